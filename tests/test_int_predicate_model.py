"""CPU: the integer form of the dispersion predicate (kernels_stream.hpp: int_predicate, used when nsig_b and nsig_s are
integers) against the oracle's float64 predicate, operation for operation (baseline/spotfinder/standalone.cc:165-170), on
random windows and on windows constructed to sit at or next to a tie -- where the integer form must either agree or say
"not certain" (the kernel then evaluates the float64 form itself).  A model of the device code in exact Python integers."""
import math

import numpy as np


def oracle(m, x, y, p, nb, ns):
    """standalone.cc:165-170 on doubles, one rounding per operation (numpy float64 does exactly that)."""
    m, x, y, p = (np.float64(v) for v in (m, x, y, p))
    a = (m * y - x * x) - x * (m - 1.0)
    b = m * p - x
    c = (x * np.float64(nb)) * np.sqrt(2.0 * (m - 1.0))
    d = np.float64(ns) * np.sqrt(x * m)
    return bool(a > c and b > d)


def int_predicate(m, x, y, p, nb, ns):
    """-> (strong, certain): the device function, in Python integers."""
    my, xx, bv = m * y, x * (x + m - 1), m * p - x
    if my <= xx or bv <= 0:
        return False, True
    av = my - xx
    c2 = nb * nb * 2 * (m - 1) * x * x
    disp_yes, disp_close = True, False
    if av < (1 << 25):
        a2 = av * av
        disp_yes, disp_close = a2 > c2, abs(a2 - c2) < 16
    b2, d2 = bv * bv, ns * ns * x * m
    return (disp_yes and b2 > d2), not (disp_close or abs(b2 - d2) < 16)


def _check(cases, nb, ns):
    n_certain = 0
    for m, x, y, p in cases:
        strong, certain = int_predicate(m, x, y, p, nb, ns)
        if certain:
            n_certain += 1
            assert strong == oracle(m, x, y, p, nb, ns), (m, x, y, p)
    return n_certain


def test_random_windows_agree():
    rng = np.random.default_rng(5)
    cases = []
    for _ in range(60000):
        m = int(rng.integers(2, 50))
        pix = rng.poisson(float(rng.choice([0.5, 2.0, 20.0, 800.0])), m)
        if rng.random() < 0.3:
            pix[rng.integers(0, m)] += int(rng.integers(10, 20000))
        x, y = int(pix.sum()), int((pix.astype(np.int64) ** 2).sum())
        if x >= 65536:
            continue
        cases.append((m, x, y, int(pix[0])))
    n = _check(cases, 6, 3)
    assert n > 0.99 * len(cases)          # ties are rare on random data


def test_windows_at_and_next_to_a_tie():
    """2 (m - 1) a perfect square (m = 3, 9, 19, 33) makes c an integer: choose y so that a = c + delta for small delta, and
    p so that b^2 is next to nsig_s^2 x m."""
    cases = []
    for nb, ns in ((6, 3), (1, 1), (2, 5)):
        for m in (3, 9, 19, 33, 49, 48, 2):
            root = math.isqrt(2 * (m - 1))
            for x in list(range(1, 400)) + [1000, 4097, 30000, 65535]:
                c_floor = (nb * x * root) if root * root == 2 * (m - 1) else math.isqrt(nb * nb * x * x * 2 * (m - 1))
                for delta in (-3, -1, 0, 1, 2, 5):
                    a_target = c_floor + delta
                    # a = m y - x (x + m - 1)  ->  y = (a + x (x + m - 1)) / m, rounded to the nearest integer window
                    y = (a_target + x * (x + m - 1) + m // 2) // m
                    if y < 0 or y >= 1 << 32:
                        continue
                    d_floor = math.isqrt(ns * ns * x * m)
                    for pd in (-1, 0, 1, 3):
                        p = (x + d_floor + pd + m - 1) // m          # b = m p - x close to d
                        if 0 <= p < 65536:
                            cases.append((m, x, y, p))
        n = _check(cases, nb, ns)
        assert n > 0

"""GPU: the fp64 square root the exact predicate uses must be correctly rounded (the oracle's is
libm's).  Exhaustive over every integer n = x*m a uint16 frame can produce (n <= 49*49*65535 <
2^28), sampled blocks up to 2^37 for the uint32 path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _host_sum(begin, end, chunk=1 << 24):
    tot = np.uint64(0)
    with np.errstate(over="ignore"):
        for b in range(begin, end, chunk):
            e = min(end, b + chunk)
            r = np.sqrt(np.arange(b, e, dtype=np.uint64).astype(np.float64))
            tot += r.view(np.uint64).sum(dtype=np.uint64)
    return int(tot)


def test_fp64_sqrt_correctly_rounded_exhaustive_u16_domain(ffs):
    ctx = ffs.Context(64, 64)
    top = 49 * 49 * 65535 + 1
    assert top < (1 << 28)
    step = 1 << 26
    for b in range(0, top, step):
        e = min(top, b + step)
        assert ctx.selftest_sqrt(b, e) == _host_sum(b, e), f"sqrt differs somewhere in [{b},{e})"


def test_fp64_sqrt_sampled_u32_domain(ffs):
    ctx = ffs.Context(64, 64)
    rng = np.random.default_rng(1)
    for _ in range(24):
        b = int(rng.integers(1 << 28, 1 << 37))
        e = b + (1 << 20)
        assert ctx.selftest_sqrt(b, e) == _host_sum(b, e)

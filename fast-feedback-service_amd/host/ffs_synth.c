/*
 * ffs_synth.c -- deterministic synthetic detector frames and masks (see
 * include/ffs_synth.h).  Pure C11 + libm's sqrt/floor/ldexp (all exactly
 * specified), so the same seed gives the same bytes on every IEEE-754 host.
 */
#include "ffs_synth.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- integer RNG ---------------------------------------------------------- */

static inline uint64_t mix64(uint64_t z) { /* splitmix64 finaliser */
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline uint64_t hash3(uint64_t a, uint64_t b, uint64_t c) {
    return mix64(mix64(mix64(a + 0x9E3779B97F4A7C15ULL) ^ (b + 0x632BE59BD9B4E019ULL))
                 ^ (c * 0xD1342543DE82EF95ULL + 1));
}

typedef struct {
    uint64_t s;
} rng_t;

static inline uint64_t rng_next(rng_t *r) {
    r->s += 0x9E3779B97F4A7C15ULL;
    return mix64(r->s);
}

static inline double rng_unit(rng_t *r) { /* [0,1) with 53 bits */
    return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0);
}

/* ---- deterministic exp (only + - * / floor ldexp) -------------------------- */

static double det_exp(double x) {
    if (x < -700.0) return 0.0;
    if (x > 700.0) x = 700.0;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    const double INV_LN2 = 1.44269504088896338700e+00;
    double kf = floor(x * INV_LN2 + 0.5);
    double r = (x - kf * LN2_HI) - kf * LN2_LO;
    /* Taylor to degree 14 on |r| <= ln2/2: error < 1e-17 relative */
    double p = 1.0;
    for (int i = 14; i >= 1; --i) p = 1.0 + p * r / (double)i;
    return ldexp(p, (int)kf);
}

/* ---- Poisson sampling -------------------------------------------------------- */

static double rng_normal(rng_t *r) { /* Irwin-Hall(12) - 6 */
    double s = 0.0;
    for (int i = 0; i < 12; ++i) s += rng_unit(r);
    return s - 6.0;
}

static uint32_t poisson(rng_t *r, double mu) {
    if (mu <= 0.0) return 0;
    if (mu < 30.0) {
        double u = rng_unit(r);
        double p = det_exp(-mu);
        double cdf = p;
        uint32_t k = 0;
        while (u > cdf && k < 1000) {
            ++k;
            p *= mu / (double)k;
            cdf += p;
        }
        return k;
    }
    double v = floor(mu + sqrt(mu) * rng_normal(r) + 0.5);
    return v < 0.0 ? 0u : (uint32_t)v;
}

/* ---- frame generator ---------------------------------------------------------- */

static inline void add_px(void *out, int pb, size_t k, uint32_t add, uint32_t maxv) {
    if (pb == 2) {
        uint16_t *o = (uint16_t *)out;
        uint64_t v = (uint64_t)o[k] + add;
        o[k] = (uint16_t)(v > maxv ? maxv : v);
    } else {
        uint32_t *o = (uint32_t *)out;
        uint64_t v = (uint64_t)o[k] + add;
        o[k] = (uint32_t)(v > maxv ? maxv : v);
    }
}

int ffs_synth_frame(const ffs_synth_params *p, uint32_t frame, void *out) {
    if (!p || !out || (p->pixel_bytes != 2 && p->pixel_bytes != 4)) return -1;
    const uint32_t W = p->width, H = p->height;
    const int pb = p->pixel_bytes;
    uint32_t maxv = p->max_value ? p->max_value : (pb == 2 ? 65535u : 0xFFFFFFFFu);
    if (pb == 2 && maxv > 65535u) maxv = 65535u;

    /* background: table inversion with 32-bit thresholds */
    enum { NTAB = 256 };
    uint32_t thr[NTAB];
    int ntab = 0;
    const int tabulated = p->background < 40.0;
    if (tabulated && p->background > 0.0) {
        double pr = det_exp(-p->background), cdf = 0.0;
        for (int k = 0; k < NTAB; ++k) {
            cdf += pr;
            double t = floor(cdf * 4294967296.0);
            thr[k] = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
            ntab = k + 1;
            if (thr[k] == 0xFFFFFFFFu) break;
            pr *= p->background / (double)(k + 1);
        }
    }
    for (uint32_t y = 0; y < H; ++y) {
        rng_t r = {hash3(p->seed, frame, y)};
        size_t k0 = (size_t)y * W;
        if (p->background <= 0.0) {
            if (pb == 2)
                memset((uint16_t *)out + k0, 0, (size_t)W * 2);
            else
                memset((uint32_t *)out + k0, 0, (size_t)W * 4);
            continue;
        }
        for (uint32_t x = 0; x < W; x += 2) {
            uint32_t v[2];
            if (tabulated) {
                uint64_t bits = rng_next(&r);
                uint32_t u[2] = {(uint32_t)bits, (uint32_t)(bits >> 32)};
                for (int j = 0; j < 2; ++j) {
                    uint32_t c = 0;
                    while (c < (uint32_t)ntab - 1 && u[j] >= thr[c]) ++c;
                    v[j] = c;
                }
            } else {
                v[0] = poisson(&r, p->background);
                v[1] = poisson(&r, p->background);
            }
            for (int j = 0; j < 2 && x + j < W; ++j) {
                uint32_t c = v[j] > maxv ? maxv : v[j];
                if (pb == 2)
                    ((uint16_t *)out)[k0 + x + j] = (uint16_t)c;
                else
                    ((uint32_t *)out)[k0 + x + j] = c;
            }
        }
    }

    /* spots */
    const int sweep = p->n_frames > 0 && p->sigma_z_max > 0.0;
    rng_t rs = {sweep ? hash3(p->seed, 0xC0FFEEULL, 0x5EEDULL)
                      : hash3(p->seed, frame, 0xC0FFEEULL)};
    for (uint32_t s = 0; s < p->n_spots; ++s) {
        double cx = rng_unit(&rs) * W;
        double cy = rng_unit(&rs) * H;
        double sg = p->sigma_min + (p->sigma_max - p->sigma_min) * rng_unit(&rs);
        double u = rng_unit(&rs);
        double peak = p->peak_min + (p->peak_max - p->peak_min) * u * u * u;
        double scale = 1.0;
        if (sweep) {
            double zc = rng_unit(&rs) * p->n_frames;
            double sz = p->sigma_z_min + (p->sigma_z_max - p->sigma_z_min) * rng_unit(&rs);
            double dz = ((double)frame + 0.5) - zc;
            scale = det_exp(-dz * dz / (2.0 * sz * sz));
        }
        if (sg <= 0.0 || peak * scale < 0.05) continue;
        rng_t rp = {hash3(p->seed ^ 0xABCDEF12345ULL, frame, s)};
        int R = (int)ceil(4.0 * sg) + 1;
        int xc = (int)floor(cx), yc = (int)floor(cy);
        for (int yy = yc - R; yy <= yc + R; ++yy) {
            if (yy < 0 || yy >= (int)H) continue;
            for (int xx = xc - R; xx <= xc + R; ++xx) {
                if (xx < 0 || xx >= (int)W) continue;
                double dx = (xx + 0.5) - cx, dy = (yy + 0.5) - cy;
                double mu = peak * scale * det_exp(-(dx * dx + dy * dy) / (2.0 * sg * sg));
                if (mu < 1e-3) continue;
                uint32_t c = poisson(&rp, mu);
                if (c) add_px(out, pb, (size_t)yy * W + xx, c, maxv);
            }
        }
    }
    return 0;
}

/* ---- masks ------------------------------------------------------------------------ */

int ffs_synth_mask_modules(uint8_t *mask, uint32_t width, uint32_t height,
                           uint32_t mod_fast, uint32_t mod_slow, uint32_t gap_fast,
                           uint32_t gap_slow) {
    if (!mask || !mod_fast || !mod_slow) return -1;
    memset(mask, 1, (size_t)width * height);
    /* horizontal gaps: rows [g*mod_slow + (g-1)*gap_slow, +gap_slow), h5read.c:1141-1146 */
    for (uint32_t y = mod_slow; y < height; y += mod_slow + gap_slow)
        for (uint32_t yy = y; yy < y + gap_slow && yy < height; ++yy)
            memset(mask + (size_t)yy * width, 0, width);
    /* vertical gaps, h5read.c:1148-1155 */
    for (uint32_t x = mod_fast; x < width; x += mod_fast + gap_fast)
        for (uint32_t y = 0; y < height; ++y)
            for (uint32_t xx = x; xx < x + gap_fast && xx < width; ++xx)
                mask[(size_t)y * width + xx] = 0;
    return 0;
}

int ffs_synth_mask_dead_pixels(uint8_t *mask, uint32_t width, uint32_t height,
                               uint64_t seed, uint32_t n_dead) {
    if (!mask) return -1;
    rng_t r = {hash3(seed, 0xDEADULL, 0)};
    uint64_t n = (uint64_t)width * height;
    for (uint32_t i = 0; i < n_dead; ++i) mask[rng_next(&r) % n] = 0;
    return 0;
}

int ffs_synth_mask_rect(uint8_t *mask, uint32_t width, uint32_t height, uint32_t x0,
                        uint32_t x1, uint32_t y0, uint32_t y1) {
    if (!mask) return -1;
    if (x1 > width) x1 = width;
    if (y1 > height) y1 = height;
    for (uint32_t y = y0; y < y1; ++y)
        for (uint32_t x = x0; x < x1; ++x) mask[(size_t)y * width + x] = 0;
    return 0;
}

/* ---- the reference's generated sample images (h5read.c:203-276) ------------------- */

/* PCG32 XSH-RR (pcg-random.org), as the reference seeds it: state = inc = 0
 * (h5read.c:189-201, :254). */
static uint32_t pcg32_step(uint64_t *state) {
    uint64_t old = *state;
    *state = old * 6364136223846793005ULL + 1ULL; /* inc | 1 with inc = 0 */
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((32u - rot) & 31u));
}

int ffs_synth_reference_sample(uint32_t n, int32_t pb, void *out) {
    enum { FAST = 4148, SLOW = 4362, MF = 1028, MS = 512, GF = 12, GS = 38, NF = 4, NS = 8 };
    if (!out || (pb != 2 && pb != 4) || n > 5) return -1;
    size_t npx = (size_t)FAST * SLOW;
    memset(out, 0, npx * (size_t)pb);
#define PUT(k, v)                                      \
    do {                                               \
        if (pb == 2)                                   \
            ((uint16_t *)out)[k] = (uint16_t)(v);      \
        else                                           \
            ((uint32_t *)out)[k] = (uint32_t)(v);      \
    } while (0)
    if (n == 1 || n == 5) { /* per module, row-major inside the module (:213-225, :252-268) */
        uint64_t st = 0;
        for (int my = 0; my < NS; ++my) {
            size_t row0 = (size_t)my * (MS + GS);
            for (int mx = 0; mx < NF; ++mx) {
                size_t col0 = (size_t)mx * (MF + GF);
                for (int row = 0; row < MS; ++row)
                    for (int x = 0; x < MF; ++x) {
                        size_t k = FAST * (row0 + row) + col0 + x;
                        if (n == 1)
                            PUT(k, 1);
                        else
                            PUT(k, pcg32_step(&st) % 10);
                    }
            }
        }
    } else if (n == 2) { /* 100 every 42 px, :226-234 */
        for (int y = 0; y < SLOW; y += 42)
            for (int x = 0; x < FAST; x += 42) PUT((size_t)y * FAST + x, 100);
    } else if (n == 3) { /* I = x, :235-242 */
        for (int y = 0; y < SLOW; ++y)
            for (int x = 0; x < FAST; ++x) PUT((size_t)y * FAST + x, x);
    } else if (n == 4) { /* I = y, :243-250 */
        for (int y = 0; y < SLOW; ++y)
            for (int x = 0; x < FAST; ++x) PUT((size_t)y * FAST + x, y);
    }
#undef PUT
    return 0;
}

#define _GNU_SOURCE
/*
 * ffs_oracle.c -- CPU restatement of the reference spot-finder hot path.
 * TEST INFRASTRUCTURE ONLY (see ffs_oracle.h).  Build with
 *   gcc -std=c11 -O2 -ffp-contract=off -fPIC -shared
 * (-O2 is the reference's RelWithDebInfo default; contraction off so that every
 * floating-point operation rounds exactly where the reference's source does).
 */
#include "ffs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void ffs_oracle_default_disp_params(ffs_oracle_disp_params *p) {
    /* baseline/spotfinder/standalone.cc:16-20 */
    p->kernel_half_x = 3;
    p->kernel_half_y = 3;
    p->min_count = 2;
    p->threshold = 0.0;
    p->nsig_b = 6.0;
    p->nsig_s = 3.0;
}

/* struct Data, standalone.cc:34-38 */
typedef struct {
    int m;
    double x;
    double y;
} sat_entry;

/* compute_sat, standalone.cc:74-105 */
static void compute_sat(sat_entry *table, const double *src, const uint8_t *mask,
                        size_t xsize, size_t ysize) {
    const double BIG = (double)(1 << 24); /* :78 */
    size_t k = 0;
    for (size_t j = 0; j < ysize; ++j) {
        int m = 0;
        double x = 0;
        double y = 0;
        for (size_t i = 0; i < xsize; ++i, ++k) {
            int mm = (mask[k] && src[k] < BIG) ? 1 : 0; /* :90 */
            m += mm;
            x += mm * src[k];
            y += mm * src[k] * src[k]; /* (mm*src)*src, :93 */
            if (j == 0) {
                table[k].m = m;
                table[k].x = x;
                table[k].y = y;
            } else {
                table[k].m = table[k - xsize].m + m;
                table[k].x = table[k - xsize].x + x;
                table[k].y = table[k - xsize].y + y;
            }
        }
    }
}

/* compute_threshold, standalone.cc:113-174 */
static void compute_threshold(const sat_entry *table, const double *src,
                              const uint8_t *mask, uint8_t *dst, size_t xsize,
                              size_t ysize, const ffs_oracle_disp_params *p) {
    const int kxsize = p->kernel_half_x;
    const int kysize = p->kernel_half_y;
    const int min_count = p->min_count;
    const double threshold = p->threshold, nsig_b = p->nsig_b, nsig_s = p->nsig_s;
    size_t k = 0;
    for (size_t j = 0; j < ysize; ++j) {
        for (size_t i = 0; i < xsize; ++i, ++k) {
            int i0 = (int)i - kxsize - 1, i1 = (int)i + kxsize; /* :126-127 */
            int j0 = (int)j - kysize - 1, j1 = (int)j + kysize;
            i1 = i1 < (int)xsize ? i1 : (int)xsize - 1;
            j1 = j1 < (int)ysize ? j1 : (int)ysize - 1;
            long k0 = (long)j0 * (long)xsize;
            long k1 = (long)j1 * (long)xsize;

            double m = 0;
            double x = 0;
            double y = 0;
            if (i0 >= 0 && j0 >= 0) { /* :139-145 */
                const sat_entry *d00 = &table[k0 + i0];
                const sat_entry *d10 = &table[k1 + i0];
                const sat_entry *d01 = &table[k0 + i1];
                m += d00->m - (d10->m + d01->m);
                x += d00->x - (d10->x + d01->x);
                y += d00->y - (d10->y + d01->y);
            } else if (i0 >= 0) { /* :146-150 */
                const sat_entry *d10 = &table[k1 + i0];
                m -= d10->m;
                x -= d10->x;
                y -= d10->y;
            } else if (j0 >= 0) { /* :151-156 */
                const sat_entry *d01 = &table[k0 + i1];
                m -= d01->m;
                x -= d01->x;
                y -= d01->y;
            }
            const sat_entry *d11 = &table[k1 + i1]; /* :157-160 */
            m += d11->m;
            x += d11->x;
            y += d11->y;

            dst[k] = 0; /* :163-171 */
            if (mask[k] && m >= min_count && x >= 0 && src[k] > threshold) {
                double a = m * y - x * x - x * (m - 1);
                double b = m * src[k] - x;
                double c = x * nsig_b * sqrt(2 * (m - 1));
                double d = nsig_s * sqrt(x * m);
                dst[k] = (a > c && b > d) ? 1 : 0;
            }
        }
    }
}

/* The reference keeps its table for the life of the object (standalone.cc:66,
 * :226-247); so does this context, so that repeated calls are timed alike. */
struct ffs_oracle_disp_ctx {
    int width, height;
    ffs_oracle_disp_params params;
    sat_entry *table;
};

ffs_oracle_disp_ctx *ffs_oracle_disp_create(int width, int height,
                                            const ffs_oracle_disp_params *p) {
    ffs_oracle_disp_ctx *c = (ffs_oracle_disp_ctx *)malloc(sizeof(*c));
    if (!c) return NULL;
    c->width = width;
    c->height = height;
    if (p)
        c->params = *p;
    else
        ffs_oracle_default_disp_params(&c->params);
    c->table = (sat_entry *)malloc((size_t)width * (size_t)height * sizeof(sat_entry));
    if (!c->table) {
        free(c);
        return NULL;
    }
    return c;
}

void ffs_oracle_disp_destroy(ffs_oracle_disp_ctx *c) {
    if (c) {
        free(c->table);
        free(c);
    }
}

/* DispersionThreshold::threshold, standalone.cc:182-196 */
int ffs_oracle_disp_run(ffs_oracle_disp_ctx *c, const double *image, const uint8_t *mask,
                        uint8_t *dst) {
    compute_sat(c->table, image, mask, (size_t)c->width, (size_t)c->height);
    compute_threshold(c->table, image, mask, dst, (size_t)c->width, (size_t)c->height,
                      &c->params);
    return 0;
}

int ffs_oracle_dispersion_f64(const double *image, const uint8_t *mask, int width,
                              int height, const ffs_oracle_disp_params *p, uint8_t *dst) {
    ffs_oracle_disp_ctx *c = ffs_oracle_disp_create(width, height, p);
    if (!c) return -1;
    int rc = ffs_oracle_disp_run(c, image, mask, dst);
    ffs_oracle_disp_destroy(c);
    return rc;
}

int ffs_oracle_dispersion_u16(const uint16_t *image, const uint8_t *mask, int width,
                              int height, const ffs_oracle_disp_params *p, uint8_t *dst) {
    size_t n = (size_t)width * (size_t)height;
    double *img = (double *)malloc(n * sizeof(double));
    if (!img) return -1;
    for (size_t i = 0; i < n; ++i) img[i] = (double)image[i]; /* spotfinder.cc:1024 */
    int rc = ffs_oracle_dispersion_f64(img, mask, width, height, p, dst);
    free(img);
    return rc;
}

int ffs_oracle_dispersion_u32(const uint32_t *image, const uint8_t *mask, int width,
                              int height, const ffs_oracle_disp_params *p, uint8_t *dst) {
    size_t n = (size_t)width * (size_t)height;
    double *img = (double *)malloc(n * sizeof(double));
    if (!img) return -1;
    for (size_t i = 0; i < n; ++i) img[i] = (double)image[i];
    int rc = ffs_oracle_dispersion_f64(img, mask, width, height, p, dst);
    free(img);
    return rc;
}

/* ---------------------------------------------------------------------------
 * Extended dispersion: DispersionExtendedThreshold, baseline/spotfinder/baseline.cpp:325-776
 * (the DIALS class; it does not compile here -- scitbx/dials headers -- so this part is a
 * restatement from the text, "parity unpinned").  threshold() at :730-761 is
 *   compute_sat(mask) -> compute_dispersion_threshold -> erode_dispersion_mask
 *   -> compute_sat(eroded mask) -> compute_final_threshold.
 * ------------------------------------------------------------------------- */

/* the SAT window walk shared by baseline.cpp:432-467 and :597-626 (identical to
 * standalone.cc:126-160 above) */
static void sat_window(const sat_entry *table, size_t i, size_t j, int kxsize, int kysize,
                       size_t xsize, size_t ysize, double *pm, double *px, double *py) {
    int i0 = (int)i - kxsize - 1, i1 = (int)i + kxsize;
    int j0 = (int)j - kysize - 1, j1 = (int)j + kysize;
    i1 = i1 < (int)xsize ? i1 : (int)xsize - 1;
    j1 = j1 < (int)ysize ? j1 : (int)ysize - 1;
    long k0 = (long)j0 * (long)xsize;
    long k1 = (long)j1 * (long)xsize;
    double m = 0, x = 0, y = 0;
    if (i0 >= 0 && j0 >= 0) {
        const sat_entry *d00 = &table[k0 + i0];
        const sat_entry *d10 = &table[k1 + i0];
        const sat_entry *d01 = &table[k0 + i1];
        m += d00->m - (d10->m + d01->m);
        x += d00->x - (d10->x + d01->x);
        y += d00->y - (d10->y + d01->y);
    } else if (i0 >= 0) {
        const sat_entry *d10 = &table[k1 + i0];
        m -= d10->m;
        x -= d10->x;
        y -= d10->y;
    } else if (j0 >= 0) {
        const sat_entry *d01 = &table[k0 + i1];
        m -= d01->m;
        x -= d01->x;
        y -= d01->y;
    }
    const sat_entry *d11 = &table[k1 + i1];
    m += d11->m;
    x += d11->x;
    y += d11->y;
    *pm = m;
    *px = x;
    *py = y;
}

/* compute_dispersion_threshold, baseline.cpp:415-475: dst = 1 where the window's index of
 * dispersion is above the background threshold ("not background"). */
static void ext_dispersion_threshold(const sat_entry *table, const double *src,
                                     const uint8_t *mask, uint8_t *dst, size_t xsize,
                                     size_t ysize, const ffs_oracle_disp_params *p,
                                     double max_valid) {
    size_t k = 0;
    for (size_t j = 0; j < ysize; ++j) {
        for (size_t i = 0; i < xsize; ++i, ++k) {
            double m, x, y;
            sat_window(table, i, j, p->kernel_half_x, p->kernel_half_y, xsize, ysize, &m, &x, &y);
            dst[k] = 0; /* :468-473 */
            if (mask[k] && m >= p->min_count && x >= 0) {
                double a = m * y - x * x - x * (m - 1);
                double c = x * p->nsig_b * sqrt(2 * (m - 1));
                dst[k] = (a > c) ? 1 : 0;
            }
            /* the device kernels' validity guard (thresholding.cu:318-325): not in baseline.cpp,
             * active only when the caller sets max_valid >= 0 */
            if (max_valid >= 0 && src[k] > max_valid) dst[k] = 0;
        }
    }
}

/* chebyshev_distance(src, value, dst): DIALS dials/algorithms/image/filter/distance.h -- not
 * under /root/reference (baseline.cpp:558 calls it).  Its published algorithm is the two-pass
 * chamfer transform below, which yields the exact Chebyshev distance to the nearest pixel equal
 * to `value`; pixels outside the image are not sources (distance height + width). */
static void chebyshev_distance(const uint8_t *src, uint8_t value, int *dst, size_t xsize,
                               size_t ysize) {
    const int max_distance = (int)(xsize + ysize);
    for (size_t j = 0; j < ysize; ++j) {
        for (size_t i = 0; i < xsize; ++i) {
            size_t k = j * xsize + i;
            if ((src[k] != 0) == (value != 0)) {
                dst[k] = 0;
            } else {
                int N = max_distance, NW = max_distance, NE = max_distance, W = max_distance;
                if (j > 0) N = dst[k - xsize];
                if (i > 0) W = dst[k - 1];
                if (j > 0 && i > 0) NW = dst[k - xsize - 1];
                if (j > 0 && i + 1 < xsize) NE = dst[k - xsize + 1];
                int a = N < NW ? N : NW, b = NE < W ? NE : W;
                dst[k] = 1 + (a < b ? a : b);
            }
        }
    }
    for (size_t j = ysize; j > 0; --j) {
        for (size_t i = xsize; i > 0; --i) {
            size_t k = (j - 1) * xsize + (i - 1);
            int S = max_distance, SE = max_distance, SW = max_distance, E = max_distance;
            if (j < ysize) S = dst[k + xsize];
            if (i < xsize) E = dst[k + 1];
            if (j < ysize && i < xsize) SE = dst[k + xsize + 1];
            if (j < ysize && i > 1) SW = dst[k + xsize - 1];
            int a = S < SE ? S : SE, b = SW < E ? SW : E;
            int d = 1 + (a < b ? a : b);
            if (d < dst[k]) dst[k] = d;
        }
    }
}

/* erode_dispersion_mask, baseline.cpp:552-571.  In: dst = 1 for "not background".  Out: dst = 1
 * for every valid pixel that counts as background in the second pass, i.e. everything except
 * the not-background pixels at Chebyshev distance >= min(kernel) = 3 from the nearest pixel that
 * is not one (masked pixels included: their dst is 0). */
static int ext_erode_baseline(const uint8_t *mask, uint8_t *dst, size_t xsize, size_t ysize,
                              const ffs_oracle_disp_params *p) {
    size_t n = xsize * ysize;
    int *distance = (int *)malloc(n * sizeof(int));
    if (!distance) return -1;
    chebyshev_distance(dst, 0, distance, xsize, ysize);
    int erosion_distance = p->kernel_half_x < p->kernel_half_y ? p->kernel_half_x : p->kernel_half_y;
    for (size_t k = 0; k < n; ++k) {
        if (mask[k])
            dst[k] = !(dst[k] && distance[k] >= erosion_distance);
        else
            dst[k] = 0;
    }
    free(distance);
    return 0;
}

/* The device flavour of the same step, spotfinder/kernels/erosion.cu:53-143: a not-background
 * pixel is given back to the background when a VALID, in-image pixel within Chebyshev distance 2
 * is background; masked neighbours are skipped (:101-105) instead of eroding.  Output in the
 * baseline's polarity (1 = background for the second pass, 0 for masked pixels). */
static int ext_erode_device(const uint8_t *mask, uint8_t *dst, size_t xsize, size_t ysize) {
    size_t n = xsize * ysize;
    uint8_t *first = (uint8_t *)malloc(n);
    if (!first) return -1;
    memcpy(first, dst, n);
    for (size_t j = 0; j < ysize; ++j)
        for (size_t i = 0; i < xsize; ++i) {
            size_t k = j * xsize + i;
            if (!first[k]) { /* :76-84 (then excluded again by the pixel mask in the second pass) */
                dst[k] = mask[k] ? 1 : 0;
                continue;
            }
            int erase = 0;
            for (int dj = -2; dj <= 2 && !erase; ++dj)
                for (int di = -2; di <= 2; ++di) {
                    long lx = (long)i + di, ly = (long)j + dj;
                    if (lx < 0 || ly < 0 || lx >= (long)xsize || ly >= (long)ysize) continue;
                    size_t kk = (size_t)ly * xsize + (size_t)lx;
                    if (mask[kk] == 0) continue;
                    if (!first[kk]) {
                        erase = 1;
                        break;
                    }
                }
            dst[k] = erase ? 1 : 0;
        }
    free(first);
    return 0;
}

/* compute_final_threshold, baseline.cpp:580-645.  table = SAT over the eroded mask (m, x only);
 * dst in: eroded mask, out: strong pixels.  kernel + 2 (:591-592). */
static void ext_final_threshold(const sat_entry *table, const double *src, const uint8_t *mask,
                                uint8_t *dst, size_t xsize, size_t ysize,
                                const ffs_oracle_disp_params *p, int flavour, double max_valid) {
    const int kxsize = p->kernel_half_x + 2, kysize = p->kernel_half_y + 2;
    size_t k = 0;
    for (size_t j = 0; j < ysize; ++j) {
        for (size_t i = 0; i < xsize; ++i, ++k) {
            double m, x, y;
            sat_window(table, i, j, kxsize, kysize, xsize, ysize, &m, &x, &y);
            (void)y;
            int ok = mask[k] && m >= 0 && x >= 0; /* :636 */
            if (flavour == 1 && !(m > 0)) ok = 0; /* thresholding.cu:472 `n > 0` */
            if (max_valid >= 0 && src[k] > max_valid) ok = 0; /* thresholding.cu:440-441 */
            if (ok) {
                int dispersion_mask = !dst[k];
                int global_mask = src[k] > p->threshold;
                double mean = (m >= 2 ? (x / m) : 0);
                int local_mask = src[k] >= (mean + p->nsig_s * sqrt(mean));
                dst[k] = dispersion_mask && global_mask && local_mask;
            } else {
                dst[k] = 0;
            }
        }
    }
}

int ffs_oracle_dispersion_extended_f64(const double *image, const uint8_t *mask, int width,
                                       int height, const ffs_oracle_disp_params *pp, int flavour,
                                       double max_valid, uint8_t *dst, uint8_t *dbg_first,
                                       uint8_t *dbg_eroded) {
    ffs_oracle_disp_params dflt;
    if (!pp) {
        ffs_oracle_default_disp_params(&dflt);
        pp = &dflt;
    }
    size_t xsize = (size_t)width, ysize = (size_t)height, n = xsize * ysize;
    sat_entry *table = (sat_entry *)malloc(n * sizeof(sat_entry));
    if (!table) return -1;
    int rc = 0;
    compute_sat(table, image, mask, xsize, ysize);                               /* :744 */
    ext_dispersion_threshold(table, image, mask, dst, xsize, ysize, pp, max_valid); /* :749 */
    if (dbg_first) memcpy(dbg_first, dst, n);
    rc = flavour == 1 ? ext_erode_device(mask, dst, xsize, ysize)
                      : ext_erode_baseline(mask, dst, xsize, ysize, pp);         /* :752 */
    if (rc == 0) {
        if (dbg_eroded) /* 1 = pixel stays in the dispersion-masked (signal) region */
            for (size_t k = 0; k < n; ++k) dbg_eroded[k] = mask[k] && !dst[k];
        compute_sat(table, image, dst, xsize, ysize);                            /* :755 */
        ext_final_threshold(table, image, mask, dst, xsize, ysize, pp, flavour, max_valid); /* :758 */
    }
    free(table);
    return rc;
}

int ffs_oracle_dispersion_extended_u16(const uint16_t *image, const uint8_t *mask, int width,
                                       int height, const ffs_oracle_disp_params *p, int flavour,
                                       double max_valid, uint8_t *dst, uint8_t *dbg_first,
                                       uint8_t *dbg_eroded) {
    size_t n = (size_t)width * (size_t)height;
    double *img = (double *)malloc(n * sizeof(double));
    if (!img) return -1;
    for (size_t i = 0; i < n; ++i) img[i] = (double)image[i];
    int rc = ffs_oracle_dispersion_extended_f64(img, mask, width, height, p, flavour, max_valid, dst,
                                                dbg_first, dbg_eroded);
    free(img);
    return rc;
}

int ffs_oracle_dispersion_extended_u32(const uint32_t *image, const uint8_t *mask, int width,
                                       int height, const ffs_oracle_disp_params *p, int flavour,
                                       double max_valid, uint8_t *dst, uint8_t *dbg_first,
                                       uint8_t *dbg_eroded) {
    size_t n = (size_t)width * (size_t)height;
    double *img = (double *)malloc(n * sizeof(double));
    if (!img) return -1;
    for (size_t i = 0; i < n; ++i) img[i] = (double)image[i];
    int rc = ffs_oracle_dispersion_extended_f64(img, mask, width, height, p, flavour, max_valid, dst,
                                                dbg_first, dbg_eroded);
    free(img);
    return rc;
}

/* ---------------------------------------------------------------------------
 * Connected components.
 *
 * The reference builds a Boost adjacency_list whose vertex ids are assigned in
 * ascending linear-index order (connected_components.cc:51-58; 3D: slice by
 * slice, :292-309) and calls boost::connected_components, a depth-first search
 * that numbers components in the order their first vertex is met when walking
 * vertex ids upward.  So: label(component) = rank of its minimum vertex id.
 * That partition + ordering is restated here with a union-find.
 * ------------------------------------------------------------------------- */

static size_t uf_find(size_t *parent, size_t v) {
    size_t r = v;
    while (parent[r] != r) r = parent[r];
    while (parent[v] != r) {
        size_t nx = parent[v];
        parent[v] = r;
        v = nx;
    }
    return r;
}

static void uf_union(size_t *parent, size_t a, size_t b) {
    size_t ra = uf_find(parent, a), rb = uf_find(parent, b);
    if (ra == rb) return;
    if (ra < rb)
        parent[rb] = ra;
    else
        parent[ra] = rb;
}

/* position of key in ascending array, or (size_t)-1 */
static size_t find_index(const uint64_t *keys, size_t n, uint64_t key) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (keys[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (lo < n && keys[lo] == key) ? lo : (size_t)-1;
}

/* Adds the 4-connectivity edges of one slice: right = k+1 (no row-end check)
 * and below = k+width, connected_components.cc:61-78. */
static void union_slice_edges(size_t *parent, size_t base, const uint64_t *k, size_t n,
                              uint32_t width) {
    for (size_t v = 0; v < n; ++v) {
        if (v + 1 < n && k[v + 1] == k[v] + 1) uf_union(parent, base + v, base + v + 1);
        size_t below = find_index(k, n, k[v] + width);
        if (below != (size_t)-1) uf_union(parent, base + v, base + below);
    }
}

/* labels[v] = component number in first-vertex order; returns count */
static size_t assign_labels(size_t *parent, size_t n, int64_t *labels) {
    int64_t *root_label = (int64_t *)malloc((n ? n : 1) * sizeof(int64_t));
    for (size_t v = 0; v < n; ++v) root_label[v] = -1;
    size_t next = 0;
    for (size_t v = 0; v < n; ++v) {
        size_t r = uf_find(parent, v);
        if (root_label[r] < 0) root_label[r] = (int64_t)next++;
        labels[v] = root_label[r];
    }
    free(root_label);
    return next;
}

int ffs_oracle_cc2d(const uint8_t *result_image, const void *pixels, int pixel_bytes,
                    uint32_t width, uint32_t height, uint32_t min_spot_size,
                    ffs_oracle_box **boxes_out, size_t *n_boxes_out,
                    size_t *n_unfiltered_out, uint32_t *num_strong_out,
                    uint32_t *num_strong_filtered_out, uint64_t **out_k,
                    uint32_t **out_intensity) {
    size_t npx = (size_t)width * height;
    /* signals, connected_components.cc:24-32 */
    size_t n = 0;
    for (size_t k = 0; k < npx; ++k)
        if (result_image[k]) ++n;
    uint64_t *ks = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    uint32_t *inten = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    size_t *parent = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
    int64_t *labels = (int64_t *)malloc((n ? n : 1) * sizeof(int64_t));
    if (!ks || !inten || !parent || !labels) return -1;
    size_t v = 0;
    for (size_t k = 0; k < npx; ++k) {
        if (result_image[k]) {
            ks[v] = k;
            inten[v] = pixel_bytes == 2 ? ((const uint16_t *)pixels)[k]
                                        : ((const uint32_t *)pixels)[k];
            parent[v] = v;
            ++v;
        }
    }
    union_slice_edges(parent, 0, ks, n, width); /* build_graph, :47-79 */
    size_t num_labels = assign_labels(parent, n, labels);

    /* generate_boxes, :87-139 */
    ffs_oracle_box *boxes =
        (ffs_oracle_box *)malloc((num_labels ? num_labels : 1) * sizeof(ffs_oracle_box));
    if (!boxes) return -1;
    for (size_t i = 0; i < num_labels; ++i) {
        boxes[i].l = width;
        boxes[i].t = height;
        boxes[i].r = 0;
        boxes[i].b = 0;
        boxes[i].num_pixels = 0;
    }
    for (v = 0; v < n; ++v) {
        uint32_t x = (uint32_t)(ks[v] % width), y = (uint32_t)(ks[v] / width);
        ffs_oracle_box *bx = &boxes[labels[v]];
        if (x < bx->l) bx->l = x;
        if (x > bx->r) bx->r = x;
        if (y < bx->t) bx->t = y;
        if (y > bx->b) bx->b = y;
        ++bx->num_pixels;
    }
    uint32_t nsf = 0;
    size_t kept = num_labels;
    if (min_spot_size > 0) { /* :122-135 */
        kept = 0;
        for (size_t i = 0; i < num_labels; ++i) {
            if ((uint32_t)boxes[i].num_pixels >= min_spot_size) {
                boxes[kept++] = boxes[i];
                nsf += (uint32_t)boxes[i].num_pixels;
            }
        }
    } else {
        nsf = (uint32_t)n; /* :137 */
    }
    *boxes_out = boxes;
    *n_boxes_out = kept;
    if (n_unfiltered_out) *n_unfiltered_out = num_labels;
    if (num_strong_out) *num_strong_out = (uint32_t)n;
    if (num_strong_filtered_out) *num_strong_filtered_out = nsf;
    if (out_k)
        *out_k = ks;
    else
        free(ks);
    if (out_intensity)
        *out_intensity = inten;
    else
        free(inten);
    free(parent);
    free(labels);
    return 0;
}

typedef struct {
    /* Reflection3D state, connected_components.hpp:34-42,254-259 */
    uint32_t x_min, x_max, y_min, y_max;
    int z_min, z_max;
    int num_pixels;
    /* center_of_mass accumulators, :81-90 (double, in signal order) */
    double wx, wy, wz, total;
    /* peak search state, :122-170 */
    int have_peak;
    double max_intensity;
    uint32_t px, py;
    int pz;
    uint32_t pint;
} refl_acc;

static int cc3d_impl(const ffs_oracle_slice *slices, size_t n_slices, uint32_t width,
                     uint32_t height, uint32_t min_spot_size,
                     float max_peak_centroid_separation, ffs_oracle_reflection **out,
                     size_t *n_out, size_t *n_calculated, size_t *n_filtered_size_out,
                     size_t *n_filtered_sep_out, int32_t *signal_reflection) {
    (void)height;
    size_t total = 0;
    for (size_t s = 0; s < n_slices; ++s) total += slices[s].n;
    size_t *base = (size_t *)malloc((n_slices + 1) * sizeof(size_t));
    size_t *parent = (size_t *)malloc((total ? total : 1) * sizeof(size_t));
    int64_t *labels = (int64_t *)malloc((total ? total : 1) * sizeof(int64_t));
    if (!base || !parent || !labels) return -1;
    /* global vertex ids slice by slice, connected_components.cc:292-309 */
    base[0] = 0;
    for (size_t s = 0; s < n_slices; ++s) base[s + 1] = base[s] + slices[s].n;
    for (size_t v = 0; v < total; ++v) parent[v] = v;
    /* in-plane edges, :316-343 */
    for (size_t s = 0; s < n_slices; ++s)
        union_slice_edges(parent, base[s], slices[s].linear_index, slices[s].n, width);
    /* inter-slice edges: same linear index in slice i and i+1, :352-370 */
    for (size_t s = 0; s + 1 < n_slices; ++s) {
        const ffs_oracle_slice *a = &slices[s], *b = &slices[s + 1];
        for (size_t v = 0; v < a->n; ++v) {
            size_t j = find_index(b->linear_index, b->n, a->linear_index[v]);
            if (j != (size_t)-1) uf_union(parent, base[s] + v, base[s + 1] + j);
        }
    }
    size_t num_labels = assign_labels(parent, total, labels); /* :381-384 */

    refl_acc *acc = (refl_acc *)calloc(num_labels ? num_labels : 1, sizeof(refl_acc));
    if (!acc) return -1;
    for (size_t i = 0; i < num_labels; ++i) {
        acc[i].x_min = UINT32_MAX; /* hpp:35-40 */
        acc[i].x_max = 0;
        acc[i].y_min = UINT32_MAX;
        acc[i].y_max = 0;
        acc[i].z_min = INT32_MAX;
        acc[i].z_max = INT32_MIN;
        acc[i].max_intensity = 2.2250738585072014e-308; /* numeric_limits<double>::min(), hpp:122 */
    }
    /* walk slices in z order, add_signal, cc:409-446 / hpp:43-64 */
    for (size_t z = 0; z < n_slices; ++z) {
        for (size_t v = 0; v < slices[z].n; ++v) {
            refl_acc *r = &acc[labels[base[z] + v]];
            uint64_t k = slices[z].linear_index[v];
            uint32_t x = (uint32_t)(k % width), y = (uint32_t)(k / width);
            uint32_t inten = slices[z].intensity[v];
            int zi = (int)z;
            if (zi < r->z_min) r->z_min = zi;
            if (zi > r->z_max) r->z_max = zi;
            if (x < r->x_min) r->x_min = x;
            if (x > r->x_max) r->x_max = x;
            if (y < r->y_min) r->y_min = y;
            if (y > r->y_max) r->y_max = y;
            ++r->num_pixels;
            /* center_of_mass, hpp:86-90 */
            r->wx += ((double)x + 0.5) * inten;
            r->wy += ((double)y + 0.5) * inten;
            r->wz += (zi + 0.5) * inten;
            r->total += inten;
            /* peak search, hpp:125-170; is_signal_preferred cc:143-157 */
            double di = (double)inten;
            if (di < r->max_intensity) continue;
            if (di > r->max_intensity) {
                r->max_intensity = di;
                r->have_peak = 1;
                r->px = x;
                r->py = y;
                r->pz = zi;
                r->pint = inten;
                continue;
            }
            if (!r->have_peak) continue;
            int preferred;
            if (zi != r->pz)
                preferred = zi < r->pz;
            else if (y != r->py)
                preferred = y < r->py;
            else
                preferred = x < r->px;
            if (preferred) {
                r->px = x;
                r->py = y;
                r->pz = zi;
                r->pint = inten;
            }
        }
    }

    ffs_oracle_reflection *res = (ffs_oracle_reflection *)malloc(
        (num_labels ? num_labels : 1) * sizeof(ffs_oracle_reflection));
    if (!res) return -1;
    size_t kept = 0, f_size = 0, f_sep = 0;
    for (size_t i = 0; i < num_labels; ++i) {
        refl_acc *r = &acc[i];
        ffs_oracle_reflection o;
        memset(&o, 0, sizeof(o));
        o.x_min = r->x_min;
        o.x_max = r->x_max;
        o.y_min = r->y_min;
        o.y_max = r->y_max;
        o.z_min = r->z_min;
        o.z_max = r->z_max;
        o.num_pixels = r->num_pixels;
        o.sum_intensity = (uint64_t)r->total;
        /* hpp:98-100: double quotient narrowed into tuple<float,float,float> */
        o.com_x = (float)(r->wx / r->total);
        o.com_y = (float)(r->wy / r->total);
        o.com_z = (float)(r->wz / r->total);
        o.peak_x = r->px;
        o.peak_y = r->py;
        o.peak_z = r->pz;
        o.peak_intensity = r->pint;
        /* hpp:194-198, all float */
        float dx = ((float)r->px + 0.5f) - o.com_x;
        float dy = ((float)r->py + 0.5f) - o.com_y;
        float dz = (r->pz + 0.5f) - o.com_z;
        o.peak_centroid_distance = sqrtf(dx * dx + dy * dy + dz * dz);
        /* filter_reflections, cc:207-236: size first, then separation */
        if (min_spot_size > 0 && (uint32_t)o.num_pixels < min_spot_size) {
            ++f_size;
            continue;
        }
        if (max_peak_centroid_separation > 0
            && o.peak_centroid_distance > max_peak_centroid_separation) {
            ++f_sep;
            continue;
        }
        acc[i].num_pixels = -(int)kept - 1; /* reuse: -(index of the kept reflection) - 1 */
        res[kept++] = o;
    }
    if (signal_reflection) /* membership of every signal, in vertex order (Reflection3D::signals_) */
        for (size_t v = 0; v < total; ++v) {
            int np = acc[labels[v]].num_pixels;
            signal_reflection[v] = np < 0 ? -np - 1 : -1;
        }
    *out = res;
    *n_out = kept;
    if (n_calculated) *n_calculated = num_labels;
    if (n_filtered_size_out) *n_filtered_size_out = f_size;
    if (n_filtered_sep_out) *n_filtered_sep_out = f_sep;
    free(acc);
    free(base);
    free(parent);
    free(labels);
    return 0;
}

int ffs_oracle_cc3d(const ffs_oracle_slice *slices, size_t n_slices, uint32_t width,
                    uint32_t height, uint32_t min_spot_size,
                    float max_peak_centroid_separation, ffs_oracle_reflection **out,
                    size_t *n_out, size_t *n_calculated, size_t *n_filtered_size_out,
                    size_t *n_filtered_sep_out) {
    return cc3d_impl(slices, n_slices, width, height, min_spot_size, max_peak_centroid_separation,
                     out, n_out, n_calculated, n_filtered_size_out, n_filtered_sep_out, NULL);
}

int ffs_oracle_cc3d_signals(const ffs_oracle_slice *slices, size_t n_slices, uint32_t width,
                            uint32_t height, uint32_t min_spot_size,
                            float max_peak_centroid_separation, int32_t *signal_reflection) {
    ffs_oracle_reflection *tmp = NULL;
    size_t n = 0;
    int rc = cc3d_impl(slices, n_slices, width, height, min_spot_size,
                       max_peak_centroid_separation, &tmp, &n, NULL, NULL, NULL, signal_reflection);
    free(tmp);
    return rc;
}

/* Spot variances in Kabsch space: the loop of spotfinder/spotfinder.cc:1152-1215 around
 * Reflection3D::variances_in_kabsch_space (connected_components.cc:159-203).
 * Panel / Scan come from the dx2 submodule, which is not under /root/reference ("parity unpinned");
 * restated here is the flat single panel its (distance, beam centre, pixel size, image size)
 * constructor describes: fast axis +x, slow axis -y, origin (-bx px, +by py, -distance) in mm,
 * px_to_mm = pixel * pixel_size (no parallax correction), lab = origin + x fast + y slow. */
static void panel_lab(const ffs_oracle_geometry *g, double xpx, double ypx, double s[3]) {
    double xmm = xpx * g->pixel_size_x_mm, ymm = ypx * g->pixel_size_y_mm;
    s[0] = -g->beam_center_x_px * g->pixel_size_x_mm + xmm;
    s[1] = g->beam_center_y_px * g->pixel_size_y_mm - ymm;
    s[2] = -g->distance_mm;
}
static void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void normalize3(double a[3]) {
    double n = sqrt(dot3(a, a));
    a[0] /= n;
    a[1] /= n;
    a[2] /= n;
}

int ffs_oracle_kabsch_variances(const ffs_oracle_slice *slices, size_t n_slices, uint32_t width,
                                const int32_t *signal_reflection,
                                const ffs_oracle_reflection *refl, size_t n_refl,
                                const ffs_oracle_geometry *g, double *sigma_b_variance,
                                double *sigma_m_variance, int32_t *bbox_depth) {
    const double deg_to_rad = M_PI / 180.0;
    const int image_range_0 = 1; /* Scan({1, num_images}, ...), spotfinder.cc:1164-1166 */
    double *acc = (double *)calloc(n_refl ? n_refl * 4 : 1, sizeof(double));
    double *frame = (double *)malloc((n_refl ? n_refl : 1) * 12 * sizeof(double));
    if (!acc || !frame) return -1;
    const double s0[3] = {0.0, 0.0, -1.0 / g->wavelength};
    const double m2[3] = {1.0, 0.0, 0.0};
    for (size_t r = 0; r < n_refl; ++r) { /* spotfinder.cc:1187-1194, cc.cc:166-175 */
        double *f = frame + r * 12; /* s1[3] e1[3] e2[3] mags1 zeta phi */
        panel_lab(g, (double)refl[r].com_x, (double)refl[r].com_y, f);
        cross3(f, s0, f + 3);
        normalize3(f + 3);
        cross3(f, f + 3, f + 6);
        normalize3(f + 6);
        f[9] = sqrt(dot3(f, f));
        f[10] = dot3(m2, f + 3);
        f[11] = (g->oscillation_start + ((double)refl[r].com_z - image_range_0) * g->oscillation_width) * deg_to_rad;
    }
    size_t v = 0;
    for (size_t z = 0; z < n_slices; ++z)
        for (size_t i = 0; i < slices[z].n; ++i, ++v) { /* cc.cc:177-193, signals in vertex order */
            int32_t r = signal_reflection[v];
            if (r < 0) continue;
            const double *f = frame + (size_t)r * 12;
            uint64_t k = slices[z].linear_index[i];
            double x = (double)(k % width) + 0.5, y = (double)(k / width) + 0.5, zz = (double)z + 0.5;
            double s1p[3], d[3];
            panel_lab(g, x, y, s1p);
            d[0] = s1p[0] - f[0];
            d[1] = s1p[1] - f[1];
            d[2] = s1p[2] - f[2];
            double eps1 = dot3(f + 3, d) / f[9];
            double eps2 = dot3(f + 6, d) / f[9];
            double phi_dash = (g->oscillation_start + (zz - image_range_0) * g->oscillation_width) * deg_to_rad;
            double eps3 = (phi_dash - f[11]) * f[10];
            double inten = (double)slices[z].intensity[i];
            acc[4 * r + 0] += inten * eps1 * eps1;
            acc[4 * r + 1] += inten * eps2 * eps2;
            acc[4 * r + 2] += inten * eps3 * eps3;
            acc[4 * r + 3] += inten;
        }
    for (size_t r = 0; r < n_refl; ++r) { /* cc.cc:194-199 */
        double varx = acc[4 * r] / acc[4 * r + 3], vary = acc[4 * r + 1] / acc[4 * r + 3];
        double varz = acc[4 * r + 2] / acc[4 * r + 3];
        sigma_b_variance[r] = (varx + vary) / 2.0;
        sigma_m_variance[r] = varz;
        bbox_depth[r] = refl[r].z_max - refl[r].z_min + 1;
    }
    free(acc);
    free(frame);
    return 0;
}

/* Resolution mask: spotfinder/kernels/masking.cu:37-73 (distance from the beam centre, d-spacing), :99-147
 * (pixels outside [dmin, dmax] are masked; masked pixels stay masked), all in float32 as the kernel writes it.
 * `0.5 * atanf(..)` is evaluated in double there and narrowed: halving is exact, so 0.5f * gives the same float.
 * The transcendental functions are the host libm's (correctly rounded or within an ulp); a device library may
 * differ from them in the last place, which can flip pixels whose d-spacing sits on a threshold -- the GPU
 * test states and bounds those.  `resolution` (optional, W*H floats) receives the d-spacing of every pixel. */
int ffs_oracle_resolution_mask(uint8_t *mask, int width, int height, float wavelength, float distance,
                               float beam_center_x, float beam_center_y, float pixel_size_x, float pixel_size_y,
                               float dmin, float dmax, float *resolution) {
    if (!mask || width <= 0 || height <= 0) return -1;
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const size_t k = (size_t)y * width + x;
            const float dx = (((float)x + 0.5f) - beam_center_x) * pixel_size_x; /* :50-52 */
            const float dy = (((float)y + 0.5f) - beam_center_y) * pixel_size_y;
            const float r = sqrtf(dx * dx + dy * dy);
            const float theta = 0.5f * atanf(r / distance);                      /* :71 */
            const float res = wavelength / (2 * sinf(theta));                    /* :72 */
            if (resolution) resolution[k] = res;
            if (mask[k] == 0) continue;                                           /* :120-126 */
            if (dmin > 0 && res < dmin) { mask[k] = 0; continue; }                /* :133-136 */
            if (dmax > 0 && res > dmax) { mask[k] = 0; continue; }                /* :139-142 */
            mask[k] = 1;                                                          /* :145 */
        }
    return 0;
}

void ffs_oracle_free(void *p) {
    free(p);
}

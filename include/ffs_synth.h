/*
 * ffs_synth.h -- deterministic synthetic detector frames and masks.
 *
 * The reference's only synthetic source is h5read_generate_samples()
 * (h5read/src/h5read.c:203-276: six fixed Eiger-16M frames, PCG32 at :189-201;
 * module-gap mask at :1131-1156).  This generator produces the workloads
 * BASELINE.json / SURVEY.md section 8(d) name (Poisson background + Gaussian
 * spots, optional rocking curve across a sweep, u16 or u32 pixels) from a seed,
 * with integer RNG and home-grown exp so that the same bytes come out on every
 * machine.  It also restates the reference's sample images 0..5 and its
 * Eiger-2XE-16M module-gap mask so the reference's own synthetic cases can be
 * run through the pipeline.
 */
#ifndef FFS_SYNTH_H
#define FFS_SYNTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t width;       /* fast axis */
    uint32_t height;      /* slow axis */
    int32_t pixel_bytes;  /* 2 (uint16) or 4 (uint32) */
    uint64_t seed;        /* dataset seed */
    double background;    /* Poisson mean per pixel */
    uint32_t n_spots;     /* spots per frame (stills) or per sweep (rotation) */
    double sigma_min, sigma_max; /* in-plane Gaussian sigma, px */
    double peak_min, peak_max;   /* expected peak counts, log-uniform */
    uint32_t max_value;   /* clamp (e.g. 65535) */
    /* rotation sweep: if n_frames > 0 and sigma_z_max > 0 the spot list is
     * drawn once per dataset and every spot gets a centre frame and a Gaussian
     * rocking width; otherwise spots are redrawn for every frame. */
    uint32_t n_frames;
    double sigma_z_min, sigma_z_max;
} ffs_synth_params;

/* Fill `out` (width*height pixels of pixel_bytes) with frame number `frame`. */
int ffs_synth_frame(const ffs_synth_params *p, uint32_t frame, void *out);

/* Module-gap mask (1 = valid): n_fast x n_slow modules of mod_fast x mod_slow
 * pixels separated by gap_fast / gap_slow masked pixels.
 * Eiger 2XE 16M = (1028, 512, 12, 38, 4, 8), h5read/include/eiger2xe.h:6-19. */
int ffs_synth_mask_modules(uint8_t *mask, uint32_t width, uint32_t height,
                           uint32_t mod_fast, uint32_t mod_slow, uint32_t gap_fast,
                           uint32_t gap_slow);

/* Knock out `n_dead` pseudo-random pixels (sets them to 0). */
int ffs_synth_mask_dead_pixels(uint8_t *mask, uint32_t width, uint32_t height,
                               uint64_t seed, uint32_t n_dead);

/* Mask a rectangle [x0,x1) x [y0,y1). */
int ffs_synth_mask_rect(uint8_t *mask, uint32_t width, uint32_t height, uint32_t x0,
                        uint32_t x1, uint32_t y0, uint32_t y1);

/* The reference's generated sample images n = 0..5 (h5read.c:203-276), always
 * Eiger-2XE-16M shaped (4148 x 4362), written as pixel_bytes-wide pixels. */
int ffs_synth_reference_sample(uint32_t n, int32_t pixel_bytes, void *out);

#ifdef __cplusplus
}
#endif
#endif

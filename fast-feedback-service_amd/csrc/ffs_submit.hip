// ffs_submit.hip -- everything that puts a batch on the device: launch geometry, the launches of the threshold stage
// and of the sparse stage (compaction -> connected components -> records), the submit entry points and compressed
// input.  All kernels of the hot path are included here and nowhere else (see ffs_internal.hpp).
//
// Reference: the per-frame section of spotfinder/spotfinder.cc:751-1008 (H2D, kernel launch wrapper
// spotfinder/spotfinder.cu:148-189, D2H of the mask, host connected components).
#include "ffs_internal.hpp"
#include "kernels_threshold.hpp"
#include "kernels_extended.hpp"
#include "kernels_stream.hpp"
#include "kernels_ccl.hpp"
#include "kernels_chain.hpp"
#include "kernels_band.hpp"
#include "kernels_decode.hpp"

bool chain_prepare_device() {   // more than 64 KB of dynamic LDS has to be asked for, per device
    const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frame_chain<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, kChainDynBytes);
    const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frame_chain<uint32_t>), hipFuncAttributeMaxDynamicSharedMemorySize, kChainDynBytes);
    const hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frame_chain<uint16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kChainDynBytes);
    const hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frame_chain<uint16_t, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kChainDynBytes);
    const hipError_t e5 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frame_chain<uint32_t, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kChainDynBytes);
    (void)hipGetLastError();
    return e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess && e4 == hipSuccess && e5 == hipSuccess;
}

ThresholdArgs make_threshold_args(ffs_stream* s, const void* img, size_t pitch, size_t fstride, uint32_t n_frames) {
    const ffs_ctx* c = s->ctx;
    const Layout& L = c->L;
    const ffs_params& p = s->batch_params;
    ThresholdArgs a{};
    a.image = img;
    a.frame_stride = fstride;
    a.pitch = (uint32_t)pitch;
    a.maskbits = c->d_maskbits;
    a.bits = s->d_bits;
    a.strong_bytes = s->d_sbytes;
    a.tile_counts = s->d_tile_counts;
    a.W = L.W;
    a.H = L.H;
    a.pitch_px = L.pitch_px;
    a.mpitch = L.mpitch;
    a.bpitch = L.bpitch;
    a.plane_frame_stride = L.plane_frame_stride;
    a.bytes_frame_stride = L.bytes_frame_stride;
    a.n_tiles = c->n_tiles;
    a.kS = (float)(p.nsig_s * p.nsig_s * (1.0 - 1.0 / 65536.0));
    a.kB = (float)(p.nsig_b * (1.0 - 1.0 / 1048576.0));
    a.min_count = p.min_count;
    a.nsig_b = p.nsig_b;
    a.nsig_s = p.nsig_s;
    a.nsig_b2 = p.nsig_b * p.nsig_b;
    a.nsig_s2 = p.nsig_s * p.nsig_s;
    a.threshold = p.threshold;
    a.max_valid = p.max_valid;
    {
        const double b2 = p.nsig_b * p.nsig_b, s2 = p.nsig_s * p.nsig_s;
        a.int_pred = (b2 == std::floor(b2) && s2 == std::floor(s2) && b2 <= 1024.0 && s2 <= 1024.0 && p.threshold < 2147483648.0
                      && std::sqrt(b2) == p.nsig_b && std::sqrt(s2) == p.nsig_s) ? 1 : 0;   // (integer nsig: the squares are exact)
        a.ib2 = a.int_pred ? (uint32_t)b2 : 0u;
        a.is2 = a.int_pred ? (uint32_t)s2 : 0u;
        a.thr_floor = a.int_pred ? (uint32_t)std::floor(p.threshold) : 0u;
    }
    // bright windows (sum p >= 65536; 32-bit pixels >= 2^24): onto the list k_bright_fix works off, or -- tuning
    // "threshold_path" = 1, and whenever that list overflowed (ffs_wait re-runs the batch) -- into the plane as candidates
    a.bright_to_plane = s->force_path >= 0 ? s->force_path : c->tune.threshold_path;
    a.overflow = s->d_overflow;
    a.bright_n = s->d_tile_counts + tile_counts_bytes(s) / 4 - 1;
    a.bright_list = s->d_bright;
    a.bright_cap = std::min<uint32_t>(kBrightCap, (uint32_t)c->tune.bright_cap);
    a.occ = s->d_occ;
    a.occ_frame_words = occ_frame_words(L);
    a.occ_spr = L.mpitch / 16;
    // The byte mask (the reference kernel's result_strong, 1 byte per pixel) is an OUTPUT only when it was asked for
    // (want_strong_mask: --writeout, parity tests): the hot path's own strong mask is the bit plane, and the 0.58 GB of
    // zeros per 32 Eiger frames cost the streaming kernel 15 %.  (The exact kernel of path 1 sets its 1s: zero-filled then too.)
    a.dense_mask = (p.want_strong_mask || c->tune.dense_mask || a.bright_to_plane) ? 1 : 0;
#ifdef FFS_EXPERIMENTS
    a.dbg = c->tune.exp.k1_debug;
    if (a.dbg & 16) a.dense_mask = 1;
    if (a.dbg & 8) a.dense_mask = 0;
#endif
    a.dbg_prio = c->tune.stream_prio;
    a.ginfo = c->d_ginfo;
    a.mmap = c->d_mmap;
    a.gpitch = (uint32_t)L.pitch_px * (uint32_t)c->pixel_bytes / 4;
    a.gpf = c->pixel_bytes == 2 ? (L.W + 7) / 8 : (L.W + 3) / 4;
    a.n_frames = (int)n_frames;
    {   // Streaming kernels: frames side by side in one super row, as many as keep every buffer of the group below 2 GiB.
        // Bands: enough waves to fill the 256 CUs several times over, bands no shorter than 72 rows (the 6-row warm-up of
        // every band stays below 8 %).  The default stays a multiple of eight bands (what rounds 1-4's round-robin map, band = xcd + 8 k,
        // needed: with 29 bands three XCDs had a band less to do than the others, 511 us per 32 Eiger frames against 430-440 with 48
        // or 56); the unit map of round 5 (ffs_device.h, stream_unit) balances any number -- tuning "stream_bands".
        const uint64_t per_frame = std::max<uint64_t>(fstride, L.bytes_frame_stride);
        a.group_frames = (int)std::max<uint64_t>(1, std::min<uint64_t>(n_frames, ((1ull << 31) - 1) / per_frame));
        a.group_frames = std::min(a.group_frames, c->tune.frames_per_group);
        const int n_groups = ((int)n_frames + a.group_frames - 1) / a.group_frames;
        const long long lanes = (long long)a.group_frames * (a.gpf + 1);
        const long long lines = (long long)a.group_frames * (L.bpitch / 128);  // byte-mask lines to zero per row
        const int lines_per_wave = c->pixel_bytes == 2 ? 4 : 2;  // a wave zero-fills 512 / 256 bytes of the byte mask per row
        a.n_strips = (int)std::max<long long>((lanes + kSOwned - 1) / kSOwned, (lines + lines_per_wave - 1) / lines_per_wave);
        const long long per_band = std::max<long long>(1, (long long)a.n_strips * n_groups);
        long long nb = std::max<long long>(1, std::min<long long>(c->tune.target_waves / per_band, L.H / 72));
        if (nb >= 8) nb = nb / 8 * 8;
        // tuning "stream_bands" > 0: that many bands (any number: the unit map balances the XCDs)
        if (c->tune.stream_bands > 0) nb = std::max<long long>(1, std::min<long long>(c->tune.stream_bands, L.H / 8));
        a.band_rows = (int)std::min<long long>(1024, (L.H + nb - 1) / nb);
        a.n_bands = (L.H + a.band_rows - 1) / a.band_rows;
        a.band_rows2 = a.band_rows;
        a.band_split = a.n_bands;
        // Tapered bands (tuning "band_taper" = t per cent, 0 = off): the last two bands of every XCD are t % as tall as the others.
        // The waves of a launch all take about the same time and there are 3-4 times as many of them as the machine has slots, so
        // the last round leaves slots idle; handed out last and short, the final waves fill that tail with less work each.
        const int taper = c->tune.band_taper;
        if (taper > 0 && taper < 100 && nb >= 32 && nb % 8 == 0 && a.n_bands == (int)nb) {
            const int K = (int)nb / 8;                       // bands per XCD
            const int K2 = 2, K1 = K - K2;
            // 8 (K1 h1 + K2 h2) >= H with h2 = taper h1 / 100
            const double h1f = (double)L.H / (8.0 * (K1 + K2 * taper / 100.0));
            int h1 = std::min(1024, (int)std::ceil(h1f));
            int h2 = (int)std::ceil((L.H / 8.0 - (double)K1 * h1) / K2);
            while (h2 < 24) { --h1; h2 = (int)std::ceil((L.H / 8.0 - (double)K1 * h1) / K2); }
            if (h1 >= h2 && h2 >= 24 && 8 * (K1 * h1 + K2 * h2) >= L.H && 8 * K1 * h1 < L.H) {
                a.band_rows = h1;
                a.band_rows2 = h2;
                a.band_split = 8 * K1;
                a.n_bands = a.band_split + (L.H - a.band_split * h1 + h2 - 1) / h2;
            }
        }
    }
    a.dplane = s->d_dplane;
    a.eplane = s->d_eplane;
    a.eplane_clean = s->ext_e_clean ? 1 : 0;
    a.ext_flavour = p.extended_flavour;
    a.ext_variant = (c->pixel_bytes == 2 && s->force_path < 0) ? c->tune.ext_first_pass : 0;
    a.ext_strips = (L.pitch_px + kExtOwnedPx - 1) / kExtOwnedPx;
    {   // one pixel per lane: bands of 64..256 rows keep the 6-row warm-up below 10 %
        const long long ext_target = 8192;
        long long er = ((long long)L.H * a.ext_strips * n_frames + 4 * ext_target - 1) / (4 * ext_target);
        er = std::max<long long>(64, std::min<long long>(er, 256));
        a.ext_band_rows = (int)er;
        a.ext_bands = (L.H + a.ext_band_rows - 1) / a.ext_band_rows;
    }
    return a;
}

// ---- the threshold stage's launches -----------------------------------------------------------------------------
static dim3 stream_grid(const ThresholdArgs& a, uint32_t n_frames) {
    const unsigned n_groups = (n_frames + (unsigned)a.group_frames - 1) / (unsigned)a.group_frames;
    return dim3(8u * stream_chunk(a), n_groups);   // (the units in eight equal chunks, one per XCD: ffs_device.h, stream_unit)
}
// logs of a launch: one per (super row, band, strip) -- log_slot()
static size_t stream_log_slots(const ThresholdArgs& a, uint32_t n_frames) {
    const unsigned n_groups = (n_frames + (unsigned)a.group_frames - 1) / (unsigned)a.group_frames;
    return (size_t)n_groups * (size_t)a.n_bands * (size_t)a.n_strips;
}

// The whole standard threshold in one kernel: final strong plane + per-tile counts (atomics into zeroed counters).
// Start and stop events ride on the dispatch itself (its completion signal): no marker packets around it.
static void launch_stream(ffs_stream* s, const ThresholdArgs& a, uint32_t n_frames, hipEvent_t start, hipEvent_t stop, hipStream_t st = nullptr) {
    const dim3 grid = stream_grid(a, n_frames);
    if (!st) st = s->st;
    if (s->ctx->pixel_bytes == 4 && a.dense_mask) hipExtLaunchKernelGGL((k_stream_u32<2, true>), grid, dim3(64), 0, st, start, stop, 0, a);
    else if (s->ctx->pixel_bytes == 4 && s->ctx->tune.rows_ahead >= 3) hipExtLaunchKernelGGL((k_stream_u32<3, false>), grid, dim3(64), 0, st, start, stop, 0, a);
    else if (s->ctx->pixel_bytes == 4) hipExtLaunchKernelGGL((k_stream_u32<2, false>), grid, dim3(64), 0, st, start, stop, 0, a);
    else if (a.dense_mask) hipExtLaunchKernelGGL((k_stream_u16<2, false, true>), grid, dim3(64), 0, st, start, stop, 0, a);
    else if (s->ctx->tune.rows_ahead == 3) hipExtLaunchKernelGGL((k_stream_u16<3, false, false>), grid, dim3(64), 0, st, start, stop, 0, a);
    else if (s->ctx->tune.rows_ahead >= 4) hipExtLaunchKernelGGL((k_stream_u16<4, false, false>), grid, dim3(64), 0, st, start, stop, 0, a);
    else hipExtLaunchKernelGGL((k_stream_u16<2, false, false>), grid, dim3(64), 0, st, start, stop, 0, a);
}
static void launch_bright_fix(ffs_stream* s, const ThresholdArgs& a, hipStream_t st) {
    if (s->ctx->pixel_bytes == 4) hipLaunchKernelGGL(k_bright_fix<uint32_t>, dim3(32), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_bright_fix<uint16_t>, dim3(32), dim3(256), 0, st, a);
}
// path 1: every pixel marked in the plane (decided strong pixels and bright-window candidates alike) takes the gathered
// predicate; rewrites the plane, the per-tile counts and sets the byte mask's 1s
static void launch_exact(ffs_stream* s, const ThresholdArgs& a, uint32_t n_frames, hipStream_t st) {
    const dim3 grid((unsigned)a.n_tiles, n_frames);
    if (s->ctx->pixel_bytes == 4) hipLaunchKernelGGL(k_exact<uint32_t>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_exact<uint16_t>, grid, dim3(256), 0, st, a);
}

// Extended dispersion: first pass -> erosion -> final threshold (kernels_extended.hpp).  Leaves the strong plane in
// a.bits, the byte mask and the per-tile counts as the exact stage does.  First pass, 16-bit pixels: the streaming kernel
// in its extended mode decides it exactly in its drain (ext_variant 2, default); k_ext_first is the plain one-pixel-
// per-lane kernel that computes the same plane directly (32-bit pixels, tuning "ext_first_pass" = 0, and the fall-back
// when the bright-window list of a batch overflowed).
static bool ext_stream_first(const ThresholdArgs& a) { return a.ext_variant >= 2; }

static void launch_ext_first(ffs_stream* s, const ThresholdArgs& a, uint32_t n_frames, hipEvent_t start, hipEvent_t stop, bool fix_here = true,
                             bool plane_clean = false, bool counts_clean = false) {
    if (ext_stream_first(a)) {
        // the kernel writes the non-zero bytes of the first-pass plane; the bright-list count sits behind the tile counts
        // (both are usually clean already: the plane was cleared behind the previous batch's sparse launch, which also zeroed the counts)
        if (!plane_clean) (void)hipMemsetAsync(a.dplane, 0, (size_t)n_frames * a.plane_frame_stride, s->st);
        if (!counts_clean) (void)hipMemsetAsync(a.tile_counts, 0, tile_counts_bytes(s), s->st);
        if (a.dense_mask) hipExtLaunchKernelGGL((k_stream_u16<2, true, true>), stream_grid(a, n_frames), dim3(64), 0, s->st, start, stop, 0, a);
        else hipExtLaunchKernelGGL((k_stream_u16<2, true, false>), stream_grid(a, n_frames), dim3(64), 0, s->st, start, stop, 0, a);
        if (fix_here) hipLaunchKernelGGL((k_bright_fix<uint16_t, true>), dim3(32), dim3(256), 0, s->st, a);
        return;
    }
    dim3 g1((unsigned)(a.ext_strips * a.ext_bands), n_frames);
    if (s->ctx->pixel_bytes == 2) hipExtLaunchKernelGGL(k_ext_first<uint16_t>, g1, dim3(64), 0, s->st, start, stop, 0, a);
    else hipExtLaunchKernelGGL(k_ext_first<uint32_t>, g1, dim3(64), 0, s->st, start, stop, 0, a);
}
static void launch_ext_rest(ffs_stream* s, const ThresholdArgs& a, uint32_t n_frames, hipStream_t st) {
    const bool u16 = s->ctx->pixel_bytes == 2;
    // The byte mask: the streaming kernel zero-filled it if somebody wants it (k_ext_final sets 1s in it either way; without a
    // taker they land in a buffer nobody reads); after k_ext_first it is always produced, so zero it here.
    if (!ext_stream_first(a)) (void)hipMemsetAsync(a.strong_bytes, 0, (size_t)n_frames * a.bytes_frame_stride, st);
    dim3 g3((unsigned)a.n_tiles, n_frames);
    if (u16 && s->ctx->tune.ext_fused) {   // erosion inside the final pass's tiles: one launch, the plane crosses memory once
        hipLaunchKernelGGL(k_ext_erode_final, g3, dim3(256), (size_t)(kTileRows + 10) * a.mpitch, st, a);
        return;
    }
    const int erode = s->ctx->tune.ext_erode;
    if (erode != 0) {
        const unsigned strips = (a.mpitch / 4 + 61) / 62, rows = erode == 1 ? 32 : 16;
        const dim3 ge(strips * 8u * (((unsigned)((a.H + rows - 1) / rows) + 7u) / 8u), n_frames);
        if (erode == 1 && a.eplane_clean) hipLaunchKernelGGL((k_ext_erode_strips<32, true>), ge, dim3(64), 0, st, a);
        else if (erode == 1) hipLaunchKernelGGL((k_ext_erode_strips<32, false>), ge, dim3(64), 0, st, a);
        else if (a.eplane_clean) hipLaunchKernelGGL((k_ext_erode_strips<16, true>), ge, dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_ext_erode_strips<16, false>), ge, dim3(64), 0, st, a);
    } else {
        const unsigned erode_lanes = (a.mpitch / 4) * (unsigned)((a.H + kErodeRows - 1) / kErodeRows);
        hipLaunchKernelGGL(k_ext_erode, dim3((erode_lanes + 255) / 256, n_frames), dim3(256), 0, st, a);
    }
    if (u16) hipLaunchKernelGGL(k_ext_final<uint16_t>, g3, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_ext_final<uint32_t>, g3, dim3(256), 0, st, a);
}

int ensure_extended_buffers(ffs_stream* s) {
    if (s->d_dplane) return FFS_OK;
    ffs_ctx* c = s->ctx;
    const size_t bytes = (size_t)s->max_batch * c->L.plane_frame_stride;
    s->dplane2_clean = false;
    if (dmalloc(&s->d_ext_pair[0], 2 * bytes) != hipSuccess || dmalloc(&s->d_ext_pair[1], 2 * bytes) != hipSuccess) {
        (void)hipGetLastError();
        c->err = "hipMalloc(extended dispersion planes) failed";
        return FFS_ERR_NOMEM;
    }
    s->d_dplane = s->d_ext_pair[0];
    s->d_eplane = s->d_ext_pair[0] + bytes;
    s->d_dplane2 = s->d_ext_pair[1];
    s->d_eplane2 = s->d_ext_pair[1] + bytes;
    return FFS_OK;
}

// Wave logs for this launch (tuning "strong_log"): the 16-bit standard path on a context with sparse streams, a geometry
// kernels_chain.hpp's merge holds (at most twelve strips per frame).  Allocates the logs for the launch's waves on first use
// and puts them into `a`; false: the plane.
bool wave_logs_for(ffs_stream* s, ThresholdArgs& a, uint32_t n_frames) {
    ffs_ctx* c = s->ctx;
    const Layout& L = c->L;
    if (!(c->tune.strong_log != 0 && !a.bright_to_plane && !s->log_off && !s->plane_once && s->st2 != s->st && c->chain_ok
          && s->batch_params.algorithm != FFS_ALGO_DISPERSION_EXTENDED && c->n_tiles <= kChainMaxTiles && L.H <= kChainMaxRows
          && (uint32_t)a.gpf / (uint32_t)kSOwned + 2u <= 16u && a.band_rows <= 1024 && L.W <= 65535))
        return false;
    size_t waves = stream_log_slots(a, n_frames);
    if (waves > s->wlog_waves) {
        // Sized ONCE, for the largest launch any batch of this stream can make (1 .. max_batch frames), so that a batch of another
        // size never re-allocates: hipFree synchronises the whole device, i.e. every other worker's batches in flight.  (A stream is
        // idle here -- submit refuses a busy one, and a re-run inside ffs_wait comes after the batch's last event -- so nothing of
        // its own has to be waited for if it does happen: a tuning change between batches.)
        for (uint32_t nf = 1; nf <= s->max_batch; ++nf) {
            const ThresholdArgs t = make_threshold_args(s, a.image, a.pitch, a.frame_stride, nf);
            waves = std::max(waves, stream_log_slots(t, nf));
        }
        if (s->d_wlog) {
            (void)hipFree(s->d_wlog);
            (void)hipFree(s->d_wlog_n);
            (void)hipFree(s->d_wpix);
            s->d_wlog = nullptr;
            s->d_wlog_n = nullptr;
            s->d_wpix = nullptr;
        }
        s->wlog_waves = 0;
        if (hipMalloc(reinterpret_cast<void**>(&s->d_wlog), waves * kWlogCap * sizeof(uint2)) != hipSuccess
            || hipMalloc(reinterpret_cast<void**>(&s->d_wpix), waves * kWlogCap * sizeof(uint4)) != hipSuccess
            || hipMalloc(reinterpret_cast<void**>(&s->d_wlog_n), waves * 4) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        s->wlog_waves = waves;
    }
    a.wlog = s->d_wlog;
    a.wlog_n = s->d_wlog_n;
    a.wpix = s->d_wpix;
    return true;
}

void bench_launch_dense(ffs_stream* s, const ThresholdArgs& a, uint32_t n_frames, hipEvent_t start, hipEvent_t stop) {
    if (s->batch_params.algorithm == FFS_ALGO_DISPERSION_EXTENDED) launch_ext_first(s, a, n_frames, start, stop);
    else launch_stream(s, a, n_frames, start, stop);
}
void bench_launch_rest(ffs_stream* s, const ThresholdArgs& a, uint32_t n_frames) {
    if (s->batch_params.algorithm == FFS_ALGO_DISPERSION_EXTENDED) launch_ext_rest(s, a, n_frames, s->st);
    else if (a.bright_to_plane) launch_exact(s, a, n_frames, s->st);
    else if (!a.wlog) launch_bright_fix(s, a, s->st);   // (with wave logs the sparse launch decides the bright windows)
}

int check_layout(ffs_stream* s, size_t pitch, size_t fstride, uint32_t n_frames) {
    ffs_ctx* c = s->ctx;
    if (n_frames == 0 || n_frames > s->max_batch) {
        c->err = "n_frames must be in 1..max_batch";
        return FFS_ERR_INVALID;
    }
    if (pitch % 16 || pitch < (size_t)c->L.pitch_px * c->pixel_bytes || pitch >= (1ull << 32)
        || fstride < pitch * c->L.H || (pitch * c->L.H) >= (1ull << 32)) {
        c->err = "device layout: pitch must be a multiple of 16 bytes and >= round_up(width,128)*pixel_bytes; "
                 "frame_stride >= pitch*height";
        return FFS_ERR_INVALID;
    }
    return FFS_OK;
}

#ifdef FFS_EXPERIMENTS
// (experiment) occupies slots for a given time without touching memory
__global__ void k_dummy_spin(uint32_t ticks, uint32_t* sink) {
    extern __shared__ uint32_t s_dummy[];
    const uint64_t t0 = wall_clock64();
    uint32_t it = 0;
    while (wall_clock64() - t0 < ticks && it < (1u << 20)) { __builtin_amdgcn_s_sleep(20); ++it; }
    if (it == 0xFFFFFFFFu) { s_dummy[threadIdx.x] = it; *sink = s_dummy[0]; }
}
#endif

// The sparse stage in small workgroups (kernels_band.hpp, tuning "sparse_bands"): can this launch geometry take it, and are the
// buffers between its two kernels there (allocated once, for the most bands any batch of this stream can have).
static int band_split(const ThresholdArgs& a) { return (std::max(a.band_rows, a.band_rows2) + kBandSplitRows - 1) / kBandSplitRows; }
static bool band_stage_for(ffs_stream* s, const ThresholdArgs& a, uint32_t n_frames) {
    ffs_ctx* c = s->ctx;
    const uint32_t strips = (uint32_t)a.gpf / (uint32_t)kSOwned + 2u;   // strips a frame's groups can touch
    const int sub = band_split(a), sub_rows = (std::max(a.band_rows, a.band_rows2) + sub - 1) / sub;
    if (sub_rows > kBandMaxRows || (uint32_t)sub_rows * std::min(strips, 16u) > (uint32_t)kBandCw || a.n_bands * sub > kMergeMaxBands || c->L.W > 65535)
        return false;
    const uint32_t need = n_frames * (uint32_t)(a.n_bands * sub);
    if (need > s->band_slots) {
        uint32_t slots = need;
        for (uint32_t nf = 1; nf <= s->max_batch; ++nf) {   // (sized once: hipFree synchronises the device -- see wave_logs_for)
            const ThresholdArgs t = make_threshold_args(s, a.image, a.pitch, a.frame_stride, nf);
            slots = std::max(slots, nf * (uint32_t)(t.n_bands * band_split(t)));
        }
        if (s->d_band_hdr) {
            (void)hipFree(s->d_band_hdr); (void)hipFree(s->d_band_acc); (void)hipFree(s->d_band_seam);
            s->d_band_hdr = nullptr; s->d_band_acc = nullptr; s->d_band_seam = nullptr;
        }
        s->band_slots = 0;
        if (hipMalloc(reinterpret_cast<void**>(&s->d_band_hdr), (size_t)slots * sizeof(uint4)) != hipSuccess
            || hipMalloc(reinterpret_cast<void**>(&s->d_band_acc), (size_t)slots * kBandCompStride * sizeof(ChainAcc)) != hipSuccess
            || hipMalloc(reinterpret_cast<void**>(&s->d_band_seam), (size_t)slots * 2 * kBandSeamCap * 4) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        s->band_slots = slots;
    }
    return true;
}

// ---- one batch ------------------------------------------------------------------------------------------------------
int enqueue_batch(ffs_stream* s, const void* d_img, size_t pitch, size_t fstride, uint32_t n, const ffs_params* snapshot) {
    ffs_ctx* c = s->ctx;
    const Layout& L = c->L;
    s->batch_params = snapshot ? *snapshot : c->params;
    const ffs_params& p = s->batch_params;
    s->cur_img = d_img;
    s->cur_pitch = pitch;
    s->cur_fstride = fstride;
    const bool ext = p.algorithm == FFS_ALGO_DISPERSION_EXTENDED;

    (void)hipGetLastError();  // drop any stale error state: the check below is for OUR launches
    const bool wait_upload = s->st_up != s->st && !s->dev_input;   // the frames are in place behind ev[1] (upload / decode stream): waited for below,
                                                                    // in the dense stream this batch's first kernel takes
    bool ext_plane_clean = false;
    if (ext) {
        const int rc = ensure_extended_buffers(s);
        if (rc != FFS_OK) return rc;
        std::swap(s->d_dplane, s->d_dplane2);   // this batch's planes: the ones cleared behind the previous batch (d_dplane2 / d_eplane2 keep that batch's)
        std::swap(s->d_eplane, s->d_eplane2);
        ext_plane_clean = s->dplane2_clean;
        s->ext_e_clean = s->dplane2_clean && s->eplane2_clean && c->tune.ext_erode != 0 && c->tune.ext_e_sparse;
        s->dplane2_clean = false;
        s->eplane2_clean = false;
    } else {
        s->ext_e_clean = false;
    }
    const bool counts_were_clean = !s->counts_dirty;
    const ThresholdArgs ta = make_threshold_args(s, d_img, pitch, fstride, n);
    // "streamed": the plane the sparse stage reads was produced by a streaming kernel into a zeroed plane (and is zeroed
    // again by the compaction); path 0 also keeps the occupancy bitmap in step with it
    const bool streamed = !ext;
    const bool list_path = streamed && !ta.bright_to_plane;
    bool dense_resets = false;   // fills went into the stream's own dense stream: this batch's kernel has to follow them there
    if (wait_upload && ext) HIP_TRY(c, hipStreamWaitEvent(s->st, s->ev[1], 0));
    if (streamed && s->bits_dirty) {  // (another algorithm or a failed batch left bits behind)
        HIP_TRY(c, hipMemsetAsync(s->d_bits, 0, (size_t)s->max_batch * L.plane_frame_stride, s->st));
        dense_resets = true;
    }
    if (streamed && s->counts_dirty) {
        HIP_TRY(c, hipMemsetAsync(s->d_tile_counts, 0, tile_counts_bytes(s), s->st));
        dense_resets = true;
    }
    if (s->occ_dirty) {
        HIP_TRY(c, hipMemsetAsync(s->d_occ, 0, (size_t)s->max_batch * occ_frame_words(L) * 4, s->st));
        s->occ_dirty = false;
        dense_resets = true;
    }
    s->counts_dirty = true;
    s->bits_dirty = true;  // until every launch of this batch is enqueued (a failure in between leaves bits behind)
    // The whole sparse stage in one launch, one workgroup per frame (kernels_chain.hpp) ...
    bool will_chain = c->tune.sparse_stage >= 2 && L.H <= 65535 && c->chain_ok && s->direct_recs && s->h_counts_dev
                      && c->n_tiles <= kChainMaxTiles && L.H <= kChainMaxRows;
    // ... as long as the frames' strong pixels fit its LDS forest.  A frame beyond that runs the same stages on global arrays
    // inside its one workgroup (2.8 ms for 32 frames of 61 k strong pixels, the extended algorithm on the bench frames),
    // where the four grid-wide kernels spread the work over the machine: the path of this batch follows what the stream's
    // previous batch held (data that is dense stays dense; a single dense frame costs one slow batch).
    // Denser frames of 16-bit pixels stay in the one launch while their RUNS fit LDS (kernels_chain.hpp, the RUNS instantiation:
    // 61 k strong pixels are 12 k runs on the bench frames of the extended algorithm); a frame with more runs than that raises
    // flag 16, ffs_wait() runs the batch again through the grid-wide kernels and the stream stays with them for dense batches.
    const bool runs_ok = c->pixel_bytes == 2 && L.W <= kChainRunMaxW && c->tune.chain_runs != 0 && !s->runs_overflowed;
    bool dense_batch = false;
    if (will_chain && c->tune.sparse_stage == 2 && s->n_frames > 0) {
        uint32_t prev_max = 0;
        for (uint32_t f = 0; f < s->n_frames; ++f) prev_max = std::max(prev_max, s->h_counts[f]);
        dense_batch = prev_max > (uint32_t)kChainLdsEntries;
        if (dense_batch && !runs_ok) will_chain = false;
    }
    if (s->force_grid) will_chain = false;
    // Wave logs instead of the plane (tuning "strong_log"): the standard 16-bit path, sparse stage in the one launch, frames
    // that fit its LDS forest.  The streaming kernel then leaves plane, counters, occupancy bitmap and bright list alone.
    ThresholdArgs ta_launch = ta;
    const bool use_log = list_path && will_chain && !dense_batch && wave_logs_for(s, ta_launch, n);
#ifdef FFS_EXPERIMENTS
    if (c->tune.exp.chain_skip) will_chain = false;
#endif
    // ... which then also does the bright-window fix-up, and whose workgroups (a whole CU each) should get their CUs
    // BEFORE the next batch's streaming kernel floods the dispatcher: that kernel waits for this launch to have STARTED.
    // (Without it a batch's sparse launch sits out the whole next streaming kernel: 0.35 ms more latency per batch.)
    // Only while few batches are in flight: with a deep pipeline the latency is hidden anyway, the wait costs the dense
    // stream ~15 us per batch and the fix-up inside the one-workgroup-per-frame launch ~25 us of its CUs (4 batches in
    // flight: 0.369-0.377 against 0.353 ms per step; 2 in flight: 0.385 against 0.523).
    const bool aside = streamed && s->st2 != s->st;   // the context has sparse streams: the dense stream holds streaming kernels only
    const int depth = c->inflight.load() + (s->busy ? 0 : 1);
    // The sparse stage in small workgroups (kernels_band.hpp): wave logs, nobody reads the pixel lists or the byte mask, a geometry
    // its LDS plan holds, and the stream's recent batches did not overflow that plan.
    const bool need_lists = p.want_strong_list || c->tune.device_lists == 1 || (c->tune.device_lists == 2 && g_live_stacks.load() > 0);
    // By pipeline depth (tuning "sparse_bands" = 1): its two launches each become ready behind a streaming kernel that is already
    // being dispatched, so a batch's results are two steps away -- hidden with four batches in flight (101 k against 95.6 k frames/s),
    // not with two or three (79 k / 89 k against 88 k / 95 k for the one-workgroup launch with its head start); alone in flight the
    // band waves win again (58 k against 53 k: nothing to wait behind).  profiles/r05n_bands_by_pipeline_depth.log
    // A context with four or more streams is a pipeline that FILLS through depths two and three (the start of a run, of a timed
    // region): there the band launches win at every depth (+1.2 % on the driver-style line, profiles/r05zo_tune_ab.log).
    const bool depth_ok = c->tune.sparse_bands >= 2 || depth >= 4 || depth <= 1 || c->n_streams_made >= 4;
    const bool banded = use_log && c->tune.sparse_bands != 0 && depth_ok && !need_lists && !ta.dense_mask && !s->bands_once_off && s->band_backoff == 0
                        && band_stage_for(s, ta_launch, n);
    if (use_log && s->band_backoff > 0 && !s->bands_once_off) --s->band_backoff;
    s->band_mode = banded;
    // (band waves fit wherever a streaming wave has left: they need no head start)
    const bool chain_first = list_path && aside && will_chain && !banded && c->tune.chain_first > 0 && depth <= c->tune.chain_first;
    if (chain_first) {
        std::lock_guard<std::mutex> lock(c->stream_mu);   // (the newest start event cannot be re-recorded between the two lines)
        const int slot = c->chain_ev_newest.load();
        if (slot >= 0) HIP_TRY(c, hipStreamWaitEvent(s->st, c->chain_ev[slot], 0));
    }
    hipEvent_t ev_start = nullptr;
    if (s->ev1_pending) { ev_start = s->ev[1]; s->ev1_pending = false; }
    if (ext && s->st2 != s->st && c->tune.ext_rest_aside) {
        // The dense stream carries the first pass alone; erosion and the final pass -- a latency-bound gather over the signal
        // region that keeps the vector units half busy -- go to the batch's sparse stream, ahead of its sparse launch, and run
        // BESIDE the next batch's first pass (an issue-bound stream of the whole frame) instead of between two of them.
        if (ext_stream_first(ta)) {   // (the first pass's stop event rides on its dispatch; the bright-window fix-up goes aside too)
            launch_ext_first(s, ta, n, ev_start, s->ev[2], false, ext_plane_clean, counts_were_clean);
            HIP_TRY(c, hipStreamWaitEvent(s->st2, s->ev[2], 0));
            hipLaunchKernelGGL((k_bright_fix<uint16_t, true>), dim3(32), dim3(256), 0, s->st2, ta);
        } else {
            launch_ext_first(s, ta, n, ev_start, nullptr);
            HIP_TRY(c, hipEventRecord(s->ev[2], s->st));
            HIP_TRY(c, hipStreamWaitEvent(s->st2, s->ev[2], 0));
        }
        launch_ext_rest(s, ta, n, s->st2);
    } else if (ext) {
        launch_ext_first(s, ta, n, ev_start, nullptr, true, ext_plane_clean, counts_were_clean);
        launch_ext_rest(s, ta, n, s->st);
        HIP_TRY(c, hipEventRecord(s->ev[2], s->st));
        if (s->st2 != s->st) HIP_TRY(c, hipStreamWaitEvent(s->st2, s->ev[2], 0));
    } else if (list_path && aside) {
        // the bright-window fix-up goes to the sparse stream (or into the sparse launch itself: chain_first; with wave logs the
        // sparse launch decides those pixels as it reads the logs)
        // Wave-log path with a deep pipeline: the context's two dense HIP streams take the streaming kernels alternately (ffs_internal.hpp,
        // "dense_overlap"): this launch waits for the value the PREVIOUS launch's last workgroup wrote as it started, not for that
        // kernel's end, and writes its own.  Everything that must precede the kernel (the upload's event) goes into the same stream.
        bool overlap = use_log && !chain_first && !dense_resets && c->tune.dense_overlap != 0 && s->st == c->dense_st;
        std::unique_lock<std::mutex> dense_lock(c->dense_mu, std::defer_lock);   // (several threads submit to one context: a launch's number, stream and wait are one step)
        if (overlap) {
            dense_lock.lock();
            if (!c->dense_st2) {   // the partner stream and the hand-over word: made on first use (nothing of this exists in a context that never asks)
                int lo = 0, hi = 0, can = 0;
                bool ok = hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device) == hipSuccess && can
                          && hipExtMallocWithFlags(reinterpret_cast<void**>(&c->d_handoff), 8, hipMallocSignalMemory) == hipSuccess;
                ok = ok && hipMemsetAsync(c->d_handoff, 0, 8, c->up_st) == hipSuccess && hipStreamSynchronize(c->up_st) == hipSuccess;
                ok = ok && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess
                     && hipStreamCreateWithPriority(&c->dense_st2, hipStreamNonBlocking, (lo + hi) / 2) == hipSuccess;
                if (!ok) {
                    (void)hipGetLastError();
                    if (c->d_handoff) (void)hipFree(c->d_handoff);
                    c->d_handoff = nullptr;
                    c->dense_st2 = nullptr;
                    c->tune.dense_overlap = 0;   // (not on this device)
                    overlap = false;
                    dense_lock.unlock();
                }
            }
        }
        if (overlap) {
            const uint32_t seq = ++c->handoff_seq;
            const int which = (int)(seq & 1u);
            hipStream_t dst = which ? c->dense_st2 : c->dense_st;
            if (wait_upload) HIP_TRY(c, hipStreamWaitEvent(dst, s->ev[1], 0));
            if (c->handoff_last >= 0 && c->handoff_last != which)
                HIP_TRY(c, hipStreamWaitValue32(dst, c->d_handoff, seq - 1, hipStreamWaitValueGte, 0xFFFFFFFFu));
            ta_launch.handoff = c->d_handoff;
            ta_launch.handoff_seq = seq;
            launch_stream(s, ta_launch, n, ev_start, s->ev[2], dst);
            c->handoff_last = which;
        } else {
            if (wait_upload) HIP_TRY(c, hipStreamWaitEvent(s->st, s->ev[1], 0));
            launch_stream(s, ta_launch, n, ev_start, s->ev[2]);
        }
        HIP_TRY(c, hipStreamWaitEvent(s->st2, s->ev[2], 0));
        if (!chain_first && !use_log) launch_bright_fix(s, ta, s->st2);
    } else if (ta.bright_to_plane == 2) {
        // the cross-check path of `spotfinder --validate` (tuning "threshold_path" = 2): no streaming kernel, no screen, no LDS
        // queue -- the plane starts as the valid-pixel mask, so k_exact gathers the window of EVERY valid pixel from memory and
        // applies the oracle's predicate to 64-bit sums (exact_strong).  Shares nothing with the hot path but that predicate.
        if (wait_upload) HIP_TRY(c, hipStreamWaitEvent(s->st, s->ev[1], 0));
        if (ev_start) HIP_TRY(c, hipEventRecord(ev_start, s->st));
        HIP_TRY(c, hipMemsetAsync(s->d_sbytes, 0, (size_t)n * L.bytes_frame_stride, s->st));
        for (uint32_t f = 0; f < n; ++f)
            HIP_TRY(c, hipMemcpyAsync(s->d_bits + (size_t)f * L.plane_frame_stride, c->d_maskbits, L.plane_frame_stride, hipMemcpyDeviceToDevice, s->st));
        launch_exact(s, ta, n, s->st);
        HIP_TRY(c, hipEventRecord(s->ev[2], s->st));
        if (s->st2 != s->st) HIP_TRY(c, hipStreamWaitEvent(s->st2, s->ev[2], 0));
    } else {
        if (wait_upload) HIP_TRY(c, hipStreamWaitEvent(s->st, s->ev[1], 0));
        launch_stream(s, ta, n, ev_start, nullptr);
        if (list_path) launch_bright_fix(s, ta, s->st);
        else launch_exact(s, ta, n, s->st);
        HIP_TRY(c, hipEventRecord(s->ev[2], s->st));
        if (s->st2 != s->st) HIP_TRY(c, hipStreamWaitEvent(s->st2, s->ev[2], 0));
    }
    HIP_TRY(c, hipGetLastError());

    CclArgs ca{};
    ca.image = d_img;
    ca.frame_stride = fstride;
    ca.pitch = (uint32_t)pitch;
    ca.bits = s->d_bits;
    ca.clear_bits = streamed ? 1 : 0;
    s->bits_cleared = ca.clear_bits != 0;
    ca.tile_counts = s->d_tile_counts;
    ca.num_strong = s->d_num_strong;
    ca.row_off = s->d_row_off;
    ca.list_k = s->d_list_k;
    ca.list_i = s->d_list_i;
    ca.parent = s->d_parent;
    ca.n_comp = s->d_n_comp;
    ca.overflow = s->d_overflow;
    ca.W = L.W;
    ca.H = L.H;
    ca.pitch_px = L.pitch_px;
    ca.mpitch = L.mpitch;
    ca.plane_frame_stride = L.plane_frame_stride;
    ca.n_tiles = c->n_tiles;
    ca.cap = s->cap;
    ca.max_comp = s->max_comp;
    ca.pixel_bytes = c->pixel_bytes;
    ca.strong_bytes = s->d_sbytes;
    ca.bpitch = L.bpitch;
    ca.bytes_frame_stride = L.bytes_frame_stride;
    ca.acc2 = s->d_acc2;
    ca.summary = s->d_summary;
    // (the byte mask: zero-filled by the streaming kernels only when asked for; the exact stages always produce it)
    ca.dense_bytes = ((list_path || (ext && ext_stream_first(ta))) ? ta.dense_mask : 1) ? 1 : 0;
    ca.need_lists = need_lists ? 1 : 0;
    // (the launch that merges wave logs and the run-based launch of dense frames can do without the lists)
    const bool runs_launch = will_chain && !use_log && c->pixel_bytes == 2 && runs_ok && (dense_batch || c->tune.chain_runs == 2);
    s->lists_valid = !(use_log || runs_launch) || ca.need_lists != 0;
    s->dense_valid = ca.dense_bytes != 0;
    ca.occ = s->d_occ;
    ca.occ_frame_words = occ_frame_words(L);
    ca.occ_spr = L.mpitch / 16;
    // only the streaming kernels and their fix-up keep the bitmap (path 1: a superset of the final plane, which is fine)
    ca.use_occ = (c->tune.occupancy_bitmap && ta.bright_to_plane != 2) ? 1 : 0;   // (kept by the streaming kernels, their fix-up and the extended algorithm's final pass)
    s->occ_dirty = !(ca.use_occ && will_chain);      // nobody consumes (and clears) the bits this batch sets

    SegArgs sa{};
    sa.list_k = s->d_list_k;
    sa.list_i = s->d_list_i;
    sa.parent = s->d_parent;
    sa.seg_n = s->d_num_strong;
    sa.seg_stride = s->cap;
    sa.n_comp = s->d_n_comp;
    sa.max_comp = s->max_comp;
    sa.overflow = s->d_overflow;
    sa.W = (uint32_t)L.W;
    sa.H = (uint32_t)L.H;
    sa.row_off = s->d_row_off;
    sa.n_slices = 1;
    sa.min_spot_size = p.min_spot_size;
    sa.max_sep = p.max_peak_centroid_separation;
    sa.summary = s->d_summary;
    sa.acc2 = s->d_acc2;
    sa.zero_counts = s->d_tile_counts;
    sa.zero_per_seg = (uint32_t)c->n_tiles;
    sa.zero_word = s->d_tile_counts + tile_counts_bytes(s) / 4 - 1;

    s->chain_mode = will_chain;
    s->path_bits = (use_log ? FFS_PATH_WAVE_LOGS : 0u) | (will_chain && !banded ? FFS_PATH_FRAME_CHAIN : 0u) | (banded ? FFS_PATH_BANDS : 0u)
                   | (runs_launch ? FFS_PATH_RUNS : 0u) | (!will_chain ? FFS_PATH_GRID_KERNELS : 0u) | (ext ? FFS_PATH_EXTENDED : 0u);
    if (s->chain_mode) {
        ChainArgs A{};
        A.c = ca;
        A.s = sa;
        A.s.recs = s->h_recs_dev;
        A.h_counts = s->h_counts_dev;
        A.max_batch = (uint32_t)s->max_batch;
        A.rec_stride = s->max_comp;
#ifdef FFS_EXPERIMENTS
        A.stop_after = c->tune.exp.chain_stop;
        if (std::getenv("FFS_EXP_CHAIN_TS")) {
            if (!s->h_phase_ts && hipHostMalloc(reinterpret_cast<void**>(&s->h_phase_ts), (size_t)s->max_batch * 64, hipHostMallocDefault) == hipSuccess) {
                std::memset(s->h_phase_ts, 0, (size_t)s->max_batch * 64);
                (void)hipHostGetDevicePointer(reinterpret_cast<void**>(&s->h_phase_ts_dev), s->h_phase_ts, 0);
            }
            A.phase_ts = s->h_phase_ts_dev;
        }
#endif
        A.t = ta_launch;
        A.fix_bright = (chain_first && !use_log) ? 1 : 0;
        A.fix_done = s->d_tile_counts + tile_counts_bytes(s) / 4 - 2;
        A.runs_ok = runs_ok ? (c->tune.chain_runs == 2 ? 2 : 1) : 0;
        {
            // the launch's start event belongs to the context (ffs_internal.hpp); published under the lock the waiting side takes
            std::lock_guard<std::mutex> lock(c->stream_mu);
            const int slot = (int)(c->chain_ev_next.fetch_add(1) % ffs_ctx::kChainEvents);
            if (banded) {
                BandArgs BA{};
                BA.A = A;
                BA.hdr = s->d_band_hdr;
                BA.acc = reinterpret_cast<ChainAcc*>(s->d_band_acc);
                BA.seam = s->d_band_seam;
                BA.sub = band_split(ta_launch);
                BA.sub_rows = (std::max(ta_launch.band_rows, ta_launch.band_rows2) + BA.sub - 1) / BA.sub;
                const dim3 gb((unsigned)(ta_launch.n_bands * BA.sub), n);
                if (c->pixel_bytes == 2) hipExtLaunchKernelGGL(k_band_cc<uint16_t>, gb, dim3(64), 0, s->st2, c->chain_ev[slot], nullptr, 0, BA);
                else hipExtLaunchKernelGGL(k_band_cc<uint32_t>, gb, dim3(64), 0, s->st2, c->chain_ev[slot], nullptr, 0, BA);
                hipLaunchKernelGGL(k_frame_merge, dim3(n), dim3(kMergeThreads), 0, s->st2, BA);
            }
            else if (use_log && c->pixel_bytes == 2) hipExtLaunchKernelGGL((k_frame_chain<uint16_t, false, true>), dim3(n), dim3(kChainThreads), kChainDynBytes, s->st2, c->chain_ev[slot], nullptr, 0, A);
            else if (use_log) hipExtLaunchKernelGGL((k_frame_chain<uint32_t, false, true>), dim3(n), dim3(kChainThreads), kChainDynBytes, s->st2, c->chain_ev[slot], nullptr, 0, A);
            else if (c->pixel_bytes == 2 && runs_ok && (dense_batch || c->tune.chain_runs == 2)) hipExtLaunchKernelGGL((k_frame_chain<uint16_t, true>), dim3(n), dim3(kChainThreads), kChainDynBytes, s->st2, c->chain_ev[slot], nullptr, 0, A);
            else if (c->pixel_bytes == 2) hipExtLaunchKernelGGL(k_frame_chain<uint16_t>, dim3(n), dim3(kChainThreads), kChainDynBytes, s->st2, c->chain_ev[slot], nullptr, 0, A);
            else hipExtLaunchKernelGGL(k_frame_chain<uint32_t>, dim3(n), dim3(kChainThreads), kChainDynBytes, s->st2, c->chain_ev[slot], nullptr, 0, A);
            if (aside) c->chain_ev_newest.store(slot);
        }
        HIP_TRY(c, hipGetLastError());
        if (ext && ext_stream_first(ta) && s->st2 != s->st) {
            // the plane the previous batch used (nobody reads it any more) is cleared here, beside the dense kernels, for the next batch
            // (with the strip erosion the signal-region plane behind it too: one fill, the two are one allocation)
            HIP_TRY(c, hipMemsetAsync(s->d_dplane2, 0, (size_t)s->max_batch * L.plane_frame_stride * (c->tune.ext_erode != 0 && c->tune.ext_e_sparse ? 2u : 1u), s->st2));
            s->dplane2_clean = true;
            s->eplane2_clean = c->tune.ext_erode != 0 && c->tune.ext_e_sparse;
        }
        HIP_TRY(c, hipEventRecord(s->ev[4], s->st2));
        s->ev3_is_ev4 = true;
        s->spec_recs_copied = (uint64_t)s->max_batch * s->max_comp;
        s->bits_dirty = !streamed;
        s->counts_dirty = false;
        mark_busy(s);
        s->n_frames = n;
        return FFS_OK;
    }
    // the same stages as four grid-wide kernels (frames taller than k_frame_chain's LDS plan, records not written to the
    // host directly, tuning "sparse_stage" = 1)
    bool skip_sparse = false;
#ifdef FFS_EXPERIMENTS
    if (c->tune.exp.chain_skip) {
        skip_sparse = true;
        if (c->tune.exp.dummy_us > 0)
            hipLaunchKernelGGL(k_dummy_spin, dim3(c->tune.exp.dummy_wg), dim3(c->tune.exp.dummy_threads),
                               (size_t)c->tune.exp.dummy_lds, s->st2, (uint32_t)c->tune.exp.dummy_us * 100u, s->d_tile_counts);
    }
#endif
    if (!skip_sparse) {
        sa.recs = s->direct_recs ? (void*)s->h_recs_dev : (void*)s->d_recs;
        sa.chunk_roots = s->d_chunk_roots;
        sa.chunks_max = s->cap / kRootChunk + 1;
        const dim3 gseg((unsigned)c->tune.ccl_grid, n), b256(256);
        if (c->pixel_bytes == 2) hipLaunchKernelGGL(k_emit_list_w<uint16_t>, dim3(c->n_tiles, n), dim3(64), 0, s->st2, ca);
        else hipLaunchKernelGGL(k_emit_list_w<uint32_t>, dim3(c->n_tiles, n), dim3(64), 0, s->st2, ca);
        hipLaunchKernelGGL(k_union<false>, gseg, b256, 0, s->st2, sa);
        hipLaunchKernelGGL(k_reduce_roots, gseg, b256, 0, s->st2, sa);
        hipLaunchKernelGGL(k_finalize_roots, gseg, b256, 0, s->st2, sa);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(s->ev[3], s->st2));
    s->ev3_is_ev4 = false;

    // small counts first; ffs_wait() sizes the record copy from them
    const size_t B = s->max_batch;
    HIP_TRY(c, hipMemcpyAsync(s->h_counts, s->d_num_strong, (B * 10 + 1) * 4, hipMemcpyDeviceToHost, s->st2));
    if (s->direct_recs) {
        s->spec_recs_copied = (uint64_t)B * s->max_comp;  // everything is on the host already
    } else {
        s->spec_recs_copied = std::min<uint64_t>((uint64_t)s->spec_recs_per_frame * n, (uint64_t)B * s->max_comp);
        HIP_TRY(c, hipMemcpyAsync(s->h_recs, s->d_recs, s->spec_recs_copied * sizeof(WireRec2), hipMemcpyDeviceToHost, s->st2));
    }
    HIP_TRY(c, hipEventRecord(s->ev[4], s->st2));
    s->bits_dirty = !streamed || skip_sparse;  // the compaction of a streamed batch leaves the plane all zero again
    s->counts_dirty = skip_sparse;  // k_union cleared the counts of the frames of this batch (all the streaming kernel touched)
    mark_busy(s);
    s->n_frames = n;
    return FFS_OK;
}

extern "C" int ffs_submit_device(ffs_stream* s, const void* device_pixels, size_t pitch, size_t fstride,
                                 uint32_t n_frames, int64_t first_frame_id) {
    if (!s || !device_pixels || !stream_handle_ok(s)) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    if (s->busy) {
        c->err = "stream already has a batch in flight: call ffs_wait() first";
        return FFS_ERR_INVALID;
    }
    int rc = check_layout(s, pitch, fstride, n_frames);
    if (rc != FFS_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    // (no marker here: every packet in the dense stream is ~5 us between two streaming kernels; enqueue_batch attaches
    // the start event to its first kernel where it can, or records it)
    s->dev_input = true;
    s->ev1_pending = true;
    s->first_id = first_frame_id;
    s->reruns = 0;
    rc = enqueue_batch(s, device_pixels, pitch, fstride, n_frames);
    if (rc == FFS_OK) ahead_register(s);
    return rc;
}

extern "C" int ffs_submit(ffs_stream* s, const void* host_pixels, uint32_t n_frames, int64_t first_frame_id) {
    if (!s || !host_pixels || !stream_handle_ok(s)) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    if (s->busy) {
        c->err = "stream already has a batch in flight: call ffs_wait() first";
        return FFS_ERR_INVALID;
    }
    if (n_frames == 0 || n_frames > s->max_batch) {
        c->err = "n_frames must be in 1..max_batch";
        return FFS_ERR_INVALID;
    }
    const Layout& L = c->L;
    HIP_TRY(c, hipSetDevice(c->device));
    s->dev_input = false;
    HIP_TRY(c, hipEventRecord(s->ev[0], s->st_up));
    // one 2D copy: the default device layout keeps frames contiguous (frame_stride = H * pitch)
    const size_t row = (size_t)L.W * c->pixel_bytes;
    HIP_TRY(c, hipMemcpy2DAsync(s->d_img, L.pitch, host_pixels, row, row, (size_t)L.H * n_frames,
                                hipMemcpyHostToDevice, s->st_up));
    HIP_TRY(c, hipEventRecord(s->ev[1], s->st_up));
    s->first_id = first_frame_id;
    s->reruns = 0;
    const int rc = enqueue_batch(s, s->d_img, L.pitch, L.frame_stride, n_frames);
    if (rc == FFS_OK) ahead_register(s);
    return rc;
}


// ---- compressed input -------------------------------------------------------------------------------

static int ensure_decode_buffers(ffs_stream* s) {
    ffs_ctx* c = s->ctx;
    if (s->d_tab) return FFS_OK;
    const size_t es = c->pixel_bytes, nelem = (size_t)c->L.W * c->L.H;
    // bitshuffle's blocking (bshuf_default_block_size, and the loop of bshuf_blocked_wrap_fun)
    const size_t block = (size_t)kDecBlockBytes / es;
    const size_t n_full = nelem / block, rem = nelem - n_full * block;
    s->dec_block_elems = (uint32_t)block;
    s->dec_blocks = (uint32_t)(n_full + (rem >= 8 ? 1 : 0));
    s->dec_last = (uint32_t)(rem >= 8 ? rem / 8 * 8 : block);
    s->dec_tail = (uint32_t)(rem % 8);
    const size_t tab_bytes = (size_t)s->max_batch * (s->dec_blocks + 1) * sizeof(uint2);
    if (dmalloc(&s->d_tab, tab_bytes) != hipSuccess
        || hipHostMalloc(reinterpret_cast<void**>(&s->h_tab), tab_bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        c->err = "allocation of the compressed-chunk buffers failed";
        return FFS_ERR_NOMEM;
    }
    return FFS_OK;
}

static inline uint32_t be32(const uint8_t* p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

// Step 1 (caller's thread): validates the headers, places the chunks in the pinned staging buffer
// (unless they already are there) and starts their copy to the device.  Fills base[] = offset of every
// chunk in the staging / device buffer.
static int stage_chunks(ffs_stream* s, const void* const* chunks, const size_t* chunk_bytes, uint32_t n,
                        std::vector<size_t>& base) {
    ffs_ctx* c = s->ctx;
    const Layout& L = c->L;
    int rc = ensure_decode_buffers(s);
    if (rc != FFS_OK) return rc;
    const size_t es = c->pixel_bytes, raw_bytes = (size_t)L.W * L.H * es;
    uint32_t in_place = 0;
    for (uint32_t f = 0; f < n; ++f) {
        if (!chunks[f] || chunk_bytes[f] < 12) {
            c->err = "ffs_submit_compressed: a chunk is shorter than its 12-byte header";
            return FFS_ERR_INVALID;
        }
        const uint8_t* p = static_cast<const uint8_t*>(chunks[f]);
        uint64_t total = 0;
        for (int i = 0; i < 8; ++i) total = (total << 8) | p[i];
        if (total != raw_bytes) {
            c->err = "ffs_submit_compressed: chunk header says " + std::to_string(total) + " bytes, the context's frames have "
                     + std::to_string(raw_bytes);
            return FFS_ERR_INVALID;
        }
        if (p >= s->h_img && p + chunk_bytes[f] <= s->h_img + s->h_img_bytes) ++in_place;
    }
    if (in_place != 0 && in_place != n) {
        c->err = "ffs_submit_compressed: either all chunks lie in the stream's host buffer or none";
        return FFS_ERR_INVALID;
    }
    base.assign(n, 0);
    size_t lo = 0, hi = 0;
    if (in_place) {
        lo = SIZE_MAX;
        for (uint32_t f = 0; f < n; ++f) {
            base[f] = (size_t)(static_cast<const uint8_t*>(chunks[f]) - s->h_img);
            lo = std::min(lo, base[f]);
            hi = std::max(hi, base[f] + chunk_bytes[f]);
        }
        lo &= ~(size_t)15;
    } else {
        size_t need = 0;
        for (uint32_t f = 0; f < n; ++f) need += (chunk_bytes[f] + 15) & ~(size_t)15;
        rc = ensure_host_staging(s, need + need / 4 + 4096);   // (grows with headroom; kept from then on)
        if (rc != FFS_OK) return rc;
        size_t cur = 0;
        for (uint32_t f = 0; f < n; ++f) {
            std::memcpy(s->h_img + cur, chunks[f], chunk_bytes[f]);
            base[f] = cur;
            cur = (cur + chunk_bytes[f] + 15) & ~(size_t)15;
        }
        hi = cur;
    }
    if (hi > 0xFFFFFFF0ull) {
        c->err = "ffs_submit_compressed: more than 4 GiB of chunks in one batch";
        return FFS_ERR_INVALID;
    }
    if (s->d_comp_bytes < hi + 64) {   // the device side of the staging area follows its size (hipMalloc is cheap)
        if (s->d_comp) {
            HIP_TRY(c, hipStreamSynchronize(s->st_up));
            (void)hipFree(s->d_comp);
            s->d_comp = nullptr;
        }
        const size_t want = std::max(hi + hi / 4 + 4096, s->h_img_bytes) + 64;
        if (dmalloc(&s->d_comp, want) != hipSuccess) {
            (void)hipGetLastError();
            s->d_comp_bytes = 0;
            c->err = "allocation of the device buffer for compressed chunks failed";
            return FFS_ERR_NOMEM;
        }
        s->d_comp_bytes = want;
    }
    if (in_place) {
        // chunks placed by the caller (a driver's fixed slots leave gaps between them): only the bytes of the chunks cross PCIe --
        // ranges closer than 512 KB travel as one copy (the gap costs what a copy of its own would), each copy costs ~10 us of the caller's time
        std::vector<std::pair<size_t, size_t>> r(n);
        for (uint32_t f = 0; f < n; ++f) r[f] = {base[f] & ~(size_t)15, base[f] + chunk_bytes[f]};
        std::sort(r.begin(), r.end());
        size_t a = r[0].first, b = r[0].second;
        for (uint32_t f = 1; f <= n; ++f) {
            if (f < n && r[f].first <= b + 524288) { b = std::max(b, r[f].second); continue; }
            HIP_TRY(c, hipMemcpyAsync(s->d_comp + a, s->h_img + a, b - a, hipMemcpyHostToDevice, s->st_up));
            if (f < n) { a = r[f].first; b = r[f].second; }
        }
    } else {
        HIP_TRY(c, hipMemcpyAsync(s->d_comp + lo, s->h_img + lo, hi - lo, hipMemcpyHostToDevice, s->st_up));
    }
    // "this batch's chunks are on the device", recorded HERE: the upload stream is shared by the context's streams, and an event
    // recorded later (by the helper thread, after the block index) would also wait for every other batch's chunks queued meanwhile
    HIP_TRY(c, hipEventRecord(s->ev[6], s->st_up));
    return FFS_OK;
}

// Step 2 (may run on the stream's helper thread while the chunks cross PCIe): indexes the blocks --
// each frame is a chain of [4-byte length][payload], a pointer chase of ~2 ms for 32 Eiger frames;
// the frames' chains are walked side by side so that their cache misses overlap -- and enqueues the
// table copy.  Errors go to `err`, not to the context (another thread may own that string).
static int index_blocks(ffs_stream* s, const std::vector<size_t>& base, const std::vector<size_t>& chunk_bytes,
                        std::string& err, hipStream_t tab_stream) {
    ffs_ctx* c = s->ctx;
    const uint32_t n = (uint32_t)base.size();
    const size_t es = c->pixel_bytes;
    const uint32_t nb = s->dec_blocks, stride = nb + 1;
    std::vector<size_t> pos(n, 12);
    std::atomic<bool> ok{true};
    static const bool trace = std::getenv("FFS_TRACE_SUBMIT") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto walk = [&](uint32_t f0, uint32_t f1) {  // frames [f0, f1), chains interleaved
        for (uint32_t b = 0; b < nb; ++b)
            for (uint32_t f = f0; f < f1; ++f) {
                if (pos[f] + 4 > chunk_bytes[f]) { ok = false; return; }
                const uint32_t clen = be32(s->h_img + base[f] + pos[f]);
                s->h_tab[(size_t)f * stride + b] = make_uint2((uint32_t)(base[f] + pos[f] + 4), clen);
                pos[f] += 4 + (size_t)clen;
            }
        for (uint32_t f = f0; f < f1; ++f) {
            const size_t tail = (size_t)s->dec_tail * es;
            if (pos[f] + tail > chunk_bytes[f]) ok = false;
            s->h_tab[(size_t)f * stride + nb] = make_uint2((uint32_t)(base[f] + pos[f]), (uint32_t)tail);
        }
    };
    const uint32_t n_thr = (uint64_t)n * nb >= 32768 ? std::min<uint32_t>(4, n) : 1;
    if (n_thr <= 1) {
        walk(0, n);
    } else {
        std::vector<std::thread> th;
        for (uint32_t t = 1; t < n_thr; ++t) th.emplace_back(walk, n * t / n_thr, n * (t + 1) / n_thr);
        walk(0, n / n_thr);
        for (auto& t : th) t.join();
    }
    if (trace)
        std::fprintf(stderr, "[ffs] indexed %u blocks in %.3f ms (%u threads)\n", n * nb,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), n_thr);
    if (!ok) {
        err = "ffs_submit_compressed: block lengths run past the end of a chunk";
        return FFS_ERR_INVALID;
    }
    // (the table goes up in the stream the decode kernel runs in: half a megabyte that must not queue behind other batches' chunks)
    const hipError_t e = hipMemcpyAsync(s->d_tab, s->h_tab, (size_t)n * stride * sizeof(uint2), hipMemcpyHostToDevice, tab_stream);
    if (e != hipSuccess) {
        err = std::string("hipMemcpyAsync(block table): ") + hipGetErrorString(e);
        return FFS_ERR_DEVICE;
    }
    return FFS_OK;
}

static void launch_decode(ffs_stream* s, uint32_t n, hipStream_t st) {
    ffs_ctx* c = s->ctx;
    DecodeArgs da{};
    da.comp = s->d_comp;
    da.table = s->d_tab;
    da.image = s->d_img;
    da.frame_stride = c->L.frame_stride;
    da.pitch = c->L.pitch;
    da.W = c->L.W;
    da.H = c->L.H;
    da.elem_bytes = c->pixel_bytes;
    da.blocks_per_frame = s->dec_blocks;
    da.block_elems = s->dec_block_elems;
    da.last_block_elems = s->dec_last;
    da.tail_elems = s->dec_tail;
    da.error = s->d_overflow;
    const dim3 grid(s->dec_blocks + 1, n);
    if (c->pixel_bytes == 2) hipLaunchKernelGGL(k_bshuf_lz4_decode<2>, grid, dim3(64), 0, st, da);
    else hipLaunchKernelGGL(k_bshuf_lz4_decode<4>, grid, dim3(64), 0, st, da);
}

static int ffs_submit_compressed_impl(ffs_stream* s, const void* const* chunks, const size_t* chunk_bytes,
                                     uint32_t n_frames, int64_t first_frame_id) {
    if (!s || !chunks || !chunk_bytes) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    if (s->busy) {
        c->err = "stream already has a batch in flight: call ffs_wait() first";
        return FFS_ERR_INVALID;
    }
    if (n_frames == 0 || n_frames > s->max_batch) {
        c->err = "n_frames must be in 1..max_batch";
        return FFS_ERR_INVALID;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    s->dev_input = false;
    HIP_TRY(c, hipEventRecord(s->ev[0], s->st_up));
    std::vector<size_t> base;
    int rc = stage_chunks(s, chunks, chunk_bytes, n_frames, base);
    if (rc != FFS_OK) return rc;
    // The rest -- block index, table copy, decode kernel and the hot path's launches -- is enqueued by a
    // helper thread, so that the caller gets its thread back while the index is built; ffs_wait joins it.
    s->first_id = first_frame_id;
    s->reruns = 0;
    s->n_frames = n_frames;
    mark_busy(s);
    s->job_rc = FFS_OK;
    s->job_err.clear();
    const ffs_params snap = c->params;
    std::vector<size_t> sizes(chunk_bytes, chunk_bytes + n_frames);
    s->job = std::thread([s, c, snap, n_frames, base = std::move(base), sizes = std::move(sizes)]() {
        if (hipSetDevice(c->device) != hipSuccess) {
            s->job_rc = FFS_ERR_DEVICE;
            s->job_err = "hipSetDevice failed on the stream's helper thread";
            return;
        }
        // the decode kernel runs with the dense kernels (in their order), behind the copies of its input
        hipStream_t dst = c->tune.decode_in_dense_stream ? s->st : s->st_up;
        int r = index_blocks(s, base, sizes, s->job_err, dst);
        if (r == FFS_OK) {
            (void)hipGetLastError();
            hipError_t e = hipSuccess;
            if (dst != s->st_up) e = hipStreamWaitEvent(dst, s->ev[6], 0);
            launch_decode(s, n_frames, dst);
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipEventRecord(s->ev[1], dst);
            if (e != hipSuccess) {
                s->job_err = std::string("decode launch: ") + hipGetErrorString(e);
                r = FFS_ERR_DEVICE;
            }
        }
        if (r == FFS_OK) {
            r = enqueue_batch(s, s->d_img, c->L.pitch, c->L.frame_stride, n_frames, &snap);
            if (r != FFS_OK) s->job_err = c->err;
        }
        s->job_rc = r;
        if (r == FFS_OK) ahead_register(s);
    });
    return FFS_OK;
}

extern "C" int ffs_decode_only(ffs_stream* s, const void* const* chunks, const size_t* chunk_bytes, uint32_t n_frames,
                               uint32_t iters, float* ms_decode, void* host_out) {
    if (!s || !chunks || !chunk_bytes || iters == 0) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    if (s->busy || n_frames == 0 || n_frames > s->max_batch) {
        c->err = "ffs_decode_only: stream busy or n_frames out of range";
        return FFS_ERR_INVALID;
    }
    const Layout& L = c->L;
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<size_t> base;
    int rc = stage_chunks(s, chunks, chunk_bytes, n_frames, base);
    if (rc == FFS_OK) {
        std::string err;
        rc = index_blocks(s, base, std::vector<size_t>(chunk_bytes, chunk_bytes + n_frames), err, s->st_up);
        if (rc != FFS_OK) c->err = err;
    }
    if (rc != FFS_OK) {
        (void)hipStreamSynchronize(s->st_up);
        return rc;
    }
    (void)hipGetLastError();
    HIP_TRY(c, hipEventRecord(s->ev[0], s->st_up));
    for (uint32_t i = 0; i < iters; ++i) launch_decode(s, n_frames, s->st_up);
    HIP_TRY(c, hipEventRecord(s->ev[1], s->st_up));
    HIP_TRY(c, hipGetLastError());
    uint32_t flag = 0;
    HIP_TRY(c, hipMemcpyAsync(&flag, s->d_overflow, 4, hipMemcpyDeviceToHost, s->st_up));
    HIP_TRY(c, hipStreamSynchronize(s->st_up));
    float ms = 0;
    HIP_TRY(c, hipEventElapsedTime(&ms, s->ev[0], s->ev[1]));
    if (ms_decode) *ms_decode = ms / iters;
    if (host_out) {
        const size_t row = (size_t)L.W * c->pixel_bytes;
        HIP_TRY(c, hipMemcpy2D(host_out, row, s->d_img, L.pitch, row, (size_t)L.H * n_frames, hipMemcpyDeviceToHost));
    }
    if (flag & 4u) {
        HIP_TRY(c, hipMemset(s->d_overflow, 0, 4));
        c->err = "corrupt bitshuffle-LZ4 chunk: an LZ4 block did not decode to its block size";
        return FFS_ERR_INVALID;
    }
    return FFS_OK;
}

extern "C" int ffs_submit_compressed(ffs_stream* s, const void* const* chunks, const size_t* chunk_bytes,
                                     uint32_t n_frames, int64_t first_frame_id) {
    if (!s || !stream_handle_ok(s)) return FFS_ERR_INVALID;
    return guarded(s->ctx, [&] { return ffs_submit_compressed_impl(s, chunks, chunk_bytes, n_frames, first_frame_id); });
}

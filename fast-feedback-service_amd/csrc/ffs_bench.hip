// ffs_bench.hip -- measurement entry points of libffs_hip.so (bench.py, tools/): per-launch kernel durations from HIP events
// on the dispatch, the memory ceiling of the box, a native submit/wait loop, and the sqrt self-test of the parity suite.
#include "ffs_internal.hpp"

// Average duration of ONE launch of the threshold stage's dense kernel (`ms_dense`) and of what follows it inside the stage
// (`ms_rest`: k_bright_fix; extended algorithm: erosion + final pass), over `iters` launches.  Every launch runs on the state
// the hot path gives it -- zeroed per-tile counts and bright-list count, empty plane and occupancy bitmap -- and is timed by
// start / stop events that ride on its own dispatch, so the resets between launches are outside the measurement.
extern "C" int ffs_bench_threshold(ffs_stream* s, const void* device_pixels, size_t pitch, size_t fstride,
                                   uint32_t n_frames, uint32_t iters, float* ms_dense, float* ms_rest) {
    if (!s || !device_pixels || iters == 0) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    if (s->busy) {
        c->err = "stream busy";
        return FFS_ERR_INVALID;
    }
    int rc = check_layout(s, pitch, fstride, n_frames);
    if (rc != FFS_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    s->batch_params = c->params;
    const bool ext = c->params.algorithm == FFS_ALGO_DISPERSION_EXTENDED;
    if (ext) {
        rc = ensure_extended_buffers(s);
        if (rc != FFS_OK) return rc;
    }
    s->ext_e_clean = false;
    ThresholdArgs ta = make_threshold_args(s, device_pixels, pitch, fstride, n_frames);
    const bool e_sparse = ext && c->tune.ext_erode != 0 && c->tune.ext_e_sparse;   // (the hot path clears the signal-region plane behind the previous batch)
    ta.eplane_clean = e_sparse ? 1 : 0;
    if (!ext) (void)wave_logs_for(s, ta, n_frames);   // (the kernel as the hot path launches it)
    const Layout& L = c->L;
    std::vector<hipEvent_t> ev(4 * (size_t)iters, nullptr);
    auto cleanup = [&]() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); };
    for (auto& e : ev)
        if (hipEventCreate(&e) != hipSuccess) {
            cleanup();
            c->err = "hipEventCreate failed";
            return FFS_ERR_DEVICE;
        }
    (void)hipGetLastError();
    hipError_t err = hipSuccess;
    for (uint32_t i = 0; i < iters && err == hipSuccess; ++i) {
        // the state a batch of the hot path starts from
        err = hipMemsetAsync(s->d_bits, 0, (size_t)s->max_batch * L.plane_frame_stride, s->st);
        if (err == hipSuccess) err = hipMemsetAsync(s->d_tile_counts, 0, tile_counts_bytes(s), s->st);
        if (err == hipSuccess) err = hipMemsetAsync(s->d_occ, 0, (size_t)s->max_batch * occ_frame_words(L) * 4, s->st);
        if (err == hipSuccess && e_sparse) err = hipMemsetAsync(s->d_eplane, 0, (size_t)s->max_batch * L.plane_frame_stride, s->st);
        if (err != hipSuccess) break;
        bench_launch_dense(s, ta, n_frames, ev[4 * i], ev[4 * i + 1]);
        err = hipEventRecord(ev[4 * i + 2], s->st);
        bench_launch_rest(s, ta, n_frames);
        if (err == hipSuccess) err = hipEventRecord(ev[4 * i + 3], s->st);
        if (err == hipSuccess) err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemsetAsync(s->d_overflow, 0, 4, s->st);   // (a bright list that overflowed here is not the next batch's business)
    if (err == hipSuccess) err = hipStreamSynchronize(s->st);
    double t_dense = 0, t_rest = 0;
    for (uint32_t i = 0; i < iters && err == hipSuccess; ++i) {
        float a = 0, b = 0;
        err = hipEventElapsedTime(&a, ev[4 * i], ev[4 * i + 1]);
        if (err == hipSuccess) err = hipEventElapsedTime(&b, ev[4 * i + 2], ev[4 * i + 3]);
        t_dense += a;
        t_rest += b;
    }
    cleanup();
    s->bits_dirty = true;  // no compaction ran: plane, counts and bitmap still hold the last launch's output
    s->counts_dirty = true;
    s->occ_dirty = true;
    if (err != hipSuccess) {
        c->err = std::string("ffs_bench_threshold: ") + hipGetErrorString(err);
        return FFS_ERR_DEVICE;
    }
    if (ms_dense) *ms_dense = (float)(t_dense / iters);
    if (ms_rest) *ms_rest = (float)(t_rest / iters);
    return FFS_OK;
}

// The submit / wait loop bench.py runs in Python for one GPU, natively: `steps` batches of device-resident frames through
// `n_streams` streams of one context, all in flight.  For drivers with one host thread per GPU (bench.py --single-process):
// no interpreter lock is held while it runs.  Sums over all frames: boxes (spots after the size filter) and strong pixels.
static int bench_pipeline_impl(ffs_stream* const* streams, uint32_t n_streams, const void* device_pixels, size_t pitch, size_t fstride,
                               uint32_t n_frames, uint32_t steps, int64_t first_frame_id, uint64_t* n_boxes, uint64_t* n_strong) {
    uint64_t boxes = 0, strong = 0;
    std::vector<uint32_t> inflight;   // stream indices, oldest first
    auto reap = [&]() -> int {
        const ffs_frame_result* res = nullptr;
        uint32_t n = 0;
        const int rc = ffs_wait_impl(streams[inflight.front()], &res, &n);
        inflight.erase(inflight.begin());
        if (rc != FFS_OK) return rc;
        for (uint32_t i = 0; i < n; ++i) { boxes += res[i].n_boxes; strong += res[i].num_strong_pixels; }
        return FFS_OK;
    };
    for (uint32_t step = 0; step < steps; ++step) {
        const uint32_t i = step % n_streams;
        if (inflight.size() == n_streams) {
            const int rc = reap();
            if (rc != FFS_OK) return rc;
        }
        const int rc = ffs_submit_device(streams[i], device_pixels, pitch, fstride, n_frames, first_frame_id + (int64_t)step * n_frames);
        if (rc != FFS_OK) return rc;
        inflight.push_back(i);
    }
    while (!inflight.empty()) {
        const int rc = reap();
        if (rc != FFS_OK) return rc;
    }
    if (n_boxes) *n_boxes = boxes;
    if (n_strong) *n_strong = strong;
    return FFS_OK;
}

extern "C" int ffs_bench_pipeline(ffs_stream* const* streams, uint32_t n_streams, const void* device_pixels, size_t pitch, size_t fstride,
                                  uint32_t n_frames, uint32_t steps, int64_t first_frame_id, uint64_t* n_boxes, uint64_t* n_strong) {
    if (!streams || n_streams == 0 || !device_pixels) return FFS_ERR_INVALID;
    for (uint32_t i = 0; i < n_streams; ++i)
        if (!streams[i] || streams[i]->ctx != streams[0]->ctx) return FFS_ERR_INVALID;
    return guarded(streams[0]->ctx, [&] {
        return bench_pipeline_impl(streams, n_streams, device_pixels, pitch, fstride, n_frames, steps, first_frame_id, n_boxes, n_strong);
    });
}

// ---- measured memory ceiling (a ceiling measured on the box beside the nominal 8 TB/s) ----
// k_probe<0>: every byte of the batch's pixel buffer is read once (16 B per lane, consecutive).
// k_probe<1|2>: the same reads plus one 8-byte zero store per 16 bytes read into the byte-mask buffer -- the 2:1
// read/write mix of the threshold kernel (2 B pixel in, 1 B mask out), again perfectly linear.
template <int WRITE>  // 0 = reads only, 1 = plain stores, 2 = non-temporal stores
__global__ __launch_bounds__(256) void k_probe(const uint4* src, uint2* dst, uint64_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const uint4 v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
        if (WRITE == 1) dst[i] = make_uint2(0u, 0u);
        if (WRITE == 2) {
            __builtin_nontemporal_store(0u, &dst[i].x);
            __builtin_nontemporal_store(0u, &dst[i].y);
        }
    }
    if (acc == 0x9E3779B9u) *sink = acc;  // keeps the loads alive
}

extern "C" int ffs_bench_hbm(ffs_stream* s, uint32_t iters, float* read_gbps, float* mix_gbps) {
    if (!s || iters == 0) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    if (s->busy) {
        c->err = "stream busy";
        return FFS_ERR_INVALID;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    const uint64_t n16 = (uint64_t)s->max_batch * c->L.frame_stride / 16;
    // the byte-mask buffer holds half as many bytes as the pixel buffer for 16-bit pixels, a quarter for 32-bit
    const uint64_t n16w = std::min<uint64_t>(n16, (uint64_t)s->max_batch * c->L.bytes_frame_stride / 8);
    const uint4* src = reinterpret_cast<const uint4*>(s->d_img);
    uint2* dst = reinterpret_cast<uint2*>(s->d_sbytes);
    float out[3] = {0, 0, 0};
    for (int mode = 0; mode < 3; ++mode) {
        const uint64_t n = mode ? n16w : n16;
        auto launch = [&]() {
            if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(4096), dim3(256), 0, s->st, src, dst, n, s->d_tile_counts);
            else if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(4096), dim3(256), 0, s->st, src, dst, n, s->d_tile_counts);
            else hipLaunchKernelGGL(k_probe<0>, dim3(4096), dim3(256), 0, s->st, src, dst, n, s->d_tile_counts);
        };
        launch();  // warm-up
        HIP_TRY(c, hipEventRecord(s->ev[0], s->st));
        for (uint32_t i = 0; i < iters; ++i) launch();
        HIP_TRY(c, hipEventRecord(s->ev[1], s->st));
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventSynchronize(s->ev[1]));
        float ms = 0;
        HIP_TRY(c, hipEventElapsedTime(&ms, s->ev[0], s->ev[1]));
        const double bytes = (double)n * 16.0 + (mode ? (double)n * 8.0 : 0.0);
        out[mode] = (float)(bytes * iters / (ms * 1e-3) / 1e9);
    }
    if (read_gbps) *read_gbps = out[0];
    if (mix_gbps) *mix_gbps = std::max(out[1], out[2]);  // the better of plain and non-temporal stores
    return FFS_OK;
}

__global__ void k_selftest_sqrt(unsigned long long begin, unsigned long long end, unsigned long long* out) {
    unsigned long long acc = 0;
    for (unsigned long long n = begin + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; n < end;
         n += (unsigned long long)gridDim.x * blockDim.x)
        acc += (unsigned long long)__double_as_longlong(__builtin_sqrt((double)n));
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

extern "C" int ffs_selftest_sqrt(ffs_ctx* c, uint64_t begin, uint64_t end, uint64_t* sum_of_bits) {
    if (!c || !sum_of_bits || end < begin) return FFS_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long* d = nullptr;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&d), 8));
    HIP_TRY(c, hipMemset(d, 0, 8));
    hipLaunchKernelGGL(k_selftest_sqrt, dim3(2048), dim3(256), 0, 0, begin, end, d);
    hipError_t e = hipMemcpy(sum_of_bits, d, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIP_TRY(c, e);
    return FFS_OK;
}

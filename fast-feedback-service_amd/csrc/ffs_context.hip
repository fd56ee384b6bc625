// ffs_context.hip -- contexts, masks, parameters and tuning, streams (see ffs_internal.hpp for the map of the library).
// The C ABI is declared in include/ffs_hip.h; every entry point there names the reference interface it replaces.
#include <sys/mman.h>

#include "ffs_internal.hpp"
#include "kernels_mask.hpp"

static_assert(sizeof(ReflOut) == sizeof(ffs_reflection), "record layout");
static_assert(offsetof(ReflOut, sum_intensity) == offsetof(ffs_reflection, sum_intensity), "record layout");

thread_local std::string g_create_error;

// ---- lifecycle (ffs_internal.hpp) -------------------------------------------------------------------------------
// The registry is never destructed (a leaked singleton): destroy calls may arrive from static destructors of the caller that run
// after this library's own.
static void ctx_destroy_internal(ffs_ctx* c);
namespace {
struct Registry {
    std::mutex mu;
    std::unordered_set<const void*> live[3];
    std::atomic<bool> exiting{false};
    bool exit_hook_installed = false;
};
Registry& registry() {
    static Registry* r = new Registry();
    return *r;
}
// Runs when the process exits (exit() or return from main) with the HIP runtime still up: installed by the first ffs_ctx_create,
// i.e. after the runtime registered its own exit handlers, and handlers run in reverse order.  Nothing is freed here -- the
// process is going away -- but nothing of ours may still be running when the runtime takes its queues and the pinned memory down:
// the helper threads of compressed submits and of ffs_wait's assembly are joined, and each context's device is drained (a sparse
// launch in flight writes its records straight into pinned host memory).
static void on_process_exit() {
    Registry& r = registry();
    std::vector<ffs_ctx*> ctxs;
    {
        std::lock_guard<std::mutex> lock(r.mu);
        r.exiting.store(true);
        for (const void* h : r.live[kHandleCtx]) ctxs.push_back(static_cast<ffs_ctx*>(const_cast<void*>(h)));
    }
    for (ffs_ctx* c : ctxs) {
        std::vector<ffs_stream*> streams;
        {
            std::lock_guard<std::mutex> lock(c->stream_mu);
            streams = c->live_streams;
        }
        for (ffs_stream* s : streams)
            if (s->job.joinable()) s->job.join();
        if (hipSetDevice(c->device) == hipSuccess) (void)hipDeviceSynchronize();
        (void)hipGetLastError();
        ahead_stop(c, false);   // (behind the drain: the thread may be waiting for a batch's last event)
        if (AssemblyPool* pool = c->assembly.load(std::memory_order_acquire)) pool->stop_and_join();
    }
}
}  // namespace

void handle_add(HandleKind kind, const void* h) {
    Registry& r = registry();
    std::lock_guard<std::mutex> lock(r.mu);
    r.live[kind].insert(h);
    if (!r.exit_hook_installed) {
        r.exit_hook_installed = true;
        std::atexit(on_process_exit);
    }
}
bool handle_take(HandleKind kind, const void* h) {
    Registry& r = registry();
    std::lock_guard<std::mutex> lock(r.mu);
    if (r.exiting.load()) return false;
    return r.live[kind].erase(h) != 0;
}
bool handle_live(HandleKind kind, const void* h) {
    Registry& r = registry();
    std::lock_guard<std::mutex> lock(r.mu);
    return r.live[kind].count(h) != 0;
}
bool process_exiting() { return registry().exiting.load(); }

extern "C" void ffs_default_params(ffs_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->min_count = 2;  // baseline/spotfinder/standalone.cc:17
    p->nsig_b = 6.0;   // :19
    p->nsig_s = 3.0;   // :20
    p->threshold = 0.0;
    p->max_valid = -1;
    p->min_spot_size = 3;     // spotfinder/spotfinder.cc:321
    p->min_spot_size_3d = 3;  // :327
    p->max_peak_centroid_separation = 2.0f;  // :335
    p->want_reflections = 1;
    p->want_strong_list = 0;
    p->want_strong_mask = 0;
    p->algorithm = FFS_ALGO_DISPERSION;
    p->extended_flavour = 0;
}

extern "C" int ffs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int ffs_device_name(int device, char* buf, size_t buflen) {
    hipDeviceProp_t prop;
    if (!buf || buflen == 0) return FFS_ERR_INVALID;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return FFS_ERR_NODEVICE;
    std::snprintf(buf, buflen, "%s (%s)", prop.name, prop.gcnArchName);
    return FFS_OK;
}

extern "C" int ffs_device_total_mem(int device, uint64_t* bytes) {
    hipDeviceProp_t prop;
    if (!bytes) return FFS_ERR_INVALID;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return FFS_ERR_NODEVICE;
    *bytes = prop.totalGlobalMem;
    return FFS_OK;
}

extern "C" const char* ffs_last_error(const ffs_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();  // both are per-thread texts
}

extern "C" int ffs_ctx_create(int device, uint32_t width, uint32_t height, int pixel_bytes,
                              uint32_t max_batch, uint32_t max_strong, ffs_ctx** out) {
    if (!out) return FFS_ERR_INVALID;
    *out = nullptr;
    if (width == 0 || height == 0 || (pixel_bytes != 2 && pixel_bytes != 4) || max_batch == 0
        || width > 10240 || (uint64_t)width * height >= (1ull << 32)) {
        g_create_error = "ffs_ctx_create: need 0 < width <= 10240, width*height < 2^32, pixel_bytes 2 or 4, max_batch > 0";
        return FFS_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        g_create_error = "no HIP device visible (libffs_hip.so has no CPU fallback)";
        return FFS_ERR_NODEVICE;
    }
    if (device < 0 || device >= ndev) {
        g_create_error = "ffs_ctx_create: device index out of range";
        return FFS_ERR_NODEVICE;
    }
    ffs_ctx* c = new (std::nothrow) ffs_ctx();
    if (!c) return FFS_ERR_NOMEM;
    c->device = device;
#ifdef FFS_EXPERIMENTS
    {   // timing experiments: from the environment, in this build only
        auto env_int = [](const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; };
        Tuning::Exp& x = c->tune.exp;
        x.k1_debug = env_int("FFS_EXP_K1_DEBUG", 0);
        x.chain_skip = env_int("FFS_EXP_CHAIN_SKIP", 0);
        x.chain_stop = env_int("FFS_EXP_CHAIN_STOP", 0);
        x.dummy_us = env_int("FFS_EXP_DUMMY_US", 0);
        x.dummy_wg = std::max(1, env_int("FFS_EXP_DUMMY_WG", 32));
        x.dummy_threads = std::max(64, std::min(1024, env_int("FFS_EXP_DUMMY_THREADS", 1024)));
        x.dummy_lds = std::max(0, std::min(65536, env_int("FFS_EXP_DUMMY_LDS", 0)));
    }
#endif
    c->pixel_bytes = pixel_bytes;
    c->max_batch = max_batch;
    Layout& L = c->L;
    L.W = (int)width;
    L.H = (int)height;
    L.pitch_px = round_up((int)width, 128);  // byte-mask rows start on 128-byte lines
    L.pitch = (uint32_t)L.pitch_px * (uint32_t)pixel_bytes;
    L.mpitch = (uint32_t)L.pitch_px / 8;
    L.bpitch = (uint32_t)L.pitch_px;
    L.frame_stride = (uint64_t)L.pitch * height;
    L.plane_frame_stride = (uint64_t)L.mpitch * height;
    L.bytes_frame_stride = (uint64_t)L.bpitch * height;
    if (L.frame_stride >= (1ull << 32)) {
        g_create_error = "frame larger than 4 GiB";
        delete c;
        return FFS_ERR_INVALID;
    }
    const uint64_t npx = (uint64_t)width * height;
    c->cap = max_strong ? max_strong : (uint32_t)std::min<uint64_t>(npx, 1u << 18);
    c->cap = (uint32_t)std::min<uint64_t>(c->cap, npx);
    c->max_comp = std::min<uint32_t>(c->cap, 1u << 14);   // (a frame with more is run again with room for it, as for the lists)
    c->n_tiles = ((int)height + kTileRows - 1) / kTileRows;
    ffs_default_params(&c->params);
    if (hipSetDevice(device) != hipSuccess) {
        g_create_error = "hipSetDevice failed";
        delete c;
        return FFS_ERR_DEVICE;
    }
    hipError_t e = hipMalloc(&c->d_maskbits, L.plane_frame_stride + 256);
    // one dword per lane group of 16 bytes of pixels: 8 pixels (16-bit) or 4 pixels (32-bit)
    if (e == hipSuccess) e = hipMalloc(&c->d_ginfo, (size_t)(L.H + kInfoExtraRows) * ((size_t)L.pitch_px * pixel_bytes / 4) + 256);
    if (e == hipSuccess) e = hipMalloc(&c->d_mmap, (size_t)L.H * L.pitch_px + 256);
    if (e != hipSuccess) {
        g_create_error = std::string("hipMalloc(mask): ") + hipGetErrorString(e);
        ctx_destroy_internal(c);
        return FFS_ERR_NOMEM;
    }
    // the hot path's code object is loaded here, not by the first worker's first stream (~30 ms), and k_frame_chain's
    // dynamic LDS is asked for once per device
    c->chain_ok = chain_prepare_device();
    {   // ... and so are the context's shared HIP streams (a hardware queue takes milliseconds to create): DESIGN.md section 3.4
        int lo = 0, hi = 0;
        hipError_t es = hipDeviceGetStreamPriorityRange(&lo, &hi);  // (least, greatest)
        if (es == hipSuccess) es = hipStreamCreateWithPriority(&c->dense_st, hipStreamNonBlocking, (lo + hi) / 2);
        if (es == hipSuccess) es = hipStreamCreateWithFlags(&c->up_st, hipStreamNonBlocking);
        for (auto& sp : c->sparse_st)
            if (es == hipSuccess) es = hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, hi);
        if (es == hipSuccess)
            for (auto& e : c->chain_ev)
                if (es == hipSuccess) es = hipEventCreate(&e);
        if (es != hipSuccess) {
            g_create_error = std::string("creating the context's HIP streams: ") + hipGetErrorString(es);
            ctx_destroy_internal(c);
            return FFS_ERR_DEVICE;
        }
    }
    int rc = ffs_ctx_set_mask(c, nullptr);
    if (rc != FFS_OK) {
        g_create_error = c->err;
        ctx_destroy_internal(c);
        return rc;
    }
    handle_add(kHandleCtx, c);   // (the runtime is up by now: the exit handler this installs runs before the runtime's own)
    *out = c;
    return FFS_OK;
}

// ---- helper threads of ffs_wait's result assembly (ffs_internal.hpp) ---------------------------------------------------
void AssemblyPool::start(int n_threads) {
    for (int t = 0; t < n_threads; ++t)
        threads.emplace_back([this] {
            uint64_t seen = 0;
            for (;;) {
                const std::function<void(uint32_t)>* fn = nullptr;
                uint32_t n = 0;
                {
                    std::unique_lock<std::mutex> lock(mu);
                    cv.wait(lock, [&] { return stop || generation != seen; });
                    if (stop) return;
                    seen = generation;
                    fn = job;
                    n = n_items;
                    if (fn) active.fetch_add(1);   // (under the lock: run() clears `job` under it before it waits for `active`)
                }
                if (!fn) continue;
                for (;;) {
                    const uint32_t i = next.fetch_add(1);
                    if (i >= n) break;
                    (*fn)(i);
                    done.fetch_add(1, std::memory_order_release);
                }
                active.fetch_sub(1, std::memory_order_release);
            }
        });
}
void AssemblyPool::run(uint32_t n, const std::function<void(uint32_t)>& fn) {
    {
        std::lock_guard<std::mutex> lock(mu);
        job = &fn;
        n_items = n;
        next.store(0);
        done.store(0);
        ++generation;
    }
    cv.notify_all();
    for (;;) {   // the caller works too: with sleepy helpers this is the whole job at the speed of one thread
        const uint32_t i = next.fetch_add(1);
        if (i >= n) break;
        fn(i);
        done.fetch_add(1, std::memory_order_release);
    }
    while (done.load(std::memory_order_acquire) < n) __builtin_ia32_pause();   // (the helpers' last items: microseconds)
    {
        std::lock_guard<std::mutex> lock(mu);
        job = nullptr;      // a helper that wakes late finds no job ...
    }
    while (active.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();   // ... and those that took this one have left it: `fn` may go
}
void AssemblyPool::stop_and_join() {
    {
        std::lock_guard<std::mutex> lock(mu);
        stop = true;
    }
    cv.notify_all();
    for (auto& t : threads)
        if (t.joinable()) t.join();   // (a helper inside a job finishes its items first)
}
AssemblyPool::~AssemblyPool() { stop_and_join(); }

extern "C" void ffs_ctx_destroy(ffs_ctx* c) {
    if (!c || !handle_take(kHandleCtx, c)) return;   // (destroyed already, or the process is exiting: ffs_internal.hpp, lifecycle)
    // the streams and stacks the caller still holds go first: each keeps a pointer to this context.  Their handles leave the registry
    // here, so the caller's own destroy calls on them -- in whatever order its teardown makes them -- find nothing to do.
    std::vector<ffs_stream*> streams;
    std::vector<ffs_stack3d*> stacks;
    {
        std::lock_guard<std::mutex> lock(c->stream_mu);
        streams.swap(c->live_streams);
        stacks.swap(c->live_stacks);
    }
    for (ffs_stream* s : streams)
        if (handle_take(kHandleStream, s)) stream_destroy_internal(s);
    for (ffs_stack3d* st : stacks)
        if (handle_take(kHandleStack, st)) {
            g_live_stacks.fetch_sub(1);
            stack3d_free(st);
        }
    ctx_destroy_internal(c);
}

static void ctx_destroy_internal(ffs_ctx* c) {
    (void)hipSetDevice(c->device);
    ahead_stop(c, true);
    for (auto* st : c->stack_pool) stack3d_free(st);
    c->stack_pool.clear();
    if (AssemblyPool* pool = c->assembly.exchange(nullptr)) {
        pool->owner.lock();     // (a wait of another thread that is using the helpers -- the caller's mistake -- is let finish)
        pool->owner.unlock();
        delete pool;            // joins the helpers
    }
    if (c->d_maskbits) (void)hipFree(c->d_maskbits);
    if (c->d_ginfo) (void)hipFree(c->d_ginfo);
    if (c->d_mmap) (void)hipFree(c->d_mmap);
    if (c->dense_st) (void)hipStreamDestroy(c->dense_st);
    if (c->dense_st2) (void)hipStreamDestroy(c->dense_st2);
    if (c->d_handoff) (void)hipFree(c->d_handoff);
    if (c->up_st) (void)hipStreamDestroy(c->up_st);
    for (auto st : c->sparse_st) if (st) (void)hipStreamDestroy(st);
    for (auto e : c->chain_ev) if (e) (void)hipEventDestroy(e);
    for (auto& pb : c->pinned_pool) pinned_free(pb);
    delete c;
}

// The tables of the one-kernel threshold path depend on the mask alone: rebuilt whenever it changes.
static int rebuild_mask_tables(ffs_ctx* c) {
    const Layout& L = c->L;
    const uint32_t gpitch = (uint32_t)L.pitch_px * (uint32_t)c->pixel_bytes / 4;
    HIP_TRY(c, hipMemset(c->d_ginfo, 0, (size_t)(L.H + kInfoExtraRows) * gpitch));
    const int groups = L.pitch_px / (c->pixel_bytes == 2 ? 8 : 4);
    (void)hipGetLastError();  // drop any stale error state: the check below is for this launch
    if (c->pixel_bytes == 2)
        hipLaunchKernelGGL(k_build_maps, dim3((groups + 255) / 256, L.H), dim3(256), 0, 0, c->d_maskbits, L.mpitch, L.W, L.H,
                           L.pitch_px, c->d_mmap, c->d_ginfo, gpitch);
    else
        hipLaunchKernelGGL(k_build_maps4, dim3((groups + 255) / 256, L.H), dim3(256), 0, 0, c->d_maskbits, L.mpitch, L.W, L.H,
                           L.pitch_px, c->d_mmap, c->d_ginfo, gpitch);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipDeviceSynchronize());
    return FFS_OK;
}

static int ffs_ctx_set_mask_impl(ffs_ctx* c, const uint8_t* host_mask) {
    if (!c) return FFS_ERR_INVALID;
    const Layout& L = c->L;
    // bytes -> bits, eight pixels at a time (18 M pixels one by one were 30 ms of every process's start: the driver uploads a mask
    // per request, spotfinder.cc:61-108): a byte's "non-zero" into its top bit, the eight top bits gathered by one multiply
    std::vector<uint8_t> bits(L.plane_frame_stride, 0);
    const int full = L.W / 8;
    for (int y = 0; y < L.H; ++y) {
        uint8_t* row = bits.data() + (size_t)y * L.mpitch;
        if (host_mask) {
            const uint8_t* m = host_mask + (size_t)y * L.W;
            for (int g = 0; g < full; ++g) {
                uint64_t v;
                std::memcpy(&v, m + 8 * g, 8);
                const uint64_t nz = (((v & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | v) & 0x8080808080808080ull;   // top bit of every non-zero byte
                row[g] = (uint8_t)(((nz >> 7) * 0x0102040810204080ull) >> 56);                                      // byte i -> bit i
            }
            for (int x = full * 8; x < L.W; ++x)
                if (m[x]) row[x >> 3] |= (uint8_t)(1u << (x & 7));
        } else {
            std::memset(row, 0xFF, (size_t)full);
            for (int x = full * 8; x < L.W; ++x) row[x >> 3] |= (uint8_t)(1u << (x & 7));
        }
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(c->d_maskbits, bits.data(), bits.size(), hipMemcpyHostToDevice));
    return rebuild_mask_tables(c);
}

static int ffs_ctx_get_mask_impl(ffs_ctx* c, uint8_t* host_mask) {
    if (!c || !host_mask) return FFS_ERR_INVALID;
    const Layout& L = c->L;
    std::vector<uint8_t> bits(L.plane_frame_stride);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(bits.data(), c->d_maskbits, bits.size(), hipMemcpyDeviceToHost));
    for (int y = 0; y < L.H; ++y)
        for (int x = 0; x < L.W; ++x)
            host_mask[(size_t)y * L.W + x] = (bits[(size_t)y * L.mpitch + (x >> 3)] >> (x & 7)) & 1;
    return FFS_OK;
}

extern "C" int ffs_ctx_apply_resolution_mask(ffs_ctx* c, float wavelength, float distance_m,
                                             float bcx, float bcy, float psx, float psy, float dmin,
                                             float dmax) {
    if (!c) return FFS_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const Layout& L = c->L;
    dim3 block(64, 1), grid((L.mpitch + 63) / 64, L.H);
    hipLaunchKernelGGL(k_resolution_mask, grid, block, 0, 0, c->d_maskbits, L.mpitch, L.W, L.H,
                       wavelength, distance_m, bcx, bcy, psx, psy, dmin, dmax);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipDeviceSynchronize());
    return rebuild_mask_tables(c);
}

extern "C" int ffs_ctx_set_params(ffs_ctx* c, const ffs_params* p) {
    if (!c || !p) return FFS_ERR_INVALID;
    if (p->min_count < 2 || p->min_count > 49 || p->nsig_b < 0 || p->nsig_s < 0 || p->threshold < 0) {
        c->err = "ffs_ctx_set_params: need 2 <= min_count <= 49, nsig_b >= 0, nsig_s >= 0, threshold >= 0";
        return FFS_ERR_INVALID;  // the asserts of standalone.cc:52-63
    }
    if ((p->algorithm != FFS_ALGO_DISPERSION && p->algorithm != FFS_ALGO_DISPERSION_EXTENDED)
        || (p->extended_flavour != 0 && p->extended_flavour != 1)) {
        c->err = "ffs_ctx_set_params: unknown algorithm / extended_flavour";
        return FFS_ERR_INVALID;
    }
    c->params = *p;
    return FFS_OK;
}

extern "C" int ffs_ctx_device_layout(const ffs_ctx* c, size_t* pitch, size_t* fstride) {
    if (!c) return FFS_ERR_INVALID;
    if (pitch) *pitch = c->L.pitch;
    if (fstride) *fstride = c->L.frame_stride;
    return FFS_OK;
}

// ---- tuning (ffs_internal.hpp: Tuning) -------------------------------------------------------------------------
extern "C" int ffs_ctx_set_tuning(ffs_ctx* c, const char* key, long long value) {
    if (!c || !key) return FFS_ERR_INVALID;
    Tuning& t = c->tune;
    const std::string k = key;
    auto in = [&](long long lo, long long hi) { return value >= lo && value <= hi; };
    bool ok = true;
    if (k == "threshold_path") { if ((ok = in(0, 2))) t.threshold_path = (int)value; }
    else if (k == "ext_first_pass") { if ((ok = value == 0 || value == 2)) t.ext_first_pass = (int)value; }
    else if (k == "sparse_stage") { if ((ok = in(1, 3))) t.sparse_stage = (int)value; }
    else if (k == "device_lists") { if ((ok = in(0, 2))) t.device_lists = (int)value; }
    else if (k == "strong_log") { if ((ok = in(0, 1))) t.strong_log = (int)value; }
    else if (k == "chain_runs") { if ((ok = in(0, 2))) t.chain_runs = (int)value; }
    else if (k == "dense_overlap") { if ((ok = in(0, 1))) t.dense_overlap = (int)value; }
    else if (k == "stream_prio") { if ((ok = in(0, 3))) t.stream_prio = (int)value; }
    else if (k == "wait_ahead") { if ((ok = in(0, 1))) t.wait_ahead = (int)value; }
    else if (k == "assembly_threads") { if ((ok = in(1, 31))) t.assembly_threads = (int)value; }
    else if (k == "sparse_bands") { if ((ok = in(0, 2))) t.sparse_bands = (int)value; }
    else if (k == "sparse_priority") {
        // priority of the context's two sparse HIP streams: 0 = highest (default), 1 = lowest, 2 = the dense stream's; before the first stream is created
        if ((ok = in(0, 2) && c->n_streams_made == 0)) {
            int lo = 0, hi = 0;
            HIP_TRY(c, hipSetDevice(c->device));
            HIP_TRY(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
            const int prio = value == 0 ? hi : value == 1 ? lo : (lo + hi) / 2;
            for (auto& sp : c->sparse_st) {
                if (sp) (void)hipStreamDestroy(sp);
                sp = nullptr;
                HIP_TRY(c, hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, prio));
            }
            t.sparse_priority = (int)value;
        }
    }
    else if (k == "sched") { if ((ok = (value == 0 || value == 3) && c->n_streams_made == 0)) t.sched = (int)value; }
    else if (k == "chain_first") { if ((ok = in(0, 64))) t.chain_first = (int)value; }
    else if (k == "bright_cap") { if ((ok = in(0, kBrightCap))) t.bright_cap = (int)value; }
    else if (k == "frames_per_group") { if ((ok = in(1, 1 << 30))) t.frames_per_group = (int)value; }
    else if (k == "target_waves") { if ((ok = in(1, 1 << 24))) t.target_waves = value; }
    else if (k == "stream_bands") { if ((ok = in(0, 4096))) t.stream_bands = (int)value; }
    else if (k == "dense_mask") { if ((ok = in(0, 1))) t.dense_mask = (int)value; }
    else if (k == "occupancy_bitmap") { if ((ok = in(0, 1))) t.occupancy_bitmap = (int)value; }
    else if (k == "direct_records") { if ((ok = in(0, 1) && c->n_streams_made == 0)) t.direct_records = (int)value; }
    else if (k == "decode_in_dense_stream") { if ((ok = in(0, 1))) t.decode_in_dense_stream = (int)value; }
    else if (k == "ext_fused") { if ((ok = in(0, 1))) t.ext_fused = (int)value; }
    else if (k == "ext_erode") { if ((ok = in(0, 2))) t.ext_erode = (int)value; }
    else if (k == "ext_e_sparse") { if ((ok = in(0, 1))) t.ext_e_sparse = (int)value; }
    else if (k == "ext_rest_aside") { if ((ok = in(0, 1))) t.ext_rest_aside = (int)value; }
    else if (k == "band_taper") { if ((ok = in(0, 99))) t.band_taper = (int)value; }
    else if (k == "rows_ahead") { if ((ok = in(2, 4))) t.rows_ahead = (int)value; }
    else if (k == "ccl_grid") { if ((ok = in(1, 1024))) t.ccl_grid = (int)value; }
    else {
        c->err = "ffs_ctx_set_tuning: unknown key '" + k + "'";
        return FFS_ERR_INVALID;
    }
    if (!ok) {
        c->err = "ffs_ctx_set_tuning: value out of range for '" + k + "' (sched / direct_records: before the first stream is created)";
        return FFS_ERR_INVALID;
    }
    return FFS_OK;
}

// ---- streams ---------------------------------------------------------------------------------------

extern "C" void ffs_stream_destroy(ffs_stream* s) {
    if (!s || !handle_take(kHandleStream, s)) return;   // (destroyed already -- by its context's ffs_ctx_destroy, or twice)
    {
        ffs_ctx* c = s->ctx;   // alive: a context takes its streams with it, and this one was still in the registry
        std::lock_guard<std::mutex> lock(c->stream_mu);
        auto& v = c->live_streams;
        v.erase(std::remove(v.begin(), v.end(), s), v.end());
    }
    stream_destroy_internal(s);
}

void stream_destroy_internal(ffs_stream* s) {
    (void)hipSetDevice(s->ctx->device);
    if (s->job.joinable()) s->job.join();
    (void)ahead_take(s);   // (a batch in flight that the context's own thread is assembling: let it finish with the stream)
    gather_scratch_free(s);
    mark_idle(s);   // (a stream may be closed with its batch still in flight)
#ifdef FFS_EXPERIMENTS
    if (s->phase_n) {
        const double k = 1.0 / (double)s->phase_n;
        std::fprintf(stderr, "[ffs exp] sparse launch, %llu frames: init %.1f | L1 / E %.1f | S + L2 %.1f | X U %.1f | P %.1f | R %.1f | whole %.1f us per frame\n",
                     s->phase_n, s->phase_sum[0] * k, s->phase_sum[1] * k, s->phase_sum[2] * k, s->phase_sum[3] * k, s->phase_sum[4] * k, s->phase_sum[5] * k, s->phase_sum[6] * k);
    }
    if (s->h_phase_ts) (void)hipHostFree(s->h_phase_ts);
#endif
    if (s->big) stream_destroy_internal(s->big);
    if (s->st_up && s->st_up != s->st) (void)hipStreamSynchronize(s->st_up);
    if (s->st) (void)hipStreamSynchronize(s->st);
    if (s->st_shared && s->ctx->dense_st2) (void)hipStreamSynchronize(s->ctx->dense_st2);   // (the dense stream's partner may hold this stream's kernel)
    if (s->st2 && s->st2 != s->st) { (void)hipStreamSynchronize(s->st2); if (!s->st2_shared) (void)hipStreamDestroy(s->st2); }
    // (d_n_comp, d_summary and d_overflow live inside the d_num_strong allocation)
    if (s->h_pack_tab) (void)hipHostFree(s->h_pack_tab);
    // (the stream's device buffers are one slab; what is allocated on first use is freed by itself)
    void* dev[] = {s->d_slab, s->d_pack_k, s->d_pack_i, s->d_pack_tab, s->d_comp, s->d_tab, s->d_ext_pair[0], s->d_ext_pair[1], s->d_wlog, s->d_wlog_n, s->d_wpix,
                   s->d_band_hdr, s->d_band_acc, s->d_band_seam};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    void* host[] = {s->h_tab, s->h_counts, s->h_recs, s->h_list_k, s->h_list_i, s->h_mask};
    for (void* p : host)
        if (p) (void)hipHostFree(p);
    if (s->h_img) {   // kept for the next stream of the context
        std::lock_guard<std::mutex> lock(s->ctx->stream_mu);
        s->ctx->pinned_pool.push_back(s->h_img_buf);
    }
    for (auto& e : s->ev)
        if (e) (void)hipEventDestroy(e);
    if (s->ev_pack) (void)hipEventDestroy(s->ev_pack);
    if (s->ev_sent) (void)hipEventDestroy(s->ev_sent);
    // (the start events of its sparse launches belong to the context: a thread about to wait on one is safe)
    if (s->st && !s->st_shared) (void)hipStreamDestroy(s->st);
    delete s;
}

extern "C" int ffs_stream_create(ffs_ctx* c, ffs_stream** out) {
    if (!c || !out) return FFS_ERR_INVALID;
    const int rc = stream_create_sized(c, c->max_batch, c->cap, c->max_comp, out);
    if (rc != FFS_OK) return rc;
    {
        std::lock_guard<std::mutex> lock(c->stream_mu);
        c->live_streams.push_back(*out);
    }
    handle_add(kHandleStream, *out);
    return FFS_OK;
}

int stream_create_sized(ffs_ctx* c, uint32_t max_batch, uint32_t cap, uint32_t max_comp, ffs_stream** out) {
    *out = nullptr;
    HIP_TRY(c, hipSetDevice(c->device));
    ffs_stream* s = new (std::nothrow) ffs_stream();
    if (!s) return FFS_ERR_NOMEM;
    s->ctx = c;
    s->max_batch = max_batch;
    s->cap = cap;
    s->max_comp = max_comp;
    const Layout& L = c->L;
    const size_t B = s->max_batch;
#define STREAM_TRY(expr)                                                        \
    do {                                                                        \
        hipError_t e_ = (expr);                                                 \
        if (e_ != hipSuccess) {                                                 \
            c->err = std::string(#expr) + ": " + hipGetErrorString(e_);         \
            stream_destroy_internal(s);                                         \
            return e_ == hipErrorOutOfMemory ? FFS_ERR_NOMEM : FFS_ERR_DEVICE;  \
        }                                                                       \
    } while (0)
    // HIP streams.  The sparse stage (compaction, union-find, reductions) is latency-bound and keeps a
    // few CUs busy; the streaming kernel wants the whole machine.  sched 3 (default): every streaming kernel of the context
    // goes through ONE stream (first in, first out: nothing of another batch behind which it could queue), the sparse work
    // of a batch is one launch on one of two shared high-priority streams, uploads and decoding have a stream of their own
    // -- four hardware queues in all.  sched 0: one HIP stream per ffs_stream for everything.
    // (Measured and dropped: CU masks for the two stages, a stream per batch for the sparse work, two dense streams.)
    {
        std::lock_guard<std::mutex> lock(c->stream_mu);
        if (c->tune.sched >= 3) {   // (the shared streams were created with the context)
            s->st = c->dense_st;
            s->st_shared = true;
            s->st_up = c->up_st;
            s->st2 = c->sparse_st[c->n_streams_made & 1];
            s->st2_shared = true;
        } else {
            STREAM_TRY(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
            s->st2 = s->st;
        }
        ++c->n_streams_made;
    }
    if (!s->st_up) s->st_up = s->st;
    for (auto& e : s->ev) STREAM_TRY(hipEventCreate(&e));
    {   // ONE device allocation per stream, carved up: hipMalloc is cheap here, but every hipFree costs ~0.25 ms and a
        // driver that closes 16 streams at the end of a run paid 16 x 20 of them (tools/ubench/alloc_cost.hip)
        size_t at = 0;
        auto carve = [&](size_t bytes) { const size_t o = at; at += (bytes + 255) & ~(size_t)255; return o; };
        const size_t o_img = carve(B * L.frame_stride), o_bits = carve(B * L.plane_frame_stride), o_sbytes = carve(B * L.bytes_frame_stride);
        const size_t o_counts = carve(tile_counts_bytes(s)), o_occ = carve(B * (size_t)occ_frame_words(L) * 4);
        const size_t o_bright = carve((size_t)kBrightCap * sizeof(uint2));
        // per-frame counters in the layout of h_counts, so that one copy brings them all back:
        // [B] strong pixels | [B] components | [B][8] summary | [1] overflow / error flag
        const size_t o_ns = carve((B * 10 + 1) * 4), o_row = carve(B * (size_t)(L.H + 1) * 4);
        const size_t o_k = carve(B * (size_t)s->cap * 4), o_i = carve(B * (size_t)s->cap * 4), o_par = carve(B * (size_t)s->cap * 4);
        const size_t o_acc = carve(B * (size_t)s->cap * sizeof(CompAcc2)), o_roots = carve(B * (size_t)(s->cap / 512 + 1) * 4);
        const size_t o_recs = carve(B * (size_t)s->max_comp * sizeof(WireRec2));
        STREAM_TRY(dmalloc(&s->d_slab, at));
        uint8_t* base = s->d_slab;
        s->d_img = base + o_img;
        s->d_bits = base + o_bits;
        s->d_sbytes = base + o_sbytes;
        s->d_tile_counts = reinterpret_cast<uint32_t*>(base + o_counts);
        s->d_occ = reinterpret_cast<uint32_t*>(base + o_occ);
        s->d_bright = reinterpret_cast<uint2*>(base + o_bright);
        s->d_num_strong = reinterpret_cast<uint32_t*>(base + o_ns);
        s->d_n_comp = s->d_num_strong + B;
        s->d_summary = s->d_num_strong + 2 * B;
        s->d_overflow = s->d_num_strong + 10 * B;
        s->d_row_off = reinterpret_cast<uint32_t*>(base + o_row);
        s->d_list_k = reinterpret_cast<uint32_t*>(base + o_k);
        s->d_list_i = reinterpret_cast<uint32_t*>(base + o_i);
        s->d_parent = reinterpret_cast<uint32_t*>(base + o_par);
        s->d_acc2 = reinterpret_cast<CompAcc2*>(base + o_acc);
        s->d_chunk_roots = reinterpret_cast<uint32_t*>(base + o_roots);
        s->d_recs = reinterpret_cast<ReflOut*>(base + o_recs);
#ifdef FFS_EXPERIMENTS
        if (std::getenv("FFS_EXP_PRINT_ADDR"))   // (kernel time follows where the buffers lie: see DESIGN.md section 3.2)
            std::fprintf(stderr, "ffs addresses: maskbits %p ginfo %p mmap %p | slab %p (%zu MB) bits %p counts %p occ %p bright %p\n", (void*)c->d_maskbits,
                         (void*)c->d_ginfo, (void*)c->d_mmap, (void*)base, at >> 20, (void*)s->d_bits, (void*)s->d_tile_counts, (void*)s->d_occ, (void*)s->d_bright);
#endif
    }
    // (the pinned staging area for frames / chunks is allocated on first use: ensure_host_staging)
    STREAM_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_counts), (B * 11 + 1) * 4, hipHostMallocDefault));  // (+ [B] per-frame flags, k_frame_chain)
    std::memset(s->h_counts, 0, (B * 11 + 1) * 4);
    STREAM_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_recs), B * (size_t)s->max_comp * sizeof(WireRec2),
                             hipHostMallocDefault));
    // The sparse kernels write the (few MB of) records and the counters straight into these pinned, device-visible
    // buffers: no copy follows them.  Tuning "direct_records" = 0 keeps the device buffers + copies (A/B).
    s->direct_recs = c->tune.direct_records != 0;
    if (s->direct_recs
        && hipHostGetDevicePointer(reinterpret_cast<void**>(&s->h_recs_dev), s->h_recs, 0) != hipSuccess) {
        (void)hipGetLastError();
        s->direct_recs = false;
    }
    if (s->direct_recs && hipHostGetDevicePointer(reinterpret_cast<void**>(&s->h_counts_dev), s->h_counts, 0) != hipSuccess) {
        (void)hipGetLastError();
        s->h_counts_dev = nullptr;
    }
    STREAM_TRY(hipMemsetAsync(s->d_overflow, 0, 4, s->st));
    STREAM_TRY(hipMemsetAsync(s->d_occ, 0, B * (size_t)occ_frame_words(L) * 4, s->st));
    // bits beyond the image width (x >= W up to the row pitch) are never written by the threshold kernels and must read 0
    STREAM_TRY(hipMemsetAsync(s->d_bits, 0, B * L.plane_frame_stride, s->st));
    STREAM_TRY(hipStreamSynchronize(s->st));
#undef STREAM_TRY
    *out = s;
    return FFS_OK;
}

size_t default_staging_bytes(const ffs_stream* s) {
    // raw frames, or bitshuffle-LZ4 chunks (which can exceed the raw size by < 1 % when incompressible)
    const size_t frame = (size_t)s->ctx->L.W * s->ctx->L.H * s->ctx->pixel_bytes;
    return (size_t)s->max_batch * (frame + frame / 128 + 4096);
}

PinnedBuf pinned_alloc(size_t bytes) {
    PinnedBuf b;
    constexpr size_t kHuge = (size_t)2 << 20;
    const size_t len = (bytes + kHuge - 1) / kHuge * kHuge + kHuge;
    void* base = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base != MAP_FAILED) {
        uint8_t* p = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(base) + kHuge - 1) / kHuge * kHuge);
        (void)madvise(p, len - kHuge, MADV_HUGEPAGE);
        for (size_t o = 0; o < bytes; o += 4096) p[o] = 0;   // first touch here: on the calling thread's NUMA node
        if (hipHostRegister(p, (bytes + 4095) & ~(size_t)4095, hipHostRegisterDefault) == hipSuccess) {
            b.p = p;
            b.bytes = bytes;
            b.map_base = base;
            b.map_len = len;
            return b;
        }
        (void)hipGetLastError();
        munmap(base, len);
    }
    uint8_t* p = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&p), bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return b;
    }
    b.p = p;
    b.bytes = bytes;
    return b;
}

void pinned_free(PinnedBuf& b) {
    if (!b.p) return;
    if (b.map_base) {
        (void)hipHostUnregister(b.p);
        munmap(b.map_base, b.map_len);
    } else {
        (void)hipHostFree(b.p);
    }
    b = PinnedBuf{};
}

int ensure_host_staging(ffs_stream* s, size_t bytes) {
    ffs_ctx* c = s->ctx;
    if (s->h_img && s->h_img_bytes >= bytes) return FFS_OK;
    if (s->busy) {
        c->err = "the stream's host buffer cannot grow while a batch is in flight";
        return FFS_ERR_INVALID;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    {
        std::lock_guard<std::mutex> lock(c->stream_mu);
        if (s->h_img) {
            c->pinned_pool.push_back(s->h_img_buf);
            s->h_img = nullptr;
            s->h_img_bytes = 0;
            s->h_img_buf = PinnedBuf{};
        }
        int best = -1;   // the smallest pooled buffer that is large enough
        for (size_t i = 0; i < c->pinned_pool.size(); ++i)
            if (c->pinned_pool[i].bytes >= bytes && (best < 0 || c->pinned_pool[i].bytes < c->pinned_pool[(size_t)best].bytes)) best = (int)i;
        if (best >= 0) {
            s->h_img_buf = c->pinned_pool[(size_t)best];
            c->pinned_pool.erase(c->pinned_pool.begin() + best);
        } else if (!c->pinned_pool.empty()) {   // nothing fits: give one too-small buffer back rather than hoarding both
            pinned_free(c->pinned_pool.back());
            c->pinned_pool.pop_back();
        }
    }
    if (!s->h_img_buf.p) s->h_img_buf = pinned_alloc(bytes);   // (outside the lock: page faults of several threads run side by side)
    if (!s->h_img_buf.p) {
        c->err = "allocation of the stream's pinned staging buffer (" + std::to_string(bytes >> 20) + " MiB) failed";
        return FFS_ERR_NOMEM;
    }
    s->h_img = s->h_img_buf.p;
    s->h_img_bytes = s->h_img_buf.bytes;
    return FFS_OK;
}

extern "C" int ffs_stream_host_buffer(ffs_stream* s, void** ptr, size_t* bytes) {
    if (!s) return FFS_ERR_INVALID;
    if (!s->h_img) {
        const int rc = ensure_host_staging(s, default_staging_bytes(s));
        if (rc != FFS_OK) return rc;
    }
    if (ptr) *ptr = s->h_img;
    if (bytes) *bytes = s->h_img_bytes;
    return FFS_OK;
}

extern "C" int ffs_stream_reserve_host(ffs_stream* s, size_t bytes) {
    if (!s || bytes == 0) return FFS_ERR_INVALID;
    return ensure_host_staging(s, bytes);
}

// ---- guarded entry points (the mask conversions grow std::vectors) ---------------------------------------------
extern "C" int ffs_ctx_set_mask(ffs_ctx* c, const uint8_t* host_mask) {
    return guarded(c, [&] { return ffs_ctx_set_mask_impl(c, host_mask); });
}
extern "C" int ffs_ctx_get_mask(ffs_ctx* c, uint8_t* host_mask) {
    return guarded(c, [&] { return ffs_ctx_get_mask_impl(c, host_mask); });
}

// ffs_wait.hip -- ffs_wait: the batch's counters and records are on the host when its last event has fired; frames that
// did not fit the stream's lists are run again, results are assembled, and the accessors hand them out.
// Reference: what follows the kernel in spotfinder/spotfinder.cc:887-1008 (D2H, ConnectedComponents, JSON fields).
#include "ffs_internal.hpp"

static int ensure_list_host(ffs_stream* s) {
    ffs_ctx* c = s->ctx;
    if (!s->h_list_k) {
        const size_t bytes = (size_t)s->max_batch * s->cap * 4;
        HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&s->h_list_k), bytes, hipHostMallocDefault));
        HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&s->h_list_i), bytes, hipHostMallocDefault));
    }
    return FFS_OK;
}

// Turns the batch's wire records and summaries (pinned host memory, complete) into the arrays ffs_wait hands out.  Runs on the
// caller's thread inside ffs_wait, or ahead of it on the context's AheadThread (then into the stream's *_n arrays).
// `ovf`: the frames of this batch that were run again on the one-frame stream (only ffs_wait's own path has any).
static int assemble_batch(ffs_stream* s, uint64_t total_recs, const std::vector<OverflowFrame>* ovf, std::vector<ffs_frame_result>& out_results,
                          std::vector<ffs_box>& out_boxes, std::vector<ffs_reflection>& out_refls, std::vector<float>& out_centres) {
    ffs_ctx* c = s->ctx;
    const uint32_t n = s->n_frames;
    const size_t B = s->max_batch;
    const Layout& L = c->L;
    const ffs_params& p = s->batch_params;
    const uint32_t* h_ns = s->h_counts;
    const uint32_t* h_nc = s->h_counts + B;
    const uint32_t* h_sm = s->h_counts + 2 * B;
    auto overflow_frame = [&](uint32_t f) -> const OverflowFrame* {
        if (ovf)
            for (const OverflowFrame& o : *ovf)
                if (o.frame == f) return &o;
        return nullptr;
    };
    // assemble: boxes = components surviving the min-size filter (connected_components.cc:122-135),
    // reflections = components surviving filter_reflections (:207-236); both keep label order.
    // (written through raw pointers into arrays sized for the worst case: 45 000 records per batch of the bench frames, and the
    // last wait of a run does this on the clock -- push_back's capacity check per record was a third of it)
    out_results.assign(n, ffs_frame_result{});
    const uint32_t min_size = p.min_spot_size;
    const bool want_refl = p.want_reflections != 0;
    // Where each frame's boxes / reflections / records lie follows from the per-frame summaries the sparse stage wrote (boxes =
    // summary[0], reflections = summary[2]; a frame that was re-run on the one-frame stream: what that run returned), so the
    // frames can be assembled independently -- by the caller and the context's helper threads (AssemblyPool) when the batch is large.
    std::vector<size_t> box_at(n + 1), refl_at(n + 1), rec_at(n);
    {
        size_t nb = 0, nr = 0, at = 0;
        for (uint32_t f = 0; f < n; ++f) {
            box_at[f] = nb;
            refl_at[f] = nr;
            const uint32_t nc = std::min<uint32_t>(h_nc[f], s->max_comp);
            rec_at[f] = s->chain_mode ? (size_t)f * s->max_comp : at;   // k_frame_chain: every frame has its own record area
            at += nc;
            if (const OverflowFrame* o = overflow_frame(f)) {
                nb += o->boxes.size();
                nr += want_refl ? o->refls.size() : 0;
            } else {
                nb += h_sm[(size_t)f * 8 + 0];
                nr += want_refl ? h_sm[(size_t)f * 8 + 2] : 0;
            }
        }
        box_at[n] = nb;
        refl_at[n] = nr;
    }
    out_boxes.resize(box_at[n]);
    out_refls.resize(refl_at[n]);
    // the centres as (frame id, x, y, z) rows, written while each record is in hand: ffs_stream_spot_centres hands them out with one
    // memcpy (walking the 72-byte reflections again cost a caller 0.3 ms per batch of 45 000 -- as long as the GPU takes for the batch)
    out_centres.resize(refl_at[n] * 4);
    ffs_box* const bo = out_boxes.data();
    ffs_reflection* const ro = out_refls.data();
    float* const co = out_centres.data();
    const WireRec2* const recs0 = reinterpret_cast<const WireRec2*>(s->h_recs);
    std::atomic<bool> consistent{true};
    const std::function<void(uint32_t)> assemble = [&](uint32_t f) {
        size_t nbx = box_at[f], nrf = refl_at[f];
        const uint32_t id_bits = (uint32_t)((uint64_t)(s->first_id + f) & 0xFFFFFFFFull);   // (a bit pattern: as a float VALUE ids collide from 2^24 on)
        float id_lane;
        std::memcpy(&id_lane, &id_bits, 4);
        if (const OverflowFrame* o = overflow_frame(f)) {  // re-run on the one-frame stream: its cut records are skipped
            for (const ffs_box& b : o->boxes) bo[nbx++] = b;
            if (want_refl)
                for (const ffs_reflection& r : o->refls) {
                    co[4 * nrf] = id_lane; co[4 * nrf + 1] = r.com_x; co[4 * nrf + 2] = r.com_y; co[4 * nrf + 3] = r.com_z;
                    ro[nrf++] = r;
                }
            return;
        }
        const uint32_t nc = std::min<uint32_t>(h_nc[f], s->max_comp);
        const WireRec2* wrec = recs0 + rec_at[f];
        const size_t box_end = box_at[f + 1], refl_end = refl_at[f + 1];
        for (uint32_t q = 0; q < nc; ++q, ++wrec) {
            const uint32_t npx = wrec->npx_flags & 0x3FFFFFFFu, flags = wrec->npx_flags >> 30;
            if (min_size == 0 || npx >= min_size) {
                if (nbx >= box_end) { consistent = false; return; }   // (never: the summary counts what this loop counts)
                bo[nbx++] = ffs_box{wrec->x_min, wrec->y_min, wrec->x_max, wrec->y_max, (int32_t)npx};
            }
            if (want_refl && flags == 0) {
                if (nrf >= refl_end) { consistent = false; return; }
                ffs_reflection r{};
                r.x_min = wrec->x_min; r.x_max = wrec->x_max; r.y_min = wrec->y_min; r.y_max = wrec->y_max;
                r.z_min = 0; r.z_max = 0;
                r.num_pixels = (int32_t)npx;
                r.com_x = wrec->com_x; r.com_y = wrec->com_y; r.com_z = 0.5f;  // z = 0 for a single frame
                r.peak_x = wrec->peak_x; r.peak_y = wrec->peak_y; r.peak_z = 0;
                r.peak_intensity = wrec->peak_intensity;
                r.peak_centroid_distance = wrec->peak_centroid_distance;
                r.flags = 0;
                r.sum_intensity = wrec->sum_intensity;
                co[4 * nrf] = id_lane; co[4 * nrf + 1] = r.com_x; co[4 * nrf + 2] = r.com_y; co[4 * nrf + 3] = 0.5f;
                ro[nrf++] = r;
            }
        }
        if (nbx != box_end || nrf != refl_end) consistent = false;
    };
    // assemble: boxes = components surviving the min-size filter (connected_components.cc:122-135),
    // reflections = components surviving filter_reflections (:207-236); both keep label order.
    bool pooled = false;
    if (total_recs >= 8192 && n >= 4) {
        AssemblyPool* pool = c->assembly.load(std::memory_order_acquire);
        if (!pool && !process_exiting()) {
            std::lock_guard<std::mutex> lock(c->stream_mu);
            pool = c->assembly.load(std::memory_order_acquire);
            if (!pool) {
                pool = new (std::nothrow) AssemblyPool();
                if (pool) {
                    try {
                        pool->start(std::max(1, std::min(31, c->tune.assembly_threads)));
                    } catch (...) {   // (no more threads to be had: whatever did start is joined, and this wait assembles on its own)
                        delete pool;
                        pool = nullptr;
                    }
                }
                if (pool) c->assembly.store(pool, std::memory_order_release);   // (published with its members built and its threads started)
            }
        }
        if (pool && pool->owner.try_lock()) {   // (another stream's wait has the helpers: assemble here)
            pool->run(n, assemble);
            pool->owner.unlock();
            pooled = true;
        }
    }
    if (!pooled)
        for (uint32_t f = 0; f < n; ++f) assemble(f);
    if (!consistent.load()) return FFS_ERR_DEVICE;   // (the caller words the error: this may run on the context's own thread)
    for (uint32_t f = 0; f < n; ++f) {
        ffs_frame_result& r = out_results[f];
        const uint32_t* sm = h_sm + (size_t)f * 8;
        r.frame_id = s->first_id + f;
        r.num_strong_pixels = h_ns[f];
        r.num_strong_pixels_filtered = sm[1];
        r.n_components = h_nc[f];
        r.n_boxes = sm[0];
        r.boxes = out_boxes.data() + box_at[f];
        r.n_reflections = p.want_reflections ? sm[2] : 0;
        r.reflections = p.want_reflections ? out_refls.data() + refl_at[f] : nullptr;
        r.n_filtered_size = sm[3];
        r.n_filtered_sep = sm[4];
        if (p.want_strong_list) {
            r.strong_k = s->h_list_k ? s->h_list_k + (size_t)f * s->cap : nullptr;
            r.strong_intensity = s->h_list_i ? s->h_list_i + (size_t)f * s->cap : nullptr;
        }
        if (p.want_strong_mask) r.strong_mask = s->h_mask + (size_t)f * L.W * L.H;
        if (const OverflowFrame* o = overflow_frame(f)) {
            const ffs_frame_result& b = o->res;
            r.num_strong_pixels = b.num_strong_pixels;
            r.num_strong_pixels_filtered = b.num_strong_pixels_filtered;
            r.n_components = b.n_components;
            r.n_boxes = b.n_boxes;
            r.n_reflections = p.want_reflections ? b.n_reflections : 0;
            r.n_filtered_size = b.n_filtered_size;
            r.n_filtered_sep = b.n_filtered_sep;
            if (p.want_strong_list) {
                r.strong_k = o->k.data();
                r.strong_intensity = o->inten.data();
            }
        }
    }
    return FFS_OK;
}

// ---- assembly ahead of ffs_wait (ffs_internal.hpp: AheadThread) ------------------------------------------------------------------
static void ahead_main(ffs_ctx* c) {
    AheadThread* A = c->ahead.load(std::memory_order_acquire);
    (void)hipSetDevice(c->device);
    for (;;) {
        ffs_stream* s = nullptr;
        {
            std::unique_lock<std::mutex> lock(A->mu);
            A->cv_work.wait(lock, [&] { return A->stop || !A->q.empty(); });
            if (A->q.empty()) return;   // (stop: what is still queued is left to its ffs_wait, below)
            s = A->q.front();
            A->q.pop_front();
        }
        int verdict = 3;
        try {
            const ffs_params& p = s->batch_params;
            if (hipEventSynchronize(s->ev[4]) == hipSuccess && !p.want_strong_list && !p.want_strong_mask && s->direct_recs) {
                const uint32_t n = s->n_frames;
                const size_t B = s->max_batch;
                uint32_t overflow = s->h_counts[10 * B];
                if (s->chain_mode) {
                    overflow = 0;
                    for (uint32_t f = 0; f < n; ++f) overflow |= s->h_counts[10 * B + 1 + f];
                }
                if (overflow == 0) {
                    uint64_t total_recs = 0;
                    for (uint32_t f = 0; f < n; ++f) total_recs += std::min<uint32_t>(s->h_counts[B + f], s->max_comp);
                    if (assemble_batch(s, total_recs, nullptr, s->results_n, s->boxes_n, s->refls_n, s->centres_n) == FFS_OK) verdict = 2;
                }
            } else {
                (void)hipGetLastError();
            }
        } catch (...) {
            verdict = 3;   // (out of memory while the arrays grew: the caller's ffs_wait tries again and reports it)
        }
        {
            std::lock_guard<std::mutex> lock(A->mu);
            s->ahead_state = verdict;
        }
        A->cv_done.notify_all();
    }
}

void ahead_register(ffs_stream* s) {
    ffs_ctx* c = s->ctx;
    if (!c->tune.wait_ahead || process_exiting()) return;
#ifdef FFS_EXPERIMENTS
    if (s->h_phase_ts) return;   // (the phase times of the sparse launch are summed up by ffs_wait's own path)
#endif
    AheadThread* A = c->ahead.load(std::memory_order_acquire);
    if (!A) {
        std::lock_guard<std::mutex> lock(c->stream_mu);
        A = c->ahead.load(std::memory_order_acquire);
        if (!A) {
            A = new (std::nothrow) AheadThread();
            if (!A) return;
            c->ahead.store(A, std::memory_order_release);
            try {
                A->th = std::thread(ahead_main, c);
            } catch (...) {   // (no thread to be had: ffs_wait assembles, as before)
                c->ahead.store(nullptr, std::memory_order_release);
                delete A;
                return;
            }
        }
    }
    {
        std::lock_guard<std::mutex> lock(A->mu);
        if (A->stop) return;
        s->ahead_state = 1;
        A->q.push_back(s);
    }
    A->cv_work.notify_one();
}

int ahead_take(ffs_stream* s) {
    AheadThread* A = s->ctx->ahead.load(std::memory_order_acquire);
    if (!A) return 0;
    std::unique_lock<std::mutex> lock(A->mu);
    if (s->ahead_state == 1 && A->stop) {   // the thread has left (process exit): take the batch back if it is still queued
        for (auto it = A->q.begin(); it != A->q.end(); ++it)
            if (*it == s) { A->q.erase(it); s->ahead_state = 3; break; }
    }
    A->cv_done.wait(lock, [&] { return s->ahead_state != 1; });
    const int v = s->ahead_state;
    s->ahead_state = 0;
    return v;
}

void ahead_stop(ffs_ctx* c, bool destroy) {
    AheadThread* A = c->ahead.load(std::memory_order_acquire);
    if (!A) return;
    {
        std::lock_guard<std::mutex> lock(A->mu);
        A->stop = true;
        for (ffs_stream* s : A->q) s->ahead_state = 3;   // (still queued: their ffs_wait does the work)
        A->q.clear();
    }
    A->cv_work.notify_all();
    A->cv_done.notify_all();
    if (A->th.joinable()) A->th.join();
    if (destroy) {
        c->ahead.store(nullptr, std::memory_order_release);
        delete A;
    }
}

int ffs_wait_impl(ffs_stream* s, const ffs_frame_result** results, uint32_t* n_results) {
    if (!s) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    if (!s->busy) {
        c->err = "ffs_wait: nothing submitted";
        return FFS_ERR_INVALID;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (s->job.joinable()) {
        s->job.join();
        if (s->job_rc != FFS_OK) {
            (void)hipStreamSynchronize(s->st_up);
            mark_idle(s);
            c->err = s->job_err;
            return s->job_rc;
        }
    }
    if (ahead_take(s) == 2) {
        // the context's own thread has assembled this batch behind its last event: its arrays become the stream's
        s->results.swap(s->results_n);
        s->boxes.swap(s->boxes_n);
        s->refls.swap(s->refls_n);
        s->centres.swap(s->centres_n);
        // the arrays handed back now take the NEXT batch: sized (and their pages touched) here, once, rather than by the assembly
        // thread in the middle of a run (a stream's second batch paid 0.5 ms of first-touch page faults for its 7 MB)
        if (s->boxes_n.size() < s->boxes.size()) s->boxes_n.resize(s->boxes.size());
        if (s->refls_n.size() < s->refls.size()) s->refls_n.resize(s->refls.size());
        if (s->centres_n.size() < s->centres.size()) s->centres_n.resize(s->centres.size());
        s->ovf.clear();
        mark_idle(s);
        s->timing_last = s->ev[4];
        s->timings_stale = true;
        if (results) *results = s->results.data();
        if (n_results) *n_results = s->n_frames;
        return FFS_OK;
    }
    HIP_TRY(c, hipEventSynchronize(s->ev[4]));
    const uint32_t n = s->n_frames;
    const size_t B = s->max_batch;
    const Layout& L = c->L;
    const ffs_params& p = s->batch_params;
    const uint32_t* h_ns = s->h_counts;
    const uint32_t* h_nc = s->h_counts + B;
    uint32_t overflow = s->h_counts[10 * B];
    if (s->chain_mode) {  // k_frame_chain: one flag word per frame
        overflow = 0;
        for (uint32_t f = 0; f < n; ++f) overflow |= s->h_counts[10 * B + 1 + f];
    }
    mark_idle(s);
#ifdef FFS_EXPERIMENTS
    if (s->h_phase_ts && s->chain_mode) {   // phase durations of this batch's sparse launch, averaged over its frames (100 MHz counter)
        for (uint32_t f = 0; f < n; ++f) {
            const unsigned long long* t = s->h_phase_ts + (size_t)f * 8;
            if (t[6] <= t[0]) continue;
            for (int k = 0; k < 6; ++k) s->phase_sum[k] += (double)(t[k + 1] - t[k]) * 0.01;
            s->phase_sum[6] += (double)(t[6] - t[0]) * 0.01;
            ++s->phase_n;
        }
    }
#endif
    s->ovf.clear();
    if (overflow) {
        s->bits_dirty = true;
        (void)hipMemsetAsync(s->d_overflow, 0, 4, s->st2);
        (void)hipStreamSynchronize(s->st2);
        if (overflow & 4u) {
            c->err = "corrupt bitshuffle-LZ4 chunk: an LZ4 block did not decode to its block size";
            return FFS_ERR_INVALID;
        }
        if (overflow & 8u) {
            // more bright-window pixels than the list k_stream_u16 hands to k_bright_fix holds (a batch of
            // saturated frames): run the batch again with those pixels marked in the plane as candidates for the exact kernel
            // (threshold path 1; the extended algorithm: its plain first pass), which has no such list
            s->force_path = 1;
            int rc = enqueue_batch(s, s->cur_img, s->cur_pitch, s->cur_fstride, s->n_frames, &s->batch_params);
            s->force_path = -1;
            if (rc != FFS_OK) return rc;
            ++s->reruns;
            return ffs_wait_impl(s, results, n_results);
        }
        if ((overflow & 128u) && !(overflow & 32u)) {
            // a band of a frame beyond the plan of the small-workgroup sparse stage (kernels_band.hpp: strong pixels, log entries,
            // components or seam pixels of ONE band): the batch again through the one-workgroup launch, and so the stream's next batches
            // (data that is dense stays dense)
            s->bands_once_off = true;
            s->band_backoff = 32;
            int rc = enqueue_batch(s, s->cur_img, s->cur_pitch, s->cur_fstride, s->n_frames, &s->batch_params);
            s->bands_once_off = false;
            if (rc != FFS_OK) return rc;
            ++s->reruns;
            return ffs_wait_impl(s, results, n_results);
        }
        if (overflow & (32u | 64u)) {
            // the wave logs could not serve a frame of the batch: the batch again through the plane.  A wave with more strong groups
            // than its log holds (32): the stream stays with the plane.  A frame with more strong pixels than the one launch's LDS
            // forest (64): only this batch -- the next one follows what this one held (dense data takes the plane by itself,
            // and the logs are back when the data is sparse again)
            if (overflow & 32u) s->log_off = true;
            s->plane_once = true;
            int rc = enqueue_batch(s, s->cur_img, s->cur_pitch, s->cur_fstride, s->n_frames, &s->batch_params);
            s->plane_once = false;
            if (rc != FFS_OK) return rc;
            ++s->reruns;
            return ffs_wait_impl(s, results, n_results);
        }
        if (overflow & 16u) {
            // a dense frame with more runs than the one-launch sparse stage holds in LDS (kernels_chain.hpp): the batch again, its
            // sparse stage as the four grid-wide kernels; the stream's later dense batches go there directly
            s->runs_overflowed = true;
            s->force_grid = true;
            int rc = enqueue_batch(s, s->cur_img, s->cur_pitch, s->cur_fstride, s->n_frames, &s->batch_params);
            s->force_grid = false;
            if (rc != FFS_OK) return rc;
            ++s->reruns;
            return ffs_wait_impl(s, results, n_results);
        }
        // A frame with more strong pixels than the stream's lists hold (flag 1) or more components than its
        // record buffers (flag 2) -- an ice ring, the direct beam.  The reference has no such limit (std::map of
        // signals, connected_components.cc:24-32), so neither may the drop-in: the other frames of the batch are
        // complete (lists and records are per frame), and each frame that did not fit is run again on its own on
        // a one-frame stream with room for it (kept for the next time).
        for (uint32_t f = 0; f < n; ++f) {
            if (h_ns[f] <= s->cap && h_nc[f] <= s->max_comp) continue;
            uint64_t need_cap = std::max<uint64_t>(h_ns[f], s->cap);
            uint64_t need_comp = h_ns[f] > s->cap ? need_cap : std::max<uint64_t>(h_nc[f], s->max_comp);  // list truncated: count unknown
            for (int attempt = 0;; ++attempt) {
                if (s->big && (s->big->cap < need_cap || s->big->max_comp < need_comp)) {
                    stream_destroy_internal(s->big);
                    s->big = nullptr;
                }
                if (!s->big) {
                    const uint64_t npx = (uint64_t)L.W * L.H;
                    const uint32_t bc = (uint32_t)std::min<uint64_t>(npx, need_cap + need_cap / 4 + 1024);
                    const uint32_t bm = (uint32_t)std::min<uint64_t>(bc, need_comp + need_comp / 4 + 1024);
                    int rc = stream_create_sized(c, 1, bc, bm, &s->big);
                    if (rc != FFS_OK) return rc;  // a real out-of-memory
                }
                ffs_stream* b = s->big;
                const uint8_t* img = static_cast<const uint8_t*>(s->cur_img) + (size_t)f * s->cur_fstride;
                b->first_id = s->first_id + f;
                b->dev_input = true;
                b->ev1_pending = true;
                // (the frame's strong-pixel list comes back to the host whenever somebody may read it: the caller, or a 3D stack
                // that is alive -- ffs_stack3d_add_batch takes an overflow frame's list from here, tuning "device_lists")
                ffs_params bp = s->batch_params;
                if (c->tune.device_lists == 1 || (c->tune.device_lists == 2 && g_live_stacks.load() > 0)) bp.want_strong_list = 1;
                int rc = enqueue_batch(b, img, s->cur_pitch, s->cur_fstride, 1, &bp);
                if (rc != FFS_OK) return rc;
                HIP_TRY(c, hipEventSynchronize(b->ev[4]));
                const uint32_t b_ovf = b->chain_mode ? b->h_counts[10 * (size_t)b->max_batch + 1] : b->h_counts[10 * (size_t)b->max_batch];
                if (b_ovf & 3u) {  // only the component count can still be short (it was a guess while the list was cut)
                    (void)hipMemsetAsync(b->d_overflow, 0, 4, b->st2);
                    (void)hipStreamSynchronize(b->st2);
                    mark_idle(b);
                    b->bits_dirty = true;
                    need_cap = std::max<uint64_t>(need_cap, b->h_counts[0]);
                    need_comp = std::max<uint64_t>(need_comp * 2, b->h_counts[b->max_batch]);
                    if (attempt >= 4) {
                        c->err = "a frame still overflows its one-frame stream";
                        return FFS_ERR_OVERFLOW;
                    }
                    continue;
                }
                const ffs_frame_result* br = nullptr;
                uint32_t bn = 0;
                rc = ffs_wait_impl(b, &br, &bn);
                if (rc != FFS_OK) return rc;
                s->ovf.emplace_back();
                OverflowFrame& o = s->ovf.back();
                o.frame = f;
                o.res = br[0];
                o.boxes.assign(br[0].boxes, br[0].boxes + br[0].n_boxes);
                if (br[0].reflections) o.refls.assign(br[0].reflections, br[0].reflections + br[0].n_reflections);
                if (br[0].strong_k) {
                    o.k.assign(br[0].strong_k, br[0].strong_k + br[0].num_strong_pixels);
                    o.inten.assign(br[0].strong_intensity, br[0].strong_intensity + br[0].num_strong_pixels);
                }
                break;
            }
        }
    }
    uint64_t total_recs = 0;
    uint32_t max_ns = 0;
    for (uint32_t f = 0; f < n; ++f) {
        total_recs += std::min<uint32_t>(h_nc[f], s->max_comp);  // (the kernels never write more than max_comp per frame)
        max_ns = std::max(max_ns, std::min<uint32_t>(h_ns[f], s->cap));
    }
    bool second_phase = false;
    if (total_recs > s->spec_recs_copied) {  // more records than the speculative copy brought: fetch the rest
        const size_t rb = sizeof(WireRec2);
        HIP_TRY(c, hipMemcpyAsync(reinterpret_cast<uint8_t*>(s->h_recs) + s->spec_recs_copied * rb,
                                  reinterpret_cast<const uint8_t*>(s->d_recs) + s->spec_recs_copied * rb,
                                  (total_recs - s->spec_recs_copied) * rb, hipMemcpyDeviceToHost, s->st2));
        second_phase = true;
    }
    s->spec_recs_per_frame = std::max<uint32_t>(s->spec_recs_per_frame,
                                                (uint32_t)std::min<uint64_t>(s->max_comp, (total_recs / n + 1) * 5 / 4));
    if (p.want_strong_list && max_ns) {
        second_phase = true;
        int rc = ensure_list_host(s);
        if (rc != FFS_OK) return rc;
        HIP_TRY(c, hipMemcpy2DAsync(s->h_list_k, (size_t)s->cap * 4, s->d_list_k, (size_t)s->cap * 4,
                                    (size_t)max_ns * 4, n, hipMemcpyDeviceToHost, s->st2));
        HIP_TRY(c, hipMemcpy2DAsync(s->h_list_i, (size_t)s->cap * 4, s->d_list_i, (size_t)s->cap * 4,
                                    (size_t)max_ns * 4, n, hipMemcpyDeviceToHost, s->st2));
    }
    if (p.want_strong_mask) {
        second_phase = true;
        if (!s->h_mask)
            HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&s->h_mask), B * (size_t)L.W * L.H, hipHostMallocDefault));
        // the reference's full-mask D2H (spotfinder.cc:887-894), all frames of the batch in one 2D copy
        HIP_TRY(c, hipMemcpy2DAsync(s->h_mask, L.W, s->d_sbytes, L.bpitch, L.W, (size_t)L.H * n,
                                    hipMemcpyDeviceToHost, s->st2));
    }
    hipEvent_t last = s->ev[4];
    if (second_phase) {
        HIP_TRY(c, hipEventRecord(s->ev[5], s->st2));
        HIP_TRY(c, hipEventSynchronize(s->ev[5]));
        last = s->ev[5];
    }
    // (the five stage times are read out of the events by ffs_stream_timings() when somebody asks: five runtime calls per
    // batch that the last wait of a run pays on the clock)
    s->timing_last = last;
    s->timings_stale = true;

    {
        const int rc = assemble_batch(s, total_recs, &s->ovf, s->results, s->boxes, s->refls, s->centres);
        if (rc != FFS_OK) {
            c->err = "ffs_wait: the records of a frame do not match its summary counts";
            return rc;
        }
    }
    if (results) *results = s->results.data();
    if (n_results) *n_results = n;
    return FFS_OK;
}

extern "C" int ffs_stream_batch_arrays(ffs_stream* s, const ffs_box** boxes, uint32_t* n_boxes,
                                       const ffs_reflection** refls, uint32_t* n_refls) {
    if (!s) return FFS_ERR_INVALID;
    if (boxes) *boxes = s->boxes.data();
    if (n_boxes) *n_boxes = (uint32_t)s->boxes.size();
    if (refls) *refls = s->refls.data();
    if (n_refls) *n_refls = (uint32_t)s->refls.size();
    return FFS_OK;
}

extern "C" int ffs_stream_last_path(ffs_stream* s, uint32_t* path_bits, uint32_t* reruns) {
    if (!s) return FFS_ERR_INVALID;
    if (path_bits) *path_bits = s->path_bits;
    if (reruns) *reruns = s->reruns;
    return FFS_OK;
}

extern "C" int ffs_stream_timings(ffs_stream* s, float ms[5]) {
    if (!s || !ms) return FFS_ERR_INVALID;
    if (s->timings_stale && !s->busy) {   // (the events of the last batch waited for; a new submit re-records them)
        (void)hipSetDevice(s->ctx->device);
        s->timings[0] = 0.0f;
        if (!s->dev_input) (void)hipEventElapsedTime(&s->timings[0], s->ev[0], s->ev[1]);
        (void)hipEventElapsedTime(&s->timings[1], s->ev[1], s->ev[2]);
        (void)hipEventElapsedTime(&s->timings[2], s->ev[2], s->ev3_is_ev4 ? s->ev[4] : s->ev[3]);
        (void)hipEventElapsedTime(&s->timings[3], s->ev3_is_ev4 ? s->ev[4] : s->ev[3], s->timing_last);
        (void)hipEventElapsedTime(&s->timings[4], s->dev_input ? s->ev[1] : s->ev[0], s->timing_last);
        (void)hipGetLastError();
        s->timings_stale = false;
    }
    std::memcpy(ms, s->timings, sizeof(s->timings));
    return FFS_OK;
}

extern "C" int ffs_stream_spot_centres(ffs_stream* s, float* rows4, uint32_t cap, uint32_t* n_written) {
    if (!s || !rows4) return FFS_ERR_INVALID;
    // (reads the result arrays of the last ffs_wait only: they stay as they are while the NEXT batch is in flight, so a caller may
    // pack one batch's centres on a helper thread while it submits the next -- until the next ffs_wait on this stream)
    const uint64_t wanted = s->centres.size() / 4;
    const uint32_t n = (uint32_t)std::min<uint64_t>(wanted, cap);
    if (n) std::memcpy(rows4, s->centres.data(), (size_t)n * 16);
    // last row: (rows written, rows wanted) as uint32 bit patterns -- wanted > written tells the receiver
    // that `cap` was too small (nothing is dropped silently)
    float* last = rows4 + (size_t)cap * 4;
    const uint32_t tail[4] = {n, (uint32_t)std::min<uint64_t>(wanted, 0xFFFFFFFFull), 0u, 0u};
    std::memcpy(last, tail, sizeof(tail));
    if (n_written) *n_written = n;
    return wanted > n ? FFS_ERR_OVERFLOW : FFS_OK;
}

extern "C" int ffs_stream_debug_planes(ffs_stream* s, const uint8_t** strong_bytes, size_t* mask_pitch,
                                       size_t* mask_fstride) {
    if (!s) return FFS_ERR_INVALID;
    if (strong_bytes) *strong_bytes = s->d_sbytes;
    if (mask_pitch) *mask_pitch = s->ctx->L.bpitch;
    if (mask_fstride) *mask_fstride = s->ctx->L.bytes_frame_stride;
    return FFS_OK;
}

extern "C" int ffs_stream_debug_bitplane(ffs_stream* s, uint32_t frame, int which, uint8_t* host_out) {
    if (!s || !host_out || which < 0 || which > 2) return FFS_ERR_INVALID;
    ffs_ctx* c = s->ctx;
    const Layout& L = c->L;
    if (s->busy || frame >= s->max_batch) {
        c->err = "ffs_stream_debug_bitplane: stream busy or frame out of range";
        return FFS_ERR_INVALID;
    }
    const uint8_t* src = which == 0 ? s->d_bits : which == 1 ? s->d_dplane : s->d_eplane;
    if (!src) {
        c->err = "ffs_stream_debug_bitplane: that plane exists only after an extended-dispersion batch";
        return FFS_ERR_INVALID;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (which == 0 && s->bits_cleared && !s->dense_valid && !s->lists_valid) {
        c->err = "ffs_stream_debug_bitplane: the batch kept neither the byte mask nor the strong-pixel list (want_strong_mask / want_strong_list)";
        return FFS_ERR_INVALID;
    }
    if (which == 0 && s->bits_cleared && !s->dense_valid) {
        // the compaction consumed (and cleared) the plane and nobody asked for the byte mask: the strong-pixel list has them
        const uint32_t ns = std::min<uint32_t>(s->h_counts[frame], s->cap);
        std::vector<uint32_t> ks(ns);
        if (ns) HIP_TRY(c, hipMemcpy(ks.data(), s->d_list_k + (size_t)frame * s->cap, (size_t)ns * 4, hipMemcpyDeviceToHost));
        std::memset(host_out, 0, (size_t)L.W * L.H);
        for (uint32_t k : ks) host_out[k] = 1;
        return FFS_OK;
    }
    if (which == 0 && s->bits_cleared) {
        // the compaction consumed (and cleared) the plane; the byte mask holds the same pixels
        HIP_TRY(c, hipMemcpy2D(host_out, L.W, s->d_sbytes + (size_t)frame * L.bytes_frame_stride, L.bpitch, L.W, L.H,
                               hipMemcpyDeviceToHost));
        return FFS_OK;
    }
    std::vector<uint8_t> packed(L.plane_frame_stride);
    HIP_TRY(c, hipMemcpy(packed.data(), src + (size_t)frame * L.plane_frame_stride, packed.size(), hipMemcpyDeviceToHost));
    for (int y = 0; y < L.H; ++y)
        for (int x = 0; x < L.W; ++x)
            host_out[(size_t)y * L.W + x] = (packed[(size_t)y * L.mpitch + (x >> 3)] >> (x & 7)) & 1u;
    return FFS_OK;
}

extern "C" int ffs_wait(ffs_stream* s, const ffs_frame_result** results, uint32_t* n_results) {
    if (!s || !stream_handle_ok(s)) return FFS_ERR_INVALID;
    return guarded(s->ctx, [&] { return ffs_wait_impl(s, results, n_results); });
}

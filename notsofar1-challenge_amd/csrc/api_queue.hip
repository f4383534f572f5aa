// libcss_mi355.so, host side 3 / 3: the fused pass (css_run / css_run_device / css_run_pcm16: css/css.py:110 as a pipeline of units), the queue
// of sessions (css_run_enqueue* / css_wait / css_wait_sessions) and the estimator batches queued sessions share.  (api_ctx.hpp: shared.)
#include "api_ctx.hpp"

// ---- the fused pass --------------------------------------------------------------------------------------------------
// Where the samples of a pass come from and where its result goes (exactly one source, exactly one sink).
struct RunIo {
    const float* pcm_host = nullptr;             // [n][C] float32 in host memory   (css_run)
    const float* pcm_dev = nullptr;              // [n][C] float32 in HBM           (css_run_device)
    const int16_t* const* planes_host = nullptr; // C mono PCM16 planes in host memory (css_run_pcm16)
    float* wav_host = nullptr;                   // [S][cap] float32
    float* wav_dev = nullptr;
    int16_t* wav16_host = nullptr;               // [S][cap] peak-normalised PCM16
    float* peaks_host = nullptr;
    int64_t cap = 0;
    bool enqueue_only = false;                   // css_run_enqueue: return once everything is on the streams
};

// device address of page-locked (hipHostMalloc / css_host_alloc / registered) host memory, nullptr for pageable memory
void* mapped_host(const void* p) {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return at.type == hipMemoryTypeHost ? at.devicePointer : nullptr;
}

// the completion event of the session whose last output copy was just enqueued on `st` (css_wait_sessions)
void mark_session_done(css_ctx* h, hipStream_t st) {
    if (h->sess_ev_used == h->sess_ev_pool.size()) {
        hipEvent_t e = nullptr;
        hipEventCreateWithFlags(&e, hipEventDisableTiming);
        h->sess_ev_pool.push_back(e);
    }
    hipEvent_t e = h->sess_ev_pool[h->sess_ev_used++];
    hipEventRecord(e, st);
    h->sess_done.push_back(e);
}

hipEvent_t pool_event(css_ctx* h) {
    if (h->ev_pool_used == h->ev_pool.size()) {
        hipEvent_t e = nullptr;
        hipEventCreateWithFlags(&e, hipEventDisableTiming);
        h->ev_pool.push_back(e);
    }
    return h->ev_pool[h->ev_pool_used++];
}

// the per-launch event brackets recorded since the session began (css_set_profile) -> per-family sums; the streams
// they were recorded on must have been synchronised
void reduce_profile(css_ctx* h) {
    CssTimings& t = h->tim;
    t.gemm_ms = 0.f; t.gemm_launches = 0; t.gemm_flops = h->gemm_flops;
    for (int c = 0; c < CSS_PROF_COUNT; ++c) { h->prof_ms[c] = 0.f; h->prof_launches[c] = 0; }
    if (h->profile_gemm) {
        for (size_t i = 0; i < h->prof_used; ++i) {
            float v = 0.f;
            hipEventElapsedTime(&v, h->prof_events[i].a, h->prof_events[i].b);
            h->prof_ms[h->prof_events[i].cat] += v;
            h->prof_launches[h->prof_events[i].cat] += 1;
        }
        t.gemm_ms = h->prof_ms[CSS_PROF_LINEAR];
        t.gemm_launches = h->prof_launches[CSS_PROF_LINEAR];
    }
    h->prof_reduced = h->prof_used;
}

// stage times of the pass just synchronised (HIP events on the handle's streams) and the per-family kernel profile
using HostClock = std::chrono::steady_clock::time_point;
int finish_timings(css_ctx* h, HostClock t0, HostClock t1, HostClock t2, bool staged) {
    auto ms = [&](int a, int b) { float v = 0.f; hipEventElapsedTime(&v, h->ev[a], h->ev[b]); return v; };
    CssTimings& t = h->tim;
    t.host_enqueue = std::chrono::duration<float, std::milli>(t1 - t0).count();
    t.host_total = std::chrono::duration<float, std::milli>(t2 - t0).count();
    // (pipelined pass: the stages overlap -- masknet = first chain's begin .. last chain's end, beamformer included;
    //  stitch / istft = the LAST batch's tail)
    t.upload = ms(0, 1); t.stft = ms(1, 2); t.masknet = ms(2, 3); t.mvdr = staged ? ms(3, 4) : 0.f; t.stitch = ms(4, 5);
    t.istft = ms(5, 6); t.download = ms(6, 7); t.total = ms(0, 7); t.features = 0.f;
    reduce_profile(h);
    return CSS_OK;
}

// One pass of css/css.py:110 separate_and_stitch as a pipeline.  The recording's segments go through the mask estimator
// in batches, each cut into lanes (css_ctx::lanes); a (batch, lane) UNIT owns the frames no earlier unit reads.
//   in    its samples cross PCIe on the copy stream as one piece; the lane's chain waits for that piece only, transforms
//         the unit's frames, and starts the estimator on its segments while the later pieces are still in flight;
//   lane  features -> Conformer -> masks, then covariances, MVDR solve and beamformer of the same segments;
//   tail  batch by batch on the tail stream: stitching costs of the batch's boundaries, the permutation scan CONTINUED over
//         them (css.py:266-285 is sequential, but only forwards), overlap-add of the frames no later segment covers,
//         gate and synthesis of those frames less the dilate / erode halo, and their samples back over PCIe --
//         while the lanes work on the next batch.  Only the last batch's tail is not hidden.
constexpr int CSS_QUEUE_LEAD = 3;   // queued passes the host may be ahead of the device (css_run_enqueue blocks beyond)
int run_once(css_handle_t h, int64_t n, int32_t n_ch, const CssRunCfg* cfg, const RunIo& io) {
    int rc;
    if (!h) return CSS_ERR_INVALID_ARG;
    const auto host_t0 = std::chrono::steady_clock::now();
    // page-locked output (css_host_alloc): its device address, for the zero-copy output path
    float* wav_mapped = nullptr;
    if (io.wav_host && (h->tune[CSS_TUNE_OUT_MAPPED] || io.enqueue_only)) {
        if (h->mapped_key != io.wav_host) { h->mapped_key = io.wav_host; h->mapped_val = mapped_host(io.wav_host); }
        wav_mapped = (float*)h->mapped_val;
    }
    // queued passes OVERLAP when the output is page-locked (see css_ctx::pass_no); otherwise they just queue up
    // (with the beamformer on the tail stream -- CSS_TUNE_MVDR_ON_LANES = 0 -- a tail also reads the spectra X, which the
    // next pass's transform overwrites: such passes queue up without overlapping)
    const bool piped = h->fft512 && io.enqueue_only && io.pcm_host && wav_mapped && h->tune[CSS_TUNE_MVDR_ON_LANES];
    if (io.enqueue_only && h->queued && h->last_piped != (int)piped) {
        // the overlap mode changes inside a queue (a page-locked output follows a pageable one or the reverse): the two
        // modes order the level word, the mask buffers and the tail differently, so the queue is drained on the device
        // first (its bookkeeping -- css_wait, the range verdict -- stays with the caller)
        HIPCHK(h, hipSetDevice(h->device));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipStreamSynchronize(h->tail_stream));
        HIPCHK(h, hipStreamSynchronize(h->copy_stream));
        h->tail_pending = false;
    }
    if (io.enqueue_only) h->last_piped = (int)piped;
    const int par = piped ? (int)(h->pass_no & 1) : 0;
    h->peak_dev = (unsigned int*)h->level.p + 8 * par;
    h->piped_now = piped;
    // (overlapping passes alternate between the two sets of planes: a grouped pass in front of this one may still read
    // its own on the tail stream -- run_group -- while this pass's transform writes)
    if (piped) std::swap(h->X, h->X_alt);
    rc = begin_impl(h, n, n_ch, cfg);
    h->piped_now = false;
    if (rc != CSS_OK) return rc;
    const CssPlan& pl = h->plan;
    if (io.cap < pl.n_out) return fail(h, CSS_ERR_INVALID_ARG, "output buffer too small: need " + std::to_string(pl.n_out) + " samples per stream");
    const int64_t nseg = pl.num_segments, TL = pl.mix_frames;
    const int S = h->d.num_spks, F = h->d.num_bins, N = h->d.frame_len, fhop = h->d.frame_hop;
    const int T = h->cfg.segment_frames, hop = h->cfg.hop_frames;
    const bool from_host = io.pcm_host || io.planes_host;
    h->ev_pool_used = 0;
    if (io.pcm_host) {
        const size_t need = ((size_t)n * n_ch * sizeof(float) + 255) / 256 * 256;
        if ((rc = ensure(h, h->pcm_in, piped ? 2 * need : need)) != CSS_OK) return rc;
        // (queued passes alternate between the two halves of the allocation, whatever their lengths)
        h->pcm_src = (const float*)((const char*)h->pcm_in.p + (piped && par ? h->pcm_in.cap / 2 / 256 * 256 : 0));
    } else if (io.planes_host) {
        if ((rc = ensure(h, h->in16, (size_t)n * n_ch * sizeof(int16_t))) != CSS_OK) return rc;
        for (int c = 0; c < n_ch; ++c)
            if (!io.planes_host[c]) return fail(h, CSS_ERR_INVALID_ARG, "null channel plane");
    } else {
        h->pcm_src = io.pcm_dev;
    }
    if (io.wav16_host && (rc = ensure(h, h->enc, (size_t)S * pl.n_out * sizeof(int16_t) + 64)) != CSS_OK) return rc;
    const int16_t* planes_dev = io.planes_host ? (const int16_t*)h->in16.p : nullptr;
    hipEventRecord(h->ev[1], h->stream);
    hipEventRecord(h->ev[2], h->stream);   // the analysis transform is part of the lanes' chains (CssTimings.stft = 0)

    // ---- nothing to hide: with the samples already in HBM the plain stage sequence (whole transform, estimator with
    // its lanes, beamformer, costs, scan, overlap-add, gate, synthesis on one stream) measures 2 % ahead of the unit
    // pipeline below (profiles/r02_shard_overhead.md: 5.35 vs 5.43 ms per 60 s meeting, 143.0 vs 146.2 ms per 30 min)
    if (!h->fft512) {
        // Other frame sizes (ExtractorCfg.frame_len / frame_hop): the plain stage sequence on one stream, samples up and
        // waveforms down as whole copies -- the pipelined schedules below are built around frame_len = 2 hop
        if (io.pcm_host) {
            if ((rc = ensure(h, h->pcm_in, (size_t)n * n_ch * sizeof(float))) != CSS_OK) return rc;
            h->pcm_src = (const float*)h->pcm_in.p;
            HIPCHK(h, hipMemcpyAsync(h->pcm_in.p, io.pcm_host, (size_t)n * n_ch * sizeof(float), hipMemcpyHostToDevice, h->stream));
        }
        if (io.planes_host) {   // the first wav edge (round 6: any frame geometry): int16 planes up, scaled by 2^-15 on the way to channel-major
            for (int c = 0; c < n_ch; ++c) {
                HIPCHK(h, hipMemcpyAsync((int16_t*)h->in16.p + (size_t)c * n, io.planes_host[c], (size_t)n * sizeof(int16_t), hipMemcpyHostToDevice, h->stream));
                launch_pcm_peak_i16(planes_dev + (size_t)c * n, peak_len(h, 0, n), h->peak_dev, h->stream);
            }
            if (pl.stft_frames < TL)   // short input: zero-padded frames (css.py:159-164)
                HIPCHK(h, hipMemsetAsync(h->X.p, 0, (size_t)h->n_ch * X_ROWS_PER_BIN * F * h->T_ld * sizeof(float), h->stream));
            if ((rc = stft_frames(h, 0, TL, planes_dev, h->stream)) != CSS_OK) return rc;
            hipEventRecord(h->ev[2], h->stream);
            h->stft_done = true;
        } else {
            launch_pcm_peak_f32(h->pcm_src, peak_len(h, 0, n) * n_ch, h->peak_dev, h->stream);
            if ((rc = css_stage_stft(h)) != CSS_OK) return rc;
        }
        if ((rc = css_stage_masknet(h, 0, nseg)) != CSS_OK) return rc;
        if ((rc = css_stage_mvdr(h, 0, nseg)) != CSS_OK) return rc;
        if ((rc = css_stage_pit_costs(h, 0, nseg - 1)) != CSS_OK) return rc;
        if ((rc = css_stage_pit_scan(h)) != CSS_OK) return rc;
        if ((rc = css_stage_stitch(h, 0, TL)) != CSS_OK) return rc;
        float* dst = io.wav_dev;
        int64_t dst_ld = io.cap;
        if (!dst) {
            if ((rc = ensure(h, h->wav, (size_t)S * pl.n_out * sizeof(float))) != CSS_OK) return rc;
            dst = (float*)h->wav.p; dst_ld = pl.n_out;
        }
        if ((rc = istft_impl(h, 0, TL, 0, TL - 1 + h->ovl, dst, dst_ld, 0, h->stream)) != CSS_OK) return rc;
        if (io.wav16_host) {    // the second wav edge: peak normalisation + PCM16 encoding on the device (utils/audio_utils.py:37-49)
            const int64_t n_out = pl.n_out;
            unsigned int* pk = (unsigned int*)h->enc.p;
            int16_t* o16 = (int16_t*)((char*)h->enc.p + 64);
            { CSS_PROF(CSS_PROF_ENCODE, h->stream); launch_encode_pcm16(dst, S, n_out, pk, o16, n_out, h->stream); }
            HIPCHK(h, hipMemcpy2DAsync(io.wav16_host, (size_t)io.cap * sizeof(int16_t), o16, (size_t)n_out * sizeof(int16_t),
                                       (size_t)n_out * sizeof(int16_t), S, hipMemcpyDeviceToHost, h->stream));
            if (io.peaks_host) HIPCHK(h, hipMemcpyAsync(io.peaks_host, pk, (size_t)S * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        }
        if (io.wav_host)
            for (int sp = 0; sp < S; ++sp)
                HIPCHK(h, hipMemcpyAsync(io.wav_host + (size_t)sp * io.cap, dst + (size_t)sp * dst_ld, (size_t)pl.n_out * sizeof(float),
                                         hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->range_flag_host, h->range_flag_dev, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
        hipEventRecord(h->ev[7], h->stream);
        const auto host_t1 = std::chrono::steady_clock::now();
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipGetLastError());
        const auto host_t2 = std::chrono::steady_clock::now();
        return finish_timings(h, host_t0, host_t1, host_t2, true);
    }
    if (io.pcm_dev && io.wav_dev && !h->tune[CSS_TUNE_PIPELINE_DEVICE]) {
        launch_pcm_peak_f32(h->pcm_src, peak_len(h, 0, n) * n_ch, h->peak_dev, h->stream);
        if ((rc = css_stage_stft(h)) != CSS_OK) return rc;
        if ((rc = css_stage_masknet(h, 0, nseg)) != CSS_OK) return rc;
        if ((rc = css_stage_mvdr(h, 0, nseg)) != CSS_OK) return rc;
        if ((rc = css_stage_pit_costs(h, 0, nseg - 1)) != CSS_OK) return rc;
        if ((rc = css_stage_pit_scan(h)) != CSS_OK) return rc;
        if ((rc = css_stage_stitch(h, 0, TL)) != CSS_OK) return rc;
        if ((rc = istft_impl(h, 0, TL, 0, TL + 1, io.wav_dev, io.cap, 0, h->stream)) != CSS_OK) return rc;
        HIPCHK(h, hipMemcpyAsync(h->range_flag_host, h->range_flag_dev, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
        hipEventRecord(h->ev[7], h->stream);
        const auto host_t1 = std::chrono::steady_clock::now();
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipGetLastError());
        const auto host_t2 = std::chrono::steady_clock::now();
        return finish_timings(h, host_t0, host_t1, host_t2, true);
    }

    // ---- units, their frames and samples
    struct Unit { int64_t seg_lo; int n; int64_t f_lo, f_hi, s_lo, s_hi; hipEvent_t up, x, v, m; };   // pieces landed, planes, beamformer, costs
    std::vector<Unit> units;
    const int64_t cap = batch_len(nseg, std::min<int64_t>(batch_cap(h, h->cfg.segment_frames), nseg));
    int64_t f_prev = 0, s_prev = 0;
    for (int64_t s0 = 0; s0 < nseg; s0 += cap) {
        const int nb = (int)std::min<int64_t>(cap, nseg - s0);
        const LaneSplit ls = lane_split(h, nb, T);
        for (int l = 0; l < ls.nl; ++l) {
            const int lo = l * ls.per, cnt = std::min(ls.per, nb - lo);
            if (cnt <= 0) continue;
            Unit u{};
            u.seg_lo = s0 + lo; u.n = cnt;
            const bool last = u.seg_lo + cnt == nseg;
            u.f_lo = f_prev;
            u.f_hi = last ? TL : std::min<int64_t>((u.seg_lo + cnt - 1) * hop + T, TL);
            const int64_t fr = std::min<int64_t>(u.f_hi, pl.stft_frames);   // frames that exist
            u.s_lo = s_prev;
            u.s_hi = last ? n : std::max<int64_t>(s_prev, std::min<int64_t>(fr > 0 ? (fr - 1) * fhop + N : 0, n));
            f_prev = u.f_hi; s_prev = u.s_hi;
            u.up = from_host ? pool_event(h) : nullptr;
            u.x = pool_event(h);
            u.v = pool_event(h);
            u.m = pool_event(h);
            units.push_back(u);
        }
    }
    // ---- everything starts after whatever the previous pass left on the three streams
    if (!piped) {
        hipEvent_t start = pool_event(h);
        HIPCHK(h, hipEventRecord(start, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->copy_stream, start, 0));
        HIPCHK(h, hipStreamWaitEvent(h->tail_stream, start, 0));
    } else {
        // the samples go into the buffer the pass before last used: free once that pass has transformed its frames; the
        // level word of this parity is cleared here, in front of the pieces' peak scans (each stream is in order in itself)
        for (int b = 0; b < 2; ++b) {
            if (!h->pcm_free[b]) HIPCHK(h, hipEventCreateWithFlags(&h->pcm_free[b], hipEventDisableTiming));
            if (!h->level_free[b]) HIPCHK(h, hipEventCreateWithFlags(&h->level_free[b], hipEventDisableTiming));
        }
        if (!h->tail_end) HIPCHK(h, hipEventCreateWithFlags(&h->tail_end, hipEventDisableTiming));
        for (auto& e : h->pass_end)
            if (!e) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        // back-pressure: the host stays at most CSS_QUEUE_LEAD passes ahead of the device.  It enqueues a pass in 2 ms, the
        // device runs one in 5; an unbounded lead only makes the runtime grow its command and signal pools (measured:
        // 3.4 instead of 2.1 ms of enqueue time per pass while they grow, 5.7 instead of 5.4 ms per pass) and buys nothing.
        if (h->pass_no >= CSS_QUEUE_LEAD) HIPCHK(h, hipEventSynchronize(h->pass_end[(h->pass_no - CSS_QUEUE_LEAD) & 3]));
        if (h->pass_no >= 2) {
            HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->pcm_free[par], 0));
            // ... and the level word when that pass's TAIL has read it (the host may be several passes ahead of the device)
            HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->level_free[par], 0));
        }
        HIPCHK(h, hipMemsetAsync(h->peak_dev, 0, sizeof(unsigned int), h->copy_stream));
    }
    // ---- PCIe pieces, in unit order, on the copy stream
    if (from_host) {
        for (const Unit& u : units) {
            if (io.pcm_host) {
                if ((rc = upload_pcm(h, io.pcm_host, u.s_lo, u.s_hi, h->copy_stream)) != CSS_OK) return rc;
            } else if (u.s_hi > u.s_lo) {
                for (int c = 0; c < n_ch; ++c)
                    HIPCHK(h, hipMemcpyAsync((int16_t*)h->in16.p + (size_t)c * n + u.s_lo, io.planes_host[c] + u.s_lo,
                                             (size_t)(u.s_hi - u.s_lo) * sizeof(int16_t), hipMemcpyHostToDevice, h->copy_stream));
            }
            HIPCHK(h, hipEventRecord(u.up, h->copy_stream));
            // the recording's level (power-of-two gain of the split synthesis operand) piece by piece, beside the next upload
            if (io.pcm_host) launch_pcm_peak_f32(h->pcm_src + u.s_lo * n_ch, peak_len(h, u.s_lo, u.s_hi) * n_ch, h->peak_dev, h->copy_stream);
            else
                for (int c = 0; c < n_ch; ++c)
                    launch_pcm_peak_i16(planes_dev + (size_t)c * n + u.s_lo, peak_len(h, u.s_lo, u.s_hi), h->peak_dev, h->copy_stream);
        }
        hipEvent_t level = pool_event(h);
        HIPCHK(h, hipEventRecord(level, h->copy_stream));
        HIPCHK(h, hipStreamWaitEvent(h->tail_stream, level, 0));
    } else {
        launch_pcm_peak_f32(h->pcm_src, peak_len(h, 0, n) * n_ch, h->peak_dev, h->tail_stream);
    }
    if (pl.stft_frames < TL)   // short input: zero-padded frames (css.py:159-164)
        HIPCHK(h, hipMemsetAsync(h->X.p, 0, (size_t)h->n_ch * X_ROWS_PER_BIN * F * h->T_ld * sizeof(float), h->stream));

    // ---- the tail of unit `k` (all earlier tails are already enqueued on the tail stream)
    const int64_t halo = h->cfg.dilation_frames + h->cfg.erosion_frames;
    const StitchArgs sa = stitch_args(h);
    int64_t t_done = 0, g_done = 0;     // frames overlap-added / gated and synthesised so far
    hipEvent_t out_done = nullptr;
    // page-locked output: the overlap-add of the synthesis writes the samples straight into the caller's buffer over
    // PCIe (no device-side copy of the waveforms, no copy call: the runtime's device-to-host copies made the host wait
    // for the events they depend on); pageable output: into the device buffer, then a copy
    if (!piped && !h->tune[CSS_TUNE_OUT_MAPPED]) wav_mapped = nullptr;
    // (one tail per BATCH, not per unit: these kernels are latency-bound chains of small launches -- a third of the
    // frames takes the same ~120 us -- and the lanes of a batch finish together, so per-unit tails only queue up)
    auto tail_of = [&](size_t k0, size_t k1) -> int {   // units [k0, k1)
        hipStream_t ts = h->tail_stream;
        const bool last = k1 == units.size();
        for (size_t k = k0; k < k1; ++k) HIPCHK(h, hipStreamWaitEvent(ts, units[k].m, 0));
        struct { int64_t seg_lo; int64_t n; } u{units[k0].seg_lo, units[k1 - 1].seg_lo + units[k1 - 1].n - units[k0].seg_lo};
        const int64_t b_lo = std::max<int64_t>(u.seg_lo - 1, 0), b_hi = u.seg_lo + u.n - 1;
        if (!h->tune[CSS_TUNE_MVDR_ON_LANES]) {   // beamformer and costs here, after the lanes, instead of on them
            if (int e = mvdr_on(h, u.seg_lo, u.seg_lo + u.n, ts)) return e;
            pit_costs_on(h, b_lo, b_hi, ts);
        }
        // (the boundaries' costs were computed on the lanes, behind each unit's beamformer)
        pit_scan_on(h, b_lo, b_hi, ts);
        const int64_t t_hi = last ? TL : std::min<int64_t>((u.seg_lo + u.n) * hop, TL);   // no later segment covers these
        if (t_hi > t_done) { CSS_PROF(CSS_PROF_OLA_MASKS, ts); launch_ola_masks(sa, t_done, t_hi, ts); }
        t_done = std::max(t_done, t_hi);
        const int64_t g_end = last ? TL : std::max<int64_t>(t_done - halo, g_done);
        if (g_end > g_done || last) {
            { CSS_PROF(CSS_PROF_GATE, ts); launch_morphology(sa, g_done, g_end, ts); }
            // the last range may leave in CSS_TUNE_TAIL_PIECES pieces (default 1), the first piece's download beside the
            // second's synthesis: measured no gain -- these launches are latency-bound, a fifth of the frames costs what all cost
            const int pieces = (last && io.wav_host && g_end - g_done >= 512) ? std::max(h->tune[CSS_TUNE_TAIL_PIECES], 1) : 1;
            const int64_t g_first = g_done;
            for (int pc = 0; pc < pieces; ++pc) {
                // (two pieces: 3/5 + 2/5, the second download is the exposed one; more: equal parts)
                const int64_t g_hi = pc + 1 == pieces ? g_end
                                     : (pieces == 2 ? g_first + (g_end - g_first) * 3 / 5 : g_first + (g_end - g_first) * (pc + 1) / pieces);
                { CSS_PROF(CSS_PROF_OLA_STFT, ts); launch_ola_stft(sa, g_done, g_hi, ts); }
                if (last && pc + 1 == pieces) hipEventRecord(h->ev[5], ts);
                const int64_t q_hi = (g_hi == TL) ? TL + 1 : g_hi;   // the last range also writes the tail half-frame
                const int64_t f_lo = std::max<int64_t>(g_done - 1, 0);
                istft_gemm_on(h, f_lo, g_hi, ts);
                if (wav_mapped && piped) {   // (the copy stream belongs to the NEXT pass's samples by now)
                    wave_ola_on(h, f_lo, g_hi, g_done, q_hi, wav_mapped, io.cap, 0, ts);
                    hipEventRecord(h->ev[6], ts);
                } else if (wav_mapped) {   // the PCIe-bound overlap-add goes to the copy stream: the next piece's kernels run beside it
                    hipEvent_t done = pool_event(h);
                    HIPCHK(h, hipEventRecord(done, ts));
                    HIPCHK(h, hipStreamWaitEvent(h->copy_stream, done, 0));
                    wave_ola_on(h, f_lo, g_hi, g_done, q_hi, wav_mapped, io.cap, 0, h->copy_stream);
                    hipEventRecord(h->ev[6], h->copy_stream);
                    if (last && pc + 1 == pieces) {
                        out_done = pool_event(h);
                        HIPCHK(h, hipEventRecord(out_done, h->copy_stream));
                    }
                } else if (io.wav_dev) {   // device-resident output: straight into the caller's buffer
                    wave_ola_on(h, f_lo, g_hi, g_done, q_hi, io.wav_dev, io.cap, 0, ts);
                    hipEventRecord(h->ev[6], ts);
                } else {
                    wave_ola_on(h, f_lo, g_hi, g_done, q_hi, (float*)h->wav.p, pl.n_out, 0, ts);
                    hipEventRecord(h->ev[6], ts);
                }
                if (io.wav_host && !wav_mapped) {
                    const int64_t a = g_done * fhop, b = (g_hi == TL) ? pl.n_out : g_hi * fhop;
                    hipEvent_t done = pool_event(h);
                    HIPCHK(h, hipEventRecord(done, ts));
                    HIPCHK(h, hipStreamWaitEvent(h->copy_stream, done, 0));
                    for (int sp = 0; sp < S; ++sp)
                        HIPCHK(h, hipMemcpyAsync(io.wav_host + (size_t)sp * io.cap + a, (const float*)h->wav.p + (size_t)sp * pl.n_out + a,
                                                 (size_t)(b - a) * sizeof(float), hipMemcpyDeviceToHost, h->copy_stream));
                    if (last && pc + 1 == pieces) {
                        out_done = pool_event(h);
                        HIPCHK(h, hipEventRecord(out_done, h->copy_stream));
                    }
                }
                g_done = g_hi;
            }
        }
        return CSS_OK;
    };

    // ---- the estimator, unit by unit; each lane appends the beamformer of its own segments
    MaskIo mio{(const float*)h->X.p, h->T_ld, pl.stft_frames, hop, T, h->masks_v, h->mask_ld_v};
    mio.PH = (const float*)h->X.p + (int64_t)h->n_ch * 2 * F * h->T_ld;   // (every frame a segment reads was transformed in this pass)
    size_t ui = 0;
    const LanePrep prep = [&](int64_t seg_lo, int cnt, hipStream_t st) -> int {
        Unit& u = units[ui];
        if (u.seg_lo != seg_lo || u.n != cnt) return fail(h, CSS_ERR_STATE, "internal: unit schedule out of step");
        if (u.up) HIPCHK(h, hipStreamWaitEvent(st, u.up, 0));
        // this unit's segments also read frames (and its transform samples) that the units just before it produced on
        // other streams: a frame is read by at most ceil(T / hop) segments (each in its own unit at worst), a batch has
        // at most MAX_LANES lanes
        const size_t back = (size_t)std::max<int>(css_ctx::MAX_LANES, (T + hop - 1) / hop);
        for (size_t k = ui >= back ? ui - back : 0; k < ui; ++k)
            HIPCHK(h, hipStreamWaitEvent(st, units[k].x, 0));
        if (int e = stft_frames(h, u.f_lo, u.f_hi, planes_dev, st)) return e;
        HIPCHK(h, hipEventRecord(u.x, st));
        ++ui;
        return CSS_OK;
    };
    size_t first = 0;
    const LanePost post = [&](int64_t seg_lo, int cnt, hipStream_t st) -> int {
        Unit* u = nullptr;
        for (size_t k = first; k < ui; ++k)
            if (units[k].seg_lo == seg_lo && units[k].n == cnt) u = &units[k];
        if (!u) return fail(h, CSS_ERR_STATE, "internal: unit schedule out of step");
        if (h->tune[CSS_TUNE_MVDR_ON_LANES])
            if (int e = mvdr_on(h, seg_lo, seg_lo + cnt, st)) return e;
        HIPCHK(h, hipEventRecord(u->v, st));
        // raw stitching costs of this unit's boundaries (losses.py:50-71); the first one joins the previous unit's last
        // segment, whose masks / separated spectra are final once that unit's beamformer is
        if (u != &units[0]) HIPCHK(h, hipStreamWaitEvent(st, (u - 1)->v, 0));
        if (h->tune[CSS_TUNE_MVDR_ON_LANES]) pit_costs_on(h, std::max<int64_t>(seg_lo - 1, 0), seg_lo + cnt - 1, st);
        HIPCHK(h, hipEventRecord(u->m, st));
        return CSS_OK;
    };
    for (int64_t s0 = 0; s0 < nseg; s0 += cap) {
        first = ui;
        if ((rc = masknet_batch(h, mio, s0, (int)std::min<int64_t>(cap, nseg - s0), prep, post,
                                (piped && h->tail_pending && s0 == 0) ? h->tail_end : nullptr)) != CSS_OK) return rc;
        if (h->tune[CSS_TUNE_TAIL_PER_UNIT]) {
            for (size_t k = first; k < ui; ++k)
                if ((rc = tail_of(k, k + 1)) != CSS_OK) return rc;
        } else if ((rc = tail_of(first, ui)) != CSS_OK) {
            return rc;
        }
    }
    h->stft_done = h->perms_done = true;
    hipEventRecord(h->ev[3], h->stream);
    hipEventRecord(h->ev[4], h->stream);
    if (piped) {   // no join: the next queued pass's estimator runs beside this pass's tail; css_wait waits for all streams
        HIPCHK(h, hipEventRecord(h->pcm_free[par], h->stream));
        HIPCHK(h, hipEventRecord(h->tail_end, h->tail_stream));
        HIPCHK(h, hipEventRecord(h->level_free[par], h->tail_stream));
        HIPCHK(h, hipEventRecord(h->pass_end[h->pass_no & 3], h->tail_stream));
        hipEventRecord(h->ev[7], h->tail_stream);
        mark_session_done(h, h->tail_stream);
        h->tail_pending = true;
        h->pass_no += 1;
        h->queued += 1;
        HIPCHK(h, hipGetLastError());
        return CSS_OK;
    }
    // ---- join: the main stream continues after the tail (and the last download)
    hipEvent_t tail_done = pool_event(h);
    HIPCHK(h, hipEventRecord(tail_done, h->tail_stream));
    HIPCHK(h, hipStreamWaitEvent(h->stream, tail_done, 0));
    if (io.wav16_host) {
        const int64_t n_out = pl.n_out;
        unsigned int* pk = (unsigned int*)h->enc.p;
        int16_t* o16 = (int16_t*)((char*)h->enc.p + 64);
        { CSS_PROF(CSS_PROF_ENCODE, h->stream); launch_encode_pcm16((const float*)h->wav.p, S, n_out, pk, o16, n_out, h->stream); }
        HIPCHK(h, hipMemcpy2DAsync(io.wav16_host, (size_t)io.cap * sizeof(int16_t), o16, (size_t)n_out * sizeof(int16_t),
                                   (size_t)n_out * sizeof(int16_t), S, hipMemcpyDeviceToHost, h->stream));
        if (io.peaks_host) HIPCHK(h, hipMemcpyAsync(io.peaks_host, pk, (size_t)S * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    }
    // range check (split_f16.hpp): a split GEMM whose operand left the format's range raised this word
    HIPCHK(h, hipMemcpyAsync(h->range_flag_host, h->range_flag_dev, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
    if (out_done) HIPCHK(h, hipStreamWaitEvent(h->stream, out_done, 0));
    hipEventRecord(h->ev[7], h->stream);
    const auto host_t1 = std::chrono::steady_clock::now();
    if (io.enqueue_only) {   // css_wait synchronises, reads the range word and the timings of the last queued pass
        mark_session_done(h, h->stream);
        h->queued += 1;
        HIPCHK(h, hipGetLastError());
        return CSS_OK;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    const auto host_t2 = std::chrono::steady_clock::now();
    return finish_timings(h, host_t0, host_t1, host_t2, false);
}

// Several queued sessions as ONE pass of the mask estimator (css_run_enqueue).  Segments are independent through the whole
// network and every kernel of it is batch invariant (a row's bits do not depend on the launch shape), so the segments of
// G sessions go through features -> Conformer -> mask head as one [sum of segments x T, .] problem: every Linear-layer
// launch then has three times the rows of a 60 s meeting's (M = 22 320 for three of them), the regime where the same
// kernel runs at 0.31 - 0.34 of its ceiling instead of 0.26.  Everything around the estimator stays per
// session, on that session's own buffers (SessState): upload and analysis transform before, covariances / MVDR /
// beamformer / stitching costs after (the sessions dealt over the lanes' streams), then -- session by session on the tail
// stream, beside the NEXT pass's estimator -- permutation scan, overlap-add, gate, synthesis and the zero-copy overlap-add
// into the session's page-locked output.  The overlap protocol between consecutive passes is run_once's (sample-buffer
// halves and level words by pass parity, the mask head waits for the previous tail), so grouped and single passes may
// follow each other in one queue.  Results are bit for bit those of css_run on each session.
namespace {
// session j of a group of G lives in the handle itself (j == G - 1: the last session stays the handle's session, as after
// a single pass) or in slots[j]; Active swaps it in for the scope
struct Active {
    css_ctx* h; SessState* other;
    Active(css_ctx* h_, int j, int G) : h(h_), other(j == G - 1 ? nullptr : &h_->slots[(size_t)j]) {
        if (other) std::swap(static_cast<SessState&>(*h), *other);
    }
    ~Active() { if (other) std::swap(static_cast<SessState&>(*h), *other); }
};
}  // namespace

int run_group(css_handle_t h, std::vector<css_ctx::Pending>& grp) {
    const int G = (int)grp.size();
    const auto host_t0 = std::chrono::steady_clock::now();
    HIPCHK(h, hipSetDevice(h->device));
    if (h->queued && h->last_piped != 1) {   // a non-overlapping pass is queued in front: drain it on the device (see run_once)
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipStreamSynchronize(h->tail_stream));
        HIPCHK(h, hipStreamSynchronize(h->copy_stream));
        h->tail_pending = false;
    }
    h->last_piped = 1;
    const int par = (int)(h->pass_no & 1);
    if ((int)h->slots.size() < G - 1) h->slots.resize((size_t)(G - 1));
    const int S = h->d.num_spks, F = h->d.num_bins;
    int rc;
    // ---- the sessions: plans, buffers
    std::vector<int64_t> off((size_t)G), pcm_off((size_t)G);
    int64_t total = 0;
    size_t pcm_bytes = 0;
    for (int j = 0; j < G; ++j) {
        Active act(h, j, G);
        std::swap(h->X, h->X_alt);   // (every grouped pass takes the planes the previous one did not)
        h->peak_dev = (unsigned int*)h->level.p + 8 * par + j;
        h->piped_now = true;
        rc = begin_impl(h, grp[(size_t)j].n, grp[(size_t)j].n_ch, &grp[(size_t)j].cfg);
        h->piped_now = false;
        if (rc != CSS_OK) return rc;
        off[(size_t)j] = total;
        total += h->plan.num_segments;
        pcm_off[(size_t)j] = (int64_t)pcm_bytes;
        const bool s16 = !grp[(size_t)j].planes.empty();
        pcm_bytes += ((size_t)grp[(size_t)j].n * grp[(size_t)j].n_ch * (s16 ? sizeof(int16_t) : sizeof(float)) + 255) / 256 * 256;
        if (grp[(size_t)j].wav16 && (rc = ensure(h, h->enc, (size_t)h->d.num_spks * h->plan.n_out * sizeof(int16_t) + 64)) != CSS_OK) return rc;
    }
    const int T = grp[0].cfg.segment_frames, hop = grp[0].cfg.hop_frames;
    // a shared batch runs on at most TWO lanes: measured equal to three (profiles/r04_queue_group_ab.md), and it leaves the
    // hardware queue the tail stream shares with lane 2 (deal_streams) to the tail alone
    struct LaneGuard { css_ctx* h; int keep; ~LaneGuard() { h->lanes = keep; } } lane_guard{h, h->lanes};
    h->lanes = std::min(h->lanes, std::max(h->tune[CSS_TUNE_GROUP_LANES], 1));
    const bool xf_main = h->tune[CSS_TUNE_GROUP_TRANSFORM_ON_MAIN] != 0, mvdr_lanes = h->tune[CSS_TUNE_GROUP_MVDR_ON_LANES] != 0;
    if ((rc = ensure(h, h->pcm_in, 2 * pcm_bytes)) != CSS_OK) return rc;
    if ((rc = ensure(h, h->masks, (size_t)(S + 1) * F * total * T * sizeof(float))) != CSS_OK) return rc;
    if ((rc = ensure_activations(h, total, T)) != CSS_OK) return rc;
    const char* pcm_base = (const char*)h->pcm_in.p + (par ? h->pcm_in.cap / 2 / 256 * 256 : 0);
    std::vector<GroupSess> gs((size_t)G);
    for (int j = 0; j < G; ++j) {
        Active act(h, j, G);
        h->pcm_src = (const float*)(pcm_base + pcm_off[(size_t)j]);
        h->src16 = !grp[(size_t)j].planes.empty();
        h->masks_v = (float*)h->masks.p + off[(size_t)j] * T;
        h->mask_ld_v = total * T;
        gs[(size_t)j] = GroupSess{(const float*)h->X.p, h->T_ld, h->plan.stft_frames, off[(size_t)j], (int)h->plan.num_segments,
                                  (const float*)h->X.p + (int64_t)h->n_ch * 2 * F * h->T_ld};
    }
    h->ev_pool_used = 0;
    std::vector<hipEvent_t> planes((size_t)G), done((size_t)G);
    for (int j = 0; j < G; ++j) { planes[(size_t)j] = pool_event(h); done[(size_t)j] = pool_event(h); }
    // ---- the overlap protocol of queued passes (run_once, `piped`)
    for (int b = 0; b < 2; ++b) {
        if (!h->pcm_free[b]) HIPCHK(h, hipEventCreateWithFlags(&h->pcm_free[b], hipEventDisableTiming));
        if (!h->level_free[b]) HIPCHK(h, hipEventCreateWithFlags(&h->level_free[b], hipEventDisableTiming));
    }
    if (!h->tail_end) HIPCHK(h, hipEventCreateWithFlags(&h->tail_end, hipEventDisableTiming));
    for (auto& e : h->pass_end)
        if (!e) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (h->pass_no >= CSS_QUEUE_LEAD) HIPCHK(h, hipEventSynchronize(h->pass_end[(h->pass_no - CSS_QUEUE_LEAD) & 3]));
    if (!h->queued) {
        // whatever the handle's stream holds from BEFORE the queue (weights, an earlier synchronous pass) comes first.  Only
        // the first pass of a queue waits for it: a later pass's uploads are ordered by pcm_free / level_free / tail_end, and
        // a wait on the main stream here would put them behind the previous pass's estimator instead of beside it
        hipEvent_t opened = pool_event(h);
        HIPCHK(h, hipEventRecord(opened, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->copy_stream, opened, 0));
    }
    if (h->pass_no >= 2) {
        // this parity's sample buffer, level words and planes were last used by the pass before last: its transforms are
        // on this very stream; its beamformers (readers of the planes) and its tail (reader of the level words) ended with
        // level_free
        HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->pcm_free[par], 0));
        HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->level_free[par], 0));
    }
    HIPCHK(h, hipMemsetAsync((unsigned int*)h->level.p + 8 * par, 0, 8 * sizeof(unsigned int), h->copy_stream));
    // ---- copy stream: every session's samples as one piece, its level scanned and its analysis transform behind it --
    // all of it beside the PREVIOUS pass's estimator (the host runs passes ahead), so that the main stream carries nothing
    // but estimators, back to back
    for (int j = 0; j < G; ++j) {
        Active act(h, j, G);
        const css_ctx::Pending& q = grp[(size_t)j];
        if (h->src16) {   // the session's mono PCM16 planes, half the PCIe bytes (css_run_enqueue_pcm16)
            int16_t* dst16 = (int16_t*)const_cast<float*>(h->pcm_src);
            for (int c = 0; c < q.n_ch; ++c) {
                HIPCHK(h, hipMemcpyAsync(dst16 + (size_t)c * q.n, q.planes[(size_t)c], (size_t)q.n * sizeof(int16_t), hipMemcpyHostToDevice, h->copy_stream));
                launch_pcm_peak_i16(dst16 + (size_t)c * q.n, peak_len(h, 0, q.n), h->peak_dev, h->copy_stream);
            }
        } else {
            HIPCHK(h, hipMemcpyAsync(const_cast<float*>(h->pcm_src), q.pcm, (size_t)q.n * q.n_ch * sizeof(float), hipMemcpyHostToDevice,
                                     h->copy_stream));
            launch_pcm_peak_f32(h->pcm_src, peak_len(h, 0, q.n) * q.n_ch, h->peak_dev, h->copy_stream);
        }
        if (xf_main) {   // (A/B: the transforms as a prefix of the main stream)
            HIPCHK(h, hipEventRecord(planes[(size_t)j], h->copy_stream));
            continue;
        }
        if (h->plan.stft_frames < h->plan.mix_frames)   // short input: zero-padded frames (css.py:159-164)
            HIPCHK(h, hipMemsetAsync(h->X.p, 0, (size_t)h->n_ch * X_ROWS_PER_BIN * F * h->T_ld * sizeof(float), h->copy_stream));
        if ((rc = stft_frames(h, 0, h->plan.mix_frames, h->src16 ? (const int16_t*)h->pcm_src : nullptr, h->copy_stream)) != CSS_OK) return rc;
        h->stft_done = true;
        HIPCHK(h, hipEventRecord(planes[(size_t)j], h->copy_stream));
    }
    if (!xf_main) HIPCHK(h, hipEventRecord(h->pcm_free[par], h->copy_stream));
    hipEventRecord(h->ev[1], h->stream);
    for (int j = 0; j < G; ++j) {
        HIPCHK(h, hipStreamWaitEvent(h->stream, planes[(size_t)j], 0));
        if (!xf_main) continue;
        Active act(h, j, G);
        if (h->plan.stft_frames < h->plan.mix_frames)
            HIPCHK(h, hipMemsetAsync(h->X.p, 0, (size_t)h->n_ch * X_ROWS_PER_BIN * F * h->T_ld * sizeof(float), h->stream));
        if ((rc = stft_frames(h, 0, h->plan.mix_frames, h->src16 ? (const int16_t*)h->pcm_src : nullptr, h->stream)) != CSS_OK) return rc;
        h->stft_done = true;
    }
    if (xf_main) HIPCHK(h, hipEventRecord(h->pcm_free[par], h->stream));
    hipEventRecord(h->ev[2], h->stream);
    // ---- one estimator batch over all their segments (the mask head waits for the previous pass's tail: it overwrites
    // the mask buffer that tail reads)
    MaskIo io{nullptr, 0, 0, hop, T, (float*)h->masks.p, total * T, &gs};
    const LanePrep none = [](int64_t, int, hipStream_t) { return (int)CSS_OK; };
    if ((rc = masknet_batch(h, io, 0, (int)total, none, none, h->tail_pending ? h->tail_end : nullptr)) != CSS_OK) return rc;
    hipEventRecord(h->ev[3], h->stream);
    hipEvent_t masks_ready = pool_event(h);
    HIPCHK(h, hipEventRecord(masks_ready, h->stream));
    hipEventRecord(h->ev[4], h->stream);
    // ---- tail stream, session by session, beside the NEXT pass's estimator: covariances, MVDR, beamformer, stitching
    // costs, permutation scan, overlap-add, gate, synthesis, zero-copy overlap-add into the session's page-locked output
    hipStream_t ts = h->tail_stream;
    HIPCHK(h, hipStreamWaitEvent(ts, masks_ready, 0));
    if (mvdr_lanes) {   // (A/B: covariances / MVDR / costs dealt over the lanes' streams, the main stream waits for them)
        const LaneSplit ls = lane_split(h, (int)total, T);
        for (int l = 1; l < ls.nl && l < G; ++l) HIPCHK(h, hipStreamWaitEvent(h->lane_stream[l], masks_ready, 0));
        for (int j = 0; j < G; ++j) {
            Active act(h, j, G);
            hipStream_t st = (ls.nl > 1 && j % ls.nl) ? h->lane_stream[j % ls.nl] : h->stream;
            if ((rc = mvdr_on(h, 0, h->plan.num_segments, st)) != CSS_OK) return rc;
            pit_costs_on(h, 0, h->plan.num_segments - 1, st);
            HIPCHK(h, hipEventRecord(done[(size_t)j], st));
        }
        for (int j = 0; j < G; ++j)
            if (ls.nl > 1 && j % ls.nl) HIPCHK(h, hipStreamWaitEvent(h->stream, done[(size_t)j], 0));
    }
    if (!mvdr_lanes) {
        // Stage by stage over the group's sessions, so that the two stages that are chains per thread or per block -- the 7 x 7
        // solves (one thread per system, ~20 us whatever the launch holds) and the stitching costs -- are ONE launch for the
        // group instead of one per session (bit for bit the per-session launches' results: every system / boundary is computed
        // by the same code on the same operands).
        std::vector<MvdrArgs> ma((size_t)G), solve;
        for (int j = 0; j < G; ++j) {
            Active act(h, j, G);
            ma[(size_t)j] = mvdr_args(h, 0, (int)h->plan.num_segments);
            if (!ma[(size_t)j].use_mvdr || ma[(size_t)j].nseg <= 0) continue;
            CSS_PROF(CSS_PROF_SCM, ts);
            if (!launch_scm(ma[(size_t)j], ts)) return fail(h, CSS_ERR_HIP, "the covariance kernel's LDS could not be reserved");
            solve.push_back(ma[(size_t)j]);
        }
        if (!solve.empty()) { CSS_PROF(CSS_PROF_MVDR_SOLVE, ts); launch_mvdr_solve_multi(solve.data(), (int)solve.size(), ts); }
        std::vector<StitchArgs> sas((size_t)G);
        std::vector<double*> scr((size_t)G), cst((size_t)G);
        bool one_loss = true;
        int loss0 = 0, input0 = 0;
        for (int j = 0; j < G; ++j) {
            Active act(h, j, G);
            if (ma[(size_t)j].nseg > 0) {
                CSS_PROF(CSS_PROF_BEAMFORM, ts);
                launch_beamform(ma[(size_t)j], ts);
                if (h->cfg.normalize_segment_power) launch_segment_power_norm(ma[(size_t)j], (double*)h->pnorm.p, ts);
            }
            sas[(size_t)j] = stitch_args(h); scr[(size_t)j] = (double*)h->pit_part.p; cst[(size_t)j] = (double*)h->costs.p;
            if (j == 0) { loss0 = h->cfg.stitching_loss; input0 = h->cfg.stitching_input; }
            else one_loss = one_loss && loss0 == h->cfg.stitching_loss && input0 == h->cfg.stitching_input;
        }
        if (one_loss) {
            CSS_PROF(CSS_PROF_PIT, ts);
            launch_pit_costs_multi(sas.data(), scr.data(), cst.data(), G, loss0, input0, ts);
        } else {   // (sessions of one group with different stitching losses: their costs per session)
            for (int j = 0; j < G; ++j) { Active act(h, j, G); pit_costs_on(h, 0, h->plan.num_segments - 1, ts); }
        }
    }
    for (int j = 0; j < G; ++j) {
        Active act(h, j, G);
        const css_ctx::Pending& q = grp[(size_t)j];
        const int64_t nseg = h->plan.num_segments, TL = h->plan.mix_frames;
        if (mvdr_lanes) HIPCHK(h, hipStreamWaitEvent(ts, done[(size_t)j], 0));
        pit_scan_on(h, 0, nseg - 1, ts);
        const StitchArgs sa = stitch_args(h);
        { CSS_PROF(CSS_PROF_OLA_MASKS, ts); launch_ola_masks(sa, 0, TL, ts); }
        { CSS_PROF(CSS_PROF_GATE, ts); launch_morphology(sa, 0, TL, ts); }
        { CSS_PROF(CSS_PROF_OLA_STFT, ts); launch_ola_stft(sa, 0, TL, ts); }
        if (j == G - 1) hipEventRecord(h->ev[5], ts);
        istft_gemm_on(h, 0, TL, ts);
        if (q.wav16) {
            // the second wav edge on the device (utils/audio_utils.py:37-49 write_wav): peak normalisation and PCM16 encoding of
            // the session's streams, then half the PCIe bytes back -- css_run_pcm16's arithmetic, launch for launch
            const int64_t n_out = h->plan.n_out;
            if ((rc = ensure(h, h->wav, (size_t)S * n_out * sizeof(float))) != CSS_OK) return rc;
            wave_ola_on(h, 0, TL, 0, TL + 1, (float*)h->wav.p, n_out, 0, ts);
            unsigned int* pk = (unsigned int*)h->enc.p;
            int16_t* o16 = (int16_t*)((char*)h->enc.p + 64);
            { CSS_PROF(CSS_PROF_ENCODE, ts); launch_encode_pcm16((const float*)h->wav.p, S, n_out, pk, o16, n_out, ts); }
            HIPCHK(h, hipMemcpy2DAsync(q.wav16, (size_t)q.cap * sizeof(int16_t), o16, (size_t)n_out * sizeof(int16_t),
                                       (size_t)n_out * sizeof(int16_t), S, hipMemcpyDeviceToHost, ts));
            if (q.peaks) HIPCHK(h, hipMemcpyAsync(q.peaks, pk, (size_t)S * sizeof(float), hipMemcpyDeviceToHost, ts));
        } else if (h->tune[CSS_TUNE_GROUP_OUT_DMA]) {
            // the PCIe leg as copies behind a 12 us kernel: written by the kernel itself the same samples keep 11 250
            // workgroups resident for 0.21 ms per session, beside the next pass's estimator (profiles/r04_queue_group_ab.md)
            const int64_t n_out = h->plan.n_out;
            if ((rc = ensure(h, h->wav, (size_t)S * n_out * sizeof(float))) != CSS_OK) return rc;
            wave_ola_on(h, 0, TL, 0, TL + 1, (float*)h->wav.p, n_out, 0, ts);
            for (int sp = 0; sp < S; ++sp)
                HIPCHK(h, hipMemcpyAsync(q.wav + (size_t)sp * q.cap, (const float*)h->wav.p + (size_t)sp * n_out,
                                         (size_t)n_out * sizeof(float), hipMemcpyDeviceToHost, ts));
        } else {
            wave_ola_on(h, 0, TL, 0, TL + 1, q.wav_mapped, q.cap, 0, ts);
        }
        mark_session_done(h, ts);
        h->perms_done = true;
    }
    hipEventRecord(h->ev[6], ts);
    HIPCHK(h, hipEventRecord(h->tail_end, ts));
    HIPCHK(h, hipEventRecord(h->level_free[par], ts));
    HIPCHK(h, hipEventRecord(h->pass_end[h->pass_no & 3], ts));
    hipEventRecord(h->ev[7], ts);
    h->tail_pending = true;
    h->pass_no += 1;
    h->queued += 1;
    HIPCHK(h, hipGetLastError());
    h->tim.host_enqueue = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
    return CSS_OK;
}

// the sessions css_run_enqueue has accepted and not yet put on the streams: one alone takes run_once's pipeline (its
// lanes hide a single meeting's PCIe legs better), several take run_group
int flush_pending(css_handle_t h) {
    if (h->pending.empty()) return CSS_OK;
    std::vector<css_ctx::Pending> grp;
    grp.swap(h->pending);
    h->pending_segments = 0;
    int rc;
    if (grp.size() == 1) {
        RunIo io; io.pcm_host = grp[0].pcm; io.wav_host = grp[0].wav; io.cap = grp[0].cap; io.enqueue_only = true;
        if (!grp[0].planes.empty()) { io.planes_host = grp[0].planes.data(); io.wav16_host = grp[0].wav16; io.peaks_host = grp[0].peaks; }
        rc = run_once(h, grp[0].n, grp[0].n_ch, &grp[0].cfg, io);
    } else {
        rc = run_group(h, grp);
    }
    if (rc != CSS_OK) {
        // sessions css_run_enqueue had accepted are dropped with this error: take them out of the repeat log (they are its
        // last grp.size() entries -- nothing is logged between an acceptance and its flush) and name them
        const size_t drop = std::min(grp.size(), h->queue_log.size());
        const size_t first = h->queue_log.size() - drop;
        h->queue_log.resize(first, css_ctx::QueuedPass(nullptr, 0, 0, CssRunCfg{}, nullptr, 0));
        const std::string why = h->err;
        return fail(h, rc, "queued session(s) " + std::to_string(first) + " .. " + std::to_string(first + drop - 1) +
                               " (counted from the last css_wait) were accepted and could not be started; they are dropped: " + why);
    }
    return CSS_OK;
}

// The pass, and -- when an operand left the split-f16 range (a split GEMM saw a non-finite accumulator) -- the same pass
// again on the exact float32 kernels (css_set_range_fallback(h, 0): CSS_ERR_RANGE instead).
int run_impl(css_handle_t h, int64_t n, int32_t n_ch, const CssRunCfg* cfg, const RunIo& io) {
    int rc;
    if (h && (h->queued || !h->pending.empty()) && (rc = css_wait(h)) != CSS_OK) return rc;   // queued passes first (and their range verdict)
    rc = run_once(h, n, n_ch, cfg, io);
    if (rc != CSS_OK) return rc;
    h->range_last = 0;
    if (!*h->range_flag_host || !h->split) return CSS_OK;
    h->range_last = 1;
    if (!h->range_fallback)
        return fail(h, CSS_ERR_RANGE, "an operand of a Linear layer left the split-f16 range (|x| > 65504): use CSS_LINEAR_EXACT_F32");
    const CssTimings first = h->tim;
    if ((rc = css_set_linear_mode(h, CSS_LINEAR_EXACT_F32)) != CSS_OK) return rc;
    rc = run_once(h, n, n_ch, cfg, io);
    const int rc2 = css_set_linear_mode(h, CSS_LINEAR_SPLIT_F16);
    h->range_fallbacks += 1;
    h->tim.total += first.total;
    return rc != CSS_OK ? rc : rc2;
}

int css_run(css_handle_t h, const float* pcm_host, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg, float* wav_host,
            int64_t cap) {
    if (!h || !pcm_host || !wav_host) return fail(h, CSS_ERR_INVALID_ARG, "null argument");
    RunIo io; io.pcm_host = pcm_host; io.wav_host = wav_host; io.cap = cap;
    return run_impl(h, n_samples, n_ch, cfg, io);
}

// css_run_enqueue (float PCM -> float waveforms) and css_run_enqueue_pcm16 (PCM16 planes -> peak-normalised PCM16 streams): one
// queue, one grouping rule; a session is one or the other (planes == nullptr: float)
static int enqueue_impl(css_handle_t h, const float* pcm_host, const int16_t* const* planes, int64_t n_samples, int32_t n_ch,
                        const CssRunCfg* cfg, float* wav_host, int16_t* wav16, float* peaks, int64_t cap) {
    CssPlan pl{};
    int rc = check_run_args(h, n_samples, n_ch, cfg, &pl);
    if (rc != CSS_OK) return rc;
    if (cap < pl.n_out) return fail(h, CSS_ERR_INVALID_ARG, "output buffer too small: need " + std::to_string(pl.n_out) + " samples per stream");
    if (planes)
        for (int c = 0; c < n_ch; ++c)
            if (!planes[c]) return fail(h, CSS_ERR_INVALID_ARG, "null channel plane");
    // Can the session share an estimator batch with its neighbours in the queue?  It must take the overlapping form of a
    // queued pass (page-locked output, beamformer on the lanes) and fit a batch; sessions of another segmentation or
    // window start a new group.  A pass under the per-launch profile stays alone only when grouping is off.
    const void* out_key = wav16 ? (const void*)wav16 : (const void*)wav_host;
    if (h->mapped_key != out_key) { h->mapped_key = out_key; h->mapped_val = mapped_host(out_key); }
    float* mapped = (float*)h->mapped_val;   // (PCM16 output: only WHETHER it is page-locked matters -- it leaves by DMA)
    const bool groupable = h->fft512 && h->group_limit > 1 && mapped && h->tune[CSS_TUNE_MVDR_ON_LANES] && pl.num_segments <= batch_cap(h, cfg->segment_frames);
    auto log_entry = [&]() {
        h->queue_log.emplace_back(pcm_host, n_samples, n_ch, *cfg, wav_host, cap);
        if (planes) { css_ctx::QueuedPass& e = h->queue_log.back(); e.planes.assign(planes, planes + n_ch); e.wav16 = wav16; e.peaks = peaks; }
    };
    auto io_of = [&](bool enqueue_only) {
        RunIo io; io.pcm_host = pcm_host; io.wav_host = wav_host; io.cap = cap; io.enqueue_only = enqueue_only;
        if (planes) { io.planes_host = planes; io.wav16_host = wav16; io.peaks_host = peaks; }
        return io;
    };
    if (!h->fft512) {
        // Frame sizes other than 512 / 256 run the plain stage sequence to its end inside the call (run_once): nothing stays
        // queued, so css_wait would never look at the range word.  The pass therefore takes css_run's own rule here -- queued
        // passes first, then this one, repeated in float32 or refused with CSS_ERR_RANGE when it left the split-f16 range.
        rc = run_impl(h, n_samples, n_ch, cfg, io_of(false));
        if (rc == CSS_OK) h->sess_done.push_back(nullptr);   // (finished inside the call)
        return rc;
    }
    if (!groupable) {
        if ((rc = flush_pending(h)) != CSS_OK) return rc;
        rc = run_once(h, n_samples, n_ch, cfg, io_of(true));
        if (rc == CSS_OK) log_entry();
        return rc;
    }
    const int T = cfg->segment_frames;
    if (!h->pending.empty()) {
        const css_ctx::Pending& f = h->pending.front();
        const bool same = f.cfg.segment_frames == T && f.cfg.hop_frames == cfg->hop_frames &&
                          std::memcmp(f.w.data(), cfg->w_first, T * sizeof(float)) == 0 &&
                          std::memcmp(f.w.data() + T, cfg->w_mid, T * sizeof(float)) == 0 &&
                          std::memcmp(f.w.data() + 2 * T, cfg->w_last, T * sizeof(float)) == 0;
        if (!same || h->pending_segments + pl.num_segments > batch_cap(h, T) || (int)h->pending.size() >= h->group_limit)
            if ((rc = flush_pending(h)) != CSS_OK) return rc;
    }
    css_ctx::Pending q{pcm_host, n_samples, n_ch, *cfg, {}, wav_host, cap, mapped, pl.num_segments};
    if (planes) { q.planes.assign(planes, planes + n_ch); q.wav16 = wav16; q.peaks = peaks; }
    q.w.resize(3 * (size_t)T);
    std::memcpy(q.w.data(), cfg->w_first, T * sizeof(float));
    std::memcpy(q.w.data() + T, cfg->w_mid, T * sizeof(float));
    std::memcpy(q.w.data() + 2 * T, cfg->w_last, T * sizeof(float));
    h->pending.push_back(std::move(q));
    {   // (the vector may have moved: point the copies of the configuration at their own windows)
        for (css_ctx::Pending& e : h->pending) {
            e.cfg.w_first = e.w.data(); e.cfg.w_mid = e.w.data() + e.cfg.segment_frames; e.cfg.w_last = e.w.data() + 2 * e.cfg.segment_frames;
        }
    }
    h->pending_segments += pl.num_segments;
    log_entry();
    // no session of this length would still fit, or the group is full: off it goes -- nothing waits for a css_wait that
    // could already run
    if (h->pending_segments + pl.num_segments > batch_cap(h, cfg->segment_frames) || (int)h->pending.size() >= h->group_limit) return flush_pending(h);
    return CSS_OK;
}

int css_run_enqueue(css_handle_t h, const float* pcm_host, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg,
                    float* wav_host, int64_t cap) {
    if (!h || !pcm_host || !wav_host) return fail(h, CSS_ERR_INVALID_ARG, "null argument");
    return enqueue_impl(h, pcm_host, nullptr, n_samples, n_ch, cfg, wav_host, nullptr, nullptr, cap);
}

int css_run_enqueue_pcm16(css_handle_t h, const int16_t* const* planes_host, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg,
                          int16_t* wav_pcm16_host, int64_t cap, float* peaks_host) {
    if (!h || !planes_host || !wav_pcm16_host || n_samples < 1 || n_ch < 1) return fail(h, CSS_ERR_INVALID_ARG, "bad argument");
    return enqueue_impl(h, nullptr, planes_host, n_samples, n_ch, cfg, nullptr, wav_pcm16_host, peaks_host, cap);
}

int css_set_queue_group(css_handle_t h, int max_sessions) {
    CSS_DRAIN(h);
    if (!h || max_sessions < 1 || max_sessions > css_ctx::MAX_GROUP) return fail(h, CSS_ERR_INVALID_ARG, "max_sessions must be in [1, 8]");
    h->group_limit = max_sessions;
    return CSS_OK;
}

int css_wait(css_handle_t h) {
    if (!h) return CSS_ERR_INVALID_ARG;
    if (!h->pending.empty()) {
        const int rc_flush = flush_pending(h);
        if (rc_flush != CSS_OK) { h->queue_log.clear(); return rc_flush; }
    }
    if (!h->queued) { h->queue_log.clear(); h->sess_done.clear(); h->sess_ev_used = 0; return CSS_OK; }
    HIPCHK(h, hipSetDevice(h->device));
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->tail_stream));
    HIPCHK(h, hipStreamSynchronize(h->copy_stream));
    HIPCHK(h, hipMemcpy(h->range_flag_host, h->range_flag_dev, sizeof(unsigned int), hipMemcpyDeviceToHost));
    HIPCHK(h, hipGetLastError());
    const auto t1 = std::chrono::steady_clock::now();
    h->queued = 0;
    h->sess_done.clear();
    h->sess_ev_used = 0;
    h->tail_pending = false;
    h->last_piped = -1;
    h->pass_no = 0;
    h->peak_dev = (unsigned int*)h->level.p;
    finish_timings(h, t0, t0, t1, false);
    h->range_last = 0;
    std::vector<css_ctx::QueuedPass> log;
    log.swap(h->queue_log);
    if (*h->range_flag_host && h->split) {
        // the same rule as css_run: the queued passes accumulate into one range word, so every pass queued since the last
        // css_wait is repeated, one by one, on the exact float32 kernels (their inputs are still the caller's to keep)
        h->range_last = 1;
        if (!h->range_fallback)
            return fail(h, CSS_ERR_RANGE, "an operand of a Linear layer left the split-f16 range in one of the queued passes "
                                          "(|x| > 65504): use CSS_LINEAR_EXACT_F32");
        int rc = css_set_linear_mode(h, CSS_LINEAR_EXACT_F32);
        size_t repeated = 0;
        for (; repeated < log.size() && rc == CSS_OK; ++repeated) {
            const css_ctx::QueuedPass& q = log[repeated];
            RunIo io; io.pcm_host = q.pcm; io.wav_host = q.wav; io.cap = q.cap;
            if (!q.planes.empty()) { io.planes_host = q.planes.data(); io.wav16_host = q.wav16; io.peaks_host = q.peaks; }
            const CssRunCfg own = q.own_cfg();
            rc = run_once(h, q.n, q.n_ch, &own, io);
        }
        const std::string why = h->err;
        const int rc2 = css_set_linear_mode(h, CSS_LINEAR_SPLIT_F16);
        h->range_fallbacks += (int64_t)repeated;
        if (rc != CSS_OK)   // (which outputs are float32 results and which still hold the overflowed split-f16 ones)
            return fail(h, rc, "float32 repeat of the queued sessions stopped at session " + std::to_string(repeated - 1) + " of " +
                                   std::to_string(log.size()) + " (sessions before it hold their float32 results, it and the later ones do not): " + why);
        return rc2;
    }
    return CSS_OK;
}

int css_wait_sessions(css_handle_t h, int64_t n) {
    if (!h || n < 0) return fail(h, CSS_ERR_INVALID_ARG, "bad argument");
    if (n == 0) return CSS_OK;
    HIPCHK(h, hipSetDevice(h->device));
    if ((int64_t)h->sess_done.size() < n && !h->pending.empty()) {   // sessions still held back for company: off they go
        const int rc = flush_pending(h);
        if (rc != CSS_OK) return rc;
    }
    if ((int64_t)h->sess_done.size() < n)
        return fail(h, CSS_ERR_INVALID_ARG, "css_wait_sessions(" + std::to_string(n) + "): only " + std::to_string(h->sess_done.size()) +
                                                " sessions have been queued since the last css_wait");
    for (int64_t i = 0; i < n; ++i)
        if (h->sess_done[(size_t)i]) HIPCHK(h, hipEventSynchronize(h->sess_done[(size_t)i]));
    return CSS_OK;
}

int css_run_device(css_handle_t h, const float* pcm_dev, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg,
                   float* wav_dev, int64_t cap) {
    if (!h || !pcm_dev || !wav_dev) return fail(h, CSS_ERR_INVALID_ARG, "null argument");
    RunIo io; io.pcm_dev = pcm_dev; io.wav_dev = wav_dev; io.cap = cap;
    return run_impl(h, n_samples, n_ch, cfg, io);
}

int css_run_pcm16(css_handle_t h, const int16_t* const* planes_host, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg,
                  int16_t* wav_pcm16_host, int64_t cap, float* peaks_host) {
    if (!h || !planes_host || !wav_pcm16_host || n_samples < 1 || n_ch < 1) return fail(h, CSS_ERR_INVALID_ARG, "bad argument");
    RunIo io; io.planes_host = planes_host; io.wav16_host = wav_pcm16_host; io.peaks_host = peaks_host; io.cap = cap;
    return run_impl(h, n_samples, n_ch, cfg, io);
}


// libcss_mi355.so, host side 1 / 3: the handle's life cycle (css_create / css_destroy: weights, streams, tables), every setter and getter,
// timings and the per-launch profile, buffer access, the RCCL communicator.  (api_ctx.hpp: what the three units share.)
#include "api_ctx.hpp"

thread_local std::string g_create_error;

int fail(css_ctx* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    else g_create_error = msg;
    return code;
}

int ensure(css_ctx* h, DevBuf& b, size_t bytes, bool zero) {
    if (bytes <= b.cap) return CSS_OK;
    if (b.p && h->queued) HIPCHK(h, hipDeviceSynchronize());   // queued passes may still use the old allocation
    if (b.p) HIPCHK(h, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    HIPCHK(h, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    if (zero) HIPCHK(h, hipMemsetAsync(b.p, 0, bytes, h->stream));
    return CSS_OK;
}

// Walks the blob in the order documented in css_mi355.h; returns the number of floats consumed.
int64_t bind_weights(const CssModelDesc& d, const float* base, Weights* w) {
    const int64_t D = d.attention_dim, FF = d.linear_units, Kp = round_up(d.in_features, 32);
    const int64_t dk = D / d.attention_heads, ks = d.kernel_size;
    int64_t off = 0;
    auto take = [&](int64_t n) {
        const float* p = base ? base + off : nullptr;
        off += pad16(n);
        return p;
    };
    Weights tmp;
    Weights& W = w ? *w : tmp;
    W.input_bias = take(Kp);
    W.input_scale = take(Kp);
    W.embed_w = take(D * Kp);
    W.embed_b = take(D);
    W.embed_ln_w = take(D);
    W.embed_ln_b = take(D);
    W.pe_k = take(2 * (int64_t)d.maxlen * dk);
    W.blocks.resize(d.num_blocks);
    for (int l = 0; l < d.num_blocks; ++l) {
        BlockWeights& b = W.blocks[l];
        b.ffi_ln_w = take(D); b.ffi_ln_b = take(D); b.ffi_w1 = take(FF * D); b.ffi_b1 = take(FF);
        b.ffi_w2 = take(D * FF); b.ffi_b2 = take(D);
        b.att_ln_w = take(D); b.att_ln_b = take(D); b.wqkv = take(3 * D * D); b.bqkv = take(3 * D);
        b.wo = take(D * D); b.bo = take(D);
        b.conv_ln_w = take(D); b.conv_ln_b = take(D); b.pw = take(8); b.dw_wt = take(ks * D); b.dw_b = take(D);
        b.bn_alpha = take(D); b.bn_beta = take(D);
        b.ffo_ln_w = take(D); b.ffo_ln_b = take(D); b.ffo_w1 = take(FF * D); b.ffo_b1 = take(FF);
        b.ffo_w2 = take(D * FF); b.ffo_b2 = take(D);
        b.fin_ln_w = take(D); b.fin_ln_b = take(D);
    }
    const int64_t nout = (int64_t)d.num_bins * (d.num_spks + d.num_nois);
    W.head_w = take(nout * D);
    W.head_b = take(nout);
    return off;
}

const char* validate_desc(const CssModelDesc& d) {
    if (d.num_mics != 1 && d.num_mics != 7) return "num_mics must be 1 or 7";
    // init_kernel (feature.py:19-45): N FFT points (frame_len rounded up to a power of two, or frame_len itself), N/2 + 1 bins, a
    // window of frame_len samples, any hop.  frame_len 512 / hop 256 takes the FFT kernel and every pipelined schedule; other
    // sizes take the DFT-matrix product and the plain stage sequence
    if (d.num_bins < 2 || d.frame_len < 32 || d.frame_len > 2 * (d.num_bins - 1) || d.frame_len % 4)
        return "frame_len must be a multiple of 4, at least 32 and at most the FFT size 2 * (num_bins - 1)";
    if (d.frame_hop < 4 || d.frame_hop > d.frame_len || d.frame_hop % 4) return "frame_hop must be a multiple of 4 in [4, frame_len]";
    // magnitude block + one block per IPD pair (ipd_index; the shipped models: one pair per extra microphone -> 1799 / 257)
    if (d.num_bins <= 0 || d.in_features % d.num_bins || d.in_features / d.num_bins < 1 ||
        d.in_features / d.num_bins > 1 + CSS_MAX_IPD_PAIRS || (d.num_mics == 1 && d.in_features != d.num_bins))
        return "in_features must be num_bins * (1 + IPD pairs), at most 16 pairs (single-channel: num_bins)";
    if (d.attention_dim % 256 || d.attention_dim > 1024 || d.attention_dim <= 0) return "attention_dim must be a multiple of 256, at most 1024";
    if (d.attention_heads <= 0 || d.attention_dim / d.attention_heads != 64 || d.attention_dim % d.attention_heads) return "head size (attention_dim / attention_heads) must be 64";
    if (d.linear_units % 32 || d.linear_units <= 0) return "linear_units must be a multiple of 32";
    if (d.kernel_size != 33 && d.kernel_size != 31 && d.kernel_size != 17) return "kernel_size must be 33, 31 or 17";
    if (d.num_spks < 1 || d.num_spks > 3 || d.num_nois != 1) return "num_spks must be 1..3 and num_nois 1";
    if (d.num_blocks < 1 || d.maxlen < 1) return "num_blocks >= 1 and maxlen >= 1 required";
    return nullptr;
}

// cos/sin of 2 pi k / N with exact zeros / ones at the multiples of pi/2 (the DC and Nyquist sine rows
// must be exactly zero: see frontend.hip, PHASE_NEG_REAL)
void exact_cs(int64_t k, int N, double* c, double* s) {
    k %= N;
    const double ang = 2.0 * M_PI * (double)k / (double)N;
    *c = cos(ang);
    *s = sin(ang);
    if ((2 * k) % N == 0) *s = 0.0;
    if ((4 * k) % N == 0) *c = std::round(*c);
}

// ---- which hardware queue a stream lands on ------------------------------------------------------------------------
// The runtime deals its streams onto a few hardware queues (four by default) in creation order, and two streams on one
// queue run strictly one after the other: an upload on a "copy stream" that shares the main stream's queue starts
// only when the main stream's kernels are through, lanes that share a queue are no lanes at all.  Which streams collide
// depends on how many streams the PROCESS created before -- a handle created second, or on a stream torch made first, got
// a different deal (round 3: the same 30-min pass 7 % slower on such a handle, the first 22 MB piece of a sharded upload
// "taking" 14.9 ms because it waited for the 784 MB behind it; tools/rccl_slowdown_probe.py).  So css_create does not
// take the streams as they come: it creates candidates, MEASURES which ones can run beside the main stream and beside each
// other (a 200 us spin kernel on one, an empty kernel on the other), and deals them out itself.  With the usual four queues
// M (the main stream's), A, B, C:  lane 1 -> A,  lane 2 -> B,  copy -> C,  tail -> B,  lane 3 -> A.  The copy stream gets a
// queue to itself: in a queue of passes its uploads (and, for grouped passes, transforms) of pass P + 1 must run beside pass
// P's estimator, and anything else on its queue would hold them back -- the tail of pass P, enqueued before them, starts
// only when P's estimator ends (measured with copy and tail on one queue: 5.07 -> 5.65 ms per session).  The tail shares
// with lane 2: grouped passes use two lanes (run_group), which leaves that queue to the tail alone; a single queued pass
// with three lanes has its third lane start behind the previous tail, as it always did.
__global__ void css_spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();   // constant 100 MHz
    while (wall_clock64() - t0 < ticks) {}
}
__global__ void css_nop_kernel() {}

bool streams_share_a_queue(hipStream_t a, hipStream_t b, hipEvent_t e0, hipEvent_t e1) {
    hipStreamSynchronize(a);
    hipStreamSynchronize(b);
    hipEventRecord(e0, a);
    hipLaunchKernelGGL(css_spin_kernel, dim3(1), dim3(64), 0, a, (long long)20000);   // 200 us
    hipLaunchKernelGGL(css_nop_kernel, dim3(1), dim3(1), 0, b);
    hipEventRecord(e1, b);
    hipStreamSynchronize(a);
    hipStreamSynchronize(b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { (void)hipGetLastError(); return false; }
    return ms > 0.1f;
}

// streams for lanes 1 .. 3, the copy stream and the tail stream, none of them on `main`'s hardware queue where that can
// be had; false: something failed, the caller creates them plainly
bool deal_streams(hipStream_t main, hipStream_t lane[4], hipStream_t* copy, hipStream_t* tail) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return false;
    std::vector<std::vector<hipStream_t>> cls;   // classes of candidates that share a queue; none shares the main stream's
    std::vector<hipStream_t> with_main;
    auto enough = [&]() {
        if (cls.size() < 3) return false;
        std::vector<size_t> n;
        for (auto& c : cls) n.push_back(c.size());
        std::sort(n.begin(), n.end());
        return n[n.size() - 1] >= 2 && n[n.size() - 2] >= 2;   // two classes with two streams (lane 1 + lane 3, lane 2 + tail), one more for the copy stream
    };
    bool ok = true;
    for (int k = 0; k < 20 && ok && !enough(); ++k) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { ok = false; break; }
        if (streams_share_a_queue(main, st, e0, e1)) { with_main.push_back(st); continue; }
        bool placed = false;
        for (auto& c : cls)
            if (streams_share_a_queue(c[0], st, e0, e1)) { c.push_back(st); placed = true; break; }
        if (!placed) cls.push_back({st});
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (hipGetLastError() != hipSuccess) ok = false;
    std::vector<hipStream_t> take;   // lane1, lane2, copy, tail, lane3
    if (ok && !cls.empty()) {
        std::sort(cls.begin(), cls.end(), [](const std::vector<hipStream_t>& x, const std::vector<hipStream_t>& y) { return x.size() > y.size(); });
        // cls[0] (two streams or more) -> lane 1, lane 3;  cls[1] -> lane 2, tail;  cls[2] -> copy
        auto pop = [&](size_t c) -> hipStream_t {
            for (size_t q = 0; q < cls.size(); ++q) {
                auto& v = cls[(c + q) % cls.size()];
                if (!v.empty()) { hipStream_t s_ = v.back(); v.pop_back(); return s_; }
            }
            if (!with_main.empty()) { hipStream_t s_ = with_main.back(); with_main.pop_back(); return s_; }
            return nullptr;
        };
        const size_t nc = cls.size();
        const size_t cA = 0, cB = nc > 1 ? 1 : 0, cC = nc > 2 ? 2 : cB;
        take.resize(5);
        take[2] = pop(cC);   // copy first: a queue of its own if there is one
        take[0] = pop(cA);   // lane 1
        take[1] = pop(cB);   // lane 2
        take[3] = pop(cB);   // tail
        take[4] = pop(cA);   // lane 3
        for (hipStream_t s_ : take) ok = ok && s_ != nullptr;
    } else {
        ok = false;
    }
    for (auto& c : cls)
        for (hipStream_t s_ : c) hipStreamDestroy(s_);
    for (hipStream_t s_ : with_main) hipStreamDestroy(s_);
    if (!ok) {
        for (hipStream_t s_ : take)
            if (s_) hipStreamDestroy(s_);
        return false;
    }
    lane[1] = take[0]; lane[2] = take[1]; *copy = take[2]; *tail = take[3]; lane[3] = take[4];
    return true;
}

int plan_impl(const CssModelDesc& d, const CssRunCfg& cfg, int64_t n, CssPlan* p) {
    const int T = cfg.segment_frames, hop = cfg.hop_frames;
    if (T <= 0 || hop <= 0 || hop > T) return CSS_ERR_INVALID_ARG;
    p->n_samples = n;
    p->stft_frames = n < d.frame_len ? 0 : (n - d.frame_len) / d.frame_hop + 1;
    p->mix_frames = std::max<int64_t>(p->stft_frames, T);
    const int64_t ov = T - hop;
    p->num_segments = (p->mix_frames - ov + hop - 1) / hop;  // ceil((mix - ov)/hop)
    p->n_out = (p->mix_frames - 1) * d.frame_hop + d.frame_len;
    const int64_t st = (p->num_segments - 1) * hop;
    int64_t en = st + T;
    if (en >= p->mix_frames) en = p->mix_frames;
    p->last_valid = (int32_t)(en - st);
    // css.py:297: every frame must collect a total weight > 1e-5
    p->zero_weight = 0;
    if (cfg.w_first && cfg.w_mid && cfg.w_last) {
        for (int64_t t = 0; t < p->mix_frames && !p->zero_weight; ++t) {
            float ws = 0.f;
            for (int64_t seg = std::max<int64_t>(0, (t - T + hop) / hop); seg <= t / hop && seg < p->num_segments; ++seg) {
                const int64_t tl = t - seg * hop;
                if (tl < 0 || tl >= T) continue;
                const float* w = seg == 0 ? cfg.w_first : (seg == p->num_segments - 1 ? cfg.w_last : cfg.w_mid);
                ws += w[tl];
            }
            if (!(ws > 1e-5f)) p->zero_weight = 1;
        }
    }
    return CSS_OK;
}

void gemm(css_ctx* h, const GemmArgs& g, hipStream_t st) {
    CSS_PROF(CSS_PROF_LINEAR, st);
    if (h->profile_gemm) h->gemm_flops += 2.0 * g.M * (double)g.N * g.K * g.batch;
    if (!g.split_in && !g.layout && h->tune[CSS_TUNE_F32_GEMM]) {   // (A/B and tests: which exact float32 kernel; same bits)
        GemmArgs q = g;
        const int t = h->tune[CSS_TUNE_F32_GEMM];
        q.layout = t == 1 ? 2 : (t == 6 ? 1 : 10 + std::min(t - 1, 4));
        launch_gemm(q, st);
        return;
    }
    launch_gemm(g, st);
}

GemmArgs linear(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                int M, int N, int K, int act) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.strideA = 0;
    g.B = W; g.ldb = ldw; g.strideB = 0;
    g.C = C; g.ldc = ldc; g.strideC = 0;
    g.M = M; g.N = N; g.K = K; g.batch = 1;
    g.bias = bias; g.bias_along_m = 0; g.act = act;
    g.residual = nullptr; g.ldr = 0; g.alpha = 1.f;
    return g;
}

StitchArgs stitch_args(css_ctx* h) {
    StitchArgs a{};
    const int T = h->cfg.segment_frames;
    a.masks = h->masks_v;
    a.mask_ld = h->mask_ld_v;
    a.sep = (const float*)h->sep.p;
    a.S = h->d.num_spks; a.F = h->d.num_bins; a.T = T; a.hop = h->cfg.hop_frames;
    a.num_segments = h->plan.num_segments; a.T_long = h->plan.mix_frames;
    a.w_first = (const float*)h->segw.p; a.w_mid = a.w_first + T; a.w_last = a.w_mid + T;
    a.perms = (const int32_t*)h->perms.p;
    a.mask_st = (float*)h->mask_st.p; a.activity = (float*)h->activity.p;
    a.act_b = (uint8_t*)h->act_b.p; a.act_tmp = (uint8_t*)h->act_tmp.p; a.act_final = (uint8_t*)h->act_final.p;
    a.activity_th = h->cfg.activity_th; a.dilation = h->cfg.dilation_frames; a.erosion = h->cfg.erosion_frames;
    a.Y = (float*)h->Y.p; a.KIp = h->KIp;
    a.y_split = h->split ? 1 : 0;
    a.level = h->split ? h->peak_dev : nullptr;
    return a;
}

MvdrArgs mvdr_args(css_ctx* h, int64_t lo, int nseg) {
    MvdrArgs a{};
    const int T = h->cfg.segment_frames;
    a.X = (const float*)h->X.p; a.T_ld = h->T_ld; a.stft_frames = h->plan.stft_frames;
    a.C = h->n_ch; a.F = h->d.num_bins;
    a.masks = h->masks_v; a.mask_ld = h->mask_ld_v;
    a.S = h->d.num_spks; a.T = T; a.hop = h->cfg.hop_frames;
    a.seg_lo = lo; a.nseg = nseg;
    a.wta_override = h->have_override ? (const uint8_t*)h->wta.p : nullptr;
    a.scm = (double*)h->scm.p; a.bfw = (double*)h->bfw.p; a.sep = (float*)h->sep.p;
    a.mask_floor = h->cfg.mask_floor;
    a.use_mvdr = (h->n_ch > 1 && h->cfg.mc_mvdr) ? 1 : 0;
    return a;
}

// Segments per estimator batch.  max_batch_segments is the caller's bound (and the size of the workspace); in the split-f16
// mode a batch is also kept to CSS_TUNE_SPLIT_BATCH_ROWS token rows (24 576 = 128 segments of 3 s): its kernels are paced by
// the memory system, and beyond that the activations of a batch no longer pass from producer to consumer inside the 256 MB
// Infinity Cache (six 60 s sessions per batch instead of three: - 2 %, four: - 4.5 %).  The exact float32 mode is bound by
// its matrix products and gains from every row a launch adds (six sessions per batch: + 2.7 %).  Results do not depend on
// the batch (every kernel is batch invariant).
int64_t batch_cap(const css_ctx* h, int T) {
    int64_t cap = h->max_batch;
    const int rows = h->tune[CSS_TUNE_SPLIT_BATCH_ROWS];
    if (h->split && rows > 0 && T > 0) cap = std::min<int64_t>(cap, std::max<int64_t>(1, rows / T));
    return cap;
}

// activation workspace of the mask estimator for batches of up to `nb` segments of T frames
int ensure_activations(css_ctx* h, int64_t nb, int T) {
    const int64_t Mb = nb * T;
    const int D = h->d.attention_dim, FF = h->d.linear_units;
    int rc;
    if ((rc = ensure(h, h->feat, (size_t)Mb * h->Kp * sizeof(float), true)) != CSS_OK) return rc;
    if ((rc = ensure(h, h->hx, (size_t)Mb * D * sizeof(float))) != CSS_OK) return rc;
    if ((rc = ensure(h, h->hu, (size_t)Mb * D * sizeof(float))) != CSS_OK) return rc;
    if ((rc = ensure(h, h->ht, (size_t)Mb * FF * sizeof(float))) != CSS_OK) return rc;
    if ((rc = ensure(h, h->qkv, (size_t)Mb * 3 * D * sizeof(float))) != CSS_OK) return rc;
    // q and k in the attention kernel's operand order (split mode); zeroed once: rows past T of a last tile are never written
    if ((rc = ensure(h, h->qkf, (size_t)qk_fragment_floats(nb, T, h->d.attention_heads) * sizeof(float), true)) != CSS_OK) return rc;
    if ((rc = ensure(h, h->ctxb, (size_t)Mb * D * sizeof(float))) != CSS_OK) return rc;
    for (int l = 1; l < h->lanes; ++l) {   // lanes 1.. hold at most ceil(nb / 2) segments (lane_split may use fewer lanes than h->lanes)
        const int64_t M2 = ((nb + 1) / 2) * T;
        if ((rc = ensure(h, h->lfeat[l], (size_t)M2 * h->Kp * sizeof(float), true)) != CSS_OK) return rc;
        if ((rc = ensure(h, h->lhx[l], (size_t)M2 * D * sizeof(float))) != CSS_OK) return rc;
        if ((rc = ensure(h, h->lhu[l], (size_t)M2 * D * sizeof(float))) != CSS_OK) return rc;
        if ((rc = ensure(h, h->lht[l], (size_t)M2 * FF * sizeof(float))) != CSS_OK) return rc;
        if ((rc = ensure(h, h->lqkv[l], (size_t)M2 * 3 * D * sizeof(float))) != CSS_OK) return rc;
        if ((rc = ensure(h, h->lqkf[l], (size_t)qk_fragment_floats(M2 / T, T, h->d.attention_heads) * sizeof(float), true)) != CSS_OK) return rc;
        if ((rc = ensure(h, h->lctx[l], (size_t)M2 * D * sizeof(float))) != CSS_OK) return rc;
    }
    return CSS_OK;
}

// split-f16 images of every Linear weight, at the blob's own offsets (a split matrix has the size of its source):
// tile-major (gemm_split_wd.hip: the weight operand goes straight from global memory into MFMA registers) for the
// layers whose weight is the B operand, row-major for the mask head, where the weight is the A operand.
int make_split_weights(css_ctx* h) {
    if (h->wsplit) return CSS_OK;
    const CssModelDesc& d = h->d;
    const int64_t need = bind_weights(d, nullptr, nullptr);
    HIPCHK(h, hipMalloc((void**)&h->wsplit, need * sizeof(float)));
    const int D = d.attention_dim, FF = d.linear_units;
    auto conv = [&](const float* w, int rows, int K) {
        launch_split_convert_tiled(w, K, h->wsplit + (w - h->blob), rows, K, h->stream);
    };
    conv(h->w.embed_w, D, h->Kp);
    for (const BlockWeights& b : h->w.blocks) {
        conv(b.ffi_w1, FF, D); conv(b.ffi_w2, D, FF);
        conv(b.wqkv, 3 * D, D); conv(b.wo, D, D);
        conv(b.ffo_w1, FF, D); conv(b.ffo_w2, D, FF);
    }
    launch_split_convert(h->w.head_w, D, h->wsplit + (h->w.head_w - h->blob), (int64_t)d.num_bins * (d.num_spks + d.num_nois), D,
                         D, h->stream);
    {
        const int nout = d.num_bins * (d.num_spks + d.num_nois);
        HIPCHK(h, hipMalloc((void**)&h->head_tiled, (size_t)((nout + 31) / 32 * 32) * D * sizeof(float)));
        launch_split_convert_tiled(h->w.head_w, D, h->head_tiled, nout, D, h->stream);
    }
    // the synthesis transform matrix, row-major split
    HIPCHK(h, hipMalloc((void**)&h->dft_split, (size_t)d.frame_len * h->KIp * sizeof(float)));
    launch_split_convert(h->dft_inv_t, h->KIp, h->dft_split, d.frame_len, h->KIp, h->KIp, h->stream);
    HIPCHK(h, hipMalloc((void**)&h->dft_tiled, (size_t)((d.frame_len + 31) / 32 * 32) * h->KIp * sizeof(float)));
    launch_split_convert_tiled(h->dft_inv_t, h->KIp, h->dft_tiled, d.frame_len, h->KIp, h->stream);
    // the relative-position table, row-major split: the attention kernel uses its rows like key rows
    const int dk = D / d.attention_heads;
    launch_split_convert(h->w.pe_k, dk, h->wsplit + (h->w.pe_k - h->blob), 2 * (int64_t)d.maxlen, dk, dk, h->stream);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

// CSS_LINEAR_EXACT_F32: the Linear weights once more, float32 in the fragment order of gemm_f32.hip (GemmArgs::b_frag32): a
// wave reads its 32 columns' operands of a slab as two coalesced 1 KiB loads straight into registers.  Built on first use.
int make_frag_weights(css_ctx* h) {
    if (h->wfrag) return CSS_OK;
    const CssModelDesc& d = h->d;
    const int64_t need = bind_weights(d, nullptr, nullptr);
    HIPCHK(h, hipMalloc((void**)&h->wfrag, need * sizeof(float)));
    const int D = d.attention_dim, FF = d.linear_units;
    auto conv = [&](const float* w, int rows, int K) { launch_f32_fragments(w, K, h->wfrag + (w - h->blob), rows, K, h->stream); };
    conv(h->w.embed_w, D, h->Kp);
    for (const BlockWeights& b : h->w.blocks) {
        conv(b.ffi_w1, FF, D); conv(b.ffi_w2, D, FF);
        conv(b.wqkv, 3 * D, D); conv(b.wo, D, D);
        conv(b.ffo_w1, FF, D); conv(b.ffo_w2, D, FF);
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

int check_session(css_ctx* h) {
    if (!h) return CSS_ERR_INVALID_ARG;
    if (!h->has_session) return fail(h, CSS_ERR_STATE, "no session: call css_begin first");
    return CSS_OK;
}


// =================================================================================================

const char* css_version(void) { return "css_mi355 0.1 (gfx950)"; }

const char* css_last_error(css_handle_t h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int css_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int64_t css_blob_num_floats(const CssModelDesc* desc) {
    if (!desc || validate_desc(*desc)) return -1;
    return bind_weights(*desc, nullptr, nullptr);
}

// The analysis kernel of any frame size as a matrix (feature.py:19-45): rows f < F: cos(2 pi f n / NF) w[n], rows F + f:
// -sin(2 pi f n / NF) w[n], n < frame_len, zero up to Lp.  The sines of DC and Nyquist are exact zeros (hazard 2).
int upload_analysis_matrix(css_ctx* h, int window) {
    const int L = h->d.frame_len, F = h->d.num_bins, NF = 2 * (F - 1), Lp = h->Lp;
    std::vector<float> m((size_t)2 * F * Lp, 0.f);
    const double S = window == CSS_WINDOW_SQRT_HANN ? 0.5 * std::sqrt((double)NF * NF / h->d.frame_hop) : 1.0;
    for (int n = 0; n < L; ++n) {
        const double wn = 0.5 - 0.5 * cos(2.0 * M_PI * n / L);
        const double w = window == CSS_WINDOW_SQRT_HANN ? (double)(float)std::sqrt((float)wn) / S : wn;
        for (int f = 0; f < F; ++f) {
            double c, s_;
            exact_cs((int64_t)f * n, NF, &c, &s_);
            m[(size_t)f * Lp + n] = (float)(c * w);
            m[(size_t)(F + f) * Lp + n] = (float)(0.0 - s_ * w);
        }
    }
    return hipMemcpy(h->dft_fwd, m.data(), m.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess ? (int)CSS_OK : (int)CSS_ERR_HIP;
}

// Frames [t_lo, t_hi) of C channels (channel c's samples at x + c x_stride, zero or finite up to 32 floats past the last
// frame) -> planes out[(c 2F + r) row_ld + t].  frame_len 512 / hop 256: the LDS-staged FFT (+ the phase planes when asked);
// any other size: DFT matrix x overlapping frames on the exact float32 GEMM -- the frames ARE the rows of the B operand,
// row stride = hop -- and no phase planes (*phase_done = false: the feature kernel forms the angles itself).
bool analysis_transform(css_ctx* h, const float* x, int64_t x_stride, int C, int64_t t_lo, int64_t t_hi, float* out,
                               int64_t row_ld, hipStream_t st, float* phase, bool* phase_done) {
    if (phase_done) *phase_done = false;
    if (t_hi <= t_lo) return true;
    if (h->fft512) {
        if (phase_done) *phase_done = phase != nullptr;
        return launch_stft_fft(x, x_stride, C, t_lo, t_hi, h->stft_tab, out, row_ld, st, phase);
    }
    const int F = h->d.num_bins;
    GemmArgs g{};
    g.A = h->dft_fwd; g.lda = h->Lp; g.strideA = 0;
    g.B = x + t_lo * h->d.frame_hop; g.ldb = h->d.frame_hop; g.strideB = x_stride;
    g.C = out + t_lo; g.ldc = row_ld; g.strideC = (int64_t)2 * F * row_ld;
    g.M = 2 * F; g.N = (int)(t_hi - t_lo); g.K = h->Lp; g.batch = C; g.alpha = 1.f;
    launch_gemm(g, st);
    return true;
}

int css_create(const CssModelDesc* desc, const float* blob_host, int64_t blob_floats, int device, void* stream,
               int32_t max_batch_segments, css_handle_t* out) {
    if (!desc || !blob_host || !out) return fail(nullptr, CSS_ERR_INVALID_ARG, "null argument");
    if (const char* why = validate_desc(*desc)) return fail(nullptr, CSS_ERR_INVALID_ARG, why);
    const int64_t need = bind_weights(*desc, nullptr, nullptr);
    if (blob_floats != need)
        return fail(nullptr, CSS_ERR_INVALID_ARG, "weight blob has " + std::to_string(blob_floats) + " floats, expected " + std::to_string(need));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, CSS_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(nullptr, CSS_ERR_NO_DEVICE, "device index out of range");
    css_ctx* h = new css_ctx();
    h->d = *desc;
    // feature extractor: the shipped configuration (ExtractorCfg defaults, conformer_wrapper.py:11-24): magnitude with mean /
    // variance normalisation, IPD version 1 as raw angles, pairs (m, 0)
    h->feat_opts = FeatOpts{};
    h->feat_opts.mvn = 1; h->feat_opts.ipd_norm = 1; h->feat_opts.ipd_version = 1;
    // (a model with another number of pairs gets its ipd_index through css_set_feature_options; until then pair p = (p + 1, 0))
    h->feat_opts.num_pairs = desc->in_features / desc->num_bins - 1;
    for (int p_ = 0; p_ < h->feat_opts.num_pairs && p_ < 16; ++p_) {
        h->feat_opts.pair_l[p_] = (unsigned char)std::min(p_ + 1, desc->num_mics - 1);
        h->feat_opts.pair_r[p_] = 0;
    }
    h->device = device;
    h->max_batch = max_batch_segments > 0 ? max_batch_segments : 64;
    h->Kp = round_up(desc->in_features, 32);
    h->KIp = round_up(2 * desc->num_bins, 32);
    auto bail = [&](int code, const std::string& msg) {
        g_create_error = msg.empty() ? h->err : msg;
        css_destroy(h);
        return code;
    };
    if (hipSetDevice(device) != hipSuccess) return bail(CSS_ERR_HIP, "hipSetDevice failed");
    if (stream) {
        h->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreate(&h->stream) != hipSuccess) return bail(CSS_ERR_HIP, "hipStreamCreate failed");
        h->own_stream = true;
    }
    for (auto& e : h->ev)
        if (hipEventCreate(&e) != hipSuccess) return bail(CSS_ERR_HIP, "hipEventCreate failed");
    if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess)
        return bail(CSS_ERR_HIP, "hipEventCreate failed");
    const bool dealt = deal_streams(h->stream, h->lane_stream, &h->copy_stream, &h->tail_stream);
    for (int l = 1; l < css_ctx::MAX_LANES; ++l)
        if ((!dealt && hipStreamCreateWithFlags(&h->lane_stream[l], hipStreamNonBlocking) != hipSuccess) ||
            hipEventCreateWithFlags(&h->ev_join[l], hipEventDisableTiming) != hipSuccess)
            return bail(CSS_ERR_HIP, "lane stream / event could not be created");
    if (!dealt && (hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess ||
                   hipStreamCreateWithFlags(&h->tail_stream, hipStreamNonBlocking) != hipSuccess))
        return bail(CSS_ERR_HIP, "copy / tail stream could not be created");
    if (hipMalloc(&h->level.p, 64) != hipSuccess || hipMemset(h->level.p, 0, 64) != hipSuccess ||
        hipMalloc((void**)&h->range_flag_dev, 64) != hipSuccess ||
        hipHostMalloc((void**)&h->range_flag_host, 64, hipHostMallocDefault) != hipSuccess)
        return bail(CSS_ERR_HIP, "range flag could not be allocated");
    *h->range_flag_host = 0;
    h->level.cap = 64;
    h->peak_dev = (unsigned int*)h->level.p;
    if (hipMalloc((void**)&h->blob, need * sizeof(float)) != hipSuccess) return bail(CSS_ERR_HIP, "hipMalloc(weights) failed");
    if (hipMemcpy(h->blob, blob_host, need * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return bail(CSS_ERR_HIP, "weight upload failed");
    bind_weights(*desc, h->blob, &h->w);
    // transforms (feature.py:19-45): analysis = Hann-windowed 512-point FFT (stft.hip); synthesis = a GEMM with the
    // matrix sqrt-Hann * DFT / 16 (its output rows overlap-add, and its input is the stitched spectra in GEMM row format)
    // L window samples, NF FFT points (feature.py:27: frame_len rounded up to a power of two, or frame_len), F = NF / 2 + 1
    const int L = desc->frame_len, F = desc->num_bins, KI = h->KIp, NF = 2 * (F - 1);
    h->fft512 = L == 512 && desc->frame_hop == 256 && F == 257;
    h->Lp = round_up(L, 32);
    h->ovl = (L + desc->frame_hop - 1) / desc->frame_hop;
    std::vector<float> inv((size_t)L * KI, 0.f), tab(stft_table_floats());
    stft_build_tables(tab.data());
    const double S = 0.5 * std::sqrt((double)NF * NF / desc->frame_hop);
    for (int n = 0; n < L; ++n) {
        const double wn = 0.5 - 0.5 * cos(2.0 * M_PI * n / L);  // torch.hann_window(frame_len) (periodic)
        const double ws = (double)(float)std::sqrt((float)wn);  // W ** 0.5 on the float32 window
        for (int f = 0; f < F; ++f) {
            double c, s;
            exact_cs((int64_t)f * n, NF, &c, &s);
            inv[(size_t)n * KI + f] = (float)(c * ws / S);
            inv[(size_t)n * KI + F + f] = (float)(0.0 - s * ws / S);
        }
    }
    if (hipMalloc((void**)&h->stft_tab, tab.size() * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&h->dft_inv_t, inv.size() * sizeof(float)) != hipSuccess)
        return bail(CSS_ERR_HIP, "hipMalloc(transform tables) failed");
    if (hipMemcpy(h->stft_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->dft_inv_t, inv.data(), inv.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return bail(CSS_ERR_HIP, "transform table upload failed");
    if (!h->fft512) {
        if (hipMalloc((void**)&h->dft_fwd, (size_t)2 * F * h->Lp * sizeof(float)) != hipSuccess) return bail(CSS_ERR_HIP, "hipMalloc(analysis matrix) failed");
        if (upload_analysis_matrix(h, CSS_WINDOW_HANN) != CSS_OK) return bail(CSS_ERR_HIP, "analysis matrix upload failed");
    }
    // a weight beyond the split-f16 operand range (never seen in a trained checkpoint; weights are O(1)): this model
    // runs on the exact float32 kernels only, css_set_linear_mode(h, CSS_LINEAR_SPLIT_F16) is refused
    h->split_ok = true;
    for (int64_t i = 0; i < need && h->split_ok; ++i)
        if (!(std::fabs(blob_host[i]) <= 65504.f)) h->split_ok = false;
    // the handle starts in the reference's arithmetic (float32 operands); the split-f16 images of the weights are built by the
    // first css_set_linear_mode(h, CSS_LINEAR_SPLIT_F16)
    h->split = false;
    if (make_frag_weights(h) != CSS_OK) { (void)hipGetLastError(); if (h->wfrag) { hipFree(h->wfrag); h->wfrag = nullptr; } }
    *out = h;
    return CSS_OK;
}

int css_destroy(css_handle_t h) {
    if (!h) return CSS_OK;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->comm) css_comm_destroy(h);
    DevBuf* bufs[] = {&h->pcm_in, &h->pcm_cm, &h->X, &h->feat, &h->hx, &h->hu, &h->ht, &h->qkv, &h->qkf, &h->ctxb, &h->masks,
                      &h->scm, &h->bfw, &h->sep, &h->costs, &h->perms, &h->mask_st, &h->activity, &h->act_b,
                      &h->act_tmp, &h->act_final, &h->Y, &h->G, &h->wav, &h->wta, &h->pnorm, &h->segw, &h->stage, &h->pit_part,
                      &h->in16, &h->pcm_f, &h->enc, &h->level, &h->mel_tab, &h->mel_work, &h->X_alt};
    for (int l = 1; l < css_ctx::MAX_LANES; ++l) {
        for (DevBuf* b : {&h->lfeat[l], &h->lhx[l], &h->lhu[l], &h->lht[l], &h->lqkv[l], &h->lqkf[l], &h->lctx[l]})
            if (b->p) hipFree(b->p);
        if (h->lane_stream[l]) { hipStreamSynchronize(h->lane_stream[l]); hipStreamDestroy(h->lane_stream[l]); }
        if (h->ev_join[l]) hipEventDestroy(h->ev_join[l]);
    }
    for (DevBuf* b : bufs)
        if (b->p) hipFree(b->p);
    for (SessState& sl : h->slots)
        for (DevBuf* b : {&sl.pcm_cm, &sl.X, &sl.scm, &sl.bfw, &sl.sep, &sl.costs, &sl.perms, &sl.mask_st, &sl.activity, &sl.act_b,
                          &sl.act_tmp, &sl.act_final, &sl.Y, &sl.G, &sl.wav, &sl.wta, &sl.pnorm, &sl.pit_part, &sl.X_alt})
            if (b->p) hipFree(b->p);
    if (h->blob) hipFree(h->blob);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->tail_end) hipEventDestroy(h->tail_end);
    for (auto& e : h->pcm_free) if (e) hipEventDestroy(e);
    for (auto& e : h->level_free) if (e) hipEventDestroy(e);
    for (auto& e : h->pass_end) if (e) hipEventDestroy(e);
    if (h->copy_stream) { hipStreamSynchronize(h->copy_stream); hipStreamDestroy(h->copy_stream); }
    if (h->tail_stream) { hipStreamSynchronize(h->tail_stream); hipStreamDestroy(h->tail_stream); }
    if (h->range_flag_dev) hipFree(h->range_flag_dev);
    if (h->range_flag_host) hipHostFree(h->range_flag_host);
    for (auto& e : h->ev_pool) hipEventDestroy(e);
    for (auto& e : h->sess_ev_pool) hipEventDestroy(e);
    if (h->wsplit) hipFree(h->wsplit);
    if (h->wfrag) hipFree(h->wfrag);
    if (h->dft_split) hipFree(h->dft_split);
    if (h->dft_tiled) hipFree(h->dft_tiled);
    if (h->head_tiled) hipFree(h->head_tiled);
    for (auto& b : h->pe_frag)
        if (b.p) hipFree(b.p);
    if (h->stft_tab) hipFree(h->stft_tab);
    if (h->dft_fwd) hipFree(h->dft_fwd);
    if (h->dft_inv_t) hipFree(h->dft_inv_t);
    for (auto& e : h->ev)
        if (e) hipEventDestroy(e);
    for (auto& pr : h->prof_events) { hipEventDestroy(pr.a); hipEventDestroy(pr.b); }
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
    return CSS_OK;
}

int css_plan(const CssModelDesc* desc, const CssRunCfg* cfg, int64_t n_samples, CssPlan* out) {
    if (!desc || !cfg || !out || n_samples < 0) return CSS_ERR_INVALID_ARG;
    return plan_impl(*desc, *cfg, n_samples, out);
}

// css.py:341-390 calc_segment_weight; the taper is torch.linspace(0.1, 1, m1 - m0) in float32: ATen evaluates the lower half as
// start + step * i and the upper half as end - step * (steps - 1 - i), each one fused multiply-add
static void segment_weight(int T, int m0, int m1, bool first, bool last, float* w) {
    for (int t = 0; t < T; ++t) w[t] = 1.f;
    for (int t = 0; t < m0; ++t) w[t] = w[T - 1 - t] = 0.f;
    const int steps = m1 - m0;
    const float start = 0.1f, end = 1.f;
    const float step = steps > 1 ? (end - start) / (float)(steps - 1) : 0.f;
    for (int i = 0; i < steps; ++i) {
        const float v = steps == 1 ? start : (i < steps / 2 ? fmaf(step, (float)i, start) : fmaf(-step, (float)(steps - 1 - i), end));
        w[m0 + i] = v;
        w[T - 1 - m0 - i] = v;
    }
    if (first) for (int t = 0; t < m0; ++t) w[t] = 0.1f;
    if (last) for (int t = 0; t < m0; ++t) w[T - 1 - t] = 0.1f;
}

int css_make_run_cfg(const CssModelDesc* desc, const CssCfgSeconds* c, int32_t fs, CssRunCfg* out, float* windows, int64_t cap) {
    if (!desc || !c || !out || !windows || fs <= 0 || desc->frame_len < 2 || desc->frame_hop < 1 || !(c->segment_size_sec > 0)) return CSS_ERR_INVALID_ARG;
    // (a double outside int64's range has no defined conversion: every duration must be a finite, sane number of seconds;
    //  the floor in dB may be -inf -- css.py:41 sc_mask_floor_db -- but not NaN or +inf)
    const double secs[6] = {c->segment_size_sec, c->hop_size_sec, c->seg_weight_m0_sec, c->seg_weight_m1_sec, c->activity_dilation_sec, c->activity_erosion_sec};
    for (double v : secs)
        if (!std::isfinite(v) || v < 0.0 || v > 86400.0) return CSS_ERR_INVALID_ARG;
    if (!std::isfinite(c->activity_th) || std::isnan(c->mask_floor_db)) return CSS_ERR_INVALID_ARG;
    if (c->stitching_loss < 0 || c->stitching_loss > 1 || c->stitching_input < 0 || c->stitching_input > 1) return CSS_ERR_INVALID_ARG;
    // Python's int() truncates towards zero, // floors (css.py:145-152; the operands are non-negative here)
    const int64_t seg_samples = (int64_t)(c->segment_size_sec * (double)fs);
    if (seg_samples < desc->frame_len) return CSS_ERR_INVALID_ARG;
    const int64_t T = (seg_samples - desc->frame_len) / desc->frame_hop + 1;
    const int64_t hop = (int64_t)((double)T * c->hop_size_sec / c->segment_size_sec);
    if (T < 2 || T > CSS_MAX_SEGMENT_FRAMES || hop < 1 || hop >= T) return CSS_ERR_INVALID_ARG;
    const int m0 = (int)((double)T * c->seg_weight_m0_sec / c->segment_size_sec), m1 = (int)((double)T * c->seg_weight_m1_sec / c->segment_size_sec);
    if (m0 < 0 || m1 < m0) return CSS_ERR_INVALID_ARG;
    if (!(T > 2 * m1)) return CSS_ERR_WEIGHT_WINDOW;   // css.py:374
    if (c->mask_floor_db > 0) return CSS_ERR_MASK_FLOOR;   // css.py:224
    if (cap < 3 * T) return CSS_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    out->segment_frames = (int32_t)T;
    out->hop_frames = (int32_t)hop;
    out->dilation_frames = (int32_t)((double)T * c->activity_dilation_sec / c->segment_size_sec);
    out->erosion_frames = (int32_t)((double)T * c->activity_erosion_sec / c->segment_size_sec);
    out->mc_mvdr = c->mc_mvdr != 0;
    out->stitching_loss = c->stitching_loss;
    out->stitching_input = c->stitching_input;
    out->normalize_segment_power = c->normalize_segment_power != 0;
    out->mask_floor = (float)std::pow(10.0, c->mask_floor_db / 20.0);   // css.py:225 (-inf dB -> 0)
    out->activity_th = (float)c->activity_th;
    segment_weight((int)T, m0, m1, true, false, windows);
    segment_weight((int)T, m0, m1, false, false, windows + T);
    segment_weight((int)T, m0, m1, false, true, windows + 2 * T);
    out->w_first = windows; out->w_mid = windows + T; out->w_last = windows + 2 * T;
    return CSS_OK;
}

int css_pit_scan(const double* costs, int64_t n_boundaries, int32_t num_spks, int32_t* perms) {
    if (!perms || n_boundaries < 0 || num_spks < 1 || num_spks > 4 || (n_boundaries > 0 && !costs)) return CSS_ERR_INVALID_ARG;
    pit_scan_host(costs, n_boundaries, num_spks, perms);
    return CSS_OK;
}


int css_get_stream(css_handle_t h, void** stream_out) {
    if (!h || !stream_out) return CSS_ERR_INVALID_ARG;
    *stream_out = (void*)h->stream;
    return CSS_OK;
}

int css_set_lanes(css_handle_t h, int lanes) {
    CSS_DRAIN(h);
    if (!h || lanes < 1 || lanes > css_ctx::MAX_LANES) return fail(h, CSS_ERR_INVALID_ARG, "lanes must be in [1, 4]");
    h->lanes = lanes;   // the lanes' activation buffers are sized by the next css_begin / css_run* / css_*_host call
    return CSS_OK;
}

int css_get_lanes(css_handle_t h) { return h ? h->lanes : (int)CSS_ERR_INVALID_ARG; }

int css_set_tuning(css_handle_t h, int which, int value) {
    if (!h || which < 0 || which >= CSS_TUNE_COUNT || value < 0 || (value > 16 && which != CSS_TUNE_SPLIT_BATCH_ROWS && which != CSS_TUNE_F32_LANE_ROWS))
        return fail(h, CSS_ERR_INVALID_ARG, "unknown tuning option / value");
    h->tune[which] = value;
    return CSS_OK;
}

int css_set_range_fallback(css_handle_t h, int enable) {
    if (!h) return CSS_ERR_INVALID_ARG;
    h->range_fallback = enable != 0;
    return CSS_OK;
}

int css_range_status(css_handle_t h, int64_t* fallbacks, int32_t* last_hit) {
    if (!h) return CSS_ERR_INVALID_ARG;
    if (fallbacks) *fallbacks = h->range_fallbacks;
    if (last_hit) *last_hit = h->range_last;
    return CSS_OK;
}

int css_check_range(css_handle_t h) {
    int rc = check_session(h);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(h->range_flag_host, h->range_flag_dev, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (*h->range_flag_host && h->split)
        return fail(h, CSS_ERR_RANGE, "an operand of a Linear layer left the split-f16 range (|x| > 65504): use CSS_LINEAR_EXACT_F32");
    return CSS_OK;
}

// torch.nn.Linear on caller data through one of the path's three GEMM kernels (unit tests of the arithmetic).
int css_linear_host(css_handle_t h, const float* x, const float* w, const float* bias, int32_t M, int32_t N, int32_t K,
                    int32_t kernel, int32_t layout, float* y) {
    CSS_DRAIN(h);
    if (!h || !x || !w || !y || M < 1 || N < 1 || K < 32 || K % 32) return fail(h, CSS_ERR_INVALID_ARG, "bad argument (K must be a multiple of 32)");
    if (kernel < 0 || kernel > 2) return fail(h, CSS_ERR_INVALID_ARG, "kernel must be 0 (split, weights direct), 1 (split, LDS staged) or 2 (exact float32)");
    HIPCHK(h, hipSetDevice(h->device));
    const int Np = (N + 31) / 32 * 32;
    const size_t xf = (size_t)M * K, wf = (size_t)Np * K, yf = (size_t)M * N;
    int rc;
    if ((rc = ensure(h, h->stage, (2 * xf + 2 * wf + yf + (size_t)N + 256) * sizeof(float))) != CSS_OK) return rc;
    float* xd = (float*)h->stage.p;
    float* xs = xd + (xf + 15) / 16 * 16;
    float* wd = xs + (xf + 15) / 16 * 16;
    float* ws = wd + (wf + 15) / 16 * 16;
    float* yd = ws + (wf + 15) / 16 * 16;
    float* bd = yd + (yf + 15) / 16 * 16;
    HIPCHK(h, hipMemcpyAsync(xd, x, xf * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(wd, w, (size_t)N * K * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (bias) HIPCHK(h, hipMemcpyAsync(bd, bias, (size_t)N * sizeof(float), hipMemcpyHostToDevice, h->stream));
    GemmArgs g = linear(xd, K, wd, K, bias ? bd : nullptr, yd, N, M, N, K, ACT_NONE);
    if (kernel != 2) {
        launch_split_convert(xd, K, xs, M, K, K, h->stream);
        if (kernel == 0) launch_split_convert_tiled(wd, K, ws, N, K, h->stream);
        else launch_split_convert(wd, K, ws, N, K, K, h->stream);
        g.A = xs; g.B = ws; g.split_in = 1; g.b_tiled = kernel == 0;
        if (kernel == 0) g.tile_rows = layout; else g.layout = layout;
    } else {
        g.layout = layout;
    }
    launch_gemm(g, h->stream);
    HIPCHK(h, hipMemcpyAsync(y, yd, yf * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

int css_host_alloc(size_t bytes, void** out) {
    if (!out) return CSS_ERR_INVALID_ARG;
    *out = nullptr;
    return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? CSS_OK : CSS_ERR_HIP;
}

int css_host_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? CSS_OK : CSS_ERR_HIP; }

int css_set_linear_mode(css_handle_t h, int mode) {
    CSS_DRAIN(h);
    if (!h || (mode != CSS_LINEAR_SPLIT_F16 && mode != CSS_LINEAR_EXACT_F32)) return fail(h, CSS_ERR_INVALID_ARG, "unknown linear mode");
    const bool split = mode == CSS_LINEAR_SPLIT_F16;
    if (split == h->split) return CSS_OK;
    if (split && !h->split_ok) return fail(h, CSS_ERR_RANGE, "a weight of this model lies outside the split-f16 operand range (|w| > 65504)");
    HIPCHK(h, hipSetDevice(h->device));
    if (split) {
        int rc = make_split_weights(h);
        if (rc) return rc;
    } else if (make_frag_weights(h) != CSS_OK) {
        (void)hipGetLastError();   // (no room for the second image: the float32 kernel takes the row-major weights through LDS)
        if (h->wfrag) { hipFree(h->wfrag); h->wfrag = nullptr; }
    }
    // the feature rows change format; their K padding must read as zero in either
    if (h->feat.p) HIPCHK(h, hipMemsetAsync(h->feat.p, 0, h->feat.cap, h->stream));
    for (int l = 1; l < css_ctx::MAX_LANES; ++l)
        if (h->lfeat[l].p) HIPCHK(h, hipMemsetAsync(h->lfeat[l].p, 0, h->lfeat[l].cap, h->stream));
    h->split = split;
    return CSS_OK;
}

int css_set_feature_options(css_handle_t h, const CssFeatureCfg* c) {
    CSS_DRAIN(h);
    if (!h || !c) return fail(h, CSS_ERR_INVALID_ARG, "null argument");
    const int C = h->d.num_mics, F = h->d.num_bins;
    if (c->num_pairs < 0 || c->num_pairs > CSS_MAX_IPD_PAIRS) return fail(h, CSS_ERR_INVALID_ARG, "at most 16 IPD pairs");
    if (C == 1 && c->num_pairs != 0) return fail(h, CSS_ERR_INVALID_ARG, "a single-channel model has no IPD pairs");
    if (h->d.in_features != F * (1 + c->num_pairs))
        return fail(h, CSS_ERR_SHAPE, "in_features = " + std::to_string(h->d.in_features) + " does not match num_bins * (1 + " +
                                          std::to_string(c->num_pairs) + " IPD pairs)");
    if (c->ipd_mean_normalize && (c->ipd_mean_normalize_version < 1 || c->ipd_mean_normalize_version > 3))
        return fail(h, CSS_ERR_INVALID_ARG, "ipd_mean_normalize_version must be 1, 2 or 3 (feature.py:228-231)");
    FeatOpts o{};
    o.log_mag = c->log_spectrogram != 0; o.mvn = c->mvn_spectrogram != 0; o.ipd_norm = c->ipd_mean_normalize != 0;
    o.ipd_version = c->ipd_mean_normalize_version; o.ipd_cos = c->ipd_cos != 0; o.num_pairs = c->num_pairs;
    for (int p = 0; p < c->num_pairs; ++p) {
        if (c->pair_l[p] < 0 || c->pair_l[p] >= C || c->pair_r[p] < 0 || c->pair_r[p] >= C)
            return fail(h, CSS_ERR_INVALID_ARG, "IPD pair index outside the model's microphones");
        o.pair_l[p] = (unsigned char)c->pair_l[p];
        o.pair_r[p] = (unsigned char)c->pair_r[p];
    }
    h->feat_opts = o;
    return CSS_OK;
}

int css_set_analysis_window(css_handle_t h, int32_t window) {
    CSS_DRAIN(h);
    if (!h) return CSS_ERR_INVALID_ARG;
    if (window != CSS_WINDOW_HANN && window != CSS_WINDOW_SQRT_HANN)
        return fail(h, CSS_ERR_INVALID_ARG, "the analysis window is 'hann' or 'sqrt_hann' (feature.py:24-25)");
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<float> tab(stft_table_floats());
    stft_build_tables(tab.data(), window);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(h->stft_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!h->fft512 && upload_analysis_matrix(h, window) != CSS_OK) return fail(h, CSS_ERR_HIP, "analysis matrix upload failed");
    return CSS_OK;
}

int css_get_linear_mode(css_handle_t h) {
    if (!h) return CSS_ERR_INVALID_ARG;
    return h->split ? CSS_LINEAR_SPLIT_F16 : CSS_LINEAR_EXACT_F32;
}

int css_set_profile(css_handle_t h, int enable) {
    CSS_DRAIN(h);
    if (!h) return CSS_ERR_INVALID_ARG;
    h->profile_gemm = enable != 0;
    if (enable && !h->prof_pairs) {
        // calibration: 64 empty brackets behind a kernel each (the bracket's cost depends on the stream being busy)
        HIPCHK(h, hipSetDevice(h->device));
        constexpr int NP = 64;
        hipEvent_t ev[2 * NP];
        for (auto& e : ev) HIPCHK(h, hipEventCreate(&e));
        unsigned int* scratch = h->range_flag_dev + 8;   // (a word of the 64-byte allocation nobody reads)
        for (int i = 0; i < NP; ++i) {
            HIPCHK(h, hipMemsetAsync(scratch, 0, 4, h->stream));
            HIPCHK(h, hipEventRecord(ev[2 * i], h->stream));
            HIPCHK(h, hipEventRecord(ev[2 * i + 1], h->stream));
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        float tot = 0.f;
        for (int i = 0; i < NP; ++i) { float v = 0.f; hipEventElapsedTime(&v, ev[2 * i], ev[2 * i + 1]); tot += v; }
        for (auto& e : ev) hipEventDestroy(e);
        h->prof_pair_ms = tot;
        h->prof_pairs = NP;
    }
    return CSS_OK;
}

// staged sessions (css_begin + css_stage_*) have no closing call: their brackets are summed when the figures are asked for
static void reduce_pending_profile(css_ctx* h) {
    if (!h->profile_gemm || h->prof_reduced == h->prof_used) return;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);   // (the lanes' streams are joined into it behind every estimator batch)
    reduce_profile(h);
}

int css_get_kernel_stats(css_handle_t h, CssKernelStat* out, int32_t cap, int32_t* count) {
    if (!h || !count || (cap > 0 && !out)) return CSS_ERR_INVALID_ARG;
    reduce_pending_profile(h);
    int n = 0;
    for (int c = 0; c < CSS_PROF_COUNT; ++c) {
        if (!h->prof_launches[c]) continue;
        if (n < cap) {
            std::snprintf(out[n].name, sizeof(out[n].name), "%s", kProfNames[c]);
            out[n].ms = h->prof_ms[c];
            out[n].launches = h->prof_launches[c];
        }
        ++n;
    }
    if (h->prof_pairs) {   // not a kernel family: the empty-bracket calibration (ms over `launches` empty brackets)
        if (n < cap) {
            std::snprintf(out[n].name, sizeof(out[n].name), "%s", "event_pair_overhead");
            out[n].ms = h->prof_pair_ms;
            out[n].launches = h->prof_pairs;
        }
        ++n;
    }
    *count = n;
    return CSS_OK;
}

int css_get_timings(css_handle_t h, CssTimings* out) {
    if (!h || !out) return CSS_ERR_INVALID_ARG;
    reduce_pending_profile(h);
    *out = h->tim;
    return CSS_OK;
}

int css_get_plan(css_handle_t h, CssPlan* out) {
    CSS_DRAIN(h);   // (a session css_run_enqueue holds back becomes the handle's session when it runs)
    int rc = check_session(h);
    if (rc) return rc;
    if (!out) return CSS_ERR_INVALID_ARG;
    *out = h->plan;
    return CSS_OK;
}


// -------------------------------------------------------------------------------------------------
static int buffer_info(css_ctx* h, int which, DevBuf** buf, int64_t dims[4], int32_t* elem) {
    const int F = h->d.num_bins, S = h->d.num_spks, T = h->cfg.segment_frames;
    const int64_t nseg = h->plan.num_segments, TL = h->plan.mix_frames;
    dims[0] = dims[1] = dims[2] = dims[3] = 1;
    *elem = 4;
    switch (which) {
        case CSS_BUF_X: *buf = &h->X; dims[0] = h->n_ch; dims[1] = 2 * F; dims[2] = h->T_ld; break;
        case CSS_BUF_FEATURES: *buf = &h->feat; dims[0] = h->last_batch_tokens; dims[1] = h->Kp; break;
        case CSS_BUF_MASKS:   // (a session of a queued group holds its masks as columns of the group's buffer: not readable)
            if (h->masks_v != (float*)h->masks.p || h->mask_ld_v != nseg * T)
                return fail(h, CSS_ERR_STATE, "the masks of a session that shared a queued estimator batch are columns of the group's "
                                              "buffer: not addressable as CSS_BUF_MASKS (css_write_buffer re-homes them)");
            *buf = &h->masks; dims[0] = (int64_t)(S + 1) * F; dims[1] = nseg * T; break;
        case CSS_BUF_SCM: *buf = &h->scm; dims[0] = nseg; dims[1] = S + 1; dims[2] = F; dims[3] = 49; *elem = 8; break;
        case CSS_BUF_BFW: *buf = &h->bfw; dims[0] = nseg; dims[1] = S; dims[2] = F; dims[3] = 14; *elem = 8; break;
        case CSS_BUF_SEP: *buf = &h->sep; dims[0] = nseg; dims[1] = S; dims[2] = F; dims[3] = (int64_t)T * 2; break;
        case CSS_BUF_PIT_COST: *buf = &h->costs; dims[0] = std::max<int64_t>(nseg - 1, 0); dims[1] = S * S; *elem = 8; break;
        case CSS_BUF_PERMS: *buf = &h->perms; dims[0] = nseg; dims[1] = S; break;
        case CSS_BUF_MASK_ST: *buf = &h->mask_st; dims[0] = S; dims[1] = F; dims[2] = TL; break;
        case CSS_BUF_ACTIVITY: *buf = &h->activity; dims[0] = S; dims[1] = TL; break;
        case CSS_BUF_ACT_B: *buf = &h->act_b; dims[0] = S; dims[1] = TL; *elem = 1; break;
        case CSS_BUF_ACT_FINAL: *buf = &h->act_final; dims[0] = S; dims[1] = TL; *elem = 1; break;
        case CSS_BUF_Y: *buf = &h->Y; dims[0] = S; dims[1] = TL; dims[2] = h->KIp; break;
        case CSS_BUF_WAV: *buf = &h->wav; dims[0] = S; dims[1] = h->plan.n_out; break;
        case CSS_BUF_HIDDEN: *buf = &h->hx; dims[0] = h->last_batch_tokens; dims[1] = h->d.attention_dim; break;
        case CSS_BUF_WTA_OVERRIDE: *buf = &h->wta; dims[0] = nseg; dims[1] = F; dims[2] = T; *elem = 1; break;
        case CSS_BUF_LEVEL: *buf = &h->level; dims[0] = 1; break;
        default: return CSS_ERR_INVALID_ARG;
    }
    return CSS_OK;
}

int css_buffer_dims(css_handle_t h, int which, int64_t dims[4], int32_t* elem_bytes) {
    CSS_DRAIN(h);
    int rc = check_session(h);
    if (rc) return rc;
    DevBuf* b;
    if (!dims || !elem_bytes) return fail(h, CSS_ERR_INVALID_ARG, "null argument");
    if ((rc = buffer_info(h, which, &b, dims, elem_bytes)) != CSS_OK) return rc == CSS_ERR_STATE ? rc : fail(h, CSS_ERR_INVALID_ARG, "unknown buffer");
    return CSS_OK;
}

int css_read_buffer(css_handle_t h, int which, void* host, int64_t nbytes) {
    CSS_DRAIN(h);
    int rc = check_session(h);
    if (rc) return rc;
    DevBuf* b;
    int64_t dims[4];
    int32_t el;
    if (!host) return fail(h, CSS_ERR_INVALID_ARG, "null host pointer");
    if ((rc = buffer_info(h, which, &b, dims, &el)) != CSS_OK) return rc == CSS_ERR_STATE ? rc : fail(h, CSS_ERR_INVALID_ARG, "unknown buffer");
    const int64_t need = dims[0] * dims[1] * dims[2] * dims[3] * el;
    if (nbytes != need || !b->p) return fail(h, CSS_ERR_INVALID_ARG, "buffer size mismatch: expected " + std::to_string(need) + " bytes");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(host, b->p, (size_t)need, hipMemcpyDeviceToHost));
    if ((which == CSS_BUF_FEATURES || which == CSS_BUF_Y) && h->split) {
        // the device holds the rows as split-f16 GEMM operands (split_f16.hpp); hand out float32 = hi + lo * 2^-11
        const int64_t rows = which == CSS_BUF_Y ? dims[0] * dims[1] : dims[0], K = which == CSS_BUF_Y ? dims[2] : dims[1];
        float unscale = 1.f;
        if (which == CSS_BUF_Y) {   // the rows carry the session's level gain (split_f16.hpp level_gain)
            float peak = 0.f;
            HIPCHK(h, hipMemcpy(&peak, h->peak_dev, sizeof(float), hipMemcpyDeviceToHost));
            if (peak > 0.f && peak < 3.0e38f) {
                int e;
                std::frexp(peak, &e);
                unscale = std::ldexp(1.f, std::min(std::max(e, -100), 100));
            }
        }
        std::vector<float> row((size_t)K);
        for (int64_t r = 0; r < rows; ++r) {
            float* dst = (float*)host + r * K;
            const _Float16* src = (const _Float16*)dst;
            for (int64_t k = 0; k < K; ++k) {
                const int64_t i = ((k >> 5) << 6) | (k & 31);
                row[(size_t)k] = ((float)src[i] + (float)src[i + 32] * (1.0f / 2048.0f)) * unscale;
            }
            std::memcpy(dst, row.data(), (size_t)K * sizeof(float));
        }
    }
    return CSS_OK;
}

int css_write_buffer(css_handle_t h, int which, const void* host, int64_t nbytes) {
    CSS_DRAIN(h);
    int rc = check_session(h);
    if (rc) return rc;
    DevBuf* b;
    int64_t dims[4];
    int32_t el;
    if (!host) return fail(h, CSS_ERR_INVALID_ARG, "null host pointer");
    // written masks are the session's own [(S+1) F][nseg T] matrix (also after a grouped pass, whose sessions see columns of the
    // group's buffer): the size is checked against THAT shape, and the session's view moves only once the call can no longer fail
    const float* keep_v = h->masks_v;
    const int64_t keep_ld = h->mask_ld_v;
    if (which == CSS_BUF_MASKS) { h->masks_v = (float*)h->masks.p; h->mask_ld_v = h->plan.num_segments * h->cfg.segment_frames; }
    auto restore = [&](int code) { if (which == CSS_BUF_MASKS) { h->masks_v = const_cast<float*>(keep_v); h->mask_ld_v = keep_ld; } return code; };
    if ((rc = buffer_info(h, which, &b, dims, &el)) != CSS_OK) return restore(rc == CSS_ERR_STATE ? rc : fail(h, CSS_ERR_INVALID_ARG, "unknown buffer"));
    const int64_t need = dims[0] * dims[1] * dims[2] * dims[3] * el;
    if (nbytes != need) return restore(fail(h, CSS_ERR_INVALID_ARG, "buffer size mismatch: expected " + std::to_string(need) + " bytes"));
    if (hipSetDevice(h->device) != hipSuccess) return restore(fail(h, CSS_ERR_HIP, "hipSetDevice failed"));
    if ((rc = ensure(h, *b, (size_t)need)) != CSS_OK) return restore(rc);
    if (which == CSS_BUF_MASKS) h->masks_v = (float*)h->masks.p;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(b->p, host, (size_t)need, hipMemcpyHostToDevice));
    if (which == CSS_BUF_X) h->ph_valid = false;   // (the planes no longer come from the transform: phases are formed in the feature kernel)
    if (which == CSS_BUF_WTA_OVERRIDE) h->have_override = true;
    if (which == CSS_BUF_PERMS) h->perms_done = true;
    return CSS_OK;
}

int css_buffer_devptr(css_handle_t h, int which, void** out) {
    CSS_DRAIN(h);
    int rc = check_session(h);
    if (rc) return rc;
    DevBuf* b;
    int64_t dims[4];
    int32_t el;
    if (!out) return fail(h, CSS_ERR_INVALID_ARG, "null argument");
    if ((rc = buffer_info(h, which, &b, dims, &el)) != CSS_OK) return rc == CSS_ERR_STATE ? rc : fail(h, CSS_ERR_INVALID_ARG, "unknown buffer");
    *out = b->p;
    return CSS_OK;
}

// -------------------------------------------------------------------------------------------------
// RCCL without Python (css_mi355.h "the exchanges of the sharded path").  librccl.so is loaded on first use: the entry
// points are looked up by name, so that this library loads (and everything else works) on a box without RCCL.
extern "C++" {
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, ncclUniqueId_bytes, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.why = std::string("librccl.so could not be loaded: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p && r.why.empty()) r.why = std::string("librccl.so lacks ") + n; return p; };
        r.GetUniqueId = (int (*)(void*))sym("ncclGetUniqueId");
        r.CommInitRank = (int (*)(void**, int, ncclUniqueId_bytes, int))sym("ncclCommInitRank");
        r.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
        r.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))sym("ncclAllGather");
        r.GetVersion = (int (*)(int*))sym("ncclGetVersion");
        r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    });
    return r;
}
int rccl_fail(css_ctx* h, const char* what, int code) {
    Rccl& r = rccl();
    return fail(h, CSS_ERR_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(code) : "RCCL error") + " (" + std::to_string(code) + ")");
}
}  // namespace
}  // extern "C++"

int css_comm_unique_id(void* id_out) {
    if (!id_out) return CSS_ERR_INVALID_ARG;
    Rccl& r = rccl();
    if (!r.why.empty()) return CSS_ERR_STATE;
    ncclUniqueId_bytes id{};
    if (r.GetUniqueId(&id) != 0) return CSS_ERR_HIP;
    std::memcpy(id_out, &id, CSS_COMM_ID_BYTES);
    return CSS_OK;
}

int css_comm_init(css_handle_t h, const void* id, int32_t nranks, int32_t rank) {
    CSS_DRAIN(h);
    if (!h || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(h, CSS_ERR_INVALID_ARG, "bad communicator arguments");
    if (h->comm) return fail(h, CSS_ERR_STATE, "the handle already has a communicator (css_comm_destroy first)");
    Rccl& r = rccl();
    if (!r.why.empty()) return fail(h, CSS_ERR_STATE, r.why);
    {   // the entry points are bound by hand (ncclUniqueId = 128 opaque bytes by value, ncclInt8 = 0): the ABI of NCCL / RCCL 2.x
        int v = 0;
        if (r.GetVersion(&v) != 0 || v < 20000 || v >= 30000)
            return fail(h, CSS_ERR_STATE, "librccl.so reports version code " + std::to_string(v) + ": the hand-bound ABI is that of RCCL 2.x");
    }
    HIPCHK(h, hipSetDevice(h->device));
    ncclUniqueId_bytes uid;
    std::memcpy(&uid, id, CSS_COMM_ID_BYTES);
    void* comm = nullptr;
    const int rc = r.CommInitRank(&comm, nranks, uid, rank);
    if (rc != 0) return rccl_fail(h, "ncclCommInitRank", rc);
    h->comm = comm; h->comm_ranks = nranks; h->comm_rank = rank;
    return CSS_OK;
}

int css_comm_destroy(css_handle_t h) {
    if (!h) return CSS_ERR_INVALID_ARG;
    if (!h->comm) return CSS_OK;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    const int rc = rccl().CommDestroy(h->comm);
    h->comm = nullptr; h->comm_ranks = 0; h->comm_rank = -1;
    return rc == 0 ? (int)CSS_OK : rccl_fail(h, "ncclCommDestroy", rc);
}

int css_comm_info(css_handle_t h, int32_t* nranks, int32_t* rank, int32_t* device, int32_t* rccl_version) {
    if (!h) return CSS_ERR_INVALID_ARG;
    if (!h->comm) return fail(h, CSS_ERR_STATE, "no communicator: css_comm_init first");
    if (nranks) *nranks = h->comm_ranks;
    if (rank) *rank = h->comm_rank;
    if (device) *device = h->device;
    if (rccl_version) { int v = 0; rccl().GetVersion(&v); *rccl_version = v; }
    return CSS_OK;
}

int css_comm_all_gather(css_handle_t h, const void* send_dev, void* recv_dev, int64_t bytes_per_rank) {
    if (!h || !send_dev || !recv_dev || bytes_per_rank < 1) return fail(h, CSS_ERR_INVALID_ARG, "bad all-gather arguments");
    if (!h->comm) return fail(h, CSS_ERR_STATE, "no communicator: css_comm_init first");
    HIPCHK(h, hipSetDevice(h->device));
    const int rc = rccl().AllGather(send_dev, recv_dev, (size_t)bytes_per_rank, /* ncclInt8 */ 0, h->comm, h->stream);
    return rc == 0 ? (int)CSS_OK : rccl_fail(h, "ncclAllGather", rc);
}


// libcss_mi355.so, host side 2 / 3: one stage per reference function (css_begin .. css_stage_*: css/css.py:110-338 piece by piece), the mask
// estimator's lanes and batches, and the separator protocol on caller data (conformer_wrapper.py:79-146).  (api_ctx.hpp: shared.)
#include "api_ctx.hpp"

// -------------------------------------------------------------------------------------------------
// Opens a session: validates the configuration, fixes the plan, sizes the workspace.  No sample moves here.
int check_run_args(css_handle_t h, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg, CssPlan* plan_out) {
    if (!h || !cfg) return fail(h, CSS_ERR_INVALID_ARG, "null argument");
    if (n_ch != h->d.num_mics)
        return fail(h, CSS_ERR_SHAPE, "input has " + std::to_string(n_ch) + " channels, the model expects " + std::to_string(h->d.num_mics));
    if (!cfg->w_first || !cfg->w_mid || !cfg->w_last) return fail(h, CSS_ERR_INVALID_ARG, "segment weights missing");
    if (cfg->mask_floor > 1.0f || cfg->mask_floor < 0.f) return fail(h, CSS_ERR_MASK_FLOOR, "mask_floor_db must be <= 0");
    const int T = cfg->segment_frames, hop = cfg->hop_frames;
    if (T < 2 || T > CSS_MAX_SEGMENT_FRAMES)
        return fail(h, CSS_ERR_INVALID_ARG, "segment_frames must be in [2, " + std::to_string(CSS_MAX_SEGMENT_FRAMES) + "]");
    if (cfg->stitching_loss < 0 || cfg->stitching_loss > 1 || cfg->stitching_input < 0 || cfg->stitching_input > 1)
        return fail(h, CSS_ERR_INVALID_ARG, "unexpected stitching_loss / stitching_input");
    if (hop <= 0 || hop >= T) return fail(h, CSS_ERR_INVALID_ARG, "hop_frames must satisfy 1 <= hop < T (at least one frame of overlap for the stitching cost, css.py:276)");
    CssPlan p{};
    if (plan_impl(h->d, *cfg, n_samples, &p) != CSS_OK) return fail(h, CSS_ERR_INVALID_ARG, "bad segment configuration");
    if (p.zero_weight) return fail(h, CSS_ERR_ZERO_WEIGHT, "zero weights found. check hop_size, segment_size or m0, m1");
    *plan_out = p;
    return CSS_OK;
}

int begin_impl(css_handle_t h, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg) {
    CssPlan p{};
    int rc0 = check_run_args(h, n_samples, n_ch, cfg, &p);
    if (rc0 != CSS_OK) return rc0;
    const int T = cfg->segment_frames;
    HIPCHK(h, hipSetDevice(h->device));
    h->cfg = *cfg;
    h->w_host.assign(3 * (size_t)T, 0.f);
    std::memcpy(h->w_host.data(), cfg->w_first, T * sizeof(float));
    std::memcpy(h->w_host.data() + T, cfg->w_mid, T * sizeof(float));
    std::memcpy(h->w_host.data() + 2 * T, cfg->w_last, T * sizeof(float));
    h->cfg.w_first = h->w_host.data();
    h->cfg.w_mid = h->w_host.data() + T;
    h->cfg.w_last = h->w_host.data() + 2 * T;
    h->plan = p;
    h->n_ch = n_ch;
    h->n_pad = (n_samples + (h->fft512 ? 0 : 64) + 31) / 32 * 32;   // whole 32-sample groups (+ slack the generic analysis product reads past the last frame)
    h->T_ld = (p.mix_frames + 3) / 4 * 4;
    h->stft_done = h->perms_done = h->have_override = false;
    h->has_session = true;
    h->uploads.clear();
    h->ev_pool_used = 0;
    h->tim = CssTimings{};
    h->prof_used = 0;
    h->prof_reduced = 0;
    h->gemm_flops = 0.0;
    h->pcm_src = nullptr;
    h->src16 = false;

    const int F = h->d.num_bins, S = h->d.num_spks;
    const int64_t nseg = p.num_segments, TL = p.mix_frames;
    int rc;
#define ENS(buf, bytes, ...)                                              \
    if ((rc = ensure(h, h->buf, (size_t)(bytes), ##__VA_ARGS__)) != CSS_OK) return rc;
    hipEventRecord(h->ev[0], h->stream);
    if (!h->piped_now) HIPCHK(h, hipMemsetAsync(h->peak_dev, 0, sizeof(unsigned int), h->stream));
    if (!h->queued) HIPCHK(h, hipMemsetAsync(h->range_flag_dev, 0, sizeof(unsigned int), h->stream));   // queued passes accumulate
    ENS(pcm_cm, (size_t)n_ch * h->n_pad * sizeof(float))
    ENS(X, (size_t)n_ch * X_ROWS_PER_BIN * F * h->T_ld * sizeof(float))
    h->ph_valid = false;
    if ((rc = ensure_activations(h, std::min<int64_t>(h->max_batch, nseg), T)) != CSS_OK) return rc;
    ENS(masks, (size_t)(S + 1) * F * nseg * T * sizeof(float))
    ENS(scm, (size_t)nseg * (S + 1) * F * 49 * sizeof(double))
    ENS(bfw, (size_t)nseg * S * F * 7 * 2 * sizeof(double))
    ENS(sep, (size_t)nseg * S * F * T * 2 * sizeof(float))
    ENS(costs, (size_t)std::max<int64_t>(nseg - 1, 1) * S * S * sizeof(double))
    ENS(pit_part, pit_cost_scratch_bytes(nseg - 1))
    ENS(perms, (size_t)nseg * S * sizeof(int32_t))
    ENS(mask_st, (size_t)S * F * TL * sizeof(float))
    ENS(activity, (size_t)S * TL * sizeof(float))
    ENS(act_b, (size_t)S * TL)
    ENS(act_tmp, (size_t)S * TL)
    ENS(act_final, (size_t)S * TL)
    ENS(Y, (size_t)S * TL * h->KIp * sizeof(float))
    ENS(G, (size_t)S * TL * h->d.frame_len * sizeof(float))
    ENS(wav, (size_t)S * p.n_out * sizeof(float))
    ENS(pnorm, (size_t)nseg * sizeof(double))
    ENS(segw, (size_t)3 * T * sizeof(float))
#undef ENS
    h->masks_v = (float*)h->masks.p;
    h->mask_ld_v = nseg * T;
    // (a copy from pageable memory makes the host wait for the stream: paid only when the windows change)
    if (h->w_on_device != h->w_host) {
        // (the tail of an overlapping queued pass may still read the previous windows)
        if (h->tail_pending && h->tail_end) HIPCHK(h, hipStreamWaitEvent(h->stream, h->tail_end, 0));
        HIPCHK(h, hipMemcpyAsync(h->segw.p, h->w_host.data(), 3 * (size_t)T * sizeof(float), hipMemcpyHostToDevice, h->stream));
        h->w_on_device = h->w_host;
    }
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

// samples [s_lo, s_hi) of a host recording -> their place in the device copy (pcm_in), on `st`
int upload_pcm(css_handle_t h, const float* pcm_host, int64_t s_lo, int64_t s_hi, hipStream_t st) {
    if (s_hi <= s_lo) return CSS_OK;
    const size_t row = (size_t)h->n_ch * sizeof(float);
    HIPCHK(h, hipMemcpyAsync((char*)const_cast<float*>(h->pcm_src) + (size_t)s_lo * row, (const char*)pcm_host + (size_t)s_lo * row,
                             (size_t)(s_hi - s_lo) * row, hipMemcpyHostToDevice, st));
    return CSS_OK;
}

// The level (power-of-two gain of the split synthesis operand) is the peak of the samples some FRAME reads,
// [0, (stft_frames - 1) * hop + frame_len): the up to hop - 1 trailing samples no frame covers are left out, so that
// the fused pass, the staged pass and every sharding of it (parallel.py uploads exactly the covered ranges) scan the
// same samples and agree bit for bit whatever the tail holds.
int64_t covered_end(const css_ctx* h) {
    const int64_t fr = h->plan.stft_frames;
    return fr > 0 ? std::min<int64_t>((fr - 1) * h->d.frame_hop + h->d.frame_len, h->plan.n_samples) : 0;
}
int64_t peak_len(const css_ctx* h, int64_t s_lo, int64_t s_hi) {   // samples of [s_lo, s_hi) to scan
    return std::max<int64_t>(std::min<int64_t>(s_hi, covered_end(h)) - s_lo, 0);
}

int css_begin(css_handle_t h, const float* pcm, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg, int pcm_is_device) {
    CSS_DRAIN(h);
    if (!h || !pcm) return fail(h, CSS_ERR_INVALID_ARG, "null argument");
    if (!pcm_is_device) return css_begin_range(h, pcm, n_samples, n_ch, cfg, 0, n_samples);
    const int rc = begin_impl(h, n_samples, n_ch, cfg);
    if (rc != CSS_OK) return rc;
    h->pcm_src = pcm;
    launch_pcm_peak_f32(pcm, peak_len(h, 0, n_samples) * n_ch, h->peak_dev, h->stream);
    hipEventRecord(h->ev[1], h->stream);
    return CSS_OK;
}

int css_begin_range(css_handle_t h, const float* pcm_host, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg,
                    int64_t s_lo, int64_t s_hi) {
    CSS_DRAIN(h);
    if (!h || !pcm_host) return fail(h, CSS_ERR_INVALID_ARG, "null argument");
    if (s_lo < 0 || s_hi > n_samples || s_lo > s_hi) return fail(h, CSS_ERR_INVALID_ARG, "sample range out of bounds");
    int rc = begin_impl(h, n_samples, n_ch, cfg);
    if (rc != CSS_OK) return rc;
    if ((rc = ensure(h, h->pcm_in, (size_t)n_samples * n_ch * sizeof(float))) != CSS_OK) return rc;
    h->pcm_src = (const float*)h->pcm_in.p;
    if ((rc = upload_pcm(h, pcm_host, s_lo, s_hi, h->stream)) != CSS_OK) return rc;
    launch_pcm_peak_f32(h->pcm_src + s_lo * n_ch, peak_len(h, s_lo, s_hi) * n_ch, h->peak_dev, h->stream);
    hipEventRecord(h->ev[1], h->stream);
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}


int css_upload_range(css_handle_t h, const float* pcm_host, int64_t s_lo, int64_t s_hi) {
    int rc = check_session(h);
    if (rc) return rc;
    if (!pcm_host) return fail(h, CSS_ERR_INVALID_ARG, "null argument");
    if (h->pcm_src != (const float*)h->pcm_in.p || !h->pcm_in.p) return fail(h, CSS_ERR_STATE, "the session was not opened by css_begin_range");
    if (s_lo < 0 || s_hi > h->plan.n_samples || s_lo > s_hi) return fail(h, CSS_ERR_INVALID_ARG, "sample range out of bounds");
    if (s_hi == s_lo) return CSS_OK;
    HIPCHK(h, hipSetDevice(h->device));
    // behind whatever the handle's stream had enqueued when the session began (the previous session's readers of pcm_in)
    hipEvent_t landed = pool_event(h);
    if (h->uploads.empty()) {
        hipEvent_t opened = pool_event(h);
        HIPCHK(h, hipEventRecord(opened, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->copy_stream, opened, 0));
    }
    if ((rc = upload_pcm(h, pcm_host, s_lo, s_hi, h->copy_stream)) != CSS_OK) return rc;
    launch_pcm_peak_f32(h->pcm_src + s_lo * h->n_ch, peak_len(h, s_lo, s_hi) * h->n_ch, h->peak_dev, h->copy_stream);
    HIPCHK(h, hipEventRecord(landed, h->copy_stream));
    h->uploads.push_back({s_lo, s_hi, landed});
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

// Analysis transform of frames [t_lo, t_hi) on stream `st`: channel-major copy of exactly the samples these frames read
// (from the sample-major float PCM, or straight from the session's PCM16 planes), then DFT-matrix x frames.
// The analysis transform stays on the exact float32 MFMA path in either mode: the 7x7 MVDR solve amplifies
// rounding noise of X by the condition number of the noise covariance (~200x on the test meetings), and
// split-f16 operands (22 significant bits) tripled that noise -- measured: waveform distance to the reference
// on identical decisions 4e-5 -> 1.1e-4.  The synthesis transform has no such amplifier and does use it.
int stft_frames(css_ctx* h, int64_t t_lo, int64_t t_hi, const int16_t* planes16, hipStream_t st) {
    const int F = h->d.num_bins, N = h->d.frame_len, hop = h->d.frame_hop;
    const int64_t f_hi = std::min<int64_t>(t_hi, h->plan.stft_frames);
    if (f_hi <= t_lo) return CSS_OK;
    const int64_t i_lo = t_lo * hop, i_hi = std::min<int64_t>((f_hi - 1) * hop + N, h->n_pad);
    {
        CSS_PROF(CSS_PROF_DEINTERLEAVE, st);
        if (planes16) launch_pcm16_to_channel_major(planes16, (float*)h->pcm_cm.p, h->plan.n_samples, h->n_ch, h->n_pad, i_lo, i_hi, st);
        else launch_deinterleave(h->pcm_src, (float*)h->pcm_cm.p, h->plan.n_samples, h->n_ch, h->n_pad, i_lo, i_hi, 0, st);
    }
    if (!h->fft512)   // the generic product reads up to 31 samples past a frame (times zero columns): they must be finite
        HIPCHK(h, hipMemset2DAsync((float*)h->pcm_cm.p + h->plan.n_samples, (size_t)h->n_pad * sizeof(float), 0,
                                   (size_t)(h->n_pad - h->plan.n_samples) * sizeof(float), (size_t)h->n_ch, st));
    CSS_PROF(CSS_PROF_STFT, st);
    bool ph = false;
    if (!analysis_transform(h, (const float*)h->pcm_cm.p, h->n_pad, h->n_ch, t_lo, f_hi, (float*)h->X.p, h->T_ld, st,
                            (float*)h->X.p + (int64_t)h->n_ch * 2 * F * h->T_ld, &ph))
        return fail(h, CSS_ERR_HIP, "the analysis transform's LDS could not be reserved");
    h->ph_valid = ph;
    return CSS_OK;
}

int css_stage_stft_range(css_handle_t h, int64_t t_lo, int64_t t_hi) {
    int rc = check_session(h);
    if (rc) return rc;
    if (t_lo < 0 || t_hi > h->plan.mix_frames || t_lo > t_hi) return fail(h, CSS_ERR_INVALID_ARG, "frame range out of bounds");
    if (!h->pcm_src) return fail(h, CSS_ERR_STATE, "the session holds no samples");
    HIPCHK(h, hipSetDevice(h->device));
    const int F = h->d.num_bins;
    if (h->plan.stft_frames < h->plan.mix_frames && !h->stft_done)  // short input: zero-padded frames (css.py:159-164)
        HIPCHK(h, hipMemsetAsync(h->X.p, 0, (size_t)h->n_ch * X_ROWS_PER_BIN * F * h->T_ld * sizeof(float), h->stream));
    {   // pieces of the recording still crossing PCIe (css_upload_range): wait for the ones these frames read
        const int64_t f_hi = std::min<int64_t>(t_hi, h->plan.stft_frames);
        const int64_t i_lo = t_lo * h->d.frame_hop, i_hi = f_hi > t_lo ? (f_hi - 1) * h->d.frame_hop + h->d.frame_len : i_lo;
        for (const auto& u : h->uploads)
            if (u.s_lo < i_hi && i_lo < u.s_hi) HIPCHK(h, hipStreamWaitEvent(h->stream, u.landed, 0));
    }
    if ((rc = stft_frames(h, t_lo, t_hi, nullptr, h->stream)) != CSS_OK) return rc;
    hipEventRecord(h->ev[2], h->stream);
    HIPCHK(h, hipGetLastError());
    h->stft_done = true;
    return CSS_OK;
}

int css_stage_stft(css_handle_t h) {
    int rc = check_session(h);
    if (rc) return rc;
    return css_stage_stft_range(h, 0, h->plan.mix_frames);
}


// The mask estimator over `nb` segments starting at `s0` on one lane (stream + activation set), phases [ph_lo, ph_hi):
// phase -1 = features + embed, phase l = Conformer block l, phase num_blocks = mask head.
int masknet_lane(css_ctx* h, const MaskIo& io, int64_t s0, int nb, int lane, int ph_lo, int ph_hi,
                        bool concurrent) {
    const CssModelDesc& d = h->d;
    const int T = io.T, D = d.attention_dim, FF = d.linear_units, F = d.num_bins;
    const int M = nb * T;
    hipStream_t st = lane ? h->lane_stream[lane] : h->stream;
    float* feat = (float*)(lane ? h->lfeat[lane].p : h->feat.p); float* x = (float*)(lane ? h->lhx[lane].p : h->hx.p);
    float* u = (float*)(lane ? h->lhu[lane].p : h->hu.p); float* t1 = (float*)(lane ? h->lht[lane].p : h->ht.p);
    float* qkf = (float*)(lane ? h->lqkf[lane].p : h->qkf.p);
    float* qkv = (float*)(lane ? h->lqkv[lane].p : h->qkv.p); float* cb = (float*)(lane ? h->lctx[lane].p : h->ctxb.p);
    const Weights& W = h->w;
    // Linear layers: split-f16 operands (h->split) -- every producer of a GEMM input writes the split format
    // directly (features, LayerNorm, the FFN's first GEMM, attention), the residual stream x stays float32.
    const int sp = h->split ? 1 : 0;
    auto WS = [&](const float* w) { return sp ? h->wsplit + (w - h->blob) : w; };
    // exact float32: the weights in fragment order for gemm_f32.hip (CSS_TUNE_F32_GEMM 0 and 2..5; 1 = round 4's kernel and
    // 6 = gemm_f32.hip with both operands through LDS read the row-major weights) -- same bits either way
    const int f32_tune = h->tune[CSS_TUNE_F32_GEMM];
    const bool frag = !sp && h->wfrag && f32_tune != 1 && f32_tune != 6 && (int64_t)M * std::max(h->Kp, FF) * 4 < ((int64_t)1 << 30);
    auto lin = [&](const float* A, int64_t lda, const float* Wt, const float* bias, float* C, int64_t ldc, int n, int k,
                   int act, int split_out) {
        GemmArgs g = linear(A, lda, frag ? h->wfrag + (Wt - h->blob) : WS(Wt), lda, bias, C, ldc, M, n, k, act);
        g.split_in = sp; g.split_out = sp ? split_out : 0; g.b_tiled = sp; g.concurrent = concurrent ? 1 : 0;
        g.b_frag32 = frag ? 1 : 0;
        g.B_rows = frag ? Wt : nullptr;   // (the row-major weight, should gemm_f32.hip decline the launch: launch_gemm)
        g.range_flag = sp ? h->range_flag_dev : nullptr;
        return g;
    };
    if (ph_lo < 0) {
        if (io.group) {
            for (const GroupSess& gs : *io.group) {
                const int64_t lo = std::max<int64_t>(s0, gs.off), hi = std::min<int64_t>(s0 + nb, gs.off + gs.n);
                if (hi <= lo) continue;
                CSS_PROF(CSS_PROF_FEATURES, st);
                launch_features(gs.X, gs.T_ld, gs.stft_frames, d.num_mics, F, feat + (lo - s0) * (int64_t)T * h->Kp, h->Kp, W.input_bias,
                                W.input_scale, lo - gs.off, (int)(hi - lo), T, io.hop, sp, h->feat_opts, st, gs.PH);
            }
        } else {
            CSS_PROF(CSS_PROF_FEATURES, st);
            launch_features(io.X, io.T_ld, io.stft_frames, d.num_mics, F, feat, h->Kp, W.input_bias, W.input_scale, s0, nb, T,
                            io.hop, sp, h->feat_opts, st, io.PH);
        }
        // embed: Linear -> LayerNorm -> ReLU (conformer.py:205-210)
        gemm(h, lin(feat, h->Kp, W.embed_w, W.embed_b, u, D, D, h->Kp, ACT_NONE, 0), st);
        { CSS_PROF(CSS_PROF_LAYERNORM, st); launch_layernorm(u, x, nullptr, W.embed_ln_w, W.embed_ln_b, M, D, 1, st); }
    }
    for (int l = std::max(ph_lo, 0); l < std::min(ph_hi, d.num_blocks); ++l) {
        const BlockWeights& b = W.blocks[l];
        const bool last = l + 1 == d.num_blocks;
        // x = xin + 0.5 * ff(xin)  (conformer.py:179,182).  with_ln = false: u already holds LN(xin), written by the fused
        // LayerNorm pair that closed the previous block.
        auto ffn = [&](bool with_ln, const float* xin, const float* lnw, const float* lnb, const float* w1, const float* b1,
                       const float* w2, const float* b2) {
            if (with_ln) { CSS_PROF(CSS_PROF_LAYERNORM, st); launch_layernorm(xin, sp ? nullptr : u, sp ? u : nullptr, lnw, lnb, M, D, 0, st); }
            gemm(h, lin(u, D, w1, b1, t1, FF, FF, D, ACT_RELU, FF), st);
            GemmArgs g = lin(t1, FF, w2, b2, x, D, D, FF, ACT_NONE, 0);
            g.residual = xin; g.ldr = D; g.alpha = 0.5f;
            gemm(h, g, st);
        };
        ffn(l == 0, x, b.ffi_ln_w, b.ffi_ln_b, b.ffi_w1, b.ffi_b1, b.ffi_w2, b.ffi_b2);
        // self attention (conformer.py:65-92)
        { CSS_PROF(CSS_PROF_LAYERNORM, st); launch_layernorm(x, sp ? nullptr : u, sp ? u : nullptr, b.att_ln_w, b.att_ln_b, M, D, 0, st); }
        // q, k and v leave the QKV GEMM as split operands for the MFMAs of the attention kernel (q and k in fragment order);
        // segments beyond 512 frames: as plain float32 rows for the any-length kernel (encoder.hip relpos_attn_long_kernel)
        const bool long_seg = T > 512 || css_force_long_path();
        {
            GemmArgs g = lin(u, D, b.wqkv, b.bqkv, qkv, 3 * D, 3 * D, D, ACT_NONE, long_seg ? 0 : 3 * D);
            if (sp && !long_seg) { g.frag_out = qkf; g.frag_D = D; g.frag_T = T; g.frag_heads = d.attention_heads; g.frag_invT = 1.0f / T; }
            gemm(h, g, st);
        }
        {
            CSS_PROF(CSS_PROF_ATTENTION, st);
            if (long_seg) {
                if (!launch_relpos_attention_long(qkv, W.pe_k, cb, nb, T, D, d.attention_heads, d.maxlen, sp, st)) return CSS_ERR_INVALID_ARG;
            } else {
                launch_relpos_attention(qkv, sp ? qkf : nullptr, (const float*)h->pe_frag[sp].p, cb, nb, T, D, d.attention_heads, d.maxlen, sp, sp, st);
            }
        }
        {
            GemmArgs g = lin(cb, D, b.wo, b.bo, x, D, D, D, ACT_NONE, 0);
            g.residual = x; g.ldr = D; g.alpha = 1.f;
            gemm(h, g, st);
        }
        // conv module (conformer.py:113-127): one kernel, x -> cb (the attention context buffer is free again; the
        // kernel must not write where neighbouring blocks still read), and the second feed-forward takes cb as its
        // input and residual and writes x.  Uncovered (D, taps): LayerNorm+GLU -> u, depthwise conv in place.
        // The LayerNorm of the second feed-forward (conformer.py:139) rides on the conv module's output pass.
        const float* xc = cb;
        bool ffo_ln = false;
        {
        CSS_PROF(CSS_PROF_CONV, st);
        if (!launch_conv_module(x, cb, b.conv_ln_w, b.conv_ln_b, b.pw, b.dw_wt, b.dw_b, b.bn_alpha, b.bn_beta, b.ffo_ln_w,
                                b.ffo_ln_b, sp ? nullptr : u, sp ? u : nullptr, nb, T, D, d.kernel_size, st)) {
            launch_ln_glu(x, u, b.conv_ln_w, b.conv_ln_b, b.pw, M, D, st);
            launch_dwconv(u, x, b.dw_wt, b.dw_b, b.bn_alpha, b.bn_beta, b.pw, nb, T, D, d.kernel_size, st);
            xc = x;
            ffo_ln = true;
        }
        }
        ffn(ffo_ln, xc, b.ffo_ln_w, b.ffo_ln_b, b.ffo_w1, b.ffo_b1, b.ffo_w2, b.ffo_b2);
        if (!last) {
            // conformer.py:184 and the next block's feed-forward LayerNorm (conformer.py:139) in one pass over x
            const BlockWeights& nb_ = W.blocks[l + 1];
            { CSS_PROF(CSS_PROF_LAYERNORM, st); launch_layernorm2(x, x, b.fin_ln_w, b.fin_ln_b, sp ? nullptr : u, sp ? u : nullptr, nb_.ffi_ln_w, nb_.ffi_ln_b, M, D, st); }
        } else {
            // conformer.py:184; the last block's output also feeds the mask head, as a split operand in u
            { CSS_PROF(CSS_PROF_LAYERNORM, st); launch_layernorm(x, x, sp ? u : nullptr, b.fin_ln_w, b.fin_ln_b, M, D, 0, st); }
        }
    }
    if (ph_hi <= d.num_blocks) return CSS_OK;
    // mask head (conformer.py:302-310), transposed so that time is the fastest axis of every mask:
    // masks[(k*F + f)][segment*T + t] = sigmoid(head_w[k*F + f] . x[token] + head_b[k*F + f])
    GemmArgs g{};
    const int nout = F * (d.num_spks + d.num_nois);
    if (sp) {
        // split mode: tokens x weights on the weights-direct kernel like every Linear layer, the result written transposed
        // (kernel trace, A/B on one box: 87 us per 60 segments against 101 us for the LDS-staged kernel with the weights as
        // its A operand; the step itself did not move measurably, 4.903 vs 4.907 ms)
        g.A = u; g.lda = D; g.strideA = 0;
        g.B = h->head_tiled; g.ldb = D; g.strideB = 0; g.b_tiled = 1;
        g.C = io.masks + s0 * T; g.ldc = io.mask_ld; g.strideC = 0; g.c_transposed = 1;
        g.M = M; g.N = nout; g.K = D; g.batch = 1;
        g.bias = W.head_b; g.bias_along_m = 0; g.act = ACT_SIGMOID; g.residual = nullptr; g.alpha = 1.f;
        g.split_in = 1; g.concurrent = concurrent ? 1 : 0;
        g.range_flag = h->range_flag_dev;
    } else {
        g.A = WS(W.head_w); g.lda = D; g.strideA = 0;
        g.B = sp ? u : x; g.ldb = D; g.strideB = 0;
        g.C = io.masks + s0 * T; g.ldc = io.mask_ld; g.strideC = 0;
        g.M = nout; g.N = M; g.K = D; g.batch = 1;
        g.bias = W.head_b; g.bias_along_m = 1; g.act = ACT_SIGMOID; g.residual = nullptr; g.alpha = 1.f;
        g.split_in = sp;
        g.range_flag = sp ? h->range_flag_dev : nullptr;
        // (the tokens are the large operand here: 17 row tiles of weights against hundreds of token panels; walked row panel by
        // row panel every XCD streamed all tokens twice -- 376 MB fetched per launch of 120 segments against 48 MB of operands;
        // kernel trace, A/B on one box: 107.8 -> 101.0 us at 60 segments per lane, 161.7 -> 137.6 us at 120)
        g.m_fastest = 1;
    }
    gemm(h, g, st);
    if (!lane) h->last_batch_tokens = M;
    return CSS_OK;
}

LaneSplit lane_split(const css_ctx* h, int nb, int T) {
    int nl = h->lanes;
    // Exact float32: the pass is its matrix products, and those want rows per launch more than they want a second chain to
    // fill their gaps -- a lane is worth it from ~14 000 token rows (75 segments of 3 s) per lane, and never a third.  One box,
    // A/B: a 60 s meeting (40 segments) 11.05 ms on one lane, 11.8 on two, 12.4 on three; a shared batch of 128 segments
    // 8.93 ms per session on one lane, 9.06 on two; of 256 segments 8.93 on one, 8.81 on two, 9.07 on three.
    const int lane_rows = h->tune[CSS_TUNE_F32_LANE_ROWS];
    if (!h->split && lane_rows > 1) nl = std::min(nl, std::min(2, std::max(1, (int)((int64_t)nb * T / lane_rows))));
    if (nl < 2 || h->profile_gemm || nb < 4 * nl) return {1, nb};   // the per-launch profile needs one ordered stream
    return {nl, (nb + nl - 1) / nl};
}
// A recording's segments [seg_lo, seg_hi) in batches of at most `cap`, equally long (9 x 128 + 57 becomes 10 x 121).
int64_t batch_len(int64_t n, int64_t cap) {
    const int64_t nbat = (n + cap - 1) / cap;
    return nbat ? (n + nbat - 1) / nbat : 0;
}

// One batched pass of the mask estimator over `nb` segments starting at `s0`: `lanes` part batches on as many streams
// (see css_ctx::lanes).  prep(first segment, count, stream), when given, is enqueued at the head of each lane's chain:
// the fused path puts the analysis transform of the frames that lane is the first to read there (run_impl).
int masknet_batch(css_ctx* h, const MaskIo& io, int64_t s0, int nb, const LanePrep& prep, const LanePost& post,
                         hipEvent_t before_head) {
    const int L = h->d.num_blocks;
    const int sp = h->split ? 1 : 0;
    int rc;
    const bool long_seg = io.T > 512 || css_force_long_path();   // (the any-length attention reads the position table itself)
    if (!long_seg && h->pe_frag_T[sp] != io.T) {   // the attention kernel's position operands depend on the segment length only
        if ((rc = ensure(h, h->pe_frag[sp], (size_t)pe_fragment_tiles(io.T) * 2048 * sizeof(float))) != CSS_OK) return rc;
        launch_pe_fragments(sp ? h->wsplit + (h->w.pe_k - h->blob) : h->w.pe_k, (float*)h->pe_frag[sp].p, io.T, h->d.maxlen,
                            sp, h->stream);
        h->pe_frag_T[sp] = io.T;
    }
    const LaneSplit ls = lane_split(h, nb, io.T);
    if (ls.nl == 1) {
        if ((rc = prep(s0, nb, h->stream)) != CSS_OK) return rc;
        if (before_head) {   // the mask head and what follows write buffers an earlier pass's tail may still read
            if ((rc = masknet_lane(h, io, s0, nb, 0, -1, L)) != CSS_OK) return rc;
            HIPCHK(h, hipStreamWaitEvent(h->stream, before_head, 0));
            if ((rc = masknet_lane(h, io, s0, nb, 0, L, L + 1)) != CSS_OK) return rc;
        } else if ((rc = masknet_lane(h, io, s0, nb, 0, -1, L + 1)) != CSS_OK) {
            return rc;
        }
        return post(s0, nb, h->stream);
    }
    HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));    // everything the estimator reads is ordered before this
    for (int l = 1; l < ls.nl; ++l) HIPCHK(h, hipStreamWaitEvent(h->lane_stream[l], h->ev_fork, 0));
    for (int l = 0; l < ls.nl; ++l) {
        const int lo = l * ls.per, n = std::min(ls.per, nb - lo);
        if (n > 0 && (rc = prep(s0 + lo, n, l ? h->lane_stream[l] : h->stream)) != CSS_OK) return rc;
    }
    // the chains are enqueued phase by phase, in turn, so that no stream starts far behind the others
    for (int ph = -1; ph <= L; ++ph)
        for (int l = 0; l < ls.nl; ++l) {
            const int lo = l * ls.per, n = std::min(ls.per, nb - lo);
            if (n > 0 && ph == L && before_head) HIPCHK(h, hipStreamWaitEvent(l ? h->lane_stream[l] : h->stream, before_head, 0));
            if (n > 0 && (rc = masknet_lane(h, io, s0 + lo, n, l, ph, ph + 1, true)) != CSS_OK) return rc;
        }
    for (int l = 0; l < ls.nl; ++l) {
        const int lo = l * ls.per, n = std::min(ls.per, nb - lo);
        if (n > 0 && (rc = post(s0 + lo, n, l ? h->lane_stream[l] : h->stream)) != CSS_OK) return rc;
    }
    for (int l = 1; l < ls.nl; ++l) {
        HIPCHK(h, hipEventRecord(h->ev_join[l], h->lane_stream[l]));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join[l], 0));
    }
    return CSS_OK;
}
int masknet_batch(css_ctx* h, const MaskIo& io, int64_t s0, int nb) {
    const LanePrep none = [](int64_t, int, hipStream_t) { return (int)CSS_OK; };
    return masknet_batch(h, io, s0, nb, none, none);
}

int css_stage_masknet(css_handle_t h, int64_t seg_lo, int64_t seg_hi) {
    int rc = check_session(h);
    if (rc) return rc;
    if (!h->stft_done) return fail(h, CSS_ERR_STATE, "css_stage_stft must run before css_stage_masknet");
    if (seg_lo < 0 || seg_hi > h->plan.num_segments || seg_lo > seg_hi) return fail(h, CSS_ERR_INVALID_ARG, "segment range out of bounds");
    HIPCHK(h, hipSetDevice(h->device));
    const int T = h->cfg.segment_frames;
    const int64_t cap = batch_len(seg_hi - seg_lo, std::min<int64_t>(batch_cap(h, T), h->plan.num_segments));
    MaskIo io{(const float*)h->X.p, h->T_ld, h->plan.stft_frames, h->cfg.hop_frames, T, h->masks_v, h->mask_ld_v};
    io.PH = h->ph_valid ? (const float*)h->X.p + (int64_t)h->n_ch * 2 * h->d.num_bins * h->T_ld : nullptr;
    for (int64_t s0 = seg_lo; s0 < seg_hi; s0 += cap) {
        const int nb = (int)std::min<int64_t>(cap, seg_hi - s0);
        if ((rc = masknet_batch(h, io, s0, nb)) != CSS_OK) return rc;
    }
    hipEventRecord(h->ev[3], h->stream);
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

// make_mvdr + mask floor / multiply (+ power normalisation) for segments [lo, hi) on `st`
int mvdr_on(css_ctx* h, int64_t seg_lo, int64_t seg_hi, hipStream_t st) {
    if (seg_hi <= seg_lo) return CSS_OK;
    MvdrArgs a = mvdr_args(h, seg_lo, (int)(seg_hi - seg_lo));
    if (a.use_mvdr) {
        {
            CSS_PROF(CSS_PROF_SCM, st);
            if (!launch_scm(a, st)) return fail(h, CSS_ERR_HIP, "the covariance kernel's LDS could not be reserved");
        }
        { CSS_PROF(CSS_PROF_MVDR_SOLVE, st); launch_mvdr_solve(a, st); }
    }
    CSS_PROF(CSS_PROF_BEAMFORM, st);
    launch_beamform(a, st);
    if (h->cfg.normalize_segment_power) launch_segment_power_norm(a, (double*)h->pnorm.p, st);
    return CSS_OK;
}

int css_stage_mvdr(css_handle_t h, int64_t seg_lo, int64_t seg_hi) {
    int rc = check_session(h);
    if (rc) return rc;
    if (seg_lo < 0 || seg_hi > h->plan.num_segments || seg_lo > seg_hi) return fail(h, CSS_ERR_INVALID_ARG, "segment range out of bounds");
    HIPCHK(h, hipSetDevice(h->device));
    if ((rc = mvdr_on(h, seg_lo, seg_hi, h->stream)) != CSS_OK) return rc;
    hipEventRecord(h->ev[4], h->stream);
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

void pit_costs_on(css_ctx* h, int64_t b_lo, int64_t b_hi, hipStream_t st) {
    if (b_hi <= b_lo) return;
    CSS_PROF(CSS_PROF_PIT, st);
    launch_pit_costs(stitch_args(h), h->cfg.stitching_loss, h->cfg.stitching_input, b_lo, b_hi, (double*)h->pit_part.p,
                     (double*)h->costs.p, st);
}

int css_stage_pit_costs(css_handle_t h, int64_t b_lo, int64_t b_hi) {
    int rc = check_session(h);
    if (rc) return rc;
    if (b_lo < 0 || b_hi > h->plan.num_segments - 1 || b_lo > b_hi) return fail(h, CSS_ERR_INVALID_ARG, "boundary range out of bounds");
    HIPCHK(h, hipSetDevice(h->device));
    pit_costs_on(h, b_lo, b_hi, h->stream);
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

// permutations of segments b_lo + 1 .. b_hi (b_lo == 0: also the identity of segment 0)
void pit_scan_on(css_ctx* h, int64_t b_lo, int64_t b_hi, hipStream_t st) {
    CSS_PROF(CSS_PROF_PIT, st);
    launch_pit_scan((const double*)h->costs.p, b_lo, b_hi, h->d.num_spks, (int32_t*)h->perms.p, st);
}

int css_stage_pit_scan(css_handle_t h) {
    int rc = check_session(h);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    pit_scan_on(h, 0, h->plan.num_segments - 1, h->stream);
    HIPCHK(h, hipGetLastError());
    h->perms_done = true;
    return CSS_OK;
}

int check_frames(css_ctx* h, int64_t t_lo, int64_t t_hi) {
    int rc = check_session(h);
    if (rc) return rc;
    if (t_lo < 0 || t_hi > h->plan.mix_frames || t_lo > t_hi) return fail(h, CSS_ERR_INVALID_ARG, "frame range out of bounds");
    HIPCHK(h, hipSetDevice(h->device));
    return CSS_OK;
}

int css_stage_stitch_masks(css_handle_t h, int64_t t_lo, int64_t t_hi) {
    int rc = check_frames(h, t_lo, t_hi);
    if (rc) return rc;
    if (!h->perms_done) return fail(h, CSS_ERR_STATE, "permutations missing: run css_stage_pit_scan or write CSS_BUF_PERMS");
    { CSS_PROF(CSS_PROF_OLA_MASKS, h->stream); launch_ola_masks(stitch_args(h), t_lo, t_hi, h->stream); }
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

int css_stage_stitch_gate(css_handle_t h, int64_t t_lo, int64_t t_hi) {
    int rc = check_frames(h, t_lo, t_hi);
    if (rc) return rc;
    if (!h->perms_done) return fail(h, CSS_ERR_STATE, "permutations missing: run css_stage_pit_scan or write CSS_BUF_PERMS");
    StitchArgs a = stitch_args(h);
    { CSS_PROF(CSS_PROF_GATE, h->stream); launch_morphology(a, t_lo, t_hi, h->stream); }
    { CSS_PROF(CSS_PROF_OLA_STFT, h->stream); launch_ola_stft(a, t_lo, t_hi, h->stream); }
    hipEventRecord(h->ev[5], h->stream);
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

int css_stage_stitch(css_handle_t h, int64_t t_lo, int64_t t_hi) {
    int rc = check_frames(h, t_lo, t_hi);
    if (rc) return rc;
    // the inverse transform of frame range [t_lo, t_hi) also needs frame t_lo - 1 (2-frame overlap-add),
    // and the dilate/erode gate needs activity `dilation + erosion` frames to either side
    const int64_t TL = h->plan.mix_frames;
    const int64_t y_lo = std::max<int64_t>(t_lo - (h->ovl - 1), 0);   // (ovl = ceil(frame_len / hop) frames over a sample: 2 as shipped)
    const int64_t halo = h->cfg.dilation_frames + h->cfg.erosion_frames;
    if ((rc = css_stage_stitch_masks(h, std::max<int64_t>(y_lo - halo, 0), std::min<int64_t>(t_hi + halo, TL))) != CSS_OK) return rc;
    return css_stage_stitch_gate(h, y_lo, t_hi);
}

// synthesis GEMM over frames [f_lo, f_hi) into G
void istft_gemm_on(css_ctx* h, int64_t f_lo, int64_t f_hi, hipStream_t st) {
    if (f_hi <= f_lo) return;
    const int S = h->d.num_spks, N = h->d.frame_len;
    const int64_t TL = h->plan.mix_frames;
    GemmArgs g{};
    g.split_in = h->split ? 1 : 0;   // Y rows were written as split operands by the stitch stage
    g.range_flag = h->split ? h->range_flag_dev : nullptr;
    g.A = (const float*)h->Y.p + f_lo * h->KIp; g.lda = h->KIp; g.strideA = TL * h->KIp;
    g.B = h->split ? h->dft_split : h->dft_inv_t; g.ldb = h->KIp; g.strideB = 0;
    g.C = (float*)h->G.p + f_lo * N; g.ldc = N; g.strideC = TL * N;
    g.M = (int)(f_hi - f_lo); g.N = N; g.K = h->KIp; g.batch = S;
    g.bias = nullptr; g.act = ACT_NONE; g.residual = nullptr; g.alpha = 1.f;
    // The whole meeting at once (a queued / grouped pass, css_stage_istft over everything): the S speakers' rows are one
    // contiguous [S TL][KIp] operand and the synthesis matrix is a static weight, so the launch takes the weights-direct
    // kernel like every Linear layer (tile-major matrix from css_create; the same k order, the same bits as the LDS-staged
    // kernel the frame ranges of the pipelined schedules use -- tests/test_hip_schedules.py holds the schedules together;
    // A/B on one box, interleaved: 35 vs 43 us per 60 s meeting)
    if (h->split && f_lo == 0 && f_hi == TL && N % 32 == 0 && (int64_t)S * TL < (int64_t)1 << 31) {
        g.M = (int)(S * TL); g.batch = 1; g.strideA = 0; g.strideC = 0;
        g.B = h->dft_tiled; g.b_tiled = 1;
    }
    CSS_PROF(CSS_PROF_ISTFT_GEMM, st);
    launch_gemm(g, st);
}
// overlap-add of output blocks [q_lo, q_hi) from the frames [f_lo, f_hi) of G; `out` may be mapped host memory
void wave_ola_on(css_ctx* h, int64_t f_lo, int64_t f_hi, int64_t q_lo, int64_t q_hi, float* out, int64_t out_ld,
                        int64_t out_q0, hipStream_t st) {
    if (f_hi <= f_lo) return;
    CSS_PROF(CSS_PROF_WAVE_OLA, st);
    launch_wave_ola((const float*)h->G.p, out, h->d.num_spks, h->plan.mix_frames, h->d.frame_hop, h->d.frame_len, q_lo, q_hi, f_lo, f_hi, out_ld,
                    out_q0, h->split ? h->peak_dev : nullptr, st);
}
int istft_impl(css_ctx* h, int64_t f_lo, int64_t f_hi, int64_t q_lo, int64_t q_hi, float* out, int64_t out_ld,
                      int64_t out_q0, hipStream_t st) {
    istft_gemm_on(h, f_lo, f_hi, st);
    wave_ola_on(h, f_lo, f_hi, q_lo, q_hi, out, out_ld, out_q0, st);
    hipEventRecord(h->ev[6], st);
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

int css_stage_istft(css_handle_t h, int64_t t_lo, int64_t t_hi) {
    int rc = check_frames(h, t_lo, t_hi);
    if (rc) return rc;
    const int64_t TL = h->plan.mix_frames;
    const int64_t q_hi = (t_hi == TL) ? TL - 1 + h->ovl : t_hi;  // the last range also writes the closing (half) frame(s)
    return istft_impl(h, std::max<int64_t>(t_lo - (h->ovl - 1), 0), t_hi, t_lo, q_hi, (float*)h->wav.p, h->plan.n_out, 0, h->stream);
}

int css_stage_istft_partial(css_handle_t h, int64_t t_lo, int64_t t_hi, float* shard_dev, int64_t shard_ld) {
    int rc = check_frames(h, t_lo, t_hi);
    if (rc) return rc;
    if (h->d.frame_len != 2 * h->d.frame_hop)
        return fail(h, CSS_ERR_INVALID_ARG, "partial output blocks compose only for frame_len = 2 * frame_hop (a two-term sum): other geometries exchange "
                                            "synthesis rows (css_stage_synthesis / css_stage_seam_rows / css_stage_overlap_add)");
    if (!shard_dev || shard_ld < (t_hi - t_lo + 1) * h->d.frame_hop) return fail(h, CSS_ERR_INVALID_ARG, "shard buffer too small");
    return istft_impl(h, t_lo, t_hi, t_lo, t_hi + 1, shard_dev, shard_ld, t_lo, h->stream);
}

// ---- the seam of a frame-sharded meeting for ANY frame geometry (round 6).  With frame_len = 2 hop an output block sums two
// frames and the two-term sum commutes, so ranks exchange partial BLOCKS (css_stage_istft_partial).  With ceil(frame_len / hop)
// = ovl > 2 frames over a sample the float sum is ordered (oldest frame first, wave_ola_kernel) and partial sums do not
// compose: the ranks exchange the synthesis ROWS of their last ovl - 1 frames instead (frame_len floats per frame and stream),
// the receiver puts them where its left neighbour's frames belong and runs the very overlap-add of the single-GPU pass.
int css_stage_synthesis(css_handle_t h, int64_t t_lo, int64_t t_hi) {
    int rc = check_frames(h, t_lo, t_hi);
    if (rc) return rc;
    istft_gemm_on(h, t_lo, t_hi, h->stream);
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

int css_stage_seam_rows(css_handle_t h, int64_t t_lo, int64_t t_hi, float* rows_dev, int32_t write) {
    int rc = check_frames(h, t_lo, t_hi);
    if (rc) return rc;
    if (!rows_dev) return fail(h, CSS_ERR_INVALID_ARG, "null argument");
    if (t_hi == t_lo) return CSS_OK;
    const int S = h->d.num_spks, L = h->d.frame_len;
    const int64_t TL = h->plan.mix_frames, nf = t_hi - t_lo;
    // G [S][TL][L]  <->  rows_dev [S][nf][L]
    float* g = (float*)h->G.p + t_lo * L;
    const size_t row = (size_t)nf * L * sizeof(float);
    if (write) HIPCHK(h, hipMemcpy2DAsync(g, (size_t)TL * L * sizeof(float), rows_dev, row, row, (size_t)S, hipMemcpyDeviceToDevice, h->stream));
    else HIPCHK(h, hipMemcpy2DAsync(rows_dev, row, g, (size_t)TL * L * sizeof(float), row, (size_t)S, hipMemcpyDeviceToDevice, h->stream));
    return CSS_OK;
}

int css_stage_overlap_add(css_handle_t h, int64_t f_lo, int64_t f_hi, int64_t q_lo, int64_t q_hi, float* out_dev, int64_t out_ld,
                          int64_t out_q0) {
    int rc = check_frames(h, f_lo, f_hi);
    if (rc) return rc;
    const int64_t TL = h->plan.mix_frames;
    if (!out_dev || q_lo < 0 || q_hi < q_lo || q_hi > TL - 1 + h->ovl || out_q0 > q_lo) return fail(h, CSS_ERR_INVALID_ARG, "output block range out of bounds");
    // (the last block may be a short one: n_out = (TL - 1) hop + frame_len; the kernel stops at out_ld)
    if (out_ld < std::min<int64_t>((q_hi - out_q0) * h->d.frame_hop, h->plan.n_out - out_q0 * h->d.frame_hop))
        return fail(h, CSS_ERR_INVALID_ARG, "out_ld shorter than the blocks asked for");
    wave_ola_on(h, f_lo, f_hi, q_lo, q_hi, out_dev, out_ld, out_q0, h->stream);
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

int css_stage_join_shards(css_handle_t h, const float* gathered_dev, int32_t world, int64_t shard_ld, const int64_t* t_lo,
                          const int64_t* t_hi, float* out_dev, int64_t out_ld) {
    int rc = check_session(h);
    if (rc) return rc;
    if (!gathered_dev || !t_lo || !t_hi || !out_dev || world < 1 || world > 64) return fail(h, CSS_ERR_INVALID_ARG, "bad argument (world <= 64)");
    if (out_ld < h->plan.n_out) return fail(h, CSS_ERR_INVALID_ARG, "out_ld shorter than the streams");
    const int hop = h->d.frame_hop;
    for (int k = 0; k < world; ++k)
        if (t_lo[k] < 0 || t_hi[k] < t_lo[k] || t_hi[k] > h->plan.mix_frames || (t_hi[k] - t_lo[k] + 1) * hop > shard_ld)
            return fail(h, CSS_ERR_INVALID_ARG, "rank frame range out of bounds / shard_ld too small");
    HIPCHK(h, hipSetDevice(h->device));
    launch_join_shards(gathered_dev, shard_ld, t_lo, t_hi, world, h->d.num_spks, hop, h->plan.n_out, out_dev, out_ld, h->stream);
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

int css_sync(css_handle_t h) {
    CSS_DRAIN(h);
    if (!h) return CSS_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CSS_OK;
}


// -------------------------------------------------------------------------------------------------
// separator-protocol helpers on caller data
int css_stft_host(css_handle_t h, const float* pcm, int64_t n_samples, int32_t n_ch, float* x_planes, int64_t t_frames) {
    CSS_DRAIN(h);
    if (!h || !pcm || !x_planes) return fail(h, CSS_ERR_INVALID_ARG, "null argument");
    if (n_ch < 1) return fail(h, CSS_ERR_SHAPE, "n_ch must be >= 1");
    const int F = h->d.num_bins, N = h->d.frame_len, hop = h->d.frame_hop;
    const int64_t T = n_samples < N ? 0 : (n_samples - N) / hop + 1;
    if (t_frames != T) return fail(h, CSS_ERR_SHAPE, "t_frames must be floor((n - frame_len)/hop) + 1 = " + std::to_string(T));
    if (T == 0) return CSS_OK;
    HIPCHK(h, hipSetDevice(h->device));
    const int64_t n_pad = (n_samples + (h->fft512 ? 0 : 64) + 3) / 4 * 4;
    const size_t in_b = (size_t)n_samples * n_ch * sizeof(float), cm_b = (size_t)n_pad * n_ch * sizeof(float);
    const size_t out_b = (size_t)n_ch * 2 * F * T * sizeof(float);
    int rc;
    if ((rc = ensure(h, h->stage, in_b + cm_b + out_b + 64)) != CSS_OK) return rc;
    float* in = (float*)h->stage.p;
    float* cm = in + ((size_t)n_samples * n_ch + 3) / 4 * 4;
    float* out = cm + (size_t)n_pad * n_ch;
    HIPCHK(h, hipMemcpyAsync(in, pcm, in_b, hipMemcpyHostToDevice, h->stream));
    if (!h->fft512) HIPCHK(h, hipMemsetAsync(cm, 0, cm_b, h->stream));
    launch_deinterleave(in, cm, n_samples, n_ch, n_pad, 0, n_pad, 0, h->stream);
    if (!analysis_transform(h, cm, n_pad, n_ch, 0, T, out, T, h->stream, nullptr, nullptr))
        return fail(h, CSS_ERR_HIP, "the analysis transform's LDS could not be reserved");
    HIPCHK(h, hipMemcpyAsync(x_planes, out, out_b, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CSS_OK;
}

int css_separate_host(css_handle_t h, const float* x_planes, int32_t batch, int32_t t_frames, float* masks) {
    CSS_DRAIN(h);
    if (!h || !x_planes || !masks || batch < 1) return fail(h, CSS_ERR_INVALID_ARG, "bad argument");
    if (t_frames < 2 || t_frames > CSS_MAX_SEGMENT_FRAMES)
        return fail(h, CSS_ERR_INVALID_ARG, "segment length must be in [2, " + std::to_string(CSS_MAX_SEGMENT_FRAMES) + "] frames");
    HIPCHK(h, hipSetDevice(h->device));
    const int F = h->d.num_bins, C = h->d.num_mics, T = t_frames, nm = h->d.num_spks + h->d.num_nois;
    const int64_t TT = (int64_t)batch * T;
    const size_t x_f = (size_t)C * 2 * F * TT, m_f = (size_t)nm * F * TT;
    int rc;
    if ((rc = ensure(h, h->stage, (x_f + m_f + 16) * sizeof(float))) != CSS_OK) return rc;
    float* X = (float*)h->stage.p;
    float* M = X + (x_f + 3) / 4 * 4;
    HIPCHK(h, hipMemcpyAsync(X, x_planes, x_f * sizeof(float), hipMemcpyHostToDevice, h->stream));
    const int64_t cap = std::min<int64_t>(h->max_batch, batch);
    if ((rc = ensure_activations(h, cap, T)) != CSS_OK) return rc;
    MaskIo io{X, TT, TT, T, T, M, TT};  // item b is the "segment" starting at frame b*T
    for (int64_t s0 = 0; s0 < batch; s0 += cap)
        if ((rc = masknet_batch(h, io, s0, (int)std::min<int64_t>(cap, batch - s0))) != CSS_OK) return rc;
    HIPCHK(h, hipMemcpyAsync(masks, M, m_f * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

// ConformerCssWrapper.forward (conformer_wrapper.py:58-77) for a batch of equally long clips, fused on the device:
// the training loop's validation forward.  pcm [batch][n_samples][n_ch] -> planes X [C][2F][batch * T']
// and masks M [(S+1) F][batch * T'] in the handle's staging buffer, T' = (n_samples - frame_len) / hop + 1, clip b in
// columns [b T', (b+1) T'); `extra_floats` more floats are reserved behind them (*extra).
static int forward_staged(css_handle_t h, const float* pcm, int32_t batch, int64_t n_samples, int32_t n_ch, size_t extra_floats,
                          float** Xo, float** Mo, float** extra, int* To) {
    if (!h || !pcm || batch < 1) return fail(h, CSS_ERR_INVALID_ARG, "bad argument");
    const int F = h->d.num_bins, C = h->d.num_mics, N = h->d.frame_len, hop = h->d.frame_hop;
    const int nm = h->d.num_spks + h->d.num_nois;
    if (n_ch != C) return fail(h, CSS_ERR_SHAPE, "the model expects " + std::to_string(C) + " channels");
    if (n_samples < N) return fail(h, CSS_ERR_INVALID_ARG, "clip shorter than one frame");
    const int64_t T64 = (n_samples - N) / hop + 1;
    if (T64 < 2 || T64 > CSS_MAX_SEGMENT_FRAMES)
        return fail(h, CSS_ERR_INVALID_ARG, "clip length must give 2.." + std::to_string(CSS_MAX_SEGMENT_FRAMES) + " frames");
    const int T = (int)T64;
    HIPCHK(h, hipSetDevice(h->device));
    const int64_t TT = (int64_t)batch * T;
    const int64_t n_pad = (n_samples + (h->fft512 ? 0 : 64) + 31) / 32 * 32;
    const size_t in_f = (size_t)batch * n_samples * C, cm_f = (size_t)batch * C * n_pad;
    const size_t x_f = (size_t)C * 2 * F * TT, ph_f = (size_t)C * F * TT, m_f = (size_t)nm * F * TT;
    int rc;
    if ((rc = ensure(h, h->stage, (in_f + cm_f + x_f + ph_f + m_f + extra_floats + 160) * sizeof(float))) != CSS_OK) return rc;
    float* in = (float*)h->stage.p;
    float* cm = in + (in_f + 15) / 16 * 16;
    float* X = cm + (cm_f + 15) / 16 * 16;
    float* PH = X + (x_f + 15) / 16 * 16;
    float* M = PH + (ph_f + 15) / 16 * 16;
    HIPCHK(h, hipMemcpyAsync(in, pcm, in_f * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (!h->fft512) HIPCHK(h, hipMemsetAsync(cm, 0, cm_f * sizeof(float), h->stream));
    bool ph = false;
    for (int b = 0; b < batch; ++b) {
        // analysis transform of clip b into columns [b T, (b+1) T) of the planes [C][2F][batch * T]
        launch_deinterleave(in + (size_t)b * n_samples * C, cm + (size_t)b * C * n_pad, n_samples, C, n_pad, 0, n_pad, 0, h->stream);
        if (!analysis_transform(h, cm + (size_t)b * C * n_pad, n_pad, C, 0, T, X + (int64_t)b * T, TT, h->stream, PH + (int64_t)b * T, &ph))
            return fail(h, CSS_ERR_HIP, "the analysis transform's LDS could not be reserved");
    }
    const int64_t cap = std::min<int64_t>(h->max_batch, batch);
    if ((rc = ensure_activations(h, cap, T)) != CSS_OK) return rc;
    MaskIo io{X, TT, TT, T, T, M, TT};  // clip b is the "segment" starting at frame b*T
    io.PH = ph ? PH : nullptr;
    for (int64_t s0 = 0; s0 < batch; s0 += cap)
        if ((rc = masknet_batch(h, io, s0, (int)std::min<int64_t>(cap, batch - s0))) != CSS_OK) return rc;
    *Xo = X; *Mo = M; *To = T;
    if (extra) *extra = M + (m_f + 15) / 16 * 16;
    return CSS_OK;
}

int css_forward_host(css_handle_t h, const float* pcm, int32_t batch, int64_t n_samples, int32_t n_ch, float* masks) {
    CSS_DRAIN(h);
    if (!h || !masks) return fail(h, CSS_ERR_INVALID_ARG, "bad argument");
    float *X, *M;
    int T;
    int rc = forward_staged(h, pcm, batch, n_samples, n_ch, 0, &X, &M, nullptr, &T);
    if (rc != CSS_OK) return rc;
    const size_t m_f = (size_t)(h->d.num_spks + h->d.num_nois) * h->d.num_bins * batch * T;
    HIPCHK(h, hipMemcpyAsync(masks, M, m_f * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

// css/training/train.py:411 _calc_loss for a validation batch (train.py:529 eval_model): forward, |STFT| of the mixture's
// and the ground truths' reference channel, S x S base-loss matrix per clip -> PIT (losses.py:32-48: the assignment of
// least mean loss, found as scipy's linear_sum_assignment finds it: lsap.hpp), noise loss, weighted mean.
int css_validation_loss_host(css_handle_t h, const float* mix, const float* gt_spk, const float* gt_noise, int32_t batch,
                             int64_t n_samples, int32_t n_ch, int32_t loss_name, int32_t base_loss, int32_t clip_gt,
                             float noise_weight, float* spk_loss, float* noise_loss, int32_t* perms, float* loss) {
    CSS_DRAIN(h);
    if (!h || !gt_spk || !gt_noise || !loss) return fail(h, CSS_ERR_INVALID_ARG, "null argument");
    if (loss_name < 0 || loss_name > 1 || base_loss < 0 || base_loss > 1) return fail(h, CSS_ERR_INVALID_ARG, "unknown loss_name / base_loss");
    const int F = h->d.num_bins, S = h->d.num_spks;
    if (S > 3 || h->d.num_nois != 1) return fail(h, CSS_ERR_INVALID_ARG, "at most three speaker outputs and one noise output");
    const int64_t n_pad = (n_samples + (h->fft512 ? 0 : 64) + 31) / 32 * 32;
    const int nsig = batch * (S + 1), chunks = val_loss_chunks(F);
    const int64_t T64 = n_samples >= h->d.frame_len ? (n_samples - h->d.frame_len) / h->d.frame_hop + 1 : 0;
    const size_t sig_f = (size_t)nsig * n_pad, g_f = (size_t)nsig * 2 * F * std::max<int64_t>(T64, 1), p_f = (size_t)batch * chunks * 16 * 2;
    float *X, *M, *extra;
    int T;
    int rc = forward_staged(h, mix, batch, n_samples, n_ch, sig_f + g_f + p_f + 64, &X, &M, &extra, &T);
    if (rc != CSS_OK) return rc;
    float* sig = extra;                                   // [batch][S + 1][n_pad]: the speakers, then the noise
    float* G = sig + (sig_f + 15) / 16 * 16;              // their planes [batch * (S + 1)][2F][T]
    double* partial = reinterpret_cast<double*>(G + (g_f + 15) / 16 * 16);
    HIPCHK(h, hipMemsetAsync(sig, 0, sig_f * sizeof(float), h->stream));
    for (int b = 0; b < batch; ++b) {
        HIPCHK(h, hipMemcpy2DAsync(sig + (size_t)b * (S + 1) * n_pad, (size_t)n_pad * sizeof(float), gt_spk + (size_t)b * S * n_samples,
                                   (size_t)n_samples * sizeof(float), (size_t)n_samples * sizeof(float), S, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(sig + ((size_t)b * (S + 1) + S) * n_pad, gt_noise + (size_t)b * n_samples,
                                 (size_t)n_samples * sizeof(float), hipMemcpyHostToDevice, h->stream));
    }
    if (!analysis_transform(h, sig, n_pad, nsig, 0, T, G, T, h->stream, nullptr, nullptr))
        return fail(h, CSS_ERR_HIP, "the analysis transform's LDS could not be reserved");
    launch_val_loss(X, M, G, batch, T, F, S, loss_name, base_loss, clip_gt ? 1 : 0, partial, h->stream);
    std::vector<double> part((size_t)batch * chunks * 16);
    HIPCHK(h, hipMemcpyAsync(part.data(), partial, part.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    const double inv = 1.0 / ((double)F * T);
    double total = 0.0;
    for (int b = 0; b < batch; ++b) {
        double mat[9] = {0}, noise = 0.0;
        for (int c = 0; c < chunks; ++c) {
            const double* q = part.data() + ((size_t)b * chunks + c) * 16;
            for (int i = 0; i < 9; ++i) mat[i] += q[i];
            noise += q[9];
        }
        // the assignment of least mean loss (losses.py:43, linear_sum_assignment): perm[a] = ground truth assigned to prediction a
        V4<V4<double>> cm;
        for (int a = 0; a < SMAX; ++a) {
            V4<double> row;
            row.fill(0.0);
            for (int k = 0; k < S; ++k)
                if (a < S) row.set(k, mat[a * 3 + k]);
            cm.set(a, row);
        }
        V4<int> assigned;
        lsap_small(cm, S, assigned);
        int best_p[SMAX];
        for (int a = 0; a < SMAX; ++a) best_p[a] = assigned.get(a);
        double best = 0.0;
        for (int a = 0; a < S; ++a) best += mat[a * 3 + best_p[a]];
        const double sl = best * inv / S, nl = noise * inv;
        if (spk_loss) spk_loss[b] = (float)sl;
        if (noise_loss) noise_loss[b] = (float)nl;
        if (perms) for (int a = 0; a < S; ++a) perms[b * S + a] = best_p[a];
        total += sl + (double)noise_weight * nl;
    }
    *loss = (float)(total / batch);
    return CSS_OK;
}

// the frames of stream `stream` that the activity gate kept (css.py:303-312) -> sample regions (the time
// map back) -> their concatenation -> Whisper's log-mel features, all from device-resident samples.
int css_handoff_logmel(css_handle_t h, const float* wav_dev, int64_t wav_ld, int32_t stream, int32_t n_mels, int32_t pad_frames,
                       int32_t drop_silence, float* mel_host, int64_t mel_capacity_frames, int64_t* n_mel_frames,
                       int64_t* regions_host, int32_t max_regions, int32_t* n_regions) {
    CSS_DRAIN(h);
    int rc = check_session(h);
    if (rc) return rc;
    if (!wav_dev || !mel_host || !n_mel_frames || !regions_host || !n_regions || max_regions < 1)
        return fail(h, CSS_ERR_INVALID_ARG, "null argument");
    if (stream < 0 || stream >= h->d.num_spks || (n_mels != 80 && n_mels != 128) || pad_frames < 0)
        return fail(h, CSS_ERR_INVALID_ARG, "stream out of range, or n_mels not 80 / 128");
    if (!h->perms_done) return fail(h, CSS_ERR_STATE, "no finished pass in this session");
    HIPCHK(h, hipSetDevice(h->device));
    const int64_t TL = h->plan.mix_frames, n_out = h->plan.n_out;
    const int hop = h->d.frame_hop, N = h->d.frame_len;
    if (wav_ld < n_out) return fail(h, CSS_ERR_INVALID_ARG, "wav_ld shorter than the streams");
    // ---- regions: maximal runs of active frames, widened by pad_frames, merged; frame t spans samples [t hop, t hop + N)
    std::vector<uint8_t> act((size_t)TL);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(act.data(), (const uint8_t*)h->act_final.p + (size_t)stream * TL, (size_t)TL, hipMemcpyDeviceToHost));
    std::vector<int64_t> reg;
    if (!drop_silence) {
        reg = {0, n_out};
    } else {
        for (int64_t t = 0; t < TL;) {
            if (!act[(size_t)t]) { ++t; continue; }
            int64_t e = t;
            while (e < TL && act[(size_t)e]) ++e;
            const int64_t a = std::max<int64_t>(t - pad_frames, 0) * hop, b = std::min<int64_t>((e - 1 + pad_frames) * hop + N, n_out);
            if (!reg.empty() && a <= reg.back()) reg.back() = std::max(reg.back(), b);
            else { reg.push_back(a); reg.push_back(b); }
            t = e;
        }
    }
    const int nr = (int)(reg.size() / 2);
    *n_regions = nr;
    if (nr > max_regions) return fail(h, CSS_ERR_INVALID_ARG, "more regions than max_regions: " + std::to_string(nr));
    std::vector<int64_t> offs((size_t)std::max(nr, 1), 0);
    int64_t n_act = 0;
    for (int r = 0; r < nr; ++r) { offs[(size_t)r] = n_act; n_act += reg[2 * r + 1] - reg[2 * r]; regions_host[2 * r] = reg[2 * r]; regions_host[2 * r + 1] = reg[2 * r + 1]; }
    const int64_t nfr = n_act / 160;                        // whisper: 1 + n // hop frames, the last one dropped
    *n_mel_frames = nfr;
    if (nfr == 0) return CSS_OK;
    if (nfr > mel_capacity_frames) return fail(h, CSS_ERR_INVALID_ARG, "mel buffer too small: need " + std::to_string(nfr) + " frames");
    // ---- tables (once per filterbank size)
    const size_t dft_f = (size_t)402 * 416, melw_f = (size_t)128 * 201;
    if ((rc = ensure(h, h->mel_tab, (dft_f + melw_f) * sizeof(float))) != CSS_OK) return rc;
    float* dftm = (float*)h->mel_tab.p;
    float* melw = dftm + dft_f;
    if (h->mel_bands != n_mels) {
        std::vector<float> t(dft_f + melw_f, 0.f);
        handoff_build_dft(t.data());
        handoff_build_mel(t.data() + dft_f, n_mels);
        HIPCHK(h, hipMemcpy(dftm, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
        h->mel_bands = n_mels;
    }
    // ---- work: region table | gathered samples | spectra [402][ld] | mel [n_mels][nfr] | max
    const int64_t ld = (nfr + 3) / 4 * 4, total = n_act + 400 + 416;
    const size_t tab_b = (size_t)nr * 3 * sizeof(int64_t) + 64;
    const size_t need = tab_b + ((size_t)total + (size_t)402 * ld + (size_t)n_mels * nfr + 64) * sizeof(float);
    if ((rc = ensure(h, h->mel_work, need)) != CSS_OK) return rc;
    int64_t* regs_d = (int64_t*)h->mel_work.p;
    int64_t* offs_d = regs_d + 2 * nr;
    float* gath = (float*)((char*)h->mel_work.p + (tab_b + 63) / 64 * 64);
    float* spec = gath + (total + 15) / 16 * 16;
    float* mel = spec + (size_t)402 * ld;
    int* gmax = (int*)(mel + (size_t)n_mels * nfr + 8);
    HIPCHK(h, hipMemcpyAsync(regs_d, reg.data(), (size_t)nr * 2 * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(offs_d, offs.data(), (size_t)nr * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    launch_handoff_gather(wav_dev + (size_t)stream * wav_ld, regs_d, offs_d, nr, n_act, gath, total, h->stream);
    GemmArgs g{};
    g.A = dftm; g.lda = 416; g.B = gath; g.ldb = 160; g.C = spec; g.ldc = ld;
    g.M = 402; g.N = (int)nfr; g.K = 416; g.batch = 1; g.alpha = 1.f;
    launch_gemm(g, h->stream);                              // exact float32 matrix cores: 402 x 416 per frame
    launch_handoff_mel(spec, ld, nfr, melw, n_mels, mel, gmax, h->stream);
    HIPCHK(h, hipMemcpyAsync(mel_host, mel, (size_t)n_mels * nfr * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return CSS_OK;
}

int css_istft_host(css_handle_t h, const float* y_planes, int32_t batch, int64_t t_frames, float* wav) {
    CSS_DRAIN(h);
    if (!h || !y_planes || !wav || batch < 1 || t_frames < 1) return fail(h, CSS_ERR_INVALID_ARG, "bad argument");
    const int F = h->d.num_bins, N = h->d.frame_len, hop = h->d.frame_hop, KI = h->KIp;
    HIPCHK(h, hipSetDevice(h->device));
    const int64_t n_out = (t_frames - 1) * hop + N;
    const size_t in_f = (size_t)batch * 2 * F * t_frames, rows_f = (size_t)batch * t_frames * KI;
    const size_t g_f = (size_t)batch * t_frames * N, w_f = (size_t)batch * n_out;
    int rc;
    if ((rc = ensure(h, h->stage, (in_f + rows_f + g_f + w_f + 16) * sizeof(float))) != CSS_OK) return rc;
    float* in = (float*)h->stage.p;
    float* rows = in + (in_f + 3) / 4 * 4;
    float* G = rows + rows_f;
    float* wv = G + g_f;
    HIPCHK(h, hipMemcpyAsync(in, y_planes, in_f * sizeof(float), hipMemcpyHostToDevice, h->stream));
    launch_planes_to_rows(in, rows, batch, 2 * F, t_frames, KI, h->stream);
    GemmArgs g{};
    g.A = rows; g.lda = KI; g.strideA = t_frames * KI;
    g.B = h->dft_inv_t; g.ldb = KI; g.strideB = 0;
    g.C = G; g.ldc = N; g.strideC = t_frames * N;
    g.M = (int)t_frames; g.N = N; g.K = KI; g.batch = batch;
    g.alpha = 1.f;
    launch_gemm(g, h->stream);
    launch_wave_ola(G, wv, batch, t_frames, hop, N, 0, t_frames - 1 + h->ovl, 0, t_frames, n_out, 0, nullptr, h->stream);
    HIPCHK(h, hipMemcpyAsync(wav, wv, w_f * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CSS_OK;
}


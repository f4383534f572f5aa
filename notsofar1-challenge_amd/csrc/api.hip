// C ABI of libcss_mi355.so (include/css_mi355.h): handle, weights, session state, stage drivers.
#include "../../include/css_mi355.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <functional>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <mutex>

#include "kernels.hpp"
#include "lsap.hpp"

using namespace css;

// kernel families of the per-launch profile (css_set_profile / css_get_kernel_stats), in the order of the pass
enum ProfCat : int {
    CSS_PROF_DEINTERLEAVE = 0, CSS_PROF_STFT, CSS_PROF_FEATURES, CSS_PROF_LINEAR, CSS_PROF_LAYERNORM, CSS_PROF_ATTENTION,
    CSS_PROF_CONV, CSS_PROF_SCM, CSS_PROF_MVDR_SOLVE, CSS_PROF_BEAMFORM, CSS_PROF_PIT, CSS_PROF_OLA_MASKS, CSS_PROF_GATE,
    CSS_PROF_OLA_STFT, CSS_PROF_ISTFT_GEMM, CSS_PROF_WAVE_OLA, CSS_PROF_ENCODE, CSS_PROF_COUNT
};
static const char* const kProfNames[CSS_PROF_COUNT] = {
    "deinterleave", "stft", "features", "linear_gemm", "layernorm", "attention", "conv_module", "scm", "mvdr_solve",
    "beamform", "pit", "ola_masks", "gate", "ola_stft", "istft_gemm", "wave_ola", "encode_pcm16"};

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct BlockWeights {
    const float *ffi_ln_w, *ffi_ln_b, *ffi_w1, *ffi_b1, *ffi_w2, *ffi_b2;
    const float *att_ln_w, *att_ln_b, *wqkv, *bqkv, *wo, *bo;
    const float *conv_ln_w, *conv_ln_b, *pw, *dw_wt, *dw_b, *bn_alpha, *bn_beta;
    const float *ffo_ln_w, *ffo_ln_b, *ffo_w1, *ffo_b1, *ffo_w2, *ffo_b2;
    const float *fin_ln_w, *fin_ln_b;
};

struct Weights {
    const float *input_bias, *input_scale, *embed_w, *embed_b, *embed_ln_w, *embed_ln_b, *pe_k;
    std::vector<BlockWeights> blocks;
    const float *head_w, *head_b;
};

inline int64_t pad16(int64_t n) { return (n + 15) / 16 * 16; }
// X holds the planes [C][2F][T_ld] (Re | Im) and, behind them, the phase planes [C][F][T_ld] the analysis transform writes
// with them (kernels.hpp launch_stft_fft): one allocation, so that whatever swaps or re-sizes X takes the phases along
constexpr int X_ROWS_PER_BIN = 3;
struct ncclUniqueId_bytes { char internal[CSS_COMM_ID_BYTES]; };   // ncclUniqueId (rccl.h: 128 opaque bytes, passed by value)
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

// What belongs to ONE session (recording) on a handle: its plan, its configuration and the device buffers the stages
// after the mask estimator work on.  The handle IS a SessState (the session the stage entry points see); a group of
// queued sessions that shares one estimator batch (run_group) parks the others in css_ctx::slots and swaps them in one
// at a time, so every stage helper keeps addressing `h->X`, `h->plan` ... unchanged.
struct SessState {
    bool has_session = false;
    CssRunCfg cfg{};
    CssPlan plan{};
    int n_ch = 0;
    int64_t n_pad = 0, T_ld = 0;
    bool stft_done = false, perms_done = false, have_override = false;
    bool ph_valid = false;       // the phase planes behind X belong to X (false after css_write_buffer(CSS_BUF_X): the feature kernel forms them itself)
    std::vector<float> w_host;   // segment weights of the session (cfg.w_* point into it)
    DevBuf pcm_cm, X, scm, bfw, sep, costs, perms, mask_st, activity, act_b, act_tmp, act_final, Y, G, wav, wta, pnorm, pit_part;
    DevBuf X_alt;   // run_group: consecutive grouped passes alternate between X and X_alt (the beamformer of pass P reads its
                    // planes on the tail stream while pass P + 1's transform already writes the other set)
    const float* pcm_src = nullptr;       // sample-major PCM on the device for the current session
    bool src16 = false;                   // run_group: pcm_src holds the session's n_ch mono PCM16 planes [C][n] instead (css_run_enqueue_pcm16)
    unsigned int* peak_dev = nullptr;     // max |sample| of the session's PCM as float bits (split_f16.hpp level_gain)
    // the session's masks [(S+1)F][mask_ld], segment s at column s*T: the handle's mask buffer, or -- inside a group -- this
    // session's columns of the group's buffer (the mask head of the shared estimator batch writes all of them at once)
    float* masks_v = nullptr;
    int64_t mask_ld_v = 0;
};

struct css_ctx : SessState {
    CssModelDesc d{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int max_batch = 64;
    int Kp = 0, KIp = 0;
    float* blob = nullptr;
    Weights w;
    // Linear-layer arithmetic.  Default (round 6): float32 operands on the float32 matrix instruction (gemm_f32.hip) -- the
    // reference's own operand precision (conformer.py:137-150 runs torch.nn.Linear in float32).  Opt-in, after
    // css_set_linear_mode(h, CSS_LINEAR_SPLIT_F16): split-f16 operands on the f16 matrix cores (22-bit operands, gemm_split*.hip).
    bool split = false;
    bool split_ok = true;        // false: a weight lies outside the split-f16 operand range, CSS_LINEAR_SPLIT_F16 is refused
    float* wsplit = nullptr;     // split-f16 images of the Linear weights, at the blob's own offsets
    float* dft_split = nullptr;  // split-f16 image of dft_inv_t (row-major)
    float* wfrag = nullptr;      // exact float32 mode: the Linear weights in gemm_f32.hip's fragment order (same offsets as blob)
    float* dft_tiled = nullptr;  // ... and in the tile-major layout of the weights-direct GEMM (whole-meeting synthesis)
    float* head_tiled = nullptr; // the mask head's weights in that layout (rows rounded up to 32; wsplit keeps the row-major image)
    DevBuf pe_frag[2];           // relative-position rows in attention-operand order for segment length pe_frag_T
    int pe_frag_T[2] = {0, 0};   // ([0] from the float32 table, [1] from the split-f16 one; encoder.hip pe_fragments_kernel)
    float* stft_tab = nullptr;   // window and twiddles of the analysis FFT (stft.hip)
    bool fft512 = true;          // frame_len 512 / hop 256 / 257 bins: the FFT kernel and the pipelined schedules; else the generic forms
    float* dft_fwd = nullptr;    // generic analysis: [2F][Lp] = (cos | -sin)(2 pi f n / N) * window[n], n < frame_len (zero beyond)
    int Lp = 0, ovl = 2;         // frame_len rounded up to 32; frames over an output sample = ceil(frame_len / hop)
    float* dft_inv_t = nullptr;  // [frame_len][KIp]

    // shared by the sessions of a handle: upload staging, the estimator's activations, the mask buffer, small tables
    std::vector<float> w_on_device;   // what segw holds (uploaded when a session's windows differ)
    DevBuf pcm_in, feat, hx, hu, ht, qkv, qkf, ctxb, masks, segw, stage, in16, pcm_f, enc, level, mel_tab, mel_work;
    // sessions of a queued group other than the active one (run_group)
    static constexpr int MAX_GROUP = 8;
    std::vector<SessState> slots;
    int group_limit = MAX_GROUP;      // css_set_queue_group: sessions merged into one estimator batch (1: none)
    // Second lane of the mask estimator: segments are independent through the whole network, so a batch is cut in `lanes`
    // parts that run as independent chains of kernels on as many streams.  One chain alone leaves the GPU idle in every
    // launch's prologue and epilogue (its waves are parked 51 % of the time, profiles/); two or three chains drift out of
    // phase and fill each other's bubbles (measured: 6.9 -> 6.4 ms per 60 s meeting with two).  Results do not change: every
    // kernel is batch invariant.  css_set_lanes(h, 1) turns it off.
    // (css_set_lanes: 1..4, default 3; lane 0 is `stream` with the buffers above)
    static constexpr int MAX_LANES = 4;
    int lanes = 3;
    hipStream_t lane_stream[MAX_LANES] = {};   // [0] unused
    hipEvent_t ev_fork = nullptr, ev_join[MAX_LANES] = {};
    DevBuf lfeat[MAX_LANES], lhx[MAX_LANES], lhu[MAX_LANES], lht[MAX_LANES], lqkv[MAX_LANES], lqkf[MAX_LANES], lctx[MAX_LANES];   // [0] unused
    int64_t last_batch_tokens = 0;
    // PCIe pieces of css_run* travel on their own stream, beside the kernels: the upload of the samples a lane's segments
    // read is followed by that lane's analysis transform and mask-estimator chain while the next piece is in flight, and
    // finished ranges of the output leave while the last ranges are still being synthesised.
    // range check of the split-f16 operand format (split_f16.hpp): a device word set when the stitched activity or the
    // waveforms hold a non-finite value, mirrored into page-locked host memory at the end of every pass
    int mel_bands = 0;                    // the filterbank mel_tab holds (0: none yet)
    unsigned int* range_flag_dev = nullptr;
    unsigned int* range_flag_host = nullptr;
    bool range_fallback = true;      // repeat such a pass on the exact float32 kernels (else: CSS_ERR_RANGE)
    int64_t range_fallbacks = 0;     // passes repeated so far
    int range_last = 0;              // the last pass hit the range limit
    hipStream_t copy_stream = nullptr;
    // css_run*: what follows the mask estimator (covariances and beamformer per segment on the lanes, then -- in segment
    // order, on this stream -- stitching costs, the permutation scan, overlap-add, gate, synthesis) trails the lanes unit
    // by unit instead of waiting for the last segment of the recording
    hipStream_t tail_stream = nullptr;
    // schedule choices of that pipeline (css_set_tuning; defaults = what measured best, A/B on one box: tools/ab_tuning.py)
    int tune[CSS_TUNE_COUNT] = {1, 0, 0, 1, 0, 2, 1, 0, 1, 0, 24576, 14000};
    const void* mapped_key = nullptr;   // last page-locked output buffer looked up, and its device address
    void* mapped_val = nullptr;
    // css_upload_range: further pieces of the recording on their way over PCIe (copy stream); css_stage_stft_range makes
    // the handle's stream wait for exactly the pieces its frames read
    // css_run_enqueue / css_wait: passes enqueued and not yet waited for
    int queued = 0;
    // queued passes overlap: pass P's samples cross PCIe while pass P - 1's kernels run, and P - 1's stitching / synthesis /
    // download run beside P's estimator.  The sample buffer and the level word alternate (pass parity); `pcm_free[b]` =
    // the last transform of the pass that used sample buffer b; `tail_end` = the end of the last queued pass's tail
    int64_t pass_no = 0;
    hipEvent_t pcm_free[2] = {nullptr, nullptr};
    hipEvent_t pass_end[4] = {nullptr, nullptr, nullptr, nullptr};   // ends of the last four queued passes (back-pressure)
    hipEvent_t level_free[2] = {nullptr, nullptr};   // end of the tail of the pass that used level word b (its last reader)
    hipEvent_t tail_end = nullptr;
    bool tail_pending = false;
    bool piped_now = false;   // run_once -> begin_impl: the level word is cleared on the copy stream, not here
    int last_piped = -1;      // overlap mode of the last queued pass (-1: nothing queued): a queue never mixes modes un-drained
    // css_run_enqueue's arguments since the last css_wait: a queued pass that left the split-f16 range is repeated from
    // them on the exact float32 kernels (the caller keeps pcm_host valid and wav_host untouched until css_wait anyway)
    struct QueuedPass {
        const float* pcm; int64_t n; int32_t n_ch; CssRunCfg cfg; float* wav; int64_t cap;
        std::vector<const int16_t*> planes; int16_t* wav16 = nullptr; float* peaks = nullptr;   // css_run_enqueue_pcm16 (pcm == wav == nullptr)
        std::vector<float> w;   // the three stitching windows of cfg, copied at css_run_enqueue (the caller may free its own)
        QueuedPass(const float* pcm_, int64_t n_, int32_t n_ch_, const CssRunCfg& c, float* wav_, int64_t cap_)
            : pcm(pcm_), n(n_), n_ch(n_ch_), cfg(c), wav(wav_), cap(cap_) {
            const size_t T = (size_t)std::max(c.segment_frames, 0);
            w.resize(3 * T);
            if (T && c.w_first && c.w_mid && c.w_last) {
                std::memcpy(w.data(), c.w_first, T * sizeof(float));
                std::memcpy(w.data() + T, c.w_mid, T * sizeof(float));
                std::memcpy(w.data() + 2 * T, c.w_last, T * sizeof(float));
            }
        }
        CssRunCfg own_cfg() const {   // cfg with its window pointers at this entry's copies
            CssRunCfg c = cfg;
            const size_t T = w.size() / 3;
            c.w_first = w.data(); c.w_mid = w.data() + T; c.w_last = w.data() + 2 * T;
            return c;
        }
    };
    std::vector<QueuedPass> queue_log;
    // css_run_enqueue: sessions accepted and not yet on the streams -- they wait for company: sessions of one segment
    // length are merged into ONE estimator batch (run_group) as long as their segments fit max_batch_segments
    struct Pending { const float* pcm; int64_t n; int32_t n_ch; CssRunCfg cfg; std::vector<float> w; float* wav; int64_t cap;
                     float* wav_mapped; int64_t nseg;
                     std::vector<const int16_t*> planes; int16_t* wav16 = nullptr; float* peaks = nullptr; };   // PCM16 edges: pcm == wav == nullptr
    std::vector<Pending> pending;
    int64_t pending_segments = 0;
    // css_wait_sessions: one event per session put on the streams since the last css_wait, in queue order, recorded behind the
    // session's last output copy (nullptr: the session had finished inside its call)
    std::vector<hipEvent_t> sess_done;
    std::vector<hipEvent_t> sess_ev_pool;
    size_t sess_ev_used = 0;
    void* comm = nullptr;          // ncclComm_t of css_comm_init (RCCL, loaded lazily)
    int comm_ranks = 0, comm_rank = -1;
    struct PendingUpload { int64_t s_lo, s_hi; hipEvent_t landed; };
    std::vector<PendingUpload> uploads;
    std::vector<hipEvent_t> ev_pool;   // untimed events of the pipeline (uploads landed, planes ready, ranges finished)
    size_t ev_pool_used = 0;

    // timing
    hipEvent_t ev[10]{};
    CssTimings tim{};
    // css_set_profile: every kernel launch of a pass is bracketed by a pair of HIP events on its stream (one lane, so
    // that the pairs are ordered); durations are summed per kernel family (css_get_kernel_stats)
    bool profile_gemm = false;
    struct ProfEvent { hipEvent_t a, b; int cat; };
    std::vector<ProfEvent> prof_events;
    size_t prof_used = 0;
    size_t prof_reduced = 0;   // brackets already summed into prof_ms (a staged session has no closing call that does it)
    double gemm_flops = 0.0;
    float prof_ms[CSS_PROF_COUNT] = {};
    int32_t prof_launches[CSS_PROF_COUNT] = {};
    // what an event pair measures with NOTHING between the two records (css_set_profile calibrates it): the part of every
    // bracketed launch that is the bracket, not the kernel
    float prof_pair_ms = 0.f;
    int32_t prof_pairs = 0;
    FeatOpts feat_opts{};   // css_set_feature_options (css_create: the shipped configuration)

    std::string err;
};

namespace {

int fail(css_ctx* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    else g_create_error = msg;
    return code;
}

#define HIPCHK(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(h, CSS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

int ensure(css_ctx* h, DevBuf& b, size_t bytes, bool zero = false) {
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

// Scope around one kernel launch: with the profile on, an event pair on the launch's stream, tagged with its family.
struct Prof {
    css_ctx* h; hipStream_t st; hipEvent_t stop = nullptr;
    Prof(css_ctx* h_, int cat, hipStream_t st_) : h(h_), st(st_) {
        if (!h->profile_gemm) return;
        if (h->prof_used == h->prof_events.size()) {
            css_ctx::ProfEvent e{};
            hipEventCreate(&e.a);
            hipEventCreate(&e.b);
            h->prof_events.push_back(e);
        }
        css_ctx::ProfEvent& e = h->prof_events[h->prof_used++];
        e.cat = cat;
        hipEventRecord(e.a, st);
        stop = e.b;
    }
    ~Prof() { if (stop) hipEventRecord(stop, st); }
};
#define CSS_PROF(cat, st) Prof prof_scope_(h, cat, st)

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
static int64_t batch_cap(const css_ctx* h, int T) {
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

}  // namespace

// queued passes (css_run_enqueue) finish before anything else touches the handle's state or buffers
#define CSS_DRAIN(h)                                                  \
    do {                                                              \
        if ((h) && ((h)->queued || !(h)->pending.empty())) {          \
            const int rc_drain_ = css_wait(h);                        \
            if (rc_drain_ != CSS_OK) return rc_drain_;                \
        }                                                             \
    } while (0)

// =================================================================================================
extern "C" {

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
static int upload_analysis_matrix(css_ctx* h, int window) {
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
static bool analysis_transform(css_ctx* h, const float* x, int64_t x_stride, int C, int64_t t_lo, int64_t t_hi, float* out,
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

int css_pit_scan(const double* costs, int64_t n_boundaries, int32_t num_spks, int32_t* perms) {
    if (!perms || n_boundaries < 0 || num_spks < 1 || num_spks > 4 || (n_boundaries > 0 && !costs)) return CSS_ERR_INVALID_ARG;
    pit_scan_host(costs, n_boundaries, num_spks, perms);
    return CSS_OK;
}

// -------------------------------------------------------------------------------------------------
// Opens a session: validates the configuration, fixes the plan, sizes the workspace.  No sample moves here.
static int check_run_args(css_handle_t h, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg, CssPlan* plan_out) {
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

static int begin_impl(css_handle_t h, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg) {
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
static int upload_pcm(css_handle_t h, const float* pcm_host, int64_t s_lo, int64_t s_hi, hipStream_t st) {
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
static inline int64_t covered_end(const css_ctx* h) {
    const int64_t fr = h->plan.stft_frames;
    return fr > 0 ? std::min<int64_t>((fr - 1) * h->d.frame_hop + h->d.frame_len, h->plan.n_samples) : 0;
}
static inline int64_t peak_len(const css_ctx* h, int64_t s_lo, int64_t s_hi) {   // samples of [s_lo, s_hi) to scan
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

static hipEvent_t pool_event(css_ctx* h);

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
static int stft_frames(css_ctx* h, int64_t t_lo, int64_t t_hi, const int16_t* planes16, hipStream_t st) {
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

// Where one batched pass of the mask estimator reads its spectra and writes its masks.
struct GroupSess { const float* X; int64_t T_ld, stft_frames; int64_t off; int n; const float* PH; };   // a session's planes (+ phase planes); its segments are the batch's [off, off + n)
struct MaskIo {
    const float* X; int64_t T_ld; int64_t stft_frames; int hop; int T;   // planes [C][2F][T_ld], segment s at s*hop
    float* masks; int64_t mask_ld;                                       // [(S+1)F][mask_ld], segment s at column s*T
    // a batch over the segments of SEVERAL sessions (run_group): the features of batch segment c come from the session
    // that holds it, everything behind them is one [segments * T, .] problem; X / T_ld / stft_frames above are unused
    const std::vector<GroupSess>* group = nullptr;
    const float* PH = nullptr;   // phase planes [C][F][T_ld] beside X (nullptr: the feature kernel forms the phases itself)
};

// The mask estimator over `nb` segments starting at `s0` on one lane (stream + activation set), phases [ph_lo, ph_hi):
// phase -1 = features + embed, phase l = Conformer block l, phase num_blocks = mask head.
static int masknet_lane(css_ctx* h, const MaskIo& io, int64_t s0, int nb, int lane, int ph_lo, int ph_hi,
                        bool concurrent = false) {
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

// How a batch of `nb` segments is cut into lanes: `nl` chains of `per` segments (the last one shorter).
struct LaneSplit { int nl, per; };
static LaneSplit lane_split(const css_ctx* h, int nb, int T) {
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
static int64_t batch_len(int64_t n, int64_t cap) {
    const int64_t nbat = (n + cap - 1) / cap;
    return nbat ? (n + nbat - 1) / nbat : 0;
}

// One batched pass of the mask estimator over `nb` segments starting at `s0`: `lanes` part batches on as many streams
// (see css_ctx::lanes).  prep(first segment, count, stream), when given, is enqueued at the head of each lane's chain:
// the fused path puts the analysis transform of the frames that lane is the first to read there (run_impl).
using LanePrep = std::function<int(int64_t, int, hipStream_t)>;
using LanePost = LanePrep;   // post(first segment, count, stream): enqueued at the END of each lane's chain
static int masknet_batch(css_ctx* h, const MaskIo& io, int64_t s0, int nb, const LanePrep& prep, const LanePost& post,
                         hipEvent_t before_head = nullptr) {
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
static int masknet_batch(css_ctx* h, const MaskIo& io, int64_t s0, int nb) {
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
static int mvdr_on(css_ctx* h, int64_t seg_lo, int64_t seg_hi, hipStream_t st) {
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

static void pit_costs_on(css_ctx* h, int64_t b_lo, int64_t b_hi, hipStream_t st) {
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
static void pit_scan_on(css_ctx* h, int64_t b_lo, int64_t b_hi, hipStream_t st) {
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

static int check_frames(css_ctx* h, int64_t t_lo, int64_t t_hi) {
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
static void istft_gemm_on(css_ctx* h, int64_t f_lo, int64_t f_hi, hipStream_t st) {
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
static void wave_ola_on(css_ctx* h, int64_t f_lo, int64_t f_hi, int64_t q_lo, int64_t q_hi, float* out, int64_t out_ld,
                        int64_t out_q0, hipStream_t st) {
    if (f_hi <= f_lo) return;
    CSS_PROF(CSS_PROF_WAVE_OLA, st);
    launch_wave_ola((const float*)h->G.p, out, h->d.num_spks, h->plan.mix_frames, h->d.frame_hop, h->d.frame_len, q_lo, q_hi, f_lo, f_hi, out_ld,
                    out_q0, h->split ? h->peak_dev : nullptr, st);
}
static int istft_impl(css_ctx* h, int64_t f_lo, int64_t f_hi, int64_t q_lo, int64_t q_hi, float* out, int64_t out_ld,
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
static void* mapped_host(const void* p) {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return at.type == hipMemoryTypeHost ? at.devicePointer : nullptr;
}

// the completion event of the session whose last output copy was just enqueued on `st` (css_wait_sessions)
static void mark_session_done(css_ctx* h, hipStream_t st) {
    if (h->sess_ev_used == h->sess_ev_pool.size()) {
        hipEvent_t e = nullptr;
        hipEventCreateWithFlags(&e, hipEventDisableTiming);
        h->sess_ev_pool.push_back(e);
    }
    hipEvent_t e = h->sess_ev_pool[h->sess_ev_used++];
    hipEventRecord(e, st);
    h->sess_done.push_back(e);
}

static hipEvent_t pool_event(css_ctx* h) {
    if (h->ev_pool_used == h->ev_pool.size()) {
        hipEvent_t e = nullptr;
        hipEventCreateWithFlags(&e, hipEventDisableTiming);
        h->ev_pool.push_back(e);
    }
    return h->ev_pool[h->ev_pool_used++];
}

// the per-launch event brackets recorded since the session began (css_set_profile) -> per-family sums; the streams
// they were recorded on must have been synchronised
static void reduce_profile(css_ctx* h) {
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
static int finish_timings(css_ctx* h, HostClock t0, HostClock t1, HostClock t2, bool staged) {
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
static int run_once(css_handle_t h, int64_t n, int32_t n_ch, const CssRunCfg* cfg, const RunIo& io) {
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

static int run_group(css_handle_t h, std::vector<css_ctx::Pending>& grp) {
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
static int flush_pending(css_handle_t h) {
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
static int run_impl(css_handle_t h, int64_t n, int32_t n_ch, const CssRunCfg* cfg, const RunIo& io) {
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

}  // extern "C"

// Internal header of libcss_mi355.so's host side (round 6: api.hip split into three units).  The C ABI is include/css_mi355.h;
// nothing here is exported on purpose.  css_ctx is the handle: the model, its streams and workspaces, and -- through SessState --
// the session the stage entry points see.
//   api_core.hip    handle life cycle (css_create / css_destroy), weights, setters, timings, buffers, the RCCL communicator
//   api_stages.hip  one stage per reference function (css_begin .. css_stage_*), the mask estimator's lanes, the separator protocol
//   api_queue.hip   the fused pass (css_run*), the queue of sessions (css_run_enqueue* / css_wait*), shared estimator batches
#pragma once
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


thread_local extern std::string g_create_error;   // css_last_error(NULL): why the last css_create of this thread failed

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

#define HIPCHK(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(h, CSS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)


// ---- api_core.hip
int fail(css_ctx* h, int code, const std::string& msg);
int ensure(css_ctx* h, DevBuf& b, size_t bytes, bool zero = false);
int64_t bind_weights(const CssModelDesc& d, const float* base, Weights* w);
const char* validate_desc(const CssModelDesc& d);
void exact_cs(int64_t k, int N, double* c, double* s);
bool deal_streams(hipStream_t main, hipStream_t lane[4], hipStream_t* copy, hipStream_t* tail);
int plan_impl(const CssModelDesc& d, const CssRunCfg& cfg, int64_t n, CssPlan* p);
void gemm(css_ctx* h, const GemmArgs& g, hipStream_t st);
GemmArgs linear(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc, int M, int N, int K, int act);
StitchArgs stitch_args(css_ctx* h);
MvdrArgs mvdr_args(css_ctx* h, int64_t lo, int nseg);
int64_t batch_cap(const css_ctx* h, int T);
int ensure_activations(css_ctx* h, int64_t nb, int T);
int make_split_weights(css_ctx* h);
int make_frag_weights(css_ctx* h);
int check_session(css_ctx* h);
int upload_analysis_matrix(css_ctx* h, int window);
bool analysis_transform(css_ctx* h, const float* x, int64_t x_stride, int C, int64_t t_lo, int64_t t_hi, float* out, int64_t row_ld,
                        hipStream_t st, float* phase, bool* phase_done);

// ---- api_stages.hip
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
struct LaneSplit { int nl, per; };   // how a batch of segments is cut into lanes: `nl` chains of `per` segments (the last one shorter)
using LanePrep = std::function<int(int64_t, int, hipStream_t)>;
using LanePost = LanePrep;   // post(first segment, count, stream): enqueued at the END of each lane's chain
int check_run_args(css_handle_t h, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg, CssPlan* plan_out);
int begin_impl(css_handle_t h, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg);
int upload_pcm(css_handle_t h, const float* pcm_host, int64_t s_lo, int64_t s_hi, hipStream_t st);
int64_t covered_end(const css_ctx* h);
int64_t peak_len(const css_ctx* h, int64_t s_lo, int64_t s_hi);
int stft_frames(css_ctx* h, int64_t t_lo, int64_t t_hi, const int16_t* planes16, hipStream_t st);
int masknet_lane(css_ctx* h, const MaskIo& io, int64_t s0, int nb, int lane, int ph_lo, int ph_hi, bool concurrent = false);
LaneSplit lane_split(const css_ctx* h, int nb, int T);
int64_t batch_len(int64_t n, int64_t cap);
int masknet_batch(css_ctx* h, const MaskIo& io, int64_t s0, int nb, const LanePrep& prep, const LanePost& post, hipEvent_t before_head = nullptr);
int masknet_batch(css_ctx* h, const MaskIo& io, int64_t s0, int nb);
int mvdr_on(css_ctx* h, int64_t seg_lo, int64_t seg_hi, hipStream_t st);
void pit_costs_on(css_ctx* h, int64_t b_lo, int64_t b_hi, hipStream_t st);
void pit_scan_on(css_ctx* h, int64_t b_lo, int64_t b_hi, hipStream_t st);
int check_frames(css_ctx* h, int64_t t_lo, int64_t t_hi);
void istft_gemm_on(css_ctx* h, int64_t f_lo, int64_t f_hi, hipStream_t st);
void wave_ola_on(css_ctx* h, int64_t f_lo, int64_t f_hi, int64_t q_lo, int64_t q_hi, float* out, int64_t out_ld, int64_t out_q0, hipStream_t st);
int istft_impl(css_ctx* h, int64_t f_lo, int64_t f_hi, int64_t q_lo, int64_t q_hi, float* out, int64_t out_ld, int64_t out_q0, hipStream_t st);

// ---- api_queue.hip
hipEvent_t pool_event(css_ctx* h);
void reduce_profile(css_ctx* h);


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

// queued passes (css_run_enqueue) finish before anything else touches the handle's state or buffers
#define CSS_DRAIN(h)                                                  \
    do {                                                              \
        if ((h) && ((h)->queued || !(h)->pending.empty())) {          \
            const int rc_drain_ = css_wait(h);                        \
            if (rc_drain_ != CSS_OK) return rc_drain_;                \
        }                                                             \
    } while (0)


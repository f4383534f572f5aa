/*
 * css_mi355.h -- C ABI of the MI355X-native continuous speech separation (CSS) front end.
 *
 * This is the drop-in boundary for the NOTSOFAR baseline's CSS hot path.  The reference is pure
 * Python and has no FFI of its own; its boundary is a Python calling convention
 * (css/css.py:51 css_inference, css/css.py:110 separate_and_stitch, and the separator protocol
 * stft/separate/istft of css/training/conformer_wrapper.py:79-146).  Each entry point below names the
 * reference function (file:line under the reference root) whose arithmetic it replaces; the Python
 * shim in notsofar1-challenge_amd/ re-exposes them under the reference's names via ctypes
 * (see INTEGRATION.md for the binding a maintainer would add on the reference side).
 *
 * Conventions
 *   - plain C types only; every buffer is caller-owned; "host" pointers are ordinary memory, "dev"
 *     pointers are HIP device memory of the handle's device;
 *   - every function returns a css_status (0 = ok, negative = error); css_last_error() gives text;
 *     the negative codes map 1:1 onto the reference's asserts (css/css.py:139,196,202,224,297);
 *   - a handle owns one HIP stream and its workspace; it is not thread-safe; there is no global state;
 *   - all arithmetic is float32 like the reference, except the 7x7 spatial-covariance accumulation
 *     and MVDR solve, which run in float64 (the reference's complex64 solve is
 *     itself ~2e-5 from the exact answer; float64 keeps our distance from it at that floor).
 *
 * Device data layout (HBM, all row-major, last index fastest)
 *   pcm_cm   [C][n_pad]                 channel-major samples (deinterleaved once on upload)
 *   X        [C][2*F][T_ld]             STFT planes: rows 0..F-1 = Re, rows F..2F-1 = Im; time fastest
 *   feat     [tokens][K_pad]            network input, token = local_segment*T_seg + t
 *   masks    [(S+1)*F][tokens_total]    sigmoid masks, row = k*F + f, column = segment*T_seg + t
 *   sep      [segments][S][F][T_seg]    separated segment spectra (interleaved re,im float2)
 *   scm      [segments][S+1][F][49]     Hermitian 7x7 packed (7 real diag + 21 complex), float64
 *   bfw      [segments][S][F][C]        MVDR weights, complex float64
 *   mask_st  [S][F][T_long]             stitched masks
 *   Y        [S][T_long][KI_pad]        gated stitched spectra for the inverse transform (Re | Im | 0-pad)
 *   wav      [S][n_out]
 */
#ifndef CSS_MI355_H
#define CSS_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the entry points declared in this header are exported. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef struct css_ctx* css_handle_t;

typedef enum css_status {
    CSS_OK = 0,
    CSS_ERR_INVALID_ARG = -1,    /* css.py:139 (ndim), :196 (batch == 1), bad pointers / sizes          */
    CSS_ERR_HIP = -2,            /* a HIP runtime call failed (text in css_last_error)                     */
    CSS_ERR_ZERO_WEIGHT = -3,    /* css.py:297  "zero weights found. check hop_size, segment_size..."     */
    CSS_ERR_MASK_FLOOR = -4,     /* css.py:224  assert mask_floor_db <= 0                                   */
    CSS_ERR_SHAPE = -5,          /* css.py:202-203 mask / stft shape mismatch, channel count vs model      */
    CSS_ERR_STATE = -6,          /* stage called before the stage it depends on                            */
    CSS_ERR_NO_DEVICE = -7,      /* no usable gfx950 device                                                */
    CSS_ERR_WEIGHT_WINDOW = -8,  /* css.py:374  not enough frames to fit the weighting window              */
    CSS_ERR_RANGE = -9           /* an operand left the split-f16 range and the float32 fallback is off    */
} css_status;

/* Architecture of the mask estimator.  Mirrors the dataclasses of
 * css/training/conformer_wrapper.py:11-48 (ExtractorCfg / ConformerCfg / NnetCfg). */
typedef struct CssModelDesc {
    int32_t num_mics;        /* 7 (multi-channel) or 1 (single-channel)                 */
    int32_t num_bins;        /* F = frame_len/2 + 1 = 257                               */
    int32_t in_features;     /* 1799 = F*(1 + 6 IPD pairs), or 257                      */
    int32_t attention_dim;   /* D = 512                                                 */
    int32_t attention_heads; /* H = 8                                                   */
    int32_t linear_units;    /* FF = 1024                                               */
    int32_t num_blocks;      /* 18                                                      */
    int32_t kernel_size;     /* depthwise conv taps, 33                                 */
    int32_t num_spks;        /* S = 3                                                   */
    int32_t num_nois;        /* 1                                                       */
    int32_t frame_len;       /* 512                                                     */
    int32_t frame_hop;       /* 256                                                     */
    int32_t maxlen;          /* relative-position table half size, 1000 (conformer.py:213); offsets past it are clamped (:24) */
} CssModelDesc;

/* Feature-extractor options beyond what the shipped v1.0 models use (ExtractorCfg, conformer_wrapper.py:11-24;
 * implemented by FeatureExtractor / IPDFeature, css/css_with_conformer/executor/feature.py:198-249,478-508).  css_create
 * starts from the shipped configuration: {0, 1, 1, 1, 0, C - 1 pairs (m, 0)}. */
#define CSS_MAX_IPD_PAIRS 16
typedef struct CssFeatureCfg {
    int32_t log_spectrogram;            /* feature.py:500-501: log of the clamped magnitude before the normalisation          */
    int32_t mvn_spectrogram;            /* feature.py:503-507: mean / (unbiased) std normalisation over the segment's frames  */
    int32_t ipd_mean_normalize;         /* feature.py:214: remove the time-mean of the phase difference ...                   */
    int32_t ipd_mean_normalize_version; /* ... 1: atan2(sin - mean sin, cos - mean cos) (:220-221); 2: minus atan2(mean sin,
                                         * mean cos) (:222-224); 3: minus the mean angle (:225-227)                            */
    int32_t ipd_cos;                    /* feature.py:234-236: cos of the (normalised) difference instead of the raw angle     */
    int32_t num_pairs;                  /* microphone pairs of ipd_index, e.g. "1,4;2,5;3,6" -> 3; in_features = F (1 + pairs) */
    int32_t pair_l[CSS_MAX_IPD_PAIRS];  /* phase difference = phase[pair_l] - phase[pair_r] (feature.py:212)                   */
    int32_t pair_r[CSS_MAX_IPD_PAIRS];
} CssFeatureCfg;

/* Run-time knobs.  Mirrors the arithmetic-relevant fields of CssCfg (css/css.py:24-48) after the
 * seconds->frames conversion of css/css.py:144-152, which the host shim performs with the same
 * Python float expressions. */
#define CSS_MAX_SEGMENT_FRAMES 16384   /* 262 s: a sanity bound (one query's score rows must fit the LDS) */
typedef struct CssRunCfg {
    int32_t segment_frames;          /* 186  (css.py:147).  2 .. CSS_MAX_SEGMENT_FRAMES.  Up to 512 frames (8 s) the attention
                                      * kernel keeps a query tile's scores over all keys in registers and the feature /
                                      * covariance kernels a segment's rows in LDS; longer segments (the reference accepts
                                      * any segment_size_sec) run on any-length forms of those three kernels -- the same
                                      * results to the path's tolerances, several times slower per second of audio    */
    int32_t hop_frames;              /* 93   (css.py:148).  1 <= hop < segment_frames (css.py:276 needs an overlap)   */
    int32_t dilation_frames;         /* 24   (css.py:151)                                         */
    int32_t erosion_frames;          /* 12   (css.py:152)                                         */
    int32_t mc_mvdr;                 /* css.py:211                                                */
    int32_t stitching_loss;          /* 0 = 'l1', 1 = 'mse'                  (css.py:263)         */
    int32_t stitching_input;         /* 0 = 'mask', 1 = 'separation_result'  (css.py:267-271)     */
    int32_t normalize_segment_power; /* css.py:233                                                */
    float mask_floor;                /* 10^(mask_floor_db/20), 0 for -inf   (css.py:225)          */
    float activity_th;               /* css.py:304                                                */
    const float* w_first;            /* [segment_frames] calc_segment_weight(is_first_seg=True)   */
    const float* w_mid;              /* [segment_frames] calc_segment_weight()                    */
    const float* w_last;             /* [segment_frames] calc_segment_weight(is_last_seg=True)    */
} CssRunCfg;

/* Index arithmetic of css/css.py:155-171 for one input length. */
typedef struct CssPlan {
    int64_t n_samples;
    int64_t stft_frames;   /* floor((N - frame_len)/hop) + 1 (feature.py:116), 0 if N < frame_len */
    int64_t mix_frames;    /* max(stft_frames, segment_frames)     (css.py:159-164)                */
    int64_t num_segments;  /* ceil((mix_frames - overlap)/hop)     (css.py:166-169)                */
    int64_t n_out;         /* (mix_frames - 1)*hop + frame_len     (feature.py:162)                */
    int32_t last_valid;    /* valid frames of the last segment     (css.py:185-190)                */
    int32_t zero_weight;   /* 1 if css.py:297 would assert                                         */
} CssPlan;

/* Per-stage wall time of the last css_run*, measured with HIP events on the handle's stream (ms). */
typedef struct CssTimings {
    float upload, stft, features, masknet, mvdr, stitch, istft, download, total;
    float gemm_ms;          /* sum of the durations of the MFMA GEMM launches inside masknet          */
    int64_t gemm_launches;
    double gemm_flops;      /* algorithmic FLOPs of those launches (2*M*N*K each)                     */
    float host_enqueue;     /* host wall time from the call to the last enqueue (before the final wait), ms */
    float host_total;       /* host wall time of the whole call, ms                                         */
} CssTimings;

/* Identifiers of device buffers readable / writable through css_read_buffer / css_write_buffer
 * (stage-level parity tests; SURVEY.md 8(b) "stage-level entry points"). */
typedef enum css_buffer {
    CSS_BUF_X = 0,          /* float  [C][2F][T_ld]                  (see css_buffer_dims)          */
    CSS_BUF_FEATURES = 1,   /* float  [batch tokens][K_pad]          last processed batch           */
    CSS_BUF_MASKS = 2,      /* float  [(S+1)F][num_segments*T_seg]                                  */
    CSS_BUF_SCM = 3,        /* double [segments][S+1][F][49]                                        */
    CSS_BUF_BFW = 4,        /* double [segments][S][F][C][2]                                        */
    CSS_BUF_SEP = 5,        /* float  [segments][S][F][T_seg][2]                                    */
    CSS_BUF_PIT_COST = 6,   /* double [segments-1][S*S]  raw (unpermuted) costs                     */
    CSS_BUF_PERMS = 7,      /* int32  [segments][S]                                                 */
    CSS_BUF_MASK_ST = 8,    /* float  [S][F][T_long]                                                */
    CSS_BUF_ACTIVITY = 9,   /* float  [S][T_long]  mean over F of mask_st                           */
    CSS_BUF_ACT_B = 10,     /* uint8  [S][T_long]  activity >= th                                   */
    CSS_BUF_ACT_FINAL = 11, /* uint8  [S][T_long]  after dilate/erode                               */
    CSS_BUF_Y = 12,         /* float  [S][T_long][KI_pad]                                           */
    CSS_BUF_WAV = 13,       /* float  [S][n_out]                                                    */
    CSS_BUF_HIDDEN = 14,    /* float  [batch tokens][D]  encoder output of the last batch           */
    CSS_BUF_WTA_OVERRIDE = 15,/* uint8 [segments][F][T_seg]  (write-only) injected WTA decisions: 0..3 = the one winning mask;
                               * 16 + bits = the SET of winners (bit j: mask j wins) -- mvdr_util.py:53-54 keeps every mask that
                               * equals the maximum, and a trained model's saturated masks tie exactly                        */
    CSS_BUF_LEVEL = 16      /* float  [1]  max |sample| of the PCM laid out so far in this session: sets the power-of-two
                               gain of the split-f16 synthesis operand; ranks of a sharded meeting exchange its maximum */
} css_buffer;

/* ---- library ------------------------------------------------------------------------------ */
const char* css_version(void);
/* Text of the last error of `h` (or of the last failed css_create when h == NULL). */
const char* css_last_error(css_handle_t h);
/* Number of visible HIP devices (0 when there is none; never fails). */
int css_device_count(void);

/* ---- model -------------------------------------------------------------------------------- */
/* Size in floats of the weight blob css_create expects for `desc` (layout below). */
int64_t css_blob_num_floats(const CssModelDesc* desc);

/* Replaces css/helpers.py:14 load_css_model + nn.Module.to(device) (css.py:178): uploads the weights
 * once; they stay resident in HBM for the life of the handle.
 * `stream` is an existing hipStream_t to launch on (e.g. torch's current stream) or NULL to create one.
 * `max_batch_segments` bounds the activation workspace (segments per batched network pass).
 *
 * Weight blob: float32 sections, each starting on a 16-float boundary, in this order
 *   input_bias[Kp] input_scale[Kp] embed_w[D][Kp] embed_b[D] embed_ln_w[D] embed_ln_b[D] pe_k[2*maxlen][D/H]
 *   per block: ffi{ln_w,ln_b,w1[FF][D],b1[FF],w2[D][FF],b2[D]} att_ln_w att_ln_b wqkv[3D][D] bqkv[3D] wo[D][D] bo[D]
 *              conv_ln_w conv_ln_b pw[8]={pw1.w0,pw1.b0,pw1.w1,pw1.b1,pw2.w,pw2.b,0,0} dw_wt[taps][D] dw_b[D]
 *              bn_alpha[D] bn_beta[D] ffo{...} fin_ln_w fin_ln_b
 *   head_w[(S+1)F][D] head_b[(S+1)F]
 * with Kp = in_features rounded up to a multiple of 32 (zero padded).  Built by
 * notsofar1-challenge_amd/weights.py::pack_blob from a reference state_dict. */
int css_create(const CssModelDesc* desc, const float* blob_host, int64_t blob_floats, int device,
               void* stream, int32_t max_batch_segments, css_handle_t* out);
int css_destroy(css_handle_t h);
/* The hipStream_t every kernel of `h` is ordered on (the one passed to css_create, or the one it created): a caller
 * that enqueues its own work -- torch.distributed collectives between the stages of a sharded meeting, say -- wraps it
 * (torch.cuda.ExternalStream) and needs no host synchronisation between its work and the handle's. */
int css_get_stream(css_handle_t h, void** stream_out);
/* Number of independent kernel chains ("lanes", 1..4, default 3) a batch of segments is cut into inside the mask
 * estimator (segments are independent through the network, css.py:182-250 carries no state between them).  Results do
 * not depend on it, bit for bit.  It is an upper bound: the exact float32 mode takes a second lane only from ~14 000 token
 * rows per lane and never a third (its matrix products want rows per launch more than a chain beside them). */
int css_set_lanes(css_handle_t h, int lanes);
int css_get_lanes(css_handle_t h);
/* Schedule choices of the css_run* pipeline.  They change WHEN work is enqueued, never a result bit; the defaults are what
 * measured best on the 60 s / 30 min meetings (tools/ab_tuning.py runs the alternatives on one box). */
enum css_tuning {
    CSS_TUNE_TAIL_PIECES = 0,   /* pieces the last frame range is synthesised and downloaded in (1..4, default 1)          */
    CSS_TUNE_OUT_MAPPED = 1,    /* 1: the overlap-add kernel writes page-locked output over PCIe itself; 0 (default): DMA  */
    CSS_TUNE_TAIL_PER_UNIT = 2, /* 1: stitch / synthesise after every lane's unit; 0 (default): once per batch             */
    CSS_TUNE_MVDR_ON_LANES = 3, /* 1 (default): covariances / MVDR / stitching costs at the end of each lane's chain; 0: after  */
    CSS_TUNE_PIPELINE_DEVICE = 4, /* 1: css_run_device also takes the unit pipeline; 0 (default): the plain stage sequence    */
    CSS_TUNE_GROUP_LANES = 5,   /* lanes of a batch shared by queued sessions (css_run_enqueue), 1..4, default 2               */
    CSS_TUNE_GROUP_TRANSFORM_ON_MAIN = 6, /* 1 (default): such a pass's analysis transforms as a prefix of the main stream; 0: on
                                           * the copy stream, behind each session's upload, beside the previous pass's estimator
                                           * (1 % slower in interleaved A/B: what the copy stream runs competes with the estimator) */
    CSS_TUNE_GROUP_MVDR_ON_LANES = 7,     /* 1: its covariances / MVDR / stitching costs on the lanes' streams, joined; 0 (default):
                                           * on the tail stream with everything else behind the mask head                        */
    CSS_TUNE_GROUP_OUT_DMA = 8,           /* 1 (default): its waveforms are overlap-added into HBM and copied out by DMA; 0: the
                                           * overlap-add kernel writes the page-locked output over PCIe itself                   */
    CSS_TUNE_F32_GEMM = 9,                /* CSS_LINEAR_EXACT_F32 products: 0 (default): gemm_f32.hip, four independent blocks per CU and
                                           * tile heights balanced over the CUs; 1: the round-4 kernel (gemm.hip); 2..5: gemm_f32.hip with
                                           * every tile 32 / 64 / 96 / 128 rows; 6: gemm_f32.hip with the weights row-major through LDS
                                           * (the default reads them as register fragments).  Same bits whichever                  */
    CSS_TUNE_SPLIT_BATCH_ROWS = 10,       /* split-f16 mode: token rows (segments x frames) per estimator batch, whatever max_batch_segments
                                           * allows (default 24576 = 128 segments of 3 s: beyond it a batch's activations leave the
                                           * Infinity Cache between producer and consumer); 0: no such bound.  The exact float32 mode
                                           * always batches up to max_batch_segments                                               */
    CSS_TUNE_F32_LANE_ROWS = 11,          /* exact float32 mode: token rows per lane from which a batch takes a second lane (default
                                           * 14000 = 75 segments of 3 s; never a third); 1: as many lanes as css_set_lanes gives (tests) */
    CSS_TUNE_COUNT = 12
};
int css_set_tuning(css_handle_t h, int which, int value);
/* Page-locked host memory for PCM / waveform buffers: css_run* on such buffers moves the samples over PCIe by DMA,
 * asynchronously, in pieces that overlap the first and last kernels of the pass; pageable memory works too, at the
 * driver's staged-copy rate.  (hipHostMalloc / hipHostFree; no handle needed.) */
int css_host_alloc(size_t bytes, void** out);
int css_host_free(void* p);

/* ---- planning (pure host arithmetic, no GPU needed) ---------------------------------------- */
/* css/css.py:155-171 + css.py:297: frames, segments, output length for an input of n_samples. */
int css_plan(const CssModelDesc* desc, const CssRunCfg* cfg, int64_t n_samples, CssPlan* out);
/* Sequential permutation scan of css/css.py:266-285 over raw PIT cost matrices (losses.py:32-48):
 * costs [n_boundaries][S*S] -> perms [(n_boundaries+1)][S]; perms[0] = identity. */
int css_pit_scan(const double* costs, int64_t n_boundaries, int32_t num_spks, int32_t* perms);

/* CssCfg's fields in the reference's own units (css/css.py:24-48), for a host without the Python shim. */
typedef struct CssCfgSeconds {
    double segment_size_sec;        /* 3.0   css.py:26                                                */
    double hop_size_sec;            /* 1.5   css.py:27                                                */
    double seg_weight_m0_sec;       /* 0.15  css.py:30                                                */
    double seg_weight_m1_sec;       /* 0.3   css.py:31                                                */
    double activity_dilation_sec;   /* 0.4   css.py:33                                                */
    double activity_erosion_sec;    /* 0.2   css.py:34                                                */
    double activity_th;             /* 0.4   css.py:32 (configs/inference/inference_v1.yaml:17: 0.3)  */
    double mask_floor_db;           /* mc_mask_floor_db (0) or sc_mask_floor_db (-inf), css.py:223    */
    int32_t mc_mvdr, stitching_loss /* 0 'l1', 1 'mse' */, stitching_input /* 0 'mask', 1 'separation_result' */, normalize_segment_power;
} CssCfgSeconds;
/* css/css.py:144-152 (seconds -> frames, with the reference's own double expressions and truncations) and css.py:341-390
 * (calc_segment_weight: the three trapezoid windows, torch.linspace's float32 evaluation order) -- what
 * notsofar1-challenge_amd/css.py::make_run_cfg does in Python, bit for bit (tests/test_cabi.py).  `windows` receives
 * 3 * segment_frames floats (first | middle | last segment) and out->w_first / w_mid / w_last point into it; `cap` is its size
 * in floats.  CSS_ERR_WEIGHT_WINDOW: css.py:374 "not enough frames to fit weighting window"; CSS_ERR_MASK_FLOOR: css.py:224;
 * CSS_ERR_INVALID_ARG: a segmentation outside 2 .. CSS_MAX_SEGMENT_FRAMES frames with 1 <= hop < segment, or cap too small. */
int css_make_run_cfg(const CssModelDesc* desc, const CssCfgSeconds* cfg, int32_t fs, CssRunCfg* out, float* windows, int64_t cap);

/* ---- the hot path: css/css.py:110 separate_and_stitch -------------------------------------- */
/* pcm_host [n_samples][n_ch] float32 (exactly css/helpers.py:40 load_audio's layout, batch squeezed)
 * -> wav_host [S][n_out].  One H2D of the PCM, one D2H of the waveforms. */
int css_run(css_handle_t h, const float* pcm_host, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg,
            float* wav_host, int64_t wav_capacity_per_stream);
/* A queue of sessions: css_run without the closing host synchronisation.  The call returns once the pass is on the
 * handle's streams; pcm_host must stay valid and wav_host untouched until css_wait(h) returns.  With page-locked buffers
 * (css_host_alloc) consecutive queued passes OVERLAP: pass P's samples cross PCIe (into the half of the sample buffer the
 * pass before last used) while pass P - 1's estimator runs, and P - 1's stitching, synthesis and download (zero-copy
 * into wav_host) run beside P's estimator -- both PCIe legs of a session hide under its neighbours' kernels, results bit
 * for bit those of css_run (tests/test_hip_session.py, bench.py).  With pageable output the passes simply queue up.
 * The call blocks while three queued passes are still unfinished (back-pressure: the host never runs more than three
 * passes ahead of the device).  A queue never mixes the two modes un-drained: when a pageable output follows a page-locked
 * one (or the reverse) the call first waits for the queued passes on the device.  Range rule as css_run: if a queued pass
 * left the split-f16 range, css_wait repeats every pass queued since the last css_wait on the exact float32 kernels (from
 * the caller's buffers, which it still holds) and css_range_status reports it; with css_set_range_fallback(h, 0) css_wait
 * returns CSS_ERR_RANGE instead. */
int css_run_enqueue(css_handle_t h, const float* pcm_host, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg,
                    float* wav_host, int64_t cap);
/* Queued sessions SHARE mask-estimator batches (round 4).  Segments are independent through the network
 * (css/css.py:182-250 carries no state between them) and every kernel of it is batch invariant, so css_run_enqueue merges
 * the segments of consecutive queued sessions -- same segmentation and windows, page-locked output -- into one
 * [segments x T, .] problem for as long as they fit max_batch_segments (exact float32 mode: six 60 s meetings at the bench's
 * 256) and, in the split-f16 mode, CSS_TUNE_SPLIT_BATCH_ROWS token rows (three such meetings): every Linear-layer launch then
 * has M >= 22 k rows instead of 7 k.  Everything outside the estimator stays per session; each
 * session's result is bit for bit its css_run result (tests/test_hip_schedules.py).  A session is accepted (arguments
 * checked, CSS_ERR_* returned at once) and may be held back until its group is full, a session that cannot join arrives, or
 * css_wait / any other call on the handle: LIFETIME -- pcm_host, wav_host AND the recording they describe must stay
 * valid and untouched from css_run_enqueue until css_wait returns, for every session queued in between (css_wait may
 * also re-read them: range rule above); cfg and its windows are copied at the call.
 * css_set_queue_group(h, n): at most n sessions per estimator batch (1 .. 8, default 8; 1 = every session its own pass). */
int css_set_queue_group(css_handle_t h, int max_sessions);
/* Blocks until every pass queued on h has finished (results in their wav_host buffers); CssTimings describe the last. */
int css_wait(css_handle_t h);
/* Blocks until the FIRST n sessions queued since the last css_wait have finished -- their outputs are in host memory -- and
 * leaves the later ones running (round 6: a session loop hands finished sessions to its file writers and keeps enqueueing
 * without ever draining the device; inference_pipeline/inference.py:59-63).  Sessions still held back for company are put on
 * the streams first.  It is NOT a css_wait: the queue's bookkeeping, the CssTimings and -- in CSS_LINEAR_SPLIT_F16 mode -- the
 * range verdict stay with css_wait, which may still repeat a session in float32 from the caller's input buffers; so in that
 * mode the inputs and outputs of a session stay the library's until css_wait, in CSS_LINEAR_EXACT_F32 (no repeat exists) they
 * are the caller's again once css_wait_sessions has returned for it. */
int css_wait_sessions(css_handle_t h, int64_t n);
/* Same with input and output resident in HBM (pcm_dev [n_samples][n_ch], wav_dev [S][n_out]). */
int css_run_device(css_handle_t h, const float* pcm_dev, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg,
                   float* wav_dev, int64_t wav_capacity_per_stream);
/* The same path between the two wav edges: n_ch mono PCM16 planes in host memory -- what
 * css/helpers.py:40 load_audio reads from a session's wav files, scaled by 2^-15 on the device as libsndfile scales
 * them -- to the S separated streams as the PCM16 samples utils/audio_utils.py:37 write_wav would put into
 * sep_stream{i}.wav: peak normalisation x * 0.99 / (max|x| + 1e-7) in float32, then lrint(x * 32767), both on the
 * device.  Half the PCIe bytes of css_run and no host pass over the samples.  peaks_host: NULL or S floats, max|x|
 * of each stream before normalisation. */
int css_run_pcm16(css_handle_t h, const int16_t* const* planes_host, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg,
                  int16_t* wav_pcm16_host, int64_t wav_capacity_per_stream, float* peaks_host);
/* css_run_pcm16 as a QUEUED session (round 6): the session loop of inference_pipeline/inference.py:59-63 -- one css_inference
 * per session: css/css.py:51-107 load_audio -> separate_and_stitch -> write_wav -- with both wav edges on the device AND
 * the sessions sharing mask-estimator batches like css_run_enqueue's (same queue: float and PCM16 sessions may alternate).
 * The n_ch plane pointers are copied at the call; the planes, wav_pcm16_host [S][cap] and peaks_host (NULL or S floats) must
 * stay valid and untouched until css_wait.  With page-locked planes / output (css_host_alloc) the session joins a shared
 * batch and its PCIe legs hide under its neighbours' kernels; with pageable output it runs as a pass of its own.  Results
 * are bit for bit css_run_pcm16's (tests/test_hip_session.py).  frame_len 512 / frame_hop 256 only, like css_run_pcm16. */
int css_run_enqueue_pcm16(css_handle_t h, const int16_t* const* planes_host, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg,
                          int16_t* wav_pcm16_host, int64_t wav_capacity_per_stream, float* peaks_host);
int css_get_timings(css_handle_t h, CssTimings* out);
/* enable != 0: bracket every MFMA GEMM launch of the mask estimator with HIP events on the handle's
 * stream, so that CssTimings.gemm_ms / gemm_launches report the live average launch duration. */
int css_set_profile(css_handle_t h, int enable);
/* With the profile on, the same event pairs bracket EVERY kernel launch of a css_run* pass; their durations summed per
 * kernel family ("stft", "features", "linear_gemm", "attention", "scm", "mvdr_solve", "beamform", "ola_stft", ...), for
 * the roofline figures of the memory-bound stages.  out[cap]; *count = families that ran in the last profiled pass. */
typedef struct CssKernelStat { char name[32]; float ms; int32_t launches; } CssKernelStat;
int css_get_kernel_stats(css_handle_t h, CssKernelStat* out, int32_t cap, int32_t* count);
/* Arithmetic of the Conformer's Linear layers (torch.nn.Linear in conformer.py:49-53,139-142,206,285):
 *   CSS_LINEAR_EXACT_F32 (default)  float32 operands on the float32 matrix instruction (bit-identical to an fmaf loop over
 *                                   k): the reference's own operand precision (its Linear layers run in float32);
 *   CSS_LINEAR_SPLIT_F16 (opt-in)   operands carried as hi + 2^-11 lo float16 pairs (22 significant bits), three f16 MFMAs per
 *                                   product, float32 accumulation: ~2x the throughput, operands NARROWER than the reference's.
 * A new handle is in CSS_LINEAR_EXACT_F32 (round 6; css_inference / separate_and_stitch / HipSeparator never leave it unless
 * asked to: HipSeparator(..., linear_mode="split_f16")).  CSS_LINEAR_SPLIT_F16 is refused for a model with a weight outside the
 * float16 range.  May be switched between runs; no environment variable is read. */
enum css_linear_mode { CSS_LINEAR_SPLIT_F16 = 0, CSS_LINEAR_EXACT_F32 = 1 };
int css_set_linear_mode(css_handle_t h, int mode);
int css_get_linear_mode(css_handle_t h);   /* css_linear_mode, or a negative css_status */
/* Operand range of CSS_LINEAR_SPLIT_F16: |x| <= 65504 (float16).  Nothing is clamped: a larger activation becomes
 * inf / NaN, the GEMM that consumes it raises a device flag, and css_run* then repeats the whole pass on the exact
 * float32 kernels (enable = 1, the default) or returns CSS_ERR_RANGE (enable = 0).  css_range_status: passes repeated
 * so far, and whether the last pass was one.  css_check_range does the same test after a staged (css_stage_*) run.
 * A model with a WEIGHT outside the range cannot leave CSS_LINEAR_EXACT_F32 (css_set_linear_mode returns CSS_ERR_RANGE). */
int css_set_range_fallback(css_handle_t h, int enable);
int css_range_status(css_handle_t h, int64_t* fallbacks, int32_t* last_hit);
int css_check_range(css_handle_t h);
/* torch.nn.Linear (conformer.py:49-53,139-142,206,285) on caller data, y[M][N] = x[M][K] w[N][K]^T + bias[N] (bias may
 * be NULL; K % 32 == 0), through one GEMM kernel of the path: kernel 0 = split-f16 operands, weights tile-major and read
 * straight into the matrix cores (the encoder's Linear layers); 1 = split-f16 operands, both through LDS (mask head,
 * inverse transform); 2 = exact float32.  layout: 0 = the launcher's choice, else a tile layout to force (kernel 0:
 * 32 / 64 / 96 / 4 / 128; kernels 1, 2: 8 / 4 / 64).  For unit tests of the arithmetic; not on the hot path. */
int css_linear_host(css_handle_t h, const float* x, const float* w, const float* bias, int32_t M, int32_t N, int32_t K,
                    int32_t kernel, int32_t layout, float* y);
int css_get_plan(css_handle_t h, CssPlan* out);

/* ---- stages (each replaces one reference function; state lives in the handle) -------------- */
/* Begin a session: upload + deinterleave PCM, fix the plan.  (css.py:141-171) */
int css_begin(css_handle_t h, const float* pcm, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg, int pcm_is_device);
/* css_begin for a rank that owns a slice of a long meeting held in host memory: the plan is that of the whole
 * recording, but only samples [s_lo, s_hi) of pcm_host (which still points at sample 0) cross PCIe -- the samples of
 * the frames this rank will pass to css_stage_stft_range. */
int css_begin_range(css_handle_t h, const float* pcm_host, int64_t n_samples, int32_t n_ch, const CssRunCfg* cfg,
                    int64_t s_lo, int64_t s_hi);
/* Further samples [s_lo, s_hi) of the same host recording for the session css_begin_range opened (pcm_host again points
 * at sample 0; page-locked memory makes the copy asynchronous).  They cross PCIe on the handle's copy stream while the
 * stages of the samples already there run; css_stage_stft_range waits for exactly the pieces its frames read.  The level
 * (CSS_BUF_LEVEL) is complete once every piece's frames have been transformed. */
int css_upload_range(css_handle_t h, const float* pcm_host, int64_t s_lo, int64_t s_hi);
/* ConformerCssWrapper.stft (conformer_wrapper.py:106, feature.py:88) over the whole recording. */
int css_stage_stft(css_handle_t h);
/* Same for frames [t_lo, t_hi) only (a rank that owns a slice of the meeting transforms just the
 * frames its segments read). */
int css_stage_stft_range(css_handle_t h, int64_t t_lo, int64_t t_hi);
/* ConformerCssWrapper.separate (conformer_wrapper.py:79): features + Conformer for segments [lo, hi). */
int css_stage_masknet(css_handle_t h, int64_t seg_lo, int64_t seg_hi);
/* make_mvdr (mvdr_util.py:5) + mask floor/multiply (css.py:222-227) for segments [lo, hi). */
int css_stage_mvdr(css_handle_t h, int64_t seg_lo, int64_t seg_hi);
/* Raw PIT costs (losses.py:50-71) for boundaries [lo, hi) (boundary b joins segments b, b+1). */
int css_stage_pit_costs(css_handle_t h, int64_t b_lo, int64_t b_hi);
/* Permutation scan on device over all boundaries (css.py:266-285). */
int css_stage_pit_scan(css_handle_t h);
/* Weighted overlap-add + activity gating (css.py:254-312) for frames [t_lo, t_hi). */
int css_stage_stitch(css_handle_t h, int64_t t_lo, int64_t t_hi);
/* ConformerCssWrapper.istft (conformer_wrapper.py:131, feature.py:138) for output frames
 * [t_lo, t_hi): samples [t_lo*hop, (t_hi)*hop) (+ the tail when t_hi == mix_frames). */
int css_stage_istft(css_handle_t h, int64_t t_lo, int64_t t_hi);
/* The two halves of css_stage_stitch, for frame-sharded runs that exchange the activity bits in between:
 * masks:  overlap-add of the masks, mean over frequency, threshold (css.py:295,299,303-304) on [t_lo, t_hi);
 * gate:   dilate/erode (css.py:305-308; reads CSS_BUF_ACT_B `dilation+erosion` frames to either side),
 *         overlap-add of the spectra, gating (css.py:294,298,312) on [t_lo, t_hi). */
int css_stage_stitch_masks(css_handle_t h, int64_t t_lo, int64_t t_hi);
int css_stage_stitch_gate(css_handle_t h, int64_t t_lo, int64_t t_hi);
/* Inverse transform of frames [t_lo, t_hi) ONLY, written to a caller-owned device shard
 * shard_dev [S][shard_ld]: column hop*(q - t_lo) + r for output blocks q in [t_lo, t_hi].  The first and
 * last block hold one frame's contribution each; adding the overlapping blocks of adjacent shards
 * reproduces css_stage_istft bit for bit (a two-term float sum commutes). */
int css_stage_istft_partial(css_handle_t h, int64_t t_lo, int64_t t_hi, float* shard_dev, int64_t shard_ld);
/* The seam for ANY frame geometry (round 6; ExtractorCfg.frame_len / frame_hop, feature.py:19-45,138-167).  With
 * ovl = ceil(frame_len / frame_hop) > 2 frames over an output sample the overlap-add is an ORDERED float sum (oldest frame
 * first) and partial blocks of two ranks do not compose bit for bit; the ranks exchange the synthesis rows of their last
 * ovl - 1 frames instead and the receiver runs the single-GPU overlap-add over them:
 *   css_stage_synthesis    frames [t_lo, t_hi) of the gated spectra -> their rows G[s][t][frame_len] (feature.py:162 without
 *                          the overlap-add); after css_stage_stitch_gate of those frames;
 *   css_stage_seam_rows    write == 0: G rows of frames [t_lo, t_hi) -> rows_dev [S][t_hi - t_lo][frame_len] (the piece a rank
 *                          sends); write == 1: the reverse (a left neighbour's rows take their place in G);
 *   css_stage_overlap_add  output blocks [q_lo, q_hi) (one hop each; q_hi <= mix_frames - 1 + ovl) from the frames
 *                          [f_lo, f_hi) of G into out_dev [S][out_ld], block q at column (q - out_q0) * hop. */
int css_stage_synthesis(css_handle_t h, int64_t t_lo, int64_t t_hi);
int css_stage_seam_rows(css_handle_t h, int64_t t_lo, int64_t t_hi, float* rows_dev, int32_t write);
int css_stage_overlap_add(css_handle_t h, int64_t f_lo, int64_t f_hi, int64_t q_lo, int64_t q_hi, float* out_dev, int64_t out_ld,
                          int64_t out_q0);
/* Exchange 3 of a segment-sharded meeting (parallel.py): gathered_dev [world][S][shard_ld] holds every rank's
 * css_stage_istft_partial shard (rank r: output blocks t_lo[r] .. t_hi[r]); out_dev [S][out_ld] receives the stitched
 * streams -- a rank's inner blocks as they are, the block at a seam as the sum of its two neighbours' partial blocks. */
int css_stage_join_shards(css_handle_t h, const float* gathered_dev, int32_t world, int64_t shard_ld, const int64_t* t_lo,
                          const int64_t* t_hi, float* out_dev, int64_t out_ld);
int css_sync(css_handle_t h);

/* Selects the feature-extractor options of the handle's model (see CssFeatureCfg); num_pairs must agree with the model's
 * in_features, the pair indices with its microphones.  Single-channel models take the spectral options only. */
int css_set_feature_options(css_handle_t h, const CssFeatureCfg* cfg);

/* The analysis window of the handle's model: ExtractorCfg.window (conformer_wrapper.py:24), one of the two init_kernel
 * builds (css_with_conformer/executor/feature.py:19-45): 'hann' (every shipped model; the state after css_create) or
 * 'sqrt_hann' -- the square root of the float32 Hann window with the kernel divided by 0.5 sqrt(N N / hop) = 16.  The
 * synthesis transform is not affected (the reference builds its iSTFT without `window`: feature.py:422-425). */
#define CSS_WINDOW_HANN 0
#define CSS_WINDOW_SQRT_HANN 1
int css_set_analysis_window(css_handle_t h, int32_t window);

/* Separator-protocol helpers operating on caller data (host pointers):
 * stft: pcm [n][C] -> X planes [C][2F][T] (T = stft_frames, tightly packed). */
int css_stft_host(css_handle_t h, const float* pcm, int64_t n_samples, int32_t n_ch, float* x_planes, int64_t t_frames);
/* separate (conformer_wrapper.py:79): X planes [C][2F][B*T] holding B independent segments of T frames
 * back to back along time -> masks [(S+1)F][B*T] (row k*F + f, column b*T + t). */
int css_separate_host(css_handle_t h, const float* x_planes, int32_t batch, int32_t t_frames, float* masks);
/* ConformerCssWrapper.forward (conformer_wrapper.py:58-77: stft -> separate) for a batch of equally long clips, fused
 * on the device -- the validation forward of the reference's training loop (train.py:529 eval_model).
 * pcm_host [batch][n_samples][n_ch] -> masks_host [(S+1) F][batch * T'], T' = (n_samples - frame_len) / hop + 1
 * (2 <= T' <= CSS_MAX_SEGMENT_FRAMES), clip b in columns [b T', (b+1) T'); mask k of bin f in row k F + f (speakers first). */
int css_forward_host(css_handle_t h, const float* pcm_host, int32_t batch, int64_t n_samples, int32_t n_ch, float* masks_host);
/* The validation loss of the reference's training loop for a batch of equally long clips (css/training/train.py:411-481
 * _calc_loss as train.py:529 eval_model calls it; no backward pass): forward as css_forward_host, |STFT| of microphone 0
 * of the mixture and of the ground truths, PIT over the speaker outputs (css/training/losses.py:50-97 PitWrapper: the
 * assignment of least mean loss), the noise loss, *loss = mean_b(spk_loss[b] + noise_weight * noise_loss[b]).
 *   mix_host [batch][n][n_ch];  gt_spk_host [batch][S][n], gt_noise_host [batch][n]: the ground truths AT microphone 0
 *   (train.py:421-425 slices them out of the batch's [.., Mics, ..] tensors);
 *   loss_name 0 = 'masked_mag' (train.py:449), 1 = 'mask' (:464);  base_loss 0 = l1, 1 = mse (losses.py:100-106);
 *   clip_gt: TrainCfg.clip_gt_to_mixture (train.py:431).  spk_loss / noise_loss [batch], perms [batch][S] may be NULL. */
int css_validation_loss_host(css_handle_t h, const float* mix_host, const float* gt_spk_host, const float* gt_noise_host,
                             int32_t batch, int64_t n_samples, int32_t n_ch, int32_t loss_name, int32_t base_loss, int32_t clip_gt,
                             float noise_weight, float* spk_loss, float* noise_loss, int32_t* perms, float* loss);
/* Downstream hand-off.  The reference writes the streams to wav files for Whisper to read back
 * (asr/asr.py:58,73-74) and notes "potential optimization: drop silent parts to save ASR compute" (css/css.py:313).  After a
 * css_run_device pass, stream `stream` of the caller's device buffer wav_dev [S][wav_ld] is cut to the frames the activity
 * gate kept (css.py:303-312), each run widened by pad_frames, runs merged: regions_host [n_regions][2] are the sample
 * ranges kept -- the time map back to the meeting -- and mel_host [n_mels][*n_mel_frames] receives Whisper's input
 * features of their concatenation (whisper/audio.py log_mel_spectrogram: reflect-padded 400-point Hann STFT, hop 160,
 * power, slaney mel bank with n_mels = 80 or 128, log10 floored at 1e-10, max - 8 clamp, (x + 4) / 4), computed on the
 * device.  drop_silence = 0: one region, the whole stream.  Whisper is not under the reference tree: the algorithm is held to
 * transformers.WhisperFeatureExtractor (tests/test_oracle_whisper_pin.py), not to openai-whisper itself. */
int css_handoff_logmel(css_handle_t h, const float* wav_dev, int64_t wav_ld, int32_t stream, int32_t n_mels, int32_t pad_frames,
                       int32_t drop_silence, float* mel_host, int64_t mel_capacity_frames, int64_t* n_mel_frames,
                       int64_t* regions_host, int32_t max_regions, int32_t* n_regions);
/* istft: Y [B][2F][T] planes (Re rows then Im rows, time fastest) -> wav [B][(T-1)*hop + frame_len]. */
int css_istft_host(css_handle_t h, const float* y_planes, int32_t batch, int64_t t_frames, float* wav);

/* ---- buffer access for stage-level parity tests -------------------------------------------- */
/* dims[0..3] (unused = 1) and element size of a buffer in the current session. */
int css_buffer_dims(css_handle_t h, int which, int64_t dims[4], int32_t* elem_bytes);
int css_read_buffer(css_handle_t h, int which, void* host, int64_t nbytes);
int css_write_buffer(css_handle_t h, int which, const void* host, int64_t nbytes);
/* Device address of a buffer (for zero-copy wrapping, e.g. by torch for the RCCL all-gather). */
int css_buffer_devptr(css_handle_t h, int which, void** out);

/* ---- the exchanges of the sharded path without Python ------------------------------------------------
 * parallel.py shards a long meeting by sliding-window segment over the GPUs of a node and stitches with three all-gathers
 * (raw PIT costs, thresholded activity, seam blocks / waveform shards; css.py:110-338 itself has no collective -- it is the
 * single-process loop these replace).  Python reaches them through torch.distributed; a host in another language uses
 * these: one RCCL communicator per handle, created from a unique id the caller distributes over its own channel (what
 * ncclGetUniqueId / ncclCommInitRank expect), every collective enqueued on the handle's stream behind its kernels.
 * librccl.so is loaded on first use (dlopen): the library itself carries no link-time dependency on it, and every entry
 * point returns CSS_ERR_STATE with a message when it cannot be loaded. */
#define CSS_COMM_ID_BYTES 128
int css_comm_unique_id(void* id_out);   /* rank 0: CSS_COMM_ID_BYTES bytes to hand to every rank */
int css_comm_init(css_handle_t h, const void* id, int32_t nranks, int32_t rank);
int css_comm_destroy(css_handle_t h);
/* nranks / rank / HIP device / RCCL version code of the handle's communicator (any pointer may be NULL) */
int css_comm_info(css_handle_t h, int32_t* nranks, int32_t* rank, int32_t* device, int32_t* rccl_version);
/* recv_dev [nranks][bytes_per_rank] <- every rank's send_dev [bytes_per_rank]; device pointers; on the handle's stream */
int css_comm_all_gather(css_handle_t h, const void* send_dev, void* recv_dev, int64_t bytes_per_rank);
/* (The pieces themselves -- which rows of CSS_BUF_PIT_COST / CSS_BUF_ACTIVITY a rank owns, the seam block of its shard -- are
 * parallel.py's plan; a host in another language addresses them through css_buffer_devptr.  parallel.HipShardBackend takes this
 * route with comm="cabi": the same driver, RCCL reached through these entry points instead of torch.distributed.) */

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* CSS_MI355_H */

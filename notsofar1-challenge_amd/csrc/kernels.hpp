// Internal launch interface of the gfx950 kernels (one declaration per kernel family).
// Everything here is device-pointer based and asynchronous on the given stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace css {

// ------------------------------------------------------------------------------------------------
// gemm.hip -- fp32 MFMA GEMM  C[m][n] = epilogue( sum_k A[m*lda + k] * B[n*ldb + k] )
// Both operands are K-contiguous ("NT"): A is an activation / DFT matrix, B a weight [out][in] or a
// strided view of frames.  K must be a multiple of 32; M, N arbitrary.
// ------------------------------------------------------------------------------------------------
enum GemmAct : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };

struct GemmArgs {
    const float* A; int64_t lda; int64_t strideA;   // batch stride (0 = shared)
    const float* B; int64_t ldb; int64_t strideB;
    float* C; int64_t ldc; int64_t strideC;
    int M, N, K, batch;
    const float* bias;      // nullptr or [N] (bias_along_m == 0) / [M] (bias_along_m == 1)
    int bias_along_m;
    int act;                // GemmAct
    const float* residual;  // nullptr or [M][ldr]: C = residual + alpha * (acc + bias)
    int64_t ldr;
    float alpha;            // only with residual
    // split-f16 operands (split_f16.hpp): A and B are split matrices addressed as float matrices (lda / ldb = their
    // padded K); three f16 MFMAs per product, float32 accumulation.  split_out = n > 0: columns [0, n) of C are
    // written in the split format for the next consumer (n % 32 == 0), the remaining columns as float32.
    int split_in;
    int split_out;
    // b_tiled (with split_in): B is a weight matrix in the tile-major split layout of gemm_split_wd.hip
    // (launch_split_convert_tiled); batch must be 1 and N % 32 == 0.
    int b_tiled;
    // b_frag32 (exact float32, gemm_f32.hip): B is a weight in float32 FRAGMENT order (launch_f32_fragments); batch 1, N % 32 == 0
    int b_frag32;
    const float* B_rows;   // with b_frag32: the same weight row-major (ld = ldb), for the launches gemm_f32.hip does not take
                           // (an operand of 2 GiB and more, a misaligned A): launch_gemm then runs gemm.hip's kernel on it
    // hint: other kernels run beside this launch (two-lane mask estimator): prefer 4-wave 64-row tiles, two of which
    // -- from different launches -- share a CU, over one 8-wave 128-row tile per CU
    int concurrent;
    int nt_store;   // epilogue stores bypass the caches (nontemporal)
    // weights-direct kernel only: columns n < 2 * frag_D (the q and k projections of the attention, D = heads * 64) leave
    // in the attention kernel's operand order instead of row-major C (encoder.hip qk fragment layout); rows are tokens
    // of segments of frag_T frames, frag_invT = 1.0f / frag_T
    float* frag_out;
    int frag_D, frag_T, frag_heads;
    float frag_invT;
    int tile_rows;         // weights-direct kernel: 0 = choose, else 32 / 64 / 96 / 128 (8 waves) / 4 (128 rows, 4 waves) / 65 (64 rows, 64-bit global loads)
    int narrow_epilogue;   // tools: keep the 4-byte-per-lane epilogue of the weights-direct kernel (A/B timing)
    // split kernels: *range_flag |= 1 when a finished accumulator is not finite -- an operand left the split-f16 range
    // (split_f16.hpp: nothing is clamped); checked here, in the consumer, because a ReLU downstream would launder a NaN
    unsigned int* range_flag;
    int layout;            // LDS-staged kernels (gemm.hip, gemm_split.hip): 0 = choose, else 8 / 4 (128-row tiles, 8 / 4 waves) / 64;
                           // exact float32 only: 1 = gemm_f32.hip with its balanced plan (what 0 chooses), 11 .. 14 = gemm_f32.hip with
                           // every tile 32 / 64 / 96 / 128 rows, 2 = gemm.hip's kernel with its own choice among 8 / 4 / 64
    // LDS-staged kernels: an XCD's consecutive tiles walk the ROW tiles of one column panel (they share the B panel) instead
    // of the column tiles of one row panel: for launches whose B operand is the large one.  Same tiles, same bits.
    int m_fastest;
    // weights-direct kernel: the result leaves transposed, C^T[n][m] at C + n * ldc + m (column bias and activation only;
    // no residual, no split output)
    int c_transposed;
};
void launch_gemm(const GemmArgs& g, hipStream_t s);        // dispatches on g.split_in
// gemm_f32.hip: the exact float32 product on four independent 4-wave blocks per CU with balanced tile heights (same bits as
// gemm.hip's kernel); forced_tm = 1..4: every tile 32 * forced_tm rows, 0: the balanced plan.  false: not launched (an
// operand beyond the 32-bit buffer offsets, K % 16): the caller takes gemm.hip's kernel
bool launch_gemm_f32(const GemmArgs& g, hipStream_t s, int forced_tm);
// float32 W [N][K] (row stride ld_src; N % 32 == 0, K % 8 == 0) -> the fragment order GemmArgs::b_frag32 names, N * K floats
void launch_f32_fragments(const float* src, int64_t ld_src, float* dst, int N, int K, hipStream_t s);
void launch_gemm_split(const GemmArgs& g, hipStream_t s);  // gemm_split.hip
void launch_gemm_split_wd(const GemmArgs& g, hipStream_t s);  // gemm_split_wd.hip (g.b_tiled)
// float32 W [N][K] (row stride ld_src, K % 32 == 0) -> tile-major split-f16 weights, (N rounded up to 32) * K floats
void launch_split_convert_tiled(const float* src, int64_t ld_src, float* dst, int N, int K, hipStream_t s);
// float32 [rows][K] (row stride ld_src) -> split-f16 [rows][Kp] (Kp % 32 == 0, zero padded past K)
void launch_split_convert(const float* src, int64_t ld_src, float* dst, int64_t rows, int K, int Kp, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// encoder.hip -- the non-GEMM pieces of the Conformer
// ------------------------------------------------------------------------------------------------
// y = LN(x) * w + b over rows of length D (D % 256 == 0, D <= 1024); optional ReLU.  y (float32) and y_split
// (the same rows as a split-f16 GEMM operand, split_f16.hpp) are each optional.
void launch_layernorm(const float* x, float* y, float* y_split, const float* w, const float* b, int rows, int D,
                      int relu, hipStream_t s);
// y = LN(x; w1, b1) (float32, may alias x), then z / z_split = LN(y; w2, b2) (each optional), one pass over the rows
void launch_layernorm2(const float* x, float* y, const float* w1, const float* b1, float* z, float* z_split,
                       const float* w2, const float* b2, int rows, int D, hipStream_t s);
// conv-module front half: u = LN(x); z = (pw[0]*u + pw[1]) * sigmoid(pw[2]*u + pw[3])
void launch_ln_glu(const float* x, float* z, const float* w, const float* b, const float* pw, int rows, int D,
                   hipStream_t s);
// conv-module back half: depthwise conv over time inside each segment (zero padded), eval-BatchNorm,
// ReLU, scalar pointwise conv, residual:  h += pw[4] * relu((conv(z) + dw_b) * alpha + beta) + pw[5]
void launch_dwconv(const float* z, float* h, const float* dw_wt, const float* dw_b, const float* bn_alpha,
                   const float* bn_beta, const float* pw, int nseg, int T, int D, int taps, hipStream_t s);
// the whole conv module in one kernel: x_out = x_in + conv_module(x_in); x_out must not alias x_in.  Returns false when
// (D, taps) is not covered and nothing was launched (use launch_ln_glu + launch_dwconv).  z / z_split (either may be
// null) = LayerNorm(x_out; ln2_w, ln2_b) as float32 / split-f16 rows: the LayerNorm of the module that follows.
bool launch_conv_module(const float* x_in, float* x_out, const float* ln_w, const float* ln_b, const float* pw,
                        const float* dw_wt, const float* dw_b, const float* bn_alpha, const float* bn_beta,
                        const float* ln2_w, const float* ln2_b, float* z, float* z_split, int nseg, int T,
                        int D, int taps, hipStream_t s);
// relative-position multi-head attention: qkv [tokens][3D] -> ctx [tokens][D] (float32, or split-f16 rows).
// qk_split: the q and k columns of qkv and the position rows are split-f16 (scores on the f16 matrix cores with
// float32-grade accuracy); v is float32 either way.  pe_frag: the relative-position table rearranged for this T by
// launch_pe_fragments from the row-major table of the same arithmetic (float32 or split-f16 rows of d_k = 64):
// pe_fragment_tiles(T) tiles of 32 rows x 64, 2048 floats each.
inline int pe_fragment_tiles(int T) { return 2 * ((T + 31) / 32); }
void launch_pe_fragments(const float* pe, float* frag, int T, int maxlen, int split, hipStream_t s);
// qk_frag (split mode only, may be null): q and k in operand order, written by the QKV GEMM (GemmArgs::frag_out):
// float4 index ((((seg * H + head) * ceil(T / 32) + tile) * 2 + which) * 8 + chunk) * 64 + lane, which = 0 q / 1 k;
// qk_fragment_floats(nseg, T, H) floats.  The q and k columns of qkv are then not read.
inline int64_t qk_fragment_floats(int64_t nseg, int T, int H) { return nseg * H * ((T + 31) / 32) * 2 * 8 * 64 * 4; }
void launch_relpos_attention(const float* qkv, const float* qk_frag, const float* pe_frag, float* ctx, int nseg, int T, int D,
                             int H, int maxlen, int qk_split, int split_out, hipStream_t s);
// segments beyond 512 frames (encoder.hip relpos_attn_long_kernel): qkv = float32 rows [token][3 D], pe = the float32
// table [2 maxlen][64]; false: the head width is not 64 or one query's rows do not fit the LDS
bool launch_relpos_attention_long(const float* qkv, const float* pe, float* ctx, int nseg, int T, int D, int H, int maxlen,
                                  int split_out, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// stft.hip -- the analysis transform as an LDS-staged FFT (frame_len 512, hop 256, 257 bins)
// ------------------------------------------------------------------------------------------------
size_t stft_table_floats();
void stft_build_tables(float* host, int window = 0);   // window (0: hann, 1: sqrt_hann / 16) | radix-4 twiddles | real-spectrum twiddles
// frames 0 .. nf-1 of C channels (channel c's first sample at x + c * x_stride, frame t at sample 256 t) -> planes
// out[(c * 514 + r) * row_ld + t], r = f (Re) / 257 + f (Im).  false: the kernel's LDS could not be reserved.
// phase != nullptr: also the PHASE planes phase[(c * 257 + f) * row_ld + t] = css_phase_of(Re, Im) -- the angle the IPD
// features are made of (feature.py:214-221), formed once per frame here instead of once per segment and microphone pair in
// the feature kernel (a frame lies in two segments, microphone 0 in six pairs: 18 atan2 per feature element became 6 + 7/2).
bool launch_stft_fft(const float* x, int64_t x_stride, int C, int64_t t_lo, int64_t t_hi, const float* tables, float* out,
                     int64_t row_ld, hipStream_t s, float* phase = nullptr);

// XCD-aware work order: workgroup L of a 1-D grid runs on XCD L % 8 (round robin), and each XCD has its own L2.  Work items
// that read the same operands should therefore be neighbours IN ONE XCD: every XCD takes a contiguous range of the n items,
// consecutive workgroups of an XCD consecutive items of its range.  (A bijection of [0, n) for every n.)
__device__ __forceinline__ int css_xcd_item(int L, int n) {
    const int q = n >> 3, rem = n & 7, xcd = L & 7;
    return xcd * q + (xcd < rem ? xcd : rem) + (L >> 3);
}

// Phase of a spectrum value as the reference reaches it.  An exactly real negative bin (DC / Nyquist have Im == +0 by
// construction of the transform) goes through polar() -> angle() in the reference (conformer_wrapper.py:124,94), which
// maps it to the float32 value just inside -pi; the side of the atan2 branch cut of the IPD feature depends on that
// (DESIGN.md "Numerical hazards").  CSS_PHASE_NEG_REAL is that value.
#define CSS_PHASE_NEG_REAL (-3.14159250259399414f) /* 0xC0490FDA */
__device__ __forceinline__ float css_phase_of(float re, float im) {
    return (im == 0.f && re < 0.f) ? CSS_PHASE_NEG_REAL : atan2f(im, re);
}

// ------------------------------------------------------------------------------------------------
// frontend.hip -- PCM layout, features, inverse-transform overlap-add
// ------------------------------------------------------------------------------------------------
// samples [i_lo, i_hi) of every channel: pcm [n][C] -> pcm_cm [C][n_pad] (zeros past n)
// (split_out: channel rows as split-f16 GEMM operands, n_pad % 32 == 0)
void launch_deinterleave(const float* pcm, float* pcm_cm, int64_t n, int C, int64_t n_pad, int64_t i_lo, int64_t i_hi,
                         int split_out, hipStream_t s);
// *peak = max(*peak, max |x|) as float bits (int16 samples scaled by 2^-15): the recording's level (split_f16.hpp level_gain)
void launch_pcm_peak_f32(const float* x, int64_t count, unsigned int* peak, hipStream_t s);
void launch_pcm_peak_i16(const int16_t* x, int64_t count, unsigned int* peak, hipStream_t s);
// wav edges on the device: C mono PCM16 planes [C][n] -> sample-major float32 [n][C]; and peak-normalised PCM16
// encoding of the S output streams (peak_bits: S words of scratch, holds max|x| as float bits afterwards)
void launch_pcm16_to_float(const int16_t* planes, float* pcm, int64_t n, int C, hipStream_t s);
void launch_pcm16_to_channel_major(const int16_t* planes, float* pcm_cm, int64_t n, int C, int64_t n_pad, int64_t i_lo,
                                   int64_t i_hi, hipStream_t s);
void launch_encode_pcm16(const float* wav, int S, int64_t n, unsigned int* peak_bits, int16_t* out, int64_t out_ld,
                         hipStream_t s);
// features for segments [seg_lo, seg_lo + nseg): X planes -> feat [nseg*T][Kp] (bias/scale folded); float32 rows
// or split-f16 rows (split_f16.hpp; the padding columns are never written and must be zero)
// Segments beyond 512 frames (8 s) take kernels written for any length (features_long_kernel, relpos_attn_long_kernel,
// scm_long_kernel).  CSS_FORCE_LONG_PATH=1 in the environment sends EVERY segment length through them: the tests hold
// them to the fixtures of the fast kernels that way.
bool css_force_long_path();

struct FeatOpts {   // CssFeatureCfg in kernel-argument form
    int log_mag, mvn, ipd_norm, ipd_version, ipd_cos, num_pairs;
    unsigned char pair_l[16], pair_r[16];
};
// PH: the phase planes launch_stft_fft wrote beside X (same T_ld), or nullptr: the kernel forms the phases itself (same
// function on the same values: the features are the same bits either way)
void launch_features(const float* X, int64_t T_ld, int64_t stft_frames, int C, int F, float* feat, int Kp,
                     const float* in_bias, const float* in_scale, int64_t seg_lo, int nseg, int T, int hop,
                     int split_out, const FeatOpts& opts, hipStream_t s, const float* PH = nullptr);
// out[b][hop*(q - out_q0) + r] = G[b][q][r] + G[b][q-1][hop + r] for output blocks q in [q_lo, q_hi), taking
// only frames in [f_lo, f_hi) (frame_len == 2*hop); out has row stride out_ld
// level (may be null): the samples are multiplied by 1 / level_gain(level) (undoes the scaling of the split spectra rows)
void launch_wave_ola(const float* G, float* out, int B, int64_t T_frames, int hop, int frame_len, int64_t q_lo, int64_t q_hi,
                     int64_t f_lo, int64_t f_hi, int64_t out_ld, int64_t out_q0, const unsigned int* level, hipStream_t s);
// gathered [world][S][ld] waveform shards (rank r: output blocks t_lo[r] .. t_hi[r]) -> out [S][out_ld] (world <= 64)
void launch_join_shards(const float* gathered, int64_t ld, const int64_t* t_lo, const int64_t* t_hi, int world, int S, int hop,
                        int64_t n_out, float* out, int64_t out_ld, hipStream_t s);
// [B][2F][T] planes -> [B][T][KIp] rows for the inverse GEMM
void launch_planes_to_rows(const float* planes, float* rows, int B, int F2, int64_t T, int KIp, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// handoff.hip -- active regions of a separated stream -> Whisper log-mel features on the device
// ------------------------------------------------------------------------------------------------
void handoff_build_dft(float* m);                 // [402][416]
void handoff_build_mel(float* w, int n_mels);     // [n_mels][201]
void launch_handoff_gather(const float* wav, const int64_t* regions, const int64_t* offs, int nr, int64_t n_act, float* out,
                           int64_t total, hipStream_t s);
void launch_handoff_mel(const float* spec, int64_t ld, int64_t nfr, const float* w, int n_mels, float* mel, int* gmax, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// loss.hip -- validation loss of the training loop (train.py:411 _calc_loss): S x S base-loss sums + the noise term
// ------------------------------------------------------------------------------------------------
int val_loss_chunks(int F);
// partial: B * val_loss_chunks(F) * 16 doubles (entries a * 3 + s and [15 -> 9]: see loss.hip)
void launch_val_loss(const float* X, const float* masks, const float* G, int B, int T, int F, int S, int loss_name, int base,
                     int clip, double* partial, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// mvdr.hip -- WTA masks, spatial covariance, MVDR solve, beamform + mask
// ------------------------------------------------------------------------------------------------
struct MvdrArgs {
    const float* X; int64_t T_ld; int64_t stft_frames; int C; int F;
    const float* masks; int64_t mask_ld;   // [(S+1)F][mask_ld]; column = segment*T + t
    int S; int T; int hop;
    int64_t seg_lo; int nseg;
    const uint8_t* wta_override;           // nullptr or [segments][F][T]
    double* scm;                           // [segments][S+1][F][49]
    double* bfw;                           // [segments][S][F][C][2]
    float* sep;                            // [segments][S][F][T][2]
    float mask_floor;
    int use_mvdr;
};
bool launch_scm(const MvdrArgs& a, hipStream_t s);   // false: the LDS of a long-segment launch could not be reserved
void launch_mvdr_solve(const MvdrArgs& a, hipStream_t s);
// the same for n sessions' argument sets in one launch (a queue group's sessions; each result is the per-session launch's)
constexpr int MVDR_MULTI_MAX = 8;
void launch_mvdr_solve_multi(const MvdrArgs* a, int n, hipStream_t s);
void launch_beamform(const MvdrArgs& a, hipStream_t s);
// optional power normalisation (css.py:233-247): scales sep of each segment in place
void launch_segment_power_norm(const MvdrArgs& a, double* scratch, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// stitch.hip -- PIT costs, permutation scan, weighted overlap-add, activity gate
// ------------------------------------------------------------------------------------------------
struct StitchArgs {
    const float* masks; int64_t mask_ld;
    const float* sep;
    int S; int F; int T; int hop;
    int64_t num_segments; int64_t T_long;
    const float* w_first; const float* w_mid; const float* w_last;  // device [T]
    const int32_t* perms;      // [segments][S]
    float* mask_st;            // [S][F][T_long]
    float* activity;           // [S][T_long]
    uint8_t* act_b; uint8_t* act_tmp; uint8_t* act_final;  // [S][T_long]
    float activity_th; int dilation; int erosion;
    float* Y; int KIp;         // [S][T_long][KIp]
    int y_split;               // Y rows as split-f16 GEMM operands (split_f16.hpp) instead of float32
    const unsigned int* level; // with y_split: the rows are scaled by level_gain(level) (split_f16.hpp); null = 1
};
// scratch: pit_cost_scratch_bytes(total boundaries) bytes, indexed by absolute boundary
size_t pit_cost_scratch_bytes(int64_t n_boundaries);
void launch_pit_costs(const StitchArgs& a, int loss, int input, int64_t b_lo, int64_t b_hi, double* scratch, double* costs,
                      hipStream_t s);
// every boundary of n sessions (a queue group: same S) in one pair of launches; scratch[i] / costs[i] as above per session
constexpr int PIT_MULTI_MAX = 8;
void launch_pit_costs_multi(const StitchArgs* a, double* const* scratch, double* const* costs, int n, int loss, int input, hipStream_t s);
// permutations of segments b_lo + 1 .. b_hi from the raw costs of boundaries [b_lo, b_hi), continuing from the
// permutation of segment b_lo already in perms (b_lo == 0: the identity, written here)
void launch_pit_scan(const double* costs, int64_t b_lo, int64_t b_hi, int S, int32_t* perms, hipStream_t s);
void pit_scan_host(const double* costs, int64_t n_boundaries, int S, int32_t* perms);
void launch_ola_masks(const StitchArgs& a, int64_t t_lo, int64_t t_hi, hipStream_t s);
void launch_morphology(const StitchArgs& a, int64_t t_lo, int64_t t_hi, hipStream_t s);
void launch_ola_stft(const StitchArgs& a, int64_t t_lo, int64_t t_hi, hipStream_t s);

}  // namespace css

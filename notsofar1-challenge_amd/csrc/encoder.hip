// Non-GEMM pieces of the Conformer mask estimator (css/css_with_conformer/nnet/conformer.py):
// LayerNorm (+ReLU / +scalar GLU), the depthwise-conv module and relative-position attention.
#include "kernels.hpp"
#include "split_f16.hpp"

namespace css {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Sum over the 64 lanes, every lane gets the total: four DPP steps inside each row of 16 lanes (quad swaps, then
// the 8- and 16-lane mirrors), then the four row totals through scalar registers.  No LDS: __shfl_xor compiles to
// ds_bpermute, and the six dependent LDS round trips of a butterfly (~100 cycles each) were most of a LayerNorm row.
template <int CTRL>
__device__ __forceinline__ float dpp_addf(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_addf<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_addf<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_addf<0x141>(v);   // row_half_mirror
    v = dpp_addf<0x140>(v);   // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over rows of D = 256*NV floats: one wave per row, NV float4 per lane, statistics by
// wavefront shuffles.  (nn.LayerNorm, eps 1e-5: conformer.py:48,98,137,170,207)
// MODE 0: y = LN(x)            MODE 1: y = relu(LN(x))   (embed: Linear->LN->ReLU, conformer.py:205-210)
// MODE 2: u = LN(x); y = (pw0*u + pw1) * sigmoid(pw2*u + pw3)   (scalar Conv2d(1,2,1) + GLU,
//         conformer.py:100,116-117)
// ------------------------------------------------------------------------------------------------
#ifndef CSS_LN_ROWS
#define CSS_LN_ROWS 4
#endif
constexpr int LN_ROWS = CSS_LN_ROWS;   // rows (waves) per LayerNorm block
template <int NV, int MODE>
__global__ __launch_bounds__(64 * LN_ROWS) void layernorm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        float* __restrict__ ys, const float* __restrict__ w,
                                                        const float* __restrict__ b, const float* __restrict__ pw,
                                                        int rows) {
    constexpr int D = 256 * NV;
    const int row = blockIdx.x * LN_ROWS + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * D);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = xr[lane + 64 * i];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + 1e-5f);
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    if (MODE == 2) { p0 = pw[0]; p1 = pw[1]; p2 = pw[2]; p3 = pw[3]; }
    // y: float32 rows (may be null); ys: the same rows in the split-f16 GEMM operand format (may be null)
    float4* yr = reinterpret_cast<float4*>(y + (int64_t)row * D);
    _Float16* ysr = reinterpret_cast<_Float16*>(ys + (int64_t)row * D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float4 g = reinterpret_cast<const float4*>(w)[lane + 64 * i];
        const float4 be = reinterpret_cast<const float4*>(b)[lane + 64 * i];
        float o[4] = {v[i].x * rstd * g.x + be.x, v[i].y * rstd * g.y + be.y, v[i].z * rstd * g.z + be.z,
                      v[i].w * rstd * g.w + be.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (MODE == 1) o[e] = fmaxf(o[e], 0.f);
            if (MODE == 2) o[e] = (p0 * o[e] + p1) * (1.0f / (1.0f + expf(-(p2 * o[e] + p3))));
        }
        if (y) yr[lane + 64 * i] = make_float4(o[0], o[1], o[2], o[3]);
        if (ys) split_store4(ysr, 4 * (lane + 64 * i), o[0], o[1], o[2], o[3]);
    }
}

// Two LayerNorms back to back on one row held in registers: y = LN(x; w1, b1) -> float32 rows (may alias x), then
// z = LN(y; w2, b2) -> float32 and / or split-f16 rows.  Block l's closing LayerNorm (conformer.py:184) and block
// l+1's feed-forward LayerNorm (conformer.py:139) read the same row; the arithmetic is that of the two separate
// kernels, value for value.
template <int NV>
__global__ __launch_bounds__(64 * LN_ROWS) void layernorm2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         float* __restrict__ z, float* __restrict__ zs,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         int rows) {
    constexpr int D = 256 * NV;
    const int row = blockIdx.x * LN_ROWS + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * D);
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = xr[lane + 64 * i];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + 1e-5f);
        const float* w = pass ? w2 : w1;
        const float* b = pass ? b2 : b1;
        float* yo = pass ? z : y;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float4 g = reinterpret_cast<const float4*>(w)[lane + 64 * i];
            const float4 be = reinterpret_cast<const float4*>(b)[lane + 64 * i];
            v[i] = make_float4(v[i].x * rstd * g.x + be.x, v[i].y * rstd * g.y + be.y, v[i].z * rstd * g.z + be.z,
                               v[i].w * rstd * g.w + be.w);
            if (yo) reinterpret_cast<float4*>(yo + (int64_t)row * D)[lane + 64 * i] = v[i];
            if (pass && zs)
                split_store4(reinterpret_cast<_Float16*>(zs + (int64_t)row * D), 4 * (lane + 64 * i), v[i].x, v[i].y, v[i].z, v[i].w);
        }
    }
}

void launch_layernorm2(const float* x, float* y, const float* w1, const float* b1, float* z, float* z_split,
                       const float* w2, const float* b2, int rows, int D, hipStream_t s) {
    const dim3 grid((rows + LN_ROWS - 1) / LN_ROWS), block(64 * LN_ROWS);
    switch (D) {
        case 256: hipLaunchKernelGGL((layernorm2_kernel<1>), grid, block, 0, s, x, y, w1, b1, z, z_split, w2, b2, rows); break;
        case 512: hipLaunchKernelGGL((layernorm2_kernel<2>), grid, block, 0, s, x, y, w1, b1, z, z_split, w2, b2, rows); break;
        case 768: hipLaunchKernelGGL((layernorm2_kernel<3>), grid, block, 0, s, x, y, w1, b1, z, z_split, w2, b2, rows); break;
        case 1024: hipLaunchKernelGGL((layernorm2_kernel<4>), grid, block, 0, s, x, y, w1, b1, z, z_split, w2, b2, rows); break;
        default: break;
    }
}

template <int MODE>
static void launch_ln_mode(const float* x, float* y, float* ys, const float* w, const float* b, const float* pw,
                           int rows, int D, hipStream_t s) {
    const dim3 grid((rows + LN_ROWS - 1) / LN_ROWS), block(64 * LN_ROWS);
    switch (D) {
        case 256: hipLaunchKernelGGL((layernorm_kernel<1, MODE>), grid, block, 0, s, x, y, ys, w, b, pw, rows); break;
        case 512: hipLaunchKernelGGL((layernorm_kernel<2, MODE>), grid, block, 0, s, x, y, ys, w, b, pw, rows); break;
        case 768: hipLaunchKernelGGL((layernorm_kernel<3, MODE>), grid, block, 0, s, x, y, ys, w, b, pw, rows); break;
        case 1024: hipLaunchKernelGGL((layernorm_kernel<4, MODE>), grid, block, 0, s, x, y, ys, w, b, pw, rows); break;
        default: break;  // rejected at css_create (attention_dim must be a multiple of 256, <= 1024)
    }
}

void launch_layernorm(const float* x, float* y, float* y_split, const float* w, const float* b, int rows, int D,
                      int relu, hipStream_t s) {
    if (relu) launch_ln_mode<1>(x, y, y_split, w, b, nullptr, rows, D, s);
    else launch_ln_mode<0>(x, y, y_split, w, b, nullptr, rows, D, s);
}

void launch_ln_glu(const float* x, float* z, const float* w, const float* b, const float* pw, int rows, int D,
                   hipStream_t s) {
    launch_ln_mode<2>(x, z, nullptr, w, b, pw, rows, D, s);
}

// ------------------------------------------------------------------------------------------------
// Depthwise conv module, back half (conformer.py:119-126): 33-tap depthwise Conv1d over time with
// zero padding at the SEGMENT edges (segments are independent sequences), eval-mode BatchNorm folded
// to alpha/beta as ATen folds it, ReLU, scalar Conv2d(1,1,1), residual add into h.
// One lane = one channel (coalesced across channels), one thread produces RUN consecutive frames
// from RUN + TAPS - 1 loaded inputs (2x read amplification instead of 33x).
// ------------------------------------------------------------------------------------------------
template <int TAPS, int RUN>
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ z, float* __restrict__ h,
                                                     const float* __restrict__ wt, const float* __restrict__ dwb,
                                                     const float* __restrict__ alpha, const float* __restrict__ beta,
                                                     const float* __restrict__ pw, int T, int D, int runs_per_seg) {
    constexpr int PAD = (TAPS - 1) / 2;
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= D) return;
    const int seg = blockIdx.y / runs_per_seg, run = blockIdx.y % runs_per_seg;
    const int t0 = run * RUN;
    const float* zs = z + (int64_t)seg * T * D + ch;
    float wk[TAPS];
#pragma clang loop unroll(full)
    for (int k = 0; k < TAPS; ++k) wk[k] = wt[k * D + ch];
    float acc[RUN];
#pragma clang loop unroll(full)
    for (int j = 0; j < RUN; ++j) acc[j] = 0.f;
#pragma clang loop unroll(full)
    for (int p = 0; p < RUN + TAPS - 1; ++p) {
        const int t = t0 + p - PAD;
        const float v = (t >= 0 && t < T) ? zs[(int64_t)t * D] : 0.f;
        // output j uses tap k = p - j; both indices are compile-time constants after full unrolling
#pragma clang loop unroll(full)
        for (int j = (p - (TAPS - 1) > 0 ? p - (TAPS - 1) : 0); j <= (p < RUN - 1 ? p : RUN - 1); ++j)
            acc[j] = fmaf(wk[p - j], v, acc[j]);
    }
    const float bb = dwb[ch], al = alpha[ch], be = beta[ch], w2 = pw[4], c2 = pw[5];
    float* hs = h + (int64_t)seg * T * D + ch;
#pragma unroll
    for (int j = 0; j < RUN; ++j) {
        const int t = t0 + j;
        if (t < T) {
            const float y = fmaxf((acc[j] + bb) * al + be, 0.f);
            hs[(int64_t)t * D] += w2 * y + c2;
        }
    }
}

void launch_dwconv(const float* z, float* h, const float* dw_wt, const float* dw_b, const float* bn_alpha,
                   const float* bn_beta, const float* pw, int nseg, int T, int D, int taps, hipStream_t s) {
    constexpr int RUN = 31;
    const int runs = (T + RUN - 1) / RUN;
    const dim3 grid((D + 255) / 256, nseg * runs), block(256);
    if (taps == 33)
        hipLaunchKernelGGL((dwconv_kernel<33, RUN>), grid, block, 0, s, z, h, dw_wt, dw_b, bn_alpha, bn_beta, pw, T, D, runs);
    else if (taps == 31)
        hipLaunchKernelGGL((dwconv_kernel<31, RUN>), grid, block, 0, s, z, h, dw_wt, dw_b, bn_alpha, bn_beta, pw, T, D, runs);
    else if (taps == 17)
        hipLaunchKernelGGL((dwconv_kernel<17, RUN>), grid, block, 0, s, z, h, dw_wt, dw_b, bn_alpha, bn_beta, pw, T, D, runs);
    // other tap counts are rejected at css_create
}

// ------------------------------------------------------------------------------------------------
// The whole conv module in one kernel (conformer.py:113-127): LayerNorm -> scalar Conv2d(1,2,1) + GLU -> 33-tap
// depthwise conv over time -> eval BatchNorm -> ReLU -> scalar Conv2d(1,1,1) -> residual.
// Block = (segment, run of RUN output frames), one thread per channel.  Phase 1: the block's waves normalise the
// RUN + TAPS - 1 input frames the run needs (frames outside the segment are the conv's zero padding) and leave
// the GLU outputs in LDS (63 frames x 512 channels = 126 KB); phase 2 is dwconv_kernel's arithmetic reading LDS
// instead of global memory.  Saves the write + read of the GLU tensor and one launch per block of the network
// (27.5 -> 14 us per layer for 40 segments); each frame's LayerNorm is computed by two blocks, which is noise.
// x_out must not alias x_in: neighbouring runs read each other's input frames.  The LayerNorm of the module that
// FOLLOWS (ln2_*, z / z_split; optional) is applied to the outgoing rows in the same pass.
// ------------------------------------------------------------------------------------------------
template <int NV, int TAPS, int RUN, int HV>
__global__ __launch_bounds__(256 * NV * HV) void conv_module_kernel(const float* __restrict__ x_in, float* __restrict__ x_out,
                                                              const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                              const float* __restrict__ pw, const float* __restrict__ wt,
                                                              const float* __restrict__ dwb, const float* __restrict__ alpha,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ ln2w, const float* __restrict__ ln2b,
                                                              float* __restrict__ z, float* __restrict__ zs, int T, int runs_per_seg) {
    // HV thread groups of D threads share a block: phase 1 and the output pass simply have HV times the waves, phase 2
    // gives group g the output frames g * JH .. g * JH + JH - 1 of the run.  The kernel is a latency chain (54 blocks per
    // launch on 256 CUs), so the block is made wide, not numerous.
    constexpr int D = 256 * NV, PAD = (TAPS - 1) / 2, ROWS = RUN + TAPS - 1, NW = 4 * NV * HV, JH = (RUN + HV - 1) / HV;
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [ROWS][D]
    // (XCD-contiguous order: neighbouring runs of a segment read each other's frames as halo -- behind one L2 the second
    //  reader hits; dealt round robin over the XCDs every run fetched its 32 halo frames from the fabric again)
    const int item = css_xcd_item((int)blockIdx.x, (int)gridDim.x);
    const int seg = item / runs_per_seg, run = item % runs_per_seg;
    const int t0 = run * RUN;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float p0 = pw[0], p1 = pw[1], p2 = pw[2], p3 = pw[3];
    // phase 2's per-channel operands (taps, BatchNorm, the residual values of the run) are requested now, so that
    // their latency runs under phase 1 instead of after the barrier
    const int ch = threadIdx.x % D, j0 = (threadIdx.x / D) * JH;   // the group index is wave-uniform (D % 64 == 0)
    float wk[TAPS];
#pragma clang loop unroll(full)
    for (int k = 0; k < TAPS; ++k) wk[k] = wt[k * D + ch];
    const float bb = dwb[ch], al = alpha[ch], be_ = beta[ch], w2 = pw[4], c2 = pw[5];
    const float* xs = x_in + (int64_t)seg * T * D + ch;
    float xres[JH];
#pragma clang loop unroll(full)
    for (int j = 0; j < JH; ++j) xres[j] = xs[(int64_t)min(t0 + j0 + j, T - 1) * D];
    // ---- phase 1: LayerNorm + GLU of frames t0 - PAD .. t0 + RUN + PAD - 1 (one wave per frame, as layernorm_kernel).
    // A wave owns frames wave, wave + NW, ...; ALL of them are requested before the first is reduced, so the wave
    // pays one memory round trip, not one per frame.
    constexpr int RPW = (ROWS + NW - 1) / NW;
    float4 v[RPW][NV];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int p = wave + i * NW, t = t0 + p - PAD;
        const bool ok = p < ROWS && t >= 0 && t < T;
        const float4* xr = reinterpret_cast<const float4*>(x_in + ((int64_t)seg * T + (ok ? t : 0)) * D);
#pragma unroll
        for (int j = 0; j < NV; ++j) v[i][j] = ok ? xr[lane + 64 * j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int p = wave + i * NW, t = t0 + p - PAD;
        if (p >= ROWS) break;
        float4* dst = reinterpret_cast<float4*>(tile + p * D);
        if (t < 0 || t >= T) {   // the conv's zero padding
#pragma unroll
            for (int j = 0; j < NV; ++j) dst[lane + 64 * j] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += (v[i][j].x + v[i][j].y) + (v[i][j].z + v[i][j].w);
        const float mean = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[i][j].x -= mean; v[i][j].y -= mean; v[i][j].z -= mean; v[i][j].w -= mean;
            q += (v[i][j].x * v[i][j].x + v[i][j].y * v[i][j].y) + (v[i][j].z * v[i][j].z + v[i][j].w * v[i][j].w);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + 1e-5f);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float4 g = reinterpret_cast<const float4*>(lnw)[lane + 64 * j];
            const float4 be = reinterpret_cast<const float4*>(lnb)[lane + 64 * j];
            float o[4] = {v[i][j].x * rstd * g.x + be.x, v[i][j].y * rstd * g.y + be.y, v[i][j].z * rstd * g.z + be.z,
                          v[i][j].w * rstd * g.w + be.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // GLU gate sigmoid(z) = 1 / (1 + 2^(-z log2 e)) as v_exp_f32 + v_rcp_f32 (1 ulp each; the argument's
                // rounding adds |z| * 6e-8): the library expf and the IEEE division were 25 of the ~40 instructions this
                // phase spends per element, and the phase is the kernel's critical path
                const float z = p2 * o[e] + p3;
                o[e] = (p0 * o[e] + p1) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.44269504088896341f));
            }
            dst[lane + 64 * j] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    // ---- phase 2: depthwise conv of this thread's channel over the run (tap order as dwconv_kernel)
    float acc[JH];
#pragma clang loop unroll(full)
    for (int j = 0; j < JH; ++j) acc[j] = 0.f;
#pragma clang loop unroll(full)
    for (int pp = 0; pp < JH + TAPS - 1; ++pp) {   // LDS row j0 + pp feeds output j0 + j with tap pp - j
        const float zv = tile[min(j0 + pp, ROWS - 1) * D + ch];
#pragma clang loop unroll(full)
        for (int j = (pp - (TAPS - 1) > 0 ? pp - (TAPS - 1) : 0); j <= (pp < JH - 1 ? pp : JH - 1); ++j)
            acc[j] = fmaf(wk[pp - j], zv, acc[j]);
    }
    // the RUN x D outputs go back through LDS (the GLU tile is dead once every thread has finished its taps) and leave
    // as 16-byte row pieces: one 4-byte store per lane and frame made the kernel's tail store-issue bound
    __syncthreads();
#pragma unroll
    for (int j = 0; j < JH; ++j) {
        const float y = fmaxf((acc[j] + bb) * al + be_, 0.f);
        if (j0 + j < RUN) tile[(j0 + j) * D + ch] = xres[j] + (w2 * y + c2);
    }
    __syncthreads();
    // One wave per frame: the frame leaves as 16-byte pieces, and -- the whole row being in the wave's registers -- the
    // LayerNorm of the feed-forward module that follows (conformer.py:139) is applied on the way out, value for value
    // what layernorm_kernel computes from the stored row (z: float32 rows, zs: split-f16 operand rows; either may be null).
    for (int j = wave; j < RUN && t0 + j < T; j += NW) {
        const int64_t row = (int64_t)seg * T + t0 + j;
        float4 r[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            r[i] = reinterpret_cast<const float4*>(tile + j * D)[lane + 64 * i];
            reinterpret_cast<float4*>(x_out + row * D)[lane + 64 * i] = r[i];
            s += (r[i].x + r[i].y) + (r[i].z + r[i].w);
        }
        if (!z && !zs) continue;
        const float mean = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            r[i].x -= mean; r[i].y -= mean; r[i].z -= mean; r[i].w -= mean;
            q += (r[i].x * r[i].x + r[i].y * r[i].y) + (r[i].z * r[i].z + r[i].w * r[i].w);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + 1e-5f);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float4 g = reinterpret_cast<const float4*>(ln2w)[lane + 64 * i];
            const float4 be = reinterpret_cast<const float4*>(ln2b)[lane + 64 * i];
            const float4 o = make_float4(r[i].x * rstd * g.x + be.x, r[i].y * rstd * g.y + be.y, r[i].z * rstd * g.z + be.z,
                                         r[i].w * rstd * g.w + be.w);
            if (z) reinterpret_cast<float4*>(z + row * D)[lane + 64 * i] = o;
            if (zs) split_store4(reinterpret_cast<_Float16*>(zs + row * D), 4 * (lane + 64 * i), o.x, o.y, o.z, o.w);
        }
    }
}

// false: this (D, taps) is not covered (the caller falls back to launch_ln_glu + launch_dwconv)
bool launch_conv_module(const float* x_in, float* x_out, const float* ln_w, const float* ln_b, const float* pw,
                        const float* dw_wt, const float* dw_b, const float* bn_alpha, const float* bn_beta,
                        const float* ln2_w, const float* ln2_b, float* z, float* z_split, int nseg, int T,
                        int D, int taps, hipStream_t s) {
#ifndef CSS_CONV_RUN
#define CSS_CONV_RUN 31   /* output frames per block; 186 = 6 x 31 (tools: -DCSS_CONV_RUN=n for A/B builds) */
#endif
    constexpr int RUN = CSS_CONV_RUN;
    if (taps != 33 || (D != 256 && D != 512)) return false;
    const int runs = (T + RUN - 1) / RUN;
    const size_t lds = (size_t)(RUN + 32) * D * sizeof(float);
    // The LDS reservation is a per-device function attribute: set (a table lookup on the host) on the launch's own device
    // every time, not behind a process-wide flag that a second GPU or a second thread's first launch would miss; if the
    // device refuses, the caller falls back to the two-kernel form.
    const void* fn = D == 512 ? reinterpret_cast<const void*>(&conv_module_kernel<2, 33, RUN, 2>)
                              : reinterpret_cast<const void*>(&conv_module_kernel<1, 33, RUN, 2>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (D == 512)
        hipLaunchKernelGGL((conv_module_kernel<2, 33, RUN, 2>), dim3(nseg * runs), dim3(1024), lds, s, x_in, x_out, ln_w, ln_b, pw, dw_wt,
                           dw_b, bn_alpha, bn_beta, ln2_w, ln2_b, z, z_split, T, runs);
    else
        hipLaunchKernelGGL((conv_module_kernel<1, 33, RUN, 2>), dim3(nseg * runs), dim3(512), lds, s, x_in, x_out, ln_w, ln_b, pw, dw_wt,
                           dw_b, bn_alpha, bn_beta, ln2_w, ln2_b, z, z_split, T, runs);
    return true;
}

// ------------------------------------------------------------------------------------------------
// Relative-position multi-head attention (conformer.py:65-92, 229-233):
//   scores[i][j] = (q_i . k_j + q_i . pe[i - j + maxlen]) / sqrt(dk);  ctx = softmax_j(scores) v
// One wave per (segment, head, 32-query tile); d_k = 64; T <= 32*NJT.
// Everything is computed TRANSPOSED (keys/offsets on the MFMA row axis, queries on the column axis) so
// that a lane owns one query column: the softmax row-reduction is 96 in-register values plus one
// cross-half exchange, and the probabilities are already in the B-operand layout the P.V product
// needs.  The position term uses the Toeplitz structure: R[i][r] = q_i . pe[r0 + r] over the 217
// offsets a 32-query tile can see (never the reference's [T,T,64] gather), staged through LDS to
// apply the per-row skew B[i][j] = R[i][i - j - r0].
// ------------------------------------------------------------------------------------------------
// Two waves per SIMD: the kernel is a chain of dependent MFMA groups, LDS round trips and loads with little to overlap
// inside one wave, so a second resident wave is worth more than the third operand buffer it costs (NTB 3 -> 2 keeps the
// register count at 234 for six key tiles; measured: attention 0.71 -> 0.67 ms per pass).  Eight key tiles (T up to 256)
// would spill at that bound and keep one wave.
template <int NJT, bool QKS, bool FRAG, int W>
__global__ __launch_bounds__(64 * W, (NJT <= 7 ? 2 : 1)) void relpos_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ qkf,
                                                         const float* __restrict__ pe,
                                                         float* __restrict__ ctx, int T, int D, int maxlen,
                                                         int split_out, int heads, int n_items) {
    constexpr int DK = 64;
    // The position term lives in an LDS ring of three 32-offset tiles per query row (row stride 98 floats:
    // the skewed reads of 32 lanes land on addresses 3c + const (mod 32), i.e. 32 distinct banks).  A key tile
    // only ever needs three consecutive offset tiles, and the window slides down by one tile per key tile, so
    // the ring replaces the full [32][217] table (29 KB, 5 waves per CU) by 12.5 KB (register-limited 8 per CU).
    constexpr int LDR = 98;
    // W waves per block, each an item of its own with its own ring; they never meet (no barrier).  The wave index goes through
    // readfirstlane so that segment, head and every base address stay in scalar registers.
    const int wv = W > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    __shared__ __attribute__((aligned(16))) float lds_all[W][32 * LDR];
    float (&lds)[32 * LDR] = lds_all[wv];
    // Block order (round 4): workgroup L runs on XCD L % 8, and the NJT query tiles of one (segment, head) read the SAME
    // keys, values and position tiles.  Dealt round robin they sat in six different L2s, each of which fetched those
    // operands again: 579 MB per launch of 120 segments against 183 MB algorithmic, 6.2 TB/s -- the launch was bound by
    // the fabric (profiles/r04_pmc.md).  Every XCD now takes a contiguous range of the (segment, head, query tile) space,
    // so the tiles of a (segment, head) run back to back behind one L2.
    const int item = css_xcd_item((int)blockIdx.x, (int)gridDim.x) * W + wv;
    if (item >= n_items) return;
    const int qt = item % NJT, head = (item / NJT) % heads, seg = item / (NJT * heads);
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int ld = 3 * D;
    const float* qb = qkv + (int64_t)seg * T * ld + head * DK;
    const float* kb = qb + D;
    const float* vb = qb + 2 * D;
    const int i0 = qt * 32;
    const int iq = min(i0 + c, T - 1);

    // QKS: q, k and pe rows arrive as split-f16 operands (split_f16.hpp; the QKV GEMM writes them, css_create
    // converts pe).  Chunk 2*kk + part of a row is the 16 bytes this lane feeds to v_mfma_f32_32x32x16_f16 for
    // k = 16 kk + 8 h .. + 7 (part 0 = hi, 1 = lo): float offset (kk >> 1) * 32 + (kk & 1) * 8 + 4 h + 16 part.
    float4 q[8];
#define CSS_ATT_LOAD8(dst, rowptr)                                                                      \
    _Pragma("unroll") for (int ch = 0; ch < 8; ++ch)                                                    \
        dst[ch] = *reinterpret_cast<const float4*>((rowptr) + (QKS ? ((ch >> 2) * 32 + ((ch >> 1) & 1) * 8 + (ch & 1) * 16) : 8 * ch) + 4 * h);
    // FRAG: q and k tiles arrive in operand order from the QKV GEMM (kernels.hpp qk_fragment_floats): like the position
    // tiles, 1 KiB contiguous per load.  Rows past T of the last tile were never written (zero or stale, finite): those
    // keys are masked and those query columns never stored.
    const float4* qkt = reinterpret_cast<const float4*>(qkf) +
                        ((int64_t)(seg * heads + head) * NJT) * (2 * 8 * 64) + lane;
    if constexpr (FRAG) {
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) q[ch] = qkt[(qt * 2 + 0) * 512 + ch * 64];
    } else {
        CSS_ATT_LOAD8(q, qb + (int64_t)iq * ld)
    }

    const int rel0 = i0 - (T - 1);
    // Offset tile rt of query tile qt holds the position rows 32 (qt + rt) - (T - 1) + c, c = 0..31: a function of
    // qt + rt only.  `pe` is the fragment-major table pe_fragments_kernel builds for this T: tile m, chunk ch, lane l
    // at float4 index (8 m + ch) 64 + l, so an operand load is 1 KiB contiguous instead of 32 rows x 32 bytes.
    auto pe_tile = [&](int rt) { return pe + ((int64_t)(qt + rt) * 8 * 64 + lane) * 4; };
    auto k_row = [&](int jt) { return kb + (int64_t)min(jt * 32 + c, T - 1) * ld; };
    (void)rel0; (void)maxlen;

    // ---- position term R^T[r][i] = pe[rel0 + r] . q_i (offset tiles, descending) interleaved with the
    //      content term S^T[j][i] = k_j . q_i (key tiles, ascending); skew B[i][j] = R[i][i - j - rel0]
    // highest offset tile a 32-query tile can see: (T - 1 + 31) >> 5 = NJT for every T of this instantiation
    // except T = 32 (NJT - 1) + 1, where tile NJT is computed but never read.  A compile-time constant keeps the
    // whole tile schedule static (a run-time RT0 tripled the code size through duplicated branches).
    constexpr int RT0 = NJT;
    // Operand tiles are consumed in a fixed order -- offset tiles RT0, RT0-1, RT0-2, then key tile jt followed by
    // offset tile RT0-3-jt while that exists -- and fetched one step ahead into two rotating register buffers
    // (tb[step % NTB]; `step` is a compile-time constant after unrolling, so no copies).  With one wave per SIMD the
    // tiles were fetched two steps ahead into three buffers; the second resident wave now covers the L2 latency.
    // A one-tile segment (T <= 32, NJT = 1) sees only the offset tiles 1 and 0 (offsets below -(T - 1) belong to masked
    // keys), so its prologue has two offset tiles, not three: NPRO offset tiles, then NJT key tiles with NI offset tiles
    // interleaved.
    constexpr int NPRO = NJT >= 2 ? 3 : 2;
    constexpr int NI = NJT >= 2 ? NJT - 2 : 0;
    constexpr int NS = NPRO + NJT + NI;
    // step s_ (a compile-time constant wherever it is used): offset tile (first) or key tile (second), -1 = not that kind
    auto step_rt = [](int s_) {
        if (s_ < NPRO) return RT0 - s_;
        const int u_ = s_ - NPRO;
        return (u_ < 2 * NI && (u_ & 1)) ? RT0 - 3 - (u_ >> 1) : -1;
    };
    auto step_jt = [](int s_) {
        const int u_ = s_ - NPRO;
        return u_ < 2 * NI ? (u_ >> 1) : NI + (u_ - 2 * NI);
    };
#define CSS_ATT_LOAD_STEP(dst, s_)                                                                         \
    {                                                                                                      \
        if (step_rt(s_) >= 0) {                                                                            \
            const float4* pt_ = reinterpret_cast<const float4*>(pe_tile(step_rt(s_)));                     \
            _Pragma("unroll") for (int ch = 0; ch < 8; ++ch) dst[ch] = pt_[ch * 64];                       \
        } else if constexpr (FRAG) {                                                                       \
            _Pragma("unroll") for (int ch = 0; ch < 8; ++ch) dst[ch] = qkt[(step_jt(s_) * 2 + 1) * 512 + ch * 64]; \
        } else {                                                                                           \
            CSS_ATT_LOAD8(dst, k_row(step_jt(s_)))                                                         \
        }                                                                                                  \
    }
    constexpr int NTB = 2;   // operand buffers: tile step + NTB - 1 is requested while tile step is consumed
    float4 tb[NTB][8];
    f32x16 S[NJT];
    int step = 0;
    // One step = the 32 (12) MFMAs of the operand tile in tb[step % NTB], with the NEXT tile's eight 16-byte loads issued ONE PER
    // FOUR MFMAs (two per three in split mode) instead of as a burst in front of them (round 5: a burst of vector-memory
    // instructions stalls the in-order wave at the memory pipe's queue with its MFMAs behind it; tools/mfma_f32_chain.hip,
    // DESIGN.md 3.2 -- float32 attention 173 -> 154 us per 120 segments from this alone).
#define CSS_ATT_LOAD_CH(dst, s_, ch)                                                                       \
    {                                                                                                      \
        if (step_rt(s_) >= 0) dst[ch] = reinterpret_cast<const float4*>(pe_tile(step_rt(s_)))[(ch) * 64];  \
        else if constexpr (FRAG) dst[ch] = qkt[(step_jt(s_) * 2 + 1) * 512 + (ch) * 64];                   \
        else dst[ch] = *reinterpret_cast<const float4*>(k_row(step_jt(s_)) + (QKS ? (((ch) >> 2) * 32 + (((ch) >> 1) & 1) * 8 + ((ch) & 1) * 16) : 8 * (ch)) + 4 * h); \
    }
#define CSS_ATT_STEP(acc)                                                                                  \
    {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r] = 0.f;                                       \
        f32x16 cor_ = {0};                                                                                 \
        _Pragma("unroll") for (int ch = 0; ch < 8; ++ch) {                                                 \
            if (step + NTB - 1 < NS) { CSS_ATT_LOAD_CH(tb[(step + NTB - 1) % NTB], min(step + NTB - 1, NS - 1), ch) } \
            const float4 s4_ = tb[step % NTB][ch];                                                         \
            if constexpr (QKS) {   /* hi*hi into acc, hi*lo + lo*hi into a second accumulator folded in with 2^-11 */ \
                if (ch & 1) {                                                                              \
                    const f16x8 sh_ = __builtin_bit_cast(f16x8, tb[step % NTB][ch - 1]), sl_ = __builtin_bit_cast(f16x8, s4_); \
                    const f16x8 qh_ = __builtin_bit_cast(f16x8, q[ch - 1]), ql_ = __builtin_bit_cast(f16x8, q[ch]);         \
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh_, qh_, acc, 0, 0, 0);                  \
                    cor_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh_, ql_, cor_, 0, 0, 0);                \
                    cor_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl_, qh_, cor_, 0, 0, 0);                \
                    __builtin_amdgcn_sched_barrier(0);                                                     \
                }                                                                                          \
            } else {                                                                                       \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s4_.x, q[ch].x, acc, 0, 0, 0);                  \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s4_.y, q[ch].y, acc, 0, 0, 0);                  \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s4_.z, q[ch].z, acc, 0, 0, 0);                  \
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s4_.w, q[ch].w, acc, 0, 0, 0);                  \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
        }                                                                                                  \
        if constexpr (QKS) acc += cor_ * SPLIT_LO_INV;                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        ++step;                                                                                            \
    }
#define CSS_ATT_RING_WRITE(acc, rt)                                                          \
    {                                                                                        \
        const int slot_ = (((rt) % 3) + 3) % 3;                                              \
        _Pragma("unroll") for (int r = 0; r < 16; ++r)                                       \
            lds[c * LDR + slot_ * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = acc[r];             \
    }
    CSS_ATT_LOAD_STEP(tb[0], 0)
    if constexpr (NTB > 2) { CSS_ATT_LOAD_STEP(tb[1], 1) }
#pragma unroll
    for (int u = 0; u < NPRO; ++u) {
        const int rt = RT0 - u;
        f32x16 acc;
        CSS_ATT_STEP(acc)
        CSS_ATT_RING_WRITE(acc, rt)
    }
    float mx = -INFINITY;
    // Skewed read of the position term: this lane needs ring column (i - j) - rel0 = base_c - K with
    // base_c = c + (T - 1) - 4 h and K = 32 jt + (r & 3) + 8 (r >> 2) a compile-time constant; the slot arithmetic
    // ((rr >> 5) % 3) * 32 + (rr & 31) is rr mod 96, so one `mod` per lane and a wrap per element replace the
    // shift / multiply-high / mask chain per element (the kernel is bound by instruction issue).
    // Only the LAST key tile can hold keys past T (masked) or a negative offset (clamped).
    const int base_c = c + (T - 1) - 4 * h;
    const int m0 = base_c >= 0 ? base_c % 96 : 0;
    const float* ringc = lds + c * LDR;
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
        const int rt_new = RT0 - 3 - jt;  // the offset tile key tile jt + 1 adds to the window
        CSS_ATT_STEP(S[jt])
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int K = jt * 32 + (r & 3) + 8 * (r >> 2);
            float bpos, sc;
            if (jt < NJT - 1) {
                const unsigned t = (unsigned)(m0 - K % 96);
                bpos = ringc[min(t, t + 96u)];
                sc = S[jt][r] + bpos;
            } else {
                const int rr = max(base_c - K, 0);
                bpos = ringc[((rr >> 5) % 3) * 32 + (rr & 31)];
                sc = S[jt][r] + bpos;
                sc = K + 4 * h < T ? sc : -INFINITY;
            }
            S[jt][r] = sc;
            mx = fmaxf(mx, sc);
        }
        if (rt_new >= 0) {
            f32x16 acc;
            CSS_ATT_STEP(acc)
            CSS_ATT_RING_WRITE(acc, rt_new)  // overwrites tile RT0 - jt, which key tile jt was the last to read
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    // softmax of scores / sqrt(d_k): p = 2^((s - max) * log2(e) / 8) as one fma + v_exp_f32 per element (the scores above
    // are left unscaled; a positive scale does not move the maximum).  The rounding of the exponent argument costs
    // |arg| * 6e-8 relative (1e-6 for a probability of 1e-8); the library expf this replaces spent 12 instructions per
    // element on range reduction, a fifth of the kernel's instruction stream.
    const float c1 = 0.125f * 1.44269504088896341f, mc = mx * c1;
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(S[jt][r], c1, -mc));
            S[jt][r] = p;
            sum += p;
        }
    }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    // all skewed reads done before the LDS region is reused for the output tile: the region belongs to this wave alone and a
    // wave's LDS instructions execute in order, so nothing is waited for -- the compiler must only keep the order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- O^T[d][i] = sum_j v[j][d] * P[i][j]; the MFMA k index of half h at step (jt, r) is the key
    //      row this lane's S[jt][r] belongs to, so P feeds the B operand straight from registers.
    constexpr int OLD = 65;
    if constexpr (QKS) {
        // Split mode: P.V on the f16 matrix cores too -- 72 f16 MFMAs (2.3 k cycles) per wave instead of 192 float32 ones
        // (12.3 k).  The QKV GEMM writes the v columns as split rows (split_f16.hpp: per 32 features, 32 hi halves then 32
        // lo halves = one 128-byte line per key and feature half), so a (key tile, feature half) tile arrives as four
        // fully coalesced 1 KiB loads, goes to LDS as two row-major [32 keys][32 features] f16 images (hi, lo) and is read
        // back TRANSPOSED by ds_read_b64_tr_b16 (tools/tr_probe.hip): lane i of a 16-lane group points at four halves of
        // row i / 4 and receives column i of the 4 x 16 block -- four consecutive keys of one feature, half an MFMA A
        // operand.  k slot e of lane half h in MFMA kk of key tile jt is the key this lane's S[jt][8 kk + e] belongs to:
        // (e & 3) + 8 (2 kk + (e >> 2)) + 4 h.  (Two-byte loads of the same operands straight from global memory were
        // measured first: 384 load instructions per wave, attention +4.5 %.)  Key tiles outermost: a tile's probabilities
        // are split when its turn comes (the float32 values die there) and feed both feature halves.
        typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
        constexpr int VIMG = 32 * 32;             // halves per image
        constexpr int VLO = VIMG + 32;            // lo image 64 bytes further: its rows' 16-byte pieces miss the hi rows' banks
        constexpr int VBUF = VLO + VIMG;
        // the two tile buffers live where the position ring was (its last reader is behind the barrier above; the output
        // tile goes there again after the last tile has been read)
        static_assert(2 * VBUF * sizeof(_Float16) <= sizeof(lds), "V tile buffers must fit in the ring");
        _Float16* vlds = reinterpret_cast<_Float16*>(lds);
        // tile g = 2 jt + dt: this lane's four 16-byte pieces (key 8 m + lane / 8, piece lane % 8 of the 128-byte line)
        float4 vr[4];
        const int vkey = lane >> 3, vpc = lane & 7;
#define CSS_ATT_VLOAD(g_)                                                                                          \
    _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                               \
        const int ju_ = ((g_) >> 1) * 32 + 8 * m + vkey;                                                          \
        const int row_ = ((g_) >> 1) < NJT - 1 ? ju_ : min(ju_, T - 1);                                           \
        vr[m] = *reinterpret_cast<const float4*>(vb + (int64_t)row_ * ld + ((g_) & 1) * 32 + 4 * vpc);            \
    }
#define CSS_ATT_VSTORE(g_)                                                                                         \
    _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                                 \
        *reinterpret_cast<float4*>(vlds + ((g_) & 1) * VBUF + (vpc >> 2) * VLO + (8 * m + vkey) * 32 + (vpc & 3) * 8) = vr[m];
        const int tq = c >> 4, ti = c & 15;
        // halves offset of this lane's piece for (kk, e-half): row 16 kk + 8 eh + 4 h + ti / 4, columns 16 tq + 4 (ti % 4)
        const int troff = (4 * h + (ti >> 2)) * 32 + 16 * tq + 4 * (ti & 3);
#define CSS_ATT_TR(g_, img_, kk_, eh_)                                                                             \
    __builtin_amdgcn_ds_read_tr16_b64_v4f16(reinterpret_cast<__attribute__((address_space(3))) h4*>(              \
        (__attribute__((address_space(3))) _Float16*)vlds + ((g_) & 1) * VBUF + (img_) * VLO + (16 * (kk_) + 8 * (eh_)) * 32 + troff))
        CSS_ATT_VLOAD(0)
        CSS_ATT_VSTORE(0)
        CSS_ATT_VLOAD(1)
        f32x16 o0 = {0}, o1 = {0}, cor0 = {0}, cor1 = {0};
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            f16x8 ph[2], pl[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 a, b;
                    split_f16(S[jt][8 * kk + e], a, b);
                    ph[kk][e] = a;
                    pl[kk][e] = b;
                }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int g = 2 * jt + dt;
                if (g + 1 < 2 * NJT) { CSS_ATT_VSTORE(g + 1) }       // tile g + 1: registers -> the other LDS buffer
                if (g + 2 < 2 * NJT) { CSS_ATT_VLOAD(g + 2) }        // tile g + 2: on its way while tile g is consumed
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const h4 a0 = CSS_ATT_TR(g, 0, kk, 0), a1 = CSS_ATT_TR(g, 0, kk, 1);
                    const h4 b0 = CSS_ATT_TR(g, 1, kk, 0), b1 = CSS_ATT_TR(g, 1, kk, 1);
                    f16x8 vh, vl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vh[e] = (_Float16)a0[e]; vh[4 + e] = (_Float16)a1[e];
                        vl[e] = (_Float16)b0[e]; vl[4 + e] = (_Float16)b1[e];
                    }
                    if (dt == 0) {
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[kk], o0, 0, 0, 0);
                        cor0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[kk], cor0, 0, 0, 0);
                        cor0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[kk], cor0, 0, 0, 0);
                    } else {
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[kk], o1, 0, 0, 0);
                        cor1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[kk], cor1, 0, 0, 0);
                        cor1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[kk], cor1, 0, 0, 0);
                    }
                }
            }
        }
        o0 += cor0 * SPLIT_LO_INV;
        o1 += cor1 * SPLIT_LO_INV;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = (r & 3) + 8 * (r >> 2) + 4 * h;
            lds[c * OLD + d] = o0[r] * inv;
            lds[c * OLD + 32 + d] = o1[r] * inv;
        }
#undef CSS_ATT_VLOAD
#undef CSS_ATT_VSTORE
#undef CSS_ATT_TR
    } else {
    // Exact float32 P.V.  The value rows arrive as 8-BYTE loads: lane (c, h) takes features 2c, 2c + 1 of the key its S[jt][r]
    // belongs to (half h: 4 rows further) -- a load instruction is two whole 256-byte rows of the head -- and feeds two
    // accumulators, the even and the odd features.  Until round 5 these were 192 four-byte loads per wave, one per MFMA:
    // beside a stream of these MFMAs a 4-byte load costs the matrix pipe ~60 cycles at that density, an 8-byte load ~8, a
    // 16-byte load nothing (tools/mfma_f32_chain.hip), and two waves of a SIMD spent 46 k clocks in this phase for 24.6 k of
    // MFMA issue.  Buffer loads: one per-lane offset, the row as a scalar offset, rows past T read zero (their probabilities
    // are zero); the next key tile's 16 loads are issued one per MFMA pair.  Every output element still sums its keys in
    // ascending order: bit for bit the results of the 4-byte form.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 vb2[2][16];
    const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vb), 0, (int)(((int64_t)(T - 1) * ld + DK) * 4), 0x00020000);
    const int vvo = (4 * h * ld + 2 * c) * 4, ldb = ld * 4;
#define CSS_ATT_LOADV(jt_, r_) __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsv, vvo, ((jt_) * 32 + ((r_) & 3) + 8 * ((r_) >> 2)) * ldb, 0))
#pragma unroll
    for (int r = 0; r < 16; ++r) vb2[0][r] = CSS_ATT_LOADV(0, r);
    f32x16 oe = {0}, oo = {0};
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (jt + 1 < NJT) vb2[(jt + 1) & 1][r] = CSS_ATT_LOADV(jt + 1, r);
            oe = __builtin_amdgcn_mfma_f32_32x32x2f32(vb2[jt & 1][r].x, S[jt][r], oe, 0, 0, 0);
            oo = __builtin_amdgcn_mfma_f32_32x32x2f32(vb2[jt & 1][r].y, S[jt][r], oo, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = 2 * ((r & 3) + 8 * (r >> 2) + 4 * h);
        lds[c * OLD + d] = oe[r] * inv;
        lds[c * OLD + d + 1] = oo[r] * inv;
    }
#undef CSS_ATT_LOADV
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 32 query rows x 64 features leave in 16-byte pieces (the store tail of a row-per-lane epilogue is bound by store
    // issue, not bandwidth): lane l takes features 4 (l % 16) .. + 3 of rows l / 16 + 4 k
    {
        const int d4 = (lane & 15) * 4;
        float* ob = ctx + ((int64_t)seg * T + i0) * D;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int il = (lane >> 4) + 4 * k;
            if (i0 + il >= T) continue;
            const float* src = lds + il * OLD + d4;
            const float v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
            if (split_out)   // ctx rows in the split-f16 GEMM operand format (split_f16.hpp) for the output projection
                split_store4(reinterpret_cast<_Float16*>(ob + (int64_t)il * D), head * DK + d4, v0, v1, v2, v3);
            else
                *reinterpret_cast<float4*>(ob + (int64_t)il * D + head * DK + d4) = make_float4(v0, v1, v2, v3);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same attention for segments of ANY length (css/css.py:144-171 takes any segment_size_sec; relpos_attn_kernel keeps a
// query tile's scores in registers and stops at 16 key tiles = 512 frames = 8 s).  Written for reach, not speed: float32
// fused multiply-adds on the vector ALU in both arithmetic modes, one block of 256 threads per (segment, head, NQ queries),
// everything that depends on T in a loop or in LDS:
//   r[i][u] = q_i . pe[clamp(i0 - (T - 1) + u)]     u = 0 .. NQ + T - 2   (the Toeplitz form: B[i][j] = r[i][i + T - 1 - j])
//   s[i][j] = (q_i . k_j + r[i][i + T - 1 - j]) / sqrt(d_k),  softmax over j in place,  ctx_i = sum_j p[i][j] v_j
// A quad of lanes holds one key row (or one position row) in registers and walks the NQ queries in LDS; the position products
// are added into the score rows where they belong; the host picks the largest NQ <= 16 whose q and s rows fit the LDS.  qkv: float32 rows [token][3 D] (the QKV GEMM writes them plain for this path),
// pe: the float32 table [2 maxlen][64].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relpos_attn_long_kernel(const float* __restrict__ qkv, const float* __restrict__ pe,
                                                               float* __restrict__ ctx, int T, int D, int maxlen, int split_out,
                                                               int heads, int nq) {
    constexpr int DK = 64;
    extern __shared__ __attribute__((aligned(16))) float att_lds[];
    const int qtiles = (T + nq - 1) / nq;
    const int item = blockIdx.x;
    const int qt = item % qtiles, head = (item / qtiles) % heads, seg = item / (qtiles * heads);
    const int i0 = qt * nq, nv = min(nq, T - i0);
    float* qs = att_lds;                         // [nq][64]
    float* ss = qs + nq * DK;                    // [nq][T]
    const int tid = threadIdx.x;
    const int64_t ld = 3 * (int64_t)D;
    const float* base = qkv + (int64_t)seg * T * ld + head * DK;
    for (int e = tid; e < nv * DK; e += 256) qs[e] = base[(int64_t)(i0 + e / DK) * ld + (e % DK)];
    __syncthreads();
    // Four lanes share a key (or position) row: lane `sub` of a quad holds features 16 m + 4 sub .. + 3, m = 0 .. 3, so a wave's
    // load instruction covers 64 contiguous bytes of 16 rows.  (A row per lane -- 64 lanes x 256 bytes in flight per wave --
    // thrashed the vector L1: every 16-byte piece fetched its 128-byte line again, 3.0 ms per layer on 10 s segments.)  The
    // quad's partial products meet by two DPP steps; four queries run side by side because a block is one wave per SIMD and
    // has nothing else to cover a chain of dependent multiply-adds.
    const int sub = tid & 3, kq = tid >> 2;
    auto quad_sum = [](float v) {
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
        return v;
    };
    auto load_row = [&](const float* rowp, float (&row)[16]) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float4 v = *reinterpret_cast<const float4*>(rowp + 16 * m + 4 * sub);
            row[4 * m] = v.x; row[4 * m + 1] = v.y; row[4 * m + 2] = v.z; row[4 * m + 3] = v.w;
        }
    };
    // partial products of queries i .. i + 3 with the lane's 16 features, summed over the quad
    auto dots4 = [&](int i, const float (&row)[16], float (&acc)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 q = *reinterpret_cast<const float4*>(qs + min(i + k, nv - 1) * DK + 16 * m + 4 * sub);
                acc[k] = fmaf(q.x, row[4 * m], acc[k]); acc[k] = fmaf(q.y, row[4 * m + 1], acc[k]);
                acc[k] = fmaf(q.z, row[4 * m + 2], acc[k]); acc[k] = fmaf(q.w, row[4 * m + 3], acc[k]);
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = quad_sum(acc[k]);
    };
    // key products: 64 keys per step
    for (int j0 = 0; j0 < T; j0 += 64) {
        const int j = j0 + kq;
        float row[16];
        load_row(base + (int64_t)min(j, T - 1) * ld + D, row);
        for (int i = 0; i < nv; i += 4) {
            float acc[4];
            dots4(i, row, acc);
            // (lane `sub` of the quad stores query i + sub: every lane holds all four sums)
            const float mine = sub == 0 ? acc[0] : sub == 1 ? acc[1] : sub == 2 ? acc[2] : acc[3];
            if (j < T && i + sub < nv) ss[(size_t)(i + sub) * T + j] = mine;
        }
    }
    __syncthreads();
    // position products, added where they belong -- query i0 + i and key j meet at offset (i0 + i) - j = i0 - (T - 1) + u,
    // u = i + T - 1 - j: for a query, different offsets are different keys, so no two lanes touch one score
    const int nu = nv + T - 1;
    for (int u0 = 0; u0 < nu; u0 += 64) {
        const int u = u0 + kq;
        const int rel = max(-maxlen, min(i0 - (T - 1) + u, maxlen - 1)) + maxlen;     // conformer.py:24-29
        float row[16];
        load_row(pe + (int64_t)rel * DK, row);
        for (int i = 0; i < nv; i += 4) {
            float acc[4];
            dots4(i, row, acc);
            const float mine = sub == 0 ? acc[0] : sub == 1 ? acc[1] : sub == 2 ? acc[2] : acc[3];
            const int j = i + sub + T - 1 - u;
            if (u < nu && i + sub < nv && j >= 0 && j < T) ss[(size_t)(i + sub) * T + j] = (ss[(size_t)(i + sub) * T + j] + mine) * 0.125f;
        }
    }
    __syncthreads();
    // softmax of each query row, in place (one wave per row)
    const int wave = tid >> 6, lane = tid & 63;
    for (int i = wave; i < nv; i += 4) {
        float* row = ss + (size_t)i * T;
        float mx = -INFINITY;
        for (int j = lane; j < T; j += 64) mx = fmaxf(mx, row[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int j = lane; j < T; j += 64) { const float e = expf(row[j] - mx); row[j] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int j = lane; j < T; j += 64) row[j] *= inv;
    }
    __syncthreads();
    // context: thread = (feature d, query group); the value rows are read along d (coalesced), p[i][j] is an LDS broadcast
    {
        const int d = tid & 63, ig = tid >> 6;
        constexpr int MAXQ = 4;                  // queries per thread: nq <= 16
        float acc[MAXQ] = {0.f, 0.f, 0.f, 0.f};
        const float* vb = base + 2 * D + d;
#pragma unroll 8
        for (int j = 0; j < T; ++j) {   // (unrolled: eight value rows on their way at a time)
            const float v = vb[(int64_t)j * ld];
#pragma unroll
            for (int k = 0; k < MAXQ; ++k) {
                const int i = ig + 4 * k;
                if (i < nv) acc[k] = fmaf(ss[(size_t)i * T + j], v, acc[k]);
            }
        }
        float* ob = ctx + ((int64_t)seg * T + i0) * D;
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int i = ig + 4 * k;
            if (i >= nv) continue;
            if (split_out) split_store(reinterpret_cast<_Float16*>(ob + (int64_t)i * D), head * DK + d, acc[k]);
            else ob[(int64_t)i * D + head * DK + d] = acc[k];
        }
    }
}

bool launch_relpos_attention_long(const float* qkv, const float* pe, float* ctx, int nseg, int T, int D, int H, int maxlen,
                                  int split_out, hipStream_t s) {
    if (D != H * 64) return false;
    // queries per block: up to 16, as many as leave room for three blocks per CU (a block is one wave per SIMD and its loops
    // are chains; 20 s segments: 16 queries at one block per CU 6.0 ms per layer)
    int nq = 16;
    auto bytes = [&](int n) { return ((size_t)n * 64 + (size_t)n * T) * sizeof(float); };
    while (nq > 1 && bytes(nq) > 52 * 1024) nq >>= 1;
    if (bytes(nq) > 150 * 1024) return false;    // (one query per block still does not fit: segments beyond ~ 19 000 frames)
    if (bytes(nq) > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(relpos_attn_long_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes(nq)) != hipSuccess)
        return false;
    const unsigned qtiles = (unsigned)((T + nq - 1) / nq);
    hipLaunchKernelGGL(relpos_attn_long_kernel, dim3(qtiles * (unsigned)H * (unsigned)nseg), dim3(256), bytes(nq), s, qkv, pe, ctx, T, D,
                       maxlen, split_out, H, nq);
    return true;
}

// Position rows in the order the attention kernel's MFMA operands want them (see pe_tile there): tile m holds rows
// 32 m - (T - 1) + c of the relative-position table (clamped to [-maxlen, maxlen - 1] as conformer.py:24 clamps them),
// float4 index (8 m + ch) 64 + l = the 16 bytes lane l = c + 32 h reads for chunk ch of row c.  `pe` is the table as the
// kernel used to read it row by row: float32 rows (split = 0) or split-f16 rows (split = 1), d_k = 64.
__global__ __launch_bounds__(256) void pe_fragments_kernel(const float* __restrict__ pe, float* __restrict__ frag, int T,
                                                           int maxlen, int ntiles, int split) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ntiles * 8 * 64) return;
    const int lane = i & 63, ch = (i >> 6) & 7, m = i >> 9;
    const int c = lane & 31, h = lane >> 5;
    int prow = 32 * m - (T - 1) + c;
    prow = max(-maxlen, min(prow, maxlen - 1)) + maxlen;
    const int off = (split ? ((ch >> 2) * 32 + ((ch >> 1) & 1) * 8 + (ch & 1) * 16) : 8 * ch) + 4 * h;
    reinterpret_cast<float4*>(frag)[i] = *reinterpret_cast<const float4*>(pe + (int64_t)prow * 64 + off);
}

void launch_pe_fragments(const float* pe, float* frag, int T, int maxlen, int split, hipStream_t s) {
    const int ntiles = pe_fragment_tiles(T);
    hipLaunchKernelGGL(pe_fragments_kernel, dim3((ntiles * 8 * 64 + 255) / 256), dim3(256), 0, s, pe, frag, T, maxlen, ntiles, split);
}

// waves (items) per block of the tuned instantiations (NJT <= 7).  Four: consecutive items are the query tiles of one (segment,
// head) and land on one CU together (A/B of 1 / 2 / 4 / 8 on one box, libraries differing in this constant only: the float32
// step 8.82 / 8.75 / 8.80 / 8.93 ms, the split-f16 step 4.63 / 4.54 / 4.48 / 4.45-4.50 ms; eight make a 100 KB block that
// shares its CU with nothing else)
#ifndef CSS_ATT_WAVES
#define CSS_ATT_WAVES 4
#endif
void launch_relpos_attention(const float* qkv, const float* qk_frag, const float* pe_frag, float* ctx, int nseg, int T, int D,
                             int H, int maxlen, int qk_split, int split_out, hipStream_t s) {
    const int qtiles = (T + 31) / 32;
    const int n_items = qtiles * H * nseg;
    // the tile schedule is static per instantiation, so NJT must be exactly ceil(T / 32)
#define CSS_ATT_LAUNCH(n, w)                                                                                                 \
    {                                                                                                                        \
        const dim3 grid((unsigned)((n_items + (w) - 1) / (w))), block(64 * (w));                                              \
        if (qk_split && qk_frag) hipLaunchKernelGGL((relpos_attn_kernel<n, true, true, w>), grid, block, 0, s, qkv, qk_frag, pe_frag, ctx, T, D, maxlen, split_out, H, n_items);  \
        else if (qk_split) hipLaunchKernelGGL((relpos_attn_kernel<n, true, false, w>), grid, block, 0, s, qkv, qk_frag, pe_frag, ctx, T, D, maxlen, split_out, H, n_items);  \
        else hipLaunchKernelGGL((relpos_attn_kernel<n, false, false, w>), grid, block, 0, s, qkv, qk_frag, pe_frag, ctx, T, D, maxlen, split_out, H, n_items);          \
    }
#define CSS_ATT_CASE(n) case n: CSS_ATT_LAUNCH(n, (n <= 7 ? CSS_ATT_WAVES : 1)) break;
    switch (qtiles) {
        CSS_ATT_CASE(1) CSS_ATT_CASE(2) CSS_ATT_CASE(3) CSS_ATT_CASE(4)
        CSS_ATT_CASE(5) CSS_ATT_CASE(6) CSS_ATT_CASE(7) CSS_ATT_CASE(8)
        // segments of 257 .. 512 frames (4 .. 8 s): the same schedule with one wave per SIMD; the score tiles of a
        // query tile stay in registers up to NJT = 16 (256 VGPRs + up to 158 accumulation registers, no spills)
        CSS_ATT_CASE(9) CSS_ATT_CASE(10) CSS_ATT_CASE(11) CSS_ATT_CASE(12)
        CSS_ATT_CASE(13) CSS_ATT_CASE(14) CSS_ATT_CASE(15) CSS_ATT_CASE(16)
        default: break;  // longer segments are rejected at css_begin (segment_frames <= 512)
    }
#undef CSS_ATT_CASE
#undef CSS_ATT_LAUNCH
}

}  // namespace css

// Front/back end of the CSS path outside the GEMMs: PCM layout, network input features, and the
// overlap-add that finishes the inverse transform.
#include <algorithm>

#include <cstdlib>

#include "kernels.hpp"
#include "split_f16.hpp"

namespace css {

bool css_force_long_path() {   // kernels.hpp
    static const bool on = [] { const char* e = getenv("CSS_FORCE_LONG_PATH"); return e && atoi(e) != 0; }();
    return on;
}


// sum over the 64 lanes (every lane gets the total): DPP inside the rows of 16, row totals through scalar registers
// (no ds_bpermute round trips; see wave_sum in encoder.hip)
template <int CTRL>
__device__ __forceinline__ float dpp_addf_(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum_f(float v) {
    v = dpp_addf_<0xB1>(v);
    v = dpp_addf_<0x4E>(v);
    v = dpp_addf_<0x141>(v);
    v = dpp_addf_<0x140>(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// ------------------------------------------------------------------------------------------------
// [n][C] sample-major PCM (css/helpers.py:40 load_audio layout) -> channel-major [C][n_pad], so that
// every later read of the frame buffer is a contiguous run of one channel
// (conformer_wrapper.py:119 does the same moveaxis on the host).
// ------------------------------------------------------------------------------------------------
// split_out: each channel row is written as a split-f16 GEMM operand (split_f16.hpp; n_pad % 32 == 0).  The
// overlapping frames the analysis GEMM reads (row stride = hop, a multiple of 32) are valid split rows of it.
__global__ void deinterleave_kernel(const float* __restrict__ pcm, float* __restrict__ out, int64_t n, int C,
                                    int64_t n_pad, int64_t i_lo, int64_t i_hi, int split_out) {
    const int64_t i = i_lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= i_hi) return;
    for (int c = 0; c < C; ++c) {
        const float v = i < n ? pcm[i * C + c] : 0.f;
        if (split_out) split_store(reinterpret_cast<_Float16*>(out + (int64_t)c * n_pad + (i & ~(int64_t)31)), (int)(i & 31), v);
        else out[(int64_t)c * n_pad + i] = v;
    }
}

void launch_deinterleave(const float* pcm, float* pcm_cm, int64_t n, int C, int64_t n_pad, int64_t i_lo, int64_t i_hi,
                         int split_out, hipStream_t s) {
    if (i_hi <= i_lo) return;
    const int64_t blocks = (i_hi - i_lo + 255) / 256;
    hipLaunchKernelGGL(deinterleave_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pcm, pcm_cm, n, C, n_pad, i_lo, i_hi,
                       split_out);
}

// ------------------------------------------------------------------------------------------------
// The two wav edges of the path on the device.
//   in : C mono PCM16 planes (the 7 files of a session, css/helpers.py:40 load_audio; libsndfile scales int16 by
//        2^-15) -> sample-major float32 [n][C]
//   out: peak normalisation x * 0.99 / (max|x| + 1e-7) in float32 (utils/audio_utils.py:44-45), then libsndfile's
//        float -> PCM16 conversion lrint(x * 32767): the same operations, value for value, as wavio.write_wav.
// ------------------------------------------------------------------------------------------------
__global__ void pcm16_to_float_kernel(const int16_t* __restrict__ planes, float* __restrict__ pcm, int64_t n, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < C; ++c) pcm[i * C + c] = (float)planes[(int64_t)c * n + i] * (1.0f / 32768.0f);
}

void launch_pcm16_to_float(const int16_t* planes, float* pcm, int64_t n, int C, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(pcm16_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, planes, pcm, n, C);
}

// the same scaling straight into the channel-major frame buffer [C][n_pad] (zeros past n), samples [i_lo, i_hi): planes
// are already channel-major, so the wav edge needs no sample-major detour
__global__ void pcm16_to_cm_kernel(const int16_t* __restrict__ planes, float* __restrict__ out, int64_t n, int64_t n_pad,
                                   int64_t i_lo, int64_t i_hi) {
    const int64_t i = i_lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= i_hi) return;
    const int c = blockIdx.y;
    out[(int64_t)c * n_pad + i] = i < n ? (float)planes[(int64_t)c * n + i] * (1.0f / 32768.0f) : 0.f;
}

// max |sample| over count values as float bits (non-negative floats order like unsigned integers), OR-ed into *peak by
// atomicMax: the recording's level for the power-of-two gain of the split synthesis operand (split_f16.hpp level_gain)
template <class T>
__global__ __launch_bounds__(256) void pcm_peak_kernel(const T* __restrict__ x, int64_t count, float scale, unsigned int* __restrict__ peak) {
    float m = 0.f;
    constexpr int V = 16 / (int)sizeof(T);                        // values per 16-byte load
    const int64_t head = std::min<int64_t>(count, (V - (int64_t)((reinterpret_cast<uintptr_t>(x) / sizeof(T)) % V)) % V);
    const int64_t nv = (count - head) / V;
    struct alignas(16) Vec { T v[V]; };
    const Vec* xv = reinterpret_cast<const Vec*>(x + head);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        const Vec q = xv[i];
#pragma unroll
        for (int e = 0; e < V; ++e) m = fmaxf(m, fabsf((float)q.v[e]));
    }
    if (blockIdx.x == 0)   // the unaligned head and the tail
        for (int64_t i = threadIdx.x; i < head + (count - head - nv * V); i += 256) {
            const int64_t j = i < head ? i : head + nv * V + (i - head);
            m = fmaxf(m, fabsf((float)x[j]));
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    // one atomic per BLOCK, and only when it can raise the peak: thousands of waves hitting one word cost ~11 ns each
    // (8192 of them made this 10 MB scan take 97 us)
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float b = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])) * scale;
        if (b > __uint_as_float(__hip_atomic_load(peak, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) atomicMax(peak, __float_as_uint(b));
    }
}

void launch_pcm_peak_f32(const float* x, int64_t count, unsigned int* peak, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(pcm_peak_kernel<float>, dim3((unsigned)std::min<int64_t>((count / 4 + 255) / 256 + 1, 512)), dim3(256), 0, s, x, count, 1.0f, peak);
}
void launch_pcm_peak_i16(const int16_t* x, int64_t count, unsigned int* peak, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(pcm_peak_kernel<int16_t>, dim3((unsigned)std::min<int64_t>((count / 8 + 255) / 256 + 1, 512)), dim3(256), 0, s, x, count, 1.0f / 32768.0f, peak);
}

void launch_pcm16_to_channel_major(const int16_t* planes, float* pcm_cm, int64_t n, int C, int64_t n_pad, int64_t i_lo,
                                   int64_t i_hi, hipStream_t s) {
    if (i_hi <= i_lo) return;
    hipLaunchKernelGGL(pcm16_to_cm_kernel, dim3((unsigned)((i_hi - i_lo + 255) / 256), C), dim3(256), 0, s, planes, pcm_cm, n,
                       n_pad, i_lo, i_hi);
}

// peak[s] = max |wav[s][:]| (as float bits: non-negative floats order like unsigned integers; peak zeroed by the caller)
__global__ __launch_bounds__(256) void peak_kernel(const float* __restrict__ wav, int64_t n, unsigned int* __restrict__ peak) {
    const int s = blockIdx.y;
    const float* w = wav + (int64_t)s * n;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(peak + s, __float_as_uint(m));
}

__global__ void encode_pcm16_kernel(const float* __restrict__ wav, int64_t n, const unsigned int* __restrict__ peak,
                                    int16_t* __restrict__ out, int64_t out_ld) {
    const int s = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float den = __fadd_rn(__uint_as_float(peak[s]), 1e-7f);
    const float y = __fdiv_rn(__fmul_rn(wav[(int64_t)s * n + i], 0.99f), den);
    double v = rint((double)y * 32767.0);
    v = fmin(fmax(v, -32768.0), 32767.0);
    out[(int64_t)s * out_ld + i] = (int16_t)v;
}

void launch_encode_pcm16(const float* wav, int S, int64_t n, unsigned int* peak_bits, int16_t* out, int64_t out_ld,
                         hipStream_t s) {
    if (n <= 0) return;
    hipMemsetAsync(peak_bits, 0, (size_t)S * sizeof(unsigned int), s);
    hipLaunchKernelGGL(peak_kernel, dim3(256, S), dim3(256), 0, s, wav, n, peak_bits);
    hipLaunchKernelGGL(encode_pcm16_kernel, dim3((unsigned)((n + 255) / 256), S), dim3(256), 0, s, wav, n, peak_bits, out, out_ld);
}

// ------------------------------------------------------------------------------------------------
// Network input features of one segment (feature.py:478-508 compute_spectra, 198-249 IPDFeature,
// then the global affine of conformer.py:298-299), written straight into the row-major
// [token][K_pad] operand of the embed GEMM.
//   m == 0 : mean/variance-normalised clamped magnitude of mic 0 (unbiased std over the T frames)
//   m >= 1 : inter-channel phase difference of pair (m, 0), version-1 mean normalisation, raw angle
// Zero-padded frames of the last segment take part in the statistics with X = 0 (css.py:185-190).
// Block = (32-bin tile, m, segment); a wave owns 8 bins, lanes run over time so the plane reads are
// contiguous; statistics by wavefront shuffles; the [bin][t] tile is transposed through LDS so the
// stores are 128-byte runs along the feature axis.
//
// Phase of an exactly real negative bin (DC / Nyquist have Im == +0 by construction of the transform):
// the reference reaches the features through polar() -> angle() (conformer_wrapper.py:124,94), which
// maps such a bin to the float32 value just inside -pi, and the side of the atan2 branch cut of the
// IPD feature depends on that (DESIGN.md "Numerical hazards").  PHASE_NEG_REAL is that value.
// ------------------------------------------------------------------------------------------------
// FEAT_TMAX: segment lengths an instantiation covers (256: up to 4 s, the shipped 3 s; 512: up to 8 s)
#define CSS_EPS32 1.1920928955078125e-07f

template <int FEAT_TMAX>
__global__ __launch_bounds__(256) void features_kernel(const float* __restrict__ X, int64_t T_ld, int64_t stft_frames,
                                                       int F, float* __restrict__ feat, int Kp,
                                                       const float* __restrict__ in_bias,
                                                       const float* __restrict__ in_scale, int64_t seg_lo, int T,
                                                       int hop, int split_out, FeatOpts o, const float* __restrict__ PH) {
    constexpr int FEAT_LD = FEAT_TMAX + 1;
    __shared__ float tile[32 * FEAT_LD];
    // blockIdx.y = 0: the spectral rows (microphone 0); y >= 1: IPD pair y - 1 = phase[ml] - phase[mr]
    const int f0 = blockIdx.x * 32, m = blockIdx.y, segl = blockIdx.z;
    const int ml = m ? o.pair_l[m - 1] : 0, mr = m ? o.pair_r[m - 1] : 0;
    const int64_t st = (seg_lo + segl) * (int64_t)hop;
    const int64_t tv64 = stft_frames - st;
    const int tv = (int)(tv64 < 0 ? 0 : (tv64 > T ? T : tv64));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float invT = 1.0f / (float)T;
    constexpr int NT = FEAT_TMAX / 64;
    // plane values of bin b + 1 are requested while bin b is being computed (the atan2 / sincos work of a bin is long
    // enough to cover the round trip; without this every one of a wave's 8 bins started with an exposed load)
    float n_r0[NT], n_i0[NT], n_rm[NT], n_im[NT];
    auto fetch = [&](int b_) {
        const int f_ = min(f0 + wave * 8 + b_, F - 1);
        const float* re0 = X + ((int64_t)mr * 2 * F + f_) * T_ld + st;       // the pair's right microphone (0 for the spectral rows)
        const float* im0 = X + ((int64_t)mr * 2 * F + F + f_) * T_ld + st;
        const float* rem = X + ((int64_t)ml * 2 * F + f_) * T_ld + st;       // the pair's left microphone
        const float* imm = X + ((int64_t)ml * 2 * F + F + f_) * T_ld + st;
        // (with phase planes an IPD block reads the two phases of its pair -- n_rm: left, n_im: right -- instead of four planes)
        const float* phl = PH ? PH + ((int64_t)ml * F + f_) * T_ld + st : nullptr;
        const float* phr = PH ? PH + ((int64_t)mr * F + f_) * T_ld + st : nullptr;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = lane + 64 * i;
            const bool ok = t < tv;   // (tv <= T)
            if (PH && m != 0) {
                n_r0[i] = 0.f; n_i0[i] = 0.f;
                n_rm[i] = ok ? phl[t] : 0.f;
                n_im[i] = ok ? phr[t] : 0.f;
            } else {
                n_r0[i] = ok ? re0[t] : 0.f;
                n_i0[i] = ok ? im0[t] : 0.f;
                n_rm[i] = (ok && m != 0) ? rem[t] : 0.f;
                n_im[i] = (ok && m != 0) ? imm[t] : 0.f;
            }
        }
    };
    fetch(0);
    for (int b = 0; b < 8; ++b) {
        const int fl = wave * 8 + b, f = f0 + fl;
        if (f >= F) break;
        float c_r0[NT], c_i0[NT], c_rm[NT], c_im[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) { c_r0[i] = n_r0[i]; c_i0[i] = n_i0[i]; c_rm[i] = n_rm[i]; c_im[i] = n_im[i]; }
        if (b + 1 < 8) fetch(b + 1);
        float a[NT], bq[NT], dd[NT];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = lane + 64 * i;
            a[i] = 0.f; bq[i] = 0.f; dd[i] = 0.f;
            if (t < T) {
                const float r0 = c_r0[i], i0 = c_i0[i];
                if (m == 0) {
                    a[i] = fmaxf(sqrtf(r0 * r0 + i0 * i0), CSS_EPS32);
                    if (o.log_mag) a[i] = logf(a[i]);                    // feature.py:500-501
                    s0 += a[i];
                } else {
                    const float rm = c_rm[i], imv = c_im[i];
                    // (zero-padded frames: both phases 0, as css_phase_of(0, 0))
                    const float d = PH ? rm - imv : css_phase_of(rm, imv) - css_phase_of(r0, i0);
                    dd[i] = d;
                    sincosf(d, &bq[i], &a[i]);   // (one argument reduction for both; the same values as sinf / cosf)
                    s0 += a[i];
                    s1 += bq[i];
                    s2 += d;
                }
            }
        }
        s0 = wave_sum_f(s0);
        s1 = wave_sum_f(s1);
        if (m == 0) {
            if (o.mvn) {                                                 // feature.py:503-507
                const float mean = s0 * invT;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int t = lane + 64 * i;
                    if (t < T) { a[i] -= mean; q += a[i] * a[i]; }
                }
                const float sd = sqrtf(wave_sum_f(q) / (float)(T - 1));
                const float den = sd + CSS_EPS32;
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int t = lane + 64 * i;
                    if (t < T) tile[fl * FEAT_LD + t] = a[i] / den;
                }
            } else {
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int t = lane + 64 * i;
                    if (t < T) tile[fl * FEAT_LD + t] = a[i];
                }
            }
        } else if (o.ipd_norm && o.ipd_version == 1 && !o.ipd_cos) {     // the shipped configuration (feature.py:220-221,245)
            const float yrm = s0 * invT, yim = s1 * invT;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int t = lane + 64 * i;
                if (t < T) tile[fl * FEAT_LD + t] = atan2f(bq[i] - yim, a[i] - yrm);
            }
        } else {                                                         // the other IPDFeature options (feature.py:214-236)
            const float yrm = s0 * invT, yim = s1 * invT;
            float shift = 0.f;
            if (o.ipd_norm && o.ipd_version == 2) shift = atan2f(yim, yrm);
            if (o.ipd_norm && o.ipd_version == 3) shift = wave_sum_f(s2) * invT;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int t = lane + 64 * i;
                if (t < T) {
                    float v = (o.ipd_norm && o.ipd_version == 1) ? atan2f(bq[i] - yim, a[i] - yrm) : dd[i] - shift;
                    if (o.ipd_cos) v = cosf(v);
                    tile[fl * FEAT_LD + t] = v;
                }
            }
        }
    }
    __syncthreads();
    // transposed store: lanes 0..31 -> consecutive feature columns (128 B), two frames per wave-instruction
    const int fl = lane & 31, f = f0 + fl;
    if (f < F) {
        const int col = m * F + f;
        const float bi = in_bias[col], sc = in_scale[col];
        if (split_out) {   // rows in the split-f16 GEMM operand format (split_f16.hpp) for the embed Linear
            float* out = feat + (int64_t)segl * T * Kp;
            for (int t = wave * 2 + (lane >> 5); t < T; t += 8)
                split_store(reinterpret_cast<_Float16*>(out + (int64_t)t * Kp), col, (tile[fl * FEAT_LD + t] + bi) * sc);
        } else {
            float* out = feat + (int64_t)segl * T * Kp + col;
            for (int t = wave * 2 + (lane >> 5); t < T; t += 8) out[(int64_t)t * Kp] = (tile[fl * FEAT_LD + t] + bi) * sc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same features for segments of ANY length (features_kernel holds a bin's T values in registers and a [32][T] tile in
// LDS: T <= 512).  One wave per (bin, m, segment); the values of a frame are formed again in every pass instead of being
// kept (statistics, [variance,] output), lanes run over time with the same strides and the sums are formed in the same
// order as in features_kernel (measured on a 3 s session: the angle rows are the same bits, the magnitude rows differ by
// one float32 ulp in half of the elements -- the compiler contracts the two kernels' mean / deviation arithmetic
// differently).  Rows leave one element per lane (a 4-byte store every K_pad floats): written for reach, not for speed.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void features_long_kernel(const float* __restrict__ X, int64_t T_ld, int64_t stft_frames,
                                                           int F, float* __restrict__ feat, int Kp,
                                                           const float* __restrict__ in_bias, const float* __restrict__ in_scale,
                                                           int64_t seg_lo, int T, int hop, int split_out, FeatOpts o,
                                                           const float* __restrict__ PH) {
    const int f = blockIdx.x, m = blockIdx.y, segl = blockIdx.z;
    const int ml = m ? o.pair_l[m - 1] : 0, mr = m ? o.pair_r[m - 1] : 0;
    const int64_t st = (seg_lo + segl) * (int64_t)hop;
    const int64_t tv64 = stft_frames - st;
    const int tv = (int)(tv64 < 0 ? 0 : (tv64 > T ? T : tv64));
    const int lane = threadIdx.x;
    const float invT = 1.0f / (float)T;
    const float* re0 = X + ((int64_t)mr * 2 * F + f) * T_ld + st;
    const float* im0 = X + ((int64_t)mr * 2 * F + F + f) * T_ld + st;
    const float* rem = X + ((int64_t)ml * 2 * F + f) * T_ld + st;
    const float* imm = X + ((int64_t)ml * 2 * F + F + f) * T_ld + st;
    const float* phl = PH ? PH + ((int64_t)ml * F + f) * T_ld + st : nullptr;
    const float* phr = PH ? PH + ((int64_t)mr * F + f) * T_ld + st : nullptr;
    // the values of frame t: m == 0: a = magnitude (clamped, optionally log); m >= 1: a = cos d, b = sin d, d = phase difference
    auto eval = [&](int t, float& a, float& b, float& d) {
        const bool ok = t < tv;   // zero-padded frames of the last segment: X = 0
        if (m == 0) {
            const float r0 = ok ? re0[t] : 0.f, i0 = ok ? im0[t] : 0.f;
            a = fmaxf(sqrtf(r0 * r0 + i0 * i0), CSS_EPS32);
            if (o.log_mag) a = logf(a);
            b = 0.f; d = 0.f;
        } else {
            if (PH) d = (ok ? phl[t] : 0.f) - (ok ? phr[t] : 0.f);
            else d = css_phase_of(ok ? rem[t] : 0.f, ok ? imm[t] : 0.f) - css_phase_of(ok ? re0[t] : 0.f, ok ? im0[t] : 0.f);
            sincosf(d, &b, &a);
        }
    };
    const int col = m * F + f;
    const float bi = in_bias[col], sc = in_scale[col];
    float* out = feat + (int64_t)segl * T * Kp;
    auto put = [&](int t, float v) {
        const float y = (v + bi) * sc;
        if (split_out) split_store(reinterpret_cast<_Float16*>(out + (int64_t)t * Kp), col, y);
        else out[(int64_t)t * Kp + col] = y;
    };
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int t = lane; t < T; t += 64) {
        float a, b, d;
        eval(t, a, b, d);
        s0 += a; s1 += b; s2 += d;
    }
    s0 = wave_sum_f(s0);
    s1 = wave_sum_f(s1);
    if (m == 0) {
        float mean = 0.f, den = 1.f;
        if (o.mvn) {                                                 // feature.py:503-507
            mean = s0 * invT;
            float q = 0.f;
            for (int t = lane; t < T; t += 64) {
                float a, b, d;
                eval(t, a, b, d);
                a -= mean;
                q += a * a;
            }
            den = sqrtf(wave_sum_f(q) / (float)(T - 1)) + CSS_EPS32;
        }
        for (int t = lane; t < T; t += 64) {
            float a, b, d;
            eval(t, a, b, d);
            put(t, o.mvn ? (a - mean) / den : a);
        }
    } else {
        const float yrm = s0 * invT, yim = s1 * invT;
        float shift = 0.f;
        if (o.ipd_norm && o.ipd_version == 2) shift = atan2f(yim, yrm);
        if (o.ipd_norm && o.ipd_version == 3) shift = wave_sum_f(s2) * invT;
        for (int t = lane; t < T; t += 64) {
            float a, b, d;
            eval(t, a, b, d);
            float v = (o.ipd_norm && o.ipd_version == 1) ? atan2f(b - yim, a - yrm) : d - shift;
            if (o.ipd_cos) v = cosf(v);
            put(t, v);
        }
    }
}

void launch_features(const float* X, int64_t T_ld, int64_t stft_frames, int C, int F, float* feat, int Kp,
                     const float* in_bias, const float* in_scale, int64_t seg_lo, int nseg, int T, int hop,
                     int split_out, const FeatOpts& opts, hipStream_t s, const float* PH) {
    (void)C;
    const dim3 grid((F + 31) / 32, 1 + opts.num_pairs, nseg), block(256);
    if (T > 512 || css_force_long_path())
        hipLaunchKernelGGL(features_long_kernel, dim3(F, 1 + opts.num_pairs, nseg), dim3(64), 0, s, X, T_ld, stft_frames, F, feat, Kp,
                           in_bias, in_scale, seg_lo, T, hop, split_out, opts, PH);
    else if (T <= 192)   // (the shipped 3 s segments: a 24.7 KB tile, six blocks per CU instead of four -- 2 520 blocks are two
                         // rounds of 1 536 instead of three of 1 024: 72.3 -> 62.7 us per 40 segments, A/B on one box)
        hipLaunchKernelGGL(features_kernel<192>, grid, block, 0, s, X, T_ld, stft_frames, F, feat, Kp, in_bias, in_scale,
                           seg_lo, T, hop, split_out, opts, PH);
    else if (T <= 256)
        hipLaunchKernelGGL(features_kernel<256>, grid, block, 0, s, X, T_ld, stft_frames, F, feat, Kp, in_bias, in_scale,
                           seg_lo, T, hop, split_out, opts, PH);
    else
        hipLaunchKernelGGL(features_kernel<512>, grid, block, 0, s, X, T_ld, stft_frames, F, feat, Kp, in_bias, in_scale,
                           seg_lo, T, hop, split_out, opts, PH);
}

// ------------------------------------------------------------------------------------------------
// Overlap-add that completes conv_transpose1d (feature.py:162) after the synthesis GEMM produced
// G[b][q][0..L): output sample hop*q + r receives the frames q - j that cover it, j = ceil-ish(L / hop) - 1 .. 0 (frame_len =
// 2 hop: frame q - 1's second half and frame q's first half).
// Gather form: one thread per output sample, no atomics, bit-reproducible.
// ------------------------------------------------------------------------------------------------
__global__ void wave_ola_kernel(const float* __restrict__ G, float* __restrict__ out, int64_t T_frames, int hop, int L,
                                int64_t q_lo, int64_t q_hi, int64_t f_lo, int64_t f_hi, int64_t out_ld, int64_t out_q0,
                                const unsigned int* __restrict__ level) {
    const int b = blockIdx.y;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t q = q_lo + idx / hop;
    const int r = (int)(idx % hop);
    if (q >= q_hi) return;
    const int64_t n = (q - out_q0) * hop + r;
    if (n >= out_ld) return;
    const float* g = G + (int64_t)b * T_frames * L;
    // frames q - j cover this sample at offset r + j hop < L; added oldest first (frame_len = 2 hop: frame q - 1's second
    // half, then frame q's first half -- the order this kernel has always had)
    float v = 0.f;
    const int jmax = (L - 1 - r) / hop;
    for (int j = jmax; j >= 0; --j) {
        const int64_t t = q - j;
        if (t < f_lo || t >= f_hi) continue;
        const float x = g[t * L + r + (int64_t)j * hop];
        v = j == jmax ? x : v + x;   // (the oldest frame is assigned, the later ones added to what is there -- 0.f when it was missing)
    }
    if (level) v *= 1.0f / level_gain(level);   // a power of two: exact
    out[(int64_t)b * out_ld + n] = v;
}

void launch_wave_ola(const float* G, float* out, int B, int64_t T_frames, int hop, int L, int64_t q_lo, int64_t q_hi,
                     int64_t f_lo, int64_t f_hi, int64_t out_ld, int64_t out_q0, const unsigned int* level, hipStream_t s) {
    const int64_t total = (q_hi - q_lo) * hop;
    if (total <= 0) return;
    const dim3 grid((unsigned)((total + 255) / 256), B), block(256);
    hipLaunchKernelGGL(wave_ola_kernel, grid, block, 0, s, G, out, T_frames, hop, L, q_lo, q_hi, f_lo, f_hi, out_ld, out_q0, level);
}

// ------------------------------------------------------------------------------------------------
// Waveform shards of a segment-sharded meeting -> the stitched streams (parallel.py exchange 3).  gathered [world][S][ld]:
// rank r's shard holds output blocks t_lo[r] .. t_hi[r] (one hop each); the block at a seam is in both neighbours'
// shards, one frame's contribution each, and is their sum (rank order: a two-term float sum commutes, so the result is
// the single-GPU pass's bit for bit).  Ranks without frames have t_hi == t_lo and are skipped.
// ------------------------------------------------------------------------------------------------
struct JoinRanks { int64_t t_lo[64], t_hi[64]; int world; };
__global__ void join_shards_kernel(const float* __restrict__ g, int64_t ld, JoinRanks rk, int S, int hop, int64_t n_out,
                                   float* __restrict__ out, int64_t out_ld) {
    const int64_t n4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int sp = blockIdx.y;
    if (n4 >= n_out) return;
    const int64_t q = n4 / hop;
    const int r = (int)(n4 - q * hop);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < rk.world; ++k) {
        if (rk.t_hi[k] <= rk.t_lo[k] || q < rk.t_lo[k] || q > rk.t_hi[k]) continue;
        const float4 a = *reinterpret_cast<const float4*>(g + ((int64_t)k * S + sp) * ld + (q - rk.t_lo[k]) * hop + r);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    float* o = out + (int64_t)sp * out_ld + n4;
    if (n4 + 4 <= n_out) *reinterpret_cast<float4*>(o) = v;
    else { const float e[4] = {v.x, v.y, v.z, v.w}; for (int i = 0; n4 + i < n_out; ++i) o[i] = e[i]; }
}

void launch_join_shards(const float* gathered, int64_t ld, const int64_t* t_lo, const int64_t* t_hi, int world, int S, int hop,
                        int64_t n_out, float* out, int64_t out_ld, hipStream_t s) {
    JoinRanks rk{};
    rk.world = world;
    for (int k = 0; k < world; ++k) { rk.t_lo[k] = t_lo[k]; rk.t_hi[k] = t_hi[k]; }
    const int64_t groups = (n_out + 3) / 4;
    hipLaunchKernelGGL(join_shards_kernel, dim3((unsigned)((groups + 255) / 256), S), dim3(256), 0, s, gathered, ld, rk, S, hop, n_out, out, out_ld);
}

// ------------------------------------------------------------------------------------------------
// [B][rows][T] planes -> [B][T][KIp] rows (separator-protocol istft entry point only; the fused path
// writes the row layout directly from the stitcher).  32x32 LDS transpose.
// ------------------------------------------------------------------------------------------------
__global__ void planes_to_rows_kernel(const float* __restrict__ planes, float* __restrict__ rows, int F2, int64_t T,
                                      int KIp) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int64_t t0 = (int64_t)blockIdx.x * 32;
    const int r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty in 0..7
    const float* src = planes + (int64_t)b * F2 * T;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k;
        const int64_t t = t0 + tx;
        tile[k][tx] = (r < F2 && t < T) ? src[(int64_t)r * T + t] : 0.f;
    }
    __syncthreads();
    float* dst = rows + (int64_t)b * T * KIp;
    for (int k = ty; k < 32; k += 8) {
        const int64_t t = t0 + k;
        const int r = r0 + tx;
        if (t < T && r < KIp) dst[t * KIp + r] = tile[tx][k];
    }
}

void launch_planes_to_rows(const float* planes, float* rows, int B, int F2, int64_t T, int KIp, hipStream_t s) {
    const dim3 grid((unsigned)((T + 31) / 32), (KIp + 31) / 32, B), block(256);
    hipLaunchKernelGGL(planes_to_rows_kernel, grid, block, 0, s, planes, rows, F2, T, KIp);
}

}  // namespace css

// Analysis transform (feature.py:88-128: Hann-windowed 512-point DFT, hop 256) as an LDS-staged FFT.
//
// The reference evaluates the transform as a convolution with a DFT kernel (a direct O(N^2) sum); round 1 mirrored
// that as a GEMM on the float32 matrix cores (13.8 GFLOP per minute of 7-channel audio, 180 us).  The transform is
// memory-sized work -- 27 MB of samples in, 54 MB of planes out per minute -- so it is done here the cheap way:
// a block owns 16 consecutive frames of one channel, reads their samples as contiguous runs (window applied on the
// way in), packs each real frame into 256 complex points, runs four radix-4 Stockham stages over all 16 frames in LDS,
// separates the real spectrum (bins 0 .. 256) and writes it transposed -- time fastest, 16-byte pieces of 64-byte runs -- into the
// planes X[c][Re f | Im f][t] every later stage reads.  The rounding is that of an FFT (~log2 N ulps against ~sqrt N for
// the direct sum; both ~1e-7 relative to the frame's largest bin, tests/test_hip_parity.py holds the planes to 1e-6
// against the oracle and 2e-6 against the reference).  The imaginary parts of the DC and Nyquist bins are exact zeros
// by construction (the IPD feature's branch cut depends on it, frontend.hip CSS_PHASE_NEG_REAL).
#include <cmath>
#include <vector>

#include "kernels.hpp"

namespace css {

constexpr int FFT_N = 512;            // frame length
constexpr int FFT_H = 256;            // packed complex length = hop
constexpr int FFT_TB = 16;            // frames per block: 35 KB of LDS, four blocks per CU cover each other's load / store phases
constexpr int FFT_FS = FFT_H + 1;     // frame stride in LDS (complex elements): odd, so the frames of a store pass hit distinct bank pairs
constexpr int FFT_THREADS = 256;

// tables: window[512] | tw256[256] (cos, -sin of 2 pi m / 256) | tw512[257] (cos, -sin of 2 pi k / 512)
size_t stft_table_floats() { return FFT_N + 2 * FFT_H + 2 * (FFT_H + 1); }

// window 0: 'hann' (the shipped extractors); 1: 'sqrt_hann' -- feature.py:29-36: W = hann ** 0.5 on the float32 window and the
// kernel divided by S = 0.5 sqrt(N N / hop) = 16 (init_kernel's normalize, which STFTBase never overrides: feature.py:63-66);
// the power of two is folded into the window table (exact)
void stft_build_tables(float* t, int window) {
    for (int n = 0; n < FFT_N; ++n) {
        const float hann = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / FFT_N));   // torch.hann_window (periodic), float32
        t[n] = window == 1 ? std::sqrt(hann) / 16.0f : hann;
    }
    float* t256 = t + FFT_N;
    for (int m = 0; m < FFT_H; ++m) {
        t256[2 * m] = (float)std::cos(2.0 * M_PI * m / FFT_H);
        t256[2 * m + 1] = (float)(-std::sin(2.0 * M_PI * m / FFT_H));
    }
    float* t512 = t256 + 2 * FFT_H;
    for (int k = 0; k <= FFT_H; ++k) {
        t512[2 * k] = (float)std::cos(2.0 * M_PI * k / FFT_N);
        t512[2 * k + 1] = (float)(-std::sin(2.0 * M_PI * k / FFT_N));
    }
}

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// x: channel c's samples start at x + c * x_stride; frame t begins at sample t * 256.  Frames [t_lo, t_hi) are transformed.
// out: plane row r of channel c, frame t at out + (c * 2F + r) * row_ld + t   (r = f: Re, F + f: Im; F = 257).
// A block owns the 16-frame tile [16 B, 16 B + 16) of ABSOLUTE frame indices (frames of the tile outside the range are
// computed from a clamped frame and not stored), so that a group of four consecutive frames of one plane row is a
// 16-byte-aligned piece whatever range a launch covers (row_ld % 4 == 0, out 16-byte aligned): the spectrum leaves as
// float4 stores, 8 per thread instead of 32 scalar ones -- the store phase was bound by store issue.
// Block order: workgroup L runs on XCD L % 8, and a tile's 64-byte row pieces are HALF an L2 line -- the other half is the
// next tile's.  Tiles dealt round robin put the two halves into two different L2s, which then write 64-byte partial lines
// to rows that lie T_ld * 4 bytes apart (450 KB for a 30-min meeting: 1.6 TB/s there against 2.4 TB/s at 60 s, where the
// planes never leave the Infinity Cache).  So every XCD takes a CONTIGUOUS range of the (channel, tile) space: neighbouring
// tiles pass through one L2 back to back and leave as whole lines (and the hop-overlapped sample reads hit that L2 too).
__global__ __launch_bounds__(FFT_THREADS) void stft_fft_kernel(const float* __restrict__ x, int64_t x_stride, int t_lo, int t_hi,
                                                               const float* __restrict__ tab, float* __restrict__ out,
                                                               int64_t row_ld, int wide, int tiles, float* __restrict__ phase) {
    extern __shared__ __attribute__((aligned(16))) float2 fft_lds[];
    float2* bufA = fft_lds;
    float2* twl = fft_lds + FFT_TB * FFT_FS;   // the 256 stage twiddles: one coalesced load instead of three dependent gathers per stage
    const int tid = threadIdx.x;
    const int item = css_xcd_item((int)blockIdx.x, (int)gridDim.x);   // position in the (channel, tile) space
    const int c = item / tiles;
    const int t0 = (t_lo / FFT_TB + (item - c * tiles)) * FFT_TB;
    const float* xc = x + (int64_t)c * x_stride;
    const float* win = tab;
    const float2* tw256 = reinterpret_cast<const float2*>(tab + FFT_N);
    const float2* tw512 = reinterpret_cast<const float2*>(tab + FFT_N + 2 * FFT_H);
    // ---- windowed, packed load: z[n] = w[2n] x[2n] + i w[2n+1] x[2n+1]
    twl[tid] = tw256[tid];
#pragma unroll 4
    for (int e = tid; e < FFT_TB * FFT_H; e += FFT_THREADS) {
        const int fr = e >> 8, n = e & 255;
        const int t = max(min(t0 + fr, t_hi - 1), t_lo);   // frames outside the range repeat its nearest frame (never stored)
        const float2 s = *reinterpret_cast<const float2*>(xc + (int64_t)t * FFT_H + 2 * n);
        const float2 w = *reinterpret_cast<const float2*>(win + 2 * n);
        bufA[fr * FFT_FS + n] = make_float2(s.x * w.x, s.y * w.y);
    }
    __syncthreads();
    // ---- four radix-4 Stockham stages (decimation in time) IN PLACE: every thread holds its four butterflies' inputs in
    //      registers, the block meets, the outputs go back into the same buffer (half the LDS of a ping-pong pair: four
    //      blocks per CU instead of two cover each other's load, barrier and store phases)
    float2* src = bufA;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int Ns = 1 << (2 * p);              // 1, 4, 16, 64
        constexpr int NB = FFT_TB * 64 / FFT_THREADS;
        float2 r0[NB], r1[NB], r2[NB], r3[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int b = i * FFT_THREADS + tid;
            const int fr = b >> 6, j = b & 63;
            const int k = j & (Ns - 1);
            const float2* in = src + fr * FFT_FS;
            float2 u0 = in[j], u1 = in[j + 64], u2 = in[j + 128], u3 = in[j + 192];
            if (p > 0) {
                const int m = k * (64 / Ns);        // exp(-2 pi i k / (4 Ns)) = tw256[k * 64 / Ns]
                u1 = cmulf(u1, twl[m]);
                u2 = cmulf(u2, twl[2 * m]);
                u3 = cmulf(u3, twl[3 * m]);
            }
            // radix-4 butterfly (forward: multiply by -i is (x, y) -> (y, -x))
            const float2 a0 = make_float2(u0.x + u2.x, u0.y + u2.y), a1 = make_float2(u0.x - u2.x, u0.y - u2.y);
            const float2 a2 = make_float2(u1.x + u3.x, u1.y + u3.y), a3 = make_float2(u1.y - u3.y, u3.x - u1.x);   // -i (u1 - u3)
            r0[i] = make_float2(a0.x + a2.x, a0.y + a2.y);
            r1[i] = make_float2(a1.x + a3.x, a1.y + a3.y);
            r2[i] = make_float2(a0.x - a2.x, a0.y - a2.y);
            r3[i] = make_float2(a1.x - a3.x, a1.y - a3.y);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int b = i * FFT_THREADS + tid;
            const int fr = b >> 6, j = b & 63;
            const int k = j & (Ns - 1);
            const int j0 = ((j - k) << 2) + k;
            float2* o = src + fr * FFT_FS + j0;
            o[0] = r0[i];
            o[Ns] = r1[i];
            o[2 * Ns] = r2[i];
            o[3 * Ns] = r3[i];
        }
        __syncthreads();
    }
    // ---- real spectrum from the packed transform Z (in src), written time-fastest
    //   E = (Z[k] + conj Z[256-k]) / 2,  O = (Z[k] - conj Z[256-k]) / (2i),  X[k] = E + e^{-2 pi i k / 512} O
    const int F = FFT_H + 1;
    static_assert(FFT_TB == 16, "store phase indexing");
    auto bin = [&](int fr, int f, float2 w5, float& re, float& im) {
        const float2 z0 = src[fr * FFT_FS + (f & 255)];
        const float2 z1 = src[fr * FFT_FS + ((256 - f) & 255)];
        if (f == 0) { re = z0.x + z0.y; im = 0.f; }
        else if (f == FFT_H) { re = z0.x - z0.y; im = 0.f; }
        else {
            const float2 e = make_float2(0.5f * (z0.x + z1.x), 0.5f * (z0.y - z1.y));
            const float2 o = make_float2(0.5f * (z0.y + z1.y), 0.5f * (z1.x - z0.x));
            const float2 wo = cmulf(w5, o);
            re = e.x + wo.x;
            im = e.y + wo.y;
        }
    };
    float* oc = out + (int64_t)c * 2 * F * row_ld + t0;
    float* pc = phase ? phase + (int64_t)c * F * row_ld + t0 : nullptr;
    if (wide) {
        // a thread owns one bin row of a four-frame group: lanes 0..3 cover the tile's 64 bytes of that row
        const int g4 = (tid & 3) * 4;
        const int ta = t0 + g4;
        const bool all_in = ta >= t_lo && ta + 4 <= t_hi;
        for (int f = tid >> 2; f < F; f += FFT_THREADS / 4) {
            const float2 w5 = tw512[f];
            float re[4], im[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bin(g4 + q, f, w5, re[q], im[q]);
            float* pr = oc + (int64_t)f * row_ld + g4;
            float* pi = oc + (int64_t)(F + f) * row_ld + g4;
            float* pp = pc ? pc + (int64_t)f * row_ld + g4 : nullptr;
            if (all_in) {
                *reinterpret_cast<float4*>(pr) = make_float4(re[0], re[1], re[2], re[3]);
                *reinterpret_cast<float4*>(pi) = make_float4(im[0], im[1], im[2], im[3]);
                if (pp)
                    *reinterpret_cast<float4*>(pp) = make_float4(css_phase_of(re[0], im[0]), css_phase_of(re[1], im[1]),
                                                                 css_phase_of(re[2], im[2]), css_phase_of(re[3], im[3]));
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (ta + q >= t_lo && ta + q < t_hi) {
                        pr[q] = re[q]; pi[q] = im[q];
                        if (pp) pp[q] = css_phase_of(re[q], im[q]);
                    }
            }
        }
    } else {
        const int tl = tid & (FFT_TB - 1);
        const bool store = t0 + tl >= t_lo && t0 + tl < t_hi;
        for (int f = tid / FFT_TB; f < F; f += FFT_THREADS / FFT_TB) {
            float re, im;
            bin(tl, f, tw512[f], re, im);
            if (store) {
                oc[(int64_t)f * row_ld + tl] = re;
                oc[(int64_t)(F + f) * row_ld + tl] = im;
                if (pc) pc[(int64_t)f * row_ld + tl] = css_phase_of(re, im);
            }
        }
    }
}

// frames [t_lo, t_hi) of C channels; x and out are the bases of frame 0
bool launch_stft_fft(const float* x, int64_t x_stride, int C, int64_t t_lo, int64_t t_hi, const float* tables, float* out,
                     int64_t row_ld, hipStream_t s, float* phase) {
    if (t_hi <= t_lo || C <= 0) return true;
    const size_t lds = ((size_t)FFT_TB * FFT_FS + FFT_H) * sizeof(float2);   // 34.9 KB
    // (the attribute is per device: set it on every launch -- a host-side table lookup -- rather than behind a
    // process-wide flag that a second device, or a second thread's first launch, would miss)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(stft_fft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return false;
    const int tiles = (int)((t_hi + FFT_TB - 1) / FFT_TB - t_lo / FFT_TB);
    const int wide = (row_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(stft_fft_kernel, dim3((unsigned)tiles * (unsigned)C), dim3(FFT_THREADS), lds, s, x, x_stride, (int)t_lo,
                       (int)t_hi, tables, out, row_ld, wide, tiles, phase);
    return true;
}

}  // namespace css

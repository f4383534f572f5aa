// fp32 MFMA GEMM for gfx950:  C[m][n] = epilogue( sum_k A[m][k] * B[n][k] ).
//
// Used for every dense product on the CSS path: the Hann-DFT analysis transform (feature.py:116
// conv1d, restated as DFT-matrix x overlapping frames), the Conformer's embed / FFN / QKV / output /
// mask-head Linear layers (conformer.py:49-53,139-142,206,285) and the sqrt-Hann synthesis transform
// (feature.py:162 conv_transpose1d).  Exact float32 (v_mfma_f32_32x32x2_f32 is a k-ordered fmaf
// chain), which the 1e-4 waveform parity and the winner-take-all mask decisions need.
//
// Shape: 128x128 block tile, K slab 32, register-staged double-buffered LDS, one barrier per slab.
// Two wave layouts share the code (template WM = waves along M):
//   WM = 2: 4 waves, each 64x64 = 2x2 MFMA tiles   (fewest LDS reads per MFMA)
//   WM = 4: 8 waves, each 32x64 = 1x2 MFMA tiles   (two waves per SIMD from ONE block: when a launch
//           has only ~1 block per CU, the second wave covers the other's barrier / LDS-latency bubbles)
// LDS rows are padded to 36 floats so that ds_read_b128 of 16 consecutive rows hits 16 distinct 16-byte
// slots (9*row mod 16 is a bijection).
//
// K permutation: within each group of 8 consecutive k the two lane halves of the MFMA take k = 4h+s
// (h = lane>>5, s = MFMA step) instead of 2s+h, so each lane fetches its four A (and B) operands with
// one 16-byte LDS read.  A and B use the same permutation, so the dot product is unchanged.
#include <cstdlib>

#include <cstdio>

#include "gemm_common.hpp"

namespace css {

template <int BM, int WM>
__global__ __launch_bounds__(WM * 128, 2) void gemm_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    constexpr int THREADS = WM * 128;          // WM x 2 waves
    constexpr int TM = (BM / WM) / 32;         // MFMA tiles along M per wave (2 or 1)
    constexpr int LROWS = THREADS / 8;         // rows covered by one staging pass (32 or 64)
    constexpr int NLA = BM / LROWS;            // staging passes of the A tile (2 or 4)
    constexpr int NLB = BN / LROWS;            // staging passes of the B tile (2 or 4)
    static_assert((NLA == 2 || NLA == 4) && (NLB == 2 || NLB == 4) && (TM == 1 || TM == 2), "unsupported tile layout");
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDS_LD];
    // ---- XCD-aware tile mapping: consecutive tiles (which share an A row panel) go to one XCD/L2 ----
    const int n_tiles = tiles_m * tiles_n * g.batch;
    const int L = blockIdx.x;
    const int tile = xcd_tile(L, n_tiles);
    const int per_batch = tiles_m * tiles_n;
    const int bz = tile / per_batch;
    const int t2 = tile - bz * per_batch;
    // (m_fastest: the row tiles of one COLUMN panel are neighbours -- the launch whose B operand is the large one, the mask
    // head: weights are A, the tokens B)
    const int tn = g.m_fastest ? t2 / tiles_m : t2 % tiles_n, tm = g.m_fastest ? t2 % tiles_m : t2 / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const float* __restrict__ A = g.A + (int64_t)bz * g.strideA;
    const float* __restrict__ B = g.B + (int64_t)bz * g.strideB;
    float* __restrict__ C = g.C + (int64_t)bz * g.strideC;

    const int tid = threadIdx.x;
    const int lr = tid >> 3;          // row within a staging pass
    const int lc = (tid & 7) << 2;    // 0..28 : float offset within the K slab
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int c = lane & 31, h = lane >> 5;
    const int M = g.M, N = g.N;

    // Rows past M / N read the last valid row instead (always in bounds, always finite).  A row of C
    // depends on one row of A and one row of B only, so such rows merely produce outputs the guarded
    // epilogue never stores -- and, unlike a select on the loaded value, nothing consumes the loads
    // before the LDS store, so they stay in flight under the MFMAs of the current slab.
#define CSS_ROWPTR(base, ld, row, lim) ((base) + (int64_t)((row) < (lim) ? (row) : (lim) - 1) * (ld) + lc)
    const float* pa0 = CSS_ROWPTR(A, g.lda, m0 + lr, M);
    const float* pa1 = CSS_ROWPTR(A, g.lda, m0 + lr + LROWS, M);
    const float* pa2 = CSS_ROWPTR(A, g.lda, m0 + lr + 2 * LROWS, M);   // passes 2, 3 only exist when NLA / NLB == 4
    const float* pa3 = CSS_ROWPTR(A, g.lda, m0 + lr + 3 * LROWS, M);
    const float* pb0 = CSS_ROWPTR(B, g.ldb, n0 + lr, N);
    const float* pb1 = CSS_ROWPTR(B, g.ldb, n0 + lr + LROWS, N);
    const float* pb2 = CSS_ROWPTR(B, g.ldb, n0 + lr + 2 * LROWS, N);
    const float* pb3 = CSS_ROWPTR(B, g.ldb, n0 + lr + 3 * LROWS, N);
#undef CSS_ROWPTR
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define CSS_GLOAD(k0)                                              \
    ra0 = *reinterpret_cast<const float4*>(pa0 + (k0));            \
    ra1 = *reinterpret_cast<const float4*>(pa1 + (k0));            \
    rb0 = *reinterpret_cast<const float4*>(pb0 + (k0));            \
    rb1 = *reinterpret_cast<const float4*>(pb1 + (k0));            \
    if constexpr (NLA == 4) {                                      \
        ra2 = *reinterpret_cast<const float4*>(pa2 + (k0));        \
        ra3 = *reinterpret_cast<const float4*>(pa3 + (k0));        \
    }                                                              \
    if constexpr (NLB == 4) {                                      \
        rb2 = *reinterpret_cast<const float4*>(pb2 + (k0));        \
        rb3 = *reinterpret_cast<const float4*>(pb3 + (k0));        \
    }
#define CSS_LSTORE(buf)                                                         \
    {                                                                           \
        float* as_ = lds + (buf) * (BM + BN) * LDS_LD + lr * LDS_LD + lc;       \
        float* bs_ = as_ + BM * LDS_LD;                                         \
        *reinterpret_cast<float4*>(as_) = ra0;                                  \
        *reinterpret_cast<float4*>(as_ + LROWS * LDS_LD) = ra1;                 \
        *reinterpret_cast<float4*>(bs_) = rb0;                                  \
        *reinterpret_cast<float4*>(bs_ + LROWS * LDS_LD) = rb1;                 \
        if constexpr (NLA == 4) {                                               \
            *reinterpret_cast<float4*>(as_ + 2 * LROWS * LDS_LD) = ra2;         \
            *reinterpret_cast<float4*>(as_ + 3 * LROWS * LDS_LD) = ra3;         \
        }                                                                       \
        if constexpr (NLB == 4) {                                               \
            *reinterpret_cast<float4*>(bs_ + 2 * LROWS * LDS_LD) = rb2;         \
            *reinterpret_cast<float4*>(bs_ + 3 * LROWS * LDS_LD) = rb3;         \
        }                                                                       \
    }

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};   // acc1x only used when TM == 2
    const int nk = g.K / BK;
    CSS_GLOAD(0)
    CSS_LSTORE(0)
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) { CSS_GLOAD((kt + 1) * BK) }
        const float* as = lds + buf * (BM + BN) * LDS_LD + (wm * (BM / WM) + c) * LDS_LD + 4 * h;
        const float* bs = lds + buf * (BM + BN) * LDS_LD + BM * LDS_LD + (wn * 64 + c) * LDS_LD + 4 * h;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const float4 a0 = *reinterpret_cast<const float4*>(as + ch * 8);
            const float4 b0 = *reinterpret_cast<const float4*>(bs + ch * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(bs + 32 * LDS_LD + ch * 8);
            float4 a1 = a0;
            if constexpr (TM == 2) a1 = *reinterpret_cast<const float4*>(as + 32 * LDS_LD + ch * 8);
#define CSS_MFMA_STEP(e)                                                                 \
    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.e, b0.e, acc00, 0, 0, 0);            \
    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.e, b1.e, acc01, 0, 0, 0);            \
    if constexpr (TM == 2) {                                                             \
        acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.e, b0.e, acc10, 0, 0, 0);        \
        acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.e, b1.e, acc11, 0, 0, 0);        \
    }
            CSS_MFMA_STEP(x) CSS_MFMA_STEP(y) CSS_MFMA_STEP(z) CSS_MFMA_STEP(w)
#undef CSS_MFMA_STEP
        }
        if (kt + 1 < nk) CSS_LSTORE(buf ^ 1)
        __syncthreads();
    }
#undef CSS_GLOAD
#undef CSS_LSTORE

    // ---- epilogue: bias, activation, scaled residual; 128-byte row segments per half wave ----
    const float* bias = g.bias;
    const float* res = g.residual;
    const int act = g.act, bias_m = g.bias_along_m;
    const int64_t ldc = g.ldc, ldr = g.ldr;
    const float alpha = g.alpha;
    const int mrow = m0 + wm * (BM / WM) + 4 * h, ncol = n0 + wn * 64 + c;
    // 16-byte stores through a wave-private LDS patch when the output rows allow it (gemm_common.hpp)
    const bool wide = (ldc % 4 == 0) && (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    if (wide) {
        __syncthreads();   // the slab buffers are free once every wave has left the K loop
        float* patch = lds + wave * (32 * LDS_LD);
        const int mt = m0 + wm * (BM / WM), nt = n0 + wn * 64;
        emit_tile_wide(acc00, mt, h, c, nt, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, 0, patch);
        emit_tile_wide(acc01, mt, h, c, nt + 32, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, 0, patch);
        if constexpr (TM == 2) {
            emit_tile_wide(acc10, mt + 32, h, c, nt, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, 0, patch);
            emit_tile_wide(acc11, mt + 32, h, c, nt + 32, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, 0, patch);
        }
        return;
    }
    emit_tile(acc00, mrow, ncol, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, 0);
    emit_tile(acc01, mrow, ncol + 32, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, 0);
    if constexpr (TM == 2) {
        emit_tile(acc10, mrow + 32, ncol, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, 0);
        emit_tile(acc11, mrow + 32, ncol + 32, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, 0);
    }
}

// Tile layouts (block tile M x 128, waves):
//   128 x 128, 8 waves (each 32x64)  default: two waves per SIMD from one block cover each other's barrier /
//                                    LDS bubbles when a launch has about one block per CU
//   128 x 128, 4 waves (each 64x64)  fewest LDS reads per MFMA; ~3 % ahead once >= 4 blocks per CU are queued
//    64 x 128, 4 waves (each 32x64)  twice the blocks: independent 4-wave blocks drift out of phase
// GemmArgs::layout = 8 | 4 | 64 forces one (unit tests, tools/gemm_bench.hip); every layout gives the same bits.
void launch_gemm(const GemmArgs& g_in, hipStream_t s) {
    if (g_in.M <= 0 || g_in.N <= 0 || g_in.batch <= 0 || g_in.K <= 0) return;
    if (g_in.split_in) return g_in.b_tiled ? launch_gemm_split_wd(g_in, s) : launch_gemm_split(g_in, s);
    GemmArgs g = g_in;
    int forced = g.layout;
    if (forced == 0 || forced == 1 || (forced >= 11 && forced <= 14)) {
        if (launch_gemm_f32(g, s, forced >= 11 ? forced - 10 : 0)) return;
        forced = 0;   // (an operand of 2 GiB and more: the kernel below addresses with 64-bit pointers)
    }
    if (g.b_frag32) {
        // the kernel below reads B row-major: a fragment-ordered weight image would be garbage without any error.  The caller
        // hands the row-major weight along (GemmArgs::B_rows); without it the launch is refused loudly instead of computed wrongly.
        if (!g.B_rows) {
            fprintf(stderr, "css_mi355: launch_gemm: a fragment-ordered weight (b_frag32) reached the row-major kernel without B_rows "
                            "(M %d N %d K %d layout %d): launch dropped\n", g.M, g.N, g.K, g.layout);
            return;
        }
        g.B = g.B_rows;
        g.b_frag32 = 0;
    }
    if (forced == 2) forced = 0;   // layout 2: this file's kernel with its own choice of tile (the round-4 default; A/B, tests)
    const int tiles_n = (g.N + BN - 1) / BN;
    const int blocks128 = ((g.M + 127) / 128) * tiles_n * g.batch;
    int layout = forced ? forced : (blocks128 < 1000 ? 8 : 4);
    if (!forced) {
        // short M (mask head: 1028 rows, analysis transform: 514): 128-row tiles would leave the last tile
        // nearly empty (12 % / 25 % padded work); 64-row tiles halve that (measured: head 119 -> 86 us,
        // STFT 192 -> 156 us)
        const double waste128 = (double)(((g.M + 127) / 128) * 128 - g.M) / g.M;
        const double waste64 = (double)(((g.M + 63) / 64) * 64 - g.M) / g.M;
        if (waste128 - waste64 > 0.03) layout = 64;
    }
    if (layout == 64) {
        const int tiles_m = (g.M + 63) / 64;
        hipLaunchKernelGGL((gemm_kernel<64, 2>), dim3(tiles_m * tiles_n * g.batch), dim3(256), 0, s, g, tiles_m, tiles_n);
    } else if (layout == 8) {
        const int tiles_m = (g.M + 127) / 128;
        hipLaunchKernelGGL((gemm_kernel<128, 4>), dim3(blocks128), dim3(512), 0, s, g, tiles_m, tiles_n);
    } else {
        const int tiles_m = (g.M + 127) / 128;
        hipLaunchKernelGGL((gemm_kernel<128, 2>), dim3(blocks128), dim3(256), 0, s, g, tiles_m, tiles_n);
    }
}

}  // namespace css

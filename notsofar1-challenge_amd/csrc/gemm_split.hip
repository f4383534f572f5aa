// Split-f16 MFMA GEMM for gfx950:  C[m][n] = epilogue( sum_k A[m][k] * B[n][k] ), float32-grade accuracy at
// f16 matrix-core rate.
//
// Used for the Conformer's Linear layers (embed / FFN / QKV / attention output / mask head:
// conformer.py:49-53,139-142,206,285), which are 99 % of the flops on the CSS path.  Both operands arrive in
// the split-f16 format of split_f16.hpp (x = hi + 2^-11 lo, two float16 numbers per float32 value); each
// product is THREE v_mfma_f32_32x32x16_f16 with float32 accumulation:
//      main += a_hi * b_hi          corr += a_hi * b_lo          corr += a_lo * b_hi
//      C = main + 2^-11 corr
// Products of f16 numbers are exact in float32 and the dropped a_lo*b_lo term is ~2^-22 of the product, so the
// result carries float32 accumulation rounding only (tools/split_f16_numerics.py; tests/test_hip_parity.py
// hold it to the same tolerances as the exact float32 kernel in gemm.hip).  16x the f32 MFMA rate / 3 products
// = 5.3x the float32 matrix peak.
//
// Shape: BM x 128 block tile, K slab 32.  One slab row of a split matrix is 64 B of hi + 64 B of lo = the same
// 128-byte line as 32 floats, so global staging and the LDS image (rows padded to 36 floats: ds_read_b128 of
// 16 consecutive rows is conflict free) are those of the float32 kernel, and a lane's MFMA operand
// (k = 8h..8h+7) is one 16-byte LDS read.  With 5.3x less MFMA time per slab the loop is bound by global-load
// latency rather than by the matrix cores, so global loads run TWO slabs ahead of the MFMAs (two register
// stages, loop unrolled by two) in front of the double-buffered LDS image; the loop body is branch free (the
// last iterations re-load the final slab) so that the compiler's vmcnt accounting keeps the younger stage in
// flight while the older one is stored.
#include <cstdlib>

#include "gemm_common.hpp"

namespace css {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CSS_LDH(p) __builtin_bit_cast(f16x8, *reinterpret_cast<const float4*>(p))
#define CSS_MFMA16(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

// the 24 (TM = 2) / 12 (TM = 1) MFMAs of one K slab for one wave; as / bs = this lane's LDS rows (+ 4h floats)
template <int TM>
__device__ __forceinline__ void slab_mfma(const float* as, const float* bs, f32x16& acc00, f32x16& acc01, f32x16& acc10,
                                          f32x16& acc11, f32x16& cor00, f32x16& cor01, f32x16& cor10, f32x16& cor11) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const f16x8 ah0 = CSS_LDH(as + kk * 8), al0 = CSS_LDH(as + kk * 8 + 16);
        const f16x8 bh0 = CSS_LDH(bs + kk * 8), bl0 = CSS_LDH(bs + kk * 8 + 16);
        const f16x8 bh1 = CSS_LDH(bs + 32 * LDS_LD + kk * 8), bl1 = CSS_LDH(bs + 32 * LDS_LD + kk * 8 + 16);
        f16x8 ah1 = ah0, al1 = al0;
        if constexpr (TM == 2) {
            ah1 = CSS_LDH(as + 32 * LDS_LD + kk * 8);
            al1 = CSS_LDH(as + 32 * LDS_LD + kk * 8 + 16);
        }
        {
        CSS_MFMA16(ah0, bh0, acc00);
        CSS_MFMA16(ah0, bh1, acc01);
        if constexpr (TM == 2) { CSS_MFMA16(ah1, bh0, acc10); CSS_MFMA16(ah1, bh1, acc11); }
        CSS_MFMA16(ah0, bl0, cor00);
        CSS_MFMA16(ah0, bl1, cor01);
        if constexpr (TM == 2) { CSS_MFMA16(ah1, bl0, cor10); CSS_MFMA16(ah1, bl1, cor11); }
        CSS_MFMA16(al0, bh0, cor00);
        CSS_MFMA16(al0, bh1, cor01);
        if constexpr (TM == 2) { CSS_MFMA16(al1, bh0, cor10); CSS_MFMA16(al1, bh1, cor11); }
        }
    }
}

template <int BM, int WM>
__global__ __launch_bounds__(WM * 128, 2) void gemm_split_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    constexpr int THREADS = WM * 128;          // WM x 2 waves
    constexpr int TM = (BM / WM) / 32;         // MFMA tiles along M per wave (2 or 1)
    constexpr int LROWS = THREADS / 8;         // rows covered by one staging pass (32 or 64)
    constexpr int NLA = BM / LROWS;            // staging passes of the A tile (2 or 4)
    constexpr int NLB = BN / LROWS;            // staging passes of the B tile (2 or 4)
    static_assert((NLA == 2 || NLA == 4) && (NLB == 2 || NLB == 4) && (TM == 1 || TM == 2), "unsupported tile layout");
    constexpr int STAGE = (BM + BN) * LDS_LD;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
    const int n_tiles = tiles_m * tiles_n * g.batch;
    const int tile = xcd_tile(blockIdx.x, n_tiles);
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
    const int lc = (tid & 7) << 2;    // 16-byte chunk of the 128-byte slab row (chunks 0-3 hi, 4-7 lo)
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int c = lane & 31, h = lane >> 5;
    const int M = g.M, N = g.N;

    // rows past M / N re-read the last valid row (in bounds, finite, never stored by the epilogue)
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
    float4 Ra0, Ra1, Ra2, Ra3, Rb0, Rb1, Rb2, Rb3;   // register stage R
    float4 Sa0, Sa1, Sa2, Sa3, Sb0, Sb1, Sb2, Sb3;   // register stage S
#define CSS_GLOAD(P, k0)                                              \
    P##a0 = *reinterpret_cast<const float4*>(pa0 + (k0));             \
    P##a1 = *reinterpret_cast<const float4*>(pa1 + (k0));             \
    P##b0 = *reinterpret_cast<const float4*>(pb0 + (k0));             \
    P##b1 = *reinterpret_cast<const float4*>(pb1 + (k0));             \
    if constexpr (NLA == 4) {                                         \
        P##a2 = *reinterpret_cast<const float4*>(pa2 + (k0));         \
        P##a3 = *reinterpret_cast<const float4*>(pa3 + (k0));         \
    }                                                                 \
    if constexpr (NLB == 4) {                                         \
        P##b2 = *reinterpret_cast<const float4*>(pb2 + (k0));         \
        P##b3 = *reinterpret_cast<const float4*>(pb3 + (k0));         \
    }
#define CSS_LSTORE(P, buf)                                                      \
    {                                                                           \
        float* as_ = lds + (buf) * STAGE + lr * LDS_LD + lc;                    \
        float* bs_ = as_ + BM * LDS_LD;                                         \
        *reinterpret_cast<float4*>(as_) = P##a0;                                \
        *reinterpret_cast<float4*>(as_ + LROWS * LDS_LD) = P##a1;               \
        *reinterpret_cast<float4*>(bs_) = P##b0;                                \
        *reinterpret_cast<float4*>(bs_ + LROWS * LDS_LD) = P##b1;               \
        if constexpr (NLA == 4) {                                               \
            *reinterpret_cast<float4*>(as_ + 2 * LROWS * LDS_LD) = P##a2;       \
            *reinterpret_cast<float4*>(as_ + 3 * LROWS * LDS_LD) = P##a3;       \
        }                                                                       \
        if constexpr (NLB == 4) {                                               \
            *reinterpret_cast<float4*>(bs_ + 2 * LROWS * LDS_LD) = P##b2;       \
            *reinterpret_cast<float4*>(bs_ + 3 * LROWS * LDS_LD) = P##b3;       \
        }                                                                       \
    }

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};   // x1x only used when TM == 2
    f32x16 cor00 = {0}, cor01 = {0}, cor10 = {0}, cor11 = {0};   // the 2^-11-scaled cross terms
    const int nk = g.K / BK;
    const int klast = (nk - 1) * BK;
    const float* as0 = lds + (wm * (BM / WM) + c) * LDS_LD + 4 * h;
    const float* bs0 = lds + BM * LDS_LD + (wn * 64 + c) * LDS_LD + 4 * h;
#define CSS_KOFF(kt_) (((kt_) * BK) < klast ? ((kt_) * BK) : klast)

    CSS_GLOAD(R, 0)
    CSS_GLOAD(S, CSS_KOFF(1))
    CSS_LSTORE(R, 0)
    __syncthreads();
    // invariant at the top of step kt: LDS[kt & 1] holds slab kt, stage S (even kt) / R (odd kt) holds slab kt + 1
    for (int kt = 0;;) {
        CSS_GLOAD(R, CSS_KOFF(kt + 2))
        slab_mfma<TM>(as0, bs0, acc00, acc01, acc10, acc11, cor00, cor01, cor10, cor11);
        CSS_LSTORE(S, 1)
        __syncthreads();
        if (++kt >= nk) break;
        CSS_GLOAD(S, CSS_KOFF(kt + 2))
        slab_mfma<TM>(as0 + STAGE, bs0 + STAGE, acc00, acc01, acc10, acc11, cor00, cor01, cor10, cor11);
        CSS_LSTORE(R, 0)
        __syncthreads();
        if (++kt >= nk) break;
    }
#undef CSS_KOFF
#undef CSS_GLOAD
#undef CSS_LSTORE

    // ---- epilogue: merge the cross terms, then bias / activation / scaled residual as in gemm.hip ----
    const float* bias = g.bias;
    const float* res = g.residual;
    const int act = g.act, bias_m = g.bias_along_m, so = g.split_out;
    const int64_t ldc = g.ldc, ldr = g.ldr;
    const float alpha = g.alpha;
    const int mrow = m0 + wm * (BM / WM) + 4 * h, ncol = n0 + wn * 64 + c;
    acc00 += cor00 * SPLIT_LO_INV;
    acc01 += cor01 * SPLIT_LO_INV;
    if constexpr (TM == 2) {   // (before either epilogue: the 16-byte one used to emit these two tiles without their cross terms)
        acc10 += cor10 * SPLIT_LO_INV;
        acc11 += cor11 * SPLIT_LO_INV;
    }
    if (g.range_flag) {
        range_check(acc00, g.range_flag);
        range_check(acc01, g.range_flag);
        if constexpr (TM == 2) { range_check(acc10, g.range_flag); range_check(acc11, g.range_flag); }
    }
    // 16-byte stores through a wave-private LDS patch when the output rows allow it (gemm_common.hpp)
    const bool wide = (ldc % 4 == 0) && (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    if (wide) {
        __syncthreads();   // the slab buffers are free once every wave has left the K loop
        float* patch = lds + wave * (32 * LDS_LD);
        const int mt = m0 + wm * (BM / WM), nt = n0 + wn * 64;
        emit_tile_wide(acc00, mt, h, c, nt, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, so, patch);
        emit_tile_wide(acc01, mt, h, c, nt + 32, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, so, patch);
        if constexpr (TM == 2) {
            emit_tile_wide(acc10, mt + 32, h, c, nt, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, so, patch);
            emit_tile_wide(acc11, mt + 32, h, c, nt + 32, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, so, patch);
        }
        return;
    }
    emit_tile(acc00, mrow, ncol, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, so);
    emit_tile(acc01, mrow, ncol + 32, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, so);
    if constexpr (TM == 2) {
        emit_tile(acc10, mrow + 32, ncol, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, so);
        emit_tile(acc11, mrow + 32, ncol + 32, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, so);
    }
}

// float32 [rows][K] (row stride ld_src) -> split-f16 [rows][Kp] (Kp % 32 == 0, zero padded past K)
__global__ __launch_bounds__(256) void split_convert_kernel(const float* __restrict__ src, int64_t ld_src,
                                                            float* __restrict__ dst, int64_t rows, int K, int Kp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one thread per 4 elements
    const int per_row = Kp >> 2;
    if (i >= rows * per_row) return;
    const int64_t r = i / per_row;
    const int k = (int)(i - r * per_row) << 2;
    const float* s = src + r * ld_src + k;
    const float x0 = k < K ? s[0] : 0.f, x1 = k + 1 < K ? s[1] : 0.f, x2 = k + 2 < K ? s[2] : 0.f, x3 = k + 3 < K ? s[3] : 0.f;
    split_store4(reinterpret_cast<_Float16*>(dst + r * Kp), k, x0, x1, x2, x3);
}

void launch_split_convert(const float* src, int64_t ld_src, float* dst, int64_t rows, int K, int Kp, hipStream_t s) {
    const int64_t n = rows * (Kp >> 2);
    if (n <= 0) return;
    hipLaunchKernelGGL(split_convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, ld_src, dst, rows, K, Kp);
}

// Layouts as in gemm.hip (128x128 with 8 or 4 waves, 64x128 with 4 waves); GemmArgs::layout = 8 | 4 | 64 forces one.
void launch_gemm_split(const GemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return;
    const int forced = g.layout;
    const int tiles_n = (g.N + BN - 1) / BN;
    const int blocks128 = ((g.M + 127) / 128) * tiles_n * g.batch;
    int layout = forced ? forced : (blocks128 < 1000 ? 8 : 4);
    if (!forced) {
        const double waste128 = (double)(((g.M + 127) / 128) * 128 - g.M) / g.M;
        const double waste64 = (double)(((g.M + 63) / 64) * 64 - g.M) / g.M;
        if (waste128 - waste64 > 0.03) layout = 64;
    }
    if (layout == 64) {
        const int tiles_m = (g.M + 63) / 64;
        hipLaunchKernelGGL((gemm_split_kernel<64, 2>), dim3(tiles_m * tiles_n * g.batch), dim3(256), 0, s, g, tiles_m, tiles_n);
    } else if (layout == 8) {
        const int tiles_m = (g.M + 127) / 128;
        hipLaunchKernelGGL((gemm_split_kernel<128, 4>), dim3(blocks128), dim3(512), 0, s, g, tiles_m, tiles_n);
    } else {
        const int tiles_m = (g.M + 127) / 128;
        hipLaunchKernelGGL((gemm_split_kernel<128, 2>), dim3(blocks128), dim3(256), 0, s, g, tiles_m, tiles_n);
    }
}

}  // namespace css

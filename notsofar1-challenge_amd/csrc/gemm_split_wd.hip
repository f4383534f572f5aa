// Split-f16 MFMA GEMM with the WEIGHT operand read straight from global memory into MFMA registers.
//
//   C[m][n] = epilogue( sum_k A[m][k] * W[n][k] )      A: activations, row-major split-f16 (split_f16.hpp)
//                                                       W: weights, tile-major split-f16 (below)
//
// Why: in gemm_split.hip both operands go global -> registers -> LDS -> registers, and with three cheap f16 MFMAs
// per product the LDS (13 cycles per ds_write_b128, 4 reads per 3 MFMAs) -- not the matrix core -- is the busiest
// unit (measured in round 1 with a timing probe: removing the LDS stores alone shortens the K loop by 27 %).  The weights are
// static, so css_create lays them out ONCE in the order the MFMA wants them: the 16 bytes lane l feeds to
// v_mfma_f32_32x32x16_f16 for column tile j, k group kk, part p (hi / lo) live at
//        float4 index ((j * K/16 + kk) * 2 + p) * 64 + l        (row j*32 + l%32 of W, k = 16 kk + 8 (l/32) .. +7)
// i.e. one fully coalesced 1 KiB load per operand and 4 KiB of sequential stream per wave and K slab, no LDS.
// A wave owns 32 output columns (its own weight stream) and 64 or 128 rows of the 128-row tile; only the activation
// tile is staged through LDS (half the LDS stores of gemm_split.hip, 2/3 of its LDS reads).  Arithmetic,
// accumulation order within a k group and epilogue are those of gemm_split.hip -- results are bit-identical to it.
// Measured (tools/gemm_bench.hip, M = 7440): 5-10 % faster than gemm_split.hip on the Conformer's shapes; the K loop
// runs at 0.62 us per 32-k slab against 0.39 us for the same 24 MFMAs per SIMD with no memory instructions at all
// (tools/mfma_rate.hip; the sustained MFMA clock is ~2.0 GHz) and 0.69 us in gemm_split.hip.
#include <cstdlib>

#include "gemm_common.hpp"

namespace css {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CSS_LDH(p) __builtin_bit_cast(f16x8, *reinterpret_cast<const float4*>(p))
#define CSS_MFMA16(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

constexpr int WD_BM = 128;

// WMV = waves along M: 1 -> 4 waves of 128 x 32 (one per SIMD), 2 -> 8 waves of 64 x 32 (two per SIMD; each weight
// operand is then fetched by two waves, the second from L1).
//
// All A operands of a slab live in registers (two sets alternating per slab), read from LDS one slab AHEAD of the
// MFMAs that use them, and the loop body between two barriers is one basic block in which the LDS reads of slab
// kt+1, the LDS stores of slab kt+2 and the global loads of slab kt+3 / weights kt+1 are spread between the MFMAs
// of slab kt (sched_group_barrier pattern): an MFMA leaves ~5 issue slots free while it runs, and with one or two
// waves per SIMD nothing else would hide those instructions.
// BUF: operand loads as buffer loads (resource descriptor in SGPRs + 32-bit lane offset + scalar slab offset) instead of
// 64-bit-per-lane global loads: half the address registers through the issue path, no 64-bit pointer arithmetic per slab
// TR: the result is written transposed (GemmArgs::c_transposed; its own instantiation, so that the Linear layers' kernel
// is the code it was)
template <int BM, int WMV, bool BUF = false, bool TR = false>
__global__ __launch_bounds__(WMV * 256, (BM / WMV) >= 96 ? 1 : (BM / WMV) >= 64 ? 2 : 3) void gemm_split_wd_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    constexpr int THREADS = WMV * 256;
    constexpr int TM = BM / 32 / WMV;           // 32-row tiles per wave
    constexpr int LROWS = THREADS / 8;          // rows per staging pass
    constexpr int NLA = BM / LROWS;             // staging passes (4 or 2)
    constexpr int STAGE = BM * LDS_LD;
    constexpr int PATCHES = WMV * 4 * 32 * LDS_LD;   // the epilogue's wave-private patches reuse the slab buffers
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE > PATCHES ? 2 * STAGE : PATCHES];
    const int n_tiles = tiles_m * tiles_n;
    const int tile = xcd_tile(blockIdx.x, n_tiles);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const float* __restrict__ A = g.A;
    float* __restrict__ C = g.C;
    const int tid = threadIdx.x;
    const int lr = tid >> 3;          // row within a staging pass
    const int lc = (tid & 7) << 2;    // 16-byte chunk of the 128-byte slab row
    const int wave = tid >> 6, lane = tid & 63;
    const int wn = wave & 3, wm = wave >> 2;
    const int c = lane & 31, h = lane >> 5;
    const int M = g.M, N = g.N;
    const int nk = g.K / BK;
    const int klast = (nk - 1) * BK;

    // Everything below is spelled out over named registers (index lists expanded by the CSS_Ix macros, entries
    // beyond this instantiation's counts dropped by `if constexpr`): arrays indexed inside unrolled loops were left
    // in scratch memory by the compiler.
#define CSS_I4(M, ...) M(0, __VA_ARGS__) M(1, __VA_ARGS__) M(2, __VA_ARGS__) M(3, __VA_ARGS__)
#define CSS_I8(M, ...) CSS_I4(M, __VA_ARGS__) M(4, __VA_ARGS__) M(5, __VA_ARGS__) M(6, __VA_ARGS__) M(7, __VA_ARGS__)
    // rows past M re-read the last valid row (in bounds, finite, never stored by the epilogue)
#define CSS_ROWPTR(i) (A + (int64_t)((m0 + lr + (i) * LROWS) < M ? (m0 + lr + (i) * LROWS) : M - 1) * g.lda + lc)
    const float* pa0 = CSS_ROWPTR(0);
    const float* pa1 = CSS_ROWPTR(1);
    const float* pa2 = CSS_ROWPTR(NLA > 2 ? 2 : 0);
    const float* pa3 = CSS_ROWPTR(NLA > 2 ? 3 : 0);
#undef CSS_ROWPTR
    // this wave's weight stream: column tile jt, 4 operands (kk0 hi, kk0 lo, kk1 hi, kk1 lo) per K slab
    const int jt_max = (N + 31) / 32 - 1;
    const int jt = min(n0 / 32 + wn, jt_max);
    const float4* __restrict__ pw = reinterpret_cast<const float4*>(g.B) + (int64_t)jt * (g.K / 16) * 2 * 64 + lane;

    float4 stg0_0, stg0_1, stg0_2, stg0_3, stg1_0, stg1_1, stg1_2, stg1_3;   // activation staging, two register stages
    float4 wr0_0, wr0_1, wr0_2, wr0_3, wr1_0, wr1_1, wr1_2, wr1_3, wr2_0, wr2_1, wr2_2, wr2_3;   // weight operands of three K slabs
    // A operands of two K slabs: index = row tile + 4 * kk
    f16x8 ah0_0, ah0_1, ah0_2, ah0_3, ah0_4, ah0_5, ah0_6, ah0_7, al0_0, al0_1, al0_2, al0_3, al0_4, al0_5, al0_6, al0_7;
    f16x8 ah1_0, ah1_1, ah1_2, ah1_3, ah1_4, ah1_5, ah1_6, ah1_7, al1_0, al1_1, al1_2, al1_3, al1_4, al1_5, al1_6, al1_7;
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};   // row tiles x this wave's 32 columns
    f32x16 cor0 = {0}, cor1 = {0}, cor2 = {0}, cor3 = {0};   // the 2^-11-scaled cross terms
#define CSS_KOFF(kt_) (((kt_) * BK) < klast ? ((kt_) * BK) : klast)
#define CSS_KT(kt_) ((kt_) < nk ? (kt_) : nk - 1)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, (int)((int64_t)M * g.lda * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, (int)((int64_t)(jt_max + 1) * 32 * g.K * 4), 0x00020000);
#define CSS_VOFFA(i) (unsigned)(((int64_t)((m0 + lr + (i) * LROWS) < M ? (m0 + lr + (i) * LROWS) : M - 1) * g.lda + lc) * 4)
    const unsigned va0 = CSS_VOFFA(0), va1 = CSS_VOFFA(1), va2 = CSS_VOFFA(NLA > 2 ? 2 : 0), va3 = CSS_VOFFA(NLA > 2 ? 3 : 0);
#undef CSS_VOFFA
    const unsigned vw = (unsigned)(((int64_t)jt * (g.K / 16) * 2 * 64 + lane) * 16);
#define CSS_G1(i, st, k0)                                                                                                  \
    if constexpr (i < NLA) {                                                                                               \
        if constexpr (BUF) stg##st##_##i = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, va##i, (k0) * 4, 0)); \
        else stg##st##_##i = *reinterpret_cast<const float4*>(pa##i + (k0));                                               \
    }
#define CSS_GLOAD(st, k0) CSS_I4(CSS_G1, st, k0)
#define CSS_L1(i, st, buf) \
    if constexpr (i < NLA) *reinterpret_cast<float4*>(lds + (buf) * STAGE + (lr + i * LROWS) * LDS_LD + lc) = stg##st##_##i;
#define CSS_LSTORE(st, buf) CSS_I4(CSS_L1, st, buf)
#define CSS_W1(i, st, q_) wr##st##_##i = (q_)[i * 64];
#define CSS_W1B(i, st, so_) wr##st##_##i = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsW, vw + i * 1024, so_, 0));
#define CSS_WLOAD(st, kt_)                                                                       \
    {                                                                                            \
        if constexpr (BUF) { const int so_ = (kt_) * 4096; CSS_I4(CSS_W1B, st, so_) }            \
        else { const float4* q_ = pw + (int64_t)(kt_) * 4 * 64; CSS_I4(CSS_W1, st, q_) }         \
    }
    const float* as0 = lds + (wm * (BM / WMV) + c) * LDS_LD + 4 * h;
    // operand o = row tile (o & 3) + 4 * kk (o >> 2)
#define CSS_A1(o, st, buf)                                                                                    \
    if constexpr ((o & 3) < TM) {                                                                             \
        ah##st##_##o = CSS_LDH(as0 + (buf) * STAGE + (o & 3) * 32 * LDS_LD + (o >> 2) * 8);                   \
        CSS_A1_LO(o, st, buf)                                                                                 \
    }
#define CSS_A1_LO(o, st, buf) al##st##_##o = CSS_LDH(as0 + (buf) * STAGE + (o & 3) * 32 * LDS_LD + (o >> 2) * 8 + 16);
#define CSS_AREAD(st, buf) CSS_I8(CSS_A1, st, buf)
#define CSS_AREAD_LOOP(st, buf) CSS_AREAD(st, buf)
    // the 6 * TM MFMAs of one slab; each accumulator is touched every TM-th MFMA
#define CSS_M1(t, st, kk, part, w_, dst) if constexpr (t < TM) CSS_MFMA16(part##st##_##kk##t, w_, dst##t);
#define CSS_SLAB_COR(st, o0, o1, o2, o3)                                                              \
        if constexpr (0 < TM) CSS_MFMA16(ah##st##_##o0, wl_, cor0);                                   \
        if constexpr (1 < TM) CSS_MFMA16(ah##st##_##o1, wl_, cor1);                                   \
        if constexpr (2 < TM) CSS_MFMA16(ah##st##_##o2, wl_, cor2);                                   \
        if constexpr (3 < TM) CSS_MFMA16(ah##st##_##o3, wl_, cor3);                                   \
        if constexpr (0 < TM) CSS_MFMA16(al##st##_##o0, wh_, cor0);                                   \
        if constexpr (1 < TM) CSS_MFMA16(al##st##_##o1, wh_, cor1);                                   \
        if constexpr (2 < TM) CSS_MFMA16(al##st##_##o2, wh_, cor2);                                   \
        if constexpr (3 < TM) CSS_MFMA16(al##st##_##o3, wh_, cor3);
#define CSS_SLAB_KK(st, ws, kk, o0, o1, o2, o3, wi0, wi1)                                                \
    {                                                                                                 \
        const f16x8 wh_ = __builtin_bit_cast(f16x8, wr##ws##_##wi0);                                  \
        const f16x8 wl_ = __builtin_bit_cast(f16x8, wr##ws##_##wi1);                                  \
        if constexpr (0 < TM) CSS_MFMA16(ah##st##_##o0, wh_, acc0);                                   \
        if constexpr (1 < TM) CSS_MFMA16(ah##st##_##o1, wh_, acc1);                                   \
        if constexpr (2 < TM) CSS_MFMA16(ah##st##_##o2, wh_, acc2);                                   \
        if constexpr (3 < TM) CSS_MFMA16(ah##st##_##o3, wh_, acc3);                                   \
        CSS_SLAB_COR(st, o0, o1, o2, o3)                                                                  \
    }
#define CSS_SLAB(st, ws) CSS_SLAB_KK(st, ws, 0, 0, 1, 2, 3, 0, 1) CSS_SLAB_KK(st, ws, 1, 4, 5, 6, 7, 2, 3)
    // issue order inside a step: 2*TM x {MFMA, 2 LDS reads, 1 global load}, NLA x {MFMA, 1 LDS store}, the rest MFMAs
#define CSS_INTERLEAVE()                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 2 * TM; ++i_) {                \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                 \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                 \
    }                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < NLA; ++i_) {                   \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                 \
    }                                                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 6 * TM - 2 * TM - NLA, 0); \
    __builtin_amdgcn_sched_barrier(0);   /* the barrier stays behind the last MFMA: the pipe drains while waiting */

#define CSS_LOOP_BARRIER() __syncthreads()
    // epilogue operands (column bias, residual) are requested now and used after the K loop (gemm_common.hpp)
    const int mrow = m0 + wm * (BM / WMV) + 4 * h, ncol = n0 + wn * 32 + c;
    TilePre pre0, pre1, pre2, pre3;
    const bool early = !g.bias_along_m;
    if (early) {
        if constexpr (0 < TM) tile_prefetch(pre0, mrow, ncol, M, N, g.bias, g.residual, g.ldr);
        if constexpr (1 < TM) tile_prefetch(pre1, mrow + 32, ncol, M, N, g.bias, g.residual, g.ldr);
        if constexpr (2 < TM) tile_prefetch(pre2, mrow + 64, ncol, M, N, g.bias, g.residual, g.ldr);
        if constexpr (3 < TM) tile_prefetch(pre3, mrow + 96, ncol, M, N, g.bias, g.residual, g.ldr);
    }

    CSS_GLOAD(0, 0)
    CSS_WLOAD(0, 0)
    CSS_WLOAD(1, CSS_KT(1))
    CSS_GLOAD(1, CSS_KOFF(1))
    CSS_LSTORE(0, 0)
    CSS_GLOAD(0, CSS_KOFF(2))
    CSS_LSTORE(1, 1)
    __syncthreads();
    CSS_AREAD(0, 0)
    __syncthreads();   // every wave holds slab 0 in registers before slab 2 overwrites LDS[0]
    // invariant at the top of step kt: operand set kt % 2 = slab kt, LDS[(kt + 1) % 2] = slab kt + 1, stage kt % 2 =
    // slab kt + 2, weight sets kt % 3 and (kt + 1) % 3 = weights kt, kt + 1.  The weights are requested TWO slabs ahead
    // (a third register set): a CU's 8 waves keep 48 KB of operands per slab on the 64 B/clk L1 return path, a load
    // lands 0.4-0.6 us after it was issued, and with one slab of lead every wave sat at the top of each slab waiting
    // for its weights while the matrix core idled.  Periods 2 (operands, LDS) and 3 (weights) -> six steps per round.
#define CSS_STEP(a, an, w, wn)                     \
        CSS_GLOAD(an, CSS_KOFF(kt + 3))            \
        CSS_WLOAD(wn, CSS_KT(kt + 2))              \
        CSS_AREAD_LOOP(an, an)                     \
        CSS_SLAB(a, w)                             \
        CSS_LSTORE(a, a)                           \
        CSS_INTERLEAVE()                           \
        CSS_LOOP_BARRIER();                        \
        if (++kt >= nk) break;
    for (int kt = 0;;) {
        CSS_STEP(0, 1, 0, 2)
        CSS_STEP(1, 0, 1, 0)
        CSS_STEP(0, 1, 2, 1)
        CSS_STEP(1, 0, 0, 2)
        CSS_STEP(0, 1, 1, 0)
        CSS_STEP(1, 0, 2, 1)
    }
#undef CSS_STEP

    const float* bias = g.bias;
    const float* res = g.residual;
    const int act = g.act, bias_m = g.bias_along_m, so = g.split_out;
    const int64_t ldc = g.ldc, ldr = g.ldr;
    const float alpha = g.alpha;
    // wide epilogue (gemm_common.hpp): each wave's finished tiles pass through its own [32][LDS_LD] patch of the slab
    // buffers, which nobody reads any more after the loop's last barrier
    float* patch = lds + wave * (32 * LDS_LD);
    const int mtile = m0 + wm * (BM / WMV), ntile = n0 + wn * 32;
    const bool wide = early && !g.narrow_epilogue;
    const bool fragq = g.frag_out != nullptr && ntile < 2 * g.frag_D;   // wave-uniform: a wave owns 32 columns
#define CSS_E1(t, ...)                                                                                      \
    if constexpr (t < TM) {                                                                                 \
        acc##t += cor##t * SPLIT_LO_INV;                                                                    \
        if (g.range_flag) range_check(acc##t, g.range_flag);                                                \
        if constexpr (TR) emit_tile_pre_tr(acc##t, pre##t.bn, mtile + 32 * t, h, c, ntile, M, N, C, ldc, act, patch);   \
        else if (fragq) emit_tile_frag(acc##t, pre##t.bn, mtile + 32 * t, h, c, M, g.frag_out, g.frag_T, g.frag_invT, g.frag_heads, \
                                  (ntile % g.frag_D) >> 6, ntile / g.frag_D, (ntile >> 5) & 1, patch);          \
        else if (wide) emit_tile_pre_wide(acc##t, pre##t, mtile + 32 * t, h, c, ntile, M, N, C, ldc, act, res != nullptr, alpha, so, \
                                     g.nt_store, patch);                                                    \
        else if (early) emit_tile_pre(acc##t, pre##t, mrow + 32 * t, ncol, M, N, C, ldc, act, res != nullptr, alpha, so, g.nt_store); \
        else emit_tile(acc##t, mrow + 32 * t, ncol, M, N, C, ldc, bias, bias_m, act, res, ldr, alpha, so);  \
    }
    CSS_I4(CSS_E1, 0)
#undef CSS_E1
#undef CSS_AREAD
#undef CSS_AREAD_LOOP
#undef CSS_LOOP_BARRIER
#undef CSS_A1
#undef CSS_A1_LO
#undef CSS_INTERLEAVE
#undef CSS_SLAB
#undef CSS_SLAB_KK
#undef CSS_SLAB_COR
#undef CSS_M1
#undef CSS_KOFF
#undef CSS_KT
#undef CSS_GLOAD
#undef CSS_G1
#undef CSS_LSTORE
#undef CSS_L1
#undef CSS_WLOAD
#undef CSS_W1
#undef CSS_W1B
#undef CSS_I4
#undef CSS_I8
}

// float32 W [N][K] (row stride ld_src, K % 32 == 0) -> tile-major split-f16 (see the header comment); rows past N of
// the last column tile are zero.  One thread per (row, 8-k group).
__global__ __launch_bounds__(256) void split_convert_tiled_kernel(const float* __restrict__ src, int64_t ld_src,
                                                                  float* __restrict__ dst, int N, int K) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int groups = K >> 3;
    const int npad = (N + 31) / 32 * 32;
    if (i >= (int64_t)npad * groups) return;
    const int n = (int)(i / groups), kg = (int)(i - (int64_t)n * groups);
    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
    f16x8v hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = n < N ? src[(int64_t)n * ld_src + kg * 8 + e] : 0.f;
        _Float16 a, b;
        split_f16(x, a, b);
        hi[e] = a; lo[e] = b;
    }
    const int j = n >> 5, kk = kg >> 1, l = (n & 31) + 32 * (kg & 1);
    float4* d = reinterpret_cast<float4*>(dst) + ((int64_t)(j * (K / 16) + kk) * 2) * 64 + l;
    d[0] = __builtin_bit_cast(float4, hi);
    d[64] = __builtin_bit_cast(float4, lo);
}

void launch_split_convert_tiled(const float* src, int64_t ld_src, float* dst, int N, int K, hipStream_t s) {
    const int64_t n = (int64_t)((N + 31) / 32 * 32) * (K >> 3);
    if (n <= 0) return;
    hipLaunchKernelGGL(split_convert_tiled_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, ld_src, dst, N, K);
}

void launch_gemm_split_wd(const GemmArgs& g_in, hipStream_t s) {
    if (g_in.M <= 0 || g_in.N <= 0) return;
    const GemmArgs& g = g_in;
    const int tiles_m = (g.M + WD_BM - 1) / WD_BM, tiles_n = (g.N + BN - 1) / BN;
    int pick = g.tile_rows;
    // (buffer loads -- BUF = true -- measured 1-6 % ahead of 64-bit global loads on every shape of the path, same bits;
    // tile_rows = 65 is the global-load form of the 64-row kernel: A/B timing, and the operands a 32-bit buffer offset
    // cannot address -- an activation or weight extent of 2 GiB and more, i.e. M * lda beyond 2^29 floats)
    const int64_t a_bytes = (int64_t)g.M * g.lda * (int64_t)sizeof(float), w_bytes = (int64_t)((g.N + 31) / 32 * 32) * g.K * (int64_t)sizeof(float);
    if (a_bytes >= (int64_t)1 << 31 || w_bytes >= (int64_t)1 << 31) pick = 65;
    if (g.c_transposed) {   // (column bias + activation only)
        const int tm64 = (g.M + 63) / 64;
        if (pick == 65)   // an operand beyond the 32-bit buffer offsets (>= 2 GiB): the 64-bit global-load form, same bits
            hipLaunchKernelGGL((gemm_split_wd_kernel<64, 1, false, true>), dim3(tm64 * tiles_n), dim3(256), 0, s, g, tm64, tiles_n);
        else
            hipLaunchKernelGGL((gemm_split_wd_kernel<64, 1, true, true>), dim3(tm64 * tiles_n), dim3(256), 0, s, g, tm64, tiles_n);
        return;
    }
    if (pick == 32) {
        const int tm32 = (g.M + 31) / 32;
        hipLaunchKernelGGL((gemm_split_wd_kernel<32, 1, true>), dim3(tm32 * tiles_n), dim3(256), 0, s, g, tm32, tiles_n);
    } else if (pick == 96) {
        const int tm96 = (g.M + 95) / 96;
        hipLaunchKernelGGL((gemm_split_wd_kernel<96, 1, true>), dim3(tm96 * tiles_n), dim3(256), 0, s, g, tm96, tiles_n);
    } else if (pick == 4) {
        hipLaunchKernelGGL((gemm_split_wd_kernel<128, 1, true>), dim3(tiles_m * tiles_n), dim3(256), 0, s, g, tiles_m, tiles_n);
    } else if (pick == 65) {
        const int tm64 = (g.M + 63) / 64;
        hipLaunchKernelGGL((gemm_split_wd_kernel<64, 1, false>), dim3(tm64 * tiles_n), dim3(256), 0, s, g, tm64, tiles_n);
    } else if (pick == 64 || !pick) {   // best or equal on every shape of the path, alone or beside another launch (tools/gemm_tile_bench.hip)
        const int tm64 = (g.M + 63) / 64;
        hipLaunchKernelGGL((gemm_split_wd_kernel<64, 1, true>), dim3(tm64 * tiles_n), dim3(256), 0, s, g, tm64, tiles_n);
    } else {
        hipLaunchKernelGGL((gemm_split_wd_kernel<128, 2, true>), dim3(tiles_m * tiles_n), dim3(512), 0, s, g, tiles_m, tiles_n);
    }
}

}  // namespace css

// Split-f16 MFMA GEMM with BOTH operands brought into LDS by the DMA path (global_load_lds), 128 x 128 output tiles.
//
//   C[m][n] = epilogue( sum_k A[m][k] * W[n][k] )      A: activations, row-major split-f16 (split_f16.hpp)
//                                                       W: weights, tile-major split-f16 (gemm_split_wd.hip)
//
// Why a third kernel.  gemm_split_wd.hip reads the weights straight into MFMA registers, one private stream per wave:
// a 64 x 128 block pulls 64 * (64 + 128) bytes per 16-deep k step through the CU's 64 B/clk vector-memory return path
// for 3 * 64 * 128 / 1024 MFMAs of 32 clocks on 4 SIMDs -- return-path clocks / MFMA clocks = 128 (BM + BN) / (3 BM BN)
// = 1.0: the two are co-critical and the loop runs at 0.63 of the matrix rate.  The ratio only falls
// with a larger tile PER CU whose operand bytes are shared by all its waves: 128 x 128 -> 0.67.  Sharing means LDS, and
// staging 32 KB per slab through registers (ds_write_b128: 13 clocks each) is what made the LDS the busiest unit of
// gemm_split.hip.  The DMA path writes LDS without passing through registers or the DS store path:
//   * a wave-level global_load_lds_dwordx4 moves 64 lanes x 16 bytes = 1 KiB; the destination is wave-uniform base +
//     lane * 16 (linear), the SOURCE address is per lane;
//   * W is already laid out as 1 KiB MFMA fragments (tile-major): a piece is copied as it lies and a wave reads its
//     operand as base + lane * 16 -- conflict-free by construction;
//   * A is row-major with 128-byte slab rows: a piece is 8 rows x 8 chunks of 16 bytes, and lane p fetches chunk
//     (p % 8) ^ ((row / 2) % 8) of row p / 8, i.e. the XOR swizzle sits in the source address and the LDS image is
//     conflict-free for the fragment reads (lane c reads chunk q of row c at slot q ^ ((c / 2) % 8): the 16 lanes of a
//     ds_read_b128 group hit 16 distinct 16-byte bank groups);
//   * three (or two) slab buffers form a ring; a wave waits for ITS OWN pieces with a counted s_waitcnt vmcnt(N) --
//     later slabs stay in flight across the barrier -- and the ONE barrier per slab sits between the two 16-deep halves
//     of the slab's MFMAs, so the operands of the next half are read from LDS while the current half runs.
// Arithmetic, accumulation order and epilogue are those of gemm_split_wd.hip: results are bit-identical to it
// (tests/test_hip_gemm.py).
#include <algorithm>

#include "gemm_common.hpp"
#include "gemm_split_dma.hpp"

namespace css {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace dma {

constexpr int DBM = 128, DBN = 128;           // block tile
constexpr int A_BYTES = DBM * 128;            // one slab of A in LDS (128 rows x 128 bytes)
constexpr int W_BYTES = (DBN / 32) * 4096;    // one slab of W in LDS (4 column tiles x 4 fragments x 1 KiB)
constexpr int STAGE_BYTES = A_BYTES + W_BYTES;

// 16 bytes per lane, global -> LDS, no register in between.  dst: wave-uniform LDS address of the 1 KiB piece.
__device__ __forceinline__ void dma16(const void* src, void* dst) {
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// The same with the address as wave-uniform 64-bit base (SGPR pair) + 32-bit lane offset: the instruction then carries
// one address register per lane instead of two (the compiler keeps folding base + offset into per-lane 64-bit pointers,
// hence the assembly).  M0 = LDS destination of the piece; written and consumed inside the one statement.
__device__ __forceinline__ void dma16s(const void* base_uniform, unsigned lane_off, const void* dst) {
    const unsigned lds_addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)dst;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(base_uniform),
                 "s"(lds_addr)
                 : "memory", "m0");
}

// 16 bytes per lane into registers, invisible to the compiler's own vmcnt bookkeeping (beside hand-counted DMA pieces the
// compiler would wait for EVERYTHING before the first use of an ordinary load's result -- the pieces just issued
// included).  The caller waits (wait_vm) and passes the value through reg_fence() before anything reads or copies it.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v gload16_uncounted(const void* p) {
    f32x4v r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ void reg_fence(f32x4v& r) { asm volatile("" : "+v"(r)); }

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace dma

// WM x WN waves; wave tile (128 / WM) x (128 / WN); NST slab buffers.
template <int WM, int WN, int NST, int ABL = 0>   // ABL: timing ablations of tools/gemm_dma_bench.hip (0 in the product)
__global__ __launch_bounds__(WM * WN * 64, WM * WN >= 8 ? 2 : 1) void gemm_split_dma_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    using namespace dma;
    constexpr int WAVES = WM * WN;
    constexpr int TM = DBM / 32 / WM, TN = DBN / 32 / WN;    // 32 x 32 tiles per wave
    constexpr int PIECES = 32 / WAVES;                       // DMA pieces per wave and slab (16 of A + 16 of W per block)
    constexpr int PA = PIECES / 2, PW = PIECES / 2;
    static_assert(PIECES >= 2 && PIECES % 2 == 0, "wave count");
    constexpr int PATCH_BYTES = WAVES * 32 * LDS_LD * 4;
    constexpr int LDS_BYTES = NST * STAGE_BYTES > PATCH_BYTES ? NST * STAGE_BYTES : PATCH_BYTES;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];   // the ONLY shared object of the kernel

    const int n_tiles = tiles_m * tiles_n;
    const int tile = xcd_tile(blockIdx.x, n_tiles);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int m0 = tm * DBM, n0 = tn * DBN;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wn = wave % WN, wm = wave / WN;
    const int c = lane & 31, h = lane >> 5;
    const int M = g.M, N = g.N;
    const int nk = g.K / BK;

    // ---- DMA sources.  A: pieces wave * PA .. + PA - 1, piece q = rows 8q .. 8q + 7; this lane: row 8q + lane / 8,
    // chunk (lane % 8) ^ swizzle(row).  Rows past M re-read the last valid row (finite, never stored).
    // (addresses as wave-uniform base + 32-bit lane offset: the SGPR-base form of the instruction carries half the
    // address registers of the 64-bit-per-lane form through the issue path)
    unsigned offA[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (wave * PA + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int m = m0 + row < M ? m0 + row : M - 1;
        offA[i] = (unsigned)((int64_t)m * g.lda * 4 + chunk * 16);
    }
    // W: pieces wave * PW .. + PW - 1 of the slab's 16 (column tile jj = piece / 4, fragment = piece % 4); a column
    // tile's four fragments of one slab are 4 KiB contiguous in the tile-major layout
    const int jt_max = (N + 31) / 32 - 1;
    unsigned offW[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int piece = wave * PW + i;
        const int jt = min(n0 / 32 + (piece >> 2), jt_max);
        offW[i] = (unsigned)(((int64_t)jt * (g.K / 16) * 2 * 64 + (piece & 3) * 64 + lane) * 16);
    }
    const char* const baseA = reinterpret_cast<const char*>(g.A);
    const char* const baseW = reinterpret_cast<const char*>(g.B);
    auto issue = [&](int kt, int st) {   // this wave's pieces of slab kt -> ring slot st
        char* base = lds + st * STAGE_BYTES;
        const char* a = baseA + (int64_t)kt * 128;
        const char* w = baseW + (int64_t)kt * 4096;
#pragma unroll
        for (int i = 0; i < PA; ++i) dma16s(a, offA[i], base + (wave * PA + i) * 1024);
#pragma unroll
        for (int i = 0; i < PW; ++i) dma16s(w, offW[i], base + A_BYTES + (wave * PW + i) * 1024);
    };

    // ---- operand reads.  A fragment (row tile i of this wave, k half kk, part p): lane (c, h) reads chunk
    // q = 4 p + 2 kk + h of row 32 (wm TM + i) + c at slot q ^ ((c / 2) % 8).
    const int a_row_off = (wm * TM * 32 + c) * 128;
    const int a_swz = (c >> 1) & 7;
    auto lda_frag = [&](int st, int i, int kk, int p) -> f16x8 {
        const int q = (4 * p + 2 * kk + h) ^ a_swz;
        return __builtin_bit_cast(f16x8, *reinterpret_cast<const float4*>(lds + st * STAGE_BYTES + a_row_off + i * 32 * 128 + q * 16));
    };
    auto ldw_frag = [&](int st, int j, int kk, int p) -> f16x8 {
        return __builtin_bit_cast(f16x8, *reinterpret_cast<const float4*>(lds + st * STAGE_BYTES + A_BYTES +
                                                                            ((wn * TN + j) * 4 + kk * 2 + p) * 1024 + lane * 16));
    };

    // epilogue operands (column bias, residual) are requested before the first DMA and used after the K loop
    const int mrow = m0 + wm * TM * 32 + 4 * h, ncol = n0 + wn * TN * 32 + c;
    // (wave tiles of 2 x 2: the 68 registers of four tiles' operands do not fit beside 128 accumulators and the operand
    // sets -- they are requested after the K loop instead, all at once, when the operand sets are dead)
    constexpr bool PREF = TM * TN <= 2;
    TilePre pre[TM][TN];
    const bool early = !g.bias_along_m;
    if (PREF && early) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) tile_prefetch(pre[i][j], mrow + 32 * i, ncol + 32 * j, M, N, g.bias, g.residual, g.ldr);
    }

    f32x16 acc[TM][TN], cor[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; cor[i][j][r] = 0.f; }

    // ---- prologue: the first NST slabs are on their way; wait for slab 0
    const int pro = nk < NST ? nk : NST;
    for (int s = 0; s < pro; ++s) issue(s, s);
    if (pro >= 3) wait_vm<2 * PIECES>();
    else if (pro == 2) wait_vm<PIECES>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    f16x8 xa[2][TM][2], xw[2][TN][2];   // [operand set][tile][part]: set 0 = k half 0, set 1 = k half 1 of the slab
#pragma unroll
    for (int i = 0; i < TM; ++i) { xa[0][i][0] = lda_frag(0, i, 0, 0); xa[0][i][1] = lda_frag(0, i, 0, 1); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { xw[0][j][0] = ldw_frag(0, j, 0, 0); xw[0][j][1] = ldw_frag(0, j, 0, 1); }

    if constexpr (ABL & 4) {   // (ablation: no LDS reads in the loop -- both operand sets stay what the prologue read)
#pragma unroll
        for (int i = 0; i < TM; ++i) { xa[1][i][0] = xa[0][i][0]; xa[1][i][1] = xa[0][i][1]; }
#pragma unroll
        for (int j = 0; j < TN; ++j) { xw[1][j][0] = xw[0][j][0]; xw[1][j][1] = xw[0][j][1]; }
    }
    // the 3 TM TN MFMAs of one k half, in gemm_split_wd.hip's order per accumulator: acc += ah wh; cor += ah wl; cor += al wh
#define CSS_HALF(s)                                                                                                       \
    if constexpr (ABL & 1) {                                                                                              \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) { asm volatile("" ::"v"(xa[s][i][0]), "v"(xa[s][i][1])); }          \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) { asm volatile("" ::"v"(xw[s][j][0]), "v"(xw[s][j][1])); }          \
    } else CSS_HALF_MFMA(s)
#define CSS_HALF_MFMA(s) {                                                                                                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                         \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[s][i][0], xw[s][j][0], acc[i][j], 0, 0, 0);                 \
    if constexpr (!(ABL & 16))                                                                                            \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                         \
        cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[s][i][0], xw[s][j][1], cor[i][j], 0, 0, 0);                 \
    if constexpr (!(ABL & 16))                                                                                            \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                         \
        cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[s][i][1], xw[s][j][0], cor[i][j], 0, 0, 0); }

    // One slab: second half's operands from LDS while the first half's MFMAs run; the slab's ONE barrier between the
    // halves (slab kt is then in every wave's registers: its ring slot takes slab kt + NST; slab kt + 1 has landed: its
    // first half's operands are read while the second half's MFMAs run).  ISSUE / NEXT / VM are literal constants in the
    // steady-state loop, so that its body is branch-free and the compiler's own lgkmcnt bookkeeping stays exact (a
    // conditional LDS read forces a conservative lgkmcnt(0) in front of MFMAs that do not depend on it).
#define CSS_READ_SET(s, slot, kk)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) { xa[s][i][0] = lda_frag(slot, i, kk, 0); xa[s][i][1] = (ABL & 32) ? xa[s][i][0] : lda_frag(slot, i, kk, 1); } \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) { xw[s][j][0] = ldw_frag(slot, j, kk, 0); xw[s][j][1] = (ABL & 32) ? xw[s][j][0] : ldw_frag(slot, j, kk, 1); }
#define CSS_SLAB(ISSUE, NEXT, VMSTMT)                                                                                    \
    {                                                                                                                    \
        if constexpr (!(ABL & 4)) { CSS_READ_SET(1, st, 1) }                                                             \
        CSS_HALF(0)                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): this wave's reads of slab kt are complete */                  \
        if constexpr (!(ABL & 2)) { VMSTMT; }                                                                            \
        if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        const int st1 = st + 1 == NST ? 0 : st + 1;                                                                      \
        if constexpr (!(ABL & 2)) { if (ISSUE) issue(kt + NST, st); }                                                    \
        if constexpr (!(ABL & 4)) { if (NEXT) { CSS_READ_SET(0, st1, 0) } }                                              \
        CSS_HALF(1)                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        st = st1;                                                                                                        \
    }
    int st = 0;   // ring slot of slab kt
    int kt = 0;
    // steady state: NST - 1 later slabs are in flight or landed; waiting for slab kt + 1 leaves NST - 2 of them in flight
    for (; kt + NST < nk; ++kt) CSS_SLAB(1, 1, wait_vm<(NST - 2) * PIECES>())
    // the last (up to) NST slabs: nothing left to request, the ring drains
    for (; kt < nk; ++kt) {
        const int later = nk - kt - 2;   // slabs after kt + 1 still in flight
        if (later >= 1 && NST >= 3) CSS_SLAB(0, 1, wait_vm<PIECES>())
        else if (later == 0) CSS_SLAB(0, 1, wait_vm<0>())
        else CSS_SLAB(0, 0, (void)0)
    }
#undef CSS_SLAB
#undef CSS_READ_SET
#undef CSS_HALF
#undef CSS_HALF_MFMA

    // ---- epilogue (gemm_common.hpp): each wave's finished tiles pass through its own [32][LDS_LD] patch, which reuses
    // the slab ring -- every wave must be past its last operand read first
    __syncthreads();
    const float* bias = g.bias;
    const float* res = g.residual;
    const int act = g.act, bias_m = g.bias_along_m, so = g.split_out;
    const int64_t ldc = g.ldc, ldr = g.ldr;
    const float alpha = g.alpha;
    float* patch = reinterpret_cast<float*>(lds) + wave * (32 * LDS_LD);
    const int mtile = m0 + wm * TM * 32, ntile0 = n0 + wn * TN * 32;
    const bool wide = early && !g.narrow_epilogue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            acc[i][j] += cor[i][j] * SPLIT_LO_INV;
            if (g.range_flag) range_check(acc[i][j], g.range_flag);
        }
    if (!PREF && early) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) tile_prefetch(pre[i][j], mrow + 32 * i, ncol + 32 * j, M, N, g.bias, g.residual, g.ldr);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ntile = ntile0 + 32 * j;
            const bool fragq = g.frag_out != nullptr && ntile < 2 * g.frag_D;   // wave-uniform
            const f32x16 a = acc[i][j];
            if (fragq)
                emit_tile_frag(a, pre[i][j].bn, mtile + 32 * i, h, c, M, g.frag_out, g.frag_T, g.frag_invT, g.frag_heads,
                               (ntile % g.frag_D) >> 6, ntile / g.frag_D, (ntile >> 5) & 1, patch);
            else if (wide)
                emit_tile_pre_wide(a, pre[i][j], mtile + 32 * i, h, c, ntile, M, N, g.C, ldc, act, res != nullptr, alpha, so,
                                   g.nt_store, patch);
            else if (early)
                emit_tile_pre(a, pre[i][j], mrow + 32 * i, ncol + 32 * j, M, N, g.C, ldc, act, res != nullptr, alpha, so, g.nt_store);
            else
                emit_tile(a, mrow + 32 * i, ncol + 32 * j, M, N, g.C, ldc, bias, bias_m, act, res, ldr, alpha, so);
        }
}

// Loader-side epilogue of the specialised-wave kernels: the finished 128 x 128 tile lies in LDS (float32, row stride CLD);
// loader wave `ld` turns rows 32 ld .. 32 ld + 31 into the output -- bias, activation, residual (rv: this lane's 16
// residual pieces, requested earlier), float32 or split-f16 rows, 512 contiguous bytes per 32 lanes -- or into the
// attention's fragment order for the q / k columns (gemm_common.hpp emit_tile_frag).
template <int CLD>
__device__ __forceinline__ void ws_epilogue(const GemmArgs& g, const float* __restrict__ Cs, int ld, int lane, int m0, int n0,
                                            const float4* rv, bool has_res) {
    const int M = g.M, N = g.N;
    const int so = g.split_out, act = g.act;
    const float alpha = g.alpha;
    if (g.frag_out != nullptr && n0 < 2 * g.frag_D) {
        const int c = lane & 31, hp = lane >> 5;
        const int m = m0 + 32 * ld + c;
        if (m < M) {
            const int T = g.frag_T;
            const int seg = (int)(((float)m + 0.5f) * g.frag_invT);
            const int j = m - seg * T;
            const int njt = (T + 31) >> 5;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const int ntile = n0 + 32 * jt;
                const int head = (ntile % g.frag_D) >> 6, which = ntile / g.frag_D, grp = (ntile >> 5) & 1;
                float4* dst = reinterpret_cast<float4*>(g.frag_out) +
                              ((((int64_t)(seg * g.frag_heads + head) * njt + (j >> 5)) * 2 + which) * 8 + grp * 4) * 64 + (j & 31);
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int col = 32 * jt + 16 * sub + 8 * hp;
                    const float* src = Cs + (32 * ld + c) * CLD + col;
                    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
                    const float* bp = g.bias + n0 + col;   // (bias vectors sit in the weight blob: 4-byte aligned only)
                    const float v[8] = {a.x + bp[0], a.y + bp[1], a.z + bp[2], a.w + bp[3], b.x + bp[4], b.y + bp[5], b.z + bp[6], b.w + bp[7]};
                    f16x8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        _Float16 x, y;
                        split_f16(v[e], x, y);
                        hi[e] = x; lo[e] = y;
                    }
                    dst[(sub * 2 + 0) * 64 + 32 * hp] = __builtin_bit_cast(float4, hi);
                    dst[(sub * 2 + 1) * 64 + 32 * hp] = __builtin_bit_cast(float4, lo);
                }
            }
        }
        return;
    }
    const int col = 4 * (lane & 31), n = n0 + col;
    if (n >= N) return;
    const float4 bn = g.bias ? make_float4(g.bias[n], g.bias[n + 1], g.bias[n + 2], g.bias[n + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int row = 32 * ld + 2 * it + (lane >> 5), m = m0 + row;
        float4 v = *reinterpret_cast<const float4*>(Cs + row * CLD + col);
        v.x += bn.x; v.y += bn.y; v.z += bn.z; v.w += bn.w;
        if (act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        else if (act == ACT_SIGMOID) { v.x = sigmoidf_(v.x); v.y = sigmoidf_(v.y); v.z = sigmoidf_(v.z); v.w = sigmoidf_(v.w); }
        if (has_res) { v.x = rv[it].x + alpha * v.x; v.y = rv[it].y + alpha * v.y; v.z = rv[it].z + alpha * v.z; v.w = rv[it].w + alpha * v.w; }
        if (m >= M) continue;
        float* dstrow = g.C + (int64_t)m * g.ldc;
        if (n < so) split_store4(reinterpret_cast<_Float16*>(dstrow), n, v.x, v.y, v.z, v.w);
        else *reinterpret_cast<float4*>(dstrow + n) = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same tile with SPECIALISED waves.  Ablations of the kernel above (tools/gemm_dma_bench.hip, M = 7440, N = 512,
// K = 1024) put its slab at MFMAs 0.42 us + DMA issue 0.13 + operand reads 0.08 = 0.63 us: the three do not overlap,
// they ADD.  A wave that issues a global_load_lds is held ~60-70 clocks per instruction (address path), a ds_read_b128
// ~14, and the two waves of a SIMD run the same program between the same barriers, so both are held at the same time
// and the matrix pipe idles.  Here the roles are split:
//   waves 0-3  (one per SIMD) own 64 x 64 of the tile each (2 x 2 MFMA tiles: a third fewer operand bytes from LDS per
//              MFMA than 64 x 32) and execute ONLY ds_read_b128 + MFMA + one barrier per slab, the reads of the next
//              k half issued one per MFMA of the current half;
//   waves 4-7  issue the slab ring's DMA pieces (8 per wave and slab), wait for them with counted vmcnt and meet the
//              consumers at the same barrier; they never touch the matrix pipe, and their slow VMEM issue runs beside
//              the consumers' MFMAs (different issue ports of the SIMD).
// Protocol per slab kt (one s_barrier for all eight waves): consumers arrive with slab kt entirely in registers
// (lgkmcnt(0)), loaders with their pieces of slab kt + 1 landed (vmcnt); past it the loaders refill slab kt's slot with
// slab kt + NST and the consumers read slab kt + 1.
template <int NST, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_split_ws_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    using namespace dma;
    constexpr int TM = 2, TN = 2, NLOAD = 4, PL = 32 / NLOAD, PA = PL / 2, PW = PL / 2;
    constexpr int CLD = 132;                       // row stride of the finished tile in LDS (floats)
    constexpr int TILE_BYTES = 128 * CLD * 4;
    constexpr int LDS_BYTES = NST * STAGE_BYTES > TILE_BYTES ? NST * STAGE_BYTES : TILE_BYTES;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];   // the ONLY shared object of the kernel

    const int n_tiles = tiles_m * tiles_n;
    const int tile = xcd_tile(blockIdx.x, n_tiles);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int m0 = tm * DBM, n0 = tn * DBN;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int M = g.M, N = g.N;
    const int nk = g.K / BK;

    if (wave >= 4) {
        // ================================================================================ loader waves
        const int ld = wave - 4;
        unsigned offA[PA];
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = (ld * PA + i) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            const int m = m0 + row < M ? m0 + row : M - 1;
            offA[i] = (unsigned)((int64_t)m * g.lda * 4 + chunk * 16);
        }
        const int jt_max = (N + 31) / 32 - 1;
        unsigned offW[PW];
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int piece = ld * PW + i;
            const int jt = min(n0 / 32 + (piece >> 2), jt_max);
            offW[i] = (unsigned)(((int64_t)jt * (g.K / 16) * 2 * 64 + (piece & 3) * 64 + lane) * 16);
        }
        const char* const baseA = reinterpret_cast<const char*>(g.A);
        const char* const baseW = reinterpret_cast<const char*>(g.B);
        auto issue = [&](int kt, int st) {
            char* base = lds + st * STAGE_BYTES;
            const char* a = baseA + (int64_t)kt * 128;
            const char* w = baseW + (int64_t)kt * 4096;
#pragma unroll
            for (int i = 0; i < PA; ++i) dma16s(a, offA[i], base + (ld * PA + i) * 1024);
#pragma unroll
            for (int i = 0; i < PW; ++i) dma16s(w, offW[i], base + A_BYTES + (ld * PW + i) * 1024);
        };
        const int pro = nk < NST ? nk : NST;
        for (int s = 0; s < pro; ++s) issue(s, s);
        if (pro >= 3) wait_vm<2 * PL>();
        else if (pro == 2) wait_vm<PL>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        // the epilogue's residual operand (this wave's 32 rows of the tile, 16 x 16 bytes per lane) is requested now, behind
        // the first slabs: it lands under the K loop.  The loads sit between slab NST - 1 and slab NST in the (in-order)
        // return queue: the waits for slabs 1 .. NST - 1 leave them outstanding, every later wait covers them.
        const int ecol = 4 * (lane & 31), en = n0 + ecol;
        const bool has_res = g.residual != nullptr && !(g.frag_out != nullptr && n0 < 2 * g.frag_D) && en < N;
        float4 rv[16];
        if (has_res) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = 32 * ld + 2 * it + (lane >> 5);
                const int mc = m0 + row < M ? m0 + row : M - 1;
                rv[it] = *reinterpret_cast<const float4*>(g.residual + (int64_t)mc * g.ldr + en);
            }
        }
        int st = 0, kt = 0;
        for (; kt + NST < nk; ++kt) {
            if constexpr (!(ABL & 2)) {
                if (kt + 1 < NST && has_res) wait_vm<(NST - 2) * PL + 16>();
                else wait_vm<(NST - 2) * PL>();
            }
            __builtin_amdgcn_s_barrier();
            if constexpr (!(ABL & 2)) issue(kt + NST, st);
            st = st + 1 == NST ? 0 : st + 1;
        }
        for (; kt < nk; ++kt) {
            const int later = nk - kt - 2;
            if (later >= 1 && NST >= 3) wait_vm<PL>();
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
        }
        // ---- epilogue, loader side: the consumers leave the finished 128 x 128 tile in LDS (float32, row stride CLD; it
        // reuses the slab ring, hence barrier A = every consumer is past its last operand read); this wave turns rows
        // 32 ld .. 32 ld + 31 into the output: bias, activation, residual, float32 or split-f16 rows, 512 contiguous bytes
        // per 32 lanes -- or the attention's fragment order for the q / k columns (gemm_common.hpp emit_tile_frag).
        __builtin_amdgcn_s_barrier();   // A
        __builtin_amdgcn_s_barrier();   // B: the tile is in LDS
        ws_epilogue<CLD>(g, reinterpret_cast<const float*>(lds), ld, lane, m0, n0, rv, has_res);
        return;
    }

    // ==================================================================================== consumer waves
    const int wn = wave & 1, wm = wave >> 1;
    const int c = lane & 31, h = lane >> 5;
    const int a_row_off = (wm * TM * 32 + c) * 128;
    const int a_swz = (c >> 1) & 7;
    auto lda_frag = [&](int st, int i, int kk, int p) -> f16x8 {
        const int q = (4 * p + 2 * kk + h) ^ a_swz;
        return __builtin_bit_cast(f16x8, *reinterpret_cast<const float4*>(lds + st * STAGE_BYTES + a_row_off + i * 32 * 128 + q * 16));
    };
    auto ldw_frag = [&](int st, int j, int kk, int p) -> f16x8 {
        return __builtin_bit_cast(f16x8, *reinterpret_cast<const float4*>(lds + st * STAGE_BYTES + A_BYTES +
                                                                            ((wn * TN + j) * 4 + kk * 2 + p) * 1024 + lane * 16));
    };
    f32x16 acc[TM][TN], cor[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; cor[i][j][r] = 0.f; }

    __builtin_amdgcn_s_barrier();   // slab 0 has landed
    __builtin_amdgcn_sched_barrier(0);
    f16x8 xa[2][TM][2], xw[2][TN][2];
#define CSS_READ_SET(s, slot, kk)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) { xa[s][i][0] = lda_frag(slot, i, kk, 0); xa[s][i][1] = lda_frag(slot, i, kk, 1); } \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) { xw[s][j][0] = ldw_frag(slot, j, kk, 0); xw[s][j][1] = ldw_frag(slot, j, kk, 1); }
#define CSS_HALF(s)                                                                                                      \
    if constexpr (ABL & 1) {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) { asm volatile("" ::"v"(xa[s][i][0]), "v"(xa[s][i][1])); }         \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) { asm volatile("" ::"v"(xw[s][j][0]), "v"(xw[s][j][1])); }         \
    } else {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                    \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[s][i][0], xw[s][j][0], acc[i][j], 0, 0, 0);            \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                    \
            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[s][i][0], xw[s][j][1], cor[i][j], 0, 0, 0);            \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                    \
            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[s][i][1], xw[s][j][0], cor[i][j], 0, 0, 0);            \
    }
    // issue order of a half: the 8 operand reads of the NEXT half one per MFMA, then the remaining MFMAs
#define CSS_PACE()                                                      \
    if constexpr (!(ABL & 1) && !(ABL & 4)) {                            \
        _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {               \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          \
        }                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);              \
    }
    CSS_READ_SET(0, 0, 0)
    if constexpr (ABL & 4) { CSS_READ_SET(1, 0, 1) }
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if constexpr (!(ABL & 4)) { CSS_READ_SET(1, st, 1) }
        CSS_HALF(0)
        CSS_PACE()
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): slab kt is in this wave's registers
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        st = st + 1 == NST ? 0 : st + 1;
        // (past the last slab this reads a slot nobody refills: harmless, and the loop body stays branch-free)
        if constexpr (!(ABL & 4)) { CSS_READ_SET(0, st, 0) }
        CSS_HALF(1)
        CSS_PACE()
        __builtin_amdgcn_sched_barrier(0);
    }
#undef CSS_PACE
#undef CSS_HALF
#undef CSS_READ_SET

    // ---- epilogue, consumer side: correction term, range check, the 64 x 64 share of the tile to LDS; the loader waves
    // take it from there (above).  No vector-memory instruction is executed by a consumer wave in the whole kernel.
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();   // A: nobody reads the slab ring any more
    float* Cs = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            acc[i][j] += cor[i][j] * SPLIT_LO_INV;
            if (g.range_flag) range_check(acc[i][j], g.range_flag);
            float* t = Cs + (wm * 64 + 32 * i + 4 * h) * CLD + wn * 64 + 32 * j + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[((r & 3) + 8 * (r >> 2)) * CLD] = acc[i][j][r];
        }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();   // B
}

// One CHUNK of the loader-side epilogue, for the persistent kernel: the tile in LDS already carries the column bias (the
// consumers add it when they leave the tile there), so a chunk is LDS reads + arithmetic + a FIXED number of stores -- what
// lets the loader count its vector-memory queue exactly while chunks and DMA pieces alternate.
//   rows of the output (16 chunks): chunk q = rows 32 ld + 2 q + {0, 1}, 128 columns: one 16-byte store per lane
//                                   (float32 columns) or two 8-byte stores (split-f16 columns)
//   q / k columns in the attention's fragment order (8 chunks): chunk q = column tile q / 2, half q % 2 of token row
//                                   32 ld + lane % 32: two 16-byte stores per lane
template <int CLD>
__device__ __forceinline__ void ws_epilogue_chunk(const GemmArgs& g, const float* __restrict__ Cs, int ld, int lane, int m0, int n0,
                                                  float4 r, bool has_res, bool frag, int q) {
    const int M = g.M, N = g.N;
    if (frag) {
        const int c = lane & 31, hp = lane >> 5;
        const int m = m0 + 32 * ld + c;
        if (m >= M) return;
        const int T = g.frag_T;
        const int seg = (int)(((float)m + 0.5f) * g.frag_invT);
        const int j = m - seg * T;
        const int njt = (T + 31) >> 5;
        const int jt = q >> 1, sub = q & 1;
        const int ntile = n0 + 32 * jt;
        const int head = (ntile % g.frag_D) >> 6, which = ntile / g.frag_D, grp = (ntile >> 5) & 1;
        float4* dst = reinterpret_cast<float4*>(g.frag_out) +
                      ((((int64_t)(seg * g.frag_heads + head) * njt + (j >> 5)) * 2 + which) * 8 + grp * 4) * 64 + (j & 31);
        const float* src = Cs + (32 * ld + c) * CLD + 32 * jt + 16 * sub + 8 * hp;
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 x, y;
            split_f16(v[e], x, y);
            hi[e] = x; lo[e] = y;
        }
        dst[(sub * 2 + 0) * 64 + 32 * hp] = __builtin_bit_cast(float4, hi);
        dst[(sub * 2 + 1) * 64 + 32 * hp] = __builtin_bit_cast(float4, lo);
        return;
    }
    const int col = 4 * (lane & 31), n = n0 + col;
    const int row = 32 * ld + 2 * q + (lane >> 5), m = m0 + row;
    if (n >= N || m >= M) return;
    const int act = g.act;
    const float alpha = g.alpha;
    float4 v = *reinterpret_cast<const float4*>(Cs + row * CLD + col);
    if (act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    else if (act == ACT_SIGMOID) { v.x = sigmoidf_(v.x); v.y = sigmoidf_(v.y); v.z = sigmoidf_(v.z); v.w = sigmoidf_(v.w); }
    if (has_res) { v.x = r.x + alpha * v.x; v.y = r.y + alpha * v.y; v.z = r.z + alpha * v.z; v.w = r.w + alpha * v.w; }
    float* dstrow = g.C + (int64_t)m * g.ldc;
    if (n < g.split_out) split_store4(reinterpret_cast<_Float16*>(dstrow), n, v.x, v.y, v.z, v.w);
    else *reinterpret_cast<float4*>(dstrow + n) = v;
}

// ------------------------------------------------------------------------------------------------------------------
// The specialised-wave kernel PERSISTENT over a launch's tiles: one block per CU walks tiles b, b + grid, b + 2 grid, ...
// and the slabs of all of them form ONE stream through the slab ring (slab g = tile g / nk, k slab g % nk): the loaders
// keep requesting NST slabs ahead across tile boundaries, the consumers run from the last slab of a tile straight into
// the first of the next.  At a tile's end the consumers leave the finished tile in a SEPARATE 64 KB of LDS (so the ring
// keeps streaming: 3 x 32 KB + 64 KB = the CU's 160 KB) and go on; the next per-slab barrier makes it visible, and the
// loaders write it out there -- bias, activation, residual (requested half a tile earlier), 512-byte row pieces --
// while the consumers are already two slabs into the next tile.  What a launch pays once instead of once per round of
// tiles: the launch itself, the first slab's round trip and the tail's write-back; a tile's own epilogue costs the
// consumers the 0.5 us of the LDS dump plus what the ring's two slabs of lead do not cover of the loaders' ~1.5 us.
// The write-out is cut into 16 chunks, one per slab of the NEXT tile, so that the DMA stream never pauses; each chunk is a
// fixed number of stores (+ one residual load for the chunk after it), which lets the loaders keep counting their
// in-order vector-memory queue exactly (vmcnt) while chunks and DMA pieces alternate.  Same arithmetic, same order, same
// bits as gemm_split_ws_kernel.  K >= 512 (16 slabs per tile) is required: one chunk per slab must finish a tile.
template <int NST>
__global__ __launch_bounds__(512, 2) void gemm_split_wsp_kernel(GemmArgs g, int tiles_m, int tiles_n) {
    using namespace dma;
    static_assert(NST == 3, "the loaders' queue accounting assumes one slab group between a wait and its target");
    constexpr int TM = 2, TN = 2, NLOAD = 4, PL = 32 / NLOAD, PA = PL / 2, PW = PL / 2;
    constexpr int CLD = 128;                       // (no padding: the 16 lanes of a row-piece read and the 32 lanes of a column write hit distinct banks)
    constexpr int RING_BYTES = NST * STAGE_BYTES, TILE_BYTES = 128 * CLD * 4;
    __shared__ __attribute__((aligned(1024))) char lds[RING_BYTES + TILE_BYTES];   // the ONLY shared object of the kernel
    float* const Cs = reinterpret_cast<float*>(lds + RING_BYTES);

    const int n_tiles = tiles_m * tiles_n;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int M = g.M, N = g.N;
    const int nk = g.K / BK;
    // this block's tiles: L = blockIdx.x + t * gridDim.x (all on the block's XCD: the grid is a multiple of 8)
    const int my_tiles = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int G = my_tiles * nk;                   // slabs of the block's stream
    auto tile_origin = [&](int t, int& m0, int& n0) {
        const int tile = xcd_tile((int)blockIdx.x + t * (int)gridDim.x, n_tiles);
        m0 = (tile / tiles_n) * DBM;
        n0 = (tile % tiles_n) * DBN;
    };

    if (wave >= 4) {
        // ================================================================================ loader waves
        const int ld = wave - 4;
        const char* const baseA = reinterpret_cast<const char*>(g.A);
        const char* const baseW = reinterpret_cast<const char*>(g.B);
        const int jt_max = (N + 31) / 32 - 1;
        unsigned offA[PA], offW[PW];
        auto set_tile = [&](int m0, int n0) {      // DMA sources of a tile (see gemm_split_dma_kernel)
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int row = (ld * PA + i) * 8 + (lane >> 3);
                const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                const int m = m0 + row < M ? m0 + row : M - 1;
                offA[i] = (unsigned)((int64_t)m * g.lda * 4 + chunk * 16);
            }
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                const int piece = ld * PW + i;
                const int jt = min(n0 / 32 + (piece >> 2), jt_max);
                offW[i] = (unsigned)(((int64_t)jt * (g.K / 16) * 2 * 64 + (piece & 3) * 64 + lane) * 16);
            }
        };
        // the issue side of the stream: slab (it, ik) goes to ring slot is
        int it = 0, ik = 0, is = 0;
        auto issue_next = [&]() {
            if (ik == 0) { int m0, n0; tile_origin(it, m0, n0); set_tile(m0, n0); }
            char* base = lds + is * STAGE_BYTES;
            const char* a = baseA + (int64_t)ik * 128;
            const char* w = baseW + (int64_t)ik * 4096;
#pragma unroll
            for (int i = 0; i < PA; ++i) dma16s(a, offA[i], base + (ld * PA + i) * 1024);
#pragma unroll
            for (int i = 0; i < PW; ++i) dma16s(w, offW[i], base + A_BYTES + (ld * PW + i) * 1024);
            if (++ik == nk) { ik = 0; ++it; }
            is = is + 1 == NST ? 0 : is + 1;
        };
        const int pro = G < NST ? G : NST;
        for (int s = 0; s < pro; ++s) issue_next();
        if (pro >= 3) wait_vm<2 * PL>();
        else if (pro == 2) wait_vm<PL>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        // the consumption side: slab gk of tile gt; (em0, en0): the tile whose finished values lie in Cs and are being written
        // out, one chunk per slab: 16 chunks of row pairs (8 for fragment-ordered q / k columns), so a tile's epilogue
        // is over well before the consumers need Cs again (nk >= 16 slabs per tile).  A chunk's residual piece is
        // requested one slab ahead (the first one during the tile's own last slab).
        int gt = 0, gk = 0, cm0 = 0, cn0 = 0, em0 = 0, en0 = 0;
        tile_origin(0, cm0, cn0);
        f32x4v r_cur = {0.f, 0.f, 0.f, 0.f}, r_next = r_cur;
        bool ep_res = false, ep_frag = false, nx_res = false;
        int ep_chunk = -1, ep_chunks = 0;          // next chunk of the tile in Cs (-1: none in progress)
        int e_prev = 0;                            // vector-memory operations of the epilogue issued since the last DMA group
        auto res_piece = [&](int m0_, int n0_, int q) -> f32x4v {
            const int row = 32 * ld + 2 * q + (lane >> 5);
            const int mc = m0_ + row < M ? m0_ + row : M - 1;
            const int en = n0_ + 4 * (lane & 31) < N ? n0_ + 4 * (lane & 31) : N - 4;
            return gload16_uncounted(g.residual + (int64_t)mc * g.ldr + en);
        };
        bool res_pending = false;                  // an uncounted residual load was issued in the previous slab
        for (int gslab = 0; gslab < G; ++gslab) {
            // counted wait for this wave's pieces of slab gslab + 1.  The return queue is in order; YOUNGER than those pieces
            // are the epilogue operations of the previous slab (e_prev: a chunk's stores and the next chunk's residual
            // load -- counted exactly) and the pieces of slab gslab + 2.
            const int later = G - gslab - 2;
            if (later >= 1) {
                switch (e_prev) {
                    case 0: wait_vm<PL>(); break;
                    case 1: wait_vm<PL + 1>(); break;
                    case 2: wait_vm<PL + 2>(); break;
                    case 3: wait_vm<PL + 3>(); break;
                    default: wait_vm<PL>(); break;   // (never more; a smaller count only waits longer)
                }
            } else if (later == 0) {
                wait_vm<0>();
            }
            e_prev = 0;
            __builtin_amdgcn_s_barrier();
            if (res_pending) {
                // the residual piece requested one slab ago: only the DMA pieces issued behind it may still be in flight
                if (gslab + NST - 1 < G) wait_vm<PL>();
                else wait_vm<0>();
                reg_fence(r_cur);
                reg_fence(r_next);
                res_pending = false;
            }
            if (gk == 0 && gt > 0) {   // the barrier just passed was the first of tile gt: tile gt - 1 is complete in Cs
                ep_chunk = 0;
                ep_frag = g.frag_out != nullptr && en0 < 2 * g.frag_D;
                ep_chunks = ep_frag ? 8 : 16;
                ep_res = nx_res;
                r_cur = r_next;
            }
            if (ep_chunk >= 0) {
                ws_epilogue_chunk<CLD>(g, Cs, ld, lane, em0, en0, make_float4(r_cur[0], r_cur[1], r_cur[2], r_cur[3]), ep_res, ep_frag, ep_chunk);
                e_prev += (ep_frag || en0 < g.split_out) ? 2 : 1;
                if (++ep_chunk == ep_chunks) {
                    ep_chunk = -1;
                } else if (ep_res) {
                    r_cur = res_piece(em0, en0, ep_chunk);   // (lands during this slab; used after the next barrier)
                    e_prev += 1;
                    res_pending = true;
                }
            }
            if (gk == nk - 1) {   // the tile's last slab: its first residual piece for the epilogue that starts one barrier on
                nx_res = g.residual != nullptr && !(g.frag_out != nullptr && cn0 < 2 * g.frag_D);
                if (nx_res) { r_next = res_piece(cm0, cn0, 0); e_prev += 1; res_pending = true; }
            }
            if (gslab + NST < G) issue_next();
            if (++gk == nk) {
                gk = 0; ++gt;
                em0 = cm0; en0 = cn0;
                if (gt < my_tiles) tile_origin(gt, cm0, cn0);
            }
        }
        __builtin_amdgcn_s_barrier();   // the last tile is in Cs: this wave's 32 rows in one go
        {
            const bool frag = g.frag_out != nullptr && en0 < 2 * g.frag_D;
            float4 rv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {   // (ordinary loads: nothing uncounted follows them)
                const int row = 32 * ld + 2 * q + (lane >> 5);
                const int mc = em0 + row < M ? em0 + row : M - 1;
                const int en = en0 + 4 * (lane & 31) < N ? en0 + 4 * (lane & 31) : N - 4;
                rv[q] = (nx_res && !frag) ? *reinterpret_cast<const float4*>(g.residual + (int64_t)mc * g.ldr + en)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (!frag || q < 8) ws_epilogue_chunk<CLD>(g, Cs, ld, lane, em0, en0, rv[q], nx_res, frag, q);
        }
        return;
    }

    // ==================================================================================== consumer waves
    const int wn = wave & 1, wm = wave >> 1;
    const int c = lane & 31, h = lane >> 5;
    const int a_row_off = (wm * TM * 32 + c) * 128;
    const int a_swz = (c >> 1) & 7;
    auto lda_frag = [&](int st, int i, int kk, int p) -> f16x8 {
        const int q = (4 * p + 2 * kk + h) ^ a_swz;
        return __builtin_bit_cast(f16x8, *reinterpret_cast<const float4*>(lds + st * STAGE_BYTES + a_row_off + i * 32 * 128 + q * 16));
    };
    auto ldw_frag = [&](int st, int j, int kk, int p) -> f16x8 {
        return __builtin_bit_cast(f16x8, *reinterpret_cast<const float4*>(lds + st * STAGE_BYTES + A_BYTES +
                                                                            ((wn * TN + j) * 4 + kk * 2 + p) * 1024 + lane * 16));
    };
    f32x16 acc[TM][TN], cor[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; cor[i][j][r] = 0.f; }

    __builtin_amdgcn_s_barrier();   // slab 0 has landed
    __builtin_amdgcn_sched_barrier(0);
    f16x8 xa[2][TM][2], xw[2][TN][2];
#define CSS_READ_SET(s, slot, kk)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) { xa[s][i][0] = lda_frag(slot, i, kk, 0); xa[s][i][1] = lda_frag(slot, i, kk, 1); } \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) { xw[s][j][0] = ldw_frag(slot, j, kk, 0); xw[s][j][1] = ldw_frag(slot, j, kk, 1); }
#define CSS_HALF(s)                                                                                                      \
    {                                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                    \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[s][i][0], xw[s][j][0], acc[i][j], 0, 0, 0);            \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                    \
            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[s][i][0], xw[s][j][1], cor[i][j], 0, 0, 0);            \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                    \
            cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[s][i][1], xw[s][j][0], cor[i][j], 0, 0, 0);            \
    }
#define CSS_PACE()                                                      \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                   \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);              \
    }                                                                   \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    CSS_READ_SET(0, 0, 0)
    int st = 0;
    for (int t = 0; t < my_tiles; ++t) {
        for (int kt = 0; kt < nk; ++kt) {
            CSS_READ_SET(1, st, 1)
            CSS_HALF(0)
            CSS_PACE()
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the slab is in registers (and, after a tile's end, its dump in LDS)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            st = st + 1 == NST ? 0 : st + 1;
            // (the next slab may be the next tile's first -- same ring; past the stream's end: a slot nobody refills)
            CSS_READ_SET(0, st, 0)
            CSS_HALF(1)
            CSS_PACE()
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the tile is finished: correction term, range check, column bias, its values to Cs (free: the loaders wrote the
        // previous tile out during this tile's first 16 slabs), accumulators back to zero
        float bcol[TN];
        {
            int m0_, n0_;
            tile_origin(t, m0_, n0_);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0_ + wn * 64 + 32 * j + c;
                bcol[j] = g.bias ? g.bias[n < N ? n : N - 1] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] += cor[i][j] * SPLIT_LO_INV;
                if (g.range_flag) range_check(acc[i][j], g.range_flag);
                float* dst = Cs + (wm * 64 + 32 * i + 4 * h) * CLD + wn * 64 + 32 * j + c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    dst[((r & 3) + 8 * (r >> 2)) * CLD] = acc[i][j][r] + bcol[j];
                    acc[i][j][r] = 0.f; cor[i][j][r] = 0.f;
                }
            }
    }
#undef CSS_PACE
#undef CSS_HALF
#undef CSS_READ_SET
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();   // the last tile is in Cs; the loaders take it from here
}

// Whether the specialised-wave kernel can take this launch: its epilogue writes whole 16-byte row pieces.
bool gemm_split_ws_eligible(const GemmArgs& g) {
    return g.split_in && g.b_tiled && g.batch == 1 && !g.bias_along_m && !g.narrow_epilogue && !g.nt_store && g.N % 32 == 0 &&
           g.K % 32 == 0 && g.ldc % 4 == 0 && ((uintptr_t)g.C & 15) == 0 &&
           (!g.residual || (g.ldr % 4 == 0 && ((uintptr_t)g.residual & 15) == 0)) && (!g.frag_out || (g.bias && g.frag_D % 128 == 0)) &&
           (g.split_out == 0 || g.split_out % 128 == 0 || g.split_out >= g.N);
}

// ... and the persistent form (tile_rows = 33): 16 slabs per tile, one epilogue chunk per slab
bool gemm_split_wsp_eligible(const GemmArgs& g) { return gemm_split_ws_eligible(g) && g.K >= 512; }

// g.tile_rows selects the variant (tools / tests): 3 = specialised waves (gemm_split_ws_kernel, three slab buffers);
// 24 = gemm_split_dma_kernel (2 x 4 waves, 3 slab buffers; the 4 x 2 and two-buffer forms measured the same and are gone)
void launch_gemm_split_dma(const GemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return;
    const int tiles_m = (g.M + dma::DBM - 1) / dma::DBM, tiles_n = (g.N + dma::DBN - 1) / dma::DBN;
    const dim3 grid(tiles_m * tiles_n);
    switch (g.tile_rows) {
        case 24: hipLaunchKernelGGL((gemm_split_dma_kernel<2, 4, 3>), grid, dim3(512), 0, s, g, tiles_m, tiles_n); break;
        case 4: hipLaunchKernelGGL((gemm_split_ws_kernel<4>), grid, dim3(512), 0, s, g, tiles_m, tiles_n); break;
        case 33: {   // persistent over the launch's tiles: one block per CU (160 KB of LDS each), a multiple of 8 blocks
            static int cus = 0;
            if (!cus) {
                int dev = 0;
                hipDeviceProp_t pr{};
                if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
                if (cus < 8) cus = 256;
                cus = cus / 8 * 8;
            }
            const int n_tiles = tiles_m * tiles_n;
            const int blocks = n_tiles < cus ? (n_tiles + 7) / 8 * 8 : cus;
            hipLaunchKernelGGL((gemm_split_wsp_kernel<3>), dim3(std::min(blocks, std::max(n_tiles, 1))), dim3(512), 0, s, g, tiles_m, tiles_n);
            break;
        }
#ifdef CSS_GEMM_DMA_ABLATE
#define CSS_ABL(n) case 1000 + n: hipLaunchKernelGGL((gemm_split_dma_kernel<2, 4, 3, n>), grid, dim3(512), 0, s, g, tiles_m, tiles_n); break;
        CSS_ABL(1) CSS_ABL(2) CSS_ABL(4) CSS_ABL(3) CSS_ABL(5) CSS_ABL(6) CSS_ABL(7) CSS_ABL(16) CSS_ABL(18) CSS_ABL(32) CSS_ABL(34) CSS_ABL(22)
#undef CSS_ABL
#define CSS_ABL(n) case 2000 + n: hipLaunchKernelGGL((gemm_split_ws_kernel<3, n>), grid, dim3(512), 0, s, g, tiles_m, tiles_n); break;
        CSS_ABL(1) CSS_ABL(2) CSS_ABL(4) CSS_ABL(6)
#undef CSS_ABL
#endif
        default: hipLaunchKernelGGL((gemm_split_ws_kernel<3>), grid, dim3(512), 0, s, g, tiles_m, tiles_n); break;
    }
}

}  // namespace css

// Exact float32 MFMA GEMM for gfx950, second form (round 5):  C[m][n] = epilogue( sum_k A[m][k] * B[n][k] ).
//
// Same arithmetic as gemm.hip (v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain; the same K permutation inside each
// group of 8 k), so every output is bit for bit what gemm_kernel writes -- what changes is how the matrix pipe is fed:
//
//   * FOUR independent 4-wave blocks per CU instead of one 8-wave block.  The round-4 kernel needs 169 registers, i.e.
//     two waves per SIMD, both from ONE block: they run the same 32 MFMAs between the same barriers, and the pipe idles
//     while both store the next slab, meet and wait for their first operand reads (a slab takes 2 680 cycles for 2 048 of
//     MFMA issue with one wave per SIMD, tools/gemm_f32_bench.hip).  Here a wave owns 32 COLUMNS of the block's tile and all
//     of its rows (TM row tiles of 32: at most 64 accumulator registers), the slab is 16 deep and double buffered
//     (2 x (128 + 128) rows x 20 floats = 40 960 B: exactly a quarter of the CU's 160 KB), and the kernel is held to
//     128 registers (__launch_bounds__(256, 4)) -- four waves per SIMD from four blocks that fill each other's bubbles.
//     The cap is met by SPILLING: hipcc -Rpass-analysis=kernel-resource-usage reports scratch for both instantiations
//     (the epilogue's descriptors, offsets and the switch over four tile heights live across the K loop); the reloads sit
//     outside the slab loop except for a handful in the TM = 3 / 4 tiles (DESIGN.md 3.1c records the counts per round).
//   * PERSISTENT blocks over a balanced cut of the work.  The launch is 1 024 blocks (4 x 256 CUs); the output is
//     counted in UNITS of 32 rows x 128 columns, the units of a launch are a line (batch entry, column panel, row unit --
//     row unit fastest), every "virtual CU" takes an equal contiguous piece of that line and each of its four blocks a
//     quarter of the piece, walked in tiles of 1..4 units.  A block's time is proportional to its units, so every CU is
//     busy for ceil(units / 256) unit times whatever M is: M = 22 320 x N = 512 is 2 792 units = 10.9 per CU, where 700
//     tiles of 128 rows leave a quarter of the CUs with 12 units and the rest with 8.
//   * The four blocks of a CU are kept OUT OF PHASE: their first tiles have different heights, so their tile ends --
//     the epilogue's residual reads and result writes, the next tile's cold first slab -- fall into the other blocks'
//     K loops instead of all CUs leaving the matrix pipe idle together (one round of equal tiles: 25 us of a 116 us
//     launch was prologue + epilogue, fitted over K).
//   * Operands arrive by buffer loads (rows past M / N read zeros: no clamped row pointers, one offset register per
//     operand); the epilogue requests a tile's residual values one 32 x 32 tile ahead (the first under the last slab's
//     MFMAs) and writes 16-byte row pieces after a turn through a wave-private LDS patch.
#include <algorithm>

#include "gemm_common.hpp"

namespace css {

namespace {

constexpr int F_BK = 16;                     // slab depth
constexpr int F_LD = 20;                     // LDS row stride in floats: 5 * row mod 16 is a bijection -> conflict-free ds_read_b128
constexpr int F_BUF = (128 + BN) * F_LD;     // floats per slab buffer (A region sized for the tallest tile)
constexpr int F_PATCH_LD = 36;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float epi_value(float acc, float bn, float bm, int act, bool has_res, float rv, float alpha) {
    float v = acc + bn + bm;
    if (act == ACT_RELU) v = fmaxf(v, 0.f);
    else if (act == ACT_SIGMOID) v = sigmoidf_(v);
    if (has_res) v = rv + alpha * v;
    return v;
}

// BD: the B operand is a static weight in FRAGMENT order (launch_f32_fragments: float4 index ((j K/8 + kc) 64 + lane) holds
// W[32 j + (lane & 31)][8 kc + 4 (lane >> 5) .. + 3], i.e. what lane `lane` feeds to the four MFMAs of chunk kc for column
// tile j): it goes global -> registers as one coalesced 1 KiB load per wave and chunk -- no LDS store, no LDS read, half the
// slab buffer -- because every memory instruction of this loop costs matrix issue time.
template <int TM, bool BD>
__device__ __forceinline__ void f32_tile(const GemmArgs& g, const float* __restrict__ A, const float* __restrict__ B,
                                         float* __restrict__ C, const int m0, const int n0, float* __restrict__ lds) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int M = g.M, N = g.N;
    constexpr int BM = 32 * TM;
    constexpr int NPA = (BM + 63) / 64;      // staging passes over the A tile (64 rows x 16 floats per pass)
    // ---- staging: thread -> (row srow + 64 i, floats sk .. sk + 3) of the slab
    const int srow = tid >> 2, sk = (tid & 3) * 4;
    // (rows may overlap -- the analysis transform of an arbitrary frame size hands the recording's frames over as rows with
    // stride hop < K: the extent is the last row's end, not rows x stride)
    const int64_t extA = (int64_t)(M - 1) * g.lda + (g.lda > g.K ? g.lda : g.K), extB = (int64_t)(N - 1) * g.ldb + (g.ldb > g.K ? g.ldb : g.K);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (int)(extA * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, (int)((BD ? (int64_t)((N + 31) / 32 * 32) * g.K : extB) * 4), 0x00020000);
    const int voA = (int)(((int64_t)(m0 + srow) * g.lda + sk) * 4), voB = (int)(((int64_t)(n0 + srow) * g.ldb + sk) * 4);
    const int passA = (int)(64 * g.lda * 4), passB = (int)(64 * g.ldb * 4);
    const bool a1_on = (BM % 64 == 0) || srow < BM % 64;   // the last A pass of an odd TM covers 32 rows only
#ifndef F32_TOOLS
#define F32_TOOLS 0   // tools/gemm_f32_bench.hip only (-DF32_TOOLS=1): the block census below (GemmArgs::narrow_epilogue == 77 turns
                      // GemmArgs::range_flag into a census buffer).  The library never compiles that path: range_flag there is ONE word.
#endif
#ifndef F32_PROBE
#define F32_PROBE 0   // tools: wave 0 of every block adds up where its slabs' clocks go (issue of the 32 MFMAs incl. operand reads | LDS stores incl. the wait for the global loads | barrier) into the census buffer
#endif
#ifndef F32_FAST_EPILOGUE
#define F32_FAST_EPILOGUE 1   // 0: the general epilogue for every launch (A/B, tools)
#endif
#ifndef F32_SPREAD
#define F32_SPREAD 1   // the next slab's global loads one per two MFMA groups of this slab instead of as a burst in front of it
#endif
#ifndef F32_TRANSPOSED
#define F32_TRANSPOSED 0   // 1: the MFMA operands swapped (weights first): the accumulators hold the tile TRANSPOSED -- lane = token row,
                           // a register group = four consecutive output features -- and the fast epilogue stores 16-byte pieces
                           // straight from the registers, no turn through the LDS patch (same products, same bits)
#endif
#ifndef F32_STORE_GUARD
#define F32_STORE_GUARD 1   // 0 (tools): without the wait state behind the fast epilogue's 16-byte stores
#endif
// A buffer store of more than 64 bits reads its data registers AFTER it has issued: a VALU result written to one of them in the
// very next slot is what gets stored.  LLVM pads that slot unless the store's soffset is an SGPR (GCNHazardRecognizer::
// createsVALUHazard) -- and the fast epilogue's stores carry their row offset exactly there.  gfx950 does corrupt such a pair
// inside this kernel (seen twice: F32_TRANSPOSED=1, and the shipped epilogue as `-O1 -g` schedules it -- DESIGN.md 3.1c), so the
// wait state is put there by construction: the asm reads the stored registers, so nothing may redefine them in front of it, and
// it is itself one slot.  (tests/test_store_hazard_scan.py scans every unit's ISA for the pair.)
#if F32_STORE_GUARD
#define F32_AFTER_STORE(o) asm volatile("s_nop 0" : : "v"(o));
#else
#define F32_AFTER_STORE(o)
#endif
#ifndef F32_DMA
#define F32_DMA 0   // 1: global -> LDS by the DMA path (buffer_load ... lds), unpadded 64-byte LDS rows with an XOR swizzle
#endif
#if F32_DMA
    // LDS row r (16 floats = four 16-byte slots) holds its k segment s in slot s ^ ((r >> 2) & 3): the 16 lanes of a
    // ds_read_b128 group then hit 16 distinct bank quads.  The DMA writes lane l's 16 bytes at base + 16 l, so lane l
    // (row l >> 2 of its wave's 16 rows, slot l & 3) FETCHES segment (l & 3) ^ ((row >> 2) & 3).
    constexpr int D_BUF = (128 + BN) * 16;
    const int dseg = ((tid & 3) ^ ((tid >> 4) & 3)) * 4;
    const int dvoA = (int)(((int64_t)(m0 + srow) * g.lda + dseg) * 4), dvoB = (int)(((int64_t)(n0 + srow) * g.ldb + dseg) * 4);
    const bool a1_wave = (BM % 64 == 0) || w < (BM % 64) / 16;   // (wave-uniform: a wave covers 16 rows of a pass)
    typedef __attribute__((address_space(3))) void lds_void;
#define F32_GLOAD(k0, buf)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < NPA; ++i) {                                                            \
        if (i + 1 < NPA || a1_wave)                                                                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(lds + (buf) * D_BUF + (i * 64 + 16 * w) * 16), 16, dvoA, (k0) * 4 + i * passA, 0, 0); \
    }                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(lds + (buf) * D_BUF + (128 + i * 64 + 16 * w) * 16), 16, dvoB, (k0) * 4 + i * passB, 0, 0);
#define F32_LSTORE(buf) __builtin_amdgcn_s_waitcnt(0x0F70);   /* vmcnt(0): this wave's pieces of the next slab have landed */
#define F32_GLOAD1(k0, j)
#undef F32_SPREAD
#define F32_SPREAD 0
#else
    f32x4 ra[NPA], rb[2];
    f32x4 wc[2], wn[2];   // BD: this slab's / the next slab's weight fragments (two chunks of 8 k each)
    const int vw = (int)((((int64_t)(n0 / 32 + w) * (g.K / 8)) * 64 + lane) * 16);
#define F32_GLOAD(k0, buf)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < NPA; ++i) {                                                            \
        if (i + 1 < NPA || a1_on) ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voA, (k0) * 4 + i * passA, 0)); \
    }                                                                                                            \
    if constexpr (!BD) {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, voB, (k0) * 4 + i * passB, 0)); \
    } else {                                                                                                     \
        wn[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, vw, ((k0) / 8) * 1024, 0)); \
        wn[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, vw, ((k0) / 8 + 1) * 1024, 0)); \
    }
    // load j of the next slab (A passes first, then the two B / weight pieces): F32_SPREAD issues them one per two MFMA groups
#define F32_GLOAD1(k0, j)                                                                                        \
    {                                                                                                            \
        if ((j) < NPA) {                                                                                         \
            if ((j) + 1 < NPA || a1_on) ra[(j) < NPA ? (j) : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voA, (k0) * 4 + (j) * passA, 0)); \
        } else if ((j) < NPA + 2) {                                                                              \
            if constexpr (!BD) rb[(j) - NPA] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, voB, (k0) * 4 + ((j) - NPA) * passB, 0)); \
            else wn[(j) - NPA] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, vw, ((k0) / 8 + (j) - NPA) * 1024, 0)); \
        }                                                                                                        \
    }
#define F32_LSTORE(buf)                                                                                          \
    {                                                                                                            \
        float* as_ = lds + (buf) * F_BUF + srow * F_LD + sk;                                                     \
        _Pragma("unroll") for (int i = 0; i < NPA; ++i) {                                                        \
            if (i + 1 < NPA || a1_on) *reinterpret_cast<f32x4*>(as_ + i * 64 * F_LD) = ra[i];                    \
        }                                                                                                        \
        if constexpr (!BD) {                                                                                     \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(as_ + (128 + i * 64) * F_LD) = rb[i]; \
        }                                                                                                        \
    }
#endif
    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[i] = f32x16{0};
    const int nk = g.K / F_BK;
#if F32_PROBE
    const unsigned long long t_in_ = (w == 0) ? __builtin_readcyclecounter() : 0;
#endif
    F32_GLOAD(0, 0)
    F32_LSTORE(0)
    if constexpr (BD) { wc[0] = wn[0]; wc[1] = wn[1]; }
    __syncthreads();
#if F32_DMA
    constexpr int S_BUF = D_BUF, S_LD = 16;
    const int sw4 = ((c >> 2) & 3) * 4;
    const int aoff = c * 16, boff = (128 + 32 * w + c) * 16;
#define F32_SLOT(ch) (((8 * (ch) + 4 * h)) ^ sw4)
#else
    constexpr int S_BUF = F_BUF, S_LD = F_LD;
    const int aoff = c * F_LD + 4 * h, boff = (128 + 32 * w + c) * F_LD + 4 * h;
#define F32_SLOT(ch) (8 * (ch))
#endif
    // epilogue operands: lane -> rows (lane >> 3) + 8 j of each 32 x 32 tile, columns n .. n + 3
    const float* __restrict__ bias = g.bias;
    const float* __restrict__ res = g.residual;
    const int act = g.act;
    const bool bias_m = bias && g.bias_along_m, bias_n = bias && !g.bias_along_m, has_res = res != nullptr;
    const int64_t ldc = g.ldc, ldr = g.ldr;
    const float alpha = g.alpha;
    const bool wide = (ldc % 4 == 0) && (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                      (!has_res || ((ldr % 4 == 0) && ((reinterpret_cast<uintptr_t>(res) & 15) == 0)));
    // The fast form of the epilogue (round 5): a wave in its epilogue shares its SIMD with three waves issuing MFMAs back to
    // back, and EVERY instruction it issues waits for a slot between them -- the slab-clock probe (tools, F32_PROBE) put the
    // last slab + epilogue at 15 k clocks of a K = 512 tile's 100 k and at 37 k of 162 k for the QKV tiles, ~220 vector-ALU
    // instructions per 32 x 32 tile (64-bit row addresses, row / column bound compares, exec masks, the bias_m select).  With
    // the rows addressed through buffer descriptors -- one per-lane offset per tile, the row piece as a SCALAR offset, rows
    // past M dropped (stores) or read as zero (residual) by the bounds check -- a row piece costs its arithmetic and nothing
    // else.  (The scalar offset takes part in the bounds check on gfx950: tools/buffer_bounds_probe.hip, loads and stores.)
    // Conditions: 16-byte pieces, whole 128-column panels, column bias or none, no sigmoid, operands below 2 GiB.
    const bool fast = F32_FAST_EPILOGUE && wide && !bias_m && (N % BN == 0) && act != ACT_SIGMOID &&
                      (int64_t)M * ldc * 4 < ((int64_t)1 << 31) && (!has_res || (int64_t)M * ldr * 4 < ((int64_t)1 << 31));
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(C, 0, fast ? (int)((int64_t)M * ldc * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(res), 0, (fast && has_res) ? (int)((int64_t)M * ldr * 4) : 0, 0x00020000);
    f32x4 rv[2][4];   // residual pieces of the tile being written and of the next one
#if F32_TRANSPOSED
    // (accumulator layout: lane (c, h) = token row m0 + 32 i + c, registers 4 q .. 4 q + 3 = features n0 + 32 w + 8 q + 4 h .. + 3)
    auto prefetch_fast = [&](int i, int slot, int lane_) {
        const int vo_ = (int)((((int64_t)m0 + (lane_ & 31)) * ldr + n0 + 32 * w + 4 * (lane_ >> 5)) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            rv[slot][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, vo_, (int)((32 * i * ldr + 8 * q) * 4), 0));
    };
#else
    auto prefetch_fast = [&](int i, int slot, int lane_) {
        const int vo_ = (int)((((int64_t)m0 + (lane_ >> 3)) * ldr + n0 + 32 * w + (lane_ & 7) * 4) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            rv[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, vo_, (int)((32 * i + 8 * j) * ldr * 4), 0));
    };
#endif
    // (lane_ is `lane` behind a compiler barrier at the call sites below: the 64-bit row addresses of the epilogue must
    // not be formed before the K loop and carried through it -- the loop has no registers to spare)
    auto prefetch = [&](int i, int slot, int lane_) {
        const int n_ = n0 + 32 * w + (lane_ & 7) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + 32 * i + (lane_ >> 3) + 8 * j;
            const int mc = m < M ? m : M - 1;
            rv[slot][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (has_res && n_ < N) {
                if (wide) {
                    rv[slot][j] = *reinterpret_cast<const f32x4*>(res + (int64_t)mc * ldr + n_);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rv[slot][j][e] = res[(int64_t)mc * ldr + (n_ + e < N ? n_ + e : N - 1)];
                }
            }
        }
    };
#ifndef F32_ABLATE
#define F32_ABLATE 0   // tools: timing probes (wrong results): 1 no slab barrier, 2 no global loads / LDS stores, 4 no LDS operand reads; 8 (right results): accumulators in AGPRs
#endif
    f32x4 abl_a = {1.f, 2.f, 3.f, (float)lane}, abl_b = {0.5f, 0.25f, (float)w, 1.f};
#define F32_COMPUTE(buf)                                                                                         \
    {                                                                                                            \
        const float* as = lds + (buf) * S_BUF + aoff;                                                            \
        const float* bs = lds + (buf) * S_BUF + boff;                                                            \
        _Pragma("unroll") for (int ch = 0; ch < 2; ++ch) {                                                       \
            f32x4 b = abl_b;                                                                                     \
            if constexpr (BD) b = wc[ch];                                                                        \
            else if (!(F32_ABLATE & 4)) b = *reinterpret_cast<const f32x4*>(bs + F32_SLOT(ch));                  \
            f32x4 a[TM];                                                                                         \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                     \
                a[i] = abl_a;                                                                                    \
                if (!(F32_ABLATE & 4)) a[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * S_LD + F32_SLOT(ch)); \
            }                                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                      \
                if (F32_SPREAD && spread_k0 >= 0 && !(e & 1)) { F32_GLOAD1(spread_k0, ch * 2 + (e >> 1)) __builtin_amdgcn_sched_barrier(0); } \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                 \
                    if (F32_ABLATE & 8) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i][e]), "v"(b[e])); \
                    else if (F32_TRANSPOSED) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[e], a[i][e], acc[i], 0, 0, 0); \
                    else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[e], acc[i], 0, 0, 0);          \
                }                                                                                                \
                if (F32_SPREAD && spread_k0 >= 0) __builtin_amdgcn_sched_barrier(0);                             \
            }                                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
        }                                                                                                        \
    }
    // slabs 0 .. nk - 2: the next slab travels global -> registers under this slab's MFMAs, then registers -> LDS
#if F32_PROBE
    const bool probe = w == 0 && g.narrow_epilogue == 77 && g.range_flag;
    unsigned long long pc_ = 0, pl_ = 0, pb_ = 0, pt_ = 0;
    const unsigned long long t_loop_ = probe ? __builtin_readcyclecounter() : 0;
#define F32_CLK(var) if (probe) var = __builtin_readcyclecounter();
#else
#define F32_CLK(var)
#endif
    int spread_k0 = -1;
    for (int kt = 0; kt + 1 < nk; ++kt) {
#if F32_PROBE
        unsigned long long t0_ = 0, t1_ = 0, t2_ = 0, t3_ = 0;
#endif
        F32_CLK(t0_)
        if (F32_SPREAD) spread_k0 = (kt + 1) * F_BK;
        else if (!(F32_ABLATE & 2)) { F32_GLOAD((kt + 1) * F_BK, (kt + 1) & 1) }
        __builtin_amdgcn_sched_barrier(0);   // (the compiler otherwise sinks the loads below the MFMAs, next to their LDS stores)
        F32_COMPUTE(kt & 1)
        F32_CLK(t1_)
        if (!(F32_ABLATE & 2)) { F32_LSTORE((kt + 1) & 1) }
        if constexpr (BD) { wc[0] = wn[0]; wc[1] = wn[1]; }
        F32_CLK(t2_)
        if (!(F32_ABLATE & 1)) __syncthreads();
        F32_CLK(t3_)
#if F32_PROBE
        pc_ += t1_ - t0_; pl_ += t2_ - t1_; pb_ += t3_ - t2_; pt_ += 1;
#endif
    }
#if F32_PROBE
    if (probe && lane == 0) {
        unsigned long long* pr = reinterpret_cast<unsigned long long*>(g.range_flag) + 4096 + 4 * blockIdx.x;
        atomicAdd(pr + 0, pc_); atomicAdd(pr + 1, pl_); atomicAdd(pr + 2, pb_); atomicAdd(pr + 3, pt_);
        atomicAdd(pr + 4096 + 0, t_loop_ - t_in_);   // prologue: first slab global -> LDS, barrier
        atomicAdd(pr + 4096 + 1, 1ull);              // tiles
    }
    const unsigned long long t_out_ = probe ? __builtin_readcyclecounter() : 0;
#define F32_EPILOGUE_PROBE                                                                                       \
    if (probe && lane == 0) {                                                                                    \
        unsigned long long* pr = reinterpret_cast<unsigned long long*>(g.range_flag) + 4096 + 4 * blockIdx.x;    \
        atomicAdd(pr + 4096 + 2, __builtin_readcyclecounter() - t_out_);   /* the last slab + the epilogue */     \
    }
#else
#define F32_EPILOGUE_PROBE
#endif
#undef F32_CLK
    // the last slab: the first tile's epilogue operands travel under its MFMAs
    {
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));
        if (fast) { if (has_res) prefetch_fast(0, 0, lane_); }
        else prefetch(0, 0, lane_);
    }
    spread_k0 = -1;
    F32_COMPUTE((nk - 1) & 1)
    __syncthreads();
#undef F32_COMPUTE
#undef F32_SLOT
#undef F32_GLOAD
#undef F32_GLOAD1
#undef F32_LSTORE

    // ---- epilogue: every 32 x 32 tile takes a turn through the wave's LDS patch and leaves as 16-byte row pieces
    // (the slab buffers are free: the loop ended with a barrier)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    float* patch = lds + w * (32 * F_PATCH_LD);
    const int col4 = (lane_e & 7) * 4, n = n0 + 32 * w + col4, erow = lane_e >> 3;
    const bool col_ok = n < N;
    float bn[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias_n) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bn[e] = bias[n + e < N ? n + e : N - 1];
    }
#if F32_TRANSPOSED
    if (fast) {
        // The accumulators hold the tile transposed (see F32_COMPUTE): a lane's four registers of a group are four consecutive
        // output features of ITS token row -- a 16-byte piece of the output row as it stands.  No turn through LDS.
        const int ce = lane_e & 31, he = lane_e >> 5, nw = n0 + 32 * w + 4 * he;
        const int voC = (int)((((int64_t)m0 + ce) * ldc + nw) * 4);
        const bool relu = act == ACT_RELU;
        f32x4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bq[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (bias_n) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bq[q][e] = bias[nw + 8 * q + e];
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i + 1 < TM && has_res) prefetch_fast(i + 1, (i + 1) & 1, lane_e);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[i][4 * q + e] + bq[q][e];
                    if (relu) t = fmaxf(t, 0.f);
                    if (has_res) t = rv[i & 1][q][e] + alpha * t;
                    o[e] = t;
                }
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, voC, (int)((32 * i * ldc + 8 * q) * 4), 0);
                F32_AFTER_STORE(o)
#if F32_TRANSPOSED >= 2
                // the store reads its four data registers AFTER it has issued; the next piece's first VALU result must not land in
                // them before that (tools/store_war_probe.hip: more wait states than the compiler's hazard table pads on gfx950)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7");
#if F32_TRANSPOSED >= 3
                asm volatile("s_nop 7");
#endif
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
        F32_EPILOGUE_PROBE
        return;
    }
#else
    if (fast) {
        const int voC = (int)((((int64_t)m0 + erow) * ldc + n) * 4);
        const bool relu = act == ACT_RELU;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * h) * F_PATCH_LD + c] = acc[i][r];
            if (i + 1 < TM && has_res) prefetch_fast(i + 1, (i + 1) & 1, lane_e);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(patch + (erow + 8 * j) * F_PATCH_LD + col4);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = v[e] + bn[e];                      // (epi_value's arithmetic with bm = 0: acc + bn is never -0, so the
                    if (relu) t = fmaxf(t, 0.f);                 //  dropped "+ 0.f" changes no bit)
                    if (has_res) t = rv[i & 1][j][e] + alpha * t;
                    o[e] = t;
                }
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, voC, (int)((32 * i + 8 * j) * ldc * 4), 0);
                F32_AFTER_STORE(o)
            }
        }
        F32_EPILOGUE_PROBE
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#if F32_TRANSPOSED
        // (the transposed accumulator: lane c is token row c of the patch, a register group four consecutive columns)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(patch + c * F_PATCH_LD + 8 * q + 4 * h) = f32x4{acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
#else
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * h) * F_PATCH_LD + c] = acc[i][r];
#endif
        // (a wave's own LDS writes and reads are in order: no barrier)
        if (i + 1 < TM) prefetch(i + 1, (i + 1) & 1, lane_e);   // (the registers of acc[i] are free now)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = erow + 8 * j, m = m0 + 32 * i + row;
            const f32x4 v = *reinterpret_cast<const f32x4*>(patch + row * F_PATCH_LD + col4);
            if (m >= M || !col_ok) continue;
            float* dst = C + (int64_t)m * ldc + n;
            const float bm = bias_m ? bias[m] : 0.f;   // (the mask head only: one launch per batch)
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = epi_value(v[e], bn[e], bm, act, has_res, rv[i & 1][j][e], alpha);
            if (wide) {
                *reinterpret_cast<f32x4*>(dst) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) dst[e] = o[e];
            }
        }
    }
    F32_EPILOGUE_PROBE
#undef F32_EPILOGUE_PROBE
}

}  // namespace

// The work of a launch as a line of units (32 rows x 128 columns): index = (batch entry * tiles_n + column panel) * u + row
// unit.  Virtual CU v of `ncu` takes [v U / ncu, (v + 1) U / ncu), its block j of four a quarter of that.
struct F32Plan {
    int u, tiles_n, ncu, max_tm; int64_t U;
#if F32_TOOLS
    unsigned long long* dbg;   // every block records where and when it ran (shader / wall clocks)
#endif
};

template <bool BD>
__global__ __launch_bounds__(256, 4) void gemm_f32_kernel(GemmArgs g, F32Plan p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * F_BUF];
    const int b = blockIdx.x, v = b % p.ncu, j = b / p.ncu;
    const int64_t lo = p.U * v / p.ncu, hi = p.U * (v + 1) / p.ncu;
    int64_t pos = lo + (hi - lo) * j / 4;
    const int64_t end = lo + (hi - lo) * (j + 1) / 4;
    // first tile: 1 + (b + j) % 4 units -- neighbours in the grid AND blocks one stride apart differ, whichever of them the
    // dispatcher puts on one CU; afterwards the tallest tile that fits
    int want = std::min(1 + (b + j) % 4, p.max_tm);
    bool first = true;
#if F32_TOOLS
    if (p.dbg && threadIdx.x == 0) {   // (a census of the blocks -- where and when each ran)
        p.dbg[4 * b + 0] = wall_clock64();
        p.dbg[4 * b + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);   // XCC_ID | HW_ID
        p.dbg[4 * b + 3] = __builtin_readcyclecounter();
    }
#endif
    while (pos < end) {
        const int64_t panel = pos / p.u;
        const int r = (int)(pos - panel * p.u);
        const int tm = (int)std::min<int64_t>(std::min<int64_t>(want, end - pos), p.u - r);
        const int bz = (int)(panel / p.tiles_n), tn = (int)(panel - (int64_t)bz * p.tiles_n);
        const float* A = g.A + (int64_t)bz * g.strideA;
        const float* B = g.B + (int64_t)bz * g.strideB;
        float* C = g.C + (int64_t)bz * g.strideC;
        if (!first) __syncthreads();   // the previous tile's epilogue patches lie over the slab buffers
        first = false;
        switch (tm) {
            case 1: f32_tile<1, BD>(g, A, B, C, 32 * r, tn * BN, lds); break;
            case 2: f32_tile<2, BD>(g, A, B, C, 32 * r, tn * BN, lds); break;
            case 3: f32_tile<3, BD>(g, A, B, C, 32 * r, tn * BN, lds); break;
            default: f32_tile<4, BD>(g, A, B, C, 32 * r, tn * BN, lds); break;
        }
        pos += tm;
        want = p.max_tm;
    }
#if F32_TOOLS
    if (p.dbg && threadIdx.x == 0) { p.dbg[4 * b + 1] = wall_clock64(); p.dbg[4 * b + 3] = __builtin_readcyclecounter() - p.dbg[4 * b + 3]; }
#endif
}

// false: an operand the 32-bit buffer offsets cannot address, or a K that is not a multiple of 16 (use gemm_kernel)
bool launch_gemm_f32(const GemmArgs& g, hipStream_t s, int forced_tm) {
    if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return true;
    if (g.K <= 0 || g.K % F_BK) return false;
    const int64_t lim = (int64_t)1 << 31;
    if ((((int64_t)g.M + 128) * g.lda + g.K) * 4 >= lim || (((int64_t)g.N + 128) * g.ldb + g.K) * 4 >= lim) return false;
    if (g.b_frag32 && (g.N % 32 || g.K % 16 || g.batch != 1 || ((int64_t)g.N * g.K * 4 >= lim))) return false;
    if ((g.lda % 4) || (!g.b_frag32 && (g.ldb % 4)) || (reinterpret_cast<uintptr_t>(g.A) & 15) || (reinterpret_cast<uintptr_t>(g.B) & 15)) return false;   // 16-byte operand loads
    constexpr int NCU = 256;
    F32Plan p;
    p.u = (g.M + 31) / 32;
    p.tiles_n = (g.N + BN - 1) / BN;
    p.ncu = NCU;
    p.max_tm = (forced_tm >= 1 && forced_tm <= 4) ? forced_tm : 4;
    p.U = (int64_t)p.u * p.tiles_n * g.batch;
#if F32_TOOLS
    p.dbg = g.narrow_epilogue == 77 ? reinterpret_cast<unsigned long long*>(g.range_flag) : nullptr;   // (tools/gemm_f32_bench.hip)
#endif
    if (g.b_frag32) hipLaunchKernelGGL(gemm_f32_kernel<true>, dim3(4 * NCU), dim3(256), 0, s, g, p);
    else hipLaunchKernelGGL(gemm_f32_kernel<false>, dim3(4 * NCU), dim3(256), 0, s, g, p);
    return true;
}


// float32 W [N][K] (row stride ld_src; N % 32 == 0, K % 8 == 0) -> fragment order for gemm_f32_kernel<true> (see f32_tile)
__global__ void f32_fragments_kernel(const float* __restrict__ src, int64_t ld_src, float* __restrict__ dst, int N, int K) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one float4 of dst
    const int64_t total = (int64_t)(N / 32) * (K / 8) * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int64_t t = idx >> 6;
    const int kc = (int)(t % (K / 8)), j = (int)(t / (K / 8));
    const float* p = src + (int64_t)(32 * j + (lane & 31)) * ld_src + 8 * kc + 4 * (lane >> 5);
    reinterpret_cast<float4*>(dst)[idx] = make_float4(p[0], p[1], p[2], p[3]);
}
void launch_f32_fragments(const float* src, int64_t ld_src, float* dst, int N, int K, hipStream_t s) {
    const int64_t total = (int64_t)(N / 32) * (K / 8) * 64;
    if (total <= 0) return;
    hipLaunchKernelGGL(f32_fragments_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, ld_src, dst, N, K);
}

}  // namespace css

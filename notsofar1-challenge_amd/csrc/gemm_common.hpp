// Pieces shared by the two MFMA GEMM kernels (gemm.hip: exact float32; gemm_split.hip: split-f16 operands):
// tile constants, the XCD-aware tile mapping and the fused epilogue.
#pragma once
#include "kernels.hpp"
#include "split_f16.hpp"

namespace css {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BN = 128, BK = 32, LDS_LD = 36;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// XCD-aware tile mapping: workgroup L runs on XCD L % 8; give each XCD a contiguous range of tiles (which
// share A row panels), so that a panel is fetched into ONE of the eight L2s.
__device__ __forceinline__ int xcd_tile(int L, int n_tiles) {
    const int q = n_tiles >> 3, rem = n_tiles & 7, xcd = L & 7;
    return xcd * q + (xcd < rem ? xcd : rem) + (L >> 3);
}

// split-f16 operand range (split_f16.hpp): an out-of-range operand is +-inf / NaN and so is every accumulator it feeds
__device__ __forceinline__ void range_check(const f32x16& acc, unsigned int* __restrict__ flag) {
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 16; ++r) bad |= !(fabsf(acc[r]) <= 3.402823466e+38f);   // false for NaN and +-inf
    if (bad) atomicOr(flag, 1u);
}

// One 32x32 accumulator tile -> memory.  All 16 residual / row-bias operands are requested (at
// clamped, always valid addresses) before the first one is consumed, so the epilogue pays one memory
// round trip per tile instead of one per element; out-of-range elements are computed and not stored.
// split_out: columns [0, split_out) of C are written in the split-f16 format (split_f16.hpp; ldc elements = ldc
// floats per row either way, the 32-column groups of the two formats coincide), the rest as float32.
__device__ __forceinline__ void emit_tile(const f32x16& acc, int mb, int n, int M, int N, float* __restrict__ C,
                                          int64_t ldc, const float* __restrict__ bias, int bias_m, int act,
                                          const float* __restrict__ res, int64_t ldr, float alpha, int split_out) {
    const bool n_ok = n < N;
    const int nc = n_ok ? n : N - 1;
    const float bn = (bias && !bias_m) ? bias[nc] : 0.f;
    float rv[16], bm[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        const int mc = m < M ? m : M - 1;
        rv[r] = res ? res[(int64_t)mc * ldr + nc] : 0.f;
        bm[r] = (bias && bias_m) ? bias[mc] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        float v = acc[r] + bn + bm[r];
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == ACT_SIGMOID) v = sigmoidf_(v);
        if (res) v = rv[r] + alpha * v;
        if (n_ok && m < M) {
            if (n < split_out) split_store(reinterpret_cast<_Float16*>(C + (int64_t)m * ldc), n, v);
            else C[(int64_t)m * ldc + n] = v;
        }
    }
}

// emit_tile with 16-byte stores (see emit_tile_pre_wide below for why): the finished tile takes a turn through a
// wave-private LDS patch [32][LDS_LD] and leaves as float4 row pieces.  The caller guarantees ldc % 4 == 0, a
// 16-byte aligned C and N % 4 == 0 (else it uses emit_tile).  mb0 / nb = first row / column of the tile.
__device__ __forceinline__ void emit_tile_wide(const f32x16& acc, int mb0, int h, int c, int nb, int M, int N,
                                               float* __restrict__ C, int64_t ldc, const float* __restrict__ bias,
                                               int bias_m, int act, const float* __restrict__ res, int64_t ldr,
                                               float alpha, int split_out, float* __restrict__ patch) {
    const int n = nb + c, mb = mb0 + 4 * h;
    const int nc = n < N ? n : N - 1;
    const float bn = (bias && !bias_m) ? bias[nc] : 0.f;
    float rv[16], bm[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        const int mc = m < M ? m : M - 1;
        rv[r] = res ? res[(int64_t)mc * ldr + nc] : 0.f;
        bm[r] = (bias && bias_m) ? bias[mc] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = acc[r] + bn + bm[r];
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == ACT_SIGMOID) v = sigmoidf_(v);
        if (res) v = rv[r] + alpha * v;
        patch[((r & 3) + 8 * (r >> 2) + 4 * h) * LDS_LD + c] = v;
    }
    const int lane = c + 32 * h;
    const int col4 = (lane & 7) * 4, n4 = nb + col4;
    if (n4 >= N) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (lane >> 3) + 8 * i, m = mb0 + row;
        const float4 v = *reinterpret_cast<const float4*>(patch + row * LDS_LD + col4);
        if (m >= M) continue;
        float* dst = C + (int64_t)m * ldc;
        if (n4 < split_out) split_store4(reinterpret_cast<_Float16*>(dst), n4, v.x, v.y, v.z, v.w);
        else *reinterpret_cast<float4*>(dst + n4) = v;
    }
}

// The same epilogue with its memory operands requested EARLY: tile_prefetch before the K loop (the bias and the 16
// residual values of a tile are loop invariant -- an in-place residual C = x + alpha*(...) reads x before anyone
// writes it), emit_tile_pre after it.  Saves the dependent global-load round trip(s) that otherwise sit between the
// last MFMA and the first store of every launch (~1.5 us each on launches that are ~20 us long).
struct TilePre {
    float rv[16];
    float bn;
};
__device__ __forceinline__ void tile_prefetch(TilePre& p, int mb, int n, int M, int N, const float* __restrict__ bias,
                                              const float* __restrict__ res, int64_t ldr) {
    const int nc = n < N ? n : N - 1;
    p.bn = bias ? bias[nc] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        const int mc = m < M ? m : M - 1;
        p.rv[r] = res ? res[(int64_t)mc * ldr + nc] : 0.f;
    }
}
__device__ __forceinline__ void emit_tile_pre(const f32x16& acc, const TilePre& p, int mb, int n, int M, int N,
                                              float* __restrict__ C, int64_t ldc, int act, bool has_res, float alpha,
                                              int split_out, int nt = 0) {
    const bool n_ok = n < N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        float v = acc[r] + p.bn;
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == ACT_SIGMOID) v = sigmoidf_(v);
        if (has_res) v = p.rv[r] + alpha * v;
        if (n_ok && m < M) {
            if (nt) {
            if (n < split_out) {
                _Float16 hi_, lo_;
                split_f16(v, hi_, lo_);
                _Float16* row_ = reinterpret_cast<_Float16*>(C + (int64_t)m * ldc);
                const int i_ = split_index(n);
                __builtin_nontemporal_store(hi_, row_ + i_);
                __builtin_nontemporal_store(lo_, row_ + i_ + 32);
            } else {
                __builtin_nontemporal_store(v, C + (int64_t)m * ldc + n);
            }
            } else {
            if (n < split_out) split_store(reinterpret_cast<_Float16*>(C + (int64_t)m * ldc), n, v);
            else C[(int64_t)m * ldc + n] = v;
            }
        }
    }
}

// The same epilogue with 16-byte stores.  In the accumulator layout a lane owns ONE column of 16 rows, so the plain
// epilogue issues 16 four-byte (or 32 two-byte) store instructions per tile, and on this hardware the epilogue of a
// launch is bound by store ISSUE, not bandwidth (MI355X_MICROARCH.md: ~7 B/clk/CU for narrow row-per-lane stores;
// a round-1 timing probe: the epilogue is 5.5 of the 9 us a K = 32 launch takes).  The finished tile
// therefore takes a turn through a wave-private LDS patch [32][LDS_LD] and leaves as 4 float4 row segments per lane
// (split columns: 4 x {8-byte hi, 8-byte lo}).  Requires N % 4 == 0 for the columns of this tile.
__device__ __forceinline__ void emit_tile_pre_wide(const f32x16& acc, const TilePre& p, int mb0, int h, int c, int nb, int M,
                                                   int N, float* __restrict__ C, int64_t ldc, int act, bool has_res,
                                                   float alpha, int split_out, int nt, float* __restrict__ patch) {
    // mb0 = first row of the tile, nb = first column; this lane computed column nb + c of rows mb0 + 4h + {0..3} + 8 i
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = acc[r] + p.bn;
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == ACT_SIGMOID) v = sigmoidf_(v);
        if (has_res) v = p.rv[r] + alpha * v;
        patch[((r & 3) + 8 * (r >> 2) + 4 * h) * LDS_LD + c] = v;
    }
    const int lane = c + 32 * h;
    const int col4 = (lane & 7) * 4, n = nb + col4;
    if (n >= N) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (lane >> 3) + 8 * i, m = mb0 + row;
        const float4 v = *reinterpret_cast<const float4*>(patch + row * LDS_LD + col4);
        if (m >= M) continue;
        float* dst = C + (int64_t)m * ldc;
        if (n < split_out) {
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            f16x4 hi, lo;
            _Float16 a, b;
            split_f16(v.x, a, b); hi[0] = a; lo[0] = b;
            split_f16(v.y, a, b); hi[1] = a; lo[1] = b;
            split_f16(v.z, a, b); hi[2] = a; lo[2] = b;
            split_f16(v.w, a, b); hi[3] = a; lo[3] = b;
            _Float16* rowh = reinterpret_cast<_Float16*>(dst) + split_index(n);
            if (nt) {
                __builtin_nontemporal_store(hi, reinterpret_cast<f16x4*>(rowh));
                __builtin_nontemporal_store(lo, reinterpret_cast<f16x4*>(rowh + 32));
            } else {
                *reinterpret_cast<f16x4*>(rowh) = hi;
                *reinterpret_cast<f16x4*>(rowh + 32) = lo;
            }
        } else {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const f32x4 vv = {v.x, v.y, v.z, v.w};
            if (nt) __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(dst + n));
            else *reinterpret_cast<f32x4*>(dst + n) = vv;
        }
    }
}

// The finished tile written TRANSPOSED: C^T[n][m] at Ct + n * ldct + m (the mask head: rows of the product are tokens, the
// masks want time as their fastest axis).  Through the wave-private patch like the wide epilogue; a store instruction is two
// output rows (columns n of the tile) of 32 consecutive m: 128-byte runs.  Column bias and activation only.
__device__ __forceinline__ void emit_tile_pre_tr(const f32x16& acc, float bn, int mb0, int h, int c, int nb, int M, int N,
                                                 float* __restrict__ Ct, int64_t ldct, int act, float* __restrict__ patch) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = acc[r] + bn;
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == ACT_SIGMOID) v = sigmoidf_(v);
        patch[((r & 3) + 8 * (r >> 2) + 4 * h) * LDS_LD + c] = v;
    }
    const int m = mb0 + c;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int nl = 2 * i + h, n = nb + nl;
        const float v = patch[c * LDS_LD + nl];
        if (m < M && n < N) Ct[(int64_t)n * ldct + m] = v;
    }
}

// Epilogue of a 32 x 32 tile of the attention's q / k projections (bias only) in the ATTENTION kernel's operand order
// (encoder.hip): the tile is one 32-k group `grp` of one head; token m = segment * T + j lands in row tile j / 32, lane
// (j % 32) + 32 h', chunk 4 grp + 2 sub + part (part 0 = hi halves, 1 = lo halves of k = 32 grp + 16 sub + 8 h' .. + 7).
// Through the wave-private patch as above; a lane handles one token row: two (sub, h') items, each 8 values -> one
// 16-byte hi store and one 16-byte lo store; consecutive lanes = consecutive tokens = consecutive 16-byte pieces.
__device__ __forceinline__ void emit_tile_frag(const f32x16& acc, float bn, int mb0, int h, int c, int M, float* __restrict__ frag,
                                               int T, float invT, int heads, int head, int which, int grp,
                                               float* __restrict__ patch) {
#pragma unroll
    for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * h) * LDS_LD + c] = acc[r] + bn;
    const int m = mb0 + c;
    if (m >= M) return;
    const int seg = (int)(((float)m + 0.5f) * invT);   // == m / T for every m < 4e6 and T in 2..256 (checked exhaustively)
    const int j = m - seg * T;
    const int njt = (T + 31) >> 5;
    float4* dst = reinterpret_cast<float4*>(frag) +
                  ((((int64_t)(seg * heads + head) * njt + (j >> 5)) * 2 + which) * 8 + grp * 4) * 64 + (j & 31);
    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int sub = pass, hp = h;   // this lane's items: (sub = 0, h' = h) and (sub = 1, h' = h)
        const float* src = patch + c * LDS_LD + 16 * sub + 8 * hp;
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        f16x8v hi, lo;
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

}  // namespace css

// Stitching of the separated segments (css/css.py:254-312): permutation alignment of adjacent
// segments (training/losses.py PitWrapper), weighted overlap-add, activity gating with
// dilate/erode (utils/numpy_utils.py), and the hand-off layout for the inverse transform.
//
// All kernels are in gather form: an output frame t collects the segments i with i*hop <= t < i*hop + T
// (two with the shipped configuration; bit-exact segment indexing st = i * hop, css.py:183,287), so
// nothing is accumulated with atomics and a frame range can be produced by any rank that holds the
// segments covering it.
#include <algorithm>

#include "kernels.hpp"
#include "lsap.hpp"
#include "split_f16.hpp"

namespace css {

__device__ __forceinline__ double wave_sum_dd(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// The same sum without LDS round trips (__shfl_xor compiles to ds_bpermute: six dependent ones per double and sum):
// four DPP steps inside each row of 16 lanes, the four row totals through scalar registers.  Every lane gets the total.
template <int CTRL>
__device__ __forceinline__ double dpp_add_d(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return v + __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false),
                                __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v = dpp_add_d<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add_d<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add_d<0x141>(v);   // row_half_mirror
    v = dpp_add_d<0x140>(v);   // row_mirror
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * i), __builtin_amdgcn_readlane(lo, 16 * i));
    return (r[0] + r[1]) + (r[2] + r[3]);
}

// ------------------------------------------------------------------------------------------------
// Raw PIT cost of boundary b (segments b | b+1):  cost[a][c] = mean_{F, overlap} loss(L_a - R_c)
// with L = last `overlap` frames of segment b, R = first `overlap` frames of segment b+1, both in
// their RAW channel order (losses.py:50-71; css.py:267-276).  The cost of an already-permuted left
// segment is a row permutation of this matrix, so all boundaries are computed in parallel and the
// sequential dependence of css.py:266-285 is confined to the tiny scan below.
//   input 0: masks   input 1: |separated spectrum|     loss 0: L1   loss 1: squared error
// ------------------------------------------------------------------------------------------------
// frequency chunks per boundary: 39 boundaries alone would occupy 39 of 256 CUs; with 48 a thread takes two (bin, frame)
// elements, all of whose loads are in flight at once (16 chunks: six dependent rounds of loads per thread).  A long
// meeting has boundaries enough (1 208 in 30 min: 58 k blocks of 48 chunks took 383 us against 305 us with 16), so the
// count follows the MEETING's number of boundaries -- not the range a call or a rank computes: the chunk sums are added in
// a fixed order and a cost must be the same bit pattern wherever it is computed.
constexpr int PIT_CH = 48;   // the most (scratch layout)
__host__ __device__ inline int pit_chunks(int64_t num_segments) { return num_segments - 1 < 128 ? PIT_CH : 16; }

// One block per (boundary, frequency chunk) -> partial[b][chunk][16]; a second kernel adds the chunks in a fixed
// order, so the cost is the same bit pattern whatever boundary range or GPU computes it.
__device__ __forceinline__ void pit_cost_body(const StitchArgs& a, int loss, int input, const int64_t b, const int ch, const int nch,
                                              double* __restrict__ partial) {
    __shared__ double red[4][SMAX * SMAX];
    const int S = a.S, F = a.F, T = a.T, ov = a.T - a.hop;
    const int f_lo = (int)((int64_t)F * ch / nch), f_hi = (int)((int64_t)F * (ch + 1) / nch);
    double acc[SMAX * SMAX];
#pragma unroll
    for (int i = 0; i < SMAX * SMAX; ++i) acc[i] = 0.0;
    for (int idx = threadIdx.x; idx < (f_hi - f_lo) * ov; idx += 256) {
        const int f = f_lo + idx / ov, t = idx % ov;
        float l[SMAX], r[SMAX];
#pragma unroll
        for (int k = 0; k < SMAX; ++k) {
            if (k < S) {
                if (input == 0) {
                    l[k] = a.masks[((int64_t)k * F + f) * a.mask_ld + b * T + (T - ov) + t];
                    r[k] = a.masks[((int64_t)k * F + f) * a.mask_ld + (b + 1) * T + t];
                } else {
                    const float2 lv = reinterpret_cast<const float2*>(a.sep)[((b * S + k) * (int64_t)F + f) * T + (T - ov) + t];
                    const float2 rv = reinterpret_cast<const float2*>(a.sep)[(((b + 1) * S + k) * (int64_t)F + f) * T + t];
                    l[k] = sqrtf(lv.x * lv.x + lv.y * lv.y);
                    r[k] = sqrtf(rv.x * rv.x + rv.y * rv.y);
                }
            } else { l[k] = 0.f; r[k] = 0.f; }
        }
#pragma unroll
        for (int i = 0; i < SMAX; ++i)
#pragma unroll
            for (int j = 0; j < SMAX; ++j) {
                const float d = l[i] - r[j];
                acc[i * SMAX + j] += loss == 0 ? (double)fabsf(d) : (double)(d * d);
            }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < SMAX * SMAX; ++i) {
        const double v = wave_sum_dpp(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < SMAX * SMAX) {
        const int q = threadIdx.x;
        partial[(b * PIT_CH + ch) * (SMAX * SMAX) + q] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
    }
}

__global__ __launch_bounds__(256) void pit_cost_kernel(StitchArgs a, int loss, int input, int64_t b_lo,
                                                       double* __restrict__ partial) {
    pit_cost_body(a, loss, input, b_lo + blockIdx.x, blockIdx.y, (int)gridDim.y, partial);
}

__device__ __forceinline__ void pit_cost_final_body(const double* __restrict__ partial, int S, int F, int ov, int64_t b_lo, int64_t b_hi,
                                                    double* __restrict__ costs, int nch, const int64_t i) {
    const int ss = S * S;
    if (i >= (b_hi - b_lo) * ss) return;
    const int64_t b = b_lo + i / ss;
    const int e = (int)(i % ss), q = (e / S) * SMAX + e % S;
    // (all loads first, over the compile-time maximum: a loop over the run-time count is 48 dependent round trips, 18 us)
    double part[PIT_CH];
#pragma unroll
    for (int ch = 0; ch < PIT_CH; ++ch) part[ch] = ch < nch ? partial[(b * PIT_CH + ch) * (SMAX * SMAX) + q] : 0.0;
    double v = 0.0;
#pragma unroll
    for (int ch = 0; ch < PIT_CH; ++ch) v += part[ch];   // (the unused chunks add + 0.0 to a non-negative sum: the same bits)
    costs[b * ss + e] = v / ((double)F * ov);
}

__global__ __launch_bounds__(64) void pit_cost_final_kernel(const double* __restrict__ partial, int S, int F, int ov,
                                                            int64_t b_lo, int64_t b_hi, double* __restrict__ costs, int nch) {
    pit_cost_final_body(partial, S, F, ov, b_lo, b_hi, costs, nch, (int64_t)blockIdx.x * 64 + threadIdx.x);
}

// All boundaries of the sessions of a queue group in one pair of launches (blockIdx.z / .y = session); every session keeps
// its own chunk count, scratch and cost matrix, so each cost is the bit pattern the per-session launch gives.
struct PitMulti { StitchArgs a[PIT_MULTI_MAX]; double* partial[PIT_MULTI_MAX]; double* costs[PIT_MULTI_MAX]; int loss, input; };
__global__ __launch_bounds__(256) void pit_cost_multi_kernel(PitMulti m) {
    const StitchArgs& a = m.a[blockIdx.z];
    const int nch = pit_chunks(a.num_segments);
    if ((int64_t)blockIdx.x >= a.num_segments - 1 || (int)blockIdx.y >= nch) return;
    pit_cost_body(a, m.loss, m.input, blockIdx.x, blockIdx.y, nch, m.partial[blockIdx.z]);
}
__global__ __launch_bounds__(64) void pit_cost_final_multi_kernel(PitMulti m) {
    const StitchArgs& a = m.a[blockIdx.y];
    pit_cost_final_body(m.partial[blockIdx.y], a.S, a.F, a.T - a.hop, 0, a.num_segments - 1, m.costs[blockIdx.y], pit_chunks(a.num_segments),
                        (int64_t)blockIdx.x * 64 + threadIdx.x);
}
void launch_pit_costs_multi(const StitchArgs* a, double* const* scratch, double* const* costs, int n, int loss, int input, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += PIT_MULTI_MAX) {
        const int cnt = std::min(PIT_MULTI_MAX, n - i0);
        PitMulti m{};
        m.loss = loss; m.input = input;
        int64_t nb = 0; int nch = 0;
        for (int i = 0; i < cnt; ++i) {
            m.a[i] = a[i0 + i]; m.partial[i] = scratch[i0 + i]; m.costs[i] = costs[i0 + i];
            nb = std::max<int64_t>(nb, a[i0 + i].num_segments - 1);
            nch = std::max(nch, pit_chunks(a[i0 + i].num_segments));
        }
        if (nb <= 0) continue;
        hipLaunchKernelGGL(pit_cost_multi_kernel, dim3((unsigned)nb, nch, cnt), dim3(256), 0, s, m);
        hipLaunchKernelGGL(pit_cost_final_multi_kernel, dim3((unsigned)((nb * a[i0].S * a[i0].S + 63) / 64), cnt), dim3(64), 0, s, m);
    }
}

size_t pit_cost_scratch_bytes(int64_t n_boundaries) { return (size_t)std::max<int64_t>(n_boundaries, 1) * PIT_CH * SMAX * SMAX * sizeof(double); }

void launch_pit_costs(const StitchArgs& a, int loss, int input, int64_t b_lo, int64_t b_hi, double* scratch, double* costs,
                      hipStream_t s) {
    if (b_hi <= b_lo) return;
    const int nch = pit_chunks(a.num_segments);
    hipLaunchKernelGGL(pit_cost_kernel, dim3((unsigned)(b_hi - b_lo), nch), dim3(256), 0, s, a, loss, input, b_lo, scratch);
    const int64_t n = (b_hi - b_lo) * a.S * a.S;
    hipLaunchKernelGGL(pit_cost_final_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, scratch, a.S, a.F, a.T - a.hop, b_lo,
                       b_hi, costs, nch);
}

// ------------------------------------------------------------------------------------------------
// Sequential permutation scan (css.py:266-285 with losses.py:32-48): perm[0] = identity;
// perm[i+1] = linear_sum_assignment(cost_i with its rows in the order perm[i]).
// The reference minimises with scipy.optimize.linear_sum_assignment (losses.py:43; scipy 1.11.4): the shortest-
// augmenting-path algorithm of scipy's rectangular_lsap (Crouse 2016).  Any exact method finds the same optimum; WHICH
// optimum on exact ties is decided by that algorithm's scan order (columns in descending order, an unassigned column
// wins among equally cheap ones, later ones otherwise) -- and exact ties are the normal case where two speakers are
// silent through a whole overlap (every cost between them is 0: scipy swaps them, a lexicographic search would not).
// So the algorithm is restated step by step, for S <= 4, shared by host and device; tests/test_cabi.py holds it to
// scipy itself on tie-laden matrices.  c[a][b] = cost of assigning row a to column b; out: col4row.
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline void pit_scan_impl(const double* costs, int64_t n_boundaries, int S, int32_t* perms) {
    for (int a = 0; a < S; ++a) perms[a] = a;
    for (int64_t b = 0; b < n_boundaries; ++b) {
        const double* c = costs + b * S * S;
        const int32_t* lp = perms + b * S;
        int32_t* rp = perms + (b + 1) * S;
        V4<V4<double>> m;
        for (int a = 0; a < SMAX; ++a) {
            V4<double> row;
            row.fill(0.0);
            for (int k = 0; k < S; ++k)
                if (a < S) row.set(k, c[lp[a] * S + k]);   // the left segment's channels in their stitched order
            m.set(a, row);
        }
        V4<int> sig;
        lsap_small(m, S, sig);
        for (int a = 0; a < S; ++a) rp[a] = sig.get(a);
    }
}

// Device form of the scan.  A permutation is a state (its lexicographic index, S! <= 24 states), and
// boundary b is a transition table next[b][state_in] -> state_out that depends only on the raw cost
// matrix of b.  All tables are built in parallel (one thread per (boundary, incoming state), in LDS);
// what remains sequential is one LDS lookup per boundary by a single lane, instead of a chain of
// dependent global-memory reads (180 us for 39 boundaries became a few us; 1208 boundaries of a 30-min
// meeting stay under 0.1 ms).  Chunked so that any meeting length fits the LDS.
constexpr int SCAN_CHUNK = 4096;
constexpr int NPMAX = 24;

__device__ __forceinline__ V4<int> perm_from_index(int idx, int S) {
    // idx-th permutation of 0..S-1 in lexicographic order (factorial number system)
    V4<int> avail{0, 1, 2, 3}, p;
    p.fill(0);
    int f = 1;
    for (int i = 2; i < S; ++i) f *= i;  // (S-1)!
    int n = S;
    for (int a = 0; a < S; ++a) {
        const int d = idx / f;
        idx -= d * f;
        p.set(a, avail.get(d));
        for (int i = 0; i < SMAX - 1; ++i)
            if (i >= d) avail.set(i, avail.get(i + 1));
        --n;
        if (n > 1) f /= (n);
    }
    return p;
}
// lexicographic index of a permutation of 0..S-1 (its Lehmer code in the factorial number system)
__device__ __forceinline__ int index_of_perm(const V4<int>& p, int S) {
    int idx = 0;
#pragma unroll
    for (int a = 0; a < SMAX; ++a) {
        int smaller = 0, fact = 1;
#pragma unroll
        for (int b = a + 1; b < SMAX; ++b) {
            if (b < S && a < S && p.get(b) < p.get(a)) ++smaller;
            if (b < S) fact *= (b - a);     // (S - 1 - a)!
        }
        idx += smaller * fact;
    }
    return idx;
}
__device__ __forceinline__ V4<int> unpack_perm(unsigned v) { return V4<int>{(int)(v & 15u), (int)((v >> 4) & 15u), (int)((v >> 8) & 15u), (int)((v >> 12) & 15u)}; }

__global__ __launch_bounds__(256) void pit_scan_kernel(const double* __restrict__ costs, int64_t b_lo, int64_t n_boundaries, int S,
                                                       int32_t* __restrict__ perms) {
    __shared__ uint8_t next[SCAN_CHUNK * NPMAX];
    __shared__ uint8_t state_of[SCAN_CHUNK];
    __shared__ unsigned short ptab[NPMAX];   // the S! permutations in lexicographic order, four bits per entry
    __shared__ int carry;
    int np = 1;
    for (int i = 2; i <= S; ++i) np *= i;
    if ((int)threadIdx.x < np) {
        const V4<int> p = perm_from_index((int)threadIdx.x, S);
        ptab[threadIdx.x] = (unsigned short)(p.a | (p.b << 4) | (p.c << 8) | (p.d << 12));
    }
    if (threadIdx.x == 0) {
        carry = 0;  // identity = lexicographic index 0
        if (b_lo > 0) {   // resume: the lexicographic index of the permutation segment b_lo already has
            V4<int> want;
            want.fill(0);
            for (int a = 0; a < S; ++a) want.set(a, perms[b_lo * S + a]);
            carry = index_of_perm(want, S);
        }
    }
    if (b_lo == 0 && threadIdx.x < S) perms[threadIdx.x] = threadIdx.x;
    __syncthreads();
    for (int64_t b0 = b_lo; b0 < n_boundaries; b0 += SCAN_CHUNK) {
        const int nb = (int)(n_boundaries - b0 < SCAN_CHUNK ? n_boundaries - b0 : SCAN_CHUNK);
        for (int e = threadIdx.x; e < nb * np; e += blockDim.x) {
            const int b = e / np, pin = e - b * np;
            const double* c = costs + (b0 + b) * S * S;
            const V4<int> lp = unpack_perm(ptab[pin]);
            V4<V4<double>> m;
            for (int a = 0; a < SMAX; ++a) {
                V4<double> row;
                row.fill(0.0);
                for (int k = 0; k < SMAX; ++k)
                    if (a < S && k < S) row.set(k, c[lp.get(a) * S + k]);
                m.set(a, row);
            }
            V4<int> sig;
            lsap_small(m, S, sig);
            next[b * np + pin] = (uint8_t)index_of_perm(sig, S);   // the state (lexicographic index) of the assignment found
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int st = carry;
            for (int b = 0; b < nb; ++b) {
                st = next[b * np + st];
                state_of[b] = (uint8_t)st;
            }
            carry = st;
        }
        __syncthreads();
        for (int b = threadIdx.x; b < nb; b += blockDim.x) {
            const V4<int> p = unpack_perm(ptab[state_of[b]]);
            for (int a = 0; a < S; ++a) perms[(b0 + b + 1) * S + a] = p.get(a);
        }
        __syncthreads();
    }
}

void launch_pit_scan(const double* costs, int64_t b_lo, int64_t b_hi, int S, int32_t* perms, hipStream_t s) {
    hipLaunchKernelGGL(pit_scan_kernel, dim3(1), dim3(256), 0, s, costs, b_lo, b_hi, S, perms);
}

void pit_scan_host(const double* costs, int64_t n_boundaries, int S, int32_t* perms) {
    pit_scan_impl(costs, n_boundaries, S, perms);
}

// ------------------------------------------------------------------------------------------------
// helpers shared by the two overlap-add kernels
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float seg_weight(const StitchArgs& a, int64_t seg, int tl) {
    // css.py:258,290: segment 0 is built with is_first_seg, the last one with is_last_seg
    const float* w = seg == 0 ? a.w_first : (seg == a.num_segments - 1 ? a.w_last : a.w_mid);
    return w[tl];
}

// The segments covering frame t, in ascending segment order.  NC = ceil(T / hop) of them at most: two with the shipped
// 3 s / 1.5 s configuration, up to four take the same code (the kernels are instantiated for NC = 2 and 4: slot i is
// ALWAYS segment t / hop - (NC - 1) + i, present or not, so every index is a compile-time constant and nothing lives in
// private scratch); denser segmentations (any hop >= 1 the reference accepts, css.py:144-171) take the NC = 0
// instantiation, which walks first_seg(t) .. last_seg(t) per value.  The sums start from zero exactly as the reference's
// `zeros += w * x` does (css.py:254-295): 0 + w x is w x, so the float32 operation order is the reference's.
constexpr int MAXC = 4;
__host__ __device__ inline int contributor_class(int T, int hop) {
    const int nc = (T + hop - 1) / hop;
    return nc <= 2 ? 2 : (nc <= MAXC ? MAXC : 0);
}
__device__ __forceinline__ int64_t first_seg(const StitchArgs& a, int64_t t) {
    const int64_t lo = t - a.T + 1;
    const int64_t s0 = lo <= 0 ? 0 : (lo + a.hop - 1) / a.hop;
    return s0;
}
__device__ __forceinline__ int64_t last_seg(const StitchArgs& a, int64_t t) {
    const int64_t s1 = t / a.hop;
    return s1 < a.num_segments ? s1 : a.num_segments - 1;
}
template <int NC>
struct Covering {
    int64_t seg[NC];
    int tl[NC];
    float w[NC];      // 0 for an absent slot
    bool on[NC];
    __device__ __forceinline__ void build(const StitchArgs& a, int64_t t) {
        const int64_t i1 = t / a.hop;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int64_t sg = i1 - (NC - 1) + i;
            const int64_t l = t - sg * a.hop;
            on[i] = sg >= 0 && sg < a.num_segments && l >= 0 && l < a.T;
            seg[i] = on[i] ? sg : 0;
            tl[i] = on[i] ? (int)l : 0;
            w[i] = on[i] ? seg_weight(a, sg, (int)l) : 0.f;
        }
    }
    __device__ __forceinline__ float weight_sum() const {
        float ws = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i)
            if (on[i]) ws = __fadd_rn(ws, w[i]);
        return ws;
    }
};

// ------------------------------------------------------------------------------------------------
// Weighted overlap-add of the (permuted) masks + mean over frequency (css.py:254-299, 303-304):
//   mask_st[s][f][t] = (sum_i w_i[t - st_i] * mask_i[perm_i[s]][f][t - st_i]) / (sum_i w_i[t - st_i])
//   activity[s][t]   = mean_f mask_st[s][f][t];   act_b = activity >= th
// The float32 operation order of the reference (multiply, add in ascending segment order, divide) is
// reproduced with explicit round-to-nearest intrinsics so no FMA contraction changes a bit.
// Block = 16 frames x 16 frequency groups (a 60 s meeting is only 3749 frames: 64-frame blocks gave 177 blocks for
// 256 CUs); lanes run over time (contiguous mask reads); the frequency mean is accumulated in float64 and the 16
// partial sums are added in a fixed order.
// ------------------------------------------------------------------------------------------------
constexpr int OM_T = 16, OM_FG = 16;
template <int NC>
__global__ __launch_bounds__(256) void ola_masks_kernel(StitchArgs a, int64_t t_lo, int64_t t_hi) {
    __shared__ double red[OM_FG][OM_T];
    const int s = blockIdx.y;
    const int lane = threadIdx.x & (OM_T - 1), fg = threadIdx.x / OM_T;
    const int64_t t = t_lo + (int64_t)blockIdx.x * OM_T + lane;
    const bool active = t < t_hi;
    double sum = 0.0;
    if constexpr (NC == 0) {
        if (active) {
            const int64_t s0 = first_seg(a, t), s1 = last_seg(a, t);
            float ws = 0.f;
            for (int64_t seg = s0; seg <= s1; ++seg) ws = __fadd_rn(ws, seg_weight(a, seg, (int)(t - seg * a.hop)));
            float* out = a.mask_st + (int64_t)s * a.F * a.T_long + t;
            for (int f = fg; f < a.F; f += OM_FG) {
                float v = 0.f;
                for (int64_t seg = s0; seg <= s1; ++seg) {
                    const int tl = (int)(t - seg * a.hop);
                    const float m = a.masks[((int64_t)a.perms[seg * a.S + s] * a.F + f) * a.mask_ld + seg * a.T + tl];
                    v = __fadd_rn(v, __fmul_rn(seg_weight(a, seg, tl), m));
                }
                v = __fdiv_rn(v, ws);
                out[(int64_t)f * a.T_long] = v;
                sum += (double)v;
            }
        }
    } else {
        if (active) {
            Covering<NC> c;
            c.build(a, t);
            const float wsum = c.weight_sum();
            const float* mp[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i)
                mp[i] = a.masks + (int64_t)(c.on[i] ? a.perms[c.seg[i] * a.S + s] : 0) * a.F * a.mask_ld + c.seg[i] * a.T + c.tl[i];
            float* out = a.mask_st + (int64_t)s * a.F * a.T_long + t;
            for (int f = fg; f < a.F; f += OM_FG) {
                float v = 0.f;
#pragma unroll
                for (int i = 0; i < NC; ++i)
                    if (c.on[i]) v = __fadd_rn(v, __fmul_rn(c.w[i], mp[i][(int64_t)f * a.mask_ld]));
                v = __fdiv_rn(v, wsum);
                out[(int64_t)f * a.T_long] = v;
                sum += (double)v;
            }
        }
    }
    red[fg][lane] = sum;
    __syncthreads();
    if (fg == 0 && active) {
        double tot = 0.0;
#pragma unroll
        for (int g = 0; g < OM_FG; ++g) tot += red[g][lane];
        const float act = (float)(tot / (double)a.F);
        a.activity[(int64_t)s * a.T_long + t] = act;
        a.act_b[(int64_t)s * a.T_long + t] = act >= a.activity_th ? 1 : 0;
    }
}

void launch_ola_masks(const StitchArgs& a, int64_t t_lo, int64_t t_hi, hipStream_t s) {
    if (t_hi <= t_lo) return;
    const dim3 grid((unsigned)((t_hi - t_lo + OM_T - 1) / OM_T), a.S), block(OM_T * OM_FG);
    switch (contributor_class(a.T, a.hop)) {
        case 2: hipLaunchKernelGGL(ola_masks_kernel<2>, grid, block, 0, s, a, t_lo, t_hi); break;
        case 4: hipLaunchKernelGGL(ola_masks_kernel<4>, grid, block, 0, s, a, t_lo, t_hi); break;
        default: hipLaunchKernelGGL(ola_masks_kernel<0>, grid, block, 0, s, a, t_lo, t_hi); break;
    }
}

// ------------------------------------------------------------------------------------------------
// dilate (zero padding) then erode (one padding) of the per-speaker activity (css.py:305-308,
// numpy_utils.py:4-13).  act_final over [t_lo, t_hi) needs act_b over [t_lo - E - Dl, t_hi + E + Dl).
// ------------------------------------------------------------------------------------------------
__global__ void morph_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int64_t T_long, int radius,
                             int erode, int64_t t_lo, int64_t t_hi) {
    const int s = blockIdx.y;
    const int64_t t = t_lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= t_hi) return;
    const uint8_t* row = in + (int64_t)s * T_long;
    uint8_t v = erode ? 1 : 0;
    for (int64_t u = t - radius; u <= t + radius; ++u) {
        const uint8_t x = (u < 0 || u >= T_long) ? (erode ? 1 : 0) : row[u];
        v = erode ? (v & x) : (v | x);
    }
    out[(int64_t)s * T_long + t] = v;
}

void launch_morphology(const StitchArgs& a, int64_t t_lo, int64_t t_hi, hipStream_t s) {
    if (t_hi <= t_lo) return;
    const int64_t d_lo = t_lo - a.erosion < 0 ? 0 : t_lo - a.erosion;
    const int64_t d_hi = t_hi + a.erosion > a.T_long ? a.T_long : t_hi + a.erosion;
    hipLaunchKernelGGL(morph_kernel, dim3((unsigned)((d_hi - d_lo + 255) / 256), a.S), dim3(256), 0, s, a.act_b,
                       a.act_tmp, a.T_long, a.dilation, 0, d_lo, d_hi);
    hipLaunchKernelGGL(morph_kernel, dim3((unsigned)((t_hi - t_lo + 255) / 256), a.S), dim3(256), 0, s, a.act_tmp,
                       a.act_final, a.T_long, a.erosion, 1, t_lo, t_hi);
}

// ------------------------------------------------------------------------------------------------
// Weighted overlap-add of the (permuted) separated spectra, activity gating (css.py:294,298,312) and
// the hand-off layout of the synthesis GEMM: Y[s][t][0..F) = Re, [F..2F) = Im, zero padding to KIp.
// Block = 16 frames of one stream; reads run over time (128-byte runs of float2), the tile is turned
// through LDS so the writes run over the frequency axis.
// ------------------------------------------------------------------------------------------------
constexpr int OT = 16;

template <int NC>
__global__ __launch_bounds__(256) void ola_stft_kernel(StitchArgs a, int64_t t_lo, int64_t t_hi) {
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [OT][2F + 1] (odd stride: no bank conflicts)
    const int TS = 2 * a.F + 1;
    const int s = blockIdx.y;
    const int64_t t0 = t_lo + (int64_t)blockIdx.x * OT;
    const int tx = threadIdx.x & (OT - 1), fy = threadIdx.x >> 4;  // 16 frames x 16 bins per pass
    const int64_t t = t0 + tx;
    const bool active = t < t_hi;
    const float2* sep = reinterpret_cast<const float2*>(a.sep);
    const float gate = active && a.act_final[(int64_t)s * a.T_long + t] ? 1.f : 0.f;
    if constexpr (NC == 0) {
        int64_t gs0 = 0, gs1 = -1;
        float wsum = 0.f;
        if (active) {
            gs0 = first_seg(a, t); gs1 = last_seg(a, t);
            for (int64_t seg = gs0; seg <= gs1; ++seg) wsum = __fadd_rn(wsum, seg_weight(a, seg, (int)(t - seg * a.hop)));
        }
        for (int f = fy; f < a.F; f += 16) {
            float re = 0.f, im = 0.f;
            if (active) {
                for (int64_t seg = gs0; seg <= gs1; ++seg) {
                    const int tl = (int)(t - seg * a.hop);
                    const float w = seg_weight(a, seg, tl);
                    const float2 v = sep[((seg * a.S + a.perms[seg * a.S + s]) * (int64_t)a.F + f) * a.T + tl];
                    re = __fadd_rn(re, __fmul_rn(w, v.x));
                    im = __fadd_rn(im, __fmul_rn(w, v.y));
                }
                if (gs1 >= gs0) {
                    re = __fmul_rn(__fdiv_rn(re, wsum), gate);
                    im = __fmul_rn(__fdiv_rn(im, wsum), gate);
                }
            }
            tile[tx * TS + f] = re;
            tile[tx * TS + a.F + f] = im;
        }
    } else {
        Covering<NC> c;
        const float2* pp[NC];
        float wsum = 1.f;
        bool any = false;
        if (active) {
            c.build(a, t);
            wsum = c.weight_sum();
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                any |= c.on[i];
                pp[i] = sep + (c.seg[i] * a.S + (c.on[i] ? a.perms[c.seg[i] * a.S + s] : 0)) * (int64_t)a.F * a.T + c.tl[i];
            }
        }
        for (int f = fy; f < a.F; f += 16) {
            float re = 0.f, im = 0.f;
            if (active && any) {
#pragma unroll
                for (int i = 0; i < NC; ++i)
                    if (c.on[i]) {
                        const float2 v = pp[i][(int64_t)f * a.T];
                        re = __fadd_rn(re, __fmul_rn(c.w[i], v.x));
                        im = __fadd_rn(im, __fmul_rn(c.w[i], v.y));
                    }
                re = __fmul_rn(__fdiv_rn(re, wsum), gate);
                im = __fmul_rn(__fdiv_rn(im, wsum), gate);
            }
            tile[tx * TS + f] = re;
            tile[tx * TS + a.F + f] = im;
        }
    }
    __syncthreads();
    const float lvl = a.y_split ? level_gain(a.level) : 1.f;
    for (int r = 0; r < OT; ++r) {
        if (t0 + r >= t_hi) break;
        float* out = a.Y + ((int64_t)s * a.T_long + t0 + r) * a.KIp;
        for (int j = threadIdx.x; j < a.KIp; j += 256) {
            const float v = j < 2 * a.F ? tile[r * TS + j] : 0.f;  // zero K padding
            if (a.y_split) split_store(reinterpret_cast<_Float16*>(out), j, v * lvl);   // operand rows of the synthesis GEMM
            else out[j] = v;
        }
    }
}

void launch_ola_stft(const StitchArgs& a, int64_t t_lo, int64_t t_hi, hipStream_t s) {
    if (t_hi <= t_lo) return;
    const size_t lds = (size_t)OT * (2 * a.F + 1) * sizeof(float);
    const dim3 grid((unsigned)((t_hi - t_lo + OT - 1) / OT), a.S), block(256);
    switch (contributor_class(a.T, a.hop)) {
        case 2: hipLaunchKernelGGL(ola_stft_kernel<2>, grid, block, lds, s, a, t_lo, t_hi); break;
        case 4: hipLaunchKernelGGL(ola_stft_kernel<4>, grid, block, lds, s, a, t_lo, t_hi); break;
        default: hipLaunchKernelGGL(ola_stft_kernel<0>, grid, block, lds, s, a, t_lo, t_hi); break;
    }
}

}  // namespace css

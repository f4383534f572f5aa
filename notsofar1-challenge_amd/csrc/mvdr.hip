// Mask-based MVDR beamformer of one segment (css/css_with_conformer/utils/mvdr_util.py):
//   make_wta (:50) -> get_mask_scm (:58) -> calc_bfcoeffs (:69) -> get_bf (:78), then the floored mask
//   multiplication of css/css.py:222-227.
// The 7x7 spatial covariance matrices are accumulated and solved in float64 (the reference does
// this in complex64, which leaves it ~2e-5 from the exact answer; float64 keeps our
// own distance from the exact answer negligible, so the distance to the reference is the reference's).
#include <algorithm>
#include <cstdlib>

#include "kernels.hpp"

namespace css {

constexpr int NC = 7;        // microphones of the NOTSOFAR array (utils/mic_array_model.py:4)
constexpr int NPACK = 49;    // Hermitian 7x7: 7 real diagonal + 21 complex upper-triangle entries

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ int valid_frames(int64_t stft_frames, int64_t seg, int hop, int T) {
    const int64_t tv = stft_frames - seg * (int64_t)hop;
    return (int)(tv < 0 ? 0 : (tv > T ? T : tv));
}

// ------------------------------------------------------------------------------------------------
// Winner-take-all masks + masked spatial covariance.
//   w_k[f,t] = mask_k[f,t] if it is the maximum over the S speaker masks and the summed noise mask,
//              else 1e-10                                            (mvdr_util.py:50-55)
//   Phi_k[f] = sum_t w_k[f,t] x[f,t] x[f,t]^H  (+ 1e-15 I)            (mvdr_util.py:61-65)
//
// One wave per (segment, bin).  The 14 Re / Im plane rows and the mask rows of the bin are read ONCE, as whole
// contiguous rows, into a wave-private LDS tile (the four masks used to re-read the planes 64 bytes at a time).  A frame
// has one winner, so
//   Phi_k = sum_{t: k wins} (m_k - 1e-10) P_t  +  1e-10 sum_t P_t,        P_t = x_t x_t^H  (7 real + 21 complex entries)
// and the outer product of a frame is formed once, for its winner, instead of once per mask: the frames are sorted by
// winner into four lists (wave ballots), the 16 lanes of group k walk list k -- every lane busy on every step, a third
// of the float64 work of the mask-major loop -- and keep two accumulator sets, the weighted sum and the plain sum; the
// plain sums of the four groups add up to sum_t P_t (a frame with tied winners is in each of their lists, but counts
// towards the plain sum only in the first one).  The 16 lanes' partial sums are combined by a four-step DPP
// reduce-scatter and leave through LDS as 49 consecutive doubles per mask.
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}
// the partner lane's value under DPP control CTRL
template <int CTRL>
__device__ __forceinline__ double dpp_get(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row; every lane ends up with the total
__device__ __forceinline__ double row16_sum(double v) {
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror: lane i <-> 7 - i
    v = dpp_add<0x140>(v);   // row_mirror:      lane i <-> 15 - i
    return v;
}

constexpr int SCM_WAVES = 2;   // bins per block

template <int NI, int TS_ = NI * 64>   // 64-frame pieces of a segment: 4 (T <= 256) or 8 (T <= 512); TS_: the tile's row length
__global__ __launch_bounds__(64 * SCM_WAVES) void scm_kernel(MvdrArgs a) {
    extern __shared__ __attribute__((aligned(16))) float scm_lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;   // (wave-uniform for the compiler: the rows' buffer descriptors live in scalar registers)
    const int k = lane >> 4, l16 = lane & 15;
    const int f = blockIdx.x * SCM_WAVES + wave, segl = blockIdx.y;
    const int64_t seg = a.seg_lo + segl;
    const int nm = a.S + 1;
    const int F = a.F, T = a.T;
    if (f >= F) return;   // (whole waves; no block-wide barrier below)
    const int tv = valid_frames(a.stft_frames, seg, a.hop, T);
    const int64_t st = seg * (int64_t)a.hop;
    // wave-private tile: x[14][TS] (rows c: Re, 7 + c: Im), m[4][TS], TS = NI * 64 (whole 64-frame pieces, so that a piece is
    // stored without a lane mask), then the four frame lists (uint16) and 4 x 98 doubles
    // (TS_ = 192 for the shipped 3 s segments, with the group totals laid over the tile -- dead by then: 15.4 KB per wave, ten
    // waves per CU instead of eight.  The launch is 10 280 waves: 5.02 rounds of 2 048 slots are six rounds, 4.02 of 2 560 five.)
    constexpr int TS = TS_;
    constexpr bool TOT_OVER_TILE = TS_ < NI * 64;
    float* xs = scm_lds + (size_t)wave * ((14 + 4) * TS + 2 * TS + (TOT_OVER_TILE ? 0 : 4 * 2 * NPACK * 2));
    float* ms = xs + 14 * TS;
    unsigned short* lists = reinterpret_cast<unsigned short*>(ms + 4 * TS);    // [4][TS]
    double* tot = reinterpret_cast<double*>(TOT_OVER_TILE ? xs : ms + 4 * TS + 2 * TS);   // [4][2 * NPACK] group totals (8-byte aligned)
    // ---- the bin's rows, contiguous along time: every load of the 18 rows is in flight before the first LDS store
    // (row by row, a wave waited out 18 memory round trips).  Buffer loads: a row is a buffer of `tv` floats (0 for a mask
    // row the model does not have), frames past it read as zero by the hardware's bounds check -- no lane mask, no branch
    // per load (72 loads and stores of this phase each sat in their own exec-mask bracket: ~600 scalar instructions per
    // wave); pieces entirely past the segment are skipped with a wave-uniform test.
    {
        float v[2 * NC + 4][NI];
#pragma unroll
        for (int r = 0; r < 2 * NC + 4; ++r) {
            const float* src = r < 2 * NC ? a.X + ((int64_t)(r % NC) * 2 * F + (r / NC) * F + f) * a.T_ld + st
                                          : a.masks + ((int64_t)(r - 2 * NC) * F + f) * a.mask_ld + seg * (int64_t)T;
            const bool row_ok = r < 2 * NC + nm;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, row_ok ? tv * 4 : 0, 0x00020000);
#pragma unroll
            for (int i = 0; i < NI; ++i)
                v[r][i] = (64 * i < tv) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (lane + 64 * i) * 4, 0, 0)) : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 2 * NC + 4; ++r)
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (64 * i < tv) xs[r * TS + lane + 64 * i] = v[r][i];   // rows 14 .. 17 are the mask rows (ms = xs + 14 TS)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- frames by winner: list j holds the frames mask j wins (bit 15: the frame's first winner)
    const uint8_t* ov = a.wta_override ? a.wta_override + (seg * F + f) * (int64_t)T : nullptr;
    int cnt[4] = {0, 0, 0, 0};
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int t0 = 0; t0 < tv; t0 += 64) {
        const int t = t0 + lane;
        const bool ok = t < tv;
        float mv[4], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mv[j] = (ok && j < nm) ? ms[j * TS + t] : -INFINITY;
            mx = fmaxf(mx, mv[j]);
        }
        const int ovv = (ok && ov) ? ov[t] : -1;
        bool seen = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // ties keep every tied mask, like mask == mask_max in the reference
            const bool win = ok && j < nm && (ov ? (ovv >= 16 ? ((ovv >> j) & 1) != 0 : ovv == j) : (mv[j] == mx));
            const unsigned long long b = __ballot(win);
            if (win) lists[j * TS + cnt[j] + __popcll(b & lt)] = (unsigned short)(t | (seen ? 0 : 0x8000));
            cnt[j] += __popcll(b);
            seen |= win;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- group k walks its list.  Two accumulator sets per lane: the weighted sum and the plain sum -- held CROSSWISE in
    // the two halves of the group (lanes 0..7: acc = weighted, pl = plain; lanes 8..15: acc = plain, pl = weighted), so that
    // the first step of the reduce-scatter below needs no selects: every lane keeps `acc` and receives its partner's `pl`
    double acc[NPACK], pl[NPACK];
#pragma unroll
    for (int i = 0; i < NPACK; ++i) { acc[i] = 0.0; pl[i] = 0.0; }
    const int nk = k == 0 ? cnt[0] : (k == 1 ? cnt[1] : (k == 2 ? cnt[2] : cnt[3]));
    const bool upper = l16 >= 8;
    for (int i = l16; i < nk; i += 16) {
        const unsigned e = lists[k * TS + i];
        const int t = e & 0x7fff;
        const double first_ = (e & 0x8000) ? 1.0 : 0.0;
        const double w_ = (double)ms[k * TS + t] - 1e-10;
        const double w = upper ? first_ : w_, first = upper ? w_ : first_;   // (names as in the lower half)
        double xr[NC], xi[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { xr[c] = (double)xs[c * TS + t]; xi[c] = (double)xs[(NC + c) * TS + t]; }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double p = xr[c] * xr[c] + xi[c] * xi[c];
            acc[c] += w * p;
            pl[c] += first * p;
        }
        int p = NC;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int d = c + 1; d < NC; ++d) {
                const double pr = xr[c] * xr[d] + xi[c] * xi[d];      // Re x_c conj(x_d)
                const double pi = xi[c] * xr[d] - xr[c] * xi[d];      // Im
                acc[p] += w * pr; pl[p] += first * pr;
                acc[p + 1] += w * pi; pl[p + 1] += first * pi;
                p += 2;
            }
        }
    }
    // ---- reduce-scatter over the group's 16 lanes: four DPP exchanges, each halving the number of values a lane keeps
    // (summing all 98 values into all 16 lanes cost 4 x 98 exchanges and 16 copies of every total; this costs
    // 49 + 25 + 13 + 7 and leaves each total in one lane).  A lane keeps the lower or upper half of its values by one bit
    // of its position; its partner keeps the other half and sends the half this lane keeps.
    // value list: acc[0..48] then pl[0..48]; after the steps lane l holds the totals of original indices
    //   j + 49 b0 + 25 b1 + 13 b2 + 7 b3,  j = 0..6,  b0 = l16 >= 8, b1 = (l16 & 7) >= 4, b2 = bit 1, b3 = bit 0
    double v1[49], v2[25], v3[13], v4[7];
    {
        // partner: 15 - l16 (row_mirror), in the other half: its `pl` is what this lane's `acc` is (see above)
#pragma unroll
        for (int j = 0; j < 49; ++j) v1[j] = acc[j] + dpp_get<0x140>(pl[j]);
    }
    {
        const bool up = (l16 & 7) >= 4;      // partner: 7 - (l16 & 7) inside the half row (row_half_mirror)
#pragma unroll
        for (int j = 0; j < 25; ++j) {
            const double lo = v1[j], hi = j + 25 < 49 ? v1[j + 25] : 0.0;
            const double keep = up ? hi : lo, send = up ? lo : hi;
            v2[j] = keep + dpp_get<0x141>(send);
        }
    }
    {
        const bool up = (l16 & 2) != 0;      // partner: l16 ^ 2 (quad_perm [2,3,0,1])
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            const double lo = v2[j], hi = j + 13 < 25 ? v2[j + 13] : 0.0;
            const double keep = up ? hi : lo, send = up ? lo : hi;
            v3[j] = keep + dpp_get<0x4E>(send);
        }
    }
    {
        const bool up = (l16 & 1) != 0;      // partner: l16 ^ 1 (quad_perm [1,0,3,2])
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const double lo = v3[j], hi = j + 7 < 13 ? v3[j + 7] : 0.0;
            const double keep = up ? hi : lo, send = up ? lo : hi;
            v4[j] = keep + dpp_get<0xB1>(send);
        }
    }
    // the totals meet in LDS: tot[k][0..48] weighted sums, tot[k][49..97] plain sums
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        const int base = (l16 >= 8 ? 49 : 0);
        const int o1 = ((l16 & 7) >= 4 ? 25 : 0), o2 = ((l16 & 2) ? 13 : 0), o3 = ((l16 & 1) ? 7 : 0);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            // position inside the 49-value half after steps 2..4; slots past a half's end were padding (zero sums)
            const int q3 = j + o3, q2 = q3 + o2, q1 = q2 + o1;
            const bool real = q3 < 13 && q2 < 25 && q1 < 49;
            if (real) tot[k * 98 + base + q1] = v4[j];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- Phi_k = weighted + 1e-10 * (sum of the four plain sums) (+ 1e-15 on the diagonal), 49 consecutive doubles per mask
    for (int e = lane; e < nm * NPACK; e += 64) {
        const int kk = e / NPACK, i = e - kk * NPACK;
        double v = tot[kk * 98 + i] + 1e-10 * ((tot[49 + i] + tot[98 + 49 + i]) + (tot[2 * 98 + 49 + i] + tot[3 * 98 + 49 + i]));
        if (i < NC) v += 1e-15;  // Ri += 1e-15 * I   (mvdr_util.py:63-65)
        a.scm[((seg * nm + kk) * (int64_t)F + f) * NPACK + i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// The same covariances for segments of ANY length (scm_kernel holds a whole segment of the bin in a wave's LDS tile and
// register pieces: T <= 512).  One block per (segment, bin); thread (k, e) owns entry e of mask k's packed matrix and
// walks the frames in order, 128 at a time through LDS; the winners of a chunk are decided by its first 128 threads
// (ties keep every tied mask, like mask == mask_max).  Written for reach, not for speed.
// ------------------------------------------------------------------------------------------------
constexpr int SCM_LONG_CH = 128;

__global__ __launch_bounds__(256) void scm_long_kernel(MvdrArgs a) {
    __shared__ float xs[2 * NC][SCM_LONG_CH];
    __shared__ double ws[4][SCM_LONG_CH];      // (m_k - 1e-10) where mask k wins the frame, else 0
    const int f = blockIdx.x, segl = blockIdx.y;
    const int64_t seg = a.seg_lo + segl;
    const int nm = a.S + 1, F = a.F, T = a.T;
    const int tv = valid_frames(a.stft_frames, seg, a.hop, T);
    const int64_t st = seg * (int64_t)a.hop;
    const uint8_t* ov = a.wta_override ? a.wta_override + (seg * F + f) * (int64_t)T : nullptr;
    const int tid = threadIdx.x;
    const int k = tid / NPACK, e = tid - k * NPACK;
    // entry e: the diagonal (c, c), or (Re | Im) of (c, d), c < d, row by row
    int c = e, d = e, im = 0;
    if (e >= NC) {
        int p = (e - NC) >> 1;
        im = (e - NC) & 1;
        c = 0;
        for (int s_ = 0; s_ < NC - 2; ++s_) if (p >= NC - 1 - c) { p -= NC - 1 - c; ++c; }
        d = c + 1 + p;
    }
    const bool owner = k < nm;   // 4 x 49 = 196 of the 256 threads own an entry
    double acc = 0.0, plain = 0.0;
    for (int t0 = 0; t0 < tv; t0 += SCM_LONG_CH) {
        const int n = min(SCM_LONG_CH, tv - t0);
        __syncthreads();   // the previous chunk has been read
        for (int i = tid; i < 2 * NC * SCM_LONG_CH; i += 256) {
            const int r = i / SCM_LONG_CH, t = i - r * SCM_LONG_CH;
            xs[r][t] = t < n ? a.X[((int64_t)(r % NC) * 2 * F + (r / NC) * F + f) * a.T_ld + st + t0 + t] : 0.f;
        }
        if (tid < SCM_LONG_CH) {
            const int t = tid;
            float mv[4], mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mv[j] = (t < n && j < nm) ? a.masks[((int64_t)j * F + f) * a.mask_ld + seg * (int64_t)T + t0 + t] : -INFINITY;
                mx = fmaxf(mx, mv[j]);
            }
            const int ovv = (t < n && ov) ? ov[t0 + t] : -1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool win = t < n && j < nm && (ov ? (ovv >= 16 ? ((ovv >> j) & 1) != 0 : ovv == j) : (mv[j] == mx));
                ws[j][t] = win ? (double)mv[j] - 1e-10 : 0.0;
            }
        }
        __syncthreads();
        if (owner) {
            for (int t = 0; t < n; ++t) {
                const double xrc = (double)xs[c][t], xic = (double)xs[NC + c][t];
                const double xrd = (double)xs[d][t], xid = (double)xs[NC + d][t];
                const double p = im ? xic * xrd - xrc * xid : xrc * xrd + xic * xid;      // x_c conj(x_d)
                acc += ws[k][t] * p;
                plain += p;
            }
        }
    }
    if (owner) {
        double v = acc + 1e-10 * plain;
        if (e < NC) v += 1e-15;  // Ri += 1e-15 * I   (mvdr_util.py:63-65)
        a.scm[((seg * nm + k) * (int64_t)F + f) * NPACK + e] = v;
    }
}

bool launch_scm(const MvdrArgs& a, hipStream_t s) {
    if (a.T > 512 || css_force_long_path()) {
        hipLaunchKernelGGL(scm_long_kernel, dim3(a.F, a.nseg), dim3(256), 0, s, a);
        return true;
    }
    // per wave: 18 rows of TS floats, 4 lists of TS uint16 (= 2 TS floats), 4 x 98 doubles; TS = 256 or 512
    const int TS = a.T <= 256 ? 256 : 512;
    const size_t per_wave = ((size_t)(14 + 4) * TS + 2 * TS + 4 * 2 * NPACK * 2) * sizeof(float);
    const dim3 grid((a.F + SCM_WAVES - 1) / SCM_WAVES, a.nseg), block(64 * SCM_WAVES);
    if (a.T <= 192) {   // (A/B on one box: 90.2 -> 74.5 us per 40 segments)
        hipLaunchKernelGGL((scm_kernel<4, 192>), grid, block, (size_t)(20 * 192) * sizeof(float) * SCM_WAVES, s, a);
    } else if (a.T <= 256) {
        hipLaunchKernelGGL(scm_kernel<4>, grid, block, per_wave * SCM_WAVES, s, a);
    } else {   // up to 8 s segments: 82 KB of LDS per block
        // (the attribute is per device: set on every launch -- a host-side table lookup -- not behind a process-wide flag
        // that a second device, or a second thread's first launch, would miss; stft.hip does the same)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(scm_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(per_wave * SCM_WAVES)) != hipSuccess)
            return false;
        hipLaunchKernelGGL(scm_kernel<8>, grid, block, per_wave * SCM_WAVES, s, a);
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// Souden MVDR (mvdr_util.py:69-75): for speaker i, Phi_n = Phi_noise + sum_{j != i} Phi_j,
//   Z = solve(Phi_n, Phi_i);  W = Z[:, 0] / (trace(Z) [+ 1e-15 at bin 0 only])
// One thread per (segment, speaker, bin): complex LU with partial pivoting (the algorithm behind
// numpy.linalg.solve / LAPACK gesv; pivot = max |re|+|im| like i?amax) held entirely in registers --
// row exchanges are predicated swaps so every index is a compile-time constant -- then the seven
// right-hand-side columns are pushed through one at a time (only column 0 and the trace are needed).
// ------------------------------------------------------------------------------------------------
struct cplx { double re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx cinv(cplx a) {
    const double d = a.re * a.re + a.im * a.im;
    return {a.re / d, -a.im / d};
}
__device__ __forceinline__ void cswap_if(bool p, cplx& a, cplx& b) {
    const cplx ta = a, tb = b;
    a.re = p ? tb.re : ta.re; a.im = p ? tb.im : ta.im;
    b.re = p ? ta.re : tb.re; b.im = p ? ta.im : tb.im;
}
// element (r, c) of a packed Hermitian matrix
__device__ __forceinline__ cplx herm_at(const double* __restrict__ m, int r, int c) {
    if (r == c) return {m[r], 0.0};
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    const int p = NC + 2 * (lo * NC - lo * (lo + 1) / 2 + (hi - lo - 1));
    return {m[p], r < c ? m[p + 1] : -m[p + 1]};
}

__device__ __forceinline__ void mvdr_solve_body(const MvdrArgs& a, const int64_t gid) {
    const int F = a.F, S = a.S, nm = S + 1;
    const int64_t total = (int64_t)a.nseg * S * F;
    if (gid >= total) return;
    const int f = (int)(gid % F);
    const int spk = (int)((gid / F) % S);
    const int64_t seg = a.seg_lo + gid / ((int64_t)F * S);
    const double* base = a.scm + seg * nm * (int64_t)F * NPACK + (int64_t)f * NPACK;
    const int64_t kstride = (int64_t)F * NPACK;

    // Phi_n = noise + other speakers, expanded to a full matrix
    cplx A[NC][NC];
#pragma unroll
    for (int r = 0; r < NC; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) A[r][c] = {0.0, 0.0};
    for (int j = 0; j < nm; ++j) {
        if (j == spk) continue;
        const double* m = base + j * kstride;
#pragma unroll
        for (int r = 0; r < NC; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const cplx v = herm_at(m, r, c);
                A[r][c].re += v.re;
                A[r][c].im += v.im;
            }
    }
    // LU factorisation, partial pivoting
    int piv[NC];
#pragma unroll
    for (int p = 0; p < NC; ++p) {
        int best = p;
        double bv = fabs(A[p][p].re) + fabs(A[p][p].im);
#pragma unroll
        for (int r = p + 1; r < NC; ++r) {
            const double v = fabs(A[r][p].re) + fabs(A[r][p].im);
            if (v > bv) { bv = v; best = r; }
        }
        piv[p] = best;
#pragma unroll
        for (int r = p + 1; r < NC; ++r) {
            const bool sw = (best == r);
#pragma unroll
            for (int c = 0; c < NC; ++c) cswap_if(sw, A[p][c], A[r][c]);
        }
        const cplx ip = cinv(A[p][p]);
#pragma unroll
        for (int r = p + 1; r < NC; ++r) {
            const cplx l = cmul(A[r][p], ip);
            A[r][p] = l;
#pragma unroll
            for (int c = p + 1; c < NC; ++c) A[r][c] = csub(A[r][c], cmul(l, A[p][c]));
        }
    }
    // right-hand sides: the columns of Phi_spk
    const double* ms = base + spk * kstride;
    cplx w0[NC];
    cplx tr = {0.0, 0.0};
#pragma unroll
    for (int col = 0; col < NC; ++col) {
        cplx z[NC];
#pragma unroll
        for (int r = 0; r < NC; ++r) z[r] = herm_at(ms, r, col);
#pragma unroll
        for (int p = 0; p < NC; ++p)
#pragma unroll
            for (int r = p + 1; r < NC; ++r) cswap_if(piv[p] == r, z[p], z[r]);
#pragma unroll
        for (int r = 1; r < NC; ++r)
#pragma unroll
            for (int c = 0; c < r; ++c) z[r] = csub(z[r], cmul(A[r][c], z[c]));
#pragma unroll
        for (int r = NC - 1; r >= 0; --r) {
#pragma unroll
            for (int c = r + 1; c < NC; ++c) z[r] = csub(z[r], cmul(A[r][c], z[c]));
            z[r] = cmul(z[r], cinv(A[r][r]));
        }
        tr.re += z[col].re;
        tr.im += z[col].im;
        if (col == 0) {
#pragma unroll
            for (int r = 0; r < NC; ++r) w0[r] = z[r];
        }
    }
    if (f == 0) tr.re += 1e-15;  // den[0] += 1e-15 touches frequency bin 0 only (mvdr_util.py:73)
    const cplx it = cinv(tr);
    double* out = a.bfw + ((seg * S + spk) * (int64_t)F + f) * NC * 2;
#pragma unroll
    for (int r = 0; r < NC; ++r) {
        const cplx w = cmul(w0[r], it);
        out[2 * r] = w.re;
        out[2 * r + 1] = w.im;
    }
}

__global__ __launch_bounds__(64) void mvdr_solve_kernel(MvdrArgs a) { mvdr_solve_body(a, (int64_t)blockIdx.x * 64 + threadIdx.x); }
// The sessions of a queue group in ONE launch (blockIdx.y = session): a thread is a chain of ~20 us whatever the launch
// holds, and a 60 s meeting's 30 840 systems are 482 waves for 1 024 SIMDs -- three or six sessions still fit one round.
struct MvdrMulti { MvdrArgs a[MVDR_MULTI_MAX]; };
__global__ __launch_bounds__(64) void mvdr_solve_multi_kernel(MvdrMulti m) { mvdr_solve_body(m.a[blockIdx.y], (int64_t)blockIdx.x * 64 + threadIdx.x); }

void launch_mvdr_solve(const MvdrArgs& a, hipStream_t s) {
    const int64_t total = (int64_t)a.nseg * a.S * a.F;
    hipLaunchKernelGGL(mvdr_solve_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, s, a);
}
void launch_mvdr_solve_multi(const MvdrArgs* a, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += MVDR_MULTI_MAX) {
        const int cnt = std::min(MVDR_MULTI_MAX, n - i0);
        MvdrMulti m{};
        int64_t most = 0;
        for (int i = 0; i < cnt; ++i) { m.a[i] = a[i0 + i]; most = std::max<int64_t>(most, (int64_t)a[i0 + i].nseg * a[i0 + i].S * a[i0 + i].F); }
        if (most > 0) hipLaunchKernelGGL(mvdr_solve_multi_kernel, dim3((unsigned)((most + 63) / 64), cnt), dim3(64), 0, s, m);
    }
}

// ------------------------------------------------------------------------------------------------
// get_bf (mvdr_util.py:78-80): y[f,t] = sum_c conj(W[f,c]) x[c,f,t], then the floored mask
// multiplication sep = y * clip(mask, min=floor) (css.py:226-227).  Without MVDR (single channel or
// mc_mvdr = False) the masked signal is the reference microphone 0 (css.py:204,220).
// Block = (bin, segment), threads run over time.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void beamform_kernel(MvdrArgs a) {
    const int f = blockIdx.x, segl = blockIdx.y;
    const int64_t seg = a.seg_lo + segl;
    const int F = a.F, T = a.T, S = a.S;
    const int tv = valid_frames(a.stft_frames, seg, a.hop, T);
    const int64_t st = seg * (int64_t)a.hop;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const bool ok = t < tv;
        float xr[NC], xi[NC];
        const int nc = a.use_mvdr ? NC : 1;
        for (int c = 0; c < nc; ++c) {
            xr[c] = ok ? a.X[((int64_t)c * 2 * F + f) * a.T_ld + st + t] : 0.f;
            xi[c] = ok ? a.X[((int64_t)c * 2 * F + F + f) * a.T_ld + st + t] : 0.f;
        }
        for (int k = 0; k < S; ++k) {
            float yr, yi;
            if (a.use_mvdr) {
                const double* w = a.bfw + ((seg * S + k) * (int64_t)F + f) * NC * 2;
                double sr = 0.0, si = 0.0;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const double wr = w[2 * c], wi = w[2 * c + 1];
                    sr += wr * xr[c] + wi * xi[c];
                    si += wr * xi[c] - wi * xr[c];
                }
                yr = (float)sr;
                yi = (float)si;
            } else {
                yr = xr[0];
                yi = xi[0];
            }
            const float m = fmaxf(a.masks[((int64_t)k * F + f) * a.mask_ld + seg * (int64_t)T + t], a.mask_floor);
            float2* o = reinterpret_cast<float2*>(a.sep) + ((seg * S + k) * (int64_t)F + f) * T + t;
            *o = make_float2(yr * m, yi * m);
        }
    }
}

void launch_beamform(const MvdrArgs& a, hipStream_t s) {
    // as many waves as the segment has 64-frame pieces (3 s segments: 186 frames on 192 threads instead of 256; ten blocks
    // per CU instead of eight -- 40 x 257 blocks are 4.02 rounds of 2 560 slots instead of 5.02 of 2 048: 50.3 -> 47.3 us)
    const int threads = std::min(256, (a.T + 63) / 64 * 64);
    hipLaunchKernelGGL(beamform_kernel, dim3(a.F, a.nseg), dim3(threads), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// Optional power normalisation (css.py:233-247): scale the separated segment so that the energy of
// the sum of its streams matches the reference microphone over the valid frames.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void segment_energy_kernel(MvdrArgs a, double* __restrict__ scratch) {
    __shared__ double red[2][4];
    const int segl = blockIdx.x;
    const int64_t seg = a.seg_lo + segl;
    const int F = a.F, T = a.T, S = a.S;
    const int tv = valid_frames(a.stft_frames, seg, a.hop, T);
    const int64_t st = seg * (int64_t)a.hop;
    double em = 0.0, es = 0.0;
    for (int idx = threadIdx.x; idx < F * tv; idx += 256) {
        const int f = idx / tv, t = idx % tv;
        const float xr = a.X[(int64_t)f * a.T_ld + st + t], xi = a.X[(int64_t)(F + f) * a.T_ld + st + t];
        em += (double)xr * xr + (double)xi * xi;
        float sr = 0.f, si = 0.f;
        for (int k = 0; k < S; ++k) {
            const float2 v = reinterpret_cast<const float2*>(a.sep)[((seg * S + k) * (int64_t)F + f) * T + t];
            sr += v.x;
            si += v.y;
        }
        es += (double)sr * sr + (double)si * si;
    }
    em = wave_sum_d(em);
    es = wave_sum_d(es);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = em; red[1][wave] = es; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double m = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const double s = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        scratch[segl] = sqrt(m) / sqrt(s);  // the 1/(F*tv) of both means cancels
    }
}

__global__ void segment_scale_kernel(MvdrArgs a, const double* __restrict__ scratch) {
    const int segl = blockIdx.y;
    const int64_t seg = a.seg_lo + segl;
    const int64_t n = (int64_t)a.S * a.F * a.T;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g = (float)scratch[segl];
    float2* p = reinterpret_cast<float2*>(a.sep) + seg * n + i;
    float2 v = *p;
    v.x *= g;
    v.y *= g;
    *p = v;
}

void launch_segment_power_norm(const MvdrArgs& a, double* scratch, hipStream_t s) {
    hipLaunchKernelGGL(segment_energy_kernel, dim3(a.nseg), dim3(256), 0, s, a, scratch);
    const int64_t n = (int64_t)a.S * a.F * a.T;
    hipLaunchKernelGGL(segment_scale_kernel, dim3((unsigned)((n + 255) / 256), a.nseg), dim3(256), 0, s, a, scratch);
}

}  // namespace css

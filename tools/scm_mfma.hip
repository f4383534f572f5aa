// EXPERIMENT (round 4, measured and not taken -- DESIGN.md section 3.3): the masked spatial covariances on the float64
// matrix core, v_mfma_f64_16x16x4_f64, instead of the vector ALU (notsofar1-challenge_amd/csrc/mvdr.hip scm_kernel).
// Bit-for-bit it is a different summation order (2.8e-14 absolute on values of 44 against scm_kernel); tools/scm_bench.hip
// builds both and times them.  What it showed on an MI355X: a dense float64 MFMA stream sustains 45-65 ns per instruction
// and SIMD (32-47 TFLOP/s, not the 78.6 of the data sheet; tools/bin/f64rate), the ~100 instructions per (segment, bin)
// wave therefore cost 45-56 us per 40 segments by themselves, and the launch takes 86 us (load + sort + write-out alone:
// 30 us) against 79 us for the vector-ALU kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -mllvm -amdgpu-mfma-vgpr-form tools/scm_bench.hip tools/scm_mfma.hip \
//         notsofar1-challenge_amd/csrc/mvdr.hip -Inotsofar1-challenge_amd/csrc -o tools/bin/scm_bench
#include <cstdlib>

#include "kernels.hpp"

namespace css {

constexpr int NC = 7;
constexpr int NPACK = 49;
constexpr int SCM_WAVES = 2;

__device__ __forceinline__ int valid_frames(int64_t stft_frames, int64_t seg, int hop, int T) {
    const int64_t tv = stft_frames - seg * (int64_t)hop;
    return (int)(tv < 0 ? 0 : (tv > T ? T : tv));
}

// ------------------------------------------------------------------------------------------------
// The same covariances on the float64 matrix core.  With z = [Re x (7); Im x (7)] a frame's outer product is the real
// 14 x 14 matrix z z^T (Re x_c conj(x_d) = G[c][d] + G[7+c][7+d], Im = G[7+c][d] - G[c][7+d]), so
//   sum_{t in list k} w_t z_t z_t^T  =  (w .* Z_k)^T Z_k
// is a [14 x n_k] x [n_k x 14] product: v_mfma_f64_16x16x4_f64 takes four frames per instruction (lane l feeds
// A[l & 15][l >> 4] = w z and B[l >> 4][l & 15] = z of the SAME element: one LDS gather serves both operands; rows / columns 14, 15
// are zero), the accumulators are 4 doubles per lane and mask instead of 98, and nothing is left to reduce across lanes.
// The plain sum of all frames (the 1e-10 regulariser) runs in frame order beside the four lists, an independent
// accumulator chain that keeps the matrix pipe busy between two dependent instructions of a list.
// ------------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NI>   // 64-frame pieces of a segment, T <= 64 NI (every piece is loaded and stored without a test: a test per
                    // load costs a branch and, inside the item loop, a conservative wait for the loads before it)
__global__ __launch_bounds__(64 * SCM_WAVES) void scm_mfma_kernel(MvdrArgs a, int per_wave, int n_items, int dbg) {
    constexpr int TS = 64 * NI + 4;
    extern __shared__ __attribute__((aligned(16))) float scm_lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int nm = a.S + 1;
    const int F = a.F, T = a.T;
    // wave-private tile: x[14][TS] (rows c: Re, 7 + c: Im), m[4][TS], then the four frame lists (uint16, TS entries each).
    // TS = 64 * ceil(T / 64) + 4: lane (i, q) of an operand reads x[i][4 g + q] -- 64 distinct banks.
    float* xs = scm_lds + (size_t)wave * per_wave;
    float* ms = xs + 14 * TS;
    unsigned short* lists = reinterpret_cast<unsigned short*>(ms + 4 * TS);    // [4][TS]
    const int i16 = lane & 15, q = lane >> 4;
    const bool rowok = i16 < 2 * NC;
    const float* xrow = xs + (rowok ? i16 : 0) * TS;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // A wave walks the (segment, bin) items item, item + step, ...: the 18 rows of the NEXT item are requested -- into
    // registers -- before the matrix instructions of the current one start, so a wave's memory phase lies under its own
    // matrix phase (independent waves of one launch start together and stay in step: all loading, then all computing)
    const int step = (int)gridDim.x * SCM_WAVES;
    int item = (int)blockIdx.x * SCM_WAVES + wave;
    if (item >= n_items) return;   // (whole waves; no block-wide barrier below)
    float v[2 * NC + 4][NI];
#define CSS_SCM_REQUEST(it_)                                                                                               \
    {                                                                                                                      \
        const int f_ = (it_) % F;                                                                                          \
        const int64_t seg_ = a.seg_lo + (it_) / F;                                                                         \
        const int tv_ = valid_frames(a.stft_frames, seg_, a.hop, T);                                                       \
        const int64_t st_ = seg_ * (int64_t)a.hop;                                                                         \
        _Pragma("unroll") for (int r = 0; r < 2 * NC + 4; ++r) {                                                           \
            const float* src = r < 2 * NC ? a.X + ((int64_t)(r % NC) * 2 * F + (r / NC) * F + f_) * a.T_ld + st_           \
                                          : a.masks + ((int64_t)(r - 2 * NC) * F + f_) * a.mask_ld + seg_ * (int64_t)T;    \
            const bool row_ok = r < 2 * NC + nm;                                                                           \
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, row_ok ? tv_ * 4 : 0, 0x00020000); \
            _Pragma("unroll") for (int i = 0; i < NI; ++i)                                                                 \
                v[r][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (lane + 64 * i) * 4, 0, 0)); \
        }                                                                                                                  \
    }
    CSS_SCM_REQUEST(item)
    for (;;) {
        const int f = item % F;
        const int64_t seg = a.seg_lo + item / F;
        const int tv = valid_frames(a.stft_frames, seg, a.hop, T);
        // ---- the rows (buffer loads: frames past the segment's end read as zero) -> LDS
#pragma unroll
        for (int r = 0; r < 2 * NC + 4; ++r)
#pragma unroll
            for (int i = 0; i < NI; ++i)
                xs[r * TS + lane + 64 * i] = v[r][i];   // rows 14 .. 17 are the mask rows (ms = xs + 14 TS)
        // ---- frames by winner, from the registers: list j holds the frames mask j wins (a tie is in every tied mask's
        // list, like mask == mask_max in the reference)
        const uint8_t* ov = a.wta_override ? a.wta_override + (seg * F + f) * (int64_t)T : nullptr;
        int cnt[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            {
                const int t = 64 * i + lane;
                const bool ok = t < tv;
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = fmaxf(mx, (ok && j < nm) ? v[2 * NC + j][i] : -INFINITY);
                const int ovv = (ok && ov) ? ov[t] : -1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool win = ok && j < nm && (ov ? (ovv == j) : (v[2 * NC + j][i] == mx));
                    const unsigned long long b = __ballot(win);
                    if (win) lists[j * TS + cnt[j] + __popcll(b & lt)] = (unsigned short)t;
                    cnt[j] += __popcll(b);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int next = item + step;
        if (next < n_items) CSS_SCM_REQUEST(next)
        // ---- four frames per matrix instruction.  Every LDS read of the loop is unconditional (clamped address, selected
        // value): a conditional one becomes an exec-mask bracket with its own wait
        f64x4 accp = {0.0, 0.0, 0.0, 0.0}, acc0 = accp, acc1 = accp, acc2 = accp, acc3 = accp;
        const int np = (tv + 3) >> 2;     // (frames tv .. 64 ceil(tv / 64) of the tile are zero: the loads' bounds check)
        int pg = 0;
#define CSS_SCM_LIST_STEP(acc_)                                                            \
            const bool in_ = e0_ + q < n_;                                                 \
            const int t_ = in_ ? (int)nxt_ : 0;                                            \
            nxt_ = lk_[e0_ + 4];              /* (inside the list's TS entries: TS >= T + 4) */ \
            const float mf_ = mk_[t_], zf_ = xrow[t_];                                     \
            const double w_ = in_ ? (double)mf_ - 1e-10 : 0.0;                             \
            const double z_ = rowok ? (double)zf_ : 0.0;                                   \
            acc_ = __builtin_amdgcn_mfma_f64_16x16x4f64(w_ * z_, z_, acc_, 0, 0, 0);
        // (two loops per list: with the plain sum beside it while that has groups left, alone afterwards -- ties make the
        // lists longer than the segment)
#define CSS_SCM_LIST(k, acc_)                                                              \
        {                                                                                  \
            const int n_ = __builtin_amdgcn_readfirstlane(cnt[k]);                         \
            const unsigned short* lk_ = lists + k * TS + q;                                \
            const float* mk_ = ms + k * TS;                                                \
            unsigned nxt_ = lk_[0];                                                        \
            int e0_ = 0;                                                                   \
            for (; e0_ < n_ && pg < np; e0_ += 4) {                                        \
                const float zpf_ = xrow[4 * pg + q];                                       \
                CSS_SCM_LIST_STEP(acc_)                                                    \
                const double zp_ = rowok ? (double)zpf_ : 0.0;                             \
                accp = __builtin_amdgcn_mfma_f64_16x16x4f64(zp_, zp_, accp, 0, 0, 0);      \
                ++pg;                                                                      \
            }                                                                              \
            for (; e0_ < n_; e0_ += 4) { CSS_SCM_LIST_STEP(acc_) }                         \
        }
        if (dbg & 1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; pg = np; }
        CSS_SCM_LIST(0, acc0)
        CSS_SCM_LIST(1, acc1)
        CSS_SCM_LIST(2, acc2)
        CSS_SCM_LIST(3, acc3)
        for (; pg < np; ++pg) {   // (a segment without winners, or lists shorter than the segment)
            const double zp_ = rowok ? (double)xrow[4 * pg + q] : 0.0;
            accp = __builtin_amdgcn_mfma_f64_16x16x4f64(zp_, zp_, accp, 0, 0, 0);
        }
#undef CSS_SCM_LIST
#undef CSS_SCM_LIST_STEP
        // ---- the five 16 x 16 results meet in LDS (the tile is dead): g[s][row][col], row = q + 4 * reg, col = i16
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double* g = reinterpret_cast<double*>(xs);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = (q + 4 * r) * 16 + i16;
            g[o] = acc0[r]; g[256 + o] = acc1[r]; g[512 + o] = acc2[r]; g[768 + o] = acc3[r]; g[1024 + o] = accp[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- Phi_k = weighted + 1e-10 * plain (+ 1e-15 on the diagonal), 49 consecutive doubles per mask:
        //      7 real diagonal entries, then (Re, Im) of the 21 entries above it, row by row
        for (int e = lane; e < nm * NPACK; e += 64) {
            const int kk = e / NPACK, i = e - kk * NPACK;
            int c, d, im = 0;
            if (i < NC) { c = d = i; }
            else {
                int p = (i - NC) >> 1;
                im = (i - NC) & 1;
                c = 0;
#pragma unroll
                for (int s = 0; s < NC - 2; ++s) if (p >= NC - 1 - c) { p -= NC - 1 - c; ++c; }
                d = c + 1 + p;
            }
            const double* gw = g + kk * 256;
            const double* gp = g + 1024;
            const int o1 = im ? (NC + c) * 16 + d : c * 16 + d, o2 = im ? c * 16 + NC + d : (NC + c) * 16 + NC + d;
            const double wv = im ? gw[o1] - gw[o2] : gw[o1] + gw[o2];
            const double pv = im ? gp[o1] - gp[o2] : gp[o1] + gp[o2];
            double val = wv + 1e-10 * pv;
            if (i < NC) val += 1e-15;  // Ri += 1e-15 * I   (mvdr_util.py:63-65)
            a.scm[((seg * nm + kk) * (int64_t)F + f) * NPACK + i] = val;
        }
        if (next >= n_items) break;
        item = next;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next tile overwrites g
        __builtin_amdgcn_wave_barrier();
    }
#undef CSS_SCM_REQUEST
}

template <int NI>
static bool launch_scm_mfma_ni(const MvdrArgs& a, hipStream_t s) {
    constexpr int TS = 64 * NI + 4;
    int per_wave = 18 * TS + 2 * TS;                // floats: tile + lists
    if (per_wave < 5 * 256 * 2) per_wave = 5 * 256 * 2;   // the five 16 x 16 float64 results
    const size_t bytes = (size_t)per_wave * SCM_WAVES * sizeof(float);
    const int n_items = a.nseg * a.F;
    // as many blocks as the device holds at once (LDS-bound), each wave walking its items
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    const char* ov_ = getenv("CSS_SCM_BLOCKS_PER_CU");
    int per_cu = ov_ ? atoi(ov_) : (int)((160 * 1024) / bytes);
    if (per_cu < 1) per_cu = 1;
    int blocks = cus * per_cu;
    const int need = (n_items + SCM_WAVES - 1) / SCM_WAVES;
    if (blocks > need) blocks = need;
    const int dbg = getenv("CSS_SCM_DBG") ? atoi(getenv("CSS_SCM_DBG")) : 0;
    // (the attribute is per device: set on every launch that needs it -- a host-side table lookup -- not behind a
    // process-wide flag that a second device, or a second thread's first launch, would miss; stft.hip does the same)
    if (bytes > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(scm_mfma_kernel<NI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)bytes) != hipSuccess)
        return false;
    hipLaunchKernelGGL(scm_mfma_kernel<NI>, dim3(blocks), dim3(64 * SCM_WAVES), bytes, s, a, per_wave, n_items, dbg);
    return true;
}

bool launch_scm_mfma(const MvdrArgs& a, hipStream_t s) {
    if (a.nseg <= 0 || a.F <= 0) return true;
    const int ni = (a.T + 63) / 64;
    return ni <= 1 ? launch_scm_mfma_ni<1>(a, s) : ni == 2 ? launch_scm_mfma_ni<2>(a, s) : ni == 3 ? launch_scm_mfma_ni<3>(a, s)
         : ni == 4 ? launch_scm_mfma_ni<4>(a, s) : ni <= 6 ? launch_scm_mfma_ni<6>(a, s) : launch_scm_mfma_ni<8>(a, s);
}

}  // namespace css

// Validation loss of the reference's training loop on the device (css/training/train.py:411-470 _calc_loss as
// train.py:529 eval_model calls it): per clip, the S x S matrix of mean base losses between every predicted speaker
// output and every ground-truth speaker (the input of PitWrapper, losses.py:50-71) and the noise loss.
#include "kernels.hpp"

namespace css {

constexpr int LOSS_FB = 8;   // bins per block

// X: mixture planes [C][2F][B*T] (clip b in columns [bT, (b+1)T)); masks [(S+1)F][B*T]; G: ground-truth planes
// [B*(S+1)][2F][T] (speaker s of clip b at b*(S+1) + s, the noise at b*(S+1) + S).
// partial[(b * chunks + chunk) * 16 + a*3 + s] = sum over the chunk's bins and all frames of base(pred_a, target_s)
// (a, s < S <= 3; row stride 3 whatever S), [.. + 9] = the noise term, [.. + 10 .. 15] unused; summed by the host in
// chunk order and divided by F*T.
__global__ __launch_bounds__(256) void val_loss_kernel(const float* __restrict__ X, const float* __restrict__ masks,
                                                       const float* __restrict__ G, int B, int T, int F, int S,
                                                       int loss_name, int base, int clip, double* __restrict__ partial) {
    __shared__ double red[4][16];
    const int chunk = blockIdx.x, b = blockIdx.y, chunks = gridDim.x;
    const int64_t ld = (int64_t)B * T;
    double acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.0;
    const float eps = 1.1920928955078125e-07f;
    const int f_lo = chunk * LOSS_FB, f_hi = min(f_lo + LOSS_FB, F);
    for (int e = threadIdx.x; e < (f_hi - f_lo) * T; e += 256) {
        const int f = f_lo + e / T, t = e - (e / T) * T;
        const int64_t col = (int64_t)b * T + t;
        const float mixmag = hypotf(X[(int64_t)f * ld + col], X[(int64_t)(F + f) * ld + col]);   // |STFT| of microphone 0
        float gt[4], pr[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s > S) continue;
            const float* g = G + ((int64_t)(b * (S + 1) + s) * 2 * F) * T;
            float m = hypotf(g[(int64_t)f * T + t], g[(int64_t)(F + f) * T + t]);
            if (clip) m = fminf(m, mixmag);                                    // train.py:431-434
            gt[s] = loss_name == 0 ? m : m / (mixmag + eps);                   // 'masked_mag' / 'mask' targets
            const float mk = masks[((int64_t)s * F + f) * ld + col];
            pr[s] = loss_name == 0 ? mk * mixmag : mk;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if (a >= S || s >= S) continue;
                const double d = (double)pr[a] - (double)gt[s];
                acc[a * 3 + s] += base == 0 ? fabs(d) : d * d;
            }
        const double dn = (double)pr[S] - (double)gt[S];
        acc[9] += base == 0 ? fabs(dn) : dn * dn;
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        double v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 10)
        partial[((int64_t)b * chunks + chunk) * 16 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

int val_loss_chunks(int F) { return (F + LOSS_FB - 1) / LOSS_FB; }

void launch_val_loss(const float* X, const float* masks, const float* G, int B, int T, int F, int S, int loss_name, int base,
                     int clip, double* partial, hipStream_t s) {
    hipLaunchKernelGGL(val_loss_kernel, dim3(val_loss_chunks(F), B), dim3(256), 0, s, X, masks, G, B, T, F, S, loss_name, base,
                       clip, partial);
}

}  // namespace css

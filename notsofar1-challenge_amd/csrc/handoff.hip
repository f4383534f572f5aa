// Device hand-off of the separated streams to the ASR front end.
//
// The reference writes the streams to wav files and Whisper reads them back (asr/asr.py:58,73-74); it notes itself that
// the silent parts could be dropped to save ASR compute (css/css.py:313).  Here a stream stays in HBM: the frames the
// activity gate kept (css.py:303-312, plus a margin) are cut out and concatenated -- the region table is the time map
// back -- and turned into Whisper's input features on the device: reflect-padded 400-point Hann STFT at hop 160, power
// spectrum, slaney mel filterbank (80 or 128 bands), log10 with the 1e-10 floor, the max - 8 clamp, (x + 4) / 4
// (whisper/audio.py log_mel_spectrogram -- a dependency that is not under the reference tree and not installed here:
// its published algorithm is restated; tests compare with the oracle's numpy restatement, which is held to
// transformers.WhisperFeatureExtractor in tests/test_oracle_whisper_pin.py).
#include <cmath>
#include <vector>

#include "kernels.hpp"

namespace css {

constexpr int MEL_NFFT = 400, MEL_BINS = 201, MEL_K = 416;   // hop 160 (the row stride of the frame operand);   // K padded to the GEMM's 32

// ---- host: tables ---------------------------------------------------------------------------------------------------
// analysis matrix [2 * 201][416]: rows f = cos, 201 + f = -sin of 2 pi f n / 400, times the periodic Hann window
void handoff_build_dft(float* m) {
    for (int f = 0; f < MEL_BINS; ++f)
        for (int n = 0; n < MEL_K; ++n) {
            double c = 0.0, s = 0.0;
            if (n < MEL_NFFT) {
                const double w = (double)(float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / MEL_NFFT));
                const double a = 2.0 * M_PI * (double)((int64_t)f * n % MEL_NFFT) / MEL_NFFT;
                c = std::cos(a) * w;
                s = -std::sin(a) * w;
            }
            m[(size_t)f * MEL_K + n] = (float)c;
            m[(size_t)(MEL_BINS + f) * MEL_K + n] = (float)s;
        }
}

static double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}
// librosa.filters.mel(sr=16000, n_fft=400, n_mels) (slaney scale and normalisation), float32 [n_mels][201]
void handoff_build_mel(float* w, int n_mels) {
    const double sr = 16000.0;
    std::vector<double> mel_f(n_mels + 2);
    const double lo = hz_to_mel(0.0), hi = hz_to_mel(sr / 2);
    for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz(lo + (hi - lo) * i / (n_mels + 1));
    for (int i = 0; i < n_mels; ++i) {
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int f = 0; f < MEL_BINS; ++f) {
            const double hz = sr / 2 * f / (MEL_BINS - 1);
            const double lower = (hz - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
            const double upper = (mel_f[i + 2] - hz) / (mel_f[i + 2] - mel_f[i + 1]);
            w[(size_t)i * MEL_BINS + f] = (float)(std::fmax(0.0, std::fmin(lower, upper)) * enorm);
        }
    }
}

// ---- device ---------------------------------------------------------------------------------------------------------
// out[i] = concatenation of the regions of `wav`, reflect-padded by 200 samples at both ends (torch.stft center=True),
// i in [0, n_act + 400); zeros up to `total`.  regions: [nr][2] sample ranges, offs[r] = samples before region r.
__global__ void handoff_gather_kernel(const float* __restrict__ wav, const int64_t* __restrict__ regions,
                                      const int64_t* __restrict__ offs, int nr, int64_t n_act, float* __restrict__ out,
                                      int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float v = 0.f;
    if (i < n_act + MEL_NFFT && n_act > 0) {
        int64_t j = i - MEL_NFFT / 2;
        if (j < 0) j = -j;                                   // reflect (no edge repeat)
        if (j >= n_act) j = 2 * (n_act - 1) - j;
        j = j < 0 ? 0 : j;
        int lo = 0, hi = nr - 1;
        while (lo < hi) {                                    // last region with offs[r] <= j
            const int mid = (lo + hi + 1) >> 1;
            if (offs[mid] <= j) lo = mid; else hi = mid - 1;
        }
        v = wav[regions[2 * lo] + (j - offs[lo])];
    }
    out[i] = v;
}

// spec [402][ld] (rows f: Re, 201 + f: Im; time fastest) -> mel[m][j] = log10(max(sum_f w[m][f] |X|^2, 1e-10)) and the
// running maximum (as an order-preserving integer)
__global__ __launch_bounds__(256) void handoff_mel_kernel(const float* __restrict__ spec, int64_t ld, int64_t nfr,
                                                          const float* __restrict__ w, int n_mels, float* __restrict__ mel,
                                                          int* __restrict__ gmax) {
    __shared__ float pw[MEL_BINS][33];
    const int64_t j0 = (int64_t)blockIdx.x * 32;
    const int tj = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int f = ty; f < MEL_BINS; f += 8) {
        const int64_t j = j0 + tj;
        float p = 0.f;
        if (j < nfr) {
            const float re = spec[(int64_t)f * ld + j], im = spec[(int64_t)(MEL_BINS + f) * ld + j];
            p = re * re + im * im;
        }
        pw[f][tj] = p;
    }
    __syncthreads();
    float best = -INFINITY;
    for (int m = ty; m < n_mels; m += 8) {
        float acc = 0.f;
        const float* wm = w + (size_t)m * MEL_BINS;
        for (int f = 0; f < MEL_BINS; ++f) acc = fmaf(wm[f], pw[f][tj], acc);
        const float v = log10f(fmaxf(acc, 1e-10f));
        if (j0 + tj < nfr) {
            mel[(int64_t)m * nfr + j0 + tj] = v;
            best = fmaxf(best, v);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
    if ((threadIdx.x & 63) == 0 && best > -INFINITY) {
        const int b = __float_as_int(best);
        atomicMax(gmax, b >= 0 ? b : b ^ 0x7fffffff);       // monotone map float -> int
    }
}

__global__ void handoff_norm_kernel(float* __restrict__ mel, int64_t count, const int* __restrict__ gmax) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int b = *gmax;
    const float mx = __int_as_float(b >= 0 ? b : b ^ 0x7fffffff);
    mel[i] = (fmaxf(mel[i], mx - 8.0f) + 4.0f) * 0.25f;
}

void launch_handoff_gather(const float* wav, const int64_t* regions, const int64_t* offs, int nr, int64_t n_act, float* out,
                           int64_t total, hipStream_t s) {
    if (total <= 0) return;
    hipLaunchKernelGGL(handoff_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wav, regions, offs, nr, n_act, out, total);
}
void launch_handoff_mel(const float* spec, int64_t ld, int64_t nfr, const float* w, int n_mels, float* mel, int* gmax, hipStream_t s) {
    if (nfr <= 0) return;
    hipMemsetAsync(gmax, 0x80, sizeof(int), s);             // 0x80808080: below every mapped finite value
    hipLaunchKernelGGL(handoff_mel_kernel, dim3((unsigned)((nfr + 31) / 32)), dim3(256), 0, s, spec, ld, nfr, w, n_mels, mel, gmax);
    const int64_t count = nfr * n_mels;
    hipLaunchKernelGGL(handoff_norm_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, mel, count, gmax);
}

}  // namespace css

#include <cstdio>
#include <hip/hip_runtime.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* out, int n) {
    double a = threadIdx.x * 1e-3, b = 1e-3;
    f64x4 d0 = {0, 0, 0, 0}, d1 = d0;
    for (int it = 0; it < n; ++it) { d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, d1, 0, 0, 0); }
    if (d0[0] + d1[1] == 12345.0) out[0] = d0[0];
}
int main() {
    double* o; hipMalloc(&o, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd)
    for (int n : {500, 5000, 50000}) {
        const int blocks = 256 * waves_per_simd;
        k<<<blocks, 256>>>(o, n); hipDeviceSynchronize();
        hipEventRecord(e0); k<<<blocks, 256>>>(o, n); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mf = 2.0 * n * waves_per_simd;   // MFMAs per SIMD
        printf("waves/SIMD %d, %d x 2 MFMA per wave: %.1f us -> %.1f ns per MFMA per SIMD = %.2f GHz at 64 clk; %.1f TFLOP/s f64\n", waves_per_simd, n, ms * 1e3, ms * 1e6 / mf, 64.0 / (ms * 1e6 / mf), mf * 1024 * 2048 / (ms * 1e-3) / 1e12);
    }
    return 0;
}

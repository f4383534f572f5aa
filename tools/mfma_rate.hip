// Micro-benchmark: sustained rate of v_mfma_f32_32x32x16_f16 in the 24-MFMA / 8-accumulator pattern of
// gemm_split_wd.hip with no memory traffic at all (tools only).  Variants: 1 or 2 waves per SIMD.
#include <cstdio>
#include <hip/hip_runtime.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define M16(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    f16x8 a0, a1, a2, a3, b0, b1;
    for (int i = 0; i < 8; ++i) { a0[i] = (_Float16)(threadIdx.x * 0.001f + i); a1[i] = a0[i] + (_Float16)1; a2[i] = a0[i] + (_Float16)2; a3[i] = a0[i] + (_Float16)3; b0[i] = (_Float16)(i * 0.01f); b1[i] = (_Float16)(i * 0.02f); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0}, d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            M16(a0, b0, c0); M16(a1, b0, c1); M16(a2, b0, c2); M16(a3, b0, c3);
            M16(a0, b1, d0); M16(a1, b1, d1); M16(a2, b1, d2); M16(a3, b1, d3);
            M16(a1, b0, d0); M16(a2, b0, d1); M16(a3, b0, d2); M16(a0, b0, d3);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i] + d0[i] + d1[i] + d2[i] + d3[i];
    if (s == 12345.f) out[0] = s;
}
int main() {
    float* o; hipMalloc(&o, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512}) for (int blocks : {236, 256, 512}) {
        const int iters = 4096;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, o, 16);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double mf = (double)blocks * (threads / 64) * iters * 24;
        printf("blocks %3d x %d waves: %.1f us, %.0f TFLOP/s, %.1f ns per MFMA per wave (= %.1f cycles at 2.4 GHz)\n", blocks, threads / 64,
               ms * 1e3, mf * 32768 / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * 24.0), ms * 1e6 / (iters * 24.0) * 2.4);
    }
}

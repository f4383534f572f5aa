// Semantics probe of ds_read_b64_tr_b16 (gfx950): which four halves does lane l receive when lane i of a 16-lane group
// points at halves [4 (i & 3), +4) of row (i >> 2) of a row-major 4 x 16 block?   (tools only, not shipped)
//   hipcc --offload-arch=gfx950 -O3 tools/tr_probe.hip -o gpurun_out/tr_probe && gpurun_out/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(float* out) {
    __shared__ __attribute__((aligned(16))) __fp16 lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (__fp16)(float)i;   // element value = its index
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(lds + g * 64 + (i >> 2) * 16 + 4 * (i & 3)));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)v[j];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %3.0f %3.0f %3.0f %3.0f\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    return 0;
}

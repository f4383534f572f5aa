#include <cstdio>
#include <hip/hip_runtime.h>
__global__ void empty_k(float* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
int main() {
    hipStream_t st; hipStreamCreate(&st); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 236, 472, 2048}) for (int thr : {256, 512}) {
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(empty_k, dim3(blocks), dim3(thr), 0, st, nullptr);
        hipEventRecord(e0, st);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_k, dim3(blocks), dim3(thr), 0, st, nullptr);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel blocks=%4d threads=%d: %.2f us per launch\n", blocks, thr, 1e3 * ms / 200);
    }
}

// Does a page-locked H2D copy on stream number k (streams created one after the other; the runtime deals them onto its
// hardware queues round robin) run while an earlier stream is busy with a long chain of kernels?  And what does a
// cross-stream event dependency cost?   (tools only)   hipcc --offload-arch=gfx950 -O3 tools/stream_alias_probe.hip -o /tmp/sap && /tmp/sap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0000001f + 1e-9f;
    p[threadIdx.x] = v;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int NS = 12;
    std::vector<hipStream_t> st(NS);
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    float* d; hipMalloc(&d, 1 << 20);
    const size_t bytes = 27u << 20;
    void *hp, *dp; hipHostMalloc(&hp, bytes, hipHostMallocDefault); hipMalloc(&dp, bytes);
    hipEvent_t e0, e1, c0, c1;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&c0); hipEventCreate(&c1);
    // a chain of ~5 ms on stream 0: 500 launches of ~10 us
    auto chain = [&](hipStream_t s, int n) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, s, d, 400); };
    chain(st[0], 50); hipDeviceSynchronize();
    printf("H2D of 27 MB on stream k while stream 0 runs a ~5 ms chain of 500 kernels (copy enqueued AFTER the chain):\n");
    for (int k = 1; k < NS; ++k) {
        hipDeviceSynchronize();
        hipEventRecord(e0, st[0]);
        chain(st[0], 500);
        hipEventRecord(e1, st[0]);
        hipEventRecord(c0, st[k]);
        hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, st[k]);
        hipEventRecord(c1, st[k]);
        hipDeviceSynchronize();
        float chain_ms, copy_start, copy_end;
        hipEventElapsedTime(&chain_ms, e0, e1); hipEventElapsedTime(&copy_start, e0, c0); hipEventElapsedTime(&copy_end, e0, c1);
        printf("  stream %2d: chain %.2f ms; copy starts at %.2f ms, ends at %.2f ms\n", k, chain_ms, copy_start, copy_end);
    }
    printf("cross-stream event dependency: stream 0 launches a kernel, stream k waits for its event and launches one, back and forth 200 times:\n");
    hipEvent_t ev[2]; hipEventCreateWithFlags(&ev[0], hipEventDisableTiming); hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
    for (int k = 0; k < NS; ++k) {
        hipDeviceSynchronize();
        const double t0 = now_ms();
        for (int i = 0; i < 200; ++i) {
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st[0], d, 10);
            hipEventRecord(ev[0], st[0]); hipStreamWaitEvent(st[k], ev[0], 0);
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st[k], d, 10);
            hipEventRecord(ev[1], st[k]); hipStreamWaitEvent(st[0], ev[1], 0);
        }
        hipDeviceSynchronize();
        printf("  stream 0 <-> stream %2d: %.1f us per hop\n", k, 1e3 * (now_ms() - t0) / 400);
    }
    return 0;
}

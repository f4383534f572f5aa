// Does a captured hipGraph shorten a chain of dependent small kernels on this runtime?  One stream, 300 launches of a
// ~3 us kernel (and of an empty one): plain launches vs one graph launch.  (tools only)
//   hipcc --offload-arch=gfx950 -O3 tools/graph_probe.hip -o /tmp/gp && /tmp/gp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void work(float* p, int iters) {
    float v = p[threadIdx.x + blockIdx.x * blockDim.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0000001f + 1e-9f;
    p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 300;
    for (int iters : {0, 200, 2000}) {
        for (int blocks : {64, 2048}) {
            auto chain = [&]() { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(work, dim3(blocks), dim3(256), 0, s, d, iters); };
            chain(); hipStreamSynchronize(s);
            double best_plain = 1e9, best_graph = 1e9;
            for (int r = 0; r < 5; ++r) { const double t0 = now_ms(); chain(); hipStreamSynchronize(s); best_plain = std::min(best_plain, now_ms() - t0); }
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(s, hipStreamCaptureModeGlobal); chain(); hipStreamEndCapture(s, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            hipGraphLaunch(ge, s); hipStreamSynchronize(s);
            for (int r = 0; r < 5; ++r) { const double t0 = now_ms(); hipGraphLaunch(ge, s); hipStreamSynchronize(s); best_graph = std::min(best_graph, now_ms() - t0); }
            printf("iters %4d blocks %4d: plain %.3f ms (%.2f us per launch), graph %.3f ms (%.2f us per launch)\n", iters, blocks,
                   best_plain, 1e3 * best_plain / N, best_graph, 1e3 * best_graph / N);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    }
    return 0;
}

// Stand-alone timing of css::launch_relpos_attention (tools only).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <hip/hip_runtime.h>
#include "kernels.hpp"
using namespace css;
int main(int argc, char** argv) {
    const int nseg = argc > 1 ? atoi(argv[1]) : 40, T = 186, D = 512, H = 8, maxlen = 1000;
    const size_t nq = (size_t)nseg * T * 3 * D, npe = 2 * maxlen * 64, nc = (size_t)nseg * T * D;
    float *qkv, *pe, *ctx;
    hipMalloc(&qkv, nq * 4); hipMalloc(&pe, npe * 4); hipMalloc(&ctx, nc * 4);
    std::vector<float> h(nq);
    unsigned s = 1; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xFFFF) / 32768.0f - 1.0f) * 0.5f; }
    hipMemcpy(qkv, h.data(), nq * 4, hipMemcpyHostToDevice); hipMemcpy(pe, h.data(), npe * 4, hipMemcpyHostToDevice);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // qk_split = 0: q, k, pe read as float32 (exact mode); 1: the same bits interpreted as split-f16 operands --
    // timing only, the values are then not a split of anything
    for (int qks = 0; qks < 2; ++qks) {
        for (int i = 0; i < 3; ++i) launch_relpos_attention(qkv, qks ? qkv : nullptr, pe, ctx, nseg, T, D, H, maxlen, qks, 0, st);
        hipEventRecord(e0, st);
        const int it = 50;
        for (int i = 0; i < it; ++i) launch_relpos_attention(qkv, qks ? qkv : nullptr, pe, ctx, nseg, T, D, H, maxlen, qks, 0, st);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<float> o(nc); hipMemcpy(o.data(), ctx, nc * 4, hipMemcpyDeviceToHost);
        double cs = 0; for (int i = 0; i < 64; ++i) cs += o[12345 + i] * (i + 1);
        unsigned long long hsh = 1469598103934665603ull;   // FNV-1a over every output bit: builds of the same arithmetic must agree
        for (size_t i = 0; i < nc; ++i) { unsigned u; memcpy(&u, &o[i], 4); hsh = (hsh ^ u) * 1099511628211ull; }
        printf("attention %d seg, %s scores: %.2f us per launch, checksum %.9g, hash of all outputs %016llx\n", nseg, qks ? "split-f16" : "float32", 1e3 * ms / it, cs, hsh);
    }
    return 0;
}

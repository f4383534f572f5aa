// Stand-alone timing of the EXACT float32 GEMM (css::gemm_kernel, v_mfma_f32_32x32x2_f32) on the Linear-layer shapes of the
// mask estimator, at the launch heights of one session (M = 7440), two lanes of a shared batch (11160) and a shared batch
// (22320), for every tile layout, next to the sustained rate of the instruction itself (tools only, not shipped).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_f32_bench.hip notsofar1-challenge_amd/csrc/gemm.hip notsofar1-challenge_amd/csrc/gemm_f32.hip \
//         notsofar1-challenge_amd/csrc/gemm_split.hip notsofar1-challenge_amd/csrc/gemm_split_wd.hip -Inotsofar1-challenge_amd/csrc -o tools/bin/gemm_f32_bench
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <hip/hip_runtime.h>
#include "kernels.hpp"
using namespace css;

typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void mfma_rate_kernel(float* out, int iters) {
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1.f, b0 = 0.5f, b1 = 0.25f;
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.f) out[0] = s;
}

int main(int argc, char** argv) {
    struct Shape { int N, K; const char* name; int mode; };   // mode 1: bias + residual (alpha 0.5), 0: bias only
    Shape shapes[] = {{512, 512, "attn-out", 1}, {1024, 512, "ffn-up", 0}, {512, 1024, "ffn-down", 1}, {1536, 512, "qkv", 0}, {512, 1824, "embed", 0}};
    int Ms[] = {7440, 22320, 22320, 22320, 29760};
    if (getenv("GEMM_BENCH_M")) Ms[1] = atoi(getenv("GEMM_BENCH_M"));   // the second height of the table (the headline's shared batch: 44640)
    const int layouts[] = {8, 1, 13, 12, 0};   // round-4 kernel (8 waves) | gemm_f32.hip (tiles up to 128 rows) | up to 96 | up to 64
    const size_t maxA = (size_t)44640 * 2048, maxB = (size_t)2048 * 1824, maxC = (size_t)44640 * 2048;
    float *A, *B, *C, *R;
    hipMalloc(&A, maxA * 4); hipMalloc(&B, maxB * 4 + 4096); hipMalloc(&C, maxC * 4); hipMalloc(&R, maxC * 4);
    std::vector<float> h(maxA);
    unsigned s = 1; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    hipMemcpy(A, h.data(), maxA * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data() + 12345, maxB * 4, hipMemcpyHostToDevice);
    hipMemcpy(R, h.data() + 999, maxC * 4, hipMemcpyHostToDevice);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    {   // the instruction's own sustained rate
        for (int threads : {256, 512}) for (int blocks : {256, 512}) {
            const int iters = 2048;
            hipLaunchKernelGGL(mfma_rate_kernel, dim3(blocks), dim3(threads), 0, st, C, 16);
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(mfma_rate_kernel, dim3(blocks), dim3(threads), 0, st, C, iters);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)blocks * (threads / 64) * iters * 16;
            printf("mfma_f32_32x32x2 rate: %3d blocks x %d waves: %.1f us, %.1f TFLOP/s\n", blocks, threads / 64, ms * 1e3, n * 4096 / (ms * 1e-3) / 1e12);
        }
    }
    {   // census of the persistent blocks of one launch: start / end (100 MHz wall clock), shader cycles, physical CU
        unsigned long long* dbg; hipMalloc(&dbg, 1024 * 96);   // [1024][4] census + [1024][4] slab-clock sums of wave 0 (builds with -DF32_PROBE=1)
        float* Bfc = nullptr; hipMalloc(&Bfc, maxB * 4 + 4096);
        struct Cs { int M, N, K; } cases[] = {{22320, 512, 1824}, {22320, 512, 512}, {22320, 1536, 512}, {7440, 512, 512}};
        for (auto& cs : cases) {
            GemmArgs g{};
            g.A = A; g.lda = cs.K; g.B = B; g.ldb = cs.K; g.C = C; g.ldc = cs.N; g.M = cs.M; g.N = cs.N; g.K = cs.K; g.batch = 1; g.alpha = 1.f;
            g.layout = 1; g.narrow_epilogue = 77; g.range_flag = (unsigned int*)dbg;
            if (getenv("GEMM_BENCH_FRAG")) { launch_f32_fragments(B, cs.K, Bfc, cs.N, cs.K, st); g.B = Bfc; g.b_frag32 = 1; }
            for (int i = 0; i < 3; ++i) launch_gemm(g, st);
            hipMemsetAsync(dbg + 4096, 0, 1024 * 64, st);
            hipEventRecord(e0, st);
            for (int i = 0; i < 10; ++i) launch_gemm(g, st);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> d(12288); hipMemcpy(d.data(), dbg, 1024 * 96, hipMemcpyDeviceToHost);
            {   // -DF32_PROBE=1: where wave 0's slab clocks went, summed over the blocks (10 launches)
                double pc = 0, pl = 0, pb = 0, pn = 0;
                for (int b = 0; b < 1024; ++b) { pc += d[4096 + 4 * b]; pl += d[4096 + 4 * b + 1]; pb += d[4096 + 4 * b + 2]; pn += d[4096 + 4 * b + 3]; }
                double pp = 0, ptl = 0, pe = 0;
                for (int b = 0; b < 1024; ++b) { pp += d[8192 + 4 * b]; ptl += d[8192 + 4 * b + 1]; pe += d[8192 + 4 * b + 2]; }
                if (ptl > 0) printf("   per tile of wave 0 (mean over %.0f tiles): prologue %.0f clocks, %.1f counted slabs x %.0f, last slab + epilogue %.0f\n", ptl, pp / ptl, pn / ptl, (pc + pl + pb) / pn, pe / ptl);
                if (pn > 0) printf("   slab clocks of wave 0 (mean over %.0f slabs): %.0f issuing the slab's MFMAs (operand reads, loads) + %.0f LDS stores (wait for the global loads) + %.0f barrier = %.0f per slab; 32 MFMAs x 4 waves per SIMD = 8192\n",
                                   pn, pc / pn, pl / pn, pb / pn, (pc + pl + pb) / pn);
            }
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int b = 0; b < 1024; ++b) { t0 = std::min(t0, d[4 * b]); t1 = std::max(t1, d[4 * b + 1]); }
            printf("census M=%d N=%d K=%d: %.1f us per launch (events), first start -> last end %.1f us\n", cs.M, cs.N, cs.K, 1e3 * ms / 10, (t1 - t0) / 100.0);
            char name[128]; snprintf(name, sizeof name, "gpurun_out/census_%d_%d_%d.txt", cs.M, cs.N, cs.K);
            FILE* f = fopen(name, "w");
            for (int b = 0; b < 1024; ++b)
                fprintf(f, "%d %.2f %.2f %llu %u %u\n", b, (d[4 * b] - t0) / 100.0, (d[4 * b + 1] - t0) / 100.0, d[4 * b + 3], (unsigned)(d[4 * b + 2] >> 32), (unsigned)d[4 * b + 2]);
            fclose(f);
        }
    }
    const char* only = getenv("GEMM_BENCH_ONLY");
    double tot[8][8] = {{0}};
    for (int mi = 0; mi < 2; ++mi) {
        const int M = Ms[mi];
        for (auto& sh : shapes) {
            if (only && !strstr(sh.name, only)) continue;
            printf("%-9s M=%6d N=%5d K=%5d :", sh.name, M, sh.N, sh.K);
            int li = 0;
            static float* Bf = nullptr;   // GEMM_BENCH_FRAG=1: the weights in fragment order for gemm_f32.hip (GemmArgs::b_frag32), as the library runs it
            if (!Bf) hipMalloc(&Bf, maxB * 4 + 4096);
            const bool fragw = getenv("GEMM_BENCH_FRAG") != nullptr;
            if (fragw) launch_f32_fragments(B, sh.K, Bf, sh.N, sh.K, st);
            for (int layout : layouts) {
                GemmArgs g{};
                g.A = A; g.lda = sh.K; g.B = B; g.ldb = sh.K; g.C = C; g.ldc = sh.N; g.M = M; g.N = sh.N; g.K = sh.K; g.batch = 1; g.alpha = 1.f;
                if (fragw && layout != 8) { g.B = Bf; g.b_frag32 = 1; }
                g.bias = A;
                if (sh.mode == 1) { g.residual = R; g.ldr = sh.N; g.alpha = 0.5f; }
                g.layout = layout;
                for (int i = 0; i < 2; ++i) launch_gemm(g, st);
                hipEventRecord(e0, st);
                const int it = 10;
                for (int i = 0; i < it; ++i) launch_gemm(g, st);
                hipEventRecord(e1, st); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double fl = 2.0 * M * sh.N * sh.K, us = 1e3 * ms / it;
                printf("  L%-2d %7.1f us %5.1f TF %.3f |", layout, us, fl / (us * 1e-6) / 1e12, fl / (us * 1e-6) / 1e12 / 157.3);
                {   // every layout must give the bits of the first one
                    static std::vector<float> ref, got;
                    const size_t ne = (size_t)M * sh.N;
                    got.resize(ne);
                    if (getenv("GEMM_BENCH_NAN")) {   // which elements does one launch write?  (C poisoned first)
                        hipMemsetAsync(C, 0xFF, maxC * 4, st); launch_gemm(g, st); hipStreamSynchronize(st);
                        hipMemcpy(got.data(), C, ne * 4, hipMemcpyDeviceToHost);
                        size_t nan = 0, firstbad = 0; for (size_t q = 0; q < ne; ++q) if (got[q] != got[q]) { if (!nan) firstbad = q; ++nan; }
                        if (nan) printf(" UNWRITTEN %zu (first at row %zu col %zu) |", nan, firstbad / sh.N, firstbad % sh.N);
                    }
                    hipMemcpy(got.data(), C, ne * 4, hipMemcpyDeviceToHost);
                    if (li == 0) ref = got;
                    else if (memcmp(ref.data(), got.data(), ne * 4) != 0) { size_t bad = 0; for (size_t q = 0; q < ne; ++q) { const bool d_ = memcmp(&ref[q], &got[q], 4) != 0; if (d_ && bad < 6 && getenv("GEMM_BENCH_NAN")) printf(" [r %zu c %zu: %.9g vs %.9g]", q / sh.N, q % sh.N, ref[q], got[q]); bad += d_; } printf(" BITS DIFFER (%zu) |", bad); }
                }
                const int mult = (sh.name[0] == 'e') ? 1 : (sh.name[0] == 'f' ? 36 : 18);
                tot[mi][li++] += us * mult;
            }
            printf("\n");
        }
        {   // the mask head: weights [1028][512] are the A operand, the tokens B; row bias + sigmoid; tokens fastest
            printf("%-9s M=%6d N=%5d K=%5d :", "head", 1028, M, 512);
            int li = 0;
            for (int layout : layouts) {
                GemmArgs g{};
                g.A = B; g.lda = 512; g.B = A; g.ldb = 512; g.C = C; g.ldc = M; g.M = 1028; g.N = M; g.K = 512; g.batch = 1; g.alpha = 1.f;
                g.bias = R; g.bias_along_m = 1; g.act = ACT_SIGMOID; g.m_fastest = 1; g.layout = layout;
                for (int i = 0; i < 2; ++i) launch_gemm(g, st);
                hipEventRecord(e0, st);
                const int it = 10;
                for (int i = 0; i < it; ++i) launch_gemm(g, st);
                hipEventRecord(e1, st); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double fl = 2.0 * 1028 * M * 512, us = 1e3 * ms / it;
                printf("  L%-2d %7.1f us %5.1f TF %.3f |", layout, us, fl / (us * 1e-6) / 1e12, fl / (us * 1e-6) / 1e12 / 157.3);
                static std::vector<float> ref, got;
                const size_t ne = (size_t)1028 * M;
                got.resize(ne);
                hipMemcpy(got.data(), C, ne * 4, hipMemcpyDeviceToHost);
                if (li == 0) ref = got;
                else if (memcmp(ref.data(), got.data(), ne * 4) != 0) printf(" BITS DIFFER |");
                ++li;
            }
            printf("\n");
        }
        // the estimator's launch mix: embed x 1, (ffn-up, ffn-down) x 36, (qkv, attn-out) x 18
        const double fl_mix = 2.0 * M * (512.0 * 1824 + 36 * 2 * 512.0 * 1024 + 18 * (1536.0 * 512 + 512.0 * 512));
        printf("   => estimator mix at M=%d:", M);
        for (int li = 0; li < 4; ++li) printf("  L%-2d %8.1f us  frac %.3f |", layouts[li], tot[mi][li], fl_mix / (tot[mi][li] * 1e-6) / 1e12 / 157.3);
        printf("\n");
    }
    return 0;
}

// Is an MFMA's source operand safe from a load that returns into the same register right after the MFMA was issued?  (tools only)
// Four queued v_mfma_f32_32x32x2_f32 (four accumulators) read register X as their FIRST or SECOND source; an LDS load into
// X is issued immediately behind them (inline asm, same register).  Expected accumulators assume every MFMA saw the value X held
// when it was issued.  gfx950 result (profiles/r05_mfma_operand_hazard.txt) decides which operand a kernel may reload early.
#include <cstdio>
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool SECOND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[64 * 4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float other = 1.0f, x = 1.0f;
    for (int it = 0; it < iters; ++it) {
        lds[w * 64 + lane] = (float)(it + 2);          // the value the load will bring: x becomes it + 2 AFTER this iteration's MFMAs
        __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0)
        if (SECOND) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n v_mfma_f32_32x32x2_f32 %1, %4, %5, %1\n v_mfma_f32_32x32x2_f32 %2, %4, %5, %2\n v_mfma_f32_32x32x2_f32 %3, %4, %5, %3\n ds_read_b32 %5, %6\n s_waitcnt lgkmcnt(0)\n s_nop 7\n s_nop 7"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(other), "v"(x), "v"((unsigned)((w * 64 + lane) * 4)) : "memory");
        } else {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %5, %4, %0\n v_mfma_f32_32x32x2_f32 %1, %5, %4, %1\n v_mfma_f32_32x32x2_f32 %2, %5, %4, %2\n v_mfma_f32_32x32x2_f32 %3, %5, %4, %3\n ds_read_b32 %5, %6\n s_waitcnt lgkmcnt(0)\n s_nop 7\n s_nop 7"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(other), "v"(x), "v"((unsigned)((w * 64 + lane) * 4)) : "memory");
        }
        // (x was overwritten by the asm's load behind the compiler's back: tell it)
        asm volatile("" : "=v"(x) : "0"(x));
        x = (float)(it + 2);   // what the register now holds (kept consistent for the compiler)
        asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(x));
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int iters = 200, blocks = 1024;
    float* o; hipMalloc(&o, blocks * 256 * 4);
    // expected: each MFMA adds 2 (k) * 1 * x to every element, x = it + 1 in iteration it: per accumulator element sum_{it} 2 (it + 1); 64 elements summed
    double e = 0; for (int it = 0; it < iters; ++it) e += 2.0 * (it + 1);
    const double expect = e * 64;
    for (int second = 0; second < 2; ++second) {
        if (second) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(256), 0, 0, o, iters);
        else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), 0, 0, o, iters);
        static float h[1024 * 256]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
        long bad = 0; for (int i = 0; i < blocks * 256; ++i) bad += h[i] != (float)expect;
        printf("reloaded register as the %s source of four queued MFMAs: %ld of %d lanes differ from the expected sum %.0f (first value %.0f)\n",
               second ? "SECOND" : "FIRST", bad, blocks * 256, expect, h[0]);
    }
    return 0;
}

// Micro-benchmark (tools only): v_mfma_f32_32x32x2_f32 issued as ONE dependent accumulator chain per wave (what the float32
// attention does) against 2 / 4 independent chains, with 1, 2 or 4 waves per SIMD, with and without memory instructions
// between the MFMAs -- what does a second resident wave buy, and what does a load cost?
//   tools/bin/mfma_f32_chain [kinds|density|valu|chains]   (outputs: profiles/r05_mfma_f32_chain_*.txt, read in
//   profiles/r05_attention_f32.md and DESIGN.md 3.1c / 3.2)
#include <cstdio>
#include <cstring>
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
template <int CH, bool LD, bool LDS>
__global__ __launch_bounds__(64) void k(float* out, const float4* __restrict__ src, int iters) {
    __shared__ float lds[64 * 17];
    const int lane = threadIdx.x;
    float a = lane * 0.001f, b = lane * 0.002f + 1.f;
    f32x16 c[CH];
    for (int i = 0; i < CH; ++i) c[i] = f32x16{0};
    float4 t0 = src[lane], t1 = src[64 + lane];
    lds[lane] = a;
    for (int it = 0; it < iters; ++it) {
        float4 n0 = t0, n1 = t1;
        if (LD) {   // 8 loads per 32 MFMAs, as an operand tile of the attention
            int z = it & 7; asm volatile("" : "+v"(z));
#pragma unroll
            for (int q = 0; q < 4; ++q) { n0.x += src[(z + q) * 64 + lane].x; n1.y += src[(z + q + 4) * 64 + lane].y; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 32 / CH; ++r) {
#pragma unroll
            for (int i = 0; i < CH; ++i) MF(a + t0.x, b + t1.y, c[i]);
            if (LDS && (r & 1)) { lds[lane * 17 + (r & 15)] = c[0][r & 15]; a += lds[((lane + 1) & 63) * 17 + (r & 15)] * 1e-30f; }
        }
        __builtin_amdgcn_sched_barrier(0);
        t0 = n0; t1 = n1;
    }
    float s = 0;
    for (int i = 0; i < CH; ++i) for (int j = 0; j < 16; ++j) s += c[i][j];
    if (s == 12345.f) out[0] = s;
}
template <int CH, bool LD, bool LDS>
void run(float* o, const float4* src, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4}) {
        const int blocks = 1024 * wps, iters = 512;
        hipLaunchKernelGGL((k<CH, LD, LDS>), dim3(blocks), dim3(64), 0, 0, o, src, 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<CH, LD, LDS>), dim3(blocks), dim3(64), 0, 0, o, src, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mf = (double)blocks * iters * 32;
        printf("%-34s %d waves/SIMD: %7.1f us, %6.1f TFLOP/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", name, wps, ms * 1e3,
               mf * 4096 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (iters * 32.0 * wps));
    }
}
// one wave = 32 MFMAs, then NV independent VALU instructions (fma), per iteration: do another wave's MFMAs run under them?
template <int NV, int DW>
__global__ __launch_bounds__(64) void kv(float* out, const float* __restrict__ src, int iters) {
    const int lane = threadIdx.x;
    float a = lane * 0.001f, b = lane * 0.002f + 1.f;
    f32x16 c = {0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane + i;
    float dsum = 0.f;
    for (int it = 0; it < iters; ++it) {
        float d[DW > 0 ? DW : 1];
        if (DW) {
            int z = it & 7; asm volatile("" : "+v"(z));
#pragma unroll
            for (int q = 0; q < DW; ++q) d[q] = src[(z + q) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 32; ++r) MF(a, b, c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
        if (DW) {
#pragma unroll
            for (int q = 0; q < DW; ++q) dsum += d[q];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = dsum;
    for (int j = 0; j < 16; ++j) s += c[j];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.f) out[0] = s;
}
template <int NV, int DW>
void runv(float* o, const float* src, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4}) {
        const int blocks = 1024 * wps, iters = 512;
        hipLaunchKernelGGL((kv<NV, DW>), dim3(blocks), dim3(64), 0, 0, o, src, 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL((kv<NV, DW>), dim3(blocks), dim3(64), 0, 0, o, src, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %d waves/SIMD: %7.1f us, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", name, wps, ms * 1e3, ms * 1e-3 * 2.4e9 / (iters * 32.0 * wps));
    }
}
// what does ONE vector-memory instruction cost the matrix pipe, by kind?  NL loads per 32 MFMAs, consumed one iteration later
//   KIND 0: global_load_dwordx4, 64-bit per-lane address   1: buffer_load_dwordx4 (32-bit offset)   2: global_load_dword
//   3: buffer_load_dword   4: buffer_load_dwordx4 ... lds (LDS DMA, no VGPR return) + nothing reads it   5: ds_read_b128
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int NL>
__global__ __launch_bounds__(64) void km(float* out, const float* __restrict__ src, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 9];
    const int lane = threadIdx.x;
    float a = lane * 0.001f, b = lane * 0.002f + 1.f;
    f32x16 c = {0};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 64 * 64 * 16, 0x00020000);
    f32x4 cur[NL], nxt[NL];
    for (int q = 0; q < NL; ++q) cur[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4 * 9; ++i) lds[lane * 36 + i] = i;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        int z = it & 7; asm volatile("" : "+v"(z));
        const float* p = src + (z * 64 + lane) * 4;
        const int vo = (z * 64 + lane) * 16;
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            if (KIND == 0) nxt[q] = *reinterpret_cast<const f32x4*>(p + q * 256);
            if (KIND == 1) nxt[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, q * 1024, 0));
            if (KIND == 2) nxt[q] = f32x4{p[q * 64], 0.f, 0.f, 0.f};
            if (KIND == 3) nxt[q] = f32x4{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, q * 256, 0)), 0.f, 0.f, 0.f};
            if (KIND == 4) { __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (q & 7) * 256), 16, vo, q * 1024, 0, 0); nxt[q] = cur[q]; }
            if (KIND == 5) nxt[q] = *reinterpret_cast<const f32x4*>(lds + lane * 36 + 4 * ((q + z) & 7));
            if (KIND == 6) { typedef float f32x2 __attribute__((ext_vector_type(2))); const f32x2 t2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, q * 512, 0)); nxt[q] = f32x4{t2.x, t2.y, 0.f, 0.f}; }
            if (KIND == 7) {   // P.V through LDS: per 8 "loads" one 16-byte DMA piece per lane (1 KiB per wave) + 8 ds_read_b32
                if ((q & 7) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + ((q >> 3) & 3) * 256), 16, vo, q * 128, 0, 0);
                nxt[q] = f32x4{lds[((q >> 3) & 3) * 256 + ((lane + 33 * q) & 255)], 0.f, 0.f, 0.f};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 32; ++r) MF(a + cur[r % NL].x, b, c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NL; ++q) cur[q] = nxt[q];
    }
    float s = acc;
    for (int j = 0; j < 16; ++j) s += c[j];
    if (s == 12345.f) out[0] = s;
}
template <int KIND, int NL>
void runm(float* o, const float* src, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4}) {
        const int blocks = 1024 * wps, iters = 512;
        hipLaunchKernelGGL((km<KIND, NL>), dim3(blocks), dim3(64), 0, 0, o, src, 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL((km<KIND, NL>), dim3(blocks), dim3(64), 0, 0, o, src, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9 / (iters * 32.0 * wps);
        printf("%-44s %d waves/SIMD: %7.1f us, %.1f cycles per MFMA, %.1f per memory instruction beyond 73.0\n", name, wps, ms * 1e3, cyc, (cyc - 73.0) * 32 / NL);
    }
}
int main(int argc, char** argv) {
    // sections: kinds (one memory instruction kind at a time, 8 / 16 per 32 MFMAs) | density (the P.V phase's load densities) |
    // valu (VALU instructions behind the MFMAs) | chains (dependent chains, loads and LDS traffic per wave); default: all
    const char* only = argc > 1 ? argv[1] : nullptr;
    auto want = [&](const char* n) { return !only || !strcmp(only, n); };
    float* o; hipMalloc(&o, 4);
    float* src; hipMalloc(&src, 64 * 64 * 16); hipMemset(src, 0, 64 * 64 * 16);
    if (want("density")) {
        runm<2, 32>(o, src, "32 global_load_dword / 32 MFMA");
        runm<3, 32>(o, src, "32 buffer_load_dword / 32 MFMA");
        runm<6, 16>(o, src, "16 buffer_load_dwordx2 / 32 MFMA");
        runm<1, 8>(o, src, "8 buffer_load_dwordx4 / 32 MFMA");
        runm<7, 32>(o, src, "4 DMA x4 + 32 ds_read_b32 / 32 MFMA");
        runm<3, 16>(o, src, "16 buffer_load_dword / 32 MFMA");
    }
    if (want("kinds")) {
        runm<0, 8>(o, src, "8 global_load_dwordx4 / 32 MFMA");
        runm<1, 8>(o, src, "8 buffer_load_dwordx4 / 32 MFMA");
        runm<2, 8>(o, src, "8 global_load_dword / 32 MFMA");
        runm<3, 8>(o, src, "8 buffer_load_dword / 32 MFMA");
        runm<4, 8>(o, src, "8 buffer_load_dwordx4 lds / 32 MFMA");
        runm<5, 8>(o, src, "8 ds_read_b128 / 32 MFMA");
        runm<0, 16>(o, src, "16 global_load_dwordx4 / 32 MFMA");
        runm<1, 16>(o, src, "16 buffer_load_dwordx4 / 32 MFMA");
        runm<5, 16>(o, src, "16 ds_read_b128 / 32 MFMA");
    }
    if (want("valu")) {
        runv<0, 0>(o, src, "32 MFMA");
        runv<64, 0>(o, src, "32 MFMA then 64 VALU");
        runv<256, 0>(o, src, "32 MFMA then 256 VALU");
        runv<0, 32>(o, src, "32 MFMA + 32 dword loads");
    }
    if (want("chains")) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        run<1, false, false>(o, s4, "1 chain");
        run<2, false, false>(o, s4, "2 chains");
        run<4, false, false>(o, s4, "4 chains");
        run<1, true, false>(o, s4, "1 chain + 8 loads / 32 MFMA");
        run<4, true, false>(o, s4, "4 chains + 8 loads / 32 MFMA");
        run<1, false, true>(o, s4, "1 chain + LDS rw / 2 MFMA");
        run<1, true, true>(o, s4, "1 chain + loads + LDS");
    }
    return 0;
}

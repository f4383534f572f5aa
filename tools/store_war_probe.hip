// Probe (tools only, round 6): how many wait states does gfx950 need between a 128-bit buffer store and a VALU instruction that
// OVERWRITES the store's data registers?
//
// Why: the transposed-accumulator epilogue of gemm_f32.hip (F32_TRANSPOSED) stores 16-byte pieces straight from registers and the
// next piece's first v_add lands in the first data register two wait states later (the compiler pads `s_nop 0` / `s_nop 1`:
// LLVM's VmemStoreHazard for stores of more than 64 bits).  Round 5 saw "a few hundred wrong elements per launch, the FIRST
// element of a piece, a few neighbouring rows" and found no cause.  The first element is exactly the register that is overwritten
// first.  This probe issues  buffer_store_dwordx4 v[20:23]  followed by K wait states and  v_mov_b32 v20..v23, <poison>  from one
// wave per SIMD, alone and with three sibling waves per SIMD issuing v_mfma_f32_32x32x2_f32 back to back (the GEMM's situation:
// an epilogue wave shares its SIMD with three waves in their K loops), and counts the stored dwords that hold the poison.
//
//   hipcc --offload-arch=gfx950 -O2 tools/store_war_probe.hip -o tools/bin/store_war_probe && tools/bin/store_war_probe
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PROBE_BODY(NOPS) PROBE_BODY_(NOPS, "0")
#define PROBE_BODY_S(NOPS) PROBE_BODY_(NOPS, "s20")
#define PROBE_BODY_(NOPS, SOFF)                                                              \
    asm volatile(                                                                            \
        "s_mov_b32 s20, 0\n"                                                                 \
        "v_mov_b32 v20, %1\n"                                                                \
        "v_add_u32 v21, 1, %1\n"                                                             \
        "v_add_u32 v22, 2, %1\n"                                                             \
        "v_add_u32 v23, 3, %1\n"                                                             \
        "s_nop 7\n"                                                                          \
        "buffer_store_dwordx4 v[20:23], %0, %2, " SOFF " offen\n"                            \
        NOPS                                                                                 \
        "v_mov_b32 v20, %3\n"                                                                \
        "v_mov_b32 v21, %3\n"                                                                \
        "v_mov_b32 v22, %3\n"                                                                \
        "v_mov_b32 v23, %3\n"                                                                \
        "s_nop 7\n"                                                                          \
        :                                                                                    \
        : "v"(off), "v"(tag), "s"(rs), "v"(poison)                                           \
        : "v20", "v21", "v22", "v23", "s20", "memory");

// the form both failing builds of gemm_f32.hip contain: the soffset SGPR is an SGPR-spill reload -- written by v_readlane_b32, the
// compiler's 5 wait states (s_nop 4), the store, the overwrite
#define PROBE_BODY_R(NOPS)                                                                   \
    asm volatile(                                                                            \
        "v_mov_b32 v24, 0\n"                                                                 \
        "v_mov_b32 v20, %1\n"                                                                \
        "v_add_u32 v21, 1, %1\n"                                                             \
        "v_add_u32 v22, 2, %1\n"                                                             \
        "v_add_u32 v23, 3, %1\n"                                                             \
        "s_nop 7\n"                                                                          \
        "v_readlane_b32 s20, v24, 17\n"                                                      \
        "s_nop 4\n"                                                                          \
        "buffer_store_dwordx4 v[20:23], %0, %2, s20 offen\n"                                 \
        NOPS                                                                                 \
        "v_max_f32_e32 v20, %3, %3\n"                                                        \
        "v_mov_b32 v21, %3\n"                                                                \
        "v_mov_b32 v22, %3\n"                                                                \
        "v_mov_b32 v23, %3\n"                                                                \
        "s_nop 7\n"                                                                          \
        :                                                                                    \
        : "v"(off), "v"(tag), "s"(rs), "v"(poison)                                           \
        : "v20", "v21", "v22", "v23", "v24", "s20", "memory");

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int K>
__device__ __forceinline__ void store_then_overwrite_readlane(int off, unsigned tag, i32x4 rs, unsigned poison) {
    if constexpr (K == 0) { PROBE_BODY_R("") }
    else if constexpr (K == 1) { PROBE_BODY_R("s_nop 0\n") }
    else if constexpr (K == 2) { PROBE_BODY_R("s_nop 1\n") }
    else { PROBE_BODY_R("s_nop 3\n") }
}

template <int K>
__device__ __forceinline__ void store_then_overwrite_sreg(int off, unsigned tag, i32x4 rs, unsigned poison) {
    // the same with the store's soffset in an SGPR -- the form LLVM's hazard table exempts (createsVALUHazard: "this hazard only
    // exists if the instruction is not using a register in the soffset field") and therefore never pads
    if constexpr (K == 0) { PROBE_BODY_S("") }
    else if constexpr (K == 1) { PROBE_BODY_S("s_nop 0\n") }
    else if constexpr (K == 2) { PROBE_BODY_S("s_nop 1\n") }
    else { PROBE_BODY_S("s_nop 3\n") }
}

template <int K>
__device__ __forceinline__ void store_then_overwrite(int off, unsigned tag, i32x4 rs, unsigned poison) {
    if constexpr (K == 0) { PROBE_BODY("") }
    else if constexpr (K == 1) { PROBE_BODY("s_nop 0\n") }
    else if constexpr (K == 2) { PROBE_BODY("s_nop 1\n") }
    else if constexpr (K == 3) { PROBE_BODY("s_nop 2\n") }
    else if constexpr (K == 4) { PROBE_BODY("s_nop 3\n") }
    else if constexpr (K == 6) { PROBE_BODY("s_nop 5\n") }
    else if constexpr (K == 8) { PROBE_BODY("s_nop 7\n") }
    else if constexpr (K == 16) { PROBE_BODY("s_nop 7\ns_nop 7\n") }
    else { PROBE_BODY("s_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\n") }
}

// 16 waves per block, one block per CU: waves 0..3 (one per SIMD) run the store test, waves 4..15 (three per SIMD) the MFMA stream
template <int K, int SREG = 0>
__global__ __launch_bounds__(1024) void probe_kernel(unsigned* out, int iters, int with_mfma, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4) {
        if (!with_mfma) return;
        f32x16 c0 = {0}, c1 = {0};
        float a = lane * 0.001f, b = 0.5f;
        for (int it = 0; it < iters * 6; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
            }
        }
        float s = 0;
        for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
        if (s == 12345.f) sink[0] = s;
        return;
    }
    // test wave: its own region of `out`: [block][wave][iter][lane][4]
    const size_t per_wave = (size_t)iters * 64 * 4;
    unsigned* base = out + ((size_t)blockIdx.x * 4 + wave) * per_wave;
    // raw buffer descriptor: base, stride 0, num_records = bytes, the flags gemm_f32.hip uses
    const unsigned long long bp = (unsigned long long)base;
    i32x4 rs = {(int)(unsigned)bp, (int)(unsigned)((bp >> 32) & 0xFFFF), (int)(per_wave * 4), 0x00020000};
    rs[0] = __builtin_amdgcn_readfirstlane(rs[0]); rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);
    rs[2] = __builtin_amdgcn_readfirstlane(rs[2]); rs[3] = __builtin_amdgcn_readfirstlane(rs[3]);
    for (int it = 0; it < iters; ++it) {
        const int off = (it * 64 + lane) * 16;
        const unsigned tag = (unsigned)(it * 64 + lane) * 4u + 0x100u;
        if constexpr (SREG == 2) store_then_overwrite_readlane<K>(off, tag, rs, 0xDEADBEEFu);
        else if constexpr (SREG == 1) store_then_overwrite_sreg<K>(off, tag, rs, 0xDEADBEEFu);
        else store_then_overwrite<K>(off, tag, rs, 0xDEADBEEFu);
    }
}

// ---- scenario 2 (the GEMM epilogue's shape): FOUR 16-byte pieces in a row from the SAME data registers -- piece p + 1's values are
// written K wait states behind piece p's store -- while the sibling waves of the SIMD run MFMAs AND keep the vector-memory and LDS
// pipes busy (buffer loads, ds reads: an epilogue wave's stores queue up behind the K loops' operand traffic).  A stored dword
// that holds the NEXT piece's value is the hazard.
#define BURST_PIECE(P, NOPS)                                                                 \
        "v_add_u32 v20, " #P "*4+0, %1\n"                                                    \
        "v_add_u32 v21, " #P "*4+1, %1\n"                                                    \
        "v_add_u32 v22, " #P "*4+2, %1\n"                                                    \
        "v_add_u32 v23, " #P "*4+3, %1\n"                                                    \
        "buffer_store_dwordx4 v[20:23], %0, %2, 0 offen offset:" #P "*16\n"                  \
        NOPS
// (the epilogue's own form: the piece's offset in an SGPR, the values formed by two-source VALU instructions)
#define BURST_PIECE_S(P, NOPS)                                                               \
        "s_movk_i32 s20, " #P "*16\n"                                                        \
        "v_add_u32 v20, %2, %0\n"                                                            \
        "v_add_u32 v21, 1, v20\n"                                                            \
        "v_add_u32 v22, 2, v20\n"                                                            \
        "v_add_u32 v23, 3, v20\n"                                                            \
        "v_add_u32 %0, 4, %0\n"                                                              \
        "buffer_store_dwordx4 v[20:23], %1, %3, s20 offen\n"                                 \
        NOPS
#define BURST_BODY_S(NOPS)                                                                   \
    {                                                                                        \
        unsigned step_ = 0;                                                                  \
        asm volatile(BURST_PIECE_S(0, NOPS) BURST_PIECE_S(1, NOPS) BURST_PIECE_S(2, NOPS) BURST_PIECE_S(3, NOPS) \
            "v_add_u32 v20, %2, %0\n" "s_nop 7\n"                                            \
            : "+v"(step_) : "v"(off), "v"(tag), "s"(rs), "v"(poison) : "v20", "v21", "v22", "v23", "s20", "memory"); \
    }
#define BURST_BODY(NOPS)                                                                     \
    asm volatile(BURST_PIECE(0, NOPS) BURST_PIECE(1, NOPS) BURST_PIECE(2, NOPS) BURST_PIECE(3, NOPS) \
        "v_mov_b32 v20, %3\n" "s_nop 7\n"                                                    \
        : : "v"(off), "v"(tag), "s"(rs), "v"(poison) : "v20", "v21", "v22", "v23", "memory");

template <int K>
__device__ __forceinline__ void burst_sreg(int off, unsigned tag, i32x4 rs, unsigned poison) {
    if constexpr (K == 0) { BURST_BODY_S("") }
    else if constexpr (K == 1) { BURST_BODY_S("s_nop 0\n") }
    else if constexpr (K == 2) { BURST_BODY_S("s_nop 1\n") }
    else { BURST_BODY_S("s_nop 7\n") }
}

template <int K>
__device__ __forceinline__ void burst(int off, unsigned tag, i32x4 rs, unsigned poison) {
    if constexpr (K == 0) { BURST_BODY("") }
    else if constexpr (K == 1) { BURST_BODY("s_nop 0\n") }
    else if constexpr (K == 2) { BURST_BODY("s_nop 1\n") }
    else if constexpr (K == 3) { BURST_BODY("s_nop 2\n") }
    else if constexpr (K == 4) { BURST_BODY("s_nop 3\n") }
    else if constexpr (K == 6) { BURST_BODY("s_nop 5\n") }
    else if constexpr (K == 8) { BURST_BODY("s_nop 7\n") }
    else if constexpr (K == 16) { BURST_BODY("s_nop 7\ns_nop 7\n") }
    else { BURST_BODY("s_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\n") }
}

template <int K, bool SREG = false>
__global__ __launch_bounds__(1024) void burst_kernel(unsigned* out, int iters, int siblings, float* sink, const float* src) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4) {
        if (!siblings) return;
        f32x16 c0 = {0}, c1 = {0};
        float a = lane * 0.001f, b = 0.5f;
        float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = threadIdx.x; i < 4096; i += 1024) lds[i] = (float)i;
        for (int it = 0; it < iters * 8; ++it) {
            if (siblings >= 2) {   // the K loop's operand traffic: 16-byte global loads and LDS reads between the MFMAs
                const float4 g = *reinterpret_cast<const float4*>(src + ((size_t)((it * 1024 + threadIdx.x) & 0xFFFFF)) * 4);
                const float4 l4 = *reinterpret_cast<const float4*>(lds + ((it * 64 + lane) & 1023) * 4);
                acc4.x += g.x + l4.x; acc4.y += g.y + l4.y; acc4.z += g.z; acc4.w += g.w;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
            }
        }
        float s = acc4.x + acc4.y + acc4.z + acc4.w;
        for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
        if (s == 12345.f) sink[0] = s;
        return;
    }
    // test wave: [block][wave][iter][lane][4 pieces][4 dwords]; lane's 64 bytes are contiguous (16-byte pieces, 32-byte... whole lines)
    const size_t per_wave = (size_t)iters * 64 * 16;
    unsigned* base = out + ((size_t)blockIdx.x * 4 + wave) * per_wave;
    const unsigned long long bp = (unsigned long long)base;
    i32x4 rs = {(int)(unsigned)bp, (int)(unsigned)((bp >> 32) & 0xFFFF), (int)(per_wave * 4), 0x00020000};
    rs[0] = __builtin_amdgcn_readfirstlane(rs[0]); rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);
    rs[2] = __builtin_amdgcn_readfirstlane(rs[2]); rs[3] = __builtin_amdgcn_readfirstlane(rs[3]);
    for (int it = 0; it < iters; ++it) {
        const int off = (it * 64 + lane) * 64;
        const unsigned tag = (unsigned)(it * 64 + lane) * 16u + 0x100u;
        if constexpr (SREG) burst_sreg<K>(off, tag, rs, 0xDEADBEEFu);
        else burst<K>(off, tag, rs, 0xDEADBEEFu);
    }
}

template <int K, bool SREG = false>
static void run_burst(unsigned* dev, float* sink, const float* src, int blocks, int iters, int siblings) {
    const size_t n = (size_t)blocks * 4 * iters * 64 * 16;
    hipMemset(dev, 0, n * 4);
    hipLaunchKernelGGL((burst_kernel<K, SREG>), dim3(blocks), dim3(1024), 0, 0, dev, iters, siblings, sink, src);
    hipDeviceSynchronize();
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), dev, n * 4, hipMemcpyDeviceToHost);
    size_t next[4] = {0, 0, 0, 0}, other = 0;   // dwords of a piece holding the NEXT piece's value, by position in the piece
    for (size_t w = 0; w < (size_t)blocks * 4; ++w)
        for (size_t i = 0; i < (size_t)iters * 64; ++i)
            for (int e = 0; e < 16; ++e) {
                const unsigned v = h[(w * iters * 64 + i) * 16 + e], want = (unsigned)i * 16u + 0x100u + e;
                if (v == want) continue;
                if (v == want + 4 || v == 0xDEADBEEFu) ++next[e & 3];
                else ++other;
            }
    const char* who = siblings == 0 ? "test waves alone                          " : (siblings == 1 ? "3 MFMA waves beside each test wave        " : "3 MFMA + load + LDS waves beside each one ");
    printf("  %2d wait states, soffset %s, %s: dwords holding the NEXT piece's value, by position %zu %zu %zu %zu of %zu pieces (other mismatches %zu)\n", K, SREG ? "in an SGPR " : "immediate  ", who,
           next[0], next[1], next[2], next[3], (size_t)blocks * 4 * iters * 64 * 4, other);
}

template <int K, int SREG = 0>
static void run(unsigned* dev, float* sink, int blocks, int iters, int with_mfma) {
    const size_t n = (size_t)blocks * 4 * iters * 64 * 4;
    hipMemset(dev, 0, n * 4);
    hipLaunchKernelGGL((probe_kernel<K, SREG>), dim3(blocks), dim3(1024), 0, 0, dev, iters, with_mfma, sink);
    hipDeviceSynchronize();
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), dev, n * 4, hipMemcpyDeviceToHost);
    size_t bad[4] = {0, 0, 0, 0}, other = 0;
    for (size_t w = 0; w < (size_t)blocks * 4; ++w)
        for (size_t i = 0; i < (size_t)iters * 64; ++i)
            for (int e = 0; e < 4; ++e) {
                const unsigned v = h[(w * iters * 64 + i) * 4 + e], want = (unsigned)i * 4u + 0x100u + e;
                if (v == 0xDEADBEEFu) ++bad[e];
                else if (v != want) ++other;
            }
    printf("  %2d wait states, soffset %s, %s: poisoned dwords by position %zu %zu %zu %zu of %zu pieces (other mismatches %zu)\n", K,
           SREG == 2 ? "SGPR<-readlane" : (SREG ? "in an SGPR " : "immediate 0"), with_mfma ? "3 MFMA waves beside each test wave" : "test waves alone               ", bad[0], bad[1], bad[2], bad[3],
           (size_t)blocks * 4 * iters * 64, other);
}

int main() {
    const int blocks = 256, iters = 512;
    unsigned* dev; float* sink;
    hipMalloc(&dev, (size_t)blocks * 4 * iters * 64 * 16);
    hipMalloc(&sink, 64);
    printf("buffer_store_dwordx4 v[20:23] ; K wait states ; v_mov_b32 v20..v23, poison   (gfx950, one test wave per SIMD, %d CUs x 4 x %d stores of 64 lanes)\n", blocks, iters);
    for (int with_mfma = 0; with_mfma < 2; ++with_mfma) {
        run<0>(dev, sink, blocks, iters, with_mfma);
        run<1>(dev, sink, blocks, iters, with_mfma);
        run<2>(dev, sink, blocks, iters, with_mfma);
        run<3>(dev, sink, blocks, iters, with_mfma);
        run<4>(dev, sink, blocks, iters, with_mfma);
        run<6>(dev, sink, blocks, iters, with_mfma);
        run<8>(dev, sink, blocks, iters, with_mfma);
        run<16>(dev, sink, blocks, iters, with_mfma);
        run<32>(dev, sink, blocks, iters, with_mfma);
        run<0, 1>(dev, sink, blocks, iters, with_mfma);
        run<1, 1>(dev, sink, blocks, iters, with_mfma);
        run<2, 1>(dev, sink, blocks, iters, with_mfma);
        run<4, 1>(dev, sink, blocks, iters, with_mfma);
        run<0, 2>(dev, sink, blocks, iters, with_mfma);
        run<1, 2>(dev, sink, blocks, iters, with_mfma);
        run<2, 2>(dev, sink, blocks, iters, with_mfma);
        run<4, 2>(dev, sink, blocks, iters, with_mfma);
    }
    printf("four pieces in a row from the same registers: { v_add_u32 v20..v23 ; buffer_store_dwordx4 v[20:23] ; K wait states } x 4\n");
    float* src; hipMalloc(&src, (size_t)(1 << 20) * 16 + 64); hipMemset(src, 0, (size_t)(1 << 20) * 16 + 64);
    for (int siblings = 0; siblings < 3; ++siblings) {
        run_burst<0>(dev, sink, src, blocks, iters / 4, siblings);
        run_burst<1>(dev, sink, src, blocks, iters / 4, siblings);
        run_burst<2>(dev, sink, src, blocks, iters / 4, siblings);
        run_burst<3>(dev, sink, src, blocks, iters / 4, siblings);
        run_burst<4>(dev, sink, src, blocks, iters / 4, siblings);
        run_burst<6>(dev, sink, src, blocks, iters / 4, siblings);
        run_burst<8>(dev, sink, src, blocks, iters / 4, siblings);
        run_burst<16>(dev, sink, src, blocks, iters / 4, siblings);
        run_burst<0, true>(dev, sink, src, blocks, iters / 4, siblings);
        run_burst<1, true>(dev, sink, src, blocks, iters / 4, siblings);
        run_burst<2, true>(dev, sink, src, blocks, iters / 4, siblings);
        run_burst<8, true>(dev, sink, src, blocks, iters / 4, siblings);
    }
    return 0;
}

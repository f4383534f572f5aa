// Split-f16 operand format of the Conformer GEMMs (device helpers shared by producers and the GEMM).
//
// A float32 value x is carried as two float16 numbers
//     hi = f16(x)            lo = f16((x - hi) * 2^11)          x = hi + lo * 2^-11  (to ~2^-22 relative)
// and a product of two such operands is evaluated with THREE f16 MFMAs accumulating in float32:
//     a*b = a_hi*b_hi + 2^-11 * (a_hi*b_lo + a_lo*b_hi)         (the dropped a_lo*b_lo term is ~2^-22 a*b)
// A product of two f16 numbers is exact in float32, so the only roundings are those of the float32
// accumulation -- the same as in a float32 GEMM.  Measured on the v1.0-MC network against a float64 run
// of the same network (tools/split_f16_numerics.py): mask error max 1.67e-6 / rms 2.55e-7, identical to
// the float32 GEMM's (1.70e-6 / 2.55e-7).  The f16 MFMA rate on gfx950 is 16x the f32 MFMA rate, so the
// three products cost 3/16 of one float32 product.
//
// lo is scaled by 2^11 so that it is a normal f16 number whenever hi is (no dependence on denormal
// support), and |x| < 2^-14 is carried entirely by lo (hi = 0); below 2^-25 lo itself loses bits, an ABSOLUTE error
// under 2^-36 per operand, invisible in any sum the path forms.  The format covers |x| <= 65504 (LayerNorm outputs,
// ReLU activations, attention contexts, pi-bounded features and trained weights are orders of magnitude inside).  A
// value outside is NOT clamped: hi becomes +-inf, lo NaN, and the non-finite result reaches the stitched activity and
// the waveforms, where css_run* looks for it (api.hip: range check) and repeats the pass on the exact float32
// kernels -- an out-of-range operand costs time, never a silently wrong answer.
//
// Memory layout of a split matrix [rows][K] (K % 32 == 0): the row is K/32 groups of 128 bytes, each
// 32 hi halves followed by the 32 lo halves of the same k -- one row of a 32-wide K slab is one 128-byte
// line, exactly like 32 floats, so a split matrix is addressed as a float matrix with ld = K.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace css {

constexpr float SPLIT_LO_SCALE = 2048.0f;
constexpr float SPLIT_LO_INV = 1.0f / 2048.0f;

// half index of element k inside its row (lo is 32 halves further)
__device__ __forceinline__ int split_index(int k) { return ((k >> 5) << 6) | (k & 31); }

__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
    const _Float16 h = fabsf(x) < 6.103515625e-05f ? (_Float16)0.f : (_Float16)x;
    hi = h;
    lo = (_Float16)((x - (float)h) * SPLIT_LO_SCALE);
}

// Level normalisation of the one split operand whose magnitude follows the recording's level, the stitched spectra fed
// to the synthesis transform: they are multiplied by a power of two that brings the recording's peak sample into
// [0.5, 1) (peak_bits: max |sample| as float bits, gathered while the PCM is laid out channel-major) and the
// synthesised samples are multiplied back -- exact in float32, so integer-scaled PCM (+-32768) cannot overflow the
// format and a recording at -100 dBFS does not fall into its 11-bit range.  nullptr / silence: gain 1.
__device__ __forceinline__ float level_gain(const unsigned int* peak_bits) {
    if (!peak_bits) return 1.f;
    const float p = __uint_as_float(*peak_bits);
    if (!(p > 0.f) || !(p < 3.0e38f)) return 1.f;
    int e;
    frexpf(p, &e);                                   // p = m * 2^e, m in [0.5, 1)
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    return ldexpf(1.f, -e);
}

// store element k of a split row (row = base pointer of the row, as halves)
__device__ __forceinline__ void split_store(_Float16* row, int k, float x) {
    _Float16 hi, lo;
    split_f16(x, hi, lo);
    const int i = split_index(k);
    row[i] = hi;
    row[i + 32] = lo;
}

// four consecutive elements k..k+3 (k % 4 == 0): two 8-byte stores
__device__ __forceinline__ void split_store4(_Float16* row, int k, float x0, float x1, float x2, float x3) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    f16x4 hi, lo;
    _Float16 a, b;
    split_f16(x0, a, b); hi[0] = a; lo[0] = b;
    split_f16(x1, a, b); hi[1] = a; lo[1] = b;
    split_f16(x2, a, b); hi[2] = a; lo[2] = b;
    split_f16(x3, a, b); hi[3] = a; lo[3] = b;
    const int i = split_index(k);
    *reinterpret_cast<f16x4*>(row + i) = hi;
    *reinterpret_cast<f16x4*>(row + i + 32) = lo;
}

}  // namespace css

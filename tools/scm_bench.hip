// Stand-alone timing of css::launch_scm (and the other MVDR-stage kernels) on the 60 s meeting's shapes (tools only).
//   (build line: tools/scm_mfma.hip; it also holds the float64-MFMA form of the covariance kernel this compares)
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>
#include "kernels.hpp"
using namespace css;
namespace css { bool launch_scm_mfma(const MvdrArgs& a, hipStream_t s); }
int main() {
    const int C = 7, F = 257, T = 186, hop = 93, S = 3, nseg = 40;
    const int64_t TL = 3749, T_ld = 3752, mask_ld = (int64_t)nseg * T;
    std::vector<float> hx((size_t)C * 2 * F * T_ld), hm((size_t)(S + 1) * F * mask_ld);
    unsigned s = 1;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    for (auto& v : hm) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f; }
    float *X, *M, *sep; double *scm, *bfw;
    hipMalloc(&X, hx.size() * 4); hipMalloc(&M, hm.size() * 4);
    hipMalloc(&scm, (size_t)nseg * (S + 1) * F * 49 * 8); hipMalloc(&bfw, (size_t)nseg * S * F * 14 * 8);
    hipMalloc(&sep, (size_t)nseg * S * F * T * 8);
    hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(M, hm.data(), hm.size() * 4, hipMemcpyHostToDevice);
    MvdrArgs a{};
    a.X = X; a.T_ld = T_ld; a.stft_frames = TL; a.C = C; a.F = F; a.masks = M; a.mask_ld = mask_ld; a.S = S; a.T = T; a.hop = hop;
    a.seg_lo = 0; a.nseg = nseg; a.scm = scm; a.bfw = bfw; a.sep = sep; a.mask_floor = 1.f; a.use_mvdr = 1;
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto fn) {
        for (int i = 0; i < 3; ++i) fn();
        hipEventRecord(e0, st);
        for (int i = 0; i < 20; ++i) fn();
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-12s %8.2f us\n", name, ms * 1000 / 20);
    };
    // launch_scm is the product (vector-ALU) kernel, launch_scm_mfma the matrix-core experiment of tools/scm_mfma.hip
    const size_t nscm = (size_t)nseg * (S + 1) * F * 49;
    std::vector<double> ref(nscm), got(nscm);
    launch_scm(a, st); hipStreamSynchronize(st);
    hipMemcpy(ref.data(), scm, nscm * 8, hipMemcpyDeviceToHost);
    hipMemset(scm, 0, nscm * 8);
    launch_scm_mfma(a, st); hipStreamSynchronize(st);
    hipMemcpy(got.data(), scm, nscm * 8, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (size_t i = 0; i < nscm; ++i) { scale = fmax(scale, fabs(ref[i])); }
    for (size_t i = 0; i < nscm; ++i) { worst = fmax(worst, fabs(ref[i] - got[i])); }
    {   // where the two differ: by packed index and by mask
        int bad_i[49] = {0}, bad_k[4] = {0}; size_t nbad = 0;
        for (size_t i = 0; i < nscm; ++i) if (fabs(ref[i] - got[i]) > 1e-9 * scale) { ++bad_i[i % 49]; ++bad_k[(i / 49 / F) % (S + 1)]; ++nbad; }
        printf("differing entries %zu of %zu; by mask:", nbad, nscm);
        for (int k = 0; k < 4; ++k) printf(" %d", bad_k[k]);
        printf("\nby packed index:");
        for (int i = 0; i < 49; ++i) printf(" %d", bad_i[i]);
        printf("\nfirst matrix (seg 0, mask 0, bin 0): vector / matrix\n");
        for (int i = 0; i < 49; ++i) printf("  [%2d] % .12g  % .12g\n", i, ref[i], got[i]);
    }
    printf("scm: max |vector - matrix| = %.3g (max |value| %.3g) err %s\n", worst, scale, hipGetErrorString(hipGetLastError()));
    timeit("scm", [&] { launch_scm(a, st); });
    timeit("scm_mfma", [&] { launch_scm_mfma(a, st); });
    timeit("scm", [&] { launch_scm(a, st); });
    timeit("scm_mfma", [&] { launch_scm_mfma(a, st); });
    timeit("mvdr_solve", [&] { launch_mvdr_solve(a, st); });
    timeit("beamform", [&] { launch_beamform(a, st); });
    std::vector<double> out(49 * 4);
    hipMemcpy(out.data(), scm, out.size() * 8, hipMemcpyDeviceToHost);
    printf("checksum %.10g %.10g\n", out[0], out[48]);
    return 0;
}

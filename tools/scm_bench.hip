// Stand-alone timing of css::launch_scm (and the other MVDR-stage kernels) on the 60 s meeting's shapes (tools only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/scm_bench.hip notsofar1-challenge_amd/csrc/mvdr.hip -Inotsofar1-challenge_amd/csrc -o /tmp/scm_bench
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>
#include "kernels.hpp"
using namespace css;
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
    timeit("scm", [&] { launch_scm(a, st); });
    timeit("mvdr_solve", [&] { launch_mvdr_solve(a, st); });
    timeit("beamform", [&] { launch_beamform(a, st); });
    std::vector<double> out(49 * 4);
    hipMemcpy(out.data(), scm, out.size() * 8, hipMemcpyDeviceToHost);
    printf("checksum %.10g %.10g\n", out[0], out[48]);
    return 0;
}

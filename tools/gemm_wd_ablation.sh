# K-loop ablations of the weights-direct split GEMM on one long-K shape (M = 7440, N = 512, K = 4096), both tile heights
S="tools/gemm_bench.hip notsofar1-challenge_amd/csrc/gemm.hip notsofar1-challenge_amd/csrc/gemm_split.hip notsofar1-challenge_amd/csrc/gemm_split_wd.hip -Inotsofar1-challenge_amd/csrc"
export GEMM_BENCH_ONLY="K=4096"
for v in ${ABL_LIST:-"" "-DCSS_ABL_MFMA_THIRD" "-DCSS_ABL_NO_BARRIER" "-DCSS_ABL_NO_AREAD" "-DCSS_ABL_HALF_AREAD" "-DCSS_ABL_ASAME -DCSS_ABL_WSAME"}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 $v $S -o /tmp/gb 2>/dev/null
  for w in 64 8; do
    echo "[$v] tile $w: $(CSS_GEMM_WD_WAVES=$w /tmp/gb | grep W-direct | cut -c26-60)"
  done
done

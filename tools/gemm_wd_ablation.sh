S="tools/gemm_bench.hip notsofar1-challenge_amd/csrc/gemm.hip notsofar1-challenge_amd/csrc/gemm_split.hip notsofar1-challenge_amd/csrc/gemm_split_wd.hip -Inotsofar1-challenge_amd/csrc"
export GEMM_BENCH_ONLY="K=4096"
for v in "" "-DCSS_ABL_ASAME" "-DCSS_ABL_WSAME" "-DCSS_ABL_ASAME -DCSS_ABL_WSAME"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 $v $S -o /tmp/gb 2>/dev/null
  echo "[$v] $(/tmp/gb | grep W-direct | cut -c1-60)"
done

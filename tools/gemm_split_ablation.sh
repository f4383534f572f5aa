S="tools/gemm_bench.hip notsofar1-challenge_amd/csrc/gemm.hip notsofar1-challenge_amd/csrc/gemm_split.hip notsofar1-challenge_amd/csrc/gemm_split_wd.hip -Inotsofar1-challenge_amd/csrc"
for v in "" "-DCSS_ABL_NO_GLOAD" "-DCSS_ABL_NO_LSTORE" "-DCSS_ABL_NO_BARRIER" "-DCSS_ABL_NO_MFMA" "-DCSS_ABL_NO_GLOAD -DCSS_ABL_NO_LSTORE" "-DCSS_ABL_NO_GLOAD -DCSS_ABL_NO_LSTORE -DCSS_ABL_NO_BARRIER" "-DCSS_ABL_NO_GLOAD -DCSS_ABL_NO_LSTORE -DCSS_ABL_NO_MFMA" "-DCSS_ABL_NO_LSTORE -DCSS_ABL_NO_BARRIER -DCSS_ABL_NO_MFMA"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 $v $S -o /tmp/gb 2>/dev/null
  for l in 8 4; do echo "[$v] layout $l: $(CSS_GEMM_SPLIT_LAYOUT=$l /tmp/gb | grep -A2 '^K=4096' | grep split | cut -c1-70)"; done
done

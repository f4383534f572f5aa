# Attribute the per-launch fixed cost of gemm_split_wd: full kernel vs no epilogue at all (GPU box, through gpurun)
S="tools/gemm_bench.hip notsofar1-challenge_amd/csrc/gemm.hip notsofar1-challenge_amd/csrc/gemm_split.hip notsofar1-challenge_amd/csrc/gemm_split_wd.hip -Inotsofar1-challenge_amd/csrc"
for v in "" "-DCSS_ABL_NO_EMIT"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 $v $S -o /tmp/gb 2>/dev/null
  for w in 8 64; do for nt in 0 1; do echo "[$v] wd $w nt $nt: $(CSS_EPI_NT=$nt CSS_GEMM_WD_WAVES=$w /tmp/gb | awk '/^[A-Za-z0-9=]/ && !/split/ {name=$1" "$2} /W-direct/ {if (name ~ /^K=32 M|^K=32 \+res|^wo 40|^wo \+res|^ffn1|^ffn2 \+res/) printf "%s ", $3}')"; done; done
done
echo "(columns: K=32, K=32 +res, wo, wo +res, ffn1, ffn2 +res)"

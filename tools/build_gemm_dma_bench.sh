#!/bin/bash
# builds tools/bin/gemm_dma_bench (cross-compiles here; the binary travels to the GPU box with the tree)
cd "$(dirname "$0")/.." && mkdir -p tools/bin && \
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCSS_GEMM_DMA_ABLATE -Wno-inline-asm tools/gemm_dma_bench.hip notsofar1-challenge_amd/csrc/gemm.hip \
  notsofar1-challenge_amd/csrc/gemm_split.hip notsofar1-challenge_amd/csrc/gemm_split_wd.hip tools/gemm_split_dma.hip \
  -Inotsofar1-challenge_amd/csrc -Itools -o tools/bin/gemm_dma_bench 2>&1 | grep -E "error|scratch" ; ls -la tools/bin/gemm_dma_bench

// Declarations of tools/gemm_split_dma.hip: the LDS-DMA / specialised-wave / persistent GEMM kernels of round 3.  They are
// bit-identical to the product's weights-direct kernel (gemm_split_wd.hip) and tie or lose inside the mask estimator
// (profiles/r03_gemm_ws_in_situ.txt), so round 4 took them out of libcss_mi355.so; they stay here with
// their bench (tools/gemm_dma_bench.hip) as the record of the experiment.
#pragma once
#include "kernels.hpp"
namespace css {
void launch_gemm_split_dma(const GemmArgs& g, hipStream_t s);   // g.tile_rows: 3 specialised waves, 33 persistent, 24 plain DMA
bool gemm_split_ws_eligible(const GemmArgs& g);
bool gemm_split_wsp_eligible(const GemmArgs& g);
}  // namespace css

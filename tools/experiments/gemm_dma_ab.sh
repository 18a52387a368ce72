#!/bin/bash
# A/B of the k_gemm staging: registers (default) vs LDS-DMA (SF_GEMM_DMA=1).  usage (GPU box, repo root): bash tools/experiments/gemm_dma_ab.sh
P=sparse-matrix-factorization-library_amd
hipcc -O3 --offload-arch=gfx950 -I$P/csrc -Iinclude tools/gemm_bench.hip -L$P -lsparseframe_hip -Wl,-rpath,$PWD/$P -o /tmp/gemm_bench 2>/dev/null
for dma in 0 1; do
  echo "== SF_GEMM_DMA=$dma"
  SF_GEMM_DMA=$dma /tmp/gemm_bench 16384 16384 4096 4
  SF_GEMM_DMA=$dma /tmp/gemm_bench 16384 4096 8192 4
  SF_GEMM_DMA=$dma /tmp/gemm_bench 24576 512 8192 4
  SF_GEMM_DMA=$dma /tmp/gemm_bench 16384 16384 512 6 1
  SF_GEMM_DMA=$dma /tmp/gemm_bench 16384 16384 100 6 1
done

#!/bin/bash
# usage (on the GPU box, from the repo root): tools/run_gemm_bench.sh
set -e
P=sparse-matrix-factorization-library_amd
hipcc -O3 --offload-arch=gfx950 -I$P/csrc -Iinclude tools/gemm_bench.hip -L$P -lsparseframe_hip -Wl,-rpath,$PWD/$P -o /tmp/gemm_bench 2>/dev/null
/tmp/gemm_bench 16384 16384 4096 4
/tmp/gemm_bench 16384 4096 8192
/tmp/gemm_bench 24576 512 8192
/tmp/gemm_bench 8192 8192 64 20
/tmp/gemm_bench 2048 2048 1024 20
echo "--- plain vs atomic epilogue at small K"
for k in 128 256 512 1024; do /tmp/gemm_bench 16384 16384 $k 6 0; /tmp/gemm_bench 16384 16384 $k 6 1; done

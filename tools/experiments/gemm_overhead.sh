#!/bin/bash
# Per-tile fixed cost of k_gemm: K sweep on one 16384 x 16384 lower trapezoid (8256 tiles = 16 rounds + remainder), with the
# library kernel and with compile-time ablations (the kernels source compiled straight into the bench).
#   bash tools/experiments/gemm_overhead.sh  ->  gpurun_out/gemm_overhead.txt
set -e
cd "$(dirname "$0")/../.."
C=sparse-matrix-factorization-library_amd/csrc
out=gpurun_out/gemm_overhead.txt; : > $out
for v in ${VARIANTS:-"" "-DSF_EXP_TIMING" "-DSF_EXP_SKIP_EPILOGUE"}; do
  v=${v//,/ }          # a variant may carry several flags, comma-separated
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -I$C -Iinclude $v tools/gemm_bench.hip $C/sf_kernels.hip -o /tmp/gemm_bench_v
  echo "== variant '$v'" >> $out
  for K in ${KS:-128 256 512 1024 2048 4096}; do
    /tmp/gemm_bench_v 16384 16384 $K 4 1 >> $out
  done
done
cat $out

#!/bin/bash
# What the device-scope release / acquire of the fused step's hand-off cost (timing ablations, results are not checked):
#   bash tools/experiments/step_fence.sh  ->  gpurun_out/step_fence.txt
cd "$(dirname "$0")/../.."
C=sparse-matrix-factorization-library_amd/csrc
out=gpurun_out/step_fence.txt; : > $out
for v in "" "-DSF_EXP_NO_RELEASE_FENCE" "-DSF_EXP_NO_RELEASE_FENCE,-DSF_EXP_NO_ACQUIRE_FENCE"; do
  vv=${v//,/ }
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -I$C -Iinclude $vv tools/step_bench.hip $C/sf_kernels.hip -o /tmp/step_bench_v || exit 1
  echo "== variant '$vv'" >> $out
  /tmp/step_bench_v >> $out
done
cat $out

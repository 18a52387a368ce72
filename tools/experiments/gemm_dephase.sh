#!/bin/bash
# k_gemm with the second workgroup of every CU started half a K loop late (-DSF_EXP_DEPHASE=<sixteenths of the first tile's K loop>):
#   bash tools/experiments/gemm_dephase.sh  ->  gpurun_out/gemm_dephase.txt   (rebuilds the library on the box; the tree is a scratch copy there)
cd "$(dirname "$0")/../.."
C=sparse-matrix-factorization-library_amd/csrc
O=gpurun_out/gemm_dephase.txt
: > $O
for v in "" "-DSF_EXP_DEPHASE=8" "-DSF_EXP_DEPHASE=4"; do
  touch $C/sf_kernels.hip
  make -C $C CXXFLAGS="-O3 -std=c++17 -fPIC -I$PWD/include -Wall -Wno-unused-function $v" > /dev/null 2>&1 || { echo "build failed for $v" | tee -a $O; exit 1; }
  for i in 1 2; do
    python bench.py --cpu-grid 0 --no-pcie --no-secondary --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('variant [$v]', d['ms_per_step'], 'schur', r['kernel_ms'], 'outer', r['outer_gemm_ms'], 'step', r['fused_step_ms'], 'frac', r['frac'], 'res', d['config']['residual_device_solve'])" | tee -a $O
  done
done

#!/bin/bash
# Where the diagonal workgroup of a fused step spends its time (k_step compiled with -DSF_EXP_STEP_STAMPS):
#   bash tools/experiments/step_stamps.sh  ->  gpurun_out/step_stamps.txt
cd "$(dirname "$0")/../.."
C=sparse-matrix-factorization-library_amd/csrc
hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -I$C -Iinclude -DSF_EXP_STEP_STAMPS tools/step_stamps_bench.hip $C/sf_kernels.hip -o /tmp/step_stamps 2> gpurun_out/step_stamps.err || { tail -20 gpurun_out/step_stamps.err; exit 1; }
/tmp/step_stamps | tee gpurun_out/step_stamps.txt

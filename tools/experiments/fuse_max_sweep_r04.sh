#!/bin/bash
# SF_FUSE_MAX sweep with the round-4 kernels (LU config 5, config 3): bash tools/experiments/fuse_max_sweep_r04.sh -> gpurun_out/fuse_max_r04.txt
cd "$(dirname "$0")/../.."
O=gpurun_out/fuse_max_r04.txt
: > $O
for fm in 1024 2048 4096 8192 16384 32768; do
  SF_FUSE_MAX=$fm python bench.py --method lu --cpu-grid 0 --no-pcie --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('LU fuse_max $fm', d['ms_per_step'], 'step', r['fused_step_ms'], 'inner', r['inner_gemm_ms'], 'potrf', r['potrf_ms'], 'trsm', r['trsm_ms'])" | tee -a $O
done
for fm in 4096 8192 16384; do
  SF_FUSE_MAX=$fm python bench.py --workload stencil2d --cpu-grid 0 --no-pcie --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['schur_gemm_mfma']; print('C3 fuse_max $fm', d['ms_per_step'], 'step', r['fused_step_ms'], 'inner', r['inner_gemm_ms'], 'potrf', r['potrf_ms'], 'trsm', r['trsm_ms'])" | tee -a $O
done

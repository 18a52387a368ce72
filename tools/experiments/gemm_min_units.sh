# SF_GEMM_MIN_UNITS: the smallest share (in 16-deep K steps) a workgroup takes in the stream-K split of a GEMM launch's tail.
# 1 = equal shares over the whole grid (the form up to round 4).  Sweep on the three bench lines.
cd $GRAFT_REPO_ROOT
for M in 1 8 16 32 64 1; do
  echo "SF_GEMM_MIN_UNITS=$M"
  SF_GEMM_MIN_UNITS=$M python bench.py --workload stencil2d --cpu-grid 0 --no-pcie 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline'].get('schur_gemm_mfma', d['roofline']); print('  C3', d['ms_per_step'], r.get('outer_gemm_ms'), r.get('kernel_ms'), d['config'].get('residual_device_solve'))"
  SF_GEMM_MIN_UNITS=$M python bench.py --method lu --cpu-grid 0 --no-pcie 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('  LU', d['ms_per_step'], r.get('outer_gemm_ms'), r.get('kernel_ms'), d['config'].get('residual_device_solve'))"
done
for M in 1 16 32; do
  SF_GEMM_MIN_UNITS=$M python bench.py --cpu-grid 0 --no-pcie --no-secondary --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('SF_GEMM_MIN_UNITS=$M 128', d['ms_per_step'], r.get('outer_gemm_ms'), r.get('kernel_ms'), d['config'].get('residual_device_solve'))"
done

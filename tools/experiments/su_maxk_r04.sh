# SF_SU_MAXK (EXP build): the K up to which a Schur update goes to the one-wave k_update_small instead of k_gemm<1>; round-4 re-sweep
cd $GRAFT_REPO_ROOT
for K in 64 96 128 192; do
  echo "SF_SU_MAXK=$K"
  SF_SU_MAXK=$K python bench.py --workload stencil2d --cpu-grid 0 --no-pcie 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline'].get('schur_gemm_mfma', d['roofline']); print('  C3', d['ms_per_step'], r.get('kernel_ms'), r.get('small_update_ms'))"
  SF_SU_MAXK=$K python bench.py --method lu --cpu-grid 0 --no-pcie 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('  LU', d['ms_per_step'], r.get('kernel_ms'), r.get('small_update_ms'))"
  SF_SU_MAXK=$K python bench.py --cpu-grid 0 --no-pcie --no-secondary --steps 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('  128', d['ms_per_step'], r.get('kernel_ms'), r.get('small_update_ms'))"
done

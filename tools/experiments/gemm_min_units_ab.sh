cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for M in 1 16; do
  SF_GEMM_MIN_UNITS=$M python bench.py --cpu-grid 0 --no-pcie --no-secondary --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('M=$M 128', d['ms_per_step'], r.get('outer_gemm_ms'), r.get('kernel_ms'))"
done; done
for i in 1 2; do for M in 1 12 16 24; do
  SF_GEMM_MIN_UNITS=$M python bench.py --workload stencil2d --cpu-grid 0 --no-pcie 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('M=$M C3', d['ms_per_step'])"
  SF_GEMM_MIN_UNITS=$M python bench.py --method lu --cpu-grid 0 --no-pcie 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('M=$M LU', d['ms_per_step'])"
done; done

# SF_GRAPH=1 (a resident factorization replayed as one hipGraph) against the eager launches: parity tests, then the three bench lines
cd $GRAFT_REPO_ROOT
SF_GRAPH=1 python -m pytest tests/test_gpu_parity.py tests/test_edge_cases.py tests/test_lu.py tests/test_lu_pivot.py -m gpu -x -q > gpurun_out/graph_tests.log 2>&1 || { tail -30 gpurun_out/graph_tests.log; exit 1; }
tail -1 gpurun_out/graph_tests.log
for G in 0 1 0 1; do
  echo "SF_GRAPH=$G"
  SF_GRAPH=$G python bench.py --workload stencil2d --cpu-grid 0 --no-pcie 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('  C3', d['ms_per_step'], d['config'].get('residual_device_solve'))"
  SF_GRAPH=$G python bench.py --method lu --cpu-grid 0 --no-pcie 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('  LU', d['ms_per_step'], d['config'].get('residual_device_solve'))"
done
for G in 0 1; do
  SF_GRAPH=$G python bench.py --cpu-grid 0 --no-pcie --no-secondary --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('SF_GRAPH=$G 128', d['ms_per_step'], d['config'].get('residual_device_solve'))"
done

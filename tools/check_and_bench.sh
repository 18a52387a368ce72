# GPU suite + the three bench lines (LU config 5, config 3, 128^3) in one gpurun call: bash tools/check_and_bench.sh TAG
set -e
cd $GRAFT_REPO_ROOT
T=$1
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputests.log 2>&1 || { tail -30 gpurun_out/${T}_gputests.log; exit 1; }
tail -1 gpurun_out/${T}_gputests.log
python bench.py --method lu --cpu-grid 0 --no-pcie > gpurun_out/${T}_lu_config5_bench.json 2> gpurun_out/${T}_lu.err
python -c "
import json; d=json.load(open('gpurun_out/${T}_lu_config5_bench.json')); print('LU', d['ms_per_step'], d['value'], d['roofline']['fused_step_ms'], d['roofline']['kernel_ms'], d['roofline']['outer_gemm_ms'], d['config'].get('residual_device_solve'))"
python bench.py --workload stencil2d --cpu-grid 0 --no-pcie > gpurun_out/${T}_config3_bench.json 2> gpurun_out/${T}_c3.err
python -c "
import json; d=json.load(open('gpurun_out/${T}_config3_bench.json')); r=d['roofline'].get('schur_gemm_mfma', d['roofline']); print('C3', d['ms_per_step'], d['value'], r['fused_step_ms'], r['kernel_ms'], r['small_update_ms'], d['config'].get('residual_device_solve'))"
python bench.py --cpu-grid 0 --no-pcie --no-secondary --steps 5 > gpurun_out/${T}_128_bench.json 2> gpurun_out/${T}_128.err
python -c "
import json; d=json.load(open('gpurun_out/${T}_128_bench.json')); r=d['roofline']; print('128', d['ms_per_step'], d['value'], r['fused_step_ms'], r['kernel_ms'], r['outer_gemm_ms'], r['frac'], d['config'].get('residual_device_solve'))"

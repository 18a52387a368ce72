set -e
cd $GRAFT_REPO_ROOT
T=$1
python -m pytest tests/test_lu.py tests/test_lu_pivot.py -m gpu -x -q > gpurun_out/${T}_lutests.log 2>&1 || { tail -30 gpurun_out/${T}_lutests.log; exit 1; }
tail -1 gpurun_out/${T}_lutests.log
bash tools/experiments/step_stamps.sh > /dev/null 2>&1 || true
grep "LU " gpurun_out/step_stamps.txt | head -3
for i in 1 2; do
python bench.py --method lu --cpu-grid 0 --no-pcie > gpurun_out/${T}_lu_config5_bench.json 2> gpurun_out/${T}_lu.err
python -c "
import json; d=json.load(open('gpurun_out/${T}_lu_config5_bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['fused_step_ms'], d['roofline']['kernel_ms'], d['roofline']['outer_gemm_ms'], d['config'].get('residual_device_solve'))"
done
python tools/profile_launches.py lu > gpurun_out/${T}_launches_lu.txt 2>&1; cp gpurun_out/launches_lu.csv gpurun_out/${T}_launches_lu.csv; tail -8 gpurun_out/${T}_launches_lu.txt

set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_lu.py tests/test_lu_pivot.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r04_c_lutests.log 2>&1 || { tail -30 gpurun_out/r04_c_lutests.log; exit 1; }
tail -3 gpurun_out/r04_c_lutests.log
bash tools/experiments/step_stamps.sh > /dev/null 2>&1 || true
cp gpurun_out/step_stamps.txt gpurun_out/r04_c_step_stamps.txt
grep "LU" gpurun_out/r04_c_step_stamps.txt
python bench.py --method lu --cpu-grid 0 --no-pcie > gpurun_out/r04_c_lu_config5_bench.json 2> gpurun_out/r04_c_lu.err
python -c "
import json; d=json.load(open('gpurun_out/r04_c_lu_config5_bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['fused_step_ms'], d['roofline']['kernel_ms'], d['roofline']['outer_gemm_ms'], d['config'].get('residual_device_solve'))"

#!/bin/bash
# HBM traffic and MFMA counters of config 5 (LU, unsymmetric 79^3) from PMC counters: bash tools/collect_pmc_lu.sh TAG
# separate --pmc passes, no tracing flags; ONE factorization + one solve per pass
set -e
TAG=${1:-r04_k}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_ATOMIC_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"; do
  d=$O/lupmc_$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $d -- python3 bench.py --method lu --steps 1 --warmup 0 --cpu-grid 0 --no-roofline --no-pcie > /dev/null 2> $d.err
done
python tools/summarize_pmc.py --factorizations 1 --what "rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE TCC_EA0_ATOMIC_sum; SQ_* GRBM_GUI_ACTIVE; one pass per directory, no tracing flags) on python3 bench.py --method lu --steps 1 --warmup 0 --cpu-grid 0 --no-roofline --no-pcie (ONE LU factorization of config 5 + one device solve + the on-device validate), summed per kernel. fetch bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction)." $O/pmc_traffic_lu_config5.json $O/lupmc_FETCH_SIZE $O/lupmc_WRITE_SIZE $O/lupmc_SQ_WAVE_CYCLES > /dev/null
rm -rf $O/lupmc_FETCH_SIZE $O/lupmc_WRITE_SIZE $O/lupmc_SQ_WAVE_CYCLES
python - $O/pmc_traffic_lu_config5.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if isinstance(v, dict) and "hbm_bytes_total" in v: print(k[:60], {a: v[a] for a in list(v)[:8]})
PY

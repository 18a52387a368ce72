#!/bin/bash
# The host-buffer boundary as bench.py measures it (a process with PyTorch's own streams alive), several runs per setting:
#   bash tools/bench_pcie_ab.sh "SF_DL_HOST_WAIT=1" "SF_DL_HOST_WAIT=0" ...  ->  gpurun_out/bench_pcie_ab.txt
cd "$(dirname "$0")/.."
out=gpurun_out/bench_pcie_ab.txt; : > $out
for cfg in "$@"; do
  for rep in 1 2 3; do
    env $cfg timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-grid 0 --no-roofline --no-secondary > /tmp/pcie_ab.json 2>/dev/null || exit 1
    python - "$cfg" >> $out <<'PY'
import json, sys
d = json.loads(open("/tmp/pcie_ab.json").read().strip().splitlines()[-1]); pc = d["config"]["pcie_inclusive"]
print(sys.argv[1], "resident", d["ms_per_step"], "struct", pc["struct_first_call_ms"], pc["struct_second_call_ms"], "plan", pc["plan_first_call_ms"], pc["plan_second_call_ms"])
PY
  done
done
cat $out

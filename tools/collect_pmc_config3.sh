#!/bin/bash
# HBM traffic of config 3 (2-D 1000 x 1000 stencil) from PMC counters: bash tools/collect_pmc_config3.sh TAG
# separate --pmc passes, no tracing flags; ONE factorization + one solve per pass
set -e
TAG=${1:-r02_e}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_ATOMIC_sum"; do
  d=$O/c3pmc_$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $d -- python3 bench.py --workload stencil2d --steps 1 --warmup 0 --cpu-grid 0 --no-roofline --no-pcie > /dev/null 2> $d.err
done
python tools/summarize_pmc.py --factorizations 1 --what "rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE TCC_EA0_ATOMIC_sum; one pass per directory, no tracing flags) on python3 bench.py --workload stencil2d --steps 1 --warmup 0 --cpu-grid 0 --no-roofline --no-pcie (ONE factorization + one device solve + the on-device validate), summed per kernel. fetch bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction)." $O/pmc_traffic_config3.json $O/c3pmc_FETCH_SIZE $O/c3pmc_WRITE_SIZE > /dev/null
rm -rf $O/c3pmc_FETCH_SIZE $O/c3pmc_WRITE_SIZE
python - $O/pmc_traffic_config3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
tot = d.get("_total") or {}
print({k: v for k, v in tot.items()} if tot else list(d.keys())[:6])
PY

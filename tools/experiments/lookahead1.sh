#!/bin/bash
# One-GPU look-ahead (far part of block jo+2's update on the second stream beside near(jo+1) + chain(jo+1)): 128^3 and config 3 / 5,
# parity check + timing for several side-grid sizes.   bash tools/experiments/lookahead1.sh  ->  gpurun_out/lookahead1.txt
cd "$(dirname "$0")/../.."
out=gpurun_out/lookahead1.txt; : > $out
run() {  # label env...
  local label=$1; shift
  a=$(env "$@" python bench.py --cpu-grid 0 --no-pcie --no-secondary --no-roofline --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['residual_device_solve'])")
  echo "$label 128^3: $a" | tee -a $out
}
run "baseline            "  SF_X=0
run "lookahead1 grid 512 "  SF_LOOKAHEAD1=1
run "lookahead1 grid 448 "  SF_LOOKAHEAD1=1 SF_LOOKAHEAD1_GRID=448
run "lookahead1 grid 384 "  SF_LOOKAHEAD1=1 SF_LOOKAHEAD1_GRID=384
run "lookahead1 grid 256 "  SF_LOOKAHEAD1=1 SF_LOOKAHEAD1_GRID=256

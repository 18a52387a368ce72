#!/bin/bash
# The struct entry point's slow mode (0.55 s -> 0.9 s per call, sporadic): which knob does it follow?
#   bash tools/struct_mode_ab.sh "SF_DL_2D=1" "SF_DL_2D=0" ...   ->  gpurun_out/struct_mode_ab.txt   (12 calls per setting, call 0 builds the plan)
cd "$(dirname "$0")/.."
out=gpurun_out/struct_mode_ab.txt; : > $out
for cfg in "$@"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 200 python tools/struct_probe.py 128 12 > /tmp/mode_ab.log 2>&1 || exit 1
  grep -E "^call" /tmp/mode_ab.log | awk '{printf "%s ", $3} END {print ""}' >> $out
done
cat $out

#!/bin/bash
# The struct entry point's slow mode (0.55 s -> 0.9 s per call, sporadic): copy workers confined to the device's NUMA node
# (SF_DL_PIN, default 1) against left to the scheduler, and 6 / 10 / 14 workers; 10 calls each, call 0 (plan build) not counted
#   bash tools/struct_pin_ab.sh  ->  gpurun_out/struct_pin_ab.txt
cd "$(dirname "$0")/.."
out=gpurun_out/struct_pin_ab.txt; : > $out
for cfg in "1 6" "0 6" "1 10" "0 10" "1 14" "0 14" "1 6" "0 6"; do
  set -- $cfg
  echo "== SF_DL_PIN=$1 SF_DL_WORKERS=$2" >> $out
  SF_DL_PIN=$1 SF_DL_WORKERS=$2 SF_TRACE=1 timeout -k 10 200 python tools/struct_probe.py 128 10 > /tmp/pin_ab.log 2>&1 || exit 1
  grep -E "NUMA node" /tmp/pin_ab.log | sort -u >> $out
  grep -E "copy workers" /tmp/pin_ab.log | tail -1 >> $out
  grep -E "^call" /tmp/pin_ab.log | awk '{printf "%s ", $3} END {print ""}' >> $out
done
cat $out

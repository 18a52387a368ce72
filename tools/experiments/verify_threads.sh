#!/bin/bash
# threads of the host fingerprint pass of the verified struct-path solve: bash tools/experiments/verify_threads.sh -> gpurun_out/verify_threads.txt
cd "$(dirname "$0")/../.."
O=gpurun_out/verify_threads.txt
: > $O
for t in 8 16 32 48 64; do
  SF_VERIFY_THREADS=$t python bench.py --cpu-grid 0 --no-secondary --steps 2 --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pc=d['config']['pcie_inclusive']; print('SF_VERIFY_THREADS $t: struct_solve_ms', pc['struct_solve_ms'], 'trusted', pc['struct_solve_trusted_ms'], 'second call', pc['struct_second_call_ms'])" | tee -a $O
done

#!/bin/bash
# Owner-computes prototype (SF_TOP_OWNER=1) against the default replicate-and-all-reduce schedule:
#   bash tools/experiments/top_owner.sh  ->  gpurun_out/top_owner.txt
# (1) parity of the prototype on emulated handlers, (2) compute side of single ranks of 256^3 / 8 and 128^3 / 8 on the one GPU with both
# schedules (tools/emulate_rank.py: that rank's launches exactly, collectives skipped)
cd "$(dirname "$0")/../.."
O=gpurun_out/top_owner.txt
: > $O
python -m pytest tests/test_multi_handlers.py -m gpu -q -k "owner_computes" 2>&1 | tail -2 | tee -a $O
for cfg in "128 8 0" "128 8 5" "256 8 0" "256 8 5"; do
  set -- $cfg
  for own in 0 1; do
    echo "== grid $1 world $2 rank $3 SF_TOP_OWNER=$own" | tee -a $O
    SF_TOP_OWNER=$own python tools/emulate_rank.py --grid $1 --world $2 --rank $3 --steps 2 2>/dev/null | tee -a $O
  done
done

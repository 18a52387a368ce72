#!/bin/bash
# every rank of the default multi-GPU sizes, one after the other on ONE GPU (tools/emulate_rank.py, all-reduces skipped)
#   bash tools/emulate_all_ranks.sh  ->  gpurun_out/r04_s_emulated_all_ranks.jsonl
O=gpurun_out/r04_s_emulated_all_ranks.jsonl
: > $O
for r in 0 1 2 3 4 5 6 7; do python tools/emulate_rank.py --grid 256 --world 8 --rank $r --steps 1 >> $O 2>>gpurun_out/r04_s_emul.err || exit 1; echo "256/8 rank $r done"; done
for r in 0 1 2 3; do python tools/emulate_rank.py --grid 203 --world 4 --rank $r --steps 1 >> $O 2>>gpurun_out/r04_s_emul.err || exit 1; echo "203/4 rank $r done"; done
for r in 0 1; do python tools/emulate_rank.py --grid 161 --world 2 --rank $r --steps 1 >> $O 2>>gpurun_out/r04_s_emul.err || exit 1; echo "161/2 rank $r done"; done

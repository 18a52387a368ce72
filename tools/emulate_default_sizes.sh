#!/bin/bash
# rank 0 of the sizes `bench.py --gpus N` runs by default (rows per GPU fixed: 161^3 / 2, 203^3 / 4, 256^3 / 8 = BASELINE config 4),
# measured on ONE GPU with the all-reduces skipped (tools/emulate_rank.py): per-rank compute, storage, collective volume
O=gpurun_out/r03_e_emulated_rank_default_sizes.jsonl
: > $O
python tools/emulate_rank.py --grid 161 --world 2 --steps 1 >> $O 2>gpurun_out/r03_e_emul.err && \
python tools/emulate_rank.py --grid 203 --world 4 --steps 1 >> $O 2>>gpurun_out/r03_e_emul.err && \
python tools/emulate_rank.py --grid 256 --world 8 --steps 1 >> $O 2>>gpurun_out/r03_e_emul.err
cat $O

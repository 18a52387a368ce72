#!/bin/bash
# LDS-side counters of one 128^3 factorization (separate --pmc pass, no tracing flags): bank conflicts, unaligned stalls, LDS busy
#   bash tools/collect_lds_counters.sh  ->  gpurun_out/lds_counters.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/lds_pmc
rm -rf $O
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O -- python3 bench.py --grid 128 --steps 1 --warmup 0 --cpu-grid 0 --no-roofline --no-pcie --no-secondary > /dev/null 2> gpurun_out/lds_pmc.err
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/lds_pmc/*/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for path in f:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void sf::", "").replace("sf::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open("gpurun_out/lds_counters.txt", "w") as out:
    for k in sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", 0))[:6]:
        line = k + ": " + ", ".join("%s %.4g" % (c, v) for c, v in sorted(acc[k].items()))
        print(line); out.write(line + "\n")
PY
rm -rf $O

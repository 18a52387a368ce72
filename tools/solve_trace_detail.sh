#!/bin/bash
# per-launch (workgroups, microseconds) of the solve's step kernels, backward and forward: bash tools/solve_trace_detail.sh [grid]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_solve3
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_solve3 -- python3 tools/solve_probe.py ${1:-128} 2 > /dev/null 2>&1
t=$(ls gpurun_out/prof_solve3/*/*kernel_trace.csv | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "k_solve" in r["Kernel_Name"] and "small" not in r["Kernel_Name"]]
rows=rows[len(rows)//2:]
for kind in ("fwd","bwd"):
    xs=[(int(r["Grid_Size_X"])//256, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows if kind in r["Kernel_Name"]]
    print(kind, " ".join("%d:%.0f"%x for x in xs))
PY
rm -rf gpurun_out/prof_solve3

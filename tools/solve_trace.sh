#!/bin/bash
# per-launch durations of the device solve from a rocprofv3 kernel trace: bash tools/solve_trace.sh [grid]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_solve2
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_solve2 -- python3 tools/solve_probe.py ${1:-128} 2 2>&1 | grep -E "solve device"
t=$(ls gpurun_out/prof_solve2/*/*kernel_trace.csv | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "k_solve" in r["Kernel_Name"]]
rows=rows[len(rows)//2:]
for kind in ("fwd","bwd"):
    xs=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]), int(r["Grid_Size_X"])//256, "big" if "ILb1" in r["Kernel_Name"] or "<true>" in r["Kernel_Name"] else "small") for r in rows if kind in r["Kernel_Name"]]
    print(kind, len(xs), "launches, total ms %.2f" % (sum(x[0] for x in xs)/1e6))
    for lo,hi in ((0,2),(2,64),(64,512),(512,4096),(4096,1<<30)):
        sel=[x for x in xs if lo<=x[1]<hi]
        if sel: print("   workgroups [%d,%d): %d launches, total %.2f ms, median %.1f us"%(lo,hi,len(sel),sum(x[0] for x in sel)/1e6, sorted(x[0] for x in sel)[len(sel)//2]/1e3))
PY
rm -rf gpurun_out/prof_solve2

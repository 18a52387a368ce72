#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per kernel -> the JSON kept under profiles/.

usage: summarize_pmc.py [--factorizations K] [--what TEXT] OUT.json DIR [DIR ...]     (each DIR = one rocprofv3 -d directory, one pass each)

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE reads half of the bytes for this code's 8-byte-per-lane
loads (tools/fetch_calib.hip: 4 GiB read -> 2,097,163.6 KiB), so fetch bytes = 2 * FETCH_SIZE * 1024.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z_]+)(<[^>]*>)?", name)
    if not m:
        return name.split("(")[0]
    if m.group(1) == "k_gemm":
        t = re.search(r"k_gemm<(\d)", name)
        return "k_gemm<%s>" % (t.group(1) if t else "?")
    return m.group(1)


def main():
    argv = sys.argv[1:]
    nfact, what = 1, None
    while argv and argv[0].startswith("--"):
        if argv[0] == "--factorizations":
            nfact = int(argv[1]); argv = argv[2:]
        elif argv[0] == "--what":
            what = argv[1]; argv = argv[2:]
        else:
            raise SystemExit("unknown option " + argv[0])
    out, dirs = argv[0], argv[1:]
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(lambda: defaultdict(set))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    c = row["Counter_Name"]
                    acc[k][c] += float(row["Counter_Value"])
                    launches[k][c].add(row["Dispatch_Id"])
    res = {"_what": "rocprofv3 --pmc passes (one pass per directory, no tracing flags) on `python3 bench.py --grid 128 --steps 1 "
                    "--warmup 0 --cpu-grid 0 --no-roofline` (one factorization), summed per kernel. FETCH_SIZE/WRITE_SIZE in "
                    "KiB; fetch bytes = 2 * FETCH_SIZE * 1024 (gfx950 correction calibrated with tools/fetch_calib.hip).",
           "calibration": {"bytes_read": 4294967296, "FETCH_SIZE_KiB": 2097163.625, "factor": 2.0}}
    for k in sorted(acc, key=lambda k: -acc[k].get("FETCH_SIZE", 0)):
        e = {"launches": max(len(s) for s in launches[k].values())}
        for c, v in sorted(acc[k].items()):
            e[c if not c.endswith("_SIZE") else c + "_KiB"] = v
        if "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
            tot = (2.0 * acc[k]["FETCH_SIZE"] + acc[k]["WRITE_SIZE"]) * 1024.0
            e["hbm_bytes_total"] = tot
            e["hbm_bytes_per_launch"] = tot / e["launches"]
        if "SQ_BUSY_CYCLES" in acc[k] and "SQ_VALU_MFMA_BUSY_CYCLES" in acc[k] and acc[k]["SQ_BUSY_CYCLES"] > 0:
            e["mfma_busy_over_sq_busy"] = acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / acc[k]["SQ_BUSY_CYCLES"]
        res[k] = e
    if what:
        res["_what"] = what
    tot = sum(e.get("hbm_bytes_total", 0.0) for k, e in res.items() if isinstance(e, dict) and not k.startswith("_") and k != "calibration")
    res["_total"] = {"factorizations_in_the_run": nfact, "hbm_bytes_per_factorization": tot / nfact,
                     "note": "sum over all kernels of (2 x FETCH_SIZE + WRITE_SIZE) x 1024, divided by the factorizations in the profiled run"}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

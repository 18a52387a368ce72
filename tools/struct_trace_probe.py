#!/usr/bin/env python3
"""Per-piece download trace (SF_DL_TRACE) of every struct call, to compare a slow call with a fast one:
    python tools/struct_trace_probe.py [grid=128] [calls=12] -> gpurun_out/dltrace/call_<k>_<ms>.csv"""
import importlib, os, shutil, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "dltrace")
os.makedirs(out, exist_ok=True)
os.environ["SF_DL_TRACE"] = "/tmp/sf_dl_trace.csv"
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n, Cp, Ci, Cx = sf.gen.laplacian_lower(N, N, N)
common = sf.CommonInfo(dev_slot_size=sf.REFERENCE_SLOT_1GPU)
mi = sf.MatrixInfo()
mi.set_csc(n, Cp, Ci, Cx)
mi.set_perm(sf.grid_nd_perm(N, N, N, 3, 1))
mi.analyze(common)
kept = {"fast": 0, "slow": 0}
for k in range(calls):
    t0 = time.perf_counter(); mi.factorize(common); ms = 1e3 * (time.perf_counter() - t0)
    kind = "slow" if ms > 750 else "fast"
    print(f"call {k}: {ms:.1f} ms", flush=True)
    if k > 0 and kept[kind] < 2:
        kept[kind] += 1
        shutil.copy("/tmp/sf_dl_trace.csv", os.path.join(out, f"call_{k}_{kind}_{ms:.0f}.csv"))
mi.cleanup(); common.close()

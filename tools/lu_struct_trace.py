#!/usr/bin/env python3
"""Per-piece copy-back trace (SF_DL_TRACE) of the LU struct call at config 5 (79^3): where do the copy workers spend their time?
    python tools/lu_struct_trace.py [grid=79]"""
import importlib, os, sys, time, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SF_DL_TRACE"] = "/tmp/sf_dl_trace_lu.csv"
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 79
n, Cp, Ci, Cx = sf.gen.unsymmetric_stencil(M, M, M, extra_per_row=0, seed=2024, drop=0.05)
common = sf.CommonInfo(dev_slot_size=sf.REFERENCE_SLOT_1GPU)
mi = sf.LUMatrixInfo()
mi.set_csc(n, Cp, Ci, Cx, symmetric=False)
mi.set_perm(sf.grid_nd_perm(M, M, M, 3, 1))
mi.analyze(common)
for k in range(3):
    t0 = time.perf_counter(); mi.factorize(common); print(f"call {k}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
rows = list(csv.DictReader(open("/tmp/sf_dl_trace_lu.csv")))
cp = [(float(r["t_copied_ms"]) - float(r["t_dma_done_ms"]), float(r["doubles"]) * 8 / 1e6, float(r["t_published_ms"]), float(r["t_dma_done_ms"]), float(r["t_copied_ms"])) for r in rows]
print("pieces", len(cp), "MB", sum(c[1] for c in cp), "sum host-copy ms", sum(c[0] for c in cp), "last published", max(c[2] for c in cp), "last dma", max(c[3] for c in cp), "last copied", max(c[4] for c in cp))
slow = sorted(cp, key=lambda c: -c[0])[:8]
for c in slow: print("  host copy %.2f ms for %.1f MB (%.1f GB/s), published at %.1f ms" % (c[0], c[1], c[1] / max(c[0], 1e-6), c[2]))
import numpy as np
a = np.array(cp)
for lo, hi in ((0, 20), (20, 40), (40, 60), (60, 90), (90, 130), (130, 400)):
    sel = a[(a[:, 2] >= lo) & (a[:, 2] < hi)]
    if len(sel): print(f"  published in [{lo},{hi}) ms: {len(sel)} pieces, {sel[:,1].sum():.0f} MB, host copy {sel[:,0].sum():.1f} ms, rate {sel[:,1].sum()/max(sel[:,0].sum(),1e-6):.1f} GB/s per worker-time")
mi.cleanup(); common.close()

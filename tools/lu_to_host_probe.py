#!/usr/bin/env python3
"""LU: is the factorization itself slower while its factor is being copied back?  Wall time of sf_chol_plan_factorize_to_host and
the compute stream's own time (ev0 -> ev1) inside it, against the resident step.   python tools/lu_to_host_probe.py [grid=110] [lu|chol]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 110
lu = (sys.argv[2] if len(sys.argv) > 2 else "lu") == "lu"
if lu:
    n, Cp, Ci, Cx = sf.gen.unsymmetric_stencil(M, M, M, extra_per_row=0, seed=2024, drop=0.05)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(M, M, M, 3, 1), sf.REFERENCE_SLOT_1GPU, "lu", False)
    plan = sf.LUPlan(sym); plan.set_values(sym.Lx, sym.Ux)
else:
    n, Cp, Ci, Cx = sf.gen.laplacian_lower(M, M, M)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(M, M, M, 3, 1), sf.REFERENCE_SLOT_1GPU)
    plan = sf.CholPlan(sym); plan.set_values(sym.Lx)
for _ in range(3):
    plan.factorize()
    print("resident: compute %.1f ms" % plan.stat("last_ms"), flush=True)
host = np.empty(sym.xsize)
for k in range(4):
    t0 = time.perf_counter()
    if lu: plan.factorize_to_host(sym.Lx, sym.Ux, host)
    else: plan.factorize_to_host(sym.Lx, host)
    print("to_host %d: wall %.1f ms, compute stream %.1f ms, factor %.1f GB" % (k, 1e3 * (time.perf_counter() - t0), plan.stat("last_ms"), sym.xsize * 8 / 1e9), flush=True)

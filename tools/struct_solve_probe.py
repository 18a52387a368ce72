#!/usr/bin/env python3
"""Time of the struct path's stages at a given grid: python tools/struct_solve_probe.py [grid]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n, Cp, Ci, Cx = sf.gen.laplacian_lower(N, N, N)
common = sf.CommonInfo(dev_slot_size=sf.REFERENCE_SLOT_1GPU)
mi = sf.MatrixInfo()
mi.set_csc(n, Cp, Ci, Cx)
mi.set_perm(sf.grid_nd_perm(N, N, N, 3, 1))
t0 = time.perf_counter(); mi.analyze(common); print(f"analyze {time.perf_counter() - t0:.3f} s", flush=True)
for k in range(2):
    t0 = time.perf_counter(); mi.factorize(common); print(f"factorize {time.perf_counter() - t0:.3f} s", flush=True)
    t0 = time.perf_counter(); r = mi.validate(); print(f"validate (solve + residual) {time.perf_counter() - t0:.3f} s, solveTime {mi.c.solveTime:.3f}, residual {r:.2e}", flush=True)
mi.cleanup(); common.close()

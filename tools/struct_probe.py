#!/usr/bin/env python3
"""Where the time of the struct entry point goes (SF_TRACE=1): python tools/struct_probe.py [grid] [calls]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SF_TRACE"] = "1"
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n, Cp, Ci, Cx = sf.gen.laplacian_lower(N, N, N)
common = sf.CommonInfo(dev_slot_size=sf.REFERENCE_SLOT_1GPU)
mi = sf.MatrixInfo()
mi.set_csc(n, Cp, Ci, Cx)
mi.set_perm(sf.grid_nd_perm(N, N, N, 3, 1))
t0 = time.perf_counter(); mi.analyze(common); print(f"analyze {time.perf_counter() - t0:.3f} s", flush=True)
for k in range(calls):
    t0 = time.perf_counter(); mi.factorize(common)
    print(f"call {k}: {1e3 * (time.perf_counter() - t0):.1f} ms (factorizeTime {1e3 * mi.c.factorizeTime:.1f})", flush=True)
mi.cleanup(); common.close()

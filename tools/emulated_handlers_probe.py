import importlib, os, sys, time
import ctypes as C
sys.path.insert(0, os.getcwd())
os.environ["SF_EMULATE_HANDLERS"] = sys.argv[1] if len(sys.argv) > 1 else "8"
os.environ["SF_TRACE"] = "1"
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n, Cp, Ci, Cx = sf.gen.laplacian_lower(N, N, N)
common = sf.CommonInfo(dev_slot_size=sf.REFERENCE_SLOT_1GPU)
print("handlers", common.c.numGPU, flush=True)
mi = sf.MatrixInfo()
mi.set_csc(n, Cp, Ci, Cx)
mi.set_perm(sf.grid_nd_perm(N, N, N, 3, 1))
mi.analyze(common)
C.memset(mi.c.Lsx, 0xff, 8 * int(mi.c.xsize))
for k in range(2):
    t0 = time.perf_counter(); mi.factorize(common); t1 = time.perf_counter()
    r = mi.validate(); t2 = time.perf_counter()
    print(f"call {k}: factorize {t1-t0:.3f} s, validate {t2-t1:.3f} s (solve {mi.c.solveTime:.3f}), residual {r:.2e}", flush=True)
mi.cleanup(); common.close()

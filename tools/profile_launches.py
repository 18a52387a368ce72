"""Per-launch timing of one 128^3 factorization (HIP events around every launch, SF_PROFILE_DUMP csv):
    python tools/profile_launches.py   ->  gpurun_out/launches.csv  (launch, kind, tasks, units, flops, ms)
kinds: 0 POTRF, 1 TRSM, 2 inner GEMM, 3 Schur (k_gemm<1>), 4 outer GEMM, 5 fused step (k_step), 6 small Schur (k_update_small)."""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
g = 128
n, Cp, Ci, Cx = sf.gen.laplacian_lower(g, g, g)
sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(g, g, g, 3, 1), sf.REFERENCE_SLOT_1GPU)
plan = sf.CholPlan(sym); plan.set_values(sym.Lx)
plan.factorize(); plan.factorize()
os.environ["SF_PROFILE_DUMP"] = "gpurun_out/launches.csv"
plan.set_profiling(True); plan.factorize(); plan.set_profiling(False)
print("total", plan.stat("last_ms"))

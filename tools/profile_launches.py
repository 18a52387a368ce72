"""Per-launch timing of one factorization (HIP events around every launch, SF_PROFILE_DUMP csv):
    python tools/profile_launches.py [lap3d|stencil2d|lu] [grid]  ->  gpurun_out/launches_<workload>.csv  (launch, kind, tasks, units, flops, ms)
kinds: 0 POTRF, 1 TRSM, 2 inner GEMM, 3 Schur (k_gemm<1>), 4 outer GEMM, 5 fused step (k_step), 6 small Schur (k_update_small)."""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
wl = sys.argv[1] if len(sys.argv) > 1 else "lap3d"
g = int(sys.argv[2]) if len(sys.argv) > 2 else {"lap3d": 128, "stencil2d": 1000, "lu": 79}[wl]
if wl == "stencil2d":
    n, Cp, Ci, Cx = sf.gen.stencil_spd_lower(g, g)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(g, g, 1, 3, 2), sf.REFERENCE_SLOT_1GPU)
    plan = sf.CholPlan(sym); plan.set_values(sym.Lx)
elif wl == "lu":
    n, Cp, Ci, Cx = sf.gen.unsymmetric_stencil(g, g, g, extra_per_row=0, seed=2024, drop=0.05)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(g, g, g, 3, 1), sf.REFERENCE_SLOT_1GPU, "lu", False)
    plan = sf.LUPlan(sym); plan.set_values(sym.Lx, sym.Ux); plan.set_pivoting(0.1)
else:
    n, Cp, Ci, Cx = sf.gen.laplacian_lower(g, g, g)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(g, g, g, 3, 1), sf.REFERENCE_SLOT_1GPU)
    plan = sf.CholPlan(sym); plan.set_values(sym.Lx)
plan.factorize(); plan.factorize()
print("total (unprofiled)", plan.stat("last_ms"), "ms; nsuper", sym.nsuper, "levels", plan.stat("levels"), "launches", plan.stat("launches"))
os.makedirs("gpurun_out", exist_ok=True)
os.environ["SF_PROFILE_DUMP"] = f"gpurun_out/launches_{wl}.csv"
plan.set_profiling(True); plan.factorize(); plan.set_profiling(False)
import csv, collections
agg = collections.defaultdict(lambda: [0, 0.0])
rows = list(csv.DictReader(open(os.environ["SF_PROFILE_DUMP"])))
for r in rows:
    agg[int(r["kind"])][0] += 1; agg[int(r["kind"])][1] += float(r["ms"])
print("load/memset ms", plan.stat("last_load_ms"))
for k in sorted(agg): print("kind", k, "launches", agg[k][0], "ms %.3f" % agg[k][1])
if wl != "lu":
    import numpy as np
    x = plan.solve(1 + np.arange(n) / n)
    print("device solve ms", plan.stat("last_solve_ms"), "residual", sf.validate_solution(sym, x))

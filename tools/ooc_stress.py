"""Out-of-core plans under repetition: the same matrix factorized `reps` times through one out-of-core plan with many small groups
and small staging slots; every result must equal the first one's defined entries to 1e-13 (the scatter's atomics reorder sums) and
the in-core plan's.  A buffer re-used too early or a piece copied before it is final shows up as a mismatch in some repetition.
    python tools/ooc_stress.py [N=40] [reps=40] [budget_fraction=0.05] [top_mode = the partition's]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SF_DL_SLOT_MB", "1")
import numpy as np
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
gen = importlib.import_module("sparse-matrix-factorization-library_amd.gen")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
S = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N), sf.REFERENCE_SLOT_1GPU)
total = int((np.diff(S.Super) * np.diff(S.Lsip)).sum())
plan0 = sf.CholPlan(S)
ref = np.full(S.xsize, np.nan)
plan0.factorize_to_host(S.Lx, out=ref)
plan0.close()
mask = np.isfinite(ref)
scale = np.abs(ref[mask]).max()
cut = sf.ooc_partition(S, int(total * frac))
g, ng, ge, te, nd, fits = cut
mode = int(sys.argv[4]) if len(sys.argv) > 4 else cut.top_mode
print(f"N {N}: top mode {mode}, {ng} groups, buffers 2 x {8 * ge / 1e6:.1f} MB, top {8 * te / 1e6:.1f} MB of {8 * total / 1e6:.1f} MB", flush=True)
plan = sf.CholPlan(S, ooc_group=g, ooc_ngroups=ng, ooc_top_mode=mode)
worst = 0.0
for r in range(reps):
    out = np.full(S.xsize, np.nan)
    plan.factorize_to_host(S.Lx * (1.0 + 0.0 * r), out=out)
    assert np.array_equal(np.isfinite(out), mask) or not np.isnan(out[mask]).any(), f"rep {r}: unwritten entries"
    err = float(np.max(np.abs(out[mask] - ref[mask])) / scale)
    worst = max(worst, err)
    assert err <= 1e-13, f"rep {r}: {err}"
print(f"{reps} repetitions, worst relative difference to the in-core factor {worst:.2e}")

"""Negative control for sf_handlers_replica_mismatches: with SF_GEMM_WHOLE_TILES=0 the replicated near parts of the look-ahead
schedule split their tiles by K like every other launch (several atomic additions per target element, in an order that differs
from rank to rank) and the ranks' copies of a shared panel differ in their last bits; with the default they are bit-identical.

    python tools/experiments/replica_identity.py [N=40] [handlers=2]
"""
import importlib, os, subprocess, sys

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.getcwd())
    sf = importlib.import_module("sparse-matrix-factorization-library_amd")
    N = int(sys.argv[2])
    for method in ("cholesky", "lu"):
        if method == "cholesky":
            n, Cp, Ci, Cx = sf.gen.laplacian_lower(N, N, N)
            mi = sf.MatrixInfo(); mi.set_csc(n, Cp, Ci, Cx)
        else:
            n, Cp, Ci, Cx = sf.gen.unsymmetric_stencil(N, N, N, seed=13)
            mi = sf.LUMatrixInfo(); mi.set_csc(n, Cp, Ci, Cx, symmetric=False)
        common = sf.CommonInfo(dev_slot_size=8 << 30)
        mi.set_perm(sf.grid_nd_perm(N, N, N))
        mi.analyze(common)
        mi.factorize(common)
        bad = sf._lib.lib.sf_handlers_replica_mismatches(mi.c.Lsx)
        print(f"  {method}: {bad} values of the shared panels differ between ranks, residual {mi.validate():.2e}", flush=True)
        mi.cleanup(); common.close()
    sys.exit(0)

N = sys.argv[1] if len(sys.argv) > 1 else "40"
nh = sys.argv[2] if len(sys.argv) > 2 else "2"
for whole in ("1", "0"):
    print(f"SF_GEMM_WHOLE_TILES={whole} ({nh} emulated handlers, {N}^3)", flush=True)
    env = dict(os.environ, SF_EMULATE_HANDLERS=nh, SF_GEMM_WHOLE_TILES=whole)
    subprocess.run([sys.executable, __file__, "--child", N], env=env, check=True)

"""How many MFMA flops do the 128 x 128 GEMM tiles issue, against the flops of the exact trapezoids they cover?

    python tools/tile_quantisation.py [grid=128]
"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sf = importlib.import_module("sparse-matrix-factorization-library_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n, Cp, Ci, Cx = sf.gen.laplacian_lower(N, N, N)
sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N), sf.REFERENCE_SLOT_1GPU)
plan = sf.CholPlan(sym)
g = plan.stat
upd = g("flops_update") - g("flops_update_small")
out = g("flops_outer_gemm")
print(f"grid {N}: Schur updates through k_gemm<1>: exact {upd:.4e}, tiles {g('flops_tiles_update'):.4e} ({g('flops_tiles_update') / upd:.4f}x)")
print(f"          outer blocks through k_gemm<0>:  exact {out:.4e}, tiles {g('flops_tiles') - g('flops_tiles_update'):.4e} "
      f"({(g('flops_tiles') - g('flops_tiles_update')) / out:.4f}x)")
print(f"          F_exec {g('flops_exec'):.4e}, gemm tasks {g('gemm_tasks'):.0f}")

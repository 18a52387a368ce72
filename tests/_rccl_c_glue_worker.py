"""worker of test_gpu_parity.py::test_rccl_c_path_glue_one_rank: the product path of a multi-GPU run -- ShardedFactorization with
the nccl backend: unique id broadcast over torch.distributed, communicator created inside the C library, proportionally mapped
plan, sf_chol_plan_factorize_distributed -- forced onto a ONE-rank group (SF_FORCE_DISTRIBUTED=1; two ranks cannot share the one
GPU of the test box).  What a one-rank run can check: the glue executes, the C driver is the one that ran, and the factor equals
the single-GPU plan's."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from util import sf, gen, nd_perm_py  # noqa: E402
from importlib import import_module  # noqa: E402


def main():
    sharded = import_module("sparse-matrix-factorization-library_amd.sharded")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    for method in ("cholesky", "lu"):
        N = 16
        if method == "lu":
            n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=2)
            sym = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(N, N, N), 1 << 30, "lu", False)
            ref_plan = sf.LUPlan(sym); ref_plan.set_values(sym.Lx, sym.Ux)
        else:
            n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
            sym = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(N, N, N), 1 << 30)
            ref_plan = sf.CholPlan(sym); ref_plan.set_values(sym.Lx)
        ref_plan.factorize()
        ref = ref_plan.get_factor().copy()
        ref_plan.close()
        S = sharded.ShardedFactorization(sym, 0, 1, device=0, mode="distributed")
        assert S.engine.comm_kind == "rccl-c", S.engine.comm_kind
        info = S.plan_info()
        assert info["collectives"] == "rccl-c" and info["mode"] == "distributed"
        if method == "lu":
            S.set_values(sym.Lx, sym.Ux)
        else:
            S.set_values(sym.Lx)
        for _ in range(2):
            S.factorize()
        got = S.gather_factor()
        assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref)), method
        b = 1.0 + np.arange(n) / n
        x = S.solve(b)                              # sf_chol_plan_solve_distributed through the same glue
        lc = np.repeat(np.arange(n), np.diff(sym.Lp))
        import scipy.sparse as sp
        A = sp.coo_matrix((sym.Lx, (sym.Li, lc)), shape=(n, n)).tocsr()
        if method == "lu":
            ur = np.repeat(np.arange(n), np.diff(sym.Up))
            off = sym.Ui != ur
            A = A + sp.coo_matrix((sym.Ux[off], (ur[off], sym.Ui[off])), shape=(n, n)).tocsr()
        else:
            off = sym.Li != lc
            A = A + sp.coo_matrix((sym.Lx[off], (lc[off], sym.Li[off])), shape=(n, n)).tocsr()
        r = A @ x - b
        assert np.abs(r).max() / (abs(A).sum(axis=0).max() * np.abs(x).max() + np.abs(b).max()) <= 1e-13, method
        S.close()
    # the launcher's fall-back: when the communicator check fails (here: a test hook), every rank closes its mapped plan and its
    # communicator and the run goes through torch.distributed's collectives instead -- same factor
    os.environ["SF_TEST_FAIL_COMM_CHECK"] = "1"
    N = 16
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    sym = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(N, N, N), 1 << 30)
    ref_plan = sf.CholPlan(sym); ref_plan.set_values(sym.Lx); ref_plan.factorize()
    ref = ref_plan.get_factor().copy()
    ref_plan.close()
    S = sharded.ShardedFactorization(sym, 0, 1, device=0, mode="distributed")
    assert S.engine.comm_kind == "torch" and S.engine.comm is None, S.engine.comm_kind
    S.set_values(sym.Lx)
    S.factorize()
    got = S.gather_factor()
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref))
    S.close()
    del os.environ["SF_TEST_FAIL_COMM_CHECK"]
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_C_GLUE_OK")


if __name__ == "__main__":
    main()

"""worker of tests/test_sharded_gloo.py: one rank of a gloo group running ShardedCholesky on the numpy engine"""
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from util import sf, gen, nd_perm_py, rel_err  # noqa: E402
from cpu_engine import NumpyEngine  # noqa: E402
import oracle  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle.blas_init("builtin")
    results = []
    for dims, mode in (((12, 12, 1), "replicated"), ((8, 8, 8), "replicated"), ((12, 12, 1), "distributed"),
                       ((8, 8, 8), "distributed")):
        n, Cp, Ci, Cx = gen.laplacian_lower(*dims)
        sym = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(*dims), 1 << 30)
        sh = sf.ShardedCholesky(sym, rank, world, engine_factory=NumpyEngine, mode=mode)
        info = sh.plan_info()
        assert info["mode"] == mode and (info["segments"] > 0) == (mode == "distributed"), info
        assert info["top_supernodes"] > 0 and all(c > 0 for c in info["subtrees_per_rank"]), info
        # every supernode is stored by its owner, top supernodes by everybody, nothing else
        assert np.array_equal(sh.phase == 0, sh.owner == rank) and np.array_equal(sh.phase == 1, sh.owner < 0)
        sh.set_values(sym.Lx)
        sh.factorize()
        full = sh.gather_factor()
        ref, inf, _ = oracle.chol_factorize(sym)
        assert inf == 0
        err = rel_err(full, ref, oracle.lower_mask(sym))
        res, _ = oracle.chol_residual(sym, full)
        assert err <= 1e-12 and res <= 1e-13, (err, res)
        # second factorization with new values re-uses the plan
        sh.set_values(sym.Lx * 2.0)
        sh.factorize()
        full2 = sh.gather_factor()
        assert rel_err(full2, ref * np.sqrt(2.0), oracle.lower_mask(sym)) <= 1e-12
        results.append((dims, mode, err, res, info["model_speedup_bound"]))
    dist.barrier()
    if rank == 0:
        print("SHARDED_OK", results)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""One Cholesky factorization sharded over the ranks of a torch.distributed group by elimination-tree subtrees
(SURVEY 8e).  One process per GPU; the only data-path exchange is ONE sum all-reduce (RCCL over xGMI on GPUs,
gloo in the CPU tests) of the contiguous top-panel region between the two phases:

    phase 0   every rank assembles and factorizes its own subtrees; their Schur updates into the (replicated,
              zero-initialised except on rank 0, which also holds the matrix entries) top panels accumulate locally
    all-reduce(sum) over the top region
    phase 1   every rank factorizes the top supernodes (replicated; the subtree-only Amdahl limit is reported
              by `plan_info`)

The numeric engine is pluggable: `HipEngine` (the product, libsparseframe_hip.so) or any object with the same
three methods (tests use a numpy engine so that the orchestration runs under gloo without a GPU).
"""
import numpy as np

from .api import CholPlan, subtree_partition, phases_for_rank


class _DevArray:
    """exposes a raw device pointer through __cuda_array_interface__ so that torch can alias it"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 3,
                                         "strides": None}


class HipEngine:
    def __init__(self, sym, phase, load_top, device):
        self.plan = CholPlan(sym, device=device, phase=phase, load_top=load_top)
        self.device = device
        self._top = None

    def set_values(self, Lx):
        self.plan.set_values(Lx)

    def factorize_phase(self, which):
        # synchronous on the plan's own stream: the collective that follows runs on torch's stream
        self.plan.factorize_phase(which, sync=True)

    def top_tensor(self):
        import torch
        if self._top is None:
            ptr, cnt = self.plan.top_region()
            self._top = torch.as_tensor(_DevArray(ptr, cnt), device=f"cuda:{self.device}") if cnt > 0 else \
                torch.zeros(0, dtype=torch.float64, device=f"cuda:{self.device}")
        return self._top

    def get_factor(self, out=None):
        return self.plan.get_factor(out)

    def close(self):
        self.plan.close()


class ShardedCholesky:
    def __init__(self, sym, rank, world, device=0, engine_factory=None, group=None):
        self.sym, self.rank, self.world, self.group = sym, rank, world, group
        self.owner, self.top_fraction, self.max_load_fraction = subtree_partition(sym, world)
        phase = phases_for_rank(self.owner, rank)
        factory = engine_factory or (lambda s, ph, lt: HipEngine(s, ph, lt, device))
        self.engine = factory(sym, phase, rank == 0)
        self.phase = phase

    def plan_info(self):
        tf, ml = self.top_fraction, self.max_load_fraction
        return {"subtrees_per_rank": [int(np.count_nonzero(self.owner == r)) for r in range(self.world)],
                "top_supernodes": int(np.count_nonzero(self.owner < 0)),
                "top_flop_fraction": tf, "max_rank_subtree_flop_fraction": ml,
                "amdahl_speedup_bound": 1.0 / (tf + ml) if tf + ml > 0 else float(self.world)}

    def set_values(self, Lx):
        self.engine.set_values(Lx)

    def factorize(self):
        import torch.distributed as dist
        self.engine.factorize_phase(0)
        if self.world > 1:
            top = self.engine.top_tensor()
            if top.numel() > 0:
                dist.all_reduce(top, op=dist.ReduceOp.SUM, group=self.group)
                if top.is_cuda:
                    import torch
                    torch.cuda.synchronize(top.device)
        self.engine.factorize_phase(1)

    def gather_factor(self):
        """full factor in the reference layout on every rank (all-reduce of the disjoint subtree panels; the
        replicated top panels are taken from this rank).  Test / validation helper, not part of the timed path."""
        import torch
        import torch.distributed as dist
        mine = self.engine.get_factor()
        if self.world == 1:
            return mine
        S = self.sym
        sub = np.zeros_like(mine)
        for s in np.flatnonzero(self.phase == 0):
            sub[S.Lsxp[s]:S.Lsxp[s + 1]] = mine[S.Lsxp[s]:S.Lsxp[s + 1]]
        t = torch.from_numpy(sub)
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        full = t.cpu().numpy()
        for s in np.flatnonzero(self.phase == 1):
            full[S.Lsxp[s]:S.Lsxp[s + 1]] = mine[S.Lsxp[s]:S.Lsxp[s + 1]]
        return full

    def close(self):
        self.engine.close()

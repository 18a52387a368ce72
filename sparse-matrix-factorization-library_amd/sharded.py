"""One factorization (Cholesky, or no-pivot LU in the distributed mode) sharded over the ranks of a torch.distributed group by elimination-tree subtrees
(SURVEY 8e).  One process per GPU; the data-path exchange is sum all-reduces (RCCL over xGMI on GPUs, gloo in the
CPU tests) of top-panel regions.  Two ways of handling the top supernodes (those above the subtrees):

mode "distributed" (default for world > 1)
    phase 0       every rank assembles and factorizes its own subtrees; their Schur updates into the top panels
                  accumulate in the rank's own copy (zero-initialised except on rank 0, which holds the matrix entries)
    per segment   (one per top level and 512-column block) all-reduce(sum) of that block of the level's panels (its
                  rows from the block's first column down, packed into one buffer: one collective per segment), then
                  the block's sequential 64-column POTRF/TRSM chain on every rank (replicated: latency-bound), then this
                  rank's 1/world share of the large GEMMs that follow (next block's left-looking GEMM, the level's
                  Schur updates) -- additive, so any split is valid and nothing is exchanged until the target
                  block's own reduce point.  Every top panel crosses xGMI exactly once.

mode "replicated"
    phase 0, ONE all-reduce over the contiguous top region, phase 1 replicated on every rank (the subtree-only
    Amdahl limit is reported by `plan_info`).

The numeric engine is pluggable: `HipEngine` (the product, libsparseframe_hip.so) or any object with the same
methods (tests use a numpy engine so that the orchestration runs under gloo without a GPU).
"""
import numpy as np

from .api import CholPlan, LUPlan, subtree_partition, phases_for_rank, top_groups

#: cost of a top flop relative to a subtree flop in the distributed mode: the split share plus the replicated
#: 64-column chain and the all-reduce (about a quarter of the top's single-GPU time at 128^3)
TOP_CHAIN_SHARE = 0.25


class _DevArray:
    """exposes a raw device pointer through __cuda_array_interface__ so that torch can alias it"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 3,
                                         "strides": None}


class HipEngine:
    def __init__(self, sym, phase, load_top, device, rank=0, world=1, distributed=False, group=None, owner=None):
        import os
        # SF_FORCE_DISTRIBUTED=1 (tests): take the distributed path even with one rank, so that the glue below -- unique id over
        # torch.distributed, the C-side communicator, the mapped plan, sf_chol_plan_factorize_distributed -- runs on a one-GPU box
        forced = os.environ.get("SF_FORCE_DISTRIBUTED") == "1"
        self.distributed = bool(distributed and (world > 1 or forced))
        self.lu = bool(getattr(sym, "lu", False))
        if self.lu and world > 1 and not self.distributed:
            raise ValueError("sharded LU needs mode='distributed'")
        cls = LUPlan if self.lu else CholPlan
        self.device = device
        self._top = None
        self._seg = {}
        self.comm = None
        self.comm_kind = "none"
        if self.distributed:
            import os
            import torch
            import torch.distributed as dist
            want_c = (dist.is_initialized() and dist.get_backend(group) == "nccl" and dist.get_world_size(group) == world
                      and owner is not None and os.environ.get("SF_BENCH_COMM", "rccl-c") != "torch")
            if want_c:
                # the product path: the collectives are issued by the C library itself (RCCL on the plan's stream,
                # sf_chol_plan_factorize_distributed); torch.distributed only carries the 128-byte unique id.  Every rank
                # reports whether its communicator came up; unless all did, all fall back to torch's collectives together.
                from .api import Comm
                ok = 1
                try:
                    uid = torch.zeros(128, dtype=torch.uint8, device=f"cuda:{device}")
                    if rank == 0:
                        uid.copy_(torch.frombuffer(bytearray(Comm.unique_id()), dtype=torch.uint8))
                    dist.broadcast(uid, 0, group=group)
                    self.comm = Comm(device, rank, world, bytes(uid.cpu().numpy().tobytes()))
                except Exception as e:          # noqa: BLE001 -- any failure here means "use the torch path"
                    import sys
                    print(f"[sparseframe-hip] rank {rank}: C-side RCCL communicator unavailable ({e}); falling back to torch.distributed",
                          file=sys.stderr)
                    ok = 0
                def agreed(ok):
                    flag = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{device}")
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                    return int(flag.item()) != 0

                def give_up(what, e):
                    import sys
                    print(f"[sparseframe-hip] rank {rank}: {what} failed ({e}); falling back to torch.distributed", file=sys.stderr)
                    return 0

                plan = None
                if not agreed(ok):
                    ok = 0
                if ok:
                    # proportionally mapped plan (a top supernode lives on the ranks below it), groups and their collectives
                    # handled by the C library
                    try:
                        plan = cls(sym, device=device, owner=owner, rank=rank, nranks=world)
                    except Exception as e:      # noqa: BLE001
                        ok = give_up("the mapped plan", e)
                    if not agreed(ok):          # (before the collective below: a rank without a plan must not leave the others in it)
                        ok = 0
                if ok:
                    try:
                        plan.prepare_comm(self.comm)        # sub-communicators + one checked sum per communicator, now
                        if os.environ.get("SF_TEST_FAIL_COMM_CHECK") == "1":     # tests: the fall-back below
                            raise RuntimeError("SF_TEST_FAIL_COMM_CHECK")
                    except Exception as e:      # noqa: BLE001
                        ok = give_up("the communicator check", e)
                    if not agreed(ok):
                        ok = 0
                if not ok:
                    # (communicator first: RCCL still refers to the stream of its last collective, which is the plan's)
                    if self.comm is not None:
                        self.comm.close()
                    self.comm = None
                    if plan is not None:
                        plan.close()
                else:
                    self.plan = plan
                    self.comm_kind = "rccl-c"
        if self.comm is not None:
            pass
        elif self.lu and world == 1:
            self.plan = LUPlan(sym, device=device)
        else:
            self.plan = cls(sym, device=device, phase=phase, load_top=load_top,
                            rank=rank if self.distributed else 0, nranks=world if self.distributed else 1)
            if self.distributed:
                # the collective is torch's (gloo rehearsals with several ranks on one GPU; the fallback of the nccl path): run
                # on the stream it is ordered with
                import torch
                self.plan.set_stream(torch.cuda.current_stream(device).cuda_stream)
                self.comm_kind = "torch"

    def set_values(self, Lx, Ux=None):
        if self.lu:
            self.plan.set_values(Lx, Ux)
        else:
            self.plan.set_values(Lx)

    def factorize_phase(self, which):
        # replicated mode: synchronous on the plan's own stream, the collective that follows runs on torch's stream
        self.plan.factorize_phase(which, sync=not self.distributed)

    def _alias(self, offset, count):
        import torch
        return torch.as_tensor(_DevArray(self.plan.factor_device_ptr + 8 * offset, count), device=f"cuda:{self.device}")

    def top_tensor(self):
        import torch
        if self._top is None:
            ptr, cnt = self.plan.top_region()
            self._top = torch.as_tensor(_DevArray(ptr, cnt), device=f"cuda:{self.device}") if cnt > 0 else \
                torch.zeros(0, dtype=torch.float64, device=f"cuda:{self.device}")
        return self._top

    def num_segments(self):
        return self.plan.num_segments()

    def segment_tensors(self, k):
        """tensors to sum over the ranks before factorize_segment(k): ONE packed buffer per segment (the blocks' rows from
        their first column down; the structurally zero rows above stay off the wire), gathered on the plan's stream"""
        import torch
        ptr, cnt = self.plan.segment_pack(k)
        if cnt <= 0:
            return []
        if k not in self._seg or self._seg[k][0] != (ptr, cnt):
            self._seg[k] = ((ptr, cnt), torch.as_tensor(_DevArray(ptr, cnt), device=f"cuda:{self.device}"))
        return [self._seg[k][1]]

    def factorize_segment(self, k):
        self.plan.factorize_segment(k, sync=False)

    def finish(self):
        self.plan.sync()

    def get_factor(self, out=None):
        return self.plan.get_factor(out)

    def factorize_distributed(self):
        self.plan.factorize_distributed(self.comm, sync=False)

    def close(self):
        if self.comm is not None:
            self.comm.close()
        self.plan.close()


class ShardedFactorization:
    def __init__(self, sym, rank, world, device=0, engine_factory=None, group=None, mode="distributed"):
        if mode not in ("distributed", "replicated"):
            raise ValueError("mode must be 'distributed' or 'replicated'")
        self.sym, self.rank, self.world, self.group = sym, rank, world, group
        import os
        self.mode = mode if (world > 1 or os.environ.get("SF_FORCE_DISTRIBUTED") == "1") else "replicated"
        self.top_weight = (1.0 / world + TOP_CHAIN_SHARE) if self.mode == "distributed" else 1.0
        self.owner, self.top_fraction, self.max_load_fraction = subtree_partition(sym, world, self.top_weight)
        phase = phases_for_rank(self.owner, rank)
        dist_mode = self.mode == "distributed"
        factory = engine_factory or (lambda s, ph, lt, r, w, d: HipEngine(s, ph, lt, device, r, w, d, group, self.owner))
        self.engine = factory(sym, phase, rank == 0, rank, world, dist_mode)
        self.phase = phase

    def plan_info(self):
        tf, ml = self.top_fraction, self.max_load_fraction
        cost = self.top_weight * tf + ml
        return {"mode": self.mode, "collectives": getattr(self.engine, "comm_kind", "engine"),
                "subtrees_per_rank": [int(np.count_nonzero(self.owner == r)) for r in range(self.world)],
                "top_supernodes": int(np.count_nonzero(self.owner < 0)),
                "top_flop_fraction": tf, "max_rank_subtree_flop_fraction": ml,
                "segments": self.engine.num_segments() if self.mode == "distributed" else 0,
                "model_speedup_bound": 1.0 / cost if cost > 0 else float(self.world)}

    def set_values(self, Lx, Ux=None):
        if Ux is None:
            self.engine.set_values(Lx)
        else:
            self.engine.set_values(Lx, Ux)

    def factorize(self):
        import torch.distributed as dist
        eng = self.engine
        if self.mode == "distributed" and getattr(eng, "comm", None) is not None:
            # C driver + RCCL: sf_chol_plan_factorize_distributed runs phase 0 AND the segments; asynchronous, finish() waits
            eng.factorize_distributed()
            eng.finish()
            return
        eng.factorize_phase(0)
        if self.mode == "distributed":
            for k in range(eng.num_segments()):
                for t in eng.segment_tensors(k):
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                eng.factorize_segment(k)
            eng.finish()
            return
        if self.world > 1:
            top = eng.top_tensor()
            if top.numel() > 0:
                dist.all_reduce(top, op=dist.ReduceOp.SUM, group=self.group)
                if top.is_cuda:
                    import torch
                    torch.cuda.synchronize(top.device)
        eng.factorize_phase(1)

    def solve(self, b):
        """x = A^{-1} b (permuted numbering) with the factor left where the factorization put it: the C driver's distributed solve
        (one small sum per shared supernode inside the library), then the ranks' parts of x are merged with one all-reduce"""
        import torch
        import torch.distributed as dist
        eng = self.engine
        if getattr(eng, "comm", None) is None:
            if self.world > 1:
                raise RuntimeError("the distributed solve needs the C-side communicator (nccl backend)")
            return eng.plan.solve(b)
        x = eng.plan.solve_distributed(eng.comm, b)
        if self.world > 1:
            t = torch.from_numpy(x).cuda(eng.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            x = t.cpu().numpy()
        return x

    def gather_factor(self):
        """full factor in the reference layout on every rank (all-reduce of the disjoint subtree panels; the
        top panels, identical on every rank, are taken from this rank).  Test / validation helper, not part of the
        timed path."""
        import torch
        import torch.distributed as dist
        mine = self.engine.get_factor()
        if self.world == 1:
            return mine
        S = self.sym
        sub = np.zeros_like(mine)
        for s in np.flatnonzero(self.phase == 0):
            sub[S.Lsxp[s]:S.Lsxp[s + 1]] = mine[S.Lsxp[s]:S.Lsxp[s + 1]]
        # a top panel is identical on the ranks of its group: the group's first rank contributes it (a rank outside the
        # group does not hold it under the proportional mapping)
        masks = top_groups(S, self.owner)
        for s in np.flatnonzero(self.owner < 0):
            m = int(masks[s])
            if m and (m & -m) == (1 << self.rank):
                sub[S.Lsxp[s]:S.Lsxp[s + 1]] = mine[S.Lsxp[s]:S.Lsxp[s + 1]]
        t = torch.from_numpy(sub)
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def close(self):
        self.engine.close()


#: the Cholesky name the rest of the package and the tests use; LU goes through the same class (sym.lu selects the plan)
ShardedCholesky = ShardedFactorization

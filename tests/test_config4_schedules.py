"""BASELINE config 4 (3-D Laplacian 256^3 over 8 GPUs) checked WITHOUT eight GPUs: every rank's proportionally mapped plan is
built schedule-only (sf_chol_plan_schedule_mapped: the code path of sf_chol_plan_create_mapped up to the uploads, no device,
nothing allocated) and the W plans are checked against each other for what a real RCCL run depends on:

  (i)   for every group of ranks, all members issue the same sequence of collectives -- same order, same element count
        (packed doubles and the per-region counts of sf_chol_plan_segment_regions), same stream role (look-ahead or in line) --
        and the ranks' sequences embed in ONE global order (no circular wait between groups);
  (ii)  inside a segment all members run the same launches, and the unit windows [lo, hi) of every split launch partition the
        launch's units exactly once over the group, in group-index order;
  (iii) every supernode's panel is stored by exactly its owner (subtrees) or exactly its group (top);
  (iv)  the bytes a rank would allocate stay inside a stated budget;
  (v)   the sums of the distributed forward solve agree the same way.

The reference has no counterpart: all its handlers take tasks from one host queue (Cholesky/Source/SparseFrame.c:2267) and
exchange panels through host memory, with the accumulate-then-subtract algebra of C:2505-2508 / C:2748 that the segments keep.

CPU part: small grids (Cholesky and LU) and config 4 itself.  GPU part (`-m gpu`): the other two sizes `bench.py --gpus N` runs
(161^3 / 2, 203^3 / 4), a real device plan's tables against the schedule-only ones (they are the same code, this pins it), and
eight emulated handlers on the largest grid whose eight plans fit one GPU together."""
import ctypes as C
import importlib

import numpy as np
import pytest

from util import sf, gen

sharded = importlib.import_module("sparse-matrix-factorization-library_amd.sharded")
lib = importlib.import_module("sparse-matrix-factorization-library_amd._lib").lib

F = {name: i for i, name in enumerate(sf.Schedule.LAUNCH_FIELDS)}
G = {name: i for i, name in enumerate(sf.Schedule.SEGMENT_FIELDS)}


def popcount(m):
    return bin(int(m)).count("1")


def group_index(m, r):
    return popcount(int(m) & ((1 << r) - 1))


def expected_groups(sym, owner):
    """mask of the ranks that must store supernode s: its owner, or for a top supernode the ranks owning a subtree below it"""
    ns = sym.nsuper
    Super, Lsip = np.asarray(sym.Super), np.asarray(sym.Lsip)
    nscol, nsrow = np.diff(Super), np.diff(Lsip)
    has_parent = nscol < nsrow
    first_below = np.asarray(sym.Lsi)[np.where(has_parent, Lsip[:-1] + nscol, 0)]
    parent = np.where(has_parent, np.asarray(sym.SuperMap)[first_below], -1).tolist()
    own = np.asarray(owner).tolist()
    below = [(1 << o) if o >= 0 else 0 for o in own]
    for s in range(ns):                      # parents follow their children (postorder)
        if parent[s] >= 0:
            below[parent[s]] |= below[s]
    return [(1 << own[s]) if own[s] >= 0 else below[s] for s in range(ns)]


def embeds_in_one_order(sequences):
    """the ranks' sequences of collective ids are all sub-sequences of one total order (Kahn's algorithm on the union of their
    consecutive pairs): nobody can wait in a collective for a rank that waits, directly or not, for it"""
    succ, indeg = {}, {}
    for seq in sequences:
        for a in seq:
            indeg.setdefault(a, 0)
            succ.setdefault(a, set())
        for a, b in zip(seq, seq[1:]):
            if b not in succ[a]:
                succ[a].add(b)
                indeg[b] += 1
    ready = [a for a, d in indeg.items() if d == 0]
    done = 0
    while ready:
        a = ready.pop()
        done += 1
        for b in succ[a]:
            indeg[b] -= 1
            if indeg[b] == 0:
                ready.append(b)
    return done == len(indeg)


def check_schedules(sym, owner, W, lu=False, budget_bytes=None, plans=None):
    """builds (or takes) the W plans and asserts (i) .. (v); returns a summary dict"""
    own_plans = plans is None
    if plans is None:
        plans = [sf.Schedule(sym, owner, r, W, lu=lu) for r in range(W)]
    try:
        seg = [p.segment_table() for p in plans]
        lau = [p.launch_table() for p in plans]
        red = [p.solve_reduce_table() for p in plans]
        regions = [[tuple(c for (_, c) in p.segment_regions(k)) for k in range(len(seg[r]))] for r, p in enumerate(plans)]
        masks = sorted({int(m) for r in range(W) for m in seg[r][:, G["mask"]]})
        all_ranks = (1 << W) - 1
        n_collectives = n_split = 0
        # ---- (i) + (ii): group by group
        for m in masks:
            members = [r for r in range(W) if (m >> r) & 1]
            assert m & ~all_ranks == 0 and members
            for r in range(W):
                rows = np.nonzero(seg[r][:, G["mask"]] == m)[0]
                if r not in members:
                    assert len(rows) == 0, f"rank {r} lists a collective of group {m:#x} it does not belong to"
            per_member = {r: np.nonzero(seg[r][:, G["mask"]] == m)[0] for r in members}
            r0 = members[0]
            for r in members[1:]:
                assert len(per_member[r]) == len(per_member[r0]), f"group {m:#x}: ranks {r0} and {r} issue a different number of collectives"
            n_collectives += len(per_member[r0])
            for j in range(len(per_member[r0])):
                k0 = per_member[r0][j]
                l0a, l1a = seg[r0][k0, G["l0"]], seg[r0][k0, G["l1"]]
                for r in members[1:]:
                    k = per_member[r][j]
                    for fld in ("packed", "early", "regions"):
                        assert seg[r][k, G[fld]] == seg[r0][k0, G[fld]], f"group {m:#x} collective {j}: {fld} differs between ranks {r0} and {r}"
                    assert regions[r][k] == regions[r0][k0], f"group {m:#x} collective {j}: region sizes differ between ranks {r0} and {r}"
                    l0b, l1b = seg[r][k, G["l0"]], seg[r][k, G["l1"]]
                    assert l1b - l0b == l1a - l0a, f"group {m:#x} segment {j}: ranks {r0} and {r} run a different number of launches"
                    same = ("kind", "tasks", "items", "split", "replicated")
                    a = lau[r0][l0a:l1a][:, [F[f] for f in same]]
                    b = lau[r][l0b:l1b][:, [F[f] for f in same]]
                    assert np.array_equal(a, b), f"group {m:#x} segment {j}: launches differ between ranks {r0} and {r}"
                # the launches of the segment carry its index
                for r in members:
                    k = per_member[r][j]
                    assert np.all(lau[r][seg[r][k, G["l0"]]:seg[r][k, G["l1"]], F["segment"]] == k)
                # split launches: the members' windows tile [0, items) once, in group-index order
                split_rows = np.nonzero(lau[r0][l0a:l1a, F["split"]] == 1)[0]
                n_split += len(split_rows)
                if len(split_rows):
                    items = lau[r0][l0a:l1a][split_rows, F["items"]]
                    edge = np.zeros(len(split_rows), dtype=np.int64)
                    for gi, r in enumerate(members):
                        k = per_member[r][j]
                        rows = lau[r][seg[r][k, G["l0"]]:seg[r][k, G["l1"]]][split_rows]
                        assert np.all(rows[:, F["group_index"]] == gi) and np.all(rows[:, F["group_size"]] == len(members))
                        assert np.array_equal(rows[:, F["lo"]], edge), f"group {m:#x} segment {j}: rank {r}'s window does not start where rank {members[gi - 1]}'s ends"
                        assert np.all(rows[:, F["hi"]] >= rows[:, F["lo"]])
                        edge = rows[:, F["hi"]].copy()
                    assert np.array_equal(edge, items), f"group {m:#x} segment {j}: the windows do not cover the launch"
                assert popcount(m) > 1 or len(split_rows) == 0
        for r in range(W):
            # own subtrees: nothing shared, nothing split, no segment; every top launch inside a segment
            own = lau[r][lau[r][:, F["segment"]] < 0]
            assert np.all(own[:, F["split"]] == 0) and np.all(own[:, F["replicated"]] == 0)
            if len(seg[r]):
                assert seg[r][0, G["l0"]] == len(own) and seg[r][-1, G["l1"]] == len(lau[r])
                assert np.array_equal(seg[r][1:, G["l0"]], seg[r][:-1, G["l1"]])
            # a look-ahead segment follows a segment of its own group (its sum is issued while that one's chain runs)
            early = np.nonzero(seg[r][:, G["early"]] == 1)[0]
            assert np.all(early > 0) and np.all(seg[r][early, G["mask"]] == seg[r][early - 1, G["mask"]])
        sequences = []
        for r in range(W):
            count = {}
            seq = []
            for m in seg[r][:, G["mask"]]:
                seq.append((int(m), count.get(int(m), 0)))
                count[int(m)] = count.get(int(m), 0) + 1
            sequences.append(seq)
        assert embeds_in_one_order(sequences), "the ranks' collective sequences do not embed in one global order"
        # ---- (v) the forward solve's sums
        solve_seqs = []
        for m in sorted({int(x) for r in range(W) for x in red[r][:, 0]}):
            members = [r for r in range(W) if (m >> r) & 1]
            ref = red[members[0]][red[members[0]][:, 0] == m][:, 1:]
            for r in range(W):
                mine = red[r][red[r][:, 0] == m][:, 1:]
                if r in members:
                    assert np.array_equal(mine, ref), f"solve: group {m:#x} sums differ between ranks {members[0]} and {r}"
                else:
                    assert len(mine) == 0
        for r in range(W):
            solve_seqs.append([(int(a), int(b)) for a, b in red[r][:, :2]])
        assert embeds_in_one_order(solve_seqs)
        # ---- (iii) storage
        want = np.array(expected_groups(sym, owner), dtype=np.int64)
        stored = np.zeros(sym.nsuper, dtype=np.int64)
        for r, p in enumerate(plans):
            stored |= (p.panel_offsets(sym.nsuper) >= 0).astype(np.int64) << r
        bad = np.nonzero(stored != want)[0]
        assert len(bad) == 0, f"supernode {bad[:5]}: stored by {stored[bad[:5]]}, its group is {want[bad[:5]]}"
        # the masks the segments use are the groups of the top supernodes
        top_masks = {int(x) for x in want[np.asarray(owner) < 0]}
        assert set(masks) <= top_masks
        # ---- (iv) bytes
        bytes_dev = [p.stat("bytes_device") for p in plans]
        if budget_bytes is not None:
            assert max(bytes_dev) <= budget_bytes, f"a rank would allocate {max(bytes_dev) / 1e9:.1f} GB, budget {budget_bytes / 1e9:.1f} GB"
        return {"groups": len(masks), "collectives": n_collectives, "split_launches": n_split,
                "max_bytes": max(bytes_dev), "launches": [len(x) for x in lau]}
    finally:
        if own_plans:
            for p in plans:
                p.close()


def _laplacian_case(g, W):
    n, Cp, Ci, Cx = gen.laplacian_lower(g, g, g)
    slot = int(lib.sf_reference_slot_size(W, 288 << 30))         # the reference's devSlotSize for W 288 GiB devices (C:82-87, C:199)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(g, g, g, 3, 1), slot)
    owner, _, _ = sf.subtree_partition(sym, W, 1.0 / W + sharded.TOP_CHAIN_SHARE)
    return sym, owner, slot


@pytest.mark.parametrize("g,W", [(24, 2), (34, 3), (40, 4), (48, 8), (64, 8)])
def test_small_grids_all_ranks_agree(g, W):
    sym, owner, _ = _laplacian_case(g, W)
    out = check_schedules(sym, owner, W)
    assert out["collectives"] > 0 and out["split_launches"] > 0


def test_more_ranks_than_subtrees_and_a_forest():
    """7 ranks on a small tree (some ranks own nothing), and a block-diagonal matrix (a forest: no top at all)"""
    sym, owner, _ = _laplacian_case(12, 7)
    check_schedules(sym, owner, 7)
    n1, Cp1, Ci1, Cx1 = gen.laplacian_lower(6, 6, 6)
    Cp = np.concatenate([Cp1, Cp1[1:] + Cp1[-1], Cp1[1:] + 2 * Cp1[-1]])
    Ci = np.concatenate([Ci1, Ci1 + n1, Ci1 + 2 * n1])
    Cx = np.concatenate([Cx1, Cx1, Cx1])
    sym = sf.analyze(3 * n1, Cp, Ci, Cx, None, 1 << 30)
    owner, _, _ = sf.subtree_partition(sym, 3, 0.5)
    check_schedules(sym, owner, 3)


@pytest.mark.parametrize("N,W", [(14, 2), (24, 4)])
def test_lu_all_ranks_agree(N, W):
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=5)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N), 8 << 30, method="lu", symmetric=False)
    owner, _, _ = sf.subtree_partition(sym, W, 1.0 / W + sharded.TOP_CHAIN_SHARE)
    out = check_schedules(sym, owner, W, lu=True)
    assert out["collectives"] > 0


def test_the_checker_notices_a_rank_that_disagrees(monkeypatch):
    """negative controls: one rank built with the older in-line schedule (no look-ahead: other early flags, other launch lists), one
    rank built from another owner map (stores the wrong panels), one rank whose windows were cut with other shares"""
    sym, owner, _ = _laplacian_case(40, 4)

    def plans_with(r_bad, make):
        return [make() if r == r_bad else sf.Schedule(sym, owner, r, 4) for r in range(4)]

    def fails(plans):
        try:
            with pytest.raises(AssertionError):
                check_schedules(sym, owner, 4, plans=plans)
        finally:
            for p in plans:
                p.close()

    check_schedules(sym, owner, 4)
    monkeypatch.setenv("SF_LOOKAHEAD", "0")
    bad = sf.Schedule(sym, owner, 1, 4)
    monkeypatch.delenv("SF_LOOKAHEAD")
    fails(plans_with(1, lambda: bad))
    swapped = np.where(owner == 2, 3, np.where(owner == 3, 2, owner)).astype(np.int32)
    fails(plans_with(2, lambda: sf.Schedule(sym, swapped, 2, 4)))
    fails(plans_with(3, lambda: sf.Schedule(sym, owner, 2, 4)))            # rank 3 running rank 2's plan: windows overlap, panels missing


def test_owner_computes_prototype_all_ranks_agree(monkeypatch):
    """SF_TOP_OWNER=1: every member of the root's group names the same owner and the same owner's part for every block; the
    owners rotate over the group; the cross-rank invariants (i) .. (v) hold for this schedule too"""
    monkeypatch.setenv("SF_TOP_OWNER", "1")
    sym, owner, _ = _laplacian_case(64, 8)
    plans = [sf.Schedule(sym, owner, r, 8) for r in range(8)]
    try:
        tabs = [np.c_[p.segment_table()[:, G["mask"]], p.segment_owner_table()] for p in plans]
        root = [t[t[:, 0] == 255][:, 1:] for t in tabs]
        assert len(root[0]) >= 8 and all(np.array_equal(root[0], r) for r in root[1:])
        assert np.array_equal(root[0][:, 0], np.arange(len(root[0])) % 8) and np.all(root[0][:, 1] > 0)
        for t in tabs:
            assert np.all(t[t[:, 0] != 255][:, 1] == -1)            # smaller groups keep the replicated schedule
        check_schedules(sym, owner, 8, plans=plans)
    finally:
        for p in plans:
            p.close()


def test_a_schedule_only_plan_cannot_compute():
    sym, owner, _ = _laplacian_case(12, 2)
    sc = sf.Schedule(sym, owner, 0, 2)
    assert lib.sf_chol_plan_set_values(sc._h, sym.Lx.ctypes.data_as(C.POINTER(C.c_double))) == 1        # SF_ERR_ARG
    assert lib.sf_chol_plan_factorize(sc._h, 1) == 1
    assert lib.sf_chol_plan_sync(sc._h) == 1
    sc.close()


# bytes one rank may allocate at 256^3 / 8 (factor panels + task tables + staging): 121.6 GB today, of 288 GB per device
CONFIG4_BUDGET = 125e9


def test_config4_256cubed_over_8_all_ranks_agree():
    """BASELINE config 4 at its real size, devSlotSize 34,781,265,920 (the reference formula at 8 devices)"""
    sym, owner, slot = _laplacian_case(256, 8)
    assert slot == 34781265920
    out = check_schedules(sym, owner, 8, budget_bytes=CONFIG4_BUDGET)
    assert out["groups"] >= 7            # the root's group, two groups of four, four pairs (more when the tree is lopsided)
    assert out["collectives"] >= 200 and out["split_launches"] >= 200
    print("config 4:", out)


@pytest.mark.gpu
@pytest.mark.parametrize("g,W,budget", [(161, 2, 50e9), (203, 4, 88e9), (256, 8, CONFIG4_BUDGET)])
def test_bench_default_sizes_all_ranks_agree(g, W, budget):
    """the three multi-GPU sizes `bench.py --gpus N` runs by default (rows per GPU fixed): 161^3 / 2, 203^3 / 4, 256^3 / 8"""
    sym, owner, _ = _laplacian_case(g, W)
    print(f"{g}^3 / {W}:", check_schedules(sym, owner, W, budget_bytes=budget))


@pytest.mark.gpu
@pytest.mark.parametrize("g,W", [(40, 4), (64, 8)])
def test_device_plans_have_the_schedule_only_tables(g, W):
    """a real device plan (sf_chol_plan_create_mapped) reports exactly the tables of its schedule-only twin, including the byte count"""
    sym, owner, _ = _laplacian_case(g, W)
    real = [sf.CholPlan(sym, device=0, owner=owner, rank=r, nranks=W) for r in range(W)]
    try:
        for r in range(W):
            dry = sf.Schedule(sym, owner, r, W)
            assert np.array_equal(dry.launch_table(), real[r].launch_table())
            assert np.array_equal(dry.segment_table(), real[r].segment_table())
            assert np.array_equal(dry.solve_reduce_table(), real[r].solve_reduce_table())
            assert np.array_equal(dry.panel_offsets(sym.nsuper), real[r].panel_offsets(sym.nsuper))
            assert dry.stat("bytes_device") == real[r].stat("bytes_device")
            dry.close()
        check_schedules(sym, owner, W, plans=real)
    finally:
        for p in real:
            p.close()


@pytest.mark.gpu
def test_eight_emulated_handlers_at_the_largest_grid_one_gpu_holds(monkeypatch):
    """144^3 over eight emulated handlers with the 8-device devSlotSize: 131 GB of the eight ranks' plans together on the one
    device, the real per-rank plans, segment loop, look-ahead and dealt-out copy-back, the all-reduce a kernel (k_sum_ranks) instead
    of RCCL.  Checked: every entry of Lsx written, shared panels bit-identical on the ranks that hold them, the verified resident
    solve accepted (every panel's fingerprint on every rank that stores it against the host copy), residual of the known answer."""
    if sf.device_count() != 1:
        pytest.skip("emulated handlers are for one-GPU boxes")
    nh, N = 8, 144
    monkeypatch.setenv("SF_EMULATE_HANDLERS", str(nh))
    slot = int(lib.sf_reference_slot_size(nh, 288 << 30))
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    common = sf.CommonInfo(dev_slot_size=slot)
    assert common.c.numGPU == nh
    mi = sf.MatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx)
    mi.set_perm(sf.grid_nd_perm(N, N, N, 3, 1))
    mi.analyze(common)
    xsize = int(mi.c.xsize)
    C.memset(mi.c.Lsx, 0xff, 8 * xsize)
    mi.factorize(common)
    assert lib.sf_handlers_replica_mismatches(mi.c.Lsx) == 0
    Lsx = mi.array("Lsx", xsize)
    nan_chunks = sum(int(np.isnan(Lsx[o:o + (1 << 26)]).sum()) for o in range(0, xsize, 1 << 26))
    assert nan_chunks == 0
    k, fb = lib.sf_handlers_resident_solves(), lib.sf_handlers_fingerprint_fallbacks()
    assert mi.validate() <= 1e-13
    assert lib.sf_handlers_resident_solves() == k + 1 and lib.sf_handlers_fingerprint_fallbacks() == fb
    mi.cleanup()
    common.close()

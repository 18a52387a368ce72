"""A factor larger than the device budget (the reference's answer: slot-sized stages streamed through the device, the factor on the
host, C:1721-1846 / C:2421-2467; here: whole subtree groups through two alternating buffers under a resident top -- DESIGN 7b).
CPU: the grouping and the schedule of an out-of-core plan.  GPU: its factor against the oracle, through the flat C ABI and through
SparseFrame_factorize with a lowered budget (SF_DEVICE_BUDGET_MB)."""
import ctypes as C

import numpy as np
import pytest

from util import sf, gen, nd_perm_py, rel_err

TOL_FACTOR = 1e-12
TOL_RESIDUAL = 1e-13


def panel_entries(S):
    return np.diff(S.Super) * np.diff(S.Lsip)


def parents(S):
    ns = int(S.nsuper)
    par = np.full(ns, -1, dtype=np.int64)
    nscol, nsrow = np.diff(S.Super), np.diff(S.Lsip)
    for s in range(ns):
        if nscol[s] < nsrow[s]:
            par[s] = S.SuperMap[S.Lsi[S.Lsip[s] + nscol[s]]]
    return par


def chol_cases():
    out = []
    N = 20
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    out.append(("lap3d_20", n, Cp, Ci, Cx, nd_perm_py(N, N, N), 1 << 30))
    n, Cp, Ci, Cx = gen.stencil_spd_lower(90, 90)
    out.append(("stencil2d_90", n, Cp, Ci, Cx, sf.grid_nd_perm(90, 90, 1, 3, 2), sf.REFERENCE_SLOT_1GPU))
    # a forest: three independent grids (no top at all above the trees' roots)
    n1, Cp1, Ci1, Cx1 = gen.laplacian_lower(9, 9, 9)
    k = 3
    Cp = np.concatenate([[0]] + [Cp1[1:] + i * Cp1[-1] for i in range(k)]).astype(np.int64)
    Ci = np.concatenate([Ci1 + i * n1 for i in range(k)]).astype(np.int64)
    Cx = np.concatenate([Cx1 * (1 + i) for i in range(k)])
    p1 = np.asarray(nd_perm_py(9, 9, 9))
    out.append(("forest_3x9cubed", k * n1, Cp, Ci, Cx, np.concatenate([p1 + i * n1 for i in range(k)]), 1 << 30))
    return out


@pytest.mark.parametrize("case", chol_cases(), ids=lambda c: c[0])
def test_grouping_properties(case):
    """groups are whole subtrees, consecutive in the postorder; the top is closed upwards; the reported need is top + 2 x largest
    group and it fits the budget whenever the call says so; a budget that holds everything gives one group"""
    name, n, Cp, Ci, Cx, perm, slot = case
    S = sf.analyze(n, Cp, Ci, Cx, perm, slot)
    ent, par = panel_entries(S), parents(S)
    total = int(ent.sum())
    g, ng, ge, te, nd, fits = sf.ooc_partition(S, total)
    assert fits and ng == 1 and te == 0 and nd == total and (g == 0).all()
    seen_cut = False
    for frac in (0.9, 0.75, 0.6, 0.45, 0.3):
        cut = sf.ooc_partition(S, int(total * frac))
        g, ng, ge, te, nd, fits = cut
        assert ng >= 2
        seen_cut = True
        top = g < 0
        # (top mode 0: every top panel resident; 1 / 2 -- only chosen when that does not fit: the arena of the active ones)
        assert int(ent[top].sum()) == te if cut.top_mode == 0 else te <= int(ent[top].sum())
        sizes = np.bincount(g[~top], weights=ent[~top], minlength=ng).astype(np.int64)
        assert sizes.max() == ge and nd == te + 2 * ge
        assert (not fits) or nd <= int(total * frac)
        for s in range(int(S.nsuper)):
            if par[s] >= 0:
                assert top[par[s]] or g[par[s]] == g[s], "a group is a union of whole subtrees"
                assert not (top[s] and not top[par[s]]), "the top is closed upwards"
        # consecutive in the postorder: group numbers never decrease along the supernode order
        gg = g[~top]
        assert (np.diff(gg) >= 0).all()
        assert set(np.unique(gg)) == set(range(ng))
    assert seen_cut


@pytest.mark.parametrize("case", chol_cases(), ids=lambda c: c[0])
def test_schedule_of_an_out_of_core_plan(case):
    """without a device: one (zero + assemble) launch per group, in group order, each before the first launch that touches the group;
    the groups' panels alias two buffers of the largest group's size, the top panels follow; the plan's byte count is what the
    partition promised (+ the plan's tables)"""
    name, n, Cp, Ci, Cx, perm, slot = case
    S = sf.analyze(n, Cp, Ci, Cx, perm, slot)
    ent = panel_entries(S)
    total = int(ent.sum())
    g, ng, ge, te, nd, fits = sf.ooc_partition(S, int(total * 0.6))
    sch = sf.Schedule(S, None, 0, 1, ooc_group=g, ooc_ngroups=ng)
    lt = sch.launch_table()
    k7 = np.nonzero(lt[:, 0] == 7)[0]
    assert len(k7) == ng
    xp = sch.panel_offsets(S.nsuper)
    for s in range(int(S.nsuper)):
        if g[s] >= 0:
            b = (g[s] & 1) * ge
            assert b <= xp[s] and xp[s] + ent[s] <= b + ge
        else:
            assert 2 * ge <= xp[s] and xp[s] + ent[s] <= 2 * ge + te
    # inside one group (and inside the top) panels do not overlap
    for grp in list(range(ng)) + [-1]:
        idx = np.nonzero(g == grp)[0]
        o = np.argsort(xp[idx])
        a, e = xp[idx][o], ent[idx][o]
        assert (a[1:] >= a[:-1] + e[:-1]).all()
    in_core = sf.Schedule(S, None, 0, 1, ooc_group=np.zeros(S.nsuper, dtype=np.int32), ooc_ngroups=1)
    saved = in_core.stat("bytes_device") - sch.stat("bytes_device")
    # (one assembly mask per group is the only table an out-of-core plan adds; it has no solve schedule)
    assert saved >= 8 * (total - nd) - (ng + 1) * int(S.nsuper) - 64
    sch.close()
    in_core.close()


# ------------------------------------------------------------------ GPU

@pytest.mark.gpu
@pytest.mark.parametrize("frac", [0.75, 0.45])
@pytest.mark.parametrize("case", chol_cases(), ids=lambda c: c[0])
def test_out_of_core_cholesky_matches_the_oracle(oracle, case, frac):
    name, n, Cp, Ci, Cx, perm, slot = case
    S = sf.analyze(n, Cp, Ci, Cx, perm, slot)
    total = int(panel_entries(S).sum())
    g, ng, ge, te, nd, fits = sf.ooc_partition(S, int(total * frac))
    assert ng >= 2
    plan = sf.CholPlan(S, ooc_group=g, ooc_ngroups=ng)
    assert plan.stat("bytes_device") < 8 * total + 64 * len(S.Lsi) or nd >= total      # the factor is not resident as a whole
    ref, info, _ = oracle.chol_factorize(S)
    assert info == 0
    mask = oracle.lower_mask(S)
    for rep in range(2):                            # the second run finds both buffers used
        out = np.full(S.xsize, np.nan)
        plan.factorize_to_host(S.Lx, out=out)
        assert not np.isnan(out[mask]).any()
        assert rel_err(out, ref, mask) <= TOL_FACTOR
        res, _ = oracle.chol_residual(S, np.where(mask, out, 0.0))
        assert res <= TOL_RESIDUAL
    # the factor never exists on the device as a whole: everything that needs it there refuses
    with pytest.raises(sf.SparseFrameError, match="SF_ERR_ARG"):
        plan.factorize()
    with pytest.raises(sf.SparseFrameError, match="SF_ERR_ARG"):
        plan.solve(np.ones(n))
    with pytest.raises(sf.SparseFrameError, match="SF_ERR_ARG"):
        plan.get_factor()
    plan.close()


def test_active_top_layout_never_shares_a_place_between_two_live_panels():
    """top modes 1 / 2 (schedule-only plans): two top panels whose places in the arena overlap are never active at the same time --
    mode 1: the second one starts at least two groups after the first one's last group, mode 2: at least one; the arena is what the
    partition promised and, in mode 2, no larger than the largest set of simultaneously active panels would need with perfect packing
    x 1.5; every group's first launch comes before the launches of the top supernodes that start with it"""
    N = 24
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    S = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(N, N, N), 1 << 30)
    ent, par = panel_entries(S), parents(S)
    total = int(ent.sum())
    cuts = {}
    for frac in np.arange(0.60, 0.30, -0.01):         # the first budget at which each mode is the partition's answer
        cut = sf.ooc_partition(S, int(total * frac))
        if cut[5] and cut.top_mode not in cuts:
            cuts[cut.top_mode] = cut
    assert set(cuts) == {0, 1, 2}, sorted(cuts)
    for want_mode in (1, 2):
        cut = cuts[want_mode]
        g, ng, ge, te, nd, fits = cut
        ns = int(S.nsuper)
        first, last, prev = np.full(ns, 10 ** 9), np.full(ns, -1), 0
        for s in range(ns):
            if g[s] >= 0:
                first[s] = last[s] = prev = g[s]
            elif last[s] < 0:
                first[s] = last[s] = prev
            if par[s] >= 0:
                first[par[s]], last[par[s]] = min(first[par[s]], first[s]), max(last[par[s]], last[s])
        sch = sf.Schedule(S, None, 0, 1, ooc_group=g, ooc_ngroups=ng, ooc_top_mode=cut.top_mode)
        xp = sch.panel_offsets(S.nsuper)
        tops = np.nonzero(g < 0)[0]
        lag = 2 if cut.top_mode == 1 else 1
        assert (xp[tops] >= 2 * ge).all() and int((xp[tops] + ent[tops]).max()) - 2 * ge == te
        for i, a in enumerate(tops):
            for b in tops[i + 1:]:
                if xp[a] < xp[b] + ent[b] and xp[b] < xp[a] + ent[a]:
                    assert first[b] >= last[a] + lag or first[a] >= last[b] + lag, (a, b)
        active = max(int(ent[tops][(first[tops] <= u) & (last[tops] + lag - 1 >= u)].sum()) for u in range(ng))
        assert active <= te <= 1.5 * active
        sch.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("case", chol_cases(), ids=lambda c: c[0])
def test_active_top_modes_match_the_oracle(oracle, case, mode, monkeypatch):
    """top panels resident only while active (mode 1), with their places re-used one group later and the waits that takes (mode 2):
    same factor; small staging slots so that every group and every top run has several pieces"""
    monkeypatch.setenv("SF_DL_SLOT_MB", "1")
    name, n, Cp, Ci, Cx, perm, slot = case
    S = sf.analyze(n, Cp, Ci, Cx, perm, slot)
    total = int(panel_entries(S).sum())
    g, ng, ge, te, nd, fits = sf.ooc_partition(S, int(total * 0.25))        # a fine cut (whatever mode it reports)
    assert ng >= 3
    plan = sf.CholPlan(S, ooc_group=g, ooc_ngroups=ng, ooc_top_mode=mode)
    ref, info, _ = oracle.chol_factorize(S)
    mask = oracle.lower_mask(S)
    for rep in range(2):
        out = np.full(S.xsize, np.nan)
        plan.factorize_to_host(S.Lx, out=out)
        assert not np.isnan(out[mask]).any()
        assert rel_err(out, ref, mask) <= TOL_FACTOR
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
def test_active_top_modes_lu_with_interchanges(oracle, mode):
    import golden_large
    c = golden_large.build_case("piv_zero_diag_12")
    n, S = c["n"], c["sym"]
    total = int(panel_entries(S).sum())
    g, ng, ge, te, nd, fits = sf.ooc_partition(S, int(total * 0.3))
    assert ng >= 3
    plan = sf.LUPlan(S, ooc_group=g, ooc_ngroups=ng, ooc_top_mode=mode)
    plan.set_pivoting(0.1)
    out = np.full(S.xsize, np.nan)
    plan.factorize_to_host(S.Lx, S.Ux, out=out)
    ref, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=0.1)
    assert info == 0 and np.array_equal(plan.get_pivots(), pivpos)
    assert not np.isnan(out).any() and rel_err(out, ref) <= 1e-10
    plan.close()


@pytest.mark.gpu
def test_out_of_core_with_many_small_groups(oracle, monkeypatch):
    """more groups than copy workers, 1 MiB staging slots (several pieces per group), a wide root: every buffer is re-used many
    times and the launch that re-uses it has to wait for the copy of the group before the last"""
    monkeypatch.setenv("SF_DL_SLOT_MB", "1")
    N = 26
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    S = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N), 4 << 30)
    assert np.diff(S.Super).max() > 512
    total = int(panel_entries(S).sum())
    g, ng, ge, te, nd, fits = sf.ooc_partition(S, int(total * 0.05))        # (does not "fit": the cheapest cut of the ladder, small groups)
    assert ng >= 24
    plan = sf.CholPlan(S, ooc_group=g, ooc_ngroups=ng)
    out = np.full(S.xsize, np.nan)
    plan.factorize_to_host(S.Lx, out=out)
    ref, info, _ = oracle.chol_factorize(S)
    mask = oracle.lower_mask(S)
    assert rel_err(out, ref, mask) <= TOL_FACTOR
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tol", [0.0, 0.1])
def test_out_of_core_lu_matches_the_oracle(oracle, tol):
    """LU: both panels of a group alias the two buffers (L and U^T halves); with threshold pivoting inside the blocks (the matrix
    of the large fixture whose permuted diagonal has zeros: rows really are interchanged)"""
    if tol > 0:
        import golden_large
        c = golden_large.build_case("piv_zero_diag_12")
        n, S = c["n"], c["sym"]
    else:
        N = 20          # (no random long-range entries: they leave no subtree to stream, the whole tree is "top")
        n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, extra_per_row=0, drop=0.2, seed=5)
        S = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N, 3, 2), 1 << 30, "lu", False)
    total = int(panel_entries(S).sum())
    g, ng, ge, te, nd, fits = sf.ooc_partition(S, int(total * 0.7))
    assert ng >= 2 and (tol > 0 or (fits and ng >= 6))
    plan = sf.LUPlan(S, ooc_group=g, ooc_ngroups=ng)
    if tol > 0:
        plan.set_pivoting(tol)
    out = np.full(S.xsize, np.nan)
    plan.factorize_to_host(S.Lx, S.Ux, out=out)
    assert not np.isnan(out).any()
    if tol > 0:
        ref, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=tol)
        assert info == 0 and np.array_equal(plan.get_pivots(), pivpos)
        assert np.count_nonzero(pivpos != np.arange(n)) > 0
    else:
        ref, info, _ = oracle.lu_factorize(S)
        assert info == 0
    assert rel_err(out, ref) <= (1e-10 if tol > 0 else TOL_FACTOR)
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("frac,mode", [(0.6, 0), (0.46, 1), (0.35, 2)])
def test_struct_entry_point_goes_out_of_core_under_a_budget(oracle, monkeypatch, capfd, frac, mode):
    """SparseFrame_factorize with a device budget smaller than the factor: same Lsx, plan cached per pattern, the solve answers from
    the host copy (nothing resident to solve with).  The tighter the budget, the further down the ladder: top panels resident
    (mode 0), resident while active (1), their places re-used at once (2)"""
    monkeypatch.setenv("SF_TRACE", "1")
    N = 32
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    perm = nd_perm_py(N, N, N)
    S = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30)
    ref, info, _ = oracle.chol_factorize(S)
    mask = oracle.lower_mask(S)
    total_mb = 8 * int(panel_entries(S).sum()) / 2**20
    # the budget counts the plan's tables and staging rings too (384 MiB + structure): leave the panels 60 % of their size
    overhead_mb = 384 + (12 * int(S.Lp[-1]) + 24 * len(S.Lsi)) / 2**20
    monkeypatch.setenv("SF_DEVICE_BUDGET_MB", str(int(overhead_mb + frac * total_mb) + 1))
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    builds0 = common.plan_builds()
    solves0 = sf.lib.sf_handlers_resident_solves()
    for scale in (1.0, 2.0):
        mi = sf.MatrixInfo()
        mi.set_csc(n, Cp, Ci, Cx * scale)
        mi.set_perm(perm)
        mi.analyze(common)
        C.memset(mi.c.Lsx, 0xff, 8 * S.xsize)
        mi.factorize(common)
        got = mi.array("Lsx", S.xsize).copy()
        assert not np.isnan(got[mask]).any()
        assert rel_err(got, ref * np.sqrt(scale), mask) <= TOL_FACTOR
        assert mi.validate() <= TOL_RESIDUAL
        mi.cleanup()
    assert common.plan_builds() == builds0 + 1                      # one out-of-core plan for both calls
    assert sf.lib.sf_handlers_resident_solves() == solves0          # host sweep: no resident factor
    assert f"(top mode {mode})" in capfd.readouterr().err
    common.close()


@pytest.mark.gpu
def test_lu_struct_entry_point_under_a_budget(oracle, monkeypatch):
    """the LU library's SparseFrame_factorize under a budget: packed (L \\ U) panels assembled by the copy workers from the two
    aliased halves, value by value against the oracle; the second call finds the cached out-of-core plan"""
    N = 20
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, extra_per_row=0, drop=0.2, seed=3)
    perm = sf.grid_nd_perm(N, N, N, 3, 2)
    S = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
    total_mb = 16 * int(panel_entries(S).sum()) / 2**20
    overhead_mb = 384 + (12 * int(S.Lp[-1]) + 12 * int(S.Up[-1]) + 24 * len(S.Lsi)) / 2**20
    monkeypatch.setenv("SF_DEVICE_BUDGET_MB", str(int(overhead_mb + 0.72 * total_mb)))
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    builds0 = common.plan_builds()
    for scale in (1.0, 3.0):
        mi = sf.LUMatrixInfo()
        mi.set_csc(n, Cp, Ci, Cx * scale, symmetric=False)
        mi.set_perm(perm)
        mi.analyze(common)
        C.memset(mi.c.Lsx, 0xff, 8 * S.xsize)
        mi.factorize(common)
        got = mi.array("Lsx", S.xsize).copy()
        assert not np.isnan(got).any()
        S2 = sf.analyze(n, Cp, Ci, Cx * scale, perm, 1 << 30, "lu", False)
        ref, info, _ = oracle.lu_factorize(S2)
        assert info == 0 and rel_err(got, ref) <= TOL_FACTOR
        assert mi.validate() <= TOL_RESIDUAL
        mi.cleanup()
    assert common.plan_builds() == builds0 + 1
    common.close()


@pytest.mark.gpu
def test_a_budget_below_the_resident_top_is_an_allocation_error(monkeypatch, capfd):
    """no cut of the tree fits: SparseFrame_factorize fails with SF_ERR_ALLOC and says what it would have taken; nothing is written,
    nothing crashes, and the same handler list factorizes the matrix once the budget is lifted"""
    N = 20
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    perm = nd_perm_py(N, N, N)
    S = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30)
    total_mb = 8 * int(panel_entries(S).sum()) / 2**20
    overhead_mb = 384 + (12 * int(S.Lp[-1]) + 24 * len(S.Lsi)) / 2**20
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    mi = sf.MatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx)
    mi.set_perm(perm)
    mi.analyze(common)
    monkeypatch.setenv("SF_DEVICE_BUDGET_MB", str(int(overhead_mb + 0.2 * total_mb)))
    with pytest.raises(sf.SparseFrameError):
        mi.factorize(common)
    assert "DOES NOT FIT" in capfd.readouterr().err
    monkeypatch.delenv("SF_DEVICE_BUDGET_MB")
    mi.factorize(common)
    assert mi.validate() <= TOL_RESIDUAL
    mi.cleanup()
    common.close()


def test_a_grouping_that_splits_a_subtree_is_refused():
    """foreign group arrays are checked, not trusted: a streamed supernode whose ancestor sits in ANOTHER group would update a panel
    that is not on the device any more (or not yet) -- plan creation refuses (no device needed: the schedule-only entry point)"""
    N = 12
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    S = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(N, N, N), 1 << 30)
    total = int(panel_entries(S).sum())
    g, ng, ge, te, nd, fits = sf.ooc_partition(S, int(total * 0.7))
    assert ng >= 2
    par = parents(S)
    bad = g.copy()
    s = next(s for s in range(int(S.nsuper)) if g[s] >= 0 and par[s] >= 0 and g[par[s]] == g[s])
    bad[s] = (g[s] + 1) % ng                       # a child in another group than its parent
    with pytest.raises(sf.SparseFrameError, match="SF_ERR_ARG"):
        sf.Schedule(S, None, 0, 1, ooc_group=bad, ooc_ngroups=ng)
    bad = g.copy()
    t = next(s for s in range(int(S.nsuper)) if g[s] < 0 and par[s] >= 0)
    bad[par[t]] = 0                                # a resident supernode below a streamed one
    with pytest.raises(sf.SparseFrameError, match="SF_ERR_ARG"):
        sf.Schedule(S, None, 0, 1, ooc_group=bad, ooc_ngroups=ng)


@pytest.mark.gpu
def test_struct_entry_point_goes_out_of_core_when_the_device_is_full(oracle, monkeypatch, capfd):
    """no budget variable: most of the device is taken by somebody else (a torch tensor), what is left holds about 60 % of the
    factor -- SparseFrame_factorize has to notice by itself (hipMemGetInfo), stream the factor and still deliver the same Lsx"""
    import torch
    monkeypatch.delenv("SF_DEVICE_BUDGET_MB", raising=False)
    monkeypatch.setenv("SF_DEVICE_POOL_MB", "0")        # (the handler's own pool would hold this small factor)
    monkeypatch.setenv("SF_TRACE", "1")
    N = 40
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    perm = sf.grid_nd_perm(N, N, N)
    S = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30)
    ref, info, _ = oracle.chol_factorize(S)
    mask = oracle.lower_mask(S)
    panels = 8 * int(panel_entries(S).sum())
    overhead = (384 << 20) + 12 * int(S.Lp[-1]) + 24 * len(S.Lsi)
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    torch.cuda.init()
    free, total = torch.cuda.mem_get_info(0)
    want_free = (1 << 30) + overhead + int(0.6 * panels)
    ballast = torch.empty(max(free - want_free, 0), dtype=torch.uint8, device="cuda:0")
    try:
        free2, _ = torch.cuda.mem_get_info(0)
        assert free2 < (1 << 30) + overhead + panels, (free2, want_free)        # the in-core plan can not fit
        mi = sf.MatrixInfo()
        mi.set_csc(n, Cp, Ci, Cx)
        mi.set_perm(perm)
        mi.analyze(common)
        C.memset(mi.c.Lsx, 0xff, 8 * S.xsize)
        mi.factorize(common)
        err = capfd.readouterr().err
        assert "out of core" in err and "DOES NOT FIT" not in err, err[-2000:]
        got = mi.array("Lsx", S.xsize).copy()
        assert rel_err(got, ref, mask) <= TOL_FACTOR
        assert mi.validate() <= TOL_RESIDUAL
        mi.cleanup()
    finally:
        del ballast
        torch.cuda.empty_cache()
        common.close()


@pytest.mark.gpu
def test_the_handler_pool_lends_its_buffer_to_one_pattern_at_a_time(oracle, monkeypatch):
    """SparseFrame_allocate_gpu allocates the device pool (as the reference allocates its slots there); the first pattern's plan
    places its factor in it, a second pattern cached beside it allocates its own, and when the first is evicted (third pattern: two
    plans are cached per handler) the pool is lent again.  Results are those of plans with their own buffers."""
    monkeypatch.setenv("SF_DEVICE_POOL_MB", "512")
    common = sf.CommonInfo(dev_slot_size=1 << 30)

    def pool():
        out = (C.c_int64 * 3)()
        assert sf.lib.sf_handlers_pool_info(common.gpu_list, 0, out) == 0
        return list(out)

    assert pool() == [512 << 20, 0, 0]
    users = []
    for N in (14, 16, 18, 14):
        n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
        perm = nd_perm_py(N, N, N)
        S = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30)
        ref, info, _ = oracle.chol_factorize(S)
        mi = sf.MatrixInfo()
        mi.set_csc(n, Cp, Ci, Cx)
        mi.set_perm(perm)
        mi.analyze(common)
        mi.factorize(common)
        assert rel_err(mi.array("Lsx", S.xsize).copy(), ref, oracle.lower_mask(S)) <= TOL_FACTOR
        assert mi.validate() <= TOL_RESIDUAL
        users.append(pool()[2])
        mi.cleanup()
    # pattern 1 borrows; pattern 2 finds the pool taken; pattern 3 evicts pattern 1 (least recently used) and borrows; pattern 4 (= 1
    # again, rebuilt) evicts pattern 2 and allocates its own: the pool stays with pattern 3
    assert users == [14 ** 3, 14 ** 3, 18 ** 3, 18 ** 3]
    common.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("seed,n_,per_col,band", [(1, 1500, 3, None), (3, 3000, 6, 400), (4, 6000, 4, 60), (6, 2500, 5, 1200)])
def test_out_of_core_on_random_patterns(oracle, monkeypatch, seed, n_, per_col, band, mode):
    """irregular trees (the built-in ordering on random SPD patterns: wide roots over heavy fill, banded chains, lopsided subtrees)
    through every top mode: whatever the cut looks like -- a chain that is all top, groups of very different sizes, top supernodes
    without grouped descendants -- the factor is the oracle's"""
    monkeypatch.setenv("SF_DL_SLOT_MB", "1")
    n, Cp, Ci, Cx = gen.random_spd_lower(n_, per_col, seed=seed, bandwidth=band)
    perm = sf.graph_nd_perm(n, Cp, Ci)
    S = sf.analyze(n, Cp, Ci, Cx, perm, sf.REFERENCE_SLOT_1GPU)
    total = int(panel_entries(S).sum())
    g, ng, ge, te, nd, fits = sf.ooc_partition(S, int(total * 0.3))
    if ng < 2:
        pytest.skip("this tree has nothing to stream")
    plan = sf.CholPlan(S, ooc_group=g, ooc_ngroups=ng, ooc_top_mode=mode)
    out = np.full(S.xsize, np.nan)
    plan.factorize_to_host(S.Lx, out=out)
    ref, info, _ = oracle.chol_factorize(S)
    assert info == 0
    mask = oracle.lower_mask(S)
    assert not np.isnan(out[mask]).any()
    assert rel_err(out, ref, mask) <= TOL_FACTOR
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_out_of_core_lu_general_matrix_with_partial_pivoting(oracle, mode):
    """a general unsymmetric matrix (no diagonal dominance) with partial pivoting inside the blocks (tol 1), streamed: the pivot
    sequence and the packed factor are the oracle's in every top mode"""
    n, Cp, Ci, Cx = gen.unsymmetric_general(14, 14, 14, seed=22, diag_scale=1.0)
    S = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(14, 14, 14), 1 << 30, "lu", False)
    total = int(panel_entries(S).sum())
    g, ng, ge, te, nd, fits = sf.ooc_partition(S, int(total * 0.3))
    if ng < 2:
        pytest.skip("this tree has nothing to stream")
    ref, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=0.3)
    assert info == 0 and np.count_nonzero(pivpos != np.arange(n)) > 0
    plan = sf.LUPlan(S, ooc_group=g, ooc_ngroups=ng, ooc_top_mode=mode)
    plan.set_pivoting(0.3)
    out = np.full(S.xsize, np.nan)
    plan.factorize_to_host(S.Lx, S.Ux, out=out)
    assert np.array_equal(plan.get_pivots(), pivpos)
    assert not np.isnan(out).any() and rel_err(out, ref) <= 1e-10
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 2])
def test_out_of_core_with_the_three_launch_step_form(oracle, monkeypatch, mode):
    """SF_FUSE_MAX=0: every 64-column step as stream-K GEMM + one-wave POTRF + TRSM launches instead of the fused step kernel -- the
    other form of the chain, streamed"""
    monkeypatch.setenv("SF_FUSE_MAX", "0")
    N = 20
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    S = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(N, N, N), 1 << 30)
    total = int(panel_entries(S).sum())
    g, ng, ge, te, nd, fits = sf.ooc_partition(S, int(total * 0.4))
    assert ng >= 3
    plan = sf.CholPlan(S, ooc_group=g, ooc_ngroups=ng, ooc_top_mode=mode)
    out = np.full(S.xsize, np.nan)
    plan.factorize_to_host(S.Lx, out=out)
    ref, info, _ = oracle.chol_factorize(S)
    mask = oracle.lower_mask(S)
    assert rel_err(out, ref, mask) <= TOL_FACTOR
    plan.close()

"""Built-in ordering (sf_graph_nd_perm): a self-contained stand-in for the reference's METIS call.  Orderings are
outside the parity contract; the tests check validity, usefulness (less fill than the identity) and that the whole
pipeline works with it (oracle factorization of the reordered matrix solves the system)."""
import numpy as np
import pytest

from util import sf, gen


@pytest.mark.parametrize("dims", [(30, 30, 1), (12, 12, 12), (50, 3, 1), (1, 1, 1)])
def test_graph_nd_is_a_permutation_and_reduces_fill(dims):
    n, Cp, Ci, Cx = gen.laplacian_lower(*dims)
    p = sf.graph_nd_perm(n, Cp, Ci)
    assert sorted(p.tolist()) == list(range(n))
    if n > 100:
        ident = sf.analyze(n, Cp, Ci, Cx, None, 8 << 30)
        nd = sf.analyze(n, Cp, Ci, Cx, p, 8 << 30)
        assert nd.flops_struct < ident.flops_struct


def test_graph_nd_disconnected_and_dense_and_diagonal():
    # many components (diagonal), a dense block, block-diagonal mix
    n = 5000
    p = sf.graph_nd_perm(n, np.arange(n + 1), np.arange(n))
    assert sorted(p.tolist()) == list(range(n))
    nd_ = 40
    rows, cols = np.tril_indices(nd_)
    order = np.lexsort((rows, cols))
    Cp = np.zeros(nd_ + 1, dtype=np.int64)
    np.add.at(Cp, cols + 1, 1)
    p = sf.graph_nd_perm(nd_, np.cumsum(Cp), rows[order])
    assert sorted(p.tolist()) == list(range(nd_))
    n, Cp, Ci, Cx = gen.random_spd_lower(3000, 3, seed=5)
    p = sf.graph_nd_perm(n, Cp, Ci, leaf=16)
    assert sorted(p.tolist()) == list(range(n))


def test_analyze_uses_builtin_ordering_when_none_is_supplied(oracle):
    n, Cp, Ci, Cx = gen.laplacian_lower(14, 14, 14)
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    mi = sf.MatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx)
    assert mi.c.permMethod == 2      # the default after initialize_matrix: ordered, as the reference (C:1937)
    mi.analyze(common)
    perm = mi.array("Perm", n).copy()
    assert sorted(perm.tolist()) == list(range(n))
    ident = sf.analyze(n, Cp, Ci, Cx, None, 1 << 30)
    assert mi.c.xsize < ident.xsize
    # the struct arrays are those of the flat analysis with the same ordering
    S = sf.analyze(n, Cp, Ci, Cx, sf.graph_nd_perm(n, Cp, Ci), 1 << 30)
    assert np.array_equal(perm, S.Perm) and mi.c.nsuper == S.nsuper
    Lsx, info, _ = oracle.chol_factorize(S)
    res, _ = oracle.chol_residual(S, Lsx)
    assert info == 0 and res <= 1e-13
    mi.cleanup()


def test_default_ordering_fill_is_close_to_geometric_nested_dissection():
    """reference call order (initialize, set matrix, analyze) with no ordering supplied must not factorize a 3-D problem in
    natural order: the built-in ordering's fill stays within 2x of the geometric nested dissection's, natural order is
    several times worse; set_perm(None) is the explicit natural-order opt-in"""
    N = 16
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    mi = sf.MatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx)
    mi.analyze(common)
    geo = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N), 1 << 30)
    ident = sf.analyze(n, Cp, Ci, Cx, None, 1 << 30)
    fill_default = int(np.sum(mi.array("ColCount", n)))
    fill_geo, fill_ident = int(np.sum(geo.ColCount)), int(np.sum(ident.ColCount))
    assert fill_default <= 2.0 * fill_geo, (fill_default, fill_geo)
    assert fill_ident >= 2.0 * fill_default, (fill_ident, fill_default)
    mi.set_perm(None)
    mi.analyze(common)
    assert int(np.sum(mi.array("ColCount", n))) == fill_ident
    mi.cleanup()


def test_builtin_ordering_fill_quality_figure():
    """fill-quality figure of the built-in ordering: against the geometric nested dissection that
    BASELINE config 2 prescribes, on the same grids.  Measured: nnz(L) 0.65-0.69x and flops 0.53-0.55x in 3-D
    (32^3 .. 96^3), nnz(L) 0.86-0.88x and flops 0.69-0.71x in 2-D (300^2, 1000^2)."""
    for dims, bound in (((32, 32, 32), 0.75), ((200, 200, 1), 0.95)):
        n, Cp, Ci, Cx = gen.laplacian_lower(*dims)
        geo = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(*dims), 8 << 30)
        blt = sf.analyze(n, Cp, Ci, Cx, sf.graph_nd_perm(n, Cp, Ci), 8 << 30)
        assert blt.flops_struct <= bound * geo.flops_struct, (dims, blt.flops_struct / geo.flops_struct)
        assert blt.ColCount.sum() <= geo.ColCount.sum()


def _isolated_boundary_grid(g):
    """g^3 7-point grid whose boundary rows are identity rows (Dirichlet rows of a finite-element matrix): one big component and
    6 g^2 - 12 g + 8 isolated vertices, the smallest vertex numbers among them"""
    n, Cp, Ci, Cx = gen.laplacian_lower(g, g, g)
    idx = np.arange(n)
    x, y, z = idx % g, (idx // g) % g, idx // (g * g)
    bnd = (x == 0) | (x == g - 1) | (y == 0) | (y == g - 1) | (z == 0) | (z == g - 1)
    cols = np.repeat(idx, np.diff(Cp))
    keep = (Ci == cols) | (~bnd[Ci] & ~bnd[cols])
    Cp2 = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(cols[keep], minlength=n), out=Cp2[1:])
    return n, Cp2, Ci[keep], Cx[keep]


def test_many_components_in_a_big_piece_stay_linear():
    """pieces of >= 100,000 vertices peeled their components off one team-BFS at a time -- a diagonal matrix of 104,000
    rows took 34 s (0.009 s at 99,999), 10^6 did not finish.  Now one sweep labels everything that is left."""
    import time
    for n in (104000, 1000000):
        t0 = time.perf_counter()
        p = sf.graph_nd_perm(n, np.arange(n + 1), np.arange(n))
        dt = time.perf_counter() - t0
        assert len(np.unique(p)) == n and p.min() == 0 and p.max() == n - 1
        assert dt < 5.0, (n, dt)        # 0.01 s / 0.14 s measured; the quadratic form needed minutes
    n, Cp, Ci, Cx = _isolated_boundary_grid(52)        # 140,608 vertices, 15,608 of them isolated, vertex 0 among them
    t0 = time.perf_counter()
    p = sf.graph_nd_perm(n, Cp, Ci)
    dt = time.perf_counter() - t0
    assert len(np.unique(p)) == n and dt < 5.0, dt
    # the big component (125,000 interior vertices) still gets a real dissection: far less fill than the natural order
    ident = sf.analyze(n, Cp, Ci, Cx, None, 8 << 30)
    nd = sf.analyze(n, Cp, Ci, Cx, p, 8 << 30)
    assert nd.flops_struct < 0.2 * ident.flops_struct


def test_big_component_behind_small_ones_is_independent_of_the_thread_count(monkeypatch):
    n, Cp, Ci, _ = _isolated_boundary_grid(52)
    perms = []
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("SF_ANALYZE_THREADS", threads)
        perms.append(sf.graph_nd_perm(n, Cp, Ci))
    assert np.array_equal(perms[0], perms[1]) and np.array_equal(perms[0], perms[2])

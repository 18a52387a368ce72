"""Degenerate and extreme shapes through the HIP path (plan ABI and struct entry points): the empty matrix, 1 x 1, a diagonal
matrix (n supernodes of one column, no update at all), one dense supernode (no tree), a chain (tridiagonal: tree of depth n),
a block-diagonal forest (several roots), panels whose width / height straddle the 64 / 512 blocking boundaries."""
import numpy as np
import pytest
import scipy.sparse as sp

from util import sf, gen, rel_err, nd_perm_py

pytestmark = pytest.mark.gpu


def lower_csc(A):
    L = sp.tril(sp.csc_matrix(A)).tocsc()
    L.sort_indices()
    return L.shape[0], L.indptr.astype(np.int64), L.indices.astype(np.int64), L.data.astype(np.float64)


def chol_check(n, Cp, Ci, Cx, perm=None, tol=1e-12):
    sym = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30)
    plan = sf.CholPlan(sym)
    plan.set_values(sym.Lx)
    plan.factorize()
    if n == 0:
        assert plan.get_factor().size == 0
        plan.close()
        return sym
    b = 1.0 + np.arange(n) / n
    x = plan.solve(b)
    assert plan.validate() <= 1e-13
    assert sf.validate_solution(sym, x, b) <= 1e-13
    # dense cross-check of the solution
    if n <= 2000:
        lc = np.repeat(np.arange(n), np.diff(sym.Lp))
        A = sp.coo_matrix((sym.Lx, (sym.Li, lc)), shape=(n, n)).toarray()
        A = A + A.T - np.diag(np.diag(A))
        assert np.allclose(A @ x, b, rtol=0, atol=tol * np.abs(A).sum(axis=0).max() * np.abs(x).max())
    plan.close()
    return sym


def test_empty_matrix():
    sym = chol_check(0, np.array([0]), np.zeros(0, dtype=np.int64), np.zeros(0))
    assert sym.nsuper == 0


def test_one_by_one():
    sym = sf.analyze(1, np.array([0, 1]), np.array([0]), np.array([4.0]), None, 1 << 30)
    plan = sf.CholPlan(sym)
    plan.set_values(sym.Lx)
    plan.factorize()
    assert plan.get_factor()[0] == 2.0
    assert plan.solve(np.array([8.0]))[0] == 2.0
    plan.close()
    # LU of the same
    sym = sf.analyze(1, np.array([0, 1]), np.array([0]), np.array([4.0]), None, 1 << 30, "lu", False)
    plan = sf.LUPlan(sym)
    plan.set_values(sym.Lx, sym.Ux)
    plan.factorize()
    assert plan.solve(np.array([8.0]))[0] == 2.0
    plan.close()


def test_diagonal_matrix():
    n = 300
    d = np.arange(1, n + 1, dtype=np.float64)
    sym = chol_check(n, np.arange(n + 1), np.arange(n), d)
    plan = sf.CholPlan(sym)
    plan.set_values(sym.Lx)
    plan.factorize()
    # the factor's diagonal is sqrt(d) in the permuted order
    L = plan.get_factor()
    cols = np.arange(n)
    s = sym.SuperMap[cols]
    nsrow = (sym.Lsip[1:] - sym.Lsip[:-1])[s]
    diag = L[sym.Lsxp[s] + (cols - sym.Super[s]) * (nsrow + 1)]
    assert np.allclose(np.sort(diag ** 2), d, rtol=1e-15)
    plan.close()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 129, 511, 512, 513, 700])
def test_one_dense_supernode(n):
    """a dense SPD matrix: ONE supernode of n columns -- widths around the 64-column step and the 512-column outer block"""
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n))
    A = B @ B.T + n * np.eye(n)
    nn, Cp, Ci, Cx = lower_csc(A)
    sym = chol_check(nn, Cp, Ci, Cx, np.arange(n))
    assert sym.nsuper == 1


def test_tridiagonal_chain():
    """tree of depth n / (relaxed supernode width): the level schedule degenerates to a chain"""
    n = 3000
    A = sp.diags([-np.ones(n - 1), 2.5 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])
    nn, Cp, Ci, Cx = lower_csc(A)
    chol_check(nn, Cp, Ci, Cx, np.arange(n))


def test_block_diagonal_forest():
    """independent blocks of different sizes: several roots, no separator at all"""
    rng = np.random.default_rng(7)
    blocks = []
    for m in (1, 3, 70, 130, 17, 600):
        B = rng.standard_normal((m, m))
        blocks.append(B @ B.T + m * np.eye(m))
    A = sp.block_diag(blocks)
    nn, Cp, Ci, Cx = lower_csc(A)
    chol_check(nn, Cp, Ci, Cx, np.arange(nn))


@pytest.mark.parametrize("n", [1, 64, 65, 513])
def test_dense_lu_single_supernode(n):
    rng = np.random.default_rng(100 + n)
    A = rng.standard_normal((n, n)) + n * np.eye(n)
    C_ = sp.csc_matrix(A)
    C_.sort_indices()
    sym = sf.analyze(n, C_.indptr.astype(np.int64), C_.indices.astype(np.int64), C_.data, np.arange(n), 1 << 30, "lu", False)
    plan = sf.LUPlan(sym)
    plan.set_values(sym.Lx, sym.Ux)
    plan.factorize()
    b = 1.0 + np.arange(n) / n
    x = plan.solve(b)
    assert np.max(np.abs(A @ x - b)) <= 1e-11 * np.abs(A).sum(axis=0).max() * max(np.abs(x).max(), 1.0)
    assert plan.validate() <= 1e-12
    plan.close()


def test_struct_path_on_tiny_matrices(tmp_path):
    """1 x 1 and 2 x 2 through the reference's entry points and the demo driver's stage order"""
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    for n, Cp, Ci, Cx in ((1, [0, 1], [0], [9.0]), (2, [0, 2, 3], [0, 1, 1], [4.0, 1.0, 3.0])):
        mi = sf.MatrixInfo()
        mi.set_csc(n, np.array(Cp), np.array(Ci), np.array(Cx))
        mi.analyze(common)
        mi.factorize(common)
        assert mi.validate() <= 1e-15
        mi.cleanup()
    common.close()


@pytest.mark.parametrize("case", ["one", "diag", "dense70", "forest", "lap4"])
def test_more_handlers_than_work(monkeypatch, case):
    """4 emulated handlers on matrices with fewer supernodes than handlers / no tree / no top: ranks without any supernode, groups
    that never form, segments that do not exist"""
    if sf.device_count() != 1:
        pytest.skip("emulated handlers are for one-GPU boxes")
    monkeypatch.setenv("SF_EMULATE_HANDLERS", "4")
    rng = np.random.default_rng(3)
    if case == "one":
        n, Cp, Ci, Cx = 1, np.array([0, 1]), np.array([0]), np.array([4.0])
    elif case == "diag":
        n = 10
        n, Cp, Ci, Cx = n, np.arange(n + 1), np.arange(n), np.arange(1.0, n + 1)
    elif case == "dense70":
        B = rng.standard_normal((70, 70))
        n, Cp, Ci, Cx = lower_csc(B @ B.T + 70 * np.eye(70))
    elif case == "forest":
        blocks = []
        for m in (2, 40, 90):
            B = rng.standard_normal((m, m))
            blocks.append(B @ B.T + m * np.eye(m))
        n, Cp, Ci, Cx = lower_csc(sp.block_diag(blocks))
    else:
        n, Cp, Ci, Cx = gen.laplacian_lower(4, 4, 4)
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    assert common.c.numGPU == 4
    mi = sf.MatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx)
    mi.analyze(common)
    mi.factorize(common)
    assert mi.validate() <= 1e-13
    mi.cleanup()
    common.close()


def test_graph_replay_of_a_resident_factorization(oracle, monkeypatch):
    """SF_GRAPH=1: the launches of a resident factorization captured once into a hipGraph and replayed (measured: no gain over the
    eager launches, profiles/r04_ad_graph_replay.txt -- kept as an option, and kept correct): new values every time, an eager run
    in between (profiling turns the graph off for that call), LU with a pivot perturbation whose scale changes with the values"""
    monkeypatch.setenv("SF_GRAPH", "1")
    N = 14
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    S = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(N, N, N), 1 << 30)
    ref, info, _ = oracle.chol_factorize(S)
    mask = oracle.lower_mask(S)
    plan = sf.CholPlan(S)
    for k, scale in enumerate((1.0, 4.0, 9.0, 16.0)):
        plan.set_values(S.Lx * scale)
        if k == 2:
            plan.set_profiling(True)
        plan.factorize()
        if k == 2:
            plan.set_profiling(False)
        assert rel_err(plan.get_factor(), ref * np.sqrt(scale), mask) <= 1e-12
        b = 1.0 + np.arange(n) / n
        assert sf.validate_solution(S, plan.solve(b) * scale, b) <= 1e-12
    plan.close()
    import golden_large
    c = golden_large.build_case("piv_zero_diag_12")
    SL = c["sym"]
    plan = sf.LUPlan(SL)
    plan.set_pivoting(0.1)
    for scale in (1.0, 3.0, 1.0):
        plan.set_values(SL.Lx * scale, SL.Ux * scale)
        plan.factorize()
        want, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(SL, tol=0.1)
        assert info == 0 and np.array_equal(plan.get_pivots(), pivpos)
        got = plan.get_factor()
        # L is scale-invariant, U scales: compare through the solve
        b = 1.0 + np.arange(SL.n) / SL.n
        x = plan.solve(b)
        xr = oracle.lu_solve_pivot(SL, want, pivpos, b) / scale
        assert np.max(np.abs(x - xr)) <= 1e-8 * np.max(np.abs(xr))
    plan.close()

"""LU pivoting (BASELINE config 5 "with partial pivoting").  The reference never pivots (magma_dgetrf_nopiv,
LU/Source/SparseFrame.c:2653; devIpiv = NULL, :3344; the static pre-pivot :589-673 is disabled), so there is nothing to be in
parity with: PARITY UNPINNED by construction.  Acceptance, as for any pivoted sparse LU:
  * on diagonally dominant inputs the pivoted factorization takes exactly the no-pivot path's pivots (they pass the threshold)
    -- and that one is in parity with the oracle (tests/test_lu.py);
  * on inputs that break the no-pivot path (exact zero pivots) or make it inaccurate (non-dominant), the scaled residual of the
    solve with the recorded interchanges is <= 1e-10 (device solve and the struct library's host solve)."""
import numpy as np
import pytest
import scipy.sparse as sp

from util import sf, gen, nd_perm_py

pytestmark = pytest.mark.gpu


def permuted_matrix(S):
    """P A P^T (the matrix the factorization works on) from the analysis: L part by column, U part by row"""
    n = S.n
    lc = np.repeat(np.arange(n), np.diff(S.Lp))
    A = sp.coo_matrix((S.Lx, (S.Li, lc)), shape=(n, n)).tocsr()
    ur = np.repeat(np.arange(n), np.diff(S.Up))
    off = S.Ui != ur
    return (A + sp.coo_matrix((S.Ux[off], (ur[off], S.Ui[off])), shape=(n, n)).tocsr()).tocsr()


def scaled_residual(A, x, b):
    r = A @ x - b
    return float(np.abs(r).max() / (abs(A).sum(axis=0).max() * np.abs(x).max() + np.abs(b).max()))


def test_dominant_input_takes_the_natural_pivots():
    """same pivots, hence the same arithmetic as the no-pivot path; the two runs differ only as two runs of ONE path do (the
    order of the fp64 atomics of the scatter is not fixed)"""
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(14, 14, 14, seed=5)
    S = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(14, 14, 14), 1 << 30, "lu", False)
    plan = sf.LUPlan(S)
    plan.set_values(S.Lx, S.Ux)
    plan.set_pivoting(0.0, 0.0)
    plan.factorize()
    ref = plan.get_factor().copy()
    plan.set_pivoting(0.1)
    plan.factorize()
    got = plan.get_factor()
    assert np.array_equal(plan.get_pivots(), np.arange(n))
    assert np.max(np.abs(got - ref)) <= 1e-13 * np.max(np.abs(ref))
    assert plan.stat("perturbed_pivots") == 0
    plan.close()


@pytest.mark.parametrize("fuse", ["", "0"])
def test_zero_diagonal_entries_need_the_interchanges(monkeypatch, fuse):
    """exact zeros on the diagonal at the first column of leaf supernodes: the no-pivot path stops at a zero pivot, the
    in-block interchanges find the neighbour's entry in the same column.  Both step forms (fused k_step / three launches)."""
    if fuse:
        monkeypatch.setenv("SF_FUSE_MAX", fuse)
    N = 12
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=9)
    perm = nd_perm_py(N, N, N)
    S0 = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
    widths = np.diff(S0.Super)
    leaves = [s for s in range(S0.nsuper) if widths[s] >= 4][::2]
    assert len(leaves) >= 5
    Cx = Cx.copy()
    cols = np.repeat(np.arange(n), np.diff(Cp))
    for s in leaves:
        g = S0.Perm[S0.Super[s]]                      # original index of the supernode's first column
        Cx[(Ci == g) & (cols == g)] = 0.0
    S = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
    A = permuted_matrix(S)
    b = 1.0 + np.arange(n) / n
    plan = sf.LUPlan(S)
    plan.set_values(S.Lx, S.Ux)
    plan.set_pivoting(0.0, 0.0)
    with pytest.raises(sf.SparseFrameError, match="SF_ERR_NOT_POSDEF"):
        plan.factorize()
    plan.set_pivoting(0.1)
    plan.factorize()
    piv = plan.get_pivots()
    assert sorted(piv.tolist()) == list(range(n)) and np.count_nonzero(piv != np.arange(n)) >= len(leaves)
    assert np.all(piv // 1 >= 0) and plan.stat("perturbed_pivots") == 0
    x = plan.solve(b)
    assert scaled_residual(A, x, b) <= 1e-10
    plan.close()


def test_non_dominant_random_matrix_residual():
    """random entries, weak diagonal: with the interchanges the residual is small; refinement steps absorb perturbed pivots"""
    N = 10
    n, Cp, Ci, Cx = gen.unsymmetric_general(N, N, N, seed=21)
    S = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(N, N, N), 1 << 30, "lu", False)
    A = permuted_matrix(S)
    b = 1.0 + np.arange(n) / n
    plan = sf.LUPlan(S)
    plan.set_values(S.Lx, S.Ux)
    plan.set_pivoting(1.0)              # partial pivoting inside the blocks
    plan.factorize()
    assert np.count_nonzero(plan.get_pivots() != np.arange(n)) > 0
    x = plan.solve(b)
    for _ in range(3):                  # iterative refinement on the host residual
        x = x + plan.solve(b - A @ x)
    res = scaled_residual(A, x, b)
    assert res <= 1e-10, (res, plan.stat("perturbed_pivots"))
    plan.close()


def test_struct_entry_points_with_interchanges():
    """LU library: SparseFrame_factorize records the interchanges in matrix_info->PivInv, SparseFrame_solve_supernodal
    (host) applies them block by block; validate()'s residual is the reference's formula"""
    N = 10
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=4)
    perm = nd_perm_py(N, N, N)
    S0 = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
    widths = np.diff(S0.Super)
    Cx = Cx.copy()
    cols = np.repeat(np.arange(n), np.diff(Cp))
    for s in [s for s in range(S0.nsuper) if widths[s] >= 4][::4]:
        g = S0.Perm[S0.Super[s]]
        Cx[(Ci == g) & (cols == g)] = 0.0
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    mi = sf.LUMatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx, symmetric=False)
    mi.set_perm(perm)
    mi.analyze(common)
    mi.factorize(common)
    piv = mi.array("PivInv", n)
    assert np.count_nonzero(piv != np.arange(n)) > 0
    assert mi.validate() <= 1e-10
    mi.cleanup()
    common.close()


def test_interchanges_in_shared_top_panels_emulated_ranks(monkeypatch):
    """two emulated handlers: exact zeros on the diagonal inside the TOP supernodes, whose 64-column chains run replicated
    on both ranks after the all-reduce -- both must take the same pivots (their inputs are bit-identical), the factor pieces
    come back from different ranks, PivInv is assembled from both, and the host solve must still reproduce b"""
    if sf.device_count() != 1:
        pytest.skip("emulated handlers are for one-GPU boxes")
    monkeypatch.setenv("SF_EMULATE_HANDLERS", "2")
    N = 12
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=31)
    perm = nd_perm_py(N, N, N)
    S0 = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
    owner, _, _ = sf.subtree_partition(S0, 2, 0.75)
    tops = [s for s in range(S0.nsuper) if owner[s] < 0]
    assert tops
    Cx = Cx.copy()
    cols = np.repeat(np.arange(n), np.diff(Cp))
    zeroed = 0
    for s in tops:
        for j in range(S0.Super[s], S0.Super[s + 1], 37):         # a few columns of every top supernode, never two in a row
            g = S0.Perm[j]
            Cx[(Ci == g) & (cols == g)] = 0.0
            zeroed += 1
    assert zeroed >= 3
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    assert common.c.numGPU == 2
    mi = sf.LUMatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx, symmetric=False)
    mi.set_perm(perm)
    mi.analyze(common)
    mi.factorize(common)
    piv = mi.array("PivInv", n)
    assert sorted(piv.tolist()) == list(range(n)) and np.count_nonzero(piv != np.arange(n)) >= 2
    assert mi.validate() <= 1e-10
    mi.cleanup()
    common.close()

"""LU pivoting (BASELINE config 5 "with partial pivoting").  The reference never pivots (magma_dgetrf_nopiv,
LU/Source/SparseFrame.c:2653; devIpiv = NULL, :3344; the static pre-pivot :589-673 is disabled), so there is nothing to be in
parity with: PARITY UNPINNED by construction.  Acceptance, as for any pivoted sparse LU:
  * on diagonally dominant inputs the pivoted factorization takes exactly the no-pivot path's pivots (they pass the threshold)
    -- and that one is in parity with the oracle (tests/test_lu.py);
  * on inputs that break the no-pivot path (exact zero pivots) or make it inaccurate (non-dominant), the scaled residual of the
    solve with the recorded interchanges is <= 1e-10 (device solve and the struct library's host solve);
  * round 3: the rule itself (threshold test, tie rule, perturbation, what moves with a row) is restated in scalar C in the oracle
    (sfo_lu_factorize_pivot; pinned on the CPU in tests/test_lu_pivot_oracle.py against LAPACK partial pivoting and a numpy
    statement of the block-restricted rule): the HIP path's pivot sequence and PivInv must EQUAL the oracle's and its L / U
    values agree to 1e-12 of the largest entry (1e-10 on the non-dominant random matrices, whose element growth under
    block-restricted pivoting amplifies the rounding differences of two summation orders), also against committed vectors
    (tests/golden/lu_small.json)."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from util import sf, gen, nd_perm_py, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture
def struct_pivoting():
    """opt in to pivoting on the struct path (process-wide policy), and back to the reference's behaviour afterwards"""
    sf.LUMatrixInfo.set_pivoting(0.1, 1.4901161193847656e-08)
    yield
    sf.LUMatrixInfo.set_pivoting(0.0, 0.0)


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


def test_struct_entry_points_with_interchanges(struct_pivoting):
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


def test_interchanges_in_shared_top_panels_emulated_ranks(monkeypatch, struct_pivoting):
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


# ---------------------------------------------------------------------------------------------------------------------
# round 3: the HIP path against the oracle's statement of the pivoting rule
# ---------------------------------------------------------------------------------------------------------------------
def _dense_csc(A):
    n = A.shape[0]
    return n, np.arange(0, n * n + 1, n, dtype=np.int64), np.tile(np.arange(n, dtype=np.int64), n), np.ascontiguousarray(A.T).ravel()


def _zeroed_diagonal_case(N, seed, every):
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=seed)
    perm = nd_perm_py(N, N, N)
    S0 = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
    Cx = Cx.copy()
    cols = np.repeat(np.arange(n), np.diff(Cp))
    for j in range(0, n, every):
        g = S0.Perm[j]
        Cx[(Ci == g) & (cols == g)] = 0.0
    return n, Cp, Ci, Cx, perm


def _blockdiag_dense_csc(sizes, seed, weak_every):
    """independent dense unsymmetric blocks (one supernode each, all in ONE level set, widths that are no multiples of 64, one of them
    wider than an outer block): the fused LU steps of that set carry panels that drop out at different steps, pre-update tasks for
    some and not for others.  Every weak_every-th diagonal entry is made tiny, so rows really are interchanged."""
    rng = np.random.default_rng(seed)
    n = sum(sizes)
    Cp, Ci, Cx = [0], [], []
    o = 0
    for k in sizes:
        B = rng.uniform(-1, 1, (k, k)) + 0.5 * k ** 0.5 * np.eye(k)
        B[np.arange(0, k, weak_every), np.arange(0, k, weak_every)] *= 1e-3
        for j in range(k):
            Ci.append(np.arange(o, o + k, dtype=np.int64))
            Cx.append(B[:, j])
            Cp.append(Cp[-1] + k)
        o += k
    return n, np.array(Cp, dtype=np.int64), np.concatenate(Ci), np.concatenate(Cx)


def pivot_cases():
    c = []
    rng = np.random.default_rng(11)
    c.append(("dense_64_tol1", *_dense_csc(rng.uniform(-1, 1, (64, 64))), None, 1.0, 1e-12))
    c.append(("dense_200_tol01", *_dense_csc(rng.uniform(-1, 1, (200, 200))), None, 0.1, 1e-10))
    c.append(("dense_700_tol05", *_dense_csc(rng.uniform(-1, 1, (700, 700)) + 4 * np.eye(700)), None, 0.5, 1e-11))
    c.append(("zero_diag_12", *_zeroed_diagonal_case(12, 9, 7), 0.1, 1e-12))
    c.append(("zero_diag_16", *_zeroed_diagonal_case(16, 10, 5), 0.3, 1e-11))      # multipliers of 1e6 next to the zeroed entries
    n, Cp, Ci, Cx = gen.unsymmetric_general(10, 10, 10, seed=21)
    c.append(("general_10_tol1", n, Cp, Ci, Cx, nd_perm_py(10, 10, 10), 1.0, 1e-10))
    n, Cp, Ci, Cx = gen.unsymmetric_general(14, 14, 14, seed=22, diag_scale=1.0)
    c.append(("general_14_tol03", n, Cp, Ci, Cx, nd_perm_py(14, 14, 14), 0.3, 1e-10))
    # mixed panel widths in one level set, no multiples of 64, more than one outer block, pivoting on
    c.append(("blockdiag_700_130_577_65_tol03", *_blockdiag_dense_csc((700, 130, 577, 65), 41, 5), None, 0.3, 1e-10))
    return c


@pytest.mark.parametrize("fuse", ["", "0"], ids=["fused_step", "three_launches"])
@pytest.mark.parametrize("case", pivot_cases(), ids=lambda c: c[0])
def test_pivot_sequence_and_factor_match_the_oracle(oracle, monkeypatch, case, fuse):
    name, n, Cp, Ci, Cx, perm, tol, vtol = case
    if fuse:
        monkeypatch.setenv("SF_FUSE_MAX", fuse)
    S = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
    ref, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=tol)
    assert info == 0
    plan = sf.LUPlan(S)
    plan.set_values(S.Lx, S.Ux)
    plan.set_pivoting(tol)
    plan.factorize()
    got_piv = plan.get_pivots()
    assert np.array_equal(got_piv, pivpos), (name, np.flatnonzero(got_piv != pivpos)[:10])
    assert np.count_nonzero(pivpos != np.arange(n)) > 0
    assert int(plan.stat("perturbed_pivots")) == nper
    Lsx = plan.get_factor()
    assert rel_err(Lsx, ref) <= vtol, (name, rel_err(Lsx, ref))
    # The device solve (interchanges applied block by block in the forward sweep) against the oracle's solve applied to THE DEVICE'S OWN
    # factor and pivots: same inputs, so the comparison isolates the solve from the factor's conditioning.  What remains is the
    # rounding of two summation orders, which a triangular solve amplifies by its own condition number: the yardstick is measured
    # here, not assumed -- the spread of the ORACLE's solution when every entry of that factor moves by one ulp at random.  CPU
    # measurements of that spread: dense_64 2e-15, dense_200 1e-13, dense_700 7e-13, general 3e-13, zero_diag_12 7e-12, zero_diag_16
    # 1.0-1.6e-10 (multipliers of 2e6 next to the zeroed entries: there 1e-12 is below what ONE ulp in the factor does, whatever the
    # solver).  Bound: 1e-12, or 8 x that spread where the spread is larger (the device solve rounds in another order than the oracle's
    # loops AND starts from a factor that is itself a few ulps away: ratios seen on zero_diag_16 over the rounds: 0.7 - 4.1).
    b = 1.0 + np.arange(n) / n
    x = plan.solve(b)
    want = oracle.lu_solve_pivot(S, Lsx, got_piv, b)
    rng = np.random.default_rng(5)
    spread = 0.0
    for _ in range(5):
        moved = Lsx * (1.0 + rng.integers(-1, 2, Lsx.size) * 1.1102230246251565e-16)
        spread = max(spread, float(np.max(np.abs(oracle.lu_solve_pivot(S, moved, got_piv, b) - want)) / np.abs(want).max()))
    err = float(np.max(np.abs(x - want)) / np.abs(want).max())
    assert err <= max(1e-12, 8.0 * spread), (name, err, spread)
    # ... and against the oracle's solve with the ORACLE's factor (factor conditioning included): the looser end-to-end statement
    want_ref = oracle.lu_solve_pivot(S, ref, pivpos, b)
    assert np.max(np.abs(x - want_ref)) <= max(vtol, 1e-11) * 1e2 * max(1.0, spread / 1e-12) * np.abs(want_ref).max(), name
    plan.close()


def test_perturbed_pivots_match_the_oracle(oracle):
    """a column that is zero in its whole diagonal block: both sides replace the pivot by +sqrt(eps) max|a_ij| and count it"""
    n = 40
    rng = np.random.default_rng(3)
    A = rng.uniform(-1, 1, (n, n))
    A[:, 7] = 0.0
    A[:, 23] = 0.0
    A[5, 7] = 1e-30          # below the perturbation threshold, not zero: sign kept
    S = sf.analyze(*_dense_csc(A), None, 1 << 30, "lu", False)
    ref, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=1.0)
    assert nper == 2 and info == 0
    plan = sf.LUPlan(S)
    plan.set_values(S.Lx, S.Ux)
    plan.set_pivoting(1.0)
    plan.factorize()
    assert int(plan.stat("perturbed_pivots")) == 2
    assert np.array_equal(plan.get_pivots(), pivpos)
    assert rel_err(plan.get_factor(), ref) <= 1e-9      # 1 / sqrt(eps) multipliers
    plan.close()


def test_lu_golden_fixtures():
    """committed vectors (tests/golden/make_golden_lu.py: oracle output accepted there only after agreeing with dense no-pivot
    LU / LAPACK partial pivoting / the numpy block rule / a solve); no oracle code runs here"""
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lu_small.json")) as f:
        G = json.load(f)
    assert any(k.startswith("piv_") for k in G) and any(k.startswith("nopiv_") for k in G)
    for name, g in G.items():
        S = sf.analyze(g["n"], g["Cp"], g["Ci"], g["Cx"], g["perm"], g["devSlotSize"], "lu", g["symmetric"])
        for k in ("Super", "Lsip", "Lsxp", "Lsi", "Perm"):
            assert np.array_equal(getattr(S, k), np.asarray(g[k], dtype=np.int64)), (name, k)
        plan = sf.LUPlan(S)
        plan.set_values(S.Lx, None if g["symmetric"] else S.Ux)
        plan.set_pivoting(g["tol"], g["perturb"])
        plan.factorize()
        if g["pivpos"] is not None:
            assert np.array_equal(plan.get_pivots(), np.asarray(g["pivpos"])), name
        else:
            assert np.array_equal(plan.get_pivots(), np.arange(g["n"])), name
        assert int(plan.stat("perturbed_pivots")) == g["perturbed"]
        assert rel_err(plan.get_factor(), np.asarray(g["Lsx"])) <= (1e-10 if name.startswith("piv_general") else 1e-12), name
        plan.close()


@pytest.mark.parametrize("fuse", ["", "0"], ids=["fused_step", "three_launches"])
@pytest.mark.parametrize("name", ["nopiv_lu_stencil_16", "piv_dense_200_tol01", "piv_zero_diag_12"])
def test_lu_golden_fixtures_large(name, fuse, monkeypatch):
    """tests/golden/large_sampled.json: the reference's own LU behaviour (NO pivoting) on an unsymmetric 16^3 stencil with a
    1340-column root (accepted entry by entry against SuperLU in natural order without pivoting), a pivoted front of four 64-column
    blocks (accepted against the numpy statement of the block rule) and a pivoted sparse factorization with a 666-column root
    (accepted by its solve against SuperLU's): full pivot sequence, perturbation count, sampled factor entries, sum log|pivots|,
    sum |entries|.  No oracle code runs here."""
    import golden_large as GL
    if fuse:
        monkeypatch.setenv("SF_FUSE_MAX", fuse)
    g = GL.load()[name]
    c = GL.build_case(name)
    GL.check_inputs_and_symbolic(name, g, c)
    S = c["sym"]
    plan = sf.LUPlan(S)
    plan.set_values(S.Lx, S.Ux)
    plan.set_pivoting(g["tol"], 0.0 if g["tol"] == 0.0 else 1.4901161193847656e-08)
    plan.factorize()
    want_piv = np.arange(g["n"]) if g["pivpos"] is None else np.asarray(g["pivpos"])      # no pivoting (the reference's LU): identity
    assert np.array_equal(plan.get_pivots(), want_piv), name
    assert int(plan.stat("perturbed_pivots")) == g["perturbed"]
    GL.check_factor(name, g, S, plan.get_factor(), 1e-11)
    plan.close()


@pytest.mark.parametrize("handlers", [1, 2])
def test_struct_path_pivinv_and_factor_match_the_oracle(oracle, monkeypatch, handlers, struct_pivoting):
    """SparseFrame_factorize (LU library): matrix_info->PivInv and every value of matrix_info->Lsx against the oracle, with
    one handler and with two emulated handlers whose SHARED top panels carry zeroed diagonal entries (both ranks must take the
    oracle's pivots; the pieces of Lsx come back from different ranks)"""
    if handlers > 1:
        if sf.device_count() != 1:
            pytest.skip("emulated handlers are for one-GPU boxes")
        monkeypatch.setenv("SF_EMULATE_HANDLERS", str(handlers))
    N = 12
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=31)
    perm = nd_perm_py(N, N, N)
    S0 = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
    owner, _, _ = sf.subtree_partition(S0, 2, 0.75)
    Cx = Cx.copy()
    cols = np.repeat(np.arange(n), np.diff(Cp))
    for s in [s for s in range(S0.nsuper) if owner[s] < 0]:
        for j in range(S0.Super[s], S0.Super[s + 1], 37):
            g = S0.Perm[j]
            Cx[(Ci == g) & (cols == g)] = 0.0
    S = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
    ref, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=0.1)
    assert info == 0 and np.count_nonzero(pivpos != np.arange(n)) >= 2
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    assert common.c.numGPU == handlers
    mi = sf.LUMatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx, symmetric=False)
    mi.set_perm(perm)
    mi.analyze(common)
    mi.factorize(common)
    assert np.array_equal(mi.array("PivInv", n), pivpos)        # (the field holds original row -> position, see sparseframe_lu_hip.h)
    assert mi.perturbed_pivots() == 0 and nper == 0
    Lsx = mi.array("Lsx", int(S.xsize))
    assert rel_err(Lsx, ref) <= 1e-12
    assert mi.validate() <= 1e-10
    mi.cleanup()
    common.close()


def test_struct_path_default_is_the_reference_no_pivot_behaviour(oracle):
    """without SparseFrame_set_pivoting the LU library factors as the reference does: PivInv = identity, Lsx = the no-pivot
    oracle's, and an exact zero pivot is reported instead of being worked around"""
    N = 10
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=4)
    perm = nd_perm_py(N, N, N)
    S = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
    ref, info, _ = oracle.lu_factorize(S)
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    mi = sf.LUMatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx, symmetric=False)
    mi.set_perm(perm)
    mi.analyze(common)
    mi.factorize(common)
    assert np.array_equal(mi.array("PivInv", n), np.arange(n)) and mi.perturbed_pivots() == 0
    assert rel_err(mi.array("Lsx", int(S.xsize)), ref) <= 1e-12
    mi.cleanup()
    Cx = Cx.copy()
    cols = np.repeat(np.arange(n), np.diff(Cp))
    g = S.Perm[S.Super[0]]
    Cx[(Ci == g) & (cols == g)] = 0.0
    mi = sf.LUMatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx, symmetric=False)
    mi.set_perm(perm)
    mi.analyze(common)
    with pytest.raises(sf.SparseFrameError, match="SF_ERR_NOT_POSDEF"):
        mi.factorize(common)
    mi.cleanup()
    common.close()


def test_reanalysis_of_a_larger_matrix_resizes_pivinv(struct_pivoting):
    """PivInv was allocated once per matrix_info; a re-analysis with a larger matrix overflowed it"""
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    mi = sf.LUMatrixInfo()
    for N in (4, 9):
        n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=N)
        mi.set_csc(n, Cp, Ci, Cx, symmetric=False)
        mi.set_perm(nd_perm_py(N, N, N))
        mi.analyze(common)
        mi.factorize(common)
        assert np.array_equal(mi.array("PivInv", n), np.arange(n))
        assert mi.validate() <= 1e-13
    mi.cleanup()
    common.close()


def test_two_matrices_with_their_own_pivot_settings_on_one_handler_list(oracle):
    """per-matrix_info policy (SparseFrame_set_matrix_pivoting): matrix A pivots, matrix B -- the same pattern, so the same cached
    device plan -- must not inherit that, and the other way round; the process-wide default stays the reference's behaviour"""
    N = 10
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=4)
    perm = nd_perm_py(N, N, N)
    S0 = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
    widths = np.diff(S0.Super)
    Cz = Cx.copy()
    cols = np.repeat(np.arange(n), np.diff(Cp))
    for s in [s for s in range(S0.nsuper) if widths[s] >= 4][::4]:
        g = S0.Perm[S0.Super[s]]
        Cz[(Ci == g) & (cols == g)] = 0.0                   # exact zero pivots: only a pivoting factorization gets through
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    a, b = sf.LUMatrixInfo(), sf.LUMatrixInfo()
    a.set_csc(n, Cp, Ci, Cz, symmetric=False)
    a.set_perm(perm)
    a.set_matrix_pivoting(0.1, 1.4901161193847656e-08)
    b.set_csc(n, Cp, Ci, Cx, symmetric=False)
    b.set_perm(perm)
    a.analyze(common)
    b.analyze(common)
    builds = common.plan_builds()
    for _ in range(2):
        a.factorize(common)
        assert np.count_nonzero(a.array("PivInv", n) != np.arange(n)) > 0 and a.validate() <= 1e-10
        b.factorize(common)                                  # same plan, no policy of its own: the reference's no-pivot factor
        assert np.array_equal(b.array("PivInv", n), np.arange(n)) and b.validate() <= 1e-13
    assert common.plan_builds() == builds + 1               # one pattern, one plan
    ref, info, _ = oracle.lu_factorize(sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False))
    assert rel_err(b.array("Lsx", int(b.c.xsize)), ref) <= 1e-12
    # without its policy matrix A meets its zero pivot, as the reference would
    lu_lib = sf._lib.lu_lib
    assert lu_lib.SparseFrame_clear_matrix_pivoting(__import__("ctypes").byref(a.c)) == 0
    with pytest.raises(sf.SparseFrameError, match="SF_ERR_NOT_POSDEF"):
        a.factorize(common)
    a.cleanup()
    b.cleanup()
    common.close()

"""The oracle against everything that pins it:
 (a) survey-recorded counts: printed by a survey-time build against stand-in headers, which by the
     rules pins nothing -- a consistency check with the survey's numbers only; parity with the actual reference is UNPINNED,
 (b) brute-force definitions of the symbolic quantities,
 (c) dense LAPACK Cholesky (the factor is unique), for both BLAS back ends of the oracle."""
import json
import os

import numpy as np
import pytest

from util import gen, nd_perm_py, small_cases, dense_reference_factor, panel_entries_from_dense, rel_err

HERE = os.path.dirname(os.path.abspath(__file__))
RECORDED = json.load(open(os.path.join(HERE, "golden", "reference_recorded.json")))


def brute_symbolic(n, Lp, Li):
    """column structures of L by straightforward symbolic elimination (small n)"""
    cols = [set() for _ in range(n)]
    for j in range(n):
        for p in range(Lp[j], Lp[j + 1]):
            if Li[p] > j:
                cols[j].add(int(Li[p]))
    parent = [-1] * n
    for j in range(n):
        if cols[j]:
            p = min(cols[j])
            parent[j] = p
            cols[p] |= (cols[j] - {p})
    return parent, [len(c) + 1 for c in cols], cols


def test_survey_recorded_counts_2d(oracle):
    """survey-recorded, stub-header build; parity unpinned"""
    rec = RECORDED["lap2d_100x100_identity_1GiB"]
    n, Cp, Ci, Cx = gen.laplacian_lower(100, 100)
    assert len(Ci) == rec["mtx_entries"]
    S = oracle.symbolic.analyze(n, Cp, Ci, Cx, None, 1 << 30)
    assert (S["nfsuper"], S["nsuper"], S["nstage"]) == (rec["nfsuper"], rec["nsuper"], rec["nstage"])


def test_survey_recorded_counts_3d_32(oracle):
    """survey-recorded, stub-header build; parity unpinned"""
    rec = RECORDED["lap3d_32_geomND_8GiB"]
    n, Cp, Ci, Cx = gen.laplacian_lower(32, 32, 32)
    S = oracle.symbolic.analyze(n, Cp, Ci, Cx, nd_perm_py(32, 32, 32), 8 << 30)
    assert (S["nfsuper"], S["nsuper"]) == (rec["nfsuper"], rec["nsuper"])


@pytest.mark.parametrize("case", small_cases(), ids=lambda c: c[0])
def test_symbolic_against_brute_force(oracle, case):
    name, n, Cp, Ci, Cx, perm, slot = case
    S = oracle.symbolic.analyze(n, Cp, Ci, Cx, perm, slot)
    parent, counts, cols = brute_symbolic(n, S["Lp"], S["Li"])
    assert S["Parent"] == parent
    assert S["ColCount"] == counts
    # every supernode's row list = its columns, then the union of its columns' structures
    for s in range(S["nsuper"]):
        c0, c1 = S["Super"][s], S["Super"][s + 1]
        rows = S["Lsi"][S["Lsip"][s]:S["Lsip"][s + 1]]
        want = set(range(c0, c1))
        for j in range(c0, c1):
            want |= cols[j]
        assert rows == sorted(want)
        assert rows[:c1 - c0] == list(range(c0, c1))
    # postorder: children before parents, and Perm is a permutation
    assert sorted(S["Perm"]) == list(range(n))
    assert all(p == -1 or p > j for j, p in enumerate(S["Parent"]))
    assert sorted(S["LeafQueue"][:S["nsleaf"]]) == sorted(
        s for s in range(S["nsuper"]) if s not in {S["SuperMap"][S["Lsi"][S["Lsip"][t] + S["Super"][t + 1] - S["Super"][t]]]
                                                   for t in range(S["nsuper"]) if S["Lsip"][t + 1] - S["Lsip"][t] > S["Super"][t + 1] - S["Super"][t]})


@pytest.mark.parametrize("blas", ["builtin", "auto"])
@pytest.mark.parametrize("case", small_cases(), ids=lambda c: c[0])
def test_numeric_against_dense_lapack(oracle, case, blas):
    name, n, Cp, Ci, Cx, perm, slot = case
    if n > 1200:
        pytest.skip("dense check only for small n")
    oracle.blas_init(blas, threads=2)
    S = oracle.symbolic.analyze(n, Cp, Ci, Cx, perm, slot)
    Lsx, info, stats = oracle.chol_factorize(S)
    assert info == 0
    A, L = dense_reference_factor(S)
    want = panel_entries_from_dense(S, L)
    mask = oracle.lower_mask(S)
    assert rel_err(Lsx, want, mask) <= 1e-12          # tolerance of SURVEY 8(c)
    res, x = oracle.chol_residual(S, Lsx)
    assert res <= 1e-13
    b = 1 + np.arange(n) / n
    assert np.allclose(A @ x, b, rtol=0, atol=1e-9 * np.max(np.abs(b)))
    oracle.blas_init("auto", threads=4)


@pytest.mark.parametrize("name", ["chol_lap3d_24", "chol_stencil2d_200", "nopiv_lu_stencil_16", "piv_dense_200_tol01", "piv_zero_diag_12"])
def test_oracle_against_the_large_sampled_fixtures(oracle, name):
    """the oracle (threaded BLAS back end; the fixtures were made with its built-in loops) against tests/golden/large_sampled.json,
    whose values were accepted against dense LAPACK / SuperLU / the numpy block rule (tests/golden/make_golden_large.py)"""
    import golden_large as GL
    g = GL.load()[name]
    c = GL.build_case(name)
    GL.check_inputs_and_symbolic(name, g, c)
    S = c["sym"]
    oracle.blas_init("auto", threads=4)
    if g["method"] == "cholesky":
        Lsx, info, _ = oracle.chol_factorize(S)
        assert info == 0
        GL.check_factor(name, g, S, Lsx, 1e-12)
    elif g["pivpos"] is None:
        Lsx, info, _ = oracle.lu_factorize(S)                 # the no-pivot path (the reference's behaviour, L:2653)
        assert info == 0
        GL.check_factor(name, g, S, Lsx, 1e-12)
    else:
        Lsx, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=g["tol"])
        assert info == 0 and nper == g["perturbed"] and np.array_equal(pivpos, np.asarray(g["pivpos"]))
        GL.check_factor(name, g, S, Lsx, 1e-12)


def test_not_positive_definite_is_reported(oracle):
    n, Cp, Ci, Cx = gen.laplacian_lower(6, 6)
    Cx = Cx.copy()
    Cx[Cp[20]] = -1.0   # a negative diagonal entry
    S = oracle.symbolic.analyze(n, Cp, Ci, Cx, None, 1 << 30)
    _, info, _ = oracle.chol_factorize(S)
    assert info != 0

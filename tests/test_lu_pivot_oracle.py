"""CPU pins of the pivoted-LU oracle (oracle/sf_oracle_numeric.c: sfo_lu_factorize_pivot / sfo_lu_solve_pivot).

PARITY UNPINNED by construction: the reference never pivots (magma_dgetrf_nopiv LU/Source/SparseFrame.c:2653, devIpiv = NULL
:3344, static pre-pivot :589-673 compiled out), so threshold pivoting inside the 64 x 64 diagonal blocks is the product's own
rule.  The oracle restates that rule in scalar C; here it is pinned against what CAN be pinned:
  * tol = 1 on a dense front of <= 64 columns IS LAPACK partial pivoting: pivot sequence, L and U against scipy.linalg.lu;
  * tol = 0 is the no-pivot oracle (which is pinned against dense no-pivot LU in tests/test_lu.py);
  * a second, independent numpy statement of the block-restricted rule on dense matrices of several blocks;
  * solves with the recorded interchanges reproduce b on matrices the no-pivot path cannot factor.
The GPU tests (tests/test_lu_pivot.py) then compare the HIP path's pivots, PivInv and L / U values with this oracle."""
import numpy as np
import pytest
import scipy.linalg

from util import sf, gen, nd_perm_py, rel_err

SQRT_EPS = 1.4901161193847656e-08


def dense_csc(A):
    n = A.shape[0]
    Cp = np.arange(0, n * n + 1, n, dtype=np.int64)
    Ci = np.tile(np.arange(n, dtype=np.int64), n)
    return n, Cp, Ci, np.ascontiguousarray(A.T).ravel()


def unpack_single_front(S, Lsx):
    """dense unit-lower L and upper U of a matrix whose analysis is ONE supernode (packed L11 \\ U11)"""
    assert S.nsuper == 1
    n = S.n
    P = Lsx[:n * n].reshape(n, n).T          # column-major, lda = n
    return np.tril(P, -1) + np.eye(n), np.triu(P)


def block_rule_numpy(A, tol, eps, nb=64):
    """independent statement of the rule on a dense square matrix that is one supernode: explicit row SWAPS inside the block,
    applied to the block's own columns and everything right of them only; `orig` tracks which original row sits where so that
    'natural row' and the tie rule (lowest ORIGINAL position) mean what they mean in the implicit form.
    Returns (packed LU, pivpos, perturbed)."""
    A = A.copy()
    n = len(A)
    pivpos = np.arange(n)
    nper = 0
    for k0 in range(0, n, nb):
        k1 = min(k0 + nb, n)
        orig = list(range(k0, k1))            # orig[q - k0] = original row now at position q
        for j in range(k0, k1):
            cand = list(range(j, k1))         # positions not used yet
            p = None
            if tol > 0:
                m = max(abs(A[q, j]) for q in cand)
                natq = [q for q in cand if orig[q - k0] == j]
                if natq and A[natq[0], j] != 0 and abs(A[natq[0], j]) >= tol * m:
                    p = natq[0]
                else:
                    p = min((q for q in cand if abs(A[q, j]) == m), key=lambda q: orig[q - k0])
            else:
                p = [q for q in cand if orig[q - k0] == j][0]
            if p != j:
                A[[j, p], k0:] = A[[p, j], k0:]
                orig[j - k0], orig[p - k0] = orig[p - k0], orig[j - k0]
            if eps > 0 and not (abs(A[j, j]) >= eps):
                A[j, j] = -eps if A[j, j] < 0 else eps
                nper += 1
            A[j + 1:, j] /= A[j, j]
            A[j + 1:, j + 1:] -= np.outer(A[j + 1:, j], A[j, j + 1:])
        for q in range(k0, k1):
            pivpos[orig[q - k0]] = q
    return A, pivpos, nper


@pytest.mark.parametrize("n,seed", [(1, 0), (5, 1), (33, 2), (64, 3)])
def test_tol_one_on_a_single_block_is_lapack_partial_pivoting(oracle, n, seed):
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1, 1, (n, n))
    S = sf.analyze(*dense_csc(A), None, 1 << 30, "lu", False)
    Lsx, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=1.0)
    assert info == 0 and nper == 0
    L, U = unpack_single_front(S, Lsx)
    p, l, u = scipy.linalg.lu(A)              # A = p l u
    assert np.array_equal(np.argmax(p, axis=0), pivinv)          # row at position q is original row pivinv[q]
    assert np.array_equal(pivpos[pivinv], np.arange(n))
    assert np.max(np.abs(L - l)) <= 1e-12 * max(1.0, np.abs(l).max())
    assert np.max(np.abs(U - u)) <= 1e-12 * np.abs(u).max()
    assert np.allclose(A[pivinv] , L @ U, rtol=0, atol=1e-13 * n)


@pytest.mark.parametrize("n,tol,seed", [(64, 0.1, 4), (150, 1.0, 5), (150, 0.1, 6), (200, 0.5, 7), (130, 0.0, 8)])
def test_block_restricted_rule_against_a_numpy_statement(oracle, n, tol, seed):
    """several 64-column blocks inside ONE supernode: interchanges stay inside their block, entries left of a block stay put"""
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1, 1, (n, n))
    if tol == 0.0:
        A += n * np.eye(n)
    S = sf.analyze(*dense_csc(A), None, 1 << 30, "lu", False)
    assert S.nsuper == 1
    Lsx, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=tol)
    want, wpos, wper = block_rule_numpy(A, tol, SQRT_EPS * np.abs(A).max())
    assert info == 0 and nper == wper
    assert np.array_equal(pivpos, wpos)
    assert np.array_equal(pivpos // 64, np.arange(n) // 64)      # nothing leaves its block
    got = Lsx[:n * n].reshape(n, n).T
    assert np.max(np.abs(got - want)) <= 1e-11 * np.abs(want).max()
    if tol > 0:
        assert np.count_nonzero(pivpos != np.arange(n)) > 0
    b = 1 + np.arange(n) / n
    x = oracle.lu_solve_pivot(S, Lsx, pivpos, b)
    assert np.max(np.abs(A @ x - b)) <= 1e-8 * np.abs(x).max() * n


def test_tol_zero_is_the_no_pivot_oracle(oracle):
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(8, 8, 8, seed=4)
    S = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(8, 8, 8), 1 << 30, "lu", False)
    ref, info0, _ = oracle.lu_factorize(S)
    for tol in (0.0, 0.1):                    # dominant input: the natural pivots pass any threshold <= its dominance
        Lsx, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=tol)
        assert info == 0 and info0 == 0 and nper == 0
        assert np.array_equal(pivpos, np.arange(n)) and np.array_equal(pivinv, np.arange(n))
        assert rel_err(Lsx, ref) <= 1e-13
        b = 1 + np.arange(n) / n
        assert np.allclose(oracle.lu_solve_pivot(S, Lsx, pivpos, b), oracle.lu_solve(S, ref, b), rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("N,seed,tol", [(6, 21, 1.0), (8, 22, 1.0), (8, 23, 0.3)])
def test_sparse_non_dominant_matrix_solves_with_the_interchanges(oracle, N, seed, tol):
    n, Cp, Ci, Cx = gen.unsymmetric_general(N, N, N, seed=seed)
    S = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(N, N, N), 1 << 30, "lu", False)
    A = gen.dense_from_csc(n, Cp, Ci, Cx)[np.ix_(S.Perm, S.Perm)]
    Lsx, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=tol)
    assert info == 0
    assert sorted(pivpos.tolist()) == list(range(n)) and np.count_nonzero(pivpos != np.arange(n)) > 0
    assert np.array_equal(pivinv[pivpos], np.arange(n))
    # an interchange never leaves its supernode's 64-column block
    sup = S.SuperMap
    assert np.array_equal(sup[pivpos], sup)
    assert np.array_equal((pivpos - S.Super[sup]) // 64, (np.arange(n) - S.Super[sup]) // 64)
    b = 1 + np.arange(n) / n
    x = oracle.lu_solve_pivot(S, Lsx, pivpos, b)
    for _ in range(3):                        # refinement absorbs perturbed pivots and the growth of restricted pivoting
        x = x + oracle.lu_solve_pivot(S, Lsx, pivpos, b - A @ x)
    r = A @ x - b
    res = np.abs(r).max() / (np.abs(A).sum(axis=0).max() * np.abs(x).max() + np.abs(b).max())
    assert res <= 1e-10, (res, nper)


def test_zero_diagonal_and_perturbation(oracle):
    """exact zeros on the diagonal: tol = 0 reports the zero pivot (info), the interchanges find a neighbour; a column that is
    zero in its whole block is perturbed to +eps and counted"""
    N = 6
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=9)
    perm = nd_perm_py(N, N, N)
    S0 = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
    Cx = Cx.copy()
    cols = np.repeat(np.arange(n), np.diff(Cp))
    widths = np.diff(S0.Super)
    hit = [s for s in range(S0.nsuper) if widths[s] >= 4][::2]
    for s in hit:
        g = S0.Perm[S0.Super[s]]
        Cx[(Ci == g) & (cols == g)] = 0.0
    S = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
    _, info, *_ = oracle.lu_factorize_pivot(S, tol=0.0, perturb=0.0)
    assert info > 0
    Lsx, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=0.1)
    assert info == 0 and nper == 0 and np.count_nonzero(pivpos != np.arange(n)) >= len(hit)
    A = gen.dense_from_csc(n, Cp, Ci, Cx)[np.ix_(S.Perm, S.Perm)]
    b = 1 + np.arange(n) / n
    x = oracle.lu_solve_pivot(S, Lsx, pivpos, b)
    assert np.abs(A @ x - b).max() <= 1e-10 * np.abs(A).sum(axis=0).max() * np.abs(x).max()
    # a 1 x 1 supernode with a zero diagonal: nothing to interchange with -> perturbed
    D = np.diag([2.0, 0.0, 3.0])
    S1 = sf.analyze(*dense_csc(D + 0.0), None, 1 << 30, "lu", False)
    Lsx, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S1, tol=0.1)
    assert nper >= 1 and info == 0


def test_oracle_reproduces_the_committed_lu_vectors(oracle):
    """tests/golden/lu_small.json on the CPU: the product's analysis gives the committed integers, the oracle (both the no-pivot path
    and the pivoted one) the committed pivots and values -- the same file the GPU test compares the HIP path with"""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lu_small.json")) as f:
        G = json.load(f)
    for name, g in G.items():
        S = sf.analyze(g["n"], g["Cp"], g["Ci"], g["Cx"], g["perm"], g["devSlotSize"], "lu", g["symmetric"])
        for k in ("Super", "Lsip", "Lsxp", "Lsi", "Perm"):
            assert np.array_equal(getattr(S, k), np.asarray(g[k], dtype=np.int64)), (name, k)
        if g["pivpos"] is None:
            Lsx, info, _ = oracle.lu_factorize(S)
            assert info == 0
        else:
            Lsx, info, pivpos, _, nper = oracle.lu_factorize_pivot(S, tol=g["tol"], perturb=g["perturb"])
            assert info == 0 and nper == g["perturbed"] and np.array_equal(pivpos, np.asarray(g["pivpos"]))
        assert rel_err(Lsx, np.asarray(g["Lsx"])) <= 1e-13, name

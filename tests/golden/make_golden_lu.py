"""Generates tests/golden/lu_small.json: LU vectors, no-pivot and pivoted.

The reference cannot run in this image and ships no fixtures, so these vectors come from the CPU oracle (oracle/, built-in C
loops) and are accepted only after agreeing with something independent of it:
  no-pivot cases  : dense no-pivot LU of the same permuted matrix (numpy), 1e-13;
  pivoted, 1 block: scipy.linalg.lu (LAPACK partial pivoting = tol 1 on a front of <= 64 columns), pivots exact, values 1e-12;
  pivoted, blocks : the numpy statement of the block-restricted threshold rule in tests/test_lu_pivot_oracle.py, pivots exact;
  pivoted, sparse : the solve with the recorded interchanges reproduces b.
The pivoting rule is the product's own (the reference never pivots: LU/Source/SparseFrame.c:2653, :3344, :589-673 disabled):
PARITY UNPINNED by construction for those cases.  Run from the repo root:  python tests/golden/make_golden_lu.py
"""
import json
import os
import sys

import numpy as np
import scipy.linalg

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from util import gen, nd_perm_py, rel_err  # noqa: E402
from test_lu import dense_lu_nopiv, reference_layout_from_dense  # noqa: E402
from test_lu_pivot_oracle import dense_csc, block_rule_numpy, SQRT_EPS  # noqa: E402
import oracle  # noqa: E402


class Sym(dict):
    __getattr__ = dict.__getitem__


def analyze(n, Cp, Ci, Cx, perm, slot, symm):
    S = Sym(oracle.symbolic.analyze(n, Cp, Ci, Cx, perm, slot, lu=True, symmetric=symm))
    for k in ("Super", "Lsip", "Lsxp", "Lsi", "Perm"):
        S[k] = np.asarray(S[k], dtype=np.int64)
    return S


def record(n, Cp, Ci, Cx, perm, slot, symm, S, Lsx, **extra):
    d = dict(n=int(n), Cp=np.asarray(Cp).tolist(), Ci=np.asarray(Ci).tolist(), Cx=np.asarray(Cx).tolist(),
             perm=None if perm is None else np.asarray(perm).tolist(), devSlotSize=int(slot), symmetric=bool(symm),
             Super=S["Super"].tolist(), Lsip=S["Lsip"].tolist(), Lsxp=S["Lsxp"].tolist(), Lsi=S["Lsi"].tolist(),
             Perm=S["Perm"].tolist(), nsuper=int(S["nsuper"]), Lsx=np.asarray(Lsx).tolist())
    d.update(extra)
    return d


def main():
    oracle.blas_init("builtin")
    out = {}
    # ---- no-pivot ------------------------------------------------------------------------------------------------
    cases = []
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(12, 9, 1, seed=2)
    cases.append(("nopiv_st2d_12x9_nd", n, Cp, Ci, Cx, nd_perm_py(12, 9, 1), 1 << 30, False))
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(5, 5, 5, seed=3)
    cases.append(("nopiv_st3d_5_nd_smallslot", n, Cp, Ci, Cx, nd_perm_py(5, 5, 5), 9000, False))
    n, Cp, Ci, Cx = gen.laplacian_lower(5, 5, 5)
    cases.append(("nopiv_sym_lap3d_5_via_lu", n, Cp, Ci, Cx, nd_perm_py(5, 5, 5), 1 << 30, True))
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(40, 1, 1, extra_per_row=3, seed=5)
    cases.append(("nopiv_rand_40", n, Cp, Ci, Cx, None, 1 << 30, False))
    for name, n, Cp, Ci, Cx, perm, slot, symm in cases:
        S = analyze(n, Cp, Ci, Cx, perm, slot, symm)
        Lsx, info, _ = oracle.lu_factorize(S)
        assert info == 0
        A = gen.dense_from_lower(n, Cp, Ci, Cx) if symm else gen.dense_from_csc(n, Cp, Ci, Cx)
        Ap = A[np.ix_(S.Perm, S.Perm)]
        assert rel_err(Lsx, reference_layout_from_dense(S, dense_lu_nopiv(Ap))) <= 1e-13, name
        out[name] = record(n, Cp, Ci, Cx, perm, slot, symm, S, Lsx, tol=0.0, perturb=0.0, pivpos=None, perturbed=0)
    # ---- pivoted, one block: LAPACK --------------------------------------------------------------------------------
    rng = np.random.default_rng(2)
    A = rng.uniform(-1, 1, (33, 33))
    n, Cp, Ci, Cx = dense_csc(A)
    S = analyze(n, Cp, Ci, Cx, None, 1 << 30, False)
    Lsx, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=1.0)
    p, l, u = scipy.linalg.lu(A)
    P = Lsx[:n * n].reshape(n, n).T
    assert info == 0 and nper == 0 and np.array_equal(np.argmax(p, axis=0), pivinv)
    assert np.abs(np.tril(P, -1) + np.eye(n) - l).max() <= 1e-12 and np.abs(np.triu(P) - u).max() <= 1e-12 * np.abs(u).max()
    out["piv_dense_33_tol1"] = record(n, Cp, Ci, Cx, None, 1 << 30, False, S, Lsx, tol=1.0, perturb=SQRT_EPS,
                                      pivpos=pivpos.tolist(), perturbed=nper)
    # ---- pivoted, two blocks in one supernode -- ------------------------------------------------------------------------
    rng = np.random.default_rng(6)
    A = rng.uniform(-1, 1, (100, 100))
    n, Cp, Ci, Cx = dense_csc(A)
    S = analyze(n, Cp, Ci, Cx, None, 1 << 30, False)
    assert S.nsuper == 1
    Lsx, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=0.1)
    want, wpos, wper = block_rule_numpy(A, 0.1, SQRT_EPS * np.abs(A).max())
    assert info == 0 and nper == wper and np.array_equal(pivpos, wpos)
    assert np.abs(Lsx[:n * n].reshape(n, n).T - want).max() <= 1e-11 * np.abs(want).max()
    out["piv_dense_100_tol01"] = record(n, Cp, Ci, Cx, None, 1 << 30, False, S, Lsx, tol=0.1, perturb=SQRT_EPS,
                                        pivpos=pivpos.tolist(), perturbed=nper)
    # ---- pivoted, sparse non-dominant --------------------------------------------------------------------------------
    N = 6
    n, Cp, Ci, Cx = gen.unsymmetric_general(N, N, N, seed=21)
    perm = nd_perm_py(N, N, N)
    S = analyze(n, Cp, Ci, Cx, perm, 1 << 30, False)
    Lsx, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=1.0)
    Ap = gen.dense_from_csc(n, Cp, Ci, Cx)[np.ix_(S.Perm, S.Perm)]
    b = 1 + np.arange(n) / n
    x = oracle.lu_solve_pivot(S, Lsx, pivpos, b)
    for _ in range(3):
        x = x + oracle.lu_solve_pivot(S, Lsx, pivpos, b - Ap @ x)
    assert info == 0 and np.abs(Ap @ x - b).max() <= 1e-10 * np.abs(Ap).sum(axis=0).max() * np.abs(x).max()
    assert np.count_nonzero(pivpos != np.arange(n)) > 0
    out["piv_general_6_tol1"] = record(n, Cp, Ci, Cx, perm, 1 << 30, False, S, Lsx, tol=1.0, perturb=SQRT_EPS,
                                       pivpos=pivpos.tolist(), perturbed=nper)
    with open(os.path.join(HERE, "lu_small.json"), "w") as f:
        json.dump(out, f)
    print("wrote", {k: (v["n"], len(v["Lsx"])) for k, v in out.items()})


if __name__ == "__main__":
    main()

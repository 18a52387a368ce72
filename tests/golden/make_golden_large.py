"""Generates tests/golden/large_sampled.json: fixtures that reach the kernels the small vectors cannot -- a Cholesky panel wider
than one 512-column outer block (k_gemm<0>, several fused k_step launches per block), a config-3-like 2-D stencil, a pivoted LU
front of several 64-column blocks and a pivoted sparse LU.  Too large to commit whole, so (SURVEY 8c's own suggestion): SHA-256 of
every integer array (inputs and symbolic outputs: bit-exact), sampled entries of the factor (positions + values), two aggregates
(sum of log|diagonal|, sum of |entries| over the defined entries), and the full pivot sequence of the pivoted cases.

The reference cannot run in this image and ships no fixtures, so the values come from the CPU oracle (oracle/, built-in C loops)
and are accepted only after agreeing with something that shares no code with it:
  chol_lap3d_24        dense LAPACK Cholesky of the permuted matrix (numpy), 1e-12
  chol_stencil2d_200   SuperLU (scipy.sparse.linalg.splu, natural order, no pivoting, symmetric mode): L sqrt(diag U), 1e-12
  nopiv_lu_stencil_16  SuperLU in natural order without pivoting: the packed L \ U panels entry by entry, 1e-12 (the reference's LU never pivots)
  piv_dense_200_tol01  the numpy statement of the block-restricted threshold rule (tests/test_lu_pivot_oracle.py), pivots exact
  piv_zero_diag_12     the solve with the recorded interchanges against SuperLU's solution of the same system
The pivoting rule is the product's own (the reference never pivots: LU/Source/SparseFrame.c:2653, :3344): PARITY UNPINNED by
construction for those two.  Run from the repo root:  python tests/golden/make_golden_large.py   (about a minute, 2 GB)
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from util import sf, gen, nd_perm_py, rel_err  # noqa: E402
from golden_large import CASES, build_case, sha, defined_mask  # noqa: E402
from test_lu_pivot_oracle import block_rule_numpy, SQRT_EPS  # noqa: E402
import oracle  # noqa: E402

SAMPLES = 3000


def panel_positions(S, lu=False):
    """(row, column) of every entry of the supernodal value array, in array order (Cholesky layout / LU packed layout L:2514-2517:
    rows [0, nsrow) of a column = L part, rows [nsrow, 2 nsrow - nscol) = U12^T, i.e. entry (column, row') of U)"""
    rows, cols, upper = [], [], []
    Super, Lsip, Lsi = np.asarray(S.Super), np.asarray(S.Lsip), np.asarray(S.Lsi)
    for s in range(S.nsuper):
        nscol, r = Super[s + 1] - Super[s], Lsi[Lsip[s]:Lsip[s + 1]]
        nsrow = len(r)
        c = np.arange(Super[s], Super[s + 1])
        if not lu:
            rows.append(np.tile(r, nscol)); cols.append(np.repeat(c, nsrow)); upper.append(np.zeros(nsrow * nscol, bool))
        else:
            col_rows = np.concatenate([r, r[nscol:]])
            rows.append(np.tile(col_rows, nscol)); cols.append(np.repeat(c, 2 * nsrow - nscol))
            upper.append(np.tile(np.concatenate([np.zeros(nsrow, bool), np.ones(nsrow - nscol, bool)]), nscol))
    return np.concatenate(rows), np.concatenate(cols), np.concatenate(upper)


def main():
    oracle.blas_init("builtin")
    out = {}
    for name in CASES:
        c = build_case(name)
        S = c["sym"]
        n = c["n"]
        rec = dict(c["spec"])
        rec.update(n=int(n), nsuper=int(S.nsuper), xsize=int(S.xsize),
                   sha_inputs={k: sha(c[k]) for k in ("Cp", "Ci", "Cx") if c[k] is not None},
                   sha_perm=None if c["perm"] is None else sha(np.asarray(c["perm"], dtype=np.int64)),
                   sha_symbolic={k: sha(np.asarray(getattr(S, k), dtype=np.int64)) for k in ("Super", "Lsip", "Lsxp", "Lsi", "Perm")})
        if c["method"] == "cholesky":
            Lsx, info, _ = oracle.chol_factorize(S)
            assert info == 0
            mask = defined_mask(S)
            r, col, _ = panel_positions(S)
            if name == "chol_lap3d_24":
                A = sp.coo_matrix((S.Lx, (S.Li, np.repeat(np.arange(n), np.diff(S.Lp)))), shape=(n, n)).toarray()
                A = A + np.tril(A, -1).T
                Ld = np.linalg.cholesky(A)
                want = Ld[r, col]
                del A, Ld
                how = "dense LAPACK Cholesky (numpy.linalg.cholesky) of the permuted matrix"
            else:
                lc = np.repeat(np.arange(n), np.diff(S.Lp))
                Al = sp.coo_matrix((S.Lx, (S.Li, lc)), shape=(n, n)).tocsc()
                A = (Al + sp.tril(Al, -1).T).tocsc()
                lu = spla.splu(A, permc_spec="NATURAL", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
                assert np.array_equal(lu.perm_r, np.arange(n)) and np.array_equal(lu.perm_c, np.arange(n))
                Lc = (lu.L @ sp.diags(np.sqrt(lu.U.diagonal()))).tocsr()
                want = np.asarray(Lc[r, col]).ravel()
                how = "SuperLU (scipy splu, natural order, no pivoting, symmetric mode): L sqrt(diag U)"
            err = rel_err(Lsx, want, mask)
            assert err <= 1e-12, (name, err)
            rec["accepted_by"] = f"{how}: max rel err {err:.2e}"
            diag_idx = np.asarray(S.Lsxp)[:-1][np.asarray(S.SuperMap)] + \
                (np.arange(n) - np.asarray(S.Super)[np.asarray(S.SuperMap)]) * (np.diff(S.Lsip)[np.asarray(S.SuperMap)] + 1)
        elif name == "nopiv_lu_stencil_16":
            # the reference's own LU behaviour (no pivoting): accepted against SuperLU in natural order with diag_pivot_thresh = 0
            # (it keeps the diagonal pivots of this diagonally dominant matrix: perm_r = identity is asserted), entry by entry of the
            # packed panels (L:2514-2517: rows [0, nscol) = L11 \ U11, [nscol, nsrow) = L21, [nsrow, 2 nsrow - nscol) = U12^T)
            oracle.blas_init("auto", threads=4)
            Lsx, info, _ = oracle.lu_factorize(S)
            oracle.blas_init("builtin")
            assert info == 0
            mask = np.ones(S.xsize, dtype=bool)
            Ap = sp.csc_matrix((c["Cx"], c["Ci"], c["Cp"]), shape=(n, n))[S.Perm][:, S.Perm].tocsc()
            lu = spla.splu(Ap, permc_spec="NATURAL", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
            assert np.array_equal(lu.perm_r, np.arange(n)) and np.array_equal(lu.perm_c, np.arange(n))
            Lm, Um = lu.L.tocsr(), lu.U.tocsr()
            r, col, upper = panel_positions(S, lu=True)
            inblock_upper = ~upper & (r <= col)                     # U11 (diagonal included) sits in the L11 rows of the packed panel
            want = np.where(upper, np.asarray(Um[col, r]).ravel(),                                  # U12^T: U(column, row')
                            np.where(inblock_upper, np.asarray(Um[r, col]).ravel(), np.asarray(Lm[r, col]).ravel()))
            err = rel_err(Lsx, want, mask)
            assert err <= 1e-12, (name, err)
            rec["accepted_by"] = f"SuperLU (scipy splu, natural order, no pivoting): packed L \\ U panels, max rel err {err:.2e}"
            rec.update(pivpos=None, perturbed=0)
            Xp = np.asarray(S.Lsxp)[:-1]
            sm = np.asarray(S.SuperMap)
            diag_idx = Xp[sm] + (np.arange(n) - np.asarray(S.Super)[sm]) * (2 * np.diff(S.Lsip)[sm] - np.diff(S.Super)[sm] + 1)
        else:
            tol = rec["tol"]
            Lsx, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=tol)
            assert info == 0 and np.count_nonzero(pivpos != np.arange(n)) > 0
            mask = np.ones(S.xsize, dtype=bool)
            Ap = gen.dense_from_csc(n, c["Cp"], c["Ci"], c["Cx"])[np.ix_(S.Perm, S.Perm)]
            if name == "piv_dense_200_tol01":
                assert S.nsuper == 1
                want, wpos, wper = block_rule_numpy(Ap, tol, SQRT_EPS * np.abs(Ap).max())
                assert nper == wper and np.array_equal(pivpos, wpos)
                err = float(np.abs(Lsx[:n * n].reshape(n, n).T - want).max() / np.abs(want).max())
                assert err <= 1e-10, err
                rec["accepted_by"] = f"numpy statement of the block-restricted rule: pivots equal, values max rel err {err:.2e}"
            else:
                b = 1 + np.arange(n) / n
                x = oracle.lu_solve_pivot(S, Lsx, pivpos, b)
                for _ in range(2):          # multipliers of 2e6 next to the zeroed entries: two refinement steps, as a user of such a factor would
                    x = x + oracle.lu_solve_pivot(S, Lsx, pivpos, b - Ap @ x)
                xs = spla.splu(sp.csc_matrix(Ap)).solve(b)
                err = float(np.abs(x - xs).max() / np.abs(xs).max())
                assert err <= 1e-12, err
                rec["accepted_by"] = (f"solve with the recorded interchanges (+ 2 refinement steps) vs SuperLU's (partial pivoting) solution: "
                                      f"max rel diff {err:.2e}")
            rec.update(pivpos=pivpos.tolist(), perturbed=int(nper))
            Xp = np.asarray(S.Lsxp)[:-1]
            sm = np.asarray(S.SuperMap)
            diag_idx = Xp[sm] + (np.arange(n) - np.asarray(S.Super)[sm]) * (2 * np.diff(S.Lsip)[sm] - np.diff(S.Super)[sm] + 1)
        rng = np.random.default_rng(2026)
        cand = np.flatnonzero(mask)
        pick = rng.choice(cand, size=min(SAMPLES, cand.size), replace=False)
        # plus entries of the widest supernode right of its first 512 columns (second outer block and later)
        w = int(np.argmax(np.diff(S.Super)))
        nscol_w, nsrow_w = int(np.diff(S.Super)[w]), int(np.diff(S.Lsip)[w])
        ld = nsrow_w if c["method"] == "cholesky" else 2 * nsrow_w - nscol_w
        if nscol_w > 512:
            lo, hi = int(S.Lsxp[w]) + 512 * ld, int(S.Lsxp[w + 1])
            extra = lo + rng.choice(hi - lo, size=500, replace=False)
            pick = np.concatenate([pick, extra[mask[extra]]])
        pick = np.unique(np.concatenate([pick, diag_idx[rng.choice(n, size=min(n, 300), replace=False)]]))
        rec.update(sample_idx=pick.tolist(), sample_val=Lsx[pick].tolist(), max_abs=float(np.abs(Lsx[mask]).max()),
                   diag_logsum=float(np.log(np.abs(Lsx[diag_idx])).sum()), abs_sum=float(np.abs(Lsx[mask]).sum()),
                   widest_supernode_columns=nscol_w)
        out[name] = rec
        print(name, "n", n, "nsuper", S.nsuper, "xsize", S.xsize, "widest", nscol_w, "samples", len(pick), rec["accepted_by"])
    with open(os.path.join(HERE, "large_sampled.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()

"""The cases of tests/golden/large_sampled.json (made by tests/golden/make_golden_large.py) and the checks a factor has to pass
against them; shared by the generator, the GPU tests (HIP path) and the CPU tests (oracle).  The inputs are regenerated from the
deterministic generators and pinned by the fixture's SHA-256 sums."""
import hashlib
import json
import os

import numpy as np

from util import sf, gen, nd_perm_py

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ("chol_lap3d_24", "chol_stencil2d_200", "nopiv_lu_stencil_16", "piv_dense_200_tol01", "piv_zero_diag_12")


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def load():
    with open(os.path.join(HERE, "golden", "large_sampled.json")) as f:
        return json.load(f)


def build_case(name):
    """inputs + the product's symbolic analysis of one case"""
    if name == "chol_lap3d_24":          # root separator 576 columns: two outer blocks
        n, Cp, Ci, Cx = gen.laplacian_lower(24, 24, 24)
        perm, slot, method, symm = nd_perm_py(24, 24, 24), 1 << 30, "cholesky", True
        spec = dict(generator="laplacian_lower(24, 24, 24)", ordering="nd_perm_py(24, 24, 24)")
    elif name == "chol_stencil2d_200":   # config-3-like: 2-D 21-point random SPD stencil, 2-line separators
        n, Cp, Ci, Cx = gen.stencil_spd_lower(200, 200)
        perm, slot, method, symm = sf.grid_nd_perm(200, 200, 1, 3, 2), sf.REFERENCE_SLOT_1GPU, "cholesky", True
        spec = dict(generator="stencil_spd_lower(200, 200)", ordering="grid_nd_perm(200, 200, 1, leaf 3, separator width 2)")
    elif name == "nopiv_lu_stencil_16":  # the reference's LU (no pivoting, L:2653): unsymmetric 19-point stencil, root of > 1000 columns
        N = 16
        n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=13)
        perm, slot, method, symm = nd_perm_py(N, N, N), 1 << 30, "lu", False
        spec = dict(generator="unsymmetric_stencil(16, 16, 16, seed=13)", ordering="nd_perm_py(16, 16, 16)", tol=0.0)
    elif name == "piv_dense_200_tol01":  # one front of four 64-column blocks
        A = np.random.default_rng(11).uniform(-1, 1, (200, 200))
        n = 200
        Cp, Ci, Cx = np.arange(0, n * n + 1, n, dtype=np.int64), np.tile(np.arange(n, dtype=np.int64), n), np.ascontiguousarray(A.T).ravel()
        perm, slot, method, symm = None, 1 << 30, "lu", False
        spec = dict(generator="default_rng(11).uniform(-1, 1, (200, 200)), dense CSC", ordering="identity", tol=0.1)
    elif name == "piv_zero_diag_12":     # sparse, every 7th diagonal entry of the permuted matrix zeroed
        N = 12
        n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=9)
        perm = nd_perm_py(N, N, N)
        S0 = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
        Cx = Cx.copy()
        cols = np.repeat(np.arange(n), np.diff(Cp))
        for j in range(0, n, 7):
            g = S0.Perm[j]
            Cx[(Ci == g) & (cols == g)] = 0.0
        slot, method, symm = 1 << 30, "lu", False
        spec = dict(generator="unsymmetric_stencil(12, 12, 12, seed=9), diagonal entries of permuted columns 0, 7, 14, ... zeroed",
                    ordering="nd_perm_py(12, 12, 12)", tol=0.1)
    else:
        raise KeyError(name)
    spec.update(devSlotSize=int(slot), method=method)
    sym = sf.analyze(n, Cp, Ci, Cx, perm, slot, method, symm)
    return dict(n=n, Cp=np.asarray(Cp, dtype=np.int64), Ci=np.asarray(Ci, dtype=np.int64), Cx=np.asarray(Cx, dtype=np.float64),
                perm=perm, method=method, sym=sym, spec=spec)


def defined_mask(S):
    """Cholesky: the entries the reference defines = everything but the strict upper triangles of the diagonal blocks"""
    mask = np.ones(int(S.xsize), dtype=bool)
    Super, Lsip, Lsxp = np.asarray(S.Super), np.asarray(S.Lsip), np.asarray(S.Lsxp)
    for s in range(int(S.nsuper)):
        nscol, nsrow = int(Super[s + 1] - Super[s]), int(Lsip[s + 1] - Lsip[s])
        if nscol > 1:
            blk = mask[Lsxp[s]:Lsxp[s] + nscol * nsrow].reshape(nscol, nsrow)      # [column, row]
            blk[:, :nscol][np.triu_indices(nscol, 1)[::-1]] = False              # row < column
    return mask


def check_inputs_and_symbolic(name, g, c):
    """inputs regenerated bit for bit, integer outputs of the analysis bit-exact (SHA-256)"""
    S = c["sym"]
    assert (int(c["n"]), int(S.nsuper), int(S.xsize)) == (g["n"], g["nsuper"], g["xsize"]), name
    for k, h in g["sha_inputs"].items():
        assert sha(c[k]) == h, (name, k)
    if g["sha_perm"] is not None:
        assert sha(np.asarray(c["perm"], dtype=np.int64)) == g["sha_perm"], name
    for k, h in g["sha_symbolic"].items():
        assert sha(np.asarray(getattr(S, k), dtype=np.int64)) == h, (name, k)


def check_factor(name, g, S, Lsx, tol):
    """sampled entries and the two aggregates"""
    idx, val = np.asarray(g["sample_idx"]), np.asarray(g["sample_val"])
    err = float(np.max(np.abs(Lsx[idx] - val)) / g["max_abs"])
    assert err <= tol, (name, "samples", err)
    lu = g["method"] == "lu"
    n = g["n"]
    sm, Sup, Xp = np.asarray(S.SuperMap), np.asarray(S.Super), np.asarray(S.Lsxp)[:-1]
    nsrow, nscol = np.diff(S.Lsip), np.diff(S.Super)
    ld = (2 * nsrow - nscol) if lu else nsrow
    diag_idx = Xp[sm] + (np.arange(n) - Sup[sm]) * (ld[sm] + 1)
    logsum = float(np.log(np.abs(Lsx[diag_idx])).sum())
    assert abs(logsum - g["diag_logsum"]) <= 1e-10 * max(1.0, abs(g["diag_logsum"])), (name, "log|diag|", logsum, g["diag_logsum"])
    mask = np.ones(len(Lsx), dtype=bool) if lu else defined_mask(S)
    asum = float(np.abs(Lsx[mask]).sum())
    assert abs(asum - g["abs_sum"]) <= max(tol, 1e-11) * g["abs_sum"], (name, "sum |entries|", asum, g["abs_sum"])
    return err

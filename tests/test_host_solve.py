"""The struct path's HOST solve on several threads (csrc/sf_host_solve.h) against the scalar sweep it replaces for large factors
(the reference's own loops, C:3036-3139 / L:3592-3700) and against the oracle: same Lsx (made by the oracle), same right-hand side.
No device is involved: this is where SparseFrame_solve_supernodal goes when no factor is resident -- after an out-of-core
factorization, with several handlers, or when the caller changed Lsx."""
import ctypes as C

import numpy as np
import pytest

from util import sf, gen, nd_perm_py


def _solve(mi_cls, n, Cp, Ci, Cx, perm, symmetric, Lsx, threads, monkeypatch, pivinv=None):
    monkeypatch.setenv("SF_HOST_SOLVE_MIN", "0")
    monkeypatch.setenv("SF_HOST_SOLVE_THREADS", str(threads))
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    mi = mi_cls()
    mi.set_csc(n, Cp, Ci, Cx, symmetric=symmetric)
    mi.set_perm(perm)
    mi.analyze(common)
    C.memmove(mi.c.Lsx, Lsx.ctypes.data, Lsx.nbytes)
    if pivinv is not None:
        libc = C.CDLL(None)
        libc.malloc.restype = C.c_void_p
        buf = libc.malloc(C.c_size_t(8 * n))
        C.memmove(buf, np.ascontiguousarray(pivinv, dtype=np.int64).ctypes.data, 8 * n)
        mi.c.PivInv = C.cast(buf, type(mi.c.PivInv))          # released by SparseFrame_cleanup_matrix like the library's own
    res = mi.validate()
    x = mi.array("Xx", n).copy()
    mi.cleanup()
    common.close()
    return res, x


def chol_cases():
    out = []
    N = 20
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    out.append(("lap3d_20", n, Cp, Ci, Cx, nd_perm_py(N, N, N)))
    n, Cp, Ci, Cx = gen.stencil_spd_lower(70, 70)
    out.append(("stencil2d_70", n, Cp, Ci, Cx, sf.grid_nd_perm(70, 70, 1, 3, 2)))
    n, Cp, Ci, Cx = gen.arrow_spd_lower(300, 40)            # one wide root above a forest of singletons
    out.append(("arrow_300", n, Cp, Ci, Cx, None))
    n, Cp, Ci, Cx = gen.laplacian_lower(50)                 # a chain: nothing to cut, everything is "top" or one subtree
    out.append(("chain_50", n, Cp, Ci, Cx, None))
    return out


@pytest.mark.parametrize("case", chol_cases(), ids=lambda c: c[0])
def test_threaded_cholesky_solve_equals_the_scalar_sweep(oracle, monkeypatch, case):
    name, n, Cp, Ci, Cx, perm = case
    S = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30)
    Lsx, info, _ = oracle.chol_factorize(S)
    assert info == 0
    Lsx = np.where(oracle.lower_mask(S), Lsx, 0.0)
    res1, x1 = _solve(sf.MatrixInfo, n, Cp, Ci, Cx, perm, True, Lsx, 1, monkeypatch)
    assert res1 <= 1e-13
    for T in (2, 5, 16):
        resT, xT = _solve(sf.MatrixInfo, n, Cp, Ci, Cx, perm, True, Lsx, T, monkeypatch)
        assert resT <= 1e-13, (name, T, resT)
        assert np.max(np.abs(xT - x1)) <= 1e-12 * np.max(np.abs(x1)), (name, T)


@pytest.mark.parametrize("pivot", [False, True], ids=["no_pivot", "interchanges"])
def test_threaded_lu_solve_equals_the_scalar_sweep(oracle, monkeypatch, pivot):
    if pivot:
        import golden_large
        c = golden_large.build_case("piv_zero_diag_12")
        n, Cp, Ci, Cx, perm, S = c["n"], c["Cp"], c["Ci"], c["Cx"], c["perm"], c["sym"]
        Lsx, info, pivpos, pivinv, nper = oracle.lu_factorize_pivot(S, tol=0.1)
        assert info == 0 and np.count_nonzero(pivpos != np.arange(n)) > 0
        pivinv = pivpos         # (the struct's PivInv field holds what sf_lu_plan_get_pivots reports: row -> position)
    else:
        N = 14
        n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, extra_per_row=0, drop=0.2, seed=4)
        perm = sf.grid_nd_perm(N, N, N, 3, 2)
        S = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
        Lsx, info, _ = oracle.lu_factorize(S)
        pivinv = None
        assert info == 0
    res1, x1 = _solve(sf.LUMatrixInfo, n, Cp, Ci, Cx, perm, False, Lsx, 1, monkeypatch, pivinv)
    assert res1 <= (1e-9 if pivot else 1e-13)
    for T in (3, 16):
        resT, xT = _solve(sf.LUMatrixInfo, n, Cp, Ci, Cx, perm, False, Lsx, T, monkeypatch, pivinv)
        assert resT <= max(2 * res1, 1e-13), (T, resT, res1)
        assert np.max(np.abs(xT - x1)) <= (1e-8 if pivot else 1e-12) * np.max(np.abs(x1)), T

"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

CPU oracle of the supernodal Cholesky path: `symbolic` (pure-Python restatement of the reference's
analysis) and the ctypes binding of libsf_oracle.so (C restatement of the reference's CPU numeric
path, oracle/sf_oracle_numeric.c).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this package.
"""
import ctypes as C
import glob
import os

import numpy as np

from . import symbolic  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsf_oracle.so")
_lp = C.POINTER(C.c_int64)
_dp = C.POINTER(C.c_double)
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise ImportError(f"{_LIB} missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = C.CDLL(_LIB)
        _lib.sfo_blas_init.argtypes = [C.c_char_p]
        _lib.sfo_blas_init.restype = C.c_int
        _lib.sfo_blas_name.restype = C.c_char_p
        _lib.sfo_blas_kind.restype = C.c_int
        _lib.sfo_blas_set_threads.argtypes = [C.c_int]
        _lib.sfo_blas_get_threads.restype = C.c_int
        _lib.sfo_chol_factorize.argtypes = [C.c_int64, C.c_int64] + [_lp] * 7 + [_dp, _lp, C.c_int64, C.c_int64, _dp, _dp]
        _lib.sfo_chol_factorize.restype = C.c_int
        _lib.sfo_chol_solve.argtypes = [C.c_int64] + [_lp] * 4 + [_dp, C.c_int64, _dp, _dp]
        _lib.sfo_chol_solve.restype = None
        _lib.sfo_chol_residual.argtypes = [C.c_int64, _lp, _lp, _dp, C.c_int64] + [_lp] * 4 + [_dp, _dp]
        _lib.sfo_chol_residual.restype = C.c_double
        _lib.sfo_lu_factorize.argtypes = [C.c_int64, C.c_int64] + [_lp] * 7 + [_dp, _lp, _lp, _dp, _lp, C.c_int64, C.c_int64, _dp, _dp]
        _lib.sfo_lu_factorize.restype = C.c_int
        _lib.sfo_lu_solve.argtypes = [C.c_int64] + [_lp] * 4 + [_dp, C.c_int64, _dp, _dp]
        _lib.sfo_lu_solve.restype = None
        _lib.sfo_lu_residual.argtypes = [C.c_int64, _lp, _lp, _dp, _lp, _lp, _dp, C.c_int64] + [_lp] * 4 + [_dp, _dp]
        _lib.sfo_lu_residual.restype = C.c_double
    return _lib


def find_openblas():
    """the OpenBLAS scipy bundles (LP64), else the one numpy bundles (ILP64), else None"""
    try:
        import scipy
        hits = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
        if hits:
            return sorted(hits)[0]
    except Exception:
        pass
    hits = glob.glob(os.path.join(os.path.dirname(os.path.dirname(np.__file__)), "numpy.libs", "libscipy_openblas*.so"))
    return sorted(hits)[0] if hits else None


def blas_init(which="auto", threads=None):
    """which: 'auto' (OpenBLAS if found, else built-in loops), 'builtin', or a library path."""
    lib = _load()
    path = None
    if which == "auto":
        path = find_openblas()
    elif which != "builtin":
        path = which
    kind = lib.sfo_blas_init(path.encode() if path else None)
    if kind < 0:
        if which == "auto":
            kind = lib.sfo_blas_init(None)
        else:
            raise RuntimeError(f"cannot bind BLAS from {path}")
    if threads is not None:
        lib.sfo_blas_set_threads(int(threads))
    return {"kind": kind, "name": lib.sfo_blas_name().decode(), "threads": lib.sfo_blas_get_threads()}


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def chol_factorize(sym):
    """sym: any object/dict exposing n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, Lx,
    LeafQueue, nsleaf, csize, xsize.  Returns (Lsx, info, stats)."""
    lib = _load()
    g = (lambda k: sym[k]) if isinstance(sym, dict) else (lambda k: getattr(sym, k))
    arrs = [_i64(g(k)) for k in ("Super", "SuperMap", "Lsip", "Lsi", "Lsxp", "Lp", "Li")]
    Lx = _f64(g("Lx"))
    LeafQueue = _i64(g("LeafQueue"))
    Lsx = np.empty(max(int(g("xsize")), 1), dtype=np.float64)
    stats = np.zeros(6)
    info = lib.sfo_chol_factorize(int(g("n")), int(g("nsuper")), *[a.ctypes.data_as(_lp) for a in arrs],
                                  Lx.ctypes.data_as(_dp), LeafQueue.ctypes.data_as(_lp), int(g("nsleaf")),
                                  int(g("csize")), Lsx.ctypes.data_as(_dp), stats.ctypes.data_as(_dp))
    names = ("flops_syrk", "flops_gemm", "flops_potrf", "flops_trsm", "scatter_elems", "seconds")
    return Lsx[:int(g("xsize"))], info, dict(zip(names, stats.tolist()))


def chol_solve(sym, Lsx, b):
    lib = _load()
    g = (lambda k: sym[k]) if isinstance(sym, dict) else (lambda k: getattr(sym, k))
    arrs = [_i64(g(k)) for k in ("Super", "Lsip", "Lsi", "Lsxp")]
    Lsx, b = _f64(Lsx), _f64(b)
    x = np.empty_like(b)
    lib.sfo_chol_solve(int(g("nsuper")), *[a.ctypes.data_as(_lp) for a in arrs], Lsx.ctypes.data_as(_dp),
                       int(g("n")), b.ctypes.data_as(_dp), x.ctypes.data_as(_dp))
    return x


def chol_residual(sym, Lsx):
    """reference validate(): returns (residual, x)"""
    lib = _load()
    g = (lambda k: sym[k]) if isinstance(sym, dict) else (lambda k: getattr(sym, k))
    Lp, Li = _i64(g("Lp")), _i64(g("Li"))
    Lx = _f64(g("Lx"))
    arrs = [_i64(g(k)) for k in ("Super", "Lsip", "Lsi", "Lsxp")]
    Lsx = _f64(Lsx)
    x = np.empty(max(int(g("n")), 1))
    r = lib.sfo_chol_residual(int(g("n")), Lp.ctypes.data_as(_lp), Li.ctypes.data_as(_lp), Lx.ctypes.data_as(_dp),
                              int(g("nsuper")), *[a.ctypes.data_as(_lp) for a in arrs],
                              Lsx.ctypes.data_as(_dp), x.ctypes.data_as(_dp))
    return float(r), x[:int(g("n"))]


def lower_mask(sym):
    """boolean mask over Lsx selecting the entries the reference defines (the strict upper
    triangle of every diagonal block is unspecified)."""
    g = (lambda k: sym[k]) if isinstance(sym, dict) else (lambda k: getattr(sym, k))
    Super, Lsip, Lsxp = _i64(g("Super")), _i64(g("Lsip")), _i64(g("Lsxp"))
    mask = np.ones(int(g("xsize")), dtype=bool)
    for s in range(int(g("nsuper"))):
        nscol = int(Super[s + 1] - Super[s])
        nsrow = int(Lsip[s + 1] - Lsip[s])
        base = int(Lsxp[s])
        for c in range(1, nscol):
            mask[base + c * nsrow: base + c * nsrow + c] = False
    return mask


# ---------------------------------------------------------------------------------------------
# LU (no pivoting)
# ---------------------------------------------------------------------------------------------
def _getter(sym):
    return (lambda k: sym[k]) if isinstance(sym, dict) else (lambda k: getattr(sym, k))


def _u_arrays(g):
    """U by row; a symmetric input aliases U to L (reference LU/Source/SparseFrame.c:2718-2729)"""
    if int(g("symmetric")):
        return _i64(g("Lp")), _i64(g("Li")), _f64(g("Lx"))
    return _i64(g("Up")), _i64(g("Ui")), _f64(g("Ux"))


def lu_factorize(sym):
    """sym from the LU analysis (lu=1).  Returns (Lsx in the reference's (2*nsrow-nscol) x nscol layout, info, stats)."""
    lib = _load()
    g = _getter(sym)
    arrs = [_i64(g(k)) for k in ("Super", "SuperMap", "Lsip", "Lsi", "Lsxp", "Lp", "Li")]
    Lx = _f64(g("Lx"))
    Up, Ui, Ux = _u_arrays(g)
    LeafQueue = _i64(g("LeafQueue"))
    Lsx = np.empty(max(int(g("xsize")), 1), dtype=np.float64)
    stats = np.zeros(5)
    info = lib.sfo_lu_factorize(int(g("n")), int(g("nsuper")), *[a.ctypes.data_as(_lp) for a in arrs],
                                Lx.ctypes.data_as(_dp), Up.ctypes.data_as(_lp), Ui.ctypes.data_as(_lp), Ux.ctypes.data_as(_dp),
                                LeafQueue.ctypes.data_as(_lp), int(g("nsleaf")), int(g("csize")),
                                Lsx.ctypes.data_as(_dp), stats.ctypes.data_as(_dp))
    names = ("flops_gemm", "flops_getrf", "flops_trsm", "scatter_elems", "seconds")
    return Lsx[:int(g("xsize"))], info, dict(zip(names, stats.tolist()))


def lu_solve(sym, Lsx, b):
    lib = _load()
    g = _getter(sym)
    arrs = [_i64(g(k)) for k in ("Super", "Lsip", "Lsi", "Lsxp")]
    Lsx, b = _f64(Lsx), _f64(b)
    x = np.empty_like(b)
    lib.sfo_lu_solve(int(g("nsuper")), *[a.ctypes.data_as(_lp) for a in arrs], Lsx.ctypes.data_as(_dp),
                     int(g("n")), b.ctypes.data_as(_dp), x.ctypes.data_as(_dp))
    return x


def lu_residual(sym, Lsx):
    lib = _load()
    g = _getter(sym)
    Lp, Li, Lx = _i64(g("Lp")), _i64(g("Li")), _f64(g("Lx"))
    Up, Ui, Ux = _u_arrays(g)
    arrs = [_i64(g(k)) for k in ("Super", "Lsip", "Lsi", "Lsxp")]
    Lsx = _f64(Lsx)
    x = np.empty(max(int(g("n")), 1))
    r = lib.sfo_lu_residual(int(g("n")), Lp.ctypes.data_as(_lp), Li.ctypes.data_as(_lp), Lx.ctypes.data_as(_dp),
                            Up.ctypes.data_as(_lp), Ui.ctypes.data_as(_lp), Ux.ctypes.data_as(_dp),
                            int(g("nsuper")), *[a.ctypes.data_as(_lp) for a in arrs],
                            Lsx.ctypes.data_as(_dp), x.ctypes.data_as(_dp))
    return float(r), x[:int(g("n"))]


# ---------------------------------------------------------------------------------------------
# LU with threshold pivoting inside the 64 x 64 diagonal blocks (the product's own rule restated in scalar C: the reference
# never pivots, LU/Source/SparseFrame.c:2653, :3344, :589-673 disabled -- PARITY UNPINNED by construction)
# ---------------------------------------------------------------------------------------------
def lu_factorize_pivot(sym, tol=0.1, perturb=1.4901161193847656e-08):
    """Returns (Lsx in the reference's packed layout, info, pivpos, pivinv, perturbed_pivots).  `perturb` is relative to
    max |a_ij| of the analysed matrix, as in the product (sf_lu_plan_set_pivoting)."""
    lib = _load()
    g = _getter(sym)
    arrs = [_i64(g(k)) for k in ("Super", "SuperMap", "Lsip", "Lsi", "Lsxp", "Lp", "Li")]
    Lx = _f64(g("Lx"))
    Up, Ui, Ux = _u_arrays(g)
    LeafQueue = _i64(g("LeafQueue"))
    n = int(g("n"))
    amax = max(float(np.abs(Lx).max()) if len(Lx) else 0.0, float(np.abs(Ux).max()) if len(Ux) else 0.0)
    Lsx = np.empty(max(int(g("xsize")), 1), dtype=np.float64)
    pivpos = np.arange(max(n, 1), dtype=np.int64)
    pivinv = np.arange(max(n, 1), dtype=np.int64)
    nper = np.zeros(1, dtype=np.int64)
    lib.sfo_lu_factorize_pivot.restype = C.c_int
    info = lib.sfo_lu_factorize_pivot(C.c_int64(n), C.c_int64(int(g("nsuper"))), *[a.ctypes.data_as(_lp) for a in arrs],
                                      Lx.ctypes.data_as(_dp), Up.ctypes.data_as(_lp), Ui.ctypes.data_as(_lp), Ux.ctypes.data_as(_dp),
                                      LeafQueue.ctypes.data_as(_lp), C.c_int64(int(g("nsleaf"))), C.c_int64(int(g("csize"))),
                                      C.c_double(float(tol)), C.c_double(float(perturb) * amax),
                                      Lsx.ctypes.data_as(_dp), pivpos.ctypes.data_as(_lp), pivinv.ctypes.data_as(_lp),
                                      nper.ctypes.data_as(_lp))
    return Lsx[:int(g("xsize"))], int(info), pivpos[:n], pivinv[:n], int(nper[0])


def lu_solve_pivot(sym, Lsx, pivpos, b):
    lib = _load()
    g = _getter(sym)
    arrs = [_i64(g(k)) for k in ("Super", "Lsip", "Lsi", "Lsxp")]
    Lsx, b, pivpos = _f64(Lsx), _f64(b), _i64(pivpos)
    x = np.empty_like(b)
    lib.sfo_lu_solve_pivot.restype = None
    lib.sfo_lu_solve_pivot(C.c_int64(int(g("nsuper"))), *[a.ctypes.data_as(_lp) for a in arrs], Lsx.ctypes.data_as(_dp),
                           pivpos.ctypes.data_as(_lp), C.c_int64(int(g("n"))), b.ctypes.data_as(_dp), x.ctypes.data_as(_dp))
    return x

"""BASELINE.json's configurations at FULL size on the GPU (configs 2, 3, 5), checked through size-independent properties --
no oracle run is possible at these sizes in test time:
  * known answer: b = A e  =>  x = e (the all-ones vector), through the device solve;
  * the reference's validate() residual, computed on the device (sf_chol_plan_validate);
  * closed form: log det of the Dirichlet 7-point Laplacian = sum of log(6 - 2cos - 2cos - 2cos) = 2 sum log L_ii (config 2);
  * homogeneity: factor(4 A) solves to x / 4.
Sizes follow bench.py (config 2: 128^3 Cholesky; config 3: 2-D 1000 x 1000 21-point stencil; config 5: 79^3 unsymmetric LU)."""
import numpy as np
import pytest

from util import sf, gen

pytestmark = pytest.mark.gpu


def sym_matvec_ones(sym):
    """A e for the symmetric matrix whose lower triangle the analysis holds (permuted order)"""
    n = sym.n
    lc = np.repeat(np.arange(n), np.diff(sym.Lp))
    b = np.bincount(sym.Li, weights=sym.Lx, minlength=n)
    off = sym.Li != lc
    b += np.bincount(lc[off], weights=sym.Lx[off], minlength=n)
    return b


def test_config2_laplacian_128cubed_properties():
    N = 128
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N, 3, 1), sf.REFERENCE_SLOT_1GPU)
    plan = sf.CholPlan(sym)
    Lsx = plan.factorize_to_host(sym.Lx)                   # the drop-in boundary's path: overlapped copy-back, 30 GB
    assert plan.validate() <= 1e-13
    # closed form of log det
    c = 2.0 * np.cos(np.arange(1, N + 1) * np.pi / (N + 1))
    lam = 6.0 - c[:, None, None] - c[None, :, None] - c[None, None, :]
    logdet = float(np.sum(np.log(lam)))
    cols = np.arange(n)
    s = sym.SuperMap[cols]
    nsrow = (sym.Lsip[1:] - sym.Lsip[:-1])[s]
    d = Lsx[sym.Lsxp[s] + (cols - sym.Super[s]) * (nsrow + 1)]
    assert np.all(d > 0)
    assert abs(2.0 * float(np.sum(np.log(d))) - logdet) <= 1e-11 * abs(logdet)
    del Lsx
    # known answer
    x = plan.solve(sym_matvec_ones(sym))
    assert np.max(np.abs(x - 1.0)) <= 1e-10
    # homogeneity
    b = 1.0 + np.arange(n) / n
    x1 = plan.solve(b)
    plan.set_values(4.0 * sym.Lx)
    plan.factorize()
    x4 = plan.solve(b)
    assert np.max(np.abs(4.0 * x4 - x1)) <= 1e-12 * np.max(np.abs(x1))
    plan.close()


def test_config3_stencil2d_1000_properties():
    g = 1000
    n, Cp, Ci, Cx = gen.stencil_spd_lower(g, g)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(g, g, 1, 3, 2), sf.REFERENCE_SLOT_1GPU)
    plan = sf.CholPlan(sym)
    plan.set_values(sym.Lx)
    plan.factorize()
    res, x = plan.validate(return_x=True)
    assert res <= 1e-13
    assert abs(res - sf.validate_solution(sym, x)) <= 16 * np.finfo(np.float64).eps
    xe = plan.solve(sym_matvec_ones(sym))
    assert np.max(np.abs(xe - 1.0)) <= 1e-9
    # the factor's diagonal is positive and the factorization repeatable to rounding (atomic order only)
    L1 = plan.get_factor().copy()
    plan.factorize()
    L2 = plan.get_factor()
    assert np.max(np.abs(L1 - L2)) <= 1e-12 * np.max(np.abs(L1))
    plan.close()


def test_config5_unsymmetric_lu_79cubed_properties():
    g = 79
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(g, g, g, extra_per_row=0, seed=2024, drop=0.05)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(g, g, g, 3, 1), sf.REFERENCE_SLOT_1GPU, "lu", False)
    plan = sf.LUPlan(sym)
    plan.set_values(sym.Lx, sym.Ux)
    plan.factorize()
    assert plan.validate() <= 1e-13
    assert np.array_equal(plan.get_pivots(), np.arange(n))          # diagonally dominant: the natural pivots pass the threshold
    assert plan.stat("perturbed_pivots") == 0
    # known answer with the unsymmetric matrix: L part by column, U part by row
    lc = np.repeat(np.arange(n), np.diff(sym.Lp))
    b = np.bincount(sym.Li, weights=sym.Lx, minlength=n)
    ur = np.repeat(np.arange(n), np.diff(sym.Up))
    off = sym.Ui != ur
    b += np.bincount(ur[off], weights=sym.Ux[off], minlength=n)
    x = plan.solve(b)
    assert np.max(np.abs(x - 1.0)) <= 1e-10
    plan.close()

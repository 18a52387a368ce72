"""No-pivot supernodal LU (SURVEY 8a row a-LU).  CPU part: product analysis vs the oracle's restatement (bit-exact),
oracle numeric vs dense no-pivot LU (the factors are unique).  GPU part: HIP path vs the oracle, through the C ABI."""
import numpy as np
import pytest

from util import sf, gen, nd_perm_py, INT_ARRAYS, INT_SCALARS, rel_err

TOL_FACTOR = 1e-12
TOL_RESIDUAL = 1e-13


def lu_cases():
    c = []
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(6, 6, 1, seed=1)
    c.append(("st2d_6x6_id", n, Cp, Ci, Cx, None, 1 << 30, False))
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(12, 9, 1, seed=2)
    c.append(("st2d_12x9_nd", n, Cp, Ci, Cx, nd_perm_py(12, 9, 1), 1 << 30, False))
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(5, 5, 5, seed=3)
    c.append(("st3d_5_nd", n, Cp, Ci, Cx, nd_perm_py(5, 5, 5), 1 << 30, False))
    c.append(("st3d_5_nd_smallslot", n, Cp, Ci, Cx, nd_perm_py(5, 5, 5), 9000, False))
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(8, 8, 8, seed=4)
    c.append(("st3d_8_nd", n, Cp, Ci, Cx, nd_perm_py(8, 8, 8), 1 << 30, False))
    n, Cp, Ci, Cx = gen.laplacian_lower(7, 7, 7)
    c.append(("sym_lap3d_7_via_lu", n, Cp, Ci, Cx, nd_perm_py(7, 7, 7), 1 << 30, True))
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(40, 1, 1, extra_per_row=3, seed=5)
    c.append(("rand_40", n, Cp, Ci, Cx, None, 1 << 30, False))
    c.append(("one_by_one", 1, np.array([0, 1]), np.array([0]), np.array([3.0]), None, 1 << 30, False))
    return c


def dense_lu_nopiv(A):
    A = A.copy()
    for k in range(len(A)):
        A[k + 1:, k] /= A[k, k]
        A[k + 1:, k + 1:] -= np.outer(A[k + 1:, k], A[k, k + 1:])
    return A


def reference_layout_from_dense(S, LU):
    out = np.zeros(S.xsize)
    for s in range(S.nsuper):
        c0, c1 = S.Super[s], S.Super[s + 1]
        nscol = c1 - c0
        rows = S.Lsi[S.Lsip[s]:S.Lsip[s + 1]]
        nsrow = len(rows)
        lda = 2 * nsrow - nscol
        P = np.zeros((lda, nscol))
        P[:nsrow, :] = LU[np.ix_(rows, range(c0, c1))]
        if nsrow > nscol:
            P[nsrow:, :] = LU[np.ix_(range(c0, c1), rows[nscol:])].T
        out[S.Lsxp[s]:S.Lsxp[s + 1]] = P.T.ravel()
    return out


@pytest.mark.parametrize("case", lu_cases(), ids=lambda c: c[0])
def test_lu_symbolic_bit_exact(oracle, case):
    name, n, Cp, Ci, Cx, perm, slot, symm = case
    P = sf.analyze(n, Cp, Ci, Cx, perm, slot, "lu", symm)
    O = oracle.symbolic.analyze(n, Cp, Ci, Cx, perm, slot, lu=True, symmetric=symm)
    for k in INT_SCALARS + ("unz",):
        assert getattr(P, k) == O[k], k
    for k in INT_ARRAYS + ("Up", "Ui", "UTp", "UTi"):
        assert np.array_equal(getattr(P, k), np.asarray(O[k], dtype=np.int64)), k
    assert np.array_equal(P.Lx, np.asarray(O["Lx"])) and np.array_equal(P.Ux, np.asarray(O["Ux"]))


@pytest.mark.parametrize("blas", ["builtin", "auto"])
@pytest.mark.parametrize("case", lu_cases(), ids=lambda c: c[0])
def test_lu_oracle_against_dense(oracle, case, blas):
    name, n, Cp, Ci, Cx, perm, slot, symm = case
    oracle.blas_init(blas, threads=2)
    S = sf.analyze(n, Cp, Ci, Cx, perm, slot, "lu", symm)
    Lsx, info, st = oracle.lu_factorize(S)
    assert info == 0
    A = gen.dense_from_lower(n, Cp, Ci, Cx) if symm else gen.dense_from_csc(n, Cp, Ci, Cx)
    Ap = A[np.ix_(S.Perm, S.Perm)]
    want = reference_layout_from_dense(S, dense_lu_nopiv(Ap))
    assert rel_err(Lsx, want) <= TOL_FACTOR
    res, x = oracle.lu_residual(S, Lsx)
    assert res <= TOL_RESIDUAL
    b = 1 + np.arange(n) / n
    assert np.allclose(Ap @ x, b, rtol=0, atol=1e-9 * np.abs(b).max())
    assert abs(st["flops_gemm"] + st["flops_getrf"] + st["flops_trsm"] - S.flops_exec) <= 1e-9 * max(S.flops_exec, 1)
    oracle.blas_init("auto", threads=4)


@pytest.mark.gpu
@pytest.mark.parametrize("case", lu_cases(), ids=lambda c: c[0])
def test_lu_gpu_matches_oracle(oracle, case):
    name, n, Cp, Ci, Cx, perm, slot, symm = case
    S = sf.analyze(n, Cp, Ci, Cx, perm, slot, "lu", symm)
    plan = sf.LUPlan(S)
    plan.set_values(S.Lx, None if symm else S.Ux)
    plan.factorize()
    Lsx = plan.get_factor()
    ref, info, _ = oracle.lu_factorize(S)
    assert info == 0
    assert rel_err(Lsx, ref) <= TOL_FACTOR
    res, _ = oracle.lu_residual(S, Lsx)
    assert res <= TOL_RESIDUAL
    # device-side solve (unit-lower forward, U backward) against the reference's host loops (oracle restatement)
    b = 1 + np.arange(n) / n
    x = plan.solve(b)
    want = oracle.lu_solve(S, Lsx, b)
    assert np.allclose(x, want, rtol=1e-12, atol=1e-13 * np.abs(want).max())
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [20, 32])
def test_lu_gpu_medium(oracle, N):
    """supernodes of several hundred columns: blocked panels, outer/inner updates, multi-tile Schur updates"""
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=7)
    S = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N), sf.REFERENCE_SLOT_1GPU, "lu", False)
    plan = sf.LUPlan(S)
    plan.set_values(S.Lx, S.Ux)
    plan.factorize()
    Lsx = plan.get_factor()
    ref, info, _ = oracle.lu_factorize(S)
    assert info == 0
    assert rel_err(Lsx, ref) <= TOL_FACTOR
    res, _ = oracle.lu_residual(S, Lsx)
    assert res <= TOL_RESIDUAL
    assert abs(plan.stat("flops_exec") - S.flops_exec) <= 1e-9 * S.flops_exec
    plan.close()


def _dense_unsym_csc(A):
    n_ = A.shape[0]
    rows, cols = np.nonzero(A.T)          # column-major order of the non-zeros of A: (col, row) pairs of A.T's row-major walk
    cols_, rows_ = rows, cols
    Cp = np.zeros(n_ + 1, dtype=np.int64)
    np.add.at(Cp, cols_ + 1, 1)
    return n_, np.cumsum(Cp), rows_.astype(np.int64), A[rows_, cols_]


def lu_wide_cases():
    """wide supernodes for the fused LU step (several 64-column steps / outer blocks per panel, narrow last blocks)"""
    rng = np.random.default_rng(21)
    out = []

    def dd(n_, band=None):
        A = rng.uniform(-1, 1, (n_, n_))
        if band is not None:
            i, j = np.indices((n_, n_))
            A[np.abs(i - j) > band] = 0.0
        A[np.arange(n_), np.arange(n_)] = np.abs(A).sum(axis=1) + 1.0       # strictly diagonally dominant: no pivoting needed
        return A

    out.append(("dense_unsym_330",) + _dense_unsym_csc(dd(330)))
    out.append(("dense_unsym_577",) + _dense_unsym_csc(dd(577)))
    out.append(("band_unsym_900_130",) + _dense_unsym_csc(dd(900, 130)))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("case", lu_wide_cases(), ids=lambda c: c[0])
def test_lu_gpu_wide_supernodes(oracle, case):
    name, n, Cp, Ci, Cx = case
    S = sf.analyze(n, Cp, Ci, Cx, None, 1 << 30, "lu", False)
    assert np.diff(S.Super).max() > 64
    plan = sf.LUPlan(S)
    plan.set_values(S.Lx, S.Ux)
    plan.factorize()
    Lsx = plan.get_factor()
    ref, info, _ = oracle.lu_factorize(S)
    assert info == 0
    assert rel_err(Lsx, ref) <= TOL_FACTOR
    res, _ = oracle.lu_residual(S, Lsx)
    assert res <= TOL_RESIDUAL
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_lu_distributed_emulated_ranks_on_one_gpu(oracle, world):
    """multi-GPU LU (sf_lu_plan_create_distributed) with every rank's plan on the single test GPU: phase 0, then per
    segment the packed L / U^T blocks are summed through torch tensors aliasing the plans' scratch buffers (the RCCL
    all-reduce between GPUs) and every rank runs the segment with its share of the split GEMM launches"""
    from importlib import import_module
    sharded = import_module("sparse-matrix-factorization-library_amd.sharded")
    N = 20
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=9)
    S = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N), 1 << 30, "lu", False)
    owner, tf, ml = sf.subtree_partition(S, world, 1.0 / world + sharded.TOP_CHAIN_SHARE)
    assert (owner < 0).any()
    engines = [sharded.HipEngine(S, sf.phases_for_rank(owner, r), r == 0, 0, r, world, True) for r in range(world)]
    nseg = engines[0].num_segments()
    assert nseg >= 1
    ref, info, _ = oracle.lu_factorize(S)
    assert info == 0
    for rep in range(2):
        for e in engines:
            e.set_values(S.Lx, S.Ux)
            e.factorize_phase(0)
        for k in range(nseg):
            parts = [e.segment_tensors(k) for e in engines]
            for i in range(len(parts[0])):
                total = parts[0][i].clone()
                for p in parts[1:]:
                    total += p[i]
                for p in parts:
                    p[i].copy_(total)
            for e in engines:
                e.factorize_segment(k)
        for e in engines:
            e.finish()
        full = np.zeros(S.xsize)
        for r, e in enumerate(engines):
            mine = e.get_factor()
            for s_ in np.flatnonzero((owner == r) | ((owner < 0) & (r == 0))):
                full[S.Lsxp[s_]:S.Lsxp[s_ + 1]] = mine[S.Lsxp[s_]:S.Lsxp[s_ + 1]]
        assert rel_err(full, ref) <= TOL_FACTOR
        res, _ = oracle.lu_residual(S, full)
        assert res <= TOL_RESIDUAL
    for e in engines:
        e.close()


@pytest.mark.gpu
def test_lu_zero_pivot_is_reported():
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(5, 5, 1, seed=9)
    Cx = Cx.copy()
    d = [p for j in range(n) for p in range(Cp[j], Cp[j + 1]) if Ci[p] == j]
    Cx[d[0]] = 0.0
    S = sf.analyze(n, Cp, Ci, Cx, None, 1 << 30, "lu", False)
    plan = sf.LUPlan(S)
    plan.set_values(S.Lx, S.Ux)
    plan.set_pivoting(0.0, 0.0)          # the reference's behaviour: no pivoting, no perturbation
    with pytest.raises(sf.SparseFrameError, match="SF_ERR_NOT_POSDEF"):
        plan.factorize()
    plan.close()


def test_lu_struct_api_host_side(oracle):
    """LU library, reference call order (LU/Source/SparseFrame.c:3997-4024) with an oracle-made factor"""
    import ctypes as C
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(6, 6, 6, seed=11)
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    mi = sf.LUMatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx, symmetric=False)
    mi.set_perm(nd_perm_py(6, 6, 6))
    mi.analyze(common)
    S = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(6, 6, 6), 1 << 30, "lu", False)
    assert (mi.c.nsuper, mi.c.xsize, mi.c.csize, mi.c.isize) == (S.nsuper, S.xsize, S.csize, S.isize)
    for k, ln in (("Ui", S.unz), ("Up", n + 1), ("Li", S.nnz), ("Lsi", S.isize), ("Lsxp", S.nsuper + 1), ("Perm", n)):
        assert np.array_equal(mi.array(k, ln), getattr(S, k)), k
    Lsx, info, _ = oracle.lu_factorize(S)
    assert info == 0
    C.memmove(mi.c.Lsx, Lsx.ctypes.data, Lsx.nbytes)
    res = mi.validate()
    want, x = oracle.lu_residual(S, Lsx)
    assert res <= TOL_RESIDUAL and abs(res - want) <= 1e-16
    assert np.allclose(mi.array("Xx", n), x, rtol=1e-14, atol=0)
    mi.cleanup()
    assert not mi.c.Lsx and not mi.c.Up


@pytest.mark.gpu
def test_lu_struct_entry_points_end_to_end(oracle, tmp_path):
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(10, 10, 10, seed=12)
    path = tmp_path / "u.mtx"
    gen.write_matrix_market(path, n, Cp, Ci, Cx, symmetric=False)
    common = sf.CommonInfo()
    common.c.devSlotSize = 1 << 30
    mi = sf.LUMatrixInfo()
    mi.read(path)
    assert (mi.c.isSymmetric, mi.c.nzmax) == (0, len(Ci))
    mi.set_perm(nd_perm_py(10, 10, 10))
    mi.analyze(common)
    mi.factorize(common)
    res = mi.validate()
    assert res <= TOL_RESIDUAL
    S = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(10, 10, 10), 1 << 30, "lu", False)
    ref, info, _ = oracle.lu_factorize(S)
    assert rel_err(mi.array("Lsx", S.xsize).copy(), ref) <= TOL_FACTOR
    mi.cleanup()
    common.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["direct", "direct-small-slots", "pack"])
def test_lu_struct_copy_back_forms(oracle, monkeypatch, form):
    """The LU factor's way back to the host, compared value by value with the oracle's packed panels (L:2514-2517):
    direct            -- U11 filled into the L panel's upper triangle on the compute stream (k_lu_fill_u11), the L and U^T runs
                         copied as they are, the copy workers interleave the columns (default);
    direct-small-slots - the same with 1 MiB staging slots, so that this small matrix also has supernodes cut into column
                         ranges (L run + 2-D copy of the U^T rows below the diagonal block) as the big ones of 79^3 / 110^3 are;
    pack              -- the older gather kernel per piece (SF_DL_LU_PACK=1)."""
    if form == "pack":
        if not sf.lib.sf_build_experiments():
            pytest.skip("the gather-kernel copy-back is an A/B switch compiled out of release builds (make EXP=1)")
        monkeypatch.setenv("SF_DL_LU_PACK", "1")
    if form == "direct-small-slots":
        monkeypatch.setenv("SF_DL_SLOT_MB", "1")
    N = 24
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=21)
    perm = nd_perm_py(N, N, N)
    common = sf.CommonInfo(dev_slot_size=4 << 30)
    S = sf.analyze(n, Cp, Ci, Cx, perm, 4 << 30, "lu", False)
    assert np.diff(S.Super).max() > 512            # a supernode of two outer blocks
    ref, info, _ = oracle.lu_factorize(S)
    assert info == 0
    for scale in (1.0, 3.0):                         # second call: cached plan, the L panels' scratch triangles hold the last U11
        mi = sf.LUMatrixInfo()
        mi.set_csc(n, Cp, Ci, Cx * scale, symmetric=False)
        mi.set_perm(perm)
        mi.analyze(common)
        import ctypes as C
        C.memset(mi.c.Lsx, 0xff, 8 * S.xsize)       # NaNs: every value must be written by the copy-back
        mi.factorize(common)
        got = mi.array("Lsx", S.xsize).copy()
        assert not np.isnan(got).any()
        # L is scale-invariant (unit lower), U scales: compare through the oracle on the scaled matrix
        S2 = sf.analyze(n, Cp, Ci, Cx * scale, perm, 4 << 30, "lu", False)
        ref2, info2, _ = oracle.lu_factorize(S2)
        assert rel_err(got, ref2) <= TOL_FACTOR
        assert mi.validate() <= TOL_RESIDUAL
        mi.cleanup()
    common.close()

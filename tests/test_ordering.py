"""Built-in ordering (sf_graph_nd_perm): a self-contained stand-in for the reference's METIS call.  Orderings are
outside the parity contract; the tests check validity, usefulness (less fill than the identity) and that the whole
pipeline works with it (oracle factorization of the reordered matrix solves the system)."""
import numpy as np
import pytest

from util import sf, gen


@pytest.mark.parametrize("dims", [(30, 30, 1), (12, 12, 12), (50, 3, 1), (1, 1, 1)])
def test_graph_nd_is_a_permutation_and_reduces_fill(dims):
    n, Cp, Ci, Cx = gen.laplacian_lower(*dims)
    p = sf.graph_nd_perm(n, Cp, Ci)
    assert sorted(p.tolist()) == list(range(n))
    if n > 100:
        ident = sf.analyze(n, Cp, Ci, Cx, None, 8 << 30)
        nd = sf.analyze(n, Cp, Ci, Cx, p, 8 << 30)
        assert nd.flops_struct < ident.flops_struct


def test_graph_nd_disconnected_and_dense_and_diagonal():
    # many components (diagonal), a dense block, block-diagonal mix
    n = 5000
    p = sf.graph_nd_perm(n, np.arange(n + 1), np.arange(n))
    assert sorted(p.tolist()) == list(range(n))
    nd_ = 40
    rows, cols = np.tril_indices(nd_)
    order = np.lexsort((rows, cols))
    Cp = np.zeros(nd_ + 1, dtype=np.int64)
    np.add.at(Cp, cols + 1, 1)
    p = sf.graph_nd_perm(nd_, np.cumsum(Cp), rows[order])
    assert sorted(p.tolist()) == list(range(nd_))
    n, Cp, Ci, Cx = gen.random_spd_lower(3000, 3, seed=5)
    p = sf.graph_nd_perm(n, Cp, Ci, leaf=16)
    assert sorted(p.tolist()) == list(range(n))


def test_analyze_uses_builtin_ordering_when_none_is_supplied(oracle):
    n, Cp, Ci, Cx = gen.laplacian_lower(14, 14, 14)
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    mi = sf.MatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx)
    mi.use_builtin_ordering()
    mi.analyze(common)
    perm = mi.array("Perm", n).copy()
    assert sorted(perm.tolist()) == list(range(n))
    ident = sf.analyze(n, Cp, Ci, Cx, None, 1 << 30)
    assert mi.c.xsize < ident.xsize
    # the struct arrays are those of the flat analysis with the same ordering
    S = sf.analyze(n, Cp, Ci, Cx, sf.graph_nd_perm(n, Cp, Ci), 1 << 30)
    assert np.array_equal(perm, S.Perm) and mi.c.nsuper == S.nsuper
    Lsx, info, _ = oracle.chol_factorize(S)
    res, _ = oracle.chol_residual(S, Lsx)
    assert info == 0 and res <= 1e-13
    mi.cleanup()

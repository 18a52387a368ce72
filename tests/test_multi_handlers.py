"""Multi-GPU through the reference's own C entry point: SparseFrame_factorize_supernodal uses EVERY handler of
gpu_info_list for one matrix (reference C:2267): elimination-tree subtrees per handler, the parent-front merge as a
collective issued by the C library on the plans' streams (sf_multi.hip).

On a one-GPU box SF_EMULATE_HANDLERS=N makes SparseFrame_allocate_gpu return N handlers that share device 0; the
orchestration, the per-rank distributed plans, the segment loop and the dealt-out overlapped copy-back are the real
ones, only the all-reduce is a kernel on that device instead of RCCL.  With >= 2 devices the same test body runs over
RCCL.  The RCCL binding itself (dlopen, unique id, communicator, ncclAllReduce on a caller's stream) is exercised
with a one-rank communicator."""
import os
import ctypes as C

import numpy as np
import pytest

from util import sf, gen, nd_perm_py, rel_err

pytestmark = pytest.mark.gpu
TOL_FACTOR = 1e-12
TOL_RESIDUAL = 1e-13


def _chol_case(oracle, N, nhandlers_expected):
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    perm = sf.grid_nd_perm(N, N, N)
    common = sf.CommonInfo(dev_slot_size=8 << 30)
    assert common.c.numGPU == nhandlers_expected
    sym = sf.analyze(n, Cp, Ci, Cx, perm, 8 << 30)
    ref, info, _ = oracle.chol_factorize(sym)
    mask = oracle.lower_mask(sym)
    for scale in (1.0, 2.0):            # second call: cached per-rank plans and communicators
        mi = sf.MatrixInfo()
        mi.set_csc(n, Cp, Ci, Cx * scale)
        mi.set_perm(perm)
        mi.analyze(common)
        C.memset(mi.c.Lsx, 0xff, 8 * sym.xsize)          # NaNs: every entry must be written by some rank's copy-back
        mi.factorize(common)
        assert sf._lib.lib.sf_handlers_replica_mismatches(mi.c.Lsx) == 0          # shared panels: bit-identical on the ranks of a group
        got = mi.array("Lsx", sym.xsize).copy()
        assert not np.isnan(got[mask]).any()
        assert rel_err(got, ref * np.sqrt(scale), mask) <= TOL_FACTOR
        assert mi.validate() <= TOL_RESIDUAL
        mi.cleanup()
    common.close()


@pytest.mark.parametrize("nh", [2, 3, 8])
def test_all_handlers_factorize_one_matrix_emulated(oracle, monkeypatch, nh):
    if sf.device_count() != 1:
        pytest.skip("emulated handlers are for one-GPU boxes")
    monkeypatch.setenv("SF_EMULATE_HANDLERS", str(nh))
    _chol_case(oracle, 24, nh)


@pytest.mark.parametrize("nh,lookahead", [(2, "1"), (4, "1"), (2, "0")])
def test_lookahead_schedule_three_outer_blocks(oracle, monkeypatch, nh, lookahead):
    """34^3: the root separator has 1156 columns = 3 outer blocks, so the look-ahead schedule of a shared panel has all its
    parts: the far part of block 2's update issued after the chain of block 0 (split over the ranks), the parts of the block
    just before (replicated, after the reduce point), sums issued one segment ahead on the second stream.  SF_LOOKAHEAD=0 is
    the older schedule (whole update split, sum in line)."""
    if sf.device_count() != 1:
        pytest.skip("emulated handlers are for one-GPU boxes")
    monkeypatch.setenv("SF_EMULATE_HANDLERS", str(nh))
    monkeypatch.setenv("SF_LOOKAHEAD", lookahead)
    _chol_case(oracle, 34, nh)


@pytest.mark.parametrize("nh,N", [(2, 34), (3, 34), (4, 40), (8, 40)])
def test_owner_computes_prototype_matches_the_oracle(oracle, monkeypatch, nh, N):
    """SF_TOP_OWNER=1 (SURVEY 8f rank 4 prototype): in the root separator's set the near GEMM and the 64-column
    chain of a 512-column block run on ONE rank (block number mod group size), the finished block column is broadcast (a sum whose
    other terms are zero) before the split far GEMMs that read it.  Same factor as the replicated schedule to rounding, replicas
    bit-identical, every entry of Lsx written, cached plans on the second call.  34^3: 3 outer blocks in the root, 40^3: 4."""
    if sf.device_count() != 1:
        pytest.skip("emulated handlers are for one-GPU boxes")
    monkeypatch.setenv("SF_EMULATE_HANDLERS", str(nh))
    monkeypatch.setenv("SF_TOP_OWNER", "1")
    _chol_case(oracle, N, nh)


@pytest.mark.parametrize("nh,N", [(2, 64), (4, 64), (8, 96)])
def test_emulated_handlers_large_by_residual(monkeypatch, nh, N):
    """64^3 over 2 / 4 emulated handlers: the root separator has 8 outer blocks, its children 4 (groups of 2 at nh = 4), and the
    split Schur launches of the top levels have thousands of tiles per rank window, i.e. the dynamic whole-tile rounds of k_gemm run
    inside a rank's unit window.  Too big for the oracle in test time: checked by the residual of the reference's validate() on the
    factor the ranks copied back (every entry of Lsx written)."""
    if sf.device_count() != 1:
        pytest.skip("emulated handlers are for one-GPU boxes")
    monkeypatch.setenv("SF_EMULATE_HANDLERS", str(nh))
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    common = sf.CommonInfo(dev_slot_size=8 << 30)
    assert common.c.numGPU == nh
    mi = sf.MatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx)
    mi.set_perm(sf.grid_nd_perm(N, N, N))
    mi.analyze(common)
    xsize = int(mi.c.xsize)
    C.memset(mi.c.Lsx, 0xff, 8 * xsize)
    mi.factorize(common)
    assert mi.validate() <= TOL_RESIDUAL
    mi.cleanup()
    common.close()


@pytest.mark.parametrize("method", ["cholesky", "lu"])
def test_struct_solve_after_multi_handler_factorization(monkeypatch, method):
    """after a factorization by several handlers the factor is spread over their plans; with SF_SOLVE=gather
    SparseFrame_solve_supernodal gathers it (device to device) into a whole plan on the first handler's device and solves there,
    again after a refactorization; same solution as the reference's host solve over Lsx.  (Default: the distributed solve, next
    test.)"""
    monkeypatch.setenv("SF_SOLVE", "gather")
    if sf.device_count() != 1:
        pytest.skip("emulated handlers are for one-GPU boxes")
    from importlib import import_module
    lib = import_module("sparse-matrix-factorization-library_amd._lib").lib
    monkeypatch.setenv("SF_EMULATE_HANDLERS", "3")
    if method == "lu":
        N = 14
        n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=5)
        mi = sf.LUMatrixInfo()
        sym_flag = dict(symmetric=False)
    else:
        N = 24
        n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
        mi = sf.MatrixInfo()
        sym_flag = {}
    common = sf.CommonInfo(dev_slot_size=8 << 30)
    assert common.c.numGPU == 3
    for scale in (1.0, 3.0):
        mi.set_csc(n, Cp, Ci, Cx * scale, **sym_flag)
        mi.set_perm(nd_perm_py(N, N, N))
        mi.analyze(common)
        mi.factorize(common)
        k = lib.sf_handlers_resident_solves()
        assert mi.validate() <= TOL_RESIDUAL
        assert lib.sf_handlers_resident_solves() == k + 1
        x_dev = mi.array("Xx", n).copy()
        monkeypatch.setenv("SF_SOLVE", "host")
        assert mi.validate() <= TOL_RESIDUAL
        assert lib.sf_handlers_resident_solves() == k + 1
        x_host = mi.array("Xx", n).copy()
        monkeypatch.setenv("SF_SOLVE", "gather")
        assert np.max(np.abs(x_dev - x_host)) <= 1e-12 * np.max(np.abs(x_host))
        mi.cleanup()
    common.close()


@pytest.mark.parametrize("method,nh,N", [("cholesky", 2, 24), ("cholesky", 3, 34), ("cholesky", 8, 40), ("lu", 2, 14), ("lu", 4, 24)])
def test_distributed_solve_with_the_factor_left_on_the_ranks(monkeypatch, method, nh, N):
    """after a multi-handler factorization the ranks solve together with the panels they hold -- one small
    sum per shared supernode in the forward sweep, none in the backward one -- instead of gathering the factor on one device;
    same solution as the reference's host solve over Lsx.  34^3 / 40^3: shared supernodes of several 256-column steps, groups of
    2, 4 and 8"""
    if sf.device_count() != 1:
        pytest.skip("emulated handlers are for one-GPU boxes")
    from importlib import import_module
    lib = import_module("sparse-matrix-factorization-library_amd._lib").lib
    monkeypatch.setenv("SF_EMULATE_HANDLERS", str(nh))
    if method == "lu":
        n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=11)
        mi = sf.LUMatrixInfo()
        kw = dict(symmetric=False)
    else:
        n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
        mi = sf.MatrixInfo()
        kw = {}
    common = sf.CommonInfo(dev_slot_size=8 << 30)
    assert common.c.numGPU == nh
    mi.set_csc(n, Cp, Ci, Cx, **kw)
    mi.set_perm(nd_perm_py(N, N, N))
    mi.analyze(common)
    mi.factorize(common)
    k = lib.sf_handlers_resident_solves()                  # default for a multi-handler factor: the distributed solve
    assert mi.validate() <= TOL_RESIDUAL
    assert lib.sf_handlers_resident_solves() == k + 1
    x_dist = mi.array("Xx", n).copy()
    monkeypatch.setenv("SF_SOLVE", "host")
    assert mi.validate() <= TOL_RESIDUAL
    x_host = mi.array("Xx", n).copy()
    assert np.max(np.abs(x_dist - x_host)) <= 1e-12 * np.max(np.abs(x_host))
    # one changed value in the caller's copy (the root panel's last entry): every panel's fingerprint is compared with every rank
    # that stores it, so the ranks are NOT asked and the host sweep answers for the data the caller holds
    monkeypatch.delenv("SF_SOLVE")
    Lsx = mi.array("Lsx", int(mi.c.xsize))
    Lsx[-1] *= 1.0 + 1e-9
    k = lib.sf_handlers_resident_solves()
    mi.validate()
    assert lib.sf_handlers_resident_solves() == k
    mi.cleanup()
    common.close()


def test_lookahead_schedule_lu(oracle, monkeypatch):
    """LU with the look-ahead schedule (the default): the ranks of a group must hold bit-identical copies of a shared panel --
    threshold pivot decisions may not depend on a rank's own rounding -- so the replicated near parts run k_gemm without
    K-splitting (whole_tiles).  Checked with the library's replica comparison, then the factor against the oracle"""
    if sf.device_count() != 1:
        pytest.skip("emulated handlers are for one-GPU boxes")
    monkeypatch.setenv("SF_EMULATE_HANDLERS", "2")
    monkeypatch.delenv("SF_LOOKAHEAD", raising=False)
    N = 33
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=13)
    perm = nd_perm_py(N, N, N)
    common = sf.CommonInfo(dev_slot_size=8 << 30)
    mi = sf.LUMatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx, symmetric=False)
    mi.set_perm(perm)
    mi.analyze(common)
    mi.factorize(common)
    assert sf._lib.lib.sf_handlers_replica_mismatches(mi.c.Lsx) == 0
    assert mi.validate() <= TOL_RESIDUAL
    S = sf.analyze(n, Cp, Ci, Cx, perm, 8 << 30, "lu", False)
    assert np.diff(S.Super).max() > 1024
    ref, info, _ = oracle.lu_factorize(S)
    assert rel_err(mi.array("Lsx", S.xsize).copy(), ref) <= TOL_FACTOR
    mi.cleanup()
    common.close()


def test_all_handlers_lu_emulated(oracle, monkeypatch):
    if sf.device_count() != 1:
        pytest.skip("emulated handlers are for one-GPU boxes")
    monkeypatch.setenv("SF_EMULATE_HANDLERS", "2")
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(12, 12, 12, seed=3)
    perm = nd_perm_py(12, 12, 12)
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    assert common.c.numGPU == 2
    mi = sf.LUMatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx, symmetric=False)
    mi.set_perm(perm)
    mi.analyze(common)
    mi.factorize(common)
    assert mi.validate() <= TOL_RESIDUAL
    S = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", False)
    ref, info, _ = oracle.lu_factorize(S)
    assert rel_err(mi.array("Lsx", S.xsize).copy(), ref) <= TOL_FACTOR
    mi.cleanup()
    common.close()


def test_one_matrix_per_handler_mode(oracle, monkeypatch):
    """SF_MULTI=matrix: the matrices of the caller's matrix threads are spread over the handlers instead"""
    if sf.device_count() != 1:
        pytest.skip("emulated handlers are for one-GPU boxes")
    monkeypatch.setenv("SF_EMULATE_HANDLERS", "2")
    monkeypatch.setenv("SF_MULTI", "matrix")
    n, Cp, Ci, Cx = gen.laplacian_lower(12, 12, 12)
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    for serial in (0, 1):
        mi = sf.MatrixInfo(serial=serial)
        mi.set_csc(n, Cp, Ci, Cx)
        mi.analyze(common)
        mi.factorize(common)
        assert mi.validate() <= TOL_RESIDUAL
        mi.cleanup()
    common.close()


def test_two_device_handlers_over_rccl(oracle):
    """the real thing: needs >= 2 MI355X in this process (skips on the one-GPU test boxes)"""
    if sf.device_count() < 2:
        pytest.skip("needs two devices")
    _chol_case(oracle, 24, sf.device_count())


def test_rccl_binding_one_rank_communicator():
    """librccl.so.1 through the library's own binding: unique id, ncclCommInitRank, ncclAllReduce(sum, fp64) on the caller's
    stream, destroy.  One rank: the sum is the input."""
    import torch
    uid = sf.Comm.unique_id()
    assert len(uid) == 128
    comm = sf.Comm(0, 0, 1, uid)
    t = torch.arange(1 << 20, dtype=torch.float64, device="cuda:0")
    want = t.clone()
    s = torch.cuda.Stream(device=0)
    s.wait_stream(torch.cuda.current_stream(0))
    comm.allreduce_sum(t.data_ptr(), t.numel(), s.cuda_stream)
    s.synchronize()
    assert torch.equal(t, want)
    # the sub-communicator calls of the proportional mapping (ncclCommSplit + a collective on the child)
    from util import sf as _sf
    lib = __import__("importlib").import_module("sparse-matrix-factorization-library_amd._lib").lib
    assert lib.sf_comm_selftest_split(comm._h, C.c_void_p(t.data_ptr()), t.numel(), C.c_void_p(s.cuda_stream)) == 0
    assert torch.equal(t, want)
    comm.close()


def test_mapped_plans_store_less_and_group_their_segments():
    """proportional mapping (sf_chol_plan_create_mapped): a rank holds its subtrees and the top supernodes above them, not
    the whole top; its segments name the group of ranks that sums them, the root's group is everybody"""
    N, W = 24, 8
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N), 8 << 30)
    owner, _, _ = sf.subtree_partition(sym, W, 1.0 / W + 0.25)
    assert np.count_nonzero(owner < 0) >= 3
    stored_mapped, stored_repl, groups = [], [], set()
    for r in range(W):
        pm = sf.CholPlan(sym, owner=owner, rank=r, nranks=W)
        pr = sf.CholPlan(sym, phase=sf.phases_for_rank(owner, r), load_top=(r == 0), rank=r, nranks=W)
        stored_mapped.append(pm.stat("stored_doubles"))
        stored_repl.append(pr.stat("stored_doubles"))
        gs = [pm.segment_group(k) for k in range(pm.num_segments())]
        assert all((g >> r) & 1 for g in gs)            # a rank only meets segments of groups it belongs to
        assert gs[-1] == (1 << W) - 1                   # the root is shared by everybody
        groups.update(gs)
        pm.close()
        pr.close()
    assert len(groups) >= 3                             # nested groups: pairs, quads, all
    assert max(stored_mapped) < 0.9 * max(stored_repl)


def test_one_handler_list_two_patterns_emulated(oracle, monkeypatch):
    """two matrices with different elimination trees through ONE handler list: the second one's groups of ranks are not the
    first one's, their sub-communicators are made when it arrives; both plans stay cached"""
    if sf.device_count() != 1:
        pytest.skip("emulated handlers are for one-GPU boxes")
    monkeypatch.setenv("SF_EMULATE_HANDLERS", "4")
    common = sf.CommonInfo(dev_slot_size=8 << 30)
    for dims in ((20, 20, 20), (40, 30, 6), (20, 20, 20)):
        n, Cp, Ci, Cx = gen.laplacian_lower(*dims)
        perm = sf.grid_nd_perm(*dims)
        sym = sf.analyze(n, Cp, Ci, Cx, perm, 8 << 30)
        ref, info, _ = oracle.chol_factorize(sym)
        mi = sf.MatrixInfo()
        mi.set_csc(n, Cp, Ci, Cx)
        mi.set_perm(perm)
        mi.analyze(common)
        mi.factorize(common)
        assert rel_err(mi.array("Lsx", sym.xsize).copy(), ref, oracle.lower_mask(sym)) <= TOL_FACTOR
        mi.cleanup()
    common.close()


@pytest.mark.parametrize("where", [1, 2, 3, 4])
@pytest.mark.parametrize("method", ["cholesky", "lu"])
def test_a_failing_rank_does_not_leave_its_peers_waiting(monkeypatch, method, where):
    """a rank that fails before its first collective (where = 1 factorization, 3 solve) or in the middle of a run (2, 4)
    used to return alone -- its peers then waited for ever (in LocalGroup::barrier here, in an RCCL kernel on real ranks).  Now the
    ranks agree on their state before the first data collective and keep the hand-shake going after a mid-run failure: every
    handler thread returns, the call reports an error, and the NEXT factorization / solve on the same handlers works."""
    if sf.device_count() != 1:
        pytest.skip("emulated handlers are for one-GPU boxes")
    import threading
    lib = sf._lib.lib
    monkeypatch.setenv("SF_EMULATE_HANDLERS", "3")
    N = 20
    if method == "lu":
        n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=3)
        mi = sf.LUMatrixInfo()
        kw = dict(symmetric=False)
    else:
        n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
        mi = sf.MatrixInfo()
        kw = {}
    common = sf.CommonInfo(dev_slot_size=8 << 30)
    mi.set_csc(n, Cp, Ci, Cx, **kw)
    mi.set_perm(nd_perm_py(N, N, N))
    mi.analyze(common)
    mi.factorize(common)                                   # builds plans and communicators
    assert mi.validate() <= TOL_RESIDUAL
    result = {}

    def run():
        try:
            if where <= 2:
                mi.factorize(common)
                result["rc"] = "ok"
            else:
                result["res"] = mi.validate()          # solve: a failed distributed solve falls back to the host sweep
                result["rc"] = "ok"
        except sf.SparseFrameError as e:
            result["rc"] = str(e)

    lib.sf_test_inject_failure(1, where)
    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(120)
    assert not t.is_alive(), "a handler thread is still waiting for the failed rank"
    if where <= 2:
        assert "SF_ERR" in result["rc"], result
    else:
        assert result["rc"] == "ok" and result["res"] <= TOL_RESIDUAL       # answered by the host solve
    lib.sf_test_inject_failure(-1, 0)
    mi.factorize(common)                                   # the handlers are still usable
    assert mi.validate() <= TOL_RESIDUAL
    mi.cleanup()
    common.close()


def test_three_launch_chain_steps_keep_replicas_identical(oracle, monkeypatch):
    """with SF_FUSE_MAX=0 the 64-column steps of a SHARED panel run as three launches whose K = 64 t GEMM used to split
    tiles by K (several atomic additions per element, in a rank-dependent order): replicas could differ in the last bits, and an LU
    threshold-pivot decision with them.  The launch now runs whole tiles for shared sets."""
    if sf.device_count() != 1:
        pytest.skip("emulated handlers are for one-GPU boxes")
    monkeypatch.setenv("SF_EMULATE_HANDLERS", "2")
    monkeypatch.setenv("SF_FUSE_MAX", "0")
    N = 24
    n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=13)
    perm = nd_perm_py(N, N, N)
    S = sf.analyze(n, Cp, Ci, Cx, perm, 8 << 30, "lu", False)
    ref, info, _ = oracle.lu_factorize(S)
    common = sf.CommonInfo(dev_slot_size=8 << 30)
    mi = sf.LUMatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx, symmetric=False)
    mi.set_perm(perm)
    mi.analyze(common)
    mi.factorize(common)
    assert sf._lib.lib.sf_handlers_replica_mismatches(mi.c.Lsx) == 0
    assert rel_err(mi.array("Lsx", S.xsize).copy(), ref) <= TOL_FACTOR
    assert mi.validate() <= TOL_RESIDUAL
    mi.cleanup()
    common.close()

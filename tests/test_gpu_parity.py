"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the same
inputs (fp64 tolerance of SURVEY 8(c): max |Lsx - ref| / max |ref| <= 1e-12 on the entries the
reference defines, residual <= 1e-13), against the committed golden fixtures, and -- at sizes the
dense check cannot reach -- through size-independent properties."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from util import sf, gen, nd_perm_py, small_cases, wide_cases, rel_err

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOL_FACTOR = 1e-12
TOL_RESIDUAL = 1e-13


EXPERIMENT_KNOBS = ("SF_DL_2D", "SF_DL_LU_PACK", "SF_GEMM_DMA", "SF_GEMM_DYNAMIC", "SF_SOLVE_BWD_AHEAD", "SF_SOLVE_BWD_FUSED",
                    "SF_SOLVE_DIAGT", "SF_SOLVE_FAR_GROUPS", "SF_SOLVE_FAR_WGS", "SF_SOLVE_FWD_AHEAD", "SF_SOLVE_FWD_FAR_FIRST",
                    "SF_SU_MAXK", "SF_WEIGHTED_SHARES", "SF_DL_HOST_WAIT", "SF_DL_PIN", "SF_STREAM_PRIORITY", "SF_LOOKAHEAD1", "SF_LOOKAHEAD1_GRID")


def needs_experiments(knobs):
    """the A/B switches of finished experiments are compiled out of release builds (make EXP=1 keeps them): a case that sets one
    is skipped unless the library has them -- setting it would silently test the default path a second time"""
    if any(k in EXPERIMENT_KNOBS for k in knobs) and not sf.lib.sf_build_experiments():
        pytest.skip("A/B switch compiled out of this build (make -C sparse-matrix-factorization-library_amd/csrc EXP=1)")


def gpu_factor(sym):
    plan = sf.CholPlan(sym, device=0)
    plan.set_values(sym.Lx)
    plan.factorize()
    Lsx = plan.get_factor()
    return plan, Lsx


@pytest.mark.parametrize("case", small_cases(), ids=lambda c: c[0])
def test_factor_matches_oracle(oracle, case):
    name, n, Cp, Ci, Cx, perm, slot = case
    sym = sf.analyze(n, Cp, Ci, Cx, perm, slot)
    plan, Lsx = gpu_factor(sym)
    ref, info, _ = oracle.chol_factorize(sym)
    assert info == 0
    mask = oracle.lower_mask(sym)
    assert rel_err(Lsx, ref, mask) <= TOL_FACTOR
    res, _ = oracle.chol_residual(sym, Lsx)
    assert res <= TOL_RESIDUAL
    plan.close()


@pytest.mark.parametrize("case", wide_cases(), ids=lambda c: c[0])
def test_wide_supernodes_match_oracle(oracle, case):
    """multi-step panels through the fused 64-column step kernel (k_step) and the blocked outer GEMMs"""
    name, n, Cp, Ci, Cx, perm, slot = case
    sym = sf.analyze(n, Cp, Ci, Cx, perm, slot)
    assert np.diff(sym.Super).max() > 64
    plan, Lsx = gpu_factor(sym)
    assert plan.stat("last_step_ms") >= 0                 # stat exists; the fused path is part of this plan
    ref, info, _ = oracle.chol_factorize(sym)
    assert info == 0
    assert rel_err(Lsx, ref, oracle.lower_mask(sym)) <= TOL_FACTOR
    res, _ = oracle.chol_residual(sym, Lsx)
    assert res <= TOL_RESIDUAL
    # device solve on the same factor
    b = 1.0 + np.arange(n) / n
    x = plan.solve(b)
    assert sf.validate_solution(sym, x, b) <= 1e-12
    plan.close()


def test_golden_fixtures():
    """fixtures made by tests/golden/make_golden.py (oracle with its built-in C loops, checked there
    against dense LAPACK); no oracle code runs here"""
    with open(os.path.join(HERE, "golden", "chol_small.json")) as f:
        G = json.load(f)
    for name, g in G.items():
        sym = sf.analyze(g["n"], g["Cp"], g["Ci"], g["Cx"], g["perm"], g["devSlotSize"])
        for k in ("Super", "Lsip", "Lsxp", "Lsi", "Perm", "LeafQueue"):
            assert np.array_equal(getattr(sym, k), np.asarray(g[k], dtype=np.int64)), (name, k)
        plan, Lsx = gpu_factor(sym)
        want = np.asarray(g["Lsx"])
        mask = np.asarray(g["mask"], dtype=bool)
        assert rel_err(Lsx, want, mask) <= TOL_FACTOR, name
        plan.close()


@pytest.mark.parametrize("name", ["chol_lap3d_24", "chol_stencil2d_200"])
def test_golden_fixtures_large(name, monkeypatch):
    """tests/golden/large_sampled.json (make_golden_large.py: oracle output accepted there only after agreeing with dense LAPACK /
    SuperLU): an 895-column root panel (two outer blocks: k_gemm<0>, fused k_step launches with pushes into future diagonal
    blocks, K > 64 Schur updates) and a config-3-like 2-D stencil (swarm levels, k_update_small).  SHA-256 of inputs and of every
    integer output, sampled entries of the factor at 1e-12, sum log|diag| and sum |entries|.  No oracle code runs here."""
    import golden_large as GL
    g = GL.load()[name]
    c = GL.build_case(name)
    GL.check_inputs_and_symbolic(name, g, c)
    plan, Lsx = gpu_factor(c["sym"])
    GL.check_factor(name, g, c["sym"], Lsx, TOL_FACTOR)
    plan.close()


def test_struct_entry_points_end_to_end(oracle, tmp_path):
    """the reference driver's call order (SparseFrame.c:3396-3423) through the struct ABI:
    BASELINE config 1 = 2-D 5-pt Laplacian 100x100 from a MatrixMarket file"""
    n, Cp, Ci, Cx = gen.laplacian_lower(100, 100)
    path = tmp_path / "lap100.mtx"
    gen.write_matrix_market(path, n, Cp, Ci, Cx)
    common = sf.CommonInfo()
    assert common.c.numGPU >= 1 and common.c.devSlotSize > (1 << 30)
    common.c.devSlotSize = 1 << 30
    mi = sf.MatrixInfo()
    mi.read(path)
    mi.set_perm(None)            # config 1 is quoted in natural order (explicit opt-in; the default orders)
    mi.analyze(common)
    assert (mi.c.nsuper, mi.c.nstage) == (155, 1)
    mi.factorize(common)
    res = mi.validate()
    assert res <= TOL_RESIDUAL
    sym = sf.analyze(n, Cp, Ci, Cx, None, 1 << 30)
    ref, info, _ = oracle.chol_factorize(sym)
    Lsx = mi.array("Lsx", sym.xsize).copy()
    assert rel_err(Lsx, ref, oracle.lower_mask(sym)) <= TOL_FACTOR
    assert mi.c.factorizeTime > 0
    mi.cleanup()
    common.close()


@pytest.mark.parametrize("knobs", [
    {"SF_DL_HOST_WAIT": "0"}, {"SF_DL_WORKERS": "1"}, {"SF_DL_WORKERS": "8", "SF_DL_SLOT_MB": "1"}, {"SF_DL_2D": "0"},
    {"SF_STREAM_PRIORITY": "0"}, {"SF_DL_PIN": "1"}], ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()))
def test_struct_copy_back_knobs(oracle, monkeypatch, knobs):
    """every alternative of the overlapped copy-back (device-side event waits, 1 / 8 workers, 1 MiB slots so that this matrix has
    many pieces and 2-D pieces, 1-D copies only, ordinary streams, NUMA-confined workers): SparseFrame_factorize must leave the
    same Lsx -- every stored value, including the zero-filled rows above the block columns -- twice in a row"""
    needs_experiments(knobs)
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    N = 26
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    perm = sf.grid_nd_perm(N, N, N)
    sym = sf.analyze(n, Cp, Ci, Cx, perm, 4 << 30)
    assert np.diff(sym.Super).max() > 512                # a panel of two outer blocks: 2-D pieces exist
    ref, info, _ = oracle.chol_factorize(sym)
    mask = oracle.lower_mask(sym)
    common = sf.CommonInfo(dev_slot_size=4 << 30)
    for scale in (1.0, 4.0):
        mi = sf.MatrixInfo()
        mi.set_csc(n, Cp, Ci, Cx * scale)
        mi.set_perm(perm)
        mi.analyze(common)
        C.memset(mi.c.Lsx, 0xff, 8 * sym.xsize)
        mi.factorize(common)
        got = mi.array("Lsx", sym.xsize).copy()
        assert not np.isnan(got[mask]).any()
        assert rel_err(got, ref * np.sqrt(scale), mask) <= TOL_FACTOR
        assert mi.validate() <= TOL_RESIDUAL
        mi.cleanup()
    common.close()


def test_not_positive_definite_is_reported():
    n, Cp, Ci, Cx = gen.laplacian_lower(6, 6)
    Cx = Cx.copy()
    Cx[Cp[20]] = -1.0
    sym = sf.analyze(n, Cp, Ci, Cx, None, 1 << 30)
    plan = sf.CholPlan(sym)
    plan.set_values(sym.Lx)
    with pytest.raises(sf.SparseFrameError, match="SF_ERR_NOT_POSDEF"):
        plan.factorize()
    plan.close()


def test_refactorize_same_pattern_new_values(oracle):
    """a plan is reusable: new values, same structure; and repeated runs agree to rounding"""
    n, Cp, Ci, Cx = gen.laplacian_lower(10, 10, 10)
    sym = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(10, 10, 10), 1 << 30)
    plan = sf.CholPlan(sym)
    mask = oracle.lower_mask(sym)
    for scale in (1.0, 3.5):
        plan.set_values(sym.Lx * scale)
        plan.factorize()
        a = plan.get_factor()
        plan.factorize()
        b = plan.get_factor()
        assert rel_err(a, b, mask) <= 1e-14
        ref, info, _ = oracle.chol_factorize(dict(
            n=n, nsuper=sym.nsuper, Super=sym.Super, SuperMap=sym.SuperMap, Lsip=sym.Lsip, Lsi=sym.Lsi,
            Lsxp=sym.Lsxp, Lp=sym.Lp, Li=sym.Li, Lx=sym.Lx * scale, LeafQueue=sym.LeafQueue,
            nsleaf=sym.nsleaf, csize=sym.csize, xsize=sym.xsize))
        assert rel_err(a, ref, mask) <= TOL_FACTOR
    plan.close()


@pytest.mark.parametrize("N", [24, 40])
def test_medium_3d_against_oracle(oracle, N):
    """sizes with supernodes of several hundred to ~2500 columns: multi-step blocked panels, trailing
    updates, multi-tile Schur updates"""
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N), sf.REFERENCE_SLOT_1GPU)
    plan, Lsx = gpu_factor(sym)
    ref, info, _ = oracle.chol_factorize(sym)
    assert info == 0
    assert rel_err(Lsx, ref, oracle.lower_mask(sym)) <= TOL_FACTOR
    res, _ = oracle.chol_residual(sym, Lsx)
    assert res <= TOL_RESIDUAL
    assert abs(plan.stat("flops_exec") - sym.flops_exec) <= 1e-9 * sym.flops_exec
    plan.close()


def test_config3_like_2d_wide_stencil(oracle):
    """BASELINE config 3's generator at reduced size (200x200 grid, 21-point stencil, 2-line separators)"""
    n, Cp, Ci, Cx = gen.stencil_spd_lower(200, 200)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(200, 200, 1, 3, 2), sf.REFERENCE_SLOT_1GPU)
    plan, Lsx = gpu_factor(sym)
    ref, info, _ = oracle.chol_factorize(sym)
    assert info == 0
    assert rel_err(Lsx, ref, oracle.lower_mask(sym)) <= TOL_FACTOR
    res, _ = oracle.chol_residual(sym, Lsx)
    assert res <= TOL_RESIDUAL
    plan.close()


def test_large_properties_64cubed(oracle):
    """64^3 (n = 262,144): too big for a dense check; properties instead --
    residual of the reference's validate(), and (L L^T)_{ij} = A_{ij} on sampled rows via the solve"""
    N = 64
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N), sf.REFERENCE_SLOT_1GPU)
    plan, Lsx = gpu_factor(sym)
    res, x = oracle.chol_residual(sym, Lsx)
    assert res <= TOL_RESIDUAL
    # linearity of the solve: (LL^T)^{-1}(2b) = 2 (LL^T)^{-1} b to rounding
    b = 1 + np.arange(n) / n
    x2 = oracle.chol_solve(sym, Lsx, 2 * b)
    assert np.allclose(x2, 2 * x, rtol=1e-13, atol=0)
    plan.close()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_plans_emulated_ranks_on_one_gpu(oracle, world):
    """the multi-GPU sharding with all ranks' plans built on the single test GPU: phase 0 on every
    rank, the top regions summed through torch tensors aliasing the plans' device memory (what the RCCL
    all-reduce does between GPUs), phase 1 on every rank, partial downloads merged on the host."""
    import torch
    from importlib import import_module
    sharded = import_module("sparse-matrix-factorization-library_amd.sharded")
    N = 20
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N), 1 << 30)
    owner, tf, ml = sf.subtree_partition(sym, world)
    assert (owner < 0).any()
    engines = [sharded.HipEngine(sym, sf.phases_for_rank(owner, r), r == 0, 0) for r in range(world)]
    for rep in range(2):                                   # plans are reusable
        for e in engines:
            e.set_values(sym.Lx)
            e.factorize_phase(0)
        tops = [e.top_tensor() for e in engines]
        assert tops[0].is_cuda and tops[0].dtype == torch.float64 and tops[0].numel() == engines[0].plan.top_region()[1]
        total = tops[0].clone()
        for t in tops[1:]:
            total += t
        for t in tops:
            t.copy_(total)
        torch.cuda.synchronize()
        for e in engines:
            e.factorize_phase(1)
        full = np.zeros(sym.xsize)
        for e in engines:
            e.get_factor(full)
        ref, info, _ = oracle.chol_factorize(sym)
        assert rel_err(full, ref, oracle.lower_mask(sym)) <= TOL_FACTOR
        res, _ = oracle.chol_residual(sym, full)
        assert res <= TOL_RESIDUAL
    stored = [e.plan.stat("stored_doubles") for e in engines]
    assert max(stored) < sym.xsize                          # compact per-rank storage
    for e in engines:
        e.close()


@pytest.mark.parametrize("N,world", [(24, 2), (24, 3), (24, 8), (10, 8)])
def test_distributed_top_emulated_ranks_on_one_gpu(oracle, N, world):
    """distributed top (sf_chol_plan_create_distributed) with every rank's plan on the single test GPU: phase 0,
    then per segment the regions are summed through torch tensors aliasing the plans' device memory (the RCCL
    all-reduce between GPUs) and every rank runs the segment with ITS share of the split GEMM launches.
    24^3: the root separator has 576 > 512 columns, so a split left-looking outer GEMM is part of the run.
    10^3 on 8 ranks: top separators of 25 and 50 columns, so split k_update_small launches (K <= 64) are part of it."""
    import torch
    from importlib import import_module
    sharded = import_module("sparse-matrix-factorization-library_amd.sharded")
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N), 1 << 30)
    owner, tf, ml = sf.subtree_partition(sym, world, 1.0 / world + sharded.TOP_CHAIN_SHARE)
    ntop = int((owner < 0).sum())
    assert ntop > 0
    engines = [sharded.HipEngine(sym, sf.phases_for_rank(owner, r), r == 0, 0, r, world, True) for r in range(world)]
    nseg = engines[0].num_segments()
    assert nseg >= (2 if N >= 24 else 1) and all(e.num_segments() == nseg for e in engines)
    ref, info, _ = oracle.chol_factorize(sym)
    for rep in range(2):                                   # plans are reusable
        for e in engines:
            e.set_values(sym.Lx)
            e.factorize_phase(0)
        for k in range(nseg):
            parts = [e.segment_tensors(k) for e in engines]
            assert all(len(p) == len(parts[0]) for p in parts)
            for i in range(len(parts[0])):
                total = parts[0][i].clone()
                for p in parts[1:]:
                    assert p[i].numel() == total.numel()
                    total += p[i]
                for p in parts:
                    p[i].copy_(total)
            for e in engines:
                e.factorize_segment(k)
        for e in engines:
            e.finish()
        full = np.zeros(sym.xsize)
        for e in engines:
            e.get_factor(full)
        assert rel_err(full, ref, oracle.lower_mask(sym)) <= TOL_FACTOR
        res, _ = oracle.chol_residual(sym, full)
        assert res <= TOL_RESIDUAL
        # the replicated chains leave the same top panels on every rank (to rounding: stream-K partial tiles use atomics)
        t0 = engines[0].get_factor()
        for e in engines[1:]:
            te = e.get_factor()
            for s_ in np.flatnonzero(owner < 0):
                a, b = sym.Lsxp[s_], sym.Lsxp[s_ + 1]
                assert np.abs(t0[a:b] - te[a:b]).max() <= 1e-13 * np.abs(t0[a:b]).max()
    # a distributed plan refuses the replicated phase-1 entry point
    with pytest.raises(RuntimeError):
        engines[0].plan.factorize_phase(1)
    # packing a second segment before the first packed one has run, or running another segment than the packed one, is an error
    engines[0].set_values(sym.Lx)
    engines[0].factorize_phase(0)
    engines[0].plan.segment_pack(0)
    with pytest.raises(RuntimeError):
        engines[0].plan.segment_pack(0)
    if nseg > 1:
        with pytest.raises(RuntimeError):
            engines[0].plan.factorize_segment(1)
    engines[0].plan.factorize_segment(0, sync=True)
    # region queries: capacity too small / segment out of range
    assert len(engines[0].plan.segment_regions(0)) >= 1
    with pytest.raises(RuntimeError):
        engines[0].plan.segment_regions(nseg)
    for e in engines:
        e.close()


@pytest.mark.parametrize("seed,n_,per_col,band", [(1, 1500, 3, None), (2, 4000, 2, None), (3, 3000, 6, 400), (4, 6000, 4, 60),
                                                   (5, 800, 12, None), (6, 2500, 5, 1200)])
def test_random_patterns_with_builtin_ordering(oracle, seed, n_, per_col, band):
    """random SPD patterns (uniformly random: heavy fill and wide root supernodes; banded: chains of panels) ordered by the
    built-in nested dissection -- irregular supernode widths, update shapes and tree depths through every kernel"""
    n, Cp, Ci, Cx = gen.random_spd_lower(n_, per_col, seed=seed, bandwidth=band)
    perm = sf.graph_nd_perm(n, Cp, Ci)
    sym = sf.analyze(n, Cp, Ci, Cx, perm, sf.REFERENCE_SLOT_1GPU)
    plan, Lsx = gpu_factor(sym)
    ref, info, _ = oracle.chol_factorize(sym)
    assert info == 0
    assert rel_err(Lsx, ref, oracle.lower_mask(sym)) <= TOL_FACTOR
    res, _ = oracle.chol_residual(sym, Lsx)
    assert res <= TOL_RESIDUAL
    b = 1.0 + np.arange(n) / n
    assert sf.validate_solution(sym, plan.solve(b), b) <= 1e-12
    plan.close()


def test_rccl_world1_collectives_on_plan_memory():
    """the real RCCL backend on the plan's memory and stream (one-rank group; see tests/_nccl_world1_worker.py)"""
    import socket
    import subprocess
    import sys
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, os.path.join(here, "_nccl_world1_worker.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0 and "RCCL_WORLD1_OK" in p.stdout, p.stdout[-3000:]


def test_rccl_c_path_glue_one_rank():
    """the multi-GPU product path end to end on a one-rank nccl group (tests/_rccl_c_glue_worker.py): unique id over
    torch.distributed, communicator made by the C library, mapped plan, sf_chol_plan_factorize_distributed, Cholesky and LU"""
    import socket
    import subprocess
    import sys
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               SF_FORCE_DISTRIBUTED="1")
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, os.path.join(here, "_rccl_c_glue_worker.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0 and "RCCL_C_GLUE_OK" in p.stdout, p.stdout[-3000:]


@pytest.mark.parametrize("comm,method", [("rccl-c", "cholesky"), ("rccl-c", "lu"), ("torch", "cholesky")])
def test_bench_multi_gpu_code_path_on_one_rank(comm, method):
    """bench.py's N > 1 branch (nccl group, sharded factorization, barriers and the max-over-ranks timing, the distributed
    solve and its residual) forced onto ONE rank (SF_FORCE_DISTRIBUTED=1): the line must come out, name the collectives that
    ran, and carry a residual at rounding level.  `torch` = the fall-back with torch.distributed's all-reduce."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               SF_FORCE_DISTRIBUTED="1", SF_BENCH_COMM=comm)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--grid", "24",
           "--method", method, "--cpu-grid", "0"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0
    sh = line["config"]["sharding"]
    assert sh is not None and sh["mode"] == "distributed" and sh["collectives"] == comm, sh
    if comm == "rccl-c":
        assert line["config"]["residual_distributed_solve"] <= 1e-13, line["config"]


@pytest.mark.parametrize("case", small_cases(), ids=lambda c: c[0])
def test_device_solve_matches_oracle(oracle, case):
    """sf_chol_plan_solve (level-scheduled, factor resident) vs the reference's host loops (oracle restatement)"""
    name, n, Cp, Ci, Cx, perm, slot = case
    sym = sf.analyze(n, Cp, Ci, Cx, perm, slot)
    plan, Lsx = gpu_factor(sym)
    b = 1 + np.arange(n) / n
    x = plan.solve(b)
    want = oracle.chol_solve(sym, Lsx, b)
    assert np.allclose(x, want, rtol=1e-12, atol=1e-13 * np.abs(want).max())
    plan.close()


@pytest.mark.parametrize("knobs", [
    {"SF_SOLVE_FAR_WGS": "1", "SF_SOLVE_FAR_GROUPS": "64"},      # far tiles merged into multi-group tasks even on this small matrix
    {"SF_SOLVE_FAR_WGS": "1", "SF_SOLVE_FAR_GROUPS": "3", "SF_SOLVE_FWD_FAR_FIRST": "0"},
    {"SF_SOLVE_BWD_AHEAD": "0", "SF_SOLVE_FWD_AHEAD": "0", "SF_SOLVE_DIAGT": "0"},     # the sweeps without look-ahead
    {"SF_SOLVE_BWD_FUSED": "0"}], ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()))
@pytest.mark.parametrize("method", ["cholesky", "lu"])
def test_device_solve_schedules(oracle, monkeypatch, knobs, method):
    """the solve's schedule variants (look-ahead of the far row tiles, multi-group far tasks, list order, row-major diagonal
    copies, two-launch backward steps) against the reference's host loops (oracle restatement); the matrix has supernodes of
    several 256-column steps, so every variant has something to do"""
    needs_experiments(knobs)
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    N = 34
    if method == "lu":
        n, Cp, Ci, Cx = gen.unsymmetric_stencil(N, N, N, seed=5)
        sym = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(N, N, N), 4 << 30, "lu", False)
        plan = sf.LUPlan(sym)
        plan.set_values(sym.Lx, sym.Ux)
    else:
        n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
        sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N), 4 << 30)
        plan = sf.CholPlan(sym)
        plan.set_values(sym.Lx)
    assert np.diff(sym.Super).max() > 1024
    plan.factorize()
    Lsx = plan.get_factor()
    b = 1 + np.arange(n) / n
    x = plan.solve(b)
    want = (oracle.lu_solve if method == "lu" else oracle.chol_solve)(sym, Lsx, b)
    assert np.allclose(x, want, rtol=1e-11, atol=1e-12 * np.abs(want).max())
    plan.close()


def test_device_solve_residual_48cubed(oracle):
    N = 48
    n, Cp, Ci, Cx = gen.laplacian_lower(N, N, N)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(N, N, N), sf.REFERENCE_SLOT_1GPU)
    plan = sf.CholPlan(sym)
    plan.set_values(sym.Lx)
    plan.factorize()
    b = 1 + np.arange(n) / n
    x = plan.solve(b)
    # residual of the reference's validate() computed from the device solution (A = stored triangle used symmetrically)
    r = -b.copy()
    Lp, Li, Lx = sym.Lp, sym.Li, sym.Lx
    cols = np.repeat(np.arange(n), np.diff(Lp))
    np.add.at(r, Li, Lx * x[cols])
    off = Li != cols
    np.add.at(r, cols[off], Lx[off] * x[Li[off]])
    colsum = np.zeros(n)
    np.add.at(colsum, cols, np.abs(Lx))
    np.add.at(colsum, Li[off], np.abs(Lx[off]))
    res = np.abs(r).max() / (colsum.max() * np.abs(x).max() + np.abs(b).max())
    assert res <= TOL_RESIDUAL
    assert plan.stat("last_solve_ms") > 0
    plan.close()


def test_three_launch_form_forced(oracle, monkeypatch):
    """SF_FUSE_MAX=0 (read at plan creation) turns every fused k_step launch into its three-launch form -- stream-K GEMM,
    one-wave POTRF, TRSM workgroups -- so that both forms stay covered; same factor either way"""
    n, Cp, Ci, Cx = gen.laplacian_lower(14, 14, 14)
    sym = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(14, 14, 14), 8 << 30)
    assert np.diff(sym.Super).max() > 128
    ref, info, _ = oracle.chol_factorize(sym)
    mask = oracle.lower_mask(sym)
    monkeypatch.delenv("SF_FUSE_MAX", raising=False)       # (the suite may be running under the knob itself)
    plan, fused = gpu_factor(sym)
    n_fused = plan.stat("launches")
    plan.close()
    monkeypatch.setenv("SF_FUSE_MAX", "0")
    plan, three = gpu_factor(sym)
    assert plan.stat("launches") > n_fused
    plan.close()
    assert rel_err(fused, ref, mask) <= TOL_FACTOR and rel_err(three, ref, mask) <= TOL_FACTOR


def test_two_plans_concurrently_on_one_gpu(oracle):
    """the flag hand-off of k_step under contention: two plans (two streams) factorize at the same time on one GPU, each
    with fused steps of more workgroups than fit the chip next to the other's.  Tasks are claimed by ticket in execution
    order, so no wait can starve its producer; SF_ERR_HIP (info bit 2: a bounded wait ran out) would fail the check."""
    n, Cp, Ci, Cx = gen.laplacian_lower(40, 40, 40)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(40, 40, 40), 8 << 30)
    plans = [sf.CholPlan(sym, device=0) for _ in range(2)]
    for k, p in enumerate(plans):
        p.set_values(sym.Lx * (1.0 + k))
    for _ in range(3):
        for p in plans:
            p.factorize(sync=False)
    for p in plans:
        p.sync()                                   # raises SparseFrameError on SF_ERR_HIP / SF_ERR_NOT_POSDEF
    ref, info, _ = oracle.chol_factorize(sym)
    mask = oracle.lower_mask(sym)
    for k, p in enumerate(plans):
        Lsx = p.get_factor()
        assert rel_err(Lsx, ref * np.sqrt(1.0 + k), mask) <= TOL_FACTOR
        p.close()


def test_factorize_to_host_overlapped_download(oracle):
    """sf_chol_plan_factorize_to_host: the factor copied back piece by piece while the upper levels compute must be the
    factor a plain factorize + get_factor returns, bit for bit where the kernels are deterministic (no atomics race on a
    single panel chain) and to rounding elsewhere; run twice (the plan, its pinned ring and the piece events are re-used)"""
    n, Cp, Ci, Cx = gen.laplacian_lower(24, 24, 24)
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(24, 24, 24), 8 << 30)
    plan = sf.CholPlan(sym, device=0)
    assert plan.stat("download_pieces") >= 1
    ref, info, _ = oracle.chol_factorize(sym)
    mask = oracle.lower_mask(sym)
    out = np.full(sym.xsize, np.nan)
    for scale in (1.0, 2.0):
        got = plan.factorize_to_host(sym.Lx * scale, out)
        assert not np.isnan(got[mask]).any()
        assert rel_err(got, ref * np.sqrt(scale), mask) <= TOL_FACTOR
        assert plan.stat("last_to_host_ms") > 0
    plan.close()


def test_struct_entry_point_reuses_cached_plan(oracle):
    """SparseFrame_factorize twice on one handler list with the same pattern and new values: the second call finds the
    plan in the handler's cache (no rebuild) and still returns the right factor; a third matrix with another pattern
    evicts nothing it should not"""
    n, Cp, Ci, Cx = gen.laplacian_lower(16, 16, 16)
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    perm = sf.grid_nd_perm(16, 16, 16)
    sym = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30)
    ref, info, _ = oracle.chol_factorize(sym)
    mask = oracle.lower_mask(sym)
    builds = []
    for scale in (1.0, 4.0, 9.0):
        mi = sf.MatrixInfo()
        mi.set_csc(n, Cp, Ci, Cx * scale)
        mi.set_perm(perm)
        mi.analyze(common)
        mi.factorize(common)
        builds.append(common.plan_builds())
        assert rel_err(mi.array("Lsx", sym.xsize).copy(), ref * np.sqrt(scale), mask) <= TOL_FACTOR
        assert mi.validate() <= TOL_RESIDUAL
        mi.cleanup()
    n2, Cp2, Ci2, Cx2 = gen.laplacian_lower(30, 30)
    mi = sf.MatrixInfo()
    mi.set_csc(n2, Cp2, Ci2, Cx2)
    mi.analyze(common)
    mi.factorize(common)
    assert mi.validate() <= TOL_RESIDUAL
    mi.cleanup()
    assert builds == [1, 1, 1] and common.plan_builds() == 2    # one plan per pattern, however often it is factorized
    common.close()


def test_gemm_register_staged_form(oracle, monkeypatch):
    """k_gemm stages its operand tiles by LDS-DMA (global_load_lds_dwordx4) by default; SF_GEMM_DMA=0 selects the
    register-staged form.  Both must give the oracle's factor (wide supernodes: K tails, odd panel offsets, edge tiles)."""
    n, Cp, Ci, Cx = gen.laplacian_lower(18, 17, 19)
    sym = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(18, 17, 19), 8 << 30)
    ref, info, _ = oracle.chol_factorize(sym)
    mask = oracle.lower_mask(sym)
    for dma in ("1", "0"):
        if dma == "0" and not sf.lib.sf_build_experiments():
            continue                        # the register-staged form is compiled out of release builds (make EXP=1)
        monkeypatch.setenv("SF_GEMM_DMA", dma)
        plan, Lsx = gpu_factor(sym)
        plan.close()
        assert rel_err(Lsx, ref, mask) <= TOL_FACTOR, dma


def test_two_matrix_threads_share_one_handler_list(oracle):
    """the reference's driver runs MATRIX_THREAD_NUM = 2 matrices at a time over ONE gpu_info_list (SparseFrame.c:3371-3375):
    two host threads call SparseFrame_factorize concurrently (ctypes releases the GIL); the handler's lock serialises them,
    both patterns stay cached, both factors are right"""
    import threading
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    cases = []
    for dims in ((14, 14, 14), (60, 50, 1)):
        n, Cp, Ci, Cx = gen.laplacian_lower(*dims)
        perm = sf.grid_nd_perm(*dims)
        sym = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30)
        ref, info, _ = oracle.chol_factorize(sym)
        cases.append((n, Cp, Ci, Cx, perm, sym, ref, oracle.lower_mask(sym)))
    errors = []

    def work(k, serial):
        n, Cp, Ci, Cx, perm, sym, ref, mask = cases[k]
        try:
            for rep in range(3):
                mi = sf.MatrixInfo(serial=serial)
                mi.set_csc(n, Cp, Ci, Cx)
                mi.set_perm(perm)
                mi.analyze(common)
                mi.factorize(common)
                if rel_err(mi.array("Lsx", sym.xsize).copy(), ref, mask) > TOL_FACTOR:
                    errors.append((k, rep))
                mi.cleanup()
        except Exception as e:      # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k, k)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    common.close()
    assert not errors, errors


@pytest.mark.parametrize("case", ["chol2d", "chol3d", "lu_unsym", "lu_sym", "lu_pivoted"])
def test_device_validate_matches_host_residual(oracle, case):
    """sf_chol_plan_validate (b_i = 1 + i/n, device solve, residual kernels over the plan's copy of A) against the host
    statements of the reference's validate() (C:3182-3263 / L:3702-3858): numpy on the same x, and the oracle's own
    residual of the downloaded factor"""
    import scipy.sparse as sp
    lu = case.startswith("lu")
    if case == "chol2d":
        n, Cp, Ci, Cx = gen.laplacian_lower(48, 48); perm = nd_perm_py(48, 48, 1)
    elif case in ("chol3d", "lu_sym"):
        n, Cp, Ci, Cx = gen.laplacian_lower(12, 12, 12); perm = nd_perm_py(12, 12, 12)
    elif case == "lu_unsym":
        n, Cp, Ci, Cx = gen.unsymmetric_stencil(11, 11, 11, seed=3); perm = nd_perm_py(11, 11, 11)
    else:
        n, Cp, Ci, Cx = gen.unsymmetric_general(10, 10, 10, seed=21); perm = nd_perm_py(10, 10, 10)
    if lu:
        sym = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30, "lu", case == "lu_sym")
        plan = sf.LUPlan(sym)
        plan.set_values(sym.Lx, None if case == "lu_sym" else sym.Ux)
        if case == "lu_pivoted":
            plan.set_pivoting(1.0)
    else:
        sym = sf.analyze(n, Cp, Ci, Cx, perm, 1 << 30)
        plan = sf.CholPlan(sym)
        plan.set_values(sym.Lx)
    plan.factorize()
    res, x = plan.validate(return_x=True)
    b = 1.0 + np.arange(n) / n
    x2 = plan.solve(b)                                    # two runs of one solve differ by the order of its fp64 atomics
    assert np.max(np.abs(x - x2)) <= 1e-12 * np.max(np.abs(x2))
    # the matrix the factorization works on, from the analysis
    lc = np.repeat(np.arange(n), np.diff(sym.Lp))
    A = sp.coo_matrix((sym.Lx, (sym.Li, lc)), shape=(n, n)).tocsr()
    if lu and case != "lu_sym":
        ur = np.repeat(np.arange(n), np.diff(sym.Up))
        off = sym.Ui != ur
        A = A + sp.coo_matrix((sym.Ux[off], (ur[off], sym.Ui[off])), shape=(n, n)).tocsr()
    else:
        off = sym.Li != lc
        A = A + sp.coo_matrix((sym.Lx[off], (lc[off], sym.Li[off])), shape=(n, n)).tocsr()
    r = A @ x - b
    ref = float(np.abs(r).max() / (abs(A).sum(axis=0).max() * np.abs(x).max() + np.abs(b).max()))
    assert res <= (1e-10 if case == "lu_pivoted" else 1e-13)
    eps = np.finfo(np.float64).eps
    assert abs(res - ref) <= 16 * eps                    # same formula; only the summation order inside r differs
    if not lu:
        assert abs(sf.validate_solution(sym, x) - ref) <= 16 * eps
    plan.close()


def test_struct_solve_runs_on_the_resident_factor(oracle, monkeypatch):
    """SparseFrame_solve_supernodal only receives matrix_info; after SparseFrame_factorize on one handler it finds the factor still
    resident in the handler's plan (by the address of Lsx) and solves there.  Same solution as the reference's host solve
    (SF_SOLVE=host); a host copy the caller changed, or a plan that meanwhile holds another factorization, falls back to the host."""
    from importlib import import_module
    lib = import_module("sparse-matrix-factorization-library_amd._lib").lib
    n, Cp, Ci, Cx = gen.laplacian_lower(20, 20, 20)
    perm = sf.grid_nd_perm(20, 20, 20)
    common = sf.CommonInfo(dev_slot_size=1 << 30)
    mi = sf.MatrixInfo()
    mi.set_csc(n, Cp, Ci, Cx)
    mi.set_perm(perm)
    mi.analyze(common)
    mi.factorize(common)
    before = lib.sf_handlers_resident_solves()
    assert mi.validate() <= TOL_RESIDUAL
    assert lib.sf_handlers_resident_solves() == before + 1
    x_dev = mi.array("Xx", n).copy()
    monkeypatch.setenv("SF_SOLVE", "host")
    assert mi.validate() <= TOL_RESIDUAL
    assert lib.sf_handlers_resident_solves() == before + 1
    x_host = mi.array("Xx", n).copy()
    assert np.max(np.abs(x_dev - x_host)) <= 1e-12 * np.max(np.abs(x_host))
    monkeypatch.delenv("SF_SOLVE")
    # the caller changes ONE value of its host copy, anywhere (here: the last bit of a diagonal entry of a mid-size panel, far from
    # the three windows the round-2 check sampled): every panel's fingerprint is compared, the solve follows the HOST data
    Lsx = mi.array("Lsx", int(mi.c.xsize))
    s_mid = int(np.argmin(np.abs(np.asarray(mi.array("Lsxp", int(mi.c.nsuper) + 1)) - int(mi.c.xsize) // 3)))
    at = int(mi.array("Lsxp", int(mi.c.nsuper) + 1)[s_mid])
    keep = float(Lsx[at])
    Lsx[at] = np.nextafter(keep, 2 * keep)
    fb = lib.sf_handlers_fingerprint_fallbacks()
    mi.validate()
    assert lib.sf_handlers_resident_solves() == before + 1            # host fallback
    assert lib.sf_handlers_fingerprint_fallbacks() == fb + 1           # ... and it is counted, not silent
    Lsx[at] = keep
    mi.factorize(common)                                               # resident again
    # trusted mode: the caller vouches for Lsx, nothing is compared -- the device's factor answers even though the host copy differs
    assert lib.sf_handlers_set_resident_solve(2) == 0
    Lsx[at] = 3.0 * keep
    mi.validate()
    assert lib.sf_handlers_resident_solves() == before + 2
    assert np.max(np.abs(mi.array("Xx", n) - x_host)) <= 1e-12 * np.max(np.abs(x_host))
    Lsx[at] = keep
    assert lib.sf_handlers_set_resident_solve(0) == 0                  # never: always the reference's host sweep
    mi.validate()
    assert lib.sf_handlers_resident_solves() == before + 2
    assert lib.sf_handlers_set_resident_solve(1) == 0 and lib.sf_handlers_set_resident_solve(7) != 0
    mi.validate()
    assert lib.sf_handlers_resident_solves() == before + 3            # verified mode, unmodified copy: resident
    before += 2
    # the caller scales its host copy: the solve follows the HOST data (x scales by 1/4)
    Lsx *= 2.0
    mi.validate()
    assert lib.sf_handlers_resident_solves() == before + 1
    assert np.max(np.abs(mi.array("Xx", n) - x_host / 4.0)) <= 1e-12 * np.max(np.abs(x_host))
    # a second matrix of the same pattern takes over the cached plan: the first matrix's entry is gone
    mi2 = sf.MatrixInfo()
    mi2.set_csc(n, Cp, Ci, Cx * 9.0)
    mi2.set_perm(perm)
    mi2.analyze(common)
    mi.factorize(common)                    # mi resident again ...
    mi2.factorize(common)                   # ... until mi2 re-uses the plan
    k = lib.sf_handlers_resident_solves()
    mi.validate()
    assert lib.sf_handlers_resident_solves() == k            # host fallback, still right
    assert mi.c.residual <= TOL_RESIDUAL
    mi2.validate()
    assert lib.sf_handlers_resident_solves() == k + 1 and mi2.c.residual <= TOL_RESIDUAL
    mi.cleanup(); mi2.cleanup(); common.close()

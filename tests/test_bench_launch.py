"""bench.py --gpus N must never print a value that fewer than N ranks produced (round-2 verdict, item 1).

CPU part (this container has no GPU): the guards that run before any device is touched.
GPU part (one-GPU box): `python3 bench.py --gpus 2` with NO launcher and NO WORLD_SIZE starts its two ranks itself; they share
device 0 over gloo (SF_BENCH_BACKEND / SF_BENCH_DEVICE are rehearsal hooks: RCCL refuses two ranks on one device), and the line
must say it saw two ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SF_FORCE_DISTRIBUTED", "SF_BENCH_SELF_LAUNCHED")}
    env.update(extra)
    return env


def _json_lines(text):
    out = []
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                out.append(json.loads(ln))
            except ValueError:
                pass
    return out


def test_world_size_must_equal_gpus():
    """one process, --gpus 2: a launcher-style environment that says WORLD_SIZE=1 is refused before anything runs"""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--grid", "8", "--cpu-grid", "0"],
                       env=_clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0
    assert "WORLD_SIZE=1" in p.stderr and "--gpus 2" in p.stderr, p.stderr[-2000:]
    assert not _json_lines(p.stdout), p.stdout[-2000:]


def test_no_launcher_starts_the_ranks_itself_and_never_prints_a_one_rank_value():
    """no WORLD_SIZE: the parent starts two ranks (torch.distributed.run).  In this container they find no GPU, so the job must
    fail as a whole -- non-zero exit, no JSON line; what may NOT happen is one process printing a value multiplied by --gpus"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side guard test (the GPU variant below runs the two ranks for real)")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--grid", "8", "--cpu-grid", "0"],
                       env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert p.returncode != 0, (p.stdout[-1000:], p.stderr[-2000:])
    assert "starting 2 ranks" in p.stderr, p.stderr[-3000:]
    assert not _json_lines(p.stdout), p.stdout[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("mp", ["subtree", "replicas"])
def test_gpus_2_without_launcher_runs_two_ranks(mp):
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--grid", "24", "--cpu-grid", "0",
                        "--mp", mp],
                       env=_clean_env(SF_BENCH_BACKEND="gloo", SF_BENCH_DEVICE="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout[-3000:]
    line = lines[0]
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["value"] > 0, line
    assert line["collectives"] == "torch"            # gloo rehearsal: the C-side RCCL communicator needs one device per rank
    if mp == "subtree":
        assert line["config"]["sharding"]["mode"] == "distributed"
        assert line["scaling"] == "strong" and line["config"]["grid_rule"] == "explicit --grid"
        sec = line["secondary"]["weak_flops"]
        assert "error" not in sec and sec["grid"] == 27 and sec["GFLOPs"] > 0, sec
    else:
        assert line["config"]["sharding"] is None and "independent" in line["config"]["parallelism"]


@pytest.mark.gpu
def test_ranks_seen_mismatch_gives_no_value():
    """a job whose communicator counts fewer ranks than --gpus exits non-zero without a line: forced one-rank group
    (SF_FORCE_DISTRIBUTED) but --gpus 1 is the only consistent call; WORLD_SIZE=2 with one live rank cannot even rendezvous, so
    the check exercised here is the in-process one, through the test hook SF_TEST_RANKS_SEEN_DELTA"""
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = _clean_env(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                     SF_FORCE_DISTRIBUTED="1", SF_TEST_RANKS_SEEN_DELTA="-1")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1", "--warmup", "0", "--grid", "16", "--cpu-grid", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 3, (p.returncode, p.stderr[-3000:])
    assert "counted 0 ranks" in p.stderr and not _json_lines(p.stdout)

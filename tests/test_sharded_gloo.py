"""N > 1 path on CPU: gloo ranks run ShardedCholesky (subtree partition, sum all-reduces of top-panel regions,
replicated and distributed top) with the test-only numpy engine and must reproduce the oracle's factor."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from util import sf, gen, nd_perm_py

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_partition_properties():
    n, Cp, Ci, Cx = gen.laplacian_lower(10, 10, 10)
    sym = sf.analyze(n, Cp, Ci, Cx, nd_perm_py(10, 10, 10), 1 << 30)
    ncol, nrow = np.diff(sym.Super), np.diff(sym.Lsip)
    parent = np.array([sym.SuperMap[sym.Lsi[sym.Lsip[s] + ncol[s]]] if nrow[s] > ncol[s] else -1 for s in range(sym.nsuper)])
    for world in (1, 2, 3, 4, 8):
        for tw in (1.0, 1.0 / world + 0.25):
            owner, tf, ml = sf.subtree_partition(sym, world, tw)
            owner2, _, _ = sf.subtree_partition(sym, world, tw)
            assert np.array_equal(owner, owner2)                      # deterministic: every rank computes the same map
            assert owner.min() >= -1 and owner.max() < world
            if world == 1:
                assert (owner == 0).all() and tf == 0.0
            for s in range(sym.nsuper):
                p = parent[s]
                if p >= 0:
                    assert owner[p] == -1 or owner[p] == owner[s]     # a subtree is closed under descendants
                    if owner[s] == -1:
                        assert owner[p] == -1                         # the top set is closed under ancestors
            # every update target of a stored supernode is stored on the same rank (own subtree or top)
            for r in range(world):
                ph = sf.phases_for_rank(owner, r)
                for s in np.flatnonzero(ph >= 0):
                    tg = sym.SuperMap[sym.Lsi[sym.Lsip[s] + ncol[s]:sym.Lsip[s + 1]]]
                    assert (ph[tg] >= 0).all()
            assert 0 <= tf < 1 and 0 < ml <= 1


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ranks_match_oracle(world):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_sharded_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARDED_OK" in outs[0], outs[0]

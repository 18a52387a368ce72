#!/usr/bin/env python3
"""Compute-side timing of ONE rank of a W-rank distributed factorization on a single GPU (no communication).

    python tools/emulate_rank.py --grid 128 --world 8 [--rank 0]

Runs rank `--rank`'s phase 0 and all its top segments back to back WITHOUT the all-reduces, so the numbers in the
factor are meaningless (the other ranks' contributions are missing) but the launches, their sizes and therefore the
time are exactly those of that rank in a real W-GPU run.  T(W GPUs) ~ max over ranks of this time + the all-reduce
time of `top_bytes` (printed).  Used for the scaling model in DESIGN.md section 6; the real curve comes from the
driver's N = 1, 2, 4, 8 runs of bench.py.
"""
import argparse
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--slot", type=int, default=-1,
                    help="devSlotSize of the analysis (default: the reference formula for `world` 288 GiB devices, as bench.py --gpus N "
                         "uses; 0 = the 1-GPU value the round-2 tables were measured with)")
    ap.add_argument("--replicated-top", action="store_true",
                    help="round-1 layout: every rank stores every top panel (default: proportional mapping, a top supernode lives on "
                         "the ranks whose subtrees lie below it)")
    args = ap.parse_args()
    import numpy as np
    import torch
    sf = importlib.import_module("sparse-matrix-factorization-library_amd")
    sharded = importlib.import_module("sparse-matrix-factorization-library_amd.sharded")
    g, W = args.grid, args.world
    n, Cp, Ci, Cx = sf.gen.laplacian_lower(g, g, g)
    slot = sf.REFERENCE_SLOT_1GPU if args.slot == 0 else (int(sf.lib.sf_reference_slot_size(W, 288 << 30)) if args.slot < 0 else args.slot)
    t_a = time.time()
    sym = sf.analyze(n, Cp, Ci, Cx, sf.grid_nd_perm(g, g, g, 3, 1), slot)
    t_a = time.time() - t_a
    owner, tf, ml = sf.subtree_partition(sym, W, 1.0 / W + sharded.TOP_CHAIN_SHARE)
    t_p = time.time()
    if args.replicated_top:
        eng = sharded.HipEngine(sym, sf.phases_for_rank(owner, args.rank), args.rank == 0, 0, args.rank, W, True)
    else:
        # (the mapped plan directly: building the replicated-top engine first would allocate every top panel -- 256^3 does not fit)
        eng = sharded.HipEngine.__new__(sharded.HipEngine)
        eng.device, eng.lu, eng.distributed, eng.comm, eng.comm_kind, eng._top, eng._seg = 0, False, True, None, "none", None, {}
        eng.plan = sf.CholPlan(sym, device=0, owner=owner, rank=args.rank, nranks=W)
        eng.plan.set_stream(torch.cuda.current_stream(0).cuda_stream)
    t_p = time.time() - t_p
    eng.set_values(sym.Lx)
    nseg = eng.num_segments()
    full_doubles = sum(c for k in range(nseg) for (_, c) in eng.plan.segment_regions(k))

    def run():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.factorize_phase(0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        packed = 0
        for k in range(nseg):
            packed += sum(t.numel() for t in eng.segment_tensors(k))     # pack (as in a real run), no all-reduce
            eng.factorize_segment(k)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return (t1 - t0) * 1e3, (t2 - t1) * 1e3, packed

    run()
    res = [run() for _ in range(args.steps)]
    p0 = min(r[0] for r in res)
    p1 = min(r[1] for r in res)
    print(json.dumps({"grid": g, "world": W, "rank": args.rank, "devSlotSize": slot, "n": int(n), "nsuper": int(sym.nsuper),
                      "F_struct": sym.flops_struct, "analyze_s": round(t_a, 2), "plan_create_s": round(t_p, 2),
                      "phase0_ms": round(p0, 2), "segments_ms": round(p1, 2), "segments": nseg,
                      "allreduce_calls": nseg, "allreduce_bytes": 8 * res[0][2], "top_panel_bytes": 8 * full_doubles,
                      "top_flop_fraction": tf, "max_rank_subtree_flop_fraction": ml,
                      "stored_doubles": eng.plan.stat("stored_doubles"),
                      "layout": "replicated top" if args.replicated_top else "proportional mapping",
                      "segment_groups": sorted({bin(eng.plan.segment_group(k)).count("1") for k in range(nseg)})}))
    eng.close()


if __name__ == "__main__":
    main()

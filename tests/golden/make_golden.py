"""Generates tests/golden/chol_small.json.

The reference cannot run in this image (it needs CUDA/cuBLAS/cuSOLVER/MAGMA/METIS/SuiteSparse headers
and ships no fixtures), so these vectors come from the CPU oracle (oracle/, built-in C loops, single
thread) and are accepted only after agreeing with a dense LAPACK Cholesky of the same permuted matrix
to 1e-13.  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from util import small_cases, dense_reference_factor, panel_entries_from_dense, rel_err  # noqa: E402
import oracle  # noqa: E402

PICK = ("lap2d_8x8_nd", "lap3d_4_nd", "arrow_40_3", "blockdiag", "lap3d_8_nd_smallslot")


def main():
    oracle.blas_init("builtin")
    out = {}
    for name, n, Cp, Ci, Cx, perm, slot in small_cases():
        if name not in PICK:
            continue
        S = oracle.symbolic.analyze(n, Cp, Ci, Cx, perm, slot)
        Lsx, info, _ = oracle.chol_factorize(S)
        assert info == 0
        _, L = dense_reference_factor(S)
        mask = oracle.lower_mask(S)
        assert rel_err(Lsx, panel_entries_from_dense(S, L), mask) <= 1e-13
        out[name] = dict(
            n=n, Cp=np.asarray(Cp).tolist(), Ci=np.asarray(Ci).tolist(), Cx=np.asarray(Cx).tolist(),
            perm=None if perm is None else np.asarray(perm).tolist(), devSlotSize=slot,
            Super=S["Super"], Lsip=S["Lsip"], Lsxp=S["Lsxp"], Lsi=S["Lsi"], Perm=S["Perm"],
            LeafQueue=S["LeafQueue"], nsuper=S["nsuper"], nfsuper=S["nfsuper"], nstage=S["nstage"],
            Lsx=[float(v) if m else 0.0 for v, m in zip(Lsx.tolist(), mask.tolist())],
            mask=[int(m) for m in mask.tolist()])
    with open(os.path.join(HERE, "chol_small.json"), "w") as f:
        json.dump(out, f)
    print("wrote", {k: v["n"] for k, v in out.items()})


if __name__ == "__main__":
    main()

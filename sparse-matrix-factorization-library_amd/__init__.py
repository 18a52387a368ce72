"""MI355X-native numeric-factorization core behind SparseFrame's C entry points.

The product is ``libsparseframe_hip.so`` (C ABI declared in ``include/sparseframe_hip.h``); this
package is the thin Python host layer above it: a ctypes binding that mirrors the reference's
operator interface (analyze / factorize / solve / validate on a ``matrix_info`` object) plus the
synthetic-matrix generators used by the tests and by ``bench.py``.

Nothing here computes a factorization on the CPU: if the shared library or a HIP device is missing,
the numeric entry points raise.
"""
from ._lib import lib, LIB_PATH, SparseFrameError  # noqa: F401
from .api import (  # noqa: F401
    Symbolic, CholPlan, LUPlan, Schedule, Comm, MatrixInfo, LUMatrixInfo, CommonInfo, analyze, grid_nd_perm, graph_nd_perm, device_count, subtree_partition, ooc_partition, phases_for_rank, top_groups, validate_solution,
    REFERENCE_SLOT_1GPU, REFERENCE_SLOT_8GPU,
)
from . import gen  # noqa: F401
from .sharded import ShardedCholesky, ShardedFactorization  # noqa: F401

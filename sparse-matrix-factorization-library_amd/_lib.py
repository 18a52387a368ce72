"""ctypes loader for libsparseframe_hip.so.  Fails loudly when the library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsparseframe_hip.so")


class SparseFrameError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C sparse-matrix-factorization-library_amd/csrc`). There is no Python/CPU fallback.")

# One HIP runtime per process: PyTorch bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  If this
# library pulled in the system runtime first and torch then loaded its bundled copy, the second runtime would see
# no device.  So when torch is installed, its runtime is loaded first (without importing torch) and both share it.
def _preload_torch_hip_runtime():
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


_preload_torch_hip_runtime()
lib = C.CDLL(LIB_PATH)
LU_LIB_PATH = os.path.join(_HERE, "libsparseframe_lu_hip.so")
if not os.path.exists(LU_LIB_PATH):
    raise ImportError(f"{LU_LIB_PATH} is missing: run __graft_entry__.build()")
lu_lib = C.CDLL(LU_LIB_PATH)      # same entry-point names over the LU struct layout (as the reference's LU/Lib)

c_long_p = C.POINTER(C.c_int64)
c_double_p = C.POINTER(C.c_double)

# ---- flat ABI signatures (include/sparseframe_hip.h, layer 2) ----
lib.sf_version.restype = C.c_char_p
lib.sf_device_count.restype = C.c_int

lib.sf_symbolic_create.argtypes = [C.POINTER(C.c_void_p), C.c_int64, c_long_p, c_long_p, c_double_p, c_long_p, C.c_size_t]
lib.sf_symbolic_create.restype = C.c_int
lib.sf_symbolic_create_lu.argtypes = [C.POINTER(C.c_void_p), C.c_int64, c_long_p, c_long_p, c_double_p, c_long_p, C.c_size_t, C.c_int]
lib.sf_symbolic_create_lu.restype = C.c_int
lib.sf_symbolic_destroy.argtypes = [C.c_void_p]
lib.sf_symbolic_destroy.restype = None
lib.sf_symbolic_scalar.argtypes = [C.c_void_p, C.c_char_p]
lib.sf_symbolic_scalar.restype = C.c_int64
lib.sf_symbolic_long_array.argtypes = [C.c_void_p, C.c_char_p, c_long_p]
lib.sf_symbolic_long_array.restype = c_long_p
lib.sf_symbolic_float_array.argtypes = [C.c_void_p, C.c_char_p, c_long_p]
lib.sf_symbolic_float_array.restype = c_double_p
lib.sf_symbolic_flops.argtypes = [C.c_void_p, C.c_int]
lib.sf_symbolic_flops.restype = C.c_double
lib.sf_device_memory.argtypes = [C.c_int]
lib.sf_device_memory.restype = C.c_size_t
lib.sf_reference_slot_size.argtypes = [C.c_int, C.c_size_t]
lib.sf_reference_slot_size.restype = C.c_size_t
lib.sf_graph_nd_perm.argtypes = [C.c_int64, c_long_p, c_long_p, C.c_int64, c_long_p]
lib.sf_graph_nd_perm.restype = C.c_int
lib.sf_grid_nd_perm.argtypes = [C.c_int64] * 5 + [c_long_p]
lib.sf_grid_nd_perm.restype = C.c_int

lib.sf_chol_plan_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int64] + [c_long_p] * 7
lib.sf_chol_plan_create.restype = C.c_int
lib.sf_chol_plan_set_values.argtypes = [C.c_void_p, c_double_p]
lib.sf_chol_plan_set_values.restype = C.c_int
lib.sf_chol_plan_factorize.argtypes = [C.c_void_p, C.c_int]
lib.sf_chol_plan_factorize.restype = C.c_int
lib.sf_chol_plan_sync.argtypes = [C.c_void_p]
lib.sf_chol_plan_sync.restype = C.c_int
lib.sf_chol_plan_get_factor.argtypes = [C.c_void_p, c_double_p]
lib.sf_chol_plan_get_factor.restype = C.c_int
lib.sf_chol_plan_factorize_to_host.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p]
lib.sf_chol_plan_factorize_to_host.restype = C.c_int
lib.sf_chol_plan_factor_device_ptr.argtypes = [C.c_void_p]
lib.sf_chol_plan_factor_device_ptr.restype = C.c_void_p
lib.sf_chol_plan_solve.argtypes = [C.c_void_p, c_double_p, c_double_p]
lib.sf_chol_plan_solve.restype = C.c_int
lib.sf_chol_plan_solve_distributed.argtypes = [C.c_void_p, C.c_void_p, c_double_p, c_double_p]
lib.sf_chol_plan_solve_distributed.restype = C.c_int
lib.sf_handlers_replica_mismatches.argtypes = [C.c_void_p]
lib.sf_handlers_replica_mismatches.restype = C.c_int64
lib.sf_handlers_set_lu_pivoting.argtypes = [C.c_double, C.c_double]
lib.sf_handlers_set_lu_pivoting.restype = C.c_int
lib.sf_handlers_perturbed_pivots.argtypes = [C.c_void_p]
lib.sf_handlers_perturbed_pivots.restype = C.c_int64
lu_lib.SparseFrame_set_pivoting.argtypes = [C.c_double, C.c_double]
lu_lib.SparseFrame_set_pivoting.restype = C.c_int
lu_lib.SparseFrame_set_matrix_pivoting.argtypes = [C.c_void_p, C.c_double, C.c_double]
lu_lib.SparseFrame_set_matrix_pivoting.restype = C.c_int
lu_lib.SparseFrame_clear_matrix_pivoting.argtypes = [C.c_void_p]
lu_lib.SparseFrame_clear_matrix_pivoting.restype = C.c_int
lu_lib.SparseFrame_perturbed_pivots.argtypes = [C.c_void_p]
lu_lib.SparseFrame_perturbed_pivots.restype = C.c_int64
lib.sf_build_experiments.argtypes = []
lib.sf_build_experiments.restype = C.c_int
lib.sf_test_inject_failure.argtypes = [C.c_int, C.c_int]
lib.sf_test_inject_failure.restype = None
lib.sf_handlers_set_resident_solve.argtypes = [C.c_int]
lib.sf_handlers_set_resident_solve.restype = C.c_int
lib.sf_handlers_pool_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
lib.sf_handlers_pool_info.restype = C.c_int
lib.sf_handlers_resident_solves.argtypes = []
lib.sf_handlers_resident_solves.restype = C.c_int64
lib.sf_handlers_fingerprint_fallbacks.argtypes = []
lib.sf_handlers_fingerprint_fallbacks.restype = C.c_int64
lib.sf_handlers_plan_builds.argtypes = [C.c_void_p, C.c_int]
lib.sf_handlers_plan_builds.restype = C.c_int64
lib.sf_chol_plan_validate.argtypes = [C.c_void_p, c_double_p, c_double_p]
lib.sf_chol_plan_validate.restype = C.c_int
lib.sf_chol_plan_stat.argtypes = [C.c_void_p, C.c_char_p]
lib.sf_chol_plan_stat.restype = C.c_double
lib.sf_chol_plan_set_profiling.argtypes = [C.c_void_p, C.c_int]
lib.sf_chol_plan_set_profiling.restype = C.c_int
lib.sf_chol_plan_destroy.argtypes = [C.c_void_p]
lib.sf_chol_plan_destroy.restype = C.c_int

lib.sf_subtree_partition.argtypes = [C.c_int64, c_long_p, c_long_p, c_long_p, c_long_p, C.c_int, C.POINTER(C.c_int32),
                                     c_double_p, c_double_p]
lib.sf_subtree_partition.restype = C.c_int
lib.sf_ooc_partition.argtypes = [C.c_int64, c_long_p, c_long_p, c_long_p, c_long_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int),
                                 C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
lib.sf_ooc_partition.restype = C.c_int
lib.sf_chol_plan_create_ooc.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int64] + [c_long_p] * 7 + [C.POINTER(C.c_int32), C.c_int, C.c_int]
lib.sf_chol_plan_create_ooc.restype = C.c_int
lib.sf_chol_plan_schedule_ooc.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int64] + [c_long_p] * 7 + [C.POINTER(C.c_int32), C.c_int, C.c_int]
lib.sf_chol_plan_schedule_ooc.restype = C.c_int
lib.sf_lu_plan_create_ooc.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int64] + [c_long_p] * 9 + [C.POINTER(C.c_int32), C.c_int, C.c_int]
lib.sf_lu_plan_create_ooc.restype = C.c_int
lib.sf_chol_plan_create_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int64] + [c_long_p] * 7 + [C.POINTER(C.c_int32), C.c_int]
lib.sf_chol_plan_create_sharded.restype = C.c_int
lib.sf_chol_plan_factorize_phase.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.sf_chol_plan_factorize_phase.restype = C.c_int
lib.sf_chol_plan_top_region.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), c_long_p]
lib.sf_chol_plan_top_region.restype = C.c_int
lib.sf_subtree_partition_weighted.argtypes = [C.c_int64, c_long_p, c_long_p, c_long_p, c_long_p, C.c_int, C.c_double,
                                              C.POINTER(C.c_int32), c_double_p, c_double_p]
lib.sf_subtree_partition_weighted.restype = C.c_int
lib.sf_chol_plan_create_distributed.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int64] + [c_long_p] * 7 + \
    [C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int]
lib.sf_chol_plan_create_distributed.restype = C.c_int
lib.sf_chol_plan_create_mapped.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int64] + [c_long_p] * 7 + \
    [C.POINTER(C.c_int32), C.c_int, C.c_int]
lib.sf_chol_plan_create_mapped.restype = C.c_int
lib.sf_lu_plan_create_mapped.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int64] + [c_long_p] * 9 + \
    [C.POINTER(C.c_int32), C.c_int, C.c_int]
lib.sf_lu_plan_create_mapped.restype = C.c_int
lib.sf_chol_plan_schedule_mapped.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int64] + [c_long_p] * 7 + \
    [C.POINTER(C.c_int32), C.c_int, C.c_int]
lib.sf_chol_plan_schedule_mapped.restype = C.c_int
lib.sf_lu_plan_schedule_mapped.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int64] + [c_long_p] * 9 + \
    [C.POINTER(C.c_int32), C.c_int, C.c_int]
lib.sf_lu_plan_schedule_mapped.restype = C.c_int
lib.sf_chol_plan_num_launches.argtypes = [C.c_void_p]
lib.sf_chol_plan_num_launches.restype = C.c_int64
lib.sf_chol_plan_launch_info.argtypes = [C.c_void_p, C.c_int64, c_long_p]
lib.sf_chol_plan_launch_info.restype = C.c_int
lib.sf_chol_plan_segment_info.argtypes = [C.c_void_p, C.c_int64, c_long_p]
lib.sf_chol_plan_segment_info.restype = C.c_int
lib.sf_chol_plan_segment_owner.argtypes = [C.c_void_p, C.c_int64, c_long_p]
lib.sf_chol_plan_segment_owner.restype = C.c_int
lib.sf_chol_plan_panel_offsets.argtypes = [C.c_void_p, c_long_p]
lib.sf_chol_plan_panel_offsets.restype = C.c_int
lib.sf_chol_plan_num_solve_reduces.argtypes = [C.c_void_p]
lib.sf_chol_plan_num_solve_reduces.restype = C.c_int64
lib.sf_chol_plan_solve_reduce_info.argtypes = [C.c_void_p, C.c_int64, c_long_p]
lib.sf_chol_plan_solve_reduce_info.restype = C.c_int
lib.sf_chol_plan_segment_group.argtypes = [C.c_void_p, C.c_int64]
lib.sf_chol_plan_segment_group.restype = C.c_uint32
lib.sf_chol_plan_num_segments.argtypes = [C.c_void_p]
lib.sf_chol_plan_num_segments.restype = C.c_int64
lib.sf_chol_plan_segment_regions.argtypes = [C.c_void_p, C.c_int64, C.c_int64, c_long_p, c_long_p, c_long_p]
lib.sf_chol_plan_segment_regions.restype = C.c_int
lib.sf_chol_plan_segment_pack.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), c_long_p]
lib.sf_chol_plan_segment_pack.restype = C.c_int
lib.sf_chol_plan_factorize_segment.argtypes = [C.c_void_p, C.c_int64, C.c_int]
lib.sf_chol_plan_factorize_segment.restype = C.c_int
lib.sf_chol_plan_set_stream.argtypes = [C.c_void_p, C.c_void_p]
lib.sf_chol_plan_set_stream.restype = C.c_int
lib.sf_comm_unique_id.argtypes = [C.c_char_p]
lib.sf_comm_unique_id.restype = C.c_int
lib.sf_comm_create_rccl.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_char_p]
lib.sf_comm_create_rccl.restype = C.c_int
lib.sf_comm_allreduce_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
lib.sf_comm_allreduce_sum.restype = C.c_int
lib.sf_comm_selftest_split.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
lib.sf_comm_selftest_split.restype = C.c_int
lib.sf_comm_destroy.argtypes = [C.c_void_p]
lib.sf_comm_destroy.restype = C.c_int
lib.sf_comm_rank.argtypes = [C.c_void_p]
lib.sf_comm_size.argtypes = [C.c_void_p]
lib.sf_chol_plan_prepare_comm.argtypes = [C.c_void_p, C.c_void_p]
lib.sf_chol_plan_factorize_distributed.argtypes = [C.c_void_p, C.c_void_p, c_double_p, C.c_int]
lib.sf_chol_plan_factorize_distributed.restype = C.c_int
lib.sf_lu_plan_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int64] + [c_long_p] * 9
lib.sf_lu_plan_create.restype = C.c_int
lib.sf_lu_plan_create_distributed.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int64] + [c_long_p] * 9 + \
    [C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int]
lib.sf_lu_plan_create_distributed.restype = C.c_int
lib.sf_lu_plan_set_values.argtypes = [C.c_void_p, c_double_p, c_double_p]
lib.sf_lu_plan_set_values.restype = C.c_int
lib.sf_lu_plan_set_pivoting.argtypes = [C.c_void_p, C.c_double, C.c_double]
lib.sf_lu_plan_set_pivoting.restype = C.c_int
lib.sf_lu_plan_get_pivots.argtypes = [C.c_void_p, c_long_p]
lib.sf_lu_plan_get_pivots.restype = C.c_int
lib.sf_lu_plan_factorize.argtypes = [C.c_void_p, C.c_int]
lib.sf_lu_plan_factorize.restype = C.c_int
lib.sf_lu_plan_sync.argtypes = [C.c_void_p]
lib.sf_lu_plan_sync.restype = C.c_int
lib.sf_lu_plan_get_factor.argtypes = [C.c_void_p, c_double_p]
lib.sf_lu_plan_get_factor.restype = C.c_int
lib.sf_lu_plan_solve.argtypes = [C.c_void_p, c_double_p, c_double_p]
lib.sf_lu_plan_solve.restype = C.c_int
lib.sf_lu_plan_stat.argtypes = [C.c_void_p, C.c_char_p]
lib.sf_lu_plan_stat.restype = C.c_double
lib.sf_lu_plan_set_profiling.argtypes = [C.c_void_p, C.c_int]
lib.sf_lu_plan_set_profiling.restype = C.c_int
lib.sf_lu_plan_destroy.argtypes = [C.c_void_p]
lib.sf_lu_plan_destroy.restype = C.c_int

ERR_NAMES = {0: "SF_OK", 1: "SF_ERR_ARG", 2: "SF_ERR_NO_DEVICE", 3: "SF_ERR_ALLOC",
             4: "SF_ERR_NOT_POSDEF", 5: "SF_ERR_HIP", 6: "SF_ERR_PEER"}


def check(rc, what):
    if rc != 0:
        raise SparseFrameError(f"{what} failed: {ERR_NAMES.get(rc, rc)}")

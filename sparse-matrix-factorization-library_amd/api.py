"""Python mirror of the reference's operator interface for the numeric-factorization path.

Two views of the same C ABI (include/sparseframe_hip.h):

* ``MatrixInfo`` / ``CommonInfo`` wrap ``struct matrix_info_struct`` / ``struct common_info_struct``
  (reference Cholesky/Include/info.h:12-29, :70-150) and call the struct-based entry points with the
  reference's own names: ``analyze`` -> SparseFrame_analyze (SparseFrame.c:1916), ``factorize`` ->
  SparseFrame_factorize (:3019), ``solve`` -> SparseFrame_solve_supernodal (:3036), ``validate`` ->
  SparseFrame_validate (:3141), ``cleanup`` -> SparseFrame_cleanup_matrix (:3268).
* ``Symbolic`` / ``CholPlan`` wrap the flat plan ABI (sf_symbolic_*, sf_chol_plan_*), which keeps the
  factor resident in HBM between calls -- this is what bench.py times.
"""
import ctypes as C

import numpy as np

from ._lib import lib, lu_lib, check, c_long_p, c_double_p, SparseFrameError

# devSlotSize the reference computes (SparseFrame.c:82-87,199) for 288 GiB devices
REFERENCE_SLOT_1GPU = 8_694_792_192     # 1 device  -> numSplit 4
REFERENCE_SLOT_8GPU = 34_781_265_920    # >=4 devices -> numSplit 1


def _lp(a):
    return a.ctypes.data_as(c_long_p)


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    return int(lib.sf_device_count())


def grid_nd_perm(nx, ny=1, nz=1, leaf=3, sep_width=1):
    """Deterministic geometric nested dissection of a regular grid; perm[new] = old."""
    perm = np.empty(nx * ny * nz, dtype=np.int64)
    check(lib.sf_grid_nd_perm(nx, ny, nz, leaf, sep_width, _lp(perm)), "sf_grid_nd_perm")
    return perm


def graph_nd_perm(n, Cp, Ci, leaf=64):
    """Built-in nested dissection (BFS level-structure separators) for a general symmetric pattern; perm[new] = old."""
    Cp, Ci = _i64(Cp), _i64(Ci)
    perm = np.empty(max(n, 1), dtype=np.int64)
    check(lib.sf_graph_nd_perm(n, _lp(Cp), _lp(Ci), leaf, _lp(perm)), "sf_graph_nd_perm")
    return perm[:n]


class Symbolic:
    """Result of the host-side symbolic analysis (flat ABI)."""

    LONG_ARRAYS = ("Perm", "Parent", "Parent0", "Post", "ColCount", "ColCount0", "Lp", "Li", "LTp", "LTi",
                   "Up", "Ui", "UTp", "UTi", "Super", "SuperMap", "Sparent", "Lsip", "Lsxp", "Lsi", "LeafQueue",
                   "ST_Map", "ST_Pointer", "ST_Index", "Aoffset", "Moffset")
    SCALARS = ("n", "nnz", "nfsuper", "nsuper", "nstage", "isize", "xsize", "csize", "nsleaf", "lu", "symmetric", "unz")

    def __init__(self, n, Cp, Ci, Cx, perm=None, dev_slot_size=REFERENCE_SLOT_1GPU, method="cholesky", symmetric=True):
        """method 'cholesky' (input = one triangle) or 'lu' (symmetric=True: one triangle, else the whole matrix)"""
        Cp, Ci, Cx = _i64(Cp), _i64(Ci), _f64(Cx)
        if perm is not None:
            perm = _i64(perm)
        h = C.c_void_p()
        if method == "cholesky":
            check(lib.sf_symbolic_create(C.byref(h), n, _lp(Cp), _lp(Ci), _dp(Cx),
                                         _lp(perm) if perm is not None else None, dev_slot_size),
                  "sf_symbolic_create")
        elif method == "lu":
            check(lib.sf_symbolic_create_lu(C.byref(h), n, _lp(Cp), _lp(Ci), _dp(Cx),
                                            _lp(perm) if perm is not None else None, dev_slot_size,
                                            1 if symmetric else 0), "sf_symbolic_create_lu")
        else:
            raise ValueError(method)
        self.method = method
        self._h = h
        self._cache = {}
        self.dev_slot_size = dev_slot_size

    def close(self):
        if getattr(self, "_h", None):
            lib.sf_symbolic_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        if name in Symbolic.SCALARS:
            return int(lib.sf_symbolic_scalar(self._h, name.encode()))
        if name in Symbolic.LONG_ARRAYS:
            if name not in self._cache:
                ln = C.c_int64()
                p = lib.sf_symbolic_long_array(self._h, name.encode(), C.byref(ln))
                self._cache[name] = (np.ctypeslib.as_array(p, shape=(ln.value,)).copy()
                                     if ln.value else np.zeros(0, np.int64))
            return self._cache[name]
        if name in ("Lx", "LTx", "Ux", "UTx"):
            if name not in self._cache:
                ln = C.c_int64()
                p = lib.sf_symbolic_float_array(self._h, name.encode(), C.byref(ln))
                self._cache[name] = (np.ctypeslib.as_array(p, shape=(ln.value,)).copy()
                                     if ln.value else np.zeros(0))
            return self._cache[name]
        raise AttributeError(name)

    @property
    def flops_struct(self):
        return float(lib.sf_symbolic_flops(self._h, 0))

    @property
    def flops_exec(self):
        return float(lib.sf_symbolic_flops(self._h, 1))

    @property
    def flops_update(self):
        return float(lib.sf_symbolic_flops(self._h, 2))

    @property
    def scatter_elems(self):
        return float(lib.sf_symbolic_flops(self._h, 3))


def analyze(n, Cp, Ci, Cx, perm=None, dev_slot_size=REFERENCE_SLOT_1GPU, method="cholesky", symmetric=True):
    return Symbolic(n, Cp, Ci, Cx, perm, dev_slot_size, method, symmetric)


def validate_solution(sym, x, b=None):
    """the reference's SparseFrame_validate residual (SparseFrame.c:3182-3263) for a given solution x of the permuted
    system: b_i = 1 + i/n, r = A x - b with the stored triangle used symmetrically,
    |r|_inf / (|A|_1 |x|_inf + |b|_inf).  Vectorised numpy (host check of a device solve)."""
    n = sym.n
    if b is None:
        b = 1 + np.arange(n) / n
    Lp, Li, Lx = sym.Lp, sym.Li, sym.Lx
    cols = np.repeat(np.arange(n), np.diff(Lp))
    # (np.bincount with weights: the same sums as np.add.at, an order of magnitude faster at n = 16.8 M)
    r = -np.asarray(b, dtype=np.float64).copy()
    r += np.bincount(Li, weights=Lx * x[cols], minlength=n)
    off = Li != cols
    r += np.bincount(cols[off], weights=Lx[off] * x[Li[off]], minlength=n)
    ax = np.abs(Lx)
    colsum = np.bincount(cols, weights=ax, minlength=n) + np.bincount(Li[off], weights=ax[off], minlength=n)
    return float(np.abs(r).max() / (colsum.max() * np.abs(x).max() + np.abs(b).max()))


def subtree_partition(sym, nranks, top_weight=1.0):
    """owner[s] = rank of the elimination-tree subtree holding supernode s, -1 for the top supernodes.
    top_weight: cost of a top flop relative to a subtree flop (1 = replicated top, ~1/nranks = distributed top).
    Returns (owner int32[nsuper], top flop fraction, heaviest rank's subtree flop fraction)."""
    owner = np.empty(max(sym.nsuper, 1), dtype=np.int32)
    tf, ml = C.c_double(), C.c_double()
    check(lib.sf_subtree_partition_weighted(sym.nsuper, _lp(sym.Super), _lp(sym.SuperMap), _lp(sym.Lsip), _lp(sym.Lsi),
                                            nranks, float(top_weight), owner.ctypes.data_as(C.POINTER(C.c_int32)),
                                            C.byref(tf), C.byref(ml)),
          "sf_subtree_partition_weighted")
    return owner[:sym.nsuper], tf.value, ml.value


class OocCut(tuple):
    """(group, ngroups, largest group, top entries, need, fits) of ooc_partition, + .top_mode (0: top panels resident throughout,
    1: only while active)"""
    top_mode = 0


def ooc_partition(sym, budget_entries):
    """Out-of-core grouping (sf_ooc_partition): group[s] = streamed group of supernode s or -1 (top) for a budget of
    `budget_entries` resident panel entries.  Returns (group int32[nsuper], ngroups, entries of the largest group, top entries,
    entries the cut needs, fits) with the attribute .top_mode to pass on to the plan."""
    group = np.zeros(max(sym.nsuper, 1), dtype=np.int32)
    ng, mode = C.c_int(), C.c_int()
    ge, te, nd = C.c_int64(), C.c_int64(), C.c_int64()
    rc = lib.sf_ooc_partition(sym.nsuper, _lp(sym.Super), _lp(sym.SuperMap), _lp(sym.Lsip), _lp(sym.Lsi), int(budget_entries),
                              group.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ng), C.byref(ge), C.byref(te), C.byref(nd), C.byref(mode))
    if rc not in (0, 3):          # 3 = SF_ERR_ALLOC: no cut fits, the cheapest one is returned
        check(rc, "sf_ooc_partition")
    out = OocCut((group[:sym.nsuper], ng.value, ge.value, te.value, nd.value, rc == 0))
    out.top_mode = mode.value
    return out


def top_groups(sym, owner):
    """mask[s] for the top supernodes (owner[s] < 0): bit r set = rank r owns a subtree below s -- the group of ranks that
    holds s under the proportional mapping (what sf_chol_plan_create_mapped derives internally); 0 for subtree supernodes"""
    ns = sym.nsuper
    mask = np.zeros(ns, dtype=np.uint32)
    own = np.asarray(owner)
    mask[own >= 0] = np.left_shift(np.uint32(1), own[own >= 0].astype(np.uint32))
    Super, Lsip, Lsi, SuperMap = sym.Super, sym.Lsip, sym.Lsi, sym.SuperMap
    for s in range(ns):
        nscol, nsrow = Super[s + 1] - Super[s], Lsip[s + 1] - Lsip[s]
        if nscol < nsrow:
            mask[SuperMap[Lsi[Lsip[s] + nscol]]] |= mask[s]
    mask[own >= 0] = 0
    return mask


def phases_for_rank(owner, rank):
    """phase array of sf_chol_plan_create_sharded for `rank`: 0 own subtree, 1 top (replicated), -1 elsewhere"""
    return np.where(owner == rank, 0, np.where(owner < 0, 1, -1)).astype(np.int32)


class Comm:
    """one rank's RCCL communicator, created and used inside the C library (sf_comm_*): the unique id comes from rank 0
    (`Comm.unique_id()`) and reaches the other ranks by whatever the launcher offers (torch.distributed here)"""

    def __init__(self, device, rank, nranks, uid):
        h = C.c_void_p()
        check(lib.sf_comm_create_rccl(C.byref(h), device, rank, nranks, uid), "sf_comm_create_rccl")
        self._h, self.rank, self.nranks = h, rank, nranks

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(lib.sf_comm_unique_id(buf), "sf_comm_unique_id")
        return buf.raw

    def allreduce_sum(self, device_ptr, count, stream=0):
        check(lib.sf_comm_allreduce_sum(self._h, C.c_void_p(device_ptr), count, C.c_void_p(stream)), "sf_comm_allreduce_sum")

    def close(self):
        if getattr(self, "_h", None):
            lib.sf_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class _ShardedPlanMixin:
    """multi-GPU entry points shared by CholPlan and LUPlan (one handle type in the C ABI)"""

    def factorize_distributed(self, comm, host_out=None, sync=True):
        """the whole sharded factorization from this rank's side, driven in C (sf_chol_plan_factorize_distributed): own
        subtrees, then per segment pack -> RCCL all-reduce -> chain + this rank's share of the split GEMMs"""
        out = _dp(host_out) if host_out is not None else None
        check(lib.sf_chol_plan_factorize_distributed(self._h, comm._h, out, 1 if sync else 0), "sf_chol_plan_factorize_distributed")

    def prepare_comm(self, comm):
        """collective: sub-communicators of the plan's groups + one checked 8-byte sum per communicator (sf_chol_plan_prepare_comm)"""
        check(lib.sf_chol_plan_prepare_comm(self._h, comm._h), "sf_chol_plan_prepare_comm")

    def solve_distributed(self, comm, b):
        """the solve with the factor left distributed (sf_chol_plan_solve_distributed): returns x with this rank's entries filled
        in (its subtrees' columns, the shared supernodes it leads) and zeros elsewhere -- the sum over the ranks is the solution"""
        b = _f64(b)
        x = np.zeros_like(b)
        check(lib.sf_chol_plan_solve_distributed(self._h, comm._h, _dp(b), _dp(x)), "sf_chol_plan_solve_distributed")
        return x

    def factorize_phase(self, which, sync=True):
        check(lib.sf_chol_plan_factorize_phase(self._h, which, 1 if sync else 0), "sf_chol_plan_factorize_phase")

    def num_segments(self):
        return int(lib.sf_chol_plan_num_segments(self._h))

    def segment_group(self, k):
        """bit mask of the ranks that sum segment k's block columns"""
        return int(lib.sf_chol_plan_segment_group(self._h, k))

    def segment_regions(self, k):
        """[(offset, count)] in doubles relative to factor_device_ptr: regions to sum over the ranks before segment k"""
        nr = C.c_int64()
        check(lib.sf_chol_plan_segment_regions(self._h, k, 0, C.byref(nr), None, None), "sf_chol_plan_segment_regions")
        off = np.zeros(max(nr.value, 1), dtype=np.int64)
        cnt = np.zeros(max(nr.value, 1), dtype=np.int64)
        check(lib.sf_chol_plan_segment_regions(self._h, k, nr.value, C.byref(nr), _lp(off), _lp(cnt)),
              "sf_chol_plan_segment_regions")
        return [(int(off[i]), int(cnt[i])) for i in range(nr.value)]

    def segment_pack(self, k):
        """gather segment k's possibly non-zero block parts into the plan's contiguous scratch buffer (enqueued on the
        plan's stream); returns (device pointer, number of doubles) to all-reduce before factorize_segment(k)"""
        ptr, cnt = C.c_void_p(), C.c_int64()
        check(lib.sf_chol_plan_segment_pack(self._h, k, C.byref(ptr), C.byref(cnt)), "sf_chol_plan_segment_pack")
        return ptr.value or 0, cnt.value

    def factorize_segment(self, k, sync=False):
        check(lib.sf_chol_plan_factorize_segment(self._h, k, 1 if sync else 0), "sf_chol_plan_factorize_segment")

    def set_stream(self, stream_handle):
        """run on a caller-owned HIP stream (integer handle, e.g. torch.cuda.current_stream().cuda_stream)"""
        check(lib.sf_chol_plan_set_stream(self._h, C.c_void_p(stream_handle)), "sf_chol_plan_set_stream")

    def top_region(self):
        """(device pointer, number of doubles) of the contiguous top-panel region"""
        ptr, cnt = C.c_void_p(), C.c_int64()
        check(lib.sf_chol_plan_top_region(self._h, C.byref(ptr), C.byref(cnt)), "sf_chol_plan_top_region")
        return ptr.value or 0, cnt.value

    @property
    def factor_device_ptr(self):
        return lib.sf_chol_plan_factor_device_ptr(self._h)


class _ScheduleMixin:
    """schedule inspection (sf_chol_plan_launch_info & co.): works on real plans and on schedule-only ones"""

    LAUNCH_FIELDS = ("kind", "tasks", "items", "split", "lo", "hi", "replicated", "segment", "group_index", "group_size")
    SEGMENT_FIELDS = ("mask", "l0", "l1", "packed", "early", "regions")

    def _table(self, count, fn, width):
        out = np.zeros((max(count, 0), width), dtype=np.int64)
        row = np.zeros(width, dtype=np.int64)
        for k in range(count):
            check(fn(self._h, k, _lp(row)), fn.__name__)
            out[k] = row
        return out

    def launch_table(self):
        """int64[launches, 10]: LAUNCH_FIELDS per launch, in issue order"""
        return self._table(int(lib.sf_chol_plan_num_launches(self._h)), lib.sf_chol_plan_launch_info, 10)

    def segment_table(self):
        """int64[segments, 6]: SEGMENT_FIELDS per segment = per all-reduce of the factorization, in issue order"""
        return self._table(int(lib.sf_chol_plan_num_segments(self._h)), lib.sf_chol_plan_segment_info, 6)

    def segment_owner_table(self):
        """int64[segments, 2]: (owner's group index or -1, launches of the owner's part) -- the owner-computes prototype, SF_TOP_OWNER=1"""
        return self._table(int(lib.sf_chol_plan_num_segments(self._h)), lib.sf_chol_plan_segment_owner, 2)

    def solve_reduce_table(self):
        """int64[reduces, 3]: (group mask, first column, columns) of the forward sweep's sums, in issue order"""
        return self._table(int(lib.sf_chol_plan_num_solve_reduces(self._h)), lib.sf_chol_plan_solve_reduce_info, 3)

    def panel_offsets(self, nsuper):
        xp = np.zeros(max(nsuper, 1), dtype=np.int64)
        check(lib.sf_chol_plan_panel_offsets(self._h, _lp(xp)), "sf_chol_plan_panel_offsets")
        return xp[:nsuper]


class Schedule(_ScheduleMixin, _ShardedPlanMixin):
    """Rank `rank`'s plan of an nranks-way factorization WITHOUT a device (sf_chol_plan_schedule_mapped /
    sf_lu_plan_schedule_mapped): the launch list, segments, storage map and byte counts of the real plan, nothing allocated."""

    def __init__(self, sym, owner, rank, nranks, lu=False, ooc_group=None, ooc_ngroups=0, ooc_top_mode=0):
        """ooc_group / ooc_ngroups (owner None): the schedule of an out-of-core plan (sf_chol_plan_schedule_ooc)"""
        h = C.c_void_p()
        self._keep = [sym.Super, sym.SuperMap, sym.Lsip, sym.Lsi, sym.Lsxp, sym.Lp, sym.Li]
        if ooc_group is not None:
            grp = np.ascontiguousarray(ooc_group, dtype=np.int32)
            check(lib.sf_chol_plan_schedule_ooc(C.byref(h), sym.n, sym.nsuper, *[_lp(a) for a in self._keep],
                                                grp.ctypes.data_as(C.POINTER(C.c_int32)), int(ooc_ngroups), int(ooc_top_mode)), "sf_chol_plan_schedule_ooc")
            self._h, self.rank, self.nranks, self.nsuper = h, 0, 1, sym.nsuper
            return
        owner = np.ascontiguousarray(owner, dtype=np.int32)
        if lu:
            sym_in = bool(sym.symmetric)
            self._keep += [None, None] if sym_in else [sym.Up, sym.Ui]
            args = [_lp(a) if a is not None else None for a in self._keep]
            check(lib.sf_lu_plan_schedule_mapped(C.byref(h), sym.n, sym.nsuper, *args,
                                                 owner.ctypes.data_as(C.POINTER(C.c_int32)), int(rank), int(nranks)),
                  "sf_lu_plan_schedule_mapped")
        else:
            check(lib.sf_chol_plan_schedule_mapped(C.byref(h), sym.n, sym.nsuper, *[_lp(a) for a in self._keep],
                                                   owner.ctypes.data_as(C.POINTER(C.c_int32)), int(rank), int(nranks)),
                  "sf_chol_plan_schedule_mapped")
        self._h, self.rank, self.nranks, self.nsuper = h, rank, nranks, sym.nsuper

    def stat(self, name):
        return float(lib.sf_chol_plan_stat(self._h, name.encode()))

    def close(self):
        if getattr(self, "_h", None):
            lib.sf_chol_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class _ValidateMixin:
    def validate(self, return_x=False):
        """SparseFrame_validate on the device: b_i = 1 + i/n, solve, residual |Ax - b|_inf / (|A|_1 |x|_inf + |b|_inf)"""
        res = C.c_double()
        x = np.empty(max(self.n, 1), dtype=np.float64) if return_x else None
        check(lib.sf_chol_plan_validate(self._h, C.byref(res), _dp(x) if return_x else None), "sf_chol_plan_validate")
        return (res.value, x[:self.n]) if return_x else res.value


class CholPlan(_ShardedPlanMixin, _ValidateMixin, _ScheduleMixin):
    """Device-resident supernodal Cholesky (flat ABI).  Raises if no HIP device is present.
    phase/load_top: multi-GPU sharding (sf_chol_plan_create_sharded); default = the whole matrix on one device.
    rank/nranks (with phase): distributed top (sf_chol_plan_create_distributed), run with factorize_phase(0) and then
    factorize_segment(k) after summing segment_regions(k) over the ranks."""

    def __init__(self, sym, device=0, phase=None, load_top=True, rank=0, nranks=1, owner=None, ooc_group=None, ooc_ngroups=0, ooc_top_mode=0):
        """owner (with rank / nranks): the owner map of subtree_partition -> proportionally mapped plan
        (sf_chol_plan_create_mapped): own subtrees + the top supernodes above them, groups of ranks per top supernode.
        ooc_group / ooc_ngroups (from ooc_partition): out-of-core plan -- factorize_to_host only"""
        h = C.c_void_p()
        self._keep = [sym.Super, sym.SuperMap, sym.Lsip, sym.Lsi, sym.Lsxp, sym.Lp, sym.Li]
        if ooc_group is not None:
            grp = np.ascontiguousarray(ooc_group, dtype=np.int32)
            check(lib.sf_chol_plan_create_ooc(C.byref(h), device, sym.n, sym.nsuper, *[_lp(a) for a in self._keep],
                                              grp.ctypes.data_as(C.POINTER(C.c_int32)), int(ooc_ngroups), int(ooc_top_mode)), "sf_chol_plan_create_ooc")
        elif owner is not None:
            owner = np.ascontiguousarray(owner, dtype=np.int32)
            check(lib.sf_chol_plan_create_mapped(C.byref(h), device, sym.n, sym.nsuper, *[_lp(a) for a in self._keep],
                                                 owner.ctypes.data_as(C.POINTER(C.c_int32)), int(rank), int(nranks)),
                  "sf_chol_plan_create_mapped")
        elif phase is None:
            check(lib.sf_chol_plan_create(C.byref(h), device, sym.n, sym.nsuper, *[_lp(a) for a in self._keep]),
                  "sf_chol_plan_create")
        elif nranks <= 1:
            phase = np.ascontiguousarray(phase, dtype=np.int32)
            check(lib.sf_chol_plan_create_sharded(C.byref(h), device, sym.n, sym.nsuper, *[_lp(a) for a in self._keep],
                                                  phase.ctypes.data_as(C.POINTER(C.c_int32)), 1 if load_top else 0),
                  "sf_chol_plan_create_sharded")
        else:
            phase = np.ascontiguousarray(phase, dtype=np.int32)
            check(lib.sf_chol_plan_create_distributed(C.byref(h), device, sym.n, sym.nsuper, *[_lp(a) for a in self._keep],
                                                      phase.ctypes.data_as(C.POINTER(C.c_int32)), 1 if load_top else 0,
                                                      int(rank), int(nranks)),
                  "sf_chol_plan_create_distributed")
        self.device = device
        self._h = h
        self.xsize = sym.xsize
        self.n = sym.n

    def set_values(self, Lx):
        Lx = _f64(Lx)
        check(lib.sf_chol_plan_set_values(self._h, _dp(Lx)), "sf_chol_plan_set_values")

    def factorize(self, sync=True):
        check(lib.sf_chol_plan_factorize(self._h, 1 if sync else 0), "sf_chol_plan_factorize")

    def sync(self):
        check(lib.sf_chol_plan_sync(self._h), "sf_chol_plan_sync")

    def get_factor(self, out=None):
        """D2H in the reference layout; a sharded plan fills only the panels stored on this rank"""
        if out is None:
            out = np.zeros(max(self.xsize, 1), dtype=np.float64)
        check(lib.sf_chol_plan_get_factor(self._h, _dp(out)), "sf_chol_plan_get_factor")
        return out[:self.xsize]

    def factorize_to_host(self, Lx, out=None):
        """values H2D + factorize + factor D2H into `out` (pageable numpy memory), the download overlapped with the
        computation -- what one SparseFrame_factorize call does once its plan exists"""
        Lx = _f64(Lx)
        if out is None:
            out = np.empty(max(self.xsize, 1), dtype=np.float64)
        check(lib.sf_chol_plan_factorize_to_host(self._h, _dp(Lx), None, _dp(out)), "sf_chol_plan_factorize_to_host")
        return out[:self.xsize]

    def solve(self, b):
        b = _f64(b)
        x = np.empty_like(b)
        check(lib.sf_chol_plan_solve(self._h, _dp(b), _dp(x)), "sf_chol_plan_solve")
        return x

    def stat(self, name):
        return float(lib.sf_chol_plan_stat(self._h, name.encode()))

    def set_profiling(self, on=True):
        check(lib.sf_chol_plan_set_profiling(self._h, 1 if on else 0), "sf_chol_plan_set_profiling")

    def close(self):
        if getattr(self, "_h", None):
            lib.sf_chol_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class LUPlan(_ShardedPlanMixin, _ValidateMixin, _ScheduleMixin):
    """Device-resident supernodal no-pivot LU (flat ABI, sf_lu_plan_*).  `sym` comes from analyze(..., method='lu').
    phase/load_top/rank/nranks: distributed multi-GPU plan (sf_lu_plan_create_distributed), as CholPlan."""

    def __init__(self, sym, device=0, phase=None, load_top=True, rank=0, nranks=1, owner=None, ooc_group=None, ooc_ngroups=0, ooc_top_mode=0):
        if not sym.lu:
            raise ValueError("LUPlan needs an LU symbolic analysis (method='lu')")
        h = C.c_void_p()
        self._alias = bool(sym.symmetric)
        self._keep = [sym.Super, sym.SuperMap, sym.Lsip, sym.Lsi, sym.Lsxp, sym.Lp, sym.Li]
        if not self._alias:
            self._keep += [sym.Up, sym.Ui]
        args = [_lp(a) for a in self._keep] + ([None, None] if self._alias else [])
        if ooc_group is not None:
            grp = np.ascontiguousarray(ooc_group, dtype=np.int32)
            check(lib.sf_lu_plan_create_ooc(C.byref(h), device, sym.n, sym.nsuper, *args,
                                            grp.ctypes.data_as(C.POINTER(C.c_int32)), int(ooc_ngroups), int(ooc_top_mode)), "sf_lu_plan_create_ooc")
        elif owner is not None:
            owner = np.ascontiguousarray(owner, dtype=np.int32)
            check(lib.sf_lu_plan_create_mapped(C.byref(h), device, sym.n, sym.nsuper, *args,
                                               owner.ctypes.data_as(C.POINTER(C.c_int32)), int(rank), int(nranks)),
                  "sf_lu_plan_create_mapped")
        elif phase is None:
            check(lib.sf_lu_plan_create(C.byref(h), device, sym.n, sym.nsuper, *args), "sf_lu_plan_create")
        else:
            phase = np.ascontiguousarray(phase, dtype=np.int32)
            check(lib.sf_lu_plan_create_distributed(C.byref(h), device, sym.n, sym.nsuper, *args,
                                                    phase.ctypes.data_as(C.POINTER(C.c_int32)), 1 if load_top else 0,
                                                    int(rank), int(nranks)), "sf_lu_plan_create_distributed")
        self.device = device
        self._h = h
        self.xsize = sym.xsize
        self.n = sym.n

    def set_values(self, Lx, Ux=None):
        Lx = _f64(Lx)
        if self._alias:
            check(lib.sf_lu_plan_set_values(self._h, _dp(Lx), None), "sf_lu_plan_set_values")
        else:
            Ux = _f64(Ux)
            check(lib.sf_lu_plan_set_values(self._h, _dp(Lx), _dp(Ux)), "sf_lu_plan_set_values")

    def set_pivoting(self, tol=0.1, perturb=1.4901161193847656e-08):
        """threshold partial pivoting inside the 64 x 64 diagonal blocks (tol in [0, 1]; 0 = none, the reference's behaviour)
        and the perturbation of tiny pivots (relative to max|a_ij|; 0 = a zero pivot is an error).  A new plan has (0, 0): the
        reference never pivots (L:2653)."""
        check(lib.sf_lu_plan_set_pivoting(self._h, float(tol), float(perturb)), "sf_lu_plan_set_pivoting")

    def get_pivots(self):
        """pivpos[g] = row position of original row g after the in-block interchanges (identity where nothing moved)"""
        out = np.arange(max(self.n, 1), dtype=np.int64)
        check(lib.sf_lu_plan_get_pivots(self._h, _lp(out)), "sf_lu_plan_get_pivots")
        return out[:self.n]

    def factorize(self, sync=True):
        check(lib.sf_lu_plan_factorize(self._h, 1 if sync else 0), "sf_lu_plan_factorize")

    def sync(self):
        check(lib.sf_lu_plan_sync(self._h), "sf_lu_plan_sync")

    def get_factor(self, out=None):
        """D2H in the reference's packed layout; a sharded plan returns zeros for the panels of other ranks"""
        if out is None:
            out = np.empty(max(self.xsize, 1), dtype=np.float64)
        check(lib.sf_lu_plan_get_factor(self._h, _dp(out)), "sf_lu_plan_get_factor")
        return out[:self.xsize]

    def factorize_to_host(self, Lx, Ux=None, out=None):
        """as CholPlan.factorize_to_host; the factor arrives in the reference's packed LU layout"""
        Lx = _f64(Lx)
        if out is None:
            out = np.empty(max(self.xsize, 1), dtype=np.float64)
        ux = None if self._alias else _dp(_f64(Ux))
        check(lib.sf_chol_plan_factorize_to_host(self._h, _dp(Lx), ux, _dp(out)), "sf_chol_plan_factorize_to_host")
        return out[:self.xsize]

    def solve(self, b):
        b = _f64(b)
        x = np.empty_like(b)
        check(lib.sf_lu_plan_solve(self._h, _dp(b), _dp(x)), "sf_lu_plan_solve")
        return x

    def stat(self, name):
        return float(lib.sf_lu_plan_stat(self._h, name.encode()))

    def set_profiling(self, on=True):
        check(lib.sf_lu_plan_set_profiling(self._h, 1 if on else 0), "sf_lu_plan_set_profiling")

    def close(self):
        if getattr(self, "_h", None):
            lib.sf_lu_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


# -------------------------------------------------------------------------------------------------
# struct-based interface (reference names)
# -------------------------------------------------------------------------------------------------
class CommonInfoStruct(C.Structure):       # info.h:12-29
    _fields_ = [("numCPU", C.c_int), ("numGPU", C.c_int), ("numGPU_physical", C.c_int),
                ("minDevMemSize", C.c_size_t), ("minHostMemSize", C.c_size_t),
                ("matrixThreadNum", C.c_int), ("numSparseMatrix", C.c_int),
                ("devSlotSize", C.c_size_t),
                ("allocateTime", C.c_double), ("computeTime", C.c_double), ("freeTime", C.c_double)]


class MatrixInfoStruct(C.Structure):       # info.h:70-150
    _fields_ = [("serial", C.c_int), ("path", C.c_char_p), ("file", C.c_void_p),
                ("factorizeType", C.c_int), ("isSymmetric", C.c_int), ("isComplex", C.c_int),
                ("ncol", C.c_int64), ("nrow", C.c_int64), ("nzmax", C.c_int64),
                ("Tj", c_long_p), ("Ti", c_long_p), ("Tx", c_double_p),
                ("Cp", c_long_p), ("Ci", c_long_p), ("Cx", c_double_p),
                ("Lp", c_long_p), ("Li", c_long_p), ("Lx", c_double_p),
                ("LTp", c_long_p), ("LTi", c_long_p), ("LTx", c_double_p),
                ("permMethod", C.c_int),
                ("Perm", c_long_p), ("Parent", c_long_p), ("Post", c_long_p), ("ColCount", c_long_p),
                ("nsuper", C.c_int64), ("Super", c_long_p), ("SuperMap", c_long_p), ("Sparent", c_long_p),
                ("nsleaf", C.c_int64), ("LeafQueue", c_long_p),
                ("isize", C.c_int64), ("xsize", C.c_int64),
                ("Lsip", c_long_p), ("Lsxp", c_long_p), ("Lsi", c_long_p), ("Lsx", c_double_p),
                ("csize", C.c_int64), ("nstage", C.c_int64),
                ("ST_Map", c_long_p), ("ST_Pointer", c_long_p), ("ST_Index", c_long_p), ("ST_Parent", c_long_p),
                ("Aoffset", C.POINTER(C.c_size_t)), ("Moffset", C.POINTER(C.c_size_t)),
                ("workspace", C.c_void_p), ("workSize", C.c_size_t),
                ("Bx", c_double_p), ("Xx", c_double_p), ("Rx", c_double_p),
                ("residual", C.c_double),
                ("readTime", C.c_double), ("analyzeTime", C.c_double),
                ("factorizeTime", C.c_double), ("solveTime", C.c_double)]


for _name, _args in (("SparseFrame_allocate_gpu", [C.POINTER(CommonInfoStruct), C.POINTER(C.c_void_p)]),
                     ("SparseFrame_free_gpu", [C.POINTER(CommonInfoStruct), C.POINTER(C.c_void_p)]),
                     ("SparseFrame_initialize_matrix", [C.POINTER(MatrixInfoStruct)]),
                     ("SparseFrame_read_matrix", [C.POINTER(MatrixInfoStruct)]),
                     ("SparseFrame_set_matrix_csc", [C.POINTER(MatrixInfoStruct), C.c_int64, C.c_int64,
                                                     c_long_p, c_long_p, c_double_p, C.c_int]),
                     ("SparseFrame_set_perm", [C.POINTER(MatrixInfoStruct), c_long_p]),
                     ("SparseFrame_analyze", [C.POINTER(CommonInfoStruct), C.POINTER(MatrixInfoStruct)]),
                     ("SparseFrame_factorize", [C.POINTER(CommonInfoStruct), C.c_void_p, C.POINTER(MatrixInfoStruct)]),
                     ("SparseFrame_factorize_supernodal", [C.POINTER(CommonInfoStruct), C.c_void_p, C.POINTER(MatrixInfoStruct)]),
                     ("SparseFrame_solve_supernodal", [C.POINTER(MatrixInfoStruct)]),
                     ("SparseFrame_validate", [C.POINTER(MatrixInfoStruct)]),
                     ("SparseFrame_cleanup_matrix", [C.POINTER(MatrixInfoStruct)])):
    _f = getattr(lib, _name)
    _f.argtypes = _args
    _f.restype = C.c_int


class LUMatrixInfoStruct(C.Structure):     # LU/Include/info.h:70-163
    _fields_ = [("serial", C.c_int), ("path", C.c_char_p), ("file", C.c_void_p),
                ("factorizeType", C.c_int), ("isSymmetric", C.c_int), ("isComplex", C.c_int),
                ("ncol", C.c_int64), ("nrow", C.c_int64), ("nzmax", C.c_int64),
                ("Tj", c_long_p), ("Ti", c_long_p), ("Tx", c_double_p),
                ("Cp", c_long_p), ("Ci", c_long_p), ("Cx", c_double_p),
                ("nzCPCT", C.c_int64), ("CPCTp", c_long_p), ("CPCTi", c_long_p),
                ("Lp", c_long_p), ("Li", c_long_p), ("Lx", c_double_p),
                ("LTp", c_long_p), ("LTi", c_long_p), ("LTx", c_double_p),
                ("Up", c_long_p), ("Ui", c_long_p), ("Ux", c_double_p),
                ("UTp", c_long_p), ("UTi", c_long_p), ("UTx", c_double_p),
                ("permMethod", C.c_int), ("PivInv", c_long_p),
                ("Perm", c_long_p), ("Parent", c_long_p), ("Post", c_long_p), ("ColCount", c_long_p),
                ("nsuper", C.c_int64), ("Super", c_long_p), ("SuperMap", c_long_p), ("Sparent", c_long_p),
                ("nsleaf", C.c_int64), ("LeafQueue", c_long_p),
                ("isize", C.c_int64), ("xsize", C.c_int64),
                ("Lsip", c_long_p), ("Lsxp", c_long_p), ("Lsi", c_long_p), ("Lsx", c_double_p),
                ("csize", C.c_int64), ("nstage", C.c_int64),
                ("ST_Map", c_long_p), ("ST_Pointer", c_long_p), ("ST_Index", c_long_p), ("ST_Parent", c_long_p),
                ("Aoffset", C.POINTER(C.c_size_t)), ("Moffset", C.POINTER(C.c_size_t)),
                ("workspace", C.c_void_p), ("workSize", C.c_size_t),
                ("Bx", c_double_p), ("Xx", c_double_p), ("Rx", c_double_p),
                ("residual", C.c_double),
                ("readTime", C.c_double), ("analyzeTime", C.c_double),
                ("factorizeTime", C.c_double), ("solveTime", C.c_double)]


for _name, _args in (("SparseFrame_allocate_gpu", [C.POINTER(CommonInfoStruct), C.POINTER(C.c_void_p)]),
                     ("SparseFrame_free_gpu", [C.POINTER(CommonInfoStruct), C.POINTER(C.c_void_p)]),
                     ("SparseFrame_initialize_matrix", [C.POINTER(LUMatrixInfoStruct)]),
                     ("SparseFrame_read_matrix", [C.POINTER(LUMatrixInfoStruct)]),
                     ("SparseFrame_set_matrix_csc", [C.POINTER(LUMatrixInfoStruct), C.c_int64, C.c_int64,
                                                     c_long_p, c_long_p, c_double_p, C.c_int]),
                     ("SparseFrame_set_perm", [C.POINTER(LUMatrixInfoStruct), c_long_p]),
                     ("SparseFrame_analyze", [C.POINTER(CommonInfoStruct), C.POINTER(LUMatrixInfoStruct)]),
                     ("SparseFrame_factorize", [C.POINTER(CommonInfoStruct), C.c_void_p, C.POINTER(LUMatrixInfoStruct)]),
                     ("SparseFrame_factorize_supernodal", [C.POINTER(CommonInfoStruct), C.c_void_p, C.POINTER(LUMatrixInfoStruct)]),
                     ("SparseFrame_solve_supernodal", [C.POINTER(LUMatrixInfoStruct)]),
                     ("SparseFrame_validate", [C.POINTER(LUMatrixInfoStruct)]),
                     ("SparseFrame_cleanup_matrix", [C.POINTER(LUMatrixInfoStruct)])):
    _f = getattr(lu_lib, _name)
    _f.argtypes = _args
    _f.restype = C.c_int


class CommonInfo:
    """common_info_struct + the opaque gpu_info list (SparseFrame_allocate_gpu / _free_gpu)."""

    def __init__(self, dev_slot_size=None):
        self.c = CommonInfoStruct()
        self.gpu_list = C.c_void_p()
        check(lib.SparseFrame_allocate_gpu(C.byref(self.c), C.byref(self.gpu_list)), "SparseFrame_allocate_gpu")
        if dev_slot_size is not None:
            self.c.devSlotSize = dev_slot_size

    def plan_builds(self):
        """device plans the handlers have built so far (a repeated sparsity pattern must not add to it)"""
        return int(lib.sf_handlers_plan_builds(self.gpu_list, self.c.numGPU))

    def close(self):
        if self.gpu_list:
            lib.SparseFrame_free_gpu(C.byref(self.c), C.byref(self.gpu_list))
            self.gpu_list = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MatrixInfo:
    """matrix_info_struct driven through the reference's stage functions (Cholesky library)."""
    _lib = lib
    _struct = MatrixInfoStruct

    def __init__(self, serial=0):
        self.c = self._struct()
        self.c.serial = serial
        self._lib.SparseFrame_initialize_matrix(C.byref(self.c))

    def set_csc(self, n, Cp, Ci, Cx, symmetric=True):
        """symmetric=True: one triangle (Cholesky, or LU of a symmetric matrix); False: whole matrix (LU only)"""
        Cp, Ci, Cx = _i64(Cp), _i64(Ci), _f64(Cx)
        check(self._lib.SparseFrame_set_matrix_csc(C.byref(self.c), n, len(Ci), _lp(Cp), _lp(Ci), _dp(Cx),
                                             1 if symmetric else 0), "SparseFrame_set_matrix_csc")

    def read(self, path):
        self._path = str(path).encode()
        self.c.path = self._path
        check(self._lib.SparseFrame_read_matrix(C.byref(self.c)), "SparseFrame_read_matrix")

    def set_perm(self, perm):
        """perm[new] = old; None = natural order (explicit opt-in: the default is the built-in nested dissection)"""
        if perm is None:
            check(self._lib.SparseFrame_set_perm(C.byref(self.c), None), "SparseFrame_set_perm")
            return
        perm = _i64(perm)
        check(self._lib.SparseFrame_set_perm(C.byref(self.c), _lp(perm)), "SparseFrame_set_perm")

    def use_builtin_ordering(self):
        """permMethod = PERM_METIS with no Perm supplied (the default after initialize_matrix): SparseFrame_analyze orders
        with the built-in nested dissection"""
        self.c.permMethod = 2

    def analyze(self, common):
        check(self._lib.SparseFrame_analyze(C.byref(common.c), C.byref(self.c)), "SparseFrame_analyze")

    def factorize(self, common):
        check(self._lib.SparseFrame_factorize(C.byref(common.c), common.gpu_list, C.byref(self.c)), "SparseFrame_factorize")

    def validate(self):
        check(self._lib.SparseFrame_validate(C.byref(self.c)), "SparseFrame_validate")
        return float(self.c.residual)

    def cleanup(self):
        self._lib.SparseFrame_cleanup_matrix(C.byref(self.c))

    def array(self, name, length):
        p = getattr(self.c, name)
        if not p or length == 0:
            return np.zeros(0, dtype=np.float64 if name.endswith("x") else np.int64)
        return np.ctypeslib.as_array(p, shape=(length,))

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass


class LUMatrixInfo(MatrixInfo):
    """the same stage functions from the LU library (libsparseframe_lu_hip.so, LU struct layout)"""
    _lib = lu_lib
    _struct = LUMatrixInfoStruct

    @staticmethod
    def set_pivoting(tol=0.0, perturb=0.0):
        """process-wide policy of the LU struct entry points (SparseFrame_set_pivoting): (0, 0) = the reference's behaviour"""
        check(lu_lib.SparseFrame_set_pivoting(float(tol), float(perturb)), "SparseFrame_set_pivoting")

    def set_matrix_pivoting(self, tol=0.0, perturb=0.0):
        """this matrix's own policy (SparseFrame_set_matrix_pivoting); overrides the process-wide one for this matrix_info"""
        check(lu_lib.SparseFrame_set_matrix_pivoting(C.byref(self.c), float(tol), float(perturb)), "SparseFrame_set_matrix_pivoting")

    def perturbed_pivots(self):
        return int(lu_lib.SparseFrame_perturbed_pivots(C.byref(self.c)))

/*
 * sparseframe_flat.h -- the flat ("plan") C ABI of the MI355X-native numeric-factorization core, shared by the
 * Cholesky and the LU struct headers.  extern "C", plain pointers and sizes only; everything here is exported by
 * libsparseframe_hip.so.  Citations: C: = Cholesky/Source/SparseFrame.c, L: = LU/Source/SparseFrame.c of the reference.
 */
#ifndef SPARSEFRAME_FLAT_H
#define SPARSEFRAME_FLAT_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar types: reference arch.h:6-16 (Int=int, Long=long, Float=double) ---- */
typedef int64_t sf_long;   /* reference `Long`  (LP64 long)  */
typedef double  sf_float;  /* reference `Float`               */

/* reference type.h:4-5 */
enum FactorizeType { TYPE_CHOLESKY, TYPE_QR, TYPE_LU };
enum PermMethod { PERM_IDENTITY, PERM_AMD, PERM_METIS };

/* ---- reference I:12-29.  Field order and types are the ABI. ---- */
struct common_info_struct
{
    int numCPU;
    int numGPU;
    int numGPU_physical;
    size_t minDevMemSize;
    size_t minHostMemSize;
    int matrixThreadNum;
    int numSparseMatrix;
    size_t devSlotSize;      /* INPUT of symbolic analysis (C:1402): caps supernode/stage size */
    double allocateTime;
    double computeTime;
    double freeTime;
};

/* ---- reference I:31-68 embeds CUDA/cuBLAS/cuSOLVER handles and is only touched by
 * allocate_gpu / free_gpu / factorize.  Here it is opaque HIP state: one element per
 * device handler, allocated and freed by this library only. ---- */
struct gpu_info_struct;

/* =====================================================================================
 * Layer 2: flat ABI.
 * ===================================================================================== */

#define SF_OK                0
#define SF_ERR_ARG           1
#define SF_ERR_NO_DEVICE     2   /* no HIP device / HIP runtime error */
#define SF_ERR_ALLOC         3
#define SF_ERR_NOT_POSDEF    4   /* non-positive pivot met in a diagonal block */
#define SF_ERR_HIP           5
#define SF_ERR_PEER          6   /* multi-GPU: another rank of the group reported a failure; this rank stopped with it */

/* ---- host-side symbolic analysis on plain arrays (what SparseFrame_analyze forwards to).
 * The result object owns its arrays; read them through sf_symbolic_get. ---- */
typedef struct sf_symbolic sf_symbolic;

int sf_symbolic_create(sf_symbolic **out, sf_long n, const sf_long *Cp, const sf_long *Ci, const sf_float *Cx,
                       const sf_long *perm /* NULL = identity */, size_t devSlotSize);
/* LU variant (reference LU/Source/SparseFrame.c:1068-2231): elimination tree / counts / row structures of the
 * pattern of L + U^T, panels of (2*nsrow - nscol) x nscol values (L:1946).  is_symmetric != 0: Cp/Ci/Cx hold one
 * triangle of a symmetric matrix (then U aliases L, L:2718-2729); otherwise the whole matrix in CSC.
 * Extra arrays: Long "Up","Ui","UTp","UTi" (U by ROW and its transpose, L:1179-1282), double "Ux","UTx". */
int sf_symbolic_create_lu(sf_symbolic **out, sf_long n, const sf_long *Cp, const sf_long *Ci, const sf_float *Cx,
                          const sf_long *perm /* NULL = identity */, size_t devSlotSize, int is_symmetric);
void sf_symbolic_destroy(sf_symbolic *sym);
/* scalar outputs: "n","nnz","nfsuper","nsuper","nstage","isize","xsize","csize","nsleaf","lu","symmetric","unz" */
sf_long sf_symbolic_scalar(const sf_symbolic *sym, const char *name);
/* array outputs (borrowed pointers, valid until destroy):
 * Long arrays: "Perm","Parent","Post","ColCount","ColCount0","Lp","Li","LTp","LTi","Super","SuperMap","Sparent",
 *              "Lsip","Lsxp","Lsi","LeafQueue","ST_Map","ST_Pointer","ST_Index","Aoffset","Moffset"
 * "Post" and "ColCount0"/"Parent0" are the PRE-supernodal values (before C:1429-1445 renumbers them).
 * double arrays: "Lx","LTx" */
const sf_long *sf_symbolic_long_array(const sf_symbolic *sym, const char *name, sf_long *len);
const sf_float *sf_symbolic_float_array(const sf_symbolic *sym, const char *name, sf_long *len);
/* algorithmic flop counts: which = 0 -> F_struct = sum_j ColCount_j^2 (unrelaxed),
 * 1 -> F_exec (executed, relaxed supernodes), 2 -> executed SYRK/GEMM update flops only */
double sf_symbolic_flops(const sf_symbolic *sym, int which);

/* deterministic geometric nested dissection of an nx*ny*nz grid (node id = x + nx*(y + ny*z)):
 * recursive longest-axis bisection with sep_width-plane separators, leaf boxes of at most
 * leaf^3 nodes in natural order.  perm[new] = old.  Stands in for METIS (unpinned third party). */
int sf_grid_nd_perm(sf_long nx, sf_long ny, sf_long nz, sf_long leaf, sf_long sep_width, sf_long *perm);

/* Built-in fill-reducing ordering for a general symmetric pattern (any triangle or both in Cp/Ci): nested dissection
 * by BFS level-structure separators, pieces of at most `leaf` vertices in reverse BFS order.  perm[new] = old.
 * A self-contained stand-in for the reference's METIS_NodeND call (C:942); orderings are outside the parity contract. */
int sf_graph_nd_perm(sf_long n, const sf_long *Cp, const sf_long *Ci, sf_long leaf, sf_long *perm);

/* ---- device plan for supernodal Cholesky (replaces C:2150-3017 + CK:22-158) ---- */
typedef struct sf_chol_plan sf_chol_plan;

/* Uploads the symbolic structure to `device`, builds the level schedule and the grouped
 * task tables, allocates the device-resident factor (xsize doubles).  Depends on the
 * structure only: reusable for any number of numeric factorizations of the same pattern. */
int sf_chol_plan_create(sf_chol_plan **plan, int device,
                        sf_long n, sf_long nsuper,
                        const sf_long *Super, const sf_long *SuperMap,
                        const sf_long *Lsip, const sf_long *Lsi, const sf_long *Lsxp,
                        const sf_long *Lp, const sf_long *Li);
/* H2D copy of the CSC values of lower(P A P^T) (nnz = Lp[n] doubles). */
int sf_chol_plan_set_values(sf_chol_plan *plan, const sf_float *Lx);
/* The timed hot path: assemble (loadA) + factor every supernode + Schur updates with the
 * mapped scatter, entirely on the device.  Asynchronous on the plan's stream unless
 * `sync` != 0.  Returns SF_ERR_NOT_POSDEF (after sync) when a pivot <= 0 was met. */
int sf_chol_plan_factorize(sf_chol_plan *plan, int sync);
/* waits for the plan's stream and returns the factorization status */
int sf_chol_plan_sync(sf_chol_plan *plan);
/* D2H copy of the factor into the reference layout (xsize doubles). */
int sf_chol_plan_get_factor(sf_chol_plan *plan, sf_float *Lsx);
/* values H2D + numeric factorization + factor D2H into host_out (reference layout, xsize doubles), the download of
 * finished blocks overlapped with the computation of the levels above them (the reference's copy-back stream,
 * C:2888-2895).  host_out may be ordinary pageable memory (it is never pinned or registered).  Ux: LU plans only (NULL
 * when U aliases L).  "last_to_host_ms" reports the wall time of the call. */
int sf_chol_plan_factorize_to_host(sf_chol_plan *plan, const sf_float *Lx, const sf_float *Ux, sf_float *host_out);
/* device pointer of the resident factor (for device-side consumers) */
void *sf_chol_plan_factor_device_ptr(sf_chol_plan *plan);
/* device-side supernodal solve with the resident factor: x <- (L L^T)^{-1} b, permuted space */
int sf_chol_plan_solve(sf_chol_plan *plan, const sf_float *b_host, sf_float *x_host);
/* values [e_begin, e_end) of the factor in the reference layout (whole plans only) */
int sf_chol_plan_get_factor_range(sf_chol_plan *plan, sf_long e_begin, sf_long e_end, sf_float *out);
/* SparseFrame_validate on the device (C:3141-3266; LU plans: L:3702-3858): b_i = 1 + i/n, solve with the resident factor,
 * r = A x - b from the plan's copy of the matrix values, *residual = |r|_inf / (|A|_1 |x|_inf + |b|_inf).  x_host may be NULL. */
int sf_chol_plan_validate(sf_chol_plan *plan, sf_float *residual, sf_float *x_host);
/* statistics: "levels","launches","gemm_tasks","update_pairs","flops_exec","flops_update",
 * "scatter_elems","bytes_device","last_ms" (device time of the last factorize, HIP events),
 * "last_update_ms","last_panel_ms","last_load_ms" (only when profiling is on) */
double sf_chol_plan_stat(const sf_chol_plan *plan, const char *name);
/* 1 -> record HIP events around each phase of the next factorize calls */
int sf_chol_plan_set_profiling(sf_chol_plan *plan, int on);
int sf_chol_plan_destroy(sf_chol_plan *plan);

/* ---- multi-GPU sharding of ONE factorization by elimination-tree subtrees (the reference has no
 * inter-GPU path: its handlers exchange panels through host memory, C:2267, C:2421-2467).
 * sf_subtree_partition: owner[s] = rank whose subtree holds supernode s, or -1 for the replicated top supernodes.
 * Rank r then builds a plan with phase[s] = 0 (owner[s] == r), 1 (owner[s] == -1), -1 (otherwise) and runs
 *     sf_chol_plan_factorize_phase(plan, 0)   assemble + own subtrees; their updates of top panels accumulate locally
 *     all-reduce(sum) of sf_chol_plan_top_region over the ranks (RCCL)   -- the one exchange step
 *     sf_chol_plan_factorize_phase(plan, 1)   top supernodes, replicated on every rank
 * load_top != 0 on exactly ONE rank (it contributes the matrix entries of the top panels to the sum). ---- */
int sf_subtree_partition(sf_long nsuper, const sf_long *Super, const sf_long *SuperMap, const sf_long *Lsip,
                         const sf_long *Lsi, int nranks, int32_t *owner, double *top_fraction, double *max_load_fraction);
int sf_chol_plan_create_sharded(sf_chol_plan **plan, int device, sf_long n, sf_long nsuper,
                                const sf_long *Super, const sf_long *SuperMap,
                                const sf_long *Lsip, const sf_long *Lsi, const sf_long *Lsxp,
                                const sf_long *Lp, const sf_long *Li, const int32_t *phase, int load_top);
int sf_chol_plan_factorize_phase(sf_chol_plan *plan, int phase /* 0, 1, or -1 = both */, int sync);
/* device pointer and length (doubles) of the contiguous region holding every top panel */
int sf_chol_plan_top_region(sf_chol_plan *plan, void **device_ptr, sf_long *count);

/* ---- a factor LARGER than the device's memory (the reference streams slot-sized "stages" through the device and keeps the factor
 * on the host, C:1721-1846, C:2421-2467; here: whole subtrees).  sf_ooc_partition cuts the supernodal tree for a budget of
 * `budget_entries` resident panel entries (doubles; LU stores two per entry): group[s] = streamed group of supernode s (whole
 * subtrees, consecutive in the postorder) or -1 = top.  The plan keeps the top panels resident and streams the groups through two
 * alternating buffers: group g is factorized while group g - 1 is copied to the host (device need = top + 2 x largest group;
 * *need_entries).  *top_mode (pass it on to the plan): 0 = all top panels resident, factorized after the last group; 1 = when
 * that does not fit: a top panel only while it is ACTIVE (from the first group below it until its own factorization, right after
 * the last one) -- *top_entries is then the arena the active panels share; slower, reaches factors of about three times the budget.  SF_OK: fits (ngroups == 1: in core); SF_ERR_ALLOC: no cut fits, group[] holds the cheapest one.
 * An out-of-core plan runs through sf_chol_plan_factorize_to_host only (the factor exists on the host, never as a whole on the
 * device): it refuses solve / validate / get_factor (SF_ERR_ARG).  SparseFrame_factorize picks this path by itself when the
 * in-core plan does not fit (SF_DEVICE_BUDGET_MB lowers the budget for tests). ---- */
int sf_ooc_partition(sf_long nsuper, const sf_long *Super, const sf_long *SuperMap, const sf_long *Lsip, const sf_long *Lsi,
                     sf_long budget_entries, int32_t *group, int *ngroups, sf_long *group_entries, sf_long *top_entries,
                     sf_long *need_entries, int *top_mode);
int sf_chol_plan_create_ooc(sf_chol_plan **plan, int device, sf_long n, sf_long nsuper,
                            const sf_long *Super, const sf_long *SuperMap,
                            const sf_long *Lsip, const sf_long *Lsi, const sf_long *Lsxp,
                            const sf_long *Lp, const sf_long *Li, const int32_t *group, int ngroups, int top_mode);
/* the same plan without a device (launch list, storage map, byte counts only; see sf_chol_plan_schedule_mapped) */
int sf_chol_plan_schedule_ooc(sf_chol_plan **plan, sf_long n, sf_long nsuper,
                              const sf_long *Super, const sf_long *SuperMap,
                              const sf_long *Lsip, const sf_long *Lsi, const sf_long *Lsxp,
                              const sf_long *Lp, const sf_long *Li, const int32_t *group, int ngroups, int top_mode);

/* ---- distributed top: the top supernodes' large GEMMs are SPLIT over the ranks instead of
 * replicated.  Every top panel is summed over the ranks exactly once, one 512-column block at a time, right before
 * the block's sequential 64-column POTRF/TRSM chain (the only replicated work); everything that updates a block
 * before that -- subtree Schur updates, top-level Schur updates, the block's large-K left-looking GEMM -- is additive,
 * so each rank applies only ITS share (units [rank, rank+1) * units / nranks of the stream-K launch) to its own copy.
 *     sf_chol_plan_factorize_phase(plan, 0)                    own subtrees (as above)
 *     for k in 0 .. sf_chol_plan_num_segments(plan) - 1:
 *         all-reduce(sum) of every region of segment k         (RCCL over xGMI; regions are offsets into
 *         sf_chol_plan_factorize_segment(plan, k)               sf_chol_plan_factor_device_ptr, in doubles)
 * sf_subtree_partition_weighted: partition cost = top_weight * top flops + heaviest rank's subtree flops
 * (top_weight 1 = replicated top; about 1/nranks + the chain's share for the distributed top).
 * sf_chol_plan_set_stream: run the plan on a caller-owned HIP stream (e.g. the stream the collectives are ordered
 * with), so that segments and all-reduces need no host synchronisation between them. ---- */
int sf_subtree_partition_weighted(sf_long nsuper, const sf_long *Super, const sf_long *SuperMap, const sf_long *Lsip,
                                  const sf_long *Lsi, int nranks, double top_weight, int32_t *owner,
                                  double *top_fraction, double *max_load_fraction);
int sf_chol_plan_create_distributed(sf_chol_plan **plan, int device, sf_long n, sf_long nsuper,
                                    const sf_long *Super, const sf_long *SuperMap,
                                    const sf_long *Lsip, const sf_long *Lsi, const sf_long *Lsxp,
                                    const sf_long *Lp, const sf_long *Li, const int32_t *phase, int load_top,
                                    int rank, int nranks);
/* PROPORTIONAL MAPPING of the top: rank `rank`'s plan straight from the owner map of
 * sf_subtree_partition[_weighted].  A top supernode belongs to the GROUP of ranks whose subtrees lie below it: only they
 * store its panel, split its GEMMs and sum its block columns (a sub-communicator per group), and the groups of one level
 * work concurrently.  A rank stores its own subtrees plus the top supernodes on its path to the root instead of the
 * whole top.  Run with sf_chol_plan_factorize_distributed (it creates the sub-communicators on first use). */
int sf_chol_plan_create_mapped(sf_chol_plan **plan, int device, sf_long n, sf_long nsuper,
                               const sf_long *Super, const sf_long *SuperMap,
                               const sf_long *Lsip, const sf_long *Lsi, const sf_long *Lsxp,
                               const sf_long *Lp, const sf_long *Li, const int32_t *owner, int rank, int nranks);
/* ---- schedule inspection: what a rank WILL do, readable without running it.  sf_chol_plan_schedule_mapped /
 * sf_lu_plan_schedule_mapped build rank `rank`'s plan exactly as sf_*_plan_create_mapped does but touch no device and allocate
 * nothing (no GPU needed; the handle only serves the inspection calls below, "bytes_device" of sf_chol_plan_stat -- the bytes the
 * real plan would allocate -- and sf_chol_plan_destroy; everything else returns SF_ERR_ARG).  With them all N plans of a
 * factorization that needs N GPUs (BASELINE config 4: 256^3 over 8) can be checked against each other on a box with one GPU or
 * none: same collective sequence inside every group, split launches partitioned exactly once, every top panel stored by exactly
 * its group (tests/test_config4_schedules.py).  The reference has no counterpart (one host task queue, C:2267).
 *   launch_info   out[10] = kind (0 potrf/getrf, 1 trsm, 2 in-block GEMM, 3 Schur GEMM, 4 outer GEMM, 5 fused step, 6 small Schur),
 *                 tasks, divisible items (GEMM: stream-K units, kind 6: tiles), split (0/1), this rank's window [lo, hi) of the
 *                 items, replicated-bit-identical (0/1), segment index (-1: the rank's own subtrees), index in group, group size
 *   segment_info  out[6] = group mask, first launch, end launch, packed doubles of its all-reduce, early (issued one segment ahead
 *                 on the second stream: 0/1), regions
 *   panel_offsets xp[nsuper] = offset (doubles) of supernode s's panel on this rank, -1 = not stored here
 *   solve_reduce_info  out[3] = group mask, first column, columns: the sums of the distributed forward sweep, in issue order */
int sf_chol_plan_schedule_mapped(sf_chol_plan **plan, sf_long n, sf_long nsuper,
                                 const sf_long *Super, const sf_long *SuperMap,
                                 const sf_long *Lsip, const sf_long *Lsi, const sf_long *Lsxp,
                                 const sf_long *Lp, const sf_long *Li, const int32_t *owner, int rank, int nranks);
sf_long sf_chol_plan_num_launches(const sf_chol_plan *plan);
int sf_chol_plan_launch_info(const sf_chol_plan *plan, sf_long k, sf_long *out10);
int sf_chol_plan_segment_info(const sf_chol_plan *plan, sf_long k, sf_long *out6);
int sf_chol_plan_panel_offsets(const sf_chol_plan *plan, sf_long *xp);
/* OWNER-COMPUTES PROTOTYPE (environment SF_TOP_OWNER=1 at plan creation, Cholesky): in the sets every rank takes
 * part in (the root separator) the near GEMM and the 64-column chain of a 512-column block run on ONE rank (block number mod group
 * size) and the finished block column is broadcast before the split far GEMMs that read it -- instead of running replicated on
 * every rank.  out2[0] = owner's index in the group (-1: replicated, the default), out2[1] = launches of the owner's part.
 * Measured and NOT the default: DESIGN.md section 7. */
int sf_chol_plan_segment_owner(const sf_chol_plan *plan, sf_long k, sf_long *out2);
sf_long sf_chol_plan_num_solve_reduces(const sf_chol_plan *plan);
int sf_chol_plan_solve_reduce_info(const sf_chol_plan *plan, sf_long k, sf_long *out3);
/* bit r set = rank r takes part in the all-reduce of segment k */
uint32_t sf_chol_plan_segment_group(const sf_chol_plan *plan, sf_long k);
sf_long sf_chol_plan_num_segments(const sf_chol_plan *plan);
/* offsets/counts may be NULL to query *nregions only */
int sf_chol_plan_segment_regions(const sf_chol_plan *plan, sf_long k, sf_long capacity, sf_long *nregions,
                                 sf_long *offsets, sf_long *counts);
/* alternative to reducing the regions in place: gather the part of segment k's regions that can be non-zero (rows from
 * each block's first column down) into ONE contiguous device buffer; all-reduce(sum) *device_ptr[0 .. *count) and call
 * sf_chol_plan_factorize_segment(plan, k), which scatters it back first.  One collective per segment and the
 * structurally zero rows above the blocks' diagonals (half of a square root panel) stay off the wire. */
int sf_chol_plan_segment_pack(sf_chol_plan *plan, sf_long k, void **device_ptr, sf_long *count);
int sf_chol_plan_factorize_segment(sf_chol_plan *plan, sf_long k, int sync);
int sf_chol_plan_set_stream(sf_chol_plan *plan, void *hip_stream);

/* ---- the parent-front merge in C: a communicator per rank, RCCL (= NCCL on ROCm, over xGMI) underneath,
 * loaded at run time (librccl.so.1) so that single-GPU users do not need it.
 *   one process per GPU:   rank 0 calls sf_comm_unique_id, hands the 128 bytes to the other ranks by any means (MPI,
 *                          torch.distributed, a file), every rank calls sf_comm_create_rccl(&comm, device, rank, nranks, id)
 *   one process, N GPUs:   SparseFrame_allocate_gpu / sf_handlers_allocate create the communicators themselves
 * sf_chol_plan_factorize_distributed (Cholesky and LU plans): own subtrees, then per segment pack -> all-reduce(sum) ->
 * chain + this rank's share of the split GEMMs, all on the plan's stream; host_out != NULL also copies this rank's pieces
 * of the factor (own panels + its share of the top panels) into host_out while it computes. ---- */
typedef struct sf_comm sf_comm;
int sf_comm_unique_id(char *id128);
int sf_comm_create_rccl(sf_comm **comm, int device, int rank, int nranks, const char *id128);
int sf_comm_rank(const sf_comm *comm);
int sf_comm_size(const sf_comm *comm);
int sf_comm_allreduce_sum(sf_comm *comm, void *device_buf, sf_long count, void *hip_stream);
int sf_comm_destroy(sf_comm *comm);
/* test hook: split the communicator (all ranks, one colour), all-reduce on the child, destroy it */
int sf_comm_selftest_split(sf_comm *comm, void *device_buf, sf_long count, void *hip_stream);
/* Optional, collective: create the sub-communicators of the plan's groups now and check the world and every group of this rank
 * with one 8-byte sum (SF_ERR_HIP when a result is wrong) -- so that a launcher can still fall back when RCCL is not usable. */
int sf_chol_plan_prepare_comm(sf_chol_plan *plan, sf_comm *comm);
int sf_chol_plan_factorize_distributed(sf_chol_plan *plan, sf_comm *comm, sf_float *host_out /* or NULL */, int sync);
/* The solve with a factor that stays distributed (mapped plans after sf_chol_plan_factorize_distributed; Cholesky and LU): b_host =
 * the whole right-hand side in the permuted numbering on every rank; every rank writes into x_host the entries it is responsible
 * for (its subtrees' columns + the columns of the shared supernodes whose group it leads), the others are left alone.  One small
 * sum per shared supernode in the forward sweep is all that travels. */
int sf_chol_plan_solve_distributed(sf_chol_plan *plan, sf_comm *comm, const sf_float *b_host, sf_float *x_host);
/* Failure behaviour of the two distributed drivers: everything that can fail on one rank alone is done first and the ranks agree on
 * the outcome with one 8-byte sum BEFORE the first data collective (a failed rank always joins that sum; its word is allocated with
 * the plan); if any rank failed, every rank returns (its own code, or SF_ERR_PEER) with nothing enqueued on the communicator.  A
 * failure in the MIDDLE of a run (a device or link failure by then) aborts THIS rank's communicator (ncclCommAbort, a local
 * operation): this rank returns its error and later calls on the communicator return SF_ERR_PEER; RCCL peers already inside a
 * collective are NOT released by that -- they rely on their own timeout / the job's supervisor.  Emulated ranks keep their host-side
 * hand-shakes going, so all of them return.  The RCCL paths have been exercised with one rank only (no multi-GPU box so far).
 * sf_test_inject_failure (test hook): rank `rank` fails once at `where` = 1 before the first collective of a factorization, 2 in the
 * middle of its segments, 3 before the first collective of a solve, 4 in the middle of its forward sweep. */
void sf_test_inject_failure(int rank, int where);

/* ---- device plan for supernodal no-pivot LU (replaces L:2668-3573 + LU/Source/cuda_kernel.cu:22-176).
 * Symbolic arrays from sf_symbolic_create_lu; Lsxp is the reference's (packed (2*nsrow-nscol) x nscol) offsets.
 * Up/Ui = U by row; pass NULL for both when the input is symmetric (U aliases L, L:2718-2729).
 * The reference never pivots (magma_dgetrf_nopiv L:2653, cusolverDnDgetrf with devIpiv = NULL L:3344): inputs must
 * be factorizable without pivoting (e.g. diagonally dominant); a zero pivot returns SF_ERR_NOT_POSDEF. ---- */
typedef struct sf_chol_plan sf_lu_plan;
int sf_lu_plan_create(sf_lu_plan **plan, int device, sf_long n, sf_long nsuper,
                      const sf_long *Super, const sf_long *SuperMap,
                      const sf_long *Lsip, const sf_long *Lsi, const sf_long *Lsxp,
                      const sf_long *Lp, const sf_long *Li, const sf_long *Up, const sf_long *Ui);
/* multi-GPU: as sf_chol_plan_create_distributed; the phase / segment / pack / stream entry points of sf_chol_plan_*
 * take an LU plan as they are (same handle type); each segment carries the L and the U^T blocks */
int sf_lu_plan_create_distributed(sf_lu_plan **plan, int device, sf_long n, sf_long nsuper,
                                  const sf_long *Super, const sf_long *SuperMap,
                                  const sf_long *Lsip, const sf_long *Lsi, const sf_long *Lsxp,
                                  const sf_long *Lp, const sf_long *Li, const sf_long *Up, const sf_long *Ui,
                                  const int32_t *phase, int load_top, int rank, int nranks);
int sf_lu_plan_create_mapped(sf_lu_plan **plan, int device, sf_long n, sf_long nsuper,
                             const sf_long *Super, const sf_long *SuperMap,
                             const sf_long *Lsip, const sf_long *Lsi, const sf_long *Lsxp,
                             const sf_long *Lp, const sf_long *Li, const sf_long *Up, const sf_long *Ui,
                             const int32_t *owner, int rank, int nranks);
int sf_lu_plan_schedule_mapped(sf_lu_plan **plan, sf_long n, sf_long nsuper,
                               const sf_long *Super, const sf_long *SuperMap,
                               const sf_long *Lsip, const sf_long *Lsi, const sf_long *Lsxp,
                               const sf_long *Lp, const sf_long *Li, const sf_long *Up, const sf_long *Ui,
                               const int32_t *owner, int rank, int nranks);   /* schedule-only, see sf_chol_plan_schedule_mapped */
/* out of core: as sf_chol_plan_create_ooc (budget: two device doubles per panel entry) */
int sf_lu_plan_create_ooc(sf_lu_plan **plan, int device, sf_long n, sf_long nsuper,
                          const sf_long *Super, const sf_long *SuperMap,
                          const sf_long *Lsip, const sf_long *Lsi, const sf_long *Lsxp,
                          const sf_long *Lp, const sf_long *Li, const sf_long *Up, const sf_long *Ui,
                          const int32_t *group, int ngroups, int top_mode);
int sf_lu_plan_set_values(sf_lu_plan *plan, const sf_float *Lx, const sf_float *Ux /* NULL if U aliases L */);
/* Pivoting (BASELINE config 5 asks for it, the reference has none: magma_dgetrf_nopiv L:2653, devIpiv = NULL
 * L:3344, static pre-pivot L:589-673 disabled).  The symbolic structure is static, so rows can only be exchanged where that
 * keeps the structure: INSIDE the 64 x 64 diagonal block of a 64-column step.  Threshold partial pivoting there -- the natural
 * row keeps the pivot while |a_jj| >= tol * max over the block's unused rows of |a_ij| -- and a pivot smaller than
 * perturb * max|a_ij| is replaced by +- that value ("perturbed_pivots" stat; refine the solution iteratively then).
 * tol in [0, 1]: 0 = no pivoting (exactly the reference's behaviour; a zero pivot is SF_ERR_NOT_POSDEF when perturb is 0 too),
 * 1 = partial pivoting.  Defaults: tol 0, perturb 0 = the reference (env SF_LU_PIVOT_TOL at plan creation: that tol and perturb sqrt(eps); SF_LU_PERTURB).  On a
 * matrix whose natural pivots pass the threshold (e.g. diagonally dominant) the factor is bit-identical to the no-pivot one.
 * The interchanges are recorded per block and applied block by block in the forward solve (LINPACK-style: the L entries to
 * the left of a block keep their rows). */
int sf_lu_plan_set_pivoting(sf_lu_plan *plan, double tol, double perturb);
/* pivpos[g] = row position (global permuted index, inside the same 64-column block) of original row g; n entries */
int sf_lu_plan_get_pivots(sf_lu_plan *plan, sf_long *pivpos);
int sf_lu_plan_factorize(sf_lu_plan *plan, int sync);
int sf_lu_plan_sync(sf_lu_plan *plan);
/* D2H copy of the factor gathered into the reference layout: panel s = (2*nsrow-nscol) x nscol column-major,
 * rows [0,nscol) packed L11\U11, [nscol,nsrow) L21, [nsrow,2*nsrow-nscol) U12^T (L:2514-2517) */
int sf_lu_plan_get_factor(sf_lu_plan *plan, sf_float *Lsx);
/* device-side solve with the resident factors: x <- (L U)^{-1} b, permuted space (device twin of L:3592-3700) */
int sf_lu_plan_solve(sf_lu_plan *plan, const sf_float *b_host, sf_float *x_host);
double sf_lu_plan_stat(const sf_lu_plan *plan, const char *name);
int sf_lu_plan_set_profiling(sf_lu_plan *plan, int on);
int sf_lu_plan_destroy(sf_lu_plan *plan);

/* ---- device handlers: what SparseFrame_allocate_gpu / _free_gpu / _factorize_supernodal of BOTH struct libraries
 * forward to (reference C:16-366, C:2150-3017).  A handler keeps a lock and a cache of device plans keyed by the
 * symbolic pattern, so repeated factorizations of one pattern pay for the plan once.  lu != 0: LU arrays (Up/Ui/Ux
 * NULL when U aliases L).  serial selects the handler (matrix threads spread over the devices). ---- */
int sf_handlers_allocate(struct common_info_struct *common_info, struct gpu_info_struct **gpu_info_list_ptr);
int sf_handlers_free(struct common_info_struct *common_info, struct gpu_info_struct **gpu_info_list_ptr);
/* device plans built so far by the handlers of a list (a repeated sparsity pattern must not add to it) */
int64_t sf_handlers_plan_builds(struct gpu_info_struct *gpu_info_list, int n_handlers);
/* The device pool of handler d (allocated once in SparseFrame_allocate_gpu, where the reference allocates its eight device slots,
 * C:92-283: min(8 x devSlotSize, a quarter of the device); SF_DEVICE_POOL_MB overrides, 0 = none; one-matrix-per-handler lists only).
 * The handler lends it to the plan of ONE cached pattern at a time for its factor, so that the first SparseFrame_factorize of a
 * pattern does not start with a hipMalloc of tens of GB.  out[0] = bytes, out[1] = 1 while lent, out[2] = n of the borrowing plan. */
int sf_handlers_pool_info(struct gpu_info_struct *list, int d, sf_long *out);
int sf_handlers_factorize(struct common_info_struct *common_info, struct gpu_info_struct *gpu_info_list, int lu, int serial,
                          sf_long n, sf_long nsuper, const sf_long *Super, const sf_long *SuperMap,
                          const sf_long *Lsip, const sf_long *Lsi, const sf_long *Lsxp,
                          const sf_long *Lp, const sf_long *Li, const sf_long *Up, const sf_long *Ui,
                          const sf_float *Lx, const sf_float *Ux, sf_float *Lsx_out,
                          sf_long *PivOut /* LU: row positions after the in-block interchanges (n entries), or NULL */);

/* LU pivoting policy of the struct entry points (process-wide, takes effect at the next SparseFrame_factorize of the LU library).
 * Default: none set = the reference's behaviour, no pivoting and no perturbation (L:2653); tol in (0, 1] switches on threshold partial
 * pivoting inside the 64 x 64 diagonal blocks, perturb > 0 the replacement of pivots below perturb * max|a_ij| (see
 * sf_lu_plan_set_pivoting).  sf_handlers_perturbed_pivots: perturbed pivots of the factorization that filled this host array
 * (-1: unknown array; with several handlers a shared panel's perturbations are counted once per rank that stores it). */
int sf_handlers_set_lu_pivoting(double tol, double perturb);
/* the policy of the NEXT sf_handlers_factorize call made by the calling thread only (how the LU struct library passes a
 * matrix_info's own setting, SparseFrame_set_matrix_pivoting); precedence: this, the process-wide one, the plan's creation default */
int sf_handlers_set_lu_pivoting_next_call(double tol, double perturb);
int64_t sf_handlers_perturbed_pivots(const sf_float *Lsx_host);

/* The struct path's solve with the RESIDENT factor: after SparseFrame_factorize the factor is still in the handler's cached plan (or
 * spread over the handlers' plans); SparseFrame_solve_supernodal (which only receives matrix_info) asks here, by the address of the
 * host copy.  Solves on the device(s) (b, x in the permuted numbering, n doubles each way) when the plan still holds THAT factorization
 * (generation counter) AND the caller's array still is what the device holds: a 64-bit fingerprint of every panel -- sum over its
 * values of bits(v) * (2 index + 1) * K mod 2^64, any single changed value changes it -- is computed over the WHOLE host array (one
 * threaded pass, ~0.1 s for 30 GB) and compared with the device's (computed once per factorization, about one read of the factor).
 * Otherwise returns SF_ERR_ARG and the caller solves on the host as the reference does (C:3036-3139).
 * sf_handlers_set_resident_solve(mode): 0 = never (always the host sweep), 1 = verified as above (default), 2 = trusted: the caller
 * guarantees it does not modify Lsx between factorize and solve; the comparison is skipped (0.02 s instead of 0.13 s at 128^3).
 * SF_SOLVE=host in the environment forces the host solve.  forget: the host copy is being freed. */
int sf_handlers_set_resident_solve(int mode);
/* 1 if this build keeps the A/B environment switches of finished experiments (make EXP=1), 0 for a release build */
int sf_build_experiments(void);
int sf_handlers_solve_resident(const sf_float *Lsx_host, const sf_float *b, sf_float *x);
/* the same with the symbolic arrays of the matrix at hand: after a factorization by SEVERAL handlers the factor is spread over
 * their plans; with these arrays the library can build a whole plan on the first handler's device, gather the panels into it
 * (device to device) and solve there -- when one device has room for the whole factor; otherwise SF_ERR_ARG (host solve) */
int sf_handlers_solve_resident_sym(const sf_float *Lsx_host, const sf_float *b, sf_float *x, int lu, sf_long n, sf_long nsuper,
                                   const sf_long *Super, const sf_long *SuperMap, const sf_long *Lsip, const sf_long *Lsi,
                                   const sf_long *Lsxp, const sf_long *Lp, const sf_long *Li, const sf_long *Up, const sf_long *Ui);
void sf_handlers_forget(const sf_float *Lsx_host);
int64_t sf_handlers_replica_mismatches(const sf_float* Lsx_host); /* values of shared panels that differ bitwise between ranks (tests; -1: not a multi-handler factor) */
int64_t sf_handlers_resident_solves(void);     /* how many solves were served from a resident factor so far (tests) */
int64_t sf_handlers_fingerprint_fallbacks(void); /* verified solves that found Lsx changed (fingerprint mismatch) and left the solve to the host sweep */

/* number of HIP devices visible (0 on a CPU-only box; never fails) */
int sf_device_count(void);
/* version string */
const char *sf_version(void);
/* layout probe for FFI authors: "sizeof_common", "sizeof_matrix", "offsetof_Lsx", "offsetof_workspace",
 * "offsetof_residual", "offsetof_devSlotSize" as this library was compiled; -1 for an unknown name */
long sf_abi_layout(const char *name);

/* total memory of HIP device `device` in bytes (0 if it does not exist) */
size_t sf_device_memory(int device);
/* the reference's slot size for `ndev` devices whose smallest memory is min_mem (C:36-41, C:82-87, C:199) */
size_t sf_reference_slot_size(int ndev, size_t min_mem);

#ifdef __cplusplus
}
#endif
#endif /* SPARSEFRAME_FLAT_H */

// Private view of the device plan shared by the translation units of libsparseframe_hip.so (not part of the C ABI).
#pragma once
#include <sparseframe_hip.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "sf_kernels.h"

using sf::GemmProb;
using sf::GemmTask;
using sf::PotrfTask;
using sf::TrsmTask;
using sf::StepTask;

struct Launch {
    int kind;       // 0 potrf, 1 trsm, 2 gemm panel (inner, K = NB), 3 gemm scatter, 4 gemm panel (outer, large K),
                    // 6 Schur updates with K <= SU_MAXK (k_update_small, tasks in stasks),
                    // 5 fused step k_step (Cholesky, steps of at most GEMM_GRID workgroups; otherwise and for LU: 2, 0, 1)
    int64_t first;  // first task
    int count;
    int64_t prefix_first = 0;   // GEMM launches: first entry of this launch's K-step prefix (count + 1 entries)
    uint32_t units = 0;         // GEMM launches: total number of (tile, K step) units
    double flops = 0;           // GEMM launches: algorithmic flops of the problems in this launch
    bool split = false;         // distributed top: this rank executes the units [share_lo, share_hi) x units of this launch
    int share_idx = 0, share_cnt = 1;
    bool whole_tiles = false;   // GEMM launch executed in full by every rank of a group: no K-splitting, bit-identical results (see k_gemm)
    double share_lo = 0.0, share_hi = 1.0;      // = share_idx / share_cnt, (share_idx + 1) / share_cnt unless the set's shares are weighted
    int ticket = 0;             // k_step launches: index of the launch's task-claim counter (d_info[1 + ticket])
    // one-GPU look-ahead (sf_chol_plan::lookahead1): lane 1 = the plan's second stream; wait_ev / rec_ev index sf_chol_plan::la_events
    // (the launch's stream waits for wait_ev before it, records rec_ev after it); -1 = none
    int8_t lane = 0;
    int wait_ev = -1, rec_ev = -1;
};

// Distributed top phase (sf_chol_plan_create_distributed): phase 1 is cut into segments.  Before a segment runs, the
// regions it lists (the 512-column block of every top panel of the level whose 64-column chain is about to start) are
// summed over the ranks; until then a block only ever received additive updates (subtree Schur updates, split top
// Schur updates, split outer GEMMs), so its true value is the sum of the ranks' copies.
struct Segment {
    uint32_t mask = 0;                     // the ranks that hold (and sum) these block columns
    size_t l0 = 0, l1 = 0;                 // launches [l0, l1)
    std::vector<int64_t> off, cnt;         // whole block columns: doubles, relative to the factor base pointer
    // the part of each region that can be non-zero (rows >= the block's first column): `cols` pieces of `rows` doubles,
    // `ld` apart, starting at `src` -- what sf_chol_plan_segment_pack gathers into one contiguous buffer
    std::vector<int64_t> src, rows, cols, ld;
    int64_t packed = 0;                    // doubles in the packed buffer
    // look-ahead schedule: every additive contribution to these block columns except the one of the block column just before
    // them (which the segment's own first launch adds, replicated) is enqueued before the PREVIOUS segment starts, so the sum
    // over the ranks may run beside the previous segment's chain
    bool early = false;
    // OWNER-COMPUTES prototype (SF_TOP_OWNER=1): the near GEMM and the 64-column chain of this block column,
    // launches [l0, lc), run on ONE rank of the group only (group index owner_gi = block number mod group size); the finished block
    // column is then broadcast to the group before the launches [lc, l1) (the split far GEMMs, which read it).  -1: everybody runs
    // everything (the default: replicate and all-reduce).
    int owner_gi = -1;
    size_t lc = 0;
};

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "[sparseframe-hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SF_ERR_HIP;                                                              \
        }                                                                                   \
    } while (0)

template <class T>
static inline int upload(T** dptr, const std::vector<T>& h, size_t* bytes_total) {
    *dptr = nullptr;
    const size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc((void**)dptr, bytes));
    if (!h.empty()) HIP_TRY(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *bytes_total += bytes;
    return SF_OK;
}

// Overlapped copy-back of the factor (sf_chol_plan_factorize_to_host; the reference overlaps the D2H of finished blocks
// with compute on s_cudaStream_copyback, C:2888-2895).  A piece is a run of finished panel columns that is contiguous in
// the host layout (and, for Cholesky, in the device layout), at most DL_SLOT doubles; it may be copied once launch
// `ready` - 1 has completed (event `ev`).  Pieces are sorted by `ready`.
constexpr int64_t DL_SLOT_DEFAULT = (int64_t)4 << 20;   // doubles per staging slot (32 MiB)
constexpr int DL_WORKERS_DEFAULT = 4;                   // copy workers: own HIP stream + 2 pinned slots each.  4 = the number of hardware
                                                        // queues the runtime gives the ordinary-priority streams: with 6 two pairs of
                                                        // workers share a queue and the struct call at 128^3 takes 553-603 ms instead of
                                                        // 552-568 (profiles/r02_f_struct_slow_mode.txt, part 5)
constexpr int DL_WORKERS_FRESH = 6;
constexpr int DL_WORKERS_MAX = 16;                      // SF_DL_WORKERS / SF_DL_SLOT_MB (read at plan creation) tune both
struct DlPiece {
    int64_t dev_off, host_off, count;   // doubles; LU: dev_off unused (the piece is packed from the (L, U^T) panels)
    size_t ready;
    int ev;
    // Cholesky block columns right of the first one: the rows above the block's first column are structurally zero (never written on
    // the device either), so only rows [skip, ld) of each of the `ncols` columns cross PCIe (count = ncols * (ld - skip), a 2-D
    // copy) and the host side zero-fills the prefixes.  ld == 0: a plain contiguous piece.
    int64_t skip = 0, ld = 0, ncols = 0;
    // LU (direct form, no pack kernel): the piece is the columns [j0, j0 + ncols) of supernode s0 (s1 == s0 + 1, ld = nsrow: the
    // L part is one contiguous run of the L panel, the U part -- rows [nscol, nsrow) of the same columns of the U^T panel -- a 2-D
    // copy) or the whole supernodes [s0, s1) (ld == 0: both panels' runs copied as they are, dev_count doubles each); the copy
    // worker interleaves the columns into the reference layout (L:2514-2517).  s0 < 0: not an LU-direct piece.
    int32_t s0 = -1, s1 = -1;
    int64_t j0 = 0, dev_count = 0;
    int32_t group = -1;     // out-of-core plans: the streamed group the piece belongs to (-1: a top panel)
};

// struct path's device pool (sf_handlers.hip): the next plan_create of this thread may use `ptr` for its factor
extern "C" void sf_plan_offer_factor_buffer(double* ptr, size_t bytes);
extern "C" int sf_plan_factor_borrowed(const sf_chol_plan* p);

struct sf_comm;      // one rank's end of a multi-GPU group (sf_multi.hip)
struct sf_chol_plan;
// one step of the solve sweeps (sf_chol_plan.hip; shared by sf_chol_plan_solve and sf_chol_plan_solve_distributed)
void sf_solve_step_fwd(sf_chol_plan* p, size_t k, const double* base, int* sync, int* tickets, hipStream_t st);
void sf_solve_step_bwd(sf_chol_plan* p, size_t k, const double* base, int* sync, int* tickets, hipStream_t st);

struct sf_chol_plan {
    // ---- overlapped download schedule (built once) and the state of a running download ----
    std::vector<DlPiece> dl_pieces;
    std::vector<size_t> dl_ev_ready;            // launch count after which event k is recorded
    std::vector<hipEvent_t> dl_events;
    bool dl_active = false;                     // run_launches records the events and publishes them
    size_t dl_next_ev = 0;
    std::mutex dl_mu;
    std::condition_variable dl_cv;
    size_t dl_published = 0;                    // events [0, dl_published) have been recorded (guarded by dl_mu)
    bool dl_abort = false;
    double* h_ring = nullptr;                   // pinned staging ring: dl_workers x 2 slots of dl_slot doubles
    double* d_ring = nullptr;                   // LU: device staging of packed pieces, same shape
    int64_t dl_slot = DL_SLOT_DEFAULT;
    int dl_workers = DL_WORKERS_DEFAULT;
    int dl_workers_fresh = DL_WORKERS_FRESH;    // ... when the destination's pages have never been touched (sf_dl_begin)
    int dl_workers_last = 0;                    // workers of the last download
    hipStream_t dl_streams[DL_WORKERS_MAX] = {};
    // LU direct download: before a piece's event is recorded, the upper triangle (with the diagonal) of its columns' part of the
    // supernode's diagonal block is filled into the L panel from the U^T panel (k_lu_fill_u11), so that the L panel holds the
    // reference's packed L11 \ U11 and the download needs no gather kernel
    bool dl_host_wait = true;                   // copy workers wait for a piece's event on the host (see dl_worker)
    bool dl_lu_direct = false;
    void* d_fill = nullptr;                     // sf::FillTile[], grouped by download event
    std::vector<int64_t> fill_first;            // tiles of event k: [fill_first[k], fill_first[k + 1])
    int dl_cpus_known = 0;                      // 0 not looked up yet, 1 dl_cpus holds the CPUs of the device's NUMA node, -1 none / disabled
    std::vector<int> dl_cpus;
    int dl_last_cpu[DL_WORKERS_MAX] = {};       // CPU each copy worker finished its last download on (SF_TRACE)
    hipEvent_t dl_done[DL_WORKERS_MAX][2] = {};
    double last_to_host_ms = 0;                 // wall time of the last sf_chol_plan_factorize_to_host
    std::vector<std::thread> dl_threads;
    std::vector<double> dl_trace;               // SF_DL_TRACE: 3 times per piece
    double dl_t0 = 0;
    std::atomic<int> dl_error{0};
    std::atomic<size_t> dl_next_piece{0};
    double* dl_host = nullptr;                  // destination of the running download (reference layout)
    // LU pivoting (sf_kernels.h, PivotCtl): threshold inside the 64 x 64 diagonal blocks, perturbation = piv_perturb * max|a_ij|
    // Defaults = the reference's behaviour: no pivoting, no perturbation (magma_dgetrf_nopiv, L:2653); opt-in by sf_lu_plan_set_pivoting,
    // SparseFrame_set_pivoting (struct path) or SF_LU_PIVOT_TOL / SF_LU_PERTURB at plan creation
    double piv_tol = 0.0, piv_perturb = 0.0, amax = 0;
    double piv_tol0 = 0.0, piv_perturb0 = 0.0;      // the policy the plan was created with (environment or none): what a struct call without a policy of its own gets
    int32_t* d_piv = nullptr;   // pivpos[n] | pivinv[n] | perturbation counter
    int last_perturbed = 0;
    bool dry = false;           // schedule-only plan (plan_create's dry mode): no device resources, inspection only
    bool lu = false;            // no-pivot LU: every supernode has an L panel and a U^T panel (see plan_create)
    int64_t xC = 0;             // doubles in one set of nsrow x nscol panels (Cholesky: == xsize)
    int64_t unz = 0;            // LU: entries of U (by row)
    int64_t* d_Up = nullptr;    // LU, unsymmetric input: U by row
    int32_t* d_Ui = nullptr;
    double* d_Ux = nullptr;
    int64_t* d_Xp = nullptr;    // LU: offsets of the nsrow x nscol panels (the reference's Lsxp has the packed sizes)
    double* d_pack = nullptr;   // LU: staging buffer in the reference layout for the download
    bool u_alias = false;       // LU with a symmetric input: U aliases L (reference L:2718-2729)
    // multi-GPU sharding: phase[s] = 0 owned subtree supernode, 1 top supernode (replicated), -1 not on this rank
    std::vector<int8_t> phase;
    std::vector<int64_t> h_XP;  // device panel offsets (doubles), -1 when the panel is not stored on this rank
    bool partial = false;       // some supernodes are absent or the top panels are not loaded here
    int64_t top_off = 0, top_size = 0;   // contiguous region of the top panels inside one panel set
    size_t launch_split = 0;    // launches [0, launch_split) belong to phase 0, the rest to phase 1
    // out-of-core plan (plan_create, ooc_group): number of streamed groups (0: in core), entries of one of the two group buffers,
    // pieces per group and how many of them are still on the device during a download
    // SF_GRAPH=1 (sf_chol_plan_factorize): the resident factorization as one hipGraph
    bool use_graph = false, capturing = false;
    hipGraphExec_t graph_exec = nullptr;
    int graph_epoch = 0;
    int32_t n_flags = 1;
    double graph_tol = 0, graph_eps = 0;
    hipStream_t graph_stream = nullptr;
    bool factor_borrowed = false;       // d_Lsx is a buffer the creator lent (sf_plan_offer_factor_buffer): never freed here
    int ooc_groups = 0;
    int64_t ooc_buf = 0;
    // top mode 1 (active top panels only): first / last group below every top supernode, and per group the (offset, length) ranges its
    // first launch zeroes for the top panels that start with it
    int ooc_top_mode = 0;
    std::vector<int32_t> ooc_first, ooc_last, ooc_wait;     // ooc_wait[g]: the last group whose copies group g's first launch waits for
    std::vector<std::vector<std::pair<int64_t, int64_t>>> ooc_zero;
    std::vector<int64_t> dl_group_pieces;
    std::unique_ptr<std::atomic<int64_t>[]> dl_group_left;
    int rank = 0, nranks = 1;   // distributed top: this rank's share of the split launches
    std::vector<Segment> segments;
    std::vector<uint32_t> all_masks;    // every group of the factorization (all ranks build the same sorted list)
    double* d_status = nullptr;    // one word: the ranks' status agreement before the first data collective (sf_multi.hip)
    double* d_scratch = nullptr;   // packed segment buffers (2 x max over the segments: segment k uses half k & 1)
    int64_t scratch_elems = 0;
    bool lookahead1 = false;       // ONE GPU: the far part of block jo+2's update on the second stream beside the chain of block jo+1 (experiment)
    int la_grid = 0;               // workgroups of a lane-1 GEMM (0 = the full persistent grid)
    std::vector<hipEvent_t> la_events;
    bool top_owner = false;        // SF_TOP_OWNER=1: owner-computes chains for the sets every rank takes part in (prototype, see Segment)
    bool lookahead = true;         // shared top panels: outer GEMM split into a far part (ahead of the previous chain) and the last block's
    hipStream_t stream2 = nullptr; // the collectives of look-ahead segments and their pack copies
    hipEvent_t ev_contrib[2] = {nullptr, nullptr}, ev_reduced[2] = {nullptr, nullptr}, ev_unpacked[2] = {nullptr, nullptr};
    bool unpacked_recorded[2] = {false, false};
    int64_t packed_pending = -1;   // segment whose packed buffer has to be scattered back before it runs
    bool own_stream = true;
    int8_t* d_loadmask = nullptr;
    int64_t *d_loadmapL = nullptr, *d_loadmapU = nullptr;      // whole plans: offset into d_Lsx of every entry of L (/ U), -1 = not loaded
    // device solve (Cholesky, whole matrix on one device): task lists per (level, 64-column step)
    sf::SolveTask* d_solve = nullptr;
    // backward sweep: row-major copies of the diagonal blocks of the top levels' steps (made at the start of every solve)
    double* d_solveT = nullptr;
    int64_t* d_solveT_list = nullptr;
    int64_t n_solveT = 0;
    double* d_x = nullptr;
    double* d_resid = nullptr;  // sf_chol_plan_validate: r | column sums | b | 4 norms
    // the forward launch of a step has fwd_count tasks, the backward one `count` (equal unless the plan is one rank's part of a
    // distributed factor: the forward tiles that update ancestors' rows are dealt out over the group, the backward ones are not);
    // big: a panel of the step is wider than 64 columns; red_first / red_count: sums over a group that precede the forward launch
    struct SolveStep { int64_t fwd_first, bwd_first; int count; int big; int nrows_tasks; int small; int ndiag; int fwd_count; int red_first, red_count; };
    static constexpr int SOLVE_TICKETS = 3;     // ticket words per step: forward, backward, the backward sweep's second launch (two-launch form)
    struct SolveReduce { int64_t off, cnt; uint32_t mask; };        // x[off .. off + cnt) summed over the ranks of `mask`
    std::vector<SolveReduce> solve_reduces;
    std::vector<std::pair<int64_t, int64_t>> solve_own;           // column ranges whose solution this rank reports (distributed solve)
    std::vector<std::pair<int64_t, int64_t>> solve_load;          // column ranges whose right-hand side this rank loads
    int* d_solve_sync = nullptr;
    bool solve_bwd_fused = false;
    int n_solve_sync = 0;
    std::vector<SolveStep> solve_steps;
    double last_solve_ms = 0;
    int device = 0;
    int64_t n = 0, nsuper = 0, nnz = 0, isize = 0, xsize = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_s0 = nullptr, ev_s1 = nullptr;

    // device copies of the structure
    int64_t* d_Lp = nullptr;
    int32_t* d_Li = nullptr;
    double* d_Lx = nullptr;
    int32_t* d_Super = nullptr;
    int32_t* d_SuperMap = nullptr;
    int64_t* d_Lsip = nullptr;
    int32_t* d_Lsi = nullptr;
    int64_t* d_Lsxp = nullptr;
    double* d_Lsx = nullptr;
    int* d_info = nullptr;      // [0]: status bits of the running factorization; [1 ..]: task-claim counters of the k_step launches
    int n_tickets = 0;
    bool gemm_dynamic = true;   // k_gemm: whole-tile rounds claimed from per-XCD counters (SF_GEMM_DYNAMIC=0: static deal)

    PotrfTask* d_potrf = nullptr;
    TrsmTask* d_trsm = nullptr;
    StepTask* d_steps = nullptr;
    int* d_flags = nullptr;     // k_step: one flag per (panel, fused step); == epoch once its diagonal block is factored
    double* d_tinv = nullptr;   // k_step: inverses of the 16 x 16 diagonal sub-blocks, per diagonal task of the running launch
    int epoch = 0;
    std::vector<uint64_t> h_hash;    // per-supernode fingerprints of the factor of epoch hash_epoch (sf_plan_panel_hashes)
    int hash_epoch = -1;
    GemmProb* d_probs = nullptr;
    GemmTask* d_gtasks = nullptr;
    GemmTask* d_stasks = nullptr;   // tiles of k_update_small
    uint32_t* d_ktprefix = nullptr;
    int32_t* d_relmap = nullptr;

    std::vector<Launch> launches;
    int nlevels = 0;
    int64_t n_gemm_tasks = 0, n_pairs = 0;
    double flops_tiles = 0, flops_tiles_update = 0;   // MFMA flops the GEMM tiles issue (full 128 x 128 x 16 steps): all launches / Schur updates
    double flops_update_small = 0;      // part of flops_update done by k_update_small (K <= SU_MAXK)
    double flops_exec = 0, flops_update = 0, scatter_elems = 0, flops_panel_gemm = 0, flops_outer_gemm = 0;
    size_t bytes_device = 0;
    bool values_set = false;

    bool profiling = false;
    double last_ms = 0, last_load_ms = 0, last_panel_ms = 0, last_update_ms = 0;
    double last_kind_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int last_status = SF_OK;

    // host copies needed by the device solve
    std::vector<int64_t> h_Lsip, h_Lsxp;
    std::vector<int32_t> h_Super;
    std::vector<int> level_of;
};


// Overlapped download (sf_chol_plan.hip).  sf_dl_begin starts the copy workers of a factorization whose launches are
// about to be enqueued with run_launches (which records and publishes the piece events while dl_active is set);
// sf_dl_end publishes what is left, waits for the workers and returns SF_OK or the first error.
int sf_dl_begin(sf_chol_plan* p, double* host_out);
// per-supernode fingerprints (k_factor_hash) of the factor the plan holds, in the reference layout's index space; 0 for panels this
// rank does not store.  Computed once per factorization (about one read of the factor), then cached.
extern "C" int sf_plan_panel_hashes(sf_chol_plan* p, const uint64_t** out);
int sf_dl_end(sf_chol_plan* p);
// gathers the panels of the parts (plans of several ranks, one pattern) into the whole plan dst (sf_chol_plan.hip)
int sf_plan_import_from(sf_chol_plan* dst, sf_chol_plan* const* parts, int nparts);
// pipelined segment sums (sf_chol_plan.hip), used by sf_chol_plan_factorize_distributed
int sf_seg_begin(sf_chol_plan* p, sf_long k, void** dptr, sf_long* count);
void* sf_plan_stream2(sf_chol_plan* p);
int sf_seg_reduced(sf_chol_plan* p, sf_long k);
int sf_seg_finish(sf_chol_plan* p, sf_long k);
// owner-computes segments (Segment::owner_gi >= 0): sf_seg_finish runs only the owner's part [l0, lc); then the broadcast --
// sf_seg_bcast_begin packs the finished block column on the owner / zeroes the buffer elsewhere (main stream), the caller sums the
// buffer over the group (x + 0 + ... + 0: every rank gets the owner's values), sf_seg_bcast_finish scatters it back and runs [lc, l1)
int sf_seg_is_owner_segment(const sf_chol_plan* p, sf_long k);
int sf_seg_bcast_begin(sf_chol_plan* p, sf_long k, void** dptr, sf_long* count);
int sf_seg_bcast_finish(sf_chol_plan* p, sf_long k);
int sf_seg_early(const sf_chol_plan* p, sf_long k);

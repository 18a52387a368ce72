// Device task descriptors and kernel launchers for the supernodal Cholesky numeric phase (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include <cstdlib>
// A/B switches of finished experiments (the alternative each one selects lost its measurement: DESIGN "Environment variables") are
// compiled OUT of release builds -- `make EXP=1` (-DSF_EXPERIMENTS) brings them back for re-measuring; sf_build_experiments() tells.
static inline const char* sf_exp_env(const char* name) {
#ifdef SF_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

namespace sf {

constexpr int NB = 64;          // diagonal block size of the in-panel right-looking factorization
constexpr int OUTER_NB = 512;      // outer (left-looking) block-column width of the two-level panel factorization
constexpr int TRSM_ROWS = 256;  // rows per TRSM workgroup (one row per lane)
constexpr int ST_ROWS = 64;     // rows per workgroup of the fused 64-column step (k_step)
constexpr int GEMM_BM = 128;    // tile extent along ci (target rows; contiguous in memory)
constexpr int GEMM_BN = 128;    // tile extent along cj (target columns)
constexpr int GEMM_BK = 16;
constexpr int GEMM_WAVES = 8;     // waves per GEMM workgroup (2 along ci x GEMM_WAVES/2 along cj)
constexpr int GEMM_THREADS = 64 * GEMM_WAVES;
#ifndef SF_GEMM_GRID
#define SF_GEMM_GRID 512
#endif
constexpr int GEMM_GRID = SF_GEMM_GRID;   // persistent GEMM grid: 2 workgroups per CU x 256 CUs (the macro: experiment builds)
constexpr int GEMM_SLICE = 1 << 30;  // K steps per task (a tile's K range could be cut into slices; measured: it does not pay)
constexpr int SU_TM = 64, SU_TN = 32;   // tile of k_update_small (one wave): rows x columns
constexpr int SV_B = 256;         // columns per step of the device solve (4 sub-blocks of NB, one per wave)
constexpr int SV_ROWS = 64;       // rows per row tile of the device solve
constexpr int SU_MAXK = 64;      // Schur updates with K <= SU_MAXK go to k_update_small

// One C -= Y * X^T problem on rows of ONE source panel (column-major, leading dimension lda):
//   C[ci][cj] = sum_k src[y_off + ci + k*lda] * src[x_off + cj + k*lda],   0<=ci<M, 0<=cj<N, 0<=k<K
// only elements with ci >= cj are produced (lower trapezoid).
// mode 0 (panel):   target = Lsx[c_off + ci + cj*ldc], plain read-modify-write (tile owned by one workgroup)
// mode 1 (scatter): target = Lsx[c_off + rowmap(ci) + colmap(cj)*ldc] with the relative map of the
//                   source rows Lsi[src_rows + .] inside the target supernode's row list, fp64 atomic add.
struct GemmProb {
    int64_t y_off;      // doubles, into Lsx
    int64_t x_off;
    int64_t c_off;
    int64_t src_rows;   // index into Lsi of the source row id of ci = 0 (scatter mode)
    int64_t tgt_rows;   // index into Lsi of the target supernode's first below-diagonal row
    int32_t lda, ldc;
    int32_t M, N, K;
    int32_t tgt_first_col;  // Super[a]
    int32_t tgt_nscol;
    int32_t tgt_nbelow;     // nsrow_a - nscol_a
    int32_t strict;         // 1: only ci > cj is produced (LU: the L panel does not own the diagonal)
    int32_t pad_;
    int64_t map_off;        // scatter mode: offset of this (source, ancestor) pair's relative map (M entries) in RelMap
};

struct GemmTask {   // one 128x128 tile of a problem over ONE slice of its K range (k_update_small: a 64x32 tile, whole K)
    int32_t prob;
    uint16_t tm, tn;    // tile coordinates along ci / cj
    uint32_t kt0;       // first 16-deep K step of the slice
    uint32_t nkt;       // K steps in the slice
};

struct PotrfTask {  // factor the b x b diagonal block at (diag, diag) of a panel
    int64_t panel;      // doubles, into Lsx
    int32_t ld, diag, b;
    int32_t first_col;  // Super[s]: the block's rows are the global (permuted) indices first_col + diag + [0, b)  (LU pivot records)
};

// LU pivoting (the reference never pivots, L:2653 / L:3344, and keeps a disabled static pre-pivot,
// L:589-673): threshold partial pivoting RESTRICTED to the 64 x 64 diagonal block of a step -- the symbolic structure is
// static, rows outside the block cannot be exchanged -- with a perturbation fallback.
//   tol  > 0 : at column j the natural row keeps the pivot while |a_jj| >= tol * max_i |a_ij| (i over the block's rows not
//              yet used); otherwise the row of the largest entry takes it.  tol = 0: no pivoting (the reference's behaviour).
//   eps  > 0 : a pivot smaller than eps in magnitude is replaced by +-eps (counted in *nperturb); eps = 0: a zero pivot is an
//              error (info bit 0) as before.
//   pivpos[g] = position (global index, inside the same block) that original row g of the block ends up in; pivinv = inverse.
// Only the block's rows of the block's own columns and of the columns to its RIGHT move (the U rows); the L entries to the
// left of the block stay where they are, so the interchanges are applied LINPACK-style, block by block, in the forward solve.
struct PivotCtl {
    double tol, eps;
    int32_t* pivpos;    // n entries each, or nullptr when tol == 0 (then no record is written: positions are the identity)
    int32_t* pivinv;
    int* nperturb;
};

struct TrsmTask {   // rows [row0, row0+nrows) of panel columns [diag, diag+b) <- X * D^{-T}
    int64_t panel;      // panel holding X
    int64_t dpanel;     // panel holding the triangular block D at (diag, diag) (same ld); Cholesky: == panel
    int32_t ld, diag, b;
    int32_t row0, nrows;
    int32_t unit;       // 1: D has an implicit unit diagonal (LU: U12^T <- U12^T * L11^{-T}); with pivoting the tile's columns are
                        // first brought into pivot order (they are rows of U)
    int32_t first_col;  // Super[s]
    int32_t pad;
};

struct StepTask {   // fused 64-column step on rows [row0, row0+nrows) of panel columns [diag, diag+b), updated by columns [J, diag)
    int64_t panel;          // panel of the tile (and of its own rows' operand)
    int64_t xpanel;         // panel of the other operand = the diagonal block's rows, and of the triangular block the rows are
                            // solved against (Cholesky: == panel; LU: the U^T panel for L rows and vice versa)
    int32_t ld, J, diag, b; // J: first column of the left-looking update (Cholesky diagonal tasks: J == diag, their block is kept
                            // up to date right-looking by the row tasks' pushes, see next_b)
    int32_t row0, nrows;    // row0 == diag: the diagonal block (POTRF / GETRF, publishes `flag`); else rows below it (wait for `flag`)
    int32_t flag;           // index of this (panel, step)'s flag
    int32_t mode;           // bit 0: the triangular block has an implicit unit diagonal (LU: U12^T <- U12^T L11^{-T})
    int32_t slot;           // index (within the launch) of the diagonal task whose 16 x 16 inverses the rows use
    int32_t next_b;         // row task, Cholesky: > 0 = these rows are a future diagonal block of the outer block, next_b wide:
                            // push X X^T into it
    int32_t first_col;      // Super[s] (LU pivot records)
    int32_t pad;
};

// skip_diag != 0: entries with row == column are not stored (LU: the L panel)
void launch_build_loadmap(const int64_t* Lp, const int32_t* Li, int32_t n, const int32_t* Super, const int32_t* SuperMap,
                          const int64_t* Lsip, const int32_t* Lsi, const int64_t* Lsxp, int64_t base, int skip_diag, int64_t* map, hipStream_t st);
void launch_load_mapped(const double* Lx, const int64_t* map, int64_t nnz, double* Lsx, hipStream_t st);
void launch_load_panels(const int64_t* Lp, const int32_t* Li, const double* Lx, int32_t n,
                        const int32_t* Super, const int32_t* SuperMap, const int64_t* Lsip, const int32_t* Lsi,
                        const int64_t* Lsxp, double* Lsx, int skip_diag, const int8_t* load_mask, hipStream_t st);
void launch_potrf(const PotrfTask* tasks, int ntasks, double* Lsx, int* info, hipStream_t st);
// LU: the diagonal block is split over two panels, L (strictly lower, at Lsx + task.panel) and U^T (lower
// including the diagonal, at Lsx + task.panel + u_shift)
void launch_getrf(const PotrfTask* tasks, int ntasks, double* Lsx, int64_t u_shift, int* info, PivotCtl pc, hipStream_t st);
// LU: gather the (L, U^T) panel pairs into the reference's (2*nsrow - nscol) x nscol panels (LU/Source/SparseFrame.c:2514-2517):
// values [e_begin, e_end) of that layout -> out[0 .. e_end - e_begin)
// LU download without a gather kernel: PL(R, j) <- PU(j, R) for R <= j inside a supernode's diagonal block, i.e. the L panel
// receives the reference's packed L11 \ U11 (L:2514-2517).  One workgroup per 64 x 64 tile: rows [r0, r0 + 64), columns
// [max(c0, cb), min(c0 + 64, ce)), transposed through LDS.  Nothing reads that part of an L panel (the solves mask it by index).
struct FillTile { int64_t xp; int32_t nsrow, r0, c0, cb, ce; };
void launch_factor_hash(const int32_t* Super, const int64_t* Lsip, const int64_t* Xp, const int64_t* RefXp, int32_t nsuper,
                        const double* PL, const double* PU, int lu, int64_t total, unsigned long long* H, hipStream_t st);
void launch_lu_fill_u11(const FillTile* tiles, int64_t ntiles, double* PL, const double* PU, hipStream_t st);
void launch_pack_lu(const int32_t* Super, const int64_t* Lsip, const int64_t* Xp, const int64_t* RefXp, int32_t nsuper,
                    const double* PL, const double* PU, double* out, int64_t e_begin, int64_t e_end, hipStream_t st);
// pivinv != nullptr (LU with pivoting): the unit tasks (U^T rows) permute their columns by the block's interchanges first
void launch_trsm(const TrsmTask* tasks, int ntasks, double* Lsx, const int32_t* pivinv, hipStream_t st);
// tasks: the diagonal blocks first, then the 64-row tiles below them, any number (the grid need not be co-resident:
// workgroups claim tasks in execution order through *ticket, which must be 0 at launch and is private to the launch);
// flags[task.flag] == epoch once that diagonal block is factored
// lu != 0: the diagonal tasks hold (L panel, U^T panel) and are factored without pivoting
// tinv: scratch for the 16 x 16 inverses, 1024 (Cholesky) / 2048 (LU: of U11^T, then of L11) doubles per diagonal task of the launch
void launch_step(const StepTask* tasks, int ntasks, int lu, double* Lsx, int* flags, int epoch, int* info, double* tinv, int* ticket,
                 PivotCtl pc, hipStream_t st);
// One-time (plan creation): relative maps of all scatter problems [first, first+count) -- the device form of the
// reference's createRelativeMap (cuda_kernel.cu:42-60): RelMap[map_off + ci] = position of source row ci in the
// target supernode's row list.
void launch_build_relmaps(const GemmProb* probs, int nprobs, const int32_t* Lsi, int32_t* RelMap, hipStream_t st);

// kt_prefix[0..ntasks]: running count of 16-deep K steps of the launch's tiles (kt_prefix[ntasks] = total units);
// this call executes the units [u_lo, u_hi) of the launch.  ticket: 8 zeroed counters (one per XCD) for the dynamic deal of
// the whole-tile rounds, or nullptr for the static deal; whole_tiles != 0 (only with u_lo = 0, u_hi = all units): no tile is split by
// units -- an order-independent (bit-reproducible) result, for launches that several ranks execute redundantly
void launch_gemm(const GemmProb* probs, const GemmTask* tasks, const uint32_t* kt_prefix, int ntasks, uint32_t u_lo, uint32_t u_hi,
                 int mode, double* Lsx, const int32_t* RelMap, int* ticket, hipStream_t st, int whole_tiles = 0, int grid_cap = 0);

// Schur updates with K <= SU_MAXK: tasks are SU_TM x SU_TN tiles (GemmTask.tm / .tn in those units), one wave each
void launch_update_small(const GemmProb* probs, const GemmTask* tasks, int ntasks, double* Lsx, const int32_t* RelMap, hipStream_t st);

// ---- device-side supernodal triangular solves with the resident factor (reference: scalar host loops,
// Cholesky/Source/SparseFrame.c:3074-3134).  SV_B-column steps level by level; one launch per step and direction, the
// diagonal solve (one workgroup, wave w = 64-column sub-block w) and the SV_ROWS-row tiles hand over inside the launch.
struct SolveTask {
    int64_t panel;      // doubles, into Lsx
    int64_t rows;       // index into Lsi of the supernode's row list
    int32_t ld, diag, b;    // the step's columns [diag, diag + b), b <= SV_B
    int32_t row0, nrows;    // row tile: rows [row0, row0 + nrows) of the panel (below the block), nrows <= SV_ROWS; nrows == 0: the diagonal task
    int32_t first_col;      // Super[s]
    int32_t flag;           // index into the solve's sync words: forward = "x_blk is solved" flag, backward = tile counter
    int32_t expect;         // backward diagonal task: number of row tiles to wait for
    int64_t tdiag;          // backward diagonal task: 1 + offset (doubles) of the step's diagonal block stored ROW by row (b x b) in the
                            // solve's scratch, 0: none -- the task then reads the block's columns out of the panel
};
// forward launch: tasks = the step's diagonal tasks, then its row tiles; backward launch: the row tiles, then the diagonal
// tasks.  sync: one word per (panel, step) and direction, zero at the start of the solve; ticket: zero, private to the launch
// pivpos != nullptr (LU with pivoting): x_blk is brought into the block's pivot order before the unit-lower solve
// steps whose panels are all narrow (nscol <= 64): one wave per supernode does its diagonal solve and all its rows, no
// hand-off; tasks = the step's diagonal tasks only (task.ld = nsrow, task.b = nscol)
void launch_solve_small_fwd(const SolveTask* t, int nt, const double* Lsx, const int32_t* Lsi, double* x, int unit, const int32_t* pivpos,
                            hipStream_t st);
void launch_solve_small_bwd(const SolveTask* t, int nt, const double* Lsx, const int32_t* Lsi, double* x, hipStream_t st);
// big != 0: some panel of the step has more than one 64-column sub-block (the variant with the prefetch registers)
void launch_solve_fwd(const SolveTask* t, int nt, int big, const double* Lsx, const int32_t* Lsi, double* x, int unit, const int32_t* pivpos,
                      int* sync, int* ticket, int* info, hipStream_t st);
void launch_solve_bwd(const SolveTask* t, int nt, int big, const double* Lsx, const int32_t* Lsi, double* x, int* sync, int* ticket, int* info,
                      hipStream_t st, const double* Tbase = nullptr);
// row-major copies of the diagonal blocks of the backward diagonal tasks list[0 .. ntasks) (indices into `tasks`) into T
void launch_solve_transpose_diag(const SolveTask* tasks, const int64_t* list, int64_t ntasks, const double* Lsx, double* T, hipStream_t st);

void launch_noop(hipStream_t st);

// device twin of SparseFrame_validate's residual (C:3182-3263): b_i = 1 + i/n is written to b, r = A x - b, the four maxima
// |r|_inf, |A|_1, |x|_inf, |b|_inf to norms[0..3] (zero them first).  Up == nullptr: (Lp, Li, Lx) is one triangle used symmetrically.
void launch_residual(const int64_t* Lp, const int32_t* Li, const double* Lx, const int64_t* Up, const int32_t* Ui, const double* Ux,
                     int32_t n, const double* x, double* r, double* colsum, double* b, double* norms, hipStream_t st);

}  // namespace sf

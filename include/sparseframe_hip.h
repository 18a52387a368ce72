/*
 * sparseframe_hip.h -- C ABI of the MI355X-native numeric-factorization core.
 *
 * This library is a drop-in for ONE path of zrjer/Sparse-Matrix-Factorization-Library
 * ("SparseFrame"): the supernodal numeric factorization step and the entry points
 * either side of it.  Everything here is extern "C", plain pointers and sizes; no
 * torch / HIP types appear in a signature.
 *
 * Reference citations:  C: = Cholesky/Source/SparseFrame.c   L: = LU/Source/SparseFrame.c
 *                       I: = Cholesky/Include/info.h          (LU/Include/info.h for LU)
 *
 * Two layers are exported:
 *   (1) the struct-based entry points the reference itself defines (same names,
 *       same argument meaning, same "0 == OK" convention) so that the reference's
 *       driver (C:3323-3467) can link against this library instead of its own
 *       SparseFrame.c for analyze / factorize / solve / validate;
 *   (2) a flat "plan" ABI (sf_chol_* / sf_lu_*) that (1) forwards to and that any FFI
 *       (ctypes, cgo, JNI ...) can bind without knowing the struct layout.
 */
#ifndef SPARSEFRAME_HIP_H
#define SPARSEFRAME_HIP_H

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

/* ---- reference I:70-150 (Cholesky) with the LU additions of LU/Include/info.h:95-117
 * appended at the SAME positions the LU header uses when SF_ABI_LU is defined.
 * The Cholesky library and the LU library of the reference are two separate .so files
 * with two different struct layouts; we keep that split: the default layout below is
 * the Cholesky one, `struct lu_matrix_info_struct` is the LU one. ---- */
struct matrix_info_struct
{
    int serial;
    const char *path;
    FILE *file;
    enum FactorizeType factorizeType;
    int isSymmetric;
    int isComplex;           /* only 0 (real fp64) is supported: reference solve/validate are TODO for complex (C:3086-3100) */
    sf_long ncol;
    sf_long nrow;
    sf_long nzmax;
    sf_long *Tj;  sf_long *Ti;  sf_float *Tx;      /* triplets (consumed by compress, C:526) */
    sf_long *Cp;  sf_long *Ci;  sf_float *Cx;      /* input CSC, one triangle of the symmetric matrix */
    sf_long *Lp;  sf_long *Li;  sf_float *Lx;      /* lower(P A P^T) by column, C:1048-1053 */
    sf_long *LTp; sf_long *LTi; sf_float *LTx;     /* its transpose, C:1055-1060 */
    enum PermMethod permMethod;
    sf_long *Perm;                                  /* Perm[new] = old (C:1006), post-order composed (C:1438) */
    sf_long *Parent;
    sf_long *Post;
    sf_long *ColCount;
    sf_long nsuper;
    sf_long *Super;                                 /* [nsuper+1] first column of each supernode */
    sf_long *SuperMap;                              /* [n] column -> supernode */
    sf_long *Sparent;
    sf_long nsleaf;
    sf_long *LeafQueue;
    sf_long isize;
    sf_long xsize;
    sf_long *Lsip;                                  /* [nsuper+1] */
    sf_long *Lsxp;                                  /* [nsuper+1] */
    sf_long *Lsi;                                   /* [isize] ascending global rows per supernode */
    sf_float *Lsx;                                  /* [xsize] OUTPUT: panel s = nsrow x nscol col-major, lda = nsrow */
    sf_long csize;
    sf_long nstage;
    sf_long *ST_Map;
    sf_long *ST_Pointer;
    sf_long *ST_Index;
    sf_long *ST_Parent;
    size_t *Aoffset;
    size_t *Moffset;
    void *workspace;
    size_t workSize;
    sf_float *Bx;
    sf_float *Xx;
    sf_float *Rx;
    sf_float residual;
    double readTime;
    double analyzeTime;
    double factorizeTime;
    double solveTime;
};

/* =====================================================================================
 * Layer 1: the reference's own entry points (struct based).
 * ===================================================================================== */

/* replaces C:16-285.  Probes HIP devices, creates one handler per device, computes
 * common_info->devSlotSize with the reference formula (C:82-87,199) from the device's
 * total memory.  With no device it leaves numGPU = 0 and takes devSlotSize from the
 * environment variable SF_DEVSLOT (bytes) or 1 GiB; it never divides by numGPU_physical
 * (reference C:38 does and traps with no GPU). */
int SparseFrame_allocate_gpu(struct common_info_struct *common_info, struct gpu_info_struct **gpu_info_list_ptr);
/* replaces C:287-366 */
int SparseFrame_free_gpu(struct common_info_struct *common_info, struct gpu_info_struct **gpu_info_list_ptr);

/* replaces C:589-650 */
int SparseFrame_initialize_matrix(struct matrix_info_struct *matrix_info);
/* replaces C:652-691 (MatrixMarket coordinate real, symmetric or general) */
int SparseFrame_read_matrix(struct matrix_info_struct *matrix_info);
/* in-memory alternative to read_matrix: copies a CSC triangle into Cp/Ci/Cx and sizes
 * the workspace exactly as C:679-684 does.  Not in the reference. */
int SparseFrame_set_matrix_csc(struct matrix_info_struct *matrix_info, sf_long nrow, sf_long nz,
                               const sf_long *Cp, const sf_long *Ci, const sf_float *Cx, int isSymmetric);
/* caller-supplied fill-reducing ordering (Perm[new] = old).  The reference calls
 * METIS_NodeND here (C:1937), which is third-party and unpinned; when no ordering is
 * supplied SparseFrame_analyze uses the identity. */
int SparseFrame_set_perm(struct matrix_info_struct *matrix_info, const sf_long *perm);

/* replaces C:1916-1978: perm -> etree -> postorder -> colcount -> postorder ->
 * analyze_supernodal.  Host only.  Every integer output is bit-exact with the reference
 * for the same Perm and devSlotSize. */
int SparseFrame_analyze(struct common_info_struct *common_info, struct matrix_info_struct *matrix_info);

/* replaces C:3019-3034 / C:2150-3017.  Numeric factorization on the MI355X; output is
 * matrix_info->Lsx on the HOST in the reference layout (lower trapezoid valid).
 * Returns 0 on success, SF_ERR_* otherwise (the reference always returns 0). */
int SparseFrame_factorize(struct common_info_struct *common_info, struct gpu_info_struct *gpu_info_list, struct matrix_info_struct *matrix_info);
int SparseFrame_factorize_supernodal(struct common_info_struct *common_info, struct gpu_info_struct *gpu_info_list, struct matrix_info_struct *matrix_info);

/* replaces C:3036-3139 (host triangular solves, reads Lsx/Bx, writes Xx) */
int SparseFrame_solve_supernodal(struct matrix_info_struct *matrix_info);
/* replaces C:3141-3266 (b_i = 1 + i/n, residual |Ax-b|_inf / (|A|_1 |x|_inf + |b|_inf)) */
int SparseFrame_validate(struct matrix_info_struct *matrix_info);
/* replaces C:3268-3321 */
int SparseFrame_cleanup_matrix(struct matrix_info_struct *matrix_info);

/* =====================================================================================
 * Layer 2: flat ABI.
 * ===================================================================================== */

#define SF_OK                0
#define SF_ERR_ARG           1
#define SF_ERR_NO_DEVICE     2   /* no HIP device / HIP runtime error */
#define SF_ERR_ALLOC         3
#define SF_ERR_NOT_POSDEF    4   /* non-positive pivot met in a diagonal block */
#define SF_ERR_HIP           5

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
/* device pointer of the resident factor (for device-side consumers) */
void *sf_chol_plan_factor_device_ptr(sf_chol_plan *plan);
/* device-side supernodal solve with the resident factor: x <- (L L^T)^{-1} b, permuted space */
int sf_chol_plan_solve(sf_chol_plan *plan, const sf_float *b_host, sf_float *x_host);
/* statistics: "levels","launches","gemm_tasks","update_pairs","flops_exec","flops_update",
 * "scatter_elems","bytes_device","last_ms" (device time of the last factorize, HIP events),
 * "last_update_ms","last_panel_ms","last_load_ms" (only when profiling is on) */
double sf_chol_plan_stat(const sf_chol_plan *plan, const char *name);
/* 1 -> record HIP events around each phase of the next factorize calls */
int sf_chol_plan_set_profiling(sf_chol_plan *plan, int on);
int sf_chol_plan_destroy(sf_chol_plan *plan);

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
int sf_lu_plan_set_values(sf_lu_plan *plan, const sf_float *Lx, const sf_float *Ux /* NULL if U aliases L */);
int sf_lu_plan_factorize(sf_lu_plan *plan, int sync);
int sf_lu_plan_sync(sf_lu_plan *plan);
/* D2H copy of the factor gathered into the reference layout: panel s = (2*nsrow-nscol) x nscol column-major,
 * rows [0,nscol) packed L11\U11, [nscol,nsrow) L21, [nsrow,2*nsrow-nscol) U12^T (L:2514-2517) */
int sf_lu_plan_get_factor(sf_lu_plan *plan, sf_float *Lsx);
double sf_lu_plan_stat(const sf_lu_plan *plan, const char *name);
int sf_lu_plan_set_profiling(sf_lu_plan *plan, int on);
int sf_lu_plan_destroy(sf_lu_plan *plan);

/* number of HIP devices visible (0 on a CPU-only box; never fails) */
int sf_device_count(void);
/* version string */
const char *sf_version(void);
/* layout probe for FFI authors: "sizeof_common", "sizeof_matrix", "offsetof_Lsx", "offsetof_workspace",
 * "offsetof_residual", "offsetof_devSlotSize" as this library was compiled; -1 for an unknown name */
long sf_abi_layout(const char *name);

#ifdef __cplusplus
}
#endif
#endif /* SPARSEFRAME_HIP_H */

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
 *   (2) a flat "plan" ABI (sparseframe_flat.h: sf_symbolic_*, sf_chol_plan_*, sf_lu_plan_*) that (1)
 *       forwards to and that any FFI (ctypes, cgo, JNI ...) can bind without knowing the struct layout.
 * This header is the CHOLESKY struct layout; the LU library of the reference is a separate .so with a
 * different struct (LU/Include/info.h): see sparseframe_lu_hip.h / libsparseframe_lu_hip.so.
 */
#ifndef SPARSEFRAME_HIP_H
#define SPARSEFRAME_HIP_H

#include "sparseframe_flat.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference I:70-150 (Cholesky layout).  Field order and types are the ABI. ---- */
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
 * supplied SparseFrame_analyze orders with the built-in nested dissection (permMethod = PERM_METIS is what
 * SparseFrame_initialize_matrix sets, as the reference always orders); perm == NULL here selects the natural order. */
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

/* replaces C:368-398 (LU: the same functions of LU/Source/SparseFrame.c): the driver's list of matrixThreadNum matrix_info objects */
int SparseFrame_allocate_matrix(struct common_info_struct *common_info, struct matrix_info_struct **matrix_info_list_ptr);
int SparseFrame_free_matrix(struct common_info_struct *common_info, struct matrix_info_struct **matrix_info_list_ptr);
/* replaces C:3323-3467, the reference's public entry point (Include/SparseFrame.h:43; Demo/demo.c calls it): argv[1..] are
 * MatrixMarket files; each is read, analysed, factorized on the GPU(s), solved and validated by one of up to two matrix threads
 * that share the handler list; prints the reference's report lines.  Returns the number of matrices that failed (the reference: 0). */
int SparseFrame(int argc, char **argv);

#ifdef __cplusplus
}
#endif
#endif /* SPARSEFRAME_HIP_H */

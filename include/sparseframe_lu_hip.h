/*
 * sparseframe_lu_hip.h -- struct-based entry points of the LU library (libsparseframe_lu_hip.so).
 *
 * The reference builds Cholesky/ and LU/ into two separate libSparseFrame.so files that export the SAME function
 * names over DIFFERENT `struct matrix_info_struct` layouts (LU/Include/info.h adds the U arrays).  This header is
 * the LU layout; do not include it together with sparseframe_hip.h.  The flat ABI both forward to is in
 * sparseframe_flat.h (sf_symbolic_create_lu, sf_lu_plan_*).
 * The reference's LU never pivots (L:2653, L:3344): inputs must be factorizable without pivoting.
 */
#ifndef SPARSEFRAME_LU_HIP_H
#define SPARSEFRAME_LU_HIP_H

#include "sparseframe_flat.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference LU/Include/info.h:70-163 (LU layout: adds nzCPCT/CPCTp/CPCTi, Up..UTx, PivInv). ---- */
struct matrix_info_struct
{
    int serial;
    const char *path;
    FILE *file;
    enum FactorizeType factorizeType;
    int isSymmetric;
    int isComplex;           /* only 0 (real fp64): the reference's LU solve/validate are TODO for complex (L:3646-3693) */
    sf_long ncol;
    sf_long nrow;
    sf_long nzmax;
    sf_long *Tj;  sf_long *Ti;  sf_float *Tx;
    sf_long *Cp;  sf_long *Ci;  sf_float *Cx;      /* symmetric: one triangle; otherwise the whole matrix in CSC */
    sf_long nzCPCT;                                 /* pattern of C + C^T for the ordering (L:2254-2396): not produced */
    sf_long *CPCTp;                                 /*   here, the ordering is supplied by the caller               */
    sf_long *CPCTi;
    sf_long *Lp;  sf_long *Li;  sf_float *Lx;      /* L part of P A P^T by column, rows i >= j (L:1178-1196) */
    sf_long *LTp; sf_long *LTi; sf_float *LTx;
    sf_long *Up;  sf_long *Ui;  sf_float *Ux;      /* U part by ROW, columns j >= i (L:1198-1215); NULL if isSymmetric */
    sf_long *UTp; sf_long *UTi; sf_float *UTx;
    enum PermMethod permMethod;
    sf_long *PivInv;                                /* static pivoting is compiled out in the reference (L:784-787) */
    sf_long *Perm;
    sf_long *Parent;
    sf_long *Post;
    sf_long *ColCount;
    sf_long nsuper;
    sf_long *Super;
    sf_long *SuperMap;
    sf_long *Sparent;
    sf_long nsleaf;
    sf_long *LeafQueue;
    sf_long isize;
    sf_long xsize;
    sf_long *Lsip;
    sf_long *Lsxp;                                  /* panel s holds nscol * (2*nsrow - nscol) values (L:1946) */
    sf_long *Lsi;
    sf_float *Lsx;                                  /* OUTPUT: rows [0,nscol) L11\\U11, [nscol,nsrow) L21, [nsrow,2nsrow-nscol) U12^T */
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

/* replaces L:16-285.  Probes HIP devices, creates one handler per device, computes
 * common_info->devSlotSize with the reference formula (L:82-87,199) from the device's
 * total memory.  With no device it leaves numGPU = 0 and takes devSlotSize from the
 * environment variable SF_DEVSLOT (bytes) or 1 GiB; it never divides by numGPU_physical
 * (reference L:38 does and traps with no GPU). */
int SparseFrame_allocate_gpu(struct common_info_struct *common_info, struct gpu_info_struct **gpu_info_list_ptr);
/* replaces L:287-366 */
int SparseFrame_free_gpu(struct common_info_struct *common_info, struct gpu_info_struct **gpu_info_list_ptr);

/* replaces L:675-746 */
int SparseFrame_initialize_matrix(struct matrix_info_struct *matrix_info);
/* replaces L:748-792 (MatrixMarket coordinate real, symmetric or general) */
int SparseFrame_read_matrix(struct matrix_info_struct *matrix_info);
/* in-memory alternative to read_matrix: copies a CSC triangle into Cp/Ci/Cx and sizes
 * the workspace exactly as L:775-780 does.  Not in the reference. */
int SparseFrame_set_matrix_csc(struct matrix_info_struct *matrix_info, sf_long nrow, sf_long nz,
                               const sf_long *Cp, const sf_long *Ci, const sf_float *Cx, int isSymmetric);
/* caller-supplied fill-reducing ordering (Perm[new] = old).  The reference calls
 * METIS_NodeND here (L:2398), which is third-party and unpinned; when no ordering is
 * supplied SparseFrame_analyze orders with the built-in nested dissection (permMethod = PERM_METIS is what
 * SparseFrame_initialize_matrix sets, as the reference always orders); perm == NULL here selects the natural order. */
int SparseFrame_set_perm(struct matrix_info_struct *matrix_info, const sf_long *perm);

/* replaces L:2233-2458: perm -> etree -> postorder -> colcount -> postorder ->
 * analyze_supernodal.  Host only.  Every integer output is bit-exact with the reference
 * for the same Perm and devSlotSize. */
int SparseFrame_analyze(struct common_info_struct *common_info, struct matrix_info_struct *matrix_info);

/* replaces L:3575-3590 / L:2668-3573.  Numeric factorization on the MI355X; output is
 * matrix_info->Lsx on the HOST in the reference layout (lower trapezoid valid).
 * Returns 0 on success, SF_ERR_* otherwise (the reference always returns 0). */
int SparseFrame_factorize(struct common_info_struct *common_info, struct gpu_info_struct *gpu_info_list, struct matrix_info_struct *matrix_info);
int SparseFrame_factorize_supernodal(struct common_info_struct *common_info, struct gpu_info_struct *gpu_info_list, struct matrix_info_struct *matrix_info);

/* Pivoting -- NOT in the reference, which never pivots (magma_dgetrf_nopiv L:2653, devIpiv = NULL L:3344; its static pre-pivot
 * L:589-673 is compiled out at L:784).  DEFAULT = the reference's behaviour: no interchanges, no perturbation, matrix_info->Lsx is
 * the reference's factor and matrix_info->PivInv the identity.  SparseFrame_set_pivoting(tol, perturb) (process-wide default, before
 * SparseFrame_factorize; or SF_LU_PIVOT_TOL in the environment) opts in to threshold partial pivoting inside the 64 x 64 diagonal
 * blocks of a supernode (tol in (0, 1], 1 = partial pivoting) and to the replacement of pivots below perturb * max|a_ij| by that
 * value.  CONTRACT when it is on: PivInv[g] = the row position original row g was given (same 64-column block; identity where
 * nothing moved); a row's entries in its block's own columns and to the right moved with it, the L entries LEFT of the block did
 * not -- a consumer of Lsx must apply the interchanges block by block in its forward sweep (LINPACK-style), as
 * SparseFrame_solve_supernodal here does; a consumer that ignores PivInv gets a wrong x whenever PivInv is not the identity.
 * SparseFrame_perturbed_pivots: how many pivots the last factorization into this matrix_info's Lsx replaced (0 = the factor is
 * exact; > 0: refine the solution iteratively; -1 unknown). */
int SparseFrame_set_pivoting(double tol, double perturb);
/* the same for ONE matrix_info (overrides the process-wide setting for that matrix; dropped by SparseFrame_initialize_matrix /
 * _cleanup_matrix).  The reference's driver factorizes MATRIX_THREAD_NUM matrices at a time over one handler list (L:3375): each
 * may carry its own policy; a matrix without one gets the process-wide setting, else the reference's behaviour. */
int SparseFrame_set_matrix_pivoting(struct matrix_info_struct *matrix_info, double tol, double perturb);
int SparseFrame_clear_matrix_pivoting(struct matrix_info_struct *matrix_info);
sf_long SparseFrame_perturbed_pivots(const struct matrix_info_struct *matrix_info);

/* replaces L:3592-3700 (host triangular solves, reads Lsx/Bx, writes Xx) */
int SparseFrame_solve_supernodal(struct matrix_info_struct *matrix_info);
/* replaces L:3702-3858 (b_i = 1 + i/n, residual |Ax-b|_inf / (|A|_1 |x|_inf + |b|_inf)) */
int SparseFrame_validate(struct matrix_info_struct *matrix_info);
/* replaces L:3860-3922 */
int SparseFrame_cleanup_matrix(struct matrix_info_struct *matrix_info);

/* layout probe, as sf_abi_layout() of the Cholesky library */
long sf_lu_abi_layout(const char *name);

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
#endif /* SPARSEFRAME_LU_HIP_H */

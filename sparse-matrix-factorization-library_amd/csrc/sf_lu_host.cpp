// libsparseframe_lu_hip.so: the reference's struct-based entry points over the LU layout of matrix_info_struct
// (LU/Include/info.h), forwarding to the flat ABI of libsparseframe_hip.so (sf_symbolic_create_lu, sf_lu_plan_*).
// Reference file: LU/Source/SparseFrame.c ("L:").
#include "sf_host_solve.h"
#include <sparseframe_lu_hip.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <mutex>
#include <thread>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <string>
#include <time.h>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

double wall_seconds() {
    struct timespec tp;
    clock_gettime(CLOCK_REALTIME, &tp);
    return tp.tv_sec + (double)tp.tv_nsec / 1.0e9;
}

void free_and_null(void** p) {
    if (*p) free(*p);
    *p = nullptr;
}
#define SF_FREE(field) free_and_null((void**)&(mi->field))

// Returning a factor of tens of GB to the system (free -> munmap: the kernel tears down millions of page-table entries) takes 0.85 s for
// the 30 GB of 128^3 -- as long as the factorization.  Nobody waits for it: big blocks are released by a detached thread.
void free_big_async(void** p, size_t bytes) {
    if (!p || !*p) return;
    void* q = *p;
    *p = nullptr;
    if (bytes < ((size_t)1 << 30)) { free(q); return; }
    try { std::thread([q] { free(q); }).detach(); } catch (...) { free(q); }
}

sf_long* dup_long(const sf_symbolic* S, const char* name, sf_long min_len = 1) {
    sf_long len = 0;
    const sf_long* src = sf_symbolic_long_array(S, name, &len);
    sf_long* p = (sf_long*)calloc((size_t)(len > min_len ? len : min_len), sizeof(sf_long));
    if (p && src && len > 0) memcpy(p, src, (size_t)len * sizeof(sf_long));
    return p;
}
sf_float* dup_float(const sf_symbolic* S, const char* name) {
    sf_long len = 0;
    const sf_float* src = sf_symbolic_float_array(S, name, &len);
    sf_float* p = (sf_float*)calloc((size_t)(len > 1 ? len : 1), sizeof(sf_float));
    if (p && src && len > 0) memcpy(p, src, (size_t)len * sizeof(sf_float));
    return p;
}

}  // namespace

extern "C" {

long sf_lu_abi_layout(const char* name) {
    if (!name) return -1;
    const std::string k(name);
    if (k == "sizeof_common") return (long)sizeof(struct common_info_struct);
    if (k == "sizeof_matrix") return (long)sizeof(struct matrix_info_struct);
    if (k == "offsetof_Lsx") return (long)offsetof(struct matrix_info_struct, Lsx);
    if (k == "offsetof_Up") return (long)offsetof(struct matrix_info_struct, Up);
    if (k == "offsetof_workspace") return (long)offsetof(struct matrix_info_struct, workspace);
    if (k == "offsetof_residual") return (long)offsetof(struct matrix_info_struct, residual);
    return -1;
}

// the handler list is the main library's (sf_handlers.hip): one handler per device, plan cache, overlapped copy-back
int SparseFrame_allocate_gpu(struct common_info_struct* common, struct gpu_info_struct** list) { return sf_handlers_allocate(common, list); }   // L:16-285
int SparseFrame_free_gpu(struct common_info_struct* common, struct gpu_info_struct** list) { return sf_handlers_free(common, list); }       // L:287-366

namespace {
// per-matrix_info pivoting policy (SparseFrame_set_matrix_pivoting): side table keyed by the struct's address
std::mutex g_mi_piv_mu;
std::unordered_map<const void*, std::pair<double, double>> g_mi_piv;
}

int SparseFrame_initialize_matrix(struct matrix_info_struct* mi) {   // L:675-746
    if (!mi) return 1;
    { std::lock_guard<std::mutex> g(g_mi_piv_mu); g_mi_piv.erase((const void*)mi); }
    const int serial = mi->serial;
    const char* path = mi->path;
    memset(mi, 0, sizeof(*mi));
    mi->serial = serial;
    mi->path = path;
    mi->factorizeType = TYPE_LU;
    // as the Cholesky library: ordered by default (the reference always calls METIS, L:2271), identity by opt-in
    mi->permMethod = PERM_METIS;
    return 0;
}

int SparseFrame_set_matrix_csc(struct matrix_info_struct* mi, sf_long nrow, sf_long nz,
                               const sf_long* Cp, const sf_long* Ci, const sf_float* Cx, int isSymmetric) {
    if (!mi || nrow < 0 || nz < 0 || !Cp || (nz > 0 && (!Ci || !Cx))) return 1;
    if (Cp[0] != 0 || Cp[nrow] != nz) return 1;
    SF_FREE(Cp); SF_FREE(Ci); SF_FREE(Cx); SF_FREE(workspace);
    mi->isSymmetric = isSymmetric;
    mi->isComplex = 0;
    mi->nrow = mi->ncol = nrow;
    mi->nzmax = nz;
    mi->Cp = (sf_long*)malloc((nrow + 1) * sizeof(sf_long));
    mi->Ci = (sf_long*)malloc((nz > 0 ? nz : 1) * sizeof(sf_long));
    mi->Cx = (sf_float*)malloc((nz > 0 ? nz : 1) * sizeof(sf_float));
    mi->workSize = (size_t)(10 * nrow + (2 * nz - nrow > 0 ? 2 * nz - nrow : 0) + 1) * sizeof(sf_long);   // L:775-780
    mi->workspace = malloc(mi->workSize);
    if (!mi->Cp || !mi->Ci || !mi->Cx || !mi->workspace) return 1;
    memcpy(mi->Cp, Cp, (nrow + 1) * sizeof(sf_long));
    if (nz > 0) {
        memcpy(mi->Ci, Ci, nz * sizeof(sf_long));
        memcpy(mi->Cx, Cx, nz * sizeof(sf_float));
    }
    return 0;
}

// MatrixMarket coordinate real {symmetric|general}; explicit zeros dropped (L:496); counting-sort compress (L:526-587)
int SparseFrame_read_matrix(struct matrix_info_struct* mi) {
    if (!mi || !mi->path) return 1;
    const double t0 = wall_seconds();
    FILE* f = fopen(mi->path, "r");
    if (!f) return 1;
    char* line = nullptr;
    size_t cap = 0;
    int rc = 1, symmetric = 0;
    long nrow = 0, ncol = 0, nzmax = 0;
    std::vector<sf_long> Ti, Tj;
    std::vector<double> Tx;
    do {
        ssize_t got;
        while ((got = getline(&line, &cap, f)) != -1 && line[0] == '\n') {}
        if (got == -1 || strncmp(line, "%%MatrixMarket", 14) != 0) break;
        char s0[64], s1[64], s2[64], s3[64], s4[64];
        s3[0] = s4[0] = 0;
        sscanf(line, "%63s %63s %63s %63s %63s", s0, s1, s2, s3, s4);
        symmetric = strcmp(s4, "symmetric") == 0;
        if (strcmp(s3, "real") != 0 && strcmp(s3, "integer") != 0) break;
        while ((got = getline(&line, &cap, f)) != -1 && (line[0] == '%' || line[0] == '\n')) {}
        if (got == -1 || sscanf(line, "%ld %ld %ld", &nrow, &ncol, &nzmax) != 3) break;
        if (nrow != ncol || nrow < 0 || nzmax < 0) break;
        bool bad = false;
        while (getline(&line, &cap, f) != -1) {
            if (line[0] == '\n' || line[0] == 0) continue;
            long i, j; double x;
            {   // "%ld %ld %lg" by hand: strtol / strtod are several times faster than sscanf on files of 10^7 lines
                char *e1, *e2, *e3;
                i = strtol(line, &e1, 10);
                j = strtol(e1, &e2, 10);
                x = strtod(e2, &e3);
                if (e1 == line || e2 == e1 || e3 == e2) { bad = true; break; }
            }
            if (x != 0) {
                if ((long)Tx.size() >= nzmax || i < 1 || j < 1 || i > nrow || j > ncol) { bad = true; break; }
                Ti.push_back(i - 1); Tj.push_back(j - 1); Tx.push_back(x);
            }
        }
        if (!bad) rc = 0;
    } while (0);
    free(line);
    fclose(f);
    if (rc) return rc;
    const sf_long nz = (sf_long)Tx.size();
    std::vector<sf_long> Cp(ncol + 1, 0), Ci(nz > 0 ? nz : 1);
    std::vector<double> Cx(nz > 0 ? nz : 1);
    for (sf_long k = 0; k < nz; ++k) Cp[Tj[k] + 1]++;
    for (sf_long j = 0; j < ncol; ++j) Cp[j + 1] += Cp[j];
    std::vector<sf_long> fill(Cp.begin(), Cp.end() - 1);
    for (sf_long k = 0; k < nz; ++k) { const sf_long p = fill[Tj[k]]++; Ci[p] = Ti[k]; Cx[p] = Tx[k]; }
    rc = SparseFrame_set_matrix_csc(mi, nrow, nz, Cp.data(), Ci.data(), Cx.data(), symmetric);
    mi->readTime = wall_seconds() - t0;
    return rc;
}

int SparseFrame_set_perm(struct matrix_info_struct* mi, const sf_long* perm) {
    if (!mi || mi->nrow <= 0) return 1;
    SF_FREE(Perm);
    if (!perm) { mi->permMethod = PERM_IDENTITY; return 0; }
    mi->Perm = (sf_long*)malloc(mi->nrow * sizeof(sf_long));
    if (!mi->Perm) return 1;
    memcpy(mi->Perm, perm, mi->nrow * sizeof(sf_long));
    mi->permMethod = PERM_METIS;
    return 0;
}

int SparseFrame_analyze(struct common_info_struct* common, struct matrix_info_struct* mi) {   // L:2233-2458
    if (!common || !mi || !mi->Cp) return 1;
    const double t0 = wall_seconds();
    sf_symbolic* S = nullptr;
    const sf_long* perm = (mi->permMethod != PERM_IDENTITY) ? mi->Perm : nullptr;
    std::vector<sf_long> builtin;
    if (mi->permMethod != PERM_IDENTITY && !mi->Perm) {
        // built-in nested dissection of the pattern of A + A^T (stands in for METIS_NodeND, L:2271)
        builtin.resize(mi->nrow > 0 ? mi->nrow : 1);
        if (sf_graph_nd_perm(mi->nrow, mi->Cp, mi->Ci, 64, builtin.data())) return 1;
        perm = builtin.data();
    }
    if (sf_symbolic_create_lu(&S, mi->nrow, mi->Cp, mi->Ci, mi->Cx, perm, common->devSlotSize, mi->isSymmetric)) return 1;

    SF_FREE(Lp); SF_FREE(Li); SF_FREE(Lx); SF_FREE(LTp); SF_FREE(LTi); SF_FREE(LTx);
    SF_FREE(Up); SF_FREE(Ui); SF_FREE(Ux); SF_FREE(UTp); SF_FREE(UTi); SF_FREE(UTx);
    SF_FREE(Perm); SF_FREE(Parent); SF_FREE(Post); SF_FREE(ColCount);
    SF_FREE(Super); SF_FREE(SuperMap); SF_FREE(Sparent); SF_FREE(LeafQueue);
    SF_FREE(Lsip); SF_FREE(Lsxp); SF_FREE(Lsi); SF_FREE(Lsx);
    SF_FREE(ST_Map); SF_FREE(ST_Pointer); SF_FREE(ST_Index); SF_FREE(Aoffset); SF_FREE(Moffset);
    SF_FREE(PivInv);            // sized by nrow at the next factorization (a re-analysis may follow a read of a larger matrix)

    mi->Lp = dup_long(S, "Lp"); mi->Li = dup_long(S, "Li"); mi->Lx = dup_float(S, "Lx");
    mi->LTp = dup_long(S, "LTp"); mi->LTi = dup_long(S, "LTi"); mi->LTx = dup_float(S, "LTx");
    if (!mi->isSymmetric) {
        mi->Up = dup_long(S, "Up"); mi->Ui = dup_long(S, "Ui"); mi->Ux = dup_float(S, "Ux");
        mi->UTp = dup_long(S, "UTp"); mi->UTi = dup_long(S, "UTi"); mi->UTx = dup_float(S, "UTx");
    }
    mi->Perm = dup_long(S, "Perm");
    mi->Parent = dup_long(S, "Parent");
    mi->Post = dup_long(S, "Post");
    mi->ColCount = dup_long(S, "ColCount");
    mi->nsuper = sf_symbolic_scalar(S, "nsuper");
    mi->Super = dup_long(S, "Super", mi->nrow + 1);      // the reference allocates nrow+1 / nrow entries (L:2449-2451)
    mi->SuperMap = dup_long(S, "SuperMap");
    mi->Sparent = dup_long(S, "Sparent", mi->nrow);
    mi->nsleaf = sf_symbolic_scalar(S, "nsleaf");
    mi->LeafQueue = dup_long(S, "LeafQueue");
    mi->isize = sf_symbolic_scalar(S, "isize");
    mi->xsize = sf_symbolic_scalar(S, "xsize");
    mi->Lsip = dup_long(S, "Lsip"); mi->Lsxp = dup_long(S, "Lsxp"); mi->Lsi = dup_long(S, "Lsi");
    mi->Lsx = (sf_float*)malloc((size_t)(mi->xsize > 0 ? mi->xsize : 1) * sizeof(sf_float));
    mi->csize = sf_symbolic_scalar(S, "csize");
    mi->nstage = sf_symbolic_scalar(S, "nstage");
    mi->ST_Map = dup_long(S, "ST_Map"); mi->ST_Pointer = dup_long(S, "ST_Pointer"); mi->ST_Index = dup_long(S, "ST_Index");
    mi->ST_Parent = nullptr;
    mi->Aoffset = (size_t*)dup_long(S, "Aoffset");      // size_t and int64_t have the same size and non-negative values
    mi->Moffset = (size_t*)dup_long(S, "Moffset");
    sf_symbolic_destroy(S);
    mi->analyzeTime = wall_seconds() - t0;
    return mi->Lsx ? 0 : 1;
}

int SparseFrame_factorize_supernodal(struct common_info_struct* common, struct gpu_info_struct* list,
                                     struct matrix_info_struct* mi) {   // L:2668-3573
    if (!common || !mi || !mi->Lsx) return SF_ERR_ARG;
    // PivInv (LU/Include/info.h: the field of the reference's disabled static pre-pivot, L:589-673) carries the record of the
    // in-block interchanges: PivInv[g] = row position of original row g (same 64-column block); identity when nothing moved
    if (!mi->PivInv) mi->PivInv = (sf_long*)malloc((size_t)(mi->nrow > 0 ? mi->nrow : 1) * sizeof(sf_long));
    if (!mi->PivInv) return SF_ERR_ALLOC;
    for (sf_long j = 0; j < mi->nrow; ++j) mi->PivInv[j] = j;
    {
        std::lock_guard<std::mutex> g(g_mi_piv_mu);
        auto it = g_mi_piv.find((const void*)mi);
        if (it != g_mi_piv.end()) (void)sf_handlers_set_lu_pivoting_next_call(it->second.first, it->second.second);
    }
    return sf_handlers_factorize(common, list, 1, mi->serial, mi->nrow, mi->nsuper, mi->Super, mi->SuperMap, mi->Lsip, mi->Lsi,
                                 mi->Lsxp, mi->Lp, mi->Li, mi->isSymmetric ? nullptr : mi->Up, mi->isSymmetric ? nullptr : mi->Ui,
                                 mi->Lx, mi->isSymmetric ? nullptr : mi->Ux, mi->Lsx, mi->PivInv);
}

// Pivoting is opt-in: by default the factor is the reference's (no interchanges, PivInv = identity, nothing perturbed).
// SparseFrame_set_pivoting: the process-wide default of the struct path.  SparseFrame_set_matrix_pivoting: ONE matrix_info's own
// setting (the reference's driver runs MATRIX_THREAD_NUM matrices at a time over one handler list, L:3375: two matrices may want
// different policies).  matrix_info_struct is the reference's layout and has no field for it, so the setting lives in a side table
// keyed by the struct's address; SparseFrame_initialize_matrix / _cleanup_matrix drop it.
int SparseFrame_set_pivoting(double tol, double perturb) { return sf_handlers_set_lu_pivoting(tol, perturb); }
int SparseFrame_set_matrix_pivoting(struct matrix_info_struct* mi, double tol, double perturb) {
    if (!mi || !(tol >= 0.0) || tol > 1.0 || !(perturb >= 0.0)) return SF_ERR_ARG;
    std::lock_guard<std::mutex> g(g_mi_piv_mu);
    g_mi_piv[(const void*)mi] = {tol, perturb};
    return SF_OK;
}
int SparseFrame_clear_matrix_pivoting(struct matrix_info_struct* mi) {
    std::lock_guard<std::mutex> g(g_mi_piv_mu);
    g_mi_piv.erase((const void*)mi);
    return SF_OK;
}
sf_long SparseFrame_perturbed_pivots(const struct matrix_info_struct* mi) {
    return (mi && mi->Lsx) ? (sf_long)sf_handlers_perturbed_pivots(mi->Lsx) : -1;
}

int SparseFrame_factorize(struct common_info_struct* common, struct gpu_info_struct* list, struct matrix_info_struct* mi) {
    const double t0 = wall_seconds();
    const int rc = SparseFrame_factorize_supernodal(common, list, mi);
    if (mi) mi->factorizeTime = wall_seconds() - t0;
    return rc;
}

// unit-lower forward substitution, then backward with U11 (packed in the diagonal block, accessed transposed)
// and the U12^T block at row offset nsrow - nscol (L:3592-3700)
int SparseFrame_solve_supernodal(struct matrix_info_struct* mi) {
    if (!mi || !mi->Lsx || !mi->Bx || !mi->Xx) return 1;
    const double t0 = wall_seconds();
    // the factor SparseFrame_factorize copied into Lsx is normally still resident in the handler's plan: solve there (two sweeps
    // over the factor in HBM instead of host memory).  Falls through to the reference's host solve when it is not (several
    // handlers, plan evicted or re-used, Lsx changed by the caller, SF_SOLVE=host).
    if (sf_handlers_solve_resident_sym(mi->Lsx, mi->Bx, mi->Xx, 1, mi->nrow, mi->nsuper, mi->Super, mi->SuperMap, mi->Lsip, mi->Lsi,
                                       mi->Lsxp, mi->Lp, mi->Li, mi->isSymmetric ? nullptr : mi->Up, mi->isSymmetric ? nullptr : mi->Ui) == SF_OK) {
        mi->solveTime = wall_seconds() - t0;
        return 0;
    }
    double* x = mi->Xx;
    memcpy(x, mi->Bx, mi->nrow * sizeof(double));
    // a large factor: the same two sweeps on several threads (sf_host_solve.h); the scalar sweep below is the reference's (L:3592-3700)
    if (const int T = sf_host_solve::threads_for((sf_host_solve::Long)mi->xsize); T > 1) {
        std::vector<int32_t> owner((size_t)(mi->nsuper > 0 ? mi->nsuper : 1), 0);
        // (the top is shared by all threads: a top flop costs 1 / T of a subtree flop, plus the barriers)
        if (sf_subtree_partition_weighted(mi->nsuper, mi->Super, mi->SuperMap, mi->Lsip, mi->Lsi, T, 1.0 / T + 0.05, owner.data(), nullptr, nullptr) == SF_OK) {
            sf_host_solve::solve_parallel<true>(mi->nrow, mi->nsuper, mi->Super, mi->SuperMap, mi->Lsip, mi->Lsi, mi->Lsxp, mi->Lsx, mi->PivInv,
                                                owner.data(), T, x);
            mi->solveTime = wall_seconds() - t0;
            return 0;
        }
    }
    for (sf_long s = 0; s < mi->nsuper; ++s) {
        const sf_long nscol = mi->Super[s + 1] - mi->Super[s], nsrow = mi->Lsip[s + 1] - mi->Lsip[s], lda = 2 * nsrow - nscol;
        const sf_long* rows = mi->Lsi + mi->Lsip[s];
        const double* P = mi->Lsx + mi->Lsxp[s];
        for (sf_long c = 0; c < nscol; ++c) {
            if (mi->PivInv && c % 64 == 0) {
                // the row interchanges of this 64-column block (restricted to its diagonal block), applied as the sweep
                // reaches it: the L entries to the left of the block were stored at the rows' original places
                const sf_long c0 = mi->Super[s] + c, bw = (nscol - c < 64) ? nscol - c : 64;
                double tmp[64];
                bool moved = false;
                for (sf_long k = 0; k < bw; ++k) moved = moved || mi->PivInv[c0 + k] != c0 + k;
                if (moved) {
                    for (sf_long k = 0; k < bw; ++k) tmp[mi->PivInv[c0 + k] - c0] = x[c0 + k];
                    for (sf_long k = 0; k < bw; ++k) x[c0 + k] = tmp[k];
                }
            }
            const double xj = x[rows[c]];
            const double* col = P + c * lda;
            for (sf_long r = c + 1; r < nsrow; ++r) x[rows[r]] -= col[r] * xj;
        }
    }
    for (sf_long s = mi->nsuper - 1; s >= 0; --s) {
        const sf_long nscol = mi->Super[s + 1] - mi->Super[s], nsrow = mi->Lsip[s + 1] - mi->Lsip[s], lda = 2 * nsrow - nscol;
        const sf_long* rows = mi->Lsi + mi->Lsip[s];
        const double* P = mi->Lsx + mi->Lsxp[s];
        for (sf_long c = nscol - 1; c >= 0; --c) {
            double acc = x[rows[c]];
            for (sf_long r = c + 1; r < nscol; ++r) acc -= P[r * lda + c] * x[rows[r]];                        // U11(c,r)
            for (sf_long r = nscol; r < nsrow; ++r) acc -= P[(nsrow - nscol) + c * lda + r] * x[rows[r]];       // U12^T
            x[rows[c]] = acc / P[c * lda + c];
        }
    }
    mi->solveTime = wall_seconds() - t0;
    return 0;
}

int SparseFrame_validate(struct matrix_info_struct* mi) {   // L:3702-3858
    if (!mi || !mi->Lp || !mi->Lsx) return 1;
    const sf_long n = mi->nrow;
    SF_FREE(Bx); SF_FREE(Xx); SF_FREE(Rx);
    mi->Bx = (double*)malloc((n > 0 ? n : 1) * sizeof(double));
    mi->Xx = (double*)malloc((n > 0 ? n : 1) * sizeof(double));
    mi->Rx = (double*)malloc((n > 0 ? n : 1) * sizeof(double));
    if (!mi->Bx || !mi->Xx || !mi->Rx) return 1;
    for (sf_long i = 0; i < n; ++i) mi->Bx[i] = 1 + i / (double)n;
    if (SparseFrame_solve_supernodal(mi)) return 1;
    const sf_long* Up = mi->isSymmetric ? mi->Lp : mi->Up;
    const sf_long* Ui = mi->isSymmetric ? mi->Li : mi->Ui;
    const double* Ux = mi->isSymmetric ? mi->Lx : mi->Ux;
    std::vector<double> colsum(n, 0.0);
    for (sf_long i = 0; i < n; ++i) mi->Rx[i] = -mi->Bx[i];
    for (sf_long j = 0; j < n; ++j) {
        for (sf_long p = mi->Lp[j]; p < mi->Lp[j + 1]; ++p) { mi->Rx[mi->Li[p]] += mi->Lx[p] * mi->Xx[j]; colsum[j] += std::fabs(mi->Lx[p]); }
        for (sf_long p = Up[j]; p < Up[j + 1]; ++p) {
            const sf_long i = Ui[p];
            if (i != j) { mi->Rx[j] += Ux[p] * mi->Xx[i]; colsum[i] += std::fabs(Ux[p]); }
        }
    }
    double anorm = 0, bnorm = 0, xnorm = 0, rnorm = 0;
    for (sf_long i = 0; i < n; ++i) {
        anorm = std::fmax(anorm, colsum[i]);
        bnorm = std::fmax(bnorm, std::fabs(mi->Bx[i]));
        xnorm = std::fmax(xnorm, std::fabs(mi->Xx[i]));
        rnorm = std::fmax(rnorm, std::fabs(mi->Rx[i]));
    }
    mi->residual = rnorm / (anorm * xnorm + bnorm);
    return 0;
}

int SparseFrame_cleanup_matrix(struct matrix_info_struct* mi) {   // L:3860-3922
    if (!mi) return 1;
    sf_handlers_forget(mi->Lsx);
    SF_FREE(Tj); SF_FREE(Ti); SF_FREE(Tx); SF_FREE(Cp); SF_FREE(Ci); SF_FREE(Cx);
    SF_FREE(CPCTp); SF_FREE(CPCTi);
    SF_FREE(Lp); SF_FREE(Li); SF_FREE(Lx); SF_FREE(LTp); SF_FREE(LTi); SF_FREE(LTx);
    SF_FREE(Up); SF_FREE(Ui); SF_FREE(Ux); SF_FREE(UTp); SF_FREE(UTi); SF_FREE(UTx);
    SF_FREE(PivInv); SF_FREE(Perm); SF_FREE(Post); SF_FREE(Parent); SF_FREE(ColCount);
    SF_FREE(Super); SF_FREE(SuperMap); SF_FREE(Sparent); SF_FREE(LeafQueue);
    SF_FREE(Lsip); SF_FREE(Lsxp); SF_FREE(Lsi);
    sf_float* big_lsx = mi->Lsx;            // released LAST (below): a concurrent munmap of tens of GB makes every other munmap wait
    const size_t big_bytes = (size_t)(mi->xsize > 0 ? mi->xsize : 0) * sizeof(sf_float);
    mi->Lsx = nullptr;
    SF_FREE(ST_Map); SF_FREE(ST_Pointer); SF_FREE(ST_Index); SF_FREE(Aoffset); SF_FREE(Moffset);
    SF_FREE(workspace); SF_FREE(Bx); SF_FREE(Xx); SF_FREE(Rx);
    free_big_async((void**)&big_lsx, big_bytes);
    const double rt = mi->readTime, at = mi->analyzeTime, ft = mi->factorizeTime, st = mi->solveTime, res = mi->residual;
    SparseFrame_initialize_matrix(mi);
    mi->readTime = rt; mi->analyzeTime = at; mi->factorizeTime = ft; mi->solveTime = st; mi->residual = res;
    return 0;
}

#include "sf_driver.inc"

}  // extern "C"

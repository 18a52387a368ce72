// Host side of the C ABI: the reference's struct-based entry points (matrix set-up, analysis,
// triangular solve, validation, clean-up) and the flat sf_symbolic_* accessors.
// Reference files: Cholesky/Source/SparseFrame.c (C:), Cholesky/Include/info.h (I:).
#include "sf_host_solve.h"
#include <sparseframe_hip.h>

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
#include <vector>

#include "sf_symbolic.h"

using sf::Long;

struct sf_symbolic {
    sf::Symbolic S;
};

namespace {

double wall_seconds() {
    struct timespec tp;
    clock_gettime(CLOCK_REALTIME, &tp);
    return tp.tv_sec + (double)tp.tv_nsec / 1.0e9;
}

template <class T, class A>
T* dup_array(const std::vector<T, A>& v, size_t min_elems = 1) {
    const size_t n = v.size() > min_elems ? v.size() : min_elems;
    T* p = (T*)malloc(n * sizeof(T));
    // (one thread: splitting the copy of the big arrays -- 16 - 80 MB each, first touch of fresh pages -- over four threads made
    // SparseFrame_analyze 0.45 s SLOWER at 128^3, 1.41 - 1.49 s against 1.10: concurrent first-touch faults of one process serialise on
    // its address-space lock, the effect that also ruled out page-populating helpers for Lsx)
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

// hand a RawVec's buffer to the caller as it is (malloc'ed: sf_symbolic.h); the vector keeps pointing at it until it is destroyed, its
// deallocate then skips the stolen pointer.  Empty vectors get the one-element block the reference's callers may index.
template <class T>
T* steal_array(sf::RawVec<T>& v) {
    if (v.empty()) return (T*)malloc(sizeof(T));
    sf::raw_mark_stolen(v.data());
    return v.data();
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

}  // namespace

extern "C" {

const char* sf_version(void) { return "sparseframe-hip 0.1 (gfx950)"; }

long sf_abi_layout(const char* name) {
    if (!name) return -1;
    const std::string k(name);
    if (k == "sizeof_common") return (long)sizeof(struct common_info_struct);
    if (k == "sizeof_matrix") return (long)sizeof(struct matrix_info_struct);
    if (k == "offsetof_Lsx") return (long)offsetof(struct matrix_info_struct, Lsx);
    if (k == "offsetof_workspace") return (long)offsetof(struct matrix_info_struct, workspace);
    if (k == "offsetof_residual") return (long)offsetof(struct matrix_info_struct, residual);
    if (k == "offsetof_devSlotSize") return (long)offsetof(struct common_info_struct, devSlotSize);
    return -1;
}

// ------------------------------------------------------------------------------------------
// flat symbolic ABI
// ------------------------------------------------------------------------------------------
int sf_symbolic_create(sf_symbolic** out, sf_long n, const sf_long* Cp, const sf_long* Ci, const sf_float* Cx,
                       const sf_long* perm, size_t devSlotSize) {
    if (!out) return SF_ERR_ARG;
    *out = nullptr;
    sf_symbolic* h = new (std::nothrow) sf_symbolic();
    if (!h) return SF_ERR_ALLOC;
    int rc = sf::analyze_cholesky(n, Cp, Ci, Cx, perm, devSlotSize, h->S);
    if (rc) { delete h; return SF_ERR_ARG; }
    *out = h;
    return SF_OK;
}

int sf_symbolic_create_lu(sf_symbolic** out, sf_long n, const sf_long* Cp, const sf_long* Ci, const sf_float* Cx,
                          const sf_long* perm, size_t devSlotSize, int is_symmetric) {
    if (!out) return SF_ERR_ARG;
    *out = nullptr;
    sf_symbolic* h = new (std::nothrow) sf_symbolic();
    if (!h) return SF_ERR_ALLOC;
    int rc = sf::analyze_lu(n, Cp, Ci, Cx, perm, devSlotSize, is_symmetric != 0, h->S);
    if (rc) { delete h; return SF_ERR_ARG; }
    *out = h;
    return SF_OK;
}

void sf_symbolic_destroy(sf_symbolic* sym) { delete sym; }

sf_long sf_symbolic_scalar(const sf_symbolic* sym, const char* name) {
    if (!sym || !name) return -1;
    const sf::Symbolic& S = sym->S;
    const std::string k(name);
    if (k == "n") return S.n;
    if (k == "nnz") return S.Lp.empty() ? 0 : S.Lp[S.n];
    if (k == "nfsuper") return S.nfsuper;
    if (k == "nsuper") return S.nsuper;
    if (k == "nstage") return S.nstage;
    if (k == "isize") return S.isize;
    if (k == "xsize") return S.xsize;
    if (k == "csize") return S.csize;
    if (k == "nsleaf") return S.nsleaf;
    if (k == "lu") return S.lu ? 1 : 0;
    if (k == "symmetric") return S.symmetric ? 1 : 0;
    if (k == "unz") return S.Up.empty() ? 0 : S.Up[S.n];
    return -1;
}

const sf_long* sf_symbolic_long_array(const sf_symbolic* sym, const char* name, sf_long* len) {
    if (len) *len = 0;
    if (!sym || !name) return nullptr;
    const sf::Symbolic& S = sym->S;
    const std::string k(name);
    const std::vector<Long>* v = nullptr;
    const sf::RawVec<Long>* w = (k == "Li") ? &S.Li : (k == "LTi") ? &S.LTi : (k == "Ui") ? &S.Ui : (k == "UTi") ? &S.UTi : (k == "Lsi") ? &S.Lsi : nullptr;
    if (w) {
        if (len) *len = (sf_long)w->size();
        return (const sf_long*)w->data();
    }
    if (k == "Perm") v = &S.Perm;
    else if (k == "Parent") v = &S.Parent;
    else if (k == "Parent0") v = &S.Parent0;
    else if (k == "Post") v = &S.Post;
    else if (k == "ColCount") v = &S.ColCount;
    else if (k == "ColCount0") v = &S.ColCount0;
    else if (k == "Lp") v = &S.Lp;
    else if (k == "LTp") v = &S.LTp;
    else if (k == "Up") v = &S.Up;
    else if (k == "UTp") v = &S.UTp;
    else if (k == "Super") v = &S.Super;
    else if (k == "SuperMap") v = &S.SuperMap;
    else if (k == "Sparent") v = &S.Sparent;
    else if (k == "Lsip") v = &S.Lsip;
    else if (k == "Lsxp") v = &S.Lsxp;
    else if (k == "LeafQueue") v = &S.LeafQueue;
    else if (k == "ST_Map") v = &S.ST_Map;
    else if (k == "ST_Pointer") v = &S.ST_Pointer;
    else if (k == "ST_Index") v = &S.ST_Index;
    else if (k == "Aoffset") v = &S.Aoffset;
    else if (k == "Moffset") v = &S.Moffset;
    if (!v) return nullptr;
    if (len) *len = (sf_long)v->size();
    return (const sf_long*)v->data();
}

const sf_float* sf_symbolic_float_array(const sf_symbolic* sym, const char* name, sf_long* len) {
    if (len) *len = 0;
    if (!sym || !name) return nullptr;
    const std::string k(name);
    const sf::RawVec<double>* v = nullptr;
    if (k == "Lx") v = &sym->S.Lx;
    else if (k == "LTx") v = &sym->S.LTx;
    else if (k == "Ux") v = &sym->S.Ux;
    else if (k == "UTx") v = &sym->S.UTx;
    if (!v) return nullptr;
    if (len) *len = (sf_long)v->size();
    return v->data();
}

double sf_symbolic_flops(const sf_symbolic* sym, int which) {
    if (!sym) return 0;
    if (which == 0) return sf::flops_struct(sym->S);
    double upd = 0, sc = 0;
    const double all = sf::flops_exec(sym->S, &upd, &sc);
    if (which == 1) return all;
    if (which == 2) return upd;
    if (which == 3) return sc;
    return 0;
}

int sf_subtree_partition(sf_long nsuper, const sf_long* Super, const sf_long* SuperMap, const sf_long* Lsip,
                         const sf_long* Lsi, int nranks, int32_t* owner, double* top_fraction, double* max_load_fraction) {
    if (!Super || !Lsip || (nsuper > 0 && (!SuperMap || !Lsi))) return SF_ERR_ARG;
    return sf::subtree_partition(nsuper, Super, SuperMap, Lsip, Lsi, nranks, owner, top_fraction, max_load_fraction) ? SF_ERR_ARG : SF_OK;
}

int sf_subtree_partition_weighted(sf_long nsuper, const sf_long* Super, const sf_long* SuperMap, const sf_long* Lsip,
                                  const sf_long* Lsi, int nranks, double top_weight, int32_t* owner, double* top_fraction,
                                  double* max_load_fraction) {
    if (!Super || !Lsip || (nsuper > 0 && (!SuperMap || !Lsi))) return SF_ERR_ARG;
    return sf::subtree_partition(nsuper, Super, SuperMap, Lsip, Lsi, nranks, owner, top_fraction, max_load_fraction, top_weight) ? SF_ERR_ARG : SF_OK;
}

int sf_ooc_partition(sf_long nsuper, const sf_long* Super, const sf_long* SuperMap, const sf_long* Lsip, const sf_long* Lsi,
                     sf_long budget_entries, int32_t* group, int* ngroups, sf_long* group_entries, sf_long* top_entries, sf_long* need_entries,
                     int* top_mode) {
    if (!Super || !Lsip || !group || !ngroups || (nsuper > 0 && (!SuperMap || !Lsi))) return SF_ERR_ARG;
    int64_t ge = 0, te = 0, nd = 0;
    int mode = 0;
    const int rc = sf::ooc_partition(nsuper, Super, SuperMap, Lsip, Lsi, budget_entries, group, ngroups, &ge, &te, &nd, &mode);
    if (group_entries) *group_entries = ge;
    if (top_entries) *top_entries = te;
    if (need_entries) *need_entries = nd;
    if (top_mode) *top_mode = mode;
    return rc == 0 ? SF_OK : (rc == 2 ? SF_ERR_ALLOC : SF_ERR_ARG);
}

int sf_graph_nd_perm(sf_long n, const sf_long* Cp, const sf_long* Ci, sf_long leaf, sf_long* perm) {
    return sf::graph_nd_perm(n, Cp, Ci, leaf, perm) ? SF_ERR_ARG : SF_OK;
}

int sf_grid_nd_perm(sf_long nx, sf_long ny, sf_long nz, sf_long leaf, sf_long sep_width, sf_long* perm) {
    return sf::grid_nd_perm(nx, ny, nz, leaf, sep_width, perm) ? SF_ERR_ARG : SF_OK;
}

// ------------------------------------------------------------------------------------------
// struct-based entry points
// ------------------------------------------------------------------------------------------
int SparseFrame_initialize_matrix(struct matrix_info_struct* mi) {  // C:589-650
    if (!mi) return 1;
    const int serial = mi->serial;
    const char* path = mi->path;
    memset(mi, 0, sizeof(*mi));
    mi->serial = serial;
    mi->path = path;
    mi->factorizeType = TYPE_CHOLESKY;
    // the reference's analyze always orders with METIS (C:1937); here PERM_METIS with no Perm supplied selects the
    // built-in nested dissection, and natural order is an explicit opt-in: SparseFrame_set_perm(mi, NULL)
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
    // workspace sized as C:679-684 (the idx_t term is the same size with 64-bit idx_t)
    mi->workSize = (size_t)(10 * nrow + (2 * nz - nrow) + 1) * sizeof(sf_long);
    // factorize carves 8*nsuper Longs out of it (C:2221-2228); nsuper <= nrow so 10*nrow covers it
    mi->workspace = malloc(mi->workSize);
    if (!mi->Cp || !mi->Ci || !mi->Cx || !mi->workspace) return 1;
    memcpy(mi->Cp, Cp, (nrow + 1) * sizeof(sf_long));
    if (nz > 0) {
        memcpy(mi->Ci, Ci, nz * sizeof(sf_long));
        memcpy(mi->Cx, Cx, nz * sizeof(sf_float));
    }
    return 0;
}

// MatrixMarket "coordinate real {symmetric|general}" reader; drops explicit zeros (C:496) and
// compresses the triplets to CSC with a counting sort (C:526-587).
int SparseFrame_read_matrix(struct matrix_info_struct* mi) {
    if (!mi || !mi->path) return 1;
    const double t0 = wall_seconds();
    FILE* f = fopen(mi->path, "r");
    if (!f) return 1;
    char* line = nullptr;
    size_t cap = 0;
    int rc = 1;
    std::vector<Long> Ti, Tj;
    std::vector<double> Tx;
    long nrow = 0, ncol = 0, nzmax = 0;
    int symmetric = 0;
    do {
        ssize_t got;
        while ((got = getline(&line, &cap, f)) != -1 && line[0] == '\n') {}
        if (got == -1 || strncmp(line, "%%MatrixMarket", 14) != 0) break;
        char s0[64], s1[64], s2[64], s3[64], s4[64];
        s3[0] = s4[0] = 0;
        sscanf(line, "%63s %63s %63s %63s %63s", s0, s1, s2, s3, s4);
        symmetric = strcmp(s4, "symmetric") == 0;
        if (strcmp(s3, "real") != 0 && strcmp(s3, "integer") != 0) break;  // complex: unsupported (reference solve is TODO)
        while ((got = getline(&line, &cap, f)) != -1 && (line[0] == '%' || line[0] == '\n')) {}
        if (got == -1 || sscanf(line, "%ld %ld %ld", &nrow, &ncol, &nzmax) != 3) break;
        if (nrow != ncol || nrow < 0 || nzmax < 0) break;
        Ti.reserve(nzmax); Tj.reserve(nzmax); Tx.reserve(nzmax);
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
        if (bad) break;
        rc = 0;
    } while (0);
    free(line);
    fclose(f);
    if (rc) return rc;

    const Long nz = (Long)Tx.size();
    std::vector<Long> Cp(ncol + 1, 0), Ci(nz > 0 ? nz : 1);
    std::vector<double> Cx(nz > 0 ? nz : 1);
    for (Long k = 0; k < nz; ++k) Cp[Tj[k] + 1]++;
    for (Long j = 0; j < ncol; ++j) Cp[j + 1] += Cp[j];
    std::vector<Long> fill(Cp.begin(), Cp.end() - 1);
    for (Long k = 0; k < nz; ++k) {
        const Long p = fill[Tj[k]]++;
        Ci[p] = Ti[k];
        Cx[p] = Tx[k];
    }
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
    mi->permMethod = PERM_METIS;  // "externally ordered"
    return 0;
}

int SparseFrame_analyze(struct common_info_struct* common, struct matrix_info_struct* mi) {  // C:1916-1978
    if (!common || !mi || !mi->Cp) return 1;
    const double t0 = wall_seconds();
    sf::Symbolic S;
    const sf_long* perm = (mi->permMethod != PERM_IDENTITY) ? mi->Perm : nullptr;
    std::vector<Long> builtin;
    if (mi->permMethod != PERM_IDENTITY && !mi->Perm) {
        // the reference orders with METIS_NodeND here (C:1937); the built-in stand-in is BFS nested dissection
        builtin.resize(mi->nrow);
        if (sf::graph_nd_perm(mi->nrow, mi->Cp, mi->Ci, 64, builtin.data())) return 1;
        perm = builtin.data();
    }
    int rc = sf::analyze_cholesky(mi->nrow, mi->Cp, mi->Ci, mi->Cx, perm, common->devSlotSize, S);
    if (rc) return rc;

    SF_FREE(Lp); SF_FREE(Li); SF_FREE(Lx); SF_FREE(LTp); SF_FREE(LTi); SF_FREE(LTx);
    SF_FREE(Perm); SF_FREE(Parent); SF_FREE(Post); SF_FREE(ColCount);
    SF_FREE(Super); SF_FREE(SuperMap); SF_FREE(Sparent); SF_FREE(LeafQueue);
    SF_FREE(Lsip); SF_FREE(Lsxp); SF_FREE(Lsi); SF_FREE(Lsx);
    SF_FREE(ST_Map); SF_FREE(ST_Pointer); SF_FREE(ST_Index); SF_FREE(Aoffset); SF_FREE(Moffset);

    mi->Lp = dup_array(S.Lp);   mi->Li = steal_array(S.Li);   mi->Lx = steal_array(S.Lx);
    mi->LTp = dup_array(S.LTp); mi->LTi = steal_array(S.LTi); mi->LTx = steal_array(S.LTx);
    mi->Perm = dup_array(S.Perm);
    mi->Parent = dup_array(S.Parent);
    mi->Post = dup_array(S.Post);
    mi->ColCount = dup_array(S.ColCount);
    mi->nsuper = S.nsuper;
    // the reference allocates Super with nrow+1 entries, SuperMap/Sparent with nrow (C:1969-1971)
    mi->Super = (sf_long*)calloc(mi->nrow + 1, sizeof(sf_long));
    mi->Sparent = (sf_long*)calloc(mi->nrow > 0 ? mi->nrow : 1, sizeof(sf_long));
    if (mi->Super) memcpy(mi->Super, S.Super.data(), S.Super.size() * sizeof(sf_long));
    if (mi->Sparent && S.nsuper) memcpy(mi->Sparent, S.Sparent.data(), S.nsuper * sizeof(sf_long));
    mi->SuperMap = dup_array(S.SuperMap);
    mi->nsleaf = S.nsleaf;
    mi->LeafQueue = dup_array(S.LeafQueue);
    mi->isize = S.isize;
    mi->xsize = S.xsize;
    mi->Lsip = dup_array(S.Lsip);
    mi->Lsxp = dup_array(S.Lsxp);
    mi->Lsi = steal_array(S.Lsi);
    mi->Lsx = (sf_float*)malloc((S.xsize > 0 ? S.xsize : 1) * sizeof(sf_float));  // output, C:1647-1651
    mi->csize = S.csize;
    mi->nstage = S.nstage;
    mi->ST_Map = dup_array(S.ST_Map);
    mi->ST_Pointer = dup_array(S.ST_Pointer);
    mi->ST_Index = dup_array(S.ST_Index);
    mi->ST_Parent = nullptr;
    mi->Aoffset = (size_t*)malloc((S.nsuper > 0 ? S.nsuper : 1) * sizeof(size_t));
    mi->Moffset = (size_t*)malloc((S.nsuper > 0 ? S.nsuper : 1) * sizeof(size_t));
    if (!mi->Lsx || !mi->Aoffset || !mi->Moffset || !mi->Super || !mi->Sparent) return 1;
    for (Long s = 0; s < S.nsuper; ++s) {
        mi->Aoffset[s] = (size_t)S.Aoffset[s];
        mi->Moffset[s] = (size_t)S.Moffset[s];
    }
    mi->analyzeTime = wall_seconds() - t0;
    return 0;
}

// Forward and backward supernodal substitution on the host, permuted space (C:3036-3139).
int SparseFrame_solve_supernodal(struct matrix_info_struct* mi) {
    if (!mi || !mi->Lsx || !mi->Bx || !mi->Xx) return 1;
    const double t0 = wall_seconds();
    // the factor SparseFrame_factorize copied into Lsx is normally still resident in the handler's plan: solve there (two sweeps
    // over the factor in HBM instead of host memory).  Falls through to the reference's host solve when it is not (several
    // handlers, plan evicted or re-used, Lsx changed by the caller, SF_SOLVE=host).
    if (sf_handlers_solve_resident_sym(mi->Lsx, mi->Bx, mi->Xx, 0, mi->nrow, mi->nsuper, mi->Super, mi->SuperMap, mi->Lsip, mi->Lsi,
                                       mi->Lsxp, mi->Lp, mi->Li, nullptr, nullptr) == SF_OK) {
        mi->solveTime = wall_seconds() - t0;
        return 0;
    }
    const Long ns = mi->nsuper;
    double* x = mi->Xx;
    memcpy(x, mi->Bx, mi->nrow * sizeof(double));
    // a large factor: the same two sweeps on several threads (sf_host_solve.h); the scalar sweep below is the reference's (C:3036-3139)
    if (const int T = sf_host_solve::threads_for((sf_host_solve::Long)mi->xsize); T > 1) {
        std::vector<int32_t> owner((size_t)(ns > 0 ? ns : 1), 0);
        // (the top is shared by all threads: a top flop costs 1 / T of a subtree flop, plus the barriers)
        if (sf_subtree_partition_weighted(ns, mi->Super, mi->SuperMap, mi->Lsip, mi->Lsi, T, 1.0 / T + 0.05, owner.data(), nullptr, nullptr) == SF_OK) {
            sf_host_solve::solve_parallel<false>(mi->nrow, ns, mi->Super, mi->SuperMap, mi->Lsip, mi->Lsi, mi->Lsxp, mi->Lsx, nullptr,
                                                 owner.data(), T, x);
            mi->solveTime = wall_seconds() - t0;
            return 0;
        }
    }
    for (Long s = 0; s < ns; ++s) {
        const Long nscol = mi->Super[s + 1] - mi->Super[s];
        const Long nsrow = mi->Lsip[s + 1] - mi->Lsip[s];
        const Long* rows = mi->Lsi + mi->Lsip[s];
        const double* P = mi->Lsx + mi->Lsxp[s];
        for (Long c = 0; c < nscol; ++c) {
            const double* col = P + c * nsrow;
            const double xj = (x[rows[c]] /= col[c]);
            for (Long r = c + 1; r < nsrow; ++r) x[rows[r]] -= col[r] * xj;
        }
    }
    for (Long s = ns - 1; s >= 0; --s) {
        const Long nscol = mi->Super[s + 1] - mi->Super[s];
        const Long nsrow = mi->Lsip[s + 1] - mi->Lsip[s];
        const Long* rows = mi->Lsi + mi->Lsip[s];
        const double* P = mi->Lsx + mi->Lsxp[s];
        for (Long c = nscol - 1; c >= 0; --c) {
            const double* col = P + c * nsrow;
            double acc = x[rows[c]];
            for (Long r = c + 1; r < nsrow; ++r) acc -= col[r] * x[rows[r]];
            x[rows[c]] = acc / col[c];
        }
    }
    mi->solveTime = wall_seconds() - t0;
    return 0;
}

// b_i = 1 + i/n, solve, r = A x - b over the stored triangle used symmetrically,
// residual = |r|_inf / (|A|_1 |x|_inf + |b|_inf)   (C:3141-3266)
int SparseFrame_validate(struct matrix_info_struct* mi) {
    if (!mi || !mi->Lp || !mi->Lsx) return 1;
    const Long n = mi->nrow;
    SF_FREE(Bx); SF_FREE(Xx); SF_FREE(Rx);
    mi->Bx = (double*)malloc((n > 0 ? n : 1) * sizeof(double));
    mi->Xx = (double*)malloc((n > 0 ? n : 1) * sizeof(double));
    mi->Rx = (double*)malloc((n > 0 ? n : 1) * sizeof(double));
    if (!mi->Bx || !mi->Xx || !mi->Rx) return 1;
    for (Long i = 0; i < n; ++i) mi->Bx[i] = 1 + i / (double)n;
    int rc = SparseFrame_solve_supernodal(mi);
    if (rc) return rc;
    std::vector<double> colsum(n, 0.0);
    for (Long i = 0; i < n; ++i) mi->Rx[i] = -mi->Bx[i];
    for (Long j = 0; j < n; ++j) {
        for (Long p = mi->Lp[j]; p < mi->Lp[j + 1]; ++p) {
            const Long i = mi->Li[p];
            const double a = mi->Lx[p];
            mi->Rx[i] += a * mi->Xx[j];
            colsum[j] += std::fabs(a);
            if (i != j) {
                mi->Rx[j] += a * mi->Xx[i];
                colsum[i] += std::fabs(a);
            }
        }
    }
    double anorm = 0, bnorm = 0, xnorm = 0, rnorm = 0;
    for (Long i = 0; i < n; ++i) {
        anorm = std::fmax(anorm, colsum[i]);
        bnorm = std::fmax(bnorm, std::fabs(mi->Bx[i]));
        xnorm = std::fmax(xnorm, std::fabs(mi->Xx[i]));
        rnorm = std::fmax(rnorm, std::fabs(mi->Rx[i]));
    }
    mi->residual = rnorm / (anorm * xnorm + bnorm);
    return 0;
}

int SparseFrame_cleanup_matrix(struct matrix_info_struct* mi) {  // C:3268-3321
    if (!mi) return 1;
    sf_handlers_forget(mi->Lsx);
    SF_FREE(Tj); SF_FREE(Ti); SF_FREE(Tx);
    SF_FREE(Cp); SF_FREE(Ci); SF_FREE(Cx);
    SF_FREE(Lp); SF_FREE(Li); SF_FREE(Lx);
    SF_FREE(LTp); SF_FREE(LTi); SF_FREE(LTx);
    SF_FREE(Perm); SF_FREE(Post); SF_FREE(Parent); SF_FREE(ColCount);
    SF_FREE(Super); SF_FREE(SuperMap); SF_FREE(Sparent); SF_FREE(LeafQueue);
    SF_FREE(Lsip); SF_FREE(Lsxp); SF_FREE(Lsi);
    sf_float* big_lsx = mi->Lsx;            // released LAST (below): a concurrent munmap of tens of GB makes every other munmap wait
    const size_t big_bytes = (size_t)(mi->xsize > 0 ? mi->xsize : 0) * sizeof(sf_float);
    mi->Lsx = nullptr;
    SF_FREE(ST_Map); SF_FREE(ST_Pointer); SF_FREE(ST_Index);
    SF_FREE(Aoffset); SF_FREE(Moffset);
    SF_FREE(workspace);
    SF_FREE(Bx); SF_FREE(Xx); SF_FREE(Rx);
    free_big_async((void**)&big_lsx, big_bytes);
    // keep the timers and the residual readable after clean-up, as the reference's driver
    // prints them after calling this (C:3423-3433)
    const double rt = mi->readTime, at = mi->analyzeTime, ft = mi->factorizeTime, st = mi->solveTime;
    const double res = mi->residual;
    SparseFrame_initialize_matrix(mi);
    mi->readTime = rt; mi->analyzeTime = at; mi->factorizeTime = ft; mi->solveTime = st;
    mi->residual = res;
    return 0;
}

#include "sf_driver.inc"

}  // extern "C"

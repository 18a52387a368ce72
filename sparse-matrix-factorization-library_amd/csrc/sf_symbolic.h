// Internal C++ view of the symbolic analysis result.  Host only.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <new>
#include <utility>
#include <vector>

namespace sf {

using Long = int64_t;

// std::vector whose resize() leaves new elements uninitialized: the index / value arrays of the triangles are tens to hundreds of
// MB that are overwritten entry by entry right after; zero-filling them first is a single-threaded pass over fresh pages.
// Its storage comes from malloc, and a buffer can be STOLEN (raw_steal): the struct entry point SparseFrame_analyze hands the big
// arrays to matrix_info as they are -- the reference's cleanup free()s them -- instead of copying 340 MB into fresh pages
// (0.1 -> 0.03 s of the 128^3 analysis); the vector's destructor then finds the pointer in the stolen set and leaves it alone.
void raw_mark_stolen(void* p);
bool raw_take_if_stolen(void* p);       // true (and forgets p) if p was stolen: the caller must not free it
template <class T>
struct default_init_allocator {
    using value_type = T;
    default_init_allocator() noexcept = default;
    template <class U> default_init_allocator(const default_init_allocator<U>&) noexcept {}
    template <class U> struct rebind { using other = default_init_allocator<U>; };
    T* allocate(std::size_t n) {
        void* p = std::malloc(n > 0 ? n * sizeof(T) : 1);
        if (!p) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, std::size_t) noexcept { if (p && !raw_take_if_stolen(p)) std::free(p); }
    template <class U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const default_init_allocator<U>&) const noexcept { return true; }
    template <class U> bool operator!=(const default_init_allocator<U>&) const noexcept { return false; }
};
template <class T> using RawVec = std::vector<T, default_init_allocator<T>>;

struct Symbolic {
    Long n = 0;
    size_t devSlotSize = 0;

    bool lu = false;         // LU path (reference LU/Source/SparseFrame.c): panels are (2*nsrow - nscol) x nscol
    bool symmetric = true;   // input holds one triangle of a symmetric matrix (LU: U aliases L, L:2718-2729)

    // permuted lower triangle (by column) and its transpose; reference C:956-1066
    std::vector<Long> Lp, LTp;
    RawVec<Long> Li, LTi;
    RawVec<double> Lx, LTx;
    // LU, unsymmetric input only: U by ROW (Up[i]..: column indices j >= i) and its transpose (L:1179-1282)
    std::vector<Long> Up, UTp;
    RawVec<Long> Ui, UTi;
    RawVec<double> Ux, UTx;

    std::vector<Long> Perm;      // final (post-order composed), C:1438
    std::vector<Long> Parent;    // final numbering, C:1439
    std::vector<Long> ColCount;  // final numbering, C:1440
    std::vector<Long> Post;      // the second (weighted) postorder, C:1967
    std::vector<Long> Parent0;   // etree before renumbering
    std::vector<Long> ColCount0; // column counts before renumbering

    Long nfsuper = 0, nsuper = 0;
    std::vector<Long> Super, SuperMap, Sparent;
    std::vector<Long> Lsip, Lsxp;
    RawVec<Long> Lsi;         // (stealable, see RawVec)
    Long isize = 0, xsize = 0, csize = 0;

    Long nstage = 0;
    std::vector<Long> ST_Map, ST_Pointer, ST_Index;
    std::vector<Long> Aoffset, Moffset;   // bytes, as size_t in the reference

    Long nsleaf = 0;
    std::vector<Long> LeafQueue;
};

// Cp/Ci/Cx: one triangle of the symmetric matrix in CSC.  perm may be null (identity).
// Returns 0 on success.
int analyze_cholesky(Long n, const Long* Cp, const Long* Ci, const double* Cx,
                     const Long* perm, size_t devSlotSize, Symbolic& out);
// LU variant (reference LU/Source/SparseFrame.c:1068-2231).  symmetric != 0: Cp/Ci/Cx hold one triangle;
// otherwise the whole (structurally unsymmetric) matrix in CSC.
int analyze_lu(Long n, const Long* Cp, const Long* Ci, const double* Cx,
               const Long* perm, size_t devSlotSize, bool symmetric, Symbolic& out);

// flop counters
double flops_struct(const Symbolic& S);
double flops_exec(const Symbolic& S, double* update_flops, double* scatter_elems);

int subtree_partition(Long nsuper, const Long* Super, const Long* SuperMap, const Long* Lsip, const Long* Lsi,
                      int nranks, int32_t* owner, double* top_fraction, double* max_load_fraction, double top_weight = 1.0);

// out-of-core grouping (sf_symbolic.cpp): group[s] = streamed group of supernode s or -1 (top); *top_mode 0 = the top panels are
// resident throughout, 1 / 2 = only while active (ooc_top_layout gives their offsets; 2: places re-used one group earlier); 0 = fits `budget` panel entries
int ooc_partition(Long nsuper, const Long* Super, const Long* SuperMap, const Long* Lsip, const Long* Lsi, int64_t budget,
                  int32_t* group, int* ngroups, int64_t* group_entries, int64_t* top_entries, int64_t* need, int* top_mode);
int ooc_top_layout(Long nsuper, const Long* Super, const Long* SuperMap, const Long* Lsip, const Long* Lsi, const int32_t* group, int ngroups,
                   int mode, int32_t* first, int32_t* last, int64_t* off, int32_t* wait, int64_t* arena);

int graph_nd_perm(Long n, const Long* Cp, const Long* Ci, Long leaf, Long* perm);

int grid_nd_perm(Long nx, Long ny, Long nz, Long leaf, Long sepw, Long* perm);

}  // namespace sf

// Device plan of the supernodal Cholesky numeric factorization.
//
// Replaces the reference's task-queue scheduler and GPU branch (Cholesky/Source/SparseFrame.c:2150-3017).
// The reference is left-looking with descendant lists, one supernode at a time, every panel staged over
// PCIe (C:2345-2954).  Here the whole factor stays resident in HBM and the elimination tree is swept
// level by level (level = height above the leaves in the supernodal tree, parent as in C:2247):
//
//   memset(Lsx) ; k_load_panels                                          (loadA, C:1998-2028)
//   for level l = 0 .. L-1:
//       for every outer block column J (512 columns) of the panels of the level:
//           k_gemm<0>      left-looking update of the block column by all columns to its left (K = J)
//           for t = 0 .. 7:                            64-column steps inside the block, also left-looking
//               k_gemm<0>      update of block column t by block columns 0..t-1 of this outer block (K = 64 t)
//               k_potrf_block  64x64 diagonal block of every panel that still has one
//               k_trsm_block   rows below that block
//       k_gemm<1>          every (supernode of the level -> ancestor) Schur update, scatter fused
//
// A supernode's updates are pushed to all its ancestors as soon as it is factored (right-looking);
// the sums are the same as the reference's left-looking sums, applied in a different order, so the
// factor agrees to rounding (the reference itself is order-nondeterministic: qsort + atomicAdd).
// All task tables depend on the structure only and are built once at plan creation.
#include <sparseframe_hip.h>

#include <algorithm>
#include <time.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sf_plan_internal.h"
#include "sf_symbolic.h"

// hipEventElapsedTime whose failure (an event that was never recorded) is expected and must not stay behind as the thread's
// "last error": callers that poll hipGetLastError after their own launches (PyTorch does) would report it as theirs
static bool elapsed_ms(float* ms, hipEvent_t a, hipEvent_t b) {
    if (hipEventElapsedTime(ms, a, b) == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}

extern "C" {

int sf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

size_t sf_device_memory(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
    return (size_t)prop.totalGlobalMem;
}

size_t sf_reference_slot_size(int ndev, size_t min_mem) {
    if (ndev <= 0) return 0;
    // C:36-41 numSplit = max(GPU_SPLIT_LIMIT / numGPU_physical, 1), GPU_SPLIT_LIMIT = 4 (parameter.h:19);
    // C:82-87: (mem - 64 MiB) * 0.9 / numSplit / 8 slots, rounded down to 1 MiB; C:199 devSlotSize = that
    const int numSplit = std::max(4 / ndev, 1);
    size_t m = (size_t)(((double)min_mem - 64.0 * (0x400 * 0x400)) * 0.9);
    m /= numSplit;
    m /= 8;
    m -= m % (0x400 * 0x400);
    return m;
}

// The NEXT plan this thread creates may place its factor in a buffer of the caller's instead of allocating one (the struct path's
// per-handler pool, sf_handlers.hip: the reference allocates its device slots once, in SparseFrame_allocate_gpu, C:92-283 -- so does
// the pool; a 30 GB hipMalloc inside the first SparseFrame_factorize of a pattern costs 12 - 790 ms depending on the box's state).
// The offer is consumed by that one plan_create whether or not the buffer is large enough (sf_plan_factor_borrowed tells).
static thread_local double* t_offer_ptr = nullptr;
static thread_local size_t t_offer_bytes = 0;
void sf_plan_offer_factor_buffer(double* ptr, size_t bytes) { t_offer_ptr = ptr; t_offer_bytes = bytes; }
int sf_plan_factor_borrowed(const sf_chol_plan* p) { return p && p->factor_borrowed ? 1 : 0; }

int sf_chol_plan_destroy(sf_chol_plan* p) {
    if (!p) return SF_OK;
    if (p->dry) { delete p; return SF_OK; }
    (void)hipSetDevice(p->device);
    void* ptrs[] = {p->d_Lp, p->d_Li, p->d_Lx, p->d_Super, p->d_SuperMap, p->d_Lsip, p->d_Lsi, p->d_Lsxp,
                    p->d_Lsx, p->d_info, p->d_potrf, p->d_trsm, p->d_steps, p->d_flags, p->d_tinv, p->d_probs, p->d_gtasks, p->d_stasks, p->d_ktprefix,
                    p->d_Up, p->d_Ui, p->d_Ux, p->d_Xp, p->d_pack, p->d_piv, p->d_resid, p->d_loadmask, p->d_loadmapL, p->d_loadmapU, p->d_solve, p->d_solve_sync, p->d_x, p->d_relmap, p->d_scratch, p->d_status, p->d_fill, p->d_solveT, p->d_solveT_list};
    for (void* q : ptrs)
        if (q && !(q == (void*)p->d_Lsx && p->factor_borrowed)) (void)hipFree(q);      // (a borrowed factor buffer goes back to its lender)
    if (p->graph_exec) (void)hipGraphExecDestroy(p->graph_exec);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->ev_s0) (void)hipEventDestroy(p->ev_s0);
    if (p->ev_s1) (void)hipEventDestroy(p->ev_s1);
    for (int k = 0; k < 2; ++k) {
        if (p->ev_contrib[k]) (void)hipEventDestroy(p->ev_contrib[k]);
        if (p->ev_reduced[k]) (void)hipEventDestroy(p->ev_reduced[k]);
        if (p->ev_unpacked[k]) (void)hipEventDestroy(p->ev_unpacked[k]);
    }
    for (hipEvent_t e : p->la_events) if (e) (void)hipEventDestroy(e);
    if (p->stream2) (void)hipStreamDestroy(p->stream2);
    for (hipEvent_t e : p->dl_events)
        if (e) (void)hipEventDestroy(e);
    for (int w = 0; w < DL_WORKERS_MAX; ++w) {
        if (p->dl_streams[w]) (void)hipStreamDestroy(p->dl_streams[w]);
        for (int k = 0; k < 2; ++k)
            if (p->dl_done[w][k]) (void)hipEventDestroy(p->dl_done[w][k]);
    }
    if (p->h_ring) (void)hipHostFree(p->h_ring);
    if (p->d_ring) (void)hipFree(p->d_ring);
    if (p->stream && p->own_stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return SF_OK;
}

// Common plan builder.  lu == false: Cholesky, one nsrow x nscol panel per supernode at Lsxp[s] (the reference layout).
// lu == true: no-pivot LU kept on the device as a PAIR of nsrow x nscol panels per supernode,
//     PL(i,j) = L(i,j) for i > j (unit diagonal implied), PU(i,j) = U(j,i) for i >= j  (i.e. U^T),
// all PL panels first, then all PU panels (shift xC).  With that pairing every LU operation is the Cholesky one
// with the two operand roles taken from different panels:
//     L-panel update   C(ci,cj) -= sum_k PL(ci,k) PU(cj,k), ci >  cj      U-panel update  C -= sum_k PU(ci,k) PL(cj,k), ci >= cj
//     L21 <- L21 U11^{-1}  = k_trsm_block with D = PU's diagonal block    U12^T <- U12^T L11^{-T} = same kernel, D = PL's, unit
// which are exactly the reference's two GEMMs per update (L:2570-2577), its two-target scatter (L:2583-2604) and its
// two triangular solves (L:2653-2662), and the factor is gathered into the reference's (2*nsrow-nscol) x nscol panels
// (L:2514-2517) only when it is downloaded.
static int plan_create(sf_chol_plan** out, int device, bool lu, sf_long n, sf_long nsuper,
                       const sf_long* Super, const sf_long* SuperMap,
                       const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                       const sf_long* Lp, const sf_long* Li, const sf_long* Up, const sf_long* Ui,
                       const int32_t* phase_in = nullptr, int load_top = 1, int rank = 0, int nranks = 1,
                       const uint32_t* top_mask = nullptr, const double* root_cum = nullptr, bool dry = false,
                       const int32_t* ooc_group = nullptr, int ooc_ngroups = 0, int ooc_top_mode = 0) {
    // ooc_group (one device, nranks == 1, no phase array): OUT-OF-CORE plan -- the factor does not stay on the device.  ooc_group[s] in
    // [0, ooc_ngroups) = the streamed group of supernode s (sf::ooc_partition: whole subtrees, consecutive in the postorder), -1 = top.
    // The top panels are resident; the groups' panels alias TWO buffers of the largest group's size (group g lives in buffer g & 1)
    // and are factorized group by group, group g while group g - 1 is copied to the host; a group's buffer is zeroed and assembled
    // again (launch kind 7) once the copy of group g - 2 has left the device.  Such a plan only runs through the overlapped download
    // (sf_chol_plan_factorize_to_host); it cannot solve or hand out its factor (partial).
    // ooc_top_mode 1 (sf::ooc_partition decides): the top panels are not resident throughout either -- a top supernode is assembled
    // with the first group below it, factorized right after the last one, and its place in the top arena (sf::ooc_top_layout) is
    // given to another two groups later.  For factors that mode 0's resident top does not fit.
    // dry: build the SCHEDULE only (launch list, segments, solve reduces, storage map, byte counts) -- no device is touched, nothing is
    // allocated or uploaded, and the resulting plan can only be inspected (sf_chol_plan_launch_info & co.) and destroyed.  It is the
    // same code path as a real plan up to the uploads, which is the point: what a rank WOULD do at a size or rank count this box cannot
    // run (tests/test_config4_schedules.py walks all eight plans of 256^3 / 8 and checks them against each other).
    // root_cum (optional, nranks + 1 values from 0 to 1): the shares of the split launches of the sets that ALL ranks take part in
    // are [root_cum[r], root_cum[r + 1]) instead of equal ones -- create_mapped uses them to even out ranks whose other groups
    // differ in weight (an elimination tree the amalgamation made lopsided)
    // top_mask[s] (phase-1 supernodes, optional): bit r set = rank r takes part in supernode s (proportional mapping: the
    // ranks whose subtrees lie below s).  Its panel exists only on those ranks, their GEMM shares and the all-reduce run
    // inside that group.  Without it every rank takes part in every top supernode.
    if (!out) return SF_ERR_ARG;
    *out = nullptr;
    if (nranks < 1 || nranks > 32 || rank < 0 || rank >= nranks || (nranks > 1 && !phase_in)) return SF_ERR_ARG;
    const bool ooc = ooc_group != nullptr && ooc_ngroups > 1;
    if (ooc && (nranks != 1 || phase_in || ooc_ngroups > 32767)) return SF_ERR_ARG;
    if (n < 0 || nsuper < 0 || !Super || !Lsip || !Lsxp || !Lp || (n > 0 && (!SuperMap || !Lsi || !Li))) return SF_ERR_ARG;
    if (n >= (sf_long)0x7fffffff) return SF_ERR_ARG;   // device row indices are 32-bit
    if (!dry) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
            fprintf(stderr, "[sparseframe-hip] no HIP device: the numeric factorization has no CPU fallback\n");
            return SF_ERR_NO_DEVICE;
        }
        if (device < 0 || device >= ndev) return SF_ERR_ARG;
        HIP_TRY(hipSetDevice(device));
    }

    sf_chol_plan* p = new (std::nothrow) sf_chol_plan();
    if (!p) return SF_ERR_ALLOC;
    p->device = dry ? -1 : device;
    p->dry = dry;
    p->lu = lu;
    p->rank = rank;
    p->nranks = nranks;
    p->n = n;
    p->nsuper = nsuper;
    p->nnz = Lp[n];
    p->isize = Lsip[nsuper];
    p->xsize = Lsxp[nsuper];
    p->u_alias = lu && (!Up || !Ui);
    p->unz = (lu && !p->u_alias) ? Up[n] : 0;
    // Device offsets of the nsrow x nscol panels: the owned (phase 0) panels in supernode order, then the top
    // (phase 1) panels, contiguous, so that the multi-GPU merge is ONE all-reduce over [top_off, top_off+top_size).
    // With no phase array every supernode is phase 0 and the layout is the reference's Lsxp (Cholesky).
    p->phase.assign(nsuper, 0);
    if (ooc) {
        for (sf_long s = 0; s < nsuper; ++s) {
            if (ooc_group[s] < -1 || ooc_group[s] >= ooc_ngroups) { delete p; return SF_ERR_ARG; }
            p->phase[s] = ooc_group[s] < 0 ? 1 : 0;
        }
        p->ooc_groups = ooc_ngroups;
    }
    if (phase_in)
        for (sf_long s = 0; s < nsuper; ++s) {
            if (phase_in[s] < -1 || phase_in[s] > 1) { delete p; return SF_ERR_ARG; }
            p->phase[s] = (int8_t)phase_in[s];
        }
    // group of every top supernode stored here: mask, this rank's index in it and its size
    const uint32_t all_ranks = nranks >= 32 ? 0xffffffffu : ((1u << nranks) - 1u);
    std::vector<uint32_t> gmask(nsuper, 0);
    for (sf_long s = 0; s < nsuper; ++s) {
        if (p->phase[s] != 1) continue;
        const uint32_t m = (top_mask && nranks > 1) ? (top_mask[s] & all_ranks) : all_ranks;
        if (!((m >> rank) & 1u)) { delete p; return SF_ERR_ARG; }      // a stored top supernode must list this rank
        gmask[s] = m;
    }
    if (top_mask && nranks > 1) {
        for (sf_long s = 0; s < nsuper; ++s)
            if (top_mask[s] & all_ranks) p->all_masks.push_back(top_mask[s] & all_ranks);
        std::sort(p->all_masks.begin(), p->all_masks.end());
        p->all_masks.erase(std::unique(p->all_masks.begin(), p->all_masks.end()), p->all_masks.end());
    } else if (nranks > 1) {
        p->all_masks.push_back(all_ranks);
    }
    auto group_idx = [&](uint32_t m) { return __builtin_popcount(m & ((1u << rank) - 1u)); };
    std::vector<int64_t> XP(nsuper + 1, -1);
    if (ooc) {
        // two buffers of the largest group's size, then the top; group g's panels from the start of buffer g & 1
        std::vector<int64_t> gsz((size_t)ooc_ngroups, 0);
        for (sf_long s = 0; s < nsuper; ++s)
            if (ooc_group[s] >= 0) gsz[(size_t)ooc_group[s]] += (Super[s + 1] - Super[s]) * (Lsip[s + 1] - Lsip[s]);
        p->ooc_buf = *std::max_element(gsz.begin(), gsz.end());
        std::vector<int64_t> run((size_t)ooc_ngroups);
        for (int g = 0; g < ooc_ngroups; ++g) run[(size_t)g] = (g & 1) * p->ooc_buf;
        int64_t top = 2 * p->ooc_buf;
        p->top_off = top;
        if (ooc_top_mode >= 1) {
            p->ooc_first.assign((size_t)std::max<sf_long>(nsuper, 1), 0);
            p->ooc_last.assign((size_t)std::max<sf_long>(nsuper, 1), 0);
            std::vector<int64_t> toff((size_t)std::max<sf_long>(nsuper, 1), -1);
            std::vector<int32_t> twait((size_t)std::max<sf_long>(nsuper, 1), -1);
            int64_t arena = 0;
            if (ooc_top_mode > 2 || sf::ooc_top_layout(nsuper, Super, SuperMap, Lsip, Lsi, ooc_group, ooc_ngroups, ooc_top_mode, p->ooc_first.data(),
                                                       p->ooc_last.data(), toff.data(), twait.data(), &arena)) {
                delete p;
                return SF_ERR_ARG;
            }
            // the group whose copies have to be over before group g's first launch may write: g - 2 for its buffer, later ones for the
            // places of the top panels that start with it (mode 2)
            p->ooc_wait.assign((size_t)ooc_ngroups, -1);
            for (int g = 0; g < ooc_ngroups; ++g) p->ooc_wait[(size_t)g] = g - 2;
            for (sf_long s = 0; s < nsuper; ++s)
                if (ooc_group[s] < 0) p->ooc_wait[(size_t)p->ooc_first[(size_t)s]] = std::max(p->ooc_wait[(size_t)p->ooc_first[(size_t)s]], twait[(size_t)s]);
            for (sf_long s = 0; s < nsuper; ++s) {
                if (ooc_group[s] >= 0) { int64_t& r = run[(size_t)ooc_group[s]]; XP[s] = r; r += (Super[s + 1] - Super[s]) * (Lsip[s + 1] - Lsip[s]); }
                else XP[s] = top + toff[(size_t)s];
            }
            top += arena;
        } else {
            for (sf_long s = 0; s < nsuper; ++s) {
                int64_t& r = ooc_group[s] >= 0 ? run[(size_t)ooc_group[s]] : top;
                XP[s] = r;
                r += (Super[s + 1] - Super[s]) * (Lsip[s + 1] - Lsip[s]);
            }
        }
        p->ooc_top_mode = ooc_top_mode >= 1 ? ooc_top_mode : 0;
        p->top_size = top - p->top_off;
        p->xC = top;
        XP[nsuper] = top;
    } else {
        int64_t run = 0;
        for (int ph = 0; ph < 2; ++ph) {
            if (ph == 1) p->top_off = run;
            for (sf_long s = 0; s < nsuper; ++s)
                if (p->phase[s] == ph) { XP[s] = run; run += (Super[s + 1] - Super[s]) * (Lsip[s + 1] - Lsip[s]); }
        }
        p->top_size = run - p->top_off;
        p->xC = run;
        XP[nsuper] = run;
    }
    p->h_XP = XP;
    for (sf_long s = 0; s < nsuper; ++s)
        if (p->phase[s] != 0) p->partial = true;
    if (load_top != 1) p->partial = true;
    if (ooc) p->partial = true;         // (also without a top -- a forest: the panels sit at aliased offsets, and there is no resident factor)
    const int64_t ushift = p->xC;     // PU(s) = PL(s) + ushift

    const bool trace_pc = getenv("SF_TRACE") != nullptr;
    auto pc_now = [] { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec / 1e6; };
    const double pc_t0 = pc_now();
    // The factor's allocation (tens of GB: tens of ms in the driver, more on a box's first large allocation) runs on a thread of its
    // own while this one builds the task tables.  + 2 doubles: the GEMM stages row PAIRS with 16-byte loads and may touch 8 bytes
    // past the last panel.
    const size_t xb_factor = (std::max<int64_t>((lu ? 2 : 1) * p->xC, 1) + 2) * sizeof(double);
    hipError_t factor_alloc_err = hipSuccess;
    double* const lent = (!dry && t_offer_ptr && t_offer_bytes >= xb_factor) ? t_offer_ptr : nullptr;
    t_offer_ptr = nullptr; t_offer_bytes = 0;
    p->factor_borrowed = lent != nullptr;
    double* factor_mem = nullptr;           // handed to the plan once everything else has succeeded; freed by the guard otherwise
    // ... and, behind the allocation, the device copies of the symbolic structure (narrowed to 32-bit indices on the way): 150 MB of
    // host loops and pageable H2D copies at 128^3 that depend on the inputs only, i.e. ~35 ms of the first call of a pattern that now
    // run beside the validation and the task tables instead of after them (round 4)
    struct EarlyUploads {
        int64_t *d_Lp = nullptr, *d_Lsip = nullptr, *d_Lsxp = nullptr;
        int32_t *d_Li = nullptr, *d_Super = nullptr, *d_SuperMap = nullptr, *d_Lsi = nullptr;
        size_t bytes = 0;
        int rc = SF_OK;
        void release() {
            void* q[] = {d_Lp, d_Lsip, d_Lsxp, d_Li, d_Super, d_SuperMap, d_Lsi};
            for (void* x : q) if (x) (void)hipFree(x);
            d_Lp = d_Lsip = d_Lsxp = nullptr; d_Li = d_Super = d_SuperMap = d_Lsi = nullptr;
        }
    } early;
    const sf_long nnz_in = Lp[n], isize_in = Lsip[nsuper];
    std::thread factor_alloc([&factor_alloc_err, &factor_mem, &early, lent, xb_factor, device, dry, n, nsuper, nnz_in, isize_in, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li] {
        if (dry) return;
        factor_alloc_err = hipSetDevice(device);
        if (factor_alloc_err == hipSuccess && !lent) factor_alloc_err = hipMalloc((void**)&factor_mem, xb_factor);
        if (factor_alloc_err != hipSuccess) return;
        auto narrow = [](const sf_long* a, sf_long len) { std::vector<int32_t> v((size_t)std::max<sf_long>(len, 0)); for (sf_long k = 0; k < len; ++k) v[(size_t)k] = (int32_t)a[k]; return v; };
        int rc = SF_OK;
        if (!rc) rc = upload(&early.d_Lp, std::vector<int64_t>(Lp, Lp + n + 1), &early.bytes);
        if (!rc) rc = upload(&early.d_Li, narrow(Li, nnz_in), &early.bytes);
        if (!rc) rc = upload(&early.d_Super, narrow(Super, nsuper + 1), &early.bytes);
        if (!rc) rc = upload(&early.d_SuperMap, narrow(SuperMap, n), &early.bytes);
        if (!rc) rc = upload(&early.d_Lsip, std::vector<int64_t>(Lsip, Lsip + nsuper + 1), &early.bytes);
        if (!rc) rc = upload(&early.d_Lsi, narrow(Lsi, isize_in), &early.bytes);
        if (!rc) rc = upload(&early.d_Lsxp, std::vector<int64_t>(Lsxp, Lsxp + nsuper + 1), &early.bytes);
        early.rc = rc;
    });
    struct JoinAlloc {
        std::thread& t; double*& mem; EarlyUploads& e;
        ~JoinAlloc() { if (t.joinable()) t.join(); if (mem) (void)hipFree(mem); e.release(); }
    } join_alloc{factor_alloc, factor_mem, early};
    // ---------------- validate the structure the kernels index with ----------------
    {
        // (row indices and SuperMap: 10^8 entries at 128^3 -- a few threads over ranges of supernodes; the pointer arrays first, alone)
        std::atomic<bool> bad{false};
        for (sf_long s = 0; s < nsuper; ++s) {
            const sf_long nscol = Super[s + 1] - Super[s], nsrow = Lsip[s + 1] - Lsip[s];
            const sf_long want = lu ? nscol * (2 * nsrow - nscol) : nscol * nsrow;
            if (nscol <= 0 || nsrow < nscol || Lsxp[s + 1] - Lsxp[s] != want || nsrow >= (sf_long)0x7fffffff || Lsip[s] < 0) bad = true;
        }
        if (bad || Super[0] != 0 || Super[nsuper] != n) { delete p; return SF_ERR_ARG; }
        auto check_range = [&](sf_long s0, sf_long s1) {
            for (sf_long s = s0; s < s1 && !bad.load(std::memory_order_relaxed); ++s) {
                const sf_long nscol = Super[s + 1] - Super[s], nsrow = Lsip[s + 1] - Lsip[s];
                for (sf_long k = 0; k < nsrow; ++k) {
                    const sf_long g = Lsi[Lsip[s] + k];
                    const bool ok = (k < nscol) ? (g == Super[s] + k) : (g > Lsi[Lsip[s] + k - 1] && g < n);
                    if (!ok) { bad = true; return; }
                }
                // SuperMap indexes level[] / phase[] below: it must be the inverse of Super (foreign arrays: checked, not trusted)
                for (sf_long j = Super[s]; j < Super[s + 1]; ++j)
                    if (SuperMap[j] != s) { bad = true; return; }
            }
        };
        const int64_t work = (int64_t)Lsip[nsuper] + n;
        const int T = work < (1 << 22) ? 1 : (int)std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()));
        if (T == 1) check_range(0, nsuper);
        else {
            std::vector<std::thread> th;
            sf_long s0 = 0;
            for (int t = 0; t < T; ++t) {       // equal shares of the row-index array
                const int64_t upto = (int64_t)Lsip[nsuper] * (t + 1) / T;
                sf_long s1 = (t == T - 1) ? nsuper : (sf_long)(std::upper_bound(Lsip + s0, Lsip + nsuper, (sf_long)upto) - Lsip);
                s1 = std::min(std::max(s1, s0), nsuper);
                th.emplace_back(check_range, s0, s1);
                s0 = s1;
            }
            for (std::thread& t : th) t.join();
        }
        if (bad) { delete p; return SF_ERR_ARG; }
    }

    // ---------------- levels of the supernodal tree ----------------
    std::vector<int> level(nsuper, 0);
    int nlevels = nsuper > 0 ? 1 : 0;
    for (sf_long s = 0; s < nsuper; ++s) {
        const sf_long nscol = Super[s + 1] - Super[s], nsrow = Lsip[s + 1] - Lsip[s];
        if (nscol < nsrow) {
            const sf_long par = SuperMap[Lsi[Lsip[s] + nscol]];
            if (par <= s) { delete p; return SF_ERR_ARG; }   // must be postordered
            level[par] = std::max(level[par], level[s] + 1);
            nlevels = std::max(nlevels, level[par] + 1);
        }
    }
    p->nlevels = nlevels;
    p->level_of = level;
    // sharding consistency: every row of a stored supernode must belong to a stored supernode (its updates have
    // a local target), and a top supernode only has top ancestors
    for (sf_long s = 0; s < nsuper && (phase_in || ooc); ++s) {         // (nothing to check when every supernode is simply stored: 10^8 row indices at 128^3)
        if (p->phase[s] < 0) continue;
        const sf_long nscol = Super[s + 1] - Super[s], nsrow = Lsip[s + 1] - Lsip[s];
        for (sf_long k = nscol; k < nsrow; ++k) {
            // out of core: a streamed supernode only updates panels of its own group (alive in the same buffer) or of the resident top
            if (ooc && ooc_group[s] >= 0) {
                const int32_t ga = ooc_group[SuperMap[Lsi[Lsip[s] + k]]];
                if (ga >= 0 && ga != ooc_group[s]) { delete p; return SF_ERR_ARG; }
            }
            const int8_t pa = p->phase[SuperMap[Lsi[Lsip[s] + k]]];
            if (pa < 0 || (p->phase[s] == 1 && pa != 1)) { delete p; return SF_ERR_ARG; }
        }
    }

    const double pc_t1 = pc_now();
    // ---------------- task tables ----------------
    std::vector<PotrfTask> potrf;
    std::vector<TrsmTask> trsm;
    std::vector<StepTask> steps;
    std::vector<GemmProb> probs;
    std::vector<GemmTask> gtasks;
    std::vector<GemmTask> stasks;           // 64 x 32 tiles of the Schur updates with K <= SU_MAXK (k_update_small)
    int64_t relmap_size = 0;
    std::vector<int64_t> scatter_probs;     // index of the first problem of every (s, a) pair
    // Tiles of one problem are emitted in "supertile" order: blocks of (up to) 8 tile columns x 8 tile rows.
    // The kernel hands each XCD a contiguous run of tasks, so the ~64 workgroups resident on one XCD at a
    // time work on one supertile and march through K together: per K step they touch 8 + 8 operand slices
    // instead of 1 + 64, i.e. each slice is fetched into that XCD's L2 once and re-used 8 times.
    // A task = one tile over one slice of its problem's K range, emitted slice by slice (the kernel deals consecutive tasks
    // out in rounds, so a round would be 512 adjacent tiles over the SAME K range).  Measured at 128^3 on one box: no
    // slicing 556 ms; slices of 512 / 256 / 128 steps 563 / 569 / 569 ms -- every extra slice is one more atomic pass over
    // the target tile, which costs more than the extra rounds gain.  GEMM_SLICE is therefore "never" (one slice).
    const int gemm_slice = sf::GEMM_SLICE;
    // kt_lo / kt_hi: the K steps [kt_lo, kt_hi) of the problem only (look-ahead: the far part and the last block's part of an outer
    // GEMM are separate launches); default = all of K
    auto add_tiles = [&](int32_t prob_id, int M, int N, int K, int kt_lo = 0, int kt_hi = -1) {
        const int tmn = (M + sf::GEMM_BM - 1) / sf::GEMM_BM, tnn = (N + sf::GEMM_BN - 1) / sf::GEMM_BN;
        const int sw = std::min(tnn, 8), sh = std::max(1, 64 / sw);
        const int nkt_all = kt_hi >= 0 ? kt_hi : (K + sf::GEMM_BK - 1) / sf::GEMM_BK;
        for (int k0 = kt_lo; k0 < nkt_all; k0 += gemm_slice)
        for (int sj = 0; sj < tnn; sj += sw)
            for (int si = 0; si < tmn; si += sh)
                for (int tn = sj; tn < std::min(sj + sw, tnn); ++tn)
                    for (int tm = si; tm < std::min(si + sh, tmn); ++tm) {
                        // keep tiles that contain at least one element with ci >= cj
                        if ((tm + 1) * sf::GEMM_BM - 1 < tn * sf::GEMM_BN) continue;
                        gtasks.push_back(GemmTask{prob_id, (uint16_t)tm, (uint16_t)tn, (uint32_t)k0, (uint32_t)std::min(gemm_slice, nkt_all - k0)});
                        p->flops_tiles += 2.0 * sf::GEMM_BM * sf::GEMM_BN * sf::GEMM_BK * (double)std::min(gemm_slice, nkt_all - k0);
                    }
    };

    // download schedule: launch count after which block column jo of supernode s is final (its 64-column chain is done)
    std::vector<int64_t> blk_first(nsuper + 1, 0);
    for (sf_long s = 0; s < nsuper; ++s) blk_first[s + 1] = blk_first[s] + (Super[s + 1] - Super[s] + sf::OUTER_NB - 1) / sf::OUTER_NB;
    std::vector<size_t> blk_ready(blk_first[nsuper], 0);

    int su_maxk = sf::SU_MAXK;              // SF_SU_MAXK: experiment knob
    if (const char* env = sf_exp_env("SF_SU_MAXK")) su_maxk = atoi(env);
    int32_t n_flags = 0;
    // LU fused steps: the step for which panel s's diagonal block last received its pre-update (outer block * 16 + step), see below
    std::vector<int32_t> pre_updated(lu ? (size_t)std::max<sf_long>(nsuper, 1) : 0, -1);
    int64_t max_diag_tasks = 0;     // k_step launches: scratch for the 16 x 16 inverses, 1024 doubles per diagonal task
    // steps of up to this many workgroups run as ONE k_step launch; beyond it (swarms of tiny panels at the bottom levels)
    // the three-launch form -- stream-K GEMM, one-wave POTRF, 256-row TRSM workgroups -- has the better throughput.
    // Measured: 128^3 608 ms at 2048, 603 at 8192, 600 at 16384, 602 unlimited; 2-D 1000^2 (config 3) 12.0 ms up to 16384,
    // 12.3 at 32768, 13.3 unlimited
    // (LU config 5: 91.7 ms at 2048, 91.2 at 16384)
    // SF_FUSE_MAX (debug knob, read at plan creation): 0 forces the three-launch form everywhere, so that both forms
    // stay covered by the tests
    int64_t fuse_max = 32 * sf::GEMM_GRID;
    if (const char* env = getenv("SF_FUSE_MAX")) fuse_max = strtoll(env, nullptr, 10);
    // (The ranks of a group must keep bit-identical copies of a shared panel: an LU threshold pivot decision may not depend on a
    // rank's own rounding.  Everything a chain reads is either an all-reduced sum or computed in a fixed order -- the step kernels,
    // and the look-ahead schedule's replicated near parts, which run k_gemm with whole_tiles: one addition per target element.)
    if (const char* env = getenv("SF_LOOKAHEAD")) p->lookahead = atoi(env) != 0;      // 0: the round-1 schedule (tests cover both)
    if (const char* env = getenv("SF_GRAPH")) p->use_graph = atoi(env) != 0;           // resident factorizations replayed as one hipGraph
    if (const char* env = getenv("SF_TOP_OWNER")) p->top_owner = atoi(env) != 0 && !lu;   // prototype schedule (Cholesky), see Segment::owner_gi
    if (const char* env = sf_exp_env("SF_LOOKAHEAD1")) p->lookahead1 = atoi(env) != 0 && nranks == 1 && !p->partial;
    if (const char* env = sf_exp_env("SF_LOOKAHEAD1_GRID")) p->la_grid = std::max(0, atoi(env));
    int n_la_events = 0;
    // The sweep: one set of independent supernodes at a time -- phase 0: a level of the owned subtrees; phase 1: the top
    // supernodes of one level that share one group of ranks (ascending mask: every rank meets the sets it shares with
    // another rank in the same order, so the groups' collectives cannot wait for each other in a circle).
    struct LevelSet { int ph; std::vector<sf_long> sn; uint32_t mask; int share_idx, share_cnt; double lo = 0.0, hi = 1.0; int group = -1; };
    std::vector<LevelSet> sets;
    if (ooc) {      // out of core: the groups one after the other, each with its own level sets (then the top, below)
        std::vector<std::vector<std::vector<sf_long>>> by_gl((size_t)ooc_ngroups);
        for (sf_long s = 0; s < nsuper; ++s)
            if (ooc_group[s] >= 0) {
                auto& bl = by_gl[(size_t)ooc_group[s]];
                if ((int)bl.size() <= level[s]) bl.resize((size_t)level[s] + 1);
                bl[(size_t)level[s]].push_back(s);
            }
        std::vector<std::vector<std::vector<sf_long>>> top_gl((size_t)ooc_ngroups);      // mode 1: the top supernodes by (last group below, level)
        if (ooc_top_mode >= 1)
            for (sf_long s = 0; s < nsuper; ++s)
                if (ooc_group[s] < 0) {
                    auto& bl = top_gl[(size_t)p->ooc_last[(size_t)s]];
                    if ((int)bl.size() <= level[s]) bl.resize((size_t)level[s] + 1);
                    bl[(size_t)level[s]].push_back(s);
                }
        for (int g = 0; g < ooc_ngroups; ++g) {
            for (auto& v : by_gl[(size_t)g])
                if (!v.empty()) { sets.push_back(LevelSet{0, std::move(v), 0, 0, 1}); sets.back().group = g; }
            for (auto& v : top_gl[(size_t)g])
                if (!v.empty()) sets.push_back(LevelSet{1, std::move(v), all_ranks, 0, 1});
        }
        // (Measured and not kept: every top supernode right after the last group below it -- the lower separators factorized and
        // on their way to the host while later groups still stream.  128^3 with 8 groups: 0.69 s against 0.60 s with the top at the end:
        // a top supernode then runs alone instead of with its level, and it delays the next group, whose download is what the
        // group phase is bound by.)
    }
    for (int ph = ooc ? 1 : 0; ph < 2 && !(ooc && ooc_top_mode >= 1); ++ph) {
        std::vector<std::vector<sf_long>> by_level(nlevels);
        for (sf_long s = 0; s < nsuper; ++s)
            if (p->phase[s] == ph) by_level[level[s]].push_back(s);
        for (int l = 0; l < nlevels; ++l) {
            if (by_level[l].empty()) continue;
            if (ph == 0) { sets.push_back(LevelSet{0, by_level[l], 0, 0, 1}); continue; }
            std::vector<uint32_t> ms;
            for (sf_long s : by_level[l]) ms.push_back(gmask[s]);
            std::sort(ms.begin(), ms.end());
            ms.erase(std::unique(ms.begin(), ms.end()), ms.end());
            for (uint32_t m : ms) {
                LevelSet LS{1, {}, m, group_idx(m), __builtin_popcount(m)};
                LS.lo = (double)LS.share_idx / LS.share_cnt;
                LS.hi = (double)(LS.share_idx + 1) / LS.share_cnt;
                if (root_cum && LS.share_cnt == nranks && nranks > 1) { LS.lo = root_cum[rank]; LS.hi = root_cum[rank + 1]; }
                for (sf_long s : by_level[l])
                    if (gmask[s] == m) LS.sn.push_back(s);
                sets.push_back(std::move(LS));
            }
        }
    }
    p->launch_split = 0;
    bool split_set = false;
    int ooc_open = -1;          // out of core: the group whose (zero + assemble) launch has been issued last
    for (const LevelSet& LS : sets) {
    const int ph = LS.ph;
    if (ph == 1 && !split_set) { p->launch_split = p->launches.size(); split_set = true; }
    // out of core: every group starts by zeroing and assembling its buffer (kind 7, first = the group); groups without supernodes
    // still get theirs, so that the buffers' turn-taking is the same whatever the tree looks like
    while (ooc && LS.group > ooc_open) p->launches.push_back(Launch{7, (int64_t)++ooc_open, 0});
    {
        const std::vector<sf_long>& Sl = LS.sn;
        const bool shared = ph == 1 && nranks > 1;          // additive updates split over the group, block columns reduced
        sf_long maxcol = 0;
        for (sf_long s : Sl) maxcol = std::max(maxcol, Super[s + 1] - Super[s]);
        // Two-level blocking of the in-panel factorization.  Outer block columns of OUTER_NB columns are
        // brought up to date left-looking with ONE large-K GEMM (K = all columns to their left), then
        // factored right-looking in NB-column steps whose trailing update stays inside the outer block.
        // The K = NB updates are HBM-bound read-modify-writes; confining them to OUTER_NB columns cuts
        // their traffic by n / OUTER_NB, the rest of the flops run at large K out of LDS/registers.
        const int nouter = (int)((maxcol + sf::OUTER_NB - 1) / sf::OUTER_NB);
        // The left-looking update of the outer block at column J2: K steps [kt_lo, kt_hi) of its K = J2 columns (all of them:
        // kt_hi < 0), one launch for the panels of the set; `split`: shared among the ranks of the group (its result is part of a
        // sum that is reduced later) or executed in full by every rank (its target has already been reduced)
        auto outer_gemm = [&](int J2, int kt_lo, int kt_hi, bool split, bool count_flops) {
            const int64_t g0 = (int64_t)gtasks.size();
            for (sf_long s : Sl) {
                const int nscol = (int)(Super[s + 1] - Super[s]), nsrow = (int)(Lsip[s + 1] - Lsip[s]);
                if (J2 >= nscol) continue;
                GemmProb g{};
                g.lda = nsrow; g.ldc = nsrow;
                g.M = nsrow - J2; g.N = std::min(sf::OUTER_NB, nscol - J2); g.K = J2;
                const int64_t src = XP[s] + J2;                    // rows J2.. , columns 0..J2-1
                const int64_t dst = XP[s] + J2 + (int64_t)J2 * nsrow;
                const double kk = (kt_hi < 0 ? g.K : (kt_hi - kt_lo) * sf::GEMM_BK);
                const double fl = (double)g.N * (g.N + 1) * kk + 2.0 * (g.M - g.N) * (double)g.N * kk;
                for (int side = 0; side < (lu ? 2 : 1); ++side) {     // side 0: (L) panel, side 1: U^T panel
                    g.y_off = src + (side ? ushift : 0);
                    g.x_off = src + ((lu && !side) ? ushift : 0);
                    g.c_off = dst + (side ? ushift : 0);
                    g.strict = (lu && !side) ? 1 : 0;
                    if (count_flops) { p->flops_outer_gemm += fl; p->flops_panel_gemm += fl; }
                    probs.push_back(g);
                    add_tiles((int32_t)probs.size() - 1, g.M, g.N, g.K, kt_lo, kt_hi);
                }
            }
            if ((int64_t)gtasks.size() > g0) {
                p->launches.push_back(Launch{4, g0, (int)(gtasks.size() - g0)});
                p->launches.back().split = split && LS.share_cnt > 1;
                p->launches.back().whole_tiles = shared && !split && LS.share_cnt > 1;      // replicated inside a group: reproducible
                p->launches.back().share_idx = LS.share_idx; p->launches.back().share_cnt = LS.share_cnt;
                p->launches.back().share_lo = LS.lo; p->launches.back().share_hi = LS.hi;
            }
        };
        // Look-ahead for shared panels (several ranks): the update of block jo is cut into the FAR part (columns of the blocks
        // 0 .. jo-2, issued right after the chain of block jo-2, shared among the ranks) and the part of block jo-1 (issued after
        // the reduce point of block jo, executed in full by every rank: 512 columns of K).  The far part is all a block's sum over
        // the ranks has to wait for, so that sum travels while the chain of block jo-1 runs (sf_chol_plan_factorize_distributed).
        const bool ahead = shared && p->lookahead && LS.share_cnt > 1;
        // ONE GPU (experiment, SF_LOOKAHEAD1, EXP builds only): the same cut, the far part on the plan's second stream -- far(jo+2) needs
        // the chain of block jo only, so it runs beside near(jo+1) + chain(jo+1); events order the two lanes.  MEASURED AND NOT ADOPTED:
        // 546.7 ms against 540.7 at 128^3 with the full side grid, 549.9 / 567.8 / 613.6 with 448 / 384 / 256 workgroups
        // (profiles/r03_m_one_gpu_lookahead.txt; DESIGN 5.2: a persistent GEMM grid leaves no registers for the chain's workgroups)
        const bool ahead1 = !shared && p->lookahead1 && nouter >= 3;
        std::vector<int> chain_ev(ahead1 ? nouter : 0, -1), far_ev(ahead1 ? nouter : 0, -1);
        constexpr int KT_BLOCK = sf::OUTER_NB / sf::GEMM_BK;
        for (int jo = 0; jo < nouter; ++jo) {
            const int J = jo * sf::OUTER_NB;
            if (jo > 0 && !ahead && !ahead1) outer_gemm(J, 0, -1, shared, true);
            if (ahead1 && jo > 0) {
                const size_t l = p->launches.size();
                outer_gemm(J, (jo - 1) * KT_BLOCK, jo * KT_BLOCK, false, true);        // block jo-1 -> block jo (main lane)
                if (p->launches.size() > l && far_ev[jo] >= 0) p->launches.back().wait_ev = far_ev[jo];
            }
            if (shared) {
                // reduce point: block column jo of every panel of the set is complete up to the sum over the group's ranks
                if (!p->segments.empty()) p->segments.back().l1 = p->launches.size();
                Segment sg;
                sg.mask = LS.mask;
                sg.l0 = p->launches.size();
                for (sf_long s : Sl) {
                    const int64_t nscol = Super[s + 1] - Super[s], nsrow = Lsip[s + 1] - Lsip[s];
                    if (J >= nscol) continue;
                    const int64_t w = std::min<int64_t>(sf::OUTER_NB, nscol - J);
                    for (int side = 0; side < (lu ? 2 : 1); ++side) {         // LU: the L panel and the U^T panel
                        const int64_t base = XP[s] + (side ? ushift : 0);
                        sg.off.push_back(base + (int64_t)J * nsrow);
                        sg.cnt.push_back(w * nsrow);
                        sg.src.push_back(base + (int64_t)J * nsrow + J);
                        sg.rows.push_back(nsrow - J);
                        sg.cols.push_back(w);
                        sg.ld.push_back(nsrow);
                        sg.packed += (nsrow - J) * w;
                    }
                }
                sg.early = ahead && jo > 0;
                // owner-computes prototype: the sets EVERY rank takes part in (the root separator), look-ahead schedule only
                if (p->top_owner && ahead && LS.share_cnt == nranks) sg.owner_gi = jo % LS.share_cnt;
                p->segments.push_back(std::move(sg));
                if (ahead && jo > 0) outer_gemm(J, (jo - 1) * KT_BLOCK, jo * KT_BLOCK, false, true);       // block jo-1 -> block jo, replicated
            }
            // Inside the outer block the 64-column steps are LEFT-looking as well: block column t is first
            // updated by the t block columns of this outer block already factored (K = 64 t, written once),
            // then its diagonal block is factored and the rows below are solved.  (A right-looking trailing
            // update would read-modify-write the rest of the outer block at every step with K = 64.)
            const int ninner = sf::OUTER_NB / sf::NB;
            bool block_fused = false;       // decided at the block's first step (the largest), kept for all its steps: the
                                            // fused Cholesky steps push updates into the block's FUTURE diagonal blocks
            for (int ti = 0; ti < ninner; ++ti) {
                const int diag = J + ti * sf::NB;
                if (diag >= maxcol) break;
                int64_t step_wgs = 0;
                for (sf_long s : Sl) {
                    const int64_t nscol = Super[s + 1] - Super[s], nsrow = Lsip[s + 1] - Lsip[s];
                    if (diag < nscol) step_wgs += 1 + (lu ? 2 : 1) * ((nsrow - std::min<int64_t>(nscol, diag + sf::NB) + sf::ST_ROWS - 1) / sf::ST_ROWS);
                }
                if (ti == 0) block_fused = step_wgs <= fuse_max;
                if (block_fused) {
                    // latency-bound step: the (update, POTRF / GETRF, TRSM) triple is ONE launch of k_step.  Steps with
                    // more tiles are throughput-bound and keep the three launches (stream-K GEMM over all tiles,
                    // 256-row TRSM workgroups).
                    const int64_t d0 = (int64_t)steps.size();
                    std::vector<int32_t> flag_of, slot_of;
                    for (sf_long s : Sl) {
                        const int nscol = (int)(Super[s + 1] - Super[s]), nsrow = (int)(Lsip[s + 1] - Lsip[s]);
                        if (diag >= nscol) { flag_of.push_back(-1); slot_of.push_back(-1); continue; }
                        const int b = std::min(sf::NB, nscol - diag);
                        flag_of.push_back(n_flags);
                        // (the two halves of an LU diagonal block's update are tied together here, not by convention: a diagonal task that
                        // applies only the last 64 columns must find the pre-update task of the launch before it)
                        if (lu && ti >= 2 && pre_updated[(size_t)s] != jo * 16 + ti) {
                            fprintf(stderr, "[sparseframe-hip] plan_create: LU step %d of outer block %d of supernode %lld has no pre-update task\n", ti, jo, (long long)s);
                            delete p;
                            return SF_ERR_ARG;
                        }
                        // Cholesky: J = diag (no left-looking update): the diagonal block arrives up to date, see k_step
                        // LU: the diagonal block's own left-looking update is cut in two -- its far part (columns [J, diag - 64): final
                        // before the PREVIOUS step starts) was applied by a pre-update task of that step's launch (below), the
                        // diagonal workgroup only applies the last 64 columns before it factors
                        steps.push_back(StepTask{XP[s], XP[s] + (lu ? ushift : 0), nsrow, lu ? (ti >= 2 ? diag - sf::NB : J) : diag, diag, b, diag, b, n_flags++, 0, (int32_t)(steps.size() - d0), 0, (int32_t)Super[s], 0});
                        slot_of.push_back((int32_t)(steps.size() - 1 - d0));
                        if (ti > 0) p->flops_panel_gemm += (lu ? 2.0 : 1.0) * ((double)b * (b + 1) * (diag - J) + 2.0 * (nsrow - diag - b) * (double)b * (diag - J));
                    }
                    if (lu && ti >= 1 && ti + 1 < ninner) {
                        // pre-update tasks (mode bit 1): the far part of the NEXT step's diagonal-block update, K = [J, diag) -- the
                        // columns of the steps before this one, final when this launch starts.  Right behind the diagonal tasks:
                        // they never wait and are the longest tasks of the launch.
                        const int dnext = diag + sf::NB;
                        for (sf_long s : Sl) {
                            const int nscol = (int)(Super[s + 1] - Super[s]), nsrow = (int)(Lsip[s + 1] - Lsip[s]);
                            if (dnext >= nscol) continue;
                            const int bn = std::min(sf::NB, nscol - dnext);
                            steps.push_back(StepTask{XP[s], XP[s] + ushift, nsrow, J, dnext, bn, dnext, bn, -1, 2, 0, 0, (int32_t)Super[s], 0});
                            pre_updated[(size_t)s] = jo * 16 + ti + 1;
                        }
                    }
                    size_t si = 0;
                    for (sf_long s : Sl) {
                        const int nscol = (int)(Super[s + 1] - Super[s]), nsrow = (int)(Lsip[s + 1] - Lsip[s]);
                        const int32_t fl = flag_of[si], sl = slot_of[si];
                        ++si;
                        if (fl < 0) continue;
                        const int b = std::min(sf::NB, nscol - diag);
                        for (int r = diag + b; r < nsrow; r += sf::ST_ROWS) {
                            const int nr = std::min(sf::ST_ROWS, nsrow - r);
                            if (!lu) {
                                // rows that are a later diagonal block of this outer block get this step's X X^T pushed into them
                                const int nb = (r < std::min(nscol, J + sf::OUTER_NB)) ? std::min(sf::NB, nscol - r) : 0;
                                steps.push_back(StepTask{XP[s], XP[s], nsrow, J, diag, b, r, nr, fl, 0, sl, nb, (int32_t)Super[s], 0});
                            } else {
                                steps.push_back(StepTask{XP[s], XP[s] + ushift, nsrow, J, diag, b, r, nr, fl, 0, sl, 0, (int32_t)Super[s], 0});     // L21 <- (L21 - ..) U11^{-1}
                                steps.push_back(StepTask{XP[s] + ushift, XP[s], nsrow, J, diag, b, r, nr, fl, 1, sl, 0, (int32_t)Super[s], 0});     // U12^T <- (U12^T - ..) L11^{-T}
                            }
                        }
                    }
                    if ((int64_t)steps.size() > d0) {
                        p->launches.push_back(Launch{5, d0, (int)(steps.size() - d0)});
                        p->launches.back().ticket = p->n_tickets++;
                    }
                    max_diag_tasks = std::max<int64_t>(max_diag_tasks, (int64_t)slot_of.size());
                    continue;
                }
                const int64_t p0 = (int64_t)potrf.size(), t0 = (int64_t)trsm.size(), g0 = (int64_t)gtasks.size();
                for (sf_long s : Sl) {
                    const int nscol = (int)(Super[s + 1] - Super[s]), nsrow = (int)(Lsip[s + 1] - Lsip[s]);
                    if (diag >= nscol) continue;
                    const int b = std::min(sf::NB, nscol - diag);
                    if (ti > 0) {
                        GemmProb g{};
                        g.lda = nsrow; g.ldc = nsrow;
                        g.M = nsrow - diag; g.N = b; g.K = diag - J;
                        const int64_t src = XP[s] + diag + (int64_t)J * nsrow;     // rows diag.., columns J..diag-1
                        const int64_t dst = XP[s] + diag + (int64_t)diag * nsrow;
                        for (int side = 0; side < (lu ? 2 : 1); ++side) {
                            g.y_off = src + (side ? ushift : 0);
                            g.x_off = src + ((lu && !side) ? ushift : 0);
                            g.c_off = dst + (side ? ushift : 0);
                            g.strict = (lu && !side) ? 1 : 0;
                            p->flops_panel_gemm += (double)g.N * (g.N + 1) * g.K + 2.0 * (g.M - g.N) * (double)g.N * g.K;
                            probs.push_back(g);
                            add_tiles((int32_t)probs.size() - 1, g.M, g.N, g.K);
                        }
                    }
                    potrf.push_back(PotrfTask{XP[s], nsrow, diag, b, (int32_t)Super[s]});
                    const int below = diag + b;
                    for (int r = below; r < nsrow; r += sf::TRSM_ROWS) {
                        const int nr = std::min(sf::TRSM_ROWS, nsrow - r);
                        if (!lu) {
                            trsm.push_back(TrsmTask{XP[s], XP[s], nsrow, diag, b, r, nr, 0, (int32_t)Super[s], 0});
                        } else {
                            trsm.push_back(TrsmTask{XP[s], XP[s] + ushift, nsrow, diag, b, r, nr, 0, (int32_t)Super[s], 0});            // L21 <- L21 U11^{-1}
                            trsm.push_back(TrsmTask{XP[s] + ushift, XP[s], nsrow, diag, b, r, nr, 1, (int32_t)Super[s], 0});            // U12^T <- U12^T L11^{-T}
                        }
                    }
                }
                if ((int64_t)gtasks.size() > g0) {
                    p->launches.push_back(Launch{2, g0, (int)(gtasks.size() - g0)});
                    // a chain step of a shared panel runs replicated on every rank of its group: no tile split by K, so that each target
                    // element receives ONE addition and the replicas stay bit-identical (LU pivot decisions rest on that)
                    p->launches.back().whole_tiles = shared && LS.share_cnt > 1;
                }
                if ((int64_t)potrf.size() > p0) p->launches.push_back(Launch{0, p0, (int)(potrf.size() - p0)});
                if ((int64_t)trsm.size() > t0) p->launches.push_back(Launch{1, t0, (int)(trsm.size() - t0)});
            }
            const bool owner_block = shared && !p->segments.empty() && p->segments.back().owner_gi >= 0;
            if (owner_block) p->segments.back().lc = p->launches.size();       // near GEMM + chain = the owner's part; then the broadcast
            for (sf_long s : Sl)        // (owner-computes: final on every rank only after the broadcast, i.e. once the next launch is in)
                if (J < Super[s + 1] - Super[s]) blk_ready[blk_first[s] + jo] = p->launches.size() + (owner_block ? 1 : 0);
            if (ahead && jo + 2 < nouter) outer_gemm(J + 2 * sf::OUTER_NB, 0, (jo + 1) * KT_BLOCK, true, true);    // blocks 0 .. jo -> block jo+2
            if (ahead1 && jo + 2 < nouter && !p->launches.empty()) {
                chain_ev[jo] = n_la_events++;
                p->launches.back().rec_ev = chain_ev[jo];                               // the chain's last launch
                const size_t l = p->launches.size();
                outer_gemm(J + 2 * sf::OUTER_NB, 0, (jo + 1) * KT_BLOCK, false, true);
                if (p->launches.size() > l) {
                    far_ev[jo + 2] = n_la_events++;
                    p->launches.back().lane = 1;
                    p->launches.back().wait_ev = chain_ev[jo];
                    p->launches.back().rec_ev = far_ev[jo + 2];
                }
            }
        }
        // Schur updates of every supernode of this level into its ancestors
        const int64_t g0 = (int64_t)gtasks.size(), s0 = (int64_t)stasks.size();
        double level_small_flops = 0;
        // longest K first: the tiles are claimed in list order (k_gemm's dynamic rounds), so the launch ends on its short tiles
        std::vector<sf_long> Su(Sl.begin(), Sl.end());
        std::stable_sort(Su.begin(), Su.end(), [&](sf_long a, sf_long b) { return Super[a + 1] - Super[a] > Super[b + 1] - Super[b]; });
        for (sf_long s : Su) {
            const int nscol = (int)(Super[s + 1] - Super[s]), nsrow = (int)(Lsip[s + 1] - Lsip[s]);
            const sf_long* rows = Lsi + Lsip[s];
            const double nk = nscol;
            p->flops_exec += lu ? ((double)nsrow * nk * nk - nk * nk * nk / 3.0 + (double)(nsrow - nscol) * nk * nk)
                                : (nk * nk * nk / 3.0 + (double)(nsrow - nscol) * nk * nk);
            int i = nscol;
            while (i < nsrow) {
                const sf_long a = SuperMap[rows[i]];
                int e = i;
                while (e < nsrow && SuperMap[rows[e]] == a) ++e;
                const int dn = e - i, dnm = nsrow - i;
                GemmProb g{};
                g.src_rows = Lsip[s] + i;
                const int a_nscol = (int)(Super[a + 1] - Super[a]), a_nsrow = (int)(Lsip[a + 1] - Lsip[a]);
                g.tgt_rows = Lsip[a] + a_nscol;
                g.lda = nsrow; g.ldc = a_nsrow;
                g.M = dnm; g.N = dn; g.K = nscol;
                g.tgt_first_col = (int32_t)Super[a];
                g.tgt_nscol = a_nscol;
                g.tgt_nbelow = a_nsrow - a_nscol;
                g.map_off = relmap_size;            // one relative map per (s, a) pair, shared by the L and U^T sides
                relmap_size += dnm;
                scatter_probs.push_back((int64_t)probs.size());
                for (int side = 0; side < (lu ? 2 : 1); ++side) {
                    g.y_off = XP[s] + i + (side ? ushift : 0);
                    g.x_off = XP[s] + i + ((lu && !side) ? ushift : 0);
                    g.c_off = XP[a] + (side ? ushift : 0);
                    g.strict = (lu && !side) ? 1 : 0;
                    probs.push_back(g);
                    if (g.K <= su_maxk) {
                        // short inner dimension: one wave per 64 x 32 tile (k_update_small); tiles entirely above the diagonal are skipped
                        const int tmn = (g.M + sf::SU_TM - 1) / sf::SU_TM, tnn = (g.N + sf::SU_TN - 1) / sf::SU_TN;
                        for (int tn = 0; tn < tnn; ++tn)
                            for (int tm = 0; tm < tmn; ++tm)
                                if ((tm + 1) * sf::SU_TM - 1 >= tn * sf::SU_TN)
                                    stasks.push_back(GemmTask{(int32_t)probs.size() - 1, (uint16_t)tm, (uint16_t)tn, 0u, 0u});
                    } else {
                        { const double t0f = p->flops_tiles; add_tiles((int32_t)probs.size() - 1, g.M, g.N, g.K); p->flops_tiles_update += p->flops_tiles - t0f; }
                    }
                }
                // executed flops of the tiles' useful part: Cholesky dn(dn+1)dk + 2 dm dn dk; LU twice minus the
                // diagonal the L side skips (the reference's two GEMMs do 2(dn+dm)dn dk + 2 dm dn dk, L:2570-2577)
                const double fl = lu ? (2.0 * (double)dnm * dn * nk + 2.0 * (double)(dnm - dn) * dn * nk)
                                     : ((double)dn * (dn + 1) * nk + 2.0 * (double)(dnm - dn) * dn * nk);
                p->flops_update += fl;
                if (g.K <= su_maxk) { p->flops_update_small += fl; level_small_flops += fl; }
                p->flops_exec += fl;
                p->scatter_elems += lu ? ((double)dnm * dn + (double)(dnm - dn) * dn)
                                       : ((double)dn * (dn + 1) / 2.0 + (double)(dnm - dn) * dn);
                p->n_pairs++;
                i = e;
            }
        }
        if ((int64_t)gtasks.size() > g0) {
            p->launches.push_back(Launch{3, g0, (int)(gtasks.size() - g0)});
            p->launches.back().split = shared && LS.share_cnt > 1;
            p->launches.back().share_idx = LS.share_idx; p->launches.back().share_cnt = LS.share_cnt;
                p->launches.back().share_lo = LS.lo; p->launches.back().share_hi = LS.hi;
        }
        if ((int64_t)stasks.size() > s0) {
            p->launches.push_back(Launch{6, s0, (int)(stasks.size() - s0)});
            p->launches.back().split = shared && LS.share_cnt > 1;
            p->launches.back().share_idx = LS.share_idx; p->launches.back().share_cnt = LS.share_cnt;
                p->launches.back().share_lo = LS.lo; p->launches.back().share_hi = LS.hi;
            p->launches.back().flops = level_small_flops;
        }
    }
    }   // level sets
    if (!split_set) p->launch_split = p->launches.size();
    if (!p->segments.empty()) p->segments.back().l1 = p->launches.size();
    p->n_gemm_tasks = (int64_t)gtasks.size() + (int64_t)stasks.size();

    const double pc_t2 = pc_now();
    // ---------------- device solve schedule (unsharded plans) ----------------
    // per (level, SV_B-column step): a forward launch [diagonal tasks, SV_ROWS-row tiles] and a backward launch [tiles, diagonal tasks]
    std::vector<sf::SolveTask> solve;
    std::vector<int64_t> solveT_list;      // backward diagonal tasks that read a row-major copy of their block (indices into `solve`)
    int64_t solveT_size = 0;
    int32_t n_solve_sync = 0;
    // A mapped plan (create_mapped: one rank's subtrees + the top supernodes above them) gets a schedule too, over the panels it
    // stores -- the distributed solve (sf_chol_plan_solve_distributed, sf_multi.hip):
    //   forward   x[columns of a shared supernode] is summed over its group before the supernode's first step (the right-hand side
    //             is loaded by the group's first rank only, every rank adds the updates of its own descendants); the diagonal
    //             solves and the tiles inside the supernode's own columns are replicated in the group, the tiles that update
    //             ANCESTORS' rows are dealt out over the group (their sums meet again at the ancestor's reduce point);
    //   backward  no communication: every rank of a group solves the group's supernodes in full (it holds the panels and, by
    //             then, the solution of all their ancestors).
    const bool solve_mapped = p->partial && top_mask != nullptr && load_top == 2;
    if (!p->partial || solve_mapped) {
        std::vector<std::vector<sf_long>> by_level(nlevels);
        for (sf_long s = 0; s < nsuper; ++s)
            if (XP[s] >= 0) by_level[level[s]].push_back(s);
        auto shared_sn = [&](sf_long s) { return solve_mapped && p->phase[s] == 1 && __builtin_popcount(gmask[s]) > 1; };
        if (solve_mapped) {
            for (sf_long s = 0; s < nsuper; ++s) {
                if (XP[s] < 0) continue;
                const bool first = p->phase[s] == 0 || group_idx(gmask[s]) == 0;
                if (first) {
                    if (!p->solve_own.empty() && p->solve_own.back().second == Super[s]) p->solve_own.back().second = Super[s + 1];
                    else p->solve_own.push_back({Super[s], Super[s + 1]});
                }
            }
            p->solve_load = p->solve_own;
        }
        const int tile = sf::SV_ROWS;
        // default: fused (35.3 ms at 128^3); SF_SOLVE_BWD_FUSED=0: two launches per backward step (38.2 ms)
        const bool solve_diagT = !(sf_exp_env("SF_SOLVE_DIAGT") && atoi(sf_exp_env("SF_SOLVE_DIAGT")) == 0);
        const bool bwd_ahead_env = !(sf_exp_env("SF_SOLVE_BWD_AHEAD") && atoi(sf_exp_env("SF_SOLVE_BWD_AHEAD")) == 0);
        const bool bwd_fused = !(sf_exp_env("SF_SOLVE_BWD_FUSED") && atoi(sf_exp_env("SF_SOLVE_BWD_FUSED")) == 0);
        p->solve_bwd_fused = bwd_fused;
        const bool bwd_ahead = bwd_ahead_env && bwd_fused;
        const bool fwd_ahead = !(sf_exp_env("SF_SOLVE_FWD_AHEAD") && atoi(sf_exp_env("SF_SOLVE_FWD_AHEAD")) == 0);
        const bool fwd_far_first = !(sf_exp_env("SF_SOLVE_FWD_FAR_FIRST") && atoi(sf_exp_env("SF_SOLVE_FWD_FAR_FIRST")) == 0);
        int solve_far_wgs = 512;
        if (const char* env = sf_exp_env("SF_SOLVE_FAR_WGS")) solve_far_wgs = std::max(1, atoi(env));
        int solve_far_groups = 8;
        if (const char* env = sf_exp_env("SF_SOLVE_FAR_GROUPS")) solve_far_groups = std::max(1, std::min(64, atoi(env)));
        for (int l = 0; l < nlevels; ++l) {
            sf_long maxcol = 0;
            for (sf_long s : by_level[l]) maxcol = std::max(maxcol, Super[s + 1] - Super[s]);
            // the narrow panels of the level (nscol <= 64; the swarm levels consist of nothing else) go to the
            // one-wave-per-supernode kernels as a step of their own, the wide ones through the general 256-column steps
            std::vector<sf_long> narrow, wide;
            for (sf_long s : by_level[l]) ((Super[s + 1] - Super[s] <= sf::NB && !shared_sn(s)) ? narrow : wide).push_back(s);
            if (narrow.size() < 64) { wide = by_level[l]; narrow.clear(); }     // not worth a launch of their own
            // sums that precede this level's first step: the columns of its shared supernodes, each inside its group
            const int red_first = (int)p->solve_reduces.size();
            for (sf_long s : wide)
                if (shared_sn(s)) p->solve_reduces.push_back(sf_chol_plan::SolveReduce{Super[s], Super[s + 1] - Super[s], gmask[s]});
            int red_count = (int)p->solve_reduces.size() - red_first;
            if (!narrow.empty()) {
                sf_chol_plan::SolveStep st{};
                st.small = 1;
                st.fwd_first = st.bwd_first = (int64_t)solve.size();
                for (sf_long s : narrow) {
                    const int nscol = (int)(Super[s + 1] - Super[s]), nsrow = (int)(Lsip[s + 1] - Lsip[s]);
                    solve.push_back(sf::SolveTask{XP[s], Lsip[s], nsrow, 0, nscol, 0, 0, (int32_t)Super[s], 0, 0});
                }
                st.count = st.ndiag = st.fwd_count = (int)narrow.size();
                st.red_first = red_first; st.red_count = 0;
                p->solve_steps.push_back(st);
            }
            maxcol = 0;
            for (sf_long s : wide) maxcol = std::max(maxcol, Super[s + 1] - Super[s]);
            // Backward look-ahead: the sum a diagonal task waits for, S_J = sum over the row blocks I > J of L_IJ^T x_I, only
            // needs x_{J+1} -- solved one launch earlier -- for the tiles whose rows lie in block J + 1 ("near").  All other tiles of
            // step J ("far": rows two blocks further down, or below the panel) read values that were final a launch earlier, so
            // they ride in the launch of step J + 1, behind its diagonal task, and the diagonal task of step J waits for its few
            // near tiles only.  (Forward sweep: unchanged.  SF_SOLVE_BWD_AHEAD=0 or the two-launch form: every tile is near.)
            // The forward sweep looks ahead the same way: the tiles of step J whose rows lie two blocks further down (or below the
            // panel) are not needed by the next diagonal task, so they ride in the launch of step J + 1 -- there their flag (step J's,
            // set one launch earlier, never reset within a solve) is already up and they run at once, next to that launch's
            // diagonal task, instead of holding a CU slot while they spin on it.  (SF_SOLVE_FWD_AHEAD=0: off.)
            std::vector<sf::SolveTask> pending_far;        // far tiles of the step built in the previous iteration
            std::vector<sf::SolveTask> pending_far_fwd;
            // Far tiles are merged into tasks of up to 8 consecutive 64-row groups of one panel (k_solve_*: one workgroup streams
            // through them; backward: ONE butterfly and ONE set of atomics per task instead of one per 64 rows, on the 256 words
            // all tiles of a step add to), as long as that leaves ~512 workgroups to the launch.  SF_SOLVE_FAR_GROUPS=1: off.
            auto group_far = [&](std::vector<sf::SolveTask>& v) {
                const int G = std::max(1, std::min(solve_far_groups, (int)(v.size() / (size_t)solve_far_wgs)));
                if (G <= 1 || v.empty()) return false;
                std::vector<sf::SolveTask> out;
                for (const sf::SolveTask& t : v) {
                    if (!out.empty() && out.back().panel == t.panel && out.back().diag == t.diag && out.back().flag == t.flag &&
                        out.back().row0 + out.back().nrows == t.row0 && out.back().nrows % tile == 0 && out.back().nrows + t.nrows <= G * tile)
                        out.back().nrows += t.nrows;
                    else out.push_back(t);
                }
                v.swap(out);
                return true;
            };
            for (int diag = 0; diag < maxcol; diag += sf::SV_B) {
                sf_chol_plan::SolveStep st{};
                std::vector<sf::SolveTask> dg, rows, rows_fwd, far_next, far_next_fwd;
                for (sf_long s : wide) {
                    const int nscol = (int)(Super[s + 1] - Super[s]), nsrow = (int)(Lsip[s + 1] - Lsip[s]);
                    if (diag >= nscol) continue;
                    const int b = std::min(sf::SV_B, nscol - diag);
                    const bool sh = shared_sn(s);
                    // a shared supernode's tiles are cut at the end of its own columns: the ones inside are replicated in the
                    // group, the ones below (ancestors' rows) are dealt out in the forward sweep
                    int ntiles = 0;
                    const int cut = sh ? std::max(nscol, diag + b) : nsrow;
                    const int g = sh ? __builtin_popcount(gmask[s]) : 1, gi = sh ? group_idx(gmask[s]) : 0;
                    int below = 0;
                    // (a supernode with no later block in this level has nothing to look ahead to: all its tiles are near)
                    const int near_end = (bwd_ahead && nscol > diag + b) ? std::min(nscol, diag + b + sf::SV_B) : nsrow;
                    const int near_end_fwd = (fwd_ahead && nscol > diag + b) ? std::min(nscol, diag + b + sf::SV_B) : nsrow;
                    for (int r0 = diag + b, r1 = cut; r0 < nsrow; r0 = r1, r1 = nsrow) {
                        for (int r = r0; r < r1; r += tile) {
                            const sf::SolveTask t{XP[s], Lsip[s], nsrow, diag, b, r, std::min(tile, r1 - r), (int32_t)Super[s], n_solve_sync, 0};
                            if (r < near_end) { rows.push_back(t); ++ntiles; }
                            else { far_next.push_back(t); far_next.back().flag = -1; }       // counts for nobody: a scratch word, set below
                            if (!sh || r < nscol || (below++ % g) == gi) ((fwd_ahead && r >= near_end_fwd) ? far_next_fwd : rows_fwd).push_back(t);
                        }
                        if (r1 >= nsrow) break;
                    }
                    // sync words: [flag] forward "solved" flag, [flag + 1] backward tile counter
                    dg.push_back(sf::SolveTask{XP[s], Lsip[s], nsrow, diag, b, 0, 0, (int32_t)Super[s], n_solve_sync, ntiles});
                    n_solve_sync += 2;
                }
                st.fwd_first = (int64_t)solve.size();
                solve.insert(solve.end(), dg.begin(), dg.end());
                // order of the forward list = order in which workgroups take their tasks: diagonal tasks, then the far tiles of the
                // step before (nothing to wait for: they start at once), the near tiles last -- taken first they would sit in
                // the CU slots spinning on the diagonal tasks' flags and keep the far tiles out until those are up
                const bool grouped_fwd = group_far(pending_far_fwd);
                if (fwd_far_first) solve.insert(solve.end(), pending_far_fwd.begin(), pending_far_fwd.end());
                solve.insert(solve.end(), rows_fwd.begin(), rows_fwd.end());
                if (!fwd_far_first) solve.insert(solve.end(), pending_far_fwd.begin(), pending_far_fwd.end());
                st.fwd_count = (int)(dg.size() + rows_fwd.size() + pending_far_fwd.size());
                pending_far_fwd.swap(far_next_fwd);
                st.red_first = red_first; st.red_count = red_count;
                red_count = 0;                      // the sums belong to the level's first step
                st.bwd_first = (int64_t)solve.size();
                for (sf::SolveTask t : rows) { t.flag += 1; solve.push_back(t); }
                // backward: one launch, the diagonal task waits for a tile counter (SF_SOLVE_BWD_FUSED=0: the row tiles and
                // the diagonal tasks as TWO launches -- measured slower)
                for (sf::SolveTask t : dg) {
                    t.flag += 1;
                    if (!bwd_fused) t.expect = 0;
                    // the steps of the top levels (few supernodes per level: their chain of diagonal tasks IS the backward
                    // sweep's critical path) read their diagonal block from a row-major copy: coalesced instead of one cache line
                    // per lane (128^3: backward sweep 19.8 -> see DESIGN 8; SF_SOLVE_DIAGT=0: off)
                    if (solve_diagT && wide.size() <= 16 && t.b > sf::NB) {
                        t.tdiag = 1 + solveT_size;
                        solveT_size += (int64_t)t.b * t.b;
                        solveT_list.push_back((int64_t)solve.size());
                    }
                    solve.push_back(t);
                }
                // the far tiles of the step before this one (lower column block): their rows' x is final when this launch starts
                const bool grouped_bwd = group_far(pending_far);
                solve.insert(solve.end(), pending_far.begin(), pending_far.end());
                st.count = (int)(dg.size() + rows.size() + pending_far.size());
                pending_far.swap(far_next);
                st.nrows_tasks = (int)rows.size();
                st.big = 0;
                for (const sf::SolveTask& t : dg) st.big |= t.b > sf::NB;
                if (grouped_bwd || grouped_fwd) st.big = 1;     // only that instantiation of the kernels walks through row groups
                st.small = 0;
                st.ndiag = (int)dg.size();
                p->solve_steps.push_back(st);
            }
        }
    }
    for (sf::SolveTask& t : solve)
        if (t.flag < 0) t.flag = n_solve_sync;        // the scratch word the far tiles count on
    p->n_solve_sync = n_solve_sync + 1;

    // K-step prefix of every GEMM launch (stream-K work distribution, see k_gemm)
    std::vector<uint32_t> ktprefix;
    for (Launch& L : p->launches) {
        if (L.kind < 2 || L.kind > 4) continue;
        L.prefix_first = (int64_t)ktprefix.size();
        uint64_t run = 0;
        int32_t last_prob = -1;
        for (int k = 0; k < L.count; ++k) {
            ktprefix.push_back((uint32_t)run);
            const int32_t pi = gtasks[L.first + k].prob;
            const GemmProb& g = probs[pi];
            run += (uint64_t)gtasks[L.first + k].nkt;
            if (pi != last_prob) {      // tasks of one problem are contiguous
                L.flops += (double)g.N * (g.N + 1) * g.K + 2.0 * (double)(g.M - g.N) * g.N * g.K;
                last_prob = pi;
            }
        }
        if (run >= (uint64_t)0x7fffffff) { delete p; return SF_ERR_ARG; }   // one launch holds < 2^31 units
        ktprefix.push_back((uint32_t)run);
        L.units = (uint32_t)run;
    }

    std::vector<sf::FillTile> fill_tiles;
    // ---------------- download schedule (sf_chol_plan_factorize_to_host) ----------------
    // Block columns in host order, merged while contiguous in the host layout (Cholesky: and in the device layout) up to
    // one staging slot, larger runs cut into slot-sized pieces.  Top panels of a sharded plan are identical on every
    // rank once factored: their pieces are dealt out over the ranks so that every PCIe link carries a share.
    {
        if (const char* env = getenv("SF_DL_WORKERS")) p->dl_workers = p->dl_workers_fresh = std::max(1, std::min(DL_WORKERS_MAX, atoi(env)));
        if (const char* env = sf_exp_env("SF_DL_HOST_WAIT")) p->dl_host_wait = atoi(env) != 0;
        if (const char* env = getenv("SF_DL_SLOT_MB")) p->dl_slot = (int64_t)std::max(1, atoi(env)) << 17;
        const int64_t DL_SLOT = p->dl_slot;
        bool dl_2d = true;
        if (const char* env = sf_exp_env("SF_DL_2D")) dl_2d = atoi(env) != 0;
        std::vector<DlPiece> runs;
        std::vector<uint32_t> run_mask;
        int last_phase = -2;
        uint32_t last_mask = 0;
        int32_t last_group = -2;        // out of core: a piece never spans two groups (each group's copy is counted on its own)
        auto grp = [&](sf_long s) { return ooc ? ooc_group[s] : (int32_t)-1; };
        // LU: the reference keeps one packed panel per supernode, column j = [ L11 \ U11 (nscol) | L21 | U12^T ] (L:2514-2517), the
        // device an L panel and a U^T panel.  DIRECT form (default): the pieces are whole columns; the compute stream writes U11 into
        // the (unused) upper triangle of the L panel's diagonal block before a piece's event (k_lu_fill_u11), the L and U^T runs
        // travel as they are and the copy worker interleaves them column by column.  SF_DL_LU_PACK=1 (and any matrix with a
        // packed column longer than a staging slot): the older form, a gather kernel per piece on the worker's stream -- which has
        // to find free CUs next to the factorization's persistent kernels and made the LU struct call 2x its resident step.
        bool lu_direct = lu;
        if (const char* env = sf_exp_env("SF_DL_LU_PACK"))
            if (atoi(env) != 0) lu_direct = false;
        for (sf_long s = 0; s < nsuper && lu_direct; ++s)
            if (XP[s] >= 0 && 2 * (Lsip[s + 1] - Lsip[s]) - (Super[s + 1] - Super[s]) > DL_SLOT) lu_direct = false;
        p->dl_lu_direct = lu_direct;
        for (sf_long s = 0; s < nsuper; ++s) {
            if (XP[s] < 0) continue;
            const int64_t nscol = Super[s + 1] - Super[s], nsrow = Lsip[s + 1] - Lsip[s];
            const int64_t hld = lu ? 2 * nsrow - nscol : nsrow;
            if (lu_direct) {
                const int64_t cntL = nsrow * nscol;
                if (2 * cntL <= DL_SLOT) {                       // a whole supernode, merged with its neighbours while they fit a slot
                    size_t ready = 0;
                    for (int64_t jo = 0; jo * sf::OUTER_NB < nscol; ++jo) ready = std::max(ready, blk_ready[blk_first[s] + jo]);
                    DlPiece pc{XP[s], Lsxp[s], nscol * hld, ready, 0};
                    pc.s0 = (int32_t)s; pc.s1 = (int32_t)s + 1; pc.dev_count = cntL;
                    const bool can_merge = !runs.empty() && runs.back().s0 >= 0 && runs.back().ld == 0 && runs.back().s1 == (int32_t)s &&
                                           last_phase == p->phase[s] && last_mask == gmask[s] && last_group == grp(s) &&
                                           runs.back().dev_off + runs.back().dev_count == pc.dev_off &&
                                           runs.back().host_off + runs.back().count == pc.host_off &&
                                           2 * (runs.back().dev_count + cntL) <= DL_SLOT;
                    if (can_merge) {
                        runs.back().count += pc.count;
                        runs.back().dev_count += cntL;
                        runs.back().s1 = (int32_t)s + 1;
                        runs.back().ready = std::max(runs.back().ready, pc.ready);
                    } else {
                        pc.ev = p->phase[s];
                        runs.push_back(pc);
                        run_mask.push_back(gmask[s]);
                    }
                } else {
                    for (int64_t jo = 0; jo * sf::OUTER_NB < nscol; ++jo) {     // block columns, cut to slot size below
                        const int64_t J = jo * sf::OUTER_NB, w = std::min<int64_t>(sf::OUTER_NB, nscol - J);
                        DlPiece pc{XP[s] + J * nsrow, Lsxp[s] + J * hld, w * hld, blk_ready[blk_first[s] + jo], 0};
                        pc.s0 = (int32_t)s; pc.s1 = (int32_t)s + 1; pc.j0 = J; pc.ncols = w; pc.ld = nsrow; pc.dev_count = w * nsrow;
                        pc.ev = p->phase[s];
                        runs.push_back(pc);
                        run_mask.push_back(gmask[s]);
                    }
                }
                last_phase = p->phase[s];
                last_mask = gmask[s];
                last_group = grp(s);
                continue;
            }
            for (int64_t jo = 0; jo * sf::OUTER_NB < nscol; ++jo) {
                const int64_t J = jo * sf::OUTER_NB, w = std::min<int64_t>(sf::OUTER_NB, nscol - J);
                DlPiece pc{XP[s] + J * nsrow, Lsxp[s] + J * hld, w * hld, blk_ready[blk_first[s] + jo], 0};
                // Cholesky, block columns right of the first: rows [0, J) of these columns are zeros on both sides -- 7 % of the
                // factor at 128^3 -- and stay off the link (SF_DL_2D=0: copy them like everything else)
                const bool two_d = !lu && J > 0 && dl_2d;
                if (two_d) { pc.skip = J; pc.ld = nsrow; pc.ncols = w; pc.count = w * (nsrow - J); }
                const bool can_merge = !two_d && !runs.empty() && runs.back().ld == 0 && last_phase == p->phase[s] && last_mask == gmask[s] &&
                                       last_group == grp(s) &&
                                       runs.back().host_off + runs.back().count == pc.host_off &&
                                       (lu || runs.back().dev_off + runs.back().count == pc.dev_off) &&
                                       runs.back().count + pc.count <= DL_SLOT;
                if (can_merge) {
                    runs.back().count += pc.count;
                    runs.back().ready = std::max(runs.back().ready, pc.ready);
                } else {
                    pc.ev = p->phase[s];        // phase kept here until the pieces are dealt out below
                    runs.push_back(pc);
                    run_mask.push_back(gmask[s]);
                }
                last_phase = p->phase[s];
                last_mask = gmask[s];
                last_group = grp(s);
            }
        }
        std::vector<std::pair<uint32_t, int64_t>> seen_by_mask;     // pieces of a group's panels are dealt out inside the group
        for (size_t ri = 0; ri < runs.size(); ++ri) {
            const DlPiece& r = runs[ri];
            if (r.ev == 1 && nranks > 1) {
                const uint32_t m = run_mask[ri];
                size_t k = 0;
                while (k < seen_by_mask.size() && seen_by_mask[k].first != m) ++k;
                if (k == seen_by_mask.size()) seen_by_mask.push_back({m, 0});
                const int64_t seq = seen_by_mask[k].second++;
                if ((int)(seq % __builtin_popcount(m)) != group_idx(m)) continue;
            }
            if (r.s0 >= 0) {        // LU, direct form
                if (r.ld == 0) { p->dl_pieces.push_back(r); p->dl_pieces.back().ev = 0; continue; }
                const int64_t hld = 2 * r.ld - (Super[r.s0 + 1] - Super[r.s0]), cper = std::max<int64_t>(1, DL_SLOT / hld);
                for (int64_t c = 0; c < r.ncols; c += cper) {
                    DlPiece q = r;
                    q.ev = 0;
                    q.j0 = r.j0 + c; q.ncols = std::min(cper, r.ncols - c);
                    q.dev_off = r.dev_off + c * r.ld; q.host_off = r.host_off + c * hld; q.count = q.ncols * hld; q.dev_count = q.ncols * r.ld;
                    p->dl_pieces.push_back(q);
                }
                continue;
            }
            if (r.ld > 0) {         // 2-D piece: cut by whole columns
                const int64_t rows = r.ld - r.skip, cper = std::max<int64_t>(1, DL_SLOT / std::max<int64_t>(rows, 1));
                if (rows > DL_SLOT) {       // one column longer than a slot (never with the default 32 MiB slot): plain pieces
                    for (int64_t o = 0; o < r.ncols * r.ld; o += DL_SLOT)
                        p->dl_pieces.push_back(DlPiece{r.dev_off + o, r.host_off + o, std::min(DL_SLOT, r.ncols * r.ld - o), r.ready, 0});
                    continue;
                }
                for (int64_t c = 0; c < r.ncols; c += cper) {
                    DlPiece q{r.dev_off + c * r.ld, r.host_off + c * r.ld, std::min(cper, r.ncols - c) * rows, r.ready, 0};
                    q.skip = r.skip; q.ld = r.ld; q.ncols = std::min(cper, r.ncols - c);
                    p->dl_pieces.push_back(q);
                }
                continue;
            }
            for (int64_t o = 0; o < r.count; o += DL_SLOT)
                p->dl_pieces.push_back(DlPiece{r.dev_off + o, r.host_off + o, std::min(DL_SLOT, r.count - o), r.ready, 0});
        }
        std::stable_sort(p->dl_pieces.begin(), p->dl_pieces.end(), [](const DlPiece& a, const DlPiece& b) { return a.ready < b.ready; });
        if (ooc) {
            // the group of every piece (from the supernode its first host entry belongs to) and the number of pieces per group: what
            // the launch that re-uses a buffer waits for
            p->dl_group_pieces.assign((size_t)ooc_ngroups, 0);
            for (DlPiece& pc : p->dl_pieces) {
                const sf_long s = (sf_long)(std::upper_bound(Lsxp, Lsxp + nsuper + 1, (sf_long)pc.host_off) - Lsxp) - 1;
                pc.group = (s >= 0 && s < nsuper) ? (ooc_group[s] >= 0 ? ooc_group[s] : (ooc_top_mode >= 1 ? p->ooc_last[(size_t)s] : -1)) : -1;
                if (pc.group >= 0) ++p->dl_group_pieces[(size_t)pc.group];
            }
        }
        for (DlPiece& pc : p->dl_pieces) {
            if (p->dl_ev_ready.empty() || p->dl_ev_ready.back() != pc.ready) p->dl_ev_ready.push_back(pc.ready);
            pc.ev = (int)p->dl_ev_ready.size() - 1;
        }
        if (lu_direct) {            // the U11 fill tiles of every piece, grouped by the piece's event
            std::vector<std::vector<sf::FillTile>> by_ev(p->dl_ev_ready.size());
            for (const DlPiece& pc : p->dl_pieces)
                for (int32_t s = pc.s0; s < pc.s1; ++s) {
                    const int32_t nscol = (int32_t)(Super[s + 1] - Super[s]), nsrow = (int32_t)(Lsip[s + 1] - Lsip[s]);
                    const int32_t cb = pc.ld > 0 ? (int32_t)pc.j0 : 0, ce = pc.ld > 0 ? (int32_t)(pc.j0 + pc.ncols) : nscol;
                    for (int32_t c0 = cb / 64 * 64; c0 < ce; c0 += 64)
                        for (int32_t r0 = 0; r0 <= c0; r0 += 64) by_ev[pc.ev].push_back(sf::FillTile{XP[s], nsrow, r0, c0, cb, ce});
                }
            p->fill_first.assign(1, 0);
            for (const auto& v : by_ev) {
                fill_tiles.insert(fill_tiles.end(), v.begin(), v.end());
                p->fill_first.push_back((int64_t)fill_tiles.size());
            }
        }
    }

    const double pc_t3 = pc_now();
    // ---------------- upload ----------------
    std::vector<int32_t> Super32(nsuper + 1);
    for (sf_long k = 0; k <= nsuper; ++k) Super32[k] = (int32_t)Super[k];
    std::vector<int64_t> Lsip64(Lsip, Lsip + nsuper + 1), Lsxp64(Lsxp, Lsxp + nsuper + 1);
    p->h_Lsip = Lsip64; p->h_Lsxp = Lsxp64; p->h_Super = Super32;
    std::vector<int64_t> Up64;
    std::vector<int32_t> Ui32;
    if (lu && !p->u_alias) {
        Up64.assign(Up, Up + n + 1);
        Ui32.resize(p->unz);
        for (sf_long k = 0; k < p->unz; ++k) Ui32[k] = (int32_t)Ui[k];
    }

    int rc = SF_OK;
    // dry plans count the bytes a real plan would allocate and touch nothing
    auto up = [&](auto** dptr, const auto& h) -> int {
        if (!dry) return upload(dptr, h, &p->bytes_device);
        *dptr = nullptr;
        p->bytes_device += std::max<size_t>(h.size(), 1) * sizeof(typename std::decay_t<decltype(h)>::value_type);
        return SF_OK;
    };
    auto dalloc = [&](void** ptr, size_t bytes) -> bool {
        if (dry) { *ptr = nullptr; return true; }
        return hipMalloc(ptr, bytes) == hipSuccess;
    };
    do {
        // The plan's streams are HIGH-PRIORITY streams: the runtime multiplexes all streams of one priority over a few hardware
        // queues (4 by default), and the copy-back workers bring six streams of their own.  Whenever the compute stream landed on
        // a hardware queue together with one of those, its kernels queued up behind the worker's event waits and copies and a
        // struct call took 0.89 s instead of 0.555 s (bimodal, about one call in four; always with GPU_MAX_HW_QUEUES=2, never
        // with 12 -- `tools/struct_mode_ab.sh`).  High-priority streams get hardware queues of their own.  SF_STREAM_PRIORITY=0:
        // ordinary streams.
        int prio_least = 0, prio_greatest = 0;
        bool prio = true;
        if (const char* env = sf_exp_env("SF_STREAM_PRIORITY")) prio = atoi(env) != 0;
        if (dry) prio = false;
        if (prio && (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess || prio_greatest >= prio_least)) {
            (void)hipGetLastError();
            prio = false;
        }
        auto new_stream = [&](hipStream_t* st, unsigned flags) {
            if (prio && hipStreamCreateWithPriority(st, flags, prio_greatest) == hipSuccess) return true;
            (void)hipGetLastError();
            return hipStreamCreateWithFlags(st, flags) == hipSuccess;
        };
        if (!dry && (!new_stream(&p->stream, hipStreamDefault) || hipEventCreate(&p->ev0) != hipSuccess ||
            hipEventCreate(&p->ev1) != hipSuccess || hipEventCreate(&p->ev_s0) != hipSuccess ||
            hipEventCreate(&p->ev_s1) != hipSuccess)) { rc = SF_ERR_HIP; break; }
        p->dl_events.assign(p->dl_ev_ready.size(), nullptr);
        if (!dry) {
            bool ok = true;
            for (hipEvent_t& e : p->dl_events) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync) == hipSuccess;    // the copy workers sleep on them
            if (!ok) { rc = SF_ERR_HIP; break; }
        }
        // the structure arrays were uploaded by the helper thread (behind the factor's allocation); a schedule-only plan counts them
        if (dry) {
            p->bytes_device += (size_t)(n + 1) * 8 + (size_t)std::max<int64_t>(p->nnz, 1) * 4 + (size_t)(nsuper + 1) * 4 + (size_t)std::max<sf_long>(n, 1) * 4
                               + (size_t)(nsuper + 1) * 8 + (size_t)std::max<int64_t>(p->isize, 1) * 4 + (size_t)(nsuper + 1) * 8;
        } else {
            factor_alloc.join();
            if (factor_alloc_err != hipSuccess) { (void)hipGetLastError(); rc = SF_ERR_ALLOC; break; }
            if (early.rc) { rc = early.rc; break; }
            p->d_Lp = early.d_Lp; p->d_Li = early.d_Li; p->d_Super = early.d_Super; p->d_SuperMap = early.d_SuperMap;
            p->d_Lsip = early.d_Lsip; p->d_Lsi = early.d_Lsi; p->d_Lsxp = early.d_Lsxp;
            early.d_Lp = early.d_Lsip = early.d_Lsxp = nullptr; early.d_Li = early.d_Super = early.d_SuperMap = early.d_Lsi = nullptr;
            p->bytes_device += early.bytes;
        }
        if ((rc = up(&p->d_potrf, potrf))) break;
        if ((rc = up(&p->d_trsm, trsm))) break;
        if ((rc = up(&p->d_steps, steps))) break;
        {
            p->n_flags = std::max<int32_t>(n_flags, 1);
            std::vector<int> zeros(std::max<int32_t>(n_flags, 1), 0);
            if ((rc = up(&p->d_flags, zeros))) break;
            const size_t tb = (size_t)std::max<int64_t>(max_diag_tasks, 1) * (lu ? 2048 : 1024) * sizeof(double);
            if (!dalloc((void**)&p->d_tinv, tb)) { rc = SF_ERR_ALLOC; break; }
            p->bytes_device += tb;
        }
        if ((rc = up(&p->d_probs, probs))) break;
        if ((rc = up(&p->d_gtasks, gtasks))) break;
        if ((rc = up(&p->d_stasks, stasks))) break;
        if ((rc = up(&p->d_ktprefix, ktprefix))) break;
        {   // relative maps of all Schur updates, built on the device (createRelativeMap, CK:42-60, once per plan)
            const size_t mb = (size_t)std::max<int64_t>(relmap_size, 1) * sizeof(int32_t);
            if (!dalloc((void**)&p->d_relmap, mb)) { rc = SF_ERR_ALLOC; break; }
            p->bytes_device += mb;
            if (!dry) {
                std::vector<GemmProb> firsts;
                firsts.reserve(scatter_probs.size());
                for (int64_t k : scatter_probs) firsts.push_back(probs[k]);
                GemmProb* d_firsts = nullptr;
                size_t dummy = 0;
                if ((rc = upload(&d_firsts, firsts, &dummy))) break;
                sf::launch_build_relmaps(d_firsts, (int)firsts.size(), p->d_Lsi, p->d_relmap, p->stream);
                const hipError_t e1 = hipStreamSynchronize(p->stream), e2 = hipGetLastError();
                (void)hipFree(d_firsts);
                if (e1 != hipSuccess || e2 != hipSuccess) { rc = SF_ERR_HIP; break; }
            }
        }
        if (!solve.empty()) {
            if ((rc = up(&p->d_solve, solve))) break;
            if (!solveT_list.empty()) {
                if ((rc = up(&p->d_solveT_list, solveT_list))) break;
                if (!dalloc((void**)&p->d_solveT, (size_t)solveT_size * sizeof(double))) { rc = SF_ERR_ALLOC; break; }
                p->bytes_device += (size_t)solveT_size * sizeof(double);
                p->n_solveT = (int64_t)solveT_list.size();
            }
            // sync words of the solve: [0] status, then the flags / counters, then two launch tickets per step
            const size_t sb = (size_t)(1 + p->n_solve_sync + sf_chol_plan::SOLVE_TICKETS * p->solve_steps.size()) * sizeof(int);
            if (!dalloc((void**)&p->d_solve_sync, sb)) { rc = SF_ERR_ALLOC; break; }
            p->bytes_device += sb;
            if (!dalloc((void**)&p->d_x, std::max<int64_t>(n, 1) * sizeof(double))) { rc = SF_ERR_ALLOC; break; }
            p->bytes_device += std::max<int64_t>(n, 1) * sizeof(double);
        }
        if (lu || p->partial) {
            if ((rc = up(&p->d_Xp, XP))) break;
        }
        if (!fill_tiles.empty()) {
            const size_t fb = fill_tiles.size() * sizeof(sf::FillTile);
            if (!dalloc(&p->d_fill, fb)) { rc = SF_ERR_ALLOC; break; }
            if (!dry && hipMemcpy(p->d_fill, fill_tiles.data(), fb, hipMemcpyHostToDevice) != hipSuccess) { rc = SF_ERR_HIP; break; }
            p->bytes_device += fb;
        }
        if (n_la_events > 0) {
            bool ok = dry || p->stream2 || new_stream(&p->stream2, hipStreamNonBlocking);
            p->la_events.assign((size_t)n_la_events, nullptr);
            for (hipEvent_t& e : p->la_events) ok = ok && (dry || hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess);
            if (!ok) { rc = SF_ERR_HIP; break; }
        }
        if (!p->segments.empty()) {
            int64_t mx = 1;
            for (const Segment& sg : p->segments) mx = std::max(mx, sg.packed);
            p->scratch_elems = mx;
            bool any_owner = false;
            for (const Segment& sg : p->segments) any_owner = any_owner || sg.owner_gi >= 0;
            const size_t nbuf = any_owner ? 3 : 2;          // owner-computes: the broadcast has a buffer of its own (the next block's sum may be in flight)
            if (!dalloc((void**)&p->d_scratch, nbuf * mx * sizeof(double))) { rc = SF_ERR_ALLOC; break; }
            p->bytes_device += nbuf * mx * sizeof(double);
            bool ok = dry || p->stream2 || new_stream(&p->stream2, hipStreamNonBlocking);
            for (int k = 0; k < 2 && !dry; ++k) {
                ok = ok && hipEventCreateWithFlags(&p->ev_contrib[k], hipEventDisableTiming) == hipSuccess;
                ok = ok && hipEventCreateWithFlags(&p->ev_reduced[k], hipEventDisableTiming) == hipSuccess;
                ok = ok && hipEventCreateWithFlags(&p->ev_unpacked[k], hipEventDisableTiming) == hipSuccess;
            }
            if (!ok) { rc = SF_ERR_HIP; break; }
        }
        if (nranks > 1) {
            // the word of the ranks' status agreement (sf_multi.hip, agree_status) exists from the start: the agreement itself must not
            // depend on an allocation that can fail on one rank alone
            if (!dalloc((void**)&p->d_status, sizeof(double))) { rc = SF_ERR_ALLOC; break; }
            p->bytes_device += sizeof(double);
        }
        if (ooc) {
            // one assembly mask per group, then the top's: d_loadmask + g * nsuper (g = ooc_ngroups: the top)
            std::vector<int8_t> mask((size_t)(ooc_ngroups + 1) * (size_t)std::max<sf_long>(nsuper, 1), 0);
            for (sf_long s = 0; s < nsuper; ++s) {
                const int unit = ooc_group[s] >= 0 ? ooc_group[s] : (ooc_top_mode >= 1 ? p->ooc_first[(size_t)s] : ooc_ngroups);
                mask[(size_t)unit * (size_t)nsuper + (size_t)s] = 1;
            }
            if ((rc = up(&p->d_loadmask, mask))) break;
            if (ooc_top_mode >= 1) {        // what a group's first launch zeroes besides its buffer: the places of the top panels that start with it
                p->ooc_zero.assign((size_t)ooc_ngroups, {});
                for (sf_long s = 0; s < nsuper; ++s)
                    if (ooc_group[s] < 0) {
                        auto& z = p->ooc_zero[(size_t)p->ooc_first[(size_t)s]];
                        const int64_t o = XP[s], len = (Super[s + 1] - Super[s]) * (Lsip[s + 1] - Lsip[s]);
                        if (!z.empty() && z.back().first + z.back().second == o) z.back().second += len;
                        else z.push_back({o, len});
                    }
            }
        } else if (p->partial) {
            std::vector<int8_t> mask(std::max<sf_long>(nsuper, 1), 0);
            // the matrix entries of a shared top panel enter the sum once: on the first rank of its group (load_top == 2),
            // or on the rank the caller names (load_top 0 / 1: the older interface, one group of all ranks)
            for (sf_long s = 0; s < nsuper; ++s)
                mask[s] = (p->phase[s] == 0 || (p->phase[s] == 1 && (load_top == 2 ? group_idx(gmask[s]) == 0 : load_top != 0))) ? 1 : 0;
            if ((rc = up(&p->d_loadmask, mask))) break;
        }
        if (lu) {
            // pivot records: pivpos | pivinv, n entries each, + the perturbation counter; identity until a factorization with
            // pivoting overwrites the blocks it interchanges
            const size_t pb = (size_t)(2 * std::max<int64_t>(n, 1) + 1) * sizeof(int32_t);
            if (!dalloc((void**)&p->d_piv, pb)) { rc = SF_ERR_ALLOC; break; }
            p->bytes_device += pb;
            if (const char* env = getenv("SF_LU_PIVOT_TOL")) {
                p->piv_tol = std::min(1.0, std::max(0.0, atof(env)));
                if (p->piv_tol > 0.0) p->piv_perturb = 1.4901161193847656e-08;      // sqrt(eps): pivoting comes with its fallback
            }
            if (const char* env = getenv("SF_LU_PERTURB")) p->piv_perturb = std::max(0.0, atof(env));
            p->piv_tol0 = p->piv_tol; p->piv_perturb0 = p->piv_perturb;
            if (!p->u_alias) {
                if ((rc = up(&p->d_Up, Up64))) break;
                if ((rc = up(&p->d_Ui, Ui32))) break;
                const size_t ub = std::max<int64_t>(p->unz, 1) * sizeof(double);
                if (!dalloc((void**)&p->d_Ux, ub)) { rc = SF_ERR_ALLOC; break; }
                p->bytes_device += ub;
            }
        }
        if (!p->partial) {
            // where every matrix entry goes (k_build_loadmap, once): the assembly of a whole plan is then one coalesced pass per
            // factorization.  Plans with assembly masks (several ranks, out of core) keep the searching kernel.
            const int64_t* xpm = lu ? p->d_Xp : p->d_Lsxp;
            const size_t mbL = (size_t)std::max<int64_t>(p->nnz, 1) * sizeof(int64_t);
            if (!dalloc((void**)&p->d_loadmapL, mbL)) { rc = SF_ERR_ALLOC; break; }
            p->bytes_device += mbL;
            if (!dry) sf::launch_build_loadmap(p->d_Lp, p->d_Li, (int32_t)n, p->d_Super, p->d_SuperMap, p->d_Lsip, p->d_Lsi, xpm, 0, lu ? 1 : 0, p->d_loadmapL, p->stream);
            if (lu) {
                const int64_t nzu = p->u_alias ? p->nnz : p->unz;
                const size_t mbU = (size_t)std::max<int64_t>(nzu, 1) * sizeof(int64_t);
                if (!dalloc((void**)&p->d_loadmapU, mbU)) { rc = SF_ERR_ALLOC; break; }
                p->bytes_device += mbU;
                if (!dry) sf::launch_build_loadmap(p->u_alias ? p->d_Lp : p->d_Up, p->u_alias ? p->d_Li : p->d_Ui, (int32_t)n, p->d_Super, p->d_SuperMap,
                                                   p->d_Lsip, p->d_Lsi, xpm, p->xC, 0, p->d_loadmapU, p->stream);
            }
            if (!dry && (hipStreamSynchronize(p->stream) != hipSuccess || hipGetLastError() != hipSuccess)) { rc = SF_ERR_HIP; break; }
        }
        // GEMM launches: 8 claim counters each (one per XCD) for the dynamic deal of their whole-tile rounds
        for (Launch& L : p->launches)
            if (L.kind == 2 || L.kind == 3 || L.kind == 4) { L.ticket = p->n_tickets; p->n_tickets += 8; }
        if (const char* env = sf_exp_env("SF_GEMM_DYNAMIC")) p->gemm_dynamic = atoi(env) != 0;
        const size_t xb = xb_factor, vb = std::max<int64_t>(p->nnz, 1) * sizeof(double);
        if (factor_alloc.joinable()) factor_alloc.join();
        if (factor_alloc_err != hipSuccess) { (void)hipGetLastError(); rc = SF_ERR_ALLOC; break; }
        p->d_Lsx = lent ? lent : factor_mem;
        factor_mem = nullptr;
        if (!dalloc((void**)&p->d_Lx, vb) ||
            !dalloc((void**)&p->d_info, (1 + p->n_tickets) * sizeof(int))) { rc = SF_ERR_ALLOC; break; }
        p->bytes_device += xb + vb + (1 + p->n_tickets) * sizeof(int);
    } while (0);
    if (rc) { sf_chol_plan_destroy(p); return rc; }
    if (trace_pc)
        fprintf(stderr, "[sparseframe-hip] plan_create: validate + levels %.1f ms, task tables %.1f ms, solve + download schedules + K prefixes %.1f ms, "
                        "uploads + relative maps + allocations %.1f ms\n", pc_t1 - pc_t0, pc_t2 - pc_t1, pc_t3 - pc_t2, pc_now() - pc_t3);
    *out = p;
    return SF_OK;
}

int sf_chol_plan_create(sf_chol_plan** out, int device, sf_long n, sf_long nsuper,
                        const sf_long* Super, const sf_long* SuperMap,
                        const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                        const sf_long* Lp, const sf_long* Li) {
    return plan_create(out, device, false, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, nullptr, nullptr);
}

int sf_chol_plan_create_sharded(sf_chol_plan** out, int device, sf_long n, sf_long nsuper,
                                const sf_long* Super, const sf_long* SuperMap,
                                const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                                const sf_long* Lp, const sf_long* Li, const int32_t* phase, int load_top) {
    return plan_create(out, device, false, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, nullptr, nullptr, phase, load_top);
}

// Out-of-core plans (plan_create, ooc_group): group[] from sf_ooc_partition; ngroups <= 1 gives the ordinary in-core plan.
int sf_chol_plan_create_ooc(sf_chol_plan** out, int device, sf_long n, sf_long nsuper,
                            const sf_long* Super, const sf_long* SuperMap,
                            const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                            const sf_long* Lp, const sf_long* Li, const int32_t* group, int ngroups, int top_mode) {
    if (ngroups > 1 && !group) return SF_ERR_ARG;
    return plan_create(out, device, false, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, nullptr, nullptr, nullptr, 1, 0, 1,
                       nullptr, nullptr, false, group, ngroups, top_mode);
}

// schedule-only out-of-core plan (no device; see sf_chol_plan_schedule_mapped): what the launch list and the storage would be
int sf_chol_plan_schedule_ooc(sf_chol_plan** out, sf_long n, sf_long nsuper,
                              const sf_long* Super, const sf_long* SuperMap,
                              const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                              const sf_long* Lp, const sf_long* Li, const int32_t* group, int ngroups, int top_mode) {
    if (ngroups > 1 && !group) return SF_ERR_ARG;
    return plan_create(out, 0, false, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, nullptr, nullptr, nullptr, 1, 0, 1,
                       nullptr, nullptr, true, group, ngroups, top_mode);
}

int sf_lu_plan_create_ooc(sf_chol_plan** out, int device, sf_long n, sf_long nsuper,
                          const sf_long* Super, const sf_long* SuperMap,
                          const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                          const sf_long* Lp, const sf_long* Li, const sf_long* Up, const sf_long* Ui, const int32_t* group, int ngroups, int top_mode) {
    if (ngroups > 1 && !group) return SF_ERR_ARG;
    return plan_create(out, device, true, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, Up, Ui, nullptr, 1, 0, 1,
                       nullptr, nullptr, false, group, ngroups, top_mode);
}

int sf_chol_plan_create_distributed(sf_chol_plan** out, int device, sf_long n, sf_long nsuper,
                                    const sf_long* Super, const sf_long* SuperMap,
                                    const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                                    const sf_long* Lp, const sf_long* Li, const int32_t* phase, int load_top,
                                    int rank, int nranks) {
    return plan_create(out, device, false, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, nullptr, nullptr, phase, load_top,
                       rank, nranks);
}

// Rank `rank`'s plan of an nranks-way factorization from the owner map of sf_subtree_partition* (owner[s] = rank of the
// subtree holding s, -1 = top): PROPORTIONAL MAPPING of the top -- a top supernode belongs to the ranks whose subtrees lie
// below it, only they store its panel, split its GEMMs and sum its block columns.
static int create_mapped(sf_chol_plan** out, int device, bool lu, sf_long n, sf_long nsuper,
                         const sf_long* Super, const sf_long* SuperMap, const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                         const sf_long* Lp, const sf_long* Li, const sf_long* Up, const sf_long* Ui,
                         const int32_t* owner, int rank, int nranks, bool dry = false) {
    if (!out || !owner || nranks < 1 || nranks > 32 || rank < 0 || rank >= nranks) return SF_ERR_ARG;
    if (nsuper < 0 || !Super || !Lsip || (nsuper > 0 && (!SuperMap || !Lsi))) return SF_ERR_ARG;
    std::vector<uint32_t> mask(std::max<sf_long>(nsuper, 1), 0);
    std::vector<int32_t> phase(std::max<sf_long>(nsuper, 1), 0);
    for (sf_long s = 0; s < nsuper; ++s) {
        if (owner[s] < -1 || owner[s] >= nranks) return SF_ERR_ARG;
        if (owner[s] >= 0) mask[s] |= 1u << owner[s];
        const sf_long nscol = Super[s + 1] - Super[s], nsrow = Lsip[s + 1] - Lsip[s];
        if (nscol < nsrow) {        // parents follow their children in the postorder
            const sf_long row = Lsi[Lsip[s] + nscol];
            if (row < 0 || row >= n) return SF_ERR_ARG;
            const sf_long par = SuperMap[row];
            if (par <= s || par >= nsuper) return SF_ERR_ARG;
            mask[par] |= mask[s];
        }
    }
    for (sf_long s = 0; s < nsuper; ++s) {
        if (owner[s] >= 0) { phase[s] = owner[s] == rank ? 0 : -1; mask[s] = 0; }
        else phase[s] = ((mask[s] >> rank) & 1u) ? 1 : -1;
    }
    // Weighted shares of the sets every rank takes part in (the root's GEMMs: the largest pool of divisible work).  The relaxed
    // amalgamation can make the tree lopsided -- at 161^3 over 4 ranks one pair of ranks shares a 14,641-column separator, the other
    // pair a 6,720-column one -- so the ranks arrive at the root with different loads.  Model (every rank evaluates the same
    // numbers): time of a rank before the root = its subtrees' flops at 42 TFLOP/s + for each smaller group it belongs to the
    // replicated part (chains: 45 us per 64 columns; in-block and near-part updates) + its equal share of that group's divisible
    // flops at 48 TFLOP/s (calibrated on tools/emulate_rank.py, 128^3 / 8 and 161^3 / 4); the root's divisible flops, at 58 TFLOP/s, are
    // then dealt out so that the ranks finish together (shares clamped to [0.2, 3] / nranks).  SF_WEIGHTED_SHARES=0: equal shares.
    std::vector<double> cum;
    const char* wenv = sf_exp_env("SF_WEIGHTED_SHARES");
    if (nranks > 1 && !(wenv && atoi(wenv) == 0)) {
        const uint32_t all = nranks >= 32 ? 0xffffffffu : ((1u << nranks) - 1u);
        std::vector<double> before(nranks, 0.0);
        double root_ms = 0.0;
        for (sf_long s = 0; s < nsuper; ++s) {
            const double k = (double)(Super[s + 1] - Super[s]), r = (double)(Lsip[s + 1] - Lsip[s]), m = r - k;
            double f = k * k * k / 3.0 + m * k * k;
            {
                const sf_long* rows = Lsi + Lsip[s];
                sf_long i = (sf_long)k;
                const sf_long nsrow = (sf_long)r;
                while (i < nsrow) {
                    const sf_long o = SuperMap[rows[i]];
                    sf_long e = i;
                    while (e < nsrow && SuperMap[rows[e]] == o) ++e;
                    const double dn = (double)(e - i), dm = (double)(nsrow - e);
                    f += dn * (dn + 1) * k + 2.0 * dm * dn * k;
                    i = e;
                }
            }
            if (lu) f *= 2.0;
            if (owner[s] >= 0) { before[owner[s]] += f / 42e9; continue; }
            const double rep = std::min(f, (lu ? 2.0 : 1.0) * 1472.0 * (r * k - 0.5 * k * k));
            const double chain_ms = std::ceil(k / 64.0) * 0.045;
            const int g = __builtin_popcount(mask[s]);
            if (mask[s] == all) { root_ms += (f - rep) / 58e9; continue; }
            for (int q = 0; q < nranks; ++q)
                if ((mask[s] >> q) & 1u) before[q] += chain_ms + (rep + (f - rep) / std::max(g, 1)) / 48e9;
        }
        if (root_ms > 0.0) {
            const double smin = 0.2 / nranks, smax = 3.0 / nranks;
            double lo = *std::min_element(before.begin(), before.end()), hi = *std::max_element(before.begin(), before.end()) + root_ms;
            std::vector<double> sh(nranks, 1.0 / nranks);
            for (int it = 0; it < 60; ++it) {            // bisection on the common finishing time
                const double T = 0.5 * (lo + hi);
                double sum = 0.0;
                for (int q = 0; q < nranks; ++q) sum += std::min(smax, std::max(smin, (T - before[q]) / root_ms));
                if (sum > 1.0) hi = T; else lo = T;
            }
            double sum = 0.0;
            for (int q = 0; q < nranks; ++q) { sh[q] = std::min(smax, std::max(smin, (0.5 * (lo + hi) - before[q]) / root_ms)); sum += sh[q]; }
            cum.assign(nranks + 1, 0.0);
            for (int q = 0; q < nranks; ++q) cum[q + 1] = cum[q] + sh[q] / sum;
            cum[nranks] = 1.0;
        }
    }
    return plan_create(out, device, lu, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, Up, Ui, phase.data(), 2, rank, nranks, mask.data(),
                       cum.empty() ? nullptr : cum.data(), dry);
}

int sf_chol_plan_create_mapped(sf_chol_plan** out, int device, sf_long n, sf_long nsuper,
                               const sf_long* Super, const sf_long* SuperMap,
                               const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                               const sf_long* Lp, const sf_long* Li, const int32_t* owner, int rank, int nranks) {
    return create_mapped(out, device, false, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, nullptr, nullptr, owner, rank, nranks);
}

int sf_lu_plan_create_mapped(sf_lu_plan** out, int device, sf_long n, sf_long nsuper,
                             const sf_long* Super, const sf_long* SuperMap,
                             const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                             const sf_long* Lp, const sf_long* Li, const sf_long* Up, const sf_long* Ui,
                             const int32_t* owner, int rank, int nranks) {
    return create_mapped(out, device, true, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, Up, Ui, owner, rank, nranks);
}

// Schedule-only twins of the two calls above: no device, nothing allocated (plan_create's dry mode)
int sf_chol_plan_schedule_mapped(sf_chol_plan** out, sf_long n, sf_long nsuper,
                                 const sf_long* Super, const sf_long* SuperMap,
                                 const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                                 const sf_long* Lp, const sf_long* Li, const int32_t* owner, int rank, int nranks) {
    return create_mapped(out, -1, false, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, nullptr, nullptr, owner, rank, nranks, true);
}

int sf_lu_plan_schedule_mapped(sf_lu_plan** out, sf_long n, sf_long nsuper,
                               const sf_long* Super, const sf_long* SuperMap,
                               const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                               const sf_long* Lp, const sf_long* Li, const sf_long* Up, const sf_long* Ui,
                               const int32_t* owner, int rank, int nranks) {
    return create_mapped(out, -1, true, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, Up, Ui, owner, rank, nranks, true);
}

// the group of ranks that sums segment k's block columns (bit r = rank r); 0 for a plan that is not distributed
uint32_t sf_chol_plan_segment_group(const sf_chol_plan* p, sf_long k) {
    return (p && k >= 0 && k < (sf_long)p->segments.size()) ? p->segments[k].mask : 0u;
}

int sf_lu_plan_create(sf_lu_plan** out, int device, sf_long n, sf_long nsuper,
                      const sf_long* Super, const sf_long* SuperMap,
                      const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                      const sf_long* Lp, const sf_long* Li, const sf_long* Up, const sf_long* Ui) {
    return plan_create(out, device, true, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, Up, Ui);
}

int sf_chol_plan_set_values(sf_chol_plan* p, const sf_float* Lx) {
    if (!p || p->lu || (!Lx && p->nnz > 0)) return SF_ERR_ARG;
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    HIP_TRY(hipSetDevice(p->device));
    if (p->nnz > 0) HIP_TRY(hipMemcpyAsync(p->d_Lx, Lx, p->nnz * sizeof(double), hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->values_set = true;
    return SF_OK;
}

int sf_lu_plan_create_distributed(sf_lu_plan** out, int device, sf_long n, sf_long nsuper,
                                  const sf_long* Super, const sf_long* SuperMap,
                                  const sf_long* Lsip, const sf_long* Lsi, const sf_long* Lsxp,
                                  const sf_long* Lp, const sf_long* Li, const sf_long* Up, const sf_long* Ui,
                                  const int32_t* phase, int load_top, int rank, int nranks) {
    return plan_create(out, device, true, n, nsuper, Super, SuperMap, Lsip, Lsi, Lsxp, Lp, Li, Up, Ui, phase, load_top, rank, nranks);
}

int sf_lu_plan_set_values(sf_lu_plan* p, const sf_float* Lx, const sf_float* Ux) {
    if (!p || !p->lu || (!Lx && p->nnz > 0) || (!p->u_alias && !Ux && p->unz > 0)) return SF_ERR_ARG;
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    HIP_TRY(hipSetDevice(p->device));
    if (p->nnz > 0) HIP_TRY(hipMemcpyAsync(p->d_Lx, Lx, p->nnz * sizeof(double), hipMemcpyHostToDevice, p->stream));
    if (!p->u_alias && p->unz > 0) HIP_TRY(hipMemcpyAsync(p->d_Ux, Ux, p->unz * sizeof(double), hipMemcpyHostToDevice, p->stream));
    {   // max |a_ij|: scale of the pivot perturbation
        double m = 0;
        for (int64_t k = 0; k < p->nnz; ++k) m = std::max(m, std::fabs(Lx[k]));
        if (!p->u_alias) for (int64_t k = 0; k < p->unz; ++k) m = std::max(m, std::fabs(Ux[k]));
        p->amax = m;
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->values_set = true;
    return SF_OK;
}

int sf_chol_plan_sync(sf_chol_plan* p) {
    if (!p) return SF_ERR_ARG;
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    int info = 0;
    HIP_TRY(hipMemcpy(&info, p->d_info, sizeof(int), hipMemcpyDeviceToHost));
    float ms = 0;
    // (a run driven segment by segment never records ev1: the call then fails, and the error must not stay behind as the thread's
    // "last error" -- a caller that checks hipGetLastError after its own launches, as PyTorch does, would trip over it)
    if (elapsed_ms(&ms, p->ev0, p->ev1)) p->last_ms = ms;
    if (p->lu) HIP_TRY(hipMemcpy(&p->last_perturbed, p->d_piv + 2 * std::max<int64_t>(p->n, 1), sizeof(int), hipMemcpyDeviceToHost));
    // 1: non-positive / zero pivot; 2: a fused step's flag wait timed out (internal error, never seen)
    p->last_status = (info & 2) ? SF_ERR_HIP : (info ? SF_ERR_NOT_POSDEF : SF_OK);
    return p->last_status;
}

// overlapped download: record the events of everything that is final once `done` launches have been enqueued and
// tell the copy workers about them
static hipError_t dl_publish(sf_chol_plan* p, size_t done) {
    size_t k = p->dl_next_ev;
    while (k < p->dl_ev_ready.size() && p->dl_ev_ready[k] <= done) {
        if (p->dl_lu_direct && p->d_fill && k + 1 < p->fill_first.size()) {
            sf::launch_lu_fill_u11((const sf::FillTile*)p->d_fill + p->fill_first[k], p->fill_first[k + 1] - p->fill_first[k],
                                   p->d_Lsx, p->d_Lsx + p->xC, p->stream);
            const hipError_t ef = hipGetLastError();
            if (ef != hipSuccess) return ef;
        }
        const hipError_t e = hipEventRecord(p->dl_events[k], p->stream);
        if (e != hipSuccess) return e;
        ++k;
    }
    if (k != p->dl_next_ev) {
        p->dl_next_ev = k;
        {
            std::lock_guard<std::mutex> g(p->dl_mu);
            p->dl_published = k;
        }
        p->dl_cv.notify_all();
    }
    return hipSuccess;
}

// This rank's window [lo, hi) of a launch's `total` divisible items (GEMM launches: stream-K units; k_update_small: tiles); all of
// them unless the launch is split over a group.  The boundaries are the same doubles on the two ranks they separate, so the windows
// of a group's members tile [0, total) exactly (checked for every launch of 256^3 / 8 by tests/test_config4_schedules.py).
static inline void launch_window(const Launch& L, int64_t total, int64_t* lo, int64_t* hi) {
    *lo = 0; *hi = total;
    if (!L.split) return;
    *lo = (int64_t)((double)total * L.share_lo);
    *hi = L.share_hi >= 1.0 ? total : (int64_t)((double)total * L.share_hi);
}

// launches [l0, l1); first: start of a factorization (timer, memset, assembly); last: its end (timer, status)
static int run_launches(sf_chol_plan* p, size_t l0, size_t l1, bool first, bool last, int sync) {
    if (!p->values_set) return SF_ERR_ARG;
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    if (p->ooc_groups > 1 && !p->dl_active) return SF_ERR_ARG;      // an out-of-core plan only exists together with its copy-back
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t st = p->stream;
    struct EventList {      // profiling events; destroyed on every exit path
        std::vector<hipEvent_t> v;
        ~EventList() { for (hipEvent_t e : v) (void)hipEventDestroy(e); }
    } evlist;
    std::vector<hipEvent_t>& evs = evlist.v;
    auto mark = [&]() {
        if (!p->profiling) return;
        hipEvent_t e;
        if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, st); evs.push_back(e); }
    };
    if (first) {
        if (!p->capturing) HIP_TRY(hipEventRecord(p->ev0, st));
        p->packed_pending = -1;
        p->epoch = (p->epoch == 0x7fffffff) ? 1 : p->epoch + 1;     // flag value of this factorization's fused steps (never 0)
        if (p->capturing) {
            // a captured factorization is replayed with the SAME kernel arguments: its flag value is fixed (never seen in the array
            // again: epochs only grow) and the flags are cleared by a node of the graph itself
            p->graph_epoch = p->epoch;
            HIP_TRY(hipMemsetAsync(p->d_flags, 0, (size_t)p->n_flags * sizeof(int), st));
        }
    }
    sf::PivotCtl pc{0.0, 0.0, nullptr, nullptr, nullptr};
    if (p->lu) {
        const bool piv = p->piv_tol > 0.0;
        pc.tol = p->piv_tol;
        pc.eps = p->piv_perturb * p->amax;
        pc.pivpos = piv ? p->d_piv : nullptr;
        pc.pivinv = piv ? p->d_piv + std::max<int64_t>(p->n, 1) : nullptr;
        pc.nperturb = (int*)(p->d_piv + 2 * std::max<int64_t>(p->n, 1));
        if (first) HIP_TRY(hipMemsetAsync(pc.nperturb, 0, sizeof(int), st));
    }
    // the matrix entries of the panels `mask` selects (nullptr: all) into the zeroed panels
    auto assemble = [&](const int8_t* mask, hipStream_t s_) {
        if (!mask && p->d_loadmapL) {           // a whole plan: the entries' places are known (plan_create)
            sf::launch_load_mapped(p->d_Lx, p->d_loadmapL, p->nnz, p->d_Lsx, s_);
            if (p->lu) sf::launch_load_mapped(p->u_alias ? p->d_Lx : p->d_Ux, p->d_loadmapU, p->u_alias ? p->nnz : p->unz, p->d_Lsx, s_);
            return;
        }
        const int64_t* xp = (p->lu || p->partial) ? p->d_Xp : p->d_Lsxp;
        if (!p->lu) {
            sf::launch_load_panels(p->d_Lp, p->d_Li, p->d_Lx, (int32_t)p->n, p->d_Super, p->d_SuperMap, p->d_Lsip,
                                   p->d_Lsi, xp, p->d_Lsx, 0, mask, s_);
        } else {
            // L panel: strictly lower entries of the columns of L; U^T panel: row j of U (diagonal included) goes to
            // column j of PU at the positions of its column indices (reference loadA, L:2490-2533)
            sf::launch_load_panels(p->d_Lp, p->d_Li, p->d_Lx, (int32_t)p->n, p->d_Super, p->d_SuperMap, p->d_Lsip,
                                   p->d_Lsi, xp, p->d_Lsx, 1, mask, s_);
            if (p->u_alias)
                sf::launch_load_panels(p->d_Lp, p->d_Li, p->d_Lx, (int32_t)p->n, p->d_Super, p->d_SuperMap, p->d_Lsip,
                                       p->d_Lsi, xp, p->d_Lsx + p->xC, 0, mask, s_);
            else
                sf::launch_load_panels(p->d_Up, p->d_Ui, p->d_Ux, (int32_t)p->n, p->d_Super, p->d_SuperMap, p->d_Lsip,
                                       p->d_Lsi, xp, p->d_Lsx + p->xC, 0, mask, s_);
        }
    };
    if (first) {
        HIP_TRY(hipMemsetAsync(p->d_info, 0, (1 + p->n_tickets) * sizeof(int), st));
        if (p->xC > 0) HIP_TRY(hipMemsetAsync(p->d_Lsx, 0, (p->lu ? 2 : 1) * p->xC * sizeof(double), st));
        // (out of core: the top panels only; every group assembles its own buffer when its turn comes, launch kind 7)
        assemble(p->ooc_groups > 1 ? p->d_loadmask + (size_t)p->ooc_groups * (size_t)p->nsuper : p->d_loadmask, st);
    }
    mark();
    std::vector<int> kinds;
    hipStream_t const st_main = st;
    for (size_t li = l0; li < l1; ++li) {
        const Launch& L = p->launches[li];
        // one-GPU look-ahead: a lane-1 launch goes to the second stream; events order the lanes.  Under profiling everything stays on
        // the main stream (per-launch events need one stream): the profiled breakdown is that of the serial schedule.
        st = (L.lane == 1 && !p->profiling && p->stream2) ? p->stream2 : st_main;
        if (L.wait_ev >= 0 && !p->profiling) HIP_TRY(hipStreamWaitEvent(st, p->la_events[(size_t)L.wait_ev], 0));
        switch (L.kind) {
            case 0:
                if (p->lu) sf::launch_getrf(p->d_potrf + L.first, L.count, p->d_Lsx, p->xC, p->d_info, pc, st);
                else sf::launch_potrf(p->d_potrf + L.first, L.count, p->d_Lsx, p->d_info, st);
                break;
            case 1: sf::launch_trsm(p->d_trsm + L.first, L.count, p->d_Lsx, pc.pivinv, st); break;
            case 7: {       // out of core: group L.first takes over buffer L.first & 1
                const int g = (int)L.first;
                // ... once every piece of group g - 2 has LEFT the device (its DMA into the pinned ring is complete; the copy workers
                // count) -- with top mode 2 possibly of a later group, whose top panels' places are taken over here.  Everything those
                // pieces wait for has been enqueued and published by now.
                const int upto = p->ooc_wait.empty() ? g - 2 : p->ooc_wait[(size_t)g];
                for (int w = std::max(0, g - 2); w <= upto && w < g; ++w) {
                    std::unique_lock<std::mutex> lk(p->dl_mu);
                    p->dl_cv.wait(lk, [&] { return p->dl_abort || p->dl_group_left[(size_t)w].load() <= 0; });
                    if (p->dl_abort) return SF_ERR_HIP;
                }
                if (g >= 2 && p->ooc_buf > 0) {      // (the first factorization step zeroed everything: groups 0 and 1 find clean buffers)
                    const size_t off = (size_t)(g & 1) * (size_t)p->ooc_buf, nb = (size_t)p->ooc_buf * sizeof(double);
                    HIP_TRY(hipMemsetAsync(p->d_Lsx + off, 0, nb, st));
                    if (p->lu) HIP_TRY(hipMemsetAsync(p->d_Lsx + p->xC + off, 0, nb, st));
                }
                if (p->ooc_top_mode >= 1)       // the top panels that become active with this group (their places may have held others)
                    for (const auto& z : p->ooc_zero[(size_t)g]) {
                        HIP_TRY(hipMemsetAsync(p->d_Lsx + z.first, 0, (size_t)z.second * sizeof(double), st));
                        if (p->lu) HIP_TRY(hipMemsetAsync(p->d_Lsx + p->xC + z.first, 0, (size_t)z.second * sizeof(double), st));
                    }
                assemble(p->d_loadmask + (size_t)g * (size_t)p->nsuper, st);
                break;
            }
            case 6: {       // k_update_small; a split launch (distributed top): this rank's share of the tiles (the update is a sum)
                int64_t lo, hi;
                launch_window(L, L.count, &lo, &hi);
                sf::launch_update_small(p->d_probs, p->d_stasks + L.first + lo, (int)(hi - lo), p->d_Lsx, p->d_relmap, st);
                break;
            }
            case 5:
                sf::launch_step(p->d_steps + L.first, L.count, p->lu ? 1 : 0, p->d_Lsx, p->d_flags, p->epoch, p->d_info, p->d_tinv,
                                p->d_info + 1 + L.ticket, pc, st);
                break;
            case 2:
            case 3:
            case 4: {
                int64_t w0, w1;
                launch_window(L, L.units, &w0, &w1);
                const uint32_t u0 = (uint32_t)w0, u1 = (uint32_t)w1;
                sf::launch_gemm(p->d_probs, p->d_gtasks + L.first, p->d_ktprefix + L.prefix_first, L.count, u0, u1,
                                L.kind == 3 ? 1 : 0, p->d_Lsx, p->d_relmap, p->gemm_dynamic ? p->d_info + 1 + L.ticket : nullptr, st,
                                (L.whole_tiles && !L.split) ? 1 : 0, (L.lane == 1 && !p->profiling) ? p->la_grid : 0);
                break;
            }
        }
        if (L.rec_ev >= 0 && !p->profiling) HIP_TRY(hipEventRecord(p->la_events[(size_t)L.rec_ev], st));
        st = st_main;
        if (p->profiling) { kinds.push_back(L.kind); mark(); }
        if (p->dl_active) HIP_TRY(dl_publish(p, li + 1));
    }
    if (last && !p->capturing) HIP_TRY(hipEventRecord(p->ev1, st));
    HIP_TRY(hipGetLastError());
    if (p->profiling) {
        HIP_TRY(hipStreamSynchronize(st));
        if (first) {
            p->last_load_ms = p->last_panel_ms = p->last_update_ms = 0;
            for (double& v : p->last_kind_ms) v = 0;
        }
        float ms = 0;
        if (first && !evs.empty() && elapsed_ms(&ms, p->ev0, evs[0])) p->last_load_ms = ms;
        FILE* dump = nullptr;
        if (const char* path = getenv("SF_PROFILE_DUMP")) dump = fopen(path, first ? "w" : "a");
        if (dump && first) fprintf(dump, "launch,kind,tasks,units,flops,ms\n");
        for (size_t k = 0; k + 1 < evs.size(); ++k) {
            if (!elapsed_ms(&ms, evs[k], evs[k + 1])) continue;
            if (kinds[k] == 3) p->last_update_ms += ms; else if (kinds[k] != 6) p->last_panel_ms += ms;
            p->last_kind_ms[kinds[k]] += ms;
            if (dump) {
                const Launch& L = p->launches[l0 + k];
                fprintf(dump, "%zu,%d,%d,%u,%.6e,%.4f\n", l0 + k, L.kind, L.count, L.units, L.flops, ms);
            }
        }
        if (dump) fclose(dump);
    }
    if (sync) {
        if (!last) { HIP_TRY(hipStreamSynchronize(st)); return SF_OK; }
        return sf_chol_plan_sync(p);
    }
    return SF_OK;
}

// phase 0: assemble + owned subtrees; phase 1: top supernodes (replicated); -1: both (single-GPU path)
int sf_chol_plan_factorize_phase(sf_chol_plan* p, int which, int sync) {
    if (!p || which < -1 || which > 1) return SF_ERR_ARG;
    if (p->nranks > 1 && which != 0) return SF_ERR_ARG;      // distributed top: phase 1 runs segment by segment
    const size_t l0 = (which == 1) ? p->launch_split : 0;
    const size_t l1 = (which == 0) ? p->launch_split : p->launches.size();
    // a distributed plan without top supernodes (a forest of independent trees) is finished after phase 0
    const bool last = (which != 0) || (p->nranks > 1 && p->segments.empty());
    return run_launches(p, l0, l1, which != 1, last, sync);
}

sf_long sf_chol_plan_num_segments(const sf_chol_plan* p) { return p ? (sf_long)p->segments.size() : 0; }

// ---- schedule inspection (real and schedule-only plans; tests/test_config4_schedules.py, tools/) ----
sf_long sf_chol_plan_num_launches(const sf_chol_plan* p) { return p ? (sf_long)p->launches.size() : 0; }

int sf_chol_plan_launch_info(const sf_chol_plan* p, sf_long k, sf_long* out) {
    if (!p || !out || k < 0 || k >= (sf_long)p->launches.size()) return SF_ERR_ARG;
    const Launch& L = p->launches[(size_t)k];
    const bool gemm = L.kind >= 2 && L.kind <= 4;
    const int64_t total = gemm ? (int64_t)L.units : (int64_t)L.count;
    int64_t lo, hi;
    launch_window(L, total, &lo, &hi);
    sf_long seg = -1;
    if ((size_t)k >= p->launch_split && !p->segments.empty()) {
        size_t a = 0, b = p->segments.size();       // last segment with l0 <= k
        while (b - a > 1) { const size_t m = (a + b) / 2; if (p->segments[m].l0 <= (size_t)k) a = m; else b = m; }
        if (p->segments[a].l0 <= (size_t)k && (size_t)k < p->segments[a].l1) seg = (sf_long)a;
    }
    out[0] = L.kind; out[1] = L.count; out[2] = total; out[3] = L.split ? 1 : 0; out[4] = lo; out[5] = hi;
    out[6] = (L.whole_tiles && !L.split) ? 1 : 0; out[7] = seg; out[8] = L.share_idx; out[9] = L.share_cnt;
    return SF_OK;
}

int sf_chol_plan_segment_info(const sf_chol_plan* p, sf_long k, sf_long* out) {
    if (!p || !out || k < 0 || k >= (sf_long)p->segments.size()) return SF_ERR_ARG;
    const Segment& sg = p->segments[(size_t)k];
    out[0] = (sf_long)sg.mask; out[1] = (sf_long)sg.l0; out[2] = (sf_long)sg.l1; out[3] = sg.packed; out[4] = sg.early ? 1 : 0;
    out[5] = (sf_long)sg.off.size();
    return SF_OK;
}

// owner-computes prototype (SF_TOP_OWNER=1): out[0] = group index of the rank that runs the block's near GEMM and chain (-1: everybody,
// the default), out[1] = number of launches of that part (the segment's first out[1] launches)
int sf_chol_plan_segment_owner(const sf_chol_plan* p, sf_long k, sf_long* out) {
    if (!p || !out || k < 0 || k >= (sf_long)p->segments.size()) return SF_ERR_ARG;
    const Segment& sg = p->segments[(size_t)k];
    out[0] = sg.owner_gi;
    out[1] = sg.owner_gi >= 0 ? (sf_long)(sg.lc - sg.l0) : 0;
    return SF_OK;
}

int sf_chol_plan_panel_offsets(const sf_chol_plan* p, sf_long* xp) {
    if (!p || (!xp && p->nsuper > 0)) return SF_ERR_ARG;
    for (int64_t s = 0; s < p->nsuper; ++s) xp[s] = p->h_XP[(size_t)s];
    return SF_OK;
}

sf_long sf_chol_plan_num_solve_reduces(const sf_chol_plan* p) { return p ? (sf_long)p->solve_reduces.size() : 0; }

int sf_chol_plan_solve_reduce_info(const sf_chol_plan* p, sf_long k, sf_long* out) {
    if (!p || !out || k < 0 || k >= (sf_long)p->solve_reduces.size()) return SF_ERR_ARG;
    out[0] = (sf_long)p->solve_reduces[(size_t)k].mask; out[1] = p->solve_reduces[(size_t)k].off; out[2] = p->solve_reduces[(size_t)k].cnt;
    return SF_OK;
}

int sf_chol_plan_segment_regions(const sf_chol_plan* p, sf_long k, sf_long capacity, sf_long* nregions, sf_long* offsets, sf_long* counts) {
    if (!p || k < 0 || k >= (sf_long)p->segments.size() || !nregions) return SF_ERR_ARG;
    const Segment& sg = p->segments[k];
    *nregions = (sf_long)sg.off.size();
    if (offsets && counts) {
        if (capacity < (sf_long)sg.off.size()) return SF_ERR_ARG;
        for (size_t i = 0; i < sg.off.size(); ++i) { offsets[i] = sg.off[i]; counts[i] = sg.cnt[i]; }
    }
    return SF_OK;
}

// gather the possibly non-zero part of segment k's regions into one contiguous buffer (strided device copies on
// the plan's stream): ONE all-reduce per segment, and the structurally zero rows above each block's diagonal
// (half of a square root panel) stay off the wire
static int seg_copy(sf_chol_plan* p, const Segment& sg, double* buf, bool pack, hipStream_t st) {
    int64_t pos = 0;
    for (size_t i = 0; i < sg.src.size(); ++i) {
        if (pack)
            HIP_TRY(hipMemcpy2DAsync(buf + pos, sg.rows[i] * sizeof(double), p->d_Lsx + sg.src[i], sg.ld[i] * sizeof(double),
                                     sg.rows[i] * sizeof(double), sg.cols[i], hipMemcpyDeviceToDevice, st));
        else
            HIP_TRY(hipMemcpy2DAsync(p->d_Lsx + sg.src[i], sg.ld[i] * sizeof(double), buf + pos, sg.rows[i] * sizeof(double),
                                     sg.rows[i] * sizeof(double), sg.cols[i], hipMemcpyDeviceToDevice, st));
        pos += sg.rows[i] * sg.cols[i];
    }
    return SF_OK;
}

int sf_chol_plan_segment_pack(sf_chol_plan* p, sf_long k, void** dptr, sf_long* count) {
    if (!p || k < 0 || k >= (sf_long)p->segments.size() || !dptr || !count) return SF_ERR_ARG;
    if (p->packed_pending >= 0) return SF_ERR_ARG;           // the previous packed segment has not been run
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    HIP_TRY(hipSetDevice(p->device));
    const Segment& sg = p->segments[k];
    double* buf = p->d_scratch + (k & 1) * p->scratch_elems;
    int rc = seg_copy(p, sg, buf, true, p->stream);
    if (rc) return rc;
    p->packed_pending = k;
    *dptr = (void*)buf;
    *count = sg.packed;
    return SF_OK;
}

int sf_chol_plan_factorize_segment(sf_chol_plan* p, sf_long k, int sync) {
    if (!p || k < 0 || k >= (sf_long)p->segments.size()) return SF_ERR_ARG;
    const Segment& sg = p->segments[k];
    if (p->packed_pending >= 0) {
        if (p->packed_pending != k) return SF_ERR_ARG;
        if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
        HIP_TRY(hipSetDevice(p->device));
        int rc = seg_copy(p, sg, p->d_scratch + (k & 1) * p->scratch_elems, false, p->stream);
        if (rc) return rc;
        p->packed_pending = -1;
    }
    const bool last = k + 1 == (sf_long)p->segments.size();
    if (sg.owner_gi >= 0) {
        // owner-computes prototype driven from outside (tools/emulate_rank.py: no collectives, timing only): the owner's part, then the rest
        const bool mine = sg.owner_gi == __builtin_popcount(sg.mask & ((1u << p->rank) - 1u));
        if (mine && sg.lc > sg.l0) { const int rc = run_launches(p, sg.l0, sg.lc, false, false, 0); if (rc) return rc; }
        return run_launches(p, sg.lc, sg.l1, false, last, sync);
    }
    return run_launches(p, sg.l0, sg.l1, false, last, sync);
}

}  // extern "C"

// Gathers a factor that is distributed over the plans of several ranks (sharded / mapped plans of ONE pattern, factorized and
// synchronised) into a whole plan of that pattern: every panel from the first part that stores it (a shared top panel is complete
// on every rank of its group once its chain has run), device to device over xGMI (hipMemcpyPeerAsync), LU: both panels and the
// pivot records.  After it `dst` solves as if it had factorized itself.  Used by the struct path's solve after a multi-handler
// factorization (sf_handlers.hip).
int sf_plan_import_from(sf_chol_plan* dst, sf_chol_plan* const* parts, int nparts) {
    if (!dst || dst->partial || !parts || nparts < 1) return SF_ERR_ARG;
    for (int r = 0; r < nparts; ++r) {
        const sf_chol_plan* P = parts[r];
        if (!P || P->n != dst->n || P->nsuper != dst->nsuper || P->lu != dst->lu || P->ooc_groups > 1) return SF_ERR_ARG;
        HIP_TRY(hipSetDevice(P->device));
        HIP_TRY(hipStreamSynchronize(P->stream));
    }
    HIP_TRY(hipSetDevice(dst->device));
    sf_long s = 0;
    while (s < dst->nsuper) {
        int src = -1;
        for (int r = 0; r < nparts && src < 0; ++r)
            if (parts[r]->h_XP[s] >= 0) src = r;
        if (src < 0) return SF_ERR_ARG;                 // a panel nobody stores
        const sf_chol_plan* P = parts[src];
        sf_long e = s;
        int64_t len = 0;
        while (e < dst->nsuper && P->h_XP[e] >= 0 && P->h_XP[e] == P->h_XP[s] + len && dst->h_XP[e] == dst->h_XP[s] + len) {
            bool earlier = false;                       // keep "the first part that stores it" for every panel of the run
            for (int r = 0; r < src; ++r) earlier = earlier || parts[r]->h_XP[e] >= 0;
            if (earlier) break;
            len += (dst->h_Super[e + 1] - dst->h_Super[e]) * (dst->h_Lsip[e + 1] - dst->h_Lsip[e]);
            ++e;
        }
        if (e == s) return SF_ERR_ARG;
        HIP_TRY(hipMemcpyPeerAsync(dst->d_Lsx + dst->h_XP[s], dst->device, P->d_Lsx + P->h_XP[s], P->device, len * sizeof(double), dst->stream));
        if (dst->lu) {
            HIP_TRY(hipMemcpyPeerAsync(dst->d_Lsx + dst->xC + dst->h_XP[s], dst->device, P->d_Lsx + P->xC + P->h_XP[s], P->device,
                                       len * sizeof(double), dst->stream));
            if (dst->d_piv && P->d_piv)
                HIP_TRY(hipMemcpyPeerAsync(dst->d_piv + dst->h_Super[s], dst->device, P->d_piv + dst->h_Super[s], P->device,
                                           (size_t)(dst->h_Super[e] - dst->h_Super[s]) * sizeof(int32_t), dst->stream));
        }
        s = e;
    }
    dst->piv_tol = parts[0]->piv_tol;
    dst->piv_perturb = parts[0]->piv_perturb;
    dst->hash_epoch = -1;           // the factor changed without a factorization of dst's own: cached fingerprints are stale
    HIP_TRY(hipStreamSynchronize(dst->stream));
    return SF_OK;
}

// The pieces of the pipelined driver (sf_multi.hip).  Segment k's sum uses half k & 1 of the scratch buffer and the plan's second
// stream:  begin (everything the sum needs has been enqueued on the main stream) -> pack on the second stream, returns the buffer for
// the collective, which the caller issues on sf_plan_stream2 -> reduced (marks the collective's end on the second stream) ->
// finish: the main stream waits for it, scatters the sums back and runs the segment's launches.
int sf_seg_begin(sf_chol_plan* p, sf_long k, void** dptr, sf_long* count) {
    if (!p || k < 0 || k >= (sf_long)p->segments.size() || !dptr || !count || !p->stream2) return SF_ERR_ARG;
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    HIP_TRY(hipSetDevice(p->device));
    const int h = (int)(k & 1);
    HIP_TRY(hipEventRecord(p->ev_contrib[h], p->stream));
    HIP_TRY(hipStreamWaitEvent(p->stream2, p->ev_contrib[h], 0));
    // the half's previous user (segment k - 2) has been scattered back on the main stream
    if (p->unpacked_recorded[h]) HIP_TRY(hipStreamWaitEvent(p->stream2, p->ev_unpacked[h], 0));
    double* buf = p->d_scratch + h * p->scratch_elems;
    int rc = seg_copy(p, p->segments[k], buf, true, p->stream2);
    if (rc) return rc;
    *dptr = (void*)buf;
    *count = p->segments[k].packed;
    return SF_OK;
}
void* sf_plan_stream2(sf_chol_plan* p) { return p ? (void*)p->stream2 : nullptr; }
int sf_seg_reduced(sf_chol_plan* p, sf_long k) {
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventRecord(p->ev_reduced[k & 1], p->stream2));
    return SF_OK;
}
int sf_seg_finish(sf_chol_plan* p, sf_long k) {
    if (!p || k < 0 || k >= (sf_long)p->segments.size()) return SF_ERR_ARG;
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    HIP_TRY(hipSetDevice(p->device));
    const int h = (int)(k & 1);
    const Segment& sg = p->segments[k];
    HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_reduced[h], 0));
    int rc = seg_copy(p, sg, p->d_scratch + h * p->scratch_elems, false, p->stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(p->ev_unpacked[h], p->stream));
    p->unpacked_recorded[h] = true;
    if (sg.owner_gi >= 0) {
        // the owner's part only (near GEMM + chain); sf_seg_bcast_begin / _finish follow
        const bool mine = sg.owner_gi == __builtin_popcount(sg.mask & ((1u << p->rank) - 1u));
        return (mine && sg.lc > sg.l0) ? run_launches(p, sg.l0, sg.lc, false, false, 0) : SF_OK;
    }
    return run_launches(p, sg.l0, sg.l1, false, k + 1 == (sf_long)p->segments.size(), 0);
}
int sf_seg_is_owner_segment(const sf_chol_plan* p, sf_long k) {
    return (p && k >= 0 && k < (sf_long)p->segments.size() && p->segments[(size_t)k].owner_gi >= 0) ? 1 : 0;
}
int sf_seg_bcast_begin(sf_chol_plan* p, sf_long k, void** dptr, sf_long* count) {
    if (!p || k < 0 || k >= (sf_long)p->segments.size() || !dptr || !count) return SF_ERR_ARG;
    if (p->dry) return SF_ERR_ARG;
    HIP_TRY(hipSetDevice(p->device));
    const Segment& sg = p->segments[(size_t)k];
    if (sg.owner_gi < 0) return SF_ERR_ARG;
    double* buf = p->d_scratch + 2 * p->scratch_elems;
    const bool mine = sg.owner_gi == __builtin_popcount(sg.mask & ((1u << p->rank) - 1u));
    if (mine) { const int rc = seg_copy(p, sg, buf, true, p->stream); if (rc) return rc; }
    else if (sg.packed > 0) HIP_TRY(hipMemsetAsync(buf, 0, (size_t)sg.packed * sizeof(double), p->stream));
    *dptr = (void*)buf;
    *count = sg.packed;
    return SF_OK;
}
int sf_seg_bcast_finish(sf_chol_plan* p, sf_long k) {
    if (!p || k < 0 || k >= (sf_long)p->segments.size()) return SF_ERR_ARG;
    if (p->dry) return SF_ERR_ARG;
    HIP_TRY(hipSetDevice(p->device));
    const Segment& sg = p->segments[(size_t)k];
    if (sg.owner_gi < 0) return SF_ERR_ARG;
    const int rc = seg_copy(p, sg, p->d_scratch + 2 * p->scratch_elems, false, p->stream);
    if (rc) return rc;
    // the block column is final on this rank NOW: its copy-back pieces carry "ready = lc + 1"; with launches behind the broadcast
    // run_launches publishes them after the first of those, without any it is done here
    if (p->dl_active && sg.lc == sg.l1) HIP_TRY(dl_publish(p, sg.lc + 1));
    return run_launches(p, sg.lc, sg.l1, false, k + 1 == (sf_long)p->segments.size(), 0);
}
int sf_seg_early(const sf_chol_plan* p, sf_long k) { return (p && k >= 0 && k < (sf_long)p->segments.size() && p->segments[k].early) ? 1 : 0; }

extern "C" {

int sf_chol_plan_set_stream(sf_chol_plan* p, void* stream) {
    if (!p) return SF_ERR_ARG;
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->stream && p->own_stream) HIP_TRY(hipStreamDestroy(p->stream));
    p->stream = (hipStream_t)stream;
    p->own_stream = false;
    return SF_OK;
}


// SF_GRAPH=1: a whole resident factorization (memsets, assembly, every launch of the plan) is captured ONCE into a hipGraph and
// replayed -- one submission instead of ~1,500 at 128^3.  Only the plain resident form: no overlapped download (its events are
// recorded between the launches for host threads to wait on), no profiling, one rank, one lane.  The graph is rebuilt when a kernel
// argument it froze changes (the LU pivot threshold / perturbation scale, the plan's stream).
static int factorize_graph(sf_chol_plan* p, int sync) {
    HIP_TRY(hipSetDevice(p->device));
    const double eps = p->lu ? p->piv_perturb * p->amax : 0.0, tol = p->lu ? p->piv_tol : 0.0;
    if (p->graph_exec && (p->graph_tol != tol || p->graph_eps != eps || p->graph_stream != p->stream)) {
        (void)hipGraphExecDestroy(p->graph_exec);
        p->graph_exec = nullptr;
    }
    bool fresh = false;
    if (!p->graph_exec) {
        fresh = true;
        hipGraph_t g = nullptr;
        HIP_TRY(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
        p->capturing = true;
        const int rc = run_launches(p, 0, p->launches.size(), true, true, 0);
        p->capturing = false;
        const hipError_t e = hipStreamEndCapture(p->stream, &g);
        if (rc || e != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); (void)hipGetLastError(); return rc ? rc : SF_ERR_HIP; }
        const hipError_t ei = hipGraphInstantiate(&p->graph_exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ei != hipSuccess) { p->graph_exec = nullptr; (void)hipGetLastError(); return SF_ERR_HIP; }
        p->graph_tol = tol; p->graph_eps = eps; p->graph_stream = p->stream;
    }
    // (the captured kernels compare the flags with graph_epoch and the graph clears the flags itself; p->epoch stays what it is for
    //  everybody else -- the generation of the factor on the device -- and moves on with every replay)
    if (!fresh) p->epoch = (p->epoch == 0x7fffffff) ? 1 : p->epoch + 1;
    p->packed_pending = -1;
    HIP_TRY(hipEventRecord(p->ev0, p->stream));
    HIP_TRY(hipGraphLaunch(p->graph_exec, p->stream));
    HIP_TRY(hipEventRecord(p->ev1, p->stream));
    return sync ? sf_chol_plan_sync(p) : SF_OK;
}

int sf_chol_plan_factorize(sf_chol_plan* p, int sync) {
    if (p && !p->dry && p->values_set && p->use_graph && p->nranks == 1 && !p->partial && !p->dl_active && !p->profiling && !p->lookahead1) {
        const int rc = factorize_graph(p, sync);
        if (rc != SF_ERR_HIP) return rc;
        p->use_graph = false;               // capture is not available here: the eager path from now on
        (void)hipGetLastError();
    }
    return sf_chol_plan_factorize_phase(p, -1, sync);
}

// The whole of SparseFrame_validate on the device (C:3141-3266, L:3702-3858): b_i = 1 + i/n, the supernodal solve with the
// resident factor, r = A x - b from the plan's copy of the matrix, residual = |r|_inf / (|A|_1 |x|_inf + |b|_inf).  Nothing
// but the scalar comes back (x_host may be NULL).
int sf_chol_plan_validate(sf_chol_plan* p, sf_float* residual, sf_float* x_host) {
    if (!p || !residual) return SF_ERR_ARG;
    if (p->partial || (p->nsuper > 0 && !p->d_solve) || !p->values_set) return SF_ERR_ARG;
    *residual = 0.0;
    if (p->n <= 0) return SF_OK;
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    HIP_TRY(hipSetDevice(p->device));
    if (!p->d_resid) {
        HIP_TRY(hipMalloc((void**)&p->d_resid, (3 * (size_t)p->n + 4) * sizeof(double)));
        p->bytes_device += (3 * (size_t)p->n + 4) * sizeof(double);
    }
    std::vector<double> b(p->n), x(p->n);
    for (int64_t i = 0; i < p->n; ++i) b[i] = 1.0 + (double)i / (double)p->n;
    int rc = sf_chol_plan_solve(p, b.data(), x.data());          // leaves the solution in d_x
    if (rc) return rc;
    double* r = p->d_resid, *colsum = r + p->n, *bb = colsum + p->n, *norms = bb + p->n;
    HIP_TRY(hipMemsetAsync(norms, 0, 4 * sizeof(double), p->stream));
    const bool unsym = p->lu && !p->u_alias;
    sf::launch_residual(p->d_Lp, p->d_Li, p->d_Lx, unsym ? p->d_Up : nullptr, unsym ? p->d_Ui : nullptr, unsym ? p->d_Ux : nullptr,
                        (int32_t)p->n, p->d_x, r, colsum, bb, norms, p->stream);
    HIP_TRY(hipGetLastError());
    double h[4];
    HIP_TRY(hipMemcpyAsync(h, norms, sizeof(h), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    *residual = h[0] / (h[1] * h[2] + h[3]);
    if (x_host) memcpy(x_host, x.data(), p->n * sizeof(double));
    return SF_OK;
}

// ---------------------------------------------------------------------------------------------------
// Overlapped copy-back.  The reference copies finished blocks back on a second stream while it computes
// (s_cudaStream_copyback, C:2888-2895).  Here: a few host threads, each with its own HIP stream and two pinned
// staging slots.  A worker pulls the next piece (pieces are sorted by the launch that finishes them): it
// waits until the main thread has recorded the piece's event on the compute stream, makes its stream wait for that
// event, starts the D2H into one slot and, while that DMA runs, copies the previous slot into the caller's buffer.
// The caller's memory is never pinned or registered: fresh (never touched) pages of a just-malloc'ed Lsx are faulted in
// by the workers in parallel with the factorization (measured: hipHostRegister of fresh memory is a serial 0.06 s/GiB,
// 1.7 s for the 128^3 factor; tools/host_xfer_bench.hip).
// ---------------------------------------------------------------------------------------------------
} // extern "C"

#include <sys/mman.h>
#include <unistd.h>
#include <sched.h>
#include <cctype>

static void dl_fail(sf_chol_plan* p, int code) {
    int expect = 0;
    p->dl_error.compare_exchange_strong(expect, code);
    {
        std::lock_guard<std::mutex> g(p->dl_mu);
        p->dl_abort = true;
    }
    p->dl_cv.notify_all();
}

static double dl_now() {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec / 1e6;
}

// The copy workers move the whole factor (30 GB at 128^3) from the pinned staging ring into the caller's pageable Lsx while the
// factorization runs: ~55 GB/s of memcpy, which on a two-socket host is cheapest from the socket the device hangs on (the ring
// lives there).  SF_DL_PIN=1 confines the workers to the CPUs of the device's NUMA node (hipDeviceAttributeHostNumaId or the PCI
// device's sysfs entry), intersected with the affinity mask the process was given.  OFF by default: measured neutral on the
// two-socket box of this project (profiles/r02_f_struct_slow_mode.txt, part 3 -- the slow calls seen there had another cause, see
// the stream priorities in plan_create), and a library should not move its caller's threads around without being asked.
static void dl_lookup_cpus(sf_chol_plan* p) {
    p->dl_cpus_known = -1;
    const char* pin_env = sf_exp_env("SF_DL_PIN");
    if (!pin_env || atoi(pin_env) == 0) return;
    int node = -1;
    if (hipDeviceGetAttribute(&node, hipDeviceAttributeHostNumaId, p->device) != hipSuccess) { (void)hipGetLastError(); node = -1; }
    if (node < 0) {                                              // older runtimes: the PCI device's own sysfs entry
        char bdf[32] = {0}, path0[96];
        if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, p->device) == hipSuccess) {
            for (char* q = bdf; *q; ++q) *q = (char)tolower((unsigned char)*q);
            snprintf(path0, sizeof path0, "/sys/bus/pci/devices/%s/numa_node", bdf);
            if (FILE* f0 = fopen(path0, "r")) {
                if (fscanf(f0, "%d", &node) != 1) node = -1;
                fclose(f0);
            }
        } else (void)hipGetLastError();
    }
    if (getenv("SF_TRACE")) fprintf(stderr, "[sparseframe-hip]   device %d hangs on NUMA node %d\n", p->device, node);
    if (node < 0) return;
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return;
    char buf[4096];
    const bool got = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!got) return;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    for (char* q = buf; *q;) {                                   // "0-63,128-191"
        char* e = nullptr;
        const long a = strtol(q, &e, 10);
        if (e == q) break;
        long b = a;
        if (*e == '-') { q = e + 1; b = strtol(q, &e, 10); if (e == q) break; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
            if (c >= 0 && CPU_ISSET((int)c, &allowed)) p->dl_cpus.push_back((int)c);
        q = (*e == ',') ? e + 1 : e;
        if (*e != ',') break;
    }
    if (!p->dl_cpus.empty()) p->dl_cpus_known = 1;
}

static void dl_worker(sf_chol_plan* p, int w) {
    if (p->dl_cpus_known == 1) {
        cpu_set_t set;
        CPU_ZERO(&set);
        for (int c : p->dl_cpus) CPU_SET(c, &set);
        (void)sched_setaffinity(0, sizeof set, &set);            // this thread only; a failure leaves it where it was
    }
    struct CpuNote { sf_chol_plan* p; int w; ~CpuNote() { p->dl_last_cpu[w] = sched_getcpu(); } } note{p, w};
    if (hipSetDevice(p->device) != hipSuccess) { dl_fail(p, SF_ERR_HIP); return; }
    hipStream_t ws = p->dl_streams[w];
    const size_t np = p->dl_pieces.size();
    const int64_t DL_SLOT = p->dl_slot;
    const int W = p->dl_workers;
    size_t prev = np;
    int prev_slot = 0, slot = 0;
    auto drain = [&](size_t k, int sl) -> bool {
        const DlPiece& pc = p->dl_pieces[k];
        if (hipEventSynchronize(p->dl_done[w][sl]) != hipSuccess) return false;
        if (pc.group >= 0 && p->dl_group_left) {        // out of core: this piece has left the device
            if (p->dl_group_left[(size_t)pc.group].fetch_sub(1) == 1) {
                { std::lock_guard<std::mutex> g(p->dl_mu); }
                p->dl_cv.notify_all();
            }
        }
        if (!p->dl_trace.empty()) p->dl_trace[3 * k + 1] = dl_now() - p->dl_t0;
        const double* ring = p->h_ring + ((int64_t)w * 2 + sl) * DL_SLOT;
        if (pc.s0 >= 0) {
            // LU, direct form: ring = [ L run | U^T run ]; column j of the reference panel = L column (nsrow) followed by rows
            // [nscol, nsrow) of the U^T column
            const double* rl = ring;
            const double* ru = ring + pc.dev_count;
            for (int32_t s = pc.s0; s < pc.s1; ++s) {
                const int64_t nscol = p->h_Super[s + 1] - p->h_Super[s], nsrow = p->h_Lsip[s + 1] - p->h_Lsip[s];
                const int64_t nb = nsrow - nscol, lda = nsrow + nb;
                const int64_t j0 = pc.ld > 0 ? pc.j0 : 0, nc = pc.ld > 0 ? pc.ncols : nscol;
                double* dst = p->dl_host + p->h_Lsxp[s] + j0 * lda;
                // 2-D U part (one supernode): nb values per column; whole panels: full columns, the first nscol rows skipped
                const int64_t ustride = pc.ld > 0 ? nb : nsrow, uskip = pc.ld > 0 ? 0 : nscol;
                for (int64_t c = 0; c < nc; ++c) {
                    memcpy(dst + c * lda, rl + c * nsrow, (size_t)nsrow * sizeof(double));
                    if (nb > 0) memcpy(dst + c * lda + nsrow, ru + c * ustride + uskip, (size_t)nb * sizeof(double));
                }
                rl += nc * nsrow;
                ru += nc * ustride;
            }
        } else if (pc.ld > 0) {
            const int64_t rows = pc.ld - pc.skip;
            for (int64_t c = 0; c < pc.ncols; ++c) {
                double* col = p->dl_host + pc.host_off + c * pc.ld;
                memset(col, 0, (size_t)pc.skip * sizeof(double));
                memcpy(col + pc.skip, ring + c * rows, (size_t)rows * sizeof(double));
            }
        } else {
            memcpy(p->dl_host + pc.host_off, ring, (size_t)pc.count * sizeof(double));
        }
        if (!p->dl_trace.empty()) p->dl_trace[3 * k + 2] = dl_now() - p->dl_t0;
        return true;
    };
    // pieces are pulled from one shared counter, in ready order: streams do not all get the same share of the SDMA
    // engines (measured: with a static round-robin split one worker of three finished at 570 ms, the others at 940 and
    // 1070 ms), so the split has to be dynamic
    (void)W;
    for (size_t k = p->dl_next_piece.fetch_add(1); k < np; k = p->dl_next_piece.fetch_add(1)) {
        const DlPiece& pc = p->dl_pieces[k];
        {
            std::unique_lock<std::mutex> g(p->dl_mu);
            if (prev < np && !(p->dl_abort || p->dl_published > (size_t)pc.ev)) {
                // nothing to fetch yet: finish the piece in hand first (an out-of-core plan's enqueue thread may be waiting for
                // exactly that piece before it publishes anything further)
                g.unlock();
                if (!drain(prev, prev_slot)) { dl_fail(p, SF_ERR_HIP); return; }
                prev = np;
                g.lock();
            }
            p->dl_cv.wait(g, [&] { return p->dl_abort || p->dl_published > (size_t)pc.ev; });
            if (p->dl_abort) return;
        }
        if (!p->dl_trace.empty()) p->dl_trace[3 * k] = dl_now() - p->dl_t0;
        double* hslot = p->h_ring + ((int64_t)w * 2 + slot) * DL_SLOT;
        // The piece's event is waited for on the HOST, by this thread, and the copy is enqueued with nothing in front of it.  A
        // device-side wait (hipStreamWaitEvent, SF_DL_HOST_WAIT=0) sits in the hardware queue the runtime has put this stream on
        // and holds up whatever else shares that queue -- another worker's copy of a piece that has long been ready, or a stream
        // of the caller's (the runtime multiplexes all streams of one priority over 4 hardware queues; plan_create has the story).
        // While the event is still in the future, the previous piece is copied out of the ring first.
        bool ok = true;
        if (p->dl_host_wait) {
            if (prev < np) {
                const hipError_t q = hipEventQuery(p->dl_events[pc.ev]);
                if (q != hipSuccess) {
                    (void)hipGetLastError();            // hipErrorNotReady is not an error
                    ok = drain(prev, prev_slot);
                    prev = np;
                }
            }
            ok = ok && hipEventSynchronize(p->dl_events[pc.ev]) == hipSuccess;
        } else {
            ok = hipStreamWaitEvent(ws, p->dl_events[pc.ev], 0) == hipSuccess;
        }
        const double* src = p->d_Lsx + pc.dev_off;
        if (pc.s0 >= 0) {
            // LU, direct form: the L run as it is; the U^T run as it is (whole supernodes) or rows [nscol, nsrow) of its columns
            const double* usrc = src + p->xC;
            ok = ok && hipMemcpyAsync(hslot, src, (size_t)pc.dev_count * sizeof(double), hipMemcpyDeviceToHost, ws) == hipSuccess;
            if (pc.ld > 0) {
                const int64_t nscol = p->h_Super[pc.s0 + 1] - p->h_Super[pc.s0], nb = pc.ld - nscol;
                if (nb > 0)
                    ok = ok && hipMemcpy2DAsync(hslot + pc.dev_count, (size_t)nb * sizeof(double), usrc + nscol, (size_t)pc.ld * sizeof(double),
                                                (size_t)nb * sizeof(double), (size_t)pc.ncols, hipMemcpyDeviceToHost, ws) == hipSuccess;
            } else {
                ok = ok && hipMemcpyAsync(hslot + pc.dev_count, usrc, (size_t)pc.dev_count * sizeof(double), hipMemcpyDeviceToHost, ws) == hipSuccess;
            }
            ok = ok && hipEventRecord(p->dl_done[w][slot], ws) == hipSuccess;
            if (ok && prev < np) ok = drain(prev, prev_slot);
            if (!ok) { dl_fail(p, SF_ERR_HIP); return; }
            prev = k;
            prev_slot = slot;
            slot ^= 1;
            continue;
        }
        if (ok && p->lu) {
            double* dslot = p->d_ring + ((int64_t)w * 2 + slot) * DL_SLOT;
            sf::launch_pack_lu(p->d_Super, p->d_Lsip, p->d_Xp, p->d_Lsxp, (int32_t)p->nsuper, p->d_Lsx, p->d_Lsx + p->xC,
                               dslot, pc.host_off, pc.host_off + pc.count, ws);
            ok = hipGetLastError() == hipSuccess;
            src = dslot;
        }
        if (pc.ld > 0) {
            const size_t rb = (size_t)(pc.ld - pc.skip) * sizeof(double);
            ok = ok && hipMemcpy2DAsync(hslot, rb, src + pc.skip, (size_t)pc.ld * sizeof(double), rb, (size_t)pc.ncols, hipMemcpyDeviceToHost, ws) == hipSuccess;
        } else {
            ok = ok && hipMemcpyAsync(hslot, src, (size_t)pc.count * sizeof(double), hipMemcpyDeviceToHost, ws) == hipSuccess;
        }
        ok = ok && hipEventRecord(p->dl_done[w][slot], ws) == hipSuccess;
        if (ok && prev < np) ok = drain(prev, prev_slot);
        if (!ok) { dl_fail(p, SF_ERR_HIP); return; }
        prev = k;
        prev_slot = slot;
        slot ^= 1;
    }
    if (prev < np && !drain(prev, prev_slot)) dl_fail(p, SF_ERR_HIP);
}

int sf_dl_begin(sf_chol_plan* p, double* host_out) {
    if (!p || !host_out || p->dl_active) return SF_ERR_ARG;
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    HIP_TRY(hipSetDevice(p->device));
    if (p->dl_cpus_known == 0) dl_lookup_cpus(p);
    if (!p->h_ring) {
        const size_t rb = (size_t)std::max(p->dl_workers, p->dl_workers_fresh) * 2 * p->dl_slot * sizeof(double);
        HIP_TRY(hipHostMalloc((void**)&p->h_ring, rb, hipHostMallocDefault));
        if (p->lu && !p->dl_lu_direct) {
            HIP_TRY(hipMalloc((void**)&p->d_ring, rb));
            p->bytes_device += rb;
        }
        for (int w = 0; w < std::max(p->dl_workers, p->dl_workers_fresh); ++w) {
            HIP_TRY(hipStreamCreateWithFlags(&p->dl_streams[w], hipStreamNonBlocking));
            for (int k = 0; k < 2; ++k) HIP_TRY(hipEventCreateWithFlags(&p->dl_done[w][k], hipEventDisableTiming));
        }
    }
    // transparent huge pages for the destination (this box: THP = madvise): 512x fewer first-touch faults when the
    // caller hands over a just-malloc'ed Lsx, harmless otherwise
    if (p->xsize > 0) {
        const uintptr_t pg = (uintptr_t)2 << 20;
        const uintptr_t a = ((uintptr_t)host_out + pg - 1) & ~(pg - 1), b = ((uintptr_t)(host_out + p->xsize)) & ~(pg - 1);
        if (b > a) (void)madvise((void*)a, b - a, MADV_HUGEPAGE);
    }
    p->dl_host = host_out;
    p->dl_next_ev = 0;
    p->dl_next_piece.store(0);
    if (p->ooc_groups > 1) {
        if (!p->dl_group_left) p->dl_group_left.reset(new std::atomic<int64_t>[(size_t)p->ooc_groups]);
        for (int g = 0; g < p->ooc_groups; ++g) p->dl_group_left[(size_t)g].store(p->dl_group_pieces[(size_t)g]);
    }
    p->dl_published = 0;
    p->dl_abort = false;
    p->dl_error.store(0);
    p->dl_active = true;
    p->dl_threads.clear();
    // SF_DL_TRACE=path: per piece, the times (ms since the start of the download) at which its event was published, its DMA
    // finished and its copy into the caller's buffer finished
    p->dl_trace.clear();
    if (getenv("SF_DL_TRACE")) p->dl_trace.assign(3 * p->dl_pieces.size(), 0.0);
    p->dl_t0 = dl_now();
    // A destination whose pages have never been touched (the first call after SparseFrame_analyze malloc'ed Lsx) makes every worker's
    // memcpy fault its pages in as it goes, and 4 workers no longer keep up with the factorization: 6 bring the first struct call at
    // 128^3 from 714 to 662 ms; on touched pages 4 are enough and 6 only share hardware queues (sf_plan_internal.h).  mincore() on a
    // sample of the destination's pages tells the two apart.
    int nw_call = p->dl_workers;
    if (p->dl_workers_fresh > nw_call && p->xsize > (int64_t)(64 << 20)) {
        const long pg = sysconf(_SC_PAGESIZE);
        int resident = 0, probed = 0;
        for (int k = 0; k < 16 && pg > 0; ++k) {
            const uintptr_t a = ((uintptr_t)(host_out + (p->xsize / 16) * k + p->xsize / 32)) & ~((uintptr_t)pg - 1);
            unsigned char vec = 0;
            if (mincore((void*)a, (size_t)pg, &vec) == 0) { ++probed; resident += vec & 1; }
        }
        if (probed > 0 && 2 * resident < probed) nw_call = p->dl_workers_fresh;
    }
    p->dl_workers_last = nw_call;
    const int nw = (int)std::min<size_t>(nw_call, p->dl_pieces.size());
    for (int w = 0; w < nw; ++w) p->dl_threads.emplace_back(dl_worker, p, w);
    // (First touch of a just-malloc'ed destination is left to the copy workers.  Helper threads that populate the page tables ahead
    // of them -- MADV_POPULATE_WRITE over the array in address order -- were measured and make the first call SLOWER: 637 ms of
    // copy-back with none, 857 ms with 2, 1100 ms with 8 or 16 at 128^3 (profiles/r03_d_first_call_touch_threads.txt): the page
    // allocator serialises them with the workers' own faults.)
    return SF_OK;
}

int sf_dl_end(sf_chol_plan* p) {
    if (!p || !p->dl_active) return SF_ERR_ARG;
    // a factorization that stopped early (an error in the enqueue path) must still release the workers
    if (p->dl_next_ev < p->dl_ev_ready.size()) {
        // (everything: an owner-computes block at the very end of the plan is ready "one launch after" its chain, see plan_create)
        if (hipSetDevice(p->device) != hipSuccess || dl_publish(p, (size_t)-1) != hipSuccess) dl_fail(p, SF_ERR_HIP);
        if (p->dl_next_ev < p->dl_ev_ready.size()) dl_fail(p, SF_ERR_HIP);
    }
    for (std::thread& t : p->dl_threads) t.join();
    p->dl_threads.clear();
    p->dl_active = false;
    p->dl_host = nullptr;
    if (!p->dl_trace.empty()) {
        if (FILE* f = fopen(getenv("SF_DL_TRACE") ? getenv("SF_DL_TRACE") : "/dev/null", "w")) {
            fprintf(f, "piece,ready_launch,doubles,t_published_ms,t_dma_done_ms,t_copied_ms\n");
            for (size_t k = 0; k < p->dl_pieces.size(); ++k)
                fprintf(f, "%zu,%zu,%lld,%.3f,%.3f,%.3f\n", k, p->dl_pieces[k].ready, (long long)p->dl_pieces[k].count,
                        p->dl_trace[3 * k], p->dl_trace[3 * k + 1], p->dl_trace[3 * k + 2]);
            fclose(f);
        }
    }
    return p->dl_error.load();
}

extern "C" {

// values H2D + numeric factorization + factor D2H into `host_out` (reference layout, xsize doubles), the download
// overlapped with the computation.  What one SparseFrame_factorize call does once its plan exists.
int sf_chol_plan_factorize_to_host(sf_chol_plan* p, const sf_float* Lx, const sf_float* Ux, sf_float* host_out) {
    if (!p || (!host_out && p->xsize > 0) || p->nranks > 1) return SF_ERR_ARG;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int rc = p->lu ? sf_lu_plan_set_values(p, Lx, Ux) : sf_chol_plan_set_values(p, Lx);
    if (rc) return rc;
    if (p->xsize <= 0) return sf_chol_plan_factorize(p, 1);
    if ((rc = sf_dl_begin(p, host_out))) return rc;
    rc = sf_chol_plan_factorize(p, 0);
    const int rc_dl = sf_dl_end(p);
    const int rc_sync = sf_chol_plan_sync(p);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    p->last_to_host_ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) / 1e6;
    return rc ? rc : (rc_sync ? rc_sync : rc_dl);
}

int sf_chol_plan_top_region(sf_chol_plan* p, void** dptr, sf_long* count) {
    if (!p || !dptr || !count) return SF_ERR_ARG;
    *dptr = (void*)(p->d_Lsx + p->top_off);
    *count = p->top_size;
    return SF_OK;
}

int sf_chol_plan_get_factor(sf_chol_plan* p, sf_float* Lsx) {
    if (!p || (!Lsx && p->xsize > 0)) return SF_ERR_ARG;
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    if (p->ooc_groups > 1) return SF_ERR_ARG;       // an out-of-core plan's factor never exists on the device as a whole
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->xsize <= 0) return SF_OK;
    if (!p->lu && !p->partial) {
        HIP_TRY(hipMemcpy(Lsx, p->d_Lsx, p->xsize * sizeof(double), hipMemcpyDeviceToHost));
        return SF_OK;
    }
    if (!p->lu) {
        // sharded plan: copy the panels stored here to their reference positions, one copy per run of
        // supernodes that is contiguous in both layouts; panels of other ranks are left untouched
        sf_long s = 0;
        while (s < p->nsuper) {
            if (p->h_XP[s] < 0) { ++s; continue; }
            sf_long e = s;
            int64_t len = 0;
            while (e < p->nsuper && p->h_XP[e] == p->h_XP[s] + len) {
                len += p->h_Lsxp[e + 1] - p->h_Lsxp[e];
                ++e;
            }
            HIP_TRY(hipMemcpy(Lsx + p->h_Lsxp[s], p->d_Lsx + p->h_XP[s], len * sizeof(double), hipMemcpyDeviceToHost));
            s = e;
        }
        return SF_OK;
    }
    // (sharded LU: panels not stored on this rank come back as zeros, k_pack_lu skips them)
    if (!p->d_pack) {
        HIP_TRY(hipMalloc((void**)&p->d_pack, p->xsize * sizeof(double)));
        p->bytes_device += p->xsize * sizeof(double);
    }
    sf::launch_pack_lu(p->d_Super, p->d_Lsip, p->d_Xp, p->d_Lsxp, (int32_t)p->nsuper, p->d_Lsx, p->d_Lsx + p->xC,
                       p->d_pack, 0, p->xsize, p->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(Lsx, p->d_pack, p->xsize * sizeof(double), hipMemcpyDeviceToHost));
    return SF_OK;
}

// values [e_begin, e_end) of the factor in the reference layout (a whole plan only): what the struct path samples to make sure a
// host copy still is what the device holds before it solves with the resident factor
int sf_chol_plan_get_factor_range(sf_chol_plan* p, sf_long e_begin, sf_long e_end, sf_float* out) {
    if (!p || p->partial || e_begin < 0 || e_end > p->xsize || e_end < e_begin || (!out && e_end > e_begin)) return SF_ERR_ARG;
    if (e_end == e_begin) return SF_OK;
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    HIP_TRY(hipSetDevice(p->device));
    if (!p->lu) {
        HIP_TRY(hipMemcpyAsync(out, p->d_Lsx + e_begin, (e_end - e_begin) * sizeof(double), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        return SF_OK;
    }
    double* tmp = nullptr;
    HIP_TRY(hipMalloc((void**)&tmp, (e_end - e_begin) * sizeof(double)));
    sf::launch_pack_lu(p->d_Super, p->d_Lsip, p->d_Xp, p->d_Lsxp, (int32_t)p->nsuper, p->d_Lsx, p->d_Lsx + p->xC, tmp, e_begin, e_end, p->stream);
    hipError_t e1 = hipGetLastError();
    hipError_t e2 = hipMemcpyAsync(out, tmp, (e_end - e_begin) * sizeof(double), hipMemcpyDeviceToHost, p->stream);
    hipError_t e3 = hipStreamSynchronize(p->stream);
    (void)hipFree(tmp);
    return (e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess) ? SF_OK : SF_ERR_HIP;
}

int sf_plan_panel_hashes(sf_chol_plan* p, const uint64_t** out) {
    if (!p || !out || p->ooc_groups > 1) return SF_ERR_ARG;
    if (p->hash_epoch != p->epoch || p->h_hash.size() != (size_t)std::max<int64_t>(p->nsuper, 1)) {
        if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
        HIP_TRY(hipSetDevice(p->device));
        const size_t nb = (size_t)std::max<int64_t>(p->nsuper, 1) * sizeof(unsigned long long);
        unsigned long long* d_h = nullptr;
        HIP_TRY(hipMalloc((void**)&d_h, nb));
        const int64_t* xp = (p->lu || p->partial) ? p->d_Xp : p->d_Lsxp;
        hipError_t e0 = hipMemsetAsync(d_h, 0, nb, p->stream);
        sf::launch_factor_hash(p->d_Super, p->d_Lsip, xp, p->d_Lsxp, (int32_t)p->nsuper, p->d_Lsx, p->d_Lsx + p->xC, p->lu ? 1 : 0,
                               p->xsize, d_h, p->stream);
        hipError_t e1 = hipGetLastError();
        p->h_hash.assign((size_t)std::max<int64_t>(p->nsuper, 1), 0);
        hipError_t e2 = hipMemcpyAsync(p->h_hash.data(), d_h, nb, hipMemcpyDeviceToHost, p->stream);
        hipError_t e3 = hipStreamSynchronize(p->stream);
        (void)hipFree(d_h);
        if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { p->hash_epoch = -1; return SF_ERR_HIP; }
        p->hash_epoch = p->epoch;
    }
    *out = p->h_hash.data();
    return SF_OK;
}

int sf_lu_plan_set_pivoting(sf_lu_plan* p, double tol, double perturb) {
    if (!p || !p->lu || !(tol >= 0.0) || tol > 1.0 || !(perturb >= 0.0)) return SF_ERR_ARG;
    p->piv_tol = tol;
    p->piv_perturb = perturb;
    return SF_OK;
}

// pivpos[g] (global permuted index) = the row position original row g was given by the interchanges of its 64-column
// block; the identity where nothing moved, for panels not stored on this rank (left untouched) and when pivoting is off
int sf_lu_plan_get_pivots(sf_lu_plan* p, sf_long* pivpos) {
    if (!p || !p->lu || (!pivpos && p->n > 0)) return SF_ERR_ARG;
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (!(p->piv_tol > 0.0)) {
        for (int64_t s = 0; s < p->nsuper; ++s)
            if (p->h_XP[s] >= 0)
                for (int64_t j = p->h_Super[s]; j < p->h_Super[s + 1]; ++j) pivpos[j] = j;
        return SF_OK;
    }
    std::vector<int32_t> h(std::max<int64_t>(p->n, 1));
    HIP_TRY(hipMemcpy(h.data(), p->d_piv, p->n * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int64_t s = 0; s < p->nsuper; ++s)
        if (p->h_XP[s] >= 0)
            for (int64_t j = p->h_Super[s]; j < p->h_Super[s + 1]; ++j) pivpos[j] = h[j];
    return SF_OK;
}

int sf_lu_plan_factorize(sf_lu_plan* p, int sync) { return (p && p->lu) ? sf_chol_plan_factorize(p, sync) : SF_ERR_ARG; }
int sf_lu_plan_sync(sf_lu_plan* p) { return sf_chol_plan_sync(p); }
int sf_lu_plan_get_factor(sf_lu_plan* p, sf_float* Lsx) { return (p && p->lu) ? sf_chol_plan_get_factor(p, Lsx) : SF_ERR_ARG; }
double sf_lu_plan_stat(const sf_lu_plan* p, const char* name) { return sf_chol_plan_stat(p, name); }
int sf_lu_plan_set_profiling(sf_lu_plan* p, int on) { return sf_chol_plan_set_profiling(p, on); }
int sf_lu_plan_solve(sf_lu_plan* p, const sf_float* b_host, sf_float* x_host) { return (p && p->lu) ? sf_chol_plan_solve(p, b_host, x_host) : SF_ERR_ARG; }
int sf_lu_plan_destroy(sf_lu_plan* p) { return sf_chol_plan_destroy(p); }

void* sf_chol_plan_factor_device_ptr(sf_chol_plan* p) { return p ? (void*)p->d_Lsx : nullptr; }

int sf_chol_plan_set_profiling(sf_chol_plan* p, int on) {
    if (!p) return SF_ERR_ARG;
    p->profiling = on != 0;
    return SF_OK;
}

double sf_chol_plan_stat(const sf_chol_plan* p, const char* name) {
    if (!p || !name) return -1;
    const std::string k(name);
    if (k == "levels") return p->nlevels;
    if (k == "last_solve_ms") return p->last_solve_ms;
    if (k == "perturbed_pivots") return (double)p->last_perturbed;
    if (k == "pivot_tol") return p->piv_tol;
    if (k == "last_to_host_ms") return p->last_to_host_ms;
    if (k == "download_pieces") return (double)p->dl_pieces.size();
    if (k == "top_doubles") return (double)p->top_size;
    if (k == "stored_doubles") return (double)p->xC;
    if (k == "launches") return (double)p->launches.size();
    if (k == "gemm_tasks") return (double)p->n_gemm_tasks;
    if (k == "update_pairs") return (double)p->n_pairs;
    if (k == "flops_exec") return p->flops_exec;
    if (k == "flops_update") return p->flops_update;
    if (k == "flops_panel_gemm") return p->flops_panel_gemm;
    if (k == "scatter_elems") return p->scatter_elems;
    if (k == "bytes_device") return (double)p->bytes_device;
    if (k == "last_ms") return p->last_ms;
    if (k == "last_load_ms") return p->last_load_ms;
    if (k == "last_panel_ms") return p->last_panel_ms;
    if (k == "last_update_ms") return p->last_update_ms;
    if (k == "last_potrf_ms") return p->last_kind_ms[0];
    if (k == "last_trsm_ms") return p->last_kind_ms[1];
    if (k == "last_inner_gemm_ms") return p->last_kind_ms[2];
    if (k == "last_outer_gemm_ms") return p->last_kind_ms[4];
    if (k == "last_step_ms") return p->last_kind_ms[5];
    if (k == "last_small_update_ms") return p->last_kind_ms[6];
    if (k == "flops_update_small") return p->flops_update_small;
    if (k == "flops_outer_gemm") return p->flops_outer_gemm;
    if (k == "flops_tiles") return p->flops_tiles;
    if (k == "flops_tiles_update") return p->flops_tiles_update;
    return -1;
}

// x <- (L L^T)^{-1} b (Cholesky, C:3036-3139) or (L U)^{-1} b (LU, L:3592-3700) with the resident factor, permuted
// space.  LU: unit-lower forward sweep over the L panels, backward sweep over the U^T panels (U x = y <=> (U^T)^T x = y).
}   // extern "C"

void sf_solve_step_fwd(sf_chol_plan* p, size_t k, const double* base, int* sync, int* tickets, hipStream_t st) {
    const auto& s = p->solve_steps[k];
    const int32_t* piv = (p->lu && p->piv_tol > 0.0) ? p->d_piv : nullptr;
    const int unit = p->lu ? 1 : 0;
    int* tk = tickets + sf_chol_plan::SOLVE_TICKETS * k;
    if (s.small) {
        sf::launch_solve_small_fwd(p->d_solve + s.fwd_first, s.ndiag, base, p->d_Lsi, p->d_x, unit, piv, st);
    } else {
        sf::launch_solve_fwd(p->d_solve + s.fwd_first, s.fwd_count, s.big, base, p->d_Lsi, p->d_x, unit, piv, sync, tk, p->d_solve_sync, st);
    }
}

void sf_solve_step_bwd(sf_chol_plan* p, size_t k, const double* base, int* sync, int* tickets, hipStream_t st) {
    const auto& s = p->solve_steps[k];
    int* tk = tickets + sf_chol_plan::SOLVE_TICKETS * k;
    if (s.small) {
        sf::launch_solve_small_bwd(p->d_solve + s.bwd_first, s.ndiag, base, p->d_Lsi, p->d_x, st);
    } else if (p->solve_bwd_fused) {
        sf::launch_solve_bwd(p->d_solve + s.bwd_first, s.count, s.big, base, p->d_Lsi, p->d_x, sync, tk + 1, p->d_solve_sync, st, p->d_solveT);
    } else {
        sf::launch_solve_bwd(p->d_solve + s.bwd_first, s.nrows_tasks, 0, base, p->d_Lsi, p->d_x, sync, tk + 1, p->d_solve_sync, st);
        sf::launch_solve_bwd(p->d_solve + s.bwd_first + s.nrows_tasks, s.count - s.nrows_tasks, s.big, base, p->d_Lsi, p->d_x, sync, tk + 2,
                             p->d_solve_sync, st, p->d_solveT);
    }
}

extern "C" {

int sf_chol_plan_solve(sf_chol_plan* p, const sf_float* b_host, sf_float* x_host) {
    if (!p || !b_host || !x_host) return SF_ERR_ARG;
    if (p->partial || (p->nsuper > 0 && !p->d_solve)) return SF_ERR_ARG;
    const double* fwd_base = p->d_Lsx;
    const double* bwd_base = p->lu ? p->d_Lsx + p->xC : p->d_Lsx;
    if (p->dry) return SF_ERR_ARG;        // a schedule-only plan has no device side
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t st = p->stream;
    if (p->n <= 0) return SF_OK;
    HIP_TRY(hipMemcpyAsync(p->d_x, b_host, p->n * sizeof(double), hipMemcpyHostToDevice, st));
    // the plan's own event pair (ev0/ev1 time the factorization; sf_chol_plan_sync has read them by now): nothing is
    // created here, so an error return leaks nothing
    hipEvent_t e0 = p->ev_s0, e1 = p->ev_s1;
    HIP_TRY(hipEventRecord(e0, st));
    const size_t nst = p->solve_steps.size();
    int* sync = p->d_solve_sync + 1;
    int* tickets = sync + p->n_solve_sync;
    HIP_TRY(hipMemsetAsync(p->d_solve_sync, 0, (size_t)(1 + p->n_solve_sync + sf_chol_plan::SOLVE_TICKETS * nst) * sizeof(int), st));
    for (size_t k = 0; k < nst; ++k) sf_solve_step_fwd(p, k, fwd_base, sync, tickets, st);
    // (row-major copies of the top steps' diagonal blocks, from the factor as it is now: 117 MB at 128^3, ~0.1 ms)
    sf::launch_solve_transpose_diag(p->d_solve, p->d_solveT_list, p->n_solveT, bwd_base, p->d_solveT, st);
    for (size_t k = nst; k-- > 0;) sf_solve_step_bwd(p, k, bwd_base, sync, tickets, st);
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(x_host, p->d_x, p->n * sizeof(double), hipMemcpyDeviceToHost, st));
    int sinfo = 0;
    HIP_TRY(hipMemcpyAsync(&sinfo, p->d_solve_sync, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (sinfo) return SF_ERR_HIP;       // a bounded in-launch wait ran out (never seen)
    float ms = 0;
    if (elapsed_ms(&ms, e0, e1)) p->last_solve_ms = ms;
    return SF_OK;
}

}  // extern "C"

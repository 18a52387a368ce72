// Hand-written gfx950 (CDNA4) kernels of the supernodal Cholesky / no-pivot LU numeric phase.
// Reference citations: C: = Cholesky/Source/SparseFrame.c, L: = LU/Source/SparseFrame.c, CK: = Cholesky/Source/cuda_kernel.cu.
//
//   k_load_panels   : SparseFrame_loadA (C:1998-2028, L:2478-2536) for ALL supernodes in one launch; the inverse row map
//                     (createMap, CK:22-30) is replaced by a binary search in the supernode's sorted row list.
//   k_potrf_block   : dpotrf_('L') on one <= 64x64 diagonal block, one wavefront, registers only   (C:2135, C:2766)
//   k_getrf_block   : no-pivot LU of the block (magma_dgetrf_nopiv L:2653, cusolverDnDgetrf/NULL L:3344), same scheme
//   k_trsm_block    : dtrsm_('R','L','C','N') of a row tile against that block (C:2142, C:2773); LU: D from the
//                     other panel, optional unit diagonal (L:2660)
//   k_gemm<mode>    : fp64 MFMA (v_mfma_f64_16x16x4_f64) C -= Y X^T on LDS-staged panels, persistent grid (tiles in rounds
//                     + stream-K remainder)
//       mode 0 : in-panel updates (the reference's blocked-potrf SYRK/GEMM, C:2854-2863)
//       mode 1 : Schur-complement update of an ancestor panel: dsyrk+dgemm (C:2061-2070; LU: L:2570-2577) with the
//                mapped scatter-subtract (mappedSubtract, CK:62-124; CPU loops C:2073-2086, L:2583-2604) fused into
//                the epilogue as native global_atomic_add_f64, lower trapezoid only
//   k_update_small  : the mode-1 update for K <= 64 (bottom-level supernodes): one wave per 64x32 tile, no LDS
//   k_step<LU>      : one 64-column step of the in-panel factorization in ONE launch: left-looking update (MFMA) +
//                     POTRF / GETRF of the diagonal block + TRSM of the rows below (MFMA, from 16x16 inverses), with a
//                     flag hand-off from the diagonal workgroup to the row workgroups inside the launch
//   k_build_relmaps : createRelativeMap (CK:42-60), once per plan for every (descendant, ancestor) pair
//   k_pack_lu       : gathers the device's (L, U^T) panel pairs into the reference's packed LU panels (L:2514-2517)
//   k_solve_*       : level-scheduled triangular solves with the resident factor (host loops C:3074-3134)
//
// Wavefronts are 64 wide; all tilings below are written for that.
#include "sf_kernels.h"
#include <cstdlib>
#include "sf_wave.h"

namespace sf {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2), aligned(8)));   // 16-byte vector that may sit on an 8-byte boundary

// position of `key` in the ascending array a[0..n); key is known to be present
__device__ __forceinline__ int lower_bound_i32(const int32_t* __restrict__ a, int n, int32_t key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------------------------------
// assemble: one lane per matrix column; A[(j-Super[s])*nsrow + pos(i)] = Lx[p]
// The panels were zeroed by a memset on the same stream.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_load_panels(const int64_t* __restrict__ Lp, const int32_t* __restrict__ Li, const double* __restrict__ Lx, int32_t n,
              const int32_t* __restrict__ Super, const int32_t* __restrict__ SuperMap,
              const int64_t* __restrict__ Lsip, const int32_t* __restrict__ Lsi,
              const int64_t* __restrict__ Lsxp, double* __restrict__ Lsx, int skip_diag,
              const int8_t* __restrict__ load_mask) {
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int32_t s = SuperMap[j];
    if (load_mask && !load_mask[s]) return;     // multi-GPU: panel not stored here, or loaded by another rank
    const int32_t c0 = Super[s], c1 = Super[s + 1];
    const int64_t r0 = Lsip[s];
    const int32_t nsrow = (int32_t)(Lsip[s + 1] - r0);
    const int32_t nscol = c1 - c0;
    double* col = Lsx + Lsxp[s] + (int64_t)(j - c0) * nsrow;
    const int32_t* below = Lsi + r0 + nscol;
    for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) {
        const int32_t i = Li[p];
        if (skip_diag && i == j) continue;
        const int32_t si = (i < c1) ? (i - c0) : nscol + lower_bound_i32(below, nsrow - nscol, i);
        col[si] = Lx[p];
    }
}

// The same walk once per plan, keeping where every entry goes (offset into Lsx, -1: not loaded) -- so that the assembly of every
// later factorization is one coalesced pass over the values and the map (k_load_mapped) instead of a binary search per entry in
// dependent loads: config 3 (n = 10^6, 10^7 entries) 0.46 -> 0.07 ms per factorization.
__global__ void __launch_bounds__(256)
k_build_loadmap(const int64_t* __restrict__ Lp, const int32_t* __restrict__ Li, int32_t n,
                const int32_t* __restrict__ Super, const int32_t* __restrict__ SuperMap,
                const int64_t* __restrict__ Lsip, const int32_t* __restrict__ Lsi,
                const int64_t* __restrict__ Lsxp, int64_t base, int skip_diag, int64_t* __restrict__ map) {
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int32_t s = SuperMap[j];
    const int32_t c0 = Super[s], c1 = Super[s + 1];
    const int64_t r0 = Lsip[s];
    const int32_t nsrow = (int32_t)(Lsip[s + 1] - r0);
    const int32_t nscol = c1 - c0;
    const int64_t col = base + Lsxp[s] + (int64_t)(j - c0) * nsrow;
    const int32_t* below = Lsi + r0 + nscol;
    for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) {
        const int32_t i = Li[p];
        if (skip_diag && i == j) { map[p] = -1; continue; }
        const int32_t si = (i < c1) ? (i - c0) : nscol + lower_bound_i32(below, nsrow - nscol, i);
        map[p] = col + si;
    }
}

__global__ void __launch_bounds__(256)
k_load_mapped(const double* __restrict__ Lx, const int64_t* __restrict__ map, int64_t nnz, double* __restrict__ Lsx) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = map[p];
        if (m >= 0) Lsx[m] = Lx[p];
    }
}

void launch_build_loadmap(const int64_t* Lp, const int32_t* Li, int32_t n, const int32_t* Super, const int32_t* SuperMap,
                          const int64_t* Lsip, const int32_t* Lsi, const int64_t* Lsxp, int64_t base, int skip_diag, int64_t* map, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_build_loadmap, dim3((n + 255) / 256), dim3(256), 0, st, Lp, Li, n, Super, SuperMap, Lsip, Lsi, Lsxp, base, skip_diag, map);
}

void launch_load_mapped(const double* Lx, const int64_t* map, int64_t nnz, double* Lsx, hipStream_t st) {
    if (nnz <= 0) return;
    const int64_t blocks = std::min<int64_t>((nnz + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_load_mapped, dim3((unsigned)blocks), dim3(256), 0, st, Lx, map, nnz, Lsx);
}

void launch_load_panels(const int64_t* Lp, const int32_t* Li, const double* Lx, int32_t n,
                        const int32_t* Super, const int32_t* SuperMap, const int64_t* Lsip, const int32_t* Lsi,
                        const int64_t* Lsxp, double* Lsx, int skip_diag, const int8_t* load_mask, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_load_panels, dim3((n + 255) / 256), dim3(256), 0, st,
                       Lp, Li, Lx, n, Super, SuperMap, Lsip, Lsi, Lsxp, Lsx, skip_diag, load_mask);
}

// ---------------------------------------------------------------------------------------------------
// Cholesky of a b x b (b <= 64) diagonal block, lower.  ONE wavefront per block, no LDS, no barrier:
// lane r keeps row r of the block in registers (a[c] = A(r,c)); at step j the pivot and the multipliers
// l(c,j) are broadcast out of lane j / lane c with v_readlane (the lane index is a compile-time
// constant after unrolling, so the broadcast lands in SGPRs and feeds v_fma_f64 directly).
// Rows/columns beyond b are padded with the identity.
// ---------------------------------------------------------------------------------------------------
// 1 / sqrt(v) and 1 / v to full fp64 accuracy from the hardware's 24-bit approximations (v_rsq_f64, v_rcp_f64: 5e-8 relative) and ONE
// third-order step --  r (1 + e/2 + 3 e^2 / 8), e = 1 - v r^2;  c (1 + e + e^2), e = 1 - v c  -- instead of two Newton steps: one
// dependent operation less on the sequential chains of the panel factorizations (5 instead of 6, 3 instead of 4) AND closer to the
// correctly rounded value: 1.24 / 1.00 ulp worst case over 4 M arguments against 2.18 / 1.69 (tools/experiments/rsq_accuracy.hip).
__device__ __forceinline__ double rsqrt_full(double v) {
    const double r = __builtin_amdgcn_rsq(v);
    const double e = __builtin_fma(-(v * r), r, 1.0);
    return __builtin_fma(r, e * __builtin_fma(e, 0.375, 0.5), r);
}
__device__ __forceinline__ double rcp_full(double v) {
    const double c = __builtin_amdgcn_rcp(v);
    const double e = __builtin_fma(-v, c, 1.0);
    return __builtin_fma(c, __builtin_fma(e, e, e), c);
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// W = the smallest of 8 / 16 / 32 / 64 that holds the block: the elimination is fully unrolled over W columns (registers), and
// at the bottom levels of a 2-D problem most blocks have a handful of columns -- with W = 64 for all, a 3-column block cost the
// 2,016 FMAs + 4,032 v_readlane of a 64-column one (config 3: 21,000 such blocks = 0.30 ms, compute-bound on padding)
template <int W>
__device__ __forceinline__ void potrf_block_w(const PotrfTask& t, double* __restrict__ Lsx, int* __restrict__ info) {
    double* A = Lsx + t.panel + t.diag + (int64_t)t.diag * t.ld;
    const int b = t.b;
    const int lane = threadIdx.x;
    const int64_t ld = t.ld;

    double a[W];
#pragma unroll
    for (int c = 0; c < W; ++c) {
        double v = (c == lane) ? 1.0 : 0.0;
        if (lane < b && c <= lane) v = A[lane + c * ld];
        a[c] = v;
    }
    bool bad = false;
    double dnext = a[0];        // the next column's diagonal entry, formed in its own lane (see k_step's panel)
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const double djj = readlane_f64(dnext, j);
        bad = bad || !(djj > 0.0);          // also catches NaN; padded rows have djj = 1
        // 1/sqrt(djj) from v_rsq_f64 + one third-order step (rsqrt_full), the column scaled by it: the IEEE sqrt and divide sequences
        // are ~10x longer and sit on the sequential critical path.  Every lane multiplies: lane j's own entry IS djj, so it gets
        // djj / sqrt(djj) = the diagonal without a select; lanes above the diagonal carry values nobody reads (they only ever feed
        // other entries above the diagonal, and the store below keeps to the lower triangle)
        const double rinv = rsqrt_full(djj);
        const double lj = a[j] * rinv;
        a[j] = lj;
        if (j + 1 < W) dnext = __builtin_fma(-lj, lj, a[j + 1]);
#pragma unroll
        for (int c = j + 1; c < W; ++c) a[c] = __builtin_fma(-lj, readlane_f64(lj, c), a[c]);
    }
    if (bad && lane == 0) atomicOr(info, 1);
#pragma unroll
    for (int c = 0; c < W; ++c)
        if (lane < b && c <= lane) A[lane + c * ld] = a[c];
}

__global__ void __launch_bounds__(64)
k_potrf_block(const PotrfTask* __restrict__ tasks, double* __restrict__ Lsx, int* __restrict__ info) {
    const PotrfTask t = tasks[blockIdx.x];
    if (t.b <= 8) potrf_block_w<8>(t, Lsx, info);
    else if (t.b <= 16) potrf_block_w<16>(t, Lsx, info);
    else if (t.b <= 32) potrf_block_w<32>(t, Lsx, info);
    else potrf_block_w<NB>(t, Lsx, info);
}

void launch_potrf(const PotrfTask* tasks, int ntasks, double* Lsx, int* info, hipStream_t st) {
    if (ntasks <= 0) return;
    hipLaunchKernelGGL(k_potrf_block, dim3(ntasks), dim3(64), 0, st, tasks, Lsx, info);
}

// ---------------------------------------------------------------------------------------------------
// LU of a b x b (b <= 64) diagonal block in ONE wavefront, lane r holds row r (a[c] = D(r,c), identity padding), the same
// register scheme as k_potrf_block.  Without pivoting this is the reference's magma_dgetrf_nopiv (LU/Source/SparseFrame.c:2653)
// / cusolverDnDgetrf with devIpiv = NULL (:3344).  With pivoting (PivotCtl, sf_kernels.h) the interchanges are IMPLICIT: rows
// never move between lanes; at column j a pivot lane p is chosen among the lanes not used yet, its row is broadcast with
// v_readlane (p is wave-uniform), and the lane remembers the position it was given.  The permutation is applied when the
// rows are stored.  RCP: multipliers by v_rcp_f64 + two Newton steps (k_step's variant) instead of the IEEE division.
// Returns this lane's final position; bad: a zero / NaN pivot was met and not perturbed.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_dyn_f64(double v, int l) {      // l wave-uniform, not a compile-time constant
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// W columns J0 .. J0 + W - 1 of the block (a[u] = D(lane, J0 + u)); `active` / `pos` carry the state of the lanes across the panels
// of a blocked factorization (k_step<true>), piv_rows[J] receives the row chosen at column J.  W = NB, J0 = 0 is the whole block.
template <bool RCP, int W>
__device__ __forceinline__ void getrf_panel_wave(double (&a)[W], int lane, int J0, int b, double tol, double eps, bool& bad, int& nperturbed,
                                                 int& pos, bool& active, int* piv_rows) {
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int J = J0 + j;
        int p = J;
        if (tol > 0.0 && J < b) {               // wave-uniform: no pivot search when pivoting is off or in the padding
            // The natural row keeps the pivot iff it is still free, non-zero, and NO free row has tol * |a_ij| > |a_jj| -- one
            // multiply, one compare and a ballot; no reduction.  (fl(tol * x) is monotonic in x, so this IS |a_jj| >= tol * max:
            // the decision of the reduction it replaces, which cost 2 us per 16-column panel.)  A NaN counts as larger.  Only when
            // the natural row fails is the arg-max looked for (two 32-bit DPP reductions and a ballot).
            const double nat = readlane_dyn_f64(a[j], J);
            const unsigned long long act = __ballot(active);
            const bool nat_free = ((act >> J) & 1ull) != 0;
            const double anat = fabs(nat);
            const unsigned long long larger = __ballot(active && !(tol * fabs(a[j]) <= anat));
            if (!(nat_free && larger == 0ull && nat != 0.0)) {
                double m;
                const int pm = wave_argmax_abs(a[j], active, &m);
                if (!(nat_free && fabs(nat) >= tol * m && nat != 0.0) && pm >= 0) p = pm;
            }
            p = __builtin_amdgcn_readfirstlane(p);
        }
        double piv = readlane_dyn_f64(a[j], p);
        if (J < b && eps > 0.0 && !(fabs(piv) >= eps) && piv == piv) {      // tiny (or zero) pivot: perturb
            piv = (piv < 0.0) ? -eps : eps;
            ++nperturbed;
            if (lane == p) a[j] = piv;
        }
        bad = bad || !(piv != 0.0);             // zero or NaN pivot; padded rows have piv = 1
        const bool elim = active && lane != p;
        double l;
        if (RCP) {
            const double rp = rcp_full(piv);
            l = elim ? a[j] * rp : 0.0;
        } else {
            l = elim ? a[j] / piv : 0.0;
        }
        if (elim) a[j] = l;
        if (lane == p) { pos = J; active = false; }
        if (piv_rows != nullptr && lane == 0) piv_rows[J] = p;
#pragma unroll
        for (int c = j + 1; c < W; ++c) a[c] -= l * readlane_dyn_f64(a[c], p);
    }
}

// NATURAL-PIVOT FAST PATH of getrf_panel_wave (k_step<true>): the same W columns eliminated with the natural rows as pivots -- the
// arithmetic of the general path when every natural pivot passes, instruction for instruction (same reciprocal, same fused
// multiply-adds: bit-identical results) -- as STRAIGHT-LINE code: the threshold test of column j (does a free row have
// tol * |a_ij| > |a_jj|?  is the pivot zero, NaN, or below the perturbation threshold?) only accumulates into a wave-uniform mask
// instead of steering a branch, so nothing on the column's critical path waits for a vector compare to reach the scalar unit and
// the compiler schedules the 16 columns as one block (the general form's per-column branches cost it a copy of all 16 registers per
// column and 48 spilled SGPRs; profiles/r03_k_step_stamps.txt: 5.9 us per 16-column panel against 2.4 us for Cholesky's).
// Valid while every earlier pivot of the block was natural too (lanes < J0 used, lanes >= J0 free).  Returns false when some
// natural pivot does NOT pass: the caller then reloads the panel and runs getrf_panel_wave on it (a[] is garbage in that case).
template <int W>
__device__ __forceinline__ bool getrf_panel_natural(double (&a)[W], int lane, int J0, int b, double tol, double eps) {
    unsigned long long viol = 0ull;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int J = J0 + j;
        const double piv = readlane_dyn_f64(a[j], J);
        const bool below = lane > J;
        viol |= __ballot(below && !(tol * fabs(a[j]) <= fabs(piv)));            // a NaN entry counts as larger
        if (J < b && (!(fabs(piv) >= eps) || piv == 0.0)) viol |= 1ull;           // zero, NaN or tiny pivot (wave-uniform test)
        const double rp = rcp_full(piv);
        const double l = below ? a[j] * rp : 0.0;
        if (below) a[j] = l;
#pragma unroll
        for (int c = j + 1; c < W; ++c) a[c] -= l * readlane_dyn_f64(a[c], J);
    }
    return viol == 0ull;
}

// The block lives in two panels: D(r,c), c < r (L, unit diagonal implied) in the L panel at (diag+r, diag+c); D(r,c), c >= r
// (U) in the U^T panel at (diag+c, diag+r).
template <int W>
__device__ __forceinline__ void getrf_block_w(const PotrfTask& t, double* __restrict__ Lsx, int64_t u_shift, int* __restrict__ info, const PivotCtl& pc) {
    double* PLd = Lsx + t.panel + t.diag + (int64_t)t.diag * t.ld;
    double* PUd = PLd + u_shift;
    const int b = t.b;
    const int lane = threadIdx.x;
    const int64_t ld = t.ld;

    double a[W];
#pragma unroll
    for (int c = 0; c < W; ++c) {
        double v = (c == lane) ? 1.0 : 0.0;
        if (lane < b && c < b) v = (c < lane) ? PLd[lane + c * ld] : PUd[c + lane * ld];
        a[c] = v;
    }
    bool bad = false, active = lane < b;
    int np = 0, pos = lane;
    getrf_panel_wave<false, W>(a, lane, 0, b, pc.tol, pc.eps, bad, np, pos, active, nullptr);
    if (bad && lane == 0) atomicOr(info, 1);
    if (np > 0 && lane == 0) atomicAdd(pc.nperturb, np);
#pragma unroll
    for (int c = 0; c < W; ++c) {
        if (lane < b && c < b) {
            if (c < pos) PLd[pos + c * ld] = a[c]; else PUd[c + pos * ld] = a[c];
        }
    }
    if (pc.pivpos && lane < b) {
        const int g0 = t.first_col + t.diag;
        pc.pivpos[g0 + lane] = g0 + pos;
        pc.pivinv[g0 + pos] = g0 + lane;
    }
}

// width-specialised like k_potrf_block
__global__ void __launch_bounds__(64)
k_getrf_block(const PotrfTask* __restrict__ tasks, double* __restrict__ Lsx, int64_t u_shift, int* __restrict__ info, PivotCtl pc) {
    const PotrfTask t = tasks[blockIdx.x];
    if (t.b <= 8) getrf_block_w<8>(t, Lsx, u_shift, info, pc);
    else if (t.b <= 16) getrf_block_w<16>(t, Lsx, u_shift, info, pc);
    else if (t.b <= 32) getrf_block_w<32>(t, Lsx, u_shift, info, pc);
    else getrf_block_w<NB>(t, Lsx, u_shift, info, pc);
}

void launch_getrf(const PotrfTask* tasks, int ntasks, double* Lsx, int64_t u_shift, int* info, PivotCtl pc, hipStream_t st) {
    if (ntasks <= 0) return;
    hipLaunchKernelGGL(k_getrf_block, dim3(ntasks), dim3(64), 0, st, tasks, Lsx, u_shift, info, pc);
}

// ---------------------------------------------------------------------------------------------------
// LU download: thread per value of the reference layout; supernode found by binary search in RefXp.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pack_lu(const int32_t* __restrict__ Super, const int64_t* __restrict__ Lsip, const int64_t* __restrict__ Xp,
          const int64_t* __restrict__ RefXp, int32_t nsuper, const double* __restrict__ PL, const double* __restrict__ PU,
          double* __restrict__ out, int64_t e_begin, int64_t e_end) {
    // values [e_begin, e_end) of the reference layout -> out[0 .. e_end - e_begin)
    for (int64_t e = e_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < e_end; e += (int64_t)gridDim.x * blockDim.x) {
        double* __restrict__ dst = out + (e - e_begin);
        int lo = 0, hi = nsuper;            // largest s with RefXp[s] <= e
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (RefXp[mid] <= e) lo = mid; else hi = mid;
        }
        const int s = lo;
        if (Xp[s] < 0) { *dst = 0.0; continue; }      // sharded plan: the panel lives on another rank
        const int64_t nscol = Super[s + 1] - Super[s], nsrow = Lsip[s + 1] - Lsip[s];
        const int64_t lda = 2 * nsrow - nscol;
        const int64_t off = e - RefXp[s];
        const int64_t j = off / lda, R = off % lda;
        const double* pl = PL + Xp[s] + j * nsrow;       // column j of the L panel
        double v;
        if (R < nscol) v = (R > j) ? pl[R] : PU[Xp[s] + j + R * nsrow];          // packed L11 \ U11: U(R,j) = PU(j,R)
        else if (R < nsrow) v = pl[R];                                             // L21
        else v = PU[Xp[s] + (R - nsrow + nscol) + j * nsrow];                      // U12^T
        *dst = v;
    }
}

void launch_pack_lu(const int32_t* Super, const int64_t* Lsip, const int64_t* Xp, const int64_t* RefXp, int32_t nsuper,
                    const double* PL, const double* PU, double* out, int64_t e_begin, int64_t e_end, hipStream_t st) {
    if (e_end <= e_begin) return;
    const int64_t blocks = (e_end - e_begin + 255) / 256;
    hipLaunchKernelGGL(k_pack_lu, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st,
                       Super, Lsip, Xp, RefXp, nsuper, PL, PU, out, e_begin, e_end);
}

// ---------------------------------------------------------------------------------------------------
// Fingerprint of the factor as the HOST sees it (reference layout), one 64-bit word per supernode:
//   H[s] = sum over the values e of panel s of  bits(v_e) * (2 e + 1) * K   (mod 2^64),  e = index in the reference layout.
// Order-independent (a sum), position-dependent (odd multiplier per index), a changed value always changes it, zeros add nothing.
// The struct path's solve compares it with the same sum over the caller's host array before it trusts the resident factor
// (sf_handlers.hip).  A workgroup walks a contiguous range of `chunk` values; a thread keeps the supernode of its current value
// (monotone walk) and flushes its partial sum when the supernode changes.  LU: the value is gathered from the (L, U^T) panel pair
// exactly as k_pack_lu does, so the result does not depend on which block columns k_lu_fill_u11 has touched.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_factor_hash(const int32_t* __restrict__ Super, const int64_t* __restrict__ Lsip, const int64_t* __restrict__ Xp,
              const int64_t* __restrict__ RefXp, int32_t nsuper, const double* __restrict__ PL, const double* __restrict__ PU, int lu,
              int64_t total, int64_t chunk, unsigned long long* __restrict__ H) {
    const int64_t begin = (int64_t)blockIdx.x * chunk, end = min(begin + chunk, total);
    if (begin >= end) return;               // (whole workgroup)
    const bool idle = begin + threadIdx.x >= end;           // last chunk: these threads add nothing but join the barriers below
    int64_t e = idle ? end - 1 : begin + threadIdx.x;
    int lo = 0, hi = nsuper;                // largest s with RefXp[s] <= e
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (RefXp[mid] <= e) lo = mid; else hi = mid;
    }
    int s = lo;
    unsigned long long acc = 0;
    int64_t s_end = RefXp[s + 1], s_beg = RefXp[s], xp = Xp[s];
    int64_t nscol = Super[s + 1] - Super[s], nsrow = Lsip[s + 1] - Lsip[s];
    if (idle) e = end;                      // nothing to add; stays for the barriers below
    for (; e < end; e += 256) {
        if (e >= s_end) {
            if (acc) atomicAdd(&H[s], acc);
            acc = 0;
            while (e >= RefXp[s + 1]) ++s;
            s_end = RefXp[s + 1]; s_beg = RefXp[s]; xp = Xp[s];
            nscol = Super[s + 1] - Super[s]; nsrow = Lsip[s + 1] - Lsip[s];
        }
        if (xp < 0) {                       // not stored on this rank: on to this thread's first value behind the panel
            e += ((s_end - e + 255) / 256 - 1) * 256;
            continue;
        }
        const int64_t off = e - s_beg;
        double v;
        if (!lu) {
            v = PL[xp + off];
        } else {
            const int64_t lda = 2 * nsrow - nscol;
            const int64_t j = off / lda, R = off % lda;
            if (R < nscol) v = (R > j) ? PL[xp + j * nsrow + R] : PU[xp + j + R * nsrow];
            else if (R < nsrow) v = PL[xp + j * nsrow + R];
            else v = PU[xp + (R - nsrow + nscol) + j * nsrow];
        }
        acc += (unsigned long long)__double_as_longlong(v) * ((2ull * (unsigned long long)e + 1ull) * 0x9E3779B97F4A7C15ull);
    }
    // end of the chunk: inside a big panel all 256 threads hold partial sums of ONE supernode -- one atomic per workgroup instead of
    // 256 on the same word (the root panel alone would otherwise take 59 M serialised atomics); mixed workgroups add per thread
    __shared__ int s_first;
    __shared__ unsigned long long s_part[4];
    if (threadIdx.x == 0) s_first = s;
    __syncthreads();
    const bool uniform = __syncthreads_and(s == s_first) != 0;
    if (!uniform) { if (acc) atomicAdd(&H[s], acc); return; }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long tot = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (tot) atomicAdd(&H[s], tot);
    }
}

void launch_factor_hash(const int32_t* Super, const int64_t* Lsip, const int64_t* Xp, const int64_t* RefXp, int32_t nsuper,
                        const double* PL, const double* PU, int lu, int64_t total, unsigned long long* H, hipStream_t st) {
    if (total <= 0 || nsuper <= 0) return;
    const int64_t chunk = 256 * 64;
    const int64_t blocks = (total + chunk - 1) / chunk;
    hipLaunchKernelGGL(k_factor_hash, dim3((unsigned)blocks), dim3(256), 0, st, Super, Lsip, Xp, RefXp, nsuper, PL, PU, lu, total, chunk, H);
}

__global__ void __launch_bounds__(256)
k_lu_fill_u11(const FillTile* __restrict__ tiles, double* __restrict__ PL, const double* __restrict__ PU) {
    __shared__ double tile[64][65];
    const FillTile t = tiles[blockIdx.x];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int jlo = max(t.c0, t.cb), jhi = min(t.c0 + 64, t.ce);
    {
        const int j = t.c0 + tx;                    // consecutive lanes: consecutive rows of the U^T panel
        for (int rr = ty; rr < 64; rr += 4) {
            const int R = t.r0 + rr;
            if (j >= jlo && j < jhi && R <= j) tile[rr][tx] = PU[t.xp + j + (int64_t)R * t.nsrow];
        }
    }
    __syncthreads();
    {
        const int R = t.r0 + tx;                    // consecutive lanes: consecutive rows of the L panel
        for (int jj = ty; jj < 64; jj += 4) {
            const int j = t.c0 + jj;
            if (j >= jlo && j < jhi && R <= j) PL[t.xp + R + (int64_t)j * t.nsrow] = tile[tx][jj];
        }
    }
}

void launch_lu_fill_u11(const FillTile* tiles, int64_t ntiles, double* PL, const double* PU, hipStream_t st) {
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_lu_fill_u11, dim3((unsigned)ntiles), dim3(256), 0, st, tiles, PL, PU);
}

// ---------------------------------------------------------------------------------------------------
// X <- X * D^{-T} for a tile of rows, D = lower-triangular b x b block already factored.
// One row per lane (rows are contiguous in memory: coalesced 8-byte accesses per column).
// The row is solved 8 columns at a time: the 8 running sums live in registers, the contributions of
// the columns already solved are re-read from the panel (the lane's own earlier stores, L2-resident)
// and D is broadcast from LDS, transposed so that the 8 multipliers of one k are 64 contiguous bytes.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TRSM_ROWS)
k_trsm_block(const TrsmTask* __restrict__ tasks, double* __restrict__ Lsx, const int32_t* __restrict__ pivinv) {
    __shared__ __attribute__((aligned(16))) double Dt[NB][NB];   // Dt[k][j] = L(j,k) for k < j, else 0
    __shared__ double Dinv[NB];
    const TrsmTask t = tasks[blockIdx.x];
    const double* Dg = Lsx + t.dpanel + t.diag + (int64_t)t.diag * t.ld;
    const int b = t.b;
    const int tid = threadIdx.x;

    // (columns k >= b are never multiplied with: a 3-column panel of the bottom levels fills 3 x 64 entries, not 64 x 64 -- config 3 has
    //  26,000 such panels; the entries Dt[k][j >= b] only reach sums that are not stored)
    for (int e = tid; e < b * NB; e += (int)blockDim.x) {
        const int k = e / NB, j = e % NB;   // column k, row j of the block: coalesced along j
        Dt[k][j] = (j < b && k < j) ? Dg[j + (int64_t)k * t.ld] : 0.0;
    }
    if (tid < NB) Dinv[tid] = (tid < b && !t.unit) ? 1.0 / Dg[tid + (int64_t)tid * t.ld] : 1.0;
    __syncthreads();

    if (tid >= t.nrows) return;
    double* X = Lsx + t.panel + t.row0 + tid + (int64_t)t.diag * t.ld;
    const int64_t ld = t.ld;
    if (pivinv && t.unit) {
        // LU with pivoting: these are rows of U^T, i.e. the tile's columns are the block's rows of U -- bring them into
        // pivot order first (column at position p <- original column pivinv[p]); all loads precede the stores
        const int g0 = t.first_col + t.diag;
        const int32_t* pv = pivinv + g0;
        double tmp[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) tmp[c] = X[(int64_t)(pv[min(c, b - 1)] - g0) * ld];
#pragma unroll
        for (int c = 0; c < NB; ++c)
            if (c < b) X[(int64_t)c * ld] = tmp[c];
    }

    for (int jb = 0; jb < b; jb += 8) {
        double acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = X[min(jb + u, b - 1) * ld];      // unconditional, clamped: 8 loads in flight
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = (jb + u < b) ? acc[u] : 0.0;
        for (int k0 = 0; k0 < jb; k0 += 8) {       // jb is a multiple of 8: 8 independent re-reads in flight per round trip
            double xk[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) xk[v] = X[(k0 + v) * ld];
#pragma unroll
            for (int v = 0; v < 8; ++v)
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] -= xk[v] * Dt[k0 + v][jb + u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int v = 0; v < u; ++v) acc[u] -= acc[v] * Dt[jb + v][jb + u];
            acc[u] *= Dinv[jb + u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (jb + u < b) X[(jb + u) * ld] = acc[u];
    }
}

// (64-thread workgroups for launches whose tiles have at most 64 rows -- the bottom levels -- were measured: SLOWER, 0.47 against
// 0.395 ms on config 3: the block's fill is what the other three waves are for)
void launch_trsm(const TrsmTask* tasks, int ntasks, double* Lsx, const int32_t* pivinv, hipStream_t st) {
    if (ntasks <= 0) return;
    hipLaunchKernelGGL(k_trsm_block, dim3(ntasks), dim3(TRSM_ROWS), 0, st, tasks, Lsx, pivinv);
}

// ---------------------------------------------------------------------------------------------------
// Triangular solves with the resident factor (L L^T x = b or L U x = b, permuted space; reference: scalar host loops,
// C:3074-3134, L:3592-3700).  The sweep is a chain of dependent steps, so the step is made BIG and its inside cheap:
// one launch per (level, 256-column step) and direction (SV_B = 256),
//   forward : x_blk <- D^{-1} x_blk  (256 x 256 lower-triangular block)  ;  x[rows below] -= L[rows, blk] x_blk
//   backward: x_blk -= L[rows below, blk]^T x[rows below]               ;  x_blk <- D^{-T} x_blk
// and both halves hand over INSIDE the launch (tasks claimed by ticket in execution order, producers first in the list).
// Diagonal task = one workgroup, wave w owns the 64-column sub-block w: its 64 x 64 triangle sits in the lane's registers
// from the start (all four waves load at once), the off-diagonal 64 x 64 blocks are prefetched one sub-step ahead, the
// solved sub-vector goes round through LDS: 4 substitution chains of 64 and 3 barriers instead of 4 launches with 4
// device-scope hand-offs.  No division on the chains (lane j forms 1 / D(j,j) up front).
// Row tiles = 64 rows x the step's columns: thread (lane, wave) = (row, 64-column chunk) forward, (column, chunk)
// backward, ALL its 64 matrix entries are in flight before the hand-off, after it 64 FMAs and one atomic.
// ---------------------------------------------------------------------------------------------------
constexpr int SV_SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ void sv_publish(int* flag, int value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void sv_wait(const int* flag, int value, int* info) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != value) {
        __builtin_amdgcn_s_sleep(16);       // hundreds of waiting workgroups poll ONE address: keep the L2 channel usable for its writer
        if (++spins > SV_SPIN_LIMIT) { atomicOr(info, 2); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool BIG>
__global__ void __launch_bounds__(256, BIG ? 1 : 2)
k_solve_fwd(const SolveTask* __restrict__ tasks, const double* __restrict__ Lsx, const int32_t* __restrict__ Lsi,
            double* __restrict__ x, int unit, const int32_t* __restrict__ pivpos, int* __restrict__ sync, int* __restrict__ ticket,
            int* __restrict__ info) {
    __shared__ int s_ticket;
    __shared__ double xs[SV_B];
    __shared__ double part[4][NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const SolveTask t = tasks[__builtin_amdgcn_readfirstlane(s_ticket)];
    const int b = t.b;                  // <= SV_B columns in this step
    const int64_t ld = t.ld;
    const int o = NB * wave;
    const int bw = min(NB, max(0, b - o));      // this wave's part of the step's columns
    if (t.nrows == 0) {                 // ---- diagonal task ----
        const double* P = Lsx + t.panel;
        double a[NB];       // row `lane` of the sub-block's triangle; unit: the diagonal is implied (LU: the L panel)
        if (bw > 0) {
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                // unconditional loads from clamped addresses, then select (a load under a per-element condition becomes a
                // branch plus its own s_waitcnt: 64 dependent round trips)
                const double v = P[(t.diag + o + min(lane, bw - 1)) + (int64_t)(t.diag + o + min(c, bw - 1)) * ld];
                a[c] = (lane < bw && c + unit <= lane) ? v : ((c == lane) ? 1.0 : 0.0);
            }
        } else {
#pragma unroll
            for (int c = 0; c < NB; ++c) a[c] = (c == lane) ? 1.0 : 0.0;
        }
        double* xq = x + t.first_col + t.diag + o;
        double v = (lane < bw) ? xq[lane] : 0.0;
        double dinv = 1.0;
#pragma unroll
        for (int c = 0; c < NB; ++c) dinv = (c == lane) ? 1.0 / a[c] : dinv;
        const int nsub = (b + NB - 1) / NB;
        for (int tt = 0; tt < nsub; ++tt) {
            const bool below = BIG && wave > tt && bw > 0;
            double blk[BIG ? NB : 1];       // L(this wave's row, columns of sub-block tt): in flight while wave tt solves
            if (BIG && below) {
#pragma unroll
                for (int k = 0; k < NB; ++k) blk[k] = P[(t.diag + o + min(lane, bw - 1)) + (int64_t)(t.diag + NB * tt + k) * ld];
            }
            if (wave == tt) {
                if (pivpos) {
                    // LU with pivoting: the row interchanges of this 64-column block, applied as the sweep reaches it
                    // (LINPACK-style: the L entries to the left of a block were stored at their rows' original places)
                    const int g0 = t.first_col + t.diag + o;
                    if (lane < bw) part[0][pivpos[g0 + lane] - g0] = v;
                    v = (lane < bw) ? part[0][lane] : 0.0;       // one wave: LDS operations complete in order
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const double xj = readlane_f64(v, j) * readlane_f64(dinv, j);
                    if (lane == j) v = xj;
                    if (lane > j) v -= a[j] * xj;
                }
                xs[o + lane] = (lane < bw) ? v : 0.0;
            }
            if (BIG) {
                __syncthreads();
                if (below) {
#pragma unroll
                    for (int k = 0; k < NB; ++k) v -= blk[k] * xs[NB * tt + k];
                }
            }
        }
        if (lane < bw) xq[lane] = v;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) sv_publish(sync + t.flag, 1);
        return;
    }
    // ---- row tile: lane = row, wave = 64-column chunk; the 64 entries are in flight while the diagonal block is solved ----
    // (a "far" tile of a look-ahead step may hold several 64-row groups, t.nrows > 64: one workgroup streams through them -- one
    // ticket, one task, one flag poll, one load of x_blk for all of them)
    int nr = min(t.nrows, SV_ROWS);
    int r = t.row0 + min(lane, nr - 1);
    double lr[NB];
    if (bw > 0) {
        const double* Lr = Lsx + t.panel + r + (int64_t)(t.diag + o) * ld;
#pragma unroll
        for (int k = 0; k < NB; ++k) lr[k] = Lr[(int64_t)min(k, bw - 1) * ld];
    }
    int32_t gi = Lsi[t.rows + r];
    if (tid == 0) sv_wait(sync + t.flag, 1, info);
    __syncthreads();
    xs[tid] = (tid < b) ? __builtin_nontemporal_load(x + t.first_col + t.diag + tid) : 0.0;
    __syncthreads();
    {
        double acc = 0.0;
        if (bw > 0) {
#pragma unroll
            for (int k = 0; k < NB; ++k) acc += lr[k] * xs[o + k];       // columns beyond b meet xs = 0
        }
        part[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && lane < nr) unsafeAtomicAdd(x + gi, -(part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]));
    }
    if (BIG) {          // (only this instantiation walks through further row groups; the plan makes sure of it)
#pragma unroll 1
        for (int g0 = SV_ROWS; g0 < t.nrows; g0 += SV_ROWS) {
            nr = min(t.nrows - g0, SV_ROWS);
            r = t.row0 + g0 + min(lane, nr - 1);
            if (bw > 0) {
                const double* Lr = Lsx + t.panel + r + (int64_t)(t.diag + o) * ld;
#pragma unroll
                for (int k = 0; k < NB; ++k) lr[k] = Lr[(int64_t)min(k, bw - 1) * ld];
            }
            gi = Lsi[t.rows + r];
            double acc = 0.0;
            if (bw > 0) {
#pragma unroll
                for (int k = 0; k < NB; ++k) acc += lr[k] * xs[o + k];
            }
            __syncthreads();            // part[] of the previous group has been read
            part[wave][lane] = acc;
            __syncthreads();
            if (wave == 0 && lane < nr) unsafeAtomicAdd(x + gi, -(part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]));
        }
    }
}

template <bool BIG>
__global__ void __launch_bounds__(256, BIG ? 1 : 2)
k_solve_bwd(const SolveTask* __restrict__ tasks, const double* __restrict__ Lsx, const int32_t* __restrict__ Lsi,
            double* __restrict__ x, int* __restrict__ sync, int* __restrict__ ticket, int* __restrict__ info,
            const double* __restrict__ Tbase) {
    __shared__ int s_ticket;
    __shared__ double xs[SV_B];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const SolveTask t = tasks[__builtin_amdgcn_readfirstlane(s_ticket)];
    const int b = t.b;
    const int64_t ld = t.ld;
    const int o = NB * wave;
    const int bw = min(NB, max(0, b - o));
    if (t.nrows > 0) {
        // ---- row tile: lane = row (coalesced loads), wave = 64-column chunk.  p[k] = L(row, column k) x_row has to be summed
        // over the 64 lanes for every k: a transposing butterfly -- in the step with mask m a lane keeps the half of its
        // array that matches its bit m and adds the partner's other half -- leaves lane l with the sum of ONE column after
        // 63 exchanges instead of 64 full reductions.
        // (a "far" tile of a look-ahead step may hold several 64-row groups, t.nrows > 64: their products are summed in
        // registers first -- the butterfly is linear -- so the group of tiles costs ONE butterfly and ONE set of atomics on the
        // 256 words every tile of the step adds to)
        double p[NB];
        if (bw > 0) {
            int nr = min(t.nrows, SV_ROWS);
            int rr = t.row0 + min(lane, nr - 1);
            {
                const double* Lr = Lsx + t.panel + rr + (int64_t)(t.diag + o) * ld;
#pragma unroll
                for (int k = 0; k < NB; ++k) p[k] = Lr[(int64_t)min(k, bw - 1) * ld];
                const double xr = (lane < nr) ? x[Lsi[t.rows + rr]] : 0.0;
#pragma unroll
                for (int k = 0; k < NB; ++k) p[k] *= xr;
            }
            if (BIG) {
#pragma unroll 1
                for (int g0 = SV_ROWS; g0 < t.nrows; g0 += SV_ROWS) {
                    nr = min(t.nrows - g0, SV_ROWS);
                    rr = t.row0 + g0 + min(lane, nr - 1);
                    const double* Lr = Lsx + t.panel + rr + (int64_t)(t.diag + o) * ld;
                    double q[NB];
#pragma unroll
                    for (int k = 0; k < NB; ++k) q[k] = Lr[(int64_t)min(k, bw - 1) * ld];
                    const double xr = (lane < nr) ? x[Lsi[t.rows + rr]] : 0.0;
#pragma unroll
                    for (int k = 0; k < NB; ++k) p[k] += q[k] * xr;
                }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const bool up = (lane & m) != 0;
#pragma unroll
                for (int i = 0; i < m; ++i) {
                    const double keep = up ? p[i + m] : p[i];
                    const double give = up ? p[i] : p[i + m];
                    p[i] = keep + __shfl_xor(give, m, 64);
                }
            }
            // lane l now holds the column whose index has bit m set exactly where l has it: column l
            if (lane < bw) unsafeAtomicAdd(x + t.first_col + t.diag + o + lane, -p[0]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(sync + t.flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    // ---- diagonal task: x_blk <- D^{-T} x_blk, sub-blocks from the last to the first; lane = column ----
    const double* P = Lsx + t.panel;
    double bcol[NB];    // bcol[c] = D(c, lane): column `lane` of the sub-block's triangle, rows c >= lane (one contiguous run per
                        // lane: 64 cache lines per load instruction, ~7 us per block -- measured cheaper than coalesced row loads
                        // plus an in-wave transpose through LDS, which made the backward sweep 23 -> 36 ms)
    // (steps of the top levels come with a ROW-major copy of their diagonal block, t.tdiag, made at the start of the solve:
    // D(c, lane) is then one contiguous run across the lanes, i.e. coalesced, for bcol and for blk below)
    const double* __restrict__ Td = (Tbase && t.tdiag) ? Tbase + (t.tdiag - 1) : nullptr;
    if (bw > 0) {
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const double v = Td ? Td[(int64_t)(o + min(c, bw - 1)) * b + (o + min(lane, bw - 1))]
                                : P[(t.diag + o + min(c, bw - 1)) + (int64_t)(t.diag + o + min(lane, bw - 1)) * ld];
            bcol[c] = (lane < bw && c < bw && c >= lane) ? v : ((c == lane) ? 1.0 : 0.0);
        }
    } else {
#pragma unroll
        for (int c = 0; c < NB; ++c) bcol[c] = (c == lane) ? 1.0 : 0.0;
    }
    double dinv = 1.0;
#pragma unroll
    for (int c = 0; c < NB; ++c) dinv = (c == lane) ? 1.0 / bcol[c] : dinv;
    if (t.expect > 0) {
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(sync + t.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != t.expect) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > SV_SPIN_LIMIT) { atomicOr(info, 2); break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    double* xq = x + t.first_col + t.diag + o;
    double v = (lane < bw) ? __builtin_nontemporal_load(xq + min(lane, max(bw, 1) - 1)) : 0.0;
    const int nsub = (b + NB - 1) / NB;
    for (int tt = nsub - 1; tt >= 0; --tt) {
        const bool above = BIG && wave < tt && bw > 0;
        const int bt = min(NB, b - NB * tt);            // rows of sub-block tt
        double blk[BIG ? NB : 1];                       // L(rows of sub-block tt, this lane's column): one contiguous run
        if (BIG && above) {
#pragma unroll
            for (int k = 0; k < NB; ++k)
                blk[k] = Td ? Td[(int64_t)(NB * tt + min(k, bt - 1)) * b + (o + min(lane, bw - 1))]
                            : P[(t.diag + NB * tt + min(k, bt - 1)) + (int64_t)(t.diag + o + min(lane, bw - 1)) * ld];
        }
        if (wave == tt) {
#pragma unroll
            for (int j = NB - 1; j >= 0; --j) {
                const double xj = readlane_f64(v, j) * readlane_f64(dinv, j);
                if (lane == j) v = xj;
                if (lane < j) v -= bcol[j] * xj;           // D(j, lane) * x_j
            }
            xs[o + lane] = (lane < bw) ? v : 0.0;
        }
        if (BIG) {
            __syncthreads();
            if (above) {
#pragma unroll
                for (int k = 0; k < NB; ++k) v -= blk[k] * xs[NB * tt + k];       // rows beyond bt meet xs = 0
            }
        }
    }
    if (lane < bw) xq[lane] = v;
}

// ---------------------------------------------------------------------------------------------------
// Steps in which every panel is narrow (nscol <= 64: the swarm levels, tens of thousands of supernodes of a few dozen
// columns): ONE WAVE per supernode does its whole part of the sweep -- diagonal solve and all its rows -- with no hand-off,
// four supernodes per workgroup.  (Through the general kernels such a supernode costs a diagonal workgroup plus one workgroup
// per 64 rows, three of four waves idle in each, and a device-scope hand-off.)  task.ld = nsrow, task.b = nscol.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
k_solve_small_fwd(const SolveTask* __restrict__ tasks, int ntasks, const double* __restrict__ Lsx, const int32_t* __restrict__ Lsi,
                  double* __restrict__ x, int unit, const int32_t* __restrict__ pivpos) {
    __shared__ double ptmp[4][NB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the task index is uniform over the wave: say so, or every field of the task (and all address arithmetic) lives in VGPRs
    const int ti = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (ti >= ntasks) return;
    const SolveTask t = tasks[ti];
    const int b = t.b;
    const int64_t ld = t.ld;
    const double* P = Lsx + t.panel;
    double a[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        const double v = P[min(lane, b - 1) + (int64_t)min(c, b - 1) * ld];
        a[c] = (lane < b && c + unit <= lane) ? v : ((c == lane) ? 1.0 : 0.0);
    }
    double* xq = x + t.first_col;
    double v = (lane < b) ? xq[lane] : 0.0;
    if (pivpos) {
        if (lane < b) ptmp[wave][pivpos[t.first_col + lane] - t.first_col] = v;
        v = (lane < b) ? ptmp[wave][lane] : 0.0;
    }
    double dinv = 1.0;
#pragma unroll
    for (int c = 0; c < NB; ++c) dinv = (c == lane) ? 1.0 / a[c] : dinv;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double xj = readlane_f64(v, j) * readlane_f64(dinv, j);
        if (lane == j) v = xj;
        if (lane > j) v -= a[j] * xj;
    }
    if (lane < b) xq[lane] = v;
    // the rows below: 64 at a time, lane = row; x_blk[k] is broadcast out of lane k's register
    for (int r0 = b; r0 < (int)ld; r0 += NB) {
        const int row = min(r0 + lane, (int)ld - 1);
#pragma unroll
        for (int k = 0; k < NB; ++k) a[k] = P[row + (int64_t)min(k, b - 1) * ld];
        const int32_t gi = Lsi[t.rows + row];
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NB; ++k) acc += a[k] * readlane_f64(v, k);       // lanes >= b hold v = 0
        if (r0 + lane < (int)ld) unsafeAtomicAdd(x + gi, -acc);
    }
}

__global__ void __launch_bounds__(256, 2)
k_solve_small_bwd(const SolveTask* __restrict__ tasks, int ntasks, const double* __restrict__ Lsx, const int32_t* __restrict__ Lsi,
                  double* __restrict__ x) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ti = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (ti >= ntasks) return;
    const SolveTask t = tasks[ti];
    const int b = t.b;
    const int64_t ld = t.ld;
    const double* P = Lsx + t.panel;
    // s_c = sum over the rows below of L(row, c) x[row]: lane = row (coalesced), then the transposing butterfly
    double s = 0.0;
    double p[NB];
    for (int r0 = b; r0 < (int)ld; r0 += NB) {
        const int row = min(r0 + lane, (int)ld - 1);
#pragma unroll
        for (int k = 0; k < NB; ++k) p[k] = P[row + (int64_t)min(k, b - 1) * ld];
        const double xr = (r0 + lane < (int)ld) ? x[Lsi[t.rows + row]] : 0.0;
#pragma unroll
        for (int k = 0; k < NB; ++k) p[k] *= xr;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const bool up = (lane & m) != 0;
#pragma unroll
            for (int i = 0; i < m; ++i) {
                const double keep = up ? p[i + m] : p[i];
                const double give = up ? p[i] : p[i + m];
                p[i] = keep + __shfl_xor(give, m, 64);
            }
        }
        s += p[0];
    }
    // D^T x_blk = x_blk - s, lane = column
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        const double v = P[min(c, b - 1) + (int64_t)min(lane, b - 1) * ld];
        p[c] = (lane < b && c < b && c >= lane) ? v : ((c == lane) ? 1.0 : 0.0);
    }
    double dinv = 1.0;
#pragma unroll
    for (int c = 0; c < NB; ++c) dinv = (c == lane) ? 1.0 / p[c] : dinv;
    double* xq = x + t.first_col;
    double v = (lane < b) ? xq[lane] - s : 0.0;
#pragma unroll
    for (int j = NB - 1; j >= 0; --j) {
        const double xj = readlane_f64(v, j) * readlane_f64(dinv, j);
        if (lane == j) v = xj;
        if (lane < j) v -= p[j] * xj;
    }
    if (lane < b) xq[lane] = v;
}

void launch_solve_small_fwd(const SolveTask* t, int nt, const double* Lsx, const int32_t* Lsi, double* x, int unit, const int32_t* pivpos,
                            hipStream_t st) {
    if (nt > 0) hipLaunchKernelGGL(k_solve_small_fwd, dim3((nt + 3) / 4), dim3(256), 0, st, t, nt, Lsx, Lsi, x, unit, pivpos);
}
void launch_solve_small_bwd(const SolveTask* t, int nt, const double* Lsx, const int32_t* Lsi, double* x, hipStream_t st) {
    if (nt > 0) hipLaunchKernelGGL(k_solve_small_bwd, dim3((nt + 3) / 4), dim3(256), 0, st, t, nt, Lsx, Lsi, x);
}

void launch_solve_fwd(const SolveTask* t, int nt, int big, const double* Lsx, const int32_t* Lsi, double* x, int unit, const int32_t* pivpos,
                      int* sync, int* ticket, int* info, hipStream_t st) {
    if (nt <= 0) return;
    if (big) hipLaunchKernelGGL(k_solve_fwd<true>, dim3(nt), dim3(256), 0, st, t, Lsx, Lsi, x, unit, pivpos, sync, ticket, info);
    else hipLaunchKernelGGL(k_solve_fwd<false>, dim3(nt), dim3(256), 0, st, t, Lsx, Lsi, x, unit, pivpos, sync, ticket, info);
}
void launch_solve_bwd(const SolveTask* t, int nt, int big, const double* Lsx, const int32_t* Lsi, double* x, int* sync, int* ticket, int* info,
                      hipStream_t st, const double* Tbase) {
    if (nt <= 0) return;
    if (big) hipLaunchKernelGGL(k_solve_bwd<true>, dim3(nt), dim3(256), 0, st, t, Lsx, Lsi, x, sync, ticket, info, Tbase);
    else hipLaunchKernelGGL(k_solve_bwd<false>, dim3(nt), dim3(256), 0, st, t, Lsx, Lsi, x, sync, ticket, info, Tbase);
}

// T(r, c) = D(r, c), row-major b x b, for the lower triangle's 64 x 64 tiles of a step's diagonal block (one workgroup per tile,
// transposed through LDS: reads run down the panel's columns, writes along the copy's rows)
__global__ void __launch_bounds__(256)
k_solve_transpose_diag(const SolveTask* __restrict__ tasks, const int64_t* __restrict__ list, const double* __restrict__ Lsx,
                       double* __restrict__ T) {
    __shared__ double tile[64][65];
    const SolveTask t = tasks[list[blockIdx.x >> 4]];
    const int ti = (blockIdx.x & 15) >> 2, tj = blockIdx.x & 3, b = t.b;
    if (tj > ti || 64 * ti >= b || 64 * tj >= b || !t.tdiag) return;
    const double* __restrict__ P = Lsx + t.panel;
    double* __restrict__ Td = T + (t.tdiag - 1);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int cc = ty; cc < 64; cc += 4) {
        const int r = 64 * ti + tx, c = 64 * tj + cc;
        if (r < b && c < b) tile[cc][tx] = P[(t.diag + r) + (int64_t)(t.diag + c) * t.ld];
    }
    __syncthreads();
    for (int rr = ty; rr < 64; rr += 4) {
        const int r = 64 * ti + rr, c = 64 * tj + tx;
        if (r < b && c < b) Td[(int64_t)r * b + c] = tile[tx][rr];
    }
}

void launch_solve_transpose_diag(const SolveTask* tasks, const int64_t* list, int64_t ntasks, const double* Lsx, double* T, hipStream_t st) {
    if (ntasks <= 0) return;
    hipLaunchKernelGGL(k_solve_transpose_diag, dim3((unsigned)(ntasks * 16)), dim3(256), 0, st, tasks, list, Lsx, T);
}

// ---------------------------------------------------------------------------------------------------
// fp64 MFMA GEMM  C[ci][cj] -= sum_k Y[ci][k] X[cj][k]   (lower trapezoid ci >= cj)
//
// Workgroup = 4 waves (2 x 2), tile 128 (ci) x 128 (cj), K step 16, double-buffered LDS.
// Each wave owns a 64 x 64 sub-tile = 4 x 4 MFMA tiles of v_mfma_f64_16x16x4_f64:
//     A operand (row index of D)  <- X rows (cj)      lane l: X[cj = l&15][k = l>>4]
//     B operand (col index of D)  <- Y rows (ci)      lane l: Y[ci = l&15][k = l>>4]
//     D[row = (l>>4) + 4*reg][col = l&15]             (f64 layout: NOT the f32 one)
// so that the 16 consecutive lanes of a quarter-wave hold 16 consecutive TARGET ROWS (ci), which are
// contiguous in the column-major target panel: the epilogue's atomics / stores hit 128-byte runs.
// LDS image of an operand tile: [k][row] with the row stride padded to 144 doubles, which makes the
// ds_read_b64 fragment reads (16 rows x 2 k per half-wave) hit all 64 banks exactly once.
// ---------------------------------------------------------------------------------------------------
constexpr int LDS_LD = GEMM_BM + 16;

// largest i in [0, n) with a[i] <= key   (a ascending, a[0] = 0 <= key)
__device__ __forceinline__ int last_le_u32(const uint32_t* __restrict__ a, int n, uint32_t key) {
    int lo = 0, hi = n;     // invariant: a[lo] <= key, (hi == n or a[hi] > key)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= key) lo = mid; else hi = mid;
    }
    return lo;
}

// Persistent launch: the work of one launch is the list of (tile, 16-deep K step) units of all its tiles, in
// task order; kt_prefix[i] = number of units before tile i.  The grid is a fixed number of workgroups (2 per CU).
// Whole tiles are dealt out in rounds (workgroup `share` takes tile r G + share in round r -- see the kernel body for
// why that matters for the L2); what does not fill a round is split "stream-K" style: each workgroup takes one
// contiguous, equal share of the remaining units, so the chip stays full whatever the mix of tile counts and K
// lengths (no tail of half-empty rounds).  Every tile is combined into its target with fp64 atomics (a K range
// split between workgroups needs them anyway).
// DMA: the operand tiles go global -> LDS directly (global_load_lds_dwordx4, "LDS-DMA"): the [k][row] LDS image of one k is
// 128 doubles = 1 KiB = exactly what one wave instruction writes (lane l: rows 2l, 2l+1), and it is 1 KiB of one panel column in
// memory too.  No staging registers, no ds_write, no select instructions; a k beyond K reads the X operand from a page of
// zeros (a DMA cannot mask), rows beyond M / N read valid memory whose products only reach entries that are never stored.
__device__ double g_zero_page[GEMM_BM + 16];

// One LDS-DMA of 16 bytes per lane (1 KiB per wave) as inline asm: the builtin form makes hipcc wait vmcnt(0) before the NEXT
// ds_read (it cannot tell the DMA's LDS destination from the buffer being read), which exposes the whole memory latency once per
// K step; an asm statement is outside its s_waitcnt bookkeeping, so the DMA stays in flight until the explicit
// `s_waitcnt vmcnt(0)` in front of the K step's barrier.  (Untracked operations can only make hipcc's own counted waits
// stricter: vmcnt retires in order.)  M0 = the wave-uniform LDS byte address, saved and restored around the instruction.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p);
}

#ifdef SF_EXP_TIMING            // tools/experiments/gemm_overhead.sh: per-tile time stamps of every workgroup (s_memtime)
__device__ unsigned long long* g_exp_stamps = nullptr;      // [workgroup][64 tiles][4]
void exp_set_stamps(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_exp_stamps), &p, sizeof(p)); }
#define SF_STAMP(slot) do { if (tid == 0 && g_exp_stamps && exp_tile < 64) \
        g_exp_stamps[((size_t)blockIdx.x * 64 + exp_tile) * 4 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SF_STAMP(slot) do { } while (0)
#endif

template <int MODE, bool DMA>
__global__ void __launch_bounds__(GEMM_THREADS, GEMM_WAVES / 2)
k_gemm(const GemmProb* __restrict__ probs, const GemmTask* __restrict__ tasks,
       const uint32_t* __restrict__ kt_prefix, int ntasks, uint32_t u_lo, uint32_t u_hi,
       double* __restrict__ Lsx, const int32_t* __restrict__ RelMap, int* __restrict__ ticket, int whole_tiles, uint32_t min_units) {
    __shared__ int s_claim;
    __shared__ __attribute__((aligned(16))) double Ys[2][GEMM_BK][LDS_LD];
    __shared__ __attribute__((aligned(16))) double Xs[2][GEMM_BK][LDS_LD];
    __shared__ int32_t rowmap[GEMM_BM];
    __shared__ int32_t colmap[GEMM_BN];

    constexpr int WCJ = GEMM_BN / (GEMM_WAVES / 2);     // columns (cj) per wave: 64 with 4 waves, 32 with 8
    constexpr int TMN = WCJ / 16;                         // MFMA tiles per wave along cj
    constexpr int SQ = GEMM_BK / GEMM_WAVES;              // staging passes per K step
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;            // wave (wm, wn) owns rows [64 wm, +64) x columns [WCJ wn, +WCJ)
    const int fr = lane & 15, fk = lane >> 4;

    // XCD-aware share: workgroups b, b+8, b+16 ... run on one XCD (one L2); give each XCD a contiguous
    // run of shares so that the tiles it works on at any time are neighbours (supertile order).
    // [u_lo, u_hi): the units of this launch that this device executes (the whole launch on one GPU; one rank's
    // share when the launch is split over the ranks of a sharded factorization -- the update is a sum, any split is valid)
    const uint32_t T = u_hi - u_lo;
    const uint32_t G = gridDim.x;
    const uint32_t xcd = blockIdx.x & 7;
    const uint32_t xcd_n = (G >> 3) + (xcd < (G & 7) ? 1u : 0u);                              // workgroups of this XCD
    const uint32_t xcd_base = xcd < (G & 7) ? xcd * ((G >> 3) + 1) : (G & 7) * ((G >> 3) + 1) + (xcd - (G & 7)) * (G >> 3);
    const uint32_t share = xcd_base + (blockIdx.x >> 3);
    // Whole tiles are dealt out in ROUNDS: in round r workgroup `share` takes tile t0 + r G + share, so the 64 workgroups
    // of an XCD (consecutive shares) work on 64 CONSECUTIVE tiles -- one supertile -- at the same time and march through K
    // together: each operand slice is fetched into that XCD's L2 once per round and re-used by the 8 tiles of its
    // row / column.  (Contiguous shares of ~16 tiles each, the plain stream-K split, put concurrently running
    // workgroups 16 tiles apart: no operand was ever shared, PMC FETCH_SIZE = the no-reuse byte count.)
    // What does not fill a round -- the partial tiles at the ends of a rank's unit window and the last < G tiles -- is
    // split by (tile, K step) units as before, so the tail is still balanced.
    (void)T;
    int t0 = last_le_u32(kt_prefix, ntasks + 1, u_lo);
    if (kt_prefix[t0] < u_lo) ++t0;
    const int t1 = last_le_u32(kt_prefix, ntasks + 1, u_hi);          // tiles [t0, t1) lie inside [u_lo, u_hi)
    // whole_tiles: every tile is multiplied over its full K range by ONE workgroup (the last round is partial, nothing is split by
    // units), so each target element receives exactly one addition from this launch and the result does not depend on the order
    // workgroups run in -- what a launch that several ranks execute redundantly needs (the ranks' copies must stay bit-identical)
    const int R = (t1 > t0) ? (int)(((uint32_t)(t1 - t0) + (whole_tiles ? G - 1 : 0)) / G) : 0;
    const uint32_t head_end = (R > 0) ? kt_prefix[t0] : u_hi;
    const uint32_t tail_beg = (R > 0 && !whole_tiles) ? kt_prefix[t0 + R * (int)G] : u_hi;

    // ticket != nullptr: the rounds are DYNAMIC -- the workgroups of an XCD claim the tiles of that XCD's slots (the same
    // tiles as in the static deal, so a supertile still shares one L2) from the XCD's counter in the order they get free; tiles of
    // different K (different source supernodes in one launch) no longer leave a workgroup idle while its neighbour works off a
    // round of long ones.  Head and tail are split statically as before.
    int ph = 0, rr = 0;
#ifdef SF_EXP_DEPHASE
    bool exp_dephased = false;
#endif
#ifdef SF_EXP_TIMING
    int exp_tile = 0;
    // per workgroup, behind the tile stamps: shader-clock and constant 100 MHz stamps at its first and last instruction
    if (tid == 0 && g_exp_stamps) {
        g_exp_stamps[(size_t)gridDim.x * 256 + blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
        g_exp_stamps[(size_t)gridDim.x * 256 + blockIdx.x * 4 + 1] = wall_clock64();
    }
#endif
    for (;;) {
    uint32_t u, u_end;
    int ti;
    SF_STAMP(0);
    if (ph == 1) {
        uint32_t slot = share;
        if (ticket != nullptr && R > 0) {
            if (tid == 0) s_claim = __hip_atomic_fetch_add(ticket + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();                     // the barrier that ends a tile separates this read from the next claim
            const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane(s_claim);
            rr = (int)(c / xcd_n);
            slot = xcd_base + c % xcd_n;
        }
        if (rr >= R) { ph = 2; continue; }
        ti = t0 + rr * (int)G + (int)slot;
        ++rr;
        if (ti >= t1) continue;                 // (whole_tiles: the last round is partial)
        u = kt_prefix[ti];
        u_end = kt_prefix[ti + 1];
#ifdef SF_EXP_DEPHASE           // experiment (round 4): the second workgroup of every CU (blockIdx >= grid / 2: the dispatcher fills the CUs
        // round-robin) starts its first whole tile half a K loop late, so that the two co-resident workgroups do not sit in their
        // prologues and atomic epilogues -- no MFMA work -- at the same time for the rest of the launch
        if (!exp_dephased && blockIdx.x >= gridDim.x / 2) {
            const int naps = (int)(u_end - u) * SF_EXP_DEPHASE / 16;       // one s_sleep 127 = 8128 cycles; a 16-deep K step of two workgroups ~ 8.5k
            for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
        }
        exp_dephased = true;
#endif
    } else {
        const uint32_t ra = (ph == 0) ? u_lo : tail_beg, rb = (ph == 0) ? head_end : u_hi;
        // (a share of fewer than min_units K steps costs more in its epilogue -- a whole 128 x 128 tile of atomics, on elements that up
        //  to G / tiles other workgroups add to at the same time -- than it saves: small launches run on fewer workgroups instead)
        const uint32_t Ueq = (rb > ra) ? (rb - ra + G - 1) / G : 0, U = max(Ueq, min_units);
        // (when the floor applies only some workgroups get a share: take them round-robin over the XCDs -- blockIdx -- instead of
        //  XCD by XCD, or the first XCDs would do all the work)
        const uint32_t sh = (U > Ueq) ? blockIdx.x : share;
        if (rb <= ra || sh * U >= rb - ra) {
            if (ph == 0) { ph = 1; continue; }
            break;
        }
        u = ra + sh * U;
        u_end = min(rb, u + U);
        ti = last_le_u32(kt_prefix, ntasks + 1, u);
    }

    while (u < u_end) {
        const GemmTask tk = tasks[ti];
        const GemmProb pb = probs[tk.prob];
        const uint32_t tbase = kt_prefix[ti];
        const int nkt = (int)(kt_prefix[ti + 1] - tbase);
        const int kt0 = (int)(u - tbase);
        const int kt1 = min(nkt, kt0 + (int)(u_end - u));
        u += (uint32_t)(kt1 - kt0);
        ++ti;

        const int ci0 = tk.tm * GEMM_BM, cj0 = tk.tn * GEMM_BN;
        // the task covers the K steps [tk.kt0, tk.kt0 + nkt) of its tile: operand pointers and K are those of the slice
        const int M = pb.M, N = pb.N;
        const int lda = pb.lda;
        const int kfirst = (int)tk.kt0 * GEMM_BK;
        const int K = min(pb.K - kfirst, nkt * GEMM_BK);
        const double* __restrict__ Yg = Lsx + pb.y_off + ci0 + (int64_t)kfirst * lda;
        const double* __restrict__ Xg = Lsx + pb.x_off + cj0 + (int64_t)kfirst * lda;

        if (MODE == 1) {
            // relative map of this tile's rows/columns inside the target panel (precomputed once per plan)
            if (tid < GEMM_BM) {
                const int ci = ci0 + tid;
                rowmap[tid] = (ci < M) ? RelMap[pb.map_off + ci] : 0;
            } else if (tid < GEMM_BM + GEMM_BN) {
                const int cj = cj0 + (tid - GEMM_BM);
                colmap[tid - GEMM_BM] = (cj < N) ? RelMap[pb.map_off + cj] : 0;
            }
        }

        // global -> register staging: lane handles the ROW PAIR (2*(tid & 63), +1) for k = (tid >> 6) + GEMM_WAVES*q:
        // one 16-byte load and one ds_write_b128 per pair (a wave reads 128 consecutive rows = 1 KiB of one panel
        // column).  gfx950 services 16-byte global loads from 8-byte-aligned addresses (tools/unaligned_load_test.hip),
        // so no alignment of the panel offsets is required.  Loads are unconditional: rows are clamped to a valid
        // pair and k to K-1; the values of invalid rows / k are zeroed when they are staged into LDS.
        const int prow = 2 * (tid & 63), pk0 = tid >> 6;
        const bool y0_ok = (ci0 + prow) < M, y1_ok = (ci0 + prow + 1) < M;
        const bool x0_ok = (cj0 + prow) < N, x1_ok = (cj0 + prow + 1) < N;
        const double* __restrict__ yp = Yg + (y0_ok ? prow : 0);
        const double* __restrict__ xp = Xg + (x0_ok ? prow : 0);
        double2_t ry[DMA ? 1 : SQ], rx[DMA ? 1 : SQ];

        auto load_tile = [&](int k0) {
            if (DMA) return;
#pragma unroll
            for (int q = 0; q < SQ; ++q) {
                const int64_t off = (int64_t)min(k0 + pk0 + GEMM_WAVES * q, K - 1) * lda;
                ry[q] = *reinterpret_cast<const double2_t*>(yp + off);
                rx[q] = *reinterpret_cast<const double2_t*>(xp + off);
            }
        };
        auto store_tile = [&](int buf, int k0) {
            if (DMA) {
                // wave `pk0` fills k rows pk0 and pk0 + 8 of both operands: four 1 KiB DMAs per K step and wave
#pragma unroll
                for (int q = 0; q < SQ; ++q) {
                    const int kl = pk0 + GEMM_WAVES * q, k = k0 + kl;
                    const bool kin = k < K;                                     // wave-uniform
                    const double* ysrc = yp + (int64_t)min(k, K - 1) * lda;
                    const double* xsrc = kin ? xp + (int64_t)k * lda : g_zero_page + prow;
                    glds16(ysrc, lds_addr(&Ys[buf][kl][0]));
                    glds16(xsrc, lds_addr(&Xs[buf][kl][0]));
                }
                return;
            }
#pragma unroll
            for (int q = 0; q < SQ; ++q) {
                const bool kin = (k0 + pk0 + GEMM_WAVES * q) < K;
                double2_t vy = ry[q], vx = rx[q];
                vy.x = (kin && y0_ok) ? vy.x : 0.0; vy.y = (kin && y1_ok) ? vy.y : 0.0;
                vx.x = (kin && x0_ok) ? vx.x : 0.0; vx.y = (kin && x1_ok) ? vx.y : 0.0;
                *reinterpret_cast<double2_t*>(&Ys[buf][pk0 + GEMM_WAVES * q][prow]) = vy;
                *reinterpret_cast<double2_t*>(&Xs[buf][pk0 + GEMM_WAVES * q][prow]) = vx;
            }
        };

        // a wave whose 64x64 quadrant lies entirely outside the lower trapezoid does no MFMA work
        // (wave-uniform by construction; readfirstlane lets the compiler branch on it with the scalar unit)
        const int qci0 = ci0 + wm * 64, qcj0 = cj0 + wn * WCJ;
        const bool quad_active = __builtin_amdgcn_readfirstlane((int)((qci0 < M) && (qcj0 < N) && (qci0 + 63 >= qcj0))) != 0;

        double4_t acc[TMN][4];
#pragma unroll
        for (int a = 0; a < TMN; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

        // Software pipeline, one barrier per K step: while step kt is multiplied out of LDS buffer `buf`, the
        // registers holding step kt+1 are written to the other buffer and re-filled with step kt+2 -- both in
        // the shadow of this wave's own MFMAs (an MFMA occupies the matrix pipe for 64 cycles after it issues).
        load_tile(kt0 * GEMM_BK);
        store_tile(0, kt0 * GEMM_BK);
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        SF_STAMP(1);
        load_tile((kt0 + 1) * GEMM_BK);
        int buf = 0;
        if (quad_active) {
            for (int kt = kt0; kt < kt1; ++kt) {
#pragma unroll
                for (int kk = 0; kk < GEMM_BK / 4; ++kk) {
                    double a[TMN], b[4];
#pragma unroll
                    for (int t = 0; t < TMN; ++t) a[t] = Xs[buf][kk * 4 + fk][wn * WCJ + t * 16 + fr];
#pragma unroll
                    for (int t = 0; t < 4; ++t) b[t] = Ys[buf][kk * 4 + fk][wm * 64 + t * 16 + fr];
                    if (kk == 0) store_tile(buf ^ 1, (kt + 1) * GEMM_BK);
                    if (kk == 1) load_tile((kt + 2) * GEMM_BK);
#pragma unroll
                    for (int tm = 0; tm < TMN; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 4; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
                }
                if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next K step's tile has landed
                __syncthreads();
                buf ^= 1;
            }
        } else {
            for (int kt = kt0; kt < kt1; ++kt) {     // staging only, same barriers
                store_tile(buf ^ 1, (kt + 1) * GEMM_BK);
                load_tile((kt + 2) * GEMM_BK);
                if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                buf ^= 1;
            }
        }

        SF_STAMP(2);
#ifdef SF_EXP_SKIP_EPILOGUE      // ablation for tools/experiments/gemm_overhead.sh: one store per lane keeps the accumulators live
        if (quad_active) {
            double v = 0.0;
            for (int tm = 0; tm < TMN; ++tm) for (int tn = 0; tn < 4; ++tn) for (int r = 0; r < 4; ++r) v += acc[tm][tn][r];
            if (v == 12345.678) Lsx[pb.c_off] = v;
        }
#else
        if (quad_active) {
            double* __restrict__ Cg = Lsx + pb.c_off;
            const int64_t ldc = pb.ldc;
#pragma unroll
            for (int tm = 0; tm < TMN; ++tm) {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    const int lci = wm * 64 + tn * 16 + fr;
                    const int ci = ci0 + lci;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int lcj = wn * WCJ + tm * 16 + fk + 4 * r;
                        const int cj = cj0 + lcj;
                        if (ci < M && cj < N && ci >= cj + (pb.strict & 1)) {
                            const double v = acc[tm][tn][r];
                            if (MODE == 1) {
                                unsafeAtomicAdd(Cg + rowmap[lci] + (int64_t)colmap[lcj] * ldc, -v);
                            } else {
                                double* dst = Cg + ci + (int64_t)cj * ldc;
                                // always the atomic form: a plain read-modify-write makes every one of the lane's 32 loads a
                                // dependent round trip (wait, subtract, store); the atomic needs no return value (measured:
                                // 49.6 vs 46.3 TFLOP/s at K = 256, profiles/r01_g_gemm_microbench_8wave.txt)
                                unsafeAtomicAdd(dst, -v);
                            }
                        }
                    }
                }
            }
        }
#endif
        // the next tile re-uses the LDS buffers and the relative maps.  (A barrier that orders LDS only -- s_waitcnt lgkmcnt(0) +
        // s_barrier, letting the tile's atomics drain under the next claim -- measured neutral: 544.2 vs 543.1 ms at 128^3.)
        __syncthreads();
        SF_STAMP(3);
#ifdef SF_EXP_TIMING
        ++exp_tile;
        SF_STAMP(0);
#endif
    }
    if (ph == 2) break;
    if (ph == 0) ph = 1;
    }   // phases
#ifdef SF_EXP_TIMING
    if (tid == 0 && g_exp_stamps) {
        g_exp_stamps[(size_t)gridDim.x * 256 + blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();
        g_exp_stamps[(size_t)gridDim.x * 256 + blockIdx.x * 4 + 3] = wall_clock64();
    }
#endif
}

// ---------------------------------------------------------------------------------------------------
// Fused 64-column step of the in-panel factorization (latency-bound steps: up to a few rounds of workgroups).
// ONE launch does what used to be three (left-looking update K = 64 t, POTRF / GETRF of the diagonal block, TRSM of
// the rows below):
//   diagonal workgroup (one per panel):   D <- D - Y_D Y_D^T (MFMA) ; D <- chol(D)  (blocked, wave 0 + MFMA) ; publish
//   row workgroups (one per 64 rows):     R <- R - Y_R Y_D^T (MFMA) ; wait for D ; R <- R D^{-T} (blocked, MFMA)
// The row workgroups' update -- most of the step's work -- runs WHILE the diagonal block is being factored; they
// pick the factored block up through a per-(panel, step) flag (release fence + relaxed store by the diagonal
// workgroup, relaxed polls + one acquire fence by the waiting one, device scope; the flag value is the
// factorization's epoch, so flags are never reset).  Liveness: tasks are handed out by an atomic ticket in the order
// the workgroups actually start, and the diagonal tasks (which never wait) come first in the list -- see the top of
// the kernel.  No assumption about the dispatch order or about co-residency of the grid is made.  The wait is
// bounded all the same (info |= 2 -> SF_ERR_HIP instead of a hang).
// Every element of the block column is read and written once.  4 waves (2 x 2), each a 32 x 32 sub-tile = 2 x 2
// v_mfma_f64_16x16x4_f64 tiles; K is short (<= 448), so the MFMA fragments are loaded straight from the panel
// (16 consecutive rows x 4 k per load), 32 k ahead in registers, no LDS staging and no barriers in the K loop.
// The updated 64 x 64 tile then goes to LDS (U[column][row]) where the POTRF wave / the blocked solve picks it up.
// ---------------------------------------------------------------------------------------------------
#ifndef SF_LU_STEP_WGS
#define SF_LU_STEP_WGS 3
#ifndef SF_GEMM_MIN_UNITS_DEFAULT
#define SF_GEMM_MIN_UNITS_DEFAULT 16     // 16-deep K steps; swept in round 4: config 3 11.20 -> 10.99 ms, config 5 and 128^3 unchanged (tools/experiments/gemm_min_units.sh)
#endif
#ifndef SF_POTRF_PW
#define SF_POTRF_PW 16        // columns per panel of the fused step's 64 x 64 POTRF (16 or 32; 32 measured slower, see k_step)
#endif      // workgroups per CU the LU variant of k_step is compiled for (168 VGPRs; the throughput-bound launches of the lower levels want the third)
#endif
constexpr int ST_ULD = ST_ROWS + 1;      // LDS column stride of the updated tile U[c][r]
constexpr int ST_KC = 32;                // K chunk of the update's LDS-staged operand
constexpr int ST_XLD = ST_ROWS + 16;     // its LDS row stride
constexpr int ST_SPIN_LIMIT = 1 << 22;   // ~ seconds

#ifdef SF_EXP_STEP_STAMPS       // tools/experiments/step_stamps.sh: where the diagonal workgroup of a step spends its time (100 MHz stamps)
__device__ unsigned long long* g_step_stamps = nullptr;
void exp_set_step_stamps(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_step_stamps), &p, sizeof(p)); }
#define ST_STAMP(slot) do { if (is_diag && tid == 0 && g_step_stamps) g_step_stamps[slot] = wall_clock64(); } while (0)
#else
#define ST_STAMP(slot) do { } while (0)
#endif

// Agent-scope coherent load (global_load ... sc1): sees what another workgroup of the RUNNING launch -- possibly on another XCD, whose
// L2 is not coherent with this one's -- stored and released before it raised a flag this workgroup has observed.  The row tasks
// read the few values they take from their diagonal task this way (the factored 64 x 64 block, the 16 x 16 inverses, LU: the pivot
// list) INSTEAD of an agent-scope acquire fence, which invalidates the whole XCD's L2 (buffer_inv sc1): 1.7 us per waiting
// workgroup, 4 us with 500 of them polling (tools/experiments/README.md, step_fence.sh), and every co-resident task's cached
// operands with it.
__device__ __forceinline__ double ld_agent(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_agent(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool LU>
__global__ void __launch_bounds__(256, LU ? SF_LU_STEP_WGS : 3)   // LU: the unblocked GETRF keeps a 64-value row per lane
k_step(const StepTask* __restrict__ tasks, double* __restrict__ Lsx, int* __restrict__ flags, int epoch, int* __restrict__ info,
       double* __restrict__ tinv, int* __restrict__ ticket, PivotCtl pc) {
    // ONE LDS array, re-used by the phases of a task:
    //   update:            X staging buffers Xs[2][32][80]
    //   diagonal task:     U[c][r], the updated block (POTRF / GETRF works on it)
    //   row task:          Dt[k][j] = D(j,k) | Tl[4][16][16], the inverses of the four 16 x 16 diagonal sub-blocks of D
    constexpr int SMEM = 2 * ST_KC * ST_XLD;
    __shared__ __attribute__((aligned(16))) double smem[SMEM];
    static_assert(NB == ST_ROWS && NB == 64, "one wavefront per 64 x 64 tile");
    static_assert(NB * ST_ULD <= SMEM && NB * NB + 4 * 256 <= SMEM, "phases must fit the LDS array");
    double* __restrict__ U = smem;
    double (*Dt)[NB] = reinterpret_cast<double (*)[NB]>(smem);

    // Tasks are claimed in EXECUTION order (one atomic ticket per workgroup), not by blockIdx: the diagonal tasks come
    // first in the list, so every one of them is held by a workgroup that is already running -- and never waits -- by
    // the time any row task is claimed.  A waiting workgroup can therefore never keep the one it waits for off the
    // chip, whatever order the hardware dispatches the grid in and whatever else shares the GPU.
    __shared__ int s_ticket;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef SF_EXP_STEP_STAMPS
    const unsigned long long st_entry = wall_clock64();
#endif
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const StepTask t = tasks[__builtin_amdgcn_readfirstlane(s_ticket)];
    // LU, mode bit 1: a PRE-UPDATE task -- the far part (columns [J, diag - 64)) of the left-looking update of the diagonal block of
    // the NEXT step, done one launch early and off the critical path (everything it reads is final when this launch starts); that
    // step's diagonal workgroup then only applies the last 64 columns before it factors (see the task list in sf_chol_plan.hip)
    const bool is_pre = LU && (t.mode & 2);
    const bool is_diag = t.row0 == t.diag && !is_pre;
#ifdef SF_EXP_STEP_STAMPS
    if (is_diag && tid == 0 && g_step_stamps) g_step_stamps[0] = st_entry;
#endif
    ST_STAMP(1);
    const int fr = lane & 15, fk = lane >> 4;
    const int64_t ld = t.ld;
    const int b = t.b, nrows = t.nrows;
    const int nhp = ((is_pre ? t.diag - NB : t.diag) - t.J) / NB;             // K = 64 nhp = 2 nhp chunks of ST_KC = 32
    double* __restrict__ Ag = Lsx + t.panel + t.row0 + (int64_t)t.diag * ld;          // this tile: rows row0.., columns diag..
    // the diagonal block in the panel the OTHER operand comes from (Cholesky: the same panel; LU: L rows are updated with
    // and solved against the U^T panel's block and vice versa)
    const double* __restrict__ Dg = Lsx + t.xpanel + t.diag + (int64_t)t.diag * ld;

    // Update: wave w owns rows 16 w .. 16 w + 15 of the tile x all 64 columns (4 MFMA tiles).  Its own rows' fragments
    // (B operand) come straight from the panel, one chunk ahead in registers -- each element is loaded once; the
    // diagonal block's rows (A operand, shared by the 4 waves) go through LDS in 32-deep chunks, staged like k_gemm
    // (16-byte row-pair loads, [k][row] image, double-buffered, one barrier per chunk).
    double4_t acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // (Cholesky diagonal tasks come with J == diag, i.e. no update: the block was brought up to date right-looking by the
    // earlier steps of this outer block -- each step's row tile that holds a future diagonal block pushes its own X X^T
    // into it, see the end of the kernel -- so the diagonal workgroup, the step's critical path, starts its POTRF at once)
    if (nhp > 0) {
        double (*Xs)[ST_KC][ST_XLD] = reinterpret_cast<double (*)[ST_KC][ST_XLD]>(smem);
        const int nch = 2 * nhp;
        const int prow = 2 * (tid & 31), pk0 = tid >> 5;       // row pair, k = pk0 + 8 q
        // rows beyond b / nrows are clamped: their values only reach accumulator entries replaced by the padding below
        const double* __restrict__ xp = Lsx + t.xpanel + t.diag + (int64_t)t.J * ld + ((prow < b) ? prow : 0);
        const double* __restrict__ yp = Lsx + t.panel + t.row0 + (int64_t)(t.J + fk) * ld + min(16 * wave + fr, nrows - 1);
        // K is short and most launches are a few hundred workgroups (one wave per SIMD): the loop lives on the distance of its
        // prefetches, not on occupancy.  Y fragments: a ring of 4 HALF chunks (16 k each) in registers, every half chunk loaded
        // 1.5 chunks before its use; X: two register sets, loaded 3 chunks ahead of their use and stored to the other LDS buffer
        // one chunk ahead.
        const int nhalf = 2 * nch;
        double2_t rx[2][4];
        double fy[4][4];
        auto load_x = [&](int sel, int h) __attribute__((always_inline)) {
            const int hc = min(h, nch - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) rx[sel][q] = *reinterpret_cast<const double2_t*>(xp + (int64_t)(hc * ST_KC + pk0 + 8 * q) * ld);
        };
        auto store_x = [&](int buf, int sel) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<double2_t*>(&Xs[buf][pk0 + 8 * q][prow]) = rx[sel][q];
        };
        auto load_y = [&](int slot, int hh) __attribute__((always_inline)) {          // half chunk hh -> ring slot
            const int hc = min(hh, nhalf - 1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) fy[slot][kk] = yp[(int64_t)(hc * (ST_KC / 2) + 4 * kk) * ld];
        };
        // chunk h (h & 1 == par): X from Xs[par], Y from the ring slots 2 par, 2 par + 1; stores chunk h + 1 (register set
        // par ^ 1) and reloads that set with chunk h + 3; half chunks 2 h + 3 and 2 h + 4 are requested on the way
        auto compute = [&](int par, int h) __attribute__((always_inline)) {
            load_y((2 * par + 3) & 3, 2 * h + 3);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                double a[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = Xs[par][4 * kk + fk][16 * q + fr];
                if (kk == 0) store_x(par ^ 1, par ^ 1);
                if (kk == 1) load_x(par ^ 1, h + 3);
                if (kk == 4) load_y(2 * par, 2 * h + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], fy[2 * par + (kk >> 2)][kk & 3], acc[q], 0, 0, 0);
            }
        };
        load_x(0, 0);
        load_x(1, 1);
        load_y(0, 0);
        load_y(1, 1);
        load_y(2, 2);
        store_x(0, 0);
        load_x(0, 2);
        __syncthreads();
        for (int hp = 0; hp < nhp; ++hp) {
            compute(0, 2 * hp);
            __syncthreads();
            compute(1, 2 * hp + 1);
            __syncthreads();
        }
    }

    if (is_pre) {
        // D <- D - (far part of the update), in place: D(ci,cj) lives in the L panel for cj < ci, in the U^T panel (transposed)
        // otherwise.  One writer: this launch's row tasks write other columns of these rows, the block's own step comes later.
        double* __restrict__ Dw = Lsx + t.xpanel + t.diag + (int64_t)t.diag * ld;
        const int ci = 16 * wave + fr;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cj = 16 * q + fk + 4 * r;
                const int cic = min(ci, b - 1), cjc = min(cj, b - 1);
                const double dl = Ag[cic + (int64_t)cjc * ld], du = Dw[cjc + (int64_t)cic * ld];       // unconditional, clamped
                if (ci < b && cj < b) {
                    if (cj < ci) Ag[ci + (int64_t)cj * ld] = dl - acc[q][r];
                    else Dw[cj + (int64_t)ci * ld] = du - acc[q][r];
                }
            }
        return;
    }

    // Row task: the updated tile stays in the MFMA accumulator layout (wave w: rows 16 w + fr, column tile q:
    // columns 16 q + fk + 4 r), which is also the B-operand layout of the next MFMA -- the solve below runs on registers
    double4_t rt[4];
    if (!is_diag) {
        const int ci = 16 * wave + fr, cic = min(ci, nrows - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cj = 16 * q + fk + 4 * r;
                const double av = Ag[cic + (int64_t)min(cj, b - 1) * ld];          // unconditional, clamped (see below)
                rt[q][r] = (ci < nrows && cj < b) ? av - acc[q][r] : 0.0;
            }
    }
    // accumulators -> U[cj][ci] = A(ci, cj) - update (padded with the identity / zeros); the staging buffers are dead
    // (the chunk loop ends with a barrier)
    if (LU && is_diag) {
        // LU diagonal task: the 64 x 64 block lives in two panels -- D(ci,cj), cj < ci in the L panel, the rest transposed in the U^T
        // panel.  Both halves are read column by column (L) and row by row (U^T) with the lane along the panels' contiguous
        // direction: 32 fully coalesced loads per thread in flight, then the image U[cj][ci] = D(ci, cj) (identity padding beyond
        // b), then the update (the last 64 columns' contribution, in the accumulator layout) subtracted in LDS.  (The first form
        // read the U half in the accumulator layout -- 64 cache lines per load instruction: 3.4 us from entry to the first panel
        // against 2.6 now, profiles/r04_*_step_stamps.txt.  Requested BEFORE the update's K loop the loads overlap it, but their 64
        // registers stay live through the loop for every task of the launch: 247 VGPRs, 2 workgroups per CU instead of 3, and the
        // throughput-bound launches of the lower levels lose more than the diagonal workgroup gains.)
        {
            // (one half at a time: 32 VGPRs each; both in flight at once pushed the kernel past 168 VGPRs = 3 workgroups per CU)
            const int lc = min(lane, b - 1);
            double blk[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) blk[i] = Ag[lc + (int64_t)min(16 * wave + i, b - 1) * ld];     // D(lane, k): column k below its diagonal
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = 16 * wave + i;
                if (lane > k) U[k * ST_ULD + lane] = (lane < b && k < b) ? blk[i] : 0.0;                // (ci = lane, cj = k)
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) blk[i] = Dg[lc + (int64_t)min(16 * wave + i, b - 1) * ld];     // D(k, lane) = PU(lane, k): row k from its diagonal on
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = 16 * wave + i;
                if (lane >= k) U[lane * ST_ULD + k] = (lane < b && k < b) ? blk[i] : ((lane == k) ? 1.0 : 0.0);    // (ci = k, cj = lane)
            }
        }
        if (nhp > 0) {
            __syncthreads();
            const int ci = 16 * wave + fr;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cj = 16 * q + fk + 4 * r;
                    if (ci < b && cj < b) U[cj * ST_ULD + ci] -= acc[q][r];
                }
        }
    } else if (is_diag) {
        const int ci = 16 * wave + fr;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cj = 16 * q + fk + 4 * r;
                // loads are unconditional (clamped addresses) and selected afterwards: a load under a per-element
                // condition costs a branch and its own wait, i.e. 16 dependent round trips per lane
                const int cic = min(ci, nrows - 1), cjc = min(cj, b - 1);
                double v = (ci == cj) ? 1.0 : 0.0;
                const double av = Ag[cic + (int64_t)cjc * ld];
                if (ci < nrows && cj < b && cj <= ci) v = av - acc[q][r];
                U[cj * ST_ULD + ci] = v;
            }
    }

    if (LU && is_diag) {
        // LU of the updated block with threshold pivoting inside it (pc.tol > 0; implicit interchanges: rows stay where they are
        // until the final store), BLOCKED by 16 columns like the Cholesky path below:
        //   panel    wave 0, lane r holds the 16 panel entries of row r: getrf_panel_wave (pivot search over the rows not used yet,
        //            v_readlane broadcasts of the pivot row, multipliers by v_rcp_f64 + two Newton steps);
        //   U12      the 16 pivot rows of the panel in the columns to its right: one thread per column, forward substitution
        //            with the panel's multipliers (LDS broadcasts);
        //   trailing every wave its 16 rows x the columns to the right with MFMA out of LDS, the multipliers of rows that are
        //            already used (in this or an earlier panel) masked to zero, the U12 rows gathered through the pivot list.
        // The unblocked form (a 64-value row per lane in getrf_panel_wave<true, 64>, 4,000 v_readlane pairs on the critical path) cost 68 us per
        // step and 256 VGPRs; see DESIGN 6b for the figures of this one.
        __shared__ int s_piv[NB], s_pos[NB];
        if (tid < NB) { s_pos[tid] = -1; s_piv[tid] = tid; }
        __syncthreads();
        ST_STAMP(2);
        bool bad = false, active = lane < b;
        int np = 0, pos = lane;
        bool nat_all = true;        // wave 0: every pivot so far was the natural row (wave-uniform)
#pragma unroll 1
        for (int q = 0; q < NB / 16; ++q) {
            const int c0 = 16 * q;
            if (wave == 0) {
                double a[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) a[u] = U[(c0 + u) * ST_ULD + lane];
                // natural pivots first (straight-line code, see getrf_panel_natural); the general search only when one of them fails
                bool done = false;
                if (nat_all) {
                    done = getrf_panel_natural<16>(a, lane, c0, b, pc.tol, pc.eps);
                    if (done) {
                        if (lane >= c0 && lane < c0 + 16) { active = false; s_piv[lane] = lane; }       // pos stays = lane
                    } else {
#pragma unroll
                        for (int u = 0; u < 16; ++u) a[u] = U[(c0 + u) * ST_ULD + lane];
                    }
                }
                if (!done) {
                    nat_all = false;
                    getrf_panel_wave<true, 16>(a, lane, c0, b, pc.tol, pc.eps, bad, np, pos, active, s_piv);
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) U[(c0 + u) * ST_ULD + lane] = a[u];
                s_pos[lane] = active ? -1 : pos;
            }
            if (q == NB / 16 - 1 || c0 + 16 >= b) break;      // nothing but identity padding to the right (narrow panel)
            __syncthreads();
            const int ntr = NB - c0 - 16;               // columns to the right of the panel
            if (tid < ntr) {
                const int c = c0 + 16 + tid;
                double x[16];
                int pr[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) { pr[k] = s_piv[c0 + k]; x[k] = U[c * ST_ULD + pr[k]]; }
#pragma unroll
                for (int k = 1; k < 16; ++k)
#pragma unroll
                    for (int jj = 0; jj < k; ++jj) x[k] -= U[(c0 + jj) * ST_ULD + pr[k]] * x[jj];
#pragma unroll
                for (int k = 1; k < 16; ++k) U[c * ST_ULD + pr[k]] = x[k];
            }
            __syncthreads();
            {
                const int ci = 16 * wave + fr;          // this wave's 16 rows
                const bool free_row = s_pos[ci] < 0;
                double lf[4];
                int pk[4];
#pragma unroll
                for (int sgm = 0; sgm < 4; ++sgm) {
                    const double v = U[(c0 + 4 * sgm + fk) * ST_ULD + ci];
                    lf[sgm] = free_row ? v : 0.0;                                                   // B[k][j = ci]
                    pk[sgm] = s_piv[c0 + 4 * sgm + fk];
                }
                for (int ct = q + 1; ct < NB / 16; ++ct) {
                    const int cb = 16 * ct;
                    double4_t d;
#pragma unroll
                    for (int r = 0; r < 4; ++r) d[r] = U[(cb + fk + 4 * r) * ST_ULD + ci];          // D[i = column][j = row ci]
#pragma unroll
                    for (int sgm = 0; sgm < 4; ++sgm)
                        d = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[(cb + fr) * ST_ULD + pk[sgm]], lf[sgm], d, 0, 0, 0);   // A[i][k] = U12(k, cb + i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) U[(cb + fk + 4 * r) * ST_ULD + ci] = d[r];
                }
            }
            __syncthreads();
            ST_STAMP(3 + q);
        }
        ST_STAMP(6);
        if (wave == 0) {
            if (bad && lane == 0) atomicOr(info, 1);
            if (np > 0 && lane == 0) atomicAdd(pc.nperturb, np);
            if (pc.pivpos && lane < b) {
                const int g0 = t.first_col + t.diag;
                pc.pivpos[g0 + lane] = g0 + pos;
                pc.pivinv[g0 + pos] = g0 + lane;
            }
        }
        __syncthreads();
        {
            // The factored block goes to the two panels, rows at their pivot positions, every store instruction along a panel's
            // contiguous direction.  L part: wave w takes the columns 16 w .. 16 w + 15, lane = row POSITION (its values come from
            // the row s_piv[position] of the LDS image); with interchanges the image itself is brought into pivot order on the way
            // (a wave's reads of a column precede its writes, no other wave touches these columns).  U part: from the ordered image,
            // wave w takes the rows 16 w .. 16 w + 15, lane = column.  (Stored in the accumulator-like layout, the first form issued
            // 64 cache lines per store instruction for the U half: 2.0 us + a longer drain before the flag.)
            const int src = s_piv[lane];                     // the row that ended at position `lane`
            const bool moved = !__all(src == lane);         // same answer in every wave (one pivot list)
            double* __restrict__ PUd = Lsx + t.xpanel + t.diag + (int64_t)t.diag * ld;
            double a[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) a[u] = U[(16 * wave + u) * ST_ULD + src];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = 16 * wave + u;
                if (lane < b && c < lane) Ag[lane + (int64_t)c * ld] = a[u];
                if (moved) U[c * ST_ULD + lane] = a[u];
            }
            if (moved) __syncthreads();
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r = 16 * wave + u;
                const double v = U[lane * ST_ULD + r];
                if (lane < b && r <= lane) PUd[lane + (int64_t)r * ld] = v;
            }
        }
        ST_STAMP(7);
        {
            // inverses of the 16 x 16 diagonal sub-blocks the row tasks solve against: of U11^T (lower, for the L rows) at
            // tinv[slot][0][w], of the unit-lower L11 (for the U^T rows) at tinv[slot][1][w]; wave w does block w of both -- the two
            // sets side by side in ONE substitution: lanes 0..15 hold the columns of the first, lanes 16..31 of the second (the
            // matrix entry a step multiplies by is an LDS broadcast per set: two addresses per read), lanes 32..63 repeat them.
            // (One set after the other in every lane, the first form, was 3.3 us of the diagonal workgroup's 40; Cholesky's one set 2.0.)
            const int o = 16 * wave, j = lane & 15;
            const bool lset = (lane & 16) != 0;
            // entry (r, c) of the set's matrix: U11^T(r,c) = U11(c,r) = row o+c, column o+r of the block (U[column][row] image:
            // offset r * ULD + c); L11(r,c) = row o+r, column o+c (offset c * ULD + r) -- per-lane strides, ONE read per step
            const int sr = lset ? 1 : ST_ULD, sc = lset ? ST_ULD : 1;
            const double* __restrict__ Ub = U + o * ST_ULD + o;
            double w[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double sacc = (r == j) ? 1.0 : 0.0;
#pragma unroll
                for (int c = 0; c < r; ++c) sacc -= Ub[r * sr + c * sc] * w[c];
                const double trr = Ub[r * (ST_ULD + 1)];
                const double rp = rcp_full(trr);
                w[r] = lset ? sacc : sacc * rp;
            }
            double* __restrict__ out = tinv + (int64_t)t.slot * 2048 + (lset ? 1024 : 0) + wave * 256 + j * 16;
            if (lane < 32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) out[r] = w[r];
            }
        }
        ST_STAMP(8);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(flags + t.flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ST_STAMP(9);
        return;
    }
    if (is_diag) {
        __syncthreads();
        ST_STAMP(2);
        // POTRF of the updated block, blocked by 16 columns.  Panel part: wave 0, lane r holds row r of the 16 columns
        // (k_potrf_block's scheme; the column scaling of all 64 rows comes with it, so there is no separate TRSM).
        // Trailing part: every wave updates its 16 rows of the columns to the right with MFMA out of LDS,
        //   U[cj][ci] -= sum_k L(cj,k) L(ci,k),  k in the panel  (A operand = -L rows cj, B operand = L rows ci)
        bool bad = false;
        // PW columns per panel.  32 = two panels per block instead of four -- half the barriers and trailing passes on the step's
        // critical path, at the price of more of the elimination in the broadcast form (496 instead of 120 (column, column) pairs per
        // panel) -- was measured (round 4) and is SLOWER: the panel becomes issue-bound on its v_readlane / v_fma pairs; config 3
        // 11.6 ms against 11.2 (fused steps 5.4 against 5.0), 128^3 fused steps 58.0 against 55.7 ms.
        constexpr int PW = SF_POTRF_PW;
#pragma unroll
        for (int q = 0; q < NB / PW; ++q) {
            const int c0 = q * PW;
            if (wave == 0) {
                double a[PW];
#pragma unroll
                for (int u = 0; u < PW; ++u) a[u] = U[(c0 + u) * ST_ULD + lane];
                // dnext: what the NEXT column's diagonal entry will be, formed in its own lane (a[j+1] - lj^2 there: the same fused
                // multiply-add as the general update below, whose multiplier for that lane is the lane's own lj) -- so the chain from one
                // pivot to the next has ONE lane broadcast in it instead of two
                double dnext = a[0];
#pragma unroll
                for (int j = 0; j < PW; ++j) {
                    if (PW > 16 && j == 16 && c0 + 16 >= b) break;          // narrow block: the rest of the panel is identity padding
                    const double djj = readlane_f64(dnext, c0 + j);
                    bad = bad || !(djj > 0.0);
                    // (rsqrt_full; no select: the diagonal lane's own entry is djj, lanes above the diagonal carry values nobody reads --
                    //  see potrf_block_w)
                    const double rinv = rsqrt_full(djj);
                    const double lj = a[j] * rinv;
                    a[j] = lj;
                    if (j + 1 < PW) dnext = __builtin_fma(-lj, lj, a[j + 1]);
#pragma unroll
                    for (int c = j + 1; c < PW; ++c) a[c] = __builtin_fma(-lj, readlane_f64(lj, c0 + c), a[c]);
                }
#pragma unroll
                for (int u = 0; u < PW; ++u) {
                    U[(c0 + u) * ST_ULD + lane] = a[u];
                    if (lane < b && c0 + u <= lane) Ag[lane + (int64_t)(c0 + u) * ld] = a[u];
                }
            }
            if (q == NB / PW - 1 || c0 + PW >= b) break;      // nothing but identity padding to the right (narrow panel)
            __syncthreads();
            if (16 * wave >= c0 + PW) {
                const int ci = wave * 16 + fr;                  // this wave's 16 rows
                double lf[PW / 4];
#pragma unroll
                for (int sgm = 0; sgm < PW / 4; ++sgm) lf[sgm] = U[(c0 + 4 * sgm + fk) * ST_ULD + ci];            // B[k][j = ci]
                for (int ct = (c0 + PW) / 16; ct <= wave; ++ct) {
                    const int cb = ct * 16;
                    double4_t d;
#pragma unroll
                    for (int r = 0; r < 4; ++r) d[r] = U[(cb + fk + 4 * r) * ST_ULD + ci];                    // D[i = cj][j = ci]
#pragma unroll
                    for (int sgm = 0; sgm < PW / 4; ++sgm)
                        d = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[(c0 + 4 * sgm + fk) * ST_ULD + cb + fr], lf[sgm], d, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) U[(cb + fk + 4 * r) * ST_ULD + ci] = d[r];
                }
            }
            __syncthreads();
            ST_STAMP(3 + q);
        }
        ST_STAMP(6);
        if (bad && wave == 0 && lane == 0) atomicOr(info, 1);
        // Inverses of the four 16 x 16 diagonal sub-blocks of L (what MAGMA-style TRSMs use): wave w inverts block w by
        // forward substitution, lane j (< 16) holds column j of the inverse, the entries of T are LDS broadcasts.  The
        // row tasks then solve with MFMA only -- X_q = R_q T_q^{-T} -- instead of a one-wave substitution.
        __syncthreads();
        {
            const int o = 16 * wave;
            double wv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double sacc = (r == (lane & 15)) ? 1.0 : 0.0;
#pragma unroll
                for (int c = 0; c < r; ++c) sacc -= U[(o + c) * ST_ULD + o + r] * wv[c];
                const double trr = U[(o + r) * ST_ULD + o + r];
                const double rp = rcp_full(trr);
                wv[r] = sacc * rp;          // rows above the diagonal come out as exact zeros (zero right-hand side so far)
            }
            // Tinv_w(r, j) at tinv[slot][w][j][r]: the [k][i] image the consumers' MFMA A operand reads
            double* __restrict__ out = tinv + (int64_t)t.slot * 1024 + wave * 256 + (lane & 15) * 16;
            if (lane < 16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) out[r] = wv[r];
            }
        }
        ST_STAMP(8);
        // publish: EVERY storing wave drains, the barrier collects them, then ONE device-scope release by lane 0 and
        // the flag (the explicit waits keep the order whatever the compiler does with the fence's own wait)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
#ifndef SF_EXP_NO_RELEASE_FENCE     // timing ablation only (tools/experiments/step_fence.sh): what the device-scope release costs
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(flags + t.flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ST_STAMP(9);
        return;
    }

    // wait for this panel's diagonal block of this step
    if (tid == 0) {
        int spins = 0;
        // relaxed polls (an acquire load would invalidate this CU's caches at every iteration, and with hundreds of
        // waiting workgroups that slows the whole chip down).  No acquire fence follows: everything this task reads of the
        // diagonal task's output is read with agent-scope loads (ld_agent), issued after the barrier below, i.e. after the flag
        // has been SEEN; the diagonal task released its stores (L2 write-back) before it raised the flag.
        while (__hip_atomic_load(flags + t.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > ST_SPIN_LIMIT) { atomicOr(info, 2); break; }
        }
#ifdef SF_EXP_ACQUIRE_FENCE         // the former protocol (timing comparison only)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (LU && (t.mode & 1) && pc.pivinv) {
        // U^T rows: the tile's 64 columns are the block's rows of U; the diagonal workgroup interchanged rows, so the columns
        // are brought into pivot order (position p <- original column pivinv[p]).  Through LDS ([column][row] image), the
        // accumulator layout is per-lane fixed.  Skipped (wave-uniform test) when the block kept its natural order.
        const int g0 = t.first_col + t.diag;
        int src[16];
        bool ident = true;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cj = 16 * q + fk + 4 * r;
                src[4 * q + r] = (cj < b) ? ld_agent(pc.pivinv + g0 + cj) - g0 : cj;
                ident = ident && src[4 * q + r] == cj;
            }
        if (!__all(ident)) {            // per wave; the waves' rows are disjoint, so is their part of the LDS image
            const int ci = 16 * wave + fr;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) U[(16 * q + fk + 4 * r) * ST_ULD + ci] = rt[q][r];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) rt[q][r] = U[src[4 * q + r] * ST_ULD + ci];
        }
        __syncthreads();                // the image is overwritten by Dt / Tl below
    }
    {
        // X <- R D^{-T} with MFMA only, per wave (its 16 rows are independent of the other waves'): for every 16-column
        // block q   X_q = R_q T_q^{-T}   (A operand = the block's inverse, B operand = R_q as it sits in the registers),
        // then   R_q' -= X_q D(q', q)^T  for the blocks q' to the right (A operand = -D from Dt, B operand = X_q).
        // D = L11 (Cholesky), U11^T (LU, L rows) or the unit-lower L11 (LU, U^T rows: mode bit 0)
        double* __restrict__ Tl = smem + NB * NB;
        const double* __restrict__ tsrc = tinv + (int64_t)t.slot * (LU ? 2048 : 1024) + ((LU && (t.mode & 1)) ? 1024 : 0);
        {
            double dv[NB * NB / 256], tv[4];
#pragma unroll
            for (int i = 0; i < NB * NB / 256; ++i) {       // all loads in flight (clamped addresses), then select + store
                const int e = tid + 256 * i, k = e / NB, j = e % NB;
                dv[i] = ld_agent(Dg + min(j, b - 1) + (int64_t)min(k, b - 1) * ld);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) tv[i] = ld_agent(tsrc + tid + 256 * i);
#pragma unroll
            for (int i = 0; i < NB * NB / 256; ++i) {
                const int e = tid + 256 * i, k = e / NB, j = e % NB;
                Dt[k][j] = (j < b && k < j) ? dv[i] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) Tl[tid + 256 * i] = tv[i];
        }
        __syncthreads();
        const int ci = 16 * wave + fr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (16 * q >= b) break;                              // narrow panel: the remaining blocks are padding
            double4_t x = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int sg = 0; sg < 4; ++sg)
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(Tl[q * 256 + (4 * sg + fk) * 16 + fr], rt[q][sg], x, 0, 0, 0);
#pragma unroll
            for (int qq = q + 1; qq < 4; ++qq)
#pragma unroll
                for (int sg = 0; sg < 4; ++sg)
                    rt[qq] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Dt[16 * q + 4 * sg + fk][16 * qq + fr], x[sg], rt[qq], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cj = 16 * q + fk + 4 * r;
                if (ci < nrows && cj < b) Ag[ci + (int64_t)cj * ld] = x[r];
            }
            if (!LU && t.next_b > 0) {
                // (push, see below) park X_q in LDS where Dt rows 16 q .. 16 q + 15 were: [k][row] image, dead once every
                // wave has passed this iteration.  next_b > 0 implies b == 64: all waves run all four iterations.
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) Dt[16 * q + fk + 4 * r][ci] = x[r];
            }
        }
        if (!LU && t.next_b > 0) {
            // These 64 rows are a FUTURE diagonal block of this outer block (rows = columns [row0, row0 + next_b) of the
            // panel): subtract this step's contribution X X^T from it now (right-looking, lower triangle), so that its own
            // step finds it up to date and its diagonal workgroup -- the step's critical path -- starts the POTRF at once.
            // One workgroup per (step, future block), steps are separate launches: plain read-modify-write.
            __syncthreads();
            double* __restrict__ Dn = Lsx + t.panel + t.row0 + (int64_t)t.row0 * ld;
            const int nb = t.next_b;
            for (int ct = 0; ct <= wave; ++ct) {
                double4_t d = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
                for (int sg = 0; sg < 16; ++sg)         // A[i = cj][k], B[k][j = ci]
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(Dt[4 * sg + fk][16 * ct + fr], Dt[4 * sg + fk][ci], d, 0, 0, 0);
                double old[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cj = 16 * ct + fk + 4 * r;
                    old[r] = Dn[min(ci, nb - 1) + (int64_t)min(cj, nb - 1) * ld];               // unconditional, clamped
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cj = 16 * ct + fk + 4 * r;
                    if (ci < nb && cj <= ci) Dn[ci + (int64_t)cj * ld] = old[r] - d[r];
                }
            }
        }
    }
}

void launch_step(const StepTask* tasks, int ntasks, int lu, double* Lsx, int* flags, int epoch, int* info, double* tinv, int* ticket,
                 PivotCtl pc, hipStream_t st) {
    if (ntasks <= 0) return;
    if (lu) hipLaunchKernelGGL(k_step<true>, dim3(ntasks), dim3(256), 0, st, tasks, Lsx, flags, epoch, info, tinv, ticket, pc);
    else hipLaunchKernelGGL(k_step<false>, dim3(ntasks), dim3(256), 0, st, tasks, Lsx, flags, epoch, info, tinv, ticket, pc);
}

// ---------------------------------------------------------------------------------------------------
// Schur updates with a short inner dimension (K <= 64: the small supernodes of the bottom levels, tens of thousands
// of (descendant, ancestor) pairs of a few dozen rows each).  k_gemm's 128 x 128 tile per 8-wave workgroup spends
// ~8 us per such pair with 7 of its 8 waves idle; here ONE WAVE owns a 64 x 32 tile (4 x 2 MFMA tiles), loads its
// fragments straight from the source panel (no LDS, no barrier, up to 48 loads in flight) and scatters with the same
// relative maps and fp64 atomics.  4 independent tiles per 256-thread workgroup.
// (Measured variants: forcing 3 waves per SIMD / hoisting the relative-map loads made the compiler spill: 9.9 vs 7.7 ms.
// Round 4: PERSISTENT waves walking through chunks of 4 consecutive tiles with the next tile's descriptors prefetched, the map
// entries requested before the K loop and 4-step fragment batches (100 VGPRs + 64 AGPRs, 3 waves per SIMD) -- 7.94 vs 7.68 ms at
// 128^3, 1.65 vs 1.59 ms on config 3, 3.27 vs 3.06 ms on config 5 (gpurun_out r04_h / r04_i): the kernel is not bound by the
// per-tile latency chain but by its load and atomic instructions through the texture path -- 24 eight-byte loads per k-group and
// up to 32 atomics per lane and tile, nothing shared between waves.)
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_update_small(const GemmProb* __restrict__ probs, const GemmTask* __restrict__ tasks, int ntasks,
               double* __restrict__ Lsx, const int32_t* __restrict__ RelMap) {
    const int lane = threadIdx.x & 63;
    // uniform over the wave: say so, or the task, the problem and all address arithmetic live in VGPRs
    const int ti = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (ti >= ntasks) return;
    const GemmTask tk = tasks[ti];
    const GemmProb pb = probs[tk.prob];
    const int fr = lane & 15, fk = lane >> 4;
    const int ci0 = tk.tm * SU_TM, cj0 = tk.tn * SU_TN;
    const int M = pb.M, N = pb.N, K = pb.K;
    const int64_t lda = pb.lda;
    // fragment rows, clamped into the problem: values of rows beyond M / N only reach outputs that are not stored
    const double* __restrict__ yq[4];
    const double* __restrict__ xq[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) yq[q] = Lsx + pb.y_off + min(ci0 + 16 * q + fr, M - 1);
#pragma unroll
    for (int q = 0; q < 2; ++q) xq[q] = Lsx + pb.x_off + min(cj0 + 16 * q + fr, N - 1);

    double4_t acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = (double4_t){0.0, 0.0, 0.0, 0.0};

    const int nkk = (K + 3) >> 2;
    for (int kk0 = 0; kk0 < nkk; kk0 += 8) {
        double fa[8][2], fb[8][4];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = 4 * (kk0 + kk) + fk;
            const int64_t off = (int64_t)min(k, K - 1) * lda;       // unconditional loads; k beyond K is zeroed in the A operand
#pragma unroll
            for (int q = 0; q < 2; ++q) fa[kk][q] = xq[q][off];
#pragma unroll
            for (int q = 0; q < 4; ++q) fb[kk][q] = yq[q][off];
#pragma unroll
            for (int q = 0; q < 2; ++q) fa[kk][q] = (k < K) ? fa[kk][q] : 0.0;
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (kk0 + kk < nkk) {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[kk][tm], fb[kk][tn], acc[tm][tn], 0, 0, 0);
            }
        }
    }

    double* __restrict__ Cg = Lsx + pb.c_off;
    const int64_t ldc = pb.ldc;
    const int32_t* __restrict__ rm = RelMap + pb.map_off;
    int32_t colm[2][4];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) colm[tm][r] = rm[min(cj0 + 16 * tm + fk + 4 * r, N - 1)];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
        const int ci = ci0 + 16 * tn + fr;
        const int32_t rowm = rm[min(ci, M - 1)];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cj = cj0 + 16 * tm + fk + 4 * r;
                if (ci < M && cj < N && ci >= cj + (pb.strict & 1))
                    unsafeAtomicAdd(Cg + rowm + (int64_t)colm[tm][r] * ldc, -acc[tm][tn][r]);
            }
    }
}

void launch_update_small(const GemmProb* probs, const GemmTask* tasks, int ntasks, double* Lsx, const int32_t* RelMap, hipStream_t st) {
    if (ntasks <= 0) return;
    hipLaunchKernelGGL(k_update_small, dim3((ntasks + 3) / 4), dim3(256), 0, st, probs, tasks, ntasks, Lsx, RelMap);
}

// ---------------------------------------------------------------------------------------------------
// On-device validation (the device twin of SparseFrame_validate, C:3141-3266 / L:3702-3858): r = A x - b with the stored
// triangle(s) of P A P^T, then |r|_inf / (|A|_1 |x|_inf + |b|_inf).  One lane per column (row for U); the four maxima are
// taken with integer atomicMax on the bit patterns (non-negative doubles order like unsigned integers).
//   norms[0] = |r|_inf, [1] = |A|_1 (max column sum), [2] = |x|_inf, [3] = |b|_inf
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_resid_init(int32_t n, double* __restrict__ r, double* __restrict__ colsum, double* __restrict__ b_out) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double b = 1.0 + (double)i / (double)n;
    b_out[i] = b;
    r[i] = -b;
    colsum[i] = 0.0;
}

// sym != 0: (Lp, Li, Lx) is one triangle of a symmetric matrix, used for both (Cholesky; LU of a symmetric input);
// sym == 0: only its own entries (the L part by column of an unsymmetric matrix)
__global__ void __launch_bounds__(256)
k_resid_cols(const int64_t* __restrict__ Lp, const int32_t* __restrict__ Li, const double* __restrict__ Lx, int32_t n, int sym,
             const double* __restrict__ x, double* __restrict__ r, double* __restrict__ colsum) {
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const double xj = x[j];
    double rj = 0.0, cj = 0.0;
    for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) {
        const int32_t i = Li[p];
        const double a = Lx[p];
        unsafeAtomicAdd(r + i, a * xj);
        cj += fabs(a);
        if (sym && i != j) {
            rj += a * x[i];
            unsafeAtomicAdd(colsum + i, fabs(a));
        }
    }
    if (rj != 0.0) unsafeAtomicAdd(r + j, rj);
    unsafeAtomicAdd(colsum + j, cj);
}

// U by ROW (unsymmetric LU): row i lists columns j >= i; the diagonal is already in the L part
__global__ void __launch_bounds__(256)
k_resid_urows(const int64_t* __restrict__ Up, const int32_t* __restrict__ Ui, const double* __restrict__ Ux, int32_t n,
              const double* __restrict__ x, double* __restrict__ r, double* __restrict__ colsum) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double ri = 0.0;
    for (int64_t p = Up[i]; p < Up[i + 1]; ++p) {
        const int32_t j = Ui[p];
        if (j == i) continue;
        ri += Ux[p] * x[j];
        unsafeAtomicAdd(colsum + j, fabs(Ux[p]));
    }
    if (ri != 0.0) unsafeAtomicAdd(r + i, ri);
}

__global__ void __launch_bounds__(256)
k_resid_norms(int32_t n, const double* __restrict__ r, const double* __restrict__ colsum, const double* __restrict__ x,
              const double* __restrict__ b, unsigned long long* __restrict__ norms) {
    double m[4] = {0.0, 0.0, 0.0, 0.0};
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        m[0] = fmax(m[0], fabs(r[i])); m[1] = fmax(m[1], colsum[i]); m[2] = fmax(m[2], fabs(x[i])); m[3] = fmax(m[3], fabs(b[i]));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double v = m[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(norms + k, (unsigned long long)__double_as_longlong(v));
    }
}

__global__ void k_noop() {}
// one empty launch: loads this library's code object onto the current device (the first launch of a process pays for that)
void launch_noop(hipStream_t st) { hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, st); }

void launch_residual(const int64_t* Lp, const int32_t* Li, const double* Lx, const int64_t* Up, const int32_t* Ui, const double* Ux,
                     int32_t n, const double* x, double* r, double* colsum, double* b, double* norms, hipStream_t st) {
    if (n <= 0) return;
    const int g = (n + 255) / 256;
    hipLaunchKernelGGL(k_resid_init, dim3(g), dim3(256), 0, st, n, r, colsum, b);
    hipLaunchKernelGGL(k_resid_cols, dim3(g), dim3(256), 0, st, Lp, Li, Lx, n, Up ? 0 : 1, x, r, colsum);
    if (Up) hipLaunchKernelGGL(k_resid_urows, dim3(g), dim3(256), 0, st, Up, Ui, Ux, n, x, r, colsum);
    hipLaunchKernelGGL(k_resid_norms, dim3(g < 1024 ? g : 1024), dim3(256), 0, st, n, r, colsum, x, b, (unsigned long long*)norms);
}

// relative map of every scatter problem: one workgroup per problem, lanes stride over its M source rows
__global__ void __launch_bounds__(256)
k_build_relmaps(const GemmProb* __restrict__ probs, const int32_t* __restrict__ Lsi, int32_t* __restrict__ RelMap) {
    const GemmProb pb = probs[blockIdx.x];
    for (int ci = threadIdx.x; ci < pb.M; ci += blockDim.x) {
        const int32_t g = Lsi[pb.src_rows + ci];
        RelMap[pb.map_off + ci] = (ci < pb.N) ? (g - pb.tgt_first_col)
                                              : pb.tgt_nscol + lower_bound_i32(Lsi + pb.tgt_rows, pb.tgt_nbelow, g);
    }
}

void launch_build_relmaps(const GemmProb* probs, int nprobs, const int32_t* Lsi, int32_t* RelMap, hipStream_t st) {
    if (nprobs > 0) hipLaunchKernelGGL(k_build_relmaps, dim3(nprobs), dim3(256), 0, st, probs, Lsi, RelMap);
}

void launch_gemm(const GemmProb* probs, const GemmTask* tasks, const uint32_t* kt_prefix, int ntasks, uint32_t u_lo, uint32_t u_hi,
                 int mode, double* Lsx, const int32_t* RelMap, int* ticket, hipStream_t st, int whole_tiles, int grid_cap) {
    if (ntasks <= 0 || u_hi <= u_lo) return;
    const uint32_t units = u_hi - u_lo;
    const uint32_t cap = (grid_cap > 0 && grid_cap < GEMM_GRID) ? (uint32_t)grid_cap : (uint32_t)GEMM_GRID;
    const uint32_t grid = units < cap ? units : cap;
    // LDS-DMA staging is the default (68.9 vs 67.4 TFLOP/s at 16k x 16k x 4k, 552 vs 554 ms at 128^3); SF_GEMM_DMA=0 selects the
    // register-staged form (read per launch: the tests flip it)
    static const uint32_t min_units = [] { const char* m = getenv("SF_GEMM_MIN_UNITS"); return (uint32_t)(m ? std::max(1, atoi(m)) : SF_GEMM_MIN_UNITS_DEFAULT); }();
    const char* e = sf_exp_env("SF_GEMM_DMA");
    const bool dma = e ? atoi(e) != 0 : true;
    if (dma) {
        if (mode == 1)
            hipLaunchKernelGGL((k_gemm<1, true>), dim3(grid), dim3(GEMM_THREADS), 0, st, probs, tasks, kt_prefix, ntasks, u_lo, u_hi, Lsx, RelMap, ticket, whole_tiles, min_units);
        else
            hipLaunchKernelGGL((k_gemm<0, true>), dim3(grid), dim3(GEMM_THREADS), 0, st, probs, tasks, kt_prefix, ntasks, u_lo, u_hi, Lsx, RelMap, ticket, whole_tiles, min_units);
        return;
    }
    if (mode == 1)
        hipLaunchKernelGGL((k_gemm<1, false>), dim3(grid), dim3(GEMM_THREADS), 0, st, probs, tasks, kt_prefix, ntasks, u_lo, u_hi, Lsx, RelMap, ticket, whole_tiles, min_units);
    else
        hipLaunchKernelGGL((k_gemm<0, false>), dim3(grid), dim3(GEMM_THREADS), 0, st, probs, tasks, kt_prefix, ntasks, u_lo, u_hi, Lsx, RelMap, ticket, whole_tiles, min_units);
}

}  // namespace sf

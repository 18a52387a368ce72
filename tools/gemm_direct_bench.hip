// Experiment: an LDS-free, barrier-free fp64 MFMA GEMM -- every wave loads its own MFMA fragments straight from global memory
// (16 consecutive rows x 4 k per quarter-wave: 128-byte runs), software-pipelined DEPTH k-groups ahead, no workgroup
// synchronisation at all.  Question: does the vector L1 / L2 carry the ~3x operand re-reads well enough to beat the LDS-staged
// k_gemm (68-69 TFLOP/s at 16k x 16k x 4k)?   C = Y X^T on rows of one column-major panel, full rectangle (no triangle).
// build: hipcc -O3 --offload-arch=gfx950 tools/gemm_direct_bench.hip -o /tmp/gemm_direct   run: /tmp/gemm_direct M N K
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int DEPTH, int WCI, int WCJ, bool ATOMIC>      // wave tile = (16 WCI) x (16 WCJ)
__global__ void __launch_bounds__(512)
k_direct(const double* __restrict__ P, int64_t lda, int M, int N, int K, double* __restrict__ C, int64_t ldc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave & 1, wn = wave >> 1;                  // 2 x 4 waves
    const int fr = lane & 15, fk = lane >> 4;
    // XCD-friendly tile order: blockIdx -> (tm, tn) in 8 x 8 supertiles
    const int tiles_m = M / (32 * WCI), tiles_n = N / (64 * WCJ);
    for (int b = blockIdx.x; b < tiles_m * tiles_n; b += gridDim.x) {      // persistent when the grid is smaller than the tile count
    int tm, tn;
    if (tiles_n % 8 == 0 && tiles_m % 8 == 0) {
        const int st = b / 64, in = b % 64;
        const int stn = tiles_n / 8;
        tm = (st / stn) * 8 + (in % 8); tn = (st % stn) * 8 + (in / 8);
    } else {
        tn = b % tiles_n; tm = b / tiles_n;      // the tile columns of one tile row next to each other
    }
    if (tm >= tiles_m || tn >= tiles_n) continue;
    const int ci0 = tm * 32 * WCI + wm * 16 * WCI, cj0 = tn * 64 * WCJ + wn * 16 * WCJ;
    const double* yq = P + ci0 + fr + (int64_t)fk * lda;
    const double* xq = P + cj0 + fr + (int64_t)fk * lda;
    double4_t acc[WCJ][WCI];
#pragma unroll
    for (int a = 0; a < WCJ; ++a)
#pragma unroll
        for (int c = 0; c < WCI; ++c) acc[a][c] = (double4_t){0, 0, 0, 0};
    double fa[DEPTH][WCJ], fb[DEPTH][WCI];
    const int nkk = K / 4;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const int64_t off = (int64_t)(4 * d) * lda;
#pragma unroll
        for (int t = 0; t < WCJ; ++t) fa[d][t] = xq[off + 16 * t];
#pragma unroll
        for (int t = 0; t < WCI; ++t) fb[d][t] = yq[off + 16 * t];
    }
    for (int kk0 = 0; kk0 < nkk; kk0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            double a[WCJ], bb[WCI];
#pragma unroll
            for (int t = 0; t < WCJ; ++t) a[t] = fa[d][t];
#pragma unroll
            for (int t = 0; t < WCI; ++t) bb[t] = fb[d][t];
            const int kn = min(kk0 + d + DEPTH, nkk - 1);
            const int64_t off = (int64_t)(4 * kn) * lda;
#pragma unroll
            for (int t = 0; t < WCJ; ++t) fa[d][t] = xq[off + 16 * t];
#pragma unroll
            for (int t = 0; t < WCI; ++t) fb[d][t] = yq[off + 16 * t];
#pragma unroll
            for (int tmm = 0; tmm < WCJ; ++tmm)
#pragma unroll
                for (int tnn = 0; tnn < WCI; ++tnn)
                    acc[tmm][tnn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tmm], bb[tnn], acc[tmm][tnn], 0, 0, 0);
        }
    }
#pragma unroll
    for (int tmm = 0; tmm < WCJ; ++tmm)
#pragma unroll
        for (int tnn = 0; tnn < WCI; ++tnn)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (ATOMIC) unsafeAtomicAdd(&C[(ci0 + 16 * tnn + fr) + (int64_t)(cj0 + 16 * tmm + fk + 4 * r) * ldc], -acc[tmm][tnn][r]);
                else C[(ci0 + 16 * tnn + fr) + (int64_t)(cj0 + 16 * tmm + fk + 4 * r) * ldc] = acc[tmm][tnn][r];
    }
}

// library-like variant: lower trapezoid only (quadrants above the diagonal skipped), scatter through a relative map with fp64
// atomics (the map is the identity here, but it is read per lane from memory like the real one)
template <int DEPTH>
__global__ void __launch_bounds__(512)
k_direct_lib(const double* __restrict__ P, int64_t lda, int M, int N, int K, double* __restrict__ C, int64_t ldc,
             const int* __restrict__ rmap) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = wave >> 2, w4 = wave & 3;
    const int wm = w4 & 1, wn = w4 >> 1;
    const int fr = lane & 15, fk = lane >> 4;
    const int tiles_m = M / 128, tiles_n = N / 128;
    // workgroup b = tiles 2b, 2b + 1 in supertile order (8 x 8 tiles, tm fastest): the two halves are vertical neighbours
    const int t = 2 * blockIdx.x + half;
    const int st = t / 64, in = t % 64;
    const int stn = tiles_n / 8;
    const int tm = (st / stn) * 8 + (in % 8), tn = (st % stn) * 8 + (in / 8);
    if (tm >= tiles_m || tn >= tiles_n) return;
    const int qci0 = tm * 128 + wm * 64, qcj0 = tn * 128 + wn * 64;
    if (qci0 + 63 < qcj0) return;
    const double* yb = P + qci0 + fr + (int64_t)fk * lda;
    const double* xb = P + qcj0 + fr + (int64_t)fk * lda;
    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = (double4_t){0, 0, 0, 0};
    double fa[DEPTH][4], fb[DEPTH][4];
    const int nkk = K / 4;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const int64_t off = (int64_t)(4 * d) * lda;
#pragma unroll
        for (int q = 0; q < 4; ++q) { fa[d][q] = xb[off + 16 * q]; fb[d][q] = yb[off + 16 * q]; }
    }
    for (int kk0 = 0; kk0 < nkk; kk0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            double a[4], bb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { a[q] = fa[d][q]; bb[q] = fb[d][q]; }
            const int64_t off = (int64_t)(4 * min(kk0 + d + DEPTH, nkk - 1)) * lda;
#pragma unroll
            for (int q = 0; q < 4; ++q) { fa[d][q] = xb[off + 16 * q]; fb[d][q] = yb[off + 16 * q]; }
#pragma unroll
            for (int tmm = 0; tmm < 4; ++tmm)
#pragma unroll
                for (int tnn = 0; tnn < 4; ++tnn)
                    acc[tmm][tnn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tmm], bb[tnn], acc[tmm][tnn], 0, 0, 0);
        }
    }
    int rowm[4];
#pragma unroll
    for (int tnn = 0; tnn < 4; ++tnn) rowm[tnn] = rmap[qci0 + 16 * tnn + fr];
#pragma unroll
    for (int tmm = 0; tmm < 4; ++tmm) {
        int colm[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) colm[r] = rmap[qcj0 + 16 * tmm + fk + 4 * r];
#pragma unroll
        for (int tnn = 0; tnn < 4; ++tnn) {
            const int ci = qci0 + 16 * tnn + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cj = qcj0 + 16 * tmm + fk + 4 * r;
                if (ci >= cj) unsafeAtomicAdd(&C[rowm[tnn] + (int64_t)colm[r] * ldc], -acc[tmm][tnn][r]);
            }
        }
    }
}

template <int DEPTH>
int run_lib(const double* d, int M, int N, int K, double* c, const int* rmap, int reps) {
    const int tiles = (M / 128) * (N / 128);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_direct_lib<DEPTH>), dim3(tiles / 2), dim3(512), 0, 0, d, (int64_t)M, M, N, K, c, (int64_t)M, rmap);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) best = ms < best ? ms : best;
    }
    const double alg = (double)N * (N + 1) * K + 2.0 * (double)(M - N) * N * K;
    printf("library-like (trapezoid, mapped atomic scatter) depth %d: M=%d N=%d K=%d  %.3f ms  algorithmic %.2f TFLOP/s\n", DEPTH, M, N, K, best,
           alg / best / 1e9);
    return 0;
}

template <int DEPTH, int WCI, int WCJ, bool ATOMIC>
int run(const double* d, int M, int N, int K, double* c, int reps, int pgrid) {
    const int tiles_m = M / (32 * WCI), tiles_n = N / (64 * WCJ);
    const int grid = pgrid > 0 ? pgrid : tiles_m * tiles_n;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_direct<DEPTH, WCI, WCJ, ATOMIC>), dim3(grid), dim3(512), 0, 0, d, (int64_t)M, M, N, K, c, (int64_t)M);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) best = ms < best ? ms : best;
    }
    printf("direct depth %d wave tile %dx%d (workgroup %dx%d) grid %d %s: M=%d N=%d K=%d  %.3f ms  %.2f TFLOP/s\n", DEPTH, 16 * WCI, 16 * WCJ,
           32 * WCI, 64 * WCJ, grid, ATOMIC ? "atomic" : "store", M, N, K, best, 2.0 * M * N * K / best / 1e9);
    return 0;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 16384, K = argc > 3 ? atoi(argv[3]) : 4096;
    double *d, *c;
    const size_t elems = (size_t)M * K;
    CK(hipMalloc(&d, (elems + 64) * sizeof(double)));
    CK(hipMalloc(&c, (size_t)M * N * sizeof(double)));
    std::vector<double> h((size_t)M * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((double)rand() / RAND_MAX - 0.5) * 1e-3;
    for (size_t off = 0; off < elems; off += h.size())
        CK(hipMemcpy(d + off, h.data(), (off + h.size() <= elems ? h.size() : elems - off) * sizeof(double), hipMemcpyHostToDevice));
    {
        std::vector<int> hm(M);
        for (int i = 0; i < M; ++i) hm[i] = i;
        int* rmap;
        CK(hipMalloc(&rmap, M * sizeof(int)));
        CK(hipMemcpy(rmap, hm.data(), M * sizeof(int), hipMemcpyHostToDevice));
        if (run_lib<2>(d, M, N, K, c, rmap, 4)) return 1;
        if (run_lib<3>(d, M, N, K, c, rmap, 4)) return 1;
    }
    if (run<2, 4, 4, false>(d, M, N, K, c, 4, 0)) return 1;
    if (run<2, 4, 4, false>(d, M, N, K, c, 4, 256)) return 1;
    if (run<2, 4, 4, false>(d, M, N, K, c, 4, 512)) return 1;
    if (run<2, 4, 4, true>(d, M, N, K, c, 4, 0)) return 1;
    if (run<2, 4, 4, true>(d, M, N, K, c, 4, 256)) return 1;
    if (run<4, 4, 2, false>(d, M, N, K, c, 4, 0)) return 1;
    if (run<4, 4, 2, false>(d, M, N, K, c, 4, 512)) return 1;
    return 0;
}

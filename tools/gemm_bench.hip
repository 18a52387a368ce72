// Stand-alone timing of the fp64 MFMA GEMM kernel (k_gemm<0>, in-panel mode) on one large synthetic problem.
// Build on the GPU box (links the in-tree library):
//   hipcc -O3 --offload-arch=gfx950 -Isparse-matrix-factorization-library_amd/csrc -Iinclude tools/gemm_bench.hip \
//         -Lsparse-matrix-factorization-library_amd -lsparseframe_hip -Wl,-rpath,$PWD/sparse-matrix-factorization-library_amd -o /tmp/gemm_bench
//   /tmp/gemm_bench M N K
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "sf_kernels.h"

namespace sf { void exp_set_stamps(unsigned long long* p); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 8192;
    const int reps = argc > 4 ? atoi(argv[4]) : 5;
    const int force_atomic = argc > 5 ? atoi(argv[5]) : 0;
    const int64_t ld = M;
    const size_t elems = (size_t)ld * (K + N);
    double* d;
    CK(hipMalloc(&d, (elems + 2) * sizeof(double)));
    std::vector<double> h((size_t)ld * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((double)rand() / RAND_MAX - 0.5) * 1e-3;
    for (size_t off = 0; off < elems; off += h.size())
        CK(hipMemcpy(d + off, h.data(), std::min(h.size(), elems - off) * sizeof(double), hipMemcpyHostToDevice));

    sf::GemmProb pb{};
    pb.y_off = 0; pb.x_off = 0; pb.c_off = (int64_t)K * ld;
    pb.lda = M; pb.ldc = M; pb.M = M; pb.N = N; pb.K = K;
    pb.strict = force_atomic ? 2 : 0;   // 5th argument != 0: force the atomic epilogue (cost of the fused scatter)
    std::vector<sf::GemmTask> tasks;
    const int tmn = (M + sf::GEMM_BM - 1) / sf::GEMM_BM, tnn = (N + sf::GEMM_BN - 1) / sf::GEMM_BN;
    const int sw = std::min(tnn, 8), sh = std::max(1, 64 / sw);
    const int nkt_all = (K + sf::GEMM_BK - 1) / sf::GEMM_BK;
    for (int k0 = 0; k0 < nkt_all; k0 += sf::GEMM_SLICE)
    for (int sj = 0; sj < tnn; sj += sw)
        for (int si = 0; si < tmn; si += sh)
            for (int tn = sj; tn < std::min(sj + sw, tnn); ++tn)
                for (int tm = si; tm < std::min(si + sh, tmn); ++tm) {
                    if ((tm + 1) * sf::GEMM_BM - 1 < tn * sf::GEMM_BN) continue;
                    tasks.push_back(sf::GemmTask{0, (uint16_t)tm, (uint16_t)tn, (uint32_t)k0, (uint32_t)std::min(sf::GEMM_SLICE, nkt_all - k0)});
                }
    std::vector<uint32_t> pre(tasks.size() + 1);
    pre[0] = 0;
    for (size_t i = 0; i < tasks.size(); ++i) pre[i + 1] = pre[i] + tasks[i].nkt;
    uint32_t* dpre;
    CK(hipMalloc(&dpre, pre.size() * sizeof(uint32_t)));
    CK(hipMemcpy(dpre, pre.data(), pre.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    sf::GemmProb* dp; sf::GemmTask* dt;
    CK(hipMalloc(&dp, sizeof(pb))); CK(hipMalloc(&dt, tasks.size() * sizeof(sf::GemmTask)));
    CK(hipMemcpy(dp, &pb, sizeof(pb), hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, tasks.data(), tasks.size() * sizeof(sf::GemmTask), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double alg = (double)N * (N + 1) * K + 2.0 * (double)(M - N) * N * K;
    const double exec = (double)pre.back() * 128.0 * 128.0 * 2.0 * sf::GEMM_BK;
    float best = 1e30f;
    int* dticket;                                   // 6th argument 0: static deal of the rounds (default: dynamic, as in the library)
    CK(hipMalloc(&dticket, 8 * sizeof(int)));
    const int dynamic = argc > 6 ? atoi(argv[6]) : 1;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemset(dticket, 0, 8 * sizeof(int)));
        CK(hipEventRecord(e0));
        sf::launch_gemm(dp, dt, dpre, (int)tasks.size(), 0, pre.back(), 0, d, nullptr, dynamic ? dticket : nullptr, 0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 || reps == 1) best = std::min(best, ms);
    }
#ifdef SF_EXP_TIMING
    {
        // per-tile stamps of one more launch: [0] tile start (claim, task / problem fetch, first tile load), [1] main loop starts,
        // [2] main loop done, [3] epilogue + closing barrier done
        const int G = sf::GEMM_GRID;
        unsigned long long* ds;
        CK(hipMalloc(&ds, (size_t)G * 65 * 4 * 8));
        CK(hipMemset(ds, 0, (size_t)G * 65 * 4 * 8));
        sf::exp_set_stamps(ds);
        CK(hipMemset(dticket, 0, 8 * sizeof(int)));
        sf::launch_gemm(dp, dt, dpre, (int)tasks.size(), 0, pre.back(), 0, d, nullptr, dynamic ? dticket : nullptr, 0);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hs((size_t)G * 65 * 4);
        CK(hipMemcpy(hs.data(), ds, hs.size() * 8, hipMemcpyDeviceToHost));
        double pro = 0, loop = 0, epi = 0; long cnt = 0;
        unsigned long long tmin = ~0ull, tmax = 0, first_end_min = ~0ull, last_start_max = 0;
        for (int g = 0; g < G; ++g) {
            unsigned long long wg_end = 0;
            for (int t = 0; t < 64; ++t) {
                const unsigned long long* q = &hs[((size_t)g * 64 + t) * 4];
                if (!q[3]) break;
                pro += (double)(q[1] - q[0]); loop += (double)(q[2] - q[1]); epi += (double)(q[3] - q[2]); ++cnt;
                tmin = std::min(tmin, q[0]); tmax = std::max(tmax, q[3]); wg_end = q[3];
            }
            if (wg_end) { first_end_min = std::min(first_end_min, wg_end); last_start_max = std::max(last_start_max, wg_end); }
        }
        // per workgroup: busy ticks / real time, first start and last end on the chip-wide 100 MHz clock
        unsigned long long r0 = ~0ull, r1 = 0, rs_max = 0, re_min = ~0ull; double rate = 0, busy = 0;
        for (int g = 0; g < G; ++g) {
            const unsigned long long* w = &hs[(size_t)G * 256 + g * 4];
            r0 = std::min(r0, w[1]); r1 = std::max(r1, w[3]); rs_max = std::max(rs_max, w[1]); re_min = std::min(re_min, w[3]);
            rate += (double)(w[2] - w[0]) / (double)(w[3] - w[1]); busy += (double)(w[3] - w[1]);
        }
        printf("  workgroups: shader ticks per 100 MHz tick %.2f; first start .. last start %.1f us; first end .. last end %.1f us; "
               "kernel %.1f us; mean workgroup life %.1f us\n", rate / G, (rs_max - r0) / 100.0, (r1 - re_min) / 100.0, (r1 - r0) / 100.0, busy / G / 100.0);
        const double span = (double)(tmax - tmin);
        printf("  stamps: %ld tiles, span %.0f ticks; per tile: prologue %.0f, main loop %.0f (%.1f per K step), epilogue %.0f ticks; "
               "workgroups end between %.3f and 1.000 of the span; ticks per ms %.0f\n", cnt, span, pro / cnt, loop / cnt,
               loop / cnt / ((K + 15) / 16), epi / cnt, (double)(first_end_min - tmin) / span, span / best);
    }
#endif
    printf("gemm<0> atomic=%d M=%d N=%d K=%d tiles=%zu  %.3f ms  algorithmic %.2f TFLOP/s  (tile-executed %.2f TFLOP/s)\n", force_atomic, M, N, K,
           tasks.size(), best, alg / best / 1e9, exec / best / 1e9);
    return 0;
}

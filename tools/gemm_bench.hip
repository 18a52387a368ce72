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
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        sf::launch_gemm(dp, dt, dpre, (int)tasks.size(), 0, pre.back(), 0, d, nullptr, 0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 || reps == 1) best = std::min(best, ms);
    }
    printf("gemm<0> atomic=%d M=%d N=%d K=%d tiles=%zu  %.3f ms  algorithmic %.2f TFLOP/s  (tile-executed %.2f TFLOP/s)\n", force_atomic, M, N, K,
           tasks.size(), best, alg / best / 1e9, exec / best / 1e9);
    return 0;
}

// Where the DIAGONAL workgroup of a fused 64-column step (k_step) spends its time: 100 MHz stamps written by the kernel when it is
// compiled with -DSF_EXP_STEP_STAMPS (tools/experiments/step_stamps.sh compiles this file together with csrc/sf_kernels.hip).
// One diagonal task + `ntiles` row tiles, step `ti` of its outer block; Cholesky and LU (tol 0.1, natural pivots pass).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "sf_kernels.h"
namespace sf { void exp_set_step_stamps(unsigned long long* p); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const int nscol = 512;
    const int64_t nsrow = 64 * 512 + 512;
    const size_t pan = (size_t)nsrow * nscol;
    double* d;
    CK(hipMalloc(&d, (2 * pan + 2) * sizeof(double)));        // [L panel | U^T panel]
    {
        std::vector<double> h((size_t)nsrow);
        for (int p = 0; p < 2; ++p)
            for (int c = 0; c < nscol; ++c) {
                for (int64_t r = 0; r < nsrow; ++r) h[r] = (r == c) ? 8.0 : ((double)rand() / RAND_MAX - 0.5) * 1e-2;
                CK(hipMemcpy(d + p * pan + (size_t)c * nsrow, h.data(), nsrow * sizeof(double), hipMemcpyHostToDevice));
            }
    }
    int *info, *flags, *tickets, *piv;
    CK(hipMalloc(&info, sizeof(int))); CK(hipMemset(info, 0, sizeof(int)));
    CK(hipMalloc(&flags, sizeof(int))); CK(hipMemset(flags, 0, sizeof(int)));
    CK(hipMalloc(&tickets, 64 * sizeof(int)));
    CK(hipMalloc(&piv, (2 * 40000 + 1) * sizeof(int)));
    double* tinv; CK(hipMalloc(&tinv, 2048 * sizeof(double)));
    sf::StepTask* dt; CK(hipMalloc(&dt, 1024 * sizeof(sf::StepTask)));
    unsigned long long* st; CK(hipMalloc(&st, 16 * sizeof(unsigned long long)));
    sf::exp_set_step_stamps(st);
    int epoch = 0;
    const char* names[9] = {"entry->task", "update+tile", "panel 0", "panel 1", "panel 2", "panel 3", "moves (LU)", "inverses", "publish"};
    for (int lu = 0; lu < 3; ++lu)          // 0 Cholesky, 1 LU with threshold pivoting (tol 0.1), 2 LU without (tol 0: the reference's behaviour)
        for (int ntiles : {0, 8, 200})
            for (int ti : {0, 7}) {
                std::vector<sf::StepTask> t;
                const int diag = 64 * ti;
                // Cholesky diagonal tasks come up to date (J == diag); LU ones update themselves (J = 0)
                t.push_back(sf::StepTask{0, lu ? (int64_t)pan : 0, (int32_t)nsrow, lu ? 0 : diag, diag, 64, diag, 64, 0, 0, 0, 0, 0, 0});
                for (int k = 0; k < ntiles; ++k) {
                    t.push_back(sf::StepTask{0, lu ? (int64_t)pan : 0, (int32_t)nsrow, 0, diag, 64, 512 + 64 * k, 64, 0, 0, 0, 0, 0, 0});
                    if (lu) t.push_back(sf::StepTask{(int64_t)pan, 0, (int32_t)nsrow, 0, diag, 64, 512 + 64 * k, 64, 0, 1, 0, 0, 0, 0});
                }
                CK(hipMemcpy(dt, t.data(), t.size() * sizeof(sf::StepTask), hipMemcpyHostToDevice));
                sf::PivotCtl pc{lu == 1 ? 0.1 : 0.0, lu == 1 ? 1e-12 : 0.0, lu == 1 ? piv : nullptr, lu == 1 ? piv + 40000 : nullptr, piv + 80000};
                double acc[9] = {0};
                const int reps = 10;
                for (int r = 0; r < reps + 1; ++r) {
                    CK(hipMemset(tickets, 0, 64 * sizeof(int)));
                    sf::launch_step(dt, (int)t.size(), lu ? 1 : 0, d, flags, ++epoch, info, tinv, tickets, pc, 0);
                    CK(hipDeviceSynchronize());
                    unsigned long long h[16];
                    CK(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
                    if (r == 0) continue;
                    // slots: 0 entry, 1 task, 2 tile ready, 3..5 panels 0..2 (+trailing), 6 all panels, 7 moves (LU), 8 inverses, 9 published
                    if (!lu) h[7] = h[6];
                    const unsigned long long seq[10] = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9]};
                    for (int k = 0; k < 9; ++k) acc[k] += (double)(seq[k + 1] - seq[k]) * 0.01;     // 100 MHz -> us
                }
                double tot = 0;
                for (int k = 0; k < 9; ++k) tot += acc[k] / reps;
                printf("%s ntiles %3d ti %d: total %.2f us |", lu == 0 ? "Chol" : (lu == 1 ? "LU  " : "LU0 "), ntiles, ti, tot);
                for (int k = 0; k < 9; ++k) printf(" %s %.2f", names[k], acc[k] / reps);
                printf("\n");
            }
    return 0;
}

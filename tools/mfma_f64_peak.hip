// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on the whole chip with register-only operands.
// Build+run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Prints TFLOP/s for several operand values / occupancies and the f64 vector-FMA rate for comparison.
// (__launch_bounds__(256, 2) keeps the accumulators in VGPRs: with (256) alone hipcc parks them in AGPRs and
//  copies all 128 registers in and out every iteration, which halves the measured rate.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256, 2) k_mfma(double* out, int iters, double a0, double b0) {
    double4_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_fma(double* out, int iters, double a0, double b0) {
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = a0 + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fma(x[i], b0, a0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    double* out;
    CK(hipMalloc(&out, sizeof(double) * 256 * 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    struct Case { const char* name; double a, b; int grid; };
    const Case cases[] = {
        {"all CUs, 1 wave/SIMD, operands ~1.0 x 0.5", 1.0, 0.5, 256},
        {"all CUs, 2 waves/SIMD, operands ~1.0 x 0.5", 1.0, 0.5, 512},
        {"all CUs, 1 wave/SIMD, operands 0 x 0 (+1e-9 lane noise)", 0.0, 0.0, 256},
        {"all CUs, 1 wave/SIMD, operands pi x e", 3.14159265358979, 2.718281828459045, 256},
        {"64 workgroups only (quarter of the chip)", 1.0, 0.5, 64},
        {"8 workgroups only", 1.0, 0.5, 8},
    };
    for (const Case& c : cases) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_mfma<16>, dim3(c.grid), dim3(256), 0, 0, out, iters, c.a, c.b);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double flops = (double)c.grid * 4 * iters * 16 * 2048.0;
        const double waves_per_simd = c.grid > 256 ? c.grid / 256.0 : 1.0;
        printf("mfma_f64_16x16x4 | %-55s | %7.2f TFLOP/s | %.3f ms | %.1f cycles/MFMA/SIMD at 2.4 GHz\n", c.name,
               flops / best / 1e9, best, best * 1e-3 * 2.4e9 / ((double)iters * 16 * waves_per_simd));
    }
    {   // sustained: ~1.5 s of back-to-back launches (power / clock management has time to react)
        const int grid = 512, n_launch = 450;
        for (int phase = 0; phase < 3; ++phase) {
            CK(hipEventRecord(e0));
            for (int l = 0; l < n_launch; ++l)
                hipLaunchKernelGGL(k_mfma<16>, dim3(grid), dim3(256), 0, 0, out, iters, 3.14159265358979, 2.718281828459045);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("mfma_f64_16x16x4 | sustained, %d back-to-back launches, 2 waves/SIMD (phase %d) | %7.2f TFLOP/s | %.1f ms\n",
                   n_launch, phase, (double)n_launch * grid * 4 * iters * 16 * 2048.0 / ms / 1e9, ms);
        }
    }
    {
        float best = 1e30f;
        const int grid = 256 * 8;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 0.5);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("v_fma_f64 (8 waves/SIMD, all CUs): %.2f TFLOP/s (%.3f ms)\n", (double)grid * 256 * iters * 16 * 2.0 / best / 1e9, best);
    }
    return 0;
}

// Accuracy of v_rsq_f64 / v_rcp_f64 followed by one or two Newton steps, against correctly rounded results (long double on the host).
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/rsq_accuracy.hip -o /tmp/rsq_acc && /tmp/rsq_acc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double* x, double* o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r0 = __builtin_amdgcn_rsq(v);
    double r1 = r0 * (1.5 - 0.5 * v * r0 * r0);
    double r2 = r1 * (1.5 - 0.5 * v * r1 * r1);
    // one step in the fma form: e = 1 - v r0^2 (one rounding), r = r0 + r0 * (e / 2)
    double e = __builtin_fma(-v * r0, r0, 1.0);
    double r1f = __builtin_fma(r0 * 0.5, e, r0);
    double c0 = __builtin_amdgcn_rcp(v);
    double c1 = c0 * (2.0 - v * c0);
    double c2 = c1 * (2.0 - v * c1);
    double ce = __builtin_fma(-v, c0, 1.0);
    double c1f = __builtin_fma(c0, ce, c0);
    // third-order steps: rsqrt  r (1 + e/2 + 3 e^2 / 8), e = 1 - v r^2;  reciprocal  c (1 + e + e^2), e = 1 - v c
    {
        const double t = v * r0, e3 = __builtin_fma(-t, r0, 1.0), p3 = __builtin_fma(e3, 0.375, 0.5), q3 = e3 * p3;
        r1f = __builtin_fma(r0, q3, r0);
        const double ec = __builtin_fma(-v, c0, 1.0), pc = __builtin_fma(ec, ec, ec);
        c1f = __builtin_fma(c0, pc, c0);
    }
    o[8 * i + 0] = r0; o[8 * i + 1] = r1; o[8 * i + 2] = r2; o[8 * i + 3] = r1f;
    o[8 * i + 4] = c0; o[8 * i + 5] = c1; o[8 * i + 6] = c2; o[8 * i + 7] = c1f;
}

int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), o(8 * (size_t)n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;         // [0, 1)
        x[i] = std::ldexp(1.0 + u, (int)(s % 41) - 20);                  // 2^-20 .. 2^21
    }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 8 * (size_t)n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 8 * (size_t)n * 8, hipMemcpyDeviceToHost);
    const char* names[8] = {"rsq", "rsq + 1 Newton", "rsq + 2 Newton", "rsq + 1 third-order step", "rcp", "rcp + 1 Newton", "rcp + 2 Newton", "rcp + 1 third-order step"};
    double worst[8] = {0};
    for (int i = 0; i < n; ++i) {
        const long double ex_r = 1.0L / sqrtl((long double)x[i]), ex_c = 1.0L / (long double)x[i];
        for (int q = 0; q < 8; ++q) {
            const long double ex = q < 4 ? ex_r : ex_c;
            const double err = (double)fabsl(((long double)o[8 * (size_t)i + q] - ex) / ex);
            if (err > worst[q]) worst[q] = err;
        }
    }
    for (int q = 0; q < 8; ++q) printf("%-26s max relative error %.3e (%.2f ulp of 2^-53)\n", names[q], worst[q], worst[q] / 1.1102230246251565e-16);
    return 0;
}

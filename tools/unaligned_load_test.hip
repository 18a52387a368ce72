// Does gfx950 service 16-byte global loads (global_load_dwordx4) from addresses that are only 8-byte aligned?
// (ROCm sets SH_MEM_CONFIG.ALIGNMENT_MODE = unaligned; this checks it on the box before the GEMM staging relies on it.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double2_t __attribute__((ext_vector_type(2)));
__global__ void k(const double* __restrict__ p, double* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n) {
        double2_t v;
        // force a single 16-byte load from an address that is 8 (not 16) byte aligned when p is offset by one double
        asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p + 2 * i) : "memory");
        out[2 * i] = v.x; out[2 * i + 1] = v.y;
    }
}
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n + 2);
    for (int i = 0; i < n + 2; ++i) h[i] = i * 0.5;
    double *d, *o;
    if (hipMalloc(&d, (n + 2) * 8) != hipSuccess || hipMalloc(&o, n * 8) != hipSuccess) return 1;
    hipMemcpy(d, h.data(), (n + 2) * 8, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 2; ++shift) {
        hipMemset(o, 0, n * 8);
        hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, d + shift, o, n);
        hipError_t e = hipDeviceSynchronize();
        std::vector<double> r(n);
        hipMemcpy(r.data(), o, n * 8, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n; ++i) bad += (r[i] != h[i + shift]);
        printf("shift %d doubles (address %% 16 = %d): %s, mismatches %ld\n", shift, shift * 8, hipGetErrorString(e), bad);
    }
    return 0;
}

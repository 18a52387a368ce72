// Calibration of rocprofv3 FETCH_SIZE for THIS code's access pattern: coalesced 8-byte-per-lane streaming reads
// (global_load_dwordx2), as k_gemm's operand staging issues them.  Reads 4 GiB once; run under
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- /tmp/fetch_calib
// and compare the counter (KiB) with 4,194,304 KiB.  (MI355X_MICROARCH.md: 16-byte-per-lane streams read exactly 1/2.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k_read8(const double* __restrict__ p, size_t n, double* out) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 123.456) out[0] = s;
}
int main() {
    const size_t n = (size_t)1 << 29;
    double *p, *o;
    if (hipMalloc(&p, n * 8) != hipSuccess || hipMalloc(&o, 8) != hipSuccess) return 1;
    if (hipMemset(p, 0, n * 8) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_read8, dim3(2048), dim3(256), 0, 0, p, n, o);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    printf("read %zu bytes\n", n * 8);
    return 0;
}

// Where do the four waves of a 256-thread workgroup land?  If wave 0 of every workgroup sits on the same SIMD of its CU, the one-wave
// panel factorizations of co-resident k_step diagonal tasks time-share ONE SIMD while three idle.
//   hipcc -O2 --offload-arch=gfx950 tools/experiments/simd_placement.hip -o /tmp/simd_placement && /tmp/simd_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256, 3) k(unsigned* out, int spin) {
    __shared__ double pad[5000];            // ~40 KB like k_step: 3 workgroups per CU
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) { out[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = hw; out[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = xcc; }
    pad[threadIdx.x] = spin;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    if (pad[(threadIdx.x * 7) % 5000] == -1.0) out[0] = 0;
}
int main() {
    const int G = 768;
    unsigned* d; hipMalloc(&d, G * 4 * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int simd_of_wave[4][4] = {};
    for (int b = 0; b < G; ++b)
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[2 * (b * 4 + w)];
            const int simd = (hw >> 4) & 3;         // HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13] ...
            simd_of_wave[w][simd]++;
        }
    for (int w = 0; w < 4; ++w) printf("wave %d: SIMD0 %d SIMD1 %d SIMD2 %d SIMD3 %d\n", w, simd_of_wave[w][0], simd_of_wave[w][1], simd_of_wave[w][2], simd_of_wave[w][3]);
    for (int b = 0; b < 6; ++b) printf("wg %d: hw %08x %08x %08x %08x xcc %u\n", b, h[8 * b], h[8 * b + 2], h[8 * b + 4], h[8 * b + 6], h[8 * b + 1]);
    return 0;
}

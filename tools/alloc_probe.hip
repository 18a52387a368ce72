// How long do large device allocations take in a plain HIP process (system ROCm runtime) AFTER the device has run kernels, by API?
//   hipcc -O2 --offload-arch=gfx950 tools/alloc_probe.hip -o /tmp/alloc_probe && for m in 0 1 2 3 4; do /tmp/alloc_probe $m; done
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_touch(double* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 1.0; }
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int warm = argc > 2 ? atoi(argv[2]) : 1;
    hipSetDevice(0); hipFree(nullptr);
    const size_t G = (size_t)1 << 30;
    if (warm) {     // what a plan build does before its big allocation: small allocations, uploads, a kernel
        double* q; hipMalloc((void**)&q, 64 << 20);
        std::vector<double> h(8 << 20, 1.0);
        hipMemcpy(q, h.data(), 64 << 20, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, q, (size_t)1 << 18);
        hipDeviceSynchronize();
    }
    const char* names[] = {"hipMalloc", "hipMallocAsync (pool keeps memory)", "VMM reserve+create+map", "hipExtMallocWithFlags(uncached)", "30 x hipMalloc 1 GB"};
    for (int rep = 0; rep < 3; ++rep) {
        double* p = nullptr;
        double t = now();
        hipError_t e = hipSuccess;
        std::vector<double*> parts;
        hipStream_t s = nullptr;
        if (mode == 0) e = hipMalloc((void**)&p, 30 * G);
        else if (mode == 1) {
            hipStreamCreate(&s);
            hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
            uint64_t thr = ~0ull; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
            e = hipMallocAsync((void**)&p, 30 * G, s); hipStreamSynchronize(s);
        } else if (mode == 2) {
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            size_t gran = 0; hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
            void* va = nullptr; e = hipMemAddressReserve(&va, 30 * G, gran, nullptr, 0);
            hipMemGenericAllocationHandle_t hh; if (!e) e = hipMemCreate(&hh, 30 * G, &prop, 0);
            if (!e) e = hipMemMap(va, 30 * G, 0, hh, 0);
            hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
            if (!e) e = hipMemSetAccess(va, 30 * G, &ad, 1);
            p = (double*)va;
        } else if (mode == 3) e = hipExtMallocWithFlags((void**)&p, 30 * G, hipDeviceMallocUncached);
        else { parts.assign(30, nullptr); for (auto& q : parts) e = hipMalloc((void**)&q, G); p = parts[0]; }
        const double t1 = now();
        // use it: a kernel over the first GB and one over the last (first touch on the device)
        const size_t n1 = G / 8;
        double* last = (mode == 4) ? parts[29] : p + 29 * n1;
        hipLaunchKernelGGL(k_touch, dim3((unsigned)(n1 / 256)), dim3(256), 0, 0, p, n1);
        hipLaunchKernelGGL(k_touch, dim3((unsigned)(n1 / 256)), dim3(256), 0, 0, last, n1);
        hipError_t es = hipDeviceSynchronize();
        const double t2 = now();
        hipMemset(p, 0, mode == 4 ? G : 30 * G); hipDeviceSynchronize();
        const double t3 = now();
        if (mode == 0 || mode == 3) hipFree(p);
        else if (mode == 1) { hipFreeAsync(p, s); hipStreamSynchronize(s); }
        else if (mode == 4) for (auto& q : parts) hipFree(q);
        printf("%-36s rep %d: allocate %.1f ms (%s), first kernels on it %.1f ms (%s), memset %.1f ms, free %.1f ms\n", names[mode], rep, (t1 - t) * 1e3,
               hipGetErrorString(e), (t2 - t1) * 1e3, hipGetErrorString(es), (t3 - t2) * 1e3, (now() - t3) * 1e3);
        if (mode == 2) break;       // (not unmapped here)
    }
    return 0;
}

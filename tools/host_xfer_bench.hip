// Host-boundary measurements behind the design of the overlapped factor copy-back ("Host-buffer boundary"):
//   1. hipHostRegister of a freshly malloc'ed buffer (untouched pages) and of a touched one: seconds per GiB
//   2. D2H bandwidth into registered / hipHostMalloc'ed / pageable memory
//   3. multi-threaded memcpy from pinned staging memory into fresh pageable memory (the alternative to registering)
// build: hipcc -O2 --offload-arch=gfx950 tools/host_xfer_bench.hip -o /tmp/host_xfer_bench -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const size_t gib = argc > 1 ? (size_t)atol(argv[1]) : 8;
    const size_t bytes = gib << 30;
    void* d = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 1, bytes));
    CK(hipDeviceSynchronize());
    {
        FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
        char buf[128] = {0};
        if (f) { if (fgets(buf, sizeof buf, f)) printf("THP: %s", buf); fclose(f); }
    }
    // 1. register fresh
    char* h = (char*)malloc(bytes);
    double t0 = now();
    CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
    double t1 = now();
    printf("hipHostRegister fresh malloc %zu GiB: %.3f s (%.3f s/GiB)\n", gib, t1 - t0, (t1 - t0) / gib);
    t0 = now();
    CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
    t1 = now();
    printf("D2H into registered: %.3f s = %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e9);
    t0 = now();
    CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
    t1 = now();
    printf("D2H into registered (2nd): %.3f s = %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e9);
    t0 = now();
    CK(hipHostUnregister(h));
    t1 = now();
    printf("hipHostUnregister: %.3f s\n", t1 - t0);
    t0 = now();
    CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
    t1 = now();
    printf("hipHostRegister touched %zu GiB: %.3f s (%.3f s/GiB)\n", gib, t1 - t0, (t1 - t0) / gib);
    CK(hipHostUnregister(h));
    // pageable D2H
    t0 = now();
    CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
    t1 = now();
    printf("D2H into pageable (touched): %.3f s = %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e9);
    free(h);
    // 2. hipHostMalloc
    void* hp = nullptr;
    t0 = now();
    CK(hipHostMalloc(&hp, bytes, hipHostMallocDefault));
    t1 = now();
    printf("hipHostMalloc %zu GiB: %.3f s\n", gib, t1 - t0);
    t0 = now();
    CK(hipMemcpy(hp, d, bytes, hipMemcpyDeviceToHost));
    t1 = now();
    printf("D2H into hipHostMalloc: %.3f s = %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e9);
    // 3. threaded memcpy pinned -> fresh pageable
    for (int nt : {1, 4, 8, 16}) {
        char* dst = (char*)malloc(bytes);
        t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
            th.emplace_back([=]() {
                const size_t a = bytes / nt * t, b = (t + 1 == nt) ? bytes : bytes / nt * (t + 1);
                memcpy(dst + a, (char*)hp + a, b - a);
            });
        for (auto& x : th) x.join();
        t1 = now();
        printf("memcpy pinned -> FRESH pageable, %2d threads: %.3f s = %.1f GB/s\n", nt, t1 - t0, bytes / (t1 - t0) / 1e9);
        t0 = now();
        th.clear();
        for (int t = 0; t < nt; ++t)
            th.emplace_back([=]() {
                const size_t a = bytes / nt * t, b = (t + 1 == nt) ? bytes : bytes / nt * (t + 1);
                memcpy(dst + a, (char*)hp + a, b - a);
            });
        for (auto& x : th) x.join();
        t1 = now();
        printf("memcpy pinned -> touched pageable, %2d threads: %.3f s = %.1f GB/s\n", nt, t1 - t0, bytes / (t1 - t0) / 1e9);
        free(dst);
    }
    // 4. concurrent D2H on two streams (does one stream saturate the link?)
    {
        hipStream_t s0, s1;
        CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
        t0 = now();
        CK(hipMemcpyAsync(hp, d, bytes / 2, hipMemcpyDeviceToHost, s0));
        CK(hipMemcpyAsync((char*)hp + bytes / 2, (char*)d + bytes / 2, bytes / 2, hipMemcpyDeviceToHost, s1));
        CK(hipDeviceSynchronize());
        t1 = now();
        printf("D2H two streams: %.3f s = %.1f GB/s\n", t1 - t0, bytes / (t1 - t0) / 1e9);
        // many 32 MiB chunks on one stream
        t0 = now();
        const size_t ch = 32u << 20;
        for (size_t o = 0; o < bytes; o += ch) CK(hipMemcpyAsync((char*)hp + o, (char*)d + o, ch, hipMemcpyDeviceToHost, s0));
        double tq = now();
        CK(hipDeviceSynchronize());
        t1 = now();
        printf("D2H 32 MiB chunks: enqueue %.3f s, total %.3f s = %.1f GB/s\n", tq - t0, t1 - t0, bytes / (t1 - t0) / 1e9);
    }
    CK(hipHostFree(hp));
    CK(hipFree(d));
    return 0;
}

// mb_pin.cpp -- what does page-locking the ingest blocks cost, and which way of doing it is cheapest?  (tools/README.md)
//   hipcc -O2 -o build/mb_pin tools/mb_pin.cpp -lpthread && build/mb_pin [threads] [MB per thread]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
int main(int argc, char** argv)
{
    const int T = argc > 1 ? std::atoi(argv[1]) : 32;
    const size_t MB = argc > 2 ? (size_t)std::atoi(argv[2]) : 32, n = MB << 20;
    double t = now();
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    std::printf("hip init %.1f ms\n", (now() - t) * 1e3);
    void* d = nullptr;
    CK(hipMalloc(&d, n));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    auto h2d = [&](void* p, const char* what) {
        CK(hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double t0 = now();
        for (int i = 0; i < 4; ++i) CK(hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        std::printf("   H2D from %s: %.1f GB/s\n", what, 4.0 * n / (now() - t0) / 1e9);
    };
    std::vector<void*> ptr((size_t)T);
    auto par = [&](auto fn) {
        double t0 = now();
        std::vector<std::thread> th;
        for (int i = 0; i < T; ++i) th.emplace_back([&, i] { fn(i); });
        for (auto& x : th) x.join();
        return (now() - t0) * 1e3;
    };
    for (int rep = 0; rep < 2; ++rep) {
        double ms = par([&](int i) { CK(hipHostMalloc(&ptr[i], n, hipHostMallocPortable)); });
        std::printf("A%d. %d threads x hipHostMalloc(%zu MB): %.1f ms (%.1f GB/s)\n", rep, T, MB, ms, T * (double)n / ms / 1e6);
        if (rep == 0) h2d(ptr[0], "hipHostMalloc");
        ms = par([&](int i) { CK(hipHostFree(ptr[i])); });
        std::printf("    free: %.1f ms\n", ms);
    }
    {
        double t0 = now();
        void* p;
        CK(hipHostMalloc(&p, n * T, hipHostMallocPortable));
        std::printf("B. one hipHostMalloc(%zu MB): %.1f ms\n", MB * T, (now() - t0) * 1e3);
        t0 = now();
        CK(hipHostFree(p));
        std::printf("    free: %.1f ms\n", (now() - t0) * 1e3);
    }
    for (int huge = 0; huge < 2; ++huge) {
        double ms_touch = par([&](int i) {
            ptr[i] = std::aligned_alloc(2u << 20, n);
            if (huge) madvise(ptr[i], n, MADV_HUGEPAGE);
            std::memset(ptr[i], 1, n);
        });
        double ms = par([&](int i) { CK(hipHostRegister(ptr[i], n, hipHostRegisterPortable)); });
        std::printf("C%d. %d threads x (aligned_alloc%s + touch: %.1f ms) + hipHostRegister: %.1f ms (%.1f GB/s)\n", huge, T, huge ? " + MADV_HUGEPAGE" : "", ms_touch, ms,
            T * (double)n / ms / 1e6);
        h2d(ptr[0], huge ? "registered huge pages" : "registered pages");
        ms = par([&](int i) { CK(hipHostUnregister(ptr[i])); std::free(ptr[i]); });
        std::printf("    unregister + free: %.1f ms\n", ms);
    }
    {
        // one registration of one big huge-page area
        double t0 = now();
        void* p = std::aligned_alloc(2u << 20, n * T);
        madvise(p, n * T, MADV_HUGEPAGE);
        double ms_touch = par([&](int i) { std::memset((char*)p + (size_t)i * n, 1, n); });
        double t1 = now();
        CK(hipHostRegister(p, n * T, hipHostRegisterPortable));
        std::printf("D. one area of %zu MB: alloc+touch %.1f ms (touch %.1f), one hipHostRegister %.1f ms\n", MB * T, (t1 - t0) * 1e3, ms_touch, (now() - t1) * 1e3);
        CK(hipHostUnregister(p));
        std::free(p);
    }
    {
        void* p = std::malloc(n);
        std::memset(p, 1, n);
        h2d(p, "pageable memory");
        std::free(p);
    }
    FILE* f = std::fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char buf[128] = { 0 };
    if (f) { if (std::fgets(buf, sizeof buf, f)) std::printf("THP: %s", buf); std::fclose(f); }
    return 0;
}

// How fast does pinning go, and is registered memory as good a D2H destination as hipHostMalloc memory?  (The first lgmi_run
// of a context pinned ~400 MB of result buffers with hipHostMalloc: 57 - 70 ms of its 90 - 110.)
//   hipcc -O2 --offload-arch=gfx950 tools/ubench_pin.hip -o /tmp/pin -pthread && /tmp/pin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <sys/mman.h>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) printf("  !! %s -> %s\n", #x, hipGetErrorString(e_)); } while (0)
static void touch(char* p, size_t n, unsigned T) {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t) th.emplace_back([=] { for (size_t i = n * t / T; i < n * (t + 1) / T; i += 4096) p[i] = 1; });
    for (auto& x : th) x.join();
}
int main() {
    CK(hipSetDevice(0)); CK(hipFree(0));
    const size_t total = 400u << 20;
    void* d = nullptr; CK(hipMalloc(&d, total)); CK(hipMemset(d, 1, total));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 2; ++rep) {
        void* p = nullptr;
        double t0 = now();
        CK(hipHostMalloc(&p, total, hipHostMallocDefault));
        double t1 = now();
        CK(hipMemcpyAsync(p, d, total, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        double t2 = now();
        CK(hipMemcpyAsync(p, d, total, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        double t3 = now();
        CK(hipHostFree(p));
        printf("hipHostMalloc 400 MB: %.1f ms; D2H into it %.1f ms, again %.1f ms (%.1f GB/s); free %.1f ms\n", t1 - t0, t2 - t1, t3 - t2, total / (t3 - t2) / 1e6, now() - t3);
    }
    for (unsigned T : {1u, 8u, 8u}) {
        double t0 = now();
        char* buf = (char*)aligned_alloc(2u << 20, total);
        touch(buf, total, T);
        double t1 = now();
        CK(hipHostRegister(buf, total, hipHostRegisterDefault));
        double t2 = now();
        CK(hipMemcpyAsync(buf, d, total, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        double t3 = now();
        CK(hipMemcpyAsync(buf, d, total, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        double t4 = now();
        CK(hipHostUnregister(buf));
        double t5 = now();
        free(buf);
        printf("aligned_alloc + touch on %u threads %.1f ms; ONE hipHostRegister %.1f ms; D2H %.1f ms, again %.1f ms (%.1f GB/s); unregister %.1f ms, free %.1f ms\n",
               T, t1 - t0, t2 - t1, t3 - t2, t4 - t3, total / (t4 - t3) / 1e6, t5 - t4, now() - t5);
    }
    for (unsigned T : {4u, 8u}) {
        double t0 = now();
        char* buf = (char*)aligned_alloc(2u << 20, total);
        touch(buf, total, 8);
        double t1 = now();
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t) th.emplace_back([=] { CK(hipSetDevice(0)); CK(hipHostRegister(buf + t * (total / T), total / T, hipHostRegisterDefault)); });
        for (auto& x : th) x.join();
        double t2 = now();
        // one copy per piece (a copy never spans two registrations)
        for (unsigned t = 0; t < T; ++t) CK(hipMemcpyAsync(buf + t * (total / T), (char*)d + t * (total / T), total / T, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double t3 = now();
        for (unsigned t = 0; t < T; ++t) CK(hipMemcpyAsync(buf + t * (total / T), (char*)d + t * (total / T), total / T, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double t4 = now();
        for (unsigned t = 0; t < T; ++t) CK(hipHostUnregister(buf + t * (total / T)));
        double t5 = now();
        free(buf);
        printf("touch on 8 threads %.1f ms; %u pieces registered on %u threads %.1f ms; D2H piecewise %.1f ms, again %.1f ms (%.1f GB/s); unregister %.1f ms\n",
               t1 - t0, T, T, t2 - t1, t3 - t2, t4 - t3, total / (t4 - t3) / 1e6, t5 - t4);
    }
    for (int huge : {0, 1, 1}) {   // untouched memory registered directly; with transparent huge pages asked for
        double t0 = now();
        char* buf = (char*)aligned_alloc(2u << 20, total);
        if (huge) madvise(buf, total, MADV_HUGEPAGE);
        double t1 = now();
        CK(hipHostRegister(buf, total, hipHostRegisterDefault));
        double t2 = now();
        CK(hipMemcpyAsync(buf, d, total, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        double t3 = now();
        CK(hipHostUnregister(buf));
        free(buf);
        printf("untouched memory (MADV_HUGEPAGE %d): alloc %.1f ms, hipHostRegister %.1f ms, D2H %.1f ms\n", huge, t1 - t0, t2 - t1, t3 - t2);
    }
    for (int huge : {1, 1}) {
        double t0 = now();
        char* buf = (char*)aligned_alloc(2u << 20, total);
        madvise(buf, total, MADV_HUGEPAGE);
        touch(buf, total, 8);
        double t1 = now();
        CK(hipHostRegister(buf, total, hipHostRegisterDefault));
        double t2 = now();
        CK(hipMemcpyAsync(buf, d, total, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        double t3 = now();
        CK(hipHostUnregister(buf));
        free(buf);
        printf("MADV_HUGEPAGE + touch on 8 threads %.1f ms, hipHostRegister %.1f ms, D2H %.1f ms\n", t1 - t0, t2 - t1, t3 - t2);
    }
    {   // pageable destination, for scale
        char* buf = (char*)aligned_alloc(2u << 20, total);
        double t0 = now();
        CK(hipMemcpy(buf, d, total, hipMemcpyDeviceToHost));
        double t1 = now();
        CK(hipMemcpy(buf, d, total, hipMemcpyDeviceToHost));
        printf("pageable destination: first %.1f ms, again %.1f ms\n", t1 - t0, now() - t1);
        free(buf);
    }
    return 0;
}

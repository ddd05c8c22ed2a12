// D2H / H2D rates of this box: pageable vs pinned host memory, and what pinning itself costs.  Decides how
// lgmi_dresult_fetch should bring 10+ GB of rows to the host.   hipcc -O2 --offload-arch=gfx950 tools/ubench_copy.hip -o /tmp/uc && /tmp/uc
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t GB = 1ull << 30, n = 4 * GB;
    void* d; hipMalloc(&d, n); hipMemset(d, 1, n); hipDeviceSynchronize();
    double t = now(); char* pg = (char*)malloc(n); memset(pg, 0, n); printf("malloc+touch 4 GiB pageable: %.3f s\n", now() - t);
    for (int k = 0; k < 2; ++k) { t = now(); hipMemcpy(pg, d, n, hipMemcpyDeviceToHost); printf("D2H pageable: %.2f GB/s\n", n / (now() - t) / 1e9); }
    t = now(); hipMemcpy(d, pg, n, hipMemcpyHostToDevice); printf("H2D pageable: %.2f GB/s\n", n / (now() - t) / 1e9);
    t = now(); void* pin; hipHostMalloc(&pin, n, hipHostMallocDefault); printf("hipHostMalloc 4 GiB: %.3f s\n", now() - t);
    for (int k = 0; k < 2; ++k) { t = now(); hipMemcpy(pin, d, n, hipMemcpyDeviceToHost); printf("D2H pinned: %.2f GB/s\n", n / (now() - t) / 1e9); }
    t = now(); hipMemcpy(d, pin, n, hipMemcpyHostToDevice); printf("H2D pinned: %.2f GB/s\n", n / (now() - t) / 1e9);
    t = now(); hipHostRegister(pg, n, hipHostRegisterDefault); printf("hipHostRegister 4 GiB (touched): %.3f s\n", now() - t);
    t = now(); hipMemcpy(pg, d, n, hipMemcpyDeviceToHost); printf("D2H registered: %.2f GB/s\n", n / (now() - t) / 1e9);
    t = now(); hipHostUnregister(pg); printf("unregister: %.3f s\n", now() - t);
    t = now(); hipHostFree(pin); printf("hipHostFree: %.3f s\n", now() - t);
    t = now(); memcpy(pg, pg + n / 2, n / 2); printf("host memcpy 2 GiB: %.2f GB/s\n", (n / 2) / (now() - t) / 1e9);
    return 0;
}

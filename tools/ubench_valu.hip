// ubench_valu.hip — issue rate of the VALU ops the count kernel is made of (gfx950).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9E3779B9u, a2 = a0 * 3u + 1u, a3 = a1 * 5u + 7u;
    uint32_t b0 = a0 + 11u, b1 = a1 + 13u, b2 = a2 + 17u, b3 = a3 + 19u;
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) {  // v_and_b32 + v_bcnt_u32_b32 (accumulate): the count kernel's pair
                c0 += __popc(a0 & b0); c1 += __popc(a1 & b1); c2 += __popc(a2 & b2); c3 += __popc(a3 & b3);
                c4 += __popc(a0 & b1); c5 += __popc(a1 & b2); c6 += __popc(a2 & b3); c7 += __popc(a3 & b0);
            } else if (OP == 1) {  // bcnt only
                c0 += __popc(a0); c1 += __popc(a1); c2 += __popc(a2); c3 += __popc(a3);
                c4 += __popc(b0); c5 += __popc(b1); c6 += __popc(b2); c7 += __popc(b3);
            } else if (OP == 2) {  // and + add (full-rate reference)
                c0 += (a0 & b0); c1 += (a1 & b1); c2 += (a2 & b2); c3 += (a3 & b3);
                c4 += (a0 & b1); c5 += (a1 & b2); c6 += (a2 & b3); c7 += (a3 & b0);
            } else if (OP == 3) {  // xor chain only
                c0 ^= a0 + c1; c1 ^= a1 + c2; c2 ^= a2 + c3; c3 ^= a3 + c4;
                c4 ^= b0 + c5; c5 ^= b1 + c6; c6 ^= b2 + c7; c7 ^= b3 + c0;
            } else if (OP == 4) {  // v_mul_hi_u32 (Philox)
                c0 += __umulhi(a0, b0 + c1); c1 += __umulhi(a1, b1 + c2); c2 += __umulhi(a2, b2 + c3); c3 += __umulhi(a3, b3 + c4);
                c4 += __umulhi(a0, b1 + c5); c5 += __umulhi(a1, b2 + c6); c6 += __umulhi(a2, b3 + c7); c7 += __umulhi(a3, b0 + c0);
            }
            // keep operands changing so nothing is hoisted
            a0 += c7; a1 ^= c0; a2 += c1; a3 ^= c2;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3;
}

template <int OP>
void run(const char* name, double ops_per_inner) {
    uint32_t* d;
    const int blocks = 256 * 8, iters = 2000;
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double inner = (double)blocks * 256 * iters * 16;
    printf("%-28s %8.3f ms  %.3e lane-inner/s  => %.2f T lane-ops/s if %g ops per inner (4 fixed-cost adds/xors included)\n",
           name, ms, inner / (ms * 1e-3), inner * ops_per_inner / (ms * 1e-3) / 1e12, ops_per_inner);
    hipFree(d);
}

int main() {
    run<2>("and+add (16 ops) +4", 20);
    run<0>("and+bcnt_acc (16 ops) +4", 20);
    run<1>("bcnt_acc (8 ops) +4", 12);
    run<3>("add+xor (16 ops) +4", 20);
    run<4>("add+mulhi+add (24 ops) +4", 28);
    return 0;
}

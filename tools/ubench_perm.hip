// ubench_perm.hip — issue rates of the building blocks of the permutation kernels (gfx950):
// 64-bit multiply forms used by Philox, f64 fma chains, f64 division.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, uint32_t seed, int iters) {
    uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9E3779B9u, a2 = a0 * 3u + 1u, a3 = a1 * 5u + 7u;
    double f0 = 1.0 + a0 * 1e-10, f1 = 1.0 + a1 * 1e-10, f2 = 1.0 + a2 * 1e-10, f3 = 1.0 + a3 * 1e-10;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) {          // 4 x (64-bit product, both halves used): what `(uint64_t)a*b` compiles to
                uint64_t p0 = (uint64_t)a0 * 0xD2511F53u, p1 = (uint64_t)a1 * 0xCD9E8D57u;
                uint64_t p2 = (uint64_t)a2 * 0xD2511F53u, p3 = (uint64_t)a3 * 0xCD9E8D57u;
                a0 = (uint32_t)(p1 >> 32) ^ (uint32_t)p0; a1 = (uint32_t)(p0 >> 32) ^ (uint32_t)p1;
                a2 = (uint32_t)(p3 >> 32) ^ (uint32_t)p2; a3 = (uint32_t)(p2 >> 32) ^ (uint32_t)p3;
            } else if (OP == 1) {   // explicit v_mul_hi_u32 + v_mul_lo_u32
                uint32_t h0 = __umulhi(a0, 0xD2511F53u), l0 = a0 * 0xD2511F53u, h1 = __umulhi(a1, 0xCD9E8D57u), l1 = a1 * 0xCD9E8D57u;
                uint32_t h2 = __umulhi(a2, 0xD2511F53u), l2 = a2 * 0xD2511F53u, h3 = __umulhi(a3, 0xCD9E8D57u), l3 = a3 * 0xCD9E8D57u;
                a0 = h1 ^ l0; a1 = h0 ^ l1; a2 = h3 ^ l2; a3 = h2 ^ l3;
            } else if (OP == 2) {   // 8 independent-ish f64 fma
                f0 = f0 * 1.0000001 + f1; f1 = f1 * 0.9999999 + f2; f2 = f2 * 1.0000001 + f3; f3 = f3 * 0.9999999 + f0;
                f0 = f0 * 0.5 + 0.25; f1 = f1 * 0.5 + 0.25; f2 = f2 * 0.5 + 0.25; f3 = f3 * 0.5 + 0.25;
            } else if (OP == 3) {   // 4 f64 divisions
                f0 = 1.0 + f1 / (f0 + 2.0); f1 = 1.0 + f2 / (f1 + 2.0); f2 = 1.0 + f3 / (f2 + 2.0); f3 = 1.0 + f0 / (f3 + 2.0);
            } else if (OP == 4) {   // 8 f32-ish integer adds as a reference
                a0 += a1; a1 ^= a2; a2 += a3; a3 ^= a0; a0 += 7u; a1 ^= 9u; a2 += 11u; a3 ^= 13u;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1 + f2 + f3 + a0 + a1 + a2 + a3;
}

template <int OP>
void run(const char* name, double units_per_inner) {
    double* d;
    const int blocks = 256 * 8, iters = 2000;
    hipMalloc(&d, blocks * 256 * 8);
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
    double inner = (double)blocks * 256 * iters * 8;
    double wave_instr_slots = ms * 1e-3 * 2.4e9 * 1024 / 4;   // 4-cycle issue slots available chip-wide
    printf("%-34s %8.3f ms  -> %.1f issue slots (4 cyc) per inner step per wave; %g units per inner\n", name, ms,
           wave_instr_slots / (inner / 64), units_per_inner);
    hipFree(d);
}

int main() {
    run<4>("8 int add/xor", 8);
    run<0>("4 x u64 product (mad_u64_u32?)", 4);
    run<1>("4 x (mul_hi + mul_lo)", 4);
    run<2>("8 f64 fma", 8);
    run<3>("4 f64 div", 4);
    return 0;
}

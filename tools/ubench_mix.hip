// ubench_mix.hip — do scalar instructions and cheap 32-bit vector instructions ride along for free beside f64 vector
// instructions, or does a SIMD issue one instruction of any kind per ~4 cycles?  (the question
// behind k_perm_general's "time follows the instruction count").  Each variant repeats a fixed group of instructions on
// independent registers; reported: SIMD cycles per GROUP, with 8 x v_xor_b32 at one wave per SIMD = 32 cycles as the clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define F(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g[i]));
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define S(i) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s[i]) : : "scc");
#define M(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(q[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
#define B(i) asm volatile("s_and_saveexec_b64 %0, -1\n\ts_or_b64 exec, exec, %0" : "=s"(e[i]) : : "scc");

template <int V>
__global__ __launch_bounds__(256) void k(double* out, uint32_t seed, int iters)
{
    uint32_t a[8], b[8], s[8];
    double f[8], g[8];
    unsigned long long q[8], e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 2654435761u + seed + i; b[i] = a[i] ^ 0x9E3779B9u; s[i] = seed + i;
        f[i] = 1.0 + (double)(a[i] & 1023u) * 1e-3; g[i] = 1.5 + i; q[i] = a[i]; e[i] = 0;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (V == 0) { X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) }                                   // 8 xor
            if (V == 1) { F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) }                                   // 8 fma64
            if (V == 2) { F(0) S(0) F(1) S(1) F(2) S(2) F(3) S(3) F(4) S(4) F(5) S(5) F(6) S(6) F(7) S(7) }   // 8 fma64 + 8 salu
            if (V == 3) { F(0) X(0) F(1) X(1) F(2) X(2) F(3) X(3) F(4) X(4) F(5) X(5) F(6) X(6) F(7) X(7) }   // 8 fma64 + 8 xor
            if (V == 4) { X(0) S(0) X(1) S(1) X(2) S(2) X(3) S(3) X(4) S(4) X(5) S(5) X(6) S(6) X(7) S(7) }   // 8 xor + 8 salu
            if (V == 5) { S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) }                                   // 8 salu
            if (V == 6) { M(0) X(0) X(1) M(1) X(2) X(3) M(2) X(4) X(5) M(3) X(6) X(7) }               // philox-like: 4 mad + 8 xor
            if (V == 7) { F(0) B(0) F(1) B(1) F(2) B(2) F(3) B(3) F(4) B(4) F(5) B(5) F(6) B(6) F(7) B(7) }   // 8 fma64 + 8 x (saveexec, or exec)
        }
    }
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += f[i] + (double)a[i] + (double)q[i] + (double)s[i] + (double)e[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

typedef void (*kern_t)(double*, uint32_t, int);
static float run(kern_t fn, double* d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d, 1u, 8);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d, 2u, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    kern_t tab[8] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>};
    const char* names[8] = {"8 v_xor_b32", "8 v_fma_f64", "8 v_fma_f64 + 8 s_add_u32", "8 v_fma_f64 + 8 v_xor_b32",
                            "8 v_xor_b32 + 8 s_add_u32", "8 s_add_u32", "4 v_mad_u64_u32 + 8 v_xor_b32",
                            "8 v_fma_f64 + 8 x (s_and_saveexec, s_or exec)"};
    double* d;
    (void)hipMalloc(&d, (size_t)256 * 16 * 256 * 8);
    const int iters = 4000;
    const float ref = run(tab[0], d, 256, iters);        // one wave per SIMD: 8 xor = 32 cycles
    for (int wps = 1; wps <= 4; wps *= 2) {
        for (int v = 0; v < 8; ++v) {
            const float ms = run(tab[v], d, 256 * wps, iters);
            printf("wps %d  %-48s %8.3f ms  %7.2f SIMD cycles per group\n", wps, names[v], ms, 32.0 * ms / ref / wps);
        }
    }
    return 0;
}

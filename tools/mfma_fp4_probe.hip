// mfma_fp4_probe.hip — checks v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands on gfx950 before
// count_mfma.hip relies on it: format codes, unit scales, the operand lane map (lane l: row/col l & 31, k-half
// l >> 5, 32 nibbles in the first 4 dwords) and the accumulator map, with exact data made of the nibble codes
// the count kernel uses (0, 0x1 = 0.5, 0x2 = 1, 0x4 = 2; 0x8 = -0).  Also times a dependent-free MFMA stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

static __host__ __device__ float fp4_value(unsigned code) {
    static const float tab[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    const float v = tab[code & 7u];
    return (code & 8u) ? -v : v;
}

// A[32][64], B[64][32] as nibble codes, one per byte
__global__ void probe(const uint8_t* A, const uint8_t* B, float* D) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < 4; ++q) {
        unsigned wa = 0, wb = 0;
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * h + 8 * q + j;
            wa |= (unsigned)(A[r * 64 + k] & 15u) << (4 * j);
            wb |= (unsigned)(B[k * 32 + r] & 15u) << (4 * j);
        }
        a[q] = (int)wa; b[q] = (int)wb;
    }
    v16f c = {0};
    // cbsz = 4, blgp = 4: FP4 e2m1 on both sides; scale bytes 0x7F = 2^0
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h, col = r;
        D[row * 32 + col] = c[reg];
    }
}

__global__ __launch_bounds__(256) void rate(float* out, int iters) {
    v8i a = {0x22222222, 0x12121212, 0x41414141, 0x22222222, 0, 0, 0, 0};
    v8i b = {0x11111111, 0x22222222, 0x44444444, 0x21212121, 0, 0, 0, 0};
    v16f c[8];
    for (int i = 0; i < 8; ++i) c[i] = v16f{0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[i], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0];
    if (s == 12345.f) out[0] = s;
}

// how many independent VALU operations fit under one FP4 MFMA: NV v_and/v_lshrrev per MFMA, 8 accumulators
template <int NV>
__global__ __launch_bounds__(256) void rate_valu(float* out, int iters, int seedv) {
    v8i a = {0x22222222, 0x12121212, 0x41414141, 0x22222222, 0, 0, 0, 0};
    v8i b = {0x11111111, 0x22222222, 0x44444444, 0x21212121, 0, 0, 0, 0};
    v16f c[8];
    int x[8];
    for (int i = 0; i < 8; ++i) { c[i] = v16f{0}; x[i] = seedv + i * (int)threadIdx.x; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[i], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
#pragma unroll
            for (int v = 0; v < NV; ++v) x[(i + v) & 7] = (x[(i + v) & 7] >> 1) & (0x11111111 + it);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0] + (float)x[i];
    if (s == 12345.f) out[0] = s;
}

template <int NV>
static void run_rate_valu(float* dD, int blocks_per_cu) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(rate_valu<NV>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, dD, 10, 3);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_valu<NV>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, dD, iters, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = 256.0 * blocks_per_cu * 4 * 8 * iters;
    printf("  %d waves/SIMD, %d x 2 VALU per MFMA: %.1f ms, cycles per MFMA per SIMD at 2.4 GHz: %.1f\n", blocks_per_cu, NV, ms, 2.4e9 * (ms * 1e-3) / (mfmas / 1024));
}

int main() {
    uint8_t hA[32 * 64], hB[64 * 32];
    float hD[1024], ref[1024];
    const unsigned codes[5] = {0u, 1u, 2u, 4u, 8u};
    for (int r = 0; r < 32; ++r) for (int k = 0; k < 64; ++k) hA[r * 64 + k] = (uint8_t)codes[(r * 7 + k * 3 + (k >> 3)) % 5];
    for (int k = 0; k < 64; ++k) for (int c = 0; c < 32; ++c) hB[k * 32 + c] = (uint8_t)codes[(k * 5 + c * 2 + (c >> 2)) % 4];
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
        float s = 0;
        for (int k = 0; k < 64; ++k) s += fp4_value(hA[r * 64 + k]) * fp4_value(hB[k * 32 + c]);
        ref[r * 32 + c] = s;
    }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += hD[i] != ref[i];
    printf("mfma_scale_f32_32x32x64_f8f6f4 (fp4 x fp4) check: %d mismatches of 1024 (D[0]=%g ref %g, D[37]=%g ref %g)\n", bad, hD[0], ref[0], hD[37], ref[37]);
    // issue rate: 256 CUs x 4 waves x 8 accumulators
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(rate, dim3(256 * 4), dim3(256), 0, 0, dD, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate, dim3(256 * 4), dim3(256), 0, 0, dD, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = 256.0 * 4 * 4 * 8 * iters;
    printf("rate: %.1f ms, %.3g MFMA/s, %.2f POP/s (2*32*32*64 per MFMA), cycles per MFMA per SIMD at 2.4 GHz: %.1f\n", ms,
           mfmas / (ms * 1e-3), mfmas * 2 * 32 * 32 * 64 / (ms * 1e-3) / 1e15, 2.4e9 * (ms * 1e-3) / (mfmas / 1024));
    for (int w = 1; w <= 2; ++w) { run_rate_valu<0>(dD, w); run_rate_valu<1>(dD, w); run_rate_valu<2>(dD, w); run_rate_valu<3>(dD, w); run_rate_valu<4>(dD, w); }
    return bad != 0;
}

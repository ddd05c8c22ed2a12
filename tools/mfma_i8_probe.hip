// mfma_i8_probe.hip — checks the operand / accumulator lane maps of v_mfma_i32_32x32x32_i8 on gfx950
// with exact integer data (asymmetric A and B), before count.hip relies on them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void probe(const int8_t* A /*[32][32] row-major r,k*/, const int8_t* B /*[32][32] k,c*/, int* D /*[32][32]*/) {
    const int l = threadIdx.x;
    v4i a, b;
    int8_t* ab = (int8_t*)&a;
    int8_t* bb = (int8_t*)&b;
    for (int j = 0; j < 16; ++j) {
        ab[j] = A[(l & 31) * 32 + 16 * (l >> 5) + j];       // A[row = l&31][k = 16*(l>>5) + j]
        bb[j] = B[(16 * (l >> 5) + j) * 32 + (l & 31)];     // B[k][col = l&31]
    }
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5), col = l & 31;
        D[row * 32 + col] = c[reg];
    }
}

int main() {
    int8_t hA[1024], hB[1024];
    int hD[1024], ref[1024];
    for (int r = 0; r < 32; ++r) for (int k = 0; k < 32; ++k) hA[r * 32 + k] = (int8_t)((r * 7 + k * 3) % 5 - 1);
    for (int k = 0; k < 32; ++k) for (int c = 0; c < 32; ++c) hB[k * 32 + c] = (int8_t)((k * 5 + c * 2) % 7 - 2);
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
        int s = 0;
        for (int k = 0; k < 32; ++k) s += hA[r * 32 + k] * hB[k * 32 + c];
        ref[r * 32 + c] = s;
    }
    int8_t *dA, *dB; int* dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += hD[i] != ref[i];
    printf("mfma_i32_32x32x32_i8 layout check: %d mismatches of 1024\n", bad);
    return bad != 0;
}

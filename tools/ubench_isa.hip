// ubench_isa.hip — issue cost of the single instructions the permutation kernels are made of (gfx950).
// Each kernel repeats ONE instruction on eight independent register sets (no dependent chain shorter than 8
// instructions), 256-thread workgroups, WPS waves per SIMD.  Reported: SIMD cycles per wave-instruction, taking
// v_xor_b32 at 4.0 waves x cycles as the clock reference of the same launch geometry (the shader clock under load
// is not known to the host).   hipcc --offload-arch=gfx950 -O3 tools/ubench_isa.hip -o /tmp/ubench_isa
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Op { XOR32, ADD32, MAD_U64_U32, MUL_HI_U32, MUL_LO_U32, MUL_U32_U24, ADD_F64, MUL_F64, FMA_F64, CVT_F64_U32,
          CVT_U32_F64, FLOOR_F64, LDEXP_F64, RCP_F64, DIV_SCALE_F64, DIV_FMAS_F64, DIV_FIXUP_F64, CMP_LT_F64,
          EXP_F32, CVT_F32_F64, CVT_F64_F32, LSHL_B64, CNDMASK, MOV_B64, FMA_F32, RCP_F32, CVT_F32_U32, MUL_F32,
          ADD_CO_U32, ADDC_CO_U32, LSHL_ADD_U32, BFE_U32, PERM_B32, ALIGNBIT, N_OPS };
static const char* NAMES[] = { "v_xor_b32", "v_add_u32", "v_mad_u64_u32", "v_mul_hi_u32", "v_mul_lo_u32", "v_mul_u32_u24",
    "v_add_f64", "v_mul_f64", "v_fma_f64", "v_cvt_f64_u32", "v_cvt_u32_f64", "v_floor_f64", "v_ldexp_f64", "v_rcp_f64",
    "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_cmp_lt_f64", "v_exp_f32", "v_cvt_f32_f64", "v_cvt_f64_f32",
    "v_lshlrev_b64", "v_cndmask_b32", "v_mov_b64", "v_fma_f32", "v_rcp_f32", "v_cvt_f32_u32", "v_mul_f32",
    "v_add_co_u32", "v_addc_co_u32", "v_lshl_add_u32", "v_bfe_u32", "v_perm_b32", "v_alignbit_b32" };

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, uint32_t seed, int iters)
{
    uint32_t a[8], b[8];
    double f[8], g[8];
    unsigned long long q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 2654435761u + seed + i; b[i] = a[i] ^ 0x9E3779B9u;
        f[i] = 1.0 + (double)(a[i] & 1023u) * 1e-3; g[i] = 1.5 + i;
        q[i] = ((unsigned long long)a[i] << 20) | b[i];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#define X_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define X_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define X_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(q[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
#define X_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define X_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define X_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define X_ADDF64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[i]) : "v"(g[i]));
#define X_MULF64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[i]) : "v"(g[i]));
#define X_FMAF64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g[i]));
#define X_CVTF64U32(i) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(f[i]) : "v"(a[i]));
#define X_CVTU32F64(i) asm volatile("v_cvt_u32_f64 %0, %1" : "+v"(a[i]) : "v"(f[i]));
#define X_FLOORF64(i) asm volatile("v_floor_f64 %0, %0" : "+v"(f[i]));
#define X_LDEXP(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(f[i]) : "v"(a[i]));
#define X_RCPF64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(f[i]));
#define X_DIVSCALE(i) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(f[i]) : "v"(g[i]) : "vcc");
#define X_DIVFMAS(i) asm volatile("v_div_fmas_f64 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g[i]) : "vcc");
#define X_DIVFIXUP(i) asm volatile("v_div_fixup_f64 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g[i]));
#define X_CMPF64(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(f[i]), "v"(g[i]) : "vcc");
#define X_EXPF32(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define X_CVTF32F64(i) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(a[i]) : "v"(f[i]));
#define X_CVTF64F32(i) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(f[i]) : "v"(a[i]));
#define X_LSHL64(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q[i]));
#define X_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
#define X_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "+v"(f[i]) : "v"(g[i]));
#define X_FMAF32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
#define X_RCPF32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define X_CVTF32U32(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
#define X_MULF32(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define X_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b[i]) : "vcc");
#define X_ADDCCO(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
#define X_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b[i]));
#define X_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a[i]));
#define X_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
#define X_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(a[i]));
            if (OP == XOR32) { REP8(X_XOR) }
            else if (OP == ADD32) { REP8(X_ADD) }
            else if (OP == MAD_U64_U32) { REP8(X_MAD64) }
            else if (OP == MUL_HI_U32) { REP8(X_MULHI) }
            else if (OP == MUL_LO_U32) { REP8(X_MULLO) }
            else if (OP == MUL_U32_U24) { REP8(X_MUL24) }
            else if (OP == ADD_F64) { REP8(X_ADDF64) }
            else if (OP == MUL_F64) { REP8(X_MULF64) }
            else if (OP == FMA_F64) { REP8(X_FMAF64) }
            else if (OP == CVT_F64_U32) { REP8(X_CVTF64U32) }
            else if (OP == CVT_U32_F64) { REP8(X_CVTU32F64) }
            else if (OP == FLOOR_F64) { REP8(X_FLOORF64) }
            else if (OP == LDEXP_F64) { REP8(X_LDEXP) }
            else if (OP == RCP_F64) { REP8(X_RCPF64) }
            else if (OP == DIV_SCALE_F64) { REP8(X_DIVSCALE) }
            else if (OP == DIV_FMAS_F64) { REP8(X_DIVFMAS) }
            else if (OP == DIV_FIXUP_F64) { REP8(X_DIVFIXUP) }
            else if (OP == CMP_LT_F64) { REP8(X_CMPF64) }
            else if (OP == EXP_F32) { REP8(X_EXPF32) }
            else if (OP == CVT_F32_F64) { REP8(X_CVTF32F64) }
            else if (OP == CVT_F64_F32) { REP8(X_CVTF64F32) }
            else if (OP == LSHL_B64) { REP8(X_LSHL64) }
            else if (OP == CNDMASK) { REP8(X_CNDMASK) }
            else if (OP == MOV_B64) { REP8(X_MOV64) }
            else if (OP == FMA_F32) { REP8(X_FMAF32) }
            else if (OP == RCP_F32) { REP8(X_RCPF32) }
            else if (OP == CVT_F32_U32) { REP8(X_CVTF32U32) }
            else if (OP == MUL_F32) { REP8(X_MULF32) }
            else if (OP == ADD_CO_U32) { REP8(X_ADDCO) }
            else if (OP == ADDC_CO_U32) { REP8(X_ADDCCO) }
            else if (OP == LSHL_ADD_U32) { REP8(X_LSHLADD) }
            else if (OP == BFE_U32) { REP8(X_BFE) }
            else if (OP == PERM_B32) { REP8(X_PERM) }
            else if (OP == ALIGNBIT) { REP8(X_ALIGN) }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i] + (double)a[i] + (double)q[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*kern_t)(double*, uint32_t, int);
template <int OP> struct Tab { static void fill(kern_t* t) { t[OP] = k<OP>; Tab<OP + 1>::fill(t); } };
template <> struct Tab<N_OPS> { static void fill(kern_t*) {} };

static float run(kern_t fn, double* d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d, 1u, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d, 2u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

int main()
{
    kern_t tab[N_OPS];
    Tab<0>::fill(tab);
    double* d;
    hipMalloc(&d, (size_t)256 * 16 * 256 * 8);
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;             // one 256-thread workgroup = one wave on each SIMD of a CU
        const float ref = run(tab[XOR32], d, blocks, iters);
        printf("# %d wave(s) per SIMD; v_xor_b32 reference %.3f ms = 4.0 cycles per wave-instruction and wave\n", wps, ref);
        for (int op = 0; op < N_OPS; ++op) {
            const float ms = run(tab[op], d, blocks, iters);
            printf("%-18s wps %d  %8.3f ms  %6.2f cycles per wave-instruction (SIMD busy)\n", NAMES[op], wps, ms, 4.0 * ms / ref);
        }
    }
    hipFree(d);
    return 0;
}

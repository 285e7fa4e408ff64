// valu_rates.hip -- measured issue cost of the instructions the Gibbs kernels are made of (gfx950).
// Every kernel runs N_ITER iterations of UNROLL independent instances of one instruction per wave,
// 8 waves per SIMD on every CU, so the result is the throughput cost in cycles per wave64 instruction
// per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define N_ITER 2048
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define KERNEL(NAME, DECL, BODY, SINK)                                               \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)        \
    {                                                                                \
        DECL;                                                                        \
        for (int it = 0; it < N_ITER; ++it) {                                        \
            BODY BODY BODY BODY                                                      \
        }                                                                            \
        out[blockIdx.x * 256 + threadIdx.x] = SINK;                                  \
    }

// 8 independent chains per body -> 32 instructions per loop trip
#define U8(a) uint32_t a##0 = seed + threadIdx.x, a##1 = a##0 * 3, a##2 = a##0 * 5, a##3 = a##0 * 7, a##4 = a##0 * 11, a##5 = a##0 * 13, a##6 = a##0 * 17, a##7 = a##0 * 19
#define D8(a) double a##0 = seed + threadIdx.x, a##1 = a##0 * 3, a##2 = a##0 * 5, a##3 = a##0 * 7, a##4 = a##0 * 11, a##5 = a##0 * 13, a##6 = a##0 * 17, a##7 = a##0 * 19
#define F8(a) float a##0 = seed + threadIdx.x, a##1 = a##0 * 3, a##2 = a##0 * 5, a##3 = a##0 * 7, a##4 = a##0 * 11, a##5 = a##0 * 13, a##6 = a##0 * 17, a##7 = a##0 * 19
#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define XSUM(a) (uint32_t)(a##0 + a##1 + a##2 + a##3 + a##4 + a##5 + a##6 + a##7)

#define OP_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x##i) : "v"(k));
KERNEL(k_add_u32, U8(x); uint32_t k = seed | 1, R8(OP_ADD), XSUM(x))
#define OP_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x##i) : "v"(k));
KERNEL(k_xor_b32, U8(x); uint32_t k = seed | 1, R8(OP_XOR), XSUM(x))
#define OP_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x##i) : "v"(k));
KERNEL(k_mul_lo_u32, U8(x); uint32_t k = seed | 1, R8(OP_MULLO), XSUM(x))
#define OP_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x##i) : "v"(k));
KERNEL(k_mul_hi_u32, U8(x); uint32_t k = seed | 1, R8(OP_MULHI), XSUM(x))
#define OP_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y##i) : "v"(x##i), "v"(k) : "vcc");
#define U64_8 uint64_t y0 = seed, y1 = 1, y2 = 2, y3 = 3, y4 = 4, y5 = 5, y6 = 6, y7 = 7
KERNEL(k_mad_u64_u32, U8(x); U64_8; uint32_t k = seed | 1, R8(OP_MAD64), (uint32_t)(y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7))
#define OP_CMPADDC(i) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(x##i) : "v"(k), "v"(x##i) : "vcc");
KERNEL(k_cmp_addc_pair, U8(x); uint32_t k = seed | 1, R8(OP_CMPADDC), XSUM(x))
#define OP_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d##i) : "v"(kd), "v"(kd2));
KERNEL(k_fma_f64, D8(d); double kd = 1.0000001; double kd2 = 1e-9, R8(OP_FMA64), (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7))
#define OP_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##i) : "v"(kd));
KERNEL(k_add_f64, D8(d); double kd = 1.0000001, R8(OP_ADD64), (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7))
#define OP_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##i) : "v"(kd));
KERNEL(k_mul_f64, D8(d); double kd = 1.0000001, R8(OP_MUL64), (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7))
#define OP_RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d##i));
KERNEL(k_rcp_f64, D8(d), R8(OP_RCP64), (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7))
#define OP_CVTU(i) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(x##i) : "v"(d##i));
KERNEL(k_cvt_u32_f64, D8(d); U8(x), R8(OP_CVTU), XSUM(x))
#define OP_CVTD(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d##i) : "v"(x##i));
KERNEL(k_cvt_f64_u32, D8(d); U8(x), R8(OP_CVTD), (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7))
#define OP_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f##i) : "v"(kf), "v"(kf2));
KERNEL(k_fma_f32, F8(f); float kf = 1.0001f; float kf2 = 1e-5f, R8(OP_FMA32), (uint32_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7))
#define OP_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x##i) : "v"(k) : "vcc");
KERNEL(k_cndmask_b32, U8(x); uint32_t k = seed | 1, R8(OP_CND), XSUM(x))
#define OP_LSH(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x##i));
KERNEL(k_lshl_b32, U8(x), R8(OP_LSH), XSUM(x))
#define OP_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x##i));
KERNEL(k_mov_dpp, U8(x), R8(OP_DPP), XSUM(x))
#define OP_LOG32(i) asm volatile("v_log_f32 %0, %0" : "+v"(f##i));
KERNEL(k_log_f32, F8(f), R8(OP_LOG32), (uint32_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7))
#define OP_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(x##i));
KERNEL(k_bfe_u32, U8(x), R8(OP_BFE), XSUM(x))
#define OP_LSH64(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(y##i));
KERNEL(k_lshr_b64, U64_8, R8(OP_LSH64), (uint32_t)(y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7))

#define OP_BITOP3(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x##i) : "v"(k), "v"(k2));
KERNEL(k_bitop3_b32, U8(x); uint32_t k = seed | 1; uint32_t k2 = seed * 7, R8(OP_BITOP3), XSUM(x))
#define OP_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %0, 21" : "+v"(x##i));
KERNEL(k_alignbit_b32, U8(x), R8(OP_ALIGNBIT), XSUM(x))
#define OP_BFREV(i) asm volatile("v_bfrev_b32 %0, %0" : "+v"(x##i));
KERNEL(k_bfrev_b32, U8(x), R8(OP_BFREV), XSUM(x))
#define OP_CMPI(i) asm volatile("v_cmp_gt_i32 vcc, 0, %0" : : "v"(x##i) : "vcc");
KERNEL(k_cmp_gt_i32, U8(x), R8(OP_CMPI), XSUM(x))
#define OP_OR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x##i) : "v"(k), "v"(k2));
KERNEL(k_or3_b32, U8(x); uint32_t k = seed | 1; uint32_t k2 = seed * 7, R8(OP_OR3), XSUM(x))
#define OP_LDEXP64(i) asm volatile("v_ldexp_f64 %0, %0, 3" : "+v"(d##i));
KERNEL(k_ldexp_f64, D8(d), R8(OP_LDEXP64), (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7))
#define OP_CMP64(i) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d##i), "v"(kd) : "vcc");
KERNEL(k_cmp_f64, D8(d); double kd = 1.0000001, R8(OP_CMP64), (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7))
#define OP_DIVSCALE(i) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(d##i) : "v"(kd) : "vcc");
KERNEL(k_div_scale_f64, D8(d); double kd = 1.0000001, R8(OP_DIVSCALE), (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7))
#define OP_DIVFIXUP(i) asm volatile("v_div_fixup_f64 %0, %0, %1, %1" : "+v"(d##i) : "v"(kd));
KERNEL(k_div_fixup_f64, D8(d); double kd = 1.0000001, R8(OP_DIVFIXUP), (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7))
#define OP_DIVFMAS(i) asm volatile("v_div_fmas_f64 %0, %0, %1, %1" : "+v"(d##i) : "v"(kd) : "vcc");
KERNEL(k_div_fmas_f64, D8(d); double kd = 1.0000001, R8(OP_DIVFMAS), (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7))
#define OP_ADDSUB(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x##i) : "v"(k));
KERNEL(k_sub_u32, U8(x); uint32_t k = seed | 1, R8(OP_ADDSUB), XSUM(x))

typedef void (*kfn)(uint32_t *, uint32_t);

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8;                 // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    uint32_t *out;
    CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    struct { const char *name; kfn fn; int per_body; } ks[] = {
        {"v_bitop3_b32 (xor3)", k_bitop3_b32, 8}, {"v_alignbit_b32", k_alignbit_b32, 8}, {"v_bfrev_b32", k_bfrev_b32, 8}, {"v_cmp_gt_i32", k_cmp_gt_i32, 8}, {"v_or3_b32", k_or3_b32, 8},
        {"v_ldexp_f64", k_ldexp_f64, 8}, {"v_cmp_gt_f64", k_cmp_f64, 8}, {"v_div_scale_f64", k_div_scale_f64, 8}, {"v_div_fixup_f64", k_div_fixup_f64, 8}, {"v_div_fmas_f64", k_div_fmas_f64, 8}, {"v_sub_u32", k_sub_u32, 8},
        {"v_add_u32", k_add_u32, 8}, {"v_xor_b32", k_xor_b32, 8}, {"v_lshlrev_b32", k_lshl_b32, 8}, {"v_bfe_u32", k_bfe_u32, 8},
        {"v_cndmask_b32", k_cndmask_b32, 8}, {"v_mov_b32_dpp", k_mov_dpp, 8},
        {"v_mul_lo_u32", k_mul_lo_u32, 8}, {"v_mul_hi_u32", k_mul_hi_u32, 8}, {"v_mad_u64_u32", k_mad_u64_u32, 8},
        {"v_cmp_lt_u32+v_addc_co_u32 (pair)", k_cmp_addc_pair, 8}, {"v_lshrrev_b64", k_lshr_b64, 8},
        {"v_fma_f32", k_fma_f32, 8}, {"v_log_f32", k_log_f32, 8},
        {"v_fma_f64", k_fma_f64, 8}, {"v_add_f64", k_add_f64, 8}, {"v_mul_f64", k_mul_f64, 8}, {"v_rcp_f64", k_rcp_f64, 8},
        {"v_cvt_u32_f64", k_cvt_u32_f64, 8}, {"v_cvt_f64_u32", k_cvt_f64_u32, 8},
    };
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    int clk_khz = 0;
    CHK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    printf("device %s, %d CUs, clock attr %.0f MHz\n", prop.gcnArchName, cus, clk_khz / 1000.0);
    printf("%-36s %10s %14s %18s\n", "instruction", "us", "Ginst/s/SIMD", "cycles@2.4GHz/inst");
    for (auto &k : ks) {
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 12345u);
        CHK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 12345u);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // wave-instructions per SIMD: 8 waves/SIMD * N_ITER * 4 bodies * per_body
        const double winst = 8.0 * N_ITER * 4.0 * k.per_body;
        const double sec = best * 1e-3;
        printf("%-36s %10.1f %14.3f %18.2f\n", k.name, best * 1e3, winst / sec * 1e-9, sec * 2.4e9 / winst);
    }
    return 0;
}

// mfma_f64_ab.hip -- A/B of the NMFT contraction R = tau . gamma (Init_NMFT.py:163,171) on gfx950:
//   (A) VALU form used by nmft_wave_kernel: lane = sample, the tau row of (variant, base) is an LDS broadcast,
//       one v_fma_f64 per (row, haplotype) per lane -> 4 variants = 16 rows x G = 8: 128 FMAs per lane
//   (B) v_mfma_f64_16x16x4_f64: 16 rows (4 variants x 4 bases) x 16 samples x K = 4 haplotypes per instruction,
//       8 instructions per 4 variants x 64 samples; tau fragment from LDS (1 double per lane per K-block), gamma
//       fragments resident in registers
// Both compute the same 16 x 64 tile per step (checked against each other), many steps per wavefront, every SIMD of
// the chip loaded with `wps` wavefronts.  Prints time per step and FMA/s; under rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64
// SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES the MFMA-pipe occupancy can be read beside it.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_f64_ab.hip -o mfma_f64_ab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

typedef double double4_t __attribute__((ext_vector_type(4)));
#define G 8
#define STEPS 512
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// tau [nsteps][16 rows][G], gamma [G][64]; out [blocks*4 waves][16 rows][64 samples] = sum over the wave's steps of R
__global__ __launch_bounds__(256) void k_valu(const double *__restrict__ tau, const double *__restrict__ gamma, double *__restrict__ out, int nsteps)
{
    __shared__ double ts[4][16 * G];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double g[G], acc[16];
    for (int k = 0; k < G; ++k) g[k] = gamma[k * 64 + lane];
    for (int r = 0; r < 16; ++r) acc[r] = 0.0;
    const int w = blockIdx.x * 4 + wv;
    for (int st = 0; st < nsteps; ++st) {
        const double *t = tau + ((size_t)((w + st) % nsteps)) * 16 * G;
        ts[wv][lane] = t[lane]; ts[wv][lane + 64] = t[lane + 64];           // 128 doubles per step
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double R = 0.0;
#pragma unroll
            for (int k = 0; k < G; ++k) R = fma(ts[wv][r * G + k], g[k], R);
            acc[r] += R;
        }
        __builtin_amdgcn_wave_barrier();
    }
    for (int r = 0; r < 16; ++r) out[((size_t)w * 16 + r) * 64 + lane] = acc[r];
}

__global__ __launch_bounds__(256) void k_mfma(const double *__restrict__ tau, const double *__restrict__ gamma, double *__restrict__ out, int nsteps)
{
    __shared__ double ts[4][16 * G];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // B fragments: B[k][n] with k = lane / 16, n = lane % 16 -> gamma[4 kb + lane/16][16 t + lane%16]
    double b[4][2];
    for (int t = 0; t < 4; ++t)
        for (int kb = 0; kb < 2; ++kb) b[t][kb] = gamma[(4 * kb + lane / 16) * 64 + 16 * t + lane % 16];
    double4_t acc[4];
    for (int t = 0; t < 4; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const int w = blockIdx.x * 4 + wv;
    for (int st = 0; st < nsteps; ++st) {
        const double *t = tau + ((size_t)((w + st) % nsteps)) * 16 * G;
        ts[wv][lane] = t[lane]; ts[wv][lane + 64] = t[lane + 64];
        __builtin_amdgcn_wave_barrier();
        // A fragments: A[m][k] with m = lane % 16 (row), k = lane / 16 -> tau[row][4 kb + lane/16]
        const double a0 = ts[wv][(lane % 16) * G + lane / 16], a1 = ts[wv][(lane % 16) * G + 4 + lane / 16];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[tt][0], acc[tt], 0, 0, 0);
            acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[tt][1], acc[tt], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // D[i][j]: lane = j + 16 * (i % 4), element i / 4  (scripts/ubench/mfma_layout_probe.hip prints this on the device)
    for (int tt = 0; tt < 4; ++tt)
        for (int e = 0; e < 4; ++e) out[((size_t)w * 16 + 4 * e + lane / 16) * 64 + 16 * tt + lane % 16] = acc[tt][e];
}

int main(int argc, char **argv)
{
    const int wps = argc > 1 ? atoi(argv[1]) : 4;                  // wavefronts per SIMD
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * wps;             // 4 waves per block -> wps waves per SIMD
    const int nwaves = blocks * 4;
    double *tau, *gam, *o1, *o2;
    CHK(hipMalloc(&tau, (size_t)STEPS * 16 * G * 8)); CHK(hipMalloc(&gam, G * 64 * 8));
    CHK(hipMalloc(&o1, (size_t)nwaves * 16 * 64 * 8)); CHK(hipMalloc(&o2, (size_t)nwaves * 16 * 64 * 8));
    double *h = (double *)malloc((size_t)STEPS * 16 * G * 8), hg[G * 64];
    srand(1);
    for (size_t i = 0; i < (size_t)STEPS * 16 * G; ++i) h[i] = rand() / (double)RAND_MAX;
    for (int i = 0; i < G * 64; ++i) hg[i] = rand() / (double)RAND_MAX;
    CHK(hipMemcpy(tau, h, (size_t)STEPS * 16 * G * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(gam, hg, sizeof hg, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms[2];
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0, 0));
            if (v == 0) hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, tau, gam, o1, STEPS);
            else hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, tau, gam, o2, STEPS);
            CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms[v], e0, e1));
        }
    }
    double *r1 = (double *)malloc((size_t)nwaves * 1024 * 8), *r2 = (double *)malloc((size_t)nwaves * 1024 * 8);
    CHK(hipMemcpy(r1, o1, (size_t)nwaves * 1024 * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(r2, o2, (size_t)nwaves * 1024 * 8, hipMemcpyDeviceToHost));
    double md = 0; for (size_t i = 0; i < (size_t)nwaves * 1024; ++i) md = fmax(md, fabs(r1[i] - r2[i]) / fmax(1.0, fabs(r1[i])));
    const double fma_total = (double)nwaves * STEPS * 16 * 64 * G;
    printf("%s, %d CUs, %d wavefronts per SIMD, %d steps of a 16x64x%d tile per wavefront; max rel diff VALU vs MFMA %.2e\n", prop.gcnArchName,
           prop.multiProcessorCount, wps, STEPS, G, md);
    const char *nm[2] = {"VALU v_fma_f64 (LDS-broadcast tau)", "v_mfma_f64_16x16x4_f64"};
    for (int v = 0; v < 2; ++v)
        printf("%-36s %8.1f us  %7.2f TFMA/s  (%.1f ns per 16x64x%d step per wavefront; fp64 vector peak 39.3 TFMA/s)\n", nm[v], ms[v] * 1e3,
               fma_total / (ms[v] * 1e-3) * 1e-12, ms[v] * 1e6 / STEPS, G);
    return 0;
}

// mfma_f64_rate.hip -- what a v_mfma_f64_16x16x4_f64 costs on gfx950, in shader cycles (s_memtime) and wall time:
//   dep:   one accumulator, N dependent MFMAs               -> dependent-accumulator latency
//   ind4:  four accumulators round robin                    -> issue rate of one wavefront
//   chain3: groups of three dependent MFMAs from zero (the R = tau . gamma chain of the NMF kernels), four groups in flight
//   mix:   ind4 with K independent v_fma_f64 between MFMAs  -> how much VALU work hides under the matrix pipe
// every SIMD loaded with `wps` wavefronts (argv[1]).  Build: hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define N 4096

template <int MODE, int K>
__global__ __launch_bounds__(256) void k(double *out, long long *cyc, double a0, double b0)
{
    const int lane = threadIdx.x & 63;
    double a = a0 + lane * 1e-9, b = b0 + lane * 1e-9;
    double4_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < N / 16; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
            else if (MODE == 1 || MODE == 3) acc[j & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j & 3], 0, 0, 0);
            else if (MODE == 2) {
                // j = 4 g + s: group g of four, step s: chains of three, the fourth MFMA is the consumer (another accumulator)
                const int s = j & 3;
                if (s < 3) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, s == 0 ? (double4_t){0.0, 0.0, 0.0, 0.0} : acc[0], 0, 0, 0);
                else acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[0][0], b, acc[1], 0, 0, 0);
            }
            // round 6: the four-block 4x4x4 form (v_mfma_f64_4x4x4_4b_f64: 512 flops against the 16x16x4's 2048) -- one output per lane
            if (MODE == 4) v[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, v[0], 0, 0, 0);
            if (MODE == 5 || MODE == 6) v[j & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, v[j & 3], 0, 0, 0);
            if (MODE == 6) {
#pragma unroll
                for (int kk = 0; kk < K; ++kk) v[4 + (kk & 3)] = fma(v[4 + (kk & 3)], a, b);
            }
            if (MODE == 7) {           // plain v_fma_f64, eight accumulators: the vector ALU's own rate
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) v[kk] = fma(v[kk], a, b);
            }
            if (MODE == 3) {
#pragma unroll
                for (int kk = 0; kk < K; ++kk) v[kk & 7] = fma(v[kk & 7], a, b);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0.0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

template <int MODE, int K>
static int run(const char *name, int blocks, double *out, long long *cyc)
{
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k<MODE, K>), dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0, 1e-3);
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    long long cc[2] = {0, 0}; CHK(hipMemcpy(cc, cyc, 16, hipMemcpyDeviceToHost)); const long long c = cc[0];
    printf("%-28s %8.1f us   %7.1f memtime ticks per MFMA (one wavefront's view)   %6.1f ns per MFMA per wavefront   s_memtime runs at %.0f MHz (against s_memrealtime = 100 MHz)\n", name, ms * 1e3, (double)c / N, ms * 1e6 / N, 100.0 * (double)cc[0] / (double)cc[1]);
    return 0;
}

int main(int argc, char **argv)
{
    const int wps = argc > 1 ? atoi(argv[1]) : 1;
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * wps;
    double *out; long long *cyc;
    CHK(hipMalloc(&out, (size_t)blocks * 256 * 8)); CHK(hipMalloc(&cyc, 16));
    printf("%s, %d CUs, %d wavefront(s) per SIMD, %d MFMAs per wavefront, clock %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, wps, N, prop.clockRate / 1000);
    if (run<0, 0>("dep (1 accumulator)", blocks, out, cyc)) return 1;
    if (run<1, 0>("ind4 (4 accumulators)", blocks, out, cyc)) return 1;
    if (run<2, 0>("chain3 + consumer", blocks, out, cyc)) return 1;
    if (run<3, 4>("ind4 + 4 v_fma_f64 each", blocks, out, cyc)) return 1;
    if (run<3, 8>("ind4 + 8 v_fma_f64 each", blocks, out, cyc)) return 1;
    if (run<3, 16>("ind4 + 16 v_fma_f64 each", blocks, out, cyc)) return 1;
    if (run<3, 32>("ind4 + 32 v_fma_f64 each", blocks, out, cyc)) return 1;
    if (run<4, 0>("4x4x4_4b dep", blocks, out, cyc)) return 1;
    if (run<5, 0>("4x4x4_4b ind4", blocks, out, cyc)) return 1;
    if (run<6, 4>("4x4x4_4b ind4 + 4 v_fma_f64", blocks, out, cyc)) return 1;
    if (run<6, 8>("4x4x4_4b ind4 + 8 v_fma_f64", blocks, out, cyc)) return 1;
    if (run<7, 0>("8 v_fma_f64 (per 'MFMA')", blocks, out, cyc)) return 1;
    return 0;
}

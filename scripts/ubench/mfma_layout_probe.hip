// prints the register layout of v_mfma_f64_16x16x4_f64 on this device: which (i,j) of D = A.B each (lane, element) holds,
// assuming A[i][k] at lane i + 16 k and B[k][j] at lane j + 16 k (checked: every D element must be found exactly once)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *out)
{
    const int l = threadIdx.x;
    double4_t acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], acc, 0, 0, 0);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = acc[e];
}
int main()
{
    double A[64], B[64], D[256], *dA, *dB, *dO, O[256];
    for (int i = 0; i < 64; ++i) { A[i] = 1.0 + i * 0.37 + (i % 7) * 0.011; B[i] = 2.0 + i * 0.53 + (i % 5) * 0.007; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; D[i * 16 + j] = s; }
    hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dO, sizeof O);
    hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dO);
    hipMemcpy(O, dO, sizeof O, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l += 1) for (int e = 0; e < 4; ++e) {
        int fi = -1, fj = -1;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (fabs(D[i * 16 + j] - O[l * 4 + e]) < 1e-9 * fabs(D[i * 16 + j])) { fi = i; fj = j; }
        if (fi < 0) bad++;
        if (l < 2 || l == 16 || l == 17 || l == 63) printf("lane %2d elem %d -> D[%d][%d]\n", l, e, fi, fj);
        if (fi >= 0 && !(fj == l % 16 && fi == 4 * (l / 16) + e)) { static int shown = 0; if (shown++ < 3) printf("  (not i = 4*(l/16)+e, j = l%%16)\n"); }
    }
    printf("unmatched: %d\n", bad);
    return 0;
}

// the register layout of v_mfma_f64_4x4x4_4b_f64 on this device (four independent 4 x 4 x 4 products, one D element per lane), found
// without assuming anything: for every pair (la, lb) the instruction runs with A = 1 in lane la only and B = 1 in lane lb only; the lanes
// of D that come back 1 say which output that (A element, B element) pair feeds.  Printed: for each D lane the A lanes and B lanes of
// its four terms.  Also: the same operands through the 16x16x4 instruction -- are the 4-term sums bitwise the same?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(unsigned long long *hit /* [64 la][64 lb] mask of D lanes */)
{
    const int l = threadIdx.x;
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == la ? 1.0 : 0.0, l == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
        const unsigned long long m = __ballot(d != 0.0);
        if (l == 0) hit[la * 64 + lb] = m;
    }
}
__global__ void kv(const double *A, const double *B, double *out, double *out16)
{
    const int l = threadIdx.x;
    out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
    double4_t acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[l], B[l], acc, 0, 0, 0);      // A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16]
    for (int e = 0; e < 4; ++e) out16[l * 4 + e] = acc[e];                      // D[4 (l / 16) + e][l % 16]
}
int main()
{
    static unsigned long long H[64 * 64];
    unsigned long long *dH; hipMalloc(&dH, sizeof H);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dH);
    hipMemcpy(H, dH, sizeof H, hipMemcpyDeviceToHost);
    int terms[64] = {0}, ta[64][8], tb[64][8];
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) for (int d = 0; d < 64; ++d) if ((H[la * 64 + lb] >> d) & 1ull) { if (terms[d] < 8) { ta[d][terms[d]] = la; tb[d][terms[d]] = lb; } terms[d]++; }
    for (int d = 0; d < 64; ++d) {
        printf("D lane %2d: %d terms:", d, terms[d]);
        for (int t = 0; t < terms[d] && t < 8; ++t) printf("  A%-2d.B%-2d", ta[d][t], tb[d][t]);
        printf("\n");
    }
    double A[64], B[64], O[64], O16[256], *dA, *dB, *dO, *dO16;
    for (int i = 0; i < 64; ++i) { A[i] = 1.0 + i * 0.37 + (i % 7) * 0.011 + 1e-9 * i * i; B[i] = 2.0 + i * 0.53 + (i % 5) * 0.007 + 3e-10 * i * i * i; }
    hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dO, sizeof O); hipMalloc(&dO16, sizeof O16);
    hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kv, dim3(1), dim3(64), 0, 0, dA, dB, dO, dO16);
    hipMemcpy(O, dO, sizeof O, hipMemcpyDeviceToHost); hipMemcpy(O16, dO16, sizeof O16, hipMemcpyDeviceToHost);
    // every 4x4x4 output whose four (A lane, B lane) pairs are those of a 16x16x4 output D[i][j] = sum_k A[i + 16 k] B[j + 16 k]: same bits?
    int cmp = 0, same = 0;
    for (int d = 0; d < 64; ++d) {
        if (terms[d] != 4) continue;
        const int i = ta[d][0] % 16, j = tb[d][0] % 16;
        bool ok = true;
        for (int t = 0; t < 4; ++t) ok = ok && ta[d][t] % 16 == i && tb[d][t] % 16 == j && ta[d][t] / 16 == tb[d][t] / 16;
        if (!ok) continue;
        const double v16 = O16[(j + 16 * (i / 4)) * 4 + i % 4];
        cmp++; same += (O[d] == v16);
    }
    printf("%d outputs are also outputs of the 16x16x4 instruction on the same registers; %d of them bitwise equal\n", cmp, same);
    return 0;
}

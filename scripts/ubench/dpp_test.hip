#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *out)
{
    int x = threadIdx.x * 10;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x130, 0xf, 0xf, false);        // wave_shl:1
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x138, 0xf, 0xf, false);   // wave_shr:1
    out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x101, 0xf, 0xf, false);  // row_shl:1
    out[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x134, 0xf, 0xf, false);  // wave_rol:1
}
int main()
{
    int *d, h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *nm[4] = {"wave_shl:1", "wave_shr:1", "row_shl:1", "wave_rol:1"};
    for (int r = 0; r < 4; ++r) { printf("%s:", nm[r]); for (int i = 0; i < 64; ++i) printf(" %d", h[r * 64 + i]); printf("\n"); }
    return 0;
}

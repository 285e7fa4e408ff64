// grid_barrier.hip -- cost of an in-kernel grid barrier on MI355X for the geometry of the NMFT update kernel
// (625 workgroups x 256 threads, ~45 KB of LDS each: 3 per CU), in isolation:
//   A  two-level counter barrier: arrival on one of 8 group counters (group = workgroup % 8), the last arriver of a group
//      arrives on a top counter, the last of those publishes the generation word every workgroup polls (one lane, relaxed
//      agent-scope loads + s_sleep)
//   B  the same + what one NMFT update moves through it: every workgroup publishes 521 doubles (write-through stores), barrier,
//      workgroup w reduces output column w over all workgroups (coalesced reads of 625 doubles), publishes one double, barrier,
//      every workgroup reads the 521 totals
// build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip ; run: ./grid_barrier [workgroups] [iterations]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Bar { unsigned *gcnt /* [8 * 16] */, *top, *gen, *timeout; int members[8]; };

__device__ __forceinline__ bool grid_barrier(const Bar &b, unsigned epoch, int tid)
{
    __syncthreads();
    bool ok = true;
    if (tid == 0) {
        const unsigned g = blockIdx.x & 7u;
        const unsigned t = __hip_atomic_fetch_add((gu32 *)(b.gcnt + g * 16), 1u, RLX_AGENT);
        if (t + 1u == epoch * (unsigned)b.members[g]) {
            const unsigned t2 = __hip_atomic_fetch_add((gu32 *)b.top, 1u, RLX_AGENT);
            if (t2 + 1u == epoch * 8u) __hip_atomic_store((gu32 *)b.gen, epoch, RLX_AGENT);
        }
        unsigned spins = 0;
        while (__hip_atomic_load((gu32 *)b.gen, RLX_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) { __hip_atomic_store((gu32 *)b.timeout, 1u, RLX_AGENT); ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256, 3) void bar_only(Bar b, int iters)
{
    extern __shared__ char smem[];
    smem[threadIdx.x] = 0;
    for (int it = 0; it < iters; ++it) grid_barrier(b, (unsigned)it + 1u, threadIdx.x);
}

// the NMFT exchange: part [nout][nwg] doubles, stat [nout]
__global__ __launch_bounds__(256, 3) void bar_exchange(Bar b, int iters, int nout, double *part, double *stat, double *check, int mode)
{
    extern __shared__ char smem[];
    double *ls = reinterpret_cast<double *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nwg = gridDim.x, wg = blockIdx.x;
    double acc = 0.0;
    unsigned epoch = 0;
    for (int it = 0; it < iters; ++it) {
        if (mode & 1) for (int o = tid; o < nout; o += 256)
            __hip_atomic_store((gu64 *)(part + (size_t)o * nwg + wg), (unsigned long long)__double_as_longlong((double)(wg + o + it)), RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        grid_barrier(b, ++epoch, tid);
        const int out = wg * 4 + wv;
        if ((mode & 2) && out < nout) {
            double a = 0.0;
            for (int k = lane; k < nwg; k += 64)
                a += __longlong_as_double((long long)__hip_atomic_load((gu64 *)(part + (size_t)out * nwg + k), RLX_AGENT));
            for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
            if (lane == 0) __hip_atomic_store((gu64 *)(stat + out), (unsigned long long)__double_as_longlong(a), RLX_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (mode & 8) grid_barrier(b, ++epoch, tid);
        if (mode & 4) for (int o = tid; o < nout; o += 256) ls[o] = __longlong_as_double((long long)__hip_atomic_load((gu64 *)(stat + o), RLX_AGENT));
        __syncthreads();
        acc += ls[(it * 7 + tid) % nout];
        __syncthreads();
    }
    if (tid == 0 && wg == 0) {
        // last iteration's column 0 total must be sum_wg (wg + 0 + iters - 1)
        *check = ls[0];
    }
    if (acc == -1.0) stat[0] = acc;
}

int main(int argc, char **argv)
{
    const int nwg = argc > 1 ? atoi(argv[1]) : 625, iters = argc > 2 ? atoi(argv[2]) : 2000, nout = 521;
    const size_t lds = 45 * 1024;
    unsigned *d_state;
    CHECK(hipMalloc(&d_state, 4096));
    Bar b;
    b.gcnt = d_state; b.top = d_state + 8 * 16; b.gen = d_state + 9 * 16; b.timeout = d_state + 10 * 16;
    for (int g = 0; g < 8; ++g) b.members[g] = (nwg - g + 7) / 8;
    double *part, *stat, *check;
    CHECK(hipMalloc(&part, sizeof(double) * nout * nwg)); CHECK(hipMalloc(&stat, sizeof(double) * nout)); CHECK(hipMalloc(&check, 8));
    int occ = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, bar_only, 256, lds));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("workgroups %d, resident capacity %d x %d\n", nwg, occ, prop.multiProcessorCount);
    if (nwg > occ * prop.multiProcessorCount) { printf("grid does not fit\n"); return 1; }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(d_state, 0, 4096));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bar_only, dim3(nwg), dim3(256), lds, 0, b, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h[176]; CHECK(hipMemcpy(h, d_state, sizeof h, hipMemcpyDeviceToHost));
        printf("A barrier only: %.2f us per barrier (timeout flag %u)\n", 1e3 * ms / iters, h[160]);
    }
    for (int mode : {15, 15, 9, 11, 14, 13, 8, 0}) {
        CHECK(hipMemset(d_state, 0, 4096));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bar_exchange, dim3(nwg), dim3(256), lds, 0, b, iters, nout, part, stat, check, mode);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double c; CHECK(hipMemcpy(&c, check, 8, hipMemcpyDeviceToHost));
        double want = 0; for (int w = 0; w < nwg; ++w) want += w + iters - 1;
        unsigned h[176]; CHECK(hipMemcpy(h, d_state, sizeof h, hipMemcpyDeviceToHost));
        printf("B mode %2d (1 publish | 2 reduce | 4 read | 8 second barrier): %.2f us per update; check %s (timeout flag %u)\n",
               mode, 1e3 * ms / iters, c == want ? "ok" : "n/a", h[160]);
    }
    return 0;
}

// kernels_nmft.hip -- gfx950 kernels of the NMF-tensor initialiser (desman/Init_NMFT.py).
//
// Device layout (HBM): F [V][4][S] f64, tau [V][4][G] f64, gamma [G][S] f64
// (the reference's row index v + a*V becomes (v, a); dsm_nmft_set/get convert).
//
// One multiplicative update (Init_NMFT.py:158-181) is two streaming passes
// over F plus a small reduction/update kernel:
//   pass A  : R = tau.gamma, objective (:152-156), Q = F (/) R, gamma numerators
//             tau^T.Q and H1 = colsum(tau), reduced per workgroup (fixed order)
//   gamma   : cross-workgroup reduction, convergence test of the factorize loop
//             (:106) ON THE DEVICE (no host sync per iteration), gamma update (:163-166)
//   pass B  : R' = tau.gamma_new, Q' = F (/) R', tau *= (Q'.gamma^T) (/) rowsum(gamma),
//             per-(v,g) renormalisation over the four bases (:170-181), _adjustment (:88-91)
#include "dsm_device.h"
#include "dsm_host.h"

#define NMFT_CTL(c) ((c)->nstat + (size_t)(c)->nG * (c)->S + 2 * (c)->nG)

__device__ __forceinline__ double nzd(double x) { return x == 0.0 ? DSM_EPS : x; }   // du.elop

// F[v][a][s] = (x_vsa + 1) / (n_vs + 4)          (Init_NMFT.py:49-60)
__global__ __launch_bounds__(256) void nmft_freq_kernel(const int32_t *__restrict__ cnt_vs, double *__restrict__ F,
                                                        int V, int S)
{
    const size_t n = (size_t)V * S;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int v = (int)(i / S), s = (int)(i % S);
        const int4 c = reinterpret_cast<const int4 *>(cnt_vs)[i];
        const double x[4] = {(double)c.x + 1.0, (double)c.y + 1.0, (double)c.z + 1.0, (double)c.w + 1.0};
        const double tot = ((x[0] + x[1]) + x[2]) + x[3];
#pragma unroll
        for (int a = 0; a < 4; ++a) F[((size_t)v * 4 + a) * S + s] = x[a] / tot;
    }
}

__global__ void clamp_min_kernel(double *__restrict__ x, size_t n, double lo)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (x[i] < lo) x[i] = lo;
}

// ---------------------------------------------------------------------------
// pass A.  thread = (row group, sample); SPAD lanes per row; a row group walks
// rows n = (v,a) with a grid stride.  Per-lane accumulators: G numerators for
// its sample, the objective, H1.  Partials layout per workgroup:
//   [G*S numerators][G H1][1 objective]
// ---------------------------------------------------------------------------
template <int GMAX>
__global__ __launch_bounds__(256) void nmft_pass_a_kernel(const double *__restrict__ F, const double *__restrict__ tau,
                                                          const double *__restrict__ gam, int V, int S, int G,
                                                          int SPAD, const double *__restrict__ ctl,
                                                          double *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) char smem_a[];
    if (ctl[2] != 0.0) return;                         // factorize loop already stopped
    double *red = reinterpret_cast<double *>(smem_a);  // [RG][GMAX + 2][SPAD]
    const int tid = threadIdx.x;
    const int RG = 256 / SPAD;
    const int rg = tid / SPAD, sl = tid % SPAD;
    const size_t N = (size_t)4 * V;
    const size_t pstride = (size_t)G * S + G + 1;
    double *mypart = partial + (size_t)blockIdx.x * pstride;
    double obj = 0.0;
    double h1[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) h1[g] = 0.0;

    for (int s0 = 0; s0 < S; s0 += SPAD) {
        const int s = s0 + sl;
        const bool live = s < S;
        double gc[GMAX], num[GMAX];
#pragma unroll
        for (int g = 0; g < GMAX; ++g) { gc[g] = (live && g < G) ? gam[(size_t)g * S + s] : 0.0; num[g] = 0.0; }
        for (size_t n = (size_t)blockIdx.x * RG + rg; n < N; n += (size_t)gridDim.x * RG) {
            const double *trow = tau + n * G;
            double tv[GMAX];
            double r = 0.0;
#pragma unroll
            for (int g = 0; g < GMAX; ++g) {
                tv[g] = (g < G) ? trow[g] : 0.0;
                r = fma(tv[g], gc[g], r);
            }
            if (s0 == 0) {
#pragma unroll
                for (int g = 0; g < GMAX; ++g) h1[g] += tv[g];
            }
            if (live) {
                const double f = F[n * S + s];
                const double pa = r < DSM_EPS ? DSM_EPS : r;
                obj += f * log(nzd(f) / pa) - f + pa;
                const double q = nzd(f) / nzd(r);
#pragma unroll
                for (int g = 0; g < GMAX; ++g) num[g] = fma(tv[g], q, num[g]);
            }
        }
        // cross-row-group reduction (fixed order) -> partial numerators of this chunk
#pragma unroll
        for (int g = 0; g < GMAX; ++g) red[((size_t)rg * (GMAX + 2) + g) * SPAD + sl] = num[g];
        __syncthreads();
        if (rg == 0 && live) {
            for (int g = 0; g < G; ++g) {
                double a = 0.0;
                for (int k = 0; k < RG; ++k) a += red[((size_t)k * (GMAX + 2) + g) * SPAD + sl];
                mypart[(size_t)g * S + s] = a;
            }
        }
        __syncthreads();
    }
    // objective + H1
    red[((size_t)rg * (GMAX + 2) + GMAX) * SPAD + sl] = obj;
    if (sl == 0) {
#pragma unroll
        for (int g = 0; g < GMAX; ++g) red[((size_t)rg * (GMAX + 2) + GMAX + 1) * SPAD + g] = h1[g];   // SPAD >= 16 >= ... see launcher
    }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0;
        for (int k = 0; k < RG; ++k)
            for (int j = 0; j < SPAD; ++j) a += red[((size_t)k * (GMAX + 2) + GMAX) * SPAD + j];
        mypart[(size_t)G * S + G] = a;
    }
    if (tid < G) {
        double a = 0.0;
        for (int k = 0; k < RG; ++k) a += red[((size_t)k * (GMAX + 2) + GMAX + 1) * SPAD + tid];
        mypart[(size_t)G * S + tid] = a;
    }
}

// ---------------------------------------------------------------------------
// gamma / control kernel.  Workgroup j owns samples [j*SB, (j+1)*SB); every
// workgroup re-derives the (identical) loop decision from the same partials,
// workgroup 0 records it.  ctl: [0] div  [2] done  [3] updates run
// [4 + (it&1)] div of iteration it (parity slots: no intra-launch race).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nmft_gamma_kernel(const double *__restrict__ partial, int nblk, int S, int G,
                                                         int it, int max_iter, double min_change, int fix_gamma,
                                                         int adjust, double *__restrict__ gam, double *__restrict__ ctl)
{
    __shared__ double red[256];
    __shared__ double h1s[DSM_MAX_G];
    __shared__ double gnew[256];
    __shared__ int go;
    if (ctl[2] != 0.0) return;
    const int tid = threadIdx.x;
    const size_t pstride = (size_t)G * S + G + 1;
    // objective: fixed-order tree over the workgroup partials
    double a = 0.0;
    for (int b = tid; b < nblk; b += 256) a += partial[(size_t)b * pstride + (size_t)G * S + G];
    red[tid] = a;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const double div = red[0];
    if (tid < G) {
        double h = 0.0;
        for (int b = 0; b < nblk; ++b) h += partial[(size_t)b * pstride + (size_t)G * S + tid];
        h1s[tid] = h;
    }
    if (tid == 0) {
        const double prev = (it == 0) ? 0.0 : ctl[4 + ((it - 1) & 1)];
        go = (it < max_iter) && (fabs(prev - div) > min_change);         // Init_NMFT.py:106
        if (blockIdx.x == 0) {
            ctl[0] = div;
            ctl[4 + (it & 1)] = div;
            ctl[3] = (double)it;
            if (!go) ctl[2] = 1.0;
        }
    }
    __syncthreads();
    if (!go || fix_gamma) return;
    // gamma update for this workgroup's samples: thread = (s_local, g), g fastest
    const int SB = 256 / DSM_MAX_G;
    const int sloc = tid / DSM_MAX_G, g = tid % DSM_MAX_G;
    const int s = blockIdx.x * SB + sloc;
    double val = 0.0;
    if (s < S && g < G) {
        if (G > 1) {
            double num = 0.0;
            for (int b = 0; b < nblk; ++b) num += partial[(size_t)b * pstride + (size_t)g * S + s];
            val = gam[(size_t)g * S + s] * (nzd(num) / nzd(h1s[g]));       // :163
        } else {
            val = 1.0;                                                   // :168
        }
    }
    gnew[tid] = val;
    __syncthreads();
    if (s < S && g < G) {
        if (G > 1) {
            double tot = 0.0;
            for (int k = 0; k < G; ++k) tot += gnew[sloc * DSM_MAX_G + k];   // :165
            val = val / tot;                                                 // :166
        }
        if (adjust && val < DSM_EPS) val = DSM_EPS;                          // :91
        gam[(size_t)g * S + s] = val;
    }
}

// ---------------------------------------------------------------------------
// pass B.  A workgroup takes VT variants (4*VT rows) at a time:
//   step 1  threads <-> (row, s):  Q'[row][s] = F (/) (tau.gamma)        -> LDS
//   step 2  threads <-> (row, g):  num[row][g] = sum_s Q'[row][s] gamma[g][s]
//   step 3  threads <-> (v, g):    tau update, renormalise over a, clamp
// gamma [G][S+1] and the tau tile live in LDS; tau1[g] = rowsum(gamma) is
// recomputed per workgroup (G*S adds).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nmft_pass_b_kernel(const double *__restrict__ F, double *__restrict__ tau,
                                                          const double *__restrict__ gam, int V, int S, int G, int VT,
                                                          int adjust, const double *__restrict__ ctl)
{
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    if (ctl[2] != 0.0) return;
    const int SP = S + 1, tid = threadIdx.x;
    const int rows = 4 * VT;
    double *gs = reinterpret_cast<double *>(smem_b);     // [G][SP]
    double *t1 = gs + (size_t)G * SP;                     // [G]
    double *ts = t1 + G;                                  // [rows][G]
    double *nm = ts + (size_t)rows * G;                   // [rows][G]
    double *qs = nm + (size_t)rows * G;                   // [rows][SP]
    for (int i = tid; i < G * S; i += 256) gs[(i / S) * SP + (i % S)] = gam[i];
    __syncthreads();
    if (tid < G) {
        double a = 0.0;
        for (int s = 0; s < S; ++s) a += gs[tid * SP + s];    // gamma.sum(1)  (:170)
        t1[tid] = a;
    }
    for (int v0 = blockIdx.x * VT; v0 < V; v0 += gridDim.x * VT) {
        const int nv = (V - v0 < VT) ? V - v0 : VT;
        const int nr = 4 * nv;
        __syncthreads();
        for (int i = tid; i < nr * G; i += 256) ts[i] = tau[(size_t)v0 * 4 * G + i];
        __syncthreads();
        for (int i = tid; i < nr * S; i += 256) {
            const int r = i / S, s = i % S;
            double acc = 0.0;
            for (int g = 0; g < G; ++g) acc = fma(ts[r * G + g], gs[g * SP + s], acc);
            qs[r * SP + s] = nzd(F[((size_t)v0 * 4 + r) * S + s]) / nzd(acc);
        }
        __syncthreads();
        for (int i = tid; i < nr * G; i += 256) {
            const int r = i / G, g = i % G;
            double acc = 0.0;
            for (int s = 0; s < S; ++s) acc = fma(qs[r * SP + s], gs[g * SP + s], acc);
            nm[i] = acc;
        }
        __syncthreads();
        for (int i = tid; i < nv * G; i += 256) {
            const int vl = i / G, g = i % G;
            double tn[4], tot = 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int r = vl * 4 + a;
                tn[a] = ts[r * G + g] * (nzd(nm[r * G + g]) / nzd(t1[g]));     // :171-172
                tot += tn[a];                                                    // :176-178
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                double x = tn[a] / tot;                                          // :180-181
                if (adjust && x < DSM_EPS) x = DSM_EPS;
                tau[((size_t)(v0 + vl) * 4 + a) * G + g] = x;
            }
        }
    }
}

// get_tau (Init_NMFT.py:230-245): strict '>' against a running max from 0.0
__global__ void nmft_get_tau_kernel(const double *__restrict__ tau, int V, int G, uint64_t *__restrict__ packed)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    uint64_t t = 0;
    for (int g = 0; g < G; ++g) {
        double best = 0.0; int arg = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double x = tau[((size_t)v * 4 + a) * G + g];
            if (x > best) { best = x; arg = a; }
        }
        t |= (uint64_t)arg << (2 * g);
    }
    packed[v] = t;
}

// ---------------------------------------------------------------------------
int nmft_grid(dsm_ctx *c)
{
    const size_t N = (size_t)4 * c->V;
    size_t g = (N + 31) / 32;
    if (g > 256) g = 256;
    if (g < 1) g = 1;
    return (int)g;
}

int k_nmft_freq(dsm_ctx *c)
{
    const size_t n = (size_t)c->V * c->S;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(nmft_freq_kernel, dim3(grid), dim3(256), 0, c->stream, c->cnt_vs, c->F, c->V, c->S);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

int k_nmft_clamp(dsm_ctx *c)
{
    hipLaunchKernelGGL(clamp_min_kernel, dim3(256), dim3(256), 0, c->stream, c->ntau, (size_t)c->V * 4 * c->nG, DSM_EPS);
    hipLaunchKernelGGL(clamp_min_kernel, dim3(8), dim3(256), 0, c->stream, c->ngam, (size_t)c->nG * c->S, DSM_EPS);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

static int spad_for(int S)
{
    int p = 32;                      // >= DSM_MAX_G lanes are needed for the H1 hand-off
    while (p < S && p < 256) p <<= 1;
    return p;
}

int k_nmft_pass_a(dsm_ctx *c)
{
    KTimer tm(c, DSM_K_NMFT_A);
    const int G = c->nG, S = c->S, SPAD = spad_for(S);
#define LAUNCH_A(GM)                                                                                              \
    hipLaunchKernelGGL(nmft_pass_a_kernel<GM>, dim3(c->nmft_blocks), dim3(256),                                   \
                       (size_t)(256 / SPAD) * (GM + 2) * SPAD * sizeof(double), c->stream, c->F, c->ntau, c->ngam, \
                       c->V, S, G, SPAD, NMFT_CTL(c), c->npart)
    if (G <= 4) LAUNCH_A(4);
    else if (G <= 8) LAUNCH_A(8);
    else if (G <= 16) LAUNCH_A(16);
    else LAUNCH_A(32);
#undef LAUNCH_A
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

int k_nmft_gamma(dsm_ctx *c, int it, int max_iter, double min_change, int fix_gamma, int adjust)
{
    KTimer tm(c, DSM_K_NMFT_G);
    const int SB = 256 / DSM_MAX_G;
    const int grid = (c->S + SB - 1) / SB;
    hipLaunchKernelGGL(nmft_gamma_kernel, dim3(grid), dim3(256), 0, c->stream, c->npart, c->nmft_blocks, c->S, c->nG,
                       it, max_iter, min_change, fix_gamma, adjust, c->ngam, NMFT_CTL(c));
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

int k_nmft_pass_b(dsm_ctx *c, int adjust)
{
    KTimer tm(c, DSM_K_NMFT_B);
    const int G = c->nG, S = c->S, SP = S + 1;
    int VT = 8192 / (4 * SP);
    if (VT > 16) VT = 16;
    if (VT < 1) VT = 1;
    const size_t sh = ((size_t)G * SP + G + 2 * (size_t)4 * VT * G + (size_t)4 * VT * SP) * sizeof(double);
    if (sh > 160 * 1024) { dsm_set_error("NMFT tile (%zu B) exceeds LDS", sh); return DSM_ERR_UNSUPPORTED; }
    int grid = (c->V + VT - 1) / VT;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(nmft_pass_b_kernel, dim3(grid), dim3(256), sh, c->stream, c->F, c->ntau, c->ngam, c->V, S, G, VT,
                       adjust, NMFT_CTL(c));
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

int k_nmft_get_tau(dsm_ctx *c, uint64_t *d_packed)
{
    hipLaunchKernelGGL(nmft_get_tau_kernel, dim3((c->V + 255) / 256), dim3(256), 0, c->stream, c->ntau, c->V, c->nG,
                       d_packed);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

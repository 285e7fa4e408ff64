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
//   pass B  : R' = tau.gamma_new (the normalised gamma BEFORE _adjustment, as in the reference),
//             Q' = F (/) R', tau *= (Q'.gamma^T) (/) rowsum(gamma),
//             per-(v,g) renormalisation over the four bases (:170-181), _adjustment (:88-91)
#include <stdlib.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>

#include "dsm_device.h"
#include "dsm_host.h"
#include "log_table.h"

#define NMFT_CTL(c) ((c)->nstat + (size_t)(c)->nG * (c)->S + 2 * (c)->nG)

// du.elop's zero rule: x == 0 ? eps : x.  eps = 2^-52 = 0x3CB00000'00000000 and a zero's low word is zero already, so one select on the high word does it
__device__ __forceinline__ double nzd(double x)
{
    static_assert(DSM_EPS == 0x1p-52, "nzd: DSM_EPS is not 2^-52");
    return __hiloint2double(x == 0.0 ? 0x3CB00000 : __double2hiint(x), __double2loint(x));
}

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
// pass A.  1024-thread workgroups (16 wavefronts, 4 per SIMD: the pass is
// latency-bound, not ALU-bound); thread = (row group, sample), SPAD lanes per
// row; a row group walks rows n = (v,a) with a grid stride.  The tau row of
// the wavefront's current n is wave-uniform: it is fetched through scalar
// loads (readfirstlane'd row index).  Per-lane accumulators: G numerators for
// its sample, the objective, H1.  Partials are written TRANSPOSED,
// partial[out][workgroup], out in [G*S numerators][G H1][1 objective], so the
// reduction kernel reads them coalesced.
// ---------------------------------------------------------------------------
// (workgroup size shrinks with GMAX so that the per-lane accumulators stay in registers)
template <int GMAX, int NMFT_A_THREADS>
__global__ __launch_bounds__(NMFT_A_THREADS) void nmft_pass_a_kernel(const double *__restrict__ F,
                                                                     const double *__restrict__ tau,
                                                                     const double *__restrict__ gam, int V, int S, int G,
                                                                     int SPAD, const double *__restrict__ ctl,
                                                                     const double *__restrict__ log_tab,
                                                                     double *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) char smem_a[];
    if (ctl[2] != 0.0) return;                         // factorize loop already stopped
    double2 *ltab = reinterpret_cast<double2 *>(smem_a);                    // [256]
    double *red = reinterpret_cast<double *>(smem_a) + 2 * DSM_LOG_TAB_N;   // [RG][GMAX + 2][SPAD]
    const int tid = threadIdx.x;
    if (tid < DSM_LOG_TAB_N) ltab[tid] = reinterpret_cast<const double2 *>(log_tab)[tid];
    __syncthreads();
    const int RG = NMFT_A_THREADS / SPAD;
    const int rg = tid / SPAD, sl = tid % SPAD;
    const size_t N = (size_t)4 * V;
    const int nblk = gridDim.x;
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
        // a row group takes the 4 rows of one variant per step: the 4 F loads (and the 4 scalar
        // tau-row loads) are independent and issued together -> 4x the memory-level parallelism
        for (size_t nb = ((size_t)blockIdx.x * RG + rg) * 4; nb < N; nb += (size_t)nblk * RG * 4) {
            double f[4];
            double tv[4][GMAX];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t n = nb + u;
                // rows are wave-uniform whenever a row spans >= 64 lanes: scalar-load the tau row
                const double *trow = tau + ((SPAD >= 64) ? (size_t)__builtin_amdgcn_readfirstlane((int)n) : n) * G;
#pragma unroll
                for (int g = 0; g < GMAX; ++g) tv[u][g] = (g < G) ? trow[g] : 0.0;
                f[u] = live ? F[n * S + s] : 1.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double r = 0.0;
#pragma unroll
                for (int g = 0; g < GMAX; ++g) r = fma(tv[u][g], gc[g], r);
                if (s0 == 0) {
#pragma unroll
                    for (int g = 0; g < GMAX; ++g) h1[g] += tv[u][g];
                }
                if (live) {
                    const double pa = r < DSM_EPS ? DSM_EPS : r;
                    obj += f[u] * dsm_log(nzd(f[u]) / pa, ltab) - f[u] + pa;
                    const double q = nzd(f[u]) / nzd(r);
#pragma unroll
                    for (int g = 0; g < GMAX; ++g) num[g] = fma(tv[u][g], q, num[g]);
                }
            }
        }
        // cross-row-group reduction (fixed order) -> partial numerators of this chunk
#pragma unroll
        for (int g = 0; g < GMAX; ++g) red[((size_t)rg * (GMAX + 2) + g) * SPAD + sl] = num[g];
        __syncthreads();
        for (int i = tid; i < G * SPAD; i += NMFT_A_THREADS) {
            const int g = i / SPAD, ss = i % SPAD;
            if (s0 + ss < S) {
                double a = 0.0;
                for (int k = 0; k < RG; ++k) a += red[((size_t)k * (GMAX + 2) + g) * SPAD + ss];
                partial[((size_t)g * S + s0 + ss) * nblk + blockIdx.x] = a;
            }
        }
        __syncthreads();
    }
    // objective + H1
    red[((size_t)rg * (GMAX + 2) + GMAX) * SPAD + sl] = obj;
    if (sl == 0) {
#pragma unroll
        for (int g = 0; g < GMAX; ++g) red[((size_t)rg * (GMAX + 2) + GMAX + 1) * SPAD + g] = h1[g];   // SPAD >= 32 >= GMAX
    }
    __syncthreads();
    if (tid < 64) {                                    // one wavefront: fixed-order butterfly over RG*SPAD terms
        double a = 0.0;
        for (int i = tid; i < RG * SPAD; i += 64) a += red[((size_t)(i / SPAD) * (GMAX + 2) + GMAX) * SPAD + (i % SPAD)];
        a = group_allreduce_sum<64>(a);
        if (tid == 0) partial[((size_t)G * S + G) * nblk + blockIdx.x] = a;
    } else if (tid < 64 + G) {
        const int g = tid - 64;
        double a = 0.0;
        for (int k = 0; k < RG; ++k) a += red[((size_t)k * (GMAX + 2) + GMAX + 1) * SPAD + g];
        partial[((size_t)G * S + g) * nblk + blockIdx.x] = a;
    }
}

// ---------------------------------------------------------------------------
// reduction of the transposed partials: one wavefront per output, coalesced
// reads, fixed-order butterfly -> stat[out].
// ---------------------------------------------------------------------------
// THE summation order of a statistic over the workgroup partials of an update kernel, shared by every reduction of this file
// (nmft_reduce_kernel, nmft_rg_kernel, nmft_persist_kernel), so that all paths produce the same bits: consecutive TRIPLES of
// partials first, (p[3t] + p[3t+1]) + p[3t+2] -- a workgroup of the persistent kernel covers the quads of three workgroups of the
// update kernel and publishes exactly that sum -- then the triples lane-strided, then a fixed-order butterfly.
__device__ __forceinline__ double nmft_sum_partials(const double *__restrict__ p, int nblk, int lane)
{
    const int ntrip = (nblk + 2) / 3;
    double a = 0.0;
    for (int t = lane; t < ntrip; t += 64) {
        const int b = 3 * t;
        double x = p[b];
        if (b + 1 < nblk) x += p[b + 1];
        if (b + 2 < nblk) x += p[b + 2];
        a += x;
    }
    return group_allreduce_sum<64>(a);
}

struct NmftReduceParams { const double *partial; int nblk, nout; const double *ctl; double *stat; };
__device__ __forceinline__ void nmft_reduce_body(const NmftReduceParams &q)
{
    const double *__restrict__ partial = q.partial, *__restrict__ ctl = q.ctl;
    double *__restrict__ stat = q.stat;
    const int nblk = q.nblk, nout = q.nout;
    if (ctl[2] != 0.0) return;
    const int out = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (out >= nout) return;
    const double a = nmft_sum_partials(partial + (size_t)out * nblk, nblk, lane);
    if (lane == 0) stat[out] = a;
}
__global__ __launch_bounds__(256) void nmft_reduce_kernel(NmftReduceParams q) { nmft_reduce_body(q); }
__global__ __launch_bounds__(256) void nmft_reduce_kernel_b(BatchArgs<NmftReduceParams> b) { nmft_reduce_body(b.p[blockIdx.y]); }

// ---------------------------------------------------------------------------
// gamma / control kernel (one workgroup): the stop test of the factorize loop
// (Init_NMFT.py:106) on the device, then the gamma update (:163-168).
// ctl: [0] div  [2] done  [3] updates run  [4 + (it&1)] div of iteration it  [6] iteration counter.
// ---------------------------------------------------------------------------
struct NmftGammaParams {
    const double *stat; int S, G, max_iter; double min_change; int fix_gamma, adjust;
    double *gam, *gam_raw, *ctl, *div_trace;
};
__device__ __forceinline__ void nmft_gamma_body(const NmftGammaParams &q)
{
    const double *__restrict__ stat = q.stat;
    const int S = q.S, G = q.G, max_iter = q.max_iter, fix_gamma = q.fix_gamma, adjust = q.adjust;
    const double min_change = q.min_change;
    double *__restrict__ gam = q.gam, *__restrict__ gam_raw = q.gam_raw, *__restrict__ ctl = q.ctl, *__restrict__ div_trace = q.div_trace;
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    double *val = reinterpret_cast<double *>(smem_g);          // [G][SC] one chunk of sample columns
    __shared__ int go;
    if (ctl[2] != 0.0) return;
    const int tid = threadIdx.x;
    if (tid == 0) {
        // the iteration index is a device word, so every iteration is the SAME launch and a batch of
        // iterations can be replayed as one hipGraph
        const int it = (int)ctl[6];
        ctl[6] = (double)(it + 1);
        const double div = stat[(size_t)G * S + G];
        const double prev = (it == 0) ? 0.0 : ctl[4 + ((it - 1) & 1)];
        go = (it < max_iter) && (fabs(prev - div) > min_change);         // Init_NMFT.py:106
        ctl[0] = div;
        ctl[4 + (it & 1)] = div;
        ctl[3] = (double)it;
        if (!go) ctl[2] = 1.0;
        if (div_trace) div_trace[it] = div;
    }
    __syncthreads();
    if (!go || fix_gamma) return;
    const int SC = 1024 / G;                                             // sample columns per chunk
    for (int s0 = 0; s0 < S; s0 += SC) {
        const int g = tid / SC, s = s0 + tid % SC;
        const bool live = g < G && s < S;
        double v = 1.0;                                                  // :168
        if (live && G > 1) v = gam[(size_t)g * S + s] * (nzd(stat[(size_t)g * S + s]) / nzd(stat[(size_t)G * S + g]));   // :163
        if (g < G) val[g * SC + tid % SC] = v;
        __syncthreads();
        if (live) {
            if (G > 1) {
                double tot = 0.0;
                for (int k = 0; k < G; ++k) tot += val[k * SC + tid % SC];   // :165
                v = v / tot;                                                 // :166
            }
            gam_raw[(size_t)g * S + s] = v;                                  // the tau update of this iteration sees it unclamped
            if (adjust && v < DSM_EPS) v = DSM_EPS;                          // _adjustment follows the whole div_update (:88-91,:108)
            gam[(size_t)g * S + s] = v;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void nmft_gamma_kernel(NmftGammaParams q) { nmft_gamma_body(q); }
__global__ __launch_bounds__(1024) void nmft_gamma_kernel_b(BatchArgs<NmftGammaParams> b) { nmft_gamma_body(b.p[blockIdx.y]); }

// gamma fixed (factorize_tau, and the stand-alone objective): of the statistics only the objective is wanted, so the reduction of
// the partials and the control step are ONE launch of one wavefront -- the objective's partials summed as every reduction of this
// file sums them (nmft_sum_partials), then the stop test of nmft_gamma_body.  An update is then two launches, not three.
struct NmftObjCtlParams { const double *partial; int nblk, nout, max_iter; double min_change; double *stat, *ctl, *div_trace; };
__device__ __forceinline__ void nmft_objctl_body(const NmftObjCtlParams &q)
{
    double *__restrict__ ctl = q.ctl;
    if (ctl[2] != 0.0) return;
    const int lane = threadIdx.x & 63;
    const double div = nmft_sum_partials(q.partial + (size_t)(q.nout - 1) * q.nblk, q.nblk, lane);
    if (lane == 0) {
        q.stat[q.nout - 1] = div;
        const int it = (int)ctl[6];
        ctl[6] = (double)(it + 1);
        const double prev = (it == 0) ? 0.0 : ctl[4 + ((it - 1) & 1)];
        const bool go = (it < q.max_iter) && (fabs(prev - div) > q.min_change);     // Init_NMFT.py:140
        ctl[0] = div;
        ctl[4 + (it & 1)] = div;
        ctl[3] = (double)it;
        if (!go) ctl[2] = 1.0;
        else ctl[10] = 1.0 - ctl[10];                  // the fused pass's candidate rows become the current ones (NmftMfmaParams.fix_gamma == 2; unused otherwise)
        if (q.div_trace) q.div_trace[it] = div;
    }
}
__global__ __launch_bounds__(64) void nmft_objctl_kernel(NmftObjCtlParams q) { nmft_objctl_body(q); }
__global__ __launch_bounds__(64) void nmft_objctl_kernel_b(BatchArgs<NmftObjCtlParams> b) { nmft_objctl_body(b.p[blockIdx.y]); }

// ---------------------------------------------------------------------------
// reduce + gamma / control in ONE launch of NMFT_RG_WGS workgroups (an update is then two dependent launches, not three).
// Workgroup b owns the sample columns [b cw, (b+1) cw): it sums the workgroup partials of the statistics it needs -- its
// own G x cw numerators, the G row sums and the objective -- exactly as nmft_reduce_kernel does (lane-strided partial sums,
// fixed-order butterfly: the same bits), runs the stop test (every workgroup for itself, from words nobody writes in this
// launch) and updates its columns of gamma.  Control words: the iteration index and the previous objective live in two
// parity slots; launch number n reads slot n & 1 and workgroup 0 writes slot 1 - (n & 1) for the next launch, so the launch
// has no reader of a word it writes.  The parity is a kernel argument (a captured batch of 64 iterations replays with the
// same arguments per node).
//   ctl: [0] div  [2] done  [3] updates run  [4 + p] previous div  [6 + p] iteration index  [8 + p] done, as seen by launches of parity p
// ---------------------------------------------------------------------------
#define NMFT_RG_WGS 8
struct NmftRgParams {
    const double *partial; int nblk;
    double *stat; int S, G, max_iter; double min_change; int fix_gamma, adjust, parity;
    double *gam, *gam_raw, *ctl, *div_trace;
};
__device__ __forceinline__ void nmft_rg_body(const NmftRgParams &q)
{
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    double *red = reinterpret_cast<double *>(smem_g);              // [G cw] numerators of the own columns, [G] row sums, [1] objective
    const double *__restrict__ partial = q.partial;
    double *__restrict__ ctl = q.ctl;
    // stopped in an earlier launch?  Read from the stop word of THIS launch's parity (written by the launch before it), never from
    // ctl[2], which workgroup 0 of this very launch may set while other workgroups are still starting: every workgroup of a launch
    // takes the same decision and q.stat is complete after the last update.  A stopped launch hands the flag on to the other parity.
    if (ctl[8 + (q.parity & 1)] != 0.0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) ctl[8 + (1 - (q.parity & 1))] = 1.0;
        return;
    }
    const int S = q.S, G = q.G, nblk = q.nblk, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nwg = (int)gridDim.x, cw = (S + nwg - 1) / nwg, s_lo = (int)blockIdx.x * cw;
    const int ncol = s_lo < S ? (S - s_lo < cw ? S - s_lo : cw) : 0;
    const int nown = G * ncol, nneed = nown + G + 1;
    for (int i = (q.fix_gamma ? nown + G : 0) + wv; i < nneed; i += 16) {            // (gamma fixed: the objective alone)
        // statistic number `out` of the update kernels' partial table: numerator (g, s) = g S + s, row sum g = G S + g,
        // objective = G S + G
        int out;
        if (i < nown) { const int g = i / ncol, s = s_lo + i % ncol; out = g * S + s; }
        else out = G * S + (i - nown);
        const double a = nmft_sum_partials(partial + (size_t)out * nblk, nblk, lane);
        if (lane == 0) {
            red[i] = a;
            if (i < nown || blockIdx.x == 0) q.stat[out] = a;
        }
    }
    __syncthreads();
    const int p = q.parity & 1;
    const int it = (int)ctl[6 + p];
    const double div = red[nown + G];
    const double prev = (it == 0) ? 0.0 : ctl[4 + p];
    const bool go = (it < q.max_iter) && (fabs(prev - div) > q.min_change);      // Init_NMFT.py:106
    if (blockIdx.x == 0 && tid == 0) {
        ctl[0] = div;
        ctl[4 + (1 - p)] = div;
        ctl[6 + (1 - p)] = (double)(it + 1);
        ctl[3] = (double)it;
        if (!go) { ctl[2] = 1.0; ctl[8 + (1 - p)] = 1.0; }       // [2]: read by the update kernels of later launches and by the host
        else if (q.fix_gamma) ctl[10] = 1.0 - ctl[10];           // gamma fixed: the fused pass's candidate rows become the current ones
        if (q.div_trace) q.div_trace[it] = div;
    }
    if (!go || q.fix_gamma) return;                                               // uniform over the workgroup
    double *val = red + nneed;                                                    // [G][SC] one chunk of the own columns
    const int SC = 1024 / G;                                                      // columns per chunk (all G values of a column at once)
    for (int c0 = 0; c0 < ncol; c0 += SC) {
        const int g = tid / SC, cc = tid % SC, c = c0 + cc, s = s_lo + c;
        const bool live = g < G && c < ncol;
        double v = 1.0;                                                           // :168
        if (live && G > 1) v = q.gam[(size_t)g * S + s] * (nzd(red[g * ncol + c]) / nzd(red[nown + g]));   // :163
        if (g < G) val[g * SC + cc] = v;
        __syncthreads();
        if (live) {
            if (G > 1) {
                double tot = 0.0;
                for (int k = 0; k < G; ++k) tot += val[k * SC + cc];              // :165
                v = v / tot;                                                      // :166
            }
            q.gam_raw[(size_t)g * S + s] = v;                // the tau update of this iteration sees it unclamped
            if (q.adjust && v < DSM_EPS) v = DSM_EPS;        // _adjustment follows the whole div_update (:88-91,:108)
            q.gam[(size_t)g * S + s] = v;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void nmft_rg_kernel(NmftRgParams q) { nmft_rg_body(q); }
__global__ __launch_bounds__(1024) void nmft_rg_kernel_b(BatchArgs<NmftRgParams> b) { nmft_rg_body(b.p[blockIdx.y]); }

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

// ---------------------------------------------------------------------------
// nmft_wave_kernel: one update in ONE pass over F, without workgroup barriers in the loop.
// A wavefront owns a variant (its 4 rows of F and tau); lanes = samples (NSL per lane).
//   1  R = tau.gamma_raw per lane (tau rows read as LDS broadcasts), Q' = F (/) R
//   2  num[r][g] = sum_s Q'[r][s] gamma_raw[g][s]: 4 x 8 per-lane products per g-chunk, summed over
//      the wavefront by a transposing butterfly that leaves ONE (r,g) total per lane
//   3  that lane updates tau[r][g] (:171-172); the four bases of a (v,g) sit in one quad, so the
//      renormalisation (:174-181) is two DPP exchanges; _adjustment (:88-91); tau -> HBM and LDS
//   4  R2 = tau_new.gamma, objective terms (:152-156), Q = F (/) R2
//   5  gamma numerators of the NEXT iteration accumulate in per-lane registers: num[g] += tau_new[r][g] Q[r]
// do_update = 0 runs 4-5 only (statistics of the current state: the first iteration, div_objective).
// gamma_raw / gamma: the running update's normalised gamma before / after _adjustment.
// Partials are written transposed, [G*S numerators][G H1][1 objective] x workgroups.
// ---------------------------------------------------------------------------
template <int NSL, int GMAX>
__global__ __launch_bounds__(256) void nmft_wave_kernel(const double *__restrict__ F, double *__restrict__ tau,
                                                        const double *__restrict__ gam_raw,
                                                        const double *__restrict__ gam, int V, int S, int G,
                                                        int adjust, int do_update, const double *__restrict__ ctl,
                                                        const double *__restrict__ log_tab, double *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    if (ctl[2] != 0.0) return;
    constexpr int SPAD = 64 * NSL;
    constexpr int GCH = GMAX < 8 ? GMAX : 8;             // haplotypes per reduction chunk
    constexpr int NV = 4 * GCH;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nblk = gridDim.x;
    double2 *ltab = reinterpret_cast<double2 *>(smem_w);                       // [256]
    double *gr = reinterpret_cast<double *>(smem_w) + 2 * DSM_LOG_TAB_N;        // [GMAX][SPAD] gamma_raw
    double *gs = gr + GMAX * SPAD;                                              // [GMAX][SPAD] gamma
    double *t1 = gs + GMAX * SPAD;                                              // [GMAX] rowsum(gamma_raw)
    double *told = t1 + GMAX + wv * (8 * GMAX);                                 // per wavefront [4][GMAX]
    double *tnew = told + 4 * GMAX;                                             // per wavefront [4][GMAX]
    double *red = t1 + GMAX + 4 * (8 * GMAX);                                   // [4][GMAX + 2][SPAD]
    if (tid < DSM_LOG_TAB_N) ltab[tid] = reinterpret_cast<const double2 *>(log_tab)[tid];
    for (int i = tid; i < GMAX * SPAD; i += 256) {
        const int g = i / SPAD, s = i % SPAD;
        const bool in = g < G && s < S;
        gr[i] = in ? gam_raw[(size_t)g * S + s] : 0.0;
        gs[i] = in ? gam[(size_t)g * S + s] : 0.0;
    }
    __syncthreads();
    if (tid < GMAX) {
        double a = 0.0;
        for (int s = 0; s < S; ++s) a += gr[tid * SPAD + s];     // gamma.sum(1)  (:170)
        t1[tid] = a;
    }
    __syncthreads();

    double num[NSL][GMAX];
#pragma unroll
    for (int j = 0; j < NSL; ++j)
#pragma unroll
        for (int g = 0; g < GMAX; ++g) num[j][g] = 0.0;
    double obj = 0.0, h1 = 0.0;
    const int myidx = transpose_index<NV>(lane);         // (r, gg) this lane owns after the reduction
    const int my_r = myidx / GCH, my_gg = myidx % GCH;

    // software prefetch: the F rows and the tau rows of the wavefront's NEXT variant are requested
    // while the current one is processed (a wavefront handles only a handful of variants, so the
    // load latency would otherwise be exposed once per variant)
    double f_nx[NSL][4];
    double t_nx[(4 * GMAX + 63) / 64];
    auto prefetch = [&](int v) {
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            const int s = lane + 64 * j;
#pragma unroll
            for (int r = 0; r < 4; ++r) f_nx[j][r] = (v < V && s < S) ? F[((size_t)v * 4 + r) * S + s] : 1.0;
        }
#pragma unroll
        for (int k = 0; k < (4 * GMAX + 63) / 64; ++k) {
            const int i = lane + 64 * k, r = i / GMAX, g = i % GMAX;
            t_nx[k] = (v < V && i < 4 * GMAX && g < G) ? tau[((size_t)v * 4 + r) * G + g] : 0.0;
        }
    };
    prefetch(blockIdx.x * 4 + wv);
    for (int v = blockIdx.x * 4 + wv; v < V; v += nblk * 4) {
        // tau rows of the variant -> this wavefront's LDS scratch (uniform reads afterwards)
#pragma unroll
        for (int k = 0; k < (4 * GMAX + 63) / 64; ++k) {
            const int i = lane + 64 * k;
            if (i < 4 * GMAX) { told[i] = t_nx[k]; if (!do_update) tnew[i] = t_nx[k]; }
        }
        double f[NSL][4];
        bool live[NSL];
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            live[j] = lane + 64 * j < S;
#pragma unroll
            for (int r = 0; r < 4; ++r) f[j][r] = f_nx[j][r];
        }
        prefetch(v + nblk * 4);
        if (do_update) {
            double q[NSL][4];
#pragma unroll
            for (int j = 0; j < NSL; ++j) {
                double R[4] = {0.0, 0.0, 0.0, 0.0};
                for (int g = 0; g < G; ++g) {
                    const double gm = gr[g * SPAD + lane + 64 * j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) R[r] = fma(told[r * GMAX + g], gm, R[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) q[j][r] = live[j] ? nzd(f[j][r]) / nzd(R[r]) : 0.0;
            }
            for (int g0 = 0; g0 < G; g0 += GCH) {
                double val[NV];
#pragma unroll
                for (int gg = 0; gg < GCH; ++gg) {
                    double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int j = 0; j < NSL; ++j) {
                        const double gm = gr[(g0 + gg) * SPAD + lane + 64 * j];   // rows >= G are zero
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[r] = fma(q[j][r], gm, a[r]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) val[r * GCH + gg] = a[r];
                }
                const double tot_rg = wave_transpose_reduce<NV>(val);             // num[my_r][g0 + my_gg]
                const int g = g0 + my_gg;
                const bool ok = g < G;
                double tn = 0.0;
                if (ok) tn = told[my_r * GMAX + g] * (nzd(tot_rg) / nzd(t1[g]));  // :171-172
                // the four bases of (v,g) live in one quad: sum over a in the reference's order
                const double t_a0 = dpp_mov<0x00>(tn), t_a1 = dpp_mov<0xAA>(tn);  // lanes with (b0,b1) = (0,0) / (0,1) -> r = 0 / 1
                const double t_a2 = dpp_mov<0x55>(tn), t_a3 = dpp_mov<0xFF>(tn);  //                    (1,0) / (1,1) -> r = 2 / 3
                const double tot = ((t_a0 + t_a1) + t_a2) + t_a3;                 // :176-178
                if (ok) {
                    double x = tn / tot;                                          // :180-181
                    if (adjust && x < DSM_EPS) x = DSM_EPS;                       // :88-91
                    if (lane < 32) tau[((size_t)v * 4 + my_r) * G + g] = x;
                    tnew[my_r * GMAX + g] = x;
                }
            }
        }
        // statistics of the (new) state for the next iteration
        h1 += (lane < 4 * GMAX) ? tnew[lane] : 0.0;          // GMAX <= 16: one (r,g) column per lane
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            double R[4] = {0.0, 0.0, 0.0, 0.0};
            for (int g = 0; g < G; ++g) {
                const double gm = gs[g * SPAD + lane + 64 * j];
#pragma unroll
                for (int r = 0; r < 4; ++r) R[r] = fma(tnew[r * GMAX + g], gm, R[r]);
            }
            double q2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double pa = R[r] < DSM_EPS ? DSM_EPS : R[r];
                const double ratio = nzd(f[j][r]) / pa;
                q2[r] = live[j] ? ((R[r] < DSM_EPS) ? nzd(f[j][r]) / nzd(R[r]) : ratio) : 0.0;
                if (live[j]) obj += f[j][r] * dsm_log(ratio, ltab) - f[j][r] + pa;
            }
#pragma unroll
            for (int g = 0; g < GMAX; ++g) {
                if (g < G) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) num[j][g] = fma(tnew[r * GMAX + g], q2[r], num[j][g]);
                }
            }
        }
    }
    // workgroup reduction over the 4 wavefronts (fixed order) -> transposed partials
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NSL; ++j)
#pragma unroll
        for (int g = 0; g < GMAX; ++g) red[((size_t)wv * (GMAX + 2) + g) * SPAD + lane + 64 * j] = num[j][g];
    red[((size_t)wv * (GMAX + 2) + GMAX) * SPAD + lane] = obj;
    // H1: lane i holds the (r = i / GMAX, g = i % GMAX) column sums
    red[((size_t)wv * (GMAX + 2) + GMAX + 1) * SPAD + lane] = (lane < 4 * GMAX) ? h1 : 0.0;
    __syncthreads();
    for (int i = tid; i < G * S; i += 256) {
        const int g = i / S, s = i % S;
        double a = 0.0;
        for (int k = 0; k < 4; ++k) a += red[((size_t)k * (GMAX + 2) + g) * SPAD + s];
        partial[(size_t)i * nblk + blockIdx.x] = a;
    }
    if (tid < G) {
        double a = 0.0;
        for (int k = 0; k < 4; ++k)
            for (int r = 0; r < 4; ++r) a += red[((size_t)k * (GMAX + 2) + GMAX + 1) * SPAD + r * GMAX + tid];
        partial[((size_t)G * S + tid) * nblk + blockIdx.x] = a;
    }
    if (tid == 64) {
        double a = 0.0;
        for (int k = 0; k < 4; ++k)
            for (int l = 0; l < 64; ++l) a += red[((size_t)k * (GMAX + 2) + GMAX) * SPAD + l];
        partial[((size_t)G * S + G) * nblk + blockIdx.x] = a;
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
    size_t g = (N + 63) / 64;                 // at least a few rows per row group
    if (g > 512) g = 512;                     // ~2 workgroups per CU
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
#define LAUNCH_A(GM, TH)                                                                                          \
    hipLaunchKernelGGL((nmft_pass_a_kernel<GM, TH>), dim3(c->nmft_blocks), dim3(TH),                              \
                       ((size_t)(TH / SPAD) * (GM + 2) * SPAD + 2 * DSM_LOG_TAB_N) * sizeof(double), c->stream,   \
                       c->F, c->ntau, c->ngam, c->V, S, G, SPAD, NMFT_CTL(c), c->log_tab, c->npart)
    if (G <= 4) LAUNCH_A(4, 1024);
    else if (G <= 8) LAUNCH_A(8, 512);
    else if (G <= 16) LAUNCH_A(16, 512);
    else LAUNCH_A(32, 256);
#undef LAUNCH_A
    HIP_TRY(hipGetLastError());
    c->npart_cols = c->nmft_blocks;
    return DSM_OK;
}

int k_nmft_gamma(dsm_ctx *c, int max_iter, double min_change, int fix_gamma, int adjust, int parity)
{
    KTimer tm(c, DSM_K_NMFT_G);
    const int nout = c->nG * c->S + c->nG + 1;
    // Few workgroup partials (V <= ~2000 positions, where a chain's ~4 500 updates are most of its run): reduce + gamma /
    // control as ONE launch of up to 32 workgroups (nmft_rg_kernel) -- V = 1000, S = 32, G = 4: 12.7 instead of 16.8 us per
    // update.  More partials: the reduction as its own launch over all its outputs, then the one-workgroup gamma / control
    // launch (config 3: 34 us per update; fused, with one workgroup per sample column, 35).
    // dsm_ctx_set_nmft_fused overrides the size rule per context (tests assert that the two forms agree bit for bit).
    const bool fuse = (c->nmft_fused < 0 || c->nmft_fused == 3) ? c->npart_cols <= 128 : c->nmft_fused != 0;     // (3: the update kernel's own step where it applies -- not here)
    if (fix_gamma && !fuse) {
        // gamma fixed: the objective's reduction and the control step as one launch of one wavefront (nmft_objctl_kernel)
        const NmftObjCtlParams q{c->npart, c->npart_cols, nout, max_iter, min_change, c->nstat, NMFT_CTL(c), c->ndiv_trace};
        LAUNCH_OR_COLLECT(NmftObjCtlParams, q,
                          hipLaunchKernelGGL(nmft_objctl_kernel, dim3(1), dim3(64), 0, c->stream, q),
                          hipLaunchKernelGGL(nmft_objctl_kernel_b, dim3(1, K), dim3(64), 0, c->stream, acc));
        HIP_TRY(hipGetLastError());
        return DSM_OK;
    }
    if (fuse) {
        const int nwg = fix_gamma ? 1 : std::min(c->S, 32);       // (gamma fixed: every workgroup would sum the same objective)
        const int cw = (c->S + nwg - 1) / nwg;
        const size_t sh = ((size_t)c->nG * cw + c->nG + 1 + 1024) * sizeof(double);
        const NmftRgParams q{c->npart, c->npart_cols, c->nstat, c->S, c->nG, max_iter, min_change, fix_gamma, adjust, parity,
                             c->ngam, c->ngam_raw, NMFT_CTL(c), c->ndiv_trace};
        LAUNCH_OR_COLLECT(NmftRgParams, q,
                          hipLaunchKernelGGL(nmft_rg_kernel, dim3(nwg), dim3(1024), sh, c->stream, q),
                          hipLaunchKernelGGL(nmft_rg_kernel_b, dim3(nwg, K), dim3(1024), sh, c->stream, acc));
        HIP_TRY(hipGetLastError());
        return DSM_OK;
    }
    const NmftReduceParams r{c->npart, c->npart_cols, nout, NMFT_CTL(c), c->nstat};
    const NmftGammaParams g{c->nstat, c->S, c->nG, max_iter, min_change, fix_gamma, adjust, c->ngam, c->ngam_raw, NMFT_CTL(c), c->ndiv_trace};
    if (g_batch.K) {                                         // both launches of K chains at once (dsm_host.h: BatchCtl)
        static thread_local BatchArgs<NmftReduceParams> ar;
        static thread_local BatchArgs<NmftGammaParams> ag;
        ar.p[g_batch.k] = r; ag.p[g_batch.k] = g;
        if (g_batch.k == g_batch.K - 1) {
            hipLaunchKernelGGL(nmft_reduce_kernel_b, dim3((nout + 3) / 4, g_batch.K), dim3(256), 0, c->stream, ar);
            hipLaunchKernelGGL(nmft_gamma_kernel_b, dim3(1, g_batch.K), dim3(1024), (size_t)1024 * sizeof(double), c->stream, ag);
        }
        HIP_TRY(hipGetLastError());
        return DSM_OK;
    }
    hipLaunchKernelGGL(nmft_reduce_kernel, dim3((nout + 3) / 4), dim3(256), 0, c->stream, r);
    hipLaunchKernelGGL(nmft_gamma_kernel, dim3(1), dim3(1024), (size_t)1024 * sizeof(double), c->stream, g);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

// the reduction alone: the update kernel that follows begins with the gamma / control step itself (NmftMfmaParams.gstep)
int k_nmft_reduce(dsm_ctx *c)
{
    KTimer tm(c, DSM_K_NMFT_G);
    const int nout = c->nG * c->S + c->nG + 1;
    const NmftReduceParams r{c->npart, c->npart_cols, nout, NMFT_CTL(c), c->nstat};
    hipLaunchKernelGGL(nmft_reduce_kernel, dim3((nout + 3) / 4), dim3(256), 0, c->stream, r);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}
// does the update of this context take that form?  The matrix-core kernel (S <= 128, G <= 16), gamma updating, more workgroup partials
// than the column-workgroup form of the step is used for (k_nmft_gamma); dsm_ctx_set_nmft_fused: 3 = wherever the kernel allows, 0 / 1 = never
bool nmft_gstep_applies(const dsm_ctx *c, int fix_gamma)
{
    if (fix_gamma || !nmft_use_mfma(c) || !c->ngam2 || !c->ngam_raw2) return false;
    static const bool off = DSM_AB_ENV("DESMAN_HIP_NMFT_NO_GSTEP") != nullptr;     // A/B switch
    if (off) return false;
    return c->nmft_fused < 0 ? nmft_mfma_grid(c, false) > 128 : c->nmft_fused == 3;
}

int k_nmft_pass_b(dsm_ctx *c, int adjust)
{
    KTimer tm(c, DSM_K_NMFT_B);
    const int G = c->nG, S = c->S, SP = S + 1;
    int VT = 8192 / (4 * SP);
    if (VT > 8) VT = 8;
    if (VT < 1) VT = 1;
    const size_t sh = ((size_t)G * SP + G + 2 * (size_t)4 * VT * G + (size_t)4 * VT * SP) * sizeof(double);
    if (sh > 160 * 1024) { dsm_set_error("NMFT tile (%zu B) exceeds LDS", sh); return DSM_ERR_UNSUPPORTED; }
    int grid = (c->V + VT - 1) / VT;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(nmft_pass_b_kernel, dim3(grid), dim3(256), sh, c->stream, c->F, c->ntau, c->ngam_raw, c->V, S, G, VT,
                       adjust, NMFT_CTL(c));
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

// which (NSL, GMAX) the one-pass wavefront kernel is compiled for (per-lane accumulators NSL*GMAX <= 32)
static bool wave_shape(const dsm_ctx *c, int *nsl, int *gmax)
{
    const int need = (c->S + 63) / 64;
    *nsl = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 8;
    *gmax = c->nG <= 4 ? 4 : c->nG <= 8 ? 8 : c->nG <= 16 ? 16 : 32;
    return need <= 8 && (*nsl) * (*gmax) <= 32 && (*gmax) <= 16;
}

bool nmft_use_mfma(const dsm_ctx *c);
bool nmft_use_wide(const dsm_ctx *c);
bool nmft_use_wave(const dsm_ctx *c) { int a, b; return wave_shape(c, &a, &b) || nmft_use_mfma(c) || nmft_use_wide(c); }      // the one-pass kernels

int nmft_wave_grid(const dsm_ctx *c)
{
    int g = (c->V + 3) / 4;
    if (g > 768) g = 768;                     // ~3 wavefronts per SIMD (the kernel's register occupancy)
    return g < 1 ? 1 : g;
}

template <int NSL, int GMAX>
static void launch_wave(dsm_ctx *c, int adjust, int do_update, int grid)
{
    const size_t sh = (2 * DSM_LOG_TAB_N + 2 * (size_t)GMAX * 64 * NSL + GMAX + 4 * 8 * GMAX +
                       4 * (size_t)(GMAX + 2) * 64 * NSL) * sizeof(double);
    hipLaunchKernelGGL((nmft_wave_kernel<NSL, GMAX>), dim3(grid), dim3(256), sh, c->stream, c->F, c->ntau, c->ngam_raw,
                       c->ngam, c->V, c->S, c->nG, adjust, do_update, NMFT_CTL(c), c->log_tab, c->npart);
}

// do_update = 1: tau half of the running update + statistics of the next; 0: statistics only
int k_nmft_mfma(dsm_ctx *c, int adjust, int do_update);
bool nmft_use_mfma(const dsm_ctx *c);

int k_nmft_wide(dsm_ctx *c, int adjust, int do_update);
int k_nmft_wave(dsm_ctx *c, int adjust, int do_update)
{
    if (nmft_use_mfma(c)) return k_nmft_mfma(c, adjust, do_update);
    if (nmft_use_wide(c)) return k_nmft_wide(c, adjust, do_update);
    KTimer tm(c, do_update ? DSM_K_NMFT_B : DSM_K_NMFT_A);
    int nsl, gmax;
    if (!wave_shape(c, &nsl, &gmax)) { dsm_set_error("nmft_wave: unsupported shape"); return DSM_ERR_UNSUPPORTED; }
    const int grid = nmft_wave_grid(c);
#define WCASE(N, GM) if (nsl == N && gmax == GM) launch_wave<N, GM>(c, adjust, do_update, grid)
    WCASE(1, 4); WCASE(1, 8); WCASE(1, 16); WCASE(2, 4); WCASE(2, 8); WCASE(2, 16); WCASE(4, 4); WCASE(4, 8);
    WCASE(8, 4);
#undef WCASE
    HIP_TRY(hipGetLastError());
    c->npart_cols = grid;
    return DSM_OK;
}

int k_nmft_get_tau(dsm_ctx *c, uint64_t *d_packed)
{
    hipLaunchKernelGGL(nmft_get_tau_kernel, dim3((c->V + 255) / 256), dim3(256), 0, c->stream, c->ntau, c->V, c->nG,
                       d_packed);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

// ===========================================================================
// nmft_mfma_kernel: the one-pass update of nmft_wave_kernel with its three dense contractions on the matrix cores
// (v_mfma_f64_16x16x4_f64; measured 2x the VALU form of the same contraction on gfx950: the VALU form is bound by
// operand delivery -- one LDS broadcast per FMA -- not by the FMA rate; profiles/r02_mfma_f64_ab.txt).
//
// A wavefront owns a QUAD of variants = 16 rows of F / tau, row index i = 4 r + vv (base r, variant vv of the quad).
// Register layout of everything sample-shaped ("L2"): lane = n + 16 vv (n = sample inside a 16-sample tile),
// element e of a double4 = base r, one double4 per tile t -- which is exactly how the instruction returns
// D[i][j] (lane = j + 16 (i % 4), element i / 4: scripts/ubench/mfma_layout_probe.hip) when rows are ordered i = 4 r + vv:
//   R' = tau_old . gamma_raw   A[i][k] = tau_old[i][4 kb + k] (lane i + 16 k, from this wavefront's LDS copy of the
//                              16 rows), B[k][j] = gamma_raw[4 kb + k][16 t + j] (lane j + 16 k, staged per lane in LDS)
//   Q' = F (/) R'              element-wise in L2 (F is loaded in L2: 16 consecutive doubles per (variant, base, tile))
//   num[i][g] = sum_s Q'[i][s] gamma_raw[g][s]: contraction over SAMPLES = over the 16 lanes of a row group and the
//                              tiles: per-lane products (gamma_raw[g][16 t + n] from LDS), then ONE transposing butterfly
//                              per 4 haplotypes over the 16 lanes of each variant -- DPP only (quad_perm, row_ror:8,
//                              row_shl/shr:4), the four variants of the quad reduce simultaneously in the four DPP rows
//                              (the lane = sample layout paid a 64-lane butterfly per variant: 4x the exchange steps)
//   tau update                 lane (n, vv) ends with base e = 2 b0 + b1, haplotype 4 c + 2 b3 + b2 (b_k = bit k of n): the
//                              four bases of a (variant, haplotype) sit in one quad -> renormalisation by quad broadcasts
//   R2 = tau_new . gamma, Q2 = F (/) R2, objective: as above with the new rows
//   gamma numerators           num_g[g][s] += sum_rows tau_new[row][g] Q2[row][s]: contraction over ROWS, K-block = base e:
//                              A[m][k] = tau_new[vv = k][e][g = m] (lane m + 16 k), B[k][j] = Q2[vv = k][e][16 t + j] = the L2
//                              register itself.  The accumulators D[g][s] stay in registers for the whole kernel.
// No operand ever needs a transposition through LDS.  Shapes: S <= 128 (NT <= 8 tiles), G <= 16 (KB <= 4 K-blocks).  Up to three
// tiles the F tiles stay in registers between the halves (KEEPF); at four they are re-read from L2 (keeping them costs 96 B/lane
// of scratch at 3 wavefronts per SIMD: 38 -> 35 us per update at V = 10k); five and six tiles run at 2 wavefronts per SIMD, where
// 256 registers hold the F tiles again (236-254 VGPRs, no spills: 202 -> 184 us per update at 50k x 96 x 12, and 119 us once the
// end-of-kernel reduction shared the loop's LDS -- see `red` below); seven and eight tiles re-read F (224-249 VGPRs).  Other
// shapes run nmft_wave_kernel / the two-pass kernels.
// ===========================================================================
typedef double double4_t __attribute__((ext_vector_type(4)));
#ifdef DSM_AB_SWITCHES           // phase clocks of the update kernel's quad loop (experiment build only; DESMAN_HIP_NMFT_STAMPS=1 prints them per launch)
__device__ unsigned long long nm_dbg[12];
#define NM_CLK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long _t = __builtin_amdgcn_s_memtime(); nm_dbg[k] += _t - nm_t; nm_t = _t; } } while (0)
#define NM_CLK0() unsigned long long nm_t = __builtin_amdgcn_s_memtime()
#define NM_CLKQ() do { if (blockIdx.x == 0 && threadIdx.x == 0) nm_dbg[7] += 1; } while (0)
#else
#define NM_CLK(k) do { } while (0)
#define NM_CLK0() do { } while (0)
#define NM_CLKQ() do { } while (0)
#endif
#define NM_MFMA(a, b, c, x, y, z) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z)
// Round 6: v_mfma_f64_4x4x4_4b_f64 -- four independent 4 x 4 x 4 products, one D element per lane.  Operand lanes as measured on this
// device (scripts/ubench/mfma_4x4x4_probe.hip, profiles/r06_mfma_4x4x4_layout.txt): A[b][i][k] at lane i + 4 b + 16 k, B[b][k][j] at lane
// j + 4 b + 16 k, D[b][i][j] at lane j + 4 b + 16 i -- with the registers of the 16 x 16 x 4 form it is that product's four diagonal 4 x 4
// blocks, at 16.5 cycles of the SIMD's fp64 datapath against 71 (scripts/ubench/mfma_f64_rate.hip): the datapath does ~30 flops a cycle
// whatever the instruction, so a contraction whose free index is the HAPLOTYPE (N = G <= 8 of the 16 x 16 x 4 form's 16 columns) is run
// as ceil(G / 4) of these per step instead of one padded wide one.
#define NM_MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)

// a / b for the operands of the update (positive, far from the ends of the exponent range): hardware reciprocal, one
// Newton step, one residual correction -- 6 instructions / ~40 issue cycles instead of the 11 / ~80 of the IEEE expansion
// (no v_div_scale / v_div_fmas / v_div_fixup).  The quotient is within 1 ulp of a / b (faithful, not always correctly
// rounded); the factors stay within the 1e-7 of the reference goldens after 100 updates that the tests ask for.
__device__ __forceinline__ double fdiv(double a, double b)
{

    // v_rcp_f64 is good to ~2^-23; one Newton step makes 2^-46, and the residual correction of the QUOTIENT below is itself a
    // Newton step on it (its error is the product of r's and q's: 2^-92) -- a second step on r (rounds 2-3) bought nothing
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
// ... and where the operands ARE at the ends of the exponent range (round 5, found by scripts/dbg/fuzz_nmft.py: a tau row of subnormal start
// values -- 8e-4 of a Dirichlet(0.01) draw's components are below 1e-308 -- against gamma columns that are as small: R = 5.6e-310, F / R =
// 8.9e307 in the reference, but v_rcp_f64 of a subnormal is inf and the Newton step makes NaN of it): the operands are brought to within
// 2^+-512 of one by exact powers of two, the powers put back on the quotient; where the scales are one (always, but for such operands)
// the bits of fdiv.  fdiv_lo: a in (0, 1] (an element of F) over any b > 0; fdiv_ext: any a, b >= 0.
__device__ __forceinline__ double nm_pow2_sel(bool c, int hi_c, int hi_else) { return __hiloint2double(c ? hi_c : hi_else, 0); }    // one select on the high word
__device__ __forceinline__ double fdiv_lo(double a, double b)
{
    const double s = nm_pow2_sel(b < 0x1p-500, 0x5FF00000, 0x3FF00000);        // 2^512 : 1
    return fdiv(a, b * s) * s;
}
// a tile's four quotients F (/) R of the tau half: the scaled form only for a wavefront that holds such a divisor (one test per tile)
__device__ __forceinline__ double4_t nm_div_tile(const double4_t ft, const double4_t R)
{
    double4_t Rz, qv;
#pragma unroll
    for (int e = 0; e < 4; ++e) Rz[e] = nzd(R[e]);
    double m01, m23, m;
    asm("v_min_f64 %0, %1, %2" : "=v"(m01) : "v"(Rz[0]), "v"(Rz[1]));
    asm("v_min_f64 %0, %1, %2" : "=v"(m23) : "v"(Rz[2]), "v"(Rz[3]));
    asm("v_min_f64 %0, %1, %2" : "=v"(m) : "v"(m01), "v"(m23));
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(m < 0x1p-500) != 0ull, 0)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) qv[e] = fdiv_lo(ft[e], Rz[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) qv[e] = fdiv(ft[e], Rz[e]);
    }
    return qv;
}
__device__ __forceinline__ double fdiv_ext(double a, double b)
{
    // Round 6: mantissa over mantissa (both in [1/2, 1): v_frexp_mant_f64 is exact and takes subnormals), the exponents' difference put on the
    // quotient by v_ldexp_f64 -- which overflows to inf and rounds into the subnormal range as the division itself does.  The bits of fdiv(a, b)
    // wherever that neither overflows nor underflows on the way (powers of two scale v_rcp_f64 and every fma of fdiv exactly).  Round 5 scaled
    // by 2^+-512 where an operand was beyond 2^+-500 and put the two scales back one after the other: two huge or two tiny operands with an
    // ordinary quotient passed through the subnormal range or infinity on the way, and a quotient that does overflow (a >= 1 over a
    // subnormal b) came out NaN where the reference's division gives inf (inf - inf inside the Newton step; tests/test_gpu_fuzz.py:
    // test_fuzz_nmft_divisions_at_both_ends_of_the_exponent_range).
    const double q = fdiv(__builtin_amdgcn_frexp_mant(a), __builtin_amdgcn_frexp_mant(b));
    return ldexp(q, __builtin_amdgcn_frexp_exp(a) - __builtin_amdgcn_frexp_exp(b));
}
// test hook (dsm_debug_fdiv): out[i] = fdiv_ext(a[i], b[i]) / fdiv_lo / fdiv
__global__ __launch_bounds__(256) void fdiv_test_kernel(int kind, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = kind == 0 ? fdiv_ext(a[i], b[i]) : (kind == 1 ? fdiv_lo(a[i], b[i]) : fdiv(a[i], b[i]));
}
extern "C" int dsm_debug_fdiv(int kind, const double *a, const double *b, double *out, int n)
{
    if (!a || !b || !out || n < 1 || kind < 0 || kind > 2) { dsm_set_error("debug_fdiv: bad arguments"); return DSM_ERR_ARG; }
    double *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)3 * n * sizeof(double)) != hipSuccess) { dsm_set_error("debug_fdiv: hipMalloc failed"); return DSM_ERR_NOMEM; }
    int r = DSM_OK;
    if (hipMemcpy(d, a, (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d + n, b, (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) r = DSM_ERR_HIP;
    if (r == DSM_OK) {
        hipLaunchKernelGGL(fdiv_test_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, kind, d, d + n, d + 2 * (size_t)n, n);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out, d + 2 * (size_t)n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) r = DSM_ERR_HIP;
    }
    (void)hipFree(d);
    if (r != DSM_OK) dsm_set_error("debug_fdiv: HIP error");
    return r;
}
// Q2 = F (/) max(R2, eps) and the objective terms of one 16-sample tile (Init_NMFT.py:152-156, du.elop), element e = base.
// F is a count + 1 over a depth + 4, never zero: elop's zero test can only fire on R.  Lanes without a cell (padded samples,
// variants past the end) carry some F of the table and R = 0: their quotient F / eps is finite, meets a zero row of tau in the
// contraction that follows (variants) or lands in a column nobody reads (samples), and is kept out of the objective by `live`.
//
// Round 5: the tile is STRAIGHT-LINE code.  Rounds 2-4 tested every element for elop's rare cases (0 < R < eps divides by R itself;
// a quotient outside the table logarithm's domain takes libm's) behind wave-uniform branches -- five basic-block ends per element,
// an exposed LDS round trip per table look-up, the exec-mask bookkeeping of `if (live)`: 72 instructions per element, a third of
// them scalar (counted in the assembly of round 4's tile; the before / after of the rewrite is profiles/r05_nmft_q2_straightline_ab.txt).  Now ONE test per tile -- a live lane with !(R >= eps), NaN included -- sends the
// whole tile through the element-by-element code (nm_tile_q2_rare: never with the adjustment on, where tau >= eps and the columns of
// gamma sum to one); the common path is four independent division / logarithm chains the scheduler interleaves, their table
// look-ups in flight together.  The operations on live elements and their order are those of the element-by-element form: same bits.
// log of any double without a call (libm's log as a callee costs the calling kernel its scratch-free register allocation): the table
// logarithm where it applies, else subnormals rescaled by 2^64, log 0 = -inf, log of a negative number or NaN = NaN, log inf = inf
__device__ __forceinline__ double nm_log_any(double x, const double2 *__restrict__ ltab)
{
    if (dsm_log_ok(x)) return dsm_log_core(x, ltab);
    if (x > 0.0 && x < 0x1p-1022) return dsm_log_core(x * 0x1p64, ltab) - 64.0 * 0x1.62e42fefa39efp-1;
    return x == 0.0 ? -__builtin_inf() : (x > 0.0 ? x : __builtin_nan(""));
}
// the element-by-element form of the tile (rounds 2-4), kept as a loop for the rare tile
__device__ __forceinline__ double4_t nm_tile_q2_rare(const double4_t &ft, const double4_t &R, bool live, const double2 *__restrict__ ltab, double &obj)
{
    double4_t q2 = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int e = 0; e < 4; ++e) {
        const double Re = e == 0 ? R[0] : e == 1 ? R[1] : e == 2 ? R[2] : R[3], fe = e == 0 ? ft[0] : e == 1 ? ft[1] : e == 2 ? ft[2] : ft[3];
        const bool tiny = Re < DSM_EPS;
        const double pa = tiny ? DSM_EPS : Re;
        const double ratio = fdiv(fe, pa);                                       // (pa >= eps, F in (0, 1]: a normal number)
        // elop divides by R itself when 0 < R < eps; nzd: the lanes without a cell have R = 0
        const double qq = (tiny && Re != 0.0) ? fdiv_ext(fe, nzd(Re)) : ratio;       // (0 < R < eps may be a subnormal)
        q2[0] = e == 0 ? qq : q2[0]; q2[1] = e == 1 ? qq : q2[1]; q2[2] = e == 2 ? qq : q2[2]; q2[3] = e == 3 ? qq : q2[3];
        const double o2 = obj + (fe * nm_log_any(ratio, ltab) - fe + pa);
        obj = live ? o2 : obj;
    }
    return q2;
}
__device__ __forceinline__ double4_t nm_tile_q2(const double4_t &ft, const double4_t &R, bool live, const double2 *__restrict__ ltab,
                                                double &obj)
{
    const bool ok = (R[0] >= DSM_EPS) & (R[1] >= DSM_EPS) & (R[2] >= DSM_EPS) & (R[3] >= DSM_EPS);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(live && !ok) != 0ull, 0)) {
        return nm_tile_q2_rare(ft, R, live, ltab, obj);
    }
    double4_t q2;
    double term[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        double pa;                                                              // max(R, eps) -- lanes without a cell: R = 0 -- as ONE instruction
        asm("v_max_f64 %0, %1, %2" : "=v"(pa) : "v"(R[e]), "v"(DSM_EPS));       // (the builtin quiets a signalling NaN first: an instruction more)
        const double ratio = fdiv(ft[e], pa);                                   // (pa >= eps, F in (0, 1]: the quotient is a normal number)
        q2[e] = ratio;
        term[e] = ft[e] * dsm_log_core(ratio, ltab) - ft[e] + pa;
    }
    double o2 = obj;
#pragma unroll
    for (int e = 0; e < 4; ++e) o2 += term[e];
    obj = live ? o2 : obj;
    return q2;
}
// ---- tau numerators on the matrix cores (round 4) --------------------------------------------------------------------------
// num[i][g] = sum_s Q'[i][s] gamma_raw[g][s] contracts over SAMPLES, which the L2 layout keeps in the low four lane bits -- where
// the instruction wants a free index.  Rounds 2-3 therefore ran it on the VALU (16 FMAs per tile and K-block on per-lane LDS
// operands, then a 15-add / 30-DPP transposing butterfly per K-block: a quarter of the update kernel's VALU instructions at
// S = 96, G = 12).  Now a tile of Q' (16 rows x 16 samples) crosses the wavefront's own 2.3 KB of LDS once -- written in L2 (four
// conflict-free 512 B row groups), read back as A[i = lane % 16][k = lane / 16] = Q'[i][4 k + j], j = 0..3 (two 16 B reads) -- and
// meets B_j[k][g] = gamma_raw[g][16 t + 4 k + j] in four MFMAs per tile.  D[i][g] returns on lane g + 16 vv, element e (i = 4 e +
// vv): the four bases of a (variant, haplotype) pair end in ONE lane, so the renormalisation over bases needs no exchange at all.
// Columns g >= G of D are never read; their lanes supply any in-range operand.
#define NM_XS 18                      // row stride of the transposition tile in doubles (16 + 2: the 16 B reads of a row group spread over the banks)
#define NM_XQ (16 * NM_XS)            // doubles per wavefront

// gamma operands in LDS: ONE zero-padded matrix [GP][LDG = SPAD + 1] per factor serves both contractions that read it -- the odd row
// stride puts the 4 rows x 16 samples of an R-operand fetch and the 16 rows x 4 samples of a numerator-operand fetch on every bank
// exactly four times (512 B each: the minimum) -- where rounds 2-3 kept a plain copy and two fragment-ordered copies (a third of the
// kernel's LDS at S = 96, G = 12: two workgroups per CU; now three).
__device__ __forceinline__ void nm_stage_gamma_p(double *graw_p, double *ggam_p, const double *__restrict__ graw, const double *__restrict__ ggam,
                                                 int GP, int LDG, int G, int S, int tid, int nthr)
{
    for (int i = tid; i < GP * LDG; i += nthr) {
        const int g = i / LDG, sidx = i - g * LDG;
        const bool in = g < G && sidx < S;
        graw_p[i] = in ? graw[(size_t)g * S + sidx] : 0.0;
        ggam_p[i] = in ? ggam[(size_t)g * S + sidx] : 0.0;
    }
}

// Which contractions run on the four-block instruction: up to NM_B4_MAXKB blocks of four haplotypes.  Measured at 50k x 96 (us per update,
// wide form -> four-block form, profiles/r06_nmft_b4.txt): G = 5 83.2 -> 78.8, G = 8 87.2 -> 82.0; G = 9 92.5 -> 94.5 and G = 12 95.5 -> 97.7 --
// at three blocks the saving on the fp64 datapath (12 x 16.5 against 4 x 71 cycles a tile) is less than what 8 more operand fetches a tile
// and their waits cost a kernel that is bound by instruction issue; at sixteen haplotypes the wide instruction has no padded column.  A
// function of KB alone: every kernel family takes the same sums in the same order at a given shape.
#ifndef NM_B4_MAXKB
#define NM_B4_MAXKB 2
#endif
constexpr bool nm_b4(int KB) { return KB <= NM_B4_MAXKB; }
#ifndef NM_B4G_MAXKB
#define NM_B4G_MAXKB NM_B4_MAXKB
#endif
// the gamma numerators' form, measured apart (its operands can be held in registers, so three blocks cost no extra fetch per tile there): at
// 50k x 96 x 12 the update kernel 81.2 -> 80.7 us, 80.4 -> 79.9 at G = 9 -- the wide instruction's padded quarter is not what the time is
// (profiles/r06_nmft_b4_g3.txt); same bound as the tau numerators'
constexpr bool nm_b4g(int KB) { return KB <= NM_B4G_MAXKB; }

// one tile: qv = Q' of tile t in L2 -> num += Q'_t . gamma_raw_t^T; graw_t = graw_p + 16 t.
// Wide form (KB = 4): four 16 x 16 x 4 instructions, B_j[k = q][g = n] = gamma_raw[g][16 t + 4 k + j]; num[e] of lane (n, q) is base e of
// variant q, haplotype n.
// Round 6, four-block form (NM_MFMA4): the A operands are the wide form's (lane (n, q) holds row n of the transposition tile, samples
// 4 q + j for step j: row block b = n / 4, row in block i = n % 4, k = q); every block meets the SAME four haplotypes,
// B[k = q][j = n % 4] = gamma_raw[g = 4 h + n % 4][16 t + 4 q + step], h = 0 .. KB - 1: KB instructions per step, no padded column up to
// G = 4 KB.  num[h] of lane (n, q) is then the numerator of row 4 (n / 4) + q -- base n / 4 of variant q -- and haplotype 4 h + n % 4
// (components h >= KB stay zero): every lane has work in nm_tau_finish, where the wide form leaves lanes n >= G idle.
// (The B operands are fetched before the tile crosses LDS: they wait for nothing, and behind the wave barriers their latency would be a
// second exposed one per tile.)
template <int KB>
__device__ __forceinline__ double4_t nm_num_tile(double4_t num, const double4_t &qv, double *xq, const double *graw_t, int LDG, int n, int q)
{
    constexpr int GP = 4 * KB;
    double b[nm_b4(KB) ? KB : 1][4];
    if constexpr (nm_b4(KB)) {
#pragma unroll
        for (int h = 0; h < KB; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) b[h][j] = graw_t[(4 * h + (n & 3)) * LDG + 4 * q + j];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) xq[(4 * e + q) * NM_XS + n] = qv[e];
    __builtin_amdgcn_wave_barrier();
    const double2 lo = *reinterpret_cast<const double2 *>(xq + n * NM_XS + 4 * q);
    const double2 hi = *reinterpret_cast<const double2 *>(xq + n * NM_XS + 4 * q + 2);
    __builtin_amdgcn_wave_barrier();
    const double a[4] = {lo.x, lo.y, hi.x, hi.y};
    if constexpr (nm_b4(KB)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < KB; ++h) num[h] = NM_MFMA4(a[j], b[h][j], num[h]);
    } else {
        const double *bl = graw_t + (n < GP ? n : GP - 1) * LDG + 4 * q;          // B_j[k = q][g = n] = gamma_raw[g][16 t + 4 k + j]
#pragma unroll
        for (int j = 0; j < 4; ++j) num = NM_MFMA(a[j], bl[j], num, 0, 0, 0);
    }
    return num;
}

// the tau rows of the quad from their numerators (Init_NMFT.py:171-181, :88-91).  Wide form: lane (g = n, vv = q) holds the four bases.
// Four-block form: lane (n, q) holds, for h < KB, base n / 4 of variant q and haplotype 4 h + n % 4; the four bases of a (variant,
// haplotype) pair are the lanes n % 4 + 4 e of the lane's row of 16.
template <int KB, bool TO_GLOBAL>
__device__ __forceinline__ void nm_tau_finish(const double4_t &num, const double *told, double *tnew, const double *t1, int G, int n, int q,
                                              int adjust, bool store, bool vok, double *tau_v /* tau + (v0 + q) * 4 * G, or null */)
{
    constexpr int GP = 4 * KB;
    if constexpr (nm_b4(KB)) {
        const int e = n >> 2, row = 4 * e + q, l0 = 16 * q + (n & 3);
#pragma unroll
        for (int h = 0; h < KB; ++h) {
            const int g = 4 * h + (n & 3);
            const bool okg = g < G;
            const double tn = okg ? told[row * GP + g] * fdiv_ext(nzd(num[h]), nzd(t1[okg ? g : 0])) : 0.0;                      // :171-172
            const double tot = ((__shfl(tn, l0, 64) + __shfl(tn, l0 + 4, 64)) + __shfl(tn, l0 + 8, 64)) + __shfl(tn, l0 + 12, 64);   // :176-178
            if (okg) {
                double x = fdiv_ext(tn, tot);                                                                                    // :180-181
                if (adjust && x < DSM_EPS) x = DSM_EPS;                                                                        // :88-91
                if (TO_GLOBAL && store) tau_v[(size_t)e * G + g] = x;
                tnew[row * GP + g] = vok ? x : 0.0;
            }
        }
    } else {
        const bool okg = n < G;
        double tn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) tn[e] = okg ? told[(4 * e + q) * GP + n] * fdiv_ext(nzd(num[e]), nzd(t1[okg ? n : 0])) : 0.0;   // :171-172
        const double tot = ((tn[0] + tn[1]) + tn[2]) + tn[3];                                                                  // :176-178
        if (okg) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double x = fdiv_ext(tn[e], tot);                                                                                 // :180-181
                if (adjust && x < DSM_EPS) x = DSM_EPS;                                                                        // :88-91
                if (TO_GLOBAL && store) tau_v[(size_t)e * G + n] = x;
                tnew[(4 * e + q) * GP + n] = vok ? x : 0.0;
            }
        }
    }
}

// The gamma numerators' contraction over the quad's 16 rows, out[g][s] += sum_rows tau_new[row][g] Q2[row][s]; B is a statistics tile as it
// stands (lane (n, q), element e: base e of variant q, sample n; k = q, one step per base).
// Wide form: A[i = n][k = q] = tau_new[4 e + q][g = n]; acc[e'] of lane (n, q) is haplotype 4 e' + q at sample n.
// Four-block form (round 6): sample block n / 4, A[i][k = q] = tau_new[4 e + q][g = 4 h + i] the same in every block: acc[h] of lane (n, q)
// is haplotype 4 h + q at sample n -- the wide form's acc[e'] with e' = h, so the launch's closing reduction is the one it was, on KB
// components.  The 4 KB values of A are held over the tile loop where the registers allow (HOLD: the callers with 256 registers and more),
// else read from the quad's LDS rows tile by tile behind a wave barrier (which keeps the compiler from holding them all the same: 8 KB - 8
// registers the kernels with 128 / 168 do not have).
template <int KB> struct NmAg { double a[nm_b4g(KB) ? 4 * KB : 4]; };
template <int KB>
__device__ __forceinline__ void nm_ag_load(NmAg<KB> &ag, const double *tnew, bool gnum, int n, int q)
{
    constexpr int GP = 4 * KB;
    if constexpr (nm_b4g(KB)) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int h = 0; h < KB; ++h) ag.a[e * KB + h] = gnum ? tnew[(4 * e + q) * GP + 4 * h + (n & 3)] : 0.0;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) ag.a[e] = (gnum && n < GP) ? tnew[(4 * e + q) * GP + n] : 0.0;
    }
}
template <int KB>
__device__ __forceinline__ void nm_gnum_tile(double4_t &acc, const NmAg<KB> &ag, const double4_t &q2)
{
    if constexpr (nm_b4g(KB)) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int h = 0; h < KB; ++h) acc[h] = NM_MFMA4(ag.a[e * KB + h], q2[e], acc[h]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = NM_MFMA(ag.a[e], q2[e], acc, 0, 0, 0);
    }
}
// H1's terms: lane (n = g, q) adds tau_new[vv = q][e][g] over the bases (the wide form's A values themselves)
template <int KB>
__device__ __forceinline__ void nm_h1_add(double &h1, const NmAg<KB> &ag, const double *tnew, bool gnum, int n, int q)
{
    constexpr int GP = 4 * KB;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if constexpr (nm_b4g(KB)) h1 += (gnum && n < GP) ? tnew[(4 * e + q) * GP + n] : 0.0;
        else h1 += ag.a[e];
    }
}
// per-tile fetch of the operands where they are not held
template <int KB, bool HOLD>
__device__ __forceinline__ void nm_ag_tile(NmAg<KB> &ag, const double *tnew, bool gnum, int n, int q)
{
    if constexpr (!HOLD) {
        __builtin_amdgcn_wave_barrier();
        nm_ag_load<KB>(ag, tnew, gnum, n, q);
    }
}

struct NmftMfmaParams {
    const double *F; double *tau; const double *gam_raw, *gam;
    int V, S, G, adjust, do_update;
    const double *ctl, *log_tab; double *partial;
    double *tau2;       // fix_gamma == 2: the second tau buffer of the fused pass (below)
    int fix_gamma;      // factorize_tau (Init_NMFT.py:134-149; the `-r` path, bin/desman:181-206): gamma stays as it is, so only the objective
                        // of the statistics is wanted -- no gamma numerators (24 of a quad's 84 MFMAs at six tiles), no row sums, one partial
                        // per workgroup instead of G S + G + 1.
                        // 2 = the FUSED pass (this kernel; the persistent loop has its own): with gamma unchanged the product tau . gamma that
                        // an update's tau half divides F by is the very product its predecessor's statistics half took the objective of, so one
                        // pass per update does both -- R = tau_k . gamma, objective of tau_k AND Q = F (/) R -> numerators -> candidate
                        // tau_(k+1), written to the OTHER buffer; the control step that follows (stop test on the objective of tau_k) accepts
                        // the candidate by flipping the parity word ctl[10], or stops and leaves tau_k current.  One contraction, 4 NT divisions
                        // and a pass over the quad's LDS rows less per update; the same objective trace, update count and factors bit for bit
                        // (nm_tile_q2's quotient IS the tau half's: both divide by R with elop's zero rule).
    // Round 6, gstep: the launch BEGINS with the gamma / control step of the update (what nmft_gamma_kernel does as a launch of its own
    // between the reduction and this kernel: 4.7 us of a large table's 81): every workgroup takes the stop decision and forms the new gamma
    // for itself from the reduced statistics -- the same operations in the same order: the same bits -- straight into the LDS matrices it
    // would have staged; workgroup 0 also writes the control words and the new gamma to the OTHER of two global buffers (launch n reads the
    // buffers and control slots of parity n & 1 and writes those of parity 1 - (n & 1): no word is read and written in one launch; the
    // protocol of nmft_rg_body).  gam / gam_raw are then the buffers of this launch's parity.
    int gstep, parity, max_iter;
    double min_change;
    const double *stat;
    double *gam_out, *gam_raw_out, *div_trace;
};
template <int NT, int KB, bool KEEPF, bool FIXF = false>      // FIXF: the fused pass of factorize_tau as an instantiation of its own (registers)
__device__ __forceinline__ void nmft_mfma_body(const NmftMfmaParams &prm)
{
    const double *__restrict__ F = prm.F, *__restrict__ gam_raw = prm.gam_raw, *__restrict__ gam = prm.gam;
    double *tau = prm.tau, *__restrict__ partial = prm.partial;
    const double *__restrict__ ctl = prm.ctl, *__restrict__ log_tab = prm.log_tab;
    const int V = prm.V, S = prm.S, G = prm.G, adjust = prm.adjust, do_update = prm.do_update;
    const bool gnum = !FIXF && prm.fix_gamma == 0;                              // the gamma numerators and row sums are wanted
    // (Round 5 ran the two contractions over HAPLOTYPES of tables with up to four of them on the vector ALU at five / six tiles -- 16 v_fma_f64
    // a tile against a wide matrix instruction with 12 empty columns.  Round 6's four-block instruction does that quarter in 4 x 16.5 cycles
    // with neither the 32 + 32 operand registers nor the transposing butterfly: 50k x 96 x 4 69 -> 65 us per update, profiles/r06_nmft_b4.txt.)
    constexpr bool PF = KEEPF && (NT == 5 || NT == 6);                          // the quad loop that looks ahead (below)
    constexpr bool fusedfix = FIXF;                                             // gamma fixed: one pass per update (NmftMfmaParams)
    extern __shared__ __attribute__((aligned(16))) char smem_m[];
    const bool gstep = !FIXF && prm.gstep != 0;
    bool gs_go = true;
    if (gstep) {
        // the control step (nmft_gamma_body / nmft_rg_body), taken by every workgroup alike from words no workgroup of this launch writes
        double *ctlw = const_cast<double *>(ctl);
        const int p = prm.parity & 1;
        if (ctl[8 + p] != 0.0) {                                                // stopped in an earlier launch: hand the flag on
            if (blockIdx.x == 0 && threadIdx.x == 0) ctlw[8 + (1 - p)] = 1.0;
            return;
        }
        const int it = (int)ctl[6 + p];
        const double div = prm.stat[(size_t)G * S + G];
        const double prev = (it == 0) ? 0.0 : ctl[4 + p];
        gs_go = (it < prm.max_iter) && (fabs(prev - div) > prm.min_change);     // Init_NMFT.py:106
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ctlw[0] = div;
            ctlw[4 + (1 - p)] = div;
            ctlw[6 + (1 - p)] = (double)(it + 1);
            ctlw[3] = (double)it;
            if (!gs_go) { ctlw[2] = 1.0; ctlw[8 + (1 - p)] = 1.0; }
            else ctlw[11] = (double)(1 - p);                                    // where the current gamma is from now on
            if (prm.div_trace) prm.div_trace[it] = div;
        }
        if (!gs_go) return;
    } else if (ctl[2] != 0.0) return;
    // fused pass: which buffer holds the current rows (flipped by the control step when it accepts a candidate)
    const bool par1 = fusedfix && ctl[10] != 0.0;
    const double *tau_in = par1 ? prm.tau2 : prm.tau;
    double *tau_out = fusedfix ? (par1 ? prm.tau : prm.tau2) : prm.tau;
    constexpr int GP = 4 * KB, SPAD = 16 * NT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, q = lane >> 4;
    const int nblk = gridDim.x;
    double2 *ltab = reinterpret_cast<double2 *>(smem_m);                        // [256]
    constexpr int LDG = SPAD + 1;
    double *graw_p = reinterpret_cast<double *>(ltab + DSM_LOG_TAB_N);          // [GP][LDG] gamma_raw, zero-padded (nm_stage_gamma_p)
    double *ggam_p = graw_p + (FIXF ? 0 : GP * LDG);                            // [GP][LDG] gamma (the fused pass reads no other: one matrix)
    double *t1 = ggam_p + GP * LDG;                                             // [GP] rowsum(gamma_raw)
    double *told = t1 + GP + wv * (2 * 16 * GP);                                // per wavefront [16][GP], row i = 4 r + vv
    double *tnew = told + 16 * GP;                                              // per wavefront [16][GP]
    double *xq = t1 + GP + 4 * (2 * 16 * GP) + wv * NM_XQ;                      // per wavefront [16][NM_XS] transposition tile of Q'
    double *red = reinterpret_cast<double *>(smem_m);                           // [4][GP + 2][SPAD] end-of-kernel reduction: takes the place
                                                                                // of everything above once the quads are done (at S = 96, G = 12 its
                                                                                // 43 KB on top of the rest made 87 KB = ONE workgroup per CU)
#ifdef DSM_AB_SWITCHES
    const unsigned long long nm_r0 = __builtin_amdgcn_s_memrealtime(), nm_m0 = __builtin_amdgcn_s_memtime();
#endif
    {
        // operands of the whole launch: every load of a thread is issued before its first store (round 5: the staging loop and the row
        // sums' own global reads were 9 400 cycles of a launch's start, one exposed latency after the other); the row sums
        // gamma.sum(1) (:170) are then taken from the staged matrix -- the same numbers in the same order, lane-strided + butterfly
        constexpr int NST = (GP * LDG + 255) / 256;
        double gr[FIXF ? 1 : NST], gg[NST];
        const double2 lt = reinterpret_cast<const double2 *>(log_tab)[tid];
        if (gstep) {
            if constexpr (!FIXF) {
                // the gamma update (Init_NMFT.py:163-168; nmft_gamma_body's operations in its order): v = gamma (.) num / H1, the column's
                // total over the haplotypes in their order, v / total; gamma_raw = that, gamma = that clamped at eps where the adjustment applies
#pragma unroll
                for (int j = 0; j < NST; ++j) {
                    const int i = tid + 256 * j, g = i / LDG, sidx = i - g * LDG;
                    const bool in = i < GP * LDG && g < G && sidx < S;
                    double v = in ? 1.0 : 0.0;                                                                    // :168
                    if (in && G > 1) v = gam[(size_t)g * S + sidx] * (nzd(prm.stat[(size_t)g * S + sidx]) / nzd(prm.stat[(size_t)G * S + g]));   // :163
                    gr[j] = v;
                    if (i < GP * LDG) graw_p[i] = v;
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < NST; ++j) {
                    const int i = tid + 256 * j, g = i / LDG, sidx = i - g * LDG;
                    const bool in = i < GP * LDG && g < G && sidx < S;
                    if (in && G > 1) {
                        double tot = 0.0;
                        for (int k = 0; k < G; ++k) tot += graw_p[k * LDG + sidx];                                  // :165
                        gr[j] = gr[j] / tot;                                                                      // :166
                    }
                    gg[j] = (in && adjust && gr[j] < DSM_EPS) ? DSM_EPS : gr[j];                                  // :88-91, :108
                    if (in && blockIdx.x == 0) { prm.gam_raw_out[(size_t)g * S + sidx] = gr[j]; prm.gam_out[(size_t)g * S + sidx] = gg[j]; }
                }
                __syncthreads();
            }
        } else {
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int i = tid + 256 * j, g = i / LDG, sidx = i - g * LDG;
            const bool in = i < GP * LDG && g < G && sidx < S;
            if constexpr (!FIXF) gr[j] = in ? gam_raw[(size_t)g * S + sidx] : 0.0;
            gg[j] = in ? gam[(size_t)g * S + sidx] : 0.0;
        }
        }
        ltab[tid] = lt;
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int i = tid + 256 * j;
            if (i < GP * LDG) {
                if constexpr (!FIXF) graw_p[i] = gr[j];
                ggam_p[i] = gg[j];
            }
        }
        __syncthreads();
        const double *src = fusedfix ? ggam_p : graw_p;
        for (int g = wv; g < GP; g += 4) {
            double a = 0.0;
            for (int s = lane; s < SPAD; s += 64) a += src[g * LDG + s];
            a = group_allreduce_sum<64>(a);
            if (lane == 0) t1[g] = a;
        }
    }
    __syncthreads();

    for (int k = lane; k < 2 * 16 * GP; k += 64) told[k] = 0.0;                 // incl. tnew and the padded haplotype columns
    double4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double obj = 0.0, h1 = 0.0;

    // ---- a quad's work: the tau half of the running update and the statistics of the next (or, gamma fixed, the fused pass), on F tiles
    // handed in by tile(t); used(t) runs when tile t has been read for the last time, rows_read() when the quad's old tau rows have
    const int nquad = (V + 3) >> 2;
    NM_CLK0();
#ifdef DSM_AB_SWITCHES
    if (blockIdx.x == 0 && threadIdx.x == 0) nm_dbg[5] += nm_t - nm_m0;        // prologue
#endif
    auto quad_work = [&](auto tile, auto livef, auto used, auto rows_read, const int v0, const bool vok) __attribute__((always_inline)) {
        if constexpr (fusedfix) {
            // gamma fixed: objective of the current rows and the candidate rows of the next update from ONE product
            double a_cur[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) a_cur[kb] = told[n * GP + 4 * kb + q];
            double4_t num = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                double4_t R = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) R = NM_MFMA(a_cur[kb], ggam_p[(4 * kb + q) * LDG + 16 * t + n], R, 0, 0, 0);
                const double4_t ft = tile(t);
                const double4_t q2 = nm_tile_q2(ft, R, livef(t), ltab, obj);
                used(t);
                num = nm_num_tile<KB>(num, q2, xq, ggam_p + 16 * t, LDG, n, q);
            }
            nm_tau_finish<KB, true>(num, told, tnew, t1, G, n, q, adjust, vok, vok, tau_out + (size_t)(v0 + q) * 4 * G);
            __builtin_amdgcn_wave_barrier();
            rows_read();
        } else {
        if (do_update) {
            double a_old[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) a_old[kb] = told[n * GP + 4 * kb + q];
            double4_t num = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                double4_t R = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) R = NM_MFMA(a_old[kb], graw_p[(4 * kb + q) * LDG + 16 * t + n], R, 0, 0, 0);
                const double4_t ft = tile(t);
                const double4_t qv = nm_div_tile(ft, R);                                // nm_tile_q2: F > 0; lanes without a cell stay finite
                num = nm_num_tile<KB>(num, qv, xq, graw_p + 16 * t, LDG, n, q);
            }
            nm_tau_finish<KB, true>(num, told, tnew, t1, G, n, q, adjust, vok, vok, tau + (size_t)(v0 + q) * 4 * G);
            __builtin_amdgcn_wave_barrier();
            rows_read();
            NM_CLK(3);
        }
        // statistics of the (new) state: R2, objective, Q2, gamma numerators, H1
        double a_new[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) a_new[kb] = tnew[n * GP + 4 * kb + q];
        constexpr bool HOLDAG = !nm_b4g(KB) || NT >= 5;                          // (five tiles and more: 256 registers)
        NmAg<KB> a_g;                                                           // A of the row contraction (nm_gnum_tile)
        if constexpr (HOLDAG) nm_ag_load<KB>(a_g, tnew, gnum, n, q);
        nm_h1_add<KB>(h1, a_g, tnew, gnum, n, q);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            double4_t R = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) R = NM_MFMA(a_new[kb], ggam_p[(4 * kb + q) * LDG + 16 * t + n], R, 0, 0, 0);
            double4_t q2;
            const double4_t ft = tile(t);
            q2 = nm_tile_q2(ft, R, livef(t), ltab, obj);
            used(t);
            nm_ag_tile<KB, HOLDAG>(a_g, tnew, gnum, n, q);
            if (gnum) nm_gnum_tile<KB>(acc[t], a_g, q2);
        }
        __builtin_amdgcn_wave_barrier();
        NM_CLK(4);
        }       // (not the fused pass)
    };
    if constexpr (!PF) {
    for (int qd = blockIdx.x * 4 + wv; qd < nquad; qd += nblk * 4) {
        NM_CLK(0); NM_CLKQ();
        const int v0 = qd * 4;
        const bool vok = v0 + q < V;                                            // this lane's variant exists
        // the 16 tau rows of the quad (contiguous in HBM: [vv][r][g]) -> LDS [i = 4 r + vv][g]
        for (int k = lane; k < 16 * G; k += 64) {
            const int vv = k / (4 * G), r = (k / G) & 3, g = k % G;
            const double x = (v0 + vv < V) ? tau_in[(size_t)v0 * 4 * G + k] : 0.0;
            told[(4 * r + vv) * GP + g] = x;
            if (!do_update) tnew[(4 * r + vv) * GP + g] = x;
        }
        // F in L2: f[t][e] = F[variant v0 + q][base e][16 t + n]; kept in registers for both halves while NT <= 3,
        // re-read (L2) by the second half at four, seven and eight tiles
        double4_t f[KEEPF ? NT : 1];
        bool live[NT];
        auto load_f = [&](int t) {
            double4_t x;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = live[t] ? F[((size_t)(v0 + q) * 4 + e) * S + 16 * t + n] : 1.0;
            return x;
        };
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            live[t] = vok && (16 * t + n < S);
            if constexpr (KEEPF) f[t] = load_f(t);
        }
        __builtin_amdgcn_wave_barrier();
        NM_CLK(1);
        quad_work([&](int t) { return KEEPF ? f[KEEPF ? t : 0] : load_f(t); }, [&](int t) { return live[t]; }, [](int) {}, []() {}, v0, vok);
    }
    } else {
    // ---- five and six tiles (round 5): the loop that looks ahead.  Measured with the phase clocks of the experiment build
    // (profiles/r05_nmft_phase_clocks.txt, one wavefront per SIMD, 50k x 96 x 12): of a quad's 21 000 cycles 4 300 went by at the TOP of
    // the round -- the old tau rows fetched and staged with their latency exposed, 24 exec-masked loads with 64-bit vector address
    // arithmetic each -- and 1 100 more waiting for the first F tile.  Now: (i) the next quad's tau rows are fetched at the start of the
    // round and staged the moment this quad's have been read for the last time; (ii) tile t of the next quad is loaded into the
    // registers tile t of this one has just left (the statistics half / the fused pass reads each tile once more and last); (iii) an
    // address is a wave-uniform base (the quad's first row) + a 32-bit lane offset, no branch, no select: lanes without a cell read a
    // cell that exists (the last variant, the last sample) -- R = 0 there and any finite F will do (nm_tile_q2).
    const int qstride = nblk * 4;
    int qd = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
    const uint32_t rowb = (uint32_t)S * 8u;
    uint32_t so[NT];                                                            // byte offset of this lane's sample in tile t
#pragma unroll
    for (int t = 0; t < NT; ++t) so[t] = (uint32_t)min(16 * t + n, S - 1) * 8u;
    const bool slive_last = 16 * (NT - 1) + n < S;
    auto f_off = [&](int qd_) -> uint32_t { return (uint32_t)min(q, V - 1 - 4 * qd_) * 4u * rowb; };      // variants past the end read the last one
    auto load_tile = [&](int qd_, uint32_t off, int t) -> double4_t {
        const char *base = reinterpret_cast<const char *>(F) + (size_t)qd_ * 16 * S * 8;
        double4_t x;
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = *reinterpret_cast<const double *>(base + (size_t)(off + (uint32_t)e * rowb + so[t]));
        return x;
    };
    constexpr int TPRE = (16 * GP + 63) / 64;
    int tk_src[TPRE], tk_vv[TPRE], tk_dst[TPRE];                                // value k = lane + 64 i of a quad's rows: where it is read, its variant, where it goes
#pragma unroll
    for (int i = 0; i < TPRE; ++i) {
        const int k = lane + 64 * i;
        const int kc = k < 16 * G ? k : 16 * G - 1;
        const int vv = kc / (4 * G), r = (kc / G) & 3, g = kc % G;
        tk_src[i] = kc; tk_vv[i] = vv;
        tk_dst[i] = k < 16 * G ? (4 * r + vv) * GP + g : -1;
    }
    auto tau_fetch = [&](int qd_, double (&tp)[TPRE]) {
#pragma unroll
        for (int i = 0; i < TPRE; ++i) tp[i] = (4 * qd_ + tk_vv[i] < V) ? tau_in[(size_t)qd_ * 16 * G + tk_src[i]] : 0.0;
    };
    auto tau_stage = [&](const double (&tp)[TPRE]) {
#pragma unroll
        for (int i = 0; i < TPRE; ++i) {
            if (tk_dst[i] >= 0) {
                told[tk_dst[i]] = tp[i];
                if (!do_update) tnew[tk_dst[i]] = tp[i];
            }
        }
    };
    if (qd < nquad) {
        uint32_t off = f_off(qd);
        double4_t f[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) f[t] = load_tile(qd, off, t);
        const bool ahead = fusedfix || do_update;            // (the statistics-only launch, once per factorize, reads the rows to the end of its round: it stages there)
        double tp[TPRE];
        tau_fetch(qd, tp);
        tau_stage(tp);
        __builtin_amdgcn_wave_barrier();
        for (;;) {
            NM_CLK(0); NM_CLKQ();
            const int qn = qd + qstride;
            const bool more = qn < nquad;
            const int qp = more ? qn : qd;                                      // what is fetched ahead (after the last quad: itself again, from cache)
            const uint32_t offp = f_off(qp);
            const int v0 = qd * 4;
            const bool vok = v0 + q < V;                                        // this lane's variant exists
            const bool live_last = vok && slive_last;
            if (ahead) tau_fetch(qp, tp);
            NM_CLK(1);
            quad_work([&](int t) { return f[t]; }, [&](int t) { return t == NT - 1 ? live_last : vok; },
                      [&](int t) { f[t] = load_tile(qp, offp, t); }, [&]() { if (ahead) { tau_stage(tp); __builtin_amdgcn_wave_barrier(); } }, v0, vok);
            if (!ahead) { tau_fetch(qp, tp); tau_stage(tp); __builtin_amdgcn_wave_barrier(); }
            if (!more) break;
            qd = qn;
            off = offp;
        }
    }
    }
    // workgroup reduction over the 4 wavefronts (fixed order) -> transposed partials.  acc[t][e]: g = 4 e + q, s = 16 t + n
    NM_CLK(0);
    __syncthreads();
    if (gnum) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int g = 4 * e + q;
                if (g < GP) red[((size_t)wv * (GP + 2) + g) * SPAD + 16 * t + n] = acc[t][e];
            }
    }
    // objective: one value per lane; H1: lane (n = g, q) holds the sum over bases and this lane's variants of tau_new[.][g]
    {
        const double o = group_allreduce_sum<64>(obj);
        double hh = h1;                                                        // sum over q (lanes n, n + 16, n + 32, n + 48)
        hh += __shfl_xor(hh, 16, 64);
        hh += __shfl_xor(hh, 32, 64);
        if (lane == 0) red[FIXF ? (size_t)wv : ((size_t)wv * (GP + 2) + GP) * SPAD] = o;     // (the fused pass reduces nothing else: four words)
        if (gnum && lane < GP) red[((size_t)wv * (GP + 2) + GP + 1) * SPAD + lane] = hh;
    }
    __syncthreads();
    for (int i = tid; gnum && i < GP * SPAD; i += 256) {                        // (walks the padded tile: the divisions are by constants)
        const int g = i / SPAD, s = i - g * SPAD;
        if (g < G && s < S) {
            double a = 0.0;
            for (int k = 0; k < 4; ++k) a += red[((size_t)k * (GP + 2) + g) * SPAD + s];
            partial[((size_t)g * S + s) * nblk + blockIdx.x] = a;
        }
    }
    if (gnum && tid < G) {
        double a = 0.0;
        for (int k = 0; k < 4; ++k) a += red[((size_t)k * (GP + 2) + GP + 1) * SPAD + tid];
        partial[((size_t)G * S + tid) * nblk + blockIdx.x] = a;
    }
    if (tid == 64) {
        double a = 0.0;
        for (int k = 0; k < 4; ++k) a += red[FIXF ? (size_t)k : ((size_t)k * (GP + 2) + GP) * SPAD];
        partial[((size_t)G * S + G) * nblk + blockIdx.x] = a;
    }
#ifdef DSM_AB_SWITCHES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NM_CLK(8);
    if (blockIdx.x == 0 && threadIdx.x == 0) { nm_dbg[9] += __builtin_amdgcn_s_memrealtime() - nm_r0; nm_dbg[10] += __builtin_amdgcn_s_memtime() - nm_m0; }
#endif
}

constexpr int nmft_mfma_wgs(int NT, int KB)
{
    // ((3, 2) and (4, 3) need 129 registers: at four wavefronts per SIMD they would spill one -- and a kernel with scratch pays its
    // first launch ~120 us for the allocation)
#ifdef NM_WGS56
    if (NT >= 5) return NM_WGS56;
#endif
    // five tiles and more: two (256 registers: the loop that looks ahead holds the next quad's rows and addresses too; measured at
    // 50k x 96 x 12 before it: 101.7 us per update at two against 106.6 at three with 22 registers spilled)
    // up to four tiles: four where 128 registers hold the kernel without scratch (round 5, with the statistics tile as straight-line code: not at
    // three and four tiles), else three
    // (round 6: up to four haplotypes the four-block contractions need 148 / 158 registers at five / six tiles -- three again: 56.5 -> 55.2 us
    // at 50k x 96 x 4, profiles/r06_nmft_b4.txt)
    return NT <= 4 ? ((KB <= 3 && NT <= 2) ? 4 : 3) : (KB == 1 ? 3 : 2);
}

// the fused pass of factorize_tau (no gamma numerators in registers, one gamma matrix and a four-word reduction in LDS): one
// workgroup per CU more
constexpr int nmft_mfma_fix_wgs(int NT, int KB) { return NT <= 4 ? (((NT == 1 && KB <= 3) || (NT == 2 && KB <= 2)) ? 5 : 4) : 3; }      // (five / six tiles: the loop that looks ahead)
static bool mfma_shape(const dsm_ctx *c, int *nt, int *kb)
{
    *nt = (c->S + 15) / 16;
    *kb = (c->nG + 3) / 4;
    static const bool off = DSM_AB_ENV("DESMAN_HIP_NMFT_NO_MFMA") != nullptr;      // A/B switch: the VALU one-pass kernel
    // measured against the VALU one-pass kernel: 1.0-1.3x at (NT, KB) = (4, 2), 1.96x at (6, 3) [V = 50k, S = 96, G = 12:
    // 211 vs 413 us per update]; at (8, 4) the 140 KB of LDS leave one workgroup per CU and the VALU kernel wins (449 vs 491 us)
    return !off && *nt >= 1 && *nt <= 8 && *kb >= 1 && *kb <= 4;          // S <= 128, G <= 16
}

bool nmft_use_mfma(const dsm_ctx *c) { int a, b; return mfma_shape(c, &a, &b); }

int nmft_mfma_grid(const dsm_ctx *c, bool fix)
{
    int g = ((c->V + 3) / 4 + 3) / 4;             // quads / 4 wavefronts
    // at most the workgroups that are resident at once (the quad loop is grid-strided): three per CU by registers up to four
    // sample tiles, two from five tiles on
    int nt, kb, cus = 256;
    (void)mfma_shape(c, &nt, &kb);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus < 1) cus = 256;
    // (a context's partial table is laid out for the larger of the two grids: dsm_nmft_set asks with fix = true)
    int cap = (fix ? nmft_mfma_fix_wgs(nt, kb) : nmft_mfma_wgs(nt, kb)) * cus;
    static const int wgs_env = DSM_AB_ENV("DESMAN_HIP_NMFT_WGS") ? atoi(DSM_AB_ENV("DESMAN_HIP_NMFT_WGS")) : 0;       // A/B switch: fewer resident workgroups per CU
    if (wgs_env > 0 && wgs_env * cus < cap) cap = wgs_env * cus;
    if (g > cap) g = cap;
    return g < 1 ? 1 : g;
}
static size_t mfma_lds_bytes(int NT, int KB, bool fix = false)
{
    const size_t GP = 4 * KB, SPAD = 16 * NT;
    const size_t loop = 2 * DSM_LOG_TAB_N + (fix ? 1 : 2) * GP * (SPAD + 1) + GP + 4 * 2 * 16 * GP + 4 * NM_XQ, red = fix ? 4 : 4 * (GP + 2) * SPAD;
    return std::max(loop, red) * sizeof(double);
}
// workgroups per CU the register allocation leaves room for (round 4, with the numerators on the matrix cores -- no qp[NT], no
// butterfly -- and one padded gamma matrix per factor in LDS): four up to four sample tiles (three at sixteen haplotypes), three from
// five tiles on (two at sixteen haplotypes: LDS)
template <int NT, int KB, bool KEEPF>
__global__ __launch_bounds__(256, nmft_mfma_wgs(NT, KB)) void nmft_mfma_kernel(NmftMfmaParams q) { nmft_mfma_body<NT, KB, KEEPF>(q); }
template <int NT, int KB, bool KEEPF>
__global__ __launch_bounds__(256, nmft_mfma_fix_wgs(NT, KB)) void nmft_mfma_fix_kernel(NmftMfmaParams q) { nmft_mfma_body<NT, KB, KEEPF, true>(q); }
template <int NT, int KB, bool KEEPF>
__global__ __launch_bounds__(256, nmft_mfma_fix_wgs(NT, KB)) void nmft_mfma_fix_kernel_b(BatchArgs<NmftMfmaParams> b)
{
    nmft_mfma_body<NT, KB, KEEPF, true>(b.p[blockIdx.y]);
}
template <int NT, int KB, bool KEEPF>
__global__ __launch_bounds__(256, nmft_mfma_wgs(NT, KB)) void nmft_mfma_kernel_b(BatchArgs<NmftMfmaParams> b)
{
    nmft_mfma_body<NT, KB, KEEPF>(b.p[blockIdx.y]);
}

template <int NT, int KB>
static void launch_mfma(dsm_ctx *c, int adjust, int do_update, int grid)
{
    const size_t sh = mfma_lds_bytes(NT, KB);
    NmftMfmaParams q{c->F, c->ntau, c->ngam_raw, c->ngam, c->V, c->S, c->nG, adjust, do_update, NMFT_CTL(c), c->log_tab, c->npart, c->ntau2, c->nmft_fix_gamma};
    if (c->nmft_gstep && do_update && !c->nmft_fix_gamma && !g_batch.K) {       // the launch begins with the gamma / control step (NmftMfmaParams.gstep)
        const int p = c->nmft_gstep_parity & 1;
        q.gstep = 1; q.parity = p; q.max_iter = c->nmft_gstep_max_iter; q.min_change = c->nmft_gstep_min_change;
        q.stat = c->nstat; q.div_trace = c->ndiv_trace;
        q.gam = p ? c->ngam2 : c->ngam; q.gam_raw = p ? c->ngam_raw2 : c->ngam_raw;
        q.gam_out = p ? c->ngam : c->ngam2; q.gam_raw_out = p ? c->ngam_raw : c->ngam_raw2;
    }
    if (c->nmft_fix_gamma == 2 && do_update) {           // gamma fixed: the fused pass (its own instantiation)
        const size_t shf = mfma_lds_bytes(NT, KB, true);
        LAUNCH_OR_COLLECT(NmftMfmaParams, q,
                          hipLaunchKernelGGL((nmft_mfma_fix_kernel<NT, KB, (NT <= 3 || NT == 5 || NT == 6)>), dim3(grid), dim3(256), shf, c->stream, q),
                          hipLaunchKernelGGL((nmft_mfma_fix_kernel_b<NT, KB, (NT <= 3 || NT == 5 || NT == 6)>), dim3(grid, K), dim3(256), shf, c->stream, acc));
        return;
    }
    LAUNCH_OR_COLLECT(NmftMfmaParams, q,
                      hipLaunchKernelGGL((nmft_mfma_kernel<NT, KB, (NT <= 3 || NT == 5 || NT == 6)>), dim3(grid), dim3(256), sh, c->stream, q),
                      hipLaunchKernelGGL((nmft_mfma_kernel_b<NT, KB, (NT <= 3 || NT == 5 || NT == 6)>), dim3(grid, K), dim3(256), sh, c->stream, acc));
}

int k_nmft_mfma(dsm_ctx *c, int adjust, int do_update)
{
    KTimer tm(c, do_update ? DSM_K_NMFT_B : DSM_K_NMFT_A);
    int nt, kb;
    if (!mfma_shape(c, &nt, &kb)) { dsm_set_error("nmft_mfma: unsupported shape"); return DSM_ERR_UNSUPPORTED; }
    const int grid = nmft_mfma_grid(c, c->nmft_fix_gamma == 2 && do_update);
#define MCASE(N, K) if (nt == N && kb == K) launch_mfma<N, K>(c, adjust, do_update, grid)
    MCASE(1, 1); MCASE(1, 2); MCASE(2, 1); MCASE(2, 2); MCASE(3, 1); MCASE(3, 2); MCASE(4, 1); MCASE(4, 2);
    MCASE(1, 3); MCASE(2, 3); MCASE(3, 3); MCASE(4, 3); MCASE(5, 1); MCASE(5, 2); MCASE(5, 3); MCASE(6, 1); MCASE(6, 2); MCASE(6, 3);
    MCASE(7, 1); MCASE(7, 2); MCASE(7, 3); MCASE(8, 1); MCASE(8, 2); MCASE(8, 3);
    MCASE(1, 4); MCASE(2, 4); MCASE(3, 4); MCASE(4, 4); MCASE(5, 4); MCASE(6, 4); MCASE(7, 4); MCASE(8, 4);
#undef MCASE
    HIP_TRY(hipGetLastError());
    c->npart_cols = grid;
#ifdef DSM_AB_SWITCHES
    if (DSM_AB_ENV("DESMAN_HIP_NMFT_STAMPS") && do_update) {
        static int calls = 0;
        unsigned long long h[12];
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(nm_dbg), sizeof h));
        if (++calls == 20 && h[7]) {
            const double qn = (double)h[7];
            fprintf(stderr, "nmft_mfma quad loop, workgroup 0 wavefront 0, cycles per quad over %.0f quads: top+issue %.0f | load wait %.0f | tau half %.0f | statistics %.0f | loop %.0f\n",
                    qn, h[1] / qn, h[2] / qn, h[3] / qn, h[4] / qn, h[0] / qn);
            fprintf(stderr, "   whole kernel (that wavefront): %llu cycles = %.1f us (s_memrealtime), i.e. %.0f MHz; prologue %llu, epilogue (reduction + partials) %llu cycles\n",
                    h[10], h[9] / 100.0, h[9] ? 100.0 * (double)h[10] / (double)h[9] : 0.0, h[5], h[8]);
        }
        unsigned long long z[12] = {0, 0, 0, 0, 0, 0, (unsigned long long)(atoi(DSM_AB_ENV("DESMAN_HIP_NMFT_STAMPS")) > 1), 0, 0, 0, 0, 0};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(nm_dbg), z, sizeof z));
    }
#endif
    return DSM_OK;
}

// ===========================================================================
// nmft_split_kernel (round 5; takes the place of round 3's nmft_wide_kernel): the update of nmft_mfma_kernel for tables wider than
// eight sample tiles (128 < S <= 512).
//
// A wavefront cannot hold more than eight tiles of per-sample state, so a quad of variants is SHARED by NCB = 4 wavefronts, each
// owning a block of NT consecutive sample tiles (tile tg = cb NT + t): everything per sample -- R', Q', R2, Q2, the objective terms,
// the gamma numerators of its columns -- is the narrow kernel's code on the block, in its round-5 form: tau numerators on the matrix
// cores through the per-wavefront transposition tile (nm_num_tile; round 3's wide kernel ran them on the vector ALU with a butterfly
// per K-block), ONE zero-padded gamma matrix per factor for every contraction that reads it (was: two fragment-ordered copies), the
// statistics tile as straight-line code (nm_tile_q2), F addressed by a wave-uniform base + a 32-bit lane offset with clamped rows /
// samples instead of exec-masked loads, and tile t of the NEXT quad loaded into the registers tile t of this one has just left.
// 10k x 300 x 8 114 -> 75 us per update, 5k x 512 x 8 83 -> 74, 10k x 192 x 8 59 -> 52 (profiles/r05_nmft_split_ab.txt).
//
// What crosses blocks is the tau numerator num[row][g] = sum over ALL samples: every wavefront leaves its block's part in LDS,
// ONE workgroup barrier per quad (round 3: three), and every wavefront of the quad adds the NCB parts in block order and runs the
// (tiny) tau update for itself -- identical numbers in every block; block 0 stores them.  Everything else a block reads it wrote
// itself or is written with identical values by its peers (the quad's tau rows: old rows in two buffers by quad parity, so that a
// wavefront may stage the next quad's rows while a peer still reads this one's; new rows in one).  The exchange area doubles as
// the transposition tile of nm_num_tile and alternates by quad parity for the same reason (XPAR; where LDS is short -- the widest
// tables -- one buffer and a second barrier).
// A workgroup is eight wavefronts = NQ = 8 / NCB quads in flight; every wavefront of it runs the same number of rounds (a slot
// past the end works on a quad of absent variants).  Statistics: a wavefront owns its columns' gamma numerators outright; the NQ
// quads in flight add up slot by slot through one [GP][SPAD] buffer that takes the operands' place at the end.  Same partial
// layout as the other update kernels: the reduce / gamma / control launches are shared.  Batched entry: nmft_split_kernel_b.
// ===========================================================================
template <int NT, int KB, int NCB, bool XPAR>
__device__ __forceinline__ void nmft_split_body(const NmftMfmaParams &prm)
{
    constexpr int NQ = 8 / NCB, NW = 8, NTT = NT * NCB, GP = 4 * KB, SPAD = 16 * NTT, NTHR = 64 * NW, LDG = SPAD + 1;
    const double *__restrict__ F = prm.F, *__restrict__ gam_raw = prm.gam_raw, *__restrict__ gam = prm.gam;
    double *__restrict__ tau = prm.tau, *__restrict__ partial = prm.partial;
    const double *__restrict__ ctl = prm.ctl, *__restrict__ log_tab = prm.log_tab;
    const int V = prm.V, S = prm.S, G = prm.G, adjust = prm.adjust, do_update = prm.do_update;
    const bool gnum = prm.fix_gamma == 0;            // gamma fixed (factorize_tau): no gamma numerators, no row sums, the objective's partial alone
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    if (ctl[2] != 0.0) return;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = wv / NCB, cb = wv % NCB, nblk = gridDim.x;
    double2 *ltab = reinterpret_cast<double2 *>(smem_w);                        // [256]
    double *graw_p = reinterpret_cast<double *>(ltab + DSM_LOG_TAB_N);          // [GP][LDG] gamma_raw, zero-padded (nm_stage_gamma_p)
    double *ggam_p = graw_p + GP * LDG;                                         // [GP][LDG] gamma
    double *t1 = ggam_p + GP * LDG;                                             // [GP] rowsum(gamma_raw)
    double *rows = t1 + GP + slot * (3 * 16 * GP);                              // per quad in flight: old rows x 2 (quad parity), new rows
    double *tnew = rows + 2 * 16 * GP;
    double *xall = t1 + GP + NQ * (3 * 16 * GP);                                // per wavefront [XPAR ? 2 : 1][NM_XQ]: transposition tile / part of num
    constexpr int XW = (XPAR ? 2 : 1) * NM_XQ;
    double *objw = xall + NW * XW;                                              // [NW]
    double *h1w = objw + NW;                                                    // [NQ][GP]
    double *red = graw_p;                                                       // [GP][SPAD] at the end
    for (int i = tid; i < DSM_LOG_TAB_N; i += NTHR) ltab[i] = reinterpret_cast<const double2 *>(log_tab)[i];
    nm_stage_gamma_p(graw_p, ggam_p, gam_raw, gam, GP, LDG, G, S, tid, NTHR);
    for (int k = tid; k < NQ * 3 * 16 * GP; k += NTHR) t1[GP + k] = 0.0;        // the rows incl. the padded haplotype columns
    for (int g = wv; g < GP; g += NW) {                                         // gamma.sum(1) (:170), lane-parallel
        double a = 0.0;
        if (g < G) for (int s2 = lane; s2 < S; s2 += 64) a += gam_raw[(size_t)g * S + s2];
        a = group_allreduce_sum<64>(a);
        if (lane == 0) t1[g] = a;
    }
    __syncthreads();

    double4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double obj = 0.0, h1 = 0.0;

    // F in L2 (nmft_mfma_kernel): f[t][e] = F[variant v0 + q][base e][16 tg + n].  Address = the quad's first row (wave-uniform) + a
    // 32-bit lane offset; lanes without a cell read a cell that exists (the last variant, the last sample): R = 0 there, any finite
    // F will do (nm_tile_q2).
    const int nquad = (V + 3) >> 2;
    const int qstride = nblk * NQ;
    const uint32_t rowb = (uint32_t)S * 8u;
    uint32_t so[NT];                                                            // byte offset of this lane's sample in tile t of its block
    bool slive[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int s2 = 16 * (cb * NT + t) + n;
        slive[t] = s2 < S;
        so[t] = (uint32_t)(s2 < S ? s2 : S - 1) * 8u;
    }
    auto quad_addr = [&](int qd_) { return min(qd_, nquad - 1); };              // a slot past the end reads the last quad (and keeps nothing of it)
    auto f_off = [&](int qa) -> uint32_t { return (uint32_t)min(q, V - 1 - 4 * qa) * 4u * rowb; };
    auto load_tile = [&](int qa, uint32_t off, int t) -> double4_t {
        const char *base = reinterpret_cast<const char *>(F) + (size_t)qa * 16 * S * 8;
        double4_t x;
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = *reinterpret_cast<const double *>(base + (size_t)(off + (uint32_t)e * rowb + so[t]));
        return x;
    };
    constexpr int TPRE = (16 * GP + 63) / 64;
    auto tau_fetch = [&](int qd_, double (&tp)[TPRE]) {
#pragma unroll
        for (int i = 0; i < TPRE; ++i) {
            const int k = lane + 64 * i;
            const int kc = k < 16 * G ? k : 16 * G - 1;
            const int vv = kc / (4 * G);
            tp[i] = (4 * qd_ + vv < V) ? tau[(size_t)min(qd_, nquad - 1) * 16 * G + kc] : 0.0;
        }
    };
    auto tau_stage = [&](const double (&tp)[TPRE], double *told_, bool also_new) {
#pragma unroll
        for (int i = 0; i < TPRE; ++i) {
            const int k = lane + 64 * i;
            if (k < 16 * G) {
                const int vv = k / (4 * G), r = (k / G) & 3, g = k % G;
                told_[(4 * r + vv) * GP + g] = tp[i];
                if (also_new) tnew[(4 * r + vv) * GP + g] = tp[i];
            }
        }
    };

    int qd = blockIdx.x * NQ + slot;                                            // wave-uniform
    int par = 0;
    {
        double tp0[TPRE];
        tau_fetch(qd, tp0);
        tau_stage(tp0, rows, !do_update);
    }
    const int qa0 = quad_addr(qd);
    uint32_t off = f_off(qa0);
    double4_t f[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) f[t] = load_tile(qa0, off, t);
    __builtin_amdgcn_wave_barrier();

    for (int qd0 = blockIdx.x * NQ; qd0 < nquad; qd0 += qstride) {              // every wavefront of the workgroup runs the same rounds
        const int qn = qd + qstride;
        const int qpa = quad_addr(qn);
        const uint32_t offp = f_off(qpa);
        const int v0 = qd * 4;
        const bool vok = v0 + q < V;                                            // this lane's variant exists
        double *told = rows + par * (16 * GP), *toldn = rows + (par ^ 1) * (16 * GP);
        double *xq = xall + wv * XW + (XPAR ? par * NM_XQ : 0);
        double tp[TPRE];
        if (do_update) {
            tau_fetch(qn, tp);                                                  // the next quad's rows: staged when this one's have been read for the last time
            double a_old[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) a_old[kb] = told[n * GP + 4 * kb + q];
            double4_t num = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int tg = cb * NT + t;
                double4_t R = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) R = NM_MFMA(a_old[kb], graw_p[(4 * kb + q) * LDG + 16 * tg + n], R, 0, 0, 0);
                const double4_t qv = nm_div_tile(f[t], R);                              // nm_tile_q2: F > 0; lanes without a cell stay finite
                num = nm_num_tile<KB>(num, qv, xq, graw_p + 16 * tg, LDG, n, q);
            }
            // this block's part of num -> LDS; all parts of the quad in block order
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 4; ++e) xq[e * 64 + lane] = num[e];
            __syncthreads();
            double4_t tot = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int b = 0; b < NCB; ++b) {
                const double *xb = xall + (slot * NCB + b) * XW + (XPAR ? par * NM_XQ : 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) tot[e] = (b == 0) ? xb[e * 64 + lane] : tot[e] + xb[e * 64 + lane];
            }
            if constexpr (!XPAR) __syncthreads();                               // one buffer: nobody overwrites a part a peer has not read yet
            nm_tau_finish<KB, true>(tot, told, tnew, t1, G, n, q, adjust, vok && cb == 0, vok, tau + (size_t)(v0 + q) * 4 * G);
            __builtin_amdgcn_wave_barrier();
            tau_stage(tp, toldn, false);
            __builtin_amdgcn_wave_barrier();
        }
        // statistics of the (new) state on this block's columns: R2, objective, Q2, gamma numerators; H1 by block 0
        double a_new[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) a_new[kb] = tnew[n * GP + 4 * kb + q];
        constexpr bool HOLDAG = !nm_b4g(KB);
        NmAg<KB> a_g;                                                           // A of the row contraction (nm_gnum_tile); the four-block form's fetched tile by tile
        if constexpr (HOLDAG) nm_ag_load<KB>(a_g, tnew, gnum, n, q);
        if (cb == 0) nm_h1_add<KB>(h1, a_g, tnew, gnum, n, q);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tg = cb * NT + t;
            double4_t R = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) R = NM_MFMA(a_new[kb], ggam_p[(4 * kb + q) * LDG + 16 * tg + n], R, 0, 0, 0);
            const double4_t q2 = nm_tile_q2(f[t], R, vok && slive[t], ltab, obj);
            f[t] = load_tile(qpa, offp, t);                                     // this tile of the next quad
            nm_ag_tile<KB, HOLDAG>(a_g, tnew, gnum, n, q);
            if (gnum) nm_gnum_tile<KB>(acc[t], a_g, q2);
        }
        __builtin_amdgcn_wave_barrier();
        if (!do_update) {                                                       // statistics only (once per factorize): the next quad's rows, in place
            __syncthreads();                                                    // every block of the quad has read the rows
            tau_fetch(qn, tp);
            tau_stage(tp, rows, true);
            __builtin_amdgcn_wave_barrier();
        } else {
            par ^= 1;
        }
        qd = qn;
        off = offp;
    }
    // objective: per wavefront; H1: lane (n = g, q) of a block-0 wavefront holds the sum over bases and its variants of tau_new[.][g]
    {
        const double o = group_allreduce_sum<64>(obj);
        double hh = h1;
        hh += __shfl_xor(hh, 16, 64);
        hh += __shfl_xor(hh, 32, 64);
        if (lane == 0) objw[wv] = o;
        if (cb == 0 && lane < GP) h1w[slot * GP + lane] = hh;
    }
    // gamma numerators: acc[t][e] is (g = 4 e + q, s = 16 tg + n) of this wavefront's columns; the quads in flight add up slot by slot
#pragma unroll 1
    for (int sl = 0; gnum && sl < NQ; ++sl) {
        __syncthreads();
        if (slot == sl) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int g = 4 * e + q;
                    if (g < GP) {
                        double *r = red + (size_t)g * SPAD + 16 * (cb * NT + t) + n;
                        *r = (sl == 0) ? acc[t][e] : *r + acc[t][e];
                    }
                }
        }
    }
    __syncthreads();
    for (int i = tid; gnum && i < G * S; i += NTHR) {
        const int g = i / S, s2 = i % S;
        partial[(size_t)i * nblk + blockIdx.x] = red[(size_t)g * SPAD + s2];
    }
    if (gnum && tid < G) {
        double a = 0.0;
        for (int k = 0; k < NQ; ++k) a += h1w[k * GP + tid];
        partial[((size_t)G * S + tid) * nblk + blockIdx.x] = a;
    }
    if (tid == 64) {
        double a = 0.0;
        for (int k = 0; k < NW; ++k) a += objw[k];
        partial[((size_t)G * S + G) * nblk + blockIdx.x] = a;
    }
}

static size_t split_lds_bytes(int NT, int KB, int NCB, bool xpar)
{
    const size_t NQ = 8 / NCB, NTT = (size_t)NT * NCB, GP = 4 * KB, SPAD = 16 * NTT;
    return (2 * DSM_LOG_TAB_N + 2 * GP * (SPAD + 1) + GP + NQ * 3 * 16 * GP + 8 * (xpar ? 2 : 1) * NM_XQ + 8 + NQ * GP) * sizeof(double);
}
// two exchange buffers where they fit beside two workgroups per CU's worth of everything else, or at least into the CU
constexpr bool split_xpar(int NT, int KB, int NCB)
{
    return (2 * DSM_LOG_TAB_N + 2 * (4 * KB) * (16 * NT * NCB + 1) + 4 * KB + (8 / NCB) * 3 * 16 * (4 * KB) + 8 * 2 * NM_XQ + 8 + (8 / NCB) * 4 * KB) * 8 <= 160 * 1024;
}
// two wavefronts per SIMD (the second argument of HIP's __launch_bounds__) = one workgroup per CU, 256 registers: measured against four
// (two workgroups per CU where three-tile blocks and the LDS allow it: 128 registers, 23-30 of them spilled) 52 vs 57 us per update at
// 10k x 192 x 8, equal at 10k x 300 x 8 (profiles/r05_nmft_split_ab.txt)
constexpr int split_wgs(int, int, int) { return 2; }
template <int NT, int KB, int NCB>
__global__ __launch_bounds__(512, split_wgs(NT, KB, NCB)) void nmft_split_kernel(NmftMfmaParams q) { nmft_split_body<NT, KB, NCB, split_xpar(NT, KB, NCB)>(q); }
template <int NT, int KB, int NCB>
__global__ __launch_bounds__(512, split_wgs(NT, KB, NCB)) void nmft_split_kernel_b(BatchArgs<NmftMfmaParams> b)
{
    nmft_split_body<NT, KB, NCB, split_xpar(NT, KB, NCB)>(b.p[blockIdx.y]);
}

// Which tables take the split kernel, and how: 128 < S <= 512 -- four blocks of three to eight tiles, the smallest that hold S.  G <= 16.
// (Two blocks of three / four tiles for 64 < S <= 128 were measured too: 104 vs 103 us per update at 50k x 96 x 12, 70 vs 61 at
// 30k x 80 x 10 with its padded tile -- one wavefront per quad stays the form there, profiles/r05_nmft_split_ab.txt.)
static bool wide_shape(const dsm_ctx *c, int *nt, int *kb, int *ncb)
{
    static const bool off = DSM_AB_ENV("DESMAN_HIP_NMFT_NO_MFMA") != nullptr || DSM_AB_ENV("DESMAN_HIP_NMFT_NO_WIDE") != nullptr;
    const int tiles = (c->S + 15) / 16;
    *kb = (c->nG + 3) / 4;
    if (off || tiles <= 8 || tiles > 32 || *kb < 1 || *kb > 4) return false;
    static const int shapes[5][2] = {{3, 4}, {4, 4}, {5, 4}, {6, 4}, {8, 4}};       // by capacity: 12, 16, 20, 24, 32 tiles
    for (int i = 0; i < 5; ++i)
        if (shapes[i][0] * shapes[i][1] >= tiles) {
            *nt = shapes[i][0]; *ncb = shapes[i][1];
            return split_lds_bytes(*nt, *kb, *ncb, false) <= 160 * 1024;       // (not S > 384 with G > 12: two padded gamma matrices of 66 KB each; the two-pass kernels take it)
        }
    return false;
}
bool nmft_use_wide(const dsm_ctx *c) { int a, b, d; return wide_shape(c, &a, &b, &d); }

int nmft_wide_grid(const dsm_ctx *c)
{
    int nt, kb, ncb, cus = 256;
    if (!wide_shape(c, &nt, &kb, &ncb)) return 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus < 1) cus = 256;
    const int nq = 8 / ncb;
    int g = ((c->V + 3) / 4 + nq - 1) / nq;
    const int cap = cus * (split_wgs(nt, kb, ncb) / 2);
    if (g > cap) g = cap;                          // the workgroups that are resident at once (the rounds are grid-strided)
    return g < 1 ? 1 : g;
}

template <int NT, int KB, int NCB>
static int launch_wide(dsm_ctx *c, int adjust, int do_update, int grid)
{
    const size_t sh = split_lds_bytes(NT, KB, NCB, split_xpar(NT, KB, NCB));
    if (sh > 160 * 1024) { dsm_set_error("nmft_split: %zu B of LDS", sh); return DSM_ERR_UNSUPPORTED; }
    static bool attr_set = false;                  // more than 64 KB of dynamic LDS has to be asked for, once per instantiation
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&nmft_split_kernel<NT, KB, NCB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&nmft_split_kernel_b<NT, KB, NCB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const NmftMfmaParams q{c->F, c->ntau, c->ngam_raw, c->ngam, c->V, c->S, c->nG, adjust, do_update, NMFT_CTL(c), c->log_tab, c->npart, c->ntau2, c->nmft_fix_gamma};
    LAUNCH_OR_COLLECT(NmftMfmaParams, q,
                      hipLaunchKernelGGL((nmft_split_kernel<NT, KB, NCB>), dim3(grid), dim3(512), sh, c->stream, q),
                      hipLaunchKernelGGL((nmft_split_kernel_b<NT, KB, NCB>), dim3(grid, K), dim3(512), sh, c->stream, acc));
    return DSM_OK;
}

int k_nmft_wide(dsm_ctx *c, int adjust, int do_update)
{
    KTimer tm(c, do_update ? DSM_K_NMFT_B : DSM_K_NMFT_A);
    int nt, kb, ncb;
    if (!wide_shape(c, &nt, &kb, &ncb)) { dsm_set_error("nmft_split: unsupported shape"); return DSM_ERR_UNSUPPORTED; }
    const int grid = nmft_wide_grid(c);
    int rc = DSM_ERR_UNSUPPORTED;
#define WCASE(N, B) if (nt == N && ncb == B) { if (kb == 1) rc = launch_wide<N, 1, B>(c, adjust, do_update, grid); else if (kb == 2) rc = launch_wide<N, 2, B>(c, adjust, do_update, grid); \
                                               else if (kb == 3) rc = launch_wide<N, 3, B>(c, adjust, do_update, grid); else rc = launch_wide<N, 4, B>(c, adjust, do_update, grid); }
    WCASE(3, 4); WCASE(4, 4); WCASE(5, 4); WCASE(6, 4); WCASE(8, 4);
#undef WCASE
    if (rc != DSM_OK) return rc;
    HIP_TRY(hipGetLastError());
    c->npart_cols = grid;
    return DSM_OK;
}

// ===========================================================================
// nmft_persist_kernel: the WHOLE factorize loop (Init_NMFT.py:98-115, :134-149) in ONE launch for tables whose quads of
// variants all fit on the machine at once (V <= 48 x compute units = 12 288 on MI355X).
//
// Why: at config 3 an update of the three-launch path is 34 us of which ~11 us are the two halves of the update itself --
// the rest is two dependent launches (reduction, gamma / control) and the update kernel's own ramp.  Here the workgroups stay
// resident and an update crosses workgroups through two in-kernel grid barriers instead:
//   statistics of the current state  ->  every workgroup publishes its 521 partial sums (write-through stores)
//   barrier 1                         ->  wavefront j of the machine reduces statistic j over the workgroups (the arithmetic of
//                                         nmft_reduce_kernel: lane-strided sums, fixed-order butterfly) and publishes it
//   barrier 2                         ->  every workgroup reads the 521 totals, runs the stop test (:106) and the gamma update
//                                         (:163-168) FOR ITSELF -- the same numbers everywhere, nothing to broadcast --
//                                         re-stages its MFMA operands and runs the tau half of the update on its own quads,
//                                         whose tau rows never leave LDS between updates (HBM sees tau once, at the end)
// Measured in isolation (scripts/ubench/grid_barrier.hip, profiles/r03_grid_barrier.txt): a barrier of 625 x 256-thread
// workgroups costs 3.7 us and the exchange of one update 14.7 us (publish 3.1, reduce 3.2, read 0.7, two barriers 7.7); with
// 209 x 768-thread workgroups -- twelve quads each, one workgroup per CU, the geometry used here -- a third of the partial
// rows cross the machine and a barrier costs 2.6 us.
//
// The barrier (grid_barrier below): arrival on one of eight group counters (group = workgroup % 8: on this part workgroup b
// runs on XCD b % 8, so a group's counter stays in one XCD's reach -- a matter of speed only, nothing depends on placement),
// the last arriver of a group arrives on the top counter, the last of those publishes the generation word that one lane of
// every workgroup polls (relaxed agent-scope loads + s_sleep).  All shared words are agent-scope atomics; payloads are 8-byte
// write-through (sc1) stores drained by every storing wavefront before the arrival, and are read with sc1 loads, so no
// release / acquire fence is needed (cdna_hip_programming.md, Guideline 16).  Polls are bounded: a workgroup that times out
// raises an error word and leaves; the host then restores the factors it saved before the launch and runs the three-launch loop.  The launch requires every workgroup to be resident: the grid is
// checked against the occupancy query, and a process-wide gate (PersistGate) admits concurrent persistent launches on a device
// only while together they ask for at most one workgroup per CU (more could starve each other of CUs); other kernels only delay them.
//
// Arithmetic: the update is that of nmft_mfma_body, statement for statement, and a workgroup publishes the sum of the three
// partial rows the three-launch kernel's workgroups would have written for its twelve quads, added in the order in which
// every reduction of this file adds a triple of partials (nmft_sum_partials), so the totals -- hence the factors, the
// update count and the objective trace -- equal the three-launch path's bit for bit (a chain's NMF start does not depend on
// which path ran it: batched replicates, timing mode and large tables take the three-launch loop).
// ===========================================================================
typedef __attribute__((address_space(1))) unsigned nm_gu32;
typedef __attribute__((address_space(1))) unsigned long long nm_gu64;
#define NM_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define NMFT_P_WAVES 12       // wavefronts of a workgroup of the large form (three update-kernel workgroups' worth of quads); the small form has 4

struct NmftBarrier { unsigned *gcnt /* [8] x 64 B */, *top, *gen, *err; int members[8], ngroups /* groups with members */; };

// ok_s: one word of the caller's dynamic LDS (a static __shared__ here would shift the base of the dynamic region off its
// 16-byte alignment)
__device__ __forceinline__ bool nm_grid_barrier(const NmftBarrier &b, unsigned epoch, int tid, int *ok_s)
{
    __syncthreads();
    if (tid == 0) {
        int ok = 1;
        const unsigned g = blockIdx.x & 7u;
        const unsigned t = __hip_atomic_fetch_add((nm_gu32 *)(b.gcnt + g * 16), 1u, NM_RLX_AGENT);
        if (t + 1u == epoch * (unsigned)b.members[g]) {
            const unsigned t2 = __hip_atomic_fetch_add((nm_gu32 *)b.top, 1u, NM_RLX_AGENT);
            if (t2 + 1u == epoch * (unsigned)b.ngroups) __hip_atomic_store((nm_gu32 *)b.gen, epoch, NM_RLX_AGENT);
        }
        unsigned spins = 0;
        while (__hip_atomic_load((nm_gu32 *)b.gen, NM_RLX_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (__hip_atomic_load((nm_gu32 *)b.err, NM_RLX_AGENT) != 0u || ++spins > (1u << 24)) {     // ~seconds: not resident / a peer gave up
                __hip_atomic_store((nm_gu32 *)b.err, 1u, NM_RLX_AGENT);
                ok = 0;
                break;
            }
        }
        *ok_s = ok;
    }
    __syncthreads();
    return *ok_s != 0;
}

__device__ __forceinline__ void nm_store(double *p, double x) { __hip_atomic_store((nm_gu64 *)p, (unsigned long long)__double_as_longlong(x), NM_RLX_AGENT); }
__device__ __forceinline__ double nm_load(const double *p) { return __longlong_as_double((long long)__hip_atomic_load((nm_gu64 *)p, NM_RLX_AGENT)); }

struct NmftPersistParams {
    const double *F; double *tau; double *gam_raw, *gam;
    int V, S, G, adjust, fix_gamma, max_iter;
    double min_change;
    double *ctl, *div_trace; const double *log_tab;
    double *partial;               // [nout][workgroups]: one row per workgroup = the sum of a triple of partial rows of the three-launch kernel
    double *stat;                  // [nout]
    NmftBarrier bar;
    double *stamps;                // debug: s_memrealtime (100 MHz) at the phase boundaries of update 5 in workgroup 0, or null
};

// NWV = wavefronts per workgroup: 12 (large tables: one workgroup per CU, a third of the partial rows and arrivals) or 4 (tables of
// at most one update-kernel workgroup per CU: a wavefront has its SIMD to itself and the phases run ~1.5x faster)
template <int NT, int KB, bool KEEPF, int NWV>
__global__ __launch_bounds__(64 * NWV, (NWV == 12 ? 3 : 1)) void nmft_persist_kernel(NmftPersistParams prm)
{
    const double *__restrict__ F = prm.F;
    const int V = prm.V, S = prm.S, G = prm.G, adjust = prm.adjust;
    extern __shared__ __attribute__((aligned(16))) char smem_p[];
    constexpr int GP = 4 * KB, SPAD = 16 * NT, NW = NWV, NTHR = 64 * NWV;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, q = lane >> 4;
    const int nwg = gridDim.x, wg = blockIdx.x;
    const int nout = G * S + G + 1;
    double2 *ltab = reinterpret_cast<double2 *>(smem_p);                        // [256]
    constexpr int LDG = SPAD + 1;
    double *graw_p = reinterpret_cast<double *>(ltab + DSM_LOG_TAB_N);          // [GP][LDG] gamma_raw, zero-padded (nm_stage_gamma_p)
    double *ggam_p = graw_p + GP * LDG;                                         // [GP][LDG] gamma
    double *t1 = ggam_p + GP * LDG;                                             // [GP] rowsum(gamma_raw)
    double *tl = t1 + GP;                                                       // per wavefront [2][16][GP]: tau rows, old and new
    constexpr int REDW = ((GP + 2) * SPAD > NM_XQ) ? (GP + 2) * SPAD : NM_XQ;   // doubles per wavefront of the region below
    double *red = tl + NW * (2 * 16 * GP);                                      // [NW][GP + 2][SPAD] cross-wavefront reduction, also scratch; in the
                                                                                // tau half (nothing of it is live then) wavefront w's transposition
                                                                                // tile of Q' (nm_num_tile) sits at red + w * NM_XQ
    double *stat = red + NW * REDW;                                             // [nout] the reduced statistics
    double *gm = stat + ((nout + 1) & ~1);                                      // [G][S] gamma after _adjustment
    double *grw = gm + G * S;                                                   // [G][S] normalised gamma before _adjustment
    int *ok_s = reinterpret_cast<int *>(grw + G * S);                           // [1] verdict of a barrier for the whole workgroup
    for (int i = tid; i < DSM_LOG_TAB_N; i += NTHR) ltab[i] = reinterpret_cast<const double2 *>(prm.log_tab)[i];
    for (int i = tid; i < G * S; i += NTHR) { gm[i] = prm.gam[i]; grw[i] = prm.gam_raw[i]; }
    double *told = tl + wv * (2 * 16 * GP), *tnew = told + 16 * GP;
    for (int k = lane; k < 2 * 16 * GP; k += 64) told[k] = 0.0;                 // incl. tnew and the padded haplotype columns
    const int qd = wg * NW + wv, nquad = (V + 3) >> 2;
    const bool have = qd < nquad;
    const int v0 = qd * 4;
    const bool vok = have && v0 + q < V;
    __builtin_amdgcn_wave_barrier();
    if (have) {
        for (int k = lane; k < 16 * G; k += 64) {
            const int vv = k / (4 * G), r = (k / G) & 3, g = k % G;
            tnew[(4 * r + vv) * GP + g] = (v0 + vv < V) ? prm.tau[(size_t)v0 * 4 * G + k] : 0.0;
        }
    }
    double *xq = red + wv * NM_XQ;
    bool live[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) live[t] = vok && (16 * t + n < S);
    auto load_f = [&](int t) {
        double4_t x;
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = live[t] ? F[((size_t)(v0 + q) * 4 + e) * S + 16 * t + n] : 1.0;
        return x;
    };
    double4_t f[KEEPF ? NT : 1];
    if constexpr (KEEPF) {
#pragma unroll
        for (int t = 0; t < NT; ++t) f[t] = load_f(t);
    }
    // MFMA operands of the current gamma: gr / braw from grw, bgam from gm, t1 = rowsum(grw)
    auto stage_gamma = [&]() {
        nm_stage_gamma_p(graw_p, ggam_p, grw, gm, GP, LDG, G, S, tid, NTHR);
        for (int g = wv; g < GP; g += NW) {                                     // gamma.sum(1) (:170), lane-parallel
            double a = 0.0;
            for (int s = lane; s < SPAD; s += 64) a += (g < G && s < S) ? (prm.fix_gamma ? gm : grw)[g * S + s] : 0.0;
            a = group_allreduce_sum<64>(a);
            if (lane == 0) t1[g] = a;
        }
        __syncthreads();
    };
    __syncthreads();
    stage_gamma();

    unsigned epoch = 0;
    int it = 0;
    double prev = 0.0;
    bool alive = true;
#define NM_STAMP(k) do { if (prm.stamps && wg == 0 && tid == 0 && it == 5) prm.stamps[k] = (double)__builtin_amdgcn_s_memrealtime(); } while (0)
    for (;;) {
        NM_STAMP(0);
        // ---- statistics of the current state (tnew, gamma): R2, objective, Q2, gamma numerators, H1
        double4_t acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
        double obj = 0.0, h1 = 0.0;
        const bool gnum = prm.fix_gamma == 0;                                   // gamma fixed (factorize_tau): the objective alone
        if (have) {
            double a_new[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) a_new[kb] = tnew[n * GP + 4 * kb + q];
            constexpr bool HOLDAG = !nm_b4g(KB) || NWV == 4;                     // (the small form: one wavefront per SIMD)
            NmAg<KB> a_g;                                                       // A of the row contraction (nm_gnum_tile)
            if constexpr (HOLDAG) nm_ag_load<KB>(a_g, tnew, gnum, n, q);
            nm_h1_add<KB>(h1, a_g, tnew, gnum, n, q);
            double4_t num = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                double4_t R = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) R = NM_MFMA(a_new[kb], ggam_p[(4 * kb + q) * LDG + 16 * t + n], R, 0, 0, 0);
                double4_t q2;
                const double4_t ft = KEEPF ? f[KEEPF ? t : 0] : load_f(t);
                q2 = nm_tile_q2(ft, R, live[t], ltab, obj);
                if (gnum) nm_ag_tile<KB, HOLDAG>(a_g, tnew, gnum, n, q);
                if (gnum) {
                    nm_gnum_tile<KB>(acc[t], a_g, q2);
                } else {
                    // gamma fixed: this product is also the one the tau half of the update divides F by (NmftMfmaParams.fix_gamma == 2):
                    // the candidate rows of the next update come out of the same pass
                    num = nm_num_tile<KB>(num, q2, xq, ggam_p + 16 * t, LDG, n, q);
                }
            }
            if (!gnum) nm_tau_finish<KB, false>(num, tnew, told, t1, G, n, q, adjust, false, vok, nullptr);     // candidate -> the spare rows
        }
        NM_STAMP(1);
        // Workgroup reduction: wavefronts 4 r .. 4 r + 3 are workgroup 3 wg + r of the three-launch kernel (the same four quads, summed
        // in the same fixed order), and the three sums are added as nmft_sum_partials adds a triple of partials -- so this
        // workgroup's ONE published row is a term of the three-launch path's reduction, and the totals agree bit for bit.
        const double o_w = group_allreduce_sum<64>(obj);
        double hh = h1;
        hh += __shfl_xor(hh, 16, 64);
        hh += __shfl_xor(hh, 32, 64);
        __syncthreads();
        double div;
        if (!gnum) {
            // gamma fixed: ONE value crosses the machine per update -- the workgroup's objective (the same triple sum of its wavefronts'
            // parts as below), published in the row of this update's parity (a workgroup that is through the barrier may publish its
            // next value while a slower one still reads this update's: two rows, the barrier keeps them at most one update apart);
            // ONE barrier; then every workgroup sums the row for itself exactly as the reduction below sums statistic G S + G --
            // the same bits everywhere, the same bits as the three-launch path.  No second barrier, no read-back of 521 totals.
            if (lane == 0) red[((size_t)wv * (GP + 2) + GP) * SPAD] = o_w;
            __syncthreads();
            double *const orow = prm.partial + (size_t)(it & 1) * nwg;
            if (tid == 0) {
                double x[NW / 4];
#pragma unroll
                for (int r = 0; r < NW / 4; ++r) {
                    double a = 0.0;
                    for (int k = 0; k < 4; ++k) a += red[((size_t)(4 * r + k) * (GP + 2) + GP) * SPAD];
                    x[r] = a;
                }
                double ps = x[0];
                if constexpr (NW == 12) ps = (x[0] + x[1]) + x[2];
                nm_store(orow + wg, ps);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            NM_STAMP(2);
            alive = nm_grid_barrier(prm.bar, ++epoch, tid, ok_s);
            if (!alive) break;
            NM_STAMP(3);
            if (wv == 0) {
                double a = 0.0;
                if constexpr (NW == 12) {
                    for (int b = lane; b < nwg; b += 64) a += nm_load(orow + b);
                } else {
                    const int ntrip = (nwg + 2) / 3;
                    for (int t = lane; t < ntrip; t += 64) {
                        const int b = 3 * t;
                        double x = nm_load(orow + b);
                        if (b + 1 < nwg) x += nm_load(orow + b + 1);
                        if (b + 2 < nwg) x += nm_load(orow + b + 2);
                        a += x;
                    }
                }
                a = group_allreduce_sum<64>(a);
                if (lane == 0) stat[G * S + G] = a;
            }
            __syncthreads();
            NM_STAMP(6);
            div = stat[G * S + G];
        } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int g = 4 * e + q;
                if (g < GP) red[((size_t)wv * (GP + 2) + g) * SPAD + 16 * t + n] = acc[t][e];
            }
        if (lane == 0) red[((size_t)wv * (GP + 2) + GP) * SPAD] = o_w;
        if (lane < GP) red[((size_t)wv * (GP + 2) + GP + 1) * SPAD + lane] = hh;
        __syncthreads();
        for (int o = tid; o < nout; o += NTHR) {
            // statistic o of the (GP + 2) x SPAD tile: numerator (g, s), row sum g, or the objective
            int row, col;
            if (o < G * S) { row = o / S; col = o % S; }
            else if (o < G * S + G) { row = GP + 1; col = o - G * S; }
            else { row = GP; col = 0; }
            double x[NW / 4];
#pragma unroll
            for (int r = 0; r < NW / 4; ++r) {
                double a = 0.0;
                for (int k = 0; k < 4; ++k) a += red[((size_t)(4 * r + k) * (GP + 2) + row) * SPAD + col];
                x[r] = a;
            }
            double ps = x[0];
            if constexpr (NW == 12) ps = (x[0] + x[1]) + x[2];
            nm_store(prm.partial + (size_t)o * nwg + wg, ps);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NM_STAMP(2);
        alive = nm_grid_barrier(prm.bar, ++epoch, tid, ok_s);
        if (!alive) break;
        NM_STAMP(3);
        // ---- statistic `out` over the workgroups: one wavefront each (the arithmetic of nmft_reduce_body)
        {
            for (int out = wg * NW + wv; out < nout; out += nwg * NW) {          // (small tables: more statistics than wavefronts)
                double a = 0.0;
                const double *row = prm.partial + (size_t)out * nwg;
                if constexpr (NW == 12) {                                        // the rows are triples already
                    for (int b = lane; b < nwg; b += 64) a += nm_load(row + b);
                } else {                                                         // nmft_sum_partials over the workgroup rows
                    const int ntrip = (nwg + 2) / 3;
                    for (int t = lane; t < ntrip; t += 64) {
                        const int b = 3 * t;
                        double x = nm_load(row + b);
                        if (b + 1 < nwg) x += nm_load(row + b + 1);
                        if (b + 2 < nwg) x += nm_load(row + b + 2);
                        a += x;
                    }
                }
                a = group_allreduce_sum<64>(a);
                if (lane == 0) nm_store(prm.stat + out, a);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NM_STAMP(4);
        alive = nm_grid_barrier(prm.bar, ++epoch, tid, ok_s);
        if (!alive) break;
        NM_STAMP(5);
        for (int i = tid; i < nout; i += NTHR) stat[i] = nm_load(prm.stat + i);
        __syncthreads();
        div = stat[G * S + G];
        }
        // ---- the stop test of the factorize loop (Init_NMFT.py:106), by every workgroup for itself
        const bool go = (it < prm.max_iter) && (fabs(prev - div) > prm.min_change);
        if (wg == 0 && tid == 0) {
            if (prm.div_trace) prm.div_trace[it] = div;
            if (!go) { prm.ctl[0] = div; prm.ctl[3] = (double)it; prm.ctl[2] = 1.0; }
        }
        if (!go) break;
        NM_STAMP(6);
        // ---- gamma update (:163-168), every workgroup its own copy
        if (!prm.fix_gamma) {
            double *val = red;                                                   // [G][S] scratch
            for (int i = tid; i < G * S; i += NTHR) {
                const int g = i / S;
                val[i] = (G > 1) ? gm[i] * (nzd(stat[i]) / nzd(stat[G * S + g])) : 1.0;      // :163 / :168
            }
            __syncthreads();
            for (int i = tid; i < G * S; i += NTHR) {
                const int s = i % S;
                double v = val[i];
                if (G > 1) {
                    double tot = 0.0;
                    for (int k = 0; k < G; ++k) tot += val[k * S + s];          // :165
                    v = v / tot;                                                 // :166
                }
                grw[i] = v;                                                      // the tau update of this iteration sees it unclamped
                gm[i] = (adjust && v < DSM_EPS) ? DSM_EPS : v;                   // _adjustment follows the whole div_update (:88-91,:108)
            }
            __syncthreads();
            stage_gamma();
        }
        NM_STAMP(7);
        // ---- tau half of the update on this wavefront's quad (tau rows stay in LDS)
        if (have && prm.fix_gamma) {                                             // the candidate rows are the current ones now
            double *tsw = told; told = tnew; tnew = tsw;
            __builtin_amdgcn_wave_barrier();
        } else if (have) {
            double *tsw = told; told = tnew; tnew = tsw;                         // the rows of the last update are the old ones now
            __builtin_amdgcn_wave_barrier();
            double a_old[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) a_old[kb] = told[n * GP + 4 * kb + q];
            double4_t num = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                double4_t R = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) R = NM_MFMA(a_old[kb], graw_p[(4 * kb + q) * LDG + 16 * t + n], R, 0, 0, 0);
                const double4_t ft = KEEPF ? f[KEEPF ? t : 0] : load_f(t);
                const double4_t qv = nm_div_tile(ft, R);                                // nm_tile_q2: F > 0; lanes without a cell stay finite
                num = nm_num_tile<KB>(num, qv, xq, graw_p + 16 * t, LDG, n, q);
            }
            nm_tau_finish<KB, false>(num, told, tnew, t1, G, n, q, adjust, false, vok, nullptr);
            __builtin_amdgcn_wave_barrier();
        }
        NM_STAMP(8);
        prev = div;
        ++it;
    }
    // ---- the factors leave the chip once: tau rows of this wavefront's quad; gamma by workgroup 0
    if (have) {
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < 16 * G; k += 64) {
            const int vv = k / (4 * G), r = (k / G) & 3, g = k % G;
            if (v0 + vv < V) prm.tau[(size_t)v0 * 4 * G + k] = tnew[(4 * r + vv) * GP + g];
        }
    }
    if (wg == 0) for (int i = tid; i < G * S; i += NTHR) { prm.gam[i] = gm[i]; prm.gam_raw[i] = grw[i]; }
}

// Admission of persistent launches, per device: their workgroups must all be resident, so concurrent ones (several chains of a
// sweep in host threads) may together ask for at most one workgroup per CU -- then every workgroup of every admitted launch has a
// CU it can be placed on whatever the others do; a launch that does not fit next to the running ones waits for them.
struct PersistGate {
    std::mutex mu;
    std::condition_variable cv;
    int used = 0;
    void enter(int want, int cap)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return used == 0 || used + want <= cap; });
        used += want;
    }
    void leave(int want)
    {
        { std::lock_guard<std::mutex> lk(mu); used -= want; }
        cv.notify_all();
    }
};
static PersistGate g_persist_gate[16];

template <int NT, int KB, int NWV>
static int launch_persist(dsm_ctx *c, const NmftPersistParams &q, int grid, size_t sh, int *fits)
{
    auto fn = nmft_persist_kernel<NT, KB, true, NWV>;          // F stays in registers for the whole loop (round 4: the tau numerators on the matrix cores freed the registers; before, three tiles spilled at 3 wavefronts per SIMD and F was re-read from L2 twice per update)
    int occ = 0, cus = 0;
    HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 64 * NWV, sh));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    *fits = grid <= occ * cus;
    if (!*fits) return DSM_OK;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * NWV), sh, c->stream, q);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

// the whole factorize loop as one launch; *used = 0 when this shape / device does not take the persistent path (the caller
// then runs the three-launch loop).  Control words (ctl), trace and factors are left as dsm_nmft_factorize expects them.
// whether the persistent loop takes this context's table, and with what launch: workgroups (0 = it does not), wavefronts per workgroup,
// tiles / K-blocks, dynamic LDS.
static int nmft_persist_shape(const dsm_ctx *c, int fix_gamma, int *nt_out, int *kb_out, int *nwv_out, size_t *sh_out, int *cus_out)
{
    static const bool off = DSM_AB_ENV("DESMAN_HIP_NMFT_NO_PERSIST") != nullptr;
    int nt, kb;
    if (off || c->nmft_persist == 0 || !mfma_shape(c, &nt, &kb) || nt > 6 || kb > 3 || c->timing) return 0;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) return 0;
    const int G = c->nG, S = c->S, nquad = (c->V + 3) / 4, nblk = (nquad + 3) / 4;
    // up to one update-kernel workgroup per CU: four wavefronts per workgroup; above: twelve (three of those workgroups each)
    const int nwv = nblk <= cus ? 4 : NMFT_P_WAVES;
    // 65..96 samples: the four-wavefront form (V <= 16 x compute units; LDS), and only with gamma fixed -- one value per workgroup
    // crosses the machine then (12.8 against 20.7 us per update at 3000 x 96 x 8); with G S + G + 1 statistics to exchange the loop is
    // no faster than three launches there (22.7 vs 22.6)
    if (nt > 4 && (nwv != 4 || !fix_gamma)) return 0;
    const int grid = (nquad + nwv - 1) / nwv;
    const int nout = G * S + G + 1;
    if (grid < 2 || grid > cus) return 0;                   // (neither form holds more than one workgroup per CU worth of table: no buffers
                                                            //  are allocated for tables that cannot take this path)
    const int GP = 4 * kb, SPAD = 16 * nt;
    const size_t sh = (2 * DSM_LOG_TAB_N + 2 * (size_t)GP * (SPAD + 1) + GP + (size_t)nwv * 2 * 16 * GP +
                       (size_t)nwv * std::max<size_t>((size_t)(GP + 2) * SPAD, NM_XQ) + ((nout + 1) & ~1) + 2 * (size_t)G * S + 2) * sizeof(double);
    if (sh > 160 * 1024) return 0;
    *nt_out = nt; *kb_out = kb; *nwv_out = nwv; *sh_out = sh; *cus_out = cus;
    return grid;
}
int k_nmft_persist(dsm_ctx *c, int max_iter, double min_change, int fix_gamma, int adjust, int *used)
{
    *used = 0;
    if (g_batch.K) return DSM_OK;
    int nt, kb, nwv, cus;
    size_t sh;
    const int grid = nmft_persist_shape(c, fix_gamma, &nt, &kb, &nwv, &sh, &cus);
    if (grid == 0) return DSM_OK;
    const int G = c->nG, S = c->S;
    const int nout = G * S + G + 1;
    // exchange buffers + barrier words (zeroed before every launch)
    // ... + a copy of the factors as they are now: should the launch not come to an end (a barrier timed out: its workgroups
    // were not all resident, e.g. behind another process's long-running kernels) the factors are put back and the caller runs
    // the three-launch loop instead
    const size_t n_tau = (size_t)c->V * 4 * G, n_gam = (size_t)G * S;
    const size_t need = (size_t)nout * grid + nout + 16 + n_tau + 2 * n_gam;
    if (!c->np_part || c->np_cap < need) {
        if (c->np_part) { (void)hipFree(c->np_part); c->np_part = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->np_part, need * sizeof(double)));
        c->np_cap = need;
    }
    double *const bak = c->np_part + (size_t)nout * grid + nout + 16;
    HIP_TRY(hipMemcpyAsync(bak, c->ntau, n_tau * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(bak + n_tau, c->ngam, n_gam * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(bak + n_tau + n_gam, c->ngam_raw, n_gam * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (!c->np_bar) HIP_TRY(hipMalloc((void **)&c->np_bar, 1024));
    NmftPersistParams q;
    q.F = c->F; q.tau = c->ntau; q.gam_raw = c->ngam_raw; q.gam = c->ngam;
    q.V = c->V; q.S = S; q.G = G; q.adjust = adjust; q.fix_gamma = fix_gamma; q.max_iter = max_iter; q.min_change = min_change;
    q.ctl = NMFT_CTL(c); q.div_trace = c->ndiv_trace; q.log_tab = c->log_tab;
    q.partial = c->np_part; q.stat = c->np_part + (size_t)nout * grid;
    q.bar.gcnt = c->np_bar; q.bar.top = c->np_bar + 8 * 16; q.bar.gen = c->np_bar + 9 * 16; q.bar.err = c->np_bar + 10 * 16;
    q.stamps = DSM_AB_ENV("DESMAN_HIP_NMFT_STAMPS") ? q.stat + nout : nullptr;       // 8 spare doubles behind the totals
    for (int g = 0; g < 8; ++g) q.bar.members[g] = (grid - g + 7) / 8;
    q.bar.ngroups = std::min(grid, 8);
    if (DSM_AB_ENV("DESMAN_HIP_NMFT_FORCE_TIMEOUT")) q.bar.ngroups += 1;          // test hook: the first barrier never completes
    PersistGate &gate = g_persist_gate[c->device & 15];
    gate.enter(grid, cus);
    struct GateGuard { PersistGate &g; int n; ~GateGuard() { g.leave(n); } } gate_guard{gate, grid};     // held until the workgroups are gone
    HIP_TRY(hipMemsetAsync(c->np_bar, 0, 1024, c->stream));
    int fits = 0, rc = DSM_OK;
    KTimer tm(c, DSM_K_NMFT_B);
#define PCASE(N, K) if (nt == N && kb == K) rc = (nwv == 4) ? launch_persist<N, K, 4>(c, q, grid, sh, &fits) : launch_persist<N, K, NMFT_P_WAVES>(c, q, grid, sh, &fits)
    PCASE(1, 1); PCASE(1, 2); PCASE(1, 3); PCASE(2, 1); PCASE(2, 2); PCASE(2, 3); PCASE(3, 1); PCASE(3, 2); PCASE(3, 3); PCASE(4, 1); PCASE(4, 2); PCASE(4, 3);
#define PCASE4(N, K) if (nt == N && kb == K) rc = launch_persist<N, K, 4>(c, q, grid, sh, &fits)
    PCASE4(5, 1); PCASE4(5, 2); PCASE4(5, 3); PCASE4(6, 1); PCASE4(6, 2); PCASE4(6, 3);
#undef PCASE4
#undef PCASE
    if (rc != DSM_OK) return rc;
    if (!fits) return DSM_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));           // the lock is held until the resident workgroups are gone
    unsigned err = 0;
    HIP_TRY(hipMemcpy(&err, c->np_bar + 10 * 16, sizeof err, hipMemcpyDeviceToHost));
    if (q.stamps) {
        double st[9] = {0};
        HIP_TRY(hipMemcpy(st, q.stamps, sizeof st, hipMemcpyDeviceToHost));
        fprintf(stderr, "nmft_persist phases of update 5, workgroup 0 (us): stats %.2f | wg-reduce+publish %.2f | barrier1 %.2f | reduce %.2f | barrier2 %.2f | read+control %.2f | gamma+stage %.2f | tau half %.2f\n",
                (st[1] - st[0]) / 100.0, (st[2] - st[1]) / 100.0, (st[3] - st[2]) / 100.0, (st[4] - st[3]) / 100.0, (st[5] - st[4]) / 100.0,
                (st[6] - st[5]) / 100.0, (st[7] - st[6]) / 100.0, (st[8] - st[7]) / 100.0);
    }
    if (err) {
        static bool told = false;
        if (!told) { fprintf(stderr, "desman_hip: the persistent NMF launch timed out at a grid barrier (its workgroups were not all resident); "
                                     "falling back to the three-launch loop\n"); told = true; }
        HIP_TRY(hipMemcpyAsync(c->ntau, bak, n_tau * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->ngam, bak + n_tau, n_gam * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->ngam_raw, bak + n_tau + n_gam, n_gam * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemsetAsync(NMFT_CTL(c), 0, 16 * sizeof(double), c->stream));
        return DSM_OK;                                  // *used stays 0: the caller's loop takes over from the untouched start
    }
    *used = 1;
    return DSM_OK;
}

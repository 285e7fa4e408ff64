// kernels_stats.hip -- A2, the auxiliary-count sums (HaploSNP_Sampler.py:284-309 via :266,:276), "spec v2":
// the aggregated sampler restated in oracle/stats_agg.c (the law and its derivation are in that header).
//
//   stats_agg_kernel     stage 1: one wavefront per (variant, 64 samples); lane = sample, the 16 B count slab
//                        of (v,s) is one coalesced int4 load (the tau kernel's layout: no second copy of the
//                        tensor).  tau_v is wave-uniform, so the haplotype sets H_a(v) and the branches of the
//                        Gamma_a accumulation are scalar.  Per lane: Multinomial(x_b; eta[a,b] Gamma_a) for the
//                        four observed bases (dsm_binom.h: mult4) -> Esum (lane-private LDS columns) and the
//                        subset counts N[H_a(v)][s] (one coalesced row of global atomics per true base).
//                        Cost per cell ~ O(G + errors), independent of the read depth.
//   stats_stage2_kernel  stage 2: one workgroup per sample spreads N[.][s] over the haplotypes by recursive
//                        halving of the haplotype range (one large-count binomial per (node, subset): BTRS).
//   stats v1 (per-read draws, kernels_gibbs.hip: stats_kernel) remains for G > 16 / tables above 64 MB.
#include "dsm_binom.h"
#include "dsm_host.h"
#include "log_table.h"

#include <algorithm>
#include <vector>

struct StatsAggParams {
    const int32_t *cnt_vs;
    const uint64_t *tau;
    const double *gamma, *eta;
    int V, S, G;
    uint32_t k0, k1, iter;
    uint32_t *ntab;                 // [2^G][S]
    unsigned long long *esum;       // [16]
};

__device__ __forceinline__ uint64_t wave_uniform_u64(uint64_t x)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void stats_agg_kernel(StatsAggParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_s[];
    const int S = p.S, G = p.G, V = p.V;
    const int NCH = (S + 63) >> 6, SP = NCH << 6;
    double *gT = reinterpret_cast<double *>(smem_s);                 // [G][SP] gamma transposed
    double *rcp = gT + (size_t)G * SP;                                // [64]   1/k
    double *es = rcp + DSM_RCP_TAB_N;                                 // [16]   eta
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(es + 16);   // [16]
    uint32_t *eacc = reinterpret_cast<uint32_t *>(acc + 16);          // [16][256] lane-private Esum columns
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < G * SP; i += 256) {
        const int g = i / SP, s = i - g * SP;
        gT[i] = (s < S) ? p.gamma[(size_t)s * G + g] : 0.0;
    }
    if (tid < DSM_RCP_TAB_N) rcp[tid] = tid ? 1.0 / (double)tid : 0.0;
    if (tid < 16) { es[tid] = p.eta[tid]; acc[tid] = 0ull; }
#pragma unroll
    for (int i = 0; i < 16; ++i) eacc[i * 256 + tid] = 0u;
    __syncthreads();

    const int nwaves = gridDim.x * 4;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (tid >> 6));
    const int ntask = V * NCH;
    for (int task = wid; task < ntask; task += nwaves) {
        const int v = task / NCH, j = task - v * NCH;
        const int s = (j << 6) + lane;
        const bool active = s < S;
        const uint64_t t = wave_uniform_u64(p.tau[v]);
        int4 c = make_int4(0, 0, 0, 0);
        if (active) c = reinterpret_cast<const int4 *>(p.cnt_vs)[(size_t)v * S + s];
        // haplotype sets of the four bases and the abundance each base carries in this sample; t is
        // wave-uniform, so the base of haplotype g and the branches below are scalar
        uint32_t H0 = 0, H1 = 0, H2 = 0, H3 = 0;
        double G0 = 0.0, G1 = 0.0, G2 = 0.0, G3 = 0.0;
        const double *gcol = gT + s;
        for (int g = 0; g < G; ++g) {
            const int a = (int)((t >> (2 * g)) & 3);
            const double x = gcol[g * SP];
            const uint32_t bit = 1u << g;
            if (a == 0) { H0 |= bit; G0 = G0 + x; }
            else if (a == 1) { H1 |= bit; G1 = G1 + x; }
            else if (a == 2) { H2 |= bit; G2 = G2 + x; }
            else { H3 |= bit; G3 = G3 + x; }
        }
        const double Gam[4] = {G0, G1, G2, G3};
        Xo128 rng = xo_seed((uint32_t)s * (uint32_t)V + (uint32_t)v, 0u, p.iter, DSM_STREAM_STA1, p.k0, p.k1);
        uint32_t nacc[4] = {0, 0, 0, 0};
#pragma unroll 1
        for (int b = 0; b < 4; ++b) {
            const int xb = (b == 0) ? c.x : (b == 1) ? c.y : (b == 2) ? c.z : c.w;
            if (xb > 0) {
                double W[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) W[a] = es[a * 4 + b] * Gam[a];
                const double Wt = ((W[0] + W[1]) + W[2]) + W[3];
                if (!(Wt > 0.0)) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) W[a] = Gam[a];
                }
                uint32_t n[4];
                mult4(rng, (uint32_t)xb, W, n, rcp);
                uint32_t *erow = eacc + (b * 4) * 256 + tid;
#pragma unroll
                for (int a = 0; a < 4; ++a) { erow[a * 256] += n[a]; nacc[a] += n[a]; }
            }
        }
        // N[H_a(v)][s] += reads whose true base is a: adjacent lanes -> adjacent words of one table row
        if (nacc[0]) atomicAdd(p.ntab + (size_t)H0 * S + s, nacc[0]);
        if (nacc[1]) atomicAdd(p.ntab + (size_t)H1 * S + s, nacc[1]);
        if (nacc[2]) atomicAdd(p.ntab + (size_t)H2 * S + s, nacc[2]);
        if (nacc[3]) atomicAdd(p.ntab + (size_t)H3 * S + s, nacc[3]);
    }
    // Esum: lane-private columns -> one transposing butterfly per wavefront -> one global atomic per workgroup and counter
    {
        uint32_t e[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i] = eacc[i * 256 + tid];
        const uint32_t tot = wave_transpose_reduce<16>(e);
        const int idx = transpose_index<16>(lane);
        if (lane < 16 && tot) atomicAdd(&acc[idx], (unsigned long long)tot);
    }
    __syncthreads();
    if (tid < 16 && acc[tid]) atomicAdd(&p.esum[tid], acc[tid]);
}

// ---------------------------------------------------------------------------------------------------
// stage 2: one workgroup per sample.  Node (level, idx) owns the haplotype range [lo,hi) and a table of
// 2^(hi-lo) subset counts; its lower child gets floor(w/2) haplotypes.  A subset that lies in one half
// passes through; one that straddles is split by one binomial with odds (sum gamma lower : sum gamma upper),
// drawn from the stream Philox({subset, s | idx << 16 | level << 24, iter, 'STA2'}).  Leaves (w = 1) are
// sum_mu[s][lo].  The level-0 table is read from (and zeroed in) HBM, deeper levels live in LDS.
// ---------------------------------------------------------------------------------------------------
#define S2_MAX_NODES 32
struct Stage2Params {
    uint32_t *ntab;                 // [2^G][S], zeroed after reading
    const double *gamma;            // [S][G]
    unsigned long long *sum_mu;     // [S][G] accumulated into
    const double *log_tab;
    int S, G;
    uint32_t k0, k1, iter;
};

__device__ void stage2_sample(const Stage2Params &p, int s, char *smem)
{
    const int G = p.G, S = p.S, tid = threadIdx.x;
    double2 *ltab = reinterpret_cast<double2 *>(smem);                         // [256]
    double *rcp = reinterpret_cast<double *>(ltab + DSM_LOG_TAB_N);            // [64]
    double *gs = rcp + DSM_RCP_TAB_N;                                          // [32]
    uint32_t *tabA = reinterpret_cast<uint32_t *>(gs + 32);                    // [512] level tables, ping
    uint32_t *tabB = tabA + 512;                                               // [512] pong
    int *nlo = reinterpret_cast<int *>(tabB + 512);                            // node arrays of the current level [S2_MAX_NODES]
    int *nhi = nlo + S2_MAX_NODES, *noff = nhi + S2_MAX_NODES, *nidx = noff + S2_MAX_NODES;
    int *mlo = nidx + S2_MAX_NODES, *mhi = mlo + S2_MAX_NODES, *moff = mhi + S2_MAX_NODES, *midx = moff + S2_MAX_NODES;
    int *ncount = midx + S2_MAX_NODES;                                         // [2] nodes at the current / next level

    ltab[tid] = reinterpret_cast<const double2 *>(p.log_tab)[tid];
    if (tid < DSM_RCP_TAB_N) rcp[tid] = tid ? 1.0 / (double)tid : 0.0;
    if (tid < 32) gs[tid] = (tid < G) ? p.gamma[(size_t)s * G + tid] : 0.0;
    for (int i = tid; i < 1024; i += 256) tabA[i] = 0u;
    if (tid == 0) { nlo[0] = 0; nhi[0] = G; noff[0] = 0; nidx[0] = 0; ncount[0] = 1; }
    __syncthreads();

    uint32_t *cur = tabA, *nxt = tabB;      // level >= 1 tables (level 0 is in HBM)
    for (int level = 0;; ++level) {
        const int nn = ncount[0];
        if (nn == 0) break;
        // thread 0 lays out the next level
        if (tid == 0) {
            int m = 0, off = 0;
            for (int i = 0; i < nn; ++i) {
                const int lo = nlo[i], hi = nhi[i], w = hi - lo;
                if (w == 1) continue;
                const int wl = w / 2, mid = lo + wl;
                mlo[m] = lo; mhi[m] = mid; moff[m] = off; midx[m] = 2 * nidx[i]; off += 1 << wl; ++m;
                mlo[m] = mid; mhi[m] = hi; moff[m] = off; midx[m] = 2 * nidx[i] + 1; off += 1 << (w - wl); ++m;
            }
            ncount[1] = m;
        }
        __syncthreads();
        int child = 0;
        for (int i = 0; i < nn; ++i) {
            const int lo = nlo[i], hi = nhi[i], w = hi - lo;
            const uint32_t *T = cur + noff[i];
            if (w == 1) {
                if (tid == 0) {
                    const uint32_t cnt = (level == 0) ? p.ntab[(size_t)1 * S + s] : T[1];
                    if (level == 0) p.ntab[(size_t)1 * S + s] = 0u;
                    if (cnt) p.sum_mu[(size_t)s * G + lo] += cnt;
                }
                continue;
            }
            const int wl = w / 2, wh = w - wl, mid = lo + wl;
            uint32_t *L = nxt + moff[child], *R = nxt + moff[child + 1];
            child += 2;
            for (uint32_t Hs = 1u + tid; Hs < (1u << w); Hs += 256) {
                uint32_t n;
                if (level == 0) {
                    uint32_t *cell = p.ntab + (size_t)Hs * S + s;
                    n = *cell;
                    if (n) *cell = 0u;
                } else n = T[Hs];
                if (!n) continue;
                const uint32_t HL = Hs & ((1u << wl) - 1u), HR = Hs >> wl;
                if (!HR) { atomicAdd(&L[HL], n); continue; }
                if (!HL) { atomicAdd(&R[HR], n); continue; }
                double wL = 0.0, wR = 0.0;
                for (int jj = 0; jj < wl; ++jj) if ((HL >> jj) & 1u) wL = wL + gs[lo + jj];
                for (int jj = 0; jj < wh; ++jj) if ((HR >> jj) & 1u) wR = wR + gs[mid + jj];
                Xo128 rng = xo_seed(Hs, (uint32_t)s | ((uint32_t)nidx[i] << 16) | ((uint32_t)level << 24), p.iter,
                                    DSM_STREAM_STA2, p.k0, p.k1);
                const uint32_t k = binom_big(rng, n, wL, wR, rcp, ltab);
                if (k) atomicAdd(&L[HL], k);
                if (n - k) atomicAdd(&R[HR], n - k);
            }
        }
        __syncthreads();
        // next level becomes current: metadata, tables (the old current table is cleared for re-use)
        const int m = ncount[1];
        if (tid < m) { nlo[tid] = mlo[tid]; nhi[tid] = mhi[tid]; noff[tid] = moff[tid]; nidx[tid] = midx[tid]; }
        if (tid == 0) ncount[0] = m;
        for (int i = tid; i < 512; i += 256) cur[i] = 0u;
        uint32_t *tswap = cur; cur = nxt; nxt = tswap;
        __syncthreads();
    }
}

#define S2_SMEM_BYTES (DSM_LOG_TAB_N * 16 + DSM_RCP_TAB_N * 8 + 32 * 8 + 1024 * 4 + (8 * S2_MAX_NODES + 2) * 4)

__global__ __launch_bounds__(256) void stats_stage2_kernel(Stage2Params p)
{
    __shared__ __attribute__((aligned(16))) char smem2[S2_SMEM_BYTES];
    stage2_sample(p, blockIdx.x, smem2);
}

// test hook: variate i of a sampler from the stream Philox({i, 0, 0, 'TEST'})  (oracle: orc_binom_test / orc_mult4_test)
__global__ __launch_bounds__(256) void binom_test_kernel(int kind, uint32_t n, double wa, double wb, double w2, double w3,
                                                         uint32_t k0, uint32_t k1, int nsamp, const double *log_tab,
                                                         uint32_t *out)
{
    __shared__ double2 ltab[DSM_LOG_TAB_N];
    __shared__ double rcp[DSM_RCP_TAB_N];
    const int tid = threadIdx.x;
    ltab[tid] = reinterpret_cast<const double2 *>(log_tab)[tid];
    if (tid < DSM_RCP_TAB_N) rcp[tid] = tid ? 1.0 / (double)tid : 0.0;
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    if (i >= nsamp) return;
    Xo128 rng = xo_seed((uint32_t)i, 0u, 0u, DSM_STREAM_TEST, k0, k1);
    if (kind == 0) out[i] = binom_small(rng, n, wa, wb, rcp);
    else if (kind == 1) out[i] = binom_big(rng, n, wa, wb, rcp, ltab);
    else {
        const double W[4] = {wa, wb, w2, w3};
        uint32_t m[4];
        mult4(rng, n, W, m, rcp);
        out[i * 4 + 0] = m[0]; out[i * 4 + 1] = m[1]; out[i * 4 + 2] = m[2]; out[i * 4 + 3] = m[3];
    }
}

// =====================================================================
// host side
// =====================================================================
// spec v2 applies when the subset table fits: G <= 16, 2^G * S * 4 B <= 64 MB, every sample's depth < 2^32
int stats_spec(const dsm_ctx *c)
{
    if (c->force_stats_v1) return 1;
    if (c->G < 1 || c->G > 16) return 1;
    if (((size_t)1 << c->G) * (size_t)c->S * 4 > ((size_t)64 << 20)) return 1;
    if (c->max_depth >= ((uint64_t)1 << 32)) return 1;
    return 2;
}

static int ensure_ntab(dsm_ctx *c)
{
    const size_t need = ((size_t)1 << c->G) * (size_t)c->S;
    if (c->ntab && c->ntab_len == need) return DSM_OK;
    if (c->ntab) { (void)hipFree(c->ntab); c->ntab = nullptr; }
    hipError_t e = hipMalloc((void **)&c->ntab, need * sizeof(uint32_t));
    if (e != hipSuccess) { dsm_set_error("hipMalloc(%zu B) failed: %s", need * 4, hipGetErrorString(e)); return DSM_ERR_NOMEM; }
    c->ntab_len = need;
    HIP_TRY(hipMemsetAsync(c->ntab, 0, need * sizeof(uint32_t), c->stream));
    return DSM_OK;
}

int k_stats_stage1(dsm_ctx *c, uint32_t iter)
{
    int r = ensure_ntab(c);
    if (r != DSM_OK) return r;
    KTimer tm(c, DSM_K_STATS);
    const int S = c->S, G = c->G, V = c->V;
    const int NCH = (S + 63) / 64, SP = NCH * 64;
    const size_t sh = ((size_t)G * SP + DSM_RCP_TAB_N + 16 + 16) * sizeof(double) + 16 * 256 * sizeof(uint32_t);
    if (sh > 160 * 1024) { dsm_set_error("stats_agg: gamma tile (%zu B) exceeds LDS", sh); return DSM_ERR_UNSUPPORTED; }
    if (c->stats_grid == 0) {
        int occ = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, stats_agg_kernel, 256, sh));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, c->device));
        c->stats_grid = std::max(1, occ) * prop.multiProcessorCount;
    }
    const long ntask = (long)V * NCH;
    const int grid = (int)std::max<long>(1, std::min<long>((ntask + 3) / 4, c->stats_grid));
    StatsAggParams p;
    p.cnt_vs = c->cnt_vs; p.tau = c->tau; p.gamma = c->gamma; p.eta = c->eta;
    p.V = V; p.S = S; p.G = G;
    p.k0 = (uint32_t)c->ctr_seed; p.k1 = (uint32_t)(c->ctr_seed >> 32); p.iter = iter;
    p.ntab = c->ntab; p.esum = c->esum;
    hipLaunchKernelGGL(stats_agg_kernel, dim3(grid), dim3(256), sh, c->stream, p);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

int k_stats_stage2(dsm_ctx *c, uint32_t iter)
{
    KTimer tm(c, DSM_K_STATS2);
    Stage2Params p;
    p.ntab = c->ntab; p.gamma = c->gamma; p.sum_mu = c->sum_mu; p.log_tab = c->log_tab;
    p.S = c->S; p.G = c->G;
    p.k0 = (uint32_t)c->ctr_seed; p.k1 = (uint32_t)(c->ctr_seed >> 32); p.iter = iter;
    hipLaunchKernelGGL(stats_stage2_kernel, dim3(c->S), dim3(256), 0, c->stream, p);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

// A2 for the resident state: spec v2 where it applies, else the per-read pass (spec v1)
int k_stats(dsm_ctx *c, uint32_t iter)
{
    if (stats_spec(c) == 2) {
        int r = k_stats_stage1(c, iter);
        if (r != DSM_OK) return r;
        return k_stats_stage2(c, iter);
    }
    return k_stats_v1(c, iter);
}

int k_binom_test(dsm_ctx *c, int kind, uint32_t n, const double *w, uint64_t seed, int nsamp, uint32_t *d_out)
{
    hipLaunchKernelGGL(binom_test_kernel, dim3((nsamp + 255) / 256), dim3(256), 0, c->stream, kind, n, w[0], w[1], w[2], w[3],
                       (uint32_t)seed, (uint32_t)(seed >> 32), nsamp, c->log_tab, d_out);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

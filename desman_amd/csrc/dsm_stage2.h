// dsm_stage2.h -- stage 2 of the aggregated mu/E pass (spec: oracle/stats_agg.c, stage2_sample): one workgroup
// per sample spreads the subset counts N[.][s] over the haplotypes by halving the haplotype range recursively.
// Node (level, idx) owns the range [lo,hi) and a table of 2^(hi-lo) subset counts; its lower child gets
// floor(w/2) haplotypes.  A subset that lies in one half passes through; one that straddles is split by one
// binomial with odds (sum gamma lower : sum gamma upper) from the stream
// Philox({subset, s | idx << 16 | level << 24, iter, 'STA2'}).  Leaves (w = 1) are sum_mu[s][lo].
// The tree depends on G only: the host lays it out (S2Plan, kernel argument -> scalar registers), every node
// below the root has its own zero-initialised LDS table, so a level costs one workgroup barrier.
// Shared by stats_stage2_kernel (kernels_stats.hip) and dirichlet_kernel (kernels_gibbs.hip, fused form).
#pragma once
#include "dsm_binom.h"
#include "log_table.h"

#define S2_MAX_NODES 32
#define S2_MAX_LEVELS 6
#define S2_TAB_ENTRIES 1024
struct S2Plan {
    int nlevels, tab_entries;
    int level_start[S2_MAX_LEVELS + 2];
    signed char lo[S2_MAX_NODES], hi[S2_MAX_NODES], child[S2_MAX_NODES];
    unsigned char idx[S2_MAX_NODES];
    short off[S2_MAX_NODES];
};

struct Stage2Params {
    uint32_t *ntab;                 // [rep][2^G][S], zeroed after reading
    int rep, ld;                    // copies of the table (their sum is the count); row stride in words (>= S)
    const double *gamma;            // [S][G]
    unsigned long long *sum_mu;     // [S][G] accumulated into (stand-alone kernel)
    const double *log_tab;
    int S, G;
    uint32_t k0, k1, iter;
    uint32_t hmul, swz;             // row of (subset H, sample s) in ntab: (H * hmul + (s >> 4) * swz) mod 2^G (kernels_stats.hip: stats_ntab_hmul / _swz)
    uint32_t *big_count;            // work-list counter of stage 1: consumed by now, reset here for the next pass (or null)
    int nsplit;                     // workgroups that share a sample's root level (stand-alone kernel, G >= 11), else 1
    uint32_t *scratch, *ticket;     // [S][S2_TAB_ENTRIES] level-1 tables handed over, [S] arrival counters; zero between passes
};                                  // the plan of the halving tree (a function of G) travels beside it: one per launch

S2Plan make_stage2_plan(int G);     // kernels_stats.hip

#define S2_SMEM_BYTES (DSM_LOG_TAB_N * 16 + DSM_EXP_TAB_N * 8 + DSM_RCP_TAB_N * 8 + 32 * 8 + S2_TAB_ENTRIES * 4 + 32 * 4 + 5 * S2_MAX_NODES * 4)

#ifdef DSM_AB_SWITCHES           // phase stamps of the Dirichlet launch's gamma rows (experiment build): [row][entry, tables staged, level 0, 1, 2, 3, stage 2 done, draw done]
__device__ unsigned long long s2_clk[1024 * 8];
#define S2_CLK(row, k) do { if (threadIdx.x == 0 && blockIdx.y == 0 && (row) < 1024) s2_clk[(size_t)(row) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define S2_CLK(row, k) do { } while (0)
#endif
// leaf counts end in LDS (returned pointer, [G] u32, valid after the function's final barrier); to_global also adds
// them to p.sum_mu[s][.]
template <int SPEC>
__device__ __forceinline__ const uint32_t *stage2_sample(const Stage2Params &p, const S2Plan &pl, int s, char *smem, bool to_global, int part = 0)
{
    const int G = p.G, tid = threadIdx.x, nthr = blockDim.x;      // 256 (fused form) or 1024 (many subsets)
    double2 *ltab = reinterpret_cast<double2 *>(smem);                         // [256] log table, then [64] exp table
    double *etab = reinterpret_cast<double *>(ltab + DSM_LOG_TAB_N);
    double *rcp = etab + DSM_EXP_TAB_N;                                        // [256]
    double *gs = rcp + DSM_RCP_TAB_N;                                          // [32]
    uint32_t *tab = reinterpret_cast<uint32_t *>(gs + 32);                     // [S2_TAB_ENTRIES] tables of all nodes below the root
    uint32_t *leaf = tab + S2_TAB_ENTRIES;                                     // [32]
    int *n_lo = reinterpret_cast<int *>(leaf + 32);                            // the plan's node arrays (lane-indexed below)
    int *n_hi = n_lo + S2_MAX_NODES, *n_off = n_hi + S2_MAX_NODES, *n_idx = n_off + S2_MAX_NODES, *n_child = n_idx + S2_MAX_NODES;

    const int nsplit = p.nsplit > 1 ? p.nsplit : 1;
    S2_CLK(s, 0);
    // Round 6: a thread's first root-level count is on its way while the tables are staged (the load has nothing to do with them: 0.9 us of
    // the launch's ~21 us critical path, phase stamps of the experiment build: profiles/r06_s2_clocks.txt).  One copy of the table only.
    uint32_t pre_n = 0;
    bool pre_ok = false;
    {
        const int total0 = 1 << G, share0 = nsplit > 1 ? (total0 + nsplit - 1) / nsplit : total0;
        const int j0 = (nsplit > 1 ? part * share0 : 0) + tid, jh0 = nsplit > 1 ? min(total0, part * share0 + share0) : total0;
        if (p.rep == 1 && j0 < jh0 && j0 != 0) {
            pre_n = p.ntab[(size_t)(((uint32_t)j0 * p.hmul + ((uint32_t)s >> 4) * p.swz) & ((1u << G) - 1u)) * (size_t)p.ld + s];
            pre_ok = true;
        }
    }
    if (s == 0 && part == 0 && tid < DSM_BIG_NT * DSM_BIG_NL && p.big_count) p.big_count[tid * DSM_BIG_STRIDE] = 0u;
    if (tid < DSM_LOG_TAB_N) ltab[tid] = reinterpret_cast<const double2 *>(p.log_tab)[tid];
    if (tid < DSM_EXP_TAB_N) etab[tid] = p.log_tab[2 * DSM_LOG_TAB_N + tid];
    for (int k = tid; k < DSM_RCP_TAB_N; k += nthr) rcp[k] = k ? 1.0 / (double)k : 0.0;
    if (tid < 32) {
        gs[tid] = (tid < G) ? p.gamma[(size_t)s * G + tid] : 0.0; leaf[tid] = 0u;
        n_lo[tid] = pl.lo[tid]; n_hi[tid] = pl.hi[tid]; n_off[tid] = pl.off[tid]; n_idx[tid] = pl.idx[tid];
        n_child[tid] = pl.child[tid];
    }
    for (int i = tid; i < S2_TAB_ENTRIES; i += nthr) tab[i] = 0u;
    __syncthreads();
    S2_CLK(s, 1);

    // Round 6: a node's count goes straight into leaf[] where its child is a single haplotype, so the tree's last level -- leaves only: a pass
    // over the level and a barrier to copy table entries, 1.0 us of the launch -- is not walked (integer adds: the same sums)
    const int nlev = pl.nlevels > 1 ? pl.nlevels - 1 : pl.nlevels;
    for (int level = 0; level < nlev; ++level) {
        // all (node, subset) pairs of the level at once: the tables of a level are contiguous in `tab`, so entry j of
        // the level belongs to the node whose table covers it (root: the 2^G words of this sample in HBM)
        const int n0 = pl.level_start[level], n1 = pl.level_start[level + 1];
        const int base = pl.off[n0];
        const int total = (level == 0) ? (1 << G) : (pl.off[n1 - 1] + (1 << (pl.hi[n1 - 1] - pl.lo[n1 - 1]))) - base;
        // (the root level of a sample that several workgroups share: this one's contiguous share of the subsets)
        const int share = (level == 0 && nsplit > 1) ? (total + nsplit - 1) / nsplit : total;
        const int j_lo = (level == 0 && nsplit > 1) ? part * share : 0, j_hi = (level == 0 && nsplit > 1) ? min(total, j_lo + share) : total;
        for (int j = j_lo + tid; j < j_hi; j += nthr) {
            int i = n0;
            for (int k = n0 + 1; k < n1; ++k) if (j + base >= pl.off[k]) i = k;      // scalar plan, <= 16 nodes per level
            const int lo = n_lo[i], hi = n_hi[i], w = hi - lo;
            const uint32_t Hs = (uint32_t)(j + base - n_off[i]);
            if (Hs == 0) continue;
            uint32_t n;
            if (level == 0) {
                uint32_t *cell = p.ntab + (size_t)((Hs * p.hmul + ((uint32_t)s >> 4) * p.swz) & ((1u << G) - 1u)) * (size_t)p.ld + s;
                const size_t cstride = ((size_t)1 << G) * (size_t)p.ld;
                if (p.rep == 8) {
                    // one copy per XCD (kernels_stats.hip): all eight loads in flight at once -- read one after the other (a store may
                    // alias the next load, so the compiler keeps the order) they cost eight memory round trips: +3.5 us on the launch
                    uint32_t m[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) m[r] = __builtin_nontemporal_load(cell + r * cstride);
                    n = 0;
#pragma unroll
                    for (int r = 0; r < 8; ++r) { n += m[r]; if (m[r]) cell[r * cstride] = 0u; }
                } else {
                    n = (pre_ok && j == j_lo + tid) ? pre_n : *cell;
                    if (n) *cell = 0u;
                    for (int r = 1; r < p.rep; ++r) {           // few subsets, many positions: the atomics of stage 1 were spread over copies
                        const uint32_t m = cell[r * cstride];
                        if (m) { cell[r * cstride] = 0u; n += m; }
                    }
                }
            } else n = tab[base + j];
            if (w == 1) {                                   // leaf: Hs == 1.  Only the root can be one here (G = 1): every other leaf was added to by its parent
                if (level == 0) leaf[lo] = n;
                continue;
            }
            if (!n) continue;
            const int wl = w / 2, wh = w - wl, mid = lo + wl;
            const int ch = n_child[i];
            uint32_t *L = tab + n_off[ch], *R = tab + n_off[ch + 1];
            const uint32_t HL = Hs & ((1u << wl) - 1u), HR = Hs >> wl;
            if (wl == 1) L = leaf + lo - 1;                  // (HL is 1 wherever it is added to: L[HL] = leaf[lo])
            if (wh == 1) R = leaf + mid - 1;
            if (!HR) { atomicAdd(&L[HL], n); continue; }
            if (!HL) { atomicAdd(&R[HR], n); continue; }
            double wL = 0.0, wR = 0.0;
            for (int jj = 0; jj < wl; ++jj) if ((HL >> jj) & 1u) wL = wL + gs[lo + jj];
            for (int jj = 0; jj < wh; ++jj) if ((HR >> jj) & 1u) wR = wR + gs[mid + jj];
            Xo128 rng = xo_seed(Hs, (uint32_t)s | ((uint32_t)n_idx[i] << 16) | ((uint32_t)level << 24), p.iter, DSM_STREAM_STA2,
                                p.k0, p.k1);
            bool dummy = false;
            const uint32_t k = binom<true, SPEC>(rng, n, wL, wR, rcp, ltab, dummy, DSM_BINV_MEAN_CAP_S2);
            if (k) atomicAdd(&L[HL], k);
            if (n - k) atomicAdd(&R[HR], n - k);
        }
        __syncthreads();
        if (level < 4) S2_CLK(s, 2 + level);
        if (level == 0 && nsplit > 1) {
            // hand the level-1 tables over: device-scope atomics onto the sample's scratch rows (they execute at the memory side, so
            // the reader below sees every one of them), then a ticket; whoever draws the last ticket owns the rest of the tree
            const int l1 = pl.level_start[1], l2 = pl.level_start[2];
            const int b1 = pl.off[l1], n1 = (pl.off[l2 - 1] + (1 << (pl.hi[l2 - 1] - pl.lo[l2 - 1]))) - b1;
            uint32_t *row = p.scratch + (size_t)s * S2_TAB_ENTRIES;
            // (the adds RETURN their old value: a thread passes the barrier below only when its adds have been performed, so the
            // ticket -- issued after the barrier -- is ordered behind every add of this workgroup without a cache-flushing fence)
            uint32_t sink = 0;
            for (int i = tid; i < n1; i += nthr) { const uint32_t v = tab[b1 + i]; if (v) sink += atomicAdd(&row[i], v); }
            asm volatile("" :: "v"(sink));
            __syncthreads();
            __shared__ int last_wg;
            if (tid == 0) last_wg = (atomicAdd(&p.ticket[s], 1u) == (uint32_t)(nsplit - 1)) ? 1 : 0;
            __syncthreads();
            if (!last_wg) return leaf;                         // (callers of the split form use nothing of the returned tables)
            for (int i = tid; i < n1; i += nthr) tab[b1 + i] = atomicExch(&row[i], 0u);      // read at the memory side, left zero for the next pass
            if (tid == 0) p.ticket[s] = 0u;
            __syncthreads();
        }
    }
    if (to_global && tid < G && leaf[tid]) p.sum_mu[(size_t)s * G + tid] += leaf[tid];
    return leaf;
}

// genes.hip -- f4: accessory-gene assignment (desman/Eta_Sampler.py, desman/GeneAssign.py) on gfx950.
//
//   gene_sweep_kernel     Eta_Sampler.sampleTauC (:355-369) = the A1 tau sweep (c_sample_tau.c:95-204)
//                         with the gene's masked, re-normalised gamma (maskGamma :147-157), for many
//                         genes and both candidates of a copy-number step in one launch, plus the
//                         variant log-probability sum x log p of computeVarLLContrib (:204-216)
//   gene_choose_kernel    the copy-number draw of update (:226-262) for every gene, the per-gene
//                         log-likelihood (:182-202) and the MAP bookkeeping (storeStarState :551-555)
//   gene_nmft_kernel      Init_NMFT.factorize_tau (:134-149,192-205) for every gene, one workgroup each
//   kl_*_kernel           GeneAssign.KLAssign.factorize (GeneAssign.py:85-120)
//
// Layout: all genes' variant rows live in ONE [Vtot][S][4] int32 tensor (gene c = rows
// gene_off[c]..gene_off[c+1]); tau is packed 2 bits/haplotype in three [Vtot] u64 buffers: cur[c] names
// the buffer holding gene c's current tau, the two candidates of a step are written to the other two
// and "keeping" a candidate is one integer store.
#include <math.h>
#include <string.h>

#include <vector>

#include "dsm_device.h"
#include "dsm_host.h"
#include "log_table.h"

#define DSM_STREAM_GENE 0x47454E45u   // 'GENE'  tau-sweep uniforms of the batched gene sampler
#define DSM_STREAM_GETA 0x47455441u   // 'GETA'  copy-number draws
#define GENE_VPG 1                    // variants a lane group sweeps per workgroup
#define GENE_MIN_DELTA 1.0e-10        // Eta_Sampler.py:18
#define GENE_ETA_PENALTY (-1.0e3)     // Eta_Sampler.py:19

#define TRY(x) do { int _r = (x); if (_r != DSM_OK) return _r; } while (0)

namespace {

template <typename T>
struct DBuf {                         // device array owned by the host object
    T *p = nullptr;
    size_t n = 0;
    ~DBuf() { if (p) (void)hipFree(p); }
    int resize(size_t count)
    {
        if (count <= n && p) return DSM_OK;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        const size_t bytes = (count ? count : 1) * sizeof(T);
        const hipError_t e = hipMalloc((void **)&p, bytes);
        if (e != hipSuccess) { dsm_set_error("hipMalloc(%zu B) failed: %s", bytes, hipGetErrorString(e)); return DSM_ERR_NOMEM; }
        n = count;
        return DSM_OK;
    }
    operator T *() const { return p; }
    DBuf() = default;
    DBuf(const DBuf &) = delete;
    DBuf &operator=(const DBuf &) = delete;
};

// numpy's add.reduce over a contiguous axis of length n (what gammaR.sum(axis=1) does, maskGamma :154):
// sequential below 8 values, otherwise 8 running sums combined pairwise and a sequential tail.
__host__ __device__ inline double masked_row_sum(const double *a, int n, uint32_t mask)
{
#define AT(i) (((mask >> (i)) & 1u) ? a[(i)] : 0.0)
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += AT(i);
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = AT(j);
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += AT(i + j);
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += AT(i);
    return res;
#undef AT
}

// =====================================================================
// counts int64 -> int32 with range check (counts must be exact as float: < 2^24, c_sample_tau.c:164)
// =====================================================================
__global__ void gene_convert_kernel(const int64_t *__restrict__ in, int32_t *__restrict__ out, size_t n, int *flag)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t x = in[i];
    if (x < 0 || x >= (1 << 24)) *flag = 1;
    out[i] = (int32_t)x;
}

__global__ void gene_pack_kernel(const int64_t *__restrict__ onehot, uint64_t *__restrict__ packed, int V, int G)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    uint64_t t = 0;
    for (int g = 0; g < G; ++g) {
        const int64_t *o = onehot + ((size_t)v * G + g) * 4;
        const int a = o[1] ? 1 : o[2] ? 2 : o[3] ? 3 : 0;
        t |= (uint64_t)a << (2 * g);
    }
    packed[v] = t;
}

// gathers every gene's current tau (buffer cur[c]) into one-hot int64
__global__ void gene_unpack_kernel(const uint64_t *t0, const uint64_t *t1, const uint64_t *t2, const int32_t *cur,
                                   const int32_t *gene_of, int64_t *__restrict__ onehot, int V, int G)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)V * G) return;
    const int v = (int)(i / G), g = (int)(i % G);
    const int b = cur[gene_of[v]];
    const uint64_t t = (b == 0 ? t0 : b == 1 ? t1 : t2)[v];
    const int a = (int)((t >> (2 * g)) & 3);
    int64_t *o = onehot + i * 4;
    o[0] = a == 0; o[1] = a == 1; o[2] = a == 2; o[3] = a == 3;
}

// =====================================================================
// masked tau sweep + variant log-probability
// =====================================================================
struct GeneSweepParams {
    const int32_t *cnt_vs;      // [Vtot][S][4]
    const int32_t *gene_off;    // [C+1]
    const int32_t *task_tab;    // [ntask][2] {gene, first variant row}: <= GENE_VPG rows of one gene
    uint64_t *tau[3];
    const int32_t *cur;         // [C]
    const int32_t *eta;         // [C][G] mask source
    const double *gamma, *eps;  // [S][G], [4][4]
    const double *log_tab;
    const uint32_t *u_raw;      // raw 32-bit words (GSL stream or test vectors); null -> Philox
    const int64_t *u_off;       // [C][2] first word of (gene, candidate) in u_raw
    double *v_ll;               // [2][Vtot] x log p of every variant after the sweep
    int32_t *nchange;           // [C][2]
    int S, G, Vtot, task_base, task_end, step_g;
    int gene_base;              // global index of gene 0 (counter-based draws are keyed by global gene, row in gene)
    uint32_t k0, k1, iter;
};

// One lane group (LPV lanes; lane = sample, NSL samples per lane) per task, candidate blockIdx.y.  Genes are
// small (a handful of variant rows), so the masked gamma is built per GROUP, not per workgroup: every lane
// only ever reads the gamma column of its own samples, so it re-normalises those itself (no barrier) into the
// group's LDS tile.  step_g >= 0: copy-number step of haplotype step_g, candidate k forces mask bit step_g to k
// and writes tau to buffer (cur+1+k)%3; step_g < 0: the gene's own mask, in place.
template <int LPV, int NSL, bool SWEEP>
__global__ __launch_bounds__(256) void gene_sweep_kernel(GeneSweepParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    constexpr int SP = LPV * NSL;
    const int GPB = blockDim.x / LPV;
    double2 *ltab = reinterpret_cast<double2 *>(smem_g);                 // [128]
    double *eS = reinterpret_cast<double *>(ltab + DSM_LOG_TAB_N);       // [16]
    const int tid = threadIdx.x, G = p.G, S = p.S;
    const int grp = tid / LPV, lig = tid % LPV, k = blockIdx.y;
    double *gT = eS + 16 + (size_t)grp * G * SP;                         // [G][SP] of this group
    for (int i = tid; i < DSM_LOG_TAB_N; i += blockDim.x) ltab[i] = reinterpret_cast<const double2 *>(p.log_tab)[i];
    if (tid < 16) eS[tid] = p.eps[tid];
    __syncthreads();
    const int task = p.task_base + blockIdx.x * GPB + grp;
    if (task >= p.task_end) return;
    const int c = p.task_tab[2 * task], vfirst = p.task_tab[2 * task + 1];
    const int g0 = p.gene_off[c], g1 = p.gene_off[c + 1];
    const int vend = (vfirst + GENE_VPG < g1) ? vfirst + GENE_VPG : g1;
    uint32_t mask = 0;
    for (int h = 0; h < G; ++h) mask |= (uint32_t)(p.eta[(size_t)c * G + h] > 0) << h;
    if (p.step_g >= 0) mask = (mask & ~(1u << p.step_g)) | ((uint32_t)k << p.step_g);
    if (mask == 0u) return;                                              // no haplotype carries the gene: nothing to sweep
    const int src = p.cur[c];
    const int dst = (p.step_g >= 0) ? (src + 1 + k) % 3 : src;
    const uint64_t *tsrc = p.tau[src];
    uint64_t *tdst = p.tau[dst];
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
        const int s = lig + j * LPV;
        if (s < S) {
            const double rsum = masked_row_sum(p.gamma + (size_t)s * G, G, mask);
            for (int g = 0; g < G; ++g) gT[g * SP + s] = (((mask >> g) & 1u) ? p.gamma[(size_t)s * G + g] : 0.0) / rsum;
        } else {
            for (int g = 0; g < G; ++g) gT[g * SP + s] = 1.0;            // pad: p > 0, count = 0
        }
    }

    int nchg = 0;
    for (int v = vfirst; v < vend; ++v) {
        uint64_t t = tsrc[v];
        int xi[NSL][4];
        double xf[NSL][4];
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            const int s = lig + j * LPV;
            int4 cc = make_int4(0, 0, 0, 0);
            if (s < S) cc = reinterpret_cast<const int4 *>(p.cnt_vs)[(size_t)v * S + s];
            xi[j][0] = cc.x; xi[j][1] = cc.y; xi[j][2] = cc.z; xi[j][3] = cc.w;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) xf[j][bb] = (double)(float)xi[j][bb];   // c_sample_tau.c:164
        }
        if (SWEEP) {
            uint32_t uw_lane = 0;
            // prefix of the h-ascending rest-mixture chain over the haplotypes already re-drawn (see tau_kernel)
            double pre[NSL][4];
#pragma unroll
            for (int j = 0; j < NSL; ++j)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) pre[j][bb] = 0.0;
            // log-probability of the row's current configuration = the candidate chosen at the previous step that had a
            // choice (a haplotype without the gene changes nothing): that candidate is not evaluated again (tau_kernel)
            bool have_cur = false;
            double l_cur = 0.0;
            for (int g = 0; g < G; ++g) {
                const int told = (int)((t >> (2 * g)) & 3);
                uint32_t uw;
                if (p.u_raw) {
                    uw = p.u_raw[p.u_off[(size_t)c * 2 + k] + (size_t)(v - g0) * G + g];
                } else {
                    // counter-based word of (row, haplotype): lane i of the group draws the word of haplotype
                    // g + i once per LPV haplotypes, every step then reads its word from the owning lane
                    if (g % LPV == 0) {
                        uint32_t r[4] = {0, 0, 0, 0};
                        if (g + lig < G)
                            philox4x32_10((uint32_t)((v - g0) * G + g + lig), (uint32_t)(p.gene_base + c), p.iter,
                                          DSM_STREAM_GENE + (uint32_t)((p.step_g + 1) * 2 + k), p.k0, p.k1, r);
                        uw_lane = r[0];
                    }
                    uw = (uint32_t)__shfl((int)uw_lane, g % LPV, LPV);
                }
                const double u = (double)uw * 2.3283064365386963e-10;
                if (!((mask >> g) & 1u)) {
                    // a haplotype without the gene has gamma = 0: the four candidates are the same mixture, their
                    // log-probabilities are equal bit for bit, so the draw is uniform: ex = {1,1,1,1}, sum = 4
                    const double us = u * 4.0;
                    const int tn = (us < 1.0) ? 0 : (us < 2.0) ? 1 : (us < 3.0) ? 2 : 3;
                    nchg += (lig == 0) & (tn != told);
                    t = (t & ~(3ull << (2 * g))) | ((uint64_t)tn << (2 * g));
                    continue;
                }
                double st[NSL][4];
#pragma unroll
                for (int j = 0; j < NSL; ++j)
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) st[j][bb] = pre[j][bb];
                for (int h = g + 1; h < G; ++h) {        // rest mixture, h ascending (c_sample_tau.c:136-150)
                    if (!((mask >> h) & 1u)) continue;               // gamma = 0 adds exactly nothing
                    const double *er = eS + (int)((t >> (2 * h)) & 3) * 4;
                    const double e0 = er[0], e1 = er[1], e2 = er[2], e3 = er[3];
#pragma unroll
                    for (int j = 0; j < NSL; ++j) {
                        const double gm = gT[h * SP + lig + j * LPV];
                        st[j][0] = fma(e0, gm, st[j][0]);
                        st[j][1] = fma(e1, gm, st[j][1]);
                        st[j][2] = fma(e2, gm, st[j][2]);
                        st[j][3] = fma(e3, gm, st[j][3]);
                    }
                }
                double gg[NSL];
#pragma unroll
                for (int j = 0; j < NSL; ++j) gg[j] = gT[g * SP + lig + j * LPV];
                double l[4];
                const bool reuse = have_cur;
                if constexpr (LPV == 64) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        l[a] = 0.0;
                        if (!(reuse && a == told)) l[a] = sweep_candidate<NSL>(a, xf, st, gg, eS, ltab, lig, LPV, S);
                    }
                    group_allreduce_sum4<LPV>(l[0], l[1], l[2], l[3]);
                } else {
                    // several rows per wavefront, each with its own current base: candidates in the rotated order
                    // told + 1, told + 2, told + 3; the totals come back in base order (dsm_device.h)
                    const int rot = reuse ? told + 1 : 0;
                    double cv[4];
#pragma unroll
                    for (int i = 0; i < 3; ++i) cv[i] = sweep_candidate<NSL>((rot + i) & 3, xf, st, gg, eS, ltab, lig, LPV, S);
                    cv[3] = 0.0;
                    if (!reuse) cv[3] = sweep_candidate<NSL>(3, xf, st, gg, eS, ltab, lig, LPV, S);
                    group_allreduce_sum4_unrotate<LPV>(cv, rot, l);
                }
                if (reuse) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) if (a == told) l[a] = l_cur;
                }
                const int tn = sweep_draw(l, uw);
                l_cur = (tn == 0) ? l[0] : (tn == 1) ? l[1] : (tn == 2) ? l[2] : l[3];
                have_cur = true;
                nchg += (lig == 0) & (tn != told);
                t = (t & ~(3ull << (2 * g))) | ((uint64_t)tn << (2 * g));
                {                                                   // link g of the chain, with the new base
                    const double *er = eS + tn * 4;
                    const double e0 = er[0], e1 = er[1], e2 = er[2], e3 = er[3];
#pragma unroll
                    for (int j = 0; j < NSL; ++j) {
                        pre[j][0] = fma(e0, gg[j], pre[j][0]);
                        pre[j][1] = fma(e1, gg[j], pre[j][1]);
                        pre[j][2] = fma(e2, gg[j], pre[j][2]);
                        pre[j][3] = fma(e3, gg[j], pre[j][3]);
                    }
                }
            }
            if (lig == 0) tdst[v] = t;
        }
        // sum x log p of the (new) configuration: computeVarLLContrib :213-215, logLikelihoodGene :166-173
        double vacc = 0.0;
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            double P[4] = {0.0, 0.0, 0.0, 0.0};
            for (int g = 0; g < G; ++g) {
                if (!((mask >> g) & 1u)) continue;
                const double *er = eS + (int)((t >> (2 * g)) & 3) * 4;
                const double gm = gT[g * SP + lig + j * LPV];
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) P[bb] = fma(gm, er[bb], P[bb]);
            }
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) vacc = fma((double)xi[j][bb], dsm_log(P[bb], ltab), vacc);
        }
        const double tot = group_allreduce_sum<LPV>(vacc);
        if (lig == 0) p.v_ll[(size_t)k * p.Vtot + v] = tot;
    }
    if (SWEEP && lig == 0 && nchg) atomicAdd(&p.nchange[c * 2 + k], nchg);
}

// =====================================================================
// copy-number step for every gene (one wavefront per gene)
// =====================================================================
struct GeneChooseParams {
    const int32_t *gene_off;             // [C+1]
    int32_t *eta, *cur;
    const double *cov, *delta;           // [C][S], [G][S]
    const double *prior;                 // [max_eta]
    const double *cov_const, *mult_const;
    const double *v_ll;                  // [2][Vtot] per-variant x log p of the two candidates
    double *lv_keep;                     // [C] x log p of the kept configuration
    double *gene_ll, *gene_llstar;
    int32_t *eta_star;
    int32_t *eta_store;                  // slot of this iteration or null
    double *gene_ll_trace;               // slot of this iteration or null
    const double *u_ext;                 // [C][G] uniforms of this iteration or null
    int C, S, G, max_eta, Vtot, gene_base;
    int step_g;                          // >= 0: draw eta[., step_g]; < 0: evaluate only (x log p in partial slot 0)
    int finish;                          // compute gene_ll (+ MAP bookkeeping, stores)
    int reset_star;                      // star := this state
    uint32_t k0, k1, iter;
};

// sum_s ( -ce + cov log ce ), ce = max(ce, MIN_DELTA)  (log_Poisson :34-39 without the lgamma constant)
__device__ __forceinline__ double poisson_term(double cov, double ce) { return cov * log(ce) - ce; }

__global__ __launch_bounds__(256) void gene_choose_kernel(GeneChooseParams p)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= p.C) return;
    const int G = p.G, S = p.S, g = p.step_g;
    const int Vc = p.gene_off[c + 1] - p.gene_off[c];
    int32_t *eta = p.eta + (size_t)c * G;
    // x log p of the candidates: the gene's per-variant values, lane-strided then a fixed butterfly
    // (the order depends only on the gene's size -> deterministic)
    double lv[2] = {0.0, 0.0};
    for (int k = 0; k < (g >= 0 ? 2 : 1); ++k) {
        double a = 0.0;
        for (int v = p.gene_off[c] + lane; v < p.gene_off[c + 1]; v += 64) a += p.v_ll[(size_t)k * p.Vtot + v];
        lv[k] = group_allreduce_sum<64>(a);
    }
    double kept = (g >= 0) ? 0.0 : lv[0];
    int pick = 0;
    if (g >= 0) {
        bool any0 = false;
        for (int h = 0; h < G; ++h) any0 |= (h != g) && (eta[h] > 0);
        // coverage terms of the max_eta states (update :246-259); the clamp of state 0 is applied in
        // place upstream, so states s >= 1 start from the clamped base (log_Poisson mutates its argument)
        double acc[DSM_MAX_ETA];
#pragma unroll
        for (int s = 0; s < DSM_MAX_ETA; ++s) acc[s] = 0.0;
        for (int i = lane; i < S; i += 64) {
            double base = 0.0;
            for (int h = 0; h < G; ++h)
                if (h != g) base = fma((double)eta[h], p.delta[(size_t)h * S + i], base);
            if (base < GENE_MIN_DELTA) base = GENE_MIN_DELTA;
            const double cv = p.cov[(size_t)c * S + i], dg = p.delta[(size_t)g * S + i];
            acc[0] += poisson_term(cv, base);
            for (int s = 1; s < p.max_eta; ++s) {
                double ce = base + (double)s * dg;
                if (ce < GENE_MIN_DELTA) ce = GENE_MIN_DELTA;
                acc[s] += poisson_term(cv, ce);
            }
        }
        double lp[DSM_MAX_ETA];
        const double lv0 = (Vc > 0) ? (any0 ? lv[0] : -1.0e20) : 0.0;
        const double lv1 = (Vc > 0) ? lv[1] : 0.0;
        double mx = -INFINITY;
        for (int s = 0; s < p.max_eta; ++s) {
            const double tot = group_allreduce_sum<64>(acc[s]);
            lp[s] = p.prior[s] + ((p.cov_const[c] + tot) + (s == 0 ? lv0 : lv1));
            if (lp[s] > mx) mx = lp[s];
        }
        // sampleLogProb :349-352 as an inverse-CDF draw
        double ex[DSM_MAX_ETA], sum = 0.0;
        for (int s = 0; s < p.max_eta; ++s) { ex[s] = exp(lp[s] - mx); sum += ex[s]; }
        double u;
        if (p.u_ext) u = p.u_ext[(size_t)c * G + g];
        else {
            uint32_t r[4];
            philox4x32_10((uint32_t)(p.gene_base + c), (uint32_t)g, p.iter, DSM_STREAM_GETA, p.k0, p.k1, r);
            u = u01_open(r[0], r[1]);
        }
        const double us = u * sum;
        pick = p.max_eta - 1;
        double cum = 0.0;
        for (int s = 0; s < p.max_eta - 1; ++s) { cum += ex[s]; if (us < cum) { pick = s; break; } }
        kept = (pick == 0) ? lv0 : lv1;
        if (lane == 0) {
            eta[g] = pick;
            if (Vc > 0) p.cur[c] = (p.cur[c] + 1 + (pick > 0)) % 3;
            p.lv_keep[c] = kept;
        }
    }
    if (!p.finish) return;
    // Eta_Sampler.logLikelihood :182-202 for this gene (every lane holds the value just drawn)
    double pri = 0.0, cacc = 0.0;
    int esum = 0;
    int ev[DSM_MAX_G];
    for (int h = 0; h < G; ++h) ev[h] = eta[h];
    if (g >= 0) ev[g] = pick;
    for (int h = 0; h < G; ++h) { pri += p.prior[ev[h]]; esum += ev[h]; }
    for (int i = lane; i < S; i += 64) {
        double ce = 0.0;
        for (int h = 0; h < G; ++h) ce = fma((double)ev[h], p.delta[(size_t)h * S + i], ce);
        if (ce < GENE_MIN_DELTA) ce = GENE_MIN_DELTA;
        cacc += poisson_term(p.cov[(size_t)c * S + i], ce);
    }
    const double ctot = group_allreduce_sum<64>(cacc);
    double ll = pri + (p.cov_const[c] + ctot);
    if (Vc > 0) ll += (esum > 0) ? (p.mult_const[c] + kept) : (double)Vc * GENE_ETA_PENALTY;
    if (lane == 0) {
        p.gene_ll[c] = ll;
        if (p.reset_star || ll > p.gene_llstar[c]) {
            p.gene_llstar[c] = ll;
            for (int h = 0; h < G; ++h) p.eta_star[(size_t)c * G + h] = ev[h];
        }
        if (p.gene_ll_trace) p.gene_ll_trace[c] = ll;
        if (p.eta_store) for (int h = 0; h < G; ++h) p.eta_store[(size_t)c * G + h] = ev[h];
    }
}

__global__ void gene_commit_kernel(int32_t *eta, int32_t *cur, int c, int G, int g, int value, int has_variants)
{
    eta[(size_t)c * G + g] = value;
    if (has_variants) cur[c] = (cur[c] + 1 + (value > 0)) % 3;
}

// =====================================================================
// Init_NMFT.factorize_tau for every gene: one workgroup per gene, one thread per row (variant, base).
// The tau update of a row needs only that row (gamma is fixed), so the whole multiplicative loop
// -- update, normalise over the four bases, objective, convergence test -- runs inside the kernel;
// sums over samples are sequential in the order of oracle/desman_oracle.c:orc_nmft_update_tau.
// =====================================================================
#define NM_EPS 2.220446049250313e-16
__device__ __forceinline__ double nz(double x) { return x == 0.0 ? NM_EPS : x; }

struct GeneNmftParams {
    const int32_t *cnt_vs, *gene_off, *eta;
    const double *gamma;        // [S][G]
    double *tauf;               // [Vtot][4][G] in/out
    double *F;                  // [Vtot*4*S] scratch, per gene [s][row]
    uint64_t *tau[3];
    const int32_t *cur;
    int32_t *n_iter;            // [C]
    int S, G, max_iter;
    double min_change;
};

template <int GMAX>
__global__ __launch_bounds__(256) void gene_nmft_kernel(GeneNmftParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_n[];
    double *gam = reinterpret_cast<double *>(smem_n);    // [G][S] masked + normalised, transposed
    double *t1 = gam + (size_t)p.G * p.S;                // [GMAX] sum_s gam
    double *red = t1 + GMAX;                             // [256]
    const int tid = threadIdx.x, G = p.G, S = p.S, c = blockIdx.x;
    const int g0 = p.gene_off[c], Vc = p.gene_off[c + 1] - g0;
    uint32_t mask = 0;
    for (int h = 0; h < G; ++h) mask |= (uint32_t)(p.eta[(size_t)c * G + h] > 0) << h;
    if (Vc == 0 || mask == 0u) { if (tid == 0) p.n_iter[c] = -1; return; }
    const int rows = 4 * Vc;
    double *F = p.F + (size_t)g0 * 4 * S;
    for (int s = tid; s < S; s += 256) {
        const double rsum = masked_row_sum(p.gamma + (size_t)s * G, G, mask);
        for (int g = 0; g < G; ++g) gam[g * S + s] = (((mask >> g) & 1u) ? p.gamma[(size_t)s * G + g] : 0.0) / rsum;
    }
    // F[s][row], row = 4 v + a: (x + 1) / (n + 4)  (Init_NMFT.py:49-60)
    for (int i = tid; i < Vc * S; i += 256) {
        const int v = i / S, s = i % S;
        const int4 x = reinterpret_cast<const int4 *>(p.cnt_vs)[(size_t)(g0 + v) * S + s];
        const double xa[4] = {(double)x.x + 1.0, (double)x.y + 1.0, (double)x.z + 1.0, (double)x.w + 1.0};
        const double tot = ((xa[0] + xa[1]) + xa[2]) + xa[3];
        for (int a = 0; a < 4; ++a) F[(size_t)s * rows + 4 * v + a] = xa[a] / tot;
    }
    __syncthreads();
    if (tid < GMAX) {
        double t = 0.0;
        if (tid < G) for (int s = 0; s < S; ++s) t += gam[tid * S + s];
        t1[tid] = t;
    }
    __syncthreads();

    auto objective = [&]() -> double {                   // div_objective :152-156 over this gene's rows
        double d = 0.0;
        for (int r = tid; r < rows; r += 256) {
            const double *tr = p.tauf + ((size_t)(g0 + r / 4) * 4 + (r & 3)) * G;
            double tg[GMAX];
#pragma unroll
            for (int g = 0; g < GMAX; ++g) tg[g] = (g < G) ? tr[g] : 0.0;
            double dr = 0.0;
            for (int s = 0; s < S; ++s) {
                double pa = 0.0;
#pragma unroll
                for (int g = 0; g < GMAX; ++g) if (g < G) pa += tg[g] * gam[g * S + s];
                if (pa < NM_EPS) pa = NM_EPS;
                const double f = F[(size_t)s * rows + r];
                dr += f * log(nz(f) / nz(pa)) - f + pa;
            }
            d += dr;
        }
        red[tid] = d;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
        const double tot = red[0];
        __syncthreads();
        return tot;
    };

    double divl = 0.0, div = objective();
    int it = 0;
    while (it < p.max_iter && fabs(divl - div) > p.min_change) {
        // rows r = 4 v + a sit in quads of lanes: the base normalisation is a quad exchange.
        // All 256 threads run the same number of passes (inactive rows idle) so the shuffles stay converged.
        for (int r0 = 0; r0 < rows; r0 += 256) {
            const int r = r0 + tid;
            const bool on = r < rows;
            double *tr = p.tauf + ((size_t)(g0 + (on ? r : 0) / 4) * 4 + (r & 3)) * G;
            double tg[GMAX], acc[GMAX];
#pragma unroll
            for (int g = 0; g < GMAX; ++g) { tg[g] = (on && g < G) ? tr[g] : 0.0; acc[g] = 0.0; }
            if (on) {
                for (int s = 0; s < S; ++s) {
                    double rr = 0.0;
#pragma unroll
                    for (int g = 0; g < GMAX; ++g) if (g < G) rr += tg[g] * gam[g * S + s];
                    const double q = nz(F[(size_t)s * rows + r]) / nz(rr);
#pragma unroll
                    for (int g = 0; g < GMAX; ++g) if (g < G) acc[g] += q * gam[g * S + s];
                }
            }
#pragma unroll
            for (int g = 0; g < GMAX; ++g) {
                if (g < G) {
                    const double nv = tg[g] * (nz(acc[g]) / nz(t1[g]));
                    const int q0 = (tid & 63) & ~3;
                    const double a0 = __shfl(nv, q0, 64), a1 = __shfl(nv, q0 + 1, 64);
                    const double a2 = __shfl(nv, q0 + 2, 64), a3 = __shfl(nv, q0 + 3, 64);
                    const double tot = ((a0 + a1) + a2) + a3;
                    if (on) tr[g] = nv / tot;
                }
            }
        }
        __syncthreads();
        divl = div;
        div = objective();
        ++it;
    }
    if (tid == 0) p.n_iter[c] = it;
    // get_tau :230-245: strict '>' against a running maximum that starts at 0.0
    uint64_t *tdst = p.tau[p.cur[c]];
    for (int v = tid; v < Vc; v += 256) {
        uint64_t t = 0;
        for (int g = 0; g < G; ++g) {
            double best = 0.0;
            int arg = 0;
            for (int a = 0; a < 4; ++a) {
                const double x = p.tauf[((size_t)(g0 + v) * 4 + a) * G + g];
                if (x > best) { best = x; arg = a; }
            }
            t |= (uint64_t)arg << (2 * g);
        }
        tdst[g0 + v] = t;
    }
}

// =====================================================================
// KLAssign: eta [C][G] >= 0 with cov ~ eta delta^T under the KL divergence.  One thread per gene row
// (rows are independent given delta); the divergence is the only global quantity.
// =====================================================================
struct KlParams {
    const double *covT;     // [S][C]
    const double *deltaT;   // [G][S]
    double *etaT;           // [G][C]
    double *partial;        // [grid]
    double *ctl;            // {divl, div, iter, stop}
    int C, S, G;
};

template <int GMAX>
__global__ __launch_bounds__(256) void kl_update_kernel(KlParams p, int do_update)
{
    extern __shared__ __attribute__((aligned(16))) char smem_k[];
    double *dl = reinterpret_cast<double *>(smem_k);     // [G][S]
    double *e1 = dl + (size_t)p.G * p.S;                 // [GMAX] delta^T row sums (div_update: eta1)
    double *red = e1 + GMAX;                             // [256]
    const int tid = threadIdx.x, G = p.G, S = p.S;
    const int c = blockIdx.x * 256 + tid;
    if (p.ctl[3] != 0.0) return;                         // converged: the remaining launches of a batch are no-ops
    for (int i = tid; i < G * S; i += 256) dl[i] = p.deltaT[i];
    __syncthreads();
    if (tid < GMAX) {
        double t = 0.0;
        if (tid < G) for (int s = 0; s < S; ++s) t += dl[tid * S + s];
        e1[tid] = t;
    }
    __syncthreads();
    double d = 0.0;
    if (c < p.C) {
        double e[GMAX], acc[GMAX];
#pragma unroll
        for (int g = 0; g < GMAX; ++g) { e[g] = (g < G) ? p.etaT[(size_t)g * p.C + c] : 0.0; acc[g] = 0.0; }
        if (do_update) {
            for (int s = 0; s < S; ++s) {
                double r = 0.0;
#pragma unroll
                for (int g = 0; g < GMAX; ++g) if (g < G) r += e[g] * dl[g * S + s];
                const double q = nz(p.covT[(size_t)s * p.C + c]) / nz(r);
#pragma unroll
                for (int g = 0; g < GMAX; ++g) if (g < G) acc[g] += q * dl[g * S + s];
            }
#pragma unroll
            for (int g = 0; g < GMAX; ++g) {
                if (g < G) {
                    double nv = e[g] * (nz(acc[g]) / nz(e1[g]));
                    if (!(nv >= NM_EPS)) nv = NM_EPS;                      // _adjustment :105-107 (np.maximum)
                    e[g] = nv;
                    p.etaT[(size_t)g * p.C + c] = nv;
                }
            }
        }
        for (int s = 0; s < S; ++s) {                                       // div_objective :113-117
            double ca = 0.0;
#pragma unroll
            for (int g = 0; g < GMAX; ++g) if (g < G) ca += e[g] * dl[g * S + s];
            const double cv = p.covT[(size_t)s * p.C + c];
            d += cv * log(nz(cv) / nz(ca)) - cv + ca;
        }
    }
    red[tid] = d;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
    if (tid == 0) p.partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void kl_reduce_kernel(const double *partial, int n, double *ctl, int counted,
                                                        int max_iter, double min_change)
{
    __shared__ double red[256];
    const int tid = threadIdx.x;
    if (ctl[3] != 0.0) return;
    double d = 0.0;
    for (int i = tid; i < n; i += 256) d += partial[i];
    red[tid] = d;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
    if (tid == 0) {
        const double div = red[0];
        ctl[0] = ctl[1];            // divl = div
        ctl[1] = div;
        if (counted) ctl[2] += 1.0;
        // while iter < max_iter and |divl - div| > min_change  (GeneAssign.py:93)
        if (!(ctl[2] < (double)max_iter && fabs(ctl[0] - ctl[1]) > min_change)) ctl[3] = 1.0;
    }
}

}   // namespace

// =====================================================================
// host object
// =====================================================================
struct dsm_genes {
    dsm_ctx *base = nullptr;        // stream, MT19937 state, log table
    int device = 0;
    int Vtot = 0, S = 0, C = 0, G = 0, max_eta = 2;
    int LPV = 64, NSL = 1, ntask = 0, block = 256;
    bool have_data = false, have_model = false, have_state = false;
    std::vector<int32_t> gene_off_h, task_off_h, eta_h;
    DBuf<int32_t> cnt_vs, gene_off, gene_of, task_tab, eta, cur, eta_star, nchange, n_iter;
    DBuf<uint64_t> tau0, tau1, tau2;
    DBuf<double> cov, gamma, eps, delta, prior, cov_const, mult_const, lv_keep, gene_ll, gene_llstar, v_ll;
    DBuf<int64_t> u_off;
    DBuf<uint32_t> u_raw;
    uint64_t ctr_seed = 0x13198A2E03707344ull;
    uint32_t iter_ctr = 0;
    int gene_base = 0;              // this object holds genes gene_base.. of a larger, sharded set
};

#define GBIND(gs) HIP_TRY(hipSetDevice((gs)->device))

static int genes_need(dsm_genes *gs, bool model, bool state)
{
    if (!gs) { dsm_set_error("null gene context"); return DSM_ERR_ARG; }
    if (!gs->have_data) { dsm_set_error("no gene data: call dsm_genes_set_data first"); return DSM_ERR_STATE; }
    if (model && !gs->have_model) { dsm_set_error("no model: call dsm_genes_set_model first"); return DSM_ERR_STATE; }
    if (state && !gs->have_state) { dsm_set_error("no state: call dsm_genes_set_state first"); return DSM_ERR_STATE; }
    return DSM_OK;
}

extern "C" int dsm_genes_create(dsm_genes **out, int device)
{
    if (!out) { dsm_set_error("null out pointer"); return DSM_ERR_ARG; }
    *out = nullptr;
    dsm_ctx *base = nullptr;
    TRY(dsm_ctx_create(&base, device));
    dsm_genes *gs = new dsm_genes();
    gs->base = base;
    gs->device = device;
    *out = gs;
    return DSM_OK;
}

extern "C" int dsm_genes_destroy(dsm_genes *gs)
{
    if (!gs) return DSM_OK;
    (void)hipSetDevice(gs->device);
    if (gs->base) { (void)hipStreamSynchronize(gs->base->stream); dsm_ctx_destroy(gs->base); }
    delete gs;
    return DSM_OK;
}

static void pick_tile(int S, int *LPV, int *NSL)
{
    if (S <= 16) { *LPV = 16; *NSL = 1; }
    else if (S <= 32) { *LPV = 16; *NSL = 2; }          // as k_tau_sweep: more rows per wavefront share the per-step draw work
    else if (S <= 48) { *LPV = 16; *NSL = 3; }
    else if (S <= 64) { *LPV = 32; *NSL = 2; }
    else if (S > 64 && S <= 96) { *LPV = 32; *NSL = 3; }
    else {
        *LPV = 64;
        const int need = (S + 63) / 64;
        *NSL = need <= 4 ? need : (need <= 6 ? 6 : 8);
    }
}

extern "C" int dsm_genes_set_data(dsm_genes *gs, const int64_t *variants, int Vtot, int S, int C,
                                  const int32_t *gene_off, const double *cov)
{
    if (!gs || !gene_off || !cov || C < 1 || S < 1 || Vtot < 0 || (Vtot > 0 && !variants)) {
        dsm_set_error("set_data: bad arguments (C=%d S=%d Vtot=%d)", C, S, Vtot);
        return DSM_ERR_ARG;
    }
    if (S > DSM_MAX_S) { dsm_set_error("S=%d exceeds DSM_MAX_S=%d", S, DSM_MAX_S); return DSM_ERR_UNSUPPORTED; }
    if (gene_off[0] != 0 || gene_off[C] != Vtot) { dsm_set_error("gene_off must run from 0 to Vtot"); return DSM_ERR_ARG; }
    for (int c = 0; c < C; ++c)
        if (gene_off[c + 1] < gene_off[c]) { dsm_set_error("gene_off not monotone at gene %d", c); return DSM_ERR_ARG; }
    GBIND(gs);
    hipStream_t st = gs->base->stream;
    gs->have_data = gs->have_state = false;
    gs->Vtot = Vtot; gs->S = S; gs->C = C;
    pick_tile(S, &gs->LPV, &gs->NSL);
    gs->gene_off_h.assign(gene_off, gene_off + C + 1);
    // task table: <= GENE_VPG consecutive variant rows of one gene per lane group
    std::vector<int32_t> tab, gof((size_t)(Vtot ? Vtot : 1));
    gs->task_off_h.assign(C + 1, 0);
    for (int c = 0; c < C; ++c) {
        for (int v = gene_off[c]; v < gene_off[c + 1]; v += GENE_VPG) { tab.push_back(c); tab.push_back(v); }
        for (int v = gene_off[c]; v < gene_off[c + 1]; ++v) gof[v] = c;
        gs->task_off_h[c + 1] = (int32_t)(tab.size() / 2);
    }
    gs->ntask = (int)(tab.size() / 2);
    TRY(gs->gene_off.resize(C + 1));
    TRY(gs->task_tab.resize(tab.size()));
    TRY(gs->gene_of.resize(gof.size()));
    TRY(gs->cov.resize((size_t)C * S));
    TRY(gs->cnt_vs.resize((size_t)Vtot * S * 4));
    TRY(gs->tau0.resize(Vtot)); TRY(gs->tau1.resize(Vtot)); TRY(gs->tau2.resize(Vtot));
    TRY(gs->cur.resize(C));
    TRY(gs->nchange.resize((size_t)C * 2));
    TRY(gs->n_iter.resize(C));
    TRY(gs->lv_keep.resize(C)); TRY(gs->gene_ll.resize(C)); TRY(gs->gene_llstar.resize(C));
    TRY(gs->v_ll.resize((size_t)2 * (Vtot ? Vtot : 1)));
    TRY(gs->u_off.resize((size_t)C * 2));
    HIP_TRY(hipMemcpyAsync(gs->gene_off, gene_off, (C + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (!tab.empty()) HIP_TRY(hipMemcpyAsync(gs->task_tab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(gs->gene_of, gof.data(), gof.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(gs->cov, cov, (size_t)C * S * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(gs->cur, 0, C * sizeof(int32_t), st));
    HIP_TRY(hipMemsetAsync(gs->v_ll, 0, (size_t)2 * (Vtot ? Vtot : 1) * sizeof(double), st));
    HIP_TRY(hipMemsetAsync(gs->lv_keep, 0, C * sizeof(double), st));
    if (Vtot > 0) {
        const size_t n = (size_t)Vtot * S * 4;
        DBuf<int64_t> raw;
        DBuf<int> flag;
        TRY(raw.resize(n));
        TRY(flag.resize(1));
        HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), st));
        HIP_TRY(hipMemcpyAsync(raw, variants, n * sizeof(int64_t), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(gene_convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, raw, gs->cnt_vs, n, flag);
        HIP_TRY(hipGetLastError());
        int bad = 0;
        HIP_TRY(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (bad) { dsm_set_error("counts must be in [0, 2^24)"); return DSM_ERR_ARG; }
        HIP_TRY(hipMemsetAsync(gs->tau0, 0, Vtot * sizeof(uint64_t), st));
        HIP_TRY(hipMemsetAsync(gs->tau1, 0, Vtot * sizeof(uint64_t), st));
        HIP_TRY(hipMemsetAsync(gs->tau2, 0, Vtot * sizeof(uint64_t), st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    gs->have_data = true;
    return DSM_OK;
}

extern "C" int dsm_genes_set_model(dsm_genes *gs, const double *gamma, const double *epsilon, const double *delta,
                                   int G, int max_eta, const double *eta_log_prior, const double *cov_const,
                                   const double *mult_const)
{
    TRY(genes_need(gs, false, false));
    if (!gamma || !epsilon || !delta || !eta_log_prior || !cov_const || !mult_const) { dsm_set_error("set_model: null pointer"); return DSM_ERR_ARG; }
    if (G < 1 || G > DSM_MAX_G) { dsm_set_error("G=%d outside 1..%d", G, DSM_MAX_G); return DSM_ERR_UNSUPPORTED; }
    if (max_eta < 2 || max_eta > DSM_MAX_ETA) { dsm_set_error("max_eta=%d outside 2..%d", max_eta, DSM_MAX_ETA); return DSM_ERR_UNSUPPORTED; }
    GBIND(gs);
    hipStream_t st = gs->base->stream;
    const int S = gs->S, C = gs->C;
    if (G != gs->G) gs->have_state = false;
    gs->G = G; gs->max_eta = max_eta;
    TRY(gs->gamma.resize((size_t)S * G)); TRY(gs->eps.resize(16)); TRY(gs->delta.resize((size_t)G * S));
    TRY(gs->prior.resize(DSM_MAX_ETA)); TRY(gs->cov_const.resize(C)); TRY(gs->mult_const.resize(C));
    TRY(gs->eta.resize((size_t)C * G)); TRY(gs->eta_star.resize((size_t)C * G));
    HIP_TRY(hipMemcpyAsync(gs->gamma, gamma, (size_t)S * G * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(gs->eps, epsilon, 16 * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(gs->delta, delta, (size_t)G * S * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(gs->prior, eta_log_prior, max_eta * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(gs->cov_const, cov_const, C * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(gs->mult_const, mult_const, C * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    gs->have_model = true;
    return DSM_OK;
}

extern "C" int dsm_genes_set_state(dsm_genes *gs, const int32_t *eta, const int64_t *tau)
{
    TRY(genes_need(gs, true, false));
    GBIND(gs);
    hipStream_t st = gs->base->stream;
    const int C = gs->C, G = gs->G, V = gs->Vtot;
    if (!eta && !gs->have_state) { dsm_set_error("set_state: eta required the first time"); return DSM_ERR_ARG; }
    if (eta) {
        for (size_t i = 0; i < (size_t)C * G; ++i)
            if (eta[i] < 0 || eta[i] >= gs->max_eta) { dsm_set_error("eta[%zu]=%d outside 0..%d", i, eta[i], gs->max_eta - 1); return DSM_ERR_ARG; }
        gs->eta_h.assign(eta, eta + (size_t)C * G);
        HIP_TRY(hipMemcpyAsync(gs->eta, eta, (size_t)C * G * sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    if (tau && V > 0) {
        DBuf<int64_t> raw;
        TRY(raw.resize((size_t)V * G * 4));
        HIP_TRY(hipMemcpyAsync(raw, tau, (size_t)V * G * 4 * sizeof(int64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(gs->cur, 0, C * sizeof(int32_t), st));
        hipLaunchKernelGGL(gene_pack_kernel, dim3((V + 255) / 256), dim3(256), 0, st, raw, gs->tau0, V, G);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    gs->have_state = true;
    return DSM_OK;
}

extern "C" int dsm_genes_get_state(dsm_genes *gs, int32_t *eta, int64_t *tau)
{
    TRY(genes_need(gs, true, true));
    GBIND(gs);
    hipStream_t st = gs->base->stream;
    const int C = gs->C, G = gs->G, V = gs->Vtot;
    if (eta) HIP_TRY(hipMemcpyAsync(eta, gs->eta, (size_t)C * G * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (tau && V > 0) {
        DBuf<int64_t> raw;
        const size_t n = (size_t)V * G;
        TRY(raw.resize(n * 4));
        hipLaunchKernelGGL(gene_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, gs->tau0, gs->tau1, gs->tau2,
                           gs->cur, gs->gene_of, raw, V, G);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(tau, raw, n * 4 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (eta) gs->eta_h.assign(eta, eta + (size_t)C * G);
    return DSM_OK;
}

extern "C" int dsm_genes_seed(dsm_genes *gs, unsigned long mt_seed, uint64_t ctr_seed)
{
    if (!gs) { dsm_set_error("null gene context"); return DSM_ERR_ARG; }
    gs->ctr_seed = ctr_seed;
    gs->iter_ctr = 0;
    return dsm_ctx_seed(gs->base, mt_seed, ctr_seed);
}
extern "C" int dsm_genes_set_gene_base(dsm_genes *gs, int gene_base)
{
    if (!gs || gene_base < 0) { dsm_set_error("set_gene_base: bad arguments"); return DSM_ERR_ARG; }
    gs->gene_base = gene_base;
    return DSM_OK;
}
extern "C" int dsm_genes_get_mt_state(dsm_genes *gs, uint32_t *state625)
{
    if (!gs) { dsm_set_error("null gene context"); return DSM_ERR_ARG; }
    return dsm_ctx_get_mt_state(gs->base, state625);
}
extern "C" int dsm_genes_set_mt_state(dsm_genes *gs, const uint32_t *state625)
{
    if (!gs) { dsm_set_error("null gene context"); return DSM_ERR_ARG; }
    return dsm_ctx_set_mt_state(gs->base, state625);
}

// ---------------------------------------------------------------- launch helpers
// lane groups per workgroup and dynamic LDS: log table + epsilon + one [G][SP] gamma tile per group; groups are
// dropped until the workgroup fits 64 KB (two or more workgroups per CU), the hard limit is the 160 KB of a CU
static int sweep_geometry(const dsm_genes *gs, int *block, size_t *lds)
{
    const size_t tile = (size_t)gs->G * gs->LPV * gs->NSL * sizeof(double);
    const size_t fixed = (2 * DSM_LOG_TAB_N + 16) * sizeof(double);
    int gpb = 256 / gs->LPV;
    while (gpb > 1 && fixed + gpb * tile > 64 * 1024) gpb >>= 1;
    if (fixed + gpb * tile > 160 * 1024) { dsm_set_error("gamma tile (%zu B) exceeds LDS", tile); return DSM_ERR_UNSUPPORTED; }
    *block = gpb * gs->LPV;
    *lds = fixed + gpb * tile;
    return DSM_OK;
}

static GeneSweepParams sweep_params(dsm_genes *gs, const int32_t *d_eta, const uint32_t *u_raw, int step_g, int task_base,
                                    int task_end, uint32_t iter)
{
    GeneSweepParams p;
    p.cnt_vs = gs->cnt_vs; p.gene_off = gs->gene_off; p.task_tab = gs->task_tab;
    p.tau[0] = gs->tau0; p.tau[1] = gs->tau1; p.tau[2] = gs->tau2;
    p.cur = gs->cur; p.eta = d_eta; p.gamma = gs->gamma; p.eps = gs->eps; p.log_tab = gs->base->log_tab;
    p.u_raw = u_raw; p.u_off = gs->u_off; p.v_ll = gs->v_ll; p.nchange = gs->nchange;
    p.S = gs->S; p.G = gs->G; p.Vtot = gs->Vtot; p.task_base = task_base; p.task_end = task_end; p.step_g = step_g;
    p.gene_base = gs->gene_base;
    p.k0 = (uint32_t)gs->ctr_seed; p.k1 = (uint32_t)(gs->ctr_seed >> 32); p.iter = iter;
    return p;
}

template <int LPV, int NSL>
static void launch_sweep_t(const GeneSweepParams &p, bool sweep, int nb, int ncand, int block, size_t sh, hipStream_t st)
{
    if (sweep) hipLaunchKernelGGL((gene_sweep_kernel<LPV, NSL, true>), dim3(nb, ncand), dim3(block), sh, st, p);
    else hipLaunchKernelGGL((gene_sweep_kernel<LPV, NSL, false>), dim3(nb, ncand), dim3(block), sh, st, p);
}

static int launch_sweep(dsm_genes *gs, const GeneSweepParams &p, bool sweep, int ncand)
{
    const int ntask = p.task_end - p.task_base;
    if (ntask <= 0) return DSM_OK;
    int block = 256;
    size_t sh = 0;
    TRY(sweep_geometry(gs, &block, &sh));
    const int gpb = block / gs->LPV, nb = (ntask + gpb - 1) / gpb;
    hipStream_t st = gs->base->stream;
#define GS_CASE(L, N) if (gs->LPV == L && gs->NSL == N) launch_sweep_t<L, N>(p, sweep, nb, ncand, block, sh, st)
    GS_CASE(16, 1); GS_CASE(16, 2); GS_CASE(16, 3); GS_CASE(32, 1); GS_CASE(32, 2); GS_CASE(32, 3); GS_CASE(64, 1); GS_CASE(64, 2); GS_CASE(64, 3); GS_CASE(64, 4); GS_CASE(64, 6); GS_CASE(64, 8);
#undef GS_CASE
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

static GeneChooseParams choose_params(dsm_genes *gs, int step_g, int finish, int reset_star, uint32_t iter)
{
    GeneChooseParams p;
    p.gene_off = gs->gene_off; p.eta = gs->eta; p.cur = gs->cur;
    p.cov = gs->cov; p.delta = gs->delta; p.prior = gs->prior; p.cov_const = gs->cov_const; p.mult_const = gs->mult_const;
    p.v_ll = gs->v_ll; p.lv_keep = gs->lv_keep; p.gene_ll = gs->gene_ll; p.gene_llstar = gs->gene_llstar;
    p.eta_star = gs->eta_star; p.eta_store = nullptr; p.gene_ll_trace = nullptr; p.u_ext = nullptr;
    p.C = gs->C; p.S = gs->S; p.G = gs->G; p.max_eta = gs->max_eta; p.Vtot = gs->Vtot; p.gene_base = gs->gene_base;
    p.step_g = step_g; p.finish = finish; p.reset_star = reset_star;
    p.k0 = (uint32_t)gs->ctr_seed; p.k1 = (uint32_t)(gs->ctr_seed >> 32); p.iter = iter;
    return p;
}

static int launch_choose(dsm_genes *gs, const GeneChooseParams &p)
{
    hipLaunchKernelGGL(gene_choose_kernel, dim3((gs->C + 3) / 4), dim3(256), 0, gs->base->stream, p);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

// mask source on the device: the resident eta or a host array
static int mask_source(dsm_genes *gs, const int32_t *eta_mask, DBuf<int32_t> &tmp, const int32_t **d_eta, std::vector<int32_t> &host)
{
    const size_t n = (size_t)gs->C * gs->G;
    if (eta_mask) {
        TRY(tmp.resize(n));
        HIP_TRY(hipMemcpyAsync(tmp, eta_mask, n * sizeof(int32_t), hipMemcpyHostToDevice, gs->base->stream));
        host.assign(eta_mask, eta_mask + n);
        *d_eta = tmp;
    } else {
        host.resize(n);
        HIP_TRY(hipMemcpyAsync(host.data(), gs->eta, n * sizeof(int32_t), hipMemcpyDeviceToHost, gs->base->stream));
        HIP_TRY(hipStreamSynchronize(gs->base->stream));
        *d_eta = gs->eta;
    }
    return DSM_OK;
}

static bool row_active(const std::vector<int32_t> &eta, int c, int G)
{
    for (int h = 0; h < G; ++h) if (eta[(size_t)c * G + h] > 0) return true;
    return false;
}

// ---------------------------------------------------------------- per-gene NMFT
extern "C" int dsm_genes_nmft_tau(dsm_genes *gs, const int32_t *eta_mask, const double *tau_init, int max_iter,
                                  double min_change, int32_t *n_iter)
{
    TRY(genes_need(gs, true, true));
    if (!tau_init) { dsm_set_error("nmft_tau: null tau_init"); return DSM_ERR_ARG; }
    GBIND(gs);
    hipStream_t st = gs->base->stream;
    const int V = gs->Vtot, G = gs->G, S = gs->S, C = gs->C;
    if (V == 0) { if (n_iter) for (int c = 0; c < C; ++c) n_iter[c] = -1; return DSM_OK; }
    DBuf<int32_t> tmp;
    const int32_t *d_eta = nullptr;
    std::vector<int32_t> host;
    TRY(mask_source(gs, eta_mask, tmp, &d_eta, host));
    DBuf<double> tauf, F;
    TRY(tauf.resize((size_t)V * 4 * G));
    TRY(F.resize((size_t)V * 4 * S));
    HIP_TRY(hipMemcpyAsync(tauf, tau_init, (size_t)V * 4 * G * sizeof(double), hipMemcpyHostToDevice, st));
    GeneNmftParams p;
    p.cnt_vs = gs->cnt_vs; p.gene_off = gs->gene_off; p.eta = d_eta; p.gamma = gs->gamma; p.tauf = tauf; p.F = F;
    p.tau[0] = gs->tau0; p.tau[1] = gs->tau1; p.tau[2] = gs->tau2; p.cur = gs->cur; p.n_iter = gs->n_iter;
    p.S = S; p.G = G; p.max_iter = max_iter; p.min_change = min_change;
    const int GM = G <= 4 ? 4 : G <= 8 ? 8 : G <= 16 ? 16 : 32;
    const size_t sh = ((size_t)G * S + GM + 256) * sizeof(double);
    if (sh > 160 * 1024) { dsm_set_error("gamma tile (%zu B) exceeds LDS", sh); return DSM_ERR_UNSUPPORTED; }
    if (GM == 4) hipLaunchKernelGGL(gene_nmft_kernel<4>, dim3(C), dim3(256), sh, st, p);
    else if (GM == 8) hipLaunchKernelGGL(gene_nmft_kernel<8>, dim3(C), dim3(256), sh, st, p);
    else if (GM == 16) hipLaunchKernelGGL(gene_nmft_kernel<16>, dim3(C), dim3(256), sh, st, p);
    else hipLaunchKernelGGL(gene_nmft_kernel<32>, dim3(C), dim3(256), sh, st, p);
    HIP_TRY(hipGetLastError());
    if (n_iter) HIP_TRY(hipMemcpyAsync(n_iter, gs->n_iter, C * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DSM_OK;
}

// ---------------------------------------------------------------- sweeps of all genes
extern "C" int dsm_genes_sweep_all(dsm_genes *gs, const int32_t *eta_mask, int sweep, int32_t *nchange, double *logvar,
                                   double *v_ll)
{
    TRY(genes_need(gs, true, true));
    GBIND(gs);
    hipStream_t st = gs->base->stream;
    const int C = gs->C, G = gs->G;
    DBuf<int32_t> tmp;
    const int32_t *d_eta = nullptr;
    std::vector<int32_t> host;
    TRY(mask_source(gs, eta_mask, tmp, &d_eta, host));
    const uint32_t *u = nullptr;
    if (sweep == 1 && gs->ntask > 0) {
        // GSL stream in gene order, genes without variants or with an empty mask draw nothing
        std::vector<int64_t> off((size_t)C * 2, 0);
        int64_t pos = 0;
        for (int c = 0; c < C; ++c) {
            const int64_t n = (int64_t)(gs->gene_off_h[c + 1] - gs->gene_off_h[c]) * G;
            off[(size_t)c * 2] = pos;
            if (n > 0 && row_active(host, c, G)) pos += n;
        }
        if (!gs->base->mt_seeded) { dsm_set_error("tau RNG not seeded: call dsm_genes_seed"); return DSM_ERR_STATE; }
        TRY(gs->u_raw.resize((size_t)pos));
        HIP_TRY(hipMemcpyAsync(gs->u_off, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
        TRY(k_mt_fill(gs->base, gs->u_raw, (size_t)pos, st));
        u = gs->u_raw;
        HIP_TRY(hipMemsetAsync(gs->nchange, 0, (size_t)C * 2 * sizeof(int32_t), st));
    }
    if (sweep == 2 && gs->ntask > 0) HIP_TRY(hipMemsetAsync(gs->nchange, 0, (size_t)C * 2 * sizeof(int32_t), st));
    const uint32_t iter = (sweep == 2) ? gs->iter_ctr++ : 0u;     // counter-based draws: a fresh iteration index per sweep
    const GeneSweepParams p = sweep_params(gs, d_eta, u, -1, 0, gs->ntask, iter);
    TRY(launch_sweep(gs, p, sweep != 0, 1));
    std::vector<double> vl((size_t)(gs->Vtot ? gs->Vtot : 1));
    std::vector<int32_t> nch((size_t)C * 2, 0);
    if (gs->Vtot > 0) HIP_TRY(hipMemcpyAsync(vl.data(), gs->v_ll, gs->Vtot * sizeof(double), hipMemcpyDeviceToHost, st));
    if (sweep && gs->ntask > 0) HIP_TRY(hipMemcpyAsync(nch.data(), gs->nchange, nch.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int c = 0; c < C; ++c) {
        const int lo = gs->gene_off_h[c], hi = gs->gene_off_h[c + 1];
        const bool on = hi > lo && row_active(host, c, G);
        double t = 0.0;
        for (int v = lo; v < hi; ++v) {
            if (!on) vl[v] = 0.0;                       // skipped genes: nothing was evaluated
            t += vl[v];
        }
        if (logvar) logvar[c] = t;
        if (nchange) nchange[c] = on ? nch[(size_t)c * 2] : -1;
    }
    if (v_ll && gs->Vtot > 0) memcpy(v_ll, vl.data(), gs->Vtot * sizeof(double));
    return DSM_OK;
}

// ---------------------------------------------------------------- reference-order single step
extern "C" int dsm_genes_step_candidates(dsm_genes *gs, int c, int g, double *logvar, int *swept)
{
    TRY(genes_need(gs, true, true));
    if (c < 0 || c >= gs->C || g < 0 || g >= gs->G || !logvar) { dsm_set_error("step_candidates: bad arguments"); return DSM_ERR_ARG; }
    GBIND(gs);
    hipStream_t st = gs->base->stream;
    const int G = gs->G;
    const int Vc = gs->gene_off_h[c + 1] - gs->gene_off_h[c];
    logvar[0] = logvar[1] = 0.0;
    if (swept) swept[0] = swept[1] = 0;
    if (Vc == 0) return DSM_OK;
    if (gs->eta_h.size() != (size_t)gs->C * G) { dsm_set_error("host eta mirror missing"); return DSM_ERR_STATE; }
    bool any0 = false;
    for (int h = 0; h < G; ++h) any0 |= (h != g) && gs->eta_h[(size_t)c * G + h] > 0;
    if (!gs->base->mt_seeded) { dsm_set_error("tau RNG not seeded: call dsm_genes_seed"); return DSM_ERR_STATE; }
    const int64_t n = (int64_t)Vc * G;
    const int64_t off[2] = {0, any0 ? n : 0};          // candidate 0 draws first, only if it sweeps (update:237-238)
    const size_t words = (size_t)(any0 ? 2 * n : n);
    TRY(gs->u_raw.resize(words));
    HIP_TRY(hipMemcpyAsync(gs->u_off.p + (size_t)c * 2, off, sizeof off, hipMemcpyHostToDevice, st));
    TRY(k_mt_fill(gs->base, gs->u_raw, words, st));
    const GeneSweepParams p = sweep_params(gs, gs->eta, gs->u_raw, g, gs->task_off_h[c], gs->task_off_h[c + 1], 0);
    TRY(launch_sweep(gs, p, true, 2));
    const int lo = gs->gene_off_h[c];
    std::vector<double> vl((size_t)2 * Vc);
    HIP_TRY(hipMemcpyAsync(vl.data(), gs->v_ll.p + lo, Vc * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(vl.data() + Vc, gs->v_ll.p + gs->Vtot + lo, Vc * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int k = 0; k < 2; ++k) {
        double t = 0.0;
        for (int v = 0; v < Vc; ++v) t += vl[(size_t)k * Vc + v];
        logvar[k] = t;
    }
    if (!any0) logvar[0] = -1.0e20;
    if (swept) { swept[0] = any0; swept[1] = 1; }
    return DSM_OK;
}

extern "C" int dsm_genes_step_choose(dsm_genes *gs, int c, int g, int eta_value)
{
    TRY(genes_need(gs, true, true));
    if (c < 0 || c >= gs->C || g < 0 || g >= gs->G || eta_value < 0 || eta_value >= gs->max_eta) {
        dsm_set_error("step_choose: bad arguments"); return DSM_ERR_ARG;
    }
    GBIND(gs);
    const int Vc = gs->gene_off_h[c + 1] - gs->gene_off_h[c];
    hipLaunchKernelGGL(gene_commit_kernel, dim3(1), dim3(1), 0, gs->base->stream, gs->eta, gs->cur, c, gs->G, g, eta_value, Vc > 0);
    HIP_TRY(hipGetLastError());
    gs->eta_h[(size_t)c * gs->G + g] = eta_value;
    return DSM_OK;
}

// ---------------------------------------------------------------- log-likelihood / batched update
static int eval_genes(dsm_genes *gs, int reset_star)
{
    const GeneSweepParams sp = sweep_params(gs, gs->eta, nullptr, -1, 0, gs->ntask, 0);
    TRY(launch_sweep(gs, sp, false, 1));
    const GeneChooseParams cp = choose_params(gs, -1, 1, reset_star, 0);
    return launch_choose(gs, cp);
}

extern "C" int dsm_genes_loglik(dsm_genes *gs, double *gene_ll)
{
    TRY(genes_need(gs, true, true));
    GBIND(gs);
    hipStream_t st = gs->base->stream;
    DBuf<double> keep_ll;
    DBuf<int32_t> keep_eta;
    // the evaluation must not disturb the MAP record: run it against scratch copies
    TRY(keep_ll.resize(gs->C));
    TRY(keep_eta.resize((size_t)gs->C * gs->G));
    HIP_TRY(hipMemcpyAsync(keep_ll, gs->gene_llstar, gs->C * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(keep_eta, gs->eta_star, (size_t)gs->C * gs->G * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    TRY(eval_genes(gs, 1));
    HIP_TRY(hipMemcpyAsync(gs->gene_llstar, keep_ll, gs->C * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(gs->eta_star, keep_eta, (size_t)gs->C * gs->G * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (gene_ll) HIP_TRY(hipMemcpyAsync(gene_ll, gs->gene_ll, gs->C * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DSM_OK;
}

extern "C" int dsm_genes_update(dsm_genes *gs, int n_iter, int reset_star, int32_t *eta_store, double *gene_ll_trace,
                                const uint32_t *u_tau_ext, const double *u_eta_ext)
{
    TRY(genes_need(gs, true, true));
    if (n_iter < 0) { dsm_set_error("update: n_iter < 0"); return DSM_ERR_ARG; }
    GBIND(gs);
    hipStream_t st = gs->base->stream;
    const int C = gs->C, G = gs->G;
    const size_t cg = (size_t)C * G, vg = (size_t)gs->Vtot * G;
    DBuf<int32_t> d_store;
    DBuf<double> d_trace, d_ue;
    DBuf<uint32_t> d_ut;
    TRY(d_store.resize(cg * (n_iter ? n_iter : 1)));
    TRY(d_trace.resize((size_t)C * (n_iter ? n_iter : 1)));
    if (u_eta_ext) {
        TRY(d_ue.resize(cg * n_iter));
        HIP_TRY(hipMemcpyAsync(d_ue, u_eta_ext, cg * n_iter * sizeof(double), hipMemcpyHostToDevice, st));
    }
    if (u_tau_ext && vg > 0) {
        TRY(d_ut.resize((size_t)n_iter * G * 2 * vg));
        HIP_TRY(hipMemcpyAsync(d_ut, u_tau_ext, (size_t)n_iter * G * 2 * vg * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        std::vector<int64_t> off((size_t)C * 2);
        for (int c = 0; c < C; ++c) {                      // test layout: [it][g][k][Vtot*G], gene rows in place
            off[(size_t)c * 2] = (int64_t)gs->gene_off_h[c] * G;
            off[(size_t)c * 2 + 1] = (int64_t)vg + (int64_t)gs->gene_off_h[c] * G;
        }
        HIP_TRY(hipMemcpyAsync(gs->u_off, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (reset_star) TRY(eval_genes(gs, 1));
    for (int it = 0; it < n_iter; ++it) {
        const uint32_t iter = gs->iter_ctr++;
        for (int g = 0; g < G; ++g) {
            const uint32_t *u = (u_tau_ext && vg > 0) ? d_ut.p + ((size_t)it * G + g) * 2 * vg : nullptr;
            const GeneSweepParams sp = sweep_params(gs, gs->eta, u, g, 0, gs->ntask, iter);
            TRY(launch_sweep(gs, sp, true, 2));
            GeneChooseParams cp = choose_params(gs, g, g == G - 1, 0, iter);
            if (u_eta_ext) cp.u_ext = d_ue.p + (size_t)it * cg;
            if (g == G - 1) { cp.eta_store = d_store.p + (size_t)it * cg; cp.gene_ll_trace = d_trace.p + (size_t)it * C; }
            TRY(launch_choose(gs, cp));
        }
    }
    if (eta_store && n_iter) HIP_TRY(hipMemcpyAsync(eta_store, d_store, cg * n_iter * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (gene_ll_trace && n_iter) HIP_TRY(hipMemcpyAsync(gene_ll_trace, d_trace, (size_t)C * n_iter * sizeof(double), hipMemcpyDeviceToHost, st));
    gs->eta_h.resize(cg);
    HIP_TRY(hipMemcpyAsync(gs->eta_h.data(), gs->eta, cg * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DSM_OK;
}

extern "C" int dsm_genes_get_star(dsm_genes *gs, int32_t *eta_star, double *gene_llstar)
{
    TRY(genes_need(gs, true, true));
    GBIND(gs);
    hipStream_t st = gs->base->stream;
    if (eta_star) HIP_TRY(hipMemcpyAsync(eta_star, gs->eta_star, (size_t)gs->C * gs->G * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (gene_llstar) HIP_TRY(hipMemcpyAsync(gene_llstar, gs->gene_llstar, gs->C * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DSM_OK;
}

extern "C" int dsm_genes_set_star(dsm_genes *gs, const int32_t *eta_star, const double *gene_llstar)
{
    TRY(genes_need(gs, true, true));
    GBIND(gs);
    hipStream_t st = gs->base->stream;
    if (eta_star) HIP_TRY(hipMemcpyAsync(gs->eta_star, eta_star, (size_t)gs->C * gs->G * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (gene_llstar) HIP_TRY(hipMemcpyAsync(gs->gene_llstar, gene_llstar, gs->C * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DSM_OK;
}

// ---------------------------------------------------------------- KLAssign
extern "C" int dsm_kl_assign(int device, const double *cov, const double *delta, double *eta, int C, int S, int G,
                             int max_iter, double min_change, int *n_done, double *div)
{
    if (!cov || !delta || !eta || C < 1 || S < 1 || G < 1) { dsm_set_error("kl_assign: bad arguments"); return DSM_ERR_ARG; }
    if (G > DSM_MAX_G) { dsm_set_error("G=%d exceeds DSM_MAX_G=%d", G, DSM_MAX_G); return DSM_ERR_UNSUPPORTED; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { dsm_set_error("no HIP device"); return DSM_ERR_NODEVICE; }
    HIP_TRY(hipSetDevice(device));
    // transposed copies: lanes = consecutive genes -> coalesced
    std::vector<double> covT((size_t)S * C), deltaT((size_t)G * S), etaT((size_t)G * C);
    for (int c = 0; c < C; ++c) for (int s = 0; s < S; ++s) covT[(size_t)s * C + c] = cov[(size_t)c * S + s];
    for (int s = 0; s < S; ++s) for (int g = 0; g < G; ++g) deltaT[(size_t)g * S + s] = delta[(size_t)s * G + g];
    for (int c = 0; c < C; ++c) for (int g = 0; g < G; ++g) etaT[(size_t)g * C + c] = eta[(size_t)c * G + g];
    const int grid = (C + 255) / 256;
    DBuf<double> d_cov, d_delta, d_eta, d_part, d_ctl;
    TRY(d_cov.resize(covT.size())); TRY(d_delta.resize(deltaT.size())); TRY(d_eta.resize(etaT.size()));
    TRY(d_part.resize(grid)); TRY(d_ctl.resize(4));
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreate(&st));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{st};
    HIP_TRY(hipMemcpyAsync(d_cov, covT.data(), covT.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_delta, deltaT.data(), deltaT.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_eta, etaT.data(), etaT.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_ctl, 0, 4 * sizeof(double), st));
    KlParams p;
    p.covT = d_cov; p.deltaT = d_delta; p.etaT = d_eta; p.partial = d_part; p.ctl = d_ctl; p.C = C; p.S = S; p.G = G;
    const int GM = G <= 4 ? 4 : G <= 8 ? 8 : G <= 16 ? 16 : 32;
    const size_t sh = ((size_t)G * S + GM + 256) * sizeof(double);
    if (sh > 160 * 1024) { dsm_set_error("delta tile (%zu B) exceeds LDS", sh); return DSM_ERR_UNSUPPORTED; }
    auto launch = [&](int upd) {
        if (GM == 4) hipLaunchKernelGGL(kl_update_kernel<4>, dim3(grid), dim3(256), sh, st, p, upd);
        else if (GM == 8) hipLaunchKernelGGL(kl_update_kernel<8>, dim3(grid), dim3(256), sh, st, p, upd);
        else if (GM == 16) hipLaunchKernelGGL(kl_update_kernel<16>, dim3(grid), dim3(256), sh, st, p, upd);
        else hipLaunchKernelGGL(kl_update_kernel<32>, dim3(grid), dim3(256), sh, st, p, upd);
        hipLaunchKernelGGL(kl_reduce_kernel, dim3(1), dim3(256), 0, st, d_part, grid, d_ctl, upd, max_iter, min_change);
    };
    launch(0);                                            // div of the start values (divl = 0)
    HIP_TRY(hipGetLastError());
    double ctl[4] = {0, 0, 0, 0};
    for (;;) {
        HIP_TRY(hipMemcpyAsync(ctl, d_ctl, sizeof ctl, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (ctl[3] != 0.0) break;
        for (int i = 0; i < 64; ++i) launch(1);           // a batch; launches after convergence are no-ops
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(etaT.data(), d_eta, etaT.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int c = 0; c < C; ++c) for (int g = 0; g < G; ++g) eta[(size_t)c * G + g] = etaT[(size_t)g * C + c];
    if (n_done) *n_done = (int)ctl[2];
    if (div) *div = ctl[1];
    return DSM_OK;
}

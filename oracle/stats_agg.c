/*
 * stats_agg.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY (same rules as desman_oracle.c).
 *
 * SPECIFICATION of the product's aggregated auxiliary-count (mu/E) sampler, "spec v2"
 * (kernels: stats_agg_kernel + the stage-2 part of dirichlet_kernel).  Like orc_stats_counter
 * (spec v1, per-read draws) this is NOT a reference function: the reference draws mu/E from numpy's
 * serial RandomState stream (desman/HaploSNP_Sampler.py:284-309), which a parallel sampler cannot
 * replay.  Only the sums  sum_mu[s,g] = sum_{v,b} mu  and  Esum[b,a] = sum_{v,s} E  are ever consumed
 * (HaploSNP_Sampler.py:266, :276), and spec v2 samples THOSE from their exact joint law at a cost that
 * does not grow with the read depth:
 *
 *   Law.  For a cell (v,s) and observed base b with x reads, the reference assigns every read a true
 *   base a and a haplotype g with probability  gamma[s,g] * eta[tau_vg,b] / sum  (one-stage form,
 *   SURVEY App. A2).  Because eta[tau_vg,b] depends on g only through the base a = tau_vg:
 *     stage 1   (n_a)_a ~ Multinomial(x; W_a),  W_a = eta[a,b] * Gamma_a,  Gamma_a = sum_{g: tau_vg=a} gamma[s,g]
 *               Esum[b,a] += n_a
 *     stage 2   the n_a reads spread over H_a(v) = {g: tau_vg = a} with probabilities gamma[s,g]/Gamma_a.
 *   Stage 2 has the same probability vector for every (v, b, a) of sample s that shares the SET H, and
 *   a sum of independent multinomials with one probability vector is one multinomial of the summed
 *   count: N[s,H] = sum of the n_a with H_a(v) = H, then  m ~ Multinomial(N[s,H]; gamma[s,g], g in H),
 *   sum_mu[s,g] += m_g.  The multinomial over H is drawn by halving the haplotype range recursively
 *   (one binomial per (node, subset)), and the halves are aggregated again, so the number of large-count
 *   binomials per sample is about 2^G, not V.
 *   The joint law of (sum_mu, Esum) is exactly the reference's (checked statistically against the
 *   reference-pinned rn.sample_mu and against the exact moments, tests/).
 *
 *   Draws.  Everything is counter-based: the stage-1 draws of a (cell, observed base) item come from one
 *   xoshiro128+ stream, consumed in the order written below (items are independent work units: the kernels
 *   process the few items that need the rejection sampler in a separate, compacted launch).  Its seed is
 *   Philox4x32-10(ctr = {cell, 0, iter, 'STA1'}, key = seed) -- shared by the four items of the cell --
 *   followed by three more Philox rounds whose keys depend on the observed base (item_seed); every
 *   stage-2 binomial owns the stream Philox(ctr = {subset, s | node << 16 | level << 24, iter, 'STA2'}).
 *   A cell's x reads of one observed base are split by ONE binomial into "heaviest true base" / "others"
 *   and the (few) others are drawn read by read against 32-bit thresholds (<= XS of them) or by two more
 *   binomials.  Binomial(n, p): sequential-search inversion on the rarer outcome while its mean is <= 128
 *   (no transcendental function: (1-q)^n by repeated squaring), Hoermann's BTRS transformed rejection
 *   (1993) above -- cost O(1) in the depth either way; the stage-2 counts go up to 2^32-1.
 *   All arithmetic is IEEE double with the operation order written here (-ffp-contract=off), the
 *   logarithm of BTRS is the table-driven orc_tlog (same table and FMA sequence as the device's
 *   dsm_log), so the kernels reproduce this file bit for bit.
 *
 *   Two versions of the draws, selected by `spec` (same law; tests/test_law_cpu.py pins both to the exact pmf and to the
 *   reference's sampleMu).  The product runs spec 2 by default; spec 3 was written to shorten stage 1 and measured no
 *   faster on MI355X -- it stays selectable and tested:
 *     spec 2   as above: (1-q)^n by repeated squaring, q/(1-q) and 1-q by two divisions, item streams three Philox
 *              rounds off the cell's Philox-10 block;
 *     spec 3   the start of the inversion search is f0 = exp(-n ln(1 + r)), r = q/(1-q) the ONE division of the
 *              draw, with the table logarithm orc_tlog and a table exponential orc_texp (32-entry 2^(j/32) table,
 *              degree-6 polynomial: |relative error| < 1e-15, hence < 1e-13 on f0) -- a fixed ~35 instructions
 *              instead of a loop over the bits of n that runs to the deepest lane of the wavefront; item streams
 *              two Philox rounds off the cell's block.  Everything else is spec 2.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "orc_log_table.h"

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);   /* desman_oracle.c */

#define STREAM_STA1 0x53544131u   /* 'STA1' */
#define STREAM_STA2 0x53544132u   /* 'STA2' */
#define STREAM_TEST 0x54455354u   /* 'TEST' */
#define XS 128u                   /* up to XS non-dominant reads are drawn read by read */
#define BINV_MEAN_CAP 128.0       /* stage 1: inversion while the mean of the rarer outcome is <= 128 (throughput-bound wavefronts) */
#define BINV_MEAN_CAP_S2 16.0     /* stage 2: <= 16 (one latency-bound binomial per lane: BTRS is the shorter chain) */
#define BINV_KMAX 255u            /* the search stops at 255 (> 11 sigma at mean 128): 1/k comes from a 256-entry table */

typedef struct { uint32_t s[4]; } xo_t;

static inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

static inline uint32_t xo_next(xo_t *r)     /* xoshiro128+ */
{
    uint32_t *s = r->s;
    uint32_t res = s[0] + s[3];
    uint32_t t = s[1] << 9;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl32(s[3], 11);
    return res;
}

static void xo_seed(xo_t *r, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, const uint32_t key[2])
{
    uint32_t ctr[4] = { c0, c1, c2, c3 };
    orc_philox4x32_10(ctr, key, r->s);
    if ((r->s[0] | r->s[1] | r->s[2] | r->s[3]) == 0) r->s[0] = 1;
}

/* one Philox4x32 round (Salmon et al. 2011) */
static void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}

/* stream of the item (cell, observed base b): `base` = Philox4x32-10({cell, 0, iter, 'STA1'}) of the cell, then three
 * rounds with the key (key + (b + 1) * Weyl constants), bumped per round as in Philox */
static void item_seed(xo_t *r, const uint32_t base[4], uint32_t b, const uint32_t key[2], int spec)
{
    uint32_t c[4] = { base[0], base[1], base[2], base[3] };
    uint32_t k0 = key[0] ^ (0x9E3779B9u * (b + 1u)), k1 = key[1] ^ (0xBB67AE85u * (b + 1u));
    const int rounds = spec >= 3 ? 2 : 3;            /* two rounds put every output word behind a keyed multiplication */
    for (int i = 0; i < rounds; i++) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    memcpy(r->s, c, sizeof c);
    if ((r->s[0] | r->s[1] | r->s[2] | r->s[3]) == 0) r->s[0] = 1;
}

/* uniform in (0,1): two words, first word = high part */
static double xo_u01(xo_t *r)
{
    uint32_t a = xo_next(r);
    uint32_t b = xo_next(r);
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6) + 0.5) * (1.0 / 9007199254740992.0);
}

/* the device's dsm_log (desman_amd/csrc/dsm_device.h) restated: x = 2^e m, table on the top 8 mantissa
 * bits, r = m*invc - 1 by one fma, degree-4 polynomial in r. */
double orc_tlog(double x)
{
    uint64_t bits;
    memcpy(&bits, &x, 8);
    const uint32_t hi = (uint32_t)(bits >> 32);
    if ((uint32_t)(hi - 0x00100000u) >= 0x7fe00000u) return log(x);     /* zero / subnormal / inf / nan / negative */
    const int e = (int)(hi >> 20) - 1023;
    const uint32_t idx = (hi >> 12) & 255u;
    const double invc = orc_log_table[2 * idx], logc = orc_log_table[2 * idx + 1];
    const uint64_t mb = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m;
    memcpy(&m, &mb, 8);
    const double r = fma(m, invc, -1.0);
    const double ed = (double)e;
    double p = fma(r, 0.2, -0.25);
    p = fma(r, p, 1.0 / 3.0);
    p = fma(r, p, -0.5);
    const double w = fma(ed, 0x1.62e42fefa3800p-1, logc);
    const double lo = fma(ed, 0x1.ef35793c76730p-45, (r * r) * p);
    return (w + r) + lo;
}

/* exp(y) for y <= 0 (the device's dsm_texp restated): y = (32 k + j) ln2/32 + r, |r| <= ln2/64;
 * exp(r) - 1 by a degree-6 polynomial (next term r^7/5040 < 4e-18), 2^(j/32) from the table, 2^k by ldexp. */
double orc_texp(double y)
{
    if (!(y > -700.0)) return 0.0;
    const double kd = rint(y * ORC_EXP_INV_LN2_32);
    const int ki = (int)kd;
    double r = fma(kd, -ORC_EXP_LN2_32_HI, y);
    r = fma(kd, -ORC_EXP_LN2_32_LO, r);
    double p = fma(r, 1.0 / 720.0, 1.0 / 120.0);
    p = fma(r, p, 1.0 / 24.0);
    p = fma(r, p, 1.0 / 6.0);
    p = fma(r, p, 0.5);
    p = fma(r, p, 1.0);
    p = r * p;                                       /* exp(r) - 1 */
    const double t = orc_exp_table[ki & 31];
    return ldexp(fma(t, p, t), ki >> 5);             /* arithmetic shift: floor(ki / 32) */
}

/* v >= 0 -> floor(v) saturated to 2^32-1 (what v_cvt_u32_f64 does); NaN -> 0 */
static inline uint32_t cvt_sat_u32(double v)
{
    if (!(v >= 0.0)) return 0u;
    return v >= 4294967295.0 ? 0xffffffffu : (uint32_t)v;
}

/* b^e by repeated squaring, multiplications in this order */
static double pw(double b, uint32_t e)
{
    double res = 1.0;
    while (e) {
        if (e & 1u) res = res * b;
        e >>= 1;
        if (e) b = b * b;
    }
    return res;
}

/* x reads over K (3 or 4) categories with weights w, read by read: n_j = #{reads whose word r falls in
 * [t_{j-1}, t_j)}, t_j = floor(2^32 * cum_j / total) */
static void draw_reads(xo_t *rng, uint32_t x, const double *w, int K, uint32_t *n)
{
    double cums[4], cum = 0.0;
    for (int j = 0; j < K; j++) { cum = cum + w[j]; cums[j] = cum; }
    const double scale = 4294967296.0 / cums[K - 1];
    uint32_t t[3], c[3] = { 0, 0, 0 };
    for (int j = 0; j < K - 1; j++) t[j] = cvt_sat_u32(cums[j] * scale);
    for (uint32_t i = 0; i < x; i++) {
        const uint32_t r = xo_next(rng);
        for (int j = 0; j < K - 1; j++) c[j] += (r < t[j]);
    }
    n[0] = c[0];
    for (int j = 1; j < K - 1; j++) n[j] = c[j] - c[j - 1];
    n[K - 1] = x - c[K - 2];
}

/* Binomial(c, q) by sequential search from 0; f0 = (1-q)^c, r = q/(1-q), c q <= 128.  One uniform. */
static uint32_t binv(xo_t *rng, uint32_t c, double f0, double r)
{
    /* P(k)/P(k-1) = r (c-k+1)/k = r (c+1) (1/k) - r */
    double u = xo_u01(rng), f = f0;
    const double rc1 = r * ((double)c + 1.0);
    uint32_t k = 0;
    const uint32_t kend = c < BINV_KMAX ? c : BINV_KMAX;
    while (u >= f && k < kend) {
        u = u - f;
        k = k + 1;
        f = f * fma(rc1, 1.0 / (double)k, -r);
    }
    return k;
}

/* Stirling series remainder  ln k! - [ (k+1/2) ln(k+1) - (k+1) + ln sqrt(2 pi) ]  (Hoermann 1993, fc(k)) */
static double stirling_tail(double k)
{
    static const double tab[10] = { 0.0810614667953272, 0.0413406959554092, 0.0276779256849983, 0.02079067210376509,
                                    0.0166446911898211, 0.0138761288230707, 0.0118967099458917, 0.0104112652619720,
                                    0.00925546218271273, 0.00833056343336287 };
    if (k <= 9.0) return tab[(int)k];
    const double inv = 1.0 / (k + 1.0), inv2 = inv * inv;
    return (1.0 / 12.0 - (1.0 / 360.0 - (1.0 / 1260.0) * inv2) * inv2) * inv;
}

/* Hoermann's BTRS (transformed rejection with squeeze), q <= 1/2, n q > 16 (BTRS needs n q >= 10) */
static uint32_t btrs(xo_t *rng, uint32_t n, double q)
{
    const double nd = (double)n;
    const double spq = sqrt(nd * q * (1.0 - q));
    const double b = 1.15 + 2.53 * spq;
    const double a = -0.0873 + 0.0248 * b + 0.01 * q;
    const double c = nd * q + 0.5;
    const double ib = 1.0 / b;
    const double v_r = 0.92 - 4.2 * ib;
    const double alpha = (2.83 + 5.1 * ib) * spq;
    const double r = q / (1.0 - q);
    const double m = floor((nd + 1.0) * q);
    const double nm1 = nd - m + 1.0;
    /* the part of the acceptance bound that does not depend on the proposal */
    const double h = (m + 0.5) * orc_tlog((m + 1.0) / (r * nm1)) + stirling_tail(m) + stirling_tail(nd - m);
    for (int attempt = 0; attempt < 4096; attempt++) {
        const double u = xo_u01(rng) - 0.5;
        const double v = xo_u01(rng);
        const double us = 0.5 - fabs(u);
        const double kd = floor((2.0 * a / us + b) * u + c);
        if (kd < 0.0 || kd > nd) continue;
        if (us >= 0.07 && v <= v_r) return (uint32_t)kd;
        const double nk1 = nd - kd + 1.0;
        const double lv = orc_tlog(v * alpha) - orc_tlog(a / (us * us) + b);
        const double ub = h + (nd + 1.0) * orc_tlog(nm1 / nk1) + (kd + 0.5) * orc_tlog(r * nk1 / (kd + 1.0))
                          - stirling_tail(kd) - stirling_tail(nd - kd);
        if (lv <= ub) return (uint32_t)kd;
    }
    return (uint32_t)m;      /* unreachable in practice (acceptance > 0.7 per attempt) */
}

/* successes among n trials with success : failure odds wa : wb */
static uint32_t binom(xo_t *rng, uint32_t n, double wa, double wb, double cap, int spec)
{
    if (n == 0 || !(wa > 0.0)) return 0;
    if (!(wb > 0.0)) return n;
    const int flip = wa > wb;
    const double ws = flip ? wb : wa, wl = flip ? wa : wb;      /* the rarer outcome has odds ws : wl */
    const double T = ws + wl;
    uint32_t k;
    if ((double)n * ws > cap * T) k = btrs(rng, n, ws / T);
    else if (spec >= 3) {
        const double r = ws / wl;                                /* q / (1 - q); 1 - q = 1 / (1 + r) */
        k = binv(rng, n, orc_texp(-((double)n * orc_tlog(1.0 + r))), r);
    } else k = binv(rng, n, pw(wl / T, n), ws / wl);
    return flip ? n - k : k;
}

/* x reads of one (cell, observed base) over the four true bases with weights W */
static void mult4(xo_t *rng, uint32_t x, const double W[4], uint32_t n[4], int spec)
{
    n[0] = n[1] = n[2] = n[3] = 0;
    if (x == 0) return;
    int am = 0;
    for (int a = 1; a < 4; a++) if (W[a] > W[am]) am = a;
    int o[3], j = 0;
    for (int a = 0; a < 4; a++) if (a != am) o[j++] = a;
    const double wo[3] = { W[o[0]], W[o[1]], W[o[2]] };
    const double ws = (wo[0] + wo[1]) + wo[2];
    const uint32_t m = binom(rng, x, ws, W[am], BINV_MEAN_CAP, spec);            /* reads NOT of the heaviest base */
    n[am] = x - m;
    if (m == 0) return;
    uint32_t k[3];
    if (m <= XS) draw_reads(rng, m, wo, 3, k);
    else {
        k[0] = binom(rng, m, wo[0], wo[1] + wo[2], BINV_MEAN_CAP, spec);
        k[1] = binom(rng, m - k[0], wo[1], wo[2], BINV_MEAN_CAP, spec);
        k[2] = m - k[0] - k[1];
    }
    n[o[0]] = k[0]; n[o[1]] = k[1]; n[o[2]] = k[2];
}
/* stage 1 for all cells: esum [4,4] ([observed][true]) accumulated, ntab [S][2^G] (subset counts) accumulated */
static void stage1(const uint8_t *tau_idx, const double *gamma, const double *eta, const int64_t *variants,
                   int V, int G, int S, const uint32_t key[2], uint32_t iter, uint64_t *esum, uint32_t *ntab, int spec)
{
    const size_t NH = (size_t)1 << G;
    for (int v = 0; v < V; v++) {
        const uint8_t *tv = tau_idx + (size_t)v * G;
        uint32_t H[4] = { 0, 0, 0, 0 };
        for (int g = 0; g < G; g++) H[tv[g]] |= 1u << g;
        for (int s = 0; s < S; s++) {
            const int64_t *x = variants + ((size_t)v * S + s) * 4;
            const uint32_t cell = (uint32_t)((uint64_t)s * (uint64_t)V + (uint64_t)v);
            uint32_t cctr[4] = { cell, 0u, iter, STREAM_STA1 }, cbase[4];
            orc_philox4x32_10(cctr, key, cbase);
            double Gam[4] = { 0.0, 0.0, 0.0, 0.0 };
            for (int g = 0; g < G; g++) Gam[tv[g]] = Gam[tv[g]] + gamma[(size_t)s * G + g];
            uint32_t nacc[4] = { 0, 0, 0, 0 };
            for (int b = 0; b < 4; b++) {
                if (x[b] <= 0) continue;
                double W[4];
                for (int a = 0; a < 4; a++) W[a] = eta[a * 4 + b] * Gam[a];
                const double Wt = ((W[0] + W[1]) + W[2]) + W[3];
                if (!(Wt > 0.0)) for (int a = 0; a < 4; a++) W[a] = Gam[a];     /* degenerate eta: fall back to abundance */
                uint32_t n[4];
                xo_t rng;
                item_seed(&rng, cbase, (uint32_t)b, key, spec);
                mult4(&rng, (uint32_t)x[b], W, n, spec);
                for (int a = 0; a < 4; a++) { esum[b * 4 + a] += n[a]; nacc[a] += n[a]; }
            }
            for (int a = 0; a < 4; a++) if (nacc[a]) ntab[(size_t)s * NH + H[a]] += nacc[a];
        }
    }
}

/* stage 2 for one sample: T0 [2^G] subset counts -> sum_mu_s [G] accumulated.  The haplotype range is
 * halved recursively: node (level, idx) owns the bit range [lo,hi); the lower child gets floor(w/2) bits. */
typedef struct { int lo, hi; uint32_t *tab; } node_t;

static void stage2_sample(int s, int G, const double *gam_s, const uint32_t *T0, const uint32_t key[2], uint32_t iter,
                          uint64_t *sum_mu_s, int spec)
{
    node_t *cur = (node_t *)malloc(sizeof(node_t) * 64), *nxt = (node_t *)malloc(sizeof(node_t) * 64);
    int ncur = 1;
    cur[0].lo = 0; cur[0].hi = G;
    cur[0].tab = (uint32_t *)malloc(sizeof(uint32_t) << G);
    memcpy(cur[0].tab, T0, sizeof(uint32_t) << G);
    int idx_cur[64], idx_nxt[64];
    idx_cur[0] = 0;
    for (int level = 0; ncur > 0; level++) {
        int nn = 0;
        for (int i = 0; i < ncur; i++) {
            const int lo = cur[i].lo, hi = cur[i].hi, w = hi - lo;
            uint32_t *T = cur[i].tab;
            if (w == 1) { sum_mu_s[lo] += T[1]; free(T); continue; }
            const int wl = w / 2, wh = w - wl, mid = lo + wl;
            uint32_t *L = (uint32_t *)calloc((size_t)1 << wl, sizeof(uint32_t));
            uint32_t *R = (uint32_t *)calloc((size_t)1 << wh, sizeof(uint32_t));
            for (uint32_t Hs = 1; Hs < (1u << w); Hs++) {
                const uint32_t n = T[Hs];
                if (!n) continue;
                const uint32_t HL = Hs & ((1u << wl) - 1u), HR = Hs >> wl;
                if (!HR) { L[HL] += n; continue; }
                if (!HL) { R[HR] += n; continue; }
                double wL = 0.0, wR = 0.0;
                for (int j = 0; j < wl; j++) if (HL >> j & 1u) wL = wL + gam_s[lo + j];
                for (int j = 0; j < wh; j++) if (HR >> j & 1u) wR = wR + gam_s[mid + j];
                xo_t rng;
                xo_seed(&rng, Hs, (uint32_t)s | ((uint32_t)idx_cur[i] << 16) | ((uint32_t)level << 24), iter, STREAM_STA2, key);
                const uint32_t k = binom(&rng, n, wL, wR, BINV_MEAN_CAP_S2, spec);
                L[HL] += k; R[HR] += n - k;
            }
            free(T);
            nxt[nn].lo = lo; nxt[nn].hi = mid; nxt[nn].tab = L; idx_nxt[nn] = 2 * idx_cur[i]; nn++;
            nxt[nn].lo = mid; nxt[nn].hi = hi; nxt[nn].tab = R; idx_nxt[nn] = 2 * idx_cur[i] + 1; nn++;
        }
        node_t *t = cur; cur = nxt; nxt = t;
        memcpy(idx_cur, idx_nxt, sizeof(int) * (size_t)nn);
        ncur = nn;
    }
    free(cur); free(nxt);
}

/* sum_mu [S,G] and esum [4,4] are ACCUMULATED into.  G <= 16.  ntab_out (optional, [S][2^G]) receives the
 * stage-1 subset counts.  spec = 2 or 3 (header).  Returns 0, or -1 if G / spec is out of range / out of memory. */
int orc_stats_agg(const uint8_t *tau_idx, const double *gamma, const double *eta, const int64_t *variants,
                  int V, int G, int S, uint64_t seed, uint32_t iter, uint64_t *sum_mu, uint64_t *esum,
                  uint32_t *ntab_out, int spec)
{
    if (G < 1 || G > 16 || spec < 2 || spec > 3) return -1;
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    const size_t NH = (size_t)1 << G;
    uint32_t *ntab = (uint32_t *)calloc((size_t)S * NH, sizeof(uint32_t));
    if (!ntab) return -1;
    stage1(tau_idx, gamma, eta, variants, V, G, S, key, iter, esum, ntab, spec);
    if (ntab_out) memcpy(ntab_out, ntab, (size_t)S * NH * sizeof(uint32_t));
    for (int s = 0; s < S; s++)
        stage2_sample(s, G, gamma + (size_t)s * G, ntab + (size_t)s * NH, key, iter, sum_mu + (size_t)s * G, spec);
    free(ntab);
    return 0;
}

/* test hooks: nsamp variates of one sampler, variate i from the stream Philox({i, 0, 0, 'TEST'}, seed) */
void orc_binom_test(int kind, uint32_t n, double wa, double wb, uint64_t seed, int nsamp, uint32_t *out, int spec)
{
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    for (int i = 0; i < nsamp; i++) {
        xo_t rng;
        xo_seed(&rng, (uint32_t)i, 0u, 0u, STREAM_TEST, key);
        out[i] = binom(&rng, n, wa, wb, kind == 0 ? BINV_MEAN_CAP : BINV_MEAN_CAP_S2, spec);
    }
}

void orc_mult4_test(uint32_t x, const double *W, uint64_t seed, int nsamp, uint32_t *out /* [nsamp][4] */, int spec)
{
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    for (int i = 0; i < nsamp; i++) {
        xo_t rng;
        xo_seed(&rng, (uint32_t)i, 0u, 0u, STREAM_TEST, key);
        mult4(&rng, x, W, out + (size_t)i * 4, spec);
    }
}

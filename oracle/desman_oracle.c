/*
 * desman_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the arithmetic on DESMAN's haplotype-inference hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / the timed CPU baseline.
 * The product (desman_amd + libdesman_hip.so) never links or calls it.
 *
 * Every function cites the reference lines it follows (paths relative to the
 * upstream DESMAN tree).  The code is written from the algorithm, not copied.
 *
 * Pinning status
 *   - tau sweep: conditional log-probabilities pinned against the reference's
 *     own pure-Python sampler (desman/HaploSNP_Sampler.py:148-183) through
 *     tests/golden/tau_sweep_*.npz; the uniform stream is the published
 *     MT19937 (GSL gsl_rng_mt19937: genrand_int32()/2^32, seed 0 -> 4357) and
 *     is pinned against numpy's legacy MT19937 raw stream.  The reference's
 *     sampletau/c_sample_tau.c itself is UNBUILDABLE in this image (it needs
 *     the GSL headers/library, which are absent, and no stand-in is written).
 *   - log-likelihood / log-posterior / NMFT / get_tau: pinned against the
 *     imported Python reference through the tests/golden fixtures.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off: plain IEEE C
 * semantics, no FMA contraction, so the result does not depend on -march).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* MT19937, GSL flavour (third-party dependency of the reference: GSL,       */
/* unpinned system library; algorithm = Matsumoto & Nishimura 2002 version). */
/* Used by the reference at sampletau/c_sample_tau.c:26-45 and :174.         */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint32_t mt[624];
    int pos;
} orc_mt;

void orc_mt_seed(orc_mt *r, unsigned long seed)
{
    if (seed == 0) seed = 4357;      /* gsl_rng_set(mt19937, 0) default */
    r->mt[0] = (uint32_t)(seed & 0xffffffffUL);
    for (int i = 1; i < 624; i++) {
        uint32_t p = r->mt[i - 1];
        r->mt[i] = (uint32_t)(1812433253UL * (p ^ (p >> 30)) + (uint32_t)i);
    }
    r->pos = 624;
}

static void mt_refill(orc_mt *r)
{
    uint32_t *m = r->mt;
    for (int i = 0; i < 624; i++) {
        uint32_t y = (m[i] & 0x80000000u) | (m[(i + 1) % 624] & 0x7fffffffu);
        uint32_t t = m[(i + 397) % 624] ^ (y >> 1);
        if (y & 1u) t ^= 0x9908b0dfu;
        m[i] = t;
    }
    r->pos = 0;
}

uint32_t orc_mt_u32(orc_mt *r)
{
    if (r->pos >= 624) mt_refill(r);
    uint32_t y = r->mt[r->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* gsl_rng_uniform for mt19937: 32-bit draw / 2^32, in [0,1). */
double orc_mt_uniform(orc_mt *r) { return orc_mt_u32(r) / 4294967296.0; }

void orc_mt_fill_u32(orc_mt *r, uint32_t *out, long n)
{
    for (long i = 0; i < n; i++) out[i] = orc_mt_u32(r);
}

/* process-global generator, mirroring `static gsl_rng *ptGSLRNG`
 * (sampletau/c_sample_tau.c:24) and c_initRNG/c_setRNG/c_freeRNG (:26-45). */
static orc_mt *g_rng = NULL;

void orc_initRNG(void)
{
    if (!g_rng) g_rng = (orc_mt *)malloc(sizeof(orc_mt));
    orc_mt_seed(g_rng, 0);      /* gsl_rng_alloc leaves the default seed */
}
void orc_setRNG(unsigned long seed) { if (g_rng) orc_mt_seed(g_rng, seed); }
void orc_freeRNG(void) { free(g_rng); g_rng = NULL; }

/* ------------------------------------------------------------------------ */
/* tau Gibbs sweep: sampletau/c_sample_tau.c:95-204 (+ normaliseLog4 :48-70, */
/* sample4 :72-91).                                                          */
/*   tau      [V,G,4] int64 one-hot, updated in place                        */
/*   pi       [S,G]   f64  (gamma)                                           */
/*   eta      [4,4]   f64  rows = true base, cols = observed base            */
/*   variants [V,S,4] int64 counts                                           */
/*   u        [V*G]   uniforms consumed v-major, g-minor (one per (v,g))     */
/*   logp_out [V,G,4] optional: un-normalised conditional log-probs          */
/* returns number of (v,g) whose base changed.                               */
/* ------------------------------------------------------------------------ */
static int draw4(const double *logp, double u)
{
    /* normaliseLog4: shift by max, exponentiate, divide by the sum */
    double mx = logp[0];
    for (int a = 1; a < 4; a++) if (logp[a] > mx) mx = logp[a];
    double e[4], sum = 0.0;
    for (int a = 0; a < 4; a++) { e[a] = logp[a] - mx; sum += exp(e[a]); }
    double p[4];
    for (int a = 0; a < 4; a++) p[a] = exp(e[a]) / sum;
    /* sample4: inverse CDF, last edge forced to 1.0 */
    double c0 = p[0], c1 = p[1] + c0, c2 = p[2] + c1;
    if (u < c0) return 0;
    if (u < c1) return 1;
    if (u < c2) return 2;
    return 3;
}

int orc_sample_tau_u(int64_t *tau, const double *pi, const double *eta,
                     const int64_t *variants, int V, int G, int S,
                     const double *u, double *logp_out)
{
    int nchange = 0;
    int *idx = (int *)malloc(sizeof(int) * (size_t)G);
    double *rest = (double *)malloc(sizeof(double) * (size_t)S * 4);
    if (!idx || !rest) { free(idx); free(rest); return -1; }

    for (int v = 0; v < V; v++) {
        int64_t *tv = tau + (size_t)v * G * 4;
        const int64_t *xv = variants + (size_t)v * S * 4;
        for (int g = 0; g < G; g++) {            /* :107-127 one-hot -> index */
            idx[g] = 0;
            for (int b = 0; b < 4; b++) if (tv[g * 4 + b] == 1) { idx[g] = b; break; }
        }
        for (int g = 0; g < G; g++) {
            /* mixture of all haplotypes but g, h ascending (:136-150) */
            for (int s = 0; s < S; s++)
                for (int b = 0; b < 4; b++) {
                    double acc = 0.0;
                    for (int h = 0; h < G; h++)
                        if (h != g) acc += eta[idx[h] * 4 + b] * pi[s * G + h];
                    rest[s * 4 + b] = acc;
                }
            double logp[4];
            for (int a = 0; a < 4; a++) {        /* :152-169 */
                double acc = 0.0;
                for (int s = 0; s < S; s++)
                    for (int b = 0; b < 4; b++) {
                        double p = rest[s * 4 + b];
                        p += eta[a * 4 + b] * pi[s * G + g];
                        /* the count goes through float before the product (:164) */
                        double term = ((float)xv[s * 4 + b]) * log(p);
                        acc += term;
                    }
                logp[a] = acc;
            }
            if (logp_out) memcpy(logp_out + ((size_t)v * G + g) * 4, logp, sizeof logp);
            int t = draw4(logp, u[(size_t)v * G + g]);   /* :172-176 */
            if (t != idx[g]) {                          /* :178-185 */
                tv[g * 4 + idx[g]] = 0;
                tv[g * 4 + t] = 1;
                idx[g] = t;
                nchange++;
            }
        }
    }
    free(idx); free(rest);
    return nchange;
}

/* c_sample_tau with the process-global MT19937 stream (:95, :174). */
int orc_sample_tau(int64_t *tau, const double *pi, const double *eta,
                   const int64_t *variants, int V, int G, int S)
{
    if (!g_rng) return -2;
    size_t n = (size_t)V * G;
    double *u = (double *)malloc(sizeof(double) * (n ? n : 1));
    if (!u) return -1;
    for (size_t i = 0; i < n; i++) u[i] = orc_mt_uniform(g_rng);
    int r = orc_sample_tau_u(tau, pi, eta, variants, V, G, S, u, NULL);
    free(u);
    return r;
}

/* ------------------------------------------------------------------------ */
/* logLikelihood  (desman/HaploSNP_Sampler.py:431-442,                       */
/*                 desman/Desman_Utils.py:23-33)                             */
/* logPosterior   (desman/HaploSNP_Sampler.py:444-461,                       */
/*                 desman/Desman_Utils.py:35-44)                             */
/* tau_idx [V,G] uint8 base index of each haplotype.                         */
/* ------------------------------------------------------------------------ */
double orc_loglik(const uint8_t *tau_idx, const double *gamma, const double *eta,
                  const int64_t *variants, int V, int G, int S)
{
    double ll = 0.0;
    for (int v = 0; v < V; v++)
        for (int s = 0; s < S; s++) {
            const int64_t *x = variants + ((size_t)v * S + s) * 4;
            int64_t n = x[0] + x[1] + x[2] + x[3];
            double t = lgamma((double)n + 1.0), sub = 0.0, dot = 0.0;
            for (int b = 0; b < 4; b++) {
                double p = 0.0;
                for (int g = 0; g < G; g++)
                    p += gamma[s * G + g] * eta[tau_idx[(size_t)v * G + g] * 4 + b];
                sub += lgamma((double)x[b] + 1.0);
                dot += (double)x[b] * log(p);
            }
            ll += t - sub + dot;
        }
    return ll;
}

/* data-only constant of the log-likelihood: sum_vs [lnG(n+1) - sum_b lnG(x_b+1)] */
double orc_loglik_const(const int64_t *variants, int V, int S)
{
    double c = 0.0;
    for (size_t i = 0; i < (size_t)V * S; i++) {
        const int64_t *x = variants + i * 4;
        int64_t n = x[0] + x[1] + x[2] + x[3];
        double sub = 0.0;
        for (int b = 0; b < 4; b++) sub += lgamma((double)x[b] + 1.0);
        c += lgamma((double)n + 1.0) - sub;
    }
    return c;
}

static double log_dirichlet(const double *x, int n, double a)
{
    /* Desman_Utils.py:35-44 with a constant concentration vector */
    double r = lgamma(a * n);
    for (int i = 0; i < n; i++) { r += (a - 1.0) * log(x[i]); r -= lgamma(a); }
    return r;
}

double orc_logprior(const double *gamma, const double *eta, int V, int G, int S,
                    double alpha, double delta)
{
    double lg = 0.0, le = 0.0;
    for (int s = 0; s < S; s++) lg += log_dirichlet(gamma + (size_t)s * G, G, alpha);
    for (int a = 0; a < 4; a++) le += log_dirichlet(eta + a * 4, 4, delta);
    return lg + le + (double)V * (double)G * log(1.0 / 4.0);
}

double orc_logpost(const uint8_t *tau_idx, const double *gamma, const double *eta,
                   const int64_t *variants, int V, int G, int S,
                   double alpha, double delta)
{
    return orc_loglik(tau_idx, gamma, eta, variants, V, G, S) +
           orc_logprior(gamma, eta, V, G, S, alpha, delta);
}

/* ------------------------------------------------------------------------ */
/* Init_NMFT  (desman/Init_NMFT.py)                                          */
/*   F    [4V,S]  base-major row blocks: row v + a*V  (:55-60)               */
/*   tau  [4V,G]  same row order                                             */
/*   gam  [G,S]                                                              */
/* ------------------------------------------------------------------------ */
#define ORC_EPS 2.220446049250313e-16

/* Init_NMFT.__init__ :49-60, BASE_PRIOR = 1 */
void orc_nmft_freq(const int64_t *variants, int V, int S, double *F)
{
    for (int v = 0; v < V; v++)
        for (int s = 0; s < S; s++) {
            const int64_t *x = variants + ((size_t)v * S + s) * 4;
            double tot = 0.0;
            for (int a = 0; a < 4; a++) tot += (double)x[a] + 1.0;
            for (int a = 0; a < 4; a++)
                F[((size_t)a * V + v) * S + s] = ((double)x[a] + 1.0) / tot;
        }
}

static inline double nz(double x) { return x == 0.0 ? ORC_EPS : x; }  /* du.elop */

/* div_objective :152-156 */
double orc_nmft_objective(const double *F, const double *tau, const double *gam,
                          int V, int G, int S)
{
    size_t N = (size_t)4 * V;
    double d = 0.0;
    for (size_t n = 0; n < N; n++)
        for (int s = 0; s < S; s++) {
            double pa = 0.0;
            for (int g = 0; g < G; g++) pa += tau[n * G + g] * gam[(size_t)g * S + s];
            if (pa < ORC_EPS) pa = ORC_EPS;
            double f = F[n * S + s];
            d += f * log(nz(f) / nz(pa)) - f + pa;
        }
    return d;
}

/* _adjustment :88-91 */
void orc_nmft_adjust(double *tau, double *gam, int V, int G, int S)
{
    for (size_t i = 0; i < (size_t)4 * V * G; i++) if (tau[i] < ORC_EPS) tau[i] = ORC_EPS;
    for (size_t i = 0; i < (size_t)G * S; i++) if (gam[i] < ORC_EPS) gam[i] = ORC_EPS;
}

/* gamma half of div_update :160-168 */
static void nmft_update_gamma(const double *F, const double *tau, double *gam,
                              int V, int G, int S)
{
    size_t N = (size_t)4 * V;
    double *num = (double *)calloc((size_t)G * S, sizeof(double));
    double *h1 = (double *)calloc((size_t)G, sizeof(double));
    for (size_t n = 0; n < N; n++) {
        for (int g = 0; g < G; g++) h1[g] += tau[n * G + g];
        for (int s = 0; s < S; s++) {
            double r = 0.0;
            for (int g = 0; g < G; g++) r += tau[n * G + g] * gam[(size_t)g * S + s];
            double q = nz(F[n * S + s]) / nz(r);
            for (int g = 0; g < G; g++) num[(size_t)g * S + s] += tau[n * G + g] * q;
        }
    }
    if (G > 1) {
        for (int g = 0; g < G; g++)
            for (int s = 0; s < S; s++)
                gam[(size_t)g * S + s] *= nz(num[(size_t)g * S + s]) / nz(h1[g]);
        for (int s = 0; s < S; s++) {
            double tot = 0.0;
            for (int g = 0; g < G; g++) tot += gam[(size_t)g * S + s];
            for (int g = 0; g < G; g++) gam[(size_t)g * S + s] /= tot;
        }
    } else {
        for (int s = 0; s < S; s++) gam[s] = 1.0;
    }
    free(num); free(h1);
}

/* tau half of div_update :170-181 == div_update_tau :192-205 */
void orc_nmft_update_tau(const double *F, double *tau, const double *gam,
                         int V, int G, int S)
{
    size_t N = (size_t)4 * V;
    double *t1 = (double *)calloc((size_t)G, sizeof(double));
    for (int g = 0; g < G; g++)
        for (int s = 0; s < S; s++) t1[g] += gam[(size_t)g * S + s];
    double *acc = (double *)malloc(sizeof(double) * (size_t)G);
    for (size_t n = 0; n < N; n++) {
        for (int g = 0; g < G; g++) acc[g] = 0.0;
        for (int s = 0; s < S; s++) {
            double r = 0.0;
            for (int g = 0; g < G; g++) r += tau[n * G + g] * gam[(size_t)g * S + s];
            double q = nz(F[n * S + s]) / nz(r);
            for (int g = 0; g < G; g++) acc[g] += q * gam[(size_t)g * S + s];
        }
        for (int g = 0; g < G; g++) tau[n * G + g] *= nz(acc[g]) / nz(t1[g]);
    }
    for (int v = 0; v < V; v++)
        for (int g = 0; g < G; g++) {
            double tot = 0.0;
            for (int a = 0; a < 4; a++) tot += tau[((size_t)a * V + v) * G + g];
            for (int a = 0; a < 4; a++) tau[((size_t)a * V + v) * G + g] /= tot;
        }
    free(t1); free(acc);
}

void orc_nmft_update(const double *F, double *tau, double *gam, int V, int G, int S)
{
    nmft_update_gamma(F, tau, gam, V, G, S);
    orc_nmft_update_tau(F, tau, gam, V, G, S);
}

/* factorize loop :98-115 (init already drawn by the caller, :101-102 applied
 * here).  div_trace (optional, length max_iter+1) receives div before the
 * first update and after every update.  Returns the number of updates. */
int orc_nmft_factorize(const double *F, double *tau, double *gam, int V, int G, int S,
                       int max_iter, double min_change, double *div_trace)
{
    orc_nmft_adjust(tau, gam, V, G, S);
    double divl = 0.0, div = orc_nmft_objective(F, tau, gam, V, G, S);
    int it = 0;
    if (div_trace) div_trace[0] = div;
    while (it < max_iter && fabs(divl - div) > min_change) {
        orc_nmft_update(F, tau, gam, V, G, S);
        orc_nmft_adjust(tau, gam, V, G, S);
        divl = div;
        div = orc_nmft_objective(F, tau, gam, V, G, S);
        it++;
        if (div_trace) div_trace[it] = div;
    }
    return it;
}

/* factorize_tau loop :134-149 (gamma fixed, no _adjustment) */
int orc_nmft_factorize_tau(const double *F, double *tau, const double *gam, int V, int G,
                           int S, int max_iter, double min_change, double *div_trace)
{
    double divl = 0.0, div = orc_nmft_objective(F, tau, gam, V, G, S);
    int it = 0;
    if (div_trace) div_trace[0] = div;
    while (it < max_iter && fabs(divl - div) > min_change) {
        orc_nmft_update_tau(F, tau, gam, V, G, S);
        divl = div;
        div = orc_nmft_objective(F, tau, gam, V, G, S);
        it++;
        if (div_trace) div_trace[it] = div;
    }
    return it;
}

/* get_tau :230-245: strict '>' against a running max that starts at 0.0 */
void orc_nmft_get_tau(const double *tau, int V, int G, uint8_t *tau_idx)
{
    for (int v = 0; v < V; v++)
        for (int g = 0; g < G; g++) {
            double best = 0.0; int arg = 0;
            for (int a = 0; a < 4; a++) {
                double t = tau[((size_t)a * V + v) * G + g];
                if (t > best) { best = t; arg = a; }
            }
            tau_idx[(size_t)v * G + g] = (uint8_t)arg;
        }
}

/* ------------------------------------------------------------------------ */
/* Counter-based sampler SPECIFICATION of the product's auxiliary-count      */
/* (mu/E) draw.  This is NOT a reference function: the reference draws mu/E  */
/* from numpy's serial RandomState stream (HaploSNP_Sampler.py:284-309),     */
/* which no parallel sampler can replay.  The product instead draws, for     */
/* every read of observed base b at (v,s), its haplotype g with probability  */
/* gamma[s,g]*eta[tau_vg,b]/sum (the one-stage form of the same joint law,   */
/* SURVEY App. A2) from a Philox4x32-10 / xoshiro128+ stream keyed by       */
/* (seed, iteration, cell, observed base).  The restatement below lets tests check the HIP  */
/* kernel bit-for-bit; equivalence in law to the reference's sampleMu is     */
/* checked statistically against oracle/ref_numpy.py.                        */
/* ------------------------------------------------------------------------ */
static inline uint32_t mulhi32(uint32_t a, uint32_t b)
{
    return (uint32_t)(((uint64_t)a * b) >> 32);
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint32_t h0 = mulhi32(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = mulhi32(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

typedef struct { uint32_t s[4]; } orc_xo;

static inline uint32_t xo_next(orc_xo *r)     /* xoshiro128+ (Blackman & Vigna) */
{
    uint32_t *s = r->s;
    uint32_t res = s[0] + s[3];
    uint32_t t = s[1] << 9;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl32(s[3], 11);
    return res;
}

#define ORC_STREAM_STATS 0x53544154u   /* 'STAT' */

/* sum_mu [S,G] and esum [4,4] ([observed b][true a]) are ACCUMULATED into. */
/* Reads of an item are drawn in chunks: chunk j (reads j*CH .. j*CH+CH-1) owns the stream
 * Philox(cell, j, iter, STAT + base); CH depends only on the problem shape (orc_stats_chunk): small
 * problems are cut into short chunks so that no single stream is long, large ones are not cut (j = 0). */
int64_t orc_stats_chunk(int V, int S)
{
    const int64_t cells = (int64_t)V * S;
    return cells <= 65536 ? 64 : cells <= 262144 ? 128 : ((int64_t)1 << 40);
}

void orc_stats_counter(const uint8_t *tau_idx, const double *gamma, const double *eta,
                       const int64_t *variants, int V, int G, int S,
                       uint64_t seed, uint32_t iter,
                       uint64_t *sum_mu, uint64_t *esum)
{
    const int64_t CH = orc_stats_chunk(V, S);
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t thr[64];
    uint64_t cnt[64];
    for (int s = 0; s < S; s++)
        for (int v = 0; v < V; v++) {
            const int64_t *x = variants + ((size_t)v * S + s) * 4;
            const uint8_t *tv = tau_idx + (size_t)v * G;
            uint64_t cell = (uint64_t)s * (uint64_t)V + (uint64_t)v;
            /* every (cell, observed base) pair owns one stream: the four bases of a
             * cell are independent work items and may be processed in any order */
            for (int b = 0; b < 4; b++) {
                int64_t nb = x[b];
                if (nb <= 0) continue;
                double c = 0.0, cum[64];
                for (int g = 0; g < G; g++) { c += gamma[(size_t)s * G + g] * eta[tv[g] * 4 + b]; cum[g] = c; }
                double scale = 4294967296.0 / c;
                for (int g = 0; g < G - 1; g++) {
                    double t = floor(cum[g] * scale);
                    thr[g] = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
                }
                for (int g = 0; g < G; g++) cnt[g] = 0;      /* cnt[g] = #{r < thr[g]} */
                for (int64_t j = 0; j * CH < nb; j++) {
                    /* (cell < 2^32 is checked by the caller: V*S < 2^32) */
                    uint32_t ctr[4] = { (uint32_t)cell, (uint32_t)j, iter, ORC_STREAM_STATS + (uint32_t)b };
                    orc_xo rng;
                    orc_philox4x32_10(ctr, key, rng.s);
                    if ((rng.s[0] | rng.s[1] | rng.s[2] | rng.s[3]) == 0) rng.s[0] = 1;
                    int64_t n = nb - j * CH < CH ? nb - j * CH : CH;
                    for (int64_t i = 0; i < n; i++) {
                        uint32_t r = xo_next(&rng);
                        for (int g = 0; g < G - 1; g++) cnt[g] += (r < thr[g]);
                    }
                }
                for (int g = 0; g < G; g++) {
                    /* haplotype g gets the reads with thr[g-1] <= r < thr[g] */
                    uint64_t hi = (g == G - 1) ? (uint64_t)nb : cnt[g];
                    uint64_t lo = (g == 0) ? 0 : cnt[g - 1];
                    uint64_t m = hi - lo;
                    sum_mu[(size_t)s * G + g] += m;
                    esum[b * 4 + tv[g]] += m;
                }
            }
        }
}

/* Exact conditional mean of the auxiliary-count sums (for statistical tests):
 * E[sum_mu[s,g]] and E[esum[b,a]] given (tau,gamma,eta,counts). */
void orc_stats_expect(const uint8_t *tau_idx, const double *gamma, const double *eta,
                      const int64_t *variants, int V, int G, int S,
                      double *e_mu, double *v_mu, double *e_E)
{
    for (int v = 0; v < V; v++)
        for (int s = 0; s < S; s++) {
            const int64_t *x = variants + ((size_t)v * S + s) * 4;
            const uint8_t *tv = tau_idx + (size_t)v * G;
            for (int b = 0; b < 4; b++) {
                double tot = 0.0;
                for (int g = 0; g < G; g++) tot += gamma[(size_t)s * G + g] * eta[tv[g] * 4 + b];
                for (int g = 0; g < G; g++) {
                    double p = gamma[(size_t)s * G + g] * eta[tv[g] * 4 + b] / tot;
                    e_mu[(size_t)s * G + g] += (double)x[b] * p;
                    v_mu[(size_t)s * G + g] += (double)x[b] * p * (1.0 - p);
                    e_E[b * 4 + tv[g]] += (double)x[b] * p;
                }
            }
        }
}

/* ------------------------------------------------------------------------ */
/* Counter-based SPECIFICATION of the product's gamma / eta Dirichlet draws  */
/* (dirichlet_kernel).  Not a reference function: the reference calls        */
/* numpy's RandomState.dirichlet (HaploSNP_Sampler.py:263-281), whose serial */
/* stream cannot be replayed in parallel.  Same law, restated so that tests  */
/* can compare the kernel value by value:                                    */
/*   variate id  vid = s*G + g  (gamma[s,g]),  S*G + a*4 + b  (eta[a,b])     */
/*   shape       alpha + sum_mu[s,g]      /  delta + esum[b,a]  (:266,:281:  */
/*               eta row a = true base a uses the COLUMN a of E[obs,true])   */
/*   gamma variate: Marsaglia & Tsang (2000), d = a' - 1/3, a' = shape (+1   */
/*               if shape < 1), normals by Box-Muller (cos branch), attempt  */
/*               t uses Philox4x32-10 counters (vid, 2t, iter, 'DIRI') ->    */
/*               u1,u2 and (vid, 2t+1, iter, 'DIRI') -> u3, u_boost; the     */
/*               shape<1 boost multiplies by u_boost^(1/shape) with the      */
/*               u_boost of the ACCEPTED attempt                             */
/*   uniform     ((w0>>5)*2^26 + (w1>>6) + 0.5) / 2^53                       */
/*   row tail    x = y/sum(y); gamma only: x<eps -> eps, x /= sum (:271-273) */
/*   rowprior    lgc + sum (a-1) log x   (Desman_Utils.py:35-44)             */
/* Row sums use the 64-lane xor butterfly (offsets 32,16,..,1) of the kernel.*/
/* libm here vs the device's log/cos/pow: values agree to a few ulp, not     */
/* bit for bit -- tests compare to 1e-13 relative.                           */
/* ------------------------------------------------------------------------ */
#define ORC_STREAM_DIRI 0x44495249u   /* 'DIRI' */

static double orc_u01_open(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6) + 0.5) * (1.0 / 9007199254740992.0);
}

static double orc_gamma_variate(double shape, uint32_t vid, uint32_t iter, const uint32_t key[2])
{
    const double a = shape < 1.0 ? shape + 1.0 : shape;
    const double d = a - 1.0 / 3.0;
    const double c = 1.0 / sqrt(9.0 * d);
    double res = 0.0, uboost = 1.0;
    for (uint32_t t = 0; t < 4096u; t++) {
        uint32_t c0[4] = { vid, 2u * t, iter, ORC_STREAM_DIRI }, c1[4] = { vid, 2u * t + 1u, iter, ORC_STREAM_DIRI };
        uint32_t r0[4], r1[4];
        orc_philox4x32_10(c0, key, r0);
        orc_philox4x32_10(c1, key, r1);
        const double u1 = orc_u01_open(r0[0], r0[1]), u2 = orc_u01_open(r0[2], r0[3]);
        const double u3 = orc_u01_open(r1[0], r1[1]);
        uboost = orc_u01_open(r1[2], r1[3]);
        const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        const double x2 = x * x;
        if (u3 < 1.0 - 0.0331 * x2 * x2 || log(u3) < 0.5 * x2 + d * (1.0 - v + log(v))) { res = d * v; break; }
    }
    if (shape < 1.0) res *= pow(uboost, 1.0 / shape);
    return res;
}

static double butterfly64_sum(const double *lane_vals)
{
    double x[64], y[64];
    memcpy(x, lane_vals, sizeof x);
    for (int off = 32; off >= 1; off >>= 1) {
        for (int l = 0; l < 64; l++) y[l] = x[l] + x[l ^ off];
        memcpy(x, y, sizeof x);
    }
    return x[0];
}

/* sum_mu [S,G], esum [4,4] = E[observed][true]; gamma_out [S,G], eta_out [4,4] (rows = true base),
 * rowprior_out [S+4] (may be NULL).  G <= 64. */
void orc_dirichlet_counter(const uint64_t *sum_mu, const uint64_t *esum, int S, int G,
                           double alpha, double delta, double epsilon, uint64_t seed, uint32_t iter,
                           double *gamma_out, double *eta_out, double *rowprior_out)
{
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    const double lgc_gamma = lgamma(alpha * G) - G * lgamma(alpha);
    const double lgc_eta = lgamma(delta * 4) - 4 * lgamma(delta);
    for (int row = 0; row < S + 4; row++) {
        const int is_gamma = row < S, n = is_gamma ? G : 4;
        double y[64], x[64], t[64];
        memset(y, 0, sizeof y);
        for (int l = 0; l < n; l++) {
            uint32_t vid; double shape;
            if (is_gamma) { vid = (uint32_t)(row * G + l); shape = alpha + (double)sum_mu[vid]; }
            else { int a = row - S; vid = (uint32_t)(S * G + a * 4 + l); shape = delta + (double)esum[l * 4 + a]; }
            y[l] = orc_gamma_variate(shape, vid, iter, key);
        }
        double tot = butterfly64_sum(y);
        for (int l = 0; l < 64; l++) x[l] = y[l] / tot;
        if (is_gamma) {
            for (int l = 0; l < 64; l++) { if (l < n && x[l] < epsilon) x[l] = epsilon; t[l] = l < n ? x[l] : 0.0; }
            tot = butterfly64_sum(t);
            for (int l = 0; l < 64; l++) x[l] = x[l] / tot;
        }
        const double a1 = is_gamma ? alpha : delta;
        for (int l = 0; l < 64; l++) t[l] = l < n ? (a1 - 1.0) * log(x[l]) : 0.0;
        const double lsum = butterfly64_sum(t);
        if (rowprior_out) rowprior_out[row] = (is_gamma ? lgc_gamma : lgc_eta) + lsum;
        for (int l = 0; l < n; l++) {
            if (is_gamma) gamma_out[row * G + l] = x[l];
            else eta_out[(row - S) * 4 + l] = x[l];
        }
    }
}

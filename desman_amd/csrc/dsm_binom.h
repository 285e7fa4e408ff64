// dsm_binom.h -- device-side samplers of the aggregated mu/E pass (spec v2, restated in
// oracle/stats_agg.c: every function here has a twin there with the same operation order; the library
// is built with -ffp-contract=off, division and sqrt are IEEE, so the two agree bit for bit).
//
//   draw_reads<K>   x reads over K categories, read by read against 32-bit thresholds (x <= DSM_XS)
//   binom_small     Binomial by sequential-search inversion on the rarer outcome, chunked so that a
//                   chunk's mean is <= 16; (1-q)^c by repeated squaring: no transcendental function
//   binom_big       the same below mean 16, Hoermann's BTRS (1993) above (counts up to 2^32-1)
//   mult4           x reads over the four true bases: heaviest base peeled off by one binomial
#pragma once
#include "dsm_device.h"

#define DSM_STREAM_STA1 0x53544131u   // 'STA1'  stage-1 cell streams
#define DSM_STREAM_STA2 0x53544132u   // 'STA2'  stage-2 binomial streams
#define DSM_STREAM_TEST 0x54455354u   // 'TEST'  test hook
#define DSM_XS 12u                    // counts up to DSM_XS are drawn read by read
#define DSM_BINV_MEAN_CAP 16.0
#define DSM_RCP_TAB_N 64              // 1/k for k < 64, staged in LDS by the kernels (entry 0 unused)

__device__ __forceinline__ Xo128 xo_seed(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    uint32_t w[4];
    philox4x32_10(c0, c1, c2, c3, k0, k1, w);
    Xo128 r{w[0], w[1], w[2], w[3]};
    if ((r.s0 | r.s1 | r.s2 | r.s3) == 0u) r.s0 = 1u;
    return r;
}

__device__ __forceinline__ double xo_u01(Xo128 &r)
{
    const uint32_t a = r.next();
    const uint32_t b = r.next();
    return u01_open(a, b);
}

// v >= 0: floor + clamp to 2^32-1 (NaN -> 0) is exactly v_cvt_u32_f64
__device__ __forceinline__ uint32_t cvt_sat_u32(double v)
{
    uint32_t q;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(q) : "v"(v));
    return q;
}

__device__ __forceinline__ double dsm_pw(double b, uint32_t e)
{
    double res = 1.0;
    while (e) {
        if (e & 1u) res = res * b;
        e >>= 1;
        if (e) b = b * b;
    }
    return res;
}

template <int K>
__device__ __forceinline__ void draw_reads(Xo128 &rng, uint32_t x, const double (&w)[K], uint32_t (&n)[K])
{
    double cums[K], cum = 0.0;
#pragma unroll
    for (int j = 0; j < K; ++j) { cum = cum + w[j]; cums[j] = cum; }
    const double scale = 4294967296.0 / cums[K - 1];
    uint32_t t[K - 1], c[K - 1];
#pragma unroll
    for (int j = 0; j < K - 1; ++j) { t[j] = cvt_sat_u32(cums[j] * scale); c[j] = 0; }
    for (uint32_t i = 0; i < x; ++i) {
        const uint32_t r = rng.next();
#pragma unroll
        for (int j = 0; j < K - 1; ++j) c[j] += (r < t[j]) ? 1u : 0u;
    }
    n[0] = c[0];
#pragma unroll
    for (int j = 1; j < K - 1; ++j) n[j] = c[j] - c[j - 1];
    n[K - 1] = x - c[K - 2];
}

// rcp: LDS table of the correctly rounded 1/k, k < DSM_RCP_TAB_N (the oracle divides)
__device__ __forceinline__ uint32_t binv_chunk(Xo128 &rng, uint32_t c, double f0, double r, const double *__restrict__ rcp)
{
    double u = xo_u01(rng), f = f0;
    uint32_t k = 0;
    while (u >= f && k < c) {
        u = u - f;
        k = k + 1;
        const double inv = (k < DSM_RCP_TAB_N) ? rcp[k] : 1.0 / (double)k;
        f = (f * (r * (double)(c - k + 1))) * inv;
    }
    return k;
}

__device__ __forceinline__ uint32_t binom_inv(Xo128 &rng, uint32_t n, double ws, double wl, const double *__restrict__ rcp)
{
    const double T = ws + wl;
    const double omq = wl / T;
    const double r = ws / wl;
    uint32_t cap = n;
    if ((double)n * ws > DSM_BINV_MEAN_CAP * T) {
        const double capd = floor(DSM_BINV_MEAN_CAP * T / ws);
        cap = capd >= (double)n ? n : (uint32_t)capd;
    }
    uint32_t total = 0, left = n;
    double f_full = 0.0;
    if (left >= cap) f_full = dsm_pw(omq, cap);
    while (left > 0) {
        const uint32_t c = left < cap ? left : cap;
        const double f0 = (c == cap) ? f_full : dsm_pw(omq, c);
        total += binv_chunk(rng, c, f0, r, rcp);
        left -= c;
    }
    return total;
}

__device__ __forceinline__ uint32_t binom_small(Xo128 &rng, uint32_t n, double wa, double wb, const double *__restrict__ rcp)
{
    if (n == 0 || !(wa > 0.0)) return 0;
    if (!(wb > 0.0)) return n;
    const bool flip = wa > wb;
    const double ws = flip ? wb : wa, wl = flip ? wa : wb;
    const uint32_t k = binom_inv(rng, n, ws, wl, rcp);
    return flip ? n - k : k;
}

__device__ __forceinline__ double stirling_tail(double k)
{
    if (k <= 9.0) {
        const int i = (int)k;
        // ln k! - Stirling for k = 0..9 (Hoermann 1993, fc); a select chain keeps it out of scratch memory
        return i == 0 ? 0.0810614667953272 : i == 1 ? 0.0413406959554092 : i == 2 ? 0.0276779256849983
             : i == 3 ? 0.02079067210376509 : i == 4 ? 0.0166446911898211 : i == 5 ? 0.0138761288230707
             : i == 6 ? 0.0118967099458917 : i == 7 ? 0.0104112652619720 : i == 8 ? 0.00925546218271273
             : 0.00833056343336287;
    }
    const double kp1 = k + 1.0, kp1sq = kp1 * kp1;
    return (1.0 / 12.0 - (1.0 / 360.0 - (1.0 / 1260.0) / kp1sq) / kp1sq) / kp1;
}

// q <= 1/2, n q > 16
__device__ __noinline__ uint32_t btrs(Xo128 &rng, uint32_t n, double q, const double2 *__restrict__ ltab)
{
    const double nd = (double)n;
    const double spq = sqrt(nd * q * (1.0 - q));
    const double b = 1.15 + 2.53 * spq;
    const double a = -0.0873 + 0.0248 * b + 0.01 * q;
    const double c = nd * q + 0.5;
    const double v_r = 0.92 - 4.2 / b;
    const double r = q / (1.0 - q);
    const double alpha = (2.83 + 5.1 / b) * spq;
    const double m = floor((nd + 1.0) * q);
    for (int attempt = 0; attempt < 4096; ++attempt) {
        const double u = xo_u01(rng) - 0.5;
        const double v = xo_u01(rng);
        const double us = 0.5 - fabs(u);
        const double kd = floor((2.0 * a / us + b) * u + c);
        if (kd < 0.0 || kd > nd) continue;
        if (us >= 0.07 && v <= v_r) return (uint32_t)kd;
        const double lv = dsm_log(v * alpha / (a / (us * us) + b), ltab);
        const double ub = (m + 0.5) * dsm_log((m + 1.0) / (r * (nd - m + 1.0)), ltab)
                          + (nd + 1.0) * dsm_log((nd - m + 1.0) / (nd - kd + 1.0), ltab)
                          + (kd + 0.5) * dsm_log(r * (nd - kd + 1.0) / (kd + 1.0), ltab)
                          + stirling_tail(m) + stirling_tail(nd - m) - stirling_tail(kd) - stirling_tail(nd - kd);
        if (lv <= ub) return (uint32_t)kd;
    }
    return (uint32_t)m;
}

__device__ __forceinline__ uint32_t binom_big(Xo128 &rng, uint32_t n, double wa, double wb, const double *__restrict__ rcp,
                                              const double2 *__restrict__ ltab)
{
    if (n == 0 || !(wa > 0.0)) return 0;
    if (!(wb > 0.0)) return n;
    const bool flip = wa > wb;
    const double ws = flip ? wb : wa, wl = flip ? wa : wb;
    const double T = ws + wl;
    uint32_t k;
    if ((double)n * ws > DSM_BINV_MEAN_CAP * T) k = btrs(rng, n, ws / T, ltab);
    else k = binom_inv(rng, n, ws, wl, rcp);
    return flip ? n - k : k;
}

__device__ __forceinline__ void mult4(Xo128 &rng, uint32_t x, const double (&W)[4], uint32_t (&n)[4], const double *__restrict__ rcp)
{
    n[0] = n[1] = n[2] = n[3] = 0;
    if (x == 0) return;
    if (x <= DSM_XS) { draw_reads<4>(rng, x, W, n); return; }
    int am = 0;
    double wm = W[0];
#pragma unroll
    for (int a = 1; a < 4; ++a) if (W[a] > wm) { wm = W[a]; am = a; }
    // the three other bases in ascending order: index j skips am
    double wo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int a_lo = j, a_hi = j + 1;                  // candidate indices: j if j < am else j + 1
        wo[j] = (j < am) ? W[a_lo] : W[a_hi];
    }
    const double ws = (wo[0] + wo[1]) + wo[2];
    const uint32_t m = binom_small(rng, x, ws, wm, rcp);
    uint32_t k[3] = {0, 0, 0};
    if (m != 0) {
        if (m <= DSM_XS) draw_reads<3>(rng, m, wo, k);
        else {
            k[0] = binom_small(rng, m, wo[0], wo[1] + wo[2], rcp);
            k[1] = binom_small(rng, m - k[0], wo[1], wo[2], rcp);
            k[2] = m - k[0] - k[1];
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int j = (a < am) ? a : a - 1;                 // position of a among the others
        const uint32_t kj = (j == 0) ? k[0] : (j == 1) ? k[1] : k[2];
        n[a] = (a == am) ? x - m : kj;
    }
}

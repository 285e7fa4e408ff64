// dsm_binom.h -- device-side samplers of the aggregated mu/E pass (specs 2 and 3, restated in
// oracle/stats_agg.c: every function here has a twin there with the same operation order; the library
// is built with -ffp-contract=off, division and sqrt are IEEE, so the two agree bit for bit).
// SPEC = 2 (the default, DSM_STATS_AGG): (1-q)^n by repeated squaring, two divisions, item streams three Philox rounds off the cell's block.
// SPEC = 3 (selectable: dsm_ctx_force_stats_spec(3), DESMAN_HIP_STATS_SPEC=3; measured 2 us slower at config 3): f0 = exp(-n ln(1 + r)), r = q/(1-q) the one division, table log + table exp (dsm_texp) -- a fixed
//           ~35 instructions instead of a loop over the bits of n to the deepest lane; item streams two rounds.
//
//   draw_reads<K>   x reads over K categories, read by read against 32-bit thresholds (x <= DSM_XS)
//   binom           Binomial(n, p): sequential-search inversion on the rarer outcome while its mean is <= 128
//                   ((1-q)^n by repeated squaring: no transcendental function), Hoermann's BTRS (1993) above
//   mult4           x reads over the four true bases: heaviest base peeled off by one binomial, the few
//                   others read by read
#pragma once
#include "dsm_device.h"
#include "log_table.h"

#define DSM_STREAM_STA1 0x53544131u   // 'STA1'  stage-1 cell streams
#define DSM_STREAM_STA2 0x53544132u   // 'STA2'  stage-2 binomial streams
#define DSM_STREAM_TEST 0x54455354u   // 'TEST'  test hook
#define DSM_XS 128u                   // up to DSM_XS non-dominant reads are drawn read by read
#define DSM_BINV_MEAN_CAP 128.0       // inversion while the mean of the rarer outcome is <= 128, BTRS above: on a
                                      // 64-lane wavefront BTRS costs ~1500 instructions (some lane always takes the
                                      // slow path / another attempt), the search 14 per step
#define DSM_LEAN_CAP 64.0             // stage 1 itself draws the items up to this mean (kernels_stats.hip: lean_cap); not part of the specification
#define DSM_BINV_MEAN_CAP_S2 16.0      // stage 2: one latency-bound binomial per lane, BTRS is the shorter dependent chain
#define DSM_BINV_KMAX 255u
#define DSM_RCP_TAB_N 256             // 1/k for k < 256, staged in LDS by the kernels (entry 0 unused)

__device__ __forceinline__ Xo128 xo_seed(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    uint32_t w[4];
    philox4x32_10(c0, c1, c2, c3, k0, k1, w);
    Xo128 r{w[0], w[1], w[2], w[3]};
    if ((r.s0 | r.s1 | r.s2 | r.s3) == 0u) r.s0 = 1u;
    return r;
}

// one Philox4x32 round
__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0, uint32_t k1)
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
}

// stream of the item (cell, observed base b): base = Philox4x32-10({cell, 0, iter, 'STA1'}) is shared by the four items
// of the cell; three more rounds keyed by the base make the item's xoshiro state (oracle: item_seed)
template <int SPEC>
__device__ __forceinline__ Xo128 item_seed(const uint32_t (&base)[4], uint32_t b, uint32_t key0, uint32_t key1)
{
    uint32_t c0 = base[0], c1 = base[1], c2 = base[2], c3 = base[3];
    uint32_t k0 = key0 ^ (0x9E3779B9u * (b + 1u)), k1 = key1 ^ (0xBB67AE85u * (b + 1u));
#pragma unroll
    for (int i = 0; i < (SPEC >= 3 ? 2 : 3); ++i) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    Xo128 r{c0, c1, c2, c3};
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

// b^e by repeated squaring (oracle: pw).  The loop runs to the longest exponent of the wavefront on a scalar condition; a
// lane that is done keeps squaring a base it no longer uses -- the products it did use are the oracle's, bit for bit
__device__ __forceinline__ double dsm_pw(double b, uint32_t e)
{
    double res = 1.0;
    while (__builtin_amdgcn_ballot_w64(e != 0u) != 0ull) {
        const double rb = res * b;
        res = (e & 1u) ? rb : res;
        e >>= 1;
        b = b * b;
    }
    return res;
}

// exp(y), y <= 0 (oracle: orc_texp): y = (32 k + j) ln2/32 + r, exp(r) - 1 by a degree-6 polynomial, 2^(j/32) from the
// 32-entry table `etab` (LDS copy of dsm_exp_table_host), 2^k by v_ldexp_f64
__device__ __forceinline__ double dsm_texp(double y, const double *__restrict__ etab)
{
    const double kd = __builtin_rint(y * DSM_EXP_INV_LN2_32);
    const int ki = (int)kd;
    double r = fma(kd, -DSM_EXP_LN2_32_HI, y);
    r = fma(kd, -DSM_EXP_LN2_32_LO, r);
    double p = fma(r, 1.0 / 720.0, 1.0 / 120.0);
    p = fma(r, p, 1.0 / 24.0);
    p = fma(r, p, 1.0 / 6.0);
    p = fma(r, p, 0.5);
    p = fma(r, p, 1.0);
    p = r * p;
    const double t = etab[ki & 31];
    const double v = ldexp(fma(t, p, t), ki >> 5);
    return (y > -700.0) ? v : 0.0;
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

// Binomial(c, q) by sequential search from 0 (c q <= 128): f0 = (1-q)^c, r = q/(1-q); the search stops at
// DSM_BINV_KMAX = 255, so 1/k always comes from the LDS table rcp (correctly rounded 1/k: the oracle divides)
__device__ __forceinline__ uint32_t binv(Xo128 &rng, uint32_t c, double f0, double r, const double *__restrict__ rcp)
{
    // P(k)/P(k-1) = r (c-k+1)/k = r (c+1) (1/k) - r: one fma and one multiply per step
    double u = xo_u01(rng), f = f0;
    const double rc1 = r * ((double)c + 1.0);
    uint32_t k = 0;
    const uint32_t kend = c < DSM_BINV_KMAX ? c : DSM_BINV_KMAX;
    while (u >= f && k < kend) {
        u = u - f;
        k = k + 1;
        f = f * fma(rc1, rcp[k], -r);
    }
    return k;
}

// the same search with nothing but one multiply on the loop-carried path.  A lane that is still searching after t steps has
// k = t, so 1/k comes from a wave-uniform index and the ratio P(t+1)/P(t) does not depend on the lane's progress; a lane that
// has stopped stays stopped (`go` is sticky), so its u and f may run on as garbage -- only k, counted while `go` holds, is
// kept.  The wavefront tests for the end once in four steps.  Same operations in the same order on every lane that is
// still searching, hence the same k as binv().  A step of binv() on a wavefront that runs alone on its SIMD (the compacted
// stage-1 kernel, stage 2) is ~170 cycles of dependent latency: compare -> scalar mask -> exec -> branch -> LDS -> fma -> mul.
__device__ __forceinline__ uint32_t binv_chain(Xo128 &rng, uint32_t c, double f0, double r, const double *__restrict__ rcp)
{
    double u = xo_u01(rng), f = f0;
    const double rc1 = r * ((double)c + 1.0);
    uint32_t k = 0, t = 0;
    const uint32_t kend = c < DSM_BINV_KMAX ? c : DSM_BINV_KMAX;
    bool go = true;
    do {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            go = go && u >= f && t < kend;
            k += go ? 1u : 0u;
            u = u - f;
            f = f * fma(rc1, rcp[(t + 1u) & (DSM_RCP_TAB_N - 1)], -r);
            t = t + 1u;
        }
    } while (__builtin_amdgcn_ballot_w64(go) != 0ull);
    return k;
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
    const double inv = 1.0 / (k + 1.0), inv2 = inv * inv;
    return (1.0 / 12.0 - (1.0 / 360.0 - (1.0 / 1260.0) * inv2) * inv2) * inv;
}

// Hoermann's BTRS, q <= 1/2, n q > 128 (the rare path of the stage-1 kernel once the chain has converged)
__device__ __forceinline__ uint32_t btrs(uint32_t &s0, uint32_t &s1, uint32_t &s2, uint32_t &s3, uint32_t n, double q,
                                      const double2 *__restrict__ ltab)
{
    Xo128 rng{s0, s1, s2, s3};
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
    const double h = (m + 0.5) * dsm_log_core((m + 1.0) / (r * nm1), ltab) + stirling_tail(m) + stirling_tail(nd - m);
    uint32_t res = (uint32_t)m;
    for (int attempt = 0; attempt < 4096; ++attempt) {
        const double u = xo_u01(rng) - 0.5;
        const double v = xo_u01(rng);
        const double us = 0.5 - fabs(u);
        const double kd = floor((2.0 * a / us + b) * u + c);
        if (kd < 0.0 || kd > nd) continue;
        if (us >= 0.07 && v <= v_r) { res = (uint32_t)kd; break; }
        const double nk1 = nd - kd + 1.0;
        const double lv = dsm_log_core(v * alpha, ltab) - dsm_log_core(a / (us * us) + b, ltab);     // arguments are positive normal numbers
        const double ub = h + (nd + 1.0) * dsm_log_core(nm1 / nk1, ltab) + (kd + 0.5) * dsm_log_core(r * nk1 / (kd + 1.0), ltab)
                          - stirling_tail(kd) - stirling_tail(nd - kd);
        if (lv <= ub) { res = (uint32_t)kd; break; }
    }
    s0 = rng.s0; s1 = rng.s1; s2 = rng.s2; s3 = rng.s3;
    return res;
}

// successes among n trials with success : failure odds wa : wb.  BIG = false is the inversion-only form of the
// lean stage-1 kernel: a draw that needs BTRS sets `defer` instead (the item is re-done by the compacted kernel).
// ltab / etab: LDS copies of the log and exp tables (etab = ltab + DSM_LOG_TAB_N doubles further: one staged block)
template <bool BIG, int SPEC>
__device__ __forceinline__ uint32_t binom(Xo128 &rng, uint32_t n, double wa, double wb, const double *__restrict__ rcp,
                                          const double2 *__restrict__ ltab, bool &defer, double cap = DSM_BINV_MEAN_CAP)
{
    if (n == 0 || !(wa > 0.0)) return 0;
    if (!(wb > 0.0)) return n;
    const bool flip = wa > wb;
    const double ws = flip ? wb : wa, wl = flip ? wa : wb;         // the rarer outcome has odds ws : wl
    const double T = ws + wl;
    uint32_t k;
    if ((double)n * ws > cap * T) {
        if constexpr (BIG) k = btrs(rng.s0, rng.s1, rng.s2, rng.s3, n, ws / T, ltab);
        else { defer = true; return 0; }
    } else {
        double f0, r;
        if constexpr (SPEC >= 3) {
            r = ws / wl;                                                                   // q / (1 - q); 1 - q = 1 / (1 + r)
            f0 = dsm_texp(-((double)n * dsm_log_core(1.0 + r, ltab)), reinterpret_cast<const double *>(ltab + DSM_LOG_TAB_N));
        } else { f0 = dsm_pw(wl / T, n); r = ws / wl; }
        if constexpr (BIG) k = binv_chain(rng, n, f0, r, rcp);                             // few, lonely wavefronts: latency
        else k = binv(rng, n, f0, r, rcp);                                                 // stage 1: throughput (48 vs 50 us with binv_chain)
    }
    return flip ? n - k : k;
}

template <bool BIG, int SPEC>
__device__ __forceinline__ void mult4(Xo128 &rng, uint32_t x, const double (&W)[4], uint32_t (&n)[4], const double *__restrict__ rcp,
                                      const double2 *__restrict__ ltab, bool &defer, double lean_cap = DSM_BINV_MEAN_CAP,
                                      int *kind = nullptr)
{
    n[0] = n[1] = n[2] = n[3] = 0;
    if (x == 0) return;
    int am = 0;
    double wm = W[0];
#pragma unroll
    for (int a = 1; a < 4; ++a) if (W[a] > wm) { wm = W[a]; am = a; }
    // the three other bases in ascending order: position j holds base j (j < am) or j + 1
    double wo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) wo[j] = (j < am) ? W[j] : W[j + 1];
    const double ws = (wo[0] + wo[1]) + wo[2];
    // step 0 splits off the reads that are not of the heaviest base (m of them); up to DSM_XS of them are drawn read
    // by read, more (only possible for the items of the compacted kernel, in practice) by two more binomials
    uint32_t m = 0, k[3] = {0, 0, 0};
    if constexpr (!BIG) {
        m = binom<false, SPEC>(rng, x, ws, wm, rcp, ltab, defer, lean_cap);   // lean_cap <= the cap of the specification: who draws, not what
        if (defer) {                                        // what the compacted kernel will run for it: BTRS or a long search
            const double wsm = ws < wm ? ws : wm;
            if (kind) *kind = ((double)x * wsm > DSM_BINV_MEAN_CAP * (ws + wm)) ? 0 : 1;
            return;
        }
        if (m > DSM_XS) { defer = true; if (kind) *kind = 2; return; }   // ... or a search and two more binomials
    } else {
        uint32_t nn = x;
        double wa = ws, wb = wm;
        int nst = 1;
#pragma unroll 1
        for (int st = 0; st < nst; ++st) {                  // ONE binomial site (the sampler is a large piece of code)
            const uint32_t kk = binom<true, SPEC>(rng, nn, wa, wb, rcp, ltab, defer);
            if (st == 0) {
                m = kk;
                if (m > DSM_XS) { nst = 3; nn = m; wa = wo[0]; wb = wo[1] + wo[2]; }
            } else if (st == 1) { k[0] = kk; nn = m - kk; wa = wo[1]; wb = wo[2]; }
            else { k[1] = kk; k[2] = m - k[0] - kk; }
        }
    }
    if (m != 0 && m <= DSM_XS) draw_reads<3>(rng, m, wo, k);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int j = (a < am) ? a : a - 1;                 // position of a among the others
        const uint32_t kj = (j == 0) ? k[0] : (j == 1) ? k[1] : k[2];
        n[a] = (a == am) ? x - m : kj;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 6: the item of the lean stage-1 kernel as ONE straight piece of code (the same draws as mult4<false, 2>, bit for bit --
// same operations on the same operands in the same order; what changes is how many instructions the wavefront issues for them):
//   * the quotients by the unscaled expansion (div_unscaled: v_rcp_f64, two Newton steps, quotient, residual, correction -- the
//     compiler's IEEE expansion without v_div_scale x 2 / v_div_fmas / v_div_fixup, which are identities while the operands and
//     the quotient are far from the ends of the exponent range) behind ONE wave-uniform range test per item; any lane outside
//     [2^-400, 2^400] sends the wavefront's item through the IEEE divisions;
//   * (1-q)^n with the exponent bit-reversed once: the bit test is a sign compare, one instruction less per squaring;
//   * the read-by-read loop counts up against a scalar index, the generator's step is written with three-input xors
//     (v_bitop3_b32), the uniform is two fused operations, the search's table pointer is its only counter;
//   * the rare outcomes (hand-over to the compacted kernel) leave the item by ONE flag the caller tests once per cell.
// ISA-level counts per phase: profiles/r06_stats_isa_counts.txt (scripts/isa_count.py).

__device__ __forceinline__ double div_unscaled(double a, double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    const double q = a * y;
    const double r = fma(-b, q, a);
    return fma(r, y, q);
}
// a / b; `fast` is wave-uniform: every active lane's operands (and hence the quotient's exponent) are in the safe range
__device__ __forceinline__ double sdiv(double a, double b, bool fast) { return fast ? div_unscaled(a, b) : a / b; }

// a ^= b ^ c in one instruction (gfx950: v_bitop3_b32 with the truth table of the three-input xor), in place
__device__ __forceinline__ void xor3_into(uint32_t &a, uint32_t b, uint32_t c)
{
    asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a) : "v"(b), "v"(c));
}
// xoshiro128+ step (Xo128::next) in seven instructions, every word updated in its own register (no copies at a loop's back edge):
// s3 ^= s1 first (the result and t = s1 << 9 are taken before), then s1 ^= s2 ^ s0 and s2 ^= s0 ^ t on the OLD s2, s0, then
// s0 ^= s3 (the new s3 is the old s3 ^ s1), then the rotation
__device__ __forceinline__ uint32_t xo_next7(uint32_t &s0, uint32_t &s1, uint32_t &s2, uint32_t &s3)
{
    const uint32_t res = s0 + s3;
    const uint32_t t = s1 << 9;
    s3 ^= s1;
    xor3_into(s1, s2, s0);
    xor3_into(s2, s0, t);
    s0 ^= s3;
    s3 = __builtin_amdgcn_alignbit(s3, s3, 21);  // rotl 11
    return res;
}
// u01_open(a, b) in two fused operations: (a >> 5) 2^26 + (b >> 6) is exact in either form (53 bits), and (m + 0.5) 2^-53 is m 2^-53 + 2^-54
// rounded once -- the rounding of m + 0.5 scaled by an exact power of two.  Same bits as u01_open.
__device__ __forceinline__ double u01_open_fused(uint32_t a, uint32_t b)
{
    const double m = fma((double)(a >> 5), 67108864.0, (double)(b >> 6));
    return fma(m, 0x1p-53, 0x1p-54);
}

// One (cell, observed base) item of the lean kernel.  x > 0 on the lanes that call it (the caller's exec mask).  Returns false
// when the item is handed over (kind: 0 rejection sampler, 1 long search, 2 search + two more binomials); n[] is the draw otherwise.
template <int SPEC>
__device__ __forceinline__ bool s1_item(Xo128 rng, uint32_t x, const double (&W)[4], uint32_t (&n)[4], const double *__restrict__ rcp,
                                        const double2 *__restrict__ ltab, double lean_cap, int &kind)
{
    static_assert(SPEC == 2, "the lean item is spec 2's");
    ISA_MARK("item_argmax");
    int am = 0;
    double wm = W[0];
#pragma unroll
    for (int a = 1; a < 4; ++a) if (W[a] > wm) { wm = W[a]; am = a; }
    double wo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) wo[j] = (j < am) ? W[j] : W[j + 1];
    const double c01 = wo[0] + wo[1];
    const double ws4 = c01 + wo[2];
    uint32_t m = 0, k0 = 0, k1 = 0;                       // reads not of the heaviest base; of the first / first two others
    uint32_t s0 = rng.s0, s1 = rng.s1, s2 = rng.s2, s3 = rng.s3;
    kind = 0;
    bool ok = true, fast = false;
    // binom(x, wa = ws4, wb = wm) of dsm_binom.h: !(wa > 0) -> 0, !(wb > 0) -> x
    if (ws4 > 0.0) {
        if (!(wm > 0.0)) m = x;
        else {
            ISA_MARK("item_binom_setup");
            const bool flip = ws4 > wm;
            const double ws = flip ? wm : ws4, wl = flip ? ws4 : wm;
            const double T = ws + wl;
            const double xd = (double)x;
            if (xd * ws > lean_cap * T) { ok = false; kind = (xd * ws > DSM_BINV_MEAN_CAP * T) ? 0 : 1; }
            else {
                // one range test for the item's three divisions (operands ws <= wl <= T <= 2 wl, and 2^32 / ws4 with ws4 one of ws, wl)
                fast = (__builtin_amdgcn_ballot_w64(!(ws >= 0x1p-400)) | __builtin_amdgcn_ballot_w64(!(wl <= 0x1p400))) == 0ull;
                ISA_MARK("item_div1");
                double b = sdiv(wl, T, fast);
                ISA_MARK("item_pw");
                // (1-q)^x by repeated squaring, lowest bit first (oracle: pw); the exponent is kept bit-reversed so that the bit
                // of the step is the sign bit.  A lane that is done keeps squaring a base it no longer uses.
                double f = 1.0;
                uint32_t er = __builtin_bitreverse32(x);
                do {
                    const double fb = f * b;
                    f = ((int32_t)er < 0) ? fb : f;
                    er <<= 1;
                    b = b * b;
                } while (__builtin_amdgcn_ballot_w64(er != 0u) != 0ull);
                ISA_MARK("item_u01");
                const uint32_t ua = xo_next7(s0, s1, s2, s3);
                const uint32_t ub = xo_next7(s0, s1, s2, s3);
                double u = u01_open_fused(ua, ub);
                uint32_t k = 0;
                if (u >= f) {
                    ISA_MARK("item_div2");
                    const double r = sdiv(ws, wl, fast);
                    const double rc1 = r * (xd + 1.0);
                    const uint32_t kend = x < DSM_BINV_KMAX ? x : DSM_BINV_KMAX;
                    ISA_MARK("item_binv");
                    // sequential search (oracle: binv); a lane still searching after t steps has k = t: the table pointer is the
                    // loop's only vector counter (one add per step; k is read off it after the search).  The oracle's other exit, k = kend
                    // = min(x, 255), is taken when u is above the sum of every term the search computed -- all x reads the rarer outcome,
                    // or rounding at the far end of the sum: a 1e-16 event -- so the loop tests it on a scalar counter against 255 and the
                    // lane's own end is applied after the loop: a lane that ran past it did so on terms that are zero (or rounding
                    // noise), and is set back to kend, the oracle's answer.
                    const double *pk = rcp + __builtin_amdgcn_mbcnt_lo(0u, 0u);      // (+ 0 the compiler cannot see: the pointer stays a vector register)
                    const double *const pk0 = pk;
                    uint32_t t = 0;
                    do {
                        u = u - f;
                        ++pk;
                        f = f * fma(rc1, *pk, -r);
                        t = __builtin_amdgcn_readfirstlane(t + 1u);
                    } while (u >= f && t < DSM_BINV_KMAX);
                    k = (uint32_t)(pk - pk0);
                    k = k < kend ? k : kend;
                    ISA_MARK("item_binv_end");
                }
                m = flip ? x - k : k;
            }
        }
        if (ok) {
            if (m > DSM_XS) { ok = false; kind = 2; }
            else if (m != 0) {
                ISA_MARK("item_reads_setup");
                // the m other reads, read by read against 32-bit thresholds (oracle: draw_reads, K = 3; its cums[2] is ws4)
                const double scale = sdiv(4294967296.0, ws4, fast);
                const uint32_t t0 = cvt_sat_u32(wo[0] * scale), t1 = cvt_sat_u32(c01 * scale);
                ISA_MARK("item_reads");
                uint32_t i = 0;                                     // the same on every lane that is still drawing: a scalar register
                do {
                    const uint32_t w = xo_next7(s0, s1, s2, s3);
                    k0 += (w < t0) ? 1u : 0u;
                    k1 += (w < t1) ? 1u : 0u;
                    i = __builtin_amdgcn_readfirstlane(i + 1u);
                } while (i < m);
                ISA_MARK("item_reads_end");
            }
        }
    }
    ISA_MARK("item_map");
    // n[a]: the heaviest base keeps x - m; the others, in ascending order, k0, k1 - k0, m - k1
    const uint32_t o0 = k0, o1 = k1 - k0, o2 = m - k1, nh = x - m;
    n[0] = (am == 0) ? nh : o0;
    n[1] = (am == 1) ? nh : ((am == 0) ? o0 : o1);
    n[2] = (am == 2) ? nh : ((am == 3) ? o2 : o1);
    n[3] = (am == 3) ? nh : o2;
    return ok;
}

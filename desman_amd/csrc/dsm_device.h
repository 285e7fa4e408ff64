// dsm_device.h -- device-side helpers shared by the gfx950 kernels.
//
// Everything here targets CDNA4 (wave64) directly.  The library is built with
// -ffp-contract=off: every fused multiply-add in the kernels is an explicit
// fma(), so the arithmetic spec is the source text (and matches the CPU
// oracle, which is built the same way).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DSM_STREAM_STATS 0x53544154u   // 'STAT'  mu/E per-read draws
#define DSM_STREAM_DIRI  0x44495249u   // 'DIRI'  gamma / eta Dirichlet draws
#define DSM_STREAM_TAUU  0x54415555u   // 'TAUU'  tau-sweep uniforms (Philox mode)

#define DSM_EPS 2.220446049250313e-16

// phase marks for scripts/isa_count.py: with -DDSM_ISA_MARKS a `; MARK name` comment goes into the assembly at the start of a phase
#ifdef DSM_ISA_MARKS
#define ISA_MARK(name) asm volatile("; MARK " name)
#else
#define ISA_MARK(name) do { } while (0)
#endif

// ---- Philox4x32-10 (Salmon et al. 2011); checked against the Random123 KATs
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
        const uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---- xoshiro128+ (Blackman & Vigna): the per-item stream of the mu/E draw.  The "+" scrambler
// (one add) is the variant recommended when only the upper bits matter: its known weakness is
// low linear complexity of the lowest four bits, and a word is only ever compared with a 32-bit
// threshold here, so those bits decide a draw with probability 2^-28 per comparison.
struct Xo128 {
    uint32_t s0, s1, s2, s3;
    __device__ __forceinline__ uint32_t next()
    {
        const uint32_t res = s0 + s3;
        const uint32_t t = s1 << 9;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3;
        s2 ^= t;
        s3 = (s3 << 11) | (s3 >> 21);
        return res;
    }
};

// uniform double in (0,1) from two 32-bit words (53 random bits, never 0 or 1)
__device__ __forceinline__ double u01_open(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6) + 0.5) * (1.0 / 9007199254740992.0);
}

// ---- wavefront (64-lane) butterflies.  W = lanes per group (16/32/64); the
// xor offsets stay inside the group, and a+b == b+a bitwise, so every lane of
// the group ends with the identical value.
template <int W>
__device__ __forceinline__ double group_allreduce_sum(double x)
{
#pragma unroll
    for (int off = W / 2; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}
template <int W>
__device__ __forceinline__ unsigned group_allreduce_sum_u32(unsigned x)
{
#pragma unroll
    for (int off = W / 2; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// ---- table-driven fp64 natural log (tau sweep / LL).  x = 2^e * m, m in [1,2);
// i = top 8 mantissa bits (256-entry table); r = m * invc_i - 1 (one fma, |r| <= 2^-9);
// log x = e ln2 + logc_i + log1p(r), log1p(r) = r - r^2/2 + r^3/3 - r^4/4 + r^5/5 (truncation r^6/6 < 1e-17).  Absolute error <= ~1 ulp of max(1, |log x|) -- the same
// class as libm in the sums it feeds (validated against glibc on the host).
// `tab` = the 256 x {invc, logc} table (log_table.h) staged in LDS.
// Zero, subnormal, negative, inf and NaN inputs take the libm path.
// dsm_log_core: x must be a positive normal double (callers check with dsm_log_ok).
__device__ __forceinline__ double dsm_log_core(double x, const double2 *__restrict__ tab)
{
    const uint32_t hi = (uint32_t)__double2hiint(x);
    const int e = (int)(hi >> 20) - 1023;
    const double2 t = tab[(hi >> 12) & 255u];
    const double m = __hiloint2double((int)((hi & 0x000fffffu) | 0x3ff00000u), __double2loint(x));
    const double r = fma(m, t.x, -1.0);
    const double ed = (double)e;
    double p = fma(r, 0.2, -0.25);          // |r| <= 2^-9: the r^6/6 term is below 1e-17
    p = fma(r, p, 1.0 / 3.0);
    p = fma(r, p, -0.5);
    const double w = fma(ed, 0x1.62e42fefa3800p-1, t.y);
    const double lo = fma(ed, 0x1.ef35793c76730p-45, (r * r) * p);
    return (w + r) + lo;
}
// true iff x is a positive normal finite double (the domain of dsm_log_core)
__device__ __forceinline__ bool dsm_log_ok(double x)
{
    return ((uint32_t)__double2hiint(x) - 0x00100000u) < 0x7fe00000u;
}
// cold path (zero / subnormal / inf / NaN arguments): kept out of line so that libm's log does not
// inflate the register allocation of the hot loops
__device__ __attribute__((noinline)) double dsm_log_slow(double x) { return log(x); }
__device__ __forceinline__ double dsm_log(double x, const double2 *__restrict__ tab)
{
    if (__builtin_expect(!dsm_log_ok(x), 0)) return dsm_log_slow(x);
    return dsm_log_core(x, tab);
}

// ---- sum of four per-lane values over a W-lane group, result on every lane.
// Transposing butterfly: after the first two exchange steps each lane carries ONE
// of the four sums, so the remaining log2(W)-2 steps move one value instead of
// four (7 exchanges + 4 broadcasts for W = 64 instead of 24).
// DPP lane moves (VALU only, no LDS round trip) for the in-row exchange steps
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
#define DSM_DPP_XOR1 0xB1    // quad_perm [1,0,3,2]
#define DSM_DPP_XOR2 0x4E    // quad_perm [2,3,0,1]
#define DSM_DPP_ROR4 0x124   // row_ror:4  (16-lane rows)
#define DSM_DPP_ROR8 0x128   // row_ror:8

template <int W>
__device__ __forceinline__ void group_allreduce_sum4(double &v0, double &v1, double &v2, double &v3)
{
    const int lane = __lane_id();
    const bool b0 = lane & 1, b1 = lane & 2;
    // step 1 (lane ^ 1): lanes with b0 = 0 keep (v0, v1), lanes with b0 = 1 keep (v2, v3)
    const double s0 = b0 ? v0 : v2, s1 = b0 ? v1 : v3;
    double k0 = b0 ? v2 : v0, k1 = b0 ? v3 : v1;
    k0 += dpp_mov<DSM_DPP_XOR1>(s0);
    k1 += dpp_mov<DSM_DPP_XOR1>(s1);
    // step 2 (lane ^ 2): b1 = 0 keeps k0, b1 = 1 keeps k1
    const double s = b1 ? k0 : k1;
    double k = b1 ? k1 : k0;
    k += dpp_mov<DSM_DPP_XOR2>(s);
    // lane (b0, b1) of every quad now holds the quad total of value 2*b0 + b1; rotations by 4
    // and 8 inside the 16-lane row keep (b0, b1) and finish the row; rows are combined by xor.
    k += dpp_mov<DSM_DPP_ROR4>(k);
    k += dpp_mov<DSM_DPP_ROR8>(k);
#pragma unroll
    for (int off = 16; off < W; off <<= 1) k += __shfl_xor(k, off, 64);
    if (W == 64) {
        // one variant per wavefront: the four totals are wave-uniform -> scalar broadcasts
        const int lo = __double2loint(k), hi = __double2hiint(k);
        v0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
        v1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 2), __builtin_amdgcn_readlane(lo, 2));
        v2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 1), __builtin_amdgcn_readlane(lo, 1));
        v3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 3), __builtin_amdgcn_readlane(lo, 3));
    } else {
        const int base = lane & ~(W - 1);
        v0 = __shfl(k, base + 0, 64);
        v1 = __shfl(k, base + 2, 64);
        v2 = __shfl(k, base + 1, 64);
        v3 = __shfl(k, base + 3, 64);
    }
}

// Same reduction for groups of W < 64 lanes whose four values were evaluated in a ROTATED candidate order (value i belongs to
// candidate (rot + i) & 3, rot differs between the groups of a wavefront): the total of candidate a sits in quad lane
// bitrev2((a - rot) & 3) after the butterfly, so every lane fetches its four totals in base order with one variable-index
// lane read each -- no select chain (the compiler turned it into divergent branches) after fixed broadcasts.
template <int W>
__device__ __forceinline__ void group_allreduce_sum4_unrotate(const double (&cv)[4], int rot, double (&l)[4])
{
    static_assert(W < 64, "one group per wavefront: use group_allreduce_sum4");
    const int lane = __lane_id();
    const bool b0 = lane & 1, b1 = lane & 2;
    const double s0 = b0 ? cv[0] : cv[2], s1 = b0 ? cv[1] : cv[3];
    double k0 = b0 ? cv[2] : cv[0], k1 = b0 ? cv[3] : cv[1];
    k0 += dpp_mov<DSM_DPP_XOR1>(s0);
    k1 += dpp_mov<DSM_DPP_XOR1>(s1);
    const double s = b1 ? k0 : k1;
    double k = b1 ? k1 : k0;
    k += dpp_mov<DSM_DPP_XOR2>(s);
    k += dpp_mov<DSM_DPP_ROR4>(k);
    k += dpp_mov<DSM_DPP_ROR8>(k);
#pragma unroll
    for (int off = 16; off < W; off <<= 1) k += __shfl_xor(k, off, 64);
    const int lo = __double2loint(k), hi = __double2hiint(k), qbase = (lane & ~3) << 2;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = (a - rot) & 3;
        const int src = qbase | ((((i & 1) << 1) | (i >> 1)) << 2);            // byte address of the source lane
        l[a] = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, hi), __builtin_amdgcn_ds_bpermute(src, lo));
    }
}

// ---- pieces of the tau sweep shared by tau_kernel (kernels_gibbs.hip) and gene_sweep_kernel (genes.hip)
// Lane-partial candidate log-probability: sum over this lane's NSL samples and the four observed bases of
// (float)count * log(rest + eta[a][b] * gamma_g)   (c_sample_tau.c:152-170); table log with a libm fallback for
// arguments that are not positive normal doubles.
// s0, sstride, S: slot j of this lane is sample s0 + j * sstride, a PADDED slot (no sample: count 0, abundance 1 in the staged tile) when
// that is >= S.  A padded slot's mixture value is a sum of eta entries -- positive for any error matrix without exact zeros, so the slot adds
// 0 * log(positive) = 0.  With exact zeros in eta (the identity; a zero row: the shim takes the caller's matrix verbatim, Eta_Sampler.py:
// 355-369) it can be 0, and 0 * log(0) = NaN would poison the lane's total where c_sample_tau.c has no such cell (round 6, found by
// scripts/dbg/fuzz_extreme.py: 5 of 300 cases).  Such a value takes the libm branch below anyway: that branch skips padded slots.
template <int NSL>
__device__ __forceinline__ double sweep_candidate(int a, const double (&xf)[NSL][4], const double (&st)[NSL][4],
                                                  const double (&gg)[NSL], const double *__restrict__ eS,
                                                  const double2 *__restrict__ ltab, int s0, int sstride, int S)
{
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
        double P[4];
        bool ok = true;
#pragma unroll
        for (int b = 0; b < 4; ++b) { P[b] = fma(eS[a * 4 + b], gg[j], st[j][b]); ok &= dsm_log_ok(P[b]); }
        if (__builtin_expect(ok, 1)) {
#pragma unroll
            for (int b = 0; b < 4; ++b) acc = fma(xf[j][b], dsm_log_core(P[b], ltab), acc);
        } else if (s0 + j * sstride < S) {
#pragma unroll
            for (int b = 0; b < 4; ++b) acc = fma(xf[j][b], dsm_log_slow(P[b]), acc);
        }
    }
    return acc;
}

// the same with the counts as the lean sweep keeps them, (float)count: the reference's (double)(float)count (c_sample_tau.c:164) is
// formed at each use -- the fp64 step of the sweep is rare and short of registers, not of issue slots
template <int NSL>
__device__ __forceinline__ double sweep_candidate_x(int a, const float (&xi)[NSL][4], const double (&st)[NSL][4],
                                                    const double (&gg)[NSL], const double *__restrict__ eS,
                                                    const double2 *__restrict__ ltab, int s0, int sstride, int S)
{
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
        double P[4];
        bool ok = true;
#pragma unroll
        for (int b = 0; b < 4; ++b) { P[b] = fma(eS[a * 4 + b], gg[j], st[j][b]); ok &= dsm_log_ok(P[b]); }
        // (the empty asm keeps the conversion HERE: hoisted out of the haplotype loop, the twelve doubles would live across the hot path)
        float x4[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) { x4[b] = xi[j][b]; asm volatile("" : "+v"(x4[b])); }
        if (__builtin_expect(ok, 1)) {
#pragma unroll
            for (int b = 0; b < 4; ++b) acc = fma((double)x4[b], dsm_log_core(P[b], ltab), acc);
        } else if (s0 + j * sstride < S) {               // (padded slots: see sweep_candidate)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc = fma((double)x4[b], dsm_log_slow(P[b]), acc);
        }
    }
    return acc;
}

// normaliseLog4 + sample4 (c_sample_tau.c:48-91) on the four group totals (identical on every lane of the group):
// exp(0) = 1 and exp(d < -745.2) = 0 exactly, so the usual case needs no exp; the CDF is inverted without the three
// fp64 divisions: u < ex0/sum <=> u*sum < ex0 (sum in [1,4]; the two forms can disagree only if u lies within
// ~1e-16 relative of a CDF edge).  u = raw 32-bit word / 2^32.
__device__ __forceinline__ int sweep_draw(const double (&l)[4], uint32_t uw)
{
    double mx = l[0];
#pragma unroll
    for (int a = 1; a < 4; ++a) if (l[a] > mx) mx = l[a];
    double ex[4], sum = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const double d = l[a] - mx;
        ex[a] = (d == 0.0) ? 1.0 : (d < -750.0) ? 0.0 : exp(d);
        sum += ex[a];
    }
    const double c0 = ex[0], c1 = ex[1] + c0, c2 = ex[2] + c1;
    const double us = ((double)uw * 2.3283064365386963e-10) * sum;
    return (us < c0) ? 0 : (us < c1) ? 1 : (us < c2) ? 2 : 3;
}

// ---- transposing butterfly: every lane brings NV values; afterwards lane l holds, in v[0], the
// wavefront total of value transpose_index<NV>(l).  Each exchange step halves the values a lane
// carries (it keeps the half selected by one lane-id bit and sends the other half to its partner),
// so NV + log2(64/NV)... exchanges replace 6*NV; the first two steps are DPP quad permutes.
template <typename T, int CTRL>
__device__ __forceinline__ T dpp_mov_t(T x)
{
    if constexpr (sizeof(T) == 8) {
        const double d = dpp_mov<CTRL>(__builtin_bit_cast(double, x));
        return __builtin_bit_cast(T, d);
    } else {
        return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
    }
}

// The same reduction for four fp32 values (the screening pass of the tau sweep): every lane of the group ends with all four
// totals, in value order.
template <int W>
__device__ __forceinline__ void group_allreduce_sum4_f32(float &v0, float &v1, float &v2, float &v3)
{
    const int lane = __lane_id();
    const bool b0 = lane & 1, b1 = lane & 2;
    const float s0 = b0 ? v0 : v2, s1 = b0 ? v1 : v3;
    float k0 = b0 ? v2 : v0, k1 = b0 ? v3 : v1;
    k0 += dpp_mov_t<float, DSM_DPP_XOR1>(s0);
    k1 += dpp_mov_t<float, DSM_DPP_XOR1>(s1);
    const float s = b1 ? k0 : k1;
    float k = b1 ? k1 : k0;
    k += dpp_mov_t<float, DSM_DPP_XOR2>(s);
    k += dpp_mov_t<float, DSM_DPP_ROR4>(k);
    k += dpp_mov_t<float, DSM_DPP_ROR8>(k);
#pragma unroll
    for (int off = 16; off < W; off <<= 1) k += __shfl_xor(k, off, 64);
    // quad lane (b0, b1) holds value 2 b0 + b1: value 0 in quad lane 0, 1 in lane 2, 2 in lane 1, 3 in lane 3
    v0 = dpp_mov_t<float, 0x00>(k);
    v1 = dpp_mov_t<float, 0xAA>(k);
    v2 = dpp_mov_t<float, 0x55>(k);
    v3 = dpp_mov_t<float, 0xFF>(k);
}

// ---- when do the fp32 totals of a step settle the fp64 draw?  (round 4: with the uniform taken into account)
// c32 = the four candidate totals in log2 units, each off its exact value by at most
//     E = [ X (1.4427 (G + 4) + 4) + |l| (4 NSL + 11) ] 2^-24          X = the variant's reads over all samples, |l| = the total's magnitude:
// every mixture value carries <= (G + 4) roundings of 2^-24 (fp32 inputs, one per link of the chain), i.e. 1.4427 (G + 4) 2^-24 on its
// log2, times its count; the hardware log2 is good to 2 ulp of its result or 2^-22 absolute where the result is tiny (pinned by
// tests/test_gpu_edges.py: the "+ 4" per read and two of the roundings below) and the products, the lane's running sum and the
// cross-lane butterfly add <= 4 NSL + 9 roundings of 2^-24 relative to |l|.  A gap between two totals is
// therefore known to 2 E, and with best = the largest total the three others together weigh, relative to it, at most
//     t = 3 x 2^-(gap to the second best - 2 E).
// sweep_draw (c_sample_tau.c:48-91) picks `best` whenever u sum falls inside its CDF interval: the edges below it sum to <= t, its own
// edge is >= 1, sum <= 1 + t -- so t < u < 1 - t suffices, whatever the exact values are.  Round 3 certified only gaps above 64 nats
// (t ~ 1e-28: every u but 0).  That left the low-abundance haplotypes of an over-fitted chain -- gamma ~ 1e-2, gaps of 8 ... 64 nats,
// half the steps of a G = 12 chain on a six-strain table (scripts/dbg/flat_diag.py) -- to the fp64 code, and with more than half of a
// sweep there the screen was suspended altogether: 572 us per sweep instead of 264 (scripts/misfit_scan.py).  A gap of 8 nats now
// certifies for all but 0.2 % of the uniforms.  The comparisons carry fp32 slack; NaN (the poisoned totals) fails them.
template <int NSL>
__device__ __forceinline__ bool screen_certify(const float (&c32)[4], float xtot, int G, uint32_t uw, int &best)
{
    best = 0;
    float m = c32[0];
#pragma unroll
    for (int a = 1; a < 4; ++a) if (c32[a] > m) { m = c32[a]; best = a; }
    float second = -3.0e38f;
#pragma unroll
    for (int a = 0; a < 4; ++a) second = (a == best) ? second : fmaxf(second, c32[a]);
    const float e2 = (xtot * (1.4427f * (float)(G + 4) + 4.0f) + (fabsf(m) + 128.0f) * (float)(4 * NSL + 11)) * 1.1920929e-7f;   // 2 E (2^-23)
    const float t = 3.0003f * __builtin_amdgcn_exp2f(fminf(second - m + e2, 0.0f));        // <= 3.0003 when the race is open
    const float u = (float)uw * 2.3283064365386963e-10f;                                   // rounded to 24 bits (may reach 1.0)
    return (fabsf(m) < 3.0e38f) && (u * 0.9999995f > t) && (u * 1.0000005f < 1.0f - t);
}

// ---- the screening pass of a sweep step (tau_kernel, gene_sweep_kernel).  The four candidate
// log-probabilities in fp32 (packed arithmetic, hardware log2) from the fp64 prefix `pre`, the links h > g and gamma_g; true when
// this lane's group can take `best` without the fp64 evaluation: the totals and the uniform word settle the draw (screen_certify above)
// and every mixture value is inside fp32's normal range.  xtot = the variant's reads over all samples (an upper bound).  gT32 = this group's gamma [G][SP] in fp32 (zero rows add nothing), eS32 = eta [16] then its column minima [4].
template <int LPV, int NSL>
__device__ __forceinline__ bool sweep_screen(const double (&pre)[NSL][4], const int (&xi)[NSL][4], uint64_t t, int g, int G, int lig,
                                             uint32_t uw, const float *__restrict__ gT32, const float *__restrict__ eS32, float xtot, int &best)
{
    constexpr int SP = LPV * NSL;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 s32[NSL][2];
    ISA_MARK("screen_prefix_cvt");
#pragma unroll
    for (int j = 0; j < NSL; ++j)
#pragma unroll
        for (int bp = 0; bp < 2; ++bp) s32[j][bp] = (f2){(float)pre[j][2 * bp], (float)pre[j][2 * bp + 1]};
    ISA_MARK("screen_links");
#pragma unroll 4
    for (int h = g + 1; h < G; ++h) {
        const f2 *er = reinterpret_cast<const f2 *>(eS32 + (int)((t >> (2 * h)) & 3) * 4);
        const f2 e01 = er[0], e23 = er[1];
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            const float gm = gT32[h * SP + lig + j * LPV];
            const f2 gm2 = (f2){gm, gm};
            s32[j][0] = __builtin_elementwise_fma(e01, gm2, s32[j][0]);
            s32[j][1] = __builtin_elementwise_fma(e23, gm2, s32[j][1]);
        }
    }
    ISA_MARK("screen_candidates");
    const f2 *e2 = reinterpret_cast<const f2 *>(eS32);               // [a][bp], then the column minima [bp]
    f2 acc2[4] = {(f2){0.0f, 0.0f}, (f2){0.0f, 0.0f}, (f2){0.0f, 0.0f}, (f2){0.0f, 0.0f}};
    float lbmin = 1.0f;
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
        const float g1 = gT32[g * SP + lig + j * LPV];
        const f2 g2 = (f2){g1, g1};
#pragma unroll
        for (int bp = 0; bp < 2; ++bp) {
            const f2 lb = __builtin_elementwise_fma(e2[8 + bp], g2, s32[j][bp]);     // smallest mixture value any candidate sees
            lbmin = fminf(lbmin, fminf(lb.x, lb.y));
            const f2 xs = (f2){(float)xi[j][2 * bp], (float)xi[j][2 * bp + 1]};      // the reference's (float)count
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f2 P = __builtin_elementwise_fma(e2[a * 2 + bp], g2, s32[j][bp]);
                const f2 lg = (f2){__builtin_amdgcn_logf(P.x), __builtin_amdgcn_logf(P.y)};
                acc2[a] = __builtin_elementwise_fma(xs, lg, acc2[a]);
            }
        }
    }
    float c32[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) c32[a] = acc2[a].x + acc2[a].y;
    ISA_MARK("screen_reduce");
    if (!(lbmin >= 1.0e-30f)) c32[0] = __builtin_nanf("");          // poisons the totals of the whole group
    group_allreduce_sum4_f32<LPV>(c32[0], c32[1], c32[2], c32[3]);    // log2 units
    ISA_MARK("screen_certify");
    return screen_certify<NSL>(c32, xtot, G, uw, best);
}

// the same from a prefix carried in fp32 ([sample slot][base pair]): the register-lean form of the sweep (kernels_gibbs.hip: tau_body, LEAN)
typedef float dsm_f2 __attribute__((ext_vector_type(2)));
template <int LPV, int NSL>
__device__ __forceinline__ bool sweep_screen32(const dsm_f2 (&pre32)[NSL][2], const float (&xi)[NSL][4], uint64_t t, int g, int G, int lig,
                                             uint32_t uw, const float *__restrict__ gT32, const float *__restrict__ eS32, float xtot, int &best)
{
    constexpr int SP = LPV * NSL;
    typedef dsm_f2 f2;
    f2 s32[NSL][2];
#pragma unroll
    for (int j = 0; j < NSL; ++j)
#pragma unroll
        for (int bp = 0; bp < 2; ++bp) s32[j][bp] = pre32[j][bp];
#pragma unroll 4
    for (int h = g + 1; h < G; ++h) {
        const f2 *er = reinterpret_cast<const f2 *>(eS32 + (int)((t >> (2 * h)) & 3) * 4);
        const f2 e01 = er[0], e23 = er[1];
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            const float gm = gT32[h * SP + lig + j * LPV];
            const f2 gm2 = (f2){gm, gm};
            s32[j][0] = __builtin_elementwise_fma(e01, gm2, s32[j][0]);
            s32[j][1] = __builtin_elementwise_fma(e23, gm2, s32[j][1]);
        }
    }
    const f2 *e2 = reinterpret_cast<const f2 *>(eS32);               // [a][bp], then the column minima [bp]
    f2 acc2[4] = {(f2){0.0f, 0.0f}, (f2){0.0f, 0.0f}, (f2){0.0f, 0.0f}, (f2){0.0f, 0.0f}};
    float lbmin = 1.0f;
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
        const float g1 = gT32[g * SP + lig + j * LPV];
        const f2 g2 = (f2){g1, g1};
#pragma unroll
        for (int bp = 0; bp < 2; ++bp) {
            const f2 lb = __builtin_elementwise_fma(e2[8 + bp], g2, s32[j][bp]);     // smallest mixture value any candidate sees
            lbmin = fminf(lbmin, fminf(lb.x, lb.y));
            const f2 xs = (f2){xi[j][2 * bp], xi[j][2 * bp + 1]};                    // the reference's (float)count, kept as such
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f2 P = __builtin_elementwise_fma(e2[a * 2 + bp], g2, s32[j][bp]);
                const f2 lg = (f2){__builtin_amdgcn_logf(P.x), __builtin_amdgcn_logf(P.y)};
                acc2[a] = __builtin_elementwise_fma(xs, lg, acc2[a]);
            }
        }
    }
    float c32[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) c32[a] = acc2[a].x + acc2[a].y;
    if (!(lbmin >= 1.0e-30f)) c32[0] = __builtin_nanf("");          // poisons the totals of the whole group
    group_allreduce_sum4_f32<LPV>(c32[0], c32[1], c32[2], c32[3]);    // log2 units
    return screen_certify<NSL>(c32, xtot, G, uw, best);
}

// ---- the screening pass of a NEAR-TIE step (round 4): the differences l_a - l_0 instead of the totals.
// The spare haplotypes of a chain with more haplotypes than the table has strains end up at gamma_sg ~ 1e-3 in every sample (a real
// `desman` chain at G = 12 on a six-strain table: six of them, scripts/dbg/chain_fp64.py): which base such a haplotype carries moves the
// likelihood by a fraction of a nat, its step is a near-tie that no bound on the TOTALS can settle, half of the sweep's steps ran the
// fp64 code: 675 us per sweep instead of 264.  This screen lives in a SECOND instantiation of the sweep kernel (tau_body: NT), chosen
// per call from the chain's own abundances: inside the one kernel its mere presence cost the normal path 5 % (profiles/r04_neartie_ab.txt).  With R = the rest mixture (haplotypes h != g),
//     l_a - l_0 = sum_{s,b} x_sb [log1p(r_ab) - log1p(r_0b)],      r_ab = e_ab gamma_sg / R_sb   (<~ 1e-2),
// and log1p(r) = r - r^2/2 + r^3/3 - [0, r^4/4]: a cubic in fp32 where the cell's largest r is below 1/32, the hardware log2 of 1 + r
// above (a base that no abundant haplotype carries has R ~ 0.01 and r ~ 0.1: few reads sit there, so the log's absolute error
// (2^-22, pinned by tests/test_gpu_edges.py, + the rounding of 1 + r) costs little, while the cubic's r^4 would not be small);
// D_a = sum x (f(r_ab) - f(r_0b)), together with a bound T >= sum x (r_max^4 / 4 or 3e-7, + 5e-6 r_max: at most G + 8 <= 40 roundings of 2^-24, twice over) on what any candidate's sum is
// off by (truncation / the log; the fp32 roundings of R, of the quotient, of the cubic and of the accumulation; r_max from the column
// maxima of eta, rounded up).  Every exp(l_a - max) the fp64 code forms
// (sweep_draw) is then known to a relative 2 T + 2e-6, and its comparison u sum < C_k comes out the same way as long as
// |u sum - C_k| > (8 T + 2e-5) sum for the three edges (twice the bound).  Returns true when this lane's group may take `tn` without
// the fp64 evaluation; false -- an edge too close, T not small (a haplotype that is not rare in some sample), a mixture value outside
// fp32's normal range, NaN -- leaves the step to the fp64 code as before.
// s32 = the rest mixture [sample slot][base pair], gq = gamma_g of the slots, e2 = eta in fp32 as [a][base pair], emax2 = its column maxima.
template <int LPV, int NSL>
__device__ __forceinline__ bool sweep_neartie_core(const dsm_f2 (&s32)[NSL][2], const float (&xs)[NSL][4], const float (&gq)[NSL], uint32_t uw,
                                                   const dsm_f2 *__restrict__ e2, const dsm_f2 *__restrict__ emax2, int &tn)
{
    typedef dsm_f2 f2;
    const f2 third = (f2){0.33333334f, 0.33333334f}, half = (f2){0.5f, 0.5f}, one = (f2){1.0f, 1.0f};
    f2 acc[4] = {(f2){0.0f, 0.0f}, (f2){0.0f, 0.0f}, (f2){0.0f, 0.0f}, (f2){0.0f, 0.0f}};
    f2 tb = (f2){0.0f, 0.0f};
    float smin = 1.0f;
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
#pragma unroll
        for (int bp = 0; bp < 2; ++bp) {
            const f2 R = s32[j][bp];
            smin = fminf(smin, fminf(R.x, R.y));
            const f2 q = (f2){gq[j] * __builtin_amdgcn_rcpf(R.x), gq[j] * __builtin_amdgcn_rcpf(R.y)};     // gamma / R
            const f2 x2 = (f2){xs[j][2 * bp], xs[j][2 * bp + 1]};
            const f2 rm = emax2[bp] * q;                                                                     // the largest r of this cell
            const bool px = rm.x < 0.03125f, py = rm.y < 0.03125f;                                           // cubic or hardware log
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f2 r = e2[a * 2 + bp] * q;
                f2 f = r * __builtin_elementwise_fma(r, __builtin_elementwise_fma(r, third, -half), one);         // r - r^2/2 + r^3/3
                if (!px) f.x = __builtin_amdgcn_logf(1.0f + r.x) * 0.69314718f;
                if (!py) f.y = __builtin_amdgcn_logf(1.0f + r.y) * 0.69314718f;
                acc[a] = __builtin_elementwise_fma(x2, f, acc[a]);
            }
            const f2 rm2 = rm * rm;
            f2 eb = rm2 * rm2 * (f2){0.25f, 0.25f};                                                          // per read: the cubic's truncation ...
            eb.x = px ? eb.x : 3.0e-7f; eb.y = py ? eb.y : 3.0e-7f;                                          // ... or the log's absolute error
            tb = __builtin_elementwise_fma(x2, __builtin_elementwise_fma(rm, (f2){5.0e-6f, 5.0e-6f}, eb), tb);
        }
    }
    float d1 = (acc[1].x - acc[0].x) + (acc[1].y - acc[0].y), d2 = (acc[2].x - acc[0].x) + (acc[2].y - acc[0].y),
          d3 = (acc[3].x - acc[0].x) + (acc[3].y - acc[0].y), T = tb.x + tb.y;
    if (!(smin >= 1.0e-30f)) T = __builtin_nanf("");                // poisons the bound of the whole group
    group_allreduce_sum4_f32<LPV>(d1, d2, d3, T);                     // natural units
    const float m = fmaxf(fmaxf(0.0f, d1), fmaxf(d2, d3));
    const float e0 = __builtin_amdgcn_exp2f((0.0f - m) * 1.44269504f), e1 = __builtin_amdgcn_exp2f((d1 - m) * 1.44269504f);
    const float e2x = __builtin_amdgcn_exp2f((d2 - m) * 1.44269504f), e3 = __builtin_amdgcn_exp2f((d3 - m) * 1.44269504f);
    const float c0 = e0, c1 = c0 + e1, c2 = c1 + e2x, sum = c2 + e3;
    const float us = ((float)uw * 2.3283064365386963e-10f) * sum;
    tn = (us < c0) ? 0 : (us < c1) ? 1 : (us < c2) ? 2 : 3;
    const float margin = (8.08f * T + 2.0e-5f) * sum;                // (T itself is an fp32 sum: + 1 %)
    const float gap = fminf(fminf(fabsf(us - c0), fabsf(us - c1)), fabsf(us - c2));
    return (T < 1.0e-2f) && (gap > margin);                          // NaN fails both comparisons
}

// the rest mixture of step g in fp32: the carried prefix (links h < g) + the links h > g (what sweep_screen builds for itself)
template <int LPV, int NSL>
__device__ __forceinline__ void screen_chain(dsm_f2 (&s32)[NSL][2], uint64_t t, int g, int G, int lig, const float *__restrict__ gT32,
                                             const float *__restrict__ eS32)
{
    constexpr int SP = LPV * NSL;
    typedef dsm_f2 f2;
#pragma unroll 4
    for (int h = g + 1; h < G; ++h) {
        const f2 *er = reinterpret_cast<const f2 *>(eS32 + (int)((t >> (2 * h)) & 3) * 4);
        const f2 e01 = er[0], e23 = er[1];
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            const float gm = gT32[h * SP + lig + j * LPV];
            const f2 gm2 = (f2){gm, gm};
            s32[j][0] = __builtin_elementwise_fma(e01, gm2, s32[j][0]);
            s32[j][1] = __builtin_elementwise_fma(e23, gm2, s32[j][1]);
        }
    }
}
template <int LPV, int NSL>
__device__ __forceinline__ bool sweep_neartie(const double (&pre)[NSL][4], const int (&xi)[NSL][4], uint64_t t, int g, int G, int lig,
                                              uint32_t uw, const float *__restrict__ gT32, const float *__restrict__ eS32, int &tn)
{
    dsm_f2 s32[NSL][2];
    float xs[NSL][4], gf[NSL];
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
        gf[j] = gT32[g * (LPV * NSL) + lig + j * LPV];
#pragma unroll
        for (int bp = 0; bp < 2; ++bp) s32[j][bp] = (dsm_f2){(float)pre[j][2 * bp], (float)pre[j][2 * bp + 1]};
#pragma unroll
        for (int b = 0; b < 4; ++b) xs[j][b] = (float)xi[j][b];
    }
    screen_chain<LPV, NSL>(s32, t, g, G, lig, gT32, eS32);
    return sweep_neartie_core<LPV, NSL>(s32, xs, gf, uw, reinterpret_cast<const dsm_f2 *>(eS32), reinterpret_cast<const dsm_f2 *>(eS32 + 20), tn);
}
template <int LPV, int NSL>
__device__ __forceinline__ bool sweep_neartie32(const dsm_f2 (&pre32)[NSL][2], const float (&xs)[NSL][4], uint64_t t, int g, int G, int lig,
                                                uint32_t uw, const float *__restrict__ gT32, const float *__restrict__ eS32, int &tn)
{
    dsm_f2 s32[NSL][2];
    float gf[NSL];
#pragma unroll
    for (int j = 0; j < NSL; ++j) {
        gf[j] = gT32[g * (LPV * NSL) + lig + j * LPV];
        s32[j][0] = pre32[j][0]; s32[j][1] = pre32[j][1];
    }
    screen_chain<LPV, NSL>(s32, t, g, G, lig, gT32, eS32);
    return sweep_neartie_core<LPV, NSL>(s32, xs, gf, uw, reinterpret_cast<const dsm_f2 *>(eS32), reinterpret_cast<const dsm_f2 *>(eS32 + 20), tn);
}

template <typename T, int NV, int CNT, int OFF>
__device__ __forceinline__ void transpose_reduce_step(T (&v)[NV], int lane)
{
    if constexpr (OFF < 64) {
        if constexpr (CNT > 1) {
            constexpr int H = CNT / 2;
            const bool up = lane & OFF;                 // this lane keeps the upper half
#pragma unroll
            for (int i = 0; i < H; ++i) {
                const T keep = up ? v[i + H] : v[i];
                const T send = up ? v[i] : v[i + H];
                T got;
                if constexpr (OFF == 1) got = dpp_mov_t<T, DSM_DPP_XOR1>(send);
                else if constexpr (OFF == 2) got = dpp_mov_t<T, DSM_DPP_XOR2>(send);
                else got = __shfl_xor(send, OFF, 64);
                v[i] = keep + got;
            }
            transpose_reduce_step<T, NV, H, OFF * 2>(v, lane);
        } else {
            v[0] += __shfl_xor(v[0], OFF, 64);
            transpose_reduce_step<T, NV, 1, OFF * 2>(v, lane);
        }
    }
}

template <int NV, typename T>
__device__ __forceinline__ T wave_transpose_reduce(T (&v)[NV])
{
    transpose_reduce_step<T, NV, NV, 1>(v, __lane_id());
    return v[0];
}

// index (within the NV values) that wave_transpose_reduce leaves on `lane`
template <int NV>
__device__ __forceinline__ int transpose_index(int lane)
{
    int idx = 0, h = NV / 2;
#pragma unroll
    for (int off = 1; off < 64 && h >= 1; off <<= 1, h >>= 1) idx += (lane & off) ? h : 0;
    return idx;
}

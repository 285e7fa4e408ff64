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

// ---- Philox4x32-10 (Salmon et al. 2011); checked against the Random123 KATs
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---- xoshiro128++ (Blackman & Vigna): the per-cell stream of the mu/E draw
struct Xo128 {
    uint32_t s0, s1, s2, s3;
    __device__ __forceinline__ uint32_t next()
    {
        const uint32_t a = s0 + s3;
        const uint32_t res = ((a << 7) | (a >> 25)) + s0;
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

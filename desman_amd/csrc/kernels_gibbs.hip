// kernels_gibbs.hip -- gfx950 kernels of the Gibbs sampler (HaploSNP_Sampler).
//
//   stats_kernel      A2  auxiliary-count sums  (HaploSNP_Sampler.py:284-309 via :266,:276)
//   dirichlet_kernel  A3/A4 gamma, eta draws    (HaploSNP_Sampler.py:263-281)
//   tau_kernel        A1  tau sweep             (sampletau/c_sample_tau.c:95-204)
//                     A5  log-likelihood        (HaploSNP_Sampler.py:431-442), fused epilogue
//   finalize_kernel   A5/A6 log-posterior, MAP tracking, traces (HaploSNP_Sampler.py:349-358)
//   mt_fill_kernel    GSL-compatible MT19937 stream for the tau draws (c_sample_tau.c:174)
//
// HBM layout: counts int32 in two layouts ([V][S][4] lane=sample for the tau
// sweep, [S][V][4] lane=variant for the per-read pass), tau packed 2 bits per
// haplotype in one u64 per variant, gamma [S][G] f64, eta [4][4] f64.
#include <string.h>

#include <mutex>

#include "dsm_device.h"
#include "dsm_host.h"
#include "dsm_stage2.h"
#include "log_table.h"

// =====================================================================
// one-time layout kernels
// =====================================================================
__global__ __launch_bounds__(256) void convert_counts_kernel(const int64_t *__restrict__ in,
                                                             int32_t *__restrict__ cnt_vs, int V, int S,
                                                             int *flag, double *partial,
                                                             unsigned long long *__restrict__ depth)
{
    __shared__ double red[256];
    __shared__ unsigned long long dep[DSM_MAX_S];          // reads per sample seen by this workgroup
    for (int i = threadIdx.x; i < S; i += 256) dep[i] = 0ull;
    __syncthreads();
    const size_t n = (size_t)V * S;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int4 c;
        int64_t x[4];
        int64_t tot = 0;
        bool bad = false;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            x[b] = in[i * 4 + b];
            bad |= (x[b] < 0) | (x[b] > 2147483647ll);
            tot += x[b];
        }
        bad |= tot > 2147483647ll;
        if (bad) atomicOr(flag, 1);
        else if (tot) atomicAdd(&dep[i % (size_t)S], (unsigned long long)tot);
        c.x = (int)x[0]; c.y = (int)x[1]; c.z = (int)x[2]; c.w = (int)x[3];
        reinterpret_cast<int4 *>(cnt_vs)[i] = c;
        // data-only part of the multinomial log-pdf (Desman_Utils.py:28-33)
        double t = lgamma((double)tot + 1.0);
#pragma unroll
        for (int b = 0; b < 4; ++b) t -= lgamma((double)x[b] + 1.0);
        acc += t;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
    for (int i = threadIdx.x; i < S; i += 256) if (dep[i]) atomicAdd(&depth[i], dep[i]);
}

__global__ void pack_tau_kernel(const int64_t *__restrict__ onehot, uint64_t *__restrict__ packed, int V, int G)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    uint64_t t = 0;
    for (int g = 0; g < G; ++g) {
        const int64_t *p = onehot + ((size_t)v * G + g) * 4;
        int idx = 0;                       // first 1 wins (c_sample_tau.c:116-122)
        for (int b = 3; b >= 0; --b) if (p[b] == 1) idx = b;
        t |= (uint64_t)idx << (2 * g);
    }
    packed[v] = t;
}

__global__ void unpack_tau_kernel(const uint64_t *__restrict__ packed, int64_t *__restrict__ onehot, int V, int G)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)V * G) return;
    const int v = (int)(i / G), g = (int)(i % G);
    const int idx = (int)((packed[v] >> (2 * g)) & 3);
    int64_t *p = onehot + i * 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) p[b] = (b == idx) ? 1 : 0;
}

// tau_sum[v][g][a] = #iterations with tau_vg == a over trace slots 1..n
__global__ void tau_sum_kernel(const uint64_t *__restrict__ trace, int n, int V, int G, int64_t *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)V * G) return;
    const int g = (int)(i / V), v = (int)(i % V);          // v fastest: coalesced trace reads
    int c[4] = {0, 0, 0, 0};
    for (int it = 1; it <= n; ++it) {
        const int idx = (int)((trace[(size_t)it * V + v] >> (2 * g)) & 3);
#pragma unroll
        for (int b = 0; b < 4; ++b) c[b] += (idx == b);
    }
    int64_t *p = out + ((size_t)v * G + g) * 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) p[b] = c[b];
}

// =====================================================================
// MT19937 (GSL gsl_rng_mt19937 semantics) -- one workgroup, state in LDS.
// A refill of the 624-word state has only three dependent phases
// ([0,227) [227,454) [454,624)), each fully lane-parallel.
// =====================================================================
__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// The state is double-buffered in LDS (old block / new block), so a phase has no read-after-write
// hazard inside itself and needs ONE barrier (3 per 624 words instead of 6); the tempered output of
// a word is stored by the lane that just produced it (no separate output pass).
__global__ __launch_bounds__(256) void mt_fill_kernel(uint32_t *__restrict__ state, uint32_t *__restrict__ out,
                                                      size_t n)
{
    __shared__ uint32_t buf[2][624];
    const int tid = threadIdx.x;
    int cur = 0;
    for (int i = tid; i < 624; i += 256) buf[0][i] = state[i];
    int pos = (int)state[624];
    __syncthreads();
    size_t done = 0;
    // words still unread in the resident block
    if (pos < 624) {
        const size_t take = (n < (size_t)(624 - pos)) ? n : (size_t)(624 - pos);
        for (int i = tid; i < (int)take; i += 256) out[i] = mt_temper(buf[0][pos + i]);
        pos += (int)take;
        done = take;
    }
    while (done < n) {
        const uint32_t *o = buf[cur];
        uint32_t *w = buf[cur ^ 1];
        const size_t left = n - done;
        const int lo[3] = {0, 227, 454}, hi[3] = {227, 454, 624};
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            const int i = lo[ph] + tid;
            if (i < hi[ph]) {
                // word i+1 = 624 is the new block's word 0; words i+397 >= 624 are new words i-227
                const uint32_t nxt = (i + 1 < 624) ? o[i + 1] : w[0];
                const uint32_t far = (i + 397 < 624) ? o[i + 397] : w[i - 227];
                const uint32_t y = (o[i] & 0x80000000u) | (nxt & 0x7fffffffu);
                const uint32_t val = far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                w[i] = val;
                if ((size_t)i < left) out[done + i] = mt_temper(val);
            }
            __syncthreads();
        }
        cur ^= 1;
        const size_t take = (left < 624) ? left : 624;
        pos = (int)take;
        done += take;
    }
    for (int i = tid; i < 624; i += 256) state[i] = buf[cur][i];
    if (tid == 0) state[624] = (uint32_t)pos;
}

// ---------------------------------------------------------------------
// The same stream, 4x wider per step.  MT19937's recurrence is  P x = 0  with  P = t^624 + t^397 + Q,  where t shifts
// the word sequence by one and  (Q x)[n] = A (U x[n] | L x[n+1])  is the "twist" (upper bit of one word, lower 31 of
// the next, times the companion matrix A).  All three terms are GF(2)-linear and t commutes with Q, so the
// cross terms of P^2 cancel:  P^(2^k) = t^(624 K) + t^(397 K) + Q^K  with K = 2^k, i.e.
//     x[j] = x[j - 227 K]  xor  (Q^K x)[j - 624 K]          (K = 1 is the textbook recurrence)
// which only reaches back 227 K words: 227 K new words are independent of each other.  Q^K is K twist levels over
// K + 1 consecutive words.  With K = 4 one workgroup of 16 wavefronts produces 908 words per barrier instead of 227
// (a 624-word refill has three dependent 227-word phases).  Every wavefront covers 64 consecutive positions and
// keeps the first 60 (the twist levels are neighbour exchanges inside the wavefront), 16 x 60 >= 908.  The first
// 2496 raw words after a state hand-off come from K = 1 and K = 2 steps (a step needs 624 K words of history).
// The raw words live in a 4096-word LDS ring; the kernel leaves the standard (624 words + position) state behind,
// so streams move between contexts / the host exactly as before.  Bit-identical to gsl_rng_mt19937.
// ---------------------------------------------------------------------
__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b)
{
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((b & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ void mt_fill_wide_body(uint32_t *__restrict__ state, uint32_t *__restrict__ out, size_t n)
{
    __shared__ uint32_t ring[4096];                 // raw word with absolute index a (block start = 0) at ring[a & 4095]
    __builtin_amdgcn_s_setprio(3);                  // a latency chain on one CU: do not queue behind the co-resident sweep / mu-E waves
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 624; i += 1024) ring[i] = state[i];
    const int pos = (int)state[624];
    __syncthreads();
    // outputs o = 0 .. n-1 are the tempered raw words pos + o; blocks start at multiples of 624
    const size_t E = (size_t)pos + n;                               // one past the last raw word consumed
    if (E <= 624) {                                                 // served from the resident block
        for (size_t o = tid; o < n; o += 1024) out[o] = mt_temper(ring[pos + o]);
        if (tid == 0) state[624] = (uint32_t)E;
        return;
    }
    for (int o = tid; o < 624 - pos; o += 1024) out[o] = mt_temper(ring[pos + o]);
    const size_t nb = (E - 1) / 624;                                // last block touched (>= 1): raw words up to R are needed
    const size_t R = (nb + 1) * 624;
    size_t T = 624;                                                 // raw words known so far
    while (T < R) {
        const int K = (T >= 2496) ? 4 : (T >= 1248) ? 2 : 1;
        const int W = 227 * K, keep = 64 - K;
        const int jo = wv * keep + lane;                            // offset of this lane's word inside the step
        if (wv * keep < W) {                                        // wavefront-uniform
            const size_t j = T + (size_t)jo;
            uint32_t y = ring[(j - 624 * (size_t)K) & 4095];        // level 0: x[j - 624 K + lane-relative] (consecutive)
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                // the word of lane + 1: DPP wave_shl:1 (a VALU move, no LDS round trip); lanes >= 64 - l hold garbage
                // from here on and are never kept
                const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y, 0x130, 0xf, 0xf, false);
                if (l < K) y = mt_twist(y, nx);
            }
            if (lane < keep && jo < W) {
                const uint32_t val = ring[(j - (size_t)W) & 4095] ^ y;
                ring[j & 4095] = val;
                if (j >= (size_t)pos + 0 && j < E) out[j - (size_t)pos] = mt_temper(val);
            }
        }
        T += (size_t)W;
        __syncthreads();
    }
    // standard state: the last block touched and the position inside it (1..624)
    const size_t B = R - 624;
    for (int i = tid; i < 624; i += 1024) state[i] = ring[(B + (size_t)i) & 4095];
    if (tid == 0) state[624] = (uint32_t)(E - B);
}

struct MtArgs { uint32_t *state, *out; size_t n; };
__global__ __launch_bounds__(1024) void mt_fill_wide_kernel(uint32_t *__restrict__ state, uint32_t *__restrict__ out, size_t n)
{
    mt_fill_wide_body(state, out, n);
}
// the generators of K chains, one workgroup each (blockIdx.x = chain)
__global__ __launch_bounds__(1024) void mt_fill_wide_kernel_b(BatchArgs<MtArgs> b)
{
    const MtArgs &a = b.p[blockIdx.x];
    mt_fill_wide_body(a.state, a.out, a.n);
}

// ---------------------------------------------------------------------
// The same stream from SEVERAL compute units (round 5).  One workgroup makes a sweep's V G words at ~2.5 per ns whatever its width;
// updateTau on large tables (the `-r` path, bin/desman:181-206) waited for it: 400 000 words in 153-195 us against a sweep of 184.
// MT19937's block reload -- 624 words -> the next 624 -- is GF(2)-linear in the 19 968 bits of the array, so the array D = 210 blocks
// (131 040 words) further on is M s with a fixed 19 968 x 19 968 bit matrix M, whatever s.  M is built once per device and process:
// column c is the array that 210 reloads make of unit vector c (one wavefront per column: a reload is wave-synchronous LDS code,
// no barrier; 0.3 ms for all columns), stored column by column [19 968][624] u32 = 50 MB; likewise M4 = 840 blocks.  A state is then
// moved D words on by XOR-ing the columns its set bits select (mt_jump_kernel: 208 column chunks x 3 word groups, 50 MB read,
// a GF(2) sum: exact), and a fill of n >= 6 D words runs as up to 32 chunks at a time: the starting arrays S[q] = M4 S[q - 4] in a chain,
// the three between two of them from M in three launches for all q at once, then ONE launch of a generator workgroup per chunk -- the
// generator above, on its own chunk and its own copy of the state.  The last chunk's final state is the stream's: bit-identical to the
// serial generator and to gsl_rng_mt19937 (tests/test_gpu_edges.py: 10^7 words across chunk and round boundaries, hand-offs).
// ---------------------------------------------------------------------
#define MTJ_BLOCKS 210
#define MTJ_D ((size_t)MTJ_BLOCKS * 624)
#define MTJ_BITS 19968
#define MTJ_NCH 208                    // column chunks of a jump: 96 columns = 3 state words each
#define MTJ_PMAX 32                    // chunks per round
#define MTJ_SW 640                     // words per state slot (624 + position, padded)

// one wavefront: `nreload` block reloads of the array in LDS (in place: a pass of 64 positions reads what no earlier pass has
// written -- position i needs the OLD words i and i + 1 and word i + 397 (old) or i - 227 (new))
__device__ __forceinline__ void mt_reload_wave(uint32_t *mt, int lane)
{
    for (int i0 = 0; i0 < 624; i0 += 64) {
        const int i = i0 + lane;
        uint32_t v = 0;
        if (i < 623) {
            const uint32_t far = (i < 227) ? mt[i + 397] : mt[i - 227];
            v = far ^ mt_twist(mt[i], mt[i + 1]);
        }
        __builtin_amdgcn_wave_barrier();
        if (i < 623) mt[i] = v;
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) mt[623] = mt[396] ^ mt_twist(mt[623], mt[0]);
    __builtin_amdgcn_wave_barrier();
}
__global__ __launch_bounds__(256) void mt_jump_build_kernel(uint32_t *__restrict__ M, int nreload)
{
    __shared__ uint32_t st[4][624];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + wv;                          // column = unit vector: bit c % 32 of word c / 32
    if (c >= MTJ_BITS) return;
    uint32_t *mt = st[wv];
    for (int i = lane; i < 624; i += 64) mt[i] = (i == (c >> 5)) ? (1u << (c & 31)) : 0u;
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < nreload; ++r) mt_reload_wave(mt, lane);
    for (int i = lane; i < 624; i += 64) M[(size_t)c * 624 + i] = mt[i];
}
// S[dst] ^= M S[src] for the (src, dst) pairs of the launch: pair y is src = base + step y + from, dst = src + to (skipped from dst = P on).
// grid (3 word groups, MTJ_NCH column chunks, pairs); S[dst] is zero before (one memset per round).  Chunk 0 copies the position word.
__global__ __launch_bounds__(256) void mt_jump_kernel(const uint32_t *__restrict__ M, uint32_t *__restrict__ S, int base, int step, int from, int to, int P)
{
    const int src = base + step * (int)blockIdx.z + from, dst = src + to;
    if (dst >= P) return;
    const uint32_t *__restrict__ s = S + (size_t)src * MTJ_SW;
    uint32_t *__restrict__ o = S + (size_t)dst * MTJ_SW;
    const int w = blockIdx.x * 256 + threadIdx.x;
    const int c0 = blockIdx.y * (MTJ_BITS / MTJ_NCH);           // 96 columns = 3 words of the state
    uint32_t acc = 0;
    if (w < 624) {
#pragma unroll
        for (int k = 0; k < (MTJ_BITS / MTJ_NCH) / 32; ++k) {
            const uint32_t bits = s[(c0 >> 5) + k];             // (uniform over the workgroup)
            const uint32_t *__restrict__ col = M + ((size_t)c0 + 32 * (size_t)k) * 624 + w;
            uint32_t v[32];
#pragma unroll
            for (int b = 0; b < 32; ++b) v[b] = col[(size_t)b * 624];      // 32 loads in flight (a loop over the set bits alone was one
#pragma unroll                                                            //  exposed latency per column: 55 us per jump instead of 12)
            for (int b = 0; b < 32; ++b) acc ^= ((bits >> b) & 1u) ? v[b] : 0u;
        }
        if (acc) atomicXor(&o[w], acc);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) o[624] = s[624];
}
struct MtParArgs { uint32_t *S, *out; size_t n; };
// one generator workgroup per chunk: chunk p starts from S[p] and writes words [p D, min((p + 1) D, n)); it leaves its final state in S[p]
__global__ __launch_bounds__(1024) void mt_fill_par_kernel(MtParArgs a)
{
    const size_t p = blockIdx.x, lo = p * MTJ_D;
    if (lo >= a.n) return;
    const size_t cnt = (a.n - lo < MTJ_D) ? a.n - lo : MTJ_D;
    mt_fill_wide_body(a.S + p * MTJ_SW, a.out + lo, cnt);
}

// The jump tables of a device: 2 x 50 MB, built once per process and device by the first long fill (2 x 2.2 ms), shared by every context of
// the device, kept until the process ends or dsm_release_device_caches() is called.  Devices 0 .. 15 have a slot; a higher device index
// gets no tables (the serial generator serves: same words).  Everything about a slot -- allocation, build, the fill kernel's
// shared-memory attribute -- happens under its mutex; an error after the allocations frees both tables and marks the slot failed
// (it used to return with ready = failed = false, and the next call allocated another pair).
struct MtJumpDev { std::mutex mu; uint32_t *M1 = nullptr, *M4 = nullptr; bool ready = false, failed = false, attr = false; };
#define MTJ_NDEV 16
static MtJumpDev g_mtj[MTJ_NDEV];
static int mt_jump_tables(dsm_ctx *c, hipStream_t stream, const uint32_t **M1, const uint32_t **M4)
{
    *M1 = *M4 = nullptr;
    if (c->device < 0 || c->device >= MTJ_NDEV) return DSM_OK;
    MtJumpDev &d = g_mtj[c->device];
    std::lock_guard<std::mutex> lk(d.mu);
    if (!d.ready && !d.failed) {
        const size_t bytes = (size_t)MTJ_BITS * 624 * sizeof(uint32_t);
        bool ok = hipMalloc((void **)&d.M1, bytes) == hipSuccess && hipMalloc((void **)&d.M4, bytes) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(mt_jump_build_kernel, dim3(MTJ_BITS / 4), dim3(256), 0, stream, d.M1, MTJ_BLOCKS);
            hipLaunchKernelGGL(mt_jump_build_kernel, dim3(MTJ_BITS / 4), dim3(256), 0, stream, d.M4, 4 * MTJ_BLOCKS);
            ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;   // other contexts of the device use the tables from their own streams
        }
        if (ok && !d.attr) {
            ok = hipFuncSetAttribute((const void *)mt_fill_par_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) == hipSuccess;
            d.attr = ok;
        }
        if (!ok) {
            if (d.M1) { (void)hipFree(d.M1); d.M1 = nullptr; }
            if (d.M4) { (void)hipFree(d.M4); d.M4 = nullptr; }
            d.failed = true;                                    // no tables on this device: the serial generator serves (same words)
            (void)hipGetLastError();
        } else d.ready = true;
    }
    if (d.ready) { *M1 = d.M1; *M4 = d.M4; }
    return DSM_OK;
}
// frees the jump tables of every device (callers: dsm_release_device_caches; no fill may be in flight)
void mt_jump_release()
{
    for (int i = 0; i < MTJ_NDEV; ++i) {
        MtJumpDev &d = g_mtj[i];
        std::lock_guard<std::mutex> lk(d.mu);
        if (d.M1 || d.M4) {
            int cur = 0;
            (void)hipGetDevice(&cur);
            (void)hipSetDevice(i);
            if (d.M1) (void)hipFree(d.M1);
            if (d.M4) (void)hipFree(d.M4);
            (void)hipSetDevice(cur);
        }
        d.M1 = d.M4 = nullptr; d.ready = false; d.failed = false;
    }
}

// a fill of n words as rounds of up to MTJ_PMAX chunks of D words; returns how many words it made (0: tables not available)
static int mt_fill_parallel(dsm_ctx *c, uint32_t *out, size_t n, hipStream_t stream, size_t *made)
{
    *made = 0;
    const uint32_t *M1 = nullptr, *M4 = nullptr;
    { const int r = mt_jump_tables(c, stream, &M1, &M4); if (r != DSM_OK) return r; }
    if (!M1) return DSM_OK;
    if (!c->mt_jstates) {
        hipError_t e = hipMalloc((void **)&c->mt_jstates, (size_t)MTJ_PMAX * MTJ_SW * sizeof(uint32_t));
        if (e != hipSuccess) { (void)hipGetLastError(); return DSM_OK; }
    }
    uint32_t *S = c->mt_jstates;
    while (n - *made >= 2 * MTJ_D) {
        const size_t rem = n - *made;
        const int P = (int)std::min<size_t>(MTJ_PMAX, (rem + MTJ_D - 1) / MTJ_D);
        const size_t words = std::min<size_t>(rem, (size_t)P * MTJ_D);
        HIP_TRY(hipMemcpyAsync(S, c->mt_state, 625 * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipMemsetAsync(S + MTJ_SW, 0, (size_t)(P - 1) * MTJ_SW * sizeof(uint32_t), stream));
        for (int q = 4; q < P; q += 4)                         // S[q] = M4 S[q - 4]
            hipLaunchKernelGGL(mt_jump_kernel, dim3(3, MTJ_NCH, 1), dim3(256), 0, stream, M4, S, q - 4, 0, 0, 4, P);
        const int nq = (P + 3) / 4;
        for (int r = 1; r <= 3 && r < P; ++r)                  // S[4 y + r] = M S[4 y + r - 1], every y in one launch
            hipLaunchKernelGGL(mt_jump_kernel, dim3(3, MTJ_NCH, nq), dim3(256), 0, stream, M1, S, 0, 4, r - 1, 1, P);
        const MtParArgs a{S, out + *made, words};
        hipLaunchKernelGGL(mt_fill_par_kernel, dim3(P), dim3(1024), (size_t)140 * 1024, stream, a);   // (a CU to itself each: see k_mt_fill)
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(c->mt_state, S + (size_t)(P - 1) * MTJ_SW, 625 * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
        *made += words;
    }
    return DSM_OK;
}

// =====================================================================
// A2: auxiliary-count sums.  Work item = one (variant, observed base) pair of
// one sample with a non-zero count; the item list of every sample is built
// once at upload, sorted by decreasing count, so the 64 lanes of a wavefront
// run read loops of (almost) equal length and the heaviest workgroups are
// dispatched first.  Every read draws its haplotype g with probability
// gamma[s,g]*eta[tau_vg,b]/sum from the item's xoshiro128+ stream (keyed by
// Philox(seed; cell, read chunk, iter, base)): one 32-bit word against G-1 thresholds,
// two VALU issues per threshold.  Only the sums sum_mu[s,g] and esum[b,a]
// ever leave the registers.
// Specification restated in oracle/desman_oracle.c: orc_stats_counter.
// =====================================================================
// r < thr ? cnt + 1 : cnt in two VALU issues (compare to VCC, add-with-carry of 0)
__device__ __forceinline__ void count_if_less(uint32_t &cnt, uint32_t r, uint32_t thr)
{
    asm("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt) : "v"(r), "v"(thr) : "vcc");
}

// EXACT: the haplotype count equals GMAX (a compile-time constant: no per-haplotype branches)
template <int GMAX, bool EXACT>
__global__ __launch_bounds__(256) void stats_kernel(const int2 *__restrict__ items,      // [S][stride] {v*4+b, reads [| chunk << 12]}
                                                    const int32_t *__restrict__ nitems,  // [S]
                                                    const int32_t *__restrict__ blk_tab, // [grid][3] {sample, j, n_j}
                                                    const uint64_t *__restrict__ tau,
                                                    const double *__restrict__ gamma,
                                                    const double *__restrict__ eta, int V, int S, int G,
                                                    int stride, int chunked, uint32_t k0, uint32_t k1, uint32_t iter,
                                                    unsigned long long *__restrict__ sum_mu,
                                                    unsigned long long *__restrict__ esum)
{
    __shared__ double gs[GMAX];
    __shared__ double es[16];
    __shared__ unsigned long long acc[GMAX + 16];
    __shared__ uint32_t eacc[16][256];
    const int tid = threadIdx.x;
    // workgroup -> (sample, j-th of n_j workgroups of that sample); the table gives every sample
    // a share of the resident workgroups proportional to its depth, so all workgroups carry the
    // same number of reads and the launch is exactly one resident wave of workgroups
    const int s = blk_tab[blockIdx.x * 3], bj = blk_tab[blockIdx.x * 3 + 1], bn = blk_tab[blockIdx.x * 3 + 2];
    const int n_s = nitems[s];
    if (tid < GMAX) gs[tid] = (tid < G) ? gamma[(size_t)s * G + tid] : 0.0;
    if (tid < 16) es[tid] = eta[tid];
    if (tid < GMAX + 16) acc[tid] = 0ull;
    __syncthreads();

    uint32_t mu[GMAX];
    uint32_t e[16];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) mu[g] = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) eacc[i][tid] = 0;

    const int Gc = EXACT ? GMAX : G;
    // grid-stride over the sample's sorted list: every workgroup gets heavy and light items, and
    // the 64 items a wavefront holds at any time are adjacent in the sort (equal loop lengths)
    for (int k = bj * 256 + tid; k < n_s; k += bn * 256) {
        const int2 it = items[(size_t)s * stride + k];
        const int v = it.x >> 2, b = it.x & 3;
        const int nb = chunked ? (it.y & 0xfff) : it.y;                  // reads of this item (one chunk of a count)
        const uint32_t chunk = chunked ? (uint32_t)it.y >> 12 : 0u;
        const uint64_t t = tau[v];
        const uint32_t cell = (uint32_t)s * (uint32_t)V + (uint32_t)v;   // V*S < 2^32 (checked at upload)
        uint32_t seedw[4];
        philox4x32_10(cell, chunk, iter, DSM_STREAM_STATS + (uint32_t)b, k0, k1, seedw);
        Xo128 rng{seedw[0], seedw[1], seedw[2], seedw[3]};
        if ((rng.s0 | rng.s1 | rng.s2 | rng.s3) == 0u) rng.s0 = 1u;
        // cumulative weights -> 32-bit thresholds
        double cum[GMAX];
        double run = 0.0;
#pragma unroll
        for (int g = 0; g < GMAX; ++g) {
            if (g < Gc) {
                const int ig = (int)((t >> (2 * g)) & 3);
                const double w = gs[g] * es[ig * 4 + b];
                run = run + w;
            }
            cum[g] = run;
        }
        const double scale = 4294967296.0 / run;
        uint32_t thr[GMAX], cnt[GMAX];
#pragma unroll
        for (int g = 0; g < GMAX; ++g) {
            // floor + clamp to 2^32-1 is exactly what v_cvt_u32_f64 does (truncate, saturate) for a value >= 0
            uint32_t q;
            asm("v_cvt_u32_f64 %0, %1" : "=v"(q) : "v"(cum[g] * scale));
            thr[g] = (g < Gc - 1) ? q : 0u;     // unused slots never count
            cnt[g] = 0;
        }
        int i = 0;
        for (; i + 1 < nb; i += 2) {                     // two reads per trip: half the loop overhead
            const uint32_t r0 = rng.next();
            const uint32_t r1 = rng.next();
#pragma unroll
            for (int g = 0; g < GMAX - 1; ++g) { count_if_less(cnt[g], r0, thr[g]); count_if_less(cnt[g], r1, thr[g]); }
        }
        if (i < nb) {
            const uint32_t r = rng.next();
#pragma unroll
            for (int g = 0; g < GMAX - 1; ++g) count_if_less(cnt[g], r, thr[g]);
        }
        // E[b][tau_g] += m_g goes to the lane's own column of an LDS table ([16][256]: the bank is the lane id
        // whatever the row, so the 64 atomics of a wavefront never conflict): 3 issues per haplotype instead
        // of ~16 compare/select/add issues on 16 register accumulators
        uint32_t *erow = &eacc[b * 4][tid];
#pragma unroll
        for (int g = 0; g < GMAX; ++g) {
            if (g < Gc) {
                const uint32_t hi = (g == Gc - 1) ? (uint32_t)nb : cnt[g];
                const uint32_t lo = (g == 0) ? 0u : cnt[g > 0 ? g - 1 : 0];
                const uint32_t m = hi - lo;
                mu[g] += m;
                const int ig = (int)((t >> (2 * g)) & 3);
                atomicAdd(erow + ig * 256, m);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = eacc[i][tid];
    // wavefront reduce -> LDS -> one global atomic per workgroup and counter.  The GMAX + 16 counters go
    // through one transposing butterfly (lane l ends up with the wavefront total of counter
    // transpose_index(l)), so a wavefront issues NV exchanges and ONE LDS atomic instead of 6 per counter.
    {
        constexpr int NV = (GMAX + 16 <= 32) ? 32 : 64;
        const int lane = tid & 63;
        uint32_t v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = (i < GMAX) ? mu[i < GMAX ? i : 0] : (i < GMAX + 16 ? e[(i - GMAX) & 15] : 0u);
        const uint32_t tot = wave_transpose_reduce<NV>(v);
        const int idx = transpose_index<NV>(lane);
        if (lane < NV && idx < GMAX + 16 && tot) atomicAdd(&acc[idx], (unsigned long long)tot);
    }
    __syncthreads();
    if (tid < G) { if (acc[tid]) atomicAdd(&sum_mu[(size_t)s * G + tid], acc[tid]); }
    else if (tid >= GMAX && tid < GMAX + 16) { if (acc[tid]) atomicAdd(&esum[tid - GMAX], acc[tid]); }
}

// =====================================================================
// A5/A6: finalize one iteration: fixed-order reduction of the per-workgroup LL
// partials and of the Dirichlet log-prior terms, lp, MAP tracking, traces.
// it < 0: no trace slot (entry state / plain evaluation).  Runs either as its
// own single-workgroup kernel or as one extra workgroup of the NEXT iteration's
// dirichlet launch (it only needs data that launch order already guarantees).
// =====================================================================
#ifndef DSM_NT_RARE
#define DSM_NT_RARE 0.01f     /* a haplotype whose abundance is at most this in every sample is "rare": its steps take the near-tie screen first */
#endif
struct FinalParams {
    const double *ll_partial; int nblocks;
    double ll_const, tau_prior;
    const double *prior; int S;      // [S + 4] per-row Dirichlet log-prior terms of the finalized state
    int *nchange;
    int it;
    double *ll_trace, *lp_trace; int *nchange_trace;
    double *star;                 // {lp_star, slot}
    const double *gamma_src; double *gamma_star; int SG;   // gamma / eta of the finalized state
    const double *eta_src; double *eta_star;
    int star_mode;                // 0 = keep the better lp, 1 = force (entry state), 2 = never
    double *scalars;              // [0]=ll [1]=lp of this evaluation
    const double *ll_given;       // sharded chain: {ll, nchange} summed over the shards (api.hip: dsm_ctx_gibbs_update_sharded), else null
    const uint32_t *step_cnt;     // [nblocks][2] wavefront-steps of the finalized launch: run / left to the fp64 code (~0: not screened)
    unsigned long long *sweep_stats;   // [2] running totals of the two
    uint32_t *screen_ctl;         // [0] sweeps still to run without the screening pass
    uint32_t *blk_order;          // [nblocks] out (or null): the order the next sweep of this parity runs its blocks in -- those first that
                                  // left a step to the fp64 code in the finalized one (tau_body: `order`)
    int *rare_out; int G;         // stand-alone launch at the end of a Gibbs call (or null): how many haplotypes of the finalized state are rare in
                                  // every sample (gamma_rare_kernel's count), written straight to the host's pinned word -- what the NEXT call's
                                  // choice of sweep instantiation goes by; as its own launch + copy + clear it was 15 us of every call
};

__device__ void finalize_body(const FinalParams &p, double *red, double *redp, int *flag, int tid, int nthr)
{
    double a = 0.0, b = 0.0;
    for (int i = tid; i < p.nblocks; i += nthr) a += p.ll_partial[i];
    for (int i = tid; i < p.S + 4; i += nthr) b += p.prior[i];
    red[tid] = a; redp[tid] = b;
    __syncthreads();
    for (int o = nthr / 2; o >= 1; o >>= 1) {
        if (tid < o) { red[tid] += red[tid + o]; redp[tid] += redp[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double ll = p.ll_given ? p.ll_given[0] : p.ll_const + red[0];
        const double lp = ll + redp[0] + p.tau_prior;
        p.scalars[0] = ll; p.scalars[1] = lp;
        const int nch = p.ll_given ? (int)p.ll_given[1] : *p.nchange;
        if (!p.ll_given) *p.nchange = 0;
        if (p.it >= 0) { p.ll_trace[p.it] = ll; p.lp_trace[p.it] = lp; p.nchange_trace[p.it] = nch; }
        int f = 0;
        if (p.star_mode == 1 || (p.star_mode == 0 && lp > p.star[0])) { p.star[0] = lp; p.star[1] = (double)(p.it + 1); f = 1; }
        *flag = f;
    }
    __syncthreads();
    if (*flag) {
        for (int i = tid; i < p.SG; i += nthr) p.gamma_star[i] = p.gamma_src[i];
        if (tid < 16) p.eta_star[tid] = p.eta_src[tid];
    }
    // the screening pass of the tau sweep is worth its ~20 % only while it decides enough steps: a screened
    // step costs ~720 issue cycles, an fp64 step ~3 700, one that takes both 4 420 -- the screen pays while it leaves fewer than
    // (3 700 - 720) / 3 700 = 0.8 of the steps open.  A sweep that left MORE than four fifths of its wavefront-steps to the fp64
    // code (very shallow data, a handful of samples) switches it off for the next 15 sweeps, then it is tried again.  (Round 3
    // switched at one half: the chains of a G-sweep with twice as many haplotypes as strains sit right there -- their spare
    // haplotypes, gamma ~ 1e-3, make near-ties of half the steps -- and ran most of their sweeps all-fp64: scripts/dbg/chain_fp64.py.)  Which steps are screened
    // changes no draw outside near-ties: a step after a screened one evaluates the current base's log-probability afresh where
    // an all-fp64 sweep re-uses the previous step's value -- the same number up to its last bits (a flip needs the uniform
    // within ~1e-13 of a CDF edge).  The rule itself is deterministic (counts of the previous launches only).
    if (p.step_cnt) {
        __syncthreads();
        unsigned long long steps = 0, exact = 0;
        unsigned plain = 0;
        for (int i = tid; i < p.nblocks; i += nthr) {
            const uint32_t a = p.step_cnt[2 * i], b = p.step_cnt[2 * i + 1];
            steps += a;
            if (b == 0xFFFFFFFFu) { plain = 1; exact += a; } else exact += b;
        }
        red[tid] = (double)steps; redp[tid] = (double)exact;              // < 2^53: exact in fp64
        __syncthreads();
        for (int o = nthr / 2; o >= 1; o >>= 1) {
            if (tid < o) { red[tid] += red[tid + o]; redp[tid] += redp[tid + o]; }
            __syncthreads();
        }
        const unsigned any_plain = __syncthreads_or((int)plain);
        if (tid == 0 && red[0] > 0.0) {
            p.sweep_stats[0] += (unsigned long long)red[0];
            p.sweep_stats[1] += (unsigned long long)redp[0];
            if (any_plain) { if (p.screen_ctl[0] > 0) p.screen_ctl[0] -= 1; }
            else if (5.0 * redp[0] > 4.0 * red[0]) p.screen_ctl[0] = 15;
        }
        // The rare fp64 step of a sweep is long, and a workgroup that meets one late in the launch is the launch's tail (2.4 us of 36.8 at
        // config 3).  Close races stay close from sweep to sweep, so the blocks that met one go FIRST next time: a stable
        // partition of the block indices (thread t owns a run of consecutive blocks; exclusive scan of the runs' counts).  Which block runs
        // where changes no result: partial sums and counters are filed under the block's own index.
        if (p.blk_order) {
            __syncthreads();
            int *cnt = reinterpret_cast<int *>(red);                        // [nthr] (red / redp are free again)
            const int per = (p.nblocks + nthr - 1) / nthr, lo = tid * per, hi = min(p.nblocks, lo + per);
            auto cold = [&](int i) { const uint32_t b = p.step_cnt[2 * i + 1]; return b != 0u && b != 0xFFFFFFFFu; };
            int nc = 0;
            for (int i = lo; i < hi; ++i) nc += cold(i) ? 1 : 0;
            int incl = nc;                                                  // inclusive scan over the wavefront, then over the wavefronts' totals
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += y; }
            if ((tid & 63) == 63) cnt[tid >> 6] = incl;
            __syncthreads();
            int before = incl - nc, total = 0;
            for (int k = 0; k < nthr / 64; ++k) { const int x = cnt[k]; if (k < (tid >> 6)) before += x; total += x; }
            int pc = before, pw = total + (lo - before);                      // next place among the first / among the rest
            for (int i = lo; i < hi; ++i) { if (cold(i)) p.blk_order[pc++] = (uint32_t)i; else p.blk_order[pw++] = (uint32_t)i; }
        }
    }
}

// NB the reduction tree depends on the workgroup size, so one chain must always finalize with the same
// size to stay bitwise reproducible: both users below run it with 256 threads.
__global__ __launch_bounds__(256) void finalize_kernel(FinalParams p)
{
    __shared__ double red[256], redp[256];
    __shared__ int flag;
    finalize_body(p, red, redp, &flag, threadIdx.x, 256);
    if (p.rare_out) {                                     // (gamma_rare_kernel's count on the finalized state's abundances)
        __syncthreads();
        if (threadIdx.x == 0) flag = 0;
        __syncthreads();
        for (int g = threadIdx.x; g < p.G; g += 256) {
            double m = 0.0;
            for (int s = 0; s < p.S; ++s) m = fmax(m, p.gamma_src[(size_t)s * p.G + g]);
            if ((float)m <= (float)DSM_NT_RARE) atomicAdd(&flag, 1);
        }
        __syncthreads();
        if (threadIdx.x == 0) *p.rare_out = flag;
    }
}

// =====================================================================
// A3/A4: Dirichlet draws from the sums, one workgroup.
// gamma[s,:] ~ Dir(alpha + sum_mu[s,:]); clamp < eps -> eps; renormalise
// eta[a,:]   ~ Dir(delta + esum[:,a])
// Gamma variates: Marsaglia & Tsang (2000) with the shape<1 boost, normals by
// Box-Muller, uniforms from Philox4x32-10 keyed by (seed; variate, attempt, iter).
// Also evaluates the two Dirichlet log-priors of the NEW state
// (Desman_Utils.py:35-44) and zeroes the sums for the next iteration.
// =====================================================================
__device__ double gamma_variate(double shape, uint32_t idx, uint32_t iter, uint32_t k0, uint32_t k1)
{
    const double a = (shape < 1.0) ? shape + 1.0 : shape;
    const double d = a - 1.0 / 3.0;
    const double c = 1.0 / sqrt(9.0 * d);
    double res = 0.0, uboost = 1.0;
    for (uint32_t attempt = 0; attempt < 4096u; ++attempt) {
        uint32_t r0[4], r1[4];
        philox4x32_10(idx, 2u * attempt, iter, DSM_STREAM_DIRI, k0, k1, r0);
        philox4x32_10(idx, 2u * attempt + 1u, iter, DSM_STREAM_DIRI, k0, k1, r1);
        const double u1 = u01_open(r0[0], r0[1]), u2 = u01_open(r0[2], r0[3]);
        const double u3 = u01_open(r1[0], r1[1]);
        uboost = u01_open(r1[2], r1[3]);
        const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        double vv = 1.0 + c * x;
        if (vv <= 0.0) continue;
        vv = vv * vv * vv;
        const double x2 = x * x;
        if (u3 < 1.0 - 0.0331 * x2 * x2 || log(u3) < 0.5 * x2 + d * (1.0 - vv + log(vv))) {
            res = d * vv;
            break;
        }
    }
    if (shape < 1.0) res *= pow(uboost, 1.0 / shape);
    return res;
}

// One workgroup per row (S gamma rows + 4 eta rows; its first wavefront does the work): lane g
// draws variate g, the row is normalised lane-parallel and writes its log-prior term to rowprior[row] (summed by finalize in a
// fixed order).  Rows are independent, so the launch fills S+4 CUs instead of one.
// do_s2: the gamma rows first run stage 2 of the aggregated mu/E pass for their sample (dsm_stage2.h, all 256 threads): the
// sums the draw needs never leave the workgroup's LDS and the iteration has one launch less.
struct DirParams {
    unsigned long long *sum_mu, *esum;
    int S, G;
    double alpha, delta, epsilon, lgc_gamma, lgc_eta;
    uint32_t k0, k1, iter;
    int zero_after;
    double *gamma_out, *gamma_trace, *eta_out, *eta_trace, *rowprior;
    int do_fin;
    FinalParams fin;
    int do_s2;                       // 0, or the version (2 / 3) of the aggregated specification stage 2 follows
    Stage2Params s2;
};

__device__ __forceinline__ void dirichlet_body(const DirParams &q, const S2Plan &plan)
{
    unsigned long long *__restrict__ sum_mu = q.sum_mu, *__restrict__ esum = q.esum;
    const int S = q.S, G = q.G;
    const double alpha = q.alpha, delta = q.delta, epsilon = q.epsilon, lgc_gamma = q.lgc_gamma, lgc_eta = q.lgc_eta;
    const uint32_t k0 = q.k0, k1 = q.k1, iter = q.iter;
    const int zero_after = q.zero_after;
    double *__restrict__ gamma_out = q.gamma_out, *__restrict__ gamma_trace = q.gamma_trace;
    double *__restrict__ eta_out = q.eta_out, *__restrict__ eta_trace = q.eta_trace, *__restrict__ rowprior = q.rowprior;
    const int do_fin = q.do_fin, do_s2 = q.do_s2;
    const FinalParams &fin = q.fin;
    const Stage2Params &s2 = q.s2;
    __shared__ __attribute__((aligned(16))) char smem_d[S2_SMEM_BYTES];   // stage 2 / finalize scratch (never both)
    if (do_fin && (int)blockIdx.x == S + 4) {            // extra workgroup: finalize the PREVIOUS iteration
        double *red = reinterpret_cast<double *>(smem_d), *redp = red + 256;
        int *flag = reinterpret_cast<int *>(redp + 256);
        finalize_body(fin, red, redp, flag, threadIdx.x, 256);
        return;
    }
    const int row = blockIdx.x;
    const bool is_gamma = row < S;
    const uint32_t *leaf = nullptr;
    if (do_s2 && is_gamma)                                                          // workgroup-uniform branches; both end with a barrier
        leaf = do_s2 >= 3 ? stage2_sample<3>(s2, plan, row, smem_d, false) : stage2_sample<2>(s2, plan, row, smem_d, false);
    if (!is_gamma) {
        // Esum = the context's [4][4] + the DSM_ESUM_PARTS copies the wavefronts of stage 1 added to (kernels_stats.hip: stats_agg_body,
        // round 6): this row needs Esum[., a] -- four counters of every copy, one per thread -- and leaves the copies zero for the next pass
        static_assert(DSM_ESUM_PARTS * 4 == 256, "one thread per (copy, observed base)");
        unsigned long long *ef = reinterpret_cast<unsigned long long *>(smem_d);
        if (threadIdx.x < 4) ef[threadIdx.x] = 0ull;
        __syncthreads();
        const int ob = threadIdx.x & 3, k = threadIdx.x >> 2;
        unsigned long long *pp = esum + 16 + k * 16 + ob * 4 + (row - S);
        const unsigned long long v = *pp;
        if (v) { *pp = 0ull; atomicAdd(&ef[ob], v); }
        __syncthreads();
    }
    S2_CLK(row, 6);
    if (threadIdx.x >= 64) return;                       // the draw needs one wavefront (no workgroup barriers below)
    const int lane = threadIdx.x;
    const int n = is_gamma ? G : 4;
    const int SG = S * G;
    double y = 0.0;
    if (lane < n) {
        double shape;
        uint32_t vid;                                   // variate id: the spec's flat index
        if (is_gamma) { vid = (uint32_t)(row * G + lane); shape = alpha + (double)(sum_mu[vid] + (leaf ? (unsigned long long)leaf[lane] : 0ull)); }
        else { const int a = row - S; vid = (uint32_t)(SG + a * 4 + lane); shape = delta + (double)(esum[lane * 4 + a] + reinterpret_cast<const unsigned long long *>(smem_d)[lane]); }
        y = gamma_variate(shape, vid, iter, k0, k1);
        if (zero_after) { if (is_gamma) sum_mu[row * G + lane] = 0ull; else esum[lane * 4 + (row - S)] = 0ull; }
    }
    // row normalisation, lane-parallel (inactive lanes carry 0)
    double x = y / group_allreduce_sum<64>(y);
    if (is_gamma) {
        if (lane < n && x < epsilon) x = epsilon;                       // HaploSNP_Sampler.py:271
        x = x / group_allreduce_sum<64>(lane < n ? x : 0.0);            // :272-273
    }
    const double a1 = is_gamma ? alpha : delta;
    const double lsum = group_allreduce_sum<64>(lane < n ? (a1 - 1.0) * log(x) : 0.0);
    if (lane == 0) rowprior[row] = (is_gamma ? lgc_gamma : lgc_eta) + lsum;
    if (lane < n) {
        if (is_gamma) { gamma_out[row * G + lane] = x; if (gamma_trace) gamma_trace[row * G + lane] = x; }
        else { eta_out[(row - S) * 4 + lane] = x; if (eta_trace) eta_trace[(row - S) * 4 + lane] = x; }
    }
    S2_CLK(row, 7);
}
#ifdef DSM_AB_SWITCHES
extern "C" int dsm_debug_s2_clocks(unsigned long long *out, int nrows)
{
    if (nrows > 1024) nrows = 1024;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(s2_clk), (size_t)nrows * 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return 8;
}
#endif

struct DirBatch { DirParams p[DSM_MAX_BATCH]; S2Plan plan; };        // chains of a batch have one G, hence one plan
__global__ __launch_bounds__(256) void dirichlet_kernel(DirParams q, S2Plan plan) { dirichlet_body(q, plan); }
__global__ __launch_bounds__(256) void dirichlet_kernel_b(DirBatch b) { dirichlet_body(b.p[blockIdx.y], b.plan); }
static_assert(sizeof(DirBatch) <= 4096, "kernarg segment");

// Dirichlet log-priors of a given (gamma, eta) -- entry state of update()
__global__ __launch_bounds__(256) void prior_kernel(const double *__restrict__ gamma, const double *__restrict__ eta,
                                                    int S, int G, double alpha, double delta, double lgc_gamma,
                                                    double lgc_eta, double *__restrict__ rowprior)
{
    for (int r = blockIdx.x * 256 + threadIdx.x; r < S + 4; r += gridDim.x * 256) {
        double lsum = 0.0;
        if (r < S) { for (int g = 0; g < G; ++g) lsum += (alpha - 1.0) * log(gamma[(size_t)r * G + g]); rowprior[r] = lgc_gamma + lsum; }
        else { for (int b = 0; b < 4; ++b) lsum += (delta - 1.0) * log(eta[(r - S) * 4 + b]); rowprior[r] = lgc_eta + lsum; }
    }
}

// =====================================================================
// A1 + A5: tau sweep with the log-likelihood epilogue.
// A group of LPV lanes (16/32/64) owns one variant; lane = sample (NSL samples
// per lane when S > LPV).  The 16 B count slab of (v,s) is one coalesced
// int4 load and stays in registers for the whole sweep; gamma is staged
// transposed in LDS ([G][SP], conflict-free across lanes), eta (sweep and
// likelihood versions) in LDS as well.  For each haplotype g, sequentially:
// the rest-mixture is accumulated h-ascending exactly as c_sample_tau.c:136-150,
// the four candidate log-probabilities are lane-partial sums of
// (float)count * log(p) reduced with an in-register butterfly, every lane
// normalises and inverts the CDF redundantly (no divergence), and the packed
// tau word is updated in a register.
// Each step first runs the fp32 screening pass (dsm_device.h: sweep_screen,
// DESIGN.md sec. 3d): when the four totals are far enough apart that the fp64
// draw is certain, the step is decided there; the fp64 evaluation above is the
// path of the remaining (< 1 % of the) steps.
// =====================================================================
struct TauParams {
    const int32_t *cnt_vs;
    uint64_t *tau;
    uint64_t *trace;          // may be null
    const double *gamma, *eta_sweep, *eta_ll;
    const uint32_t *u_raw;    // MT19937 words, [V*G]; null -> Philox
    double *logp;             // may be null: [V][G][4]
    double *ll_partial;       // [gridDim.x]
    const double *log_tab;    // [256][2]
    int *nchange;
    uint32_t *step_cnt;                // [gridDim.x][2] this launch's wavefront-steps: run / left to the fp64 code (~0: not screened)
    const uint32_t *screen_ctl;        // [0] != 0: the screening pass is suspended (finalize_body)
    int screen;                        // fp32 screening pass allowed (DESMAN_HIP_TAU_NO_SCREEN switches it off for A/B runs)
    int nt_skip;                       // experiment (DESMAN_HIP_NT_SKIP_TOTALS): a rare haplotype's step the difference screen left open goes to fp64 without the totals screen
    int V, S, G;
    int v_off;                // first position of this shard in the whole table (counter-based uniforms are keyed by global indices)
    uint32_t k0, k1, iter;
    const uint32_t *order;    // [blocks] which block of variants workgroup b works on (null: b) -- finalize_body: blk_order
    int do_fin;               // the last workgroup of the launch finalizes the PREVIOUS sweep (updateTau: no launch between
    FinalParams fin;          // two sweeps could carry it); it reads the other parity of ll_partial / nchange
};

// which shapes of the sweep run its register-lean form (tau_body: LEAN): three, six and eight samples per lane of a 32- or 64-lane group.
// Measured against the fp64-prefix form on one box (scripts/dbg/lean_ab.sh), ms per iteration: 32 x 3 0.722 vs 0.744 (50k x 96 x 12),
// 64 x 3 0.237 vs 0.243, 64 x 6 0.711 vs 0.819 (10k x 300 x 8), 64 x 8 0.570 vs 0.626 (5k x 512 x 8); it loses at 16 x 3 (0.167 vs 0.163)
// and 64 x 4 (0.551 vs 0.530), and at two samples per lane.
#ifdef TAU_NO_LEAN
#define TAU_LEAN(LPV, NSL) false
#else
#ifdef TAU_LEAN2          /* experiment: the lean form at two samples per lane too, four wavefronts per SIMD */
#define TAU_LEAN(LPV, NSL) ((LPV) >= 32 && ((NSL) == 2 || (NSL) == 3 || (NSL) == 6 || (NSL) == 8))
#else
#define TAU_LEAN(LPV, NSL) ((LPV) >= 32 && ((NSL) == 3 || (NSL) == 6 || (NSL) == 8))
#endif
#endif

// NT: the instantiation with the near-tie screen (dsm_device.h: sweep_neartie_core) for chains that carry haplotypes rare in every
// sample -- chosen per call by k_tau_sweep from the chain's own abundances (dsm_host.h: tau_neartie_on).  Both instantiations make the
// draws of the fp64 code, so which one runs is a matter of speed only.
template <int LPV, int NSL, bool SWEEP, bool LL, bool NT = false>
__device__ __forceinline__ void tau_body(const TauParams &p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_t[];
    const int nblk = (int)gridDim.x - p.do_fin;
    if (p.do_fin && (int)blockIdx.x == nblk) {
        double *fr = reinterpret_cast<double *>(smem_t);             // >= 4.4 KB of dynamic LDS: red[256], redp[256], flag
        finalize_body(p.fin, fr, fr + 256, reinterpret_cast<int *>(fr + 512), threadIdx.x, 256);
        return;
    }
    constexpr bool LEAN = TAU_LEAN(LPV, NSL) && SWEEP;       // the register-lean form of the sweep (below)
    const int bid = p.order ? (int)p.order[blockIdx.x] : (int)blockIdx.x;       // the block of variants this workgroup works on
    constexpr int SP = LPV * NSL;
    constexpr int GPB = 256 / LPV;
    double *gT = reinterpret_cast<double *>(smem_t);   // [G][SP]
    double *eS = gT + (size_t)p.G * SP;                  // [16]
    double *eL = eS + 16;                                // [16]
    double *red = eL + 16;                               // [4]
    int *redi = reinterpret_cast<int *>(red + 4);        // [4] (+4 pad)
    double2 *ltab = reinterpret_cast<double2 *>(red + 6);// [256] log table
    float *gT32 = reinterpret_cast<float *>(ltab + DSM_LOG_TAB_N);   // [G][SP] fp32 copies for the screening pass
    float *eS32 = gT32 + (size_t)p.G * SP;               // [16] eta_sweep, [4] its column minima; NT: [4] its column maxima (rounded up)
    uint32_t *gmaxb = reinterpret_cast<uint32_t *>(eS32 + 24);       // NT: [32] bits of max_s (float)gamma_sg (positive floats order like their bits)
    const int tid = threadIdx.x, G = p.G, S = p.S;
    if (tid < DSM_LOG_TAB_N) ltab[tid] = reinterpret_cast<const double2 *>(p.log_tab)[tid];
    if constexpr (NT) {
        if (tid < 32) gmaxb[tid] = 0u;
        __syncthreads();
    }
    for (int i = tid; i < G * SP; i += 256) {
        const int g = i / SP, s = i % SP;
        const double x = (s < S) ? p.gamma[(size_t)s * G + g] : 1.0;   // pad: p > 0, count = 0
        gT[i] = x;
        if (SWEEP) gT32[i] = (float)x;
        if constexpr (NT) { if (s < S && g < 32) atomicMax(&gmaxb[g], __float_as_uint((float)x)); }
    }
    if (tid < 16) { eS[tid] = p.eta_sweep[tid]; eL[tid] = p.eta_ll[tid]; if (SWEEP) eS32[tid] = (float)p.eta_sweep[tid]; }
    if (SWEEP && tid < 4) {
        float m = (float)p.eta_sweep[tid], mx = m;
        for (int a = 1; a < 4; ++a) { m = fminf(m, (float)p.eta_sweep[a * 4 + tid]); mx = fmaxf(mx, (float)p.eta_sweep[a * 4 + tid]); }
        eS32[16 + tid] = m;
        if constexpr (NT) eS32[20 + tid] = mx * 1.000001f;          // (it enters an error bound: sweep_neartie_core)
    }
    __syncthreads();

    const int grp = tid / LPV, lig = tid % LPV;
    double ll_acc = 0.0;
    int nchg = 0;
    int n_steps = 0, n_exact = 0;                        // wave-uniform
    const bool screen_on = SWEEP && p.logp == nullptr && p.screen && *p.screen_ctl == 0u;

    for (int v = bid * GPB + grp; v < p.V; v += nblk * GPB) {
        ISA_MARK("variant_setup");
        uint64_t t = p.tau[v];
        // LEAN keeps the counts of the sweep as (float)count only (12 registers instead of 24 at three samples per lane: the
        // screening pass multiplies by exactly that, the rare fp64 step by (double)(float)count, c_sample_tau.c:164) and the
        // likelihood epilogue, which needs the integers, loads the slab again -- an L2 hit a few microseconds after the first
        // load.  With both forms live across the haplotype loop the compiler spilled per variant (round 3: 94 MB of scratch
        // write-back per sweep at 50k x 96 x 12).
        int xi[LEAN ? 1 : NSL][4];
        float xs[LEAN ? NSL : 1][4];
        double xf[LEAN ? 1 : NSL][4];
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            int ligl = lig;
            if constexpr (LEAN) asm volatile("" : "+v"(ligl));      // the slab's address is formed afresh, not kept in a register pair
            const int s = ligl + j * LPV;
            int4 c = make_int4(0, 0, 0, 0);
            if (s < S) c = reinterpret_cast<const int4 *>(p.cnt_vs)[(size_t)v * S + s];
            if constexpr (LEAN) {
                xs[j][0] = (float)c.x; xs[j][1] = (float)c.y; xs[j][2] = (float)c.z; xs[j][3] = (float)c.w;
            } else {
                xi[j][0] = c.x; xi[j][1] = c.y; xi[j][2] = c.z; xi[j][3] = c.w;
#pragma unroll
                for (int b = 0; b < 4; ++b) xf[j][b] = (double)(float)xi[j][b];   // c_sample_tau.c:164
            }
        }
        if (SWEEP) {
            double l_cur = 0.0;                  // log-prob of the variant's current configuration
            // The rest mixture of step g is the h-ascending FMA chain over h != g (c_sample_tau.c:136-150).  Its
            // first g links use haplotypes that are already re-drawn and final, so that prefix is carried from
            // step to step (pre) and only the links h > g are re-done: the same operations in the same order --
            // the same sums -- for G(G+1)/2 instead of G(G-1) links per variant.  (The links are fma(eta, gamma, acc)
            // where c_sample_tau.c:143-149 multiplies and adds: the sums agree with the reference to rounding, not bit
            // for bit; see DESIGN.md sec. 4 for what that means for the draws.)
            // LEAN (three, six, eight samples per lane of a 32- or 64-lane group): the prefix is carried in fp32 only -- all the screening
            // pass reads -- the counts are kept as (float)count, and the rare fp64 step re-does its links h < g (same operations, same
            // order, same sums), converts its counts at each use and evaluates one candidate at a time: with the fp64 prefix and the
            // unrolled fp64 step live, three samples per lane need 219 VGPRs = two wavefronts per SIMD; lean they fit 160 = three, with
            // NO scratch (round 3's form spilled 23 registers and stored two of them per variant: 94 MB of scratch write-back per sweep
            // at 50k x 96 x 12; profiles/r04_kernel_regs.txt).  Same box, lean vs not, ms per iteration: 50k x 96 x 12 0.683 (round 3's
            // lean form 0.722) vs 0.744, x 96 x 6 0.480 vs 0.508, 10k x 192 x 8 0.225 vs 0.243.  At two samples per lane the lean form at
            // FOUR wavefronts per SIMD (-DTAU_LEAN2: 128 VGPRs, the loop's accumulators spilled per variant) measures 104.2 vs 103.1 us
            // per iteration at config 3 (sweep 39.2 vs 37.7 us): not used; at 16 lanes per variant (S <= 48: 0.167 vs 0.163) the same
            // trade loses as well.
            double pre[LEAN ? 1 : NSL][4];
            dsm_f2 pre32[LEAN ? NSL : 1][2];
#pragma unroll
            for (int j = 0; j < NSL; ++j) {
                if constexpr (LEAN) { pre32[j][0] = (dsm_f2){0.0f, 0.0f}; pre32[j][1] = (dsm_f2){0.0f, 0.0f}; }
                else {
#pragma unroll
                    for (int b = 0; b < 4; ++b) pre[j][b] = 0.0;
                }
            }
            bool have_cur = false;               // l_cur is the log-probability of the current configuration (wave-uniform)
            n_steps += G;
            const bool screen = screen_on;
            // the variant's reads over all its samples, rounded up: the error bound of the screening pass scales with it (screen_certify)
            float xtot = 0.0f;
            if (screen) {
#pragma unroll
                for (int j = 0; j < NSL; ++j)
#pragma unroll
                    for (int b = 0; b < 4; ++b) { if constexpr (LEAN) xtot += xs[j][b]; else xtot += (float)xi[j][b]; }
                float z1 = 0.0f, z2 = 0.0f, z3 = 0.0f;
                group_allreduce_sum4_f32<LPV>(xtot, z1, z2, z3);
                xtot *= 1.0001f;
            }
            for (int g = 0; g < G; ++g) {
                double gg[NSL];
                if constexpr (!LEAN) {
#pragma unroll
                    for (int j = 0; j < NSL; ++j) gg[j] = gT[g * SP + lig + j * LPV];
                }
                ISA_MARK("step_setup");
                const int told = (int)((t >> (2 * g)) & 3);
                uint32_t uw;
                const size_t ui = (size_t)v * G + g;
                if (p.u_raw) {
                    uw = p.u_raw[ui];
                } else {
                    uint32_t r[4];
                    const size_t ug = ui + (size_t)p.v_off * G;
                    philox4x32_10((uint32_t)ug, (uint32_t)(ug >> 32), p.iter, DSM_STREAM_TAUU, p.k0, p.k1, r);
                    uw = r[0];
                }
                int tn = 0;
                bool decided = false;
                // NT: a haplotype that is rare in every sample (max_s gamma_sg <= 0.01: the spare haplotypes of an over-fitted chain) makes
                // near-ties, which the totals below cannot settle: its step is screened on the differences of the candidates first;
                // what that leaves open is tried on the totals and then goes to the fp64 code like any other step.  Known per haplotype,
                // wave-uniform.
                bool rare = false;
                if constexpr (NT) rare = screen && g < 32 && __builtin_amdgcn_readfirstlane(gmaxb[g]) <= __builtin_bit_cast(uint32_t, (float)DSM_NT_RARE);
                if constexpr (NT) {
                    if (rare) {
                        int tf = 0;
                        bool c2;
                        if constexpr (LEAN) c2 = sweep_neartie32<LPV, NSL>(pre32, xs, t, g, G, lig, uw, gT32, eS32, tf);
                        else c2 = sweep_neartie<LPV, NSL>(pre, xi, t, g, G, lig, uw, gT32, eS32, tf);
                        if (__builtin_amdgcn_ballot_w64(c2) == __builtin_amdgcn_ballot_w64(true)) { tn = tf; decided = true; }
                    }
                }
                if (screen && !decided && !(NT && rare && p.nt_skip)) {
                    // ---- screening pass in fp32 (hardware log2): the four candidate log-probabilities to ~1e-6 relative.
                    // If the best one leads every other by more than 64 + 2^-13 |l| (natural units; the fp32 error is below
                    // 0.1 + 1e-6 |l|), the fp64 evaluation below would find exp(l_a - l_best) < e^-30 for the others, and its
                    // draw is then `best` for every uniform word except 0 (sum <= 1 + 3e-13, u <= 1 - 2^-32: u sum < 1 =
                    // the best candidate's CDF edge; u sum >= the edges below it as soon as u >= 2^-32).  Such a step costs
                    // ~850 issue cycles instead of ~3700; anything else -- a closer race, a zero word, a mixture component
                    // outside fp32's normal range, NaN -- is decided by the fp64 code.  More than 99 % of the steps of a
                    // converged chain and ~97 % right after the NMFT initialisation take the short way.
                    int best = 0;
                    bool cert;
                    if constexpr (LEAN) cert = sweep_screen32<LPV, NSL>(pre32, xs, t, g, G, lig, uw, gT32, eS32, xtot, best);
                    else cert = sweep_screen<LPV, NSL>(pre, xi, t, g, G, lig, uw, gT32, eS32, xtot, best);
                    if (__builtin_amdgcn_ballot_w64(cert) == __builtin_amdgcn_ballot_w64(true)) { tn = best; decided = true; }
                }
                ISA_MARK("step_fp64");
                if (!decided) {
                const bool reuse = have_cur;
                double l[4];
                if constexpr (LEAN) {
                double st[NSL][4];
#pragma unroll
                for (int j = 0; j < NSL; ++j) {
                    gg[j] = gT[g * SP + lig + j * LPV];
#pragma unroll
                    for (int b = 0; b < 4; ++b) st[j][b] = 0.0;
                }
#pragma unroll 1
                for (int h = 0; h < G; ++h) {
                    if (h == g) continue;
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
                // The candidate a == told is the variant's current configuration; when its log-probability is known from
                // the previous step (the candidate chosen there, evaluated in fp64) only the three others need their
                // 4*NSL logs.  One candidate at a time (a loop that is not unrolled): this path is rare, and what it keeps live
                // decides whether the whole sweep runs three or four wavefronts per SIMD.
                if constexpr (LPV == 64) {
                    // one variant per wavefront: told is wave-uniform, the current candidate is branched around
                    double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
#pragma unroll 1
                    for (int a = 0; a < 4; ++a) {
                        if (reuse && a == told) continue;
                        const double c = sweep_candidate_x<NSL>(a, xs, st, gg, eS, ltab, lig, LPV, S);
                        c0 = (a == 0) ? c : c0; c1 = (a == 1) ? c : c1; c2 = (a == 2) ? c : c2; c3 = (a == 3) ? c : c3;
                    }
                    l[0] = c0; l[1] = c1; l[2] = c2; l[3] = c3;
                    group_allreduce_sum4<LPV>(l[0], l[1], l[2], l[3]);
                } else {
                    // several variants per wavefront, each with its own current base: every group evaluates its
                    // candidates in the rotated order told+1, told+2, told+3 (no divergence), the sums are put
                    // back in base order afterwards (pure data movement: same values)
                    const int rot = reuse ? told + 1 : 0;
                    double cv[4] = {0.0, 0.0, 0.0, 0.0};
                    const int ncand = reuse ? 3 : 4;                // wave-uniform
#pragma unroll 1
                    for (int i = 0; i < ncand; ++i) {
                        const double c = sweep_candidate_x<NSL>((rot + i) & 3, xs, st, gg, eS, ltab, lig, LPV, S);
                        cv[0] = (i == 0) ? c : cv[0]; cv[1] = (i == 1) ? c : cv[1]; cv[2] = (i == 2) ? c : cv[2]; cv[3] = (i == 3) ? c : cv[3];
                    }
                    group_allreduce_sum4_unrotate<LPV>(cv, rot, l);
                }
                } else {
                double st[NSL][4];
#pragma unroll
                for (int j = 0; j < NSL; ++j)
#pragma unroll
                    for (int b = 0; b < 4; ++b) st[j][b] = pre[j][b];
#pragma unroll 4
                for (int h = g + 1; h < G; ++h) {
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
                // The candidate a == told is the variant's current configuration; when its log-probability is known from
                // the previous step (the candidate chosen there, evaluated in fp64) only the three others need their
                // 4*NSL logs.
                if constexpr (LPV == 64) {
                    // one variant per wavefront: told is wave-uniform, the current candidate is branched around
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        l[a] = 0.0;
                        if (!(reuse && a == told)) l[a] = sweep_candidate<NSL>(a, xf, st, gg, eS, ltab, lig, LPV, S);
                    }
                    group_allreduce_sum4<LPV>(l[0], l[1], l[2], l[3]);
                } else {
                    // several variants per wavefront, each with its own current base: every group evaluates its
                    // candidates in the rotated order told+1, told+2, told+3 (no divergence), the sums are put
                    // back in base order afterwards (pure data movement: same values)
                    const int rot = reuse ? told + 1 : 0;
                    double cv[4];
#pragma unroll
                    for (int i = 0; i < 3; ++i) cv[i] = sweep_candidate<NSL>((rot + i) & 3, xf, st, gg, eS, ltab, lig, LPV, S);
                    cv[3] = 0.0;
                    if (!reuse) cv[3] = sweep_candidate<NSL>(3, xf, st, gg, eS, ltab, lig, LPV, S);       // wave-uniform
                    group_allreduce_sum4_unrotate<LPV>(cv, rot, l);
                }
                }
                if (reuse) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) if (a == told) l[a] = l_cur;
                }
                if (p.logp && lig == 0) {
                    double *o = p.logp + ((size_t)v * G + g) * 4;
                    o[0] = l[0]; o[1] = l[1]; o[2] = l[2]; o[3] = l[3];
                }
                tn = sweep_draw(l, uw);
                l_cur = (tn == 0) ? l[0] : (tn == 1) ? l[1] : (tn == 2) ? l[2] : l[3];
                }
                ISA_MARK("step_commit");
                have_cur = !decided;
                n_exact += !decided;
                nchg += (lig == 0) & (tn != told);
                t = (t & ~(3ull << (2 * g))) | ((uint64_t)tn << (2 * g));
                if constexpr (LEAN) {                                // link g of the chain, with the new base
                    const dsm_f2 *er = reinterpret_cast<const dsm_f2 *>(eS32 + tn * 4);
                    const dsm_f2 e01 = er[0], e23 = er[1];
#pragma unroll
                    for (int j = 0; j < NSL; ++j) {
                        const float gm = gT32[g * SP + lig + j * LPV];
                        const dsm_f2 gm2 = (dsm_f2){gm, gm};
                        pre32[j][0] = __builtin_elementwise_fma(e01, gm2, pre32[j][0]);
                        pre32[j][1] = __builtin_elementwise_fma(e23, gm2, pre32[j][1]);
                    }
                } else {
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
            if constexpr (LEAN) {
                int vv = v;
                asm volatile("" : "+v"(vv));                 // &tau[v] formed afresh: it was spilled across the haplotype loop
                if (lig == 0) p.tau[vv] = t;
            } else if (lig == 0) p.tau[v] = t;
        }
        ISA_MARK("variant_ll");
        if (lig == 0 && p.trace) p.trace[v] = t;
        if (LL) {
#pragma unroll
            for (int j = 0; j < NSL; ++j) {
                int xl[4];                                   // this slab's counts as integers
                if constexpr (LEAN) {
                    int vv = v, ligl = lig;
                    asm volatile("" : "+v"(vv), "+v"(ligl)); // a second load, not the first one (or its address) kept alive across the sweep
                    const int s = ligl + j * LPV;
                    int4 c = make_int4(0, 0, 0, 0);
                    if (s < S) c = reinterpret_cast<const int4 *>(p.cnt_vs)[(size_t)vv * S + s];
                    xl[0] = c.x; xl[1] = c.y; xl[2] = c.z; xl[3] = c.w;
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b) xl[b] = xi[j][b];
                }
                double P[4] = {0.0, 0.0, 0.0, 0.0};
                for (int g = 0; g < G; ++g) {
                    const double *er = eL + (int)((t >> (2 * g)) & 3) * 4;
                    const double gm = gT[g * SP + lig + j * LPV];
#pragma unroll
                    for (int b = 0; b < 4; ++b) P[b] = fma(gm, er[b], P[b]);
                }
                bool ok = true;
#pragma unroll
                for (int b = 0; b < 4; ++b) ok &= dsm_log_ok(P[b]);
                if (__builtin_expect(ok, 1)) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) ll_acc = fma((double)xl[b], dsm_log_core(P[b], ltab), ll_acc);
                } else if (lig + j * LPV < S) {                 // (a padded slot adds nothing: see sweep_candidate, dsm_device.h)
#pragma unroll
                    for (int b = 0; b < 4; ++b) ll_acc = fma((double)xl[b], dsm_log_slow(P[b]), ll_acc);
                }
            }
        }
    }
    ISA_MARK("epilogue");
    // workgroup reduction, fixed order -> deterministic ll
    const double wsum = group_allreduce_sum<64>(ll_acc);
    const int wn = (int)group_allreduce_sum_u32<64>((unsigned)nchg);
    __shared__ uint32_t s_cnt[8];                         // wavefront-steps run / left to the fp64 code, per wavefront
    if ((tid & 63) == 0) { red[tid >> 6] = wsum; redi[tid >> 6] = wn; s_cnt[tid >> 6] = (uint32_t)n_steps; s_cnt[4 + (tid >> 6)] = (uint32_t)n_exact; }
    __syncthreads();
    if (tid == 0) {
        const uint32_t st_sum = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3], ex_sum = s_cnt[4] + s_cnt[5] + s_cnt[6] + s_cnt[7];
        p.step_cnt[2 * bid] = st_sum;
        p.step_cnt[2 * bid + 1] = screen_on ? ex_sum : (SWEEP ? 0xFFFFFFFFu : 0u);
        p.ll_partial[bid] = ((red[0] + red[1]) + red[2]) + red[3];
        const int tot = redi[0] + redi[1] + redi[2] + redi[3];
        if (SWEEP && tot) atomicAdd(p.nchange, tot);
    }
}

// three samples per lane, 32 or 64 lanes per variant: the lean form of the sweep at three wavefronts per SIMD (tau_body: LEAN)
#define TAU_MIN_WGS(LPV, NSL) (TAU_LEAN(LPV, NSL) ? ((NSL) == 2 ? 4 : (NSL) == 3 ? 3 : 2) : 1)
template <int LPV, int NSL, bool SWEEP, bool LL>
__global__ __launch_bounds__(256, TAU_MIN_WGS(LPV, NSL)) void tau_kernel(TauParams p) { tau_body<LPV, NSL, SWEEP, LL>(p); }
template <int LPV, int NSL>
__global__ __launch_bounds__(256, TAU_MIN_WGS(LPV, NSL)) void tau_kernel_nt(TauParams p) { tau_body<LPV, NSL, true, true, true>(p); }
// K chains of one shape, chain = blockIdx.y (dsm_host.h: BatchCtl)
template <int LPV, int NSL, bool SWEEP, bool LL>
__global__ __launch_bounds__(256, TAU_MIN_WGS(LPV, NSL)) void tau_kernel_b(BatchArgs<TauParams> b) { tau_body<LPV, NSL, SWEEP, LL>(b.p[blockIdx.y]); }
static_assert(sizeof(BatchArgs<TauParams>) <= 4096, "kernarg segment");

// test hook: the hardware log2 the screening pass relies on (its error bound is pinned by tests/test_gpu_edges.py)
__global__ __launch_bounds__(256) void log2f_test_kernel(const float *__restrict__ in, float *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = __builtin_amdgcn_logf(in[i]);
}

int k_log2f_test(dsm_ctx *c, const float *d_in, float *d_out, size_t n)
{
    hipLaunchKernelGGL(log2f_test_kernel, dim3(1024), dim3(256), 0, c->stream, d_in, d_out, n);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

// =====================================================================
// host launchers
// =====================================================================
int k_convert_counts(dsm_ctx *c, const int64_t *d_in, int *d_flag, double *d_partial, int nblk, unsigned long long *d_depth)
{
    hipLaunchKernelGGL(convert_counts_kernel, dim3(nblk), dim3(256), 0, c->stream, d_in, c->cnt_vs, c->V, c->S, d_flag,
                       d_partial, d_depth);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

int k_pack_tau(dsm_ctx *c, const int64_t *d_onehot, uint64_t *d_packed, int V, int G)
{
    hipLaunchKernelGGL(pack_tau_kernel, dim3((V + 255) / 256), dim3(256), 0, c->stream, d_onehot, d_packed, V, G);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

int k_unpack_tau(dsm_ctx *c, const uint64_t *d_packed, int64_t *d_onehot, int V, int G)
{
    const size_t n = (size_t)V * G;
    hipLaunchKernelGGL(unpack_tau_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_packed,
                       d_onehot, V, G);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

int k_tau_sum(dsm_ctx *c, const uint64_t *trace, int n, int64_t *d_sum)
{
    const size_t m = (size_t)c->V * c->G;
    hipLaunchKernelGGL(tau_sum_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, trace, n, c->V,
                       c->G, d_sum);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

int k_mt_fill(dsm_ctx *c, uint32_t *out, size_t n, hipStream_t stream)
{
    if (n == 0) return DSM_OK;
    KTimer tm(c, DSM_K_MT, stream);
    static const bool plain = DSM_AB_ENV("DESMAN_HIP_MT_PLAIN") != nullptr;        // A/B switch: the 227-words-per-step kernel
    if (plain) hipLaunchKernelGGL(mt_fill_kernel, dim3(1), dim3(256), 0, stream, c->mt_state, out, n);
    else {
        // The generator is one workgroup that runs next to the main stream's kernels.  It asks for (nearly) all of a CU's
        // LDS, which it does not use, so that no other workgroup is placed on its CU: a workgroup sharing a CU with these
        // 16 high-priority wavefronts runs 2-3x longer and becomes the tail of its launch (stage 1 of the mu/E pass, whose
        // wavefronts all get the same number of tasks: 73 -> 57 us; DESMAN_HIP_MT_HOG=0 switches the reservation off).
        static const int hog = DSM_AB_ENV("DESMAN_HIP_MT_HOG") ? atoi(DSM_AB_ENV("DESMAN_HIP_MT_HOG")) : 140;   // KB
        if (hog && !c->mt_attr_set) {                                  // per device, hence per context
            HIP_TRY(hipFuncSetAttribute((const void *)mt_fill_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, hog * 1024));
            c->mt_attr_set = true;
        }
        // long fills (chunks of sweeps of a large table): from several CUs (mt_fill_parallel above); what is left, if anything, serially
        static const bool no_par = DSM_AB_ENV("DESMAN_HIP_MT_SERIAL") != nullptr;        // A/B switch
        static const size_t par_min = DSM_AB_ENV("DESMAN_HIP_MT_PAR_MIN") ? (size_t)atoi(DSM_AB_ENV("DESMAN_HIP_MT_PAR_MIN")) : 6;   // A/B switch, in chunks of D words (config 3, 80 000 words per sweep, never gets there: its generator hides behind the iteration, and five CUs taken at once cost it 0.103 -> 0.106-0.112 ms per iteration)
        if (!g_batch.K && !no_par && n >= par_min * MTJ_D) {
            size_t made = 0;
            { const int r = mt_fill_parallel(c, out, n, stream, &made); if (r != DSM_OK) return r; }
            out += made; n -= made;
            if (n == 0) return DSM_OK;
        }
        if (g_batch.K) {
            static thread_local BatchArgs<MtArgs> acc;
            acc.p[g_batch.k] = MtArgs{c->mt_state, out, n};
            if (g_batch.k == g_batch.K - 1)          // no CU reservation here: small tables, K generators
                hipLaunchKernelGGL(mt_fill_wide_kernel_b, dim3(g_batch.K), dim3(1024), 0, stream, acc);
        } else
            hipLaunchKernelGGL(mt_fill_wide_kernel, dim3(1), dim3(1024), (size_t)hog * 1024, stream, c->mt_state, out, n);
    }
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

int k_stats_v1(dsm_ctx *c, uint32_t iter)
{
    if (!c->items_built) { int r = build_stats_items(c); if (r != DSM_OK) return r; }
    KTimer tm(c, DSM_K_STATS);
    if (c->max_items == 0) return DSM_OK;
    // the per-read loop issues two instructions per threshold slot: instantiate it for the exact
    // haplotype count (every G up to 8, then in steps of 2 and 4) instead of padding to a power of two
    static const int sizes[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32};
    int gm = 32;
    for (int z : sizes) if (c->G <= z) { gm = z; break; }
    const void *fn = nullptr;
    const bool exact = (c->G == gm);
#define STATS_FN(GM) if (gm == GM) fn = exact ? (const void *)stats_kernel<GM, true> : (const void *)stats_kernel<GM, (GM <= 8)>
    STATS_FN(1); STATS_FN(2); STATS_FN(3); STATS_FN(4); STATS_FN(5); STATS_FN(6); STATS_FN(7); STATS_FN(8);
    STATS_FN(10); STATS_FN(12); STATS_FN(14); STATS_FN(16); STATS_FN(20); STATS_FN(24); STATS_FN(28); STATS_FN(32);
#undef STATS_FN
    const int key = gm * 2 + (exact ? 1 : 0);             // the two variants can differ in occupancy
    if (c->blk_gmax != key) {
        // size the grid to exactly the resident workgroups and share them among the samples by depth
        int occ = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 256, 0));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, c->device));
        if (occ < 1) occ = 1;
        const int S = c->S, budget = std::max(S, occ * prop.multiProcessorCount);
        double tot = 0.0;
        for (int s = 0; s < S; ++s) tot += (double)c->depth[s];
        std::vector<int32_t> nb(S, 1);
        int used = S;
        if (tot > 0)
            for (int s = 0; s < S; ++s) {
                int extra = (int)((double)(budget - S) * (double)c->depth[s] / tot);
                const int most = (c->nitems_h[s] + 255) / 256;            // no more workgroups than item chunks
                if (1 + extra > most) extra = std::max(0, most - 1);
                nb[s] += extra; used += extra;
            }
        std::vector<int32_t> tab;
        tab.reserve((size_t)used * 3);
        // interleave the samples so that neighbouring workgroup ids belong to different samples
        int maxnb = 0;
        for (int s = 0; s < S; ++s) maxnb = std::max(maxnb, nb[s]);
        for (int j = 0; j < maxnb; ++j)
            for (int s = 0; s < S; ++s)
                if (j < nb[s]) { tab.push_back(s); tab.push_back(j); tab.push_back(nb[s]); }
        if (c->blk_tab) { (void)hipFree(c->blk_tab); c->blk_tab = nullptr; }
        HIP_TRY(hipMalloc((void **)&c->blk_tab, tab.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpyAsync(c->blk_tab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->blk_n = used;
        c->blk_gmax = key;
    }
    const dim3 grid(c->blk_n), block(256);
    const uint32_t k0 = (uint32_t)c->ctr_seed, k1 = (uint32_t)(c->ctr_seed >> 32);
#define LAUNCH_STATS(GM, EX)                                                                                          \
    hipLaunchKernelGGL((stats_kernel<GM, EX>), grid, block, 0, c->stream, reinterpret_cast<const int2 *>(c->items),   \
                       c->nitems, c->blk_tab, c->tau, c->gamma, c->eta,                                               \
                       c->V, c->S, c->G, c->item_stride, c->chunked ? 1 : 0, k0, k1, iter, c->sum_mu, c->esum)
#define STATS_CASE(GM) if (gm == GM) { if (exact) LAUNCH_STATS(GM, true); else LAUNCH_STATS(GM, (GM <= 8)); }
    STATS_CASE(1); STATS_CASE(2); STATS_CASE(3); STATS_CASE(4); STATS_CASE(5); STATS_CASE(6); STATS_CASE(7);
    STATS_CASE(8); STATS_CASE(10); STATS_CASE(12); STATS_CASE(14); STATS_CASE(16); STATS_CASE(20); STATS_CASE(24);
    STATS_CASE(28); STATS_CASE(32);
#undef STATS_CASE
#undef LAUNCH_STATS
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

static void dirichlet_consts(const dsm_ctx *c, double *lgc_gamma, double *lgc_eta)
{
    *lgc_gamma = lgamma(c->alpha * c->G) - c->G * lgamma(c->alpha);
    *lgc_eta = lgamma(c->delta * 4) - 4 * lgamma(c->delta);
}

static FinalParams make_final(dsm_ctx *c, int nblocks, int it, int star_mode, const double *prior, const double *gamma_src,
                              const double *eta_src, int slot = 0)
{
    FinalParams p;
    p.ll_partial = c->ll_partial + (size_t)slot * DSM_MAX_GRID; p.nblocks = nblocks;
    p.ll_const = c->ll_const;
    p.tau_prior = (double)(c->shard_on ? c->shard_vtot : c->V) * (double)c->G * log(1.0 / 4.0);     // HaploSNP_Sampler.py:457
    p.ll_given = c->shard_on ? c->shard_vec : nullptr;
    p.prior = prior; p.S = c->S; p.nchange = c->nchange + slot; p.it = it;
    p.ll_trace = c->ll_trace; p.lp_trace = c->lp_trace; p.nchange_trace = c->nchange_trace;
    p.star = c->star; p.gamma_src = gamma_src; p.gamma_star = c->gamma_star; p.SG = c->S * c->G;
    p.eta_src = eta_src; p.eta_star = c->eta_star;
    p.star_mode = star_mode; p.scalars = c->scalars;
    // the suspension word of the finalized launch's parity: the launch this step may ride in (updateTau) reads the OTHER word,
    // so no launch reads a word it writes and the screening decisions are the same on every run
    p.step_cnt = c->step_cnt + (size_t)slot * 2 * DSM_MAX_GRID; p.sweep_stats = c->sweep_stats; p.screen_ctl = c->screen_ctl + slot;
    static const bool order_on = !(getenv("DESMAN_HIP_TAU_ORDER") && atoi(getenv("DESMAN_HIP_TAU_ORDER")) == 0);      // A/B switch
    p.blk_order = nullptr;
    p.rare_out = nullptr; p.G = c->G;
    // (a launch whose workgroups are all resident at once has no tail to move the long steps out of: the order is left alone and the finalize
    // step stays short -- at config 2 it rides in a 8 us Dirichlet launch)
    if (order_on && c->blk_order && nblocks > 0 && nblocks <= DSM_MAX_GRID) {
        if (c->tau_resident_key != c->S * 64 + c->G) {
            int launched = 0, resident = 0;
            if (tau_launch_info(c, &launched, &resident) == DSM_OK) { c->tau_resident = resident; c->tau_resident_key = c->S * 64 + c->G; }
        }
        if (c->tau_resident > 0 && nblocks > c->tau_resident) { p.blk_order = c->blk_order + (size_t)slot * DSM_MAX_GRID; c->blk_order_n[slot] = nblocks; }
        else c->blk_order_n[slot] = 0;
    }
    return p;
}

// fin_it >= 0: the launch also finalizes iteration fin_it (traces of that iteration are its gamma/eta source)
int k_dirichlet(dsm_ctx *c, uint32_t iter, double *gamma_out, double *gamma_trace, double *eta_out, double *eta_trace,
                double *prior_out, int fin_it, int fin_nblocks, const double *fin_prior, int do_s2)
{
    KTimer tm(c, DSM_K_DIRICH);
    double lg, le;
    dirichlet_consts(c, &lg, &le);
    const uint32_t k0 = (uint32_t)c->ctr_seed, k1 = (uint32_t)(c->ctr_seed >> 32);
    const int do_fin = fin_it >= 0 ? 1 : 0;
    FinalParams fin = {};
    if (do_fin)
        fin = make_final(c, fin_nblocks, fin_it, 0, fin_prior, c->gamma_trace + (size_t)fin_it * c->S * c->G,
                         c->eta_trace + (size_t)fin_it * 16);
    Stage2Params s2 = {};
    if (do_s2) {
        // stage 2 splits the subset counts with the gamma the mu/E pass used = the resident one; gamma_out may be the same
        // buffer: workgroup s stages row s in LDS before it writes the new row s, and no other workgroup reads that row
        s2.ntab = c->ntab; s2.rep = c->ntab_rep; s2.ld = c->ntab_ld; s2.gamma = c->gamma; s2.sum_mu = c->sum_mu; s2.log_tab = c->log_tab;
        s2.S = c->S; s2.G = c->G; s2.k0 = k0; s2.k1 = k1; s2.iter = iter; s2.hmul = stats_ntab_hmul(); s2.swz = stats_ntab_swz();
        s2.big_count = c->big_count;
    }
    DirParams q;
    q.sum_mu = c->sum_mu; q.esum = c->esum; q.S = c->S; q.G = c->G;
    q.alpha = c->alpha; q.delta = c->delta; q.epsilon = c->epsilon; q.lgc_gamma = lg; q.lgc_eta = le;
    q.k0 = k0; q.k1 = k1; q.iter = iter; q.zero_after = 1;
    q.gamma_out = gamma_out; q.gamma_trace = gamma_trace; q.eta_out = eta_out; q.eta_trace = eta_trace; q.rowprior = prior_out;
    q.do_fin = do_fin; q.fin = fin; q.do_s2 = do_s2 ? stats_draw_version(stats_spec(c)) : 0; q.s2 = s2;
    if (g_batch.K == 0) {
        const S2Plan plan = do_s2 ? make_stage2_plan(c->G) : S2Plan{};
        hipLaunchKernelGGL(dirichlet_kernel, dim3(c->S + 4 + do_fin), dim3(256), 0, c->stream, q, plan);
    } else {
        static thread_local DirBatch acc;
        acc.p[g_batch.k] = q;
        if (g_batch.k == g_batch.K - 1) {
            acc.plan = do_s2 ? make_stage2_plan(c->G) : S2Plan{};
            hipLaunchKernelGGL(dirichlet_kernel_b, dim3(c->S + 4 + do_fin, g_batch.K), dim3(256), 0, c->stream, acc);
        }
    }
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

// the same for n stored states at once (updateTau: gamma_store / eta_store are known up front):
// gamma [n][S][G], eta [n][4][4] -> rowprior [n][S + 4]
__global__ __launch_bounds__(256) void prior_batch_kernel(const double *__restrict__ gamma, const double *__restrict__ eta,
                                                          int n, int S, int G, double alpha, double delta,
                                                          double lgc_gamma, double lgc_eta, double *__restrict__ rowprior)
{
    const int rows = S + 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)n * rows; i += (size_t)gridDim.x * 256) {
        const int it = (int)(i / rows), r = (int)(i % rows);
        double lsum = 0.0;
        if (r < S) {
            const double *g = gamma + ((size_t)it * S + r) * G;
            for (int k = 0; k < G; ++k) lsum += (alpha - 1.0) * log(g[k]);
            rowprior[i] = lgc_gamma + lsum;
        } else {
            const double *e = eta + (size_t)it * 16 + (r - S) * 4;
            for (int b = 0; b < 4; ++b) lsum += (delta - 1.0) * log(e[b]);
            rowprior[i] = lgc_eta + lsum;
        }
    }
}

int k_prior_batch(dsm_ctx *c, const double *gamma, const double *eta, int n, double *prior_out)
{
    double lg, le;
    dirichlet_consts(c, &lg, &le);
    const size_t tot = (size_t)n * (c->S + 4);
    const int grid = (int)std::min<size_t>((tot + 255) / 256, 1024);
    hipLaunchKernelGGL(prior_batch_kernel, dim3(grid), dim3(256), 0, c->stream, gamma, eta, n, c->S, c->G, c->alpha, c->delta,
                       lg, le, prior_out);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

int k_prior(dsm_ctx *c, const double *gamma, const double *eta, double *prior_out)
{
    double lg, le;
    dirichlet_consts(c, &lg, &le);
    hipLaunchKernelGGL(prior_kernel, dim3(1), dim3(256), 0, c->stream, gamma, eta, c->S, c->G, c->alpha, c->delta, lg,
                       le, prior_out);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

// how many haplotypes are rare in every sample (max_s gamma_sg <= 0.01): the chain then runs the sweep's NT instantiation for this call
__global__ __launch_bounds__(256) void gamma_rare_kernel(const double *__restrict__ gamma, int S, int G, int *__restrict__ out)
{
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += 256) {
        double m = 0.0;
        for (int s = 0; s < S; ++s) m = fmax(m, gamma[(size_t)s * G + g]);
        if ((float)m <= (float)DSM_NT_RARE) atomicAdd(&cnt, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) *out = cnt;
}
// Which instantiation of the sweep a Gibbs call runs: -1 = by the abundances as last known (dsm_host.h: tau_rare_n -- no device work
// here), 0 / 1 = forced (dsm_ctx_set_tau_neartie).  Never for a batch or a sharded chain.
int k_tau_neartie_hint(dsm_ctx *c)
{
    c->tau_neartie_on = false;
    if (c->tau_neartie_mode == 0 || g_batch.K || c->shard_on || !c->tau_screen || c->G < 2) return DSM_OK;
    c->tau_neartie_on = c->tau_neartie_mode == 1 || (c->tau_rare_n > 0 && c->tau_rare_n < c->G);
    return DSM_OK;
}
int tau_rare_from_host(const double *gamma, int S, int G)
{
    int n = 0;
    for (int g = 0; g < G; ++g) {
        double m = 0.0;
        for (int s = 0; s < S; ++s) m = std::max(m, gamma[(size_t)s * G + g]);
        n += (float)m <= (float)DSM_NT_RARE;
    }
    return n;
}
// the same count from the resident gamma, read back into pinned memory: enqueued before a synchronisation the caller makes anyway
// (end of a Gibbs call), or with one of its own (wait: every 64 iterations inside a call)
int k_tau_rare_count(dsm_ctx *c, bool wait)
{
    if (!c->h_rare) HIP_TRY(hipHostMalloc((void **)&c->h_rare, sizeof(int), hipHostMallocDefault));
    hipLaunchKernelGGL(gamma_rare_kernel, dim3(1), dim3(256), 0, c->stream, c->gamma, c->S, c->G, c->nchange + 1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_rare, c->nchange + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemsetAsync(c->nchange + 1, 0, sizeof(int), c->stream));
    if (wait) { HIP_TRY(hipStreamSynchronize(c->stream)); c->tau_rare_n = *c->h_rare; }
    return DSM_OK;
}

template <int LPV, int NSL>
static void launch_tau(dsm_ctx *c, int mode, const TauParams &p, int grid, size_t sh)
{
    const int g2 = grid + p.do_fin;
    if (g_batch.K) {                                     // the Gibbs loop's sweep (mode 3) of K chains in one launch
        static thread_local BatchArgs<TauParams> acc;
        acc.p[g_batch.k] = p;
        if (g_batch.k == g_batch.K - 1)
            hipLaunchKernelGGL((tau_kernel_b<LPV, NSL, true, true>), dim3(g2, g_batch.K), dim3(256), sh, c->stream, acc);
        return;
    }
    if (mode == 3 && c->tau_neartie_on) hipLaunchKernelGGL((tau_kernel_nt<LPV, NSL>), dim3(g2), dim3(256), sh, c->stream, p);
    else if (mode == 3) hipLaunchKernelGGL((tau_kernel<LPV, NSL, true, true>), dim3(g2), dim3(256), sh, c->stream, p);
    else if (mode == 1) hipLaunchKernelGGL((tau_kernel<LPV, NSL, true, false>), dim3(g2), dim3(256), sh, c->stream, p);
    else hipLaunchKernelGGL((tau_kernel<LPV, NSL, false, true>), dim3(g2), dim3(256), sh, c->stream, p);
}

// lanes per variant and sample slots per lane of the sweep kernel for a table of S samples
static int tau_shape(int S, int *lpv, int *nsl)
{
    int LPV, NSL;
    if (S <= 16) { LPV = 16; NSL = 1; }
    else if (S <= 32) { LPV = 16; NSL = 2; }            // four variants per wavefront (77.8 vs 94.5 us as 32 x 1 at V=20k, S=32, G=8)
    else if (S <= 48) { LPV = 16; NSL = 3; }            // no padded slots: 48 useful lanes-samples instead of 64
    else if (S <= 64) { LPV = 32; NSL = 2; }            // two variants per wavefront: the per-step draw / reduction work (computed
                                                        // redundantly by every lane) is shared by both: 73.7 vs 78.5 us as 64 x 1
                                                        // (16 x 4 needs 235 VGPRs: 85 us)
    else if (S > 64 && S <= 96) { LPV = 32; NSL = 3; }  // 96 slots, two variants per wavefront (measured 10 % faster
                                                        // than 64 x 2 = 128 slots even though that tile re-uses a candidate)
    else {
        LPV = 64;
        const int need = (S + 63) / 64;
        NSL = need <= 4 ? need : (need <= 6 ? 6 : 8);
        if (need > 8) { dsm_set_error("S=%d exceeds DSM_MAX_S=%d", S, DSM_MAX_S); return DSM_ERR_UNSUPPORTED; }
    }
    *lpv = LPV; *nsl = NSL;
    return DSM_OK;
}
static size_t tau_lds_bytes(int G, int LPV, int NSL)
{
    return ((size_t)G * LPV * NSL + 16 + 16 + 6 + 2 * DSM_LOG_TAB_N) * sizeof(double) +
           ((size_t)G * LPV * NSL + 24 + 32) * sizeof(float);             // + fp32 copies of gamma / eta for the screening passes (+ the NT kernel's column maxima and per-haplotype maxima)
}

int k_tau_sweep(dsm_ctx *c, int mode, const double *gamma, const double *eta_sweep, const double *eta_ll,
                uint64_t *trace_slot, double *d_logp, uint32_t iter, int *nblocks, const uint32_t *u_raw, int slot,
                const TauFinalRider *rider)
{
    KTimer tm(c, DSM_K_TAU);
    const int S = c->S, G = c->G, V = c->V;
    int LPV, NSL;
    { const int rc = tau_shape(S, &LPV, &NSL); if (rc != DSM_OK) return rc; }
    const int gpb = 256 / LPV;
    int grid = (V + gpb - 1) / gpb;
    if (grid > DSM_MAX_GRID) grid = DSM_MAX_GRID;
    {   // experiment build: DESMAN_HIP_TAU_GRID = workgroups of the sweep launch (the kernel strides over the variants; fewer workgroups make several passes)
        static const int genv = DSM_AB_ENV("DESMAN_HIP_TAU_GRID") ? atoi(DSM_AB_ENV("DESMAN_HIP_TAU_GRID")) : 0;
        if (genv > 0 && genv < grid) grid = genv;
    }
    if (grid < 1) grid = 1;
    TauParams p;
    p.cnt_vs = c->cnt_vs; p.tau = c->tau; p.trace = trace_slot;
    p.gamma = gamma; p.eta_sweep = eta_sweep; p.eta_ll = eta_ll;
    p.u_raw = (c->tau_rng == DSM_RNG_MT19937 && (mode & 1)) ? u_raw : nullptr;
    p.logp = d_logp; p.ll_partial = c->ll_partial + (size_t)slot * DSM_MAX_GRID; p.nchange = c->nchange + slot; p.log_tab = c->log_tab;
    static const bool no_screen = DSM_AB_ENV("DESMAN_HIP_TAU_NO_SCREEN") != nullptr;
    p.screen = (no_screen || !c->tau_screen) ? 0 : 1;
    { static const bool nts = DSM_AB_ENV("DESMAN_HIP_NT_SKIP_TOTALS") != nullptr; p.nt_skip = nts ? 1 : 0; }
    p.step_cnt = c->step_cnt + (size_t)slot * 2 * DSM_MAX_GRID; p.screen_ctl = c->screen_ctl + slot;
    p.order = (c->blk_order && c->blk_order_n[slot] == grid) ? c->blk_order + (size_t)slot * DSM_MAX_GRID : nullptr;
    p.do_fin = 0;
    memset(&p.fin, 0, sizeof p.fin);
    if (rider) {
        p.do_fin = 1;
        p.fin = make_final(c, rider->nblocks, rider->it, 0, rider->prior, rider->gamma_src, rider->eta_src, slot ^ 1);
    }
    p.V = V; p.S = S; p.G = G; p.v_off = c->shard_on ? c->shard_voff : 0;
    p.k0 = (uint32_t)c->ctr_seed; p.k1 = (uint32_t)(c->ctr_seed >> 32); p.iter = iter;
    const size_t sh = tau_lds_bytes(G, LPV, NSL);
    if (sh > 160 * 1024) { dsm_set_error("gamma tile (%zu B) exceeds LDS", sh); return DSM_ERR_UNSUPPORTED; }
#define TAU_CASE(L, N) if (LPV == L && NSL == N) launch_tau<L, N>(c, mode, p, grid, sh)
    TAU_CASE(16, 1); TAU_CASE(16, 2); TAU_CASE(16, 3); TAU_CASE(32, 1); TAU_CASE(32, 2); TAU_CASE(32, 3); TAU_CASE(64, 1); TAU_CASE(64, 2); TAU_CASE(64, 3); TAU_CASE(64, 4);
    TAU_CASE(64, 6); TAU_CASE(64, 8);
#undef TAU_CASE
    HIP_TRY(hipGetLastError());
    if (nblocks) *nblocks = grid;
    return DSM_OK;
}

// workgroups one sweep launches and workgroups of that kernel the device holds at once (occupancy x compute units): the
// launch runs as launched / resident rounds, and the last, partly filled round is its tail (bench.py: roofline.tail_frac)
int tau_launch_info(dsm_ctx *c, int *launched, int *resident)
{
    int LPV, NSL;
    { const int rc = tau_shape(c->S, &LPV, &NSL); if (rc != DSM_OK) return rc; }
    const int gpb = 256 / LPV;
    int grid = (c->V + gpb - 1) / gpb;
    if (grid > DSM_MAX_GRID) grid = DSM_MAX_GRID;
    const size_t sh = tau_lds_bytes(c->G, LPV, NSL);
    int per_cu = 0, cus = 0;
#define TAU_CASE(L, N) if (LPV == L && NSL == N) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tau_kernel<L, N, true, true>, 256, sh))
    TAU_CASE(16, 1); TAU_CASE(16, 2); TAU_CASE(16, 3); TAU_CASE(32, 1); TAU_CASE(32, 2); TAU_CASE(32, 3); TAU_CASE(64, 1); TAU_CASE(64, 2); TAU_CASE(64, 3); TAU_CASE(64, 4);
    TAU_CASE(64, 6); TAU_CASE(64, 8);
#undef TAU_CASE
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    *launched = grid;
    *resident = per_cu * cus;
    return DSM_OK;
}

// ---- a chain sharded by positions: what a shard contributes to the per-iteration exchange besides its subset table.
// vec [18]: [0] log-likelihood of this shard's positions (its data constant + the launch's partials, reduced in the order
// finalize_body uses), [1] changed (v, g) pairs, [2..17] Esum [observed][true] -- counts below 2^53, exact as doubles.
__global__ __launch_bounds__(256) void shard_pack_kernel(const double *__restrict__ ll_partial, int nblocks, double ll_const,
                                                         int *__restrict__ nchange, unsigned long long *__restrict__ esum,
                                                         double *__restrict__ vec)
{
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double a = 0.0;
    for (int i = tid; i < nblocks; i += 256) a += ll_partial[i];
    red[tid] = a;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) { vec[0] = ll_const + red[0]; vec[1] = (double)*nchange; *nchange = 0; }
    if (tid < 16) {
        // Esum + the copies stage 1 added to (kernels_stats.hip: stats_agg_body); the copies are zero again when the exchanged total comes back
        unsigned long long tot = esum[tid];
        for (int k = 0; k < DSM_ESUM_PARTS; ++k) { const unsigned long long v = esum[16 + k * 16 + tid]; if (v) { tot += v; esum[16 + k * 16 + tid] = 0ull; } }
        vec[2 + tid] = (double)tot;
    }
}
__global__ void shard_unpack_kernel(const double *__restrict__ vec, unsigned long long *__restrict__ esum)
{
    if (threadIdx.x < 16) esum[threadIdx.x] = (unsigned long long)vec[2 + threadIdx.x];
}
int k_shard_pack(dsm_ctx *c, int nblocks)
{
    hipLaunchKernelGGL(shard_pack_kernel, dim3(1), dim3(256), 0, c->stream, c->ll_partial, nblocks, c->ll_const, c->nchange, c->esum, c->shard_vec);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}
int k_shard_unpack(dsm_ctx *c)
{
    hipLaunchKernelGGL(shard_unpack_kernel, dim3(1), dim3(64), 0, c->stream, c->shard_vec, c->esum);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

int k_finalize(dsm_ctx *c, int nblocks, int it, int star_mode, const double *prior, const double *gamma_src,
               const double *eta_src, int slot, int *rare_out)
{
    KTimer tm(c, DSM_K_FINAL);
    FinalParams p = make_final(c, nblocks, it, star_mode, prior, gamma_src, eta_src, slot);
    p.rare_out = rare_out;
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, c->stream, p);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

// kernels_stats.hip -- A2, the auxiliary-count sums (HaploSNP_Sampler.py:284-309 via :266,:276), "spec v2":
// the aggregated sampler restated in oracle/stats_agg.c (the law and its derivation are in that header).
//
//   stats_agg_kernel     stage 1: one wavefront per (variant, 64 samples); lane = sample, the 16 B count slab
//                        of (v,s) is one coalesced int4 load (the tau kernel's layout: no second copy of the
//                        tensor).  tau_v is wave-uniform, so the haplotype sets H_a(v) and the branches of the
//                        Gamma_a accumulation are scalar.  Per lane: Multinomial(x_b; eta[a,b] Gamma_a) for the
//                        four observed bases (dsm_binom.h: mult4) -> Esum (lane-private LDS columns) and the
//                        subset counts N[H_a(v)][s] (one coalesced row of global atomics per true base).
//                        Cost per cell ~ O(G + errors).  Inversion only: an item whose rarer outcome has a mean
//                        above 64 (DSM_LEAN_CAP), or more than DSM_XS reads off its heaviest base, is pushed onto a
//                        work list instead (lean kernel: <= 80 VGPRs, no rejection loop).  The lists: one per kind of
//                        item (rejection sampler / long search / search + two more binomials) x DSM_BIG_NL, each with
//                        its own counter -- workgroup b appends to list b mod DSM_BIG_NL of the kind.
//   stats_big_kernel     the listed items, one lane each (compacted, a wavefront holds items of one kind: no idle
//                        lanes, no path run for one lane only).  Items are independent work units (own Philox
//                        stream each): which kernel draws an item does not change the draw.
//   stats_stage2_kernel  stage 2: one workgroup per sample spreads N[.][s] over the haplotypes by recursive
//                        halving of the haplotype range (one large-count binomial per (node, subset): BTRS).
//   stats v1 (per-read draws, kernels_gibbs.hip: stats_kernel) remains for G > 16 / tables above 64 MB.
#include "dsm_binom.h"
#include "dsm_host.h"
#include "log_table.h"

#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

// ablation switches of DESMAN_HIP_STATS_DBG: only in the experiment build (dsm_host.h: DSM_AB_SWITCHES); the product kernel has none of the tests
#ifdef DSM_AB_SWITCHES
#define STATS_DBG(p, mask) ((p).dbg & (mask))
#else
#define STATS_DBG(p, mask) 0
#endif
#ifdef DSM_AB_SWITCHES           // wave time stamps of stage 1 (experiment build only; dsm_debug_s1_clocks reads them): [wave][entry, tables staged, after pass 1, 2, ..., before the epilogue, end]
#define S1_NCLK 16
__device__ unsigned long long s1_clk[8192 * S1_NCLK];
#define S1_CLK(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.y == 0) s1_clk[(size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * S1_NCLK + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int dsm_debug_s1_clocks(unsigned long long *out, int nwaves)
{
    if (nwaves > 8192) nwaves = 8192;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(s1_clk), (size_t)nwaves * S1_NCLK * sizeof(unsigned long long)) != hipSuccess) return -1;
    return S1_NCLK;
}
#else
#define S1_CLK(k) do { } while (0)
#endif
#define S1_CLKP(j) do { if (s1_pass < 3) S1_CLK(2 + 4 * s1_pass + (j)); } while (0)
struct StatsAggParams {
    const int32_t *cnt_vs;
    const uint64_t *tau;
    const double *gamma, *eta;
    int V, S, G;
    int v_off, V_tot;               // a chain sharded over GPUs by positions: this context holds positions v_off .. v_off + V of V_tot; the
                                    // counter-based streams are keyed by GLOBAL cell indices, so the draws do not depend on the sharding
    uint32_t k0, k1, iter;
    uint32_t *ntab;                 // [rep][2^G][ld]; workgroup b adds to copy b mod rep
    int rep, ld;                    // ld: row stride in words, >= S (ensure_ntab)
    unsigned long long *esum;       // [16]
    const double *log_tab;
    unsigned long long *big_list;   // deferred items: cell * 4 + observed base; DSM_BIG_NL sub-lists of big_seg entries each
    uint32_t *big_count;            // [DSM_BIG_NL] counters, 64 B apart (workgroup b appends to sub-list b % DSM_BIG_NL: one
                                    // counter took every deferring wavefront's atomic in turn, ~8 ns each -- 35k of them = the
                                    // whole pass on deep data)
    size_t big_seg;
    double lean_cap;                // stage 1 hands an item whose rarer outcome has a mean above this to the compacted kernel
    uint32_t swz;                   // ... + (s >> 4) * swz: the four 64 B lines of a subset's 64 samples sit in four different rows (stats_ntab_swz)
    uint32_t hmul;                  // the row of subset H is (H * hmul) mod 2^G (dsm_stage2.h: s2_row): an odd multiplier scatters the hot
                                    // subsets over the memory channels whatever the table's address
    int xcd;                        // 1: the table has a copy per XCD (rep = 8 k); a workgroup adds to a copy of ITS XCD (HW_REG_XCC_ID) with
                                    // workgroup-scope atomics, which execute in that XCD's L2 instead of at the memory side
    // spec 4 (pattern-aggregated stage 1): non-null -> only the REPRESENTATIVE positions (lowest position of each packed tau word:
    // pat_rep[word] & 0xFFFFFFFF) carry cells, with the counts of all positions of their word summed in pat_x [V][4][S]
    const unsigned long long *pat_rep;
    uint32_t *pat_x;
    const uint32_t *pat_list, *pat_n;   // the representatives of this pass (any order) and their number: stage 1 walks this list
    int dbg;                        // timing experiments only (DESMAN_HIP_STATS_DBG; compiled out of the product kernel: STATS_DBG above): bit 0 no draws, 1 no item seeding, 2 no E/N atomics, 3 no cell Philox
};

__device__ __forceinline__ uint64_t wave_uniform_u64(uint64_t x)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// LPV = lanes per variant: a wavefront works on 64 / LPV (variant, LPV-sample chunk) tasks at a time, so that tables of 16,
// 32, 48 or 96 samples fill its lanes (lane = sample alone leaves 3/4 of a wavefront idle at S = 16 and 1/4 at S = 96).
// SPEC: 2 or 3 (dsm_binom.h).  REGG: S <= LPV and G <= 8 -- a lane keeps its sample for the whole launch, so its G abundances
// live in registers (no LDS tile of gamma: the 4 KB it took at config 3 now hold the log / exp tables of spec 3 at the same six
// workgroups per CU) and the haplotype loop of a cell is scalar compares + one add per haplotype, no LDS read.
template <int LPV, int SPEC, bool REGG, bool PAT = false>
__device__ __forceinline__ void stats_agg_body(const StatsAggParams &p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_s[];
    S1_CLK(0);
#ifdef DSM_AB_SWITCHES
    if ((threadIdx.x & 63) == 0 && blockIdx.y == 0)          // where the wavefront runs: HW_REG_HW_ID (wave, SIMD, CU, SH, SE) and the XCC id
        s1_clk[(size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * S1_NCLK + 13] =
            ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 4)) << 8) | (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) | (1ull << 63);
#endif
    constexpr int NG = 64 / LPV;
    constexpr int NTAB = SPEC >= 3 ? (2 * DSM_LOG_TAB_N + DSM_EXP_TAB_N) : 0;      // doubles: log table, then exp table
    const int S = p.S, G = p.G, V = p.V;
    const int NCH = (S + LPV - 1) / LPV, SP = NCH * LPV;
    double *tabs = reinterpret_cast<double *>(smem_s);                // [NTAB]
    double *gT = tabs + NTAB;                                         // [G][SP] gamma transposed (not with REGG)
    double *rcp = gT + (REGG ? 0 : (size_t)G * SP);                   // [256]  1/k
    double *es = rcp + DSM_RCP_TAB_N;                                 // [16]   eta
    uint32_t *eacc = reinterpret_cast<uint32_t *>(es + 16);           // [16][256] lane-private Esum columns
    const double2 *ltab = reinterpret_cast<const double2 *>(tabs);
    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = lane / LPV, lig = lane % LPV;
    const int nwaves = gridDim.x * 4;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (tid >> 6));
    // spec 4: only the representatives carry cells -- the tasks are (entry of the pass's list, chunk); the list is in no
    // particular order (atomic appends), which changes no sum: a cell's stream is keyed by its position, the sums are integers
    const int nunit = PAT ? (int)__builtin_amdgcn_readfirstlane((int)*p.pat_n) : V;
    const int ntask = nunit * NCH;                                    // (variant, chunk of LPV samples)
    const int nslot = (ntask + NG - 1) / NG;                          // NG tasks per wavefront pass
    if constexpr (!REGG) {
        for (int i = tid; i < G * SP; i += 256) {
            const int g = i / SP, s = i - g * SP;
            gT[i] = (s < S) ? p.gamma[(size_t)s * G + g] : 0.0;
        }
    }
    if constexpr (SPEC >= 3) { for (int i = tid; i < NTAB; i += 256) tabs[i] = p.log_tab[i]; }
    for (int k = tid; k < DSM_RCP_TAB_N; k += blockDim.x) rcp[k] = k ? 1.0 / (double)k : 0.0;
    if (tid < 16) es[tid] = p.eta[tid];
#pragma unroll
    for (int i = 0; i < 16; ++i) eacc[i * 256 + tid] = 0u;
    double gr[REGG ? 8 : 1];
    if constexpr (REGG) {
#pragma unroll
        for (int g = 0; g < 8; ++g) gr[g] = (g < G && lig < S) ? p.gamma[(size_t)lig * G + g] : 0.0;
    }
    __syncthreads();

    S1_CLK(1);
    ISA_MARK("slot_setup");
    int s1_pass = 0;
    (void)s1_pass;
    // which copy of the subset table this workgroup adds to
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;                // HW_REG_XCC_ID[3:0] (gfx942 / gfx950)
    const unsigned copy = p.xcd ? xcc + 8u * ((blockIdx.x >> 3) % (unsigned)(p.rep >> 3)) : blockIdx.x % (unsigned)p.rep;
    for (int slot = wid; slot < nslot; slot += nwaves) {
        const int task = slot * NG + grp;
        const bool tv = task < ntask;
        const int u = tv ? task / NCH : 0, j = tv ? task - u * NCH : 0;
        const int v = PAT ? (int)p.pat_list[u] : u;
        const int s = j * LPV + lig;
        const bool active = tv && s < S;
        S1_CLKP(0);
        ISA_MARK("cell_load");
        uint64_t t;
        int4 c = make_int4(0, 0, 0, 0);
        if constexpr (PAT) {
            t = p.tau[v];
            if (active) {
                const uint32_t *xr = p.pat_x + (size_t)v * 4 * S + s;
                c.x = (int)xr[0]; c.y = (int)xr[(size_t)S]; c.z = (int)xr[2 * (size_t)S]; c.w = (int)xr[3 * (size_t)S];
            }
        } else {
            // (issuing a pass's loads one pass ahead -- the first before the tables are staged -- was built and measured in round 6: no
            // difference at any shape, 16 instructions and six registers more: profiles/r06_stats_ab.txt, builds `new` / `nopf`)
            t = p.tau[v];
            if (active) c = reinterpret_cast<const int4 *>(p.cnt_vs)[(size_t)v * S + s];
        }
        // haplotype sets of the four bases and the abundance each base carries in this sample; t is uniform over a lane
        // group, so the base of haplotype g and the branches below are scalar -- one group after the other when the
        // wavefront holds several
        ISA_MARK("cell_gamma");
        uint32_t H0 = 0, H1 = 0, H2 = 0, H3 = 0;
        double G0 = 0.0, G1 = 0.0, G2 = 0.0, G3 = 0.0;
        const double *gcol = gT + s;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const uint32_t tlo = __builtin_amdgcn_readlane((uint32_t)t, gi * LPV), thi = __builtin_amdgcn_readlane((uint32_t)(t >> 32), gi * LPV);
            const uint64_t tg = ((uint64_t)thi << 32) | tlo;
            if (NG == 1 || grp == gi) {
                if constexpr (REGG) {
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        if (g < G) {
                            const int a = (int)((tlo >> (2 * g)) & 3);
                            const uint32_t bit = 1u << g;
                            if (a == 0) { H0 |= bit; G0 = G0 + gr[g]; }
                            else if (a == 1) { H1 |= bit; G1 = G1 + gr[g]; }
                            else if (a == 2) { H2 |= bit; G2 = G2 + gr[g]; }
                            else { H3 |= bit; G3 = G3 + gr[g]; }
                        }
                    }
                } else {
                    for (int g = 0; g < G; ++g) {
                        const int a = (int)((tg >> (2 * g)) & 3);
                        const double x = gcol[g * SP];
                        const uint32_t bit = 1u << g;
                        if (a == 0) { H0 |= bit; G0 = G0 + x; }
                        else if (a == 1) { H1 |= bit; G1 = G1 + x; }
                        else if (a == 2) { H2 |= bit; G2 = G2 + x; }
                        else { H3 |= bit; G3 = G3 + x; }
                    }
                }
            }
        }
        const double Gam[4] = {G0, G1, G2, G3};
        ISA_MARK("cell_philox");
        const uint32_t cell = (uint32_t)s * (uint32_t)p.V_tot + (uint32_t)(p.v_off + v);
        uint32_t cbase[4];
        if (STATS_DBG(p, 8)) { cbase[0] = cell; cbase[1] = p.iter; cbase[2] = p.k0; cbase[3] = p.k1; }
        else philox4x32_10(cell, 0u, p.iter, DSM_STREAM_STA1, p.k0, p.k1, cbase);         // one Philox-10 per cell
        uint32_t nacc[4] = {0, 0, 0, 0};
        S1_CLKP(1);
        // Round 6: the four observed bases are straight-line code (the base is a compile-time constant: the count is a register, the
        // Esum column an immediate offset, no loop counter kept in a vector register, no copies at the loop's back edge), and the rare
        // outcome -- an item handed to the compacted kernel -- is collected in `hand` and tested ONCE per cell below.
        uint32_t hand = 0;                                                  // bit b: item b is handed over; bits 4 + 2 b: its kind
        const uint32_t xs[4] = {(uint32_t)c.x, (uint32_t)c.y, (uint32_t)c.z, (uint32_t)c.w};     // (aggregated counts may pass 2^31)
        auto item = [&](const int b, const uint32_t xb) __attribute__((always_inline)) {
            if (xb > 0) {
                ISA_MARK("item_w");
                double W[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) W[a] = es[a * 4 + b] * Gam[a];
                const double Wt = ((W[0] + W[1]) + W[2]) + W[3];
                if (!(Wt > 0.0)) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) W[a] = Gam[a];
                }
                ISA_MARK("item_seed");
                uint32_t n[4];
                const Xo128 rng = STATS_DBG(p, 2) ? Xo128{cbase[0] + b, cbase[1], cbase[2], cbase[3] | 1u} : item_seed<SPEC>(cbase, (uint32_t)b, p.k0, p.k1);
                bool ok = true;
                int kind = 0;
                if (STATS_DBG(p, 1)) { n[0] = (uint32_t)xb + (uint32_t)(W[0] > W[1]) + rng.s0; n[1] = n[2] = n[3] = 0; }
                else if constexpr (SPEC == 2) ok = s1_item<SPEC>(rng, xb, W, n, rcp, ltab, p.lean_cap, kind);
                else {
                    Xo128 r2 = rng;
                    bool defer = false;
                    mult4<false, SPEC>(r2, xb, W, n, rcp, ltab, defer, p.lean_cap, &kind);
                    ok = !defer;
                }
#ifdef DSM_S1_PAD                  // experiment: DSM_S1_PAD dependent 32-bit adds per item (is the pass VALU-bound?)
                {
                    uint32_t pad = n[0];
#pragma unroll
                    for (int i_ = 0; i_ < DSM_S1_PAD; ++i_) asm volatile("v_add_u32 %0, %0, %1" : "+v"(pad) : "v"(xb));
                    n[0] += (pad == 0x7FFFFFF1u) ? 1u : 0u;
                }
#endif
                ISA_MARK("item_esum");
                if (ok) {
                    // an aggregated count is consumed exactly once -- here, or by the compacted kernel when the item is handed over -- and
                    // whoever consumes it leaves zero behind for the next pass
                    if constexpr (PAT) p.pat_x[((size_t)v * 4 + b) * S + s] = 0u;
                    if (STATS_DBG(p, (4 | 16))) {
#pragma unroll
                        for (int a = 0; a < 4; ++a) nacc[a] += n[a];
                    } else {
                        uint32_t *erow = eacc + (b * 4) * 256 + tid;           // lane-private column: ds_add_u32, never a conflict
#pragma unroll
                        for (int a = 0; a < 4; ++a) { atomicAdd(erow + a * 256, n[a]); nacc[a] += n[a]; }
                    }
                } else hand |= (1u << b) | ((uint32_t)kind << (4 + 2 * b));
            }
        };
        if constexpr (SPEC == 2 && !REGG) { item(0, xs[0]); item(1, xs[1]); item(2, xs[2]); item(3, xs[3]); }
        else {
            // (the table-exp variant and the gamma-in-registers experiment keep the rolled loop: four copies of their item do not fit
            // the 80 registers of six wavefronts per SIMD)
#pragma unroll 1
            for (int b = 0; b < 4; ++b) item(b, (b == 0) ? xs[0] : (b == 1) ? xs[1] : (b == 2) ? xs[2] : xs[3]);
        }
        S1_CLKP(2);
        ISA_MARK("cell_handover");
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(hand != 0u) != 0ull, 0)) {
            // needs the rejection sampler / a long search: the compacted kernel re-does the item from its own stream.  One list per kind
            // of item (and DSM_BIG_NL of each): the wavefronts of the compacted kernel then hold items that take the same path; one
            // atomic per wavefront, base and kind, the lanes that hand over take consecutive slots
#pragma unroll 1
            for (int b = 0; b < 4; ++b) {
                const bool mine = (hand >> b) & 1u;
                const int kind = (int)((hand >> (4 + 2 * b)) & 3u);
#pragma unroll 1
                for (int kd = 0; kd < DSM_BIG_NT; ++kd) {
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(mine && kind == kd);
                    if (mask == 0ull) continue;
                    const uint32_t sub = (uint32_t)kd * DSM_BIG_NL + blockIdx.x % DSM_BIG_NL;
                    const int leader = __builtin_ctzll(mask);
                    uint32_t base = 0;
                    if (lane == leader) base = atomicAdd(p.big_count + sub * DSM_BIG_STRIDE, (uint32_t)__builtin_popcountll(mask));
                    base = __builtin_amdgcn_readlane(base, leader);
                    const uint32_t slot = base + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
                    if (mine && kind == kd) p.big_list[(size_t)sub * p.big_seg + slot] = (unsigned long long)cell * 4ull + (unsigned long long)b;
                }
            }
        }
        ISA_MARK("cell_table");
        // N[H_a(v)][s] += reads whose true base is a: adjacent lanes -> adjacent words of one table row
        const size_t ld = (size_t)p.ld;                                    // row stride of the subset table in words (>= S: ensure_ntab)
        uint32_t *const nt = p.ntab + (size_t)copy * (((size_t)1 << G) * ld);
        const uint32_t hmask = (1u << G) - 1u;
        // row of (subset, sample): (H hmul + (s >> 4) swz) mod 2^G -- the subset's part is scalar, the sample's a lane constant of the slot
        const uint32_t sw = ((uint32_t)s >> 4) * p.swz;
        const uint32_t r0 = (H0 * p.hmul + sw) & hmask, r1 = (H1 * p.hmul + sw) & hmask, r2 = (H2 * p.hmul + sw) & hmask, r3 = (H3 * p.hmul + sw) & hmask;
        if (STATS_DBG(p, (4 | 32))) { if ((nacc[0] ^ nacc[1] ^ nacc[2] ^ nacc[3] ^ r0 ^ r1 ^ r2 ^ r3) == 0x12345u) atomicAdd(nt + s, 1u); continue; }
        if (STATS_DBG(p, 64)) {        // plain stores instead of atomics (wrong sums; what the adds cost beyond a store)
            nt[(size_t)r0 * ld + s] = nacc[0]; nt[(size_t)r1 * ld + s] = nacc[1]; nt[(size_t)r2 * ld + s] = nacc[2]; nt[(size_t)r3 * ld + s] = nacc[3];
            continue;
        }
        if (p.xcd) {
            // this XCD's copy: every adder of the copy shares the L2 the atomic executes in (relaxed, workgroup scope: no sc1, the
            // line stays in L2); the kernel boundary writes the lines back for stage 2, which sums the copies
            if (nacc[0]) __hip_atomic_fetch_add(nt + (size_t)r0 * ld + s, nacc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (nacc[1]) __hip_atomic_fetch_add(nt + (size_t)r1 * ld + s, nacc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (nacc[2]) __hip_atomic_fetch_add(nt + (size_t)r2 * ld + s, nacc[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (nacc[3]) __hip_atomic_fetch_add(nt + (size_t)r3 * ld + s, nacc[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            if (nacc[0]) atomicAdd(nt + (size_t)r0 * ld + s, nacc[0]);
            if (nacc[1]) atomicAdd(nt + (size_t)r1 * ld + s, nacc[1]);
            if (nacc[2]) atomicAdd(nt + (size_t)r2 * ld + s, nacc[2]);
            if (nacc[3]) atomicAdd(nt + (size_t)r3 * ld + s, nacc[3]);
        }
        S1_CLKP(3);
        ++s1_pass;
    }
    S1_CLK(14);
    ISA_MARK("epilogue");
    // Esum: a wavefront's lane-private columns -> one transposing butterfly -> 16 adds to ONE OF DSM_ESUM_PARTS COPIES of Esum (round 6).
    // Rounds 2-5 summed the four wavefronts of a workgroup in LDS (three barriers) and sent 16 adds per workgroup to Esum itself: 1 488
    // workgroups adding to the same two cache lines take their turns at the memory side (~8 ns each), and the wavefronts that left
    // the pass loop last waited 4-5 us for the launch's final adds -- 10 % of the launch at config 3 (wave time stamps,
    // profiles/r06_stats_timeline.txt).  A copy takes 1/64 of the adds, no wavefront waits for another, and the consumer (the eta
    // rows of dirichlet_kernel; esum_fold_kernel on the API paths) sums the copies and leaves them zero.
    {
        uint32_t e[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i] = eacc[i * 256 + tid];
        const uint32_t tot = wave_transpose_reduce<16>(e);
        const int idx = transpose_index<16>(lane);
        if (lane < 16 && tot) atomicAdd(&p.esum[16 + ((unsigned)wid % DSM_ESUM_PARTS) * 16 + idx], (unsigned long long)tot);
    }
    S1_CLK(15);
}

// six wavefronts per SIMD (the persistent grid's six workgroups per CU): the register allocation must leave room for them
template <int LPV, int SPEC, bool REGG>
__global__ __launch_bounds__(256, 6) void stats_agg_kernel(StatsAggParams p) { stats_agg_body<LPV, SPEC, REGG>(p); }
template <int LPV>      // (five workgroups per CU: at six the 16- and 32-lane forms spill)
__global__ __launch_bounds__(256, 5) void stats_pat_kernel(StatsAggParams p) { stats_agg_body<LPV, 2, false, true>(p); }

// ---- spec 4: the two small passes ahead of stage 1 ------------------------------------------------------------------------
// Cells (v, s) and (v', s) of positions with the same packed tau word have the same weights W_a = eta[a,b] Gamma_a for every
// observed base b, and only sums over positions are consumed (Esum, N[H][s]): Multinomial(x; W) + Multinomial(x'; W) is
// Multinomial(x + x'; W), so the counts are summed per (word, sample, base) first and stage 1 runs once per word instead of once per
// position -- the same law (oracle twin: cbind.stats_agg(spec=4) sums the counts the same way and calls the spec-2 code), other draws.
// Which position stands for a word must not depend on the order the hardware runs things in (the cell's Philox key is built from
// it): the LOWEST position carrying the word, found by atomicMin.  Table entries are (~generation << 32 | position): a later pass
// always writes smaller values than anything an earlier one left, so the table is never cleared.
__global__ __launch_bounds__(256) void pat_rep_kernel(const uint64_t *__restrict__ tau, int V, int G, unsigned long long *__restrict__ rep, uint32_t hi)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    atomicMin(&rep[tau[v] & ((1ull << (2 * G)) - 1ull)], ((unsigned long long)hi << 32) | (unsigned long long)(uint32_t)v);
}
// up to 4096 words (G <= 6): thousands of positions per word would queue on the same few addresses (16 words in two cache lines at
// G = 2: ~100 us for V = 50k), so a 1024-thread workgroup takes the minimum of its share of the positions in LDS first and sends
// one atomicMin per word it met
__global__ __launch_bounds__(1024) void pat_rep_lds_kernel(const uint64_t *__restrict__ tau, int V, int G, unsigned long long *__restrict__ rep, uint32_t hi)
{
    __shared__ uint32_t lo[4096];
    const int words = 1 << (2 * G), tid = threadIdx.x;
    for (int i = tid; i < words; i += 1024) lo[i] = 0xFFFFFFFFu;
    __syncthreads();
    const uint64_t mask = (1ull << (2 * G)) - 1ull;
    for (int v = blockIdx.x * 1024 + tid; v < V; v += gridDim.x * 1024) atomicMin(&lo[tau[v] & mask], (uint32_t)v);
    __syncthreads();
    for (int i = tid; i < words; i += 1024)
        if (lo[i] != 0xFFFFFFFFu) atomicMin(&rep[i], ((unsigned long long)hi << 32) | (unsigned long long)lo[i]);
}
// Few words (G <= 3 with 64-sample chunks, G = 4 with 32-sample chunks): thousands of positions add to the same few blocks, and
// same-address global atomics resolve one after the other (V = 50k, S = 96: 248 us at G = 2, 153 at G = 3 with pat_agg_kernel).
// Here a 1024-thread workgroup pools its share of the positions in LDS first -- table [4^G][4][LPW], indexed by the word itself, so
// no look-up on the way in -- and adds the table to the representatives' blocks once at the end: workgroups x 4^G x 4 LPW global
// atomics instead of V x S x 4.  LPW = lanes per position (64: one position per wavefront pass; 32: two).
template <int LPW>
__global__ __launch_bounds__(1024) void pat_agg_lds_kernel(const int32_t *__restrict__ cnt_vs, const uint64_t *__restrict__ tau, int V, int S, int G,
                                                           const unsigned long long *__restrict__ rep, uint32_t *__restrict__ x, int nvsplit)
{
    extern __shared__ uint32_t ptab[];                       // [words][4][LPW]
    const int words = 1 << (2 * G);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < words * 4 * LPW; i += 1024) ptab[i] = 0u;
    __syncthreads();
    constexpr int PPW = 64 / LPW;                            // positions per wavefront pass
    const int chunk = (int)blockIdx.x / nvsplit, part = (int)blockIdx.x % nvsplit;      // sample chunk, share of the positions
    const int s = chunk * LPW + (lane % LPW);
    const uint64_t mask = (1ull << (2 * G)) - 1ull;
    for (int v0 = (part * 16 + wv) * PPW; v0 < V; v0 += nvsplit * 16 * PPW) {
        const int v = v0 + lane / LPW;
        if (v < V && s < S) {
            const int4 c = reinterpret_cast<const int4 *>(cnt_vs)[(size_t)v * S + s];
            uint32_t *row = ptab + (size_t)(tau[v] & mask) * 4 * LPW + (lane % LPW);
            if (c.x) atomicAdd(row, (uint32_t)c.x);
            if (c.y) atomicAdd(row + LPW, (uint32_t)c.y);
            if (c.z) atomicAdd(row + 2 * LPW, (uint32_t)c.z);
            if (c.w) atomicAdd(row + 3 * LPW, (uint32_t)c.w);
        }
    }
    __syncthreads();
    for (int i = tid; i < words * 4 * LPW; i += 1024) {
        const uint32_t n = ptab[i];
        if (!n) continue;
        const int w = i / (4 * LPW), b = (i / LPW) & 3, sl = chunk * LPW + i % LPW;
        atomicAdd(x + ((size_t)(uint32_t)rep[w] * 4 + b) * S + sl, n);      // (n != 0: some position carries the word, so it has a representative)
    }
}

// the representatives of the pass as a list (one thread per word; appended in whatever order the atomics resolve): stage 1 then has
// an even share of them per wavefront -- walking the positions instead, the lowest positions of the words (all within the first few
// thousand) made a few wavefronts do most of the work (V = 50k, S = 96, G = 6: 175 us).  The counter of the NEXT pass is cleared here.
__global__ __launch_bounds__(256) void pat_list_kernel(const unsigned long long *__restrict__ rep, int words, uint32_t hi, uint32_t *__restrict__ list,
                                                       uint32_t *__restrict__ n_this, uint32_t *__restrict__ n_next)
{
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w == 0) *n_next = 0u;
    if (w >= words) return;
    const unsigned long long e = rep[w];
    if ((uint32_t)(e >> 32) == hi) list[atomicAdd(n_this, 1u)] = (uint32_t)e;
}

// thread = (position, sample): the slab's four counts are added to the representative's [4][S] block -- lanes are consecutive samples,
// so an instruction's atomics fall on consecutive words.  Integer sums: the order of the adds changes nothing.
__global__ __launch_bounds__(256) void pat_agg_kernel(const int32_t *__restrict__ cnt_vs, const uint64_t *__restrict__ tau, int V, int S, int G,
                                                      const unsigned long long *__restrict__ rep, uint32_t *__restrict__ x)
{
    const size_t n = (size_t)V * S;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int v = (int)(i / (size_t)S), s = (int)(i - (size_t)v * S);
        const int4 c = reinterpret_cast<const int4 *>(cnt_vs)[i];
        const uint32_t r = (uint32_t)rep[tau[v] & ((1ull << (2 * G)) - 1ull)];
        uint32_t *xr = x + (size_t)r * 4 * S + s;
        if (c.x) atomicAdd(xr, (uint32_t)c.x);
        if (c.y) atomicAdd(xr + S, (uint32_t)c.y);
        if (c.z) atomicAdd(xr + 2 * (size_t)S, (uint32_t)c.z);
        if (c.w) atomicAdd(xr + 3 * (size_t)S, (uint32_t)c.w);
    }
}
template <int LPV, int SPEC, bool REGG>
__global__ __launch_bounds__(256, 6) void stats_agg_kernel_b(BatchArgs<StatsAggParams> b) { stats_agg_body<LPV, SPEC, REGG>(b.p[blockIdx.y]); }
template <int LPV>
__global__ __launch_bounds__(256, 5) void stats_pat_kernel_b(BatchArgs<StatsAggParams> b) { stats_agg_body<LPV, 2, false, true>(b.p[blockIdx.y]); }

// ---------------------------------------------------------------------------------------------------
// the deferred items (rarer outcome with a mean above 64: burn-in states, very deep data), one lane per item:
// the same arithmetic as stats_agg_kernel with the full sampler (BTRS).  tau_v differs from lane to lane here,
// so the haplotype sets / abundances are built with vector selects.
// ---------------------------------------------------------------------------------------------------
template <int SPEC>
__device__ __forceinline__ void stats_big_body(const StatsAggParams &p)
{
    __shared__ double2 ltab[DSM_LOG_TAB_N + DSM_EXP_TAB_N / 2];          // log table, then the exp table (dsm_binom.h: binom)
    __shared__ double rcp[DSM_RCP_TAB_N];
    __shared__ double es[16];
    __shared__ unsigned long long acc[16];
    __shared__ uint32_t eacc[16 * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t sub = blockIdx.x % (DSM_BIG_NT * DSM_BIG_NL), bi = blockIdx.x / (DSM_BIG_NT * DSM_BIG_NL), nb = gridDim.x / (DSM_BIG_NT * DSM_BIG_NL);
    const uint32_t nbig = p.big_count[sub * DSM_BIG_STRIDE];
    if (bi * 256u >= nbig) return;
    const unsigned long long *list = p.big_list + (size_t)sub * p.big_seg;
    ltab[tid] = reinterpret_cast<const double2 *>(p.log_tab)[tid];
    if (tid < DSM_EXP_TAB_N / 2) ltab[DSM_LOG_TAB_N + tid] = reinterpret_cast<const double2 *>(p.log_tab)[DSM_LOG_TAB_N + tid];
    for (int k = tid; k < DSM_RCP_TAB_N; k += blockDim.x) rcp[k] = k ? 1.0 / (double)k : 0.0;
    if (tid < 16) { es[tid] = p.eta[tid]; acc[tid] = 0ull; }
#pragma unroll
    for (int i = 0; i < 16; ++i) eacc[i * 256 + tid] = 0u;
    __syncthreads();
    const int S = p.S, G = p.G;
    for (uint32_t i = bi * 256u + tid; i < nbig; i += nb * 256u) {
        const unsigned long long item = list[i];
        const uint32_t cell = (uint32_t)(item >> 2);
        const int b = (int)(item & 3ull);
        const int s = (int)(cell / (uint32_t)p.V_tot), v = (int)(cell - (uint32_t)s * (uint32_t)p.V_tot) - p.v_off;
        const uint64_t t = p.tau[v];
        uint32_t xb;
        if (p.pat_x) { uint32_t *xw = p.pat_x + ((size_t)v * 4 + b) * S + s; xb = *xw; *xw = 0u; }     // spec 4: consumed here (see stats_agg_body)
        else xb = (uint32_t)p.cnt_vs[((size_t)v * S + s) * 4 + b];
        uint32_t H[4] = {0, 0, 0, 0};
        double Gam[4] = {0.0, 0.0, 0.0, 0.0};
        for (int g = 0; g < G; ++g) {
            const int a = (int)((t >> (2 * g)) & 3);
            const double x = p.gamma[(size_t)s * G + g];
#pragma unroll
            for (int k = 0; k < 4; ++k) if (a == k) { H[k] |= 1u << g; Gam[k] = Gam[k] + x; }
        }
        double W[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) W[a] = es[a * 4 + b] * Gam[a];
        const double Wt = ((W[0] + W[1]) + W[2]) + W[3];
        if (!(Wt > 0.0)) {
#pragma unroll
            for (int a = 0; a < 4; ++a) W[a] = Gam[a];
        }
        uint32_t n[4], cbase[4];
        philox4x32_10(cell, 0u, p.iter, DSM_STREAM_STA1, p.k0, p.k1, cbase);
        Xo128 rng = item_seed<SPEC>(cbase, (uint32_t)b, p.k0, p.k1);
        bool defer = false;
        mult4<true, SPEC>(rng, xb, W, n, rcp, ltab, defer);
        uint32_t *erow = eacc + (b * 4) * 256 + tid;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            erow[a * 256] += n[a];
            if (n[a]) atomicAdd(p.ntab + (size_t)(blockIdx.x % (unsigned)p.rep) * (((size_t)1 << G) * (size_t)p.ld) + (size_t)((H[a] * p.hmul + ((uint32_t)s >> 4) * p.swz) & ((1u << G) - 1u)) * (size_t)p.ld + s, n[a]);
        }
    }
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

// the copies of Esum stage 1 adds to (stats_agg_body's last lines) -> Esum, copies zero again.  The Gibbs loop never launches this: the
// eta rows of dirichlet_kernel do the same sum on their way in.  For the paths that read Esum otherwise (dsm_ctx_sample_stats,
// dsm_ctx_debug_stage1, the exchange of a chain sharded by positions, the table-place probe).
__global__ __launch_bounds__(256) void esum_fold_kernel(unsigned long long *__restrict__ esum)
{
    static_assert(DSM_ESUM_PARTS * 4 == 256, "one thread per (copy, group of four counters)");
    const int k = threadIdx.x >> 2, q = threadIdx.x & 3;
    unsigned long long *pp = esum + 16 + k * 16 + q * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned long long v = pp[i];
        if (v) { pp[i] = 0ull; atomicAdd(&esum[q * 4 + i], v); }
    }
}
int k_esum_fold(dsm_ctx *c)
{
    hipLaunchKernelGGL(esum_fold_kernel, dim3(1), dim3(256), 0, c->stream, c->esum);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

template <int SPEC>
__global__ __launch_bounds__(256) void stats_big_kernel(StatsAggParams p) { stats_big_body<SPEC>(p); }
template <int SPEC>
__global__ __launch_bounds__(256) void stats_big_kernel_b(BatchArgs<StatsAggParams> b) { stats_big_body<SPEC>(b.p[blockIdx.y]); }

#include "dsm_stage2.h"

struct Stage2Batch { Stage2Params p[DSM_MAX_BATCH]; S2Plan plan; };
template <int SPEC>
__global__ __launch_bounds__(1024) void stats_stage2_kernel(Stage2Params p, S2Plan plan)
{
    __shared__ __attribute__((aligned(16))) char smem2[S2_SMEM_BYTES];
    stage2_sample<SPEC>(p, plan, (int)blockIdx.x / p.nsplit, smem2, true, (int)blockIdx.x % p.nsplit);
}
template <int SPEC>
__global__ __launch_bounds__(1024) void stats_stage2_kernel_b(Stage2Batch b)
{
    __shared__ __attribute__((aligned(16))) char smem2[S2_SMEM_BYTES];
    stage2_sample<SPEC>(b.p[blockIdx.y], b.plan, blockIdx.x, smem2, true);
}

// test hook: variate i of a sampler from the stream Philox({i, 0, 0, 'TEST'})  (oracle: orc_binom_test / orc_mult4_test)
template <int SPEC>
__global__ __launch_bounds__(256) void binom_test_kernel(int kind, uint32_t n, double wa, double wb, double w2, double w3,
                                                         uint32_t k0, uint32_t k1, int nsamp, const double *log_tab,
                                                         uint32_t *out)
{
    __shared__ double2 ltab[DSM_LOG_TAB_N + DSM_EXP_TAB_N / 2];
    __shared__ double rcp[DSM_RCP_TAB_N];
    const int tid = threadIdx.x;
    ltab[tid] = reinterpret_cast<const double2 *>(log_tab)[tid];
    if (tid < DSM_EXP_TAB_N / 2) ltab[DSM_LOG_TAB_N + tid] = reinterpret_cast<const double2 *>(log_tab)[DSM_LOG_TAB_N + tid];
    for (int k = tid; k < DSM_RCP_TAB_N; k += blockDim.x) rcp[k] = k ? 1.0 / (double)k : 0.0;
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    if (i >= nsamp) return;
    Xo128 rng = xo_seed((uint32_t)i, 0u, 0u, DSM_STREAM_TEST, k0, k1);
    bool dummy = false;
    if (kind != 2) out[i] = binom<true, SPEC>(rng, n, wa, wb, rcp, ltab, dummy, kind == 0 ? DSM_BINV_MEAN_CAP : DSM_BINV_MEAN_CAP_S2);
    else {
        const double W[4] = {wa, wb, w2, w3};
        uint32_t m[4];
        mult4<true, SPEC>(rng, n, W, m, rcp, ltab, dummy);
        out[i * 4 + 0] = m[0]; out[i * 4 + 1] = m[1]; out[i * 4 + 2] = m[2]; out[i * 4 + 3] = m[3];
    }
}

// =====================================================================
// host side
// =====================================================================
// spec v2 applies when the subset table fits (G <= 16, 2^G * S * 4 B <= 64 MB, every sample's depth < 2^32).  Where both
// apply the cheaper one runs, by a cost model of the two passes fitted on MI355X (us per iteration, 933 x 64 ... 50k x 96):
//   per-read pass (v1)   25 + (0.25 + 0.02 G + 0.0029 G^2) per million reads   -- O(depth x G): 0.30 at G = 2, 0.60 at 8, 1.31 at 16
//   aggregated pass (v2) 24 + 0.062 per thousand cells + 14 + 0.03 x 3V / 2^G (one table; ~0 with copies)   -- O(cells); cells = V x S rounded up to
//                        the kernel's lane groups; + 14 for stage 2 in the Dirichlet launch (G >= 10: its own launch over
//                        2^G subsets per sample, + 0.012 x 2^G); the last term is the
//                        same-address contention of the subset-table atomics (3 per cell onto 2^G x S counters: 9 375 per
//                        counter at V = 50k, G = 4 -> 410 us; 117 at config 3), spread over rep copies of the table
//                        (stats_ntab_rep below)
// so shallow data (< ~100 reads per cell) and problems below ~200k cells keep the per-read pass.
// The rule is a function of the shape and the read totals only: the same on every run and every GPU.
// lanes per variant of stats_agg_kernel: the chunk size that pads S least (ties: the larger one)
static int stats_agg_lpv(int S)
{
    int best = 64;
    for (int l : {32, 16}) if ((S + l - 1) / l * l < (S + best - 1) / best * best) best = l;
    return best;
}

int stats_spec(const dsm_ctx *c)
{
    int force = c->force_stats_spec;
    if (force == 0) {                       // DESMAN_HIP_STATS_SPEC=1|2: the choice for every context that has none of its own
        const char *e = getenv("DESMAN_HIP_STATS_SPEC");   // (2 = the draws of a batched run, also for chains run one by one)
        if (e && (e[0] == '1' || e[0] == '2' || e[0] == '3' || e[0] == '4') && e[1] == 0) force = e[0] - '0';
    }
    if (force == 1) return 1;
    if (c->G < 1 || c->G > 16) return 1;
    if (((size_t)1 << c->G) * (size_t)c->S * 4 > ((size_t)64 << 20)) return 1;
    if (c->max_depth >= ((uint64_t)1 << 32)) return 1;
    // spec 4 = spec 2's samplers over tau PATTERNS (positions sharing their packed word share a cell per sample; pat_rep_kernel above).
    // It needs the direct-address table of 4^G words (G <= 8) and a chain that is not sharded by positions (representatives would be
    // per shard: a sharded chain is the unsharded chain UNDER SPEC 2, desman_amd/vshard.py); asked for where it does not apply, spec 2
    // runs.  A batch runs what its chains would run alone (round 5: the pooling passes chain by chain, stage 1 over the words as one launch).
    const bool pat_ok = c->G <= 8 && !c->shard_on;
    if (force == 4) return pat_ok ? 4 : 2;
    if (force >= 2) return force;                        // 2 = the default version of the aggregated draws, 3 = the table exp/log variant
    double reads = 0.0;
    for (int64_t d : c->depth) reads += (double)d;
    const int lpv = stats_agg_lpv(c->S);
    const double cells = (double)c->V * (double)((c->S + lpv - 1) / lpv * lpv);
    const double g = (double)c->G;
    const double t1 = 25.0 + (0.25 + 0.02 * g + 0.0029 * g * g) * 1e-6 * reads;    // per read: its uniform + G - 1 threshold tests (0.30 / 0.60 / 1.31 at G = 2 / 8 / 16)
    const double stage2 = 14.0 + (c->G >= 10 ? 0.012 * (double)(1u << c->G) : 0.0);      // its own launch from G = 10: 62 us at G = 12
    const int rep = stats_ntab_rep(c);
    const double per = 3.0 * (double)c->V / (double)(1u << c->G);                // atomics per counter of the subset table
    const double t2 = 24.0 + stage2 + 0.062e-3 * cells + (rep == 1 ? 0.03 * per : 0.003 * per / rep);   // with copies: no measurable penalty
    // over tau words (spec 4): one more pass over the counts (~60 us per 77 MB incl. its atomics) + two small launches, then stage 1 over
    // the words the chain's tau actually holds instead of its V positions.  What it saves is stage 1's per-cell work, not its
    // per-error-read work (the pooled cells hold the same reads), so it pays on large tables with few words -- and most where stage 1 is
    // most expensive, in the chains of a G-sweep that have fewer haplotypes than the table has strains.  Measured, ms per iteration,
    // spec 4 vs 2 (profiles/r04_misfit_scan_spec4.txt): six-strain 50k x 96 table G = 2 0.14 vs 0.91, 3 0.21 vs 0.84, 4 0.27 vs 0.42,
    // 5 0.33 vs 0.44, 6 0.34 vs 0.47, 7 0.41 vs 0.49, 8 0.42 vs 0.51 (at 7 and 8 the six real strains keep the words few although
    // 4^G > V); four-strain 20k x 64: 0.10 vs 0.22, 0.15 vs 0.18, then 0.16 vs 0.13, 0.17 vs 0.14; 10k x 64: 0.10 vs 0.12, 0.13 vs 0.12,
    // 0.14 vs 0.09.  A table whose words are all different pays the pooling pass for nothing (+12 % at 50k x 96).  The rule -- shape
    // only, like everything here: G <= 2 from half a million cells, G = 3 from a million, G = 4 ... 8 from 2.5 million cells where
    // even 64 x 2^G words (several times what biallelic positions can form: 12 (2^G - 2) + 4) stay below V.  Anything else can ask for
    // it (dsm_ctx_force_stats_spec(4), DESMAN_HIP_STATS_SPEC=4).
    if (pat_ok && force == 0 && t2 < t1) {
        const bool small_g = (c->G <= 2 && cells >= 0.5e6) || (c->G == 3 && cells >= 1.0e6);
        const bool mid_g = c->G >= 4 && cells >= 2.5e6 && 64.0 * (double)((size_t)1 << c->G) <= (double)c->V;
        if (small_g || mid_g) return 4;
    }
    return t2 < t1 ? DSM_STATS_AGG : 1;
}

// Few subsets and many positions put thousands of atomics on every counter of the subset table (3 V / 2^G each: 9 375 at
// V = 50k, G = 4 -- 410 us of same-address serialisation).  The table is then kept in `rep` copies, workgroup b adds to copy
// b mod rep and stage 2 reads their sum: integers, so nothing changes but the time.  What decides is the rate of atomics per
// 64 B line of the table(s): stage 1 issues 3 per cell and takes ~0.06 us per thousand cells, so the rate is ~48 000 /
// (lines x rep) per us whatever V is, and the measurements (scripts/shape_scan.py) put the knee at
// lines x rep ~ 300: V = 100k, S = 64, G = 2 (16 lines) 585 us with 16 copies, 304 with 32; V = 50k, S = 96, G = 5 (192 lines)
// 472 us with one table, 300 with two; V = 20k, S = 32, G = 5 (64 lines) 0.127 ms per iteration with two copies, 0.094 with
// sixteen.  So: one table while a counter takes <= 128 atomics (config 3: 117, config 5: 37); otherwise at least two copies
// (V = 20k, S = 64, G = 8: 145 -> 88 us) and as many as bring lines x rep to 384 -- every copy is one more read per subset at
// stage 2's root (64 copies: +20 us on the Dirichlet launch).  DESMAN_HIP_NTAB_LINES overrides the 384.
int stats_ntab_rep(const dsm_ctx *c)
{
    static const double want = DSM_AB_ENV("DESMAN_HIP_NTAB_LINES") ? atof(DSM_AB_ENV("DESMAN_HIP_NTAB_LINES")) : 384.0;
    static const int force = DSM_AB_ENV("DESMAN_HIP_NTAB_REP") ? atoi(DSM_AB_ENV("DESMAN_HIP_NTAB_REP")) : 0;      // A/B switch
    if (force > 0) return force;
    // a chain sharded by positions: every rank must lay its table out alike (the tables are all-reduced element by element), so
    // the rule reads the WHOLE table's position count, never the shard's own (shards differ by one position: V = 21 845, G = 8 on
    // two ranks gave 10 922 -> one copy and 10 923 -> two).  More copies than a shard needs cost one read each at stage 2's root.
    const double per = 3.0 * (double)(c->shard_on ? c->shard_vtot : c->V) / (double)((size_t)1 << c->G);
    if (per <= 128.0) return 1;
    const double lines = (double)((((size_t)1 << c->G) * (size_t)c->S * 4 + 63) / 64);
    int rep = 2;
    while (lines * rep < want && rep < 64 && (size_t)(2 * rep) * ((size_t)1 << c->G) * (size_t)c->S * 4 <= ((size_t)64 << 20)) rep *= 2;
    return rep;
}

// a copy of the table per XCD (stats_agg_body: the atomics of stage 1 then execute in the XCD's L2) while 8 copies stay below 64 MB
static bool stats_ntab_xcd(const dsm_ctx *c)
{
    // measured at config 3 (rocprofv3 kernel trace of the Gibbs loop): stage 1 47.9 -> 46.1 us, but the root of stage 2 then reads
    // eight copies: Dirichlet launch 19.2 -> 22.7 us.  Off by default.
    static const int env = DSM_AB_ENV("DESMAN_HIP_NTAB_XCD") ? atoi(DSM_AB_ENV("DESMAN_HIP_NTAB_XCD")) : 0;      // A/B switch
    return env != 0 && (size_t)8 * ((size_t)1 << c->G) * (size_t)c->S * 4 <= ((size_t)64 << 20);
}

// Odd multiplier of the subset -> table row map.  Why: the subsets that take most of stage 1's atomics are the low-popcount ones, and
// with the identity map their rows sit at addresses base + 256 H whose bits line up with the memory-channel hash: depending on where
// hipMalloc put the 64 KB table, the memory-side atomics of stage 1 cost 2 us or 26 us at config 3 (stage 1 46 vs 72 us; found when
// two more allocations ahead of it -- the persistent NMFT kernel's -- moved the table; base offsets scanned: fast at 256 B and
// 32-60 KB into a 2 MB block, slow at 0, 4 KB, 64 KB ...).  Scattering the rows takes the pathology away (46-54 us at every offset
// scanned).  DESMAN_HIP_NTAB_HMUL=1 is the identity (A/B switch).
// Round 6: the sample's part of the row map.  A subset's 64 samples are one 256 B row = four 64 B lines, and a wavefront's add to it is one
// instruction = four line-atomics that arrive at the memory side together.  With (s >> 4) * swz added to the row index the four lines of a
// LOGICAL row sit in four different physical rows, i.e. 256 B blocks: the adds a hot subset takes are spread over four places of the table
// instead of queueing at one.  Measured at config 3 (scripts/dbg/r06_swz_scan.py, profiles/r06_swz_scan.txt): stage 1 40.5 us at EVERY one
// of eight table places and every odd swz tried (1, 3, 17, 85; also 64), against 45-47 us at the good places and 55-59 us at the bad ones
// without it -- the place lottery of rounds 2-5 (44 vs 54-61 us for the same launch, hipMalloc's answer deciding; stats_place_ntab's eight
// timed passes per table; a probe that kept a slow place in 2 of 6 processes) is gone, and the launch is 11 % faster than its best place
// was.  Stage 2 reads the same map (dsm_stage2.h); integer sums: nothing but the time changes.  0 = off (the map of rounds 2-5: the
// place is then measured as before).  DESMAN_HIP_NTAB_SWZ in the experiment build.
uint32_t stats_ntab_swz()
{
    static const uint32_t k = DSM_AB_ENV("DESMAN_HIP_NTAB_SWZ") ? (uint32_t)strtoul(DSM_AB_ENV("DESMAN_HIP_NTAB_SWZ"), nullptr, 0) : DSM_NTAB_SWZ;
    return k;
}
uint32_t stats_ntab_hmul()
{
    static const uint32_t k = DSM_AB_ENV("DESMAN_HIP_NTAB_HMUL") ? ((uint32_t)strtoul(DSM_AB_ENV("DESMAN_HIP_NTAB_HMUL"), nullptr, 0) | 1u) : 0x9E3779B1u;
    return k;
}

// Row stride of the subset table.  With S a multiple of 64 the rows are whole 256 B blocks and where the hot rows fall relative to the
// memory-channel interleave decides what stage 1's memory-side atomics cost: at config 3 (S = 64) 44 us or 54 us (69 at the worst) for the
// same launch, switching with every 256 B the table's base moves and with the multiplier of the row map (scripts/dbg/ntab_off_scan.py);
// S = 48 or 96 (rows of 192 / 384 B, which cut the interleave at a different place every row) show none of it.  Longer rows (320 / 384 B)
// narrowed the spread without closing it, so DSM_NTAB_PAD is 0 and the table's START is measured instead (stats_place_ntab below);
// DESMAN_HIP_NTAB_PAD adds words to such rows in the experiment build (A/B switch).
int stats_ntab_ld(int S)
{
    static const int pad = DSM_AB_ENV("DESMAN_HIP_NTAB_PAD") ? atoi(DSM_AB_ENV("DESMAN_HIP_NTAB_PAD")) : DSM_NTAB_PAD;
    return (S % 64 == 0) ? S + pad : S;
}

// Where the table starts.  What the memory-side atomics of stage 1 cost depends on how the rows that take most adds fall on the
// memory channels, i.e. on the table's PHYSICAL address: at config 3 the same launch takes 44 us at one 256 B offset and 54-61 us at
// the next (scripts/dbg/ntab_scan_inproc.py: stable to 0.3 us for a given place, the median place costs 53 us) -- and hipMalloc's
// answer moves with every allocation made before it and with the box (round 2's 0.105 ms was a lucky place, the same build measured
// 0.131 once two allocations preceded the table).  Neither an odd row multiplier, nor longer rows, nor two or four copies take the
// spread away, so the place is MEASURED: the table is allocated with DSM_NTAB_PLACES x 256 B to spare and
// stats_place_ntab() times stage 1 on the chain's own state at each place, once per table, and keeps the fastest.
// DESMAN_HIP_NTAB_OFF=<bytes> fixes the place instead (experiments); DESMAN_HIP_NTAB_TUNE=0 keeps place 0.
#define DSM_NTAB_PLACES 8
// Round 5 (VERDICT r4 "weak" 12: 55 chains of a sweep each probed again, next to whatever else ran on the GPU): a table whose place has been
// measured outlives its chain.  dsm_ctx_destroy hands it to this pool (stats_release_ntab) and the next chain of the same device and
// table size takes it, place and all, without a probe -- a table's cost is a property of its PHYSICAL address, which it keeps.  Probes
// themselves are serialised (g_ntab_probe_mu): never two of this process at a time.  The pool holds at most 32 tables (each <= 512 KB:
// larger ones are never probed and never pooled); dsm_debug_ntab_probes() counts the probes of the process.
struct NtabSlot { int device; size_t need; int ld, rep, xcd; uint32_t *raw, *base; size_t off; };
static std::mutex g_ntab_mu, g_ntab_probe_mu;
static std::vector<NtabSlot> g_ntab_pool;
static std::atomic<int> g_ntab_probes{0};
extern "C" int dsm_debug_ntab_probes(void) { return g_ntab_probes.load(); }
// frees the pooled subset tables (dsm_release_device_caches)
void stats_ntab_pool_release()
{
    std::lock_guard<std::mutex> lk(g_ntab_mu);
    for (const NtabSlot &sl : g_ntab_pool) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(sl.device);
        (void)hipFree(sl.raw);
        (void)hipSetDevice(cur);
    }
    g_ntab_pool.clear();
}
void stats_release_ntab(dsm_ctx *c)
{
    if (!c->ntab_raw) return;
    bool kept = false;
    if (c->ntab_measured) {
        // (whoever takes the table next clears it on ITS stream: nothing of this context's may still be queued on the table)
        (void)hipStreamSynchronize(c->stream);
        std::lock_guard<std::mutex> lk(g_ntab_mu);
        if (g_ntab_pool.size() < 32) { g_ntab_pool.push_back(NtabSlot{c->device, c->ntab_len, c->ntab_ld, c->ntab_rep, c->ntab_xcd, c->ntab_raw, c->ntab_base, c->ntab_off}); kept = true; }
    }
    if (!kept) (void)hipFree(c->ntab_raw);
    c->ntab_raw = nullptr; c->ntab = nullptr; c->ntab_base = nullptr; c->ntab_len = 0; c->ntab_placed = c->ntab_measured = false;
}
static int ensure_ntab(dsm_ctx *c)
{
    // (the layout the pass wants now; the context's fields keep the layout its CURRENT table was made for until that table is released)
    int rep = stats_ntab_rep(c);
    const int xcd = stats_ntab_xcd(c) ? 1 : 0;
    if (xcd) rep = std::max(8, (rep + 7) / 8 * 8);
    const int ld = stats_ntab_ld(c->S);
    const size_t need = (size_t)rep * ((size_t)1 << c->G) * (size_t)ld;
    if (c->ntab_raw && (c->ntab_len != need || c->ntab_ld != ld || c->ntab_rep != rep || c->ntab_xcd != xcd)) stats_release_ntab(c);
    c->ntab_rep = rep; c->ntab_ld = ld; c->ntab_xcd = xcd;
    static const bool scan = DSM_AB_ENV("DESMAN_HIP_NTAB_SCAN") != nullptr;          // experiments: the offset is re-read at every call
    const char *eo = DSM_AB_ENV("DESMAN_HIP_NTAB_OFF");
    const size_t off_env = eo ? ((size_t)strtoull(eo, nullptr, 0) & ~(size_t)255) : (size_t)-1;
    if (c->ntab && c->ntab_len == need && (!scan || off_env == (size_t)-1 || off_env == c->ntab_off)) return DSM_OK;
    if (c->ntab_raw) stats_release_ntab(c);
    if (off_env == (size_t)-1) {                           // a table of this size whose place was measured by an earlier chain
        std::lock_guard<std::mutex> lk(g_ntab_mu);
        for (size_t i = 0; i < g_ntab_pool.size(); ++i)
            // (the measured place belongs to the access pattern -- row stride, copies, per-XCD mode -- not to the word count alone)
            if (g_ntab_pool[i].device == c->device && g_ntab_pool[i].need == need && g_ntab_pool[i].ld == c->ntab_ld &&
                g_ntab_pool[i].rep == c->ntab_rep && g_ntab_pool[i].xcd == c->ntab_xcd) {
                const NtabSlot sl = g_ntab_pool[i];
                g_ntab_pool.erase(g_ntab_pool.begin() + (long)i);
                c->ntab_raw = sl.raw; c->ntab_base = sl.base; c->ntab_off = sl.off; c->ntab = sl.base + sl.off / 4; c->ntab_len = need;
                c->ntab_placed = c->ntab_measured = true;
                break;
            }
    }
    if (c->ntab_raw) {
        HIP_TRY(hipMemsetAsync(c->ntab, 0, need * sizeof(uint32_t), c->stream));
        return DSM_OK;
    }
    const size_t off = off_env != (size_t)-1 ? off_env : 0;
    const size_t spare = std::max<size_t>(off, (size_t)DSM_NTAB_PLACES * 256) + 4096;
    hipError_t e = hipMalloc((void **)&c->ntab_raw, need * sizeof(uint32_t) + spare);
    if (e != hipSuccess) { dsm_set_error("hipMalloc(%zu B) failed: %s", need * 4 + spare, hipGetErrorString(e)); return DSM_ERR_NOMEM; }
    c->ntab_base = reinterpret_cast<uint32_t *>(((uintptr_t)c->ntab_raw + 4095) & ~(uintptr_t)4095);
    c->ntab = c->ntab_base + off / 4;
    c->ntab_off = off;
    c->ntab_len = need;
    c->ntab_placed = off_env != (size_t)-1;               // a place given from outside is not measured again
    c->ntab_measured = false;
    HIP_TRY(hipMemsetAsync(c->ntab, 0, need * sizeof(uint32_t), c->stream));
    return DSM_OK;
}

int stats_place_ntab(dsm_ctx *c)
{
    static const bool on = !(getenv("DESMAN_HIP_NTAB_TUNE") && atoi(getenv("DESMAN_HIP_NTAB_TUNE")) == 0);
    static const bool verbose = getenv("DESMAN_HIP_NTAB_TUNE") && atoi(getenv("DESMAN_HIP_NTAB_TUNE")) == 2;
    if (stats_spec(c) < 2) return DSM_OK;
    int r = ensure_ntab(c);
    if (r != DSM_OK) return r;
    if (c->ntab_placed) return DSM_OK;
    c->ntab_placed = true;
    if (stats_ntab_swz() != 0u) return DSM_OK;          // round 6: with the sample-swizzled row map every place costs the same (stats_ntab_swz): nothing to measure
    // tables of a few hundred KB: larger ones spread over the channels whatever their place (config 5, 1.5 MB: 302 us everywhere)
    if (!on || g_batch.K || c->ntab_len * sizeof(uint32_t) > ((size_t)512 << 10)) return DSM_OK;
    std::lock_guard<std::mutex> probe_lk(g_ntab_probe_mu);
    g_ntab_probes.fetch_add(1);
    const bool timing = c->timing;
    c->timing = false;
    hipEvent_t ev[2];
    HIP_TRY(hipEventCreate(&ev[0]));
    HIP_TRY(hipEventCreate(&ev[1]));
    auto clear = [&]() -> int {
        HIP_TRY(hipMemsetAsync(c->ntab, 0, c->ntab_len * sizeof(uint32_t), c->stream));
        HIP_TRY(hipMemsetAsync(c->esum, 0, (16 + 16 * DSM_ESUM_PARTS) * sizeof(unsigned long long), c->stream));
        if (c->big_count) HIP_TRY(hipMemsetAsync(c->big_count, 0, DSM_BIG_NT * DSM_BIG_NL * DSM_BIG_STRIDE * sizeof(uint32_t), c->stream));
        return DSM_OK;
    };
    float best = 0.f;
    int best_k = 0;
    for (int k = 0; k < DSM_NTAB_PLACES && r == DSM_OK; ++k) {
        c->ntab = c->ntab_base + (size_t)k * 64;
        // the draws are counter-based (the iteration index is an argument): running the pass here moves no stream.  Sums are cleared after.
        r = clear();
        if (r == DSM_OK) r = k_stats_stage1(c, 0xFFFFFF00u + (uint32_t)k);
        if (r == DSM_OK) r = clear();
        if (r != DSM_OK) break;
        // Round 6: the stage-1 launch alone (the compacted kernel and the clears between the events added 10-90 us of work that does not
        // depend on the place -- most on the burn-in state a chain is probed in -- to a difference of 9 us), the FASTEST of three launches
        // per place (a launch is only ever slowed by what else runs): 2 of 6 processes had kept a slow place at config 3 (stage 1 52-55 us
        // instead of 45-47, profiles/r06_stats_ab.txt)
        float ms = 0.f;
        for (int j = 0; j < 3 && r == DSM_OK; ++j) {
            c->stats_probe = true;                                    // k_stats_stage1: no compacted launch behind the pass
            (void)hipEventRecord(ev[0], c->stream);
            r = k_stats_stage1(c, 0xFFFFFF80u + (uint32_t)k);
            (void)hipEventRecord(ev[1], c->stream);
            c->stats_probe = false;
            if (r == DSM_OK) r = clear();
            if (r != DSM_OK) break;
            if (hipEventSynchronize(ev[1]) != hipSuccess) { r = DSM_ERR_HIP; dsm_set_error("stats_place_ntab: hipEventSynchronize failed"); break; }
            float m1 = 0.f;
            (void)hipEventElapsedTime(&m1, ev[0], ev[1]);
            if (j == 0 || m1 < ms) ms = m1;
        }
        if (r != DSM_OK) break;
        if (k == 0 || ms < best) { best = ms; best_k = k; }
        if (verbose) fprintf(stderr, "desman_hip: subset table at +%d B: %.1f us per stage-1 launch\n", k * 256, 1000.0 * ms);
    }
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    c->timing = timing;
    c->ntab = c->ntab_base + (size_t)best_k * 64;
    c->ntab_off = (size_t)best_k * 256;
    if (r != DSM_OK) return r;
    c->ntab_measured = true;
    return clear();
}

static int ensure_big_list(dsm_ctx *c, size_t seg)
{
    const size_t need = seg * DSM_BIG_NT * DSM_BIG_NL;
    if (c->big_list && c->big_cap == need) return DSM_OK;
    if (c->big_list) { (void)hipFree(c->big_list); c->big_list = nullptr; }
    if (!c->big_count) {
        hipError_t e = hipMalloc((void **)&c->big_count, DSM_BIG_NT * DSM_BIG_NL * DSM_BIG_STRIDE * sizeof(uint32_t));
        if (e != hipSuccess) { dsm_set_error("hipMalloc failed: %s", hipGetErrorString(e)); return DSM_ERR_NOMEM; }
    }
    hipError_t e = hipMalloc((void **)&c->big_list, need * sizeof(unsigned long long));
    if (e != hipSuccess) { dsm_set_error("hipMalloc(%zu B) failed: %s", need * 8, hipGetErrorString(e)); return DSM_ERR_NOMEM; }
    c->big_cap = need;
    // the counters are zero between passes: stage 2 (its consumer-side successor) resets them
    HIP_TRY(hipMemsetAsync(c->big_count, 0, DSM_BIG_NT * DSM_BIG_NL * DSM_BIG_STRIDE * sizeof(uint32_t), c->stream));
    return DSM_OK;
}

// spec 4: the word table (4^G entries, all ones = "no pass has written here") and the aggregated counts (zero between passes)
static int ensure_pat(dsm_ctx *c)
{
    const size_t nrep = (size_t)1 << (2 * c->G), nx = (size_t)c->V * 4 * (size_t)c->S;
    if (!c->pat_rep || c->pat_rep_len != nrep) {
        if (c->pat_rep) { (void)hipFree(c->pat_rep); c->pat_rep = nullptr; }
        hipError_t e = hipMalloc((void **)&c->pat_rep, nrep * sizeof(unsigned long long));
        if (e != hipSuccess) { dsm_set_error("hipMalloc(%zu B) failed: %s", nrep * 8, hipGetErrorString(e)); return DSM_ERR_NOMEM; }
        c->pat_rep_len = nrep;
        c->pat_gen = 0;
        HIP_TRY(hipMemsetAsync(c->pat_rep, 0xFF, nrep * sizeof(unsigned long long), c->stream));
    }
    if (!c->pat_list || c->pat_rep_len != nrep || c->pat_gen == 0) {
        if (c->pat_list) { (void)hipFree(c->pat_list); c->pat_list = nullptr; }
        hipError_t e = hipMalloc((void **)&c->pat_list, (nrep + 2) * sizeof(uint32_t));
        if (e != hipSuccess) { dsm_set_error("hipMalloc(%zu B) failed: %s", (nrep + 2) * 4, hipGetErrorString(e)); return DSM_ERR_NOMEM; }
        HIP_TRY(hipMemsetAsync(c->pat_list + nrep, 0, 2 * sizeof(uint32_t), c->stream));
    }
    if (!c->pat_x || c->pat_x_len != nx) {
        if (c->pat_x) { (void)hipFree(c->pat_x); c->pat_x = nullptr; }
        hipError_t e = hipMalloc((void **)&c->pat_x, nx * sizeof(uint32_t));
        if (e != hipSuccess) { dsm_set_error("hipMalloc(%zu B) failed: %s", nx * 4, hipGetErrorString(e)); return DSM_ERR_NOMEM; }
        c->pat_x_len = nx;
        HIP_TRY(hipMemsetAsync(c->pat_x, 0, nx * sizeof(uint32_t), c->stream));
    }
    return DSM_OK;
}

int k_stats_stage1(dsm_ctx *c, uint32_t iter)
{
    int r = ensure_ntab(c);
    if (r != DSM_OK) return r;
    const int S = c->S, G = c->G, V = c->V;
    const int LPV = stats_agg_lpv(S);
    const int NCH = (S + LPV - 1) / LPV, SP = NCH * LPV, NG = 64 / LPV;
    const int spec_full = stats_spec(c);                  // 2, 3 or 4 (the caller checked that the aggregated pass applies)
    const bool pat = spec_full == 4;
    const int spec = stats_draw_version(spec_full);
    static const int regg_env = DSM_AB_ENV("DESMAN_HIP_STATS_REGG") ? atoi(DSM_AB_ENV("DESMAN_HIP_STATS_REGG")) : -1;   // A/B switch
    // (gamma in registers where a lane keeps its sample, instead of the LDS tile: measured slower -- 87-91 VGPRs, i.e. five
    // wavefronts per SIMD, or spills at six: 54-57 us against 48-49; kept as an A/B switch)
    const bool regg = NCH == 1 && G <= 8 && regg_env == 1 && !pat;
    const size_t sh = ((regg ? 0 : (size_t)G * SP) + (spec >= 3 ? 2 * DSM_LOG_TAB_N + DSM_EXP_TAB_N : 0) + DSM_RCP_TAB_N + 16) * sizeof(double) +
                      16 * 256 * sizeof(uint32_t);
    if (sh > 160 * 1024) { dsm_set_error("stats_agg: gamma tile (%zu B) exceeds LDS", sh); return DSM_ERR_UNSUPPORTED; }
    // the instantiation of this shape and specification: {single, batched}
    const void *fn = nullptr, *fn_b = nullptr;
#define AGG_CASE(L, SP_, R)                                                                                               \
    if (LPV == L && spec == SP_ && regg == R) { fn = (const void *)stats_agg_kernel<L, SP_, R>; fn_b = (const void *)stats_agg_kernel_b<L, SP_, R>; }
    AGG_CASE(16, 2, false) AGG_CASE(16, 2, true) AGG_CASE(32, 2, false) AGG_CASE(32, 2, true) AGG_CASE(64, 2, false) AGG_CASE(64, 2, true)
    AGG_CASE(16, 3, false) AGG_CASE(16, 3, true) AGG_CASE(32, 3, false) AGG_CASE(32, 3, true) AGG_CASE(64, 3, false) AGG_CASE(64, 3, true)
#undef AGG_CASE
    if (pat) {
        fn = LPV == 16 ? (const void *)stats_pat_kernel<16> : LPV == 32 ? (const void *)stats_pat_kernel<32> : (const void *)stats_pat_kernel<64>;
        fn_b = LPV == 16 ? (const void *)stats_pat_kernel_b<16> : LPV == 32 ? (const void *)stats_pat_kernel_b<32> : (const void *)stats_pat_kernel_b<64>;
    }
    if (!fn) { dsm_set_error("stats_agg: no kernel for spec %d", spec); return DSM_ERR_UNSUPPORTED; }
    if (c->stats_grid == 0 || c->stats_grid_key != (spec_full * 2 + (regg ? 1 : 0))) {
        int occ = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 256, sh));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, c->device));
        static const int wgs_env = DSM_AB_ENV("DESMAN_HIP_STATS_WGS") ? atoi(DSM_AB_ENV("DESMAN_HIP_STATS_WGS")) : 6;   // A/B switch
        // ... on all CUs but one per XCD when the sweep's uniforms are MT19937 words: the generator's workgroup keeps a CU to itself
        // (kernels_gibbs.hip: k_mt_fill) and workgroups go to the XCDs in turn whatever room they have, so a persistent grid sized for
        // every CU leaves six workgroups of that XCD waiting for others to finish -- a launch that starts while the generator runs took
        // 51 us instead of 46 (scripts/dbg/trace_stats_vs_mt.sh).  Leaving 8 CUs: 106.7 vs 107.5 us per iteration at config 3 (1 CU: no
        // change; 16 and more: slower).
        static const int leave_env = DSM_AB_ENV("DESMAN_HIP_STATS_LEAVE_CUS") ? atoi(DSM_AB_ENV("DESMAN_HIP_STATS_LEAVE_CUS")) : -1;   // A/B switch
        const int n_xcd = prop.multiProcessorCount % 8 == 0 && prop.multiProcessorCount >= 64 ? 8 : 1;
        // (only while a wavefront makes a pass or two: over many passes the smaller grid is simply 3 % less machine -- config 5, twelve
        // passes: 384 vs 368 us)
        const long passes4 = (((long)V * NCH + NG - 1) / NG) / std::max(1L, (long)std::min(wgs_env, std::max(1, occ)) * prop.multiProcessorCount);   // workgroup = 4 wavefronts: passes x 4
        const int leave = leave_env >= 0 ? leave_env : ((c->tau_rng == DSM_RNG_MT19937 && passes4 <= 12) ? n_xcd : 0);
        c->stats_grid = std::min(wgs_env, std::max(1, occ)) * std::max(1, prop.multiProcessorCount - leave);   // six workgroups per CU: measured optimum (below)
        c->stats_grid_key = spec_full * 2 + (regg ? 1 : 0);
    }
    const long ntask = ((long)V * NCH + NG - 1) / NG;            // wavefront passes (NG lane groups = NG tasks each)
    // a persistent grid of six workgroups (24 wavefronts) per CU, passes dealt round-robin: the kernel is issue-bound, and a SIMD
    // with six wavefronts of one or two passes each keeps its VALU busier than five with exactly two (48 vs 52 us at config 3,
    // 143 vs 152 at V = 20k); a seventh workgroup per CU or a second partial round costs more than it brings (7168 wavefronts
    // 53 us, 8192 57 us)
    const long max_waves = (long)c->stats_grid * 4 / (g_batch.K ? g_batch.K : 1);   // the chains of a batch share the persistent grid
    const long waves = std::min<long>(ntask, max_waves);
    const int grid = (int)std::max<long>(1, (waves + 3) / 4);
    // a sub-list holds what its workgroups can defer at most: passes per wavefront x 4 wavefronts x 64 lanes x 4 bases each
    const long per_wg = ((ntask + waves - 1) / waves) * 4 * 64 * 4;
    const size_t seg = (size_t)std::min<long>((long)V * S * 4, ((grid + DSM_BIG_NL - 1) / DSM_BIG_NL) * per_wg);
    r = ensure_big_list(c, seg);
    if (r != DSM_OK) return r;
    StatsAggParams p;
    p.cnt_vs = c->cnt_vs; p.tau = c->tau; p.gamma = c->gamma; p.eta = c->eta;
    p.V = V; p.S = S; p.G = G; p.big_seg = seg;
    p.v_off = c->shard_on ? c->shard_voff : 0; p.V_tot = c->shard_on ? c->shard_vtot : V;
    p.k0 = (uint32_t)c->ctr_seed; p.k1 = (uint32_t)(c->ctr_seed >> 32); p.iter = iter;
    p.ntab = c->ntab; p.rep = c->ntab_rep; p.ld = c->ntab_ld; p.esum = c->esum; p.log_tab = c->log_tab;
    p.xcd = stats_ntab_xcd(c) ? 1 : 0;
    p.hmul = stats_ntab_hmul(); p.swz = stats_ntab_swz();
    p.big_list = c->big_list; p.big_count = c->big_count;
    p.pat_rep = nullptr; p.pat_x = nullptr; p.pat_list = nullptr; p.pat_n = nullptr;
    if (pat) {
        r = ensure_pat(c);
        if (r != DSM_OK) return r;
        p.pat_rep = c->pat_rep; p.pat_x = c->pat_x;
        KTimer tm(c, DSM_K_STATSPAT);
        const uint32_t hi = 0xFFFFFFFFu - (++c->pat_gen);             // later passes write smaller entries: nothing to clear
        const int words = 1 << (2 * G);
        uint32_t *n_this = c->pat_list + words + (c->pat_gen & 1u), *n_next = c->pat_list + words + ((c->pat_gen + 1u) & 1u);
        p.pat_list = c->pat_list; p.pat_n = n_this;
        if (G <= 6) hipLaunchKernelGGL(pat_rep_lds_kernel, dim3(std::max(1, std::min(64, V / 4096))), dim3(1024), 0, c->stream, c->tau, V, G, c->pat_rep, hi);
        else hipLaunchKernelGGL(pat_rep_kernel, dim3((V + 255) / 256), dim3(256), 0, c->stream, c->tau, V, G, c->pat_rep, hi);
        hipLaunchKernelGGL(pat_list_kernel, dim3((words + 255) / 256), dim3(256), 0, c->stream, c->pat_rep, words, hi, c->pat_list, n_this, n_next);
        const size_t ncell = (size_t)V * S;
        // few words: pooled in LDS first (one 1024-thread workgroup per CU and share of the positions); else straight to the blocks
        const int lpw = G <= 3 ? 64 : 32;
        const size_t lds = ((size_t)1 << (2 * G)) * 4 * lpw * sizeof(uint32_t);
        if (G <= 4 && lds <= 128 * 1024 && V >= 4096) {
            const int nchunk = (S + lpw - 1) / lpw;
            const int nvsplit = std::max(1, std::min(256 / nchunk, V / (16 * (64 / lpw) * 8)));     // >= 8 passes per wavefront
            const void *fl = lpw == 64 ? (const void *)pat_agg_lds_kernel<64> : (const void *)pat_agg_lds_kernel<32>;
            static bool attr_done[2] = {false, false};
            if (!attr_done[lpw == 32]) { HIP_TRY(hipFuncSetAttribute(fl, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)); attr_done[lpw == 32] = true; }
            const int32_t *cv = c->cnt_vs; const uint64_t *tp = c->tau; const unsigned long long *rp = c->pat_rep; uint32_t *xp = c->pat_x;
            int Vv = V, Ss = S, Gg = G, nvs = nvsplit;
            void *args[] = {(void *)&cv, (void *)&tp, (void *)&Vv, (void *)&Ss, (void *)&Gg, (void *)&rp, (void *)&xp, (void *)&nvs};
            HIP_TRY(hipLaunchKernel(fl, dim3(nchunk * nvsplit), dim3(1024), args, lds, c->stream));
        } else
            hipLaunchKernelGGL(pat_agg_kernel, dim3((unsigned)std::min<size_t>((ncell + 255) / 256, 16384)), dim3(256), 0, c->stream,
                               c->cnt_vs, c->tau, V, S, G, c->pat_rep, c->pat_x);
        HIP_TRY(hipGetLastError());
    }
    {
        // who draws an item, not what is drawn: an item whose rarer outcome has a mean above lean_cap goes to the compacted
        // kernel although inversion still applies to it -- its search would hold its 63 neighbours of the wavefront for
        // ~lean_cap more steps.  Config 3 at 10 x depth (us, this pass + compacted kernel): 128 -> 150 + 35, 64 -> 119 + 53,
        // 48 -> 105 + 81, 32 -> 86 + 134.  Data of ordinary depth has no such item once the chain has converged.
        static const char *e = DSM_AB_ENV("DESMAN_HIP_LEAN_CAP");
        p.lean_cap = e ? atof(e) : DSM_LEAN_CAP;
        static const int dbg = DSM_AB_ENV("DESMAN_HIP_STATS_DBG") ? atoi(DSM_AB_ENV("DESMAN_HIP_STATS_DBG")) : 0;
        p.dbg = dbg;
        if (!(p.lean_cap > 0.0 && p.lean_cap <= DSM_BINV_MEAN_CAP)) p.lean_cap = DSM_LEAN_CAP;
    }
    // (a batch: fewer workgroups per list, the launch holds K times as many lists)
    static const int big_wgs_env = DSM_AB_ENV("DESMAN_HIP_BIG_WGS") ? atoi(DSM_AB_ENV("DESMAN_HIP_BIG_WGS")) : 0;     // A/B switch: workgroups per list
    const int big_wgs = big_wgs_env > 0 ? big_wgs_env : 16;
    const int big_grid = DSM_BIG_NT * DSM_BIG_NL *
        (int)std::max<long>(1, std::min<long>((ntask * 64 * 4 / DSM_BIG_NL + 255) / 256, g_batch.K ? std::max(2, big_wgs / g_batch.K) : big_wgs));
    if (g_batch.K) {
        // two launches of the chain's pass, each collected on its own: stage 1, then the deferred items
        static thread_local BatchArgs<StatsAggParams> acc;
        acc.p[g_batch.k] = p;
        if (g_batch.k == g_batch.K - 1) {
            const dim3 g(grid, g_batch.K);
            void *args[] = {(void *)&acc};
            HIP_TRY(hipLaunchKernel(fn_b, g, dim3(256), args, sh, c->stream));
            if (spec >= 3) hipLaunchKernelGGL(stats_big_kernel_b<3>, dim3(big_grid, g_batch.K), dim3(256), 0, c->stream, acc);
            else hipLaunchKernelGGL(stats_big_kernel_b<2>, dim3(big_grid, g_batch.K), dim3(256), 0, c->stream, acc);
        }
        HIP_TRY(hipGetLastError());
        return DSM_OK;
    }
    {
        KTimer tm(c, DSM_K_STATS);
        void *args[] = {(void *)&p};
        HIP_TRY(hipLaunchKernel(fn, dim3(grid), dim3(256), args, sh, c->stream));
    }
    if (c->stats_probe && !pat) return DSM_OK;           // (stats_place_ntab times the pass alone and clears the lists itself; over tau words the
                                                         // compacted kernel is also what leaves a handed-over item's pooled count zero: it runs)
    KTimer tm(c, DSM_K_STATSBIG);
    // the deferred items (none once the chain has converged on data of ordinary depth: the launch then returns at once);
    // up to 16 workgroups per list = 4096 items of a list per round (a list one item longer than a round doubles the launch:
    // every wavefront is one long dependent chain); the workgroups of an empty list leave at once
    if (spec >= 3) hipLaunchKernelGGL(stats_big_kernel<3>, dim3(big_grid), dim3(256), 0, c->stream, p);
    else hipLaunchKernelGGL(stats_big_kernel<2>, dim3(big_grid), dim3(256), 0, c->stream, p);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

// the halving tree of stage 2 depends on G only: laid out on the host, read through scalar registers on the device
S2Plan make_stage2_plan(int G)
{
    S2Plan pl;
    memset(&pl, 0, sizeof pl);
    int n = 0, off = 0;
    pl.lo[0] = 0; pl.hi[0] = G; pl.off[0] = 0; pl.idx[0] = 0; pl.child[0] = -1;
    n = 1;
    int lvl_begin = 0, lvl_end = 1, level = 0;
    pl.level_start[0] = 0;
    while (lvl_begin < lvl_end) {
        for (int i = lvl_begin; i < lvl_end; ++i) {
            const int w = pl.hi[i] - pl.lo[i];
            if (w == 1) { pl.child[i] = -1; continue; }
            const int wl = w / 2, mid = pl.lo[i] + wl;
            pl.child[i] = n;
            pl.lo[n] = pl.lo[i]; pl.hi[n] = mid; pl.off[n] = off; pl.idx[n] = 2 * pl.idx[i]; off += 1 << wl; ++n;
            pl.lo[n] = mid; pl.hi[n] = pl.hi[i]; pl.off[n] = off; pl.idx[n] = 2 * pl.idx[i] + 1; off += 1 << (w - wl); ++n;
        }
        lvl_begin = lvl_end; lvl_end = n;
        pl.level_start[++level] = lvl_begin;
    }
    pl.nlevels = level;                 // levels 0 .. nlevels-1 hold nodes; level_start[nlevels] = n
    pl.tab_entries = off;
    return pl;
}

int k_stats_stage2(dsm_ctx *c, uint32_t iter)
{
    KTimer tm(c, DSM_K_STATS2);
    Stage2Params p;
    p.ntab = c->ntab; p.rep = c->ntab_rep; p.ld = c->ntab_ld; p.gamma = c->gamma; p.sum_mu = c->sum_mu; p.log_tab = c->log_tab;
    p.S = c->S; p.G = c->G; p.hmul = stats_ntab_hmul(); p.swz = stats_ntab_swz();
    p.k0 = (uint32_t)c->ctr_seed; p.k1 = (uint32_t)(c->ctr_seed >> 32); p.iter = iter;
    p.big_count = c->big_count;
    // 2^G subsets per sample at the root: 256 threads up to G = 9, 1024 above
    const int nthr = c->G >= 10 ? 1024 : 256;
    const bool v3 = stats_draw_version(stats_spec(c)) >= 3;          // (spec 4 draws with version 2's samplers)
    // many subsets per sample (G >= 11): the root level -- one large-count binomial per subset that straddles the halves, 2^G / 1024 of
    // them per thread one after the other -- is dealt over `nsplit` workgroups per sample; the last one to finish takes the level-1
    // tables the others left in `s2_scratch` and runs the lower levels (dsm_stage2.h).  Same draws: a binomial's stream is keyed by
    // (subset, sample, node, level), the tables are integer sums.
    p.nsplit = 1; p.scratch = nullptr; p.ticket = nullptr;
    if (g_batch.K == 0 && c->G >= 11) {
        if (!c->s2_scratch) {
            hipError_t e = hipMalloc((void **)&c->s2_scratch, ((size_t)DSM_MAX_S * (S2_TAB_ENTRIES + 1)) * sizeof(uint32_t));
            if (e != hipSuccess) { dsm_set_error("hipMalloc failed: %s", hipGetErrorString(e)); return DSM_ERR_NOMEM; }
            HIP_TRY(hipMemsetAsync(c->s2_scratch, 0, ((size_t)DSM_MAX_S * (S2_TAB_ENTRIES + 1)) * sizeof(uint32_t), c->stream));
        }
        p.nsplit = c->G >= 12 ? 4 : 2;
        p.scratch = c->s2_scratch; p.ticket = c->s2_scratch + (size_t)DSM_MAX_S * S2_TAB_ENTRIES;
    }
    if (g_batch.K == 0) {
        if (v3) hipLaunchKernelGGL(stats_stage2_kernel<3>, dim3(c->S * p.nsplit), dim3(nthr), 0, c->stream, p, make_stage2_plan(c->G));
        else hipLaunchKernelGGL(stats_stage2_kernel<2>, dim3(c->S * p.nsplit), dim3(nthr), 0, c->stream, p, make_stage2_plan(c->G));
    } else {
        static thread_local Stage2Batch acc;
        acc.p[g_batch.k] = p;
        if (g_batch.k == g_batch.K - 1) {
            acc.plan = make_stage2_plan(c->G);
            if (v3) hipLaunchKernelGGL(stats_stage2_kernel_b<3>, dim3(c->S, g_batch.K), dim3(nthr), 0, c->stream, acc);
            else hipLaunchKernelGGL(stats_stage2_kernel_b<2>, dim3(c->S, g_batch.K), dim3(nthr), 0, c->stream, acc);
        }
    }
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

// A2 for the resident state: the aggregated pass (spec DSM_STATS_AGG = 2, or 3 when forced) where it applies, else the per-read pass (spec 1)
int k_stats(dsm_ctx *c, uint32_t iter)
{
    if (stats_spec(c) >= 2) {
        int r = k_stats_stage1(c, iter);
        if (r != DSM_OK) return r;
        return k_stats_stage2(c, iter);
    }
    return k_stats_v1(c, iter);
}

int k_binom_test(dsm_ctx *c, int kind, uint32_t n, const double *w, uint64_t seed, int nsamp, uint32_t *d_out, int spec)
{
    if (spec >= 3) hipLaunchKernelGGL(binom_test_kernel<3>, dim3((nsamp + 255) / 256), dim3(256), 0, c->stream, kind, n, w[0], w[1], w[2], w[3],
                                      (uint32_t)seed, (uint32_t)(seed >> 32), nsamp, c->log_tab, d_out);
    else hipLaunchKernelGGL(binom_test_kernel<2>, dim3((nsamp + 255) / 256), dim3(256), 0, c->stream, kind, n, w[0], w[1], w[2], w[3],
                       (uint32_t)seed, (uint32_t)(seed >> 32), nsamp, c->log_tab, d_out);
    HIP_TRY(hipGetLastError());
    return DSM_OK;
}

// dsm_host.h -- host-side context and kernel-launcher prototypes (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/desman_hip.h"

#define DSM_MAX_GRID 4096
#define DSM_ESUM_PARTS 64   // copies of Esum the workgroups of stage 1 add to (workgroup b: copy b mod 64); the compacted kernel folds them into Esum
#define DSM_BIG_NL 64        // sub-lists of deferred stage-1 items (kernels_stats.hip), one counter each ...
#define DSM_BIG_STRIDE 16    // ... 64 B apart
#define DSM_BIG_NT 3         // ... for each kind of deferred item (BTRS / long search / search + two more binomials)
#define DSM_U_CHUNK 8        // MT19937 words are generated in chunks of up to this many sweeps (api.hip: SweepWords)

#define DSM_STATS_AGG 2      // the version of the aggregated mu/E specification that runs by default (oracle/stats_agg.c); version 3 (table
                             // exp / log instead of the squaring loop) is selectable: measured 2 us slower in stage 1 at config 3
#define DSM_MAX_BATCH 8      // chains of one shape that share every launch of the Gibbs loop (api.hip: dsm_batch_gibbs_update)

void dsm_set_error(const char *fmt, ...);

// A/B switches of the timing experiments (scripts/dbg/): environment variables that move work around, replace a kernel by an
// older form or -- DESMAN_HIP_STATS_DBG -- switch parts of a kernel OFF for ablation timing (results are garbage then).  They exist only
// in the experiment build `make -C desman_amd/csrc ab` (-DDSM_AB_SWITCHES -> lib/libdesman_hip_ab.so, loaded with DESMAN_HIP_LIB=...); in
// the product library every one of them compiles to its default and no environment variable can change what a kernel computes.  What
// the product library does read: DESMAN_HIP_DEVICE, DESMAN_HIP_STATS_SPEC (which mu/E specification unforced contexts follow),
// DESMAN_HIP_ONE_STREAM / DESMAN_HIP_NMFT_GRAPH (set by desman-sweep for concurrent chains), DESMAN_HIP_NTAB_TUNE / DESMAN_HIP_TAU_ORDER
// (= 0: the measured table place / the fp64-blocks-first order off: same results, tests/test_gpu_fuzz.py), DESMAN_HIP_RCCL.
#ifdef DSM_AB_SWITCHES
#define DSM_AB_ENV(name) getenv(name)
#else
#define DSM_AB_ENV(name) ((const char *)nullptr)
#endif

// ---- batched launches: K chains of the same shape, one launch per kernel of the iteration with the chain in blockIdx.y.
// A launcher called while g_batch.K > 0 stores its parameter block in slot g_batch.k and launches -- on the K-th call -- the
// _b form of its kernel, whose argument is the array of parameter blocks (kernarg, indexed by blockIdx.y).  The chains of a
// batch share the leader's streams for the duration of the call.
struct BatchCtl { int K = 0, k = 0; };
extern thread_local BatchCtl g_batch;
template <class P> struct BatchArgs { P p[DSM_MAX_BATCH]; };
// inside a launcher: LAUNCH_OR_COLLECT(params type, param block, normal launch statement, batched launch statement using `acc` and `K`)
#define LAUNCH_OR_COLLECT(PT, P, NORMAL, BATCHED)                                          \
    do {                                                                                    \
        if (g_batch.K == 0) { NORMAL; }                                                     \
        else {                                                                              \
            static thread_local BatchArgs<PT> acc;                                          \
            acc.p[g_batch.k] = (P);                                                         \
            if (g_batch.k == g_batch.K - 1) { const int K = g_batch.K; (void)K; BATCHED; }  \
        }                                                                                   \
    } while (0)

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            dsm_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                          __LINE__);                                                        \
            return DSM_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

struct TimedSpan { int k; hipEvent_t e0, e1; };

struct dsm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream_rng = nullptr;   // MT19937 refills run here, overlapped with the mu/E pass
    hipEvent_t ev_u_ready[2] = {nullptr, nullptr}, ev_u_free[2] = {nullptr, nullptr};
    // sizes
    int V = 0, S = 0, G = 0;
    // count tensor (both layouts, int32) and data-only ll constant
    int32_t *cnt_vs = nullptr;      // [V][S][4]  tau sweep / LL: lane = sample
    int32_t *items = nullptr;       // [S][item_stride][2] mu/E pass work items {v*4+b, reads}, sorted by reads per sample
    int32_t *nitems = nullptr;      // [S] items with a non-zero count
    int max_items = 0;
    bool items_built = false;       // the work list of the per-read pass (spec v1) is built on first use
    uint64_t max_depth = 0;         // largest per-sample read total
    int force_stats_spec = 0;       // 0 = by the shape rule (1 or DSM_STATS_AGG); 1 = per-read pass everywhere; 2 / 3 = that version of
                                    // the aggregated pass wherever it applies, small problems too
    uint32_t *ntab = nullptr;       // [2^G][S] subset counts of the aggregated mu/E pass (spec v2), zero between passes
    size_t ntab_len = 0;
    // pattern-aggregated stage 1 (spec 4, kernels_stats.hip): positions that share their packed tau word share one stage-1 cell per sample
    unsigned long long *pat_rep = nullptr;   // [4^G] (~generation << 32 | lowest position that carries the word); atomicMin, never cleared
    uint32_t *pat_x = nullptr;               // [V][4][S] counts summed per (representative position, base, sample); zero between passes
    uint32_t *pat_list = nullptr;            // [4^G] the representatives of this pass, in no particular order, then [2] their number (by pass parity)
    size_t pat_rep_len = 0, pat_x_len = 0;
    uint32_t pat_gen = 0;
    uint32_t *s2_scratch = nullptr; // stage 2 as its own launch with a sample's root level shared by several workgroups (kernels_stats.hip: k_stats_stage2)
    uint32_t *ntab_raw = nullptr;   // the allocation `ntab` points into (kernels_stats.hip: ensure_ntab places the table inside it)
    uint32_t *ntab_base = nullptr;  // first place the table can start at (4 KB aligned)
    size_t ntab_off = 0;            // where past ntab_base the table starts (stats_place_ntab)
    bool ntab_placed = false;       // the place has been measured (or given)
    bool ntab_measured = false;     // ... measured: the table goes to the process's pool when the chain ends (kernels_stats.hip: stats_release_ntab)
    int ntab_ld = 0;                // row stride of the table in words (kernels_stats.hip: stats_ntab_ld)
    int ntab_xcd = 0;               // 1: the table has a copy per XCD (experiment switch; part of the pool key of a placed table)
    int ntab_rep = 1;               // copies of the table (few subsets x many positions: kernels_stats.hip, stats_ntab_rep)
    unsigned long long *big_list = nullptr;   // stage-1 items deferred to the compacted (BTRS) kernel: cell * 4 + base
    uint32_t *big_count = nullptr;            // DSM_BIG_NL counters, DSM_BIG_STRIDE words apart
    size_t big_cap = 0;
    int stats_grid = 0;             // resident workgroups of stats_agg_kernel
    int stats_grid_key = -1;        // ... of which instantiation (specification x register-gamma form)
    int item_stride = 1;            // items per sample row of `items`
    bool chunked = false;           // items carry reads | chunk << 12 (small problems, see dsm_ctx_set_counts)
    int32_t *blk_tab = nullptr;     // [blk_n][3] workgroup -> {sample, j, n_j} of the mu/E pass
    int blk_n = 0, blk_gmax = 0;
    std::vector<int64_t> depth;     // host: total reads per sample
    std::vector<int32_t> nitems_h;  // host copy of nitems
    double ll_const = 0.0;
    // chain state
    uint64_t *tau = nullptr;        // [V] packed, 2 bits per haplotype
    double *gamma = nullptr;        // [S][G]
    double *eta = nullptr;          // [4][4]
    double *eta_new = nullptr;      // [4][4]
    bool have_state = false;
    double alpha = 0.1, delta = 0.1, epsilon = 1e-6;
    // sufficient statistics of the auxiliary counts
    unsigned long long *sum_mu = nullptr;   // [S][G]
    bool stats_probe = false;               // stats_place_ntab is timing stage 1: k_stats_stage1 leaves the compacted launch out
    unsigned long long *esum = nullptr;     // [4][4] [observed][true], followed by DSM_ESUM_PARTS x [4][4] partial sums of stage 1 (kernels_stats.hip: zero between passes)
    // RNG
    uint32_t *mt_state = nullptr;   // 624 words + position
    uint32_t *mt_jstates = nullptr; // starting arrays of the chunks of a parallel fill (kernels_gibbs.hip: mt_fill_parallel), made on first use
    bool mt_seeded = false;
    uint32_t *u_raw = nullptr;      // [2][u_chunk_words] raw MT19937 words: two slots of several sweeps each
    size_t u_cap = 0;               // words per sweep = V*G
    size_t u_chunk_words = 0;
    bool mt_attr_set = false;       // dynamic-LDS limit of the MT19937 kernel raised on this context's device
    uint64_t ctr_seed = 0x243F6A8885A308D3ull;
    uint32_t iter_ctr = 0;          // global iteration counter for counter-based draws
    int tau_rng = DSM_RNG_MT19937;
    // scratch
    double *ll_partial = nullptr;   // [DSM_MAX_GRID]
    int *nchange = nullptr;         // device counter
    unsigned long long *sweep_stats = nullptr;   // [2] wavefront-steps of the tau sweeps: run / decided by the fp64 code
    uint32_t *blk_order = nullptr;  // [2 slots][DSM_MAX_GRID] block order of the next sweep of the slot (kernels_gibbs.hip: finalize_body)
    int tau_resident = 0, tau_resident_key = -1;   // workgroups of the sweep the device holds at once, for (S, G) = key
    int blk_order_n[2] = {0, 0};    // the grid each was made for (0: none yet)
    uint32_t *step_cnt = nullptr;   // [2 slots][DSM_MAX_GRID][2] per-workgroup wavefront-steps of a tau launch: run / left to fp64
    uint32_t *screen_ctl = nullptr; // [4] [0] = sweeps still to run without the screening pass (set by finalize_body)
    bool tau_screen = true;         // dsm_ctx_set_tau_screen: the fp32 screening pass of the tau sweep may run
    int tau_neartie_mode = -1;      // dsm_ctx_set_tau_neartie: -1 = by the chain's abundances at the start of each Gibbs call, 0 = never, 1 = always
    bool tau_neartie_on = false;    // this call runs the sweep's instantiation with the near-tie screen (kernels_gibbs.hip: k_tau_neartie_hint)
    int tau_rare_n = 0;             // haplotypes at or below DSM_NT_RARE in every sample, as last known: from the host's gamma (set_state /
                                    // set_gamma_eta), from the device's at the end of a Gibbs call and every 64 iterations inside one
    int *h_rare = nullptr;          // pinned word the count is read back into (no synchronisation of its own at the end of a call)
    double *prior_all = nullptr;    // [n_iter][S + 4] priors of the stored states of updateTau
    double *prior = nullptr;        // [2][DSM_MAX_S + 4] per-row Dirichlet log-prior terms, by iteration parity
    double *scalars = nullptr;      // [8] misc device scalars
    double *log_tab = nullptr;      // [256][2] table of dsm_log (log_table.h)
    // traces of the last update call
    int n_trace = 0, trace_cap = 0;
    uint64_t *tau_trace = nullptr;  // [(n+1)][V]; slot 0 = entry state
    double *ll_trace = nullptr, *lp_trace = nullptr;   // [n]
    int *nchange_trace = nullptr;   // [n]
    double *gamma_trace = nullptr;  // [n][S][G]
    double *eta_trace = nullptr;    // [n][16]
    double *gamma_in = nullptr, *eta_in = nullptr;   // updateTau inputs
    int in_cap = 0;                 // iterations gamma_in / eta_in / prior_all have room for
    // MAP ("star") tracking: {lp_star, (double)slot}
    double *star = nullptr;         // [2]
    double *gamma_star = nullptr;   // [S][G]
    double *eta_star = nullptr;     // [16]
    // a chain sharded over GPUs by positions (dsm_ctx_gibbs_update_sharded): this context holds positions shard_voff .. + V of shard_vtot
    bool shard_on = false;
    int shard_voff = 0, shard_vtot = 0;
    double *shard_vec = nullptr;    // [18] {ll, nchange, Esum[16]} of the per-iteration exchange
    // NMFT
    double *F = nullptr;            // [V][4][S]
    double *ntau = nullptr;         // [V][4][G]
    double *ntau2 = nullptr;        // [V][4][G] the other buffer of factorize_tau's fused pass (kernels_nmft.hip: NmftMfmaParams.fix_gamma == 2), made on first use
    double *ngam = nullptr;         // [G][S] (after _adjustment)
    double *ngam_raw = nullptr;     // [G][S] normalised gamma of the running update, before _adjustment
    int npart_cols = 0;             // workgroup partials currently held by npart
    double *npart = nullptr;        // per-block partials
    double *nstat = nullptr;        // reduced statistics + control words
    double *ndiv_trace = nullptr;   // objective after every update of the running factorize (or null)
    int nG = 0;
    int nmft_blocks = 0;
    int nmft_persist = -1;          // the factorize loop as one persistent launch where the table fits (kernels_nmft.hip): -1 / 1 = yes, 0 = never
    double *np_part = nullptr;      // its exchange buffers: partials [nout][workgroups] + totals [nout]
    size_t np_cap = 0;
    unsigned *np_bar = nullptr;     // its barrier words
    int nmft_fix_gamma = 0;         // set for the duration of a factorize_tau loop (api.hip: dsm_nmft_factorize): the update kernels then leave out the gamma numerators
    int nmft_fused = -1;            // the gamma/control step of an update: -1 = by size (<= 128 partials: one launch with the reduction; more: inside the update
                                    // kernel's start on the matrix-core path), 0 = a launch of its own, 1 = always with the reduction, 3 = inside the update kernel wherever it can
    double *ngam2 = nullptr, *ngam_raw2 = nullptr;     // the other parity's gamma buffers of the update kernel's own gamma step (NmftMfmaParams.gstep)
    int nmft_gstep = 0, nmft_gstep_parity = 0, nmft_gstep_max_iter = 0;    // set by dsm_nmft_factorize around the launch
    double nmft_gstep_min_change = 0.0;
    // timing
    bool timing = false;
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> free_events;
    double k_ms[DSM_K_COUNT] = {0};
    int64_t k_launches[DSM_K_COUNT] = {0};
};

struct KTimer {
    dsm_ctx *c; int k; hipStream_t st; hipEvent_t e0 = nullptr, e1 = nullptr;
    KTimer(dsm_ctx *ctx, int kid, hipStream_t stream = nullptr);
    ~KTimer();
};

// ---- launchers (kernels_gibbs.hip)
int k_convert_counts(dsm_ctx *c, const int64_t *d_in, int *d_flag, double *d_partial, int nblk, unsigned long long *d_depth);
int k_pack_tau(dsm_ctx *c, const int64_t *d_onehot, uint64_t *d_packed, int V, int G);
int k_unpack_tau(dsm_ctx *c, const uint64_t *d_packed, int64_t *d_onehot, int V, int G);
int k_tau_sum(dsm_ctx *c, const uint64_t *trace, int n, int64_t *d_sum);
int k_mt_fill(dsm_ctx *c, uint32_t *out, size_t n, hipStream_t stream);
int k_stats_v1(dsm_ctx *c, uint32_t iter);
int k_log2f_test(dsm_ctx *c, const float *d_in, float *d_out, size_t n);
int build_stats_items(dsm_ctx *c);          // api.hip: work list of the per-read pass from the resident tensor

// ---- launchers (kernels_stats.hip)
uint32_t stats_ntab_hmul();                  // odd multiplier of the subset -> table row map
int stats_place_ntab(dsm_ctx *c);         // measures where the subset table should start (once per table; kernels_stats.hip)
void stats_release_ntab(dsm_ctx *c);     // the table leaves the context: to the process's pool of placed tables, or freed
#define DSM_NTAB_PAD 0               // words added to a row of the subset table when S is a multiple of 64 (stats_ntab_ld)
int stats_ntab_ld(int S);
uint32_t stats_ntab_swz();                 // the sample's part of the subset table's row map
#define DSM_NTAB_SWZ 17u                // (any odd number: see stats_ntab_swz)
void stats_ntab_pool_release();            // kernels_stats.hip: frees the pooled subset tables
void mt_jump_release();                    // kernels_gibbs.hip: frees the MT19937 jump tables
int stats_ntab_rep(const dsm_ctx *c);       // copies of the subset table the stage-1 atomics are spread over
int stats_spec(const dsm_ctx *c);           // 2 / 3 = aggregated sampler (oracle/stats_agg.c), 4 = the same over tau patterns, 1 = per-read (orc_stats_counter)
static inline int stats_draw_version(int spec) { return spec == 4 ? 2 : spec; }   // which version of the samplers (dsm_binom.h: SPEC) a specification draws with
int k_tau_neartie_hint(dsm_ctx *c);
int k_tau_rare_count(dsm_ctx *c, bool wait);         // enqueue the count of rare haplotypes of the resident gamma -> *h_rare (wait: and take it)
int tau_rare_from_host(const double *gamma, int S, int G);
int k_stats(dsm_ctx *c, uint32_t iter);
int k_stats_stage1(dsm_ctx *c, uint32_t iter);
int k_stats_stage2(dsm_ctx *c, uint32_t iter);
int k_binom_test(dsm_ctx *c, int kind, uint32_t n, const double *w, uint64_t seed, int nsamp, uint32_t *d_out, int spec);
int k_esum_fold(dsm_ctx *c);             // kernels_stats.hip: the copies of Esum stage 1 adds to -> Esum
int k_dirichlet(dsm_ctx *c, uint32_t iter, double *gamma_out, double *gamma_trace, double *eta_out, double *eta_trace,
                double *prior_out, int fin_it, int fin_nblocks, const double *fin_prior, int do_s2 = 0);
int k_prior(dsm_ctx *c, const double *gamma, const double *eta, double *prior_out);
int k_prior_batch(dsm_ctx *c, const double *gamma, const double *eta, int n, double *prior_out);
// the previous sweep's finalize riding in a tau launch (updateTau): what k_finalize would have been called with
struct TauFinalRider { int nblocks, it; const double *prior, *gamma_src, *eta_src; };
// mode bit0 = sweep, bit1 = log-likelihood epilogue; slot = parity of the ll_partial / nchange buffers the launch writes
int k_tau_sweep(dsm_ctx *c, int mode, const double *gamma, const double *eta_sweep,
                const double *eta_ll, uint64_t *trace_slot, double *d_logp, uint32_t iter, int *nblocks,
                const uint32_t *u_raw, int slot = 0, const TauFinalRider *rider = nullptr);
int tau_launch_info(dsm_ctx *c, int *launched, int *resident);
int k_shard_pack(dsm_ctx *c, int nblocks);
int k_shard_unpack(dsm_ctx *c);
int k_finalize(dsm_ctx *c, int nblocks, int it, int star_mode, const double *prior, const double *gamma_src,
               const double *eta_src, int slot = 0, int *rare_out = nullptr);

// ---- launchers (kernels_nmft.hip)
int k_nmft_freq(dsm_ctx *c);
int k_nmft_clamp(dsm_ctx *c);
int k_nmft_pass_a(dsm_ctx *c);
int k_nmft_gamma(dsm_ctx *c, int max_iter, double min_change, int fix_gamma, int adjust, int parity);   // parity = launch number & 1 since the control words were zeroed
int k_nmft_reduce(dsm_ctx *c);
bool nmft_gstep_applies(const dsm_ctx *c, int fix_gamma);
int k_nmft_pass_b(dsm_ctx *c, int adjust);
int k_nmft_get_tau(dsm_ctx *c, uint64_t *d_packed);
int nmft_grid(dsm_ctx *c);
bool nmft_use_wave(const dsm_ctx *c);
bool nmft_use_mfma(const dsm_ctx *c);
bool nmft_use_wide(const dsm_ctx *c);                      // 128 < S <= 512: nmft_split_kernel
int nmft_wide_grid(const dsm_ctx *c);
int nmft_wave_grid(const dsm_ctx *c);
int nmft_mfma_grid(const dsm_ctx *c, bool fix = false);     // up to four workgroups per CU (five for the fused pass of factorize_tau)
int k_nmft_wave(dsm_ctx *c, int adjust, int do_update);
int k_nmft_persist(dsm_ctx *c, int max_iter, double min_change, int fix_gamma, int adjust, int *used);

/*
 * desman_hip.h -- C ABI of libdesman_hip.so, the MI355X (gfx950) implementation
 * of DESMAN's haplotype-inference hot path.
 *
 * Two layers:
 *
 *  (1) LEGACY SHIM -- the four entry points the reference's Cython module
 *      `sampletau` binds (sampletau/sampletau.pyx:13-19), with identical
 *      argument meaning: borrowed host pointers, tau mutated in place, one
 *      process-global MT19937 stream.  Differences: explicit int64_t, and
 *      errors come back as negative return codes instead of exit(1)
 *      (c_sample_tau.c:200-203).
 *
 *  (2) CONTEXT API -- the device-resident fast path.  One `dsm_ctx` owns the
 *      count tensor, the chain state (tau, gamma, eta) and all traces in HBM;
 *      the host never touches V-sized arrays per iteration.  It covers the
 *      Python-level hot loops of the reference as well
 *      (desman/HaploSNP_Sampler.py:263-365,431-461; desman/Init_NMFT.py:98-205).
 *
 * Plain pointers and sizes only; no torch / numpy types.  All functions return
 * DSM_OK (0) or a negative DSM_ERR_* code; dsm_last_error() gives the message.
 * All arrays are C-contiguous.  Layouts follow the reference:
 *   tau      [V][G][4] int64 one-hot          (HaploSNP_Sampler.py:68)
 *   gamma/pi [S][G]    f64                    (HaploSNP_Sampler.py:63)
 *   eta      [4][4]    f64, rows = true base, cols = observed base
 *   variants [V][S][4] int64 counts           (HaploSNP_Sampler.py:52)
 */
#ifndef DESMAN_HIP_H
#define DESMAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSM_OK               0
#define DSM_ERR_HIP         -1   /* a HIP runtime call failed                 */
#define DSM_ERR_ARG         -2   /* bad argument / shape                      */
#define DSM_ERR_STATE       -3   /* call order (no counts / no state / no RNG)*/
#define DSM_ERR_UNSUPPORTED -4   /* size outside the compiled kernel range    */
#define DSM_ERR_NOMEM       -5
#define DSM_ERR_NODEVICE    -6   /* no gfx950 device visible                  */
#define DSM_ERR_COMM        -7   /* an RCCL call failed (dsm_comm_*)           */

#define DSM_MAX_G  32            /* haplotypes: tau is packed 2 bits each     */
#define DSM_MAX_S  512           /* samples per variant row                   */

/* tau-sweep uniform source */
#define DSM_RNG_MT19937 0        /* GSL-compatible serial stream (replay)     */
#define DSM_RNG_PHILOX  1        /* counter-based, keyed by (seed, iter, v,g) */

const char *dsm_last_error(void);
int dsm_device_count(void);
const char *dsm_version(void);

/* ---------------------------------------------------------------- (1) ---- */
/* replaces c_initRNG  (sampletau/c_sample_tau.c:26-34)                       */
int dsm_initRNG(void);
/* replaces c_setRNG   (sampletau/c_sample_tau.c:36-40); seed 0 -> 4357 (GSL) */
int dsm_setRNG(unsigned long seed);
/* replaces c_freeRNG  (sampletau/c_sample_tau.c:42-45)                       */
int dsm_freeRNG(void);
/* replaces c_sample_tau (sampletau/c_sample_tau.c:95-204).  Returns the number
 * of (v,g) whose base changed (>= 0) or a negative DSM_ERR_* code.           */
int dsm_sample_tau(int64_t *tau, const double *pi, const double *eta,
                   const int64_t *variants, int nV, int nG, int nS);
/* the same four entry points under the reference's own names and prototypes (what the four bare `cdef extern`
 * declarations of sampletau/sampletau.pyx:13-19 bind -- there is no header in the reference; c_sample_tau.c:26,36,42,95),
 * so the unmodified .pyx links against libdesman_hip.so.  Errors go to stderr; c_sample_tau then returns -1.        */
void c_initRNG(void);
void c_setRNG(unsigned long seed);
void c_freeRNG(void);
int c_sample_tau(long *tau, double *pi, double *eta, long *variants, int nV, int nG, int nS);
/* read / write the process-global MT19937 stream of the shim (624 words +
 * position): lets a device-resident context continue the same logical stream. */
int dsm_getRNG_state(uint32_t *state625);
int dsm_setRNG_state(const uint32_t *state625);

/* ---------------------------------------------------------------- (2) ---- */
typedef struct dsm_ctx dsm_ctx;

int dsm_ctx_create(dsm_ctx **out, int device);
int dsm_ctx_destroy(dsm_ctx *ctx);
int dsm_ctx_sync(dsm_ctx *ctx);

/* upload the count tensor (HaploSNP_Sampler.py:52); builds the int32 HBM
 * layouts and the data-only log-likelihood constant.                         */
int dsm_ctx_set_counts(dsm_ctx *ctx, const int64_t *variants, int V, int S);

/* chain state in / out.  set_state also (re)defines G.                       */
int dsm_ctx_set_state(dsm_ctx *ctx, const int64_t *tau, const double *gamma,
                      const double *eta, int G);
int dsm_ctx_get_state(dsm_ctx *ctx, int64_t *tau, double *gamma, double *eta);
int dsm_ctx_set_gamma_eta(dsm_ctx *ctx, const double *gamma, const double *eta);

/* priors / clamp of the sampler (HaploSNP_Sampler.py:31: alpha_constant,
 * delta_constant, epsilon); defaults 0.1, 0.1, 1e-6.                          */
int dsm_ctx_set_priors(dsm_ctx *ctx, double alpha, double delta, double epsilon);

/* RNG: mt_seed feeds the GSL-compatible tau stream (bin/desman:131-132),
 * ctr_seed keys every counter-based draw (mu/E, gamma, eta).                 */
int dsm_ctx_seed(dsm_ctx *ctx, unsigned long mt_seed, uint64_t ctr_seed);
int dsm_ctx_set_tau_rng(dsm_ctx *ctx, int mode /* DSM_RNG_* */);
/* export / import the MT19937 stream (624 state words + position), so that one
 * logical GSL stream can continue across contexts, as the reference's single
 * global generator does across sampler objects (bin/desman:131,149-153,199-201). */
int dsm_ctx_get_mt_state(dsm_ctx *ctx, uint32_t *state625);
int dsm_ctx_set_mt_state(dsm_ctx *ctx, const uint32_t *state625);
/* test hook: the next n raw 32-bit words of the context's MT19937 stream (advances it), host buffer.  */
int dsm_ctx_debug_mt_fill(dsm_ctx *ctx, size_t n, uint32_t *out);
/* fill `state625` from a seed exactly as gsl_rng_set(mt19937, seed) would. */
int dsm_mt_seed_state(unsigned long seed, uint32_t *state625);

/* A1: one tau sweep on the resident state (c_sample_tau.c:95-204) with the
 * resident gamma / eta (see dsm_ctx_set_gamma_eta).  logp_out (optional, host,
 * [V][G][4]) receives the un-normalised conditional log-probabilities.       */
int dsm_ctx_sample_tau(dsm_ctx *ctx, int *nchange, double *logp_out);

/* A2: one draw of the auxiliary-count sums (HaploSNP_Sampler.py:284-309 via
 * :266,:276): sum_mu [S][G], esum [4][4] = [observed][true].                 */
int dsm_ctx_sample_stats(dsm_ctx *ctx, uint32_t iter, uint64_t *sum_mu, uint64_t *esum);
/* which counter-based specification dsm_ctx_sample_stats / dsm_ctx_gibbs_update follow for the resident shape:
 * 2 = aggregated sampler (oracle/stats_agg.c, spec 2; needs G <= 16, a subset table of at most 64 MB and every sample's
 * depth < 2^32), 3 = a variant of it (same law; the start of the inversion search from a table exp / log instead of
 * repeated squaring, item streams two Philox rounds off the cell's block instead of three -- selectable, not the default:
 * measured no faster), 4 = spec 2's samplers over tau WORDS: positions whose packed tau words are equal pool their counts in the
 * lowest such position and stage 1 draws once per (word, sample) -- the same law (a sum of multinomials with one probability vector),
 * restated by oracle/cbind.py: stats_agg(spec=4); G <= 8, not for chains sharded by positions (they run spec 2); by rule on large tables
 * (G <= 2 from half a million cells, G = 3 from a million, G = 4 ... 8 from 2.5 million cells with 64 x 2^G <= V) --,
 * 1 = per-read draws (oracle/desman_oracle.c: orc_stats_counter).  Where the aggregated pass applies
 * the cheaper of it and the per-read pass runs, by a cost model over the read total, the cells V x S (padded to the
 * kernel's lane groups) and the atomics per subset counter (kernels_stats.hip: stats_spec) -- a function of the shape and
 * the read totals only, the same on every run.  dsm_ctx_force_stats_spec: 0 = that rule, 1 = always spec 1, 2 / 3 / 4 = that
 * version of the aggregated sampler wherever it applies, small problems too (environment: DESMAN_HIP_STATS_SPEC=1|2|3|4 for
 * contexts without a choice of their own).  A batch (dsm_batch_gibbs_update) runs what its chains would run alone (one
 * specification per batch: chains that differ in it are refused); a chain sharded by positions runs spec 2, so it is the
 * unsharded chain under dsm_ctx_force_stats_spec(ctx, 2).  */
int dsm_ctx_stats_spec(dsm_ctx *ctx);
int dsm_ctx_force_stats_spec(dsm_ctx *ctx, int spec);
/* how often this process has measured where a subset table should start.  Always 0 since round 6: the table's row map puts the four cache lines of
 * a subset's row into four rows, and every place costs the same; the measurement of rounds 3-5 (eight timed stage-1 passes per
 * new table, measured tables pooled per process) only runs in the experiment build with DESMAN_HIP_NTAB_SWZ=0.   */
int dsm_debug_ntab_probes(void);
/* test hooks of the aggregated sampler: stage 1 only (subset counts ntab [S][2^G] u32 and esum), and nsamp variates of one
 * sampler (kind 0 binom_small, 1 binom_big: out [nsamp]; 2 mult4 with weights w[0..3]: out [nsamp][4]) of version
 * spec (2 / 3) exactly as oracle/stats_agg.c: orc_binom_test / orc_mult4_test draw them.                          */
int dsm_ctx_debug_stage1(dsm_ctx *ctx, uint32_t iter, uint32_t *ntab, uint64_t *esum);
int dsm_ctx_debug_binom(dsm_ctx *ctx, int kind, uint32_t n, const double *w4, uint64_t seed, int nsamp,
                        uint32_t *out, int spec);

/* A3+A4: gamma ~ Dir(alpha + sum_mu[s,:]) clamped/renormalised, eta[a,:] ~
 * Dir(delta + esum[:,a]) (HaploSNP_Sampler.py:263-281) from given sums.      */
int dsm_ctx_draw_gamma_eta(dsm_ctx *ctx, uint32_t iter, const uint64_t *sum_mu,
                           const uint64_t *esum, double *gamma_out, double *eta_out);

/* A5: logLikelihood / logPosterior of the resident state
 * (HaploSNP_Sampler.py:431-461).                                             */
int dsm_ctx_loglik(dsm_ctx *ctx, double *ll, double *lp);

/* A6: HaploSNP_Sampler.update (HaploSNP_Sampler.py:334-365): n_iter full Gibbs
 * iterations with MAP tracking and traces, all on the device.                */
int dsm_ctx_gibbs_update(dsm_ctx *ctx, int n_iter);
/* The same for n_ctx (1..8) chains of one shape -- same device, V, S, G, tau RNG; typically the replicate chains of one
 * G value (scripts/runDesman.sh:15-21) -- with one kernel launch per step of the iteration for all of them.  On tables
 * that leave most of the GPU idle the batch costs little more than one chain.  The mu/E pass of a batch is always the
 * aggregated sampler (dsm_ctx_stats_spec 2; needs G <= 16): every chain ends in the state dsm_ctx_gibbs_update leaves
 * it in after dsm_ctx_force_stats_spec(ctx, 2).                                                                        */
int dsm_batch_gibbs_update(dsm_ctx *const *ctxs, int n_ctx, int n_iter);
/* HaploSNP_Sampler.updateTau (HaploSNP_Sampler.py:383-407): tau-only sweeps
 * driven by host traces gamma_store [n][S][G], eta_store [n][4][4].          */
int dsm_ctx_update_tau(dsm_ctx *ctx, int n_iter, const double *gamma_store,
                       const double *eta_store);
/* ... and for n_ctx (1..8) chains of one shape at once (the -r path of replicate chains): one launch per sweep for all of
 * them; gamma_stores[k] / eta_stores[k] are chain k's traces.                                                          */
int dsm_batch_update_tau(dsm_ctx *const *ctxs, int n_ctx, int n_iter, const double *const *gamma_stores,
                         const double *const *eta_stores);

/* results of the last update call.  Any pointer may be NULL.                 */
int dsm_ctx_get_trace(dsm_ctx *ctx, double *ll, double *lp, int32_t *nchange,
                      double *gamma_store, double *eta_store);
int dsm_ctx_get_star(dsm_ctx *ctx, int64_t *tau_star, double *gamma_star,
                     double *eta_star, double *lp_star, int *iter_star);
/* sum over the stored iterations of the one-hot tau ([V][G][4] int64):
 * tauMean = tau_sum / n_iter (HaploSNP_Sampler.py:479-483).                  */
int dsm_ctx_get_tau_sum(dsm_ctx *ctx, int64_t *tau_sum);
/* tau as stored after iteration `it` of the last update ([V][G][4] int64).   */
int dsm_ctx_get_tau_at(dsm_ctx *ctx, int it, int64_t *tau);

/* A8-A12: Init_NMFT (desman/Init_NMFT.py).  F is derived from the resident
 * counts (:49-60).  tau [4V][G] row v + a*V, gamma [G][S] as in the reference.*/
int dsm_nmft_set(dsm_ctx *ctx, const double *tau, const double *gamma, int G);
int dsm_nmft_get(dsm_ctx *ctx, double *tau, double *gamma);
/* factorize (:98-115) incl. _adjustment; fix_gamma != 0 -> factorize_tau
 * (:134-149, no _adjustment).  div_trace (optional, max_iter+1) gets the
 * objective before the first and after every update; *n_done = updates run.  */
int dsm_nmft_factorize(dsm_ctx *ctx, int max_iter, double min_change, int fix_gamma,
                       int *n_done, double *div_trace);
/* The same for n_ctx (1..8) chains of one shape at once (replicate chains: one launch of each kernel of an update for all
 * of them; the stop test stays per chain).  Matrix-core path only (S <= 128, G <= 16; DSM_ERR_UNSUPPORTED otherwise).
 * n_done [n_ctx]; div_traces [n_ctx][max_iter + 1] or NULL.                                                          */
int dsm_batch_nmft_factorize(dsm_ctx *const *ctxs, int n_ctx, int max_iter, double min_change, int fix_gamma,
                             int *n_done, double *div_traces);
int dsm_nmft_objective(dsm_ctx *ctx, double *div);
/* get_tau (:230-245) -> one-hot [V][G][4] int64                              */
int dsm_nmft_get_tau(dsm_ctx *ctx, int64_t *tau_onehot);

/* f3 (upstream of the path): one inner step of Variant_Filter.get_filtered_VariantsLogRatio
 * (desman/Variant_Filter.py:348-356) for all V positions: BLL, the bounded-Brent minimiser of
 * mixNLL over p in (0, upperP) when `optimise` (scipy minimize_scalar(method='bounded')
 * semantics), and MLL at p.  Host pointers: ffreq [V][4] f64, maxA/maxB [V] int32, eta [4][4],
 * p_inout/MLL/BLL [V].                                                        */
int dsm_lrt_step(int device, const double *ffreq, const int32_t *maxA, const int32_t *maxB,
                 const double *eta, double upperP, int optimise, int V, double *p_inout,
                 double *MLL, double *BLL);

/* ------------------------------------------------------------------------ */
/* f4: accessory-gene assignment (desman/Eta_Sampler.py, desman/GeneAssign.py) */
/* C genes, gene c owns the rows gene_off[c]..gene_off[c+1]-1 of one           */
/* concatenated [Vtot][S][4] count tensor; eta[c][g] in {0..max_eta-1} is the  */
/* copy number of gene c in haplotype g.  All tau sweeps are the A1 sweep with */
/* the gene's masked, re-normalised gamma (Eta_Sampler.maskGamma :147-157).    */
/* ------------------------------------------------------------------------ */
typedef struct dsm_genes dsm_genes;
#define DSM_MAX_ETA 8

int dsm_genes_create(dsm_genes **out, int device);
int dsm_genes_destroy(dsm_genes *gs);
/* variants may be NULL when Vtot == 0 (GeneAssign without -v).  cov [C][S].   */
int dsm_genes_set_data(dsm_genes *gs, const int64_t *variants, int Vtot, int S, int C,
                       const int32_t *gene_off /*[C+1]*/, const double *cov);
/* gamma [S][G] (unmasked), epsilon [4][4], delta [G][S] (Eta_Sampler.__init__:47),
 * eta_log_prior [max_eta] (:139-145); cov_const[c] = -sum_s lgamma(cov+1) and
 * mult_const[c] = sum_{v,s} (lgamma(n+1) - sum_b lgamma(x_b+1)) are the data-only
 * constants of log_Poisson (:34-39) and log_multinomial_pdf (:27-32).         */
int dsm_genes_set_model(dsm_genes *gs, const double *gamma, const double *epsilon,
                        const double *delta, int G, int max_eta, const double *eta_log_prior,
                        const double *cov_const, const double *mult_const);
/* eta [C][G]; tau one-hot [Vtot][G][4] (either may be NULL = keep)            */
int dsm_genes_set_state(dsm_genes *gs, const int32_t *eta, const int64_t *tau);
int dsm_genes_get_state(dsm_genes *gs, int32_t *eta, int64_t *tau);
/* GSL-compatible MT19937 stream of the tau draws (shared semantics with dsm_setRNG) and the
 * Philox key of the batched sampler                                           */
int dsm_genes_seed(dsm_genes *gs, unsigned long mt_seed, uint64_t ctr_seed);
/* this object holds genes gene_base.. of a larger set sharded over several objects / ranks: the counter-based
 * draws are keyed by (global gene index, row within the gene), so results do not depend on the sharding   */
int dsm_genes_set_gene_base(dsm_genes *gs, int gene_base);
int dsm_genes_get_mt_state(dsm_genes *gs, uint32_t *state625);
int dsm_genes_set_mt_state(dsm_genes *gs, const uint32_t *state625);

/* Init_NMFT.factorize_tau per gene with the masked gamma (Eta_Sampler.__init__:128-134,
 * calcTauStar:419-426): tau_init [Vtot][4][G] f64 holds each gene's random start
 * (Init_NMFT.random_initialize_tau), genes without variants or with an all-zero mask row are
 * skipped; the arg-max (get_tau) becomes the resident tau.  n_iter [C] (optional).           */
int dsm_genes_nmft_tau(dsm_genes *gs, const int32_t *eta_mask /*[C][G] or NULL = resident*/,
                       const double *tau_init, int max_iter, double min_change, int32_t *n_iter);
/* Eta_Sampler.sampleTauC over all genes with variants and a non-empty mask, in gene order on
 * the GSL stream (:135, calcTauStar:437): nchange [C], logvar [C] = sum x log p after the
 * sweep, v_ll [Vtot] the same per variant (any may be NULL).  sweep = 0: evaluate only; 1: the GSL
 * stream; 2: counter-based draws (sharding-invariant).                                        */
int dsm_genes_sweep_all(dsm_genes *gs, const int32_t *eta_mask, int sweep, int32_t *nchange,
                        double *logvar, double *v_ll);
/* reference-order single step (Eta_Sampler.update:226-262): the two candidate sweeps
 * eta[c][g] = 0 / 1 of gene c (GSL stream: candidate 0 first, only if it keeps a haplotype),
 * logvar[2], swept[2]; then dsm_genes_step_choose commits the value drawn by the caller.      */
int dsm_genes_step_candidates(dsm_genes *gs, int c, int g, double *logvar, int *swept);
int dsm_genes_step_choose(dsm_genes *gs, int c, int g, int eta_value);
/* batched Gibbs: n_iter iterations of Eta_Sampler.update for all genes at once with
 * counter-based draws; eta_store [n_iter][C][G], gene_ll_trace [n_iter][C] (optional).
 * u_tau_ext [n_iter][G][2][Vtot*G] raw 32-bit words and u_eta_ext [n_iter][C][G] uniforms
 * replace the Philox draws when given (parity tests).  reset_star != 0: star := entry state
 * (update:216-219).                                                                          */
int dsm_genes_update(dsm_genes *gs, int n_iter, int reset_star, int32_t *eta_store,
                     double *gene_ll_trace, const uint32_t *u_tau_ext, const double *u_eta_ext);
/* Eta_Sampler.logLikelihood (:182-202) per gene for the resident state                        */
int dsm_genes_loglik(dsm_genes *gs, double *gene_ll);
int dsm_genes_get_star(dsm_genes *gs, int32_t *eta_star, double *gene_llstar);
int dsm_genes_set_star(dsm_genes *gs, const int32_t *eta_star, const double *gene_llstar);

/* GeneAssign.KLAssign.factorize (GeneAssign.py:85-120): eta [C][G] in/out (start values drawn by
 * the caller), cov [C][S], delta [S][G]; *n_done updates, *div final divergence.               */
int dsm_kl_assign(int device, const double *cov, const double *delta, double *eta, int C, int S,
                  int G, int max_iter, double min_change, int *n_done, double *div);

/* per-kernel HIP-event timing on the context's stream (bench/roofline).      */
#define DSM_K_STATS    0
#define DSM_K_DIRICH   1
#define DSM_K_TAU      2
#define DSM_K_FINAL    3
#define DSM_K_MT       4
#define DSM_K_NMFT_A   5
#define DSM_K_NMFT_G   6
#define DSM_K_NMFT_B   7
#define DSM_K_STATS2   8         /* stage 2 of the aggregated mu/E pass       */
#define DSM_K_STATSBIG 9         /* deferred stage-1 items (stats_big_kernel) */
#define DSM_K_STATSPAT 10        /* spec 4: word table + per-word count sums ahead of stage 1 (pat_rep_kernel, pat_agg_kernel) */
#define DSM_K_COUNT    11
/* evidence for the screening pass of the tau sweep: wavefront-steps run / left to the fp64 code, over the sweeps
   of dsm_ctx_gibbs_update and dsm_ctx_update_tau so far.  mode 1 = read and zero, 0 = read */
int dsm_ctx_sweep_stats(dsm_ctx *ctx, uint64_t *steps, uint64_t *exact_steps, int mode);
/* test hook: out[i] = the hardware log2 (v_log_f32) of in[i], the logarithm of the screening pass */
int dsm_ctx_debug_log2f(dsm_ctx *ctx, const float *in, float *out, size_t n);
/* Frees what the library keeps per process and device beyond the life of a context: the MT19937 jump tables of the parallel generator
   (2 x 50 MB per device, built by the first fill of >= 6 x 131 040 words: 2 x 2.2 ms) and the pool of placed subset tables (at most 32 of
   <= 512 KB).  For long-lived processes and for several rank processes sharing one GPU; nothing may be in flight.  Returns 0. */
int dsm_release_device_caches(void);
/* test hook: out[i] = a[i] / b[i] as the NMFT update divides (kernels_nmft.hip): kind 0 = fdiv_ext (any a, b >= 0: operands brought to
   within 2^+-512 of one by exact powers of two), 1 = fdiv_lo (a in (0, 1]), 2 = fdiv (operands far from the ends of the exponent range).
   Runs on the current device's default stream. */
int dsm_debug_fdiv(int kind, const double *a, const double *b, double *out, int n);
/* gamma / control step of an NMFT update (Init_NMFT.py:163-168 and the stop test :106): -1 = by size (default): one launch together
   with the reduction while the update kernel leaves <= 128 workgroup partials; above that at the start of the update kernel itself on
   the matrix-core path (S <= 128, G <= 16: an update is two launches), else a launch of its own (three); 0 = always a launch of its own;
   1 = always with the reduction; 3 = in the update kernel wherever that path runs.  The factors, update counts and objective traces do
   not depend on it (tests/test_gpu_fullsize.py asserts bit-equality). */
int dsm_ctx_set_nmft_fused(dsm_ctx *ctx, int mode);
/* dsm_nmft_factorize as ONE persistent launch (resident workgroups, in-kernel grid barriers, tau rows kept in LDS) where the
   table fits the machine (S <= 64, G <= 12, V <= 48 x compute units; factorize_tau also S <= 96 with V <= 16 x compute units): -1 / 1 = wherever it applies (default), 0 = never (the
   three-launch loop).  Same stopping rule, update counts, objective trace and factors, bit for bit: every path sums the workgroup
   partials in one canonical order (kernels_nmft.hip: nmft_sum_partials; tests/test_gpu_fullsize.py asserts the equality). */
int dsm_ctx_set_nmft_persist(dsm_ctx *ctx, int mode);
/* on = 0: every step of the tau sweep in fp64 (A/B switch).  Same law, and the same draws except in near-ties: after a screened
   step the current base's log-probability is evaluated afresh, an all-fp64 sweep re-uses the previous step's value (equal up to
   the last bits; a different draw needs the uniform within ~1e-13 of a CDF edge -- none in the ~1e7 draws of the parity tests) */
int dsm_ctx_set_tau_screen(dsm_ctx *ctx, int on);
/* the tau sweep of dsm_ctx_gibbs_update has a second instantiation with one more screen, for the steps of haplotypes that are rare in
   every sample (max_s gamma_sg <= 0.01: the spare haplotypes of a chain with more haplotypes than the table has strains -- their steps
   are near-ties that only the DIFFERENCES of the candidates can settle short of fp64).  mode -1 (default): a call runs it while the
   chain's abundances hold such a haplotype (as set by dsm_ctx_set_state / left by the previous call, and looked at every 64 iterations); 0 = never; 1 = always.  The draws are those of the fp64 code
   either way (tests/test_gpu_parity.py); not used for batches and sharded chains. */
int dsm_ctx_set_tau_neartie(dsm_ctx *ctx, int mode);
/* workgroups one tau sweep of the resident shape launches, and how many of them the device holds at once (occupancy of the
   kernel x compute units): launched / resident = rounds of the launch, the partly filled last one being its tail */
int dsm_ctx_tau_launch_info(dsm_ctx *ctx, int *launched, int *resident);
/* ---- one chain over several GPUs, sharded by positions (only worthwhile when there are fewer chains than
 * GPUs, V >~ 50k).  Each process / GPU holds a contiguous slice of the positions in its own context (dsm_ctx_set_counts with the
 * slice, dsm_ctx_set_state with the slice of tau and the shared gamma / eta, the same dsm_ctx_seed ctr_seed everywhere,
 * dsm_ctx_set_tau_rng(DSM_RNG_PHILOX)) and calls dsm_ctx_gibbs_update_sharded with its offset.  Once per iteration the library
 * hands `exchange` its subset table (uint32, to be SUMMED element-wise over the shards, in place) and an 18-double vector
 * (likewise); n_tab = 0 means the vector only.  The callback is the caller's all-reduce (RCCL: desman_amd/vshard.py); it is
 * called with the context's stream drained and must return 0 once the sums are in place, non-zero to abort.  Afterwards
 * gamma / eta / every trace are identical on all shards and equal the unsharded chain's (ll / lp to rounding), and each
 * shard's tau is the unsharded chain's slice: the counter-based streams are keyed by global indices.                       */
/* checkpoint / resume: besides (tau, gamma, eta) and the MT19937 state (dsm_ctx_get_state / dsm_ctx_get_mt_state)
 * a chain's position in its counter-based streams is the stream key and the number of iterations drawn so far.  A fresh context
 * given the same counts, state, MT19937 state, priors, tau RNG and these two words continues the chain bit for bit.          */
int dsm_ctx_get_counters(dsm_ctx *ctx, uint64_t *ctr_seed, uint32_t *iter_ctr);
int dsm_ctx_set_counters(dsm_ctx *ctx, uint64_t ctr_seed, uint32_t iter_ctr);
/* ... and the tau sweep's screening state: two words (one per launch parity) = sweeps still to run without the fp32 screening pass.
 * Which steps are screened cannot change a draw outside a ~1e-13 near-tie; carrying the words makes the resumed
 * chain take the same screening decisions as the uninterrupted one, so "bit for bit" holds without that caveat.                  */
int dsm_ctx_get_screen_state(dsm_ctx *ctx, uint32_t *out2);
int dsm_ctx_set_screen_state(dsm_ctx *ctx, const uint32_t *in2);
typedef int (*dsm_exchange_fn)(void *user, uint32_t *dev_tab, size_t n_tab, double *dev_vec, size_t n_vec);
int dsm_ctx_gibbs_update_sharded(dsm_ctx *ctx, int n_iter, int v_offset, int v_total, dsm_exchange_fn exchange, void *user);
/* ---- RCCL (xGMI) inside the library: one communicator per process / GPU, no torch.  librccl is dlopen()ed on first use
 * (DESMAN_HIP_RCCL names another copy).  Replaces the reference's "gather" -- `cat *\/fit.txt` over per-chain directories,
 * complete_example/README.md:626-627 after scripts/runDesman.sh:15-21 -- and carries the per-iteration exchange of a sharded
 * chain.  Bootstrap: rank 0 calls dsm_comm_unique_id and hands the 128 bytes to the other ranks by any
 * means (desman_amd/comm.py: a TCP socket on MASTER_ADDR); then EVERY rank calls dsm_comm_create (collective).              */
typedef struct dsm_comm dsm_comm;
int dsm_comm_unique_id(void *id128);
int dsm_comm_create(dsm_comm **out, const void *id128, int rank, int world, int device);
int dsm_comm_destroy(dsm_comm *comm);
int dsm_comm_rank(const dsm_comm *comm);
int dsm_comm_world(const dsm_comm *comm);
/* recv[world][n] <- every rank's send[n] (host buffers): the chain scheduler's one exchange, a fit record per chain */
int dsm_comm_allgather_f64(dsm_comm *comm, const double *send, double *recv, size_t n);
/* data[n] <- sum (op 0) or maximum (op 1) over the ranks, in place (host buffer); dsm_comm_barrier = the same with n = 0 */
int dsm_comm_allreduce_f64(dsm_comm *comm, double *data, size_t n, int op);
int dsm_comm_barrier(dsm_comm *comm);
/* dsm_ctx_gibbs_update_sharded with the exchange done by the library itself: the two all-reduces are ENQUEUED on the context's
 * stream between stage 1 and the Dirichlet launch (one grouped RCCL call), no host synchronisation, no callback.  The
 * communicator must live on the context's device; every rank of it must make the call (same n_iter, same v_total).          */
int dsm_ctx_gibbs_update_sharded_comm(dsm_ctx *ctx, int n_iter, int v_offset, int v_total, dsm_comm *comm);
/* plain device <-> host copies of the exchange buffers (host-side reductions, tests) */
int dsm_device_read(int device, const void *dev, void *host, size_t bytes);
int dsm_device_write(int device, void *dev, const void *host, size_t bytes);
int dsm_ctx_set_timing(dsm_ctx *ctx, int on);
int dsm_ctx_get_timing(dsm_ctx *ctx, double *ms_total /*[DSM_K_COUNT]*/,
                       int64_t *launches /*[DSM_K_COUNT]*/);
const char *dsm_kernel_name(int k);

#ifdef __cplusplus
}
#endif
#endif /* DESMAN_HIP_H */

/* include/hgibbs.h -- C ABI of the MI355X-native BayesRR single-site Gibbs hot path.
 *
 * The reference (medical-genomics-group/hydra) has no plugin/FFI boundary: the
 * sampler is one member function, BayesRRm::runMpiGibbs (src/BayesRRm.cpp:933).
 * This ABI cuts that function at the seam between data load (:1349) and output
 * (:2736): everything N-sized or M-sized lives on the GPU behind an opaque
 * handle, the host keeps only K/G-sized hyper-parameters and the RNG it shares
 * with the device.  Two layers:
 *
 *   hgibbs_*   device operators -- what a maintainer would call from
 *              runMpiGibbs in place of the CPU loops cited per entry point;
 *   hydra_*    the host driver itself (the body of runMpiGibbs restated on top
 *              of hgibbs_*), so that the CLI and language bindings share it.
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  Every call
 * returns 0 on success, non-zero on error (hgibbs_last_error() has the text;
 * like the reference's check_mpi/check_malloc, src/mpi_utils.hpp:19-36, errors
 * are fail-stop for the chain).  One caller thread per handle; calls on one
 * handle are never concurrent.  The library owns all device memory; host
 * buffers are only read/written during the call.  Arrays named *_host are host
 * pointers; nothing in this ABI takes a device pointer.
 */
#ifndef HGIBBS_H
#define HGIBBS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hgibbs_ctx* hgibbs_t;

/* MT19937 state shared between host and device (boost::mt19937 of
 * src/distributions_boost.hpp:27; host side drives std::shuffle and the
 * hyper-parameter draws, device side the per-marker draws). */
typedef struct {
    uint32_t x[624];
    uint32_t idx; /* next output position, 624 = twist first */
} hgibbs_rng_state;

const char* hgibbs_last_error(void);
int hgibbs_version(void);

/* ---- lifetime --------------------------------------------------------- */
/* One handle per GPU (one process per GPU in multi-GPU runs). */
int hgibbs_create(int device_id, hgibbs_t* out);
int hgibbs_destroy(hgibbs_t h);

/* ---- individuals sharded across GPUs (replaces the MPI_Allreduce family of
 * src/BayesRRm.cpp:2051,2456,2517-2518 -- SURVEY.md 2.2) ------------------ */
/* rank 0 fills id128 (ncclUniqueId, 128 bytes); the caller broadcasts it by
 * whatever means it has (torch.distributed, MPI, a file) and every rank calls
 * hgibbs_comm_init.  nranks == 1 needs neither call. */
int hgibbs_comm_unique_id(void* id128);
int hgibbs_comm_init(hgibbs_t h, int nranks, int rank, const void* id128);
/* Alternative to RCCL for the rare bulk reductions (load-time counts, per
 * iteration sums): the caller reduces a HOST buffer over its own transport
 * (MPI_Allreduce, gloo ...).  dtype 0 = f64, 1 = u64; returns 0 on success. */
typedef int (*hgibbs_allreduce_fn)(void* user, void* buf_host, size_t count, int dtype);
int hgibbs_comm_init_external(hgibbs_t h, int nranks, int rank, hgibbs_allreduce_fn fn, void* user);
/* In-launch exchange of the per-batch scalars over xGMI: every rank exports a
 * 64-byte IPC handle of its mailbox, the caller gathers the nranks handles
 * (rank order) and every rank imports them.  When imported, hgibbs_sweep sums
 * the ranks' (s1,s2) rows inside the sweep kernel -- each GPU pushes its rows
 * into all peers' mailboxes and adds the nranks contributions in rank order --
 * instead of splitting every batch around an ncclAllReduce launch. */
int hgibbs_p2p_export(hgibbs_t h, void* handle64);
int hgibbs_p2p_import(hgibbs_t h, const void* handles /* nranks * 64 bytes */);

/* ---- data: replaces Data::load_data_from_bed_file + sparse index build
 * (src/data.cpp:671-739, :1224-1290) -------------------------------------- */
/* bed_host: SNP-major packed columns WITHOUT the 3 magic bytes, M columns of
 * stride_in = ceil(n_total/4) bytes.  keep_host: n_total bytes, 0 drops the
 * individual (NA phenotype, src/data.cpp:1112-1158), NULL keeps all.  Of the
 * kept individuals this rank takes the half-open range [row_begin,row_end)
 * (row_begin % 4 == 0 unless keep_host is given).  n_global = number of kept
 * individuals over all ranks (the N of every formula). */
int hgibbs_load_bed(hgibbs_t h, const uint8_t* bed_host, uint64_t stride_in, uint32_t n_total, uint32_t M,
                    const uint8_t* keep_host, uint32_t row_begin, uint32_t row_end, uint32_t n_global);
/* Seeded synthetic genotypes generated directly in HBM (BASELINE.md section 4:
 * g ~ Binomial(2,p_j), p_j ~ U(0.01,0.5), missing calls at missing_rate).
 * Row i of the global matrix depends only on (seed, marker, i), so any
 * sharding yields the same matrix. */
int hgibbs_synth_bed(hgibbs_t h, uint32_t n_global, uint32_t M, uint32_t row_begin, uint32_t row_end,
                     uint64_t seed, double missing_rate);
/* Problem sizes as loaded: N over all ranks, this rank's individuals, markers,
 * first global row of this rank.  Any pointer may be NULL. */
int hgibbs_dims(hgibbs_t h, uint32_t* n_global, uint32_t* n_local, uint32_t* M, uint32_t* row_begin);
/* Copy packed columns [m0, m0+mcount) of this rank's shard back (tests). */
int hgibbs_get_bed(hgibbs_t h, uint32_t m0, uint32_t mcount, uint8_t* out_host, uint64_t out_stride);

/* a2: per-marker counts and mave/mstd (src/BayesRRm.cpp:1502-1508), counts
 * summed over ranks.  Any output pointer may be NULL. */
int hgibbs_marker_stats(hgibbs_t h, double* mave_host, double* mstd_host, uint64_t* n1_host, uint64_t* n2_host,
                        uint64_t* nmiss_host);

/* ---- residual: eps_host has this rank's row_end-row_begin entries -------- */
int hgibbs_set_residual(hgibbs_t h, const double* eps_host);
int hgibbs_get_residual(hgibbs_t h, double* eps_host);
/* sum and squared norm over ALL ranks (src/BayesRRm.cpp:1677-1678, :2685-2686) */
int hgibbs_reduce_eps(hgibbs_t h, double* sum, double* sqn);
/* eps_i += c (src/BayesRRm.cpp:1675, :1686) */
int hgibbs_add_scalar(hgibbs_t h, double c);
/* eps += x_marker * (-dbeta) ... i.e. the a8 update for one marker on its own
 * (src/BayesRRm.cpp:250-281): eps_i += {v0,v1,v2,0}[g_i] with
 * v0=-(mave*mstd*dbeta), v1=dbeta*(1-mave)*mstd, v2=dbeta*(2-mave)*mstd. */
int hgibbs_update_marker(hgibbs_t h, uint32_t marker, double dbeta);
/* a4 for one marker on its own: num = x_marker' eps (before + beta*(N-1)),
 * summed over ranks (src/BayesRRm.cpp:316-342). */
int hgibbs_dot_marker(hgibbs_t h, uint32_t marker, double* num);

/* ---- fixed-effect covariates (src/BayesRRm.cpp:2648-2681) ---------------- */
/* X_host: this rank's n_local x C covariate values, row-major (what data.X holds
 * for these individuals; hydra expects them standardised: x'x = N-1). */
int hgibbs_set_covariates(hgibbs_t h, const double* X_host, int C);
/* num_f = sum_k X(k,c) * (eps_k + gamma_old * X(k,c)) over all ranks (:2666-2668) */
int hgibbs_cov_dot(hgibbs_t h, int c, double gamma_old, double* num_f);
/* eps_k = eps_k + dgamma * X(k,c), dgamma = gamma_old - gamma_new (:2673-2676) */
int hgibbs_cov_update(hgibbs_t h, int c, double dgamma);

/* ---- model (src/BayesRRm.cpp:1037-1110) --------------------------------- */
/* groups_host[M] in [0,G); cVa/cVaI are G*K row-major with column 0 == 0. */
int hgibbs_set_model(hgibbs_t h, int G, int K, const int32_t* groups_host, const double* cVa_host,
                     const double* cVaI_host);

/* ---- marker effects (replicated on every rank) --------------------------- */
int hgibbs_set_beta(hgibbs_t h, const double* beta_host);
int hgibbs_set_components(hgibbs_t h, const int32_t* components_host); /* restart only */
int hgibbs_get_beta(hgibbs_t h, double* beta_host, int32_t* components_host, double* acum_host);
/* per-group sum of beta^2 in marker order (src/BayesRRm.cpp:2496-2499) */
int hgibbs_beta_sqnorm(hgibbs_t h, double* bsq_host /* G */);

/* ---- the sweep: src/BayesRRm.cpp:1709-2025 (+ :2468-2487) for all M markers
 * in the given order.  order_host[M]; sigmaG_host[G]; estPi_host[G*K];
 * adaV_host[M] (0 = marker frozen out, :1740,:1923-1926).  rng: in = state
 * before the first marker, out = state after the last.  cass_host[G*K] is
 * zeroed then counted (:1697,:1904).  nnz_updates = markers with
 * deltaBeta != 0 (fixes the algorithmic byte count of the sweep). */
int hgibbs_sweep(hgibbs_t h, const int32_t* order_host, double sigmaE, const double* sigmaG_host,
                 const double* estPi_host, const uint8_t* adaV_host, hgibbs_rng_state* rng, int32_t* cass_host,
                 uint64_t* nnz_updates);

/* Tuning knobs of the sweep (not part of the reference's behaviour; every setting gives the same chain up to
 * floating-point rounding, and bit-identical chains on the batch engine with gram = 0).  0 = automatic where noted.
 *   engine          0 auto (default): the resident engine where it applies, else the batch engine; 1 batch engine (one launch
 *                   per batch of markers up to an event); 2 resident engine (ONE launch per sweep; the call fails where it does
 *                   not apply: several ranks without peer mailboxes or more than eight, a
 *                   shard of more than 4096 individuals per compute unit -- 2048 with several ranks or refill = 1 --, G * K > 256).  Setting any option of the batch
 *                   engine below (a batch width, ...) while engine = 0 selects the batch engine.
 *   window          resident engine: columns kept in LDS behind the cursor, a power of two <= 256 (0 auto: 256)
 *   res_cus         resident engine: compute units to use (0 = all; one of them walks the chain, the others stream)
 *   pivots          resident engine: 1 = take the Gram terms of markers whose effect is non-zero at sweep start when a column is
 *                   streamed (their events need no round trip); default 0 (measured slower on MI355X)
 *   refill          resident engine, the streaming workgroups' form: 0 auto (default: 2, unless option pivots is on or the sweep starts
 *                   with a residual beyond the digits' range, |eps| >= 32), 1 = every wave whole columns, three vector instructions per
 *                   genotype (hg_resident.hip.h), 2 = every wave a slice of the individuals, the dots as integer matrix products over
 *                   eps's signed base-256 digits (hg_streamer2.hip.h; needs |eps| < 64: a sweep that meets a larger one fails, error 5)
 *   walker          resident engine: 0 auto (the second where it applies: every marker takes a uniform, <= 4 groups), 1 the
 *                   first walker, 2 the second (hg_walker2.hip.h; the call fails where it does not apply); walker2_ranks: 1 (default) =
 *                   several ranks run the second walker too (without early advances), 0 = they run the first
 *   announce        second walker: 1 (default) = an event that is certain (a marker whose effect is non-zero) is announced before
 *                   its draw, so that its Gram terms travel meanwhile
 *   window_end16    second walker: 1 (default) = the window ends at a multiple of sixteen positions (the second form of the streaming
 *                   workgroups then takes every group of sixteen columns once); 0 = at exactly `window` columns behind the cursor
 *   early_advance   second walker: a walk that runs out of dots moves the window on at once when at least this many positions have
 *                   passed (default 24; 0 = it waits)
 *   res_timeout_ms  resident engine: longest wait of any workgroup for another before the sweep is abandoned with an error
 *                   (default 2000); res_deadline_ms: the host's own deadline for the kernel (0 = derived from M, extended while the
 *                   walker's round counter moves)
 *   batch           speculative batch width, 1..256 (0 auto)
 *   cols_per_group  batch columns per workgroup: 2, 4 (default), 8, 16
 *   slices          most tile-group slices per column group, 1..64 (0 auto)
 *   gram            1 (default): continue past predicted events with Gram-corrected dots
 *   max_seg         segments (predicted events) per launch, 1..4 (0 auto by shard size)
 *   ext_limit       longest Gram-corrected extension in columns (default 256)
 *   gram_missing    -1 auto / 0 / 1: take columns with missing calls through the extension (four-term build)
 *   carry           hand the dots of columns behind an unplanned event to the next launch: -1 auto (default: shards of
 *                   400 000 individuals and more), 0 off, 1 on
 *   ahead           columns a launch streams ahead of its batch while its last workgroup draws, 0..256 (default 0: measured
 *                   slower on MI355X; needs carry)
 *   graph           1: replay the launches from a captured HIP graph
 *   p2p, force_split, chunk, debug_timing, w_kernel_timing   transport selection and diagnostics */
int hgibbs_set_option(hgibbs_t h, const char* name, int64_t value);
/* Statistics of the last sweep: launches, markers per launch, device time of
 * the sweep in ms (HIP events on the sweep's stream). */
typedef struct {
    uint64_t launches;        /* launches enqueued (the host enqueues in chunks: a few past the end of the sweep find nothing to do) */
    uint64_t nnz_updates;
    double device_ms;
    double kernel_ms_avg;     /* device_ms / working_launches: average period of the dominant kernel's working launches */
    uint64_t carried_columns; /* batch columns whose dot was handed on by the previous launch instead of being streamed again */
    uint64_t working_launches; /* launches that accepted markers or applied a pending update */
    uint64_t accepted_markers; /* == M after a complete sweep */
    uint64_t streamed_columns; /* batch columns whose dot product was streamed (speculative ones included, carried ones not) */
    uint32_t tiles_per_workgroup_min; /* tile groups of 4096 individuals one workgroup of the sweep kernel streamed per launch: */
    uint32_t tiles_per_workgroup_max; /* > 1 means the loop's next-tile prefetch and cross-tile accumulation ran */
    uint32_t engine;          /* 1 = batch engine (one launch per event batch), 2 = resident engine (one launch per sweep) */
    uint32_t walker;          /* resident engine: 1 = the first walker (the workgroup in lockstep), 2 = the second (one wave walks the chain, hg_walker2.hip.h) */
    double eps_sum_drift;     /* |sum(eps) at sweep end - sum(eps) at sweep start|: s2 of a column without missing calls is taken
                               * once per sweep (src/BayesRRm.cpp:331 re-sums per marker); this is what that assumption costs */
    /* resident engine: rounds of the walker (= working_launches), events (messages that carried an update), rounds that only
     * advanced the window, posterior chunks evaluated, chunks that had to wait for dots streamed behind the last message */
    uint64_t rounds, events, advances, chunks, refolds;
    uint64_t pivots;          /* resident engine: events whose Gram terms were under way before the draw (announced: option announce) or came with the columns (option pivots) */
    uint64_t predicted;       /* resident engine, the census of why rounds end: events at markers whose effect was non-zero at sweep start
                               * (certain to change); events - predicted came unannounced; advances = rounds that ran out of window */
    double shader_mhz;        /* resident engine: s_memtime ticks per microsecond over the sweep (the clock the walker's compute unit held) */
    uint64_t ticks[16];       /* resident engine with option debug_timing: 100 MHz ticks, walker [0] fold [1] collect [2] evaluate
                               * [3] scan + draw [4] message + results + prefetch; streaming workgroup 0: [8] wait [9] update [10] Gram [11] stream */
    uint32_t refill;          /* resident engine: the streaming workgroups' form -- 1 = fused multiply-adds per individual (hg_resident.hip.h),
                               * 2 = integer matrix products over signed base-256 digits of eps (hg_streamer2.hip.h; option refill) */
    uint32_t reserved_;
} hgibbs_sweep_stats;
int hgibbs_last_sweep_stats(hgibbs_t h, hgibbs_sweep_stats* out);
/* measured streaming ceiling of this GPU: device-to-device copy of `bytes` (choose well above the 256 MB
 * Infinity Cache), `reps` times; GB/s counts bytes read + bytes written */
int hgibbs_stream_ceiling(hgibbs_t h, uint64_t bytes, int reps, double* gbps);
/* diagnostic: the 48 accumulated stage-timestamp words (100 MHz ticks) of the sweep kernel's build with option
 * debug_timing = 1 since the last call (which clears them): bench.py's launch anatomy, tools/dbg_times.py */
int hgibbs_debug_times(hgibbs_t h, uint64_t* out48);
/* diagnostic: wall-clock stamps (100 MHz) of the resident engine's last 4096 messages, taken by the build with option
 * debug_timing = 1: 10 rows of 4096 words indexed by message number mod 4096 (`words` <= 10 x 4096 are copied) -- walker: [0] message
 * stored, [1] its Gram terms collected, [2] next message decided, [3] positions it consumed; streaming workgroup 0: [4] message seen
 * (an announced event: the message proper, not its announcement), [5] eps updated, [6] Gram terms sent, [7] refill streamed; rows 8, 9:
 * reserved for experiments (tools/res_anatomy.py reads the first eight) */
int hgibbs_resident_trace(hgibbs_t h, uint64_t* out, uint64_t words);

/* ======================================================================== */
/* Host driver: the body of BayesRRm::runMpiGibbs (src/BayesRRm.cpp:933-2939)
 * for --mpibayes bayesMPI, restated on top of hgibbs_*.                     */
/* ======================================================================== */
typedef struct hydra_chain* hydra_chain_t;

typedef struct {
    uint32_t seed;          /* --seed */
    int32_t shuffle;        /* --shuf-mark */
    int32_t G, K;           /* groups, mixture components incl. the zero one */
    const int32_t* groups;  /* M, or NULL for one group (src/BayesRRm.cpp:984-996) */
    const double* mS;       /* G*K, column 0 == 0.0 */
} hydra_model_desc;

/* y_host: phenotypes of the kept individuals of ALL ranks (n_global entries,
 * NA rows removed); the chain centres/scales it (src/BayesRRm.cpp:1564-1579). */
int hydra_chain_create(hgibbs_t dev, const hydra_model_desc* model, const double* y_host, hydra_chain_t* out);
int hydra_chain_destroy(hydra_chain_t c);
/* optional fixed effects (--covariates): X_host = n_global x C row-major, same
 * individuals and order as y_host; call once before the first iteration */
int hydra_chain_set_covariates(hydra_chain_t c, const double* X_host, int C);
int hydra_chain_gamma(hydra_chain_t c, double* gamma_out /* C */, int32_t* xI_out /* C, may be NULL */);
/* one Gibbs iteration: mu, shuffle, sweep, sigmaG/pi per group, covariates, sigmaE */
int hydra_chain_iterate(hydra_chain_t c);
/* hyper-parameters after the last iteration; any pointer may be NULL */
int hydra_chain_state(hydra_chain_t c, double* sigmaE, double* mu, double* sigmaG /*G*/, double* estPi /*G*K*/,
                      int32_t* m0 /*G*/, int32_t* cass /*G*K*/, hgibbs_rng_state* rng);
/* the .csv line of src/BayesRRm.cpp:2742-2760 for the given iteration number */
int hydra_chain_csv_line(hydra_chain_t c, uint32_t iteration, char* buf, size_t len);
const int32_t* hydra_chain_order(hydra_chain_t c);

/* ======================================================================== */
/* BayesW: Weibull survival model (src/BayesW.cpp); individuals shard as above  */
/* ======================================================================== */
/* The libc rand() stream the reference's ARS draws from (src/BayesW_arms.cpp:914-919,
 * srand at src/BayesW.cpp:1012, :877, :2029), one private copy per chain:
 * glibc's TYPE_3 additive-feedback generator. */
typedef struct {
    int32_t r[31];
    int32_t f, b;
} hgibbs_grand_state;
void hgibbs_grand_seed(hgibbs_grand_state* st, uint32_t seed); /* srand(seed) */
int32_t hgibbs_grand_next(hgibbs_grand_state* st);             /* rand()      */

/* arms(xinit, 4, &xl, &xr, logdens, data, &convex=1.0, 100, 0, ..., nsamp=1, ...) of
 * src/BayesW_arms.cpp as BayesW calls it: one draw from exp(logdens) on [xl, xr].
 * Returns 0 or the reference's error code (1003, 1004, 2000). */
int hgibbs_ars_sample(const double* xinit4, double xl, double xr, double (*logdens)(double, void*), void* data,
                      hgibbs_grand_state* rng, double* xsamp, int* neval);

typedef struct {
    uint64_t launches;    /* batch launches of the last sweep */
    uint64_t nnz_updates; /* markers whose effect changed */
    uint64_t ars_draws, ars_evals;
    double device_ms;       /* first launch to last, host round trips included */
    double sums_kernel_ms;  /* total of the k_bw_sums launches (option "w_kernel_timing"), else 0 */
} hgibbs_w_sweep_stats;

/* failure indicator (0/1) of the n_global kept individuals; allocates vi next to eps */
int hgibbs_w_init(hgibbs_t h, const int32_t* failure_host);
/* mean, standard deviation (not its inverse) and sum_failure per marker, src/BayesW.cpp:1201-1232 */
int hgibbs_w_marker_stats(hgibbs_t h, double* mave, double* sd, double* sum_failure);
/* groups[M] (NULL = one group), mS: G x K with column 0 == 0 (src/BayesW.cpp:760-790),
 * quad_points in {3,5,7,9,11,13,15,17,25} (:706-708) */
int hgibbs_w_set_model(hgibbs_t h, int G, int K, const int32_t* groups_host, const double* mS, int quad_points);
/* the N-length sums inside the scalar log densities, on the current residual:
 *  kind 0  sum_i exp(((eps_i + p0) - p1) * p2 - EuMasc)                mu_dens,    :77-88
 *  kind 1  sum_i exp(eps_i * p0 - EuMasc)                              alpha_dens, :132-142
 *  kind 2  sum_i exp(((eps_i + x_ic*p0) - x_ic*p1) * p2 - EuMasc)      gamma_dens, :118-129 (c = col)
 *  kind 3  sum_i eps_i * failure_i                                     alpha_dens */
int hgibbs_w_reduce(hgibbs_t h, int kind, int col, double p0, double p1, double p2, double* out);
/* vi_i = exp(alpha * eps_i - EuMasc), src/BayesW.cpp:1457-1459 */
int hgibbs_w_refresh_vi(hgibbs_t h, double alpha);
int hgibbs_w_get_vi(hgibbs_t h, double* vi_host /* n_local or NULL */, double* vi_sum);
/* the per-marker streaming operator on its own (src/BayesW.cpp:1499-1523): sums of vi over all
 * individuals / genotype 1 / genotype 2, with the marker's own effect taken out when beta_old != 0 */
int hgibbs_w_marker_sums(hgibbs_t h, uint32_t marker, double beta_old, double alpha, double* vi_sum, double* vi_1, double* vi_2);
/* one pass over all markers, src/BayesW.cpp:1461-1622.  rng: dist.rng (one uniform per marker);
 * ars_rng: the rand() stream; cass: G*K counts out; beta_sqnorm: G sums of squared effects out */
int hgibbs_w_sweep(hgibbs_t h, const int32_t* order_host, double alpha, const double* sigmaG, const double* pi, double sumSigmaG,
                   hgibbs_rng_state* rng, hgibbs_grand_state* ars_rng, int32_t* cass_host, double* beta_sqnorm, uint64_t* nnz_updates);
int hgibbs_w_last_sweep_stats(hgibbs_t h, hgibbs_w_sweep_stats* out);
/* diagnostic: the adaptive-rejection draw of an effect (src/BayesW.cpp:1562-1582, src/BayesW_arms.cpp:135-242) on ONE device lane,
 * `ndraws` times on the density beta_dens with the nine parameters dens9 = {alpha, sigmaG, sum_failure, sd, mean / sd, mixture value,
 * vi_0, vi_1, vi_2}, abscissae and bounds as the sweep sets them from beta_old and safe_limit: microseconds per draw (device clock),
 * density evaluations per draw, the last draw.  What the event's continuation on the device would cost. */
int hgibbs_w_ars_device_probe(hgibbs_t h, const double* dens9, double beta_old, double safe_limit, uint32_t seed, uint32_t ndraws,
                              double* us_per_draw, double* evals_per_draw, double* last_draw);
int hgibbs_w_get_beta(hgibbs_t h, double* beta, int32_t* components);
int hgibbs_w_set_beta(hgibbs_t h, const double* beta, const int32_t* components);

/* ---- BayesW chain driver: the body of BayesW::runMpiGibbs_bW (src/BayesW.cpp:905-2176) ---- */
typedef struct hydraw_chain* hydraw_chain_t;
typedef struct {
    uint32_t seed;         /* --seed: srand(seed) and dist.reset_rng(seed) (rank 0) */
    int32_t shuffle;       /* --shuf-mark */
    int32_t G, K;          /* groups; mixture components including the zero one */
    const int32_t* groups; /* M or NULL */
    const double* mS;      /* G*K, column 0 == 0 */
    int32_t quad_points;   /* --quad_points */
} hydraw_model_desc;
int hydraw_chain_create(hgibbs_t dev, const hydraw_model_desc* model, const double* y_host, const int32_t* failure_host, hydraw_chain_t* out);
int hydraw_chain_destroy(hydraw_chain_t c);
int hydraw_chain_set_covariates(hydraw_chain_t c, const double* X_host, int C);
int hydraw_chain_reseed_ars(hydraw_chain_t c, uint32_t seed);
int hydraw_chain_iterate(hydraw_chain_t c);
int hydraw_chain_state(hydraw_chain_t c, double* mu, double* alpha, double* sigmaG, double* pi, int32_t* m0, int32_t* cass,
                       hgibbs_rng_state* rng, hgibbs_grand_state* ars_rng);
int hydraw_chain_gamma(hydraw_chain_t c, double* gamma_out, int32_t* xI_out);
const int32_t* hydraw_chain_order(hydraw_chain_t c);
uint64_t hydraw_chain_last_nnz(hydraw_chain_t c);
int hydraw_chain_csv_line(hydraw_chain_t c, uint32_t iteration, char* buf, size_t len);
/* BayesW::init_from_restart (src/BayesW.cpp:869-903) + :1300-1311: the regular init, then the dumped
 * state; the ARS stream restarts at srand(ars_seed) (the reference passes opt.seed + iteration, :877) */
typedef struct {
    uint32_t iteration;
    double mu, alpha;
    const double* sigmaG;      /* G */
    const double* pi;          /* G*K */
    const double* beta;        /* M */
    const int32_t* components; /* M */
    const double* eps;         /* n_local: this rank's rows */
    const int32_t* order;      /* M */
    const double* gamma;       /* C or NULL */
    const int32_t* xI;         /* C or NULL */
    hgibbs_rng_state rng;
    uint32_t ars_seed;
} hydraw_restart_state;
int hydraw_chain_restore(hydraw_chain_t c, const hydraw_restart_state* st);

/* ---- checkpoint / restart (src/BayesRRm.cpp:842-928, :2802-2838) --------- */
/* State a --restart run reads back from the dump files; arrays are host
 * pointers, eps has this rank's individuals.  gamma/xI may be NULL without
 * covariates.  `iteration` = the saved iteration; the chain continues at +1. */
typedef struct {
    uint32_t iteration;
    double sigmaE, mu;
    const double* sigmaG;        /* G */
    const double* estPi;         /* G*K */
    const double* beta;          /* M */
    const int32_t* components;   /* M */
    const double* eps;           /* n_local */
    const int32_t* order;        /* M: markerI as dumped in .mrk */
    const double* gamma;         /* C or NULL */
    const int32_t* xI;           /* C or NULL */
    hgibbs_rng_state rng;
} hydra_restart_state;
int hydra_chain_restore(hydra_chain_t c, const hydra_restart_state* st);
/* dist.rng in Boost's stream form (624 decimal words, `file << rng`,
 * src/distributions_boost.cpp:38-44) and back (`file >> rng`, :46-55) */
int hydra_rng_to_boost_words(const hgibbs_rng_state* st, uint32_t* words624);
int hydra_rng_from_boost_words(const uint32_t* words624, hgibbs_rng_state* st);
/* markers with deltaBeta != 0 in the last sweep */
uint64_t hydra_chain_last_nnz(hydra_chain_t c);

#ifdef __cplusplus
}
#endif
#endif /* HGIBBS_H */

/*
 * smcnuts_hip.h -- C ABI of libsmcnuts_hip.so: the MI355X (gfx950) hot path of
 * SMC-NUTS behind the reference's own operator boundary.
 *
 * Each entry point names the reference interface it replaces
 * (paths relative to UoL-SignalProcessingGroup/SMC-NUTS @ 2024_10_08).
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; smcn_last_error()
 *     gives the message (owned by the library, valid until the next call on
 *     that context, or on a NULL context: until the next failing create);
 *   - the caller owns every host buffer; the library owns device memory;
 *   - host matrices are row-major [N, D] fp64 exactly as the reference's NumPy
 *     arrays; on the device the particle state is kept [D, N] ("dim-major")
 *     so that lane-adjacent particles are address-adjacent;
 *   - one context = one GPU = one shard of N particles; not re-entrant;
 *   - all calls are synchronous on return unless the name ends in _async.
 */
#ifndef SMCNUTS_HIP_H
#define SMCNUTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smcn_ctx smcn_ctx;

/* Device-native targets (replace smcnuts/model/bridgestan.py:7-146, whose
 * BridgeStan back end cannot be called from device code). */
#define SMCN_MODEL_GAUSS 0  /* data = [D, prior_sd, has_lik, lik_mean, lik_sd]            */
#define SMCN_MODEL_ARMA 1   /* data = [T, y_1..y_T]                 stan_models/arma/arma.stan   */
#define SMCN_MODEL_PRMWCD 2 /* data = [N, M, Clength, q, y.., Xkernel..]  stan_models/PRMwCD/PRMwCD.stan */
#define SMCN_MODEL_HOST 3   /* data = [D]: the density is the caller's (smcn_set_host_target)              */

#define SMCN_LKERNEL_FORWARD 0  /* smcnuts/lkernel/forward_lkernel.py:22-35   */
#define SMCN_LKERNEL_GAUSSIAN 1 /* smcnuts/lkernel/gaussian_lkernel.py:24-84  */

/* per-particle flag bits returned by smcn_get_tree_stats */
#define SMCN_FLAG_TAPE_OVERFLOW 1

int smcn_version(void);
const char* smcn_last_error(const smcn_ctx* ctx);

/* One context per GPU: replaces the state held by Samples.__init__ /
 * initialise_samples (smcnuts/samples/samples.py:8-88) and the target object.
 * `particle_base` is the global index of this shard's first particle (Philox
 * streams are keyed by global particle index, so a sharded run draws the same
 * numbers as a single-GPU run). */
int smcn_ctx_create(smcn_ctx** out, int device_id, int64_t n_particles, int64_t particle_base,
                    int model_id, const double* model_data, int64_t model_data_len);
void smcn_ctx_destroy(smcn_ctx* ctx);
int smcn_dim(const smcn_ctx* ctx);
int smcn_constrained_dim(const smcn_ctx* ctx);
/* 1 if this context's NUTS kernel can run several SMC iterations per launch (smcn_fuse_* /
 * smcn_block_* with B > 1), 0 if it runs one NUTSProposal.rvs (proposal/nuts.py:34-56) per launch. */
int smcn_fused_transitions(const smcn_ctx* ctx);
/* Use a caller-provided hipStream_t (e.g. the framework's current stream). */
int smcn_set_stream(smcn_ctx* ctx, void* hip_stream);
int smcn_synchronize(smcn_ctx* ctx);

/* Philox4x32-10 seed of the production RNG (replaces the shared sequential
 * np.random stream). */
int smcn_set_seed(smcn_ctx* ctx, uint64_t seed);

/* Particle state in/out (Samples.x / logw / r / x_new / r_new). NULL = skip. */
int smcn_set_state(smcn_ctx* ctx, const double* x, const double* logw);
int smcn_get_state(smcn_ctx* ctx, double* x, double* logw, double* wn);
int smcn_get_proposal(smcn_ctx* ctx, double* r, double* x_new, double* r_new, double* logw_new);
int smcn_set_momentum(smcn_ctx* ctx, const double* r);
/* A proposal computed elsewhere -- forward_kernel.rvs(x, r, phi) -> (x', r'), samples/samples.py:158 -- e.g. the
 * reference's recorded one: uploads r, x', r' and evaluates the density parts at x and x', leaving the state
 * smcn_propose_nuts leaves (the re-weight, L-kernels and tempering then run on it unchanged). */
int smcn_set_proposal(smcn_ctx* ctx, const double* r, const double* x_new, const double* r_new);

/* StanModel.logpdf / logpdfgrad / constrain (model/bridgestan.py:28-120) on
 * M rows of x; lpri/llik are the prior(+Jacobian) and likelihood parts with
 * log pi_phi = lpri + phi * llik.  Any output may be NULL. */
int smcn_target_eval(smcn_ctx* ctx, const double* x, int64_t M, double phi, double* logp,
                     double* grad, double* lpri, double* llik);
int smcn_target_constrain(smcn_ctx* ctx, const double* x, int64_t M, double* out);

/* Initial particles: x ~ N(0, I) from Philox stream 3 and
 * logw = log pi_phi(x) - N(x; 0, I)  (samples.py:77,85 with the harness'
 * sample_proposal, experiments/run_experiments.py:110). */
int smcn_init_particles_std_normal(smcn_ctx* ctx, double phi);
/* logw = log pi_phi(x) - logq0 for caller-supplied x (set_state) and q0. */
int smcn_init_weights(smcn_ctx* ctx, double phi, const double* logq0);

/* Samples.normalise_weights + calculate_ess (samples.py:91-113), split so
 * that shards can be combined: partials = [max, count_of_max, sum exp(.-max)
 * over non-max, sum exp(2(.-max)) over all finite-or-+inf].  apply() takes the
 * GLOBAL log-likelihood and writes wn = exp(logw - loglik). */
int smcn_normalise_partials(smcn_ctx* ctx, double out[4]);
int smcn_normalise_apply(smcn_ctx* ctx, double loglik);
/* Single-shard convenience: both steps; returns loglik and ESS. */
int smcn_normalise(smcn_ctx* ctx, double* loglik, double* ess);

/* Estimate.return_estimate (estimate/estimate.py:38-57,79-95) in constrained
 * space, two passes as the reference: sums[Dc] = sum_i wn_i * c(x_i), then
 * sums[Dc] = sum_i wn_i * (c(x_i) - mean)^2. */
int smcn_moment_sums(smcn_ctx* ctx, const double* mean_or_null, double* sums);

/* Samples._resample (samples.py:124-146): multinomial through
 * cdf = cumsum(wn)/sum, idx = searchsorted(cdf, u, 'right'), x <- x[idx],
 * logw <- loglik - log(n_total).  u = NULL draws Philox stream 2.
 * idx_out (int64[N], may be NULL) receives the ancestor indices. */
int smcn_resample_multinomial(smcn_ctx* ctx, const double* u, double loglik, double log_n_total,
                              int64_t iteration, int64_t* idx_out);
/* Resampling scheme for every resampling entry point of the ctx.  SMCN_RESAMPLE_MULTINOMIAL (default)
 * is the reference's rng.choice (samples.py:139): N independent uniforms, searchsorted right.
 * SMCN_RESAMPLE_SYSTEMATIC draws ONE uniform u0 per resampling and uses the keys (i + u0) / N on the
 * same prefix sum and search (lower-variance alternative; not in the reference). */
#define SMCN_RESAMPLE_MULTINOMIAL 0
#define SMCN_RESAMPLE_SYSTEMATIC 1
int smcn_set_resample_scheme(smcn_ctx* ctx, int scheme);
/* NUTSProposal.rvs (nuts.py:34-56) in the lane-per-particle kernel (arma): a launch lasts as long as its longest chain
 * of leaves, so when at most 16 (8, 4, 2, 1) lanes of a wavefront are still building trees, 4 (8, 16, 32, 64) lanes
 * share each one's T-step recurrence (segmented scan, csrc/smcn_nuts3.hpp recur_wide).  The sums are then re-associated: results agree
 * with the one-lane evaluation to rounding (~1e-15 relative on the density), not bit for bit, and WHICH evaluations
 * run wide depends on the launch's schedule (iterations per launch, wave mates).  1 (default) = on; 0 = every
 * evaluation by one lane (bit-identical results whatever the schedule: what the fused-vs-stepwise tests pin). */
int smcn_set_wide_eval(smcn_ctx* ctx, int on);
/* The lane kernel's schedule: at most `waves` wavefronts are launched (0 = default: one per SIMD, all the chip holds of this
 * kernel; < 0: no cap, one per 64 particles).  With fewer lanes than particles every wavefront owns a contiguous run of
 * particles (N / waves of them) and its 64 lanes work through it, so a population that is no multiple of 65 536 per GPU
 * does not pay a second round of wavefronts.  Which lane runs a particle never changes its draws (Philox is keyed by the
 * particle); with smcn_set_wide_eval(0) the results are bit-identical under every schedule.  (More than 4 096 particles
 * per wavefront: the launch falls back to one wavefront per 64 particles.) */
int smcn_set_lane_grid(smcn_ctx* ctx, int64_t waves);
/* ... and then a particle's block of B fused transitions is worked off in `segments` pieces (0 = auto: 4; 1 = a lane keeps
 * its particle for the whole block): a lane that starts the last transition of its segment looks for a ready job of its
 * wavefront -- another particle's next segment, the least advanced first --; if there is one it leaves its own particle's
 * (x', running log-weight) in a hand-over record at the segment's end and goes over to that job without an idle
 * iteration, else it simply goes on with its own particle.  The work is the same trees in the same order per particle;
 * only who builds them changes (bit-identical results with smcn_set_wide_eval(0)). */
int smcn_set_lane_segments(smcn_ctx* ctx, int segments);

/* Two-phase NUTS launches (group kernels whose trajectory edges live in registers: PRMwCD, Gaussians of 129..256 dimensions).
 * A launch of the group kernels lasts as long as its longest tree; with doublings > 0 a tree that still wants a doubling
 * after that many is parked at the boundary (its state: the two edges, the selected sample, a few scalars) and finished by
 * a second launch behind the first.  widen != 0: by the wavefront-per-particle functor of the model where one exists
 * (PRMwCD: 100 observations over 64 lanes; results then differ from the one-launch run by the rounding of the
 * re-associated likelihood sums), else by the kernel that parked it (bit for bit the one-launch result).
 * widen == 2 (tests, A/B): nothing is parked -- every tree runs from its start in the kernel that otherwise finishes the
 * parked ones, so that the parity tests reach the finisher's functor and tree stack on whole trees.
 * doublings <= 0: one launch (the default).  smcn_nuts_parked: how many trees the last proposal parked. */
int smcn_set_nuts_cap(smcn_ctx* ctx, int doublings, int widen);
int smcn_nuts_parked(smcn_ctx* ctx, int64_t* parked);
/* An INNER park level of the first launch (PRMwCD's lane-group kernel; 0 < doublings < the cap above, 0 = none): a tree
 * that wants more than `doublings` doublings is parked there as well and taken up again BY THE SAME LAUNCH once every fresh
 * particle has been handed out, so that the launch ends on pieces of trees (at most 2^(cap-1) leaves) instead of on whole
 * ones.  The same trees in another order: results are bit-identical with and without it. */
int smcn_set_nuts_requeue(smcn_ctx* ctx, int doublings);

/* Samples.propose_samples (samples.py:149-158) = momentum draw +
 * NUTSProposal.rvs (proposal/nuts.py:34-175) for every particle in ONE launch.
 * Momentum: Philox stream 1 unless smcn_set_momentum was called since the
 * last proposal.  Tape mode (tests): tape/tape_off as recorded from the
 * reference, tape_off has N+1 entries; NULL = Philox stream 0. */
int smcn_propose_nuts(smcn_ctx* ctx, double step_size, double phi, int max_depth, double delta_max,
                      int64_t iteration, const double* tape, const int64_t* tape_off);
/* per-particle leapfrog count, doublings, draws consumed, flags (int32[N]). */
int smcn_get_tree_stats(smcn_ctx* ctx, int32_t* nleap, int32_t* depth, int32_t* ndraws, int32_t* flags);
/* sum of leapfrogs of the last proposal (device-side reduction). */
int smcn_last_leapfrogs(smcn_ctx* ctx, int64_t* total);
/* density parts at x (before) and x_new (after) kept by the last proposal. */
int smcn_get_density_parts(smcn_ctx* ctx, double* lpri0, double* llik0, double* lpri1, double* llik1);

/* Samples._non_asympototic_reweight (samples.py:183-196):
 * logw_new = logw + pi_1(x_new) - pi_1(x) + L - q with the N(0,I) momentum
 * proposal.  FORWARD: L = N(-r_new; 0, I).  GAUSSIAN: L supplied per particle
 * through smcn_gauss_lkernel_logpdf beforehand. */
int smcn_reweight(smcn_ctx* ctx, int lkernel);
/* The asymptotic strategy (lkernel "asymptoticLKernel"):
 *  - smcn_accept_reject: the Metropolis step of NUTSProposalWithAccRej.rvs
 *    (smcnuts/proposal/nuts_acc_rej.py:42-49, proposal/utils.py:3-34) on the last
 *    proposal; u = NULL draws Philox stream 4; a rejected particle keeps (x, r);
 *  - smcn_reweight_asymptotic: Samples._asymptotic_reweight (samples.py:169-180),
 *    logw_new = logw + pi_{phi_new}(x) - pi_{phi_old}(x) at the OLD positions;
 *  - smcn_set_logw_density_ratio: logw = pi_{phi_num}(x) - pi_{phi_den}(x) at the resident
 *    x (estimate/estimate_from_tempered.py:47). */
int smcn_accept_reject(smcn_ctx* ctx, double phi, const double* u, int64_t iteration);
int smcn_reweight_asymptotic(smcn_ctx* ctx, double phi_old, double phi_new);
int smcn_set_logw_density_ratio(smcn_ctx* ctx, double phi_num, double phi_den);

/* Plug-in values for a duck-typed momentum proposal (lkernel.calculate_L /
 * forward_kernel.logpdf evaluated by the caller): per-particle L and/or q used
 * by the next smcn_reweight instead of the N(0, I) closed forms. */
int smcn_set_lkernel_values(smcn_ctx* ctx, const double* L, const double* q);

/* GaussianApproxLKernel.calculate_L (gaussian_lkernel.py:45-82), N-scaled
 * parts: sums[2D + 2D*2D] = [sum X, sum X X^T] of X = [-r_new, x_new] shifted
 * by `shift[2D]` (pass the previous mean, or zeros); then given the D x D
 * regression matrix B, offset m0[D], whitening U[D,D] and constant c0 the
 * per-particle log-density L_i = c0 - 0.5 |U^T(-r_i - m0 - B (x_i - mu_x))|^2. */
int smcn_gauss_lkernel_sums(smcn_ctx* ctx, const double* shift, double* sums);
int smcn_gauss_lkernel_logpdf(smcn_ctx* ctx, const double* mu_x, const double* m0, const double* B,
                              const double* U, double c0);
/* The same L-kernel for ONE shard with the D x D algebra on the device (two Cholesky factorisations by one wavefront; D <= 32):
 * both moment passes, the algebra and the conditional log-density are enqueued back to back, one wait at the end.
 * info = [status, c0, cond(c_xx), cond(cov)] (condition numbers: upper estimates).  status 0: the L values are set as
 * by smcn_gauss_lkernel_logpdf.  status 1 / 2: a covariance is not positive definite / too ill-conditioned for that route
 * (estimate >= 1e8) -- nothing is set, and the caller runs the reference's own pinv / eigh with their cut-offs and
 * exceptions through smcn_gauss_lkernel_sums + smcn_gauss_lkernel_logpdf (gaussian_lkernel.py:45-82). */
int smcn_gauss_lkernel_device(smcn_ctx* ctx, double info[4]);
/* The same over SEVERAL shards ("Gaussian L-kernel -- all-gather of 2D + (2D)^2 sums"): every rank runs
 *   stage 0 (this shard's un-shifted sums)  -> all-gather of nq = 2D + 2D (2D + 1) / 2 doubles, local -> gathered
 *   stage 1 (rows added in rank order, mean over n_total, this shard's centred sums)  -> the same all-gather
 *   stage 2 (rows added, the D x D algebra -- the same bits on every rank --, the log-density of this shard's particles;
 *            waits; info as above, status 1 / 2 handled as above)
 * on the two device buffers smcn_gauss_lkernel_buffers names (gathered = [world][nq], rank order).  world == 1: no
 * exchange, the stages read the local row.  On status 1 / 2 the L values the caller had set are left untouched. */
int smcn_gauss_lkernel_buffers(smcn_ctx* ctx, int world, void** local_dev, void** gathered_dev);
int smcn_gauss_lkernel_stage(smcn_ctx* ctx, int stage, int world, double n_total, double info[4] /* stage 2 */);

/* ESSTempering._ess (tempering/adaptive_tempering.py:41-56) partials at
 * new_phi for logw = new_phi*loglik + logpri - base, with base = pi_{phi_old}
 * at x_new; same 4 partials as smcn_normalise_partials. */
int smcn_temper_partials(smcn_ctx* ctx, double phi_old, double phi_new, double out[4]);
/* ESSTempering.calculate_phi (adaptive_tempering.py:18-63) with the bisection's own iteration on the device
 * (csrc/smcn_temper.hpp: scipy/optimize/Zeros/bisect.c restated; a pass evaluates the 15 trial points of the next four
 * bisection steps, a one-wavefront kernel takes the steps).  target = alpha N (the ESS the weights are tempered to);
 * the density parts at x_new must be resident (kept by the NUTS kernel / smcn_eval_proposed_parts).
 *   one shard:      smcn_temper_bisect -- one host synchronisation when ESS(1) >= target (phi stays 1), two when a
 *                   bisection runs, instead of one per trial point;
 *   several shards: per pass  smcn_temper_bisect_pass; all-gather of the [15][4] doubles of smcn_temper_bisect_buffers
 *                   (in the stream); smcn_temper_bisect_decide -- all asynchronous -- and smcn_temper_bisect_result
 *                   after pass 10 (status 1: enqueue further passes, up to 25).
 * status: 0 = *phi holds the result (1.0 if ESS(1) >= target); 2 = f(phi_old), f(1) of equal sign (scipy: ValueError);
 * 3 = no convergence in 100 steps (scipy: RuntimeError). */
int smcn_temper_bisect(smcn_ctx* ctx, double phi_old, double target, double* phi, int* status);
int smcn_temper_bisect_pass(smcn_ctx* ctx, int pass, double phi_old);
int smcn_temper_bisect_buffers(smcn_ctx* ctx, int world, void** local, void** gathered);
int smcn_temper_bisect_decide(smcn_ctx* ctx, int pass, int world, double target, double phi_old);
int smcn_temper_bisect_result(smcn_ctx* ctx, double* phi, int* status);

/* Density parts (lpri, llik) at the resident x (which = 0) or x_new (1), for
 * the first temperature of Samples.initialise_samples (samples.py:78-82). */
int smcn_eval_proposed_parts(smcn_ctx* ctx, int which);

/* Samples.update_samples (samples.py:215-222) + the acceptance statistic of
 * SMCSampler.update_sampler (smc_sampler.py:97): x <- x_new, logw <- logw_new;
 * n_moved = #particles with every coordinate changed. */
int smcn_commit(smcn_ctx* ctx, int64_t* n_moved);

/* ---- device-resident loop (forward L-kernel, fixed temperature) -------------
 * The order of SMCSampler.sample() (smc_sampler.py:109-149) with every scalar
 * (log-likelihood, ESS, the resample decision of samples.py:120, estimates,
 * leapfrog and acceptance counts) produced and consumed on the device: K
 * iterations are enqueued without a host round trip.
 *   smcn_fast_begin   allocates the per-iteration history (K+1 records of
 *                     6 + 2*Dc doubles: ll, ess, resampled, leapfrogs, moved,
 *                     phi, mean[Dc], var[Dc]) and, if save_history, x_saved /
 *                     logw_saved (smc_sampler.py:73-74) on the device.
 *   smcn_step_begin   shard partials [max, count, s1, s2, sum w c(x), sum w
 *                     (c(x)-shift)^2] of the current weights into the buffer
 *                     smcn_fast_buffers reports (4 + 2*Dc doubles).
 *   (several shards: all-gather `local_partials` of every rank into `gathered`,
 *    rank-major, on the same stream -- the ONLY exchange of an iteration)
 *   smcn_step_finish  combines the shards in rank order, then normalise,
 *                     estimate, ESS, conditional multinomial resampling (local
 *                     to the shard), momentum draw, NUTS, re-weight, commit,
 *                     history.  last != 0: only the closing normalise /
 *                     estimate / ESS of smc_sampler.py:143-149.
 *   smcn_fast_read    synchronises and downloads. */
int smcn_fast_begin(smcn_ctx* ctx, int64_t K, int save_history, int world);
int smcn_fast_buffers(smcn_ctx* ctx, void** local_partials, void** gathered, int* nq);
int smcn_set_resample_uniforms(smcn_ctx* ctx, const double* u);
int smcn_step_begin(smcn_ctx* ctx, int64_t k);
int smcn_step_finish(smcn_ctx* ctx, int64_t k, int world, int rank, double n_total, double step_size,
                     double phi, int max_depth, double delta_max, int lkernel, int last,
                     const double* tape, const int64_t* tape_off);
int smcn_fast_read(smcn_ctx* ctx, double* hist, double* x_saved, double* logw_saved);
/* The same for generations k_from .. K only, and -- beside the loop -- generations k_from .. k_to that the caller knows to be
 * final (their blocks have been waited for), on a stream of the library's own: the copies then run under the NUTS launch
 * that is already enqueued instead of behind the whole run (SMCSampler.sample(): smc_sampler.py:139-140 keeps x_saved /
 * logw_saved of every generation). */
int smcn_fast_read_from(smcn_ctx* ctx, double* hist, double* x_saved, double* logw_saved, int64_t k_from);
int smcn_history_download(smcn_ctx* ctx, int64_t k_from, int64_t k_to, double* x_saved, double* logw_saved);
/* ---- fused transitions: B SMC iterations per NUTS launch ---------------------
 * Between two resampling events a particle's next NUTS transition depends only
 * on its own sample, so B iterations of the loop (smc_sampler.py:109-140) can
 * run inside one launch, speculating that no generation in between falls below
 * the resampling threshold (samples.py:120); the speculation is checked on the
 * recorded weights and rolled back to the first generation that has to
 * resample, so results equal the one-iteration-per-launch schedule bit for bit.
 *   smcn_fuse_begin(Bmax, world)   after smcn_fast_begin
 *   per block of B <= Bmax iterations starting at k0:
 *     smcn_step_begin(k0); [exchange local_partials -> gathered]
 *     smcn_fuse_run(k0, B, ...);   [exchange the (B-1) x nq block of smcn_fuse_buffers]
 *     smcn_fuse_finish(k0, B, ..., &n_ok)    n_ok in 1..B iterations were valid;
 *                                            the resident state is generation k0 + n_ok. */
int smcn_fuse_begin(smcn_ctx* ctx, int Bmax, int world);
int smcn_fuse_buffers(smcn_ctx* ctx, void** local_partials, void** gathered, int* nq);
int smcn_fuse_run(smcn_ctx* ctx, int64_t k0, int B, int world, int rank, double n_total, double step_size,
                  double phi, int max_depth, double delta_max, int decided);
/* decided = 0: resampling decision + shard-local multinomial resampling on the device (one shard:
 * the reference's; several shards: each keeps its own mass, logw = log W_shard - log N_local).
 * Several shards, default: resampling is GLOBAL (Samples._resample, samples.py:124-146, over the whole
 * population -- the indices one shard of N_total particles would draw), so results do not depend on
 * the shard count (to rounding; BIT FOR BIT only with smcn_set_wide_eval(0): the default lets lane groups evaluate a
 * wavefront's stragglers, and which evaluations those are depends on the schedule, hence on the shard sizes).  Per block: smcn_step_begin(k0); exchange; smcn_fuse_decide(k0, .., &resample);
 * if resample: the routed global resampling (smcn_gres_*: tile totals, then keys and ancestor rows point to
 * point -- the population itself is never gathered); finally smcn_fuse_run(.., decided = 1). */
int smcn_fuse_decide(smcn_ctx* ctx, int64_t k0, int world, int rank, double n_total, double phi, int* resample);

/* Pipelined form of the fused block (same arithmetic as smcn_fuse_run/_finish; smc_sampler.py:109-140
 * for B iterations).  The statistics of ALL B generations are produced at the end of the block (one
 * combine launch; history rows come back through pinned memory behind an event), so the caller can
 * enqueue the next block -- speculating "no resampling" -- BEFORE waiting, and the device never idles
 * on the host.  Per block:
 *   [smcn_step_begin + exchange + smcn_fuse_decide (+ smcn_block_resample_local | global resampling)]
 *   smcn_block_launch(k0, B)          momentum draws, B NUTS transitions per particle
 *   smcn_block_post(k0, B, world)     generations k0+1..k0+B, re-weight, counts, shard partials [B][nq]
 *   exchange of smcn_fuse_buffers     (in stream: RCCL on the device pointers; or smcn_block_partials_get/_set)
 *   smcn_block_stats(k0, B, ..)       history rows k0+1..k0+B; read-back enqueued
 *   smcn_block_commit(k0, B) + smcn_block_launch(k0+B, B')      optional: the speculative next block
 *   smcn_block_wait(B, &n_ok, &resample_next)
 * If n_ok < B or resample_next: discard the speculative launch (smcn_synchronize), smcn_block_commit(k0,
 * n_ok) and restart from generation k0 + n_ok, which resamples. */
/* Targets evaluated by the CALLER: any model with the reference's StanModel interface
 * (smcnuts/model/bridgestan.py:28-120: log_density / log_density_gradient of an arbitrary BridgeStan
 * model) that has no device functor.  A context created with SMCN_MODEL_HOST asks this function for
 * the density wherever the device functors would be evaluated: x is [n][D] row-major; lpri/llik [n]
 * receive log prior (+ Jacobian) and log likelihood, so that log pi_phi = lpri + phi * llik; with
 * want_grad != 0 also gpri/glik [n][D].  Non-finite values are mapped to -inf as bridgestan.py:45-49
 * does.  Return 0, or non-zero to abort the calling entry point.  The NUTS proposal then runs the
 * tree building on the device and the target in lock step on the host (one call per leapfrog of the
 * longest tree): the generality path, not the fast one.  Device-resident loops (smcn_fast_*,
 * smcn_block_*) are not available for such a context. */
typedef int (*smcn_host_target_fn)(void* user, int64_t n, int D, const double* x, int want_grad, double* lpri,
                                   double* llik, double* gpri, double* glik);
int smcn_set_host_target(smcn_ctx* ctx, smcn_host_target_fn fn, void* user);
/* sum_i wn_i v_ic  or  sum_i wn_i (v_ic - shift_c)^2 over caller-supplied rows v [N][Dc] (estimate.py:79-95
 * with the caller's own constrain(), bridgestan.py:100-120); out has Dc entries. */
int smcn_moment_sums_of(smcn_ctx* ctx, const double* v, int Dc, const double* shift, double* out);

int smcn_block_resample_local(smcn_ctx* ctx, int64_t k0);
int smcn_block_launch(smcn_ctx* ctx, int64_t k0, int B, double step_size, double phi, int max_depth,
                      double delta_max);
int smcn_block_post(smcn_ctx* ctx, int64_t k0, int B, int world);
int smcn_block_partials_get(smcn_ctx* ctx, int B, double* out);
int smcn_block_partials_set(smcn_ctx* ctx, int B, int world, const double* gathered);
int smcn_block_stats(smcn_ctx* ctx, int64_t k0, int B, int world, int rank, double n_total, double phi,
                     int final_block /* the run ends with this block: the scalar history is downloaded behind it */);
int smcn_block_wait(smcn_ctx* ctx, int B, int* n_ok, int* resample_next);
int smcn_block_commit(smcn_ctx* ctx, int64_t k0, int n_ok);
/* ESS (samples.py:113) of the B generations of the block smcn_block_wait returned for. */
int smcn_block_ess(smcn_ctx* ctx, int B, double* out);
int smcn_fuse_finish(smcn_ctx* ctx, int64_t k0, int B, int world, int rank, double n_total, double phi,
                     int* n_ok);
#ifdef SMCN_LEGACY_ABI   /* round-1 generation of the block exchange: nothing in this repository calls it any more
                          * (smcn_block_partials_get / _set replaced it); exported only by builds that define the macro */
int smcn_fuse_partials_get(smcn_ctx* ctx, int B, double* out);
int smcn_fuse_partials_set(smcn_ctx* ctx, int B, int world, const double* in);
#endif

/* Host-side exchange of the shard partials, for communicators that cannot
 * all-gather device memory (e.g. gloo): read this shard's 4 + 2*Dc doubles /
 * write the world x (4 + 2*Dc) gathered block. */
int smcn_partials_get(smcn_ctx* ctx, double* out);
int smcn_partials_set_gathered(smcn_ctx* ctx, const double* in, int world);

/* NUTS kernel time measured with HIP events on the launch stream since the
 * last reset: out = [total ms, launches, 0, 0, 0, 0]; reset != 0 clears. */
int smcn_timers(smcn_ctx* ctx, double out[6], int reset);

/* Self test of the library's lean fp64 device math used inside the arma density and of its wavefront reduction:
 * out[0..n) = exp(x), out[n..2n) = log1p(|x|), out[2n..3n) = 1/x, out[3n..4n) = the sum of x over the element's
 * wavefront (64 consecutive elements; the butterfly of the NUTS kernels, last stages by v_permlane*_swap),
 * out[4n..5n) = the same butterfly through ds_bpermute (identical bits expected); out[5n..7n) = the wavefront sums of
 * x and x^2 by the fused two-value butterfly (wave_sum2), out[7n..11n) = those of x, x^2, |x|, 1 - x by the four-value
 * one (wave_sum4).  out holds 11 n doubles. */
int smcn_selftest_math(smcn_ctx* ctx, const double* x, int64_t n, double* out);
/* Measurement aid (bench.py, roofline.peak_measured; SURVEY.md 8(d) asks for nominal AND on-box denominators): the
 * device's streaming copy rate [GB/s] and its fp64 FMA rate [TFLOP/s] at one and at four wavefronts per SIMD. */
int smcn_measure_peaks(smcn_ctx* ctx, double out[3]);
/* Device memory the library holds beyond its contexts: a context's streams and buffers are not given back to the driver
 * when it is destroyed (creating a stream costs 2-9 ms; fresh buffers cost their first touch); they are pooled per device
 * and handed to the next context
 * (buffers: to a request of the same size, zeroed).  The cache holds at most `SMCN_DEVICE_CACHE_MB` megabytes per device
 * (environment, read once; default 3072; 0 = every buffer goes back to the driver at once) and is emptied when an
 * allocation fails.  smcn_device_cache_trim returns what is idle on `device` (-1: all devices) to the driver and
 * reports the bytes released; *idle_bytes (may be NULL) = what was idle before. */
int smcn_device_cache_trim(int device, int64_t* released_bytes, int64_t* idle_bytes);
/* Test hook for smcn_set_wide_eval: the four sums of the arma recurrence (sum err^2 and its three sensitivity sums) of
 * n rows x[n][4], by one lane (out[i][0..3]) and by a group of `lanes` (64, 32, 16, 8 or 4) lanes (out[i][4..7]). */
int smcn_selftest_wide(smcn_ctx* ctx, int lanes, const double* x, int64_t n, double* out);

/* ---- shards: one process per GPU; the reference has no counterpart (single thread) ----
 * In-library communicator: RCCL over xGMI (looked up at run time; no link-time dependency).  Rank 0 obtains the
 * 128-byte id and hands it to the other ranks by any means (smcnuts_amd.parallel.RcclComm: a file keyed by the
 * launch, one node); collectives run in the context's stream on device pointers. */
int smcn_comm_unique_id(char out[128]);
int smcn_comm_init(smcn_ctx* ctx, int rank, int world, const char id[128]);
int smcn_comm_destroy(smcn_ctx* ctx);
/* The communicator's own view: out = [ranks in it (ncclCommCount), this rank (ncclCommUserRank), RCCL version code];
 * -1 each when the context has no communicator.  bench.py prints it so that a multi-GPU line proves how many ranks RCCL saw. */
int smcn_comm_info(smcn_ctx* ctx, int out[3]);
int smcn_comm_allgather(smcn_ctx* ctx, const void* src_dev, void* dst_dev, int64_t n_doubles);
int smcn_comm_allgather_host(smcn_ctx* ctx, const double* src, int64_t n_doubles, double* dst /* [world][n] */);
int smcn_comm_alltoallv(smcn_ctx* ctx, const void* send_dev, const int64_t* send_counts, void* recv_dev,
                        const int64_t* recv_counts, int elem_doubles);
int smcn_buf_get(smcn_ctx* ctx, const void* dev, int64_t n_doubles, double* host);
int smcn_buf_set(smcn_ctx* ctx, void* dev, int64_t n_doubles, const double* host);
/* device -> device in the context's stream, NOT waited for: several shards of one process on one GPU exchange their
 * buffers with it (smcnuts_amd.parallel.InProcessComm -- the rehearsal of the shard protocol a one-GPU box allows). */
int smcn_buf_copy(smcn_ctx* ctx, void* dst_dev, const void* src_dev, int64_t n_doubles);

/* Samples._resample (samples/samples.py:124-146: rng.choice over the WHOLE population) across shards without
 * gathering the population: blocked scan per shard -> all-gather of the tile totals (N_local / 1024 doubles) ->
 * every rank plans its keys (Philox by global slot) and their owner ranks -> all-to-all of the keys -> the owners
 * search their tiles and gather the ancestor rows -> all-to-all of only those rows -> scatter.  The ancestors are
 * those one shard of N_total particles draws.  Shards of ANY size: where N_local is a multiple of the scan tile (1024) the
 * cdf is summed exactly as one shard of N_total sums it; otherwise the last tile of every shard is partial and an
 * ancestor can differ from the one-shard run only at keys within rounding of a cdf step. */
int smcn_gres_begin(smcn_ctx* ctx, int world, double* ttot_host /* [N_local/1024] or NULL */);
int smcn_gres_buffers(smcn_ctx* ctx, void** ttot_local, void** ttot_all, void** keys_send, void** keys_recv,
                      void** rows_send, void** rows_recv);
int smcn_gres_plan(smcn_ctx* ctx, int world, int rank, const double* ttot_all_host /* or NULL: already on the device */,
                   int64_t iteration, int32_t* dest_rank_host /* [N_local] */);
int smcn_gres_set_order(smcn_ctx* ctx, const int32_t* order /* [N_local]: slot of the k-th key in send order */);
int smcn_gres_reserve(smcn_ctx* ctx, int64_t n_requests_to_serve);
int smcn_gres_serve(smcn_ctx* ctx, int world, int rank, int64_t n_requests);
int smcn_gres_finish(smcn_ctx* ctx, int world, const double* loglik /* NULL: the device-resident loop's */);

/* Measurement aid: `reps` x the kernels of Samples._resample (samples/samples.py:124-146) on the resident
 * state, timed with HIP events on the context's stream (bench.py's roofline entry for resampling). */
int smcn_bench_resample(smcn_ctx* ctx, int reps, int64_t iteration, double* ms_total);

/* Diagnostic builds only (-DSMCN_PROFILE): in-kernel cycle sums per section of
 * the NUTS loop, summed over wavefronts (out[0..7]; out[8], out[9]: loop trips of all wavefronts
 * and of the longest one); zeros in a normal build. */
int smcn_debug_profile(smcn_ctx* ctx, uint64_t out[16], int reset);

#ifdef __cplusplus
}
#endif
#endif

/* gpbo.h — C ABI of the MI355X (gfx950) GP-posterior + acquisition engine.
 *
 * This is the drop-in boundary (seam B3).  The reference (bayes_opt 3.3.0) is pure
 * Python and has no FFI of its own: the arithmetic on its suggest() hot path is delegated to
 * scikit-learn / SciPy.  Each entry point below replaces the call named beside it, so that a
 * maintainer binds it with ctypes behind the reference's two Python seams —
 * the sklearn estimator duck type of `optimizer._gp` and `AcquisitionFunction` subclassing.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every function returns an int status: 0 = GPBO_OK, negative = error (see enum); the message is
 *     available from gpbo_last_error(ctx) (ctx may be NULL for creation errors).
 *   - host buffers are caller-owned, C-contiguous, borrowed for the duration of the call only.
 *   - device memory is owned by the context; one HIP stream per context; a context is not
 *     thread-safe (distinct contexts may be used from distinct threads).
 *   - all matrices crossing the ABI are row-major float64 (the reference is float64 throughout,
 *     bayes_opt/target_space.py:95-96); `precision` selects the on-device arithmetic.
 *   - determinism: fixed reduction trees; arg-best ties -> lowest index; NaN -> first NaN wins
 *     (numpy argmin semantics, bayes_opt/acquisition.py:313).
 *   - environment: the library reads exactly four variables, none of which changes a result —
 *       GPBO_KSTAR_GB          k* slab workspace budget in GB (default 4)
 *       GPBO_COMM_TIMEOUT_S    deadline of a wait for a collective (default 120)
 *       GPBO_GROUP_TIMEOUT_S   deadline of a device-group job (default 300; kept >= the collective deadline + 15 s)
 *       GPBO_GROUP_HOST_MERGE  1: a device group merges its shards' records on the host instead of over RCCL
 *     (tests/test_abi.py checks the sources against this list).  Kernel A/B switches, dispatch overrides, probes and
 *     fault injection exist only in the DEBUG BUILD (-DGPBO_DEBUG, libgpbo_dbg.so), together with the entry points at
 *     the end of this header.
 */
#ifndef GPBO_H
#define GPBO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPBO_ABI_VERSION 2

enum gpbo_status {
  GPBO_OK = 0,
  GPBO_ERR_INVALID = -1,      /* bad argument (shape, enum, NULL)                    -> ValueError   */
  GPBO_ERR_HIP = -2,          /* HIP runtime failure (no device, OOM, launch error)  -> RuntimeError */
  GPBO_ERR_NOT_PD = -3,       /* K + noise*I not positive definite (see `info`)      -> LinAlgError  */
  GPBO_ERR_STATE = -4,        /* call order (predict before fit, no candidates ...)  -> RuntimeError */
  GPBO_ERR_UNSUPPORTED = -5,  /* kernel/precision/size outside the HIP path          -> NotImplementedError */
  GPBO_ERR_COMM = -6,         /* RCCL failure / missed deadline: the communicator is gone -> RuntimeError */
  GPBO_ERR_PEER = -7          /* multi-GPU: another rank's LOCAL step failed; the exchange completed and the
                                 communicators are intact — this step has no result, the next one may  -> RuntimeError */
};

enum gpbo_kernel { GPBO_KERNEL_RBF = 0, GPBO_KERNEL_MATERN25 = 1 };
enum gpbo_acq { GPBO_ACQ_UCB = 0, GPBO_ACQ_EI = 1, GPBO_ACQ_POI = 2 };
enum gpbo_precision { GPBO_F64 = 0, GPBO_F32 = 1 };

#define GPBO_MAX_MODELS 8   /* slot 0 = target GP, slots 1.. = constraint GPs */
#define GPBO_MAX_DIM 64
#define GPBO_MAX_SEEDS 64
#define GPBO_LML_BATCH_MAX 8 /* theta values one gpbo_lml_batch call evaluates side by side */

typedef struct gpbo_ctx gpbo_ctx;

/* ---- lifecycle -------------------------------------------------------------------------- */
int gpbo_abi_version(void);
int gpbo_device_count(int* count);
int gpbo_create(int device, gpbo_ctx** out);
int gpbo_destroy(gpbo_ctx* ctx);
const char* gpbo_last_error(const gpbo_ctx* ctx);
int gpbo_synchronize(gpbo_ctx* ctx);
/* Device properties as a JSON string (name, CUs, clocks, memory) for bench/profile headers. */
int gpbo_device_info(gpbo_ctx* ctx, char* buf, int buflen);

/* ---- fit at fixed theta ----------------------------------------------------------------- */
/* Replaces the tail of GaussianProcessRegressor.fit (sklearn/gaussian_process/_gpr.py:346-364,
 * called from bayes_opt/acquisition.py:84 and bayes_opt/constraint.py:148,151):
 *   K = k(X/ls, X/ls) + noise*I   (Matern-2.5: kernels.py:1711-1738; RBF: kernels.py:1556-1565)
 *   L = cholesky(K, lower)        (_gpr.py:349)
 *   alpha = cho_solve(L, y_norm)  (_gpr.py:360-364)
 * and additionally forms W = L^-1, the operator the posterior kernel applies to k*.
 * X: (N,d) row-major; y_norm: (N,) already normalised by the caller (_gpr.py:272-277);
 * length_scale: n_ls == 1 (isotropic) or n_ls == d (anisotropic).
 * precision: GPBO_F64, or GPBO_F32 = the factorisation stays fp64 but gpbo_posterior rounds k* and W to fp32 and
 * runs the N^2-per-candidate contraction on v_mfma_f32_16x16x4_f32 (sigma accurate to ~1e-3 relative; mu stays fp64).
 * On GPBO_ERR_NOT_PD, *info = 1-based index of the first non-positive pivot (LAPACK potrf info). */
int gpbo_fit(gpbo_ctx* ctx, int slot, const double* X, const double* y_norm, int64_t N, int d,
             int kernel, const double* length_scale, int n_ls, double noise, int precision,
             int* info);

/* Append n_new observations to a fitted slot at UNCHANGED kernel, length scale and noise.
 * Replaces re-running the whole fixed-theta fit (_gpr.py:346-364) on X u x_new, which is what the reference's
 * maximize() loop does every iteration (bayes_opt/bayesian_optimization.py:377-388 -> acquisition.py:79-86) and what
 * ConstantLiar's dummy refits do (acquisition.py:1130-1143).  Per new row j, O(j^2) instead of O(j^3):
 *   k = k(X[:j], x_j);  l = W k;  lambda = sqrt(1 + noise - l.l);  L[j,:] = [l, lambda];
 *   W[j,:] = [-(l^T W) / lambda, 1 / lambda]
 * then alpha = W^T (W y_norm) for ALL targets.  y_norm: the n_total = N + n_new normalised targets (the
 * normalisation of every target changes when one is added, _gpr.py:272-277).  n_new = 0 re-solves alpha for new
 * targets only (same X; the dummy-target case).  When the rows do not fit the slot's 64-row padding, or n_new > 16,
 * the factorisation is redone from the device-resident scaled inputs instead (same result as gpbo_fit).
 * On GPBO_ERR_NOT_PD the slot is left unfitted and *info = 1-based index of the failing pivot. */
int gpbo_fit_append(gpbo_ctx* ctx, int slot, const double* x_new, int64_t n_new, int d,
                    const double* y_norm, int64_t n_total, int* info);

/* gpbo_fit in two halves, for the fits that one suggest() makes back to back — the target GP and the constraint GPs
 * (bayes_opt/acquisition.py:84-86: `gp.fit(...)`, then `constraint.fit(...)` -> constraint.py:148-151, one GP per
 * constraint).  gpbo_fit_begin enqueues the whole fit of `slot` on the slot's own stream and returns; gpbo_fit_wait
 * waits for it and resolves the positive-definiteness check (GPBO_ERR_NOT_PD, *info as gpbo_fit).  A factorisation at
 * these sizes is a chain of short dependent kernels that leaves most of the chip idle, so several of them in flight
 * overlap almost perfectly (N = 8192: two fits 26.6 ms one after the other).  Between the two calls the slot must not be
 * used; every slot's result is bitwise the one gpbo_fit gives.  X and y_norm are copied with asynchronous transfers on
 * the slot's stream: they must stay alive and unchanged until gpbo_fit_wait returns (pageable memory happens to be staged
 * before the call returns, pinned or registered memory is not). */
int gpbo_fit_begin(gpbo_ctx* ctx, int slot, const double* X, const double* y_norm, int64_t N, int d, int kernel,
                   const double* length_scale, int n_ls, double noise, int precision);
int gpbo_fit_wait(gpbo_ctx* ctx, int slot, int* info);

/* Log marginal likelihood and its gradient with respect to log(length_scale) at the given theta.
 * Replaces one L-BFGS-B evaluation of GaussianProcessRegressor.log_marginal_likelihood(theta,
 * eval_gradient=True) (_gpr.py:575-652; Matern/RBF gradients kernels.py:1764-1766, 1567-1582), the inner
 * loop of the theta search in fit (_gpr.py:296-338).  grad has n_ls entries.  A non-PD kernel matrix gives
 * *lml = -inf, zero gradient and *info = pivot index (status stays GPBO_OK), as sklearn does.
 * Overwrites the slot's fit state (the slot must be re-fitted with gpbo_fit before gpbo_posterior). */
int gpbo_lml(gpbo_ctx* ctx, int slot, const double* X, const double* y_norm, int64_t N, int d,
             int kernel, const double* length_scale, int n_ls, double noise, int eval_gradient,
             double* lml, double* grad, int* info);

/* n_theta evaluations of gpbo_lml on the SAME (X, y_norm) at different length scales (length_scales: n_theta x n_ls,
 * row-major), in scratch models of their own ("lanes") so that the latency-bound factorisations share the device.
 * This is what the theta search's independent L-BFGS-B runs (the initial theta + n_restarts_optimizer restarts,
 * _gpr.py:296-338) need when they advance together.  Results per theta as gpbo_lml (lml[i], grad[i * n_ls ...], info[i]);
 * every lane computes exactly what gpbo_lml computes, bit for bit.  Model slots and their fits are not touched.
 * X = y_norm = NULL re-uses the inputs the previous call uploaded (same N, d) — the rounds of one theta search.
 * Below N = 2048 all lanes run through ONE sequence of ~60 launches (lane = a grid dimension of every kernel); from there
 * on each lane gets its own stream.  From the second call with the same shape a sequence is replayed as a captured hipGraph. */
int gpbo_lml_batch(gpbo_ctx* ctx, int n_theta, const double* X, const double* y_norm, int64_t N, int d, int kernel,
                   const double* length_scales, int n_ls, double noise, int eval_gradient, double* lml, double* grad,
                   int* info);

/* Parity accessors (tests): copy device state back as (N,N) row-major / (N,) float64. */
int gpbo_get_K(gpbo_ctx* ctx, int slot, double* out);      /* kernel matrix incl. noise, full symmetric */
int gpbo_get_L(gpbo_ctx* ctx, int slot, double* out);      /* lower Cholesky factor, upper zeroed (gp.L_) */
int gpbo_get_Linv(gpbo_ctx* ctx, int slot, double* out);   /* W = L^-1, lower */
int gpbo_get_alpha(gpbo_ctx* ctx, int slot, double* out);  /* gp.alpha_ */

/* ---- candidates ------------------------------------------------------------------------- */
/* Upload the candidate matrix x_tries (M,d) row-major (bayes_opt/acquisition.py:311,
 * TargetSpace.random_sample target_space.py:565-603) and keep it resident in HBM. */
int gpbo_set_candidates(gpbo_ctx* ctx, const double* Xc, int64_t M, int d);

/* Throughput mode (NOT stream-compatible with the reference): fill the resident candidate matrix with
 * M x d uniforms in [lo[t], hi[t]) from a counter-based Philox4x32-10 generator on the device, instead of
 * TargetSpace.random_sample + upload (target_space.py:565-603).  Index parity with the reference needs the
 * host RandomState stream (gpbo_set_candidates); this entry point removes the host sampling and the H2D copy. */
int gpbo_generate_candidates(gpbo_ctx* ctx, int64_t M, int d, const double* lo, const double* hi, uint64_t seed);
/* Index-parity mode: the SAME matrix TargetSpace.random_sample would return for an all-float space — column t =
 * random_state.uniform(lo[t], hi[t], M) in key order (target_space.py:593-600, parameter.py:86-87) — generated on
 * the device from the caller's MT19937 state (key[624], *pos as in RandomState.get_state()) and left resident; key
 * and *pos come back advanced by the 2*M*d outputs consumed, so RandomState.set_state() continues the reference's
 * stream.  Removes the host sampling and the upload without changing a single candidate. */
int gpbo_generate_candidates_mt19937(gpbo_ctx* ctx, int64_t M, int d, const double* lo, const double* hi,
                                     uint32_t* key, int* pos);
/* The same for rows [row_begin, row_end) only (one shard of a candidate set drawn by several GPUs from ONE stream): this
 * context keeps (row_end - row_begin) x d candidates resident.  key / pos are read only; key_out / pos_out (may be NULL)
 * receive the state after the WHOLE (M, d) draw and are written only when row_end == M. */
int gpbo_generate_candidate_rows_mt19937(gpbo_ctx* ctx, int64_t M, int d, int64_t row_begin, int64_t row_end,
                                         const double* lo, const double* hi, const uint32_t* key, int pos,
                                         uint32_t* key_out, int* pos_out);
/* MT19937 jump-ahead on the host: key_out = the 624-word block n_blocks (>= 1) after key_in, computed as
 * (x^(624 (n_blocks - 1)) mod phi)(F) applied to the state — the same polynomial route by which
 * gpbo_generate_candidates_mt19937 starts its sub-streams on the device (csrc/mt_jump.hip).  Equivalent to drawing
 * 624 * n_blocks outputs from numpy.random.RandomState and reading get_state()[1], in O(log n_blocks) polynomial products. */
int gpbo_mt19937_jump_blocks(const uint32_t key_in[624], int64_t n_blocks, uint32_t key_out[624]);
/* Copy n rows (by index) of the resident candidate matrix back to the host (x_min and the seeds,
 * bayes_opt/acquisition.py:313-317); out is (n, d) row-major; out-of-range indices give NaN rows. */
int gpbo_get_candidate_rows(gpbo_ctx* ctx, const int64_t* idx, int n, double* out);

/* ---- spaces with Int / Categorical parameters (second half) -------------------------------------------
 * TargetSpace.random_sample (bayes_opt/target_space.py:593-600) draws parameter by parameter from ONE RandomState:
 * FloatParameter -> uniform (parameter.py:86-87), IntParameter / CategoricalParameter -> randint (:280-284, :360-377), whose
 * word consumption depends on the values drawn.  The matrix is therefore assembled group by group in key order, the float
 * groups on the device, the others on the host at the position the device hands back:
 *   gpbo_generate_candidate_columns_mt19937  columns [col0, col0 + ncols) = uniform(lo_t, hi_t, M) per column from (key, pos),
 *                                            which come back advanced past these 2 M ncols words;
 *   gpbo_set_candidate_columns               columns [col0, col0 + ncols) from a host array (M, ncols).
 * Every group of one matrix is written with the same (M, d_total).  gpbo_transform_candidates then applies
 * TargetSpace.kernel_transform (target_space.py:340-347) in place — kind 0: identity, 1: np.round (IntParameter,
 * parameter.py:308-320), 2: CategoricalParameter's one-hot (parameter.py:434-449, including its batch behaviour: a column
 * is set in ALL rows as soon as it is the argmax of any row) — and keeps the untransformed matrix aside:
 * gpbo_get_candidate_rows keeps returning the rows as drawn, gpbo_posterior sees the transformed ones.
 * Because of that batch behaviour a categorical group (kind 2) is a reduction over ALL M rows: the context must hold the whole
 * reference batch.  On a context that is one shard of a multi-device job (gpbo_comm_init with world_size > 1, a member of a
 * gpbo_group) a kind-2 group returns GPBO_ERR_UNSUPPORTED instead of transforming its rows differently from the reference
 * (sharded callers transform on the host, as GroupEngine does). */
int gpbo_generate_candidate_columns_mt19937(gpbo_ctx* ctx, int64_t M, int d_total, int col0, int ncols, const double* lo,
                                            const double* hi, uint32_t* key, int* pos);
int gpbo_set_candidate_columns(gpbo_ctx* ctx, const double* values, int64_t M, int d_total, int col0, int ncols);
int gpbo_transform_candidates(gpbo_ctx* ctx, int n_groups, const int* kind, const int* col0, const int* ncols);

/* ---- posterior -------------------------------------------------------------------------- */
/* Replaces GaussianProcessRegressor.predict(X, return_std=True) (_gpr.py:443-494; called from
 * bayes_opt/acquisition.py:205,216 and bayes_opt/constraint.py:200,213) for the resident
 * candidates: mu = y_std * (k* . alpha) + y_mean ; sd = sqrt(max(1 - |W k*|^2, 0) * y_std^2).
 * mu / sd may be NULL (results stay on the device for gpbo_acq_argbest). */
int gpbo_posterior(gpbo_ctx* ctx, int slot, double y_mean, double y_std, double* mu, double* sd);

/* Convenience: set_candidates + posterior for a host batch (the HipGPR.predict path). */
int gpbo_predict(gpbo_ctx* ctx, int slot, const double* Xc, int64_t M, int d, double y_mean,
                 double y_std, double* mu, double* sd);

/* sklearn warns "Predicted variances smaller than 0. Setting those variances to 0." when it clips a NEGATIVE
 * variance (_gpr.py:479-485).  The device clips inside its finalize kernels; this reports (and clears) whether any
 * gpbo_posterior / gpbo_predict / gpbo_predict_grad since the last call clipped one, so the host can warn on exactly
 * sklearn's condition.  Synchronises the context's stream. */
int gpbo_take_negative_variance_flag(gpbo_ctx* ctx, int* seen);

/* Replaces GaussianProcessRegressor.predict(X, return_cov=True) (_gpr.py:443-447, 458-469; reached from
 * BayesianOptimization.predict(..., return_cov=True), bayes_opt/bayesian_optimization.py:238) for a host batch:
 * cov (M,M) row-major = (k(X, X) - V^T V) * y_std^2 with V = L^-1 K*^T formed as two MFMA GEMMs on the device (no clipping,
 * as sklearn on this branch); mu (M,) optional.  M <= 16384. */
int gpbo_predict_cov(gpbo_ctx* ctx, int slot, const double* Xc, int64_t M, int d, double y_mean, double y_std,
                     double* mu, double* cov);

/* Posterior AND its gradient in the inputs for a small host batch (M <= 256): mu, sd (M,) as gpbo_predict, and
 * dmu, dsd (M,d) = d mu / d x, d sd / d x.  One evaluation gives the local search of the reference
 * (bayes_opt/acquisition.py:365-374: scipy L-BFGS-B, which forms its gradient from d + 1 predicts by finite differences)
 * value and gradient from ONE point: u = W^T (W k*), d sd^2 / d x = -2 y_std^2 u . dk* / dx.
 * A clipped (zero) variance has zero slope.  sklearn has no counterpart; parity is against finite differences of
 * gpbo_predict / the oracle. */
int gpbo_predict_grad(gpbo_ctx* ctx, int slot, const double* Xc, int64_t M, int d, double y_mean, double y_std,
                      double* mu, double* sd, double* dmu, double* dsd);

/* The local-search stage of a suggest() as ONE call.  Replaces AcquisitionFunction._smart_minimize for an all-continuous
 * space (bayes_opt/acquisition.py:322-420: one scipy L-BFGS-B run per seed, every value a GP predict, every gradient d + 1
 * of them): all `n_seeds` runs advance in lockstep, each round is one batched evaluation of
 *   f(x) = -base_acq(mu0, sd0) [* prod_j (Phi((ub_j-mu_j)/sd_j) - Phi((lb_j-mu_j)/sd_j))]   and its gradient
 * from the posteriors of slots 0..n_constraints and their input gradients (as gpbo_predict_grad), and the optimiser
 * arithmetic (a projected L-BFGS with L-BFGS-B's stopping rule as SciPy configures it: 10 corrections, projected gradient
 * 1e-5, relative reduction 1e7 eps, 20 line-search steps, max_iter <= 0 -> 15000 iterations) runs on the host in between.
 * One model of at most 256 (padded) observations: the runs are one launch, a workgroup each, evaluations and optimiser on the
 * device — the same optimiser arithmetic, the same evaluation arithmetic, the same results (bit for bit for UCB).
 * Not the reference's iterates: parity is statistical (acquisition value at the returned point).
 * y_mean / y_std: (1 + n_constraints,) the targets' normalisation per slot; seeds (n_seeds,d), clipped into the box;
 * box_lo < box_hi (d,).  Outputs per seed: x_out (n_seeds,d) inside the box, f_out, status_out (0: projected gradient
 * below tolerance, 1: relative reduction below tolerance / no further progress, 2: iteration limit or a non-finite start —
 * SciPy's success = False), n_rounds_out (optional) = batched evaluations issued by the lockstep path, or — one model of
 * NP <= 256, where the whole stage is ONE launch (a workgroup per run) — the longest run's evaluation count; n_iter_out /
 * n_eval_out (optional, per seed) = accepted steps / objective evaluations, SciPy's nit / nfev.  The one-launch form returns when
 * its longest run has stopped: the calling thread and the context's stream are held for that long (a run is bounded by
 * 4 * max_iter + 64 evaluations of 4-40 us each; with the default max_iter = 15 000 a pathological run is ~1 s) — callers that
 * need a tighter bound pass a smaller max_iter (status 2 marks the runs it cut).  Like gpbo_predict, the call uses the context's candidate
 * buffer for its trial points: the resident candidate set is gone afterwards (fetch x_min / the seeds with
 * gpbo_get_candidate_rows first, as the reference reads x_tries before it starts its local searches, acquisition.py:313-317). */
int gpbo_polish_seeds(gpbo_ctx* ctx, int acq, double acq_param, double y_max, int n_constraints, const double* lb,
                      const double* ub, const double* y_mean, const double* y_std, const double* seeds, int n_seeds, int d,
                      const double* box_lo, const double* box_hi, int max_iter, double* x_out, double* f_out, int* status_out,
                      int* n_rounds_out, int* n_iter_out, int* n_eval_out);

/* ---- acquisition + arg-best ------------------------------------------------------------- */
/* Replaces the _get_acq closure + base_acq + argmin/min/argsort[:k]
 * (bayes_opt/acquisition.py:198-217, 485, 660-661, 847-849, 312-317) and, when n_constraints > 0,
 * ConstraintModel.predict (bayes_opt/constraint.py:199-221) over the resident candidates, using
 * the posteriors left on the device by gpbo_posterior for slots 0..n_constraints:
 *   ys = -1 * base_acq(mu0, sd0) [* prod_j (Phi((ub_j-mu_j)/sd_j) - Phi((lb_j-mu_j)/sd_j))]
 * acq_param = kappa (UCB) or xi (EI/POI); y_max is ignored for UCB.
 * lb/ub: (n_constraints,), +-inf allowed (short-circuit to 0/1 as the reference does).
 * Outputs: best_idx/best_val = ys.argmin()/ys.min() (first NaN wins); seed_idx/seed_val (k_seeds,)
 * = argsort(ys)[:k] with ties -> lowest index and NaNs last; ys_out (M,) optional (NULL to skip).
 * index_offset is added to every returned index (candidate shards). */
int gpbo_acq_argbest(gpbo_ctx* ctx, int acq, double acq_param, double y_max, int n_constraints,
                     const double* lb, const double* ub, int k_seeds, int64_t index_offset,
                     int64_t* best_idx, double* best_val, int64_t* seed_idx, double* seed_val,
                     double* ys_out);

/* ---- timing (HIP events on the context stream) ------------------------------------------ */
/* Milliseconds spent in the last call's kernels: [0] fit total, [1] posterior main kernel,
 * [2] posterior finalize, [3] acquisition + arg-best, [4] kmat assembly, [5] cholesky, [6] trtri. */
int gpbo_last_timings(gpbo_ctx* ctx, float* ms, int n);
/* on = 0: the calls stop recording their event pairs (gpbo_last_timings then answers -1 everywhere); on = 1 (the state of a
 * new context): they record them.  A record is a marker packet on the stream — a step of BASELINE config 1 (0.12 ms: fit + posterior +
 * acquisition + arg-best) carries eight of them — so callers that never read the timings (accelerate(), bayes_opt's loop) switch
 * them off.  No result depends on it. */
int gpbo_set_timing(gpbo_ctx* ctx, int on);

/* ---- multi-GPU (RCCL over xGMI) --------------------------------------------------------- */
/* The candidate rows are independent (sklearn _gpr.py:443-494 is row-wise; bayes_opt/acquisition.py:312-317 needs only
 * the global argmin and the k best), so the candidate matrix is block-partitioned in index order, every GPU fits the
 * same GP redundantly (deterministic -> identical L) and evaluates its block, and ONE exchange — ncclAllGather of each
 * shard's 1 + k (value, global index) records, 16 bytes each, produced on the device — precedes an identical merge
 * on every rank: first NaN wins the arg-best, otherwise lexicographic (value, index); seeds = the k smallest.
 *
 * (a) one process per GPU.  rank 0 calls gpbo_comm_unique_id and ships the 128 bytes to its peers through any host
 * channel; every rank calls gpbo_comm_init; gpbo_comm_acq_argbest is gpbo_acq_argbest over the union of all shards
 * (index_offset = first global row of this rank's shard); outputs are identical on every rank. */
int gpbo_comm_unique_id(char id[128]);
int gpbo_comm_init(gpbo_ctx* ctx, const char id[128], int world_size, int rank);
int gpbo_comm_acq_argbest(gpbo_ctx* ctx, int acq, double acq_param, double y_max, int n_constraints,
                          const double* lb, const double* ub, int k_seeds, int64_t index_offset,
                          int64_t* best_idx, double* best_val, int64_t* seed_idx, double* seed_val,
                          double* ys_out /* this rank's shard, (M_local,) or NULL */);
/* Host records in, gathered host records out (n_records per rank, rank-major): the bare exchange, for callers that
 * select on the host. */
int gpbo_comm_allgather_best(gpbo_ctx* ctx, const double* vals, const int64_t* idxs, int n_records,
                             double* all_vals, int64_t* all_idxs);
/* Drain this rank's stream, then *value = max over ranks (ncclAllReduce): barrier + max-over-ranks timing. */
int gpbo_comm_allreduce_max(gpbo_ctx* ctx, double* value);
int gpbo_comm_destroy(gpbo_ctx* ctx);

/* (b) one process, G GPUs — what sits behind BayesianOptimization.suggest(), which is a single Python process
 * (bayes_opt/bayesian_optimization.py:323-333).  gpbo_group_create opens one context per entry of `devices`
 * (ncclCommInitAll) and one host thread per context; every gpbo_group_* call runs its per-device part on all devices
 * concurrently and returns when all are done.  Listing a device more than once gives VIRTUAL ranks (several shards on
 * one GPU, records merged on the host without RCCL): the single-GPU rehearsal of the sharded path.
 * gpbo_group_collective: "rccl-allgather" or "host-merge(virtual ranks)". */
typedef struct gpbo_group gpbo_group;
int gpbo_group_create(int n_dev, const int* devices, gpbo_group** out);
int gpbo_group_destroy(gpbo_group* grp);
int gpbo_group_size(const gpbo_group* grp);
gpbo_ctx* gpbo_group_ctx(gpbo_group* grp, int rank);          /* borrowed: parity accessors, timings, small predicts */
const char* gpbo_group_collective(const gpbo_group* grp);
const char* gpbo_group_last_error(const gpbo_group* grp);
int gpbo_group_synchronize(gpbo_group* grp);
/* gpbo_fit / gpbo_fit_append on every device (replicated model). */
int gpbo_group_fit(gpbo_group* grp, int slot, const double* X, const double* y_norm, int64_t N, int d, int kernel,
                   const double* length_scale, int n_ls, double noise, int precision, int* info);
int gpbo_group_fit_append(gpbo_group* grp, int slot, const double* x_new, int64_t n_new, int d, const double* y_norm,
                          int64_t n_total, int* info);
/* gpbo_lml_batch across the group: the theta search's 1 + n_restarts_optimizer L-BFGS-B runs are independent (sklearn
 * _gpr.py:296-338), so the lanes of a lockstep round are spread over the devices, lane i on device i mod G (n_theta <= 64).
 * X / y_norm non-NULL: made resident on every device first; NULL: the previous call's inputs.  Every lane is bitwise what
 * gpbo_lml returns on any one device, so the search ends at the single-device theta.  lane_device (optional, n_theta):
 * the device each lane ran on. */
int gpbo_group_lml_batch(gpbo_group* grp, int n_theta, const double* X, const double* y_norm, int64_t N, int d, int kernel,
                         const double* length_scales, int n_ls, double noise, int eval_gradient, double* lml, double* grad,
                         int* info, int* lane_device);
/* x_tries (M,d): device r keeps rows [r M / G, (r + 1) M / G) resident (gpbo_group_shard reports the range). */
int gpbo_group_set_candidates(gpbo_group* grp, const double* Xc, int64_t M, int d);
int gpbo_group_shard(const gpbo_group* grp, int rank, int64_t* row_begin, int64_t* row_end);
/* gpbo_generate_candidates_mt19937 across the group: every device generates ITS row block of the reference's candidate
 * matrix from the caller's MT19937 state by jump-ahead — no host sampling, no upload; key / pos come back advanced. */
int gpbo_group_generate_candidates_mt19937(gpbo_group* grp, int64_t M, int d, const double* lo, const double* hi,
                                           uint32_t* key, int* pos);
/* gpbo_posterior on every shard; mu / sd (M,) in global row order, or NULL to keep them on the devices. */
int gpbo_group_posterior(gpbo_group* grp, int slot, double y_mean, double y_std, double* mu, double* sd);
/* gpbo_acq_argbest over all M candidates: global indices, same tie/NaN rules; ys_out (M,) optional. */
int gpbo_group_acq_argbest(gpbo_group* grp, int acq, double acq_param, double y_max, int n_constraints,
                           const double* lb, const double* ub, int k_seeds, int64_t* best_idx, double* best_val,
                           int64_t* seed_idx, double* seed_val, double* ys_out);
int gpbo_group_get_candidate_rows(gpbo_group* grp, const int64_t* idx, int n, double* out);
/* Failure path.  Every gpbo_group_* job has a deadline (GPBO_GROUP_TIMEOUT_S, default 300 s) and every wait for a
 * collective has one (GPBO_COMM_TIMEOUT_S, default 120 s): a device that never comes back, or a peer that never enters
 * the all-gather, turns into GPBO_ERR_COMM (communicators aborted with ncclCommAbort, the group / communicator unusable
 * afterwards) instead of a hung suggest().  A rank whose LOCAL pass failed still enters the all-gather with a poisoned
 * record, so that every rank returns an error from the same step: the failing rank its own code, the others
 * GPBO_ERR_PEER (the exchange completed, the communicators are intact, the group stays usable).  Buffers passed to a
 * gpbo_group_* call that returned GPBO_ERR_COMM because a device missed its deadline must stay allocated until
 * gpbo_group_destroy: a worker released late may still read them (everything the workers WRITE besides the caller's
 * output arrays lives in state owned by the job itself). */

/* ---- calibration (bench.py's roofline: measured peak and sustained clock next to the datasheet numbers) ---- */
/* Sustained v_mfma_f64_16x16x4_f64 rate in TFLOP/s over `iters` dependent-chain-free MFMAs. */
int gpbo_mfma_f64_peak(gpbo_ctx* ctx, int iters, double* tflops);
/* The same MFMA stream with in-kernel clocks, one workgroup of 4 * waves_per_simd (<= 4) waves per compute unit.
 * out[4] = { TFLOP/s over the kernel's own span (first wave's start .. last wave's end, s_memrealtime), shader cycles per
 * MFMA per SIMD, sustained shader clock in MHz (s_memtime / s_memrealtime), event-bracketed kernel milliseconds }.
 * mode 0: accumulators where the compiler puts them (VGPRs); mode 1: AGPR accumulators (inline asm); mode 2: the
 * posterior GEMM's register pattern (2 x 4 tiles, six operand registers). */
int gpbo_mfma_f64_probe(gpbo_ctx* ctx, int iters, int waves_per_simd, int mode, double* out);
/* Streaming copy bandwidth in GB/s (read+write bytes) over a `bytes`-sized buffer. */
int gpbo_hbm_copy_peak(gpbo_ctx* ctx, int64_t bytes, double* gbps);

/* ==== DEBUG BUILD ONLY (-DGPBO_DEBUG: bayesianoptimization_amd/libgpbo_dbg.so) ==========================================
 * Self-test seams, single-kernel timers and micro-benchmarks the tests and scripts/ use.  The product library
 * (libgpbo.so) exports none of them, reads none of the A/B environment switches (GPBO_CHOL_*, GPBO_POST_*, GPBO_SMALL_MAX,
 * GPBO_SELECT_V2*, GPBO_MT_*, GPBO_F32_*, GPBO_GEMM128, GPBO_TRI64*, GPBO_LML_GRAPH, GPBO_POLISH_FUSED*) and contains no scratch-using kernel. */
#ifdef GPBO_DEBUG
/* The optimiser of gpbo_polish_seeds alone, over a host objective (self-test seam: no device, no context): `fg` is called
 * once per lockstep round with the trial points of the runs that are still alive — x (n_live,d) -> f (n_live), g (n_live,d) —
 * and returns 0 or an error code that ends the call.  Same outputs as gpbo_polish_seeds. */
typedef int (*gpbo_fg_callback)(const double* x, int n_live, int d, double* f, double* g, void* user);
int gpbo_debug_minimize_box(gpbo_fg_callback fg, void* user, const double* seeds, int n_seeds, int d, const double* box_lo,
                            const double* box_hi, int max_iter, double* x_out, double* f_out, int* status_out, int* n_rounds_out,
                            int* n_iter_out, int* n_eval_out);

/* The objective of gpbo_polish_seeds' one-launch path (polish_fused.hip; slot 0, no constraints) at each of n <= 64 points (n,d),
 * evaluated `repeat` >= 1 times inside one launch (repeat > 1: for timing an evaluation): out (n, 4 + 3 d) = [f, mu, sd, 0 | df/dx (d) |
 * dmu/dx (d) | dsd/dx (d)] — what gpbo_predict_grad's six kernels compute, from the one kernel (the tests require the same bits). */
int gpbo_debug_polish_eval(gpbo_ctx* ctx, int acq, double acq_param, double y_max, double y_mean, double y_std, const double* points,
                           int n, int d, int repeat, double* out);

/* Multi-GPU failure path, self-test seam (no device needed): a group of workers without contexts, and a job in which
 * rank `fail_rank` returns `fail_code` and rank `hang_rank` sleeps `hang_ms` (either may be -1). */
int gpbo_group_debug_create(int n_ranks, gpbo_group** out);
int gpbo_group_debug_run(gpbo_group* grp, int fail_rank, int fail_code, int hang_rank, int hang_ms);
/* Fault injection: the next gpbo_comm_acq_argbest on `ctx` behaves as if its local pass had failed (it still enters the
 * exchange, with a poisoned record). */
int gpbo_debug_fail_next_acq(gpbo_ctx* ctx);
/* The blocked Cholesky alone on an n x n matrix (n a multiple of 64, lower triangle read; replaces LAPACK dpotrf behind
 * sklearn _gpr.py:349): L_out = the factorised buffer (n x n row-major, lower triangle valid), dinv_out = the inverted
 * 64x64 diagonal blocks [n/64][64][64], stamps_out[16] = shader-clock stamps of the first diagonal workgroup's phases
 * , ms_out = best of `iters` device times, info_out = LAPACK-style pivot info.  variant must be 3 (the 128-column
 * schedule; 2, the round-2 schedule, is retired).  Any output pointer may be NULL. */
int gpbo_debug_cholesky(gpbo_ctx* ctx, const double* A, int64_t n, int variant, int iters, double* L_out, double* dinv_out,
                        long long* stamps_out, double* ms_out, int* info_out);
/* C = alpha * A(m,k) * op(B) + beta * C on the fit GEMM kernel; b_trans: B given as (n,k). */
int gpbo_debug_gemm(gpbo_ctx* ctx, int m, int n, int k, double alpha, const double* A,
                    const double* B, int b_trans, double beta, double* C);
/* The same kernel timed on device-resident random operands: out[2] = { ms per launch, TFLOP/s } (lower_only counts half
 * the flops); a_trans: A given as (k,m). */
int gpbo_debug_gemm_bench(gpbo_ctx* ctx, int m, int n, int k, int b_trans, int a_trans, int lower_only, int iters,
                          double* out);
/* The selection launches of gpbo_acq_argbest alone — ys.argmin(), argsort(ys)[:k] (bayes_opt/acquisition.py:313-317) — over
 * caller-supplied values ys (M,): idx_out / val_out (k,) = the k smallest keys in order (index -1 = fewer than k values),
 * first_nan_out = lowest index holding NaN or -1, ms_out = milliseconds per selection over `iters` repeats.  variant 1: k
 * block-reduction passes (what GPBO_SELECT_V2=0 switches gpbo_acq_argbest back to); variant 2: threshold + rank counting
 * (the default; GPBO_SELECT_V2_CAP bounds its LDS list, beyond which it falls back to the passes).  Same picks, bit for
 * bit.  Overwrites the context's acquisition values. */
int gpbo_debug_select(gpbo_ctx* ctx, const double* ys, int64_t M, int k, int variant, int iters, int64_t* idx_out,
                      double* val_out, int64_t* first_nan_out, float* ms_out);
/* Single-wave instruction latency / issue-cost probe: out[t] = shader cycles for 64 copies of pattern t (latency_probe.hip
 * lists the patterns; t = 0 is the empty bracket).  n <= 32. */
int gpbo_debug_latency_probe(gpbo_ctx* ctx, long long* out, int n);
/* fp64 VALU FMA throughput next to the matrix pipe, 4 waves/SIMD. cfg 0: 16 MFMA per iteration only, 1: 256
 * v_fma_f64 (scalar-operand) only, 2: 16 MFMA + 256 VALU, 3: 16 + 128, 4: 8 + 256.
 * out[3] = { kernel ms, MFMA TFLOP/s, VALU TFLOP/s }. */
int gpbo_hybrid_probe(gpbo_ctx* ctx, int iters, int cfg, double* out);
#endif /* GPBO_DEBUG */

#ifdef __cplusplus
}
#endif
#endif /* GPBO_H */

/*
 * scaml_gp.h — C ABI of libscaml_hip.so: the MI355X (gfx950) batched exact-GP inference
 * hot path of ScaML-GP.
 *
 * The reference (boschresearch/Scalable-Meta-Learning-with-Gaussian-Processes) has no
 * FFI: the path sits behind the Python call surface of scamlgp/model.py and, beneath it,
 * gpytorch's operator seam.  Each entry point below cites the reference site(s) whose
 * arithmetic it replaces (paths relative to the reference checkout).
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer into memory owned by the caller (e.g. PyTorch's
 *     caching allocator); the library allocates nothing persistent and keeps no state;
 *   - all matrices/stacks are dense, contiguous, row-major fp64:
 *       X      (T, N, D)      point stacks            theta (T, D+2) = [l_0..l_{D-1}, os, noise]
 *       y      (T, N)         standardised targets    L     (T, N, N) lower Cholesky factors
 *       alpha  (T, N)         (K + noise I)^-1 y
 *     theta holds CONSTRAINED values (lengthscales, outputscale, noise variance);
 *   - `n_points` (T) int32, may be NULL: per-task number of valid points n_t <= N (ragged
 *     tasks; rows/cols >= n_t are treated as an identity block and never written);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); work is only
 *     enqueued, never synchronised;
 *   - return value: 0 ok; <0 bad argument (SCAML_E_*); never throws.  Numerical status
 *     is per task in the caller-provided `info` (LAPACK convention: info[t] = k > 0 means
 *     the leading minor of order k is not positive definite after the last jitter try).
 */
#ifndef SCAML_GP_H
#define SCAML_GP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCAML_KIND_RBF 0      /* gpytorch RBFKernel:    exp(-r^2/2)                        */
#define SCAML_KIND_MATERN52 1 /* gpytorch MaternKernel(nu=2.5): (1+√5r+5r²/3)exp(-√5r)      */

#define SCAML_OK 0
#define SCAML_E_BADARG -1      /* NULL pointer / negative size                              */
#define SCAML_E_TOOLARGE -2    /* N or D beyond what the kernels support (see *_max_*)      */
#define SCAML_E_LAUNCH -3      /* hipLaunch / attribute call failed (see scaml_last_error)  */

/* flags for scaml_gp_fit_fused_f64 */
#define SCAML_FIT_STORE_L 1u     /* write L (T,N,N); without it only alpha + scalars (MLL-only mode) */
#define SCAML_FIT_ZERO_UPPER 2u  /* also write zeros to the strict upper triangle of L       */
#define SCAML_FIT_NO_RETRY 4u    /* single attempt: no in-kernel jitter escalation           */

/* Library / limits ------------------------------------------------------------------ */
int scaml_version(void);            /* 10000*major + 100*minor + patch */
const char* scaml_last_error(void); /* text of the last HIP error seen by this thread    */
int scaml_fit_max_n(void);          /* largest N of the register-resident fused fit       */
int scaml_fit_max_d(int N);         /* largest D for that N (LDS budget)                  */

/*
 * (3) Fused "task-posterior": kernel matrix + noise, jittered Cholesky, alpha, quad,
 * logdet and the marginal log-likelihood for a stack of T independent tasks.
 * Replaces, for the whole stack at once, the per-task chain the reference runs inside
 *   scamlgp/model.py:176-188  (for task in meta_data: SingleTaskGP + optimize_marginal_likelihood)
 *   scamlgp/utils.py:171-177  (ExactMarginalLogLikelihood -> MVN.log_prob -> inv_quad_logdet ->
 *                              psd_safe_cholesky -> torch.linalg.cholesky_ex / solve_triangular)
 * with the kernels of scamlgp/model.py:36-70 / :73-105 and the noise of :25-33.
 *   quad[t]   = y^T (K+noise I)^-1 y          logdet[t] = log det (K+noise I)
 *   mll[t]    = -(quad + logdet + n_t log 2pi) / (2 n_t)      (no prior terms)
 * Jitter: like linear_operator's psd_safe_cholesky the first attempt adds nothing; a task
 * whose factorisation hits a non-positive pivot is retried IN-KERNEL with 1e-8, 1e-7, 1e-6
 * added to its diagonal (only that task); jitter_used[t] reports the value that succeeded,
 * info[t] > 0 that all attempts failed.  `jitter_in` (T) may be NULL; if given it is added to
 * every attempt (caller-controlled extra jitter).
 * Linv_diag (T, ceil(N/16), 16, 16), optional: the inverses of the 16x16 diagonal blocks of L
 * (identity-padded past n_t); scaml_posterior_batched_f64 needs them.
 * Outputs L, alpha, quad, logdet, mll, jitter_used, Linv_diag may individually be NULL (not
 * written); info must not be NULL.
 */
int scaml_gp_fit_fused_f64(const double* X, const double* y, const double* theta,
                           const int32_t* n_points, const double* jitter_in,
                           int T, int N, int D, int kind,
                           double* L, double* alpha, double* quad, double* logdet, double* mll,
                           int32_t* info, double* jitter_used, double* Linv_diag, unsigned flags, void* stream);

/*
 * (3b) The same fit for scaml_fit_max_n() < N <= scaml_fit_blocked_max_n() points per task (the 512-point source
 * tasks of scamlgp/benchmarking/configurations/hartmann6_ablation_num_points_per_task.py:17-18, fitted in the
 * reference by the same scamlgp/model.py:176-188 / scamlgp/utils.py:171-177 chain), with no host synchronisation
 * and nothing but this library's kernels (and one memset of a few words) on the stream.  Two implementations behind
 * the one entry point, chosen by shape: while the stack leaves CUs idle (3 T <= #CUs; 2 T <= #CUs for N <= 320;
 * D <= 16) ONE launch in which up to eight workgroups share a task (csrc/gp_fit_coop.hip); otherwise a 2 x 2 block
 * factorisation enqueued as one sequence of launches -- fused fit of the first 256 points in place, L21 =
 * (L11^-1 K12)^T by column strips, Schur complement, fused factorisation of the Schur complement in place, alpha /
 * MLL assembly.  Same results to rounding (tests/test_blocked_fit_gpu.py runs every test through both; info[t] = -1:
 * the one-launch kernel gave a task up after a bounded wait, which no correct run does).  Arguments, outputs and the jitter
 * ladder (one value for the whole matrix of a failing task: 0, 1e-8, 1e-7, 1e-6) are those of
 * scaml_gp_fit_fused_f64; info[t] = k > 0 counts pivots over the full matrix.  Requirements: N a multiple of 16,
 * D <= scaml_fit_blocked_max_d(); L, alpha, info and Linv_diag must be given (the later launches read the earlier
 * ones' results from them), SCAML_FIT_STORE_L is implied.  `workspace` is device memory of at least
 * scaml_gp_fit_blocked_workspace_bytes(T, N) bytes, 16-byte aligned, owned by the caller and free again when the
 * stream has passed the call.  Returns SCAML_E_TOOLARGE for shapes it does not take.
 */
int scaml_fit_blocked_max_n(void);
int scaml_fit_blocked_max_d(void);
long long scaml_gp_fit_blocked_workspace_bytes(int T, int N);
int scaml_gp_fit_blocked_f64(const double* X, const double* y, const double* theta,
                             const int32_t* n_points, const double* jitter_in,
                             int T, int N, int D, int kind,
                             double* L, double* alpha, double* quad, double* logdet, double* mll,
                             int32_t* info, double* jitter_used, double* Linv_diag, unsigned flags,
                             void* workspace, long long workspace_bytes, void* stream);

/*
 * (1) Stand-alone kernel matrix K[t] = os_t k(X1_t / l_t, X2_t / l_t), (T, N1, N2).  X2 == NULL means
 * X2 = X1 (N2 must equal N1; with add_noise != 0 the noise variance is added to the diagonal: the
 * training matrix of scamlgp/model.py:36-70 + :25-33); x2_shared != 0 means X2 is one (N2, D) set
 * used for every task (the cross-kernel K_* of gp.posterior(x), scamlgp/model.py:128).  The fused
 * entry points never materialise K; this one exists for callers that want it.
 */
int scaml_kernel_matrix_f64(const double* X1, const double* X2, const double* theta, int T, int N1, int N2, int D,
                            int kind, int x2_shared, int add_noise, double* K, void* stream);

/*
 * (2) Batched jittered Cholesky of GIVEN symmetric matrices A (T, N, N) (lower triangle read):
 * the psd_safe_cholesky of linear_operator (reached from scamlgp/utils.py:171-177) for a whole
 * stack, same in-kernel jitter escalation, status and outputs as scaml_gp_fit_fused_f64.  With a
 * right-hand side y (T, N) it also returns alpha = A^-1 y, quad = y^T A^-1 y and logdet.  y, alpha,
 * quad, logdet, jitter_used, Linv_diag may be NULL.  N <= scaml_fit_max_n().
 */
int scaml_potrf_batched_f64(const double* A, const double* y, const int32_t* n_points, const double* jitter_in,
                            int T, int N, double* L, double* alpha, double* quad, double* logdet,
                            int32_t* info, double* jitter_used, double* Linv_diag, unsigned flags, void* stream);

/*
 * (5) Batched source-GP posteriors at M query points shared by all tasks.
 * Replaces the per-source loop `[gp.posterior(x) for gp in source_gps]` of
 *   scamlgp/model.py:128 (inside _compute_target_prior) and :281 (ScaMLGP.__init__ caches)
 * i.e. gpytorch's exact prediction mu~ = K_* alpha, Sigma~ = k(x,x) - V^T V with V = L^-1 K_*^T,
 * followed by botorch's Standardize.untransform_posterior (mu = m + s mu~, Sigma = s^2 Sigma~;
 * y_mean / y_std (T) may be NULL = 0 / 1).  No observation noise is added.
 *   Xq (M, D); L, alpha, Linv_diag as produced by scaml_gp_fit_fused_f64 with the same X, theta.
 *   mu (T, M), var (T, M) = diagonal of Sigma, V (T, N, M) = L^-1 K_*^T (un-scaled): each may be
 *   NULL.  N <= scaml_posterior_max_n().
 *   flags: SCAML_POST_XQ_PER_TASK — Xq is (T, M, D), one query set per task;
 *          SCAML_POST_MEAN_ONLY   — only mu = m + s K_* alpha (no solve; L, Linv_diag, var, V unused/NULL).
 */
#define SCAML_POST_XQ_PER_TASK 1u
#define SCAML_POST_MEAN_ONLY 2u
int scaml_posterior_max_n(void);
int scaml_posterior_batched_f64(const double* Xq, const double* X, const double* theta, const double* L,
                                const double* Linv_diag, const double* alpha, const double* y_mean,
                                const double* y_std, const int32_t* n_points, int T, int N, int M, int D, int kind,
                                double* mu, double* var, double* V, unsigned flags, void* stream);

/*
 * (5b) Covariance block of the same posteriors between the first Ma query points and all M:
 *   cov[t][a][c] = y_std[t]^2 (os k(xq_a, xq_c) - sum_i V[t][i][a] V[t][i][c]),  cov (T, Ma, M).
 * With the target's training points placed first in Xq this yields Sigma_nn and Sigma_nq of
 * scamlgp/model.py:287-289 / :131 without recomputing the train block per query.
 */
int scaml_posterior_cov_f64(const double* Xq, const double* theta, const double* V, const double* y_std,
                            int T, int N, int M, int Ma, int D, int kind, double* cov, unsigned flags, void* stream);

/*
 * (5c) The explicit inverse factor Linv = L^-1 (T, N, N; zero above the diagonal; identity rows past n_t) of a
 * fused fit / POTRF, and the source posteriors computed from it.  gpytorch keeps such a cache
 * (DefaultPredictionStrategy.covar_cache, the root of K^-1) for fast predictive variances; here it turns
 * V = L^-1 K_*^T from a substitution with a serial dependency into a triangular matrix product, which is what
 * the BO loop wants: the source GPs of scamlgp/model.py:128, :281 are fixed while every acquisition step
 * scores thousands of candidates.  scaml_posterior_linv_f64 has the semantics of
 * scaml_posterior_batched_f64 (same mu, var, V; SCAML_POST_XQ_PER_TASK allowed, SCAML_POST_MEAN_ONLY not).
 */
int scaml_linv_batched_f64(const double* L, const double* Linv_diag, const int32_t* n_points, int T, int N, double* Linv,
                           void* stream);
/* The same inverse factor with only the block rows at or below each 16-column strip's diagonal block written (the 16 x 16
 * diagonal blocks complete, zeros above the diagonal inside them): what scaml_posterior_linv*_f64 read.
 * Everything above is left untouched -- half of the matrix not written (67 of 134 MB at T = 256, N = 256). */
int scaml_linv_batched_lower_f64(const double* L, const double* Linv_diag, const int32_t* n_points, int T, int N, double* Linv,
                                 void* stream);
int scaml_posterior_linv_f64(const double* Xq, const double* X, const double* theta, const double* Linv, const double* alpha,
                             const double* y_mean, const double* y_std, const int32_t* n_points, int T, int N, int M, int D,
                             int kind, double* mu, double* var, double* V, unsigned flags, void* stream);

/*
 * Batched Cholesky solve Xout = (L L^T)^-1 B for R right-hand sides per task, B and Xout (T, N, R), with
 * the L and Linv_diag of a fused fit / POTRF (gpytorch's cholesky_solve behind prediction caches and
 * inv_quad for new right-hand sides).  Forward and backward substitution as blocked MFMA products.
 */
int scaml_cho_solve_batched_f64(const double* L, const double* Linv_diag, const double* B, const int32_t* n_points,
                                int T, int N, int R, double* Xout, void* stream);
/* The backward half alone: Xout = L^-T B (linear_operator's TriangularLinearOperator.solve with upper = True behind
 * the same caches).  With V = L^-1 K_* at hand, K^-1 K_* a = L^-T (V a) needs only this. */
int scaml_solve_lt_batched_f64(const double* L, const double* Linv_diag, const double* B, const int32_t* n_points,
                               int T, int N, int R, double* Xout, void* stream);

/*
 * (6) Weighted sum over the task axis: out[e] = sum_t c_t in[t][e], c_t = w_t (power 1) or w_t^2
 * (power 2), tasks with active[t] == 0 skipped (active may be NULL).  in (T, len), out (len).
 * Replaces scamlgp/model.py:129-135: mean = sum_i w_i mu_i (power 1), cov = sum_i w_i^2 Sigma_i
 * (power 2, PsdSumLinearOperator) over the significant tasks of model.py:368-372.
 */
int scaml_weighted_task_sum_f64(const double* in, const double* w, const uint8_t* active, int T, long long len,
                                int power, double* out, void* stream);

/*
 * (6') The ScaML-GP target prior of scamlgp/model.py:108-135 in one call, on the outputs of (5)/(5b):
 *   mu_s[q] = sum_t w_t mu[t][q]  (mu (T, M)),   cov_s[a][q] = sum_t w_t^2 cov[t][a][q]  (cov (T, Ma, M)),
 * over the tasks with active[t] != 0 (the pruning mask of model.py:368-372; NULL = all).  Either of mu / cov may
 * be NULL.  Two launches of the weighted-sum kernel.
 */
int scaml_weighted_prior_reduce_f64(const double* mu, const double* cov, const double* w, const uint8_t* active, int T, int M,
                                    int Ma, double* mu_s, double* cov_s, void* stream);

/*
 * (4) Analytic gradient of mll[t] (as defined for scaml_gp_fit_fused_f64, no prior terms) w.r.t. the
 * constrained hyper-parameters theta[t] = [l_0..l_{D-1}, os, noise].  Replaces the autograd pass of
 * botorch's fit_gpytorch_mll through kernel, Cholesky and solve (scamlgp/utils.py:175, 190).
 *   d mll / d theta_p = (1 / 2 n_t) sum_ij (alpha alpha^T - K^-1)_ij dK_ij / d theta_p.
 * Inputs are the X, theta and the L, Linv_diag, alpha produced by the fused fit.  `workspace` must
 * hold scaml_mll_backward_workspace_doubles(T, N, D) doubles (L^-1 per task, then per-tile partial
 * sums).  Result: partial sums (T, tiles, D+2), tiles = nb (nb+1)/2 with nb = ceil(N/16), written
 * to `partials_out` if given, else to the tail of the workspace; the caller adds them over the tile
 * axis and divides by 2 n_t (deterministic: no atomics).  Only the sum over the tile axis is defined: the
 * kernel works on 2 x 2 groups of tiles and leaves a group's total in the slot of its first tile.  D <= 76.
 */
long long scaml_mll_backward_workspace_doubles(int T, int N, int D);
int scaml_mll_backward_f64(const double* X, const double* theta, const double* L, const double* Linv_diag,
                           const double* alpha, const int32_t* n_points, int T, int N, int D, int kind,
                           double* workspace, double* partials_out, void* stream);

/*
 * (5c') The covariance block of (5b) fused into the posterior pass of (5c): with VA = V[:, :Ma] (T, N, Ma) -- the V of the
 * first Ma query points, produced by a call of scaml_posterior_linv_f64 on those points alone -- one pass over all M
 * query points gives mu, var AND cov (T, Ma, M) = s_t^2 (os k(xq_a, xq_c) - VA^T V); V (T, N, M) itself never reaches
 * memory (at BASELINE configs[4]: 145 MB not written and not re-read per acquisition-function evaluation).  Replaces the
 * same reference sites as (5) / (5b): scamlgp/model.py:128-134, 281-289.  Ma <= 96, Ma <= N, Ma <= M.
 */
int scaml_posterior_linv_cov_f64(const double* Xq, const double* X, const double* theta, const double* Linv, const double* alpha,
                                 const double* y_mean, const double* y_std, const int32_t* n_points, const double* VA, int T, int N,
                                 int M, int Ma, int D, int kind, double* mu, double* var, double* cov, unsigned flags, void* stream);

/*
 * (7) Target GP posterior on top of the weighted source prior (scamlgp/model.py:359-384 eval branch, then gpytorch's
 * exact prediction: botorch GPyTorchModel.posterior -> ExactGP.__call__ in eval mode).
 * The reference evaluates it per acquisition-function call with ~40 small torch ops; here it is four launches and no
 * host synchronisation:
 *   scaml_target_assemble_f64   Knn = cov_s[:, :n] / s^2 + os_t k_t(Xt, Xt) + noise I,  Knq = cov_s[:, n:] / s^2 + os_t k_t(Xt, Xq),
 *                               resid = y~ - (mean_s[:n] - m) / s,  mean_q = (mean_s[n:] - m) / s,  var_q = var_s[n:] / s^2 + os_t
 *   scaml_potrf_batched_f64     (T = 1) L L^T = Knn with psd_safe_cholesky's jitter ladder, alpha = Knn^-1 resid
 *   scaml_cho_solve_batched_f64 (T = 1, R = M) Z = Knn^-1 Knq
 *   scaml_target_finish_f64     mu[q] = m + s (mean_q + Knq[:, q] . alpha),  var[q] = s^2 (var_q - Knq[:, q] . Z[:, q] + noise_add);
 *                               `info` (1) int32 or NULL: the POTRF's status -- a factorisation that failed even with jitter
 *                               (psd_safe_cholesky would raise NotPSDError) turns every output into NaN on the device
 * Inputs: cov_s (n, n + M), mean_s (n + M), var_s (n + M) from (6') at cat(train_X, Xq); Xall (n + M, D) = cat(train_X, Xq);
 * theta (D + 2) of the TARGET kernel; train_targets (n) standardised with (m_all, s_all).  n >= 1.
 */
int scaml_target_assemble_f64(const double* cov_s, const double* mean_s, const double* var_s, const double* Xall,
                              const double* theta, const double* train_targets, double m_all, double s_all, int n, int M, int D,
                              int kind, double* Knn, double* resid, double* Knq, double* mean_q, double* var_q, void* stream);
int scaml_target_finish_f64(const double* Knq, const double* Z, const double* alpha, const double* mean_q, const double* var_q,
                            double m_all, double s_all, double noise_add, const int32_t* info, int n, int M, double* mu, double* var,
                            void* stream);

/*
 * (5d) Input gradients of the posterior -- what botorch's optimize_acqf differentiates through model.posterior for when it
 * maximises the acquisition function (the caller behind scamlgp/utils.py:215-224; SURVEY 3.3, HOT LOOP #4; the reference gets
 * them from torch autograd through kernel, solves and PsdSum per L-BFGS-B iteration).
 * scaml_posterior_linv_grad_f64: the (5c') pass with 16 output columns per query point x_q = [value, d/dx_0 .. d/dx_{D-1}, 0 ..]:
 *   mu  (T, Mq, 16): mu[t][q][0] = posterior mean of source t at x_q,   mu[t][q][1 + d]  = d mean / d x_d
 *   var (T, Mq, 16): var[t][q][0] = posterior variance,                 var[t][q][1 + d] = d variance / d x_d
 *   cov (T, Ma, Mq * 16): cov[t][a][16 q] = Cov(f(xa_a), f(x_q)),       cov[t][a][16 q + 1 + d] = d Cov / d x_d
 * (un-standardised with y_mean / y_std as in (5)), Xq (Mq, D) the query points, Xa (Ma, D) the Ma leading points (the target's
 * training inputs) and VA (T, N, Ma) their V as in (5c').  Ma = 0: no covariance block (VA, Xa, cov may be NULL).  D <= 15, Ma <= 96.
 * scaml_target_posterior_grad_f64: d mu* / d x (Mq, D) and d var* / d x (Mq, D) of the ScaML-GP TARGET posterior (original units;
 * scamlgp/model.py:359-384 eval branch + gpytorch's exact prediction) from the weighted task sums of those outputs
 * (cov_g (n, Mq * 16), mu_g / var_g (Mq * 16): (6') applied to cov / mu / var above), the target inputs Xt (n, D), the target
 * kernel theta (D + 2), and alpha (n), Z (n, Mq) = Knn^-1 Knq of the value path ((7): the POTRF's alpha, the Cholesky solve's result).
 * info (1) int32 or NULL: a failed target factorisation turns the outputs into NaN.
 */
int scaml_posterior_linv_grad_f64(const double* Xq, const double* Xa, const double* X, const double* theta, const double* Linv,
                                  const double* alpha, const double* y_mean, const double* y_std, const int32_t* n_points,
                                  const double* VA, int T, int N, int Mq, int Ma, int D, int kind, double* mu, double* var, double* cov,
                                  unsigned flags, void* stream);
int scaml_target_posterior_grad_f64(const double* cov_g, const double* mu_g, const double* var_g, const double* Xt, const double* Xq,
                                    const double* theta, const double* alpha, const double* Z, double s_all, const int32_t* info, int n,
                                    int Mq, int D, int kind, double* dmu, double* dvar, void* stream);

/*
 * (8) The target GP's training objective with its analytic gradient, and the whole refit, in ONE launch.
 * Replaces what the reference runs on every report(): scamlgp/optimizer.py:176-185 rebuilds ScaMLGP and calls
 * optimize_marginal_likelihood (scamlgp/utils.py:139-212), which drives scipy L-BFGS-B through torch autograd over
 *   mll(z) = [ log N(y~ | mean, cov + noise I) + sum log priors ] / n                       (scamlgp/model.py:360-363, 376-383)
 *   mean = (source_means w - m_all) / s_all,   cov = source_covs w^2 / s_all^2 + os k_t(X, X; l)
 * with the source terms cached at construction (scamlgp/model.py:279-289) and the priors / constraints of
 * scamlgp/model.py:25-33, 73-105, 318-338.
 *   z (B, P), P = D + 2 + T: B independent parameter vectors [raw lengthscales (D), raw outputscale, raw noise, weights (T)] --
 *       the warm start and the prior-sampled restarts of utils.py:184-199 side by side; raw = unconstrained values of the
 *       sigmoid Interval constraints, weights as they are (GreaterThan(1e-10, transform=None): a box bound for the optimiser).
 *   means_t (T, n): source posterior means at the target inputs (source_means transposed), original units.
 *   covs_packed (T, n (n + 1) / 2): source posterior covariances, lower triangle packed row-wise -- element (a, b), a >= b, at
 *       a (a + 1) / 2 + b (source_covs[a, b, t]).
 *   X (n, D), y (n): target inputs and observations standardised with (m_all, s_all) (scamlgp/model.py:264-276, 309-316).
 *   spec_host: HOST pointer to 19 doubles (read during the call; the one exception to "every pointer is a device pointer"):
 *       [ls_lo, ls_hi, os_lo, os_hi, noise_lo, noise_hi,  then (kind, p1, p2) for the lengthscale, outputscale, noise and
 *       weight priors -- kind 0 none, 1 Gamma(concentration p1, rate p2), 2 LogNormal(loc p1, scale p2) --,  w_lower].
 * scaml_target_mll_f64: value (B) = mll(z_b), grad (B, P) = d mll / d z_b, info (B) (k > 0: pivot k not positive even with
 *   jitter 1e-6 -- psd_safe_cholesky's ladder runs in-kernel --, value = NaN then), jitter_used (B, may be NULL).
 * scaml_target_fit_f64: maximises mll from every row of z (in: start points, out: optima) with an L-BFGS (history pairs,
 *   projected backtracking line search on w >= w_lower, scipy L-BFGS-B's stopping rules: projected gradient <= gtol, relative
 *   decrease <= ftol, max_iter iterations), entirely on the device; value (B) = mll at the returned point, stats (B, 4)
 *   (may be NULL) = [iterations, evaluations, status, 0], status 1 / 2 converged (gradient / decrease), 0 max_iter,
 *   3 line search failed, 4 objective not finite at the start point.  workspace: scaml_target_fit_workspace_doubles(...) doubles.
 * One workgroup per row of z, the n x n matrix in LDS: n <= scaml_target_fit_max_n(T, D) (128 at T = 32, D = 6),
 * D <= scaml_target_fit_max_d(); SCAML_E_TOOLARGE otherwise.
 */
int scaml_target_fit_max_n(int T, int D);
int scaml_target_fit_max_d(void);
long long scaml_target_fit_workspace_doubles(int B, int T, int D, int history);
int scaml_target_mll_f64(const double* means_t, const double* covs_packed, const double* X, const double* y, double m_all, double s_all,
                         const double* spec_host, const double* z, int B, int n, int T, int D, int kind, double* value, double* grad,
                         int32_t* info, double* jitter_used, void* stream);
int scaml_target_fit_f64(const double* means_t, const double* covs_packed, const double* X, const double* y, double m_all, double s_all,
                         const double* spec_host, double* z, int B, int n, int T, int D, int kind, int max_iter, int history, double gtol,
                         double ftol, double* value, int32_t* info, double* jitter_used, int32_t* stats, double* workspace,
                         long long workspace_doubles, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCAML_GP_H */

// Kernel-argument blocks of the posterior kernels, shared by device source and host launcher.
#pragma once
#include <stdint.h>

namespace scaml {

struct PosteriorParams {
  const double* Xq;         // (M, D) query points, shared by all tasks
  const double* X;          // (T, N, D)
  const double* theta;      // (T, D+2)
  const double* L;          // (T, N, N)
  const double* Linv_diag;  // (T, ceil(N/16), 16, 16)
  const double* alpha;      // (T, N)
  const double* y_mean;     // (T) or NULL
  const double* y_std;      // (T) or NULL
  const int32_t* n_points;  // (T) or NULL
  double* mu;               // (T, M) or NULL
  double* var;              // (T, M) or NULL
  double* V;                // (T, N, M) or NULL
  int T, N, M, D;
  int x_in_lds;
  int xq_per_task;  // Xq is (T, M, D) instead of (M, D)
  int mean_only;    // skip the triangular solve: only mu = m + s K_* alpha is produced
  // fused covariance block (gp_posterior_linv_kernel only): with VA = V[:, :Ma] (T, N, Ma) of the first Ma query points given,
  // cov (T, Ma, M) = s^2 (os k(xq_a, xq_c) - VA^T V) comes out of the same pass and V itself never reaches memory
  const double* VA;
  double* cov;
  int Ma;
  int pad_;
  // the Ma leading points of the covariance block when they are not the head of Xq (input-gradient pass: Xq holds the query
  // points only); NULL: the first Ma rows of Xq
  const double* Xa;
};

struct PosteriorCovParams {
  const double* Xq;     // (M, D)
  const double* theta;  // (T, D+2)
  const double* V;      // (T, N, M)
  const double* y_std;  // (T) or NULL
  double* cov;          // (T, Ma, M)
  int T, N, M, Ma, D;
  int xq_per_task;
};

struct ChoSolveParams {
  const double* L;          // (T, N, N)
  const double* Linv_diag;  // (T, ceil(N/16), 16, 16)
  const double* B;          // (T, N, R) right-hand sides
  const int32_t* n_points;  // (T) or NULL
  double* Xout;             // (T, N, R): (L L^T)^-1 B, or L^-T B (mode 1)
  int T, N, R;
  int mode;                 // 0: forward + backward substitution, 1: backward only (L^T x = b)
};

struct KernelMatrixParams {
  const double* X1;     // (T, N1, D)
  const double* X2;     // (T, N2, D), (N2, D) when x2_shared, or NULL: X2 = X1 (square, symmetric)
  const double* theta;  // (T, D+2)
  double* K;            // (T, N1, N2)
  int T, N1, N2, D;
  int x2_shared, add_noise;
};

struct LinvParams {
  const double* L;          // (T, N, N)
  const double* Linv_diag;  // (T, ceil(N/16), 16, 16)
  const int32_t* n_points;  // (T) or NULL
  double* Linv;             // (T, N, N) out: L^-1 (dense, zero above the diagonal)
  int T, N;
  int lower_only;           // 1: the block rows above a strip's diagonal block are not written (callers that never read them)
};

struct MllGradParams {
  const double* X;          // (T, N, D)
  const double* theta;      // (T, D+2)
  const double* alpha;      // (T, N)
  const double* Linv;       // (T, N, N)
  const int32_t* n_points;  // (T) or NULL
  double* partials;         // (T, tiles, D+2), tiles = nb (nb + 1) / 2, nb = ceil(N/16)
  int T, N, D;
};

struct TargetAssembleParams {
  const double* cov_s;          // (n, n + M) weighted source covariance block, original units
  const double* mean_s;         // (n + M)   weighted source mean
  const double* var_s;          // (n + M)   weighted source variances (only [n:] is read)
  const double* Xall;           // (n + M, D) target training inputs, then the query points
  const double* theta;          // (D + 2) target kernel: lengthscales, outputscale, noise
  const double* train_targets;  // (n) standardised target observations
  double m_all, s_all;
  double* Knn;                  // (n, n)
  double* resid;                // (n)
  double* Knq;                  // (n, M)
  double* mean_q;               // (M)
  double* var_q;                // (M)
  int n, M, D;
};

struct MllGradFusedParams {
  const double* X;          // (T, N, D)
  const double* theta;      // (T, D+2)
  const double* L;          // (T, N, N) lower factor (strict upper triangle never read)
  const double* Linv_diag;  // (T, ceil(N/16), 16, 16)
  const double* alpha;      // (T, N)
  const int32_t* n_points;  // (T) or NULL
  double* partials;         // (T, tiles, D+2): the task's totals go into tile slot 0, zeros into the others
  int T, N, D;
};

}  // namespace scaml

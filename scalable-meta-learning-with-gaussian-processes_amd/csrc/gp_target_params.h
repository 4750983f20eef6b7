// Kernel-argument block of the target-GP fit kernel (csrc/gp_target_fit.hip), shared by device source and host launcher.
#pragma once
#include <stdint.h>

namespace scaml {

// One hyper-prior: log-density on the CONSTRAINED value, evaluated as gpytorch does.
//   kind 0: none;  1: Gamma(concentration = p1, rate = p2);  2: LogNormal(loc = p1, scale = p2)
// c0 is the additive constant of the log-density (host: Gamma c ln r - lgamma(c); LogNormal -ln scale - ln(2 pi) / 2).
struct TargetPrior {
  int kind;
  int pad_;
  double p1, p2, c0;
};

// Constraints and priors of the target GP (scamlgp/model.py:25-33, 73-105, 318-338): sigmoid Interval on
// lengthscales / outputscale / noise, a plain box bound on the weights.
struct TargetSpec {
  double ls_lo, ls_hi, os_lo, os_hi, nz_lo, nz_hi;
  TargetPrior ls_prior, os_prior, nz_prior, w_prior;
  double w_lower;   // optimiser only: w >= w_lower (GreaterThan(1e-10, transform=None), model.py:334)
};

struct TargetFitParams {
  const double* means_t;   // (T, n)            source posterior means at the target inputs, original units
  const double* covs_p;    // (T, n (n + 1) / 2) source posterior covariances, packed lower triangle (a >= b at a (a + 1) / 2 + b)
  const double* X;         // (n, D) target inputs
  const double* y;         // (n)    target observations standardised with (m_all, s_all)
  double m_all, s_all;
  TargetSpec spec;
  double* z;               // (B, P) P = D + 2 + T: [raw lengthscales, raw outputscale, raw noise, weights]; optimiser: start points in, optima out
  double* value;           // (B) objective mll = [log N(y | mean, cov) + log priors] / n at z (optimiser: at the optimum)
  double* grad;            // (B, P) d mll / d z (evaluation mode; may be NULL in optimiser mode)
  int32_t* info;           // (B) 0 ok, k > 0: pivot k not positive even with the largest jitter
  double* jitter;          // (B) jitter that succeeded (NULL ok)
  double* workspace;       // optimiser: B * (6 + 2 history) * P doubles
  int32_t* stats;          // optimiser: (B, 4) iterations, evaluations, status, reserved (NULL ok)
  int B, n, T, D;
  int kind;
  int mode;                // 0: value + gradient at z, 1: L-BFGS from z
  int max_iter, history, max_ls;
  int use_mfma;            // 1: factorisation on the matrix cores (16 x 16 tiles in LDS; n <= 112), 0: column-by-column elimination
  double gtol, ftol;
};

constexpr int TARGET_FIT_DMAX = 16;
constexpr int TARGET_FIT_HMAX = 16;

}  // namespace scaml

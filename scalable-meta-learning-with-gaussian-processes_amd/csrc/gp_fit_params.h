// Kernel-argument block of gp_fit_fused_kernel, shared by the device source and the host launcher.
#pragma once
#include <stdint.h>

namespace scaml {

struct FitParams {
  const double* X;
  const double* y;
  const double* theta;
  const int32_t* n_points;
  const double* jitter_in;
  const double* A_in;       // (T, N, N) or NULL: factor this matrix (lower triangle read) instead of building K
  double* L;
  double* alpha;
  double* quad;
  double* logdet;
  double* mll;
  int32_t* info;
  double* jitter_used;
  double* Linv_diag;
  int T, N, D;
  unsigned flags;
};

}  // namespace scaml

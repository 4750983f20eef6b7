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

// Second argument of gp_fit_blocked_kernel: the same fit run IN PLACE on a diagonal block of a larger task
// (csrc/gp_fit_blocked.hip).  A plain fit has ldl = N, stride_x = N D, stride_y = N, stride_L = N N, stride_W = ceil(N/16) 256.
struct FitBlockParams {
  long long stride_x;       // doubles between the tasks' point sets
  long long stride_y;       // ... between the tasks' targets (and alphas)
  long long stride_L;       // ... between the tasks' factors
  long long stride_W;       // ... between the tasks' stacks of inverted diagonal blocks
  const int32_t* active;    // (T) or NULL: tasks with active[t] == 0 are left untouched
  int ldl;                  // leading dimension of L
};

}  // namespace scaml

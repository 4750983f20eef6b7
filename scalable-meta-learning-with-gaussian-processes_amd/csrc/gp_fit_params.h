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

constexpr unsigned FIT_FORWARD_ONLY = 0x100u;   // gp_fit_blocked_kernel only (not part of the C ABI): `alpha` receives v = L^-1 y

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

// Kernel-argument block of the blocked fit's own kernels (csrc/gp_fit_blocked.hip): the caller's arrays, then the workspace.
struct BlockedFitParams {
  const double* X;          // (T, N, D)
  const double* y;          // (T, N)
  const double* theta;      // (T, D+2)
  const int32_t* n_points;  // (T) or NULL
  const double* jitter_in;  // (T) or NULL
  double* L;                // (T, N, N)
  double* alpha;            // (T, N)
  double* quad;
  double* logdet;
  double* mll;
  int32_t* info;
  double* jitter_used;
  double* Linv_diag;        // (T, N/16, 16, 16)
  // workspace (scaml_gp_fit_blocked_workspace_bytes)
  double* S;                // (T, N2, N2) Schur complement, lower triangle
  double* Vimg;             // (T, 16 strips, 16 block rows, 64 lanes, 4) blocks of L21^T as the matrix core holds them
  double* r2;               // (T, N): [N1:] = y2 - K21 alpha1'
  double* q12;              // [4][T]: quad / logdet of block 1, of block 2
  double* jit_cur;          // (T) this round's jitter (caller's + ladder)
  double* jit_ladder;       // (T) the ladder value alone (-> jitter_used)
  int32_t* n1;              // (T) points in block 1
  int32_t* n2;              // (T) ... in block 2
  int32_t* active;          // (T) 0: the task is done, this round leaves it alone
  int32_t* info1;           // (T) status of block 1
  int32_t* info2;           // (T) status of block 2
  int T, N, D;
  unsigned flags;
  int round;
};

// Kernel-argument block of gp_fit_coop_kernel (csrc/gp_fit_coop.hip): the caller's arrays, then the workspace.
struct CoopFitParams {
  const double* X;          // (T, N, D)
  const double* y;          // (T, N)
  const double* theta;      // (T, D+2)
  const int32_t* n_points;  // (T) or NULL
  const double* jitter_in;  // (T) or NULL
  double* L;                // (T, N, N)
  double* alpha;            // (T, N)
  double* quad;
  double* logdet;
  double* mll;
  int32_t* info;
  double* jitter_used;
  double* Linv_diag;        // (T, N/16, 16, 16)
  // workspace
  unsigned* prog;           // (T, 32) per block column: attempt << 8 | block rows published   [zeroed before every launch]
  unsigned* status;         // (T, 4) [0] attempts that failed, [1] attempt << 20 | failing pivot   [zeroed before every launch]
  unsigned* xcc;            // (T, 8) 1 + the XCC id every part runs on   [zeroed before every launch]
  double* v;                // (T, N) L^-1 y
  double* part;             // (T, 32, 2) per block column: sum v^2, sum 2 log diag
  int T, N, D;
  unsigned flags;
  int parts;                // workgroups per task
};

}  // namespace scaml

// TEST INFRASTRUCTURE, not product code: csrc/gp_target_fit.hip compiled as single-threaded host code (SCAML_HOST_EMUL: one
// "thread", barriers are no-ops, every cooperative loop runs sequentially), so that the arithmetic of the target-GP objective,
// its analytic gradient and the L-BFGS driver can be checked against the oracle on a machine without a GPU.  Never linked into
// libscaml_hip.so; only tests/test_target_fit_emul.py builds and loads it.
#define SCAML_HOST_EMUL 1
#include <math.h>
#include <string.h>
#include <vector>
#include "gp_target_fit.hip"

extern "C" int emul_target_fit(const double* means_t, const double* covs_p, const double* X, const double* y, double m_all, double s_all,
                               const double* spec, double* z, int B, int n, int T, int D, int kind, int mode, int max_iter, int history,
                               double gtol, double ftol, double* value, double* grad, int32_t* info, double* jitter, int32_t* stats) {
  using namespace scaml;
  TargetFitParams p;
  memset(&p, 0, sizeof(p));
  p.means_t = means_t; p.covs_p = covs_p; p.X = X; p.y = y; p.m_all = m_all; p.s_all = s_all;
  TargetSpec& sp = p.spec;
  sp.ls_lo = spec[0]; sp.ls_hi = spec[1]; sp.os_lo = spec[2]; sp.os_hi = spec[3]; sp.nz_lo = spec[4]; sp.nz_hi = spec[5];
  TargetPrior* pr[4] = {&sp.ls_prior, &sp.os_prior, &sp.nz_prior, &sp.w_prior};
  for (int q = 0; q < 4; ++q) {
    pr[q]->kind = (int)spec[6 + 3 * q];
    pr[q]->p1 = spec[7 + 3 * q];
    pr[q]->p2 = spec[8 + 3 * q];
    pr[q]->c0 = pr[q]->kind == 1 ? pr[q]->p1 * log(pr[q]->p2) - lgamma(pr[q]->p1)
                                 : (pr[q]->kind == 2 ? -log(pr[q]->p2) - 0.5 * log(2.0 * M_PI) : 0.0);
  }
  sp.w_lower = spec[18];
  const int P = D + 2 + T;
  std::vector<double> ws((size_t)B * (6 + 2 * history) * P + 1);
  p.z = z; p.value = value; p.grad = grad; p.info = info; p.jitter = jitter; p.workspace = ws.data(); p.stats = stats;
  p.B = B; p.n = n; p.T = T; p.D = D; p.kind = kind; p.mode = mode; p.max_iter = max_iter; p.history = history; p.max_ls = 20;
  p.gtol = gtol; p.ftol = ftol;
  std::vector<double> lds((size_t)(n + 1) * (n + 2) + n * D + 8 * n + 2 * T + 3 * D + 200);
  for (int b = 0; b < B; ++b) {
    TfCtx c;
    c.tid = 0; c.nthr = 1; c.lane = 0; c.wave = 0; c.nwave = 1; c.solo = 0;
    c.n = n; c.T = T; c.D = D; c.P = P; c.E = n * (n + 1) / 2; c.kind = kind;
    tf_carve(c, lds.data(), n, T, D, 1, 0);
    tf_main(c, p, b);
  }
  return 0;
}

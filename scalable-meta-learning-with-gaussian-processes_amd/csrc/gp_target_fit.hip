// gp_target_fit.hip — the target GP's training objective, its analytic gradient and the whole L-BFGS refit in ONE launch.
//
// Replaces what the reference runs on every report(): scamlgp/optimizer.py:176-185 rebuilds ScaMLGP and calls
// optimize_marginal_likelihood (scamlgp/utils.py:139-212), which drives scipy L-BFGS-B over
//   mll(z) = [ log N(y~ | mean, cov + sigma^2 I) + log priors ] / n                 (scamlgp/model.py:360-363, 376-383)
//   mean = (source_means w - m_all) / s_all,  cov = source_covs w^2 / s_all^2 + os k_t(X, X; l)
// with torch autograd through kernel, Cholesky and solves per evaluation -- ~700 evaluations per BO step at BASELINE configs[4].
// Here one workgroup owns one start point (the warm start and every prior-sampled restart run side by side): the n x n matrix
// lives in LDS, an evaluation is
//   build      cov from the packed source covariances (coalesced over elements, sum over tasks), target kernel, noise, jitter
//   factorise  n <= 112: blocked Cholesky on the matrix cores -- 16 x 16 tiles in LDS, the diagonal tile factored and inverted by one
//              wave in registers (pivots by v_readlane), panel / trailing update / L^-1 / K^-1 as v_mfma_f64_16x16x4 tile products
//              (tf_factor_mfma);  112 < n <= 128 and the host-emulation build: square-root-free elimination by columns
//              (A = L~ D^-1 L~^T, same pivots as LL^T), the right-hand side riding along as row n (its Schur complement is -quad)
//              and the identity as n more right-hand sides, so ONE sweep with ONE barrier per column yields pivots, L^-1 y and L^-1
//   K^-1, alpha, G = (alpha alpha^T - K^-1) / 2, packed
//   gradient   d/dw_i = <G, 2 w_i C_i / s^2> + alpha . M_i / s   (one wave per task, coalesced over the packed elements)
//              d/d(l, os, noise) from one pass over the kernel elements; chain rule through the sigmoid Interval, priors
// and the optimiser (two-loop L-BFGS, projected backtracking line search on the box w >= w_lower, curvature pairs in the free
// subspace, scipy L-BFGS-B's stopping rules) runs in the same kernel on vectors in a small global workspace: the kernel body is one
// loop around ONE inlined call of the evaluation (tf_run).  psd_safe_cholesky's jitter ladder (0, 1e-8, 1e-7, 1e-6 on the diagonal)
// is applied per evaluation, in-kernel.
//
// Layout: packed lower triangle for everything the gradient phases touch -- element (a, b), a >= b, at a (a + 1) / 2 + b.
// Throughput is not the point of this kernel (B <= a handful of workgroups on an otherwise idle chip); latency per evaluation is:
// 101 us at n = 80, T = 32 (228 us through the column-by-column elimination; DESIGN.md 4d, profiles/r03_notes.md).
//
// SCAML_HOST_EMUL: the same source compiles as single-threaded host code (tests/host_emul: arithmetic of objective, gradient
// and optimiser checked against the oracle on CPU; never part of libscaml_hip.so).
#ifndef SCAML_HOST_EMUL
#include <hip/hip_runtime.h>
#include "scaml_common.hpp"
#define TF_DEV __device__ __forceinline__
#define TF_SYNC() __syncthreads()
#define TF_LANES 64
#else
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#define TF_DEV static inline
#define TF_SYNC() ((void)0)
#define TF_LANES 1
#endif
#include "gp_target_params.h"

// diagnostic builds (tools/target_fit_probe.hip): phase stamps of one evaluation, 100 MHz wall clock, thread 0
#if defined(SCAML_TF_STAMPS) && !defined(SCAML_HOST_EMUL)
__device__ long long* g_tf_stamps;
#define TF_STAMP(i) do { if (c.tid == 0 && g_tf_stamps) g_tf_stamps[i] = (long long)wall_clock64(); } while (0)
#else
#define TF_STAMP(i) ((void)0)
#endif

namespace scaml {

struct TfCtx {
  int tid, nthr, lane, wave, nwave;
  int n, T, D, P, E, kind;
  // LDS
  double *Ap, *Tp, *Xs, *col, *piv, *rs, *alpha, *vv, *w, *w2, *theta, *dth, *invl, *part, *red, *sc;
  // matrix-core path (n <= 112): 16 x 16 tiles at pitch 17, lower block triangle, tile (i, j) at (i (i + 1) / 2 + j) * TF_TS
  double *Lt, *Xt;
  int nb, mfma;
  int solo;   // optimiser bookkeeping by ONE wave (P <= 64: a lane per variable): reductions stay in the wave, barriers are no-ops
};
constexpr int TF_TP = 17, TF_TS = 16 * TF_TP;

TF_DEV int tf_idxL(int a, int b) { return a * (a + 1) / 2 + b; }
TF_DEV int tf_rowT(int j, int n) { return j * n - j * (j - 1) / 2 - j; }   // row j of the upper-packed block: (j, k), k >= j, at tf_rowT(j) + k
TF_DEV int tf_idxT(int j, int k, int n) { return tf_rowT(j, n) + k; }

TF_DEV void tf_decode(int e, int& a, int& b) {
  int r = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
  while (r * (r + 1) / 2 > e) --r;
  while ((r + 1) * (r + 2) / 2 <= e) ++r;
  a = r;
  b = e - r * (r + 1) / 2;
}

// total over the wave, valid in the last lane
TF_DEV double tf_wave_sum(double x) {
#ifndef SCAML_HOST_EMUL
  return wave_sum_to_lane15(x);
#else
  return x;
#endif
}
TF_DEV bool tf_last_lane(const TfCtx& c) { return c.lane == TF_LANES - 1; }

// barrier of the optimiser's bookkeeping: a real one, or nothing when a single wave does the work (its own program order is enough:
// a lane only ever touches its own vector component, dot products travel through the wave reduction)
#define TF_OSYNC(c) do { if (!(c).solo) TF_SYNC(); } while (0)

// sum over the workgroup, result in every thread (fixed order: deterministic).  Two barriers (none in solo mode).
TF_DEV double tf_block_sum(const TfCtx& c, double x) {
  const double s = tf_wave_sum(x);
#ifndef SCAML_HOST_EMUL
  if (c.solo) return readlane_f64(s, 63);
#endif
  if (tf_last_lane(c)) c.red[c.wave] = s;
  TF_SYNC();
  double t = 0.0;
  for (int v = 0; v < c.nwave; ++v) t += c.red[v];
  TF_SYNC();
  return t;
}
TF_DEV double tf_block_max(const TfCtx& c, double x) {
  // (max needs no order; waves publish their lanes' maxima through LDS)
#ifndef SCAML_HOST_EMUL
  for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_xor(x, off));
  if (c.solo) return x;
#endif
  if (c.lane == 0) c.red[c.wave] = x;
  TF_SYNC();
  double t = c.red[0];
  for (int v = 1; v < c.nwave; ++v) t = fmax(t, c.red[v]);
  TF_SYNC();
  return t;
}

TF_DEV double tf_prior_logp(const TargetPrior& p, double x) {
  if (p.kind == 1) return p.c0 + (p.p1 - 1.0) * log(x) - p.p2 * x;
  if (p.kind == 2) {
    const double lx = log(x), u = (lx - p.p1) / p.p2;
    return p.c0 - lx - 0.5 * u * u;
  }
  return 0.0;
}
TF_DEV double tf_prior_dlogp(const TargetPrior& p, double x) {
  if (p.kind == 1) return (p.p1 - 1.0) / x - p.p2;
  if (p.kind == 2) return -(1.0 + (log(x) - p.p1) / (p.p2 * p.p2)) / x;
  return 0.0;
}

// 1 / x for a pivot: f32 seed + two Newton steps (error 2^-22 -> 2^-44 -> 2^-88, then rounding) instead of the ~40-instruction
// fp64 division, which sat on every column's critical path; outside the f32 range the plain division
TF_DEV double tf_rcp(double x) {
#ifndef SCAML_HOST_EMUL
  if (x > 1e-30 && x < 1e30) {
    double y = (double)__builtin_amdgcn_rcpf((float)x);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    return y;
  }
#endif
  return 1.0 / x;
}

// k(x_a, x_b) / os and d(k / os) / d(d2) for the scaled squared distance d2 (a != b)
TF_DEV void tf_kernel(int kind, double d2, double& kk, double& dk) {
  if (kind == 0) {
    kk = exp(-0.5 * d2);
    dk = -0.5 * kk;
  } else {
    const double s5 = 2.2360679774997896964;
    const double r = sqrt(fmax(d2, 1e-30));   // gpytorch clamps the squared distance before the root
    const double e5 = exp(-s5 * r);
    kk = (1.0 + s5 * r + (5.0 / 3.0) * r * r) * e5;
    dk = -(5.0 / 6.0) * (1.0 + s5 * r) * e5;
  }
}

#ifndef SCAML_HOST_EMUL
// ---- the factorisation on the matrix cores (n <= 112) ------------------------------------------------------------------------
// The column-by-column elimination above costs a barrier, an LDS round trip for the pivot and a reciprocal PER COLUMN: 2 us each,
// 163 us of a 228-us evaluation at n = 80 -- the vector unit idles on latencies.  Here: blocked Cholesky on 16 x 16 tiles in LDS.
//   per block column K:  wave 0 factors the diagonal tile IN REGISTERS (a lane holds a row; pivots and column entries travel by
//                        v_readlane, no LDS round trip, no barrier) and inverts it (W_K = L_KK^-1, a lane per column);
//                        panel  L_iK = A_iK W_K^T  and trailing update  A_ij -= L_iK L_jK^T  as 4 v_mfma_f64_16x16x4 per tile
//   X = L^-1             by block columns (X_kk = W_k, X_ik = -W_i sum_j L_ij X_jk): the accumulator layout of one product IS the
//                        B-operand layout of the next (lane (lc, lq) register m = element [4 m + lq][lc])
//   v = X r, alpha = X^T v, K^-1 = X^T X by tiles, G = (alpha alpha^T - K^-1) / 2 packed into the (then free) L region.
// Three barriers per BLOCK column instead of one per column.  Returns 0 or the 1-based index of the failing pivot.
TF_DEV void tf_settle(d4_t& v) { asm volatile("s_nop 15\n\ts_nop 2" : "+v"(v)); }   // gfx950: an MFMA's last result pair is not interlocked
TF_DEV int tf_tile(int i, int j) { return (i * (i + 1) / 2 + j) * TF_TS; }

TF_DEV int tf_factor_mfma(const TfCtx& c, const TargetFitParams& p, double os, double noise, double jit, double& quad, double& logdet) {
  const int n = c.n, T = c.T, D = c.D, E = c.E, nb = c.nb;
  const int lane = c.lane, lc = lane & 15, lq = lane >> 4;
  const double inv_s = 1.0 / p.s_all;
  // ---- build: lower block triangle of K (+ noise + jitter), identity padding past n; residual r into vv ----
  const int ntile = nb * (nb + 1) / 2;
  for (int e = c.tid; e < ntile * 256; e += c.nthr) {
    const int t = e >> 8, r = (e >> 4) & 15, cc = e & 15;
    int ti, tj;
    tf_decode(t, ti, tj);
    const int a = 16 * ti + r, b = 16 * tj + cc;
    double val = a == b ? 1.0 : 0.0;
    if (a < n && b < n) {
      val = 0.0;
      if (b <= a) {
        double acc = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
        const double* cp = p.covs_p + tf_idxL(a, b);
        int i = 0;
        for (; i + 16 <= T; i += 16) {   // sixteen task rows in flight: the pass is a chain of L2 round trips, not of bytes
          double x[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) x[j] = cp[(size_t)(i + j) * E];
#pragma unroll
          for (int j = 0; j < 16; j += 4) {
            acc += c.w2[i + j] * x[j];
            acc1 += c.w2[i + j + 1] * x[j + 1];
            acc2 += c.w2[i + j + 2] * x[j + 2];
            acc3 += c.w2[i + j + 3] * x[j + 3];
          }
        }
        for (; i + 4 <= T; i += 4) {
          const double x0 = cp[(size_t)i * E], x1 = cp[(size_t)(i + 1) * E], x2 = cp[(size_t)(i + 2) * E], x3 = cp[(size_t)(i + 3) * E];
          acc += c.w2[i] * x0;
          acc1 += c.w2[i + 1] * x1;
          acc2 += c.w2[i + 2] * x2;
          acc3 += c.w2[i + 3] * x3;
        }
        for (; i < T; ++i) acc += c.w2[i] * cp[(size_t)i * E];
        acc = (acc + acc1) + (acc2 + acc3);
        double k = os;
        if (a != b) {
          double d2 = 0.0;
          for (int d = 0; d < D; ++d) {
            const double df = (c.Xs[a * D + d] - c.Xs[b * D + d]) * c.invl[d];
            d2 += df * df;
          }
          double kk, dk;
          tf_kernel(c.kind, d2, kk, dk);
          k = os * kk;
        } else {
          k += noise + jit;
        }
        val = acc + k;
      }
    }
    c.Lt[t * TF_TS + r * TF_TP + cc] = val;
  }
  for (int b = c.tid; b < 16 * nb; b += c.nthr) {
    double rv = 0.0;
    if (b < n) {
      double m = 0.0, m1 = 0.0;
      int i = 0;
      for (; i + 4 <= T; i += 4) {
        const double x0 = p.means_t[(size_t)i * n + b], x1 = p.means_t[(size_t)(i + 1) * n + b], x2 = p.means_t[(size_t)(i + 2) * n + b],
                     x3 = p.means_t[(size_t)(i + 3) * n + b];
        m += c.w[i] * x0 + c.w[i + 2] * x2;
        m1 += c.w[i + 1] * x1 + c.w[i + 3] * x3;
      }
      for (; i < T; ++i) m += c.w[i] * p.means_t[(size_t)i * n + b];
      rv = p.y[b] - (m + m1 - p.m_all) * inv_s;
    }
    c.vv[b] = rv;
  }
  if (c.tid == 0) c.sc[0] = 0.0;   // failure word
  TF_SYNC();
  TF_STAMP(1);
  // ---- blocked Cholesky ----
  for (int K = 0; K < nb; ++K) {
    if (c.wave == 0) {
      // diagonal tile in registers: lane l (and its three mirrors l + 16 m) holds row l & 15
      double* Dt = c.Lt + tf_tile(K, K);
      double row[16];
#pragma unroll
      for (int b = 0; b < 16; ++b) row[b] = Dt[lc * TF_TP + b];
      // Factor and invert in ONE sweep over the pivots: lane l is row l of L (registers row[]) and, at the same time, column l of
      // W = L^-1 (registers wv[], partial sums sp[]).  Column cc of L is final after pivot cc; its entries L[b][cc] travel by
      // v_readlane to every lane, where they serve both the rank-1 update of the rows (row[b] -= L[l][cc] L[b][cc]) and the forward
      // substitution of the inverse (sp[b] += L[b][cc] W[cc][l], W[cc][l] = (delta - sp[cc]) / L[cc][cc]): the inverse costs one
      // more FMA per broadcast instead of a second 120-broadcast pass (8 k -> 4.5 k cycles per block).
      int bad = 0;
      double wv[16], sp[16];
#pragma unroll
      for (int b = 0; b < 16; ++b) sp[b] = 0.0;
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) {
        const double pv = readlane_f64(row[cc], cc);
        if (!(pv > 0.0) && bad == 0) bad = 16 * K + cc + 1;
        const double rsq = rsqrt_pos(pv);                        // 1 / L[cc][cc]
        const double x = lc == cc ? pv * rsq : row[cc] * rsq;   // L[l][cc] for l >= cc (rows above hold junk that is never read)
        row[cc] = x;
        const double wcc = ((lc == cc ? 1.0 : 0.0) - sp[cc]) * rsq;   // W[cc][l] (zero for cc < l: nothing has been added to sp yet)
        wv[cc] = wcc;
#pragma unroll
        for (int b = cc + 1; b < 16; ++b) {
          const double xb = readlane_f64(x, b);                  // L[b][cc]
          row[b] = __builtin_fma(-x, xb, row[b]);                 // (only b <= l matters)
          sp[b] = __builtin_fma(xb, wcc, sp[b]);
        }
      }
      if (lane < 16) {
        double* Wt = c.Xt + tf_tile(K, K);
#pragma unroll
        for (int b = 0; b < 16; ++b) {
          Dt[lc * TF_TP + b] = b <= lc ? row[b] : 0.0;
          Wt[b * TF_TP + lc] = wv[b];
        }
        if (16 * K + lc < n) c.piv[16 * K + lc] = row[lc];   // diag(L): logdet = 2 sum log
      }
      if (lane == 0 && bad) c.sc[0] = (double)bad;
    }
    TF_SYNC();
    const int fail = (int)c.sc[0];
    if (fail) return fail;
    // panel: L_iK = A_iK W_K^T   (A operand: A_iK[lc][4 m + lq], B operand: (W^T)[4 m + lq][lc] = W[lc][4 m + lq])
    {
      const double* Wt = c.Xt + tf_tile(K, K);
      double wb[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) wb[m] = Wt[lc * TF_TP + 4 * m + lq];
      for (int i = K + 1 + c.wave; i < nb; i += c.nwave) {
        double* At = c.Lt + tf_tile(i, K);
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(At[lc * TF_TP + 4 * m + lq], wb[m], acc, 0, 0, 0);
        tf_settle(acc);
#pragma unroll
        for (int g = 0; g < 4; ++g) At[(lq + 4 * g) * TF_TP + lc] = acc[g];
      }
    }
    TF_SYNC();
    // trailing update: A_ij -= L_iK L_jK^T, K < j <= i
    {
      const int m1 = nb - K - 1, cnt = m1 * (m1 + 1) / 2;
      for (int q = c.wave; q < cnt; q += c.nwave) {
        int ii, jj;
        tf_decode(q, ii, jj);
        const int i = K + 1 + ii, j = K + 1 + jj;
        const double* Li = c.Lt + tf_tile(i, K);
        const double* Lj = c.Lt + tf_tile(j, K);
        double* At = c.Lt + tf_tile(i, j);
        d4_t acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = At[(lq + 4 * g) * TF_TP + lc];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Li[lc * TF_TP + 4 * m + lq], Lj[lc * TF_TP + 4 * m + lq], acc, 0, 0, 0);
        tf_settle(acc);
#pragma unroll
        for (int g = 0; g < 4; ++g) At[(lq + 4 * g) * TF_TP + lc] = acc[g];
      }
    }
    TF_SYNC();
  }
  TF_STAMP(2);
  // ---- X = L^-1 by block columns: X_ik = -W_i sum_{j = k}^{i - 1} L_ij X_jk ----
  for (int k = c.wave; k < nb; k += c.nwave) {
    for (int i = k + 1; i < nb; ++i) {
      d4_t sacc = {0.0, 0.0, 0.0, 0.0};
      for (int j = k; j < i; ++j) {
        const double* Lij = c.Lt + tf_tile(i, j);
        const double* Xjk = c.Xt + tf_tile(j, k);
#pragma unroll
        for (int m = 0; m < 4; ++m) sacc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lij[lc * TF_TP + 4 * m + lq], Xjk[(4 * m + lq) * TF_TP + lc], sacc, 0, 0, 0);
      }
      tf_settle(sacc);
      const double* Wi = c.Xt + tf_tile(i, i);
      d4_t x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int m = 0; m < 4; ++m) x = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wi[lc * TF_TP + 4 * m + lq], sacc[m], x, 0, 0, 0);
      tf_settle(x);
      double* Xik = c.Xt + tf_tile(i, k);
#pragma unroll
      for (int g = 0; g < 4; ++g) Xik[(lq + 4 * g) * TF_TP + lc] = x[g];
    }
  }
  TF_SYNC();
  // ---- v = X r, quad, logdet; alpha = X^T v ----
  double qd = 0.0, ld = 0.0;
  for (int a = c.tid; a < n; a += c.nthr) {
    const int ta = a >> 4, ra = a & 15;
    double sacc = 0.0;
    for (int b = 0; b <= a; ++b) sacc += c.Xt[tf_tile(ta, b >> 4) + ra * TF_TP + (b & 15)] * c.vv[b];
    c.rs[a] = sacc;   // v
    qd += sacc * sacc;
    ld += 2.0 * log(c.piv[a]);
  }
  quad = tf_block_sum(c, qd);
  logdet = tf_block_sum(c, ld);
  for (int b = c.tid; b < n; b += c.nthr) {
    const int tb = b >> 4, rb = b & 15;
    double sacc = 0.0;
    for (int a = b; a < n; ++a) sacc += c.Xt[tf_tile(a >> 4, tb) + (a & 15) * TF_TP + rb] * c.rs[a];
    c.alpha[b] = sacc;
  }
  TF_SYNC();
  TF_STAMP(3);
  TF_STAMP(4);   // (the column-by-column path's "scale U + alpha" phase has no counterpart here)
  // ---- K^-1 = X^T X by tiles, G = (alpha alpha^T - K^-1) / 2 (off-diagonal doubled) packed into the L region ----
  // (every L tile has been read for the last time before the barrier above)
  {
    const int cnt = nb * (nb + 1) / 2;
    for (int q = c.wave; q < cnt; q += c.nwave) {
      int ta, tb;
      tf_decode(q, ta, tb);
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      for (int k = ta; k < nb; ++k) {
        const double* Xa = c.Xt + tf_tile(k, ta);
        const double* Xb = c.Xt + tf_tile(k, tb);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xa[(4 * m + lq) * TF_TP + lc], Xb[(4 * m + lq) * TF_TP + lc], acc, 0, 0, 0);
      }
      tf_settle(acc);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int a = 16 * ta + lq + 4 * g, b = 16 * tb + lc;
        if (a < n && b <= a) {
          const double gv = 0.5 * (c.alpha[a] * c.alpha[b] - acc[g]);
          c.Ap[tf_idxL(a, b)] = a == b ? gv : 2.0 * gv;
        }
      }
    }
  }
  TF_SYNC();
  return 0;
}
#endif

// One evaluation at z (P doubles, global): returns mll (every thread) and, if gz != nullptr, d mll / d z.
// info_out / jit_out: status of the factorisation (thread 0 writes them if given).  A matrix that is not positive definite
// even with the largest jitter gives NaN.
TF_DEV double tf_eval(const TfCtx& c, const TargetFitParams& p, const double* z, double* gz, int32_t* info_out, double* jit_out) {
  const int n = c.n, T = c.T, D = c.D, E = c.E;
  const TargetSpec& sp = p.spec;
  const double inv_s = 1.0 / p.s_all, inv_s2 = inv_s * inv_s;
  // ---- parameters ----
  for (int i = c.tid; i < D + 2; i += c.nthr) {
    const double lo = i < D ? sp.ls_lo : (i == D ? sp.os_lo : sp.nz_lo), hi = i < D ? sp.ls_hi : (i == D ? sp.os_hi : sp.nz_hi);
    const double s = 1.0 / (1.0 + exp(-z[i]));
    const double th = lo + (hi - lo) * s;
    c.theta[i] = th;
    c.dth[i] = (hi - lo) * s * (1.0 - s);
    if (i < D) c.invl[i] = 1.0 / th;
  }
  for (int i = c.tid; i < T; i += c.nthr) {
    const double wi = z[D + 2 + i];
    c.w[i] = wi;
    c.w2[i] = wi * wi * inv_s2;
  }
  TF_SYNC();
  TF_STAMP(0);
  const double os = c.theta[D], noise = c.theta[D + 1];
  int fail = 0;
  double jit = 0.0;
  double quad_m = 0.0, logdet_m = 0.0;
  for (int attempt = 0; attempt < 4; ++attempt) {
    jit = attempt == 0 ? 0.0 : (attempt == 1 ? 1e-8 : (attempt == 2 ? 1e-7 : 1e-6));
#ifndef SCAML_HOST_EMUL
    if (c.mfma) {
      fail = tf_factor_mfma(c, p, os, noise, jit, quad_m, logdet_m);
      if (!fail) break;
      TF_SYNC();
      continue;
    }
#endif
    // ---- build ----
    for (int e = c.tid; e < E; e += c.nthr) {
      int a, b;
      tf_decode(e, a, b);
      // (eight loads in flight per trip: one exposed memory round trip per task was 37 us of a 365-us evaluation)
      double acc = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
      const double* cp = p.covs_p + e;
      int i = 0;
      for (; i + 8 <= T; i += 8) {
        double x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = cp[(size_t)(i + j) * E];
        acc += c.w2[i] * x[0] + c.w2[i + 4] * x[4];
        acc1 += c.w2[i + 1] * x[1] + c.w2[i + 5] * x[5];
        acc2 += c.w2[i + 2] * x[2] + c.w2[i + 6] * x[6];
        acc3 += c.w2[i + 3] * x[3] + c.w2[i + 7] * x[7];
      }
      for (; i < T; ++i) acc += c.w2[i] * cp[(size_t)i * E];
      acc = (acc + acc1) + (acc2 + acc3);
      double k = os;
      if (a != b) {
        double d2 = 0.0;
        for (int d = 0; d < D; ++d) {
          const double df = (c.Xs[a * D + d] - c.Xs[b * D + d]) * c.invl[d];
          d2 += df * df;
        }
        double kk, dk;
        tf_kernel(c.kind, d2, kk, dk);
        k = os * kk;
      } else {
        k += noise + jit;
      }
      c.Ap[e] = acc + k;
      c.Tp[e] = 0.0;
    }
    for (int b = c.tid; b < n; b += c.nthr) {
      double m = 0.0, m1 = 0.0;
      int i = 0;
      for (; i + 4 <= T; i += 4) {
        const double x0 = p.means_t[(size_t)i * n + b], x1 = p.means_t[(size_t)(i + 1) * n + b], x2 = p.means_t[(size_t)(i + 2) * n + b],
                     x3 = p.means_t[(size_t)(i + 3) * n + b];
        m += c.w[i] * x0 + c.w[i + 2] * x2;
        m1 += c.w[i + 1] * x1 + c.w[i + 3] * x3;
      }
      for (; i < T; ++i) m += c.w[i] * p.means_t[(size_t)i * n + b];
      m += m1;
      c.Ap[tf_idxL(n, b)] = p.y[b] - (m - p.m_all) * inv_s;
    }
    if (c.tid == 0) c.Ap[tf_idxL(n, n)] = 0.0;
    TF_SYNC();
    for (int j = c.tid; j < n; j += c.nthr) c.Tp[tf_idxT(j, j, n)] = 1.0;
    for (int b = c.tid; b <= n; b += c.nthr) c.col[b] = c.Ap[tf_idxL(b, 0)];
    TF_SYNC();
    // ---- elimination: column k of the (n + 1) x (n + 1) bordered matrix, rows of the identity block riding along ----
    TF_STAMP(1);
    fail = 0;
    int first = c.wave > 0 ? c.wave : c.nwave;
    for (int k = 0; k < n; ++k) {
      if (first <= k) first += c.nwave;
      const double* cur = c.col + (k & 1) * (n + 1);
      double* nxt = c.col + ((k + 1) & 1) * (n + 1);
      const double pv = cur[k];
      if (!(pv > 0.0)) {   // (uniform: every thread reads the same LDS word; NaN lands here too)
        fail = k + 1;
        break;
      }
      const double inv = tf_rcp(pv);
      // Rows r = wave, wave + nwave, ... (cyclic: the triangle's rows get shorter), a lane per column b.  Four rows per trip: their
      // LDS reads go out together, so a trip costs one LDS round trip instead of four (the wave executes in order; with one row
      // per trip the step was a chain of exposed ~120-cycle latencies, 7 k cycles per column at n = 80).
      for (int b = k + 1 + c.lane; b <= n; b += TF_LANES) {
        const double cb = cur[b];
        // identity block: row r <= k holds (L~^-1)^T e_r so far; columns k + 1 .. n - 1 still to go
        if (b < n) {
          for (int r0 = c.wave; r0 <= k; r0 += 4 * c.nwave) {
            double f[4], v[4];
            int ix[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = r0 + j * c.nwave;
              const int base = tf_rowT(r <= k ? r : r0, n);
              ix[j] = base + b;
              f[j] = c.Tp[base + k];
              v[j] = c.Tp[ix[j]];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (r0 + j * c.nwave <= k) c.Tp[ix[j]] = v[j] - f[j] * inv * cb;
            }
          }
        }
        // matrix rows r > k (r == n: the right-hand side, whose diagonal entry collects -quad), columns k + 1 .. r
        for (int r0 = first; r0 <= n; r0 += 4 * c.nwave) {   // first: the first row > k of this wave's residue class
          double f[4], v[4];
          int ix[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = r0 + j * c.nwave;
            const int rr = r <= n ? r : r0;
            ix[j] = tf_idxL(rr, 0) + (b <= rr ? b : rr);
            f[j] = cur[rr];
            v[j] = c.Ap[ix[j]];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = r0 + j * c.nwave;
            if (r <= n && b <= r) {
              const double u = v[j] - f[j] * inv * cb;
              c.Ap[ix[j]] = u;
              if (b == k + 1) nxt[r] = u;
            }
          }
        }
      }
      TF_SYNC();
    }
    if (!fail) break;
    TF_SYNC();
  }
  if (info_out && c.tid == 0) *info_out = fail;
  if (jit_out && c.tid == 0) *jit_out = jit;
  if (fail) {
    if (gz) {
      for (int i = c.tid; i < c.P; i += c.nthr) gz[i] = 0.0;
    }
    TF_SYNC();
#ifndef SCAML_HOST_EMUL
    return __builtin_nan("");
#else
    return NAN;
#endif
  }
  // ---- pivots, v = L^-1 r, quad, logdet ----
  if (!c.mfma) TF_STAMP(2);
  double ld = 0.0;
  if (!c.mfma) {
    for (int k = c.tid; k < n; k += c.nthr) {
      const double pv = c.Ap[tf_idxL(k, k)];
      const double r = 1.0 / sqrt(pv);
      c.piv[k] = pv;
      c.rs[k] = r;
      c.vv[k] = c.Ap[tf_idxL(n, k)] * r;
      ld += log(pv);
    }
  }
  // log priors ride in the same reduction
  double lp = 0.0;
  for (int i = c.tid; i < D + 2; i += c.nthr) lp += tf_prior_logp(i < D ? sp.ls_prior : (i == D ? sp.os_prior : sp.nz_prior), c.theta[i]);
  for (int i = c.tid; i < T; i += c.nthr) lp += tf_prior_logp(sp.w_prior, c.w[i]);
  const double quad = c.mfma ? quad_m : -c.Ap[tf_idxL(n, n)];
  double logdet = tf_block_sum(c, ld);
  if (c.mfma) logdet = logdet_m;
  const double logprior = tf_block_sum(c, lp);
  const double value = (-0.5 * (quad + logdet + n * 1.8378770664093453) + logprior) / n;
  if (!gz) return value;
  if (!c.mfma) {
  TF_STAMP(3);
  // ---- U = L^-T scaled: U[j][k] = (L^-1)[k][j], rows j, columns k >= j ----
  for (int j = c.wave; j < n; j += c.nwave) {
    const int b0 = tf_rowT(j, n);
    for (int k = j + c.lane; k < n; k += TF_LANES) c.Tp[b0 + k] *= c.rs[k];
  }
  TF_SYNC();
  // alpha_a = sum_{k >= a} U[a][k] v[k]
  for (int a = c.wave; a < n; a += c.nwave) {
    const int b0 = tf_rowT(a, n);
    double acc = 0.0;
    for (int k = a + c.lane; k < n; k += TF_LANES) acc += c.Tp[b0 + k] * c.vv[k];
    acc = tf_wave_sum(acc);
    if (tf_last_lane(c)) c.alpha[a] = acc;
  }
  TF_SYNC();
  // G = (alpha alpha^T - K^-1) / 2, off-diagonal elements doubled (they stand for both halves), into Ap
  TF_STAMP(4);
  for (int e = c.tid; e < E; e += c.nthr) {
    int a, b;
    tf_decode(e, a, b);
    const int ba = tf_rowT(a, n), bb = tf_rowT(b, n);
    double s = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = a;
    for (; k + 4 <= n; k += 4) {
      const double u0 = c.Tp[ba + k], u1 = c.Tp[ba + k + 1], u2 = c.Tp[ba + k + 2], u3 = c.Tp[ba + k + 3];
      const double w0 = c.Tp[bb + k], w1 = c.Tp[bb + k + 1], w2 = c.Tp[bb + k + 2], w3 = c.Tp[bb + k + 3];
      s += u0 * w0;
      s1 += u1 * w1;
      s2 += u2 * w2;
      s3 += u3 * w3;
    }
    for (; k < n; ++k) s += c.Tp[ba + k] * c.Tp[bb + k];
    s = (s + s1) + (s2 + s3);
    const double g = 0.5 * (c.alpha[a] * c.alpha[b] - s);
    c.Ap[e] = a == b ? g : 2.0 * g;
  }
  TF_SYNC();
  }   // (column-by-column path; the matrix-core path left G in Ap and alpha in place)
  const double inv_n = 1.0 / n;
  TF_STAMP(5);
  // ---- d / d w_i: one wave per task (two tasks per trip: eight coalesced loads in flight) ----
  for (int i0 = 2 * c.wave; i0 < T; i0 += 2 * c.nwave) {
    const int i1 = i0 + 1 < T ? i0 + 1 : i0;
    const double* cp0 = p.covs_p + (size_t)i0 * E;
    const double* cp1 = p.covs_p + (size_t)i1 * E;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    int e = c.lane;
    for (; e + 3 * TF_LANES < E; e += 4 * TF_LANES) {
      const double x0 = cp0[e], x1 = cp0[e + TF_LANES], x2 = cp0[e + 2 * TF_LANES], x3 = cp0[e + 3 * TF_LANES];
      const double y0 = cp1[e], y1 = cp1[e + TF_LANES], y2 = cp1[e + 2 * TF_LANES], y3 = cp1[e + 3 * TF_LANES];
      const double g0 = c.Ap[e], g1 = c.Ap[e + TF_LANES], g2 = c.Ap[e + 2 * TF_LANES], g3 = c.Ap[e + 3 * TF_LANES];
      a0 += g0 * x0; a1 += g1 * x1; a2 += g2 * x2; a3 += g3 * x3;
      b0 += g0 * y0; b1 += g1 * y1; b2 += g2 * y2; b3 += g3 * y3;
    }
    for (; e < E; e += TF_LANES) {
      const double g0 = c.Ap[e];
      a0 += g0 * cp0[e];
      b0 += g0 * cp1[e];
    }
    double am = 0.0, bm = 0.0;
    for (int a = c.lane; a < n; a += TF_LANES) {
      const double al = c.alpha[a];
      am += al * p.means_t[(size_t)i0 * n + a];
      bm += al * p.means_t[(size_t)i1 * n + a];
    }
    const double w0 = c.w[i0], w1 = c.w[i1];
    const double t0 = tf_wave_sum(((a0 + a1) + (a2 + a3)) * (2.0 * w0 * inv_s2) + am * inv_s);
    const double t1 = tf_wave_sum(((b0 + b1) + (b2 + b3)) * (2.0 * w1 * inv_s2) + bm * inv_s);
    if (tf_last_lane(c)) {
      gz[D + 2 + i0] = (t0 + tf_prior_dlogp(sp.w_prior, w0)) * inv_n;
      if (i1 != i0) gz[D + 2 + i1] = (t1 + tf_prior_dlogp(sp.w_prior, w1)) * inv_n;
    }
  }
  // ---- d / d (lengthscales, outputscale, noise) ----
  TF_STAMP(6);
  double gl[TARGET_FIT_DMAX];
#pragma unroll
  for (int d = 0; d < TARGET_FIT_DMAX; ++d) gl[d] = 0.0;
  double gos = 0.0, gnz = 0.0;
  for (int e = c.tid; e < E; e += c.nthr) {
    int a, b;
    tf_decode(e, a, b);
    const double ge = c.Ap[e];
    if (a == b) {
      gos += ge;
      gnz += ge;
    } else {
      double df2[TARGET_FIT_DMAX];
      double d2 = 0.0;
#pragma unroll
      for (int d = 0; d < TARGET_FIT_DMAX; ++d) {
        df2[d] = 0.0;
        if (d < D) {
          const double df = (c.Xs[a * D + d] - c.Xs[b * D + d]) * c.invl[d];
          df2[d] = df * df;
          d2 += df2[d];
        }
      }
      double kk, dk;
      tf_kernel(c.kind, d2, kk, dk);
      gos += ge * kk;
      const double h = -2.0 * ge * os * dk;   // d d2 / d l_d = -2 df_d^2 / l_d
#pragma unroll
      for (int d = 0; d < TARGET_FIT_DMAX; ++d) gl[d] += h * df2[d];
    }
  }
#pragma unroll
  for (int d = 0; d < TARGET_FIT_DMAX; ++d) {
    if (d < D) {
      const double s = tf_wave_sum(gl[d]);
      if (tf_last_lane(c)) c.part[c.wave * (TARGET_FIT_DMAX + 2) + d] = s;
    }
  }
  {
    const double s1 = tf_wave_sum(gos), s2 = tf_wave_sum(gnz);
    if (tf_last_lane(c)) {
      c.part[c.wave * (TARGET_FIT_DMAX + 2) + TARGET_FIT_DMAX] = s1;
      c.part[c.wave * (TARGET_FIT_DMAX + 2) + TARGET_FIT_DMAX + 1] = s2;
    }
  }
  TF_SYNC();
  for (int q = c.tid; q < D + 2; q += c.nthr) {
    const int slot = q < D ? q : (TARGET_FIT_DMAX + (q - D));
    double s = 0.0;
    for (int v = 0; v < c.nwave; ++v) s += c.part[v * (TARGET_FIT_DMAX + 2) + slot];
    const double th = c.theta[q];
    if (q < D) s *= c.invl[q];   // (the 1 / l_d of d d2 / d l_d)
    const TargetPrior& pr = q < D ? sp.ls_prior : (q == D ? sp.os_prior : sp.nz_prior);
    gz[q] = (s + tf_prior_dlogp(pr, th)) * c.dth[q] * inv_n;
  }
  TF_SYNC();
  TF_STAMP(7);
  return value;
}

// ---- the optimiser: minimise f = -mll over z, w >= w_lower ----------------------------------------------------------------
TF_DEV double tf_dot(const TfCtx& c, const double* a, const double* b) {
  double s = 0.0;
  for (int i = c.tid; i < c.P; i += c.nthr) s += a[i] * b[i];
  return tf_block_sum(c, s);
}

TF_DEV bool tf_finite(double x) { return x - x == 0.0; }

// The whole kernel body as ONE loop around ONE call of tf_eval (a state machine: the evaluation is inlined exactly once, its LDS
// pointers stay 32-bit LDS addresses in registers; three call sites out of line made every LDS access a flat access through a
// context struct in memory):
//   EVAL_ONLY  mode 0: value + gradient at z, done
//   INIT       first evaluation at the start point              -> direction, first trial point
//   TRIAL      evaluation at a line-search trial point          -> accept (curvature pair, stopping rules, new direction) or halve the step
//   FINAL      value (and factorisation status) at the point kept
TF_DEV void tf_run(const TfCtx& c, const TargetFitParams& p, int prob) {
  enum { EVAL_ONLY = 0, INIT = 1, TRIAL = 2, FINAL = 3 };
  const int P = c.P, D = c.D, H = p.history;
  int32_t* info_out = p.info ? p.info + prob : nullptr;
  double* jit_out = p.jitter ? p.jitter + prob : nullptr;
  double* x = p.z + (size_t)prob * P;
  double* ws = p.mode == 0 ? nullptr : p.workspace + (size_t)prob * (6 + 2 * H) * P;
  double *g = ws, *xn = ws + P, *gn = ws + 2 * P, *dvec = ws + 3 * P, *q = ws + 4 * P, *mask = ws + 5 * P;
  double* S = ws + 6 * P;
  double* Y = S + (size_t)H * P;
  double* rho = c.sc + 8;                       // [H]
  double* al = c.sc + 8 + TARGET_FIT_HMAX;      // [H]
  const double lb = p.spec.w_lower;
  const double c1 = 1e-4;
  int state = p.mode == 0 ? EVAL_ONLY : INIT;
  const double* zc = x;
  double* gout = p.mode == 0 ? p.grad + (size_t)prob * P : g;
  if (state == INIT) {   // start inside the box
    for (int i = D + 2 + c.tid; i < P; i += c.nthr) x[i] = fmax(x[i], lb);
    TF_SYNC();
  }
  TfCtx o = c;   // who does the vector bookkeeping between evaluations
#ifndef SCAML_HOST_EMUL
  if (P <= TF_LANES) {
    o.solo = 1;
    o.tid = c.lane;
    o.nthr = TF_LANES;
    o.wave = 0;
    o.nwave = 1;
  }
#endif
  double f = 0.0, t = 1.0, gd = 0.0;
  int n_eval = 0, it = 0, status = 0, hist = 0, head = 0, ls = 0;   // status 0 max_iter, 1 converged (gradient), 2 converged (decrease), 3 line search failed, 4 bad start
  for (;;) {
    const bool report = state != TRIAL;
    const double val = tf_eval(c, p, zc, gout, report ? info_out : nullptr, report ? jit_out : nullptr);
    ++n_eval;
    if (state == EVAL_ONLY || state == FINAL) {
      if (c.tid == 0) {
        p.value[prob] = val;
        if (state == FINAL && p.stats) {
          p.stats[4 * prob + 0] = it;
          p.stats[4 * prob + 1] = n_eval;
          p.stats[4 * prob + 2] = status;
          p.stats[4 * prob + 3] = 0;
        }
      }
      return;
    }
    // ---- the optimiser's bookkeeping: by every thread, or -- P <= 64, a lane per variable -- by wave 0 alone, with no barrier and
    //      every dot product a wave reduction (45 us per iteration of vector operations through barriers became ~5) ----
    if (!o.solo || c.wave == 0) {
      bool go_final = false, new_dir = false;
      if (state == INIT) {
        f = -val;
        for (int i = o.tid; i < P; i += o.nthr) g[i] = -g[i];
        TF_OSYNC(o);
        if (!tf_finite(f)) {
          status = 4;
          go_final = true;
        } else if (p.max_iter < 1) {
          go_final = true;
        } else {
          new_dir = true;
        }
      } else {   // TRIAL: xn, gn hold the trial point and +d mll / dz there
        const double fn = -val;
        double dec = 0.0;
        for (int i = o.tid; i < P; i += o.nthr) dec += g[i] * (xn[i] - x[i]);
        dec = tf_block_sum(o, dec);
        if (tf_finite(fn) && fn <= f + c1 * dec) {
          // curvature pair, then the step
          double sy = 0.0, ss = 0.0, yy = 0.0, pg = 0.0;
          const int slot = head;
          // (the pair lives in the subspace of the variables that were free in this step and still are: a weight that sat on, or
          //  ran into, the bound contributes a projected step and a gradient change that say nothing about the curvature there)
          for (int i = o.tid; i < P; i += o.nthr) {
            const double gi = -gn[i];
            const bool fr = mask[i] != 0.0 && !(i >= D + 2 && xn[i] <= lb);
            const double sv = fr ? xn[i] - x[i] : 0.0, yv = fr ? gi - g[i] : 0.0;
            sy += sv * yv;
            ss += sv * sv;
            yy += yv * yv;
          }
          sy = tf_block_sum(o, sy);
          ss = tf_block_sum(o, ss);
          yy = tf_block_sum(o, yy);
          const bool push = sy > 1e-10 * sqrt(ss) * sqrt(yy);
          for (int i = o.tid; i < P; i += o.nthr) {
            const double gi = -gn[i];
            if (push) {
              const bool fr = mask[i] != 0.0 && !(i >= D + 2 && xn[i] <= lb);
              S[(size_t)slot * P + i] = fr ? xn[i] - x[i] : 0.0;
              Y[(size_t)slot * P + i] = fr ? gi - g[i] : 0.0;
            }
            x[i] = xn[i];
            g[i] = gi;
            // projected gradient (scipy's pgtol test): |P(x - g) - x|
            double st = x[i] - gi;
            if (i >= D + 2) st = fmax(st, lb);
            pg = fmax(pg, fabs(st - x[i]));
          }
          if (push) {
            if (o.tid == 0) rho[slot] = 1.0 / sy;
            head = (head + 1) % H;
            if (hist < H) ++hist;
          }
          pg = tf_block_max(o, pg);
          const double rel = (f - fn) / fmax(fmax(fabs(f), fabs(fn)), 1.0);
          f = fn;
          if (pg <= p.gtol) { status = 1; go_final = true; }
          else if (rel <= p.ftol && it > 1) { status = 2; go_final = true; }
          else if (it >= p.max_iter) { go_final = true; }
          else new_dir = true;
        } else {
          // backtrack: the minimiser of the parabola through f, its slope along the projected step, and the trial value -- kept inside
          // [0.1 t, 0.5 t]; a trial that is not finite just halves
          double tn = 0.5 * t;
          if (tf_finite(fn) && dec < 0.0) {
            const double tq = -0.5 * dec * t / (fn - f - dec);   // slope along the step = dec / t
            if (tq > 0.1 * t && tq < 0.5 * t) tn = tq;
            else if (tq <= 0.1 * t) tn = 0.1 * t;
          }
          t = tn;
          if (++ls >= p.max_ls) { status = 3; go_final = true; }
        }
      }
      if (new_dir) {
        ++it;
        ls = 0;
        // free variables: everything except weights sitting on the bound whose gradient pushes outwards
        for (int i = o.tid; i < P; i += o.nthr) {
          const bool fixed = i >= D + 2 && x[i] <= lb && g[i] > 0.0;
          mask[i] = fixed ? 0.0 : 1.0;
          q[i] = fixed ? 0.0 : g[i];
        }
        TF_OSYNC(o);
        // two-loop recursion, newest pair first (pair h lives in slot (head - 1 - h) mod H)
        for (int h = 0; h < hist; ++h) {
          const int slot = (head - 1 - h + 2 * H) % H;
          const double a = rho[slot] * tf_dot(o, S + (size_t)slot * P, q);
          if (o.tid == 0) al[slot] = a;
          for (int i = o.tid; i < P; i += o.nthr) q[i] -= a * Y[(size_t)slot * P + i];
          TF_OSYNC(o);
        }
        double gamma = 1.0;
        if (hist > 0) {
          const int slot = (head - 1 + H) % H;
          const double yy = tf_dot(o, Y + (size_t)slot * P, Y + (size_t)slot * P);
          if (yy > 0.0) gamma = 1.0 / (rho[slot] * yy);
        }
        for (int i = o.tid; i < P; i += o.nthr) q[i] *= gamma;
        TF_OSYNC(o);
        for (int h = hist - 1; h >= 0; --h) {
          const int slot = (head - 1 - h + 2 * H) % H;
          const double b = rho[slot] * tf_dot(o, Y + (size_t)slot * P, q);
          const double a = al[slot];
          for (int i = o.tid; i < P; i += o.nthr) q[i] += (a - b) * S[(size_t)slot * P + i];
          TF_OSYNC(o);
        }
        for (int i = o.tid; i < P; i += o.nthr) dvec[i] = -q[i] * mask[i];
        TF_OSYNC(o);
        gd = tf_dot(o, g, dvec);
        t = 1.0;
        if (!(gd < 0.0) || hist == 0) {   // first step / not a descent direction: steepest descent, scipy-like first step length
          for (int i = o.tid; i < P; i += o.nthr) dvec[i] = -g[i] * mask[i];
          TF_OSYNC(o);
          gd = tf_dot(o, g, dvec);
          t = fmin(1.0, 1.0 / sqrt(fmax(-gd, 1e-24)));
          if (!(gd < 0.0)) {   // the projected gradient is zero: converged
            status = 1;
            go_final = true;
          }
        }
      }
      if (!go_final) {   // next trial point on the projected ray
        for (int i = o.tid; i < P; i += o.nthr) {
          double v = x[i] + t * dvec[i];
          if (i >= D + 2) v = fmax(v, lb);
          xn[i] = v;
        }
      }
      if (o.tid == 0) c.sc[1] = go_final ? (double)FINAL : (double)TRIAL;
    }
    TF_SYNC();
    if ((int)c.sc[1] == FINAL) {
      state = FINAL;
      zc = x;
      gout = nullptr;
      continue;
    }
    state = TRIAL;
    zc = xn;
    gout = gn;
  }
}

TF_DEV void tf_carve(TfCtx& c, double* lds, int n, int T, int D, int nwave, int mfma) {
  double* q = lds;
  c.nb = (n + 15) / 16;
  c.mfma = mfma;
  if (mfma) {
    // two block triangles of 16 x 17 tiles: L (later the packed G), X = L^-1
    const int tiles = c.nb * (c.nb + 1) / 2;
    c.Lt = q; q += tiles * TF_TS;
    c.Xt = q; q += tiles * TF_TS;
    c.Ap = c.Lt;
    c.Tp = c.Xt;
  } else {
    c.Lt = c.Xt = nullptr;
    c.Ap = q; q += (n + 1) * (n + 2) / 2;
    c.Tp = q; q += n * (n + 1) / 2;
  }
  c.Xs = q; q += n * D;
  c.col = q; q += 2 * (n + 1);
  c.piv = q; q += n;
  c.rs = q; q += n;
  c.alpha = q; q += n;
  c.vv = q; q += n + 16;
  c.w = q; q += T;
  c.w2 = q; q += T;
  c.theta = q; q += D + 2;
  c.dth = q; q += D + 2;
  c.invl = q; q += D;
  c.part = q; q += nwave * (TARGET_FIT_DMAX + 2);
  c.red = q; q += nwave;
  c.sc = q; q += 8 + 2 * TARGET_FIT_HMAX;
}

TF_DEV void tf_main(TfCtx& c, const TargetFitParams& p, int prob) {
  for (int i = c.tid; i < c.n * c.D; i += c.nthr) c.Xs[i] = p.X[i];
  TF_SYNC();
  tf_run(c, p, prob);
}

#ifndef SCAML_HOST_EMUL
#ifndef TF_MAX_THREADS
#define TF_MAX_THREADS 512
#endif
extern "C" __global__ __launch_bounds__(TF_MAX_THREADS) void scaml_target_fit_kernel(TargetFitParams p) {
  extern __shared__ double tf_lds[];
  TfCtx c;
  c.tid = threadIdx.x;
  c.nthr = blockDim.x;
  c.lane = threadIdx.x & 63;
  c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: every row index derived from it is SALU work, off the vector unit)
  c.nwave = blockDim.x >> 6;
  c.n = p.n; c.T = p.T; c.D = p.D; c.P = p.D + 2 + p.T; c.E = p.n * (p.n + 1) / 2; c.kind = p.kind;
  c.solo = 0;
  tf_carve(c, tf_lds, p.n, p.T, p.D, c.nwave, p.use_mfma);
  tf_main(c, p, blockIdx.x);
}
#endif

}  // namespace scaml

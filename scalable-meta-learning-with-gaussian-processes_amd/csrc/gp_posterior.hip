// gp_posterior.hip — batched source-GP posteriors at shared query points, their cross-covariance
// blocks, and the weighted sum over tasks that forms the ScaML-GP target prior (gfx950).
//
// Replaces, for all source tasks at once,
//   scamlgp/model.py:128, 281   posteriors = [gp.posterior(x) for gp in source_gps]
//       (botorch GPyTorchModel.posterior -> gpytorch ExactGP eval / DefaultPredictionStrategy:
//        mu~ = K_* alpha,  Sigma~ = k(x,x) - V^T V with V = L^-1 K_*^T,  then
//        Standardize.untransform_posterior: mu = m + s mu~, Sigma = s^2 Sigma~)
//   scamlgp/model.py:129-135    mean = sum_i w_i mu_i,  cov = sum_i w_i^2 Sigma_i
//
// gp_posterior_kernel: one wave per (task, strip of 16 query points), waves fully independent.
//   Row block kb of V = L^-1 K_*^T:  V_kb = W_kb (K_*^T_kb - sum_{j<kb} L_kb,j V_j), W_kb = L_kk^-1
//   (from the fused fit).  The cross-kernel block is evaluated straight into an MFMA accumulator
//   (C/D layout: col = lane & 15 = query point, row = (lane >> 4) + 4g = training point), every
//   L_kb,j V_j product is 4 v_mfma_f64_16x16x4_f64 (A = L tile from HBM/L2, B = V_j from the wave's
//   LDS strip); the accumulator registers are already in B-operand position for the final W_kb
//   product (row 4m + (lane >> 4) of the tile is register m), so no data moves between the two.
//   mean and variance are reduced on the fly; V can be kept for the covariance kernel.
#include "scaml_common.hpp"
#include "../../include/scaml_gp.h"
#include "gp_posterior_params.h"

namespace scaml {

template <int KIND>
__global__ __launch_bounds__(256) void gp_posterior_kernel(PosteriorParams p) {
  extern __shared__ double lds[];
  const int N = p.N, D = p.D, M = p.M;
  const int NB = (N + 15) / 16, NP = NB * 16;
  const int task = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = SCAML_WAVE_INDEX(tid), nwaves = blockDim.x >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  int n = p.n_points ? p.n_points[task] : N;
  n = n < 0 ? 0 : (n > N ? N : n);
  // LDS: exp table [64] | alpha [NP] | invl [D] (+pad) | xsT [D][NP] (optional) | per wave: xq [D][16], V strip [NP][16]
  double* exptab = lds;
  double* alpha_s = exptab + 64;
  double* invl = alpha_s + NP;
  double* xsT = invl + D + (D & 1);
  double* wbase = xsT + (p.x_in_lds ? D * NP : 0);
  double* xqs = wbase + wave * (16 * D + NP * 16);
  double* Vs = xqs + 16 * D;

  const double* Xg = p.X + (size_t)task * N * D;
  const double* th = p.theta + (size_t)task * (D + 2);
  const double* Lg = p.L + (size_t)task * N * N;
  const double* Wg = p.Linv_diag + (size_t)task * NB * 256;
  const double os = th[D];
  const double ym = p.y_mean ? p.y_mean[task] : 0.0;
  const double ys = p.y_std ? p.y_std[task] : 1.0;

  exp2_table_init(exptab, tid);
  for (int d = tid; d < D; d += blockDim.x) invl[d] = 1.0 / th[d];   // strided: D may exceed the 64 .. 256 threads
  for (int r = tid; r < NP; r += blockDim.x) alpha_s[r] = r < n ? p.alpha[(size_t)task * N + r] : 0.0;
  __syncthreads();
  if (p.x_in_lds) {
    for (int r = tid; r < NP; r += blockDim.x) {
      const bool in = r < n;
      for (int d = 0; d < D; ++d) xsT[d * NP + r] = in ? Xg[(size_t)r * D + d] * invl[d] : 0.0;
    }
  }
  const int strip = blockIdx.x * nwaves + wave;
  const int c0 = strip * 16;
  const int qc = c0 + lc;   // this lane's query point
  if (lq == 0) {
    const double* Xqg = p.Xq + (p.xq_per_task ? (size_t)task * M * D : 0);
    for (int d = 0; d < D; ++d) xqs[d * 16 + lc] = qc < M ? Xqg[(size_t)qc * D + d] * invl[d] : 0.0;
  }
  __syncthreads();
  if (c0 >= M) return;   // (no barrier below: waves are independent from here)

  double mean_part = 0.0, var_part = 0.0;
  for (int kb = 0; kb < NB; ++kb) {
    // cross-kernel block K(X[16kb + row], xq[c]) straight into the accumulator
    d4_t acc;
    {
      double d2[4] = {0.0, 0.0, 0.0, 0.0};
      const int row0 = 16 * kb + lq;
#pragma unroll 2
      for (int d = 0; d < D; ++d) {
        const double xc = xqs[d * 16 + lc];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = row0 + 4 * g;
          const double xr = p.x_in_lds ? xsT[d * NP + row] : (row < n ? Xg[(size_t)row * D + d] * invl[d] : 0.0);
          const double df = xr - xc;
          d2[g] = __builtin_fma(df, df, d2[g]);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = row0 + 4 * g;
        double kv = os * kernel_from_sqdist<KIND>(d2[g], exptab);
        kv = row < n ? kv : 0.0;
        acc[g] = kv;
        mean_part = __builtin_fma(kv, alpha_s[row], mean_part);
      }
    }
    if (p.mean_only) continue;
    // acc -= L[kb, j] V_j for the row blocks already solved
    const int arow = 16 * kb + lc;
    const bool arow_ok = arow < n;
    const double* Lrow = Lg + (size_t)(arow < N ? arow : 0) * N;
    acc = subst_accumulate(acc, Lrow, arow_ok, 16 * kb + 16 <= n, n, (N & 1) == 0, Vs, 0, kb, lc, lq);
    // V_kb = W_kb acc: register m of acc is row 4m + lq of the tile = the B operand of k-step m
    d4_t v = {0.0, 0.0, 0.0, 0.0};
    const double* wrow = Wg + (size_t)kb * 256 + lc * 16 + lq;
#pragma unroll
    for (int m = 0; m < 4; ++m) v = __builtin_amdgcn_mfma_f64_16x16x4f64(wrow[4 * m], acc[m], v, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      var_part = __builtin_fma(v[g], v[g], var_part);
      Vs[(16 * kb + strip_row(lq, g)) * 16 + lc] = v[g];
      const int row = 16 * kb + lq + 4 * g;
      if (p.V && row < N && qc < M) p.V[((size_t)task * N + row) * M + qc] = v[g];
    }
  }
  mean_part += __shfl_xor(mean_part, 16);
  mean_part += __shfl_xor(mean_part, 32);
  var_part += __shfl_xor(var_part, 16);
  var_part += __shfl_xor(var_part, 32);
  if (lq == 0 && qc < M) {
    if (p.mu) p.mu[(size_t)task * M + qc] = __builtin_fma(ys, mean_part, ym);
    if (p.var) p.var[(size_t)task * M + qc] = ys * ys * (os - var_part);
  }
}

// gp_posterior_linv_kernel: the same posteriors from the EXPLICIT inverse factor L^-1 (scaml_linv_batched_f64,
// computed once per fit -- the source GPs of a BO run are fixed while thousands of candidates are scored):
// V = L^-1 K_*^T is then a triangular matrix product with no dependency between its row blocks instead of a
// substitution.  One workgroup (4 waves) per (task, strip of 16 query points): the waves first evaluate the
// cross-kernel strip K_*^T (N x 16) into LDS side by side (VALU work, four row blocks each), then take the
// output row blocks in boustrophedon order (block kb costs kb + 1 products: 0,7,8,15 | 1,6,9,14 | ... is
// perfectly balanced) and stream the rows of L^-1 through the matrix cores (128-byte row segments, one block
// ahead).  LDS is 37-53 KB per workgroup, so three to four of them share a CU and hide each other's
// latencies -- the substitution kernel above holds a 32 KB strip per WAVE and runs one wave per SIMD.
// XCD-aware block map (see gp_mll_grad_kernel): the strips of one task share an L2.
// GRAD (round 3): the same pass with the INPUT GRADIENT of the posterior.  A strip's 16 columns then belong to ONE query point x:
// column 0 is k_*(x) as before, columns 1 .. D are d k_*(x) / d x_d (columns past D are zero), so that V = L^-1 [k_*, dk_*/dx]
// comes out of the same triangular product and, with it,
//   mu[.., 0] = m + s k_* . alpha,       mu[.., 1 + d]  = s (dk_*/dx_d) . alpha                    = d mu / d x_d
//   var[.., 0] = s^2 (os - V_0 . V_0),   var[.., 1 + d] = -2 s^2 V_0 . V_{1+d}                      = d var / d x_d
//   cov[a][.., 0] = s^2 (os k(x_a, x) - VA_a . V_0),  cov[a][.., 1 + d] = s^2 (os dk(x_a, x)/dx_d - VA_a . V_{1+d})
// for the Ma leading points x_a (p.Xa: the target's training inputs) -- everything botorch's optimize_acqf differentiates through
// model.posterior for (HOT LOOP #4; scamlgp/utils.py:215-224).  Outputs are (T, Mq, 16) / (T, Ma, Mq * 16): M = 16 Mq.
template <int KIND, bool COV, bool GRAD>
__global__ __launch_bounds__(512) void gp_posterior_linv_kernel(PosteriorParams p) {
  // eight waves share one K_*^T strip (the row blocks are dealt 8 ways: twice the waves per byte of LDS to hide the L^-1 segment loads)
  constexpr int NW = 8;
  extern __shared__ double lds[];
  const int N = p.N, D = p.D, M = p.M;
  const int NB = (N + 15) / 16, NP = NB * 16;
  const int strips = (M + 15) / 16;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int task = (slot / strips) * 8 + xcd;
  const int strip = slot % strips;
  if (task >= p.T) return;
  // (the wave index stays a vector value here: as a scalar -- SCAML_WAVE_INDEX, which pays in the L^-1 and gradient kernels -- this
  //  kernel measured 3 % slower at C3, 178.9 against 173.6 us)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  int n = p.n_points ? p.n_points[task] : N;
  n = n < 0 ? 0 : (n > N ? N : n);
  // LDS: exp table [64] | alpha [NP] | invl [D4] | xq [D4][16] | red [32] | |xq|^2 [16] | |x|^2 [NP] | K_*^T strip [NP][16];
  // D4 = D rounded up to the MFMA k-step, the extra entries are zeros.  (Round 2: the task's points are no longer staged -- they
  // are MFMA operands read once per block straight from memory --, which keeps four workgroups on a CU instead of two.)
  const int D4 = (D + 3) & ~3;
  double* exptab = lds;
  double* alpha_s = exptab + 64;
  double* invl = alpha_s + NP;
  double* xqs = invl + D4;
  double* red = xqs + 16 * D4;
  double* nq = red + 32;
  double* nr = nq + 16;
  double* Ks = nr + NP;

  const double* Xg = p.X + (size_t)task * N * D;
  const double* th = p.theta + (size_t)task * (D + 2);
  const double* Li = p.L + (size_t)task * N * N;   // L^-1 here
  const double os = th[D];
  const double ym = p.y_mean ? p.y_mean[task] : 0.0;
  const double ys = p.y_std ? p.y_std[task] : 1.0;
  const int qc = 16 * strip + lc;   // this lane's query point

  exp2_table_init(exptab, tid);
  for (int d = tid; d < D4; d += blockDim.x) invl[d] = d < D ? 1.0 / th[d] : 0.0;   // strided: D may exceed the 256 threads
  if (tid < 32) red[tid] = 0.0;
  for (int r = tid; r < NP; r += blockDim.x) alpha_s[r] = r < n ? p.alpha[(size_t)task * N + r] : 0.0;
  __syncthreads();
  for (int r = tid; r < NP; r += blockDim.x) {
    double nrm = 0.0;
    if (r < n) {
      for (int d = 0; d < D; ++d) {
        const double v = Xg[(size_t)r * D + d] * invl[d];
        nrm = __builtin_fma(v, v, nrm);
      }
    }
    nr[r] = nrm;
  }
  if (tid < 16) {
    const double* Xqg = p.Xq + (p.xq_per_task ? (size_t)task * (GRAD ? M / 16 : M) * D : 0);
    const int qpoint = GRAD ? strip : qc;   // GRAD: all 16 columns of the strip are the same point
    double nrm = 0.0;
    for (int d = 0; d < D4; ++d) {
      const double v = (qc < M && d < D) ? Xqg[(size_t)qpoint * D + d] * invl[d] : 0.0;
      xqs[d * 16 + lc] = v;
      nrm = __builtin_fma(v, v, nrm);
    }
    nq[lc] = nrm;
  }
  __syncthreads();

  // ---- phase 1: cross-kernel strip into LDS (rows permuted inside each 16-block, see strip_row) + the mean
  // Round 2: the squared distances of a 16 x 16 block come off the matrix core (expanded form |a|^2 + |c|^2 - 2 a.c: D4/4 k-steps
  // over the coordinates + one over the norms) -- on gfx950 a VALU instruction costs what 1/8 of an fp64 MFMA costs, and nothing
  // overlaps with it (profiles/r02_probe_dp_pipe.txt): 2 D VALU instructions per element became (D4/4 + 1) MFMAs per 256 elements;
  // the kernel function itself in the fused fit's instruction sequence, masks only where a block straddles n.
  double mean_part = 0.0;
  {
    const double kc0 = os, kc1 = 2.2360679774997896964 * os, kc2 = (5.0 / 3.0) * os;
    const double bn = lq == 0 ? 1.0 : (lq == 1 ? nq[lc] : 0.0);
    for (int kb = wave; kb < NB; kb += NW) {
      d4_t d2v = {0.0, 0.0, 0.0, 0.0};
      const int arow = 16 * kb + lc;
      const double* xa = Xg + (size_t)(arow < n ? arow : 0) * D;
      const double* xb = xqs + lq * 16 + lc;
      for (int s4 = 0; s4 < D4; s4 += 4) {
        const int d = s4 + lq;
        const double av = (arow < n && d < D) ? -2.0 * invl[d] * xa[d] : 0.0;   // A[i = lc][k = lq]: point 16 kb + lc, coordinate d
        d2v = __builtin_amdgcn_mfma_f64_16x16x4f64(av, xb[s4 * 16], d2v, 0, 0, 0);
      }
      d2v = __builtin_amdgcn_mfma_f64_16x16x4f64(lq == 0 ? nr[arow] : (lq == 1 ? 1.0 : 0.0), bn, d2v, 0, 0, 0);
      asm volatile("s_nop 15\n\ts_nop 2" : "+v"(d2v));   // (gfx950: the last result pair is not interlocked for VALU reads)
      const int row0 = 16 * kb + lq;
      if (GRAD) {
        // column 0: the kernel value; column c = 1 + d: os dk/d(d2) * 2 (x'_d - a'_d) / l_d with primes = coordinates / l_d
        const int dcol = lc - 1;
        const double xqd = (dcol >= 0 && dcol < D) ? xqs[dcol * 16] : 0.0, ild = (dcol >= 0 && dcol < D) ? invl[dcol] : 0.0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = row0 + 4 * g;
          double kv, dk;
          kernel_and_slope_scaled<KIND>(KIND == 0 ? vmax_f64(d2v[g], 0.0) : d2v[g], kc0, exptab, kv, dk);
          double val = kv;
          if (lc > 0) {
            const double xad = (row < n && dcol < D) ? Xg[(size_t)row * D + dcol] * ild : 0.0;
            val = dcol < D ? 2.0 * dk * (xqd - xad) * ild : 0.0;
          }
          val = row < n ? val : 0.0;
          Ks[(16 * kb + strip_row(lq, g)) * 16 + lc] = val;
          mean_part = __builtin_fma(val, alpha_s[row], mean_part);
        }
      } else if (16 * kb + 16 <= n) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const double kv = kernel_from_sqdist_scaled<KIND>(KIND == 0 ? vmax_f64(d2v[g], 0.0) : d2v[g], kc0, kc1, kc2, exptab);
          Ks[(16 * kb + strip_row(lq, g)) * 16 + lc] = kv;
          mean_part = __builtin_fma(kv, alpha_s[row0 + 4 * g], mean_part);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = row0 + 4 * g;
          double kv = kernel_from_sqdist_scaled<KIND>(KIND == 0 ? vmax_f64(d2v[g], 0.0) : d2v[g], kc0, kc1, kc2, exptab);
          kv = row < n ? kv : 0.0;
          Ks[(16 * kb + strip_row(lq, g)) * 16 + lc] = kv;
          mean_part = __builtin_fma(kv, alpha_s[row], mean_part);
        }
      }
    }
  }
  mean_part = sum_lane_groups(mean_part);
  if (lq == 0) __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)(red + lc), mean_part);
  __syncthreads();

  // ---- phase 2: V = L^-1 K_*^T by row blocks, the variance on the fly
  // (COV: the variant with the fused covariance block -- ~90 more registers, so the plain pass keeps its own instantiation)
  constexpr int MAXAS = COV ? 6 : 1;   // at most 96 leading query points (the target's training points)
  const int nas = (COV && p.VA) ? (p.Ma + 15) / 16 : 0;
  d4_t cacc[MAXAS];
#pragma unroll
  for (int as = 0; as < MAXAS; ++as) cacc[as] = d4_t{0.0, 0.0, 0.0, 0.0};
  if (!p.mean_only) {
    double var_part = 0.0;
    const bool n_even = (N & 1) == 0;
    const double* VAg = p.VA ? p.VA + (size_t)task * N * p.Ma : nullptr;
    for (int it = 0;; ++it) {
      const int kb = (it & 1) ? (it + 1) * NW - 1 - wave : it * NW + wave;
      if (kb >= NB) {
        if (it * NW >= NB) break;
        continue;
      }
      const int arow = 16 * kb + lc;
      const double* Lrow = Li + (size_t)(arow < N ? arow : 0) * N;
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      acc = block_row_accumulate_deep<false>(acc, Lrow, arow < N, 16 * kb + 16 <= N, N, n_even, Ks, 0, kb + 1, lc, lq);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // GRAD: V_0 . V_c instead of V_c . V_c -- the row's column-0 value sits in the lane with lc = 0 of the same lane group
        const double other = GRAD ? __shfl(acc[g], lane & 48) : acc[g];
        var_part = __builtin_fma(acc[g], other, var_part);
        const int row = 16 * kb + lq + 4 * g;
        if (!GRAD && p.V && row < N && qc < M) p.V[((size_t)task * N + row) * M + qc] = row < n ? acc[g] : 0.0;
      }
      if (nas) {
        // cov tiles: (VA^T V)[a][c] += sum over the 16 rows of this block -- A = VA[rows][16 as + lc] transposed by the
        // operand layout (A[i = lc][k = lq + 4 m] = VA[16 kb + lq + 4 m][16 as + lc]: 128-byte row segments), B = the
        // V block just computed, straight from its accumulator registers (C/D layout = B-operand layout)
        d4_t vb = acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) vb[g] = (16 * kb + lq + 4 * g) < n ? vb[g] : 0.0;
        // (the VA operands are fetched strip by strip: holding all six strips' worth across the products above cost 48 registers
        //  and with them half of the waves a CU can keep in flight)
#pragma unroll
        for (int as = 0; as < MAXAS; ++as) {
          if (as < nas) {
            const int ac = 16 * as + lc;
            double va[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const int row = 16 * kb + lq + 4 * m;
              va[m] = (row < n && ac < p.Ma) ? VAg[(size_t)row * p.Ma + ac] : 0.0;
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) cacc[as] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[m], vb[m], cacc[as], 0, 0, 0);
          }
        }
      }
    }
    var_part = sum_lane_groups(var_part);
    if (lq == 0) __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)(red + 16 + lc), var_part);
    __syncthreads();
  }
  if (tid < 16 && qc < M) {
    if (GRAD && tid > 0) {
      if (p.mu) p.mu[(size_t)task * M + qc] = ys * red[tid];
      if (p.var) p.var[(size_t)task * M + qc] = -2.0 * ys * ys * red[16 + tid];
    } else {
      if (p.mu) p.mu[(size_t)task * M + qc] = __builtin_fma(ys, red[tid], ym);
      if (p.var) p.var[(size_t)task * M + qc] = ys * ys * (os - red[16 + tid]);
    }
  }
  if (nas && !p.mean_only) {
    // the four waves' partial tiles are added in a fixed order in the (now free) K_*^T strip, then
    // cov[a][c] = s^2 (os k(xq_a, xq_c) - (VA^T V)[a][c]) for the Ma leading query points a and this strip's 16 points c
    double* cb = Ks;   // [nas][256] register images; nas * 256 <= NP * 16 is checked by the host (Ma <= N)
    for (int w = 0; w < NW; ++w) {
      if (wave == w) {
#pragma unroll
        for (int as = 0; as < MAXAS; ++as) {
          if (as < nas) {
            d4_t v = cacc[as];
            {   // gfx950: the last result pair of an fp64 MFMA is not interlocked for VALU / LDS reads -- settle first
              asm volatile("s_nop 15\n\ts_nop 2" : "+v"(v));
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              double* dst = cb + as * 256 + g * 64 + lane;
              *dst = (w == 0 ? 0.0 : *dst) + v[g];
            }
          }
        }
      }
      __syncthreads();
    }
    // the Ma leading points: the head of the query list (the caller put the target's training points first), or p.Xa
    const double* Xqg = p.Xa ? p.Xa : p.Xq + (p.xq_per_task ? (size_t)task * M * D : 0);
    for (int e = tid; e < nas * 256; e += blockDim.x) {
      const int as = e >> 8, g = (e >> 6) & 3, ln = e & 63;
      const int a = 16 * as + (ln >> 4) + 4 * g, c = ln & 15, qcc = 16 * strip + c;
      if (a < p.Ma && qcc < M) {
        double d2 = 0.0;
        for (int d = 0; d < D; ++d) {
          const double df = Xqg[(size_t)a * D + d] * invl[d] - xqs[d * 16 + c];
          d2 = __builtin_fma(df, df, d2);
        }
        double kv;
        if (GRAD && c > 0) {
          kv = 0.0;
          if (c - 1 < D) {
            double k0, dk;
            kernel_and_slope_scaled<KIND>(d2, os, exptab, k0, dk);
            kv = 2.0 * dk * (xqs[(c - 1) * 16 + c] - Xqg[(size_t)a * D + c - 1] * invl[c - 1]) * invl[c - 1];
          }
        } else {
          kv = os * kernel_from_sqdist<KIND>(d2, exptab);
        }
        p.cov[((size_t)task * p.Ma + a) * M + qcc] = ys * ys * (kv - cb[e]);
      }
    }
  }
}

// cov[t][a][c] = s_t^2 (os k(xq_a, xq_c) - sum_i V[t][i][a] V[t][i][c]) for a < Ma, c < M:
// one wave per 16x16 output tile, contraction over the N training points on the matrix cores.
template <int KIND>
__global__ __launch_bounds__(256) void gp_posterior_cov_kernel(PosteriorCovParams p) {
  __shared__ double exptab[64];
  const int N = p.N, D = p.D, M = p.M, Ma = p.Ma;
  const int task = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = SCAML_WAVE_INDEX(tid);
  const int lc = lane & 15, lq = lane >> 4;
  exp2_table_init(exptab, tid);
  __syncthreads();
  const int ta = blockIdx.y, tc = blockIdx.x * (blockDim.x >> 6) + wave;
  if (16 * tc >= M) return;
  const double* th = p.theta + (size_t)task * (D + 2);
  const double os = th[D];
  const double ys = p.y_std ? p.y_std[task] : 1.0;
  const double* Vg = p.V + (size_t)task * N * M;
  const double* Xqg = p.Xq + (p.xq_per_task ? (size_t)task * M * D : 0);
  // -V_a^T V_c: A operand lane (i = lc -> point a, k = lq -> training row), B operand lane (k = lq, j = lc -> point c)
  const int pa = 16 * ta + lc, pc = 16 * tc + lc;
  d4_t acc = {0.0, 0.0, 0.0, 0.0};
  for (int i0 = 0; i0 < N; i0 += 4) {
    const int i = i0 + lq;
    const double a = (i < N && pa < Ma) ? Vg[(size_t)i * M + pa] : 0.0;
    const double b = (i < N && pc < M) ? Vg[(size_t)i * M + pc] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 1);
  }
  // + os k(xq_a, xq_c) for the elements this lane owns: rows (points a) lq + 4g, column (point c) lc
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int a = 16 * ta + lq + 4 * g;
    if (a < Ma && pc < M) {
      double d2 = 0.0;
      for (int d = 0; d < D; ++d) {
        const double df = (Xqg[(size_t)a * D + d] - Xqg[(size_t)pc * D + d]) / th[d];
        d2 = __builtin_fma(df, df, d2);
      }
      const double kv = os * kernel_from_sqdist<KIND>(d2, exptab);
      p.cov[((size_t)task * Ma + a) * M + pc] = ys * ys * (kv + acc[g]);
    }
  }
}

}  // namespace scaml

// out[e] = sum_t coef(t) * in[t][e], coef = w_t (power 1) or w_t^2 (power 2), skipping masked tasks.
// A workgroup of 256 threads takes 64 consecutive elements; wave q sums the q-th quarter of the tasks (eight tasks' values in flight per
// thread), the four partial sums meet in LDS in a fixed order: 4 x the waves in flight of one-thread-per-element (the pass is a pure
// stream of T len doubles: 23 MB at configs[4] took 17 us with one load in flight per thread, ~5 us now), deterministic.
extern "C" __global__ __launch_bounds__(256) void scaml_weighted_task_sum_kernel(const double* __restrict__ in, const double* __restrict__ w,
                                                                              const uint8_t* __restrict__ active, int T, long long len, int power,
                                                                              double* __restrict__ out) {
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long long e = (long long)blockIdx.x * 64 + lane;
  const int chunk = (T + 3) / 4, t_lo = q * chunk, t_hi = t_lo + chunk < T ? t_lo + chunk : T;
  double s = 0.0;
  if (e < len) {
    for (int t0 = t_lo; t0 < t_hi; t0 += 8) {
      double v[8], c[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int t = t0 + k;
        const bool ok = t < t_hi && (!active || active[t]);   // (a masked task is skipped, not multiplied by zero: its values may be NaN)
        v[k] = ok ? in[(size_t)t * len + e] : 0.0;
        const double wt = ok ? w[t] : 0.0;
        c[k] = power == 2 ? wt * wt : wt;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) s = __builtin_fma(c[k], v[k], s);
    }
  }
  part[q][lane] = s;
  __syncthreads();
  if (q == 0 && e < len) out[e] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// ---- target GP (scamlgp/model.py:359-384 eval branch + gpytorch's exact prediction) ------------------
// (1) assemble: the standardised joint prior of the target GP over cat(train_X (n), Xq (M)) from the weighted source
// sums: K[i][c] = cov_s[i][c] / s^2 + os_t k_t(x_i, x_c) for the n training rows i and all n + M columns c -- the n x n
// block (+ noise on its diagonal) goes to Knn, the cross block to Knq; resid = y~ - (mean_s - m) / s on the training
// points, mean_q / var_q the standardised prior mean and variance (+ os_t) at the queries.  One element per thread.
template <int KIND>
__global__ void scaml_target_assemble_kernel(scaml::TargetAssembleParams p) {
  __shared__ double exptab[64];
  scaml::exp2_table_init(exptab, threadIdx.x);
  __syncthreads();
  const int n = p.n, M = p.M, D = p.D, W = n + M;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const double s2 = p.s_all * p.s_all;
  const double os = p.theta[D], noise = p.theta[D + 1];
  if (e < (long long)n * W) {
    const int i = (int)(e / W), c = (int)(e - (long long)i * W);
    double d2 = 0.0;
    for (int d = 0; d < D; ++d) {
      const double df = (p.Xall[(size_t)i * D + d] - p.Xall[(size_t)c * D + d]) / p.theta[d];
      d2 = __builtin_fma(df, df, d2);
    }
    const double v = p.cov_s[e] / s2 + os * scaml::kernel_from_sqdist<KIND>(d2, exptab);
    if (c < n) p.Knn[(size_t)i * n + c] = v + (i == c ? noise : 0.0);
    else p.Knq[(size_t)i * M + (c - n)] = v;
  }
  if (e < n) p.resid[e] = p.train_targets[e] - (p.mean_s[e] - p.m_all) / p.s_all;
  if (e < M) {
    p.mean_q[e] = (p.mean_s[n + e] - p.m_all) / p.s_all;
    p.var_q[e] = p.var_s[n + e] / s2 + os;
  }
}
template __global__ void scaml_target_assemble_kernel<0>(scaml::TargetAssembleParams);
template __global__ void scaml_target_assemble_kernel<1>(scaml::TargetAssembleParams);

// (2) finish: mu*_q = mean_q + Knq[:, q] . alpha, var*_q = var_q - Knq[:, q] . Z[:, q] with alpha = Knn^-1 resid and
// Z = Knn^-1 Knq from the batched Cholesky / solve kernels (T = 1), un-standardised: m + s mu*, s^2 (var* + noise_add).
extern "C" __global__ void scaml_target_finish_kernel(const double* __restrict__ Knq, const double* __restrict__ Z,
                                                      const double* __restrict__ alpha, const double* __restrict__ mean_q,
                                                      const double* __restrict__ var_q, double m_all, double s_all,
                                                      double noise_add, const int32_t* __restrict__ info, int n, int M,
                                                      double* __restrict__ mu, double* __restrict__ var) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= M) return;
  if (info && info[0] > 0) {   // not positive definite even with jitter: NaN, like the exception the reference would raise
    mu[q] = var[q] = __builtin_nan("");
    return;
  }
  double a = mean_q[q], v = var_q[q];
  // (eight rows' loads in flight: one at a time, 2 n dependent round trips made this 26 us at n = 80)
  for (int i0 = 0; i0 < n; i0 += 8) {
    double k[8], z[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool ok = i0 + u < n;
      k[u] = ok ? Knq[(size_t)(i0 + u) * M + q] : 0.0;
      z[u] = ok ? Z[(size_t)(i0 + u) * M + q] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a = __builtin_fma(k[u], i0 + u < n ? alpha[i0 + u] : 0.0, a);
      v = __builtin_fma(-k[u], z[u], v);
    }
  }
  mu[q] = __builtin_fma(s_all, a, m_all);
  var[q] = s_all * s_all * (v + noise_add);
}

// Input gradient of the target posterior (original units) at Mq query points, on top of the weighted sums of the GRAD pass:
//   d mu* / d x_d  = s ( mu_g[q][1 + d] / s + sum_a alpha_a  dk_nq[a][q][d] )
//   d var* / d x_d = s^2 ( var_g[q][1 + d] / s^2 - 2 sum_a Z[a][q] dk_nq[a][q][d] )
//   dk_nq[a][q][d] = cov_g[a][q][1 + d] / s^2 + d/dx_d os_t k_t(x_a, x_q)
// (ExactGP prediction mu* = mean_q + k_nq . alpha, var* = var_q - k_nq . Knn^-1 k_nq, SURVEY A10, differentiated in x_q; alpha and
// Z = Knn^-1 Knq come from the value path).  One wave per query point, a lane per training point a: the kernel slope at (x_a, x_q) is
// evaluated ONCE and serves all D dimensions (a thread per (query, dimension) looping over the training points -- n dependent
// evaluations each -- took 69 us for 10 points at n = 80), 2 D wave reductions at the end.
template <int KIND>
__global__ __launch_bounds__(64) void scaml_target_grad_kernel(const double* __restrict__ cov_g, const double* __restrict__ mu_g,
                                                               const double* __restrict__ var_g, const double* __restrict__ Xt,
                                                               const double* __restrict__ Xq, const double* __restrict__ theta,
                                                               const double* __restrict__ alpha, const double* __restrict__ Z, double s_all,
                                                               const int32_t* __restrict__ info, int n, int Mq, int D,
                                                               double* __restrict__ dmu, double* __restrict__ dvar) {
  constexpr int DM = 16;   // D <= 15 (the GRAD pass's column budget)
  __shared__ double exptab[64];
  const int q = blockIdx.x, lane = threadIdx.x;
  scaml::exp2_table_init(exptab, lane);
  __syncthreads();
  if (info && info[0] > 0) {
    if (lane < D) dmu[(size_t)q * D + lane] = dvar[(size_t)q * D + lane] = __builtin_nan("");
    return;
  }
  const double os = theta[D], inv_s2 = 1.0 / (s_all * s_all);
  const size_t W = (size_t)Mq * 16, col0 = (size_t)q * 16 + 1;
  double gm[DM], gv[DM];
#pragma unroll
  for (int d = 0; d < DM; ++d) gm[d] = gv[d] = 0.0;
  for (int a = lane; a < n; a += 64) {
    double df[DM];
    double d2 = 0.0;
#pragma unroll
    for (int d = 0; d < DM; ++d) {
      df[d] = 0.0;
      if (d < D) {
        const double il = 1.0 / theta[d];
        df[d] = (Xq[(size_t)q * D + d] - Xt[(size_t)a * D + d]) * il;
        d2 = __builtin_fma(df[d], df[d], d2);
        df[d] *= il;   // (x_q - x_a)_d / l_d^2
      }
    }
    double k0, dk;
    scaml::kernel_and_slope_scaled<KIND>(d2, os, exptab, k0, dk);
    const double al = alpha[a], z = Z[(size_t)a * Mq + q];
    const double* cg = cov_g + (size_t)a * W + col0;
#pragma unroll
    for (int d = 0; d < DM; ++d) {
      if (d < D) {
        const double dkn = __builtin_fma(cg[d], inv_s2, 2.0 * dk * df[d]);
        gm[d] = __builtin_fma(al, dkn, gm[d]);
        gv[d] = __builtin_fma(z, dkn, gv[d]);
      }
    }
  }
#pragma unroll
  for (int d = 0; d < DM; ++d) {
    if (d < D) {
      const double sm = scaml::wave_sum_to_lane15(gm[d]), sv = scaml::wave_sum_to_lane15(gv[d]);
      if (lane == 63) {
        dmu[(size_t)q * D + d] = mu_g[col0 + d] + s_all * sm;
        dvar[(size_t)q * D + d] = var_g[col0 + d] - 2.0 * s_all * s_all * sv;
      }
    }
  }
}
template __global__ void scaml_target_grad_kernel<0>(const double*, const double*, const double*, const double*, const double*, const double*,
                                                      const double*, const double*, double, const int32_t*, int, int, int, double*, double*);
template __global__ void scaml_target_grad_kernel<1>(const double*, const double*, const double*, const double*, const double*, const double*,
                                                      const double*, const double*, double, const int32_t*, int, int, int, double*, double*);

template __global__ void scaml::gp_posterior_kernel<0>(scaml::PosteriorParams);
template __global__ void scaml::gp_posterior_kernel<1>(scaml::PosteriorParams);
template __global__ void scaml::gp_posterior_cov_kernel<0>(scaml::PosteriorCovParams);
template __global__ void scaml::gp_posterior_cov_kernel<1>(scaml::PosteriorCovParams);

// (1) stand-alone kernel matrix: K[t][i][j] = os k(x1_i / l, x2_j / l) (+ noise on the diagonal of the
// square case), one element per thread, rows of x2 coalesced across lanes.
template <int KIND>
__global__ void gp_kernel_matrix_kernel(scaml::KernelMatrixParams p) {
  __shared__ double exptab[64];
  scaml::exp2_table_init(exptab, threadIdx.x);
  __syncthreads();
  const int task = blockIdx.z;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= p.N2) return;
  const int D = p.D;
  const double* th = p.theta + (size_t)task * (D + 2);
  const double* x1 = p.X1 + ((size_t)task * p.N1 + i) * D;
  const double* x2 = (p.X2 ? p.X2 + ((size_t)(p.x2_shared ? 0 : task) * p.N2 + j) * D : p.X1 + ((size_t)task * p.N1 + j) * D);
  double d2 = 0.0;
  for (int d = 0; d < D; ++d) {
    const double df = (x1[d] - x2[d]) / th[d];
    d2 = __builtin_fma(df, df, d2);
  }
  double kv = th[D] * scaml::kernel_from_sqdist<KIND>(d2, exptab);
  if (!p.X2 && p.add_noise && i == j) kv += th[D + 1];
  p.K[((size_t)task * p.N1 + i) * p.N2 + j] = kv;
}
template __global__ void gp_kernel_matrix_kernel<0>(scaml::KernelMatrixParams);
template __global__ void gp_kernel_matrix_kernel<1>(scaml::KernelMatrixParams);
template __global__ void scaml::gp_posterior_linv_kernel<0, false, false>(scaml::PosteriorParams);
template __global__ void scaml::gp_posterior_linv_kernel<1, false, false>(scaml::PosteriorParams);
template __global__ void scaml::gp_posterior_linv_kernel<0, true, false>(scaml::PosteriorParams);
template __global__ void scaml::gp_posterior_linv_kernel<1, true, false>(scaml::PosteriorParams);
template __global__ void scaml::gp_posterior_linv_kernel<0, true, true>(scaml::PosteriorParams);
template __global__ void scaml::gp_posterior_linv_kernel<1, true, true>(scaml::PosteriorParams);

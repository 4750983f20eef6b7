// gp_mll_grad.hip — analytic gradient of the (un-priored) marginal log-likelihood w.r.t. the
// constrained hyper-parameters [lengthscales, outputscale, noise] for a stack of tasks (gfx950).
//
// Replaces the autograd pass through kernel -> Cholesky -> solve that botorch's
// fit_gpytorch_mll runs per L-BFGS-B iteration (scamlgp/utils.py:175, 190):
//   mll = -(y^T K^-1 y + log|K| + n log 2pi) / (2n)
//   d mll / d theta_p = (1 / 2n) sum_ij G_ij dK_ij/dtheta_p,   G = alpha alpha^T - K^-1.
// Two kernels: (1) Linv = L^-1 by the same blocked MFMA substitution as the posterior kernel with an
// identity right-hand side (only row blocks at or below the strip are touched); (2) per 16x16 tile
// of the lower triangle: K^-1 tile = Linv^T Linv on the matrix cores, then the lanes re-evaluate the
// kernel derivatives for the elements they own and reduce G o dK into D + 2 partial sums per tile
// (off-diagonal tiles count twice).  Partial sums go to a (T, tiles, D+2) buffer that the caller
// adds up (deterministic, no atomics).
#include "scaml_common.hpp"
#include "../../include/scaml_gp.h"
#include "gp_posterior_params.h"

namespace scaml {

__global__ __launch_bounds__(256) void gp_linv_kernel(LinvParams p) {
  extern __shared__ double lds[];
  const int N = p.N, NB = (N + 15) / 16, NP = NB * 16;
  const int task = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = SCAML_WAVE_INDEX(tid), nwaves = blockDim.x >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  int n = p.n_points ? p.n_points[task] : N;
  n = n < 0 ? 0 : (n > N ? N : n);
  double* Vs = lds + (size_t)wave * NP * 16;
  const double* Lg = p.L + (size_t)task * N * N;
  const double* Wg = p.Linv_diag + (size_t)task * NB * 256;
  double* Og = p.Linv + (size_t)task * N * N;
  // Strip s costs (NB - s)(NB - s - 1) / 2 block products: the V = gridDim.x * nwaves (virtual) waves of a task
  // take strips in boustrophedon order (v, 2V-1-v, 2V+v, 4V-1-v, ...), which evens out their loads.
  const int vwave = blockIdx.x * nwaves + wave, V = gridDim.x * nwaves;
  for (int it = 0;; ++it) {
  const int strip = (it & 1) ? (it + 1) * V - 1 - vwave : it * V + vwave;
  if (strip >= NB) {
    if (it * V >= NB) break;
    continue;
  }
  const int qc = 16 * strip + lc;  // column of the inverse owned by this lane
  // rows above the strip are zero (half of the dense matrix: 67 MB of the 134 MB this kernel writes at T = 256, N = 256 -- skipped
  // for callers that only ever read block rows at or below the diagonal block, as the posterior kernels do)
  if (!p.lower_only) {
    for (int r = lq; r < 16 * strip && r < N; r += 4)
      if (qc < N) Og[(size_t)r * N + qc] = 0.0;
  }
  // acc -= sum_{strip <= j < kb} L[kb, j] V_j.  One workgroup per task at T >= 256 means ONE wave per SIMD: nobody hides this wave's load
  // latency, so the L row segments are fetched EIGHT tiles ahead in eight buffers that rotate in place over the FLAT sequence of (block
  // row, tile) pairs of the strip -- across the end of a block row into the first tiles of the next -- and W_kb one block ahead.
  // Round 3: unconditional 16-byte BUFFER loads (rows past n_t and the end of the sequence get an offset beyond the descriptor's end
  // and load as zeros) and no copy of the buffers: before, the four segments were copied to a second set at the top of every group of
  // four products, and the compiler waited there for the loads it had issued one group earlier -- a product is 256 cycles, an L2 round
  // trip 1,000-2,000.
  typedef unsigned lv_u4 __attribute__((ext_vector_type(4)));
  typedef double lv_d2 __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Lg), 0, n * N * 8, 0x00020000);
  int ckb = strip + 1, cj = strip;   // load cursor
  auto next_load = [&]() {
    const unsigned off = ckb < NB ? (unsigned)(((size_t)(16 * ckb + lc) * N + 16 * cj + 4 * lq) * 8) : 0xC0000000u;
    const lv_d2 x = __builtin_bit_cast(lv_d2, __builtin_amdgcn_raw_buffer_load_b128(rsL, off, 0, 0));
    const lv_d2 y = __builtin_bit_cast(lv_d2, __builtin_amdgcn_raw_buffer_load_b128(rsL, off + 16, 0, 0));
    if (++cj >= ckb) {
      ++ckb;
      cj = strip;
    }
    LRowSeg t;
    t.a[0] = x[0]; t.a[1] = x[1]; t.a[2] = y[0]; t.a[3] = y[1];
    return t;
  };
  double wn[4];
  auto fetch_w = [&](int kbn) {
#pragma unroll
    for (int m = 0; m < 4; ++m) wn[m] = kbn < NB ? Wg[(size_t)kbn * 256 + lc * 16 + lq + 4 * m] : 0.0;
  };
  d4_t acc;
  double wc[4];
  auto finish_row = [&](int kb) {   // V_kb = W_kb acc -> the wave's strip in LDS (operand position) and the output
    d4_t v = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int m = 0; m < 4; ++m) v = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[m], acc[m], v, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = 16 * kb + lq + 4 * g;
      Vs[(16 * kb + strip_row(lq, g)) * 16 + lc] = v[g];
      if (row < N && qc < N) Og[(size_t)row * N + qc] = v[g];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) { acc[g] = 0.0; wc[g] = wn[g]; }
    fetch_w(kb + 2);
  };
  fetch_w(strip);
#pragma unroll
  for (int g = 0; g < 4; ++g) { acc[g] = (lc == lq + 4 * g) ? 1.0 : 0.0; wc[g] = wn[g]; }
  fetch_w(strip + 1);
  LRowSeg q0 = next_load(), q1 = next_load(), q2 = next_load(), q3 = next_load(), q4 = next_load(), q5 = next_load(), q6 = next_load(), q7 = next_load();
  finish_row(strip);   // (block row `strip`: no products, V = W_strip)
  {
    int kb = strip + 1, j = strip;
    auto use = [&](const LRowSeg& c) {
      const double* vb = Vs + (16 * j + lq) * 16 + lc;
#pragma unroll
      for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(c.a[m], vb[4 * m * 16], acc, 0, 0, 1);
    };
#define SCAML_LINV_STEP(q)        \
    use(q);                       \
    q = next_load();              \
    if (++j == kb) {              \
      finish_row(kb);             \
      ++kb;                       \
      j = strip;                  \
      if (kb >= NB) break;        \
    }
    while (kb < NB) {
      SCAML_LINV_STEP(q0)
      SCAML_LINV_STEP(q1)
      SCAML_LINV_STEP(q2)
      SCAML_LINV_STEP(q3)
      SCAML_LINV_STEP(q4)
      SCAML_LINV_STEP(q5)
      SCAML_LINV_STEP(q6)
      SCAML_LINV_STEP(q7)
    }
#undef SCAML_LINV_STEP
  }
  }   // strips of this wave
}

// X = (L L^T)^-1 B for R right-hand-side columns per task: one wave per (task, strip of 16 columns).
// Forward: Y_kb = W_kb (B_kb - sum_{j<kb} L_kb,j Y_j); backward: X_kb = W_kb^T (Y_kb - sum_{j>kb} L_j,kb^T X_j).
// Every block product is 4 MFMAs; solved blocks live in the wave's LDS strip in B-operand position.
__global__ __launch_bounds__(256) void gp_cho_solve_kernel(ChoSolveParams p) {
  extern __shared__ double lds[];
  const int N = p.N, R = p.R, NB = (N + 15) / 16, NP = NB * 16;
  const int task = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = SCAML_WAVE_INDEX(tid), nwaves = blockDim.x >> 6;
  const int lc = lane & 15, lq = lane >> 4;
  int n = p.n_points ? p.n_points[task] : N;
  n = n < 0 ? 0 : (n > N ? N : n);
  double* Vs = lds + (size_t)wave * NP * 16;
  const double* Lg = p.L + (size_t)task * N * N;
  const double* Wg = p.Linv_diag + (size_t)task * NB * 256;
  const double* Bg = p.B + (size_t)task * N * R;
  double* Og = p.Xout + (size_t)task * N * R;
  const int strip = blockIdx.x * nwaves + wave;
  if (16 * strip >= R) return;
  const int qc = 16 * strip + lc;
  // forward substitution
  LRowSeg q[4];   // the next four L tiles (row segments) of the substitution, in flight
  auto fetch4 = [&](int kbr, int jf) {   // tiles jf .. jf + 3 of block row kbr (zeros past the diagonal / the matrix)
    const int arow = 16 * kbr + lc;
    const bool arow_ok = kbr < NB && arow < n;
    const double* Lrow = Lg + (size_t)(arow < N ? arow : 0) * N;
    const bool rows_in = 16 * kbr + 16 <= n;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = jf + u;
      if (kbr < NB && j < kbr) {
        q[u] = load_lrow_seg(Lrow, 16 * j + 4 * lq, rows_in && (N & 1) == 0 && 16 * j + 16 <= n, arow_ok, n);
      } else {
        q[u].a[0] = q[u].a[1] = q[u].a[2] = q[u].a[3] = 0.0;
      }
    }
  };
  // (W_kb and the right-hand side rows of a block are fetched one block ahead as well)
  double wn[4], bn[4];
  auto fetch_wb = [&](int kbn) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      wn[m] = kbn < NB ? Wg[(size_t)kbn * 256 + lc * 16 + lq + 4 * m] : 0.0;
      const int row = 16 * kbn + lq + 4 * m;
      bn[m] = (kbn < NB && row < n && qc < R) ? Bg[(size_t)row * R + qc] : 0.0;
    }
  };
  if (p.mode == 1) {
    // backward substitution only: the right-hand side goes into the strip as it is
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = 16 * kb + lq + 4 * g;
        Vs[(16 * kb + strip_row(lq, g)) * 16 + lc] = (row < n && qc < R) ? Bg[(size_t)row * R + qc] : 0.0;
      }
    }
  }
  fetch_wb(0);
  for (int kb = 0; kb < (p.mode == 1 ? 0 : NB); ++kb) {
    d4_t acc;
    double wc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { acc[g] = bn[g]; wc[g] = wn[g]; }
    fetch_wb(kb + 1);
    // acc -= sum_{j < kb} L[kb, j] V_j.  A wave works alone on its strip here (R <= 16 right-hand sides: ONE wave
    // per task), so the L row segments are fetched four tiles ahead -- across the end of a block row into the
    // first tiles of the next one -- instead of one (subst_accumulate): every tile product used to expose most
    // of a global-load latency.
    for (int j0 = 0; j0 < kb; j0 += 4) {
      LRowSeg cur[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) cur[u] = q[u];
      if (j0 + 4 < kb) fetch4(kb, j0 + 4); else fetch4(kb + 1, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u;
        if (j < kb) {
          const double* vb = Vs + (16 * j + lq) * 16 + lc;
#pragma unroll
          for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[u].a[m], vb[4 * m * 16], acc, 0, 0, 1);
        }
      }
    }
    if (kb == 0) fetch4(1, 0);
    d4_t v = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int m = 0; m < 4; ++m) v = __builtin_amdgcn_mfma_f64_16x16x4f64(wc[m], acc[m], v, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) Vs[(16 * kb + strip_row(lq, g)) * 16 + lc] = v[g];
  }
  // backward substitution (in place in the strip: block kb is overwritten once all j > kb are final)
  double un[4];   // A[i = lc][k = lq] = W[4 m + lq][lc] of the next block, one block ahead
  auto fetch_wt = [&](int kbn) {
#pragma unroll
    for (int m = 0; m < 4; ++m) un[m] = kbn >= 0 ? Wg[(size_t)kbn * 256 + lq * 16 + lc + 4 * m * 16] : 0.0;
  };
  fetch_wt(NB - 1);
  for (int kb = NB - 1; kb >= 0; --kb) {
    d4_t acc;
    double uc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { acc[g] = Vs[(16 * kb + strip_row(lq, g)) * 16 + lc]; uc[g] = un[g]; }
    fetch_wt(kb - 1);
    // A[i = lc][k] = L[16 j + k][16 kb + lc] (the transposed tile), k = 4 lq + m as in the strip's row order.
    // The tiles of block column kb are known in advance: four of them (16 values per lane) are in flight while
    // the MFMAs of the previous four run -- with the load issued right in front of its MFMA every one of the
    // NB (NB - 1) / 2 tile products exposed a full global-load latency.
    auto load_tile = [&](int j, double (&a)[4]) {
      const int c = 16 * kb + lc;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int r = 16 * j + 4 * lq + m;
        a[m] = (j < NB && r < n && c < n) ? Lg[(size_t)r * N + c] : 0.0;
      }
    };
    double nx[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) load_tile(kb + 1 + u, nx[u]);
    for (int j0 = kb + 1; j0 < NB; j0 += 4) {
      double cur[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int m = 0; m < 4; ++m) cur[u][m] = nx[u][m];
#pragma unroll
      for (int u = 0; u < 4; ++u) load_tile(j0 + 4 + u, nx[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u;
        if (j < NB) {
          const double* vb = Vs + (16 * j + lq) * 16 + lc;
#pragma unroll
          for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[u][m], vb[4 * m * 16], acc, 0, 0, 1);
        }
      }
    }
    d4_t v = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int m = 0; m < 4; ++m) v = __builtin_amdgcn_mfma_f64_16x16x4f64(uc[m], acc[m], v, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = 16 * kb + lq + 4 * g;
      Vs[(16 * kb + strip_row(lq, g)) * 16 + lc] = v[g];
      if (row < N && qc < R) Og[(size_t)row * R + qc] = row < n ? v[g] : 0.0;
    }
  }
}

// kernel value k (without outputscale) and h with dk/dl_d = h * delta_d^2 / l_d^3
template <int KIND>
__device__ __forceinline__ void kernel_and_dfactor(double d2, const double* exptab, double& k, double& h) {
  if (KIND == 0) {
    k = exp_neg(-0.5 * d2, exptab);
    h = k;
  } else {
    double dd = d2 < 1e-30 ? 1e-30 : d2;
    dd = dd > 1e30 ? 1e30 : dd;
    const double r = sqrt_from_rinv(dd, rsqrt_seeded(dd));
    const double s5 = 2.2360679774997896964;
    const double e = exp_neg(-s5 * r, exptab);
    k = __builtin_fma(__builtin_fma(r, 5.0 / 3.0, s5), r, 1.0) * e;
    h = (5.0 / 3.0) * __builtin_fma(s5, r, 1.0) * e;
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void gp_mll_grad_kernel(MllGradParams p) {
  __shared__ double exptab[64];
  __shared__ double invl_s[64];   // 1 / lengthscale (D <= 64: larger D take the division)
  extern __shared__ double xstage[];   // [waves][64][D | 1] scaled points of the four blocks of a super-tile
  const int N = p.N, D = p.D, NB = (N + 15) / 16;
  const int NT = NB * (NB + 1) / 2;
  // XCD-aware block -> (task, tile group) map.  Workgroups go to the 8 XCDs round-robin by linear id, and
  // each XCD has its own 4 MB L2: all tile groups of one task are therefore given ids that are congruent
  // mod 8, so that the task's L^-1 (re-read by every tile of its column / row) stays in ONE L2 instead of
  // being fetched from HBM by all eight.  1-D grid of ceil(T / 8) * 8 * groups blocks.
  // A wave owns a SUPER-tile of 2 x 2 tiles (block rows 2 SA, 2 SA + 1; block columns 2 SC, 2 SC + 1): its K^-1
  // loop loads four 16-row segments per trip for up to 16 MFMAs.  With one tile per wave the loop moved 4 KB per
  // 4 MFMAs and CU -- the vector L1's 64 B/clk exactly at the MFMA rate -- and ran load-bound.
  const int NBS = (NB + 1) / 2, NS = NBS * (NBS + 1) / 2;
  const int groups = (NS + 3) / 4;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int task = (slot / groups) * 8 + xcd;
  const int group = slot % groups;
  if (task >= p.T) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = SCAML_WAVE_INDEX(tid);
  const int lc = lane & 15, lq = lane >> 4;
  exp2_table_init(exptab, tid);
  if (tid < D && tid < 64) invl_s[tid] = 1.0 / p.theta[(size_t)task * (D + 2) + tid];
  __syncthreads();
  const int stile = group * (blockDim.x >> 6) + wave;
  if (stile >= NS) return;
  // super-tile -> (SA >= SC), column-major over the lower triangle of super-blocks
  int SC = 0, soff = 0;
  while (SC < NBS - 1 && soff + (NBS - SC) <= stile) { soff += NBS - SC; ++SC; }
  const int SA = SC + (stile - soff);
  const bool diag = SA == SC;
  const int ta0 = 2 * SA, ta1 = 2 * SA + 1, tc0 = 2 * SC, tc1 = 2 * SC + 1;   // (ta1 / tc1 may be >= NB: N % 32 in (0, 16])
  int n = p.n_points ? p.n_points[task] : N;
  n = n < 0 ? 0 : (n > N ? N : n);
  const double* th = p.theta + (size_t)task * (D + 2);
  const double os = th[D];
  const double* Xg = p.X + (size_t)task * N * D;
  const double* al = p.alpha + (size_t)task * N;
  const double* Li = p.Linv + (size_t)task * N * N;
  // K^-1 tiles: sum_r Linv[r][a] Linv[r][c] over r >= 16 ta0 (Linv is lower triangular: the rows of block ta0 are
  // zero in the columns of block ta1, one trip of zeros for that tile).  kin[i][j] <-> tile (ta_i, tc_j).
  d4_t k00 = {0.0, 0.0, 0.0, 0.0}, k01 = k00, k10 = k00, k11 = k00;
  // The 64 points of the four blocks (scaled by 1 / l) go into the wave's LDS slab once per super-tile -- before the
  // K^-1 loop, so that their global-load latency runs under it; odd pitch: the epilogue reads one dimension of 16
  // different points at a time.  Element e = lane + 64 i <-> (point e / D, dimension e % D), stepped without division.
  const int DP = (D + 1) | 1;   // column D carries alpha
  double* xs = xstage + (size_t)wave * 64 * DP;   // points 0..15 block ta0, 16..31 ta1, 32..47 tc0, 48..63 tc1
  {
    int r = lane / D, d = lane - r * D;
    const int dr = 64 / D, dd = 64 - dr * D;
    for (int e = lane; e < 64 * D; e += 64) {
      const int blk = r >> 4;
      const int tb = blk == 0 ? ta0 : (blk == 1 ? ta1 : (blk == 2 ? tc0 : tc1));
      const int row = 16 * tb + (r & 15);
      const double il = d < 64 ? invl_s[d] : 1.0 / th[d];
      xs[r * DP + d] = (tb < NB && row < n) ? Xg[(size_t)row * D + d] * il : 0.0;
      r += dr; d += dd;
      if (d >= D) { d -= D; ++r; }
    }
    {
      const int blk = lane >> 4;
      const int tb = blk == 0 ? ta0 : (blk == 1 ? ta1 : (blk == 2 ? tc0 : tc1));
      const int row = 16 * tb + (lane & 15);
      xs[lane * DP + D] = (tb < NB && row < n) ? al[row] : 0.0;
    }
  }
  const int pa0 = 16 * ta0 + lc, pa1 = 16 * ta1 + lc, pc0 = 16 * tc0 + lc, pc1 = 16 * tc1 + lc;
  int r0 = 16 * ta0;
  if (ta1 < NB && 16 * ta1 + 16 <= N) {   // every column of the four blocks is inside the matrix
    // Two operand sets, alternating: the 16 loads of the trip after next are issued before the 12 / 16 MFMAs of this one, and no set is
    // ever copied (round 3: the trip's operands used to be copied out of the set the next loads went into, and the compiler waited at
    // the copy for the loads of one trip earlier -- an exposed L2 round trip per trip).  Unconditional buffer loads: a trip past the
    // last row gets an offset beyond the descriptor's end and loads zeros.
    typedef unsigned gk_u2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Li), 0, N * N * 8, 0x00020000);
    struct OpSet {
      double a0[4], a1[4], b0[4], b1[4];
    };
    auto fetch = [&](OpSet& o, int rr) {
      const unsigned base = rr + 16 <= N ? (unsigned)(((size_t)(rr + lq) * N) * 8) : 0xC0000000u;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned ro = base + (unsigned)(4 * u * N * 8);
        o.a0[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsI, ro + (unsigned)(pa0 * 8), 0, 0));
        o.a1[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsI, ro + (unsigned)(pa1 * 8), 0, 0));
        o.b0[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsI, ro + (unsigned)(pc0 * 8), 0, 0));
        o.b1[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsI, ro + (unsigned)(pc1 * 8), 0, 0));
      }
    };
    auto mma = [&](const OpSet& o) {
#pragma unroll
      for (int u = 0; u < 4; ++u) k00 = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a0[u], o.b0[u], k00, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) k10 = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a1[u], o.b0[u], k10, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) k11 = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a1[u], o.b1[u], k11, 0, 0, 0);
      if (!diag) {   // (the tile above the diagonal of a diagonal super-tile is not needed)
#pragma unroll
        for (int u = 0; u < 4; ++u) k01 = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a0[u], o.b1[u], k01, 0, 0, 0);
      }
    };
    OpSet s0, s1;
    fetch(s0, r0);
    fetch(s1, r0 + 16);
    for (; r0 + 16 <= N; r0 += 32) {
      mma(s0);
      fetch(s0, r0 + 32);
      if (r0 + 32 <= N) mma(s1);
      fetch(s1, r0 + 48);
    }
    r0 = N & ~15;   // (every whole block row done)
  }
  for (; r0 < N; r0 += 4) {
    const int r = r0 + lq;
    const bool rin = r < N;
    const double a0 = (rin && pa0 < N) ? Li[(size_t)r * N + pa0] : 0.0;
    const double a1 = (rin && ta1 < NB && pa1 < N) ? Li[(size_t)r * N + pa1] : 0.0;
    const double b0 = (rin && pc0 < N) ? Li[(size_t)r * N + pc0] : 0.0;
    const double b1 = (rin && tc1 < NB && pc1 < N) ? Li[(size_t)r * N + pc1] : 0.0;
    k00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, k00, 0, 0, 0);
    k10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, k10, 0, 0, 0);
    k11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, k11, 0, 0, 0);
    k01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, k01, 0, 0, 0);
  }
  // epilogue per tile: this lane owns rows a = 16 ta + lq + 4g, column c = 16 tc + lc.  The 32 points of the two
  // blocks are staged (scaled by 1 / l) in the wave's LDS slab, so that both passes over the dimensions are
  // LDS reads, and the D + 2 sums are reduced over the wave at the very end, back to back.
  // Epilogue over the (up to) four tiles of the super-tile.  Pass A per tile: G = wgt (alpha alpha^T - K^-1), the
  // kernel value k and the radial factor h for the lane's four elements; the D lengthscale sums, the outputscale
  // and the noise sum are accumulated over ALL tiles of the super-tile and reduced over the wave once (the caller
  // adds the per-tile partials up anyway: the super-tile's totals go into the slot of its first tile, zeros into
  // the others).  Lane (lc, lq) owns rows a = 16 ta + lq + 4g, column c = 16 tc + lc of a tile.
  const int tiles_ta[4] = {ta0, ta1, ta1, ta0}, tiles_tc[4] = {tc0, tc0, tc1, tc1};
  const int tiles_xa[4] = {0, 16, 16, 0}, tiles_xc[4] = {32, 32, 48, 48};
  const bool tiles_on[4] = {true, ta1 < NB, ta1 < NB && tc1 < NB, !diag && tc1 < NB};
  double GH[4][4];
  double g_os = 0.0, g_noise = 0.0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const d4_t kin = t == 0 ? k00 : (t == 1 ? k10 : (t == 2 ? k11 : k01));
    const int ta = tiles_ta[t], tc = tiles_tc[t], xa = tiles_xa[t], xc = tiles_xc[t];
    const int pc = 16 * tc + lc;
    const double wgt = ta == tc ? 1.0 : 2.0;
    const double ac = xs[(xc + lc) * DP + D];   // alpha, staged with the points (0 past n)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      GH[t][g] = 0.0;
      if (tiles_on[t]) {
        const int a = 16 * ta + lq + 4 * g;
        const bool ok = a < n && pc < n;
        double d2 = 0.0;
#pragma unroll 4
        for (int d = 0; d < D; ++d) {   // (unrolled: the LDS reads of four dimensions are in flight together)
          const double df = xs[(xa + lq + 4 * g) * DP + d] - xs[(xc + lc) * DP + d];
          d2 = __builtin_fma(df, df, d2);
        }
        double k, h;
        kernel_and_dfactor<KIND>(d2, exptab, k, h);
        const double Gv = ok ? wgt * (xs[(xa + lq + 4 * g) * DP + D] * ac - kin[g]) : 0.0;
        GH[t][g] = Gv * os * h;   // (os folded in: every lengthscale term carries it)
        g_os = __builtin_fma(Gv, k, g_os);
        if (a == pc) g_noise += Gv;
      }
    }
  }
  const int tile0 = tc0 * NB - tc0 * (tc0 - 1) / 2 + (ta0 - tc0);   // column-major index over the lower triangle of tiles
  double* outp = p.partials + ((size_t)task * NT + tile0) * (D + 2);
  // lengthscales: d mll / d l_d ~ sum G os h delta_d^2 / l_d^3 with delta / l already in LDS: sum G os h (delta/l)^2 / l
  for (int d = 0; d < D; ++d) {
    double sd = 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (tiles_on[t]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const double df = xs[(tiles_xa[t] + lq + 4 * g) * DP + d] - xs[(tiles_xc[t] + lc) * DP + d];
          sd = __builtin_fma(GH[t][g], df * df, sd);
        }
      }
    }
    const double il = d < 64 ? invl_s[d] : 1.0 / th[d];
    sd = wave_sum_to_lane15(sd * il);
    if (lane == 63) outp[d] = sd;
  }
  g_os = wave_sum_to_lane15(g_os);
  g_noise = wave_sum_to_lane15(g_noise);
  if (lane == 63) {
    outp[D] = g_os;        // = sum G k  (d K / d os = k)
    outp[D + 1] = g_noise; // = tr G
  }
  // the other tiles of the super-tile contribute through the slot above: their own slots are zero
#pragma unroll
  for (int t = 1; t < 4; ++t) {
    if (tiles_on[t]) {
      const int ta = tiles_ta[t], tc = tiles_tc[t];
      const int tile = tc * NB - tc * (tc - 1) / 2 + (ta - tc);
      double* o = p.partials + ((size_t)task * NT + tile) * (D + 2);
      for (int e = lane; e < D + 2; e += 64) o[e] = 0.0;
    }
  }
}

}  // namespace scaml

template __global__ void scaml::gp_mll_grad_kernel<0>(scaml::MllGradParams);
template __global__ void scaml::gp_mll_grad_kernel<1>(scaml::MllGradParams);

// gp_mll_grad_fused.hip — the MLL hyper-gradient of a task in ONE kernel, without L^-1 or K^-1 in memory (gfx950).
//
//   d mll / d theta_p = (1 / 2n) sum_ij G_ij dK_ij/dtheta_p,   G = alpha alpha^T - K^-1,   K^-1 = (L L^T)^-1
//
// Replaces the autograd pass through kernel -> Cholesky -> solves that botorch's fit_gpytorch_mll runs per
// L-BFGS-B iteration (scamlgp/utils.py:175, 190) -- the heaviest call of the reference's meta-fit.
//
// Round 1 did this with two launches: L^-1 written to HBM dense (134 MB per 256-task launch) and streamed back
// ~3.6 times through L2 by the K^-1 tiles (490 MB), 104 + 118 us at T = 256, N = 256.  Here ONE workgroup owns a task:
//   * K^-1 is produced by column strips: strip c (16 columns) is the solution of L L^T Z = E_c -- a forward and a
//     backward substitution with the 16 columns of the identity as right-hand sides.  The strip's 16 x 16 blocks live
//     in REGISTERS (hand-managed AGPRs) in the MFMA C/D layout, which is exactly the B-operand layout of v_mfma_f64_16x16x4_f64
//     (lane (lc, lq) register m = element [lq + 4m][lc]): a finished block feeds the next products straight from the
//     registers it was computed in -- no LDS round trip, no transposes.  Strips need nothing from each other.
//   * Wave w owns strips (k, NBT-1-k): 16-k and k+1 blocks = NBT+1 blocks = 136 registers at N = 256, whatever k.
//     Block j of the first strip sits in slot NBT-1-j, of the second in slot j+1: compile-time functions of j, so the
//     fully unrolled code addresses registers statically although k is a run-time (per-wave) value.
//   * All strips advance in lockstep over the block row kb (forward) / block column kb (backward) of L.  That row /
//     column block (<= 33 KB, + the inverse W_kb of the diagonal block) is staged ONCE per step in LDS for the whole
//     workgroup, double-buffered, its global loads in flight a full step ahead: L is read from memory exactly twice
//     (2 x 272 KB per task at N = 256) instead of ~10 x through L2.
//   * The entries of a finished K^-1 block go straight into the gradient sums (kernel value and radial derivative
//     re-evaluated from the LDS-staged points, D + 2 running sums per lane): VALU work that overlaps with the MFMA
//     work of the other waves.  One wave reduction per sum at the very end, deterministic.
// Work: 2 x 816 block products of 4 MFMAs at N = 256 (the same 2 N^3 / 3 as L^-1 + L^-T L^-1).
#include "scaml_common.hpp"
#include "../../include/scaml_gp.h"
#include "gp_posterior_params.h"

#include "gf_tiles.hpp"

namespace scaml {

constexpr int GF_DP = 9;     // pitch of the staged points (D <= 8, zero-padded; odd: conflict-free row reads)
constexpr int GF_DMAX = 8;

// forward step KB of one strip (first block c): V_KB = W_KB (E_KB - sum_j L[KB][j] V_j)
template <int KB, int NBT, bool SB>
__device__ __forceinline__ void gf_fwd_strip(int c, const double* pa, int PA, int lc, int lq) {
  using Dst = GfTile<gf_slot<NBT, SB>(KB)>;
  if (KB == c) {
    // the strip's first block: V = W_KB * I -- W_KB itself, read in the C/D layout (row lq + 4 g, column lc)
    const double* w = pa - (lc * PA + lq) + lq * PA + 16 * KB + lc;
    Dst::set(w[0], w[4 * PA], w[8 * PA], w[12 * PA]);
    return;
  }
  GfOps cur = gf_load_ops(pa + 16 * c, 4);
  GF_DRAIN();        // (the previous strip step's last MFMA may still be reading ACC)
  gf_acc_zero();
  gf_fwd_chain<KB, NBT, SB>(cur, c, pa);
  GF_DRAIN();
  Dst::set_prod_acc(cur.a0, cur.a1, cur.a2, cur.a3);   // (W_KB sits in the staged row block's last 16 columns)
}

// backward step KB of one strip: Z_KB = W_KB^T (V_KB - sum_j L[j][KB]^T Z_j); the finished block also goes to the
// wave's LDS scratch (register image, lane-owned) for the gradient epilogue
template <int KB, int NBT, bool SB>
__device__ __forceinline__ void gf_bwd_strip(int last, const double* pb, double* zout, int lane) {
  using Dst = GfTile<gf_slot<NBT, SB>(KB)>;
  GfOps cur = gf_load_ops(pb + 16 * (last > KB ? last : KB) * 16, 64);
  GF_DRAIN();
  Dst::copy_to_acc();   // V_KB: written by the forward pass, many steps (and barriers) ago
  gf_bwd_chain<KB, NBT, SB>(cur, last, pb);
  GF_DRAIN();
  Dst::set_prod_acc(cur.a0, cur.a1, cur.a2, cur.a3);
  GF_DRAIN();
  const d4_t z = Dst::get();
  zout[lane] = z[0]; zout[64 + lane] = z[1]; zout[128 + lane] = z[2]; zout[192 + lane] = z[3];
}

template <int KB, int NBT>
__device__ __forceinline__ void gf_fwd_step(int cA, int cB, const double* pa, int PA, int lc, int lq) {
  if (KB >= cA) gf_fwd_strip<KB, NBT, false>(cA, pa, PA, lc, lq);
  if constexpr (KB >= NBT / 2) {   // (a second strip starts at block NBT/2 at the earliest)
    if (KB >= cB) gf_fwd_strip<KB, NBT, true>(cB, pa, PA, lc, lq);
  }
}

template <int KB, int NBT>
__device__ __forceinline__ void gf_bwd_step(int cA, int cB, int last, const double* pb, double* zs, int lane) {
  if (KB >= cA) gf_bwd_strip<KB, NBT, false>(last, pb, zs, lane);
  if constexpr (KB >= NBT / 2) {
    if (KB >= cB) gf_bwd_strip<KB, NBT, true>(last, pb, zs + 256, lane);
  }
}

// kernel value k (without outputscale) and h with dk/dl_d = GF_HSCALE<KIND> * h * delta_d^2 / l_d^3 (the caller folds the constant
// into the outputscale).  Matern: the fused fit's instruction sequence (scaml_common.hpp: kernel_from_sqdist_scaled) -- sqrt straight
// from the f32 seed, d2 clamped to [1e-30, 1e5] so that exp needs no clamp of its own.
template <int KIND>
__device__ __forceinline__ constexpr double gf_hscale() { return KIND == 0 ? 1.0 : 5.0 / 3.0; }
template <int KIND>
__device__ __forceinline__ void gf_kernel_and_dfactor(double d2, const double* exptab, double& k, double& h) {
  if (KIND == 0) {
    k = exp_neg_t<true>(-0.5 * d2, exptab);
    h = k;
  } else {
    const double dd = vmin_f64(vmax_f64(d2, 1e-30), 1e5);   // (bare v_max / v_min: one instruction each)
    const double y = (double)__builtin_amdgcn_rsqf((float)dd);
    const double g = dd * y;
    const double e = __builtin_fma(-g, y, 1.0);
    const double r = __builtin_fma(g * e, __builtin_fma(e, 0.375, 0.5), g);
    const double s5 = 2.2360679774997896964;
    const double ex = exp_neg_t<false>(-s5 * r, exptab);
    const double p1 = __builtin_fma(s5, r, 1.0);
    k = __builtin_fma(r * (5.0 / 3.0), r, p1) * ex;
    h = p1 * ex;
  }
}

#define GF_STEP_CASE(K)                                                  \
  case K:                                                                \
    if constexpr (K < NBT) {                                             \
      if (fwd) gf_fwd_step<K, NBT>(cA, cB, pa, PA, lc, lq);              \
      else gf_bwd_step<K, NBT>(cA, cB, NB - 1, pb, zs, lane);            \
    }                                                                    \
    break;

// Per-strip state of the gradient epilogue: T accumulates [X_a^T; (X_a^2)^T] GH over the strip's blocks (rows 0..7: sum_a
// x_ad GH_ac, rows 8..15: sum_a x_ad^2 GH_ac, column c on the lane), cs the lane's share of the column sums of GH.
struct GfStrip {
  d4_t T;
  double cs;
};

// Gradient contributions of the finished K^-1 block (kb, c) (register image `zt`, lane-owned): G = alpha alpha^T - K^-1,
// GH = G os h;  sum G k -> g_os, tr G -> g_noise, and for the lengthscales
//   sum_ac GH_ac (x_ad - x_cd)^2 = sum_c [ Q2_dc - 2 x_cd Q_dc + x_cd^2 CS_c ]
// with Q = X_a^T GH, Q2 = (X_a^2)^T GH accumulated on the matrix core (GH in the C/D layout IS the B operand) and the
// squared distances of the block from one MFMA chain as well (expanded form, norms as a third k-step): the VALU is left
// with the kernel function and G.  At the strip's last block (kb == c) the sums over a are complete and folded into the
// lane's two lengthscale partials (dimensions lq and lq + 4).
template <int KIND, bool SETTLE_T>
__device__ __forceinline__ void gf_epilogue(GfStrip& st, const double* zt, int kb, int c, int n, const double* Xs, const double* Xq, const double* als,
                                            const double* exptab, double os, double& g_os, double& g_noise, double& pd0,
                                            double& pd1, int lc, int lq) {
  const int col = 16 * c + lc;
  const double* xcp = Xs + col * GF_DP;
  const double* xap = Xs + (16 * kb + lc) * GF_DP;
  const double xc0 = xcp[lq], xc1 = xcp[lq + 4];           // B operand of the distance product: X_c[d = lq + 4 m][point lc]
  const double nc = xcp[8], na = xap[8];
  d4_t d2v = {0.0, 0.0, 0.0, 0.0};
  d2v = __builtin_amdgcn_mfma_f64_16x16x4f64(-2.0 * xap[lq], xc0, d2v, 0, 0, 0);
  d2v = __builtin_amdgcn_mfma_f64_16x16x4f64(-2.0 * xap[lq + 4], xc1, d2v, 0, 0, 0);
  d2v = __builtin_amdgcn_mfma_f64_16x16x4f64(lq == 0 ? na : (lq == 1 ? 1.0 : 0.0), lq == 0 ? 1.0 : (lq == 1 ? nc : 0.0), d2v, 0, 0, 0);
  d2v = gf_settle(d2v);
  const double ac = als[col];
  const double hs = os * gf_hscale<KIND>();
  d4_t GH;
  if (kb != c && 16 * kb + 16 <= n) {
    // wave-uniform fast path (120 of the 136 blocks at N = 256): an off-diagonal block entirely inside the n valid points has no
    // diagonal element and needs no masks -- ~10 VALU instructions per element less
    // (off-diagonal blocks stand for both triangles: the factor 2 rides on the outputscale factor and on a separate sum)
    const double hs2 = 2.0 * hs;
    double gk = 0.0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      double k, h;
      // (Matern clamps d2 at 1e-30 itself; RBF must not see the slightly negative values of the expanded form)
      gf_kernel_and_dfactor<KIND>(KIND == 0 ? vmax_f64(d2v[g], 0.0) : d2v[g], exptab, k, h);
      const double Gh = __builtin_fma(als[16 * kb + lq + 4 * g], ac, -zt[64 * g]);   // half of G
      GH[g] = (Gh * hs2) * h;
      gk = __builtin_fma(Gh, k, gk);
      st.cs += GH[g];
    }
    g_os = __builtin_fma(2.0, gk, g_os);
  } else {
    const double wgt = kb == c ? 1.0 : 2.0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = 16 * kb + lq + 4 * g;
      double d2 = d2v[g] > 0.0 ? d2v[g] : 0.0;
      if (row == col) d2 = 0.0;
      double k, h;
      gf_kernel_and_dfactor<KIND>(d2, exptab, k, h);
      const bool ok = row < n && col < n;
      const double Gv = ok ? wgt * (als[row] * ac - zt[64 * g]) : 0.0;
      GH[g] = (Gv * hs) * h;
      g_os = __builtin_fma(Gv, k, g_os);
      if (row == col) g_noise += Gv;
      st.cs += GH[g];
    }
  }
  // T += [X_a^T; (X_a^2)^T] GH: A[i = lc][k = lq + 4 m] = x_{a, k}[d = lc & 7], squared for the rows i >= 8
  // (the squares are staged next to the coordinates: a pointer select per block instead of a multiply and a select per operand)
  const double* xk = (lc >= 8 ? Xq : Xs) + (16 * kb + lq) * GF_DP + (lc & 7);
  d4_t T = st.T;
#pragma unroll
  for (int m = 0; m < 4; ++m) T = __builtin_amdgcn_mfma_f64_16x16x4f64(xk[4 * m * GF_DP], GH[m], T, 0, 0, 0);
  // (register-staged variants spill: hipcc once placed a scratch reload into the accumulator 16 wait states behind the last MFMA --
  //  the build's hazard audit refused the code object; those variants pay 19 idle cycles per block for a guaranteed distance)
  if constexpr (SETTLE_T) T = gf_settle(T);
  st.T = T;
  if (kb == c) {
    const d4_t Tf = gf_settle(T);
    const double cscol = sum_lane_groups(st.cs);   // column sum over all rows of the strip
    // lane (lc, lq): rows lq, lq + 4 of Q (registers 0, 1) and of Q2 (registers 2, 3)
    pd0 += Tf[2] - 2.0 * xc0 * Tf[0] + xc0 * xc0 * cscol;
    pd1 += Tf[3] - 2.0 * xc1 * Tf[1] + xc1 * xc1 * cscol;
  }
}

// SPLIT workgroups share a task (grid = (T, SPLIT)): the strips are independent, so each workgroup takes NW / SPLIT of
// the NBT / 2 strip pairs, stages L for itself and writes its own partial sums (tile slot blockIdx.y).  Used when the
// stack does not fill the CUs with one workgroup per task (BASELINE configs[3]: 128 tasks per GPU on 256 CUs).
template <int NBT, int KIND, bool DMA, int SPLIT>
__global__ __launch_bounds__(NBT * 32 / SPLIT) void gp_mll_grad_fused_kernel(MllGradFusedParams p) {
  static_assert(SPLIT == 1 || (DMA && NBT >= 8 && (SPLIT == 2 || SPLIT == 4)), "split tasks: N <= 128 / N <= 256 classes with LDS-DMA staging only");
  constexpr int NP = 16 * NBT;
  constexpr int PA = NP + 2;             // pitch of a staged row block: lc * PA + lq hits 32 different 8-byte banks
  constexpr int NWT = NBT / 2;           // strip pairs = waves per task
  constexpr int NW = NWT / SPLIT > 0 ? NWT / SPLIT : 1;   // waves of this workgroup
  constexpr int TPB = NW * 64;
  constexpr int BUF = 16 * PA;           // doubles per staging buffer (a column block, NP x 16, fits as well)
  extern __shared__ double lds[];
  double* buf = lds;                     // [2][BUF]
  double* Xs = buf + 2 * BUF;            // [NP][GF_DP] points scaled by 1 / lengthscale, zero-padded to 8 dimensions; [8] = |x|^2
  double* Xq = Xs + NP * GF_DP + 16;     // [NP][GF_DP] their squares, coordinate by coordinate (16 banks away: lanes lc and lc + 8 read the two arrays together)
  double* als = Xq + NP * GF_DP;         // [NP] alpha (0 past n)
  double* zs_all = als + NP;             // [NW][2][256] finished K^-1 blocks (register images)
  double* exptab = zs_all + NW * 512;    // [64]
  double* invl = exptab + 64;            // [8]
  double* red = invl + 8;                // [NW][10]

  const int N = p.N, D = p.D;
  const int NB = (N + 15) / 16;
  const int task = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, lq = lane >> 4;
  int n = p.n_points ? p.n_points[task] : N;
  n = n < 0 ? 0 : (n > N ? N : n);
  const double* Lg = p.L + (size_t)task * N * N;
  const double* Wg = p.Linv_diag + (size_t)task * NB * 256;
  const double* th = p.theta + (size_t)task * (D + 2);
  const double os = th[D];
  // strip pair of this wave: (k, NBT-1-k); the second half of the waves takes the pairs in reverse, so that the two
  // waves of a SIMD (w, w + NW/2) carry pairs (s, NW-1-s) whose block counts per step add up evenly
  // (split tasks: workgroup g of 2 takes the virtual waves {g, 3-g, 4+g, 7-g}, of 4 {g, g+4}: near-equal block counts again)
  const int grp = SPLIT > 1 ? (int)blockIdx.y : 0;
  const int vw = SPLIT == 1 ? wave : (SPLIT == 2 ? 2 * wave + ((wave & 1) ? 1 - grp : grp) : grp + 4 * wave);
  const int kp = (NWT < 2 || vw < NWT / 2) ? vw : (NWT + NWT / 2 - 1 - vw);
  const int cA = kp, cB = NBT - 1 - kp;

  // ---- staging of one step's slice of L (t < NB: row block t, forward; else column block 2 NB - 1 - t, backward) with
  // the inverse W of the diagonal block in place of the diagonal block itself.
  // DMA (the host picks it when N is a multiple of 16 and the stack is not ragged, so that no element needs masking):
  // global_load_lds_dwordx4 -- each wave-instruction moves 64 x 16 B from per-lane addresses to 1 KB of consecutive LDS,
  // no VGPRs, no LDS-write pass; issued at the start of step t for step t + 1, waited for (vmcnt) before the barrier that
  // ends step t.  Otherwise: registers now, LDS one step later (masked loads).
  typedef __attribute__((address_space(1))) void gvoid_t;
  typedef __attribute__((address_space(3))) void lvoid_t;
  auto dma_step = [&](int t) {
    double* b = buf + (t & 1) * BUF;
    if (t < NB) {
      const int kb = t;
      constexpr int RPW = 16 / NW;                 // rows per wave
      const int nch = 8 * kb + 8;                  // 16-byte pieces per row: 16 kb columns of L, then the 16 of W_kb
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        const double* lrow = Lg + (size_t)(16 * kb + r) * N;
        const double* wrow = Wg + (size_t)kb * 256 + r * 16 - 16 * kb;
        for (int c0 = 0; c0 < nch; c0 += 64) {
          const int ch = c0 + lane;
          if (ch < nch) {
            const double* src = (ch < 8 * kb ? lrow : wrow) + 2 * ch;
            __builtin_amdgcn_global_load_lds((const gvoid_t*)src, (lvoid_t*)(b + r * PA + 2 * c0), 16, 0, 0);
          }
        }
      }
    } else if (t < 2 * NB) {
      const int kb = 2 * NB - 1 - t;
      // rows 16 kb .. 16 NB - 1 of the column block, 8 rows (128 B each) per wave-instruction, dealt to the waves in turn
      for (int r0 = 16 * kb + 8 * wave; r0 < 16 * NB; r0 += 8 * NW) {
        const int row = r0 + (lane >> 3), c2 = 2 * (lane & 7);
        const double* src = row < 16 * kb + 16 ? Wg + (size_t)kb * 256 + (row - 16 * kb) * 16 + c2 : Lg + (size_t)row * N + 16 * kb + c2;
        __builtin_amdgcn_global_load_lds((const gvoid_t*)src, (lvoid_t*)(b + r0 * 16), 16, 0, 0);
      }
    }
  };
  double stg[DMA ? 1 : 8];
  const bool n_even = (N & 1) == 0;
  auto load_step = [&](int t) {
    if constexpr (!DMA) {
      if (t < NB) {
        const int kb = t, r = tid / (2 * NBT), ch = tid % (2 * NBT);
        const int row = 16 * kb + r, col0 = 8 * ch;
        if (col0 < 16 * kb) {
          const double* src = Lg + (size_t)row * N + col0;
          if (n_even && row < n && col0 + 8 <= n) {
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
              const double2 v = *reinterpret_cast<const double2*>(src + q);
              stg[q] = v.x; stg[q + 1] = v.y;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) stg[q] = (row < n && col0 + q < n) ? src[q] : 0.0;
          }
        } else if (col0 < 16 * kb + 16) {
          const double* src = Wg + (size_t)kb * 256 + r * 16 + (col0 - 16 * kb);
#pragma unroll
          for (int q = 0; q < 8; q += 2) {
            const double2 v = *reinterpret_cast<const double2*>(src + q);
            stg[q] = v.x; stg[q + 1] = v.y;
          }
        }
      } else if (t < 2 * NB) {
        const int kb = 2 * NB - 1 - t, row = tid >> 1, c8 = 8 * (tid & 1);
        if (row >= 16 * kb + 16) {
          const int col0 = 16 * kb + c8;
          const double* src = Lg + (size_t)row * N + col0;
          if (n_even && row < n) {   // (col0 + 8 <= 16 kb + 16 <= row < n)
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
              const double2 v = *reinterpret_cast<const double2*>(src + q);
              stg[q] = v.x; stg[q + 1] = v.y;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) stg[q] = (row < n && col0 + q < n) ? src[q] : 0.0;
          }
        } else if (row >= 16 * kb) {
          const double* src = Wg + (size_t)kb * 256 + (row - 16 * kb) * 16 + c8;
#pragma unroll
          for (int q = 0; q < 8; q += 2) {
            const double2 v = *reinterpret_cast<const double2*>(src + q);
            stg[q] = v.x; stg[q + 1] = v.y;
          }
        }
      }
    }
  };
  auto store_step = [&](int t) {
    if constexpr (!DMA) {
      double* b = buf + (t & 1) * BUF;
      if (t < NB) {
        const int kb = t, r = tid / (2 * NBT), ch = tid % (2 * NBT);
        if (8 * ch < 16 * kb + 16) {
          double* dst = b + r * PA + 8 * ch;
#pragma unroll
          for (int q = 0; q < 8; ++q) dst[q] = stg[q];
        }
      } else if (t < 2 * NB) {
        const int kb = 2 * NB - 1 - t, row = tid >> 1;
        if (row >= 16 * kb) {
          double* dst = b + row * 16 + 8 * (tid & 1);
#pragma unroll
          for (int q = 0; q < 8; ++q) dst[q] = stg[q];
        }
      }
    }
  };

  if constexpr (DMA) dma_step(0); else load_step(0);
  exp2_table_init(exptab, tid);
  if (tid < GF_DMAX) invl[tid] = tid < D ? 1.0 / th[tid] : 0.0;
  __syncthreads();
  for (int r = tid; r < NP; r += TPB) {
    double nrm = 0.0;
#pragma unroll
    for (int d = 0; d < GF_DMAX; ++d) {
      const double v = (r < n && d < D) ? p.X[((size_t)task * N + r) * D + d] * invl[d] : 0.0;
      Xs[r * GF_DP + d] = v;
      Xq[r * GF_DP + d] = v * v;
      nrm = __builtin_fma(v, v, nrm);
    }
    Xs[r * GF_DP + 8] = nrm;
    als[r] = r < n ? p.alpha[(size_t)task * N + r] : 0.0;
  }
  if constexpr (DMA) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    store_step(0);
    load_step(1);
  }
  __syncthreads();

  GfStrip stA = {d4_t{0.0, 0.0, 0.0, 0.0}, 0.0}, stB = stA;
  double g_os = 0.0, g_noise = 0.0, pd0 = 0.0, pd1 = 0.0;
  double* zs = zs_all + wave * 512;

  // (-DGF_STAGGER: the second half of the waves defers a step's epilogue to the start of the next step, so that of
  //  the two waves of a SIMD one streams MFMAs while its partner evaluates kernel functions -- measured SLOWER,
  //  133.5 vs 127.3 us at T = 256, N = 256 in interleaved A/B runs: off)
#ifdef GF_STAGGER
  const bool late = NW >= 2 && wave >= NW / 2;
#else
  const bool late = false;
#endif
  auto epilogue = [&](int kb) {
#ifndef GF_NO_EPI
    if (kb >= cA) gf_epilogue<KIND, !DMA>(stA, zs + lane, kb, cA, n, Xs, Xq, als, exptab, os, g_os, g_noise, pd0, pd1, lc, lq);
    if (kb >= cB) gf_epilogue<KIND, !DMA>(stB, zs + 256 + lane, kb, cB, n, Xs, Xq, als, exptab, os, g_os, g_noise, pd0, pd1, lc, lq);
#endif
  };
  for (int t = 0; t < 2 * NB; ++t) {
#ifndef GF_NO_STAGE
    if constexpr (DMA) {
      dma_step(t + 1);   // into the buffer step t - 1 read from: everybody is past the barrier that ended it
    } else {
      // the slice for step t + 1 (loaded during step t - 1) goes into the buffer step t - 1 read from; then the loads for
      // step t + 2 start and have this whole step to land
      store_step(t + 1);
      load_step(t + 2);
    }
#endif
    const bool fwd = t < NB;
    const int kb = fwd ? t : 2 * NB - 1 - t;
    if (late && t > NB) epilogue(kb + 1);    // the previous (backward) step's blocks
    const double* b = buf + (t & 1) * BUF;
    const double* pa = b + lc * PA + lq;     // forward A operand:  L[16 kb + lc][16 j + lq + 4 m]
    const double* pb = b + lq * 16 + lc;     // backward A operand: L[16 j + lq + 4 m][16 kb + lc]
    switch (kb) {
      GF_STEP_CASE(0) GF_STEP_CASE(1) GF_STEP_CASE(2) GF_STEP_CASE(3) GF_STEP_CASE(4) GF_STEP_CASE(5) GF_STEP_CASE(6)
      GF_STEP_CASE(7) GF_STEP_CASE(8) GF_STEP_CASE(9) GF_STEP_CASE(10) GF_STEP_CASE(11) GF_STEP_CASE(12) GF_STEP_CASE(13)
      GF_STEP_CASE(14) GF_STEP_CASE(15)
      default: break;
    }
    if (!fwd && !late) epilogue(kb);
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next slice have landed
    __syncthreads();
  }
  if (late) epilogue(0);

  // ---- D + 2 sums: lengthscale partials over the 16 columns of a lane group (dimensions lq, lq + 4), the two scalars over
  // the wave; then over the waves in a fixed order; the task's totals go into tile slot 0
  pd0 = row_sum_to_lane15(pd0);
  pd1 = row_sum_to_lane15(pd1);
  if (lc == 15) {
    red[wave * 10 + lq] = pd0 * invl[lq];
    red[wave * 10 + lq + 4] = pd1 * invl[lq + 4];
  }
  g_os = wave_sum_to_lane15(g_os);
  g_noise = wave_sum_to_lane15(g_noise);
  if (lane == 63) { red[wave * 10 + 8] = g_os; red[wave * 10 + 9] = g_noise; }
  __syncthreads();
  const int NT = NB * (NB + 1) / 2;
  double* outp = p.partials + (size_t)task * NT * (D + 2);
  if (tid < D + 2) {
    const int src = tid < D ? tid : 8 + (tid - D);
    double s = 0.0;
    for (int w = 0; w < NW; ++w) s += red[w * 10 + src];
    outp[grp * (D + 2) + tid] = s;
  }
  if (grp == 0)
    for (int e = SPLIT * (D + 2) + tid; e < NT * (D + 2); e += TPB) outp[e] = 0.0;
}
#undef GF_STEP_CASE

}  // namespace scaml

#define SCAML_INSTANTIATE_GF(NBT)                                                                             \
  template __global__ void scaml::gp_mll_grad_fused_kernel<NBT, 0, false, 1>(scaml::MllGradFusedParams);     \
  template __global__ void scaml::gp_mll_grad_fused_kernel<NBT, 1, false, 1>(scaml::MllGradFusedParams);     \
  template __global__ void scaml::gp_mll_grad_fused_kernel<NBT, 0, true, 1>(scaml::MllGradFusedParams);      \
  template __global__ void scaml::gp_mll_grad_fused_kernel<NBT, 1, true, 1>(scaml::MllGradFusedParams);
SCAML_INSTANTIATE_GF(2)
SCAML_INSTANTIATE_GF(4)
SCAML_INSTANTIATE_GF(8)
SCAML_INSTANTIATE_GF(16)
template __global__ void scaml::gp_mll_grad_fused_kernel<16, 0, true, 2>(scaml::MllGradFusedParams);
template __global__ void scaml::gp_mll_grad_fused_kernel<16, 1, true, 2>(scaml::MllGradFusedParams);
template __global__ void scaml::gp_mll_grad_fused_kernel<16, 0, true, 4>(scaml::MllGradFusedParams);
template __global__ void scaml::gp_mll_grad_fused_kernel<16, 1, true, 4>(scaml::MllGradFusedParams);
template __global__ void scaml::gp_mll_grad_fused_kernel<8, 0, true, 2>(scaml::MllGradFusedParams);
template __global__ void scaml::gp_mll_grad_fused_kernel<8, 1, true, 2>(scaml::MllGradFusedParams);
template __global__ void scaml::gp_mll_grad_fused_kernel<8, 0, true, 4>(scaml::MllGradFusedParams);
template __global__ void scaml::gp_mll_grad_fused_kernel<8, 1, true, 4>(scaml::MllGradFusedParams);

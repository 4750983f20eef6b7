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

namespace scaml {

constexpr int GF_DP = 9;     // pitch of the staged points (D <= 8, zero-padded; odd: conflict-free row reads)
constexpr int GF_DMAX = 8;

// ---- hand-managed block registers ------------------------------------------------------------------------------
// The strips' blocks live in AGPRs a[8 S : 8 S + 7] (slot S) that the COMPILER NEVER SEES as values: with the blocks as
// C++ values hipcc shuffled all 136 registers through copies at every merge of the unrolled step code (and, like in
// the fused fit, spilled).  Every access is an asm statement naming the physical registers (csrc/tile_regs.inc); the
// build sets "amdgpu-agpr-alloc"="0" so the compiler keeps out of the AGPR half, and caps the arch VGPRs so that
// VGPRs + block AGPRs <= 256 (two waves per SIMD).  Hazards hipcc cannot see inside asm: GF_DRAIN (19 wait states)
// separates the last MFMA writing a register from any non-accumulating read of it (as MFMA A/B operand,
// v_accvgpr_read, VALU); dependent accumulation into the same registers issues back to back (interlocked).
#include "tile_regs.inc"
#define GF_DRAIN() asm volatile("s_nop 15\n\ts_nop 2" ::: "memory")

// The accumulator of a strip step is a block of its own, slot 17 = a[136:143] ("ACC"): compiler-visible VGPR
// accumulators were copied around by VALU moves right behind the asm MFMAs (stale reads of the last result pair).
#define GF_ACC "a[136:143]"
__device__ __forceinline__ void gf_acc_zero() {
  const int z = 0;
  asm volatile("v_accvgpr_write_b32 a136, %0\n\tv_accvgpr_write_b32 a137, %0\n\tv_accvgpr_write_b32 a138, %0\n\t"
               "v_accvgpr_write_b32 a139, %0\n\tv_accvgpr_write_b32 a140, %0\n\tv_accvgpr_write_b32 a141, %0\n\t"
               "v_accvgpr_write_b32 a142, %0\n\tv_accvgpr_write_b32 a143, %0"
               :
               : "v"(z)
               : "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143");
}

template <int S>
struct GfTile;
#define GF_DEF_TILE(S, r0, r1, r2, r3, r4, r5, r6, r7)                                                             \
  template <>                                                                                                      \
  struct GfTile<S> {                                                                                               \
    /* ACC -= A * block, the block as B operand (k-step m reads register pair m) */                                \
    static __device__ __forceinline__ void chain_neg(double a0, double a1, double a2, double a3) {                 \
      asm volatile("s_nop 1\n\t"                                                                                   \
                   "v_mfma_f64_16x16x4_f64 " GF_ACC ", %0, a[" #r0 ":" #r1 "], " GF_ACC " neg:[1,0,0]\n\t"         \
                   "v_mfma_f64_16x16x4_f64 " GF_ACC ", %1, a[" #r2 ":" #r3 "], " GF_ACC " neg:[1,0,0]\n\t"         \
                   "v_mfma_f64_16x16x4_f64 " GF_ACC ", %2, a[" #r4 ":" #r5 "], " GF_ACC " neg:[1,0,0]\n\t"         \
                   "v_mfma_f64_16x16x4_f64 " GF_ACC ", %3, a[" #r6 ":" #r7 "], " GF_ACC " neg:[1,0,0]"              \
                   :                                                                                               \
                   : "v"(a0), "v"(a1), "v"(a2), "v"(a3)                                                            \
                   : "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143");                              \
    }                                                                                                              \
    /* block = A * ACC (the caller has drained the MFMAs that wrote ACC) */                                        \
    static __device__ __forceinline__ void set_prod_acc(double a0, double a1, double a2, double a3) {              \
      asm volatile("s_nop 1\n\t"                                                                                   \
                   "v_mfma_f64_16x16x4_f64 a[" #r0 ":" #r7 "], %0, a[136:137], 0\n\t"                              \
                   "v_mfma_f64_16x16x4_f64 a[" #r0 ":" #r7 "], %1, a[138:139], a[" #r0 ":" #r7 "]\n\t"             \
                   "v_mfma_f64_16x16x4_f64 a[" #r0 ":" #r7 "], %2, a[140:141], a[" #r0 ":" #r7 "]\n\t"             \
                   "v_mfma_f64_16x16x4_f64 a[" #r0 ":" #r7 "], %3, a[142:143], a[" #r0 ":" #r7 "]\n\t"             \
                   "s_nop 7"                                                                                       \
                   :                                                                                               \
                   : "v"(a0), "v"(a1), "v"(a2), "v"(a3)                                                            \
                   : "a" #r0, "a" #r1, "a" #r2, "a" #r3, "a" #r4, "a" #r5, "a" #r6, "a" #r7);                      \
    }                                                                                                              \
    /* ACC <- block */                                                                                             \
    static __device__ __forceinline__ void copy_to_acc() {                                                         \
      asm volatile("v_accvgpr_mov_b32 a136, a" #r0 "\n\tv_accvgpr_mov_b32 a137, a" #r1 "\n\t"                     \
                   "v_accvgpr_mov_b32 a138, a" #r2 "\n\tv_accvgpr_mov_b32 a139, a" #r3 "\n\t"                     \
                   "v_accvgpr_mov_b32 a140, a" #r4 "\n\tv_accvgpr_mov_b32 a141, a" #r5 "\n\t"                     \
                   "v_accvgpr_mov_b32 a142, a" #r6 "\n\tv_accvgpr_mov_b32 a143, a" #r7                             \
                   :                                                                                               \
                   :                                                                                               \
                   : "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143");                              \
    }                                                                                                              \
    /* block <- four doubles (element g = row lq + 4 g, column lc) */                                              \
    static __device__ __forceinline__ void set(double d0, double d1, double d2, double d3) {                       \
      asm volatile("v_accvgpr_write_b32 a" #r0 ", %0\n\tv_accvgpr_write_b32 a" #r1 ", %1\n\t"                     \
                   "v_accvgpr_write_b32 a" #r2 ", %2\n\tv_accvgpr_write_b32 a" #r3 ", %3\n\t"                     \
                   "v_accvgpr_write_b32 a" #r4 ", %4\n\tv_accvgpr_write_b32 a" #r5 ", %5\n\t"                     \
                   "v_accvgpr_write_b32 a" #r6 ", %6\n\tv_accvgpr_write_b32 a" #r7 ", %7"                          \
                   :                                                                                               \
                   : "v"(__double2loint(d0)), "v"(__double2hiint(d0)), "v"(__double2loint(d1)),                    \
                     "v"(__double2hiint(d1)), "v"(__double2loint(d2)), "v"(__double2hiint(d2)),                    \
                     "v"(__double2loint(d3)), "v"(__double2hiint(d3))                                              \
                   : "a" #r0, "a" #r1, "a" #r2, "a" #r3, "a" #r4, "a" #r5, "a" #r6, "a" #r7);                      \
    }                                                                                                              \
    /* four doubles <- block (the caller has drained the MFMAs that wrote it) */                                   \
    static __device__ __forceinline__ d4_t get() {                                                                 \
      int l0, h0, l1, h1, l2, h2, l3, h3;                                                                          \
      asm volatile("v_accvgpr_read_b32 %0, a" #r0 "\n\tv_accvgpr_read_b32 %1, a" #r1 "\n\t"                       \
                   "v_accvgpr_read_b32 %2, a" #r2 "\n\tv_accvgpr_read_b32 %3, a" #r3 "\n\t"                       \
                   "v_accvgpr_read_b32 %4, a" #r4 "\n\tv_accvgpr_read_b32 %5, a" #r5 "\n\t"                       \
                   "v_accvgpr_read_b32 %6, a" #r6 "\n\tv_accvgpr_read_b32 %7, a" #r7                               \
                   : "=v"(l0), "=v"(h0), "=v"(l1), "=v"(h1), "=v"(l2), "=v"(h2), "=v"(l3), "=v"(h3));              \
      d4_t r = {__hiloint2double(h0, l0), __hiloint2double(h1, l1), __hiloint2double(h2, l2),                      \
                __hiloint2double(h3, l3)};                                                                         \
      return r;                                                                                                    \
    }                                                                                                              \
  };
SCAML_TILE_LIST(GF_DEF_TILE)
#undef GF_DEF_TILE

// register slot of block j of a wave's first (SB = false) / second (SB = true) strip
template <int NBT, bool SB>
__device__ __forceinline__ constexpr int gf_slot(int j) { return SB ? j + 1 : NBT - 1 - j; }

// forward: acc -= sum_{j = c}^{KB-1} L[KB][j] V_j, entered at the run-time first block c and falling through
#define GF_FWD_CASE(J)                                                                                  \
  case J:                                                                                               \
    if constexpr (J < KB && J < NBT) {                                                                  \
      const double* q = pa + 16 * J;                                                                    \
      GfTile<gf_slot<NBT, SB>(J < NBT ? J : 0)>::chain_neg(q[0], q[4], q[8], q[12]);               \
    }                                                                                                   \
    [[fallthrough]];
template <int KB, int NBT, bool SB>
__device__ __forceinline__ void gf_fwd_chain(int c, const double* pa) {
  switch (c) {
    GF_FWD_CASE(0) GF_FWD_CASE(1) GF_FWD_CASE(2) GF_FWD_CASE(3) GF_FWD_CASE(4) GF_FWD_CASE(5) GF_FWD_CASE(6) GF_FWD_CASE(7)
    GF_FWD_CASE(8) GF_FWD_CASE(9) GF_FWD_CASE(10) GF_FWD_CASE(11) GF_FWD_CASE(12) GF_FWD_CASE(13) GF_FWD_CASE(14)
    default: break;
  }
}
#undef GF_FWD_CASE

// backward: acc -= sum_{j = KB+1}^{NB-1} L[j][KB]^T Z_j, entered at the run-time last block NB-1 and falling through
#define GF_BWD_CASE(J)                                                                                  \
  case J:                                                                                               \
    if constexpr (J > KB && J < NBT) {                                                                  \
      const double* q = pb + 16 * J * 16;                                                               \
      GfTile<gf_slot<NBT, SB>(J < NBT ? J : 0)>::chain_neg(q[0], q[64], q[128], q[192]);           \
    }                                                                                                   \
    [[fallthrough]];
template <int KB, int NBT, bool SB>
__device__ __forceinline__ void gf_bwd_chain(int last, const double* pb) {
  switch (last) {
    GF_BWD_CASE(15) GF_BWD_CASE(14) GF_BWD_CASE(13) GF_BWD_CASE(12) GF_BWD_CASE(11) GF_BWD_CASE(10) GF_BWD_CASE(9)
    GF_BWD_CASE(8) GF_BWD_CASE(7) GF_BWD_CASE(6) GF_BWD_CASE(5) GF_BWD_CASE(4) GF_BWD_CASE(3) GF_BWD_CASE(2) GF_BWD_CASE(1)
    default: break;
  }
}
#undef GF_BWD_CASE

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
  GF_DRAIN();        // (the previous strip step's last MFMA may still be reading ACC)
  gf_acc_zero();
  gf_fwd_chain<KB, NBT, SB>(c, pa);
  const double* q = pa + 16 * KB;   // (W_KB sits in the staged row block's last 16 columns)
  const double w0 = q[0], w1 = q[4], w2 = q[8], w3 = q[12];
  GF_DRAIN();
  Dst::set_prod_acc(w0, w1, w2, w3);
}

// backward step KB of one strip: Z_KB = W_KB^T (V_KB - sum_j L[j][KB]^T Z_j); the finished block also goes to the
// wave's LDS scratch (register image, lane-owned) for the gradient epilogue
template <int KB, int NBT, bool SB>
__device__ __forceinline__ void gf_bwd_strip(int last, const double* pb, double* zout, int lane) {
  using Dst = GfTile<gf_slot<NBT, SB>(KB)>;
  GF_DRAIN();
  Dst::copy_to_acc();   // V_KB: written by the forward pass, many steps (and barriers) ago
  gf_bwd_chain<KB, NBT, SB>(last, pb);
  const double* q = pb + 16 * KB * 16;
  const double w0 = q[0], w1 = q[64], w2 = q[128], w3 = q[192];
  GF_DRAIN();
  Dst::set_prod_acc(w0, w1, w2, w3);
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

// kernel value k (without outputscale) and h with dk/dl_d = h * delta_d^2 / l_d^3
template <int KIND>
__device__ __forceinline__ void gf_kernel_and_dfactor(double d2, const double* exptab, double& k, double& h) {
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

#define GF_STEP_CASE(K)                                                  \
  case K:                                                                \
    if constexpr (K < NBT) {                                             \
      if (fwd) gf_fwd_step<K, NBT>(cA, cB, pa, PA, lc, lq);              \
      else gf_bwd_step<K, NBT>(cA, cB, NB - 1, pb, zs, lane);            \
    }                                                                    \
    break;

template <int NBT, int KIND>
__global__ __launch_bounds__(NBT * 32) void gp_mll_grad_fused_kernel(MllGradFusedParams p) {
  constexpr int NP = 16 * NBT;
  constexpr int PA = NP + 2;             // pitch of a staged row block: lc * PA + lq hits 32 different 8-byte banks
  constexpr int NW = NBT / 2;            // waves
  constexpr int TPB = NBT * 32;
  constexpr int BUF = 16 * PA;           // doubles per staging buffer (a column block, NP x 16, fits as well)
  extern __shared__ double lds[];
  double* buf = lds;                     // [2][BUF]
  double* Xs = buf + 2 * BUF;            // [NP][GF_DP] points scaled by 1 / lengthscale, zero-padded to 8 dimensions
  double* als = Xs + NP * GF_DP;         // [NP] alpha (0 past n)
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
  const int kp = (NW < 2 || wave < NW / 2) ? wave : (NW + NW / 2 - 1 - wave);
  const int cA = kp, cB = NBT - 1 - kp;

  // ---- staging of one step's slice of L: registers now, LDS one step later
  double stg[8];
  const bool n_even = (N & 1) == 0;
  auto load_step = [&](int t) {   // t < NB: row block t (forward);  else column block 2 NB - 1 - t (backward)
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
  };
  auto store_step = [&](int t) {
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
  };

  load_step(0);
  exp2_table_init(exptab, tid);
  if (tid < GF_DMAX) invl[tid] = tid < D ? 1.0 / th[tid] : 0.0;
  __syncthreads();
  for (int e = tid; e < NP * GF_DP; e += TPB) {
    const int r = e / GF_DP, d = e - r * GF_DP;
    Xs[e] = (r < n && d < D) ? p.X[((size_t)task * N + r) * D + d] * invl[d] : 0.0;
  }
  for (int r = tid; r < NP; r += TPB) als[r] = r < n ? p.alpha[(size_t)task * N + r] : 0.0;
  store_step(0);
  load_step(1);
  __syncthreads();

  double accd[GF_DMAX];
#pragma unroll
  for (int d = 0; d < GF_DMAX; ++d) accd[d] = 0.0;
  double g_os = 0.0, g_noise = 0.0;
  double* zs = zs_all + wave * 512;

  for (int t = 0; t < 2 * NB; ++t) {
    // the slice for step t + 1 (loaded during step t - 1) goes into the buffer step t - 1 read from: everybody is past
    // the barrier that ended it; then the loads for step t + 2 start and have this whole step to land
    store_step(t + 1);
    load_step(t + 2);
    const bool fwd = t < NB;
    const int kb = fwd ? t : 2 * NB - 1 - t;
    const double* b = buf + (t & 1) * BUF;
    const double* pa = b + lc * PA + lq;     // forward A operand:  L[16 kb + lc][16 j + lq + 4 m]
    const double* pb = b + lq * 16 + lc;     // backward A operand: L[16 j + lq + 4 m][16 kb + lc]
    switch (kb) {
      GF_STEP_CASE(0) GF_STEP_CASE(1) GF_STEP_CASE(2) GF_STEP_CASE(3) GF_STEP_CASE(4) GF_STEP_CASE(5) GF_STEP_CASE(6)
      GF_STEP_CASE(7) GF_STEP_CASE(8) GF_STEP_CASE(9) GF_STEP_CASE(10) GF_STEP_CASE(11) GF_STEP_CASE(12) GF_STEP_CASE(13)
      GF_STEP_CASE(14) GF_STEP_CASE(15)
      default: break;
    }
    if (!fwd) {
      // ---- gradient epilogue for the K^-1 blocks (kb, cA) / (kb, cB) just finished: this lane owns rows
      // 16 kb + lq + 4 g, column 16 c + lc
      for (int s = 0; s < 2; ++s) {
        const int c = s ? cB : cA;
        if (kb < c) continue;
        const double* zt = zs + s * 256 + lane;
        const int col = 16 * c + lc;
        const double wgt = kb == c ? 1.0 : 2.0;   // off-diagonal blocks stand for both triangles
        double xc[GF_DMAX];
#pragma unroll
        for (int d = 0; d < GF_DMAX; ++d) xc[d] = Xs[col * GF_DP + d];
        const double ac = als[col];
#pragma clang loop unroll(disable)
        for (int g = 0; g < 4; ++g) {   // (rolled: four unrolled copies cost ~90 more live registers than the cap leaves)
          const int row = 16 * kb + lq + 4 * g;
          const double* xr = Xs + row * GF_DP;
          double d2 = 0.0;
#pragma unroll
          for (int d = 0; d < GF_DMAX; ++d) {
            const double df = xr[d] - xc[d];
            d2 = __builtin_fma(df, df, d2);
          }
          double k, h;
          gf_kernel_and_dfactor<KIND>(d2, exptab, k, h);
          const bool ok = row < n && col < n;
          const double Gv = ok ? wgt * (als[row] * ac - zt[64 * g]) : 0.0;
          const double GH = Gv * os * h;
          g_os = __builtin_fma(Gv, k, g_os);
          if (row == col) g_noise += Gv;
#pragma unroll
          for (int d = 0; d < GF_DMAX; ++d) {
            const double df = xr[d] - xc[d];
            accd[d] = __builtin_fma(GH, df * df, accd[d]);
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- D + 2 sums: over the wave, then over the waves in a fixed order; the task's totals go into tile slot 0
#pragma unroll
  for (int d = 0; d < GF_DMAX; ++d) {
    const double s = wave_sum_to_lane15(accd[d] * invl[d]);
    if (lane == 63) red[wave * 10 + d] = s;
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
    outp[tid] = s;
  }
  for (int e = D + 2 + tid; e < NT * (D + 2); e += TPB) outp[e] = 0.0;
}
#undef GF_STEP_CASE

}  // namespace scaml

#define SCAML_INSTANTIATE_GF(NBT)                                                                   \
  template __global__ void scaml::gp_mll_grad_fused_kernel<NBT, 0>(scaml::MllGradFusedParams);     \
  template __global__ void scaml::gp_mll_grad_fused_kernel<NBT, 1>(scaml::MllGradFusedParams);
SCAML_INSTANTIATE_GF(2)
SCAML_INSTANTIATE_GF(4)
SCAML_INSTANTIATE_GF(8)
SCAML_INSTANTIATE_GF(16)

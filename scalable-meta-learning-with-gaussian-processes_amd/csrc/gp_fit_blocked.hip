// gp_fit_blocked.hip — the fused fit for 256 < N <= 512 points per task as a 2 x 2 block factorisation, all on the
// device (gfx950): no host synchronisation, no framework ops between the launches.
//
//        [K11 K12]   [L11      ] [L11^T L21^T]      L11  = fused fit of block 1, in place     (gp_fit_blocked_kernel, 1 CU / task)
//        [K21 K22] = [L21   L22] [      L22^T]      L21^T = L11^-1 K12, r2 = y2 - L21 v1       (gp_blocked_solve_kernel, 4 CUs / task)
//                                                   S    = K22 + (noise + jitter) I - L21 L21^T (gp_blocked_syrk_kernel, 10 CUs / task)
//                                                   L22  = chol(S), alpha2 = S^-1 r2, in place  (gp_fit_blocked_kernel on S)
//                                                   alpha1 = L11^-T (v1 - L21^T alpha2), MLL    (scaml_blocked_finish_kernel)
// (v1 = L11^-1 y1: the first fit stops after its forward substitution, FIT_FORWARD_ONLY)
//
// The reference fits source GPs of up to 512 points (scamlgp/benchmarking/configurations/
// hartmann6_ablation_num_points_per_task.py:17-18) with gpytorch's ExactGP (kernel matrix -> psd_safe_cholesky ->
// cholesky_solve); a 512 x 512 task does not fit the registers + LDS of one CU, which is what the single-launch kernel
// lives on.  The jitter ladder of psd_safe_cholesky (one jitter value for the whole matrix of a failing task) is kept
// exact: every launch of a round is single-shot; the launches of rounds 1-3 are enqueued unconditionally, and
// scaml_blocked_round_kernel switches off (active[t] = 0) every task that is already factored, so that they cost an
// empty launch each (~2 us) when nothing failed.  Host side: scaml_gp_fit_blocked_f64 in csrc/scaml_host.cpp.
#include "../../include/scaml_gp.h"
#include "gf_tiles.hpp"
#include "gp_fit_params.h"

namespace scaml {

#ifdef BK_STAMPS   // developer build: cycle stamps of (part 0, wave 0) into the unused head of the task's r2 row (tools/dev_blk_stamps.py)
#define BK_STAMP(slot) do { if (stamp_on) p.r2[(size_t)task * p.N + (slot)] = (double)(__builtin_amdgcn_s_memtime() - stamp_t0); } while (0)
#define BK_STAMP_INIT(cond) const bool stamp_on = (cond); const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime()
#else
#define BK_STAMP(slot) do { } while (0)
#define BK_STAMP_INIT(cond) do { } while (0)
#endif

constexpr int BK_N1 = 256;   // block 1: what the single-launch kernel takes
constexpr int BK_NB1 = 16;

// ---- round prologue: per-task sizes of the two blocks, the round's jitter, and who still has to run -------------------
extern "C" __global__ void scaml_blocked_round_kernel(BlockedFitParams p) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.T) return;
  const double base = p.jitter_in ? p.jitter_in[t] : 0.0;
  if (p.round == 0) {
    int n = p.n_points ? p.n_points[t] : p.N;
    n = n < 0 ? 0 : (n > p.N ? p.N : n);
    p.n1[t] = n < BK_N1 ? n : BK_N1;
    p.n2[t] = n > BK_N1 ? n - BK_N1 : 0;
    p.active[t] = 1;
    p.jit_ladder[t] = 0.0;
    p.jit_cur[t] = base;
    p.info1[t] = 0;
    p.info2[t] = 0;
  } else {
    const bool failed = p.info1[t] > 0 || p.info2[t] > 0;
    p.active[t] = failed ? 1 : 0;
    if (failed) {
      const double j = p.round == 1 ? 1e-8 : (p.round == 2 ? 1e-7 : 1e-6);   // psd_safe_cholesky's escalation
      p.jit_ladder[t] = j;
      p.jit_cur[t] = base + j;
      p.info2[t] = 0;   // (block 2 is not attempted when block 1 fails again: no stale status)
    }
  }
}

// Workgroup -> (task, part) for the kernels that give a task several workgroups.  Workgroups are dealt to the 8 XCDs
// round-robin by their linear index; task t is kept on XCD t % 8 -- where its fused fits run (one workgroup per task,
// index t) -- so that what one launch leaves in that XCD's L2 is what the next one reads.  grid.x = 8 ceil(T/8) parts.
__device__ __forceinline__ bool bk_task_part(int parts, int T, int& task, int& part) {
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  task = (idx / parts) * 8 + xcd;
  part = idx % parts;
  return task < T;
}

// squared scaled distance of two staged points (difference form)
__device__ __forceinline__ double bk_sqdist(const double* a, const double* b, int D) {
  double s = 0.0;
  for (int d = 0; d < D; ++d) {
    const double u = a[d] - b[d];
    s = __builtin_fma(u, u, s);
  }
  return s;
}

// ---- L21^T = L11^-1 K12 by column strips, r2 = y2 - L21 v1 -------------------------------------------------------------
// One wave owns one strip (16 points of block 2): its 16 blocks Z_0..Z_15 live in hand-managed AGPRs in the MFMA C/D
// layout = the B-operand layout (csrc/gf_tiles.hpp), so a finished block feeds the next products from the registers it
// was computed in.  The four strips of a workgroup advance in lockstep over the block rows of L11; row block kb (with
// W_kb = L_kb,kb^-1 in place of the diagonal block) is staged once per workgroup in LDS by global_load_lds_dwordx4, two
// steps ahead.  Step kb:  ACC = K12 block (kernel function, VALU);  ACC -= L[kb][j] Z_j, j < kb;  Z_kb = W_kb ACC.
template <int KB>
__device__ __forceinline__ void bk_solve_step(const double* pa) {
  GfOps cur = gf_load_ops(pa, 4);
  gf_fwd_chain<KB, BK_NB1, false>(cur, 0, pa);
  GF_DRAIN();
  GfTile<gf_slot<BK_NB1, false>(KB)>::set_prod_acc(cur.a0, cur.a1, cur.a2, cur.a3);
}

// The finished strip leaves the registers twice: as register images for the Schur-complement kernel (workspace), and as
// rows of L21 -- transposed through the (by now free) LDS staging area in two halves of eight blocks, so that every
// global store is 1 KB of one row (the direct store would scatter 32-byte pieces over 16 rows 4 KB apart).
constexpr int BK_TP = 132;   // LDS pitch of a transposed half row (128 columns)
template <int J, int J1>
__device__ __forceinline__ void bk_stage_strip(double* tr, double* img, const double* v1s, double& macc, bool col_ok, int lc, int lq) {
  // tr[lc][16 (J mod 8) + lq + 4 g] = L21[16 c + lc][16 J + lq + 4 g] = Z_J[lq + 4 g][lc]
  d4_t z = GfTile<gf_slot<BK_NB1, false>(J)>::get();
  if (!col_ok) z = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int g = 0; g < 4; ++g) macc = __builtin_fma(z[g], v1s[16 * J + lq + 4 * g], macc);   // (L21 v1)[16 c + lc], this lane's rows
  asm volatile("" : "+v"(macc));   // (evaluate now: left alone, hipcc keeps all 64 values of the strip alive and sums at the very end)
  if (img) {   // the block as the matrix core holds it (lane-major, 32 bytes per lane)
    double2* q = reinterpret_cast<double2*>(img + J * 256);
    q[0] = double2{z[0], z[1]};
    q[1] = double2{z[2], z[3]};
  }
  double* t = tr + lc * BK_TP + 16 * (J & 7) + lq;
#pragma unroll
  for (int g = 0; g < 4; ++g) t[4 * g] = z[g];
  if constexpr (J + 1 < J1) bk_stage_strip<J + 1, J1>(tr, img, v1s, macc, col_ok, lc, lq);
}

#define BK_STEP_CASE(K) case K: bk_solve_step<K>(pa); break;

// Eight waves: waves 0-3 own the four strips (matrix core), waves 4-7 are their helpers on the same SIMDs -- helper w + 4
// evaluates the kernel-function block of strip w for the NEXT step into LDS and issues the staging loads, i.e. the VALU and
// memory work runs under the MFMA chain of the wave it shares a SIMD with.  One barrier per step.
// SMALLD (D <= 8): the points are staged zero-padded to 8 dimensions, the helper lane's own block-2 point sits in registers
// and the distance loops have a constant trip count.
template <int KIND, bool SMALLD>
__global__ __launch_bounds__(512) void gp_blocked_solve_kernel(BlockedFitParams p) {
  constexpr int NP = BK_N1, PA = NP + 2, BUF = 16 * PA, TPB = 512;
  extern __shared__ double lds[];
  const int N = p.N, D = p.D;
  const int DP = SMALLD ? 9 : (D | 1);   // odd pitch of the staged points
  const int DS = SMALLD ? 8 : D;         // staged dimensions
  double* buf = lds;                     // [3][BUF]: row blocks of L11, two steps ahead
  double* kbuf = buf + 3 * BUF;          // [2][4][256] kernel-function blocks (register images), one step ahead
  double* X1s = kbuf + 2 * 4 * 256;      // [NP][DP] block-1 points / lengthscale
  double* X2s = X1s + NP * DP;           // [64][DP] this workgroup's block-2 points / lengthscale
  double* a1s = X2s + 64 * DP;           // [NP] v1 = L11^-1 y1
  double* exptab = a1s + NP;             // [64]
  double* invl = exptab + 64;            // [DS]

  int task, sg;
  if (!bk_task_part((N - BK_N1 + 63) / 64, p.T, task, sg)) return;
  if (p.active[task] == 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool helper = wave >= 4;
  const int sw = wave & 3;               // strip of this wave (owner) / of the wave this one helps
  const int lc = lane & 15, lq = lane >> 4;
  const int n1 = p.n1[task], n2 = p.n2[task];
  const int NBT = (N + 15) / 16;
  const double* Lg = p.L + (size_t)task * N * N;
  const double* Wg = p.Linv_diag + (size_t)task * NBT * 256;
  const double* Xg = p.X + (size_t)task * N * D;
  const double* th = p.theta + (size_t)task * (D + 2);
  const double os = th[D];

  typedef __attribute__((address_space(1))) void gvoid_t;
  typedef __attribute__((address_space(3))) void lvoid_t;
  // row block kb of L11: 16 kb columns of L, then the 16 of W_kb (16-byte pieces; N % 16 == 0).  A helper wave moves four
  // rows: 4 wave-instructions for kb <= 7, 8 beyond (the step loop counts them in its s_waitcnt).
  auto dma_step = [&](int kb) {
    double* b = buf + (kb % 3) * BUF;
    const int nch = 8 * kb + 8;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = sw * 4 + rr;
      const double* lrow = Lg + (size_t)(16 * kb + r) * N;
      const double* wrow = Wg + (size_t)kb * 256 + r * 16 - 16 * kb;
      for (int c0 = 0; c0 < nch; c0 += 64) {
        const int ch = c0 + lane;
        const int chc = ch < nch ? ch : nch - 1;   // (lanes past the row re-fetch its last piece into a slot nobody reads: one instruction per 64 pieces, always)
        const double* src = (chc < 8 * kb ? lrow : wrow) + 2 * chc;
        __builtin_amdgcn_global_load_lds((const gvoid_t*)src, (lvoid_t*)(b + r * PA + 2 * c0), 16, 0, 0);
      }
    }
  };

  BK_STAMP_INIT(sg == 0 && tid == 0);
  if (helper) {
    dma_step(0);
    dma_step(1);
  }
  exp2_table_init(exptab, tid);
  if (tid < DS) invl[tid] = tid < D ? 1.0 / th[tid] : 0.0;
  __syncthreads();
  for (int e = tid; e < NP * DS; e += TPB) {
    const int r = e / DS, d = e - r * DS;
    X1s[r * DP + d] = (r < n1 && d < D) ? Xg[(size_t)r * D + d] * invl[d] : 0.0;
  }
  for (int e = tid; e < 64 * DS; e += TPB) {
    const int q = e / DS, d = e - q * DS, pt = 64 * sg + q;
    X2s[q * DP + d] = (pt < n2 && d < D) ? Xg[(size_t)(BK_N1 + pt) * D + d] * invl[d] : 0.0;
  }
  for (int r = tid; r < NP; r += TPB) a1s[r] = r < n1 ? p.alpha[(size_t)task * N + r] : 0.0;
  __syncthreads();
  if constexpr (SMALLD) {   // |x|^2 into the ninth slot: the squared distances of a block come off the matrix core (expanded form)
    for (int r = tid; r < NP + 64; r += TPB) {
      double* xr = r < NP ? X1s + r * 9 : X2s + (r - NP) * 9;
      double nrm = 0.0;
#pragma unroll
      for (int d = 0; d < 8; ++d) nrm = __builtin_fma(xr[d], xr[d], nrm);
      xr[8] = nrm;
    }
    __syncthreads();
  }

  const int c = 4 * sg + sw;                // the strip
  const int colpt = 16 * c + lc;            // its point on this lane (index inside block 2)
  const bool col_ok = colpt < n2;
  const double* xc = X2s + (16 * sw + lc) * DP;
  // SMALLD: B operands of the distance product, fixed for the kernel: X2[d = lq + 4 m][point lc], and the norm row
  double xb0 = 0.0, xb1 = 0.0, xb2 = 0.0;
  if constexpr (SMALLD) {
    xb0 = xc[lq];
    xb1 = xc[lq + 4];
    xb2 = lq == 0 ? 1.0 : (lq == 1 ? xc[8] : 0.0);
  }
  const bool strip_full = 16 * c + 16 <= n2;
  // scaled Matern / RBF coefficients (the outputscale rides on the polynomial)
  const double kc0 = os, kc1 = 2.2360679774997896964 * os, kc2 = (5.0 / 3.0) * os;
  // kernel-function block (kb, strip) as the matrix core wants it: lane (lc, lq) register g = os k(x1[16 kb + lq + 4 g], x2[16 c + lc])
  auto eval_block = [&](int kb) {
    double* out = kbuf + ((kb & 1) * 4 + sw) * 256 + lane;
    if constexpr (SMALLD) {
      // d2 = |a|^2 + |c|^2 - 2 a.c for the whole block from three MFMAs (A: -2 X1[point lc][d = lq + 4 m]; third k-step: the norms)
      const double* xa = X1s + (16 * kb + lc) * 9;
      d4_t d2v = {0.0, 0.0, 0.0, 0.0};
      d2v = __builtin_amdgcn_mfma_f64_16x16x4f64(-2.0 * xa[lq], xb0, d2v, 0, 0, 0);
      d2v = __builtin_amdgcn_mfma_f64_16x16x4f64(-2.0 * xa[lq + 4], xb1, d2v, 0, 0, 0);
      d2v = __builtin_amdgcn_mfma_f64_16x16x4f64(lq == 0 ? xa[8] : (lq == 1 ? 1.0 : 0.0), xb2, d2v, 0, 0, 0);
      d2v = gf_settle(d2v);
      if (strip_full && 16 * kb + 16 <= n1) {   // wave-uniform: nothing to mask
#pragma unroll
        for (int g = 0; g < 4; ++g) out[64 * g] = kernel_from_sqdist_scaled<KIND>(KIND == 0 ? vmax_f64(d2v[g], 0.0) : d2v[g], kc0, kc1, kc2, exptab);
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * kb + lq + 4 * g;
          const double k = kernel_from_sqdist_scaled<KIND>(KIND == 0 ? vmax_f64(d2v[g], 0.0) : d2v[g], kc0, kc1, kc2, exptab);
          out[64 * g] = (row < n1 && col_ok) ? k : 0.0;
        }
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = 16 * kb + lq + 4 * g;
        const double k = os * kernel_from_sqdist<KIND>(bk_sqdist(X1s + row * DP, xc, D), exptab);
        out[64 * g] = (row < n1 && col_ok) ? k : 0.0;
      }
    }
  };
  if (helper) {
    eval_block(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  BK_STAMP(0);
  for (int kb = 0; kb < BK_NB1; ++kb) {
    if (helper) {
      if (kb + 2 < BK_NB1) dma_step(kb + 2);  // into the buffer step kb - 1 read from: everybody is past the barrier that ended it
      if (kb + 1 < BK_NB1) eval_block(kb + 1);
      // this wave's pieces of row block kb + 1 have landed; those of kb + 2 (4 or 8 instructions, issued above) may be in flight
      if (kb + 2 >= BK_NB1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (kb + 2 >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      const double* kin = kbuf + ((kb & 1) * 4 + sw) * 256 + lane;
      const double k0 = kin[0], k1 = kin[64], k2 = kin[128], k3 = kin[192];
      GF_DRAIN();   // (the previous step's closing product may still be reading ACC)
      gf_acc_set(k0, k1, k2, k3);
      const double* pa = buf + (kb % 3) * BUF + lc * PA + lq;   // A operand: L[16 kb + lc][16 j + lq + 4 m]
      switch (kb) {
        BK_STEP_CASE(0) BK_STEP_CASE(1) BK_STEP_CASE(2) BK_STEP_CASE(3) BK_STEP_CASE(4) BK_STEP_CASE(5) BK_STEP_CASE(6) BK_STEP_CASE(7)
        BK_STEP_CASE(8) BK_STEP_CASE(9) BK_STEP_CASE(10) BK_STEP_CASE(11) BK_STEP_CASE(12) BK_STEP_CASE(13) BK_STEP_CASE(14)
        BK_STEP_CASE(15)
        default: break;
      }
      BK_STAMP(2 + 4 * kb);   // chain issued
    }
    __syncthreads();
    BK_STAMP(4 + 4 * kb);   // barrier passed
  }
  if (helper) return;
  GF_DRAIN();
  // the strip goes to memory as rows of L21 (and, on request, zeros into the mirrored block of the upper triangle)
  const int N2 = N - BK_N1;
  const bool in_matrix = colpt < N2;
  double* Lw = p.L + (size_t)task * N * N;
  double* urow = in_matrix ? Lw + BK_N1 + colpt : nullptr;
  double* img = in_matrix || 16 * c < N2 ? p.Vimg + ((size_t)task * BK_NB1 + c) * BK_NB1 * 256 + lane * 4 : nullptr;
  double* tr = buf + wave * (16 * BK_TP);   // (everybody is past the barrier that ended the last step: the staging area is free)
  const bool zero_upper = (p.flags & SCAML_FIT_ZERO_UPPER) != 0;
  const bool strip_in = 16 * c < N2;        // (N2 is a multiple of 16: a strip is inside the matrix or not at all)
  double macc = 0.0;                        // lane's share of (L21 v1) at its point
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h == 0) bk_stage_strip<0, 8>(tr, img, a1s, macc, col_ok, lc, lq);
    else bk_stage_strip<8, 16>(tr, img, a1s, macc, col_ok, lc, lq);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (strip_in) {
#pragma unroll 4
      for (int i = 0; i < 16; ++i) {
        const double2 v = *reinterpret_cast<const double2*>(tr + i * BK_TP + 2 * lane);
        *reinterpret_cast<double2*>(Lw + (size_t)(BK_N1 + 16 * c + i) * N + 128 * h + 2 * lane) = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (zero_upper && urow) {
    for (int r = lq; r < BK_N1; r += 4) urow[(size_t)r * N] = 0.0;   // the mirrored block of the upper triangle: 128-byte row pieces
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BK_STAMP(70);   // strip stored
  macc = sum_lane_groups(macc);
  if (lq == 0 && in_matrix) p.r2[(size_t)task * N + BK_N1 + colpt] = col_ok ? p.y[(size_t)task * N + BK_N1 + colpt] - macc : 0.0;
}
#undef BK_STEP_CASE

// ---- S = K22 + (noise + jitter) I - L21 L21^T, lower 64 x 64 tiles, one workgroup (8 waves) each -------------------------
// Wave (w, ks) of a tile takes block column w (16 columns), the four row blocks and half ks of the summation index; both
// MFMA operands are the blocks of L21^T as the strip solve left them in the workspace (register images: a wave load is
// 2 KB of consecutive memory -- rows of L21 itself are 4 KB apart, 16 of them per load land on one L2 channel).
// The loads of four block rows (40 x 16 bytes per lane) are issued together: two memory round trips per wave.  The two halves
// swap partial sums through LDS; each finishes two of the four blocks (kernel function, diagonal, store).
template <int KIND>
__global__ __launch_bounds__(512, 4) void gp_blocked_syrk_kernel(BlockedFitParams p) {
  extern __shared__ double lds[];
  const int N = p.N, D = p.D, N2 = N - BK_N1;
  const int DP = D | 1;
  double* xch = lds;                // [8 waves][2 blocks][4][64] partial sums handed to the partner wave
  double* Xr = xch + 8 * 512;       // [64][DP] the tile's row points / lengthscale
  double* Xc = Xr + 64 * DP;        // [64][DP] column points
  double* exptab = Xc + 64 * DP;    // [64]
  double* invl = exptab + 64;       // [D]
  const int nt = (N2 + 63) / 64;
  int task, tile;
  if (!bk_task_part(nt * (nt + 1) / 2, p.T, task, tile)) return;
  if (p.active[task] == 0) return;
  int bi = 0, rem = tile;           // linear index over the lower triangle of tiles, row by row
  while (rem > bi) { rem -= bi + 1; ++bi; }
  const int bj = rem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = wave & 3, ks = wave >> 2;
  const int lc = lane & 15, lq = lane >> 4;
  const double* Xg = p.X + (size_t)task * N * D;
  const double* th = p.theta + (size_t)task * (D + 2);
  const double os = th[D], diag_add = th[D + 1] + p.jit_cur[task];
  exp2_table_init(exptab, tid);
  if (tid < D) invl[tid] = 1.0 / th[tid];
  __syncthreads();
  for (int e = tid; e < 64 * D; e += 512) {
    const int q = e / D, d = e - q * D;
    const int pr = 64 * bi + q, pc = 64 * bj + q;
    Xr[q * DP + d] = pr < N2 ? Xg[(size_t)(BK_N1 + pr) * D + d] * invl[d] : 0.0;
    Xc[q * DP + d] = pc < N2 ? Xg[(size_t)(BK_N1 + pc) * D + d] * invl[d] : 0.0;
  }
  const int cb = 4 * bj + w;
  const bool live = 16 * cb < N2;   // (a block column past the matrix: the wave only keeps the barriers company)
  // block (kb, strip) of L21^T as the strip solve left it: 64 lanes x 32 bytes, register m of the image = operand of step m
  // on either side (A = block^T in the A layout, B = block in the B layout: the same registers)
  auto imgp = [&](int strip) { return p.Vimg + (((size_t)task * BK_NB1 + (strip < BK_NB1 ? strip : BK_NB1 - 1)) * BK_NB1 + 8 * ks) * 256 + lane * 4; };
  const double* bp = imgp(cb);
  const double* ap[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) ap[r] = imgp(4 * bi + r);
  d4_t acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = d4_t{0.0, 0.0, 0.0, 0.0};
  if (live) {
#pragma unroll 1
    for (int k0 = 0; k0 < 8; k0 += 4) {
      double2 bv[4][2], av[4][4][2];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bv[kk][0] = *reinterpret_cast<const double2*>(bp + 256 * (k0 + kk));
        bv[kk][1] = *reinterpret_cast<const double2*>(bp + 256 * (k0 + kk) + 2);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          av[kk][r][0] = *reinterpret_cast<const double2*>(ap[r] + 256 * (k0 + kk));
          av[kk][r][1] = *reinterpret_cast<const double2*>(ap[r] + 256 * (k0 + kk) + 2);
        }
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk][r][0].x, bv[kk][0].x, acc[r], 0, 0, 0);
          acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk][r][0].y, bv[kk][0].y, acc[r], 0, 0, 0);
          acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk][r][1].x, bv[kk][1].x, acc[r], 0, 0, 0);
          acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk][r][1].y, bv[kk][1].y, acc[r], 0, 0, 0);
        }
      }
    }
  }
  // half ks finishes row blocks 2 ks, 2 ks + 1; the other two go to the partner wave (w, 1 - ks)
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = gf_settle(acc[r]);
  {
    double* out = xch + wave * 512 + lane;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const d4_t a = ks == 0 ? acc[2 + q] : acc[q];
#pragma unroll
      for (int g = 0; g < 4; ++g) out[(4 * q + g) * 64] = a[g];
    }
  }
  __syncthreads();
  if (!live) return;
  double* Sg = p.S + (size_t)task * N2 * N2;
  const int col = 16 * cb + lc;
  const double* xc = Xc + (16 * w + lc) * DP;
  const double* in = xch + (wave ^ 4) * 512 + lane;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = 2 * ks + q, rb = 4 * bi + r;
    if (rb < cb || 16 * rb >= N2) continue;   // strictly upper block / past the matrix (wave-uniform)
    const d4_t a = ks == 0 ? acc[q] : acc[2 + q];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = 16 * rb + lq + 4 * g;
      double k = os * kernel_from_sqdist<KIND>(bk_sqdist(Xr + (16 * r + lq + 4 * g) * DP, xc, D), exptab);
      if (row == col) k += diag_add;
      if (row < N2 && col < N2) Sg[(size_t)row * N2 + col] = k - (a[g] + in[(4 * q + g) * 64]);
    }
  }
}

// ---- alpha1 = L11^-T (v1 - L21^T alpha2), the scalars, the status ---------------------------------------------------
// One workgroup of 1024 threads per task; thread (q, j) = (tid / 256, tid % 256).  u = L11^-T w runs by blocks from the
// bottom: u_kb = W_kb^T w_kb, then w_j -= sum_r L[16 kb + r][j] u_kb[r] for the columns j left of the block.  Nothing on
// that 16-step chain waits for global memory: all W_kb sit in LDS, thread (q, j) holds rows 4 q .. 4 q + 3 of the block
// rows in a ring of registers filled eight steps ahead, every wave forms u_kb for itself (no hand-over through LDS), and
// the partial sums of the four quarters meet in LDS atomics: one barrier per step.  (A ragged block 1 rides on the identity
// padding of W_kb; rows of L at or past n1 are never written by the fit -- include/scaml_gp.h -- and may hold anything, NaN bit
// patterns included: the ring is loaded before n1 is known, so the step masks those rows where it uses them.)
template <int KB>
__device__ __forceinline__ void bk_finish_load(double (&dst)[4], const double* Lg, int N, int q, int j) {
#pragma unroll
  for (int i = 0; i < 4; ++i) dst[i] = j < 16 * KB ? Lg[(size_t)(16 * KB + 4 * q + i) * N + j] : 0.0;
}

template <int KB>
__device__ __forceinline__ void bk_finish_step(const double (&lrow)[4], const double* Wl, const double* w, double (*pw)[BK_N1], double* u, double& mine,
                                               int q, int j, int lane, int n1) {
  __syncthreads();   // the four quarters' running sums for block KB are in pw
  // u_kb[c] on lane (c = lane % 16, any lane / 16): every wave computes all of it
  const int c = lane & 15, rg = lane >> 4;
  double xv = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 16 * KB + 4 * rg + i;
    const double wr = w[r] - ((pw[0][r] + pw[1][r]) + (pw[2][r] + pw[3][r]));   // fixed order: bit-reproducible
    xv = __builtin_fma(Wl[KB * 256 + (4 * rg + i) * 16 + c], wr, xv);
  }
  xv = sum_lane_groups(xv);
  if (q == 0 && j < 16) u[16 * KB + j] = xv;
  if constexpr (KB > 0) {
    // thread (q, j) keeps ITS share of sum_r L[r][j] u[r] over all steps in a register and hands it over when column j's block is next
#pragma unroll
    for (int i = 0; i < 4; ++i) mine = __builtin_fma(16 * KB + 4 * q + i < n1 ? lrow[i] : 0.0, readlane_f64(xv, 4 * q + i), mine);
    if (j >= 16 * (KB - 1) && j < 16 * KB) pw[q][j] = mine;
  }
}

extern "C" __global__ __launch_bounds__(1024) void scaml_blocked_finish_kernel(BlockedFitParams p) {
  __shared__ double Wl[BK_NB1 * 256];
  __shared__ double part[4][BK_N1];
  __shared__ double pw[4][BK_N1];
  __shared__ double a2s[BK_N1];
  __shared__ double w[BK_N1];
  __shared__ double u[BK_N1];
  const int task = blockIdx.x, tid = threadIdx.x;
  const int N = p.N, N2 = N - BK_N1;
  const double* Lg = p.L + (size_t)task * N * N;
  const double* Wg = p.Linv_diag + (size_t)task * ((N + 15) / 16) * 256;
  double* al = p.alpha + (size_t)task * N;
  const int j = tid & 255, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 8);
  // Everything that depends on nothing but the task index starts its way FIRST (the kernel is a chain of memory round trips at one
  // workgroup per task): the ring's first four block rows (15 .. 12), every W_kb, and the first 16 of this thread's 64 rows of
  // L21 for the mat-vec (rows of L21 past n2 are zeros, written by the strip solve: no mask needed, only the matrix bound).
  double ring[8][4];
  bk_finish_load<15>(ring[7], Lg, N, q, j); bk_finish_load<14>(ring[6], Lg, N, q, j);
  bk_finish_load<13>(ring[5], Lg, N, q, j); bk_finish_load<12>(ring[4], Lg, N, q, j);
  double wl[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wl[i] = Wg[tid + 1024 * i];
  const int r0 = q * 64;
  const double* col = Lg + (size_t)(BK_N1 + r0) * N + j;
  double cv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) cv[k] = r0 + k < N2 ? col[(size_t)k * N] : 0.0;
  const int n1 = p.n1[task], n2 = p.n2[task], n = n1 + n2;
  const int i1 = p.info1[task], i2 = p.info2[task];
  const int info = i1 > 0 ? i1 : (i2 > 0 ? i2 + BK_N1 : 0);
  if (tid == 0) {
    p.info[task] = info;
    if (p.jitter_used) p.jitter_used[task] = p.jit_ladder[task];
    const double nan = __builtin_nan("");
    const double qd = p.q12[task] + p.q12[2 * p.T + task], ld = p.q12[p.T + task] + p.q12[3 * p.T + task];
    if (p.quad) p.quad[task] = info ? nan : qd;
    if (p.logdet) p.logdet[task] = info ? nan : ld;
    if (p.mll) p.mll[task] = info ? nan : (n > 0 ? -0.5 * (qd + ld + n * 1.8378770664093454836) / n : 0.0);
  }
  if (info) return;
  BK_STAMP_INIT(tid == 0);
  if (tid < BK_N1) a2s[tid] = tid < n2 ? al[BK_N1 + tid] : 0.0;
  __syncthreads();
  {
    // w = L21^T alpha2: thread (q, j) sums its quarter of the rows of column j (rows read coalesced, 16 in flight)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 1
    for (int rb = 0; rb < 64; rb += 16) {
      if (r0 + rb >= n2) break;
      if (rb > 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) cv[k] = r0 + rb + k < N2 ? col[(size_t)(rb + k) * N] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 16; k += 4) {
        s0 = __builtin_fma(cv[k], a2s[r0 + rb + k], s0);
        s1 = __builtin_fma(cv[k + 1], a2s[r0 + rb + k + 1], s1);
        s2 = __builtin_fma(cv[k + 2], a2s[r0 + rb + k + 2], s2);
        s3 = __builtin_fma(cv[k + 3], a2s[r0 + rb + k + 3], s3);
      }
    }
    part[q][j] = (s0 + s1) + (s2 + s3);
  }
  // (block rows 11 .. 8 are needed four steps from now)
  bk_finish_load<11>(ring[3], Lg, N, q, j); bk_finish_load<10>(ring[2], Lg, N, q, j);
  bk_finish_load<9>(ring[1], Lg, N, q, j); bk_finish_load<8>(ring[0], Lg, N, q, j);
#pragma unroll
  for (int i = 0; i < 4; ++i) Wl[tid + 1024 * i] = wl[i];
  __syncthreads();
  if (tid < BK_N1) w[tid] = (tid < n1 ? al[tid] : 0.0) - ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));   // v1 - L21^T alpha2
  pw[q][j] = 0.0;
  double mine = 0.0;
  BK_STAMP(100);   // mat-vec done
  bk_finish_step<15>(ring[7], Wl, w, pw, u, mine, q, j, lane, n1); bk_finish_load<7>(ring[7], Lg, N, q, j);
  bk_finish_step<14>(ring[6], Wl, w, pw, u, mine, q, j, lane, n1); bk_finish_load<6>(ring[6], Lg, N, q, j);
  bk_finish_step<13>(ring[5], Wl, w, pw, u, mine, q, j, lane, n1); bk_finish_load<5>(ring[5], Lg, N, q, j);
  bk_finish_step<12>(ring[4], Wl, w, pw, u, mine, q, j, lane, n1); bk_finish_load<4>(ring[4], Lg, N, q, j);
  bk_finish_step<11>(ring[3], Wl, w, pw, u, mine, q, j, lane, n1); bk_finish_load<3>(ring[3], Lg, N, q, j);
  bk_finish_step<10>(ring[2], Wl, w, pw, u, mine, q, j, lane, n1); bk_finish_load<2>(ring[2], Lg, N, q, j);
  bk_finish_step<9>(ring[1], Wl, w, pw, u, mine, q, j, lane, n1); bk_finish_load<1>(ring[1], Lg, N, q, j);
  bk_finish_step<8>(ring[0], Wl, w, pw, u, mine, q, j, lane, n1);
  bk_finish_step<7>(ring[7], Wl, w, pw, u, mine, q, j, lane, n1); bk_finish_step<6>(ring[6], Wl, w, pw, u, mine, q, j, lane, n1);
  bk_finish_step<5>(ring[5], Wl, w, pw, u, mine, q, j, lane, n1); bk_finish_step<4>(ring[4], Wl, w, pw, u, mine, q, j, lane, n1);
  bk_finish_step<3>(ring[3], Wl, w, pw, u, mine, q, j, lane, n1); bk_finish_step<2>(ring[2], Wl, w, pw, u, mine, q, j, lane, n1);
  bk_finish_step<1>(ring[1], Wl, w, pw, u, mine, q, j, lane, n1); bk_finish_step<0>(ring[0], Wl, w, pw, u, mine, q, j, lane, n1);
  __syncthreads();
  BK_STAMP(101);   // chain done
  if (tid < n1) al[tid] = u[tid];
}

}  // namespace scaml

template __global__ void scaml::gp_blocked_solve_kernel<0, false>(scaml::BlockedFitParams);
template __global__ void scaml::gp_blocked_solve_kernel<1, false>(scaml::BlockedFitParams);
template __global__ void scaml::gp_blocked_solve_kernel<0, true>(scaml::BlockedFitParams);
template __global__ void scaml::gp_blocked_solve_kernel<1, true>(scaml::BlockedFitParams);
template __global__ void scaml::gp_blocked_syrk_kernel<0>(scaml::BlockedFitParams);
template __global__ void scaml::gp_blocked_syrk_kernel<1>(scaml::BlockedFitParams);

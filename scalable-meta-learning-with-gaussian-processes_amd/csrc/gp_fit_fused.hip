// gp_fit_fused.hip — fused "task-posterior" kernel for gfx950 (MI355X):
//   K = os * k(X/l, X/l) + (noise + jitter) I  ->  L L^T = K  ->  v = L^-1 y, alpha = L^-T v
//   quad = v.v, logdet = 2 sum log L_ii, mll = -(quad + logdet + n log 2pi) / (2 n)
// for a stack of T independent tasks, one workgroup per task.
//
// Replaces (for the whole stack at once) the reference's per-task chain
//   scamlgp/model.py:176-188 -> scamlgp/utils.py:171-177 -> gpytorch ExactMarginalLogLikelihood
//   -> linear_operator psd_safe_cholesky -> torch.linalg.cholesky_ex / solve_triangular.
//
// Design.  The kernel matrix never exists in memory: the trailing matrix of a
// right-looking Cholesky lives in MFMA accumulator registers.  The lower triangle is cut into
// 16x16 tiles; tile t (column-major over the triangle) belongs to update wave t % WU, slot t / WU,
// in the C/D layout of v_mfma_f64_16x16x4_f64 (col = lane & 15, row = (lane >> 4) + 4 * reg), and
// every lane evaluates the kernel function for the four elements it owns straight into it.
// Waves are specialised: WU "update" waves own tiles; one "panel" wave owns none and factors the
// 16x16 diagonal blocks.  Per 16-column panel k (diagonal block factored, rows below raw in LDS):
//   T   owners of column k form L_ik = A_ik W^T with 4 MFMAs per tile (W = L_kk^-1 from the panel
//       wave), put the final tiles back into the LDS panel and write them to HBM from registers
//       (128-byte row segments; the mirrored upper tile is written as zeros by the same lanes)
//   U1  tiles of column k+1 get their rank-16 update first and are spilled to the other LDS panel
//       (meanwhile the panel wave folds panel k into the running right-hand side y)
//   U2  all remaining tiles get the rank-16 update (4 MFMAs per tile, operands from the LDS
//       panel) WHILE the panel wave factors diagonal block k+1 (one-panel lookahead): 16 rank-1
//       MFMA updates on the symmetric block, the next pivot computed ahead on the VALU so the
//       64-cycle MFMA latency stays off the pivot chain; a second accumulator receives the same
//       row operations and ends as L_kk^-1, which also gives v_k = L_kk^-1 y_k as a mat-vec.
// alpha comes from a blocked back-substitution over the L tiles still held in registers.  A
// failed pivot restarts the task in-kernel with the next jitter (1e-8, 1e-7, 1e-6), as
// linear_operator's psd_safe_cholesky does on the host.
#include "scaml_common.hpp"
#include "../../include/scaml_gp.h"
#include "gp_fit_params.h"

#ifdef SCAML_STAMPS
// Diagnostic build only (python __graft_entry__.py --stamps): update wave 0 and the panel wave of
// every workgroup accumulate s_memtime deltas per phase into a caller-provided buffer [T][2][16].
// Never compiled into libscaml_hip.so.
__device__ long long* g_stamp_buf = nullptr;
#define STAMP_DECL long long st_prev = __builtin_amdgcn_s_memtime(), st_acc[16] = {0}
#define STAMP(i) do { long long st_now = __builtin_amdgcn_s_memtime(); st_acc[i] += st_now - st_prev; st_prev = st_now; } while (0)
#define STAMP_FLUSH(task) do { if (g_stamp_buf && (threadIdx.x == 0 || threadIdx.x == blockDim.x - 64)) for (int i_ = 0; i_ < 16; ++i_) g_stamp_buf[((task) * 2 + (threadIdx.x != 0)) * 16 + i_] = st_acc[i_]; } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH(task)
#endif

namespace scaml {



// LDS panel layout: PANEL[row * PP + c], PP = 17 doubles.  With this pitch the MFMA operand reads
// (lane -> row 16t + (lane & 15), column 4m + (lane >> 4)), the C/D-layout tile spills/reloads
// (lane -> row (lane >> 4) + 4g, column lane & 15) and the per-row reads are all bank-conflict free.
constexpr int PP = 17;

__device__ __forceinline__ int opaque_s(int v) {
  asm volatile("" : "+s"(v));  // keeps per-slot address arithmetic from being hoisted out of the panel loop
  return v;
}

// ---- panel wave: Cholesky of the 16x16 diagonal block k, its inverse, and v_k = L_kk^-1 y_k ----
// panel rows 16k..16k+15 hold the symmetric trailing block (both triangles).  The block sits in
// one MFMA accumulator (C/D layout); step c scales row c (= column c by symmetry, already in
// operand position on lane group c & 3), applies the rank-1 update with ONE MFMA and forms the
// next pivot ahead of it on the VALU, so the 64-cycle MFMA latency is off the pivot chain.  A
// second accumulator R (initially I) receives the same row operations and ends as W = L_kk^-1.
// Lanes outside the active group store to a per-lane trash slot instead of being masked off
// (no exec juggling in the 16-step chain).
// Outputs: panel rows <- L_kk (zero above the diagonal), Wk[c * PP + j] = W[c][j], vv.
// Returns 0 or the 1-based global index of the first non-positive / out-of-range pivot.
__device__ __forceinline__ int potf2_inv_block(double* panel, double* Wk, double* vv, double* trash,
                                               const double* ytil, int k, int lane) {
  const int lc = lane & 15, lq = lane >> 4;
  d4_t a, R;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    a[g] = panel[(16 * k + lq + 4 * g) * PP + lc];
    R[g] = (lc == lq + 4 * g) ? 1.0 : 0.0;
  }
  // LDS element offsets relative to `panel` (the trash slot is addressed through the same base)
  const int off_trash = (int)(trash - panel) + lane;
  const int off_prow = (16 * k + lc) * PP;    // + c   -> L[lc][c]
  const int off_w = (int)(Wk - panel) + lc;   // + c*PP -> W[c][lc]
  // Pivot chain per step: rsqrt(dpiv) -> ri^2 -> next dpiv.  Everything else hangs off it: the raw
  // row values are masked and read across lanes as soon as the previous MFMA lands (before ri is
  // known), the range check only feeds `bad`, the two MFMAs and the LDS stores trail behind.
  int bad = 0;
  double dpiv = readlane_f64(a[0], 0);
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int g = c & 3, rg = c >> 2;
    const bool mine = lq == g;
    const bool low = (unsigned)(lane - (16 * g + c)) < (unsigned)(16 - c);  // mine && lc >= c
    const double am = low ? a[rg] : 0.0;    // raw column c (row c of the symmetric block), masked
    const double Rm = mine ? R[rg] : 0.0;
#ifndef PROBE_NO_OOB
    // non-positive, NaN (or too small for the f32-seeded rsqrt): remember the first failing step;
    // the garbage it produces afterwards is never used
    if (!(dpiv >= 1e-30) && bad == 0) bad = 16 * k + c + 1;
#endif
    const double ri = rsqrt_seeded(dpiv);
    if (c < 15) {
      const int g1 = (c + 1) & 3, rg1 = (c + 1) >> 2;
      const double acn = readlane_f64(a[rg], 16 * g + c + 1);      // a[c+1][c] before scaling
      const double anext = readlane_f64(a[rg1], 16 * g1 + c + 1);  // a[c+1][c+1]
      dpiv = __builtin_fma(-(acn * acn), ri * ri, anext);
    }
    const double lcol = am * ri;   // L[lc][c]
    const double wrow = Rm * ri;   // W[c][lc]
    if (c < 15) {
      a = __builtin_amdgcn_mfma_f64_16x16x4f64(lcol, lcol, a, 0, 0, 1);   // blgp = 1: A operand negated
#ifndef PROBE_NO_INV
      R = __builtin_amdgcn_mfma_f64_16x16x4f64(lcol, wrow, R, 0, 0, 1);
#endif
    }
#ifndef PROBE_NO_STORE
    panel[mine ? off_prow + c : off_trash] = lcol;
    panel[mine ? off_w + c * PP : off_trash] = wrow;
#endif
  }
  // v_k = W y_k: lane (lc, lq) takes the four terms c = 4 lq .. 4 lq + 3 of row lc, then the lane
  // groups are summed
  double v = 0.0;
#pragma unroll
  for (int c = 0; c < 4; ++c) v = __builtin_fma(Wk[lc * PP + 4 * lq + c], ytil[16 * k + 4 * lq + c], v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  if (lq == 0) vv[16 * k + lc] = v;
  return bad;
}

// ---- hand-managed accumulator tiles ------------------------------------------------------------
// The trailing-matrix tiles live in AGPRs a[0:159] that the COMPILER NEVER SEES as values: every
// access is an asm statement naming the physical registers (csrc/tile_regs.inc) and listing them
// as clobbers.  Why: with the tiles as C++ values (builtin MFMA, or asm with tied "+v"/"+a"
// operands) hipcc's register allocator kept a second VGPR copy of tiles around the wave-uniform
// branches of the panel loop, ran out of registers and spilled tiles to scratch every phase
// (568-876 bytes/lane, 5x slower; profiles/r01_notes.md).  Hand-managed, the compiler only has
// to fit its own temporaries into the VGPR half (capped with amdgpu_num_vgpr so that VGPRs + 160
// AGPRs stay within the 256 registers a wave gets at 2 waves/SIMD).
// Hazards hipcc cannot see inside asm (CDNA3/4 ISA 4.5, DGEMM rows): dependent MFMAs on one tile
// issue back to back; MFMA_DRAIN (19 wait states) must separate the last MFMA on a tile from any
// v_accvgpr_read/write of it; 2 wait states cover a VALU-written operand feeding an MFMA.
#include "tile_regs.inc"

#define SCAML_CLOB8(r0, r1, r2, r3, r4, r5, r6, r7) "a" #r0, "a" #r1, "a" #r2, "a" #r3, "a" #r4, "a" #r5, "a" #r6, "a" #r7
#define SCAML_TILE(r0, r7) "a[" #r0 ":" #r7 "]"

// tile <- (d0, d1, d2, d3): element g is the value for row (lane >> 4) + 4g, column lane & 15
#define TILE_SET(r0, r1, r2, r3, r4, r5, r6, r7, d0, d1, d2, d3)                                   \
  asm volatile("v_accvgpr_write_b32 a" #r0 ", %0\n\tv_accvgpr_write_b32 a" #r1 ", %1\n\t"         \
               "v_accvgpr_write_b32 a" #r2 ", %2\n\tv_accvgpr_write_b32 a" #r3 ", %3\n\t"         \
               "v_accvgpr_write_b32 a" #r4 ", %4\n\tv_accvgpr_write_b32 a" #r5 ", %5\n\t"         \
               "v_accvgpr_write_b32 a" #r6 ", %6\n\tv_accvgpr_write_b32 a" #r7 ", %7"             \
               :                                                                                   \
               : "v"(__double2loint(d0)), "v"(__double2hiint(d0)), "v"(__double2loint(d1)),        \
                 "v"(__double2hiint(d1)), "v"(__double2loint(d2)), "v"(__double2hiint(d2)),        \
                 "v"(__double2loint(d3)), "v"(__double2hiint(d3))                                  \
               : SCAML_CLOB8(r0, r1, r2, r3, r4, r5, r6, r7))

// (d0, d1, d2, d3) <- tile
#define TILE_GET(r0, r1, r2, r3, r4, r5, r6, r7, d0, d1, d2, d3)                                   \
  do {                                                                                             \
    int l0_, h0_, l1_, h1_, l2_, h2_, l3_, h3_;                                                    \
    asm volatile("v_accvgpr_read_b32 %0, a" #r0 "\n\tv_accvgpr_read_b32 %1, a" #r1 "\n\t"         \
                 "v_accvgpr_read_b32 %2, a" #r2 "\n\tv_accvgpr_read_b32 %3, a" #r3 "\n\t"         \
                 "v_accvgpr_read_b32 %4, a" #r4 "\n\tv_accvgpr_read_b32 %5, a" #r5 "\n\t"         \
                 "v_accvgpr_read_b32 %6, a" #r6 "\n\tv_accvgpr_read_b32 %7, a" #r7                 \
                 : "=v"(l0_), "=v"(h0_), "=v"(l1_), "=v"(h1_), "=v"(l2_), "=v"(h2_), "=v"(l3_),    \
                   "=v"(h3_));                                                                     \
    d0 = __hiloint2double(h0_, l0_);                                                               \
    d1 = __hiloint2double(h1_, l1_);                                                               \
    d2 = __hiloint2double(h2_, l2_);                                                               \
    d3 = __hiloint2double(h3_, l3_);                                                               \
  } while (0)

// tile -= A * B over four k-steps (rank-16 update): neg:[1,0,0] negates the A operand in the MFMA
#define TILE_MFMA4_SUB(r0, r1, r2, r3, r4, r5, r6, r7, a0, a1, a2, a3, b0, b1, b2, b3)             \
  asm volatile("s_nop 1\n\t"                                                                       \
               "v_mfma_f64_16x16x4_f64 " SCAML_TILE(r0, r7) ", %0, %4, " SCAML_TILE(r0, r7) " neg:[1,0,0]\n\t" \
               "v_mfma_f64_16x16x4_f64 " SCAML_TILE(r0, r7) ", %1, %5, " SCAML_TILE(r0, r7) " neg:[1,0,0]\n\t" \
               "v_mfma_f64_16x16x4_f64 " SCAML_TILE(r0, r7) ", %2, %6, " SCAML_TILE(r0, r7) " neg:[1,0,0]\n\t" \
               "v_mfma_f64_16x16x4_f64 " SCAML_TILE(r0, r7) ", %3, %7, " SCAML_TILE(r0, r7) " neg:[1,0,0]"     \
               :                                                                                   \
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3)            \
               : SCAML_CLOB8(r0, r1, r2, r3, r4, r5, r6, r7))

// tile = A * B over four k-steps (first MFMA starts from the inline constant 0)
#define TILE_MFMA4_SET(r0, r1, r2, r3, r4, r5, r6, r7, a0, a1, a2, a3, b0, b1, b2, b3)             \
  asm volatile("s_nop 1\n\t"                                                                       \
               "v_mfma_f64_16x16x4_f64 " SCAML_TILE(r0, r7) ", %0, %4, 0\n\t"                      \
               "v_mfma_f64_16x16x4_f64 " SCAML_TILE(r0, r7) ", %1, %5, " SCAML_TILE(r0, r7) "\n\t" \
               "v_mfma_f64_16x16x4_f64 " SCAML_TILE(r0, r7) ", %2, %6, " SCAML_TILE(r0, r7) "\n\t" \
               "v_mfma_f64_16x16x4_f64 " SCAML_TILE(r0, r7) ", %3, %7, " SCAML_TILE(r0, r7)        \
               :                                                                                   \
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3)            \
               : SCAML_CLOB8(r0, r1, r2, r3, r4, r5, r6, r7))

#define MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 2" ::: "memory")

// switch-dispatch of one runtime slot index onto the code for the matching physical tile
#define SCAML_CASE_(S, r0, r1, r2, r3, r4, r5, r6, r7) \
  case S:                                              \
    if (S < SLOTS) { SCAML_BODY(r0, r1, r2, r3, r4, r5, r6, r7) } \
    break;
#define SCAML_DISPATCH(s) switch (s) { SCAML_TILE_LIST(SCAML_CASE_) default: break; }

// One factorisation attempt with a given diagonal jitter, start to finish (kernel matrix, panel
// loop, scalars, alpha).  Returns 0 on success or the 1-based index of the failing pivot.
// The kernel inlines this once for the first attempt (straight-line: nothing loop-invariant can be
// hoisted out of the kernel-matrix build and stay live across the panel loop) and calls the
// out-of-line copy below for the rare jitter retries.
template <int NB, int WU, int KIND>
__device__ __forceinline__ int gp_fit_attempt(const FitParams& p, const double jitter) {
  constexpr int NP = NB * 16;                 // padded matrix order
  constexpr int NT = NB * (NB + 1) / 2;       // lower-triangular tiles
  constexpr int SLOTS = (NT + WU - 1) / WU;   // tiles per update wave
  constexpr int PANEL = NP * PP;              // doubles per LDS panel buffer
  constexpr int NTHREADS = (WU + 1) * 64;
  static_assert(SLOTS <= 20, "tile_regs.inc provides 20 accumulator tiles");

  extern __shared__ double lds[];
  // region A (overlaid): xsT [D][NP] during the kernel-matrix build; PT[2][NP][PP] + WAll[NB][16][PP] afterwards
  double* xsT = lds;
  double* PT = lds;
  double* WAll = lds + 2 * PANEL;
  const int regionA = (p.D * NP > 2 * PANEL + NB * 16 * PP) ? p.D * NP : 2 * PANEL + NB * 16 * PP;
  double* ytil = lds + regionA;   // [NP] running right-hand side
  double* vv = ytil + NP;         // [NP] v = L^-1 y
  double* ww = vv + NP;           // [NP] back-substitution workspace -> alpha
  double* dl = ww + NP;           // [NP] diag(L)
  double* trash = dl + NP;        // [64] per-lane dump slot of the panel wave
  double* exptab = trash + 64;    // [64] 2^(j/64) for exp_neg
  int* rowlist = (int*)(exptab + 64);  // [WU][NB][8]: count, then up to 7 packed (slot << 8 | column) per block row
  double* invl = exptab + 64 + WU * NB * 4;  // [D]  1 / lengthscale
  int* flagp = (int*)(invl + p.D + (p.D & 1));  // [2] fail index

  const int task = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_panel = wave == WU;
  const int lc = lane & 15;   // tile column owned by this lane
  const int lq = lane >> 4;   // tile row group: rows lq + 4 * reg
  const int N = p.N, D = p.D;
  int n = p.n_points ? p.n_points[task] : N;
  n = n < 0 ? 0 : (n > N ? N : n);
  const bool from_matrix = p.A_in != nullptr;   // POTRF mode: the matrix is given, nothing to evaluate
  const double* Ag = from_matrix ? p.A_in + (size_t)task * N * N : nullptr;
  const double* Xg = from_matrix ? nullptr : p.X + (size_t)task * N * D;
  const double* yg = p.y ? p.y + (size_t)task * N : nullptr;
  const double* th = from_matrix ? nullptr : p.theta + (size_t)task * (D + 2);
  const double os = from_matrix ? 1.0 : th[D];
  const double noise = from_matrix ? 0.0 : th[D + 1];
  const double jit_in = p.jitter_in ? p.jitter_in[task] : 0.0;
  double* Lg = (p.flags & SCAML_FIT_STORE_L) ? p.L + (size_t)task * N * N : nullptr;
  const bool zero_upper = (p.flags & SCAML_FIT_ZERO_UPPER) != 0;
  const int lane_idx = lq * N + lc;  // element offset of this lane inside a 16x16 tile of L (row-major, ld = N)

  // Tiles in column-major order over the lower triangle: column j starts at tile off(j); this wave
  // owns tiles t = s * WU + wave (slot s), so its tiles of column j are the contiguous slots
  // [slo(j), slo(j+1)) and everything right of column j is the suffix starting at slo(j+1).
  auto off = [](int j) { return j * NB - j * (j - 1) / 2; };
  auto slo = [&](int j) { const int o = off(j) - wave; return o <= 0 ? 0 : (o + WU - 1) / WU; };

  STAMP_DECL;
  int fail = 0;
  {
    const double diag_add = noise + jitter + jit_in;
    __syncthreads();  // previous attempt done with region A
    if (!from_matrix && tid < D) invl[tid] = 1.0 / th[tid];
    if (tid == 0) flagp[0] = 0;
    exp2_table_init(exptab, tid);
    if (!is_panel && lane == 0) {
      // which off-diagonal tiles of each block row this wave holds (for the back-substitution)
      int* rl = rowlist + wave * NB * 8;
      for (int i = 0; i < NB; ++i) rl[8 * i] = 0;
      int j = 0, r = wave;
      for (int s = 0; s < SLOTS; ++s) {
        while (j < NB && r >= NB - j) { r -= NB - j; ++j; }
        if (j >= NB) break;
        if (r > 0) {
          const int i = j + r, c = rl[8 * i];
          if (c < 7) { rl[8 * i + 1 + c] = (s << 8) | j; rl[8 * i] = c + 1; }
        }
        r += WU;
      }
    }
    __syncthreads();
    // ---- stage X / l transposed into LDS: xsT[d][row]; y into ytil
    for (int r = tid; r < NP; r += NTHREADS) {
      const bool in = r < n;
      if (!from_matrix)
        for (int d = 0; d < D; ++d) xsT[d * NP + r] = in ? Xg[(size_t)r * D + d] * invl[d] : 0.0;
      ytil[r] = (in && yg) ? yg[r] : 0.0;
    }
    __syncthreads();
    STAMP(0);

    // ---- kernel matrix straight into the accumulator tiles (update waves)
    if (!is_panel) {
      int kj = 0, kr = wave;  // column / row-in-column of the current slot's tile
#define SCAML_KBUILD_(S, r0, r1, r2, r3, r4, r5, r6, r7)                                           \
      if (S < SLOTS) {                                                                             \
        while (kj < NB && kr >= NB - kj) { kr -= NB - kj; ++kj; }                                  \
        if (kj < NB) {                                                                             \
          const int col = 16 * kj + lc;                                                            \
          const int row0 = 16 * (kj + kr) + lq;                                                    \
          double kt[4];                                                                            \
          if (from_matrix) {                                                                       \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                        \
              const int row = row0 + 4 * g;                                                        \
              double kv = 0.0;                                                                     \
              if (row < n && col < n) kv = row >= col ? Ag[(size_t)row * N + col] : Ag[(size_t)col * N + row]; \
              if (row == col) kv += diag_add;                                                      \
              if (row >= n || col >= n) kv = row == col ? 1.0 : 0.0;                               \
              kt[g] = kv;                                                                          \
            }                                                                                      \
          } else {                                                                                 \
            double d2[4] = {0.0, 0.0, 0.0, 0.0};                                                   \
            _Pragma("unroll 2") for (int d = 0; d < D; ++d) {                                      \
              const double* xr = xsT + d * NP;                                                     \
              const double xc = xr[col];                                                           \
              _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                      \
                double df = xr[row0 + 4 * g] - xc;                                                 \
                d2[g] = __builtin_fma(df, df, d2[g]);                                              \
              }                                                                                    \
            }                                                                                      \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                        \
              const int row = row0 + 4 * g;                                                        \
              double kv = os * kernel_from_sqdist<KIND>(d2[g], exptab);                            \
              if (row == col) kv += diag_add;                                                      \
              if (row >= n || col >= n) kv = row == col ? 1.0 : 0.0;                               \
              kt[g] = kv;                                                                          \
            }                                                                                      \
          }                                                                                        \
          TILE_SET(r0, r1, r2, r3, r4, r5, r6, r7, kt[0], kt[1], kt[2], kt[3]);                    \
          kr += WU;                                                                                \
        }                                                                                          \
      }
      SCAML_TILE_LIST(SCAML_KBUILD_)
#undef SCAML_KBUILD_
    }
    __syncthreads();  // xsT dead from here: region A becomes PT / WAll
    STAMP(1);

    // ---- prologue: column 0 to LDS, diagonal block 0 factored
    fail = 0;
    if (!is_panel) {
      const int s1 = slo(1);
      for (int s = 0; s < s1; ++s) {
        const int ti = s * WU + wave;
#define SCAML_BODY(r0, r1, r2, r3, r4, r5, r6, r7)                                                 \
        double e0, e1, e2, e3;                                                                     \
        TILE_GET(r0, r1, r2, r3, r4, r5, r6, r7, e0, e1, e2, e3);                                  \
        double* dst = PT + (16 * ti + lq) * PP + lc;                                               \
        dst[0] = e0; dst[4 * PP] = e1; dst[8 * PP] = e2; dst[12 * PP] = e3;
        SCAML_DISPATCH(s)
#undef SCAML_BODY
      }
    }
    __syncthreads();
    if (is_panel) {
      int bad = potf2_inv_block(PT, WAll, vv, trash, ytil, 0, lane);
      if (bad && lane == 0) flagp[0] = bad;
    }
    __syncthreads();
    STAMP(2);
    fail = flagp[0];

    if (!fail) {
      // Finished tiles of column k: registers -> HBM as 128-byte row segments; the mirrored upper tile is
      // written as zeros by the same lanes (no separate zero-fill pass over L).
      auto store_column = [&](int k) {
        if (Lg) {
          const int sa = slo(k), sb = slo(k + 1), offk = off(k);
          for (int s = sa; s < sb; ++s) {
            const int ti = k + (s * WU + wave - offk);
            double e[4];
#define SCAML_BODY(r0, r1, r2, r3, r4, r5, r6, r7) TILE_GET(r0, r1, r2, r3, r4, r5, r6, r7, e[0], e[1], e[2], e[3]);
            SCAML_DISPATCH(s)
#undef SCAML_BODY
            {
              double* tb = Lg + ((size_t)(16 * ti) * N + 16 * k);   // tile (ti, k), wave-uniform
              double* mb = Lg + ((size_t)(16 * k) * N + 16 * ti);   // mirrored tile (k, ti)
              if (16 * ti + 16 <= n && ti != k) {                   // interior tile: no per-lane bounds
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                  tb[(size_t)g * 4 * N + lane_idx] = e[g];
                  if (zero_upper) mb[(size_t)g * 4 * N + lane_idx] = 0.0;
                }
              } else {
                const int col = 16 * k + lc, mc = 16 * ti + lc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                  const int row = 16 * ti + lq + 4 * g, mr = 16 * k + lq + 4 * g;
                  if (row < n && col < n && (zero_upper || col <= row)) tb[(size_t)g * 4 * N + lane_idx] = e[g];
                  if (zero_upper && ti != k && mr < n && mc < n) mb[(size_t)g * 4 * N + lane_idx] = 0.0;
                }
              }
            }
          }
        }
      };
      for (int k = 0; k < NB; ++k) {
        double* buf = PT + (k & 1) * PANEL;         // panel k: diagonal block final, rows below raw
        double* nbuf = PT + ((k + 1) & 1) * PANEL;  // receives column k+1
        const double* Wk = WAll + k * 16 * PP;
        if (!is_panel) {
          const int sa = slo(k), sb = slo(k + 1), offk = off(k);
          // T(k): column k becomes final.  Sub-diagonal tiles: L_ik = A_ik W^T (4 MFMAs, A rows from the
          // panel, W = L_kk^-1); the diagonal tile is read back from the panel.
          for (int s = sa; s < sb; ++s) {
            const int ti = k + (s * WU + wave - offk);
#define SCAML_BODY(r0, r1, r2, r3, r4, r5, r6, r7)                                                 \
            if (ti == k) {                                                                         \
              const double* prow = buf + (16 * ti + lq) * PP + lc;                                 \
              TILE_SET(r0, r1, r2, r3, r4, r5, r6, r7, prow[0], prow[4 * PP], prow[8 * PP], prow[12 * PP]); \
            } else {                                                                               \
              const double* pa = buf + (16 * ti + lc) * PP + lq;                                   \
              const double* pw = Wk + lc * PP + lq;                                                \
              TILE_MFMA4_SET(r0, r1, r2, r3, r4, r5, r6, r7, pa[0], pa[4], pa[8], pa[12], pw[0], pw[4], pw[8], pw[12]); \
            }
            SCAML_DISPATCH(s)
#undef SCAML_BODY
          }
          MFMA_DRAIN();
          // final tiles: back to the panel for everyone's operand reads
          for (int s = sa; s < sb; ++s) {
            const int ti = k + (s * WU + wave - offk);
            double e[4];
#define SCAML_BODY(r0, r1, r2, r3, r4, r5, r6, r7) TILE_GET(r0, r1, r2, r3, r4, r5, r6, r7, e[0], e[1], e[2], e[3]);
            SCAML_DISPATCH(s)
#undef SCAML_BODY
            if (ti != k) {
              double* prow = buf + (16 * ti + lq) * PP + lc;
              prow[0] = e[0]; prow[4 * PP] = e[1]; prow[8 * PP] = e[2]; prow[12 * PP] = e[3];
            } else {
#pragma unroll
              for (int g = 0; g < 4; ++g)
                if (lc == lq + 4 * g) dl[16 * k + lc] = e[g];
            }
          }
        }
        // (issuing these stores after barrier X instead — off the critical path — measured 4 % slower:
        //  they then compete with the bulk update for issue slots; interleaved A/B, tools/dev_ab.py)
        if (!is_panel) store_column(k);
        STAMP(3);
        __syncthreads();  // Z: panel k final in LDS
        STAMP(4);
        if (k + 1 == NB) break;
        if (is_panel) {
          // running right-hand side: y_r -= L[r, panel k] . v_k for every row below the block
          double vk[16];
          const double vmine = vv[16 * k + lc];
#pragma unroll
          for (int c = 0; c < 16; ++c) vk[c] = readlane_f64(vmine, c);
          for (int r = 16 * (k + 1) + lane; r < NP; r += 64) {
            const double* pr = buf + r * PP;
            double yr = ytil[r];
#pragma unroll
            for (int c = 0; c < 16; ++c) yr = __builtin_fma(-pr[c], vk[c], yr);
            ytil[r] = yr;
          }
        } else {
          // U1: column k+1 first: rank-16 update, then spill (raw) to the other panel buffer
          const int sa = slo(k + 1), sb = slo(k + 2), offk1 = off(k + 1);
          const double* pb = buf + (16 * (k + 1) + lc) * PP + lq;
          for (int s = sa; s < sb; ++s) {
            const int ti = k + 1 + (s * WU + wave - offk1);
            const double* pa = buf + (16 * ti + lc) * PP + lq;
#define SCAML_BODY(r0, r1, r2, r3, r4, r5, r6, r7) \
            TILE_MFMA4_SUB(r0, r1, r2, r3, r4, r5, r6, r7, pa[0], pa[4], pa[8], pa[12], pb[0], pb[4], pb[8], pb[12]);
            SCAML_DISPATCH(s)
#undef SCAML_BODY
          }
          MFMA_DRAIN();
          for (int s = sa; s < sb; ++s) {
            const int ti = k + 1 + (s * WU + wave - offk1);
            double e0, e1, e2, e3;
#define SCAML_BODY(r0, r1, r2, r3, r4, r5, r6, r7) TILE_GET(r0, r1, r2, r3, r4, r5, r6, r7, e0, e1, e2, e3);
            SCAML_DISPATCH(s)
#undef SCAML_BODY
            double* dst = nbuf + (16 * ti + lq) * PP + lc;
            dst[0] = e0; dst[4 * PP] = e1; dst[8 * PP] = e2; dst[12 * PP] = e3;
          }
        }
        STAMP(5);
        __syncthreads();  // X: column k+1 (raw) and the updated right-hand side visible to the panel wave
        STAMP(6);
        if (is_panel) {
          // the panel wave is the youngest wave on its SIMD and would lose every issue slot to the
          // update wave streaming MFMAs next to it: raise its priority for the pivot chain
          __builtin_amdgcn_s_setprio(3);
          int bad = potf2_inv_block(nbuf, WAll + (k + 1) * 16 * PP, vv, trash, ytil, k + 1, lane);
          __builtin_amdgcn_s_setprio(0);
          if (bad && lane == 0) flagp[0] = bad;
        } else {
          // U2: the bulk of the trailing update, overlapped with the panel wave: every slot from
          // slo(k+2) on, entered through one switch and then falling through slot after slot
          const int s0 = slo(k + 2);
          int uj = k + 2, ur = s0 * WU + wave - off(k + 2);
#define SCAML_U2_(S, r0, r1, r2, r3, r4, r5, r6, r7)                                               \
          case S:                                                                                  \
            if (S < SLOTS) {                                                                       \
              while (uj < NB && ur >= NB - uj) { ur -= NB - uj; ++uj; }                            \
              if (uj < NB) {                                                                       \
                const double* pa = buf + (16 * (uj + ur) + lc) * PP + lq;                          \
                const double* pb = buf + (16 * uj + lc) * PP + lq;                                 \
                TILE_MFMA4_SUB(r0, r1, r2, r3, r4, r5, r6, r7, pa[0], pa[4], pa[8], pa[12], pb[0], pb[4], pb[8], pb[12]); \
                ur += WU;                                                                          \
              }                                                                                    \
            }
          switch (s0) { SCAML_TILE_LIST(SCAML_U2_) default: break; }
#undef SCAML_U2_
          MFMA_DRAIN();
        }
        STAMP(7);
        __syncthreads();  // Y: diagonal block k+1 factored
        STAMP(8);
        fail = flagp[0];
        if (fail) break;
      }
    }
  }

  // ---- scalars: quad, logdet (panel wave), then alpha by blocked back-substitution
  __syncthreads();
  STAMP(9);
  if (!fail) {
    if (is_panel) {
      double q = 0.0, ld = 0.0;
      for (int r = lane; r < NP; r += 64) {
        const double v = vv[r];
        q = __builtin_fma(v, v, q);
        ld += log(dl[r]);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        q += __shfl_xor(q, o);
        ld += __shfl_xor(ld, o);
      }
      if (lane == 0) {
        ld *= 2.0;
        if (p.quad) p.quad[task] = q;
        if (p.logdet) p.logdet[task] = ld;
        if (p.mll) p.mll[task] = n > 0 ? -0.5 * (q + ld + n * 1.8378770664093454836) / n : 0.0;
      }
    }
    if (p.Linv_diag) {
      // W_k = L_kk^-1 for every diagonal block, (T, ceil(N/16), 16, 16): the batched posterior solves
      // L^-1 K_*^T with them on the matrix cores instead of by substitution
      const int nbn = (N + 15) / 16;
      double* Wg = p.Linv_diag + (size_t)task * nbn * 256;
      for (int e = tid; e < nbn * 256; e += NTHREADS) Wg[e] = WAll[(e >> 4) * PP + (e & 15)];
    }
    if (p.alpha) {
      // alpha = L^-T v by blocks from the bottom.  Per block k every wave forms alpha_k = W_k^T w_k
      // itself (a 16x16 mat-vec out of LDS: four terms per lane group, then two cross-group adds),
      // the update waves then fold alpha_k into w_j for the tiles (k, j) they hold in registers; one
      // barrier per block.
      for (int r = tid; r < NP; r += NTHREADS) ww[r] = vv[r];
      __syncthreads();
      STAMP(11);
      for (int k = NB - 1; k >= 0; --k) {
        const double* Wk = WAll + k * 16 * PP;
        double ak = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) ak = __builtin_fma(Wk[(4 * lq + c) * PP + lc], ww[16 * k + 4 * lq + c], ak);
        ak += __shfl_xor(ak, 16);
        ak += __shfl_xor(ak, 32);   // every lane (lc, *) now holds alpha_k[lc]
        STAMP(12);
        if (is_panel) {
          if (lq == 0) dl[16 * k + lc] = ak;   // dl is free by now: alpha is collected there
        } else {
          // alpha_k[lq + 4 g] for the four rows of this lane's tile elements
          const double a0 = __shfl(ak, lq), a1 = __shfl(ak, lq + 4), a2 = __shfl(ak, lq + 8), a3 = __shfl(ak, lq + 12);
          // tiles (k, j), j < k, held by this wave
          const int* rl = rowlist + (wave * NB + k) * 8;
          const int cnt = rl[0];
          for (int i = 0; i < cnt; ++i) {
            const int sj = rl[1 + i], s = sj >> 8, j = sj & 0xff;
            double e0, e1, e2, e3;
#define SCAML_BODY(r0, r1, r2, r3, r4, r5, r6, r7) TILE_GET(r0, r1, r2, r3, r4, r5, r6, r7, e0, e1, e2, e3);
            SCAML_DISPATCH(s)
#undef SCAML_BODY
            double part = e0 * a0;
            part = __builtin_fma(e1, a1, part);
            part = __builtin_fma(e2, a2, part);
            part = __builtin_fma(e3, a3, part);
            // four lane groups add into the same w_j entry: LDS fp64 atomic, no return value
            __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)(ww + 16 * j + lc), -part);
          }
        }
        STAMP(13);
        __syncthreads();
        STAMP(14);
      }
      for (int r = tid; r < NP; r += NTHREADS) ww[r] = dl[r];
      __syncthreads();
      for (int r = tid; r < n; r += NTHREADS) p.alpha[(size_t)task * N + r] = ww[r];
    }
  }
  STAMP(10);
  STAMP_FLUSH(task);
  return fail;
}

template <int NB, int WU, int KIND>
__device__ __noinline__ int gp_fit_retry(const FitParams& p, const double jitter) {
  return gp_fit_attempt<NB, WU, KIND>(p, jitter);
}

template <int NB, int WU, int KIND>
__global__ __launch_bounds__((WU + 1) * 64) void gp_fit_fused_kernel(FitParams p) {
  // jitter escalation of linear_operator's psd_safe_cholesky, per task, without leaving the GPU
  double jitter = 0.0;
  int fail = gp_fit_attempt<NB, WU, KIND>(p, 0.0);
#ifndef SCAML_STAMPS  // (the diagnostic build times the first attempt only)
  if (fail && !(p.flags & SCAML_FIT_NO_RETRY)) {
    const FitParams pc = p;  // only this cold copy has its address taken; `p` stays in the kernarg segment
    for (int attempt = 1; attempt < 4 && fail; ++attempt) {
      jitter = attempt == 1 ? 1e-8 : (attempt == 2 ? 1e-7 : 1e-6);
      fail = gp_fit_retry<NB, WU, KIND>(pc, jitter);
    }
  }
#endif
  if (threadIdx.x == 0) {
    const int task = blockIdx.x;
    if (fail) {
      const double nan = __builtin_nan("");
      if (p.quad) p.quad[task] = nan;
      if (p.logdet) p.logdet[task] = nan;
      if (p.mll) p.mll[task] = nan;
    }
    p.info[task] = fail;
    if (p.jitter_used) p.jitter_used[task] = jitter;
  }
}

}  // namespace scaml

// Explicit instantiations: one kernel per padded size class (NB 16-blocks, WU update waves) and
// kernel kind (see VGPR_CAPS in __graft_entry__.py for the register budgets).
#define SCAML_INSTANTIATE(NB, WU)                                                   \
  template __global__ void scaml::gp_fit_fused_kernel<NB, WU, 0>(scaml::FitParams); \
  template __global__ void scaml::gp_fit_fused_kernel<NB, WU, 1>(scaml::FitParams);
SCAML_INSTANTIATE(2, 1)
SCAML_INSTANTIATE(4, 1)
SCAML_INSTANTIATE(8, 3)
SCAML_INSTANTIATE(16, 7)

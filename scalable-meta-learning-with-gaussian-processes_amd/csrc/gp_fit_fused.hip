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
// 16x16 tiles; tile t (column-major over the triangle) belongs to update wave t % WU, slot t / WU.
// A tile is held TRANSPOSED in the C/D layout of v_mfma_f64_16x16x4_f64: lane (lc = lane & 15,
// lq = lane >> 4) register g holds A[16 i + lc][16 j + lq + 4 g] -- the lane owns a piece of a ROW.
// That makes the accumulator registers themselves the B operand of W * A^T, so the triangular solve
// of a column (L_ik = A_ik W_k^T, W_k = L_kk^-1) runs on the tile where it sits, without a trip
// through LDS, and the forward-substitution dot products are in-lane.
// Waves are specialised: WU "update" waves own tiles; one "panel" wave owns none and factors the
// 16x16 diagonal blocks.  The panel loop is a dataflow pipeline WITHOUT workgroup barriers: the waves
// hand work to each other through counters in LDS (release/acquire at workgroup scope).
//   panel wave, step j:   D_j (diagonal tile with the updates of panels <= j-2) and the raw tile
//       R_j = A[j][j-1], both parked in LDS by their owners during U1(j-2) -> L_j,j-1 = R_j W_{j-1}^T and
//       D_j -= L L^T by 8 MFMAs of its own -> factor D_j (16 rank-1 MFMA updates on the symmetric
//       block, the next pivot formed ahead on the VALU so the 64-cycle MFMA latency stays off the
//       pivot chain; a second accumulator receives the same row operations and ends as
//       W_j = L_jj^-1) -> publish flagW[j].  It runs ahead of the update waves.
//   update waves, iteration k (column k final in LDS = cntT[k] complete):
//       U1  column k+1 and the diagonal tile D_{k+2} get the rank-16 update of panel k first; D_{k+2}
//           and R_{k+2} are parked for the panel wave (cntS[k])
//       F   (after flagW[k+1]) column k+1 is finalised in registers (4 MFMAs per tile), written to
//           the LDS panel for everyone's operand reads (cntT[k+1]) and folded into the running
//           right-hand side; v_{k+1} = W y rides along
//       U2  all remaining tiles get the rank-16 update of panel k (4 MFMAs per tile, operands from
//           the LDS panel) -- the bulk, during which the other waves' F(k+1) results arrive
//       then column k+1 goes to HBM (128-byte row segments; the mirrored upper tile as zeros).
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
#ifndef SCAML_STAMP_WAVE
#define SCAML_STAMP_WAVE 0   // which update wave reports (next to the panel wave)
#endif
#define STAMP_FLUSH(task) do { if (g_stamp_buf && (threadIdx.x == 64 * SCAML_STAMP_WAVE || threadIdx.x == blockDim.x - 64)) for (int i_ = 0; i_ < 16; ++i_) g_stamp_buf[((task) * 2 + (threadIdx.x != 64 * SCAML_STAMP_WAVE)) * 16 + i_] = st_acc[i_]; } while (0)
#ifdef SCAML_STAMPS_PER_PANEL
// variant: slot k = time (since kernel start) at which panel step k was published / U1(k) was done
#undef STAMP
#define STAMP(i)
#ifdef SCAML_STAMPS_ONE_PANEL
// variant: absolute times of the sub-steps of one panel iteration (SCAML_STAMPS_ONE_PANEL = k)
#define STAMP_AT(i)
#define STAMP_K(kk, i) do { if ((kk) == SCAML_STAMPS_ONE_PANEL) st_acc[i] = __builtin_amdgcn_s_memtime() - st_prev; } while (0)
#else
#define STAMP_AT(i) do { st_acc[i] = __builtin_amdgcn_s_memtime() - st_prev; } while (0)
#endif
#else
#define STAMP_AT(i)
#endif
#else
#define STAMP_AT(i)
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH(task)
#endif
#ifndef STAMP_K
#define STAMP_K(kk, i)
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

// ---- wave-to-wave hand-off through LDS counters -------------------------------------------------
// All waves of the workgroup are resident, so a spinning wave cannot starve the one it waits for;
// s_sleep keeps the pollers off the issue ports.  Release/acquire at workgroup scope orders the LDS
// traffic (on gfx950 that is s_waitcnt lgkmcnt(0) around the atomic; HBM stores are not waited for).
#ifndef SCAML_SLEEP
#define SCAML_SLEEP 1   // s_sleep argument of the pollers (64-cycle units); A/B-tuned
#endif
typedef __attribute__((address_space(3))) int lds_int_t;
__device__ __forceinline__ int sync_peek(const int* p) {
  return __hip_atomic_load((const lds_int_t*)p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void sync_wait_ge(const int* p, int target) {
  while (sync_peek(p) < target) __builtin_amdgcn_s_sleep(SCAML_SLEEP);
}
// waiting side of a hand-off that may never come because the panel wave hit a bad pivot
__device__ __forceinline__ bool sync_wait_ge_or_fail(const int* p, int target, const int* failp) {
  while (sync_peek(p) < target) {
    if (sync_peek(failp) != 0) return false;
    __builtin_amdgcn_s_sleep(SCAML_SLEEP);
  }
  return true;
}
// Signalling side.  The LDS unit executes the DS instructions of one wave in issue order, so a counter
// update issued after the data writes is performed after them: no s_waitcnt is needed in front of it
// (a release fence would put one there and stall the wave for the full LDS write latency at every
// hand-off).  The asm statements only pin the compiler's ordering.
__device__ __forceinline__ void sync_arrive(int* p, int lane) {
  // (written as asm: for a one-lane atomic the compiler's atomic optimiser adds a dozen instructions of
  //  lane counting.  An LDS op the compiler does not count only makes its later s_waitcnt lgkmcnt(n)
  //  conservative, since LDS results return in order.)
  if (lane == 0) {
    const unsigned addr = (unsigned)(size_t)(lds_int_t*)p;
    const int one = 1;
    asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(one) : "memory");
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void sync_publish(int* p, int value, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_store((lds_int_t*)p, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}

// ---- panel wave: Cholesky of one 16x16 diagonal block and its inverse ----
// `a` holds the symmetric trailing block (both triangles) in the MFMA C/D layout.  Step c scales
// row c (= column c by symmetry, already in operand position on lane group c & 3) and applies the
// rank-1 update with ONE MFMA.  A second accumulator R (initially I) receives the same row
// operations and ends as W = L_kk^-1.  Lanes outside the active group store to a per-lane trash
// slot instead of being masked off (no exec juggling in the 16-step chain).
// Outputs: LT[r * PP + c] = L_kk[r][c] (zero above the diagonal), Wt[j * PP + c] = W[c][j] (W TRANSPOSED:
// both stores of step c then sit at the immediate offset c from a per-lane base that is fixed for the
// whole block, no address arithmetic inside the chain).
// Returns 0 or the 1-based global index of the first non-positive / out-of-range pivot.
__device__ __forceinline__ int potf2_inv_block(d4_t a, double* LT, double* Wt, double* trash, int k, int lane) {
  const int lc = lane & 15, lq = lane >> 4;
  d4_t R;
#pragma unroll
  for (int g = 0; g < 4; ++g) R[g] = (lc == lq + 4 * g) ? 1.0 : 0.0;
  // per-lane store bases: step c is written by lane group c & 3 only, the other groups dump into their
  // trash slot (trash[lane + 0..15]: 80 doubles)
  typedef __attribute__((address_space(3))) double lds_double_t;   // 32-bit LDS pointers: 8 VGPRs in all
  lds_double_t* baseL[4];
  lds_double_t* baseW[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    baseL[g] = (lds_double_t*)((lq == g) ? LT + lc * PP : trash + lane);
    baseW[g] = (lds_double_t*)((lq == g) ? Wt + lc * PP : trash + lane);
  }
  // A wave issues one VALU instruction per ~8.5 cycles whatever it is (tools/valu_rate_probe.hip) and executes
  // in order, so a step costs its instruction count plus its stalls (stand-alone, tools/potf2_probe.hip: ~290
  // cycles per pivot, of which ~150 are the floor MFMA -> read pivot -> scale column -> MFMA, ~60 the
  // inverse's MFMA -- the matrix pipe takes one fp64 MFMA per 64 cycles -- and ~50 the rsqrt):
  // * the pivot is read out of the accumulator once the previous rank-1 update has landed, rather than formed
  //   ahead of the MFMA from the un-updated values (5 instructions more per step);
  // * the MFMAs are asm, so that hipcc adds no wait states of its own behind them (it pads every read of the
  //   result to 11-15 wait states; the hardware interlock on the first three result pairs stalls exactly as
  //   long as needed);
  // * no pivot log, no running minimum: a non-positive, NaN or f32-underflowing pivot makes the f32-seeded
  //   rsqrt return NaN, which reaches every later column (0 * NaN), so the last diagonal entry tells whether
  //   anything failed, and the first NaN on the stored diagonal of L tells where.
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int g = c & 3, rg = c >> 2;
    const bool mine = lq == g;
    const bool low = (unsigned)(lane - (16 * g + c)) < (unsigned)(16 - c);  // mine && lc >= c
    if (rg == 3) {
      // gfx950: reads of the LAST result pair of an fp64 MFMA are not interlocked -- touch the first pair of
      // both accumulators before the steps that work on the last one (tools/mfma_hazard_probe.hip)
      double a0 = a[0], r0 = R[0];
      asm volatile("v_max_f64 %0, %0, %0\n\tv_max_f64 %1, %1, %1" : "+v"(a0), "+v"(r0));
      a[0] = a0; R[0] = r0;
      asm volatile("" : "+v"(a), "+v"(R));
    }
    const double dpiv = readlane_f64(a[rg], 16 * g + c);   // a[c][c], updates of the steps < c applied
    const double am = low ? a[rg] : 0.0;    // column c (row c of the symmetric block), masked
    const double Rm = mine ? R[rg] : 0.0;
    const double ri = rsqrt_seeded(dpiv);
    const double lcol = am * ri;   // L[lc][c]
    const double wrow = Rm * ri;   // W[c][lc]
    if (c < 15) {
      // (the update of `a` is the one the next pivot waits for: it enters the pipe first)
      asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %1, %0 neg:[1,0,0]" : "+v"(a) : "v"(lcol));
      asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0 neg:[1,0,0]" : "+v"(R) : "v"(lcol), "v"(wrow));
    }
    baseL[g][c] = lcol;
    baseW[g][c] = wrow;
  }
  int bad = 0;
  {
    const double last = LT[15 * PP + 15];
    if (!(last == last)) {
#pragma unroll
      for (int c = 15; c >= 0; --c) {
        const double d = LT[c * PP + c];
        if (!(d == d)) bad = 16 * k + c + 1;
      }
    }
  }
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

// out (VGPR tile) = W * tile^T-operand: the accumulator registers of the (transposed) tile are the B
// operand, k-step m reads register pair m; w0..w3 = W[lc][lq + 4m].  The result is the finalised
// tile in the same transposed layout.  Ends drained (the VALU may read `out`).
#define TILE_TRSM_TO_V(r0, r1, r2, r3, r4, r5, r6, r7, w0, w1, w2, w3, out)                        \
  asm volatile("s_nop 1\n\t"                                                                     \
               "v_mfma_f64_16x16x4_f64 %0, %1, a[" #r0 ":" #r1 "], 0\n\t"                         \
               "v_mfma_f64_16x16x4_f64 %0, %2, a[" #r2 ":" #r3 "], %0\n\t"                        \
               "v_mfma_f64_16x16x4_f64 %0, %3, a[" #r4 ":" #r5 "], %0\n\t"                        \
               "v_mfma_f64_16x16x4_f64 %0, %4, a[" #r6 ":" #r7 "], %0\n\t"                        \
               "s_nop 15\n\ts_nop 2"                                                              \
               : "=&v"(out)                                                                        \
               : "v"(w0), "v"(w1), "v"(w2), "v"(w3))

// the same without the trailing wait: several tiles are issued back to back, one MFMA_DRAIN after the last
#define TILE_TRSM_ISSUE(r0, r1, r2, r3, r4, r5, r6, r7, w0, w1, w2, w3, out)                       \
  asm volatile("s_nop 1\n\t"                                                                     \
               "v_mfma_f64_16x16x4_f64 %0, %1, a[" #r0 ":" #r1 "], 0\n\t"                         \
               "v_mfma_f64_16x16x4_f64 %0, %2, a[" #r2 ":" #r3 "], %0\n\t"                        \
               "v_mfma_f64_16x16x4_f64 %0, %3, a[" #r4 ":" #r5 "], %0\n\t"                        \
               "v_mfma_f64_16x16x4_f64 %0, %4, a[" #r6 ":" #r7 "], %0"                             \
               : "=&v"(out)                                                                        \
               : "v"(w0), "v"(w1), "v"(w2), "v"(w3))

// out (VGPR tile) = tile^T-as-A-operand * W^T: the accumulator registers of the (transposed) tile are the A operand this
// time (k-step m reads register pair m: A[i = lc][k = lq + 4m] = A_ik[lc][lq + 4m]), w0..w3 = W[lc][lq + 4m] the B operand
// (= W^T[lq + 4m][lc]).  The result is the finalised tile UN-transposed (register g of lane (lc, lq) = L[lq + 4g][lc]):
// the layout the finished tiles are kept in, straight off the matrix core -- no read-back through LDS.
#define TILE_TRSM_ISSUE_UT(r0, r1, r2, r3, r4, r5, r6, r7, w0, w1, w2, w3, out)                    \
  asm volatile("s_nop 1\n\t"                                                                     \
               "v_mfma_f64_16x16x4_f64 %0, a[" #r0 ":" #r1 "], %1, 0\n\t"                         \
               "v_mfma_f64_16x16x4_f64 %0, a[" #r2 ":" #r3 "], %2, %0\n\t"                        \
               "v_mfma_f64_16x16x4_f64 %0, a[" #r4 ":" #r5 "], %3, %0\n\t"                        \
               "v_mfma_f64_16x16x4_f64 %0, a[" #r6 ":" #r7 "], %4, %0"                             \
               : "=&v"(out)                                                                        \
               : "v"(w0), "v"(w1), "v"(w2), "v"(w3))

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
// addressing of the task's arrays: dense (everything follows from N) or given (blocked fit)
template <bool BLK>
struct FitBlk {
  __device__ __forceinline__ int ldl(int N) const { return N; }
  __device__ __forceinline__ size_t sx(size_t dense) const { return dense; }
  __device__ __forceinline__ size_t sy(size_t dense) const { return dense; }
  __device__ __forceinline__ size_t sL(size_t dense) const { return dense; }
  __device__ __forceinline__ size_t sW(size_t dense) const { return dense; }
};
template <>
struct FitBlk<true> {
  FitBlockParams v;
  __device__ __forceinline__ int ldl(int) const { return v.ldl; }
  __device__ __forceinline__ size_t sx(size_t) const { return (size_t)v.stride_x; }
  __device__ __forceinline__ size_t sy(size_t) const { return (size_t)v.stride_y; }
  __device__ __forceinline__ size_t sL(size_t) const { return (size_t)v.stride_L; }
  __device__ __forceinline__ size_t sW(size_t) const { return (size_t)v.stride_W; }
};

template <int NB, int WU, int KIND, bool BLK>
__device__ __forceinline__ int gp_fit_attempt(const FitParams& p, const double jitter, const FitBlk<BLK> b) {
  constexpr int NP = NB * 16;                 // padded matrix order
  constexpr int NT = NB * (NB + 1) / 2;       // lower-triangular tiles
  constexpr int SLOTS = (NT + WU - 1) / WU;   // tiles per update wave
  constexpr int MAXC = (NB + WU - 1) / WU;    // ... of which at most this many in one column
  constexpr int PANEL = NP * PP;              // doubles per LDS panel buffer
  constexpr int NTHREADS = (WU + 1) * 64;
  static_assert(SLOTS <= 20, "tile_regs.inc provides 20 accumulator tiles");

  extern __shared__ double lds[];
  // region A (overlaid): xsT [D][NP] during the kernel-matrix build; PT[3][NP][PP] (round 2: three column buffers) + WAll[NB][16][PP] +
  // DG[2][256] + CR[2][256] (parked diagonal / sub-diagonal tiles for the panel wave, register image) +
  // LT[2][16][PP] (factored diagonal blocks) afterwards
  double* xsT = lds;
  double* PT = lds;
  double* WAll = lds + 3 * PANEL;
  double* DG = WAll + NB * 16 * PP;
  double* CR = DG + 2 * 256;
  double* LT = CR + 2 * 256;
  constexpr int REGION_A = 3 * PANEL + (NB + 2) * 16 * PP + 4 * 256;
  // (during the build: xsT, then up to 24 tile images written by the panel wave, see PANEL_BUILDS)
  const int XROWS = ((p.D + 3) & ~3) + 4;   // staged point stack: D rounded up to the MFMA k-step + 4 tail rows
  const int buildA = XROWS * NP + ((WU == 7 && NB == 16) ? 24 * 256 : 0);
  const int regionA = (buildA > REGION_A) ? buildA : REGION_A;
  double* ytil = lds + regionA;   // [NP] running right-hand side
  double* vv = ytil + NP;         // [NP] v = L^-1 y
  double* ww = PT + 2 * PANEL;    // [NP] back-substitution workspace -> alpha: tail only, on the (then free) third column buffer
  double* dl = vv + NP;           // [NP] diag(L)
  double* trash = dl + NP;        // [80] per-lane dump slots of the panel wave (lane + step) + [16] pivot log
  double* exptab = trash + 96;    // [64] 2^(j/64) for exp_neg
  int* rowlist = (int*)(exptab + 64);  // [WU][NB][8]: count, then up to 7 packed (slot << 8 | column) per block row
  double* invl = exptab + 64 + WU * NB * 4;  // [D]  1 / lengthscale
  int* flagp = (int*)(invl + p.D + (p.D & 1));  // [2] fail index
  int* flagW = flagp + 2;        // [NB] 1: W_k, v_k, L_kk published by the panel wave; 2: failed pivot
  int* cntT = flagW + NB;        // [NB] update waves done with T(k): column k final in LDS
  int* cntS = cntT + NB;         // [NB] update waves done with U1(k): column k+1 and D_{k+2} parked
  int* cntY = cntS + NB;         // [NB] update waves done folding column k into the right-hand side
  int* cntU = cntY + NB;         // [NB] update waves done with U2(k): nobody reads column k any more
  int* cntB = cntU + NB;         // [NB] back-substitution: tiles (i, k), i > k, already folded into w_k

  const int task = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_panel = wave == WU;
  const int lc = lane & 15;   // tile column owned by this lane
  const int lq = lane >> 4;   // tile row group: rows lq + 4 * reg
  const int N = p.N, D = p.D;
  const int LD = b.ldl(N);   // leading dimension of the stored factor (N, or the full size when this is a diagonal block of a blocked fit)
  int n = p.n_points ? p.n_points[task] : N;
  n = n < 0 ? 0 : (n > N ? N : n);
  const bool from_matrix = p.A_in != nullptr;   // POTRF mode: the matrix is given, nothing to evaluate
  const double* Ag = from_matrix ? p.A_in + (size_t)task * N * N : nullptr;
  const double* Xg = from_matrix ? nullptr : p.X + (size_t)task * b.sx((size_t)N * D);
  const double* yg = p.y ? p.y + (size_t)task * b.sy((size_t)N) : nullptr;
  const double* th = from_matrix ? nullptr : p.theta + (size_t)task * (D + 2);
  const double os = from_matrix ? 1.0 : th[D];
  const double noise = from_matrix ? 0.0 : th[D + 1];
  const double jit_in = p.jitter_in ? p.jitter_in[task] : 0.0;
  double* Lg = (p.flags & SCAML_FIT_STORE_L) ? p.L + (size_t)task * b.sL((size_t)N * N) : nullptr;
  const bool zero_upper = (p.flags & SCAML_FIT_ZERO_UPPER) != 0;
  const int lane_idx = lq * LD + lc;  // element offset of this lane inside a 16x16 tile of L (row-major, ld = LD)

  // Tiles in column-major order over the lower triangle: column j starts at tile off(j); this wave
  // owns tiles t = s * WU + wave (slot s), so its tiles of column j are the contiguous slots
  // [slo(j), slo(j+1)) and everything right of column j is the suffix starting at slo(j+1).
  auto off = [](int j) { return j * NB - j * (j - 1) / 2; };
  auto slo = [&](int j) { const int o = off(j) - wave; return o <= 0 ? 0 : (o + WU - 1) / WU; };

  STAMP_DECL;
  int fail = 0;
  {
    __syncthreads();  // previous attempt done with region A
    // (sqrt 2 folded into the scaling: the cross term of the squared distance then needs no factor 2)
    if (!from_matrix)   // strided: the workgroup may have fewer threads than D (128 threads at N <= 32, D up to ~588)
      for (int d = tid; d < D; d += NTHREADS) invl[d] = 1.4142135623730951 / th[d];
    if (tid < 2) flagp[tid] = 0;
    if (tid < 6 * NB) flagW[tid] = 0;
    exp2_table_init(exptab, tid);
    __syncthreads();
    // ---- stage x' = sqrt(2) X / l transposed into LDS: xsT[d][row], rows D .. D4-1 zero, then the tail rows
    // (-|x'|^2 / 2, 1, 0, 0): with them the squared distance is ONE chain of MFMAs,
    //   d2(a, b) = -(x'_a . x'_b + h_a * 1 + 1 * h_b),  h = -|x'|^2 / 2
    // (the expansion gpytorch's sq_dist uses, model.py:44-70 -> RBFKernel / MaternKernel), instead of 2 D VALU
    // operations per matrix element.  y into ytil.
    const int D4 = (D + 3) & ~3;
    for (int r = tid; r < NP; r += NTHREADS) {
      const bool in = r < n;
      if (!from_matrix) {
        double h = 0.0;
        for (int d = 0; d < D; ++d) {
          const double v = in ? Xg[(size_t)r * D + d] * invl[d] : 0.0;
          xsT[d * NP + r] = v;
          h = __builtin_fma(v, v, h);
        }
        for (int d = D; d < D4; ++d) xsT[d * NP + r] = 0.0;
        h *= -0.5;
        xsT[D4 * NP + r] = h;
        xsT[(D4 + 1) * NP + r] = 1.0;
        xsT[(D4 + 2) * NP + r] = 0.0;
        xsT[(D4 + 3) * NP + r] = 0.0;
        // NaN / inf points or lengthscales: the clamps of the kernel function would swallow a NaN, so the
        // diagonal is poisoned instead and the factorisation fails the way psd_safe_cholesky fails on NaN
        if (!(__builtin_fabs(h) <= 1e300)) flagp[1] = 1;
      }
      ytil[r] = (in && yg) ? yg[r] : 0.0;
    }
    __syncthreads();
    const double diag_add = noise + jitter + jit_in + (flagp[1] ? __builtin_nan("") : 0.0);
    // outputscale folded into the polynomial of the Matern kernel (RBF: c0 only)
    const double kc0 = os, kc1 = os * 2.2360679774997896964, kc2 = os * (5.0 / 3.0);
    STAMP(0);

    // ---- kernel matrix straight into the accumulator tiles
    // One 16x16 tile (block row kj + kr, block column kj), the four values of this lane (transposed tile:
    // the lane owns a row piece).
    // squared distances of one whole tile on the matrix core: A = points of block column kj (the MFMA's row
    // index is the tile's column, tiles are held transposed), B = points of block row kj + kr, A negated by
    // the instruction (blgp = 1).  The operand reads run one MFMA ahead.
    auto tile_dist = [&](int kj, int kr) -> d4_t {
      // (also run when the matrix is given -- on whatever the LDS holds, result unused: a conditional here
      //  becomes a select on the MFMA result right behind the MFMA, see tile_eval)
      d4_t d2 = {0.0, 0.0, 0.0, 0.0};
      const double* pa = xsT + lq * NP + 16 * kj + lc;
      const double* pb = xsT + lq * NP + 16 * (kj + kr) + lc;
      const double ta = xsT[(D4 + lq) * NP + 16 * kj + lc], tb = xsT[(D4 + (lq ^ 1)) * NP + 16 * (kj + kr) + lc];
      double ca = pa[0], cb = pb[0];
      for (int m = 4; m < D4; m += 4) {
        const double na = pa[m * NP], nb = pb[m * NP];
        d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ca, cb, d2, 0, 0, 1);
        ca = na; cb = nb;
      }
      d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ca, cb, d2, 0, 0, 1);
      d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ta, tb, d2, 0, 0, 1);
      return d2;
    };
    // One 16x16 tile (block row kj + kr, block column kj), the four values of this lane (transposed tile:
    // the lane owns a row piece), from its squared distances.
    auto tile_eval = [&](int kj, int kr, d4_t d2, double (&kt)[4]) {
      {
        // gfx950: a VALU read of the first three result pairs of an fp64 MFMA is interlocked (it stalls until the
        // MFMA is done), a read of the LAST pair is not -- and hipcc's wait states are those of the 8-pass gfx942
        // instruction: a loop-carried copy of d2 once read stale registers (tools/mfma_hazard_probe.hip,
        // profiles/r01_notes.md).  So the first pair is touched here before anything may read the rest;
        // tools/mfma_hazard_audit.py checks the compiled code object, __graft_entry__.build() refuses a bad one.
        double first = d2[0];
        asm volatile("v_max_f64 %0, %0, %0" : "+v"(first));
        d2[0] = first;
        asm volatile("" : "+v"(d2));
      }
      const int row = 16 * (kj + kr) + lc;
      const int col0 = 16 * kj + lq;
      if (from_matrix) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = col0 + 4 * g;
          double kv = 0.0;
          if (row < n && col < n) kv = row >= col ? Ag[(size_t)row * N + col] : Ag[(size_t)col * N + row];
          if (row == col) kv += diag_add;
          if (row >= n || col >= n) kv = row == col ? 1.0 : 0.0;
          kt[g] = kv;
        }
      } else {
        // wave-uniform fast path: an off-diagonal tile entirely inside the n valid points needs neither the
        // diagonal term nor the identity padding (saves ~10 VALU instructions per element)
        if (kr != 0 && 16 * (kj + kr) + 16 <= n) {
#pragma unroll
          for (int g = 0; g < 4; ++g) kt[g] = kernel_from_sqdist_scaled<KIND>(d2[g], kc0, kc1, kc2, exptab);
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int col = col0 + 4 * g;
            double kv = kernel_from_sqdist_scaled<KIND>(row == col ? 0.0 : d2[g], kc0, kc1, kc2, exptab);
            if (row == col) kv += diag_add;
            if (row >= n || col >= n) kv = row == col ? 1.0 : 0.0;
            kt[g] = kv;
          }
        }
      }
    };
    // The build is VALU work, two waves per SIMD -- except on the SIMD of the (otherwise idle) panel
    // wave.  With 7 update waves the panel wave therefore builds the last slots of the six update waves that
    // do not share its SIMD into LDS images, which their owners pick up after the barrier.  The second wave
    // of a SIMD (4, 5, 6) only gets the issue slots its older mate leaves (measured 3.0 k cycles per tile
    // against 2.6 k; the panel wave 3.8 k), so it hands over two tiles (slots 17, 18) and the first waves
    // one (slot 19): 9 of the 136 tiles.  (Round 1: three and one, 12 tiles; re-tuned at the end of round 2 together with
    // PSTORE below, after the panel loop had changed: 106.8 -> 105.3 us in interleaved A/B runs.)
    constexpr bool PANEL_BUILDS = (WU == 7 && NB == 16);
    constexpr int KMATE = 3;   // wave KMATE shares the panel wave's SIMD (waves go round-robin)
#ifndef SCAML_KOLD
#define SCAML_KOLD 19
#define SCAML_KYOUNG 17
#endif
    auto kslot_of = [](int w) { return w < KMATE ? SCAML_KOLD : SCAML_KYOUNG; };
    double* KT = lds + XROWS * NP;   // [6][4][256] register images, behind xsT
    if (!is_panel) {
      int kj = 0, kr = wave;  // column / row-in-column of the current slot's tile
      // (issuing the distances of the next tile ahead of the evaluation of this one was tried: no gain, the
      //  MFMA chain is short against the ~120 VALU instructions of the evaluation)
#define SCAML_KBUILD_(S, r0, r1, r2, r3, r4, r5, r6, r7)                                           \
      if (S < SLOTS && !(PANEL_BUILDS && wave != KMATE && S >= kslot_of(wave))) {                  \
        while (kj < NB && kr >= NB - kj) { kr -= NB - kj; ++kj; }                                  \
        if (kj < NB) {                                                                             \
          double kt[4];                                                                            \
          tile_eval(kj, kr, tile_dist(kj, kr), kt);                                                \
          TILE_SET(r0, r1, r2, r3, r4, r5, r6, r7, kt[0], kt[1], kt[2], kt[3]);                    \
          kr += WU;                                                                                \
        }                                                                                          \
      }
      SCAML_TILE_LIST(SCAML_KBUILD_)
#undef SCAML_KBUILD_
    } else if (PANEL_BUILDS) {
      for (int idx = 0; idx < 6 * 4; ++idx) {
        const int w6 = idx >> 2, w = w6 + (w6 >= KMATE), sl = kslot_of(w) + (idx & 3);
        const int t = sl * WU + w;
        if (sl >= SLOTS || t >= NT) continue;
        int kj = 0, kr = t;
        while (kr >= NB - kj) { kr -= NB - kj; ++kj; }
        double kt[4];
        tile_eval(kj, kr, tile_dist(kj, kr), kt);
        double* img = KT + idx * 256 + lane;
        img[0] = kt[0]; img[64] = kt[1]; img[128] = kt[2]; img[192] = kt[3];
      }
    }
#ifdef SCAML_STAMPS
    if (is_panel) { STAMP(8); } else { STAMP(9); }   // own share of the build done (diagnostic build only)
#endif
    __syncthreads();  // xsT dead from here
    if (PANEL_BUILDS && !is_panel && wave != KMATE) {
      const int w6 = wave - (wave > KMATE);
#define SCAML_KLOAD_(S, r0, r1, r2, r3, r4, r5, r6, r7)                                            \
      if (S >= kslot_of(wave) && S < SLOTS && S * WU + wave < NT) {                                \
        const double* img = KT + (w6 * 4 + (S - kslot_of(wave))) * 256 + lane;                     \
        TILE_SET(r0, r1, r2, r3, r4, r5, r6, r7, img[0], img[64], img[128], img[192]);             \
      }
      SCAML_TILE_LIST(SCAML_KLOAD_)
#undef SCAML_KLOAD_
    }
    __syncthreads();  // xsT dead from here: region A becomes PT / WAll
    STAMP(1);

    // ---- prologue: the first two diagonal tiles and R_1 = tile (1, 0) parked for the panel wave
    fail = 0;
    // park tile t (index in column-major order over the triangle) as a register image at dst[256] if
    // this wave owns it
    auto park_tile = [&](int t, double* dst) {
      if (t % WU == wave) {
        const int s = t / WU;
        double e0 = 0.0, e1 = 0.0, e2 = 0.0, e3 = 0.0;
#define SCAML_BODY(r0, r1, r2, r3, r4, r5, r6, r7) TILE_GET(r0, r1, r2, r3, r4, r5, r6, r7, e0, e1, e2, e3);
        SCAML_DISPATCH(s)
#undef SCAML_BODY
        dst[lane] = e0; dst[64 + lane] = e1; dst[128 + lane] = e2; dst[192 + lane] = e3;
      }
    };
    if (!is_panel) {
      park_tile(off(0), DG);
      if (NB > 1) {
        park_tile(off(1), DG + 256);
        park_tile(off(0) + 1, CR + 256);   // R_1 = tile (1, 0)
      }
    }
    __syncthreads();
    STAMP(2);

    // One finished sub-diagonal tile (ti, c): read back from the LDS column buffer in row segments -> HBM as
    // 128-byte stores; the mirrored upper tile is written as zeros by the same lanes on request (no
    // separate zero-fill pass over L).
    auto store_tile = [&](int c, int ti) {
      const double* prow = PT + (c % 3) * PANEL + (16 * ti + lq) * PP + lc;
      const double e[4] = {prow[0], prow[4 * PP], prow[8 * PP], prow[12 * PP]};
      double* tb = Lg + ((size_t)(16 * ti) * LD + 16 * c);   // tile (ti, c), wave-uniform
      double* mb = Lg + ((size_t)(16 * c) * LD + 16 * ti);   // mirrored tile (c, ti)
      if (16 * ti + 16 <= n) {                              // interior tile: no per-lane bounds
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          tb[(size_t)g * 4 * LD + lane_idx] = e[g];
          if (zero_upper) mb[(size_t)g * 4 * LD + lane_idx] = 0.0;
        }
      } else {
        const int col = 16 * c + lc, mc = 16 * ti + lc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * ti + lq + 4 * g, mr = 16 * c + lq + 4 * g;
          if (row < n && col < n) tb[(size_t)g * 4 * LD + lane_idx] = e[g];
          if (zero_upper && mr < n && mc < n) mb[(size_t)g * 4 * LD + lane_idx] = 0.0;
        }
      }
    };
    // The first PSTORE columns go to HBM from the panel wave: in the early panels it runs a whole step ahead
    // of the update waves (which are the bottleneck there) and has the time; later it is the bottleneck
    // itself and the update waves store their own tiles.  (8 columns: re-tuned at the end of round 2; it was 12.)
#ifndef SCAML_PSTORE
#define SCAML_PSTORE 8
#endif
    constexpr int PSTORE = (WU == 7 && NB == 16) ? SCAML_PSTORE : 0;

    // Forward substitution, by the panel wave alone: v_c = W_c y~_c, then column c (final in its LDS buffer) is folded into
    // the running right-hand side, y~_i -= sum_q L[i][16 c + q] v_c[q] for the rows below block c.  v_c sits in SGPRs
    // (the same 16 values for every lane), a lane owns whole rows: no atomics, no hand-off counter -- the update waves'
    // F phase used to do this per tile (redundant v_c per wave, in-lane dot products + LDS atomics, a counter per
    // column) on their critical path; the panel wave has the slack for it in the early panels, and the late columns are short.
    auto fold_column = [&](int c) {
      const double* Wc = WAll + c * 16 * PP;
      double v = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) v = __builtin_fma(Wc[(4 * lq + q) * PP + lc], ytil[16 * c + 4 * lq + q], v);
      v = sum_lane_groups(v);
      if (lq == 0) vv[16 * c + lc] = v;
      if (16 * (c + 1) >= NP) return;
      double vq[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) vq[q] = readlane_f64(v, q);
      const double* col = PT + (c % 3) * PANEL;
      for (int i = 16 * (c + 1) + lane; i < NP; i += 64) {
        const double* row = col + i * PP;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
          s0 = __builtin_fma(row[q], vq[q], s0);
          s1 = __builtin_fma(row[q + 1], vq[q + 1], s1);
        }
        ytil[i] -= s0 + s1;
      }
    };

    if (is_panel) {
      // ================= panel wave: the chain of diagonal blocks, running ahead of the update =========
      // (it is the youngest wave on its SIMD and would lose every issue slot to the update wave
      //  streaming MFMAs next to it: raised priority for the whole chain)
      __builtin_amdgcn_s_setprio(3);
      for (int j = 0; j < NB; ++j) {
        double* Wj = WAll + j * 16 * PP;
        // D_j (updates of panels <= j-2 applied) and the raw tile R_j = A[j][j-1] were parked by their
        // owners during U1(j-2): one whole step of slack, the chain does not wait in steady state
        if (j >= 2) sync_wait_ge(cntS + j - 2, WU);
        STAMP(3);
        d4_t a;
        {
          const double* dg = DG + (j & 1) * 256 + lane;   // symmetric: the transposed image is the block
#pragma unroll
          for (int g = 0; g < 4; ++g) a[g] = dg[64 * g];
        }
        if (j >= 1) {
          // tm = W_{j-1} R_j^T = (L_j,j-1)^T: in the C/D layout that is L_j,j-1 in operand position,
          // so D_j -= L L^T follows without a transpose through LDS
          const double* pw = WAll + (j - 1) * 16 * PP + lq * PP + lc;   // W[lc][lq + 4m] out of the transposed copy
          const double* pr = CR + (j & 1) * 256 + lane;   // register image of the transposed tile = R[lc][lq + 4m]
          d4_t tm = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int m = 0; m < 4; ++m) tm = __builtin_amdgcn_mfma_f64_16x16x4f64(pw[4 * m * PP], pr[64 * m], tm, 0, 0, 0);
#pragma unroll
          for (int m = 0; m < 4; ++m) a = __builtin_amdgcn_mfma_f64_16x16x4f64(tm[m], tm[m], a, 0, 0, 1);
        }
        STAMP(4);
        double* LTj = LT + (j & 1) * 16 * PP;
        const int bad = potf2_inv_block(a, LTj, Wj, trash, j, lane);
        STAMP(5);
        if (lq == 0) dl[16 * j + lc] = LTj[lc * PP + lc];   // diag(L) for logdet
        if (bad) {
          if (lane == 0) flagp[0] = bad;
          sync_publish(flagW + j, 2, lane);
          break;
        }
        sync_publish(flagW + j, 1, lane);
        STAMP_AT(j);
        STAMP(9);
        // Column j-2 (final in its LDS buffer since cntS[j-2]) goes to HBM and into the running right-hand side AFTER
        // flagW[j] is out: with three column buffers its buffer is not reused before F(j+1), i.e. before this wave has
        // published flagW[j+1] -- the stores and the fold (3.3 k cycles per step) are off the path to flagW.
        if (Lg && j >= 2 && j - 2 < PSTORE) {
          // (SCAML_STRIP tiles per trip: their LDS reads are issued together, one exposed latency for all)
#ifndef SCAML_STRIP
#define SCAML_STRIP 4
#endif
          const int c = j - 2;
          int ti = j - 1;
          for (; ti + SCAML_STRIP - 1 < NB && 16 * (ti + SCAML_STRIP) <= n; ti += SCAML_STRIP) {
            const double* prow = PT + (c % 3) * PANEL + (16 * ti + lq) * PP + lc;
            double e[4 * SCAML_STRIP];
#pragma unroll
            for (int u = 0; u < 4 * SCAML_STRIP; ++u) e[u] = prow[4 * u * PP];
            double* tb = Lg + ((size_t)(16 * ti) * LD + 16 * c) + lane_idx;
#pragma unroll
            for (int u = 0; u < 4 * SCAML_STRIP; ++u) tb[(size_t)4 * u * LD] = e[u];
            if (zero_upper) {
              double* mb = Lg + ((size_t)(16 * c) * LD + 16 * ti) + lane_idx;
#pragma unroll
              for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int u = 0; u < SCAML_STRIP; ++u) mb[(size_t)g * 4 * LD + 16 * u] = 0.0;
            }
          }
          for (; ti < NB; ++ti) store_tile(c, ti);
        }
        if (j >= 2) fold_column(j - 2);
        if (j == 1 && lane < WU) {
          // Off the critical path (the chain waits for parked tiles next anyway): which off-diagonal tiles of
          // each block row the update waves hold, for the back-substitution at the end -- lane w lists wave w.
          int* rl = rowlist + lane * NB * 8;
#pragma clang loop unroll(disable)
          for (int i = 0; i < NB; ++i) rl[8 * i] = 0;
          int jj = 0, r = lane;
#pragma clang loop unroll(disable)
          for (int sl = 0; sl < SLOTS; ++sl) {
            while (jj < NB && r >= NB - jj) { r -= NB - jj; ++jj; }
            if (jj >= NB) break;
            if (r > 0) {
              const int i = jj + r, c = rl[8 * i];
              if (c < 7) { rl[8 * i + 1 + c] = ((jj == i - 1) << 16) | (sl << 8) | jj; rl[8 * i] = c + 1; }   // bit 16: next to the diagonal
            }
            r += WU;
          }
        }
      }
      if (!flagp[0]) {
        // the last two columns: column NB-2 once every update wave has finalised its tiles of it (nothing reuses its
        // buffer any more), column NB-1 has no rows below it
        if (sync_wait_ge_or_fail(cntT + NB - 2, WU, flagp)) {
          fold_column(NB - 2);
          fold_column(NB - 1);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    } else {
      // ================= update waves ================================================================
#ifndef SCAML_YOUNG_PRIO
#define SCAML_YOUNG_PRIO 1
#endif
      // (the second wave of a SIMD loses the issue arbitration to its older mate in every phase and is the iteration's
      //  long pole: a static raise for that half -- 107.7 -> 107.0 us in interleaved A/B, priority 2 the same)
      if (WU == 7 && wave >= 4) __builtin_amdgcn_s_setprio(SCAML_YOUNG_PRIO);
      auto store_column = [&](int c) {
        if (Lg && c >= PSTORE) {
          const int sa = slo(c), sb = slo(c + 1), offc = off(c);
          for (int s = sa; s < sb; ++s) {
            const int ti = c + (s * WU + wave - offc);
            if (ti != c) store_tile(c, ti);   // (the diagonal tile: store_diag)
          }
        }
      };
      // L_cc left its owner's registers long ago: it comes from the panel wave's LDS copy LT[c & 1], which
      // step c+2 overwrites once cntS[c] is complete -- so this runs before the wave's arrival there
      auto store_diag = [&](int c) {
        if (Lg && off(c) % WU == wave) {
          const double* lt = LT + (c & 1) * 16 * PP + lq * PP + lc;
          double* tb = Lg + ((size_t)(16 * c) * LD + 16 * c);
          const int col = 16 * c + lc;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = 16 * c + lq + 4 * g;
            const double e = lt[4 * g * PP];
            if (row < n && col < n && (zero_upper || col <= row)) tb[(size_t)g * 4 * LD + lane_idx] = e;
          }
        }
      };
      // iteration k = -1 only finalises column 0
      for (int k = -1; k + 1 < NB; ++k) {
        const int c = k + 1;                          // the column finalised in this iteration
        const double* buf = PT + ((k + 3) % 3) * PANEL;     // column k, final (operand reads)
        double* cbuf = PT + (c % 3) * PANEL;          // receives column c
        if (k >= 0) {
          // (every wait below gives up when the panel wave has reported a failed pivot: the waves it is
          //  waiting for may have left already)
          STAMP_K(k, 0);
          if (!sync_wait_ge_or_fail(cntT + k, WU, flagp)) goto update_done;   // column k final in LDS
          STAMP_K(k, 1);
          STAMP(3);
          // U1: column k+1 and the diagonal tile D_{k+2} first
          const int sa = slo(c), sb = slo(c + 1), offc = off(c);
          const double* pj = buf + (16 * c + lc) * PP + lq;         // L_{k+1,k} rows: the A operand
          for (int s = sa; s < sb; ++s) {
            const int ti = c + (s * WU + wave - offc);
            if (ti == c) continue;   // D_{k+1} left during U1(k-1)
            const double* pi = buf + (16 * ti + lc) * PP + lq;      // L_{ti,k} rows: the B operand
#define SCAML_BODY(r0, r1, r2, r3, r4, r5, r6, r7) \
            TILE_MFMA4_SUB(r0, r1, r2, r3, r4, r5, r6, r7, pj[0], pj[4], pj[8], pj[12], pi[0], pi[4], pi[8], pi[12]);
            SCAML_DISPATCH(s)
#undef SCAML_BODY
          }
          const int td = off(k + 2);     // tile (k+2, k+2)
          const bool own_d = k + 2 < NB && td % WU == wave;
          if (own_d) {
            const int s = td / WU;
            const double* pd = buf + (16 * (k + 2) + lc) * PP + lq;
#define SCAML_BODY(r0, r1, r2, r3, r4, r5, r6, r7) \
            TILE_MFMA4_SUB(r0, r1, r2, r3, r4, r5, r6, r7, pd[0], pd[4], pd[8], pd[12], pd[0], pd[4], pd[8], pd[12]);
            SCAML_DISPATCH(s)
#undef SCAML_BODY
          }
          MFMA_DRAIN();
          STAMP_K(k, 2);
          if (k + 2 < NB) park_tile(offc + 1, CR + (k & 1) * 256);   // R_{k+2} = tile (k+2, k+1), raw
          if (own_d) park_tile(td, DG + (k & 1) * 256);
          sync_arrive(cntS + k, lane);
          STAMP_K(k, 3);
          STAMP_AT(k);
          STAMP(4);
        }
        if (k >= 0) {
          // U2: the bulk of the trailing update with panel k: every slot from slo(k+2) on, entered through
          // one switch and then falling through slot after slot (the parked D_{k+2} is skipped).
          // Round 2: U2(k) runs BEFORE F(k+1) -- the time a wave used to spend in front of flagW[k+1] waiting for the
          // panel wave (1.4-2.9 k cycles per iteration) is bulk work now; cntT[k+1] arrives later, but nobody needs it
          // before his own U2(k) is through (109.2 -> 108.0 us in interleaved A/B; round 1 had F first).
          const int s0 = slo(k + 2);
          int uj = k + 2, ur = s0 * WU + wave - off(k + 2);
#define SCAML_U2_(S, r0, r1, r2, r3, r4, r5, r6, r7)                                               \
          case S:                                                                                  \
            if (S < SLOTS) {                                                                       \
              while (uj < NB && ur >= NB - uj) { ur -= NB - uj; ++uj; }                            \
              if (uj < NB) {                                                                       \
                if (ur != 0 || uj != k + 2) {                                                      \
                  const double* pj = buf + (16 * uj + lc) * PP + lq;                               \
                  const double* pi = buf + (16 * (uj + ur) + lc) * PP + lq;                        \
                  TILE_MFMA4_SUB(r0, r1, r2, r3, r4, r5, r6, r7, pj[0], pj[4], pj[8], pj[12], pi[0], pi[4], pi[8], pi[12]); \
                }                                                                                  \
                ur += WU;                                                                          \
              }                                                                                    \
            }
          switch (s0) { SCAML_TILE_LIST(SCAML_U2_) default: break; }
#undef SCAML_U2_
          MFMA_DRAIN();
          sync_arrive(cntU + k, lane);
          STAMP_K(k, 11);
          STAMP(7);
        }
        // ---- F(c): column c becomes final
        {
          int fw;
          while ((fw = sync_peek(flagW + c)) == 0) __builtin_amdgcn_s_sleep(SCAML_SLEEP);
          STAMP_K(k, 4);
          STAMP(5);
          if (fw == 2) break;
          const double* Wc = WAll + c * 16 * PP;
          const int sa = slo(c), sb = slo(c + 1), offc = off(c);
          // everyone must be done reading column c-3 (operands of U2(c-3), its HBM stores) before its buffer -- one of three -- is reused
          if (sa < sb && c >= 3 && !sync_wait_ge_or_fail(cntU + c - 3, WU, flagp)) goto update_done;
          STAMP_K(k, 6);
          const double* pw = Wc + lq * PP + lc;   // W[lc][lq + 4m] out of the transposed copy
          const double w0 = pw[0], w1 = pw[4 * PP], w2 = pw[8 * PP], w3 = pw[12 * PP];
          d4_t t[MAXC];
#pragma unroll
          for (int u = 0; u < MAXC; ++u) {
            const int s = sa + u;
            if (s < sb && c + (s * WU + wave - offc) != c) {
#define SCAML_BODY(r0, r1, r2, r3, r4, r5, r6, r7) TILE_TRSM_ISSUE_UT(r0, r1, r2, r3, r4, r5, r6, r7, w0, w1, w2, w3, t[u]);
              SCAML_DISPATCH(s)
#undef SCAML_BODY
            }
          }
          MFMA_DRAIN();
          STAMP_K(k, 7);
          // The finished tiles come off the matrix core UN-transposed (lane (lc, lq) register g = L[16 ti + lq + 4 g][16 c + lc]):
          // into the LDS column buffer for everyone's operand reads, and back into their accumulator registers as they
          // are -- nothing updates them any more, and the back-substitution at the end contracts over rows, which this
          // layout keeps inside a lane.  (The forward substitution -- v_c and the fold of column c into the running
          // right-hand side -- is the panel wave's: fold_column.)
#pragma unroll
          for (int u = 0; u < MAXC; ++u) {
            const int s = sa + u;
            const int ti = c + (s * WU + wave - offc);
            if (s < sb && ti != c) {
              asm volatile("" : "+v"(t[u]));   // the values exist only after the drain
              double* prow = cbuf + (16 * ti + lq) * PP + lc;
              prow[0] = t[u][0]; prow[4 * PP] = t[u][1]; prow[8 * PP] = t[u][2]; prow[12 * PP] = t[u][3];
            }
          }
          sync_arrive(cntT + c, lane);
          STAMP_K(k, 8);
#pragma unroll
          for (int u = 0; u < MAXC; ++u) {
            const int s = sa + u;
            const int ti = c + (s * WU + wave - offc);
            if (s < sb && ti != c) {
              const double e0 = t[u][0], e1 = t[u][1], e2 = t[u][2], e3 = t[u][3];
#define SCAML_BODY(r0, r1, r2, r3, r4, r5, r6, r7) TILE_SET(r0, r1, r2, r3, r4, r5, r6, r7, e0, e1, e2, e3);
              SCAML_DISPATCH(s)
#undef SCAML_BODY
            }
          }
          STAMP_K(k, 9);
          store_diag(c);
          STAMP_K(k, 10);
          STAMP(6);
        }
        store_column(c);
        STAMP_K(k, 12);
        STAMP(8);
      }
    update_done:;
    }
  }
  __syncthreads();
  fail = flagp[0];

  // ---- scalars: quad, logdet (panel wave), then alpha by blocked back-substitution
  STAMP(15);
#if defined(SCAML_STAMPS_PER_PANEL) && !defined(SCAML_STAMPS_ONE_PANEL)
  if (!is_panel) { st_acc[14] = __builtin_amdgcn_s_memtime() - st_prev; }   // loop left (slot 14 = U1(14) never runs... overwritten on purpose)
#endif
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
      double* Wg = p.Linv_diag + (size_t)task * b.sW((size_t)nbn * 256);
      for (int e = tid; e < nbn * 256; e += NTHREADS)   // W[r][c] of block b = WAll[b][c][r] (kept transposed)
        Wg[e] = WAll[(e >> 8) * 16 * PP + (e & 15) * PP + ((e >> 4) & 15)];
    }
    bool want_tail = p.alpha != nullptr;
    if constexpr (BLK) {
      // first block of a blocked fit: the caller continues from v = L^-1 y (the back-substitution runs once, over both
      // blocks, in gp_blocked_finish_kernel)
      if (want_tail && (p.flags & FIT_FORWARD_ONLY)) {
        for (int r = tid; r < n; r += NTHREADS) p.alpha[(size_t)task * b.sy((size_t)N) + r] = vv[r];
        want_tail = false;
      }
    }
    if (want_tail) {
      // alpha = L^-T v by blocks from the bottom: alpha_k = W_k^T w_k, w_j -= L_kj^T alpha_k for j < k.  The
      // dependency chain runs through the tiles next to the diagonal, w_{k-1} <- alpha_k <- w_k, so ONE wave
      // (the panel wave) walks it alone, out of LDS and registers, on the matrix core:
      //   w_{k-1} = (w_{k-1} with every other tile of column k-1 folded in) - M_k^T w_k,  M_k = W_k L_{k,k-1},
      // 4 dependent MFMAs per step whose result (replicated over lc, register g = element lq + 4g) is already
      // the B operand of the next step.  alpha_k itself is off the chain (VALU, published to the update waves),
      // and so are the other tiles: the update waves fold alpha_k into w_j, j < k-1, for the tiles (k, j) they
      // hold in registers -- their results are needed one chain step later at the earliest.
      double* MK = PT;   // [NB][16][PP]: M_k at lane (lc, lq) element g -> MK[k][(lq + 4g) * PP + lc]; the panels are free by now
      for (int r = tid; r < NP; r += NTHREADS) ww[r] = vv[r];
      if (tid < NB) flagW[tid] = 0;   // reused: alpha_k published
      if (!is_panel) {
        // M_k = W_k L_{k,k-1} by the owner of that tile (un-transposed in its registers = B operand position)
        // (tile (k, k-1) is number off(k-1) + 1 in the column-major tile order: wave t % WU, slot t / WU)
        for (int k = 1; k < NB; ++k) {
          const int t = off(k - 1) + 1;
          if (t % WU != wave) continue;
          const int sl = t / WU;
          const double* pw = WAll + k * 16 * PP + lq * PP + lc;   // W[lc][lq + 4m] out of the transposed copy
          const double w0 = pw[0], w1 = pw[4 * PP], w2 = pw[8 * PP], w3 = pw[12 * PP];
          d4_t mk;
#define SCAML_BODY(r0, r1, r2, r3, r4, r5, r6, r7) TILE_TRSM_TO_V(r0, r1, r2, r3, r4, r5, r6, r7, w0, w1, w2, w3, mk);
          SCAML_DISPATCH(sl)
#undef SCAML_BODY
          double* pm = MK + k * 16 * PP + lq * PP + lc;
          pm[0] = mk[0]; pm[4 * PP] = mk[1]; pm[8 * PP] = mk[2]; pm[12 * PP] = mk[3];
        }
      }
      __syncthreads();
      STAMP(11);
      if (is_panel) {
        d4_t c = {0.0, 0.0, 0.0, 0.0};
        // operands of a step are fetched one step ahead (they do not depend on the chain)
        const double* pm = MK + (NB - 1) * 16 * PP + lq * PP + lc;   // A operand: M_k[lq + 4m][lc]
        const double* Wk = WAll + (NB - 1) * 16 * PP + lc * PP + lq;  // W_k[lq + 4g][lc] out of the transposed copy
        double m0 = pm[0], m1 = pm[4 * PP], m2 = pm[8 * PP], m3 = pm[12 * PP];
        double u0 = Wk[0], u1 = Wk[4], u2 = Wk[8], u3 = Wk[12];
        for (int k = NB - 1; k >= 0; --k) {
          // every tile (i, k), i >= k + 2, folded into w_k by its owner (the one next to the diagonal is `c`).
          // The counter and w_k are read in ONE trip: the LDS unit serves a wave's requests in order and every
          // owner's atomic add precedes its arrival, so a complete count means the values read behind it are final.
          const int need = NB - 2 - k;
          const double* pwk = ww + 16 * k + lq;
          double wk0, wk1, wk2, wk3;
          for (;;) {
            const int have = need > 0 ? __hip_atomic_load((const lds_int_t*)(cntB + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
            asm volatile("" ::: "memory");
            wk0 = pwk[0]; wk1 = pwk[4]; wk2 = pwk[8]; wk3 = pwk[12];
            asm volatile("" ::: "memory");
            if (have >= need) break;
            __builtin_amdgcn_s_sleep(SCAML_SLEEP);
          }
          STAMP(13);   // (panel-wave column of the diagnostic build: time spent waiting for w_k)
          {
            // gfx950: the last result pair of an fp64 MFMA is not interlocked; touch the first one before it
            double c0 = c[0];
            asm volatile("v_max_f64 %0, %0, %0" : "+v"(c0));
            c[0] = c0;
            asm volatile("" : "+v"(c));
          }
          wk0 -= c[0]; wk1 -= c[1]; wk2 -= c[2]; wk3 -= c[3];
          if (k > 0) {
            d4_t cn = {0.0, 0.0, 0.0, 0.0};
            cn = __builtin_amdgcn_mfma_f64_16x16x4f64(m0, wk0, cn, 0, 0, 0);
            cn = __builtin_amdgcn_mfma_f64_16x16x4f64(m1, wk1, cn, 0, 0, 0);
            cn = __builtin_amdgcn_mfma_f64_16x16x4f64(m2, wk2, cn, 0, 0, 0);
            cn = __builtin_amdgcn_mfma_f64_16x16x4f64(m3, wk3, cn, 0, 0, 0);
            c = cn;
          }
          // alpha_k[lc] = sum_r W_k[r][lc] w_k[r], in the shadow of the MFMAs
          double ak = u0 * wk0;
          ak = __builtin_fma(u1, wk1, ak);
          ak = __builtin_fma(u2, wk2, ak);
          ak = __builtin_fma(u3, wk3, ak);
          if (k > 0) {
            pm -= 16 * PP; Wk -= 16 * PP;
            if (k > 1) { m0 = pm[0]; m1 = pm[4 * PP]; m2 = pm[8 * PP]; m3 = pm[12 * PP]; }
            u0 = Wk[0]; u1 = Wk[4]; u2 = Wk[8]; u3 = Wk[12];
          }
          ak = sum_lane_groups(ak);   // every lane (lc, *) now holds alpha_k[lc]
          if (lq == 0) dl[16 * k + lc] = ak;   // dl is free by now: alpha is collected there
          sync_publish(flagW + k, 1, lane);
          STAMP(12);
        }
      } else {
        // tiles (k, j), j < k - 1, held by this wave: register g of lane (lc, lq) is L_kj[lq + 4 g][lc], so the
        // lane needs alpha_k[lq + 4 g] and adds its four terms; the four lane groups meet in w_j[lc] through
        // LDS fp64 atomics.  The tile closest to the diagonal first (its column is needed first).
        for (int k = NB - 1; k >= 2; --k) {
          const int* rl = rowlist + (wave * NB + k) * 8;
          const int cnt = rl[0];
          if (cnt == 0 || (cnt == 1 && (rl[1] >> 16))) continue;   // nothing of this block row here: do not even wait
          sync_wait_ge(flagW + k, 1);
          const double* pa = dl + 16 * k + lq;
          const double a0 = pa[0], a1 = pa[4], a2 = pa[8], a3 = pa[12];
          for (int i = cnt - 1; i >= 0; --i) {
            const int sj = rl[1 + i], s = (sj >> 8) & 0xff, j = sj & 0xff;
            if (sj >> 16) continue;   // the tile next to the diagonal went into M_k
            double e0, e1, e2, e3;
#define SCAML_BODY(r0, r1, r2, r3, r4, r5, r6, r7) TILE_GET(r0, r1, r2, r3, r4, r5, r6, r7, e0, e1, e2, e3);
            SCAML_DISPATCH(s)
#undef SCAML_BODY
            double part = e0 * a0;
            part = __builtin_fma(e1, a1, part);
            part = __builtin_fma(e2, a2, part);
            part = __builtin_fma(e3, a3, part);
            __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)(ww + 16 * j + lc), -part);
            sync_arrive(cntB + j, lane);
          }
          STAMP(13);
        }
      }
      __syncthreads();
      STAMP(14);
      for (int r = tid; r < n; r += NTHREADS) p.alpha[(size_t)task * b.sy((size_t)N) + r] = dl[r];
    }
  }
  STAMP(10);
#if defined(SCAML_STAMPS_PER_PANEL) && !defined(SCAML_STAMPS_ONE_PANEL)
  if (!is_panel) { st_acc[15] = __builtin_amdgcn_s_memtime() - st_prev; }   // end of the attempt
#endif
  STAMP_FLUSH(task);
  return fail;
}

template <int NB, int WU, int KIND, bool BLK>
__device__ __noinline__ int gp_fit_retry(const FitParams& p, const double jitter, const FitBlk<BLK> b) {
  return gp_fit_attempt<NB, WU, KIND, BLK>(p, jitter, b);
}

template <int NB, int WU, int KIND, bool BLK>
__device__ __forceinline__ void gp_fit_main(const FitParams& p, const FitBlk<BLK> b) {
  // jitter escalation of linear_operator's psd_safe_cholesky, per task, without leaving the GPU
  double jitter = 0.0;
  int fail = gp_fit_attempt<NB, WU, KIND, BLK>(p, 0.0, b);
#ifndef SCAML_STAMPS  // (the diagnostic build times the first attempt only)
  if (fail && !(p.flags & SCAML_FIT_NO_RETRY)) {
    const FitParams pc = p;  // only this cold copy has its address taken; `p` stays in the kernarg segment
    for (int attempt = 1; attempt < 4 && fail; ++attempt) {
      jitter = attempt == 1 ? 1e-8 : (attempt == 2 ? 1e-7 : 1e-6);
      fail = gp_fit_retry<NB, WU, KIND, BLK>(pc, jitter, b);
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

template <int NB, int WU, int KIND>
__global__ __launch_bounds__((WU + 1) * 64) void gp_fit_fused_kernel(FitParams p) {
  gp_fit_main<NB, WU, KIND, false>(p, FitBlk<false>{});
}

// the same fit in place on a diagonal block of a larger task (strided addressing; tasks may be switched off)
template <int NB, int WU, int KIND>
__global__ __launch_bounds__((WU + 1) * 64) void gp_fit_blocked_kernel(FitParams p, FitBlockParams bp) {
  if (bp.active && bp.active[blockIdx.x] == 0) return;   // (jitter rounds: this task is already done)
  gp_fit_main<NB, WU, KIND, true>(p, FitBlk<true>{bp});
}

}  // namespace scaml

// Explicit instantiations: one kernel per padded size class (NB 16-blocks, WU update waves) and
// kernel kind (see VGPR_CAPS in __graft_entry__.py for the register budgets).
#define SCAML_INSTANTIATE(NB, WU)                                                   \
  template __global__ void scaml::gp_fit_fused_kernel<NB, WU, 0>(scaml::FitParams); \
  template __global__ void scaml::gp_fit_fused_kernel<NB, WU, 1>(scaml::FitParams); \
  template __global__ void scaml::gp_fit_blocked_kernel<NB, WU, 0>(scaml::FitParams, scaml::FitBlockParams); \
  template __global__ void scaml::gp_fit_blocked_kernel<NB, WU, 1>(scaml::FitParams, scaml::FitBlockParams);
SCAML_INSTANTIATE(2, 1)
SCAML_INSTANTIATE(4, 3)
SCAML_INSTANTIATE(8, 3)
SCAML_INSTANTIATE(8, 7)   // wide variant for N <= 128: one workgroup per CU, used when the tasks do not fill the CUs twice
SCAML_INSTANTIATE(16, 7)

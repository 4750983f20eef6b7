// gp_fit_coop.hip -- the fit beyond 256 points per task with SEVERAL compute units per task, one launch (gfx950).
//
// scamlgp/model.py:176-188 -> scamlgp/utils.py:171-177 fit every source GP by gpytorch's ExactGP chain (kernel matrix ->
// psd_safe_cholesky -> cholesky_solve); the 512-point tasks of scamlgp/benchmarking/configurations/
// hartmann6_ablation_num_points_per_task.py:17-18 do not fit one CU's registers.  gp_fit_blocked.hip factors them as a 2 x 2 block
// matrix with two single-CU factorisations in a row (212 of 319 us at T = 32 on 32 of 256 CUs).  Here P workgroups share a task:
//
//   * block columns of 32, dealt round-robin to the task's P workgroups (column j -> part j mod P);
//   * LEFT-looking: block (i, j) = K(X_i, X_j) - sum_{c < j} L_ic L_jc^T is computed once, by its owner, from finished blocks of
//     earlier columns read straight from the output L (through L2: every tile has ONE writer and is written ONCE per attempt);
//   * the diagonal block is factored and inverted by one wave in registers (two 16 x 16 sweeps of v_readlane + MFMA glue), the
//     blocks below it become L_ij = C_ij L_jj^-T on the matrix cores and are published two block rows at a time;
//   * forward substitution v = L^-1 y rides along (v_j from the row block j the owner reads anyway), the backward substitution
//     alpha = L^-T v and the scalars are done by the owner of the last column once that column is through.
//
// Inter-workgroup protocol (one launch, no grid barrier): every handed-off byte -- L blocks, v, the per-column partial sums -- is stored
// write-through (sc1) and drained (s_waitcnt vmcnt(0) by every storing wave, then the workgroup barrier) before ONE lane publishes
// prog[j] = attempt << 8 | (block rows of column j finished); consumers poll prog[] with sc1 loads and read the payload with sc1 loads
// only (L1 is bypassed: no acquire fence needed, results do not depend on which XCD a workgroup runs on).  Spins are bounded.
// psd_safe_cholesky's jitter ladder (0, 1e-8, 1e-7, 1e-6 on the whole diagonal of a failing task) is kept exactly: a non-positive pivot
// raises the task's failure count, every workgroup of the task abandons the attempt at its next poll and starts the next one; tags
// carry the attempt number, so nothing has to be reset.  All workgroups of a launch must be resident at once: the host launches
// at most one workgroup per CU (csrc/scaml_host.cpp).
#include "../../include/scaml_gp.h"
#include "scaml_common.hpp"
#include "gp_fit_params.h"

namespace scaml {

typedef unsigned cf_u4 __attribute__((ext_vector_type(4)));
typedef double cf_d2 __attribute__((ext_vector_type(2)));
#define CF_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

#ifndef CF_BGROUP
#define CF_BGROUP 4
#endif
constexpr int CF_BP = 33;               // pitch of the 32 x 32 blocks in LDS
constexpr int CF_BS = 32 * CF_BP;
constexpr int CF_SPIN_LIMIT = 4000000;  // polls (~1 us each: ~4 s) before a workgroup gives the task up: a protocol error, never a wait
constexpr unsigned CF_DEAD = 255u;      // failure count that stops every attempt (time-out)

// developer build (-DCF_STAMPS): 100 MHz wall-clock stamps of task 0, thread 0 of every part, into the workspace behind `part`
// (tools/dev_coop_stamps.py): [part][column slot][16]
#ifdef CF_STAMPS
#define CF_STAMP(k) do { if (task == 0 && tid == 0) p.part[(size_t)p.T * 64 + ((size_t)part * 32 + (j / P)) * 16 + (k)] = (double)wall_clock64(); } while (0)
#else
#define CF_STAMP(k) do { } while (0)
#endif
#define CF_FSTAMP(k) do { const int j = part + 30 * P; CF_STAMP(k); } while (0)   // (slot 30: inside the finish)

__device__ __forceinline__ void cf_settle(d4_t& v) { asm volatile("s_nop 15\n\ts_nop 2" : "+v"(v)); }   // gfx950: last MFMA result pair not interlocked
// workgroup barrier for LDS traffic only: __syncthreads() also waits for every global load in flight (its fences), which undoes a prefetch
__device__ __forceinline__ void cf_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// payload store: write-through (sc1) in general; a plain store where every workgroup of the task sits on the same XCD (`near`): the line
// then stays in that XCD's L2, which is where the consumers' sc1 loads look it up (an sc1 store drops it: every consumer read goes to memory)
__device__ __forceinline__ void cf_store(double* ptr, double v, bool near) {
  if (near) *ptr = v;
  else __hip_atomic_store(ptr, v, CF_RLX_AGENT);
}
__device__ __forceinline__ void cf_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// 1 / sqrt(d) for ANY positive finite double, without a branch (rsqrt_pos's slow path is one: with it in the middle of the pivot sweep
// hipcc sinks the multiplier FMAs below the branch and keeps every broadcast alive across it -- 600 SGPR spills): the exponent is
// split off exactly (d = m 4^h, m in [0.5, 2)), the f32-seeded step works on m.
__device__ __forceinline__ double cf_rsqrt(double d) {
  const int h = __builtin_amdgcn_frexp_exp(d) >> 1;
  return __builtin_ldexp(rsqrt_seeded(__builtin_ldexp(d, -2 * h)), -h);
}

// Cholesky factor and inverse of the 16 x 16 block at src (LDS, pitch CF_BP) by ONE wave: lane l (and its mirrors l + 16 m) is row
// l & 15 of L and column l & 15 of W = L^-1; pivots and multipliers travel by v_readlane (csrc/gp_target_fit.hip has the same sweep).
// L (zeros above the diagonal) -> Ld, W -> Wd.  Returns 0, or 1 + the index of the first pivot that is not positive.
typedef __attribute__((address_space(3))) double cf_lds_double;

// The pivot sweep's building blocks.  The multipliers L[b][cc] of a pivot step sit one per lane (lane b of every 16-lane row: the four
// rows are mirrors); every lane needs all of them.  gfx950's 64-bit DPP takes exactly that pattern as an operand modifier --
// row_newbcast:b = lane b of the row, to every lane of the row -- so one v_fmac_f64_dpp does what two v_readlane_b32, their SGPR hazard
// slot and an FMA did: ~840 instructions per sweep instead of ~2,300.  (Inline asm: hipcc has no builtin for the 64-bit form; the
// s_nop covers the two wait states a DPP read needs behind the VALU write of its source, which the compiler's hazard recogniser does
// not see inside an asm.)
template <int B>
__device__ __forceinline__ void cf_upd(double (&row)[16], double (&sp)[16], double x, double nx, double wcc) {
  if constexpr (B < 16) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(row[B]) : "v"(x), "v"(nx), "n"(B));
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(sp[B]) : "v"(x), "v"(wcc), "n"(B));
  }
}
#define CF_FENCE(v) asm volatile("" : "+v"(v))
// Pivot step CC, entered with rsq = 1 / sqrt(pivot CC); leaves with the next pivot's.  The reciprocal square root of the NEXT pivot is a
// chain of ~25 dependent instructions and needs nothing but row[CC + 1] after this step's first update: its pieces are placed between
// the remaining updates (empty asm fences pin every piece between two of them), where an in-order wave would otherwise wait for one
// result after the other before it even starts on the updates.
template <int CC>
__device__ __forceinline__ void cf_step(double (&row)[16], double (&sp)[16], double (&wv)[16], double& rsq, int& bad, int lc) {
  const double x = row[CC] * rsq;   // lane CC: pivot / sqrt(pivot)
  row[CC] = x;
  const double wcc = ((lc == CC ? 1.0 : 0.0) - sp[CC]) * rsq;
  wv[CC] = wcc;
  const double nx = -x;
  asm volatile("s_nop 1" ::"v"(x), "v"(nx), "v"(wcc));
  cf_upd<CC + 1>(row, sp, x, nx, wcc);
  if constexpr (CC < 15) {
    double pv;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(pv) : "v"(row[CC + 1]), "n"(CC + 1));
    if (!(pv > 0.0) && bad == 0) bad = CC + 2;
    cf_upd<CC + 2>(row, sp, x, nx, wcc);
    const int h = __builtin_amdgcn_frexp_exp(pv) >> 1;
    double m = __builtin_ldexp(pv, -2 * h);   // in [0.5, 2): the f32-seeded step works on it, whatever the pivot's exponent
    CF_FENCE(m);
    cf_upd<CC + 3>(row, sp, x, nx, wcc);
    double y = (double)__builtin_amdgcn_rsqf((float)m);
    CF_FENCE(y);
    cf_upd<CC + 4>(row, sp, x, nx, wcc);
    double e = __builtin_fma(-(m * y), y, 1.0);
    CF_FENCE(e);
    cf_upd<CC + 5>(row, sp, x, nx, wcc);
    double pp = __builtin_fma(e, 0.375, 0.5), ye = y * e;
    CF_FENCE(pp);
    CF_FENCE(ye);
    cf_upd<CC + 6>(row, sp, x, nx, wcc);
    rsq = __builtin_ldexp(__builtin_fma(ye, pp, y), -h);
    CF_FENCE(rsq);
    cf_upd<CC + 7>(row, sp, x, nx, wcc);
    cf_upd<CC + 8>(row, sp, x, nx, wcc);
    cf_upd<CC + 9>(row, sp, x, nx, wcc);
    cf_upd<CC + 10>(row, sp, x, nx, wcc);
    cf_upd<CC + 11>(row, sp, x, nx, wcc);
    cf_upd<CC + 12>(row, sp, x, nx, wcc);
    cf_upd<CC + 13>(row, sp, x, nx, wcc);
    cf_upd<CC + 14>(row, sp, x, nx, wcc);
    cf_upd<CC + 15>(row, sp, x, nx, wcc);
  }
}

// Cholesky factor and inverse of the 16 x 16 block at src (LDS, pitch CF_BP) by ONE wave: lane l (and its mirrors l + 16 m) is row
// l & 15 of L and column l & 15 of W = L^-1 (csrc/gp_target_fit.hip has the same sweep on v_readlane).
// L (zeros above the diagonal) -> Ld, W -> Wd.  Returns 0, or 1 + the index of the first pivot that is not positive.
#ifndef CF_SWEEP_INLINE
__device__ __attribute__((noinline)) int cf_potf2_16(const cf_lds_double* src, cf_lds_double* Ld, cf_lds_double* Wd, int lane) {
#else
__device__ __forceinline__ int cf_potf2_16(const cf_lds_double* src, cf_lds_double* Ld, cf_lds_double* Wd, int lane) {
#endif
  const int lc = lane & 15;
  double row[16], wv[16], sp[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    row[b] = src[lc * CF_BP + b];
    sp[b] = 0.0;
  }
  int bad = 0;
  double rsq;
  {
    double pv;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(pv) : "v"(row[0]));
    if (!(pv > 0.0)) bad = 1;
    rsq = cf_rsqrt(pv);
  }
  cf_step<0>(row, sp, wv, rsq, bad, lc);
  cf_step<1>(row, sp, wv, rsq, bad, lc);
  cf_step<2>(row, sp, wv, rsq, bad, lc);
  cf_step<3>(row, sp, wv, rsq, bad, lc);
  cf_step<4>(row, sp, wv, rsq, bad, lc);
  cf_step<5>(row, sp, wv, rsq, bad, lc);
  cf_step<6>(row, sp, wv, rsq, bad, lc);
  cf_step<7>(row, sp, wv, rsq, bad, lc);
  cf_step<8>(row, sp, wv, rsq, bad, lc);
  cf_step<9>(row, sp, wv, rsq, bad, lc);
  cf_step<10>(row, sp, wv, rsq, bad, lc);
  cf_step<11>(row, sp, wv, rsq, bad, lc);
  cf_step<12>(row, sp, wv, rsq, bad, lc);
  cf_step<13>(row, sp, wv, rsq, bad, lc);
  cf_step<14>(row, sp, wv, rsq, bad, lc);
  cf_step<15>(row, sp, wv, rsq, bad, lc);
  bad = __builtin_amdgcn_readfirstlane(bad);
  if (lane < 16) {
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      Ld[lc * CF_BP + b] = b <= lc ? row[b] : 0.0;
      Wd[b * CF_BP + lc] = wv[b];
    }
  }
  return bad;
}

// 32 x 32 diagonal block C (LDS, lower triangle valid; clobbered) -> L in Dg, L^-1 in Wb (both full, zeros above the diagonal).
__device__ __forceinline__ int cf_potf2_32(double* C, double* Dg, double* Wb, int lane, double* stamps) {
  const int lc = lane & 15, lq = lane >> 4;
  int bad = cf_potf2_16((const cf_lds_double*)C, (cf_lds_double*)Dg, (cf_lds_double*)Wb, lane);
  if (stamps && lane == 0) stamps[8] = (double)wall_clock64();
  if (bad) return bad;
  // L21 = C21 W11^T
  d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(C[(16 + lc) * CF_BP + 4 * m + lq], Wb[lc * CF_BP + 4 * m + lq], acc, 0, 0, 0);
  cf_settle(acc);
#pragma unroll
  for (int g = 0; g < 4; ++g) Dg[(16 + lq + 4 * g) * CF_BP + lc] = acc[g];
  // S = C22 - L21 L21^T
  d4_t s;
#pragma unroll
  for (int g = 0; g < 4; ++g) s[g] = C[(16 + lq + 4 * g) * CF_BP + 16 + lc];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const double l = Dg[(16 + lc) * CF_BP + 4 * m + lq];
    s = __builtin_amdgcn_mfma_f64_16x16x4f64(-l, l, s, 0, 0, 0);
  }
  cf_settle(s);
#pragma unroll
  for (int g = 0; g < 4; ++g) C[(16 + lq + 4 * g) * CF_BP + 16 + lc] = s[g];
  if (stamps && lane == 0) stamps[9] = (double)wall_clock64();
  bad = cf_potf2_16((const cf_lds_double*)(C + 16 * CF_BP + 16), (cf_lds_double*)(Dg + 16 * CF_BP + 16), (cf_lds_double*)(Wb + 16 * CF_BP + 16), lane);
  if (stamps && lane == 0) stamps[10] = (double)wall_clock64();
  if (bad) return 16 + bad;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    Dg[(lq + 4 * g) * CF_BP + 16 + lc] = 0.0;
    Wb[(lq + 4 * g) * CF_BP + 16 + lc] = 0.0;
  }
  // W21 = -W22 (L21 W11)
  d4_t t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m = 0; m < 4; ++m) t = __builtin_amdgcn_mfma_f64_16x16x4f64(Dg[(16 + lc) * CF_BP + 4 * m + lq], Wb[(4 * m + lq) * CF_BP + lc], t, 0, 0, 0);
  cf_settle(t);
  d4_t w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m = 0; m < 4; ++m) w = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wb[(16 + lc) * CF_BP + 16 + 4 * m + lq], t[m], w, 0, 0, 0);
  cf_settle(w);
#pragma unroll
  for (int g = 0; g < 4; ++g) Wb[(16 + lq + 4 * g) * CF_BP + lc] = w[g];
  return 0;
}

// one 16-wide slice of the operands of a tile product: four consecutive doubles of the lane's row of the A tile and of the B tile
// (contraction index k = 4 lq + m: the A and B sides use the same order, so any order is right)
struct CfOps {
  cf_u4 a0, a1, b0, b1;
};
template <bool SAME>
__device__ __forceinline__ CfOps cf_load(__amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rs, unsigned offA, unsigned offB) {
  CfOps o;
  o.a0 = __builtin_amdgcn_raw_buffer_load_b128(rsA, offA, 0, 16);        // aux 16 = sc1: served by L2, never by this CU's L1
  o.a1 = __builtin_amdgcn_raw_buffer_load_b128(rsA, offA + 16, 0, 16);
  if (SAME) {
    o.b0 = o.a0;
    o.b1 = o.a1;
  } else {
    o.b0 = __builtin_amdgcn_raw_buffer_load_b128(rs, offB, 0, 16);
    o.b1 = __builtin_amdgcn_raw_buffer_load_b128(rs, offB + 16, 0, 16);
  }
  return o;
}
__device__ __forceinline__ d4_t cf_mma(const CfOps& o, d4_t acc) {
  const cf_d2 a0 = __builtin_bit_cast(cf_d2, o.a0), a1 = __builtin_bit_cast(cf_d2, o.a1);
  const cf_d2 b0 = __builtin_bit_cast(cf_d2, o.b0), b1 = __builtin_bit_cast(cf_d2, o.b1);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], b0[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], b0[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], b1[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], b1[1], acc, 0, 0, 0);
  return acc;
}

// acc += sum over the block columns [c_lo, c_hi) (two 16-wide slices each) of A-tile(rows rowA ..) B-tile(rows rowB ..)^T.
// Three operand buffers in rotation, two block columns ahead of the products; no buffer is ever copied (a copy makes the compiler
// wait for the loads it has just issued), and no load is conditional: block columns past c_hi get an offset beyond the descriptor's
// end, which loads zeros without touching memory.
// The A tile comes through its own descriptor: for the forward substitution it is the finished part of v as a one-row matrix (the
// descriptor ends after that row, so the tile's other 15 rows load as zeros) -- v_j is one more row of the factor.
struct CfOps2 {
  CfOps s0, s1;
};
template <bool SAME>
__device__ __forceinline__ CfOps2 cf_load2(__amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rs, unsigned baseA, unsigned baseB, int c, int c_hi) {
  const unsigned off = c < c_hi ? 256u * (unsigned)c : 0xC0000000u;
  CfOps2 o;
  o.s0 = cf_load<SAME>(rsA, rs, baseA + off, baseB + off);
  o.s1 = cf_load<SAME>(rsA, rs, baseA + off + 128u, baseB + off + 128u);
  return o;
}
__device__ __forceinline__ d4_t cf_mma2(const CfOps2& o, d4_t acc) { return cf_mma(o.s1, cf_mma(o.s0, acc)); }

template <bool SAME>
__device__ __forceinline__ d4_t cf_accumulate(d4_t acc, __amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rs, int N, int rowA, int rowB, int c_lo, int c_hi,
                                              int lc, int lq) {
  const unsigned baseA = (unsigned)(((size_t)(rowA + lc) * N + 4 * lq) * 8), baseB = (unsigned)(((size_t)(rowB + lc) * N + 4 * lq) * 8);
  CfOps2 b0 = cf_load2<SAME>(rsA, rs, baseA, baseB, c_lo, c_hi), b1 = cf_load2<SAME>(rsA, rs, baseA, baseB, c_lo + 1, c_hi), b2;
  for (int c = c_lo; c < c_hi; c += 3) {
    b2 = cf_load2<SAME>(rsA, rs, baseA, baseB, c + 2, c_hi);
    acc = cf_mma2(b0, acc);
    b0 = cf_load2<SAME>(rsA, rs, baseA, baseB, c + 3, c_hi);
    if (c + 1 < c_hi) acc = cf_mma2(b1, acc);
    b1 = cf_load2<SAME>(rsA, rs, baseA, baseB, c + 4, c_hi);
    if (c + 2 < c_hi) acc = cf_mma2(b2, acc);
  }
  // (settled here, not at the use: the registers of one slot's tile are reused for the next slot's, and a VALU write into the last
  //  result pair 17 wait states behind the MFMA is as unprotected as a read -- found by the build's hazard audit)
  cf_settle(acc);
  return acc;
}

// The forward substitution's products (L_j,0:j v for the 32 rows of block row j) in ONE pipeline: the finished part of v as a one-row A tile
// (its descriptor ends after that row: the other 15 rows load as zeros), the two 16-row tiles of the block row as B.
struct CfOpsV {
  cf_u4 v0, v1, p0, p1, q0, q1;
};
__device__ __forceinline__ CfOpsV cf_loadv(__amdgpu_buffer_rsrc_t rsV, __amdgpu_buffer_rsrc_t rs, unsigned baseV, unsigned baseB, unsigned rowskip, int s, int s_hi) {
  const unsigned off = s < s_hi ? 128u * (unsigned)s : 0xC0000000u;
  CfOpsV o;
  o.v0 = __builtin_amdgcn_raw_buffer_load_b128(rsV, baseV + off, 0, 16);
  o.v1 = __builtin_amdgcn_raw_buffer_load_b128(rsV, baseV + off + 16, 0, 16);
  o.p0 = __builtin_amdgcn_raw_buffer_load_b128(rs, baseB + off, 0, 16);
  o.p1 = __builtin_amdgcn_raw_buffer_load_b128(rs, baseB + off + 16, 0, 16);
  o.q0 = __builtin_amdgcn_raw_buffer_load_b128(rs, baseB + rowskip + off, 0, 16);
  o.q1 = __builtin_amdgcn_raw_buffer_load_b128(rs, baseB + rowskip + off + 16, 0, 16);
  return o;
}
__device__ __forceinline__ void cf_mmav(const CfOpsV& o, d4_t& r0, d4_t& r1) {
  r0 = cf_mma(CfOps{o.v0, o.v1, o.p0, o.p1}, r0);
  r1 = cf_mma(CfOps{o.v0, o.v1, o.q0, o.q1}, r1);
}
__device__ __forceinline__ void cf_accumulate_v(d4_t& r0, d4_t& r1, __amdgpu_buffer_rsrc_t rsV, __amdgpu_buffer_rsrc_t rs, int N, int rowB, int c_lo, int c_hi, int lc,
                                                int lq) {
  const unsigned baseV = (unsigned)(((size_t)lc * N + 4 * lq) * 8), baseB = (unsigned)(((size_t)(rowB + lc) * N + 4 * lq) * 8), rowskip = (unsigned)(16 * N * 8);
  const int s_hi = 2 * c_hi;
  CfOpsV b0 = cf_loadv(rsV, rs, baseV, baseB, rowskip, 2 * c_lo, s_hi), b1 = cf_loadv(rsV, rs, baseV, baseB, rowskip, 2 * c_lo + 1, s_hi), b2;
  for (int s = 2 * c_lo; s < s_hi; s += 3) {
    b2 = cf_loadv(rsV, rs, baseV, baseB, rowskip, s + 2, s_hi);
    cf_mmav(b0, r0, r1);
    b0 = cf_loadv(rsV, rs, baseV, baseB, rowskip, s + 3, s_hi);
    if (s + 1 < s_hi) cf_mmav(b1, r0, r1);
    b1 = cf_loadv(rsV, rs, baseV, baseB, rowskip, s + 4, s_hi);
    if (s + 2 < s_hi) cf_mmav(b2, r0, r1);
  }
  cf_settle(r0);
  cf_settle(r1);
}

// Two tiles with the same B rows (the same tile position in two consecutive block rows) in ONE pipeline: B is loaded once, and the two
// tiles share the pipeline's start-up round trip (an even column's first chunk has two block rows to finish while the diagonal block is
// being factored; one after the other they do not fit into that time).
struct CfOpsA2 {
  cf_u4 a0, a1, c0, c1, b0, b1;
};
__device__ __forceinline__ CfOpsA2 cf_loada2(__amdgpu_buffer_rsrc_t rs, unsigned baseA, unsigned baseC, unsigned baseB, int s, int s_hi) {
  const unsigned off = s < s_hi ? 128u * (unsigned)s : 0xC0000000u;
  CfOpsA2 o;
  o.a0 = __builtin_amdgcn_raw_buffer_load_b128(rs, baseA + off, 0, 16);
  o.a1 = __builtin_amdgcn_raw_buffer_load_b128(rs, baseA + off + 16, 0, 16);
  o.c0 = __builtin_amdgcn_raw_buffer_load_b128(rs, baseC + off, 0, 16);
  o.c1 = __builtin_amdgcn_raw_buffer_load_b128(rs, baseC + off + 16, 0, 16);
  o.b0 = __builtin_amdgcn_raw_buffer_load_b128(rs, baseB + off, 0, 16);
  o.b1 = __builtin_amdgcn_raw_buffer_load_b128(rs, baseB + off + 16, 0, 16);
  return o;
}
__device__ __forceinline__ void cf_mmaa2(const CfOpsA2& o, d4_t& r0, d4_t& r1) {
  r0 = cf_mma(CfOps{o.a0, o.a1, o.b0, o.b1}, r0);
  r1 = cf_mma(CfOps{o.c0, o.c1, o.b0, o.b1}, r1);
}
__device__ __forceinline__ void cf_accumulate_a2(d4_t& r0, d4_t& r1, __amdgpu_buffer_rsrc_t rs, int N, int rowA, int rowC, int rowB, int c_lo, int c_hi, int lc, int lq) {
  const unsigned baseA = (unsigned)(((size_t)(rowA + lc) * N + 4 * lq) * 8), baseC = (unsigned)(((size_t)(rowC + lc) * N + 4 * lq) * 8);
  const unsigned baseB = (unsigned)(((size_t)(rowB + lc) * N + 4 * lq) * 8);
  const int s_hi = 2 * c_hi;
  CfOpsA2 b0 = cf_loada2(rs, baseA, baseC, baseB, 2 * c_lo, s_hi), b1 = cf_loada2(rs, baseA, baseC, baseB, 2 * c_lo + 1, s_hi), b2;
  for (int s = 2 * c_lo; s < s_hi; s += 3) {
    b2 = cf_loada2(rs, baseA, baseC, baseB, s + 2, s_hi);
    cf_mmaa2(b0, r0, r1);
    b0 = cf_loada2(rs, baseA, baseC, baseB, s + 3, s_hi);
    if (s + 1 < s_hi) cf_mmaa2(b1, r0, r1);
    b1 = cf_loada2(rs, baseA, baseC, baseB, s + 4, s_hi);
    if (s + 2 < s_hi) cf_mmaa2(b2, r0, r1);
  }
  cf_settle(r0);
  cf_settle(r1);
}

template <int KIND>
__global__ __launch_bounds__(512) void gp_fit_coop_kernel(CoopFitParams p) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int N = p.N, D = p.D, P = p.parts;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int task = (idx / P) * 8 + xcd, part = idx % P;
  if (task >= p.T) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, lq = lane >> 4;
  const int NB = (N + 31) >> 5;          // block columns (the last one is half empty when N / 16 is odd: rows / columns >= N are identity)
  int n = p.n_points ? p.n_points[task] : N;
  n = n < 0 ? 0 : (n > N ? N : n);
  const int DP = D | 1;

  double* exptab = lds;                  // [64]
  double* invl = exptab + 64;            // [16]
  double* tv = invl + 16;                // [32] right-hand side of the column's forward substitution
  double* rvs = tv + 32;                 // [32] L_j,0:j v
  double* red = rvs + 32;                // [16]
  int* ctrl = (int*)(red + 16);          // [0] abandon the attempt, [1] failing pivot (local), [2] all parts of the task on one XCD
  double* Cb = red + 16 + 8;             // [4][32][CF_BP] the chunk's blocks before the multiplication by L_jj^-T
  double* Dg = Cb + 4 * CF_BS;           // L_jj
  double* Wb = Dg + CF_BS;               // L_jj^-1
  double* Xs = Wb + CF_BS;               // [N][DP] x / lengthscale

  const double* Xg = p.X + (size_t)task * N * D;
  const double* yg = p.y + (size_t)task * N;
  const double* th = p.theta + (size_t)task * (D + 2);
  double* Lg = p.L + (size_t)task * N * N;
  double* Wg = p.Linv_diag + (size_t)task * (N / 16) * 256;
  double* vg = p.v + (size_t)task * N;
  unsigned* prog = p.prog + (size_t)task * 32;
  unsigned* status = p.status + (size_t)task * 4;
  const double os = th[D], noise = th[D + 1];
  const double base_jit = p.jitter_in ? p.jitter_in[task] : 0.0;
  const int max_attempts = (p.flags & SCAML_FIT_NO_RETRY) ? 1 : 4;

  exp2_table_init(exptab, tid);
  if (tid < 16) invl[tid] = tid < D ? 1.0 / th[tid] : 0.0;
  if (tid == 0) { ctrl[0] = 0; ctrl[1] = 0; }
  __syncthreads();
  for (int e = tid; e < N * D; e += blockDim.x) {
    const int r = e / D, d = e - r * D;
    Xs[r * DP + d] = r < n ? Xg[e] * invl[d] : 0.0;
  }
  __syncthreads();

  // XCD census: do all P workgroups of this task share an XCD (and with it an L2)?  The host's block map puts them on one (blockIdx & 7
  // is the same), but placement is the dispatcher's business: every part publishes the XCC id it actually runs on, and only if all P agree
  // are payload stores plain.  Results do not depend on the answer, only the speed of the hand-offs does.
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 15u;   // HW_REG_XCC_ID, bits [3:0]
    __hip_atomic_store(p.xcc + (size_t)task * 8 + part, xcc + 1u, CF_RLX_AGENT);
  }
  if (wave == 0) {
    bool same = false;
    for (int spins = 0; spins < CF_SPIN_LIMIT; ++spins) {
      const unsigned mine = __hip_atomic_load(p.xcc + (size_t)task * 8 + part, CF_RLX_AGENT);
      const unsigned other = lane < P ? __hip_atomic_load(p.xcc + (size_t)task * 8 + lane, CF_RLX_AGENT) : mine;
      if (__ballot(other == 0u || mine == 0u) == 0ull) {
        same = __ballot(other != mine) == 0ull;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (lane == 0) ctrl[2] = (same && !(p.flags & 0x80000000u)) ? 1 : 0;
  }
  __syncthreads();
  const bool near = ctrl[2] != 0;

  // descriptors of the two handed-off arrays this workgroup loads from (wave-uniform by construction)
  // (rows >= n are an identity block that is never written -- include/scaml_gp.h --: the descriptor ends at row n, loads past it return 0)
  const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(Lg, 0, n * N * 8, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(vg, 0, N * 8, 0x00020000);
  const double kc0 = os, kc1 = 2.2360679774997896964 * os, kc2 = (5.0 / 3.0) * os;
  const int wb = wave >> 2, wq = wave & 3, ti = wq >> 1, tj = wq & 1;   // this wave's tile inside a chunk of two blocks
  const int last_part = (NB - 1) % P;

  for (int attempt = 1;; ++attempt) {
    const double ladder = attempt == 1 ? 0.0 : (attempt == 2 ? 1e-8 : (attempt == 3 ? 1e-7 : 1e-6));
    const double diag_add = noise + base_jit + ladder;
    bool gone = false;
    // kernel values of this wave's tile of block (i, j)
    auto kvals = [&](int j, int i) -> d4_t {
      d4_t kt = {0.0, 0.0, 0.0, 0.0};
      const int col = 32 * j + 16 * tj + lc;
      const double* xc = Xs + (col < N ? col : 0) * DP;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = 32 * i + 16 * ti + lq + 4 * g;
        const double* xr = Xs + (row < N ? row : 0) * DP;
        double d2 = 0.0;
        for (int d = 0; d < D; ++d) {
          const double u = xr[d] - xc[d];
          d2 = __builtin_fma(u, u, d2);
        }
        double kv = kernel_from_sqdist_scaled<KIND>(row == col ? 0.0 : d2, kc0, kc1, kc2, exptab);
        if (row == col) kv += diag_add;
        if (row >= n || col >= n) kv = row == col ? 1.0 : 0.0;
        kt[g] = kv;
      }
      return kt;
    };
    // Left-looking sums of up to three tiles of this wave in column j, advanced together: slot 0 is the tile of the chunk being worked
    // on, slots 1 and 2 the tiles of the next two phases (the block row j + 2 that is summed while the diagonal block is factored, and
    // the first chunk behind it).  Their sums over the columns c <= j - 2 do not depend on the column the workgroup is waiting for,
    // so they are taken in the waiting time -- one bounded piece between two polls, so that the awaited tile is never kept waiting for
    // long; summed on demand instead, those tiles trail the diagonal chain by more with every column.
    // mode 0: tile (ti, tj) of the block row; 1: the same with A and B the same rows (diagonal tiles); 2: no tile, L_j,0:j v (-> acc, rv1);
    // 3: the tiles (ti, tj) of the block rows `row` and `row + 1` together (-> acc, rv1).
    struct Slot {
      d4_t acc;
      int row, mode, done;
      bool on;
    };
    d4_t rv1 = {0.0, 0.0, 0.0, 0.0};   // second accumulator of the mode-2 slot (rows 16 .. 31 of L_j,0:j v)
    auto advance = [&](int j, Slot& s0, Slot& s1, Slot& s2, Slot& s3, int target) {
      // until slot `target` (0 .. 3) has all j columns; the slots behind it take what is ready, two columns at a time
      int spins = 0;
      for (;;) {
        Slot& st_ = target == 0 ? s0 : (target == 1 ? s1 : (target == 2 ? s2 : s3));
        if (!st_.on || st_.done >= j) return;
        unsigned pv = 0xffffffffu;
        if (lane < j) pv = __hip_atomic_load(prog + lane, CF_RLX_AGENT);
        const unsigned st = __hip_atomic_load(status, CF_RLX_AGENT);
        if (st >= (unsigned)attempt) { gone = true; break; }
        const bool tag_ok = (pv >> 8) == (unsigned)attempt;
        const int cnt = (int)(pv & 255u);
        asm volatile("" ::: "memory");   // (the payload loads below stay below the poll)
        bool moved = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          Slot& sl = k == 0 ? s0 : (k == 1 ? s1 : (k == 2 ? s2 : s3));
          if (k < target || !sl.on || sl.done >= j || moved) continue;
          const unsigned long long m = __ballot(lane >= j || (tag_ok && cnt >= sl.row + (sl.mode == 3 ? 1 : 0) - lane + 1));
          int ready = m == ~0ull ? j : __builtin_ctzll(~m);
          if (ready > j) ready = j;
          if (ready > sl.done) {
            if (k != target && ready > sl.done + 2) ready = sl.done + 2;
            if (sl.mode == 2) {
              cf_accumulate_v(sl.acc, rv1, rsV, rsL, N, 32 * j, sl.done, ready, lc, lq);
            } else if (sl.mode == 3) {
              cf_accumulate_a2(sl.acc, rv1, rsL, N, 32 * sl.row + 16 * ti, 32 * (sl.row + 1) + 16 * ti, 32 * j + 16 * tj, sl.done, ready, lc, lq);
            } else if (sl.mode == 1) {
              sl.acc = cf_accumulate<true>(sl.acc, rsL, rsL, N, 32 * sl.row + 16 * ti, 32 * j + 16 * tj, sl.done, ready, lc, lq);
            } else {
              sl.acc = cf_accumulate<false>(sl.acc, rsL, rsL, N, 32 * sl.row + 16 * ti, 32 * j + 16 * tj, sl.done, ready, lc, lq);
            }
            sl.done = ready;
            moved = true;
          }
        }
        if (moved) {
          spins = 0;
        } else {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > CF_SPIN_LIMIT) {
            if (lane == 0) __hip_atomic_fetch_max(status, CF_DEAD, CF_RLX_AGENT);
            gone = true;
            break;
          }
        }
      }
      if (gone && lane == 0) ctrl[0] = 1;
    };
    // C = K - sums of a finished slot into the chunk buffer (or, mode 2, L_j,0:j v into rvs)
    auto put = [&](Slot& sl, d4_t kt, double* Cdst) {
      cf_settle(sl.acc);
      if (sl.mode == 2) {
        cf_settle(rv1);
        if (lq == 0) {   // row 0 of the two products: (L_j,0:j v)[16 tj + lc]
          rvs[lc] = sl.acc[0];
          rvs[16 + lc] = rv1[0];
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) Cdst[(16 * ti + lq + 4 * g) * CF_BP + 16 * tj + lc] = kt[g] - sl.acc[g];
      }
    };
    // one tile summed on demand (the chunks further down the column)
    auto sum_tile = [&](int j, int i, double* Cdst) {
      const d4_t kt = kvals(j, i);
      Slot s{{0.0, 0.0, 0.0, 0.0}, i, 0, 0, true}, off{{0.0, 0.0, 0.0, 0.0}, 0, 0, 0, false};
      advance(j, s, off, off, off, 0);
      if (!gone) put(s, kt, Cdst);
    };
    // tile (ti, tj) of L_ij = C_ij L_jj^-T on the matrix cores, written through (sc1): rows / columns past n are never written
    auto trsm_store = [&](int j, int i, const double* Cs) {
      d4_t out = {0.0, 0.0, 0.0, 0.0};
      for (int tk = 0; tk <= tj; ++tk) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          out = __builtin_amdgcn_mfma_f64_16x16x4f64(Cs[(16 * ti + lc) * CF_BP + 16 * tk + 4 * m + lq], Wb[(16 * tj + lc) * CF_BP + 16 * tk + 4 * m + lq], out, 0, 0, 0);
      }
      cf_settle(out);
      const int col = 32 * j + 16 * tj + lc;
      if (col < n) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 32 * i + 16 * ti + lq + 4 * g;
          if (row < n) cf_store(Lg + (size_t)row * N + col, out[g], near);
        }
      }
    };

    for (int j = part; j < NB && !gone; j += P) {
      if (p.flags & SCAML_FIT_ZERO_UPPER) {   // the strict upper triangle above this block column (nobody reads it)
        const int rows = 32 * j, c0 = 32 * j;
        for (int e = tid; e < rows * 32; e += blockDim.x) {
          const int r = e >> 5, c = c0 + (e & 31);
          if (c < n) Lg[(size_t)r * N + c] = 0.0;
        }
      }
      CF_STAMP(0);
      // ---- first chunk: the diagonal block and the TWO block rows below it (what the next column's first chunk starts from) ----
      // Block rows of the first chunk: the diagonal block and the two below it -- three in an odd column, FOUR in an even one, so that the
      // chunks behind start at an even block row in EVERY column: a chunk {2 m, 2 m + 1} then needs the same chunk of the column before
      // it and nothing else (with cuts relative to j it needs two consecutive chunks of that column, two chunk times per column, and
      // trails the diagonal chain by more with every column).
      const int want0 = (j & 1) ? 3 : 4;
      const int nb0 = NB - j < want0 ? NB - j : want0;
      const d4_t z4 = {0.0, 0.0, 0.0, 0.0};
      // slot 0: diagonal block (tiles (0,0), (1,0), (1,1); the fourth wave: L_j,0:j v) / block row j + 1; slot 1: block row j + 2 (summed
      // by waves 4-7 while wave 0 factors the diagonal block); in an even column together with block row j + 3 (mode 3);
      // slot 2 (odd columns): this wave's tile of the first chunk behind (block rows j + 3, j + 4)
      const bool four = nb0 > 3;
      Slot s0{z4, j + wb, wb == 0 ? (wq == 1 ? 2 : (wq == 2 ? 0 : 1)) : 0, 0, wb == 0 || nb0 > 1};
      Slot s1{z4, j + 2, four ? 3 : 0, 0, wb == 1 && nb0 > 2};
      Slot s2{z4, j + 3 + wb, 0, 0, (j & 1) && j + 3 + wb < NB};
      // slot 3: this wave's tile of the next chunk behind those (block rows j + 5, j + 6 in an odd column, j + 4, j + 5 in an even one)
      const int row3 = ((j & 1) ? j + 5 : j + 4) + wb;
      Slot s3{z4, row3, 0, 0, row3 < NB};
      rv1 = z4;
      const d4_t kt0 = (s0.on && s0.mode != 2) ? kvals(j, s0.row) : z4;
      const d4_t kt2 = s2.on ? kvals(j, s2.row) : z4;
      advance(j, s0, s1, s2, s3, 0);
      if (!gone && s0.on) put(s0, kt0, Cb + wb * CF_BS);
      CF_STAMP(2);
      __syncthreads();
      gone = ctrl[0] != 0;
      if (gone) break;
      CF_STAMP(3);
      if (wave == 0) {
#ifdef CF_STAMPS
        double* stp = task == 0 ? p.part + (size_t)p.T * 64 + ((size_t)part * 32 + (j / P)) * 16 : nullptr;
#else
        double* stp = nullptr;
#endif
        const int bad = cf_potf2_32(Cb, Dg, Wb, lane, stp);
        CF_STAMP(4);
        if (bad && lane == 0) {
          const unsigned piv = (unsigned)(32 * j + bad);
          __hip_atomic_fetch_max(status + 1, ((unsigned)attempt << 20) | piv, CF_RLX_AGENT);
          cf_drain();
          __hip_atomic_fetch_max(status, (unsigned)attempt, CF_RLX_AGENT);
          ctrl[0] = 1;
          ctrl[1] = (int)piv;
        }
      } else if (wave == 1) {
        if (lane < 32) {
          const int r = 32 * j + lane;
          tv[lane] = (r < n ? yg[r] : 0.0) - (j > 0 ? rvs[lane] : 0.0);
        }
      } else if (wb == 1) {
        if (s1.on) {                             // (block row j + 2, during the factorisation of the diagonal block)
          const d4_t kt1 = kvals(j, s1.row);
          advance(j, s0, s1, s2, s3, 1);
          if (!gone) {
            if (four) {
              const d4_t kt1b = kvals(j, s1.row + 1);
              cf_settle(rv1);
#pragma unroll
              for (int g = 0; g < 4; ++g) Cb[3 * CF_BS + (16 * ti + lq + 4 * g) * CF_BP + 16 * tj + lc] = kt1b[g] - rv1[g];
            }
            put(s1, kt1, Cb + 2 * CF_BS);
          }
        }
      }
      __syncthreads();
      gone = ctrl[0] != 0;
      if (gone) break;
      CF_STAMP(5);
      if (wave == 1) {
        // v_j = L_jj^-1 (y_j - L_j,0:j v): row r = lane & 31, the two half-waves take the columns k < 16 / k >= 16
        const int r = lane & 31, h = lane >> 5;
        double vj = 0.0;
        for (int k = 16 * h; k < 16 * h + 16; ++k) vj = __builtin_fma(k <= r ? Wb[r * CF_BP + k] : 0.0, tv[k], vj);
        vj += __shfl_xor(vj, 32);
        if (lane < 32 && 32 * j + lane < N) cf_store(vg + 32 * j + lane, vj, near);
      } else if (wave >= 2 && wave < 4) {
        // the inverses of the two 16 x 16 diagonal blocks (what the posterior kernels take)
        const int h = wave - 2, blk = 2 * j + h;
        if (16 * blk < N)
          for (int e = lane; e < 256; e += 64) cf_store(Wg + (size_t)blk * 256 + e, Wb[(16 * h + (e >> 4)) * CF_BP + 16 * h + (e & 15)], near);
      }
      if (wb == 0) {
        // the diagonal block's tiles go out as they are (the upper tile only on request) ...
        const int col = 32 * j + 16 * tj + lc;
        const bool upper = ti == 0 && tj == 1;
        if (col < n && (!upper || (p.flags & SCAML_FIT_ZERO_UPPER))) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = 32 * j + 16 * ti + lq + 4 * g;
            if (row < n) cf_store(Lg + (size_t)row * N + col, Dg[(16 * ti + lq + 4 * g) * CF_BP + 16 * tj + lc], near);
          }
        }
        if (nb0 > 1) trsm_store(j, j + 1, Cb + CF_BS);   // ... and the first block row below
      } else if (nb0 > 2) {
        trsm_store(j, j + 2, Cb + 2 * CF_BS);
        if (four) trsm_store(j, j + 3, Cb + 3 * CF_BS);
      }
      CF_STAMP(6);
      cf_drain();
      __syncthreads();
      CF_STAMP(7);
      if (tid == 0) __hip_atomic_store(prog + j, ((unsigned)attempt << 8) | (unsigned)nb0, CF_RLX_AGENT);
      // ---- the rest of the column, two block rows at a time ----
      for (int i0 = j + nb0; i0 < NB;) {
        const int nblk = NB - i0 < 2 ? NB - i0 : 2;
        if (i0 == j + 3) {   // (odd column: the chunk whose sums were started in the waiting time)
          if (s2.on) {
            advance(j, s0, s1, s2, s3, 2);
            if (!gone) put(s2, kt2, Cb + wb * CF_BS);
          }
        } else if (i0 + wb == row3) {
          if (s3.on) {
            const d4_t kt3 = kvals(j, row3);
            advance(j, s0, s1, s2, s3, 3);
            if (!gone) put(s3, kt3, Cb + wb * CF_BS);
          }
        } else if (wb < nblk) {
          sum_tile(j, i0 + wb, Cb + wb * CF_BS);
        }
        __syncthreads();
        gone = ctrl[0] != 0;
        if (gone) break;
        if (wb < nblk) trsm_store(j, i0 + wb, Cb + wb * CF_BS);
        cf_drain();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(prog + j, ((unsigned)attempt << 8) | (unsigned)(i0 - j + nblk), CF_RLX_AGENT);
        i0 += nblk;
      }
    }
    { const int j = part + 31 * P; CF_STAMP(0); }   // (slot 31: all columns of this part done)
    // ---- the attempt's outcome ----
    if (!gone && part != last_part) {
      // this part's columns are through: wait until the last column is, or until somebody fails
      if (wave == 0) {
        int spins = 0;
        for (;;) {
          const unsigned pv = __hip_atomic_load(prog + NB - 1, CF_RLX_AGENT);
          const unsigned st = __hip_atomic_load(status, CF_RLX_AGENT);
          if (st >= (unsigned)attempt) { if (lane == 0) ctrl[0] = 1; break; }
          if ((pv >> 8) == (unsigned)attempt && (pv & 255u) >= 1u) break;
          __builtin_amdgcn_s_sleep(8);
          if (++spins > CF_SPIN_LIMIT) {
            if (lane == 0) { __hip_atomic_fetch_max(status, CF_DEAD, CF_RLX_AGENT); ctrl[0] = 1; }
            break;
          }
        }
      }
      __syncthreads();
      gone = ctrl[0] != 0;
      if (!gone) return;   // factored: the owner of the last column finishes the task
    }
    if (gone) {
      const unsigned st = __hip_atomic_load(status, CF_RLX_AGENT);
      const bool fatal = st >= CF_DEAD || attempt >= max_attempts;
      if (fatal) {
        // the workgroup that saw the pivot fail in the last attempt (or that timed out) reports; the others just leave
        if (tid == 0 && (ctrl[1] != 0 || st >= CF_DEAD)) {
          const double nan = __builtin_nan("");
          p.info[task] = st >= CF_DEAD ? -1 : ctrl[1];
          if (p.jitter_used) p.jitter_used[task] = ladder;
          if (p.quad) p.quad[task] = nan;
          if (p.logdet) p.logdet[task] = nan;
          if (p.mll) p.mll[task] = nan;
        }
        return;
      }
      __syncthreads();
      if (tid == 0) { ctrl[0] = 0; ctrl[1] = 0; }
      __syncthreads();
      continue;
    }
    // ---- owner of the last column, everything factored: scalars and alpha = L^-T v ----
    {
      double* vs = Cb;   // [N] (N <= 2 * CF_BS)
      for (int r = tid; r < N; r += blockDim.x) vs[r] = __hip_atomic_load(vg + r, CF_RLX_AGENT);
      {
        // quad = v . v, logdet = 2 sum log L_rr: one element per thread, summed in a fixed order (waves, then lanes)
        double q = 0.0, ld = 0.0;
        for (int r = tid; r < N; r += blockDim.x) {
          const double vr = __hip_atomic_load(vg + r, CF_RLX_AGENT);
          q += vr * vr;
          if (r < n) ld += 2.0 * log(__hip_atomic_load(Lg + (size_t)r * N + r, CF_RLX_AGENT));
        }
        for (int o = 32; o > 0; o >>= 1) {
          q += __shfl_xor(q, o);
          ld += __shfl_xor(ld, o);
        }
        if (lane == 0) {
          rvs[wave] = q;
          rvs[8 + wave] = ld;
        }
        __syncthreads();
        if (tid == 0) {
          q = 0.0;
          ld = 0.0;
          for (int w = 0; w < 8; ++w) {
            q += rvs[w];
            ld += rvs[8 + w];
          }
          p.info[task] = 0;
          if (p.jitter_used) p.jitter_used[task] = ladder;
          if (p.quad) p.quad[task] = q;
          if (p.logdet) p.logdet[task] = ld;
          if (p.mll) p.mll[task] = n > 0 ? -0.5 * (q + ld + n * 1.8378770664093454836) / n : 0.0;
        }
        __syncthreads();
      }
      // row-push form by 16-row blocks, last to first: alpha_b = W_b^T v_b, then v_k -= sum_r L[16 b + r][k] alpha_b[r] for k < 16 b
      // (the rows of L are contiguous: thread k reads column k of the block row, coalesced; the next block row is fetched while this
      //  one is used)
      CF_FSTAMP(0);
      const int nb16 = N / 16;
      // Row-push form by 16-row blocks, last to first: alpha_b = W_b^T v_b, then v_k -= sum_r L[16 b + r][k] alpha_b[r] for k < 16 b.
      // Thread t < 256 owns the columns 2 t, 2 t + 1 of the pushes (N <= 512).  The 16 rows of L of a step come as 16-byte sc1 buffer
      // loads CF_FD steps ahead (they depend on nothing computed here; a step is ~150 ns of arithmetic, a load ~1 us away); all the W_b
      // are fetched into LDS up front (64 KB: the chunk buffers and the staged points are free by now).  No load is conditional (a
      // conditional load is a branch, and the compiler waits for every load in flight at its join): what a thread does not need gets
      // an offset beyond the descriptor's end and loads as zero.  ONE barrier per step: the wave whose threads own the columns of block
      // b - 1 forms alpha_{b-1} right behind its own push of step b (its v_{b-1} is final then), into the other of two alpha buffers.
      double* Wall = Cb;                        // [nb16][256]
      double* vs2 = Wall + (size_t)nb16 * 256;  // [N]: v moves behind the W blocks
      double* abuf = tv;                        // [2][16]
      constexpr int CF_FD = 3;
      cf_d2 lr[CF_FD][16];
      const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(Wg, 0, nb16 * 256 * 8, 0x00020000);
      const bool loader = tid < 256;
      auto fetch = [&](int b, cf_d2* l) {
        const unsigned base = (b >= 0 && 2 * tid < 16 * b) ? (unsigned)(((size_t)(16 * b) * N + 2 * tid) * 8) : 0xC0000000u;
#pragma unroll
        for (int r = 0; r < 16; ++r) l[r] = __builtin_bit_cast(cf_d2, __builtin_amdgcn_raw_buffer_load_b128(rsL, base + (unsigned)(r * N * 8), 0, 16));
      };
      if (loader) {
#pragma unroll
        for (int u = 0; u < CF_FD; ++u) fetch(nb16 - 1 - u, lr[u]);
      }
      {
        // v out of the chunk buffer (the W blocks go there), then every W block
        double vkeep = tid < N ? vs[tid] : 0.0;
        __syncthreads();
        if (tid < N) vs2[tid] = vkeep;
        for (int e = tid; e < nb16 * 128; e += blockDim.x) {
          const cf_d2 w2 = __builtin_bit_cast(cf_d2, __builtin_amdgcn_raw_buffer_load_b128(rsW, (unsigned)(e * 16), 0, 16));
          Wall[2 * e] = w2[0];
          Wall[2 * e + 1] = w2[1];
        }
      }
      cf_lds_barrier();
      if (tid < 16) {
        const int b = nb16 - 1;
        double a = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) a = __builtin_fma(Wall[b * 256 + r * 16 + tid], vs2[16 * b + r], a);   // (W_b^T v_b)[tid]
        abuf[(b & 1) * 16 + tid] = a;
        if (16 * b + tid < n) p.alpha[(size_t)task * N + 16 * b + tid] = a;
      }
      cf_lds_barrier();
      CF_FSTAMP(1);
      for (int b0 = nb16 - 1; b0 >= 0; b0 -= CF_FD) {
        if (((nb16 - 1 - b0) / CF_FD) < 8) CF_FSTAMP(2 + ((nb16 - 1 - b0) / CF_FD));
#pragma unroll
        for (int u = 0; u < CF_FD; ++u) {
          const int b = b0 - u;
          if (b >= 0) {
            if (loader) {
              if (2 * tid < 16 * b) {
                const double* ab = abuf + (b & 1) * 16;
                double sa = vs2[2 * tid], sb = vs2[2 * tid + 1];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  const double ar = ab[r];
                  sa = __builtin_fma(-lr[u][r][0], ar, sa);
                  sb = __builtin_fma(-lr[u][r][1], ar, sb);
                }
                vs2[2 * tid] = sa;
                vs2[2 * tid + 1] = sb;
              }
              fetch(b - CF_FD, lr[u]);
              if (b > 0 && wave == ((b - 1) >> 3) && lane < 16) {
                // (same wave as the eight threads that have just written v_{b-1}: LDS operations of a wave execute in order)
                double a = 0.0;
#pragma unroll
                for (int r = 0; r < 16; ++r) a = __builtin_fma(Wall[(b - 1) * 256 + r * 16 + lane], vs2[16 * (b - 1) + r], a);
                abuf[((b - 1) & 1) * 16 + lane] = a;
                if (16 * (b - 1) + lane < n) p.alpha[(size_t)task * N + 16 * (b - 1) + lane] = a;
              }
            }
            cf_lds_barrier();
          }
        }
      }
      { const int j = part + 31 * P; CF_STAMP(1); }   // (finish done)
    }
    return;
  }
}

template __global__ void gp_fit_coop_kernel<0>(CoopFitParams);
template __global__ void gp_fit_coop_kernel<1>(CoopFitParams);

}  // namespace scaml

// Shared device helpers for the gfx950 GP kernels (fp64 on CDNA4).
//
// Measured on MI355X (tools/mfma_f64_probe.hip, profiles/r01_probe.txt): a v_fma_f64 costs a
// single wave 8 issue cycles, v_rsq_f64 / v_rcp_f64 ~180, so sqrt()/1.0/x (320/256 cycles) are
// replaced by an f32 seed (v_rsq_f32) plus one cubically-convergent fp64 step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4_t __attribute__((ext_vector_type(4)));

// The wave's index inside its workgroup as a SCALAR: tid >> 6 is the same in all 64 lanes, but the compiler cannot know, and every
// block-row / strip index derived from it is then vector arithmetic -- on a SIMD that its waves' VALU instructions already contend
// for (profiles/r03_probe_valu_overlap.txt).  Through readfirstlane that arithmetic moves to the scalar unit.
#ifdef SCAML_VECTOR_WAVE   // (A/B: the plain expression)
#define SCAML_WAVE_INDEX(tid) ((tid) >> 6)
#else
#define SCAML_WAVE_INDEX(tid) __builtin_amdgcn_readfirstlane((tid) >> 6)
#endif

namespace scaml {

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readfirstlane_f64(double v) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// 1/sqrt(d) for d in [1e-30, 1e30]: f32 seed (rel. err ~2^-22) + one Halley step
// y <- y (1 + e/2 + 3e^2/8), e = 1 - d y^2   => rel. err ~ (5/16) e^3 < 1e-19 before rounding.
__device__ __forceinline__ double rsqrt_seeded(double d) {
  float yf = __builtin_amdgcn_rsqf((float)d);
  double y = (double)yf;
  double t = d * y;
  double e = __builtin_fma(-t, y, 1.0);
  double p = __builtin_fma(e, 0.375, 0.5);
  double ye = y * e;
  return __builtin_fma(ye, p, y);
}

// Robust 1/sqrt(d), d > 0: seeded fast path in the f32-safe range, libm otherwise.
__device__ __forceinline__ double rsqrt_pos(double d) {
  if (d >= 1e-30 && d <= 1e30) return rsqrt_seeded(d);
  return 1.0 / sqrt(d);
}

// sqrt(d) from a given rinv ~ 1/sqrt(d): one correction step makes it correctly rounded
// in all but a vanishing fraction of cases.
__device__ __forceinline__ double sqrt_from_rinv(double d, double rinv) {
  double g = d * rinv;
  double r = __builtin_fma(-g, g, d);
  return __builtin_fma(r, 0.5 * rinv, g);
}

// exp(x) for x <= 0 (kernel values), table driven to keep the fp64 constant count (= registers)
// low: n = rint(x * 64/ln2), x = n ln2/64 + r with |r| <= ln2/128, exp(x) = 2^(n>>6) * T[n&63] *
// (1 + r + r^2/2 + r^3/6 + r^4/24 + r^5/120); the truncated term r^6/720 < 4e-17.  T[j] = 2^(j/64)
// lives in LDS (exp2_table_init).  Max error vs libm ~1 ulp; returns 0 below -745.
__device__ __forceinline__ void exp2_table_init(double* tab, int tid) {
  if (tid < 64) tab[tid] = exp2((double)tid * (1.0 / 64.0));
}

__device__ __forceinline__ double exp_neg(double x, const double* tab) {
  x = __builtin_fmax(x, -746.0);   // (a NaN becomes -746 -> 0, as the select did)
  // n = rint(x * 64 / ln 2) by the 1.5 * 2^52 trick (|n| < 2^31 here): the integer sits in the low word of the sum
  const double sh = __builtin_fma(x, 92.332482616893656877, 0x1.8p52);
  const int n = __double2loint(sh);
  const double nf = sh - 0x1.8p52;
  double r = __builtin_fma(nf, -0x1.62e42fee00000p-7, x);   // ln2/64, high part (21 trailing zero bits: nf * hi is exact)
  r = __builtin_fma(nf, -0x1.a39ef35793c76p-39, r);          // ln2/64, low part
  const double t = tab[n & 63];
  double p = __builtin_fma(r, 8.3333333333333332e-03, 4.1666666666666664e-02);
  p = __builtin_fma(p, r, 1.6666666666666666e-01);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p * r, r, r);     // r + r^2 (1/2 + ...)
  return __builtin_ldexp(__builtin_fma(t, p, t), n >> 6);
}

// k(d2) for squared scaled distance d2 >= 0 (without the outputscale).  The clamps are v_max/v_min (which
// drop a NaN operand); `d2 - d2` (0, or NaN for NaN / inf input) puts a NaN back, so that bad inputs
// still fail the factorisation the way psd_safe_cholesky fails on NaN.
template <int KIND>
__device__ __forceinline__ double kernel_from_sqdist(double d2, const double* exp_tab) {
  if (KIND == 0) {  // RBF: exp(-d2/2)
    return exp_neg(-0.5 * d2, exp_tab) + (d2 - d2);
  } else {          // Matern-5/2: gpytorch clamps d2 at 1e-30 before the sqrt
    // (upper clamp: k == 0 out there; it keeps the f32-seeded rsqrt in range)
    const double dd = __builtin_fmin(__builtin_fmax(d2, 1e-30), 1e30);
    const double ri = rsqrt_seeded(dd);
    const double r = sqrt_from_rinv(dd, ri);
    const double s5 = 2.2360679774997896964;
    double poly = __builtin_fma(__builtin_fma(r, 5.0 / 3.0, s5), r, 1.0);
    return __builtin_fma(poly, exp_neg(-s5 * r, exp_tab), d2 - d2);
  }
}

// The fused fit's variant: d2 comes off the matrix core (expanded form, may be slightly negative, never NaN --
// non-finite inputs are caught when the points are staged), the outputscale is folded into the polynomial
// (c0 = os, c1 = sqrt(5) os, c2 = 5/3 os), and r = d2 * rsqrt(d2) without the correction step (<= 1.5 ulp).
// d2 is clamped to [1e-30, 1e5]: k(1e5) ~ 1e-302 os, and -sqrt(5) r stays inside the range exp_neg handles
// without its own clamp.  Bare v_max / v_min: the builtins add a canonicalising v_max in front.
__device__ __forceinline__ double vmax_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
  return r;
}
__device__ __forceinline__ double vmin_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
  return r;
}
template <bool CLAMP>
__device__ __forceinline__ double exp_neg_t(double x, const double* tab) {
  if (CLAMP) x = vmax_f64(x, -746.0);
  // n = rint(x * 64 / ln 2) by the 1.5 * 2^52 trick (|n| < 2^31 here): the integer sits in the low word of the sum
  const double sh = __builtin_fma(x, 92.332482616893656877, 0x1.8p52);
  const int n = __double2loint(sh);
  const double nf = sh - 0x1.8p52;
  double r = __builtin_fma(nf, -0x1.62e42fee00000p-7, x);
  r = __builtin_fma(nf, -0x1.a39ef35793c76p-39, r);
  const double t = tab[n & 63];
  double p = __builtin_fma(r, 8.3333333333333332e-03, 4.1666666666666664e-02);
  p = __builtin_fma(p, r, 1.6666666666666666e-01);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p * r, r, r);
  return __builtin_ldexp(__builtin_fma(t, p, t), n >> 6);
}
template <int KIND>
__device__ __forceinline__ double kernel_from_sqdist_scaled(double d2, double c0, double c1, double c2, const double* exp_tab) {
  if (KIND == 0) {  // RBF: os exp(-d2/2)
    return c0 * exp_neg_t<true>(-0.5 * d2, exp_tab);
  } else {
    const double dd = vmin_f64(vmax_f64(d2, 1e-30), 1e5);
    // r = sqrt(dd) straight from the f32 seed y ~ 1/sqrt(dd): g = dd y, e = 1 - g y, r = g (1 + e/2 + 3 e^2/8)
    const double y = (double)__builtin_amdgcn_rsqf((float)dd);
    const double g = dd * y;
    const double e = __builtin_fma(-g, y, 1.0);
    const double r = __builtin_fma(g * e, __builtin_fma(e, 0.375, 0.5), g);
    const double poly = __builtin_fma(__builtin_fma(r, c2, c1), r, c0);
    return poly * exp_neg_t<false>(-2.2360679774997896964 * r, exp_tab);
  }
}

// os k(d2) and os dk / d(d2) of the scaled squared distance d2 (input gradients of the posterior):
//   RBF      k = os exp(-d2 / 2),                         dk/dd2 = -k / 2
//   Matern   k = os (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r),  dk/dd2 = -(5/6) os (1 + sqrt5 r) exp(-sqrt5 r)   (finite at r = 0)
template <int KIND>
__device__ __forceinline__ void kernel_and_slope_scaled(double d2, double os, const double* exp_tab, double& k, double& dk) {
  if (KIND == 0) {
    k = os * exp_neg_t<true>(-0.5 * d2, exp_tab);
    dk = -0.5 * k;
  } else {
    const double s5 = 2.2360679774997896964;
    const double dd = vmin_f64(vmax_f64(d2, 1e-30), 1e5);
    const double y = (double)__builtin_amdgcn_rsqf((float)dd);
    const double g = dd * y;
    const double e = __builtin_fma(-g, y, 1.0);
    const double r = __builtin_fma(g * e, __builtin_fma(e, 0.375, 0.5), g);
    const double ex = os * exp_neg_t<false>(-s5 * r, exp_tab);
    const double lin = __builtin_fma(s5, r, 1.0);
    k = __builtin_fma((5.0 / 3.0) * r, r, lin) * ex;
    dk = (-5.0 / 6.0) * lin * ex;
  }
}

// x summed over the four lane groups lq (lanes l, l^16, l^32, l^48), result in every lane: gfx950's
// v_permlane{32,16}_swap exchange half-waves / odd-even rows in one VALU op per dword (no LDS crossbar)
__device__ __forceinline__ double sum_lane_groups(double x) {
  {
    const unsigned lo = __double2loint(x), hi = __double2hiint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  }
  {
    const unsigned lo = __double2loint(x), hi = __double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  }
  return x;
}

// sum over the 16 lanes of a DPP row (same lq): the total lands in lane lc == 15
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum_to_lane15(double x) {
  x += dpp_mov_f64<0x118>(x);   // row_shr:8
  x += dpp_mov_f64<0x114>(x);   // row_shr:4
  x += dpp_mov_f64<0x112>(x);   // row_shr:2
  x += dpp_mov_f64<0x111>(x);   // row_shr:1
  return x;
}

// sum over all 64 lanes; the total lands in lanes 15, 31, 47, 63
__device__ __forceinline__ double wave_sum_to_lane15(double x) { return row_sum_to_lane15(sum_lane_groups(x)); }

// ---- blocked forward substitution on the matrix cores (posterior, L^-1, Cholesky solve) ----------
// One rank-16 step: acc -= L[16 kb + lc][16 j .. 16 j + 15] * V_j with V_j (16 rows x 16 right-hand sides)
// in the wave's LDS strip.  The contraction index is taken in the order k = 4 lq + m (lane group lq, MFMA
// k-step m) instead of 4 m + lq: a lane then needs FOUR CONTIGUOUS doubles of its row of L -- 128-byte
// row segments across the wave, two 16-byte loads per lane instead of four scattered 8-byte ones.  The
// strip is stored with the matching row permutation inside each 16-row block (row 4a + b at position
// 4b + a, see strip_row), which keeps the B-operand reads at the conflict-free stride.
struct LRowSeg {
  double a[4];
};
__device__ __forceinline__ int strip_row(int lq, int g) { return 4 * lq + g; }  // storage row of tile row lq + 4g

// vec: wave-uniform "whole tile in range and 16-byte aligned" (16 kb + 16 <= n, 16 j + 16 <= n, N even)
__device__ __forceinline__ LRowSeg load_lrow_seg(const double* Lrow, int col0, bool vec, bool row_ok, int n) {
  LRowSeg t;
  if (vec) {
    const double2 x = *reinterpret_cast<const double2*>(Lrow + col0);
    const double2 y = *reinterpret_cast<const double2*>(Lrow + col0 + 2);
    t.a[0] = x.x; t.a[1] = x.y; t.a[2] = y.x; t.a[3] = y.y;
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m) t.a[m] = (row_ok && col0 + m < n) ? Lrow[col0 + m] : 0.0;
  }
  return t;
}

// acc -/+= sum_{j = j0}^{kb-1} L[kb, j] V_j, the L row segments double-buffered one block ahead
template <bool NEG>
__device__ __forceinline__ d4_t block_row_accumulate(d4_t acc, const double* Lrow, bool row_ok, bool rows_in, int n, bool n_even,
                                                     const double* Vs, int j0, int kb, int lc, int lq) {
  if (j0 >= kb) return acc;
  LRowSeg cur = load_lrow_seg(Lrow, 16 * j0 + 4 * lq, rows_in && n_even && 16 * j0 + 16 <= n, row_ok, n);
  for (int j = j0; j < kb; ++j) {
    LRowSeg nxt = cur;
    if (j + 1 < kb) nxt = load_lrow_seg(Lrow, 16 * (j + 1) + 4 * lq, rows_in && n_even && 16 * (j + 1) + 16 <= n, row_ok, n);
    const double* vb = Vs + (16 * j + lq) * 16 + lc;
#pragma unroll
    for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.a[m], vb[4 * m * 16], acc, 0, 0, NEG ? 1 : 0);  // blgp = 1: -A
    cur = nxt;
  }
  return acc;
}

// The same sum with the row segments FOUR blocks ahead: with one block of look-ahead a product (4 MFMAs, 256 cycles) cannot hide an
// L2 round trip (~1,000 cycles) -- the L^-1 posterior pass spent three quarters of its time waiting for segments.
template <bool NEG>
__device__ __forceinline__ d4_t block_row_accumulate_deep(d4_t acc, const double* Lrow, bool row_ok, bool rows_in, int n, bool n_even,
                                                          const double* Vs, int j0, int kb, int lc, int lq) {
  if (j0 >= kb) return acc;
  auto ld = [&](int j) {
    LRowSeg t = {{0.0, 0.0, 0.0, 0.0}};
    if (j < kb) t = load_lrow_seg(Lrow, 16 * j + 4 * lq, rows_in && n_even && 16 * j + 16 <= n, row_ok, n);
    return t;
  };
  auto use = [&](const LRowSeg& c, int j) {
    const double* vb = Vs + (16 * j + lq) * 16 + lc;
#pragma unroll
    for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(c.a[m], vb[4 * m * 16], acc, 0, 0, NEG ? 1 : 0);
  };
  LRowSeg q0 = ld(j0), q1 = ld(j0 + 1), q2 = ld(j0 + 2), q3 = ld(j0 + 3);
  for (int j = j0; j < kb; j += 4) {
    use(q0, j);
    q0 = ld(j + 4);
    if (j + 1 < kb) use(q1, j + 1);
    q1 = ld(j + 5);
    if (j + 2 < kb) use(q2, j + 2);
    q2 = ld(j + 6);
    if (j + 3 < kb) use(q3, j + 3);
    q3 = ld(j + 7);
  }
  return acc;
}

__device__ __forceinline__ d4_t subst_accumulate(d4_t acc, const double* Lrow, bool row_ok, bool rows_in, int n, bool n_even,
                                                 const double* Vs, int j0, int kb, int lc, int lq) {
  return block_row_accumulate_deep<true>(acc, Lrow, row_ok, rows_in, n, n_even, Vs, j0, kb, lc, lq);   // (four tiles ahead, buffers rotating in place)
}

}  // namespace scaml

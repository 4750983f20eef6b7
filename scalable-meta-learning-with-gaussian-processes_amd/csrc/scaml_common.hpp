// Shared device helpers for the gfx950 GP kernels (fp64 on CDNA4).
//
// Measured on MI355X (tools/mfma_f64_probe.hip, profiles/r01_probe.txt): a v_fma_f64 costs a
// single wave 8 issue cycles, v_rsq_f64 / v_rcp_f64 ~180, so sqrt()/1.0/x (320/256 cycles) are
// replaced by an f32 seed (v_rsq_f32) plus one cubically-convergent fp64 step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4_t __attribute__((ext_vector_type(4)));

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

// exp(x) for x <= 0 (kernel values).  n = rint(x log2 e), r = x - n ln2 (two-part), degree-13
// Taylor on |r| <= 0.3466 (truncation 2e-17 rel.), scaled by 2^n with v_ldexp_f64.  Max
// observed error vs libm < 2 ulp; returns 0 below -745.
__device__ __forceinline__ double exp_neg(double x) {
  x = x < -746.0 ? -746.0 : x;
  double n = __builtin_rint(x * 1.4426950408889634074);
  double r = __builtin_fma(n, -6.93147180369123816490e-01, x);
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;            // 1/13!
  p = __builtin_fma(p, r, 2.08767569878681e-09);   // 1/12!
  p = __builtin_fma(p, r, 2.505210838544172e-08);  // 1/11!
  p = __builtin_fma(p, r, 2.755731922398589e-07);  // 1/10!
  p = __builtin_fma(p, r, 2.7557319223985893e-06); // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873e-05);   // 1/8!
  p = __builtin_fma(p, r, 1.984126984126984e-04);  // 1/7!
  p = __builtin_fma(p, r, 1.388888888888889e-03);  // 1/6!
  p = __builtin_fma(p, r, 8.333333333333333e-03);  // 1/5!
  p = __builtin_fma(p, r, 4.1666666666666664e-02); // 1/4!
  p = __builtin_fma(p, r, 1.6666666666666666e-01); // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)n);
}

// k(d2) for squared scaled distance d2 >= 0 (without the outputscale).
template <int KIND>
__device__ __forceinline__ double kernel_from_sqdist(double d2) {
  if (KIND == 0) {  // RBF: exp(-d2/2)
    return exp_neg(-0.5 * d2);
  } else {          // Matern-5/2: gpytorch clamps d2 at 1e-30 before the sqrt
    double dd = d2 < 1e-30 ? 1e-30 : d2;
    double r;
    if (dd <= 1e30) {
      double ri = rsqrt_seeded(dd);
      r = sqrt_from_rinv(dd, ri);
    } else {
      r = sqrt(dd);
    }
    const double s5 = 2.2360679774997896964;
    double poly = __builtin_fma(__builtin_fma(r, 5.0 / 3.0, s5), r, 1.0);
    return poly * exp_neg(-s5 * r);
  }
}

}  // namespace scaml

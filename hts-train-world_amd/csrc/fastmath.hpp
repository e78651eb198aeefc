// fastmath.hpp -- natural logarithm and exponential in FP64 for the per-bin loops of the hot kernels.
//
// The device library's log() and exp() are about 85 and 73 instructions each (special cases, extended-precision
// tails): CheapTrick takes nine of each per lane and frame (30 % of its instructions), the synthesis pulse kernel
// eighteen logarithms per voiced pulse.  wm_log / wm_exp are the classical kernels on a reduced argument
// (fdlibm's e_log.c scheme; a degree-13 polynomial for exp), about 35 and 22 instructions, within 1 ulp / 2 ulp of
// the correctly rounded result (tests/hooks/fastmath_check.cpp measures it against long double on the host, where
// the same source compiles).  Arguments outside the plain range (zero, denormal, infinite, NaN; |x| > 700 for exp)
// go to the library function.
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define WM_FM_HD __host__ __device__ __forceinline__
#else
#define WM_FM_HD inline
#endif

namespace wm {

// x = m 2^k with m in [0.5, 1), for normal finite x: one instruction each on the device (the library's frexp() adds
// seventy for the cases excluded here)
WM_FM_HD double fm_frexp(double x, int* k) {
#if defined(__HIP_DEVICE_COMPILE__)
  *k = __builtin_amdgcn_frexp_exp(x);
  return __builtin_amdgcn_frexp_mant(x);
#else
  return frexp(x, k);
#endif
}
// A coefficient held in scalar registers on the device: as a VOP3 operand it costs the vector ALU nothing, while the
// compiler's own choice (two v_mov_b32 per 64-bit literal in front of every v_fmac_f64) doubles the vector
// instructions of a polynomial.
WM_FM_HD double fm_k(double c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(WM_FM_PLAIN)
  asm volatile("" : "+s"(c));
#endif
  return c;
}
WM_FM_HD double fm_ldexp(double x, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ldexp(x, k);
#else
  return ldexp(x, k);
#endif
}

WM_FM_HD double wm_log(double x) {
  if (!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308)) return log(x);
  int k;
  double m = fm_frexp(x, &k);                       // x = m 2^k, m in [0.5, 1)
  if (m < 0.70710678118654752440) { m += m; --k; }  // m in [sqrt(1/2), sqrt(2))
  const double f = m - 1.0;
  // s = f / (2 + f): it only scales the small term s (hfsq + R) below, so a reciprocal refined to 1 ulp will do (six
  // instructions where the IEEE division takes twelve)
#if defined(__HIP_DEVICE_COMPILE__)
  const double d = 2.0 + f;
  double rc = __builtin_amdgcn_rcp(d);
  rc = fma(fma(-d, rc, 1.0), rc, rc);
  rc = fma(fma(-d, rc, 1.0), rc, rc);
  const double s = f * rc;
#else
  const double s = f / (2.0 + f);
#endif
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, fm_k(1.531383769920937332e-01), fm_k(2.222219843214978396e-01)), fm_k(3.999999999940941908e-01));
  const double t2 = z * fma(w, fma(w, fma(w, fm_k(1.479819860511658591e-01), fm_k(1.818357216161805012e-01)),
                                   fm_k(2.857142874366239149e-01)), fm_k(6.666666666666735130e-01));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  // k ln2_hi - ((hfsq - (s (hfsq + R) + k ln2_lo)) - f)
  return dk * fm_k(6.93147180369123816490e-01) - ((hfsq - (s * (hfsq + R) + dk * fm_k(1.90821492927058770002e-10))) - f);
}

// The coefficients of wm_exp as a pack: a loop over bins loads them once (fourteen scalar register pairs) instead of
// once per call -- the scalar unit issues for all four SIMDs of a CU, so 2 x 14 s_mov per call are not free either.
struct ExpK {
  double l2e, ln2h, ln2l, c[11];
  WM_FM_HD void load() {
    l2e = fm_k(1.44269504088896338700e+00);
    ln2h = fm_k(6.93147180369123816490e-01);
    ln2l = fm_k(1.90821492927058770002e-10);
    c[0] = fm_k(1.6059043836821613e-10);                  // 1/13!
    c[1] = fm_k(2.08767569878681e-09);                    // 1/12!
    c[2] = fm_k(2.505210838544172e-08);                   // 1/11!
    c[3] = fm_k(2.755731922398589e-07);                   // 1/10!
    c[4] = fm_k(2.7557319223985893e-06);                  // 1/9!
    c[5] = fm_k(2.48015873015873e-05);                    // 1/8!
    c[6] = fm_k(1.984126984126984e-04);                   // 1/7!
    c[7] = fm_k(1.388888888888889e-03);                   // 1/6!
    c[8] = fm_k(8.333333333333333e-03);                   // 1/5!
    c[9] = fm_k(4.1666666666666664e-02);                  // 1/4!
    c[10] = fm_k(1.6666666666666666e-01);                 // 1/3!
  }
};
WM_FM_HD double wm_exp_k(double x, const ExpK& k) {
  if (!(x >= -700.0 && x <= 700.0)) return exp(x);
  const double kd = rint(x * k.l2e);
  double r = fma(-kd, k.ln2h, x);
  r = fma(-kd, k.ln2l, r);                                // |r| <= 0.3466
  double p = k.c[0];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int i = 1; i < 11; ++i) p = fma(p, r, k.c[i]);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return fm_ldexp(p, (int)kd);
}
WM_FM_HD double wm_exp(double x) {
  ExpK k;
  k.load();
  return wm_exp_k(x, k);
}

// sin(pi x) and cos(pi x) (the library's sincospi() is 71 vector instructions; this one is about 30): x is split into
// a count n of quarter turns and r = x - n / 2, |r| <= 1/4 (exact), both functions are Taylor polynomials in r with the
// coefficients pi^k / k! (their tails at |r| = 1/4: 5e-17 and 2e-18), the quadrant swaps and negates.  Within 1.5 ulp
// (tests/hooks, fastmath_check).  No branch and no call: from 2^53 on every double is an even integer (sin 0, cos 1),
// the quadrant n mod 4 is taken in floating point (exact for any n), infinities and NaN come out as NaN.
// sqrt(x) for 0 <= x < 2^500, normal or zero (the library's sqrt() is 22 vector instructions: it rescales arguments
// near the ends of the exponent range): v_rsq_f64 (26 bits) carried to 52 by one coupled step for the root g and the
// half reciprocal root h, then one correction of g by its own residual.  Negative x and x = 0 give 0.
WM_FM_HD double wm_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double d = fma(-g, g, x);
  g = fma(d, h, g);
  return x > 0.0 ? g : 0.0;
#else
  return x > 0.0 ? sqrt(x) : 0.0;
#endif
}

struct SinCosPiK {
  double s[9], c[8];
  WM_FM_HD void load() {
    s[0] = fm_k(7.952054001475513e-07);
    s[1] = fm_k(-2.1915353447830217e-05);
    s[2] = fm_k(0.00046630280576761255);
    s[3] = fm_k(-0.0073704309457143504);
    s[4] = fm_k(0.08214588661112823);
    s[5] = fm_k(-0.5992645293207921);
    s[6] = fm_k(2.5501640398773455);
    s[7] = fm_k(-5.16771278004997);
    s[8] = fm_k(3.141592653589793);
    c[0] = fm_k(4.303069587032947e-06);
    c[1] = fm_k(-0.0001046381049248457);
    c[2] = fm_k(0.0019295743094039231);
    c[3] = fm_k(-0.02580689139001406);
    c[4] = fm_k(0.2353306303588932);
    c[5] = fm_k(-1.3352627688545895);
    c[6] = fm_k(4.0587121264167685);
    c[7] = fm_k(-4.934802200544679);
  }
};
WM_FM_HD void wm_sincospi_k(double x, const SinCosPiK& k, double* sn, double* cs) {
  x = fabs(x) < 9007199254740992.0 ? x : x * 0.0;
  const double n = rint(x + x);
  const double r = fma(-0.5, n, x);
  const double z = r * r;
  double ps = k.s[0], pc = k.c[0];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int i = 1; i < 9; ++i) ps = fma(ps, z, k.s[i]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int i = 1; i < 8; ++i) pc = fma(pc, z, k.c[i]);
  const double s0 = ps * r;
  const double c0 = fma(pc, z, 1.0);
  const int q = (int)fma(-4.0, rint(n * 0.25), n) & 3;   // quarter turns mod 4: -2 .. 2 -> two's complement & 3
  const bool swap = (q & 1) != 0;
  const double sv = swap ? c0 : s0, cv = swap ? s0 : c0;
  *sn = (q & 2) ? -sv : sv;
  *cs = ((q + 1) & 2) ? -cv : cv;
}
WM_FM_HD void wm_sincospi(double x, double* sn, double* cs) {
  SinCosPiK k;
  k.load();
  wm_sincospi_k(x, k, sn, cs);
}

}  // namespace wm

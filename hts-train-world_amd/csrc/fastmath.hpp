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

WM_FM_HD double wm_log(double x) {
  if (!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308)) return log(x);
  int k;
  double m = frexp(x, &k);                          // x = m 2^k, m in [0.5, 1)
  if (m < 0.70710678118654752440) { m += m; --k; }  // m in [sqrt(1/2), sqrt(2))
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                            6.666666666666735130e-01);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  // k ln2_hi - ((hfsq - (s (hfsq + R) + k ln2_lo)) - f)
  return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

WM_FM_HD double wm_exp(double x) {
  if (!(x >= -700.0 && x <= 700.0)) return exp(x);
  const double kd = rint(x * 1.44269504088896338700e+00);
  double r = fma(-kd, 6.93147180369123816490e-01, x);
  r = fma(-kd, 1.90821492927058770002e-10, r);      // |r| <= 0.3466
  double p = 1.6059043836821613e-10;                // 1/13!
  p = fma(p, r, 2.08767569878681e-09);              // 1/12!
  p = fma(p, r, 2.505210838544172e-08);             // 1/11!
  p = fma(p, r, 2.755731922398589e-07);             // 1/10!
  p = fma(p, r, 2.7557319223985893e-06);            // 1/9!
  p = fma(p, r, 2.48015873015873e-05);              // 1/8!
  p = fma(p, r, 1.984126984126984e-04);             // 1/7!
  p = fma(p, r, 1.388888888888889e-03);             // 1/6!
  p = fma(p, r, 8.333333333333333e-03);             // 1/5!
  p = fma(p, r, 4.1666666666666664e-02);            // 1/4!
  p = fma(p, r, 1.6666666666666666e-01);            // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)kd);
}

}  // namespace wm

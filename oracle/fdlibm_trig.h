/*
 * ORACLE / TEST INFRASTRUCTURE - atan, atan2 and asin as the JavaScript engines compute them.
 *
 * V8 (and SpiderMonkey) implement Math.atan2 / Math.asin with a port of Sun's fdlibm (V8: src/base/ieee754.cc; FreeBSD
 * msun s_atan.c, e_atan2.c, e_asin.c): fixed argument reductions, fixed polynomial coefficients, plain binary64 operations -
 * deterministic on every platform.  This is a restatement of those published algorithms, operation for operation (compile with
 * -ffp-contract=off; sqrt and division correctly rounded), so that the C restatement of the hot path computes the samplers'
 * u and v (main.js:127-128, 446-447) with the SAME bits as the reference under Node instead of whatever the host's libm gives.
 * Pinned against Node itself: tests/test_oracle.py::test_fdlibm_trig_matches_node compares 0.9 M vectors (random, tiny, huge,
 * special values) bit for bit with Math.atan2 / Math.asin.  The strict kernel carries the same code (csrc/rt_kernel.hip).
 */
#ifndef FDLIBM_TRIG_H
#define FDLIBM_TRIG_H
#include <stdint.h>
#include <string.h>
#include <math.h>
static inline uint32_t fd_hi(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
static inline uint32_t fd_lo(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }
static inline double fd_zero_lo(double x) { uint64_t u; memcpy(&u, &x, 8); u &= 0xffffffff00000000ull; memcpy(&x, &u, 8); return x; }
static const double fd_atanhi[] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00};
static const double fd_atanlo[] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17};
static const double fd_aT[] = {3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
  9.09088713343650656196e-02, -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02, 4.97687799461593236017e-02,
  -3.65315727442169155270e-02, 1.62858201153657823623e-02};
static inline double fd_atan(double x) {
  const int32_t hx = (int32_t)fd_hi(x); const uint32_t ix = (uint32_t)hx & 0x7fffffffu; int id;
  if (ix >= 0x44100000u) {
    if (ix > 0x7ff00000u || (ix == 0x7ff00000u && fd_lo(x) != 0)) return x + x;
    return hx > 0 ? fd_atanhi[3] + fd_atanlo[3] : -fd_atanhi[3] - fd_atanlo[3];
  }
  if (ix < 0x3fdc0000u) { if (ix < 0x3e200000u) return x; id = -1; }
  else {
    x = fabs(x);
    if (ix < 0x3ff30000u) { if (ix < 0x3fe60000u) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); } else { id = 1; x = (x - 1.0) / (x + 1.0); } }
    else { if (ix < 0x40038000u) { id = 2; x = (x - 1.5) / (1.0 + 1.5 * x); } else { id = 3; x = -1.0 / x; } }
  }
  const double z = x * x, w = z * z;
  const double s1 = z * (fd_aT[0] + w * (fd_aT[2] + w * (fd_aT[4] + w * (fd_aT[6] + w * (fd_aT[8] + w * fd_aT[10])))));
  const double s2 = w * (fd_aT[1] + w * (fd_aT[3] + w * (fd_aT[5] + w * (fd_aT[7] + w * fd_aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  const double r = fd_atanhi[id] - ((x * (s1 + s2) - fd_atanlo[id]) - x);
  return hx < 0 ? -r : r;
}
static inline double fd_atan2(double y, double x) {
  const double tiny = 1.0e-300, pi_o_4 = 7.8539816339744827900E-01, pi_o_2 = 1.5707963267948965580E+00, pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
  const int32_t hx = (int32_t)fd_hi(x), hy = (int32_t)fd_hi(y); const uint32_t lx = fd_lo(x), ly = fd_lo(y);
  const uint32_t ix = (uint32_t)hx & 0x7fffffffu, iy = (uint32_t)hy & 0x7fffffffu;
  if ((ix | ((lx | (0u - lx)) >> 31)) > 0x7ff00000u || (iy | ((ly | (0u - ly)) >> 31)) > 0x7ff00000u) return x + y;
  if ((((uint32_t)hx - 0x3ff00000u) | lx) == 0) return fd_atan(y);
  int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
  if ((iy | ly) == 0) { switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; } }
  if ((ix | lx) == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7ff00000u) {
    if (iy == 0x7ff00000u) { switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny; case 2: return 3.0 * pi_o_4 + tiny; default: return -3.0 * pi_o_4 - tiny; } }
    else { switch (m) { case 0: return 0.0; case 1: return -0.0; case 2: return pi + tiny; default: return -pi - tiny; } }
  }
  if (iy == 0x7ff00000u) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int32_t k = (int32_t)(iy - ix) >> 20;
  double z;
  if (k > 60) { z = pi_o_2 + 0.5 * pi_lo; m &= 1; }
  else if (hx < 0 && k < -60) z = 0.0;
  else z = fd_atan(fabs(y / x));
  switch (m) { case 0: return z; case 1: return -z; case 2: return pi - (z - pi_lo); default: return (z - pi_lo) - pi; }
}
static inline double fd_asin(double x) {
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pio4_hi = 7.85398163397448278999e-01;
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
               pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
  const double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  const int32_t hx = (int32_t)fd_hi(x); const uint32_t ix = (uint32_t)hx & 0x7fffffffu;
  double t = 0.0, w, p, q, c, r, s;
  if (ix >= 0x3ff00000u) {
    if (((ix - 0x3ff00000u) | fd_lo(x)) == 0) return x * pio2_hi + x * pio2_lo;
    return (x - x) / (x - x);
  } else if (ix < 0x3fe00000u) {
    if (ix < 0x3e400000u) return x;
    t = x * x;
    p = t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
    q = 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
    w = p / q;
    return x + x * w;
  }
  w = 1.0 - fabs(x);
  t = w * 0.5;
  p = t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
  q = 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
  s = sqrt(t);
  if (ix >= 0x3FEF3333u) { w = p / q; t = pio2_hi - (2.0 * (s + s * w) - pio2_lo); }
  else { w = fd_zero_lo(s); c = (t - w * w) / (s + w); r = p / q; p = 2.0 * s * r - (pio2_lo - 2.0 * c); q = pio4_hi - 2.0 * w; t = pio4_hi - (p - q); }
  return hx > 0 ? t : -t;
}
#endif

// rt_kernel.hip — the per-pixel ray-sphere trace/shade loop as ONE hand-written gfx950 kernel.
//
// What it computes, per output pixel (reference lines in /root/reference/main.js):
//   A1  primary ray                     :102-105, :186-193   (dist indexed by component, quirk q1)
//   A2  ray-sphere intersection         :420-451
//   A3  closest hit, strict <           :223-231
//   A4  reflection / A5 refraction      :233-266
//   A6  bounded recursion               :221, :268-278       (stackless forward fold, or explicit per-lane stack)
//   A7  lights, shadows, Phong          :280-318              (light_intensity shared across lights, q2)
//   A8  samplers (colour/texture/checker) :404, :343-351, :126-133
//   A9  combine + non-linear clamp      :320-336
//   A10 RGBA8 store                     :195-198              (Uint8ClampedArray: clamp, round-half-even)
//
// MI355X mapping (no MFMA: there is no dense contraction anywhere in this path; all arithmetic is
// binary64 VALU, which is what JS numbers are):
//   * one work-item per pixel (per SAMPLE when supersampling 2x2); a wave owns a compact 8x8 block so its
//     64 rays take the same branches; 4 waves side by side make a 32x8 workgroup tile = whole
//     128-byte framebuffer lines, each line written by exactly one workgroup (no cross-XCD sharing);
//     product kernel: a FLAT grid whose workgroups look their tile up in a launch table built on the GPU (one
//     s_load_dwordx4: no tile arithmetic, no division) that lists the tiles dearest first, so a launch ends
//     on cheap sky tiles (rt_api.hip: dispatch_order); strict kernel: the plain 2-D grid;
//   * everything wave-uniform — camera, lights, loop bounds (kernarg) and the sphere tables walked by
//     the uniform object loops (typed address_space(4)) — is read with SCALAR loads into SGPRs: the
//     intersection loops issue no vector memory and no LDS instruction;
//   * the per-hit, per-lane data (material + sampler parameters of the sphere that lane hit, texture
//     descriptors), the primary-ray cull rectangles and the 10-double state of the stackless recursion
//     fold live in LDS; the LDS image is one contiguous block in HBM, loaded behind the ray generation;
//   * texels are plain global loads (gfx950 has no image/texture path); the two 128 KB textures stay
//     L2-resident;
//   * product kernel only: "anchored" line-sphere discriminants for primary rays (camera) and shadow
//     rays (walked from the light): 4 operations instead of 10; a wave-wide cull of primary-ray
//     candidates (__ballot over per-sphere screen rectangles); per-light shadow grids when the scene has
//     many spheres; an enclosing sphere (skybox) kept out of the loops; hardware rsq/rcp + Newton;
//   * divergent phases are exec-mask branches the compiler lowers to s_cbranch_execz; the bookkeeping of
//     a hit is pinned inside its branch, so a wave whose 64 rays all miss a sphere pays 4 FP64 operations
//     and one compare for it.
//
// The file is compiled twice (csrc/Makefile): RT_STRICT=0 with FMA contraction (the product kernels)
// and RT_STRICT=1 with -ffp-contract=off (operation for operation with the JS expression trees, IEEE
// sqrt/div, fdlibm atan2/asin, OCML pow, explicit recursion stack; RT_FLAG_STRICT_FP).  Both are held to <= 1 LSB on generic
// samples.  Samples whose outcome in the reference is decided by the last bit of its own arithmetic - a sampler coordinate
// within rounding of a texel / checker boundary, rays with an exactly-zero direction component (the centre row / column of an
// odd sample grid) that stay in a coordinate plane through sphere centres - can only be reproduced by the reference's own
// operation sequence.  The product kernels MARK the former while tracing (a list in HBM); the strict build has a second,
// list-driven kernel (rt_retrace) that traces the marked samples and the centre row / column again and stores over them
// (rt_api.hip: render_batch_impl launches it unless it KNOWS that a frame of this scene, camera, size and tile set has neither).
// Scenes that sit on a coincidence as a whole (a light exactly on a surface, a camera with a zero axis sum:
// rt_scene_dev::needs_strict) take the strict kernels throughout.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "rt_device.h"

#ifndef RT_STRICT
#define RT_STRICT 0
#endif
#if RT_STRICT
#define RT_LAUNCH_NAME rt_launch_trace_strict
#define RT_SCRATCH_NAME rt_scratch_trace_strict
#else
#define RT_LAUNCH_NAME rt_launch_trace_fast
#define RT_SCRATCH_NAME rt_scratch_trace_fast
#endif

// Register budget: minimum waves per SIMD the kernel must fit (second __launch_bounds__ argument).
#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 2
#endif

#define RT_INF __builtin_inf()

namespace {

struct v3 { double x, y, z; };
__device__ __forceinline__ v3 mk(double x, double y, double z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
// main.js:49-51 — (a0*b0 + a1*b1) + a2*b2
__device__ __forceinline__ double dot(const v3 a, const v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// ---- math layer ----------------------------------------------------------------------------
// RT_STRICT: IEEE-754 correctly rounded sqrt and division, fdlibm's atan2 / asin and OCML pow, operation for operation
// with the JS expression trees.  Otherwise (product kernel): the hardware estimates v_rsq_f64 /
// v_rcp_f64 refined by Newton steps in FMA arithmetic (<= ~1 ulp, no denormal pre-scaling, no
// div_scale/div_fixup), reciprocal-multiplies for divisions by constants, and integer powers by
// square-and-multiply.  All of it stays binary64; the differences are last-ulp effects, which
// the +-1 LSB tolerance exists for (tests/test_gpu_parity.py holds both kernels to it).
#if RT_STRICT
__device__ __forceinline__ double rt_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ double rt_sqrt_nn(double x) { return sqrt(x); }
__device__ __forceinline__ double rt_rcp(double x) { return 1.0 / x; }
__device__ __forceinline__ double rt_div(double a, double b) { return a / b; }
__device__ __forceinline__ double rt_pow(double x, double e) { return pow(x, e); }
#define RT_DIV_CONST(x, c) ((x) / (c))
// main.js:62-66 — multiply by 1/len; the zero vector is returned unchanged
__device__ __forceinline__ v3 unit(const v3 v, double *len_out) {
  const double l = sqrt(dot(v, v));
  *len_out = l;
  if (l != 0.0) { const double s = 1.0 / l; return mk(v.x * s, v.y * s, v.z * s); }
  return v;
}
#else
// Measured on MI355X (build/probe/prec.hip, 1M random inputs over 1e-6..1e8): v_rsq_f64 / v_rcp_f64 are good to
// 2^-24; one Newton step gives 4.1e-15 / 2.1e-15, two give 1.4e-16 / 1.1e-16; x*rsqrt(x) after ONE step plus
// the residual correction g += (x - g*g) * y/2 is a square root good to 1.1e-16.
// rt_rsqrt_pos is only ever used to NORMALISE a vector.  An error in the scale of a direction moves no hit
// point (p + d*t is invariant under rescaling d) and no reflection direction; it reaches only continuous
// quantities (a cosine, a distance) at the 4e-15 level, so one Newton step is enough there.
__device__ __forceinline__ double rt_rsqrt_pos(double m) {           // m > 0, finite
  const double y = __builtin_amdgcn_rsq(m);
  const double e = __builtin_fma(-(m * y), y, 1.0);
  return __builtin_fma(0.5 * y, e, y);
}
__device__ __forceinline__ double rt_sqrt(double x) {
  const double y = rt_rsqrt_pos(x);
  double g = x * y;
  g = __builtin_fma(__builtin_fma(-g, g, x), 0.5 * y, g);
  return (x > 0.0) ? g : x;                                           // +0 -> 0, NaN -> NaN, x < 0 -> x (callers never pass it)
}
// x >= 0 (or NaN): the +0 case is kept exact by clamping the estimate (rsq(0) = inf) instead of
// selecting afterwards: 0 * 1e100 = 0 through every step below.
__device__ __forceinline__ double rt_sqrt_nn(double x) {
  double y = __builtin_fmin(__builtin_amdgcn_rsq(x), 1e100);
  const double e = __builtin_fma(-(x * y), y, 1.0);
  y = __builtin_fma(0.5 * y, e, y);
  const double g = x * y;
  return __builtin_fma(__builtin_fma(-g, g, x), 0.5 * y, g);
}
__device__ __forceinline__ double rt_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
  return __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
}
__device__ __forceinline__ double rt_div(double a, double b) {
  const double r = rt_rcp(b);
  const double q = a * r;
  return __builtin_fma(__builtin_fma(-q, b, a), r, q);               // one correction step on the quotient
}
#define RT_DIV_CONST(x, c) rt_div_const((x), (c), 1.0 / (c))
__device__ __forceinline__ double rt_div_const(double a, double c, double rc) {
  const double q = a * rc;
  return __builtin_fma(__builtin_fma(-q, c, a), rc, q);
}
// Non-integer exponents (none in the reference scene) take OCML's pow out of line, so that its ~40
// temporaries are not part of the register budget of the loop every pixel runs.
__device__ __attribute__((noinline)) double rt_pow_generic(double x, double e) { return pow(x, e); }
// x > 0.  Integer exponents (every specular_exponent of the reference scene, main.js:108-123) by
// square-and-multiply: <= 2*log2(e) multiplies instead of OCML's ~150-instruction pow.
__device__ __forceinline__ double rt_pow(double x, double e) {
  const int n = (int)e;
  if ((double)n == e && n >= 0 && n <= 65536) {
    double r = 1.0, b = x;
    unsigned k = (unsigned)n;
    while (k) { if (k & 1u) r *= b; b *= b; k >>= 1; }
    return r;
  }
  return rt_pow_generic(x, e);
}
// main.js:62-66 — v * (1/len), len = sqrt(v.v); the zero vector is returned unchanged
__device__ __forceinline__ v3 unit(const v3 v, double *len_out) {
  const double m = dot(v, v);
  const double s = rt_rsqrt_pos(m);
  const bool ok = (m > 0.0);
  *len_out = ok ? m * s : m;
  return ok ? mk(v.x * s, v.y * s, v.z * s) : v;
}
#endif

// ---- atan2 / asin for the samplers (main.js:127-128, 446-447) --------------------------------------------------------
// RT_STRICT: fdlibm's, as the JS engines' (below).  Product kernel: OCML's argument reductions and minimax polynomials (atan: odd polynomial of degree 39
// on [0,1] after q = min/max; asin: x + x*r*P(r), r = x^2 below 1/2 and (1-|x|)/2 above with pi/2 - 2*asin(sqrt(r))), but
//   * every Horner step is ONE v_fma_f64 whose constant comes from an SGPR pair (hipcc otherwise writes the 64-bit literal
//     into the v_fmac accumulator with two v_mov_b32 per step: 64 of OCML's ~230 instructions for the pair of calls),
//   * the quotient is the kernel's rcp + Newton division, the square root its rsq + Newton one,
//   * the branch above 1/2 finishes in working precision instead of OCML's double-double tail.
// Accuracy: <= 2 ulp (OCML: <= 1); what the samplers make of it is a texel index / a checker parity, i.e. the same last-ulp
// sensitivity at boundaries that OCML, glibc and V8 already have among themselves (the one listed
// exception); the parity suite and the soaks hold the result to 1 LSB.
#if RT_STRICT
// The strict kernel computes atan2 / asin AS THE JAVASCRIPT ENGINES DO: V8 and SpiderMonkey implement Math.atan2 / Math.asin with
// a port of Sun's fdlibm (fixed argument reductions and coefficients, plain binary64 operations - deterministic everywhere), so
// restating those published algorithms operation for operation (this build has no FMA contraction; `/` and sqrt are correctly
// rounded) gives u and v the reference's own bits, where OCML's functions differ in the last ulp on a few per cent of the
// inputs - and an ulp at a texel or checker boundary is a different pixel.  Same code as
// oracle/fdlibm_trig.h, which the CPU tests compare with Node's Math.atan2 / Math.asin bit for bit on 0.9 M vectors.
__device__ __forceinline__ uint32_t fd_hi(double x) { return (uint32_t)(__builtin_bit_cast(unsigned long long, x) >> 32); }
__device__ __forceinline__ uint32_t fd_lo(double x) { return (uint32_t)__builtin_bit_cast(unsigned long long, x); }
__device__ __forceinline__ double fd_atan(double x) {
  const double hi0 = 4.63647609000806093515e-01, hi1 = 7.85398163397448278999e-01, hi2 = 9.82793723247329054082e-01, hi3 = 1.57079632679489655800e+00;
  const double lo0 = 2.26987774529616870924e-17, lo1 = 3.06161699786838301793e-17, lo2 = 1.39033110312309984516e-17, lo3 = 6.12323399573676603587e-17;
  const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01, a2 = 1.42857142725034663711e-01, a3 = -1.11111104054623557880e-01,
               a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02, a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02,
               a8 = 4.97687799461593236017e-02, a9 = -3.65315727442169155270e-02, a10 = 1.62858201153657823623e-02;
  const int32_t hx = (int32_t)fd_hi(x);
  const uint32_t ix = (uint32_t)hx & 0x7fffffffu;
  int id;
  if (ix >= 0x44100000u) {                                            // |x| >= 2^66
    if (ix > 0x7ff00000u || (ix == 0x7ff00000u && fd_lo(x) != 0u)) return x + x;
    return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
  }
  if (ix < 0x3fdc0000u) { if (ix < 0x3e200000u) return x; id = -1; }   // |x| < 0.4375 (< 2^-29: x)
  else {
    x = __builtin_fabs(x);
    if (ix < 0x3ff30000u) { if (ix < 0x3fe60000u) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); } else { id = 1; x = (x - 1.0) / (x + 1.0); } }
    else { if (ix < 0x40038000u) { id = 2; x = (x - 1.5) / (1.0 + 1.5 * x); } else { id = 3; x = -1.0 / x; } }
  }
  const double z = x * x, w = z * z;
  const double s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
  const double s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
  if (id < 0) return x - x * (s1 + s2);
  const double ahi = id == 0 ? hi0 : (id == 1 ? hi1 : (id == 2 ? hi2 : hi3)), alo = id == 0 ? lo0 : (id == 1 ? lo1 : (id == 2 ? lo2 : lo3));
  const double r = ahi - ((x * (s1 + s2) - alo) - x);
  return hx < 0 ? -r : r;
}
__device__ __forceinline__ double fd_atan2(double y, double x) {
  const double tiny = 1.0e-300, pi_o_4 = 7.8539816339744827900E-01, pi_o_2 = 1.5707963267948965580E+00, pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
  const int32_t hx = (int32_t)fd_hi(x), hy = (int32_t)fd_hi(y);
  const uint32_t lx = fd_lo(x), ly = fd_lo(y), ix = (uint32_t)hx & 0x7fffffffu, iy = (uint32_t)hy & 0x7fffffffu;
  if ((ix | ((lx | (0u - lx)) >> 31)) > 0x7ff00000u || (iy | ((ly | (0u - ly)) >> 31)) > 0x7ff00000u) return x + y;     // NaN
  if ((((uint32_t)hx - 0x3ff00000u) | lx) == 0u) return fd_atan(y);                                                    // x == 1
  int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                                                                          // 2 * sign(x) + sign(y)
  if ((iy | ly) == 0u) { if (m < 2) return y; return m == 2 ? pi + tiny : -pi - tiny; }                                   // y == 0
  if ((ix | lx) == 0u) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;                                                 // x == 0
  if (ix == 0x7ff00000u) {
    if (iy == 0x7ff00000u) return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0 * pi_o_4 + tiny : -3.0 * pi_o_4 - tiny));
    return m == 0 ? 0.0 : (m == 1 ? -0.0 : (m == 2 ? pi + tiny : -pi - tiny));
  }
  if (iy == 0x7ff00000u) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int32_t k = (int32_t)(iy - ix) >> 20;
  double z;
  if (k > 60) { z = pi_o_2 + 0.5 * pi_lo; m &= 1; }                                                                      // |y / x| > 2^60
  else if (hx < 0 && k < -60) z = 0.0;                                                                                  // 0 > |y| / x > -2^-60
  else z = fd_atan(__builtin_fabs(y / x));
  return m == 0 ? z : (m == 1 ? -z : (m == 2 ? pi - (z - pi_lo) : (z - pi_lo) - pi));
}
__device__ __forceinline__ double fd_asin(double x) {
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pio4_hi = 7.85398163397448278999e-01;
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
               pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
  const double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  const int32_t hx = (int32_t)fd_hi(x);
  const uint32_t ix = (uint32_t)hx & 0x7fffffffu;
  if (ix >= 0x3ff00000u) {                                            // |x| >= 1
    if (((ix - 0x3ff00000u) | fd_lo(x)) == 0u) return x * pio2_hi + x * pio2_lo;
    return (x - x) / (x - x);                                         // NaN
  }
  if (ix < 0x3fe00000u) {                                             // |x| < 0.5
    if (ix < 0x3e400000u) return x;
    const double t = x * x;
    const double p = t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
    const double q = 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
    return x + x * (p / q);
  }
  const double w0 = 1.0 - __builtin_fabs(x);
  double t = w0 * 0.5;
  const double p = t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
  const double q = 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
  const double s = sqrt(t);
  if (ix >= 0x3FEF3333u) { const double w = p / q; t = pio2_hi - (2.0 * (s + s * w) - pio2_lo); }                       // |x| > 0.975
  else {
    const double w = __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, s) & 0xffffffff00000000ull);
    const double c = (t - w * w) / (s + w), r = p / q;
    const double p2 = 2.0 * s * r - (pio2_lo - 2.0 * c), q2 = pio4_hi - 2.0 * w;
    t = pio4_hi - (p2 - q2);
  }
  return hx > 0 ? t : -t;
}
// atan2(y, x) and asin(w) of one surface normal (the two halves of main.js:446-447 / :127-128)
__device__ __forceinline__ void rt_atan2_asin(double y, double x, double w, double *at, double *as) { *at = fd_atan2(y, x); *as = fd_asin(w); }
#else
__device__ __forceinline__ double rt_fma_k(double a, double b, double k) {     // a*b + k, k wave-uniform: v_fma_f64 v, v, v, s[..]
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
  return r;
}
typedef const double __attribute__((address_space(4))) *rt_trig_kptr;
typedef double __attribute__((ext_vector_type(4))) rt_d4;
// The polynomial coefficients (OCML's, as 64-bit patterns) sit in a constant-memory table read with scalar loads: an
// s_load_dwordx8 brings four of them into SGPRs with ONE scalar instruction, where immediates would take two s_mov_b32 each
// (the scalar unit is shared by the CU's four SIMDs and is ~60 % busy in this kernel: measured, immediates made the kernel slower).
// The two Horner chains are independent, so their steps ALTERNATE (a dependent v_fma_f64 waits for its predecessor; the other
// chain's step fills the gap), and the coefficients come in eight 32-byte groups laid out for that order - atan c0..3 | atan c4..7 |
// then two steps of each chain per group: atan c8, asin c0, atan c9, asin c1 | ... - each fetched two groups ahead of its use
// (one ahead: -0.4 %, profiles/r03_ab_log.md), the first two before the quotient's dependent chain.  (Round 2 fetched 64-byte
// groups: 32 scalar registers of coefficients at the kernel's scalar-pressure peak, where the kernel had none to spare; this way
// it is 24.)
#define RT_TRIG_AHEAD 2u
__constant__ unsigned long long RT_TRIG_BITS[32] = {
    0x3eeba404b5e68a13ull, 0xbf23e260bd3237f4ull, 0x3f4b2bb069efb384ull, 0xbf67952daf56de9bull,
    0x3f7d6d43a595c56full, 0xbf8c6ea4a57d9582ull, 0x3f967e295f08b19full, 0xbf9e9ae6fc27006aull,
    0x3fa2c15b5711927aull, 0x3fa059859fea6a70ull, 0xbfa59976e82d3ff0ull, 0xbf90a5a378a05eafull,
    0x3fa82d5d6ef28734ull, 0x3f94052137024d6aull, 0xbfaae5ce6a214619ull, 0x3f7ab3a098a70509ull,
    0x3fae1bb48427b883ull, 0x3f88ed60a300c8d2ull, 0xbfb110e48b207f05ull, 0x3f8c6fa84b77012bull,
    0x3fb3b13657b87036ull, 0x3f91c6c111dccb70ull, 0xbfb745d119378e4full, 0x3f96e89f0a0adacfull,
    0x3fbc71c717e1913cull, 0x3f9f1c72c668963full, 0xbfc2492492376b7dull, 0x3fa6db6db41ce4bdull,
    0x3fc99999999952ccull, 0x3fb333333336fd5bull, 0xbfd5555555555523ull, 0x3fc5555555555380ull};
__device__ __forceinline__ void rt_atan2_asin(double y, double x, double w, double *at, double *as) {
  rt_trig_kptr K = (rt_trig_kptr)(const void *)RT_TRIG_BITS;
  asm volatile("" : "+s"(K));                        // opaque: the reads below stay scalar LOADS instead of being folded back into immediates
#define RT_TRIG_GROUP(I) (*(const rt_d4 __attribute__((address_space(4))) *)(K + 4 * (I)))
  rt_d4 g[8];
  g[0] = RT_TRIG_GROUP(0); g[1] = RT_TRIG_GROUP(1);
  const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
  const double hi = __builtin_fmax(ax, ay), lo = __builtin_fmin(ax, ay);
  const double q = rt_div(lo, hi);                                   // in [0,1]; 0/0 (both zero) handled below
  const double z = q * q;
  const double yw = __builtin_fabs(w);
  const double t = __builtin_fma(yw, -0.5, 0.5);                     // (1 - |w|) / 2
  const bool big = (yw >= 0.5);
  const double r = big ? t : w * w;
  g[2] = RT_TRIG_GROUP(2);
  double p = g[0][0];
  p = rt_fma_k(p, z, g[0][1]); p = rt_fma_k(p, z, g[0][2]); p = rt_fma_k(p, z, g[0][3]);
  g[3] = RT_TRIG_GROUP(3);
  p = rt_fma_k(p, z, g[1][0]); p = rt_fma_k(p, z, g[1][1]); p = rt_fma_k(p, z, g[1][2]); p = rt_fma_k(p, z, g[1][3]);
  double pa = 0.0;
#pragma unroll
  for (uint32_t i = 2; i < 8; i++) {                                 // group i: atan step, asin step, atan step, asin step
    if (i + RT_TRIG_AHEAD < 8u) g[i + RT_TRIG_AHEAD] = RT_TRIG_GROUP(i + RT_TRIG_AHEAD);
    p = rt_fma_k(p, z, g[i][0]);
    pa = (i == 2u) ? g[i][1] : rt_fma_k(pa, r, g[i][1]);
    p = rt_fma_k(p, z, g[i][2]);
    pa = rt_fma_k(pa, r, g[i][3]);
  }
#undef RT_TRIG_GROUP
  // atan2: quadrant and special cases
  double a = __builtin_fma(q, z * p, q);                             // atan(q), q in [0,1]
  a = (ay > ax) ? (M_PI / 2.0 - a) : a;
  const bool xneg = (__builtin_bit_cast(unsigned long long, x) >> 63) != 0;      // the sign BIT: atan2(+-0, -0) = +-pi
  a = xneg ? (M_PI - a) : a;
  a = (hi == 0.0) ? (xneg ? M_PI : 0.0) : a;                         // atan2(+-0, +-0)
  *at = __builtin_copysign(a, y);                                    // NaN in, NaN out (every step above propagates it)
  // asin: x + x*r*P(r) below 1/2, pi/2 - 2*asin(sqrt((1-|x|)/2)) above
  pa = pa * r;
  const double sq = big ? rt_sqrt_nn(t) : yw;
  const double ww = __builtin_fma(sq, pa, sq);                       // asin(sq)
  double b = big ? __builtin_fma(-2.0, ww, M_PI / 2.0) : ww;
  b = (yw > 1.0) ? __builtin_nan("") : b;                            // |w| > 1 by an ulp (a ray through the exact pole): NaN, as Math.asin gives
  *as = __builtin_copysign(b, w);
}
#endif

// main.js:40-43 — v + n * (-(2 * v.n))
__device__ __forceinline__ v3 reflect(const v3 v, const v3 n) {
  const double t = -(2.0 * dot(v, n));
  return mk(v.x + n.x * t, v.y + n.y * t, v.z + n.z * t);
}
// Math.min(1, x) / Math.max(a, x) as used at main.js:316-317, :333-335 (NaN in x propagates)
__device__ __forceinline__ double min1(double x) { return (x > 1.0) ? 1.0 : x; }
__device__ __forceinline__ double maxa(double a, double x) { return (x < a) ? a : x; }

// Counter-based stand-in for Math.random() in the stars sampler (main.js:135-139): lowbias32 twice over the sample's
// index in the frame and the node's position in the ray tree (root 1, reflect child 2p, refract child 2p+1).
// Identical in oracle/restate.js and oracle/rt_oracle.c.
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ double star_uniform(uint32_t pix_lo, uint32_t pix_hi, uint32_t path) {
  return (double)lowbias32(pix_lo ^ lowbias32(path + 0x9e3779b9u * (pix_hi + 1u))) * (1.0 / 4294967296.0);
}

// ECMAScript ToInt32(x) & 1   (main.js:129-130)
__device__ __forceinline__ int to_int32_bit0(double x) {
  if (fabs(x) < 2147483648.0) return (int)x & 1;     // the common case: one truncating conversion
  if (!(fabs(x) < RT_INF)) return 0;                 // NaN, +-Infinity -> 0
  double t = trunc(x);
  if (fabs(t) >= 4294967296.0) t = t - floor(t / 4294967296.0) * 4294967296.0;
  return (int)((long long)t & 1);
}

// Uint8ClampedArray store of 255*c (main.js:195-197): NaN -> 0, clamp, round half to even
__device__ __forceinline__ uint32_t to_byte(double c) {
#if RT_STRICT
  // fmax(NaN, 0) = 0 and the clamp precede the conversion, so v_cvt_u32_f64 never sees an out-of-range
  // value; v_rndne_f64 rounds half to even.
  return (uint32_t)__builtin_rint(__builtin_fmin(__builtin_fmax(255.0 * c, 0.0), 255.0));
#else
  // Four operations instead of five: the product is rounded to binary64 FIRST, exactly as the reference's `255 * rgb[c]` is
  // (main.js:195) - values of the form k + 1/2 are common there (0.04 % of all channels: 255 * (j/255) / 2 ...) and the store's
  // round-half-to-even must see them as the ties they are - then clamped, and ONE addition onto 1.5*2^52 (whose ulp is 1) does
  // the rounding to nearest-even and leaves the byte in the sum's low mantissa word (v_rndne + v_cvt in one operation).
  // (Folding the multiplication into that addition as an fma would round the EXACT product instead: measured 1.7e-4 of all
  // channels off by one against 4e-7, profiles/r02_ab_log.md.)
  const double s = __builtin_fmin(__builtin_fmax(255.0 * c, 0.0), 255.0) + 6755399441055744.0;
  return (uint32_t)__builtin_bit_cast(unsigned long long, s);
#endif
}

// main.js:420-439, as a "candidate root" test.  With thc >= 0 (or NaN) the reference's two-armed
// root selection reduces to: cand = (t0 < eps) ? t1 : t0, and the sphere is hit at cand unless
// cand < eps (both roots behind the epsilon).  NaN fails every comparison, exactly as it loses
// `check.t < hit.t` / `t < light_len` in the reference.  inside = (t0 < eps), which for an
// accepted root equals the reference's (t0 < 0.001) || (t1 < 0.001) (main.js:445).
//
// Two forms of the line-sphere discriminant:
//   generic  (any origin p):     L = o - p, tca = d.L, d2 = L.L - tca^2, miss if d2 > r2   — the reference's own
//   anchored (uniform origin a): the host precomputes La = o - a and Ca = La.La - r2 per sphere; then
//            tca = d.La and r2 - d2 = tca^2 - Ca: 4 operations instead of 10.  Used (product kernel only)
//            for primary rays (a = camera) and for shadow rays walked FROM the light (a = light k), whose
//            line is the same line, so the same discriminant decides hit or miss.
// Both helpers must be called inside `if (hit)`; RT_PIN keeps the bookkeeping in that branch so that a
// wave whose 64 rays all miss pays one s_cbranch_execz and nothing else.
#define RT_PIN() asm volatile("")

// The scene tables walked by the wave-uniform loops live in global memory that nothing writes
// during the launch.  Typing them as CONSTANT address space (4) makes every uniform-index read an
// s_load into SGPRs by construction, whatever else is in the loop.
typedef const rt_geom __attribute__((address_space(4))) *geom_kptr;
typedef const rt_sphere __attribute__((address_space(4))) *sphere_kptr;

// Which pixel (or sample) a work-item owns.  Evaluated twice from the work-item id — before the ray is generated and
// again after the trace, behind an opaque copy of the id — so that px / lrow / valid are not kept live in VGPRs across
// the whole trace (they would be the 97th register: the kernel fits the 96 of 5 waves per SIMD without them).
struct rt_pixel { uint32_t px, trow, frow, lrow, sub, rows_valid, run, cand; bool valid, sky; };
// W1 - ONE-WAVE workgroups (the reflection-only many-sphere variants, rt_trace): the entry's 32 x 8 block is rendered by FOUR
// workgroups of one wave each, workgroup b = xcd + 8 * (wave + 4 * e') for entry e = 8 e' + xcd: the four waves of a block are
// consecutive workgroups of ONE XCD, and an XCD still reads one contiguous eighth of the table.
template <bool W1>
__device__ __forceinline__ uint32_t rt_entry_slot(const rt_launch &L) {
  // entry of workgroup b at (b % 8) * ceil(n / 8) + b / 8: workgroups are dealt round-robin over the 8 XCDs (speed only, never
  // correctness), so each XCD's L2 reads one contiguous eighth of the table instead of every line of it
  return (blockIdx.x & 7u) * L.order_n8 + (blockIdx.x >> (W1 ? 5 : 3));
}
// (the entry's index in the table's order: a compact band's block number)
template <bool W1>
__device__ __forceinline__ uint32_t rt_entry_index() { return W1 ? (((blockIdx.x >> 5) << 3) | (blockIdx.x & 7u)) : blockIdx.x; }
// (which of the block's four 8-pixel columns this wave renders)
template <bool W1>
__device__ __forceinline__ uint32_t rt_wave_of(uint32_t tid) { return W1 ? ((blockIdx.x >> 3) & 3u) : tid >> 6; }

#if RT_STRICT
template <bool SS2, bool W1 = false>
__device__ __forceinline__ rt_pixel rt_pixel_of(const rt_launch &L, uint32_t tid) {
  const uint32_t wave = tid >> 6, lane = tid & 63u;
  // grid = (tiles across the frame, tiles x row blocks per tile, frames of the batch).  y splits into
  // (tile, row block) with a shift when row blocks per tile is a power of two (the 16-row tiles of the
  // multi-GPU plan), trivially for a single tile (a whole frame), else with one wave-uniform division.
  const uint32_t tile_x = blockIdx.x, by = blockIdx.y;
  uint32_t tile_i, row_block;
  if (L.n_tiles == 1u) { tile_i = 0u; row_block = by; }
  else if (L.rb_shift != ~0u) { tile_i = by >> L.rb_shift; row_block = by & ((1u << L.rb_shift) - 1u); }
  else { tile_i = by / L.rb_per_tile; row_block = by - tile_i * L.rb_per_tile; }
  rt_pixel P;
  P.sub = 0u;                                          // trow = row inside tile `tile_i`
  if (!SS2) { P.px = tile_x * RT_TILE_W + wave * 8u + (lane & 7u); P.trow = row_block * RT_TILE_H + (lane >> 3); }
  else { const uint32_t q = lane >> 2; P.sub = lane & 3u; P.px = tile_x * RT_TILE_W + wave * 8u + (q & 7u); P.trow = row_block * 2u + (q >> 3); }
  P.frow = (L.tile_first + tile_i * L.tile_stride) * L.tile_rows + P.trow;   // frame row
  P.lrow = tile_i * L.tile_rows + P.trow;                                    // row in this call's output band
  P.valid = (P.px < L.w) && (P.trow < L.tile_rows) && (P.frow < L.h);
  P.rows_valid = 0u;                                   // (product kernel only)
  P.sky = false; P.run = 1u; P.cand = 0u;
  return P;
}
#else
// Product kernel: a FLAT grid (workgroups, 1, frames of the batch) and a launch table with one 16-byte entry per workgroup:
//   word 0 = tile_x | rows_valid << 11 | first frame row << 15      word 1 = first row in this call's output band | (run - 1) << 24 | sky << 31
//   word 2 = shadow masks                                            word 3 = primary candidates
// (built on the GPU per camera, frame size and tile set: rt_tables_gpu.hip, rt_block.h).  One scalar load replaces the tile /
// row-block arithmetic of the plain grid - no division, no tile parameters in registers - and decides the ORDER in which the
// hardware hands the tiles out: dearest first, so that a launch ends on cheap sky tiles instead of on the floor.  trow is the row
// inside the workgroup's block here.  (W1: rt_entry_slot above.)
template <bool SS2, bool W1 = false>
__device__ __forceinline__ rt_pixel rt_pixel_of(const rt_launch &L, uint32_t tid) {
  const uint32_t wave = rt_wave_of<W1>(tid), lane = tid & 63u;
  typedef uint32_t __attribute__((ext_vector_type(4))) rt_entry;                                     // 16 bytes (rt_tables.h: RT_ENTRY_WORDS)
  typedef const rt_entry __attribute__((address_space(4))) *order_kptr;
  const uint32_t slot = rt_entry_slot<W1>(L);
  const rt_entry e4 = *(order_kptr)((const char __attribute__((address_space(4))) *)L.order + ((size_t)slot << 4));   // s_load_dwordx4
  const uint32_t e0 = e4.x, e1 = e4.y;
  const uint32_t tile_x = e0 & 2047u, rows_valid = (e0 >> 11) & 15u, frow0 = e0 >> 15;
  rt_pixel P;
  P.sub = 0u;
  if (!SS2) { P.px = tile_x * RT_TILE_W + wave * 8u + (lane & 7u); P.trow = lane >> 3; }
  else { const uint32_t q = lane >> 2; P.sub = lane & 3u; P.px = tile_x * RT_TILE_W + wave * 8u + (q & 7u); P.trow = q >> 3; }
  P.frow = frow0 + P.trow;
  P.lrow = (e1 & 0xffffffu) + P.trow;
  P.sky = (e1 >> 31) != 0u;                            // workgroup-uniform: no sphere can show in these blocks (rt_block.h) ...
  P.run = ((e1 >> 24) & 127u) + 1u;                    // ... a run of this many 32-pixel blocks, starting at tile_x
  P.cand = e4.w;                                       // the (at most two) loop spheres the block's primary rays can meet (count << 16 | second << 8 | first), or 0: cull
  P.rows_valid = rows_valid;                           // wave-uniform: rows of the block inside its tile and the frame
  P.valid = (P.px < L.w) && (P.trow < rows_valid);
  return P;
}
#endif

#if !RT_STRICT
// Word 2 of this workgroup's launch-table entry: per light, the 16-bit set of loop-order spheres that can shadow a primary hit of
// its block (rt_block.h), or ~0u.  Read again where it is used - the primary node's lighting - instead of being kept in a
// scalar register across the cull and the search (the kernel has none to spare).
template <bool W1>
__device__ __forceinline__ uint32_t rt_entry_shadow_masks(const rt_launch &L) {
  const uint32_t slot = rt_entry_slot<W1>(L);
  return *(const uint32_t __attribute__((address_space(4))) *)((const char __attribute__((address_space(4))) *)L.order + ((size_t)slot << 4) + 8u);
}
#endif

// A frame of the explicit recursion stack: everything intersectWorld still needs after its
// recursive calls return (main.js:320-336) — the lighting and sampler terms do not depend on the
// children, so they are evaluated before descending.
template <bool REFRACT> struct frame;
template <> struct frame<false> { double amb[3], ds[3], a3; };
template <> struct frame<true>  { double amb[3], ds[3], a3, a4, h[3], f[3], re[3]; int has_f, phase; };
// A parked two-child node of the product general kernel: its own terms, its refraction ray, and the map F that
// was accumulated above it (S, O, LO, HI), to be restored when its reflection subtree has been evaluated.
struct park { double amb[3], ds[3], a3, a4, h[3], f[3], S, O[3], LO[3], HI[3]; uint32_t path, segs_left; int level, map_valid, hcode; };

// 64-bit table word `i` of a scalar-loaded bit-set table (shadow grids, bounce table: a few MB at most)
__device__ __forceinline__ unsigned long long rt_load_word32(const void *base, uint32_t i) {
  const char __attribute__((address_space(4))) *b = (const char __attribute__((address_space(4))) *)base;
  return *(const unsigned long long __attribute__((address_space(4))) *)(b + (i << 3));
}

__device__ __forceinline__ rt_geom rt_load_geom32(geom_kptr tab, uint32_t i) {
  const rt_geom __attribute__((address_space(4))) *g =
      (const rt_geom __attribute__((address_space(4))) *)((const char __attribute__((address_space(4))) *)tab + (i << 5));
  return rt_geom{g->ox, g->oy, g->oz, g->r2};
}

// two consecutive table records with ONE scalar load (s_load_dwordx16): one memory latency per two sphere tests
struct rt_geom_pair { rt_geom a, b; };
__device__ __forceinline__ rt_geom_pair rt_load_geom_pair32(geom_kptr tab, uint32_t i) {
  typedef double __attribute__((ext_vector_type(8))) d8;
  const d8 v = *(const d8 __attribute__((address_space(4))) *)((const char __attribute__((address_space(4))) *)tab + (i << 5));
  return rt_geom_pair{rt_geom{v[0], v[1], v[2], v[3]}, rt_geom{v[4], v[5], v[6], v[7]}};
}

#if !RT_STRICT
// The launch record as the COLD paths read it: straight from the kernarg segment at the point of use (the kernel's only argument lies
// at its start), behind an opaque copy of the pointer, so that a field only the rare paths need is not loaded at kernel entry and
// held in scalar registers across the whole trace (the kernel has none to spare).
__device__ __forceinline__ const rt_launch __attribute__((address_space(4))) *rt_cold_args() {
  const rt_launch __attribute__((address_space(4))) *K = (const rt_launch __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(K));
  return K;
}
// Append this work-item's sample to the launch's mark list (the cold end of the samplers' boundary test, a handful of samples per frame): entry = sample x |
// sample y << 20 | frame of the batch << 40; the counter of THIS launch is marks[marks_slot] (rt_api.hip alternates two, so that
// rt_retrace can clear the next launch's while it reads its own); beyond the list's capacity only the count grows and rt_retrace
// traces every sample of the launch.
template <bool SS2>
__device__ __forceinline__ void rt_mark_append(const rt_pixel &P) {
  const rt_launch __attribute__((address_space(4))) *K = rt_cold_args();
  if (!P.valid || (K->mark_flags & RT_MARK_NEVER)) return;
  const uint32_t sx = SS2 ? 2u * P.px + (P.sub & 1u) : P.px, sy = SS2 ? 2u * P.frow + (P.sub >> 1) : P.frow;
  uint32_t *const marks = K->marks;
  const uint32_t i = atomicAdd(marks + K->marks_slot, 1u);
  if (i < K->marks_cap) ((unsigned long long *)(marks + 4))[i] = (unsigned long long)sx | ((unsigned long long)sy << 20) | ((unsigned long long)blockIdx.z << 40);
}
#endif

#if !RT_STRICT
// Q of a hit from its parent's (see trace_pixel: "How far this kernel's own rounding has been magnified"): x = t / r, c = |d.n|
__device__ __forceinline__ float rt_q_of(float qp, float x, float c, float rp_over_r) {
  const float ic = __builtin_amdgcn_rcpf(fmaxf(c, 1e-30f));                    // (s taken as 1: no square root on the way)
  return fminf(ic * (qp * (6.f * x + rp_over_r) + x + 0.5f * x * x) + x, 1e30f);
}
#endif

template <bool REFRACT, bool COUNT, bool GRID, bool SS2, bool ITEM = false, bool W1 = false>
__device__ __forceinline__ void trace_pixel(const rt_launch &L, const rt_mtl *mtl, const rt_texture_desc *tex,
                                            [[maybe_unused]] double *acc, [[maybe_unused]] const rt_geom *cull_lds, [[maybe_unused]] const rt_geom cull0, [[maybe_unused]] uint32_t lane,
                                            [[maybe_unused]] double blk_x0, [[maybe_unused]] double blk_x1, [[maybe_unused]] double blk_y0,
                                            [[maybe_unused]] double blk_y1, v3 p, v3 d, double rgb[3], uint32_t cnt[3],
                                            [[maybe_unused]] bool is_probe, [[maybe_unused]] uint32_t cand_host,
                                            [[maybe_unused]] uint32_t own_sx = 0u, [[maybe_unused]] uint32_t own_sy = 0u) {
#ifdef RT_TESTING
  uint32_t probe_n = 0;                                  // test build: nodes of this sample's ray tree recorded so far
  double probe_li = 0.0;
#endif
  const sphere_kptr objs = (sphere_kptr)L.objects;
  const geom_kptr geom = (geom_kptr)L.geom;
  const uint32_t N = L.n_objects, NL = L.n_lights;
  const double eps = L.epsilon;
  const uint32_t NLOOP = L.n_loop;                      // spheres the per-ray loops walk: N, or N-1 with an enclosing sphere
  const uint32_t enc = L.enclosing;                      // device index of the enclosing sphere (== NLOOP), or ~0u
  // where the primary-ray cull's rectangles are: the product kernels know at compile time (the host launches the many-sphere
  // variant exactly for the scenes whose LDS image leaves them out), the strict and counting kernels ask the launch record
  const bool cull_lds_on = (!RT_STRICT && !COUNT) ? !GRID : (L.cull_in_lds != 0u);
#if RT_STRICT
  constexpr bool FOLD_FORWARD = false;
#else
  constexpr bool FOLD_FORWARD = true;       // the recursion is folded on the way down (see the descend step)
#endif
  [[maybe_unused]] frame<REFRACT> stack[FOLD_FORWARD ? 1 : RT_MAX_SEGS];
  // product general kernel: nodes with BOTH a reflection and a refraction child are parked here while their
  // reflection subtree is traced (everything else needs no stack)
  [[maybe_unused]] park parked[(FOLD_FORWARD && REFRACT) ? RT_MAX_SEGS : 1];
  [[maybe_unused]] int sp = 0;
  [[maybe_unused]] bool map_valid = false;             // false: the accumulated map F is the identity
  int level = 0;
#if !RT_STRICT
  // How far this kernel's own rounding has been MAGNIFIED on the way to the current hit: Q bounds the error of the hit's NORMAL in units
  // of 1.1e-16 (a float, rounded up: an estimate that only widens a tolerance).  A ray with origin error P and direction error D meets
  // a sphere of radius r after t at incidence cosine c (sine s): a sideways shift of the ray moves the hit along the surface by 1/c of
  // it, so the normal inherits (P + t D) / (r c); the distance itself, t = tca - thc, is rounded to ~eps t (1 + t / (2 r c)) - the
  // near root cancels when the ray grazes - and moves the hit along the ray, the normal by s / r of it; the mirrored (or refracted)
  // ray leaves with D' <= 3 D + 4 Q and P' = r Q.  With D dominated by the previous normal's error:
  //     Q_hit = Q_parent (6 t + r_parent) / (r c)  +  (t / r) (1 / c + s (1 + t / (2 r c))),        Q = 0 at the camera
  // (r_parent Q_parent is the origin's position error: from a small sphere onto a large one it all but vanishes; the parent's radius
  // rides along as one exponent byte, rounded up).
  // A primary hit on the floor has Q ~ 1e-2, on the reference's small spheres 1e1 - 1e3; every bounce off a sphere of radius r at
  // distance t multiplies it by ~6 t / (r c), a grazing one by far more (profiles/r04_ab_log.md section 4: the adversarial soak's
  // flipped pixel had ONE bounce).  The samplers' boundary test scales its tolerance by max(1, Q / RT_Q_FLAT): RT_XY_INDEX below;
  // RT_Q_FLAT is a third of the Q at which the flat tolerance (2e-13 in u, v = Q 1.1e-16 / 2 pi) is exactly the bound.  The scaled tolerance is honoured
  // up to the hot path's prefilter band (a fraction within 2^-20 of an integer): 36 x the flat tolerance at the largest admitted frequency
  // (2^17 per unit; beyond, the scene is a strict scene), 950 x at the reference's 5000 - a bound that routes every high-Q sample to the
  // cold block marks the reference's own scene's deep internal reflections by the dozen per frame (the recurrence is a worst case: inside a
  // sphere errors do not compound the way it assumes), profiles/r04_ab_log.md section 4.  Updated per
  // BOUNCE, not per node: it is the Q of the current hit at every node below the primary, and is filled in for the primary when it
  // spawns a ray (at the node's top, where the hit's distance is at hand).
  // (it lives in the upper half of `level` as a bfloat16, rounded up: a 97th vector register would cost the kernel a wave per SIMD)
#define RT_LVL(L_) ((L_) & 255)
#define RT_Q_GET(L_) __builtin_bit_cast(float, (uint32_t)(L_) & 0xffff0000u)
#define RT_Q_SET(L_, Q_) (L_) = (int)(((uint32_t)(L_) & 0xffffu) | ((__builtin_bit_cast(uint32_t, (float)(Q_)) + 0xffffu) & 0xffff0000u))
  // bits 8..15: the exponent byte of a power of two >= the radius of the sphere this hit lies on (the next hit's r_parent)
#define RT_R_GET(L_) __builtin_bit_cast(float, ((uint32_t)(L_) & 0xff00u) << 15)
#define RT_R_SET(L_, INVR_) (L_) = (int)(((uint32_t)(L_) & 0xffff00ffu) | ((((__builtin_bit_cast(uint32_t, __builtin_amdgcn_rcpf(INVR_)) >> 23) + 1u) & 255u) << 8))
#define RT_Q_FLAT 4096.f
#define RT_Q_OF(QP, T, INVR, C, RP) rt_q_of((QP), (T) * (INVR), (C), (RP) * (INVR))
#else
#define RT_LVL(L_) (L_)
#endif
  [[maybe_unused]] uint32_t tree_path = 1u;            // general kernel: position in the ray tree (root 1, reflect 2p, refract 2p+1)
#if defined(RT_TESTING) && defined(RT_ABLATE_BOUNCE)
  uint32_t segs_left = L.segs ? 1 : 0;
#else
  uint32_t segs_left = L.segs;
#endif
  double ret[3] = {0.0, 0.0, 0.0};

  // A3: closest hit.  The winner is kept as (ht, hcode) with hcode = 2*index + inside, so a candidate costs one
  // 64-bit and one 32-bit select.
  double ht = RT_INF; int hcode = -1;
      // One candidate: the sqrt and the bookkeeping stay inside the hit branch (RT_PIN).
#define RT_CAND(IDX, TCA, DISC)                                                               \
      if (!((DISC) < 0.0)) {                                                                  \
        RT_PIN();                                                                             \
        const double thc_ = rt_sqrt_nn(DISC);                                                 \
        const double t0_ = (TCA) - thc_, t1_ = (TCA) + thc_;                                  \
        const bool in_ = (t0_ < eps);                                                         \
        const double t_ = in_ ? t1_ : t0_;                                                    \
        const bool closer_ = (t_ < ht) && !(t_ < eps);   /* strict <: first wins */           \
        ht = closer_ ? t_ : ht;                                                               \
        hcode = closer_ ? ((int)(2u * (IDX)) + (in_ ? 1 : 0)) : hcode;                        \
      }
      // generic form, the reference's own (main.js:422-425): disc = r2 - d2
#define RT_GENERIC(IDX, G)                                                                    \
      {                                                                                       \
        const v3 Lv_ = mk((G).ox - p.x, (G).oy - p.y, (G).oz - p.z);                          \
        const double tca_ = dot(d, Lv_);                                                      \
        const double disc_ = (G).r2 - (dot(Lv_, Lv_) - tca_ * tca_);                          \
        RT_CAND(IDX, tca_, disc_)                                                             \
      }
      // anchored form (origin = camera): disc = tca^2 - Ca
#define RT_ANCHORED(IDX, G)                                                                   \
      {                                                                                       \
        const double tca_ = d.x * (G).ox + d.y * (G).oy + d.z * (G).oz;                       \
        const double disc_ = __builtin_fma(tca_, tca_, -(G).r2);                              \
        RT_CAND(IDX, tca_, disc_)                                                             \
      }
      // 32-bit byte offset (at most 256 spheres x 32 bytes, times at most 16 lights in the light-anchored table): base +
      // zext(offset) lets the scalar load take its offset from an SGPR (s_load_dwordx8 s[..], s[base], s_off) instead of
      // a 64-bit address computation per load (+0.8 % on the headline)
#define RT_LOAD(TAB, I) rt_load_geom32((TAB), (uint32_t)(I))
#define RT_LOAD_PAIR(TAB, I) rt_load_geom_pair32((TAB), (uint32_t)(I))
      // Both loops are unrolled by two by hand (the pinned branches make them convergent, which rules out
      // the compiler's runtime unrolling); a pair's two records come with ONE s_load_dwordx16 (rt_load_geom_pair32).
  if (segs_left != 0) {
        // Primary rays.  First a wave-wide cull: lane j compares sphere j's conservative screen rectangle
        // (host, resolution-independent: bounds of X/D and Y/D over the pixels whose LINE meets the sphere)
        // with the rectangle of this wave's 8x8 pixel block; __ballot turns the 64 verdicts into one scalar
        // mask and only the surviving spheres are tested, in index order (the tie-break is preserved).
        // A wave of sky pixels tests nothing; a wave of floor pixels tests the floor.  The cull only prunes, so the strict
        // kernel uses it too (with the reference's own discriminant for the survivors) and stays bit-identical.
        [[maybe_unused]] const geom_kptr ga = (geom_kptr)L.geom_cam;
#if !RT_STRICT
        // A block for which the table names at most two spheres its primary rays can meet at all (word 3 of its entry;
        // a floor block names the floor) tests those and skips the cull.
        if (cand_host != 0u) {                           // count << 16 | second << 8 | first (loop indices, ascending)
          { const uint32_t i = cand_host & 255u; const rt_geom g0 = RT_LOAD(ga, i); RT_ANCHORED(i, g0) }
          if (cand_host >= (2u << 16)) { const uint32_t i = (cand_host >> 8) & 255u; const rt_geom g0 = RT_LOAD(ga, i); RT_ANCHORED(i, g0) }
        } else
#endif
        for (uint32_t base = 0; base < NLOOP; base += 64u) {
          const uint32_t j = base + lane;
          // {x_lo, x_hi, y_lo, y_hi} in units of 1/D.  Few spheres: from the LDS image.  Many: one record per lane from HBM (L2) -
          // the first 64 were fetched before the ray was generated (cull0), scenes of more spheres fetch the rest here
          const uint32_t jj = j < NLOOP ? j : 0u;
          double c0 = cull0.ox, c1 = cull0.oy, c2 = cull0.oz, c3 = cull0.r2;
          // (explicit address spaces: the compiler otherwise selects the POINTER and issues one flat load for both cases)
          if constexpr (ITEM) {
          } else if (cull_lds_on) {
            const rt_geom __attribute__((address_space(3))) *g = (const rt_geom __attribute__((address_space(3))) *)cull_lds + jj;
            c0 = g->ox; c1 = g->oy; c2 = g->oz; c3 = g->r2;
          } else if (base != 0u) {
            const rt_geom __attribute__((address_space(1))) *g = (const rt_geom __attribute__((address_space(1))) *)L.cull + jj;
            c0 = g->ox; c1 = g->oy; c2 = g->oz; c3 = g->r2;
          }
          const rt_geom cr = rt_geom{c0, c1, c2, c3};
          // five compares, their 64-bit masks combined on the scalar unit (as one boolean expression the compiler may build
          // the conjunction in vector registers instead: ~15 more vector instructions per wave in the many-sphere variant)
          unsigned long long m = __ballot(j < NLOOP) & __ballot(cr.ox * L.proj_d <= blk_x1) & __ballot(cr.oy * L.proj_d >= blk_x0) &
                                 __ballot(cr.oz * L.proj_d <= blk_y1) & __ballot(cr.r2 * L.proj_d >= blk_y0);
          // (rt_retrace: a wave's lanes hold unrelated samples and only some of them run - no wave-wide cull, every sphere in scene order)
          if constexpr (ITEM) m = (NLOOP - base >= 64u) ? ~0ull : ((1ull << (NLOOP - base)) - 1ull);
          while (m) {
            const uint32_t i = base + (uint32_t)__builtin_ctzll(m);
            m &= m - 1ull;
#if RT_STRICT
            const rt_geom g0 = RT_LOAD(geom, i);
            RT_GENERIC(i, g0)
#else
            const rt_geom g0 = RT_LOAD(ga, i);
            RT_ANCHORED(i, g0)
#endif
          }
        }
  }
  bool searched = true;                // the primary ray's candidates were found above (culled; camera-anchored in the product kernel)
#if !RT_STRICT
  // A wave none of whose primary rays met a sphere of the loops, in a scene whose enclosing sphere is flat AND constant in colour
  // (the reference's skybox with a plain colour): every pixel of the wave is that sphere's ambient term, max(color*albedo[0],
  // min(1, color*0 + color*0)) (main.js:326-336 with no light and no child), which the host evaluated once.  Nothing else runs.
  if (L.sky_fast && segs_left != 0 && __ballot(hcode >= 0) == 0ull) {
    if (COUNT) { cnt[0]++; cnt[2] += N; }
    rgb[0] = L.sky_rgb[0]; rgb[1] = L.sky_rgb[1]; rgb[2] = L.sky_rgb[2];
    return;
  }
#endif

  if (segs_left != 0) {
    for (;;) {
      // ---------------- evaluate one intersectWorld node (segs_left > 0 here) ----------------
      if (COUNT) cnt[0]++;
      if (!searched) {                              // reflection / refraction rays: any origin, generic form
        bool scanned = false;
#if !RT_STRICT
        if constexpr (GRID) {
          if (rt_cold_args()->bounce_table != nullptr) {
            // Many spheres: a bounced ray starts ON the sphere it just hit (`hcode` still names it) and its direction
            // falls in one cell of a cube map.  The host stored, per (sphere, cell), the bit set of the spheres that
            // ANY ray leaving that sphere's ball in ANY direction of that cell can meet (conservative: angle between
            // the cell and the line of centres against asin((r_i + r_j) / distance), rt_tables.cpp build_bounce_table).
            // The wave tests the UNION over its active lanes, walked like the shadow grid's cells (readlane + ballot,
            // correct under divergence), in index order, so the strict-< tie-break of the full scan is kept.
            const uint32_t from = (uint32_t)(hcode >> 1);
            const double ax = __builtin_fabs(d.x), ay = __builtin_fabs(d.y), az = __builtin_fabs(d.z);
            const bool bx = (ax >= ay) && (ax >= az), by = !bx && (ay >= az);
            const double dm = bx ? d.x : (by ? d.y : d.z);
            const double du = bx ? d.y : d.x, dv = (bx || by) ? d.z : d.y;
            const double sc = (0.5 * RT_BGRID) * __builtin_amdgcn_rcp(__builtin_fabs(dm));   // 2^-24 is plenty: the host's cells overlap by 1e-6
            const double fu = __builtin_fmin(__builtin_fmax(__builtin_fma(du, sc, 0.5 * RT_BGRID), 0.0), (double)(RT_BGRID - 1u));
            const double fv = __builtin_fmin(__builtin_fmax(__builtin_fma(dv, sc, 0.5 * RT_BGRID), 0.0), (double)(RT_BGRID - 1u));
            const uint32_t face = (bx ? 0u : (by ? 2u : 4u)) + ((dm < 0.0) ? 1u : 0u);
            const uint32_t key = from * RT_BCELLS + face * (RT_BGRID * RT_BGRID) + (uint32_t)fv * RT_BGRID + (uint32_t)fu;
            const uint32_t words = (NLOOP + 63u) >> 6;
            ht = RT_INF; hcode = -1;
            for (uint32_t wd = 0; wd < words; wd++) {
              unsigned long long cand = 0ull, todo = __ballot(true);
              uint32_t distinct = 0;
              while (todo) {
                const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)__builtin_ctzll(todo));
                cand |= rt_load_word32(rt_cold_args()->bounce_table, k0 * words + wd);
                todo &= ~__ballot(key == k0);
                if (++distinct == 16u && todo) {           // a wave whose rays fan out over many cells: scan everything
                  cand = (wd + 1u == words && (NLOOP & 63u)) ? ((1ull << (NLOOP & 63u)) - 1ull) : ~0ull;
                  break;
                }
              }
              while (cand) {
                const uint32_t j = (wd << 6) + (uint32_t)__builtin_ctzll(cand);
                cand &= cand - 1ull;
                const rt_geom g0 = RT_LOAD(geom, j);
                RT_GENERIC(j, g0)
              }
            }
            scanned = true;
          }
        }
#endif
        if (!scanned) {
          ht = RT_INF; hcode = -1;
          uint32_t i = 0;
          for (; i + 2 <= NLOOP; i += 2) {
            const rt_geom_pair gp = RT_LOAD_PAIR(geom, i);
            const rt_geom g0 = gp.a, g1 = gp.b;
            RT_GENERIC(i, g0) RT_GENERIC(i + 1, g1)
          }
          if (i < NLOOP) { const rt_geom g0 = RT_LOAD(geom, i); RT_GENERIC(i, g0) }
        }
      }
      [[maybe_unused]] const bool primary_node = searched;       // wave-uniform: this node is the primary ray's
      searched = false;
      // The enclosing sphere (every other sphere, light and the camera strictly inside it: a skybox) is
      // kept LAST in the device tables and outside the loops above: it can only be the closest hit of a
      // ray that hits nothing else.  Only the lanes still without a hit evaluate it.
      if (enc != ~0u && hcode < 0) {
        if (L.enclosing_flat) {
          // ... and when that sphere is flat - no lighting, no children, a colour that does not depend on the hit point (the
          // reference's skybox: albedo [1,0,0,0,0], main.js:124) - WHERE the ray meets it does not matter: a ray that starts
          // strictly inside it always does (main.js:429-439 returns t1 > 0.001), so the test is not evaluated at all
          ht = 1.0; hcode = (int)(2u * enc + 1u);
        } else {
          const rt_geom g0 = RT_LOAD(geom, enc);
          RT_GENERIC(enc, g0)
        }
      }
#undef RT_ANCHORED
#undef RT_GENERIC
#undef RT_CAND
      if (COUNT) cnt[2] += N;
      const int hi = hcode >> 1;
      const bool inside = (hcode & 1) != 0;
      bool descend = false;
#if defined(RT_TESTING) && defined(RT_ABLATE_SHADE)
      if (true) { ret[0] = ht; ret[1] = (double)hcode; ret[2] = 0.0; } else
#endif
      if (hcode < 0) {                                // main.js:231 (with a flat sky of constant colour: that sky's pixel term, see rt_api.hip bind_kernel)
#if RT_STRICT
        ret[0] = L.miss_color[0]; ret[1] = L.miss_color[1]; ret[2] = L.miss_color[2];
#else
        // (read where it is used, from the kernarg segment: six scalar registers less across the whole loop)
        { const rt_launch __attribute__((address_space(4))) *K = rt_cold_args(); ret[0] = K->miss_color[0]; ret[1] = K->miss_color[1]; ret[2] = K->miss_color[2]; }
#endif
#ifdef RT_TESTING
        if (is_probe && probe_n < RT_PROBE_NODES) {
          double *q = L.probe + (size_t)(probe_n++) * RT_PROBE_WORDS;
          for (uint32_t z = 0; z < RT_PROBE_WORDS; z++) q[z] = 0.0;
          q[0] = (double)(REFRACT ? tree_path : (1u << RT_LVL(level))); q[1] = -1.0; q[2] = ht; q[9] = d.x; q[10] = d.y; q[11] = d.z;
          q[17] = (double)segs_left; q[19] = p.x; q[20] = p.y; q[21] = p.z; q[23] = 1.0;
        }
#endif
      } else {
        const rt_mtl &m = *(const rt_mtl *)((const char *)mtl + (uint32_t)hi * (uint32_t)sizeof(rt_mtl));      // per-lane index, a 32-bit offset (LDS; the many-sphere variant: HBM / L2)
        // A2 ext part for the closest hit only (main.js:440-447; pure, so deferring it is exact)
        const v3 h = mk(p.x + d.x * ht, p.y + d.y * ht, p.z + d.z * ht);
#if RT_STRICT
        double nlen;
        const v3 n = unit(mk(h.x - m.origin[0], h.y - m.origin[1], h.z - m.origin[2]), &nlen);
#else
        // the hit point lies on the sphere, so |h - o| is r up to the rounding of h: scale by the stored 1/r
        const double inv_r = m.inv_r;
        const v3 n = mk((h.x - m.origin[0]) * inv_r, (h.y - m.origin[1]) * inv_r, (h.z - m.origin[2]) * inv_r);
#endif
        const v3 l = inside ? mk(-n.x, -n.y, -n.z) : n;                 // hit.l, quirk q5
        const double a0 = m.albedo[0], a1 = m.albedo[1], a2 = m.albedo[2], a3 = m.albedo[3];
        const double a4 = REFRACT ? m.albedo[4] : 0.0;
#if !RT_STRICT && !defined(RT_ABLATE_QAMP)     /* (RT_ABLATE_QAMP: timing experiment, profiles/ab_build.sh) */
        // Q of this hit (see above): for a bounced ray's hit, and for a primary hit that will spawn a ray (ht is at hand here)
        if (RT_LVL(level) != 0 || ((a3 > 0.0 || a4 > 0.0) && segs_left > 1)) {
          const float ir_ = (float)inv_r;
          const float q_ = RT_Q_OF(RT_Q_GET(level), (float)ht, ir_, __builtin_fabsf((float)dot(d, n)), RT_R_GET(level));
          RT_Q_SET(level, q_);
          RT_R_SET(level, ir_);
        }
#endif

        // A8 sampler (main.js:320).  Pure, so it is evaluated here, before the lighting, where few values
        // are live: the OCML atan2/asin bodies are the register-pressure peak of the kernel.
        double col[3];
#if defined(RT_TESTING) && defined(RT_ABLATE_SAMPLER)   /* timing experiments only (profiles/ab_build.sh); never defined in the product build */
        const int kind = RT_SAMPLER_COLOR;
#else
        const int kind = m.sampler_kind;
#endif
#if RT_STRICT
        if (kind == RT_SAMPLER_TEXTURE) {
          double t_at, t_as;
          rt_atan2_asin(-n.z, -n.x, -n.y, &t_at, &t_as);
          const double u = RT_DIV_CONST(t_at, M_PI) / 2.0 + 0.5;   // main.js:446 (q6: two divisions)
          const double v = RT_DIV_CONST(t_as, M_PI / 2.0) / 2.0 + 0.5;  // main.js:447
          const rt_texture_desc td = tex[m.texture];
          const double xd = ceil(u * (double)td.width) - 1.0, yd = ceil(v * (double)td.height) - 1.0;
          uint32_t xi = (xd > 0.0) ? (uint32_t)xd : 0u, yi = (yd > 0.0) ? (uint32_t)yd : 0u;
          xi = min(xi, td.width - 1u); yi = min(yi, td.height - 1u);   // memory safety only; u,v <= 1
          const uint32_t texel = *(const uint32_t *)(L.texel_base + td.texels_offset + ((size_t)yi * td.width + xi) * 4u);
          col[0] = RT_DIV_CONST((double)(texel & 255u), 255.0); col[1] = RT_DIV_CONST((double)((texel >> 8) & 255u), 255.0);
          col[2] = RT_DIV_CONST((double)((texel >> 16) & 255u), 255.0);
          if (xd != xd || yd != yd) col[0] = col[1] = col[2] = __builtin_nan("");   // texels[NaN] is undefined in JS
        } else if (kind == RT_SAMPLER_CHECKER) {
          double t_at, t_as;
          rt_atan2_asin(-n.y, -n.x, -n.z, &t_at, &t_as);
          const double u = RT_DIV_CONST(t_at, M_PI) / 2.0 + 0.5;   // main.js:127 (its own axes)
          const double v = RT_DIV_CONST(t_as, M_PI / 2.0) / 2.0 + 0.5;  // main.js:128
          const int c = to_int32_bit0(u * m.c[6]) ^ to_int32_bit0(v * m.c[7]);
          col[0] = m.c[3 * c]; col[1] = m.c[3 * c + 1]; col[2] = m.c[3 * c + 2];
#else
        // ---- Texture (main.js:143-145, 343-351, u, v of :446-447) and sphere-checker (main.js:126-133, its own u, v), and the
        // boundary marks.
        // A sampler coordinate x = u * frequency decides a texel (main.js:344-347) or a checker parity (main.js:129-130) by its integer
        // part, and this kernel's u, v differ from the reference's in their last bits (its hit point and normal do).  A sample with a
        // coordinate within L.flag_tol (RT_FLAG_T1 x the scene's largest sampler frequency: 1e-9 for the reference's checker) of an
        // integer is decided in the reference by the last bits of ITS arithmetic: it is appended to the launch's mark list and
        // traced again, operation for operation, by the strict build's rt_retrace (rt_api.hip).  The test on the hot path is integer work on the bits of x + 1.5 * 2^32, a sum whose ulp
        // is 2^-20: its mantissa holds floor(x) (from bit 20 up) - the texel index, the checker parity - and 20 fraction bits;
        // "fraction within 2^-20 of 0 or 1" (6e-6 of the hits) sends the sample to the precise test, which also takes floor(x)
        // again (the sum rounds a fraction above 1 - 2^-21 up).  (Checker frequencies outside [0, 2^31), where the sum does not hold
        // ToInt32's parity, make the scene a strict-kernel scene: rt_api.hip.)
        // RT_XY_INDEX: iu, iv = floor(xu), floor(xv) and the boundary mark
#define RT_XY_INDEX(XU, XV, FU, FV)                                                                                \
          const unsigned long long su = __builtin_bit_cast(unsigned long long, (XU) + 6442450944.0), sv = __builtin_bit_cast(unsigned long long, (XV) + 6442450944.0);   \
          uint32_t iu = __builtin_amdgcn_alignbit((uint32_t)(su >> 32), (uint32_t)su, 20u) ^ 0x80000000u;       /* floor(x) for x in [0, 2^31) ... */ \
          uint32_t iv = __builtin_amdgcn_alignbit((uint32_t)(sv >> 32), (uint32_t)sv, 20u) ^ 0x80000000u;       \
          /* ... unless the fraction is within 2^-20 of an integer <=> the 20 fraction bits are 0xfffff, 0 or 1 (NaN, infinity: 0) */ \
          if ((min(((uint32_t)su + 1u) & 0xfffffu, ((uint32_t)sv + 1u) & 0xfffffu) <= 2u)) {           \
            RT_PIN();                                                                                             \
            iu = (uint32_t)(XU); iv = (uint32_t)(XV);                  /* truncation = floor (x >= 0); NaN -> 0 */  \
            const rt_launch __attribute__((address_space(4))) *K = rt_cold_args();                                \
            /* the tolerance grows with what this hit's normal error has been magnified by (qamp; a primary hit's Q from the camera) */ \
            float q_here = RT_Q_GET(level);                                                                        \
            if (RT_LVL(level) == 0 && q_here == 0.f) {                  /* (a primary hit that spawns nothing: not filled in above) */ \
              const float ex = (float)(h.x - K->cam_origin[0]), ey = (float)(h.y - K->cam_origin[1]), ez = (float)(h.z - K->cam_origin[2]);   \
              q_here = RT_Q_OF(0.f, __builtin_sqrtf(ex * ex + ey * ey + ez * ez), (float)m.inv_r, __builtin_fabsf((float)dot(d, n)), 0.f);       \
            }                                                                                                     \
            double tol = (K->mark_flags & RT_MARK_ALL) ? 2.0 : K->flag_tol * (double)fmaxf(1.f, q_here * (1.f / RT_Q_FLAT));   \
            /* ... and a sample whose colour cannot move the pixel by a byte is left alone: the pixel is F(x) = max(LO, min(HI, O + S x)) \
               of this node's colour x (and of the parked nodes' maps above it), |dF| <= S |dx|, and a flipped texel / parity moves x by \
               at most 2 when every albedo and colour of the scene lies in [0, 1] (RT_MARK_WEIGHT: the host's check) */              \
            if ((K->mark_flags & (RT_MARK_WEIGHT | RT_MARK_ALL)) == RT_MARK_WEIGHT) {                              \
              double S_ = map_valid ? acc[0] : 1.0;                                                               \
              if constexpr (FOLD_FORWARD && REFRACT) for (int i_ = 0; i_ < sp; i_++) S_ *= (parked[i_].map_valid ? parked[i_].S : 1.0) * parked[i_].a3;   \
              if (__builtin_fabs(S_) * 510.0 < 0.9) tol = -1.0;                                                    \
            }                                                                                                     \
            /* (a frequency of exactly 0 - stripes - makes the coordinate exactly 0 on every hit: it carries no error and decides nothing) */ \
            const bool zf = !(K->mark_flags & RT_MARK_ZERO);                                                       \
            const bool bu = (zf && (FU) == 0.0 && (XU) == 0.0) || (__builtin_fabs((XU) - __builtin_rint(XU)) >= tol);    \
            const bool bv = (zf && (FV) == 0.0 && (XV) == 0.0) || (__builtin_fabs((XV) - __builtin_rint(XV)) >= tol);    \
            if (!(bu & bv)) {                                                                                     /* NaN: marked */ \
              uint32_t t3 = threadIdx.x;                                                                          \
              asm volatile("" : "+v"(t3));                                                                        \
              rt_mark_append<SS2>(rt_pixel_of<SS2, W1>(L, t3));                                                        \
            }                                                                                                     \
          }
        if (kind == RT_SAMPLER_TEXTURE) {
          double t_at, t_as;
          rt_atan2_asin(-n.z, -n.x, -n.y, &t_at, &t_as);
          const double u = RT_DIV_CONST(t_at, M_PI) / 2.0 + 0.5;   // main.js:446 (q6: two divisions)
          const double v = RT_DIV_CONST(t_as, M_PI / 2.0) / 2.0 + 0.5;  // main.js:447
          const rt_texture_desc td = tex[m.texture];
          const double xu = u * (double)td.width, xv = v * (double)td.height;
          // max(0, ceil(x) - 1) (main.js:344-345) is floor(x) for every x >= 0 that is not an integer, and the integers are marked:
          // the index comes out of the fixed-point sum (u, v in [0, 1]; widths and heights <= 16384)
          RT_XY_INDEX(xu, xv, 1.0, 1.0)
          const uint32_t xi = min(iu, td.width - 1u), yi = min(iv, td.height - 1u);   // memory safety only; u,v <= 1
          const uint32_t texel = *(const uint32_t *)(rt_cold_args()->texel_base + td.texels_offset + ((size_t)yi * td.width + xi) * 4u);
          col[0] = RT_DIV_CONST((double)(texel & 255u), 255.0); col[1] = RT_DIV_CONST((double)((texel >> 8) & 255u), 255.0);
          col[2] = RT_DIV_CONST((double)((texel >> 16) & 255u), 255.0);
        } else if (kind == RT_SAMPLER_CHECKER) {
          double t_at, t_as;
          rt_atan2_asin(-n.y, -n.x, -n.z, &t_at, &t_as);
          const double u = RT_DIV_CONST(t_at, M_PI) / 2.0 + 0.5;   // main.js:127 (its own axes)
          const double v = RT_DIV_CONST(t_as, M_PI / 2.0) / 2.0 + 0.5;  // main.js:128
          const double xu = u * m.c[6], xv = v * m.c[7];
          RT_XY_INDEX(xu, xv, m.c[6], m.c[7])
          const int c = (int)((iu ^ iv) & 1u);                       // the parity of floor(x) = ToInt32(x) & 1 for x in [0, 2^31)
          col[0] = m.c[3 * c]; col[1] = m.c[3 * c + 1]; col[2] = m.c[3 * c + 2];
#undef RT_XY_INDEX
#endif
        } else if (kind == RT_SAMPLER_STARS) {
          // the sample's index in the FRAME (not in this call's tiles), recomputed from the work-item id so that it
          // costs no register outside this branch; `path` is the node's position in the ray tree
          uint32_t sx = own_sx, sy = own_sy;         // rt_retrace hands the sample over
          if constexpr (!ITEM) {
            uint32_t t3 = threadIdx.x;
            asm volatile("" : "+v"(t3));
            const rt_pixel P = rt_pixel_of<SS2, W1>(L, t3);
            sx = SS2 ? 2u * P.px + (P.sub & 1u) : P.px; sy = SS2 ? 2u * P.frow + (P.sub >> 1) : P.frow;
          }
          const unsigned long long pix = (unsigned long long)sy * (SS2 ? 2u * L.w : L.w) + sx;
          const uint32_t path = REFRACT ? tree_path : (1u << RT_LVL(level));
          double c = star_uniform((uint32_t)pix, (uint32_t)(pix >> 32), path);
          c = (c >= m.c[6]) ? 0.0 : c * m.c[7];     // main.js:137-138
          col[0] = col[1] = col[2] = c;
        } else { col[0] = m.c[0]; col[1] = m.c[1]; col[2] = m.c[2]; }

        // general product kernel: the sampled colour waits in LDS (slots 10-12 of the lane's fold state) while the
        // lights are scanned: six registers fewer across the hottest loop, which is what lets this kernel fit
        // 96 VGPRs = 5 waves per SIMD
        if constexpr (FOLD_FORWARD && REFRACT) {
#pragma unroll
          for (int c = 0; c < 3; c++) acc[(10 + c) * RT_WG_THREADS] = col[c];
        }

        // A7 lighting and shadows
        double diffuse = 0.0, specular = 0.0;
#if defined(RT_TESTING) && defined(RT_ABLATE_LIGHT)
        if (false) {
#else
        if (a1 > 0.0 || a2 > 0.0) {
#endif
#if RT_STRICT
          double li = L.light_intensity;                               // shared across lights (q2)
#else
          double li = rt_cold_args()->light_intensity;                 // shared across lights (q2); read where it is used (two scalar registers less across the loop)
#endif
#if !RT_STRICT
          [[maybe_unused]] uint32_t smask = ~0u;
          if constexpr (!COUNT) { if (primary_node) smask = rt_entry_shadow_masks<W1>(L); }
#endif
          for (uint32_t k = 0; k < NL; k++) {
            double llen;
            // light k from the kernarg segment through a 32-bit byte offset (scalar load with an SGPR offset)
            const double *lk = (const double *)((const char *)&L.lights[0][0] + (uint32_t)(k * 24u));
#if !RT_STRICT
            // few spheres (no shadow grid): the scan's first two records are fetched HERE, with the light's position - their latency
            // hides behind the light vector's normalisation instead of standing in front of the scan (a wave whose lanes all
            // face away wasted one load); measured -0.3 % on the headline, and +0.4 % where the grid path made it a wasted load
            [[maybe_unused]] rt_geom_pair gp_first;
            if constexpr (!GRID && !COUNT) {
              [[maybe_unused]] const geom_kptr gl = (geom_kptr)L.geom_light;
              gp_first = RT_LOAD_PAIR(gl, k * L.n_objects);
            }
#endif
            const v3 sraw = mk(lk[0] - h.x, lk[1] - h.y, lk[2] - h.z);
            const double lmag = dot(sraw, sraw);
#if RT_STRICT
            const v3 sv = unit(sraw, &llen);
#else
            const double inv_llen = rt_rsqrt_pos(lmag);               // lights never coincide with a surface point
            llen = lmag * inv_llen;
            const v3 sv = mk(sraw.x * inv_llen, sraw.y * inv_llen, sraw.z * inv_llen);
#endif
            const double sdot = dot(sv, l);
#if RT_STRICT
            if (sdot <= 0.0) continue;                                 // surface faces away (main.js:292)
#else
            if (!(sdot > 0.0)) continue;                               // the same; and a light AT the hit point: lmag == 0 makes sdot 0 there, NaN here
#endif
            if (COUNT) cnt[1]++;
            // Shadow scan (main.js:293-304) over every sphere but the one just hit (q3).  A fully blocked lane
            // keeps li == 0 whatever follows, so leaving the loop is a pure shortcut, taken per pair.
            uint32_t tests = 0;
            bool blocked = false;
#if !RT_STRICT
            // walked from the light: origin = light k (uniform), direction = -sv, the hit point is at llen
            const geom_kptr gl = (geom_kptr)L.geom_light;
            const uint32_t glo = k * L.n_objects;                      // light k's table: a 32-bit index offset (RT_LOAD)
#define RT_SDISC(G, TC, DISC)                                                                 \
            const double TC = -(sv.x * (G).ox + sv.y * (G).oy + sv.z * (G).oz);               \
            const double DISC = __builtin_fma(TC, TC, -(G).r2);
#define RT_SROOTS(TC, THC, T0, T1) const double T0 = llen - (TC + THC), T1 = llen - (TC - THC);
#else
            const geom_kptr gl = geom;
            const uint32_t glo = 0u;
#define RT_SDISC(G, TC, DISC)                                                                 \
            const v3 Lv_ = mk((G).ox - h.x, (G).oy - h.y, (G).oz - h.z);                      \
            const double TC = dot(sv, Lv_);                                                   \
            const double DISC = (G).r2 - (dot(Lv_, Lv_) - TC * TC);
#define RT_SROOTS(TC, THC, T0, T1) const double T0 = TC - THC, T1 = TC + THC;
#endif
#define RT_SHADOW(J, G)                                                                       \
            {                                                                                 \
              const bool other_ = ((int)(J) != hi);                                           \
              if (COUNT && other_ && !blocked) tests++;                                       \
              RT_SDISC(G, tc_, disc_)                                                         \
              if (other_ && !(disc_ < 0.0) && !blocked) {                                     \
                RT_PIN();                                                                     \
                const double thc_ = rt_sqrt_nn(disc_);                                        \
                RT_SROOTS(tc_, thc_, t0_, t1_)                                                \
                const double t_ = (t0_ < eps) ? t1_ : t0_;                                    \
                if ((t_ < llen) && !(t_ < eps)) {                                             \
                  RT_PIN();                                                                   \
                  const double oa4_ = objs[J].albedo[4];                                      \
                  if (oa4_ != 0.0) li = rt_div(li, oa4_);   /* transparent occluder brightens (q2) */ \
                  else { li = 0.0; blocked = true; }                                          \
                }                                                                             \
              }                                                                               \
            }
// The scans the product and strict kernels run (everything but the counting variant): no per-lane `break`.  A lane that
// is already blocked (li == 0) keeps testing, and whatever it hits leaves li at 0 (0 / albedo, or 0), exactly where the
// reference's `break` (main.js:301) left it.  The loop is then wave-uniform: the exec-mask bookkeeping of a divergent loop
// exit - about 10 scalar instructions per iteration, for every wave - is gone (measured: +4.6 % on the headline).
#define RT_SHADOW_U(J, G)                                                                     \
            {                                                                               \
              RT_SDISC(G, tc_, disc_)                                                       \
              if (((int)(J) != hi) && !(disc_ < 0.0)) {                                     \
                RT_PIN();                                                                   \
                const double thc_ = rt_sqrt_nn(disc_);                                      \
                RT_SROOTS(tc_, thc_, t0_, t1_)                                              \
                const double t_ = (t0_ < eps) ? t1_ : t0_;                                  \
                if ((t_ < llen) && !(t_ < eps) && li != 0.0) {                              \
                  RT_PIN();                                                                 \
                  const double oa4_ = objs[J].albedo[4];                                    \
                  li = (oa4_ != 0.0) ? rt_div(li, oa4_) : 0.0;                              \
                }                                                                           \
              }                                                                             \
            }
#if defined(RT_TESTING) && defined(RT_ABLATE_SHADOW)
            const uint32_t NS = 0;
#elif defined(RT_TESTING) && defined(RT_ABLATE_SHADOW4)
            const uint32_t NS = NLOOP > 4u ? NLOOP - 4u : NLOOP;      // timing only: what skipping four tests per light would be worth
#else
            const uint32_t NS = NLOOP;
#endif
#if !RT_STRICT
            // Primary hits of a block whose table entry says that NO sphere can stand between the block's hit points and light k
            // (rt_block.h, shadow masks; most floor blocks): neither grid nor scan.
            bool no_occluder = false;
            if constexpr (!COUNT) no_occluder = primary_node && k < 2u && ((smask >> (16u * k)) & 0xffffu) == 0u;
            if (no_occluder) {
            } else
            if (GRID && rt_cold_args()->shadow_grid != nullptr && li != 0.0) {
              // Many spheres: cull the scan with the light's grid.  The host cut light k's view of the scene
              // (projective coordinates x'/z', y'/z' in a frame looking from the light at the scene) into
              // RT_SGRID x RT_SGRID cells and stored, per cell, the bit set of spheres whose conservative rectangle
              // (same construction as the primary-ray cull, with the light as the eye) touches it.  A lane's shadow
              // ray lies on the line from the light through its hit point, so only the spheres of that point's cell
              // can block it.  The wave tests the UNION over its active lanes: the distinct cells are walked with
              // readlane/ballot (correct under divergence: it never relies on inactive lanes), typically 1-4 of them.
              const void *const sgrid = rt_cold_args()->shadow_grid;     // (read where it is used: the many-sphere kernels have no scalar register to spare)
              const double __attribute__((address_space(4))) *gh = (const double __attribute__((address_space(4))) *)sgrid + 16u * k;
              const v3 vv = mk(-sraw.x, -sraw.y, -sraw.z);                                   // light -> hit point
              const double vx = gh[0] * vv.x + gh[1] * vv.y + gh[2] * vv.z, vy = gh[3] * vv.x + gh[4] * vv.y + gh[5] * vv.z;
              const double vz = gh[6] * vv.x + gh[7] * vv.y + gh[8] * vv.z;
              const double iz = rt_rcp(vz);
              const double fx = __builtin_fmin(__builtin_fmax((vx * iz - gh[9]) * gh[11], 0.0), (double)(RT_SGRID - 1));
              const double fy = __builtin_fmin(__builtin_fmax((vy * iz - gh[10]) * gh[12], 0.0), (double)(RT_SGRID - 1));
              const bool proj = (vz > 0.0) && (fx == fx) && (fy == fy);
              const uint32_t cell = proj ? (uint32_t)fy * RT_SGRID + (uint32_t)fx : (uint32_t)(RT_SGRID * RT_SGRID);   // last cell: every sphere
              const uint32_t words = (NLOOP + 63u) >> 6;
              const uint32_t cells_at = 16u * NL + k * (RT_SGRID * RT_SGRID + 1u) * words;      // in 64-bit words from the grid's start
              for (uint32_t wd = 0; wd < words; wd++) {
                unsigned long long cand = 0ull, todo = __ballot(true);
                while (todo) {
                  const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)cell, (int)__builtin_ctzll(todo));
                  cand |= rt_load_word32(sgrid, cells_at + c0 * words + wd);
                  todo &= ~__ballot(cell == c0);
                }
                while (cand) {
                  const uint32_t j = (wd << 6) + (uint32_t)__builtin_ctzll(cand);
                  cand &= cand - 1ull;
                  const rt_geom g0 = RT_LOAD(gl, glo + j);
                  RT_SHADOW_U(j, g0)                                                          // the grid variant never counts
                }
              }
            } else
#endif
            if (!COUNT) {
              if (li != 0.0) {
                uint32_t j = 0;
#if !RT_STRICT
                // a non-empty set still skips the PAIRS of the scan neither sphere of which is in it (a loop over just the named
                // spheres would cost the kernel its 96th register)
                const bool masked = false;
                [[maybe_unused]] uint32_t mk = ~0u;
                if constexpr (!GRID) { if (primary_node && k < 2u) mk = (smask >> (16u * k)) | 0xffff0000u; }
                if (!masked) {
                if constexpr (!GRID) {
                  if (NS >= 2u) { if (mk & 3u) { const rt_geom g0 = gp_first.a, g1 = gp_first.b; RT_SHADOW_U(0u, g0) RT_SHADOW_U(1u, g1) } j = 2u; }
                }
#endif
                for (; j + 2 <= NS; j += 2) {
#if !RT_STRICT
                  if (!GRID && j < 31u && ((mk >> j) & 3u) == 0u) continue;       // (sets name 16 spheres; the upper half of mk is all ones: beyond bit 30 every pair is scanned)
#endif
                  const rt_geom_pair gp = RT_LOAD_PAIR(gl, glo + j);
                  const rt_geom g0 = gp.a, g1 = gp.b;
                  RT_SHADOW_U(j, g0) RT_SHADOW_U(j + 1, g1)
                }
#if !RT_STRICT
                if (j < NS && (GRID || j >= 32u || ((mk >> j) & 1u))) { const rt_geom g0 = RT_LOAD(gl, glo + j); RT_SHADOW_U(j, g0) }
                }
#else
                if (j < NS) { const rt_geom g0 = RT_LOAD(gl, glo + j); RT_SHADOW_U(j, g0) }
#endif
              }
            } else
            if (COUNT || li != 0.0) {                  // li == 0 on entry (an earlier light was blocked) cannot change
              uint32_t j = 0;
              for (; j + 2 <= NS; j += 2) {
                const rt_geom g0 = RT_LOAD(gl, glo + j), g1 = RT_LOAD(gl, glo + j + 1);
                RT_SHADOW(j, g0) RT_SHADOW(j + 1, g1)
                if (blocked) break;
              }
              if (j < NS && !blocked) { const rt_geom g0 = RT_LOAD(gl, glo + j); RT_SHADOW(j, g0) }
            }
#undef RT_SHADOW
#undef RT_SHADOW_U
#undef RT_SROOTS
#undef RT_SDISC
            if (COUNT) cnt[2] += tests;
            if (li == 0.0) continue;
#if RT_STRICT
            diffuse += li * sdot / lmag;                               // main.js:306
#else
            diffuse += (li * sdot) * (inv_llen * inv_llen);            // 1/lmag = (1/llen)^2, already at hand
#endif
#if defined(RT_TESTING) && defined(RT_ABLATE_SPEC)
            if (false) {
#else
            if (a2 > 0.0) {                                            // main.js:307-314
#endif
#if RT_STRICT
              double ql;
              const v3 q = unit(reflect(mk(-sv.x, -sv.y, -sv.z), l), &ql);
              const double spd = d.x * -q.x + d.y * -q.y + d.z * -q.z;
#else
              // reflect(-sv, l) = -sv + l*(2 sv.l): sv.l is sdot, and the mirror image of a unit vector in
              // a unit normal is a unit vector, so the reference's re-normalisation moves it by an ulp at most
              const double t2 = 2.0 * sdot;
              const v3 q = mk(__builtin_fma(l.x, t2, -sv.x), __builtin_fma(l.y, t2, -sv.y), __builtin_fma(l.z, t2, -sv.z));
              const double spd = -(d.x * q.x + d.y * q.y + d.z * q.z);
#endif
#if !RT_STRICT
              // (materials in HBM - the reflection-only many-sphere variants -: the record's address is derived again here, from the hit
              // code, so that no 64-bit pointer lives across the shadow scans)
              double spec_e;
              if constexpr (GRID && !REFRACT && !COUNT) {
                uint32_t off_ = (uint32_t)hi * (uint32_t)sizeof(rt_mtl);
                asm volatile("" : "+v"(off_));
                spec_e = ((const rt_mtl *)((const char *)mtl + off_))->specular_exponent;
              } else spec_e = m.specular_exponent;
              if (spd > 0.0) specular += rt_pow(spd, spec_e);
#else
              if (spd > 0.0) specular += rt_pow(spd, m.specular_exponent);
#endif
            }
          }
          diffuse = min1(diffuse) * a1;
          specular = min1(specular) * a2;
#ifdef RT_TESTING
          probe_li = li;
#endif
        }

        // A4 reflection direction.  (Computed AFTER the lighting: in program order the reference does it before, but it
        // is pure, and placed here neither r nor f — nor n, which is l with its sign restored — occupies registers
        // across the shadow scans, the hottest loop of the kernel.)
#if RT_STRICT
        const v3 nq = n;
#else
        const v3 nq = inside ? mk(-l.x, -l.y, -l.z) : l;
#endif
        v3 r = mk(0, 0, 0); double rlen = 0.0;
        // (with segs_left == 1 the child returns [0,0,0] at main.js:221 whatever its direction: skip it)
        if (a3 > 0.0 && segs_left > 1) r = unit(reflect(d, nq), &rlen);
        // A5 refraction direction
        v3 f = mk(0, 0, 0); double flen = 0.0;
        if (REFRACT && a4 > 0.0 && segs_left > 1) {
          const double dn = dot(d, nq);
          double cosi = -((dn < -1.0) ? -1.0 : min1(dn));              // -Math.max(-1, Math.min(1, dot))
          v3 nn = nq; double eta;
          if (cosi < 0.0) { cosi = -cosi; nn = mk(-nq.x, -nq.y, -nq.z); eta = m.refract_index; }
          else eta = rt_rcp(m.refract_index);
          const double k = 1.0 - eta * eta * (1.0 - cosi * cosi);
          if (k > 0.0) {
            const double q = eta * cosi - rt_sqrt(k);
            f = mk(d.x * eta + nn.x * q, d.y * eta + nn.y * q, d.z * eta + nn.z * q);
          } else f = reflect(d, nn);                                   // total internal reflection
          f = unit(f, &flen);
        }

        if constexpr (FOLD_FORWARD && REFRACT) {
#pragma unroll
          for (int c = 0; c < 3; c++) col[c] = acc[(10 + c) * RT_WG_THREADS];
        }
        const bool go_r = (rlen != 0.0);
        const bool go_f = REFRACT && (flen != 0.0);
#ifdef RT_TESTING
        if (is_probe && probe_n < RT_PROBE_NODES) {
          double *q = L.probe + (size_t)(probe_n++) * RT_PROBE_WORDS;
          q[0] = (double)(REFRACT ? tree_path : (1u << RT_LVL(level))); q[1] = (double)hcode; q[2] = ht;
          q[3] = h.x; q[4] = h.y; q[5] = h.z; q[6] = n.x; q[7] = n.y; q[8] = n.z; q[9] = d.x; q[10] = d.y; q[11] = d.z;
          q[12] = col[0]; q[13] = col[1]; q[14] = col[2]; q[15] = diffuse; q[16] = specular; q[17] = (double)segs_left;
          q[18] = probe_li; q[19] = p.x; q[20] = p.y; q[21] = p.z; q[22] = (double)(go_r ? 1 : 0) + 2.0 * (go_f ? 1 : 0); q[23] = 1.0;
        }
#endif
        if (!go_r && !go_f) {
          // children are absent or return [0,0,0] (segs == 0, main.js:221): x + 0*a == x
#pragma unroll
          for (int c = 0; c < 3; c++) ret[c] = maxa(col[c] * a0, min1(col[c] * diffuse + col[c] * specular));
        } else if constexpr (FOLD_FORWARD) {
          // Each level maps its child's colour x through  f(x) = max(amb, min(1, (ds [+ other child]) + a*x))
          // (main.js:326-336), a non-decreasing clamped-affine map, and compositions of such maps are again
          // clamped-affine.  So the pixel, as a function of the colour of the ray currently being traced, is kept in
          // closed form  F(x) = max(LO, min(HI, O + S*x))  (S one scalar; O, LO, HI per channel) and updated on the
          // way DOWN: a node with ONE child (reflection-only or refraction-only: mirrors, metals, glass) needs no
          // stack and no unwinding at any depth.  The ten doubles live in LDS (lane-major, conflict-free), touched
          // once per bounce.   F o f:  S' = S*a,  O' = O + S*ds,  LO' = clampF(O + S*amb),  HI' = clampF(O + S*max(amb,1))
          // A node with BOTH children (a bubble) is parked with the map accumulated so far, its reflection subtree is
          // traced under a fresh (identity) map, and when that subtree's colour is known the node continues as a
          // one-child node through its refraction ray (main.js:268-278: reflection is evaluated before refraction).
          const uint32_t T = W1 ? 64u : RT_WG_THREADS;
          double A[3], D[3];
#pragma unroll
          for (int c = 0; c < 3; c++) { A[c] = col[c] * a0; D[c] = col[c] * diffuse + col[c] * specular; }
          const bool via_f = REFRACT && !go_r;                          // the only child is the refraction ray
          if (REFRACT && go_r && go_f) {
            park &pk = parked[sp++];
#pragma unroll
            for (int c = 0; c < 3; c++) { pk.amb[c] = A[c]; pk.ds[c] = D[c]; }
            pk.a3 = a3; pk.a4 = a4; pk.h[0] = h.x; pk.h[1] = h.y; pk.h[2] = h.z; pk.f[0] = f.x; pk.f[1] = f.y; pk.f[2] = f.z;
            pk.path = tree_path; pk.segs_left = segs_left; pk.level = level; pk.map_valid = map_valid; pk.hcode = hcode;
            if (map_valid) {
              pk.S = acc[0];
#pragma unroll
              for (int c = 0; c < 3; c++) { pk.O[c] = acc[(1 + c) * T]; pk.LO[c] = acc[(4 + c) * T]; pk.HI[c] = acc[(7 + c) * T]; }
            }
            map_valid = false;
            p = h; d = r; tree_path = 2u * tree_path;
          } else {
            const double coef = via_f ? a4 : a3;
            if (!map_valid) {
              acc[0] = coef;
#pragma unroll
              for (int c = 0; c < 3; c++) { acc[(1 + c) * T] = D[c]; acc[(4 + c) * T] = A[c]; acc[(7 + c) * T] = __builtin_fmax(A[c], 1.0); }
            } else {
              const double S = acc[0];
#pragma unroll
              for (int c = 0; c < 3; c++) {
                const double O = acc[(1 + c) * T], LO = acc[(4 + c) * T], HI = acc[(7 + c) * T];
                const double l2 = __builtin_fma(S, A[c], O), h2 = __builtin_fma(S, __builtin_fmax(A[c], 1.0), O);
                acc[(1 + c) * T] = __builtin_fma(S, D[c], O);
                acc[(4 + c) * T] = __builtin_fmax(LO, __builtin_fmin(HI, l2));
                acc[(7 + c) * T] = __builtin_fmax(LO, __builtin_fmin(HI, h2));
              }
              acc[0] = S * coef;
            }
            map_valid = true;
            p = h; d = via_f ? f : r; tree_path = 2u * tree_path + (via_f ? 1u : 0u);
          }
          level++; segs_left--;
          descend = true;
        } else {
          frame<REFRACT> &fr = stack[level];
#pragma unroll
          for (int c = 0; c < 3; c++) { fr.amb[c] = col[c] * a0; fr.ds[c] = col[c] * diffuse + col[c] * specular; }
          fr.a3 = a3;
          if constexpr (REFRACT) {
            fr.a4 = a4; fr.h[0] = h.x; fr.h[1] = h.y; fr.h[2] = h.z; fr.f[0] = f.x; fr.f[1] = f.y; fr.f[2] = f.z;
            fr.re[0] = fr.re[1] = fr.re[2] = 0.0;
            fr.has_f = go_f; fr.phase = go_r ? 0 : 1;
          }
          p = h; d = go_r ? r : f;
          tree_path = 2u * tree_path + (go_r ? 0u : 1u);
          level++; segs_left--;
          descend = true;
        }
      }
      if (descend) continue;

      if constexpr (FOLD_FORWARD) {
        // a chain of one-child nodes ended with colour `ret`: apply the accumulated map once
        const uint32_t T = W1 ? 64u : RT_WG_THREADS;
        if (map_valid) {
          const double S = acc[0];
#pragma unroll
          for (int c = 0; c < 3; c++) ret[c] = __builtin_fmax(acc[(4 + c) * T], __builtin_fmin(acc[(7 + c) * T], __builtin_fma(S, ret[c], acc[(1 + c) * T])));
        }
        bool resumed = false;
        if constexpr (REFRACT) {
          if (sp > 0) {
            // `ret` is the colour of a parked node's reflection child: fold it into the node's constant term, put the
            // map that was accumulated above the node back, and go on through the node's refraction ray
            const park &pk = parked[--sp];
            const double coef = pk.a4;
            if (!pk.map_valid) {
              acc[0] = coef;
#pragma unroll
              for (int c = 0; c < 3; c++) {
                acc[(1 + c) * T] = pk.ds[c] + ret[c] * pk.a3; acc[(4 + c) * T] = pk.amb[c]; acc[(7 + c) * T] = __builtin_fmax(pk.amb[c], 1.0);
              }
            } else {
              const double S = pk.S;
#pragma unroll
              for (int c = 0; c < 3; c++) {
                const double Dn = pk.ds[c] + ret[c] * pk.a3;
                const double l2 = __builtin_fma(S, pk.amb[c], pk.O[c]), h2 = __builtin_fma(S, __builtin_fmax(pk.amb[c], 1.0), pk.O[c]);
                acc[(1 + c) * T] = __builtin_fma(S, Dn, pk.O[c]);
                acc[(4 + c) * T] = __builtin_fmax(pk.LO[c], __builtin_fmin(pk.HI[c], l2));
                acc[(7 + c) * T] = __builtin_fmax(pk.LO[c], __builtin_fmin(pk.HI[c], h2));
              }
              acc[0] = S * coef;
            }
            map_valid = true;
            p = mk(pk.h[0], pk.h[1], pk.h[2]); d = mk(pk.f[0], pk.f[1], pk.f[2]);
            tree_path = 2u * pk.path + 1u; segs_left = pk.segs_left - 1u; level = pk.level + 1;
            hcode = pk.hcode;                          // the refraction ray starts on the parked node's sphere (bounce table)

            resumed = true;
          }
        }
        if (!resumed) break;
      } else {
        // ---------------- return `ret` to the parents (post-order fold, main.js:268-278, :326-336) ----------------
        bool resumed = false;
        while (level > 0) {
          level--; segs_left++; tree_path >>= 1;
          frame<REFRACT> &fr = stack[level];
          if constexpr (REFRACT) {
            if (fr.phase == 0) {
              fr.re[0] = ret[0] * fr.a3; fr.re[1] = ret[1] * fr.a3; fr.re[2] = ret[2] * fr.a3;
              if (fr.has_f) {                            // now the refraction child of the same node
                fr.phase = 1;
                p = mk(fr.h[0], fr.h[1], fr.h[2]); d = mk(fr.f[0], fr.f[1], fr.f[2]);
                tree_path = 2u * tree_path + 1u;
                level++; segs_left--;
                resumed = true;
                break;
              }
#pragma unroll
              for (int c = 0; c < 3; c++) ret[c] = maxa(fr.amb[c], min1(fr.ds[c] + fr.re[c]));
            } else {
#pragma unroll
              for (int c = 0; c < 3; c++) ret[c] = maxa(fr.amb[c], min1(fr.ds[c] + fr.re[c] + ret[c] * fr.a4));
            }
          } else {
#pragma unroll
            for (int c = 0; c < 3; c++) ret[c] = maxa(fr.amb[c], min1(fr.ds[c] + ret[c] * fr.a3));
          }
        }
        if (!resumed) break;
      }
    }
  }
#undef RT_LOAD
#undef RT_LOAD_PAIR
  rgb[0] = ret[0]; rgb[1] = ret[1]; rgb[2] = ret[2];
}

// W1: one-wave workgroups (rt_pixel_of), for the reflection-only variants.  A workgroup's waves are placed together: a 4-wave
// workgroup starts when its CU has room for all four, and where the waves of a launch differ in length freed slots wait for their
// neighbours - the dear two thirds of a 64-sphere frame kept 3 400 - 4 100 of the chip's 5 120 wave slots resident, the headline
// 4 550 - 4 720 (profiles/wave_timeline.py).  One wave per workgroup fills every slot as it frees.  The many-sphere variants stage
// nothing (64 spheres 2x2 -8.6 %, cfg5's frame -7.1 %); the few-sphere ones stage their image per wave, as 16-byte units - 8 spheres:
// two loads and two LDS stores per work-item, no barrier (headline -2.8 %, 8K -1.6 %; with the 8-byte staging loop of the four-wave
// form it was +0.5 %).  The general kernel keeps four waves (its 13-double fold state and image would leave a CU 16 one-wave
// workgroups), and so does the peer-store path (whole 128-byte lines across the workgroup's waves).  profiles/r04_ab_log.md section 8.
template <bool REFRACT, bool COUNT, bool SS2, bool GRID, bool W1 = false>
// Register budget: the reflection-only kernel fits 96 VGPRs = 5 waves per SIMD on its own (measured: 4 waves cost 11 %,
// more than 5 gain nothing); its shadow-grid variant is held there; the general kernel is left free (forcing it to 5
// waves spills into its loops: -9 %).
__global__ void __launch_bounds__(W1 ? 64 : RT_WG_THREADS, ((REFRACT || !GRID) ? RT_WAVES_PER_EU : (RT_WAVES_PER_EU > 5 ? RT_WAVES_PER_EU : 5))) rt_trace(const rt_launch L) {
  static_assert(!W1 || (!REFRACT && !COUNT && !RT_STRICT), "one-wave workgroups: the reflection-only product variants");
  [[maybe_unused]] constexpr uint32_t WG_WAVES = W1 ? 1u : RT_WG_THREADS / 64u;
  constexpr uint32_t KT = W1 ? 64u : RT_WG_THREADS;            // work-items of this workgroup
  extern __shared__ double lds_raw[];
#if defined(RT_WAVE_LOG) && !RT_STRICT
  // measurement build: when this wave started, and where (nothing is kept in registers: the exit stamp recomputes its slot)
  if (unsigned long long *const wl = rt_cold_args()->wave_log; wl != nullptr && (threadIdx.x & 63u) == 0u) {
    unsigned long long *q = wl + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * WG_WAVES + (threadIdx.x >> 6)) * 4u;
    q[0] = __builtin_amdgcn_s_memrealtime();
    q[2] = (unsigned long long)__builtin_amdgcn_s_getreg((4u) | (0u << 6) | (31u << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20u) | (0u << 6) | (31u << 11)) << 32);
    q[3] = rt_entry_index<W1>();
  }
#endif
  // ---- stage the per-workgroup tables into LDS: ONE contiguous image in HBM (materials | texture descriptors |
  //      cull rectangles, laid out exactly as the LDS copy), so a workgroup pays one memory latency, not three;
  //      the loads are issued first and land while the ray is being generated ----
  const uint32_t tid = threadIdx.x;
  const uint32_t mtl_words = L.n_objects * (uint32_t)(sizeof(rt_mtl) / 8);
  const uint32_t tex_words = 16u * 2u;               // RT_MAX_TEXTURES descriptors of 16 B
  const bool cull_lds_on = (!RT_STRICT && !COUNT) ? !GRID : (L.cull_in_lds != 0u);      // (see trace_pixel)
  const uint32_t cull_words = cull_lds_on ? L.n_objects * 4u : 0u;     // per-sphere screen rectangles of the primary-ray cull (few spheres)
  const uint32_t image_words = mtl_words + tex_words + cull_words;
  // The reflection-only many-sphere variants read materials and texture descriptors WHERE THEY ARE (HBM / L2 / L1: a 32-bit offset from
  // a uniform base) instead of staging 10 KB of them per workgroup: LDS is then the fold state alone, 20 KB instead of 30 - a
  // workgroup's LDS stays allocated until its slowest wave ends, and with 30 KB (five per CU) the dear two thirds of a launch kept
  // only 66-78 % of the wave slots resident - and there is no staging and no barrier.  64 spheres: -3.8 % (3840x2160), -2.1 % (2x2),
  // cfg5's full frame -2.6 % (profiles/r04_ab_log.md section 6).  (The general kernel keeps its image in LDS: it has no register to spare.)
  constexpr bool IMAGE_IN_LDS = RT_STRICT || COUNT || !(GRID && !REFRACT);
  const uint32_t lds_words = IMAGE_IN_LDS ? image_words : 0u;
  const double *__restrict__ image = (const double *)L.lds_image;
  // Few spheres (8: exactly one word per work-item): one 8-byte load each, a loop for the rest.  The many-sphere variant (64
  // spheres are 10.5 KB) moves 16 bytes per work-item and instruction, three 4 KB pieces unrolled with ALL their loads in flight
  // before the first is waited for, the loads themselves unconditional (the host pads the buffer to whole pieces, so a piece
  // that starts inside the image may be read to its end) behind a scalar test per piece: one memory latency per workgroup
  // and ~a tenth of the instructions of a word-by-word loop (profiles/r02_ab_log.md).
  typedef uint32_t __attribute__((ext_vector_type(4))) rt_u4;
  constexpr uint32_t RT_STAGE_PIECES = 3u;
  const uint32_t image_vec = image_words >> 1;                                     // 16-byte units (GRID: the image is whole units)
  [[maybe_unused]] rt_u4 piece[RT_STAGE_PIECES];
  [[maybe_unused]] double stage0 = 0.0;
  if constexpr (!IMAGE_IN_LDS) {
  } else if constexpr (GRID || W1) {           // (W1, few spheres: 8 spheres are 112 units - two loads per work-item of the one wave)
    const rt_u4 *__restrict__ image4 = (const rt_u4 *)L.lds_image;
#pragma unroll
    for (uint32_t i = 0; i < RT_STAGE_PIECES; i++) if (i * KT < image_vec) piece[i] = image4[tid + i * KT];
  } else {
    stage0 = (tid < image_words) ? image[tid] : 0.0;
  }
  const rt_mtl *mtl = IMAGE_IN_LDS ? (const rt_mtl *)lds_raw : (const rt_mtl *)L.lds_image;
  const rt_texture_desc *tex = IMAGE_IN_LDS ? (const rt_texture_desc *)(lds_raw + mtl_words) : (const rt_texture_desc *)((const double *)L.lds_image + mtl_words);
  const rt_geom *cull_lds = (const rt_geom *)(lds_raw + mtl_words + tex_words);
  double *acc = lds_raw + lds_words + tid;   // fold state: 10 (general kernel: 13) x RT_WG_THREADS doubles, lane-major
  // Many spheres: the primary-ray cull's rectangle of sphere `lane` (the wave's first 64 spheres) comes straight from HBM / L2,
  // in flight while the ray is generated, and is no part of the LDS image: 64 spheres + the fold state then fit 32 KB, five
  // workgroups per CU instead of four (64-sphere scenes -12 %; with 8 spheres the extra vector load costs 3 %, so few
  // spheres keep their rectangles in the image)
  rt_geom cull0 = rt_geom{0.0, 0.0, 0.0, 0.0};
  if (!cull_lds_on) { const uint32_t lane0 = tid & 63u; cull0 = L.cull[lane0 < L.n_loop ? lane0 : 0u]; }

  // ---- which pixel / sample this work-item owns ----
  const uint32_t lane = tid & 63u;
  const rt_pixel P0 = rt_pixel_of<SS2, W1>(L, tid);
#if !RT_STRICT
  // workgroup-uniform: a block wholly past its tile's or the frame's last row - or no entry at all: while the host does not know how
  // many entries a table built on the GPU a moment ago has, it launches one workgroup per BLOCK, and the slots behind the last
  // entry are zero (rt_tables_gpu.hip)
  if (P0.rows_valid == 0u) return;
  // (RT_FLAG_NO_SKY / RT_FLAG_SKY_ONLY - a frame assembled from several GPUs' tiles: the OWNER fills the sky blocks of the whole frame,
  // the others do not send them over the links - are launch tables of their own, without the entries the launch leaves out: this
  // kernel knows nothing of it.  As a test of the launch record here it cost the headline 1.5 %.)
#endif
  double rgb[3];
  uint32_t cnt[3] = {0u, 0u, 0u};
  // A workgroup the table build marked as showing no sphere (rt_block.h: the cone test, made once for the workgroup's
  // box) stores the background constant: no staging, no barrier, no ray, no cull.  (The staging loads issued above are simply
  // never waited for.)  45 % of the headline's workgroups.
  if (P0.sky) {
    rgb[0] = L.sky_rgb[0]; rgb[1] = L.sky_rgb[1]; rgb[2] = L.sky_rgb[2];
  } else {
  const uint32_t px = P0.px, frow = P0.frow, sub = P0.sub;
  const uint32_t sx = SS2 ? 2u * px + (sub & 1u) : px;
  const uint32_t sy = SS2 ? 2u * frow + (sub >> 1) : frow;

  // ---- A1 primary ray (main.js:186-193); dist is indexed by component k, not by axis (q1) ----
#if RT_STRICT
  const double d0 = ((double)sx - L.proj_w) + 0.5, d1 = (L.proj_h - (double)sy) - 0.5, d2 = L.proj_d;
#else
  // the same numbers with one addition each: sx, sy are integers and proj_w, proj_h half-integers far below 2^52, so
  // sx + (0.5 - proj_w) and (proj_h - 0.5) - sy are exact, like the reference's two-step forms
  const double d0 = (double)sx + L.ray_bias[0], d1 = L.ray_bias[1] - (double)sy;
#endif
  const v3 o = mk(L.cam_origin[0], L.cam_origin[1], L.cam_origin[2]);
  double rl;
#if RT_STRICT
  const v3 target = mk(o.x + L.cam_axis_x[0] * d0 + L.cam_axis_y[0] * d0 + L.cam_axis_z[0] * d0,
                       o.y + L.cam_axis_x[1] * d1 + L.cam_axis_y[1] * d1 + L.cam_axis_z[1] * d1,
                       o.z + L.cam_axis_x[2] * d2 + L.cam_axis_y[2] * d2 + L.cam_axis_z[2] * d2);
  const v3 ray = unit(mk(target.x - o.x, target.y - o.y, target.z - o.z), &rl);
#else
  // target[k] - origin[k] = (axisX[k] + axisY[k] + axisZ[k]) * dist[k]; the sums come from the host
  // (its z component is axis_sum.z * projD: never the zero vector unless the camera is degenerate, and then the
  // reference divides by zero as well, so no zero-length select here)
  const v3 rawray = mk(L.cam_axis_sum[0] * d0, L.cam_axis_sum[1] * d1, L.ray_bias[2]);      // [2] = axis_sum.z * projD, from the host
  rl = rt_rsqrt_pos(dot(rawray, rawray));
  const v3 ray = mk(rawray.x * rl, rawray.y * rl, rawray.z * rl);
#endif

  // finish the staging (first use of LDS: the cull table or the closest hit's material inside trace_pixel)
  if constexpr (!IMAGE_IN_LDS) {
  } else if constexpr (GRID || W1) {
    rt_u4 *lds4 = (rt_u4 *)lds_raw;
    const rt_u4 *__restrict__ image4 = (const rt_u4 *)L.lds_image;
#pragma unroll
    for (uint32_t i = 0; i < RT_STAGE_PIECES; i++) {
      const uint32_t k = tid + i * KT;
      if (i * KT < image_vec && k < image_vec) lds4[k] = piece[i];
    }
    for (uint32_t k = tid + RT_STAGE_PIECES * KT; k < image_vec; k += KT) lds4[k] = image4[k];     // more than 73 spheres (one-wave workgroups: 17)
  } else {
    if (tid < image_words) lds_raw[tid] = stage0;
    for (uint32_t k = tid + RT_WG_THREADS; k < image_words; k += RT_WG_THREADS) lds_raw[k] = image[k];
  }
  if constexpr (IMAGE_IN_LDS) __syncthreads();

  // this wave's pixel block in the units of d0/d1 (every lane holds the same four numbers)
  const double bw = SS2 ? 15.0 : 7.0, bh = SS2 ? 3.0 : 7.0;
  const double lx = SS2 ? (double)(2u * ((lane >> 2) & 7u) + (sub & 1u)) : (double)(lane & 7u);
  const double ly = SS2 ? (double)(2u * (lane >> 5) + (sub >> 1)) : (double)(lane >> 3);
  const double blk_x0 = d0 - lx, blk_x1 = blk_x0 + bw, blk_y1 = d1 + ly, blk_y0 = blk_y1 - bh;
#ifdef RT_TESTING
  const bool is_probe = L.probe != nullptr && sx == L.probe_x && sy == L.probe_y && blockIdx.z == 0;
#else
  const bool is_probe = false;
#endif
  trace_pixel<REFRACT, COUNT, GRID, SS2, false, W1>(L, mtl, tex, acc, cull_lds, cull0, lane, blk_x0, blk_x1, blk_y0, blk_y1, o, ray, rgb, cnt, is_probe, P0.cand);
  }

  // ---- A10 RGBA8 store ----
  uint32_t tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));                       // opaque: recompute the pixel instead of keeping it live (see rt_pixel_of)
  const rt_pixel P1 = rt_pixel_of<SS2, W1>(L, tid2);
  const uint32_t frame_i = blockIdx.z;
  const bool valid = P1.valid;
  const uint32_t r8 = to_byte(rgb[0]), g8 = to_byte(rgb[1]), b8 = to_byte(rgb[2]);
  // scatter: every frame of the batch has its own destination (the frame buffer of the rank that owns it, peer-mapped
  // over xGMI) and rows go to their place in the FRAME, so nothing is left to exchange or to de-interleave; a tile row
  // is whole 128-byte lines written by one workgroup, which is what a remote store wants
  uint32_t *__restrict__ out = L.scatter ? L.out_frames[frame_i] : L.out + (size_t)frame_i * L.frame_stride;
  const uint32_t orow = L.scatter ? P1.frow : P1.lrow;
  uint32_t rgbw;                                       // this lane's pixel as 0x00BBGGRR
  if (!SS2) rgbw = r8 | (g8 << 8) | (b8 << 16);
  else {
    // 2x2 box filter across the 4 lanes of a quad: (a+b+c+d+2)>>2 per channel (10-bit fields); all four lanes end
    // up with the pixel
    uint32_t packed = r8 | (g8 << 10) | (b8 << 20);
    packed += __shfl_xor(packed, 1);
    packed += __shfl_xor(packed, 2);
    rgbw = (((packed & 1023u) + 2u) >> 2) | ((((packed >> 10) & 1023u) + 2u) >> 2 << 8) | ((((packed >> 20) & 1023u) + 2u) >> 2 << 16);
  }
  // A sky entry of the launch table stands for a RUN of consecutive 32-pixel blocks of one row block (rt_tables_gpu.hip): the
  // workgroup stores the same constant into each of them; every other workgroup stores its one block.
#if RT_STRICT
  const uint32_t n_run = 1u;
#else
  const uint32_t n_run = P1.sky ? P1.run : 1u;
#endif
  if (!W1 && !L.rgb24 && L.scatter && !SS2) {          // workgroup-uniform (the host launches the four-wave variant for scatter stores)
    // Peer stores want whole lines: a wave's 8x8 block is eight 32-byte pieces, one per row, and memory on the far side
    // of an xGMI link has no L2 of ours in front of it to merge them.  So the workgroup transposes its 32x8 tile through
    // LDS - every lane parks its pixel in its own fold-state slot 0, dead by now - and each wave then stores two whole
    // 128-byte rows of the tile.
    uint32_t *tile = (uint32_t *)(lds_raw + lds_words);
    tile[2u * tid2] = rgbw;
    __syncthreads();
    const uint32_t xr = tid2 & 31u, rr = tid2 >> 5;     // this work-item's pixel of the tile in row-major order
    const uint32_t v = tile[2u * (((xr >> 3) << 6) + (rr << 3) + (xr & 7u))];
    const uint32_t lane2 = tid2 & 63u;
    const uint32_t dr = rr - (lane2 >> 3);              // row of the tile: difference in wrap-around arithmetic
    const uint32_t trow2 = P1.trow + dr, frow2 = P1.frow + dr;
    uint32_t px2 = P1.px - ((tid2 >> 6) * 8u + (lane2 & 7u)) + xr;
    for (uint32_t t = 0; t < n_run; t++, px2 += RT_TILE_W) {
#if RT_STRICT
      const bool row_ok2 = trow2 < L.tile_rows && frow2 < L.h;
#else
      const bool row_ok2 = trow2 < P1.rows_valid;      // rows of the block inside its tile and the frame (from the table entry)
#endif
      if (row_ok2 && px2 < L.w) out[(size_t)frow2 * L.w + px2] = v | 0xff000000u;
    }
  } else if (!L.rgb24) {                               // wave-uniform
    uint32_t pxq = P1.px;
    for (uint32_t t = 0; t < n_run; t++, pxq += RT_TILE_W) {
#if RT_STRICT
      const bool validq = valid;
#else
      const bool validq = (pxq < L.w) && (P1.trow < P1.rows_valid);
#endif
      if (validq && P1.sub == 0u) out[(size_t)orow * L.w + pxq] = rgbw | 0xff000000u;
    }
  } else {
    // RT_FLAG_RGB24: the 8 pixels a wave holds of one row are 24 bytes = 6 words.  Word j of the group takes its bytes
    // from pixels p = j + j/3 and p + 1, shifted by (j mod 3) bytes; lanes j < 6 of the group store.  w % 4 == 0 (host
    // check), so a group at the right edge holds 8 or 4 pixels = 6 or 3 whole words and rows start word-aligned.
    const uint32_t lane2 = tid2 & 63u;
    const uint32_t slot = SS2 ? lane2 >> 2 : lane2;    // pixel slot in the wave; a row's 8 slots are consecutive
    const uint32_t j = slot & 7u, j3 = (j * 11u) >> 5; // j3 = j / 3 for j < 8
    const uint32_t p = (slot & ~7u) + j + j3, sh = (j - 3u * j3) * 8u;
    const uint32_t pn = (j < 6u) ? p + 1u : p;         // lanes 6,7 store nothing; keep their source lane inside the group
    const uint32_t lo = __shfl(rgbw, SS2 ? (p << 2) : p), hi = __shfl(rgbw, SS2 ? (pn << 2) : pn);
    const uint32_t word = (lo >> sh) | (hi << (24u - sh));
    uint32_t x0 = P1.px - j;                           // first pixel of the group (a multiple of 8)
    for (uint32_t t = 0; t < n_run; t++, x0 += RT_TILE_W) {
      const uint32_t in_row = (x0 + 8u <= L.w) ? 6u : ((x0 + 4u <= L.w) ? 3u : 0u);
#if RT_STRICT
      const bool row_ok = (P1.trow < L.tile_rows) && (P1.frow < L.h);
#else
      const bool row_ok = P1.trow < P1.rows_valid;
#endif
      if (!(row_ok && j < in_row && P1.sub == 0u)) continue;
      // RT_FLAG_COMPACT: the block whole, at its place in the LAUNCH (a compact band: rows of 32 pixels = 24 words, 8 - or 2 - of them)
      if (L.compact) out[(size_t)rt_entry_index<W1>() * (RT_TILE_W * 3u / 4u * (SS2 ? 2u : RT_TILE_H)) + P1.trow * (RT_TILE_W * 3u / 4u) + ((rt_wave_of<W1>(tid2) * 8u) >> 2) * 3u + j] = word;
      else out[((P1.lrow * L.w + x0) >> 2) * 3u + j] = word;
    }
  }

#if defined(RT_WAVE_LOG) && !RT_STRICT
  if (unsigned long long *const wl = rt_cold_args()->wave_log; wl != nullptr && (tid2 & 63u) == 0u)
    wl[((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * WG_WAVES + (tid2 >> 6)) * 4u + 1u] = __builtin_amdgcn_s_memrealtime();
#endif
  if (COUNT) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      unsigned long long v = valid ? cnt[c] : 0u;
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if ((tid2 & 63u) == 0) atomicAdd(&L.counters[c], v);
    }
  }
}

#if RT_STRICT
// rt_retrace - the second, list-driven launch of a product frame (strict build only).  Its items are
//   * the samples the product launch marked (L.marks: a sampler coordinate within rounding of a texel / checker boundary),
//   * the centre row and the centre column of an odd sample grid that fall into this call's tiles: x - w/2 + 0.5 == 0 there
//     (main.js:186), so the primary ray - and every ray it spawns that stays in that plane - lies in a coordinate plane through the
//     camera, a sphere centred on that plane is met with a normal component of exactly 0, and u or v lands exactly ON a boundary,
//   * every sample of the call when the list overflowed (or the test build asks for it);
// each is traced with the reference's own operation sequence - trace_pixel in ITEM mode: every sphere in the scene's own order,
// materials read from HBM, no wave-wide step (a wave's lanes hold unrelated samples) - and stored where the product launch
// stored it (band, RGB24 band, or its row of the frame in scatter mode).  With supersample 2 the pixel's four samples are all
// traced (the product launch kept only their average).  A grid-stride loop: the number of items is only known on the device.
// Work-item 0 clears the NEXT launch's counter and publishes this launch's count to the host (rt_api.hip skips this launch from
// then on if a frame of this scene, camera, size and tile set has no item at all).
template <bool REFRACT, bool SS2>
__global__ void __launch_bounds__(RT_WG_THREADS) rt_retrace(const rt_launch L) {
  const uint32_t count = L.marks[L.marks_slot];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    L.marks[L.marks_slot ^ 1u] = 0u;
    if (L.marks_known) __hip_atomic_store(L.marks_known, ((unsigned long long)L.known_tag << 32) | (count + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const uint32_t band_rows = L.n_tiles * L.tile_rows;
  const bool everything = count > L.marks_cap || L.retrace_all != 0u;
  const unsigned long long n_list = everything ? 0ull : count;
  const unsigned long long n_row = (!everything && L.centre_row != ~0u) ? (unsigned long long)L.w * L.n_frames : 0ull;
  const unsigned long long n_col = (!everything && L.centre_col != ~0u) ? (unsigned long long)band_rows * L.n_frames : 0ull;
  const unsigned long long n_all = everything ? (unsigned long long)band_rows * L.w * L.n_frames : 0ull;
  const unsigned long long total = n_list + n_row + n_col + n_all;
  const unsigned long long *list = (const unsigned long long *)(L.marks + 4);
  const rt_mtl *mtl = (const rt_mtl *)L.lds_image;                         // (HBM: nothing is staged here)
  const rt_texture_desc *tex = (const rt_texture_desc *)((const char *)L.lds_image + (size_t)L.n_objects * sizeof(rt_mtl));
  // supersample 2: the four samples of an item's pixel in four adjacent lanes (a quad: the same item, the same control flow), their
  // bytes summed across the quad - a marked sample of a many-sphere scene is a deep tree, and four of them one after the other were
  // 0.35 ms of a 0.34 ms frame (profiles/r04_ab_log.md)
  constexpr unsigned long long SUBS = SS2 ? 4ull : 1ull;
  for (unsigned long long ii = (unsigned long long)blockIdx.x * RT_WG_THREADS + threadIdx.x; ii < total * SUBS; ii += (unsigned long long)gridDim.x * RT_WG_THREADS) {
    const unsigned long long i = ii / SUBS;
    uint32_t px, frow, lrow, f;
    bool have_lrow = false;
    lrow = 0u;
    if (i < n_list) {
      const unsigned long long e = list[i];
      const uint32_t sx = (uint32_t)(e & 0xfffffu), sy = (uint32_t)((e >> 20) & 0xfffffu);
      f = (uint32_t)(e >> 40);
      px = SS2 ? sx >> 1 : sx; frow = SS2 ? sy >> 1 : sy;
    } else if (i < n_list + n_row) {
      const unsigned long long k = i - n_list;
      f = (uint32_t)(k / L.w); px = (uint32_t)(k - (unsigned long long)f * L.w); frow = L.centre_row;
    } else {
      unsigned long long k = i - n_list - n_row;
      if (!everything) { f = (uint32_t)(k / band_rows); lrow = (uint32_t)(k - (unsigned long long)f * band_rows); px = L.centre_col; }
      else { f = (uint32_t)(k / ((unsigned long long)band_rows * L.w)); k -= (unsigned long long)f * band_rows * L.w; lrow = (uint32_t)(k / L.w); px = (uint32_t)(k - (unsigned long long)lrow * L.w); }
      const uint32_t tile_i = lrow / L.tile_rows, trow = lrow - tile_i * L.tile_rows;
      frow = (L.tile_first + tile_i * L.tile_stride) * L.tile_rows + trow;
      have_lrow = true;
    }
    if (!have_lrow) {                                    // the row's place in this call's band, if the call renders it at all
      const uint32_t t = frow / L.tile_rows;
      if (t < L.tile_first || (t - L.tile_first) % L.tile_stride != 0u || (t - L.tile_first) / L.tile_stride >= L.n_tiles) continue;
      lrow = ((t - L.tile_first) / L.tile_stride) * L.tile_rows + (frow - t * L.tile_rows);
    }
    if (px >= L.w || frow >= L.h || f >= L.n_frames) continue;
    uint32_t sum[3] = {0u, 0u, 0u};
    const uint32_t sub = (uint32_t)(ii % SUBS);
    {
      const uint32_t sx = SS2 ? 2u * px + (sub & 1u) : px, sy = SS2 ? 2u * frow + (sub >> 1) : frow;
      // A1 primary ray (main.js:186-193), literally
      const double d0 = ((double)sx - L.proj_w) + 0.5, d1 = (L.proj_h - (double)sy) - 0.5, d2 = L.proj_d;
      const v3 o = mk(L.cam_origin[0], L.cam_origin[1], L.cam_origin[2]);
      const v3 target = mk(o.x + L.cam_axis_x[0] * d0 + L.cam_axis_y[0] * d0 + L.cam_axis_z[0] * d0,
                           o.y + L.cam_axis_x[1] * d1 + L.cam_axis_y[1] * d1 + L.cam_axis_z[1] * d1,
                           o.z + L.cam_axis_x[2] * d2 + L.cam_axis_y[2] * d2 + L.cam_axis_z[2] * d2);
      double rl;
      const v3 ray = unit(mk(target.x - o.x, target.y - o.y, target.z - o.z), &rl);
      double rgb[3];
      uint32_t cnt[3] = {0u, 0u, 0u};
      trace_pixel<REFRACT, false, false, SS2, true>(L, mtl, tex, nullptr, nullptr, rt_geom{0.0, 0.0, 0.0, 0.0}, 0u, 0.0, 0.0, 0.0, 0.0, o, ray, rgb, cnt, false, 0u, sx, sy);
      sum[0] += to_byte(rgb[0]); sum[1] += to_byte(rgb[1]); sum[2] += to_byte(rgb[2]);
    }
    if (SS2) {
      uint32_t packed = sum[0] | (sum[1] << 10) | (sum[2] << 20);
      packed += __shfl_xor(packed, 1);
      packed += __shfl_xor(packed, 2);
      sum[0] = ((packed & 1023u) + 2u) >> 2; sum[1] = (((packed >> 10) & 1023u) + 2u) >> 2; sum[2] = (((packed >> 20) & 1023u) + 2u) >> 2;
      if (sub != 0u) continue;
    }
    if (L.scatter) L.out_frames[f][(size_t)frow * L.w + px] = sum[0] | (sum[1] << 8) | (sum[2] << 16) | 0xff000000u;
    else if (!L.rgb24) L.out[(size_t)f * L.frame_stride + (size_t)lrow * L.w + px] = sum[0] | (sum[1] << 8) | (sum[2] << 16) | 0xff000000u;
    else if (!L.compact) {
      uint8_t *o8 = (uint8_t *)(L.out + (size_t)f * L.frame_stride) + ((size_t)lrow * L.w + px) * 3u;
      o8[0] = (uint8_t)sum[0]; o8[1] = (uint8_t)sum[1]; o8[2] = (uint8_t)sum[2];
    } else {
      // a compact band: the sample's block is workgroup b of the product launch - class start + entries of its class in the rows
      // above + its rank in the row (rt_tables_gpu.hip: rt_table_emit) -, its pixel row r, column i of the block's 32 x RH pixels
      const uint32_t RH = SS2 ? 2u : RT_TILE_H, y = lrow / RH, x = px / RT_TILE_W, at = y * L.tiles_x + x;
      const uint32_t it = L.tb_item[at];
      if (!it || (it >> 16)) continue;                     // (inside a run of sky blocks: a compact band holds none)
      const uint32_t bin = (it & 0xffffu) - 1u, b = L.tb_bin_start[bin] + L.tb_row_hist[(size_t)y * L.tb_bins + bin] + L.tb_rank_in_row[at];
      uint8_t *o8 = (uint8_t *)(L.out + (size_t)f * L.frame_stride) + (size_t)b * (RT_TILE_W * 3u * RH) + ((size_t)(lrow - y * RH) * RT_TILE_W + (px - x * RT_TILE_W)) * 3u;
      o8[0] = (uint8_t)sum[0]; o8[1] = (uint8_t)sum[1]; o8[2] = (uint8_t)sum[2];
    }
  }
}
#endif

}  // namespace

#if RT_STRICT
// Host-side launcher of rt_retrace (n_wg workgroups of RT_WG_THREADS).  Returns a hipError_t as int.
extern "C" int rt_launch_retrace(const rt_launch *L, int refract, int ss2, unsigned n_wg, hipStream_t stream) {
  const dim3 grid(n_wg ? n_wg : 1u), block(RT_WG_THREADS);
  if (!refract) { if (!ss2) hipLaunchKernelGGL((rt_retrace<false, false>), grid, block, 0, stream, *L); else hipLaunchKernelGGL((rt_retrace<false, true>), grid, block, 0, stream, *L); }
  else          { if (!ss2) hipLaunchKernelGGL((rt_retrace<true, false>), grid, block, 0, stream, *L);  else hipLaunchKernelGGL((rt_retrace<true, true>), grid, block, 0, stream, *L); }
  return (int)hipGetLastError();
}
#endif

// Scratch (private segment) bytes per lane of the kernel instantiation the launcher below would pick, from the code object: what the
// runtime reserves for every wave slot of the device before the first launch (rt_api.hip: scratch_guard).  Returns a hipError_t as int.
extern "C" int RT_SCRATCH_NAME(int refract, int count, int ss2, int grid_variant, int one_wave, size_t *bytes_per_lane) {
  const void *f = nullptr;
#define RT_PICK(R, C, S, G) f = (const void *)&rt_trace<R, C, S, G>
  if (!RT_STRICT && grid_variant && !count) {
#if !RT_STRICT
    if (!refract && one_wave) { if (!ss2) f = (const void *)&rt_trace<false, false, false, true, true>; else f = (const void *)&rt_trace<false, false, true, true, true>; }
    else if (!refract) { if (!ss2) RT_PICK(false, false, false, true); else RT_PICK(false, false, true, true); }
    else          { if (!ss2) RT_PICK(true, false, false, true);  else RT_PICK(true, false, true, true); }
#endif
#if !RT_STRICT
  } else if (!count && !refract && one_wave) {
    if (!ss2) f = (const void *)&rt_trace<false, false, false, false, true>; else f = (const void *)&rt_trace<false, false, true, false, true>;
#endif
  } else if (!count) {
    if (!refract) { if (!ss2) RT_PICK(false, false, false, false); else RT_PICK(false, false, true, false); }
    else          { if (!ss2) RT_PICK(true, false, false, false);  else RT_PICK(true, false, true, false); }
  } else {
    if (!refract) { if (!ss2) RT_PICK(false, true, false, false); else RT_PICK(false, true, true, false); }
    else          { if (!ss2) RT_PICK(true, true, false, false);  else RT_PICK(true, true, true, false); }
  }
#undef RT_PICK
  hipFuncAttributes fa;
  const hipError_t e = hipFuncGetAttributes(&fa, f);
  if (e == hipSuccess) *bytes_per_lane = (size_t)fa.localSizeBytes;
  return (int)e;
}
#if RT_STRICT
extern "C" int rt_scratch_retrace(int refract, int ss2, size_t *bytes_per_lane) {
  const void *f = !refract ? (!ss2 ? (const void *)&rt_retrace<false, false> : (const void *)&rt_retrace<false, true>)
                           : (!ss2 ? (const void *)&rt_retrace<true, false> : (const void *)&rt_retrace<true, true>);
  hipFuncAttributes fa;
  const hipError_t e = hipFuncGetAttributes(&fa, f);
  if (e == hipSuccess) *bytes_per_lane = (size_t)fa.localSizeBytes;
  return (int)e;
}
#endif

// Host-side launcher for this translation unit's kernels.  Returns a hipError_t as int.
extern "C" int RT_LAUNCH_NAME(const rt_launch *L, int refract, int count, int ss2, unsigned lds_bytes, hipStream_t stream) {
  // x: 32-pixel tiles across the frame; y: tiles x row blocks (8 rows, or 2 when supersampling); z: frames
  (void)ss2;
  // (the product launch is flat: L->grid_x workgroups, one per launch-table entry)
  const dim3 grid(L->grid_x ? L->grid_x : (L->order ? L->tiles_x * L->n_tiles * L->rb_per_tile : L->tiles_x),
                  L->grid_y ? L->grid_y : (L->order ? 1u : L->n_tiles * L->rb_per_tile), L->n_frames), block(RT_WG_THREADS);
#define RT_CASE(R, C, S, G) hipLaunchKernelGGL((rt_trace<R, C, S, G>), grid, block, lds_bytes, stream, *L)
#if !RT_STRICT
  // one-wave workgroups (rt_pixel_of, W1): four per table entry, whole groups of eight entries (the slots behind the last entry are
  // zero: their workgroups leave at once)
  if (rt_one_wave_workgroups(false, count != 0, refract != 0, (L->scatter != 0u && !ss2) || L->four_waves != 0u)) {
    const dim3 grid1(((grid.x + 7u) / 8u) * 32u, 1u, L->n_frames), block1(64u);
    if (!L->cull_in_lds) {
      if (!ss2) hipLaunchKernelGGL((rt_trace<false, false, false, true, true>), grid1, block1, lds_bytes, stream, *L);
      else hipLaunchKernelGGL((rt_trace<false, false, true, true, true>), grid1, block1, lds_bytes, stream, *L);
    } else {
      if (!ss2) hipLaunchKernelGGL((rt_trace<false, false, false, false, true>), grid1, block1, lds_bytes, stream, *L);
      else hipLaunchKernelGGL((rt_trace<false, false, true, false, true>), grid1, block1, lds_bytes, stream, *L);
    }
    return (int)hipGetLastError();
  }
#endif
  // GRID: the shadow-grid variant, a separate instantiation so that scenes with few spheres do not carry its registers
  // (the host leaves the cull rectangles out of the LDS image exactly for the scenes that have a shadow grid or a bounce table)
  const bool grid_variant = !RT_STRICT && !count && !L->cull_in_lds;
  if (grid_variant) {
#if !RT_STRICT
    if (!refract) { if (!ss2) RT_CASE(false, false, false, true); else RT_CASE(false, false, true, true); }
    else          { if (!ss2) RT_CASE(true, false, false, true);  else RT_CASE(true, false, true, true); }
#endif
  } else if (!count) {
    if (!refract) { if (!ss2) RT_CASE(false, false, false, false); else RT_CASE(false, false, true, false); }
    else          { if (!ss2) RT_CASE(true, false, false, false);  else RT_CASE(true, false, true, false); }
  } else {
    if (!refract) { if (!ss2) RT_CASE(false, true, false, false); else RT_CASE(false, true, true, false); }
    else          { if (!ss2) RT_CASE(true, true, false, false);  else RT_CASE(true, true, true, false); }
  }
#undef RT_CASE
  return (int)hipGetLastError();
}

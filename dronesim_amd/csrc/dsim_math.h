// dsim_math.h — short fp32 elementary functions for the per-drone arithmetic.
//
// The control law needs, per drone-step: one atan2 + one asin (roll, pitch from the
// attitude quaternion), three sin/cos pairs (half target angles) and a few reciprocals
// and square roots.  Library (ocml) versions carry large-argument paths and IEEE
// division sequences that cost ~800 VALU instructions per drone-step; these versions
// are branch-free minimax polynomials (Cephes single-precision coefficients) with
// absolute error <= ~2e-7 on the ranges this path produces, ~100 instructions in all.
// The header compiles for the host too (tests/test_device_math_cpu.py checks every
// function against libm on dense grids), so it contains no HIP-only types.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__)
#define DSIM_HD __device__ __forceinline__
#define DSIM_RCP(x) __builtin_amdgcn_rcpf(x)       // v_rcp_f32, 1 ulp
#define DSIM_RSQ(x) __builtin_amdgcn_rsqf(x)       // v_rsq_f32, 1 ulp
#define DSIM_SQRT(x) __builtin_amdgcn_sqrtf(x)     // v_sqrt_f32, 1 ulp
#elif defined(__HIPCC__)
#define DSIM_HD __host__ __device__ inline
#define DSIM_RCP(x) (1.0f / (x))
#define DSIM_RSQ(x) (1.0f / sqrtf(x))
#define DSIM_SQRT(x) sqrtf(x)
#else
#define DSIM_HD static inline
#define DSIM_RCP(x) (1.0f / (x))
#define DSIM_RSQ(x) (1.0f / sqrtf(x))
#define DSIM_SQRT(x) sqrtf(x)
#endif

#define DSIM_PI 3.14159265358979323846f
#define DSIM_PI_2 1.57079632679489661923f
#define DSIM_PI_4 0.78539816339744830962f

// sin and cos of x, |x| <~ 1e4 rad (two-term Cody-Waite reduction to [-pi/4, pi/4]).
DSIM_HD void dsim_sincos(float x, float* s, float* c) {
  const float k = rintf(x * 0.636619772367581343f);            // x * 2/pi
  float r = fmaf(-k, 1.57079637050628662109375f, x);           // float(pi/2)
  r = fmaf(-k, -4.37113900018624283e-8f, r);                   // pi/2 - float(pi/2)
  const int q = (int)k;
  const float r2 = r * r;
  const float sp = fmaf(r * r2, fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  const float cp = fmaf(r2 * r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                        fmaf(-0.5f, r2, 1.0f));
  const float ss = (q & 1) ? cp : sp;
  const float cc = (q & 1) ? sp : cp;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}

// atan2(y, x), full quadrant handling; atan2(0, 0) = 0 like libm.
DSIM_HD float dsim_atan2(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const bool big = mn > 0.41421356237f * mx;                   // a = mn/mx > tan(pi/8)
  const float num = big ? mn - mx : mn;
  const float den = big ? mn + mx : mx;
  const float t = den > 0.0f ? num * DSIM_RCP(den) : 0.0f;      // (a-1)/(a+1) or a
  const float z = t * t;
  float p = fmaf(fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f) * z, t, t);
  p = big ? p + DSIM_PI_4 : p;
  p = ay > ax ? DSIM_PI_2 - p : p;
  p = x < 0.0f ? DSIM_PI - p : p;
  return copysignf(p, y);
}

// asin(x), |x| <= 1
DSIM_HD float dsim_asin(float x) {
  const float a = fabsf(x);
  const bool big = a > 0.5f;
  const float z = big ? 0.5f * (1.0f - a) : a * a;
  const float s = big ? DSIM_SQRT(z) : a;
  float p = fmaf(fmaf(fmaf(fmaf(fmaf(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z,
                      1.6666752422e-1f) * z, s, s);
  p = big ? DSIM_PI_2 - 2.0f * p : p;
  return copysignf(p, x);
}

// Deterministic elementary functions for the solver.
//
// Why: the reference's stage-2 L-BFGS run is chaotically sensitive to rounding (a 1e-14 relative difference in
// one cost evaluation grows ~10x every 10 evaluations, see DESIGN.md "Parity"), so converged results are only
// reproducible when every floating-point operation is.  +,-,*,/,sqrt,fma,floor,rint are IEEE-exact on both
// gfx950 and x86-64; libm's sin/cos/atan2 are not specified bit-for-bit.  These versions use only exact
// operations in a fixed order (build with -ffp-contract=off), so the GPU and the CPU lane-emulator build of the
// same source agree bit for bit.  Algorithms and coefficients: FreeBSD msun k_sin.c / k_cos.c / e_rem_pio2.c
// (medium-size path) / s_atan.c / e_atan2.c (Sun Microsystems, freely distributable).  Accuracy < 1 ulp for
// |x| up to ~1e5, which is what per-evaluation parity against the libm-based oracle needs.
#pragma once
#include <hip/hip_runtime.h>

namespace topay {

__device__ __forceinline__ double k_sin(double x, double y) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double z = x * x;
  const double w = z * z;
  const double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
  const double v = z * x;
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
__device__ __forceinline__ double k_cos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double z = x * x;
  const double w0 = z * z;
  const double r = z * (C1 + z * (C2 + z * C3)) + (w0 * w0) * (C4 + z * (C5 + z * C6));
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * r - x * y));
}

// sin and cos of x.  Branch-free (so that independent calls can be interleaved by the scheduler: a single wave per
// SIMD has nothing else to hide the ~15-deep dependent chain behind).
__device__ __forceinline__ void det_sincos(double x, double* sn, double* cs) {
  const bool bad = !(fabs(x) < 1.0e15);  // inf / nan / absurd: propagate a NaN deterministically
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
               pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
  const double xs = bad ? 0.0 : x;
  const double fn = rint(xs * invpio2);
  // two-step Cody-Waite reduction (e_rem_pio2.c, second iteration form)
  const double t = xs - fn * pio2_1;
  double w = fn * pio2_2;
  const double r = t - w;
  w = fn * pio2_2t - ((t - r) - w);
  const double y0 = r - w;
  const double y1 = (r - y0) - w;
  const double s = k_sin(y0, y1), c = k_cos(y0, y1);
  const int n = (int)((long long)fn & 3);
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);
  const double so = (n == 0) ? s : (n == 1) ? c : (n == 2) ? -s : -c;
  const double co = (n == 0) ? c : (n == 1) ? -s : (n == 2) ? -c : s;
  *sn = bad ? nanv : so;
  *cs = bad ? nanv : co;
}

// sin and cos of NA angles, step by step across the angles: the arithmetic of an angle is that of det_sincos (same
// operations, same order), but each of the seventeen 64-bit constants is then live for NA adjacent instructions instead of
// being formed anew for every angle (f64 instructions of gfx950 take no 64-bit literal: two scalar moves per use; the
// manipulator block takes eight sines and cosines per sample).
// A 64-bit constant held in a scalar register pair (the empty asm hides that it is a constant, so it cannot be formed anew
// at each use -- two s_mov_b32 each time, since f64 instructions of gfx950 take no 64-bit literal).
__device__ __forceinline__ double topay_hold_f64(double v) {
#ifndef TOPAY_CPU_EMU
  asm("" : "+s"(v));
#endif
  return v;
}
// keeps the machine scheduler from undoing the step-by-step order below (it would rather finish one angle at a time)
#ifndef TOPAY_CPU_EMU
#define TOPAY_STEP_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define TOPAY_STEP_FENCE() do { } while (0)
#endif
template <int NA>
__device__ __forceinline__ void det_sincos_n(const double (&xin)[NA], double (&sn)[NA], double (&cs)[NA]) {
  // (held in scalar register pairs, topay_hold_f64: the compiler forms a 64-bit constant anew at every use otherwise)
  const double invpio2 = topay_hold_f64(6.36619772367581382433e-01), pio2_1 = topay_hold_f64(1.57079632673412561417e+00),
               pio2_2 = topay_hold_f64(6.07710050630396597660e-11), pio2_2t = topay_hold_f64(2.02226624879595063154e-21);
  const double S1 = topay_hold_f64(-1.66666666666666324348e-01), S2 = topay_hold_f64(8.33333333332248946124e-03),
               S3 = topay_hold_f64(-1.98412698298579493134e-04), S4 = topay_hold_f64(2.75573137070700676789e-06),
               S5 = topay_hold_f64(-2.50507602534068634195e-08), S6 = topay_hold_f64(1.58969099521155010221e-10);
  const double C1 = topay_hold_f64(4.16666666666666019037e-02), C2 = topay_hold_f64(-1.38888888888741095749e-03),
               C3 = topay_hold_f64(2.48015872894767294178e-05), C4 = topay_hold_f64(-2.75573143513906633035e-07),
               C5 = topay_hold_f64(2.08757232129817482790e-09), C6 = topay_hold_f64(-1.13596475577881948265e-11);
  bool bad[NA];
  double fn[NA], y0[NA], y1[NA];
  {
    double xs[NA], t[NA], w[NA], r[NA];
#pragma unroll
    for (int a = 0; a < NA; a++) { bad[a] = !(fabs(xin[a]) < 1.0e15); xs[a] = bad[a] ? 0.0 : xin[a]; }
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) fn[a] = rint(xs[a] * invpio2);
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) t[a] = xs[a] - fn[a] * pio2_1;
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) w[a] = fn[a] * pio2_2;
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) r[a] = t[a] - w[a];
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) w[a] = fn[a] * pio2_2t - ((t[a] - r[a]) - w[a]);
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) { y0[a] = r[a] - w[a]; y1[a] = (r[a] - y0[a]) - w[a]; }
    TOPAY_STEP_FENCE();
  }
  double s[NA], c[NA];
  {
    // k_sin(y0, y1):  r = S2 + z (S3 + z S4) + z w (S5 + z S6);  v = z x;  x - ((z (0.5 y - v r) - y) - v S1)
    double z[NA], p[NA], q[NA];
#pragma unroll
    for (int a = 0; a < NA; a++) z[a] = y0[a] * y0[a];
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) p[a] = z[a] * S4;
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) p[a] = S3 + p[a];
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) p[a] = S2 + z[a] * p[a];            // S2 + z (S3 + z S4)
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) q[a] = z[a] * S6;
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) q[a] = S5 + q[a];
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) {
      const double w = z[a] * z[a];
      const double r = p[a] + z[a] * w * q[a];
      const double v = z[a] * y0[a];
      p[a] = (z[a] * (0.5 * y1[a] - v * r) - y1[a]);
      q[a] = v;
    }
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) s[a] = y0[a] - (p[a] - q[a] * S1);
    TOPAY_STEP_FENCE();
    // k_cos(y0, y1):  r = z (C1 + z (C2 + z C3)) + (w0 w0) (C4 + z (C5 + z C6));  hz = 0.5 z;  w = 1 - hz;
    //                 w + (((1 - w) - hz) + (z r - x y))
#pragma unroll
    for (int a = 0; a < NA; a++) p[a] = z[a] * C3;
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) p[a] = C2 + p[a];
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) p[a] = C1 + z[a] * p[a];
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) q[a] = z[a] * C6;
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) q[a] = C5 + q[a];
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) q[a] = C4 + z[a] * q[a];
    TOPAY_STEP_FENCE();
#pragma unroll
    for (int a = 0; a < NA; a++) {
      const double w0 = z[a] * z[a];
      const double r = z[a] * p[a] + (w0 * w0) * q[a];
      const double hz = 0.5 * z[a];
      const double w = 1.0 - hz;
      c[a] = w + (((1.0 - w) - hz) + (z[a] * r - y0[a] * y1[a]));
    }
    TOPAY_STEP_FENCE();
  }
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);
#pragma unroll
  for (int a = 0; a < NA; a++) {
    const int n = (int)((long long)fn[a] & 3);
    const double so = (n == 0) ? s[a] : (n == 1) ? c[a] : (n == 2) ? -s[a] : -c[a];
    const double co = (n == 0) ? c[a] : (n == 1) ? -s[a] : (n == 2) ? -c[a] : s[a];
    sn[a] = bad[a] ? nanv : so;
    cs[a] = bad[a] ? nanv : co;
  }
}

__device__ __forceinline__ double det_atan(double xin) {
  const double aT[11] = {3.33333333333329318027e-01,  -1.99999999998764832476e-01, 1.42857142725034663711e-01,
                         -1.11111104054623557880e-01, 9.09088713343650656196e-02,  -7.69187620504482999495e-02,
                         6.66107313738753120669e-02,  -5.83357013379057348645e-02, 4.97687799461593236017e-02,
                         -3.65315727442169155270e-02, 1.62858201153657823623e-02};
  const bool neg = xin < 0.0;
  double x = fabs(xin);
  if (x != x) return x;
  double hi = 0.0, lo = 0.0;
  int id = -1;
  if (x >= 7.378697629483821e19) {  // 2^66
    const double r = 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
    return neg ? -r : r;
  }
  if (x < 0.4375) {
    if (x < 7.450580596923828e-09) return xin;  // 2^-27
  } else if (x < 1.1875) {
    if (x < 0.6875) { id = 0; hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; x = (2.0 * x - 1.0) / (2.0 + x); }
    else { id = 1; hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; x = (x - 1.0) / (x + 1.0); }
  } else {
    if (x < 2.4375) { id = 2; hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; x = (x - 1.5) / (1.0 + 1.5 * x); }
    else { id = 3; hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; x = -1.0 / x; }
  }
  const double z = x * x;
  const double w = z * z;
  const double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  const double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) {
    const double r = x - x * (s1 + s2);
    return neg ? -r : r;
  }
  const double r = hi - ((x * (s1 + s2) - lo) - x);
  return neg ? -r : r;
}

__device__ __forceinline__ double det_atan2(double y, double x) {
  const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
  if (x != x || y != y) return x + y;
  const bool sy = y < 0.0 || (y == 0.0 && 1.0 / y < 0.0);
  const bool sx = x < 0.0 || (x == 0.0 && 1.0 / x < 0.0);
  if (y == 0.0) return sx ? (sy ? -pi : pi) : y;
  if (x == 0.0) return sy ? -0.5 * pi : 0.5 * pi;
  const double z = det_atan(fabs(y / x));
  if (!sx) return sy ? -z : z;
  return sy ? (z - pi_lo) - pi : pi - (z - pi_lo);
}

}  // namespace topay

// Per-point operations of the sweep, restated for the device.
//
// Each function follows the corresponding function of the reference's `mod ops`
// (src/anguelova.rs:99-171) operation by operation and in the same order.  rustc never
// contracts a*b+c into an FMA, so contraction is switched off for this file's functions:
// given identical (V, v00, v10, v11, |dV|^2) inputs the results differ from the reference's
// only through atan/tan (libm vs OCML, <= 1-2 ulp).
#pragma once
#include "inflx_device_math.h"

struct InflxModelValues {
  double V;    // potential                       (C symbol `V`)
  double v00;  // Hesse component along (v, v)     (C symbol `v00`)
  double v10;  // Hesse component along (w, v)     (C symbol `v10`, Hesse2D::v10 = fns[2])
  double v11;  // Hesse component along (w, w)     (C symbol `v11`)
  double g;    // |grad V|^2                       (C symbol `grad_norm_squared`)
  double b0;   // basis vector v, component 0      (C symbol `v`, v_out[0]; Potential::grad)
  double b1;   // basis vector v, component 1      (v_out[1])
};

// atan and tan of the epilogue (src/anguelova.rs:128,132: f64::atan / f64::tan, i.e. the platform libm).  On the
// device these are OCML's algorithms, operation for operation and constant for constant (ROCm 7.2 ocml.bc:
// __ocml_atan_f64 / __ocmlpriv_atanred_f64, __ocml_tan_f64 / __ocmlpriv_trigredsmall_f64 / __ocmlpriv_tanred2_f64),
// restricted to the arguments the epilogue can produce -- delta = atan(|v10/v00|) >= 0, and tan(delta) with
// 0 <= delta <= pi/2 -- so that everything the general entry points spend on other arguments goes away: sign
// handling, the Payne-Hanek branch for |x| >= 2^30, the infinity test.  Bit for bit OCML's results on that domain
// (tests/test_epilogue_math_gpu.py compares them on the device); NaN in, NaN out.  The host twin keeps libm.
#ifndef INFLX_HOST_TWIN
// One Horner step.  (Measured on MI355X, A/B in one session, scripts/hoist_experiment.py: spelling the step as inline
// `v_fma_f64` with the coefficient in a scalar register pair -- which avoids the v_mov_b64 + v_fmac_f64 pairs the
// compiler emits where registers are short, D5: 32 of them -- costs s_mov/s_nop hazard slots instead and is not
// faster overall: D5 0.505 / EGNO 0.453 / doc 0.244 / angular 0.279 ms against 0.497 / 0.435 / 0.244 / 0.289 as
// written here; with the coefficient pinned in a vector register 0.484 / 0.442 / 0.245 / 0.287.  Within the noise:
// the plain form stays.)
INFLX_FN double inflx_horner(double s, double acc, double coefficient) { return __builtin_fma(s, acc, coefficient); }

INFLX_FN double inflx_atan_nonneg(double t) {
#pragma clang fp contract(off)
  const bool big = t > 1.0;
  const double x = big ? 1.0 / t : t;
  const double s = x * x;
  double p = __builtin_fma(s, 0x1.ba404b5e68a13p-17, -0x1.3e260bd3237f4p-13);
  p = inflx_horner(s, p, 0x1.b2bb069efb384p-11);
  p = inflx_horner(s, p, -0x1.7952daf56de9bp-9);
  p = inflx_horner(s, p, 0x1.d6d43a595c56fp-8);
  p = inflx_horner(s, p, -0x1.c6ea4a57d9582p-7);
  p = inflx_horner(s, p, 0x1.67e295f08b19fp-6);
  p = inflx_horner(s, p, -0x1.e9ae6fc27006ap-6);
  p = inflx_horner(s, p, 0x1.2c15b5711927ap-5);
  p = inflx_horner(s, p, -0x1.59976e82d3ff0p-5);
  p = inflx_horner(s, p, 0x1.82d5d6ef28734p-5);
  p = inflx_horner(s, p, -0x1.ae5ce6a214619p-5);
  p = inflx_horner(s, p, 0x1.e1bb48427b883p-5);
  p = inflx_horner(s, p, -0x1.110e48b207f05p-4);
  p = inflx_horner(s, p, 0x1.3b13657b87036p-4);
  p = inflx_horner(s, p, -0x1.745d119378e4fp-4);
  p = inflx_horner(s, p, 0x1.c71c717e1913cp-4);
  p = inflx_horner(s, p, -0x1.2492492376b7dp-3);
  p = inflx_horner(s, p, 0x1.99999999952ccp-3);
  p = inflx_horner(s, p, -0x1.5555555555523p-2);
  const double a = __builtin_fma(x, s * p, x);
  // pi/2 - a, with pi/2 as the product OCML uses (0x1.dd9ad336a0500p-1 * 0x1.af154eeb562d6p+0, one rounding)
  return big ? __builtin_fma(0x1.dd9ad336a0500p-1, 0x1.af154eeb562d6p+0, -a) : a;
}

INFLX_FN double inflx_tan_quadrant1(double x) {
#pragma clang fp contract(off)
  // argument reduction (trigredsmall): n = rint(x * 2/pi), (r, rr) = x - n*pi/2 as a double-double
  const double dn = __builtin_rint(x * 0x1.45f306dc9c883p-1);
  const double a = __builtin_fma(dn, -0x1.921fb54442d18p+0, x);
  const double b = __builtin_fma(dn, -0x1.1a62633145c00p-54, a);
  const double ph = dn * 0x1.1a62633145c00p-54;
  const double pt = __builtin_fma(dn, 0x1.1a62633145c00p-54, -ph);
  const double th = a - ph;
  const double c = ((th - b) + ((a - th) - ph)) - pt;
  const double d = __builtin_fma(dn, -0x1.b839a252049c0p-104, c);
  const double r = b + d;
  const double rr = d - (r - b);
  // tanred2
  const double s0 = r * r;
  const double s = s0 + __builtin_fma(r, rr * 2.0, __builtin_fma(r, r, -s0));
  double p = __builtin_fma(s, 0x1.5e089c751c08cp-16, -0x1.78809a9a29f71p-15);
  p = inflx_horner(s, p, 0x1.7746f90a8aae0p-14);
  p = inflx_horner(s, p, -0x1.bb44da6fbf144p-16);
  p = inflx_horner(s, p, 0x1.1e634a7943acfp-13);
  p = inflx_horner(s, p, 0x1.d250fdeb68febp-13);
  p = inflx_horner(s, p, 0x1.37fd9b58c4d95p-11);
  p = inflx_horner(s, p, 0x1.7d5af15120e2cp-10);
  p = inflx_horner(s, p, 0x1.d6d93e09491dfp-9);
  p = inflx_horner(s, p, 0x1.226e12033784dp-7);
  p = inflx_horner(s, p, 0x1.664f49ac36ae2p-6);
  p = inflx_horner(s, p, 0x1.ba1ba1b451c21p-5);
  p = inflx_horner(s, p, 0x1.11111111185b7p-3);
  p = inflx_horner(s, p, 0x1.55555555554eep-2);
  const double u = s * p;
  const double e = r * u;
  const double el = __builtin_fma(r, u, -e);
  const double f = r + e;
  const double lo = (rr + el) + (e - (f - r));
  const double t = f + lo;        // tan of the reduced argument
  const double tl = lo - (t - f);
  // -1/t with the low part (the result when n is odd)
  double y = __builtin_amdgcn_rcp(t);
  y = __builtin_fma(__builtin_fma(-t, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-t, y, 1.0), y, y);
  const double g = t * y;
  const double gl = __builtin_fma(y, tl, __builtin_fma(y, t, -g));
  const double h = g + gl;
  const double hl = gl - (h - g);
  const double k = 1.0 - h;
  const double m = k + ((((1.0 - k) - h)) - hl);
  const double w = y + y * m;
  return dn == 0.0 ? t : -w;  // 0 <= x <= pi/2: n is 0 or 1
}
#else
INFLX_FN double inflx_atan_nonneg(double t) { return atan(t); }
INFLX_FN double inflx_tan_quadrant1(double x) { return tan(x); }
#endif

INFLX_FN double inflx_sq(double x) {
#pragma clang fp contract(off)
  return x * x;  // f64::powi(2)
}

// ops::complete_analysis, src/anguelova.rs:103-135.
// out[0] consistency, [1] epsilon_V, [2] epsilon_H, [3] eta_parallel, [4] delta, [5] omega
INFLX_FN void inflx_op_complete_analysis(const InflxModelValues& m, double out[6]) {
#pragma clang fp contract(off)
  const double v = m.V, v11 = m.v11, v10 = m.v10, v00 = m.v00;
  double consistency;
  {
    const double lhs = v11 / v;
    const double rhs = 3. + 3. * inflx_sq(v00 / v10) + (v00 / v) * inflx_sq(v10 / v00);
    consistency = fabs(lhs - rhs) / (fabs(lhs) + fabs(rhs));
  }
  const double epsilon_v = m.g / inflx_sq(v);
  const double vtt = (v00 * inflx_sq(v10) + v11 * inflx_sq(v00) - 2. * v00 * inflx_sq(v10)) / (inflx_sq(v00) + inflx_sq(v10));
  const double vt2 = epsilon_v * (1. / (1. + inflx_sq(v00 / v10)));
  // The reference divides twice by v here: |vtt| / v (below) and vtt / v (omega).  Rounding to nearest is symmetric in
  // the sign, so |vtt| / v is exactly |vtt / v| with the sign of v (zeros, infinities and NaN included): one IEEE
  // division instead of two, not a bit changed.
  const double vtt_over_v = vtt / v;
  const double abs_vtt_over_v = __builtin_copysign(fabs(vtt_over_v), v);
  const double epsilon_h = 3. * (epsilon_v - vt2) * (1. / (epsilon_v + abs_vtt_over_v - vt2));
  const double delta = inflx_atan_nonneg(fabs(v10 / v00));
  const double omega = sqrt(vtt_over_v * (3. - epsilon_h));
  const double eta_parallel = omega * inflx_tan_quadrant1(delta) - 3.;
  out[0] = consistency;
  out[1] = epsilon_v;
  out[2] = epsilon_h;
  out[3] = eta_parallel;
  out[4] = delta;
  out[5] = omega;
}

// ops::epsilon_v_only, src/anguelova.rs:138-140 (note the 1/2 that complete_analysis lacks)
INFLX_FN double inflx_op_epsilon_v_only(const InflxModelValues& m) {
#pragma clang fp contract(off)
  return 0.5 * m.g / inflx_sq(m.V);
}

// ops::consistency_rapidturn_only, src/anguelova.rs:143-154
INFLX_FN double inflx_op_consistency_rapidturn_only(const InflxModelValues& m) {
#pragma clang fp contract(off)
  const double lhs = m.v11 / m.V;
  const double rhs = 3. * inflx_sq(m.v10 / m.v00);
  return fabs(fabs(lhs) - fabs(rhs)) / (fabs(lhs) + fabs(rhs));
}

// ops::consistency_only, src/anguelova.rs:157-163
INFLX_FN double inflx_op_consistency_only(const InflxModelValues& m) {
#pragma clang fp contract(off)
  const double lhs = m.v11 / m.V - 3.;
  const double rhs = 3. * inflx_sq(m.v00 / m.v10) + (m.v00 / m.V) * inflx_sq(m.v10 / m.v00);
  return fabs(fabs(lhs) - fabs(rhs)) / (fabs(lhs) + fabs(rhs));
}

// ops::flag_quantum_diff, src/anguelova.rs:166-170: all components of the normalised gradient
// <= accuracy (no abs(); a NaN component makes the flag false, as in Rust)
INFLX_FN bool inflx_op_flag_quantum_diff(const InflxModelValues& m, double accuracy) {
  return (m.b0 <= accuracy) && (m.b1 <= accuracy);
}

// index -> field-space coordinate, src/anguelova.rs:514-516,531-533: (idx as f64) * spacing + offset,
// multiply then add (two roundings).
INFLX_FN double inflx_coord(unsigned long long idx, double spacing, double offset) {
#pragma clang fp contract(off)
  return (double)idx * spacing + offset;
}

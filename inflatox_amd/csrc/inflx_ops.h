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
  const double epsilon_h = 3. * (epsilon_v - vt2) * (1. / (epsilon_v + fabs(vtt) / v - vt2));
  const double delta = atan(fabs(v10 / v00));
  const double omega = sqrt((vtt / v) * (3. - epsilon_h));
  const double eta_parallel = omega * tan(delta) - 3.;
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

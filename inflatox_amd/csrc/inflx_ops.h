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
  double v01;  // Hesse component along (v, w)     (C symbol `v01`; filled for INFLX_OP_HESSE only -- Hesse2D never calls it on a sweep)
};

// atan and tan of the epilogue (src/anguelova.rs:128,132: f64::atan / f64::tan, i.e. the platform libm).  On the
// device these are OCML's algorithms, operation for operation and constant for constant (ROCm 7.2 ocml.bc:
// __ocml_atan_f64 / __ocmlpriv_atanred_f64, __ocml_tan_f64 / __ocmlpriv_trigredsmall_f64 / __ocmlpriv_tanred2_f64),
// restricted to the arguments the epilogue can produce -- delta = atan(|v10/v00|) >= 0, and tan(delta) with
// 0 <= delta <= pi/2 -- so that everything the general entry points spend on other arguments goes away: sign
// handling, the Payne-Hanek branch for |x| >= 2^30, the infinity test.  Bit for bit OCML's results on that domain
// (tests/test_epilogue_math_gpu.py compares them on the device); NaN in, NaN out.  The host twin keeps libm.
#ifndef INFLX_HOST_TWIN
// The coefficients of OCML's two polynomials (atan: degree 19 in x^2, leading one first; tan: degree 13 in r^2).
constexpr int kInflxAtanTerms = 20, kInflxTanTerms = 14, kInflxEpilogueConstants = kInflxAtanTerms + kInflxTanTerms;
constexpr double kInflxAtanC[kInflxAtanTerms] = {
    0x1.ba404b5e68a13p-17,  -0x1.3e260bd3237f4p-13, 0x1.b2bb069efb384p-11, -0x1.7952daf56de9bp-9,  0x1.d6d43a595c56fp-8,
    -0x1.c6ea4a57d9582p-7,  0x1.67e295f08b19fp-6,   -0x1.e9ae6fc27006ap-6, 0x1.2c15b5711927ap-5,   -0x1.59976e82d3ff0p-5,
    0x1.82d5d6ef28734p-5,   -0x1.ae5ce6a214619p-5,  0x1.e1bb48427b883p-5,  -0x1.110e48b207f05p-4,  0x1.3b13657b87036p-4,
    -0x1.745d119378e4fp-4,  0x1.c71c717e1913cp-4,   -0x1.2492492376b7dp-3, 0x1.99999999952ccp-3,   -0x1.5555555555523p-2};
constexpr double kInflxTanC[kInflxTanTerms] = {
    0x1.5e089c751c08cp-16, -0x1.78809a9a29f71p-15, 0x1.7746f90a8aae0p-14, -0x1.bb44da6fbf144p-16, 0x1.1e634a7943acfp-13,
    0x1.d250fdeb68febp-13, 0x1.37fd9b58c4d95p-11,  0x1.7d5af15120e2cp-10, 0x1.d6d93e09491dfp-9,   0x1.226e12033784dp-7,
    0x1.664f49ac36ae2p-6,  0x1.ba1ba1b451c21p-5,   0x1.11111111185b7p-3,  0x1.55555555554eep-2};

// Where the 34 coefficients live while a kernel evaluates the polynomials (template switch TABLE: read them from the
// table at `kc`; else literals).
// A Horner step is p = fma(s, p, C) with C uniform.  gfx950 has no 64-bit literal operands, so C sits in a register
// pair: as literals the compiler keeps all 34 in VECTOR registers across the row loop of the tile kernels (68 VGPRs --
// the difference between two and three wavefronts per SIMD for the heavy models) and, where registers are short,
// spells every step v_mov_b64 + v_fmac_f64 (measured with the SQ counters: +30 VALU instructions per grid point); in
// scalar registers (68 SGPRs) it spills them to VGPR lanes.  The tile kernels therefore keep the table in LDS and pass
// its address: one ds_read_b64 per step with a uniform address (2 LDS cycles per wave-instruction, the LDS pipe is
// otherwise nearly idle here) whose destination is the accumulator of v_fmac_f64 -- no move, no resident register.
// INFLX_HORNER_MODE spells the step for literal coefficients (kernels without the table): 0 plain, 1 the coefficient in
// a scalar register pair, 2 in a vector register pair with a three-address FMA (experiments; profiles/r03_experiments.txt).
#ifndef INFLX_HORNER_MODE
#define INFLX_HORNER_MODE 0
#endif
INFLX_FN double inflx_horner(double s, double acc, double coefficient) {
#if INFLX_HORNER_MODE == 1
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(s), "v"(acc), "s"(coefficient));
  return r;
#elif INFLX_HORNER_MODE == 2
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(s), "v"(acc), "v"(coefficient));
  return r;
#else
  return __builtin_fma(s, acc, coefficient);
#endif
}
// the polynomial C[0] s^(N-1) + ... + C[N-1] by Horner's rule, coefficients from `kc` (LDS) when given
template <bool TABLE, int N>
INFLX_FN double inflx_polynomial(double s, const double (&literal)[N], [[maybe_unused]] const double* kc) {
  double p;
  if constexpr (TABLE) {  // (a compile-time switch: the address of an LDS object may be 0, so `kc != NULL` says nothing)
    p = __builtin_fma(s, kc[0], kc[1]);
#pragma unroll
    for (int k = 2; k < N; ++k) p = __builtin_fma(s, p, kc[k]);
  } else {
    p = __builtin_fma(s, literal[0], literal[1]);
#pragma unroll
    for (int k = 2; k < N; ++k) p = inflx_horner(s, p, literal[k]);
  }
  return p;
}

// ---- the IEEE division without its special-case handling ------------------------------------------------
// The compiler spells a/b as: v_div_scale x2, v_rcp_f64, two Newton steps on the reciprocal (4 FMAs), q0 = a*y,
// r = fma(-b, q0, a), v_div_fmas (= fma(r, y, q0) when nothing was scaled), v_div_fixup -- 11 instructions, 13.0
// fma-equivalents of issue time (scripts/micro/valu_rates.hip).  v_div_scale only changes an operand when an exponent
// is extreme (denominator or quotient next to the denormal range or to overflow, numerator below 2^-969) and
// v_div_fixup only changes the result for zero / infinite / NaN operands.  For operands that are known to be in mid
// range the quotient is therefore EXACTLY what the eight arithmetic instructions in between deliver -- the functions
// below are those eight instructions, nothing else, so their result is the IEEE quotient bit for bit -- and
// quotients with the same denominator share the five instructions of the reciprocal.  The epilogue establishes "mid
// range" once per point for its five inputs (inflx_op_complete_analysis_quick); a point that fails is evaluated with
// the compiler's divisions instead.
INFLX_FN double inflx_rcp_newton2(double b) {
  const double y0 = __builtin_amdgcn_rcp(b);
  const double y1 = __builtin_fma(y0, __builtin_fma(-b, y0, 1.0), y0);
  return __builtin_fma(y1, __builtin_fma(-b, y1, 1.0), y1);
}
// a / b, given y = inflx_rcp_newton2(b)
INFLX_FN double inflx_quotient(double a, double b, double y) {
#pragma clang fp contract(off)
  const double q0 = a * y;
  return __builtin_fma(__builtin_fma(-b, q0, a), y, q0);
}
// 1.0 / b (q0 = 1.0 * y is y itself)
INFLX_FN double inflx_reciprocal_quick(double b) {
  const double y = inflx_rcp_newton2(b);
  return __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
}
// biased exponent field of a double: 0 for zeros and denormals, 2047 for infinities and NaNs
INFLX_FN unsigned inflx_exponent_field(double x) { return ((unsigned)__double2hiint(x) >> 20) & 0x7ffu; }
// (inflx_magnitude_word, inflx_field_word, inflx_sqrt_quick and inflx_sqrt_quick_ok: inflx_device_math.h)

// Wave-uniform specialisation (both functions below).  Along a grid row delta = atan|v10/v00| varies smoothly, so the 64
// lanes of a wavefront nearly always sit on the same side of t = 1 (delta = pi/4): measured on the 4096-column grids of
// the example models, 98 % of the wavefronts of the D5 sweep have t <= 1 in every lane, EGNO 52 % (46 % t > 1
// everywhere), angular 63 % / 28 %, doc 28 % / 70 %.  The branches of OCML's algorithms that a lane does not take are
// then taken by NO lane, and a scalar branch on a ballot skips them: atan's reciprocal for t <= 1; tan's argument
// reduction and its -1/t tail for delta < pi/4 (n = 0).  Every lane still executes exactly the operations OCML executes
// for its argument -- the results are the same bits whichever branch the wavefront takes
// (tests/test_epilogue_math_gpu.py feeds sorted, i.e. wave-uniform, and shuffled arguments).
template <bool TABLE>
INFLX_FN double inflx_atan_core(double x, const double* kc) {
#pragma clang fp contract(off)
  const double s = x * x;
  const double p = inflx_polynomial<TABLE>(s, kInflxAtanC, kc);
  return __builtin_fma(x, s * p, x);
}

template <bool QUICK = false, bool TABLE = false>
INFLX_FN double inflx_atan_nonneg(double t, const double* kc = nullptr) {
#pragma clang fp contract(off)
  const bool big = t > 1.0;
  double x = t;
  if (__builtin_amdgcn_ballot_w64(big) != 0) {  // wave-uniform: some lane needs 1/t
    // QUICK: the caller guarantees 2^-241 < t < 2^241 (or accepts nothing of this point)
    const double inv = QUICK ? inflx_reciprocal_quick(t) : 1.0 / t;
    x = big ? inv : t;
  }
  const double a = inflx_atan_core<TABLE>(x, kc);
  // pi/2 - a, with pi/2 as the product OCML uses (0x1.dd9ad336a0500p-1 * 0x1.af154eeb562d6p+0, one rounding)
  return big ? __builtin_fma(0x1.dd9ad336a0500p-1, 0x1.af154eeb562d6p+0, -a) : a;
}

// tanred2 of OCML: tan(r + rr) for the reduced argument as (t, tl), t + tl to about 2^-100 relative
struct InflxTanRed {
  double t, tl;
};
template <bool RR_IS_ZERO, bool TABLE>
INFLX_FN InflxTanRed inflx_tanred2(double r, double rr, const double* kc) {
#pragma clang fp contract(off)
  const double s0 = r * r;
  const double s_lo = __builtin_fma(r, r, -s0);  // r*r - s0 exactly (+0 when it vanishes)
  // rr = +0: fma(r, rr*2, s_lo) is s_lo itself (NaN and infinite r included: s_lo is NaN then as well)
  const double s = s0 + (RR_IS_ZERO ? s_lo : __builtin_fma(r, rr * 2.0, s_lo));
  const double p = inflx_polynomial<TABLE>(s, kInflxTanC, kc + kInflxAtanTerms);
  const double u = s * p;
  const double e = r * u;
  const double el = __builtin_fma(r, u, -e);  // exact residual: never -0
  const double f = r + e;
  // rr = +0: (rr + el) is el
  const double lo = (RR_IS_ZERO ? el : rr + el) + (e - (f - r));
  const double t = f + lo;  // tan of the reduced argument
  return InflxTanRed{t, lo - (t - f)};
}

template <bool TABLE = false>
INFLX_FN double inflx_tan_quadrant1(double x, const double* kc = nullptr) {
#pragma clang fp contract(off)
  // argument reduction (trigredsmall): n = rint(x * 2/pi), (r, rr) = x - n*pi/2 as a double-double
  const double dn = __builtin_rint(x * 0x1.45f306dc9c883p-1);
  // n = 0 in every lane (x < pi/4): the reduction returns r = x, rr = +0 exactly (every product with dn is a zero
  // and every sum of them +0), and the -1/t tail below is not wanted by any lane
  if (__builtin_amdgcn_ballot_w64(dn != 0.0) == 0) return inflx_tanred2<true, TABLE>(x, 0.0, kc).t;
  const double a = __builtin_fma(dn, -0x1.921fb54442d18p+0, x);
  const double b = __builtin_fma(dn, -0x1.1a62633145c00p-54, a);
  const double ph = dn * 0x1.1a62633145c00p-54;
  const double pt = __builtin_fma(dn, 0x1.1a62633145c00p-54, -ph);
  const double th = a - ph;
  const double c = ((th - b) + ((a - th) - ph)) - pt;
  const double d = __builtin_fma(dn, -0x1.b839a252049c0p-104, c);
  const double r = b + d;
  const double rr = d - (r - b);
  const InflxTanRed q = inflx_tanred2<false, TABLE>(r, rr, kc);
  const double t = q.t, tl = q.tl;
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
template <bool QUICK = false, bool TABLE = false>
INFLX_FN double inflx_atan_nonneg(double t, const double* = nullptr) { return atan(t); }
template <bool TABLE = false>
INFLX_FN double inflx_tan_quadrant1(double x, const double* = nullptr) { return tan(x); }
#endif

// tan(delta) for delta = atan(t) (src/anguelova.rs:128,132).  Mathematically that is t; the reference evaluates the
// two libm calls one after the other and gets t(1 + e), |e| <~ (t + 1/t) 2^-53 (the rounding of delta, amplified by
// the slope of the tangent) + the rounding of tan itself.  Compiler(tan_shortcut=T) (-DINFLX_TAN_SHORTCUT_MAX=T)
// returns t itself wherever t <= T, i.e. a value that differs from the reference's by at most ~(T + 1) 2^-53
// relative -- inside the 1e-10 bar for any T below 2^17, and closer to the exact value than the reference's own --
// and evaluates tan(delta) above T.  The rule is a function of the point alone (the same value whichever path or
// wavefront evaluates it).  Off (0) by default: the default results are OCML's tan of OCML's atan, bit for bit.
#ifndef INFLX_TAN_SHORTCUT_MAX
#define INFLX_TAN_SHORTCUT_MAX 0
#endif
template <bool TABLE = false>
INFLX_FN double inflx_tan_of_atan(double t, double delta, const double* kc = nullptr) {
#if INFLX_TAN_SHORTCUT_MAX > 0
  const bool small = t <= (double)(INFLX_TAN_SHORTCUT_MAX);  // false for NaN
#ifndef INFLX_HOST_TWIN
  if (__builtin_amdgcn_ballot_w64(!small) == 0) return t;  // wave-uniform: the whole wavefront skips the tangent
#endif
  const double full = inflx_tan_quadrant1<TABLE>(delta, kc);
  return small ? t : full;
#else
  (void)t;
  return inflx_tan_quadrant1<TABLE>(delta, kc);
#endif
}

INFLX_FN double inflx_sq(double x) {
#pragma clang fp contract(off)
  return x * x;  // f64::powi(2)
}

// ops::complete_analysis, src/anguelova.rs:103-135.
// out[0] consistency, [1] epsilon_V, [2] epsilon_H, [3] eta_parallel, [4] delta, [5] omega
template <bool TABLE = false>
INFLX_FN void inflx_op_complete_analysis(const InflxModelValues& m, double out[6], const double* kc = nullptr) {
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
  const double t = fabs(v10 / v00);
  const double delta = inflx_atan_nonneg<false, TABLE>(t, kc);
  const double omega = sqrt(vtt_over_v * (3. - epsilon_h));
  const double eta_parallel = omega * inflx_tan_of_atan<TABLE>(t, delta, kc) - 3.;
  out[0] = consistency;
  out[1] = epsilon_v;
  out[2] = epsilon_h;
  out[3] = eta_parallel;
  out[4] = delta;
  out[5] = omega;
}

// The same function with its eleven divisions spelled as inflx_quotient / inflx_reciprocal_quick: 8 (7) instructions
// each instead of 11, and one reciprocal of V for the three quotients by V.  Returns whether the point qualified; when
// it did not, `out` is unspecified and the caller evaluates inflx_op_complete_analysis instead (the tile kernels do
// that for the whole grid row of the wavefront, after their hot loop).
//
// When is every division of the epilogue a mid-range division?  Let all five inputs be normal numbers with
// 2^-120 <= |x| < 2^121 (ONE test: the smallest and the largest exponent field).  Then, division by division
// (a = numerator, b = denominator, q = quotient; needed: b and 1/b normal, |a| >= 2^-969 or a = +0, q normal or +0):
//   v11/V, v00/v10, v10/v00, v00/V     a, b in range, |q| in (2^-241, 2^241)
//   |lhs-rhs| / (|lhs|+|rhs|)          b >= |lhs| > 2^-241, b < 2^241 + 3 + 3*2^482 + 2^241*2^482 < 2^724; lhs and rhs are
//                                      multiples of 2^-293 (rhs = (3 + 3A^2) + x is 0 or >= 2^-52), so a is +0 or >= 2^-293
//                                      and q is +0 or in [2^-1017, 1]
//   g / V^2                            b in [2^-240, 2^242), |q| in (2^-362, 2^361)
//   N / (v00^2 + v10^2)                b in [2^-240, 2^243); N = (v00 v10^2 + v11 v00^2) - 2 v00 v10^2 is a sum of numbers
//                                      >= 2^-360: +0 (x - x rounds to +0) or >= 2^-412; |N| < 2^365; |q| in [2^-655, 2^605] or 0
//   1 / (1 + (v00/v10)^2)              b in [1, 2^483)
//   vtt / V                            a = vtt as above, |q| in [2^-776, 2^725] or (-)0: a = +0 gives the correctly signed zero
//                                      (q0 = +-0, r = +0, fma(+0, y, q0) = q0)
//   1 / (eps_V + |vtt/V| - vt2)        the one denominator that can cancel to anything: tested on its own, [2^-500, 2^501)
//   1 / t in atan, t = |v10/v00| > 1   b in (1, 2^241)
//   sqrt((vtt/V)(3 - eps_H))           not a division, but the same idea (inflx_sqrt_quick): tested on its own, 2^-767 <= |.| < inf
// A NaN, an infinity, a zero or a denormal among the inputs fails the test of the exponent fields.  On the 4096-column
// grids of the example models 97.5-98.4 % of the wavefronts qualify in every lane (exponents seen: doc -23..45, angular
// -112..-29, EGNO -64..-12, D5 -77..15); tests/test_epilogue_values_gpu.py compares this function with the IEEE one bit
// for bit on two million tuples that straddle every bound above.
template <bool TABLE = false>
INFLX_FN bool inflx_op_complete_analysis_quick(const InflxModelValues& m, double out[6], const double* kc = nullptr) {
#ifdef INFLX_HOST_TWIN
  (void)kc;
  inflx_op_complete_analysis(m, out);
  return true;
#else
#pragma clang fp contract(off)
  const double v = m.V, v11 = m.v11, v10 = m.v10, v00 = m.v00, g = m.g;
  const unsigned e0 = inflx_exponent_field(v), e1 = inflx_exponent_field(v11), e2 = inflx_exponent_field(v10), e3 = inflx_exponent_field(v00),
                 e4 = inflx_exponent_field(g);
  const unsigned lowest = min(min(e0, e1), min(min(e2, e3), e4)), highest = max(max(e0, e1), max(max(e2, e3), e4));
  bool ok = lowest >= 1023u - 120u && highest <= 1023u + 120u;
  const double yv = inflx_rcp_newton2(v);
  double consistency;
  const double a_over_b = inflx_quotient(v00, v10, inflx_rcp_newton2(v10));
  const double b_over_a = inflx_quotient(v10, v00, inflx_rcp_newton2(v00));
  {
    const double lhs = inflx_quotient(v11, v, yv);
    const double rhs = 3. + 3. * inflx_sq(a_over_b) + inflx_quotient(v00, v, yv) * inflx_sq(b_over_a);
    const double num = fabs(lhs - rhs), den = fabs(lhs) + fabs(rhs);
    consistency = inflx_quotient(num, den, inflx_rcp_newton2(den));
  }
  const double v2 = inflx_sq(v);
  const double epsilon_v = inflx_quotient(g, v2, inflx_rcp_newton2(v2));
  const double vtt_num = v00 * inflx_sq(v10) + v11 * inflx_sq(v00) - 2. * v00 * inflx_sq(v10);
  const double vtt_den = inflx_sq(v00) + inflx_sq(v10);
  const double vtt = inflx_quotient(vtt_num, vtt_den, inflx_rcp_newton2(vtt_den));
  const double vt2 = epsilon_v * inflx_reciprocal_quick(1. + inflx_sq(a_over_b));
  const double vtt_over_v = inflx_quotient(vtt, v, yv);
  const double abs_vtt_over_v = __builtin_copysign(fabs(vtt_over_v), v);
  const double eh_den = epsilon_v + abs_vtt_over_v - vt2;
  ok = ok && (inflx_magnitude_word(eh_den) - inflx_field_word(1023u - 500u) < inflx_field_word(1001u));
  const double epsilon_h = 3. * (epsilon_v - vt2) * inflx_reciprocal_quick(eh_den);
  const double t = fabs(b_over_a);
  const double delta = inflx_atan_nonneg<true, TABLE>(t, kc);
  const double omega2 = vtt_over_v * (3. - epsilon_h);  // anything: zero where vtt is, negative where omega is NaN
  ok = ok && inflx_sqrt_quick_ok(omega2);
  const double omega = inflx_sqrt_quick(omega2);
  const double eta_parallel = omega * inflx_tan_of_atan<TABLE>(t, delta, kc) - 3.;
  out[0] = consistency;
  out[1] = epsilon_v;
  out[2] = epsilon_h;
  out[3] = eta_parallel;
  out[4] = delta;
  out[5] = omega;
  return ok;
#endif
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

// ops::consistency_only with its five divisions spelled as inflx_quotient (see inflx_op_complete_analysis_quick): one
// reciprocal of V serves v11/V and v00/V, and no division carries its special-case handling -- 4 reciprocals + 5 x 3
// instructions instead of 5 x 11.  Returns whether the point qualified (else `out` is unspecified and the caller evaluates
// inflx_op_consistency_only).  Mid-range, division by division, for V, v00, v10, v11 normal with 2^-120 <= |x| < 2^121
// (a = numerator, b = denominator, q = quotient; needed: b and 1/b normal, |a| >= 2^-969 or a = +0, q normal or +0):
//   v11/V, v00/v10, v10/v00, v00/V     a, b in range, |q| in (2^-241, 2^241)
//   num / den, num = ||lhs| - |rhs||, den = |lhs| + |rhs|, lhs = v11/V - 3, rhs = 3 (v00/v10)^2 + (v00/V)(v10/v00)^2:
//     |lhs| is 0 or >= 2^-51 (a quotient within a factor 2 of 3 is a multiple of 2^-51) and < 2^242; |rhs| is 0 or
//     >= 2^-775 (a sum of multiples of 2^-775) and < 2^725; so den is 0 -- tested on its own -- or in [2^-775, 2^726): b and 1/b
//     normal.  With L >= S the two magnitudes: L >= 2 S gives num >= L/2 and q >= 1/4; else num = L - S is exact, 0 or >=
//     ulp(S)/2 >= 2^-53 S >= 2^-828, and q is +0 or >= 2^-53 S / (3 S) > 2^-55.
// (consistency_rapidturn_only has three divisions and no shared denominator: the range test would cost what the three
// fix-ups save, so it keeps the compiler's divisions.)
INFLX_FN bool inflx_op_consistency_only_quick(const InflxModelValues& m, double& out) {
#ifdef INFLX_HOST_TWIN
  out = inflx_op_consistency_only(m);
  return true;
#else
#pragma clang fp contract(off)
  const double v = m.V, v11 = m.v11, v10 = m.v10, v00 = m.v00;
  const unsigned e0 = inflx_exponent_field(v), e1 = inflx_exponent_field(v11), e2 = inflx_exponent_field(v10), e3 = inflx_exponent_field(v00);
  const unsigned lowest = min(min(e0, e1), min(e2, e3)), highest = max(max(e0, e1), max(e2, e3));
  bool ok = lowest >= 1023u - 120u && highest <= 1023u + 120u;
  const double yv = inflx_rcp_newton2(v);
  const double a_over_b = inflx_quotient(v00, v10, inflx_rcp_newton2(v10));
  const double b_over_a = inflx_quotient(v10, v00, inflx_rcp_newton2(v00));
  const double lhs = inflx_quotient(v11, v, yv) - 3.;
  const double rhs = 3. * inflx_sq(a_over_b) + inflx_quotient(v00, v, yv) * inflx_sq(b_over_a);
  const double num = fabs(fabs(lhs) - fabs(rhs)), den = fabs(lhs) + fabs(rhs);
  ok = ok && den != 0.0;
  out = inflx_quotient(num, den, inflx_rcp_newton2(den));
  return ok;
#endif
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

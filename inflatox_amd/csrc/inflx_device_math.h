// Device-side helpers that generated model headers may call.  gfx950 only.
//
// Everything here is IEEE-strict FP64: no fast-math, no flush-to-zero, correctly rounded
// division and sqrt.  NaN / +-Inf propagation is part of the result contract of the sweep
// (reference src/anguelova.rs:103-135 relies on IEEE semantics; its generated C is built
// without -ffast-math, python/inflatox/compiler.py:299-310).
#pragma once
// INFLX_FN qualifies every function of the model/ops headers.  tests/host_twin.cpp defines it as
// plain `inline` (and INFLX_HOST_TWIN) to compile the very same headers for the CPU, which is how
// the transpiler's staging is checked without a GPU.
#ifndef INFLX_HOST_TWIN
#include <hip/hip_runtime.h>
#define INFLX_FN __device__ __forceinline__
#endif
// the long special-function routines (continued fractions, double-double series) are real functions: a
// model may call them a dozen times, and every kernel of the code object would carry every copy
#ifndef INFLX_FN_NOINLINE
#ifdef INFLX_HOST_TWIN
#define INFLX_FN_NOINLINE static __attribute__((noinline))
#else
#define INFLX_FN_NOINLINE __device__ __noinline__
#endif
#endif

// x^N for a compile-time integer N >= 1 by binary exponentiation (at most 2*log2(N) multiplies,
// error <= (N-1) half-ulps; the reference calls libm pow(x, N) here, < 1 ulp).
template <int N>
INFLX_FN double inflx_ipow(double x) {
  static_assert(N >= 1, "inflx_ipow needs a positive exponent");
  if constexpr (N == 1) {
    return x;
  } else if constexpr (N % 2 == 0) {
    const double h = inflx_ipow<N / 2>(x);
    return h * h;
  } else {
    return x * inflx_ipow<N - 1>(x);
  }
}

// x^(N/2) for odd N >= 1: x^((N-1)/2) * sqrt(x).  sqrt of a negative base yields NaN exactly like
// pow(x, N/2.0) does.
template <int N>
INFLX_FN double inflx_hpow(double x) {
  static_assert(N >= 1 && (N % 2) == 1, "inflx_hpow needs an odd positive numerator");
  if constexpr (N == 1) {
    return sqrt(x);
  } else {
    return inflx_ipow<(N - 1) / 2>(x) * sqrt(x);
  }
}

// ---- division by a value that is known one stage earlier ---------------------------------------------
// A quotient a/b whose denominator depends on fewer grid axes than its numerator is the single most
// expensive thing left in the per-point stage: an IEEE double division is 11 instructions, one of them
// (v_rcp_f64) at a third of the full rate -- 13.0 times the issue cost of one v_fma_f64, measured
// (scripts/micro/valu_rates.hip).  With y = RN(1/b) computed once per row / column / sweep (inflx_recip,
// an IEEE division itself), the per-point work is one multiplication and two FMAs:
//     q0 = RN(a*y);   r = a - b*q0 (exact, FMA);   q = RN(q0 + r*y)
// -- Markstein's division step.  With a correctly rounded reciprocal, q is the correctly rounded quotient
// whenever q0 is within one ulp of a/b (Markstein 1990; Muller et al., Handbook of Floating-Point
// Arithmetic, section 4.7); q0 can be up to 1.5 ulp off, and then q still is RN(a/b + d) with |d| below
// 1.7e-16 ulp, which differs from RN(a/b) only if a/b lies that close to a rounding boundary: about 3 in
// 1e16 quotients, by one ulp (tests/div_hoisted_host.cpp: 228 million quotients, no difference).
//
// The three operations are only valid while nothing overflows or underflows on the way.  ONE comparison per
// quotient establishes that (4.8 fma-equivalents per quotient in all, against 13.0):
//   * inflx_recip hands over y = NaN unless 2^-500 <= |b| <= 2^500 (evaluated in the earlier stage);
//   * the quotient is accepted iff |q| >= 2^-400.  Then |a| >= 2^-901, so the residual a - b*q0 is a multiple
//     of 2^-1006 and exact; an infinite or NaN numerator, an overflowing product a*y and a NaN reciprocal all
//     end as q = NaN, which fails the comparison; zero, tiny and denormal quotients fail it by magnitude.
// A quotient that is not accepted clears `ok`: the tile kernel then evaluates that grid row again with IEEE
// divisions (inflx_stage_point_ieee), so the stored values are those of the IEEE program always.  Rows inside
// a NaN region of the model are evaluated twice for that reason -- correct, merely slower there.
INFLX_FN double inflx_recip(double b) {
  const double y = 1.0 / b;
  const double m = __builtin_fabs(b);
  return (m >= 0x1p-500 && m <= 0x1p500) ? y : __builtin_nan("");
}

// For a quotient whose numerator is a product of values of earlier stages the comparison is not needed at all:
// the stages that produce the factors test them once against [2^-E, 2^E] (inflx_out_of_range, summed into one
// flag per stage; E = 160 for at most three factors, 500 for the denominator), which bounds numerator and
// quotient away from every overflow and underflow, and the point stage tests the sum of the flags once.
template <int E>
INFLX_FN double inflx_out_of_range(double x) {
  const double m = __builtin_fabs(x);
  return (m >= __builtin_ldexp(1.0, -E) && m <= __builtin_ldexp(1.0, E)) ? 0.0 : 1.0;  // NaN: out of range
}

INFLX_FN double inflx_div_by_hoisted_in_range(double a, double b, double y) {
  const double q0 = a * y;
  const double r = __builtin_fma(-b, q0, a);
  return __builtin_fma(r, y, q0);
}

INFLX_FN double inflx_div_by_hoisted(double a, double b, double y, bool& ok) {
  const double q0 = a * y;
  const double r = __builtin_fma(-b, q0, a);
  const double q = __builtin_fma(r, y, q0);
#ifndef INFLX_DIVH_TRUST  // (experiments only: time the hot loop as if every quotient were accepted)
  ok = ok && (__builtin_fabs(q) >= 0x1p-400);
#endif
  return q;
}

// The same three operations for models whose point stage is NOT kept twice (Compiler(hoist_reciprocals="inline"): EGNO -- a dozen
// quotients by row / parameter values, too few to pay for a second copy of the stage and the `ok` bookkeeping of the row redo, which
// cost it 20 registers and the third wavefront per SIMD).  The quotient checks itself with the one comparison above; a wavefront in
// which any lane fails it -- NaN / infinite / zero / tiny operands, a reciprocal that inflx_recip refused -- divides THOSE lanes
// with the compiler's IEEE sequence on the spot.  The branch is wave-uniform (a ballot), so the eleven instructions of the division
// are off the hot path and never if-converted into it; the value is RN(a/b) either way.
#ifndef INFLX_HOST_TWIN
INFLX_FN double inflx_div_by_hoisted_inline(double a, double b, double y) {
  const double q0 = a * y;
  const double r = __builtin_fma(-b, q0, a);
  double q = __builtin_fma(r, y, q0);
  const bool regular = __builtin_fabs(q) >= 0x1p-400;
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!regular) != 0, 0)) q = regular ? q : a / b;
  return q;
}
#else
INFLX_FN double inflx_div_by_hoisted_inline(double a, double b, double y) {
  const double q0 = a * y;
  const double r = __builtin_fma(-b, q0, a);
  const double q = __builtin_fma(r, y, q0);
  return __builtin_fabs(q) >= 0x1p-400 ? q : a / b;
}
#endif

// ---- quotients of the point stage that share a PER-POINT denominator ---------------------------------------
// (D5 divides four times by r_44*p_5 at every grid point.)  The compiler's IEEE division is: two v_div_scale, v_rcp_f64
// and two Newton steps on the reciprocal (5 instructions), then q0 = a*y, r = fma(-b, q0, a), q = fma(r, y, q0) (v_div_fmas
// with nothing scaled), v_div_fixup.  For operands in mid range the scaling and the fix-up change nothing, so the
// quotient is exactly what the eight arithmetic instructions deliver (inflx_ops.h has the argument in full) -- and the
// five of the reciprocal depend on the denominator alone: quotients with the same denominator share them and cost
// three instructions and one comparison each instead of eleven.  "Mid range" is tested, not assumed: the denominator's
// exponent field once per reciprocal (2^-500 <= |b| < 2^501), |q| >= 2^-400 per quotient (a NaN, an infinity, an
// overflow or a zero anywhere fail one of the two); a point that fails clears `ok` and the tile kernel evaluates its
// grid row again with the compiler's divisions, so the stored values are the IEEE program's always.
#ifndef INFLX_HOST_TWIN
INFLX_FN double inflx_shared_reciprocal(double b, bool& ok) {
  const double y0 = __builtin_amdgcn_rcp(b);
  const double y1 = __builtin_fma(y0, __builtin_fma(-b, y0, 1.0), y0);
  ok = ok && ((((unsigned)__double2hiint(b) >> 20) & 0x7ffu) - (1023u - 500u) <= 1000u);
  return __builtin_fma(y1, __builtin_fma(-b, y1, 1.0), y1);
}
INFLX_FN double inflx_div_by_shared(double a, double b, double y, bool& ok) {
  const double q0 = a * y;
  const double q = __builtin_fma(__builtin_fma(-b, q0, a), y, q0);
  ok = ok && (__builtin_fabs(q) >= 0x1p-400);
  return q;
}
#else
INFLX_FN double inflx_shared_reciprocal(double, bool&) { return 0.0; }
INFLX_FN double inflx_div_by_shared(double a, double b, double, bool&) { return a / b; }
#endif

// ---- square roots of the point stage without their special-case handling ------------------------------------
#ifndef INFLX_HOST_TWIN
// High word of |x|: exponent field and the top 20 bits of the significand as one unsigned number, monotonic in |x|
// (field >= E  <=>  word >= E << 20): a range test of one value is an AND, a subtraction and an unsigned comparison.
// (Spelling the five-value test of the quick epilogue this way as well -- v_min3 / v_max3 over the words instead of the
// extracted fields, 11 instructions for the compiler's 25 -- changes nothing measurable for doc, angular and EGNO and
// costs D5 twelve more spilled registers, profiles/r03_experiments.txt section 10: it keeps the fields.)
INFLX_FN unsigned inflx_magnitude_word(double x) { return (unsigned)__double2hiint(x) & 0x7fffffffu; }
constexpr unsigned inflx_field_word(unsigned field) { return field << 20; }

// sqrt(x) as the compiler spells it (v_rsq_f64, then Goldschmidt's coupled iteration and two residual corrections),
// WITHOUT the operand scaling (x < 2^-767 is multiplied by 2^256 first, the root by 2^-128 afterwards) and without the
// final selection that returns x itself for zeros and +infinity: eight instructions fewer.  Bit for bit the compiler's
// result for every x with 2^-767 <= |x| < infinity -- for negative x both spellings return the NaN of v_rsq_f64 -- which
// the caller establishes (inflx_sqrt_quick_ok).
INFLX_FN double inflx_sqrt_quick(double x) {
#pragma clang fp contract(off)
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d0 = __builtin_fma(-g, g, x);
  g = __builtin_fma(d0, h, g);
  const double d1 = __builtin_fma(-g, g, x);
  return __builtin_fma(d1, h, g);
}
// exponent field of x in [1023 - 767, 2046]: one subtraction and one unsigned comparison of the magnitude word
INFLX_FN bool inflx_sqrt_quick_ok(double x) { return inflx_magnitude_word(x) - inflx_field_word(1023u - 767u) < inflx_field_word(2047u - (1023u - 767u)); }
// the point stage's sqrt(x) and x^(N/2) in the quick variant of the stage: a value outside the guard clears `ok` and the tile
// kernel evaluates the grid row again with the compiler's sqrt (inflx_stage_point_ieee)
INFLX_FN double inflx_sqrt_checked(double x, bool& ok) {
  ok = ok && inflx_sqrt_quick_ok(x);
  return inflx_sqrt_quick(x);
}
#else
INFLX_FN double inflx_sqrt_checked(double x, bool&) { return sqrt(x); }
#endif
template <int N>
INFLX_FN double inflx_hpow_checked(double x, bool& ok) {
  static_assert(N >= 1 && (N % 2) == 1, "inflx_hpow_checked needs an odd positive numerator");
  if constexpr (N == 1) {
    return inflx_sqrt_checked(x, ok);
  } else {
    return inflx_ipow<(N - 1) / 2>(x) * inflx_sqrt_checked(x, ok);
  }
}

// reciprocal hyperbolic / trigonometric functions sympy may emit without a C99 spelling
INFLX_FN double inflx_coth(double x) { return 1.0 / tanh(x); }
INFLX_FN double inflx_sech(double x) { return 1.0 / cosh(x); }
INFLX_FN double inflx_csch(double x) { return 1.0 / sinh(x); }
INFLX_FN double inflx_cot(double x) { return 1.0 / tan(x); }
INFLX_FN double inflx_sec(double x) { return 1.0 / cos(x); }
INFLX_FN double inflx_csc(double x) { return 1.0 / sin(x); }

// Bessel functions (the reference's GSL path)
#include "inflx_sf.h"

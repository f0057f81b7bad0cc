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

// reciprocal hyperbolic / trigonometric functions sympy may emit without a C99 spelling
INFLX_FN double inflx_coth(double x) { return 1.0 / tanh(x); }
INFLX_FN double inflx_sech(double x) { return 1.0 / cosh(x); }
INFLX_FN double inflx_csch(double x) { return 1.0 / sinh(x); }
INFLX_FN double inflx_cot(double x) { return 1.0 / tan(x); }
INFLX_FN double inflx_sec(double x) { return 1.0 / cos(x); }
INFLX_FN double inflx_csc(double x) { return 1.0 / sin(x); }

// Bessel functions (the reference's GSL path)
#include "inflx_sf.h"

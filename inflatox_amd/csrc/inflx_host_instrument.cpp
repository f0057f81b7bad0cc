// Host evaluation of a generated model header -- the transpiler's CONDITIONING INSTRUMENT.
//
// Not a sweep path and never a fallback for one: Compiler(regroup="auto", sample=...) uses it at transpile time to decide
// which of the five model values may be re-associated (inflatox_amd/_instrument.py).  It compiles the very stage functions
// the kernels inline (inflx_stage_uniform / _row / _col / _point of the generated header) for the CPU and evaluates the
// five model values at a list of points, in double or -- built with -DINFLX_INSTRUMENT_LONG_DOUBLE -- with every
// `double` of the generated code and of the helper headers read as `long double` (x87 extended precision on x86-64:
// 64-bit significands, 2048 times finer than float64; on hosts whose long double is float64 the instrument degrades to
// "no information" and the transpiler then keeps the reference's arithmetic).  The difference of the two builds at a
// point is the rounding error of the float64 evaluation there: what a re-association is allowed to change.
#include <math.h>  // the C++ <math.h>: every overload of sqrt / pow / sin ... in the global namespace
#include <stddef.h>
#include <stdint.h>

typedef double inflx_f64;  // the interface stays float64 whatever the arithmetic inside

#define INFLX_HOST_TWIN 1
#define INFLX_FN static inline
#ifdef INFLX_INSTRUMENT_LONG_DOUBLE
#define double long double
#endif

#include "inflx_device_math.h"
#include "inflx_ops.h"  // InflxModelValues
#include INFLX_MODEL_HEADER

static constexpr int kNU = INFLX_NU > 0 ? INFLX_NU : 1;
static constexpr int kNR = INFLX_NR > 0 ? INFLX_NR : 1;
static constexpr int kNC = INFLX_NC > 0 ? INFLX_NC : 1;
static constexpr int kNP = INFLX_N_PARAMETERS > 0 ? INFLX_N_PARAMETERS : 1;

extern "C" {

unsigned inflx_instrument_n_parameters() { return INFLX_N_PARAMETERS; }
#ifdef INFLX_INSTRUMENT_LONG_DOUBLE
unsigned inflx_instrument_mantissa_bits() { return (unsigned)__LDBL_MANT_DIG__; }  // 64 on x86-64
#else
unsigned inflx_instrument_mantissa_bits() { return 53u; }
#endif

// out: n records (V, v00, v10, v11, |dV|^2) at the n points (x0, x1) of `pts`
void inflx_instrument_raw(const inflx_f64* p, const inflx_f64* pts, size_t n, inflx_f64* out) {
  double A[kNP], U[kNU], R[kNR], C[kNC];
  for (int k = 0; k < INFLX_N_PARAMETERS; ++k) A[k] = p[k];
  inflx_stage_uniform(A, U);
  for (size_t i = 0; i < n; ++i) {
    const double x0 = pts[2 * i], x1 = pts[2 * i + 1];
    inflx_stage_row(x0, A, U, R);
    inflx_stage_col(x1, A, U, C);
    InflxModelValues mv;
    // the point stage with plain divisions (the quick variant's Markstein steps go through __builtin_fma, which is a
    // float64 operation whatever `double` reads as: the extended-precision build must not round there)
#if INFLX_HAS_QUICK_POINT
    inflx_stage_point_ieee(x0, x1, A, U, R, C, mv);
#else
    inflx_stage_point(x0, x1, A, U, R, C, mv);
#endif
    out[5 * i + 0] = (inflx_f64)mv.V;
    out[5 * i + 1] = (inflx_f64)mv.v00;
    out[5 * i + 2] = (inflx_f64)mv.v10;
    out[5 * i + 3] = (inflx_f64)mv.v11;
    out[5 * i + 4] = (inflx_f64)mv.g;
  }
}

}  // extern "C"

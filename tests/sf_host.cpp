// Host build of csrc/inflx_sf.h for the CPU test-suite -- TEST INFRASTRUCTURE (the product only uses the
// header inside gfx950 code objects).
#include <cmath>
#define INFLX_FN static inline
using std::cos;
using std::exp;
using std::fabs;
using std::fmax;
using std::floor;
using std::lgamma;
using std::tgamma;
using std::tan;
using std::log1p;
using std::sinh;
using std::tanh;
using std::cosh;
using std::log;
using std::sin;
using std::sqrt;
using std::atan;
using std::pow;
using std::fmin;
#include "inflx_sf.h"

extern "C" {
#define F1(name) \
  void sf_##name(const double* x, int n, double* out) { for (int i = 0; i < n; ++i) out[i] = inflx_sf_bessel_##name(x[i]); }
#define F2(name) \
  void sf_##name(int order, const double* x, int n, double* out) { for (int i = 0; i < n; ++i) out[i] = inflx_sf_bessel_##name(order, x[i]); }
F1(J0) F1(J1) F1(Y0) F1(Y1) F1(I0) F1(I1) F1(K0) F1(K1) F1(j0) F1(j1) F1(j2) F1(y0) F1(y1) F1(y2)
F2(Jn) F2(Yn) F2(In) F2(Kn) F2(jl) F2(yl)
#define F2R(name) \
  void sf_##name(double order, const double* x, int n, double* out) { for (int i = 0; i < n; ++i) out[i] = inflx_sf_bessel_##name(order, x[i]); }
F2R(Jnu) F2R(Ynu) F2R(Inu) F2R(Knu)
void sf_1F1(double a, double b, const double* x, int n, double* out) { for (int i = 0; i < n; ++i) out[i] = inflx_sf_hyperg_1F1(a, b, x[i]); }
void sf_2F1(double a, double b, double c, const double* x, int n, double* out) { for (int i = 0; i < n; ++i) out[i] = inflx_sf_hyperg_2F1(a, b, c, x[i]); }
void sf_2F0(double a, double b, const double* x, int n, double* out) { for (int i = 0; i < n; ++i) out[i] = inflx_sf_hyperg_2F0(a, b, x[i]); }
// the status word the functions above leave their domain / declined notes in (csrc/inflx_sf.h); reading clears it
unsigned sf_status_take() {
  const unsigned v = inflx_sf_status_host;
  inflx_sf_status_host = 0u;
  return v;
}
void sf_0F1(double c, const double* x, int n, double* out) { for (int i = 0; i < n; ++i) out[i] = inflx_sf_hyperg_0F1(c, x[i]); }
}

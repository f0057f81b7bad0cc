// Host twin of the sweep kernels -- TEST INFRASTRUCTURE.
//
// Compiles the very headers the GPU kernels are built from (generated model header,
// csrc/inflx_ops.h, csrc/inflx_device_math.h) for the CPU and walks a grid stage by stage exactly
// like csrc/inflx_sweep_kernels.hip does: U once, C once per column, R once per row, then the
// point stage and the per-point operation.  It lets the CPU test-suite check the transpiler's
// common-subexpression elimination and axis staging against the oracle without a GPU.  It is never
// used by the product.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

#define INFLX_HOST_TWIN 1
#define INFLX_FN static inline
using std::atan;
using std::cos;
using std::cosh;
using std::exp;
using std::fabs;
using std::floor;
using std::fmax;
using std::lgamma;
using std::log1p;
using std::tgamma;
using std::log;
using std::pow;
using std::sin;
using std::sinh;
using std::sqrt;
using std::tan;
using std::tanh;

#include "inflx_device_math.h"
#include "inflx_kernel_abi.h"
#include "inflx_ops.h"
#include INFLX_MODEL_HEADER

static constexpr int kNU = INFLX_NU > 0 ? INFLX_NU : 1;
static constexpr int kNR = INFLX_NR > 0 ? INFLX_NR : 1;
static constexpr int kNC = INFLX_NC > 0 ? INFLX_NC : 1;

static int width(int op) { return op == INFLX_OP_COMPLETE ? 6 : (op == INFLX_OP_RAW ? 5 : (op == INFLX_OP_HESSE ? 4 : 1)); }

static double g_accuracy = 0.0;
static void apply(int op, const InflxModelValues& mv, double* o) {
  switch (op) {
    case INFLX_OP_QDIF: o[0] = inflx_op_flag_quantum_diff(mv, g_accuracy) ? 1.0 : 0.0; break;
    case INFLX_OP_COMPLETE: inflx_op_complete_analysis(mv, o); break;
    case INFLX_OP_CONSISTENCY: o[0] = inflx_op_consistency_only(mv); break;
    case INFLX_OP_RAPIDTURN: o[0] = inflx_op_consistency_rapidturn_only(mv); break;
    case INFLX_OP_EPSILON_V: o[0] = inflx_op_epsilon_v_only(mv); break;
    case INFLX_OP_HESSE: o[0] = mv.v00; o[1] = mv.v01; o[2] = mv.v10; o[3] = mv.v11; break;
    default:
      o[0] = mv.V; o[1] = mv.v00; o[2] = mv.v10; o[3] = mv.v11; o[4] = mv.g;
  }
}

// INFLX_OP_HESSE: the reference's own v01 (csrc/inflx_sweep_kernels.hip fill_v01)
static void fill_v01(InflxModelValues& mv, double x0, double x1, const double* p) {
#if INFLX_V01_IS_V10
  (void)x0, (void)x1, (void)p;
  mv.v01 = mv.v10;
#else
  mv.v01 = inflx_v01_point(x0, x1, p);
#endif
}

extern "C" {

unsigned twin_v01_is_v10() { return INFLX_V01_IS_V10; }
unsigned twin_n_parameters() { return INFLX_N_PARAMETERS; }
void twin_set_accuracy(double a) { g_accuracy = a; }
unsigned twin_out_mask() { return INFLX_OUT_MASK; }

// out: (N0, N1, K) AoS
void twin_grid(int op, const double* p, const double* ss, size_t N0, size_t N1, double* out) {
  const int K = width(op);
  const double x0a = ss[0], dx0 = (ss[1] - ss[0]) / (double)N0, x1a = ss[2], dx1 = (ss[3] - ss[2]) / (double)N1;
  double U[kNU];
  inflx_stage_uniform(p, U);
  std::vector<double> C(N1 * kNC), R(kNR);
  for (size_t j = 0; j < N1; ++j) inflx_stage_col(inflx_coord(j, dx1, x1a), p, U, &C[j * kNC]);
  for (size_t i = 0; i < N0; ++i) {
    const double x0 = inflx_coord(i, dx0, x0a);
    inflx_stage_row(x0, p, U, R.data());
    for (size_t j = 0; j < N1; ++j) {
      InflxModelValues mv;
      inflx_stage_point(x0, inflx_coord(j, dx1, x1a), p, U, R.data(), &C[j * kNC], mv);
      fill_v01(mv, x0, inflx_coord(j, dx1, x1a), p);
      apply(op, mv, out + (i * N1 + j) * K);
    }
  }
}

// out: (n, 7) -- the records of the device kernel inflx_basis_points
void twin_basis(const double* p, const double* pts, size_t n, double* out) {
  for (size_t k = 0; k < n; ++k) inflx_basis_point(pts[2 * k], pts[2 * k + 1], p, out + 7 * k);
}

// out: (n, K)
void twin_trajectory(int op, const double* p, const double* pts, size_t n, double* out) {
  const int K = width(op);
  double U[kNU], R[kNR], C[kNC];
  inflx_stage_uniform(p, U);
  for (size_t k = 0; k < n; ++k) {
    const double x0 = pts[2 * k], x1 = pts[2 * k + 1];
    inflx_stage_row(x0, p, U, R);
    inflx_stage_col(x1, p, U, C);
    InflxModelValues mv;
    inflx_stage_point(x0, x1, p, U, R, C, mv);
    fill_v01(mv, x0, x1, p);
    apply(op, mv, out + k * K);
  }
}
}

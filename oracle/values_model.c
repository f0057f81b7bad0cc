/*
 * values_model.c -- a "model" whose functions return numbers from a table instead of evaluating expressions.
 *
 * TEST INFRASTRUCTURE ONLY (like everything under oracle/).  It exists so that the oracle's own per-point operations
 * (sweep_oracle.c: op_complete_analysis etc., the restatement of src/anguelova.rs:99-171) can be run on ARBITRARY
 * tuples (V, v00, v10, v11, grad_norm_squared) -- specials, zeros, denormals, random bit patterns -- that no real model
 * produces on demand: the artefact exports the reference's model ABI (src/dylib.rs:32-48), the oracle opens it like
 * any other model and sweeps an explicit point list whose x[0] is the record number.
 */
#include <stddef.h>
#include <stdint.h>

const uint16_t VERSION[3] = {5, 0, 0};
const uint32_t DIM = 2;
const uint32_t N_PARAMETERS = 1;
char *const MODEL_NAME = "values_table";
const char USE_GSL = 0;

/* set by oracle.ops_on_values through the symbol's address: n records of 5 doubles */
const double *inflx_values_table = 0;

#define RECORD(x) (inflx_values_table + 5 * (size_t)(x)[0])
double V(const double x[], const double args[]) { (void)args; return RECORD(x)[0]; }
double v00(const double x[], const double args[]) { (void)args; return RECORD(x)[1]; }
double v10(const double x[], const double args[]) { (void)args; return RECORD(x)[2]; }
double v01(const double x[], const double args[]) { (void)args; return RECORD(x)[2]; }
double v11(const double x[], const double args[]) { (void)args; return RECORD(x)[3]; }
double grad_norm_squared(const double x[], const double args[]) { (void)args; return RECORD(x)[4]; }
void v(const double x[], const double args[], double v_out[]) { (void)x; (void)args; v_out[0] = v_out[1] = 0.0; }
void w1(const double x[], const double args[], double v_out[]) { (void)x; (void)args; v_out[0] = v_out[1] = 0.0; }

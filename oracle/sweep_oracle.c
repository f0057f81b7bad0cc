/*
 * sweep_oracle.c -- CPU restatement of the reference's grid-sweep hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (inflatox_amd/) may import, link or
 * call this file; it is the checker used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.
 *
 * What it restates (all citations relative to /root/reference):
 *   - model-artefact loading and ABI check        src/dylib.rs:67-161, src/inflatox_version.rs:48-53
 *   - Hesse2D / Potential symbol binding          src/hesse_bindings.rs:38-47,202-210
 *   - start/stop -> spacing/offset conversion     src/anguelova.rs:84-94, src/lib.rs:117-139
 *   - flat index -> grid index -> field point     src/anguelova.rs:514-516,531-533
 *   - per-point operation ops::complete_analysis  src/anguelova.rs:103-135
 *   - the other per-point operations              src/anguelova.rs:138-170
 *   - serial / chunked-parallel sweep drivers     src/anguelova.rs:508-540
 *
 * The model functions themselves (V, v00, v10, v11, grad_norm_squared) live in a per-model
 * shared object with the reference's C ABI (src/dylib.rs:32-48) and are called through
 * function pointers, five indirect calls per grid point, exactly like the reference
 * (no inlining across the boundary).
 *
 * Parity status: the Rust crate cannot be built here (no rustc/cargo), so this file is
 * pinned (a) on its *inputs* by the reference's own known-answer test (tests/test_doc.py:50-51),
 * (b) on the one inequality that test states for the output (tests/test_doc.py:58) and (c) by
 * golden vectors generated from the reference's own Python stages (tests/golden/make_golden.py).
 * The per-point formulas are a line-by-line restatement with the reference's evaluation order;
 * compile with -ffp-contract=off so that, like rustc, no FMA contraction happens.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef double (*exfn2)(const double *, const double *);
typedef void (*exvecfn)(const double *, const double *, double *);

/* ABI version the reference's native module accepts: src/lib.rs:50 (major.minor compared only) */
static const uint16_t ORACLE_ABI[3] = {5, 0, 0};

typedef struct {
  void *handle;
  uint32_t dim;
  uint32_t n_par;
  char name[256];
  exfn2 V, grad_sq;      /* src/dylib.rs:126-140 */
  exfn2 v00, v01, v10, v11; /* src/dylib.rs:163-183 */
  exvecfn basis_v;       /* src/dylib.rs:185 (basis fn 0 = "v") */
} oracle_model;

static __thread char oracle_err[512];
const char *oracle_last_error(void) { return oracle_err; }

#define FAIL(code, ...)                                  \
  do {                                                   \
    snprintf(oracle_err, sizeof oracle_err, __VA_ARGS__); \
    return (code);                                       \
  } while (0)

/* error codes mirror the classes of src/err.rs:29-38 */
enum { ORACLE_OK = 0, ORACLE_EIO = 1, ORACLE_ESYMBOL = 2, ORACLE_EVERSION = 3, ORACLE_ESHAPE = 4 };

int oracle_open(const char *path, oracle_model **out) {
  void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) FAIL(ORACLE_EIO, "could not open %s: %s", path, dlerror());
  const uint16_t *ver = (const uint16_t *)dlsym(h, "VERSION");
  if (!ver) { dlclose(h); FAIL(ORACLE_ESYMBOL, "missing symbol VERSION in %s", path); }
  if (ver[0] != ORACLE_ABI[0] || ver[1] != ORACLE_ABI[1]) {
    int a = ver[0], b = ver[1], c = ver[2];
    dlclose(h);
    FAIL(ORACLE_EVERSION, "artefact ABI v%d.%d.%d incompatible with v5.0", a, b, c);
  }
  const uint32_t *dim = (const uint32_t *)dlsym(h, "DIM");
  const uint32_t *npar = (const uint32_t *)dlsym(h, "N_PARAMETERS");
  char *const *name = (char *const *)dlsym(h, "MODEL_NAME");
  if (!dim || !npar || !name) { dlclose(h); FAIL(ORACLE_ESYMBOL, "missing DIM/N_PARAMETERS/MODEL_NAME in %s", path); }
  oracle_model *m = (oracle_model *)calloc(1, sizeof *m);
  m->handle = h;
  m->dim = *dim;
  m->n_par = *npar;
  snprintf(m->name, sizeof m->name, "%s", *name);
  m->V = (exfn2)dlsym(h, "V");
  m->grad_sq = (exfn2)dlsym(h, "grad_norm_squared");
  if (!m->V || !m->grad_sq) { dlclose(h); free(m); FAIL(ORACLE_ESYMBOL, "missing V/grad_norm_squared in %s", path); }
  *out = m;
  return ORACLE_OK;
}

void oracle_close(oracle_model *m) {
  if (!m) return;
  dlclose(m->handle);
  free(m);
}

uint32_t oracle_dim(const oracle_model *m) { return m->dim; }
uint32_t oracle_n_par(const oracle_model *m) { return m->n_par; }
const char *oracle_name(const oracle_model *m) { return m->name; }

/* Hesse2D::new + Potential::new: resolve v00,v01,v10,v11 and basis fn "v"; n_fields must be 2 */
static int bind_2d(oracle_model *m) {
  if (m->dim != 2) FAIL(ORACLE_ESHAPE, "model has %u fields; the sweep requires a 2-field model", m->dim);
  if (m->v00) return ORACLE_OK;
  m->v00 = (exfn2)dlsym(m->handle, "v00");
  m->v01 = (exfn2)dlsym(m->handle, "v01");
  m->v10 = (exfn2)dlsym(m->handle, "v10");
  m->v11 = (exfn2)dlsym(m->handle, "v11");
  m->basis_v = (exvecfn)dlsym(m->handle, "v");
  if (!m->v00 || !m->v01 || !m->v10 || !m->v11 || !m->basis_v) {
    m->v00 = NULL;
    FAIL(ORACLE_ESYMBOL, "missing Hesse component or basis symbol");
  }
  return ORACLE_OK;
}

/* ---- per-point operations (src/anguelova.rs: mod ops) ------------------------------------ */

static inline double sq(double x) { return x * x; } /* f64::powi(2) */

/* ops::complete_analysis, src/anguelova.rs:103-135 */
static inline void op_complete_analysis(const oracle_model *m, const double x[2], const double *p, double val[6]) {
  const double v = m->V(x, p), v11 = m->v11(x, p), v10 = m->v10(x, p), v00 = m->v00(x, p);
  double consistency;
  {
    const double lhs = v11 / v;
    const double rhs = 3. + 3. * sq(v00 / v10) + (v00 / v) * sq(v10 / v00);
    consistency = fabs(lhs - rhs) / (fabs(lhs) + fabs(rhs));
  }
  const double epsilon_v = m->grad_sq(x, p) / sq(v);
  const double vtt = (v00 * sq(v10) + v11 * sq(v00) - 2. * v00 * sq(v10)) / (sq(v00) + sq(v10));
  const double vt2 = epsilon_v * (1. / (1. + sq(v00 / v10)));
  const double epsilon_h = 3. * (epsilon_v - vt2) * (1. / (epsilon_v + fabs(vtt) / v - vt2));
  const double delta = atan(fabs(v10 / v00));
  const double omega = sqrt((vtt / v) * (3. - epsilon_h));
  const double eta_parallel = omega * tan(delta) - 3.;
  val[0] = consistency;
  val[1] = epsilon_v;
  val[2] = epsilon_h;
  val[3] = eta_parallel;
  val[4] = delta;
  val[5] = omega;
}

/* ops::epsilon_v_only, src/anguelova.rs:138-140 */
static inline double op_epsilon_v_only(const oracle_model *m, const double x[2], const double *p) {
  return 0.5 * m->grad_sq(x, p) / sq(m->V(x, p));
}

/* ops::consistency_rapidturn_only, src/anguelova.rs:143-154 */
static inline double op_consistency_rapidturn_only(const oracle_model *m, const double x[2], const double *p) {
  const double v = m->V(x, p), v11 = m->v11(x, p), v10 = m->v10(x, p), v00 = m->v00(x, p);
  const double lhs = v11 / v;
  const double rhs = 3. * sq(v10 / v00);
  return fabs(fabs(lhs) - fabs(rhs)) / (fabs(lhs) + fabs(rhs));
}

/* ops::consistency_only, src/anguelova.rs:157-163 */
static inline double op_consistency_only(const oracle_model *m, const double x[2], const double *p) {
  const double v = m->V(x, p), v11 = m->v11(x, p), v10 = m->v10(x, p), v00 = m->v00(x, p);
  const double lhs = v11 / v - 3.;
  const double rhs = 3. * sq(v00 / v10) + (v00 / v) * sq(v10 / v00);
  return fabs(fabs(lhs) - fabs(rhs)) / (fabs(lhs) + fabs(rhs));
}

/* ops::flag_quantum_diff, src/anguelova.rs:166-170 (note: no abs()) */
static inline uint8_t op_flag_quantum_diff(const oracle_model *m, const double x[2], const double *p, double accuracy) {
  double g[2] = {0., 0.};
  m->basis_v(x, p, g);
  return (g[0] <= accuracy) && (g[1] <= accuracy);
}

/* raw model values at a point, in the order V, v00, v10, v11, grad_norm_squared */
static inline void op_raw(const oracle_model *m, const double x[2], const double *p, double val[5]) {
  val[0] = m->V(x, p);
  val[1] = m->v00(x, p);
  val[2] = m->v10(x, p);
  val[3] = m->v11(x, p);
  val[4] = m->grad_sq(x, p);
}

/* ---- sweep drivers ----------------------------------------------------------------------- */

enum {
  OP_COMPLETE = 0,    /* 6 f64 per point */
  OP_CONSISTENCY = 1, /* 1 f64 */
  OP_RAPIDTURN = 2,   /* 1 f64 */
  OP_EPSILON_V = 3,   /* 1 f64 */
  OP_QDIF = 4,        /* 1 u8 */
  OP_RAW = 5          /* 5 f64: V,v00,v10,v11,g (oracle-only helper, pins the model functions) */
};

typedef struct {
  const oracle_model *m;
  const double *p;
  void *out;
  const double *traj; /* NULL for grid sweeps, else (n,2) points */
  size_t n1;          /* shape[1] */
  double dx0, dx1, x0a, x1a;
  double accuracy;
  int op;
  size_t begin, end;
} job_t;

static void run_range(const job_t *j) {
  for (size_t idx = j->begin; idx < j->end; ++idx) {
    double x[2];
    if (j->traj) {
      x[0] = j->traj[2 * idx];
      x[1] = j->traj[2 * idx + 1];
    } else {
      /* src/anguelova.rs:514-516: ((idx / N1) as f64) * spacing + offset; mul then add */
      const double fi = (double)(idx / j->n1), fj = (double)(idx % j->n1);
      x[0] = fi * j->dx0 + j->x0a;
      x[1] = fj * j->dx1 + j->x1a;
    }
    switch (j->op) {
      case OP_COMPLETE: op_complete_analysis(j->m, x, j->p, (double *)j->out + 6 * idx); break;
      case OP_CONSISTENCY: ((double *)j->out)[idx] = op_consistency_only(j->m, x, j->p); break;
      case OP_RAPIDTURN: ((double *)j->out)[idx] = op_consistency_rapidturn_only(j->m, x, j->p); break;
      case OP_EPSILON_V: ((double *)j->out)[idx] = op_epsilon_v_only(j->m, x, j->p); break;
      case OP_QDIF: ((uint8_t *)j->out)[idx] = op_flag_quantum_diff(j->m, x, j->p, j->accuracy); break;
      case OP_RAW: op_raw(j->m, x, j->p, (double *)j->out + 5 * idx); break;
    }
  }
}

static void *thread_main(void *arg) {
  run_range((const job_t *)arg);
  return NULL;
}

static int run_sweep(job_t base, size_t len, int threads) {
  if (threads <= 1 || len < 2) {
    base.begin = 0;
    base.end = len;
    run_range(&base);
    return ORACLE_OK;
  }
  /* static contiguous chunks over points (rayon splits the same index space adaptively;
   * every point is independent, so the partition does not affect results) */
  if ((size_t)threads > len) threads = (int)len;
  pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * threads);
  job_t *jobs = (job_t *)malloc(sizeof(job_t) * threads);
  for (int t = 0; t < threads; ++t) {
    jobs[t] = base;
    jobs[t].begin = len * (size_t)t / (size_t)threads;
    jobs[t].end = len * (size_t)(t + 1) / (size_t)threads;
    pthread_create(&tid[t], NULL, thread_main, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
  free(tid);
  free(jobs);
  return ORACLE_OK;
}

/*
 * Grid sweep.  start_stop is the row-major (2,2) array [[x0a,x0b],[x1a,x1b]]
 * (python/inflatox/consistency_conditions.py:292-294, src/lib.rs:117-139).
 * out: OP_COMPLETE (N0,N1,6) f64; OP_RAW (N0,N1,5) f64; OP_QDIF (N0,N1) u8; else (N0,N1) f64.
 * threads: 1 = serial (src/anguelova.rs:508-521), else chunked over `threads` workers (:524-540).
 */
int oracle_grid_sweep(oracle_model *m, int op, const double *p, size_t n_p, void *out, const double start_stop[4],
                      size_t N0, size_t N1, double accuracy, int threads) {
  int rc = bind_2d(m);
  if (rc) return rc;
  if (n_p != m->n_par) /* validiate_p, src/anguelova.rs:70-79 */
    FAIL(ORACLE_ESHAPE, "model \"%s\" has %u paramters (got %zu)", m->name, m->n_par, n_p);
  job_t j;
  memset(&j, 0, sizeof j);
  j.m = m;
  j.p = p;
  j.out = out;
  j.n1 = N1;
  /* convert_ranges, src/anguelova.rs:84-94: spacing = (stop - start) / N, endpoint excluded */
  j.dx0 = (start_stop[1] - start_stop[0]) / (double)N0;
  j.dx1 = (start_stop[3] - start_stop[2]) / (double)N1;
  j.x0a = start_stop[0];
  j.x1a = start_stop[2];
  j.accuracy = accuracy;
  j.op = op;
  return run_sweep(j, N0 * N1, threads);
}

/* On-trajectory variants (src/anguelova.rs:633-977): same ops on an explicit (n,2) point list. */
int oracle_trajectory_sweep(oracle_model *m, int op, const double *p, size_t n_p, const double *traj, size_t n,
                            void *out, double accuracy, int threads) {
  int rc = bind_2d(m);
  if (rc) return rc;
  if (n_p != m->n_par) FAIL(ORACLE_ESHAPE, "model \"%s\" has %u paramters (got %zu)", m->name, m->n_par, n_p);
  job_t j;
  memset(&j, 0, sizeof j);
  j.m = m;
  j.p = p;
  j.out = out;
  j.traj = traj;
  j.n1 = 1;
  j.accuracy = accuracy;
  j.op = op;
  return run_sweep(j, n, threads);
}

/* scalar helpers used to pin against tests/test_doc.py:50-51 (src/lib.rs:309-340,384-420) */
double oracle_potential(const oracle_model *m, const double *x, const double *p) { return m->V(x, p); }

int oracle_hesse(oracle_model *m, const double *x, const double *p, double h[4]) {
  int rc = bind_2d(m);
  if (rc) return rc;
  h[0] = m->v00(x, p);
  h[1] = m->v01(x, p);
  h[2] = m->v10(x, p);
  h[3] = m->v11(x, p);
  return ORACLE_OK;
}

/* ---- extended-precision evaluation of the model values (not part of the reference) ---------------
 * Opens a model object emitted with long_double=True (oracle/model_c.py) and evaluates
 * V, v00, v10, v11, grad_norm_squared at n explicit points in x87 extended precision, rounding the
 * results to double.  Used by the tests to measure the reference's own rounding error. */
typedef long double (*exfn2l)(const long double *, const long double *);

int oracle_raw_long_double(const char *so_path, const double *p, size_t n_p, const double *pts, size_t n, double *out) {
  void *h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) FAIL(ORACLE_EIO, "could not open %s: %s", so_path, dlerror());
  const char *names[5] = {"V", "v00", "v10", "v11", "grad_norm_squared"};
  exfn2l fn[5];
  for (int k = 0; k < 5; ++k) {
    fn[k] = (exfn2l)dlsym(h, names[k]);
    if (!fn[k]) { dlclose(h); FAIL(ORACLE_ESYMBOL, "missing symbol %s in %s", names[k], so_path); }
  }
  long double *pl = (long double *)malloc(sizeof(long double) * (n_p ? n_p : 1));
  for (size_t k = 0; k < n_p; ++k) pl[k] = p[k];
  for (size_t i = 0; i < n; ++i) {
    const long double x[2] = {pts[2 * i], pts[2 * i + 1]};
    for (int k = 0; k < 5; ++k) out[5 * i + k] = (double)fn[k](x, pl);
  }
  free(pl);
  dlclose(h);
  return ORACLE_OK;
}

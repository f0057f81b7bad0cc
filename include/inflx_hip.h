/*
 * inflx_hip.h -- C ABI of libinflx_hip.so, the MI355X-native replacement for the grid-sweep
 * entry points of the reference's native module `libinflx_rs`.
 *
 * Every entry point cites the reference interface it replaces (paths relative to the reference
 * repository).  Signatures use plain pointers and sizes only; the caller owns every host buffer
 * (`p`, `out`, `start_stop`, `x`), the library owns the device memory, stream and code object
 * held by an `inflx_model`.  All functions return an `inflx_status`; on failure
 * `inflx_last_error()` returns a thread-local message.  The mapping to the Python exception
 * classes the reference raises (src/err.rs:63-74) is given with each status.
 *
 * Threads: calls on different handles may run concurrently; calls on the SAME handle are serialised
 * by the library (a per-handle lock held for the call) -- the reference gets the same effect from
 * the GIL, which it holds for a whole sweep (src/anguelova.rs:458-465) and which a ctypes / cgo /
 * FFI binding of this library does not.  Closing a handle that another thread still uses remains
 * the caller's error.
 *
 * A "model artefact" is the per-model gfx950 code object written by inflatox_amd.Compiler -- the
 * counterpart of the per-model dylib the reference compiles with zig cc
 * (python/inflatox/compiler.py:568-598) and opens with libloading (src/dylib.rs:67-161).
 */
#ifndef INFLX_HIP_H
#define INFLX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct inflx_model inflx_model;

typedef enum inflx_status {
  INFLX_OK = 0,
  INFLX_ERR_IO = 1,      /* artefact cannot be opened              -> IOError      (err.rs:65)    */
  INFLX_ERR_SYMBOL = 2,  /* artefact lacks a required symbol       -> SystemError  (err.rs:66-69) */
  INFLX_ERR_VERSION = 3, /* artefact built for another ABI version -> SystemError  (err.rs:70)    */
  INFLX_ERR_SHAPE = 4,   /* array shape / parameter count mismatch -> Exception    (err.rs:71)    */
  INFLX_ERR_DEVICE = 5,  /* HIP runtime failure (no counterpart)   -> SystemError                 */
  INFLX_ERR_ARG = 6,     /* invalid argument (NULL, bad enum)      -> ValueError; the reference
                            panics on the analogous conditions (anguelova.rs:473,501)             */
  INFLX_ERR_BASIS = 7,   /* basis not orthonormal (BasisNorm/BasisOth) -> Exception (err.rs:36-37,72) */
  INFLX_ERR_GSL = 8      /* a special function was called outside its domain (see inflx_sf_policy): the reference's GSL error
                            handler prints the reason and panics (err.rs:86-103) -> ArithmeticError       */
} inflx_status;

/* bits of a model's special-function status (csrc/inflx_sf.h) */
typedef enum inflx_sf_bits {
  INFLX_SF_EDOM = 1,      /* an argument outside the function's domain: GSL_EDOM (1) in the reference's GSL */
  INFLX_SF_EDECLINED = 2  /* inside the domain, but the device function could not deliver 1e-13 and returned NaN
                             rather than a plausible number (GSL's counterparts: GSL_ELOSS, GSL_EMAXITER, GSL_EUNIMPL) */
} inflx_sf_bits;

typedef enum inflx_sf_policy_t {
  INFLX_SF_QUIET = 0, /* NaN at the point, nothing else */
  INFLX_SF_FAIL = 1   /* the call that finds a bit set returns INFLX_ERR_GSL -- after its result is complete */
} inflx_sf_policy_t;

/* per-point operation selector; numbering shared with the kernels (csrc/inflx_kernel_abi.h) */
typedef enum inflx_op {
  INFLX_SWEEP_COMPLETE = 0,    /* ops::complete_analysis          anguelova.rs:103-135, 6 f64/point */
  INFLX_SWEEP_CONSISTENCY = 1, /* ops::consistency_only           anguelova.rs:157-163, 1 f64/point */
  INFLX_SWEEP_RAPIDTURN = 2,   /* ops::consistency_rapidturn_only anguelova.rs:143-154, 1 f64/point */
  INFLX_SWEEP_EPSILON_V = 3,   /* ops::epsilon_v_only             anguelova.rs:138-140, 1 f64/point */
  INFLX_SWEEP_RAW = 4,         /* V,v00,v10,v11,|dV|^2 (diagnostic, 5 f64/point; what Potential /
                                  Hesse2D return, hesse_bindings.rs:52-57,106-110,213-231)          */
  INFLX_SWEEP_HESSE = 6        /* v00,v01,v10,v11 (4 f64/point): the projected Hesse matrix row-major, what `hesse` /
                                  `hesse_array` return (src/lib.rs:384-462) -- v01 is the reference's own v01 expression
                                  (hesse_bindings.rs:202-210), evaluated on its own wherever it is not the very expression
                                  of v10                                                             */
} inflx_op;

typedef enum inflx_layout {
  INFLX_AOS = 0, /* [P][rows][N1][K] -- the reference's (N0,N1,6) array (consistency_conditions.py:290) */
  INFLX_SOA = 1  /* [P][K][rows][N1] -- K contiguous planes */
} inflx_layout;

/* thread-local description of the last failure on the calling thread */
const char* inflx_last_error(void);

/* number of visible HIP devices */
int inflx_device_count(int* count);

/*
 * Open a model artefact on `device` -- replaces InflatoxDylib::open (src/dylib.rs:67-161) as
 * reached from open_inflx_dylib (src/lib.rs:108-115): loads the code object, checks VERSION
 * (major.minor against 5.0, src/inflatox_version.rs:48-53, src/lib.rs:50), reads DIM,
 * N_PARAMETERS, MODEL_NAME and resolves the sweep kernels.
 */
int inflx_open(const char* artefact_path, int device, inflx_model** out);
void inflx_close(inflx_model* model);

/*
 * Kernel groups.  Every kernel that evaluates the model inlines the whole model, and a complete artefact holds ~45 of them: building
 * all of it costs a heavy model several seconds of hipcc where the reference's one `zig cc` step takes about one
 * (python/inflatox/compiler.py:568-598).  The artefact inflx_open takes is therefore the model's CORE object -- everything
 * complete_analysis, the on-trajectory complete_analysis and the basis validation need -- or a complete one
 * (Compiler(kernel_groups="all")); the kernels of the other operations live in one small code object per group:
 *   "stats" (the fused summary), "consistency", "rapidturn", "epsilon_v", "raw", "qdif", "hesse", "values" (inflx_ops_on_values).
 * An entry point that needs a group the handle lacks loads the file `<artefact>.<group>` if it exists (inflatox_amd builds the group
 * and puts it there on first use) and fails with INFLX_ERR_SYMBOL otherwise.  inflx_attach loads a group object from any path; it must
 * come from the same generated model and options as the core object (its MODEL_TAG global), else INFLX_ERR_VERSION.
 * inflx_groups: bit mask of the groups loaded so far (bit k = the k-th name of: core, stats, values, consistency, rapidturn,
 * epsilon_v, raw, qdif, hesse).
 */
int inflx_attach(inflx_model* model, const char* group_object_path);
uint32_t inflx_groups(const inflx_model* model);

/* InflatoxDylib::{n_fields,n_pars,name} (src/dylib.rs:285-301) */
uint32_t inflx_n_fields(const inflx_model* model);
uint32_t inflx_n_parameters(const inflx_model* model);
const char* inflx_model_name(const inflx_model* model);
int inflx_device_of(const inflx_model* model);
/* staging facts of the loaded kernels: exported uniform/row/column values and the axis mask of the
 * five model values (bit0 = x[0], bit1 = x[1]) */
int inflx_stage_info(const inflx_model* model, uint32_t* n_uniform, uint32_t* n_row, uint32_t* n_col, uint32_t* out_mask);

/*
 * Drop-ins for the #[pyfunction]s of src/anguelova.rs.  Host buffers, C-contiguous f64:
 *   p           (n_p,)           model parameters
 *   start_stop  (2,2) row-major  [[x0_start,x0_stop],[x1_start,x1_stop]]  (src/lib.rs:117-139)
 *   out         (N0,N1,6) for complete_analysis, (N0,N1) for the single-quantity sweeps
 * `progress` != 0 prints the reference's start/finish lines to stderr and, while a call with a result of 1 GiB or more
 * has been running for longer than 0.5 s, a progress line twice a second (time to completion, grid points/s,
 * percentage: the figures of the reference's progress bar, src/anguelova.rs:42-50); `threads` is accepted for
 * signature compatibility and ignored by these single-device entry points (the *_multi entry points below give it the
 * reference's meaning, with GPUs as the workers).
 */
int inflx_complete_analysis(inflx_model* model, const double* p, size_t n_p, double* out, const double* start_stop,
                            size_t N0, size_t N1, int progress, size_t threads); /* anguelova.rs:458-550 */
int inflx_consistency_only(inflx_model* model, const double* p, size_t n_p, double* out, const double* start_stop,
                           size_t N0, size_t N1, int progress, size_t threads); /* anguelova.rs:176-264 */
int inflx_consistency_rapidturn_only(inflx_model* model, const double* p, size_t n_p, double* out,
                                     const double* start_stop, size_t N0, size_t N1, int progress,
                                     size_t threads); /* anguelova.rs:267-356 */
int inflx_epsilon_v_only(inflx_model* model, const double* p, size_t n_p, double* out, const double* start_stop,
                         size_t N0, size_t N1, int progress, size_t threads); /* anguelova.rs:359-447 */

/* flag_quantum_dif_py (src/anguelova.rs:574-626): out is a (N0,N1) array of bytes (numpy bool), 1 where
 * every component of the normalised potential gradient (basis vector `v`) is <= accuracy */
int inflx_flag_quantum_dif(inflx_model* model, const double* p, size_t n_p, uint8_t* out, const double* start_stop,
                           size_t N0, size_t N1, int progress, double accuracy);

/* on-trajectory variants (src/anguelova.rs:633-977): x is (n,2), out is (n,K) */
int inflx_sweep_on_trajectory(inflx_model* model, int op, const double* p, size_t n_p, const double* x, size_t n,
                              double* out, int progress, size_t threads);

/*
 * Basis validation (src/lib.rs:141-300).  The reference calls the artefact's C functions `v`, `w1` and
 * `inner_prod` (compiler.py:417-472) point by point; here one launch evaluates them at n explicit points.
 *   inflx_basis_on_points: x is (n,2); out is (n,7): v.v, v.w1, w1.w1, v[0], v[1], w1[0], w1[1]
 *   inflx_validate_basis_at_random (lib.rs:142-199, run by open_inflx_dylib(check_basis=true), lib.rs:109-114):
 *     one random parameter vector in [-10,10), 100 random points in [-1,1)^2, accuracy 1e-3;
 *     seed 0 = seed from the OS (the reference's rand::random), any other value = reproducible
 *   inflx_validate_basis_on_domain (lib.rs:207-300): num_points is (n_axes,), start_stop (n_axes,2) row-major
 * Both return INFLX_ERR_BASIS with the reference's message when a norm or overlap misses its target by
 * >= accuracy, and print the reference's warnings for inner products that are not normal numbers.
 */
int inflx_basis_on_points(inflx_model* model, const double* p, size_t n_p, const double* x, size_t n, double* out);
int inflx_validate_basis_at_random(inflx_model* model, uint64_t seed);
int inflx_validate_basis_on_domain(inflx_model* model, const uint32_t* num_points, size_t n_axes, const double* p,
                                   size_t n_p, const double* start_stop, double accuracy);

/*
 * The per-point operations of src/anguelova.rs:99-171 (`mod ops`) applied to GIVEN model values instead of values the
 * kernels evaluate from a model: `values` is (n,5) -- V, v00, v10, v11, grad_norm_squared, what Potential / Hesse2D
 * return (hesse_bindings.rs:52-57,106-110,213-231) --, `out` is (n,9): [0..5] ops::complete_analysis (:103-135),
 * [6] ops::consistency_only (:157-163), [7] ops::consistency_rapidturn_only (:143-154), [8] ops::epsilon_v_only
 * (:138-140).  The model's own functions are not called.  `ieee_only` = 0 evaluates complete_analysis the way the
 * sweep kernels do (divisions without special-case handling where every operand is in mid range, csrc/inflx_ops.h),
 * != 0 with the compiler's IEEE divisions throughout; the two must agree bit for bit on every input, and both are
 * what the parity tests compare with the oracle on arbitrary tuples (specials, zeros, denormals, random bit patterns).
 */
int inflx_ops_on_values(inflx_model* model, const double* values, size_t n, double* out, int ieee_only);

/*
 * Generalised sweep, host result.  P parameter rows (p is (P,n_p)), grid rows
 * [row_begin,row_begin+row_count) of the N0 x N1 grid, result written to the host buffer `out`
 * of P*row_count*N1*K doubles in `layout`.  Rows are processed in device-sized chunks and copied
 * back while the next chunk computes.
 */
int inflx_sweep_host(inflx_model* model, int op, const double* p, size_t P, size_t n_p, double* out,
                     const double* start_stop, size_t N0, size_t N1, size_t row_begin, size_t row_count, int layout);

/*
 * As inflx_sweep_host with the planes layout, but only planes [first_plane, first_plane + n_planes) of every parameter row cross
 * PCIe: `out` holds P*n_planes*row_count*N1 doubles, (P, n_planes, row_count, N1).  The device evaluates all K values of the
 * operation in one pass; a caller that wants one of them (calc_V_array: plane 0 of the five raw values, reference
 * consistency_conditions.py:67-101; calc_H_array: planes 1-3, :119-156) does not pay the copy of the others.
 */
int inflx_sweep_host_planes(inflx_model* model, int op, const double* p, size_t P, size_t n_p, double* out,
                            const double* start_stop, size_t N0, size_t N1, size_t row_begin, size_t row_count,
                            size_t first_plane, size_t n_planes);

/*
 * Generalised sweep, device-resident result: `d_out` is device memory of at least
 * P*row_count*N1*K*8 bytes on the model's device (`d_out_bytes` is checked); the kernel is
 * enqueued on `stream` (a hipStream_t; NULL = the model's own stream) and the call returns without
 * synchronising.  This is what the multi-GPU sharding and the benchmark use.
 * `p` is copied into pinned memory owned by the model before the call returns: the caller may free or
 * change it immediately, and consecutive calls with different parameters need no synchronisation.
 */
int inflx_sweep_device(inflx_model* model, int op, const double* p, size_t P, size_t n_p, void* d_out,
                       size_t d_out_bytes, const double* start_stop, size_t N0, size_t N1, size_t row_begin,
                       size_t row_count, int layout, void* stream);

/*
 * Which kernels a sweep of this shape takes (introspection for tests and profiles; no launch):
 *   plan[0] inflx_path; for INFLX_PATH_ROW_STREAM also plan[1] = parameter rows per row-table batch,
 *   plan[2] = number of batches the call is cut into, plan[3] = replicas of a row's table entry;
 *   for INFLX_PATH_TILE plan[1] = parameter rows per launch, plan[2] = number of launches, plan[3] = grid rows per workgroup
 *   tile of the first launch (the full height for large launches, lower for launches of fewer than ~4096 tiles).
 */
typedef enum inflx_path {
  INFLX_PATH_TILE = 0,       /* inflx_sweep_tile_*: some model value depends on x[1]                    */
  INFLX_PATH_ROW_STREAM = 1, /* inflx_sweep_rowvals_* + inflx_sweep_rowstream6/_planes (row-only model) */
  INFLX_PATH_ROWS = 2,       /* inflx_sweep_rows_*: row-only model, result shape the streams do not cover */
  INFLX_PATH_COL_STREAM = 3  /* inflx_sweep_colvals_* + inflx_sweep_colstream (no value depends on x[0])      */
} inflx_path;
int inflx_sweep_plan(const inflx_model* model, int op, size_t P, size_t N1, size_t row_count, int layout, uint32_t plan[4]);

/*
 * Per-call options of the device-result sweeps (the *_ex entry points; the plain ones pass INFLX_SWEEP_DEFAULT).
 *   INFLX_SWEEP_FORCE_TILE  evaluate EVERY grid point in the tile kernels (one lane per grid point, inflx_sweep_tile_*), also for a
 *                           model whose values ignore a grid axis and would otherwise take a broadcast path (one evaluated line
 *                           + store stream).  That is what the reference does for every model -- its loop calls the five model
 *                           functions at every point whatever they depend on (src/anguelova.rs:526-539) -- and the results are
 *                           the broadcast path's bit for bit; the flag exists so that the per-lane rate of such a workload can be
 *                           measured and the two paths compared (bench.py `c1_tile_path_*`, tests/test_parity_gpu.py).
 */
typedef enum inflx_sweep_flags {
  INFLX_SWEEP_DEFAULT = 0,
  INFLX_SWEEP_FORCE_TILE = 1
} inflx_sweep_flags;
int inflx_sweep_device_ex(inflx_model* model, int op, const double* p, size_t P, size_t n_p, void* d_out,
                          size_t d_out_bytes, const double* start_stop, size_t N0, size_t N1, size_t row_begin,
                          size_t row_count, int layout, void* stream, unsigned flags);
int inflx_sweep_plan_ex(const inflx_model* model, int op, size_t P, size_t N1, size_t row_count, int layout, unsigned flags,
                        uint32_t plan[4]);

/*
 * Host helper threads (no launch, no device needed).  The host-result paths start helper threads -- page residency ahead of the
 * device-to-host copy, the streaming-store fill of results that are constant along a grid axis.  Their number comes from ONE
 * budget per process, read once: the CPUs of the scheduler affinity mask, cut down to the smallest cgroup CPU quota between the
 * process's cgroup and the root (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us); INFLX_HOST_THREADS overrides it.  The
 * reference sizes its rayon pool the same way (`threads` = 0 -> every core the process has, src/anguelova.rs:467,524-525).  A
 * multi-device call divides the budget among the devices at work.  out = {budget, residency helpers, fill helpers} of one device
 * pipeline when `devices_at_work` pipelines run at once (0 is read as 1).
 */
int inflx_host_threads(unsigned devices_at_work, unsigned out[3]);

/* As inflx_sweep_device, `repeats` times back to back between two HIP events recorded on the
 * launch stream; returns the mean duration of one sweep in milliseconds (synchronises).  A sweep of a
 * model whose values do not depend on x[1] is two launches (per-row evaluation, then the store stream);
 * with `dominant_only` == 1 only the dominant one -- the store stream -- is repeated and timed; with
 * `dominant_only` == 2 the full sweeps are enqueued as in mode 0 and the value returned is the time spent
 * between event pairs recorded around every launch of the dominant kernel, per sweep: its duration inside
 * the pipeline (side-stream evaluation overlapping, cross-stream waits in place). */
int inflx_sweep_device_timed(inflx_model* model, int op, const double* p, size_t P, size_t n_p, void* d_out,
                             size_t d_out_bytes, const double* start_stop, size_t N0, size_t N1, size_t row_begin,
                             size_t row_count, int layout, void* stream, int repeats, int dominant_only,
                             float* ms_per_launch);
/*
 * The same with the timing mode by name and the inflx_sweep_flags of inflx_sweep_device_ex:
 *   INFLX_TIME_BACK_TO_BACK   mode 0 above: `repeats` sweeps enqueued back to back between two events -- the THROUGHPUT of a scan:
 *                             the tables (per-row values) of sweep n+1 are evaluated on a side stream under the kernels of sweep n;
 *   INFLX_TIME_DOMINANT_ONLY  mode 1;   INFLX_TIME_IN_PIPELINE  mode 2;
 *   INFLX_TIME_SINGLE_CALL    what ONE call costs a caller whose handle is idle (one call = one sweep, reference
 *                             python/inflatox/consistency_conditions.py:290-300): before every repetition the call waits until
 *                             nothing of this handle is left on the device, then brackets one whole inflx_sweep_device_ex call --
 *                             stage tables / per-row evaluation AND the sweep kernel -- with two events on the launch stream;
 *                             returns the mean over `repeats` calls.
 */
typedef enum inflx_timing {
  INFLX_TIME_BACK_TO_BACK = 0,
  INFLX_TIME_DOMINANT_ONLY = 1,
  INFLX_TIME_IN_PIPELINE = 2,
  INFLX_TIME_SINGLE_CALL = 3
} inflx_timing;
int inflx_sweep_device_timed_ex(inflx_model* model, int op, const double* p, size_t P, size_t n_p, void* d_out,
                                size_t d_out_bytes, const double* start_stop, size_t N0, size_t N1, size_t row_begin,
                                size_t row_count, int layout, void* stream, int repeats, int mode, unsigned flags,
                                float* ms_per_launch);

/*
 * Running summary of the six complete_analysis outputs over a sweep (no counterpart in the reference's
 * native module; its users compute such statistics afterwards with numpy, e.g. np.nanmax in
 * tests/test_doc.py:58).  NaN values are ignored, +-Inf count as values; min/max are +Inf/-Inf for a
 * quantity that is NaN everywhere.
 */
typedef struct inflx_summary {
  double min[6];     /* per quantity k: consistency, epsilon_V, epsilon_H, eta, delta, omega */
  double max[6];
  uint64_t count[6]; /* number of non-NaN values */
} inflx_summary;

/*
 * complete_analysis sweep (AoS layout) with the summary reduced on the device inside the sweep kernels
 * (wave-wide butterfly reduction + one set of f64 atomics per wavefront).  `d_out` may be NULL: the sweep
 * then only evaluates and reduces and writes no result array.  Synchronous: returns when the summary is
 * in `*summary`.
 */
int inflx_sweep_device_stats(inflx_model* model, const double* p, size_t P, size_t n_p, void* d_out, size_t d_out_bytes,
                             const double* start_stop, size_t N0, size_t N1, size_t row_begin, size_t row_count,
                             void* stream, inflx_summary* summary);

/* wait for everything enqueued on the model's own stream (under INFLX_SF_FAIL this is where the asynchronous device-resident
 * sweeps report a special-function error) */
int inflx_synchronize(inflx_model* model);

/*
 * Special functions outside their domain.  The reference links GSL for sympy's Bessel and hypergeometric functions
 * (python/inflatox/compiler.py:123-212) and, when the artefact says USE_GSL = 1, installs an error handler that prints the reason
 * and panics (compiler.py:145-149 `err_setup`, src/dylib.rs:141-148, src/err.rs:86-103): a sweep that evaluates, say, K_nu at
 * x <= 0 anywhere on its grid does not return.  The device counterparts (csrc/inflx_sf.h) return NaN for the point and set a bit in
 * the code object's status word; what the host makes of it is the handle's policy:
 *   INFLX_SF_FAIL  (default when USE_GSL = 1) -- every call that hands results to the host (the inflx_sweep_host* family, the named
 *                  sweeps, inflx_sweep_on_trajectory, inflx_sweep_device_stats, the *_multi forms, and inflx_synchronize for
 *                  the asynchronous device-resident sweeps) returns INFLX_ERR_GSL, with the reference handler's first line in
 *                  inflx_last_error(), AFTER the result has been written: the caller's array is complete, NaN at the points;
 *   INFLX_SF_QUIET (default otherwise)        -- NaN at the points, INFLX_OK.
 * inflx_sf_status: OR of the bits (inflx_sf_bits) set since they were last cleared, after waiting for the handle's streams;
 * `clear` != 0 resets them.  Reported conditions: the argument domains GSL documents (nu >= 0; Y, K, y_l for x > 0; J_nu, I_nu, j_l
 * for x >= 0; 0F1 / 1F1 / 2F1 with the lower parameter a non-positive integer; 2F1 outside -1 <= x < 1; 2F0 for x > 0) and this
 * implementation's own refusals.  A NaN argument is not an error (it propagates, as through GSL's comparisons).  NOT reported:
 * overflow and underflow (+-inf / 0 here; GSL_EOVRFLW / GSL_EUNDRFLW reach the reference's handler too).  GSL itself is not part of the
 * reference's sources or of this image: which arguments it rejects is taken from its manual, not from a run.
 */
int inflx_sf_status(inflx_model* model, unsigned* bits, int clear);
int inflx_sf_policy(inflx_model* model, int policy);
/* the artefact's USE_GSL global (0 / 1): Compiler(link_gsl=True), compiler.py:558, dylib.rs:135-141 */
int inflx_uses_gsl(const inflx_model* model);

/*
 * One call, several GPUs.  The reference's knob for "use the whole machine" is the `threads` argument of its sweeps:
 * threads = None -> 0 -> a rayon pool over all cores (python/inflatox/consistency_conditions.py:297,
 * src/anguelova.rs:185,236,524-540).  Here the workers are the node's GPUs: an `inflx_multi` is the model artefact opened
 * on several devices (`devices` NULL or n_dev <= 0: every visible device; the same device may be listed more than once),
 * and one call splits the outermost axis of the sweep into one contiguous balanced block per device --
 *   the parameter axis when P >= devices, else the grid's row axis (inflx_shard_plan: plan = {axis (0 parameter rows,
 *   1 grid rows), p_begin, p_count, row_begin, row_count}, the first n % world parts get one item more) --
 * runs every device's pipeline (launch, device-to-host copy, page residency) on a host thread of its own and lets each
 * device copy its slab straight into its place in the caller's array.  No exchange between devices is needed: every grid
 * point and every parameter row is independent (the reference's threads do not communicate either).
 *   inflx_sweep_host_multi:        as inflx_sweep_host for the whole grid; `out` is the whole (P, N0, N1, K) [AOS] or
 *                                  (P, K, N0, N1) [SOA] array; `max_devices` = 0 uses every device of the handle, k > 0 the
 *                                  first k; `progress` != 0 prints the start / finish lines and, for calls longer than
 *                                  0.5 s, progress lines (time to completion, grid points/s, percentage: the figures of the
 *                                  reference's progress bar, src/anguelova.rs:42-50) to stderr
 *   inflx_complete_analysis_multi: the drop-in shape of libinflx_rs.complete_analysis (anguelova.rs:458-465) with
 *                                  `threads` = 0: every device of the handle, k: at most k
 *   inflx_sweep_stats_multi:       the summary of inflx_sweep_device_stats over the whole grid, every device reducing its
 *                                  block inside its sweep kernels (no array is written), combined on the host
 * Results are bit-identical to the single-device calls whatever the split: a grid point's value depends on its index
 * and parameter row only.
 */
typedef struct inflx_multi inflx_multi;
int inflx_shard_plan(size_t P, size_t N0, int world, int rank, size_t plan[5]);
int inflx_open_multi(const char* artefact_path, const int* devices, int n_dev, inflx_multi** out);
void inflx_close_multi(inflx_multi* multi);
int inflx_multi_device_count(const inflx_multi* multi);
inflx_model* inflx_multi_handle(const inflx_multi* multi, int index); /* the handle of device `index` (owned by `multi`) */
int inflx_sweep_host_multi(inflx_multi* multi, int op, const double* p, size_t P, size_t n_p, double* out,
                           const double* start_stop, size_t N0, size_t N1, int layout, int progress, size_t max_devices);
int inflx_complete_analysis_multi(inflx_multi* multi, const double* p, size_t n_p, double* out, const double* start_stop,
                                  size_t N0, size_t N1, int progress, size_t threads);
int inflx_sweep_stats_multi(inflx_multi* multi, const double* p, size_t P, size_t n_p, const double* start_stop, size_t N0,
                            size_t N1, size_t max_devices, inflx_summary* summary);
/*
 * Device-resident results on several GPUs (no counterpart in the reference; the exchange step SURVEY section 8e allows
 * "when the result must be device-resident on every GPU").
 *   inflx_sweep_device_multi:    device k sweeps its block (inflx_shard_plan with world = the handle's device count, rank = k)
 *                                into d_out[k], device memory ON device k laid out as the block; asynchronous, enqueued on
 *                                streams[k] (NULL array / entry: the handle's own stream).  No data crosses a link.
 *   inflx_sweep_allgather_multi: every d_full[k] (device memory on device k, the whole (P, N0, N1, K) AOS array) holds the
 *                                whole result on return.  Each device sweeps its block in place into its slice of its own
 *                                buffer and pushes the slice to every peer with hipMemcpyPeerAsync on a stream per peer:
 *                                xGMI is point to point, the n - 1 pushes of a device use n - 1 links at once (a direct
 *                                all-to-all broadcast, slab / link-rate instead of a ring's (n - 1) steps).  Synchronous.
 */
int inflx_sweep_device_multi(inflx_multi* multi, int op, const double* p, size_t P, size_t n_p, void* const* d_out,
                             const size_t* d_out_bytes, const double* start_stop, size_t N0, size_t N1, int layout,
                             void* const* streams);
int inflx_sweep_allgather_multi(inflx_multi* multi, int op, const double* p, size_t P, size_t n_p, void* const* d_full,
                                size_t d_full_bytes, const double* start_stop, size_t N0, size_t N1);
/*
 *   inflx_sweep_allgather_multi_ex: the same call with the exchange step chosen by the caller --
 *     INFLX_GATHER_PEER_PUSH  the direct pushes described above (what inflx_sweep_allgather_multi does; any split, also
 *                             several handles on one device);
 *     INFLX_GATHER_RCCL       ONE in-place ncclAllGather over xGMI per contiguous image (the whole array when the parameter
 *                             axis is split, one per parameter row when grid rows are), all of them fused in one RCCL group
 *                             and enqueued on every device's sweep stream behind its sweep: the collective BASELINE.json's
 *                             north_star names ("one RCCL ncclAllGather (equal slabs ...)").  Needs equal
 *                             blocks (the split axis divides by the device count), one device per handle and a buffer per
 *                             device; INFLX_ERR_SHAPE / INFLX_ERR_ARG otherwise.  RCCL is bound at run time (the copy already
 *                             mapped into the process if there is one, e.g. PyTorch's, else librccl.so.1): libinflx_hip.so
 *                             does not link it, and a caller that never asks for it never loads it.
 *                             EXPERIMENTAL until a run on two or more GPUs has passed: no node with more than one GPU has been
 *                             available to this build, so the collective has executed with one rank only; the (send, recv, count) of
 *                             every rank are checked against inflx_shard_plan on the host (tests/host_units.cpp).  On an error the
 *                             call waits for every device's streams before it returns.
 */
typedef enum inflx_gather {
  INFLX_GATHER_PEER_PUSH = 0,
  INFLX_GATHER_RCCL = 1
} inflx_gather;
int inflx_sweep_allgather_multi_ex(inflx_multi* multi, int op, const double* p, size_t P, size_t n_p, void* const* d_full,
                                   size_t d_full_bytes, const double* start_stop, size_t N0, size_t N1, int gather);

#ifdef __cplusplus
}
#endif
#endif /* INFLX_HIP_H */

// Grid-sweep kernels for one inflation model, gfx950 (MI355X / CDNA4) only.
//
// Compiled once per model by inflatox_amd.Compiler:
//   hipcc --offload-arch=gfx950 --genco -O3 -DINFLX_MODEL_HEADER="<generated>.h" this_file -o model.hsaco
// The generated header provides the model as four straight-line device functions
// (inflx_stage_uniform / _row / _col / _point, see inflatox_amd/staging.py) that are inlined
// here -- the counterpart of the reference calling V, v11, v10, v00, grad_norm_squared through
// five dlsym'd pointers per grid point (src/anguelova.rs:110,119; src/hesse_bindings.rs:213-231).
//
// Work decomposition (replaces rayon's par_chunks_exact_mut(6) over points, anguelova.rs:526-539):
//
//  * tile kernels (inflx_sweep_tile_*): a 256-thread workgroup owns a tile of TILE_ROWS grid rows
//    x 256 grid columns for one parameter row.  Lane <-> column j (the fast axis of the output),
//    so a wavefront covers 64 consecutive points of one grid row = 64*K*8 contiguous bytes.
//    Parameter-only, row-only and column-only sub-expressions are evaluated ONCE PER LAUNCH by inflx_stage_tables
//    (one lane per parameter row / grid row / grid column) into tables; a workgroup loads its column's values into
//    registers and its 32 rows' values into LDS, from where all lanes read them as a broadcast.
//    For the 6-value AoS result a wavefront transposes its 64x6 block through a private 3 KiB LDS
//    buffer so that every global store instruction writes 1 KiB of contiguous memory (16 B/lane).
//
//  * row-broadcast path, used when no model value depends on x[1] (e.g. the hyperbolic benchmark model):
//    the per-point operation is then a function of the grid row only.  inflx_sweep_rowvals_* evaluates
//    it once per row into a small table, inflx_sweep_rowstream6 / _planes broadcast it along the row:
//    a pure store stream, ONE 16-byte store per thread and 4 KiB per workgroup, which is what the
//    HBM roofline prices (6.9-7.2 TB/s measured).  inflx_sweep_rows_* is the fallback for result
//    shapes the stream kernels do not cover (odd N1 planes, the 5-value AoS diagnostic).
//
//  * column-broadcast path, used when no model value depends on x[0]: inflx_sweep_colvals_* evaluates the image of
//    one output row (one lane per grid column), inflx_sweep_colstream copies it into every grid row -- the same
//    store-stream shape, fed by an L2-resident 16-byte load per thread.
//
//  * inflx_sweep_traj_*: explicit point lists (src/anguelova.rs:633-977), one thread per point.
//
// Numerics: IEEE-strict FP64 (no fast-math, denormals kept, correctly rounded div/sqrt).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "inflx_device_math.h"
#include "inflx_kernel_abi.h"
#include "inflx_ops.h"

#ifndef INFLX_MODEL_HEADER
#error "INFLX_MODEL_HEADER must name the generated model header"
#endif
#include INFLX_MODEL_HEADER

#ifndef INFLX_TILE_ROWS
#define INFLX_TILE_ROWS 32
#endif
#ifndef INFLX_ROWS_PER_BLOCK
#define INFLX_ROWS_PER_BLOCK 4
#endif
#ifndef INFLX_NT_STORES
#define INFLX_NT_STORES 1
#endif
// tile kernels: minimum wavefronts per SIMD the register allocator must leave room for
// (second argument of __launch_bounds__; 512 registers per lane / this = the VGPR+AGPR cap)
#ifndef INFLX_MIN_WAVES
#define INFLX_MIN_WAVES 2
#endif
// tile kernels: keep the parameter-only (U) values in LDS instead of 2 VGPRs each
#ifndef INFLX_U_IN_LDS
#define INFLX_U_IN_LDS (INFLX_NU > 8)
#endif
// tile kernels: the epilogue's polynomial coefficients in LDS instead of 68 resident vector registers
#ifndef INFLX_EPILOGUE_CONSTANTS_IN_LDS
#define INFLX_EPILOGUE_CONSTANTS_IN_LDS 1
#endif

static_assert(INFLX_DIM == 2, "the sweep kernels need a two-field model (Hesse2D, hesse_bindings.rs:203)");

constexpr int kThreads = 256;
constexpr int kWave = 64;
constexpr int kTileRows = INFLX_TILE_ROWS;
constexpr int kRowsPerBlock = INFLX_ROWS_PER_BLOCK;
constexpr int kNU = INFLX_NU > 0 ? INFLX_NU : 1;
constexpr int kNR = INFLX_NR > 0 ? INFLX_NR : 1;
// stride of a row's R values in the stage table and in LDS: even, so that every pair of values is one aligned 16-byte
// LDS read (an odd stride costs ds_read2_b64 pairs at twice the LDS cycles of ds_read_b128; D5 has 69 values, EGNO 82)
constexpr int kNRs = (kNR + 1) & ~1;
constexpr int kNC = INFLX_NC > 0 ? INFLX_NC : 1;
constexpr int kNP = INFLX_N_PARAMETERS > 0 ? INFLX_N_PARAMETERS : 1;

typedef double inflx_d2 __attribute__((ext_vector_type(2)));

// ---- metadata the host reads back (counterpart of the dylib's VERSION/DIM/... symbols,
// ---- src/dylib.rs:32-48, emitted by the reference at compiler.py:546-560) ---------------------
// (non-const on purpose: a const namespace-scope object would get internal linkage and vanish from
// the code object's dynamic symbol table, where hipModuleGetGlobal looks it up)
#define INFLX_EXPORT extern "C" __device__ __attribute__((used, visibility("default")))
// (the two overrides exist for the negative ABI tests only: an artefact of another ABI version must be refused by
// inflx_open like InflatoxDylib::open refuses it, src/dylib.rs:92-104; an artefact that says it has three fields must be
// refused by the sweeps like Hesse2D::new refuses it, src/hesse_bindings.rs:203)
#ifndef INFLX_ABI_VERSION_MAJOR
#define INFLX_ABI_VERSION_MAJOR 5
#endif
#ifndef INFLX_EXPORTED_DIM
#define INFLX_EXPORTED_DIM INFLX_DIM
#endif
INFLX_EXPORT uint16_t VERSION[3] = {INFLX_ABI_VERSION_MAJOR, 0, 0};
INFLX_EXPORT uint32_t DIM = INFLX_EXPORTED_DIM;
INFLX_EXPORT uint32_t N_PARAMETERS = INFLX_N_PARAMETERS;
INFLX_EXPORT char MODEL_NAME[] = INFLX_MODEL_NAME;
#ifndef INFLX_USE_GSL
#define INFLX_USE_GSL 0
#endif
INFLX_EXPORT char USE_GSL = INFLX_USE_GSL;  // Compiler(link_gsl=True): special functions come from inflx_sf.h
INFLX_EXPORT InflxKernelInfo INFLX_KERNEL_INFO = {
    INFLX_KERNEL_ABI, INFLX_NU, INFLX_NR, INFLX_NC, INFLX_OUT_MASK, INFLX_TILE_ROWS, kThreads, INFLX_ROWS_PER_BLOCK,
    kThreads, 0};

// ---- kernel groups: which entry points this code object carries ---------------------------------------------------------------
// Every kernel that evaluates the model inlines the whole model, and a complete artefact has ~45 of them: building all of it costs
// a heavy model (D5) 8 s of hipcc, where the reference's one `zig cc` step takes about one (python/inflatox/compiler.py:568-598).
// A code object therefore carries the groups INFLX_KERNEL_GROUPS names: Compiler.compile() builds the CORE object -- everything
// complete_analysis needs -- and the groups of the other operations are built when first used and loaded beside it (inflx_attach,
// or the file `<artefact>.<group>` next to the artefact, which libinflx_hip.so loads on demand).  Default: everything.
#ifndef INFLX_KERNEL_GROUPS
#define INFLX_KERNEL_GROUPS 0xffffffffu
#endif
#define INFLX_HAS_GROUP(g) ((INFLX_KERNEL_GROUPS & (g)) != 0u)
INFLX_EXPORT uint32_t INFLX_GROUPS = INFLX_KERNEL_GROUPS & INFLX_GROUP_ALL;
// which model this is (content hash of the generated header): an attached group must belong to the same one
#ifndef INFLX_MODEL_TAG
#define INFLX_MODEL_TAG ""
#endif
INFLX_EXPORT char MODEL_TAG[] = INFLX_MODEL_TAG;

template <int OP>
struct OpWidth {
  static constexpr int K = (OP == INFLX_OP_COMPLETE) ? 6 : (OP == INFLX_OP_RAW ? 5 : (OP == INFLX_OP_HESSE ? 4 : 1));
};

// TABLE / `kc`: the polynomial coefficients of the epilogue's atan / tan come from a table in LDS (tile kernels) instead
// of literals (inflx_ops.h)
template <int OP, bool TABLE = false>
__device__ __forceinline__ void apply_op(const InflxModelValues& mv, double* o, [[maybe_unused]] double accuracy = 0.0, [[maybe_unused]] const double* kc = nullptr) {
  if constexpr (OP == INFLX_OP_QDIF) {
    o[0] = inflx_op_flag_quantum_diff(mv, accuracy) ? 1.0 : 0.0;
  } else if constexpr (OP == INFLX_OP_COMPLETE) {
    inflx_op_complete_analysis<TABLE>(mv, o, kc);
  } else if constexpr (OP == INFLX_OP_CONSISTENCY) {
    o[0] = inflx_op_consistency_only(mv);
  } else if constexpr (OP == INFLX_OP_RAPIDTURN) {
    o[0] = inflx_op_consistency_rapidturn_only(mv);
  } else if constexpr (OP == INFLX_OP_EPSILON_V) {
    o[0] = inflx_op_epsilon_v_only(mv);
  } else if constexpr (OP == INFLX_OP_HESSE) {
    o[0] = mv.v00;
    o[1] = mv.v01;
    o[2] = mv.v10;
    o[3] = mv.v11;
  } else {
    o[0] = mv.V;
    o[1] = mv.v00;
    o[2] = mv.v10;
    o[3] = mv.v11;
    o[4] = mv.g;
  }
}

// INFLX_OP_HESSE: the reference's own v01 (`hesse`, src/lib.rs:384-420, returns [v00, v01, v10, v11]).  Where the symbolic stage
// gave v01 the very expression of v10 (INFLX_V01_IS_V10: every example model -- sympy's canonical form of the symmetric
// projection) the value IS v10, bit for bit; otherwise the generated header carries v01 as a function of its own, printed and
// evaluated like the reference's C function (its own stages inline, once per point: this is a helper off the sweep path).
template <int OP>
__device__ __forceinline__ void fill_v01([[maybe_unused]] InflxModelValues& mv, [[maybe_unused]] double x0, [[maybe_unused]] double x1,
                                         [[maybe_unused]] const double* A) {
  if constexpr (OP == INFLX_OP_HESSE) {
#if INFLX_V01_IS_V10
    mv.v01 = mv.v10;
#else
    mv.v01 = inflx_v01_point(x0, x1, A);
#endif
  }
}

// The per-point operation with its divisions spelled without special-case handling (inflx_ops.h); false = this point
// needs the IEEE spelling.  complete_analysis (eleven divisions) and consistency_only (five, two of them by V) have such
// a variant; consistency_rapidturn_only (three divisions, no shared denominator) and epsilon_v_only (one) would pay as much
// for the range test as the fix-ups cost.
template <int OP, bool TABLE = false>
__device__ __forceinline__ bool apply_op_quick(const InflxModelValues& mv, double* o, [[maybe_unused]] double accuracy = 0.0, [[maybe_unused]] const double* kc = nullptr) {
#ifndef INFLX_EXPERIMENT_IEEE_EPILOGUE  // (A/B experiments: the compiler's divisions in the hot loop as well)
  if constexpr (OP == INFLX_OP_COMPLETE) {
    return inflx_op_complete_analysis_quick<TABLE>(mv, o, kc);
  } else if constexpr (OP == INFLX_OP_CONSISTENCY) {
    return inflx_op_consistency_only_quick(mv, o[0]);
  } else
#endif
  {
    apply_op<OP, TABLE>(mv, o, accuracy, kc);
    return true;
  }
}

// One 16-byte store of a store stream (row / column broadcast paths), with its cache-policy bits spelled out.
// scripts/micro/store_bw.hip, variant Q, 3.2 GB of 16-byte stores, one per thread (profiles/r03_store_policy.txt):
// no bits 6.86-6.91 TB/s, `nt` (what __builtin_nontemporal_store emits) 6.71, `sc1` 7.06-7.09, `sc0 sc1` 7.06-7.09,
// `sc1 nt` 7.03-7.04, `sc0 sc1 nt` 7.06 -- the system-scope bit is worth 5 % over the plain streaming hint.  The streaming
// hint stays: without it the stream sweeps the row table out of the Infinity Cache (section 4.1 of DESIGN.md): in the product
// kernel, hyperbolic 8192^2, `nt` 0.463 ms, `sc1 nt` 0.445, `sc0 sc1 nt` 0.444, `sc1` / `sc0 sc1` 0.668.  (The tile kernels'
// stores gain nothing from it -- doc 0.212 -> 0.216 ms, D5 0.438 -> 0.486 with the asm spelling -- and keep the builtin.)
#ifndef INFLX_OPAQUE_ROW_BASE
#define INFLX_OPAQUE_ROW_BASE 1
#endif
#if INFLX_OPAQUE_ROW_BASE
#define INFLX_KEEP_SCALAR(x) asm volatile("" : "+s"(x))
#else
#define INFLX_KEEP_SCALAR(x) (void)(x)
#endif
#ifndef INFLX_DRAIN_LOADS_BEFORE_ROW_LOOP
#define INFLX_DRAIN_LOADS_BEFORE_ROW_LOOP 1
#endif
#ifndef INFLX_STREAM_STORE
#define INFLX_STREAM_STORE 1
#endif
__device__ __forceinline__ void stream_store_d2(inflx_d2* p, inflx_d2 v) {
#if INFLX_STREAM_STORE == 0
  __builtin_nontemporal_store(v, p);
#elif INFLX_STREAM_STORE == 1
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory");
#elif INFLX_STREAM_STORE == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(v) : "memory");
#elif INFLX_STREAM_STORE == 3
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
#endif
}

__device__ __forceinline__ void store_d2(inflx_d2* p, inflx_d2 v) {
#if INFLX_NT_STORES
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// one scalar result at element offset `off`: an f64, or for the boolean flag sweep one byte
// (the reference fills a numpy bool array, consistency_conditions.py:515)
// `off` is wave-uniform and `lane_off` a 32-bit per-thread offset: the address is "scalar base + vector offset", which the
// store instruction forms by itself (no 64-bit pointer arithmetic per lane)
template <int OP>
__device__ __forceinline__ void store_scalar(double* out, uint64_t off, unsigned lane_off, double v) {
  if constexpr (OP == INFLX_OP_QDIF) {
    (reinterpret_cast<uint8_t*>(out) + off)[lane_off] = v != 0.0 ? 1 : 0;
  } else {
#if INFLX_NT_STORES
    __builtin_nontemporal_store(v, out + off + lane_off);
#else
    (out + off)[lane_off] = v;
#endif
  }
}

__device__ __forceinline__ void store_d1(double* p, double v) {
#if INFLX_NT_STORES
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

__device__ __forceinline__ void load_params(const double* __restrict__ params, unsigned p, double* A) {
#pragma unroll
  for (int k = 0; k < INFLX_N_PARAMETERS; ++k) A[k] = params[(size_t)p * INFLX_N_PARAMETERS + k];
}

// ---- running summary of the six outputs (the "consistency mask" statistics: where does the condition
// ---- hold, what range do epsilon_H, eta, omega take) -- fused into the sweep, no second pass over HBM
struct StatAcc {
  double mn[6], mx[6];
  unsigned long long cnt[6];
};

__device__ __forceinline__ void stat_init(StatAcc& s) {
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    s.mn[k] = __builtin_inf();
    s.mx[k] = -__builtin_inf();
    s.cnt[k] = 0;
  }
}

// NaN-ignoring (like np.nanmin / np.nanmax, reference tests/test_doc.py:58); +-Inf count as values
__device__ __forceinline__ void stat_add(StatAcc& s, const double* o, unsigned long long weight) {
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (o[k] == o[k]) {
      s.mn[k] = fmin(s.mn[k], o[k]);
      s.mx[k] = fmax(s.mx[k], o[k]);
      s.cnt[k] += weight;
    }
  }
}

// wave-wide butterfly reduction over the 64 lanes, then one set of device-scope atomics per wavefront
__device__ __forceinline__ void stat_flush(StatAcc& s, double* stats) {
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
      s.mn[k] = fmin(s.mn[k], __shfl_xor(s.mn[k], off, kWave));
      s.mx[k] = fmax(s.mx[k], __shfl_xor(s.mx[k], off, kWave));
      s.cnt[k] += __shfl_xor(s.cnt[k], off, kWave);
    }
  }
  if ((threadIdx.x & (kWave - 1)) == 0) {
    unsigned long long* counts = reinterpret_cast<unsigned long long*>(stats + 12);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (s.cnt[k]) {
        __hip_atomic_fetch_min(stats + k, s.mn[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(stats + 6 + k, s.mx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(counts + k, s.cnt[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// ================================================================================================
// tile kernels: general case (some value depends on x[1])
// ================================================================================================
template <int OP, bool STATS = false, bool STORE = true>
__device__ __forceinline__ void sweep_tile(const InflxSweepArgs& a) {
  constexpr int K = OpWidth<OP>::K;
  static_assert(!STATS || OP == INFLX_OP_COMPLETE, "the summary is defined for the six outputs of complete_analysis");
  StatAcc acc;
  if constexpr (STATS) stat_init(acc);
  __shared__ __attribute__((aligned(16))) double Rs[kTileRows][kNRs];
  __shared__ __attribute__((aligned(16))) double tbuf[kThreads / kWave][kWave * 6];
#if INFLX_EPILOGUE_CONSTANTS_IN_LDS
  // the 34 polynomial coefficients of the epilogue's atan / tan: read back as LDS broadcasts straight into the
  // accumulator of each Horner step instead of living in 68 vector registers across the row loop (inflx_ops.h)
  __shared__ double epilogue_constants[kInflxEpilogueConstants];
  if (OP == INFLX_OP_COMPLETE && threadIdx.x < (unsigned)kInflxEpilogueConstants)
    epilogue_constants[threadIdx.x] = threadIdx.x < (unsigned)kInflxAtanTerms ? kInflxAtanC[threadIdx.x] : kInflxTanC[threadIdx.x - kInflxAtanTerms];
  const double* const kc = epilogue_constants;  // published by the barrier below
  constexpr bool kTable = OP == INFLX_OP_COMPLETE;
#else
  const double* const kc = nullptr;
  constexpr bool kTable = false;
#endif

  const unsigned tid = threadIdx.x;
  const unsigned lane = tid & (kWave - 1);
  const unsigned wave = tid / kWave;
  const unsigned p = blockIdx.z;

  double A[kNP];
  load_params(a.params, p, A);
  // Stage values come from the tables inflx_stage_tables wrote for this launch (same stage code, evaluated once per
  // parameter row / grid row / grid column instead of once per workgroup: the U -> R chain is a serial latency of
  // thousands of dependent instructions on one lane, resp. 32 lanes, that every workgroup used to pay at its start --
  // 6 % of the D5 sweep and 8 % of the angular one, measured by skipping it).
  const uint64_t slab_rows = a.stream_units;  // grid rows covered by this launch's tables, first one = stream_row0
  const double* __restrict__ utab = a.row_table + (uint64_t)p * kNU;
  const double* __restrict__ rtab = a.row_table + (uint64_t)a.P * kNU + (uint64_t)p * slab_rows * kNRs;
  const double* __restrict__ ctab = a.row_table + (uint64_t)a.P * (kNU + slab_rows * kNRs) + (uint64_t)p * kNC * a.N1;
  // Which 256 columns does this workgroup own?  Workgroups are dealt to the 8 XCDs round-robin in launch order -- XCD = linear id
  // mod 8 = blockIdx.x mod 8 for the usual multiples of 8 column tiles -- and every XCD schedules only what it was dealt.  With
  // column tile = blockIdx.x a column tile that is structurally slow (D5: the column x1 = 0, where sin(theta) = 0 exactly sends its
  // wavefront through the IEEE loop in every row, 1.4 x the time, while the workgroup's other slots wait) lands on ONE XCD in every
  // grid row and parameter row: that XCD finishes last and the other seven idle (round 5: 12 % of D5's wave slots empty with no
  // workgroup held back by any resource, profiles/r05_experiments.txt sections 3 and 7).  Rotating the column tile with the grid
  // row and the parameter row gives every XCD every column tile equally often.  Same tiles, same values, another owner.
#ifndef INFLX_XCD_SPREAD
#define INFLX_XCD_SPREAD 1
#endif
#if INFLX_XCD_SPREAD
  const unsigned col_tile = (blockIdx.x + blockIdx.y + blockIdx.z) % gridDim.x;
#else
  const unsigned col_tile = blockIdx.x;
#endif
  const uint64_t col0 = (uint64_t)col_tile * kThreads;
  const uint64_t j = col0 + tid;
  const double x1 = inflx_coord(j, a.dx1, a.x1a);
  // relative to row_begin; stream_row0 = first slab row of this launch (grid.y is limited to 65535 tiles)
  const unsigned tile_rows = a.tile_rows;  // height of this launch's tiles (<= kTileRows; lower for small grids)
  const uint64_t row0 = (uint64_t)a.stream_row0 + (uint64_t)blockIdx.y * tile_rows;
  const uint64_t left = a.row_count - row0;
  const int nrows = left < (uint64_t)tile_rows ? (int)left : (int)tile_rows;
#ifdef INFLX_EXPERIMENT_INLINE_PROLOGUE  // (A/B experiment only: the round-1 prologue, every workgroup evaluates its own stage values)
#if INFLX_U_IN_LDS
  __shared__ __attribute__((aligned(16))) double U[kNU];
  if (tid == 0) inflx_stage_uniform(A, U);
  __syncthreads();
#else
  double U[kNU];
  inflx_stage_uniform(A, U);
#endif
  double C[kNC];
  inflx_stage_col(x1, A, U, C);
  if ((int)tid < nrows) inflx_stage_row(inflx_coord(a.row_begin + row0 + tid, a.dx0, a.x0a), A, U, Rs[tid]);
  __syncthreads();
  (void)utab, (void)rtab, (void)ctab;
#else
#if INFLX_U_IN_LDS
  // wave-uniform values live in LDS: every use is a broadcast read and costs no long-lived VGPRs
  // (a uniform f64 cannot be an operand from SGPRs more than once per instruction, and there are up to hundreds)
  __shared__ __attribute__((aligned(16))) double U[kNU];
  for (unsigned k = tid; k < (unsigned)kNU; k += kThreads) U[k] = utab[k];
#else
  double U[kNU];
#pragma unroll
  for (int k = 0; k < kNU; ++k) U[k] = utab[k];  // uniform address: scalar loads
#endif
  double C[kNC];
#pragma unroll
  for (int k = 0; k < kNC; ++k) C[k] = j < a.N1 ? ctab[(uint64_t)k * a.N1 + j] : 0.0;
  {
    const double* __restrict__ src = rtab + (uint64_t)blockIdx.y * tile_rows * kNRs;
    double* dst = &Rs[0][0];
    for (unsigned i = tid; i < (unsigned)(nrows * kNRs); i += kThreads) dst[i] = src[i];
  }
  __syncthreads();
#endif

#if INFLX_DRAIN_LOADS_BEFORE_ROW_LOOP
  // Every vector-memory LOAD of this kernel is issued above (parameters, the column values); the row loop only stores.
  // gfx950 counts loads and stores in the same counter (vmcnt), and the compiler waits for a load at the first use of its
  // value -- inside the row loop, where "vmcnt(0)" also means "until every store of the previous grid row has been
  // acknowledged by memory": each wavefront then idles one write round trip per row.  Waiting here, once, leaves the loop
  // without any wait on the stores.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt and lgkmcnt untouched
#endif
  const bool in_range = j < a.N1;
  const uint64_t wave_col0 = col0 + (uint64_t)wave * kWave;
  // 16-byte units of this wavefront's 64-point block that lie inside the row (AoS, K = 6)
  const uint64_t cols_left = wave_col0 < a.N1 ? a.N1 - wave_col0 : 0;
  const unsigned wave_units = cols_left >= kWave ? 3u * kWave : 3u * (unsigned)cols_left;
  const unsigned wave_unit0 = wave * (3u * kWave);  // first unit of the wavefront's block within the workgroup's part of a row

  // what happens to the K values of one grid row: summary, then the store in the requested layout
  auto emit = [&](const double (&o)[K], const uint64_t row) {
    if constexpr (STATS) {
      if (in_range) stat_add(acc, o, 1);
    }
    if constexpr (!STORE) return;

    // Every address below is "wave-uniform row base + 32-bit offset of the thread": the stores then take the base from
    // scalar registers and the compiler neither keeps 64-bit per-lane pointers across the row loop nor advances them
    // every row (it kept seven -- six planes and the AoS block -- when the addresses were formed per lane: 14 vector
    // registers and 7 64-bit additions per grid row, whichever layout ran).
    if (a.layout == INFLX_LAYOUT_SOA || K == 1) {
      if (in_range) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          uint64_t off = (((uint64_t)p * K + k) * a.row_count + row) * a.N1 + col0;
          INFLX_KEEP_SCALAR(off);  // (see below: keeps the row base in scalar registers)
          store_scalar<OP>(a.out, off, tid, o[k]);
        }
      }
    } else if constexpr (K == 6) {
      // wave-private transpose: lane l holds point l's 6 values; after it lane l holds the
      // l-th, (64+l)-th and (128+l)-th 16-byte unit of the block's 3072 contiguous bytes
      double* tb = tbuf[wave];
      inflx_d2* mine = reinterpret_cast<inflx_d2*>(tb + lane * 6);
      mine[0] = inflx_d2{o[0], o[1]};
      mine[1] = inflx_d2{o[2], o[3]};
      mine[2] = inflx_d2{o[4], o[5]};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // The row offset is made opaque to the loop optimiser: strength reduction would turn "scalar base(row) + thread offset"
      // back into per-lane pointers that it advances every row (it did, in the hot loop only: the redo loop's rows are not an
      // arithmetic sequence).  (The offset, not the pointer: a pointer that went through an asm loses its address space.)
      uint64_t row_off = (((uint64_t)p * a.row_count + row) * a.N1 + col0) * 6;
      INFLX_KEEP_SCALAR(row_off);
      inflx_d2* dst = reinterpret_cast<inflx_d2*>(a.out + row_off);  // the workgroup's 12 KiB of this row
      const inflx_d2* units = reinterpret_cast<const inflx_d2*>(tb);
      // all three LDS reads are issued before the first (predicated) store: one LDS round trip per row
      // instead of three (left alone, the compiler sinks every read into its store's branch)
      inflx_d2 v[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        v[s] = units[s * kWave + lane];
        asm volatile("" : "+v"(v[s]));
      }
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const unsigned q = s * kWave + lane;
        if (q < wave_units) store_d2(dst + (wave_unit0 + q), v[s]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
      if (in_range) {
        uint64_t row_off = (((uint64_t)p * a.row_count + row) * a.N1 + col0) * K;
        INFLX_KEEP_SCALAR(row_off);
        double* dst = a.out + row_off;
#pragma unroll
        for (int k = 0; k < K; ++k) store_d1(dst + (tid * (unsigned)K + k), o[k]);
      }
    }
  };

  // Hot loop: the point stage that divides by hoisted reciprocals (models with enough such quotients) and the per-point
  // operation that divides without special-case handling (inflx_op_complete_analysis_quick).  A row in which some lane
  // met an irregular case (NaN or infinite operands, overflow, a zero or denormal quotient, a model value outside
  // [2^-100, 2^100]...) is only noted here and evaluated after the loop with IEEE divisions -- by the whole wavefront,
  // since the row is stored as a block -- so that the cold code costs the hot loop neither registers nor branches.
  // Irregular cases are usually structural: a numerator that is exactly zero in one grid column (sin 0, x1 = 0)
  // fails in EVERY row of the wavefront that owns the column.  Such a wavefront would do all its rows twice
  // and hold its workgroup's slot 2.2 times as long as its neighbours (measured: +10 % on the whole D5 sweep at
  // 4096 columns); after two consecutive irregular rows it therefore stops trying and leaves the remaining rows
  // to the IEEE loop directly.
  static_assert(kTileRows <= 64, "one bit per tile row");
  // (One workgroup per tile.  A workgroup that walks several vertically consecutive tiles -- column values, parameter-only values
  // and coefficients loaded once, R rows restaged behind a barrier, wave slots held across the hand-over -- was measured in round 5,
  // bit-identical results: D5 4096^2 x 32 14.06 ms with 1 tile, 14.44 / 14.84 / 15.83 with 2 / 4 / 8; EGNO x 32 13.10 / 13.05 /
  // 13.13 / 13.43; doc x 16 3.47 / 3.47 / 3.44 / 3.49 -- profiles/r05_experiments.txt section 4.)
#ifdef INFLX_EXPERIMENT_R_SCALAR
  // (A/B experiment, profiles/r06_experiments.txt: the row's R values straight from the stage table through the scalar cache -- the
  // address is wave-uniform and the table constant for the kernel's lifetime -- instead of LDS broadcasts into vector registers)
  typedef const double __attribute__((address_space(4))) inflx_cdouble;
  const double* __restrict__ rbase = rtab + (uint64_t)blockIdx.y * tile_rows * kNRs;
#define INFLX_ROW_VALUES(r) ((const double*)(inflx_cdouble*)(rbase + (uint64_t)(r) * kNRs))
#else
#define INFLX_ROW_VALUES(r) (Rs[r])
#endif
  uint64_t redo = 0;
  int streak = 0;  // consecutive irregular rows (wave-uniform)
  for (int r = 0; r < nrows; ++r) {  // (unrolling by 2 was measured: no gain, scripts/tile_tuning.py)
    if (streak >= 2) {
      redo |= ~uint64_t(0) << r;  // rows r .. 63; rows >= nrows are masked below
      break;
    }
    const uint64_t row = row0 + r;
    const double x0 = inflx_coord(a.row_begin + row, a.dx0, a.x0a);
    InflxModelValues mv;
    bool ok = true;
#if INFLX_HAS_QUICK_POINT
    inflx_stage_point_quick(x0, x1, A, U, INFLX_ROW_VALUES(r), C, mv, ok);
#else
    inflx_stage_point(x0, x1, A, U, INFLX_ROW_VALUES(r), C, mv);
#endif
    fill_v01<OP>(mv, x0, x1, A);
    double o[K];
    ok = apply_op_quick<OP, kTable>(mv, o, a.accuracy, kc) && ok;
    // (lanes past N1 hold zeros for their column values and fail every acceptance test: they do not vote)
    if (__builtin_amdgcn_ballot_w64(!ok && in_range) != 0) {  // wave-uniform
      redo |= uint64_t(1) << r;
      ++streak;
      continue;
    }
    streak = 0;
    emit(o, row);
  }
  if (nrows < 64) redo &= (uint64_t(1) << nrows) - 1;
  while (redo != 0) {
    const int r = __builtin_ctzll(redo);
    redo &= redo - 1;
    const uint64_t row = row0 + r;
    const double x0 = inflx_coord(a.row_begin + row, a.dx0, a.x0a);
    InflxModelValues mv;
#if INFLX_HAS_QUICK_POINT
    inflx_stage_point_ieee(x0, x1, A, U, INFLX_ROW_VALUES(r), C, mv);
#else
    inflx_stage_point(x0, x1, A, U, INFLX_ROW_VALUES(r), C, mv);
#endif
    fill_v01<OP>(mv, x0, x1, A);
    double o[K];
    apply_op<OP, kTable>(mv, o, a.accuracy, kc);
    emit(o, row);
  }
#undef INFLX_ROW_VALUES
  if constexpr (STATS) stat_flush(acc, a.stats);
}

// ================================================================================================
// row kernels: no model value depends on x[1]  ->  one evaluation per grid row, broadcast along it
// ================================================================================================
// one lane evaluates everything a grid row needs (U, R and P stages + the per-point operation)
template <int OP>
__device__ __forceinline__ void eval_row(const InflxSweepArgs& a, unsigned p, uint64_t row, double* o) {
  double A[kNP];
  load_params(a.params, p, A);
  double U[kNU], R[kNR], C[kNC];
  inflx_stage_uniform(A, U);
  const double x0 = inflx_coord(a.row_begin + row, a.dx0, a.x0a);
  inflx_stage_row(x0, A, U, R);
  // by construction of the row kernels nothing below reads x1 or C
  InflxModelValues mv;
  inflx_stage_point(x0, a.x1a, A, U, R, C, mv);
  fill_v01<OP>(mv, x0, a.x1a, A);
  apply_op<OP>(mv, o, a.accuracy);
}

// Row-broadcast path for the AoS result of the 6-value operation, two launches:
//
//  1. inflx_sweep_rowvals_*: one lane per grid row evaluates the row's six values into row_table
//     (8192 rows = 32 workgroups; latency-bound, a few microseconds).
//  2. inflx_sweep_rowstream6: the output is one contiguous stream of 16-byte units in which every grid
//     row is N1 copies of the same 48 bytes, i.e. unit u of a row holds value pair (u mod 3).  Every
//     THREAD issues exactly ONE 16-byte store and every 256-thread workgroup writes 4 KiB of contiguous
//     memory; workgroups are dispatched in address order, so the resident ones always cover one compact,
//     advancing window of HBM.  Measured on MI355X for the 3.2 GB result (scripts/micro/store_bw.hip):
//     this shape 6.8-6.9 TB/s, hipMemsetAsync 6.5, several stores per thread 5.2-5.9 (one wavefront per
//     row, grid-stride chunks, 8-96 KiB per workgroup), non-temporal pairs of adjacent stores 2.1.
//     The row's values arrive by scalar loads (the table address is workgroup-uniform).
template <int OP, bool STATS = false>
__device__ __forceinline__ void sweep_rowvals(const InflxSweepArgs& a) {
  // one wavefront per workgroup (launched with 64 threads): 64 grid rows, spread over as many CUs as
  // possible because the evaluation is a ~750-instruction dependent chain per lane (latency-bound)
  constexpr int K = OpWidth<OP>::K;
  __shared__ __attribute__((aligned(16))) double vals[kWave][8];
  const unsigned lane = threadIdx.x;
  const uint64_t row0 = (uint64_t)blockIdx.x * kWave;
  const unsigned p = blockIdx.y;
  double o[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
  if (row0 + lane < a.row_count) eval_row<OP>(a, p, row0 + lane, o);
  if constexpr (STATS) {
    // every point of the row has these six values: the row counts N1 times
    StatAcc acc;
    stat_init(acc);
    if (row0 + lane < a.row_count) stat_add(acc, o, a.N1);
    stat_flush(acc, a.stats);
    if (a.row_table == nullptr) return;  // summary-only sweep: nothing to broadcast
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) vals[lane][k] = o[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // the 64 rows' table entries ([row][replica][8 doubles], replicas a power of two) are one contiguous
  // region: write it with coalesced 16-byte stores instead of 2-KiB-strided ones
  const uint64_t left = a.row_count - row0;
  const unsigned nrows = left < (uint64_t)kWave ? (unsigned)left : (unsigned)kWave;
  const unsigned shift = 31 - __builtin_clz(a.table_replicas * 4);  // log2(units per row)
  const unsigned units = nrows << shift;
  inflx_d2* dst = reinterpret_cast<inflx_d2*>(a.row_table + ((uint64_t)p * a.row_count + row0) * a.table_replicas * 8);
  for (unsigned q = lane; q < units; q += kWave) {
    const unsigned r = q >> shift, part = q & 3;
    dst[q] = *reinterpret_cast<const inflx_d2*>(&vals[r][2 * part]);
  }
  (void)K;
}

__device__ __forceinline__ void sweep_rowstream6(const InflxSweepArgs& a) {
  // grid: x = 4 KiB piece within the grid row, y = grid row (relative to stream_row0), z = parameter row;
  // no index arithmetic beyond multiply-add -- with one store per thread even a 64-bit division per
  // workgroup (flat grid -> (row, piece)) costs 30 % of the bandwidth.
  // (An XCD-aware remap -- every XCD takes one whole grid row of each group of 8, so that its L2 write-combines 384 KiB
  // runs instead of 4 KiB pieces 32 KiB apart -- gains 3 % in the pure-store microbenchmark, scripts/micro/store_bw.hip
  // variant X, but LOSES 1.7 % here, A/B in one session: 0.472 against 0.464 ms per sweep.  The identity map stays.)
  const uint64_t units_row = 3 * a.N1;
  const unsigned k = blockIdx.x;
  const uint64_t row = (uint64_t)a.stream_row0 + blockIdx.y;
  const unsigned p = blockIdx.z;
  const uint64_t slab_row = (uint64_t)p * a.row_count + row;
  // table_replicas is a power of two (the host guarantees it): a mask, not the 20-instruction modulo
  const double* __restrict__ t = a.row_table + (slab_row * a.table_replicas + (k & (a.table_replicas - 1))) * 8;
  const uint64_t u = (uint64_t)k * kThreads + threadIdx.x;
  // u mod 3 == (k + tid) mod 3 because 256 == 1 (mod 3)
  const unsigned phase = (k % 3 + threadIdx.x) % 3;
  // the six values are workgroup-uniform: fetch them with scalar loads.  (Left to itself the compiler
  // selects the *address* per lane and issues divergent vector loads inside branches; the empty asm
  // pins the values in SGPRs so that the selection below is three v_cndmask pairs.)
  double t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
  asm volatile("" : "+s"(t0), "+s"(t1), "+s"(t2), "+s"(t3), "+s"(t4), "+s"(t5));
  const inflx_d2 v = {phase == 0 ? t0 : (phase == 1 ? t2 : t4), phase == 0 ? t1 : (phase == 1 ? t3 : t5)};
  // non-temporal on purpose: a plain store stream of 3.2 GB sweeps the row table out of the Infinity
  // Cache, every table fetch then goes to HBM and the stream drops to 5.0 TB/s; with nt stores the
  // table stays cached and the stream runs at 6.7-6.8 TB/s (scripts/micro/store_bw.hip, variants G/H/P)
  if (u < units_row) stream_store_d2(reinterpret_cast<inflx_d2*>(a.out + slab_row * a.N1 * 6 + 2 * u), v);
}
static_assert(kThreads % 3 == 1, "the phase rule of sweep_rowstream6 needs kThreads == 1 (mod 3)");

// The same store stream for results made of planes ([P][K][rows][N1]: the SoA layout, and every
// single-value sweep with K = 1), N1 even: a plane row is N1/2 copies of the 16-byte unit (v, v).
// grid = (pieces per plane row, rows, P*K).
__device__ __forceinline__ void sweep_rowstream_planes(const InflxSweepArgs& a) {
  const uint64_t units_row = a.N1 / 2;
  const unsigned piece = blockIdx.x;
  const uint64_t row = (uint64_t)a.stream_row0 + blockIdx.y;
  const unsigned p = blockIdx.z / a.stream_planes;
  const unsigned k = blockIdx.z - p * a.stream_planes;
  const uint64_t slab_row = (uint64_t)p * a.row_count + row;
  const double* __restrict__ t = a.row_table + (slab_row * a.table_replicas + (piece & (a.table_replicas - 1))) * 8;
  double v = t[k];
  asm volatile("" : "+s"(v));
  const uint64_t u = (uint64_t)piece * kThreads + threadIdx.x;
  double* dst = a.out + (((uint64_t)p * a.stream_planes + k) * a.row_count + row) * a.N1;
  if (u < units_row) stream_store_d2(reinterpret_cast<inflx_d2*>(dst + 2 * u), inflx_d2{v, v});
}

template <int OP>
__device__ __forceinline__ void sweep_rows(const InflxSweepArgs& a) {
  constexpr int K = OpWidth<OP>::K;
  // every result shape except the 6-value AoS one (which takes rowvals + rowstream6): 8-byte stores,
  // one wavefront per grid row
  __shared__ double vals[kRowsPerBlock][8];
  const unsigned tid = threadIdx.x;
  const unsigned lane = tid & (kWave - 1);
  const unsigned wave = tid / kWave;
  const unsigned p = blockIdx.y;
  const uint64_t group = blockIdx.x / a.col_chunks;
  const unsigned chunk = blockIdx.x % a.col_chunks;
  const uint64_t row0 = group * kRowsPerBlock;

  if (tid < kRowsPerBlock && row0 + tid < a.row_count) {
    double o[K];
    eval_row<OP>(a, p, row0 + tid, o);
#pragma unroll
    for (int k = 0; k < K; ++k) vals[tid][k] = o[k];
  }
  __syncthreads();

  // waves stride over the block's rows (kRowsPerBlock may exceed the 4 waves of a workgroup)
  for (unsigned rr = wave; rr < (unsigned)kRowsPerBlock; rr += kThreads / kWave) {
    const uint64_t row = row0 + rr;
    if (row >= a.row_count) break;
    double v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = vals[rr][k];
    const uint64_t per_chunk = ((a.N1 + a.col_chunks - 1) / a.col_chunks + kWave - 1) / kWave * kWave;
    const uint64_t j_begin = (uint64_t)chunk * per_chunk;
    const uint64_t j_end = j_begin + per_chunk < a.N1 ? j_begin + per_chunk : a.N1;
    for (uint64_t j = j_begin + lane; j < j_end; j += kWave) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const uint64_t off = a.layout == INFLX_LAYOUT_SOA ? (((uint64_t)p * K + k) * a.row_count + row) * a.N1 + j
                                                          : (((uint64_t)p * a.row_count + row) * a.N1 + j) * K + k;
        store_scalar<OP>(a.out, off, 0u, v[k]);
      }
    }
  }
}

// ================================================================================================
// column kernels: no model value depends on x[0]  ->  one evaluation per grid column; every grid row is the
// same image, which a store stream copies row by row (the mirror image of the row-broadcast path)
// ================================================================================================
template <int OP, bool STATS = false>
__device__ __forceinline__ void sweep_colvals(const InflxSweepArgs& a) {
  constexpr int K = OpWidth<OP>::K;
  const uint64_t j = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  const unsigned p = blockIdx.y;
  const bool in_range = j < a.N1;
  double o[K];
#pragma unroll
  for (int k = 0; k < K; ++k) o[k] = 0.0;
  if (in_range) {
    double A[kNP];
    load_params(a.params, p, A);
    double U[kNU], R[kNR], C[kNC];
    inflx_stage_uniform(A, U);
    const double x0 = inflx_coord(a.row_begin, a.dx0, a.x0a);  // any row: by construction nothing below depends on it
    const double x1 = inflx_coord(j, a.dx1, a.x1a);
    inflx_stage_row(x0, A, U, R);
    inflx_stage_col(x1, A, U, C);
    InflxModelValues mv;
    inflx_stage_point(x0, x1, A, U, R, C, mv);
    fill_v01<OP>(mv, x0, x1, A);
    apply_op<OP>(mv, o, a.accuracy);
  }
  if constexpr (STATS) {
    // every grid row of this column has these six values: the column counts row_count times
    StatAcc acc;
    stat_init(acc);
    if (in_range) stat_add(acc, o, a.row_count);
    stat_flush(acc, a.stats);
    if (a.row_table == nullptr) return;  // summary-only sweep
  }
  if (!in_range) return;
  if (a.layout == INFLX_LAYOUT_SOA || K == 1) {
#pragma unroll
    for (int k = 0; k < K; ++k) a.row_table[((uint64_t)p * K + k) * a.N1 + j] = o[k];
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) a.row_table[((uint64_t)p * a.N1 + j) * K + k] = o[k];
  }
}

// grid: x = 4 KiB piece of the row image, y = grid row (relative to stream_row0), z = image.  One 16-byte load
// (the image stays in L2: at most a few hundred KiB, read once per grid row) and one 16-byte non-temporal store per
// thread -- the shape of inflx_sweep_rowstream6.
__device__ __forceinline__ void sweep_colstream(const InflxSweepArgs& a) {
  const uint64_t u = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  const uint64_t row = (uint64_t)a.stream_row0 + blockIdx.y;
  const uint64_t z = blockIdx.z;
  if (u < a.stream_units) {
    const inflx_d2 v = reinterpret_cast<const inflx_d2*>(a.row_table)[z * a.stream_units + u];
    stream_store_d2(reinterpret_cast<inflx_d2*>(a.out) + (z * a.row_count + row) * a.stream_units + u, v);
  }
}

// ================================================================================================
// on-trajectory kernels: explicit (n,2) point list (src/anguelova.rs:633-977)
// ================================================================================================
template <int OP>
__device__ __forceinline__ void sweep_trajectory(const InflxTrajectoryArgs& a) {
  constexpr int K = OpWidth<OP>::K;
  const uint64_t idx = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  const unsigned p = blockIdx.y;
  if (idx >= a.n) return;
  double A[kNP];
  load_params(a.params, p, A);
  double U[kNU], R[kNR], C[kNC];
  const double x0 = a.points[2 * idx], x1 = a.points[2 * idx + 1];
  inflx_stage_uniform(A, U);
  inflx_stage_row(x0, A, U, R);
  inflx_stage_col(x1, A, U, C);
  InflxModelValues mv;
  inflx_stage_point(x0, x1, A, U, R, C, mv);
  fill_v01<OP>(mv, x0, x1, A);
  double o[K];
  apply_op<OP>(mv, o, a.accuracy);
#pragma unroll
  for (int k = 0; k < K; ++k) store_scalar<OP>(a.out, ((uint64_t)p * a.n + idx) * K + k, 0u, o[k]);
}

// ================================================================================================
// stage tables of the tile kernels: U per parameter row, R per grid row, C per grid column
// ================================================================================================
// grid: x = ceil(slab rows / 256) row blocks followed by ceil(N1 / 256) column blocks, y = parameter row.  Table layout
// (doubles) behind a.row_table:
//   U[P][kNU]  |  R[P][slab_rows][kNRs]  |  C[P][kNC][N1]      (kNRs = kNR rounded up to even)
// (R row-major so that a tile's 32 rows are one contiguous block, C value-major so that the threads of a tile read
// every value coalesced).  The slab is rows [stream_row0, stream_row0 + stream_units) relative to row_begin.
// The kernel is a latency chain, not a throughput problem: U (one lane, dependent instructions), then a row's or a column's values
// (one lane each).  Rows and columns therefore live in DIFFERENT workgroups -- the chain a lone call waits for in front of its tile
// kernel is U + max(R, C), not U + R + C as it was while thread i evaluated row i and then column i.
#if INFLX_HAS_GROUP(INFLX_GROUP_CORE)
extern "C" __global__ __launch_bounds__(kThreads) void inflx_stage_tables(const InflxSweepArgs a) {
  const unsigned tid = threadIdx.x;
  const unsigned p = blockIdx.y;
  const uint64_t slab_rows = a.stream_units;
  const unsigned row_blocks = (unsigned)((slab_rows + kThreads - 1) / kThreads);
  const bool rows = blockIdx.x < row_blocks;
  const uint64_t idx = (uint64_t)(rows ? blockIdx.x : blockIdx.x - row_blocks) * kThreads + tid;
  double* utab = a.row_table + (uint64_t)p * kNU;
  double* rtab = a.row_table + (uint64_t)a.P * kNU + (uint64_t)p * slab_rows * kNRs;
  double* ctab = a.row_table + (uint64_t)a.P * (kNU + slab_rows * kNRs) + (uint64_t)p * kNC * a.N1;
  double A[kNP];
  load_params(a.params, p, A);
#if INFLX_U_IN_LDS
  __shared__ __attribute__((aligned(16))) double U[kNU];
  if (tid == 0) inflx_stage_uniform(A, U);
  __syncthreads();
  if (blockIdx.x == 0)
    for (unsigned k = tid; k < (unsigned)kNU; k += kThreads) utab[k] = U[k];
#else
  double U[kNU];
#pragma unroll
  for (int k = 0; k < kNU; ++k) U[k] = 0.0;
  inflx_stage_uniform(A, U);
  if (blockIdx.x == 0 && tid == 0) {
#pragma unroll
    for (int k = 0; k < kNU; ++k) utab[k] = U[k];
  }
#endif
  if (rows) {
    if (idx < slab_rows) {
      const double x0 = inflx_coord(a.row_begin + a.stream_row0 + idx, a.dx0, a.x0a);
      inflx_stage_row(x0, A, U, rtab + idx * kNRs);
    }
  } else if (idx < a.N1) {
    const double x1 = inflx_coord(idx, a.dx1, a.x1a);
    double C[kNC];
#pragma unroll
    for (int k = 0; k < kNC; ++k) C[k] = 0.0;
    inflx_stage_col(x1, A, U, C);
#pragma unroll
    for (int k = 0; k < kNC; ++k) ctab[(uint64_t)k * a.N1 + idx] = C[k];
  }
}

#endif  // INFLX_GROUP_CORE

// ---- entry points (looked up by name with hipModuleGetFunction) --------------------------------
#define INFLX_DEFINE_KERNELS(NAME, OP)                                                                   \
  extern "C" __global__ __launch_bounds__(kThreads, INFLX_MIN_WAVES) void inflx_sweep_tile_##NAME(const InflxSweepArgs a) { \
    sweep_tile<OP>(a);                                                                                   \
  }                                                                                                      \
  extern "C" __global__ __launch_bounds__(kThreads) void inflx_sweep_rows_##NAME(const InflxSweepArgs a) { \
    if constexpr ((INFLX_OUT_MASK & 2) == 0) sweep_rows<OP>(a);                                          \
  }                                                                                                      \
  extern "C" __global__ __launch_bounds__(kThreads) void inflx_sweep_rowvals_##NAME(const InflxSweepArgs a) { \
    if constexpr ((INFLX_OUT_MASK & 2) == 0) sweep_rowvals<OP>(a);                                       \
  }                                                                                                      \
  extern "C" __global__ __launch_bounds__(kThreads) void inflx_sweep_colvals_##NAME(const InflxSweepArgs a) { \
    if constexpr ((INFLX_OUT_MASK & 3) == 2) sweep_colvals<OP>(a);                                       \
  }                                                                                                      \
  extern "C" __global__ __launch_bounds__(kThreads) void inflx_sweep_traj_##NAME(const InflxTrajectoryArgs a) { \
    sweep_trajectory<OP>(a);                                                                             \
  }

#if INFLX_HAS_GROUP(INFLX_GROUP_CORE)
// validate_basis_*: v, w1 and their inner products at n explicit points, 7 doubles per point
// (reference src/lib.rs:149-163 calls the C functions v, w1 and inner_prod per point)
extern "C" __global__ __launch_bounds__(kThreads) void inflx_basis_points(const InflxTrajectoryArgs a) {
  const uint64_t idx = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= a.n) return;
  double A[kNP];
  load_params(a.params, blockIdx.y, A);
  double o[7];
  inflx_basis_point(a.points[2 * idx], a.points[2 * idx + 1], A, o);
#pragma unroll
  for (int k = 0; k < 7; ++k) a.out[((uint64_t)blockIdx.y * a.n + idx) * 7 + k] = o[k];
}
// the store streams (no model code in them)
extern "C" __global__ __launch_bounds__(kThreads) void inflx_sweep_rowstream6(const InflxSweepArgs a) { sweep_rowstream6(a); }
extern "C" __global__ __launch_bounds__(kThreads) void inflx_sweep_rowstream_planes(const InflxSweepArgs a) {
  sweep_rowstream_planes(a);
}
extern "C" __global__ __launch_bounds__(kThreads) void inflx_sweep_colstream(const InflxSweepArgs a) { sweep_colstream(a); }
INFLX_DEFINE_KERNELS(complete, INFLX_OP_COMPLETE)
#endif  // INFLX_GROUP_CORE

#if INFLX_HAS_GROUP(INFLX_GROUP_VALUES)
// ops::* of src/anguelova.rs:99-171 applied to GIVEN model values -- no model evaluation: a.points holds n records
// (V, v00, v10, v11, |dV|^2), a.out receives n records of 9 doubles: [0..5] complete_analysis, [6] consistency_only,
// [7] consistency_rapidturn_only, [8] epsilon_v_only.  a.reserved = 0: complete_analysis exactly as the sweep kernels
// evaluate it (divisions without special-case handling; the IEEE spelling for a wavefront in which some lane does not
// qualify); 1: the IEEE spelling only.  This is how the per-point operations are pinned against the oracle on arbitrary
// inputs -- specials, zeros, denormals, random bit patterns -- that no model produces on demand.
extern "C" __global__ __launch_bounds__(kThreads) void inflx_ops_on_values(const InflxTrajectoryArgs a) {
  const uint64_t idx = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  const bool live = idx < a.n;
  InflxModelValues mv;
  const double* rec = a.points + 5 * (live ? idx : 0);
  mv.V = rec[0], mv.v00 = rec[1], mv.v10 = rec[2], mv.v11 = rec[3], mv.g = rec[4];
  mv.b0 = mv.b1 = 0.0;
  double o[6];
  bool ok = false;
  if (a.reserved == 0) ok = apply_op_quick<INFLX_OP_COMPLETE>(mv, o);
  if (__builtin_amdgcn_ballot_w64(!ok && live) != 0) apply_op<INFLX_OP_COMPLETE>(mv, o);
  // consistency_only the way the tile kernel evaluates it: the quick spelling, the IEEE one for the whole wavefront otherwise
  double cons;
  bool ok_cons = false;
  if (a.reserved == 0) ok_cons = apply_op_quick<INFLX_OP_CONSISTENCY>(mv, &cons);
  if (__builtin_amdgcn_ballot_w64(!ok_cons && live) != 0) cons = inflx_op_consistency_only(mv);
  if (!live) return;
  double* dst = a.out + idx * 9;
#pragma unroll
  for (int k = 0; k < 6; ++k) dst[k] = o[k];
  dst[6] = cons;
  dst[7] = inflx_op_consistency_rapidturn_only(mv);
  dst[8] = inflx_op_epsilon_v_only(mv);
}
#endif  // INFLX_GROUP_VALUES

#if INFLX_HAS_GROUP(INFLX_GROUP_STATS)
// complete_analysis with the running summary; the *_nostore variant evaluates and reduces only
extern "C" __global__ __launch_bounds__(kThreads, INFLX_MIN_WAVES) void inflx_sweep_tile_complete_stats(const InflxSweepArgs a) {
  sweep_tile<INFLX_OP_COMPLETE, true, true>(a);
}
extern "C" __global__ __launch_bounds__(kThreads, INFLX_MIN_WAVES) void inflx_sweep_tile_complete_stats_nostore(const InflxSweepArgs a) {
  sweep_tile<INFLX_OP_COMPLETE, true, false>(a);
}
extern "C" __global__ __launch_bounds__(kThreads) void inflx_sweep_rowvals_complete_stats(const InflxSweepArgs a) {
  if constexpr ((INFLX_OUT_MASK & 2) == 0) sweep_rowvals<INFLX_OP_COMPLETE, true>(a);
}
extern "C" __global__ __launch_bounds__(kThreads) void inflx_sweep_colvals_complete_stats(const InflxSweepArgs a) {
  if constexpr ((INFLX_OUT_MASK & 3) == 2) sweep_colvals<INFLX_OP_COMPLETE, true>(a);
}
#endif  // INFLX_GROUP_STATS

#if INFLX_HAS_GROUP(INFLX_GROUP_OF_OP(1))  // INFLX_OP_CONSISTENCY (an enumerator: the preprocessor needs the number)
INFLX_DEFINE_KERNELS(consistency, INFLX_OP_CONSISTENCY)
#endif
#if INFLX_HAS_GROUP(INFLX_GROUP_OF_OP(2))  // INFLX_OP_RAPIDTURN (an enumerator: the preprocessor needs the number)
INFLX_DEFINE_KERNELS(rapidturn, INFLX_OP_RAPIDTURN)
#endif
#if INFLX_HAS_GROUP(INFLX_GROUP_OF_OP(3))  // INFLX_OP_EPSILON_V (an enumerator: the preprocessor needs the number)
INFLX_DEFINE_KERNELS(epsilon_v, INFLX_OP_EPSILON_V)
#endif
#if INFLX_HAS_GROUP(INFLX_GROUP_OF_OP(4))  // INFLX_OP_RAW (an enumerator: the preprocessor needs the number)
INFLX_DEFINE_KERNELS(raw, INFLX_OP_RAW)
#endif
#if INFLX_HAS_GROUP(INFLX_GROUP_OF_OP(5))  // INFLX_OP_QDIF (an enumerator: the preprocessor needs the number)
INFLX_DEFINE_KERNELS(qdif, INFLX_OP_QDIF)
#endif
#if INFLX_HAS_GROUP(INFLX_GROUP_OF_OP(6))  // INFLX_OP_HESSE (an enumerator: the preprocessor needs the number)
INFLX_DEFINE_KERNELS(hesse, INFLX_OP_HESSE)
#endif

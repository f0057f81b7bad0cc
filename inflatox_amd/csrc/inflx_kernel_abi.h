// Contract between the per-model code object (device side) and libinflx_hip.so (host side).
// Plain C types only; shared by inflx_sweep_kernels.hip and inflx_hip.cpp.
#pragma once
#include <stdint.h>

#define INFLX_KERNEL_ABI 19u

// which per-point operation a sweep kernel applies (reference src/anguelova.rs `mod ops`)
enum InflxOp {
  INFLX_OP_COMPLETE = 0,     // ops::complete_analysis          6 f64 per point
  INFLX_OP_CONSISTENCY = 1,  // ops::consistency_only           1 f64
  INFLX_OP_RAPIDTURN = 2,    // ops::consistency_rapidturn_only 1 f64
  INFLX_OP_EPSILON_V = 3,    // ops::epsilon_v_only             1 f64
  INFLX_OP_RAW = 4,          // V, v00, v10, v11, |dV|^2        5 f64 (diagnostic; pins the model functions)
  INFLX_OP_QDIF = 5,         // ops::flag_quantum_diff          1 byte (bool) per point
  INFLX_OP_HESSE = 6,        // v00, v01, v10, v11              4 f64: the projected Hesse matrix row-major, with the reference's
                             //                                 OWN v01 (Hesse2D loads fns=[v00,v01,v10,v11], hesse_bindings.rs:202-210;
                             //                                 `hesse` returns all four, src/lib.rs:384-420)
  INFLX_OP_COUNT = 7
};

// Kernel groups of a code object (its INFLX_GROUPS global): a model artefact is the CORE object -- everything complete_analysis
// needs -- plus, built and loaded on first use, the group of every other operation.  A full artefact carries all of them.
#define INFLX_GROUP_CORE 1u    // inflx_stage_tables, inflx_sweep_{tile,rows,rowvals,colvals,traj}_complete, the store streams, inflx_basis_points
#define INFLX_GROUP_STATS 2u   // the fused-summary kernels (*_complete_stats[_nostore])
#define INFLX_GROUP_VALUES 4u  // inflx_ops_on_values
#define INFLX_GROUP_OF_OP(op) ((op) == 0 ? INFLX_GROUP_CORE : (8u << ((op)-1)))  // consistency 8, rapidturn 16, epsilon_v 32, raw 64, qdif 128, hesse 256
#define INFLX_GROUP_ALL 511u

// output memory layout for multi-value operations
enum InflxLayout {
  INFLX_LAYOUT_AOS = 0,  // [P][rows][N1][K]   the reference's (N0,N1,6) array per parameter row
  INFLX_LAYOUT_SOA = 1   // [P][K][rows][N1]   K contiguous planes per parameter row
};

// Launch arguments of every grid-sweep kernel (passed by value).
//
// A launch evaluates rows [row_begin, row_begin + row_count) of the full N0 x N1 grid for P
// parameter rows.  Row i, column j maps to the field-space point
//   x0 = (double)i * dx0 + x0a,   x1 = (double)j * dx1 + x1a            (src/anguelova.rs:531-533)
// Output element (p, i, j, k), AOS:  out[((p*row_count + (i-row_begin))*N1 + j)*K + k]
//                               SOA:  out[((p*K + k)*row_count + (i-row_begin))*N1 + j]
struct InflxSweepArgs {
  double* out;
  const double* params;  // [P][N_PARAMETERS], device memory
  double x0a, dx0, x1a, dx1;
  uint64_t N1;
  uint64_t row_begin;
  uint64_t row_count;
  uint32_t P;
  uint32_t layout;      // InflxLayout
  uint32_t col_chunks;  // row kernels: number of column chunks a row is split into
  uint32_t stream_row0; // slab row that blockIdx.y == 0 works on (store-stream and tile kernels: grid.y <= 65535 per launch)
  // row-broadcast path: per-row results, [P][row_count][table_replicas][8] doubles (first K of 8 used).
  // Every row's 64-byte entry is stored table_replicas times, on cache lines of its own, because the
  // ~400 wavefronts that stream one grid row all fetch it with scalar loads at the same moment: served
  // from a single line that fetch throttles the store stream to 5.1 TB/s, from 32 replicas it runs at 6.8.
  double* row_table;
  uint32_t table_replicas;
  uint32_t stream_planes;  // inflx_sweep_rowstream_planes: K, the number of result planes per parameter row
  double accuracy;  // INFLX_OP_QDIF: threshold of ops::flag_quantum_diff
  // *_stats kernels: running summary of the six outputs over everything the launch evaluates --
  // [0..5] NaN-ignoring minimum, [6..11] NaN-ignoring maximum (f64), [12..17] number of non-NaN values (u64)
  double* stats;
  // column-broadcast path (no model value depends on x[0]): row_table holds the image of ONE output row per
  // image index z -- AOS: z = parameter row, stream_units = K*N1/2; planes: z = p*K + k, stream_units = N1/2 --
  // and inflx_sweep_colstream copies image z into every grid row: out unit ((z*row_count + row)*stream_units + u)
  uint64_t stream_units;  // 16-byte units per output row
  // tile kernels (some value depends on x[1]): row_table holds the stage tables inflx_stage_tables wrote for this launch,
  //   U[P][max(NU,1)] | R[P][slab_rows][NRs] | C[P][max(NC,1)][N1]   (doubles; NRs = max(NR,1) rounded up to even),
  // the slab being grid rows [stream_row0, stream_row0 + stream_units) relative to row_begin (stream_units = slab_rows here)
  // tile kernels: grid rows per workgroup tile of THIS launch, 1 ... InflxKernelInfo::tile_rows.  Large grids use the full
  // height; a grid with fewer full-height tiles than the chip has wavefront slots is cut into lower tiles, so that a
  // 256 x 256 or 1000 x 1000 sweep is not eight (128) workgroups walking 32 rows each one after the other
  uint32_t tile_rows;
  uint32_t reserved0;
};

// Launch arguments of the on-trajectory kernels: n explicit points (x0, x1) per launch
// (src/anguelova.rs:633-977); out[(p*n + idx)*K + k].
struct InflxTrajectoryArgs {
  double* out;
  const double* params;
  const double* points;  // [n][2]
  uint64_t n;
  uint32_t P;
  uint32_t reserved;  // inflx_ops_on_values: 1 = IEEE spelling of the divisions only (points = n records of 5 model values there)
  double accuracy;
};

// Read by the host from the code object's INFLX_KERNEL_INFO global after loading it.
struct InflxKernelInfo {
  uint32_t kernel_abi;  // INFLX_KERNEL_ABI
  uint32_t n_uniform;   // exported U-stage values
  uint32_t n_row;       // exported R-stage values (doubles of LDS per tile row)
  uint32_t n_col;       // exported C-stage values (registers per thread)
  uint32_t out_mask;    // bit0: some model value depends on x[0]; bit1: on x[1]
  uint32_t tile_rows;   // rows per workgroup tile of the tile kernels
  uint32_t tile_cols;   // columns per workgroup tile (= threads per workgroup)
  uint32_t rows_per_block;  // rows per workgroup of the row-broadcast kernels (SoA / single-value results)
  uint32_t row_chunk_units;  // 16-byte units each workgroup of inflx_sweep_rowstream6 writes (= its thread count)
  uint32_t reserved;
};

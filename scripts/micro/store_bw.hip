// Microbenchmark (MI355X): how fast can 3.2 GB be written with 16-byte stores, and which access
// shape gets closest to the HBM write peak?  Variants mirror candidate structures of the row kernel.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT> __device__ __forceinline__ void st(double* p, d2 v) {
  if (NT) __builtin_nontemporal_store(v, (d2*)p); else *(d2*)p = v;
}

// A: one wave per row (row = units_per_row 16-B units), 4 rows per 256-thread block  (current kernel)
template <bool NT> __global__ __launch_bounds__(256) void wave_per_row(double* out, uint64_t rows, uint64_t upr) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t row = (uint64_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  d2 v = {1.0 + lane, 2.0};
  double* dst = out + row * upr * 2;
  for (uint64_t u = lane; u < upr; u += 192) {
    st<NT>(dst + 2 * u, v);
    if (u + 64 < upr) st<NT>(dst + 2 * (u + 64), v);
    if (u + 128 < upr) st<NT>(dst + 2 * (u + 128), v);
  }
}

// B: flat grid-stride over chunks of CH units; block b takes chunks b, b+G, ...; the 4 waves of a block
// write one chunk together (each store instruction 1 KiB, a block iteration 4 KiB contiguous)
template <bool NT, int THREADS> __global__ __launch_bounds__(THREADS) void chunked(double* out, uint64_t total_units, uint64_t chunk_units) {
  const uint64_t nchunks = (total_units + chunk_units - 1) / chunk_units;
  d2 v = {1.0 + threadIdx.x, 2.0};
  for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const uint64_t b = c * chunk_units;
    const uint64_t e = b + chunk_units < total_units ? b + chunk_units : total_units;
    for (uint64_t u = b + threadIdx.x; u < e; u += THREADS) st<NT>(out + 2 * u, v);
  }
}

// C: like A but the 4 waves of a block share ONE row segment (block writes 4 KiB contiguous per iteration),
// block b owns row b (grid = rows)
template <bool NT> __global__ __launch_bounds__(256) void block_per_row(double* out, uint64_t rows, uint64_t upr) {
  const uint64_t row = blockIdx.x;
  d2 v = {1.0 + threadIdx.x, 2.0};
  double* dst = out + row * upr * 2;
  for (uint64_t u = threadIdx.x; u < upr; u += 256) st<NT>(dst + 2 * u, v);
}

// D: one chunk per block, no loop over chunks (grid = number of chunks); UNROLL stores per thread
template <bool NT, int PER_THREAD> __global__ __launch_bounds__(256) void one_chunk(double* out, uint64_t total_units, const double* table) {
  const uint64_t b = (uint64_t)blockIdx.x * (256 * PER_THREAD);
  // six values per block from a (cached) table, like the row kernel would read them
  const double* t = table + (blockIdx.x % 8192) * 6;
  d2 pr[3] = {{t[0], t[1]}, {t[2], t[3]}, {t[4], t[5]}};
#pragma unroll
  for (int i = 0; i < PER_THREAD; ++i) {
    const uint64_t u = b + threadIdx.x + (uint64_t)i * 256;
    if (u < total_units) st<NT>(out + 2 * u, pr[(threadIdx.x + i) % 3]);
  }
}

// E: one 32-byte contiguous span per thread (two adjacent 16-B stores), block = THREADS threads
template <bool NT, int THREADS> __global__ __launch_bounds__(THREADS) void two_adjacent(double* out, uint64_t total_units) {
  const uint64_t u = ((uint64_t)blockIdx.x * THREADS + threadIdx.x) * 2;
  d2 v = {1.0 + threadIdx.x, 2.0};
  if (u < total_units) st<NT>(out + 2 * u, v);
  if (u + 1 < total_units) st<NT>(out + 2 * (u + 1), v);
}
// F: one 16-B store per thread, block = THREADS threads
template <bool NT, int THREADS> __global__ __launch_bounds__(THREADS) void one_store(double* out, uint64_t total_units) {
  const uint64_t u = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
  d2 v = {1.0 + threadIdx.x, 2.0};
  if (u < total_units) st<NT>(out + 2 * u, v);
}

// Q (round 3): F with the store's cache-policy bits spelled explicitly (gfx950: sc0 / sc1 = coherence scope, nt = streaming hint)
template <int POLICY, int THREADS = 256, int PER_THREAD = 1> __global__ __launch_bounds__(THREADS) void one_store_policy(double* out, uint64_t total_units) {
  uint64_t u = (uint64_t)blockIdx.x * (THREADS * PER_THREAD) + threadIdx.x;
  if (PER_THREAD == 2) {  // two stores per thread, THREADS * 16 bytes apart: 2 * THREADS * 16 contiguous bytes per workgroup
    d2 w = {3.0 + threadIdx.x, 4.0};
    double* q = out + 2 * (u + THREADS);
    if (u + THREADS < total_units) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(q), "v"(w) : "memory");
  }
  d2 v = {1.0 + threadIdx.x, 2.0};
  if (u >= total_units) return;
  double* p = out + 2 * u;
  if (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
  if (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
  if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
  if (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
  if (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
  if (POLICY == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" : : "v"(p), "v"(v) : "memory");
  if (POLICY == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory");
  if (POLICY == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(v) : "memory");
}

// X: F with an XCD-aware remap of workgroup -> 4-KiB piece: workgroups are dispatched round-robin over the 8 XCDs
// (blockIdx % 8), so with run = G consecutive pieces per XCD each XCD's L2 write-combines contiguous runs of G * 4 KiB;
// G = nblocks/8 gives every XCD one contiguous eighth of the buffer, G = 1 is F itself
template <bool NT> __global__ __launch_bounds__(256) void one_store_xcd(double* out, uint64_t total_units, uint64_t run) {
  const uint64_t b = blockIdx.x, xcd = b & 7, k = b >> 3;          // k-th workgroup of this XCD
  const uint64_t piece = ((k / run) * 8 + xcd) * run + (k % run);  // super-chunk k/run, XCD's run inside it, position in the run
  const uint64_t u = piece * 256 + threadIdx.x;
  d2 v = {1.0 + threadIdx.x, 2.0};
  if (u < total_units) st<NT>(out + 2 * u, v);
}

// G: the product's inflx_sweep_rowstream6 logic (3-D grid, row table, phase select), parameterised
struct GArgs { double* out; const double* table; uint64_t N1; uint64_t row_count; uint32_t stream_row0; };
template <int MODE> __global__ __launch_bounds__(256) void rowstream_like(const GArgs a) {
  const uint64_t units_row = 3 * a.N1;
  const unsigned k = blockIdx.x;
  const uint64_t row = (uint64_t)a.stream_row0 + blockIdx.y;
  const unsigned p = blockIdx.z;
  const uint64_t slab_row = (uint64_t)p * a.row_count + row;
  const double* __restrict__ t = a.table + slab_row * 8;
  if (MODE == 3) t = a.table + ((slab_row * 96 + k) % 8192) * 8;           // a different line for every block
  if (MODE == 4) t = a.table + ((slab_row % 1024) * 8 + (k % 8)) * 8;      // 8 replicas per row
  if (MODE == 5) t = a.table + ((slab_row % 256) * 32 + (k % 32)) * 8;     // 32 replicas per row
  const uint64_t u = (uint64_t)k * 256 + threadIdx.x;
  const unsigned phase = (k % 3 + threadIdx.x) % 3;
  double t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
  if (MODE == 0 || MODE >= 3) asm volatile("" : "+s"(t0), "+s"(t1), "+s"(t2), "+s"(t3), "+s"(t4), "+s"(t5));
  d2 v = {phase == 0 ? t0 : (phase == 1 ? t2 : t4), phase == 0 ? t1 : (phase == 1 ? t3 : t5)};
  if (MODE == 2) v = d2{1.0 + threadIdx.x, 2.0};  // no table at all
  if (u < units_row) *(d2*)(a.out + slab_row * a.N1 * 6 + 2 * u) = v;
}

// H: cold full-size replicated table written by a pre-kernel, S contiguous 1-KiB stores per wave
__global__ void fill_table(double* table, uint64_t lines) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < lines) for (int k = 0; k < 8; ++k) table[i * 8 + k] = 1.0 + (i >> 5) + k;
}
// P: 1 store/thread, cold replicated table, optional nt stores, optional prefetch of the line that the
// block `dist` rows later (same piece index, hence the same XCD under round-robin dispatch) will need
template <bool NT, bool PREFETCH> __global__ __launch_bounds__(256) void rowstream_pf(const GArgs a, unsigned replicas, unsigned dist) {
  const uint64_t units_row = 3 * a.N1;
  const unsigned k = blockIdx.x;
  const uint64_t row = blockIdx.y;
  const double* __restrict__ t = a.table + (row * replicas + k % replicas) * 8;
  double t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
  asm volatile("" : "+s"(t0), "+s"(t1), "+s"(t2), "+s"(t3), "+s"(t4), "+s"(t5));
  (void)dist;  // (a fire-and-forget prefetch of a later row's line was tried here and faulted; removed)
  const uint64_t u = (uint64_t)k * 256 + threadIdx.x;
  const unsigned phase = (k % 3 + threadIdx.x) % 3;
  d2 v = {phase == 0 ? t0 : (phase == 1 ? t2 : t4), phase == 0 ? t1 : (phase == 1 ? t3 : t5)};
  if (u < units_row) st<NT>(a.out + row * a.N1 * 6 + 2 * u, v);
}

template <int S> __global__ __launch_bounds__(256) void rowstream_cold(const GArgs a, unsigned replicas) {
  // block covers 256*S units; wave w covers units [w*64*S, (w+1)*64*S) of it: S contiguous KiB per wave
  const uint64_t units_row = 3 * a.N1;
  const unsigned k = blockIdx.x;
  const uint64_t row = blockIdx.y;
  const double* __restrict__ t = a.table + (row * replicas + k % replicas) * 8;
  double t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
  asm volatile("" : "+s"(t0), "+s"(t1), "+s"(t2), "+s"(t3), "+s"(t4), "+s"(t5));
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t u0 = (uint64_t)k * 256 * S + (uint64_t)wave * 64 * S + lane;
#pragma unroll
  for (int i = 0; i < S; ++i) {
    const uint64_t u = u0 + 64 * i;
    const unsigned phase = (unsigned)(u % 3);
    d2 v = {phase == 0 ? t0 : (phase == 1 ? t2 : t4), phase == 0 ? t1 : (phase == 1 ? t3 : t5)};
    if (u < units_row) *(d2*)(a.out + row * a.N1 * 6 + 2 * u) = v;
  }
}

int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  const bool policy_only = argc > 1 && argv[1][0] == 'p';  // `store_bw.bin policy`: only the cache-policy variants Q
  const uint64_t N = 8192, upr = 3 * N, rows = N, total = rows * upr;
  const size_t bytes = total * 16;
  double* d; CK(hipMalloc(&d, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10; if (ms < best) best = ms;
    }
    printf("%-44s %7.3f ms  %7.1f GB/s\n", name, best, bytes / best / 1e6);
  };
  {
    const unsigned nb = (unsigned)((total + 255) / 256);
    const char* names[8] = {"(none)", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
    for (int round = 0; round < 2; ++round) {
      char nm[96];
#define RUN_Q(P) snprintf(nm, sizeof nm, "Q 1 store/thread 256thr, policy %s", names[P]); timeit(nm, [&] { one_store_policy<P><<<nb, 256>>>(d, total); });
      RUN_Q(0) RUN_Q(1) RUN_Q(2) RUN_Q(3) RUN_Q(4) RUN_Q(5) RUN_Q(6) RUN_Q(7)
    }
  }
  {
    char nm[96];
#define RUN_QT(T, PT) snprintf(nm, sizeof nm, "Q sc1 nt, %d threads, %d store(s)/thread (%d KiB/workgroup)", T, PT, T * PT * 16 / 1024); \
    timeit(nm, [&] { one_store_policy<6, T, PT><<<(unsigned)((total + (uint64_t)T * PT - 1) / ((uint64_t)T * PT)), T>>>(d, total); });
    for (int round = 0; round < 2; ++round) { RUN_QT(64, 1) RUN_QT(128, 1) RUN_QT(256, 1) RUN_QT(512, 1) RUN_QT(1024, 1) RUN_QT(256, 2) RUN_QT(128, 2) }
  }
  if (policy_only) return 0;
  timeit("A wave/row nt", [&] { wave_per_row<true><<<rows / 4, 256>>>(d, rows, upr); });
  timeit("A wave/row plain", [&] { wave_per_row<false><<<rows / 4, 256>>>(d, rows, upr); });
  timeit("C block/row nt", [&] { block_per_row<true><<<rows, 256>>>(d, rows, upr); });
  timeit("C block/row plain", [&] { block_per_row<false><<<rows, 256>>>(d, rows, upr); });
  for (int grid : {1024, 2048, 4096, 8192}) for (uint64_t ch : {256ull, 1024ull, 4096ull, 24576ull}) {
    char nm[96];
    snprintf(nm, sizeof nm, "B chunked nt grid=%d chunk=%lluKiB", grid, (unsigned long long)(ch * 16 / 1024));
    timeit(nm, [&] { chunked<true, 256><<<grid, 256>>>(d, total, ch); });
  }
  for (uint64_t ch : {1024ull, 4096ull}) {
    char nm[96];
    snprintf(nm, sizeof nm, "B chunked plain grid=2048 chunk=%lluKiB", (unsigned long long)(ch * 16 / 1024));
    timeit(nm, [&] { chunked<false, 256><<<2048, 256>>>(d, total, ch); });
    snprintf(nm, sizeof nm, "B chunked nt 512thr grid=1024 chunk=%lluKiB", (unsigned long long)(ch * 16 / 1024));
    timeit(nm, [&] { chunked<true, 512><<<1024, 512>>>(d, total, ch); });
    snprintf(nm, sizeof nm, "B chunked nt 1024thr grid=512 chunk=%lluKiB", (unsigned long long)(ch * 16 / 1024));
    timeit(nm, [&] { chunked<true, 1024><<<512, 1024>>>(d, total, ch); });
  }
  double* table; CK(hipMalloc(&table, 8192 * 8 * 8)); CK(hipMemset(table, 0, 8192 * 8 * 8));
#define RUN_D(NT, PT) { char nm[96]; snprintf(nm, sizeof nm, "D one chunk/block %s chunk=%dKiB", NT ? "nt" : "plain", 256 * PT * 16 / 1024); \
    const unsigned grid = (unsigned)((total + 256ull * PT - 1) / (256ull * PT)); \
    timeit(nm, [&] { one_chunk<NT, PT><<<grid, 256>>>(d, total, table); }); }
  RUN_D(true, 1) RUN_D(true, 2) RUN_D(true, 3) RUN_D(true, 6) RUN_D(true, 12) RUN_D(true, 24)
  RUN_D(false, 1) RUN_D(false, 2) RUN_D(false, 3) RUN_D(false, 6) RUN_D(false, 12) RUN_D(false, 24)
  {
    GArgs ga{d, table, N, rows, 0};
    timeit("G rowstream-like 3D grid (96,8192,1) sgpr table", [&] { rowstream_like<0><<<dim3(96, 8192, 1), 256>>>(ga); });
    timeit("G rowstream-like 3D grid (96,8192,1) plain table", [&] { rowstream_like<1><<<dim3(96, 8192, 1), 256>>>(ga); });
    timeit("G rowstream-like 3D grid (96,8192,1) no table", [&] { rowstream_like<2><<<dim3(96, 8192, 1), 256>>>(ga); });
    double* big; CK(hipMalloc(&big, 8192ull * 96 * 64));
    auto cold = [&](const char* nm, auto launch, unsigned replicas) {
      GArgs gb{d, big, N, rows, 0};
      timeit(nm, [&] { fill_table<<<(unsigned)((8192ull * replicas + 255) / 256), 256>>>(big, 8192ull * replicas); launch(gb, replicas); });
    };
    cold("P cold R=32 nt stores", [&](GArgs g, unsigned r) { rowstream_pf<true, false><<<dim3(96, 8192), 256>>>(g, r, 0); }, 32);
    cold("P cold R=32 plain no prefetch", [&](GArgs g, unsigned r) { rowstream_pf<false, false><<<dim3(96, 8192), 256>>>(g, r, 0); }, 32);
    cold("H cold table R=32 S=1 (4KiB/block)", [&](GArgs g, unsigned r) { rowstream_cold<1><<<dim3(96, 8192), 256>>>(g, r); }, 32);
    cold("H cold table R=96 S=1", [&](GArgs g, unsigned r) { rowstream_cold<1><<<dim3(96, 8192), 256>>>(g, r); }, 96);
    cold("H cold table R=32 S=2 (8KiB/block)", [&](GArgs g, unsigned r) { rowstream_cold<2><<<dim3(48, 8192), 256>>>(g, r); }, 32);
    cold("H cold table R=48 S=2", [&](GArgs g, unsigned r) { rowstream_cold<2><<<dim3(48, 8192), 256>>>(g, r); }, 48);
    cold("H cold table R=24 S=4 (16KiB/block)", [&](GArgs g, unsigned r) { rowstream_cold<4><<<dim3(24, 8192), 256>>>(g, r); }, 24);
    cold("H cold table R=12 S=8 (32KiB/block)", [&](GArgs g, unsigned r) { rowstream_cold<8><<<dim3(12, 8192), 256>>>(g, r); }, 12);
    cold("H cold table R=1 S=1", [&](GArgs g, unsigned r) { rowstream_cold<1><<<dim3(96, 8192), 256>>>(g, r); }, 1);
    timeit("G distinct line per block", [&] { rowstream_like<3><<<dim3(96, 8192, 1), 256>>>(ga); });
    timeit("G 8 replicas per row", [&] { rowstream_like<4><<<dim3(96, 8192, 1), 256>>>(ga); });
    timeit("G 32 replicas per row", [&] { rowstream_like<5><<<dim3(96, 8192, 1), 256>>>(ga); });
    timeit("F' 1 store/thread plain 256thr again", [&] { one_store<false, 256><<<(unsigned)((total + 255) / 256), 256>>>(d, total); });
  }
  timeit("E 2x16B adjacent/thread nt 256thr", [&] { two_adjacent<true, 256><<<(unsigned)((total / 2 + 255) / 256), 256>>>(d, total); });
  timeit("E 2x16B adjacent/thread plain 256thr", [&] { two_adjacent<false, 256><<<(unsigned)((total / 2 + 255) / 256), 256>>>(d, total); });
  for (uint64_t run : {1ull, 2ull, 4ull, 16ull, 64ull, 256ull, 4096ull, (unsigned long long)((total + 255) / 256 / 8)}) {
    char label[96];
    snprintf(label, sizeof label, "X 1 store/thread nt, XCD runs of %llu x 4 KiB", (unsigned long long)run);
    const uint64_t nb = (total + 255) / 256;
    const uint64_t nb8 = (nb + 8 * run - 1) / (8 * run) * (8 * run);  // whole super-chunks (extra workgroups store nothing)
    timeit(label, [&] { one_store_xcd<true><<<(unsigned)nb8, 256>>>(d, total, run); });
  }
  timeit("F 1 store/thread nt 64thr", [&] { one_store<true, 64><<<(unsigned)((total + 63) / 64), 64>>>(d, total); });
  timeit("F 1 store/thread nt 128thr", [&] { one_store<true, 128><<<(unsigned)((total + 127) / 128), 128>>>(d, total); });
  timeit("F 1 store/thread nt 256thr", [&] { one_store<true, 256><<<(unsigned)((total + 255) / 256), 256>>>(d, total); });
  timeit("F 1 store/thread nt 512thr", [&] { one_store<true, 512><<<(unsigned)((total + 511) / 512), 512>>>(d, total); });
  timeit("F 1 store/thread nt 1024thr", [&] { one_store<true, 1024><<<(unsigned)((total + 1023) / 1024), 1024>>>(d, total); });
  timeit("F 1 store/thread plain 256thr", [&] { one_store<false, 256><<<(unsigned)((total + 255) / 256), 256>>>(d, total); });
  timeit("F 1 store/thread plain 1024thr", [&] { one_store<false, 1024><<<(unsigned)((total + 1023) / 1024), 1024>>>(d, total); });
  CK(hipMemsetAsync(d, 0, bytes, 0));
  hipDeviceSynchronize();
  timeit("hipMemsetAsync(0)", [&] { hipMemsetAsync(d, 0, bytes, 0); });
  return 0;
}

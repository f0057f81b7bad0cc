// libinflx_hip.so -- host side of the MI355X grid sweep (C ABI declared in include/inflx_hip.h).
//
// Replaces, for the sweep path only, what the reference does in Rust:
//   src/dylib.rs          open the per-model artefact, check its ABI version, read its globals
//   src/hesse_bindings.rs bind the five model functions           (here: resolve the kernels)
//   src/anguelova.rs      validate shapes, convert ranges, drive the loop over grid points
// The loop itself runs on the GPU (csrc/inflx_sweep_kernels.hip); this file owns the code object,
// one stream, the parameter buffer and the chunk buffers of a model, and launches kernels with
// hipModuleLaunchKernel.  No CPU fallback exists: without a usable HIP device every entry point
// fails with INFLX_ERR_DEVICE.
#include "inflx_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <future>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>
#include <sys/mman.h>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif
#include <unistd.h>

#include "inflx_kernel_abi.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(INFLX_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

constexpr uint16_t kAbiMajor = 5, kAbiMinor = 0;  // src/lib.rs:50 V_INFLX_ABI
const char* const kOpNames[INFLX_OP_COUNT] = {"complete", "consistency", "rapidturn", "epsilon_v", "raw", "qdif", "hesse"};
constexpr int kOpWidth[INFLX_OP_COUNT] = {6, 1, 1, 1, 5, 1, 4};
// bytes per grid point of an operation's result (the flag sweep writes one bool per point)
constexpr size_t kOpBytes[INFLX_OP_COUNT] = {48, 8, 8, 8, 40, 1, 32};

// device chunk used by the host-result path: two buffers of this many bytes at most
size_t chunk_bytes_limit() {
  static const size_t v = [] {
    const char* e = getenv("INFLX_CHUNK_MB");  // tuning knob
    const long mb = e ? atol(e) : 0;
    return (size_t)(mb > 0 ? mb : 32) << 20;  // 32 MiB measured best (16: 34 GB/s, 32: 42, 64: 34, 128: 31)
  }();
  return v;
}
// host-result sweeps whose result is at most this large are evaluated into one device buffer and copied
// back with a single device-to-host copy; larger ones go through the chunk pipeline
size_t whole_result_limit() {
  static const size_t v = [] {
    const char* e = getenv("INFLX_WHOLE_RESULT_MB");  // tuning knob; 0 disables the path
    const long mb = e ? atol(e) : -1;
    return (size_t)(mb >= 0 ? mb : 65536) << 20;
  }();
  return v;
}
// ---- one host-thread budget per process ---------------------------------------------------------------------------------
// The helper threads of the host-result paths (page residency ahead of the DMA, streaming-store fill of broadcast results) are sized
// from the CPUs this PROCESS may use, read once: the scheduler affinity mask, cut down to the cgroup CPU quota if one is set anywhere
// between the process's own cgroup and the root (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us) -- a GPU box hands every GPU
// a share of the host (16 of 256 CPUs), which std::thread::hardware_concurrency() does not see.  The reference sizes its rayon pool
// from the same information (`threads` = 0 -> every core the process has, src/anguelova.rs:467,524-525).  In a multi-device call
// every device's pipeline runs on a host thread of its own and brings its own helpers: the budget is divided among them
// (tl_sharers), so that 8 devices on a 16-CPU share start 16 helpers, not 64+.
unsigned quota_from(const std::string& quota_file, const std::string& period_file) {
  // v2: "max 100000" / "1600000 100000" in one file; v1: two files, quota -1 = none.  0 = no limit found here.
  FILE* fh = fopen(quota_file.c_str(), "r");
  if (!fh) return 0;
  char a[64] = {0}, b[64] = {0};
  const int got = fscanf(fh, "%63s %63s", a, b);
  fclose(fh);
  if (got < 1 || strcmp(a, "max") == 0) return 0;
  double quota = atof(a), period = got >= 2 ? atof(b) : 0.0;
  if (!period_file.empty()) {
    FILE* ph = fopen(period_file.c_str(), "r");
    if (!ph) return 0;
    if (fscanf(ph, "%63s", b) == 1) period = atof(b);
    fclose(ph);
  }
  if (quota <= 0.0 || period <= 0.0) return 0;
  return (unsigned)std::max(1.0, std::floor(quota / period + 0.5));
}

// the cgroup path of this process for controller `want` ("" = the v2 unified hierarchy), from /proc/self/cgroup
std::string own_cgroup(const char* want) {
  FILE* fh = fopen("/proc/self/cgroup", "r");
  if (!fh) return "";
  char line[1024];
  std::string found;
  while (fgets(line, sizeof line, fh)) {
    // hierarchy-ID:controller-list:path
    char* c1 = strchr(line, ':');
    char* c2 = c1 ? strchr(c1 + 1, ':') : nullptr;
    if (!c2) continue;
    std::string ctrl(c1 + 1, c2), path(c2 + 1);
    while (!path.empty() && (path.back() == '\n' || path.back() == '\r')) path.pop_back();
    const bool match = *want ? (("," + ctrl + ",").find(std::string(",") + want + ",") != std::string::npos) : ctrl.empty();
    if (match) { found = path; break; }
  }
  fclose(fh);
  return found;
}

// smallest CPU quota on the way from `leaf` (a cgroup path below `mount`) up to the mount point; 0 = none
unsigned min_quota_upwards(const std::string& mount, std::string leaf, bool v2) {
  unsigned best = 0;
  for (;;) {
    const std::string dir = mount + (leaf == "/" ? "" : leaf);
    const unsigned q = v2 ? quota_from(dir + "/cpu.max", "") : quota_from(dir + "/cpu.cfs_quota_us", dir + "/cpu.cfs_period_us");
    if (q && (!best || q < best)) best = q;
    if (leaf.empty() || leaf == "/") break;
    const size_t cut = leaf.find_last_of('/');
    leaf = cut == 0 || cut == std::string::npos ? "/" : leaf.substr(0, cut);
  }
  return best;
}

unsigned detect_cpu_budget() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
  unsigned quota = min_quota_upwards("/sys/fs/cgroup", own_cgroup(""), true);  // v2 (inside a container the path may not exist below the mount: the root file still does)
  if (!quota) quota = min_quota_upwards("/sys/fs/cgroup", "/", true);
  for (const char* mount : {"/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"}) {  // v1
    unsigned q = min_quota_upwards(mount, own_cgroup("cpu"), false);
    if (!q) q = min_quota_upwards(mount, "/", false);
    if (q && (!quota || q < quota)) quota = q;
  }
  return quota ? std::min(n, quota) : n;
}

unsigned host_cpu_budget() {
  static const unsigned v = [] {
    const char* e = getenv("INFLX_HOST_THREADS");  // the budget by hand (tests, odd schedulers)
    const long n = e ? atol(e) : 0;
    return n > 0 ? (unsigned)n : detect_cpu_budget();
  }();
  return v;
}

// device pipelines that run at the same time as the calling thread's (set by run_parts for the duration of a part)
thread_local unsigned tl_sharers = 1;

// Helpers a host-result pipeline may start: {threads that make destination pages resident ahead of the DMA, threads that write a
// broadcast result with streaming stores}, for a process that may use `cpus` CPUs and runs `sharers` device pipelines at once.
//  * residency: half the budget, at most 8 (one thread touches ~10 GB/s of fresh pages; the DMA needs 50-57);
//  * fill: memory-bound streaming stores gain from two threads per CPU of a quota'd share (hyperbolic 8192^2 on 16 CPUs: 16 threads
//    15.2 ms, 32: 13.6, 64: 12.2) -- but only while nothing else competes for the share: with several devices at work every
//    pipeline gets its plain share and nothing is oversubscribed.
struct HelperPlan {
  unsigned prefault, fill;
};
HelperPlan helper_plan(unsigned cpus, unsigned sharers) {
  cpus = std::max(1u, cpus);
  sharers = std::max(1u, sharers);
  const unsigned share = std::max(1u, cpus / sharers);
  HelperPlan h;
  h.prefault = std::max(1u, std::min(8u, (share + 1) / 2));
  h.fill = std::min(64u, sharers == 1 ? 2 * share : share);
  return h;
}

unsigned prefault_threads() {
  static const long forced = [] {
    const char* e = getenv("INFLX_PREFAULT_THREADS");  // tuning knob: overrides the budget
    return e ? atol(e) : 0L;
  }();
  return forced > 0 ? (unsigned)forced : helper_plan(host_cpu_budget(), tl_sharers).prefault;
}

unsigned host_fill_threads() {
  static const long forced = [] {
    const char* e = getenv("INFLX_HOST_FILL_THREADS");  // tuning knob: overrides the budget
    return e ? atol(e) : 0L;
  }();
  return forced > 0 ? (unsigned)forced : helper_plan(host_cpu_budget(), tl_sharers).fill;
}

// Write-fault every page of a range without changing it: an atomic OR of 0 into one byte per page.  The
// atomic matters: helper threads may still be walking a buffer the DMA engine has started to fill, and a
// plain read-then-write could put a stale byte back over a freshly copied one; a locked read-modify-write
// holds the cache line for its duration, so a coherent DMA write lands entirely before or after it.
#if defined(__x86_64__)
constexpr bool kTouchIsAtomic = true;
#else
constexpr bool kTouchIsAtomic = false;  // the fallbacks below may read-then-write: never concurrently with the DMA
#endif
void touch_range(char* begin, size_t bytes) {
  const size_t page = (size_t)sysconf(_SC_PAGESIZE);
  char* lo = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(begin) + page - 1) / page * page);
  char* hi = reinterpret_cast<char*>(reinterpret_cast<uintptr_t>(begin + bytes) / page * page);
#if defined(__x86_64__)
  // spelled in assembly: compilers turn an idempotent atomic OR into a fenced *load*, which would map the
  // shared zero page instead of allocating a writable one
  for (char* q = lo; q < hi; q += page) asm volatile("lock orb $0, %0" : "+m"(*q) : : "cc");
#else
  if (hi > lo && madvise(lo, (size_t)(hi - lo), MADV_POPULATE_WRITE) != 0)
    for (volatile char* q = lo; q < hi; q += page) *q = *q;  // last resort, only safe when nothing else writes
#endif
}

void advise_huge_pages(char* begin, size_t bytes) {
#ifdef MADV_HUGEPAGE
  const size_t page = (size_t)sysconf(_SC_PAGESIZE);
  char* lo = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(begin) + page - 1) / page * page);
  char* hi = reinterpret_cast<char*>(reinterpret_cast<uintptr_t>(begin + bytes) / page * page);
  if (hi > lo) (void)madvise(lo, (size_t)(hi - lo), MADV_HUGEPAGE);
#else
  (void)begin, (void)bytes;
#endif
}

// Make the pages of a host destination range resident before the DMA engine writes to them.
// A result array fresh from np.zeros has no physical pages yet; letting the device-to-host copy fault
// them in one by one runs at 14 GB/s, copying into resident pages at 50 GB/s (scripts/pinned_probe.py).
// Contents are preserved (a read-modify-write of one byte per page; MADV_POPULATE_WRITE on request -- on the
// GPU box's host touching populates huge pages 2-5x faster, scripts/micro/prefault_probe.cpp).
void prefault_range(char* begin, size_t bytes) {
  const size_t page = (size_t)sysconf(_SC_PAGESIZE);
  char* lo = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(begin) + page - 1) / page * page);
  char* hi = reinterpret_cast<char*>(reinterpret_cast<uintptr_t>(begin + bytes) / page * page);
  if (hi <= lo) return;
#ifdef MADV_HUGEPAGE
  (void)madvise(lo, (size_t)(hi - lo), MADV_HUGEPAGE);
#endif
  const unsigned nthreads = (unsigned)std::min<size_t>(prefault_threads(), (size_t)(hi - lo) / (size_t(8) << 20) + 1);
  static const bool use_populate = [] {
    const char* e = getenv("INFLX_PREFAULT_MODE");  // tuning knob: "populate" or "touch"
    return e && strcmp(e, "populate") == 0;
  }();
  auto work = [](char* a, char* b) {
#ifdef MADV_POPULATE_WRITE
    if (use_populate && madvise(a, (size_t)(b - a), MADV_POPULATE_WRITE) == 0) return;
#endif
    touch_range(a, (size_t)(b - a));
  };
  std::vector<std::thread> pool;
  const size_t pages = (size_t)(hi - lo) / page;
  for (unsigned t = 0; t < nthreads; ++t) {
    char* a = lo + pages * t / nthreads * page;
    char* b = lo + pages * (t + 1) / nthreads * page;
    bool spawned = false;
    if (t + 1 < nthreads) {
      try {  // a thread that cannot be created (resource limits) must not take the process down: do its share here
        pool.emplace_back(work, a, b);
        spawned = true;
      } catch (...) {
      }
    }
    if (!spawned) work(a, b);
  }
  for (auto& th : pool) th.join();
}
}  // namespace

#define INFLX_SERIALISE(m) \
  std::unique_lock<std::recursive_mutex> inflx_call_lock_; \
  if (m) inflx_call_lock_ = std::unique_lock<std::recursive_mutex>((m)->mu)

struct inflx_model {
  // Every entry point that works on a handle holds this lock for the call: a handle owns streams, staging buffers, the
  // parameter-slot ring and the double-buffered tables, none of which two calls may use at once.  (The reference holds the
  // GIL for a whole sweep; ctypes releases it, so two Python threads can arrive here together.)  Recursive: the timed and
  // validating entry points call the plain ones.  Different handles never wait for each other.
  std::recursive_mutex mu;
  int device = 0;
  hipModule_t module = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;
  hipFunction_t tile[INFLX_OP_COUNT] = {};
  hipFunction_t rows[INFLX_OP_COUNT] = {};
  hipFunction_t traj[INFLX_OP_COUNT] = {};
  hipFunction_t basis_points = nullptr;
  hipFunction_t ops_on_values = nullptr;
  hipFunction_t rowvals[INFLX_OP_COUNT] = {};
  hipFunction_t rowstream6 = nullptr;
  hipFunction_t rowstream_planes = nullptr;
  hipFunction_t colvals[INFLX_OP_COUNT] = {};  // column-broadcast path: one row image per launch ...
  hipFunction_t colstream = nullptr;           // ... copied into every grid row
  hipFunction_t colvals_stats = nullptr;
  // tile path: stage tables (U per parameter row, R per grid row, C per grid column) written by inflx_stage_tables
  // on the side stream, double-buffered like the row table: the tables of launch n+1 are evaluated while the tile
  // kernel of launch n runs (the table kernel is a latency-bound chain on a few dozen workgroups)
  hipFunction_t stage_tables = nullptr;
  double* d_stage_tab[2] = {nullptr, nullptr};
  size_t d_stage_tab_cap[2] = {0, 0};  // doubles
  hipEvent_t stage_ready[2] = {nullptr, nullptr};  // inflx_stage_tables finished writing buffer b
  hipEvent_t stage_free[2] = {nullptr, nullptr};   // the tile kernel that last read buffer b finished
  bool stage_used[2] = {false, false};
  unsigned stage_turn = 0;
  // decided per call by evaluates_on_side_stream, read by launch_tiles: a tile sweep that is ONE launch builds its tables on
  // the caller's stream, in front of the tile kernel (no second stream, no event between the two)
  bool tables_on_callers_stream = false;
  // inflx_sweep_flags of the call that holds the lock (INFLX_SWEEP_FORCE_TILE: every grid point through the tile kernels even where
  // the model ignores a grid axis); set and cleared by the *_ex entry points
  unsigned call_flags = 0;
  hipFunction_t tile_stats = nullptr, tile_stats_nostore = nullptr, rowvals_stats = nullptr;
  double* d_stats = nullptr;  // 18 x 8 bytes: min[6], max[6], count[6]
  // Row-broadcast path: per-row results [P][rows][replicas][8], double-buffered.  The per-row
  // evaluation of sweep n runs on `side` and overlaps the store stream of sweep n-1 on the caller's
  // stream (it needs ~25 us of latency but hardly any bandwidth); events order table reuse.
  double* d_row_table[2] = {nullptr, nullptr};
  size_t d_row_table_cap[2] = {0, 0};
  hipStream_t side = nullptr;
  hipEvent_t table_ready[2] = {nullptr, nullptr};  // rowvals finished writing buffer b
  hipEvent_t table_free[2] = {nullptr, nullptr};   // the store stream that last read buffer b finished
  bool table_used[2] = {false, false};
  unsigned table_turn = 0;
  // inflx_sweep_device_timed(..., dominant_only = 2): event pairs recorded around every dominant-kernel launch of the sweeps it
  // enqueues -- the kernel's duration inside the full pipeline (side-stream evaluation overlapping, cross-stream waits in place)
  std::vector<std::pair<hipEvent_t, hipEvent_t>>* probe = nullptr;
  size_t probe_used = 0;
  bool probe_overflow = false;  // more dominant-kernel launches than event pairs: the timing would be partial
  InflxKernelInfo info = {};
  // kernel groups (inflx_kernel_abi.h): the artefact is the core object; the groups of the other operations are loaded beside it on
  // first use -- inflx_attach, or the file `<artefact>.<group>` if it exists (need_groups)
  uint32_t groups = 0;
  std::vector<hipModule_t> attached;
  std::string tag;  // MODEL_TAG of the core object: what an attached group must carry too
  // special-function status (csrc/inflx_sf.h): one INFLX_SF_STATUS word per loaded code object of a model that calls inflx_sf_*;
  // empty for every other model.  `sf_policy`: what a host-result call does when a word is set (inflx_sf_policy)
  std::vector<hipDeviceptr_t> sf_words;
  int sf_policy = INFLX_SF_QUIET;
  // a device-resident sweep runs on the CALLER's stream, which the handle neither owns nor may assume alive later: an event of the
  // handle's own is recorded behind the sweep, and reading the status waits for it
  hipEvent_t sf_done = nullptr;
  bool sf_pending = false;
  char use_gsl = 0;  // the artefact's USE_GSL global (Compiler(link_gsl=True), python/inflatox/compiler.py:558)
  uint16_t version[3] = {};
  uint32_t dim = 0, n_par = 0;
  std::string name, path;
  // Parameter rows travel through a small ring of (pinned host, device) buffer pairs: the caller's array is
  // copied into pinned memory before the call returns (so its lifetime ends with the call, whatever kind
  // of memory it is), the upload is a true asynchronous copy, and a slot is only overwritten once the
  // kernels that read it have finished (`last_use`) -- back-to-back sweeps with different parameters on
  // different streams never wait for each other on the host.
  struct ParamSlot {
    double* host = nullptr;         // pinned; always holds what `dev` holds (or will hold once the copy ran)
    double* dev = nullptr;
    size_t cap = 0, count = 0;      // doubles
    hipStream_t stream = nullptr;   // stream the upload was ordered on
    hipEvent_t last_use = nullptr;  // recorded behind the last kernel that reads `dev`
    hipStream_t last_reader = nullptr;  // stream `last_use` was recorded on
    bool in_flight = false;
  };
  static constexpr int kParamSlots = 4;
  ParamSlot pslot[kParamSlots];
  int pcur = 0;
  void* d_chunk[2] = {nullptr, nullptr};
  size_t d_chunk_cap[2] = {0, 0};
  void* d_whole = nullptr;  // whole-result buffer of the host path (results up to whole_result_limit())
  size_t d_whole_cap = 0;
  hipEvent_t chunk_done[2] = {nullptr, nullptr};
  hipEvent_t copy_done[2] = {nullptr, nullptr};
  hipEvent_t t0 = nullptr, t1 = nullptr;
};

namespace {

template <typename T>
int read_global(inflx_model* m, const char* sym, T* dst, size_t bytes, bool exact) {
  hipDeviceptr_t dptr = nullptr;
  size_t size = 0;
  if (hipModuleGetGlobal(&dptr, &size, m->module, sym) != hipSuccess)
    return fail(INFLX_ERR_SYMBOL, "artefact %s lacks symbol %s", m->path.c_str(), sym);
  if (exact ? size != bytes : size > bytes)
    return fail(INFLX_ERR_SYMBOL, "symbol %s in %s has %zu bytes, expected %s%zu", sym, m->path.c_str(), size,
                exact ? "" : "<= ", bytes);
  HIP_TRY(hipMemcpy(dst, dptr, size, hipMemcpyDeviceToHost));
  return INFLX_OK;
}

// Put `count` doubles of parameters where kernels enqueued on `s` can read them; *d_params is the device
// address.  A sweep re-launches with the same rows far more often than it changes them: when the current
// slot already holds these values for this stream nothing is uploaded.
int acquire_params(inflx_model* m, const double* p, size_t count, hipStream_t s, const double** d_params) {
  inflx_model::ParamSlot& cur = m->pslot[m->pcur];
  if (cur.dev && cur.count == count && cur.stream == s && memcmp(cur.host, p, count * sizeof(double)) == 0) {
    *d_params = cur.dev;
    return INFLX_OK;
  }
  const int next = (m->pcur + 1) % inflx_model::kParamSlots;
  inflx_model::ParamSlot& slot = m->pslot[next];
  if (slot.in_flight) {
    HIP_TRY(hipEventSynchronize(slot.last_use));  // four sweeps ago: long finished unless the queue is that deep
    slot.in_flight = false;
  }
  if (!slot.last_use) HIP_TRY(hipEventCreateWithFlags(&slot.last_use, hipEventDisableTiming));
  if (count > slot.cap) {
    if (slot.dev) HIP_TRY(hipFree(slot.dev));
    if (slot.host) HIP_TRY(hipHostFree(slot.host));
    slot.dev = slot.host = nullptr;
    slot.cap = slot.count = 0;
    const size_t cap = std::max<size_t>(count, 64);
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&slot.host), cap * sizeof(double), hipHostMallocDefault));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&slot.dev), cap * sizeof(double)));
    slot.cap = cap;
  }
  memcpy(slot.host, p, count * sizeof(double));
  HIP_TRY(hipMemcpyAsync(slot.dev, slot.host, count * sizeof(double), hipMemcpyHostToDevice, s));
  slot.count = count;
  slot.stream = s;
  m->pcur = next;
  *d_params = slot.dev;
  return INFLX_OK;
}

// every kernel that reads the current parameter slot has been enqueued, the last of them on `reader`
int release_params(inflx_model* m, hipStream_t reader) {
  inflx_model::ParamSlot& cur = m->pslot[m->pcur];
  if (!cur.last_use) return INFLX_OK;  // nothing was ever uploaded into this slot
  // One event per slot: when the readers of an earlier call ran on another stream (a cache hit from a different caller
  // stream), the new reader first waits for them, so that the re-recorded event still stands behind every reader.
  if (cur.in_flight && cur.last_reader != reader) HIP_TRY(hipStreamWaitEvent(reader, cur.last_use, 0));
  HIP_TRY(hipEventRecord(cur.last_use, reader));
  cur.last_reader = reader;
  cur.in_flight = true;
  return INFLX_OK;
}

// The same on an error path: kernels that were enqueued before the failure may still read the slot, so it must be
// marked in flight all the same; the first error is the one reported.
int release_params_after(inflx_model* m, hipStream_t reader, int rc) {
  if (rc == INFLX_OK) return release_params(m, reader);
  const std::string first = g_last_error;
  (void)release_params(m, reader);
  g_last_error = first;
  return rc;
}

// ---- special-function status ------------------------------------------------------------------------------------------------
// A model that calls the device counterparts of the reference's gsl_sf_* functions (csrc/inflx_sf.h) carries the word
// INFLX_SF_STATUS in each of its code objects; a function that is called outside its domain returns NaN for the point and sets a
// bit there.  The reference's GSL error handler prints the reason and panics (src/err.rs:86-103, installed by src/dylib.rs:141-148
// when USE_GSL = 1): the call does not return.  Here the host reads the words after the sweep.
void note_sf_word(inflx_model* m, hipModule_t module) {
  // only a USE_GSL artefact can call the special functions (the printer refuses them otherwise, as the reference's does): every other
  // model is left without status words, i.e. without any of the bookkeeping below
  if (!m->use_gsl) return;
  hipDeviceptr_t dptr = nullptr;
  size_t size = 0;
  if (hipModuleGetGlobal(&dptr, &size, module, "INFLX_SF_STATUS") == hipSuccess && size == sizeof(unsigned))
    m->sf_words.push_back(dptr);
  else
    (void)hipGetLastError();  // a model without special functions has no such word
}

// OR of the status words of every code object of the handle; everything the handle has enqueued is waited for first
int sf_collect(inflx_model* m, bool clear, unsigned* bits) {
  *bits = 0;
  if (m->sf_words.empty()) return INFLX_OK;
  HIP_TRY(hipSetDevice(m->device));
  HIP_TRY(hipStreamSynchronize(m->side));
  HIP_TRY(hipStreamSynchronize(m->stream));
  if (m->sf_pending) {
    HIP_TRY(hipEventSynchronize(m->sf_done));
    m->sf_pending = false;
  }
  for (hipDeviceptr_t w : m->sf_words) {
    unsigned v = 0;
    HIP_TRY(hipMemcpy(&v, w, sizeof v, hipMemcpyDeviceToHost));
    *bits |= v;
    if (v && clear) {
      const unsigned zero = 0;
      HIP_TRY(hipMemcpy(w, &zero, sizeof zero, hipMemcpyHostToDevice));
    }
  }
  return INFLX_OK;
}

int sf_error(const inflx_model* m, unsigned bits) {
  // (the first words are the reference handler's, src/err.rs:98 -- spelling included; GSL_EDOM = 1, GSL_ELOSS = 17)
  const bool dom = (bits & INFLX_SF_EDOM) != 0;
  return fail(INFLX_ERR_GSL, "a GSL exception ocurred (ERRCODE %s): model \"%s\" called a special function %s at one or more points of this call. "
              "Those points hold NaN and the result is complete otherwise; the reference's GSL error handler panics here instead of returning. "
              "inflx_sf_policy(handle, INFLX_SF_QUIET) / GeneralisedAL(..., sf_errors=\"nan\") returns the result with its NaNs.",
              dom ? "0X1" : "0X11", m->name.c_str(),
              dom ? "outside its domain (input domain error)" : "where this implementation could not deliver 1e-13 and declined (loss of accuracy)");
}

// what a host-result call returns after its result is complete
int sf_verdict(inflx_model* m) {
  if (m->sf_words.empty() || m->sf_policy != INFLX_SF_FAIL) return INFLX_OK;
  unsigned bits = 0;
  const int rc = sf_collect(m, true, &bits);
  if (rc) return rc;
  return bits ? sf_error(m, bits) : INFLX_OK;
}

// ---- kernel groups ----------------------------------------------------------------------------------------------------------
const char* const kGroupNames[9] = {"core", "stats", "values", "consistency", "rapidturn", "epsilon_v", "raw", "qdif", "hesse"};
uint32_t group_of_op(int op) { return INFLX_GROUP_OF_OP(op); }

// the kernels of the groups `mask` names, out of `module`
int resolve_group_kernels(inflx_model* m, hipModule_t module, uint32_t mask, const char* path) {
  auto get = [&](hipFunction_t* slot, const std::string& name) -> int {
    if (hipModuleGetFunction(slot, module, name.c_str()) != hipSuccess) {
      *slot = nullptr;
      return fail(INFLX_ERR_SYMBOL, "artefact %s lacks kernel %s", path, name.c_str());
    }
    return INFLX_OK;
  };
  int rc;
  for (int op = 0; op < INFLX_OP_COUNT; ++op) {
    if (!(mask & group_of_op(op))) continue;
    const std::string nm = kOpNames[op];
    if ((rc = get(&m->tile[op], "inflx_sweep_tile_" + nm)) || (rc = get(&m->rows[op], "inflx_sweep_rows_" + nm)) || (rc = get(&m->traj[op], "inflx_sweep_traj_" + nm)) ||
        (rc = get(&m->rowvals[op], "inflx_sweep_rowvals_" + nm)) || (rc = get(&m->colvals[op], "inflx_sweep_colvals_" + nm)))
      return rc;
  }
  if (mask & INFLX_GROUP_CORE) {
    if ((rc = get(&m->rowstream6, "inflx_sweep_rowstream6")) || (rc = get(&m->rowstream_planes, "inflx_sweep_rowstream_planes")) || (rc = get(&m->colstream, "inflx_sweep_colstream")) ||
        (rc = get(&m->stage_tables, "inflx_stage_tables")) || (rc = get(&m->basis_points, "inflx_basis_points")))
      return rc;
  }
  if (mask & INFLX_GROUP_STATS) {
    if ((rc = get(&m->tile_stats, "inflx_sweep_tile_complete_stats")) || (rc = get(&m->tile_stats_nostore, "inflx_sweep_tile_complete_stats_nostore")) ||
        (rc = get(&m->rowvals_stats, "inflx_sweep_rowvals_complete_stats")) || (rc = get(&m->colvals_stats, "inflx_sweep_colvals_complete_stats")))
      return rc;
  }
  if (mask & INFLX_GROUP_VALUES) {
    if ((rc = get(&m->ops_on_values, "inflx_ops_on_values"))) return rc;
  }
  return INFLX_OK;
}

template <typename T>
int read_global_of(hipModule_t module, const char* path, const char* sym, T* dst, size_t bytes, bool exact) {
  hipDeviceptr_t dptr = nullptr;
  size_t size = 0;
  if (hipModuleGetGlobal(&dptr, &size, module, sym) != hipSuccess) return fail(INFLX_ERR_SYMBOL, "artefact %s lacks symbol %s", path, sym);
  if (exact ? size != bytes : size > bytes) return fail(INFLX_ERR_SYMBOL, "symbol %s in %s has %zu bytes, expected %s%zu", sym, path, size, exact ? "" : "<= ", bytes);
  HIP_TRY(hipMemcpy(dst, dptr, size, hipMemcpyDeviceToHost));
  return INFLX_OK;
}

// Load the code object at `path` beside the core object of `m`: it must be built from the same generated model with the same options
// (MODEL_TAG) for the same kernel ABI; the kernels of the groups it carries and the handle lacks become available.
int attach_object(inflx_model* m, const char* path) {
  hipModule_t module = nullptr;
  hipError_t e = hipModuleLoad(&module, path);
  if (e != hipSuccess) return fail(INFLX_ERR_IO, "could not load %s as a gfx950 code object: %s", path, hipGetErrorString(e));
  auto bail = [&](int code) {
    const std::string first = g_last_error;
    (void)hipModuleUnload(module);
    g_last_error = first;
    return code;
  };
  uint16_t version[3] = {};
  InflxKernelInfo info = {};
  uint32_t groups = 0;
  char tag[128] = {0};
  int rc;
  if ((rc = read_global_of(module, path, "VERSION", version, sizeof version, true)) || (rc = read_global_of(module, path, "INFLX_KERNEL_INFO", &info, sizeof info, true)) ||
      (rc = read_global_of(module, path, "INFLX_GROUPS", &groups, sizeof groups, true)) || (rc = read_global_of(module, path, "MODEL_TAG", tag, sizeof tag - 1, false)))
    return bail(rc);
  if (version[0] != m->version[0] || version[1] != m->version[1] || memcmp(&info, &m->info, sizeof info) != 0 || m->tag != tag)
    return bail(fail(INFLX_ERR_VERSION, "%s does not belong to artefact %s: built from another model, with other options or for another kernel ABI (tag \"%s\", expected \"%s\")",
                     path, m->path.c_str(), tag, m->tag.c_str()));
  const uint32_t fresh = groups & ~m->groups;
  if ((rc = resolve_group_kernels(m, module, fresh, path))) return bail(rc);
  m->attached.push_back(module);
  m->groups |= fresh;
  note_sf_word(m, module);
  return INFLX_OK;
}

// Make the kernels of the groups in `mask` available: those the handle lacks are looked for next to the artefact, as
// `<artefact>.<group>` (where inflatox_amd puts a group when it builds one; a C client may put them there too).
int need_groups(inflx_model* m, uint32_t mask) {
  uint32_t missing = mask & ~m->groups & INFLX_GROUP_ALL;
  for (int b = 0; missing && b < 9; ++b) {
    const uint32_t bit = 1u << b;
    if (!(missing & bit)) continue;
    const std::string side = m->path + "." + kGroupNames[b];
    FILE* fh = fopen(side.c_str(), "rb");
    if (!fh)
      return fail(INFLX_ERR_SYMBOL, "the kernels of group \"%s\" are not part of %s and no %s exists: build that group (inflatox_amd does on first use; "
                  "Compiler(kernel_groups=\"all\") builds a complete artefact) and pass it to inflx_attach", kGroupNames[b], m->path.c_str(), side.c_str());
    fclose(fh);
    HIP_TRY(hipSetDevice(m->device));
    const int rc = attach_object(m, side.c_str());
    if (rc) return rc;
    missing = mask & ~m->groups & INFLX_GROUP_ALL;
  }
  return INFLX_OK;
}

// validate_lib + validiate_p (src/anguelova.rs:55-79) and the Hesse2D guard (hesse_bindings.rs:203)
int validate(const inflx_model* m, int op, const double* p, size_t P, size_t n_p, bool kernels = true) {
  if (!m) return fail(INFLX_ERR_ARG, "model handle is NULL");
  if (op < 0 || op >= INFLX_OP_COUNT) return fail(INFLX_ERR_ARG, "unknown sweep operation %d", op);
  if (m->dim != 2)
    return fail(INFLX_ERR_SHAPE, "the Anguelova & Lazaroiu consistency condition requires a 2-field model (model has %u fields)",
                m->dim);
  if (n_p != m->n_par)
    return fail(INFLX_ERR_SHAPE, "model \"%s\" has %u paramters (got %zu)", m->name.c_str(), m->n_par, n_p);
  if (!p && n_p) return fail(INFLX_ERR_ARG, "parameter array is NULL");
  if (P == 0) return fail(INFLX_ERR_SHAPE, "parameter array has no rows");
  // the kernels of this operation (loaded on first use when the artefact is a core object)
  if (kernels && (m->groups & group_of_op(op)) == 0) {
    inflx_model* mm = const_cast<inflx_model*>(m);
    INFLX_SERIALISE(mm);
    return need_groups(mm, group_of_op(op));
  }
  return INFLX_OK;
}

int ensure_row_table(inflx_model* m, int b, size_t doubles) {
  if (doubles <= m->d_row_table_cap[b]) return INFLX_OK;
  // a larger table is needed: nothing may still be using the old one
  HIP_TRY(hipDeviceSynchronize());
  if (m->d_row_table[b]) HIP_TRY(hipFree(m->d_row_table[b]));
  m->d_row_table[b] = nullptr;
  m->d_row_table_cap[b] = 0;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->d_row_table[b]), doubles * sizeof(double)));
  m->d_row_table_cap[b] = doubles;
  return INFLX_OK;
}

// does this sweep take the two-launch row-broadcast path (per-row evaluation + store stream)?
bool takes_row_stream(const inflx_model* m, int op, int layout, size_t P, size_t N1) {
  if ((m->info.out_mask & 2u) != 0 || op == INFLX_OP_QDIF || (m->call_flags & INFLX_SWEEP_FORCE_TILE)) return false;
  const bool aos6 = kOpWidth[op] == 6 && layout == INFLX_AOS;
  (void)P;  // the path never depends on the number of parameter rows: they are batched (row_stream_plan)
  const bool planes = (layout == INFLX_SOA || kOpWidth[op] == 1) && N1 % 2 == 0;
  return aos6 || planes;
}

// does this sweep take the two-launch column-broadcast path (row image + copy stream)?  No model value depends on
// x[0], and an output row is a whole number of 16-byte units.
bool takes_col_stream(const inflx_model* m, int op, int layout, size_t P, size_t N1) {
  if ((m->info.out_mask & 3u) != 2u || op == INFLX_OP_QDIF || (m->call_flags & INFLX_SWEEP_FORCE_TILE)) return false;
  const size_t K = kOpWidth[op];
  const bool planes = layout == INFLX_SOA || K == 1;
  (void)P;  // batched in launch_col_stream
  if (planes) return N1 % 2 == 0;
  return (K * N1) % 2 == 0;
}

// Geometry of the two-launch row-broadcast path for one call (shared by launch_grid and inflx_sweep_plan).
struct RowStreamPlan {
  size_t cpr;       // workgroups (4 KiB pieces) per grid row
  size_t replicas;  // copies of every row's table entry, a power of two
  size_t batch;     // parameter rows per table batch
};
RowStreamPlan row_stream_plan(const inflx_model* m, int op, int layout, size_t P, size_t N1, size_t row_count) {
  const bool aos6 = kOpWidth[op] == 6 && layout == INFLX_AOS;
  const size_t units_row = aos6 ? 3 * N1 : N1 / 2;
  RowStreamPlan r;
  r.cpr = (units_row + m->info.row_chunk_units - 1) / m->info.row_chunk_units;
  // replicas of every row's table entry (see inflx_kernel_abi.h); a power of two (the evaluation
  // kernel indexes with shifts), fewer for short rows
  r.replicas = 32;
  while (r.replicas > 1 && r.replicas > r.cpr) r.replicas /= 2;
  // The table has to stay in the 256 MiB Infinity Cache between its evaluation and its use (a table
  // fetch from HBM throttles the store stream from 6.6 to 5.0 TB/s, measured at P = 16): parameter
  // rows are processed in batches whose table is at most 64 MiB, each batch = evaluation + stream,
  // and the double-buffered side stream overlaps the evaluation of batch k+1 with the stream of batch k.
  const size_t line_bytes = row_count * r.replicas * 64;
  r.batch = std::max<size_t>(1, std::min<size_t>(P, (size_t(64) << 20) / std::max<size_t>(line_bytes, 1)));
  r.batch = std::min<size_t>(r.batch, 65535 / (aos6 ? 1 : kOpWidth[op]));  // grid.z of the store stream = batch x planes
  return r;
}

// Parameter rows per table batch of the column-broadcast path (images of at most 64 MiB, grid.z of the copy stream).
size_t col_stream_batch(int op, int layout, size_t P, size_t N1) {
  const size_t K = kOpWidth[op];
  const size_t images_per_p = (layout == INFLX_SOA || K == 1) ? K : 1;
  size_t batch = std::max<size_t>(1, std::min<size_t>(P, (size_t(64) << 20) / std::max<size_t>(K * N1 * 8, 1)));
  return std::min<size_t>(batch, 65535 / images_per_p);
}

// The tile path as a sequence of launches: parameter rows in batches whose tables fit 1 GiB, grid rows in slabs of at most
// 65535 tiles (grid.y).
struct TilePlan {
  size_t rows_per_launch, pbatch;
};
TilePlan tile_plan(const inflx_model* m, size_t P, size_t N1, size_t row_count) {
  const size_t nu = std::max<size_t>(m->info.n_uniform, 1), nc = std::max<size_t>(m->info.n_col, 1);
  const size_t nr = (std::max<size_t>(m->info.n_row, 1) + 1) & ~size_t(1);  // even stride of a row's values (kNRs of the kernels)
  TilePlan t;
  t.rows_per_launch = size_t(65535) * m->info.tile_rows;
  const size_t per_p = nu + std::min(t.rows_per_launch, row_count) * nr + nc * N1;  // doubles per parameter row
  t.pbatch = std::max<size_t>(1, std::min<size_t>(P, (size_t(1) << 27) / std::max<size_t>(per_p, 1)));
  return t;
}
// Which stream are the parameters uploaded on?  The first kernel that reads them runs on the side stream for the two
// broadcast paths (per-row / per-column evaluation) and for the tile path (stage tables) -- unless the sweep is a single
// pair of launches enqueued alone (below) --; the fallback row kernel of row-only models reads them on the caller's stream.
// `alone`: the sweep is enqueued on its own and waited for (a host-result call that is one launch + one copy).  Sweeps that may be
// enqueued back to back -- the asynchronous device-result entry points, the chunk pipeline -- keep the tables on the side stream, where
// the table kernel of sweep k+1 overlaps the tile kernel of sweep k like the launches of a multi-launch sweep do.
bool evaluates_on_side_stream(inflx_model* m, int op, int layout, size_t P, size_t N1, size_t row_count, bool alone = false) {
  m->tables_on_callers_stream = false;
  // the broadcast paths: per-row / per-column evaluation, then the store stream; one table batch = one pair of launches
  if (takes_row_stream(m, op, layout, P, N1)) {
    m->tables_on_callers_stream = alone && P <= row_stream_plan(m, op, layout, P, N1, row_count).batch;
    return !m->tables_on_callers_stream;
  }
  if (takes_col_stream(m, op, layout, P, N1)) {
    m->tables_on_callers_stream = alone && P <= col_stream_batch(op, layout, P, N1);
    return !m->tables_on_callers_stream;
  }
  const bool row_uniform = (m->info.out_mask & 2u) == 0 && op != INFLX_OP_QDIF && !(m->call_flags & INFLX_SWEEP_FORCE_TILE);
  if (row_uniform) return false;
  // tile path: tables, then tile kernel.  Several launches: the tables of launch k+1 are built on the side stream while the tile
  // kernel of launch k runs.  A single launch that nothing follows has nothing to overlap with, and the hop between two streams
  // (event record, wait, a second doorbell) is 10-15 us of the ~60 us a small sweep takes: it runs on the caller's stream alone.
  const TilePlan t = tile_plan(m, P, N1, row_count);
  m->tables_on_callers_stream = alone && P <= t.pbatch && row_count <= t.rows_per_launch;
  // (experiments, scripts/tables_policy_probe.py: "same" = the tables of every single-launch sweep on the caller's stream, "side" = never)
  static const int forced = [] {
    const char* e = getenv("INFLX_EXPERIMENT_TABLES");
    return !e ? 0 : (strcmp(e, "same") == 0 ? 1 : (strcmp(e, "side") == 0 ? 2 : 0));
  }();
  if (forced == 1) m->tables_on_callers_stream = P <= t.pbatch && row_count <= t.rows_per_launch;
  if (forced == 2) m->tables_on_callers_stream = false;
  return !m->tables_on_callers_stream;
}
// ... and on which stream does the LAST kernel that reads them run?  The tile kernels read the parameter rows too
// (the point stage uses args[k]), after the table kernel; on the broadcast paths the store streams do not.
bool last_reader_is_callers_stream(const inflx_model* m, int op, int layout, size_t P, size_t N1) {
  return !(takes_row_stream(m, op, layout, P, N1) || takes_col_stream(m, op, layout, P, N1));
}

// Has every kernel this handle enqueued earlier finished?  A sweep that arrives at an idle handle has nothing to overlap its
// table evaluation with: it is enqueued as `alone` (tables in front of the tile kernel on the caller's stream, no second stream, no
// event hop).  One that arrives while the previous sweep's kernels are still running keeps the side stream, where its tables are
// evaluated under them.  (A user's lone call and the back-to-back calls of a scan both get the shorter of the two shapes.)
bool handle_idle(inflx_model* m) {
  bool idle = true;
  for (int b = 0; b < 2 && idle; ++b) {
    if (m->stage_used[b] && hipEventQuery(m->stage_free[b]) != hipSuccess) idle = false;
    if (m->table_used[b] && hipEventQuery(m->table_free[b]) != hipSuccess) idle = false;
  }
  (void)hipGetLastError();  // hipErrorNotReady is an answer, not a failure
  return idle;
}

// the inflx_sweep_flags of an *_ex call, for as long as it holds the handle's lock
struct FlagScope {
  inflx_model* m;
  unsigned before;
  FlagScope(inflx_model* m_, unsigned flags) : m(m_), before(m_ ? m_->call_flags : 0u) {
    if (m) m->call_flags = flags;
  }
  ~FlagScope() {
    if (m) m->call_flags = before;
  }
  FlagScope(const FlagScope&) = delete;
  FlagScope& operator=(const FlagScope&) = delete;
};

// (in-pipeline timing of the dominant kernel, see inflx_model::probe) -- no-ops unless a probe is armed and has pairs left
hipError_t probe_begin(inflx_model* m, hipStream_t s) {
  if (!m->probe) return hipSuccess;
  if (m->probe_used >= m->probe->size()) {
    m->probe_overflow = true;
    return hipSuccess;
  }
  return hipEventRecord((*m->probe)[m->probe_used].first, s);
}
hipError_t probe_end(inflx_model* m, hipStream_t s) {
  if (!m->probe || m->probe_used >= m->probe->size()) return hipSuccess;
  return hipEventRecord((*m->probe)[m->probe_used++].second, s);
}

// ---- the four ways a sweep is enqueued; `a` arrives with the grid geometry filled in (launch_grid) -------------------

// Row-broadcast path: per-row values into the row table on the side stream, then the broadcast store stream on `s`
// (one 16-byte store per thread, 4 KiB per workgroup).  `what`: 0 = both, 1 = only the evaluation, 2 = only the
// store stream (used to time the dominant kernel on its own).
int launch_row_stream(inflx_model* m, int op, InflxSweepArgs a, const double* d_params, size_t P, double* d_out, size_t N1, size_t row_count, int layout, hipStream_t s, int what, double* d_stats) {
  void* params[] = {&a};
  const bool aos6 = kOpWidth[op] == 6 && layout == INFLX_AOS;
  hipStream_t ev = m->tables_on_callers_stream ? s : m->side;  // (decided with the stream of the parameter upload, evaluates_on_side_stream)
  const RowStreamPlan plan = row_stream_plan(m, op, layout, P, N1, row_count);
  const size_t cpr = plan.cpr, replicas = plan.replicas, batch = plan.batch;
  if (cpr > 0x7fffffffULL) return fail(INFLX_ERR_SHAPE, "grid rows too long for one launch");
  const size_t K = kOpWidth[op];
  // the timing-only mode re-runs store streams from the table of the sweep before it, and only the last
  // batch's table is still there
  if (what == 2 && batch < P)
    return fail(INFLX_ERR_ARG, "dominant_only timing needs the parameter rows to fit one table batch (%zu rows here, got %zu)", batch, P);
  for (size_t p0 = 0; p0 < P; p0 += batch) {
    const size_t pb = std::min(batch, P - p0);
    // `what` == 2 (timing only) re-runs the store streams from whatever the tables hold
    const int b = what == 2 ? (int)((m->table_turn + 1) & 1) : (int)(m->table_turn & 1);
    const bool store = d_out != nullptr;
    int rc = store ? ensure_row_table(m, b, pb * row_count * replicas * 8) : INFLX_OK;
    if (rc) return rc;
    a.params = d_params + p0 * m->n_par;
    a.out = store ? d_out + p0 * row_count * N1 * K : nullptr;  // same offset for [P][rows][N1][K] and [P][K][rows][N1]
    a.P = (uint32_t)pb;
    a.row_table = store ? m->d_row_table[b] : nullptr;
    a.table_replicas = (uint32_t)replicas;
    a.stream_planes = (uint32_t)K;
    if (what != 2) {
      // per-row evaluation on the side stream (on the caller's for a single pair of launches enqueued alone), as soon as the
      // previous reader of this table is done
      if (m->table_used[b]) HIP_TRY(hipStreamWaitEvent(ev, m->table_free[b], 0));
      HIP_TRY(hipModuleLaunchKernel(d_stats ? m->rowvals_stats : m->rowvals[op], (unsigned)((row_count + 63) / 64), (unsigned)pb, 1, 64, 1, 1, 0,
                                    ev, params, nullptr));
      if (ev != s) HIP_TRY(hipEventRecord(m->table_ready[b], ev));
      m->table_turn++;
    }
    if (what != 1 && store) {
      if (ev != s) HIP_TRY(hipStreamWaitEvent(s, m->table_ready[b], 0));
      // grid.y is limited to 65535, longer slabs take several launches
      for (size_t r0 = 0; r0 < row_count; r0 += 65535) {
        a.stream_row0 = (uint32_t)r0;
        const size_t nr = std::min<size_t>(65535, row_count - r0);
        HIP_TRY(probe_begin(m, s));
        HIP_TRY(hipModuleLaunchKernel(aos6 ? m->rowstream6 : m->rowstream_planes, (unsigned)cpr, (unsigned)nr,
                                      (unsigned)(aos6 ? pb : pb * K), m->info.tile_cols, 1, 1, 0, s, params, nullptr));
        HIP_TRY(probe_end(m, s));
      }
      HIP_TRY(hipEventRecord(m->table_free[b], s));
      m->table_used[b] = true;
    }
  }
  return INFLX_OK;
}

// Column-broadcast path: the image of one output row per parameter row (and plane) into the table on the side stream,
// then the copy stream on `s`.
int launch_col_stream(inflx_model* m, int op, InflxSweepArgs a, const double* d_params, size_t P, double* d_out, size_t N1, size_t row_count, int layout, hipStream_t s, int what, double* d_stats) {
  void* params[] = {&a};
  const size_t K = kOpWidth[op];
  hipStream_t ev = m->tables_on_callers_stream ? s : m->side;
  const bool planes = layout == INFLX_SOA || K == 1;
  const size_t images_per_p = planes ? K : 1;
  const size_t units = (planes ? N1 : K * N1) / 2;  // 16-byte units per output row
  const size_t cpr = (units + m->info.tile_cols - 1) / m->info.tile_cols;
  if (cpr > 0x7fffffffULL) return fail(INFLX_ERR_SHAPE, "grid rows too long for one launch");
  const size_t image_doubles = K * N1;  // per parameter row
  const size_t batch = col_stream_batch(op, layout, P, N1);
  if (what == 2 && batch < P)
    return fail(INFLX_ERR_ARG, "dominant_only timing needs the parameter rows to fit one table batch (%zu rows here, got %zu)", batch, P);
  const size_t gx = (N1 + m->info.tile_cols - 1) / m->info.tile_cols;
  for (size_t p0 = 0; p0 < P; p0 += batch) {
    const size_t pb = std::min(batch, P - p0);
    const int b = what == 2 ? (int)((m->table_turn + 1) & 1) : (int)(m->table_turn & 1);
    const bool store = d_out != nullptr;
    int rc = store ? ensure_row_table(m, b, pb * image_doubles) : INFLX_OK;
    if (rc) return rc;
    a.params = d_params + p0 * m->n_par;
    a.out = store ? d_out + p0 * row_count * N1 * K : nullptr;
    a.P = (uint32_t)pb;
    a.row_table = store ? m->d_row_table[b] : nullptr;
    a.stream_units = units;
    if (what != 2) {
      if (m->table_used[b]) HIP_TRY(hipStreamWaitEvent(ev, m->table_free[b], 0));
      HIP_TRY(hipModuleLaunchKernel(d_stats ? m->colvals_stats : m->colvals[op], (unsigned)gx, (unsigned)pb, 1, m->info.tile_cols, 1, 1, 0, ev, params,
                                    nullptr));
      if (ev != s) HIP_TRY(hipEventRecord(m->table_ready[b], ev));
      m->table_turn++;
    }
    if (what != 1 && store) {
      if (ev != s) HIP_TRY(hipStreamWaitEvent(s, m->table_ready[b], 0));
      for (size_t r0 = 0; r0 < row_count; r0 += 65535) {
        a.stream_row0 = (uint32_t)r0;
        const size_t nr = std::min<size_t>(65535, row_count - r0);
        HIP_TRY(probe_begin(m, s));
        HIP_TRY(hipModuleLaunchKernel(m->colstream, (unsigned)cpr, (unsigned)nr, (unsigned)(pb * images_per_p), m->info.tile_cols, 1, 1, 0, s, params, nullptr));
        HIP_TRY(probe_end(m, s));
      }
      HIP_TRY(hipEventRecord(m->table_free[b], s));
      m->table_used[b] = true;
    }
  }
  return INFLX_OK;
}

// Fallback of row-only models for result shapes the store streams do not cover: per-row evaluation inside the kernel.
int launch_rows_fallback(inflx_model* m, int op, InflxSweepArgs a, size_t P, size_t N1, size_t row_count, hipStream_t s) {
  void* params[] = {&a};
  const size_t rpb = m->info.rows_per_block;
  const size_t groups = (row_count + rpb - 1) / rpb;
  // few rows: split each row into column chunks until the grid can fill 256 CUs x 8 workgroups
  size_t chunks = 1;
  const size_t want = 4096;
  if (groups * P < want) {
    const size_t units = N1 / 64;
    chunks = std::min<size_t>(std::max<size_t>(units, 1), (want + groups * P - 1) / (groups * P));
  }
  if (groups * chunks > 0x7fffffffULL) return fail(INFLX_ERR_SHAPE, "grid too large for one launch");
  a.col_chunks = (uint32_t)chunks;
  HIP_TRY(hipModuleLaunchKernel(m->rows[op], (unsigned)(groups * chunks), (unsigned)P, 1, m->info.tile_cols, 1, 1, 0, s, params,
                                nullptr));
  return INFLX_OK;
}

// Tile path (some value depends on x[1]).  Stage tables U[P][nu] | R[P][slab][nr] | C[P][nc][N1] (doubles) are written
// by inflx_stage_tables on the side stream into one of two buffers and read by the tile kernel on `s` behind an event,
// so that in a sequence of launches the tables of launch n+1 are evaluated while the tile kernel of launch n runs.
// grid.y is limited to 65535 tiles: a taller slab takes several launches, each with tables of its own; parameter rows
// are batched so that one set of tables stays below 1 GiB.
constexpr size_t kLaunchWorkgroups = 1024;  // workgroups a tile launch is cut into when full-height tiles would give fewer (256 CUs x 3-4 resident)

// Tile height of a launch of `segments` = column tiles x grid rows x parameter rows.  The kernels walk a tile's rows one after
// the other, and a CU holds three or four workgroups: a launch of fewer than ~1000 full-height tiles leaves CUs idle or half
// occupied for the time of 32 rows (256 x 256: 8 tiles, 1000 x 1000: 128, 2048 x 2048: 512 -- two per CU, where D5 then takes
// 0.167 ms instead of 0.130).  The height is therefore what gives the launch about 1024 workgroups, between 1 row and half the
// full height; half-height tiles then stay until they number 4096 (a launch of two to four rounds of full-height workgroups ends
// with a ragged last round: 4096 x 4096 is 2.7), and large launches have the full height and its amortisation of the per-tile
// prologue (scripts/tile_rows_probe.py, profiles/r04_experiments.txt section 17).
size_t tile_height(size_t full, size_t segments) {
  return std::min(full, std::max<size_t>({size_t(1), std::min(full / 2, segments / kLaunchWorkgroups), segments / (4 * kLaunchWorkgroups)}));
}

int launch_tiles(inflx_model* m, int op, InflxSweepArgs a, const double* d_params, size_t P, double* d_out, size_t N1, size_t row_count, hipStream_t s,
                 double* d_stats) {
  void* params[] = {&a};
  const size_t gx = (N1 + m->info.tile_cols - 1) / m->info.tile_cols;
  if (gx > 0x7fffffffULL) return fail(INFLX_ERR_SHAPE, "grid rows too long for one launch (%zu column tiles)", gx);
  if (row_count > 0xffffffffULL) return fail(INFLX_ERR_SHAPE, "at most 2^32 grid rows per call (got %zu)", row_count);
  hipFunction_t f = d_stats ? (d_out ? m->tile_stats : m->tile_stats_nostore) : m->tile[op];
  const size_t nu = std::max<size_t>(m->info.n_uniform, 1), nc = std::max<size_t>(m->info.n_col, 1);
  const size_t nr = (std::max<size_t>(m->info.n_row, 1) + 1) & ~size_t(1);  // even stride of a row's values (kNRs of the kernels)
  const TilePlan plan = tile_plan(m, P, N1, row_count);
  const size_t rows_per_launch = plan.rows_per_launch, pbatch = plan.pbatch;
  // (the entry point decided where the tables are built when it chose the stream of the parameter upload: the two must agree)
  hipStream_t tables = m->tables_on_callers_stream ? s : m->side;
  for (size_t p0 = 0; p0 < P; p0 += pbatch) {
    const size_t pb = std::min(pbatch, P - p0);
    a.params = d_params + p0 * m->n_par;
    a.out = d_out ? reinterpret_cast<double*>(reinterpret_cast<char*>(d_out) + p0 * row_count * N1 * kOpBytes[op]) : nullptr;
    a.P = (uint32_t)pb;
    for (size_t r0 = 0; r0 < row_count; r0 += rows_per_launch) {
      const size_t slab = std::min(rows_per_launch, row_count - r0);
      const size_t need = pb * (nu + slab * nr + nc * N1);
      const int b = (int)(m->stage_turn & 1);
      if (need > m->d_stage_tab_cap[b]) {
        // nothing may still be reading the old tables
        if (m->stage_used[b]) HIP_TRY(hipEventSynchronize(m->stage_free[b]));
        if (m->d_stage_tab[b]) HIP_TRY(hipFree(m->d_stage_tab[b]));
        m->d_stage_tab[b] = nullptr;
        m->d_stage_tab_cap[b] = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->d_stage_tab[b]), need * sizeof(double)));
        m->d_stage_tab_cap[b] = need;
      }
      a.row_table = m->d_stage_tab[b];
      a.stream_row0 = (uint32_t)r0;
      a.stream_units = slab;
      // tables on the side stream, as soon as the tile kernel that read this buffer two launches ago is done ...
      if (m->stage_used[b]) HIP_TRY(hipStreamWaitEvent(tables, m->stage_free[b], 0));
      const size_t tx = (slab + m->info.tile_cols - 1) / m->info.tile_cols + (N1 + m->info.tile_cols - 1) / m->info.tile_cols;  // row blocks, then column blocks
      HIP_TRY(hipModuleLaunchKernel(m->stage_tables, (unsigned)tx, (unsigned)pb, 1, m->info.tile_cols, 1, 1, 0, tables, params, nullptr));
      m->stage_turn++;
      // ... and the tile kernel on the caller's stream behind them
      if (tables != s) {
        HIP_TRY(hipEventRecord(m->stage_ready[b], tables));
        HIP_TRY(hipStreamWaitEvent(s, m->stage_ready[b], 0));
      }
      const size_t full = m->info.tile_rows;
      size_t th = tile_height(full, gx * slab * pb);
      // (experiments, scripts/tile_rows_probe.py: a height forced through the environment -- looked at per launch only if the variable
      // existed when the first sweep ran, so that ordinary processes never read the environment while other threads may be writing it)
      static const bool forced = getenv("INFLX_EXPERIMENT_TILE_ROWS") != nullptr;
      if (forced) {
        const char* e = getenv("INFLX_EXPERIMENT_TILE_ROWS");
        const int rows = e ? atoi(e) : 0;
        if (rows > 0) th = std::min<size_t>(full, (size_t)rows);
      }
      a.tile_rows = (uint32_t)th;
      const size_t gy = (slab + th - 1) / th;
      HIP_TRY(probe_begin(m, s));
      HIP_TRY(hipModuleLaunchKernel(f, (unsigned)gx, (unsigned)gy, (unsigned)pb, m->info.tile_cols, 1, 1, 0, s, params, nullptr));
      HIP_TRY(probe_end(m, s));
      HIP_TRY(hipEventRecord(m->stage_free[b], s));
      m->stage_used[b] = true;
    }
  }
  return INFLX_OK;
}

// Enqueue one sweep on `s`; `d_params` points at P parameter rows in device memory.
// `what`: 0 = the whole sweep; for the two-launch broadcast paths 1 = only the evaluation,
// 2 = only the store stream (used to time the dominant kernel on its own).
int launch_grid(inflx_model* m, int op, const double* d_params, size_t P, double* d_out, const double* ss, size_t N0, size_t N1,
                size_t row_begin, size_t row_count, int layout, hipStream_t s, int what = 0, double accuracy = 0.0,
                double* d_stats = nullptr) {
  if (row_count == 0 || N1 == 0) return INFLX_OK;
  if (P > 65535) {
    // grid.z (and grid.y of the evaluation kernels) carries the parameter row: longer parameter axes take several launches
    for (size_t p0 = 0; p0 < P; p0 += 65535) {
      const size_t pb = std::min<size_t>(65535, P - p0);
      double* sub = d_out ? reinterpret_cast<double*>(reinterpret_cast<char*>(d_out) + p0 * row_count * N1 * kOpBytes[op]) : nullptr;
      const int rc = launch_grid(m, op, d_params + p0 * m->n_par, pb, sub, ss, N0, N1, row_begin, row_count, layout, s, what, accuracy, d_stats);
      if (rc) return rc;
    }
    return INFLX_OK;
  }
  InflxSweepArgs a;
  memset(&a, 0, sizeof a);
  a.out = d_out;
  a.params = d_params;
  // convert_ranges (src/anguelova.rs:84-94): spacing = (stop - start) / N, offset = start
  a.x0a = ss[0];
  a.dx0 = (ss[1] - ss[0]) / (double)N0;
  a.x1a = ss[2];
  a.dx1 = (ss[3] - ss[2]) / (double)N1;
  a.N1 = N1;
  a.row_begin = row_begin;
  a.row_count = row_count;
  a.P = (uint32_t)P;
  a.layout = (uint32_t)layout;
  a.col_chunks = 1;
  a.accuracy = accuracy;
  a.stats = d_stats;  // non-NULL: complete_analysis with the running summary (d_out may then be NULL)
  if (takes_row_stream(m, op, layout, P, N1)) return launch_row_stream(m, op, a, d_params, P, d_out, N1, row_count, layout, s, what, d_stats);
  if (takes_col_stream(m, op, layout, P, N1)) return launch_col_stream(m, op, a, d_params, P, d_out, N1, row_count, layout, s, what, d_stats);
  // the flag sweep reads the basis vector, whose axis dependence the out_mask does not describe
  const bool row_uniform = (m->info.out_mask & 2u) == 0 && op != INFLX_OP_QDIF && !(m->call_flags & INFLX_SWEEP_FORCE_TILE);
  if (row_uniform) return launch_rows_fallback(m, op, a, P, N1, row_count, s);
  return launch_tiles(m, op, a, d_params, P, d_out, N1, row_count, s, d_stats);
}

int ensure_chunk(inflx_model* m, int which, size_t bytes) {
  if (bytes <= m->d_chunk_cap[which]) return INFLX_OK;
  if (m->d_chunk[which]) HIP_TRY(hipFree(m->d_chunk[which]));
  m->d_chunk[which] = nullptr;
  m->d_chunk_cap[which] = 0;
  HIP_TRY(hipMalloc(&m->d_chunk[which], bytes));
  m->d_chunk_cap[which] = bytes;
  return INFLX_OK;
}

void say(const char* fmt, ...) {
  // the reference's BADGE_INFO lines (src/lib.rs:53-66, src/anguelova.rs:492,543-547)
  va_list ap;
  va_start(ap, fmt);
  fputs("[Inflatox Info] ", stderr);
  vfprintf(stderr, fmt, ap);
  fputc('\n', stderr);
  fflush(stderr);
  va_end(ap);
}

void warn(const char* fmt, ...) {
  // BADGE_WARN lines (src/lib.rs:58-61)
  va_list ap;
  va_start(ap, fmt);
  fputs("[Inflatox Warning] ", stderr);
  vfprintf(stderr, fmt, ap);
  fputc('\n', stderr);
  fflush(stderr);
  va_end(ap);
}

// a float the way Rust's Display/Debug prints it: NaN, inf, -inf, otherwise `digits` decimals (or %g)
std::string num_text(double v, int digits = -1) {
  if (std::isnan(v)) return "NaN";
  if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
  char buf[64];
  if (digits >= 0) snprintf(buf, sizeof buf, "%.*f", digits, v); else snprintf(buf, sizeof buf, "%.17g", v);
  return buf;
}

std::string list_text(const double* v, size_t n) {
  // Rust's {:.03?} of a Vec<f64>
  std::string s = "[";
  for (size_t k = 0; k < n; ++k) s += (k ? ", " : "") + num_text(v[k], 3);
  return s + "]";
}

double unit_random(uint64_t& state) {
  // splitmix64 -> [0, 1) with 53 random bits; the reference draws from rand::random::<f64>() (thread_rng,
  // unseeded), so no particular sequence is part of its behaviour
  uint64_t z = (state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * 0x1.0p-53;
}

// The orthonormality test of src/lib.rs:164-193 (= :262-285) over the 7-double records of
// inflx_basis_on_points; `failed` counts points at which an inner product was not a normal number.
int check_basis(const double* basis, const double* x, size_t n, double accuracy, size_t* failed) {
  for (size_t k = 0; k < n; ++k) {
    const double* b = basis + 7 * k;
    const std::string point = list_text(x + 2 * k, 2);
    bool encountered_nan = false;
    const int pair[3][2] = {{0, 0}, {0, 1}, {1, 1}};
    for (int q = 0; q < 3; ++q) {
      const int i = pair[q][0], j = pair[q][1];
      const double ip = b[q];
      const std::string vi = list_text(b + 3 + 2 * i, 2), vj = list_text(b + 3 + 2 * j, 2);
      if (i == j) {
        if (!std::isnormal(ip)) {
          warn("Norm of basisvector %d is %s at field-space point %s. v%d=%s\nAre we outside the model's domain?", i,
               num_text(ip).c_str(), point.c_str(), i, vi.c_str());
          encountered_nan = true;
        } else if (std::fabs(ip - 1.) >= accuracy) {
          return fail(INFLX_ERR_BASIS, "Expected basis vector %d to be normalised everywhere in the models domain. Instead, "
                      "found norm %s at %s.", i, num_text(ip).c_str(), point.c_str());
        }
      } else {
        if (!std::isnormal(ip) && ip != 0.0) {
          warn("w%d•w%d = %s at field-space point %s.\nv%d=%s\nv%d=%s\nAre we outside the model's domain?", i, j,
               num_text(ip).c_str(), point.c_str(), i, vi.c_str(), j, vj.c_str());
          encountered_nan = true;
        } else if (std::fabs(ip) >= accuracy) {
          return fail(INFLX_ERR_BASIS, "Expected basis vectors w%d and w%d to be orthogonal everywhere in the model's domain. "
                      "Instead, found inner product %s at %s.", i, j, num_text(ip).c_str(), point.c_str());
        }
      }
    }
    if (encountered_nan) ++*failed;
  }
  return INFLX_OK;
}

}  // namespace

extern "C" {

const char* inflx_last_error(void) { return g_last_error.c_str(); }

int inflx_device_count(int* count) {
  if (!count) return fail(INFLX_ERR_ARG, "count is NULL");
  *count = 0;
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return fail(INFLX_ERR_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  }
  return INFLX_OK;
}

int inflx_open(const char* artefact_path, int device, inflx_model** out) {
  if (!artefact_path || !out) return fail(INFLX_ERR_ARG, "artefact path / output handle is NULL");
  *out = nullptr;
  FILE* fh = fopen(artefact_path, "rb");
  if (!fh) return fail(INFLX_ERR_IO, "could not open model artefact %s: %s", artefact_path, strerror(errno));
  fclose(fh);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(INFLX_ERR_DEVICE, "no HIP device available: the sweep has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(INFLX_ERR_ARG, "device %d out of range (have %d)", device, ndev);
  HIP_TRY(hipSetDevice(device));

  inflx_model* m = new inflx_model();
  m->device = device;
  m->path = artefact_path;
  auto bail = [&](int code) {
    inflx_close(m);
    return code;
  };
  hipError_t e = hipModuleLoad(&m->module, artefact_path);
  if (e != hipSuccess) {
    m->module = nullptr;
    fail(INFLX_ERR_IO, "could not load %s as a gfx950 code object: %s", artefact_path, hipGetErrorString(e));
    return bail(INFLX_ERR_IO);
  }
  int rc;
  if ((rc = read_global(m, "VERSION", m->version, sizeof m->version, true))) return bail(rc);
  if (m->version[0] != kAbiMajor || m->version[1] != kAbiMinor) {
    fail(INFLX_ERR_VERSION, "artefact %s was built for inflatox ABI v%u.%u.%u, this library implements v%u.%u", artefact_path,
         m->version[0], m->version[1], m->version[2], kAbiMajor, kAbiMinor);
    return bail(INFLX_ERR_VERSION);
  }
  if ((rc = read_global(m, "DIM", &m->dim, sizeof m->dim, true))) return bail(rc);
  if ((rc = read_global(m, "N_PARAMETERS", &m->n_par, sizeof m->n_par, true))) return bail(rc);
  char name[512] = {0};
  if ((rc = read_global(m, "MODEL_NAME", name, sizeof name - 1, false))) return bail(rc);
  m->name = name;
  if ((rc = read_global(m, "INFLX_KERNEL_INFO", &m->info, sizeof m->info, true))) return bail(rc);
  if (m->info.kernel_abi != INFLX_KERNEL_ABI) {
    fail(INFLX_ERR_VERSION, "artefact %s uses kernel ABI %u, this library expects %u", artefact_path, m->info.kernel_abi,
         INFLX_KERNEL_ABI);
    return bail(INFLX_ERR_VERSION);
  }
  if ((rc = read_global(m, "USE_GSL", &m->use_gsl, sizeof m->use_gsl, true))) return bail(rc);
  note_sf_word(m, m->module);
  // with GSL linked the reference installs a handler that panics on the first GSL error (src/dylib.rs:141-148): a call fails
  m->sf_policy = m->use_gsl ? INFLX_SF_FAIL : INFLX_SF_QUIET;
  char tag[128] = {0};
  if ((rc = read_global(m, "INFLX_GROUPS", &m->groups, sizeof m->groups, true))) return bail(rc);
  if ((rc = read_global(m, "MODEL_TAG", tag, sizeof tag - 1, false))) return bail(rc);
  m->tag = tag;
  m->groups &= INFLX_GROUP_ALL;
  if (!(m->groups & INFLX_GROUP_CORE)) {
    fail(INFLX_ERR_SYMBOL, "artefact %s is not a model's core object (it carries kernel groups 0x%x only): open the core object and attach this one (inflx_attach)", artefact_path, m->groups);
    m->groups = 0;
    return bail(INFLX_ERR_SYMBOL);
  }
  {
    const uint32_t present = m->groups;
    m->groups = 0;
    if ((rc = resolve_group_kernels(m, m->module, present, artefact_path))) return bail(rc);
    m->groups = present;
  }
  // (experiment, scripts/tables_policy_probe.py: the side stream at the highest priority the device offers)
  int side_priority = 0;
  if (const char* e = getenv("INFLX_EXPERIMENT_SIDE_PRIORITY")) {
    int least = 0, greatest = 0;
    if (atoi(e) != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess) side_priority = greatest;
  }
  if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithPriority(&m->side, hipStreamNonBlocking, side_priority) != hipSuccess ||
      hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking) != hipSuccess) {
    fail(INFLX_ERR_DEVICE, "could not create HIP streams");
    return bail(INFLX_ERR_DEVICE);
  }
  for (int k = 0; k < 2; ++k) {
    if (hipEventCreateWithFlags(&m->chunk_done[k], hipEventDisableTiming) != hipSuccess ||
        // ordering-only events between two streams of this device: no system-scope fence needed (the
        // kernels' own agent-scope release/acquire at their boundaries publishes the row table)
        hipEventCreateWithFlags(&m->table_ready[k], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess ||
        hipEventCreateWithFlags(&m->table_free[k], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess ||
        hipEventCreateWithFlags(&m->copy_done[k], hipEventDisableTiming) != hipSuccess) {
      fail(INFLX_ERR_DEVICE, "could not create HIP events");
      return bail(INFLX_ERR_DEVICE);
    }
  }
  for (int k = 0; k < 2; ++k) {
    if (hipEventCreateWithFlags(&m->stage_ready[k], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess ||
        hipEventCreateWithFlags(&m->stage_free[k], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
      fail(INFLX_ERR_DEVICE, "could not create HIP events");
      return bail(INFLX_ERR_DEVICE);
    }
  }
  if (hipEventCreate(&m->t0) != hipSuccess || hipEventCreate(&m->t1) != hipSuccess) {
    fail(INFLX_ERR_DEVICE, "could not create HIP timing events");
    return bail(INFLX_ERR_DEVICE);
  }
  *out = m;
  return INFLX_OK;
}

void inflx_close(inflx_model* m) {
  if (!m) return;
  { std::lock_guard<std::recursive_mutex> last_call_has_returned(m->mu); }  // (closing a handle another thread still uses remains the caller's error)
  (void)hipSetDevice(m->device);
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  if (m->side) (void)hipStreamSynchronize(m->side);
  if (m->copy_stream) (void)hipStreamSynchronize(m->copy_stream);
  for (int k = 0; k < 2; ++k) {
    if (m->d_chunk[k]) (void)hipFree(m->d_chunk[k]);
    if (m->chunk_done[k]) (void)hipEventDestroy(m->chunk_done[k]);
    if (m->copy_done[k]) (void)hipEventDestroy(m->copy_done[k]);
    if (m->table_ready[k]) (void)hipEventDestroy(m->table_ready[k]);
    if (m->table_free[k]) (void)hipEventDestroy(m->table_free[k]);
    if (m->d_row_table[k]) (void)hipFree(m->d_row_table[k]);
  }
  for (int k = 0; k < 2; ++k) {
    if (m->stage_ready[k]) (void)hipEventDestroy(m->stage_ready[k]);
    if (m->stage_free[k]) (void)hipEventDestroy(m->stage_free[k]);
    if (m->d_stage_tab[k]) (void)hipFree(m->d_stage_tab[k]);
  }
  if (m->t0) (void)hipEventDestroy(m->t0);
  if (m->t1) (void)hipEventDestroy(m->t1);
  if (m->sf_done) (void)hipEventDestroy(m->sf_done);
  for (auto& slot : m->pslot) {
    if (slot.dev) (void)hipFree(slot.dev);
    if (slot.host) (void)hipHostFree(slot.host);
    if (slot.last_use) (void)hipEventDestroy(slot.last_use);
  }
  if (m->d_stats) (void)hipFree(m->d_stats);
  if (m->d_whole) (void)hipFree(m->d_whole);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  if (m->side) (void)hipStreamDestroy(m->side);
  if (m->copy_stream) (void)hipStreamDestroy(m->copy_stream);
  for (hipModule_t extra : m->attached) (void)hipModuleUnload(extra);
  if (m->module) (void)hipModuleUnload(m->module);
  delete m;
}

int inflx_attach(inflx_model* m, const char* path) {
  if (!m || !path) return fail(INFLX_ERR_ARG, "model handle / path is NULL");
  INFLX_SERIALISE(m);
  FILE* fh = fopen(path, "rb");
  if (!fh) return fail(INFLX_ERR_IO, "could not open code object %s: %s", path, strerror(errno));
  fclose(fh);
  HIP_TRY(hipSetDevice(m->device));
  return attach_object(m, path);
}

uint32_t inflx_groups(const inflx_model* m) { return m ? m->groups : 0; }

uint32_t inflx_n_fields(const inflx_model* m) { return m ? m->dim : 0; }
uint32_t inflx_n_parameters(const inflx_model* m) { return m ? m->n_par : 0; }
const char* inflx_model_name(const inflx_model* m) { return m ? m->name.c_str() : ""; }
int inflx_device_of(const inflx_model* m) { return m ? m->device : -1; }

int inflx_host_threads(unsigned devices_at_work, unsigned out[3]) {
  if (!out) return fail(INFLX_ERR_ARG, "output array is NULL");
  const unsigned saved = tl_sharers;
  tl_sharers = std::max(1u, devices_at_work);
  out[0] = host_cpu_budget();
  out[1] = prefault_threads();  // (through the same functions the pipelines ask: environment overrides included)
  out[2] = host_fill_threads();
  tl_sharers = saved;
  return INFLX_OK;
}

int inflx_stage_info(const inflx_model* m, uint32_t* nu, uint32_t* nr, uint32_t* nc, uint32_t* out_mask) {
  if (!m) return fail(INFLX_ERR_ARG, "model handle is NULL");
  if (nu) *nu = m->info.n_uniform;
  if (nr) *nr = m->info.n_row;
  if (nc) *nc = m->info.n_col;
  if (out_mask) *out_mask = m->info.out_mask;
  return INFLX_OK;
}

int inflx_sweep_plan(const inflx_model* m, int op, size_t P, size_t N1, size_t row_count, int layout, uint32_t plan[4]) {
  return inflx_sweep_plan_ex(m, op, P, N1, row_count, layout, INFLX_SWEEP_DEFAULT, plan);
}

int inflx_sweep_plan_ex(const inflx_model* cm, int op, size_t P, size_t N1, size_t row_count, int layout, unsigned flags, uint32_t plan[4]) {
  if (!cm || !plan) return fail(INFLX_ERR_ARG, "model handle / plan array is NULL");
  if (op < 0 || op >= INFLX_OP_COUNT) return fail(INFLX_ERR_ARG, "unknown sweep operation %d", op);
  if (flags & ~(unsigned)INFLX_SWEEP_FORCE_TILE) return fail(INFLX_ERR_ARG, "unknown sweep flags 0x%x", flags);
  inflx_model* m = const_cast<inflx_model*>(cm);  // (the flags live in the handle for the duration of the call)
  INFLX_SERIALISE(m);
  FlagScope scope(m, flags);
  plan[0] = plan[1] = plan[2] = plan[3] = 0;
  if (takes_row_stream(m, op, layout, P, N1)) {
    const RowStreamPlan r = row_stream_plan(m, op, layout, P, N1, row_count);
    plan[0] = INFLX_PATH_ROW_STREAM;
    plan[1] = (uint32_t)r.batch;
    plan[2] = (uint32_t)((P + r.batch - 1) / r.batch);
    plan[3] = (uint32_t)r.replicas;
  } else if (takes_col_stream(m, op, layout, P, N1)) {
    plan[0] = INFLX_PATH_COL_STREAM;
  } else if ((m->info.out_mask & 2u) == 0 && op != INFLX_OP_QDIF && !(m->call_flags & INFLX_SWEEP_FORCE_TILE)) {
    plan[0] = INFLX_PATH_ROWS;
  } else {
    plan[0] = INFLX_PATH_TILE;
    const TilePlan t = tile_plan(m, P, N1, row_count);
    const size_t gx = (N1 + m->info.tile_cols - 1) / m->info.tile_cols, slab = std::min(t.rows_per_launch, row_count);
    plan[1] = (uint32_t)t.pbatch;
    plan[2] = (uint32_t)(((P + t.pbatch - 1) / t.pbatch) * ((row_count + t.rows_per_launch - 1) / t.rows_per_launch));
    plan[3] = (uint32_t)tile_height(m->info.tile_rows, gx * slab * t.pbatch);
  }
  return INFLX_OK;
}

int inflx_sweep_device_stats(inflx_model* m, const double* p, size_t P, size_t n_p, void* d_out, size_t d_out_bytes, const double* ss,
                             size_t N0, size_t N1, size_t row_begin, size_t row_count, void* stream, inflx_summary* summary) {
  INFLX_SERIALISE(m);
  const int op = INFLX_OP_COMPLETE;
  int rc = validate(m, op, p, P, n_p);
  if (rc) return rc;
  if ((rc = need_groups(m, INFLX_GROUP_STATS))) return rc;
  if (!ss || !summary) return fail(INFLX_ERR_ARG, "start_stop / summary pointer is NULL");
  if (row_begin + row_count > N0) return fail(INFLX_ERR_SHAPE, "rows [%zu,%zu) exceed the grid (N0 = %zu)", row_begin, row_begin + row_count, N0);
  if (d_out && d_out_bytes < P * row_count * N1 * kOpBytes[op])
    return fail(INFLX_ERR_SHAPE, "output buffer has %zu bytes, the sweep writes %zu", d_out_bytes, P * row_count * N1 * kOpBytes[op]);
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m->stream;
  // `up`: the stream of the first kernel of the sweep (parameters and the initial summary are ordered before it);
  // `eval`: the stream of the kernels that accumulate (the per-row / per-column evaluation of the broadcast paths on the
  // side stream, the tile kernels on the caller's)
  hipStream_t up = evaluates_on_side_stream(m, op, INFLX_AOS, P, N1, row_count) ? m->side : s;
  hipStream_t eval = last_reader_is_callers_stream(m, op, INFLX_AOS, P, N1) ? s : m->side;
  if (!m->d_stats) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->d_stats), 18 * sizeof(double)));
  inflx_summary init;
  for (int k = 0; k < 6; ++k) {
    init.min[k] = HUGE_VAL;
    init.max[k] = -HUGE_VAL;
    init.count[k] = 0;
  }
  static_assert(sizeof(inflx_summary) == 18 * 8, "inflx_summary must match the device layout");
  HIP_TRY(hipMemcpyAsync(m->d_stats, &init, sizeof init, hipMemcpyHostToDevice, up));
  const double* d_params = nullptr;
  if ((rc = acquire_params(m, p, P * n_p, up, &d_params))) return rc;
  if (row_count && N1) {
    rc = launch_grid(m, op, d_params, P, static_cast<double*>(d_out), ss, N0, N1, row_begin, row_count, INFLX_AOS, s, 0, 0.0, m->d_stats);
  }
  if ((rc = release_params_after(m, eval, rc))) return rc;
  HIP_TRY(hipMemcpyAsync(summary, m->d_stats, sizeof *summary, hipMemcpyDeviceToHost, eval));
  HIP_TRY(hipStreamSynchronize(eval));
  if (eval != s) HIP_TRY(hipStreamSynchronize(s));
  return sf_verdict(m);
}

int inflx_synchronize(inflx_model* m) {
  INFLX_SERIALISE(m);
  if (!m) return fail(INFLX_ERR_ARG, "model handle is NULL");
  HIP_TRY(hipSetDevice(m->device));
  HIP_TRY(hipStreamSynchronize(m->side));
  HIP_TRY(hipStreamSynchronize(m->stream));
  return sf_verdict(m);  // the device-resident sweeps are asynchronous: this is where they report
}

int inflx_sf_status(inflx_model* m, unsigned* bits, int clear) {
  if (!m || !bits) return fail(INFLX_ERR_ARG, "model handle / status pointer is NULL");
  INFLX_SERIALISE(m);
  return sf_collect(m, clear != 0, bits);
}

int inflx_sf_policy(inflx_model* m, int policy) {
  if (!m) return fail(INFLX_ERR_ARG, "model handle is NULL");
  if (policy != INFLX_SF_QUIET && policy != INFLX_SF_FAIL) return fail(INFLX_ERR_ARG, "unknown special-function policy %d", policy);
  INFLX_SERIALISE(m);
  m->sf_policy = policy;
  return INFLX_OK;
}

int inflx_uses_gsl(const inflx_model* m) { return m ? (int)m->use_gsl : 0; }

int inflx_sweep_device(inflx_model* m, int op, const double* p, size_t P, size_t n_p, void* d_out, size_t d_out_bytes,
                       const double* ss, size_t N0, size_t N1, size_t row_begin, size_t row_count, int layout, void* stream) {
  return inflx_sweep_device_ex(m, op, p, P, n_p, d_out, d_out_bytes, ss, N0, N1, row_begin, row_count, layout, stream, INFLX_SWEEP_DEFAULT);
}

int inflx_sweep_device_ex(inflx_model* m, int op, const double* p, size_t P, size_t n_p, void* d_out, size_t d_out_bytes,
                          const double* ss, size_t N0, size_t N1, size_t row_begin, size_t row_count, int layout, void* stream, unsigned flags) {
  INFLX_SERIALISE(m);
  int rc = validate(m, op, p, P, n_p);
  if (rc) return rc;
  if (!d_out || !ss) return fail(INFLX_ERR_ARG, "output / start_stop pointer is NULL");
  if (layout != INFLX_AOS && layout != INFLX_SOA) return fail(INFLX_ERR_ARG, "unknown layout %d", layout);
  if (flags & ~(unsigned)INFLX_SWEEP_FORCE_TILE) return fail(INFLX_ERR_ARG, "unknown sweep flags 0x%x", flags);
  if (row_begin + row_count > N0) return fail(INFLX_ERR_SHAPE, "rows [%zu,%zu) exceed the grid (N0 = %zu)", row_begin, row_begin + row_count, N0);
  if (op == INFLX_OP_QDIF) return fail(INFLX_ERR_ARG, "the flag sweep has a byte result: use inflx_flag_quantum_dif");
  const size_t need = P * row_count * N1 * kOpBytes[op];
  if (d_out_bytes < need) return fail(INFLX_ERR_SHAPE, "output buffer has %zu bytes, the sweep writes %zu", d_out_bytes, need);
  HIP_TRY(hipSetDevice(m->device));
  FlagScope scope(m, flags);
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m->stream;
  // the parameters are read by the kernel that evaluates the model: the per-row / per-column evaluation or the stage tables -- on the
  // side stream while an earlier sweep of this handle is still running (they overlap it), on the caller's stream at an idle handle
  hipStream_t up = evaluates_on_side_stream(m, op, layout, P, N1, row_count, /*alone=*/handle_idle(m)) ? m->side : s;
  const double* d_params = nullptr;
  if ((rc = acquire_params(m, p, P * n_p, up, &d_params))) return rc;
  rc = launch_grid(m, op, d_params, P, static_cast<double*>(d_out), ss, N0, N1, row_begin, row_count, layout, s);
  rc = release_params_after(m, last_reader_is_callers_stream(m, op, layout, P, N1) ? s : m->side, rc);
  if (rc == INFLX_OK && !m->sf_words.empty() && s != m->stream) {  // (see sf_done: the stream that finishes the sweep is the caller's)
    if (!m->sf_done) HIP_TRY(hipEventCreateWithFlags(&m->sf_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(m->sf_done, s));
    m->sf_pending = true;
  }
  return rc;
}

int inflx_sweep_device_timed(inflx_model* m, int op, const double* p, size_t P, size_t n_p, void* d_out, size_t d_out_bytes,
                             const double* ss, size_t N0, size_t N1, size_t row_begin, size_t row_count, int layout, void* stream,
                             int repeats, int dominant_only, float* ms_per_launch) {
  return inflx_sweep_device_timed_ex(m, op, p, P, n_p, d_out, d_out_bytes, ss, N0, N1, row_begin, row_count, layout, stream, repeats, dominant_only,
                                     INFLX_SWEEP_DEFAULT, ms_per_launch);
}

int inflx_sweep_device_timed_ex(inflx_model* m, int op, const double* p, size_t P, size_t n_p, void* d_out, size_t d_out_bytes,
                                const double* ss, size_t N0, size_t N1, size_t row_begin, size_t row_count, int layout, void* stream,
                                int repeats, int mode, unsigned flags, float* ms_per_launch) {
  INFLX_SERIALISE(m);
  if (repeats <= 0 || !ms_per_launch) return fail(INFLX_ERR_ARG, "repeats must be positive and ms_per_launch non-NULL");
  if (mode < INFLX_TIME_BACK_TO_BACK || mode > INFLX_TIME_SINGLE_CALL) return fail(INFLX_ERR_ARG, "unknown timing mode %d", mode);
  // first call validates everything and uploads the parameters
  int rc = inflx_sweep_device_ex(m, op, p, P, n_p, d_out, d_out_bytes, ss, N0, N1, row_begin, row_count, layout, stream, flags);
  if (rc) return rc;
  FlagScope scope(m, flags);
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m->stream;
  if (mode == INFLX_TIME_SINGLE_CALL) {
    // what ONE call costs a caller whose handle is idle: every repetition waits until the device has nothing of this handle left,
    // then brackets one whole inflx_sweep_device call -- tables (or per-row values) AND the sweep kernel -- with two events
    float total = 0.f;
    for (int k = 0; k < repeats; ++k) {
      HIP_TRY(hipStreamSynchronize(m->side));
      HIP_TRY(hipStreamSynchronize(s));
      HIP_TRY(hipEventRecord(m->t0, s));
      rc = inflx_sweep_device_ex(m, op, p, P, n_p, d_out, d_out_bytes, ss, N0, N1, row_begin, row_count, layout, stream, flags);
      if (rc) return rc;
      HIP_TRY(hipEventRecord(m->t1, s));
      HIP_TRY(hipEventSynchronize(m->t1));
      float one = 0.f;
      HIP_TRY(hipEventElapsedTime(&one, m->t0, m->t1));
      total += one;
    }
    HIP_TRY(hipStreamSynchronize(m->side));
    *ms_per_launch = total / (float)repeats;
    return INFLX_OK;
  }
  const int dominant_only = mode;
  hipStream_t reader = last_reader_is_callers_stream(m, op, layout, P, N1) ? s : m->side;
  const double* d_params = m->pslot[m->pcur].dev;  // what the call above uploaded (or found in place)
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipStreamSynchronize(m->side));
  // the repetitions are enqueued back to back: the shape of a sweep that arrives while its predecessor runs (tables on the side stream)
  (void)evaluates_on_side_stream(m, op, layout, P, N1, row_count, /*alone=*/false);
  // dominant_only == 2: the full sweeps, with an event pair around every dominant-kernel launch (at most 64 per sweep)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;
  struct Disarm {
    inflx_model* m;
    std::vector<std::pair<hipEvent_t, hipEvent_t>>& pairs;
    ~Disarm() {
      m->probe = nullptr;
      for (auto& pr : pairs) {
        if (pr.first) (void)hipEventDestroy(pr.first);
        if (pr.second) (void)hipEventDestroy(pr.second);
      }
    }
  } disarm{m, pairs};
  if (dominant_only == 2) {
    pairs.assign((size_t)repeats * 64, {nullptr, nullptr});
    for (auto& pr : pairs) {
      HIP_TRY(hipEventCreate(&pr.first));
      HIP_TRY(hipEventCreate(&pr.second));
    }
    m->probe = &pairs;
    m->probe_used = 0;
    m->probe_overflow = false;
  }
  HIP_TRY(hipEventRecord(m->t0, s));
  for (int k = 0; k < repeats; ++k) {
    rc = launch_grid(m, op, d_params, P, static_cast<double*>(d_out), ss, N0, N1, row_begin, row_count, layout, s,
                     dominant_only == 1 ? 2 : 0);
    if (rc) return release_params_after(m, reader, rc);
  }
  HIP_TRY(hipEventRecord(m->t1, s));
  if ((rc = release_params(m, reader))) return rc;
  HIP_TRY(hipEventSynchronize(m->t1));
  float ms = 0.f;
  if (dominant_only == 2) {
    if (m->probe_used == 0) return fail(INFLX_ERR_ARG, "in-pipeline timing: this sweep shape has no probed kernel");
    if (m->probe_overflow) return fail(INFLX_ERR_ARG, "in-pipeline timing: more than 64 launches of the dominant kernel per sweep (use fewer parameter rows per call)");
    for (size_t k = 0; k < m->probe_used; ++k) {
      float one = 0.f;
      HIP_TRY(hipEventElapsedTime(&one, pairs[k].first, pairs[k].second));
      ms += one;
    }
  } else {
    HIP_TRY(hipEventElapsedTime(&ms, m->t0, m->t1));
  }
  *ms_per_launch = ms / (float)repeats;
  return INFLX_OK;
}

int inflx_basis_on_points(inflx_model* m, const double* p, size_t n_p, const double* x, size_t n, double* out) {
  INFLX_SERIALISE(m);
  int rc = validate(m, INFLX_OP_RAW, p, 1, n_p, /*kernels=*/false);  // (the operation stands for "any": inflx_basis_points is part of the core object)
  if (rc) return rc;
  if (n == 0) return INFLX_OK;
  if (!x || !out) return fail(INFLX_ERR_ARG, "point / output pointer is NULL");
  HIP_TRY(hipSetDevice(m->device));
  const size_t in_bytes = n * 2 * sizeof(double), out_bytes = n * 7 * sizeof(double);
  if ((rc = ensure_chunk(m, 0, out_bytes))) return rc;
  if ((rc = ensure_chunk(m, 1, in_bytes))) return rc;
  const double* d_params = nullptr;
  if ((rc = acquire_params(m, p, n_p, m->stream, &d_params))) return rc;
  HIP_TRY(hipMemcpyAsync(m->d_chunk[1], x, in_bytes, hipMemcpyHostToDevice, m->stream));
  InflxTrajectoryArgs a;
  memset(&a, 0, sizeof a);
  a.out = static_cast<double*>(m->d_chunk[0]);
  a.params = d_params;
  a.points = static_cast<const double*>(m->d_chunk[1]);
  a.n = n;
  a.P = 1;
  void* params[] = {&a};
  const size_t gx = (n + m->info.tile_cols - 1) / m->info.tile_cols;
  HIP_TRY(hipModuleLaunchKernel(m->basis_points, (unsigned)gx, 1, 1, m->info.tile_cols, 1, 1, 0, m->stream, params, nullptr));
  HIP_TRY(hipMemcpyAsync(out, m->d_chunk[0], out_bytes, hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  // The basis checks judge the values themselves (a NaN fails `is_normal`, src/lib.rs:176-205): what the special functions noted at
  // these probe points is dropped here, so that it is not reported by the next sweep.
  unsigned dropped = 0;
  return sf_collect(m, true, &dropped);
}

int inflx_ops_on_values(inflx_model* m, const double* values, size_t n, double* out, int ieee_only) {
  INFLX_SERIALISE(m);
  if (!m) return fail(INFLX_ERR_ARG, "model handle is NULL");
  if (n == 0) return INFLX_OK;
  if (!values || !out) return fail(INFLX_ERR_ARG, "values / output pointer is NULL");
  HIP_TRY(hipSetDevice(m->device));
  {
    const int rcg = need_groups(m, INFLX_GROUP_VALUES);
    if (rcg) return rcg;
  }
  const size_t in_bytes = n * 5 * sizeof(double), out_bytes = n * 9 * sizeof(double);
  int rc;
  if ((rc = ensure_chunk(m, 0, out_bytes))) return rc;
  if ((rc = ensure_chunk(m, 1, in_bytes))) return rc;
  HIP_TRY(hipMemcpyAsync(m->d_chunk[1], values, in_bytes, hipMemcpyHostToDevice, m->stream));
  InflxTrajectoryArgs a;
  memset(&a, 0, sizeof a);
  a.out = static_cast<double*>(m->d_chunk[0]);
  a.points = static_cast<const double*>(m->d_chunk[1]);
  a.n = n;
  a.P = 1;
  a.reserved = ieee_only ? 1u : 0u;
  void* params[] = {&a};
  const size_t gx = (n + m->info.tile_cols - 1) / m->info.tile_cols;
  if (gx > 0x7fffffffULL) return fail(INFLX_ERR_SHAPE, "too many records for one launch");
  HIP_TRY(hipModuleLaunchKernel(m->ops_on_values, (unsigned)gx, 1, 1, m->info.tile_cols, 1, 1, 0, m->stream, params, nullptr));
  HIP_TRY(hipMemcpyAsync(out, m->d_chunk[0], out_bytes, hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  return INFLX_OK;
}

int inflx_validate_basis_at_random(inflx_model* m, uint64_t seed) {
  INFLX_SERIALISE(m);
  if (!m) return fail(INFLX_ERR_ARG, "model handle is NULL");
  // src/lib.rs:142-162: one random parameter vector in [-10, 10), 100 random points in [-1, 1)^2
  const size_t num_points = 100;
  uint64_t state = seed ? seed : (((uint64_t)std::random_device{}() << 32) ^ std::random_device{}());
  std::vector<double> p(m->n_par), x(2 * num_points);
  for (double& v : p) v = 10. * (-1. + 2. * unit_random(state));
  for (double& v : x) v = -1. + 2. * unit_random(state);
  std::vector<double> basis(7 * num_points);
  int rc = inflx_basis_on_points(m, p.data(), p.size(), x.data(), num_points, basis.data());
  if (rc) return rc;
  size_t failed = 0;
  if ((rc = check_basis(basis.data(), x.data(), num_points, 1e-3, &failed))) return rc;
  if (failed)
    warn("Inflatox was unable to verify basis orthonormality at %zu out of %zu tested points.\nThis could be indicative of a "
         "defective model.\nUsed parameter values: p=%s", failed, num_points, list_text(p.data(), p.size()).c_str());
  return INFLX_OK;
}

int inflx_validate_basis_on_domain(inflx_model* m, const uint32_t* num_points, size_t n_axes, const double* p, size_t n_p,
                                   const double* ss, double accuracy) {
  INFLX_SERIALISE(m);
  if (!m) return fail(INFLX_ERR_ARG, "model handle is NULL");
  if (!num_points || !ss) return fail(INFLX_ERR_ARG, "num_points / start_stop array is NULL");
  say("Validating basis orthonormality on specified domain. This may take a while...");
  if (n_axes != m->dim)
    return fail(INFLX_ERR_SHAPE, "expected an array with with the same number of axes as there are field-space coordinates "
                "(model has %u fields, got %zu)", m->dim, n_axes);
  int rc = validate(m, INFLX_OP_RAW, p, 1, n_p, /*kernels=*/false);
  if (rc) return rc;
  // src/lib.rs:247-256: every axis in turn is walked from the START corner of the other axis; the walk
  // itself begins at that axis' STOP value (`stop + spacing * idx`) -- reproduced as the reference has it
  size_t failed = 0;
  double tested = 1.;
  for (size_t k = 0; k < n_axes; ++k) tested *= (double)num_points[k];
  for (size_t axis = 0; axis < n_axes; ++axis) {
    const size_t n = num_points[axis];
    const double start = ss[2 * axis], stop = ss[2 * axis + 1];
    const double spacing = (stop - start) / (double)n;
    std::vector<double> x(2 * n), basis(7 * n);
    for (size_t idx = 0; idx < n; ++idx) {
      x[2 * idx] = ss[0];
      x[2 * idx + 1] = ss[2];
      x[2 * idx + axis] = stop + spacing * (double)idx;
    }
    if ((rc = inflx_basis_on_points(m, p, n_p, x.data(), n, basis.data()))) return rc;
    if ((rc = check_basis(basis.data(), x.data(), n, accuracy, &failed))) return rc;
    if (failed)
      warn("Inflatox was unable to verify basis orthonormality at %zu out of %g tested points.\nThis could be indicative of a "
           "defective model.\nUsed parameter values: p=%s", failed, tested, list_text(p, n_p).c_str());
  }
  return INFLX_OK;
}

}  // extern "C"

namespace {

// Where a slab lands in the caller's array.  The plain entry points hand over an array that holds exactly the slab
// (dst_rows = row_count, dst_row0 = 0); a multi-device sweep hands every device the WHOLE array and the place of its
// slab in it: element (p, k, row r of the slab) lives in block (p[, k]) of `dst_rows` grid rows at row dst_row0 + r.
struct HostDest {
  size_t dst_rows = 0;
  size_t dst_row0 = 0;
  // plane subset of a planes (SoA) result: only planes [plane0, plane0 + planes) of every parameter row are copied to the host, into a
  // destination of `planes` planes per parameter row (planes = 0: all of them).  The device evaluates every plane -- the model's values
  // come out of one evaluation -- but calc_V_array wants one of five and the copy is what a host-result call pays for.
  size_t plane0 = 0;
  size_t planes = 0;
};

// Bytes that have arrived in the caller's array, for the progress lines of long host-result calls (shared by the
// devices of a multi-device sweep).
struct Progress {
  std::atomic<uint64_t> done{0};
  uint64_t total = 0;
};

// One contiguous piece of the device-to-host transfer.
struct Span {
  size_t src, dst, bytes;  // byte offsets into the device buffer / the caller's array
};

// The spans that carry a device buffer laid out as the slab ([P][rc][N1][K] or [P][K][rc][N1], one byte per point for the
// flag sweep) into the caller's array, merged where both sides are contiguous.
std::vector<Span> transfer_spans(int op, size_t P, size_t N1, size_t row_count, int layout, const HostDest& d) {
  const size_t K = kOpWidth[op];
  const bool planes = layout == INFLX_SOA && K > 1;
  const size_t blocks = planes ? P * K : P;                              // contiguous blocks of rows on both sides
  const size_t row_bytes = planes ? N1 * sizeof(double) : N1 * kOpBytes[op];
  std::vector<Span> spans;
  const bool subset = planes && d.planes != 0;
  for (size_t b = 0; b < blocks; ++b) {
    size_t dst_block = b;
    if (subset) {
      const size_t pr = b / K, k = b % K;
      if (k < d.plane0 || k >= d.plane0 + d.planes) continue;
      dst_block = pr * d.planes + (k - d.plane0);
    }
    const Span sp{b * row_count * row_bytes, (dst_block * d.dst_rows + d.dst_row0) * row_bytes, row_count * row_bytes};
    if (!spans.empty() && spans.back().src + spans.back().bytes == sp.src && spans.back().dst + spans.back().bytes == sp.dst)
      spans.back().bytes += sp.bytes;
    else
      spans.push_back(sp);
  }
  return spans;
}

// ---- host-side broadcast of results that are constant along a grid axis ------------------------------------------------
// A model none of whose values depends on x[1] (the README hyperbolic model) has a result in which every grid row is N1
// copies of one record; one that ignores x[0] has a result whose grid rows are all the same image.  Such a result need not
// cross PCIe: the one evaluated line does (hyperbolic 8192^2: 393 kB instead of 3.2 GB), and host threads write the
// caller's array from it with streaming stores -- the same bytes in the same writable array (the reference's contract,
// consistency_conditions.py:301-308), at the host's memory bandwidth instead of the link's.
// INFLX_HOST_FILL=0 switches the path off (everything then goes through the device-to-host copy), INFLX_HOST_FILL_THREADS
// sets the number of filling threads, INFLX_HOST_FILL_MIN_MB the smallest result that takes the path.
bool host_fill_enabled() {
  static const bool v = [] {
    const char* e = getenv("INFLX_HOST_FILL");
    return !(e && atoi(e) == 0);
  }();
  return v;
}
size_t host_fill_min_bytes() {
  static const size_t v = [] {
    const char* e = getenv("INFLX_HOST_FILL_MIN_MB");
    const long mb = e ? atol(e) : -1;
    return (size_t)(mb >= 0 ? mb : 8) << 20;
  }();
  return v;
}

// dst[0 .. n*rec_bytes) = n copies of the record at `rec` (rec_bytes a multiple of 8, at most 64): streaming 16-byte stores
// from a small periodic pattern, so that the freshly written array does not pass through the caches of the filling core
void repeat_record(char* dst, const char* rec, size_t rec_bytes, size_t n) {
  const size_t bytes = n * rec_bytes;
#if defined(__x86_64__)
  // pattern: 16 records (a multiple of 16 bytes whatever rec_bytes) and one more stretch, so that 64 bytes can be read at any phase
  alignas(64) char pat[16 * 64 + 128];
  const size_t period = 16 * rec_bytes;
  for (size_t off = 0; off < period + 64; off += rec_bytes) memcpy(pat + off, rec, rec_bytes);
  size_t done = 0;
  const size_t head = std::min(bytes, (size_t)((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15));
  if (head) {
    memcpy(dst, pat, head);
    done = head;
  }
  size_t phase = done % period;
  for (; done + 64 <= bytes; done += 64) {
    const char* src = pat + phase;
    __m128d a, b, c, d;
    memcpy(&a, src, 16), memcpy(&b, src + 16, 16), memcpy(&c, src + 32, 16), memcpy(&d, src + 48, 16);
    _mm_stream_pd(reinterpret_cast<double*>(dst + done), a);
    _mm_stream_pd(reinterpret_cast<double*>(dst + done + 16), b);
    _mm_stream_pd(reinterpret_cast<double*>(dst + done + 32), c);
    _mm_stream_pd(reinterpret_cast<double*>(dst + done + 48), d);
    phase += 64;
    if (phase >= period) phase -= period;
  }
  for (; done < bytes; ++done) {  // (< 64 bytes)
    dst[done] = pat[phase];
    if (++phase == period) phase = 0;
  }
#else
  for (size_t k = 0; k < n; ++k) memcpy(dst + k * rec_bytes, rec, rec_bytes);
#endif
}

// dst[0 .. bytes) = src[0 .. bytes) with streaming stores (the row image of a column-only model into one grid row)
void stream_copy(char* dst, const char* src, size_t bytes) {
#if defined(__x86_64__)
  size_t done = std::min(bytes, (size_t)((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15));
  if (done) memcpy(dst, src, done);
  for (; done + 16 <= bytes; done += 16) {
    __m128d a;
    memcpy(&a, src + done, 16);
    _mm_stream_pd(reinterpret_cast<double*>(dst + done), a);
  }
  if (done < bytes) memcpy(dst + done, src + done, bytes - done);
#else
  memcpy(dst, src, bytes);
#endif
}

// Run fill(first, last) over [0, n) in contiguous blocks on up to host_fill_threads() threads (the calling thread takes one).
template <typename F>
void parallel_blocks(size_t n, size_t bytes_per_item, F fill) {
  if (n == 0) return;
  size_t nthreads = std::min<size_t>(host_fill_threads(), n);
  // a thread costs tens of microseconds to start: up to 8 threads get at least 4 MiB each, further ones 16 MiB (hyperbolic on a
  // GPU box's 16 CPUs: 1000^2 = 48 MB fastest with 8 threads, 2048^2 = 201 MB with 12 -- 32 take half as long again --, 8192^2 with 32)
  const size_t bytes = n * bytes_per_item;
  nthreads = std::min<size_t>(nthreads, std::max<size_t>({size_t(1), bytes / (size_t(16) << 20), std::min<size_t>(8, bytes / (size_t(4) << 20))}));
  std::vector<std::thread> pool;
  auto block = [&](size_t t) {
    const size_t a = n * t / nthreads, b = n * (t + 1) / nthreads;
    if (b > a) fill(a, b);
#if defined(__x86_64__)
    _mm_sfence();
#endif
  };
  std::vector<size_t> here{0};
  for (size_t t = 1; t < nthreads; ++t) {
    try {
      pool.emplace_back(block, t);
    } catch (...) {
      here.push_back(t);
    }
  }
  for (size_t t : here) block(t);
  for (auto& th : pool) th.join();
}

int sweep_host_body(inflx_model* m, int op, const double* p, size_t P, size_t n_p, void* out_v, const double* ss, size_t N0,
                    size_t N1, size_t row_begin, size_t row_count, int layout, double accuracy, HostDest dest, Progress* progress);

// The broadcast form of a host-result sweep (see above); `axis` = 1: nothing depends on x[1], 0: nothing depends on x[0].
int sweep_host_broadcast(inflx_model* m, int op, int axis, const double* p, size_t P, size_t n_p, char* out, const double* ss, size_t N0,
                         size_t N1, size_t row_begin, size_t row_count, int layout, const HostDest& dest, Progress* progress) {
  const size_t K = kOpWidth[op];
  const bool planes = layout == INFLX_SOA && K > 1;
  const size_t rec = planes ? sizeof(double) : K * sizeof(double);  // bytes per grid point within one block of rows
  const size_t blocks = planes ? P * K : P;
  const size_t dst_row_bytes = N1 * rec;
  for (size_t b = 0; b < blocks; ++b) advise_huge_pages(out + ((b * dest.dst_rows + dest.dst_row0) * N1) * rec, row_count * dst_row_bytes);
  int rc;
  if (axis == 1) {
    // one column of the grid: (P, row_count, 1, K) resp. (P, K, row_count, 1)
    std::vector<double> line(P * row_count * K);
    if ((rc = sweep_host_body(m, op, p, P, n_p, line.data(), ss, N0, 1, row_begin, row_count, layout, 0.0, HostDest(), nullptr))) return rc;
    const char* src = reinterpret_cast<const char*>(line.data());
    parallel_blocks(blocks * row_count, dst_row_bytes, [&](size_t first, size_t last) {
      for (size_t q = first; q < last; ++q) {
        const size_t b = q / row_count, r = q % row_count;
        char* dst = out + ((b * dest.dst_rows + dest.dst_row0 + r) * N1) * rec;
        repeat_record(dst, src + q * rec, rec, N1);
        if (progress) progress->done += dst_row_bytes;
      }
    });
    return INFLX_OK;
  }
  // one row of the grid: (P, 1, N1, K) resp. (P, K, 1, N1)
  std::vector<double> image(P * N1 * K);
  if ((rc = sweep_host_body(m, op, p, P, n_p, image.data(), ss, N0, N1, row_begin, 1, layout, 0.0, HostDest(), nullptr))) return rc;
  const char* src = reinterpret_cast<const char*>(image.data());
  parallel_blocks(blocks * row_count, dst_row_bytes, [&](size_t first, size_t last) {
    for (size_t q = first; q < last; ++q) {
      const size_t b = q / row_count, r = q % row_count;
      char* dst = out + ((b * dest.dst_rows + dest.dst_row0 + r) * N1) * rec;
      stream_copy(dst, src + b * dst_row_bytes, dst_row_bytes);
      if (progress) progress->done += dst_row_bytes;
    }
  });
  return INFLX_OK;
}

// host-result sweep for every operation; the slab holds kOpBytes[op] bytes per grid point
int sweep_host_body(inflx_model* m, int op, const double* p, size_t P, size_t n_p, void* out_v, const double* ss, size_t N0,
                    size_t N1, size_t row_begin, size_t row_count, int layout, double accuracy, HostDest dest, Progress* progress) {
  INFLX_SERIALISE(m);
  char* const out = static_cast<char*>(out_v);
  int rc = validate(m, op, p, P, n_p);
  if (rc) return rc;
  if (!out || !ss) return fail(INFLX_ERR_ARG, "output / start_stop pointer is NULL");
  if (layout != INFLX_AOS && layout != INFLX_SOA) return fail(INFLX_ERR_ARG, "unknown layout %d", layout);
  if (row_begin + row_count > N0) return fail(INFLX_ERR_SHAPE, "rows [%zu,%zu) exceed the grid (N0 = %zu)", row_begin, row_begin + row_count, N0);
  if (row_count == 0 || N1 == 0) return INFLX_OK;
  if (dest.dst_rows == 0) dest.dst_rows = row_count;
  if (dest.dst_row0 + row_count > dest.dst_rows) return fail(INFLX_ERR_SHAPE, "slab rows [%zu,%zu) exceed the destination's %zu rows", dest.dst_row0, dest.dst_row0 + row_count, dest.dst_rows);
  HIP_TRY(hipSetDevice(m->device));

  const size_t K = kOpWidth[op];
  const size_t row_bytes = N1 * kOpBytes[op];
  const size_t total = P * row_count * row_bytes;
  const bool subset = dest.planes != 0;
  if (subset && (layout != INFLX_SOA || K == 1 || dest.plane0 + dest.planes > K))
    return fail(INFLX_ERR_ARG, "a plane subset [%zu, %zu) needs the planes layout of an operation with at least that many values (this one has %zu)", dest.plane0, dest.plane0 + dest.planes, K);
  if (!subset && op != INFLX_OP_QDIF && host_fill_enabled() && total >= host_fill_min_bytes()) {
    // a result that is constant along a grid axis is written by host threads from the one evaluated line
    if ((m->info.out_mask & 2u) == 0 && N1 > 1) return sweep_host_broadcast(m, op, 1, p, P, n_p, out, ss, N0, N1, row_begin, row_count, layout, dest, progress);
    if ((m->info.out_mask & 3u) == 2u && row_count > 1) return sweep_host_broadcast(m, op, 0, p, P, n_p, out, ss, N0, N1, row_begin, row_count, layout, dest, progress);
  }
  // (a plane subset prefers the whole-result path whatever its size -- one launch, few large copies -- and falls back to the chunk
  // pipeline like everything else when that much HBM is not free)
  bool whole = whole_result_limit() != 0 && (total <= whole_result_limit() || subset);  // (INFLX_WHOLE_RESULT_MB=0 switches the path off altogether)
  if (whole && total > m->d_whole_cap) {
    if (m->d_whole) HIP_TRY(hipFree(m->d_whole));
    m->d_whole = nullptr;
    m->d_whole_cap = 0;
    if (hipMalloc(&m->d_whole, total) == hipSuccess) {
      m->d_whole_cap = total;
    } else {
      (void)hipGetLastError();  // not enough free HBM for the whole result: the chunk pipeline needs 64 MiB
      m->d_whole = nullptr;
      whole = false;  // (a plane subset too: the pipeline below copies the requested planes of every chunk)
    }
  }
  // the stream of the kernels that read the parameters: decided by the P the launches below really see (the
  // whole-result path launches all P rows at once, the chunk pipeline one parameter row at a time)
  hipStream_t reader = evaluates_on_side_stream(m, op, layout, whole ? P : 1, N1, row_count, /*alone=*/whole) ? m->side : m->stream;
  const double* d_params = nullptr;
  if ((rc = acquire_params(m, p, P * n_p, reader, &d_params))) return rc;
  if (whole) {
    // One launch for everything, and as few copies as the destination allows (one when the caller's array is the slab):
    // the device buffer has the layout of the slab.  The destination pages are made resident by helper threads that run
    // ahead of the copy: the first stretch before the copy starts, the rest -- in stripes dealt round-robin, so that the
    // resident frontier advances at the aggregate rate, several times the PCIe rate -- while it is under way.
    rc = launch_grid(m, op, d_params, P, static_cast<double*>(m->d_whole), ss, N0, N1, row_begin, row_count, layout, m->stream, 0, accuracy);
    if (rc) {  // kernels enqueued before the failure may still read the parameter slot
      (void)hipStreamSynchronize(m->side);
      (void)hipStreamSynchronize(m->stream);
      return rc;
    }
    const std::vector<Span> spans = transfer_spans(op, P, N1, row_count, layout, dest);
    // the destination as a sequence of stripes of at most 16 MiB, in transfer order
    struct Stripe { char* at; size_t bytes; };
    std::vector<Stripe> stripes;
    const size_t stripe = size_t(16) << 20;
    for (const Span& sp : spans) {
      advise_huge_pages(out + sp.dst, sp.bytes);
      for (size_t off = 0; off < sp.bytes; off += stripe) stripes.push_back({out + sp.dst + off, std::min(stripe, sp.bytes - off)});
    }
    // helpers may only walk the destination while the copy is under way if their page touch is a real atomic
    // read-modify-write (touch_range); elsewhere everything is made resident before the copy starts
    size_t head_stripes = 0;
    {
      char* run_at = nullptr;  // stripes that follow each other in memory are made resident by ONE call (its threads share the run)
      size_t run_bytes = 0;
      for (size_t bytes = 0; head_stripes < stripes.size() && (!kTouchIsAtomic || bytes < (size_t(64) << 20)); ++head_stripes) {
        const Stripe& st = stripes[head_stripes];
        if (run_bytes && run_at + run_bytes != st.at) {
          prefault_range(run_at, run_bytes);
          run_bytes = 0;
        }
        if (!run_bytes) run_at = st.at;
        run_bytes += st.bytes;
        bytes += st.bytes;
      }
      if (run_bytes) prefault_range(run_at, run_bytes);
    }
    std::vector<std::thread> pool;
    if (head_stripes < stripes.size()) {
      const size_t rest = stripes.size() - head_stripes;
      const unsigned nthreads = (unsigned)std::min<size_t>(prefault_threads(), rest);
      // helpers are an optimisation: the copy is correct without them (the runtime faults pages in itself,
      // slowly), so a thread that cannot be created is simply not there
      try {
        for (unsigned t = 0; t < nthreads; ++t)
          pool.emplace_back([&stripes, head_stripes, t, nthreads] {
            for (size_t k = head_stripes + t; k < stripes.size(); k += nthreads) touch_range(stripes[k].at, stripes[k].bytes);
          });
      } catch (...) {
      }
    }
    // Copies in slices of at most 256 MiB, all enqueued at once; with a progress sink an event behind every slice tells
    // how far the transfer has come (the waits below cost nothing: the host has nothing else to do).
    const size_t slice = size_t(256) << 20;
    hipError_t e = hipSuccess;
    std::vector<std::pair<hipEvent_t, size_t>> marks;
    for (const Span& sp : spans) {
      for (size_t off = 0; off < sp.bytes && e == hipSuccess; off += slice) {
        const size_t n = std::min(slice, sp.bytes - off);
        e = hipMemcpyAsync(out + sp.dst + off, static_cast<char*>(m->d_whole) + sp.src + off, n, hipMemcpyDeviceToHost, m->stream);
        if (progress && e == hipSuccess) {
          hipEvent_t ev = nullptr;
          if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(ev, m->stream) == hipSuccess) {
            marks.push_back({ev, n});
          } else {
            if (ev) (void)hipEventDestroy(ev);
            marks.push_back({nullptr, n});
          }
        }
      }
    }
    size_t unmarked = 0;
    for (auto& mk : marks) {
      if (mk.first) {
        if (e == hipSuccess) e = hipEventSynchronize(mk.first);
        (void)hipEventDestroy(mk.first);
        if (e == hipSuccess) progress->done += mk.second;
      } else {
        unmarked += mk.second;
      }
    }
    {
      const hipError_t se = hipStreamSynchronize(m->stream);
      if (e == hipSuccess) e = se;
    }
    if (progress && e == hipSuccess) progress->done += unmarked;
    for (auto& th : pool) th.join();
    // a buffer of many GiB is not kept between calls (the model would sit on that much HBM); after a plane subset -- calc_V_array
    // keeps one plane of five -- not even one of more than 1 GiB
    if (m->d_whole_cap > (size_t(4) << 30) || (subset && m->d_whole_cap > (size_t(1) << 30))) {
      (void)hipFree(m->d_whole);
      m->d_whole = nullptr;
      m->d_whole_cap = 0;
    }
    if (e != hipSuccess) return fail(INFLX_ERR_DEVICE, "device-to-host copy failed: %s", hipGetErrorString(e));
    return INFLX_OK;
  }
  // rows per chunk: whole rows of ONE parameter row at a time (keeps every copy contiguous in the
  // AoS result; the SoA result is copied plane by plane)
  size_t rows_per_chunk = std::max<size_t>(1, chunk_bytes_limit() / row_bytes);
  rows_per_chunk = std::min(rows_per_chunk, row_count);
  const size_t chunk_bytes = rows_per_chunk * row_bytes;
  for (int k = 0; k < 2; ++k)
    if ((rc = ensure_chunk(m, k, chunk_bytes))) return rc;

  // ping-pong: kernel for chunk c on `stream` into buffer c&1, copy-back on `copy_stream`; a helper thread
  // makes the destination pages of chunk c+1 resident while chunk c is being copied
  struct Piece { size_t pr, r, nrows; };
  std::vector<Piece> pieces;
  for (size_t pr = 0; pr < P; ++pr)
    for (size_t r = 0; r < row_count; r += rows_per_chunk) pieces.push_back({pr, r, std::min(rows_per_chunk, row_count - r)});
  const bool planes = layout == INFLX_SOA && K > 1;
  // planes [k_lo, k_hi) of every chunk reach the host; the destination holds k_hi - k_lo planes per parameter row (all K unless a subset was asked for)
  const size_t k_lo = subset ? dest.plane0 : 0, k_hi = subset ? dest.plane0 + dest.planes : K;
  const size_t piece_row_bytes = planes ? (k_hi - k_lo) * N1 * sizeof(double) : row_bytes;  // host bytes per grid row of a piece (progress)
  // where rows [r, r + nrows) of parameter row pr (plane k) land in the caller's array
  auto dst_of = [&](const Piece& pc, size_t k) {
    return planes ? out + (((pc.pr * (k_hi - k_lo) + (k - k_lo)) * dest.dst_rows) + dest.dst_row0 + pc.r) * N1 * sizeof(double)
                  : out + ((pc.pr * dest.dst_rows) + dest.dst_row0 + pc.r) * row_bytes;
  };
  auto touch = [&](const Piece& pc) {
    if (!planes) {
      prefault_range(dst_of(pc, 0), pc.nrows * row_bytes);
    } else {
      for (size_t k = k_lo; k < k_hi; ++k) prefault_range(dst_of(pc, k), pc.nrows * N1 * sizeof(double));
    }
  };
  auto start_touch = [&](const Piece& pc) {
    try {
      return std::async(std::launch::async, touch, pc);
    } catch (...) {  // no thread to be had: do it here
      touch(pc);
      return std::future<void>();
    }
  };
  std::future<void> ready = start_touch(pieces[0]);
  bool used[2] = {false, false};
  // on every exit -- errors included -- nothing may still be writing to the caller's buffer
  auto drain = [&] {
    if (ready.valid()) ready.wait();
    (void)hipStreamSynchronize(m->side);
    (void)hipStreamSynchronize(m->stream);
    (void)hipStreamSynchronize(m->copy_stream);
  };
  size_t counted = 0;  // pieces whose bytes the progress sink has been told about
  for (size_t c = 0; c < pieces.size(); ++c) {
    const Piece& pc = pieces[c];
    const int b = (int)(c & 1);
    // a chunk holds rows of a single parameter row: launch with P = 1 at that row's parameters
    if (used[b]) {
      // with a progress sink the host waits for the copy that frees this buffer (piece c - 2) instead of leaving the wait
      // to the stream: it then knows how far the transfer has come, and stays two chunks ahead of it at most
      hipError_t we = progress ? hipEventSynchronize(m->copy_done[b]) : hipStreamWaitEvent(m->stream, m->copy_done[b], 0);
      if (we != hipSuccess) { drain(); return fail(INFLX_ERR_DEVICE, "waiting for a chunk copy failed: %s", hipGetErrorString(we)); }
      if (progress)
        for (; counted + 2 <= c; ++counted) progress->done += pieces[counted].nrows * piece_row_bytes;
    }
    rc = launch_grid(m, op, d_params + pc.pr * n_p, 1, static_cast<double*>(m->d_chunk[b]), ss, N0, N1, row_begin + pc.r, pc.nrows, layout,
                     m->stream, 0, accuracy);
    if (rc) { drain(); return rc; }
    if (ready.valid()) ready.wait();  // pages of this chunk's destination are resident
    if (c + 1 < pieces.size()) ready = start_touch(pieces[c + 1]);
    hipError_t e = hipEventRecord(m->chunk_done[b], m->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(m->copy_stream, m->chunk_done[b], 0);
    if (e == hipSuccess) {
      if (!planes) {
        e = hipMemcpyAsync(dst_of(pc, 0), m->d_chunk[b], pc.nrows * row_bytes, hipMemcpyDeviceToHost, m->copy_stream);
      } else {
        for (size_t k = k_lo; k < k_hi && e == hipSuccess; ++k) {
          const double* src = static_cast<const double*>(m->d_chunk[b]) + k * pc.nrows * N1;
          e = hipMemcpyAsync(dst_of(pc, k), src, pc.nrows * N1 * sizeof(double), hipMemcpyDeviceToHost, m->copy_stream);
        }
      }
    }
    if (e == hipSuccess) e = hipEventRecord(m->copy_done[b], m->copy_stream);
    if (e != hipSuccess) { drain(); return fail(INFLX_ERR_DEVICE, "device-to-host copy failed: %s", hipGetErrorString(e)); }
    used[b] = true;
  }
  if (ready.valid()) ready.wait();
  HIP_TRY(hipStreamSynchronize(m->copy_stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  if (progress)
    for (; counted < pieces.size(); ++counted) progress->done += pieces[counted].nrows * piece_row_bytes;
  return INFLX_OK;
}

// the sweep, then what the special functions had to say about it (the result is complete in the caller's memory either way)
int sweep_host_impl(inflx_model* m, int op, const double* p, size_t P, size_t n_p, void* out_v, const double* ss, size_t N0,
                    size_t N1, size_t row_begin, size_t row_count, int layout, double accuracy, HostDest dest, Progress* progress) {
  INFLX_SERIALISE(m);
  const int rc = sweep_host_body(m, op, p, P, n_p, out_v, ss, N0, N1, row_begin, row_count, layout, accuracy, dest, progress);
  return rc ? rc : sf_verdict(m);
}

// ---- progress lines of long host-result calls --------------------------------------------------------------------------
// The reference draws an indicatif bar on stderr at 2 Hz while a sweep runs: time to completion, operations per second,
// percentage (src/anguelova.rs:42-50, ticked once per grid point).  A sweep here is too short for a bar unless its result
// is tens of gigabytes (PCIe at ~50 GB/s): a reporter thread prints the same three figures, from the bytes that have
// arrived in the caller's array, first after 0.5 s and then twice a second -- calls shorter than that stay silent, and
// only calls whose result is at least 1 GiB get a reporter at all.  (INFLX_PROGRESS_MIN_MB / INFLX_PROGRESS_INTERVAL_MS:
// knobs for the tests.)
size_t progress_min_bytes() {
  static const size_t v = [] {
    const char* e = getenv("INFLX_PROGRESS_MIN_MB");
    const long mb = e ? atol(e) : -1;
    return (size_t)(mb >= 0 ? mb : 1024) << 20;
  }();
  return v;
}
long progress_interval_ms() {
  static const long v = [] {
    const char* e = getenv("INFLX_PROGRESS_INTERVAL_MS");
    const long ms = e ? atol(e) : 0;
    return ms > 0 ? ms : 500L;
  }();
  return v;
}

class Reporter {
 public:
  Reporter(Progress* pr, double points, bool wanted) : pr_(pr), points_(points) {
    if (!wanted || !pr || pr->total < progress_min_bytes()) return;
    t0_ = std::chrono::steady_clock::now();
    try {
      th_ = std::thread([this] { run(); });
    } catch (...) {  // no thread to be had: no progress lines
    }
  }
  ~Reporter() {
    if (!th_.joinable()) return;
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    th_.join();
  }
  Reporter(const Reporter&) = delete;
  Reporter& operator=(const Reporter&) = delete;
  // is anybody reading the counter?  (no: the pipelines are not asked to keep it -- no slice events, no host-side chunk waits)
  bool active() const { return th_.joinable(); }

 private:
  void run() {
    std::unique_lock<std::mutex> g(mu_);
    const auto step = std::chrono::milliseconds(progress_interval_ms());
    while (!cv_.wait_for(g, step, [this] { return stop_; })) {
      const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
      const double frac = std::min(1.0, (double)pr_->done.load() / (double)std::max<uint64_t>(pr_->total, 1));
      const double rate = frac * points_ / std::max(sec, 1e-9);
      char eta[32];
      if (frac > 0.0) snprintf(eta, sizeof eta, "%.1f s", sec * (1.0 - frac) / frac); else snprintf(eta, sizeof eta, "?");
      say("Time to completion: %s | grid points/s: %.3g | %3.0f%%", eta, rate, 100.0 * frac);
    }
  }
  Progress* pr_;
  double points_;
  std::chrono::steady_clock::time_point t0_;
  std::thread th_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool stop_ = false;
};

int grid_entry(inflx_model* m, int op, const double* p, size_t n_p, double* out, const double* ss, size_t N0, size_t N1,
               int progress, const char* what) {
  if (!out) return fail(INFLX_ERR_ARG, "output array is NULL");
  if (!ss) return fail(INFLX_ERR_ARG, "start_stop array is NULL");
  int rc = validate(m, op, p, 1, n_p);
  if (rc) return rc;
  const auto t0 = std::chrono::steady_clock::now();
  if (progress) say("Calculating %s on HIP device %d (%s).", what, m->device, m->name.c_str());
  Progress pr;
  pr.total = (uint64_t)N0 * N1 * kOpBytes[op];
  {
    Reporter reporter(&pr, (double)N0 * (double)N1, progress != 0);
    rc = sweep_host_impl(m, op, p, 1, n_p, out, ss, N0, N1, 0, N0, INFLX_AOS, 0.0, HostDest(), reporter.active() ? &pr : nullptr);
  }
  if (rc) return rc;
  if (progress) {
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    say("Calculation finished. Took %.3f s (%.3g grid points/s including the copy to host memory).", sec,
        (double)N0 * (double)N1 / std::max(sec, 1e-9));
  }
  return INFLX_OK;
}

// The contiguous balanced block of `n_items` that part `rank` of `world` owns: the first n_items % world parts get one
// item more (inflatox_amd/distributed.py block_bounds: one partition rule for the in-process and the one-process-per-GPU form).
void block_bounds(size_t n_items, size_t world, size_t rank, size_t* begin, size_t* count) {
  const size_t base = n_items / world, extra = n_items % world;
  *begin = rank * base + std::min(rank, extra);
  *count = base + (rank < extra ? 1 : 0);
}

struct ShardPlan {
  int axis;  // 0 = the parameter axis is split, 1 = the grid's row axis
  size_t p_begin, p_count, row_begin, row_count;
};
ShardPlan shard_plan(size_t P, size_t N0, size_t world, size_t rank) {
  ShardPlan s;
  if (P >= world) {
    s.axis = 0;
    block_bounds(P, world, rank, &s.p_begin, &s.p_count);
    s.row_begin = 0;
    s.row_count = N0;
  } else {
    s.axis = 1;
    s.p_begin = 0;
    s.p_count = P;
    block_bounds(N0, world, rank, &s.row_begin, &s.row_count);
  }
  return s;
}
}  // namespace

// One model artefact opened on several devices: the handles of a sweep that uses more than one GPU from one call.
struct inflx_multi {
  std::vector<inflx_model*> dev;
  std::string path;
  // device-resident all-gather (inflx_sweep_allgather_multi): per source device one copy stream per peer -- xGMI is point to
  // point, so a device pushes its slab to all peers at once, each copy on a link of its own -- and the events that order
  // the pushes behind the sweep
  std::vector<std::vector<hipStream_t>> push;  // push[k][j]: stream on device k that copies towards device j
  std::vector<hipEvent_t> swept;               // recorded on device k's sweep stream behind its sweep
  // the same gather as ONE RCCL collective (inflx_sweep_allgather_multi_ex, INFLX_GATHER_RCCL): one communicator per device,
  // created by ncclCommInitAll on first use (void*: RCCL is loaded at run time, see rccl_api())
  std::vector<void*> comms;
  std::mutex mu;
};

namespace {
// RCCL, bound at run time.  The library is not a link-time dependency of libinflx_hip.so: the single-GPU product needs no
// collective at all, and a process that has PyTorch loaded already holds a copy of RCCL (torch bundles one) -- a second copy
// pulled in by the dynamic linker would be a second set of communicator state.  So: the copy that is already mapped if
// there is one, else the system's.
struct RcclApi {
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string origin;
};
constexpr int kNcclFloat64 = 8;  // ncclDataType_t::ncclFloat64 (rccl.h)

// (`why`: filled when the library or one of its symbols could not be bound -- the text is captured where dlopen / dlsym failed, once;
// dlerror() itself is per-thread state that the next dl* call of anybody overwrites)
const RcclApi* rccl_api(std::string* why = nullptr) {
  static std::string failure;
  static const RcclApi api = [] {
    RcclApi a;
    void* h = nullptr;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names)
      if ((h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) { a.origin = std::string(nm) + " (already mapped)"; break; }
    if (!h)
      for (const char* nm : names) {
        if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) { a.origin = nm; break; }
        const char* e = dlerror();
        failure += std::string(failure.empty() ? "" : "; ") + nm + ": " + (e ? e : "not found");
      }
    if (!h) return a;
    failure.clear();
    auto bind = [&](const char* sym) -> void* {
      void* fn = dlsym(h, sym);
      if (!fn) failure += std::string(failure.empty() ? "" : ", ") + sym;
      return fn;
    };
    a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(bind("ncclCommInitAll"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(bind("ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(bind("ncclAllGather"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(bind("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(bind("ncclGroupEnd"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(bind("ncclGetErrorString"));
    if (!failure.empty()) failure = a.origin + " lacks " + failure;
    return a;
  }();
  const bool ok = api.CommInitAll && api.CommDestroy && api.AllGather && api.GroupStart && api.GroupEnd && api.GetErrorString;
  if (!ok && why) *why = failure.empty() ? "library or symbols missing" : failure;
  return ok ? &api : nullptr;
}

// What the in-place RCCL all-gather of a sharded sweep hands ncclAllGather, per (image q, rank k): byte offsets into rank k's full-size
// buffer of its own block (send) and of the image (recv), and the element count of one block.  Split along the parameter axis the whole
// (P, N0, N1, K) array is ONE image; split along grid rows every parameter row is one (its blocks are contiguous within the row's
// (N0, N1, K) image).  rccl.h's in-place rule: sendbuff == recvbuff + rank * sendcount.  (A function of its own so that the offsets can
// be checked against shard_plan without a second GPU: tests/host_units.cpp.)
struct GatherCall {
  size_t image, rank, send_off, recv_off, count;
};
std::vector<GatherCall> rccl_gather_calls(size_t P, size_t N0, size_t row_bytes, size_t world) {
  std::vector<GatherCall> calls;
  const ShardPlan s0 = shard_plan(P, N0, world, 0);
  const size_t images = s0.axis == 0 ? 1 : P;
  const size_t block_bytes = s0.axis == 0 ? s0.p_count * N0 * row_bytes : s0.row_count * row_bytes;
  const size_t image_bytes = s0.axis == 0 ? P * N0 * row_bytes : N0 * row_bytes;
  for (size_t q = 0; q < images; ++q)
    for (size_t k = 0; k < world; ++k) calls.push_back({q, k, q * image_bytes + k * block_bytes, q * image_bytes, block_bytes / sizeof(double)});
  return calls;
}

#define RCCL_TRY(api, expr)                                                                                       \
  do {                                                                                                            \
    const int r_ = (expr);                                                                                        \
    if (r_ != 0) return fail(INFLX_ERR_DEVICE, "%s failed: %s", #expr, (api)->GetErrorString(r_));               \
  } while (0)

int ensure_comms(inflx_multi* mm, const RcclApi* api) {
  std::lock_guard<std::mutex> g(mm->mu);
  const size_t n = mm->dev.size();
  if (mm->comms.size() == n) return INFLX_OK;
  std::vector<int> devs(n);
  for (size_t k = 0; k < n; ++k) devs[k] = mm->dev[k]->device;
  for (size_t k = 0; k < n; ++k)
    for (size_t j = k + 1; j < n; ++j)
      if (devs[k] == devs[j])
        return fail(INFLX_ERR_ARG, "the RCCL all-gather needs one device per handle (device %d appears twice): a communicator has one rank per GPU; "
                    "the peer-push gather (INFLX_GATHER_PEER_PUSH) has no such restriction", devs[k]);
  std::vector<void*> comms(n, nullptr);
  RCCL_TRY(api, api->CommInitAll(comms.data(), (int)n, devs.data()));
  mm->comms = std::move(comms);
  return INFLX_OK;
}

// Run `part(k)` for k in [0, n) -- one host thread per part, part 0 on the calling thread -- and report the first failure
// (status and message) on the calling thread.
template <typename F>
int run_parts(size_t n, F part) {
  std::vector<int> rcs(n, INFLX_OK);
  std::vector<std::string> msgs(n);
  auto body = [&](size_t k) {
    const unsigned before = tl_sharers;
    tl_sharers = (unsigned)n;  // n device pipelines share the process's host-thread budget for the duration of the call
    rcs[k] = part(k);
    tl_sharers = before;
    if (rcs[k] != INFLX_OK) msgs[k] = g_last_error;  // thread-local: carried back by hand
  };
  std::vector<std::thread> pool;
  std::vector<size_t> inline_parts{0};
  for (size_t k = 1; k < n; ++k) {
    try {
      pool.emplace_back(body, k);
    } catch (...) {  // no thread to be had: this part runs on the calling thread after its own
      inline_parts.push_back(k);
    }
  }
  for (size_t k : inline_parts)
    if (k < n) body(k);
  for (auto& th : pool) th.join();
  for (size_t k = 0; k < n; ++k)
    if (rcs[k] != INFLX_OK) {
      g_last_error = msgs[k];
      return rcs[k];
    }
  return INFLX_OK;
}

size_t devices_used(const inflx_multi* mm, size_t max_devices) {
  const size_t n = mm->dev.size();
  return max_devices ? std::min(n, max_devices) : n;
}
}  // namespace

extern "C" {

int inflx_sweep_host(inflx_model* m, int op, const double* p, size_t P, size_t n_p, double* out, const double* ss, size_t N0,
                     size_t N1, size_t row_begin, size_t row_count, int layout) {
  if (op == INFLX_OP_QDIF) return fail(INFLX_ERR_ARG, "the flag sweep has a byte result: use inflx_flag_quantum_dif");
  return sweep_host_impl(m, op, p, P, n_p, out, ss, N0, N1, row_begin, row_count, layout, 0.0, HostDest(), nullptr);
}

int inflx_sweep_host_planes(inflx_model* m, int op, const double* p, size_t P, size_t n_p, double* out, const double* ss, size_t N0,
                            size_t N1, size_t row_begin, size_t row_count, size_t first_plane, size_t n_planes) {
  if (op == INFLX_OP_QDIF) return fail(INFLX_ERR_ARG, "the flag sweep has a byte result: use inflx_flag_quantum_dif");
  if (n_planes == 0) return fail(INFLX_ERR_ARG, "no planes requested");
  HostDest dest;
  dest.plane0 = first_plane;
  dest.planes = n_planes;
  return sweep_host_impl(m, op, p, P, n_p, out, ss, N0, N1, row_begin, row_count, INFLX_SOA, 0.0, dest, nullptr);
}

int inflx_flag_quantum_dif(inflx_model* m, const double* p, size_t n_p, uint8_t* out, const double* ss, size_t N0, size_t N1,
                           int progress, double accuracy) {
  if (!out || !ss) return fail(INFLX_ERR_ARG, "output / start_stop pointer is NULL");
  if (!m) return fail(INFLX_ERR_ARG, "model handle is NULL");
  if (progress) say("Calculating zeros of the potential gradient on HIP device %d.", m->device);
  return sweep_host_impl(m, INFLX_OP_QDIF, p, 1, n_p, out, ss, N0, N1, 0, N0, INFLX_AOS, accuracy, HostDest(), nullptr);
}

int inflx_complete_analysis(inflx_model* m, const double* p, size_t n_p, double* out, const double* ss, size_t N0, size_t N1,
                            int progress, size_t /*threads*/) {
  return grid_entry(m, INFLX_OP_COMPLETE, p, n_p, out, ss, N0, N1, progress, "full analysis");
}
int inflx_consistency_only(inflx_model* m, const double* p, size_t n_p, double* out, const double* ss, size_t N0, size_t N1,
                           int progress, size_t /*threads*/) {
  return grid_entry(m, INFLX_OP_CONSISTENCY, p, n_p, out, ss, N0, N1, progress, "consistency condition");
}
int inflx_consistency_rapidturn_only(inflx_model* m, const double* p, size_t n_p, double* out, const double* ss, size_t N0,
                                     size_t N1, int progress, size_t /*threads*/) {
  return grid_entry(m, INFLX_OP_RAPIDTURN, p, n_p, out, ss, N0, N1, progress, "consistency condition (rapid-turn limit)");
}
int inflx_epsilon_v_only(inflx_model* m, const double* p, size_t n_p, double* out, const double* ss, size_t N0, size_t N1,
                         int progress, size_t /*threads*/) {
  return grid_entry(m, INFLX_OP_EPSILON_V, p, n_p, out, ss, N0, N1, progress, "potential slow-roll parameter ε_V");
}

int inflx_sweep_on_trajectory(inflx_model* m, int op, const double* p, size_t n_p, const double* x, size_t n, double* out,
                              int progress, size_t /*threads*/) {
  INFLX_SERIALISE(m);
  int rc = validate(m, op, p, 1, n_p);
  if (rc) return rc;
  if (op == INFLX_OP_QDIF) return fail(INFLX_ERR_ARG, "the flag sweep has no on-trajectory variant in the reference");
  if (n == 0) return INFLX_OK;
  if (!x || !out) return fail(INFLX_ERR_ARG, "trajectory / output pointer is NULL");
  HIP_TRY(hipSetDevice(m->device));
  const size_t K = kOpWidth[op];
  const size_t in_bytes = n * 2 * sizeof(double), out_bytes = n * K * sizeof(double);
  if ((rc = ensure_chunk(m, 0, out_bytes))) return rc;
  if ((rc = ensure_chunk(m, 1, in_bytes))) return rc;
  const double* d_params = nullptr;
  if ((rc = acquire_params(m, p, n_p, m->stream, &d_params))) return rc;
  if (progress) say("Calculating on trajectory (%zu points) on HIP device %d.", n, m->device);
  HIP_TRY(hipMemcpyAsync(m->d_chunk[1], x, in_bytes, hipMemcpyHostToDevice, m->stream));
  InflxTrajectoryArgs a;
  memset(&a, 0, sizeof a);
  a.out = static_cast<double*>(m->d_chunk[0]);
  a.params = d_params;
  a.points = static_cast<const double*>(m->d_chunk[1]);
  a.n = n;
  a.P = 1;
  void* params[] = {&a};
  const size_t gx = (n + m->info.tile_cols - 1) / m->info.tile_cols;
  HIP_TRY(hipModuleLaunchKernel(m->traj[op], (unsigned)gx, 1, 1, m->info.tile_cols, 1, 1, 0, m->stream, params, nullptr));
  // a long trajectory's result lands in a fresh host array: make its pages resident (huge pages, several threads) while the
  // kernel runs, instead of letting the copy fault them in one by one (14 GB/s against 50, section 5 of DESIGN.md)
  if (out_bytes >= (size_t(4) << 20)) prefault_range(reinterpret_cast<char*>(out), out_bytes);
  HIP_TRY(hipMemcpyAsync(out, m->d_chunk[0], out_bytes, hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  return sf_verdict(m);
}


// ---- one call, several GPUs -------------------------------------------------------------------------------------------
// The reference's knob for "use the whole machine" is `threads`: None -> 0 -> a rayon pool over all cores
// (python/inflatox/consistency_conditions.py:297, src/anguelova.rs:524-540).  Here the machine's parallelism is its GPUs:
// a multi-handle opens the artefact on several devices, and one call splits the outermost axis of the sweep into one
// contiguous block per device (shard_plan: the parameter axis when there are at least as many parameter rows as devices,
// the grid's rows otherwise -- the partition of inflatox_amd/distributed.py), runs every device's pipeline on a host thread
// of its own, and lets each device copy its slab straight into its place in the caller's array.  No exchange between
// devices: every grid point and every parameter row is independent.

int inflx_shard_plan(size_t P, size_t N0, int world, int rank, size_t plan[5]) {
  if (!plan) return fail(INFLX_ERR_ARG, "plan array is NULL");
  if (world < 1 || rank < 0 || rank >= world) return fail(INFLX_ERR_ARG, "rank %d outside world of size %d", rank, world);
  const ShardPlan s = shard_plan(P, N0, (size_t)world, (size_t)rank);
  plan[0] = (size_t)s.axis;
  plan[1] = s.p_begin;
  plan[2] = s.p_count;
  plan[3] = s.row_begin;
  plan[4] = s.row_count;
  return INFLX_OK;
}

int inflx_open_multi(const char* artefact_path, const int* devices, int n_dev, inflx_multi** out) {
  if (!artefact_path || !out) return fail(INFLX_ERR_ARG, "artefact path / output handle is NULL");
  *out = nullptr;
  std::vector<int> ids;
  if (!devices || n_dev <= 0) {  // every visible device
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
      return fail(INFLX_ERR_DEVICE, "no HIP device available: the sweep has no CPU fallback");
    for (int d = 0; d < count; ++d) ids.push_back(d);
  } else {
    ids.assign(devices, devices + n_dev);
  }
  inflx_multi* mm = new inflx_multi();
  mm->path = artefact_path;
  for (int d : ids) {
    inflx_model* m = nullptr;
    const int rc = inflx_open(artefact_path, d, &m);
    if (rc != INFLX_OK) {
      const std::string first = g_last_error;
      inflx_close_multi(mm);
      g_last_error = first;
      return rc;
    }
    mm->dev.push_back(m);
  }
  *out = mm;
  return INFLX_OK;
}

void inflx_close_multi(inflx_multi* mm) {
  if (!mm) return;
  for (size_t k = 0; k < mm->push.size(); ++k) {
    (void)hipSetDevice(mm->dev[k]->device);
    for (hipStream_t st : mm->push[k])
      if (st) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
      }
    if (k < mm->swept.size() && mm->swept[k]) (void)hipEventDestroy(mm->swept[k]);
  }
  if (!mm->comms.empty()) {
    if (const RcclApi* api = rccl_api())
      for (size_t k = 0; k < mm->comms.size(); ++k)
        if (mm->comms[k]) {
          (void)hipSetDevice(mm->dev[k]->device);
          (void)api->CommDestroy(mm->comms[k]);
        }
  }
  for (inflx_model* m : mm->dev) inflx_close(m);
  delete mm;
}

int inflx_multi_device_count(const inflx_multi* mm) { return mm ? (int)mm->dev.size() : 0; }

inflx_model* inflx_multi_handle(const inflx_multi* mm, int index) {
  if (!mm || index < 0 || (size_t)index >= mm->dev.size()) return nullptr;
  return mm->dev[(size_t)index];
}

int inflx_sweep_host_multi(inflx_multi* mm, int op, const double* p, size_t P, size_t n_p, double* out, const double* ss, size_t N0,
                           size_t N1, int layout, int progress, size_t max_devices) {
  if (!mm || mm->dev.empty()) return fail(INFLX_ERR_ARG, "multi-device handle is NULL or empty");
  if (op == INFLX_OP_QDIF) return fail(INFLX_ERR_ARG, "the flag sweep has a byte result: use inflx_flag_quantum_dif on one device");
  int rc = validate(mm->dev[0], op, p, P, n_p);
  if (rc) return rc;
  if (!out || !ss) return fail(INFLX_ERR_ARG, "output / start_stop pointer is NULL");
  if (layout != INFLX_AOS && layout != INFLX_SOA) return fail(INFLX_ERR_ARG, "unknown layout %d", layout);
  if (N0 == 0 || N1 == 0) return INFLX_OK;
  const size_t world = devices_used(mm, max_devices);
  const auto t0 = std::chrono::steady_clock::now();
  if (progress) say("Calculating on %zu HIP device(s) (%s): %zu parameter row(s) x %zu x %zu grid points.", world, mm->dev[0]->name.c_str(), P, N0, N1);
  Progress pr;
  pr.total = (uint64_t)P * N0 * N1 * kOpBytes[op];
  {
    Reporter reporter(&pr, (double)P * (double)N0 * (double)N1, progress != 0);
    Progress* sink = reporter.active() ? &pr : nullptr;
    rc = run_parts(world, [&](size_t k) -> int {
      const ShardPlan s = shard_plan(P, N0, world, k);
      if (s.p_count == 0 || s.row_count == 0) return INFLX_OK;
      inflx_model* m = mm->dev[k];
      if (s.axis == 0) {  // a block of parameter rows, every grid row: the slab is one contiguous piece of the array
        char* dst = reinterpret_cast<char*>(out) + s.p_begin * N0 * N1 * kOpBytes[op];
        return sweep_host_impl(m, op, p + s.p_begin * n_p, s.p_count, n_p, dst, ss, N0, N1, 0, N0, layout, 0.0, HostDest(), sink);
      }
      HostDest dest;  // every parameter row, a block of grid rows: rows [row_begin, row_begin + row_count) of every block
      dest.dst_rows = N0;
      dest.dst_row0 = s.row_begin;
      return sweep_host_impl(m, op, p, P, n_p, out, ss, N0, N1, s.row_begin, s.row_count, layout, 0.0, dest, sink);
    });
  }
  if (rc) return rc;
  if (progress) {
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    say("Calculation finished. Took %.3f s (%.3g grid points/s including the copy to host memory).", sec,
        (double)P * (double)N0 * (double)N1 / std::max(sec, 1e-9));
  }
  return INFLX_OK;
}

int inflx_complete_analysis_multi(inflx_multi* mm, const double* p, size_t n_p, double* out, const double* ss, size_t N0, size_t N1,
                                  int progress, size_t threads) {
  // `threads` keeps the meaning it has in the reference (anguelova.rs:524: 0 = everything the machine has, k = at most k
  // workers), with the handle's devices as the workers
  return inflx_sweep_host_multi(mm, INFLX_OP_COMPLETE, p, 1, n_p, out, ss, N0, N1, INFLX_AOS, progress, threads);
}

int inflx_sweep_stats_multi(inflx_multi* mm, const double* p, size_t P, size_t n_p, const double* ss, size_t N0, size_t N1,
                            size_t max_devices, inflx_summary* summary) {
  if (!mm || mm->dev.empty()) return fail(INFLX_ERR_ARG, "multi-device handle is NULL or empty");
  if (!summary) return fail(INFLX_ERR_ARG, "summary pointer is NULL");
  int rc = validate(mm->dev[0], INFLX_OP_COMPLETE, p, P, n_p);
  if (rc) return rc;
  const size_t world = devices_used(mm, max_devices);
  std::vector<inflx_summary> part(world);
  for (auto& s : part)
    for (int k = 0; k < 6; ++k) {
      s.min[k] = HUGE_VAL;
      s.max[k] = -HUGE_VAL;
      s.count[k] = 0;
    }
  rc = run_parts(world, [&](size_t k) -> int {
    const ShardPlan s = shard_plan(P, N0, world, k);
    if (s.p_count == 0 || s.row_count == 0) return INFLX_OK;
    return inflx_sweep_device_stats(mm->dev[k], p + s.p_begin * n_p, s.p_count, n_p, nullptr, 0, ss, N0, N1, s.row_begin, s.row_count, nullptr, &part[k]);
  });
  if (rc) return rc;
  *summary = part[0];
  for (size_t d = 1; d < world; ++d)
    for (int k = 0; k < 6; ++k) {
      summary->min[k] = std::fmin(summary->min[k], part[d].min[k]);
      summary->max[k] = std::fmax(summary->max[k], part[d].max[k]);
      summary->count[k] += part[d].count[k];
    }
  return INFLX_OK;
}


// ---- device-resident results on several GPUs ---------------------------------------------------------------------------
// inflx_sweep_device_multi: every device sweeps its block of the outermost axis (inflx_shard_plan) into memory of its own;
// nothing crosses a link.  inflx_sweep_allgather_multi: every device ends up with the WHOLE result.  Each device sweeps its
// block straight into its slice of its own full-size buffer and then PUSHES that slice into the same place of every
// peer's buffer with hipMemcpyPeerAsync, one stream per peer: xGMI is a point-to-point fabric (7 links of ~153 GB/s per
// GPU), so the n - 1 pushes of a device travel on n - 1 different links at the same time and the gather takes
// slab / link rate (8192^2 over 8 GPUs: 403 MB per slab, ~2.6 ms) instead of the (n - 1) slab steps of a ring (~18 ms).
namespace {
int ensure_push_streams(inflx_multi* mm) {
  std::lock_guard<std::mutex> g(mm->mu);
  const size_t n = mm->dev.size();
  if (mm->push.size() == n) return INFLX_OK;
  // built aside and published only when every stream and event exists: a failure part-way must not leave a table of the right
  // size with null entries behind (the next call would take it for complete and record / copy on null handles)
  std::vector<std::vector<hipStream_t>> push(n, std::vector<hipStream_t>(n, nullptr));
  std::vector<hipEvent_t> swept(n, nullptr);
  hipError_t e = hipSuccess;
  const char* what = "";
  for (size_t k = 0; k < n && e == hipSuccess; ++k) {
    if ((e = hipSetDevice(mm->dev[k]->device)) != hipSuccess) { what = "hipSetDevice"; break; }
    if ((e = hipEventCreateWithFlags(&swept[k], hipEventDisableTiming)) != hipSuccess) { what = "hipEventCreateWithFlags"; break; }
    for (size_t j = 0; j < n; ++j) {
      if (j == k) continue;
      if ((e = hipStreamCreateWithFlags(&push[k][j], hipStreamNonBlocking)) != hipSuccess) { what = "hipStreamCreateWithFlags"; break; }
      const int a = mm->dev[k]->device, b = mm->dev[j]->device;
      if (a != b) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) {
          const hipError_t pe = hipDeviceEnablePeerAccess(b, 0);  // "already enabled" is fine
          (void)hipGetLastError();
          if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
            warn("peer access from device %d to device %d could not be enabled (%s): the all-gather stages these copies through the host", a, b, hipGetErrorString(pe));
        } else {
          (void)hipGetLastError();
          warn("device %d cannot access device %d directly: the all-gather stages these copies through the host", a, b);
        }
      }
    }
  }
  if (e != hipSuccess) {
    for (size_t k = 0; k < n; ++k) {
      (void)hipSetDevice(mm->dev[k]->device);
      if (swept[k]) (void)hipEventDestroy(swept[k]);
      for (hipStream_t st : push[k])
        if (st) (void)hipStreamDestroy(st);
    }
    return fail(INFLX_ERR_DEVICE, "%s failed while preparing the peer copies: %s", what, hipGetErrorString(e));
  }
  mm->push = std::move(push);
  mm->swept = std::move(swept);
  return INFLX_OK;
}
}  // namespace

int inflx_sweep_device_multi(inflx_multi* mm, int op, const double* p, size_t P, size_t n_p, void* const* d_out, const size_t* d_out_bytes,
                             const double* ss, size_t N0, size_t N1, int layout, void* const* streams) {
  if (!mm || mm->dev.empty()) return fail(INFLX_ERR_ARG, "multi-device handle is NULL or empty");
  if (!d_out || !d_out_bytes) return fail(INFLX_ERR_ARG, "device buffer array is NULL");
  int rc = validate(mm->dev[0], op, p, P, n_p);
  if (rc) return rc;
  const size_t world = mm->dev.size();
  // launches are asynchronous and cheap: enqueue device after device from the calling thread
  for (size_t k = 0; k < world; ++k) {
    const ShardPlan s = shard_plan(P, N0, world, k);
    if (s.p_count == 0 || s.row_count == 0) continue;
    rc = inflx_sweep_device(mm->dev[k], op, p + s.p_begin * n_p, s.p_count, n_p, d_out[k], d_out_bytes[k], ss, N0, N1, s.row_begin, s.row_count, layout,
                            streams ? streams[k] : nullptr);
    if (rc) return rc;
  }
  return INFLX_OK;
}

int inflx_sweep_allgather_multi(inflx_multi* mm, int op, const double* p, size_t P, size_t n_p, void* const* d_full, size_t d_full_bytes,
                                const double* ss, size_t N0, size_t N1) {
  return inflx_sweep_allgather_multi_ex(mm, op, p, P, n_p, d_full, d_full_bytes, ss, N0, N1, INFLX_GATHER_PEER_PUSH);
}

int inflx_sweep_allgather_multi_ex(inflx_multi* mm, int op, const double* p, size_t P, size_t n_p, void* const* d_full, size_t d_full_bytes,
                                   const double* ss, size_t N0, size_t N1, int gather) {
  if (!mm || mm->dev.empty()) return fail(INFLX_ERR_ARG, "multi-device handle is NULL or empty");
  if (gather != INFLX_GATHER_PEER_PUSH && gather != INFLX_GATHER_RCCL) return fail(INFLX_ERR_ARG, "unknown gather %d", gather);
  if (!d_full) return fail(INFLX_ERR_ARG, "device buffer array is NULL");
  if (op == INFLX_OP_QDIF) return fail(INFLX_ERR_ARG, "the flag sweep has a byte result: use inflx_flag_quantum_dif on one device");
  int rc = validate(mm->dev[0], op, p, P, n_p);
  if (rc) return rc;
  const size_t world = mm->dev.size();
  const size_t point_bytes = kOpBytes[op];
  if (d_full_bytes < P * N0 * N1 * point_bytes) return fail(INFLX_ERR_SHAPE, "every full-size buffer needs %zu bytes (got %zu)", P * N0 * N1 * point_bytes, d_full_bytes);
  if (N0 == 0 || N1 == 0) return INFLX_OK;
  const RcclApi* api = nullptr;
  if (gather == INFLX_GATHER_RCCL) {
    // ncclAllGather moves equal counts: the split axis must divide by the number of devices (shard_plan then gives every device
    // the same block), and every device a buffer of its own
    const ShardPlan s0 = shard_plan(P, N0, world, 0);
    if ((s0.axis == 0 ? P : N0) % world != 0)
      return fail(INFLX_ERR_SHAPE, "the RCCL all-gather needs equal blocks: %zu %s do not divide by %zu devices (the peer-push gather takes unequal ones)",
                  s0.axis == 0 ? P : N0, s0.axis == 0 ? "parameter rows" : "grid rows", world);
    for (size_t k = 0; k < world; ++k)
      for (size_t j = k + 1; j < world; ++j)
        if (d_full[k] == d_full[j]) return fail(INFLX_ERR_ARG, "the RCCL all-gather needs a buffer per device (buffers %zu and %zu are the same)", k, j);
    std::string why;
    if (!(api = rccl_api(&why))) return fail(INFLX_ERR_DEVICE, "RCCL (librccl.so) could not be loaded: %s", why.c_str());
    if ((rc = ensure_comms(mm, api))) return rc;
  } else if ((rc = ensure_push_streams(mm))) {
    return rc;
  }
  // on every exit from here on -- errors included -- nothing may still be writing into the caller's buffers
  auto drain_all = [&] {
    for (size_t k = 0; k < world; ++k) {
      (void)hipSetDevice(mm->dev[k]->device);
      (void)hipStreamSynchronize(mm->dev[k]->side);
      (void)hipStreamSynchronize(mm->dev[k]->stream);
    }
  };
  // AOS only: a device's slab is then one contiguous piece per parameter row of the (P, N0, N1, K) array
  const size_t row_bytes = N1 * point_bytes;
  for (size_t k = 0; k < world; ++k) {
    const ShardPlan s = shard_plan(P, N0, world, k);
    if (s.p_count == 0 || s.row_count == 0) continue;
    inflx_model* m = mm->dev[k];
    HIP_TRY(hipSetDevice(m->device));
    char* mine = static_cast<char*>(d_full[k]);
    if (s.axis == 0) {
      // parameter rows [p_begin, p_begin + p_count): one contiguous slice; the sweep writes it in place
      char* slice = mine + s.p_begin * N0 * row_bytes;
      rc = inflx_sweep_device(m, op, p + s.p_begin * n_p, s.p_count, n_p, slice, s.p_count * N0 * row_bytes, ss, N0, N1, 0, N0, INFLX_AOS, nullptr);
      if (rc) {
        drain_all();
        return rc;
      }
    } else {
      // grid rows [row_begin, row_begin + row_count) of every parameter row: one sweep per parameter row, each into its place
      for (size_t pr = 0; pr < P; ++pr) {
        char* slice = mine + (pr * N0 + s.row_begin) * row_bytes;
        rc = inflx_sweep_device(m, op, p + pr * n_p, 1, n_p, slice, s.row_count * row_bytes, ss, N0, N1, s.row_begin, s.row_count, INFLX_AOS, nullptr);
        if (rc) {
          drain_all();
          return rc;
        }
      }
    }
    if (gather == INFLX_GATHER_RCCL) continue;  // the collective below runs on every device's sweep stream, behind its sweep
    HIP_TRY(hipEventRecord(mm->swept[k], m->stream));
    for (size_t j = 0; j < world; ++j) {
      if (j == k || d_full[j] == d_full[k]) continue;
      hipStream_t st = mm->push[k][j];
      HIP_TRY(hipStreamWaitEvent(st, mm->swept[k], 0));
      char* theirs = static_cast<char*>(d_full[j]);
      const size_t pieces = s.axis == 0 ? 1 : P;
      for (size_t q = 0; q < pieces; ++q) {
        const size_t off = s.axis == 0 ? s.p_begin * N0 * row_bytes : (q * N0 + s.row_begin) * row_bytes;
        const size_t bytes = s.axis == 0 ? s.p_count * N0 * row_bytes : s.row_count * row_bytes;
        HIP_TRY(hipMemcpyPeerAsync(theirs + off, mm->dev[j]->device, mine + off, m->device, bytes, st));
      }
    }
  }
  if (gather == INFLX_GATHER_RCCL) {
    // In place (rccl.h: sendbuff == recvbuff + rank * sendcount): the device's own block is where the sweep put it; all gathers of
    // all devices fused in one group (rccl_gather_calls has the offsets).
    int r = api->GroupStart();
    if (r != 0) {
      drain_all();
      return fail(INFLX_ERR_DEVICE, "ncclGroupStart failed: %s", api->GetErrorString(r));
    }
    for (const GatherCall& c : rccl_gather_calls(P, N0, row_bytes, world)) {
      char* base = static_cast<char*>(d_full[c.rank]);
      r = api->AllGather(base + c.send_off, base + c.recv_off, c.count, kNcclFloat64, mm->comms[c.rank], mm->dev[c.rank]->stream);
      if (r != 0) break;
    }
    const int rend = api->GroupEnd();
    if (r != 0 || rend != 0) {
      drain_all();  // the sweeps (and whatever part of the collective was enqueued) may still be running: wait before the caller gets its buffers back
      return fail(INFLX_ERR_DEVICE, "%s failed: %s", r != 0 ? "ncclAllGather" : "ncclGroupEnd", api->GetErrorString(r != 0 ? r : rend));
    }
  }
  // synchronous: every device holds the whole result when the call returns
  for (size_t k = 0; k < world; ++k) {
    HIP_TRY(hipSetDevice(mm->dev[k]->device));
    HIP_TRY(hipStreamSynchronize(mm->dev[k]->side));
    HIP_TRY(hipStreamSynchronize(mm->dev[k]->stream));
    if (gather == INFLX_GATHER_PEER_PUSH)
      for (size_t j = 0; j < world; ++j)
        if (mm->push[k][j]) HIP_TRY(hipStreamSynchronize(mm->push[k][j]));
  }
  return INFLX_OK;
}

}  // extern "C"

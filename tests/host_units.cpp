// Unit checks of the HOST-side helpers of libinflx_hip.so that touch memory by hand -- the streaming-store fill of broadcast
// results, the span / stripe arithmetic of the device-to-host transfer, the page-touching helpers, the partition of a
// multi-device sweep, the progress reporter -- built with AddressSanitizer + UBSan on the CPU (tests/test_cabi.py).  No GPU
// is needed: nothing here calls into HIP.  The translation unit includes the library source itself, so that the functions
// of its unnamed namespace are reachable.
#include "../inflatox_amd/csrc/inflx_hip.cpp"

#include <cassert>

#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) {                                                           \
      fprintf(stderr, "host_units: %s failed (line %d)\n", #cond, __LINE__); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

static uint64_t g_state = 0x1234567ull;
static uint64_t rnd() { return (uint64_t)(unit_random(g_state) * 9007199254740992.0); }

int main() {
  // ---- repeat_record: every record size the sweeps have (8 ... 48 bytes), every destination alignment, short and long runs
  for (size_t rec_bytes : {8, 16, 24, 40, 48}) {
    for (int trial = 0; trial < 200; ++trial) {
      const size_t n = trial < 20 ? (size_t)trial : (size_t)(rnd() % 3000);
      const size_t shift = 8 * (rnd() % 8);  // results are arrays of doubles: 8-byte aligned at least
      std::vector<char> buf(n * rec_bytes + 64 + 64, (char)0xEE);
      char* base = buf.data();
      base += (64 - (reinterpret_cast<uintptr_t>(base) & 63)) & 63;
      char* dst = base + shift;
      char rec[48];
      for (size_t k = 0; k < rec_bytes; ++k) rec[k] = (char)(rnd() & 0xff);
      repeat_record(dst, rec, rec_bytes, n);
      _mm_sfence();
      for (size_t q = 0; q < n; ++q) CHECK(memcmp(dst + q * rec_bytes, rec, rec_bytes) == 0);
      CHECK((unsigned char)dst[n * rec_bytes] == 0xEE);  // nothing written behind the run
      if (shift) CHECK((unsigned char)dst[-1] == 0xEE);  // ... nor in front of it
    }
  }
  // ---- stream_copy: any length and alignment
  for (int trial = 0; trial < 400; ++trial) {
    const size_t bytes = 8 * (rnd() % 700);
    const size_t shift = 8 * (rnd() % 8);
    std::vector<char> src(bytes + 8), buf(bytes + 128 + 64, (char)0xEE);
    for (auto& c : src) c = (char)(rnd() & 0xff);
    char* base = buf.data();
    base += (64 - (reinterpret_cast<uintptr_t>(base) & 63)) & 63;
    char* dst = base + 64 + shift;
    stream_copy(dst, src.data(), bytes);
    _mm_sfence();
    CHECK(memcmp(dst, src.data(), bytes) == 0);
    CHECK((unsigned char)dst[bytes] == 0xEE && (unsigned char)dst[-1] == 0xEE);
  }
  // ---- transfer_spans: the spans tile the slab on the device side, land inside the destination, and merge when both sides are contiguous
  for (int trial = 0; trial < 2000; ++trial) {
    const int op = (int)(rnd() % INFLX_OP_COUNT);
    const int layout = (int)(rnd() % 2);
    const size_t P = 1 + rnd() % 5, N1 = 1 + rnd() % 70, rc = 1 + rnd() % 40;
    HostDest d;
    d.dst_rows = rc + rnd() % 30;
    d.dst_row0 = rnd() % (d.dst_rows - rc + 1);
    const std::vector<Span> spans = transfer_spans(op, P, N1, rc, layout, d);
    size_t src_end = 0, total = 0;
    for (const Span& sp : spans) {
      CHECK(sp.src == src_end);  // device side: back to back, in order
      src_end = sp.src + sp.bytes;
      total += sp.bytes;
      CHECK(sp.dst + sp.bytes <= P * d.dst_rows * N1 * kOpBytes[op]);
    }
    CHECK(total == P * rc * N1 * kOpBytes[op]);
    if (d.dst_rows == rc) CHECK(spans.size() == 1);  // the destination IS the slab: one copy
    for (size_t k = 1; k < spans.size(); ++k) CHECK(spans[k].dst >= spans[k - 1].dst + spans[k - 1].bytes);  // disjoint, ascending
  }
  // ---- plane subsets: exactly the requested planes of every parameter row, packed into a destination of that many planes
  for (int trial = 0; trial < 500; ++trial) {
    const int op = (rnd() % 2) ? INFLX_OP_COMPLETE : INFLX_OP_RAW;
    const size_t K = kOpWidth[op], P = 1 + rnd() % 4, N1 = 1 + rnd() % 50, rc = 1 + rnd() % 30;
    HostDest d;
    d.dst_rows = rc;
    d.planes = 1 + rnd() % K;
    d.plane0 = rnd() % (K - d.planes + 1);
    const std::vector<Span> spans = transfer_spans(op, P, N1, rc, INFLX_SOA, d);
    const size_t plane_bytes = rc * N1 * sizeof(double);
    size_t total = 0, dst_end = 0;
    for (const Span& sp : spans) {
      const size_t k0 = (sp.src / plane_bytes) % K;
      CHECK(sp.src % plane_bytes == 0 && sp.bytes % plane_bytes == 0);
      if (d.planes < K) CHECK(k0 >= d.plane0 && k0 + sp.bytes / plane_bytes <= d.plane0 + d.planes);  // (all planes: one span across the parameter rows)
      CHECK(sp.dst == dst_end);  // the destination is filled back to back
      dst_end = sp.dst + sp.bytes;
      total += sp.bytes;
    }
    CHECK(total == P * d.planes * plane_bytes);
    if (d.planes == K) CHECK(spans.size() == 1);
  }
  // ---- the partition of a multi-device sweep covers the index space exactly once
  for (int trial = 0; trial < 3000; ++trial) {
    const size_t P = 1 + rnd() % 40, N0 = 1 + rnd() % 500, world = 1 + rnd() % 9;
    size_t p_seen = 0, r_seen = 0;
    int axis = -1;
    for (size_t k = 0; k < world; ++k) {
      const ShardPlan s = shard_plan(P, N0, world, k);
      if (axis < 0) axis = s.axis;
      CHECK(s.axis == axis && axis == (P >= world ? 0 : 1));
      if (axis == 0) {
        CHECK(s.p_begin == p_seen && s.row_begin == 0 && s.row_count == N0);
        p_seen += s.p_count;
      } else {
        CHECK(s.row_begin == r_seen && s.p_begin == 0 && s.p_count == P);
        r_seen += s.row_count;
      }
    }
    CHECK(axis == 0 ? p_seen == P : r_seen == N0);
    size_t plan[5];
    CHECK(inflx_shard_plan(P, N0, (int)world, (int)(world - 1), plan) == INFLX_OK);
    CHECK(inflx_shard_plan(P, N0, (int)world, (int)world, plan) != INFLX_OK);  // rank out of range
  }
  // ---- page helpers on a fresh mapping: contents preserved, partial pages at both ends left alone
  {
    const size_t bytes = (size_t(3) << 20) + 1234;
    char* p = (char*)mmap(nullptr, bytes + 8192, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    CHECK(p != MAP_FAILED);
    for (size_t k = 0; k < bytes; k += 997) p[100 + k] = (char)(k & 0x7f);
    prefault_range(p + 100, bytes);
    touch_range(p + 100, bytes);
    advise_huge_pages(p + 100, bytes);
    for (size_t k = 0; k < bytes; k += 997) CHECK(p[100 + k] == (char)(k & 0x7f));
    prefault_range(p + 5, 10);  // shorter than a page: nothing to do, nothing touched
    munmap(p, bytes + 8192);
  }
  // ---- parallel_blocks covers [0, n) once, also when n is smaller than the thread count
  for (size_t n : {size_t(0), size_t(1), size_t(3), size_t(1000), size_t(100000)}) {
    std::vector<std::atomic<int>> hits(n);
    for (auto& h : hits) h = 0;
    parallel_blocks(n, 4096, [&](size_t a, size_t b) {
      for (size_t k = a; k < b; ++k) hits[k]++;
    });
    for (size_t k = 0; k < n; ++k) CHECK(hits[k] == 1);
  }
  // ---- tile height of a launch: 1 ... full, never more workgroups than segments, ~1024 workgroups for small launches, the full height for large ones
  {
    const size_t full = 32;
    CHECK(tile_height(full, 1) == 1 && tile_height(full, 256) == 1);              // 256 x 256: one row per workgroup
    CHECK(tile_height(full, 4 * 1000) == 3);                                      // 1000 x 1000 (4 column tiles)
    CHECK(tile_height(full, 8 * 2048) == 16 && tile_height(full, 16 * 4096) == 16);  // 2048^2, 4096^2: half height
    CHECK(tile_height(full, 4 * 16 * 4096) == 32 && tile_height(full, size_t(1) << 40) == 32);
    size_t last = 1;
    for (size_t seg = 1; seg < (size_t(1) << 22); seg += 1 + seg / 37) {
      const size_t th = tile_height(full, seg);
      CHECK(th >= 1 && th <= full && th >= last);  // monotone in the size of the launch
      last = th;
      if (th < full / 2) CHECK(seg / th >= 1024 || th == 1);  // lower tiles only while they are needed to reach ~1024 workgroups
    }
  }
  // ---- the reporter starts and stops cleanly whether or not it ever prints
  {
    Progress pr;
    pr.total = 100;
    Reporter silent(&pr, 1e6, true);  // below INFLX_PROGRESS_MIN_MB: no thread
    CHECK(!silent.active());
    Progress big;
    big.total = uint64_t(8) << 30;
    {
      Reporter r(&big, 1e9, true);
      CHECK(r.active());
      big.done += uint64_t(1) << 30;
    }
    Reporter off(&big, 1e9, false);
    CHECK(!off.active());
  }
  // ---- the in-place RCCL all-gather: (send, recv, count) per rank are the rank's shard_plan block inside its own full-size buffer
  for (int trial = 0; trial < 400; ++trial) {
    const size_t world = 1 + rnd() % 8;
    const bool split_params = rnd() % 2;
    // equal blocks are the entry point's precondition: the split axis divides by the device count
    const size_t P = split_params ? world * (1 + rnd() % 5) : 1 + rnd() % (world > 1 ? world - 1 : 1);
    const size_t N0 = split_params ? 1 + rnd() % 40 : world * (1 + rnd() % 9);
    if (!split_params && P >= world) continue;
    const size_t N1 = 1 + rnd() % 30, K = 6, row_bytes = N1 * K * sizeof(double);
    const std::vector<GatherCall> calls = rccl_gather_calls(P, N0, row_bytes, world);
    const ShardPlan s0 = shard_plan(P, N0, world, 0);
    CHECK(s0.axis == (split_params ? 0 : 1));
    CHECK(calls.size() == (s0.axis == 0 ? 1 : P) * world);
    std::vector<char> covered(P * N0 * row_bytes, 0);  // every byte of the result is some rank's send block of some image, once
    for (const GatherCall& c : calls) {
      const ShardPlan s = shard_plan(P, N0, world, c.rank);
      CHECK(s.axis == s0.axis);
      // where the sweep put rank c.rank's block of image c.image (inflx_sweep_allgather_multi_ex: `slice`)
      const size_t slice = s.axis == 0 ? s.p_begin * N0 * row_bytes : (c.image * N0 + s.row_begin) * row_bytes;
      const size_t bytes = s.axis == 0 ? s.p_count * N0 * row_bytes : s.row_count * row_bytes;
      CHECK(c.send_off == slice && c.count * sizeof(double) == bytes);
      CHECK(c.send_off == c.recv_off + c.rank * c.count * sizeof(double));  // rccl.h: in place means sendbuff == recvbuff + rank * sendcount
      CHECK(c.recv_off + world * bytes <= covered.size());                  // the image lies inside the buffer
      for (size_t b = 0; b < bytes; ++b) CHECK(covered[c.send_off + b]++ == 0);
    }
    for (char v : covered) CHECK(v == 1);
  }
  // ---- one host-thread budget per process: helpers in flight never exceed the budget once several devices are at work
  {
    // 16 CPUs, 8 devices (a GPU box's share, a whole node's GPUs): at most 16 helpers in flight, whichever kind
    const HelperPlan h = helper_plan(16, 8);
    CHECK(8 * h.prefault <= 16 && 8 * h.fill <= 16 && h.prefault >= 1 && h.fill >= 1);
    for (unsigned cpus : {1u, 2u, 3u, 8u, 16u, 64u, 256u})
      for (unsigned dev : {1u, 2u, 3u, 4u, 8u}) {
        const HelperPlan q = helper_plan(cpus, dev);
        CHECK(q.prefault >= 1 && q.fill >= 1 && q.prefault <= 8 && q.fill <= 64);
        if (dev > 1) CHECK(dev * q.fill <= std::max(cpus, dev) && dev * q.prefault <= std::max(cpus, dev));  // (a pipeline always has its own thread)
        if (dev == 1) CHECK(q.fill <= 2 * cpus && q.prefault <= std::max(1u, (cpus + 1) / 2));
      }
    CHECK(helper_plan(16, 1).fill == 32 && helper_plan(16, 1).prefault == 8);  // the measured optimum of a 16-CPU share, one GPU
    // quota files: v2 "max" = none, "1600000 100000" = 16; v1 quota -1 = none
    char dir[] = "/tmp/inflx_quota_XXXXXX";
    CHECK(mkdtemp(dir) != nullptr);
    const std::string d = dir;
    auto put = [&](const char* name, const char* text) {
      FILE* fh = fopen((d + "/" + name).c_str(), "w");
      fputs(text, fh);
      fclose(fh);
    };
    put("cpu.max", "max 100000\n");
    CHECK(quota_from(d + "/cpu.max", "") == 0);
    put("cpu.max", "1600000 100000\n");
    CHECK(quota_from(d + "/cpu.max", "") == 16);
    put("cpu.max", "50000 100000\n");
    CHECK(quota_from(d + "/cpu.max", "") == 1);  // half a CPU is still one thread
    put("cpu.cfs_quota_us", "-1\n");
    put("cpu.cfs_period_us", "100000\n");
    CHECK(quota_from(d + "/cpu.cfs_quota_us", d + "/cpu.cfs_period_us") == 0);
    put("cpu.cfs_quota_us", "800000\n");
    CHECK(quota_from(d + "/cpu.cfs_quota_us", d + "/cpu.cfs_period_us") == 8);
    // a limit on an intermediate ancestor is found on the way up: <d>/a/b/c (none) -> <d>/a (4 CPUs) -> <d> (8 via v1 files above: not looked at by the v2 walk)
    CHECK(system(("mkdir -p " + d + "/a/b/c").c_str()) == 0);
    put("a/cpu.max", "400000 100000\n");
    put("a/b/cpu.max", "max 100000\n");
    put("cpu.max", "1600000 100000\n");
    CHECK(min_quota_upwards(d, "/a/b/c", true) == 4);
    CHECK(min_quota_upwards(d, "/", true) == 16);
    CHECK(min_quota_upwards(d, "/nonexistent/x", true) == 16);  // a path that does not exist below the mount: the root file still counts
    CHECK(system(("rm -rf " + d).c_str()) == 0);
    // the budget is at least one thread and the reported plan is the plan of the budget
    unsigned out[3] = {0, 0, 0};
    CHECK(inflx_host_threads(1, out) == INFLX_OK && out[0] >= 1 && out[1] >= 1 && out[2] >= 1);
    unsigned out8[3];
    CHECK(inflx_host_threads(8, out8) == INFLX_OK && out8[0] == out[0] && 8 * out8[2] <= std::max(out[0], 8u));
    CHECK(inflx_host_threads(1, nullptr) == INFLX_ERR_ARG);
    // run_parts hands every part the number of pipelines at work and restores the caller's view
    std::atomic<unsigned> seen{0};
    CHECK(run_parts(4, [&](size_t) -> int { seen += tl_sharers; return INFLX_OK; }) == INFLX_OK);
    CHECK(seen == 16 && tl_sharers == 1);
  }
  // ---- argument validation of the entry points that need no device
  CHECK(inflx_open(nullptr, 0, nullptr) == INFLX_ERR_ARG);
  {
    uint32_t plan[4];
    float ms;
    CHECK(inflx_sweep_plan_ex(nullptr, 0, 1, 1, 1, 0, INFLX_SWEEP_FORCE_TILE, plan) == INFLX_ERR_ARG);
    CHECK(inflx_sweep_device_ex(nullptr, 0, nullptr, 1, 1, nullptr, 0, nullptr, 1, 1, 0, 1, 0, nullptr, INFLX_SWEEP_FORCE_TILE) == INFLX_ERR_ARG);
    CHECK(inflx_sweep_device_timed_ex(nullptr, 0, nullptr, 1, 1, nullptr, 0, nullptr, 1, 1, 0, 1, 0, nullptr, 1, INFLX_TIME_SINGLE_CALL, 0, &ms) == INFLX_ERR_ARG);
  }
  CHECK(inflx_sweep_host_multi(nullptr, 0, nullptr, 1, 1, nullptr, nullptr, 1, 1, 0, 0, 0) == INFLX_ERR_ARG);
  CHECK(inflx_multi_device_count(nullptr) == 0 && inflx_multi_handle(nullptr, 0) == nullptr);
  inflx_close(nullptr);
  inflx_close_multi(nullptr);
  printf("host_units: all checks passed\n");
  return 0;
}

// Device-to-host copy into PAGEABLE memory: the runtime's own path (hipMemcpyAsync straight into the array) against a pipeline
// of our own -- DMA into a ring of pinned staging buffers, host threads moving each buffer into the array with streaming
// stores while the next DMA runs.  Sizes the host-result path of csrc/inflx_hip.cpp (sweep_host_impl).
//   hipcc -O2 -o staged_d2h.bin staged_d2h.cpp -lpthread ; ./staged_d2h.bin [MiB total]
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void stream_copy(char* dst, const char* src, size_t bytes) {
  size_t done = 0;
  for (; done + 64 <= bytes; done += 64) {
    __m128d a, b, c, d;
    memcpy(&a, src + done, 16), memcpy(&b, src + done + 16, 16), memcpy(&c, src + done + 32, 16), memcpy(&d, src + done + 48, 16);
    _mm_stream_pd((double*)(dst + done), a);
    _mm_stream_pd((double*)(dst + done + 16), b);
    _mm_stream_pd((double*)(dst + done + 32), c);
    _mm_stream_pd((double*)(dst + done + 48), d);
  }
  if (done < bytes) memcpy(dst + done, src + done, bytes - done);
  _mm_sfence();
}

// `threads` movers: buffer k of the ring is split evenly among them
struct Movers {
  std::vector<std::thread> pool;
  std::mutex mu;
  std::condition_variable cv, done_cv;
  const char* src = nullptr;
  char* dst = nullptr;
  size_t bytes = 0;
  uint64_t ticket = 0;
  unsigned pending = 0;
  bool stop = false;
  explicit Movers(unsigned n) {
    for (unsigned t = 0; t < n; ++t)
      pool.emplace_back([this, t, n] {
        uint64_t seen = 0;
        for (;;) {
          const char* s;
          char* d;
          size_t b;
          {
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [&] { return stop || ticket != seen; });
            if (stop) return;
            seen = ticket;
            s = src, d = dst, b = bytes;
          }
          const size_t lo = b * t / n / 64 * 64, hi = t + 1 == n ? b : b * (t + 1) / n / 64 * 64;
          if (hi > lo) stream_copy(d + lo, s + lo, hi - lo);
          {
            std::lock_guard<std::mutex> g(mu);
            if (--pending == 0) done_cv.notify_all();
          }
        }
      });
  }
  void move(char* d, const char* s, size_t b) {  // blocks until done
    std::unique_lock<std::mutex> g(mu);
    src = s, dst = d, bytes = b, pending = (unsigned)pool.size(), ++ticket;
    cv.notify_all();
    done_cv.wait(g, [&] { return pending == 0; });
  }
  ~Movers() {
    {
      std::lock_guard<std::mutex> g(mu);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : pool) t.join();
  }
};

int main(int argc, char** argv) {
  const size_t total = (size_t)(argc > 1 ? atol(argv[1]) : 768) << 20;
  char* dev = nullptr;
  CK(hipMalloc(&dev, total));
  CK(hipMemset(dev, 0x5a, total));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  auto fresh = [&]() {
    char* p = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(p, total, MADV_HUGEPAGE);
    return p;
  };
  auto touch = [&](char* p) {
    std::vector<std::thread> pool;
    for (int t = 0; t < 8; ++t)
      pool.emplace_back([=] {
        for (size_t off = total * t / 8; off < total * (t + 1) / 8; off += 4096) p[off] = 1;
      });
    for (auto& th : pool) th.join();
  };
  // (a) the runtime's path into resident pageable memory
  for (int rep = 0; rep < 3; ++rep) {
    char* host = fresh();
    touch(host);
    const double t0 = now();
    CK(hipMemcpyAsync(host, dev, total, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    const double t = now() - t0;
    printf("runtime, resident pageable destination:            %7.2f ms = %5.1f GB/s (last byte %02x)\n", t * 1e3, total / t / 1e9, (unsigned char)host[total - 1]);
    munmap(host, total);
  }
  // (b) pinned destination (upper bound)
  {
    char* pin = nullptr;
    CK(hipHostMalloc(&pin, total, hipHostMallocDefault));
    for (int rep = 0; rep < 2; ++rep) {
      const double t0 = now();
      CK(hipMemcpyAsync(pin, dev, total, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      const double t = now() - t0;
      printf("runtime, pinned destination:                        %7.2f ms = %5.1f GB/s\n", t * 1e3, total / t / 1e9);
    }
    CK(hipHostFree(pin));
  }
  // (c) staged: ring of pinned buffers + mover threads; the destination is FRESH (pages do not exist) or resident
  for (size_t buf_mb : {4, 8, 16, 32})
    for (unsigned ring : {2u, 4u})
      for (unsigned threads : {4u, 8u, 16u})
        for (int resident = 0; resident < 2; ++resident) {
          const size_t buf = buf_mb << 20;
          std::vector<char*> stage(ring);
          std::vector<hipEvent_t> ev(ring);
          for (unsigned k = 0; k < ring; ++k) {
            CK(hipHostMalloc(&stage[k], buf, hipHostMallocDefault));
            CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
          }
          Movers movers(threads);
          double best = 1e9;
          for (int rep = 0; rep < 3; ++rep) {
            char* host = fresh();
            if (resident) touch(host);
            const double t0 = now();
            const size_t pieces = (total + buf - 1) / buf;
            size_t issued = 0;
            for (; issued < pieces && issued < ring; ++issued) {
              const size_t off = issued * buf, n = std::min(buf, total - off);
              CK(hipMemcpyAsync(stage[issued % ring], dev + off, n, hipMemcpyDeviceToHost, st));
              CK(hipEventRecord(ev[issued % ring], st));
            }
            for (size_t c = 0; c < pieces; ++c) {
              const unsigned k = (unsigned)(c % ring);
              const size_t off = c * buf, n = std::min(buf, total - off);
              CK(hipEventSynchronize(ev[k]));
              movers.move(host + off, stage[k], n);
              if (issued < pieces) {  // the buffer is free again
                const size_t o2 = issued * buf, n2 = std::min(buf, total - o2);
                CK(hipMemcpyAsync(stage[k], dev + o2, n2, hipMemcpyDeviceToHost, st));
                CK(hipEventRecord(ev[k], st));
                ++issued;
              }
            }
            const double t = now() - t0;
            best = std::min(best, t);
            if ((unsigned char)host[total - 1] != 0x5a || (unsigned char)host[total / 2 + 12345] != 0x5a) printf("WRONG DATA\n");
            munmap(host, total);
          }
          printf("staged %2zu MiB x %u, %2u movers, %s destination:  %7.2f ms = %5.1f GB/s\n", buf_mb, ring, threads, resident ? "resident" : "fresh   ", best * 1e3, total / best / 1e9);
          for (unsigned k = 0; k < ring; ++k) {
            CK(hipHostFree(stage[k]));
            CK(hipEventDestroy(ev[k]));
          }
        }
  return 0;
}

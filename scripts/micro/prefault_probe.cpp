// How fast can fresh anonymous memory (what np.zeros hands to complete_analysis) be made resident?
// Measures, for an 805 MB mapping: MADV_POPULATE_WRITE and per-page touching, with and without
// MADV_HUGEPAGE, over 1..16 threads.   build: g++ -O2 -pthread prefault_probe.cpp -o prefault_probe
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t bytes = size_t(805) << 20;
  FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
  char line[256] = "?";
  if (f) { if (!fgets(line, sizeof line, f)) line[0] = 0; fclose(f); }
  printf("THP: %s", line);
  for (int huge = 0; huge < 2; ++huge)
    for (int mode = 0; mode < 2; ++mode)
      for (int threads : {1, 2, 4, 8, 16}) {
        char* p = (char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) return 1;
        if (huge) madvise(p, bytes, MADV_HUGEPAGE);
        const double t0 = now();
        std::vector<std::thread> pool;
        int rc_all = 0;
        for (int t = 0; t < threads; ++t) {
          char* a = p + bytes / threads * t;
          char* b = t + 1 == threads ? p + bytes : p + bytes / threads * (t + 1);
          pool.emplace_back([=, &rc_all] {
            if (mode == 0) {
#ifdef MADV_POPULATE_WRITE
              if (madvise(a, b - a, MADV_POPULATE_WRITE) != 0) rc_all = 1;
#else
              rc_all = 2;
#endif
            } else {
              for (volatile char* q = a; q < b; q += 4096) *q = 1;
            }
          });
        }
        for (auto& th : pool) th.join();
        const double t1 = now();
        munmap(p, bytes);
        const double t2 = now();
        printf("hugepage=%d %s threads=%2d: populate %7.2f ms (%5.1f GB/s)  munmap %6.2f ms  rc=%d\n", huge, mode == 0 ? "POPULATE_WRITE" : "touch         ", threads,
               (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, rc_all);
      }
  return 0;
}

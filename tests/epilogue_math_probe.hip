// TEST INFRASTRUCTURE: the epilogue's specialised atan / tan (csrc/inflx_ops.h) against OCML's general entry points,
// bit for bit, on the device.  usage: epilogue_math_probe [millions of random arguments]; prints mismatch counts.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "inflx_ops.h"

__global__ void probe(const double* t, size_t n, unsigned long long* bad_atan, unsigned long long* bad_tan, double* first_bad, unsigned long long* sqrt_counts) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = t[i];
  const double a_ref = atan(x), a_new = inflx_atan_nonneg(x);
  bool a_ok = __double_as_longlong(a_ref) == __double_as_longlong(a_new) || (a_ref != a_ref && a_new != a_new);
  // the variant the quick epilogue uses (1/t without special-case handling) on the range the epilogue guarantees
  const bool mid = x >= 0x1p-200 && x <= 0x1p200;
  const double a_quick = inflx_atan_nonneg<true>(mid ? x : 1.0);
  a_ok = a_ok && (!mid || __double_as_longlong(a_ref) == __double_as_longlong(a_quick));
  if (!a_ok && atomicAdd(bad_atan, 1ull) == 0) { first_bad[0] = x; first_bad[1] = a_ref; first_bad[2] = a_new; }
  // the quick epilogue's square root (no operand scaling, no zero / infinity selection) against the compiler's, wherever its
  // guard accepts the argument: x itself, -x, a random bit pattern, and magnitudes around the guard's lower bound 2^-767
  {
    unsigned long long hsh = (unsigned long long)__double_as_longlong(x) * 0x9E3779B97F4A7C15ull + i;
    hsh ^= hsh >> 29;
    const double frac = 1.0 + (double)(hsh & ((1ull << 52) - 1)) * 0x1p-52;
    const double z[5] = {x, -x, __longlong_as_double((long long)(hsh * 0xBF58476D1CE4E5B9ull)), ldexp(frac, -769 + (int)(i % 5)), ldexp(frac, 1019 + (int)(i % 5))};
    for (int k = 0; k < 5; ++k) {
      if (!inflx_sqrt_quick_ok(z[k])) continue;
      atomicAdd(sqrt_counts + 1, 1ull);
      const double want = sqrt(z[k]), got = inflx_sqrt_quick(z[k]);
      const bool same = __double_as_longlong(want) == __double_as_longlong(got) || (want != want && got != got);
      if (!same && atomicAdd(sqrt_counts, 1ull) == 0) { first_bad[6] = z[k]; first_bad[7] = want; first_bad[8] = got; }
    }
    // the guard must refuse what the plain spelling cannot do: zeros, denormals, infinities, NaNs
    const double refuse[6] = {0.0, -0.0, 5e-324, 0x1p-768, __builtin_inf(), __builtin_nan("")};
    for (int k = 0; k < 6; ++k)
      if (inflx_sqrt_quick_ok(refuse[k])) atomicAdd(sqrt_counts, 1ull);
  }
  // tan at delta = atan(x) (what the epilogue feeds it) and at x itself when 0 <= x <= pi/2
  const double d[2] = {a_ref, (x >= 0.0 && x <= 0x1.921fb54442d18p+0) ? x : a_ref};
  for (int k = 0; k < 2; ++k) {
    const double t_ref = tan(d[k]), t_new = inflx_tan_quadrant1(d[k]);
    const bool t_ok = __double_as_longlong(t_ref) == __double_as_longlong(t_new) || (t_ref != t_ref && t_new != t_new);
    if (!t_ok && atomicAdd(bad_tan, 1ull) == 0) { first_bad[3] = d[k]; first_bad[4] = t_ref; first_bad[5] = t_new; }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
int main(int argc, char** argv) {
  const size_t millions = argc > 1 ? (size_t)atol(argv[1]) : 16;
  const size_t n = millions * 1000000;
  std::vector<double> h(n);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  const double special[] = {0.0, 5e-324, 2.2250738585072014e-308, 1e-300, 0x1p-60, 0.5, 0x1.fffffffffffffp-1, 1.0, 0x1.0000000000001p+0, 2.0, 0x1p60, 1e300,
                            1.7976931348623157e308, INFINITY, NAN, 0x1.921fb54442d18p-1, 0x1.921fb54442d18p+0, 0x1.921fb54442d17p+0, 0x1.921fb54442d19p-1};
  for (size_t i = 0; i < n; ++i) {
    const uint64_t m = rnd() & ((1ull << 52) - 1);
    const unsigned kind = (unsigned)(rnd() % 8);
    double v;
    if (i < sizeof special / sizeof special[0]) v = special[i];
    else if (kind < 3) v = std::ldexp(1.0 + (double)m * 0x1p-52, (int)(rnd() % 121) - 60);          // wide range of |v10/v00|
    else if (kind < 5) v = std::ldexp(1.0 + (double)m * 0x1p-52, (int)(rnd() % 7) - 3);              // around 1
    else if (kind < 7) v = (double)m * 0x1p-52 * 0x1.921fb54442d18p+0;                                // delta uniform in [0, pi/2)
    else v = 0x1.921fb54442d18p+0 - std::ldexp((double)m * 0x1p-52, -(int)(rnd() % 50));             // next to pi/2
    h[i] = v;
  }
  double *d_t, *d_first;
  unsigned long long *d_bad;
  CK(hipMalloc(&d_t, n * 8)); CK(hipMalloc(&d_bad, 32)); CK(hipMalloc(&d_first, 72));
  CK(hipMemcpy(d_t, h.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemset(d_bad, 0, 32)); CK(hipMemset(d_first, 0, 72));
  // twice: in random order (the lanes of a wavefront disagree about every branch) and sorted (wave-uniform arguments:
  // the scalar branches of the specialised functions skip what no lane needs) -- same bits either way
  probe<<<(unsigned)((n + 255) / 256), 256>>>(d_t, n, d_bad, d_bad + 1, d_first, d_bad + 2);
  CK(hipDeviceSynchronize());
  std::sort(h.begin(), h.end(), [](double a, double b) { return (a == a) && (!(b == b) || a < b); });  // NaNs last
  CK(hipMemcpy(d_t, h.data(), n * 8, hipMemcpyHostToDevice));
  probe<<<(unsigned)((n + 255) / 256), 256>>>(d_t, n, d_bad, d_bad + 1, d_first, d_bad + 2);
  CK(hipDeviceSynchronize());
  unsigned long long bad[4]; double first[9];
  CK(hipMemcpy(bad, d_bad, 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(first, d_first, 72, hipMemcpyDeviceToHost));
  printf("%zu arguments (random order + sorted): atan mismatches %llu, tan mismatches %llu\n", n, bad[0], bad[1]);
  printf("quick sqrt: %llu accepted arguments, mismatches %llu\n", bad[3], bad[2]);
  if (bad[2]) printf("  first sqrt mismatch: x=%a compiler=%a ours=%a\n", first[6], first[7], first[8]);
  if (bad[0]) printf("  first atan mismatch: x=%a ocml=%a ours=%a\n", first[0], first[1], first[2]);
  if (bad[1]) printf("  first tan mismatch: x=%a ocml=%a ours=%a\n", first[3], first[4], first[5]);
  return (bad[0] || bad[1] || bad[2] || bad[3] == 0) ? 1 : 0;
}

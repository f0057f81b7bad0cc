// Microbenchmark (MI355X): issue cost of the FP64 VALU instructions the sweep kernels are made of, relative to
// v_fma_f64, and the accuracy of the hardware reciprocal / reciprocal square root seeds.  Decides how an IEEE
// double division should be spelled in the per-point stage.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/valu_rates.hip -o scripts/micro/valu_rates.bin && scripts/micro/valu_rates.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIter = 2048;   // loop trips
constexpr int kChains = 8;    // independent dependency chains per lane

// One kernel per instruction: kChains independent chains, kIter trips, the asm body given as a macro.
#define RATE_KERNEL(NAME, ASM, ...)                                                                      \
  __global__ __launch_bounds__(256) void k_##NAME(double* out, double seed, int n) {                     \
    double x[kChains];                                                                                   \
    for (int c = 0; c < kChains; ++c) x[c] = seed + 0.001 * (threadIdx.x + c);                           \
    double y = 1.0000001 + 1e-9 * threadIdx.x, z = 0.999999;                                             \
    int e = 1;                                                                                           \
    (void)y; (void)z; (void)e;                                                                           \
    for (int i = 0; i < n; ++i) {                                                                        \
      _Pragma("unroll") for (int c = 0; c < kChains; ++c) { asm volatile(ASM : "+v"(x[c]) : __VA_ARGS__); } \
    }                                                                                                    \
    double s = 0;                                                                                        \
    for (int c = 0; c < kChains; ++c) s += x[c];                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                      \
  }

RATE_KERNEL(fma_f64, "v_fma_f64 %0, %0, %1, %2", "v"(y), "v"(z))
RATE_KERNEL(mul_f64, "v_mul_f64 %0, %0, %1", "v"(y))
RATE_KERNEL(add_f64, "v_add_f64 %0, %0, %1", "v"(y))
RATE_KERNEL(rcp_f64, "v_rcp_f64_e32 %0, %0", "v"(y))
RATE_KERNEL(rsq_f64, "v_rsq_f64_e32 %0, %0", "v"(y))
RATE_KERNEL(sqrt_f64, "v_sqrt_f64_e32 %0, %0", "v"(y))
RATE_KERNEL(div_scale_f64, "v_div_scale_f64 %0, vcc, %0, %1, %0", "v"(y) : "vcc")
RATE_KERNEL(div_fmas_f64, "v_div_fmas_f64 %0, %0, %1, %2", "v"(y), "v"(z) : "vcc")
RATE_KERNEL(div_fixup_f64, "v_div_fixup_f64 %0, %0, %1, %2", "v"(y), "v"(z))
RATE_KERNEL(cmp_class_f64, "v_cmp_class_f64_e32 vcc, %0, %1", "v"(e) : "vcc")
RATE_KERNEL(cmp_ge_f64, "v_cmp_ge_f64_e32 vcc, %0, %1", "v"(y) : "vcc")
RATE_KERNEL(cmp_u_f64, "v_cmp_u_f64_e32 vcc, %0, %1", "v"(y) : "vcc")
RATE_KERNEL(ldexp_f64, "v_ldexp_f64 %0, %0, %1", "v"(e))
RATE_KERNEL(frexp_mant_f64, "v_frexp_mant_f64_e32 %0, %0", "v"(y))
RATE_KERNEL(trig_preop_f64, "v_trig_preop_f64 %0, %0, %1", "v"(e))
RATE_KERNEL(fract_f64, "v_fract_f64_e32 %0, %0", "v"(y))
RATE_KERNEL(rndne_f64, "v_rndne_f64_e32 %0, %0", "v"(y))
RATE_KERNEL(mov_b64, "v_mov_b64_e32 %0, %1", "v"(y))
RATE_KERNEL(max_f64, "v_max_f64 %0, %0, %1", "v"(y))

// 32-bit instructions on the low half of the register pair
#define RATE_KERNEL32(NAME, ASM, ...)                                                                    \
  __global__ __launch_bounds__(256) void k_##NAME(double* out, double seed, int n) {                     \
    float x[kChains];                                                                                    \
    for (int c = 0; c < kChains; ++c) x[c] = (float)seed + 0.001f * (threadIdx.x + c);                   \
    float y = 1.0000001f, z = 0.999999f;                                                                 \
    (void)y; (void)z;                                                                                    \
    for (int i = 0; i < n; ++i) {                                                                        \
      _Pragma("unroll") for (int c = 0; c < kChains; ++c) { asm volatile(ASM : "+v"(x[c]) : __VA_ARGS__); } \
    }                                                                                                    \
    float s = 0;                                                                                         \
    for (int c = 0; c < kChains; ++c) s += x[c];                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                      \
  }
RATE_KERNEL32(fma_f32, "v_fma_f32 %0, %0, %1, %2", "v"(y), "v"(z))
RATE_KERNEL32(rcp_f32, "v_rcp_f32_e32 %0, %0", "v"(y))
RATE_KERNEL32(mov_b32, "v_mov_b32_e32 %0, %1", "v"(y))
RATE_KERNEL32(cndmask_b32, "v_cndmask_b32_e32 %0, %0, %1, vcc", "v"(y) : "vcc")
RATE_KERNEL32(and_b32, "v_and_b32_e32 %0, %0, %1", "v"(y))

// v_cndmask_b32 in the forms the compiler emits for an f64 select (two per select)
RATE_KERNEL32(cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", "v"(y) : "s20", "s21")
__global__ __launch_bounds__(256) void k_select_f64(double* out, double seed, int n) {
  // x = (x > y) ? x*z : x  -- compare + two v_cndmask + one multiply per chain and trip
  double x[kChains];
  for (int c = 0; c < kChains; ++c) x[c] = seed + 0.001 * (threadIdx.x + c);
  double y = 1.0000001 + 1e-9 * threadIdx.x, z = 0.999999;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
      const double t = x[c] * z;
      x[c] = x[c] > y ? t : x[c];
      asm volatile("" : "+v"(x[c]), "+v"(y), "+v"(z));
    }
  }
  double s = 0;
  for (int c = 0; c < kChains; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the same without the select (multiply only): the difference is the cost of compare + 2 cndmask
__global__ __launch_bounds__(256) void k_noselect_f64(double* out, double seed, int n) {
  double x[kChains];
  for (int c = 0; c < kChains; ++c) x[c] = seed + 0.001 * (threadIdx.x + c);
  double y = 1.0000001 + 1e-9 * threadIdx.x, z = 0.999999;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
      x[c] = x[c] * z;
      asm volatile("" : "+v"(x[c]), "+v"(y), "+v"(z));
    }
  }
  double s = 0;
  for (int c = 0; c < kChains; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// quotient by a denominator whose reciprocal is known (hoisted): 3 instructions + ONE range check
__global__ __launch_bounds__(256) void k_div_hoisted1(double* out, double seed, int n) {
  double x[kChains];
  for (int c = 0; c < kChains; ++c) x[c] = seed + 0.001 * (threadIdx.x + c);
  double y = 1.0000001 + 1e-9 * threadIdx.x;
  const double r = 1.0 / y;
  bool ok = true;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
      const double q0 = x[c] * r;
      const double rem = __builtin_fma(-y, q0, x[c]);
      const double q = __builtin_fma(rem, r, q0);
      ok = ok && (__builtin_fabs(q) >= 0x1p-400);
      x[c] = q;
      asm volatile("" : "+v"(x[c]), "+v"(y));
    }
  }
  double s = ok ? 0 : 1;
  for (int c = 0; c < kChains; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- whole divisions --------------------------------------------------------------------------------
// IEEE a/b as the compiler spells it (11 instructions: 2 div_scale, rcp, 6 fma/mul, div_fmas, div_fixup)
__global__ __launch_bounds__(256) void k_div_ieee(double* out, double seed, int n) {
  double x[kChains];
  for (int c = 0; c < kChains; ++c) x[c] = seed + 0.001 * (threadIdx.x + c);
  double y = 1.0000001 + 1e-9 * threadIdx.x;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < kChains; ++c) { x[c] = x[c] / y; asm volatile("" : "+v"(x[c]), "+v"(y)); }
  }
  double s = 0;
  for (int c = 0; c < kChains; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same arithmetic without operand scaling and fix-up, plus the three checks that decide whether the
// result can be trusted (normal quotient, numerator not tiny, or a NaN operand)
__device__ __forceinline__ double div_fast(double a, double b, bool& ok) {
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  const double q0 = a * r;
  const double rem = __builtin_fma(-b, q0, a);
  const double q = __builtin_fma(rem, r, q0);
  ok = ok && ((__builtin_isnormal(q) && __builtin_fabs(a) >= 0x1p-960) || __builtin_isunordered(a, b));
  return q;
}
__global__ __launch_bounds__(256) void k_div_fast(double* out, double seed, int n) {
  double x[kChains];
  for (int c = 0; c < kChains; ++c) x[c] = seed + 0.001 * (threadIdx.x + c);
  double y = 1.0000001 + 1e-9 * threadIdx.x;
  bool ok = true;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < kChains; ++c) { x[c] = div_fast(x[c], y, ok); asm volatile("" : "+v"(x[c]), "+v"(y)); }
  }
  double s = ok ? 0 : 1;
  for (int c = 0; c < kChains; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// eight quotients by ONE denominator: reciprocal refined once, three instructions + checks per quotient
__global__ __launch_bounds__(256) void k_div_shared(double* out, double seed, int n) {
  double x[kChains];
  for (int c = 0; c < kChains; ++c) x[c] = seed + 0.001 * (threadIdx.x + c);
  double y = 1.0000001 + 1e-9 * threadIdx.x;
  bool ok = true;
  for (int i = 0; i < n; ++i) {
    double r = __builtin_amdgcn_rcp(y);
    double e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
      const double q0 = x[c] * r;
      const double rem = __builtin_fma(-y, q0, x[c]);
      const double q = __builtin_fma(rem, r, q0);
      ok = ok && ((__builtin_isnormal(q) && __builtin_fabs(x[c]) >= 0x1p-960) || __builtin_isunordered(x[c], y));
      x[c] = q;
      asm volatile("" : "+v"(x[c]), "+v"(y));
    }
  }
  double s = ok ? 0 : 1;
  for (int c = 0; c < kChains; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- accuracy of the seeds and equality of the fast division with the IEEE one ----------------------------
__global__ void k_accuracy(const double* a, const double* b, int n, double* rcp_out, double* rsq_out, unsigned long long* mismatches, unsigned long long* flagged) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  rcp_out[i] = __builtin_amdgcn_rcp(b[i]);
  rsq_out[i] = __builtin_amdgcn_rsq(fabs(b[i]));
  bool ok = true;
  const double q = div_fast(a[i], b[i], ok);
  const double want = a[i] / b[i];
  if (!ok) atomicAdd(flagged, 1ull);
  else if (!(q == want || (q != q && want != want))) atomicAdd(mismatches, 1ull);
}

typedef void (*kern_t)(double*, double, int);
struct Case { const char* name; kern_t fn; int per_trip; };

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %.0f MHz\n", prop.gcnArchName, cus, prop.clockRate / 1e3);
  double* out;
  CK(hipMalloc(&out, sizeof(double) * 256 * cus * 8));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  std::vector<Case> cases = {
#define C(n) {#n, k_##n, kChains}
      C(fma_f64), C(mul_f64), C(add_f64), C(max_f64), C(mov_b64), C(rcp_f64), C(rsq_f64), C(sqrt_f64), C(div_scale_f64), C(div_fmas_f64), C(div_fixup_f64),
      C(cmp_class_f64), C(cmp_ge_f64), C(cmp_u_f64), C(ldexp_f64), C(frexp_mant_f64), C(trig_preop_f64), C(fract_f64), C(rndne_f64),
      C(fma_f32), C(rcp_f32), C(mov_b32), C(cndmask_b32), C(cndmask_sgpr), C(and_b32), C(select_f64), C(noselect_f64), C(div_hoisted1), C(div_ieee), C(div_fast), C(div_shared)};
  double fma_ns[3] = {0, 0, 0};
  for (int wi = 0; wi < 3; ++wi) {
    const int waves_per_simd = wi == 0 ? 1 : (wi == 1 ? 2 : 4);
    printf("\n== %d wave(s) per SIMD ==\n%-16s %12s %14s %10s\n", waves_per_simd, "instruction", "ns/wave-inst", "cyc@2.4GHz", "vs fma64");
    for (const Case& c : cases) {
      const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
      c.fn<<<blocks, 256>>>(out, 1.5, 64);
      CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(t0));
        c.fn<<<blocks, 256>>>(out, 1.5, kIter);
        CK(hipEventRecord(t1));
        CK(hipEventSynchronize(t1));
        float ms;
        CK(hipEventElapsedTime(&ms, t0, t1));
        best = ms < best ? ms : best;
      }
      // per SIMD: waves_per_simd waves each issue kIter*per_trip instructions (or divisions)
      const double ns = best * 1e6 / ((double)kIter * c.per_trip * waves_per_simd);
      if (!strcmp(c.name, "fma_f64")) fma_ns[wi] = ns;
      printf("%-16s %12.3f %14.2f %10.2f\n", c.name, ns, ns * 2.4, ns / fma_ns[wi]);
    }
  }

  // accuracy
  const int n = 1 << 22;
  std::vector<double> ha(n), hb(n);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (int i = 0; i < n; ++i) {
    // mantissas uniform, exponents in [-300, 300]
    auto mk = [&] {
      const uint64_t m = rnd() & ((1ull << 52) - 1);
      const int e = (int)(rnd() % 601) - 300;
      double d = std::ldexp(1.0 + (double)m * 0x1p-52, e);
      return (rnd() & 1) ? d : -d;
    };
    ha[i] = mk();
    hb[i] = mk();
  }
  double *da, *db, *dr, *dq;
  unsigned long long *dm, *df;
  CK(hipMalloc(&da, n * 8)); CK(hipMalloc(&db, n * 8)); CK(hipMalloc(&dr, n * 8)); CK(hipMalloc(&dq, n * 8));
  CK(hipMalloc(&dm, 8)); CK(hipMalloc(&df, 8));
  CK(hipMemset(dm, 0, 8)); CK(hipMemset(df, 0, 8));
  CK(hipMemcpy(da, ha.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), n * 8, hipMemcpyHostToDevice));
  k_accuracy<<<n / 256, 256>>>(da, db, n, dr, dq, dm, df);
  CK(hipDeviceSynchronize());
  std::vector<double> hr(n), hq(n);
  unsigned long long mism = 0, flag = 0;
  CK(hipMemcpy(hr.data(), dr, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hq.data(), dq, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&mism, dm, 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&flag, df, 8, hipMemcpyDeviceToHost));
  double worst_rcp = 0, worst_rsq = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / (long double)hb[i];
    worst_rcp = std::fmax(worst_rcp, (double)fabsl(((long double)hr[i] - t) / t));
    const long double u = 1.0L / sqrtl(fabsl((long double)hb[i]));
    worst_rsq = std::fmax(worst_rsq, (double)fabsl(((long double)hq[i] - u) / u));
  }
  printf("\nv_rcp_f64: max relative error %.3e (2^%.1f); v_rsq_f64: %.3e (2^%.1f) over %d random operands\n", worst_rcp, std::log2(worst_rcp), worst_rsq,
         std::log2(worst_rsq), n);
  printf("div_fast vs a/b on %d random pairs: %llu mismatches, %llu flagged for the IEEE path\n", n, mism, flag);
  return 0;
}

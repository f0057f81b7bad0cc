// Do the sticky FP exception bits of TRAPSTS accumulate with traps disabled, for which instructions, and can a wavefront
// read them right after the instruction that raised them?   hipcc --offload-arch=gfx950 -O2 fp_flags.hip -o fp_flags.bin
//
// TRAPSTS.EXCP[5:0]: 0 invalid, 1 input denormal, 2 division by zero, 3 overflow, 4 underflow, 5 inexact.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define CLEAR_FLAGS() asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 6), 0" ::: "memory")

__device__ inline unsigned read_flags_after(double v) {
  unsigned f;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 6)" : "=s"(f) : "v"(v) : "memory");
  return f;
}
__device__ inline unsigned read_flags_after_nops(double v) {
  unsigned f;
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 6)" : "=s"(f) : "v"(v) : "memory");
  return f;
}

enum Op { MUL, ADD, FMA1, RCP, RSQ, SQRT, DIVSCALE_CHAIN, RCP_F32, N_OPS };

__device__ inline double apply(int op, double a, double b) {
  double r;
  switch (op) {
    case MUL: asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); break;
    case ADD: asm volatile("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); break;
    case FMA1: asm volatile("v_fma_f64 %0, %1, %2, 1.0" : "=v"(r) : "v"(a), "v"(b)); break;
    case RCP: asm volatile("v_rcp_f64 %0, %1" : "=v"(r) : "v"(a)); break;
    case RSQ: asm volatile("v_rsq_f64 %0, %1" : "=v"(r) : "v"(a)); break;
    case SQRT: asm volatile("v_sqrt_f64 %0, %1" : "=v"(r) : "v"(a)); break;
    default: r = a; break;
  }
  return r;
}

// one case per record: (op, a, b, lane) -> the operation runs with (a, b) in `lane` only (1.0, 1.0 elsewhere; lane < 0: every
// lane); out: flags read immediately, flags read after 16 nops, result of `lane`
struct Case {
  int op;
  int lane;
  double a, b;
};

__global__ void probe(const Case* cases, int n, unsigned* flags_now, unsigned* flags_later, double* result) {
  for (int c = 0; c < n; ++c) {
    const Case k = cases[c];
    const bool mine = k.lane < 0 || (int)threadIdx.x == k.lane;
    const double a = mine ? k.a : 1.0, b = mine ? k.b : 1.0;
    CLEAR_FLAGS();
    const double r = apply(k.op, a, b);
    const unsigned f0 = read_flags_after(r);
    const unsigned f1 = read_flags_after_nops(r);
    if (mine && (k.lane >= 0 || threadIdx.x == 0)) result[c] = r;
    if (threadIdx.x == 0) {
      flags_now[c] = f0;
      flags_later[c] = f1;
    }
  }
}

// an overflow in an EXEC-masked lane must not be seen
__global__ void masked(unsigned* out, double big) {
  CLEAR_FLAGS();
  double r = 1.0;
  if (threadIdx.x == 5) r = apply(MUL, big, big);
  const unsigned f_after_branch = read_flags_after_nops(r);
  CLEAR_FLAGS();
  double q = 1.0;
  if (threadIdx.x > 100) q = apply(MUL, big, big);  // no lane
  const unsigned f_none = read_flags_after_nops(q);
  if (threadIdx.x == 0) {
    out[0] = f_after_branch;
    out[1] = f_none;
  }
}

// hazard hunt: the raising instruction directly before the read, many times, with every flag-free filler length
__global__ void hazard(unsigned* missed, double big, int rounds) {
  unsigned miss = 0;
  for (int i = 0; i < rounds; ++i) {
    CLEAR_FLAGS();
    double r;
    unsigned f;
    // the last VALU instruction before the read raises overflow in one lane only
    const double x = ((int)threadIdx.x == (i & 63)) ? big : 1.0;
    asm volatile("v_mul_f64 %0, %2, %2\n\ts_getreg_b32 %1, hwreg(HW_REG_TRAPSTS, 0, 6)" : "=&v"(r), "=s"(f) : "v"(x) : "memory");
    if (!(f & 8u)) ++miss;
    // ... and behind a transcendental
    CLEAR_FLAGS();
    const double z = ((int)threadIdx.x == (i & 63)) ? 0.0 : 1.0;
    asm volatile("v_rcp_f64 %0, %2\n\ts_getreg_b32 %1, hwreg(HW_REG_TRAPSTS, 0, 6)" : "=&v"(r), "=s"(f) : "v"(z) : "memory");
    if (!(f & 4u)) miss += 0x10000u;
  }
  if (threadIdx.x == 0) missed[blockIdx.x] = miss;
}

int main() {
  const double inf = INFINITY, qnan = NAN, den = 5e-324, tiny = 0x1p-1000, big = 1e300;
  const Case host[] = {
      {MUL, -1, 2.0, 3.0},          // exact: nothing (or nothing but nothing)
      {MUL, -1, 1.1, 1.3},          // inexact only
      {MUL, -1, big, big},          // overflow
      {MUL, 17, big, big},          // overflow in one lane
      {MUL, -1, 1e-200, 1e-200},    // underflow to zero, inexact
      {MUL, -1, tiny, 0x1p-60},     // exact subnormal result 2^-1060
      {MUL, -1, tiny * 1.1, 0x1.3p-60},  // inexact subnormal result
      {MUL, -1, den, 2.0},          // denormal input, exact denormal result
      {MUL, -1, 0x1p-1060, 0x1p200},  // denormal input, normal exact result
      {MUL, -1, inf, 0.0},          // invalid
      {MUL, -1, qnan, 2.0},         // quiet NaN operand: no flag expected
      {ADD, -1, inf, -inf},         // invalid
      {ADD, -1, big, 1.0},          // inexact only
      {FMA1, -1, inf, 0.0},         // invalid
      {FMA1, -1, 1e200, 1e200},     // overflow
      {RCP, -1, 0.0, 0},            // division by zero
      {RCP, -1, -0.0, 0},
      {RCP, -1, den, 0},            // 1/denormal = overflow
      {RCP, -1, 0x1p-1023, 0},      // subnormal input, finite reciprocal 2^1023
      {RCP, -1, 0x1p1023, 0},       // exact subnormal result
      {RCP, -1, 0x1.8p1023, 0},     // inexact subnormal result
      {RCP, -1, inf, 0},            // 0, no flag expected
      {RCP, -1, qnan, 0},
      {RCP, -1, 3.0, 0},            // inexact only
      {RSQ, -1, 0.0, 0},            // division by zero
      {RSQ, -1, -1.0, 0},           // invalid
      {RSQ, -1, inf, 0},            // 0
      {RSQ, -1, den, 0},
      {RSQ, -1, 2.0, 0},
      {SQRT, -1, -1.0, 0},
      {SQRT, -1, -0.0, 0},
      {SQRT, -1, 4.0, 0},
  };
  const int n = sizeof(host) / sizeof(host[0]);
  Case* cases;
  unsigned *f0, *f1, *m;
  double* res;
  hipMalloc(&cases, sizeof(host));
  hipMalloc(&f0, n * 4);
  hipMalloc(&f1, n * 4);
  hipMalloc(&res, n * 8);
  hipMalloc(&m, 4096);
  hipMemcpy(cases, host, sizeof(host), hipMemcpyHostToDevice);
  hipMemset(res, 0, n * 8);
  probe<<<1, 64>>>(cases, n, f0, f1, res);
  unsigned h0[64], h1[64];
  double hr[64];
  hipMemcpy(h0, f0, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(h1, f1, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hr, res, n * 8, hipMemcpyDeviceToHost);
  const char* names[] = {"mul", "add", "fma+1", "rcp", "rsq", "sqrt"};
  printf("flags: 1 invalid, 2 input denormal, 4 div by zero, 8 overflow, 16 underflow, 32 inexact\n");
  for (int c = 0; c < n; ++c)
    printf("%-6s lane %3d a=%-24.17g b=%-24.17g -> %-24.17g flags now 0x%02x later 0x%02x\n", names[host[c].op], host[c].lane, host[c].a, host[c].b, hr[c], h0[c],
           h1[c]);
  masked<<<1, 64>>>(m, big);
  unsigned hm[2];
  hipMemcpy(hm, m, 8, hipMemcpyDeviceToHost);
  printf("overflow in lane 5 inside a branch: 0x%02x; in no lane: 0x%02x\n", hm[0], hm[1]);
  const int blocks = 512;
  hazard<<<blocks, 64>>>(m, big, 4096);
  unsigned hz[512];
  hipMemcpy(hz, m, blocks * 4, hipMemcpyDeviceToHost);
  unsigned long miss_mul = 0, miss_rcp = 0;
  for (int b = 0; b < blocks; ++b) {
    miss_mul += hz[b] & 0xffffu;
    miss_rcp += hz[b] >> 16;
  }
  printf("read directly behind the raising instruction, %d reads each: missed after v_mul_f64 %lu, after v_rcp_f64 %lu\n", blocks * 4096, miss_mul, miss_rcp);
  printf("hip status: %s\n", hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}

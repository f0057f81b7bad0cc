// TEST INFRASTRUCTURE: delta and eta_parallel the way src/anguelova.rs:128,132 spells them, with OCML's general atan / tan
// as the libm -- the device-side counterpart of the reference's `(b / a).abs().atan()` and `omega * delta.tan() - 3.0`.
// usage: ocml_sweep_probe IN OUT; IN holds n records (v00, v10, omega) of doubles, OUT receives n records (delta, eta).
// tests/test_parity_gpu.py feeds it the model values and omega a DEFAULT build's sweep produced and demands bit equality
// of that sweep's delta and eta: the default build computes OCML's tan of OCML's atan, nothing else.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void probe(const double* in, size_t n, double* out) {
#pragma clang fp contract(off)  // rustc never contracts
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = in[3 * i], b = in[3 * i + 1], omega = in[3 * i + 2];
  const double delta = atan(fabs(b / a));
  out[2 * i] = delta;
  out[2 * i + 1] = omega * tan(delta) - 3.0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  fseek(f, 0, SEEK_END);
  const size_t n = (size_t)ftell(f) / (3 * sizeof(double));
  fseek(f, 0, SEEK_SET);
  std::vector<double> h(3 * n), o(2 * n);
  if (fread(h.data(), sizeof(double), 3 * n, f) != 3 * n) return 2;
  fclose(f);
  double *d_in, *d_out;
  CK(hipMalloc(&d_in, 3 * n * sizeof(double)));
  CK(hipMalloc(&d_out, 2 * n * sizeof(double)));
  CK(hipMemcpy(d_in, h.data(), 3 * n * sizeof(double), hipMemcpyHostToDevice));
  probe<<<(unsigned)((n + 255) / 256), 256>>>(d_in, n, d_out);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(o.data(), d_out, 2 * n * sizeof(double), hipMemcpyDeviceToHost));
  f = fopen(argv[2], "wb");
  if (!f || fwrite(o.data(), sizeof(double), 2 * n, f) != 2 * n) return 2;
  fclose(f);
  return 0;
}

// check of wave_sum4 (permlane swaps) against plain sums
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../../hts-train-world_amd/csrc/common.hpp"
__global__ void k(const double* in, double* out) {
  double a = in[threadIdx.x], b = in[64 + threadIdx.x], c = in[128 + threadIdx.x], d = in[192 + threadIdx.x];
  wm::wave_sum4(a, b, c, d);
  if (threadIdx.x == 17) { out[0] = a; out[1] = b; out[2] = c; out[3] = d; }
}
int main() {
  double h[256], ref[4] = {0, 0, 0, 0};
  for (int i = 0; i < 256; ++i) { h[i] = std::sin(0.37 * i) * (1 + i % 7); ref[i / 64] += h[i]; }
  double *din, *dout; (void)hipMalloc(&din, sizeof(h)); (void)hipMalloc(&dout, 32);
  (void)hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(din, dout);
  double o[4]; (void)hipMemcpy(o, dout, 32, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 4; ++i) { printf("%d: %.15g vs %.15g\n", i, o[i], ref[i]); if (std::fabs(o[i] - ref[i]) > 1e-12) bad = 1; }
  printf(bad ? "FAIL\n" : "OK\n");
  return bad;
}

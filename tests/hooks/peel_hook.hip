// peel_hook.hip -- TEST INFRASTRUCTURE: the product's per-lane sort and peel_largest() (csrc/peel.hpp) behind a C
// entry point of their own (tests/hooks/libpeel_hook.so), used exactly as d4c_kernel uses them.  Nothing here is
// linked into libworld_mi355.so.
#include <hip/hip_runtime.h>

#include "peel.hpp"

namespace wm {

// One wavefront per case.  vals[case][lane][M + 1]: M values per lane plus one extra slot (the kernel's bin fft/2:
// a value on lane 0, -1 elsewhere).  Outputs: low[case] = sum of all values but the K largest, taken[case][lane].
template <int M>
__global__ __launch_bounds__(64) void peel_test_kernel(int count, int K, const double* __restrict__ vals,
                                                       double* __restrict__ low_out, int* __restrict__ taken_out) {
  constexpr int MB = M + 1;
  __shared__ double heads[(MB + 3) * 64];
  const int lane = threadIdx.x;
  for (int c = blockIdx.x; c < count; c += gridDim.x) {
    double p[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) p[m] = vals[((int64_t)c * 64 + lane) * MB + m];
    sort_desc<M>(p);
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {                    // the extra slot into place (as in d4c_kernel)
      const double hi = fmax(p[i], p[i + 1]), lo = fmin(p[i], p[i + 1]);
      p[i] = hi;
      p[i + 1] = lo;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MB; ++m) heads[m * 64 + lane] = p[m];
#pragma unroll
    for (int m = MB; m < MB + 3; ++m) heads[m * 64 + lane] = -1.0;
    const int taken = peel_largest(heads, K, lane);
    double low = 0.0;
#pragma unroll
    for (int m = 0; m < MB; ++m) low += (m >= taken && p[m] >= 0.0) ? p[m] : 0.0;
    low = wave_sum(low);
    if (lane == 0) low_out[c] = low;
    taken_out[(int64_t)c * 64 + lane] = taken;
    __syncthreads();
  }
}

}  // namespace wm

// m = values per lane without the extra slot (16 or 32).  Returns 0, or -1 for another m.
extern "C" int PeelHook(void* stream, int m, int count, int K, const double* vals, double* low, int* taken) {
  using namespace wm;
  hipStream_t st = (hipStream_t)stream;
  const int grid = count < 1024 ? count : 1024;
  if (m == 16) hipLaunchKernelGGL(peel_test_kernel<16>, dim3(grid), dim3(64), 0, st, count, K, vals, low, taken);
  else if (m == 32) hipLaunchKernelGGL(peel_test_kernel<32>, dim3(grid), dim3(64), 0, st, count, K, vals, low, taken);
  else return -1;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

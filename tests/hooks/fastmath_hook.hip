// fastmath_hook.hip -- TEST INFRASTRUCTURE: the product's wm_log / wm_exp / wm_sincospi / wm_sqrt (csrc/fastmath.hpp) applied to arrays, behind a
// C entry point of their own (tests/hooks/libfastmath_hook.so).  Nothing here is linked into libworld_mi355.so.
#include <hip/hip_runtime.h>

#include "fastmath.hpp"

namespace wm {
__global__ __launch_bounds__(256) void fastmath_test_kernel(int64_t n, const double* __restrict__ x,
                                                            double* __restrict__ lg, double* __restrict__ ex,
                                                            double* __restrict__ sn, double* __restrict__ cs,
                                                            double* __restrict__ sq) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    lg[i] = wm_log(x[i]);
    ex[i] = wm_exp(x[i]);
    wm_sincospi(x[i], &sn[i], &cs[i]);
    sq[i] = wm_sqrt(x[i]);
  }
}
}  // namespace wm

extern "C" int FastmathHook(void* stream, int64_t n, const double* x, double* lg, double* ex, double* sn, double* cs,
                            double* sq) {
  hipLaunchKernelGGL(wm::fastmath_test_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, n, x, lg, ex, sn, cs, sq);
  return (int)hipGetLastError();
}

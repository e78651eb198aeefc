// fft_test.hip -- test hook exposing the wavefront FFT (fft.hpp) through the C ABI
// (WorldMi355TestRfft) so that tests can check it against numpy; not on the product path.
#include "batch.hpp"
#include "fft.hpp"

namespace wm {

template <int F>
__global__ __launch_bounds__(64) void rfft_test_kernel(int count, const double* __restrict__ x,
                                                       double* __restrict__ re, double* __restrict__ im,
                                                       double* __restrict__ xb) {
  constexpr int N = F / 2, M = N / 64;
  __shared__ __attribute__((aligned(16))) double smem[2 * FftLds<N>::kElems];
  cpx* img = reinterpret_cast<cpx*>(smem);
  const int lane = threadIdx.x;
  FftTw<N> tw;
  tw.init(lane);
  for (int row = blockIdx.x; row < count; row += gridDim.x) {
    const double* xr = x + (int64_t)row * F;
    cpx v[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int n = lane + 64 * m;
      v[m] = make_double2(xr[2 * n], xr[2 * n + 1]);
    }
    rfft_forward<N>(v, img, img, tw, lane);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int k = lane + 64 * m;
      re[(int64_t)row * (N + 1) + k] = img[k].x;
      im[(int64_t)row * (N + 1) + k] = img[k].y;
    }
    if (lane == 0) {
      re[(int64_t)row * (N + 1) + N] = img[N].x;
      im[(int64_t)row * (N + 1) + N] = img[N].y;
    }
    rfft_backward<N>(img, v, img, tw, lane);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int n = lane + 64 * m;
      xb[(int64_t)row * F + 2 * n] = v[m].x;
      xb[(int64_t)row * F + 2 * n + 1] = v[m].y;
    }
    __syncthreads();
  }
}

int launch_test_rfft(Context* ctx, int n, int count, const double* x, double* re, double* im, double* xb) {
  const int grid = count < 1024 ? count : 1024;
  switch (n) {
    case 1024: hipLaunchKernelGGL(rfft_test_kernel<1024>, dim3(grid), dim3(64), 0, ctx->stream, count, x, re, im, xb); break;
    case 2048: hipLaunchKernelGGL(rfft_test_kernel<2048>, dim3(grid), dim3(64), 0, ctx->stream, count, x, re, im, xb); break;
    case 4096: hipLaunchKernelGGL(rfft_test_kernel<4096>, dim3(grid), dim3(64), 0, ctx->stream, count, x, re, im, xb); break;
    default: return WM_ERR_UNSUPPORTED_FFT;
  }
  return wm_check(hipGetLastError());
}

}  // namespace wm

// fft_hook.hip -- TEST INFRASTRUCTURE: exposes the wavefront FFT engine of the product (csrc/fft.hpp) through a
// small shared library of its own (tests/hooks/libfft_hook.so), so that tests can check it against numpy.
// Nothing here is linked into libworld_mi355.so.
#include <hip/hip_runtime.h>

#include "fft.hpp"

namespace wm {

// nz < 0: the plain transform; nz >= 0: rfft_forward_nz with that many leading packed registers declared non-zero
// (the rows must then be zero from sample 128 nz on: the pruned first pass does not read what lies behind)
template <int F>
__global__ __launch_bounds__(64) void rfft_test_kernel(int count, int nz, const double* __restrict__ x,
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
    if (nz < 0) rfft_forward<N>(v, img, img, tw, lane);
    else rfft_forward_nz<N>(v, img, img, tw, lane, nz);
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

// The same through the PAIR forms (rfft_split_pairs / rfft_backward_pairs): the lane that holds Z[k] hands on X[k] and
// X[N - k]; here every bin is written to memory from the lane that holds it, and the inverse takes the pairs back.
template <int F>
__global__ __launch_bounds__(64) void rfft_pairs_test_kernel(int count, int nz, const double* __restrict__ x,
                                                             double* __restrict__ re, double* __restrict__ im,
                                                             double* __restrict__ xb) {
  constexpr int N = F / 2, M = N / 64, MH = M / 2;
  __shared__ __attribute__((aligned(16))) double smem[2 * FftLds<N>::kElems];
  cpx* img = reinterpret_cast<cpx*>(smem);
  const int lane = threadIdx.x;
  FftTw<N> tw;
  tw.init(lane);
  for (int row = blockIdx.x; row < count; row += gridDim.x) {
    const double* xr_ = x + (int64_t)row * F;
    cpx v[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int n = lane + 64 * m;
      v[m] = make_double2(xr_[2 * n], xr_[2 * n + 1]);
    }
    cpx xk[MH], xr[MH], xh;
    if (nz < 0) rfft_forward_pairs<N>(v, img, tw, lane, xk, xr, xh);
    else rfft_forward_nz_pairs<N>(v, img, tw, lane, nz, xk, xr, xh);
    double* rr = re + (int64_t)row * (N + 1);
    double* ri = im + (int64_t)row * (N + 1);
#pragma unroll
    for (int m = 0; m < MH; ++m) {
      const int k = lane + 64 * m;
      rr[k] = xk[m].x;
      ri[k] = xk[m].y;
      rr[N - k] = xr[m].x;
      ri[N - k] = xr[m].y;
    }
    if (lane == 0) {
      rr[N / 2] = xh.x;
      ri[N / 2] = xh.y;
    }
    rfft_backward_pairs<N>(xk, xr, xh, v, img, tw, lane);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int n = lane + 64 * m;
      xb[(int64_t)row * F + 2 * n] = v[m].x;
      xb[(int64_t)row * F + 2 * n + 1] = v[m].y;
    }
    __syncthreads();
  }
}

}  // namespace wm

// The pair forms; n = 512, 1024, 2048, 4096.
extern "C" int FftHookRfftPairs(void* stream, int n, int count, int nz, const double* x, double* re, double* im,
                                double* xb) {
  using namespace wm;
  hipStream_t st = (hipStream_t)stream;
  const int grid = count < 1024 ? count : 1024;
  switch (n) {
    case 512: hipLaunchKernelGGL(rfft_pairs_test_kernel<512>, dim3(grid), dim3(64), 0, st, count, nz, x, re, im, xb); break;
    case 1024: hipLaunchKernelGGL(rfft_pairs_test_kernel<1024>, dim3(grid), dim3(64), 0, st, count, nz, x, re, im, xb); break;
    case 2048: hipLaunchKernelGGL(rfft_pairs_test_kernel<2048>, dim3(grid), dim3(64), 0, st, count, nz, x, re, im, xb); break;
    case 4096: hipLaunchKernelGGL(rfft_pairs_test_kernel<4096>, dim3(grid), dim3(64), 0, st, count, nz, x, re, im, xb); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? (hipStreamSynchronize(st) == hipSuccess ? 0 : -2) : -2;
}

// Forward / backward real FFT of `count` rows of length n (1024, 2048, 4096) on `stream`; layouts as
// externs/WORLD_v2/src/fft.cpp:26-72 (re / im split, n / 2 + 1 bins).  Returns 0, or -1 for another n.
extern "C" int FftHookRfftNz(void* stream, int n, int count, int nz, const double* x, double* re, double* im,
                             double* xb);
extern "C" int FftHookRfft(void* stream, int n, int count, const double* x, double* re, double* im, double* xb) {
  return FftHookRfftNz(stream, n, count, -1, x, re, im, xb);
}
// The same with the pruned first pass: nz leading packed registers (128 samples each) may be non-zero.
extern "C" int FftHookRfftNz(void* stream, int n, int count, int nz, const double* x, double* re, double* im,
                             double* xb) {
  using namespace wm;
  hipStream_t st = (hipStream_t)stream;
  const int grid = count < 1024 ? count : 1024;
  switch (n) {
    case 1024: hipLaunchKernelGGL(rfft_test_kernel<1024>, dim3(grid), dim3(64), 0, st, count, nz, x, re, im, xb); break;
    case 2048: hipLaunchKernelGGL(rfft_test_kernel<2048>, dim3(grid), dim3(64), 0, st, count, nz, x, re, im, xb); break;
    case 4096: hipLaunchKernelGGL(rfft_test_kernel<4096>, dim3(grid), dim3(64), 0, st, count, nz, x, re, im, xb); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? (hipStreamSynchronize(st) == hipSuccess ? 0 : -2) : -2;
}

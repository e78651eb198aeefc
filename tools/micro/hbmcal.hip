// hbmcal.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this repo's
// kernels use.  MI355X_MICROARCH.md (HBM section) says FETCH_SIZE reports half the bytes of a 16-byte-per-lane
// coalesced streaming read and that other widths are uncalibrated; the per-frame kernels here read and write
// 8 bytes per lane (one double), so the factor is measured instead of assumed:
//
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d DIR -o f --output-format csv -- tools/micro/hbmcal
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d DIR -o w --output-format csv -- tools/micro/hbmcal
//   python tools/hbmcal_to_json.py FETCH_DIR WRITE_DIR        -> profiles/hbm_counter_calibration.json
//
// Every kernel streams a buffer far larger than the 256 MiB Infinity Cache exactly once; the binary prints the
// bytes each kernel really moves ("cal <kernel> <bytes read> <bytes written>"), the script divides.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(e)                                                                              \
  do {                                                                                        \
    hipError_t err_ = (e);                                                                    \
    if (err_ != hipSuccess) {                                                                 \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(err_));            \
      exit(1);                                                                                \
    }                                                                                         \
  } while (0)

// one wavefront per workgroup, grid-stride over "rows" of 64 * PER elements, like the per-frame kernels
template <typename T, int PER>
__global__ __launch_bounds__(64) void cal_read(const T* __restrict__ src, int64_t rows, int64_t skew, double* __restrict__ sink) {
  double acc = 0.0;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const T* p = src + skew + r * (64 * PER) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const T v = p[64 * k];
      acc += *reinterpret_cast<const double*>(&v);
    }
  }
  if (acc == 1.2345e300) sink[blockIdx.x] = acc;       // never true: keeps the loads alive
}

template <typename T, int PER>
__global__ __launch_bounds__(64) void cal_write(T* __restrict__ dst, int64_t rows, int64_t skew, double seed) {
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    T* p = dst + skew + r * (64 * PER) + threadIdx.x;
    T v;
    double* d = reinterpret_cast<double*>(&v);
    for (unsigned j = 0; j < sizeof(T) / 8; ++j) d[j] = seed + (double)r;
#pragma unroll
    for (int k = 0; k < PER; ++k) p[64 * k] = v;
  }
}

// a frame-shaped access: each wave reads a window of `len` doubles starting at an arbitrary (8-byte aligned) sample and
// writes one row of 513 doubles -- d4c_kernel's pattern; windows of neighbouring frames overlap by len - hop samples
__global__ __launch_bounds__(64) void cal_frames(const double* __restrict__ x, int64_t frames, int hop, int len,
                                                  double* __restrict__ rows) {
  for (int64_t f = blockIdx.x; f < frames; f += gridDim.x) {
    const double* p = x + f * hop;
    double acc = 0.0;
    for (int i = threadIdx.x; i < len; i += 64) acc += p[i];
    double* o = rows + f * 513;
    for (int i = threadIdx.x; i < 513; i += 64) o[i] = acc + i;
  }
}

struct double2a { double a, b; } __attribute__((aligned(16)));

int main() {
  const int64_t bytes = (int64_t)2 << 30;                     // 2 GiB: eight times the Infinity Cache
  void* buf = nullptr;
  double* sink = nullptr;
  CHECK(hipMalloc(&buf, bytes + 4096));
  CHECK(hipMalloc(&sink, 1 << 20));
  CHECK(hipMemset(buf, 0x3c, bytes + 4096));
  CHECK(hipDeviceSynchronize());
  const int grid = 256 * 16;
  const int64_t n8 = bytes / 8, n16 = bytes / 16;
  // each kernel twice (the first launch of a process carries code-object loading in some counters)
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((cal_read<double, 8>), dim3(grid), dim3(64), 0, 0, (const double*)buf, n8 / 512, 0, sink);
    hipLaunchKernelGGL((cal_read<double, 8>), dim3(grid), dim3(64), 0, 0, (const double*)buf, n8 / 512, 1, sink);   // +8 B skew
    hipLaunchKernelGGL((cal_read<double2a, 4>), dim3(grid), dim3(64), 0, 0, (const double2a*)buf, n16 / 256, 0, sink);
    hipLaunchKernelGGL((cal_write<double, 8>), dim3(grid), dim3(64), 0, 0, (double*)buf, n8 / 512, 0, 1.0);
    hipLaunchKernelGGL((cal_write<double, 8>), dim3(grid), dim3(64), 0, 0, (double*)buf, n8 / 512, 1, 2.0);
    hipLaunchKernelGGL((cal_write<double2a, 4>), dim3(grid), dim3(64), 0, 0, (double2a*)buf, n16 / 256, 0, 3.0);
    // frames: 16 kHz / 5 ms hop = 80 samples, windows of 681 samples (4 fs / 94 Hz); rows behind the samples
    const int64_t frames = 300000;
    double* rows = (double*)buf + ((int64_t)1 << 26);         // 512 MiB in: the samples occupy 256 MB below
    hipLaunchKernelGGL(cal_frames, dim3(grid), dim3(64), 0, 0, (const double*)buf, frames, 80, 681, rows);
    CHECK(hipDeviceSynchronize());
    if (rep == 0) {
      printf("cal cal_read<double,8> %lld 0\n", (long long)(n8 / 512 * 512 * 8));
      printf("cal cal_read<double,8>+8B %lld 0\n", (long long)(n8 / 512 * 512 * 8));
      printf("cal cal_read<double2,4> %lld 0\n", (long long)(n16 / 256 * 256 * 16));
      printf("cal cal_write<double,8> 0 %lld\n", (long long)(n8 / 512 * 512 * 8));
      printf("cal cal_write<double,8>+8B 0 %lld\n", (long long)(n8 / 512 * 512 * 8));
      printf("cal cal_write<double2,4> 0 %lld\n", (long long)(n16 / 256 * 256 * 16));
      printf("cal cal_frames %lld %lld\n", (long long)((frames * 80 + 681) * 8), (long long)(frames * 513 * 8));
    }
  }
  CHECK(hipFree(buf));
  CHECK(hipFree(sink));
  return 0;
}

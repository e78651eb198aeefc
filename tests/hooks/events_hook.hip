// events_hook.hip -- TEST INFRASTRUCTURE: conv_block_events() (csrc/fftconv.hpp), the zero-crossing passes of a
// filtered block, behind a C entry point of its own (tests/hooks/libevents_hook.so).  Nothing here is linked into
// libworld_mi355.so.
#include <hip/hip_runtime.h>

#include "fftconv.hpp"

namespace wm {

constexpr int kC = 28;

// One wavefront per block: s[block][step + 2] filtered samples of outputs n0 = block * step ...; list_cap as the
// product uses it, or smaller to force the direct path.
__global__ __launch_bounds__(64) void events_test_kernel(int step, int ylen, int list_cap, const double* __restrict__ sin,
                                                         int* __restrict__ cnt4, double* __restrict__ slots,
                                                         int64_t slot_cap) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x, blk = blockIdx.x;
  for (int i = lane; i < step + 2; i += 64) lds[i] = sin[(int64_t)blk * (step + 2) + i];
  __syncthreads();
  conv_block_events<kC>(lds, blk * step, step, ylen, blk, ConvEvCfg<2048, kC>::lists(lds), list_cap, cnt4 + 4 * blk, slots,
                        slot_cap, lane);
}

}  // namespace wm

extern "C" int EventsHook(void* stream, int blocks, int step, int ylen, int list_cap, const double* s, int* cnt4,
                          double* slots) {
  using namespace wm;
  if (step > 64 * kC || list_cap > ConvEvCfg<2048, kC>::kListCap) return -1;
  hipLaunchKernelGGL(events_test_kernel, dim3(blocks), dim3(64), (ConvEvCfg<2048, kC>::kLdsBytes), (hipStream_t)stream,
                     step, ylen, list_cap, s, cnt4, slots, (int64_t)blocks * kZcSlot);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int EventsHookSlot(void) { return wm::kZcSlot; }
extern "C" int EventsHookListCap(void) { return wm::ConvEvCfg<2048, wm::kC>::kListCap; }

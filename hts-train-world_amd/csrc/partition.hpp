// partition.hpp -- stable two-way partition of an index range, for load balance.
//
// The per-frame kernels run a persistent grid that deals frames round-robin.  Where a frame either
// costs a full analysis or nothing (D4C and LoveTrain skip f0 == 0, d4c.cpp:231-233, :380; StoneMask
// skips f0 <= 0, stonemask.cpp:196-199; an unvoiced pulse has no periodic response,
// synthesis.cpp:197-204), the number of costly frames per wave varies by +-10 % around its mean
// and the launch lasts as long as its unluckiest wave.  Listing the costly indices first (ascending,
// so neighbours still share cache lines of the waveform) and dealing the list gives every wave the
// same count +-1.
//
//   perm[0 .. n_true)  indices with pred true, ascending
//   perm[n_true .. n)  the others, ascending
//   *n_true            written by the scatter kernel
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hpp"

namespace wm {

constexpr int kPartBlock = 1024;   // indices per workgroup (256 threads x 4 consecutive)

template <class Pred>
__global__ __launch_bounds__(256) void partition_count_kernel(Pred pred, int n, int* __restrict__ block_cnt) {
  __shared__ int wsum[4];
  const int i0 = blockIdx.x * kPartBlock + threadIdx.x * 4;
  int c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) c += (i0 + k < n && pred(i0 + k)) ? 1 : 0;
  c = wave_sum_i(c);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

template <class Pred>
__global__ __launch_bounds__(256) void partition_scatter_kernel(Pred pred, int n, const int* __restrict__ block_cnt,
                                                                int* __restrict__ perm, int* __restrict__ n_true) {
  __shared__ int wsum[4], wtot[4], wbase[4];
  const int nb = gridDim.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // trues before this block, and in total
  int before = 0, total = 0;
  for (int j = threadIdx.x; j < nb; j += 256) {
    const int v = block_cnt[j];
    total += v;
    if (j < (int)blockIdx.x) before += v;
  }
  before = wave_sum_i(before);
  total = wave_sum_i(total);
  if (lane == 0) { wsum[wv] = before; wtot[wv] = total; }
  __syncthreads();
  before = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_true = total;

  const int i0 = blockIdx.x * kPartBlock + threadIdx.x * 4;
  bool f[4];
  int c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[k] = i0 + k < n && pred(i0 + k);
    c += f[k] ? 1 : 0;
  }
  const int incl = wave_scan_incl_i(c);
  if (lane == 63) wbase[wv] = incl;
  __syncthreads();
  int rank = incl - c;                                   // trues before this thread's four, in the block
  for (int w = 0; w < wv; ++w) rank += wbase[w];
  int t_at = before + rank;                              // next true slot
  int f_at = total + (i0 - before - rank);               // next false slot: falses before i0 = i0 - trues before i0
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i0 + k >= n) break;
    if (f[k]) perm[t_at++] = i0 + k; else perm[f_at++] = i0 + k;
  }
}

// Scratch owned by the caller: perm[n], block_cnt[ceil(n / kPartBlock)], n_true[1].
template <class Pred>
inline void launch_partition(hipStream_t st, Pred pred, int n, int* block_cnt, int* perm, int* n_true) {
  if (n <= 0) {
    (void)hipMemsetAsync(n_true, 0, sizeof(int), st);
    return;
  }
  const int nb = (n + kPartBlock - 1) / kPartBlock;
  hipLaunchKernelGGL(partition_count_kernel<Pred>, dim3(nb), dim3(256), 0, st, pred, n, block_cnt);
  hipLaunchKernelGGL(partition_scatter_kernel<Pred>, dim3(nb), dim3(256), 0, st, pred, n, block_cnt, perm, n_true);
}

// readfirstlane: the entry is the same for every lane (the position depends on the workgroup only); saying so
// keeps the frame index, and every per-frame quantity loaded through it, in scalar registers
__device__ __forceinline__ int64_t listed_at(const int* __restrict__ perm, int64_t pos, int64_t count) {
  return pos < count ? (int64_t)__builtin_amdgcn_readfirstlane(perm[pos]) : -1;
}
// Persistent-grid loop over perm[0 .. count): `frame` is the listed index; XCD-aware like WM_FOR_EACH_FRAME.
// The entry of the next round is requested at the top of the body, so its latency is not in front of the
// next frame's dependent loads.
#define WM_FOR_EACH_LISTED(frame, perm, count)                                                            \
  for (int64_t base_ = 0, nxt_ = listed_at((perm), xcd_dealt(0, (count)), (count)), frame = nxt_;         \
       base_ < (int64_t)(count); base_ += gridDim.x, frame = nxt_)                                        \
    if ((nxt_ = listed_at((perm), xcd_dealt(base_ + gridDim.x, (count)), (count))), frame >= 0)

}  // namespace wm

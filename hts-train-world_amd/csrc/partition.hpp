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

// Per-frame scalars of the analysis kernels, fetched AHEAD through a three-stage pipeline.  A frame starts with a
// chain of dependent loads -- list entry -> frame -> (f0, position, utterance, randn offset) -> utterance base and
// length -> samples -- and at one or two resident waves per SIMD nothing hides those round trips (LoveTrain spent a
// third of its time in them).  Requesting the whole chain one frame ahead does not help: the dependent scalar
// loads need a wait between them, and the wave would sit through the chain at the top of every body.  So every
// level of the chain belongs to a different future frame and each iteration advances all levels by one load each:
//     frame k+3: its list entry is requested
//     frame k+2: its entry has arrived (requested one iteration ago) -> f0, position, utterance, offset requested
//     frame k+1: its utterance has arrived -> the utterance's base and length requested
//     frame k  : everything is in registers
// Every value is consumed one iteration after its request; the sample loads of a frame are the only round trip
// left on its critical path.
struct FrameScalars {
  int64_t pos;          // position in the list
  int64_t frame;        // -1 beyond the end of the list
  double f0, tpos;
  const double* xu;     // the utterance's samples
  int xlen, roff;
};
struct FramePipe {
  const int* perm;
  int64_t count;
  const double *x, *tpos, *f0;
  const int64_t* x_off;
  const int *x_len, *frame_utt, *rng_off;
  int64_t base;                       // list position (before dealing) of stage 0
  // stage 3: list entry; stage 2: + per-frame scalars; stage 1: + utterance base / length
  int raw3;                           // list entry as loaded (valid only if ok3): not looked at until the next step,
  bool ok3;                           // so that its load is not waited for in the iteration that issued it
  int64_t fr2, fr1;
  double f0_2, tp_2, f0_1, tp_1;
  int utt_2, ro_2, ro_1, xl_1;
  int64_t xo_1;
  __device__ __forceinline__ void request_entry(int64_t b) {
    const int64_t pos = xcd_dealt(b, count);
    ok3 = pos < count;
    raw3 = perm[ok3 ? pos : 0];         // unconditional load (clamped), no select on the value here
  }
  __device__ __forceinline__ int64_t entry3() const {
    return ok3 ? (int64_t)__builtin_amdgcn_readfirstlane(raw3) : -1;
  }
  __device__ __forceinline__ void scalars(int64_t fr, double& a, double& t, int& u, int& r) const {
    const int64_t i = fr < 0 ? 0 : fr;          // beyond the list: a harmless reload of frame 0
    a = f0[i]; t = tpos[i]; u = frame_utt[i]; r = rng_off[i];
  }
  __device__ __forceinline__ void init(const int* perm_, int64_t count_, const double* x_, const int64_t* x_off_,
                                       const int* x_len_, const int* frame_utt_, const double* tpos_,
                                       const double* f0_, const int* rng_off_) {
    perm = perm_; count = count_; x = x_; x_off = x_off_; x_len = x_len_; frame_utt = frame_utt_;
    tpos = tpos_; f0 = f0_; rng_off = rng_off_;
    base = 0;
    // fill: frames of rounds 0, 1, 2 (blocking, once per kernel)
    request_entry(0);
    fr1 = entry3();
    request_entry((int64_t)gridDim.x);
    fr2 = entry3();
    request_entry(2 * (int64_t)gridDim.x);
    int u1;
    scalars(fr1, f0_1, tp_1, u1, ro_1);
    xo_1 = x_off[u1];
    xl_1 = x_len[u1];
    scalars(fr2, f0_2, tp_2, utt_2, ro_2);
  }
  // the scalars of the frame of round `base`, and one step of every stage
  __device__ __forceinline__ FrameScalars next() {
    FrameScalars c;
    c.pos = xcd_dealt(base, count);
    c.frame = fr1; c.f0 = f0_1; c.tpos = tp_1; c.xu = x + xo_1; c.xlen = xl_1; c.roff = ro_1;
    const int64_t e3 = entry3();        // arrived: requested one step ago
    // new stage 3 first: its destination is then not behind this step's other requests
    request_entry(base + 3 * (int64_t)gridDim.x);
    base += gridDim.x;
    // stage 2 -> 1: the utterance of the frame after this one is known by now
    fr1 = fr2; f0_1 = f0_2; tp_1 = tp_2; ro_1 = ro_2;
    xo_1 = x_off[utt_2];
    xl_1 = x_len[utt_2];
    // stage 3 -> 2
    fr2 = e3;
    scalars(fr2, f0_2, tp_2, utt_2, ro_2);
    return c;
  }
};
// for-loop header: `sc` is the FrameScalars of the current frame; frames beyond the list are skipped
#define WM_FOR_EACH_PIPED(sc, pipe, count)                                                                   \
  for (int64_t base_ = 0; base_ < (int64_t)(count); base_ += gridDim.x)                                      \
    for (FrameScalars sc = (pipe).next(), *once_ = &sc; once_; once_ = nullptr)                              \
      if (sc.frame >= 0)

}  // namespace wm

// status.hip -- per-utterance status flags of a batch (a failed utterance must not poison
// the batch, and the caller must be able to tell which one failed).
//
// The reference has no error reporting at all (every entry point returns void).  Here the utterances of a batch
// never exchange data -- every kernel works per utterance, per frame or per pulse -- so a bad utterance cannot
// change another's results; what a batch caller still needs is to learn WHICH utterances are unusable.  One
// workgroup per utterance scans its input and outputs:
//   WM_UTT_INPUT_NONFINITE   a NaN / Inf sample in x
//   WM_UTT_TOO_SHORT         f0_length <= voice_range_minimum: Dio has no contour to fix (the reference returns
//                            with f0 unwritten, dio.cpp:266; here f0 is all zero)
//   WM_UTT_OUTPUT_NONFINITE  a NaN / Inf in the utterance's f0 / sp / ap rows (arrays given as NULL are skipped)
//   WM_UTT_D4C_DEFAULT_ROWS  kept for ABI compatibility, never set since round 5 (it marked frames with f0 >= fs / 16 at
//                            fs above 48.1 kHz, which kept D4C's default row until d4c_wide.hpp analysed them)
#include "batch.hpp"
#include "common.hpp"

namespace wm {

__global__ __launch_bounds__(256) void utterance_status_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int64_t* __restrict__ f_off, const double* __restrict__ f0, const double* __restrict__ sp,
    const double* __restrict__ ap, int bins, int vrm, double d4c_bins_per_hz, int d4c_bin_limit,
    int* __restrict__ status) {
  __shared__ int flags;
  const int u = blockIdx.x;
  if (threadIdx.x == 0) flags = 0;
  __syncthreads();
  int mine = 0;
  if (x) {
    const double* xu = x + x_off[u];
    const int n = x_len[u];
    bool bad = false;
    for (int i = threadIdx.x; i < n; i += 256) bad |= !isfinite(xu[i]);
    if (bad) mine |= WM_UTT_INPUT_NONFINITE;
  }
  const int64_t fb = f_off[u];
  const int nf = (int)(f_off[u + 1] - fb);
  if (nf <= vrm) mine |= WM_UTT_TOO_SHORT;
  bool bad = false, rare = false;
  if (f0)
    for (int i = threadIdx.x; i < nf; i += 256) {
      const double v = f0[fb + i];
      bad |= !isfinite(v);
      // d4c.hip's d4c_mirror_bins(f0) > fft_size_d4c / 16 and within what the reference defines (<= fft_size_d4c / 2)
      if (d4c_bin_limit > 0 && v > 0.0 && isfinite(v)) {
        const double q = v * d4c_bins_per_hz;
        rare |= q < 1.0e6 && (int)q + 1 > d4c_bin_limit && (int)q + 1 <= 8 * d4c_bin_limit;
      }
    }
  if (rare) mine |= WM_UTT_D4C_DEFAULT_ROWS;
  const int64_t cells = (int64_t)nf * bins;
  if (sp)
    for (int64_t i = threadIdx.x; i < cells; i += 256) bad |= !isfinite(sp[fb * bins + i]);
  if (ap)
    for (int64_t i = threadIdx.x; i < cells; i += 256) bad |= !isfinite(ap[fb * bins + i]);
  if (bad) mine |= WM_UTT_OUTPUT_NONFINITE;
  if (mine) atomicOr(&flags, mine);
  __syncthreads();
  if (threadIdx.x == 0) status[u] = flags;
}

int launch_utterance_status(Batch& b, const double* d_x, const double* d_f0, const double* d_sp, const double* d_ap,
                            int* d_status) {
  if (!d_status) return WM_ERR_BAD_ARG;
  const int vrm = (int)(0.5 + 1000.0 / b.p.frame_period / b.p.f0_floor) * 2 + 1;   // dio.cpp:263-264
  // fft_size_d4c (d4c.cpp:344-346); the rare-frame kernel exists up to 4096 points
  const int fd = (int)pow(2.0, 1.0 + (int)(log(4.0 * b.p.fs / 47.0 + 1) / log(2.0)));
  // (until round 5 frames with f0 >= fs / 16 kept D4C's default row where its transform has 8192 points, and were
  // flagged here; d4c_wide_kernel analyses them now, so WM_UTT_D4C_DEFAULT_ROWS is never set any more)
  const int bin_limit = 0;
  (void)fd;
  hipLaunchKernelGGL(utterance_status_kernel, dim3(b.n_utt), dim3(256), 0, b.ctx->stream, b.total_x > 0 ? d_x : nullptr,
                     b.d_x_off, b.d_x_len, b.d_f_off, d_f0, d_sp, d_ap, b.p.fft_size / 2 + 1, vrm, (double)fd / b.p.fs,
                     bin_limit, d_status);
  return wm_check(hipGetLastError());
}

}  // namespace wm

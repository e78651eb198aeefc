// bcommon.hpp -- workgroup-cooperative versions of the per-frame helpers (NT = 64 * NW threads per
// frame): F0-adaptive windowing (cheaptrick.cpp:87-142, d4c.cpp:21-84), DCCorrection
// (common.cpp:56-75) and LinearSmoothing (common.cpp:27-46, 77-111).  Same arithmetic as
// window.hpp / common.hpp; only the work split and the reduction trees differ.
#pragma once
#include "bfft.hpp"
#include "common.hpp"
#include "window.hpp"

namespace wm {

// wav[0..F) (LDS) <- windowed, dithered, mean-removed frame.  Ends with a barrier.
template <int TYPE, bool NORMALISE, int NW>
__device__ __forceinline__ FrameWindow windowed_waveform_blk(const double* __restrict__ xu, int xl, int fs,
                                                             double f0, double pos, double ratio,
                                                             const uint32_t* __restrict__ rtab, int roff, int tid,
                                                             double* wav, int F, double* red) {
  constexpr int NT = 64 * NW;
  FrameWindow fw;
  fw.hw = matlab_round(ratio * fs / f0 / 2.0);
  fw.L = 2 * fw.hw + 1;
  fw.origin = matlab_round(pos * fs + 0.001);
  fw.a = 2.0 * f0 / (ratio * fs);
  fw.scale = 1.0;
  CosGen g;
  if (NORMALISE) {
    g.init(fw.a, tid - fw.hw, NT);
    double e = 0.0;
    for (int i = tid; i < fw.L; i += NT) {
      const double w = window_value<TYPE>(g.c);
      e += w * w;
      g.next();
    }
    fw.scale = sqrt(BlockOps<NW>::sum(e, red, tid));
  }
  const double inv_scale = 1.0 / fw.scale;
  g.init(fw.a, tid - fw.hw, NT);
  double s1 = 0.0, s2 = 0.0;
  for (int i = tid; i < F; i += NT) {
    double val = 0.0;
    if (i < fw.L) {
      const double w = NORMALISE ? window_value<TYPE>(g.c) * inv_scale : window_value<TYPE>(g.c);
      val = xu[imin(xl - 1, imax(0, fw.origin + i - fw.hw))] * w + randn_at(rtab, roff + i) * kSafe;
      s1 += val;
      s2 += w;
    }
    wav[i] = val;
    g.next();
  }
  const double t1 = BlockOps<NW>::sum(s1, red, tid);
  const double t2 = BlockOps<NW>::sum(s2, red, tid);
  fw.coef = t1 / t2;
  g.init(fw.a, tid - fw.hw, NT);
  for (int i = tid; i < fw.L; i += NT) {       // own samples only
    wav[i] -= (NORMALISE ? window_value<TYPE>(g.c) * inv_scale : window_value<TYPE>(g.c)) * fw.coef;
    g.next();
  }
  __syncthreads();
  return fw;
}

// second transform of D4C's centroid: the same frame / nrm * (i + 1)  (d4c.cpp:96-112)
template <int TYPE, int NW>
__device__ __forceinline__ void rebuild_ramped_blk(const double* __restrict__ xu, int xl, const FrameWindow& fw,
                                                   const uint32_t* __restrict__ rtab, int roff, double rnrm, int tid,
                                                   double* wav, int F) {
  constexpr int NT = 64 * NW;
  CosGen g;
  g.init(fw.a, tid - fw.hw, NT);
  for (int i = tid; i < F; i += NT) {
    double val = 0.0;
    if (i < fw.L) {
      const double w = window_value<TYPE>(g.c);
      val = xu[imin(xl - 1, imax(0, fw.origin + i - fw.hw))] * w + randn_at(rtab, roff + i) * kSafe;
      val -= w * fw.coef;
      val = val * rnrm * (i + 1.0);
    }
    wav[i] = val;
    g.next();
  }
  __syncthreads();
}

// DCCorrection in place on pw[0..half] (LDS); scratch >= upper doubles.  Ends with a barrier.
template <int NT>
__device__ __forceinline__ void dc_correction_blk(double* pw, double f0, int fs, int fft_size, double* scratch,
                                                  int tid) {
  const int upper = 2 + (int)(f0 * fft_size / fs);
  const int nrep = upper - 1;
  for (int i = tid; i < nrep; i += NT) {
    const double axis = (double)i * fs / fft_size;
    scratch[i] = interp1q_lds(f0, -(double)fs / fft_size, pw, upper + 1, axis);
  }
  __syncthreads();
  for (int i = tid; i < nrep; i += NT) pw[i] += scratch[i];
  __syncthreads();
}

// LinearSmoothing: in[0..half] -> out[0..half] (LDS, may alias); seg >= half + 2b + 1 doubles.
template <int NW>
__device__ __forceinline__ void linear_smoothing_blk(const double* in, double width, int fs, int fft_size,
                                                     double* seg, double* out, double* red, int tid) {
  constexpr int NT = 64 * NW;
  const int half = fft_size / 2;
  const double inv_fft = 1.0 / fft_size;               // power of two: x * inv_fft == x / fft_size exactly
  const int b = (int)(width * fft_size / fs) + 1;
  const int len = half + 2 * b + 1;
  const int chunk = (len + NT - 1) / NT;
  const int beg = tid * chunk;
  const int end = imin(len, beg + chunk);
  double run = 0.0;
  for (int i = beg; i < end; ++i) {
    const int src = i < b ? b - i : (i < half + b ? i - b : half - (i - (half + b)));
    run += in[src] * fs * inv_fft;
    seg[i] = run;
  }
  const int lane = tid & 63, wv = tid >> 6;
  const double incl = wave_scan_incl(run, lane);
  __syncthreads();
  if (lane == 63) red[wv] = incl;
  __syncthreads();
  double carry = incl - run;
#pragma unroll
  for (int w = 0; w < NW; ++w)
    if (w < wv) carry += red[w];
  for (int i = beg; i < end; ++i) seg[i] += carry;
  __syncthreads();
  const double origin = -(b - 0.5) * fs / fft_size;
  const double step = (double)fs / fft_size;
  for (int i = tid; i <= half; i += NT) {
    const double lo_x = (double)i * inv_fft * fs - width / 2.0;
    const double lo = interp1q_lds(origin, step, seg, len, lo_x);
    const double hi = interp1q_lds(origin, step, seg, len, lo_x + width);
    out[i] = (hi - lo) / width;
  }
  __syncthreads();
}

}  // namespace wm

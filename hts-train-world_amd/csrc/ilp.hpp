// ilp.hpp -- per-frame helpers for the one-wavefront kernels that run at ONE wave per SIMD.
//
// With a single resident wave per SIMD nothing hides latency except the wave's own instruction
// level parallelism: a rolled loop whose body is "global load -> multiply -> LDS store" or
// "divide -> convert -> LDS load -> interpolate" costs a full memory / divider latency per trip
// (measured with s_memtime stamps: 70 % of d4c_kernel's cycles sat in such loops, only 17 % in
// its six FFTs).  Everything here is therefore fully unrolled with compile-time trip counts:
// all loads of a frame are issued before the first use, all bins of a spectrum are interpolated
// side by side.  The price is registers (300-400 VGPRs), which is exactly what a one-wave-per-SIMD
// kernel has to spare (512).
//
// Arithmetic as window.hpp / common.hpp (same reference lines).
#pragma once
#include "common.hpp"
#include "fft.hpp"
#include "window.hpp"

namespace wm {

// cos(pi a (i - hw)) for this lane's 2M samples, in the FFT's packed order:
// cw[m] = (cos at i = 2n, cos at i = 2n+1), n = lane + 64 m.
template <int M>
__device__ __forceinline__ void frame_cos(double a, int hw, int lane, cpx (&cw)[M]) {
  CosGen g0, g1;
  g0.init(a, 2 * lane - hw, 128);
  g1.init(a, 2 * lane + 1 - hw, 128);
#pragma unroll
  for (int m = 0; m < M; ++m) {
    cw[m] = make_double2(g0.c, g1.c);
    g0.next();
    g1.next();
  }
}

// GetWindowedWaveform (d4c.cpp:52-84, unnormalised windows) into registers, packed order.
// All x / randn loads of the frame are issued up front (one memory round trip per frame instead of
// one per 64 samples); slots m >= MU are beyond the window and stay zero.
template <int TYPE, int M, int MU>
__device__ __forceinline__ void build_frame_regs_t(const double* __restrict__ xu, int xl, int origin, int hw, int L,
                                                   const uint32_t* __restrict__ rtab, int roff, int lane,
                                                   const cpx (&cw)[M], cpx (&fv)[M]) {
  double xs[2 * MU];
  uint32_t rs[2 * MU];
#pragma unroll
  for (int m = 0; m < MU; ++m) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int i = 2 * (lane + 64 * m) + c;
      const int ic = i < L ? i : L - 1;                            // keep the address inside the frame's draws
      xs[2 * m + c] = xu[imin(xl - 1, imax(0, origin + ic - hw))];
      rs[2 * m + c] = rtab[roff + ic];
    }
  }
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    if (m < MU) {
      const int i0 = 2 * (lane + 64 * m);
      const double w0 = i0 < L ? window_value<TYPE>(cw[m].x) : 0.0;
      const double w1 = i0 + 1 < L ? window_value<TYPE>(cw[m].y) : 0.0;
      const double a0 = i0 < L ? xs[2 * m] * w0 + ((double)rs[2 * m] / 268435456.0 - 6.0) * kSafe : 0.0;
      const double a1 = i0 + 1 < L ? xs[2 * m + 1] * w1 + ((double)rs[2 * m + 1] / 268435456.0 - 6.0) * kSafe : 0.0;
      fv[m] = make_double2(a0, a1);
      s1 += a0 + a1;
      s2 += w0 + w1;
    } else {
      fv[m] = make_double2(0.0, 0.0);
    }
  }
  const double coef = wave_sum(s1) / wave_sum(s2);
#pragma unroll
  for (int m = 0; m < MU; ++m) {
    const int i0 = 2 * (lane + 64 * m);
    if (i0 < L) fv[m].x -= window_value<TYPE>(cw[m].x) * coef;
    if (i0 + 1 < L) fv[m].y -= window_value<TYPE>(cw[m].y) * coef;
  }
}

// tier dispatch on the window length: a quarter / half / all of the M register slots
template <int TYPE, int M>
__device__ __forceinline__ void build_frame_regs(const double* __restrict__ xu, int xl, int origin, int hw, int L,
                                                 const uint32_t* __restrict__ rtab, int roff, int lane,
                                                 const cpx (&cw)[M], cpx (&fv)[M]) {
  if (L <= 32 * M) build_frame_regs_t<TYPE, M, M / 4>(xu, xl, origin, hw, L, rtab, roff, lane, cw, fv);
  else if (L <= 64 * M) build_frame_regs_t<TYPE, M, M / 2>(xu, xl, origin, hw, L, rtab, roff, lane, cw, fv);
  else build_frame_regs_t<TYPE, M, M>(xu, xl, origin, hw, L, rtab, roff, lane, cw, fv);
}

// DCCorrection (common.cpp:56-75) in place on pw[0..half] (LDS), replica of at most 64*REP bins.
template <int REP>
__device__ __forceinline__ void dc_correction_ilp(double* pw, double f0, int fs, int fft_size, int lane) {
  const int upper = 2 + (int)(f0 * fft_size / fs);
  const int nrep = upper - 1;
  const double inv_fft = 1.0 / fft_size;
  double rep[REP];
#pragma unroll
  for (int r = 0; r < REP; ++r) {
    const int i = lane + 64 * r;
    rep[r] = 0.0;
    if (i < nrep) rep[r] = interp1q_lds(f0, -(double)fs * inv_fft, pw, upper + 1, (double)i * fs * inv_fft);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < REP; ++r) {
    const int i = lane + 64 * r;
    if (i < nrep) pw[i] += rep[r];
  }
  __syncthreads();
}

// LinearSmoothing (common.cpp:77-111): in[0..half] (LDS) -> out[m] registers (bins lane + 64 m,
// m < MB; bin half is m = MB-1 on lane 0).  seg: LDS scratch >= 64*CH doubles, CH = per-lane chunk
// bound of the blocked cumulative sum.  Ends with a barrier.
template <int MB, int CH>
__device__ __forceinline__ void linear_smoothing_ilp(const double* in, double width, int fs, int fft_size,
                                                     double* seg, double (&out)[MB], int lane) {
  const int half = fft_size / 2;
  const double inv_fft = 1.0 / fft_size;
  const int b = (int)(width * fft_size / fs) + 1;
  const int len = half + 2 * b + 1;
  const int chunk = (len + 63) / 64;                       // <= CH
  const int beg = lane * chunk;
  double vals[CH];
#pragma unroll
  for (int q = 0; q < CH; ++q) {
    const int i = beg + q;
    const int src = i < b ? b - i : (i < half + b ? i - b : half - (i - (half + b)));
    vals[q] = (q < chunk && i < len) ? in[src] * fs * inv_fft : 0.0;
  }
  double run = 0.0;
#pragma unroll
  for (int q = 0; q < CH; ++q) {
    run += vals[q];
    vals[q] = run;
  }
  const double carry = wave_scan_incl(run) - run;
#pragma unroll
  for (int q = 0; q < CH; ++q) {
    const int i = beg + q;
    if (q < chunk && i < len) seg[i] = vals[q] + carry;
  }
  __syncthreads();
  const double origin = -(b - 0.5) * fs * inv_fft;
  const double step = (double)fs * inv_fft;
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    const int i = lane + 64 * m;
    out[m] = 0.0;
    if (i <= half) {
      const double lo_x = (double)i * inv_fft * fs - width / 2.0;
      const double lo = interp1q_lds(origin, step, seg, len, lo_x);
      const double hi = interp1q_lds(origin, step, seg, len, lo_x + width);
      out[m] = (hi - lo) / width;
    }
  }
  __syncthreads();
}

}  // namespace wm

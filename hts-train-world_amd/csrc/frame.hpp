// frame.hpp -- F0-adaptive analysis frames built directly in registers (one wavefront per frame).
//
// Restates GetWindowedWaveform of D4C (externs/WORLD_v2/src/d4c.cpp:21-84: Hann or Blackman over `ratio`
// periods, not normalised) and of CheapTrick (cheaptrick.cpp:87-142: Hann over 3 periods, L2-normalised):
//
//   waveform[i] = x[clamp(origin + i - hw)] * window[i] + randn * 1e-12          i = 0 .. 2 hw
//   waveform[i] -= window[i] * (sum waveform / sum window)
//
// without staging the frame in LDS: every lane generates the samples of its own FFT operand and keeps them in
// registers (window.hpp builds the same frame in LDS; it costs three LDS passes per frame and the LDS bytes of a
// whole frame beside the FFT image).  Two layouts:
//
//   frame_strided  x[q]  = sample 64 q + lane              (operand of a complex transform over the samples)
//   frame_packed   v[m]  = (sample 2n, sample 2n + 1), n = 64 m + lane   (packed operand of a real transform)
//
// The window's cosine comes from a per-lane sincospi() base and one complex rotation per step (window.hpp CosGen).
// The loads of a frame (waveform and randn table) are issued back to back with clamped addresses, so a frame costs
// one memory round trip; register groups that lie wholly beyond the window are skipped by wave-uniform branches.
#pragma once
#include "common.hpp"
#include "window.hpp"

// wave-uniform skips of register groups beyond the window (1) or straight-line predicated code (0)
#ifndef WM_FRAME_SKIP
#define WM_FRAME_SKIP 1
#endif
#if WM_FRAME_SKIP
#define WM_FRAME_BRANCH(c) (c)
#else
#define WM_FRAME_BRANCH(c) true
#endif

namespace wm {

struct FrameGeom {
  int hw, L, origin;
  double a;          // window angle per sample, in units of pi
};

__device__ __forceinline__ FrameGeom frame_geom(int fs, double f0, double pos, double ratio) {
  FrameGeom g;
  g.hw = matlab_round(ratio * fs / f0 / 2.0);          // d4c.cpp:55-56, cheaptrick.cpp:118-119
  g.L = 2 * g.hw + 1;
  g.origin = matlab_round(pos * fs + 0.001);           // d4c.cpp:28, cheaptrick.cpp:96
  g.a = uniform_d(2.0 * f0 / (ratio * fs));            // cos(pi * a * (i - hw)): d4c.cpp:36-37, cheaptrick.cpp:101-102
  return g;
}

// x[q] = sample 64 q + lane of the frame, q < QX (zero beyond the window); returns sum of squares in `pwr`.
// KEEP_W: the window values stay in registers between the two passes (QX more doubles); otherwise the second
// pass regenerates them with a fresh rotation (long frames, where the registers are worth more than the flops).
// Loads are issued in groups of at most 16 per array.
// GMAX: registers per load group; groups wholly beyond the window are skipped (the usual D4C frame of fft 2048 is
// 4 periods = 250 ... 900 samples of the 1024 its 16 registers cover: with groups of 8 the upper half costs nothing
// from 125 Hz up).
template <int TYPE, int QX, bool KEEP_W, int GMAX = 16>
__device__ __forceinline__ void frame_strided(const double* __restrict__ xu, int xl, const FrameGeom& fg,
                                              const uint32_t* __restrict__ rtab, int roff, int lane,
                                              double (&x)[QX], double& pwr) {
  constexpr int G = QX < GMAX ? QX : GMAX;
  static_assert(QX % G == 0, "QX is a multiple of the load group");
  const int L = fg.L;
  CosGen g;
  g.init(fg.a, lane - fg.hw, 64);
  const CosGen g0 = g;
  double w[KEEP_W ? QX : 1];
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int c = 0; c < QX / G; ++c) {
    double xv[G];
    uint32_t rv[G];
    // No branch around the single loads: behind one, every register is its own trip to memory (the compiler waits
    // for the draw right where it is loaded); registers beyond the window re-read sample L - 1.  Long frames (more
    // than one group of 16) skip whole groups beyond the window: 64 unconditional loads cost the fft-4096 centroid
    // kernel 156 bytes of spilled registers per lane.
#pragma unroll
    for (int r = 0; r < G; ++r) {
      xv[r] = 0.0;
      rv[r] = 0u;
    }
    if (QX == G || 64 * G * c < L) {
#pragma unroll
      for (int r = 0; r < G; ++r) {
        const int q = c * G + r;
        const int ic = imin(64 * q + lane, L - 1);
        xv[r] = xu[imin(xl - 1, imax(0, fg.origin + ic - fg.hw))];
        rv[r] = rtab[roff + ic];
      }
    }
#pragma unroll
    for (int r = 0; r < G; ++r) {
      const int q = c * G + r;
      x[q] = 0.0;
      if (KEEP_W) w[q] = 0.0;
      if (WM_FRAME_BRANCH(64 * q < L)) {
        const bool in = 64 * q + lane < L;
        const double wv = window_value<TYPE>(g.c);
        const double val = xv[r] * wv + ((double)rv[r] / 268435456.0 - 6.0) * kSafe;
        x[q] = in ? val : 0.0;
        s1 += x[q];
        s2 += in ? wv : 0.0;
        if (KEEP_W) w[q] = in ? wv : 0.0;
        g.next();
      }
    }
  }
  const double coef = wave_sum(s1) / wave_sum(s2);
  double p = 0.0;
  g = g0;
#pragma unroll
  for (int q = 0; q < QX; ++q) {
    if (WM_FRAME_BRANCH(64 * q < L)) {
      double wv;
      if (KEEP_W) {
        wv = w[q];
      } else {
        wv = 64 * q + lane < L ? window_value<TYPE>(g.c) : 0.0;
        g.next();
      }
      x[q] -= wv * coef;
      p += x[q] * x[q];
    }
  }
  pwr = wave_sum(p);
}

// The same frame written to memory instead of registers: out[64 q + lane] = sample 64 q + lane for 64 q < L (nothing
// beyond), in two passes over groups of 16 registers (long frames: 64 samples per lane would not leave room for
// anything else).  Every lane re-reads only what it wrote itself.  Returns the sum of squares.
template <int TYPE, int QX>
__device__ __forceinline__ double frame_strided_to_memory(const double* __restrict__ xu, int xl, const FrameGeom& fg,
                                                          const uint32_t* __restrict__ rtab, int roff, int lane,
                                                          double* __restrict__ out) {
  constexpr int G = 16;
  static_assert(QX % G == 0, "QX is a multiple of the load group");
  const int L = fg.L;
  CosGen g;
  g.init(fg.a, lane - fg.hw, 64);
  const CosGen g0 = g;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll 1
  for (int c = 0; c < QX / G; ++c) {
    if (64 * G * c >= L) break;                          // wave-uniform
    double xv[G];
    uint32_t rv[G];
#pragma unroll
    for (int r = 0; r < G; ++r) {
      const int ic = imin(64 * (c * G + r) + lane, L - 1);
      xv[r] = xu[imin(xl - 1, imax(0, fg.origin + ic - fg.hw))];
      rv[r] = rtab[roff + ic];
    }
#pragma unroll
    for (int r = 0; r < G; ++r) {
      const int i = 64 * (c * G + r) + lane;
      const bool in = i < L;
      const double wv = window_value<TYPE>(g.c);
      const double val = in ? xv[r] * wv + ((double)rv[r] / 268435456.0 - 6.0) * kSafe : 0.0;
      s1 += val;
      s2 += in ? wv : 0.0;
      out[i] = val;
      g.next();
    }
  }
  const double coef = wave_sum(s1) / wave_sum(s2);
  double p = 0.0;
  g = g0;
#pragma unroll 1
  for (int c = 0; c < QX / G; ++c) {
    if (64 * G * c >= L) break;
    double v[G];
#pragma unroll
    for (int r = 0; r < G; ++r) v[r] = out[64 * (c * G + r) + lane];
#pragma unroll
    for (int r = 0; r < G; ++r) {
      const int i = 64 * (c * G + r) + lane;
      const double wv = i < L ? window_value<TYPE>(g.c) : 0.0;
      const double val = v[r] - wv * coef;
      p += val * val;
      out[i] = val;
      g.next();
    }
  }
  return wave_sum(p);
}

// v[m] = (sample 2n, sample 2n + 1), n = 64 m + lane, m < M: the packed operand of a real FFT of 128 M points.
// NORMALISE: CheapTrick's L2 normalisation of the window (cheaptrick.cpp:105-106), applied as a multiplication
// by the reciprocal of sqrt(sum w^2).
// The window values are not kept between the passes (they would double the registers of the frame): each pass
// regenerates them by rotation from the two saved bases.  Loads are issued in groups of 8 pairs.
template <int TYPE, bool NORMALISE, int M, bool KEEP_W = false>
__device__ __forceinline__ void frame_packed(const double* __restrict__ xu, int xl, const FrameGeom& fg,
                                             const uint32_t* __restrict__ rtab, int roff, int lane,
                                             cpx (&v)[M]) {
  constexpr int G = M < 8 ? M : 8;
  static_assert(M % G == 0, "M is a multiple of the load group");
  const int L = fg.L;
  CosGen ge0, go0;
  ge0.init(fg.a, 2 * lane - fg.hw, 128);
  go0.init(fg.a, 2 * lane + 1 - fg.hw, 128);
  // KEEP_W: the window values of the first pass stay in registers for the later ones (2 M doubles; CheapTrick at fft
  // 1024 has them to spare at three waves per SIMD) instead of being regenerated by rotation in every pass
  double kw0[KEEP_W ? M : 1], kw1[KEEP_W ? M : 1];
  if (KEEP_W) {
#pragma unroll
    for (int m = 0; m < (KEEP_W ? M : 1); ++m) kw0[m] = kw1[m] = 0.0;
  }
  double inv_scale = 1.0;
  if (NORMALISE) {
    CosGen he = ge0, ho = go0;
    double e = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (WM_FRAME_BRANCH(128 * m < L)) {
        const double w0 = 128 * m + 2 * lane < L ? window_value<TYPE>(he.c) : 0.0;
        const double w1 = 128 * m + 2 * lane + 1 < L ? window_value<TYPE>(ho.c) : 0.0;
        if (KEEP_W) {
          kw0[KEEP_W ? m : 0] = w0;
          kw1[KEEP_W ? m : 0] = w1;
        }
        e += w0 * w0 + w1 * w1;
        he.next();
        ho.next();
      }
    }
    inv_scale = 1.0 / sqrt(wave_sum(e));
    if (KEEP_W) {
#pragma unroll
      for (int m = 0; m < (KEEP_W ? M : 1); ++m) {
        kw0[m] *= inv_scale;
        kw1[m] *= inv_scale;
      }
    }
  }
  constexpr bool HAVE_W = KEEP_W && NORMALISE;        // the first pass that makes window values is the energy pass
  CosGen ge = ge0, go = go0;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int c = 0; c < M / G; ++c) {
    double xa[G], xb[G];
    uint32_t ra[G], rb[G];
#pragma unroll
    for (int r = 0; r < G; ++r) {
      xa[r] = xb[r] = 0.0;
      ra[r] = rb[r] = 0u;
    }
    if (M == G || 128 * G * c < L) {                    // a group wholly beyond the window is not fetched (wave-uniform)
#pragma unroll
      for (int r = 0; r < G; ++r) {
        const int m = c * G + r;
        const int i0 = imin(128 * m + 2 * lane, L - 1), i1 = imin(128 * m + 2 * lane + 1, L - 1);   // no branch: see above
        xa[r] = xu[imin(xl - 1, imax(0, fg.origin + i0 - fg.hw))];
        xb[r] = xu[imin(xl - 1, imax(0, fg.origin + i1 - fg.hw))];
        ra[r] = rtab[roff + i0];
        rb[r] = rtab[roff + i1];
      }
    }
#pragma unroll
    for (int r = 0; r < G; ++r) {
      const int m = c * G + r;
      v[m] = make_double2(0.0, 0.0);
      if (WM_FRAME_BRANCH(128 * m < L)) {
        const bool in0 = 128 * m + 2 * lane < L, in1 = 128 * m + 2 * lane + 1 < L;
        double w0, w1;
        if (HAVE_W) {                                  // zero beyond the window already
          w0 = kw0[KEEP_W ? m : 0];
          w1 = kw1[KEEP_W ? m : 0];
        } else {
          w0 = NORMALISE ? window_value<TYPE>(ge.c) * inv_scale : window_value<TYPE>(ge.c);
          w1 = NORMALISE ? window_value<TYPE>(go.c) * inv_scale : window_value<TYPE>(go.c);
          ge.next();
          go.next();
        }
        const double v0 = xa[r] * w0 + ((double)ra[r] / 268435456.0 - 6.0) * kSafe;
        const double v1 = xb[r] * w1 + ((double)rb[r] / 268435456.0 - 6.0) * kSafe;
        v[m].x = in0 ? v0 : 0.0;
        v[m].y = in1 ? v1 : 0.0;
        s1 += v[m].x + v[m].y;
        s2 += (in0 ? w0 : 0.0) + (in1 ? w1 : 0.0);
        if (KEEP_W && !HAVE_W) {
          kw0[KEEP_W ? m : 0] = in0 ? w0 : 0.0;
          kw1[KEEP_W ? m : 0] = in1 ? w1 : 0.0;
        }
      }
    }
  }
  const double coef = wave_sum(s1) / wave_sum(s2);
  ge = ge0;
  go = go0;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    if (WM_FRAME_BRANCH(128 * m < L)) {
      if (KEEP_W) {
        v[m].x -= kw0[KEEP_W ? m : 0] * coef;
        v[m].y -= kw1[KEEP_W ? m : 0] * coef;
      } else {
        const double w0 = NORMALISE ? window_value<TYPE>(ge.c) * inv_scale : window_value<TYPE>(ge.c);
        const double w1 = NORMALISE ? window_value<TYPE>(go.c) * inv_scale : window_value<TYPE>(go.c);
        v[m].x -= (128 * m + 2 * lane < L ? w0 : 0.0) * coef;
        v[m].y -= (128 * m + 2 * lane + 1 < L ? w1 : 0.0) * coef;
        ge.next();
        go.next();
      }
    }
  }
}

}  // namespace wm

// d4c_wide.hpp -- D4CGeneralBody + GetAperiodicity (d4c.cpp:290-333) for the frames NO other kernel of this library
// takes: fft_size_d4c = 8192 (fs above 48.1 kHz: 88.2 / 96 kHz) and f0 >= fs / 16, where the smoothing mirrors reach up
// to half the spectrum.  Until round 5 those frames kept the default row 1 - 1e-12 (the one place the library returned
// something else than the reference on valid input); the reference analyses them (d4c.cpp:337-397 has no such limit).
//
// Nobody's speech has an f0 of 6 kHz: this kernel exists for parity, not for speed, and is written to be READ against the
// reference -- one workgroup of 256 threads per frame, every array in LDS (143 KB of the CU's 160), no wavefront tricks:
//   * the three analysis frames are at most 2 round(2 fs / f0) + 1 <= 65 samples at f0 >= fs / 16, and a band's slice is
//     513 taps, so every spectrum is a DIRECT DFT per bin (a twiddle per bin by sincospi, rotated per sample) -- the
//     zero padding to 8192 points costs nothing that way, and no 8192-point transform has to exist;
//   * DCCorrection / LinearSmoothing as common.cpp:56-111 on an array with mirror margins of a full half spectrum, the
//     cumulative sum by a two-level block scan;
//   * "the sum of all but the bnd + 1 largest of 4097 values" (d4c.cpp:215-220) by a bisection on the bit patterns of
//     the (non-negative) doubles: the K-th largest value T, then the values below T plus the copies of T that are left.
// Included by d4c.hip.
#pragma once

namespace wm {

constexpr int kWideThreads = 256;
constexpr int kWideMaxTaps = 1024;        // samples of a frame / taps of a band slice that fit the LDS operand

struct D4cWideLds {
  // doubles: [ext: 3 H + 8 | cen: H + 1 (+pad) | wv: kWideMaxTaps | red: 2 * kWideThreads]
  __host__ __device__ static size_t doubles(int FD) {
    const size_t H = (size_t)FD / 2;
    return (3 * H + 8) + (H + 8) + kWideMaxTaps + 2 * kWideThreads;
  }
};

__device__ __forceinline__ double wide_block_sum(double v, double* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = kWideThreads / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}
__device__ __forceinline__ int wide_block_sum_i(int v, int* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = kWideThreads / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  const int r = red[0];
  __syncthreads();
  return r;
}

// GetWindowedWaveform of D4C (d4c.cpp:21-84) into wv[0 .. L): type 1 Hann / 2 Blackman over `ratio` periods.
__device__ __forceinline__ int wide_frame(const double* __restrict__ xu, int xl, int fs, double f0, double pos, int type,
                                          double ratio, const uint32_t* __restrict__ rtab, int roff, double* wv,
                                          double* red) {
  const int hw = matlab_round(ratio * fs / f0 / 2.0);
  const int L = 2 * hw + 1;
  const int origin = matlab_round(pos * fs + 0.001);
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < L; i += kWideThreads) {
    const double p = (2.0 * (i - hw) / ratio) / fs;
    const double c1 = cospi(p * f0);
    const double w = type == 1 ? 0.5 * c1 + 0.5 : 0.42 + 0.5 * c1 + 0.08 * cospi(p * f0 * 2.0);
    const int si = imin(xl - 1, imax(0, origin + i - hw));
    const double v = xu[si] * w + ((double)rtab[roff + i] / 268435456.0 - 6.0) * kSafe;
    wv[i] = v;
    s1 += v;
    s2 += w;
  }
  s1 = wide_block_sum(s1, red);
  s2 = wide_block_sum(s2, red);
  const double coef = s1 / s2;
  for (int i = threadIdx.x; i < L; i += kWideThreads) {
    const double p = (2.0 * (i - hw) / ratio) / fs;
    const double c1 = cospi(p * f0);
    const double w = type == 1 ? 0.5 * c1 + 0.5 : 0.42 + 0.5 * c1 + 0.08 * cospi(p * f0 * 2.0);
    wv[i] -= w * coef;
  }
  __syncthreads();
  return L;
}

// DCCorrection (common.cpp:56-75) in place on a[0 .. H]
__device__ __forceinline__ void wide_dc_correction(double* a, int H, double f0, int fs, int FD) {
  const int upper = 2 + (int)(f0 * FD / fs);
  const int nrep = imin(upper - 1, H + 1);
  constexpr int kPer = 17;                                   // ceil(4097 / 256)
  double rep[kPer];
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const int i = threadIdx.x + kWideThreads * q;
    rep[q] = 0.0;
    if (i < nrep) {
      const double axis = (double)i * fs / FD;
      const double qq = (axis - f0) / (-(double)fs / FD);     // interp1Q(x0 = f0, dx = -fs / FD, ...)
      const int b = (int)qq;
      const double frac = qq - b;
      const int b0 = imin(imax(b, 0), H), b1 = imin(b0 + 1, H);
      const double dy = (b == upper) ? 0.0 : a[b1] - a[b0];   // delta_y[n - 1] = 0, n = upper + 1
      rep[q] = a[b0] + dy * frac;
    }
  }
  __syncthreads();                                           // every read before any write (common.cpp:62-74)
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const int i = threadIdx.x + kWideThreads * q;
    if (i < nrep) a[i] += rep[q];
  }
  __syncthreads();
}

// LinearSmoothing (common.cpp:27-46, 77-111) in place on arr[0 .. H]; arr has H + 2 free doubles on either side.
__device__ __forceinline__ void wide_linear_smoothing(double* arr, int H, double width, int fs, int FD, double* red) {
  const double wq = width * FD / fs;
  const int b = imin((int)wq + 1, H);                         // the caller's frames keep b <= H (D4cRunRarePred)
  const int len = H + 2 * b + 1;
  for (int t = 1 + threadIdx.x; t <= b; t += kWideThreads) { // mirror (common.cpp:85-92)
    arr[-t] = arr[t];
    arr[H + t] = arr[H - t];
  }
  __syncthreads();
  double* ext = arr - b;
  // cumulative sum * fs / FD (common.cpp:38-41): every thread its run of consecutive entries, then the runs' offsets
  const int chunk = (len + kWideThreads - 1) / kWideThreads;
  const int beg = threadIdx.x * chunk, end = imin(beg + chunk, len);
  double run = 0.0;
  for (int i = beg; i < end; ++i) {
    run += ext[i] * fs / FD;
    ext[i] = run;
  }
  red[threadIdx.x] = run;
  __syncthreads();
  double offset = 0.0;
  for (int t = 0; t < (int)threadIdx.x; ++t) offset += red[t];
  __syncthreads();
  for (int i = beg; i < end; ++i) ext[i] += offset;
  if (threadIdx.x == 0) ext[len] = 0.0;                       // never used with a non-zero weight (see dy below)
  __syncthreads();
  const double c_lo = (b - 0.5) - 0.5 * wq, c_hi = c_lo + wq; // knot of bin i: i + c (common.cpp:99-108)
  const int bl = (int)c_lo, bh = (int)c_hi;
  const double fl = c_lo - bl, fh = c_hi - bh;
  constexpr int kPer = 17;
  double out[kPer];
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const int i = threadIdx.x + kWideThreads * q;
    out[q] = 0.0;
    if (i <= H) {
      const int il = i + bl, ih = i + bh;
      const double dl = il >= len - 1 ? 0.0 : ext[il + 1] - ext[il];     // interp1Q: delta_y[n - 1] = 0
      const double dh = ih >= len - 1 ? 0.0 : ext[ih + 1] - ext[ih];
      const double l = ext[imin(il, len - 1)] + dl * fl;
      const double h = ext[imin(ih, len - 1)] + dh * fh;
      out[q] = (h - l) / width;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const int i = threadIdx.x + kWideThreads * q;
    if (i <= H) arr[i] = out[q];
  }
  __syncthreads();
}

template <int FD>
__global__ __launch_bounds__(kWideThreads) void d4c_wide_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int* __restrict__ frame_utt, const double* __restrict__ tpos, const double* __restrict__ f0,
    const int* __restrict__ rng_off, const uint32_t* __restrict__ rtab, int fs, D4CTables tab, int out_fft,
    const int* __restrict__ perm, const int* __restrict__ n_listed, double* __restrict__ ap) {
  constexpr int H = FD / 2;
  static_assert((H + 1 + kWideThreads - 1) / kWideThreads <= 17, "per-thread bins of a half spectrum");
  extern __shared__ __attribute__((aligned(16))) double wide_lds[];
  double* ext0 = wide_lds;                       // 3 H + 8
  double* arr = ext0 + H + 2;                    // arr[-(H + 2) .. 2 H + 5]
  double* cen = ext0 + 3 * H + 8;                // H + 8
  double* wv = cen + H + 8;                      // kWideMaxTaps
  double* red = wv + kWideMaxTaps;               // 2 * kWideThreads
  const int n_run = *n_listed;
  const int out_bins = out_fft / 2 + 1;
  const int wl = tab.window_length, hwl = wl / 2;
  const int bnd = matlab_round(FD * 8.0 / wl);
  for (int li = blockIdx.x; li < n_run; li += gridDim.x) {
    const int frame = perm[li];
    const int u = frame_utt[frame];
    const double* xu = x + x_off[u];
    const int xl = x_len[u];
    const double f0v = f0[frame];
    const double cf0 = f0v > kFloorF0D4C ? f0v : kFloorF0D4C;         // d4c.cpp:381
    const double pos = tpos[frame];
    const int roff = rng_off[frame];
    const int Lw = 2 * matlab_round(2.0 * fs / cf0) + 1;
    double* row = ap + frame * (int64_t)out_bins;
    if (Lw > kWideMaxTaps || wl > kWideMaxTaps) {
      // cannot happen for this kernel's frames (f0 >= fs / 16: windows of at most 65 samples, slices of 513 / 557 taps);
      // if a caller ever lists such a frame it gets the default row rather than an unwritten one
      for (int i = threadIdx.x; i < out_bins; i += kWideThreads) row[i] = 1.0 - kSafe;
      continue;
    }
    // ---- GetStaticCentroid (d4c.cpp:125-142) ----
    for (int k = threadIdx.x; k <= H; k += kWideThreads) cen[k] = 0.0;
    __syncthreads();
    for (int side = 0; side < 2; ++side) {
      const double cpos = side == 0 ? pos - 0.25 / cf0 : pos + 0.25 / cf0;
      const int L = wide_frame(xu, xl, fs, cf0, cpos, 2, 4.0, rtab, roff + side * Lw, wv, red);
      double p = 0.0;
      for (int i = threadIdx.x; i < L; i += kWideThreads) p += wv[i] * wv[i];       // d4c.cpp:96-100
      p = wide_block_sum(p, red);
      const double inv = 1.0 / sqrt(p);
      for (int i = threadIdx.x; i < L; i += kWideThreads) wv[i] *= inv;
      __syncthreads();
      for (int k = threadIdx.x; k <= H; k += kWideThreads) {          // X1 = DFT(x), X2 = DFT((i + 1) x), :101-118
        double sn, cs;
        sincospi(2.0 * (double)k / FD, &sn, &cs);
        const double wr = cs, wi = -sn;
        double cr = 1.0, ci = 0.0, r1 = 0.0, i1 = 0.0, r2 = 0.0, i2 = 0.0;
        for (int n = 0; n < L; ++n) {
          const double v = wv[n], v2 = v * (n + 1.0);
          r1 += v * cr; i1 += v * ci;
          r2 += v2 * cr; i2 += v2 * ci;
          const double nr = cr * wr - ci * wi;
          ci = cr * wi + ci * wr;
          cr = nr;
        }
        cen[k] += r2 * r1 + i1 * i2;
      }
      __syncthreads();
    }
    wide_dc_correction(cen, H, cf0, fs, FD);                          // d4c.cpp:139
    // ---- GetSmoothedPowerSpectrum (d4c.cpp:148-164) ----
    {
      const int L = wide_frame(xu, xl, fs, cf0, pos, 1, 4.0, rtab, roff + 2 * Lw, wv, red);
      for (int k = threadIdx.x; k <= H; k += kWideThreads) {
        double sn, cs;
        sincospi(2.0 * (double)k / FD, &sn, &cs);
        const double wr = cs, wi = -sn;
        double cr = 1.0, ci = 0.0, re = 0.0, im = 0.0;
        for (int n = 0; n < L; ++n) {
          const double v = wv[n];
          re += v * cr; im += v * ci;
          const double nr = cr * wr - ci * wi;
          ci = cr * wi + ci * wr;
          cr = nr;
        }
        arr[k] = re * re + im * im;
      }
      __syncthreads();
    }
    wide_dc_correction(arr, H, cf0, fs, FD);
    wide_linear_smoothing(arr, H, cf0, fs, FD, red);
    // ---- GetStaticGroupDelay (d4c.cpp:170-186) ----
    for (int k = threadIdx.x; k <= H; k += kWideThreads) arr[k] = cen[k] / arr[k];
    __syncthreads();
    wide_linear_smoothing(arr, H, cf0 / 2.0, fs, FD, red);
    for (int k = threadIdx.x; k <= H; k += kWideThreads) cen[k] = arr[k];
    __syncthreads();
    wide_linear_smoothing(arr, H, cf0, fs, FD, red);
    for (int k = threadIdx.x; k <= H; k += kWideThreads) cen[k] -= arr[k];              // the group delay
    __syncthreads();
    // ---- GetCoarseAperiodicity (d4c.cpp:192-223) ----
    double coarse[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int band = 0; band < tab.nap; ++band) {
      const int center = (int)(kFreqInterval * (band + 1) * FD / fs);
      for (int j = threadIdx.x; j < wl; j += kWideThreads) wv[j] = cen[center - hwl + j] * tab.nuttall[j];
      __syncthreads();
      double tot = 0.0;
      for (int k = threadIdx.x; k <= H; k += kWideThreads) {
        double sn, cs;
        sincospi(2.0 * (double)k / FD, &sn, &cs);
        const double wr = cs, wi = -sn;
        double cr = 1.0, ci = 0.0, re = 0.0, im = 0.0;
        for (int n = 0; n < wl; ++n) {
          const double v = wv[n];
          re += v * cr; im += v * ci;
          const double nr = cr * wr - ci * wi;
          ci = cr * wi + ci * wr;
          cr = nr;
        }
        const double pw = re * re + im * im;
        arr[k] = pw;
        tot += pw;
      }
      tot = wide_block_sum(tot, red);
      // the K = bnd + 1 largest are left out (d4c.cpp:215-220: cum[H - bnd - 1] of the ascending sort): T = the K-th
      // largest bit pattern, by bisection on "how many values are >= T"
      const int K = bnd + 1;
      unsigned long long lo = 0ull, hi = 0x7ff0000000000000ull;       // cnt(lo) = H + 1 >= K; cnt(hi) = 0 (finite powers)
      for (int it = 0; it < 64 && hi - lo > 1ull; ++it) {
        const unsigned long long mid = lo + (hi - lo) / 2ull;
        int c = 0;
        for (int k = threadIdx.x; k <= H; k += kWideThreads) c += (unsigned long long)__double_as_longlong(arr[k]) >= mid ? 1 : 0;
        c = wide_block_sum_i(c, (int*)red);
        if (c >= K) lo = mid; else hi = mid;
      }
      const double T = __longlong_as_double((long long)lo);
      double below = 0.0;
      int n_gt = 0, n_eq = 0;
      for (int k = threadIdx.x; k <= H; k += kWideThreads) {
        const double v = arr[k];
        if (v < T) below += v;
        n_gt += v > T ? 1 : 0;
        n_eq += v == T ? 1 : 0;
      }
      below = wide_block_sum(below, red);
      n_gt = wide_block_sum_i(n_gt, (int*)red);
      n_eq = wide_block_sum_i(n_eq, (int*)red);
      const double low = below + (double)(n_eq - (K - n_gt)) * T;
      double c = 10.0 * log10(low / tot);
      c = c + (cf0 - 100.0) / 50.0;                                   // d4c.cpp:309-311
      c = 0.0 < c ? 0.0 : c;                                          // MyMinDouble(0.0, c): a NaN stays a NaN
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (j == band) coarse[j] = c;
      __syncthreads();
    }
    // ---- GetAperiodicity (d4c.cpp:325-333): interp1 over {0, 3000 i, fs / 2}, then 10^(x / 20) ----
    const int nap = tab.nap;
    for (int i = threadIdx.x; i < out_bins; i += kWideThreads) {
      const double f = (double)i * fs / out_fft;
      int kk = (int)(f / kFreqInterval);
      kk = kk > nap ? nap : kk;
      const double x0 = kk * kFreqInterval, x1 = kk == nap ? fs / 2.0 : (kk + 1) * kFreqInterval;
      double y0 = kk == 0 ? -60.0 : 0.0, y1 = kk + 1 > nap ? -kSafe : 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (kk == j + 1) y0 = coarse[j];
        if (kk + 1 == j + 1 && kk + 1 <= nap) y1 = coarse[j];
      }
      const double s = (f - x0) / (x1 - x0);
      row[i] = pow(10.0, (y0 + s * (y1 - y0)) / 20.0);
    }
    __syncthreads();
  }
}

}  // namespace wm

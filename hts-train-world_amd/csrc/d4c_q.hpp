// d4c_q.hpp -- D4CGeneralBody (d4c.cpp:290-316) + GetAperiodicity (:325-333) as ONE kernel on the QUARTER-size
// wavefront transform: fft_size_d4c = 2048 (fs 12.1 ... 24 kHz: the headline's 16 kHz) on the 512-point engine, at
// THREE waves per SIMD.  Included by d4c.hip after d4c_big.hpp (whose real_power_halves it shares).
//
// Why: d4c_kernel<2048, 2, false> (d4c.hip) runs the analysis on the 1024-point engine: 240 registers and 17.4 KB of
// LDS, two waves per SIMD, which issue vector instructions 74-76 % of the cycles and wait the rest (DESIGN.md
// section 3, item 30).  A third wave needs <= 168 registers and <= 13 312 bytes of LDS, i.e. every transform on the
// 512-point engine (8 complex values per lane and operand, a 9.2 KB image):
//
//   centroid   z = s x + j (i + 1) x as in d4c.hip; the FD-point complex transform is one radix-4 decimation-in-
//              frequency step by hand, Z[4 j + q] = FFT_NS(u_q)[j], u_q[n] = (z[n] + (-j)^q z[n + NS]) W_FD^(n q)
//              (the usual frame has at most FD / 2 = 2 NS samples).  Z[k] Z[FD - k] couples q = 0 and q = 2 with
//              themselves and q = 1 with q = 3: E1 waits in registers while E3 is transformed.  The lower NS samples of
//              the frame stay in registers across the four sub-transforms, the upper ones (frames longer than NS
//              samples: f0 below ~125 Hz at 16 kHz) wait in the 4 KB of LDS beside the image, every lane re-reading
//              what it stored itself.
//   spectra    the real transforms of FD points (Hann frame, band slices) by even / odd bins: the even bins are the
//              real transform of FD / 2 points as it is, the odd bins come from FFT_NS(v[n] W_{FD/2}^n) with the
//              split pairing j <-> NS - 1 - j (real_power_halves, d4c_big.hpp).
//
// The spectrum-domain stages (DCCorrection, LinearSmoothing, sort + peel, the output row) are those of d4c_kernel:
// same code, same LDS layout with mirror margins (spectrum.hpp).
//
// ONE_BAND: number_of_aperiodicities == 1 (fs < 18 kHz): the group delay goes to LDS once and is not kept in
// registers across a loop over bands.
#pragma once

namespace wm {

template <int FD> struct D4cQ {
  static constexpr int NS = FD / 4, MS = NS / 64, H = FD / 2, M = H / 64, MB = M + 1;
  static constexpr int kBM = FD / 16;
  static constexpr int kImg = 2 * FftLds<NS>::kElems;            // doubles of the FFT image
  static constexpr int kPark = kImg;                             // the parked upper half of a long frame: NS doubles
  static constexpr int kRegion = SmoothCfg<H, kBM>::kRegion;
  static constexpr int kHeads = (MB + 3) * 64;
  static constexpr int kA = kImg + NS;
  static constexpr int kB = kRegion > kHeads ? kRegion : kHeads;
  static constexpr int kTot = kA > kB ? kA : kB;
};

// Im(a b): the centroid identity of d4c.hip, Im(Z[k] Z[FD - k]) = 2 s Re(X1[k] conj X2[k])
__device__ __forceinline__ double im_prod(cpx a, cpx b) { return a.x * b.y + a.y * b.x; }

// The frame of one centroid side for the quarter engine (frame.hpp's frame_strided, in halves): sample lane + 64 m of
// the frame in xlo[m], m < MS; sample NS + lane + 64 m in park[lane + 64 m] (LDS, every lane re-reads only what it
// stored itself) when the window reaches into the second quarter (`lng`, wave-uniform).  Same sums in the same order
// as frame_strided, so the same frame bit for bit.  Returns the sum of squares.
// Not frame_strided<..., 2 MS, ...> into one array: its per-register branches turn the array into ONE 1024-bit vector
// value with a select per element (every element access then moves the whole tuple; 2 900 spilled registers here).
template <int TYPE, int MS>
__device__ __forceinline__ double frame_strided_q(const double* __restrict__ xu, int xl, const FrameGeom& fg,
                                                  const uint32_t* __restrict__ rtab, int roff, int lane,
                                                  double (&xlo)[MS], double* park, bool lng) {
  constexpr int NS = 64 * MS;
  const int L = fg.L;
  CosGen g;
  g.init(fg.a, lane - fg.hw, 64);
  const CosGen g0 = g;
  double s1 = 0.0, s2 = 0.0;
  {
    double xv[MS];
    uint32_t rv[MS];
#pragma unroll
    for (int r = 0; r < MS; ++r) {                       // unconditional, clamped: one round trip for the group
      const int ic = imin(64 * r + lane, L - 1);
      xv[r] = xu[imin(xl - 1, imax(0, fg.origin + ic - fg.hw))];
      rv[r] = rtab[roff + ic];
    }
#pragma unroll
    for (int r = 0; r < MS; ++r) {
      const bool in = 64 * r + lane < L;
      const double wv = window_value<TYPE>(g.c);
      const double val = xv[r] * wv + ((double)rv[r] / 268435456.0 - 6.0) * kSafe;
      xlo[r] = in ? val : 0.0;
      s1 += xlo[r];
      s2 += in ? wv : 0.0;
      g.next();
    }
  }
  if (lng) {
    double xv[MS];
    uint32_t rv[MS];
#pragma unroll
    for (int r = 0; r < MS; ++r) {
      const int ic = imin(NS + 64 * r + lane, L - 1);
      xv[r] = xu[imin(xl - 1, imax(0, fg.origin + ic - fg.hw))];
      rv[r] = rtab[roff + ic];
    }
#pragma unroll
    for (int r = 0; r < MS; ++r) {
      const bool in = NS + 64 * r + lane < L;
      const double wv = window_value<TYPE>(g.c);
      const double val = xv[r] * wv + ((double)rv[r] / 268435456.0 - 6.0) * kSafe;
      const double kept = in ? val : 0.0;
      park[lane + 64 * r] = kept;
      s1 += kept;
      s2 += in ? wv : 0.0;
      g.next();
    }
  }
  const double coef = wave_sum(s1) / wave_sum(s2);
  double p = 0.0;
  g = g0;
#pragma unroll
  for (int r = 0; r < MS; ++r) {
    const double wv = 64 * r + lane < L ? window_value<TYPE>(g.c) : 0.0;
    xlo[r] -= wv * coef;
    p += xlo[r] * xlo[r];
    g.next();
  }
  if (lng) {
    double pv[MS];
#pragma unroll
    for (int r = 0; r < MS; ++r) pv[r] = park[lane + 64 * r];
#pragma unroll
    for (int r = 0; r < MS; ++r) {
      const double wv = NS + 64 * r + lane < L ? window_value<TYPE>(g.c) : 0.0;
      const double val = pv[r] - wv * coef;
      p += val * val;
      park[lane + 64 * r] = val;
      g.next();
    }
  }
  return wave_sum(p);
}

// One GetCentroid (d4c.cpp:90-119) at the frame `fg`, added into c0 .. c3 / mid:
//   cq[m] += centroid at bin 4 j + q, j = lane + 64 m, m < MS / 2;   mid (lane 0) += centroid at bin FD / 2.
template <int FD>
__device__ __forceinline__ void d4cq_centroid_side(const double* __restrict__ xu, int xl, const FrameGeom& fg,
                                                   const uint32_t* __restrict__ rtab, int ro,
                                                   FftTw<D4cQ<FD>::NS>& tw, cpx* img, double* park, int lane0,
                                                   double (&c0)[D4cQ<FD>::MS / 2], double (&c1)[D4cQ<FD>::MS / 2],
                                                   double (&c2)[D4cQ<FD>::MS / 2], double (&c3)[D4cQ<FD>::MS / 2],
                                                   double& mid) {
  constexpr int NS = D4cQ<FD>::NS, MS = D4cQ<FD>::MS;
  int lane = opaque_lane(lane0);
  // s: power of two next below the half window length (the mean of the ramp i + 1): exact scaling
  const double s = (double)(1 << (31 - __clz(fg.hw | 1)));
  const bool lng = fg.L > NS;                                     // wave-uniform: the frame reaches into the second quarter
  const int nz = lng ? MS : (fg.L + 63) >> 6;                     // registers of a sub-transform's operand that may be non-zero
  double xlo[MS];
  const double pwr = frame_strided_q<kBlackman, MS>(xu, xl, fg, rtab, ro, lane, xlo, park, lng);
  // normalisation to unit energy (d4c.cpp:96-100) and the 1 / (2 s) of the identity, on the products
  const double scale = uniform_d(1.0 / (2.0 * s * pwr));
  cpx v[MS];
  // The parked half comes back as one group of loads per sub-transform, behind ONE wave-uniform branch (a branch per
  // element would make every element a trip of its own); a short frame takes zeros (what is parked there is stale).
#define WM_D4CQ_UPPER(xh)                                                    \
  double xh[MS];                                                             \
  if (lng) {                                                                 \
    _Pragma("unroll") for (int m = 0; m < MS; ++m) xh[m] = park[lane + 64 * m]; \
  } else {                                                                   \
    _Pragma("unroll") for (int m = 0; m < MS; ++m) xh[m] = 0.0;              \
  }
  // ---- q = 0: u = z[n] + z[n + NS]; pairs with itself, j <-> (NS - j) mod NS ----
  {
    WM_D4CQ_UPPER(xh)
#pragma unroll
    for (int m = 0; m < MS; ++m) {
      const double r = (double)(lane + 64 * m + 1);
      v[m] = make_double2(__builtin_fma(s, xh[m], s * xlo[m]), __builtin_fma(r + NS, xh[m], r * xlo[m]));
    }
  }
  fft_forward_nz<NS>(v, img, tw, lane, nz);
  store_upper<NS>(v, img, lane);
#pragma unroll
  for (int m = 0; m < MS / 2; ++m) {
    const int j = lane + 64 * m;
    cpx pt = img[(NS - j) & (NS - 1)];
    if (m == 0) {                                                 // E[0] pairs with itself (component-wise select:
      pt.x = lane == 0 ? v[0].x : pt.x;                           // see d4c_centroid)
      pt.y = lane == 0 ? v[0].y : pt.y;
    }
    c0[m] += im_prod(v[m], pt) * scale;
  }
  mid += 2.0 * v[MS / 2].x * v[MS / 2].y * scale;                 // lane 0: E[NS / 2] pairs with itself
  wave_sync();
  // ---- q = 2: u = (z[n] - z[n + NS]) W_FD^(2 n); pairs with itself, j <-> NS - 1 - j ----
  lane = opaque_lane(lane);
  tw.fence();
  {
    WM_D4CQ_UPPER(xh)
    cpx w = tw.wsplit;                                            // W_{2 NS}^lane = W_FD^(2 lane)
#pragma unroll
    for (int m = 0; m < MS; ++m) {
      const double r = (double)(lane + 64 * m + 1);
      const cpx d = make_double2(__builtin_fma(-s, xh[m], s * xlo[m]), __builtin_fma(-(r + NS), xh[m], r * xlo[m]));
      v[m] = cmul(d, w);
      w = cmul(w, tw.wstep());                                    // W_{2 NS}^64
    }
  }
  fft_forward_nz<NS>(v, img, tw, lane, nz);
  store_upper<NS>(v, img, lane);
#pragma unroll
  for (int m = 0; m < MS / 2; ++m) c2[m] += im_prod(v[m], img[NS - 1 - (lane + 64 * m)]) * scale;
  wave_sync();
  // ---- q = 1 and q = 3: u_1 = (z[n] - j z[n + NS]) W^n, u_3 = (z[n] + j z[n + NS]) W^(3 n); bins 4 j + 1 pair
  //      E1[j] with E3[NS - 1 - j], bins 4 j + 3 pair E3[j] with E1[NS - 1 - j] ----
  lane = opaque_lane(lane);
  tw.fence();
  const cpx wl = cis_neg2pi((double)lane / (double)FD);          // W_FD^lane
  {
    WM_D4CQ_UPPER(xh)
    cpx w = wl;
    const cpx w64 = cis64(4096 / FD);                             // W_FD^64 (FD <= 4096: an entry of the table)
#pragma unroll
    for (int m = 0; m < MS; ++m) {
      const double r = (double)(lane + 64 * m + 1);
      // - j (a + j b) = b - j a
      const cpx d = make_double2(__builtin_fma(r + NS, xh[m], s * xlo[m]), __builtin_fma(-s, xh[m], r * xlo[m]));
      v[m] = cmul(d, w);
      w = cmul(w, w64);
    }
  }
  fft_forward_nz<NS>(v, img, tw, lane, nz);
  cpx v1[MS];
#pragma unroll
  for (int m = 0; m < MS; ++m) v1[m] = v[m];
  lane = opaque_lane(lane);
  tw.fence();
  {
    WM_D4CQ_UPPER(xh)
    const cpx wl2 = csqr(wl);
    cpx w = cmul(wl2, wl);                                        // W_FD^(3 lane)
    const cpx w64 = cis64(3 * (4096 / FD));                       // W_FD^192
#pragma unroll
    for (int m = 0; m < MS; ++m) {
      const double r = (double)(lane + 64 * m + 1);
      // + j (a + j b) = - b + j a
      const cpx d = make_double2(__builtin_fma(-(r + NS), xh[m], s * xlo[m]), __builtin_fma(s, xh[m], r * xlo[m]));
      v[m] = cmul(d, w);
      w = cmul(w, w64);
    }
  }
  fft_forward_nz<NS>(v, img, tw, lane, nz);
  store_upper<NS>(v, img, lane);
#pragma unroll
  for (int m = 0; m < MS / 2; ++m) c1[m] += im_prod(v1[m], img[NS - 1 - (lane + 64 * m)]) * scale;   // E3[NS - 1 - j]
  store_upper<NS>(v1, img, lane);
#pragma unroll
  for (int m = 0; m < MS / 2; ++m) c3[m] += im_prod(v[m], img[NS - 1 - (lane + 64 * m)]) * scale;    // E1[NS - 1 - j]
  wave_sync();
#undef WM_D4CQ_UPPER
}

template <int FD, bool ONE_BAND>
#ifndef WM_D4CQ_WAVES
#define WM_D4CQ_WAVES 3
#endif
__global__ __launch_bounds__(64, WM_D4CQ_WAVES) void d4cq_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int* __restrict__ frame_utt, const double* __restrict__ tpos, const double* __restrict__ f0,
    const double* __restrict__ ap0, const int* __restrict__ rng_off, const uint32_t* __restrict__ rtab,
    int fs_arg, double threshold, D4CTables tab, int out_fft_arg, int64_t total_frames, const int* __restrict__ perm,
    const int* __restrict__ n_listed, double* __restrict__ ap) {
  typedef D4cQ<FD> G;
  constexpr int NS = G::NS, MS = G::MS, H = G::H, M = G::M, MB = G::MB, kBM = G::kBM;
  static_assert(FD <= 4096, "constant twiddles of the sub-transform operands are tabulated in 64ths of a turn");
  static_assert(kBM % 2 == 0, "the spectrum starts on a 16-byte boundary");
  __shared__ __attribute__((aligned(16))) double smem[G::kTot];
  double* arr = smem + kBM;                   // [-kBM .. H + kBM] spectrum-domain array with mirror margins
  cpx* img = reinterpret_cast<cpx*>(smem);    // FFT image (aliases it)
  double* park = smem + G::kPark;             // upper half of a long frame, beside the image

  const int lane0 = threadIdx.x;
  const int n_run = *n_listed;
  FftTw<NS> tw;
  tw.init(lane0);
  const int out_bins = out_fft_arg / 2 + 1;
  {
    const D4cRunRarePred rare{f0, ap0, threshold, FD, fs_arg};
    WM_FOR_EACH_LISTED(frame, perm + n_run, total_frames - n_run) {   // d4c.cpp:318-323, :380
      if (rare((int)frame)) continue;                                  // the RARE launch owns that row (it may run first)
      double* row = ap + frame * (int64_t)out_bins;
      for (int i = lane0; i < out_bins; i += 64) row[i] = 1.0 - kSafe;
    }
  }
  FramePipe pipe;
  pipe.init(perm, n_run, x, x_off, x_len, frame_utt, tpos, f0, rng_off);
  WM_PHASE_DECL
  WM_FOR_EACH_PIPED(sc, pipe, n_run) {
    WM_PHASE_MARK(0)                                                                  // the pipe's step
    const int64_t frame = sc.frame;
    int lane = opaque_lane(lane0);
    const int fs = opaque_uniform(fs_arg), out_fft = opaque_uniform(out_fft_arg);   // nothing derived is hoisted
    double* row = ap + frame * (int64_t)out_bins;
    const double f0v = sc.f0;
    const double cf0 = uniform_d(f0v > kFloorF0D4C ? f0v : kFloorF0D4C);   // d4c.cpp:381
    const double* xu = sc.xu;
    const int xl = sc.xlen;
    const double pos = uniform_d(sc.tpos);
    const int roff = sc.roff;
    const int Lw = 2 * matlab_round(2.0 * fs / cf0) + 1;

    // ---- GetStaticCentroid (d4c.cpp:125-142): two centroids at pos -/+ 0.25/f0 ----
    double sc_[MB];
    {
      double c0[MS / 2], c1[MS / 2], c2[MS / 2], c3[MS / 2], mid = 0.0;
#pragma unroll
      for (int m = 0; m < MS / 2; ++m) c0[m] = c1[m] = c2[m] = c3[m] = 0.0;
#pragma unroll 1
      for (int side = 0; side < 2; ++side) {
        const double cpos = uniform_d(side == 0 ? pos - 0.25 / cf0 : pos + 0.25 / cf0);
        const FrameGeom fg = frame_geom(fs, cf0, cpos, 4.0);
        d4cq_centroid_side<FD>(xu, xl, fg, rtab, roff + side * Lw, tw, img, park, lane, c0, c1, c2, c3, mid);
      }
      WM_PHASE_MARK(1)                                                                // two centroids
      cpx* arr2 = reinterpret_cast<cpx*>(arr);                   // bins 4 j .. 4 j + 3 as two 16-byte stores
#pragma unroll
      for (int m = 0; m < MS / 2; ++m) {
        arr2[2 * (lane + 64 * m)] = make_double2(c0[m], c1[m]);
        arr2[2 * (lane + 64 * m) + 1] = make_double2(c2[m], c3[m]);
      }
      if (lane == 0) arr[H] = mid;                               // q = 0, j = NS / 2: bin FD / 2
    }
    wave_sync();
    dc_correction_margin<H, kBM>(arr, cf0, fs, FD, lane);  // d4c.cpp:139
#pragma unroll
    for (int m = 0; m < M; ++m) sc_[m] = arr[lane + 64 * m];
    sc_[M] = arr[H];
    wave_sync();
    WM_PHASE_MARK(2)                                                                  // DC correction of the centroid

    // ---- GetSmoothedPowerSpectrum (d4c.cpp:148-164) ----
    lane = opaque_lane(lane);
    tw.fence();
    {
      cpx va[MS], none[MS];
      const FrameGeom fg = frame_geom(fs, cf0, pos, 4.0);
      frame_packed<kHann, false, MS>(xu, xl, fg, rtab, roff + 2 * Lw, lane, va);
      WM_PHASE_MARK(3)                                                                // Hann frame
#pragma unroll
      for (int m = 0; m < MS; ++m) none[m] = make_double2(0.0, 0.0);
      double pe[MS + 1], po[MS];
      real_power_halves<FD>(va, none, false, (fg.L + 127) >> 7, img, tw, lane, pe, po);   // L <= FD / 2: never folds
      cpx* arr2 = reinterpret_cast<cpx*>(arr);
#pragma unroll
      for (int m = 0; m < MS; ++m) arr2[lane + 64 * m] = make_double2(pe[m], po[m]);   // bins 2 j, 2 j + 1
      if (lane == 0) arr[H] = pe[MS];
      wave_sync();
    }
    WM_PHASE_MARK(4)                                                                  // its transform and power
    dc_correction_margin<H, kBM>(arr, cf0, fs, FD, lane);
    linear_smoothing_margin<H, kBM>(arr, cf0, fs, FD, lane);
    WM_PHASE_MARK(5)                                                                  // DC correction + smoothing
    // ---- GetStaticGroupDelay (d4c.cpp:170-186) ----
#pragma unroll
    for (int m = 0; m < M; ++m) arr[lane + 64 * m] = sc_[m] / arr[lane + 64 * m];
    if (lane == 0) arr[H] = sc_[M] / arr[H];
    wave_sync();
    linear_smoothing_margin<H, kBM>(arr, cf0 / 2.0, fs, FD, lane);
    double gd[MB];
#pragma unroll
    for (int m = 0; m < M; ++m) gd[m] = arr[lane + 64 * m];
    gd[M] = arr[H];
    wave_sync();
    linear_smoothing_margin<H, kBM>(arr, cf0, fs, FD, lane);
#pragma unroll
    for (int m = 0; m < M; ++m) gd[m] -= arr[lane + 64 * m];
    gd[M] -= arr[H];
    wave_sync();
    WM_PHASE_MARK(6)                                                                  // group delay: two smoothings

    // ---- GetCoarseAperiodicity (d4c.cpp:192-223) ----
    const int wl = tab.window_length;
    const int bnd = matlab_round(FD * 8.0 / wl);
    const int hwl = wl / 2;
    double coarse[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const int nap = ONE_BAND ? 1 : tab.nap;
#pragma unroll 1
    for (int band = 0; band < nap; ++band) {
#pragma unroll
      for (int m = 0; m < M; ++m) arr[lane + 64 * m] = gd[m];
      if (lane == 0) arr[H] = gd[M];
      wave_sync();
      const int center = (int)(kFreqInterval * (band + 1) * FD / fs);
      // The slice of the group delay under the Nuttall window (at most FD / 2 taps: the packed operand of NS pairs),
      // all loads issued together with clamped indices and the predicate on the value (see d4c_kernel)
      cpx vp[MS], none[MS];
      const int lw = opaque_lane(lane);
      tw.fence();
      {
        double na[MS], nb[MS], ga[MS], gb[MS];
#pragma unroll
        for (int m = 0; m < MS; ++m) {
          const int i0 = 2 * (lw + 64 * m);
          const int j0 = imin(i0, wl - 1), j1 = imin(i0 + 1, wl - 1);
          na[m] = tab.nuttall[j0];
          nb[m] = tab.nuttall[j1];
          ga[m] = arr[center - hwl + j0];
          gb[m] = arr[center - hwl + j1];
        }
#pragma unroll
        for (int m = 0; m < MS; ++m) {
          const int i0 = 2 * (lw + 64 * m);
          vp[m] = make_double2(i0 < wl ? ga[m] * na[m] : 0.0, i0 + 1 < wl ? gb[m] * nb[m] : 0.0);
          none[m] = make_double2(0.0, 0.0);
        }
      }
      WM_PHASE_MARK(7)                                                                // band window
      double pe[MS + 1], po[MS];
      real_power_halves<FD>(vp, none, false, (wl + 127) >> 7, img, tw, lw, pe, po);
      WM_PHASE_MARK(8)                                                                // band transform
      // through LDS into strided order (p[t] = bin lane + 64 t): the main lobe the peel removes is a run of
      // neighbouring bins, which then sit in different lanes and go in one or two steps of peel_largest()
      double p[MB];
      double tot = 0.0;
      {
        cpx* flat2 = reinterpret_cast<cpx*>(smem);
#pragma unroll
        for (int m = 0; m < MS; ++m) flat2[lw + 64 * m] = make_double2(pe[m], po[m]);     // bins 2 j, 2 j + 1
        wave_sync();
#pragma unroll
        for (int t = 0; t < M; ++t) {
          p[t] = smem[lw + 64 * t];
          tot += p[t];
        }
      }
      p[M] = -1.0;
      if (lw == 0) {
        p[M] = pe[MS];
        tot += pe[MS];
      }
      tot = wave_sum(tot);
      // Sum of all but the (bnd + 1) largest bins (d4c.cpp:215-220): see d4c_kernel
      wave_sync();
      sort_desc<M>(p);
#pragma unroll
      for (int i = M - 1; i >= 0; --i) {                    // Nyquist bin (lane 0 only, -1 elsewhere) into place
        const double hi = fmax(p[i], p[i + 1]), lo = fmin(p[i], p[i + 1]);
        p[i] = hi;
        p[i + 1] = lo;
      }
      double* heads = smem;                                   // [MB + 3][64]
#pragma unroll
      for (int m = 0; m < MB; ++m) heads[m * 64 + lw] = p[m];
#pragma unroll
      for (int m = MB; m < MB + 3; ++m) heads[m * 64 + lw] = -1.0;     // exhausted
      const int taken = peel_largest(heads, bnd + 1, lw);   // own column only: no barrier needed
      double low = 0.0;
#pragma unroll
      for (int m = 0; m < MB; ++m) low += (m >= taken && p[m] >= 0.0) ? p[m] : 0.0;
      low = wave_sum(low);
      double c = wm_log(low / tot) * 4.3429448190325182765;   // 10 log10(.)
      c = c + (cf0 - 100.0) / 50.0;                         // d4c.cpp:309-311
      c = 0.0 < c ? 0.0 : c;                                // MyMinDouble(0.0, c), common.h:80: a NaN stays a NaN
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (j == band) coarse[j] = c;               // static indices keep coarse[] in registers
      wave_sync();
      WM_PHASE_MARK(9)                                                                // sort + peel + log
    }

    // ---- GetAperiodicity (d4c.cpp:325-333): see d4c_kernel ----
    if (lane <= nap + 1) {
      double kv = lane == 0 ? -60.0 : -kSafe;
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (lane == j + 1 && j < nap) kv = coarse[j];
      smem[lane] = kv;
    }
    wave_sync();
    d4c_write_row([&](int k) { return smem[k]; }, nap, fs, out_fft, out_bins, lane, row);
    wave_sync();
    WM_PHASE_MARK(10)                                                                 // output row
  }
  WM_PHASE_FLUSH(0)
}

}  // namespace wm

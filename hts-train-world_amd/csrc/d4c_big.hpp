// d4c_big.hpp -- D4CGeneralBody for fft_size_d4c = 4096 (fs above 24 kHz: the recipe's own 48 kHz) and 8192 (fs above
// 48.1 kHz: 88.2 and 96 kHz), as four kernels on the quarter-size wavefront transform (1024 points, two waves per SIMD;
// 2048 points at 8192, one wave per SIMD).  Included by d4c.hip.
//
// One wavefront per frame with a 2048-point complex engine (32 complex values per lane and operand) needs all 512
// registers of a SIMD, i.e. one wave per SIMD with nothing to hide latency behind, and still keeps two 2049-bin
// spectra in registers across five band analyses: 231 us per voiced frame against 29 us at fft 2048 for ~3.7 x the
// arithmetic.  Here every transform of 4096 points is decomposed by hand into transforms of NS = 1024 complex
// points -- the size the engine runs at two waves per SIMD -- and the frame's work is cut into kernels whose state
// fits that budget, with the 2049-bin arrays passing through HBM (16 KB per frame and array):
//
//   d4cb_centroid_kernel   GetStaticCentroid (d4c.cpp:125-142).  z = s x + j (i + 1) x as in d4c.hip; the 4096-point
//                          complex transform is one radix-4 decimation-in-frequency step: Z[4 j + q] = FFT_NS(u_q)[j],
//                          u_q[n] = (sum_p (-j)^(p q) z[n + p NS]) W_4096^(n q), p over the quarters the window
//                          reaches into (typically one or two).  The pairing Z[k] Z[4096 - k] couples q = 0 and
//                          q = 2 with themselves and q = 1 with q = 3.  The frame is built in registers, feeds u_0
//                          from there and is re-read for u_2, u_1, u_3 from a scratch row of the workgroup (every
//                          lane what it stored itself); E1 stays in registers while E3 is transformed.
//                          Output: C[frame][q][j], the centroid at bin 4 j + q (both sides summed).
//   d4cb_spectrum_kernel   GetSmoothedPowerSpectrum + GetStaticGroupDelay (:148-186).  The real transform of the
//                          Hann frame: even bins are the real transform of 2048 points as it is (rfft_forward<NS>),
//                          odd bins come from FFT_NS(v[n] W_2048^n) with the split pairing j <-> NS - 1 - j.
//                          Output: GD[frame][0..2048].
//   d4cb_band_kernel       GetCoarseAperiodicity (:192-223), one wavefront per (frame, band): the same even / odd
//                          real transform of the windowed slice, the power spectrum's 2049 values sorted per lane,
//                          the largest peeled (peel_largest, peel.hpp).  Output: COARSE[frame][band].
//   d4cb_output_kernel     GetAperiodicity (:325-333) for the listed frames, the default row for all others but the
//                          RARE launch's.
//
// What these kernels were bound by, in the order it was found (MI355X, 256 utterances at 48 kHz, 272 794 frames;
// rocprofv3 SQ_WAIT_ANY / SQ_ACTIVE_INST_ANY per wave): memory round trips, not arithmetic.  A load that sits in a
// wave-uniform `if` next to its use is a dependent trip to memory per element (48 per sub-transform in the first
// version); values derived from the lane number or from a shared twiddle base are common subexpressions of all
// four sub-transforms and were kept alive across them (spilled); a parked spectrum is two more trips.  With
// loads issued in groups by window class, per-quarter fences on the lane and the twiddle base, and nothing
// parked, centroid / spectrum / band take 8.2 / 6.8 / 8.8 ms (first version 12.9 / 8.7 / 13.5) and issue
// instructions 85-100 % of the time at two waves per SIMD.
//
// Frames whose smoothing mirror exceeds 4096 / 16 bins (f0 >= fs / 16, d4c_is_usual) stay with the one-kernel form
// (d4c_kernel<4096, 1, true>); any window length up to 4096 samples is handled here.
#pragma once

namespace wm {

template <int FD> struct D4cBig {
  static constexpr int NS = FD / 4, MS = NS / 64, H = FD / 2;
  static constexpr int kQ = NS / 2 + 1;                 // entries of C per quarter (q = 0 uses all, the others NS / 2)
  static constexpr int kRow = ((H + 1 + 7) / 8) * 8;    // doubles per frame of the HBM arrays
};

// partner of lane-local element j = lane + 64 m in an NS-point spectrum held as v[m]: stores the upper half of
// the registers in plain layout (as d4c_centroid)
template <int NS>
__device__ __forceinline__ void store_upper(const cpx (&v)[NS / 64], cpx* img, int lane) {
  constexpr int MS = NS / 64;
  wave_sync();
#pragma unroll
  for (int m = MS / 2; m < MS; ++m) img[lane + 64 * m] = v[m];
  wave_sync();
}
template <int NS>
__device__ __forceinline__ void store_all(const cpx (&v)[NS / 64], cpx* img, int lane) {
  wave_sync();
#pragma unroll
  for (int m = 0; m < NS / 64; ++m) img[lane + 64 * m] = v[m];
  wave_sync();
}

// (-j)^k z
template <int K> __device__ __forceinline__ cpx mul_mj_pow(cpx z) {
  if (K == 0) return z;
  if (K == 1) return make_double2(z.y, -z.x);
  if (K == 2) return make_double2(-z.x, -z.y);
  return make_double2(-z.y, z.x);
}

// u_q[n] = (sum_p z[n + p NS] (-j)^(p q)) W_FD^(n q) for n = lane + 64 m, from the frame parked in the workgroup's
// scratch row (xs[i] = sample i of the normalised-later frame): z[i] = xs[i] (s + j (i + 1)).  P = the number of
// quarters the window reaches into (ceil(L / NS)), a template argument so that the loads of a group of elements
// are issued together: with a branch per quarter inside the element loop every load sits in its own block next
// to its use, and an element costs up to three dependent trips to memory (48 per sub-transform).
template <int FD, int Q, int P>
__device__ __forceinline__ void d4cb_quarter_input_p(const double* xs, double s, cpx wl, int lane,
                                                     cpx (&v)[D4cBig<FD>::MS]) {
  constexpr int NS = D4cBig<FD>::NS, MS = D4cBig<FD>::MS;
  constexpr int G = P == 1 ? MS : (P == 2 ? MS / 2 : MS / 4);     // elements per group: 12 to 16 loads in flight
  const cpx w64 = FD <= 4096 ? cis64(FD <= 4096 ? 4096 / FD : 1) : cis_neg2pi(64.0 / (double)FD);   // W_FD^64
  cpx w = wl;
#pragma unroll
  for (int g0 = 0; g0 < MS; g0 += G) {
    double xv[P][G];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int m = 0; m < G; ++m) xv[p][m] = xs[lane + 64 * (g0 + m) + p * NS];
#pragma unroll
    for (int m = 0; m < G; ++m) {
      const double r = (double)(lane + 64 * (g0 + m) + 1);
      cpx acc = make_double2(s * xv[0][m], r * xv[0][m]);
      if (P > 1) acc = cadd(acc, mul_mj_pow<(1 * Q) & 3>(make_double2(s * xv[P > 1 ? 1 : 0][m], (r + NS) * xv[P > 1 ? 1 : 0][m])));
      if (P > 2) acc = cadd(acc, mul_mj_pow<(2 * Q) & 3>(make_double2(s * xv[P > 2 ? 2 : 0][m], (r + 2 * NS) * xv[P > 2 ? 2 : 0][m])));
      if (P > 3) acc = cadd(acc, mul_mj_pow<(3 * Q) & 3>(make_double2(s * xv[P > 3 ? 3 : 0][m], (r + 3 * NS) * xv[P > 3 ? 3 : 0][m])));
      if (Q == 0) v[g0 + m] = acc;
      if (Q == 1) v[g0 + m] = cmul(acc, w);
      if (Q == 2) v[g0 + m] = cmul(acc, cmul(w, w));
      if (Q == 3) v[g0 + m] = cmul(acc, cmul(cmul(w, w), w));
      w = cmul(w, w64);
    }
  }
}

template <int FD, int Q>
__device__ __forceinline__ void d4cb_quarter_input(const double* xs, int L, double s, cpx wl, int lane,
                                                   cpx (&v)[D4cBig<FD>::MS]) {
  constexpr int NS = D4cBig<FD>::NS;
  // fenced: the frame is read from memory again (not forwarded from the registers that were stored, which would
  // have to live through the transforms in between), and the sample numbers and the chain of twiddles are
  // rebuilt per quarter
  asm volatile("" ::: "memory");
  lane = opaque_lane(lane);
  wl = make_double2(opaque_d(wl.x), opaque_d(wl.y));
  if (L <= NS) d4cb_quarter_input_p<FD, Q, 1>(xs, s, wl, lane, v);           // wave-uniform
  else if (L <= 2 * NS) d4cb_quarter_input_p<FD, Q, 2>(xs, s, wl, lane, v);
  else if (L <= 3 * NS) d4cb_quarter_input_p<FD, Q, 3>(xs, s, wl, lane, v);
  else d4cb_quarter_input_p<FD, Q, 4>(xs, s, wl, lane, v);
}

// u_0 from the frame still in registers (x[q] = sample lane + 64 q): no twiddles, and no trip to memory before the
// first transform.
template <int FD>
__device__ __forceinline__ void d4cb_quarter0_input(const double (&x)[FD / 64], int L, double s, int lane,
                                                    cpx (&v)[D4cBig<FD>::MS]) {
  constexpr int NS = D4cBig<FD>::NS, MS = D4cBig<FD>::MS;
#pragma unroll
  for (int m = 0; m < MS; ++m) {
    const double r = (double)(lane + 64 * m + 1);
    cpx acc = make_double2(s * x[m], r * x[m]);
    if (NS < L) acc = cadd(acc, make_double2(s * x[m + MS], (r + NS) * x[m + MS]));
    if (2 * NS < L) acc = cadd(acc, make_double2(s * x[m + 2 * MS], (r + 2 * NS) * x[m + 2 * MS]));
    if (3 * NS < L) acc = cadd(acc, make_double2(s * x[m + 3 * MS], (r + 3 * NS) * x[m + 3 * MS]));
    v[m] = acc;
  }
}

// Scratch of a workgroup of the centroid kernel (doubles): the frames of the two sides.
template <int FD> struct D4cBigWs { static constexpr int kDoubles = 2 * FD; };

// The listed frames [begin, begin + chunk) of a launch: the per-frame arrays C / GD have `chunk` rows, indexed by
// the position in that range.
__device__ __forceinline__ int d4cb_chunk_count(const int* __restrict__ n_listed, int begin, int chunk) {
  return imax(0, imin(chunk, *n_listed - begin));
}

template <int FD>
__global__ __launch_bounds__(64, FD > 4096 ? 1 : 2) void d4cb_centroid_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int* __restrict__ frame_utt, const double* __restrict__ tpos, const double* __restrict__ f0,
    const int* __restrict__ rng_off, const uint32_t* __restrict__ rtab, int fs_arg, const int* __restrict__ perm,
    const int* __restrict__ n_listed, int begin, int chunk, double* scratch, double* __restrict__ C) {
  constexpr int NS = D4cBig<FD>::NS, MS = D4cBig<FD>::MS, kQ = D4cBig<FD>::kQ;
  __shared__ __attribute__((aligned(16))) double smem[2 * FftLds<NS>::kElems];
  cpx* img = reinterpret_cast<cpx*>(smem);
  const int lane0 = threadIdx.x;
  FftTw<NS> tw;
  tw.init(lane0);
  double* ws = scratch + (int64_t)blockIdx.x * D4cBigWs<FD>::kDoubles;
  const int n_run = d4cb_chunk_count(n_listed, begin, chunk);
  FramePipe pipe;
  pipe.init(perm + begin, n_run, x, x_off, x_len, frame_utt, tpos, f0, rng_off);
  WM_FOR_EACH_PIPED(sc, pipe, n_run) {
    const int fs = opaque_uniform(fs_arg);
    const double cf0 = uniform_d(sc.f0 > kFloorF0D4C ? sc.f0 : kFloorF0D4C);
    const double pos = uniform_d(sc.tpos);
    const int Lw = 2 * matlab_round(2.0 * fs / cf0) + 1;
    double* Cf = C + sc.pos * (int64_t)(4 * kQ);
    // ---- the frames of both sides (d4c.cpp:125-142: pos -/+ 0.25 / f0), built in registers and parked in the
    //      workgroup's scratch rows: every sub-transform re-reads them from there (every lane what it stored itself)
    //      instead of 128 registers living through the transforms ----
    const FrameGeom fg0 = frame_geom(fs, cf0, uniform_d(pos - 0.25 / cf0), 4.0);
    const int L = fg0.L;                                          // the same on both sides
    const double s = (double)(1 << (31 - __clz(fg0.hw | 1)));
    double scale[2];
#pragma unroll 1
    for (int side = 0; side < 2; ++side) {
      const int lane = opaque_lane(lane0);
      const double cpos = uniform_d(side == 0 ? pos - 0.25 / cf0 : pos + 0.25 / cf0);
      const FrameGeom fg = frame_geom(fs, cf0, cpos, 4.0);
      double xr[4 * MS];
      double pwr;
      frame_strided<kBlackman, 4 * MS, false>(sc.xu, sc.xlen, fg, rtab, sc.roff + side * Lw, lane, xr, pwr);
      double* xs = ws + side * FD;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (g * NS < fg.L) {                                      // wave-uniform
#pragma unroll
          for (int m = 0; m < MS; ++m) xs[lane + 64 * (m + MS * g)] = xr[m + MS * g];
        }
      // normalisation to unit energy (d4c.cpp:96-100) and the 1 / (2 s) of the identity in d4c.hip, on the products
      const double sv = uniform_d(1.0 / (2.0 * s * pwr));
      if (side == 0) scale[0] = sv; else scale[1] = sv;
    }
    // ---- C[q][j] = sum over the sides of the centroid at bin 4 j + q, accumulated in registers and stored once
    //      (the first version added the second side to what the first had stored: a read-modify-write of 16 KB per
    //      frame through HBM) ----
    // q = 0: pairs with itself, j <-> (NS - j) mod NS
    {
      double acc[MS / 2], mid = 0.0;
#pragma unroll
      for (int m = 0; m < MS / 2; ++m) acc[m] = 0.0;
#pragma unroll 1
      for (int side = 0; side < 2; ++side) {
        const int lane = opaque_lane(lane0);
        tw.fence();
        const double sv = side == 0 ? scale[0] : scale[1];
        cpx v[MS];
        d4cb_quarter_input<FD, 0>(ws + side * FD, L, s, make_double2(1.0, 0.0), lane, v);
        fft_forward<NS>(v, img, tw, lane);
        store_upper<NS>(v, img, lane);
#pragma unroll
        for (int m = 0; m < MS / 2; ++m) {
          const int j = lane + 64 * m;
          cpx pt = img[(NS - j) & (NS - 1)];
          if (m == 0) {
            pt.x = lane == 0 ? v[0].x : pt.x;
            pt.y = lane == 0 ? v[0].y : pt.y;
          }
          acc[m] += (v[m].x * pt.y + v[m].y * pt.x) * sv;
        }
        mid += 2.0 * v[MS / 2].x * v[MS / 2].y * sv;               // lane 0: j = NS / 2 pairs with itself
        wave_sync();
      }
#pragma unroll
      for (int m = 0; m < MS / 2; ++m) Cf[0 * kQ + lane0 + 64 * m] = acc[m];
      if (lane0 == 0) Cf[NS / 2] = mid;
    }
    // q = 2: pairs with itself, j <-> NS - 1 - j
    {
      double acc[MS / 2];
#pragma unroll
      for (int m = 0; m < MS / 2; ++m) acc[m] = 0.0;
#pragma unroll 1
      for (int side = 0; side < 2; ++side) {
        const int lane = opaque_lane(lane0);
        tw.fence();
        const double sv = side == 0 ? scale[0] : scale[1];
        const cpx wl = cis_neg2pi((double)lane / (double)FD);     // W_FD^lane
        cpx v[MS];
        d4cb_quarter_input<FD, 2>(ws + side * FD, L, s, wl, lane, v);
        fft_forward<NS>(v, img, tw, lane);
        store_upper<NS>(v, img, lane);
#pragma unroll
        for (int m = 0; m < MS / 2; ++m) {
          const cpx pt = img[NS - 1 - (lane + 64 * m)];
          acc[m] += (v[m].x * pt.y + v[m].y * pt.x) * sv;
        }
        wave_sync();
      }
#pragma unroll
      for (int m = 0; m < MS / 2; ++m) Cf[2 * kQ + lane0 + 64 * m] = acc[m];
    }
    // q = 1 and q = 3: bins 4 j + 1 pair E1[j] with E3[NS - 1 - j], bins 4 j + 3 pair E3[j] with E1[NS - 1 - j];
    // E1 waits in registers while E3 is transformed
    {
      double acc1[MS / 2], acc3[MS / 2];
#pragma unroll
      for (int m = 0; m < MS / 2; ++m) acc1[m] = acc3[m] = 0.0;
#pragma unroll 1
      for (int side = 0; side < 2; ++side) {
        const int lane = opaque_lane(lane0);
        tw.fence();
        const double sv = side == 0 ? scale[0] : scale[1];
        const cpx wl = cis_neg2pi((double)lane / (double)FD);
        cpx v[MS];
        d4cb_quarter_input<FD, 1>(ws + side * FD, L, s, wl, lane, v);
        fft_forward<NS>(v, img, tw, lane);
        cpx v1[MS];
#pragma unroll
        for (int m = 0; m < MS; ++m) v1[m] = v[m];
        d4cb_quarter_input<FD, 3>(ws + side * FD, L, s, wl, lane, v);
        fft_forward<NS>(v, img, tw, lane);
        store_upper<NS>(v, img, lane);
#pragma unroll
        for (int m = 0; m < MS / 2; ++m) {
          const cpx pt = img[NS - 1 - (lane + 64 * m)];           // E3[NS - 1 - j]
          acc1[m] += (v1[m].x * pt.y + v1[m].y * pt.x) * sv;
        }
        store_upper<NS>(v1, img, lane);
#pragma unroll
        for (int m = 0; m < MS / 2; ++m) {
          const cpx pt = img[NS - 1 - (lane + 64 * m)];           // E1[NS - 1 - j]
          acc3[m] += (v[m].x * pt.y + v[m].y * pt.x) * sv;
        }
        wave_sync();
      }
#pragma unroll
      for (int m = 0; m < MS / 2; ++m) {
        Cf[1 * kQ + lane0 + 64 * m] = acc1[m];
        Cf[3 * kQ + lane0 + 64 * m] = acc3[m];
      }
    }
  }
}

// Power spectrum |X[k]|^2, k = 0 .. FD / 2, of a real frame of FD samples given as packed pairs in two halves:
// va[m] = (x[2 n], x[2 n + 1]), vb[m] the same NS pairs later (n = lane + 64 m < NS); `folded` says whether vb holds
// anything (wave-uniform).  pe[m] = power at bin 2 (lane + 64 m), pe[MS] = power at bin FD / 2 (every lane),
// po[m] = power at bin 2 (lane + 64 m) + 1.  nz: only va[0 .. nz) may be non-zero (the window's reach; MS when folded):
// both transforms take the pruned first pass (fft.hpp).  The operand of the odd bins waits in registers while the even bins are
// transformed (64 registers; parking it in memory cost 32 KB of traffic per call and a trip there and back).
// va / vb are consumed.
template <int FD>
__device__ __forceinline__ void real_power_halves(cpx (&va)[D4cBig<FD>::MS], cpx (&vb)[D4cBig<FD>::MS], bool folded,
                                                  int nz, cpx* img, const FftTw<D4cBig<FD>::NS>& tw, int lane,
                                                  double (&pe)[D4cBig<FD>::MS + 1], double (&po)[D4cBig<FD>::MS]) {
  constexpr int NS = D4cBig<FD>::NS, MS = D4cBig<FD>::MS;
  // the packed sequence vp of 2 NS points splits into even bins FFT_NS(va + vb) and odd bins
  // FFT_NS((va - vb) W_{2 NS}^n)
  cpx odd[MS];
  {
    cpx w = tw.wsplit;                                            // W_{2 NS}^lane
#pragma unroll
    for (int m = 0; m < MS; ++m) {
      const cpx d = folded ? csub(va[m], vb[m]) : va[m];
      odd[m] = cmul(d, w);
      if (folded) va[m] = cadd(va[m], vb[m]);
      w = cmul(w, tw.wstep());
    }
  }
  // even bins: the real transform of 2 NS points as it is
  rfft_forward_nz<NS>(va, img, img, tw, lane, nz);
#pragma unroll
  for (int m = 0; m < MS; ++m) {
    const cpx s = img[lane + 64 * m];
    pe[m] = s.x * s.x + s.y * s.y;
  }
  {
    const cpx s = img[NS];
    pe[MS] = s.x * s.x + s.y * s.y;
  }
  // odd bins: X[2 j + 1] from O[j] and O[NS - 1 - j] with the twiddle W_FD^(2 j + 1)
  fft_forward_nz<NS>(odd, img, tw, lane, nz);
  store_all<NS>(odd, img, lane);
  {
    const cpx w1 = cis_neg2pi(1.0 / (double)FD);                  // W_FD
    cpx w = cmul(tw.wsplit, w1);                                  // W_FD^(2 lane + 1)
#pragma unroll
    for (int m = 0; m < MS; ++m) {
      const cpx a = odd[m];
      const cpx b = cconj(img[NS - 1 - (lane + 64 * m)]);
      const cpx e = make_double2(0.5 * (a.x + b.x), 0.5 * (a.y + b.y));
      const cpx d = csub(a, b);
      const cpx o = make_double2(0.5 * d.y, -0.5 * d.x);          // (a - b) / (2 i)
      const cpx xk = cadd(e, cmul(w, o));
      po[m] = xk.x * xk.x + xk.y * xk.y;
      w = cmul(w, tw.wstep());                                    // W_FD^128 = W_{2 NS}^64
    }
  }
  wave_sync();
}

// The same power spectrum with every transform's output taken BY PAIRS (fft.hpp, rfft_split_pairs): the lane that holds
// element j of a transform also receives its partner (NS - j for the even bins' split, NS - 1 - j for the odd bins'), and
// both bins of a pair are the same sums and differences -- for the odd bins X[2 j + 1] = e + w o and
// X[2 (NS - 1 - j) + 1] = conj(e - w o) -- so only the upper half of each transform's output goes through LDS, half the
// twiddles are built, and no spectrum is parked in LDS to be read back.  put(bin, value) is called for every bin
// 0 .. FD / 2 exactly once, from whichever lane holds it, AFTER the last read of the image (it may store into it).
template <int FD, class Put>
__device__ __forceinline__ void real_power_pairs(cpx (&va)[D4cBig<FD>::MS], cpx (&vb)[D4cBig<FD>::MS], bool folded,
                                                 int nz, cpx* img, const FftTw<D4cBig<FD>::NS>& tw, int lane, Put put) {
  constexpr int NS = D4cBig<FD>::NS, MS = D4cBig<FD>::MS, MH = MS / 2;
  cpx odd[MS];
  {
    cpx w = tw.wsplit;                                            // W_{2 NS}^lane
#pragma unroll
    for (int m = 0; m < MS; ++m) {
      const cpx d = folded ? csub(va[m], vb[m]) : va[m];
      odd[m] = cmul(d, w);
      if (folded) va[m] = cadd(va[m], vb[m]);
      w = cmul(w, tw.wstep());
    }
  }
  // even bins 2 j: the real transform of 2 NS points as it is; j = lane + 64 m and NS - j, and NS / 2 in lane 0
  double ek[MH], er[MH], eh = 0.0;
  fft_forward_nz<NS>(va, img, tw, lane, nz);
  rfft_split_pairs_f<NS>(va, img, tw, lane, [&](int m, cpx a, cpx b) {
    if (m < MH) {
      ek[m < MH ? m : 0] = a.x * a.x + a.y * a.y;
      er[m < MH ? m : 0] = b.x * b.x + b.y * b.y;
    } else {
      eh = a.x * a.x + a.y * a.y;
    }
  });
  // odd bins: X[2 j + 1] from O[j] and O[NS - 1 - j] with the twiddle W_FD^(2 j + 1)
  double ok[MH], orr[MH];
  fft_forward_nz<NS>(odd, img, tw, lane, nz);
  store_upper<NS>(odd, img, lane);
  {
    const cpx w1 = cis_neg2pi(1.0 / (double)FD);                  // W_FD
    cpx w = cmul(tw.wsplit, w1);                                  // W_FD^(2 lane + 1)
#pragma unroll
    for (int m = 0; m < MH; ++m) {
      const cpx a = odd[m];
      const cpx b = cconj(img[NS - 1 - (lane + 64 * m)]);
      const cpx e = make_double2(0.5 * (a.x + b.x), 0.5 * (a.y + b.y));
      const cpx d = csub(a, b);
      const cpx o = make_double2(0.5 * d.y, -0.5 * d.x);          // (a - b) / (2 i)
      const cpx wo = cmul(w, o);
      const cpx xp = cadd(e, wo), xm = csub(e, wo);
      ok[m] = xp.x * xp.x + xp.y * xp.y;
      orr[m] = xm.x * xm.x + xm.y * xm.y;
      w = cmul(w, tw.wstep());                                    // W_FD^128 = W_{2 NS}^64
    }
  }
  wave_sync();                                                    // every partner is read: the image may be written
#pragma unroll
  for (int m = 0; m < MH; ++m) {
    const int j = lane + 64 * m;
    put(2 * j, ek[m]);
    put(2 * j + 1, ok[m]);
    put(2 * NS - 2 * j - 1, orr[m]);
    put(2 * NS - 2 * j, er[m]);
  }
  if (lane == 0) put(NS, eh);
}

template <int FD>
__global__ __launch_bounds__(64, FD > 4096 ? 1 : 2) void d4cb_spectrum_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int* __restrict__ frame_utt, const double* __restrict__ tpos, const double* __restrict__ f0,
    const int* __restrict__ rng_off, const uint32_t* __restrict__ rtab, int fs_arg, const int* __restrict__ perm,
    const int* __restrict__ n_listed, int begin, int chunk, const double* __restrict__ C, double* __restrict__ GD) {
  constexpr int NS = D4cBig<FD>::NS, MS = D4cBig<FD>::MS, H = D4cBig<FD>::H, kQ = D4cBig<FD>::kQ;
  constexpr int kRow = D4cBig<FD>::kRow;
  constexpr int kBM = FD / 16;
  constexpr int kImg = 2 * FftLds<NS>::kElems;
  constexpr int kRegion = SmoothCfg<H, kBM>::kRegion;
  constexpr int kTot = kImg > kRegion ? kImg : kRegion;
  constexpr int T = SmoothCfg<H, kBM>::kBi;                       // consecutive bins per lane of the smoothing's layout
  __shared__ __attribute__((aligned(16))) double smem[kTot];
  double* arr = smem + kBM;
  cpx* img = reinterpret_cast<cpx*>(smem);
  const int lane0 = threadIdx.x;
  FftTw<NS> tw;
  tw.init(lane0);
  const int n_run = d4cb_chunk_count(n_listed, begin, chunk);
  FramePipe pipe;
  pipe.init(perm + begin, n_run, x, x_off, x_len, frame_utt, tpos, f0, rng_off);
  WM_FOR_EACH_PIPED(sc, pipe, n_run) {
    const int lane = opaque_lane(lane0);
    const int fs = opaque_uniform(fs_arg);
    tw.fence();
    const double cf0 = uniform_d(sc.f0 > kFloorF0D4C ? sc.f0 : kFloorF0D4C);
    const int Lw = 2 * matlab_round(2.0 * fs / cf0) + 1;
    const double* Cf = C + sc.pos * (int64_t)(4 * kQ);
    double* GDf = GD + sc.pos * (int64_t)kRow;
    wave_sync();
    // ---- GetSmoothedPowerSpectrum (d4c.cpp:148-164), kept in registers (bin lane + 64 t): the static centroid
    //      passes through LDS next, and what used to be parked in HBM in between (the corrected centroid, the first
    //      smoothing of the group delay: four times 16 KB per frame) stays on the chip ----
    double ps[T];
    {
      cpx va[MS], vb[MS];
      const FrameGeom fg = frame_geom(fs, cf0, uniform_d(sc.tpos), 4.0);
      {
        cpx vp[2 * MS];
        frame_packed<kHann, false, 2 * MS>(sc.xu, sc.xlen, fg, rtab, sc.roff + 2 * Lw, lane, vp);
#pragma unroll
        for (int m = 0; m < MS; ++m) { va[m] = vp[m]; vb[m] = vp[m + MS]; }
      }
      real_power_pairs<FD>(va, vb, fg.L > 2 * NS, fg.L > 2 * NS ? MS : (fg.L + 127) >> 7, img, tw, lane,
                           [&](int bin, double v) { arr[bin] = v; });
      wave_sync();
    }
    dc_correction_margin<H, kBM>(arr, cf0, fs, FD, lane);
    // the smoothed power waits in the smoothing's own layout (BI consecutive bins per lane) for the division below
    linear_smoothing_margin<H, kBM>(arr, cf0, fs, FD, lane, [&](int q, double s) { return ps[q] = s; });
    // ---- static centroid: gather the four quarters, DCCorrection (d4c.cpp:139) ----
    {
      double cq[4][MS / 2];                                      // all 32 loads in flight, then the LDS stores
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int m = 0; m < MS / 2; ++m) cq[q][m] = Cf[q * kQ + lane + 64 * m];
      const double mid = Cf[NS / 2];
      cpx* arr2 = reinterpret_cast<cpx*>(arr);                   // bins 4 j .. 4 j + 3 as two 16-byte stores
#pragma unroll
      for (int m = 0; m < MS / 2; ++m) {
        arr2[2 * (lane + 64 * m)] = make_double2(cq[0][m], cq[1][m]);
        arr2[2 * (lane + 64 * m) + 1] = make_double2(cq[2][m], cq[3][m]);
      }
      if (lane == 0) arr[H] = mid;                               // q = 0, j = NS / 2: bin FD / 2
    }
    wave_sync();
    dc_correction_margin<H, kBM>(arr, cf0, fs, FD, lane);
    // ---- GetStaticGroupDelay (d4c.cpp:170-186): centroid / smoothed power, smoothed by f0 / 2 minus that by f0; the
    //      elementwise steps ride on the smoothings' stores, the last one straight to the frame's row in HBM ----
    {
      double cb[T];
#pragma unroll
      for (int q = 0; q < T; ++q) cb[q] = arr[lane * T + q];
      wave_sync();
#pragma unroll
      for (int q = 0; q < T; ++q) arr[lane * T + q] = cb[q] / ps[q];
      wave_sync();
    }
    linear_smoothing_margin<H, kBM>(arr, cf0 / 2.0, fs, FD, lane, [&](int q, double s) { return ps[q] = s; });
    linear_smoothing_margin<H, kBM>(arr, cf0, fs, FD, lane, [&](int q, double s) {
      const int i = lane * T + q;
      if (i <= H) GDf[i] = ps[q] - s;
      return s;
    });
  }
}

// One wavefront per (listed frame, band).
template <int FD>
__global__ __launch_bounds__(64, FD > 4096 ? 1 : 2) void d4cb_band_kernel(const double* __restrict__ f0, int fs, D4CTables tab,
                                                          const int* __restrict__ perm,
                                                          const int* __restrict__ n_listed, int begin, int chunk,
                                                          const double* __restrict__ GD,
                                                          double* __restrict__ COARSE) {
  constexpr int NS = D4cBig<FD>::NS, MS = D4cBig<FD>::MS, kRow = D4cBig<FD>::kRow;
  constexpr int NP = 2 * MS + 1;                                  // power values per lane
  constexpr int kImg = 2 * FftLds<NS>::kElems;
  constexpr int kHeads = (NP + 3) * 64;
  __shared__ __attribute__((aligned(16))) double smem[kImg > kHeads ? kImg : kHeads];
  cpx* img = reinterpret_cast<cpx*>(smem);
  const int lane0 = threadIdx.x;
  FftTw<NS> tw;
  tw.init(lane0);
  const int64_t n_task = (int64_t)d4cb_chunk_count(n_listed, begin, chunk) * tab.nap;
  const int wl = tab.window_length, hwl = wl / 2;
  const int bnd = matlab_round(FD * 8.0 / wl);
  for (int64_t task = blockIdx.x; task < n_task; task += gridDim.x) {
    const int lane = opaque_lane(lane0);
    tw.fence();
    const int64_t li = task / tab.nap;
    const int band = (int)(task - li * tab.nap);
    const int frame = __builtin_amdgcn_readfirstlane(perm[begin + li]);
    const double f0v = f0[frame];
    const double cf0 = uniform_d(f0v > kFloorF0D4C ? f0v : kFloorF0D4C);
    const double* gd = GD + li * (int64_t)kRow;                  // rows by position in the chunk
    const int center = (int)(kFreqInterval * (band + 1) * FD / fs);
    // the window has at most FD / 4 - 1 taps (launch_d4c_big checks): the upper half of the packed operand is zero.
    // All loads are issued together with clamped indices (a branch per pair would make every pair a trip to memory).
    cpx vp[MS];
    {
      double ga[MS / 2], gb[MS / 2], na[MS / 2], nb[MS / 2];
#pragma unroll
      for (int m = 0; m < MS / 2; ++m) {
        const int i0 = imin(2 * (lane + 64 * m), wl - 1), i1 = imin(2 * (lane + 64 * m) + 1, wl - 1);
        ga[m] = gd[center - hwl + i0];
        gb[m] = gd[center - hwl + i1];
        na[m] = tab.nuttall[i0];
        nb[m] = tab.nuttall[i1];
      }
#pragma unroll
      for (int m = 0; m < MS / 2; ++m) {
        const int i0 = 2 * (lane + 64 * m);
        vp[m] = make_double2(i0 < wl ? ga[m] * na[m] : 0.0, i0 + 1 < wl ? gb[m] * nb[m] : 0.0);
      }
#pragma unroll
      for (int m = MS / 2; m < MS; ++m) vp[m] = make_double2(0.0, 0.0);
    }
    cpx none[MS];
#pragma unroll
    for (int m = 0; m < MS; ++m) none[m] = make_double2(0.0, 0.0);
    // every bin's power by index into LDS, then in strided order (p[t] = bin lane + 64 t): the main lobe the peel
    // removes is a run of neighbouring bins, which then sit in different lanes and go in one or two steps of
    // peel_largest()
    real_power_pairs<FD>(vp, none, false, (wl + 127) >> 7, img, tw, lane,          // the window never folds
                         [&](int bin, double v) { smem[bin] = v; });
    wave_sync();
    double p[NP];
    double tot = 0.0;
#pragma unroll
    for (int t = 0; t < 2 * MS; ++t) {
      p[t] = smem[lane + 64 * t];
      tot += p[t];
    }
    p[2 * MS] = -1.0;
    if (lane == 0) {
      p[2 * MS] = smem[2 * MS * 64];                                                 // bin FD / 2
      tot += p[2 * MS];
    }
    tot = wave_sum(tot);
    // the (bnd + 1) largest of the FD / 2 + 1 values are left out (d4c.cpp:215-220): see d4c_kernel
    sort_desc<2 * MS>(p);
#pragma unroll
    for (int i = 2 * MS - 1; i >= 0; --i) {
      const double hi = fmax(p[i], p[i + 1]), lo = fmin(p[i], p[i + 1]);
      p[i] = hi;
      p[i + 1] = lo;
    }
    double* heads = smem;                                         // [NP + 3][64]
    wave_sync();
#pragma unroll
    for (int m = 0; m < NP; ++m) heads[m * 64 + lane] = p[m];
#pragma unroll
    for (int m = NP; m < NP + 3; ++m) heads[m * 64 + lane] = -1.0;
    const int taken = peel_largest(heads, bnd + 1, lane);
    double low = 0.0;
#pragma unroll
    for (int m = 0; m < NP; ++m) low += (m >= taken && p[m] >= 0.0) ? p[m] : 0.0;
    low = wave_sum(low);
    double c = wm_log(low / tot) * 4.3429448190325182765;   // 10 log10(.)
    c = c + (cf0 - 100.0) / 50.0;                                 // d4c.cpp:309-311
    c = 0.0 < c ? 0.0 : c;                                        // MyMinDouble(0.0, c)
    if (lane == 0) COARSE[(int64_t)frame * 8 + band] = c;
    wave_sync();
  }
}

// D4CLoveTrainSub (d4c.cpp:225-250) where its transform has 8192 points (fs above 54.6 kHz): the power spectrum by
// the same even / odd halves on the quarter-size engine; aperiodicity0 = cum[boundary1] / cum[boundary2] with the
// bins up to boundary0 left out.  One wavefront per listed (f0 != 0) frame; the others get 0 (d4c.cpp:231-233).
template <int FL>
__global__ __launch_bounds__(64, FL > 4096 ? 1 : 2) void d4cb_lovetrain_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int* __restrict__ frame_utt, const double* __restrict__ tpos, const double* __restrict__ f0,
    const int* __restrict__ rng_off, const uint32_t* __restrict__ rtab, int fs, int64_t total_frames,
    const int* __restrict__ perm, const int* __restrict__ n_listed, double* __restrict__ ap0) {
  constexpr int NS = D4cBig<FL>::NS, MS = D4cBig<FL>::MS;
  __shared__ __attribute__((aligned(16))) double smem[2 * FftLds<NS>::kElems];
  cpx* img = reinterpret_cast<cpx*>(smem);
  const int lane0 = threadIdx.x;
  FftTw<NS> tw;
  tw.init(lane0);
  const int b0 = (int)ceil(100.0 * FL / fs), b1 = (int)ceil(4000.0 * FL / fs), b2 = (int)ceil(7900.0 * FL / fs);
  const int n_run = *n_listed;
  for (int64_t i = n_run + blockIdx.x * 64 + lane0; i < total_frames; i += (int64_t)gridDim.x * 64)
    ap0[perm[i]] = 0.0;
  FramePipe pipe;
  pipe.init(perm, n_run, x, x_off, x_len, frame_utt, tpos, f0, rng_off);
  WM_FOR_EACH_PIPED(sc, pipe, n_run) {
    const int lane = opaque_lane(lane0);
    tw.fence();
    const double cf0 = uniform_d(sc.f0 > 40.0 ? sc.f0 : 40.0);
    const FrameGeom fg = frame_geom(fs, cf0, uniform_d(sc.tpos), 3.0);
    cpx va[MS], vb[MS];
    {
      cpx vp[2 * MS];
      frame_packed<kBlackman, false, 2 * MS>(sc.xu, sc.xlen, fg, rtab, sc.roff, lane, vp);
#pragma unroll
      for (int m = 0; m < MS; ++m) { va[m] = vp[m]; vb[m] = vp[m + MS]; }
    }
    double pe[MS + 1], po[MS];
    real_power_halves<FL>(va, vb, fg.L > 2 * NS, fg.L > 2 * NS ? MS : (fg.L + 127) >> 7, img, tw, lane, pe, po);
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int m = 0; m < MS; ++m) {
      const int ke = 2 * (lane + 64 * m), ko = ke + 1;
      if (ke > b0 && ke <= b1) s1 += pe[m];
      if (ke > b0 && ke <= b2) s2 += pe[m];
      if (ko > b0 && ko <= b1) s1 += po[m];
      if (ko > b0 && ko <= b2) s2 += po[m];
    }
    if (lane == 0) {                                              // bin FL / 2 only if a boundary reaches it
      if (FL / 2 <= b1) s1 += pe[MS];
      if (FL / 2 <= b2) s2 += pe[MS];
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) ap0[sc.frame] = s1 / s2;
    wave_sync();
  }
}

// ap rows: interpolation of the coarse values (GetAperiodicity, d4c.cpp:325-333) for the listed frames, the
// default 1 - 1e-12 for all others (:318-323).  One wavefront per frame, four per workgroup.
// `rare`: the frames of the RARE launch (d4c_kernel<FD, 1, true>, which may run before this kernel) keep what it
// wrote; rare.fd == 0 when there is no such launch (fft 8192: those frames get the default row).
__global__ __launch_bounds__(256) void d4cb_output_kernel(int fs, D4CTables tab, int out_fft, int64_t total_frames,
                                                          const int* __restrict__ perm,
                                                          const int* __restrict__ n_listed, D4cRunRarePred rare,
                                                          const double* __restrict__ COARSE, double* __restrict__ ap) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (int64_t)gridDim.x * 4;
  const int out_bins = out_fft / 2 + 1;
  const int n_run = *n_listed;
  for (int64_t k = wave; k < total_frames; k += waves) {
    const int frame = perm[k];
    double* row = ap + frame * (int64_t)out_bins;
    if (k >= n_run) {
      if (rare.fd != 0 && rare(frame)) continue;
      for (int i = lane; i < out_bins; i += 64) row[i] = 1.0 - kSafe;
      continue;
    }
    const double* cz = COARSE + (int64_t)frame * 8;
    d4c_write_row([&](int k) { return k == 0 ? -60.0 : (k > tab.nap ? -kSafe : cz[k - 1]); }, tab.nap, fs, out_fft, out_bins,
                  lane, row);
  }
}

}  // namespace wm

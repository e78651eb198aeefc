// synth_pulse_bp.hpp -- the pulse kernel of Synthesis in the pair layout (included by synthesis.hip).
#pragma once

namespace wm {

// Spectra live in registers BY PAIRS in this kernel (fft.hpp, rfft_split_pairs): a lane holds bins k = lane + 64 m and
// N - k for m < M / 2 (lane 0, m = 0: bins 0 and N = H), lane 0 also the middle bin N / 2.  Every transform hands its
// half spectrum to the next step in that layout, so a spectrum goes through LDS only where the algorithm itself
// re-arranges it (the cepstrum's fold into the packed operand of the next transform).
template <int MH, class T> struct BinPairs {
  T k[MH], r[MH], h;
};

// GetMinimumPhaseSpectrum (common.cpp:182-220) for one wavefront.  ls[0..H] (LDS) holds the log spectrum; on exit mp
// is the minimum-phase spectrum by pairs.  The cepstrum's imaginary parts (rounding noise of a real symmetric
// transform) are dropped, so the reference's c2c becomes a second r2c -- of a sequence that is zero above H, i.e. in
// all but the first M / 2 + 1 packed registers (pruned first pass).
template <int N>
__device__ __forceinline__ void minimum_phase_bp(const double* ls, cpx* img, const FftTw<N>& tw, int lane,
                                              BinPairs<N / 128, cpx>& mp) {
  constexpr int M = N / 64, MH = M / 2, F = 2 * N, H = N;
  lane = opaque_lane(lane);                    // the mirrored indices are rebuilt per call, not kept from the last one
  cpx v[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int i0 = 2 * (lane + 64 * m), i1 = i0 + 1;
    v[m] = make_double2(ls[i0 <= H ? i0 : F - i0], ls[i1 <= H ? i1 : F - i1]);   // mirroring :184-187
  }
  // folded cepstrum c[0] = C0, c[j] = 2 C[j] (0 < j < H), c[H] = C[H], 0 above (:193-203): real, [0 .. H] in LDS,
  // stored as the split produces it -- into the first N + 1 doubles of the image, i.e. the lower half of Z's plain
  // layout, which the split does not use (only the upper half of Z, complex elements above N / 2, travels)
  double* cf = reinterpret_cast<double*>(img);
  rfft_forward_pairs_f<N>(v, img, tw, lane, [&](int m, cpx a, cpx b) {
    if (m < MH) {
      const int k = lane + 64 * m;
      const bool ends = m == 0 && lane == 0;
      cf[k] = ends ? a.x : 2.0 * a.x;
      cf[N - k] = ends ? b.x : 2.0 * b.x;
    } else if (lane == 0) {
      cf[N / 2] = 2.0 * a.x;
    }
  });
  // packed for the next r2c: (c[2 n], c[2 n + 1]), n = lane + 64 m; n = N / 2 (lane 0) is (c[H], 0), zero beyond
#pragma unroll
  for (int m = 0; m < MH; ++m) v[m] = *reinterpret_cast<const cpx*>(cf + 2 * (lane + 64 * m));
  v[MH] = make_double2(lane == 0 ? cf[H] : 0.0, 0.0);
#pragma unroll
  for (int m = MH + 1; m < M; ++m) v[m] = make_double2(0.0, 0.0);
  auto polar = [&](cpx s) {
    const double amp = wm_exp(s.x / F);                              // :210-218
    double sn, cs;
    // the library's call here: in this loop it is 45 vector instructions; wm_sincospi is 30 plus 34 scalar moves for its
    // coefficients, and the scalar registers to keep those across the nine bins are not there (they come back as
    // v_readlane reloads): 4.86 ms against 4.40 for the kernel (tools/ab.sh).  A 64-point table in LDS with short
    // polynomials around it is 26 instructions, but the kernel then allocates 168 registers (three waves per SIMD)
    sincospi(s.y * (1.0 / (kPi * F)), &sn, &cs);                    // phase in half-turns: no Payne-Hanek path
    return make_double2(amp * cs, amp * sn);                        // (bin H: the split leaves Im = 0 exactly)
  };
  rfft_forward_pairs_f<N, MH + 1>(v, img, tw, lane, [&](int m, cpx a, cpx b) {
    if (m < MH) {
      mp.k[m < MH ? m : 0] = polar(a);
      __builtin_amdgcn_sched_barrier(0);                            // one bin at a time: keeps the VGPR peak low
      mp.r[m < MH ? m : 0] = polar(b);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      mp.h = polar(a);
    }
  });
}

// synth_pulse_kernel with every spectrum held BY PAIRS in registers (see above), for fft 2048 (48 kHz): at two waves per
// SIMD the CU's LDS pipe co-limits the kernel, and this form issues a fifth fewer ds_write_b128 per pulse.  At fft
// 1024 (four waves per SIMD, issue-bound) the same layout was 1 % slower and is not used
// (profiles/r05_z_pulse_pairs_experiment.txt).  Same arithmetic per bin; which lane holds a bin differs.
template <int F>
__global__ __launch_bounds__(64, F >= 4096 ? 1 : (F == 1024 ? 4 : (F < 1024 ? 3 : 2))) void synth_pulse_bp_kernel(
    const double* __restrict__ sp, const double* __restrict__ ap, const PulseRec* __restrict__ rec,
    const double* __restrict__ dcr, const uint32_t* __restrict__ rtab, int fs, double fp, int64_t p_begin,
    int64_t p_end, const int* __restrict__ perm, double* __restrict__ resp) {
  constexpr int N = F / 2, M = N / 64, MH = M / 2, H = F / 2;
  // LEAN (fft_size 2048: 16 complex values per lane and array): to run two waves per SIMD nothing of a spectrum's
  // size lives through a transform -- the interpolated envelope and aperiodicity are fetched again for the aperiodic
  // half instead of being kept (68 registers), the periodic response waits in the response row it is headed for
  // (32), and the log spectrum shares the LDS image of the transform that consumes it (8 KB: 9 workgroups per CU
  // instead of 6).  One wave per SIMD had nothing to hide the LDS round trips of its transforms behind.
  constexpr bool LEAN = F == 2048 || F == 1024;
  static_assert(F == 2048, "the pair layout pays at two waves per SIMD only (DESIGN.md section 3, item 46)");
  __shared__ __attribute__((aligned(16))) double smem[2 * FftLds<N>::kElems + (LEAN ? 0 : H + 2)];
  cpx* img = reinterpret_cast<cpx*>(smem);
  double* ls = LEAN ? smem : smem + 2 * FftLds<N>::kElems;
  const int lane0 = threadIdx.x;
  FftTw<N> tw;
  tw.init(lane0);

  // perm lists the chunk's voiced pulses (7 transforms) before its unvoiced ones (4): round-robin over the
  // list gives every wave the same number of each (partition.hpp)
  WM_FOR_EACH_LISTED(pi, perm, p_end - p_begin) {
    const int64_t p = p_begin + pi;
    const int lane = opaque_lane(lane0);
    const PulseRec r = rec[p];                                      // wave-uniform
    const int nf = r.nf;
    const int idx = r.idx;
    const int noise_size = r.noise_size;                            // synthesis.cpp:369
    const double cvuv = r.cvuv;
    const double ctime = idx / (double)fs;                          // pulse_locations = time_axis[i]
    const double shift = r.shift;

    // ---- GetSpectralEnvelope / GetAperiodicRatio (:140-178) ----
    const int ff = imin(nf - 1, (int)floor(ctime / fp));
    const int fc = imin(nf - 1, (int)ceil(ctime / fp));
    const double wgt = ff == fc ? 0.0 : ctime / fp - ff;    // beyond the last frame both indices are clamped: a copy there too
    const double* s0 = sp + (r.fbase + ff) * (int64_t)(H + 1);
    const double* s1 = sp + (r.fbase + fc) * (int64_t)(H + 1);
    const double* a0 = ap + (r.fbase + ff) * (int64_t)(H + 1);
    const double* a1 = ap + (r.fbase + fc) * (int64_t)(H + 1);
    // bins by pairs (BinPairs): k = lane + 64 m and H - k, m < MH, and the middle bin H / 2 (used from lane 0)
    auto spectral = [&](BinPairs<MH, double>& env, BinPairs<MH, double>& rat) {
      // synthesis.cpp:140-178 copies row ff when ff == fc and interpolates otherwise; with wgt = 0 (set above for
      // that case) the interpolation IS the copy (1 x + 0 y = x for finite y, and y is then the same row), so there
      // is one form and no branch inside the loop: behind one, every bin's loads were a trip to memory of their own
      auto bin = [&](int k, double& e, double& r) {
        e = (1.0 - wgt) * fabs(s0[k]) + wgt * fabs(s1[k]);
        const double a = (1.0 - wgt) * safe_ap(a0[k]) + wgt * safe_ap(a1[k]);
        r = a * a;
      };
#pragma unroll
      for (int m = 0; m < MH; ++m) {
        bin(lane + 64 * m, env.k[m], rat.k[m]);
        bin(H - (lane + 64 * m), env.r[m], rat.r[m]);
      }
      bin(H / 2, env.h, rat.h);
    };
    // f(value at k, value at H - k ...) over all of a lane's bins
    BinPairs<LEAN ? 1 : MH, double> env_keep, rat_keep;
    double rat0;
    if constexpr (LEAN) {
      const double a = (1.0 - wgt) * safe_ap(a0[0]) + wgt * safe_ap(a1[0]);   // bin 0, every lane
      rat0 = uniform_d(a * a);
    } else {
      spectral(env_keep, rat_keep);
      rat0 = __shfl(rat_keep.k[0], 0, 64);
    }
    double* out = resp + (p - p_begin) * (int64_t)F;

    // ---- GetPeriodicResponse (:105-138) ----
    double xp[LEAN ? 1 : M];            // periodic c2r output, x-index i = 2n + c for n < N/2 (first half)
    double dc = 0.0;
    const bool periodic = !(cvuv <= 0.5 || rat0 > 0.999);
#pragma unroll
    for (int m = 0; m < (LEAN ? 1 : M); ++m) xp[m] = 0.0;
    // A voiced pulse needs two minimum-phase spectra (periodic and aperiodic part): their phases come from ONE pair
    // of complex transforms (minimum_phase_pair), their amplitudes are the square roots of the spectra themselves
    if (periodic) {
      wave_sync();
      auto log_periodic = [&](const BinPairs<MH, double>& env, const BinPairs<MH, double>& rat) {
#pragma unroll
        for (int m = 0; m < MH; ++m) {
          ls[lane + 64 * m] = wm_log(env.k[m] * (1.0 - rat.k[m]) + kSafe) / 2.0;
          __builtin_amdgcn_sched_barrier(0);
          ls[H - (lane + 64 * m)] = wm_log(env.r[m] * (1.0 - rat.r[m]) + kSafe) / 2.0;
          __builtin_amdgcn_sched_barrier(0);
        }
        if (lane == 0) ls[H / 2] = wm_log(env.h * (1.0 - rat.h) + kSafe) / 2.0;
      };
      BinPairs<MH, cpx> mp;
      if constexpr (LEAN) {
        BinPairs<MH, double> env, rat;
        spectral(env, rat);
        log_periodic(env, rat);
      } else {
        log_periodic(env_keep, rat_keep);
      }
      wave_sync();
      minimum_phase_bp<N>(ls, img, tw, lane, mp);
      const double coef = 2.0 * kPi * shift * fs / F;               // :130-131
      // cos(coef k) for k = lane + 64 m by rotation from cos/sin(coef lane) in steps of 64 coef (the reference
      // evaluates cos per bin; the rotation is within 1e-15 of it); that of H - k from it and the angle of H; the
      // rotation's next step in lane 0 is the middle bin's
      double rc, rs, dc64, ds64;
      // in half-turns (coef / pi = 2 shift fs / F): the argument's own rounding, 1e-16 of up to 2000 half-turns,
      // moves a phase by 1e-12 rad at most -- the size of the rounding of coef * k itself
      const double ch = coef * (1.0 / kPi);
      double snH, reH;
      wm_sincospi(ch * lane, &rs, &rc);
      wm_sincospi(ch * 64.0, &ds64, &dc64);
      wm_sincospi(ch * H, &snH, &reH);
      // synthesis.cpp:96 takes the sine as sqrt(1 - cos^2): always >= 0.  The rotated cosine can pass 1 by a
      // rounding where the reference's cos() cannot: wm_sqrt returns 0 there instead of a NaN
      auto shifted = [&](cpx a, double re2) {                       // :88-100
        const double im2 = wm_sqrt(1.0 - re2 * re2);
        return make_double2(a.x * re2 + a.y * im2, a.y * re2 - a.x * im2);
      };
      cpx v[M];
      rfft_backward_pairs_f<N>([&](int m, cpx& a, cpx& b) {
        if (m < MH) {
          a = shifted(mp.k[m < MH ? m : 0], rc);
          b = shifted(mp.r[m < MH ? m : 0], (m == 0 && lane == 0) ? reH : reH * rc + snH * rs);
          const double nc = rc * dc64 - rs * ds64;
          rs = rs * dc64 + rc * ds64;
          rc = nc;
        } else {
          a = shifted(mp.h, rc);                                    // lane 0: coef H / 2
        }
      }, v, img, tw, lane);
      // fftshift + RemoveDCComponent (:73-82, :135-137): dc = sum of the shifted second half = x[0..H)
#pragma unroll
      for (int m = 0; m < M / 2; ++m) {
        if constexpr (LEAN) {                                       // waits where it is headed for (every lane re-reads its own)
          const int i0 = 2 * (lane + 64 * m);
          *reinterpret_cast<cpx*>(out + i0 + H) = v[m];
        } else {
          xp[2 * m] = v[m].x;
          xp[2 * m + 1] = v[m].y;
        }
        dc += v[m].x + v[m].y;
      }
      dc = wave_sum(dc);
      if constexpr (LEAN) dc = uniform_d(dc);
      wave_sync();
    }

    // ---- GetAperiodicResponse (:38-68) ----
    wave_sync();
    BinPairs<MH, cpx> mp;
    {
      auto log_aperiodic = [&](const BinPairs<MH, double>& env, const BinPairs<MH, double>& rat) {
        auto la = [&](double e, double r) { return wm_log(cvuv != 0.0 ? e * r : e) / 2.0; };
#pragma unroll
        for (int m = 0; m < MH; ++m) {
          ls[lane + 64 * m] = la(env.k[m], rat.k[m]);
          __builtin_amdgcn_sched_barrier(0);
          ls[H - (lane + 64 * m)] = la(env.r[m], rat.r[m]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (lane == 0) ls[H / 2] = la(env.h, rat.h);
      };
      if constexpr (LEAN) {
        BinPairs<MH, double> env, rat;
        spectral(env, rat);
        log_aperiodic(env, rat);
      } else {
        log_aperiodic(env_keep, rat_keep);
      }
    }
    wave_sync();
    minimum_phase_bp<N>(ls, img, tw, lane, mp);
    // GetNoiseSpectrum (:19-33)
    cpx v[M];
    {
      // LEAN: the draws are fetched here (their addresses hang on a fenced lane), not ahead of the transforms above
      const int ln = LEAN ? opaque_lane(lane) : lane;
      const int roff = r.roff;
      double sum = 0.0;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const int i0 = 2 * (ln + 64 * m);
        const double n0 = i0 < noise_size ? randn_at(rtab, roff + i0) : 0.0;
        const double n1 = i0 + 1 < noise_size ? randn_at(rtab, roff + i0 + 1) : 0.0;
        v[m] = make_double2(n0, n1);
        sum += n0 + n1;
      }
      const double avg = wave_sum(sum) / noise_size;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const int i0 = 2 * (ln + 64 * m);
        if (i0 < noise_size) v[m].x -= avg;
        if (i0 + 1 < noise_size) v[m].y -= avg;
      }
    }
    // the noise spectrum meets the minimum-phase spectrum pair by pair, between the split of the one transform and the
    // un-split of the other (rfft_filter_pairs); the draws up to the next pulse, zeros behind
    rfft_filter_pairs<N>(v, img, tw, lane, (noise_size + 127) >> 7, [&](int m, cpx& a, cpx& b) {
      if (m < MH) {
        a = cmul(mp.k[m < MH ? m : 0], a);
        b = cmul(mp.r[m < MH ? m : 0], b);
      } else {
        a = cmul(mp.h, a);
      }
    });

    // ---- response = (periodic * sqrt(noise_size) + aperiodic) / fft_size (:211-215), fftshifted ----
    const double sq = sqrt((double)noise_size);
    cpx xq[LEAN ? M / 2 : 1];
    const int lo = LEAN ? opaque_lane(lane) : lane;   // LEAN: the DC remover's table is fetched here, not ahead of the transforms
    if constexpr (LEAN) {
#pragma unroll
      for (int m = 0; m < M / 2; ++m)
        xq[m] = periodic ? *reinterpret_cast<const cpx*>(out + 2 * (lo + 64 * m) + H) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int n = lo + 64 * m;
      const int i0 = 2 * n;                           // x-index; shifted position j = (i + H) mod F
      double r0, r1;
      if (m < M / 2) {                                // i < H  ->  j = i + H (second half)
        const double x0 = LEAN ? xq[LEAN ? m : 0].x : xp[LEAN ? 0 : 2 * m];
        const double x1 = LEAN ? xq[LEAN ? m : 0].y : xp[LEAN ? 0 : 2 * m + 1];
        const double2 dr = *reinterpret_cast<const double2*>(dcr + i0 + H);   // unconditional: a load behind the
        const double p0 = periodic ? x0 - dc * dr.x : 0.0;                    // (uniform) branch is waited for on its own
        const double p1 = periodic ? x1 - dc * dr.y : 0.0;
        r0 = (p0 * sq + v[m].x) / F;
        r1 = (p1 * sq + v[m].y) / F;
        out[i0 + H] = r0;
        out[i0 + 1 + H] = r1;
      } else {                                        // i >= H ->  j = i - H (first half, periodic overwritten)
        const double2 dr = *reinterpret_cast<const double2*>(dcr + i0 - H);
        const double p0 = periodic ? -dc * dr.x : 0.0;
        const double p1 = periodic ? -dc * dr.y : 0.0;
        r0 = (p0 * sq + v[m].x) / F;
        r1 = (p1 * sq + v[m].y) / F;
        out[i0 - H] = r0;
        out[i0 + 1 - H] = r1;
      }
    }
    wave_sync();
  }
}


}  // namespace wm

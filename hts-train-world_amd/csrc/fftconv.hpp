// fftconv.hpp -- block FFT convolution (overlap-save) for the long FIR filters of DIO, fused with the four
// zero-crossing event passes of zcfilter.hpp.  One wavefront owns one block.
//
// The reference filters whole signals by multiplying 2^16 .. 2^18-point spectra (dio.cpp:296-343).  Every filter
// is a short FIR (40 .. 961 taps), so round 1 evaluated them directly, tile by tile (zcfilter.hpp: 72 % of the
// issued instructions are FMAs, 39 % of the FP64 peak) -- but a direct FIR costs `taps` multiply-adds per sample
// and band, 1 600 per sample over the eight filters of DIO at 16 kHz.  A block of B = 2048 samples through the
// one-wavefront real FFT (fft.hpp) costs two transforms per block and filter, i.e. about one seventh of the
// instructions at 16 kHz, and unlike the direct form it does not grow with the filter length (48 kHz: taps x 3).
//
//   out[n] = sum_{k < ntap} g[k] in(n + bias - k)         n in [n0, n0 + V),  V = B - ntap + 1
//
// is read off the circular convolution of the block blk[i] = in(n0 + bias - (ntap - 1) + i), i < B, with g:
// entries j >= ntap - 1 of the product are exact linear-convolution values, out[n0 + j - (ntap - 1)].
// Filters are applied as spectra H = FFT(g zero-padded to B) / B, built once per batch by conv_spectrum_kernel
// with the same transform (`/ B`: the inverse real transform is unnormalised, fft.cpp:27-35).
// Several filters of different length share one block: they are delayed to a common (ntap0, bias0) by leading
// zeros, so the forward transform of the block is done once and kept in registers.
//
// Numerically this is the reference's own method on shorter blocks; results differ from the direct FIR of
// round 1 by rounding (1e-16 of the block's scale), and the same parity tests hold.
#pragma once
#include "common.hpp"
#include "fft.hpp"
#include "zcfilter.hpp"

namespace wm {

template <int B> struct ConvCfg {
  static constexpr int N = B / 2, M = N / 64;
  static constexpr int kImg = 2 * FftLds<N>::kElems;           // doubles: FFT image, then the filtered block
};

// LDS of a band kernel that also extracts events (conv_block_events): the FFT image, whose first 64 CMAX + 2 doubles
// hold the filtered block while the events are extracted, and the four ordered index lists, which start right
// behind the block -- inside what is left of the image -- and end at 20 KB, so that eight workgroups (two waves per
// SIMD) fit a CU; with lists of their own behind the image (25.6 KB) it was six.  A list then holds 766 indices;
// a block can have up to 64 CMAX / 2 events of a kind (they cannot fire on consecutive samples), so blocks with
// more than the lists hold -- signals that alternate around zero nearly every sample -- take a direct path.
template <int B, int CMAX> struct ConvEvCfg {
  static constexpr int kSDoubles = (64 * CMAX + 2 + 1) & ~1;
  static constexpr size_t kImgBytes = sizeof(double) * ConvCfg<B>::kImg;
  static constexpr size_t kLdsBytes = kImgBytes > 20480 ? kImgBytes + 4096 : 20480;
  static constexpr int kListCap = (int)((kLdsBytes - sizeof(double) * kSDoubles) / (4 * sizeof(unsigned short)));
  static_assert(kListCap >= 256, "room for the event lists");
  __device__ static unsigned short* lists(double* lds) { return reinterpret_cast<unsigned short*>(lds + kSDoubles); }
};

// H[f][k] = FFT_B(g_f)[k] / B, k = 0 .. B/2, where g_f[k'] = taps[off[f] + k' - delay[f]] for
// k' in [delay[f], delay[f] + ntap[f]) and 0 elsewhere.  One wavefront per filter.
template <int B>
__global__ __launch_bounds__(64) void conv_spectrum_kernel(const double* __restrict__ taps, const int* __restrict__ off,
                                                           const int* __restrict__ ntap, const int* __restrict__ delay,
                                                           cpx* __restrict__ H) {
  constexpr int N = ConvCfg<B>::N, M = ConvCfg<B>::M;
  __shared__ __attribute__((aligned(16))) double smem[ConvCfg<B>::kImg];
  cpx* img = reinterpret_cast<cpx*>(smem);
  const int f = blockIdx.x, lane = threadIdx.x;
  FftTw<N> tw;
  tw.init(lane);
  const double* g = taps + off[f];
  const int d = delay[f], nt = ntap[f];
  cpx v[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int i0 = 2 * (lane + 64 * m) - d, i1 = i0 + 1;
    v[m] = make_double2(i0 >= 0 && i0 < nt ? g[i0] : 0.0, i1 >= 0 && i1 < nt ? g[i1] : 0.0);
  }
  rfft_forward<N>(v, img, img, tw, lane);
  cpx* Hf = H + (int64_t)f * (N + 1);
  const double inv = 1.0 / B;
  for (int k = lane; k <= N; k += 64) Hf[k] = make_double2(img[k].x * inv, img[k].y * inv);
}

// The spectrum of a block, held by PAIRS in registers (fft.hpp, rfft_split_pairs): k[m] = Z[j], r[m] = Z[N - j] for
// j = lane + 64 m < N / 2 (lane 0, m = 0: Z[0] and Z[N]), h = Z[N / 2] (lane 0's counts).
template <int B> struct ConvSpec {
  cpx k[ConvCfg<B>::M / 2], r[ConvCfg<B>::M / 2], h;
};

// Forward transform of a block given as packed pairs v[m] = (blk[2q], blk[2q + 1]), q = lane + 64 m.
template <int B>
__device__ __forceinline__ void conv_forward(cpx (&v)[ConvCfg<B>::M], cpx* img, const FftTw<ConvCfg<B>::N>& tw, int lane,
                                             ConvSpec<B>& z) {
  constexpr int N = ConvCfg<B>::N;
  rfft_forward_pairs<N>(v, img, tw, lane, z.k, z.r, z.h);
}

// out = IFFT(Z . H): v[m] = (out[2q], out[2q + 1]).  The product is taken pair by pair as the inverse transform's
// un-split consumes it: no spectrum goes through LDS on either side of it.
template <int B>
__device__ __forceinline__ void conv_apply(const ConvSpec<B>& z, const cpx* __restrict__ Hf, cpx* img,
                                           const FftTw<ConvCfg<B>::N>& tw, int lane, cpx (&v)[ConvCfg<B>::M]) {
  constexpr int N = ConvCfg<B>::N, M = ConvCfg<B>::M, MH = M / 2;
  cpx hk[MH], hr[MH];
#pragma unroll
  for (int m = 0; m < MH; ++m) {
    hk[m] = Hf[lane + 64 * m];
    hr[m] = Hf[N - (lane + 64 * m)];
  }
  const cpx hh = Hf[N / 2];
  rfft_unsplit_pairs_f<N>([&](int m, cpx& a, cpx& b) {
    if (m < MH) {
      a = cmul(z.k[m < MH ? m : 0], hk[m < MH ? m : 0]);
      b = cmul(z.r[m < MH ? m : 0], hr[m < MH ? m : 0]);
    } else {
      a = cmul(z.h, hh);
    }
  }, v, img, tw, lane);
  fft_backward<N>(v, img, tw, lane);
}

// The four zero-crossing passes (zcfilter.hpp: ZeroCrossingEngine, dio.cpp:357-393; kinds :402-435) of one block
// by one wavefront.  s[0 .. step + 2) holds the filtered samples of the block's outputs n0 .. ; events of sample
// li < step are found.  Lane l owns the consecutive samples [l c, (l + 1) c), c <= CMAX: it reads them into
// registers in one go (a rolled loop over LDS would pay one LDS round trip per sample), keeps the four kinds'
// hits as bit masks, a wave scan of the counts gives every lane its place, and the sample indices go into ordered
// LDS lists.  The fine positions (a division each) are then computed densely over the lists and go to the
// block's slots in sample order.
template <int CMAX>
__device__ __forceinline__ void conv_block_events(const double* s, int n0, int step, int ylen, int tile,
                                                  unsigned short* lists, int list_cap, int* __restrict__ tile_cnt4,
                                                  double* __restrict__ slot, int64_t slot_cap, int lane) {
  static_assert(CMAX <= 32, "one bit per sample in a 32-bit mask");
  const int c = (step + 63) >> 6;                               // <= CMAX (the caller's block geometry)
  const int lo = lane * c, hi = imin(step, lo + c);
  double xs[CMAX + 2];
#pragma unroll
  for (int r = 0; r < CMAX + 2; ++r) xs[r] = s[imin(lo + r, step + 1)];
  unsigned mask[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int r = 0; r < CMAX; ++r) {
    const int li = lo + r, i = n0 + li;
    const double x0 = xs[r], x1 = xs[r + 1], x2 = xs[r + 2];
    const bool in1 = li < hi && i < ylen - 1, in2 = in1 && i < ylen - 2;
    const double p0 = x1 - x0, p1 = x2 - x1;                    // (:424-425)
    const bool f0 = in1 && 0.0 < x0 && x1 <= 0.0;               // positive -> non-positive (:361-363)
    const bool f1 = in1 && 0.0 < -x0 && -x1 <= 0.0;             // same on the negated signal (:419-422)
    const bool f2 = in2 && 0.0 < p0 && p1 <= 0.0;
    const bool f3 = in2 && 0.0 < -p0 && -p1 <= 0.0;
    mask[0] |= (f0 ? 1u : 0u) << r;
    mask[1] |= (f1 ? 1u : 0u) << r;
    mask[2] |= (f2 ? 1u : 0u) << r;
    mask[3] |= (f3 ? 1u : 0u) << r;
  }
  int total[4], first[4];
#pragma unroll
  for (int ty = 0; ty < 4; ++ty) {
    const int cnt = __popc(mask[ty]);
    const int incl = wave_scan_incl_i(cnt);
    first[ty] = incl - cnt;
    total[ty] = __builtin_amdgcn_readlane(incl, 63);
  }
  if (lane < 4) tile_cnt4[lane] = lane == 0 ? total[0] : (lane == 1 ? total[1] : (lane == 2 ? total[2] : total[3]));
  if (imax(imax(total[0], total[1]), imax(total[2], total[3])) > list_cap) {        // wave-uniform, rare
    // more events than the lists hold: every lane computes the positions of its own events where they are found
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) {
      int at = first[ty];
      unsigned m = mask[ty];
      while (m) {
        const int r = __ffs((int)m) - 1;
        const int li = lo + r, i = n0 + li;
        const double x0 = s[li], x1 = s[li + 1];
        double fine;
        if (ty < 2) {
          fine = (i + 1) - x0 / (x1 - x0);
        } else {
          const double x2 = s[li + 2];
          const double p0 = x1 - x0, p1 = x2 - x1;
          fine = (i + 1) - p0 / (p1 - p0);
        }
        slot[(int64_t)ty * slot_cap + (int64_t)tile * kZcSlot + at++] = fine;
        m &= m - 1u;
      }
    }
    wave_sync();
    return;
  }
#pragma unroll
  for (int ty = 0; ty < 4; ++ty) {
    int at = first[ty];
    unsigned m = mask[ty];
    while (m) {                                                 // a few hits per lane and kind
      const int r = __ffs((int)m) - 1;
      lists[ty * list_cap + at++] = (unsigned short)(lo + r);
      m &= m - 1u;
    }
  }
  wave_sync();
#pragma unroll
  for (int ty = 0; ty < 4; ++ty) {
    for (int j = lane; j < total[ty]; j += 64) {
      const int li = lists[ty * list_cap + j];
      const int i = n0 + li;
      const double x0 = s[li], x1 = s[li + 1];
      double fine;
      if (ty < 2) {
        fine = (i + 1) - x0 / (x1 - x0);                        // :378-382
      } else {
        const double x2 = s[li + 2];
        const double p0 = x1 - x0, p1 = x2 - x1;
        fine = (i + 1) - p0 / (p1 - p0);
      }
      slot[(int64_t)ty * slot_cap + (int64_t)tile * kZcSlot + j] = fine;
    }
  }
  wave_sync();
}

}  // namespace wm

// spectrum.hpp -- DCCorrection and LinearSmoothing on an LDS-resident half spectrum WITH MARGINS, one wavefront.
//
// Restates common.cpp:56-75 (DCCorrection) and common.cpp:27-46, 77-111 (LinearSmoothing) like common.hpp's
// dc_correction_lds / linear_smoothing_lds, but for a spectrum stored with BM free doubles on either side:
//
//     arr[-BM .. -1] | arr[0 .. HALF] | arr[HALF+1 .. HALF+BM]        (arr points at bin 0)
//
// LinearSmoothing works on the spectrum mirrored by b = int(width F / fs) + 1 bins at both ends (b <= BM is the
// caller's duty).  With margins the mirror is two short fills instead of an index computation per element, the
// mirrored array is contiguous, and its cumulative sum can replace it in place -- no second LDS array:
//
//   fill      arr[-t] = arr[t], arr[HALF+t] = arr[HALF-t], t = 1..b            (2 b elements)
//   cumsum    lane l owns CH consecutive entries of ext = arr - b (odd CH: conflict-free strided access), sums them
//             in registers, one wave scan stitches the 64 partial sums, the result overwrites ext
//   interp    lane l owns BI consecutive bins: the two interp1Q lookups of bin i are knots i + c_lo and i + c_hi
//             with constant offsets and fractions (common.hpp), so a lane reads two runs of BI + 1 consecutive
//             knots instead of four scattered ones per bin; results wait in registers until every lane has
//             read its knots, then overwrite arr[0 .. HALF]
//
// Everything ends with a barrier.
#pragma once
#include "common.hpp"

namespace wm {

template <int HALF, int BM> struct SmoothCfg {
  static constexpr int kCh = ((HALF + 2 * BM + 1 + 63) / 64) | 1;     // entries per lane of the scan (odd)
  static constexpr int kBi = ((HALF + 1 + 63) / 64) | 1;              // bins per lane of the interpolation (odd)
  // doubles needed from (arr - BM): the scan writes 64 * kCh entries from ext = arr - b >= arr - BM, the
  // interpolation's stores reach arr[64 * kBi - 1]
  static constexpr int kRegion = BM + (64 * kCh > 64 * kBi ? 64 * kCh : 64 * kBi) + 2;
  static_assert(64 * kCh >= 64 * kBi + (3 * BM) / 2 + 3, "interpolation reads stay inside the scanned entries");
};

// DCCorrection in place on arr[0..HALF]; the corrected bins are 0 .. int(f0 * fft_size / fs), which must be at most
// COVER (by default the margin BM; CheapTrick's wide instantiation lets f0 up to fs / 2 through, i.e. HALF bins,
// while its margin is sized by the smoothing width 2 f0 / 3).
template <int HALF, int BM, int COVER = BM>
__device__ __forceinline__ void dc_correction_margin(double* arr, double f0, int fs, int fft_size, int lane) {
  constexpr int T = (COVER + 1 + 63) / 64;
  const double inv_fft = 1.0 / fft_size;               // power of two: exact
  const int upper = 2 + (int)(f0 * fft_size / fs);
  const int nrep = upper - 1;
  const double inv_dx = -(double)fft_size / fs;
  double r[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int i = lane + 64 * t;
    r[t] = 0.0;
    if (64 * t < nrep) {
      const double axis = (double)imin(i, nrep - 1) * fs * inv_fft;
      r[t] = interp1q_lds_r(f0, inv_dx, arr, upper + 1, axis);
    }
  }
  wave_sync();                                          // every read of arr happens before any write (common.cpp:62-74)
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int i = lane + 64 * t;
    if (i < nrep) arr[i] += r[t];
  }
  wave_sync();
}

// LinearSmoothing in place on arr[0..HALF]; b = int(width * fft_size / fs) + 1 <= BM.
// `fin(q, s)` is what is stored for the lane's q-th bin (bin lane * kBi + q) whose smoothed value is s: a caller whose
// next step is elementwise passes it here, with its other operand held in the SAME blocked layout (kBi consecutive bins
// per lane), instead of reading the smoothed spectrum back in another layout and storing the result again.
template <int HALF, int BM, class Fin>
__device__ __forceinline__ void linear_smoothing_margin(double* arr, double width, int fs, int fft_size, int lane,
                                                        Fin fin) {
  constexpr int CH = SmoothCfg<HALF, BM>::kCh, BI = SmoothCfg<HALF, BM>::kBi;
  const double inv_fft = 1.0 / fft_size;               // power of two: x * inv_fft == x / fft_size exactly
  const double wq = width * fft_size / fs;             // width in bins
  const int b = (int)wq + 1;
  const int len = HALF + 2 * b + 1;
  for (int t = 1 + lane; t <= b; t += 64) {             // mirror (common.cpp:85-92)
    arr[-t] = arr[t];
    arr[HALF + t] = arr[HALF - t];
  }
  wave_sync();
  double* ext = arr - b;
  const int beg = lane * CH;
  double v[CH];
#pragma unroll
  for (int q = 0; q < CH; ++q) v[q] = ext[beg + q];
#pragma unroll
  for (int q = 0; q < CH; ++q) {                        // cumsum * fs / F (common.cpp:38-41)
    const double term = (beg + q < len) ? v[q] * fs * inv_fft : 0.0;
    v[q] = q == 0 ? term : v[q - 1] + term;
  }
  const double carry = wave_scan_incl(v[CH - 1]) - v[CH - 1];
  wave_sync();                                          // all lanes have read their entries
#pragma unroll
  for (int q = 0; q < CH; ++q) ext[beg + q] = v[q] + carry;
  wave_sync();
  const double c_lo = (b - 0.5) - 0.5 * wq, c_hi = c_lo + wq;     // knot of bin i: i + c (common.cpp:99-108)
  const int bl = (int)c_lo, bh = (int)c_hi;
  const double fl = c_lo - bl, fh = c_hi - bh;
  const double inv_width = 1.0 / width;
  const int i0 = lane * BI;
  // in chunks of at most 12 bins: the two runs of knots of a chunk are in flight together, the results of all
  // chunks wait in registers (a lane's BI results; 33 at a half spectrum of 2048 bins)
  constexpr int CK = BI <= 17 ? BI : (BI + 2) / 3;
  double out[BI];
#pragma unroll
  for (int c0 = 0; c0 < BI; c0 += CK) {
    double lo[CK + 1], hi[CK + 1];
#pragma unroll
    for (int q = 0; q <= CK; ++q) {
      lo[q] = ext[i0 + bl + c0 + q];
      hi[q] = ext[i0 + bh + c0 + q];
    }
#pragma unroll
    for (int q = 0; q < CK; ++q) {
      if (c0 + q < BI) {
        const double l = lo[q] + (lo[q + 1] - lo[q]) * fl;
        const double h = hi[q] + (hi[q + 1] - hi[q]) * fh;
        out[c0 + q] = (h - l) * inv_width;
      }
    }
  }
  wave_sync();                                          // all knots read: the spectrum may be overwritten
  // unconditional: bins beyond HALF land in the right margin (rewritten by the next mirror fill)
#pragma unroll
  for (int q = 0; q < BI; ++q) arr[i0 + q] = fin(q, out[q]);
  wave_sync();
}
template <int HALF, int BM>
__device__ __forceinline__ void linear_smoothing_margin(double* arr, double width, int fs, int fft_size, int lane) {
  linear_smoothing_margin<HALF, BM>(arr, width, fs, fft_size, lane, [](int, double s) { return s; });
}

}  // namespace wm

// peel.hpp -- "the sum of all but the K largest of a wavefront's values" without a global sort (d4c.cpp:215-220:
// std::sort of the band's power spectrum only feeds that sum).  Every lane sorts its own values in registers
// (sort_desc), parks the column in LDS, and peel_largest() finds how many of its largest each lane gives up.
#pragma once
#include "common.hpp"

namespace wm {

// Descending sort of a[0..NS) in registers, NS a power of two: Batcher's odd-even merge sort.  All
// loop bounds are compile-time, so after unrolling every compare-exchange has static register indices.
template <int NS>
__device__ __forceinline__ void sort_desc(double (&a)[NS + 1]) {
#pragma unroll
  for (int p = 1; p < NS; p <<= 1) {
#pragma unroll
    for (int k = p; k >= 1; k >>= 1) {
#pragma unroll
      for (int j = k % p; j + k < NS; j += 2 * k) {
#pragma unroll
        for (int i = 0; i < k; ++i) {
          if (i + j + k < NS && (i + j) / (2 * p) == (i + j + k) / (2 * p)) {
            const double hi = fmax(a[i + j], a[i + j + k]), lo = fmin(a[i + j], a[i + j + k]);
            a[i + j] = hi;
            a[i + j + k] = lo;
          }
        }
      }
    }
  }
}

// Which entries of a wavefront's values are among the K largest, for lists sorted descending per lane and parked
// as columns in LDS: heads[m * 64 + lane] = m-th largest of the lane, followed by at least THREE rows of -1 (values
// are powers, >= 0).  Returns how many of its entries each lane gives up; the counts add up to K.  Only the sum
// of what is left matters to the caller (d4c.cpp:215-220), so ties may be broken anyhow.
//
// Round 1 peeled one value per wave-wide maximum: K = 65 dependent steps of a 6-stage reduction and an LDS read,
// 15 % of d4c_kernel's instructions at 16 kHz and 40 % of the band kernel's at 48 kHz.  Here a step takes every
// value that provably beats all values below the lanes' first two entries: T = max over lanes of the THIRD entry;
// whatever of a lane's first two entries is >= T is larger than every entry not looked at.  If that is more than
// what is left to take, the same with the second entry (fewer candidates), and if that is still too many the
// largest of those heads are taken one by one (they beat everything else, so no list advances).  On spectra
// with a main lobe or a few peaks over noise that is 3-6 reductions plus about 5 single steps instead of 65.
__device__ __forceinline__ int peel_largest(const double* heads, int K, int lane) {
  int taken = 0, r = K;
  while (r > 0) {                                               // every pass takes at least one value or leaves
    const double* col = heads + taken * 64 + lane;
    const double cur = col[0], nxt = col[64], thr = col[128];
    double T = wave_max(thr);
    bool a = cur >= T && cur >= 0.0, b = nxt >= T && nxt >= 0.0;
    int c = __popcll(__ballot(a)) + __popcll(__ballot(b));
    if (c > r) {
      T = wave_max(nxt);
      a = cur >= T && cur >= 0.0;
      b = false;
      c = __popcll(__ballot(a));
    }
    if (c == 0) break;                                          // NaNs only (the caller's total is NaN as well)
    if (c <= r) {
      taken += (a ? 1 : 0) + (b ? 1 : 0);
      r -= c;
      continue;
    }
    double cd = a ? cur : -1.0;
#pragma unroll 1
    for (int i = 0; i < r; ++i) {
      const double mx = wave_max(cd);
      const int winner = __ffsll((long long)__ballot(cd == mx)) - 1;
      const bool me = lane == winner;
      taken += me ? 1 : 0;
      cd = me ? -1.0 : cd;
    }
    r = 0;
  }
  return taken;
}

}  // namespace wm

// zcfilter.hpp -- FIR filtering fused with the four zero-crossing event passes, shared by DIO and
// Harvest (externs/WORLD_v2/src/dio.cpp:296-435 and harvest.cpp:99-238 are the same construction
// with different filters).  One 256-thread workgroup filters one (utterance, band) signal tile by
// tile and appends the fine zero-crossing positions of each of the four event kinds to ordered
// lists (ballot + popcount compaction keeps the reference's sample order).
#pragma once
#include "common.hpp"

namespace wm {

constexpr int kBandK = 8;                       // outputs per thread
constexpr int kBandTile = 256 * kBandK;         // samples per tile

// LDS doubles needed by filter_and_events() for filters of up to ntap_max taps
__host__ __device__ inline int zc_pad8(int n) { return (n + 7) & ~7; }
__host__ __device__ inline int zc_lds_doubles(int ntap_max) {
  const int zspan = kBandTile + zc_pad8(ntap_max) + 8;
  const int stride = (zspan + kBandK - 1) / kBandK + 1;
  return kBandK * stride + zc_pad8(ntap_max) + kBandTile;
}

// filtered[n] = sum_{k < ntap} taps[k] * sig[n + bias - k], n in [0, ylen), where sig[m] is read for
// m in [lo, hi) and is zero elsewhere.  Events (ZeroCrossingEngine, dio.cpp:357-393) of the four
// kinds go to ev[kind * cap + i] in order; ev_cnt4[kind] receives the number of edges.
// Kinds (dio.cpp:402-435): 0 negative-going, 1 positive-going, 2 peaks, 3 dips.
__device__ __forceinline__ void filter_and_events(const double* __restrict__ sig, int lo, int hi, int ylen,
                                                  const double* __restrict__ taps, int ntap, int bias,
                                                  double* __restrict__ ev, int cap, int* __restrict__ ev_cnt4,
                                                  double* lds) {
  // LDS: transposed signal tile (element e at [(e % K) * stride + e / K]), taps (zero-padded to a
  // multiple of 8), filtered tile.  The tile carries 8 extra leading elements so that the 8-tap
  // register rotation below never indexes below zero.
  const int ntp = zc_pad8(ntap);
  const int zspan = kBandTile + ntp + 8;                        // elements needed per tile
  const int stride = (zspan + kBandK - 1) / kBandK + 1;
  double* zt = lds;                                             // [K * stride]
  double* w = zt + kBandK * stride;                             // [ntp]
  double* s = w + ntp;                                          // [kBandTile] filtered samples
  __shared__ int wave_cnt[4][4];                                // [kind][wave]
  __shared__ int run_cnt[4];
  for (int j = threadIdx.x; j < ntp; j += 256) w[j] = j < ntap ? taps[j] : 0.0;
  if (threadIdx.x < 4) run_cnt[threadIdx.x] = 0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int step = kBandTile - 2;                               // tiles overlap by 2 (s[i+1], s[i+2] look-ahead)

  for (int n0 = 0; n0 < ylen; n0 += step) {
    __syncthreads();
    // tile element e <-> signal index zbase + e with zbase = n0 + bias - (ntp - 1) - 8
    const int zbase = n0 + bias - (ntp - 1) - 8;
    for (int e = threadIdx.x; e < zspan; e += 256) {
      const int m = zbase + e;
      const double val = (m >= lo && m < hi) ? sig[m] : 0.0;
      zt[(e % kBandK) * stride + e / kBandK] = val;
    }
    __syncthreads();
    // thread t: outputs n0 + t*K + q, q < K; output q at tap k reads element
    // e = t*K + q + (ntp - 1) + 8 - k.  Taps are consumed 8 at a time: the 8 window registers
    // r[q] (elements for tap k) and 8 freshly loaded lower elements nw[] cover all 64 products of
    // the group with static register indices, so the window slides without register moves.
    {
      double acc[kBandK];
#pragma unroll
      for (int q = 0; q < kBandK; ++q) acc[q] = 0.0;
      const int t = threadIdx.x;
      const int ebase = t * kBandK + (ntp - 1) + 8;             // element of output 0 at tap 0
      double r[kBandK];
#pragma unroll
      for (int q = 0; q < kBandK; ++q) {
        const int e = ebase + q;
        r[q] = zt[(e % kBandK) * stride + e / kBandK];
      }
      for (int k = 0; k < ntp; k += 8) {
        double nw[8], wk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int e = ebase - k - 1 - i;                      // >= 0 thanks to the 8 leading elements
          nw[i] = zt[(e % kBandK) * stride + e / kBandK];
          wk[i] = w[k + i];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
          for (int q = 0; q < kBandK; ++q) {
            const double el = (q - j >= 0) ? r[(q - j) & 7] : nw[(j - q - 1) & 7];
            acc[q] += wk[j] * el;
          }
        }
#pragma unroll
        for (int q = 0; q < kBandK; ++q) r[q] = nw[7 - q];
      }
#pragma unroll
      for (int q = 0; q < kBandK; ++q) s[t * kBandK + q] = acc[q];
    }
    __syncthreads();
    // ---- zero crossings over samples i in [n0, n0 + step) ----
    for (int rowb = 0; rowb < step; rowb += 256) {
      const int li = rowb + threadIdx.x;          // local index
      const int i = n0 + li;
      bool f[4] = {false, false, false, false};
      double fine[4] = {0.0, 0.0, 0.0, 0.0};
      if (li < step && i < ylen - 1) {
        const double a = s[li], b = s[li + 1];
        // kind 0: positive -> non-positive (dio.cpp:361-363); kind 1 on the negated signal (:419-422)
        f[0] = 0.0 < a && b <= 0.0;
        f[1] = 0.0 < -a && -b <= 0.0;
        if (f[0] || f[1]) fine[f[0] ? 0 : 1] = (i + 1) - a / (b - a);          // :378-382
        if (i < ylen - 2) {
          const double c = s[li + 2];
          const double p0 = b - a, p1 = c - b;     // (-s[i]) - (-s[i+1]) (:424-425)
          f[2] = 0.0 < p0 && p1 <= 0.0;
          f[3] = 0.0 < -p0 && -p1 <= 0.0;
          if (f[2] || f[3]) fine[f[2] ? 2 : 3] = (i + 1) - p0 / (p1 - p0);
        }
      }
      unsigned long long bal[4];
#pragma unroll
      for (int ty = 0; ty < 4; ++ty) {
        bal[ty] = __ballot(f[ty]);
        if (lane == 0) wave_cnt[ty][wv] = __popcll(bal[ty]);
      }
      __syncthreads();
#pragma unroll
      for (int ty = 0; ty < 4; ++ty) {
        int base = run_cnt[ty];
        for (int q = 0; q < wv; ++q) base += wave_cnt[ty][q];
        if (f[ty]) {
          const int rank = __popcll(bal[ty] & ((1ull << lane) - 1ull));
          const int dst = base + rank;
          if (dst < cap) ev[(int64_t)ty * cap + dst] = fine[ty];
        }
      }
      __syncthreads();
      if (threadIdx.x < 4) {
        int tot = 0;
        for (int q = 0; q < 4; ++q) tot += wave_cnt[threadIdx.x][q];
        run_cnt[threadIdx.x] += tot;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  if (threadIdx.x < 4) ev_cnt4[threadIdx.x] = imin(run_cnt[threadIdx.x], cap);
}

// interp1 (matlabfunctions.cpp:136-182) over a zero-crossing track given by its fine edges:
// locations[j] = (e[j] + e[j+1]) / 2 / fs, intervals[j] = fs / (e[j+1] - e[j]), j < n (dio.cpp:384-387)
__device__ __forceinline__ double zc_track(const double* __restrict__ e, int n, double fs, double t) {
  int lo = 0, hi = n;                         // upper_bound on locations
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const double loc = (e[mid] + e[mid + 1]) / 2.0 / fs;
    if (loc <= t) lo = mid + 1; else hi = mid;
  }
  const int k = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
  const double x0 = (e[k - 1] + e[k]) / 2.0 / fs, x1 = (e[k] + e[k + 1]) / 2.0 / fs;
  const double y0 = fs / (e[k] - e[k - 1]), y1 = fs / (e[k + 1] - e[k]);
  const double h = x1 - x0;
  const double sfrac = (t - x0) / h;
  return y0 + sfrac * (y1 - y0);
}


}  // namespace wm

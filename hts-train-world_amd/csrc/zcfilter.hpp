// zcfilter.hpp -- FIR filtering fused with the four zero-crossing event passes, shared by DIO and
// Harvest (externs/WORLD_v2/src/dio.cpp:296-435 and harvest.cpp:99-238 are the same construction
// with different filters).
//
// The ordered event lists need a running count along the signal; walking the tiles of one signal one
// after another makes that walk (tens of tiles, thousands of FMAs per thread each) the critical
// path of the whole launch.  Here every tile of every (utterance, band) signal is its own 256-thread
// workgroup and the ordering is restored afterwards:
//   filter_tile_events   filter the tile, write its events (in order) into the tile's own slot of a
//                        staging array and its four counts
//   zc_scan_tiles        per-tile counts -> exclusive offsets, list lengths
//   zc_compact_signal    move the events from the tiles' slots to offset + rank of the ordered lists
// Only actual events travel twice; the FIR runs once.
#pragma once
#include "common.hpp"

namespace wm {

constexpr int kBandK = 8;                       // outputs per thread
constexpr int kBandTile = 256 * kBandK;         // samples per tile
constexpr int kZcStep = kBandTile - 2;          // tiles overlap by 2 (s[i+1], s[i+2] look-ahead)
constexpr int kZcSlot = 1024;                   // staging slots per tile and event kind
// Row stride (doubles) of the transposed signal tile; the two users pick the smallest that holds their
// longest filter.  stride % 32 == 24 spreads the 8 rows over distinct LDS bank groups.
constexpr int kZcStrideDio = 312;               // taps <= 8 * 311 - 2056 = 432
constexpr int kZcStrideHarvest = 408;           // taps <= 8 * 407 - 2056 = 1200
constexpr int kZcStrideLong = 520;              // taps <= 8 * 519 - 2056 = 2096 (DIO low-cut at 48 kHz)

__host__ __device__ inline int zc_pad16(int n) { return (n + 15) & ~15; }
__host__ __device__ inline int zc_tiles(int ylen) { return (ylen + kZcStep - 1) / kZcStep; }
template <int STRIDE> __host__ __device__ inline int zc_max_taps() { return 8 * (STRIDE - 1) - kBandTile - 8; }
// LDS doubles needed for filters of up to ntap_max taps
template <int STRIDE> __host__ __device__ inline int zc_lds_doubles(int ntap_max) {
  return kBandK * STRIDE + zc_pad16(ntap_max) + kBandTile;
}

// acc[q] = sum_{k < ntp} w[k] * element(8 t + ntp + 7 + q - k) of the transposed tile zt (see
// filter_tile_events for the layout); ntp is a multiple of 16 and w is zero-padded to it.
template <int STRIDE>
__device__ __forceinline__ void fir_tile_accumulate(const double* zt, const double* w, int ntp, int t,
                                                    double (&acc)[kBandK]) {
#pragma unroll
  for (int q = 0; q < kBandK; ++q) acc[q] = 0.0;
  // window at tap 0: elements 8t + ntp + 7 + q -> q = 0: row 7, column t + ntp/8; q >= 1: row q - 1, one further
  const double* zc = zt + t + ntp / 8;                          // column of the current group's loads
  double a[8], b[8];
  a[0] = zc[7 * STRIDE];
#pragma unroll
  for (int q = 1; q < 8; ++q) a[q] = zc[(q - 1) * STRIDE + 1];
  const double2* w2 = reinterpret_cast<const double2*>(w);
  for (int k = 0; k < ntp; k += 16) {
    // group 1: window a[], loads b[i] = element 8t + ntp + 6 - k - i
    double wk[8];
#pragma unroll
    for (int i = 0; i < 7; ++i) b[i] = zc[(6 - i) * STRIDE];
    b[7] = zc[7 * STRIDE - 1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double2 p = w2[k / 2 + i];
      wk[2 * i] = p.x;
      wk[2 * i + 1] = p.y;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int q = 0; q < kBandK; ++q) acc[q] += wk[j] * ((q - j >= 0) ? a[(q - j) & 7] : b[(j - q - 1) & 7]);
    }
    // group 2: window is b[] reversed (b[7 - q]), loads go into a[] reversed (a[7 - i])
#pragma unroll
    for (int i = 0; i < 7; ++i) a[7 - i] = zc[(6 - i) * STRIDE - 1];
    a[0] = zc[7 * STRIDE - 2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double2 p = w2[k / 2 + 4 + i];
      wk[2 * i] = p.x;
      wk[2 * i + 1] = p.y;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int q = 0; q < kBandK; ++q)
        acc[q] += wk[j] * ((q - j >= 0) ? b[(7 - (q - j)) & 7] : a[(7 - (j - q - 1)) & 7]);
    }
    zc -= 2;
  }
}

// filtered[n] = sum_{k < ntap} taps[k] * sig[n + bias - k], n in [0, ylen), where sig[m] is read for
// m in [lo, hi) and is zero elsewhere.  Events (ZeroCrossingEngine, dio.cpp:357-393) of the four
// kinds go to ev[kind * cap + i] in order.  Kinds (dio.cpp:402-435): 0 negative-going,
// 1 positive-going, 2 peaks, 3 dips.
// The tile's events of kind K go to slot[K * slot_cap + tile * kZcSlot + rank] in sample order and
// tile_cnt4[K] receives their number (an event kind cannot fire on two consecutive samples, so a
// tile of 2046 samples holds at most 1023 of a kind).
//
// Layout.  Tile element e (signal index zbase + e) sits at zt[(e % 8) * STRIDE + e / 8], so that the
// eight consecutive outputs of a thread and their sliding window are eight conflict-free rows.
// Thread t produces outputs n0 + 8t + q; at tap k output q reads element 8t + ntp + 7 + q - k.
// Taps go 16 per trip in two groups of 8.  A group holds the 8 window elements of its first tap
// in registers and loads the 8 elements below them; all 64 products of the group then use static
// register indices.  The second group takes the first group's loads as its window and loads into
// the registers of the old window, so the window slides without a single register move, and with
// STRIDE a compile-time constant every LDS address is one moving column plus an immediate.
// Wave w owns the 512 consecutive samples [512 w, 512 w + 512) of the tile, as 8 rows of 64.
template <int STRIDE>
__device__ __forceinline__ void filter_tile_events(const double* __restrict__ sig, int lo, int hi, int ylen,
                                                   const double* __restrict__ taps, int ntap, int bias, int tile,
                                                   int* __restrict__ tile_cnt4, double* __restrict__ slot,
                                                   int64_t slot_cap, double* lds) {
  const int ntp = zc_pad16(ntap);
  const int zspan = kBandTile + ntp + 8;
  double* zt = lds;                                             // [8 * STRIDE]
  double* w = zt + kBandK * STRIDE;                             // [ntp], zero-padded taps
  double* s = w + ntp;                                          // [kBandTile] filtered samples
  __shared__ int wave_cnt[4][4];                                // [kind][wave]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n0 = tile * kZcStep;
  for (int j = threadIdx.x; j < ntp; j += 256) w[j] = j < ntap ? taps[j] : 0.0;
  {
    // 8 extra leading elements keep the lowest load of the last group inside the tile
    const int zbase = n0 + bias - (ntp - 1) - 8;
    for (int e = threadIdx.x; e < zspan; e += 256) {
      const int m = zbase + e;
      const double val = sig[imin(hi - 1, imax(lo, m))];        // clamped address, no branch around the load
      zt[(e & 7) * STRIDE + (e >> 3)] = (m >= lo && m < hi) ? val : 0.0;
    }
  }
  __syncthreads();
  {
    double acc[kBandK];
    fir_tile_accumulate<STRIDE>(zt, w, ntp, threadIdx.x, acc);
#pragma unroll
    for (int q = 0; q < kBandK; ++q) s[threadIdx.x * kBandK + q] = acc[q];
  }
  __syncthreads();
  // ---- zero crossings (ZeroCrossingEngine, dio.cpp:357-393; kinds :402-435) ----
  // Events are sparse (a few per hundred samples), so their fine positions -- a division each -- are not
  // computed where they are found (one divergent pass per row and kind, 32 of them, nearly all taken by some
  // lane).  Each wavefront lists the sample indices of its 512 samples per kind, in order (an event kind
  // cannot fire on two consecutive samples: at most 256 entries); the transposed tile is dead after the
  // barrier above and holds the 16 lists.  They are then worked off densely in list order.
  static_assert(16 * 256 * (int)sizeof(int) <= kBandK * STRIDE * (int)sizeof(double), "event lists fit the tile");
  int* evl = reinterpret_cast<int*>(zt);                        // [kind][wave][256]
  int cnt[4] = {0, 0, 0, 0};
#pragma unroll
  for (int row = 0; row < 8; ++row) {
    const int li = wv * 512 + row * 64 + lane;
    const int i = n0 + li;
    bool f[4] = {false, false, false, false};
    if (li < kZcStep && i < ylen - 1) {
      const double x0 = s[li], x1 = s[li + 1];
      f[0] = 0.0 < x0 && x1 <= 0.0;                             // positive -> non-positive (:361-363)
      f[1] = 0.0 < -x0 && -x1 <= 0.0;                           // same on the negated signal (:419-422)
      if (i < ylen - 2) {
        const double x2 = s[li + 2];
        const double p0 = x1 - x0, p1 = x2 - x1;                // (:424-425)
        f[2] = 0.0 < p0 && p1 <= 0.0;
        f[3] = 0.0 < -p0 && -p1 <= 0.0;
      }
    }
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) {
      const unsigned long long bal = __ballot(f[ty]);
      if (f[ty]) evl[(ty * 4 + wv) * 256 + cnt[ty] + __popcll(bal & ((1ull << lane) - 1ull))] = li;
      cnt[ty] += __popcll(bal);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) wave_cnt[ty][wv] = cnt[ty];
  }
  __syncthreads();
  if (threadIdx.x < 4)
    tile_cnt4[threadIdx.x] = wave_cnt[threadIdx.x][0] + wave_cnt[threadIdx.x][1] + wave_cnt[threadIdx.x][2] +
                             wave_cnt[threadIdx.x][3];
#pragma unroll
  for (int ty = 0; ty < 4; ++ty) {
    const int e1 = wave_cnt[ty][0], e2 = e1 + wave_cnt[ty][1], e3 = e2 + wave_cnt[ty][2];
    const int total = e3 + wave_cnt[ty][3];
    for (int j = threadIdx.x; j < total; j += 256) {
      const int q = (j >= e1 ? 1 : 0) + (j >= e2 ? 1 : 0) + (j >= e3 ? 1 : 0);
      const int first = q == 0 ? 0 : (q == 1 ? e1 : (q == 2 ? e2 : e3));
      const int li = evl[(ty * 4 + q) * 256 + (j - first)];
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
      slot[(int64_t)ty * slot_cap + tile * kZcSlot + j] = fine;
    }
  }
}

// per-tile event counts -> exclusive offsets (in place; entry ntiles receives the total) and the list
// lengths; one wavefront per signal.  tile_cnt has 4 * (ntiles + 1) entries.
__device__ __forceinline__ void zc_scan_tiles(int* __restrict__ tile_cnt, int ntiles, int cap,
                                              int* __restrict__ ev_cnt4, int lane) {
  int run[4] = {0, 0, 0, 0};
  for (int t0 = 0; t0 < ntiles; t0 += 64) {
    const int t = t0 + lane;
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) {
      const int c = t < ntiles ? tile_cnt[4 * t + ty] : 0;
      const int incl = wave_scan_incl_i(c);
      if (t < ntiles) tile_cnt[4 * t + ty] = run[ty] + incl - c;
      run[ty] += __builtin_amdgcn_readlane(incl, 63);
    }
  }
  if (lane < 4) {
    int v = run[0];
    if (lane == 1) v = run[1];
    if (lane == 2) v = run[2];
    if (lane == 3) v = run[3];
    ev_cnt4[lane] = imin(v, cap);
    tile_cnt[4 * ntiles + lane] = v;                            // end offset of the last tile
  }
}

// Move a signal's staged events to their places in the ordered lists, by one workgroup: its scanned offsets (4 * (ntiles + 1) ints, layout [tile][kind]) go to
// LDS once (`lds_off`, kZcCompactTiles + 1 tiles at most; longer signals read the offsets from memory), then every
// wavefront takes tiles and copies each tile's run of events to its place in the ordered list.  One workgroup per (tile, signal) -- a handful of events each, 1.6 M
// workgroups on configs[2] -- cost more in dispatch than the copies themselves.
constexpr int kZcCompactTiles = 1023;
__device__ __forceinline__ void zc_compact_signal(const double* __restrict__ slot, int64_t slot_cap, int ntiles,
                                                  const int* __restrict__ off, double* __restrict__ ev, int cap,
                                                  int* lds_off) {
  const bool staged = ntiles <= kZcCompactTiles;
  if (staged)
    for (int i = threadIdx.x; i < 4 * (ntiles + 1); i += blockDim.x) lds_off[i] = off[i];
  __syncthreads();
  const int* o4 = staged ? lds_off : off;
  // A wavefront per tile and kind: the tile's events are consecutive in its slot and consecutive in the list, so a
  // tile is two offset reads and a run of coalesced copies.  (Round 3 gave a thread a PLACE of the list and found its
  // tile by bisection of the offsets: ten dependent LDS reads per event, 80 % of the kernel's cycles waiting.)
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
#pragma unroll 1
  for (int ty = 0; ty < 4; ++ty) {
    const double* src = slot + (int64_t)ty * slot_cap;
    double* dst = ev + (int64_t)ty * cap;
    for (int tile = wv; tile < ntiles; tile += nw) {
      const int b = o4[4 * tile + ty];
      const int e = imin(o4[4 * (tile + 1) + ty], cap);
      const double* st = src + (int64_t)tile * kZcSlot;
      for (int j = b + lane; j < e; j += 64) dst[j] = st[j - b];
    }
  }
}

// largest half-sum h with fl(h / fs) <= t (the division is monotone in h); `exact` false if it could not be
// pinned down within 4 ulps of t * fs (t == 0)
__device__ __forceinline__ double zc_hmax(double fs, double t, bool& exact) {
  double hmax = t * fs;
  for (int it = 0; it < 4 && hmax / fs > t; ++it) hmax = nextafter(hmax, -HUGE_VAL);
  for (int it = 0; it < 4 && nextafter(hmax, HUGE_VAL) / fs <= t; ++it) hmax = nextafter(hmax, HUGE_VAL);
  exact = hmax / fs <= t && nextafter(hmax, HUGE_VAL) / fs > t;
  return hmax;
}
// interp1 (matlabfunctions.cpp:136-182) over a zero-crossing track given by its fine edges:
// locations[j] = (e[j] + e[j+1]) / 2 / fs, intervals[j] = fs / (e[j+1] - e[j]), j < n (dio.cpp:384-387)
__device__ __forceinline__ double zc_track(const double* __restrict__ e, int n, double fs, double t) {
  // upper_bound on locations[j] = fl(fl((e[j] + e[j+1]) / 2) / fs) <= t.  The division is monotone in its
  // numerator, so "location <= t" is "half-sum <= hmax" with hmax the largest double whose quotient by fs
  // rounds to at most t: a few divisions per query instead of one per probe, same decisions bit for bit.
  bool exact = false;
  const double hmax = zc_hmax(fs, t, exact);
  int lo = 0, hi = n;
  if (exact) {
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((e[mid] + e[mid + 1]) / 2.0 <= hmax) lo = mid + 1; else hi = mid;
    }
  } else {                                     // not pinned down within 4 ulps (t == 0): the literal form
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const double loc = (e[mid] + e[mid + 1]) / 2.0 / fs;
      if (loc <= t) lo = mid + 1; else hi = mid;
    }
  }
  const int k = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
  const double x0 = (e[k - 1] + e[k]) / 2.0 / fs, x1 = (e[k] + e[k + 1]) / 2.0 / fs;
  const double y0 = fs / (e[k] - e[k - 1]), y1 = fs / (e[k + 1] - e[k]);
  const double h = x1 - x0;
  const double sfrac = (t - x0) / h;
  return y0 + sfrac * (y1 - y0);
}


// upper_bound index alone (the first part of zc_track), literal comparisons
__device__ __forceinline__ int zc_upper(const double* __restrict__ e, int n, double fs, double t) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const double loc = (e[mid] + e[mid + 1]) / 2.0 / fs;
    if (loc <= t) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ---- the ordered lists read where they were staged (round 5) ----------------------------------------------------
// The ordered list of fine edges of one (signal, kind), read WHERE THE FILTER KERNEL STAGED IT: tile t of the signal left
// its events of this kind at slot + t * kZcSlot, and the scanned counts (zc_scan_tiles) say which list indices a tile
// holds -- off[4 t + kind] <= g < off[4 (t + 1) + kind].  Until round 5 a compaction kernel (zc_compact_signal) copied the
// staged events into one contiguous list per (signal, kind) first -- 0.58 GB written and read again per pass of Harvest's
// configs[2] (0.31 ms) -- for the sake of `e[g]`.  Harvest's consumer is a walk (consecutive indices), so a cursor on the
// current tile serves it with a compare per access: 5.63 -> 5.36 ms per pass.  DIO's consumer brackets every frame on its
// own (a thread per frame): the same lists through slots made dio_candidate_kernel 0.13 -> 0.24 ms for the 0.07 ms of
// compaction saved (profiles/r05_v_dio_slots_ab.txt), so DIO keeps its compaction.
struct SlotList {
  const double* slot;             // the kind's slot rows of the signal
  const int* off;                 // scanned tile offsets of the signal, layout [tile][kind], entry nt = the total
  int kind, nt;
  int tile, tb, te;               // cursor: list indices [tb, te) live in `tile`
  __device__ __forceinline__ void open(const double* slot_, const int* off_, int kind_, int nt_, int tile_) {
    slot = slot_; off = off_; kind = kind_; nt = nt_;
    tile = tile_ < 0 ? 0 : (tile_ > nt_ - 1 ? nt_ - 1 : tile_);
    tb = off[4 * tile + kind];
    te = off[4 * (tile + 1) + kind];
  }
  __device__ __forceinline__ double at(int g) {              // 0 <= g < the list's length
    while (g >= te && tile + 1 < nt) {                        // empty tiles are stepped over
      ++tile;
      tb = te;
      te = off[4 * (tile + 1) + kind];
    }
    while (g < tb && tile > 0) {
      --tile;
      te = tb;
      tb = off[4 * tile + kind];
    }
    return slot[(int64_t)tile * kZcSlot + (g - tb)];
  }
};
// histc's count of knots at or before t -- knot j = (e[j] + e[j+1]) / 2 / fs, j < n -- over a SlotList, with `le(a, b)`
// the caller's form of "knot of edges a, b <= t" (zc_upper's literal division, or zc_track's half-sum against hmax).
// A knot lies between its two edges and an edge lies in its tile's sample range, so the count falls inside the tiles
// around sample t fs: the bisection runs over those (a handful of probes near the cursor) and its two ends are CHECKED
// against their neighbours; only a signal with empty tiles there takes the bisection over the whole list.
template <class LE>
__device__ __forceinline__ int slot_upper(SlotList& e, int n, double fs, double t, int step, LE le) {
  const int t0 = (int)(t * fs) / step;
  const int ta = imax(0, imin(e.nt - 1, t0 - 1)), tz = imax(0, imin(e.nt, t0 + 2));
  int lo = imin(n, e.off[4 * ta + e.kind]), hi = imin(n, e.off[4 * tz + e.kind]);
  const int lo0 = lo, hi0 = hi;
  auto knot_le = [&](int j) {
    const double a = e.at(j), b = e.at(j + 1);
    return le(a, b);
  };
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (knot_le(mid)) lo = mid + 1; else hi = mid;
  }
  const bool ok_lo = lo > lo0 || lo0 == 0 || knot_le(lo0 - 1);
  const bool ok_hi = lo < hi0 || hi0 == n || !knot_le(hi0);
  if (ok_lo && ok_hi) return lo;
  lo = 0;
  hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (knot_le(mid)) lo = mid + 1; else hi = mid;
  }
  return lo;
}
}  // namespace wm

// dio.hip -- DIO F0 estimation for a batch of utterances.
//
// Replaces Dio / DioGeneralBody and everything below it
// (externs/WORLD_v2/src/dio.cpp:40-647).  The reference filters by multiplying
// whole-utterance spectra (16 FFTs of 2^16..2^18 points); because every filter is
// a short FIR (641-tap low-cut, 40..320-tap Nuttall low-pass) and
// fft_size >= y_length + filter length (dio.cpp:592-593), the same circular
// convolution is evaluated here as tiled time-domain FIRs with explicit
// wrap-around indexing (so even the aliasing corner case fft_size - y_length < 480
// is reproduced):
//
//   dio_mean_kernel      mean over y_length = N+1 samples          dio.cpp:74-79
//   dio_lowcut_kernel    y (*) low-cut filter, circular             dio.cpp:40-53, 85-101
//   dio_band_kernel      per (utterance, band): Nuttall FIR, the four zero-crossing
//                        event lists (ordered compaction)           dio.cpp:296-435
//   dio_candidate_kernel interp1 of the four interval tracks, mean/std score  dio.cpp:441-508, 562-567
//   dio_fix_kernel       best band + FixStep1..4                    dio.cpp:112-289
#include <math.h>
#include <string.h>

#include "batch.hpp"
#include "common.hpp"
#include "decimate.hpp"
#include "fftconv.hpp"
#include "zcfilter.hpp"

namespace wm {

constexpr int kMaxBands = 32;

struct DioMeta {
  int nb;                       // number_of_bands
  int hal[kMaxBands];           // half_average_length per band (dio.cpp:532)
  int win_off[kMaxBands];       // offset of the band's Nuttall window in d_win
  double boundary[kMaxBands];   // boundary_f0_list
  int cut;                      // cutoff_in_sample (dio.cpp:86)
  int pad;                      // 2 * hal[0]: how far the low-cut output is needed outside [0, y_len)
  int ratio;                    // decimation ratio (only 1 is implemented on device)
  double afs;                   // actual_fs
  int step;                     // outputs per tile of the band filters (tiles overlap by 2 samples of look-ahead)
  int band_conv;                // block size of the FFT convolution of the band filters (fftconv.hpp); 0: direct FIR
  int lc_conv;                  // same for the low-cut filter
};
__host__ __device__ inline int dio_tiles(int ylen, int step) { return (ylen + step - 1) / step; }
constexpr int kDioConvC = 28;   // samples per lane of a block's event passes: a block's step is at most 64 * 28

// src is x itself (speed 1: y_length = N + 1, the extra sample is zero) or the decimated signal
// (speed > 1: all y_length samples materialised, zeros beyond decimate's output).
// Two stages with a fixed summation tree (deterministic, no atomics): kMeanTiles partial sums per
// utterance, then one thread per utterance adds them in order.
constexpr int kMeanTiles = 32;
__global__ __launch_bounds__(256) void dio_mean_partial_kernel(const double* __restrict__ src,
                                                               const int64_t* __restrict__ src_off,
                                                               const int* __restrict__ src_len,
                                                               double* __restrict__ part) {
  __shared__ double wsum[4];
  const int u = blockIdx.y, t = blockIdx.x;
  const double* xu = src + src_off[u];
  const int n = src_len[u];
  const int chunk = (n + kMeanTiles - 1) / kMeanTiles;
  const int lo = t * chunk, hi = imin(n, lo + chunk);
  double s = 0.0;
  for (int i = lo + threadIdx.x; i < hi; i += 256) s += xu[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[u * kMeanTiles + t] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}
__global__ __launch_bounds__(256) void dio_mean_kernel(const double* __restrict__ part, const int* __restrict__ ylen_a,
                                                       int n_utt, double* __restrict__ mean) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= n_utt) return;
  double s = 0.0;
  for (int t = 0; t < kMeanTiles; ++t) s += part[u * kMeanTiles + t];
  mean[u] = s / ylen_a[u];                                        // dio.cpp:74-77
}

// circular, zero-padded, mean-removed signal of the reference (dio.cpp:63-79)
__device__ __forceinline__ double dio_y(const double* __restrict__ xu, int n, int ylen, int fftn, double mean,
                                        int i) {
  i = i < 0 ? i + fftn : (i >= fftn ? i - fftn : i);
  // the load is unconditional (clamped address) and the case is applied to the VALUE: with the load behind a branch
  // the elements of a block are fetched one dependent trip at a time (32 pairs per lane at block 4096: the low-cut
  // kernel took 4.4 ms at 48 kHz)
  const double xv = xu[imax(0, imin(n - 1, i))];              // n >= 1: CreateBatch refuses empty utterances
  return i < n ? xv - mean : (i < ylen ? 0.0 - mean : 0.0);
}

// z[m] = sum_lag h(lag) y[(m - lag) mod fft], m in [-pad, ylen + pad); stored at z[m + pad].
// Tiles of 2048 outputs, 8 per thread, on the register-rotation FIR core of zcfilter.hpp
// (h is symmetric in the lag, so h[j] with j = lag + cut serves as the tap sequence directly).
template <int STRIDE>
__global__ __launch_bounds__(256) void dio_lowcut_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int* __restrict__ ylen_a, const double* __restrict__ mean, const int* __restrict__ fft_sizes,
    const double* __restrict__ lowcut, DioMeta meta, const int64_t* __restrict__ z_off, double* __restrict__ z) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int u = blockIdx.y;
  const int n = x_len[u], ylen = ylen_a[u], fftn = fft_sizes[u];
  const int cut = meta.cut, ntap = 2 * cut + 1;
  const int total = ylen + 2 * meta.pad;
  const int m0 = blockIdx.x * kBandTile;            // tile start in z-storage coordinates
  if (m0 >= total) return;
  const int ntp = zc_pad16(ntap);
  const int zspan = kBandTile + ntp + 8;
  double* zt = lds;                                 // [8 * STRIDE] transposed tile
  double* w = lds + kBandK * STRIDE;                // [ntp] taps: w[k] = h(lag = k - cut)
  const double* xu = x + x_off[u];
  const double mu = mean[u];
  for (int j = threadIdx.x; j < ntp; j += 256) w[j] = j < ntap ? lowcut[j] : 0.0;
  // output m (storage index m0 + 8t + q) = sum_k w[k] y[(m0 + 8t + q) - pad - (k - cut)]: tile element e is
  // y[m0 - pad + cut - (ntp - 1) - 8 + e]
  const int ybase = m0 - meta.pad + cut - (ntp - 1) - 8;
  for (int e = threadIdx.x; e < zspan; e += 256)
    zt[(e & 7) * STRIDE + (e >> 3)] = dio_y(xu, n, ylen, fftn, mu, ybase + e);
  __syncthreads();
  double acc[kBandK];
  fir_tile_accumulate<STRIDE>(zt, w, ntp, threadIdx.x, acc);
  double* zu = z + z_off[u];
#pragma unroll
  for (int q = 0; q < kBandK; ++q) {
    const int m = m0 + threadIdx.x * kBandK + q;
    if (m < total) zu[m] = acc[q];
  }
}

// One workgroup per (tile, band, utterance): Nuttall FIR over the low-cut signal + the four
// ZeroCrossingEngine passes (dio.cpp:296-435), see zcfilter.hpp.
// events / staging layout per (utt, band): 4 lists of `cap` fine edges / 4 x tiles x kZcSlot slots.
template <int STRIDE>
__global__ __launch_bounds__(256) void dio_band_kernel(
    const int* __restrict__ ylen_a, const int64_t* __restrict__ z_off, const double* __restrict__ z,
    const double* __restrict__ win, DioMeta meta, int tiles_max, int* __restrict__ tile_cnt,
    const int64_t* __restrict__ slot_off, double* __restrict__ slots) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int u = blockIdx.z, band = blockIdx.y, tile = blockIdx.x;
  const int ylen = ylen_a[u];
  const int nt = dio_tiles(ylen, meta.step);
  if (tile >= nt) return;
  const int hal = meta.hal[band];
  const int64_t slot_cap = (int64_t)nt * kZcSlot;
  // filtered[n] = sum_{k < 4 hal} w[k] z[n + 2 hal - k]  (dio.cpp:310-337)
  filter_tile_events<STRIDE>(z + z_off[u] + meta.pad, -meta.pad, ylen + meta.pad, ylen, win + meta.win_off[band],
                             4 * hal, 2 * hal, tile,
                             tile_cnt + (((int64_t)u * meta.nb + band) * (tiles_max + 1) + tile) * 4,
                             slots + slot_off[u] + (int64_t)band * 4 * slot_cap, slot_cap, lds);
}

// ---- the same two stages by block FFT convolution (fftconv.hpp): one wavefront per block ---------------------
// z storage index m0 + i <-> signal index m0 + i - pad; z[m] = sum_k h[k] y[(m - pad) + cut - k], k < 2 cut + 1.
template <int B>
__global__ __launch_bounds__(64) void dio_lowcut_fft_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int* __restrict__ ylen_a, const double* __restrict__ mean, const int* __restrict__ fft_sizes,
    const cpx* __restrict__ H, DioMeta meta, const int64_t* __restrict__ z_off, double* __restrict__ z) {
  constexpr int N = ConvCfg<B>::N, M = ConvCfg<B>::M;
  __shared__ __attribute__((aligned(16))) double smem[ConvCfg<B>::kImg];
  cpx* img = reinterpret_cast<cpx*>(smem);
  const int u = blockIdx.y, lane = threadIdx.x;
  const int n = x_len[u], ylen = ylen_a[u], fftn = fft_sizes[u];
  const int ntap = 2 * meta.cut + 1, V = B - ntap + 1;
  const int total = ylen + 2 * meta.pad;
  const int m0 = blockIdx.x * V;
  if (m0 >= total) return;
  FftTw<N> tw;
  tw.init(lane);
  const double* xu = x + x_off[u];
  const double mu = mean[u];
  const int base = m0 - meta.pad + meta.cut - (ntap - 1);        // block element i is y[base + i]
  cpx v[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int i0 = base + 2 * (lane + 64 * m);
    v[m] = make_double2(dio_y(xu, n, ylen, fftn, mu, i0), dio_y(xu, n, ylen, fftn, mu, i0 + 1));
  }
  ConvSpec<B> zr;
  conv_forward<B>(v, img, tw, lane, zr);
  conv_apply<B>(zr, H, img, tw, lane, v);
  double* zu = z + z_off[u];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int j = 2 * (lane + 64 * m);
    const int a = m0 + j - (ntap - 1);                             // storage index of out[j]
    if (j >= ntap - 1 && a < total) zu[a] = v[m].x;
    if (j + 1 >= ntap - 1 && a + 1 < total) zu[a + 1] = v[m].y;
  }
}

// One wavefront per (block, utterance): the block's spectrum once, then per band the product with the band's
// (delayed) Nuttall spectrum, the inverse transform and the four zero-crossing passes.  All bands share
// ntap0 = 4 hal[0], bias0 = 2 hal[0]: band b's window is delayed by 2 (hal[0] - hal[b]) samples.
template <int B>
__global__ __launch_bounds__(64, B > 2048 ? 1 : 2) void dio_band_fft_kernel(
    const int* __restrict__ ylen_a, const int64_t* __restrict__ z_off, const double* __restrict__ z,
    const cpx* __restrict__ H, DioMeta meta, int tiles_max, int* __restrict__ tile_cnt,
    const int64_t* __restrict__ slot_off, double* __restrict__ slots) {
  constexpr int N = ConvCfg<B>::N, M = ConvCfg<B>::M;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  cpx* img = reinterpret_cast<cpx*>(lds);
  double* s = lds;                                               // the filtered block, after the inverse transform
  unsigned short* lists = ConvEvCfg<B, kDioConvC>::lists(lds);
  const int u = blockIdx.y, tile = blockIdx.x, lane_k = threadIdx.x, lane = lane_k;
  const int ylen = ylen_a[u];
  const int nt = dio_tiles(ylen, meta.step);
  if (tile >= nt) return;
  FftTw<N> tw;
  tw.init(lane);
  const int ntap0 = 4 * meta.hal[0], bias0 = 2 * meta.hal[0];
  const int n0 = tile * meta.step;
  const int base = n0 + bias0 - (ntap0 - 1);                     // block element i is sig[base + i]
  const double* zu = z + z_off[u] + meta.pad;                    // sig[m], valid for m in [-pad, ylen + pad)
  const int lo = -meta.pad, hi = ylen + meta.pad;
  cpx v[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int i0 = base + 2 * (lane + 64 * m), i1 = i0 + 1;
    const double a = zu[imin(hi - 1, imax(lo, i0))], c = zu[imin(hi - 1, imax(lo, i1))];
    v[m] = make_double2(i0 >= lo && i0 < hi ? a : 0.0, i1 >= lo && i1 < hi ? c : 0.0);
  }
  ConvSpec<B> zr;
  conv_forward<B>(v, img, tw, lane, zr);
  const int64_t slot_cap = (int64_t)nt * kZcSlot;
#pragma unroll 1
  for (int band = 0; band < meta.nb; ++band) {
    const int lane = opaque_lane(lane_k);                        // nothing lane-derived is carried across the bands
    tw.fence();
    conv_apply<B>(zr, H + (int64_t)band * (N + 1), img, tw, lane, v);
    wave_sync();
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int j = 2 * (lane + 64 * m) - (ntap0 - 1);
      if (j >= 0) s[j] = v[m].x;
      if (j + 1 >= 0) s[j + 1] = v[m].y;
    }
    wave_sync();
    conv_block_events<kDioConvC>(s, n0, meta.step, ylen, tile, lists, ConvEvCfg<B, kDioConvC>::kListCap,
                      tile_cnt + (((int64_t)u * meta.nb + band) * (tiles_max + 1) + tile) * 4,
                      slots + slot_off[u] + (int64_t)band * 4 * slot_cap, slot_cap, lane);
  }
}

__global__ __launch_bounds__(64) void dio_band_scan_kernel(const int* __restrict__ ylen_a, DioMeta meta,
                                                           int tiles_max, int* __restrict__ tile_cnt,
                                                           int* __restrict__ ev_cnt) {
  const int u = blockIdx.y, band = blockIdx.x;
  const int ylen = ylen_a[u];
  zc_scan_tiles(tile_cnt + ((int64_t)u * meta.nb + band) * (tiles_max + 1) * 4, dio_tiles(ylen, meta.step), ylen / 2 + 2,
                ev_cnt + ((int64_t)u * meta.nb + band) * 4, threadIdx.x);
}

__global__ __launch_bounds__(256) void dio_band_compact_kernel(
    const int* __restrict__ ylen_a, DioMeta meta, int tiles_max, const int* __restrict__ tile_cnt,
    const int64_t* __restrict__ slot_off, const double* __restrict__ slots, const int64_t* __restrict__ ev_off,
    double* __restrict__ events) {
  __shared__ int lds_off[4 * (kZcCompactTiles + 1)];
  const int u = blockIdx.y, band = blockIdx.x;
  const int ylen = ylen_a[u];
  const int nt = dio_tiles(ylen, meta.step);
  const int cap = ylen / 2 + 2;
  const int64_t slot_cap = (int64_t)nt * kZcSlot;
  zc_compact_signal(slots + slot_off[u] + (int64_t)band * 4 * slot_cap, slot_cap, nt,
                    tile_cnt + ((int64_t)u * meta.nb + band) * (tiles_max + 1) * 4,
                    events + ev_off[u] + (int64_t)band * 4 * cap, cap, lds_off);
}

__global__ __launch_bounds__(256) void dio_candidate_kernel(
    const int* __restrict__ ylen_a, const int64_t* __restrict__ f_off, const int* __restrict__ frame_utt,
    double frame_period, DioMeta meta, double f0_floor, double f0_ceil,
    const int64_t* __restrict__ ev_off, const double* __restrict__ events, const int* __restrict__ ev_cnt,
    int64_t total_frames, double* __restrict__ cand, double* __restrict__ score) {
  const int band = blockIdx.y;
  const int64_t frame = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (frame >= total_frames) return;
  const int u = frame_utt[frame];
  const int ylen = ylen_a[u];
  const int cap = ylen / 2 + 2;
  const int* cnt = ev_cnt + ((int64_t)u * meta.nb + band) * 4;
  const double* ev = events + ev_off[u] + (int64_t)band * 4 * cap;
  int nint[4];
  bool ok = true;
#pragma unroll
  for (int ty = 0; ty < 4; ++ty) {
    nint[ty] = cnt[ty] < 2 ? 0 : cnt[ty] - 1;       // ZeroCrossingEngine returns count - 1 (dio.cpp:372-392)
    ok = ok && nint[ty] > 2;                        // CheckEvent(n - 2) (dio.cpp:475-478)
  }
  double c = 0.0, sc = kBig;
  if (ok) {
    const double t = (int)(frame - f_off[u]) * frame_period / 1000.0;   // temporal_positions, dio.cpp:608-609
    double v[4];
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) v[ty] = zc_track(ev + (int64_t)ty * cap, nint[ty], meta.afs, t);
    c = (v[0] + v[1] + v[2] + v[3]) / 4.0;          // dio.cpp:446-457
    sc = sqrt(((v[0] - c) * (v[0] - c) + (v[1] - c) * (v[1] - c) + (v[2] - c) * (v[2] - c) +
               (v[3] - c) * (v[3] - c)) / 3.0);
    const double bf = meta.boundary[band];
    if (c > bf || c < bf / 2.0 || c > f0_ceil || c < f0_floor) { c = 0.0; sc = kBig; }   // :459-463
  }
  cand[(int64_t)band * total_frames + frame] = c;
  score[(int64_t)band * total_frames + frame] = sc / (c + kSafe);                        // :564-565
}

// SelectBestF0 (dio.cpp:190-209) with the bands spread over the lanes of a wavefront: lane b < nb
// holds candidate b of the frame.  The reference keeps the FIRST band with a strictly smaller error,
// i.e. the lowest band among the minima.  cur / past / the result are wave-uniform.
__device__ __forceinline__ double dio_select_wave(double cur, double past, double cv, int nb, int lane,
                                                  double allowed) {
  const double ref = (cur * 3.0 - past) / 2.0;
  const double err = lane < nb ? fabs(ref - cv) : HUGE_VAL;
  // exact minimum by DPP steps on the bit patterns (a chain of twelve ds_bpermute round trips before: this function is
  // the body of a loop that runs once per frame of a voiced section, on one wavefront per utterance)
  const double m = wave_min_nonneg(err);
  const unsigned long long at = __ballot(err == m);
  const int who = at ? __ffsll((long long)at) - 1 : 0;
  const double best = readlane_d(cv, __builtin_amdgcn_readfirstlane(who));
  if (fabs(1.0 - best / ref) > allowed) return 0.0;
  return best;
}

// GetBestF0Contour (dio.cpp:112-126) + FixF0Contour / FixStep1-4 (:132-289), one workgroup per
// utterance.  Steps 1-2 and the copies are per-frame parallel.  Steps 3-4 are sequential along time
// by construction (each extension step feeds the next); they run on wavefront 0 with everything that
// does not depend on the chain taken off it: the rising / falling edges of step 2's contour are
// compacted into LDS lists up front, the chain state (current and previous f0) lives in registers,
// the candidates of the next 16 frames are fetched in one round trip with the bands spread over
// lanes, and results are stored without waiting.
// ws holds 3 work arrays of the utterance's frame count.  The edge lists (2 * (max_nf / 2 + 2) ints) live in
// dynamic LDS while they fit (EDGES_IN_LDS; up to 6 k frames per utterance = 48 KB) and in a per-utterance slice
// of global memory beyond that: they are read once per voiced section, so where they live does not matter for
// speed, only for the launch to be possible at all (a 40 s utterance at a 1 ms hop has 40 k frames).
constexpr int kFixBlk = 16;
template <bool EDGES_IN_LDS>
__global__ __launch_bounds__(256) void dio_fix_kernel(const int64_t* __restrict__ f_off,
                                                      const double* __restrict__ cand,
                                                      const double* __restrict__ score, int nb,
                                                      double frame_period, double f0_floor, double allowed,
                                                      int64_t total_frames, int edge_cap, double* __restrict__ ws,
                                                      int* __restrict__ edges_global,
                                                      double* __restrict__ tpos, double* __restrict__ f0) {
  extern __shared__ int edges_lds[];                  // [2][edge_cap]: falling (negative), rising (positive)
  __shared__ int n_edges[2];
  const int u = blockIdx.x;
  int* edges = EDGES_IN_LDS ? edges_lds : edges_global + (int64_t)u * 2 * edge_cap;
  const int lane = threadIdx.x & 63;
  const int64_t base = f_off[u];
  const int nf = (int)(f_off[u + 1] - base);
  double* best = ws + base;
  double* s1 = ws + total_frames + base;
  double* s2 = ws + 2 * total_frames + base;
  double* out = f0 + base;
  for (int i = threadIdx.x; i < nf; i += 256) {
    tpos[base + i] = i * frame_period / 1000.0;                  // dio.cpp:608-609
    double sv = score[base + i], bv = cand[base + i];
    for (int b = 1; b < nb; ++b) {
      const double s = score[(int64_t)b * total_frames + base + i];
      if (sv > s) { sv = s; bv = cand[(int64_t)b * total_frames + base + i]; }
    }
    best[i] = bv;
    out[i] = 0.0;
  }
  const int vrm = (int)(0.5 + 1000.0 / frame_period / f0_floor) * 2 + 1;   // dio.cpp:263-264
  if (nf <= vrm) return;       // reference leaves f0 unwritten here (dio.cpp:266); we leave zeros
  __syncthreads();
  // step 1 (dio.cpp:132-150): f0_base is best with vrm frames zeroed at both ends
  for (int i = threadIdx.x; i < nf; i += 256) {
    double r = 0.0;
    if (i >= vrm) {
      const double bi = (i < nf - vrm) ? best[i] : 0.0;
      const double bp = (i - 1 >= vrm && i - 1 < nf - vrm) ? best[i - 1] : 0.0;
      r = fabs((bi - bp) / (kSafe + bi)) < allowed ? bi : 0.0;
    }
    s1[i] = r;
  }
  __syncthreads();
  // step 2 (dio.cpp:156-169); s1 becomes the copy step 3 works on
  const int c = (vrm - 1) / 2;
  for (int i = threadIdx.x; i < nf; i += 256) {
    double r = s1[i];
    if (i >= c && i < nf - c) {
      for (int j = -c; j <= c; ++j)
        if (s1[i + j] == 0) { r = 0.0; break; }
    }
    s2[i] = r;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nf; i += 256) s1[i] = s2[i];
  // edge lists of s2, in ascending order: falling at i (s2[i] == 0, s2[i-1] != 0), rising at i
  // (s2[i-1] == 0, s2[i] != 0)
  if (threadIdx.x < 64) {
    int cnt0 = 0, cnt1 = 0;
    for (int i0 = 1; i0 < nf; i0 += 64) {
      const int i = i0 + lane;
      const double a = i < nf ? s2[i - 1] : 0.0, bq = i < nf ? s2[i] : 0.0;
      const bool fall = i < nf && bq == 0 && a != 0;
      const bool rise = i < nf && a == 0 && bq != 0;
      const unsigned long long bf = __ballot(fall), br = __ballot(rise);
      const unsigned long long below = (1ull << lane) - 1ull;
      if (fall) edges[cnt0 + __popcll(bf & below)] = i;
      if (rise) edges[edge_cap + cnt1 + __popcll(br & below)] = i;
      cnt0 += __popcll(bf);
      cnt1 += __popcll(br);
    }
    if (lane == 0) { n_edges[0] = cnt0; n_edges[1] = cnt1; }
  }
  __syncthreads();
  const int n_fall = n_edges[0], n_rise = n_edges[1];
  const double* cu = cand + base + (int64_t)(lane < nb ? lane : 0) * total_frames;   // this lane's band
  // step 3 (dio.cpp:215-231): forward from each falling edge; result in s1
  if (threadIdx.x < 64) {
    for (int k = 0; k < n_fall; ++k) {
      const int start = edges[k] - 1;                           // negative_index
      const int limit = k + 1 < n_fall ? edges[k + 1] - 1 : nf - 1;
      __threadfence_block();                                    // an earlier extension may have written these
      double cur = s1[start], past = s1[start - 1];
      int j = start;
      bool stop = false;
      while (j < limit && !stop) {
        double blk[kFixBlk];
#pragma unroll
        for (int r = 0; r < kFixBlk; ++r) blk[r] = cu[imin(nf - 1, j + 1 + r)];
#pragma unroll
        for (int r = 0; r < kFixBlk; ++r) {
          if (!stop && j < limit) {
            const double sel = dio_select_wave(cur, past, blk[r], nb, lane, allowed);
            if (lane == 0) s1[j + 1] = sel;
            stop = sel == 0;
            past = cur;
            cur = sel;
            ++j;
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nf; i += 256) out[i] = s1[i];
  __syncthreads();
  // step 4 (dio.cpp:237-253): backward from each rising edge, last edge first
  if (threadIdx.x < 64) {
    for (int k = n_rise - 1; k >= 0; --k) {
      const int start = edges[edge_cap + k];                    // positive_index
      const int limit = k > 0 ? edges[edge_cap + k - 1] : 1;
      __threadfence_block();
      double cur = out[start], past = out[start + 1];
      int j = start;
      bool stop = false;
      while (j > limit && !stop) {
        double blk[kFixBlk];
#pragma unroll
        for (int r = 0; r < kFixBlk; ++r) blk[r] = cu[imax(0, j - 1 - r)];
#pragma unroll
        for (int r = 0; r < kFixBlk; ++r) {
          if (!stop && j > limit) {
            const double sel = dio_select_wave(cur, past, blk[r], nb, lane, allowed);
            if (lane == 0) out[j - 1] = sel;
            stop = sel == 0;
            past = cur;
            cur = sel;
            --j;
          }
        }
      }
    }
  }
}

// ---- host side ---------------------------------------------------------------
static int suitable_fft_size(int sample) {   // common.cpp:51-54
  return (int)pow(2.0, (int)(log((double)sample) / kLog2) + 1.0);
}

struct DioHost {
  DioMeta meta;
  std::vector<double> lowcut, win;
};

// frees every device buffer dio_setup() allocates and clears the pointers (the filters belong to the context)
static void dio_release(Batch& b) {
  void** ptrs[] = {(void**)&b.d_dio_desc, (void**)&b.d_dio_mean, (void**)&b.d_dio_mean_part, (void**)&b.d_dio_y,
                   (void**)&b.d_dio_tmp, (void**)&b.d_dio_z, (void**)&b.d_dio_events, (void**)&b.d_dio_ev_cnt,
                   (void**)&b.d_dio_tile_cnt, (void**)&b.d_dio_slots, (void**)&b.d_dio_cand, (void**)&b.d_dio_score,
                   (void**)&b.d_dio_ws};
  for (void** p : ptrs) {
    if (*p) dev_free(*p);
    *p = nullptr;
  }
  b.d_dio_lowcut = b.d_dio_win = nullptr;
  b.d_dio_H = nullptr;
  b.d_dio_fft = b.d_dio_ylen = nullptr;
  b.d_dio_yoff = b.d_dio_toff = b.d_dio_z_off = b.d_dio_ev_off = b.d_dio_slot_off = nullptr;
}

static int dio_setup(Batch& b) {
  if (b.dio_ready) return WM_OK;
  const WorldMi355Params& p = b.p;
  DioHost* H = new DioHost();
  DioMeta& m = H->meta;
  m.nb = 1 + (int)(log(p.f0_ceil / p.f0_floor) / kLog2 * p.channels_in_octave);      // dio.cpp:582-583
  if (m.nb < 1 || m.nb > kMaxBands) { delete H; return WM_ERR_UNSUPPORTED; }
  m.ratio = imax(imin(p.speed, 12), 1);                                                // :589
  m.afs = (double)p.fs / m.ratio;
  int woff = 0;
  for (int i = 0; i < m.nb; ++i) {
    m.boundary[i] = p.f0_floor * pow(2.0, (i + 1) / p.channels_in_octave);             // :585-586
    m.hal[i] = matlab_round(m.afs / m.boundary[i] / 2.0);                              // :532
    m.win_off[i] = woff;
    woff += 4 * m.hal[i];
  }
  m.pad = 2 * m.hal[0];
  m.cut = matlab_round(m.afs / 50.0);                                                  // :86
  // block FFT convolution where the filters fit a block comfortably (fftconv.hpp), the direct FIR otherwise
  // (blocks of 4096 carry the filters of the rates above 51 kHz: 1 912 band taps and 3 841 low-cut taps at 96 kHz
  // and speed 1 -- the low-cut then yields only 256 outputs per block: it works, at a sixteenth of the efficiency)
  m.band_conv = 4 * m.hal[0] <= 1024 ? 2048 : (4 * m.hal[0] <= 2560 ? 4096 : 0);
  m.lc_conv = 2 * m.cut + 1 <= 1024 ? 2048 : (2 * m.cut + 1 <= 4000 ? 4096 : 0);
  m.step = m.band_conv ? imin(m.band_conv - 4 * m.hal[0] + 1 - 2, 64 * kDioConvC) : kZcStep;
  // Nuttall low-pass windows (dio.cpp:301, common.cpp:113-121)
  H->win.resize((size_t)woff);
  for (int i = 0; i < m.nb; ++i) {
    const int n = 4 * m.hal[i];
    for (int j = 0; j < n; ++j) {
      double tmp = j / (n - 1.0);
      H->win[(size_t)(m.win_off[i] + j)] = 0.355768 - 0.487396 * cos(2.0 * kPi * tmp) +
                                           0.144232 * cos(4.0 * kPi * tmp) - 0.012604 * cos(6.0 * kPi * tmp);
    }
  }
  // low-cut filter as a function of lag in [-cut, cut] (DesignLowCutFilter, dio.cpp:40-53)
  {
    const int N = 2 * m.cut + 1;
    std::vector<double> lc((size_t)N);
    double sum = 0.0;
    for (int i = 1; i <= N; ++i) {
      lc[(size_t)(i - 1)] = 0.5 - 0.5 * cos(i * 2.0 * kPi / (N + 1));
      sum += lc[(size_t)(i - 1)];
    }
    H->lowcut.resize((size_t)N);
    for (int i = 0; i < N; ++i) H->lowcut[(size_t)i] = -lc[(size_t)i] / sum;   // lag = i - cut
    H->lowcut[(size_t)m.cut] += 1.0;
  }
  // per-utterance FFT size of the reference's circular convolution (dio.cpp:590-593)
  std::vector<int> fftn((size_t)b.n_utt), ylens((size_t)b.n_utt);
  std::vector<int64_t> yoff((size_t)b.n_utt + 1, 0), toff((size_t)b.n_utt + 1, 0);
  b.dio_z_off.assign((size_t)b.n_utt + 1, 0);
  b.dio_ev_off.assign((size_t)b.n_utt + 1, 0);
  for (int u = 0; u < b.n_utt; ++u) {
    const int ylen = 1 + b.x_len[u] / m.ratio;                                          // dio.cpp:590
    ylens[(size_t)u] = ylen;
    yoff[(size_t)u + 1] = yoff[(size_t)u] + ylen;
    toff[(size_t)u + 1] = toff[(size_t)u] + b.x_len[u] + 18;
    fftn[(size_t)u] = suitable_fft_size(ylen + 4 * (int)(1.0 + m.afs / m.boundary[0] / 2.0));
    b.dio_z_off[(size_t)u + 1] = b.dio_z_off[(size_t)u] + ylen + 2 * m.pad;
    b.dio_ev_off[(size_t)u + 1] = b.dio_ev_off[(size_t)u] + (int64_t)m.nb * 4 * (ylen / 2 + 2);
  }
  int rc = WM_OK;
  // slot offsets of the staged events (needed below, part of the same upload)
  std::vector<int64_t> soff((size_t)b.n_utt + 1, 0);
  for (int u = 0; u < b.n_utt; ++u)
    soff[(size_t)u + 1] = soff[(size_t)u] + (int64_t)m.nb * 4 * dio_tiles(ylens[(size_t)u], m.step) * kZcSlot;
  if (m.ratio > 1) b.dio_tot_y = yoff[(size_t)b.n_utt];
  // The per-utterance tables in ONE block and one upload (a batch per utterance length is what the drop-in API makes:
  // eight synchronous copies of a few bytes each were 0.1 ms of every Dio call).  Sections are 8-byte aligned.
  {
    const size_t n = (size_t)b.n_utt, n1 = n + 1;
    const size_t ints = (2 * n + 1) & ~(size_t)1;                     // fft, ylen (padded to a multiple of two ints)
    std::vector<int64_t> blob(ints / 2 + 5 * n1);
    int* bi = (int*)blob.data();
    for (size_t u = 0; u < n; ++u) { bi[u] = fftn[u]; bi[n + u] = ylens[u]; }
    int64_t* bl = blob.data() + ints / 2;
    memcpy(bl + 0 * n1, yoff.data(), 8 * n1);
    memcpy(bl + 1 * n1, toff.data(), 8 * n1);
    memcpy(bl + 2 * n1, b.dio_z_off.data(), 8 * n1);
    memcpy(bl + 3 * n1, b.dio_ev_off.data(), 8 * n1);
    memcpy(bl + 4 * n1, soff.data(), 8 * n1);
    rc = wm_check(dev_alloc(&b.d_dio_desc, 8 * blob.size()));
    if (!rc) rc = wm_check(hipMemcpy(b.d_dio_desc, blob.data(), 8 * blob.size(), hipMemcpyHostToDevice));
    if (!rc) {
      int* di = (int*)b.d_dio_desc;
      int64_t* dl = (int64_t*)b.d_dio_desc + ints / 2;
      b.d_dio_fft = di;
      b.d_dio_ylen = di + n;
      b.d_dio_yoff = m.ratio > 1 ? dl + 0 * n1 : nullptr;
      b.d_dio_toff = m.ratio > 1 ? dl + 1 * n1 : nullptr;
      b.d_dio_z_off = dl + 2 * n1;
      b.d_dio_ev_off = dl + 3 * n1;
      b.d_dio_slot_off = dl + 4 * n1;
    }
  }
  auto al = [&](void** dst, size_t bytes) {
    if (rc) return;
    rc = wm_check(dev_alloc(dst, bytes ? bytes : 8));
  };
  al((void**)&b.d_dio_mean, sizeof(double) * (size_t)b.n_utt);
  al((void**)&b.d_dio_mean_part, sizeof(double) * (size_t)b.n_utt * kMeanTiles);
  if (m.ratio > 1) {
    al((void**)&b.d_dio_y, sizeof(double) * (size_t)yoff[(size_t)b.n_utt]);
    al((void**)&b.d_dio_tmp, sizeof(double) * (size_t)toff[(size_t)b.n_utt]);
  }
  al((void**)&b.d_dio_z, sizeof(double) * (size_t)b.dio_z_off[(size_t)b.n_utt]);
  al((void**)&b.d_dio_events, sizeof(double) * (size_t)b.dio_ev_off[(size_t)b.n_utt]);
  al((void**)&b.d_dio_ev_cnt, sizeof(int) * (size_t)b.n_utt * m.nb * 4);
  al((void**)&b.d_dio_tile_cnt,
     sizeof(int) * (size_t)b.n_utt * m.nb * 4 * ((size_t)dio_tiles(b.max_x_len / m.ratio + 1, m.step) + 1));
  al((void**)&b.d_dio_slots, sizeof(double) * (size_t)soff[(size_t)b.n_utt]);
  al((void**)&b.d_dio_cand, sizeof(double) * (size_t)m.nb * (size_t)b.total_f);
  al((void**)&b.d_dio_score, sizeof(double) * (size_t)m.nb * (size_t)b.total_f);
  al((void**)&b.d_dio_ws, sizeof(double) * 3 * (size_t)b.total_f);
  // The filters: the context's, by configuration.  Taps and windows as the reference designs them (above); the filter
  // spectra of the FFT-convolution path: [0] low-cut (block lc_conv), [1 .. nb] bands (block band_conv).
  if (!rc) {
    Context& c = *b.ctx;
    const Context::DioFilters* hit = nullptr;
    for (const auto& f : c.dio_filters)
      if (f.fs == p.fs && f.speed == m.ratio && f.f0_floor == p.f0_floor && f.f0_ceil == p.f0_ceil &&
          f.channels == p.channels_in_octave)
        hit = &f;
    if (!hit) {
      Context::DioFilters f{p.fs, m.ratio, p.f0_floor, p.f0_ceil, p.channels_in_octave, nullptr, nullptr, nullptr};
      auto up = [&](void** dst, const void* src, size_t bytes) {
        if (rc) return;
        rc = wm_check(dev_alloc(dst, bytes ? bytes : 8));
        if (!rc && bytes) rc = wm_check(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
      };
      up((void**)&f.d_lowcut, H->lowcut.data(), sizeof(double) * H->lowcut.size());
      up((void**)&f.d_win, H->win.data(), sizeof(double) * H->win.size());
      if (!rc && (m.band_conv || m.lc_conv)) {
        const size_t n_lc = m.lc_conv ? (size_t)m.lc_conv / 2 + 1 : 0, n_bd = m.band_conv ? (size_t)m.band_conv / 2 + 1 : 0;
        rc = wm_check(dev_alloc(&f.d_H, sizeof(cpx) * (n_lc + (size_t)m.nb * n_bd)));
        std::vector<int> desc(3 * ((size_t)m.nb + 1));
        const int n1 = m.nb + 1;
        desc[0] = 0; desc[(size_t)n1] = 2 * m.cut + 1; desc[2 * (size_t)n1] = 0;
        for (int i = 0; i < m.nb; ++i) {
          desc[(size_t)i + 1] = m.win_off[i];
          desc[(size_t)n1 + i + 1] = 4 * m.hal[i];
          desc[2 * (size_t)n1 + i + 1] = 2 * (m.hal[0] - m.hal[i]);
        }
        int* d_desc = nullptr;
        up((void**)&d_desc, desc.data(), sizeof(int) * desc.size());
        if (!rc) {
          cpx* Hs = (cpx*)f.d_H;
          hipStream_t st = c.stream;
          if (m.lc_conv == 2048)
            hipLaunchKernelGGL(conv_spectrum_kernel<2048>, dim3(1), dim3(64), 0, st, f.d_lowcut, d_desc, d_desc + n1,
                               d_desc + 2 * n1, Hs);
          else if (m.lc_conv == 4096)
            hipLaunchKernelGGL(conv_spectrum_kernel<4096>, dim3(1), dim3(64), 0, st, f.d_lowcut, d_desc, d_desc + n1,
                               d_desc + 2 * n1, Hs);
          if (m.band_conv == 4096)
            hipLaunchKernelGGL(conv_spectrum_kernel<4096>, dim3(m.nb), dim3(64), 0, st, f.d_win, d_desc + 1,
                               d_desc + n1 + 1, d_desc + 2 * n1 + 1, Hs + n_lc);
          else if (m.band_conv)
            hipLaunchKernelGGL(conv_spectrum_kernel<2048>, dim3(m.nb), dim3(64), 0, st, f.d_win, d_desc + 1,
                               d_desc + n1 + 1, d_desc + 2 * n1 + 1, Hs + n_lc);
          rc = wm_check(hipGetLastError());
          if (!rc) rc = wm_check(hipStreamSynchronize(st));
        }
        if (d_desc) dev_free(d_desc);
      }
      if (rc) {
        if (f.d_lowcut) dev_free(f.d_lowcut);
        if (f.d_win) dev_free(f.d_win);
        if (f.d_H) dev_free(f.d_H);
      } else {
        // kept for the life of the context (about 150 KB per configuration; batches hold the pointers)
        c.dio_filters.push_back(f);
        hit = &c.dio_filters.back();
      }
    }
    if (!rc) {
      b.d_dio_lowcut = hit->d_lowcut;
      b.d_dio_win = hit->d_win;
      b.d_dio_H = hit->d_H;
    }
  }
  if (rc) {
    // nothing half-built stays behind: a retry starts from scratch instead of leaking H and the buffers above
    dio_release(b);
    delete H;
    return rc;
  }
  b.dio_host = H;
  b.dio_ready = true;
  return WM_OK;
}

void dio_free_host(void* h) { delete (DioHost*)h; }

int launch_dio(Batch& b, const double* d_x, double* d_t, double* d_f0) {
  int rc = dio_setup(b);
  if (rc) return rc;
  Context& c = *b.ctx;
  hipStream_t st = c.stream;
  const DioMeta& m = ((DioHost*)b.dio_host)->meta;
  // speed > 1: decimate first (dio.cpp:68-70); the rest of the chain then reads the decimated signal
  const double* src = d_x;
  const int64_t* src_off = b.d_x_off;
  const int* src_len = b.d_x_len;
  if (m.ratio > 1) {
    const DecMeta dm = make_dec_meta(m.ratio, 0);
    const int len_max = b.max_x_len + 18;
    const int blocks = (len_max + 64 * kDecChunk - 1) / (64 * kDecChunk);
    (void)hipMemsetAsync(b.d_dio_y, 0, sizeof(double) * (size_t)b.dio_tot_y, st);
    hipLaunchKernelGGL(decim_fwd_kernel, dim3(blocks, b.n_utt), dim3(64), 0, st, d_x, b.d_x_off, b.d_x_len, dm,
                       b.d_dio_toff, b.d_dio_tmp);
    hipLaunchKernelGGL(decim_bwd_kernel, dim3(blocks, b.n_utt), dim3(64), 0, st, b.d_x_len, dm, b.d_dio_toff,
                       b.d_dio_tmp, b.d_dio_yoff, b.d_dio_ylen, b.d_dio_y);
    src = b.d_dio_y; src_off = b.d_dio_yoff; src_len = b.d_dio_ylen;
  }
  hipLaunchKernelGGL(dio_mean_partial_kernel, dim3(kMeanTiles, b.n_utt), dim3(256), 0, st, src, src_off, src_len,
                     b.d_dio_mean_part);
  hipLaunchKernelGGL(dio_mean_kernel, dim3((b.n_utt + 255) / 256), dim3(256), 0, st, b.d_dio_mean_part, b.d_dio_ylen,
                     b.n_utt, b.d_dio_mean);
  {
    const int total_max = b.max_x_len / m.ratio + 1 + 2 * m.pad;
    const int tiles = (total_max + kBandTile - 1) / kBandTile;
    const int ntap = 2 * m.cut + 1;
    const bool small = ntap <= zc_max_taps<kZcStrideHarvest>();
    if (!m.lc_conv && !small && ntap > zc_max_taps<kZcStrideLong>()) return WM_ERR_UNSUPPORTED;
    const size_t lds = sizeof(double) * (size_t)((small ? kZcStrideHarvest : kZcStrideLong) * kBandK + zc_pad16(ntap));
    TimedScope ts_(b.ctx, "dio_lowcut_kernel");
    if (m.lc_conv) {
      const int V = m.lc_conv - ntap + 1;
      const dim3 grid((total_max + V - 1) / V, b.n_utt);
      if (m.lc_conv == 2048)
        hipLaunchKernelGGL(dio_lowcut_fft_kernel<2048>, grid, dim3(64), 0, st, src, src_off, src_len, b.d_dio_ylen,
                           b.d_dio_mean, b.d_dio_fft, (const cpx*)b.d_dio_H, m, b.d_dio_z_off, b.d_dio_z);
      else
        hipLaunchKernelGGL(dio_lowcut_fft_kernel<4096>, grid, dim3(64), 0, st, src, src_off, src_len, b.d_dio_ylen,
                           b.d_dio_mean, b.d_dio_fft, (const cpx*)b.d_dio_H, m, b.d_dio_z_off, b.d_dio_z);
    } else if (small)
      hipLaunchKernelGGL(dio_lowcut_kernel<kZcStrideHarvest>, dim3(tiles, b.n_utt), dim3(256), lds, st, src, src_off,
                         src_len, b.d_dio_ylen, b.d_dio_mean, b.d_dio_fft, b.d_dio_lowcut, m, b.d_dio_z_off,
                         b.d_dio_z);
    else
      hipLaunchKernelGGL(dio_lowcut_kernel<kZcStrideLong>, dim3(tiles, b.n_utt), dim3(256), lds, st, src, src_off,
                         src_len, b.d_dio_ylen, b.d_dio_mean, b.d_dio_fft, b.d_dio_lowcut, m, b.d_dio_z_off,
                         b.d_dio_z);
  }
  {
    // row stride of the LDS tile by the longest filter: 16 kHz fits the small one, 48 kHz needs the large
    const int ntap_max = 4 * m.hal[0];
    const bool small = ntap_max <= zc_max_taps<kZcStrideDio>();
    if (!m.band_conv && !small && ntap_max > zc_max_taps<kZcStrideHarvest>()) return WM_ERR_UNSUPPORTED;
    const size_t lds = sizeof(double) * (size_t)(small ? zc_lds_doubles<kZcStrideDio>(ntap_max)
                                                       : zc_lds_doubles<kZcStrideHarvest>(ntap_max));
    const int tiles_max = dio_tiles(b.max_x_len / m.ratio + 1, m.step);
    TimedScope ts_(b.ctx, "dio_band_kernel");
    if (m.band_conv) {
      const cpx* Hb = (const cpx*)b.d_dio_H + (m.lc_conv ? m.lc_conv / 2 + 1 : 0);
      if (m.band_conv == 4096) {
        allow_dynamic_lds(*b.ctx, dio_band_fft_kernel<4096>, (int)(ConvEvCfg<4096, kDioConvC>::kLdsBytes));
        hipLaunchKernelGGL(dio_band_fft_kernel<4096>, dim3(tiles_max, b.n_utt), dim3(64), (ConvEvCfg<4096, kDioConvC>::kLdsBytes), st,
                           b.d_dio_ylen, b.d_dio_z_off, b.d_dio_z, Hb, m, tiles_max, b.d_dio_tile_cnt, b.d_dio_slot_off,
                           b.d_dio_slots);
      } else {
        allow_dynamic_lds(*b.ctx, dio_band_fft_kernel<2048>, (int)(ConvEvCfg<2048, kDioConvC>::kLdsBytes));
        hipLaunchKernelGGL(dio_band_fft_kernel<2048>, dim3(tiles_max, b.n_utt), dim3(64), (ConvEvCfg<2048, kDioConvC>::kLdsBytes), st,
                           b.d_dio_ylen, b.d_dio_z_off, b.d_dio_z, Hb, m, tiles_max, b.d_dio_tile_cnt, b.d_dio_slot_off,
                           b.d_dio_slots);
      }
    } else if (small)
      hipLaunchKernelGGL((dio_band_kernel<kZcStrideDio>), dim3(tiles_max, m.nb, b.n_utt), dim3(256), lds, st,
                         b.d_dio_ylen, b.d_dio_z_off, b.d_dio_z, b.d_dio_win, m, tiles_max, b.d_dio_tile_cnt,
                         b.d_dio_slot_off, b.d_dio_slots);
    else
      hipLaunchKernelGGL((dio_band_kernel<kZcStrideHarvest>), dim3(tiles_max, m.nb, b.n_utt), dim3(256), lds, st,
                         b.d_dio_ylen, b.d_dio_z_off, b.d_dio_z, b.d_dio_win, m, tiles_max, b.d_dio_tile_cnt,
                         b.d_dio_slot_off, b.d_dio_slots);
    hipLaunchKernelGGL(dio_band_scan_kernel, dim3(m.nb, b.n_utt), dim3(64), 0, st, b.d_dio_ylen, m, tiles_max,
                       b.d_dio_tile_cnt, b.d_dio_ev_cnt);
    hipLaunchKernelGGL(dio_band_compact_kernel, dim3(m.nb, b.n_utt), dim3(256), 0, st, b.d_dio_ylen, m,
                       tiles_max, b.d_dio_tile_cnt, b.d_dio_slot_off, b.d_dio_slots, b.d_dio_ev_off,
                       b.d_dio_events);
  }
  {
    const int gx = (int)((b.total_f + 255) / 256);
    TimedScope ts_(b.ctx, "dio_candidate_kernel");
    hipLaunchKernelGGL(dio_candidate_kernel, dim3(gx, m.nb), dim3(256), 0, st, b.d_dio_ylen, b.d_f_off,
                       b.d_frame_utt, b.p.frame_period, m, b.p.f0_floor, b.p.f0_ceil, b.d_dio_ev_off, b.d_dio_events,
                       b.d_dio_ev_cnt, b.total_f, b.d_dio_cand, b.d_dio_score);
  }
  {
  TimedScope ts_(b.ctx, "dio_fix_kernel");
  const int edge_cap = b.max_f0_len / 2 + 2;
  const size_t lds = sizeof(int) * 2 * (size_t)edge_cap;
  if (lds <= 48 * 1024) {
    hipLaunchKernelGGL(dio_fix_kernel<true>, dim3(b.n_utt), dim3(256), lds, st, b.d_f_off, b.d_dio_cand, b.d_dio_score,
                       m.nb, b.p.frame_period, b.p.f0_floor, b.p.allowed_range, b.total_f, edge_cap, b.d_dio_ws,
                       (int*)nullptr, d_t, d_f0);
  } else {
    if (!b.d_dio_edges) {
      rc = wm_check(dev_alloc(&b.d_dio_edges, sizeof(int) * 2 * (size_t)edge_cap * (size_t)b.n_utt));
      if (rc) return rc;
    }
    hipLaunchKernelGGL(dio_fix_kernel<false>, dim3(b.n_utt), dim3(256), 0, st, b.d_f_off, b.d_dio_cand, b.d_dio_score,
                       m.nb, b.p.frame_period, b.p.f0_floor, b.p.allowed_range, b.total_f, edge_cap, b.d_dio_ws,
                       b.d_dio_edges, d_t, d_f0);
  }
  }
  return wm_check(hipGetLastError());
}

}  // namespace wm

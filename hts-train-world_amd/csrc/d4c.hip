// d4c.hip -- D4C band-aperiodicity estimation, one wavefront per voiced frame.
//
// Replaces D4C and everything below it (externs/WORLD_v2/src/d4c.cpp:21-397):
//   d4c_offsets_kernel<0/1>  the two randn consumption scans (d4c.cpp:340 reseed;
//                            LoveTrain first, then the frames that pass it)
//   d4c_lovetrain_kernel     D4CLoveTrain(+Sub), d4c.cpp:225-282
//   d4c_kernel               D4CGeneralBody d4c.cpp:290-316 + GetAperiodicity :325-333
// The std::sort of d4c.cpp:215 only feeds "sum of all but the (boundary+1)
// largest bins"; here the largest bins are peeled off by repeated wave-wide max
// and the rest summed directly (no sort).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "batch.hpp"
#include "bcommon.hpp"
#include "bfft.hpp"
#include "common.hpp"
#include "fft.hpp"
#include "partition.hpp"
#include "window.hpp"

namespace wm {

constexpr double kFloorF0D4C = 47.0;     // constantnumbers.h
constexpr double kFreqInterval = 3000.0;
constexpr double kUpperLimit = 15000.0;

__host__ __device__ inline int d4c_fft_size(int fs) {           // d4c.cpp:344-346
  return (int)pow(2.0, 1.0 + (int)(log(4.0 * fs / kFloorF0D4C + 1) / kLog2));
}
__host__ __device__ inline int lovetrain_fft_size(int fs) {     // d4c.cpp:261-263
  return (int)pow(2.0, 1.0 + (int)(log(3.0 * fs / 40.0 + 1) / kLog2));
}

// The frames that cost something, listed first (partition.hpp): LoveTrain works on f0 != 0
// (d4c.cpp:231-233), the body on f0 != 0 && aperiodicity0 > threshold (d4c.cpp:380).
struct VoicedPred {
  const double* f0;
  __device__ bool operator()(int i) const { return f0[i] != 0.0; }
};
struct D4cRunPred {
  const double* f0;
  const double* ap0;
  double threshold;
  __device__ bool operator()(int i) const { return f0[i] != 0.0 && ap0[i] > threshold; }
};

// MODE 0: LoveTrain consumption 2*round(1.5 fs/max(f0,40))+1 for frames with f0 != 0
//         (d4c.cpp:231-233, :272-278); writes per-utterance totals to utt_total.
// MODE 1: body consumption 3*(2*round(2 fs/max(f0,47))+1) for frames with f0 != 0 and
//         ap0 > threshold (d4c.cpp:380-383, :94-97, :152-153), offset by utt_total.
template <int MODE>
__global__ __launch_bounds__(256) void d4c_offsets_kernel(const double* __restrict__ f0,
                                                          const double* __restrict__ ap0,
                                                          const int64_t* __restrict__ f_off, int fs,
                                                          double threshold, int* __restrict__ utt_total,
                                                          int* __restrict__ rng_off) {
  __shared__ int part[256];
  __shared__ int carry_s;
  const int u = blockIdx.x;
  const int64_t base = f_off[u];
  const int nf = (int)(f_off[u + 1] - base);
  if (threadIdx.x == 0) carry_s = MODE == 0 ? 0 : utt_total[u];
  __syncthreads();
  for (int start = 0; start < nf; start += 256) {
    const int i = start + threadIdx.x;
    int c = 0;
    if (i < nf) {
      double v = f0[base + i];
      if (MODE == 0) {
        if (v != 0.0) c = 2 * matlab_round(1.5 * fs / (v > 40.0 ? v : 40.0)) + 1;
      } else {
        if (v != 0.0 && ap0[base + i] > threshold)
          c = 3 * (2 * matlab_round(2.0 * fs / (v > kFloorF0D4C ? v : kFloorF0D4C)) + 1);
      }
    }
    part[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      int tv = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
      __syncthreads();
      part[threadIdx.x] += tv;
      __syncthreads();
    }
    const int carry = carry_s;
    if (i < nf) rng_off[base + i] = carry + part[threadIdx.x] - c;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + part[255];
    __syncthreads();
  }
  if (MODE == 0 && threadIdx.x == 0) utt_total[u] = carry_s;
}

// D4CLoveTrainSub (d4c.cpp:225-250): aperiodicity0 = cum[boundary1] / cum[boundary2]
// over the power spectrum with bins <= boundary0 zeroed.
template <int FL>
__global__ __launch_bounds__(64, 2) void d4c_lovetrain_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int* __restrict__ frame_utt, const double* __restrict__ tpos, const double* __restrict__ f0,
    const int* __restrict__ rng_off, const uint32_t* __restrict__ rtab, int fs, int64_t total_frames,
    const int* __restrict__ perm, const int* __restrict__ n_listed, double* __restrict__ ap0) {
  constexpr int N = FL / 2, M = N / 64;
  __shared__ __attribute__((aligned(16))) double smem[2 * FftLds<N>::kElems];
  cpx* img = reinterpret_cast<cpx*>(smem);
  const int lane0 = threadIdx.x;
  FftTw<N> tw;
  tw.init(lane0);
  const int b0 = (int)ceil(100.0 * FL / fs), b1 = (int)ceil(4000.0 * FL / fs), b2 = (int)ceil(7900.0 * FL / fs);
  const int n_run = *n_listed;
  for (int64_t i = n_run + blockIdx.x * 64 + lane0; i < total_frames; i += (int64_t)gridDim.x * 64)
    ap0[perm[i]] = 0.0;                                     // f0 == 0 (d4c.cpp:231-233)
  WM_FOR_EACH_LISTED(frame, perm, n_run) {
    const int lane = opaque_lane(lane0);
    const double f0v = f0[frame];
    const int u = frame_utt[frame];
    const double cf0 = f0v > 40.0 ? f0v : 40.0;
    cpx v[M];
    windowed_waveform_lds<kBlackman, false>(x + x_off[u], x_len[u], fs, cf0, tpos[frame], 3.0, rtab,
                                            rng_off[frame], lane, smem, FL);
    load_packed<N>(smem, lane, v);
    rfft_forward<N>(v, img, img, tw, lane);
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int k = lane + 64 * m;
      cpx s = img[k];
      double p = s.x * s.x + s.y * s.y;
      if (k > b0 && k <= b1) s1 += p;
      if (k > b0 && k <= b2) s2 += p;
    }
    if (lane == 0) {        // bin N (Nyquist) only if a boundary reaches it
      cpx s = img[N];
      double p = s.x * s.x + s.y * s.y;
      if (N <= b1) s1 += p;
      if (N <= b2) s2 += p;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) ap0[frame] = s1 / s2;
    wave_sync();
  }
}

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

struct D4CTables {
  const double* nuttall;    // [window_length] NuttallWindow(window_length) (d4c.cpp:356-359)
  int window_length;
  int nap;                  // number_of_aperiodicities (d4c.cpp:351-353)
};

// FD = fft_size_d4c.  Rows of `ap` have out_bins = fft_size/2+1 entries (CheapTrick's size).
// Variant B: one 256-thread workgroup per frame (bfft.hpp / bcommon.hpp): each thread owns the bins
// tid + 256 q of every spectrum-domain array and one radix-4 butterfly of every FFT pass.
template <int FD, int WAVES>
__global__ __launch_bounds__(256, WAVES) void d4c_block_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int* __restrict__ frame_utt, const double* __restrict__ tpos, const double* __restrict__ f0,
    const double* __restrict__ ap0, const int* __restrict__ rng_off, const uint32_t* __restrict__ rtab,
    int fs, double threshold, D4CTables tab, int out_fft, int64_t total_frames, const int* __restrict__ perm,
    const int* __restrict__ n_listed, double* __restrict__ ap, int dbg) {
  constexpr int NW = 4, NT = 64 * NW;
  constexpr int N = FD / 2, H = FD / 2;
  constexpr int QB = (H + 1 + NT - 1) / NT;        // bins per thread
  constexpr int QS = FD / NT;                      // time samples per thread
  constexpr int kA = H + 2;
  constexpr int kBMax = FD / 16;
  constexpr int kImg = 2 * (N + 1);
  constexpr int kMain = kImg > (kA + H + 2 * kBMax + 2) ? kImg : (kA + H + 2 * kBMax + 2);
  constexpr int kTopMax = 80;                      // >= boundary + 1 (22 at 16 kHz, 65 at 48 kHz)
  __shared__ __attribute__((aligned(16))) double smem[kMain + (NW + 1) * kTopMax + 2 * NW + 8];
  double* arr = smem;                         // [H+1] spectrum-domain array
  double* seg = smem + kA;                    // scan / DC scratch
  cpx* img = reinterpret_cast<cpx*>(smem);    // FFT image / time-domain frame (aliases both)
  double* top = smem + kMain;                 // [NW][kTopMax] per-wave largest bins, then [kTopMax] global
  double* gtop = top + NW * kTopMax;
  double* red = gtop + kTopMax;               // [NW] reduction scratch

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  BFft<N, NT> bf;
  bf.init(tid);
  const int out_bins = out_fft / 2 + 1;

  const int n_run = *n_listed;
  WM_FOR_EACH_LISTED(frame, perm + n_run, total_frames - n_run) {   // d4c.cpp:318-323, :380
    double* row = ap + frame * (int64_t)out_bins;
    for (int i = tid; i < out_bins; i += NT) row[i] = 1.0 - kSafe;
  }
  WM_FOR_EACH_LISTED(frame, perm, n_run) {
    double* row = ap + frame * (int64_t)out_bins;
    const double f0v = f0[frame];
    bool run = f0v != 0.0 && ap0[frame] > threshold;                 // d4c.cpp:380
    const double cf0 = f0v > kFloorF0D4C ? f0v : kFloorF0D4C;         // d4c.cpp:381
    // LDS capacity guard for the smoothing scratch (width = f0 is the widest): f0 < fs/16
    if (run && (int)(cf0 * FD / fs) + 1 > kBMax) run = false;
    if (!run) {
      for (int i = tid; i < out_bins; i += NT) row[i] = 1.0 - kSafe;  // d4c.cpp:318-323
      continue;
    }
    const int u = frame_utt[frame];
    const double* xu = x + x_off[u];
    const int xl = x_len[u];
    const double pos = tpos[frame];
    const int roff = rng_off[frame];
    const int Lw = 2 * matlab_round(2.0 * fs / cf0) + 1;
    __syncthreads();

    // All three analysis windows of this frame (two Blackman centroid frames, one Hann power
    // frame; d4c.cpp:94-95, :152-153) span 4 periods, so they share cos(pi a (i - hw)):
    // generate it once per frame and keep this thread's QS samples in registers.
    const int hw = matlab_round(4.0 * fs / cf0 / 2.0);            // d4c.cpp:55-56
    const int L = 2 * hw + 1;
    double cw[QS];
    {
      CosGen g;
      g.init(2.0 * cf0 / (4.0 * fs), tid - hw, NT);               // d4c.cpp:36-37
#pragma unroll
      for (int q = 0; q < QS; ++q) { cw[q] = g.c; g.next(); }
    }
    // windowed, dithered, mean-removed frame of GetWindowedWaveform (d4c.cpp:52-84) into fv[]
    auto build = [&](auto type_tag, double cpos, int ro, double (&fv)[QS]) {
      constexpr int TYPE = decltype(type_tag)::value;
      const int origin = matlab_round(cpos * fs + 0.001);
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int q = 0; q < QS; ++q) {
        const int i = tid + NT * q;
        fv[q] = 0.0;
        if (i < L) {
          const double w = window_value<TYPE>(cw[q]);
          fv[q] = xu[imin(xl - 1, imax(0, origin + i - hw))] * w + randn_at(rtab, ro + i) * kSafe;
          s1 += fv[q];
          s2 += w;
        }
      }
      BlockOps<NW>::sum2(s1, s2, red, tid);
      const double coef = s1 / s2;
#pragma unroll
      for (int q = 0; q < QS; ++q) {
        const int i = tid + NT * q;
        if (i < L) fv[q] -= window_value<TYPE>(cw[q]) * coef;
      }
    };

    // ---- GetStaticCentroid (d4c.cpp:125-142): two centroids at pos -/+ 0.25/f0 ----
    double sc[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) sc[q] = 0.0;
#pragma unroll 1
    for (int side = 0; side < ((dbg & 1) ? 0 : 2); ++side) {
      const double cpos = side == 0 ? pos - 0.25 / cf0 : pos + 0.25 / cf0;
      double fv[QS];
      build(std::integral_constant<int, kBlackman>{}, cpos, roff + side * Lw, fv);
      double pwr = 0.0;                                   // d4c.cpp:96-100
#pragma unroll
      for (int q = 0; q < QS; ++q) pwr += fv[q] * fv[q];
      const double nrm = sqrt(BlockOps<NW>::sum(pwr, red, tid));
#pragma unroll
      for (int q = 0; q < QS; ++q) {
        fv[q] /= nrm;
        smem[tid + NT * q] = fv[q];
      }
      bf.rfft_forward(img, tid);
      cpx s1[QB];
#pragma unroll
      for (int q = 0; q < QB; ++q) {
        const int k = tid + NT * q;
        s1[q] = k <= H ? img[k] : make_double2(0.0, 0.0);
      }
      __syncthreads();
      // second transform of the same frame times (i + 1)  (d4c.cpp:110-112)
#pragma unroll
      for (int q = 0; q < QS; ++q) smem[tid + NT * q] = fv[q] * (tid + NT * q + 1.0);
      bf.rfft_forward(img, tid);
#pragma unroll
      for (int q = 0; q < QB; ++q) {
        const int k = tid + NT * q;
        if (k <= H) {
          const cpx s2 = img[k];
          sc[q] += s2.x * s1[q].x + s1[q].y * s2.y;          // d4c.cpp:113-115
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int k = tid + NT * q;
      if (k <= H) arr[k] = sc[q];
    }
    __syncthreads();
    dc_correction_blk<NT>(arr, cf0, fs, FD, seg, tid);     // d4c.cpp:139
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int k = tid + NT * q;
      if (k <= H) sc[q] = arr[k];
    }
    __syncthreads();

    // ---- GetSmoothedPowerSpectrum (d4c.cpp:148-164) ----
    if (!(dbg & 2)) {
      double fv[QS];
      build(std::integral_constant<int, kHann>{}, pos, roff + 2 * Lw, fv);
#pragma unroll
      for (int q = 0; q < QS; ++q) smem[tid + NT * q] = fv[q];
      bf.rfft_forward(img, tid);
      double p[QB];
#pragma unroll
      for (int q = 0; q < QB; ++q) {
        const int k = tid + NT * q;
        p[q] = 0.0;
        if (k <= H) {
          const cpx s = img[k];
          p[q] = s.x * s.x + s.y * s.y;
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < QB; ++q) {
        const int k = tid + NT * q;
        if (k <= H) arr[k] = p[q];
      }
      __syncthreads();
    }
    if (!(dbg & 4)) {
    dc_correction_blk<NT>(arr, cf0, fs, FD, seg, tid);
    linear_smoothing_blk<NW>(arr, cf0, fs, FD, seg, arr, red, tid);
    }
    // ---- GetStaticGroupDelay (d4c.cpp:170-186) ----
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int k = tid + NT * q;
      if (k <= H) arr[k] = sc[q] / arr[k];
    }
    __syncthreads();
    if (!(dbg & 4)) linear_smoothing_blk<NW>(arr, cf0 / 2.0, fs, FD, seg, arr, red, tid);
    double gd[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int k = tid + NT * q;
      gd[q] = k <= H ? arr[k] : 0.0;
    }
    __syncthreads();
    if (!(dbg & 4)) linear_smoothing_blk<NW>(arr, cf0, fs, FD, seg, arr, red, tid);
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int k = tid + NT * q;
      if (k <= H) gd[q] -= arr[k];
    }
    __syncthreads();

    // ---- GetCoarseAperiodicity (d4c.cpp:192-223) ----
    const int wl = tab.window_length;
    const int bnd = imin(matlab_round(FD * 8.0 / wl), kTopMax - 1);
    const int hwl = wl / 2;
    double coarse[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int band = 0; band < ((dbg & 8) ? 0 : tab.nap); ++band) {
#pragma unroll
      for (int q = 0; q < QB; ++q) {
        const int k = tid + NT * q;
        if (k <= H) arr[k] = gd[q];
      }
      __syncthreads();
      const int center = (int)(kFreqInterval * (band + 1) * FD / fs);
      double fr[QS];
#pragma unroll
      for (int q = 0; q < QS; ++q) {
        const int i = tid + NT * q;
        fr[q] = i < wl ? arr[center - hwl + i] * tab.nuttall[i] : 0.0;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < QS; ++q) smem[tid + NT * q] = fr[q];
      bf.rfft_forward(img, tid);
      double p[QB];
      double tot = 0.0;
#pragma unroll
      for (int q = 0; q < QB; ++q) {
        const int k = tid + NT * q;
        p[q] = -1.0;
        if (k <= H) {
          const cpx s = img[k];
          p[q] = s.x * s.x + s.y * s.y;
          tot += p[q];
        }
      }
      tot = BlockOps<NW>::sum(tot, red, tid);
      // the (bnd + 1) largest bins: cum[h - bnd - 1] keeps the h - bnd smallest (d4c.cpp:215-220).
      // Level 1: every wave peels its own bnd + 1 largest; level 2: wave 0 peels the union.
      {
        double c[QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) c[q] = p[q];
#pragma unroll 1
        for (int it = 0; it <= ((dbg & 16) ? 0 : bnd); ++it) {
          double mx = c[0];
#pragma unroll
          for (int q = 1; q < QB; ++q) mx = fmax(mx, c[q]);
          const double wmx = wave_max(mx);
          const unsigned long long vote = __ballot(mx == wmx);
          const int winner = __ffsll((long long)vote) - 1;
          if (lane == winner) {
            bool done = false;
#pragma unroll
            for (int q = 0; q < QB; ++q)
              if (!done && c[q] == wmx) { c[q] = -2.0; done = true; }
            top[wv * kTopMax + it] = wmx;
          }
        }
      }
      __syncthreads();
      if (wv == 0) {
        constexpr int QC = (NW * kTopMax + 63) / 64;
        double c[QC];
#pragma unroll
        for (int q = 0; q < QC; ++q) {
          const int idx = lane + 64 * q;                  // candidate (wave = idx / (bnd+1), rank = idx % (bnd+1))
          const int w2 = idx / (bnd + 1), r2 = idx - w2 * (bnd + 1);
          c[q] = w2 < NW ? top[w2 * kTopMax + r2] : -2.0;
        }
#pragma unroll 1
        for (int it = 0; it <= bnd; ++it) {
          double mx = c[0];
#pragma unroll
          for (int q = 1; q < QC; ++q) mx = fmax(mx, c[q]);
          const double wmx = wave_max(mx);
          const unsigned long long vote = __ballot(mx == wmx);
          const int winner = __ffsll((long long)vote) - 1;
          if (lane == winner) {
            bool done = false;
#pragma unroll
            for (int q = 0; q < QC; ++q)
              if (!done && c[q] == wmx) { c[q] = -2.0; done = true; }
            gtop[it] = wmx;
          }
        }
      }
      __syncthreads();
      const double tau = gtop[bnd];                       // the (bnd+1)-th largest value
      int n_gt = 0;
      for (int it = 0; it <= bnd; ++it) n_gt += gtop[it] > tau ? 1 : 0;
      const int need_eq = bnd + 1 - n_gt;                 // copies of tau among the removed bins
      double low = 0.0, eq = 0.0;
#pragma unroll
      for (int q = 0; q < QB; ++q) {
        if (p[q] >= 0.0 && p[q] < tau) low += p[q];
        if (p[q] == tau) eq += 1.0;
      }
      low = BlockOps<NW>::sum(low, red, tid);
      eq = BlockOps<NW>::sum(eq, red, tid);
      low += (eq - need_eq) * tau;
      double c = 10.0 * log10(low / tot);
      c = c + (cf0 - 100.0) / 50.0;                         // d4c.cpp:309-311
      c = c < 0.0 ? c : 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (j == band) coarse[j] = c;               // static indices keep coarse[] in registers
      __syncthreads();
    }

    // ---- GetAperiodicity (d4c.cpp:325-333): interp1 over {0, 3000 i, fs/2} then 10^(x/20) ----
    const int nk = tab.nap + 2;
    for (int i = tid; i < out_bins; i += NT) {
      const double f = (double)i * fs / out_fft;
      int k = 0;                                            // #{knots <= f}
      for (int j = 0; j < nk; ++j) {
        double xj = j <= tab.nap ? j * kFreqInterval : fs / 2.0;
        k += xj <= f ? 1 : 0;
      }
      k = k < 1 ? 1 : (k > nk - 1 ? nk - 1 : k);
      const double x0 = (k - 1) <= tab.nap ? (k - 1) * kFreqInterval : fs / 2.0;
      const double x1 = k <= tab.nap ? k * kFreqInterval : fs / 2.0;
      double y0 = -60.0, y1 = -kSafe;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (j < tab.nap) {
          if (k - 1 == j + 1) y0 = coarse[j];
          if (k == j + 1) y1 = coarse[j];
        }
      }
      const double s = (f - x0) / (x1 - x0);
      const double yi = y0 + s * (y1 - y0);
      row[i] = exp(yi * (2.302585092994045684 / 20.0));       // 10^(yi/20), d4c.cpp:331-332
    }
    __syncthreads();
  }
}

// FD = fft_size_d4c.  Rows of `ap` have out_bins = fft_size/2+1 entries (CheapTrick's size).
// Variant A: one wavefront per frame (fft.hpp), everything in registers + 18.5 KB of LDS.
template <int FD, int WAVES>
__global__ __launch_bounds__(64, WAVES) void d4c_wave_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int* __restrict__ frame_utt, const double* __restrict__ tpos, const double* __restrict__ f0,
    const double* __restrict__ ap0, const int* __restrict__ rng_off, const uint32_t* __restrict__ rtab,
    int fs, double threshold, D4CTables tab, int out_fft, int64_t total_frames, const int* __restrict__ perm,
    const int* __restrict__ n_listed, double* __restrict__ ap, int dbg) {
  constexpr int N = FD / 2, M = N / 64, H = FD / 2, MB = M + 1;
  constexpr int kA = H + 2;
  constexpr int kBMax = FD / 16;
  constexpr int kImg = 2 * FftLds<N>::kElems;
  constexpr int kCh = ((H + 2 * kBMax + 1 + 63) / 64) | 1;   // odd per-lane chunk of the smoothing scan
  constexpr int kTot = kImg > (kA + 64 * kCh) ? kImg : (kA + 64 * kCh);
  __shared__ __attribute__((aligned(16))) double smem[kTot];
  double* arr = smem;                         // [H+1] spectrum-domain array
  double* seg = smem + kA;                    // scan / DC scratch
  cpx* img = reinterpret_cast<cpx*>(smem);    // FFT image (aliases both)

  const int lane0 = threadIdx.x;
  FftTw<N> tw;
  tw.init(lane0);
  const int out_bins = out_fft / 2 + 1;

  const int n_run = *n_listed;
  WM_FOR_EACH_LISTED(frame, perm + n_run, total_frames - n_run) {   // d4c.cpp:318-323, :380
    double* row = ap + frame * (int64_t)out_bins;
    for (int i = lane0; i < out_bins; i += 64) row[i] = 1.0 - kSafe;
  }
  WM_FOR_EACH_LISTED(frame, perm, n_run) {
    const int lane = opaque_lane(lane0);
    double* row = ap + frame * (int64_t)out_bins;
    const double f0v = f0[frame];
    bool run = f0v != 0.0 && ap0[frame] > threshold;                 // d4c.cpp:380
    const double cf0 = f0v > kFloorF0D4C ? f0v : kFloorF0D4C;         // d4c.cpp:381
    // LDS capacity guard for the smoothing scratch (width = f0 is the widest): f0 < fs/16
    if (run && (int)(cf0 * FD / fs) + 1 > kBMax) run = false;
    if (!run) {
      for (int i = lane; i < out_bins; i += 64) row[i] = 1.0 - kSafe;  // d4c.cpp:318-323
      continue;
    }
    const int u = frame_utt[frame];
    const double* xu = x + x_off[u];
    const int xl = x_len[u];
    const double pos = tpos[frame];
    const int roff = rng_off[frame];
    const int Lw = 2 * matlab_round(2.0 * fs / cf0) + 1;

    // ---- GetStaticCentroid (d4c.cpp:125-142): two centroids at pos -/+ 0.25/f0 ----
    double sc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) sc[m] = 0.0;
#pragma unroll 1
    for (int side = 0; side < ((dbg & 1) ? 0 : 2); ++side) {
      const double cpos = side == 0 ? pos - 0.25 / cf0 : pos + 0.25 / cf0;
      const int ro = roff + side * Lw;
      cpx v[M];
      const FrameWindow fw = windowed_waveform_lds<kBlackman, false, 8>(xu, xl, fs, cf0, cpos, 4.0, rtab, ro, lane,
                                                                     smem, FD);
      double pwr = 0.0;                                   // d4c.cpp:96-100
      for (int i = lane; i < fw.L; i += 64) pwr += smem[i] * smem[i];
      // normalisation by sqrt(power) (d4c.cpp:99-100) applied as a multiplication by its reciprocal
      const double rnrm = 1.0 / sqrt(wave_sum(pwr));
      load_packed<N>(smem, lane, v);
      cpx fv[M];                                          // normalised frame, kept for the ramped transform
#pragma unroll
      for (int m = 0; m < M; ++m) { v[m].x *= rnrm; v[m].y *= rnrm; fv[m] = v[m]; }
      rfft_forward<N>(v, img, img, tw, lane);
      cpx s1[MB];
#pragma unroll
      for (int m = 0; m < M; ++m) s1[m] = img[lane + 64 * m];
      s1[M] = img[N];
      // second transform of the same frame times (i + 1)  (d4c.cpp:110-112)
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const int i0 = 2 * (lane + 64 * m);
        v[m] = make_double2(fv[m].x * (i0 + 1.0), fv[m].y * (i0 + 2.0));
      }
      rfft_forward<N>(v, img, img, tw, lane);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        cpx s2 = img[lane + 64 * m];
        sc[m] += s2.x * s1[m].x + s1[m].y * s2.y;          // d4c.cpp:113-115
      }
      {
        cpx s2 = img[N];
        sc[M] += s2.x * s1[M].x + s1[M].y * s2.y;
      }
      wave_sync();
    }
#pragma unroll
    for (int m = 0; m < M; ++m) arr[lane + 64 * m] = sc[m];
    if (lane == 0) arr[N] = sc[M];
    wave_sync();
    dc_correction_lds(arr, cf0, fs, FD, seg, lane);        // d4c.cpp:139
#pragma unroll
    for (int m = 0; m < M; ++m) sc[m] = arr[lane + 64 * m];
    sc[M] = arr[N];
    wave_sync();

    // ---- GetSmoothedPowerSpectrum (d4c.cpp:148-164) ----
    double gd[MB];
    if (!(dbg & 2)) {
      cpx v[M];
      windowed_waveform_lds<kHann, false, 8>(xu, xl, fs, cf0, pos, 4.0, rtab, roff + 2 * Lw, lane, smem, FD);
      load_packed<N>(smem, lane, v);
      rfft_forward<N>(v, img, img, tw, lane);
      double p[MB];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        cpx s = img[lane + 64 * m];
        p[m] = s.x * s.x + s.y * s.y;
      }
      {
        cpx s = img[N];
        p[M] = s.x * s.x + s.y * s.y;
      }
      wave_sync();
#pragma unroll
      for (int m = 0; m < M; ++m) arr[lane + 64 * m] = p[m];
      if (lane == 0) arr[N] = p[M];
      wave_sync();
    }
    if (!(dbg & 4)) {
    dc_correction_lds(arr, cf0, fs, FD, seg, lane);
    linear_smoothing_lds<kCh>(arr, cf0, fs, FD, seg, arr, lane);
    }
    // ---- GetStaticGroupDelay (d4c.cpp:170-186) ----
#pragma unroll
    for (int m = 0; m < M; ++m) arr[lane + 64 * m] = sc[m] / arr[lane + 64 * m];
    if (lane == 0) arr[N] = sc[M] / arr[N];
    wave_sync();
    if (!(dbg & 4)) linear_smoothing_lds<kCh>(arr, cf0 / 2.0, fs, FD, seg, arr, lane);
#pragma unroll
    for (int m = 0; m < M; ++m) gd[m] = arr[lane + 64 * m];
    gd[M] = arr[N];
    wave_sync();
    if (!(dbg & 4)) linear_smoothing_lds<kCh>(arr, cf0, fs, FD, seg, arr, lane);
#pragma unroll
    for (int m = 0; m < M; ++m) gd[m] -= arr[lane + 64 * m];
    gd[M] -= arr[N];
    wave_sync();

    // ---- GetCoarseAperiodicity (d4c.cpp:192-223) ----
    const int wl = tab.window_length;
    const int bnd = matlab_round(FD * 8.0 / wl);
    const int hwl = wl / 2;
    double coarse[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int band = 0; band < ((dbg & 8) ? 0 : tab.nap); ++band) {
#pragma unroll
      for (int m = 0; m < M; ++m) arr[lane + 64 * m] = gd[m];
      if (lane == 0) arr[N] = gd[M];
      wave_sync();
      const int center = (int)(kFreqInterval * (band + 1) * FD / fs);
      cpx v[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const int i0 = 2 * (lane + 64 * m);
        double a0 = 0.0, a1 = 0.0;
        if (i0 < wl) a0 = arr[center - hwl + i0] * tab.nuttall[i0];
        if (i0 + 1 < wl) a1 = arr[center - hwl + i0 + 1] * tab.nuttall[i0 + 1];
        v[m] = make_double2(a0, a1);
      }
      rfft_forward<N>(v, img, img, tw, lane);
      double p[MB];
      double tot = 0.0;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        cpx s = img[lane + 64 * m];
        p[m] = s.x * s.x + s.y * s.y;
        tot += p[m];
      }
      p[M] = -1.0;
      if (lane == 0) {
        cpx s = img[N];
        p[M] = s.x * s.x + s.y * s.y;
        tot += p[M];
      }
      tot = wave_sum(tot);
      // Sum of all but the (bnd + 1) largest bins: cum[h - bnd - 1] keeps the h - bnd smallest
      // (d4c.cpp:215-220).  Each lane sorts its own bins once (Batcher network, static indices) and
      // parks the sorted column in LDS (the FFT image is free again); the largest bins are then
      // peeled by repeated wave-wide max over the lanes' current heads, and the winning lane
      // advances its head with one LDS read.
      wave_sync();
      sort_desc<M>(p);
#pragma unroll
      for (int i = M - 1; i >= 0; --i) {                    // Nyquist bin (lane 0 only, -1 elsewhere) into place
        const double hi = fmax(p[i], p[i + 1]), lo = fmin(p[i], p[i + 1]);
        p[i] = hi;
        p[i + 1] = lo;
      }
      double* heads = smem;                                   // [MB + 1][64]
#pragma unroll
      for (int m = 0; m < MB; ++m) heads[m * 64 + lane] = p[m];
      heads[MB * 64 + lane] = -1.0;                           // exhausted
      int taken = 0;
      double cur = p[0];
#pragma unroll 1
      for (int it = 0; it <= ((dbg & 16) ? 0 : bnd); ++it) {
        const double wmx = wave_max(cur);
        const unsigned long long vote = __ballot(cur == wmx);
        const int winner = __ffsll((long long)vote) - 1;
        taken += lane == winner ? 1 : 0;                      // branch-free: every lane re-reads its head
        cur = heads[taken * 64 + lane];
      }
      double low = 0.0;
#pragma unroll
      for (int m = 0; m < MB; ++m) low += (m >= taken && p[m] >= 0.0) ? p[m] : 0.0;
      low = wave_sum(low);
      double c = 10.0 * log10(low / tot);
      c = c + (cf0 - 100.0) / 50.0;                         // d4c.cpp:309-311
      c = c < 0.0 ? c : 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (j == band) coarse[j] = c;               // static indices keep coarse[] in registers
      wave_sync();
    }

    // ---- GetAperiodicity (d4c.cpp:325-333): interp1 over {0, 3000 i, fs/2} then 10^(x/20) ----
    // knot values {-60, coarse..., -kSafe} go through LDS (the FFT image is free here) so that the
    // segment of every output bin is one indexed read; the segment index is floor(f / 3000) taken
    // with a reciprocal: on an exact knot it may pick the segment to the left with s = 1, which is
    // the same point of the (continuous) interpolant.
    if (lane <= tab.nap + 1) {
      double kv = lane == 0 ? -60.0 : -kSafe;
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (lane == j + 1 && j < tab.nap) kv = coarse[j];
      smem[lane] = kv;
    }
    wave_sync();
    {
      const double bin_hz = (double)fs / out_fft;
      const double last_w = fs / 2.0 - tab.nap * kFreqInterval;
      const double inv_last = 1.0 / last_w;
      for (int i = lane; i < ((dbg & 64) ? 0 : out_bins); i += 64) {
        const double f = (double)i * bin_hz;
        int kk = (int)(f * (1.0 / kFreqInterval));
        kk = kk > tab.nap ? tab.nap : kk;
        const double x0 = kk * kFreqInterval;
        const double sfr = (f - x0) * (kk == tab.nap ? inv_last : 1.0 / kFreqInterval);
        const double y0 = smem[kk], y1 = smem[kk + 1];
        const double yi = y0 + sfr * (y1 - y0);
        row[i] = exp(yi * (2.302585092994045684 / 20.0));       // 10^(yi/20), d4c.cpp:331-332
      }
    }
    wave_sync();
  }
}

int launch_d4c(Batch& b, const double* d_x, const double* d_t, const double* d_f0, double* d_ap) {
  Context& c = *b.ctx;
  hipStream_t st = c.stream;
  const int fs = b.p.fs;
  const int FD = d4c_fft_size(fs), FL = lovetrain_fft_size(fs);
  if (FD != FL || (FD != 1024 && FD != 2048 && FD != 4096)) {
    return WM_ERR_UNSUPPORTED_FFT;
  }
  int rc = c.ensure_rng(b.rng_bound_d4c());
  if (rc) return rc;
  // Nuttall window table for GetCoarseAperiodicity (d4c.cpp:356-359, common.cpp:113-121)
  const int wl = (int)(kFreqInterval * FD / fs) * 2 + 1;
  if (!b.d_d4c_window) {
    std::vector<double> w((size_t)wl);
    for (int i = 0; i < wl; ++i) {
      double tmp = i / (wl - 1.0);
      w[(size_t)i] = 0.355768 - 0.487396 * cos(2.0 * kPi * tmp) + 0.144232 * cos(4.0 * kPi * tmp) -
                     0.012604 * cos(6.0 * kPi * tmp);
    }
    rc = wm_check(hipMalloc((void**)&b.d_d4c_window, sizeof(double) * (size_t)wl));
    if (rc) return rc;
    rc = wm_check(hipMemcpyAsync(b.d_d4c_window, w.data(), sizeof(double) * (size_t)wl, hipMemcpyHostToDevice, st));
    if (rc) return rc;
    rc = wm_check(hipStreamSynchronize(st));   // w is a stack-lifetime buffer
    if (rc) return rc;
    rc = wm_check(hipMalloc((void**)&b.d_utt_total, sizeof(int) * (size_t)b.n_utt));
    if (rc) return rc;
  }
  D4CTables tab;
  tab.nuttall = b.d_d4c_window;
  tab.window_length = wl;
  double lim = fs / 2.0 - kFreqInterval;
  tab.nap = (int)((kUpperLimit < lim ? kUpperLimit : lim) / kFreqInterval);
  if (tab.nap < 1 || tab.nap > 6) return WM_ERR_UNSUPPORTED;

  const int64_t tf = b.total_f;
  const int grid = (int)(tf < (int64_t)c.frame_grid ? tf : (int64_t)c.frame_grid);
  hipLaunchKernelGGL(d4c_offsets_kernel<0>, dim3(b.n_utt), dim3(256), 0, st, d_f0, (const double*)nullptr,
                     b.d_f_off, fs, b.p.d4c_threshold, b.d_utt_total, b.d_rng_off2);
  launch_partition(st, VoicedPred{d_f0}, (int)tf, b.d_part_cnt, b.d_perm, b.d_part_n);
#define WM_LT_CASE(FF)                                                                                     \
  case FF: {                                                                                               \
    static const int per_ = persistent_grid(c, d4c_lovetrain_kernel<FF>, 64, (int64_t)1 << 40);            \
    hipLaunchKernelGGL(d4c_lovetrain_kernel<FF>, dim3(imin(grid, per_)), dim3(64), 0, st, d_x, b.d_x_off,  \
                       b.d_x_len, b.d_frame_utt, d_t, d_f0, b.d_rng_off2, c.d_rng, fs, tf,                \
                       (const int*)b.d_perm, (const int*)b.d_part_n, b.d_ap0);                             \
  } break;
  {
    TimedScope ts_(b.ctx, "d4c_lovetrain_kernel");
    switch (FL) {
      WM_LT_CASE(1024)
      WM_LT_CASE(2048)
      WM_LT_CASE(4096)
    }
  }
#undef WM_LT_CASE
  hipLaunchKernelGGL(d4c_offsets_kernel<1>, dim3(b.n_utt), dim3(256), 0, st, d_f0, (const double*)b.d_ap0,
                     b.d_f_off, fs, b.p.d4c_threshold, b.d_utt_total, b.d_rng_off);
  launch_partition(st, D4cRunPred{d_f0, b.d_ap0, b.p.d4c_threshold}, (int)tf, b.d_part_cnt, b.d_perm, b.d_part_n);
  // Variant choice: the one-wavefront kernel executes about half the instructions per FFT (radix 16/8/8 or
  // 16/16/8 in registers) and wins at every size -- at fft 4096 (48 kHz) it runs one wave per SIMD on 512
  // registers and still spills, but takes 11.8 ms where the workgroup-cooperative kernel takes 31.7 ms
  // (64 utterances, tools/rate_48k.py).  WORLD_MI355_D4C_VARIANT=wave|block overrides.
  static const int dbg = getenv("WORLD_MI355_D4C_DBG") ? atoi(getenv("WORLD_MI355_D4C_DBG")) : 0;
  static const char* var = getenv("WORLD_MI355_D4C_VARIANT");
  const bool use_block = var ? (var[0] == 'b') : false;
  const int block_waves = (var && var[0] == 'b' && var[5] >= '1' && var[5] <= '3') ? var[5] - '0' : 3;   // block1|block2|block3
#define WM_D4C_CASE(FF, WV)                                                                               \
  case FF:                                                                                                \
    if (use_block) {                                                                                      \
      if (block_waves == 1)                                                                               \
        hipLaunchKernelGGL((d4c_block_kernel<FF, 1>), dim3(grid), dim3(256), 0, st, d_x, b.d_x_off,       \
                           b.d_x_len, b.d_frame_utt, d_t, d_f0, (const double*)b.d_ap0, b.d_rng_off,      \
                           c.d_rng, fs, b.p.d4c_threshold, tab, b.p.fft_size, tf, (const int*)b.d_perm,   \
                           (const int*)b.d_part_n, d_ap, dbg);                                            \
      else if (block_waves == 2)                                                                          \
        hipLaunchKernelGGL((d4c_block_kernel<FF, 2>), dim3(grid), dim3(256), 0, st, d_x, b.d_x_off,       \
                           b.d_x_len, b.d_frame_utt, d_t, d_f0, (const double*)b.d_ap0, b.d_rng_off,      \
                           c.d_rng, fs, b.p.d4c_threshold, tab, b.p.fft_size, tf, (const int*)b.d_perm,   \
                           (const int*)b.d_part_n, d_ap, dbg);                                            \
      else                                                                                                \
      hipLaunchKernelGGL((d4c_block_kernel<FF, 3>), dim3(grid), dim3(256), 0, st, d_x, b.d_x_off,         \
                         b.d_x_len, b.d_frame_utt, d_t, d_f0, (const double*)b.d_ap0, b.d_rng_off,        \
                         c.d_rng, fs, b.p.d4c_threshold, tab, b.p.fft_size, tf, (const int*)b.d_perm,     \
                         (const int*)b.d_part_n, d_ap, dbg);                                              \
    } else {                                                                                              \
      static const int per_ = persistent_grid(c, d4c_wave_kernel<FF, WV>, 64, (int64_t)1 << 40);          \
      hipLaunchKernelGGL((d4c_wave_kernel<FF, WV>), dim3(imin(grid, per_)), dim3(64), 0, st, d_x,         \
                         b.d_x_off, b.d_x_len, b.d_frame_utt, d_t, d_f0, (const double*)b.d_ap0,          \
                         b.d_rng_off, c.d_rng, fs, b.p.d4c_threshold, tab, b.p.fft_size, tf,              \
                         (const int*)b.d_perm, (const int*)b.d_part_n, d_ap, dbg);                        \
    }                                                                                                     \
    break;
  {
    TimedScope ts_(b.ctx, "d4c_kernel");
    switch (FD) {
      WM_D4C_CASE(1024, 2)
      WM_D4C_CASE(2048, 2)
      WM_D4C_CASE(4096, 1)      // 32 complex values per lane and operand: one wave per SIMD, 512 registers
    }
  }
#undef WM_D4C_CASE
  return wm_check(hipGetLastError());
}

}  // namespace wm

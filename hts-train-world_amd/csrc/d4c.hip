// d4c.hip -- D4C band-aperiodicity estimation, one wavefront per voiced frame.
//
// Replaces D4C and everything below it (externs/WORLD_v2/src/d4c.cpp:21-397):
//   d4c_offsets_kernel<0/1>  the two randn consumption scans (d4c.cpp:340 reseed;
//                            LoveTrain first, then the frames that pass it)
//   d4c_lovetrain_kernel     D4CLoveTrain(+Sub), d4c.cpp:225-282
//   d4c_kernel               D4CGeneralBody d4c.cpp:290-316 + GetAperiodicity :325-333
//
// How this differs from the reference's arithmetic (results agree to ~1e-12, tests/test_gpu_parity.py):
//   * Frames are built in registers (frame.hpp), never staged in LDS.
//   * GetCentroid (d4c.cpp:90-119) needs Re(X1 conj X2) with X1 = FFT(x), X2 = FFT((i+1) x), both of FD real
//     points.  Both come out of ONE complex transform of z = s x + j (i+1) x (s a power of two near the ramp's
//     mean, so that neither part drowns the other): with Z = s X1 + j X2 and real sequences,
//         Im(Z[k] Z[FD-k]) = 2 s Re(X1[k] conj X2[k]).
//     The FD-point complex transform is one decimation-in-frequency step by hand (even bins = FFT of the folded
//     halves, odd bins = FFT of their twiddled difference) on top of the FD/2-point wavefront engine (fft.hpp);
//     the usual 4-period window is shorter than FD/2 samples, so the fold is empty and only long frames
//     (f0 below ~63 Hz at 16 kHz) take the general path.  Per side: 2 complex FFTs, no real-FFT split passes,
//     no second spectrum held in registers.
//   * The std::sort of d4c.cpp:215 only feeds "sum of all but the (boundary+1) largest bins": every lane sorts its
//     own bins once, the largest are peeled off by repeated wave-wide max, the rest is summed directly.
//   * DCCorrection / LinearSmoothing work in place on a spectrum stored with mirror margins (spectrum.hpp).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

// fastmath.hpp's coefficients as plain literals in this file: the D4C kernels already spill scalar registers (frame
// pipeline, kernel arguments, FFT constants), and every coefficient pinned to a scalar pair came back as a v_readlane
#define WM_FM_PLAIN 1
#include "batch.hpp"
#include "common.hpp"
#include "fastmath.hpp"
#include "fft.hpp"
#include "frame.hpp"
#include "partition.hpp"
#include "peel.hpp"
#include "spectrum.hpp"

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
        if (!(v == 0.0 || ap0[base + i] <= threshold))
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
__global__ __launch_bounds__(64, FL >= 4096 ? 1 : 2) void d4c_lovetrain_kernel(
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
  FramePipe pipe;
  pipe.init(perm, n_run, x, x_off, x_len, frame_utt, tpos, f0, rng_off);
  WM_FOR_EACH_PIPED(sc, pipe, n_run) {
    const int64_t frame = sc.frame;
    const int lane = opaque_lane(lane0);
    const double cf0 = uniform_d(sc.f0 > 40.0 ? sc.f0 : 40.0);
    cpx v[M];
    const FrameGeom fg = frame_geom(fs, cf0, uniform_d(sc.tpos), 3.0);
    frame_packed<kBlackman, false, M>(sc.xu, sc.xlen, fg, rtab, sc.roff, lane, v);
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

// D4CLoveTrain when the threshold is <= 0 (the recipe's setting, test/analysis.cpp:190).  Its ratio is used only
// in `f0 == 0 || aperiodicity0 <= threshold -> skip` (d4c.cpp:380).  The ratio of two sums of squared magnitudes
// is never negative, and it is zero only if every bin from 100 Hz to 4 kHz is exactly zero, which the 1e-12 randn
// dither of the window (d4c.cpp:66-67) rules out; a NaN ratio (non-finite samples under the window) compares
// false and does not skip either.  So for threshold <= 0 every frame with f0 != 0 is analysed whatever LoveTrain
// computes, and its transform is not run: aperiodicity0 is set to 1 on those frames.  The randn draws LoveTrain
// would have consumed are still accounted for by d4c_offsets_kernel<0>.
__global__ __launch_bounds__(256) void d4c_lovetrain_all_pass_kernel(const double* __restrict__ f0, int64_t total_frames,
                                                                     double* __restrict__ ap0) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < total_frames) ap0[i] = f0[i] != 0.0 ? 1.0 : 0.0;
}

struct D4CTables {
  const double* nuttall;    // [window_length] NuttallWindow(window_length) (d4c.cpp:356-359)
  int window_length;
  int nap;                  // number_of_aperiodicities (d4c.cpp:351-353)
};

// bins the widest LinearSmoothing of a frame mirrors at either end (width = f0; common.cpp:80)
__host__ __device__ inline int d4c_mirror_bins(double cf0, int fft_size_d4c, int fs) {
  const double q = cf0 * fft_size_d4c / fs;
  return q < 1.0e6 ? (int)q + 1 : 1 << 30;      // an infinite f0 from the caller must not wrap around the int range
}
// margin of the spectrum in LDS: the usual kernel covers f0 < fs / 16, the RARE one everything the
// reference defines (its DCCorrection reads past the spectrum from f0 ~ fs / 2 on, common.cpp:62-68)
template <int FD, bool RARE> struct D4CMargin { static constexpr int kBM = RARE ? FD / 2 : FD / 16; };

// One GetCentroid (d4c.cpp:90-119) at `cpos`, added into ce / co:
//   ce[m] += centroid at bin 2 j, co[m] += centroid at bin 2 j + 1, j = lane + 64 m, m < M / 2;
//   ce[M / 2] (lane 0): bin FD / 2.
// LONG: the frame has more than N = FD / 2 samples (fold not empty); it is then built twice, once per
// sub-transform, instead of being kept in registers across the first.
template <int N, bool LONG>
__device__ __forceinline__ void d4c_centroid(const double* __restrict__ xu, int xl, const FrameGeom& fg,
                                             const uint32_t* __restrict__ rtab, int ro, const FftTw<N>& tw, cpx* img,
                                             int lane0, double (&ce)[N / 128 + 1], double (&co)[N / 128]) {
  constexpr int M = N / 64, QX = LONG ? 2 * M : M;
  // per call: the ramp values and sample indices derived from the lane are the same for both centroids of a
  // frame, and the compiler would otherwise keep all of them in registers across the loop over the two
  const int lane = opaque_lane(lane0);
  // s: power of two next below the half window length (the mean of the ramp i + 1): exact scaling
  const double s = (double)(1 << (31 - __clz(fg.hw | 1)));
  double x[QX];
  double pwr;
  const int nzc = (fg.L + 63) >> 6;                 // registers of the operand the window reaches (not LONG: L <= N)
  frame_strided<kBlackman, QX, !LONG, (LONG ? 16 : 8)>(xu, xl, fg, rtab, ro, lane, x, pwr);
  // normalisation to unit energy (d4c.cpp:96-100) and the 1 / (2 s) of the identity above, on the result
  const double scale = uniform_d(1.0 / (2.0 * s * pwr));
  cpx v[M];
  // ---- even bins: E = FFT_N(z[n] + z[n + N]), z[n] = x[n] (s + j (n + 1)) ----
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const double r = (double)(lane + 64 * m + 1);
    if constexpr (!LONG) v[m] = make_double2(s * x[m], r * x[m]);
    else v[m] = make_double2(s * (x[m] + x[m + M]), r * x[m] + (r + N) * x[m + M]);
  }
  if constexpr (LONG) fft_forward<N>(v, img, tw, lane);
  else fft_forward_nz<N>(v, img, tw, lane, nzc);
  wave_sync();
#pragma unroll
  for (int m = M / 2; m < M; ++m) img[lane + 64 * m] = v[m];       // the partners E[N - j] of j < N / 2
  wave_sync();
#pragma unroll
  for (int m = 0; m < M / 2; ++m) {
    const int j = lane + 64 * m;
    cpx pt = img[(N - j) & (N - 1)];
    if (m == 0) {                                                   // E[0] pairs with itself (component-wise:
      pt.x = lane == 0 ? v[0].x : pt.x;                             // a select between two cpx objects keeps
      pt.y = lane == 0 ? v[0].y : pt.y;                             // v[] addressable, i.e. in scratch memory)
    }
    ce[m] += (v[m].x * pt.y + v[m].y * pt.x) * scale;
  }
  ce[M / 2] += 2.0 * v[M / 2].x * v[M / 2].y * scale;              // lane 0: E[N / 2] pairs with itself
  // ---- odd bins: O = FFT_N((z[n] - z[n + N]) W_FD^n) ----
  if constexpr (LONG) frame_strided<kBlackman, QX, !LONG>(xu, xl, fg, rtab, ro, lane, x, pwr);
  cpx w = tw.wsplit;                                                // W_FD^lane
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const double r = (double)(lane + 64 * m + 1);
    if constexpr (!LONG) {
      // x (s + j r) w
      v[m] = make_double2(x[m] * (s * w.x - r * w.y), x[m] * (s * w.y + r * w.x));
    } else {
      const cpx d = make_double2(s * (x[m] - x[m + M]), r * x[m] - (r + N) * x[m + M]);
      v[m] = cmul(d, w);
    }
    w = cmul(w, tw.wstep());                                        // W_FD^64
  }
  if constexpr (LONG) fft_forward<N>(v, img, tw, lane);
  else fft_forward_nz<N>(v, img, tw, lane, nzc);
  wave_sync();
#pragma unroll
  for (int m = M / 2; m < M; ++m) img[lane + 64 * m] = v[m];
  wave_sync();
#pragma unroll
  for (int m = 0; m < M / 2; ++m) {
    const cpx pt = img[N - 1 - (lane + 64 * m)];                    // O[N - 1 - j]
    co[m] += (v[m].x * pt.y + v[m].y * pt.x) * scale;
  }
  wave_sync();
}

// GetAperiodicity (d4c.cpp:325-333): row[i] = 10^(y(f_i) / 20), y the linear interpolation of the knot values
// kv[0 .. nap + 1] = {-60, coarse..., -1e-12} at {0, 3000, ..., 3000 nap, fs / 2}, f_i = i fs / out_fft.  Inside a
// segment y is linear in the bin number, so the row is a geometric sequence there: a lane's bins are 64 apart, so
// it needs one exp() when it enters a segment and a multiplication by that segment's ratio^64 per bin after that
// (one exp() per segment and wave) -- 513 bins cost 2 + 2 exp() calls per lane at 16 kHz instead of 9.  The products
// drift from the direct value by at most (bins of a segment / 64) roundings, 1e-15 relative.
// The segment index is floor(f / 3000) taken with a reciprocal: on an exact knot it may pick the segment to the left
// with s = 1, which is the same point of the (continuous) interpolant.
template <class Knot>
__device__ __forceinline__ void d4c_write_row(Knot kv, int nap, int fs, int out_fft, int out_bins, int lane,
                                              double* __restrict__ row) {
  const double bin_hz = (double)fs / out_fft;
  const double last_w = fs / 2.0 - nap * kFreqInterval;
  const double inv_last = 1.0 / last_w;
  const double ln10_20 = 2.302585092994045684 / 20.0;
  int seg = -1;
  bool have_ratio = false;
  double cur = 0.0, ratio64 = 1.0, slope = 0.0;
  for (int i = lane; i < out_bins; i += 64) {
    const double f = (double)i * bin_hz;
    int kk = (int)(f * (1.0 / kFreqInterval));
    kk = kk > nap ? nap : kk;
    if (kk != seg) {                                             // first bin of this lane in segment kk
      seg = kk;
      have_ratio = false;
      const double x0 = kk * kFreqInterval;
      const double inv_w = kk == nap ? inv_last : 1.0 / kFreqInterval;
      const double y0 = kv(kk), y1 = kv(kk + 1);
      slope = inv_w * (y1 - y0);
      cur = wm_exp((y0 + (f - x0) * slope) * ln10_20);
    } else {
      if (!have_ratio) {                                         // second bin in the segment: the step of 64 bins
        ratio64 = wm_exp(64.0 * bin_hz * slope * ln10_20);
        have_ratio = true;
      }
      cur *= ratio64;
    }
    row[i] = cur;
  }
}

// The frames the two instantiations of d4c_kernel work on (both: d4c.cpp:380).  The usual one takes the frames
// whose smoothing mirror fits FD / 16 bins (f0 < fs / 16) AND whose 4-period window fits FD / 2 samples (f0 above
// ~4 fs / FD): every frame behind Dio / Harvest + StoneMask at their default 71-800 Hz range from 12.8 kHz up.
// The RARE one takes the rest of what the reference defines.
__host__ __device__ inline bool d4c_is_usual(double cf0, int fd, int fs) {
  // fd 4096 (d4c_big.hpp) takes any window length; the one-kernel form only windows of at most fd / 2 samples
  return d4c_mirror_bins(cf0, fd, fs) <= fd / 16 && (fd >= 4096 || 2 * matlab_round(2.0 * fs / cf0) + 1 <= fd / 2);
}
struct D4cRunUsualPred {
  const double* f0;
  const double* ap0;
  double threshold;
  int fd, fs;
  __device__ bool operator()(int i) const {
    const double v = f0[i];
    if (v == 0.0 || ap0[i] <= threshold) return false;       // d4c.cpp:380 as written: a NaN ratio does not skip
    return d4c_is_usual(v > kFloorF0D4C ? v : kFloorF0D4C, fd, fs);
  }
};
struct D4cRunRarePred {
  const double* f0;
  const double* ap0;
  double threshold;
  int fd, fs;
  __device__ bool operator()(int i) const {
    const double v = f0[i];
    if (v == 0.0 || ap0[i] <= threshold) return false;       // d4c.cpp:380 as written: a NaN ratio does not skip
    const double cf0 = v > kFloorF0D4C ? v : kFloorF0D4C;
    // beyond fd / 2 mirror bins the reference reads past its spectrum (common.cpp:62-68, :85-92): default row stays
    return !d4c_is_usual(cf0, fd, fs) && d4c_mirror_bins(cf0, fd, fs) <= fd / 2;
  }
};

// FD = fft_size_d4c.  Rows of `ap` have out_bins = fft_size/2+1 entries (CheapTrick's size).
// One wavefront per frame (fft.hpp): registers + one LDS region that is FFT image, spectrum with margins and
// selection scratch in turn.  RARE = false: the frames d4c_is_usual() accepts (short window, narrow mirror), at
// the register and LDS budget of two waves per SIMD; RARE = true: the others (one wave per SIMD).
template <int FD, int WAVES, bool RARE>
__global__ __launch_bounds__(64, WAVES) void d4c_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int* __restrict__ frame_utt, const double* __restrict__ tpos, const double* __restrict__ f0,
    const double* __restrict__ ap0, const int* __restrict__ rng_off, const uint32_t* __restrict__ rtab,
    int fs_arg, double threshold, D4CTables tab, int out_fft_arg, int64_t total_frames, const int* __restrict__ perm,
    const int* __restrict__ n_listed, double* __restrict__ ap) {
  constexpr int N = FD / 2, M = N / 64, H = FD / 2, MB = M + 1;
  constexpr int kBM = D4CMargin<FD, RARE>::kBM;
  constexpr int kImg = 2 * FftLds<N>::kElems;
  constexpr int kRegion = SmoothCfg<H, kBM>::kRegion;
  constexpr int kHeads = (MB + 3) * 64;
  constexpr int kTot = kImg > kRegion ? (kImg > kHeads ? kImg : kHeads) : (kRegion > kHeads ? kRegion : kHeads);
  static_assert(kBM % 2 == 0, "the spectrum starts on a 16-byte boundary");
  __shared__ __attribute__((aligned(16))) double smem[kTot];
  double* arr = smem + kBM;                   // [-kBM .. H + kBM] spectrum-domain array with mirror margins
  cpx* img = reinterpret_cast<cpx*>(smem);    // FFT image (aliases it)

  const int lane0 = threadIdx.x;
  const int n_run = *n_listed;
  // the usual case of the RARE launch is an empty list: leave before the twiddle tables are built (the launch took
  // 64 us of every pass of configs[1] doing that and nothing else)
  if (RARE && n_run == 0) return;
  FftTw<N> tw;
  tw.init(lane0);
  const int out_bins = out_fft_arg / 2 + 1;

  if (!RARE) {
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
    const int lane = opaque_lane(lane0);
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
    double ce[M / 2 + 1], co[M / 2];
#pragma unroll
    for (int m = 0; m < M / 2; ++m) ce[m] = co[m] = 0.0;
    ce[M / 2] = 0.0;
#pragma unroll 1
    for (int side = 0; side < 2; ++side) {
      const double cpos = uniform_d(side == 0 ? pos - 0.25 / cf0 : pos + 0.25 / cf0);
      const FrameGeom fg = frame_geom(fs, cf0, cpos, 4.0);
      if (!RARE || fg.L <= N) d4c_centroid<N, false>(xu, xl, fg, rtab, roff + side * Lw, tw, img, lane, ce, co);
      else d4c_centroid<N, true>(xu, xl, fg, rtab, roff + side * Lw, tw, img, lane, ce, co);
    }
    WM_PHASE_MARK(1)                                                                  // two centroids
    {
      cpx* arr2 = reinterpret_cast<cpx*>(arr);
#pragma unroll
      for (int m = 0; m < M / 2; ++m) arr2[lane + 64 * m] = make_double2(ce[m], co[m]);   // bins 2 j, 2 j + 1
      if (lane == 0) arr[H] = ce[M / 2];
    }
    wave_sync();
    dc_correction_margin<H, kBM>(arr, cf0, fs, FD, lane);  // d4c.cpp:139
    // The centroid waits in the smoothing's own layout (BI consecutive bins per lane; bins past H: whatever the margin
    // holds, they only ever meet other bins past H), so that the elementwise steps between the smoothings below ride on
    // the smoothings' stores.
    constexpr int BI = SmoothCfg<H, kBM>::kBi;
    double sc[BI];
#pragma unroll
    for (int q = 0; q < BI; ++q) sc[q] = arr[lane * BI + q];
    wave_sync();
    WM_PHASE_MARK(2)                                                                  // DC correction of the centroid

    // ---- GetSmoothedPowerSpectrum (d4c.cpp:148-164) ----
    {
      cpx v[M];
      const FrameGeom fg = frame_geom(fs, cf0, pos, 4.0);
      frame_packed<kHann, false, M>(xu, xl, fg, rtab, roff + 2 * Lw, lane, v);
      WM_PHASE_MARK(3)                                                                // Hann frame
      {
        cpx xk[M / 2], xr[M / 2], xh;                    // the half spectrum stays in registers (rfft_split_pairs)
        rfft_forward_nz_pairs<N>(v, img, tw, lane, RARE ? M : (fg.L + 127) >> 7, xk, xr, xh);
#pragma unroll
        for (int m = 0; m < M / 2; ++m) {
          arr[lane + 64 * m] = xk[m].x * xk[m].x + xk[m].y * xk[m].y;
          arr[N - (lane + 64 * m)] = xr[m].x * xr[m].x + xr[m].y * xr[m].y;
        }
        if (lane == 0) arr[N / 2] = xh.x * xh.x + xh.y * xh.y;
      }
      wave_sync();
    }
    WM_PHASE_MARK(4)                                                                  // its transform and power
    dc_correction_margin<H, kBM>(arr, cf0, fs, FD, lane);
    // ---- GetStaticGroupDelay (d4c.cpp:170-186): centroid / smoothed power, smoothed by f0 / 2, minus that smoothed by f0 ----
    linear_smoothing_margin<H, kBM>(arr, cf0, fs, FD, lane, [&](int q, double s) { return sc[q] / s; });
    WM_PHASE_MARK(5)                                                                  // DC correction + smoothing
    double gd[BI];
    linear_smoothing_margin<H, kBM>(arr, cf0 / 2.0, fs, FD, lane, [&](int q, double s) { return gd[q] = s; });
    linear_smoothing_margin<H, kBM>(arr, cf0, fs, FD, lane, [&](int q, double s) { return gd[q] -= s; });
    WM_PHASE_MARK(6)                                                                  // group delay: two smoothings

    // ---- GetCoarseAperiodicity (d4c.cpp:192-223) ----
    const int wl = tab.window_length;
    const int bnd = matlab_round(FD * 8.0 / wl);
    const int hwl = wl / 2;
    double coarse[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int band = 0; band < tab.nap; ++band) {
      if (band > 0) {                 // the first band finds the group delay where the last smoothing stored it
#pragma unroll
        for (int q = 0; q < BI; ++q) arr[lane * BI + q] = gd[q];
        wave_sync();
      }
      const int center = (int)(kFreqInterval * (band + 1) * FD / fs);
      cpx v[M];
      // Window values in groups of four register pairs: the loads of a group are issued together with clamped
      // indices and the predicate goes on the value (behind a per-lane branch every pair was a trip to memory of its
      // own); groups wholly beyond the window are skipped by a wave-uniform branch.
      constexpr int GW = M < 4 ? M : 4;
      const int lw = opaque_lane(lane);       // per band: the window values are not kept in registers across the bands
#pragma unroll
      for (int g0 = 0; g0 < M; g0 += GW) {
        if (128 * g0 < wl) {
          double na[GW], nb[GW], ga[GW], gb[GW];
#pragma unroll
          for (int r = 0; r < GW; ++r) {
            const int i0 = 2 * (lw + 64 * (g0 + r));
            const int j0 = imin(i0, wl - 1), j1 = imin(i0 + 1, wl - 1);
            na[r] = tab.nuttall[j0];
            nb[r] = tab.nuttall[j1];
            ga[r] = arr[center - hwl + j0];
            gb[r] = arr[center - hwl + j1];
          }
#pragma unroll
          for (int r = 0; r < GW; ++r) {
            const int i0 = 2 * (lw + 64 * (g0 + r));
            v[g0 + r] = make_double2(i0 < wl ? ga[r] * na[r] : 0.0, i0 + 1 < wl ? gb[r] * nb[r] : 0.0);
          }
        } else {
#pragma unroll
          for (int r = 0; r < GW; ++r) v[g0 + r] = make_double2(0.0, 0.0);
        }
      }
      WM_PHASE_MARK(7)                                                                // band window
      double p[MB];
      double tot = 0.0;
      {
        cpx xk[M / 2], xr[M / 2], xh;                    // which lane holds which bin is all the same to what follows
        rfft_forward_nz_pairs<N>(v, img, tw, lane, (wl + 127) >> 7, xk, xr, xh);
        WM_PHASE_MARK(8)                                                              // band transform
#pragma unroll
        for (int m = 0; m < M / 2; ++m) {
          p[2 * m] = xk[m].x * xk[m].x + xk[m].y * xk[m].y;
          p[2 * m + 1] = xr[m].x * xr[m].x + xr[m].y * xr[m].y;
          tot += p[2 * m] + p[2 * m + 1];
        }
        p[M] = -1.0;
        if (lane == 0) {
          p[M] = xh.x * xh.x + xh.y * xh.y;
          tot += p[M];
        }
      }
      tot = wave_sum(tot);
      // Sum of all but the (bnd + 1) largest bins: cum[h - bnd - 1] keeps the h - bnd smallest
      // (d4c.cpp:215-220).  Each lane sorts its own bins once (Batcher network, static indices) and
      // parks the sorted column in LDS (the FFT image is free again); peel_largest() says how many
      // of its largest bins every lane gives up.
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
      for (int m = 0; m < MB; ++m) heads[m * 64 + lane] = p[m];
#pragma unroll
      for (int m = MB; m < MB + 3; ++m) heads[m * 64 + lane] = -1.0;     // exhausted
      const int taken = peel_largest(heads, bnd + 1, lane);   // own column only: no barrier needed
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
    d4c_write_row([&](int k) { return smem[k]; }, tab.nap, fs, out_fft, out_bins, lane, row);
    wave_sync();
    WM_PHASE_MARK(10)                                                                 // output row
  }
  WM_PHASE_FLUSH(0)
}

}  // namespace wm
#include "d4c_big.hpp"
#include "d4c_q.hpp"
#include "d4c_wide.hpp"
namespace wm {

// fft_size_d4c = 4096 / 8192: the four-kernel form of d4c_big.hpp for the usual frames
// Grid of the RARE launch: one round of resident workgroups, not `oversub` rounds -- its list is empty at the default
// f0 range, and 6 144 workgroups of 33 KB LDS took 64 us to come and go (1 024: 12 us).
static inline int rare_grid(const Context& c, int persistent) { return imax(1, persistent / imax(1, c.oversub)); }

template <int FD>
static int launch_d4c_big(Batch& b, const double* d_x, const double* d_t, const double* d_f0, D4CTables tab,
                          double* d_ap) {
  typedef D4cBig<FD> G;
  Context& c = *b.ctx;
  hipStream_t st = c.stream;
  const int64_t tf = b.total_f;
  if (tab.window_length >= FD / 4) return WM_ERR_UNSUPPORTED;    // d4cb_band_kernel: taps fill half the packed operand at most
  const int g1 = persistent_grid(c, d4cb_centroid_kernel<FD>, 64, (int64_t)1 << 40);
  const int g2 = persistent_grid(c, d4cb_spectrum_kernel<FD>, 64, (int64_t)1 << 40);
  const int g3 = persistent_grid(c, d4cb_band_kernel<FD>, 64, (int64_t)1 << 40);
  // The per-frame arrays between the kernels (C: the centroid by quarter, GD: the group delay; 33 KB per frame) hold
  // kD4cBigChunk listed frames at most and are reused chunk after chunk: the workspace is bounded whatever the batch
  // (1.5 M frames at 48 kHz used to ask for 74 GB).  The count of listed frames lives on the device, so the chunk
  // launches cover the whole frame range and the kernels clip to what is listed.
  constexpr int64_t kD4cBigChunk = 128 * 1024;
  const int64_t chunk = tf < kD4cBigChunk ? (tf > 0 ? tf : 1) : kD4cBigChunk;
  const size_t ws_rows = (size_t)g1;                               // scratch rows per workgroup of the centroid kernel
  if (!b.d_d4c_big) {
    const size_t per = (size_t)4 * G::kQ + (size_t)G::kRow;
    int rc = wm_check(dev_alloc(&b.d_d4c_big, sizeof(double) * (per * (size_t)chunk + 8 * (size_t)(tf > 0 ? tf : 1) +
                                                                        ws_rows * D4cBigWs<FD>::kDoubles)));
    if (rc) return rc;
  }
  double* C = b.d_d4c_big;
  double* GD = C + (size_t)chunk * 4 * G::kQ;
  double* COARSE = GD + (size_t)chunk * G::kRow;
  double* WS = COARSE + (size_t)tf * 8;
  const int fs = b.p.fs;
  const int* perm = (const int*)b.d_perm_d4c;
  const int* nl = (const int*)b.d_part_n_d4c;
  for (int64_t begin = 0; begin < tf; begin += chunk) {
    const int64_t left = tf - begin < chunk ? tf - begin : chunk;
    const int capc = (int)(left < (int64_t)c.frame_grid ? left : (int64_t)c.frame_grid);
    hipLaunchKernelGGL(d4cb_centroid_kernel<FD>, dim3(imin(capc, g1)), dim3(64), 0, st, d_x, b.d_x_off, b.d_x_len,
                       b.d_frame_utt, d_t, d_f0, b.d_rng_off_d4c, c.d_rng, fs, perm, nl, (int)begin, (int)chunk, WS, C);
    hipLaunchKernelGGL(d4cb_spectrum_kernel<FD>, dim3(imin(capc, g2)), dim3(64), 0, st, d_x, b.d_x_off, b.d_x_len,
                       b.d_frame_utt, d_t, d_f0, b.d_rng_off_d4c, c.d_rng, fs, perm, nl, (int)begin, (int)chunk,
                       (const double*)C, GD);
    const int64_t tasks = left * tab.nap;
    const int cap3 = (int)(tasks < (int64_t)c.frame_grid * 4 ? tasks : (int64_t)c.frame_grid * 4);
    hipLaunchKernelGGL(d4cb_band_kernel<FD>, dim3(imax(1, imin(cap3, g3))), dim3(64), 0, st, d_f0, fs, tab, perm, nl,
                       (int)begin, (int)chunk, (const double*)GD, COARSE);
  }
  const int cap = (int)(tf < (int64_t)c.frame_grid ? tf : (int64_t)c.frame_grid);
  const int64_t blocks = (tf + 3) / 4;
  hipLaunchKernelGGL(d4cb_output_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, fs, tab,
                     b.p.fft_size, tf, perm, nl,
                     D4cRunRarePred{d_f0, (const double*)b.d_ap0, b.p.d4c_threshold, FD, fs},
                     (const double*)COARSE, d_ap);
  return wm_check(hipGetLastError());
}

// D4C in three steps.  d4c_prepare(): everything that needs f0 (and the waveform, for the LoveTrain ratio) but no
// result of CheapTrick -- the randn offsets, the LoveTrain stage and the three frame lists -- on the context's stream.
// d4c_rare(): the RARE instantiation over its list (normally empty).  d4c_run(): the transforms of the usual frames.  launch_analyze_synthesize() puts the first on its second stream beside CheapTrick (ten
// short dependent launches, 0.1 ms of an otherwise idle machine between CheapTrick and the D4C kernel); the lists and
// offsets are D4C's own arrays (`*_d4c`), so nothing of CheapTrick's is touched.
static int d4c_tables(Batch& b, D4CTables& tab) {
  const int fs = b.p.fs;
  const int FD = d4c_fft_size(fs);
  tab.nuttall = b.d_d4c_window;
  tab.window_length = (int)(kFreqInterval * FD / fs) * 2 + 1;
  double lim = fs / 2.0 - kFreqInterval;
  tab.nap = (int)((kUpperLimit < lim ? kUpperLimit : lim) / kFreqInterval);
  // no band at all below 12 kHz (fs / 2 - 3000 < 3000): the reference then interpolates between its two end knots only
  if (tab.nap < 0 || tab.nap > 6) return WM_ERR_UNSUPPORTED;
  return WM_OK;
}

int d4c_prepare(Batch& b, const double* d_x, const double* d_t, const double* d_f0) {
  Context& c = *b.ctx;
  hipStream_t st = c.stream;
  const int fs = b.p.fs;
  const int FD = d4c_fft_size(fs), FL = lovetrain_fft_size(fs);
  // D4C's own transform and LoveTrain's are sized independently (d4c.cpp:344-346, :261-263): they differ for fs in
  // [12.0, 13.6), [24.1, 27.3) and [48.1, 54.6) kHz (2048 / 1024, 4096 / 2048, 8192 / 4096).  LoveTrain's transform is
  // only run for a threshold above zero.
  if (FD != 1024 && FD != 2048 && FD != 4096 && FD != 8192) return WM_ERR_UNSUPPORTED_FFT;
  if (b.p.d4c_threshold > 0.0 && FL != 1024 && FL != 2048 && FL != 4096 && FL != 8192) return WM_ERR_UNSUPPORTED_FFT;
  int rc = c.ensure_rng(b.rng_bound_d4c());
  if (rc) return rc;
  // Nuttall window table for GetCoarseAperiodicity (d4c.cpp:356-359, common.cpp:113-121)
  const int wl = (int)(kFreqInterval * FD / fs) * 2 + 1;
  if (!b.d_d4c_window) {
    double* dw = nullptr;
    for (const auto& e : c.nuttall_windows)
      if (e.first == wl) dw = e.second;
    if (!dw) {                                           // once per context and sampling rate
      std::vector<double> w((size_t)wl);
      for (int i = 0; i < wl; ++i) {
        double tmp = i / (wl - 1.0);
        w[(size_t)i] = 0.355768 - 0.487396 * cos(2.0 * kPi * tmp) + 0.144232 * cos(4.0 * kPi * tmp) -
                       0.012604 * cos(6.0 * kPi * tmp);
      }
      rc = wm_check(dev_alloc(&dw, sizeof(double) * (size_t)wl));
      if (rc) return rc;
      rc = wm_check(hipMemcpyAsync(dw, w.data(), sizeof(double) * (size_t)wl, hipMemcpyHostToDevice, st));
      if (!rc) rc = wm_check(hipStreamSynchronize(st));   // w is a stack-lifetime buffer
      if (rc) { dev_free(dw); return rc; }
      c.nuttall_windows.push_back(std::make_pair(wl, dw));
    }
    if (!b.d_utt_total) rc = wm_check(dev_alloc(&b.d_utt_total, sizeof(int) * (size_t)b.n_utt));
    if (!rc && !b.d_perm2)                               // StoneMask may have taken it already
      rc = wm_check(dev_alloc(&b.d_perm2, sizeof(int) * (size_t)(b.total_f > 0 ? b.total_f : 1)));
    if (rc) return rc;
    b.d_d4c_window = dw;
  }
  D4CTables tab;
  rc = d4c_tables(b, tab);
  if (rc) return rc;

  const int64_t tf = b.total_f;
  const int grid = (int)(tf < (int64_t)c.frame_grid ? tf : (int64_t)c.frame_grid);
  hipLaunchKernelGGL(d4c_offsets_kernel<0>, dim3(b.n_utt), dim3(256), 0, st, d_f0, (const double*)nullptr,
                     b.d_f_off, fs, b.p.d4c_threshold, b.d_utt_total, b.d_rng_off2);
  launch_partition(st, VoicedPred{d_f0}, (int)tf, b.d_part_cnt_d4c, b.d_perm_d4c, b.d_part_n_d4c);
#define WM_LT_CASE(FF)                                                                                     \
  case FF: {                                                                                               \
    const int per_ = persistent_grid(c, d4c_lovetrain_kernel<FF>, 64, (int64_t)1 << 40);            \
    hipLaunchKernelGGL(d4c_lovetrain_kernel<FF>, dim3(imin(grid, per_)), dim3(64), 0, st, d_x, b.d_x_off,  \
                       b.d_x_len, b.d_frame_utt, d_t, d_f0, b.d_rng_off2, c.d_rng, fs, tf,                \
                       (const int*)b.d_perm_d4c, (const int*)b.d_part_n_d4c, b.d_ap0);                             \
  } break;
  {
    TimedScope ts_(b.ctx, "d4c_lovetrain_kernel");
    if (b.p.d4c_threshold <= 0.0) {
      // every voiced frame passes whatever the ratio is: see d4c_lovetrain_all_pass_kernel
      hipLaunchKernelGGL(d4c_lovetrain_all_pass_kernel, dim3((unsigned)((tf + 255) / 256)), dim3(256), 0, st, d_f0, tf,
                         b.d_ap0);
    } else {
      switch (FL) {
        WM_LT_CASE(1024)
        WM_LT_CASE(2048)
        WM_LT_CASE(4096)
        case 8192: {
          const int per_ = persistent_grid(c, d4cb_lovetrain_kernel<8192>, 64, (int64_t)1 << 40);
          hipLaunchKernelGGL(d4cb_lovetrain_kernel<8192>, dim3(imin(grid, per_)), dim3(64), 0, st, d_x, b.d_x_off, b.d_x_len,
                             b.d_frame_utt, d_t, d_f0, b.d_rng_off2, c.d_rng, fs, tf, (const int*)b.d_perm_d4c,
                             (const int*)b.d_part_n_d4c, b.d_ap0);
        } break;
      }
    }
  }
#undef WM_LT_CASE
  hipLaunchKernelGGL(d4c_offsets_kernel<1>, dim3(b.n_utt), dim3(256), 0, st, d_f0, (const double*)b.d_ap0,
                     b.d_f_off, fs, b.p.d4c_threshold, b.d_utt_total, b.d_rng_off_d4c);
  launch_partition(st, D4cRunUsualPred{d_f0, b.d_ap0, b.p.d4c_threshold, FD, fs}, (int)tf, b.d_part_cnt_d4c, b.d_perm_d4c,
                   b.d_part_n_d4c);
  // the rare frames (f0 >= fs / 16, or a window longer than FD / 2 samples) are listed separately for the
  // wide-margin, long-frame instantiation; an empty list costs that launch a few microseconds
  launch_partition(st, D4cRunRarePred{d_f0, b.d_ap0, b.p.d4c_threshold, FD, fs}, (int)tf, b.d_part_cnt_d4c, b.d_perm2,
                   b.d_part_n_d4c + 1);
  return wm_check(hipGetLastError());
}

int d4c_rare(Batch& b, const double* d_x, const double* d_t, const double* d_f0, double* d_ap) {
  Context& c = *b.ctx;
  hipStream_t st = c.stream;
  const int fs = b.p.fs;
  const int FD = d4c_fft_size(fs);
  D4CTables tab;
  int rc = d4c_tables(b, tab);
  if (rc) return rc;
  const int64_t tf = b.total_f;
  const int grid = (int)(tf < (int64_t)c.frame_grid ? tf : (int64_t)c.frame_grid);
  // The RARE launch on its own: its rows are its own (the other kernels leave them alone), so it needs nothing of
  // d4c_run() but the lists -- and behind the usual kernel it cost 62 us of every pass with its list empty (the default
  // f0 range), for a kernel whose every wave leaves at its fourth instruction.  Its waves need a SIMD to themselves:
  // beside another kernel the launch sits until that one drains, so the one-call forms give it a stream of its own.  At 8192 there is no one-wavefront form (its
  // transform would be 4096 complex points on one wavefront): frames with f0 >= fs / 16 (6 kHz at 96 kHz) go to
  // d4c_wide_kernel, a workgroup per frame with direct DFTs (their windows are at most 65 samples).
#define WM_D4C_RARE(FF)                                                                                   \
  case FF: {                                                                                              \
    const int per2_ = persistent_grid(c, d4c_kernel<FF, 1, true>, 64, (int64_t)1 << 40);           \
    hipLaunchKernelGGL((d4c_kernel<FF, 1, true>), dim3(imin(grid, rare_grid(c, per2_))), dim3(64), 0, st, d_x, \
                       b.d_x_off, b.d_x_len, b.d_frame_utt, d_t, d_f0, (const double*)b.d_ap0,            \
                       b.d_rng_off_d4c, c.d_rng, fs, b.p.d4c_threshold, tab, b.p.fft_size, tf,            \
                       (const int*)b.d_perm2, (const int*)(b.d_part_n_d4c + 1), d_ap);                    \
  } break;
  switch (FD) {
    WM_D4C_RARE(1024)
    WM_D4C_RARE(2048)
    WM_D4C_RARE(4096)
    case 8192: {               // one workgroup per frame, everything in LDS, direct DFTs (d4c_wide.hpp)
      const int lds = (int)(sizeof(double) * D4cWideLds::doubles(8192));
      allow_dynamic_lds(c, d4c_wide_kernel<8192>, lds);
      hipLaunchKernelGGL(d4c_wide_kernel<8192>, dim3(imin(grid, c.num_cu)), dim3(kWideThreads), lds, st, d_x, b.d_x_off,
                         b.d_x_len, b.d_frame_utt, d_t, d_f0, b.d_rng_off_d4c, c.d_rng, fs, tab, b.p.fft_size,
                         (const int*)b.d_perm2, (const int*)(b.d_part_n_d4c + 1), d_ap);
    } break;
  }
#undef WM_D4C_RARE
  return wm_check(hipGetLastError());
}

int d4c_run(Batch& b, const double* d_x, const double* d_t, const double* d_f0, double* d_ap) {
  Context& c = *b.ctx;
  hipStream_t st = c.stream;
  const int fs = b.p.fs;
  const int FD = d4c_fft_size(fs);
  D4CTables tab;
  int rc = d4c_tables(b, tab);
  if (rc) return rc;
  const int64_t tf = b.total_f;
  const int grid = (int)(tf < (int64_t)c.frame_grid ? tf : (int64_t)c.frame_grid);
#define WM_D4C_CASE(FF, WV)                                                                               \
  case FF: {                                                                                              \
    const int per_ = persistent_grid(c, d4c_kernel<FF, WV, false>, 64, (int64_t)1 << 40);          \
    hipLaunchKernelGGL((d4c_kernel<FF, WV, false>), dim3(imin(grid, per_)), dim3(64), 0, st, d_x,         \
                       b.d_x_off, b.d_x_len, b.d_frame_utt, d_t, d_f0, (const double*)b.d_ap0,            \
                       b.d_rng_off_d4c, c.d_rng, fs, b.p.d4c_threshold, tab, b.p.fft_size, tf,                \
                       (const int*)b.d_perm_d4c, (const int*)b.d_part_n_d4c, d_ap);                               \
  } break;
  // fft_size_d4c 2048 (the headline's 16 kHz): WORLD_MI355_D4C_Q=1 selects the three-waves-per-SIMD form on the
  // 512-point engine (d4c_q.hpp).  Built in round 5 because two waves issue a vector instruction every 6.0 cycles where
  // the SIMD could take one every 4; measured (profiles/r05_b_*): a third wave brings that to 5.8 -- whatever the waves
  // wait for, another wave does not hide it -- for 5 % more instructions, at a clock 12 % lower (2.26 -> 1.98 GHz: the
  // denser issue costs power, and the kernels that follow inherit the lower clock).  5.05 -> 5.8 ms: off by default.
  static const bool use_q = getenv("WORLD_MI355_D4C_Q") && atoi(getenv("WORLD_MI355_D4C_Q")) != 0;
#define WM_D4CQ_LAUNCH(OB)                                                                                \
  {                                                                                                       \
    const int per_ = persistent_grid(c, d4cq_kernel<2048, OB>, 64, (int64_t)1 << 40);                     \
    hipLaunchKernelGGL((d4cq_kernel<2048, OB>), dim3(imin(grid, per_)), dim3(64), 0, st, d_x, b.d_x_off,  \
                       b.d_x_len, b.d_frame_utt, d_t, d_f0, (const double*)b.d_ap0, b.d_rng_off_d4c,      \
                       c.d_rng, fs, b.p.d4c_threshold, tab, b.p.fft_size, tf, (const int*)b.d_perm_d4c,   \
                       (const int*)b.d_part_n_d4c, d_ap);                                                 \
  }
  {
    TimedScope ts_(b.ctx, "d4c_kernel");
    switch (FD) {
      WM_D4C_CASE(1024, 2)
      case 2048: {
        if (use_q && tab.nap == 1) WM_D4CQ_LAUNCH(true)
        else if (use_q) WM_D4CQ_LAUNCH(false)
        else {
          switch (FD) { WM_D4C_CASE(2048, 2) }
        }
      } break;
      case 4096: {              // four kernels on the 1024-point transform (d4c_big.hpp)
        rc = launch_d4c_big<4096>(b, d_x, d_t, d_f0, tab, d_ap);
        if (rc) return rc;
      } break;
      case 8192: {              // the same on the 2048-point transform (fs above 48.1 kHz)
        rc = launch_d4c_big<8192>(b, d_x, d_t, d_f0, tab, d_ap);
        if (rc) return rc;
      } break;
    }
  }
#undef WM_D4C_CASE
#undef WM_D4CQ_LAUNCH
  return wm_check(hipGetLastError());
}

int launch_d4c(Batch& b, const double* d_x, const double* d_t, const double* d_f0, double* d_ap) {
  int rc = d4c_prepare(b, d_x, d_t, d_f0);
  rc = rc ? rc : d4c_rare(b, d_x, d_t, d_f0, d_ap);
  return rc ? rc : d4c_run(b, d_x, d_t, d_f0, d_ap);
}

#ifdef WM_PHASE
int phase_read_d4c(unsigned long long* out32) { return wm_phase_read(out32); }
#endif

}  // namespace wm

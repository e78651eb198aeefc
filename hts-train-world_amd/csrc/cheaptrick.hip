// cheaptrick.hip -- CheapTrick spectral-envelope estimation, one wavefront per frame.
//
// Replaces CheapTrick / CheapTrickGeneralBody and everything below it
// (externs/WORLD_v2/src/cheaptrick.cpp:22-228) for a whole batch of utterances:
// grid-stride over (utterance, frame) pairs, each frame handled by one 64-lane
// workgroup that keeps the frame's window, spectra and cepstrum in registers and
// LDS (3 real FFTs of fft_size, 1 blocked scan, log/lifter/exp fused).  The
// reference's global xorshift generator (matlabfunctions.cpp:247-277) is a
// table lookup at a per-frame offset computed by cheaptrick_offsets_kernel.
#include "batch.hpp"
#include "common.hpp"
#include "fastmath.hpp"
#include "fft.hpp"
#include "frame.hpp"
#include "partition.hpp"
#include "spectrum.hpp"

namespace wm {

// The f0 a frame is analysed with.
__host__ __device__ inline double ct_frame_f0(double f0v, int fs, int F) {
  const double f0_floor = 3.0 * fs / (F - 3.0);              // cheaptrick.cpp:196-198
  // f0 <= floor (cheaptrick.cpp:217); NaN too.  Above fs / 2 (and +Inf) DCCorrection indexes past its spectrum
  // in the reference (common.cpp:56-75: upper_limit > fft_size / 2): undefined there, the default f0 here.
  return !(f0v > f0_floor) || !(f0v <= fs / 2.0) ? kDefaultF0 : f0v;
}

// Per-utterance exclusive scan of each frame's randn consumption:
// (2*round(1.5 fs/f0')+1) for the window (cheaptrick.cpp:126-128) then fft_size/2+1
// for AddInfinitesimalNoise (:149-150).
__global__ __launch_bounds__(256) void cheaptrick_offsets_kernel(const double* __restrict__ f0,
                                                                 const int64_t* __restrict__ f_off,
                                                                 int fs, int fft_size, int* __restrict__ rng_off) {
  __shared__ int part[256];
  __shared__ int carry_s;
  const int u = blockIdx.x;
  const int64_t base = f_off[u];
  const int nf = (int)(f_off[u + 1] - base);
  if (threadIdx.x == 0) carry_s = 0;
  wave_sync();
  for (int start = 0; start < nf; start += 256) {
    const int i = start + threadIdx.x;
    int c = 0;
    if (i < nf) {
      const double cf0 = ct_frame_f0(f0[base + i], fs, fft_size);
      c = 2 * matlab_round(1.5 * fs / cf0) + 1 + fft_size / 2 + 1;
    }
    part[threadIdx.x] = c;
    wave_sync();
    for (int o = 1; o < 256; o <<= 1) {
      int tv = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
      wave_sync();
      part[threadIdx.x] += tv;
      wave_sync();
    }
    const int carry = carry_s;
    if (i < nf) rng_off[base + i] = carry + part[threadIdx.x] - c;
    wave_sync();
    if (threadIdx.x == 255) carry_s = carry + part[255];
    wave_sync();
  }
}

// Mirror margin of the spectrum array: LinearSmoothing (width 2 f0 / 3) mirrors int(width F / fs) + 1 bins and
// DCCorrection touches 2 + int(f0 F / fs).  Sized for any f0 up to fs / 2 (F / 3 + 2 bins) the array is 25 KB at
// fft 2048 and holds the kernel at 1.5 waves per SIMD, so the frames are split as in D4C: the usual ones
// (f0 below (F / 8 - 2) fs / F: 1.97 kHz at 16 kHz, 5.95 kHz at 48 kHz) run with F / 8 bins of margin, i.e. within
// the FFT image's own LDS, the others (WIDE) with the full margin.  Same code, same results.
template <int F, bool WIDE> struct CtMargin { static constexpr int kBM = WIDE ? (((F / 3 + 2) + 1) & ~1) : F / 8; };
struct CtUsualPred {
  const double* f0;
  int fs, F;
  __device__ bool operator()(int i) const { return 2 + (int)(ct_frame_f0(f0[i], fs, F) * F / fs) <= F / 8; }
};

// Three waves per SIMD at fft <= 1024, and no more: the kernel would fit four (125 registers), but in the one-call forms
// the f0-only kernels of Synthesis and D4C's preparation run beside it on other streams, in the registers a fourth wave
// would take -- with four they waited for CheapTrick to end and ran beside `d4c_kernel` instead (+0.3 ms there).
// (A clobbered v135 makes the allocation 136 registers; amdgpu_waves_per_eu(3, 3) does not raise it on gfx950.)
template <int F, bool WIDE>
__global__ __launch_bounds__(64, F >= 4096 ? 1 : (F >= 2048 ? 2 : 3)) void cheaptrick_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int* __restrict__ frame_utt, const double* __restrict__ tpos, const double* __restrict__ f0,
    const int* __restrict__ rng_off, const uint32_t* __restrict__ rtab, int fs_arg, double q1,
    int64_t total_frames, const int* __restrict__ perm, const int* __restrict__ n_usual,
    double* __restrict__ sp) {
  if constexpr (F <= 1024) asm volatile("; three waves per SIMD: see above" ::: "v135");
  constexpr int N = F / 2, M = N / 64, H = F / 2;
  constexpr int kBM = CtMargin<F, WIDE>::kBM;
  constexpr int kImg = 2 * FftLds<N>::kElems;                 // doubles
  constexpr int kRegion = SmoothCfg<H, kBM>::kRegion;
  constexpr int kTot = kImg > kRegion ? kImg : kRegion;
  __shared__ __attribute__((aligned(16))) double smem[kTot];
  double* pw = smem + kBM;                                     // [-kBM .. H + kBM] power / log spectrum with margins
  cpx* img = reinterpret_cast<cpx*>(smem);                     // FFT image | spectrum (aliases it)

  const int lane0 = threadIdx.x;
  FftTw<N> tw;
  tw.init(lane0);

  const int n_us = *n_usual;
  const int* list = WIDE ? perm + n_us : perm;                 // usual frames first, then the wide ones
  const int64_t n_list = WIDE ? total_frames - n_us : n_us;
  WM_PHASE_DECL
  WM_FOR_EACH_LISTED(frame, list, n_list) {
    WM_PHASE_MARK(0)
    const int lane = opaque_lane(lane0);
    const int fs = opaque_uniform(fs_arg);
    tw.fence();
    const int u = frame_utt[frame];
    const double cf0 = uniform_d(ct_frame_f0(f0[frame], fs, F));
    const int roff = rng_off[frame];
    cpx v[M];

    // ---- GetWindowedWaveform (cheaptrick.cpp:87-142), straight into the FFT operand ----
    const FrameGeom fg = frame_geom(fs, cf0, uniform_d(tpos[frame]), 3.0);
    frame_packed<kHann, true, M, (F <= 1024)>(x + x_off[u], x_len[u], fg, rtab, roff, lane, v);
    WM_PHASE_MARK(1)

    // ---- GetPowerSpectrum (cheaptrick.cpp:64-82) ----
    // The half spectrum never goes to LDS: every bin's power is stored from the lane that computed it (rfft_split_pairs).
    {
      cpx xk[M / 2], xr[M / 2], xh;
      rfft_forward_nz_pairs<N>(v, img, tw, lane, (fg.L + 127) >> 7, xk, xr, xh);   // the window reaches that many packed registers
#pragma unroll
      for (int m = 0; m < M / 2; ++m) {
        pw[lane + 64 * m] = xk[m].x * xk[m].x + xk[m].y * xk[m].y;
        pw[N - (lane + 64 * m)] = xr[m].x * xr[m].x + xr[m].y * xr[m].y;
      }
      if (lane == 0) pw[N / 2] = xh.x * xh.x + xh.y * xh.y;
      wave_sync();
    }
    WM_PHASE_MARK(2)
    dc_correction_margin<H, kBM, WIDE ? H : kBM>(pw, cf0, fs, F, lane);
    WM_PHASE_MARK(3)

    // ---- LinearSmoothing (cheaptrick.cpp:176) + AddInfinitesimalNoise (:147-151) + log (:39-40) ----
    // The noise and the logarithm ride on the smoothing's store, in the smoothing's layout (BI consecutive bins per lane;
    // what lands past bin H is margin).  The draws of a lane's bins are requested before the smoothing starts.
    {
      constexpr int BI = SmoothCfg<H, kBM>::kBi;
      uint32_t rv[BI];
#pragma unroll
      for (int q = 0; q < BI; ++q) rv[q] = rtab[roff + fg.L + imin(lane * BI + q, H)];
      linear_smoothing_margin<H, kBM>(pw, cf0 * 2.0 / 3.0, fs, F, lane, [&](int q, double s) {
        return wm_log(s + fabs((double)rv[q] / 268435456.0 - 6.0) * kEps);
      });
    }
    WM_PHASE_MARK(5)

    // ---- SmoothingWithRecovery (cheaptrick.cpp:22-57) ----
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int i0 = 2 * (lane + 64 * m), i1 = i0 + 1;
      v[m] = make_double2(pw[i0 <= H ? i0 : F - i0], pw[i1 <= H ? i1 : F - i1]);
    }
    {
      // The cepstrum stays in registers by pairs (quefrencies i and N - i in one lane), is liftered there, and goes
      // straight into the inverse transform's operand.
      cpx xk[M / 2], xr[M / 2], xh;
      rfft_forward_pairs<N>(v, img, tw, lane, xk, xr, xh);
      WM_PHASE_MARK(6)
      // lifters at quefrency i / fs: sin(pi f0 q) / (pi f0 q) and (1 - 2 q1) + 2 q1 cos(2 pi f0 q), cos(2 a) = 1 - 2 sin^2(a).
      // The angle pi f0 i / fs advances by a rotation per 64 bins; that of N - i follows from it and the (wave-uniform)
      // angle of N, itself the double of N / 2's, which lane 0 needs for the middle bin.
      const double a = cf0 / fs;
      CosGen g;
      g.init(a, lane, 64);
      double sh, ch;
      wm_sincospi(uniform_d(a * (N / 2)), &sh, &ch);
      const double sn = 2.0 * sh * ch, cn = 1.0 - 2.0 * sh * sh;
      auto lifter = [&](int i, double si) {
        const double quef = (double)i / fs;
        const double sl = si / (kPi * cf0 * quef);
        const double cl = (1.0 - 2.0 * q1) + 2.0 * q1 * (1.0 - 2.0 * si * si);
        return sl * cl;
      };
#pragma unroll
      for (int m = 0; m < M / 2; ++m) {
        const int i = lane + 64 * m;
        double lk = (1.0 - 2.0 * q1) + 2.0 * q1;                                  // i = 0
        if (i > 0) lk = lifter(i, g.s);
        const double lr = lifter(N - i, sn * g.c - cn * g.s);
        xk[m] = make_double2(xk[m].x * lk / F, 0.0);
        xr[m] = make_double2(xr[m].x * lr / F, 0.0);
        g.next();
      }
      xh = make_double2(xh.x * lifter(N / 2, sh) / F, 0.0);
      WM_PHASE_MARK(7)
      rfft_backward_pairs<N>(xk, xr, xh, v, img, tw, lane);
    }
    WM_PHASE_MARK(8)
    // x[2n], x[2n+1] for n = lane + 64 m: the first H + 1 samples are the envelope's logarithm
    double* row = sp + frame * (int64_t)(H + 1);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int i0 = 2 * (lane + 64 * m);
      if (i0 <= H) row[i0] = wm_exp(v[m].x);
      if (i0 + 1 <= H) row[i0 + 1] = wm_exp(v[m].y);
    }
    wave_sync();
    WM_PHASE_MARK(9)
  }
  WM_PHASE_FLUSH(0)
}

int launch_cheaptrick(Batch& b, const double* d_x, const double* d_t, const double* d_f0, double* d_sp) {
  hipStream_t st = b.ctx->stream;
  const int F = b.p.fft_size;
  int rc = b.ctx->ensure_rng(b.rng_bound_cheaptrick());
  if (rc) return rc;
  hipLaunchKernelGGL(cheaptrick_offsets_kernel, dim3(b.n_utt), dim3(256), 0, st, d_f0, b.d_f_off, b.p.fs, F,
                     b.d_rng_off);
  const int64_t tf = b.total_f;
  const int grid = (int)(tf < (int64_t)b.ctx->frame_grid ? tf : (int64_t)b.ctx->frame_grid);
  if (grid <= 0) return 0;
#define WM_CT_CASE(FF)                                                                                   \
  case FF: {                                                                                             \
    const int per_ = persistent_grid(*b.ctx, cheaptrick_kernel<FF, false>, 64, (int64_t)1 << 40); \
    const int perw_ = persistent_grid(*b.ctx, cheaptrick_kernel<FF, true>, 64, (int64_t)1 << 40); \
    hipLaunchKernelGGL((cheaptrick_kernel<FF, false>), dim3(imin(grid, per_)), dim3(64), 0, st, d_x,     \
                       b.d_x_off, b.d_x_len, b.d_frame_utt, d_t, d_f0, b.d_rng_off, b.ctx->d_rng,        \
                       b.p.fs, b.p.q1, tf, (const int*)b.d_perm, (const int*)b.d_part_n, d_sp);          \
    hipLaunchKernelGGL((cheaptrick_kernel<FF, true>), dim3(imin(grid, perw_)), dim3(64), 0, st, d_x,     \
                       b.d_x_off, b.d_x_len, b.d_frame_utt, d_t, d_f0, b.d_rng_off, b.ctx->d_rng,        \
                       b.p.fs, b.p.q1, tf, (const int*)b.d_perm, (const int*)b.d_part_n, d_sp);          \
  } break;
  {
  TimedScope ts_(b.ctx, "cheaptrick_kernel");
  launch_partition(st, CtUsualPred{d_f0, b.p.fs, F}, (int)tf, b.d_part_cnt, b.d_perm, b.d_part_n);
  switch (F) {
    WM_CT_CASE(512)       // fs <= 12.8 kHz (GetFFTSizeForCheapTrick, cheaptrick.cpp:191-194)
    WM_CT_CASE(1024)
    WM_CT_CASE(2048)
    WM_CT_CASE(4096)      // fs > 51.2 kHz
    default:
      return WM_ERR_UNSUPPORTED_FFT;
  }
  }
#undef WM_CT_CASE
  return wm_check(hipGetLastError());
}

#ifdef WM_PHASE
int phase_read_cheaptrick(unsigned long long* out32) { return wm_phase_read(out32); }
#endif

}  // namespace wm

// stonemask.hip -- StoneMask F0 refinement, one wavefront per frame.
//
// Replaces StoneMask / GetRefinedF0 / GetMeanF0 / GetTentativeF0 / FixF0
// (externs/WORLD_v2/src/stonemask.cpp:24-217).  The reference takes two real
// FFTs of 2^(2+int(log2(L))) points per frame but reads at most 8 bins of them
// (stonemask.cpp:101-106, :125, :130); here those bins are evaluated directly as
// windowed DFT sums over the L = 2*hw+1 samples (twiddles by per-lane rotation),
// so no FFT size binning is needed.
#include "batch.hpp"
#include "common.hpp"
#include "fft.hpp"
#include "partition.hpp"

namespace wm {

constexpr double kFloorF0StoneMask = 40.0;   // constantnumbers.h

// Sum_i a_i e^{-j 2 pi k i / n} for NB bins k[], both windows at once.  The windowed samples
// am[i] = x_i * main_window[i], ad[i] = x_i * diff_window[i] are read from LDS (FIRST = false) or
// produced on the way from the window mw[] and the samples xs[] and left in their place (FIRST).
// exp(-2 pi i k / kSmTwid), written once per batch with the function the kernel would otherwise call per bin
// (k / 2^n is exact, so a lookup returns the same bits)
constexpr int kSmTwid = 8192;
__global__ __launch_bounds__(256) void sm_twiddle_kernel(cpx* __restrict__ tw) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < kSmTwid) tw[k] = cis_neg2pi((double)k / (double)kSmTwid);
}

template <int NB, bool FIRST>
__device__ __forceinline__ void sm_bins(double* ad_mw, double* am_xs, int L, const int (&bin)[NB], int fftn,
                                        int lane, const cpx* __restrict__ twid, double (&pw)[NB],
                                        double (&num)[NB]) {
  cpx mainv[NB], diffv[NB], w[NB], st[NB];
  const double inv_fftn = 1.0 / fftn;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    mainv[b] = make_double2(0.0, 0.0);
    diffv[b] = make_double2(0.0, 0.0);
    // fftn is a power of two: the modulo is a mask; the twiddle at sample `lane` and its step of 64 samples
    // are table entries (transforms longer than the table compute them)
    if (fftn <= kSmTwid) {
      const int sc = kSmTwid / fftn;
      w[b] = twid[((bin[b] * lane) & (fftn - 1)) * sc];
      st[b] = twid[((bin[b] * 64) & (fftn - 1)) * sc];
    } else {
      w[b] = cis_neg2pi((double)((bin[b] * lane) & (fftn - 1)) * inv_fftn);
      st[b] = cis_neg2pi((double)((bin[b] * 64) & (fftn - 1)) * inv_fftn);
    }
  }
  double carry = 0.0;                                 // mw of the previous trip's last lane
  for (int i = lane; i < ((L + 63) & ~63); i += 64) {
    double am = 0.0, ad = 0.0;
    if (FIRST) {
      // differential window, stonemask.cpp:49-55; each lane overwrites only its own element and the
      // left neighbour of lane 0 travels in a register, so two LDS arrays are enough
      const bool in = i < L;
      const double m = in ? ad_mw[i] : 0.0;
      double left = (in && i > 0) ? ad_mw[i - 1] : 0.0;
      const double right = (i + 1 < L) ? ad_mw[i + 1] : 0.0;
      if (lane == 0) left = carry;
      carry = __shfl(m, 63, 64);
      const double xi = in ? am_xs[i] : 0.0;
      double d;
      if (i == 0) d = -right / 2.0;
      else if (i == L - 1) d = left / 2.0;
      else d = -(right - left) / 2.0;
      am = xi * m;
      ad = in ? xi * d : 0.0;
      if (in) { am_xs[i] = am; ad_mw[i] = ad; }
    } else if (i < L) {
      am = am_xs[i];
      ad = ad_mw[i];
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      mainv[b].x += am * w[b].x; mainv[b].y += am * w[b].y;
      diffv[b].x += ad * w[b].x; diffv[b].y += ad * w[b].y;
      w[b] = cmul(w[b], st[b]);
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    double mr = mainv[b].x, mi = mainv[b].y, dr = diffv[b].x, di = diffv[b].y;
    wave_sum4(mr, mi, dr, di);
    num[b] = mr * di - mi * dr;                                     // stonemask.cpp:159-160
    pw[b] = mr * mr + mi * mi;                                      // :161-162
  }
}

// FixF0, stonemask.cpp:96-117, from already-evaluated bins (wave-uniform values).  Lane h does harmonic h's
// division and square root -- done for every harmonic by every lane they are a fifth of the kernel -- and two
// row sums over the first NB lanes finish it.
template <int NB>
__device__ __forceinline__ double sm_fix(const double (&pw)[NB], const double (&num)[NB], const int (&bin)[NB],
                                         int fftn, int fs, int lane) {
  static_assert(NB <= 16, "one harmonic per lane of the first row");
  const double inv_fftn = 1.0 / fftn;                 // power of two: exact
  const double fs_over_2pi = fs / 2.0 / kPi;
  double pw_h = 0.0, num_h = 0.0;
  int bin_h = 0;
#pragma unroll
  for (int h = 0; h < NB; ++h)
    if (lane == h) { pw_h = pw[h]; num_h = num[h]; bin_h = bin[h]; }
  double numer = 0.0, denom = 0.0;
  if (lane < NB) {
    // bins above fftn/2 are an out-of-bounds read in the reference; they count as zero power here
    const double p = bin_h <= fftn / 2 ? pw_h : 0.0;
    const double inst = p == 0.0 ? 0.0 : (double)bin_h * fs * inv_fftn + num_h / p * fs_over_2pi;
    const double amp = sqrt(p);
    numer = amp * inst;
    denom = amp * (lane + 1);
  }
  numer = readlane_d(row_sum16(numer), 0);
  denom = readlane_d(row_sum16(denom), 0);
  return numer / (denom + kSafe);
}

// frames that are refined at all (stonemask.cpp:186-187), listed first (partition.hpp)
struct StoneMaskPred {
  const double* f0;
  double upper;
  double lower;        // kFloorF0StoneMask, or the caller's tighter guarantee (frames below it are not expected)
  int fs, l_above, l_upto;   // window lengths of this list: l_above < 2 hw + 1 <= l_upto (launch_stonemask)
  __device__ bool operator()(int i) const {
    const double f = f0[i];
    if ((f <= kFloorF0StoneMask || f > upper) || !(f >= lower * (1.0 - 1e-9))) return false;
    const int L = 2 * (int)(1.5 * fs / f + 1.0) + 1;            // stonemask.cpp:189
    return L > l_above && L <= l_upto;
  }
};

__global__ __launch_bounds__(64) void stonemask_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int* __restrict__ frame_utt, const double* __restrict__ tpos, const double* __restrict__ f0, int fs,
    int lmax, int64_t total_frames, const int* __restrict__ perm, const int* __restrict__ n_listed,
    const cpx* __restrict__ twid, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double sm_lds[];
  double* mw = sm_lds;                      // [lmax + 2] main window, then x * diff window
  double* xs = sm_lds + lmax + 2;           // [lmax + 2] samples, then x * main window
  const int lane0 = threadIdx.x;
  const double inv_fs = 1.0 / fs;
  const int n_run = *n_listed;
  for (int64_t i = n_run + blockIdx.x * 64 + lane0; i < total_frames; i += (int64_t)gridDim.x * 64)
    out[perm[i]] = 0.0;                                            // stonemask.cpp:186-187
  WM_FOR_EACH_LISTED(frame, perm, n_run) {
    const int lane = opaque_lane(lane0);
    const double f = f0[frame];
    const int u = frame_utt[frame];
    const double* xu = x + x_off[u];
    const int xl = x_len[u];
    const double pos = tpos[frame];
    const int hw = (int)(1.5 * fs / f + 1.0);                      // :189
    const int L = 2 * hw + 1;
    const double wlen = (2.0 * hw + 1.0) / fs;                     // :190
    const double inv_wlen = 1.0 / wlen;
    // :194-195; a power of two by shift (device pow() is not guaranteed exact for 2^n)
    const int fftn = 1 << (2 + (int)(log(hw * 2.0 + 1.0) / kLog2));
    wave_sync();
    // main window (GetMainWindow :33-43) and the samples under it (:67-68), four trips at a time so
    // that the x loads of a group are in flight together and hide behind the window arithmetic
    for (int i0 = 0; i0 < L; i0 += 256) {
      int raw[4];
      double xv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = imin(L - 1, i0 + 64 * q + lane);
        // GetBaseIndex :24-28.  The division stays a division: at rates where pos * fs is not an integer
        // (22.05 kHz: 110.25 samples per frame) the argument of the rounding lands on exact .5 ties and a
        // reciprocal multiplication flips them
        raw[q] = matlab_round((pos + (double)(-hw + i) / fs) * fs);
        xv[q] = xu[imax(0, imin(xl - 1, raw[q] - 1))];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + 64 * q + lane;
        const double tm = (raw[q] - 1.0) * inv_fs - pos;
        const double c1 = cospi(2.0 * tm * inv_wlen);               // cos(2 pi tm / wlen)
        if (i < L) {
          mw[i] = 0.42 + 0.5 * c1 + 0.08 * (2.0 * c1 * c1 - 1.0);   // + 0.08 cos(4 pi tm / wlen)
          xs[i] = xv[q];
        }
      }
    }
    wave_sync();

    int bin2[2];
    double pw2[2], num2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) bin2[h] = matlab_round(f * fftn / fs * (h + 1));   // :102
    sm_bins<2, true>(mw, xs, L, bin2, fftn, lane, twid, pw2, num2);
    const double tent = sm_fix<2>(pw2, num2, bin2, fftn, fs, lane);      // GetTentativeF0 :122-131
    double mean = 0.0;
    if (!(tent <= 0.0 || tent > f * 2)) {
      int bin6[6];
      double pw6[6], num6[6];
#pragma unroll
      for (int h = 0; h < 6; ++h) bin6[h] = matlab_round(tent * fftn / fs * (h + 1));
      wave_sync();
      sm_bins<6, false>(mw, xs, L, bin6, fftn, lane, twid, pw6, num6);
      mean = sm_fix<6>(pw6, num6, bin6, fftn, fs, lane);
    }
    if (fabs(mean - f) / f > 0.2) mean = f;                        // :202
    if (lane == 0) out[frame] = mean;
  }
}

// f0_lower: the caller's guarantee that every non-zero f0 is at least this (StoneMask itself accepts anything
// above 40 Hz, stonemask.cpp:186).  The window scratch is sized for it: behind Dio, whose candidates are
// confined to [f0_floor, f0_ceil] (dio.cpp:441-452), the windows are at most 3 periods of f0_floor instead of
// 3 periods of 40 Hz, and twice as many wavefronts fit a CU.
int launch_stonemask(Batch& b, const double* d_x, const double* d_t, const double* d_f0, double* d_out,
                     double f0_lower) {
  Context& c = *b.ctx;
  const int fs = b.p.fs;
  const double f_low = f0_lower > kFloorF0StoneMask ? f0_lower : kFloorF0StoneMask;
  // one bin of slack below the guarantee: hw is a floor of 1.5 fs / f0 + 1
  const int lmax = 2 * (int)(1.5 * fs / (f_low * (1.0 - 1e-9)) + 1.0) + 1;
  const size_t lds = sizeof(double) * 2 * (size_t)(lmax + 2);
  if (lds > 64 * 1024) return WM_ERR_UNSUPPORTED;
  const int64_t tf = b.total_f;
  if (tf <= 0) return WM_OK;
  hipFuncSetAttribute((const void*)stonemask_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (!b.d_sm_twid) {
    int rc = wm_check(dev_alloc(&b.d_sm_twid, sizeof(cpx) * (size_t)kSmTwid));
    if (rc) return rc;
    hipLaunchKernelGGL(sm_twiddle_kernel, dim3(kSmTwid / 256), dim3(256), 0, c.stream, (cpx*)b.d_sm_twid);
  }
  if (!b.d_perm2) {
    int rc = wm_check(dev_alloc(&b.d_perm2, sizeof(int) * (size_t)tf));
    if (rc) return rc;
  }
  // Two lists by window length.  The window scratch is sized by the LONGEST window a launch may meet: at 48 kHz and
  // a 71 Hz floor that is 32 KB, 1.25 waves per SIMD, while two thirds of the frames (f0 above twice the floor) need
  // half of it.  The frames with windows of at most half the maximum run with half the scratch, the rest as before.
  const int lhalf = lmax / 2;
  const double lower = f_low > kFloorF0StoneMask ? f_low : 0.0;
  // The output is cleared before the lists are made from d_f0: the two must not be the same array (the reference's
  // StoneMask takes them as separate arrays too, stonemask.h:27-29); refused rather than answered with zeros.
  if (d_out == d_f0) {
    set_error("StoneMask: refined_f0 must not alias f0");
    return WM_ERR_BAD_ARG;
  }
  int rc = wm_check(hipMemsetAsync(d_out, 0, sizeof(double) * (size_t)tf, c.stream));    // stonemask.cpp:186-187
  if (rc) return rc;
  TimedScope ts_(b.ctx, "stonemask_kernel");
  for (int cls = 0; cls < 2; ++cls) {
    const int lcap = cls == 0 ? lhalf : lmax;
    const size_t lds_c = sizeof(double) * 2 * (size_t)(lcap + 2);
    int* perm = cls == 0 ? b.d_perm : b.d_perm2;
    int* n_listed = b.d_part_n + cls;
    launch_partition(c.stream, StoneMaskPred{d_f0, fs / 12.0, lower, fs, cls == 0 ? 0 : lhalf, lcap}, (int)tf,
                     b.d_part_cnt, perm, n_listed);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stonemask_kernel, 64, lds_c) != hipSuccess || per_cu < 1)
      per_cu = 4;
    const int64_t resident = (int64_t)c.num_cu * per_cu;
    const int grid = (int)(tf < resident ? tf : resident);
    // total_frames = 0: the rows of the frames that are not refined were zeroed above, not by the kernel
    hipLaunchKernelGGL(stonemask_kernel, dim3(grid), dim3(64), lds_c, c.stream, d_x, b.d_x_off, b.d_x_len,
                       b.d_frame_utt, d_t, d_f0, fs, lcap, (int64_t)0, (const int*)perm, (const int*)n_listed,
                       (const cpx*)b.d_sm_twid, d_out);
  }
  return wm_check(hipGetLastError());
}

}  // namespace wm

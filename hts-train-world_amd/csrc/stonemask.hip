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

namespace wm {

constexpr double kFloorF0StoneMask = 40.0;   // constantnumbers.h

// Sum_i a_i e^{-j 2 pi k i / n} for NB bins k[], both windows at once.
template <int NB>
__device__ __forceinline__ void sm_bins(const double* __restrict__ xu, int xl, const double* mw, int L,
                                        double pos, int hw, int fs, const int (&bin)[NB], int fftn, int lane,
                                        double (&pw)[NB], double (&num)[NB]) {
  cpx mainv[NB], diffv[NB], w[NB], st[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    mainv[b] = make_double2(0.0, 0.0);
    diffv[b] = make_double2(0.0, 0.0);
    w[b] = cis_neg2pi((double)(((long long)bin[b] * lane) % fftn) / (double)fftn);
    st[b] = cis_neg2pi((double)(((long long)bin[b] * 64) % fftn) / (double)fftn);
  }
  for (int i = lane; i < L; i += 64) {
    const int raw = matlab_round((pos + (double)(-hw + i) / fs) * fs);   // GetBaseIndex, stonemask.cpp:24-28
    const double xi = xu[imax(0, imin(xl - 1, raw - 1))];           // :67-68
    const double m = mw[i];
    double d;                                                       // stonemask.cpp:49-55
    if (i == 0) d = -mw[1] / 2.0;
    else if (i == L - 1) d = mw[L - 2] / 2.0;
    else d = -(mw[i + 1] - mw[i - 1]) / 2.0;
    const double am = xi * m, ad = xi * d;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      mainv[b].x += am * w[b].x; mainv[b].y += am * w[b].y;
      diffv[b].x += ad * w[b].x; diffv[b].y += ad * w[b].y;
      w[b] = cmul(w[b], st[b]);
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const double mr = wave_sum(mainv[b].x), mi = wave_sum(mainv[b].y);
    const double dr = wave_sum(diffv[b].x), di = wave_sum(diffv[b].y);
    num[b] = mr * di - mi * dr;                                     // stonemask.cpp:159-160
    pw[b] = mr * mr + mi * mi;                                      // :161-162
  }
}

// FixF0, stonemask.cpp:96-117, from already-evaluated bins.
template <int NB>
__device__ __forceinline__ double sm_fix(const double (&pw)[NB], const double (&num)[NB], const int (&bin)[NB],
                                         int fftn, int fs) {
  double numer = 0.0, denom = 0.0;
#pragma unroll
  for (int h = 0; h < NB; ++h) {
    // bins above fftn/2 are an out-of-bounds read in the reference; they count as zero power here
    const bool ok = bin[h] <= fftn / 2;
    const double p = ok ? pw[h] : 0.0;
    const double inst = p == 0.0 ? 0.0 : (double)bin[h] * fs / fftn + num[h] / p * fs / 2.0 / kPi;
    const double amp = sqrt(p);
    numer += amp * inst;
    denom += amp * (h + 1);
  }
  return numer / (denom + kSafe);
}

__global__ __launch_bounds__(64) void stonemask_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int* __restrict__ frame_utt, const double* __restrict__ tpos, const double* __restrict__ f0, int fs,
    int64_t total_frames, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double mw[];
  const int lane = threadIdx.x;
  for (int64_t frame = blockIdx.x; frame < total_frames; frame += gridDim.x) {
    const double f = f0[frame];
    if (f <= kFloorF0StoneMask || f > fs / 12.0) {                 // stonemask.cpp:186-187
      if (lane == 0) out[frame] = 0.0;
      continue;
    }
    const int u = frame_utt[frame];
    const double* xu = x + x_off[u];
    const int xl = x_len[u];
    const double pos = tpos[frame];
    const int hw = (int)(1.5 * fs / f + 1.0);                      // :189
    const int L = 2 * hw + 1;
    const double wlen = (2.0 * hw + 1.0) / fs;                     // :190
    // :194-195; a power of two by shift (device pow() is not guaranteed exact for 2^n)
    const int fftn = 1 << (2 + (int)(log(hw * 2.0 + 1.0) / kLog2));
    __syncthreads();
    for (int i = lane; i < L; i += 64) {
      const double bt = (double)(-hw + i) / fs;
      const int raw = matlab_round((pos + bt) * fs);               // GetBaseIndex :24-28
      const double tm = (raw - 1.0) / fs - pos;                    // GetMainWindow :33-43
      mw[i] = 0.42 + 0.5 * cos(2.0 * kPi * tm / wlen) + 0.08 * cos(4.0 * kPi * tm / wlen);
    }
    __syncthreads();

    int bin2[2];
    double pw2[2], num2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) bin2[h] = matlab_round(f * fftn / fs * (h + 1));   // :102
    sm_bins<2>(xu, xl, mw, L, pos, hw, fs, bin2, fftn, lane, pw2, num2);
    const double tent = sm_fix<2>(pw2, num2, bin2, fftn, fs);      // GetTentativeF0 :122-131
    double mean = 0.0;
    if (!(tent <= 0.0 || tent > f * 2)) {
      int bin6[6];
      double pw6[6], num6[6];
#pragma unroll
      for (int h = 0; h < 6; ++h) bin6[h] = matlab_round(tent * fftn / fs * (h + 1));
      sm_bins<6>(xu, xl, mw, L, pos, hw, fs, bin6, fftn, lane, pw6, num6);
      mean = sm_fix<6>(pw6, num6, bin6, fftn, fs);
    }
    if (fabs(mean - f) / f > 0.2) mean = f;                        // :202
    if (lane == 0) out[frame] = mean;
  }
}

int launch_stonemask(Batch& b, const double* d_x, const double* d_t, const double* d_f0, double* d_out) {
  Context& c = *b.ctx;
  const int fs = b.p.fs;
  const int lmax = 2 * (int)(1.5 * fs / kFloorF0StoneMask + 1.0) + 1;
  const size_t lds = sizeof(double) * (size_t)(lmax + 2);
  if (lds > 64 * 1024) return WM_ERR_UNSUPPORTED;
  const int64_t tf = b.total_f;
  const int grid = (int)(tf < (int64_t)c.frame_grid ? tf : (int64_t)c.frame_grid);
  if (grid <= 0) return WM_OK;
  TimedScope ts_(b.ctx, "stonemask_kernel");
  hipLaunchKernelGGL(stonemask_kernel, dim3(grid), dim3(64), lds, c.stream, d_x, b.d_x_off, b.d_x_len,
                     b.d_frame_utt, d_t, d_f0, fs, tf, d_out);
  return wm_check(hipGetLastError());
}

}  // namespace wm

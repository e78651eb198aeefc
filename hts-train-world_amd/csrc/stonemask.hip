// stonemask.hip -- StoneMask F0 refinement, one LANE per frame.
//
// Replaces StoneMask / GetRefinedF0 / GetMeanF0 / GetTentativeF0 / FixF0
// (externs/WORLD_v2/src/stonemask.cpp:24-217).  The reference takes two real FFTs of 2^(2+int(log2(L))) points
// per frame but reads at most 8 bins of them (stonemask.cpp:101-106, :125, :130): here those bins are windowed sums
// over the L = 2 hw + 1 samples of the frame (3 periods: 60-680 samples at 16 kHz), and a frame is the job of ONE
// LANE (round 3; Harvest's refinement, harvest.hip, is the same computation and was rebuilt the same way):
//   * a workgroup takes 256 listed frames and orders them by window length in LDS (counting sort), so that the 64
//     frames of a wavefront run for about the same number of samples;
//   * a lane walks its window sample by sample: the Blackman window by rotation of (cos, sin) -- per sample, not
//     one cospi() per sample --, the differentiated window from its two neighbours, and per bin the two windowed
//     sums by Goertzel's recurrence s[n] = x[n] + 2 cos(w) s[n-1] - s[n-2] (two instructions per sum and sample);
//     the recurrence's phase factor is common to the main and the differentiated spectrum of a bin and cancels in
//     |main|^2 and Im(conj(main) diff), all FixF0 uses;
//   * GetTentativeF0's two stages (2 bins from the initial f0, then 6 from the tentative one) are two walks.
// With a wavefront per frame the twiddles, the divisions of FixF0 and the window's cospi() per sample were
// replicated or spread thin over 64 lanes and every sum ended in a cross-lane reduction: 2.8 k wave instructions
// per frame, now about 0.4 k.
// GetBaseIndex rounds (pos + base_time[i]) * fs per sample (stonemask.cpp:24-28).  Where pos * fs is within 1e-6 of
// a half-integer (22.05 kHz: 110.25 samples per frame) those roundings land on ties and differ from sample to
// sample: such frames take every index from the literal expression and every window value from cospi(); all others
// have index[i] = index[0] + i.
#include "batch.hpp"
#include "common.hpp"
#include "fastmath.hpp"
#include "fft.hpp"
#include "partition.hpp"

namespace wm {

constexpr double kFloorF0StoneMask = 40.0;   // constantnumbers.h

// exp(-2 pi i k / kSmTwid), written once per batch (k / 2^n is exact, so a lookup returns the bits the function
// would): cos / sin of a bin's angle for the recurrence
constexpr int kSmTwid = 8192;
__global__ __launch_bounds__(256) void sm_twiddle_kernel(cpx* __restrict__ tw) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < kSmTwid) tw[k] = cis_neg2pi((double)k / (double)kSmTwid);
}

// frames that are refined at all (stonemask.cpp:186-187), listed first (partition.hpp)
struct StoneMaskPred {
  const double* f0;
  double upper;
  __device__ bool operator()(int i) const {
    const double f = f0[i];
    // a NaN passes the reference's test and then sizes an array with it (undefined): not refined here
    return f == f && !(f <= kFloorF0StoneMask || f > upper);
  }
};

struct SmFrame {              // one frame's geometry (GetRefinedF0 :189-195, GetBaseIndex :24-28)
  const double* xu;
  int xl, hw, L, fftn, raw0;
  bool regular;               // index[i] = raw0 + i
  double pos, fs, inv_fs, inv_wlen;
  __device__ __forceinline__ int raw(int i) const {             // index_raw[i]
    return regular ? raw0 + i : matlab_round((pos + (double)(-hw + i) / fs) * fs);
  }
};

__device__ __forceinline__ double sm_blackman(double c) {        // 0.42 + 0.5 c + 0.08 (2 c^2 - 1), :33-43
  return fma(c, fma(0.16, c, 0.5), 0.34);
}

// power[b] = |main[b]|^2 and numer[b] = Im(conj(main[b]) diff[b]) (stonemask.cpp:159-162) at NB bins.
template <int NB>
__device__ __forceinline__ void sm_walk(const SmFrame& fr, const int (&bin)[NB], const cpx* __restrict__ twid,
                                        double (&pw)[NB], double (&num)[NB]) {
  // e^{-j w} of bin b: a table entry (transforms longer than the table compute it); fetched again after the walk
  // rather than kept through it (24 registers)
  auto bin_twiddle = [&](int b) -> cpx {
    if (fr.fftn <= kSmTwid) return twid[(bin[b] & (fr.fftn - 1)) * (kSmTwid / fr.fftn)];
    return cis_neg2pi((double)(bin[b] & (fr.fftn - 1)) / (double)fr.fftn);
  };
  double coef[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) coef[b] = 2.0 * bin_twiddle(b).x;
  double m1[NB], m2[NB], d1[NB], d2[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) m1[b] = m2[b] = d1[b] = d2[b] = 0.0;
  // window: cos(2 pi tm / wlen), tm = (index[i] - 1) / fs - pos, advanced by a rotation per sample on regular frames
  double c, sn, cd, sd;
  int r_cur = fr.raw(0);
  wm_sincospi(2.0 * ((r_cur - 1.0) * fr.inv_fs - fr.pos) * fr.inv_wlen, &sn, &c);
  wm_sincospi(2.0 * fr.inv_fs * fr.inv_wlen, &sd, &cd);
  double w_prev = 0.0, w_cur = sm_blackman(c);
  const int L = fr.L;
  // a lane leaves the loop after its own window (the wave runs on for the longest one): what a frame returns does
  // not depend on which other frames share its wavefront
  for (int i0 = 0; i0 < L; i0 += 4) {
    int rv[5];
    rv[0] = r_cur;
#pragma unroll
    for (int q = 1; q < 5; ++q) rv[q] = fr.raw(imin(i0 + q, L - 1));
    double xv[4];
    if (fr.regular && rv[0] >= 1 && rv[0] + 2 < fr.xl) {            // four consecutive samples inside the signal
      load4_a8(fr.xu + (rv[0] - 1), xv);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) xv[q] = fr.xu[imax(0, imin(fr.xl - 1, rv[q] - 1))];   // :67-68
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + q;
      double cn;
      if (fr.regular) {
        cn = c * cd - sn * sd;
        sn = sn * cd + c * sd;
      } else {
        cn = cospi(2.0 * ((rv[q + 1] - 1.0) * fr.inv_fs - fr.pos) * fr.inv_wlen);
      }
      c = cn;
      const double w_next = i + 1 < L ? sm_blackman(c) : 0.0;
      const double x = i < L ? xv[q] : 0.0;
      const double am = x * w_cur;
      const double ad = x * (0.5 * (w_prev - w_next));          // GetDiffWindow :49-55, both edges included
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const double tm = fma(coef[b], m1[b], am - m2[b]);
        m2[b] = m1[b];
        m1[b] = tm;
        const double td = fma(coef[b], d1[b], ad - d2[b]);
        d2[b] = d1[b];
        d1[b] = td;
      }
      w_prev = w_cur;
      w_cur = w_next;
    }
    r_cur = rv[4];
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const cpx t = bin_twiddle(b);
    const double cwb = t.x, swb = -t.y;
    const double mr = m1[b] - cwb * m2[b], mi = swb * m2[b];
    const double dr = d1[b] - cwb * d2[b], di = swb * d2[b];
    num[b] = mr * di - mi * dr;                                   // :159-160
    pw[b] = mr * mr + mi * mi;                                    // :161-162
  }
}

// FixF0, stonemask.cpp:96-117, from already-evaluated bins
template <int NB>
__device__ __forceinline__ double sm_fix(const double (&pw)[NB], const double (&num)[NB], const int (&bin)[NB],
                                         int fftn, int fs) {
  const double inv_fftn = 1.0 / fftn;                 // power of two: exact
  const double fs_over_2pi = fs / 2.0 / kPi;
  double numer = 0.0, denom = 0.0;
#pragma unroll
  for (int h = 0; h < NB; ++h) {
    // bins above fftn/2 are an out-of-bounds read in the reference; they count as zero power here
    const double p = bin[h] <= fftn / 2 ? pw[h] : 0.0;
    const double inst = p == 0.0 ? 0.0 : (double)bin[h] * fs * inv_fftn + num[h] / p * fs_over_2pi;
    const double amp = sqrt(p);
    numer += amp * inst;
    denom += amp * (h + 1);
  }
  return numer / (denom + kSafe);
}

constexpr int kSmChunk = 256;      // listed frames per workgroup: one group of 64 per wavefront
constexpr int kSmBins = 2048;      // half window lengths told apart by the ordering (longer ones share the last bin)

__global__ __launch_bounds__(256) void stonemask_kernel(
    const double* __restrict__ x, const int64_t* __restrict__ x_off, const int* __restrict__ x_len,
    const int* __restrict__ frame_utt, const double* __restrict__ tpos, const double* __restrict__ f0, int fs,
    const int* __restrict__ perm, const int* __restrict__ n_listed, const cpx* __restrict__ twid,
    double* __restrict__ out) {
  __shared__ int hist[kSmBins];
  __shared__ unsigned short order[kSmChunk];
  __shared__ int sh_w[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n_run = *n_listed;
  const float hw_scale = 1.5f * (float)fs;
  for (int64_t c0 = (int64_t)blockIdx.x * kSmChunk; c0 < n_run; c0 += (int64_t)gridDim.x * kSmChunk) {
    // ---- order the chunk's frames by window length, longest first ----
    for (int b = tid; b < kSmBins; b += 256) hist[b] = 0;
    __syncthreads();
    const bool mine = c0 + tid < n_run;
    int key = 0, rank = 0;
    if (mine) {
      const float fv = (float)f0[perm[c0 + tid]];
      key = kSmBins - 1 - imin(kSmBins - 1, (int)(hw_scale * __frcp_rn(fv) + 1.0f));
      rank = atomicAdd(&hist[key], 1);
    }
    __syncthreads();
    {
      int4 c4a = reinterpret_cast<int4*>(hist)[2 * tid], c4b = reinterpret_cast<int4*>(hist)[2 * tid + 1];
      const int minesum = c4a.x + c4a.y + c4a.z + c4a.w + c4b.x + c4b.y + c4b.z + c4b.w;
      const int incl = wave_scan_incl_i(minesum);
      if (lane == 63) sh_w[wv] = incl;
      __syncthreads();
      int base = incl - minesum;
      for (int q = 0; q < wv; ++q) base += sh_w[q];
      int4 oa, ob;
      oa.x = base; oa.y = oa.x + c4a.x; oa.z = oa.y + c4a.y; oa.w = oa.z + c4a.z;
      ob.x = oa.w + c4a.w; ob.y = ob.x + c4b.x; ob.z = ob.y + c4b.y; ob.w = ob.z + c4b.z;
      reinterpret_cast<int4*>(hist)[2 * tid] = oa;
      reinterpret_cast<int4*>(hist)[2 * tid + 1] = ob;
    }
    __syncthreads();
    if (mine) order[hist[key] + rank] = (unsigned short)tid;
    __syncthreads();
    const int n_here = (int)(n_run - c0 < kSmChunk ? n_run - c0 : kSmChunk);
    // ---- a lane per frame ----
    if (tid < n_here) {
      const int64_t frame = perm[c0 + order[tid]];
      const double f = f0[frame];
      const int u = frame_utt[frame];
      SmFrame fr;
      fr.xu = x + x_off[u];
      fr.xl = x_len[u];
      fr.pos = tpos[frame];
      fr.fs = fs;
      fr.inv_fs = 1.0 / fs;
      fr.hw = (int)(1.5 * fs / f + 1.0);                          // :189
      fr.L = 2 * fr.hw + 1;
      fr.inv_wlen = fs / (2.0 * fr.hw + 1.0);                     // 1 / window_length_in_time, :190
      // fft_size = 2^(2 + int(log2(L))) (:194-195): L is odd, so the logarithm is never within rounding of an integer
      fr.fftn = 1 << (2 + (31 - __clz(fr.L)));
      const double q = fr.pos * fs;
      fr.regular = fabs(q - floor(q) - 0.5) > 1e-6;
      fr.raw0 = matlab_round((fr.pos + (double)(-fr.hw) / fs) * fs);
      int bin2[2];
      double pw2[2], num2[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) bin2[h] = matlab_round(f * fr.fftn / fs * (h + 1));   // :102
      sm_walk<2>(fr, bin2, twid, pw2, num2);
      const double tent = sm_fix<2>(pw2, num2, bin2, fr.fftn, fs);    // GetTentativeF0 :122-131
      double mean = 0.0;
      if (!(tent <= 0.0 || tent > f * 2)) {
        int bin6[6];
        double pw6[6], num6[6];
#pragma unroll
        for (int h = 0; h < 6; ++h) bin6[h] = matlab_round(tent * fr.fftn / fs * (h + 1));
        sm_walk<6>(fr, bin6, twid, pw6, num6);
        mean = sm_fix<6>(pw6, num6, bin6, fr.fftn, fs);
      }
      if (fabs(mean - f) / f > 0.2) mean = f;                      // :202
      out[frame] = mean;
    }
    __syncthreads();
  }
}

// f0_lower is the caller's guarantee that every non-zero f0 is at least this (behind Dio: its floor); the kernel no
// longer sizes anything by it.
int launch_stonemask(Batch& b, const double* d_x, const double* d_t, const double* d_f0, double* d_out,
                     double f0_lower) {
  (void)f0_lower;
  Context& c = *b.ctx;
  const int fs = b.p.fs;
  const int64_t tf = b.total_f;
  if (tf <= 0) return WM_OK;
  if (!c.d_sm_twid) {                       // once per context (the table depends on nothing)
    int rc = wm_check(dev_alloc(&c.d_sm_twid, sizeof(cpx) * (size_t)kSmTwid));
    if (rc) return rc;
    hipLaunchKernelGGL(sm_twiddle_kernel, dim3(kSmTwid / 256), dim3(256), 0, c.stream, (cpx*)c.d_sm_twid);
    rc = wm_check(hipStreamSynchronize(c.stream));          // later calls may come on other streams
    if (rc) return rc;
  }
  b.d_sm_twid = c.d_sm_twid;
  // The output is cleared before the list is made from d_f0: the two must not be the same array (the reference's
  // StoneMask takes them as separate arrays too, stonemask.h:27-29); refused rather than answered with zeros.
  if (d_out == d_f0) {
    set_error("StoneMask: refined_f0 must not alias f0");
    return WM_ERR_BAD_ARG;
  }
  int rc = wm_check(hipMemsetAsync(d_out, 0, sizeof(double) * (size_t)tf, c.stream));    // stonemask.cpp:186-187
  if (rc) return rc;
  TimedScope ts_(b.ctx, "stonemask_kernel");
  launch_partition(c.stream, StoneMaskPred{d_f0, fs / 12.0}, (int)tf, b.d_part_cnt, b.d_perm, b.d_part_n);
  const int64_t chunks = (tf + kSmChunk - 1) / kSmChunk;
  const int64_t cap = (int64_t)c.num_cu * 64;
  hipLaunchKernelGGL(stonemask_kernel, dim3((unsigned)(chunks < cap ? chunks : cap)), dim3(256), 0, c.stream, d_x, b.d_x_off,
                     b.d_x_len, b.d_frame_utt, d_t, d_f0, fs, (const int*)b.d_perm, (const int*)b.d_part_n,
                     (const cpx*)b.d_sm_twid, d_out);
  return wm_check(hipGetLastError());
}

}  // namespace wm

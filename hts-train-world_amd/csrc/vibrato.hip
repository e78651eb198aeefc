// vibrato.hip -- the recipe's vibrato feature for a batch of utterances.
//
// Replaces the per-utterance run of data/scripts/Extract.py (data/Makefile.in:215; Extract.py:161-227) between the
// analysis and the `cmp` composition: from the utterance's lf0 (as WorldMi355RecipeFeatures writes it) and its
// label segments (frame range + note pitch each) to
//     vib [T][2]   log depth, log period per frame              (Extract.py:226-227, soprLog :86-96)
//     lf0'[T][2]   log f0, log(f0 - note pitch + 500) per frame (:189-199, written back over the lf0 file)
//
//   vib_scan_kernel     per utterance, lane 0: the script's sequential walk over segments and frames -- df0 /
//                       df02 and the voiced runs (f0 >= 55 Hz, closed by an unvoiced frame inside the segment,
//                       longer than 20 frames); at most T / 21 runs per utterance go into fixed slots
//   vib_lowess_kernel   one wavefront per run: LOWESS detrend (it = 20, frac 2/3, delta 0: Extract.py:220;
//                       statsmodels.nonparametric.lowess) with the run, its robustness weights and the sort
//                       scratch of the median in LDS; then getVibrate (:115-158) into the run's slice of a pair buffer
//   vib_resolve_kernel  per utterance: the caller-side copy of :223-225 in run order (the pairs of a run land
//                       AFTER the run's own frames -- the script appends to a list that already holds `length`
//                       zero pairs, :148,150 -- and a later run overwrites an earlier run's spill-over), soprLog
//
// PARITY UNPINNED: the reference holds no fixtures for this script, statsmodels is neither installed here nor
// pinned by the reference (DESIGN.md has the details).  What the tests compare this file with is a CPU restatement
// of the same reading of the script and of the published LOWESS algorithm.
// Limits: a run of more than kVibMaxRun frames (15 s of unbroken voicing at a 5 ms hop) is left without vibrato
// and counted in the status word.
#include <math.h>

#include <vector>

#include "batch.hpp"
#include "common.hpp"

namespace wm {

constexpr int kVibMaxRun = 3072;
constexpr int kVibSortMax = 4096;          // power of two >= kVibMaxRun
constexpr double kVibVoiced = 55.0;        // Extract.py:194, :207

struct VibRun {
  int ostart, n;       // n == 0: empty slot
};

__global__ __launch_bounds__(64) void vib_scan_kernel(
    const float* __restrict__ lf0, const int64_t* __restrict__ f_off, const int* __restrict__ seg_off,
    const int* __restrict__ seg_start, const int* __restrict__ seg_end, const double* __restrict__ seg_pitch,
    const int64_t* __restrict__ slot_off, double* __restrict__ f0hz, double* __restrict__ df0,
    double* __restrict__ df02, VibRun* __restrict__ runs) {
  const int u = blockIdx.x;
  const int64_t fb = f_off[u];
  const int T = (int)(f_off[u + 1] - fb);
  const int64_t sb = slot_off[u];
  const int nslot = (int)(slot_off[u + 1] - sb);
  // soprExp (Extract.py:98-108): exp, values below 1 become 0
  for (int i = threadIdx.x; i < T; i += 64) {
    const double e = exp((double)lf0[fb + i]);
    f0hz[fb + i] = e < 1.0 ? 0.0 : e;
    df0[2 * (fb + i)] = 0.0;
    df0[2 * (fb + i) + 1] = 0.0;
    df02[fb + i] = 0.0;
  }
  for (int r = threadIdx.x; r < nslot; r += 64) runs[sb + r] = VibRun{0, 0};
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double* f0 = f0hz + fb;
  int nrun = 0;
  for (int s = seg_off[u]; s < seg_off[u + 1]; ++s) {
    const int start = imax(seg_start[s], 0), end = imin(seg_end[s], T);
    const double base = seg_pitch[s];
    for (int j = start; j < end; ++j) {                         // :189-199
      double d = f0[j] - base + 500.0;
      if (d <= 0) d = -1.0;
      df0[2 * (fb + j)] = f0[j];
      const bool v = !(f0[j] < kVibVoiced);
      df0[2 * (fb + j) + 1] = v ? d : 0.0;
      df02[fb + j] = v ? f0[j] - base : 0.0;
    }
    int j = start;                                              // :202-218
    while (j < end) {
      int ostart = j, oend = 0;
      bool first = true;
      while (j < end) {
        if (first && f0[j] >= kVibVoiced) { ostart = j; first = false; }
        else if (!first && f0[j] < kVibVoiced) { oend = j; break; }
        ++j;
      }
      if (oend == 0) continue;
      if (oend - ostart > 20 && nrun < nslot) runs[sb + nrun++] = VibRun{ostart, oend - ostart};
    }
  }
}

// tricube weight of neighbour j of point i times its robustness weight
__device__ __forceinline__ double vib_weight(int j, double xi, double inv_radius, const double* rw) {
  double d = fabs((double)j - xi) * inv_radius;
  d = 1.0 - d * d * d;
  return d * d * d * rw[j];
}

__global__ __launch_bounds__(64) void vib_lowess_kernel(const int64_t* __restrict__ f_off,
                                                        const int64_t* __restrict__ slot_off,
                                                        const int* __restrict__ slot_utt, int64_t total_slots,
                                                        const VibRun* __restrict__ runs,
                                                        const double* __restrict__ df02,
                                                        double* __restrict__ pairs, int* __restrict__ pair_len,
                                                        int* __restrict__ too_long) {
  extern __shared__ __attribute__((aligned(16))) double vlds[];
  double* y = vlds;                          // [kVibMaxRun]
  double* rw = y + kVibMaxRun;               // [kVibMaxRun] robustness weights
  double* fit = rw + kVibMaxRun;             // [kVibMaxRun]
  double* srt = fit + kVibMaxRun;            // [kVibSortMax] |residual|, then sorted
  const int lane = threadIdx.x;
  for (int64_t slot = blockIdx.x; slot < total_slots; slot += gridDim.x) {
    const VibRun r = runs[slot];
    if (lane == 0) pair_len[slot] = 0;
    if (r.n == 0) continue;
    const int u = slot_utt[slot];
    const int64_t fb = f_off[u];
    const int n = r.n;
    if (n > kVibMaxRun) {
      if (lane == 0) atomicAdd(too_long, 1);
      continue;
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
      y[i] = df02[fb + r.ostart + i];
      rw[i] = 1.0;
    }
    __syncthreads();
    int k = (int)(2.0 / 3.0 * n + 1e-10);
    k = imin(imax(k, 2), n);
    int npow = 1;
    while (npow < n) npow <<= 1;
    for (int pass = 0; pass <= 20; ++pass) {
      for (int i = lane; i < n; i += 64) {
        // the neighbourhood the reference's sliding window ends on for point i (x = 0..n-1): it moves right while
        // i - lo > hi - i, i.e. lo is the smallest value >= i - k / 2 within [0, n - k]
        int lo = (2 * i - k + 1) >> 1;                      // ceil(i - k / 2)
        lo = imin(imax(lo, 0), n - k);
        const int hi = lo + k;
        const double xi = (double)i;
        const double r0 = xi - lo, r1 = (double)(hi - 1) - xi;
        const double inv_radius = 1.0 / (r0 > r1 ? r0 : r1);
        double sum_w = 0.0;
        int nonzero = 0;
        for (int j = lo; j < hi; ++j) {
          const double w = vib_weight(j, xi, inv_radius, rw);
          sum_w += w;
          nonzero += w > 1e-12 ? 1 : 0;
        }
        double f = y[i];
        if (sum_w > 0.0 && nonzero >= 2) {
          const double inv_sum = 1.0 / sum_w;
          double xm = 0.0;
          for (int j = lo; j < hi; ++j) xm += vib_weight(j, xi, inv_radius, rw) * inv_sum * (double)j;
          double sq = 0.0;
          for (int j = lo; j < hi; ++j) {
            const double dj = (double)j - xm;
            sq += vib_weight(j, xi, inv_radius, rw) * inv_sum * dj * dj;
          }
          f = 0.0;
          const double slope = (xi - xm) / sq;
          for (int j = lo; j < hi; ++j)
            f += vib_weight(j, xi, inv_radius, rw) * inv_sum * (1.0 + slope * ((double)j - xm)) * y[j];
        }
        fit[i] = f;
      }
      __syncthreads();
      if (pass == 20) break;
      // robustness weights: bisquare of |residual| / (6 median |residual|); median by a bitonic sort in LDS
      for (int i = lane; i < npow; i += 64) srt[i] = i < n ? fabs(y[i] - fit[i]) : 1.0e300;
      __syncthreads();
      for (int size = 2; size <= npow; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
          for (int t = lane; t < (npow >> 1); t += 64) {
            const int a = ((t / stride) * stride << 1) + (t % stride), b = a + stride;
            const bool up = ((a & size) == 0);
            const double va = srt[a], vb = srt[b];
            if ((va > vb) == up) { srt[a] = vb; srt[b] = va; }
          }
          __syncthreads();
        }
      }
      const double med = (n & 1) ? srt[n / 2] : 0.5 * (srt[n / 2 - 1] + srt[n / 2]);
      __syncthreads();
      for (int i = lane; i < n; i += 64) {
        const double res = fabs(y[i] - fit[i]);
        double s = med == 0.0 ? (res > 0.0 ? 1.0 : 0.0) : res / (6.0 * med);
        s = s >= 1.0 ? 1.0 : s;
        s = 1.0 - s * s;
        rw[i] = s * s;
      }
      __syncthreads();
    }
    // getVibrate (Extract.py:115-158) on the detrended run, sequentially (lane 0): pairs[2 * ostart ...] gets the
    // appended (peak, period) entries only; the `length` leading zero pairs are implied by pair_len
    if (lane == 0) {
      double* out = pairs + 4 * (fb + r.ostart);              // room for 2 n pairs = the run's own 4 n doubles
      int len = 0;
      bool positive = !(y[0] - fit[0] < 0);
      int prev = -1, last = -1;
      double peak = 0.0, period = 0.0;
      for (int i = 0; i < n; ++i) {
        const double v = y[i] - fit[i];
        bool cross = false;
        if (positive && v <= 0) { cross = true; positive = false; }
        else if (!positive && v >= 0) { cross = true; positive = true; }
        if (!cross) continue;
        if (prev >= 0) {                                       // half-wave [prev, i)
          peak = 0.0;
          for (int j = prev; j < i; ++j) peak = fmax(peak, fabs(y[j] - fit[j]));
          if (!(peak < 5)) {
            period = (double)i - (double)prev / 2.0;
            for (int j = prev; j < i; ++j) { out[2 * len] = peak; out[2 * len + 1] = period; ++len; }
          }
        }
        prev = i;
        last = i;
      }
      if (last >= 0)
        for (int i = last; i < n; ++i) { out[2 * len] = peak; out[2 * len + 1] = period; ++len; }
      pair_len[slot] = len;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ float sopr_log(double v) { return (float)(v <= 0 ? 1e-8 : log(v)); }   // Extract.py:86-96

__global__ __launch_bounds__(64) void vib_resolve_kernel(const int64_t* __restrict__ f_off,
                                                         const int64_t* __restrict__ slot_off,
                                                         const VibRun* __restrict__ runs,
                                                         const double* __restrict__ pairs,
                                                         const int* __restrict__ pair_len,
                                                         const double* __restrict__ df0, double* __restrict__ vraw,
                                                         float* __restrict__ vib, float* __restrict__ lf0_out) {
  const int u = blockIdx.x;
  const int64_t fb = f_off[u];
  const int T = (int)(f_off[u + 1] - fb);
  for (int i = threadIdx.x; i < 2 * T; i += 64) vraw[2 * fb + i] = 0.0;
  __syncthreads();
  // runs in the script's order; within a run the frames are independent, so the wave copies them together
  for (int64_t s = slot_off[u]; s < slot_off[u + 1]; ++s) {
    const VibRun r = runs[s];
    if (r.n == 0) break;
    const int len = pair_len[s];
    const double* src = pairs + 4 * (fb + r.ostart);
    for (int k = threadIdx.x; k < r.n + len; k += 64) {
      const int fr = r.ostart + k;
      if (fr >= T) continue;                                   // the script would raise here (:224)
      vraw[2 * (fb + fr)] = k < r.n ? 0.0 : src[2 * (k - r.n)];
      vraw[2 * (fb + fr) + 1] = k < r.n ? 0.0 : src[2 * (k - r.n) + 1];
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < 2 * T; i += 64) {
    vib[2 * fb + i] = sopr_log(vraw[2 * fb + i]);
    lf0_out[2 * fb + i] = sopr_log(df0[2 * fb + i]);
  }
}

struct VibWs {
  std::vector<void*> owned;
  int64_t* d_slot_off = nullptr;
  int* d_slot_utt = nullptr;
  VibRun* d_runs = nullptr;
  int* d_pair_len = nullptr;
  double *d_f0hz = nullptr, *d_df0 = nullptr, *d_df02 = nullptr, *d_pairs = nullptr, *d_vraw = nullptr;
  int* d_too_long = nullptr;
  int64_t total_slots = 0;
};

void vibrato_free(void* p) {
  VibWs* W = (VibWs*)p;
  if (!W) return;
  for (void* q : W->owned) dev_free(q);
  delete W;
}

int launch_vibrato(Batch& b, const float* d_lf0, const int* seg_utt_off, const int* seg_start, const int* seg_end,
                   const double* seg_pitch, float* d_vib, float* d_lf0_out, int* n_too_long) {
  if (!d_lf0 || !seg_utt_off || !d_vib || !d_lf0_out) return WM_ERR_BAD_ARG;
  hipStream_t st = b.ctx->stream;
  int rc = WM_OK;
  if (!b.vibrato_ws) {
    VibWs* W = new VibWs();
    std::vector<int64_t> soff((size_t)b.n_utt + 1, 0);
    for (int u = 0; u < b.n_utt; ++u) soff[(size_t)u + 1] = soff[(size_t)u] + b.f0_len[(size_t)u] / 21 + 1;
    W->total_slots = soff[(size_t)b.n_utt];
    std::vector<int> sutt((size_t)W->total_slots);
    for (int u = 0; u < b.n_utt; ++u)
      for (int64_t s = soff[(size_t)u]; s < soff[(size_t)u + 1]; ++s) sutt[(size_t)s] = u;
    auto al = [&](void** dst, size_t bytes) {
      if (rc) return;
      rc = wm_check(dev_alloc(dst, bytes ? bytes : 8));
      if (!rc) W->owned.push_back(*dst);
    };
    const size_t tf = (size_t)b.total_f;
    al((void**)&W->d_slot_off, sizeof(int64_t) * soff.size());
    al((void**)&W->d_slot_utt, sizeof(int) * sutt.size());
    al((void**)&W->d_runs, sizeof(VibRun) * (size_t)W->total_slots);
    al((void**)&W->d_pair_len, sizeof(int) * (size_t)W->total_slots);
    al((void**)&W->d_f0hz, sizeof(double) * tf);
    al((void**)&W->d_df0, sizeof(double) * 2 * tf);
    al((void**)&W->d_df02, sizeof(double) * tf);
    al((void**)&W->d_pairs, sizeof(double) * 4 * tf);
    al((void**)&W->d_vraw, sizeof(double) * 2 * tf);
    al((void**)&W->d_too_long, sizeof(int));
    if (!rc) rc = wm_check(hipMemcpy(W->d_slot_off, soff.data(), sizeof(int64_t) * soff.size(), hipMemcpyHostToDevice));
    if (!rc) rc = wm_check(hipMemcpy(W->d_slot_utt, sutt.data(), sizeof(int) * sutt.size(), hipMemcpyHostToDevice));
    if (!rc) rc = wm_check(hipFuncSetAttribute((const void*)vib_lowess_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)(sizeof(double) * (3 * kVibMaxRun + kVibSortMax))));
    if (rc) { vibrato_free(W); return rc; }
    b.vibrato_ws = W;
  }
  VibWs& W = *(VibWs*)b.vibrato_ws;
  // the label segments of this call (host arrays: they come from text files)
  const int nseg = seg_utt_off[b.n_utt];
  int *d_soff = nullptr, *d_ss = nullptr, *d_se = nullptr;
  double* d_sp = nullptr;
  auto up = [&](void** dst, const void* src, size_t bytes) {
    if (rc) return;
    rc = wm_check(dev_alloc(dst, bytes ? bytes : 8));
    if (!rc && bytes) rc = wm_check(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st));
  };
  up((void**)&d_soff, seg_utt_off, sizeof(int) * ((size_t)b.n_utt + 1));
  up((void**)&d_ss, seg_start, sizeof(int) * (size_t)nseg);
  up((void**)&d_se, seg_end, sizeof(int) * (size_t)nseg);
  up((void**)&d_sp, seg_pitch, sizeof(double) * (size_t)nseg);
  if (!rc) rc = wm_check(hipMemsetAsync(W.d_too_long, 0, sizeof(int), st));
  if (!rc) {
    TimedScope ts_(b.ctx, "vibrato_kernels");
    hipLaunchKernelGGL(vib_scan_kernel, dim3(b.n_utt), dim3(64), 0, st, d_lf0, b.d_f_off, d_soff, d_ss, d_se, d_sp,
                       W.d_slot_off, W.d_f0hz, W.d_df0, W.d_df02, W.d_runs);
    const int grid = (int)(W.total_slots < (int64_t)b.ctx->num_cu * 2 ? W.total_slots : (int64_t)b.ctx->num_cu * 2);
    hipLaunchKernelGGL(vib_lowess_kernel, dim3(grid), dim3(64), sizeof(double) * (3 * kVibMaxRun + kVibSortMax), st,
                       b.d_f_off, W.d_slot_off, W.d_slot_utt, W.total_slots, W.d_runs, W.d_df02, W.d_pairs,
                       W.d_pair_len, W.d_too_long);
    hipLaunchKernelGGL(vib_resolve_kernel, dim3(b.n_utt), dim3(64), 0, st, b.d_f_off, W.d_slot_off, W.d_runs,
                       W.d_pairs, W.d_pair_len, W.d_df0, W.d_vraw, d_vib, d_lf0_out);
    rc = wm_check(hipGetLastError());
  }
  int too_long = 0;
  if (!rc) rc = wm_check(hipMemcpyAsync(&too_long, W.d_too_long, sizeof(int), hipMemcpyDeviceToHost, st));
  if (!rc) rc = wm_check(hipStreamSynchronize(st));       // the segment arrays are call-local
  for (void* p : {(void*)d_soff, (void*)d_ss, (void*)d_se, (void*)d_sp})
    if (p) dev_free(p);
  if (n_too_long) *n_too_long = too_long;
  return rc;
}

}  // namespace wm

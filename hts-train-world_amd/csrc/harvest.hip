// harvest.hip -- Harvest F0 estimation for a batch of utterances.
//
// Replaces Harvest / HarvestGeneralBody and everything below it
// (externs/WORLD_v2/src/harvest.cpp:43-1262) plus decimate (matlabfunctions.cpp:27-125, 184-210):
//
//   hv_decim_fwd/bwd_kernel  zero-phase 3rd-order IIR decimation; each thread runs the recursion
//                            over its own chunk after a 512-sample warm-up (pole radius <= 0.89,
//                            so the warm-up state equals the sequential state to < 1e-26)
//   hv_mean_kernel           mean removal over y_length                       harvest.cpp:81-86
//   hv_band_kernel           152 cos-modulated Nuttall band-pass FIRs + four zero-crossing event
//                            lists per channel (zcfilter.hpp)                 :99-238
//   hv_raw_kernel            interp1 of the four tracks per channel, runs of frames walked  :240-293, 334-343
//   hv_detect_kernel         runs of >= 10 voiced channels -> candidates      :348-412
//   hv_refine_kernel         instantaneous-frequency refinement of every (frame, overlapped
//                            candidate): direct DFT at <= 6 bins instead of two FFTs  :417-631
//   hv_remove_kernel         RemoveUnreliableCandidates                       :652-688
//   hv_contour_kernel        SearchF0Base, FixStep1-4, SmoothF0Contour, final resampling
//                            (:693-1113, 1246-1251); sequential along time, one workgroup per
//                            utterance, sections of the zero-lag Butterworth in parallel.
//
// The band filters are evaluated as time-domain FIRs: the reference's FFT size leaves more than
// the filter length of zero padding (harvest.cpp:1164-1165), so its circular convolution is the
// linear one.
#include <math.h>

#include <stdlib.h>

#include "batch.hpp"
#include "common.hpp"
#include "fastmath.hpp"
#include "decimate.hpp"
#include "fft.hpp"
#include "fftconv.hpp"
#include "zcfilter.hpp"

namespace wm {

constexpr int kHvOverlap = 7;

struct HvMeta {
  int nch;          // number_of_channels (harvest.cpp:1151-1153)
  int r;            // decimation ratio
  double afs;       // actual_fs
  int lag;          // edge padding of GetWaveformAndSpectrumSub (:50-51)
  int cpf;          // candidates per frame before overlap = matlab_round(nch / 10.0)
  int maxc;         // max_candidates = cpf * 7 (:1180-1181)
  int ntap_max;
  int step;         // outputs per tile of the filter bank (tiles overlap by 2 samples of look-ahead)
  int conv;         // block size of the FFT convolution of the filter bank (fftconv.hpp); 0: direct FIR
  int half0;        // largest half window length: every channel is delayed to ntap0 = 2 half0 + 1, bias0 = half0 + 1
};
__host__ __device__ inline int hv_tiles(int ylen, int step) { return (ylen + step - 1) / step; }
constexpr int kHvConvC = 28;        // samples per lane of a block's event passes
constexpr int kHvChGroup = 19;      // channels per wavefront of the FFT filter bank (152 = 8 x 19)

struct HarvestWs {
  HvMeta m;
  std::vector<int> ylen, nb1;
  std::vector<int64_t> yoff, toff, evoff, boff, mdoff, smoff;
  int64_t tot_y = 0, tot_t = 0, tot_ev = 0, tot_b = 0, tot_md = 0, tot_sm = 0;
  int* d_ylen = nullptr; int* d_nb1 = nullptr;
  int64_t *d_yoff = nullptr, *d_toff = nullptr, *d_evoff = nullptr, *d_boff = nullptr, *d_mdoff = nullptr,
          *d_smoff = nullptr;
  int* d_bframe_utt = nullptr;                 // [tot_b]
  int* d_run_utt = nullptr; int* d_run_first = nullptr; int64_t n_runs = 0;   // runs of kRawRun frames (hv_raw_kernel)
  double* d_bf = nullptr; int* d_half = nullptr; int* d_tapoff = nullptr; double* d_taps = nullptr;
  double* d_y = nullptr; double* d_tmp = nullptr;
  double* d_mean_part = nullptr;               // [n_utt][kHvMeanTiles]
  int* d_evcnt = nullptr;
  int* d_tile_cnt = nullptr; int tiles_max = 0;
  void* d_H = nullptr;                         // channel spectra of the FFT filter bank
  double* d_slots = nullptr; int64_t* d_slot_off = nullptr;
  double* d_raw = nullptr; double* d_offc = nullptr; int* d_cnt = nullptr; int* d_ncand1 = nullptr;
  double *d_rc = nullptr, *d_rs = nullptr, *d_rc2 = nullptr, *d_rs2 = nullptr;
  double* d_work = nullptr;                    // [8][tot_b] contour work arrays
  int* d_bl = nullptr;                         // [2][tot_b + 8 n_utt] boundary lists
  double* d_md = nullptr;                      // banded multi-channel contours
  double* d_mds = nullptr;                     // their search scores
  int* d_sec = nullptr;                        // [3][tot_b/4 ...] section descriptors (lo, hi, off)
  double* d_sm = nullptr;                      // smoothing scratch
  cpx* d_twid = nullptr;                       // [kHvTwid] exp(-2 pi i k / kHvTwid), the refinement's twiddles
  std::vector<void*> owned;
};

__global__ __launch_bounds__(256) void hv_copy_kernel(const double* __restrict__ x,
                                                      const int64_t* __restrict__ x_off,
                                                      const int* __restrict__ x_len,
                                                      const int64_t* __restrict__ yoff, double* __restrict__ y) {
  const int u = blockIdx.y;
  const int n = x_len[u];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) y[yoff[u] + i] = x[x_off[u] + i];
}

// y -= mean(y) over y_length (harvest.cpp:81-86), in two launches of (tiles, utterances) workgroups: a workgroup per
// utterance walked its 16 ... 64 thousand samples twice with 256 threads (0.10 ms of every pass, 64 workgroups on 256
// CUs).  Every workgroup of an utterance adds the same partial sums in the same order, so the mean is one value.
constexpr int kHvMeanTiles = 32;
__global__ __launch_bounds__(256) void hv_mean_partial_kernel(const int64_t* __restrict__ yoff,
                                                              const int* __restrict__ ylen_a,
                                                              const double* __restrict__ y, double* __restrict__ part) {
  __shared__ double red[4];
  const int u = blockIdx.y;
  const double* yu = y + yoff[u];
  const int n = ylen_a[u];
  const int chunk = (n + kHvMeanTiles - 1) / kHvMeanTiles;
  const int lo = blockIdx.x * chunk, hi = imin(n, lo + chunk);
  double s = 0.0;
  for (int i = lo + threadIdx.x; i < hi; i += 256) s += yu[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[u * kHvMeanTiles + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void hv_mean_sub_kernel(const int64_t* __restrict__ yoff,
                                                          const int* __restrict__ ylen_a,
                                                          const double* __restrict__ part, double* __restrict__ y) {
  const int u = blockIdx.y;
  double* yu = y + yoff[u];
  const int n = ylen_a[u];
  double tot = 0.0;
#pragma unroll
  for (int t = 0; t < kHvMeanTiles; ++t) tot += part[u * kHvMeanTiles + t];
  const double mean = tot / n;
  const int chunk = (n + kHvMeanTiles - 1) / kHvMeanTiles;
  const int lo = blockIdx.x * chunk, hi = imin(n, lo + chunk);
  for (int i = lo + threadIdx.x; i < hi; i += 256) yu[i] -= mean;
}

// ---- filterbank + events --------------------------------------------------------------------
// One workgroup per (tile, channel, utterance); events are staged per tile, then scanned and
// compacted into the ordered lists (zcfilter.hpp).
__global__ __launch_bounds__(256) void hv_band_kernel(const int64_t* __restrict__ yoff,
                                                      const int* __restrict__ ylen_a, const double* __restrict__ y,
                                                      const double* __restrict__ taps,
                                                      const int* __restrict__ tapoff, const int* __restrict__ half,
                                                      int nch, int tiles_max, int* __restrict__ tile_cnt,
                                                      const int64_t* __restrict__ slot_off,
                                                      double* __restrict__ slots) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int u = blockIdx.z, ch = blockIdx.y, tile = blockIdx.x;
  const int ylen = ylen_a[u];
  const int nt = hv_tiles(ylen, kZcStep);
  if (tile >= nt) return;
  const int hf = half[ch];
  const int64_t slot_cap = (int64_t)nt * kZcSlot;
  // filtered[n] = sum_k bp[k] y[n + (half + 1) - k], k < 2 half + 1 (harvest.cpp:101-142)
  filter_tile_events<kZcStrideHarvest>(y + yoff[u], 0, ylen, ylen, taps + tapoff[ch], 2 * hf + 1, hf + 1, tile,
                                       tile_cnt + (((int64_t)u * nch + ch) * (tiles_max + 1) + tile) * 4,
                                       slots + slot_off[u] + (int64_t)ch * 4 * slot_cap, slot_cap, lds);
}

// The same filter bank by block FFT convolution (fftconv.hpp): one wavefront per (block, channel group,
// utterance) transforms the block once and applies the group's channels one after another.
template <int B>
__global__ __launch_bounds__(64, 2) void hv_band_fft_kernel(const int64_t* __restrict__ yoff,
                                                            const int* __restrict__ ylen_a,
                                                            const double* __restrict__ y, const cpx* __restrict__ H,
                                                            int nch, int half0, int step, int tiles_max,
                                                            int* __restrict__ tile_cnt,
                                                            const int64_t* __restrict__ slot_off,
                                                            double* __restrict__ slots) {
  constexpr int N = ConvCfg<B>::N, M = ConvCfg<B>::M;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  cpx* img = reinterpret_cast<cpx*>(lds);
  double* s = lds;
  unsigned short* lists = ConvEvCfg<B, kHvConvC>::lists(lds);
  const int u = blockIdx.z, grp = blockIdx.y, tile = blockIdx.x, lane_k = threadIdx.x, lane = lane_k;
  const int ylen = ylen_a[u];
  const int nt = hv_tiles(ylen, step);
  if (tile >= nt) return;
  FftTw<N> tw;
  tw.init(lane);
  const int ntap0 = 2 * half0 + 1, bias0 = half0 + 1;
  const int n0 = tile * step;
  const int base = n0 + bias0 - (ntap0 - 1);                     // block element i is y[base + i], zero outside
  const double* yu = y + yoff[u];
  cpx v[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int i0 = base + 2 * (lane + 64 * m), i1 = i0 + 1;
    const double a = yu[imin(ylen - 1, imax(0, i0))], c = yu[imin(ylen - 1, imax(0, i1))];
    v[m] = make_double2(i0 >= 0 && i0 < ylen ? a : 0.0, i1 >= 0 && i1 < ylen ? c : 0.0);
  }
  ConvSpec<B> zr;
  conv_forward<B>(v, img, tw, lane, zr);
  const int64_t slot_cap = (int64_t)nt * kZcSlot;
  const int ch_end = imin(nch, (grp + 1) * kHvChGroup);
#pragma unroll 1
  for (int ch = grp * kHvChGroup; ch < ch_end; ++ch) {
    const int lane = opaque_lane(lane_k);
    tw.fence();
    conv_apply<B>(zr, H + (int64_t)ch * (N + 1), img, tw, lane, v);
    wave_sync();
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int j = 2 * (lane + 64 * m) - (ntap0 - 1);
      if (j >= 0) s[j] = v[m].x;
      if (j + 1 >= 0) s[j + 1] = v[m].y;
    }
    wave_sync();
    conv_block_events<kHvConvC>(s, n0, step, ylen, tile, lists, ConvEvCfg<B, kHvConvC>::kListCap,
                                tile_cnt + (((int64_t)u * nch + ch) * (tiles_max + 1) + tile) * 4,
                                slots + slot_off[u] + (int64_t)ch * 4 * slot_cap, slot_cap, lane);
  }
}

__global__ __launch_bounds__(64) void hv_band_scan_kernel(const int* __restrict__ ylen_a, int nch, int step, int tiles_max,
                                                          int* __restrict__ tile_cnt, int* __restrict__ evcnt) {
  const int u = blockIdx.y, ch = blockIdx.x;
  const int ylen = ylen_a[u];
  zc_scan_tiles(tile_cnt + ((int64_t)u * nch + ch) * (tiles_max + 1) * 4, hv_tiles(ylen, step), ylen / 2 + 2,
                evcnt + ((int64_t)u * nch + ch) * 4, threadIdx.x);
}

// raw_f0_candidates[channel][frame] (harvest.cpp:240-293), stored [frame][channel]
// GetF0CandidateContour(+Sub) (harvest.cpp:240-293): per channel the four zero-crossing tracks are interpolated
// (interp1, matlabfunctions.cpp:136-182) at every basic frame time and averaged.  The frame times of an utterance
// are increasing and so are a track's knots (locations[j] = (e[j] + e[j+1]) / 2 / fs), so the knot interval of a
// frame is found by WALKING, not searching: a thread owns kRawRun consecutive frames of one channel, finds the
// interval of its first frame once (one binary search per track) and then steps forward -- a knot interval lasts
// 1 to 15 frames; entering the next one costs its right knot and value (two divisions; the edges it needs were
// requested two intervals earlier), a frame costs a comparison and the interpolation.  The predicate that moves to the next interval is the literal
// "locations[k] <= t" of the reference's histc, so the intervals are the reference's.  The form this replaces
// searched per frame (four binary searches and twenty divisions per frame and channel, 1 040 instructions).
// Lanes of a wavefront are consecutive channels of the same run, so the stores are rows of raw[frame][channel].
constexpr int kRawRun = 128;

// (SlotList: zcfilter.hpp -- the edge lists are read where the filter bank staged them, no compaction pass)
struct RawTrack {                 // one zero-crossing track of a channel, positioned on a knot interval
  SlotList e;                     // fine edges, n + 1 of them
  int n;                          // knots (locations) 0 .. n-1
  int lo;                         // number of knots at or before the current time (histc's count)
  int k;                          // lo clamped to [1, n-1]: the interval [knot k-1, knot k]
  double w[5];                    // e[k-1 .. k+3] (indices clamped to n): the two edges past the interval are fetched
                                  // when the interval is entered and used one and two intervals later
  double x0, x1, y0, y1;          // knots k-1, k and the values there (dio.cpp:384-387 via harvest.cpp:188-200)
  double h, dy, next;             // x1 - x0, y1 - y0; the knot that ends the stay (knot lo), +inf beyond the last
};
__device__ __forceinline__ void raw_next_knot(RawTrack& tr) {
  tr.next = tr.lo == 0 ? tr.x0 : (tr.lo < tr.n ? tr.x1 : HUGE_VAL);      // knot lo is x1 in the interior, x0 before knot 0
}
__device__ __forceinline__ void raw_init(RawTrack& tr, double fs) {
  const int k = tr.lo < 1 ? 1 : (tr.lo > tr.n - 1 ? tr.n - 1 : tr.lo);
  tr.k = k;
#pragma unroll
  for (int j = 0; j < 5; ++j) tr.w[j] = tr.e.at(imin(tr.n, k - 1 + j));
  tr.x0 = (tr.w[0] + tr.w[1]) / 2.0 / fs;
  tr.x1 = (tr.w[1] + tr.w[2]) / 2.0 / fs;
  tr.y0 = fs / (tr.w[1] - tr.w[0]);
  tr.y1 = fs / (tr.w[2] - tr.w[1]);
  tr.h = tr.x1 - tr.x0;
  tr.dy = tr.y1 - tr.y0;
  raw_next_knot(tr);
}
// one more knot lies at or before the current time
__device__ __forceinline__ void raw_advance(RawTrack& tr, double fs) {
  ++tr.lo;
  const int k = tr.lo > tr.n - 1 ? tr.n - 1 : tr.lo;           // lo >= 1 here
  if (k != tr.k) {                                              // the next interval: its left knot is the old right one
    tr.k = k;
    tr.w[0] = tr.w[1]; tr.w[1] = tr.w[2]; tr.w[2] = tr.w[3]; tr.w[3] = tr.w[4];
    tr.w[4] = tr.e.at(imin(tr.n, k + 3));
    tr.x0 = tr.x1;
    tr.y0 = tr.y1;
    tr.x1 = (tr.w[1] + tr.w[2]) / 2.0 / fs;
    tr.y1 = fs / (tr.w[2] - tr.w[1]);
    tr.h = tr.x1 - tr.x0;
    tr.dy = tr.y1 - tr.y0;
  }
  raw_next_knot(tr);
}
// quad_perm DPP: the value of lane (4 * (lane / 4) + Q) of the same quad
template <int Q>
__device__ __forceinline__ double quad_lane(double v) {
  return dpp_get<Q | (Q << 2) | (Q << 4) | (Q << 6), 0xf, 0xf>(v);
}

// A thread walks ONE track: the four tracks of a channel sit in the four lanes of a quad (forty registers of state
// per lane instead of 130, eight waves per SIMD to cover the trips to memory of a step that advances), and the
// average is taken across the quad in the reference's order of summation.
__global__ __launch_bounds__(256) void hv_raw_kernel(const int* __restrict__ run_utt, const int* __restrict__ run_first,
                                                     int64_t n_items, const int64_t* __restrict__ boff,
                                                     const int* __restrict__ nb1_a, const int* __restrict__ ylen_a,
                                                     HvMeta m, const double* __restrict__ bf, double f0_floor,
                                                     double f0_ceil, int tiles_max, const int* __restrict__ tile_off,
                                                     const int64_t* __restrict__ slot_off,
                                                     const double* __restrict__ slots, const int* __restrict__ evcnt,
                                                     double* __restrict__ raw) {
  // item = ((run, channel), track), track fastest; threads past the end shadow the last item and store nothing
  const int64_t tid_all = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool real = tid_all < n_items * 4;
  const int64_t item = (real ? tid_all : n_items * 4 - 1) >> 2;
  const int ty = threadIdx.x & 3;
  const int run = (int)(item / m.nch), ch = (int)(item - (int64_t)run * m.nch);
  const int u = run_utt[run], k0 = run_first[run];
  const int k1 = imin(nb1_a[u], k0 + kRawRun);
  const int nt = hv_tiles(ylen_a[u], m.step);
  const int c = evcnt[((int64_t)u * m.nch + ch) * 4 + ty];
  const double fs = m.afs;
  double* out = raw + (boff[u] + k0) * m.nch + ch;
  RawTrack tr;
  tr.n = c < 2 ? 0 : c - 1;
  // CheckEvent(n - 2) on all four tracks (:263-266): the quad's lanes agree on the outcome
  const bool mine_ok = tr.n > 2;
  const double okf = mine_ok ? 1.0 : 0.0;
  const bool ok = quad_lane<0>(okf) + quad_lane<1>(okf) + quad_lane<2>(okf) + quad_lane<3>(okf) == 4.0;
  const bool writer = real && ty == 0;
  if (!ok) {                                                     // quad-uniform
    if (writer)
      for (int k = k0; k < k1; ++k) out[(int64_t)(k - k0) * m.nch] = 0.0;
    return;
  }
  {
    const double t_first = k0 * 1 / 1000.0;                       // basic frame period 1 ms (:1174-1175)
    const int64_t slot_cap = (int64_t)nt * kZcSlot;
    tr.e.open(slots + slot_off[u] + ((int64_t)ch * 4 + ty) * slot_cap,
              tile_off + ((int64_t)u * m.nch + ch) * (tiles_max + 1) * 4, ty, nt, (int)(t_first * fs) / m.step - 1);
    tr.lo = slot_upper(tr.e, tr.n, fs, t_first, m.step,
                       [&](double a, double b) { return (a + b) / 2.0 / fs <= t_first; });     // zc_upper's literal form
  }
  raw_init(tr, fs);
  const double b = bf[ch];
  const double b_hi = b * 1.1, b_lo = b * 0.9;
  // Eight frames per trip, their values stored together: a store counts in vmcnt like a load and completes in issue
  // order with it, so a store per frame made every wait for a prefetched edge also a wait for the store before it
  // (2.3 us per frame; the kernel waited 69 % of its cycles).
  constexpr int kGrp = 8;
  for (int kb = k0; kb < k1; kb += kGrp) {
    double cv[kGrp];
#pragma unroll
    for (int j = 0; j < kGrp; ++j) {
      const double t = (kb + j) * 1 / 1000.0;
      while (tr.next <= t) raw_advance(tr, fs);                   // histc: knots at or before t
      const double sfrac = (t - tr.x0) / tr.h;
      const double v = tr.y0 + sfrac * tr.dy;
      const double c4 = (quad_lane<0>(v) + quad_lane<1>(v) + quad_lane<2>(v) + quad_lane<3>(v)) / 4.0;
      cv[j] = (c4 > b_hi || c4 < b_lo || c4 > f0_ceil || c4 < f0_floor) ? 0.0 : c4;   // :243-252
    }
    if (writer) {
#pragma unroll
      for (int j = 0; j < kGrp; ++j)
        if (kb + j < k1) out[(int64_t)(kb + j - k0) * m.nch] = cv[j];
    }
  }
}

// DetectOfficialF0Candidates (harvest.cpp:348-412): one thread per frame
// next set / clear bit at or after position p of a 192-bit mask (3 words), 192 if none
__device__ __forceinline__ int hv_next_bit(const unsigned long long (&w)[3], int p, bool want_set) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (p < 64 * (i + 1)) {
      const int sh = p > 64 * i ? p - 64 * i : 0;
      unsigned long long x = want_set ? w[i] : ~w[i];
      x = sh < 64 ? (x >> sh) << sh : 0ull;
      if (x) return 64 * i + __ffsll((long long)x) - 1;
    }
  }
  return 192;
}

// DetectOfficialF0Candidates (harvest.cpp:348-412): one wavefront per basic frame.  The channel row is
// read once (coalesced), voiced channels become a bit mask by ballot, runs of >= 10 voiced channels are
// walked on the mask, and each run's mean is accumulated in channel order (the reference's association).
__global__ __launch_bounds__(64) void hv_detect_kernel(const int* __restrict__ bframe_utt, HvMeta m,
                                                       const double* __restrict__ raw, int64_t tot_b,
                                                       double* __restrict__ offc, int* __restrict__ cnt,
                                                       int* __restrict__ ncand1) {
  const int lane = threadIdx.x;
  // a workgroup takes a contiguous range of frames, so that the per-utterance maximum of the candidate
  // count needs one atomic per (workgroup, utterance) instead of one per frame (339 k atomics on 64
  // addresses were the whole cost of this kernel)
  const int64_t per = (tot_b + gridDim.x - 1) / gridDim.x;
  const int64_t f_lo = blockIdx.x * per, f_hi = f_lo + per < tot_b ? f_lo + per : tot_b;
  int cur_u = -1, cur_max = 0;
  for (int64_t fr = f_lo; fr < f_hi; ++fr) {
    const double* row = raw + fr * m.nch;
    double* out = offc + fr * m.cpf;
    double vals[3];
    unsigned long long bits[3];
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      const int j = lane + 64 * w;
      vals[w] = j < m.nch ? row[j] : 0.0;
      // vuv[0] = 0 and vuv[nch - 1] = 0 by construction (:351-356)
      bits[w] = __ballot(j >= 1 && j < m.nch - 1 && vals[w] > 0);
    }
    int k = 0;
    int p = 1;
    while (p < m.nch) {
      const int st = hv_next_bit(bits, p, true);
      if (st >= m.nch) break;
      const int ed = hv_next_bit(bits, st, false);               // first unvoiced channel after the run
      if (ed - st >= 10) {
        double tmp = 0.0;
#pragma unroll
        for (int w = 0; w < 3; ++w) {                              // channel order, word by word
          const int lo = st > 64 * w ? st : 64 * w, hi = ed < 64 * w + 64 ? ed : 64 * w + 64;
          for (int q = lo; q < hi; ++q) tmp += readlane_d(vals[w], q - 64 * w);   // uniform lane: v_readlane
        }
        tmp /= (ed - st);
        if (k < m.cpf && lane == 0) out[k] = tmp;
        ++k;
      }
      p = ed;
    }
    if (k > m.cpf) k = m.cpf;
    for (int q = k + lane; q < m.cpf; q += 64) out[q] = 0.0;
    if (lane == 0) cnt[fr] = k;
    const int u = bframe_utt[fr];
    if (u != cur_u) {
      if (lane == 0 && cur_max > 0) atomicMax(&ncand1[cur_u], cur_max);
      cur_u = u;
      cur_max = 0;
    }
    cur_max = k > cur_max ? k : cur_max;
  }
  if (lane == 0 && cur_max > 0) atomicMax(&ncand1[cur_u], cur_max);
}

// exp(-2 pi i k / kHvTwid) with the function (and the argument: k / 2^n is exact) the kernels used to call
// per element, so a lookup returns the same bits
constexpr int kHvTwid = 2048;
__global__ __launch_bounds__(256) void hv_twiddle_kernel(cpx* __restrict__ tw) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < kHvTwid) tw[k] = cis_neg2pi((double)k / (double)kHvTwid);
}

// RefineF0Candidates (harvest.cpp:622-631) over the overlapped candidate table that
// OverlapF0Candidates (:417-429) would build: slot s = j + ncand1 * blk reads frame k - blk
// (blk = 1..3) or k + blk - 3 (blk = 4..6); out-of-range or unwritten entries are zero.
//
// RefineF0Candidates / GetRefinedF0 (harvest.cpp:434-617).  ONE LANE PER (frame, slot).  The windows are short
// at the decimated rate (three periods at 8 kHz: 30-340 samples) and everything around the sums is per-candidate
// scalar work (window geometry, transform size, six bins, FixF0's divisions and square roots): with a wavefront
// -- or a 16-lane row, the form this replaces -- per candidate that work was replicated over the lanes, rows
// without a candidate idled and every sum ended in a cross-lane reduction (3.9 G wave instructions on
// configs[2], 120 k per frame-wave).  Here a workgroup owns kRfChunk consecutive basic frames:
//   0. their candidate rows (and three frames either side) and per-frame descriptors go to LDS;
//   1. the occupied (frame, slot) pairs are listed, ordered by window length (counting sort on LDS counters,
//      longest first), so that the 64 refinements of a wavefront run for about the same number of samples;
//   2. waves fetch groups of 64 listed pairs; a lane walks its own window sample by sample.
// Per sample and lane: the Blackman window by rotation of (cos, sin), the differentiated window from its two
// neighbours, and for each of the six bins the two windowed sums by Goertzel's recurrence
//   s[n] = x[n] + 2 cos(w) s[n-1] - s[n-2],   X(w) = e^{-jw(N-1)} (s[N-1] - e^{-jw} s[N-2])
// (two instructions per sum and sample where a rotating twiddle takes four).  The phase factor is the same for
// the main and the differentiated spectrum of a bin, and FixF0 (:505-528) only uses |main|^2 and
// Im(conj(main) diff): both are unchanged by it, so it is dropped.  Rounding: the recurrence amplifies by about
// N / sin(w) <= 2e4 at the lowest f0, i.e. 1e-12 relative in the sums, 1e-9 Hz in a refined f0.
constexpr int kRfChunk = 128;      // basic frames per workgroup
constexpr int kRfBins = 1024;      // half window lengths told apart by the ordering; longer ones share the last bin

__device__ __forceinline__ double hv_blackman(double c) {           // 0.42 + 0.5 c + 0.08 (2 c^2 - 1), :446-456
  return fma(c, fma(0.16, c, 0.5), 0.34);
}

__global__ __launch_bounds__(256) void hv_refine_kernel(const int* __restrict__ bframe_utt,
                                                        const int64_t* __restrict__ boff,
                                                        const int* __restrict__ nb1_a, HvMeta m,
                                                        const int64_t* __restrict__ yoff,
                                                        const int* __restrict__ ylen_a, const double* __restrict__ y,
                                                        const double* __restrict__ offc,
                                                        const int* __restrict__ ncand1_a, double f0_floor,
                                                        double f0_ceil, int64_t tot_b,
                                                        const cpx* __restrict__ twid, double* __restrict__ rc,
                                                        double* __restrict__ rs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rf_smem[];
  double* rows = reinterpret_cast<double*>(rf_smem);                         // [kRfChunk + 6][cpf]
  int* hist = reinterpret_cast<int*>(rows + (kRfChunk + 6) * m.cpf);         // [kRfBins] counts, then cursors
  int4* meta = reinterpret_cast<int4*>(hist + kRfBins);                      // [kRfChunk] {utterance, k, nb1, nc1}
  unsigned short* list = reinterpret_cast<unsigned short*>(meta + kRfChunk); // [kRfChunk * maxc] (frame << 8) | slot
  __shared__ int sh_w[4], sh_next;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t fr0 = (int64_t)blockIdx.x * kRfChunk;
  const int nfr = (int)(tot_b - fr0 < kRfChunk ? tot_b - fr0 : kRfChunk);
  const double fs = m.afs, inv_fs = 1.0 / fs;

  // ---- 0. stage ----
  {
    const int64_t flat0 = (fr0 - 3) * m.cpf, flat_end = tot_b * m.cpf;
    for (int e = tid; e < (kRfChunk + 6) * m.cpf; e += 256) {
      const int64_t g = flat0 + e;
      rows[e] = (g >= 0 && g < flat_end) ? offc[g] : 0.0;
    }
    for (int fl = tid; fl < nfr; fl += 256) {
      const int u = bframe_utt[fr0 + fl];
      meta[fl] = make_int4(u, (int)(fr0 + fl - boff[u]), nb1_a[u], ncand1_a[u]);
    }
    for (int b = tid; b < kRfBins; b += 256) hist[b] = 0;
    if (tid == 0) sh_next = 0;
  }
  __syncthreads();
  // candidate of (local frame, slot): 0 when the slot is empty, -1 when it lies beyond the utterance's table
  auto candidate = [&](int fl, int slot) -> double {
    const int4 mt = meta[fl];
    if (slot >= mt.w * kHvOverlap) return -1.0;
    // slot / nc1 for slot < 256, nc1 <= 36: the quotient of the half-shifted value is at least 0.5 / nc1 from an integer
    const int blk = (int)(((float)slot + 0.5f) * __frcp_rn((float)mt.w));
    const int j = slot - blk * mt.w;
    const int d = blk == 0 ? 0 : (blk <= 3 ? -blk : blk - 3);
    const int src = mt.y + d;
    return (src >= 0 && src < mt.z) ? rows[(fl + 3 + d) * m.cpf + j] : 0.0;
  };
  // ordering key, longest window first: the half window length in single precision (the order only decides which
  // refinements share a wavefront)
  const float hw_scale = 1.5f * (float)fs;
  auto order_key = [&](double f0) -> int {
    const int hw = (int)(hw_scale * __frcp_rn((float)f0) + 1.0f);
    return kRfBins - 1 - imin(kRfBins - 1, hw);
  };
  // ---- 1. list the occupied pairs, longest window first: a wave per frame, a lane per slot ----
  for (int fl = wv; fl < nfr; fl += 4) {
    for (int slot = lane; slot < m.maxc; slot += 64) {
      const double f0 = candidate(fl, slot);
      if (f0 > 0.0) {
        atomicAdd(&hist[order_key(f0)], 1);
      } else if (f0 == 0.0) {                                  // GetRefinedF0 :593-597
        rc[(fr0 + fl) * m.maxc + slot] = 0.0;
        rs[(fr0 + fl) * m.maxc + slot] = 0.0;
      }
    }
  }
  __syncthreads();
  int n_items;
  {
    // exclusive scan of the kRfBins counters: four per thread, a wave scan, the four wave totals
    const int4 c4 = reinterpret_cast<int4*>(hist)[tid];
    const int mine = c4.x + c4.y + c4.z + c4.w;
    const int incl = wave_scan_incl_i(mine);
    if (lane == 63) sh_w[wv] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int q = 0; q < wv; ++q) base += sh_w[q];
    n_items = sh_w[0] + sh_w[1] + sh_w[2] + sh_w[3];
    reinterpret_cast<int4*>(hist)[tid] = make_int4(base, base + c4.x, base + c4.x + c4.y, base + c4.x + c4.y + c4.z);
  }
  __syncthreads();
  for (int fl = wv; fl < nfr; fl += 4) {
    for (int slot = lane; slot < m.maxc; slot += 64) {
      const double f0 = candidate(fl, slot);
      if (f0 > 0.0) list[atomicAdd(&hist[order_key(f0)], 1)] = (unsigned short)((fl << 8) | slot);
    }
  }
  __syncthreads();

  // ---- 2. refine, 64 listed pairs per wave and trip ----
  for (;;) {
    int grp = 0;
    if (lane == 0) grp = atomicAdd(&sh_next, 1);
    grp = __builtin_amdgcn_readfirstlane(grp);
    if (grp * 64 >= n_items) break;
    const int it = grp * 64 + lane;
    const bool live = it < n_items;
    const int code = list[live ? it : n_items - 1];
    const int fl = code >> 8, slot = code & 255;
    const double f0 = candidate(fl, slot);                     // > 0 by construction
    const int4 mt = meta[fl];
    const double* ys = y + yoff[mt.x];
    const int ylen = ylen_a[mt.x];
    const double pos = mt.y * 1 / 1000.0;
    // GetRefinedF0 :589-617
    const int hw = (int)(1.5 * fs / f0 + 1.0);
    const int L = 2 * hw + 1;
    const double inv_wlen = fs / (2.0 * hw + 1.0);
    // fft_size = 2^(2 + int(log2(L))): L is odd, so the logarithm is never within rounding of an integer
    const int fftn = 1 << (2 + (31 - __clz(L)));
    const double bt0 = (-hw + 0) / fs;
    const int basic = matlab_round((pos + bt0) * fs + 0.001);   // GetBaseIndex :434-441
    // GetMainWindow :446-456: cos(2 pi tm / wlen) at tm = (basic + i - 1) / fs - pos, advanced by a rotation per sample
    double c, sn, cd, sd;
    wm_sincospi(2.0 * ((basic - 1.0) * inv_fs - pos) * inv_wlen, &sn, &c);
    wm_sincospi(2.0 * inv_fs * inv_wlen, &sd, &cd);
    double coef[6], cw[6], sw[6];
    int bin[6];
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      bin[h] = matlab_round(f0 * fftn / fs * (h + 1));          // FixF0 :515
      cpx t;
      if (fftn <= kHvTwid) t = twid[(bin[h] & (fftn - 1)) * (kHvTwid / fftn)];
      else t = cis_neg2pi((double)(bin[h] & (fftn - 1)) / (double)fftn);
      cw[h] = t.x;
      sw[h] = -t.y;
      coef[h] = 2.0 * t.x;
    }
    double m1[6], m2[6], d1[6], d2[6];
#pragma unroll
    for (int h = 0; h < 6; ++h) m1[h] = m2[h] = d1[h] = d2[h] = 0.0;
    double w_prev = 0.0, w_cur = hv_blackman(c);
    // a lane leaves the loop after its own window (the wave runs on for the longest one): what a refinement returns
    // does not depend on which other refinements share its wavefront
    for (int i0 = 0; i0 < L; i0 += 4) {
      double xv[4];
      const int r0 = basic + i0 - 1;
      if (r0 >= 0 && r0 + 3 < ylen) {                               // four consecutive samples inside the signal
        load4_a8(ys + r0, xv);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[q] = ys[imax(0, imin(ylen - 1, r0 + q))];   // :481-484
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + q;
        const double nc = c * cd - sn * sd;
        sn = sn * cd + c * sd;
        c = nc;
        const double w_next = i + 1 < L ? hv_blackman(c) : 0.0;
        const double x = i < L ? xv[q] : 0.0;
        const double am = x * w_cur;
        const double ad = x * (0.5 * (w_prev - w_next));        // GetDiffWindow :462-468, both edges included
#pragma unroll
        for (int h = 0; h < 6; ++h) {
          const double tm = fma(coef[h], m1[h], am - m2[h]);
          m2[h] = m1[h];
          m1[h] = tm;
          const double td = fma(coef[h], d1[h], ad - d2[h]);
          d2[h] = d1[h];
          d1[h] = td;
        }
        w_prev = w_cur;
        w_cur = w_next;
      }
    }
    // FixF0 :505-528
    const int nh = imin((int)(fs / 2.0 / f0), 6);                // :571-572
    const double inv_fftn = 1.0 / fftn;                          // power of two: exact
    double numer = 0.0, denom = 0.0, sc = 0.0;
#pragma unroll
    for (int h = 0; h < 6; ++h) {
      const double mr = m1[h] - cw[h] * m2[h], mi = sw[h] * m2[h];
      const double dr = d1[h] - cw[h] * d2[h], di = sw[h] * d2[h];
      const double num = mr * di - mi * dr;                      // :565-566
      const double pwv = mr * mr + mi * mi;                      // :567-568
      const double p = bin[h] <= fftn / 2 ? pwv : 0.0;
      const double inst = p == 0.0 ? 0.0 : (double)bin[h] * fs * inv_fftn + num / p * fs / 2.0 / kPi;
      const double amp = sqrt(p);
      if (h < nh) {
        numer += amp * inst;
        denom += amp * (h + 1.0);
        sc += fabs((inst / (h + 1.0) - f0) / f0);
      }
    }
    double rf0 = numer / (denom + kSafe);
    double rscore = 1.0 / (sc / nh + kSafe);
    if (rf0 < f0_floor || rf0 > f0_ceil || rscore < 2.5) { rf0 = 0.0; rscore = 0.0; }   // :610-614
    if (live) {
      rc[(fr0 + fl) * m.maxc + slot] = rf0;
      rs[(fr0 + fl) * m.maxc + slot] = rscore;
    }
  }
}

// RemoveUnreliableCandidates (harvest.cpp:652-688): half a wavefront per frame, a lane per slot (the occupied slots
// are the first ncand1 * 7 of the row's maxc = 105: a thread per (frame, slot of the full row) left three threads in
// four without work); neighbours are read from the unmodified table.  Rows 0 and T-1 of the reference's scratch are
// uninitialised memory; they read as zero here.
constexpr int kRmFrames = 8;       // frames per workgroup
__global__ __launch_bounds__(256) void hv_remove_kernel(const int* __restrict__ bframe_utt,
                                                        const int64_t* __restrict__ boff,
                                                        const int* __restrict__ nb1_a, HvMeta m,
                                                        const int* __restrict__ ncand1_a,
                                                        const double* __restrict__ rc,
                                                        const double* __restrict__ rs, int64_t tot_b,
                                                        double* __restrict__ rc2, double* __restrict__ rs2) {
  // Every item compares its candidate with all candidates of the two neighbouring frames, so the rows of the
  // workgroup's frames (and one frame either side) are staged in LDS once instead of being fetched by every item.
  extern __shared__ double rows[];                              // [kRmFrames + 2][maxc]
  const int64_t fr_lo = (int64_t)blockIdx.x * kRmFrames;
  int64_t fr_hi = fr_lo + kRmFrames - 1;
  if (fr_hi > tot_b - 1) fr_hi = tot_b - 1;
  const int64_t st_lo = fr_lo > 0 ? fr_lo - 1 : 0, st_hi = fr_hi + 1 < tot_b ? fr_hi + 1 : tot_b - 1;
  const int n_stage = (int)((st_hi - st_lo + 1) * m.maxc);
  for (int i = threadIdx.x; i < n_stage; i += 256) rows[i] = rc[st_lo * m.maxc + i];
  __syncthreads();
  const int64_t fr = fr_lo + (threadIdx.x >> 5);
  if (fr >= tot_b) return;
  const int u = bframe_utt[fr];
  const int nc = ncand1_a[u] * kHvOverlap;
  const int k = (int)(fr - boff[u]);
  const int nb1 = nb1_a[u];
  for (int s = threadIdx.x & 31; s < nc; s += 32) {
    double c = rows[(fr - st_lo) * m.maxc + s], sc = rs[fr * m.maxc + s];
    if (k >= 1 && k < nb1 - 1 && c != 0) {
      // SelectBestF0 with allowed_range 1.0 on both neighbours (:652-688) only keeps the smaller relative
      // error: min over q of fl(|c - v_q| / c), capped at 1.  Correctly rounded division by c > 0 is
      // monotone, so that is fl(min |c - v_q| / c) -- one division instead of one per candidate.
      double dmin = HUGE_VAL;
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        const int nk = side == 0 ? k + 1 : k - 1;
        const bool zero_row = nk == 0 || nk == nb1 - 1;
        const double* row = rows + (fr + (side == 0 ? 1 : -1) - st_lo) * m.maxc;
        for (int q0 = 0; q0 < nc; q0 += 8) {
          double v[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = zero_row ? 0.0 : row[imin(nc - 1, q0 + r)];
#pragma unroll
          for (int r = 0; r < 8; ++r) dmin = fmin(dmin, fabs(c - v[r]));
        }
      }
      const double er = dmin / c;
      const double me = er > 1.0 ? 1.0 : er;
      if (!(me <= 0.05)) { c = 0; sc = 0; }
    }
    rc2[fr * m.maxc + s] = c;
    rs2[fr * m.maxc + s] = sc;
  }
}

// SearchF0Base (harvest.cpp:693-705) and FixStep1 (:710-722, allowed 0.008) for every basic frame of the batch:
// a workgroup takes 256 consecutive frames plus the two before them (FixStep1 looks two frames back).  Entries
// the reference leaves unwritten (f0_base == 0) are zero.  c1 = base, c2 = step 1, sm = 0 (per global frame).
__global__ __launch_bounds__(256) void hv_base_kernel(const int* __restrict__ bframe_utt, const int64_t* __restrict__ boff,
                                                      HvMeta m, const int* __restrict__ ncand1_a,
                                                      const double* __restrict__ rc2, const double* __restrict__ rs2,
                                                      int64_t tot_b, double* __restrict__ c1, double* __restrict__ c2,
                                                      double* __restrict__ bscore, double* __restrict__ sm) {
  __shared__ double bs_[258], bsc_[258];
  const int64_t fr0 = (int64_t)blockIdx.x * 256;
  for (int e = threadIdx.x; e < 258; e += 256) {
    const int64_t fr = fr0 - 2 + e;
    double bv = 0.0, bsc = 0.0;
    if (fr >= 0 && fr < tot_b) {
      const int nc = ncand1_a[bframe_utt[fr]] * kHvOverlap;
      const double* cr = rc2 + fr * m.maxc;
      const double* sr = rs2 + fr * m.maxc;
      for (int j0 = 0; j0 < nc; j0 += 8) {
        double cv[8], sv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int j = imin(nc - 1, j0 + r); cv[r] = cr[j]; sv[r] = sr[j]; }
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if (j0 + r < nc && sv[r] > bsc) { bv = cv[r]; bsc = sv[r]; }
      }
    }
    bs_[e] = bv;
    bsc_[e] = bsc;
  }
  __syncthreads();
  const int64_t fr = fr0 + threadIdx.x;
  if (fr >= tot_b) return;
  const int k = (int)(fr - boff[bframe_utt[fr]]);              // frame index inside its utterance
  const double b0 = bs_[threadIdx.x + 2], bm1 = bs_[threadIdx.x + 1], bm2 = bs_[threadIdx.x];
  double v = 0.0;
  if (k >= 2 && b0 != 0.0) {
    const double ref = bm1 * 2 - bm2;
    v = (fabs((b0 - ref) / ref) > 0.008 && fabs((b0 - bm1)) / bm1 > 0.008) ? 0.0 : b0;
  }
  c1[fr] = b0;
  c2[fr] = v;
  bscore[fr] = bsc_[threadIdx.x + 2];        // SearchScore of the base value: it is the best score of its frame
  sm[fr] = 0.0;
}

// ---- contour logic (sequential along time) -------------------------------------------------
struct HvCand {                     // candidate table of one utterance after pruning
  const double* c; const double* s; int nc; int stride;
};

// SelectBestF0 (harvest.cpp:636-650)
__device__ __forceinline__ double hv_select(double ref, const double* __restrict__ c, int n, double allowed) {
  double best = 0.0, be = allowed;
  for (int i = 0; i < n; ++i) {
    const double e = fabs(ref - c[i]) / ref;
    if (e > be) continue;
    best = c[i];
    be = e;
  }
  return best;
}

// GetBoundaryList (harvest.cpp:727-743)
__device__ int hv_boundaries(const double* f0, int n, int* list) {
  int cnt = 0, prev = 0;
  for (int i = 1; i < n; ++i) {
    const int v = (i == n - 1) ? 0 : (f0[i] > 0 ? 1 : 0);
    if (v - prev != 0) { list[cnt] = i - cnt % 2; cnt++; }
    prev = v;
  }
  return cnt;
}

struct HvSec { int lo, hi; int64_t off; };     // banded row of multi_channel_f0
__device__ __forceinline__ double hv_get(const double* md, const HvSec& s, int j) {
  return (j >= s.lo && j <= s.hi) ? md[s.off + (j - s.lo)] : 0.0;
}
__device__ __forceinline__ void hv_set(double* md, const HvSec& s, int j, double v) {
  if (j >= s.lo && j <= s.hi) md[s.off + (j - s.lo)] = v;
}

// ExtendF0 (harvest.cpp:791-820)
__device__ int hv_extend_f0(int origin, int last_point, int shift, const HvCand& cd, double allowed, double* md,
                            const HvSec& sec) {
  double tmp_f0 = hv_get(md, sec, origin);
  int shifted_origin = origin;
  const int distance = last_point > origin ? last_point - origin : origin - last_point;
  int count = 0;
  for (int i = 0; i <= distance; ++i) {
    const int idx = origin + shift * i;
    const double v = hv_select(tmp_f0, cd.c + (int64_t)(idx + shift) * cd.stride, cd.nc, allowed);
    hv_set(md, sec, idx + shift, v);
    if (v == 0.0) {
      count++;
    } else {
      tmp_f0 = v;
      count = 0;
      shifted_origin = idx + shift;
    }
    if (count == 4) break;
  }
  return shifted_origin;
}

// SearchScore (harvest.cpp:901-907)
__device__ __forceinline__ double hv_search_score(double f0, const double* c, const double* s, int n) {
  double score = 0.0;
  for (int i = 0; i < n; ++i)
    if (f0 == c[i] && score < s[i]) score = s[i];
  return score;
}

// ---- workgroup-wide (kCtThreads threads, uniform control flow) forms of the contour helpers --------------
constexpr int kCtThreads = 1024, kCtWaves = kCtThreads / 64;
// Boundaries of a 0/1 sequence (GetBoundaryList, harvest.cpp:727-743): transitions of
// v(i) = (i == n-1) ? 0 : voiced(i), i in [1, n), against v(i-1) (0 before the first), written in
// order as list[p] = i - (p & 1).  Ordered compaction by ballots; returns the count to every thread.
template <class F>
__device__ __forceinline__ int hv_boundaries_wg(F voiced, int n, int* __restrict__ list, int* shw) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int cnt = 0;
  for (int i0 = 1; i0 < n; i0 += kCtThreads) {
    const int i = i0 + threadIdx.x;
    bool flag = false;
    if (i < n) {
      const int v = (i == n - 1) ? 0 : (voiced(i) ? 1 : 0);
      const int p = (i - 1 >= 1) ? (voiced(i - 1) ? 1 : 0) : 0;
      flag = v != p;
    }
    const unsigned long long bal = __ballot(flag);
    if (lane == 0) shw[wv] = __popcll(bal);
    __syncthreads();
    int base = cnt;
    int all = 0;
    for (int q = 0; q < kCtWaves; ++q) {
      const int cq = shw[q];
      if (q < wv) base += cq;
      all += cq;
    }
    if (flag) {
      const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
      list[pos] = i - (pos & 1);
    }
    cnt += all;
    __syncthreads();
  }
  return cnt;
}

// SelectBestF0 (harvest.cpp:783-797) with the candidates of the row spread over the lanes of a wave.
// The sequential rule "skip if e > best_e, else take" keeps the LAST candidate among those of minimal
// error (<= allowed).  cv[q] / sv[q] are this lane's candidate lane + 64 q and its score (nc <= 64 * kSelPer).
// Also returns, in `score`, SearchScore (:901-907) of the value selected: the largest score among the candidates
// equal to it.  Reductions on DPP steps and ballots, no LDS crossbar (two shuffled reductions and three shuffles
// per step were 0.8 us of a step that ExtendF0 repeats up to 200 times per section).
constexpr int kSelPer = 3;
__device__ __forceinline__ double hv_select_wave(double ref, const double (&cv)[kSelPer], const double (&sv)[kSelPer],
                                                 int nc, double allowed, int lane, double& score) {
  double be = HUGE_VAL;
  double eq[kSelPer];
  bool any_nan = false;
#pragma unroll
  for (int q = 0; q < kSelPer; ++q) {
    eq[q] = HUGE_VAL;
    if (64 * q < nc) {                                     // wave-uniform
      bool is_nan = false;
      if (lane + 64 * q < nc) {
        const double e = fabs(ref - cv[q]) / ref;
        is_nan = e != e;
        if (e <= allowed) eq[q] = e;
      }
      any_nan = any_nan || __ballot(is_nan) != 0ull;
      be = fmin(be, eq[q]);
    }
  }
  const double mn = wave_min_nonneg(be);
  score = 0.0;
  int pick = -1;
  if (any_nan) {
    // a NaN error is taken by the sequential rule (`tmp > best` is false) and then so is everything after it: the
    // last candidate of the row comes out (non-finite samples only)
    pick = nc - 1;
  } else {
    if (!(mn <= allowed)) return 0.0;                      // nobody within the allowed range
#pragma unroll
    for (int q = kSelPer - 1; q >= 0; --q) {
      if (pick < 0 && 64 * q < nc) {
        const unsigned long long b = __ballot(eq[q] == mn);
        if (b) pick = 64 * q + 63 - __clzll((long long)b);
      }
    }
  }
  double out = 0.0;
#pragma unroll
  for (int q = 0; q < kSelPer; ++q)
    if ((pick >> 6) == q) out = readlane_d(cv[q], pick & 63);     // pick is wave-uniform
  double sc = 0.0;
#pragma unroll
  for (int q = 0; q < kSelPer; ++q)
    if (64 * q < nc) sc = fmax(sc, (lane + 64 * q < nc && cv[q] == out) ? sv[q] : 0.0);
  score = wave_max(sc);
  return out;
}

// ExtendF0 (harvest.cpp:791-820) on one wavefront: uniform control flow, candidates over lanes, the
// rows of the next step requested while the current one is evaluated.  The score of every value set goes to mds.
__device__ __forceinline__ int hv_extend_f0_wave(int origin, int last_point, int shift, const HvCand& cd,
                                                 double allowed, double* md, double* mds, const HvSec& sec, int lane) {
  double tmp_f0 = hv_get(md, sec, origin);
  int shifted_origin = origin;
  const int distance = last_point > origin ? last_point - origin : origin - last_point;
  int count = 0;
  int li[kSelPer];
#pragma unroll
  for (int q = 0; q < kSelPer; ++q) li[q] = lane + 64 * q < cd.nc ? lane + 64 * q : 0;
  const int64_t r0 = (int64_t)(origin + shift) * cd.stride;
  double nx[kSelPer], ns[kSelPer];
#pragma unroll
  for (int q = 0; q < kSelPer; ++q) { nx[q] = cd.c[r0 + li[q]]; ns[q] = cd.s[r0 + li[q]]; }
  for (int i = 0; i <= distance; ++i) {
    const int idx = origin + shift * i;
    double cur[kSelPer], cus[kSelPer];
#pragma unroll
    for (int q = 0; q < kSelPer; ++q) { cur[q] = nx[q]; cus[q] = ns[q]; }
    if (i < distance) {                                  // next step's rows
      const int64_t rn = (int64_t)(idx + 2 * shift) * cd.stride;
#pragma unroll
      for (int q = 0; q < kSelPer; ++q) { nx[q] = cd.c[rn + li[q]]; ns[q] = cd.s[rn + li[q]]; }
    }
    double sc;
    const double v = hv_select_wave(tmp_f0, cur, cus, cd.nc, allowed, lane, sc);
    if (lane == 0) { hv_set(md, sec, idx + shift, v); hv_set(mds, sec, idx + shift, sc); }
    if (v == 0.0) {
      count++;
    } else {
      tmp_f0 = v;
      count = 0;
      shifted_origin = idx + shift;
    }
    if (count == 4) break;
  }
  return shifted_origin;
}

// SearchScore (harvest.cpp:901-907) for one frame, one thread
__device__ __forceinline__ double hv_search_score_row(double f0, const double* __restrict__ c,
                                                      const double* __restrict__ sc, int n) {
  double score = 0.0;
  for (int i0 = 0; i0 < n; i0 += 16) {
    double cv[16], sv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int i = imin(n - 1, i0 + r); cv[r] = c[i]; sv[r] = sc[i]; }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (i0 + r < n && f0 == cv[r] && score < sv[r]) score = sv[r];
  }
  return score;
}

// Ordering of one wavefront's own LDS traffic inside a workgroup of SEVERAL wavefronts (wave_sync() is a workgroup
// barrier there): LDS operations of a wave complete in issue order, so a write is seen by the wave's later reads
// once the compiler keeps them apart and the write has left the queue.
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// x += v[0] + v[1] + ... + v[n-1] in exactly that order (adding 0.0 for the places beyond n is exact), the same on
// every lane: 64 values at a time are parked in the wave's LDS row `row` and read back by all lanes at uniform
// addresses, sixteen reads in flight ahead of the sixteen dependent additions (a shuffle per addend put an LDS
// round trip on the chain for every element).  Two sums at once share the trips: their chains interleave.
template <class F, class G>
__device__ __forceinline__ void hv_seq_sum2_wave(double& x, double& y, F value_x, G value_y, int n, int lane, double* row) {
  for (int j0 = 0; j0 < n; j0 += 64) {
    const bool in = j0 + lane < n;
    row[lane] = in ? value_x(j0 + lane) : 0.0;
    row[64 + lane] = in ? value_y(j0 + lane) : 0.0;
    wave_lds_fence();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      double a[16], b[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) { a[q] = row[16 * g + q]; b[q] = row[64 + 16 * g + q]; }
#pragma unroll
      for (int q = 0; q < 16; ++q) { x += a[q]; y += b[q]; }
    }
    wave_lds_fence();
  }
}
template <class F>
__device__ __forceinline__ double hv_seq_sum_wave(double x, F value, int n, int lane, double* row) {
  for (int j0 = 0; j0 < n; j0 += 64) {
    row[lane] = (j0 + lane < n) ? value(j0 + lane) : 0.0;
    wave_lds_fence();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      double a[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) a[q] = row[16 * g + q];
#pragma unroll
      for (int q = 0; q < 16; ++q) x += a[q];
    }
    wave_lds_fence();
  }
  return x;
}

constexpr int kSmLag = 300;      // SmoothF0Contour lag (harvest.cpp:1085)
constexpr int kSmPar = 32;       // sections filtered concurrently
constexpr int kSmTail = 320;     // frames beyond a section over which the smoothing filter settles (0.875^320 = 3e-19)

// One pass of FilteringF0's second-order recursion (harvest.cpp:1057-1062)
//     wt[i] = x[i] + a0 wt[i-1] + a1 wt[i-2],   y[i] = b0 wt[i] + b1 wt[i-1] + b0 wt[i-2]
// over n samples by ONE WAVEFRONT, 64 * kIirB samples per trip: lane l owns kIirB consecutive samples, runs them from
// a zero state, the block-end states are chained over the lanes by a scan of the affine maps s -> M s + e (M, the
// block's transition matrix, is the same for every lane, so the scan needs M^(2^d) only), and every lane then adds
// the response of its true incoming state, p[k] s1 + q[k] s2.  A thread per section walking its samples one by one
// -- every step a trip to memory for the input and one for the intermediate array -- took 1.35 ms for the longest
// utterance of configs[2], two thirds of the contour kernel.  The filter's poles have radius 0.875: the blocked form
// is as stable as the sequential one; results differ from it by rounding.
constexpr int kIirB = 16;
struct IirBlock {
  double a0, a1, b0, b1;
  double p[kIirB], q[kIirB];     // wt[k] of a block for the incoming states (1, 0) and (0, 1), zero input
  double mp[6][4];               // M^(2^d), row-major, M = [[p[B-1], q[B-1]], [p[B-2], q[B-2]]]
  double ml[4];                  // M^lane
  __device__ __forceinline__ void init(double fa0, double fa1, double fb0, double fb1, int lane) {
    a0 = fa0; a1 = fa1; b0 = fb0; b1 = fb1;
    double u1 = 1.0, u2 = 0.0, v1 = 0.0, v2 = 1.0;
#pragma unroll
    for (int k = 0; k < kIirB; ++k) {
      const double pu = a0 * u1 + a1 * u2, qv = a0 * v1 + a1 * v2;
      p[k] = pu; q[k] = qv;
      u2 = u1; u1 = pu; v2 = v1; v1 = qv;
    }
    mp[0][0] = p[kIirB - 1]; mp[0][1] = q[kIirB - 1]; mp[0][2] = p[kIirB - 2]; mp[0][3] = q[kIirB - 2];
#pragma unroll
    for (int d = 1; d < 6; ++d) {
      const double* m = mp[d - 1];
      mp[d][0] = m[0] * m[0] + m[1] * m[2]; mp[d][1] = m[0] * m[1] + m[1] * m[3];
      mp[d][2] = m[2] * m[0] + m[3] * m[2]; mp[d][3] = m[2] * m[1] + m[3] * m[3];
    }
    ml[0] = 1.0; ml[1] = 0.0; ml[2] = 0.0; ml[3] = 1.0;
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      if ((lane >> d) & 1) {
        const double* m = mp[d];
        const double r0 = m[0] * ml[0] + m[1] * ml[2], r1 = m[0] * ml[1] + m[1] * ml[3];
        const double r2 = m[2] * ml[0] + m[3] * ml[2], r3 = m[2] * ml[1] + m[3] * ml[3];
        ml[0] = r0; ml[1] = r1; ml[2] = r2; ml[3] = r3;
      }
    }
  }
  // x[k]: this lane's inputs (zero beyond the end); (s1, s2): the state entering the trip's first sample (uniform).
  // On exit y[k] are the outputs and (s1, s2) the state after the trip's last sample.
  __device__ __forceinline__ void trip(const double (&x)[kIirB], double (&y)[kIirB], double& s1, double& s2, int lane) const {
    double w[kIirB];
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int k = 0; k < kIirB; ++k) {
      const double wt = x[k] + a0 * a + a1 * b;
      w[k] = wt;
      b = a; a = wt;
    }
    double g1 = a, g2 = b;                                        // e = zero-state end of the block
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      double t1 = __shfl_up(g1, 1 << d, 64), t2 = __shfl_up(g2, 1 << d, 64);
      if (lane < (1 << d)) { t1 = 0.0; t2 = 0.0; }
      g1 += mp[d][0] * t1 + mp[d][1] * t2;
      g2 += mp[d][2] * t1 + mp[d][3] * t2;
    }
    double f1 = __shfl_up(g1, 1, 64), f2 = __shfl_up(g2, 1, 64);
    if (lane == 0) { f1 = 0.0; f2 = 0.0; }
    const double i1 = ml[0] * s1 + ml[1] * s2 + f1, i2 = ml[2] * s1 + ml[3] * s2 + f2;   // state entering this lane's block
    double m1 = i1, m2 = i2;
#pragma unroll
    for (int k = 0; k < kIirB; ++k) {
      const double wt = w[k] + p[k] * i1 + q[k] * i2;
      y[k] = b0 * wt + b1 * m1 + b0 * m2;
      m2 = m1; m1 = wt;
    }
    s1 = __shfl(m1, 63, 64);
    s2 = __shfl(m2, 63, 64);
  }
};

__global__ __launch_bounds__(kCtThreads) void hv_contour_kernel(
    const int64_t* __restrict__ boff, const int* __restrict__ nb1_a, HvMeta m, const int* __restrict__ ncand1_a,
    const double* __restrict__ rc2, const double* __restrict__ rs2, int64_t tot_b, int n_utt,
    double* __restrict__ work, int* __restrict__ blist, const int64_t* __restrict__ mdoff, double* __restrict__ mdata,
    double* __restrict__ mscore, int* __restrict__ secd, const int64_t* __restrict__ smoff, double* __restrict__ smbuf,
    const int64_t* __restrict__ f_off, double frame_period, double* __restrict__ t_out, double* __restrict__ f0_out) {
  const int u = blockIdx.x;
  const int nf = nb1_a[u];
  const int64_t b0 = boff[u];
  const int nc = ncand1_a[u] * kHvOverlap;
  HvCand cd;
  cd.c = rc2 + b0 * m.maxc; cd.s = rs2 + b0 * m.maxc; cd.nc = nc; cd.stride = m.maxc;
  double* c1 = work + 0 * tot_b + b0;
  double* c2 = work + 1 * tot_b + b0;
  double* best = work + 2 * tot_b + b0;
  double* smooth = work + 3 * tot_b + b0;
  int* bl = blist + (b0 + 8 * u);
  int* bl2 = blist + (tot_b + 8 * (int64_t)n_utt) + (b0 + 8 * u);      // up to nf + 600 boundaries? (<= nf+8 used)
  double* md = mdata + mdoff[u];
  double* mds = mscore + mdoff[u];           // SearchScore (:901-907) of every entry of md, filled once after Extend
  // section descriptors: lo[], hi[], off[] (as int offsets relative to md)
  const int sec_cap = nf / 4 + 8;
  int* s_lo = secd + 3 * (b0 / 4 + 8 * (int64_t)u);
  int* s_hi = s_lo + sec_cap;
  int* s_of = s_hi + sec_cap;
  __shared__ int sh_n;

  // SearchF0Base and FixStep1 have run for the whole batch (hv_base_kernel): c1 = base, c2 = step 1, smooth = 0
  // ---- from here on all threads run with uniform control flow; bulk loops are thread-parallel,
  // decisions are recomputed by every thread from values in memory, single writers are thread 0 ----
  __shared__ int sh4[kCtWaves];
  __shared__ double sh_row[kCtWaves][128];          // per wave: the parked addends of hv_seq_sum_wave
  __shared__ int sh_key[2048], sh_ord[2048];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

  // FixStep2 (:748-762, minimum 6): c2 -> c1
  for (int i = tid; i < nf; i += kCtThreads) c1[i] = c2[i];
  int nb = hv_boundaries_wg([&](int i) { return c2[i] > 0; }, nf, bl, sh4);
  for (int i = tid; i < nb / 2; i += kCtThreads) {
    const int lo = bl[i * 2], hi = bl[i * 2 + 1];
    if (hi - lo >= 6) continue;
    for (int j = lo; j <= hi; ++j) c1[j] = 0.0;
  }
  __syncthreads();
  // FixStep3 (:968-995, allowed 0.18): c1 -> c2
  for (int i = tid; i < nf; i += kCtThreads) c2[i] = c1[i];
  nb = hv_boundaries_wg([&](int i) { return c1[i] > 0; }, nf, bl, sh4);
  const int nsec = nb / 2;
  if (tid == 0) {                                                  // GetMultiChannelF0 :767-778 (banded rows)
    int64_t used = 0;
    for (int i = 0; i < nsec; ++i) {
      const int lo = imax(0, bl[i * 2] - 102), hi = imin(nf - 1, bl[i * 2 + 1] + 102);
      s_lo[i] = lo; s_hi[i] = hi; s_of[i] = (int)used;
      used += hi - lo + 1;
    }
  }
  __syncthreads();
  for (int i = 0; i < nsec; ++i) {
    const int lo = s_lo[i], hi = s_hi[i], of = s_of[i], v0 = bl[i * 2], v1 = bl[i * 2 + 1];
    for (int j = lo + tid; j <= hi; j += kCtThreads) {
      const double v = (j >= v0 && j <= v1) ? c1[j] : 0.0;
      md[of + (j - lo)] = v;
      mds[of + (j - lo)] = v != 0.0 ? best[j] : 0.0;       // `best` holds the base scores until FixStep4 (hv_base_kernel)
    }
  }
  __syncthreads();
  // Extend :861-878 (in place): a section only touches its own row, so sections go to the four waves
  for (int i = wv; i < nsec; i += kCtWaves) {
    HvSec sc; sc.lo = s_lo[i]; sc.hi = s_hi[i]; sc.off = s_of[i];
    const int o1 = bl[i * 2 + 1], o0 = bl[i * 2];
    const int e1 = hv_extend_f0_wave(o1, imin(nf - 2, o1 + 100), 1, cd, 0.18, md, mds, sc, lane);
    const int e0 = hv_extend_f0_wave(o0, imax(1, o0 - 100), -1, cd, 0.18, md, mds, sc, lane);
    if (lane == 0) { bl[i * 2 + 1] = e1; bl[i * 2] = e0; }
  }
  __syncthreads();
  // mds holds SearchScore (:901-907) of every entry of the rows by now (the base contour's from hv_base_kernel, the
  // extensions' from the selection itself): MergeF0Sub (:912-932) only sums them.  c1 is free from here on and carries
  // the scores of the merged contour c2.
  // ExtendSub :840-856: the running mean is NOT reset between sections (quirk), so the sections are
  // walked in order on wave 0 and every sum keeps the reference's sequential association
  if (wv == 0) {
    int nchn = 0;
    double mean_f0 = 0.0;
    for (int i = 0; i < nsec; ++i) {
      HvSec sc; sc.lo = s_lo[i]; sc.hi = s_hi[i]; sc.off = s_of[i];
      const int st = bl[i * 2], ed = bl[i * 2 + 1];
      mean_f0 = hv_seq_sum_wave(mean_f0, [&](int q) { return hv_get(md, sc, st + q); }, ed - st, lane, sh_row[wv]);
      mean_f0 /= ed - st;
      if (2200.0 / mean_f0 < ed - st) {                            // Swap :826-838
        if (lane == 0 && nchn != i) {
          int tv;
          tv = s_lo[nchn]; s_lo[nchn] = s_lo[i]; s_lo[i] = tv;
          tv = s_hi[nchn]; s_hi[nchn] = s_hi[i]; s_hi[i] = tv;
          tv = s_of[nchn]; s_of[nchn] = s_of[i]; s_of[i] = tv;
          tv = bl[nchn * 2]; bl[nchn * 2] = bl[i * 2]; bl[i * 2] = tv;
          tv = bl[nchn * 2 + 1]; bl[nchn * 2 + 1] = bl[i * 2 + 1]; bl[i * 2 + 1] = tv;
        }
        nchn++;
      }
    }
    if (lane == 0) sh_n = nchn;
  }
  __syncthreads();
  const int nchn = sh_n;
  __syncthreads();
  if (nchn != 0) {                                                 // MergeF0 :937-963
    const bool in_lds = nchn <= 2048;
    int* order = in_lds ? sh_ord : bl2;                            // MakeSortedOrder :883-896
    if (in_lds) {
      for (int i = tid; i < nchn; i += kCtThreads) sh_key[i] = bl[i * 2];
      __syncthreads();
    }
    if (tid == 0) {
      for (int i = 0; i < nchn; ++i) order[i] = i;
      for (int i = 1; i < nchn; ++i)
        for (int j = i - 1; j >= 0; --j) {
          const int kj = in_lds ? sh_key[order[j]] : bl[order[j] * 2];
          const int ki = in_lds ? sh_key[order[i]] : bl[order[i] * 2];
          if (kj > ki) { const int tv = order[i]; order[i] = order[j]; order[j] = tv; }
          else break;
        }
    }
    __syncthreads();
    {
      HvSec s0; s0.lo = s_lo[0]; s0.hi = s_hi[0]; s0.off = s_of[0];
      for (int i = tid; i < nf; i += kCtThreads) { c2[i] = hv_get(md, s0, i); c1[i] = hv_get(mds, s0, i); }
    }
    // boundary_list[0], [1] of the reference double as the running start / end of the merged contour
    int run_st = bl[0], run_ed = bl[1];
    __syncthreads();
    for (int i = 1; i < nchn; ++i) {
      const int o = order[i];
      HvSec so; so.lo = s_lo[o]; so.hi = s_hi[o]; so.off = s_of[o];
      const int st2 = o == 0 ? run_st : bl[o * 2], ed2 = o == 0 ? run_ed : bl[o * 2 + 1];
      if (st2 - run_ed > 0) {
        for (int j = st2 + tid; j <= ed2; j += kCtThreads) { c2[j] = hv_get(md, so, j); c1[j] = hv_get(mds, so, j); }
        run_st = st2;
        run_ed = ed2;
      } else {                                                     // MergeF0Sub :912-932
        const int st1 = run_st, ed1 = run_ed;
        if (!(st1 <= st2 && ed1 >= ed2)) {
          // the two sums of scores over the overlap, in frame order (every wave computes them: uniform result)
          double sc1 = 0.0, sc2 = 0.0;
          hv_seq_sum2_wave(sc1, sc2, [&](int q) { return c1[st2 + q]; }, [&](int q) { return hv_get(mds, so, st2 + q); },
                           ed1 - st2 + 1, lane, sh_row[wv]);
          __syncthreads();                                         // every wave has read c1 before it is overwritten
          const int from = sc1 > sc2 ? ed1 : st2;
          for (int k = from + tid; k <= ed2; k += kCtThreads) { c2[k] = hv_get(md, so, k); c1[k] = hv_get(mds, so, k); }
          run_ed = ed2;
        }
      }
      __syncthreads();
    }
  }
  // FixStep4 (:1000-1022, threshold 9): c2 -> best
  for (int i = tid; i < nf; i += kCtThreads) best[i] = c2[i];
  nb = hv_boundaries_wg([&](int i) { return c2[i] > 0; }, nf, bl, sh4);
  for (int i = tid; i < nb / 2 - 1; i += kCtThreads) {
    const int e0 = bl[i * 2 + 1], s1 = bl[(i + 1) * 2];
    const int distance = s1 - e0 - 1;
    if (distance >= 9) continue;
    const double tmp0 = c2[e0] + 1, tmp1 = c2[s1] - 1;
    const double coef = (tmp1 - tmp0) / (distance + 1.0);
    int count = 1;
    for (int j = e0 + 1; j <= s1 - 1; ++j) best[j] = tmp0 + coef * count++;
  }
  __syncthreads();
  // SmoothF0Contour (:1079-1113): boundaries of the zero-padded contour (best shifted by the lag)
  {
    const int nn = nf + 2 * kSmLag;
    const int cnt = hv_boundaries_wg([&](int i) { const int j = i - kSmLag; return j >= 0 && j < nf && best[j] > 0; },
                                     nn, bl, sh4);
    if (tid == 0) sh_n = cnt / 2;
  }
  __syncthreads();
  // FilteringF0 (:1049-1074) per section, a wavefront each.  The reference runs both passes of the zero-lag filter
  // over the whole padded contour; outside [st, ed] the input is constant (x[st] before, x[ed] after, :1054-1055) for
  // at least the 300 frames of padding, and the filter's poles have radius 0.875 (0.875^300 = 4e-18): at st the
  // forward pass is in its steady state for the constant x[st], and the backward pass, which only has to deliver
  // [st, ed], is in its steady state kSmTail frames beyond ed.  Both passes therefore run over [st, ed + kSmTail]
  // from those states, 1024 samples per trip of the wavefront (IirBlock).
  {
    const int nsec = sh_n;
    const int nn = nf + 2 * kSmLag;
    const double fb0 = 0.0078202080334971724, fb1 = 0.015640416066994345;
    const double fa0 = 1.7347257688092754, fa1 = -0.76600660094326412;
    const double dc = 1.0 - fa0 - fa1;                                // w = x / dc for a constant input x
    IirBlock f;
    f.init(fa0, fa1, fb0, fb1, lane);
    double* tmp = smbuf + smoff[u] + (int64_t)wv * nn;                // forward output at padded position st + q
    for (int sidx = wv; sidx < nsec; sidx += kCtWaves) {
      const int st = bl[sidx * 2], ed = bl[sidx * 2 + 1];            // padded coordinates
      const int last = imin(nn - 1, ed + kSmTail);
      const int n = last - st + 1;
      const double xe = best[ed - kSmLag];
      double s1 = best[st - kSmLag] / dc, s2 = s1;
      for (int t0 = 0; t0 < n; t0 += 64 * kIirB) {
        double x[kIirB], y[kIirB];
#pragma unroll
        for (int k = 0; k < kIirB; ++k) {
          const int i = st + t0 + kIirB * lane + k;
          const double v = best[imin(ed, i) - kSmLag];
          x[k] = i > last ? 0.0 : (i > ed ? xe : v);
        }
        f.trip(x, y, s1, s2, lane);
#pragma unroll
        for (int k = 0; k < kIirB; ++k) {
          const int q = t0 + kIirB * lane + k;
          if (q < n) tmp[q] = y[k];
        }
      }
      // the wave's own stores are visible to its own loads (same wavefront: in order through the same L1 / L2 path)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // backward: position r counts down from `last`; the state there is the steady state of the forward output
      s1 = s2 = tmp[n - 1] / dc;
      for (int t0 = 0; t0 < n; t0 += 64 * kIirB) {
        double x[kIirB], y[kIirB];
#pragma unroll
        for (int k = 0; k < kIirB; ++k) {
          const int r = t0 + kIirB * lane + k;                        // steps back from `last`
          x[k] = r < n ? tmp[imax(0, n - 1 - r)] : 0.0;
        }
        f.trip(x, y, s1, s2, lane);
#pragma unroll
        for (int k = 0; k < kIirB; ++k) {
          const int r = t0 + kIirB * lane + k;
          const int i = last - r;
          if (r < n && i <= ed) smooth[i - kSmLag] = y[k];
        }
      }
    }
  }
  __syncthreads();
  // final resampling to the requested frame period (:1230-1251)
  const int64_t fo = f_off[u];
  const int nout = (int)(f_off[u + 1] - fo);
  for (int i = threadIdx.x; i < nout; i += kCtThreads) {
    const double t = i * frame_period / 1000.0;
    t_out[fo + i] = t;
    f0_out[fo + i] = frame_period == 1.0 ? smooth[i] : smooth[imin(nf - 1, matlab_round(t * 1000.0))];
  }
}

// ---- host side ------------------------------------------------------------------------------
void harvest_free(void* p);

static int hv_setup(Batch& b) {
  if (b.harvest_ws) return WM_OK;
  // the workspace is published on the batch only once every allocation and upload has succeeded: a half-built
  // one would make the next call skip the set-up and launch with null device pointers
  HarvestWs* W = new HarvestWs();
  HvMeta& m = W->m;
  const WorldMi355Params& p = b.p;
  const double adj_floor = p.f0_floor * 0.9, adj_ceil = p.f0_ceil * 1.1;
  m.nch = 1 + (int)(log(adj_ceil / adj_floor) / kLog2 * 40);                 // :1151-1153
  m.r = imax(imin(matlab_round(p.fs / 8000.0), 12), 1);                      // :1227, :1160
  m.afs = (double)p.fs / m.r;
  m.lag = m.r == 1 ? 0 : (int)(ceil(140.0 / m.r) * m.r);
  m.cpf = matlab_round(m.nch / 10.0);
  m.maxc = m.cpf * kHvOverlap;
  if (m.maxc > 64 * kSelPer) { delete W; return WM_ERR_UNSUPPORTED; }   // hv_select_wave: candidates over lanes
  std::vector<double> bf((size_t)m.nch), taps;
  std::vector<int> half((size_t)m.nch), tapoff((size_t)m.nch);
  m.ntap_max = 0;
  for (int i = 0; i < m.nch; ++i) {
    bf[(size_t)i] = adj_floor * pow(2.0, (i + 1) / 40.0);                    // :1155-1157
    const int hf = matlab_round(m.afs / bf[(size_t)i] * 2.0);               // :101
    half[(size_t)i] = hf;
    tapoff[(size_t)i] = (int)taps.size();
    const int n = 2 * hf + 1;
    for (int k = 0; k < n; ++k) {                                            // :103-106
      const double tmp = k / (n - 1.0);
      const double w = 0.355768 - 0.487396 * cos(2.0 * kPi * tmp) + 0.144232 * cos(4.0 * kPi * tmp) -
                       0.012604 * cos(6.0 * kPi * tmp);
      taps.push_back(w * cos(2 * kPi * bf[(size_t)i] * (k - hf) / m.afs));
    }
    m.ntap_max = imax(m.ntap_max, n);
  }
  m.half0 = (m.ntap_max - 1) / 2;
  m.conv = m.ntap_max <= 1024 ? 2048 : 0;        // block FFT convolution where the longest filter fits half a block
  m.step = m.conv ? imin(m.conv - m.ntap_max + 1 - 2, 64 * kHvConvC) : kZcStep;
  const int n_utt = b.n_utt;
  W->ylen.resize((size_t)n_utt); W->nb1.resize((size_t)n_utt);
  W->yoff.assign((size_t)n_utt + 1, 0); W->toff.assign((size_t)n_utt + 1, 0); W->evoff.assign((size_t)n_utt + 1, 0);
  W->boff.assign((size_t)n_utt + 1, 0); W->mdoff.assign((size_t)n_utt + 1, 0); W->smoff.assign((size_t)n_utt + 1, 0);
  for (int u = 0; u < n_utt; ++u) {
    const int n = b.x_len[(size_t)u];
    const int yl = (int)ceil((double)n / m.r);                               // :1161-1162
    const int nb1 = (int)(1000.0 * n / p.fs / 1) + 1;                        // GetSamplesForHarvest, 1 ms
    W->ylen[(size_t)u] = yl; W->nb1[(size_t)u] = nb1;
    W->yoff[(size_t)u + 1] = W->yoff[(size_t)u] + yl;
    W->toff[(size_t)u + 1] = W->toff[(size_t)u] + n + 2 * m.lag + 18;
    W->evoff[(size_t)u + 1] = W->evoff[(size_t)u] + (int64_t)m.nch * 4 * (yl / 2 + 2);
    W->boff[(size_t)u + 1] = W->boff[(size_t)u] + nb1;
    W->mdoff[(size_t)u + 1] = W->mdoff[(size_t)u] + 28LL * nb1 + 512;
    W->smoff[(size_t)u + 1] = W->smoff[(size_t)u] + (int64_t)kSmPar * (nb1 + 2 * kSmLag);
  }
  W->tot_y = W->yoff[(size_t)n_utt]; W->tot_t = W->toff[(size_t)n_utt]; W->tot_ev = W->evoff[(size_t)n_utt];
  W->tot_b = W->boff[(size_t)n_utt]; W->tot_md = W->mdoff[(size_t)n_utt]; W->tot_sm = W->smoff[(size_t)n_utt];
  std::vector<int> bfu((size_t)W->tot_b);
  for (int u = 0; u < n_utt; ++u)
    for (int64_t i = W->boff[(size_t)u]; i < W->boff[(size_t)u + 1]; ++i) bfu[(size_t)i] = u;

  int rc = WM_OK;
  auto up = [&](void** dst, const void* src, size_t bytes) {
    if (rc) return;
    rc = wm_check(dev_alloc(dst, bytes ? bytes : 8));
    if (!rc) W->owned.push_back(*dst);
    if (!rc && bytes) rc = wm_check(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  };
  auto al = [&](void** dst, size_t bytes) {
    if (rc) return;
    rc = wm_check(dev_alloc(dst, bytes ? bytes : 8));
    if (!rc) W->owned.push_back(*dst);
  };
  up((void**)&W->d_ylen, W->ylen.data(), sizeof(int) * (size_t)n_utt);
  up((void**)&W->d_nb1, W->nb1.data(), sizeof(int) * (size_t)n_utt);
  up((void**)&W->d_yoff, W->yoff.data(), sizeof(int64_t) * ((size_t)n_utt + 1));
  up((void**)&W->d_toff, W->toff.data(), sizeof(int64_t) * ((size_t)n_utt + 1));
  up((void**)&W->d_evoff, W->evoff.data(), sizeof(int64_t) * ((size_t)n_utt + 1));
  up((void**)&W->d_boff, W->boff.data(), sizeof(int64_t) * ((size_t)n_utt + 1));
  up((void**)&W->d_mdoff, W->mdoff.data(), sizeof(int64_t) * ((size_t)n_utt + 1));
  up((void**)&W->d_smoff, W->smoff.data(), sizeof(int64_t) * ((size_t)n_utt + 1));
  up((void**)&W->d_bframe_utt, bfu.data(), sizeof(int) * bfu.size());
  {
    std::vector<int> ru, rf;
    for (int u = 0; u < n_utt; ++u)
      for (int k = 0; k < W->nb1[(size_t)u]; k += kRawRun) { ru.push_back(u); rf.push_back(k); }
    W->n_runs = (int64_t)ru.size();
    up((void**)&W->d_run_utt, ru.data(), sizeof(int) * ru.size());
    up((void**)&W->d_run_first, rf.data(), sizeof(int) * rf.size());
  }
  up((void**)&W->d_bf, bf.data(), sizeof(double) * bf.size());
  up((void**)&W->d_half, half.data(), sizeof(int) * half.size());
  up((void**)&W->d_tapoff, tapoff.data(), sizeof(int) * tapoff.size());
  up((void**)&W->d_taps, taps.data(), sizeof(double) * taps.size());
  al((void**)&W->d_y, sizeof(double) * (size_t)W->tot_y);
  al((void**)&W->d_mean_part, sizeof(double) * (size_t)n_utt * kHvMeanTiles);
  if (m.r > 1) al((void**)&W->d_tmp, sizeof(double) * (size_t)W->tot_t);
  // (W->d_events, the compacted lists, is gone: 0.58 GB at configs[2] that nobody reads any more)
  al((void**)&W->d_evcnt, sizeof(int) * (size_t)n_utt * m.nch * 4);
  {
    int ymax = 1;
    for (int u = 0; u < n_utt; ++u) ymax = imax(ymax, W->ylen[(size_t)u]);
    W->tiles_max = hv_tiles(ymax, m.step);
  }
  al((void**)&W->d_tile_cnt, sizeof(int) * (size_t)n_utt * m.nch * 4 * ((size_t)W->tiles_max + 1));
  {
    std::vector<int64_t> soff((size_t)n_utt + 1, 0);
    for (int u = 0; u < n_utt; ++u)
      soff[(size_t)u + 1] = soff[(size_t)u] + (int64_t)m.nch * 4 * hv_tiles(W->ylen[(size_t)u], m.step) * kZcSlot;
    up((void**)&W->d_slot_off, soff.data(), sizeof(int64_t) * soff.size());
    al((void**)&W->d_slots, sizeof(double) * (size_t)soff[(size_t)n_utt]);
  }
  al((void**)&W->d_raw, sizeof(double) * (size_t)W->tot_b * m.nch);
  al((void**)&W->d_offc, sizeof(double) * (size_t)W->tot_b * m.cpf);
  al((void**)&W->d_cnt, sizeof(int) * (size_t)W->tot_b);
  al((void**)&W->d_ncand1, sizeof(int) * (size_t)n_utt);
  al((void**)&W->d_rc, sizeof(double) * (size_t)W->tot_b * m.maxc);
  al((void**)&W->d_rs, sizeof(double) * (size_t)W->tot_b * m.maxc);
  al((void**)&W->d_rc2, sizeof(double) * (size_t)W->tot_b * m.maxc);
  al((void**)&W->d_rs2, sizeof(double) * (size_t)W->tot_b * m.maxc);
  al((void**)&W->d_work, sizeof(double) * 4 * (size_t)W->tot_b);
  al((void**)&W->d_bl, sizeof(int) * 2 * ((size_t)W->tot_b + 8 * (size_t)n_utt));
  al((void**)&W->d_md, sizeof(double) * (size_t)W->tot_md);
  al((void**)&W->d_mds, sizeof(double) * (size_t)W->tot_md);
  al((void**)&W->d_sec, sizeof(int) * 3 * ((size_t)W->tot_b / 4 + 8 * (size_t)n_utt + 8));
  al((void**)&W->d_sm, sizeof(double) * (size_t)W->tot_sm);
  al((void**)&W->d_twid, sizeof(cpx) * (size_t)kHvTwid);
  if (!rc) {
    hipLaunchKernelGGL(hv_twiddle_kernel, dim3(kHvTwid / 256), dim3(256), 0, b.ctx->stream, W->d_twid);
    rc = wm_check(hipGetLastError());
  }
  if (!rc && m.conv) {
    // channel spectra of the FFT filter bank: channel i delayed by half0 - half[i] samples
    std::vector<int> nt((size_t)m.nch), dl((size_t)m.nch);
    for (int i = 0; i < m.nch; ++i) {
      nt[(size_t)i] = 2 * half[(size_t)i] + 1;
      dl[(size_t)i] = m.half0 - half[(size_t)i];
    }
    int *d_nt = nullptr, *d_dl = nullptr;
    up((void**)&d_nt, nt.data(), sizeof(int) * nt.size());
    up((void**)&d_dl, dl.data(), sizeof(int) * dl.size());
    al((void**)&W->d_H, sizeof(cpx) * (size_t)m.nch * ((size_t)m.conv / 2 + 1));
    if (!rc) {
      hipLaunchKernelGGL(conv_spectrum_kernel<2048>, dim3(m.nch), dim3(64), 0, b.ctx->stream, W->d_taps, W->d_tapoff,
                         d_nt, d_dl, (cpx*)W->d_H);
      rc = wm_check(hipGetLastError());
      if (!rc) rc = wm_check(hipStreamSynchronize(b.ctx->stream));
    }
  }
  if (rc) {
    harvest_free(W);
    return rc;
  }
  b.harvest_ws = W;
  return WM_OK;
}

void harvest_free(void* p) {
  HarvestWs* W = (HarvestWs*)p;
  if (!W) return;
  for (void* q : W->owned) dev_free(q);
  delete W;
}

int launch_harvest(Batch& b, const double* d_x, double* d_t, double* d_f0) {
  if (b.total_x <= 0) return WM_ERR_BAD_ARG;
  int rc = hv_setup(b);
  if (rc) return rc;
  HarvestWs& W = *(HarvestWs*)b.harvest_ws;
  const HvMeta& m = W.m;
  Context& c = *b.ctx;
  hipStream_t st = c.stream;
  const int n_utt = b.n_utt;
  if (m.r > 1) {
    const int len_max = b.max_x_len + 2 * m.lag + 18;
    const int blocks = (len_max + 64 * kDecChunk - 1) / (64 * kDecChunk);
    TimedScope ts_(b.ctx, "hv_decimate");
    (void)hipMemsetAsync(W.d_y, 0, sizeof(double) * (size_t)W.tot_y, st);
    const DecMeta dm = make_dec_meta(m.r, m.lag);
    hipLaunchKernelGGL(decim_fwd_kernel, dim3(blocks, n_utt), dim3(64), 0, st, d_x, b.d_x_off, b.d_x_len, dm,
                       W.d_toff, W.d_tmp);
    hipLaunchKernelGGL(decim_bwd_kernel, dim3(blocks, n_utt), dim3(64), 0, st, b.d_x_len, dm, W.d_toff, W.d_tmp,
                       W.d_yoff, W.d_ylen, W.d_y);
  } else {
    hipLaunchKernelGGL(hv_copy_kernel, dim3(64, n_utt), dim3(256), 0, st, d_x, b.d_x_off, b.d_x_len, W.d_yoff, W.d_y);
  }
  hipLaunchKernelGGL(hv_mean_partial_kernel, dim3(kHvMeanTiles, n_utt), dim3(256), 0, st, W.d_yoff, W.d_ylen,
                     (const double*)W.d_y, W.d_mean_part);
  hipLaunchKernelGGL(hv_mean_sub_kernel, dim3(kHvMeanTiles, n_utt), dim3(256), 0, st, W.d_yoff, W.d_ylen,
                     (const double*)W.d_mean_part, W.d_y);
  {
    TimedScope ts_(b.ctx, "hv_band_kernel");
    if (m.conv) {
      allow_dynamic_lds(*b.ctx, hv_band_fft_kernel<2048>, (int)(ConvEvCfg<2048, kHvConvC>::kLdsBytes));
      const int groups = (m.nch + kHvChGroup - 1) / kHvChGroup;
      TimedScope tf_(b.ctx, "hv_band_fft_kernel");
      hipLaunchKernelGGL(hv_band_fft_kernel<2048>, dim3(W.tiles_max, groups, n_utt), dim3(64),
                         (ConvEvCfg<2048, kHvConvC>::kLdsBytes), st, W.d_yoff, W.d_ylen, W.d_y, (const cpx*)W.d_H, m.nch, m.half0,
                         m.step, W.tiles_max, W.d_tile_cnt, W.d_slot_off, W.d_slots);
    } else {
      const size_t lds = sizeof(double) * (size_t)zc_lds_doubles<kZcStrideHarvest>(m.ntap_max);
      if (m.ntap_max > zc_max_taps<kZcStrideHarvest>()) return WM_ERR_UNSUPPORTED;
      hipLaunchKernelGGL(hv_band_kernel, dim3(W.tiles_max, m.nch, n_utt), dim3(256), lds, st, W.d_yoff, W.d_ylen,
                         W.d_y, W.d_taps, W.d_tapoff, W.d_half, m.nch, W.tiles_max, W.d_tile_cnt, W.d_slot_off,
                         W.d_slots);
    }
    hipLaunchKernelGGL(hv_band_scan_kernel, dim3(m.nch, n_utt), dim3(64), 0, st, W.d_ylen, m.nch, m.step, W.tiles_max,
                       W.d_tile_cnt, W.d_evcnt);
    // (no compaction of the staged events into contiguous lists any more: hv_raw_kernel reads the slots, SlotList)
  }
  {
    TimedScope ts_(b.ctx, "hv_raw_kernel");
    const int64_t items = W.n_runs * m.nch;                      // four threads (the four tracks) each
    hipLaunchKernelGGL(hv_raw_kernel, dim3((unsigned)((items * 4 + 255) / 256)), dim3(256), 0, st, W.d_run_utt, W.d_run_first,
                       items, W.d_boff, W.d_nb1, W.d_ylen, m, W.d_bf, b.p.f0_floor, b.p.f0_ceil, W.tiles_max,
                       (const int*)W.d_tile_cnt, (const int64_t*)W.d_slot_off, (const double*)W.d_slots,
                       (const int*)W.d_evcnt, W.d_raw);
  }
  (void)hipMemsetAsync(W.d_ncand1, 0, sizeof(int) * (size_t)n_utt, st);
  if (m.nch > 192) return WM_ERR_UNSUPPORTED;                    // hv_detect_kernel: channel mask of 3 words
  {
    TimedScope ts_(b.ctx, "hv_detect_kernel");
    const int64_t gd = W.tot_b < (int64_t)c.num_cu * 64 ? W.tot_b : (int64_t)c.num_cu * 64;
    hipLaunchKernelGGL(hv_detect_kernel, dim3((unsigned)gd), dim3(64), 0, st, W.d_bframe_utt, m, W.d_raw, W.tot_b,
                       W.d_offc, W.d_cnt, W.d_ncand1);
  }
  {
    TimedScope ts_(b.ctx, "hv_refine_kernel");
    const size_t lds = sizeof(double) * (size_t)((kRfChunk + 6) * m.cpf) + sizeof(int) * kRfBins + sizeof(int4) * kRfChunk +
                       sizeof(unsigned short) * (size_t)(kRfChunk * m.maxc);
    if (m.maxc > 255) return WM_ERR_UNSUPPORTED;                  // a slot is eight bits of a listed pair
    allow_dynamic_lds(c, hv_refine_kernel, (int)lds);
    const int64_t grid = (W.tot_b + kRfChunk - 1) / kRfChunk;
    hipLaunchKernelGGL(hv_refine_kernel, dim3((unsigned)grid), dim3(256), lds, st, W.d_bframe_utt, W.d_boff, W.d_nb1, m,
                       W.d_yoff, W.d_ylen, W.d_y, W.d_offc, W.d_ncand1, b.p.f0_floor, b.p.f0_ceil, W.tot_b,
                       (const cpx*)W.d_twid, W.d_rc, W.d_rs);
  }
  {
    TimedScope ts_(b.ctx, "hv_remove_kernel");
    const size_t lds_rm = sizeof(double) * (size_t)((kRmFrames + 2) * m.maxc);
    hipLaunchKernelGGL(hv_remove_kernel, dim3((unsigned)((W.tot_b + kRmFrames - 1) / kRmFrames)), dim3(256), lds_rm, st,
                       W.d_bframe_utt, W.d_boff, W.d_nb1, m, W.d_ncand1, W.d_rc, W.d_rs, W.tot_b, W.d_rc2, W.d_rs2);
  }
  {
    TimedScope ts_(b.ctx, "hv_contour_kernel");
    hipLaunchKernelGGL(hv_base_kernel, dim3((unsigned)((W.tot_b + 255) / 256)), dim3(256), 0, st, W.d_bframe_utt, W.d_boff, m,
                       W.d_ncand1, W.d_rc2, W.d_rs2, W.tot_b, W.d_work, W.d_work + W.tot_b, W.d_work + 2 * W.tot_b,
                       W.d_work + 3 * W.tot_b);
    hipLaunchKernelGGL(hv_contour_kernel, dim3(n_utt), dim3(kCtThreads), 0, st, W.d_boff, W.d_nb1, m, W.d_ncand1, W.d_rc2,
                       W.d_rs2, W.tot_b, n_utt, W.d_work, W.d_bl, W.d_mdoff, W.d_md, W.d_mds, W.d_sec, W.d_smoff, W.d_sm,
                       b.d_f_off, b.p.frame_period, d_t, d_f0);
  }
  return wm_check(hipGetLastError());
}

}  // namespace wm

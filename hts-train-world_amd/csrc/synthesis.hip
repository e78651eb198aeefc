// synthesis.hip -- WORLD waveform synthesis for a batch of utterances.
//
// Replaces Synthesis and everything below it (externs/WORLD_v2/src/synthesis.cpp:19-397,
// GetMinimumPhaseSpectrum common.cpp:182-220):
//   synth_timebase_kernel  GetTimeBase :287-320 (+ :223-285): one wavefront per utterance;
//                          the phase accumulation keeps the reference's strictly sequential
//                          order (lane l adds increments 0..l one after another), so pulse
//                          positions are bit-identical; pulses are compacted in order.
//   synth_pulse_kernel     GetOneFrameSegment :183-221: one wavefront per pulse, 7 (voiced) or
//                          4 (unvoiced) real FFTs of fft_size in LDS/registers.
//   synth_ola_kernel       the overlap-add of :378-383 in gather form: every output sample sums
//                          the responses that cover it in pulse order (deterministic, same
//                          association as the reference's sequential +=).
// The reference's randn() draws for pulse i are R[idx_i - idx_0 ...) of the universal table
// (synthesis.cpp:341 reseed, :369 noise_size).
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "batch.hpp"
#include "common.hpp"
#include "fastmath.hpp"
#include "fft.hpp"
#include "partition.hpp"

namespace wm {

// coarse f0 / vuv knots of GetTemporalParametersForTimeBase (synthesis.cpp:223-240)
__device__ __forceinline__ double coarse_f0(const double* __restrict__ f0, int nf, int j, double lowest) {
  if (j < nf) {
    const double v = f0[j];
    return v < lowest ? 0.0 : v;
  }
  const int j2 = nf >= 2 ? nf - 2 : nf - 1;       // nf == 1 reads f0[-1] in the reference (undefined)
  const double a = f0[nf - 1] < lowest ? 0.0 : f0[nf - 1];
  const double b = f0[j2] < lowest ? 0.0 : f0[j2];
  return a * 2 - b;
}
__device__ __forceinline__ double coarse_vuv(const double* __restrict__ f0, int nf, int j, double lowest) {
  if (j < nf) return (f0[j] < lowest ? 0.0 : f0[j]) == 0.0 ? 0.0 : 1.0;
  const double a = (f0[nf - 1] < lowest ? 0.0 : f0[nf - 1]) == 0.0 ? 0.0 : 1.0;
  const int j2 = nf >= 2 ? nf - 2 : nf - 1;
  const double b = (f0[j2] < lowest ? 0.0 : f0[j2]) == 0.0 ? 0.0 : 1.0;
  return a * 2 - b;
}

// Part 1 of GetTimeBase (synthesis.cpp:287-307): per-sample interpolation of the coarse f0 / vuv
// contours and the phase increment 2 pi f0 / fs.  Fully parallel over samples.
// The f0-only kernels are latency chains of a few waves that run BESIDE the frame kernels of the analysis (CheapTrick,
// D4C) or of an earlier part's render stage, whose waves fill every SIMD: at equal priority a chain wave gets the
// issue slots the older waves leave.  Raised priority hands a chain its slot whenever it is ready -- it needs few.
#ifndef WM_CHAIN_PRIO_LEVEL
#define WM_CHAIN_PRIO_LEVEL 3
#endif
#define WM_CHAIN_PRIO __builtin_amdgcn_s_setprio(WM_CHAIN_PRIO_LEVEL);
// (All f0-only kernels take their utterances through a list: Synthesis prepares the batch in two parts, the
// shortest utterances first -- synthesis_prepare_part.)
__global__ __launch_bounds__(256) void synth_inc_kernel(
    const int* __restrict__ utts, const double* __restrict__ f0, const int64_t* __restrict__ f_off,
    const int64_t* __restrict__ y_off, int fs, double fp, double lowest_f0, double* __restrict__ vuv_out,
    double* __restrict__ inc_out) {
  // bit-exact increments are required (see synth_timebase_kernel): no FMA contraction
#pragma clang fp contract(off)
  WM_CHAIN_PRIO
  const int u = utts[blockIdx.y];
  const double* f0u = f0 + f_off[u];
  const int nf = (int)(f_off[u + 1] - f_off[u]);
  const int64_t yb = y_off[u];
  const int ylen = (int)(y_off[u + 1] - yb);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < ylen; i += gridDim.x * 256) {
    // interp1 (matlabfunctions.cpp:136-182) of the (nf+1)-knot coarse contours at t = i / fs
    const double t = i / (double)fs;
    int kg = (int)(t / fp) + 1;
    if (kg > nf + 1) kg = nf + 1;
    if (kg < 0) kg = 0;
    while (kg <= nf && kg * fp <= t) ++kg;
    while (kg > 0 && (kg - 1) * fp > t) --kg;
    const int k = kg < 1 ? 1 : (kg > nf ? nf : kg);
    const double x0 = (k - 1) * fp, x1 = k * fp;
    const double h = x1 - x0;
    const double s = (t - x0) / h;
    const double fa = coarse_f0(f0u, nf, k - 1, lowest_f0), fb = coarse_f0(f0u, nf, k, lowest_f0);
    const double va = coarse_vuv(f0u, nf, k - 1, lowest_f0), vb = coarse_vuv(f0u, nf, k, lowest_f0);
    double fi = fa + s * (fb - fa);
    const double vi = va + s * (vb - va);
    const double vv = vi > 0.5 ? 1.0 : 0.0;                 // synthesis.cpp:303-307
    fi = vv == 0.0 ? kDefaultF0 : fi;
    vuv_out[yb + i] = vv;
    inc_out[yb + i] = 2.0 * kPi * fi / fs;                  // :248-252
  }
}

// Part 2 (synthesis.cpp:242-285): one wavefront per utterance accumulates the phase in the
// reference's strictly sequential order, wraps it and compacts the pulses in order.
// The pulse positions of unvoiced stretches sit exactly on phase-wrap ties (500 Hz * 32 samples =
// one period), so the accumulated phase has to match the reference bit for bit: lane l adds
// increments 0..l one after another (adding 0.0 on the lanes that are done is exact), and
// fmod(total, 2 pi) is evaluated exactly as total - k * (2 pi) with a single FMA.
__global__ __launch_bounds__(64) void synth_timebase_kernel(const int* __restrict__ utts,
                                                            const int64_t* __restrict__ y_off, double* phase) {
#pragma clang fp contract(off)
  WM_CHAIN_PRIO
  const int u = utts[blockIdx.x], lane = threadIdx.x;
  const int64_t yb = y_off[u];
  const int ylen = (int)(y_off[u + 1] - yb);
  double carry = 0.0;
  // Increments travel global -> registers -> LDS -> broadcast reads.  A super-chunk of kSuper
  // samples is requested from HBM a whole super-chunk (~5 us) before it is needed, parked in
  // registers, written into one half of an LDS ring, and consumed 16 samples at a time by
  // uniform-address ds_reads that are issued one block ahead of the chain.
  constexpr int kSuper = 1024, kPer = kSuper / 64;
  __shared__ double ring[2 * kSuper];
  const double* p = phase + yb;
  double stage[kPer];
  auto request = [&](int s0) {
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int idx = s0 + 64 * q + lane;
      stage[q] = p[imin(idx, ylen - 1)];          // the tail repeats the last increment: never used
    }
  };
  auto park = [&](int half) {
#pragma unroll
    for (int q = 0; q < kPer; ++q) ring[half * kSuper + 64 * q + lane] = stage[q];
  };
  request(0);
  park(0);
  request(kSuper);
  __syncthreads();
  double r[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) r[q] = ring[q];
  for (int c0 = 0; c0 < ylen; c0 += 64) {
    const int i = c0 + lane;
    if ((c0 & (kSuper - 1)) == 0) {
      // entering super-chunk k: park k+1 (requested one super-chunk ago) and request k+2
      park(((c0 / kSuper) + 1) & 1);
      request(c0 + 2 * kSuper);
      __syncthreads();
    }
    // total_phase[i] = total_phase[i-1] + inc[i], strictly in order (:250-252): sample j is added
    // under exec = lanes >= j, so lane l ends with the prefix up to l and one sample costs one
    // v_add_f64 on the dependency chain plus one scalar shift of exec.
    double t = carry;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      double n[16];
      const int nb = (c0 + 16 * (g + 1)) & (2 * kSuper - 1);
#pragma unroll
      for (int q = 0; q < 16; ++q) n[q] = ring[nb + q];
#define WM_STEP(R) "v_add_f64 %[t], %[t], %[" #R "]\n\ts_lshl_b64 exec, exec, 1\n\t"
      asm volatile(
          "s_lshl_b64 exec, -1, %[sh]\n\t"
          WM_STEP(b0) WM_STEP(b1) WM_STEP(b2) WM_STEP(b3) WM_STEP(b4) WM_STEP(b5) WM_STEP(b6) WM_STEP(b7)
          WM_STEP(b8) WM_STEP(b9) WM_STEP(b10) WM_STEP(b11) WM_STEP(b12) WM_STEP(b13) WM_STEP(b14)
          "v_add_f64 %[t], %[t], %[b15]\n\t"
          "s_mov_b64 exec, -1"
          : [t] "+v"(t)
          : [sh] "n"(16 * g), [b0] "v"(r[0]), [b1] "v"(r[1]), [b2] "v"(r[2]), [b3] "v"(r[3]), [b4] "v"(r[4]),
            [b5] "v"(r[5]), [b6] "v"(r[6]), [b7] "v"(r[7]), [b8] "v"(r[8]), [b9] "v"(r[9]), [b10] "v"(r[10]),
            [b11] "v"(r[11]), [b12] "v"(r[12]), [b13] "v"(r[13]), [b14] "v"(r[14]), [b15] "v"(r[15])
          : "scc");
#undef WM_STEP
#pragma unroll
      for (int q = 0; q < 16; ++q) r[q] = n[q];
    }
    // the running phase replaces the increment it came from (the ring read it long ago); wrapping
    // and the pulse search are per-sample work and live in synth_pulse_search_kernel
    if (i < ylen) phase[yb + i] = t;
    carry = lane63(t);                            // v_readlane (a ds_bpermute round trip per 64 samples before)
  }
}

// wrap = fmod(total, 2 pi) (synthesis.cpp:249, :253): the remainder is exactly representable, so one
// fused multiply-add from the unrounded total gives it once k is right
__device__ __forceinline__ double wrap_two_pi(double t) {
  const double two_pi = 2.0 * kPi;
  double kq = floor(t * (1.0 / two_pi));
  double w = __fma_rn(-kq, two_pi, t);
  if (w < 0.0) { kq -= 1.0; w = __fma_rn(-kq, two_pi, t); }
  if (w >= two_pi) { kq += 1.0; w = __fma_rn(-kq, two_pi, t); }
  return w;
}

// Part 3 (synthesis.cpp:253-285): pulses where the wrapped phase jumps by more than pi, compacted in order.
// Every tile of kSearchTile samples is its own workgroup (an utterance of 8 s at 48 kHz has 188 of them; walking
// them in order on one workgroup was 8 ms of latency chain): WRITE = false counts the tile's pulses,
// WRITE = true adds up the counts of the tiles before it and writes the tile's pulses at that offset.
constexpr int kSearchSub = 8, kSearchTile = 256 * kSearchSub;
template <bool WRITE>
__global__ __launch_bounds__(256) void synth_pulse_search_kernel(
    const int* __restrict__ utts, const int64_t* __restrict__ y_off, const double* __restrict__ phase, int fs,
    int tiles_max, int* __restrict__ tile_cnt, int* __restrict__ pulse_idx, double* __restrict__ pulse_shift,
    int* __restrict__ pulse_cnt) {
#pragma clang fp contract(off)
  WM_CHAIN_PRIO
  __shared__ int wave_cnt[kSearchSub][4];
  __shared__ int red[4];
  const int u = utts[blockIdx.y], tile = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t yb = y_off[u];
  const int ylen = (int)(y_off[u + 1] - yb);
  const int i0 = tile * kSearchTile;
  if (i0 >= ylen && !(WRITE && tile == 0)) return;
  int* cnt_u = tile_cnt + (int64_t)u * tiles_max;
  int count = 0;
  if (WRITE) {
    // pulses of the tiles before this one (and, on tile 0, of the whole utterance)
    const int ntile = (ylen + kSearchTile - 1) / kSearchTile;
    int before = 0, all = 0;
    for (int j = threadIdx.x; j < ntile; j += 256) {
      const int c = cnt_u[j];
      all += c;
      if (j < tile) before += c;
    }
    before = wave_sum_i(before);
    all = wave_sum_i(all);
    if (lane == 0) red[wv] = before;
    __syncthreads();
    count = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    if (tile == 0) {
      if (lane == 0) red[wv] = all;
      __syncthreads();
      if (threadIdx.x == 0) pulse_cnt[u] = red[0] + red[1] + red[2] + red[3];
      __syncthreads();
    }
    if (i0 >= ylen) return;
  }
  // all loads of the tile first: one memory round trip
  double tc[kSearchSub], tp[kSearchSub];
#pragma unroll
  for (int q = 0; q < kSearchSub; ++q) {
    const int i = i0 + 256 * q + threadIdx.x;
    const bool in = i >= 1 && i < ylen;
    tc[q] = phase[yb + (in ? i : 0)];
    tp[q] = phase[yb + (in ? i - 1 : 0)];
  }
  bool hit[kSearchSub];
  double wrap[kSearchSub], prev[kSearchSub];
  unsigned long long bal[kSearchSub];
#pragma unroll
  for (int q = 0; q < kSearchSub; ++q) {
    const int i = i0 + 256 * q + threadIdx.x;
    wrap[q] = wrap_two_pi(tc[q]);
    prev[q] = wrap_two_pi(tp[q]);
    // pulse at index i-1 when |wrap[i] - wrap[i-1]| > pi (:254-259)
    hit[q] = i >= 1 && i < ylen && fabs(wrap[q] - prev[q]) > kPi;
    bal[q] = __ballot(hit[q]);
    if (lane == 0) wave_cnt[q][wv] = __popcll(bal[q]);
  }
  __syncthreads();
  if (!WRITE) {
    if (threadIdx.x == 0) {
      int c = 0;
#pragma unroll
      for (int q = 0; q < kSearchSub; ++q) c += wave_cnt[q][0] + wave_cnt[q][1] + wave_cnt[q][2] + wave_cnt[q][3];
      cnt_u[tile] = c;
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < kSearchSub; ++q) {
    const int i = i0 + 256 * q + threadIdx.x;
    int base = count;
    for (int w = 0; w < wv; ++w) base += wave_cnt[q][w];
    if (hit[q]) {
      const int dst = base + __popcll(bal[q] & ((1ull << lane) - 1ull));
      const double y1 = prev[q] - 2.0 * kPi, y2 = wrap[q];  // :271-274
      const double xx = -y1 / (y2 - y1);
      pulse_idx[yb + dst] = i - 1;
      pulse_shift[yb + dst] = xx / fs;
    }
    count += wave_cnt[q][0] + wave_cnt[q][1] + wave_cnt[q][2] + wave_cnt[q][3];
  }
}

// Pulse numbers of a part of the batch: utterance utts[k] owns [off[u], off[u] + cnt[u]), numbered from `base` on in
// list order; info[0] = the part's total, info[1] = its largest count (pinned host memory).
__global__ __launch_bounds__(256) void synth_pulse_off_kernel(const int* __restrict__ utts, const int* __restrict__ cnt,
                                                              int n_list, int64_t base, int64_t* __restrict__ off,
                                                              int64_t* __restrict__ info) {
  __shared__ int64_t part[256];
  __shared__ int mx[256];
  const int per = (n_list + 255) / 256;
  const int lo = threadIdx.x * per, hi = imin(n_list, lo + per);
  int64_t sum = 0;
  int m = 0;
  for (int k = lo; k < hi; ++k) {
    const int c = cnt[utts[k]];
    sum += c;
    m = imax(m, c);
  }
  part[threadIdx.x] = sum;
  mx[threadIdx.x] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t run = 0;
    int mm = 0;
    for (int i = 0; i < 256; ++i) {
      const int64_t v = part[i];
      part[i] = run;
      run += v;
      mm = imax(mm, mx[i]);
    }
    info[0] = run;
    info[1] = mm;
    __threadfence_system();
  }
  __syncthreads();
  int64_t run = base + part[threadIdx.x];
  for (int k = lo; k < hi; ++k) {
    const int u = utts[k];
    off[u] = run;
    run += cnt[u];
  }
}

__global__ void synth_dc_remover_kernel(int fft_size, double* __restrict__ dcr) {   // GetDCRemover :322-334
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double dc = 0.0;
  const int h = fft_size / 2;
  for (int i = 0; i < h; ++i) {
    dcr[i] = 0.5 - 0.5 * cos(2.0 * kPi * (i + 1.0) / (1.0 + fft_size));
    dcr[fft_size - i - 1] = dcr[i];
    dc += dcr[i] * 2.0;
  }
  for (int i = 0; i < h; ++i) {
    dcr[i] /= dc;
    dcr[fft_size - i - 1] = dcr[i];
  }
}

__device__ __forceinline__ double safe_ap(double v) {          // common.h:111-113
  const double m = 0.999999999999 < v ? 0.999999999999 : v;
  return 0.001 > m ? 0.001 : m;
}

// GetMinimumPhaseSpectrum (common.cpp:182-220) for one wavefront.  ls[0..H] (LDS) holds the
// log spectrum; on exit mp[m] is the minimum-phase spectrum at bins lane + 64 m (m < M) and
// mp[M] at bin H = N (all lanes compute it).  The cepstrum's imaginary parts (rounding noise of
// a real symmetric transform) are dropped, so the reference's c2c becomes a second r2c.
template <int N>
__device__ __forceinline__ void minimum_phase(const double* ls, cpx* img, const FftTw<N>& tw, int lane,
                                              cpx (&mp)[N / 64 + 1]) {
  constexpr int M = N / 64, F = 2 * N, H = N;
  lane = opaque_lane(lane);                    // the mirrored indices are rebuilt per call, not kept from the last one
  cpx v[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int i0 = 2 * (lane + 64 * m), i1 = i0 + 1;
    v[m] = make_double2(ls[i0 <= H ? i0 : F - i0], ls[i1 <= H ? i1 : F - i1]);   // mirroring :184-187
  }
  rfft_forward<N>(v, img, img, tw, lane);
  // folded cepstrum c[0]=C0, c[j]=2C[j] (0<j<H), c[H]=C[H], 0 above (:193-203), packed for the next r2c
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int i0 = 2 * (lane + 64 * m), i1 = i0 + 1;
    // reads with clamped indices and the predicate on the value (a read behind a per-lane branch is waited for on its own)
    const double xa = img[imin(i0, H)].x, xb = img[imin(i1, H)].x;
    const double a = i0 <= H ? ((i0 == 0 || i0 == H) ? xa : 2.0 * xa) : 0.0;
    const double b = i1 < H ? 2.0 * xb : 0.0;
    v[m] = make_double2(a, b);
  }
  rfft_forward<N>(v, img, img, tw, lane);
#pragma unroll
  for (int m = 0; m <= M; ++m) {
    const int k = m < M ? lane + 64 * m : N;
    const cpx s = img[k];
    const double amp = wm_exp(s.x / F);                              // :210-218
    double sn, cs;
    // the library's call here: in this loop it is 45 vector instructions; wm_sincospi is 30 plus 34 scalar moves for its
    // coefficients, and the scalar registers to keep those across the nine bins are not there (they come back as
    // v_readlane reloads): 4.86 ms against 4.40 for the kernel (tools/ab.sh).  A 64-point table in LDS with short
    // polynomials around it is 26 instructions, but the kernel then allocates 168 registers (three waves per SIMD)
    if (m < M) {
      sincospi(s.y * (1.0 / (kPi * F)), &sn, &cs);                // phase in half-turns: no Payne-Hanek path
    } else {
      sn = 0.0;                                                   // the Nyquist bin of a real sequence's transform is real
      cs = 1.0;                                                   // (rfft_forward writes its imaginary part as 0.0)
    }
    mp[m] = make_double2(amp * cs, amp * sn);
    __builtin_amdgcn_sched_barrier(0);                            // one bin at a time: keeps the VGPR peak low
  }
  wave_sync();
}

// The PHASES of the minimum-phase spectra of two log spectra at once (a voiced pulse needs both: the periodic and the
// aperiodic part, synthesis.cpp:105-138 and :38-68).
//
// GetMinimumPhaseSpectrum (common.cpp:182-220) takes the transform of the mirrored (real, even) log spectrum -- the
// cepstrum C, real -- folds it (c_j = 2 C_j for 0 < j < H, c_0 = C_0, c_H = C_H, 0 above) and transforms again:
// S_k = sum_j c_j e^(-i pi j k / H).  Its real part is F times the log spectrum it started from (the fold undoes the
// mirror), so the amplitude exp(Re S_k / F) is the square root of the spectrum value, which the caller has; only
//     Im S_k = -2 sum_{0<j<H} C_j sin(pi j k / H)
// needs the transforms: a cosine transform of the log spectrum, then a sine transform of the result.  Both are real
// and linear, so ONE complex sequence w = lp + i la carries both spectra through them, and the symmetry does the
// rest: the even (odd) extension of w to F = 2 H points has an H-point transform of its even samples E and of its
// odd samples O = e^(i pi j / H) R with E_(-j) = +-E_j, R_(-j) = +-R_j, so the ONE complex H-point transform Z of
// z_n = w_2n + i w_2n+1 separates into both from Z_j and Z_(H-j):
//     R_j = -(Z_j -+ Z_(H-j)) / (2 sin(pi j / H)),   E_j = Z_j - i e^(i pi j / H) R_j,   transform_j = E_j + R_j.
// Two complex transforms of N points and two partner exchanges replace four real transforms of 2 N points with their
// splits (4 x 490 -> 2 x ~520 vector instructions at N = 512), and the exp() of the amplitude goes.  The division by
// the sine amplifies rounding at low j by up to H / pi: 1e-13 of the cepstrum's scale, 6e-14 rad in the phases
// (tools/minphase_pair_proto.py), far below the 1e-8 at which y is compared.
// lsp / lsa: LDS, [0 .. H] each, may lie inside the FFT image (they are read before its first exchange).
// On exit php[m] / pha[m] = Im S_k / F for k = lane + 64 m; the phase of bin H is 0.
template <int N>
__device__ __forceinline__ void minimum_phase_pair(const double* lsp, const double* lsa, cpx* img, const FftTw<N>& tw,
                                                   int lane, double (&php)[N / 64], double (&pha)[N / 64]) {
  constexpr int M = N / 64, F = 2 * N, H = N;
  static_assert(M >= 8 && M % 8 == 0, "the constant factors below are eighths of a half turn");
  lane = opaque_lane(lane);
  cpx v[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int i0 = 2 * (lane + 64 * m), i1 = i0 + 1;
    const int a0 = i0 <= H ? i0 : F - i0, a1 = i1 <= H ? i1 : F - i1;             // the even extension
    const double p0 = lsp[a0], q0 = lsa[a0], p1 = lsp[a1], q1 = lsa[a1];
    v[m] = make_double2(p0 - q1, q0 + p1);                                           // w_2n + i w_2n+1
  }
  fft_forward<N>(v, img, tw, lane);
  // e^(i pi j / H), j = lane + 64 m -- the lane's base turned by 32 m / M sixty-fourths of a turn -- and 1 / (2 sin):
  // rebuilt where they are used (a product with a constant, a reciprocal with one Newton step) instead of kept in
  // 4 M registers across the transforms
  auto turn = [&](int m, cpx e0, cpx& e, double& inv) {
    const int k64 = 32 * m / M;
    e = k64 % 16 == 0 ? e0 : cmul(e0, cconj(cis64(k64 % 16)));
    if (k64 / 16 == 1) e = make_double2(-e.y, e.x);
    const double r0 = __builtin_amdgcn_rcp(e.y);
    const double r1 = __builtin_fma(__builtin_fma(-e.y, r0, 1.0), r0, r0);           // 1 / sin to 1e-16 relative
    inv = (m == 0 && lane == 0) ? 0.0 : 0.5 * r1;                                     // j = 0: nothing to separate
  };
  wave_sync();
#pragma unroll
  for (int m = 0; m < M; ++m) img[lane + 64 * m] = v[m];                            // Z, plain layout
  wave_sync();
  {
    const cpx e0 = cconj(make_double2(opaque_d(tw.wsplit.x), opaque_d(tw.wsplit.y)));   // W_2N^(-lane)
    cpx zp[M];
#pragma unroll
    for (int m = 0; m < M; ++m) zp[m] = img[(N - (lane + 64 * m)) & (N - 1)];
    wave_sync();                                                                     // partners read: the image may go
#pragma unroll
    for (int m = 0; m < M; ++m) {
      cpx e;
      double inv;
      turn(m, e0, e, inv);
      const cpx R = make_double2((zp[m].x - v[m].x) * inv, (zp[m].y - v[m].y) * inv);   // -(Z_j - Z_(H-j)) / (2 sin)
      // i e^(i th) R = (-(e.x R.y + e.y R.x), e.x R.x - e.y R.y);  A = Z - that + R;  g = 2 A
      const double ax = v[m].x + __builtin_fma(e.x, R.y, e.y * R.x) + R.x;
      const double ay = v[m].y - __builtin_fma(e.x, R.x, -(e.y * R.y)) + R.y;
      const cpx g = (m == 0 && lane == 0) ? make_double2(0.0, 0.0) : make_double2(2.0 * ax, 2.0 * ay);
      img[lane + 64 * m] = g;                                                         // folded cepstra, plain; g_0 = 0
      __builtin_amdgcn_sched_barrier(0);
    }
    if (lane == 0) img[H] = make_double2(0.0, 0.0);                                   // g_H = 0
  }
  wave_sync();
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int i0 = 2 * (lane + 64 * m), i1 = i0 + 1;
    const cpx c0 = img[i0 <= H ? i0 : F - i0], c1 = img[i1 <= H ? i1 : F - i1];
    const double s0 = i0 <= H ? 1.0 : -1.0, s1 = i1 <= H ? 1.0 : -1.0;              // the odd extension
    v[m] = make_double2(s0 * c0.x - s1 * c1.y, s0 * c0.y + s1 * c1.x);
  }
  fft_forward<N>(v, img, tw, lane);
  wave_sync();
#pragma unroll
  for (int m = 0; m < M; ++m) img[lane + 64 * m] = v[m];
  wave_sync();
  {
    const cpx e0 = cconj(make_double2(opaque_d(tw.wsplit.x), opaque_d(tw.wsplit.y)));
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int k = lane + 64 * m;
      const cpx zp = img[(N - k) & (N - 1)];
      cpx e;
      double inv;
      turn(m, e0, e, inv);
      const cpx R = make_double2(-(zp.x + v[m].x) * inv, -(zp.y + v[m].y) * inv);     // -(Z_k + Z_(H-k)) / (2 sin)
      const double gx = v[m].x + __builtin_fma(e.x, R.y, e.y * R.x) + R.x;
      const double gy = v[m].y - __builtin_fma(e.x, R.x, -(e.y * R.y)) + R.y;
      // transform of the odd extension = -2 i sum c sin = 2 X_a - 2 i X_p with Im S = -X: Im S_p = Im G / 2, Im S_a = -Re G / 2
      const bool zero = m == 0 && lane == 0;
      php[m] = zero ? 0.0 : gy * (0.5 / F);
      pha[m] = zero ? 0.0 : -gx * (0.5 / F);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  wave_sync();
}

// Everything a pulse needs that is not spectral data, gathered by a fully parallel kernel so that
// the per-pulse kernel starts with ONE (scalar) load instead of a binary search over the pulse
// offsets followed by four levels of dependent loads.
struct PulseRec {
  int64_t fbase;      // first frame of the utterance
  int nf;             // frames of the utterance
  int idx;            // pulse sample index (pulse_locations_index)
  int roff;           // idx - idx of the utterance's first pulse: randn table offset (:341, :369)
  int noise_size;     // idx of the next pulse - idx (:369; 0 for the last pulse)
  double shift;       // pulse_locations_time_shift
  double cvuv;        // interpolated vuv at the pulse
};

struct PulseVoicedPred {                       // synthesis.cpp:197: the pulses that have a periodic response
  const PulseRec* rec;
  __device__ bool operator()(int i) const { return rec[i].cvuv > 0.5; }
};

// The overlap-add finds the pulses of a stretch of output samples in a table: first[m] = number (within the
// utterance) of the first pulse whose index is at least kOlaStep * m, m = 0 .. ceil(ylen / kOlaStep); utterance u owns
// the entries from ola_table_base(yb, u) on (two spare entries per utterance keep the ranges apart whatever yb is).
constexpr int kOlaStep = 128;
__host__ __device__ inline int64_t ola_table_base(int64_t yb, int u) { return yb / kOlaStep + 2 * (int64_t)u; }

__global__ __launch_bounds__(256) void synth_pulse_rec_kernel(
    const int* __restrict__ utts, const int64_t* __restrict__ f_off, const int64_t* __restrict__ y_off,
    const int64_t* __restrict__ p_off, const int* __restrict__ p_cnt, const int* __restrict__ pulse_idx,
    const double* __restrict__ pulse_shift, const double* __restrict__ vuv, PulseRec* __restrict__ rec,
    int* __restrict__ first) {
  const int u = utts[blockIdx.y];
  const int64_t pb = p_off[u];
  const int np = p_cnt[u];
  const int64_t yb = y_off[u];
  const int ylen = (int)(y_off[u + 1] - yb);
  int* fu = first + ola_table_base(yb, u);
  const int m_last = (ylen + kOlaStep - 1) / kOlaStep;
  for (int pi = blockIdx.x * 256 + threadIdx.x; pi < np; pi += gridDim.x * 256) {
    PulseRec r;
    r.fbase = f_off[u];
    r.nf = (int)(f_off[u + 1] - f_off[u]);
    r.idx = pulse_idx[yb + pi];
    r.roff = r.idx - pulse_idx[yb];
    r.noise_size = pulse_idx[yb + imin(np - 1, pi + 1)] - r.idx;
    r.shift = pulse_shift[yb + pi];
    r.cvuv = vuv[yb + r.idx];
    rec[pb + pi] = r;
    // table entries whose sample kOlaStep * m lies in (index of the pulse before, index of this pulse]: every entry
    // is written by exactly one pulse; the last pulse also writes the entries behind it (= np: no such pulse)
    const int prev = pi > 0 ? pulse_idx[yb + pi - 1] : -1;
    for (int m = prev < 0 ? 0 : prev / kOlaStep + 1; m <= imin(m_last, r.idx / kOlaStep); ++m) fu[m] = pi;
    if (pi == np - 1)
      for (int m = r.idx / kOlaStep + 1; m <= m_last; ++m) fu[m] = np;
  }
}

// One wavefront per pulse.  resp[(p - p_begin) * F + j] = response[j] of synthesis.cpp:211-215.
template <int F>
__global__ __launch_bounds__(64, F >= 4096 ? 1 : (F == 1024 ? 4 : (F < 1024 ? 3 : 2))) void synth_pulse_kernel(
    const double* __restrict__ sp, const double* __restrict__ ap, const PulseRec* __restrict__ rec,
    const double* __restrict__ dcr, const uint32_t* __restrict__ rtab, int fs, double fp, int64_t p_begin,
    int64_t p_end, const int* __restrict__ perm, double* __restrict__ resp) {
  constexpr int N = F / 2, M = N / 64, H = F / 2, MB = M + 1;
  // LEAN (fft_size 2048: 16 complex values per lane and array): to run two waves per SIMD nothing of a spectrum's
  // size lives through a transform -- the interpolated envelope and aperiodicity are fetched again for the aperiodic
  // half instead of being kept (68 registers), the periodic response waits in the response row it is headed for
  // (32), and the log spectrum shares the LDS image of the transform that consumes it (8 KB: 9 workgroups per CU
  // instead of 6).  One wave per SIMD had nothing to hide the LDS round trips of its transforms behind.
  constexpr bool LEAN = F == 2048 || F == 1024;
  // PAIRED: the two minimum-phase spectra of a voiced pulse through one pair of complex transforms
  // (minimum_phase_pair).  At fft 2048 the pair's registers cost four spilled ones at two waves per SIMD and the
  // kernel is 1 % slower with it than without (A/B at 48 kHz: 10.08 against 9.95 ms), so only fft 1024 takes it.
  constexpr bool PAIRED = F == 1024;
  __shared__ __attribute__((aligned(16))) double smem[2 * FftLds<N>::kElems + (LEAN ? 0 : H + 2)];
  cpx* img = reinterpret_cast<cpx*>(smem);
  double* ls = LEAN ? smem : smem + 2 * FftLds<N>::kElems;
  const int lane0 = threadIdx.x;
  FftTw<N> tw;
  tw.init(lane0);

  // perm lists the chunk's voiced pulses (7 transforms) before its unvoiced ones (4): round-robin over the
  // list gives every wave the same number of each (partition.hpp)
  WM_FOR_EACH_LISTED(pi, perm, p_end - p_begin) {
    const int64_t p = p_begin + pi;
    const int lane = opaque_lane(lane0);
    const PulseRec r = rec[p];                                      // wave-uniform
    const int nf = r.nf;
    const int idx = r.idx;
    const int noise_size = r.noise_size;                            // synthesis.cpp:369
    const double cvuv = r.cvuv;
    const double ctime = idx / (double)fs;                          // pulse_locations = time_axis[i]
    const double shift = r.shift;

    // ---- GetSpectralEnvelope / GetAperiodicRatio (:140-178) ----
    const int ff = imin(nf - 1, (int)floor(ctime / fp));
    const int fc = imin(nf - 1, (int)ceil(ctime / fp));
    const double wgt = ff == fc ? 0.0 : ctime / fp - ff;    // beyond the last frame both indices are clamped: a copy there too
    const double* s0 = sp + (r.fbase + ff) * (int64_t)(H + 1);
    const double* s1 = sp + (r.fbase + fc) * (int64_t)(H + 1);
    const double* a0 = ap + (r.fbase + ff) * (int64_t)(H + 1);
    const double* a1 = ap + (r.fbase + fc) * (int64_t)(H + 1);
    auto spectral = [&](double (&env)[MB], double (&rat)[MB]) {
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int k = m < M ? lane + 64 * m : H;
        // synthesis.cpp:140-178 copies row ff when ff == fc and interpolates otherwise; with wgt = 0 (set above for
        // that case) the interpolation IS the copy (1 x + 0 y = x for finite y, and y is then the same row), so there
        // is one form and no branch inside the loop: behind one, every bin's loads were a trip to memory of their own
        env[m] = (1.0 - wgt) * fabs(s0[k]) + wgt * fabs(s1[k]);
        const double a = (1.0 - wgt) * safe_ap(a0[k]) + wgt * safe_ap(a1[k]);
        rat[m] = a * a;
      }
    };
    double env_keep[LEAN ? 1 : MB], rat_keep[LEAN ? 1 : MB];
    double rat0;
    if constexpr (LEAN) {
      const double a = (1.0 - wgt) * safe_ap(a0[0]) + wgt * safe_ap(a1[0]);   // bin 0, every lane
      rat0 = uniform_d(a * a);
    } else {
      spectral(env_keep, rat_keep);
      rat0 = __shfl(rat_keep[0], 0, 64);
    }
    double* out = resp + (p - p_begin) * (int64_t)F;

    // ---- GetPeriodicResponse (:105-138) ----
    double xp[LEAN ? 1 : M];            // periodic c2r output, x-index i = 2n + c for n < N/2 (first half)
    double dc = 0.0;
    const bool periodic = !(cvuv <= 0.5 || rat0 > 0.999);
#pragma unroll
    for (int m = 0; m < (LEAN ? 1 : M); ++m) xp[m] = 0.0;
    // A voiced pulse needs two minimum-phase spectra (periodic and aperiodic part): their phases come from ONE pair
    // of complex transforms (minimum_phase_pair), their amplitudes are the square roots of the spectra themselves
    if (periodic) {
      wave_sync();
      auto log_periodic = [&](const double (&env)[MB], const double (&rat)[MB]) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
          ls[lane + 64 * m] = wm_log(env[m] * (1.0 - rat[m]) + kSafe) / 2.0;
          __builtin_amdgcn_sched_barrier(0);
        }
        if (lane == 0) ls[H] = wm_log(env[M] * (1.0 - rat[M]) + kSafe) / 2.0;
      };
      cpx mp[MB];
      if constexpr (PAIRED) {
        double* ls2 = ls + H + 2;                                   // the aperiodic log spectrum beside the periodic one
        {
          double env[MB], rat[MB];
          spectral(env, rat);
#pragma unroll
          for (int m = 0; m < MB; ++m) {
            const int k = m < M ? lane + 64 * m : H;
            const double lp = wm_log(env[m] * (1.0 - rat[m]) + kSafe) / 2.0;
            const double la = wm_log(env[m] * rat[m]) / 2.0;        // cvuv > 0.5 here (synthesis.cpp:53-56)
            if (m < M || lane == 0) {
              ls[k] = lp;
              ls2[k] = la;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        wave_sync();
        {
          double php[M], pha[M];
          minimum_phase_pair<N>(ls, ls2, img, tw, lane, php, pha);
          // the aperiodic part's phases wait in the first half of the response row (free until the response is
          // written; every lane re-reads its own), the periodic part's in the slots of the spectrum they become
#pragma unroll
          for (int m = 0; m < M; ++m) {
            out[lane + 64 * m] = pha[m];
            mp[m] = make_double2(0.0, php[m]);
          }
          mp[M] = make_double2(0.0, 0.0);                             // bin H: real
        }
        {
          double env[MB], rat[MB];
          spectral(env, rat);
#pragma unroll
          for (int m = 0; m < MB; ++m) mp[m].x = wm_sqrt(env[m] * (1.0 - rat[m]) + kSafe);
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          double sn = 0.0, cs = 1.0;
          if (m < M) sincospi(mp[m].y * (1.0 / kPi), &sn, &cs);
          mp[m] = make_double2(mp[m].x * cs, mp[m].x * sn);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        if constexpr (LEAN) {
          double env[MB], rat[MB];
          spectral(env, rat);
          log_periodic(env, rat);
        } else {
          log_periodic(env_keep, rat_keep);
        }
        wave_sync();
        minimum_phase<N>(ls, img, tw, lane, mp);
      }
      const double coef = 2.0 * kPi * shift * fs / F;               // :130-131
      // cos(coef k) for k = lane + 64 m by rotation from cos/sin(coef lane) in steps of 64 coef
      // (the reference evaluates cos per bin; the rotation is within 1e-15 of it); bin H directly
      double rc, rs, dc64, ds64;
      // in half-turns (coef / pi = 2 shift fs / F): the argument's own rounding, 1e-16 of up to 2000 half-turns,
      // moves a phase by 1e-12 rad at most -- the size of the rounding of coef * k itself
      const double ch = coef * (1.0 / kPi);
      double snH, reH;
      wm_sincospi(ch * lane, &rs, &rc);
      wm_sincospi(ch * 64.0, &ds64, &dc64);
      wm_sincospi(ch * H, &snH, &reH);
#pragma unroll
      for (int m = 0; m < MB; ++m) {                                // :88-100
        const int k = m < M ? lane + 64 * m : H;
        const double re2 = m < M ? rc : reH;
        {
          const double nc = rc * dc64 - rs * ds64;
          rs = rs * dc64 + rc * ds64;
          rc = nc;
        }
        // synthesis.cpp:96 takes the sine as sqrt(1 - cos^2): always >= 0.  The rotated cosine can pass 1 by a
        // rounding where the reference's cos() cannot: wm_sqrt returns 0 there instead of a NaN
        const double im2 = wm_sqrt(1.0 - re2 * re2);
        const cpx s = make_double2(mp[m].x * re2 + mp[m].y * im2, mp[m].y * re2 - mp[m].x * im2);
        if (m < M || lane == 0) img[k] = s;
        __builtin_amdgcn_sched_barrier(0);
      }
      cpx v[M];
      rfft_backward<N>(img, v, img, tw, lane);
      // fftshift + RemoveDCComponent (:73-82, :135-137): dc = sum of the shifted second half = x[0..H)
#pragma unroll
      for (int m = 0; m < M / 2; ++m) {
        if constexpr (LEAN) {                                       // waits where it is headed for (every lane re-reads its own)
          const int i0 = 2 * (lane + 64 * m);
          *reinterpret_cast<cpx*>(out + i0 + H) = v[m];
        } else {
          xp[2 * m] = v[m].x;
          xp[2 * m + 1] = v[m].y;
        }
        dc += v[m].x + v[m].y;
      }
      dc = wave_sum(dc);
      if constexpr (LEAN) dc = uniform_d(dc);
      wave_sync();
    }

    // ---- GetAperiodicResponse (:38-68) ----
    wave_sync();
    cpx mp[MB];
    if (PAIRED && periodic) {
      // phases from the pair above (parked in the response row); amplitude sqrt(env * rat) (cvuv > 0.5 on a periodic pulse)
      {
        const int lp_ = opaque_lane(lane);
#pragma unroll
        for (int m = 0; m < M; ++m) mp[m] = make_double2(0.0, out[lp_ + 64 * m]);
        mp[M] = make_double2(0.0, 0.0);
      }
      {
        double env[MB], rat[MB];
        spectral(env, rat);
#pragma unroll
        for (int m = 0; m < MB; ++m) mp[m].x = wm_sqrt(env[m] * rat[m]);
      }
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        double sn = 0.0, cs = 1.0;
        if (m < M) sincospi(mp[m].y * (1.0 / kPi), &sn, &cs);
        mp[m] = make_double2(mp[m].x * cs, mp[m].x * sn);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      {
        auto log_aperiodic = [&](const double (&env)[MB], const double (&rat)[MB]) {
#pragma unroll
          for (int m = 0; m < MB; ++m) {
            const int k = m < M ? lane + 64 * m : H;
            const double val = wm_log(cvuv != 0.0 ? env[m] * rat[m] : env[m]) / 2.0;
            if (m < M || lane == 0) ls[k] = val;
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        if constexpr (LEAN) {
          double env[MB], rat[MB];
          spectral(env, rat);
          log_aperiodic(env, rat);
        } else {
          log_aperiodic(env_keep, rat_keep);
        }
      }
      wave_sync();
      minimum_phase<N>(ls, img, tw, lane, mp);
    }
    // GetNoiseSpectrum (:19-33)
    cpx v[M];
    {
      // LEAN: the draws are fetched here (their addresses hang on a fenced lane), not ahead of the transforms above
      const int ln = LEAN ? opaque_lane(lane) : lane;
      const int roff = r.roff;
      double sum = 0.0;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const int i0 = 2 * (ln + 64 * m);
        const double n0 = i0 < noise_size ? randn_at(rtab, roff + i0) : 0.0;
        const double n1 = i0 + 1 < noise_size ? randn_at(rtab, roff + i0 + 1) : 0.0;
        v[m] = make_double2(n0, n1);
        sum += n0 + n1;
      }
      const double avg = wave_sum(sum) / noise_size;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const int i0 = 2 * (ln + 64 * m);
        if (i0 < noise_size) v[m].x -= avg;
        if (i0 + 1 < noise_size) v[m].y -= avg;
      }
    }
    rfft_forward_nz<N>(v, img, img, tw, lane, (noise_size + 127) >> 7);   // the draws up to the next pulse, zeros behind
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const int k = m < M ? lane + 64 * m : H;
      const cpx ns = img[k];
      const cpx s = make_double2(mp[m].x * ns.x - mp[m].y * ns.y, mp[m].x * ns.y + mp[m].y * ns.x);
      if (m < M || lane == 0) img[k] = s;
    }
    rfft_backward<N>(img, v, img, tw, lane);

    // ---- response = (periodic * sqrt(noise_size) + aperiodic) / fft_size (:211-215), fftshifted ----
    const double sq = sqrt((double)noise_size);
    cpx xq[LEAN ? M / 2 : 1];
    const int lo = LEAN ? opaque_lane(lane) : lane;   // LEAN: the DC remover's table is fetched here, not ahead of the transforms
    if constexpr (LEAN) {
#pragma unroll
      for (int m = 0; m < M / 2; ++m)
        xq[m] = periodic ? *reinterpret_cast<const cpx*>(out + 2 * (lo + 64 * m) + H) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int n = lo + 64 * m;
      const int i0 = 2 * n;                           // x-index; shifted position j = (i + H) mod F
      double r0, r1;
      if (m < M / 2) {                                // i < H  ->  j = i + H (second half)
        const double x0 = LEAN ? xq[LEAN ? m : 0].x : xp[LEAN ? 0 : 2 * m];
        const double x1 = LEAN ? xq[LEAN ? m : 0].y : xp[LEAN ? 0 : 2 * m + 1];
        const double2 dr = *reinterpret_cast<const double2*>(dcr + i0 + H);   // unconditional: a load behind the
        const double p0 = periodic ? x0 - dc * dr.x : 0.0;                    // (uniform) branch is waited for on its own
        const double p1 = periodic ? x1 - dc * dr.y : 0.0;
        r0 = (p0 * sq + v[m].x) / F;
        r1 = (p1 * sq + v[m].y) / F;
        out[i0 + H] = r0;
        out[i0 + 1 + H] = r1;
      } else {                                        // i >= H ->  j = i - H (first half, periodic overwritten)
        const double2 dr = *reinterpret_cast<const double2*>(dcr + i0 - H);
        const double p0 = periodic ? -dc * dr.x : 0.0;
        const double p1 = periodic ? -dc * dr.y : 0.0;
        r0 = (p0 * sq + v[m].x) / F;
        r1 = (p1 * sq + v[m].y) / F;
        out[i0 - H] = r0;
        out[i0 + 1 - H] = r1;
      }
    }
    wave_sync();
  }
}

}  // namespace wm
#include "synth_pulse_bp.hpp"
namespace wm {

// y[n] += sum over pulses p (of this utterance, within [p_begin, p_end)) covering n, in pulse order:
// index = j + idx - F/2 + 1  (synthesis.cpp:378-383)  ->  j = n - idx + F/2 - 1.
// ONE WAVEFRONT owns a stretch of kOlaSeg consecutive output samples (lane l the pairs 2 l + 128 q, q < kOlaQ) and
// streams through the response rows of the pulses that reach into it, in pulse order, adding into registers: the
// association is the reference's sequential += per sample, a row is read once per stretch it touches (1 + F / kOlaSeg
// stretches: 3 at fft 1024), in 16-byte loads of consecutive lanes.  A chunk of 128 samples that a row does not reach
// is skipped (wave-uniform), one that it covers whole takes one load per lane, the two at the row's ends two
// predicated 8-byte loads.  Adding nothing where the round-3 kernel added an exact + 0.0 is the same sum: the
// accumulator starts from y (+0.0 or a sum) and x + 0.0 == x unless x is -0.0, which a sum starting at +0.0 never is.
// The pulse range of a stretch comes from the first-pulse table (synth_pulse_rec_kernel), not from a search.
// Round 3 gave a thread one sample of a tile of 256 and walked every pulse near the tile with 8-byte loads, four in
// flight: 0.8 TB/s where a stream reads at 6, 15.8 ms busy per configs[4] step beside the pulse kernel.
constexpr int kOlaQ = 4, kOlaSeg = 128 * kOlaQ, kOlaWaves = 4;
__global__ __launch_bounds__(64 * kOlaWaves) void synth_ola_kernel(const int* __restrict__ utts,
                                                                   const int64_t* __restrict__ y_off,
                                                                   const int64_t* __restrict__ p_off,
                                                                   const int* __restrict__ p_cnt,
                                                                   const int* __restrict__ pulse_idx,
                                                                   const int* __restrict__ first, int fft_size,
                                                                   int64_t p_begin, int64_t p_end,
                                                                   const double* __restrict__ resp,
                                                                   double* __restrict__ y) {
  const int u = utts[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int64_t yb = y_off[u];
  const int ylen = (int)(y_off[u + 1] - yb);
  const int n0 = (blockIdx.x * kOlaWaves + (threadIdx.x >> 6)) * kOlaSeg;      // wave-uniform
  const int64_t pu = p_off[u];
  const int np = p_cnt[u];
  // pulses of another part of the batch (their numbers lie outside this launch's piece) or none at all
  if (n0 >= ylen || np == 0 || pu >= p_end || pu + np <= p_begin) return;
  const int h = fft_size / 2;
  // pulses with n0 - h <= idx <= n0 + kOlaSeg + h - 2 reach into the stretch; the table brackets them (a few more on
  // the left, whose rows end before the stretch and are skipped chunk by chunk)
  const int* fu = first + ola_table_base(yb, u);
  const int m_last = (ylen + kOlaStep - 1) / kOlaStep;
  const int lo_s = imax(0, n0 - h), hi_s = n0 + kOlaSeg + h - 2;
  int pa = fu[lo_s / kOlaStep];
  int pb = fu[imin(m_last, hi_s / kOlaStep + 1)];
  const int64_t rel_a = p_begin - pu, rel_b = p_end - pu;        // the piece's pulses, utterance-relative
  pa = rel_a > pa ? (int)(rel_a < np ? rel_a : np) : pa;
  pb = rel_b < pb ? (int)(rel_b > 0 ? rel_b : 0) : pb;
  if (pa >= pb) return;
  const int* pidx = pulse_idx + yb;
  double2_a8 acc[kOlaQ];
#pragma unroll
  for (int q = 0; q < kOlaQ; ++q) {
    const int n = n0 + 128 * q + 2 * lane;
    acc[q].x = n < ylen ? y[yb + n] : 0.0;
    acc[q].y = n + 1 < ylen ? y[yb + n + 1] : 0.0;
  }
  for (int base = pa; base < pb; base += 64) {
    const int cnt = imin(64, pb - base);
    const int my_idx = pidx[base + imin(lane, cnt - 1)];         // the indices of up to 64 pulses, one per lane
    for (int k = 0; k < cnt; ++k) {
      const int idx = __builtin_amdgcn_readlane(my_idx, k);
      const int s = idx - h + 1 - n0;                             // stretch-relative sample of response[0]
      const double* row = resp + ((int64_t)(pu + base + k) - p_begin) * fft_size;
#pragma unroll
      for (int q = 0; q < kOlaQ; ++q) {
        const int c0 = 128 * q - s;                               // response index of the chunk's first sample
        if (c0 + 127 < 0 || c0 >= fft_size) continue;             // the row does not reach the chunk
        const int j = c0 + 2 * lane;
        if (c0 >= 0 && c0 + 127 < fft_size) {
          const double2_a8 v = *reinterpret_cast<const double2_a8*>(row + j);
          acc[q].x += v.x;
          acc[q].y += v.y;
        } else {
          const bool in0 = j >= 0 && j < fft_size, in1 = j + 1 >= 0 && j + 1 < fft_size;
          const double v0 = row[imin(fft_size - 1, imax(0, j))];
          const double v1 = row[imin(fft_size - 1, imax(0, j + 1))];
          if (in0) acc[q].x += v0;
          if (in1) acc[q].y += v1;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kOlaQ; ++q) {
    const int n = n0 + 128 * q + 2 * lane;
    if (n < ylen) y[yb + n] = acc[q].x;
    if (n + 1 < ylen) y[yb + n + 1] = acc[q].y;
  }
}

// Synthesis in two stages.  The PREPARE stage is everything that depends on f0 only: sample-rate f0 / vuv, the time
// base, the pulse list and its per-pulse records; it ends with the one host round trip of the path (the pulse count
// sizes the response scratch).  The RENDER stage turns sp / ap into responses and overlap-adds them.
//
// Both work on a PART of the batch: a list of utterances (Batch::d_syn_order holds the batch's utterances sorted by
// output length; the identity order when the batch is prepared as one part).  launch_analyze_synthesize()
// (context.cpp) prepares the whole batch as one part on a side stream while CheapTrick and D4C occupy the main one.
// launch_synthesis() -- Synthesis alone, BASELINE.json configs[4] -- has nothing to hide the prepare stage behind
// but its own render stage, so it splits the batch: part A, the shortest utterances making up a sixth of the output
// samples, is prepared on the caller's stream (its phase chain is as long as ITS longest utterance, a third of the
// batch's); while A is rendered, the rest is prepared on a third stream.  An utterance's pulses keep their order and
// every sample its order of additions, so y does not depend on the split (tests: against the one-part form, bit for bit).
struct SynPart {
  const int* d_list;      // utterance numbers of the part (device)
  int n;                  // how many
  int max_y_len;          // longest output among them
  int64_t p_base;         // number of the part's first pulse
  int64_t total_p = 0;    // its pulses (after the host round trip)
  int max_np = 0;
};

static int synthesis_arena(Batch& b) {
  Context& c = *b.ctx;
  const int F = b.p.fft_size;
  if (b.d_pulse_idx) return WM_OK;
  // one allocation for the work arrays of this batch (sections aligned to 256 bytes)
  const size_t ny = (size_t)b.total_y, nu = (size_t)b.n_utt;
  const size_t tiles = (size_t)((b.max_y_len + kSearchTile - 1) / kSearchTile + 1);
  size_t at = 0;
  auto take = [&](size_t bytes) { const size_t o = at; at = (at + (bytes ? bytes : 8) + 255) & ~(size_t)255; return o; };
  const size_t o_idx = take(4 * ny), o_shift = take(8 * ny), o_vuv = take(8 * ny), o_phase = take(8 * ny);
  const size_t o_cnt = take(4 * nu), o_tile = take(4 * nu * tiles), o_off = take(8 * (nu + 1));
  const size_t o_first = take(4 * (ny / kOlaStep + 2 * nu + 4));
  const size_t o_order = take(4 * 2 * nu);
  // the DC remover (GetDCRemover, synthesis.cpp:322-334) depends on fft_size only: once per context
  double* dcr = nullptr;
  for (const auto& e : c.dc_removers)
    if (e.first == F) dcr = e.second;
  if (!dcr) {
    int rc0 = wm_check(dev_alloc(&dcr, sizeof(double) * (size_t)F));
    if (rc0) return rc0;
    hipLaunchKernelGGL(synth_dc_remover_kernel, dim3(1), dim3(64), 0, c.stream, F, dcr);
    rc0 = wm_check(hipStreamSynchronize(c.stream));                // later calls may come on other streams
    if (rc0) { dev_free(dcr); return rc0; }
    c.dc_removers.push_back(std::make_pair(F, dcr));
  }
  unsigned char* base = nullptr;
  int rc = wm_check(dev_alloc(&base, at));
  if (rc) return rc;
  if (!c.h_pulse_info) {            // per context: two pinned, device-visible integers
    rc = wm_check(hipHostMalloc((void**)&c.h_pulse_info, sizeof(int64_t) * 2, hipHostMallocMapped));
    if (!rc) rc = wm_check(hipHostGetDevicePointer((void**)&c.d_pulse_info, c.h_pulse_info, 0));
    if (rc) { dev_free(base); return rc; }
  }
  b.d_syn_arena = base;
  b.d_pulse_idx = (int*)(base + o_idx); b.d_pulse_shift = (double*)(base + o_shift);
  b.d_vuv = (double*)(base + o_vuv); b.d_phase = (double*)(base + o_phase);
  b.d_pulse_cnt = (int*)(base + o_cnt); b.d_pulse_tile_cnt = (int*)(base + o_tile);
  b.d_pulse_off = (int64_t*)(base + o_off); b.d_dc_remover = dcr;
  b.d_pulse_first = (int*)(base + o_first);
  b.d_syn_order = (int*)(base + o_order);
  // [0, n): the identity; [n, 2 n): the utterances by output length, shortest first (stable)
  std::vector<int>& order = b.syn_order_host;                    // the batch's: alive until the copy has been made
  order.resize(2 * nu);
  for (size_t u = 0; u < nu; ++u) order[u] = order[nu + u] = (int)u;
  std::stable_sort(order.begin() + (long)nu, order.end(), [&](int x, int y) { return b.y_len[(size_t)x] < b.y_len[(size_t)y]; });
  b.syn_sorted.assign(order.begin() + (long)nu, order.end());
  // (a copy from pageable memory returns once the source has been read or staged: no wait here -- it was 20 us of
  // every Synthesis() of a new utterance length)
  rc = wm_check(hipMemcpyAsync(b.d_syn_order, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice, c.stream));
  return rc;
}

// The f0-only kernels of a part, up to its pulse numbers: asynchronous on the context's stream.
static int synthesis_prepare_launch(Batch& b, const SynPart& part, const double* d_f0) {
  Context& c = *b.ctx;
  hipStream_t st = c.stream;
  const int F = b.p.fft_size, fs = b.p.fs;
  const double fp = b.p.frame_period / 1000.0;
  const double lowest_f0 = fs / F + 1.0;                  // integer division as in synthesis.cpp:359
  {
    const int tiles = imin(64, (part.max_y_len + 255) / 256);
    {
      TimedScope ts_(b.ctx, "synth_inc_kernel");
      hipLaunchKernelGGL(synth_inc_kernel, dim3(tiles, part.n), dim3(256), 0, st, part.d_list, d_f0, b.d_f_off,
                         b.d_y_off, fs, fp, lowest_f0, b.d_vuv, b.d_phase);
    }
    TimedScope ts_(b.ctx, "synth_timebase_kernel");
    hipLaunchKernelGGL(synth_timebase_kernel, dim3(part.n), dim3(64), 0, st, part.d_list, b.d_y_off, b.d_phase);
  }
  {
    TimedScope ts_(b.ctx, "synth_search_kernel");
    const int tiles_max = (b.max_y_len + kSearchTile - 1) / kSearchTile + 1;        // the row length of the tile counts
    const int tiles_part = (part.max_y_len + kSearchTile - 1) / kSearchTile + 1;
    hipLaunchKernelGGL(synth_pulse_search_kernel<false>, dim3(tiles_part, part.n), dim3(256), 0, st, part.d_list,
                       b.d_y_off, b.d_phase, fs, tiles_max, b.d_pulse_tile_cnt, b.d_pulse_idx, b.d_pulse_shift,
                       b.d_pulse_cnt);
    hipLaunchKernelGGL(synth_pulse_search_kernel<true>, dim3(tiles_part, part.n), dim3(256), 0, st, part.d_list,
                       b.d_y_off, b.d_phase, fs, tiles_max, b.d_pulse_tile_cnt, b.d_pulse_idx, b.d_pulse_shift,
                       b.d_pulse_cnt);
  }
  // Pulse numbers stay on the device; the host needs two numbers only -- the part's total, which sizes the response
  // scratch, and the largest count, which sizes a grid -- and reads them from pinned memory the kernel writes
  // directly.  No hipMemcpy in either direction: a small copy queues on the same DMA engine as whatever bulk
  // transfer another stream has in flight (a 1 GB feature download held this synchronisation, and with it the
  // whole step, for 20 ms).
  hipLaunchKernelGGL(synth_pulse_off_kernel, dim3(1), dim3(256), 0, st, part.d_list, (const int*)b.d_pulse_cnt, part.n,
                     part.p_base, b.d_pulse_off, c.d_pulse_info);
  return wm_check(hipGetLastError());
}

// After the host round trip: the part's totals are known.  `in_flight`: kernels of an earlier part may be using the
// pulse records and the response scratch, which therefore must not move.
static int synthesis_prepare_finish(Batch& b, SynPart& part, int64_t expect_more, bool in_flight) {
  Context& c = *b.ctx;
  hipStream_t st = c.stream;
  const int F = b.p.fft_size;
  part.total_p = c.h_pulse_info[0];
  part.max_np = (int)c.h_pulse_info[1];
  if (part.total_p == 0) return WM_OK;
  // The responses of a piece of the pulse list wait in scratch memory for the overlap-add.  The scratch holds two
  // pieces: while one is added into y on the second stream the pulse kernel fills the other (synthesis_render).
  // A piece is half of the list when that fits, else what half of the scratch cap holds.  Measured (tools/syn_sweep.sh,
  // ms per pass, configs[1] | configs[4]): 1 piece 14.64 | 20.86, 2 pieces 14.64 | 20.73, 4 pieces 14.81 | 20.75,
  // 8 pieces 15.76 | 20.62 -- every launch of the pulse kernel has a tail, so a short list wants few pieces.
  int64_t cap_mb = 4096;
  if (const char* e = getenv("WORLD_MI355_SCRATCH_MB")) cap_mb = atoll(e) > 0 ? atoll(e) : cap_mb;
  int64_t chunk = (cap_mb * 1024 * 1024 / 8) / F / 2;
  if (chunk < 1) chunk = 1;
  int pieces = 2;
  if (const char* e = getenv("WORLD_MI355_SYN_PIECES")) pieces = atoi(e) > 0 ? atoi(e) : pieces;
  const int64_t list = part.total_p > expect_more ? part.total_p : expect_more;     // the longest list still to come
  const int64_t share = (list + pieces - 1) / pieces;
  if (share >= 16384 && chunk > share) chunk = share;       // short lists: one piece, nothing to overlap
  if (chunk > list) chunk = list;
  int rc = c.ensure_side();
  if (rc) return rc;
  if (in_flight) {
    // what is there stays: this part is cut into pieces of the size the scratch was laid out for
    chunk = b.syn_chunk > 0 ? b.syn_chunk : chunk;
  } else {
    // one half only when this part is the whole call AND fits one piece: a later part (expect_more > 0) starts at
    // whatever piece parity the earlier one ended on, so the split path always lays out both halves
    rc = c.ensure_scratch((chunk < list || expect_more > 0 ? 2 : 1) * chunk * F);
    if (rc) return rc;
    b.syn_chunk = chunk;
  }
  const int64_t need = part.p_base + part.total_p + (in_flight ? 0 : expect_more);
  if (need > b.pulse_rec_cap) {
    if (in_flight) {
      // the earlier part's kernels read the records: let them finish (they no longer need theirs afterwards)
      rc = wm_check(hipDeviceSynchronize());
      if (rc) return rc;
    }
    if (b.d_pulse_rec) dev_free(b.d_pulse_rec);
    b.d_pulse_rec = nullptr;
    b.pulse_rec_cap = 0;
    if (b.d_pulse_perm) dev_free(b.d_pulse_perm);
    b.d_pulse_perm = nullptr;
    const int64_t cap = need + need / 8 + 64;
    rc = wm_check(dev_alloc(&b.d_pulse_rec, sizeof(PulseRec) * (size_t)cap));
    if (rc) return rc;
    // perm[cap], then two sets of partition scratch (n_true + block counts), one per part of a split call
    rc = wm_check(dev_alloc(&b.d_pulse_perm, sizeof(int) * (size_t)(cap + 2 * (cap / kPartBlock + 8))));
    if (rc) return rc;
    b.pulse_rec_cap = cap;
  }
  hipLaunchKernelGGL(synth_pulse_rec_kernel, dim3(imin(64, (part.max_np + 255) / 256), part.n), dim3(256), 0, st,
                     part.d_list, b.d_f_off, b.d_y_off, b.d_pulse_off, (const int*)b.d_pulse_cnt, b.d_pulse_idx,
                     b.d_pulse_shift, b.d_vuv, (PulseRec*)b.d_pulse_rec, b.d_pulse_first);
  // The voiced-first order of every piece of the list, here rather than in front of each pulse kernel: two short
  // dependent launches per piece that sat between D4C and the first pulse kernel and between the pieces.  A piece's
  // order lives at its own place of the array (perm + p0); the block counts are scratch of one launch pair.
  // The second part of a split call partitions on another stream than the first: its count scratch is its own.
  const int64_t piece = b.syn_chunk, p_end = part.p_base + part.total_p;
  int* part_scratch = b.d_pulse_perm + b.pulse_rec_cap + (in_flight ? b.pulse_rec_cap / kPartBlock + 8 : 0);
  for (int64_t p0 = part.p_base; p0 < p_end; p0 += piece) {
    const int64_t np = p_end - p0 < piece ? p_end - p0 : piece;
    launch_partition(st, PulseVoicedPred{(const PulseRec*)b.d_pulse_rec + p0}, (int)np, part_scratch + 4,
                     b.d_pulse_perm + p0, part_scratch);
  }
  return wm_check(hipGetLastError());
}

// The pulses [part.p_base, part.p_base + part.total_p): responses by the pulse kernel on the caller's stream, added
// into y on the second stream.  `piece` counts the pieces of the whole call (the halves of the scratch alternate
// across parts).
static int synthesis_render_part(Batch& b, const SynPart& part, const double* d_sp, const double* d_ap, double* d_y,
                                 int& piece) {
  Context& c = *b.ctx;
  hipStream_t st = c.stream;
  const int F = b.p.fft_size, fs = b.p.fs;
  const double fp = b.p.frame_period / 1000.0;
  const int64_t chunk = b.syn_chunk;
  const int ola_tiles = (part.max_y_len + kOlaSeg * kOlaWaves - 1) / (kOlaSeg * kOlaWaves);
  // Piece k: pulse kernel on the caller's stream into half k & 1 of the scratch, overlap-add on the second stream.
  // The overlap-adds run in list order on one stream, so every sample is summed in the order of one piece per launch
  // whatever the timing; the pulse kernel of piece k + 2 waits for the overlap-add of piece k to release its half.
  int rc = WM_OK;
  const int64_t lo = part.p_base, hi = part.p_base + part.total_p;
  for (int64_t p0 = lo; p0 < hi && !rc; p0 += chunk, ++piece) {
    const int64_t p1 = p0 + chunk < hi ? p0 + chunk : hi;
    const int64_t np = p1 - p0;
    const int h = piece & 1;
    double* resp = c.d_scratch + (int64_t)h * chunk * F;
    const int grid = (int)(np < (int64_t)c.frame_grid ? np : (int64_t)c.frame_grid);
    if (piece >= 2) rc = wm_check(hipStreamWaitEvent(st, c.ev_ola[h], 0));
    if (rc) break;
#define WM_SY_CASE(FF, KERNEL)                                                                                  \
  case FF: {                                                                                                    \
    const int per_ = persistent_grid(c, KERNEL<FF>, 64, (int64_t)1 << 40);                               \
    hipLaunchKernelGGL(KERNEL<FF>, dim3(imin(grid, per_)), dim3(64), 0, st, d_sp, d_ap,                         \
                       (const PulseRec*)b.d_pulse_rec, b.d_dc_remover, c.d_rng, fs, fp, p0, p1,                 \
                       (const int*)b.d_pulse_perm + p0, resp);                                                  \
  } break;
    {
      TimedScope ts_(b.ctx, "synth_pulse_kernel");
      // fft 2048 (two waves per SIMD): spectra by pairs in registers (synth_pulse_bp.hpp); WORLD_MI355_PULSE_BP=0: the
      // strided form
      const char* bp_env = getenv("WORLD_MI355_PULSE_BP");          // read per launch: a test switches it within a process
      const bool bp = !(bp_env && atoi(bp_env) == 0);
      switch (F) {
        WM_SY_CASE(512, synth_pulse_kernel)
        WM_SY_CASE(1024, synth_pulse_kernel)
        case 2048:
          if (bp) {
            const int per_ = persistent_grid(c, synth_pulse_bp_kernel<2048>, 64, (int64_t)1 << 40);
            hipLaunchKernelGGL(synth_pulse_bp_kernel<2048>, dim3(imin(grid, per_)), dim3(64), 0, st, d_sp, d_ap,
                               (const PulseRec*)b.d_pulse_rec, b.d_dc_remover, c.d_rng, fs, fp, p0, p1,
                               (const int*)b.d_pulse_perm + p0, resp);
          } else {
            const int per_ = persistent_grid(c, synth_pulse_kernel<2048>, 64, (int64_t)1 << 40);
            hipLaunchKernelGGL(synth_pulse_kernel<2048>, dim3(imin(grid, per_)), dim3(64), 0, st, d_sp, d_ap,
                               (const PulseRec*)b.d_pulse_rec, b.d_dc_remover, c.d_rng, fs, fp, p0, p1,
                               (const int*)b.d_pulse_perm + p0, resp);
          }
          break;
        WM_SY_CASE(4096, synth_pulse_kernel)
      }
    }
#undef WM_SY_CASE
    static const bool overlap = !(getenv("WORLD_MI355_SYN_OVERLAP") && atoi(getenv("WORLD_MI355_SYN_OVERLAP")) == 0);
    hipStream_t so = overlap ? c.side : st;
    rc = wm_check(hipEventRecord(c.ev_pulse[h], st));
    rc = rc ? rc : wm_check(hipStreamWaitEvent(so, c.ev_pulse[h], 0));
    if (rc) break;
    {
      c.stream = so;                                       // the timing bracket records on the context's stream
      TimedScope ts2_(b.ctx, "synth_ola_kernel");
      hipLaunchKernelGGL(synth_ola_kernel, dim3(ola_tiles, part.n), dim3(64 * kOlaWaves), 0, so, part.d_list, b.d_y_off,
                         b.d_pulse_off, (const int*)b.d_pulse_cnt, b.d_pulse_idx, (const int*)b.d_pulse_first, F, p0,
                         p1, resp, d_y);
    }
    c.stream = st;
    rc = wm_check(hipEventRecord(c.ev_ola[h], so));
  }
  c.stream = st;
  return rc ? rc : wm_check(hipGetLastError());
}

// y is complete, and both halves of the scratch are free again, when the last overlap-add is: everything after the
// call on the caller's stream is ordered behind it
static int synthesis_join(Batch& b, int pieces) {
  Context& c = *b.ctx;
  if (pieces == 0) return WM_OK;
  return wm_check(hipStreamWaitEvent(c.stream, c.ev_ola[(pieces - 1) & 1], 0));
}

// ---- the whole batch as one part (launch_analyze_synthesize: prepare on a side stream, render on the main one) ----
// synthesis_begin() queues the f0-only kernels and returns; synthesis_prepare_wait() is the host round trip behind
// them.  Between the two the host is free: the drop-in Synthesis() gathers the caller's `double**` rows of sp / ap
// into pinned memory there, i.e. while the phase chain of the utterance runs (capi.cpp).
int synthesis_begin(Batch& b, const double* d_f0, double* d_y) {
  Context& c = *b.ctx;
  const int F = b.p.fft_size;
  if (F != 512 && F != 1024 && F != 2048 && F != 4096) return WM_ERR_UNSUPPORTED_FFT;
  int rc = c.ensure_rng(b.rng_bound_synthesis());
  rc = rc ? rc : synthesis_arena(b);
  rc = rc ? rc : wm_check(hipMemsetAsync(d_y, 0, sizeof(double) * (size_t)b.total_y, c.stream));
  if (rc) return rc;
  SynPart part{b.d_syn_order, b.n_utt, b.max_y_len, 0};
  return synthesis_prepare_launch(b, part, d_f0);
}
int synthesis_prepare_wait(Batch& b) {
  Context& c = *b.ctx;
  SynPart part{b.d_syn_order, b.n_utt, b.max_y_len, 0};
  int rc = wm_check(hipStreamSynchronize(c.stream));            // the one host round trip of the path
  b.syn_chunk = 0;
  rc = rc ? rc : synthesis_prepare_finish(b, part, 0, false);
  b.syn_total_p = part.total_p;
  return rc;
}
int synthesis_prepare(Batch& b, const double* d_f0, double* d_y) {
  int rc = synthesis_begin(b, d_f0, d_y);
  return rc ? rc : synthesis_prepare_wait(b);
}

int synthesis_render(Batch& b, const double* d_sp, const double* d_ap, double* d_y) {
  if (b.syn_total_p == 0) return WM_OK;
  SynPart part{b.d_syn_order, b.n_utt, b.max_y_len, 0};
  part.total_p = b.syn_total_p;
  int piece = 0;
  int rc = synthesis_render_part(b, part, d_sp, d_ap, d_y, piece);
  return rc ? rc : synthesis_join(b, piece);
}

// ---- Synthesis alone: the batch in two parts (see the top of this section) ----
int launch_synthesis(Batch& b, const double* d_f0, const double* d_sp, const double* d_ap, double* d_y) {
  Context& c = *b.ctx;
  static const int split_env = getenv("WORLD_MI355_SYN_SPLIT") ? atoi(getenv("WORLD_MI355_SYN_SPLIT")) : 3;
  // worth it from a few hundred thousand output samples per part on: below, the parts do not fill the machine
  if (!split_env || b.n_utt < 16 || b.total_y < (int64_t)4 << 20) {
    int rc = synthesis_prepare(b, d_f0, d_y);
    return rc ? rc : synthesis_render(b, d_sp, d_ap, d_y);
  }
  const int F = b.p.fft_size;
  if (F != 512 && F != 1024 && F != 2048 && F != 4096) return WM_ERR_UNSUPPORTED_FFT;
  int rc = c.ensure_rng(b.rng_bound_synthesis());
  rc = rc ? rc : synthesis_arena(b);
  rc = rc ? rc : c.ensure_side();
  if (!rc && !c.prep) {
    // the highest priority there is: its workgroups are few and latency-bound, and they only get the slots the
    // pulse kernel's workgroups leave as they retire
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    rc = wm_check(hipStreamCreateWithPriority(&c.prep, hipStreamNonBlocking, prio_hi));
    rc = rc ? rc : wm_check(hipEventCreateWithFlags(&c.ev_call, hipEventDisableTiming));
    rc = rc ? rc : wm_check(hipEventCreateWithFlags(&c.ev_prep_b, hipEventDisableTiming));
  }
  if (rc) return rc;
  hipStream_t st = c.stream;
  // part A: the shortest utterances up to a sixth of the output samples
  int n_a = 0;
  int64_t acc = 0;
  const int denom = split_env >= 2 ? split_env : 3;
  while (n_a < b.n_utt - 1 && acc < b.total_y / denom) acc += b.y_len[(size_t)b.syn_sorted[(size_t)n_a++]];
  const int* sorted = b.d_syn_order + b.n_utt;
  SynPart pa{sorted, n_a, b.y_len[(size_t)b.syn_sorted[(size_t)n_a - 1]], 0};
  SynPart pb{sorted + n_a, b.n_utt - n_a, b.max_y_len, 0};
  rc = wm_check(hipMemsetAsync(d_y, 0, sizeof(double) * (size_t)b.total_y, st));
  rc = rc ? rc : wm_check(hipEventRecord(c.ev_call, st));          // the caller's f0 is ready from here on
  rc = rc ? rc : synthesis_prepare_launch(b, pa, d_f0);
  rc = rc ? rc : wm_check(hipStreamSynchronize(st));               // host round trip of part A
  if (rc) return rc;
  b.syn_chunk = 0;
  const int64_t guess_b = (int64_t)((double)c.h_pulse_info[0] * (double)(b.total_y - acc) / (double)(acc > 0 ? acc : 1) * 1.25) + 1024;
  rc = synthesis_prepare_finish(b, pa, guess_b, false);
  if (rc) return rc;
  // part B's f0-only kernels on the third stream, beside part A's render stage
  pb.p_base = pa.total_p;
  rc = wm_check(hipStreamWaitEvent(c.prep, c.ev_call, 0));
  c.stream = c.prep;
  rc = rc ? rc : synthesis_prepare_launch(b, pb, d_f0);
  c.stream = st;
  int piece = 0;
  if (!rc && pa.total_p > 0) rc = synthesis_render_part(b, pa, d_sp, d_ap, d_y, piece);
  rc = rc ? rc : wm_check(hipStreamSynchronize(c.prep));           // host round trip of part B (A's render is queued)
  if (rc) return rc;
  c.stream = c.prep;
  rc = synthesis_prepare_finish(b, pb, 0, pa.total_p > 0);
  rc = rc ? rc : wm_check(hipEventRecord(c.ev_prep_b, c.prep));
  c.stream = st;
  rc = rc ? rc : wm_check(hipStreamWaitEvent(st, c.ev_prep_b, 0));
  if (!rc && pb.total_p > 0) rc = synthesis_render_part(b, pb, d_sp, d_ap, d_y, piece);
  b.syn_total_p = pa.total_p + pb.total_p;
  return rc ? rc : synthesis_join(b, piece);
}

#ifdef WM_PHASE
int phase_read_synthesis(unsigned long long* out32) { return wm_phase_read(out32); }
#endif

}  // namespace wm

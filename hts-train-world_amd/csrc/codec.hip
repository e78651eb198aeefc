// codec.hip -- WORLD's feature codec on the device.
//
// Replaces externs/WORLD_v2/src/codec.cpp:
//   codec_code_sp_kernel     CodeSpectralEnvelope :268-295 (CodeOneFrame :122-133, DCTForCodec :73-88)
//   codec_decode_sp_kernel   DecodeSpectralEnvelope :297-324 (DecodeOneFrame :138-157, IDCTForCodec :93-117)
//   codec_code_ap_kernel     CodeAperiodicity :217-235
//   codec_decode_ap_kernel   DecodeAperiodicity :237-266 (CheckVUV :30-41, GetAperiodicity :46-54)
// plus the feature packing of the analysis CLI (test/analysis.cpp:292-366: x 1e4, offsets, log f0,
// float32) as options of the coding kernel, so that the recipe's coded `lf0/mgc/bap` come straight
// out of HBM-resident sp/ap.
//
// Both interp1 calls of the spectral codec run between two FIXED axes (they depend on fs and fft_size
// only), so the knot index and the fraction of every query point are tabulated once per batch on the
// host -- with the same expressions the reference evaluates per frame -- and a frame costs one
// lookup-and-lerp per point.  The DCT is the reference's own construction: even/odd reordering and a
// real FFT of fft_size/2 points (one wavefront, fft.hpp), then the weights; the inverse is a complex
// FFT of fft_size/2 points of which only the real parts are used (fft.cpp's c2c "backward" returns the
// conjugate of the forward transform of its input).
#include <math.h>
#include <string.h>

#include <vector>

#include "batch.hpp"
#include "common.hpp"
#include "fastmath.hpp"
#include "fft.hpp"
#include "window.hpp"

namespace wm {

constexpr double kCodecM0 = 1127.01048;        // constantnumbers.h
constexpr double kCodecF0 = 700.0;
constexpr double kCodecFloorFreq = 40.0;
constexpr double kCodecCeilFreq = 20000.0;
constexpr double kCodecFreqInterval = 3000.0;
constexpr double kCodecUpperLimit = 15000.0;

struct CodeOpts {            // analysis.cpp:296-348 applied around CodeSpectralEnvelope; identity by default
  double pre_scale;          // input multiplied by this before the log
  double zero_value;         // a scaled input of exactly 0 becomes this (0 = leave it)
  double c0_add;             // added to coefficient 0
  int c0_snap;               // 1: coefficient 0 in (0, 1e-4) becomes 0 (analysis.cpp:343-345)
};

template <int F, class OUT>
__global__ __launch_bounds__(64) void codec_code_sp_kernel(const double* __restrict__ in,
                                                           const int* __restrict__ kidx,
                                                           const double* __restrict__ sfrac,
                                                           const cpx* __restrict__ weight, int ndim, CodeOpts o,
                                                           int64_t total_frames, OUT* __restrict__ out) {
  constexpr int MD = F / 2, BINS = F / 2 + 1, N = MD, M = N / 64;
  // the log spectrum and the reordered mel sequence live in the FFT image: they are consumed (into registers)
  // before the transform writes it.  With arrays of their own the kernel held 17.4 KB at fft 1024 (33.8 KB at
  // 2048: one wave per SIMD); now 9.2 / 18.4 KB.
  __shared__ __attribute__((aligned(16))) cpx img[FftLds<N>::kElems];
  static_assert(2 * FftLds<N>::kElems >= (BINS + 1) + MD, "log spectrum and mel sequence fit the image");
  double* ls = reinterpret_cast<double*>(img);
  double* wave = ls + BINS + 1;
  const int lane0 = threadIdx.x;
  FftTw<N> tw;
  tw.init(lane0);
  const double inv_norm = 1.0 / sqrt((double)MD);
  for (int64_t frame = blockIdx.x; frame < total_frames; frame += gridDim.x) {
    const int lane = opaque_lane(lane0);
    const double* row = in + frame * (int64_t)BINS;
    wave_sync();
    {
      // the row and the interpolation table in one trip to memory each (a rolled loop made every bin wait for its own)
      double rv[M + 1];
#pragma unroll
      for (int m = 0; m <= M; ++m) rv[m] = row[imin(lane + 64 * m, BINS - 1)];
#pragma unroll
      for (int m = 0; m <= M; ++m) {
        double v = rv[m] * o.pre_scale;
        if (o.zero_value != 0.0 && v == 0.0) v = o.zero_value;
        const double lv = wm_log(v);
        if (m < M || lane == 0) ls[lane + 64 * m] = lv;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    wave_sync();
    // interp1 onto the mel axis (codec.cpp:126-129) and the even/odd reordering of DCTForCodec (:77-82)
    {
      int kk[M];
      double sf[M];
#pragma unroll
      for (int q = 0; q < M; ++q) {
        kk[q] = kidx[lane + 64 * q];
        sf[q] = sfrac[lane + 64 * q];
      }
#pragma unroll
      for (int q = 0; q < M; ++q) {
        const int m = lane + 64 * q;
        const double y0 = ls[kk[q] - 1];
        const double v = y0 + sf[q] * (ls[kk[q]] - y0);
        wave[(m & 1) ? (MD - 1 - m) / 2 + MD / 2 : m / 2] = v;
      }
    }
    wave_sync();
    // DFT of the MD real points: the reference takes a real FFT; the one-wavefront engine starts at 512
    // complex points, so the sequence goes in as complex with zero imaginary parts (bins 0..MD/2 are the
    // same numbers)
    cpx v[M];
#pragma unroll
    for (int m = 0; m < M; ++m) v[m] = make_double2(wave[lane + 64 * m], 0.0);
    fft_forward<N>(v, img, tw, lane);
    OUT* orow = out + frame * (int64_t)ndim;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int i = lane + 64 * m;
      if (i < ndim) {
        const cpx w = weight[i];
        double c = (v[m].x * w.x - v[m].y * w.y) * inv_norm;
        if (i == 0) {
          c += o.c0_add;
          if (o.c0_snap && c > 0.0 && c < 1e-4) c = 0.0;
        }
        orow[i] = (OUT)c;
      }
    }
  }
}

struct DecodeOpts {          // synth.cpp:198-217 applied around DecodeSpectralEnvelope; identity by default
  double c0_add;             // added to coefficient 0 before decoding
  double post_div;           // decoded values divided by this (0 = leave them)
};

template <int F, class IN>
__global__ __launch_bounds__(64) void codec_decode_sp_kernel(const IN* __restrict__ coded, int ndim,
                                                             const int* __restrict__ kidx,
                                                             const double* __restrict__ sfrac,
                                                             const cpx* __restrict__ weight, DecodeOpts o,
                                                             int64_t total_frames, double* __restrict__ sp) {
  constexpr int MD = F / 2, BINS = F / 2 + 1, N = MD, M = N / 64;
  // the mel knots share the FFT image (the transform's results are in registers when they are written)
  __shared__ __attribute__((aligned(16))) cpx img[FftLds<N>::kElems];
  double* knots = reinterpret_cast<double*>(img);                   // [MD + 2]
  const int lane0 = threadIdx.x;
  FftTw<N> tw;
  tw.init(lane0);
  const double norm = sqrt((double)MD), inv_md = 1.0 / MD;
  for (int64_t frame = blockIdx.x; frame < total_frames; frame += gridDim.x) {
    const int lane = opaque_lane(lane0);
    const IN* crow = coded + frame * (int64_t)ndim;
    cpx v[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {                                   // IDCTForCodec :96-106
      const int i = lane + 64 * m;
      v[m] = make_double2(0.0, 0.0);
      if (i < ndim) {
        double c = (double)crow[i];
        if (i == 0) c += o.c0_add;
        const cpx w = weight[i];
        v[m] = make_double2(c * w.x * norm, -c * w.y * norm);
      }
    }
    fft_forward<N>(v, img, tw, lane);                               // real parts == the wrapper's output
    wave_sync();
    // mel_spectrum[2 i] = out[i], mel_spectrum[2 i + 1] = out[MD - i - 1] (:110-114), padded by its own
    // end values (DecodeOneFrame :147-148)
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int i = lane + 64 * m;                                  // FFT output index
      const int j = i < MD / 2 ? 2 * i : 2 * (MD - 1 - i) + 1;      // mel_spectrum index
      knots[1 + j] = v[m].x;
      if (j == 0) knots[0] = v[m].x;
      if (j == MD - 1) knots[MD + 1] = v[m].x;
    }
    wave_sync();
    double* orow = sp + frame * (int64_t)BINS;
    {
      int kk[M + 1];
      double sf[M + 1];
#pragma unroll
      for (int q = 0; q <= M; ++q) {                                // the table in one trip to memory
        kk[q] = kidx[imin(lane + 64 * q, BINS - 1)];
        sf[q] = sfrac[imin(lane + 64 * q, BINS - 1)];
      }
#pragma unroll
      for (int q = 0; q <= M; ++q) {
        const int b = lane + 64 * q;
        const double y0 = knots[kk[q] - 1];
        const double e = wm_exp((y0 + sf[q] * (knots[kk[q]] - y0)) * inv_md);
        if (q < M || lane == 0) orow[b] = o.post_div != 0.0 ? e / o.post_div : e;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    wave_sync();
  }
}

// coded[frame][k] = interp1Q(0, fs/fft, 20 log10(ap[frame][.]), 3000 (k+1))  (codec.cpp:217-235)
__global__ __launch_bounds__(256) void codec_code_ap_kernel(const double* __restrict__ ap, int bins, int fs,
                                                            int fft_size, int nap, int64_t total_frames,
                                                            double* __restrict__ coded) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total_frames * nap) return;
  const int64_t frame = idx / nap;
  const int k = (int)(idx - frame * nap);
  const double xi = kCodecFreqInterval * (k + 1.0);
  const double q = (xi - 0) / ((double)fs / fft_size);            // interp1Q, matlabfunctions.cpp:220-241
  const int b = (int)q;
  const double frac = q - b;
  const double* row = ap + frame * (int64_t)bins;
  const double y0 = 20 * log10(row[b]);
  const double dy = (b == bins - 1) ? 0.0 : 20 * log10(row[b + 1]) - y0;
  coded[idx] = y0 + dy * frac;
}

__global__ __launch_bounds__(256) void codec_decode_ap_kernel(const double* __restrict__ coded, int nap, int fs,
                                                              int fft_size, int64_t total_frames,
                                                              double* __restrict__ ap) {
  const int bins = fft_size / 2 + 1;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total_frames * bins) return;
  const int64_t frame = idx / bins;
  const int b = (int)(idx - frame * bins);
  const double* c = coded + frame * (int64_t)nap;
  double tmp = 0.0;                                                // CheckVUV :30-41
  for (int k = 0; k < nap; ++k) tmp += c[k];
  tmp /= nap;
  if (tmp > -0.5) { ap[idx] = 1.0 - kSafe; return; }                // keeps InitializeAperiodicity's value
  const double f = (double)fs / fft_size * b;
  const int n = nap + 2;                                           // knots 0, 3000 k, fs/2 (:244-252)
  int cnt = 0;
  for (int j = 0; j < n; ++j) {
    const double xj = j <= nap ? j * kCodecFreqInterval : fs / 2.0;
    cnt += xj <= f ? 1 : 0;
  }
  const int k = cnt < 1 ? 1 : (cnt > n - 1 ? n - 1 : cnt);
  const double x0 = (k - 1) <= nap ? (k - 1) * kCodecFreqInterval : fs / 2.0;
  const double x1 = k <= nap ? k * kCodecFreqInterval : fs / 2.0;
  const double y0 = k - 1 == 0 ? -60.0 : c[k - 2];
  const double y1 = k == nap + 1 ? -kSafe : c[k - 1];
  const double s = (f - x0) / (x1 - x0);
  ap[idx] = wm_exp((y0 + s * (y1 - y0)) * (2.302585092994045684 / 20.0));   // 10^(x/20), codec.cpp:52-53
}

// lf0 = log f0, 0 where unvoiced (analysis.cpp:216-224), as float32
__global__ __launch_bounds__(256) void codec_lf0_kernel(const double* __restrict__ f0, int64_t n,
                                                        float* __restrict__ lf0) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) lf0[i] = (float)(f0[i] != 0 ? log(f0[i]) : 0.0);
}

// The sample formats on either side of the path: wavread's x = s / 2^15 for 16-bit files (test/audioio.cpp:236-249)
// and wavwrite's s = clamp(int(y * 32767)) with the cast truncating towards zero (:160-167).  One pass each over the
// batch's samples (the host pipeline did them as two and four elementwise passes).  A NaN writes 0.
__global__ __launch_bounds__(256) void pcm16_to_samples_kernel(const int16_t* __restrict__ pcm, int64_t n,
                                                               double* __restrict__ x) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    x[i] = (double)pcm[i] * (1.0 / 32768.0);                      // exact: a power of two
}
__global__ __launch_bounds__(256) void samples_to_pcm16_kernel(const double* __restrict__ y, int64_t n,
                                                               int16_t* __restrict__ pcm) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double v = trunc(y[i] * 32767.0);
    pcm[i] = (int16_t)(v != v ? 0.0 : fmin(32767.0, fmax(-32768.0, v)));
  }
}
int launch_pcm16_to_samples(Batch& b, const int16_t* d_pcm, double* d_x) {
  if (b.total_x <= 0) return WM_OK;
  const int64_t blocks = (b.total_x + 1023) / 1024;
  hipLaunchKernelGGL(pcm16_to_samples_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, b.ctx->stream,
                     d_pcm, b.total_x, d_x);
  return wm_check(hipGetLastError());
}
int launch_samples_to_pcm16(Batch& b, const double* d_y, int16_t* d_pcm) {
  if (b.total_y <= 0) return WM_OK;
  const int64_t blocks = (b.total_y + 1023) / 1024;
  hipLaunchKernelGGL(samples_to_pcm16_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, b.ctx->stream,
                     d_y, b.total_y, d_pcm);
  return wm_check(hipGetLastError());
}

// ---- host side -----------------------------------------------------------------------------------
static double to_mel(double f) { return kCodecM0 * log(f / kCodecF0 + 1.0); }         // codec.cpp:59-61
static double from_mel(double m) { return kCodecF0 * (exp(m / kCodecM0) - 1.0); }     // codec.cpp:66-68

// interp1's knot choice and fraction (matlabfunctions.cpp:136-182) for fixed axes
static void interp_table(const std::vector<double>& x, const std::vector<double>& xi, std::vector<int>& k,
                         std::vector<double>& s) {
  const int n = (int)x.size();
  k.resize(xi.size());
  s.resize(xi.size());
  for (size_t i = 0; i < xi.size(); ++i) {
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) / 2;
      if (x[(size_t)mid] <= xi[i]) lo = mid + 1; else hi = mid;
    }
    const int kk = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    k[i] = kk;
    s[i] = (xi[i] - x[(size_t)kk - 1]) / (x[(size_t)kk] - x[(size_t)kk - 1]);
  }
}

struct CodecTables {
  int* d_code_k = nullptr; double* d_code_s = nullptr; cpx* d_code_w = nullptr;
  int* d_dec_k = nullptr; double* d_dec_s = nullptr; cpx* d_dec_w = nullptr;
};

static int codec_setup(Batch& b) {
  if (b.codec_tables) return WM_OK;
  const int fs = b.p.fs, F = b.p.fft_size, md = F / 2, bins = F / 2 + 1;
  const double ceilf = fs / 2.0 < kCodecCeilFreq ? fs / 2.0 : kCodecCeilFreq;
  const double floor_mel = to_mel(kCodecFloorFreq), ceil_mel = to_mel(ceilf);
  CodecTables* T = new CodecTables();
  int rc = WM_OK;
  auto up = [&](void** dst, const void* src, size_t bytes) {
    if (rc) return;
    rc = wm_check(dev_alloc(dst, bytes ? bytes : 8));
    if (!rc && bytes) rc = wm_check(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  };
  {   // coding: frequency axis in mel (knots) -> uniform mel axis (queries), GetParametersForCoding :161-180
    std::vector<double> x((size_t)bins), xi((size_t)md);
    for (int i = 0; i < bins; ++i) x[(size_t)i] = to_mel((double)i * fs / F);   // knot F/2 is never reached
    for (int i = 0; i < md; ++i) xi[(size_t)i] = (ceil_mel - floor_mel) * i / md + floor_mel;
    std::vector<int> k; std::vector<double> s;
    interp_table(x, xi, k, s);
    std::vector<cpx> w((size_t)md);
    for (int i = 0; i < md; ++i)
      w[(size_t)i] = make_double2(2.0 * cos(i * kPi / F) / sqrt((double)F), 2.0 * sin(i * kPi / F) / sqrt((double)F));
    w[0].x /= sqrt(2.0);
    up((void**)&T->d_code_k, k.data(), sizeof(int) * k.size());
    up((void**)&T->d_code_s, s.data(), sizeof(double) * s.size());
    up((void**)&T->d_code_w, w.data(), sizeof(cpx) * w.size());
  }
  {   // decoding: mel axis in Hz padded by 0 and fs/2 (knots) -> uniform Hz axis, GetParametersForDecoding :185-208
    std::vector<double> x((size_t)md + 2), xi((size_t)bins);
    for (int i = 0; i < md; ++i) x[(size_t)i + 1] = from_mel((ceil_mel - floor_mel) * i / md + floor_mel);
    x[0] = 0;
    x[(size_t)md + 1] = fs / 2.0;
    for (int i = 0; i < bins; ++i) xi[(size_t)i] = (double)i * fs / F;
    std::vector<int> k; std::vector<double> s;
    interp_table(x, xi, k, s);
    std::vector<cpx> w((size_t)md);
    for (int i = 0; i < md; ++i)
      w[(size_t)i] = make_double2(cos(i * kPi / F) * sqrt((double)F), sin(i * kPi / F) * sqrt((double)F));
    w[0].x /= sqrt(2.0);
    up((void**)&T->d_dec_k, k.data(), sizeof(int) * k.size());
    up((void**)&T->d_dec_s, s.data(), sizeof(double) * s.size());
    up((void**)&T->d_dec_w, w.data(), sizeof(cpx) * w.size());
  }
  b.codec_tables = T;
  return rc;
}

void codec_free(void* p) {
  CodecTables* T = (CodecTables*)p;
  if (!T) return;
  void* ptrs[] = {T->d_code_k, T->d_code_s, T->d_code_w, T->d_dec_k, T->d_dec_s, T->d_dec_w};
  for (void* q : ptrs)
    if (q) dev_free(q);
  delete T;
}

int codec_num_aperiodicities(int fs) {                              // codec.cpp:212-215
  const double lim = fs / 2.0 - kCodecFreqInterval;
  return (int)((kCodecUpperLimit < lim ? kCodecUpperLimit : lim) / kCodecFreqInterval);
}

template <class OUT>
static int code_sp(Batch& b, const double* d_in, int ndim, CodeOpts o, OUT* d_out, const char* name) {
  const int F = b.p.fft_size;
  if (F != 512 && F != 1024 && F != 2048 && F != 4096) return WM_ERR_UNSUPPORTED_FFT;
  if (ndim < 1 || ndim > F / 4 + 1) return WM_ERR_BAD_ARG;      // the reference reads spectrum[i], i <= fft_size/4
  int rc = codec_setup(b);
  if (rc) return rc;
  const CodecTables& T = *(CodecTables*)b.codec_tables;
  const int64_t tf = b.total_f;
  if (tf <= 0) return WM_OK;
  hipStream_t st = b.ctx->stream;
  TimedScope ts_(b.ctx, name);
#define WM_CODE_CASE(FF)                                                                                     \
  case FF: {                                                                                                 \
    const int per_ = persistent_grid(*b.ctx, codec_code_sp_kernel<FF, OUT>, 64, (int64_t)1 << 40);    \
    hipLaunchKernelGGL((codec_code_sp_kernel<FF, OUT>), dim3((int)(tf < per_ ? tf : per_)), dim3(64), 0, st, \
                       d_in, T.d_code_k, T.d_code_s, T.d_code_w, ndim, o, tf, d_out);                        \
  } break;
  switch (F) {
    WM_CODE_CASE(512)
    WM_CODE_CASE(1024)
    WM_CODE_CASE(2048)
    WM_CODE_CASE(4096)
  }
#undef WM_CODE_CASE
  return wm_check(hipGetLastError());
}

int launch_code_spectral_envelope(Batch& b, const double* d_sp, int ndim, double* d_coded) {
  const CodeOpts o = {1.0, 0.0, 0.0, 0};
  return code_sp<double>(b, d_sp, ndim, o, d_coded, "codec_code_sp_kernel");
}

template <class IN>
static int decode_sp(Batch& b, const IN* d_coded, int ndim, DecodeOpts o, double* d_sp) {
  const int F = b.p.fft_size;
  if (F != 512 && F != 1024 && F != 2048 && F != 4096) return WM_ERR_UNSUPPORTED_FFT;
  if (ndim < 1 || ndim > F / 2) return WM_ERR_BAD_ARG;
  int rc = codec_setup(b);
  if (rc) return rc;
  const CodecTables& T = *(CodecTables*)b.codec_tables;
  const int64_t tf = b.total_f;
  if (tf <= 0) return WM_OK;
  hipStream_t st = b.ctx->stream;
  TimedScope ts_(b.ctx, "codec_decode_sp_kernel");
#define WM_DEC_CASE(FF)                                                                                      \
  case FF: {                                                                                                 \
    const int per_ = persistent_grid(*b.ctx, codec_decode_sp_kernel<FF, IN>, 64, (int64_t)1 << 40);   \
    hipLaunchKernelGGL((codec_decode_sp_kernel<FF, IN>), dim3((int)(tf < per_ ? tf : per_)), dim3(64), 0, st, \
                       d_coded, ndim, T.d_dec_k, T.d_dec_s, T.d_dec_w, o, tf, d_sp);                         \
  } break;
  switch (F) {
    WM_DEC_CASE(512)
    WM_DEC_CASE(1024)
    WM_DEC_CASE(2048)
    WM_DEC_CASE(4096)
  }
#undef WM_DEC_CASE
  return wm_check(hipGetLastError());
}

int launch_decode_spectral_envelope(Batch& b, const double* d_coded, int ndim, double* d_sp) {
  return decode_sp<double>(b, d_coded, ndim, DecodeOpts{0.0, 0.0}, d_sp);
}

int launch_code_aperiodicity(Batch& b, const double* d_ap, double* d_coded) {
  const int nap = codec_num_aperiodicities(b.p.fs);
  const int64_t n = b.total_f * nap;
  if (n <= 0) return WM_OK;
  TimedScope ts_(b.ctx, "codec_code_ap_kernel");
  hipLaunchKernelGGL(codec_code_ap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, b.ctx->stream, d_ap,
                     b.p.fft_size / 2 + 1, b.p.fs, b.p.fft_size, nap, b.total_f, d_coded);
  return wm_check(hipGetLastError());
}

int launch_decode_aperiodicity(Batch& b, const double* d_coded, double* d_ap) {
  const int nap = codec_num_aperiodicities(b.p.fs);
  if (nap < 1) return WM_ERR_UNSUPPORTED;
  const int64_t n = b.total_f * (b.p.fft_size / 2 + 1);
  if (n <= 0) return WM_OK;
  TimedScope ts_(b.ctx, "codec_decode_ap_kernel");
  hipLaunchKernelGGL(codec_decode_ap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, b.ctx->stream, d_coded,
                     nap, b.p.fs, b.p.fft_size, b.total_f, d_ap);
  return wm_check(hipGetLastError());
}

// The coded feature set the recipe's `analysis wav lf0 mgc bap 5 2048 50 25` call writes
// (data/Makefile.in:214; test/analysis.cpp:292-366), as float32, from resident f0 / sp / ap:
//   mgc = CodeSpectralEnvelope(sp * 1e4, zeros -> 1e-4)[spec_dim], coefficient 0 + 12.0
//   bap = CodeSpectralEnvelope(ap * 1e4)[ap_dim], coefficient 0 - 9.210340, (0, 1e-4) -> 0
//         (the mcep result computed just before is overwritten by the reference, :332-341)
//   lf0 = log f0, 0 where unvoiced
int launch_recipe_features(Batch& b, const double* d_f0, const double* d_sp, const double* d_ap, int spec_dim,
                           int ap_dim, float* d_lf0, float* d_mgc, float* d_bap) {
  const CodeOpts osp = {1e4, 0.0001, 12.0, 0};
  int rc = code_sp<float>(b, d_sp, spec_dim, osp, d_mgc, "codec_code_sp_kernel");
  if (rc) return rc;
  const CodeOpts oap = {1e4, 0.0, -9.210340, 1};
  rc = code_sp<float>(b, d_ap, ap_dim, oap, d_bap, "codec_code_sp_kernel");
  if (rc) return rc;
  if (b.total_f > 0)
    hipLaunchKernelGGL(codec_lf0_kernel, dim3((unsigned)((b.total_f + 255) / 256)), dim3(256), 0, b.ctx->stream, d_f0,
                       b.total_f, d_lf0);
  return wm_check(hipGetLastError());
}

// ---- decode side of the synth CLI's coded features (test/synth.cpp:151-256) ------------------------
// bap rows go through the CLI's SPTK port (test/sptkfunctions.cpp): mgc2sp = freqt (:596-631) from
// `order` to F/2 coefficients with a = -0.55, c0 -> log(exp(c0)) (gnorm / ignorm with gamma 0, :331-365),
// then the real part of the F-point DFT of the zero-padded cepstrum (c2sp :256-274); the CLI keeps
// exp(x[j]) / 1e4 for j < order only (synth.cpp:242-245) and leaves the other bins of the row
// uninitialised -- they are written as 0.0 here (flagged in INTEGRATION.md).
//
// freqt is a two-dimensional recurrence, g_s[j] = g_{s-1}[j-1] + a (g_{s-1}[j] - g_s[j-1]), sequential in
// both the input step s and the output index j.  It runs as a systolic pipeline: lane s owns input step s
// and computes its row along j, one element per time step, reading the element lane s-1 produced in the
// previous time step with a one-lane DPP shift.  Every element is computed by the reference's expression in
// the reference's order (no FMA contraction), so the cepstrum is bit-identical.  W lanes serve a frame: with
// order < 32 a wavefront carries two frames side by side.  The loop is issue-bound, so it is split into the
// start-up steps, where some row is still at one of its two special first elements, and a steady phase of
// ten instructions per step without selects or exec-mask branches.
// The spectrum is one real FFT of the zero-padded cepstrum per frame on the whole wavefront (fft.hpp).
template <int W, int F>
__global__ __launch_bounds__(64) void codec_bap_decode_kernel(const float* __restrict__ bap, int ap_dim, int order,
                                                              double alpha, int64_t total_frames,
                                                              double* __restrict__ ap) {
#pragma clang fp contract(off)
  constexpr int FP = 64 / W;       // frames per wavefront
  constexpr int N = F / 2, M = N / 64, h = F / 2, bins = h + 1;
  constexpr int kC = h + 2 + 64;   // one cepstrum plus 64 dummy slots (see the store in the steady loop)
  __shared__ __attribute__((aligned(16))) double bd_lds[2 * FftLds<N>::kElems + FP * kC];
  cpx* img = reinterpret_cast<cpx*>(bd_lds);             // FFT image / spectrum
  const int lane = threadIdx.x, sub = lane / W, s = lane % W;
  double* c_all = bd_lds + 2 * FftLds<N>::kElems;
  double* c = c_all + sub * kC;                          // [h + 2] transformed cepstrum of this lane's frame
  FftTw<N> tw;
  tw.init(lane);
  const double a = (0.0 - alpha) / (1 - alpha * 0.0);          // mgc2mgc :241 with a2 = 0
  const double b = 1 - a * a;
  for (int64_t f0_ = (int64_t)blockIdx.x * FP; f0_ < total_frames; f0_ += (int64_t)gridDim.x * FP) {
    const int64_t frame = f0_ + sub;
    const bool live = frame < total_frames;
    wave_sync();
    // lane s feeds coefficient order - s (the reference walks c1[order] .. c1[0]); the row holds ap_dim
    // values, mgc2sp reads order + 1 (one past an even ap_dim: taken as 0)
    const int ci = order - s;
    double cin = 0.0;
    if (live && ci >= 0 && ci < ap_dim) cin = (double)bap[frame * (int64_t)ap_dim + ci];
    if (ci == 0) cin += 9.210340;                              // synth.cpp:241
    double last = 0.0, prev_up = 0.0, own_prev = 0.0;
    const bool store_lane = live && s == order;
    // start-up: until step order + 1 some row is still at its first (j = 0) or second (j = 1) element, which
    // have their own expressions (:620-623)
    for (int t = 0; t <= order + 1; ++t) {
      double up = dpp_get<0x138, 0xf, 0xf>(last);              // wave_shr:1 -- lane s reads lane s-1, lane 0 reads 0
      if (W < 64 && s == 0) up = 0.0;                          // g_{-1} = 0 between two packed frames too
      const int j = t - s;
      const double B = up - (j >= 2 ? own_prev : 0.0);
      const double A = j == 0 ? cin : prev_up * (j == 1 ? b : 1.0);
      const double val = A + a * B;
      // a row that has not started (j < 0) computes values nobody reads: its right neighbour is one step
      // behind it, and j = 0 takes nothing from the row's own state
      prev_up = up;
      own_prev = val;
      last = val;
      if (store_lane && j >= 0) c[j] = val;
    }
    // steady state: every row is at j >= 2, the general element g[j] = d[j-1] + a (d[j] - g[j-1]) (:624-625).
    // A row that is finished (j > h) again computes values nobody reads.  Every lane stores every step -- the
    // row that carries the result into the cepstrum, the others into a slot of their own -- so the loop has
    // no exec-mask branch.
    {
      double* dst = store_lane ? c + 2 : c_all + sub * kC + (h + 2) + s;
      const int adv = store_lane ? 1 : 0;
      for (int t = order + 2; t <= h + order; ++t) {
        double up = dpp_get<0x138, 0xf, 0xf>(last);
        if (W < 64 && s == 0) up = 0.0;
        const double val = prev_up + a * (up - own_prev);
        prev_up = up;
        own_prev = val;
        last = val;
        *dst = val;
        dst += adv;
      }
    }
    wave_sync();
    if (live && s == 0) c[0] = log(exp(c[0]));
    wave_sync();
    // c2sp (:256-274): the real part of the F-point transform of the cepstrum zero-padded to F, wanted at the
    // first `order` bins; one real transform per frame on the whole wavefront
    double mine = 0.0;                                         // bin s of this lane's frame
#pragma unroll 1
    for (int q = 0; q < FP; ++q) {
      const double* cq = c_all + q * kC;
      cpx v[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const int i0 = 2 * (lane + 64 * m);
        v[m] = make_double2(i0 <= h ? cq[i0] : 0.0, i0 + 1 <= h ? cq[i0 + 1] : 0.0);
      }
      rfft_forward<N>(v, img, img, tw, lane);
      if (sub == q && s < order) mine = img[s].x;
      wave_sync();
    }
    double* row = ap + frame * (int64_t)bins;
    if (live && s < order) row[s] = exp(mine) / 1e4;             // synth.cpp:243-245
    if (live)
      for (int j = order + s; j < bins; j += W) row[j] = 0.0;
  }
}

__global__ __launch_bounds__(256) void codec_f0_from_lf0_kernel(const float* __restrict__ lf0, int64_t n,
                                                                double* __restrict__ f0) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double l = (double)lf0[i];
  f0[i] = l != 0 ? exp(l) : 0;                                 // ToF0, synth.cpp:80-88
}

int launch_recipe_decode(Batch& b, const float* d_lf0, const float* d_mgc, const float* d_bap, int spec_dim,
                         int ap_dim, double* d_f0, double* d_sp, double* d_ap) {
  const int F = b.p.fft_size;
  if (F != 512 && F != 1024 && F != 2048 && F != 4096) return WM_ERR_UNSUPPORTED_FFT;
  const int order = (ap_dim % 2 == 1) ? ap_dim - 1 : ap_dim;   // synth.cpp:233-235
  if (ap_dim < 1 || order < 1 || order > 63) return WM_ERR_BAD_ARG;
  int rc = decode_sp<float>(b, d_mgc, spec_dim, DecodeOpts{-12.0, 1e4}, d_sp);
  if (rc) return rc;
  const int64_t tf = b.total_f;
  if (tf <= 0) return WM_OK;
  hipStream_t st = b.ctx->stream;
  {
    TimedScope ts_(b.ctx, "codec_bap_decode_kernel");
    const int64_t cap = (int64_t)b.ctx->num_cu * 16;
    const int64_t units = order < 32 ? (tf + 1) / 2 : tf;
    const dim3 grid((unsigned)(units < cap ? units : cap));
#define WM_BAP_CASE(WW, FF)                                                                                   \
  hipLaunchKernelGGL((codec_bap_decode_kernel<WW, FF>), grid, dim3(64), 0, st, d_bap, ap_dim, order, 0.55, tf, d_ap)
    if (order < 32) {
      switch (F) {
        case 512: WM_BAP_CASE(32, 512); break;
        case 1024: WM_BAP_CASE(32, 1024); break;
        case 2048: WM_BAP_CASE(32, 2048); break;
        default: WM_BAP_CASE(32, 4096); break;
      }
    } else {
      switch (F) {
        case 512: WM_BAP_CASE(64, 512); break;
        case 1024: WM_BAP_CASE(64, 1024); break;
        case 2048: WM_BAP_CASE(64, 2048); break;
        default: WM_BAP_CASE(64, 4096); break;
      }
    }
#undef WM_BAP_CASE
  }
  hipLaunchKernelGGL(codec_f0_from_lf0_kernel, dim3((unsigned)((tf + 255) / 256)), dim3(256), 0, st, d_lf0, tf, d_f0);
  return wm_check(hipGetLastError());
}

}  // namespace wm

// ---- dynamic-feature windows and `cmp` composition -------------------------
// Replaces data/scripts/window.pl:45-146 for every stream and the SPTK `merge` chain of
// data/Makefile.in:296-299 in one pass: out[frame] = [stream 0: win 1 | win 2 | ... ][stream 1: ...]...
// Frames outside an utterance are clamped to its first / last frame; a frame whose checked taps touch
// the ignore value -1e10 (unvoiced lf0) yields -1e10.  Accumulation in double, in tap order, no FMA
// contraction: the float32 results are bit-identical to the script's.
namespace wm {

constexpr int kCmpMaxStreams = 4, kCmpMaxWin = 4, kCmpMaxTaps = 15;
struct CmpMeta {
  int n_streams, total_cols;
  int dim[kCmpMaxStreams], nwin[kCmpMaxStreams], col0[kCmpMaxStreams];
  int wsize[kCmpMaxStreams][kCmpMaxWin];
  unsigned chk[kCmpMaxStreams][kCmpMaxWin];                 // bit k: tap k takes part in the boundary check
  double w[kCmpMaxStreams][kCmpMaxWin][kCmpMaxTaps];
  const float* data[kCmpMaxStreams];
};

__global__ __launch_bounds__(256) void cmp_compose_kernel(CmpMeta m, const int* __restrict__ frame_utt,
                                                          const int64_t* __restrict__ f_off, int64_t total_frames,
                                                          float* __restrict__ out) {
#pragma clang fp contract(off)
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total_frames * m.total_cols) return;
  const int64_t frame = idx / m.total_cols;
  const int col = (int)(idx - frame * m.total_cols);
  int s = 0;
#pragma unroll
  for (int q = 1; q < kCmpMaxStreams; ++q)
    if (q < m.n_streams && col >= m.col0[q]) s = q;
  const int dim = m.dim[s];
  const int c = col - m.col0[s];
  const int wi = c / dim, j = c - wi * dim;
  const int size = m.wsize[s][wi], nlr = (size - 1) / 2;
  const int u = frame_utt[frame];
  const int64_t lo = f_off[u], hi = f_off[u + 1] - 1;
  const float* src = m.data[s];
  bool boundary = false;
  double acc = 0.0;
  for (int k = 0; k < size; ++k) {
    int64_t l = frame + (k - nlr);
    l = l < lo ? lo : (l > hi ? hi : l);
    const double v = (double)src[l * dim + j];
    if (((m.chk[s][wi] >> k) & 1u) && v == -1.0e+10) boundary = true;
    acc += m.w[s][wi][k] * v;
  }
  out[idx] = boundary ? -1.0e+10f : (float)acc;
}

int launch_compose_cmp(Batch& b, int n_streams, const float* const* d_data, const int* dims, const int* n_windows,
                       const double* const* const* windows, const int* const* window_sizes, float* d_out) {
  if (n_streams < 1 || n_streams > kCmpMaxStreams) return WM_ERR_BAD_ARG;
  CmpMeta m;
  memset(&m, 0, sizeof(m));
  m.n_streams = n_streams;
  int col = 0;
  for (int s = 0; s < n_streams; ++s) {
    if (dims[s] < 1 || n_windows[s] < 1 || n_windows[s] > kCmpMaxWin) return WM_ERR_BAD_ARG;
    m.dim[s] = dims[s];
    m.nwin[s] = n_windows[s];
    m.col0[s] = col;
    m.data[s] = d_data[s];
    for (int i = 0; i < n_windows[s]; ++i) {
      const int size = window_sizes[s][i];
      if (size < 1 || size > kCmpMaxTaps || size % 2 != 1) return WM_ERR_BAD_ARG;   // window.pl:96-98
      m.wsize[s][i] = size;
      unsigned chk = (1u << size) - 1u;                           // window.pl:83-94: leading / trailing zero taps
      for (int k = 0; k < size; ++k) { if (windows[s][i][k] != 0.0) break; chk &= ~(1u << k); }
      for (int k = size - 1; k >= 0; --k) { if (windows[s][i][k] != 0.0) break; chk &= ~(1u << k); }
      m.chk[s][i] = chk;
      for (int k = 0; k < size; ++k) m.w[s][i][k] = windows[s][i][k];
    }
    col += dims[s] * n_windows[s];
  }
  m.total_cols = col;
  const int64_t n = b.total_f * col;
  if (n <= 0) return WM_OK;
  TimedScope ts_(b.ctx, "cmp_compose_kernel");
  hipLaunchKernelGGL(cmp_compose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, b.ctx->stream, m,
                     b.d_frame_utt, b.d_f_off, b.total_f, d_out);
  return wm_check(hipGetLastError());
}

}  // namespace wm

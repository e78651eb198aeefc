// capi.cpp -- the C-ABI of libworld_mi355.so.
//
// (1) WORLD's public API (include/world/*.h), one utterance per call, host
//     pointers in and out, exactly the reference's signatures
//     (externs/WORLD_v2/src/world/{dio,stonemask,cheaptrick,d4c,synthesis,harvest}.h).
//     Each call stages the utterance into HBM, runs the batch kernels with B = 1
//     and copies the result back.  There is NO CPU fallback: without a HIP device
//     the call aborts with a message, like the reference aborts on bad_alloc.
// (2) The batched device-pointer extension (include/world_mi355.h).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <mutex>
#include <vector>

#include "batch.hpp"
#include "common.hpp"
#include "world/cheaptrick.h"
#include "world/codec.h"
#include "world/d4c.h"
#include "world/dio.h"
#include "world/harvest.h"
#include "world/stonemask.h"
#include "world/synthesis.h"

namespace wm {
const char* last_error();
void set_error(const char* msg);
int launch_harvest(Batch& b, const double* d_x, double* d_t, double* d_f0);
void free_batch_buffers(Batch& b);
#ifdef WM_PHASE
int phase_read_d4c(unsigned long long* out32);
int phase_read_cheaptrick(unsigned long long* out32);
int phase_read_synthesis(unsigned long long* out32);
#endif
}  // namespace wm

using namespace wm;

// the drop-in entry points' failure handler (WorldMi355SetErrorHandler; see die() below)
// The handler and its argument are read and written as a pair under their own lock (not g_mu: die() runs while an
// entry point holds that one).
static std::mutex g_err_mu;
static WorldMi355ErrorHandler g_on_error = nullptr;
static void* g_on_error_user = nullptr;
static WorldMi355ErrorHandler error_handler(void** user) {
  std::lock_guard<std::mutex> lock(g_err_mu);
  *user = g_on_error_user;
  return g_on_error;
}

struct WorldMi355Context { Context c; };
struct WorldMi355Batch { Batch b; };

template <typename T> static int alloc_items(T** p, size_t count) {
  *p = nullptr;
  return wm_check(wm::dev_alloc(p, sizeof(T) * (count ? count : 1)));
}
template <typename T> static int upload(T** p, const std::vector<T>& v) {
  int rc = alloc_items(p, v.size());
  if (rc) return rc;
  if (v.empty()) return WM_OK;
  return wm_check(hipMemcpy(*p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
}

extern "C" {

const char* WorldMi355LastError(void) { return wm::last_error(); }

void WorldMi355DefaultParams(int fs, double frame_period, WorldMi355Params* p) {
  memset(p, 0, sizeof(*p));
  p->fs = fs;
  p->frame_period = frame_period;
  p->f0_floor = 71.0;            // analysis.cpp:111
  p->f0_ceil = 800.0;            // dio.cpp:652 (kCeilF0)
  p->channels_in_octave = 2.0;   // dio.cpp:651
  p->speed = 1;                  // analysis.cpp:106
  p->allowed_range = 0.1;        // analysis.cpp:116
  p->q1 = -0.15;                 // analysis.cpp:151
  p->fft_size = 0;
  p->d4c_threshold = 0.0;        // analysis.cpp:190
}

int WorldMi355CreateContext(int device, void* hip_stream, WorldMi355Context** out) {
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    wm::set_error("no HIP device visible: libworld_mi355 has no CPU path");
    return WM_ERR_NO_DEVICE;
  }
  int before = -1;
  (void)hipGetDevice(&before);
  if (device >= 0 && device != before) {
    int rc = wm_check(hipSetDevice(device));
    if (rc) return rc;
  }
  WorldMi355Context* h = new WorldMi355Context();
  Context& c = h->c;
  hipGetDevice(&c.device);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, c.device) == hipSuccess) c.num_cu = prop.multiProcessorCount;
  c.frame_grid = c.num_cu * 16 * c.oversub;
  if (const char* e = getenv("WORLD_MI355_OVERSUB")) {          // workgroups per resident slot (batch.hpp)
    const int v = atoi(e);
    if (v >= 1 && v <= 64) {
      c.oversub = v;
      c.frame_grid = c.num_cu * 16 * v;
    }
  }
  // NULL selects the legacy default stream (stream 0): it orders with every blocking stream, which
  // is what callers that allocate and copy with plain hipMemcpy / torch's default stream expect.
  c.stream = (hipStream_t)hip_stream;
  c.own_stream = false;
  *out = h;
  // every entry point switches to its context's device itself (OnDevice): the caller's current device stays
  if (before >= 0 && before != c.device) (void)hipSetDevice(before);
  return WM_OK;
}

void WorldMi355DestroyContext(WorldMi355Context* h) {
  if (!h) return;
  Context& c = h->c;
  OnDevice dev_(c);
  hipStreamSynchronize(c.stream);
  c.timing_clear();
  if (c.d_rng) wm::dev_free(c.d_rng);
  if (c.d_scratch) wm::dev_free(c.d_scratch);
  for (auto& f : c.dio_filters) {
    if (f.d_lowcut) wm::dev_free(f.d_lowcut);
    if (f.d_win) wm::dev_free(f.d_win);
    if (f.d_H) wm::dev_free(f.d_H);
  }
  c.dio_filters.clear();
  for (auto& w : c.nuttall_windows)
    if (w.second) wm::dev_free(w.second);
  c.nuttall_windows.clear();
  for (auto& w : c.dc_removers)
    if (w.second) wm::dev_free(w.second);
  c.dc_removers.clear();
  if (c.d_sm_twid) wm::dev_free(c.d_sm_twid);
  if (c.h_pulse_info) hipHostFree(c.h_pulse_info);
  if (c.own_stream) hipStreamDestroy(c.stream);
  if (c.side) { hipStreamSynchronize(c.side); hipStreamDestroy(c.side); }
  if (c.ev_f0) hipEventDestroy(c.ev_f0);
  if (c.ev_prep) hipEventDestroy(c.ev_prep);
  if (c.ev_d4c) hipEventDestroy(c.ev_d4c);
  if (c.ev_rare) hipEventDestroy(c.ev_rare);
  if (c.aux) { hipStreamSynchronize(c.aux); hipStreamDestroy(c.aux); }
  if (c.prep) { hipStreamSynchronize(c.prep); hipStreamDestroy(c.prep); }
  if (c.ev_call) hipEventDestroy(c.ev_call);
  if (c.ev_prep_b) hipEventDestroy(c.ev_prep_b);
  for (int h = 0; h < 2; ++h) {
    if (c.ev_pulse[h]) hipEventDestroy(c.ev_pulse[h]);
    if (c.ev_ola[h]) hipEventDestroy(c.ev_ola[h]);
  }
  delete h;
}

int WorldMi355SetStream(WorldMi355Context* h, void* hip_stream) {
  OnDevice dev_(h->c);
  Context& c = h->c;
  int rc = wm_check(hipStreamSynchronize(c.stream));
  if (rc) return rc;
  if (c.own_stream) hipStreamDestroy(c.stream);
  c.own_stream = false;
  c.stream = (hipStream_t)hip_stream;
  return WM_OK;
}

int WorldMi355Synchronize(WorldMi355Context* h) {
  OnDevice dev_(h->c);
  return wm_check(hipStreamSynchronize(h->c.stream));
}

int WorldMi355CreateBatch(WorldMi355Context* h, const WorldMi355Params* params, int n_utt,
                          const int* x_lengths, const int* f0_lengths, const int* y_lengths,
                          WorldMi355Batch** out) {
  *out = nullptr;
  if (!h || !params || n_utt <= 0 || (!x_lengths && !f0_lengths)) {
    wm::set_error("CreateBatch: bad argument");
    return WM_ERR_BAD_ARG;
  }
  OnDevice dev_(h->c);
  WorldMi355Batch* hb = new WorldMi355Batch();
  Batch& b = hb->b;
  b.ctx = &h->c;
  b.p = *params;
  if (b.p.fft_size == 0)   // GetFFTSizeForCheapTrick, cheaptrick.cpp:191-194
    b.p.fft_size = (int)pow(2.0, 1.0 + (int)(log(3.0 * b.p.fs / b.p.f0_floor + 1) / kLog2));
  b.n_utt = n_utt;
  b.x_len.assign(n_utt, 0); b.f0_len.assign(n_utt, 0); b.y_len.assign(n_utt, 0);
  b.x_off.assign(n_utt + 1, 0); b.f_off.assign(n_utt + 1, 0); b.y_off.assign(n_utt + 1, 0);
  for (int u = 0; u < n_utt; ++u) {
    b.x_len[u] = x_lengths ? x_lengths[u] : 0;
    b.f0_len[u] = f0_lengths ? f0_lengths[u]
                             : (int)(1000.0 * b.x_len[u] / b.p.fs / b.p.frame_period) + 1;   // dio.cpp:638-640
    b.y_len[u] = y_lengths ? y_lengths[u]
                           : (int)((b.f0_len[u] - 1) * b.p.frame_period / 1000.0 * b.p.fs) + 1;   // synth.cpp:259
    if (b.x_len[u] < 0 || b.f0_len[u] <= 0 || b.y_len[u] < 0) {
      delete hb;
      wm::set_error("CreateBatch: negative length");
      return WM_ERR_BAD_ARG;
    }
    if (x_lengths && b.x_len[u] == 0) {
      // the reference's caller refuses a wav without samples (test/analysis.cpp:252-259); the kernels clamp their
      // loads into [0, x_length - 1], which an empty utterance does not have
      delete hb;
      wm::set_error("CreateBatch: an utterance has no samples");
      return WM_ERR_BAD_ARG;
    }
    b.x_off[u + 1] = b.x_off[u] + b.x_len[u];
    b.f_off[u + 1] = b.f_off[u] + b.f0_len[u];
    b.y_off[u + 1] = b.y_off[u] + b.y_len[u];
    b.max_x_len = imax(b.max_x_len, b.x_len[u]);
    b.max_f0_len = imax(b.max_f0_len, b.f0_len[u]);
    b.max_y_len = imax(b.max_y_len, b.y_len[u]);
  }
  b.total_x = b.x_off[n_utt]; b.total_f = b.f_off[n_utt]; b.total_y = b.y_off[n_utt];
  // One device allocation and one upload for every descriptor and per-frame work array of the batch (a batch of
  // one utterance is created per call by the per-utterance entry points: fourteen hipMallocs and seven blocking
  // copies were most of its cost).  Layout: 256-byte aligned sections; the first `init_bytes` are initialised
  // from a host image, the rest is scratch.
  const size_t n1 = (size_t)n_utt + 1, nf = (size_t)b.total_f;
  size_t at = 0;
  auto take = [&](size_t bytes) { const size_t o = at; at = (at + (bytes ? bytes : 8) + 255) & ~(size_t)255; return o; };
  const size_t o_xoff = take(8 * n1), o_foff = take(8 * n1), o_yoff = take(8 * n1);
  const size_t o_xlen = take(4 * (size_t)n_utt), o_flen = take(4 * (size_t)n_utt), o_ylen = take(4 * (size_t)n_utt);
  const size_t o_futt = take(4 * nf);
  const size_t init_bytes = at;
  const size_t o_rng = take(4 * nf), o_rng2 = take(4 * nf), o_ap0 = take(8 * nf), o_f0t = take(8 * nf);
  const size_t o_perm = take(4 * nf), o_pcnt = take(4 * (nf / 1024 + 2)), o_pn = take(4 * 4);
  const size_t o_rng_d = take(4 * nf), o_perm_d = take(4 * nf), o_pcnt_d = take(4 * (nf / 1024 + 2)), o_pn_d = take(4 * 4);
  std::vector<unsigned char> img(init_bytes, 0);
  memcpy(&img[o_xoff], b.x_off.data(), 8 * n1);
  memcpy(&img[o_foff], b.f_off.data(), 8 * n1);
  memcpy(&img[o_yoff], b.y_off.data(), 8 * n1);
  memcpy(&img[o_xlen], b.x_len.data(), 4 * (size_t)n_utt);
  memcpy(&img[o_flen], b.f0_len.data(), 4 * (size_t)n_utt);
  memcpy(&img[o_ylen], b.y_len.data(), 4 * (size_t)n_utt);
  {
    int* fu = reinterpret_cast<int*>(&img[o_futt]);
    for (int u = 0; u < n_utt; ++u)
      for (int64_t i = b.f_off[u]; i < b.f_off[u + 1]; ++i) fu[i] = u;
  }
  unsigned char* base = nullptr;
  int rc = wm_check(wm::dev_alloc(&base, at));
  if (!rc) rc = wm_check(hipMemcpy(base, img.data(), init_bytes, hipMemcpyHostToDevice));
  if (rc) {
    if (base) wm::dev_free(base);
    delete hb;
    return rc;
  }
  b.d_arena = base;
  b.d_x_off = (int64_t*)(base + o_xoff); b.d_f_off = (int64_t*)(base + o_foff); b.d_y_off = (int64_t*)(base + o_yoff);
  b.d_x_len = (int*)(base + o_xlen); b.d_f0_len = (int*)(base + o_flen); b.d_y_len = (int*)(base + o_ylen);
  b.d_frame_utt = (int*)(base + o_futt);
  b.d_rng_off = (int*)(base + o_rng); b.d_rng_off2 = (int*)(base + o_rng2);
  b.d_ap0 = (double*)(base + o_ap0); b.d_f0_tmp = (double*)(base + o_f0t);
  b.d_perm = (int*)(base + o_perm); b.d_part_cnt = (int*)(base + o_pcnt); b.d_part_n = (int*)(base + o_pn);
  b.d_rng_off_d4c = (int*)(base + o_rng_d); b.d_perm_d4c = (int*)(base + o_perm_d);
  b.d_part_cnt_d4c = (int*)(base + o_pcnt_d); b.d_part_n_d4c = (int*)(base + o_pn_d);
  *out = hb;
  return WM_OK;
}

void WorldMi355DestroyBatch(WorldMi355Batch* hb) {
  if (!hb) return;
  Batch& b = hb->b;
  OnDevice dev_(*b.ctx);
  hipStreamSynchronize(b.ctx->stream);
  wm::free_batch_buffers(b);
  delete hb;
}

int64_t WorldMi355BatchTotalSamples(const WorldMi355Batch* b) { return b->b.total_x; }
int64_t WorldMi355BatchTotalFrames(const WorldMi355Batch* b) { return b->b.total_f; }
int64_t WorldMi355BatchTotalOutputSamples(const WorldMi355Batch* b) { return b->b.total_y; }
int WorldMi355BatchFftSize(const WorldMi355Batch* b) { return b->b.p.fft_size; }
const int64_t* WorldMi355BatchSampleOffsets(const WorldMi355Batch* b) { return b->b.x_off.data(); }
const int64_t* WorldMi355BatchFrameOffsets(const WorldMi355Batch* b) { return b->b.f_off.data(); }
const int64_t* WorldMi355BatchOutputOffsets(const WorldMi355Batch* b) { return b->b.y_off.data(); }

int WorldMi355Dio(WorldMi355Batch* b, const double* x, double* t, double* f0) {
  OnDevice dev_(b->b.ctx[0]);
  return launch_dio(b->b, x, t, f0);
}
int WorldMi355StoneMask(WorldMi355Batch* b, const double* x, const double* t, const double* f0,
                        double* refined_f0) {
  OnDevice dev_(b->b.ctx[0]);
  return launch_stonemask(b->b, x, t, f0, refined_f0);
}
int WorldMi355CheapTrick(WorldMi355Batch* b, const double* x, const double* t, const double* f0,
                         double* sp) {
  OnDevice dev_(b->b.ctx[0]);
  return launch_cheaptrick(b->b, x, t, f0, sp);
}
int WorldMi355D4C(WorldMi355Batch* b, const double* x, const double* t, const double* f0, double* ap) {
  OnDevice dev_(b->b.ctx[0]);
  return launch_d4c(b->b, x, t, f0, ap);
}
int WorldMi355Synthesis(WorldMi355Batch* b, const double* f0, const double* sp, const double* ap,
                        double* y) {
  OnDevice dev_(b->b.ctx[0]);
  return launch_synthesis(b->b, f0, sp, ap, y);
}
int WorldMi355GetNumberOfAperiodicities(int fs) { return codec_num_aperiodicities(fs); }
int WorldMi355CodeSpectralEnvelope(WorldMi355Batch* b, const double* sp, int number_of_dimensions, double* coded) {
  OnDevice dev_(b->b.ctx[0]);
  return launch_code_spectral_envelope(b->b, sp, number_of_dimensions, coded);
}
int WorldMi355DecodeSpectralEnvelope(WorldMi355Batch* b, const double* coded, int number_of_dimensions,
                                     double* sp) {
  OnDevice dev_(b->b.ctx[0]);
  return launch_decode_spectral_envelope(b->b, coded, number_of_dimensions, sp);
}
int WorldMi355CodeAperiodicity(WorldMi355Batch* b, const double* ap, double* coded) {
  OnDevice dev_(b->b.ctx[0]);
  return launch_code_aperiodicity(b->b, ap, coded);
}
int WorldMi355DecodeAperiodicity(WorldMi355Batch* b, const double* coded, double* ap) {
  OnDevice dev_(b->b.ctx[0]);
  return launch_decode_aperiodicity(b->b, coded, ap);
}
int WorldMi355RecipeFeatures(WorldMi355Batch* b, const double* f0, const double* sp, const double* ap,
                             int spec_dim, int ap_dim, float* lf0, float* mgc, float* bap) {
  OnDevice dev_(b->b.ctx[0]);
  return launch_recipe_features(b->b, f0, sp, ap, spec_dim, ap_dim, lf0, mgc, bap);
}
int WorldMi355RecipeDecode(WorldMi355Batch* b, const float* lf0, const float* mgc, const float* bap, int spec_dim,
                           int ap_dim, double* f0, double* sp, double* ap) {
  OnDevice dev_(b->b.ctx[0]);
  return launch_recipe_decode(b->b, lf0, mgc, bap, spec_dim, ap_dim, f0, sp, ap);
}
int WorldMi355ComposeCmp(WorldMi355Batch* b, int n_streams, const float* const* streams, const int* dims,
                         const int* n_windows, const double* const* const* windows,
                         const int* const* window_sizes, float* out) {
  OnDevice dev_(b->b.ctx[0]);
  return launch_compose_cmp(b->b, n_streams, streams, dims, n_windows, windows, window_sizes, out);
}
void WorldMi355HtkHeader(int n_frames, int sampling_rate, int frame_shift_samples, int bytes_per_frame,
                         int htk_type, unsigned char out12[12]) {              // addhtkheader.pl:60-75
  const int32_t a = n_frames, fs100 = (int32_t)(10000000.0 * frame_shift_samples / sampling_rate);
  const int16_t c = (int16_t)bytes_per_frame, d = (int16_t)htk_type;
  memcpy(out12, &a, 4);
  memcpy(out12 + 4, &fs100, 4);
  memcpy(out12 + 8, &c, 2);
  memcpy(out12 + 10, &d, 2);
}
int WorldMi355SamplesFromPcm16(WorldMi355Batch* b, const int16_t* pcm, double* x) {
  OnDevice dev_(b->b.ctx[0]);
  return wm::launch_pcm16_to_samples(b->b, pcm, x);
}
int WorldMi355SamplesToPcm16(WorldMi355Batch* b, const double* y, int16_t* pcm) {
  OnDevice dev_(b->b.ctx[0]);
  return wm::launch_samples_to_pcm16(b->b, y, pcm);
}
int WorldMi355Harvest(WorldMi355Batch* b, const double* x, double* t, double* f0) {
  OnDevice dev_(b->b.ctx[0]);
  return wm::launch_harvest(b->b, x, t, f0);
}
int WorldMi355Analyze(WorldMi355Batch* hb, const double* x, double* t, double* f0, double* sp,
                      double* ap) {
  OnDevice dev_(hb->b.ctx[0]);
  return launch_analyze(hb->b, x, t, f0, sp, ap);
}
int WorldMi355AnalyzeSynthesize(WorldMi355Batch* hb, const double* x, double* t, double* f0, double* sp,
                                double* ap, double* y) {
  OnDevice dev_(hb->b.ctx[0]);
  return launch_analyze_synthesize(hb->b, x, t, f0, sp, ap, y);
}
int WorldMi355Vibrato(WorldMi355Batch* hb, const float* lf0, const int* seg_utt_off, const int* seg_start,
                      const int* seg_end, const double* seg_pitch, float* vib, float* lf0_out, int* n_too_long) {
  OnDevice dev_(hb->b.ctx[0]);
  return launch_vibrato(hb->b, lf0, seg_utt_off, seg_start, seg_end, seg_pitch, vib, lf0_out, n_too_long);
}
int WorldMi355UtteranceStatus(WorldMi355Batch* hb, const double* x, const double* f0, const double* sp,
                              const double* ap, int* status) {
  OnDevice dev_(hb->b.ctx[0]);
  return launch_utterance_status(hb->b, x, f0, sp, ap, status);
}
#ifdef WM_PHASE
// debug builds only (make EXTRA=-DWM_PHASE): the shader-clock totals of the phase marks of one translation unit
// (0 = d4c.hip, 1 = cheaptrick.hip, 2 = synthesis.hip), read and cleared
__attribute__((visibility("default"))) int WorldMi355DebugPhases(int unit, unsigned long long* out32) {
  return unit == 0 ? wm::phase_read_d4c(out32) : unit == 1 ? wm::phase_read_cheaptrick(out32) : wm::phase_read_synthesis(out32);
}
#endif
void WorldMi355SetErrorHandler(WorldMi355ErrorHandler handler, void* user) {
  std::lock_guard<std::mutex> lock(g_err_mu);
  g_on_error = handler;
  g_on_error_user = user;
}
int WorldMi355TimingEnable(WorldMi355Context* h, int on) {
  OnDevice dev_(h->c);
  Context& c = h->c;
  int rc = wm_check(hipStreamSynchronize(c.stream));
  c.timing_clear();
  c.timing = on != 0;
  return rc;
}
int WorldMi355TimingQuery(WorldMi355Context* h, const char* kernel, double* total_ms, int* launches) {
  OnDevice dev_(h->c);
  Context& c = h->c;
  int rc = wm_check(hipStreamSynchronize(c.stream));
  if (rc) return rc;
  *total_ms = 0.0;
  *launches = 0;
  auto it = c.timed.find(kernel);
  if (it == c.timed.end()) return WM_OK;
  for (auto& pr : it->second) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { *total_ms += ms; *launches += 1; }
  }
  return WM_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// WORLD's own API: host pointers, one utterance, B = 1 batch.
// ---------------------------------------------------------------------------
namespace {

std::mutex g_mu;
WorldMi355Context* g_ctx = nullptr;

// A failure inside a drop-in entry point (no device, a HIP error, an unsupported size): the reference's functions
// return void, so by default it ends in a message on stderr and abort(), like the reference's own bad_alloc.  A host
// that would rather lose one utterance than the process installs a handler (WorldMi355SetErrorHandler): the failure
// then unwinds to the entry point, which calls the handler and returns with its outputs unspecified.
struct DropInError {
  const char* where;
  int code;
  WorldMi355ErrorHandler handler;   // the handler that was installed when the failure happened: a concurrent
  void* user;                       // WorldMi355SetErrorHandler(NULL) cannot make the failure vanish
};
[[noreturn]] void die(const char* where, int rc) {
  void* user = nullptr;
  if (WorldMi355ErrorHandler handler = error_handler(&user)) throw DropInError{where, rc, handler, user};
  fprintf(stderr, "libworld_mi355: %s failed (code %d): %s\n", where, rc, WorldMi355LastError());
  abort();
}

WorldMi355Context* default_context() {
  if (!g_ctx) {
    int rc = WorldMi355CreateContext(-1, nullptr, &g_ctx);
    if (rc) die("CreateContext", rc);
  }
  return g_ctx;
}

// ---- persistent workspace of the per-utterance entry points ------------------------------------------------
// The reference's API is stateless: every call brings host pointers and takes its results home.  Doing that
// literally -- a fresh batch, fresh device buffers and pageable copies per call -- cost 8 ms per utterance of 5 s,
// most of it allocation and staging rather than kernels.  What persists here between calls (per process, under the
// same mutex as the calls; nothing of it is visible to the caller):
//   * device buffers by role (x, t, f0, f0 out, sp, ap, coded, y), grow-only;
//   * one pinned staging buffer, grow-only: `double**` rows are gathered into / scattered from it, so the DMA
//     engine always sees pinned contiguous memory;
//   * the last few batches by (parameters, lengths): Dio, StoneMask, CheapTrick and D4C of one utterance, or the
//     same call on the next utterance of equal length, reuse descriptors and stage workspaces;
//   * a host copy of the last uploaded waveform: the four analysis calls of one utterance pass the same x, which
//     is then uploaded once (a memcmp of the samples decides, not the pointer).
enum Slot { kX, kT, kF0, kOut, kSp, kAp, kCoded, kY, kSlots };

struct Workspace {
  void* dev[kSlots] = {};
  size_t dev_cap[kSlots] = {};
  void* pinned = nullptr;
  size_t pinned_cap = 0;
  std::vector<double> x_copy;           // what dev[kX] holds
  bool x_valid = false;
  struct Entry { WorldMi355Params p; int xl, fl, yl; WorldMi355Batch* b; unsigned long stamp; };
  std::vector<Entry> cache;
  unsigned long clock = 0;

  double* device(Slot s, size_t n) {
    const size_t bytes = sizeof(double) * (n ? n : 1);
    if (bytes > dev_cap[s]) {
      if (s == kX) x_valid = false;
      if (dev[s]) wm::dev_free(dev[s]);
      dev[s] = nullptr;
      const size_t cap = bytes + bytes / 4;
      if (wm::dev_alloc(&dev[s], cap) != hipSuccess) die("hipMalloc", WM_ERR_HIP);
      dev_cap[s] = cap;
    }
    return (double*)dev[s];
  }
  // Pinned staging of one API call: regions are handed out one after another (no region is reused inside a call,
  // so no copy has to be waited for before the next argument is staged) and the call ends with finish(): ONE
  // stream synchronisation, then the host-side copies of the results.  Growing moves what the call has staged.
  size_t stage_at = 0;                   // doubles in use by the current call
  struct Pending { size_t off; double* flat; double** rows; size_t n; int n_rows, width; };
  std::vector<Pending> pending;
  double* stage(size_t n) {
    const size_t need = sizeof(double) * (stage_at + n + 8);
    if (need > pinned_cap) {
      if (hipStreamSynchronize(default_context()->c.stream) != hipSuccess) die("staging", WM_ERR_HIP);
      void* grown = nullptr;
      const size_t cap = need + need / 2;
      if (hipHostMalloc(&grown, cap, hipHostMallocDefault) != hipSuccess) die("hipHostMalloc", WM_ERR_HIP);
      if (pinned) {
        memcpy(grown, pinned, sizeof(double) * stage_at);
        (void)hipHostFree(pinned);
      }
      pinned = grown;
      pinned_cap = cap;
    }
    double* r = (double*)pinned + stage_at;
    stage_at += (n + 7) & ~(size_t)7;
    return r;
  }
  void finish() {
    if (hipStreamSynchronize(default_context()->c.stream) != hipSuccess) die("finish", WM_ERR_HIP);
    for (const Pending& q : pending) {
      const double* st = (const double*)pinned + q.off;
      if (q.flat) {
        if (q.n) memcpy(q.flat, st, sizeof(double) * q.n);
      } else {
        for (int i = 0; i < q.n_rows; ++i) memcpy(q.rows[i], st + (size_t)i * q.width, sizeof(double) * q.width);
      }
    }
    pending.clear();
    stage_at = 0;
  }
  WorldMi355Batch* batch(const WorldMi355Params& p, int xl, int fl, int yl) {
    ++clock;
    for (Entry& e : cache)
      if (e.xl == xl && e.fl == fl && e.yl == yl && memcmp(&e.p, &p, sizeof(p)) == 0) {
        e.stamp = clock;
        return e.b;
      }
    if (cache.size() >= 8) {                  // drop the least recently used
      size_t lru = 0;
      for (size_t i = 1; i < cache.size(); ++i)
        if (cache[i].stamp < cache[lru].stamp) lru = i;
      WorldMi355DestroyBatch(cache[lru].b);
      cache.erase(cache.begin() + (long)lru);
    }
    WorldMi355Batch* b = nullptr;
    int rc = WorldMi355CreateBatch(default_context(), &p, 1, xl >= 0 ? &xl : nullptr, fl >= 0 ? &fl : nullptr,
                                   yl >= 0 ? &yl : nullptr, &b);
    if (rc) die("CreateBatch", rc);
    Entry e;
    e.p = p; e.xl = xl; e.fl = fl; e.yl = yl; e.b = b; e.stamp = clock;
    cache.push_back(e);
    return b;
  }
};
// Never destroyed (a leaked singleton, like g_ctx): at process exit the HIP runtime may already be gone when static
// destructors run, and a destroyed cache whose batches are then unreachable is what a leak checker reports.
Workspace& g_ws = *new Workspace();

void drop_in_failed(const DropInError& e) {
  (void)hipDeviceSynchronize();               // nothing of the failed call may still write into the staging buffers
  g_ws.pending.clear();
  g_ws.stage_at = 0;
  g_ws.x_valid = false;
  e.handler(e.where, e.code, WorldMi355LastError(), e.user);
}

hipStream_t ws_stream() { return default_context()->c.stream; }

// the upload stream of Synthesis(): its sp / ap rows travel beside the f0-only kernels, not behind them
hipStream_t g_up_stream = nullptr;
hipEvent_t g_up_done = nullptr;
hipStream_t up_stream() {
  if (!g_up_stream) {
    if (hipStreamCreateWithFlags(&g_up_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&g_up_done, hipEventDisableTiming) != hipSuccess)
      die("upload stream", WM_ERR_HIP);
  }
  return g_up_stream;
}

void h2d(double* dst, const double* staged, size_t n, hipStream_t st = nullptr) {
  if (n && hipMemcpyAsync(dst, staged, sizeof(double) * n, hipMemcpyHostToDevice, st ? st : ws_stream()) != hipSuccess)
    die("H2D", WM_ERR_HIP);
}
// contiguous host array -> device slot through a staging region of this call (asynchronous: see Workspace::stage)
double* put(Slot s, const double* h, size_t n) {
  double* d = g_ws.device(s, n);
  double* st = g_ws.stage(n);
  if (n) memcpy(st, h, sizeof(double) * n);
  h2d(d, st, n);
  return d;
}
double* put_x(const double* x, size_t n) {
  if (g_ws.x_valid && g_ws.x_copy.size() == n && (n == 0 || memcmp(g_ws.x_copy.data(), x, sizeof(double) * n) == 0))
    return (double*)g_ws.dev[kX];
  g_ws.x_valid = false;
  double* d = put(kX, x, n);
  g_ws.x_copy.assign(x, x + n);
  g_ws.x_valid = true;
  return d;
}
double* put_rows(Slot s, const double* const* rows, int n_rows, int width, hipStream_t stream = nullptr) {
  const size_t n = (size_t)n_rows * width;
  double* d = g_ws.device(s, n);
  double* st = g_ws.stage(n);
  for (int i = 0; i < n_rows; ++i) memcpy(st + (size_t)i * width, rows[i], sizeof(double) * width);
  h2d(d, st, n, stream);
  return d;
}
// device -> a staging region (after the kernels on the stream); the host copy happens in finish()
size_t fetch(const double* d, size_t n) {
  double* st = g_ws.stage(n);
  const size_t off = (size_t)(st - (double*)g_ws.pinned);
  if (n && hipMemcpyAsync(st, d, sizeof(double) * n, hipMemcpyDeviceToHost, ws_stream()) != hipSuccess)
    die("D2H", WM_ERR_HIP);
  return off;
}
void get(const double* d, double* h, size_t n) {
  const size_t off = fetch(d, n);
  g_ws.pending.push_back({off, h, nullptr, n, 0, 0});
}
void get_rows(const double* d, double** rows, int n_rows, int width) {
  const size_t off = fetch(d, (size_t)n_rows * width);
  g_ws.pending.push_back({off, nullptr, rows, 0, n_rows, width});
}
void run_or_die(const char* where, int rc) {
  if (rc) die(where, rc);
}

// WORLD_MI355_TRACE=1: where the wall time of a drop-in call goes, to stderr (host staging / launches + kernels / the
// way back); synchronises after every stage, so the totals are somewhat above the untraced call's
struct CallTrace {
  bool on;
  double t0, last;
  const char* name;
  char line[256];
  int at = 0;
  static double now() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  }
  explicit CallTrace(const char* n) : name(n) {
    static const bool enabled = getenv("WORLD_MI355_TRACE") && atoi(getenv("WORLD_MI355_TRACE")) != 0;
    on = enabled;
    if (on) t0 = last = now();
    line[0] = 0;
  }
  void mark(const char* what, bool sync = false) {
    if (!on) return;
    if (sync) (void)hipStreamSynchronize(ws_stream());
    const double t = now();
    at += snprintf(line + at, sizeof(line) - (size_t)at, " %s %.3f", what, t - last);
    last = t;
  }
  ~CallTrace() {
    if (on) fprintf(stderr, "[world_mi355] %s:%s | total %.3f ms\n", name, line, now() - t0);
  }
};

}  // namespace

extern "C" {

int GetSamplesForDIO(int fs, int x_length, double frame_period) {      // dio.cpp:638-640
  return (int)(1000.0 * x_length / fs / frame_period) + 1;
}
void InitializeDioOption(DioOption* option) {                           // dio.cpp:649-665
  option->channels_in_octave = 2.0;
  option->f0_ceil = 800.0;
  option->f0_floor = 71.0;
  option->frame_period = 5;
  option->speed = 1;
  option->allowed_range = 0.1;
}
void Dio(const double* x, int x_length, int fs, const DioOption* option, double* temporal_positions,
         double* f0) {
  std::lock_guard<std::mutex> lock(g_mu);
  try {
    WorldMi355Params p;
    WorldMi355DefaultParams(fs, option->frame_period, &p);
    p.f0_floor = option->f0_floor; p.f0_ceil = option->f0_ceil;
    p.channels_in_octave = option->channels_in_octave; p.speed = option->speed;
    p.allowed_range = option->allowed_range;
    p.fft_size = 1024;   // unused by DIO
    CallTrace tr("Dio");
    WorldMi355Batch* b = g_ws.batch(p, x_length, -1, -1);
    tr.mark("batch");
    const size_t nf = (size_t)WorldMi355BatchTotalFrames(b);
    double* dx = put_x(x, (size_t)x_length);
    tr.mark("stage");
    double* dt = g_ws.device(kT, nf);
    double* df = g_ws.device(kF0, nf);
    run_or_die("Dio", WorldMi355Dio(b, dx, dt, df));
    tr.mark("launch");
    tr.mark("kernels", true);
    get(dt, temporal_positions, nf);
    get(df, f0, nf);
    g_ws.finish();
    tr.mark("back");
  } catch (const DropInError& e_) {
    drop_in_failed(e_);
  }
}

int GetSamplesForHarvest(int fs, int x_length, double frame_period) {  // harvest.cpp:1219-1221
  return (int)(1000.0 * x_length / fs / frame_period) + 1;
}
void InitializeHarvestOption(HarvestOption* option) {                   // harvest.cpp:1257-1262
  option->f0_ceil = 800.0;
  option->f0_floor = 71.0;
  option->frame_period = 5;
}
void Harvest(const double* x, int x_length, int fs, const HarvestOption* option,
             double* temporal_positions, double* f0) {
  std::lock_guard<std::mutex> lock(g_mu);
  try {
    WorldMi355Params p;
    WorldMi355DefaultParams(fs, option->frame_period, &p);
    p.f0_floor = option->f0_floor; p.f0_ceil = option->f0_ceil;
    p.fft_size = 1024;   // unused by Harvest
    WorldMi355Batch* b = g_ws.batch(p, x_length, -1, -1);
    const size_t nf = (size_t)WorldMi355BatchTotalFrames(b);
    double* dx = put_x(x, (size_t)x_length);
    double* dt = g_ws.device(kT, nf);
    double* df = g_ws.device(kF0, nf);
    run_or_die("Harvest", WorldMi355Harvest(b, dx, dt, df));
    get(dt, temporal_positions, nf);
    get(df, f0, nf);
    g_ws.finish();
  } catch (const DropInError& e_) {
    drop_in_failed(e_);
  }
}

void StoneMask(const double* x, int x_length, int fs, const double* temporal_positions, const double* f0,
               int f0_length, double* refined_f0) {
  std::lock_guard<std::mutex> lock(g_mu);
  try {
    WorldMi355Params p;
    WorldMi355DefaultParams(fs, 5.0, &p);
    p.fft_size = 1024;   // unused by StoneMask
    WorldMi355Batch* b = g_ws.batch(p, x_length, f0_length, -1);
    double* dx = put_x(x, (size_t)x_length);
    double* dt = put(kT, temporal_positions, (size_t)f0_length);
    double* df = put(kF0, f0, (size_t)f0_length);
    double* dr = g_ws.device(kOut, (size_t)f0_length);
    run_or_die("StoneMask", WorldMi355StoneMask(b, dx, dt, df, dr));
    get(dr, refined_f0, (size_t)f0_length);
    g_ws.finish();
  } catch (const DropInError& e_) {
    drop_in_failed(e_);
  }
}

int GetFFTSizeForCheapTrick(int fs, const CheapTrickOption* option) {   // cheaptrick.cpp:191-194
  return (int)pow(2.0, 1.0 + (int)(log(3.0 * fs / option->f0_floor + 1) / kLog2));
}
double GetF0FloorForCheapTrick(int fs, int fft_size) {                   // cheaptrick.cpp:196-198
  return 3.0 * fs / (fft_size - 3.0);
}
void InitializeCheapTrickOption(int fs, CheapTrickOption* option) {      // cheaptrick.cpp:230-239
  option->q1 = -0.15;
  option->f0_floor = 71.0;
  option->fft_size = GetFFTSizeForCheapTrick(fs, option);
}
void CheapTrick(const double* x, int x_length, int fs, const double* temporal_positions, const double* f0,
                int f0_length, const CheapTrickOption* option, double** spectrogram) {
  std::lock_guard<std::mutex> lock(g_mu);
  try {
    WorldMi355Params p;
    WorldMi355DefaultParams(fs, 5.0, &p);
    p.q1 = option->q1;
    p.fft_size = option->fft_size;
    const int w = option->fft_size / 2 + 1;
    CallTrace tr("CheapTrick");
    WorldMi355Batch* b = g_ws.batch(p, x_length, f0_length, -1);
    tr.mark("batch");
    double* dx = put_x(x, (size_t)x_length);
    double* dt = put(kT, temporal_positions, (size_t)f0_length);
    double* df = put(kF0, f0, (size_t)f0_length);
    tr.mark("stage");
    double* ds = g_ws.device(kSp, (size_t)f0_length * w);
    run_or_die("CheapTrick", WorldMi355CheapTrick(b, dx, dt, df, ds));
    tr.mark("launch");
    tr.mark("kernels", true);
    get_rows(ds, spectrogram, f0_length, w);
    g_ws.finish();
    tr.mark("back");
  } catch (const DropInError& e_) {
    drop_in_failed(e_);
  }
}

void InitializeD4COption(D4COption* option) { option->threshold = 0.85; }   // d4c.cpp:399-401
void D4C(const double* x, int x_length, int fs, const double* temporal_positions, const double* f0,
         int f0_length, int fft_size, const D4COption* option, double** aperiodicity) {
  std::lock_guard<std::mutex> lock(g_mu);
  try {
    WorldMi355Params p;
    WorldMi355DefaultParams(fs, 5.0, &p);
    p.fft_size = fft_size;
    p.d4c_threshold = option->threshold;
    const int w = fft_size / 2 + 1;
    WorldMi355Batch* b = g_ws.batch(p, x_length, f0_length, -1);
    double* dx = put_x(x, (size_t)x_length);
    double* dt = put(kT, temporal_positions, (size_t)f0_length);
    double* df = put(kF0, f0, (size_t)f0_length);
    CallTrace tr("D4C");
    double* da = g_ws.device(kAp, (size_t)f0_length * w);
    run_or_die("D4C", WorldMi355D4C(b, dx, dt, df, da));
    tr.mark("launch");
    tr.mark("kernels", true);
    get_rows(da, aperiodicity, f0_length, w);
    g_ws.finish();
    tr.mark("back");
  } catch (const DropInError& e_) {
    drop_in_failed(e_);
  }
}

// ---- world/codec.h --------------------------------------------------------------------------------
int GetNumberOfAperiodicities(int fs) { return codec_num_aperiodicities(fs); }   // codec.cpp:212-215

namespace {
// a frames-only batch: the codec needs fs, fft_size and the frame count
static WorldMi355Batch* codec_batch(int fs, int fft_size, int f0_length) {   // C linkage ignores the namespace
  WorldMi355Params p;
  WorldMi355DefaultParams(fs, 5.0, &p);
  p.fft_size = fft_size;
  return g_ws.batch(p, -1, f0_length, -1);
}
}  // namespace

void CodeSpectralEnvelope(const double* const* spectrogram, int f0_length, int fs, int fft_size,
                          int number_of_dimensions, double** coded_spectral_envelope) {
  std::lock_guard<std::mutex> lock(g_mu);
  try {
    const int w = fft_size / 2 + 1;
    WorldMi355Batch* b = codec_batch(fs, fft_size, f0_length);
    double* ds = put_rows(kSp, spectrogram, f0_length, w);
    double* dc = g_ws.device(kCoded, (size_t)f0_length * number_of_dimensions);
    run_or_die("CodeSpectralEnvelope", WorldMi355CodeSpectralEnvelope(b, ds, number_of_dimensions, dc));
    get_rows(dc, coded_spectral_envelope, f0_length, number_of_dimensions);
    g_ws.finish();
  } catch (const DropInError& e_) {
    drop_in_failed(e_);
  }
}

void DecodeSpectralEnvelope(const double* const* coded_spectral_envelope, int f0_length, int fs, int fft_size,
                            int number_of_dimensions, double** spectrogram) {
  std::lock_guard<std::mutex> lock(g_mu);
  try {
    const int w = fft_size / 2 + 1;
    WorldMi355Batch* b = codec_batch(fs, fft_size, f0_length);
    double* dc = put_rows(kCoded, coded_spectral_envelope, f0_length, number_of_dimensions);
    double* ds = g_ws.device(kSp, (size_t)f0_length * w);
    run_or_die("DecodeSpectralEnvelope", WorldMi355DecodeSpectralEnvelope(b, dc, number_of_dimensions, ds));
    get_rows(ds, spectrogram, f0_length, w);
    g_ws.finish();
  } catch (const DropInError& e_) {
    drop_in_failed(e_);
  }
}

void CodeAperiodicity(const double* const* aperiodicity, int f0_length, int fs, int fft_size,
                      int number_of_aperiodicities, double** coded_aperiodicity) {
  std::lock_guard<std::mutex> lock(g_mu);
  try {
    if (number_of_aperiodicities != codec_num_aperiodicities(fs))
      die("CodeAperiodicity: number_of_aperiodicities must be GetNumberOfAperiodicities(fs)", WM_ERR_BAD_ARG);
    const int w = fft_size / 2 + 1;
    WorldMi355Batch* b = codec_batch(fs, fft_size, f0_length);
    double* da = put_rows(kAp, aperiodicity, f0_length, w);
    double* dc = g_ws.device(kCoded, (size_t)f0_length * number_of_aperiodicities);
    run_or_die("CodeAperiodicity", WorldMi355CodeAperiodicity(b, da, dc));
    get_rows(dc, coded_aperiodicity, f0_length, number_of_aperiodicities);
    g_ws.finish();
  } catch (const DropInError& e_) {
    drop_in_failed(e_);
  }
}

// Positional meaning of the reference's DEFINITION (codec.cpp:237-238): the 4th argument is the number
// of aperiodicities and the 5th the FFT size, whatever the header calls them.
void DecodeAperiodicity(const double* const* coded_aperiodicity, int f0_length, int fs, int arg4_number_of_aperiodicities,
                        int arg5_fft_size, double** aperiodicity) {
  std::lock_guard<std::mutex> lock(g_mu);
  try {
    const int nap = arg4_number_of_aperiodicities, fft_size = arg5_fft_size;
    if (nap != codec_num_aperiodicities(fs))
      die("DecodeAperiodicity: 4th argument must be GetNumberOfAperiodicities(fs) (codec.cpp:237-238)", WM_ERR_BAD_ARG);
    const int w = fft_size / 2 + 1;
    WorldMi355Batch* b = codec_batch(fs, fft_size, f0_length);
    double* dc = put_rows(kCoded, coded_aperiodicity, f0_length, nap);
    double* da = g_ws.device(kAp, (size_t)f0_length * w);
    run_or_die("DecodeAperiodicity", WorldMi355DecodeAperiodicity(b, dc, da));
    get_rows(da, aperiodicity, f0_length, w);
    g_ws.finish();
  } catch (const DropInError& e_) {
    drop_in_failed(e_);
  }
}

void Synthesis(const double* f0, int f0_length, const double* const* spectrogram,
               const double* const* aperiodicity, int fft_size, double frame_period, int fs, int y_length,
               double* y) {
  std::lock_guard<std::mutex> lock(g_mu);
  try {
    WorldMi355Params p;
    WorldMi355DefaultParams(fs, frame_period, &p);
    p.fft_size = fft_size;
    const int w = fft_size / 2 + 1;
    CallTrace tr("Synthesis");
    WorldMi355Batch* b = g_ws.batch(p, -1, f0_length, y_length);
    tr.mark("batch");
    // f0 first, and the f0-only stage of Synthesis behind it (time base, pulse search: a latency chain as long as the
    // utterance); the rows of sp / ap are gathered into pinned memory and sent while that runs
    double* df = put(kF0, f0, (size_t)f0_length);
    double* dy = g_ws.device(kY, (size_t)y_length);
    {
      OnDevice dev_(b->b.ctx[0]);
      run_or_die("Synthesis", synthesis_begin(b->b, df, dy));
    }
    tr.mark("begin");
    // (the device and pinned buffers are sized before anything of this call is in flight on the upload stream: growing
    // them synchronises the context's stream only)
    const size_t cells = (size_t)f0_length * w;
    (void)g_ws.device(kSp, cells);
    (void)g_ws.device(kAp, cells);
    (void)g_ws.stage(2 * cells + 16);
    g_ws.stage_at -= (2 * cells + 16 + 7) & ~(size_t)7;               // reserved, not taken
    hipStream_t up = up_stream();
    double* ds = put_rows(kSp, spectrogram, f0_length, w, up);
    double* da = put_rows(kAp, aperiodicity, f0_length, w, up);
    if (hipEventRecord(g_up_done, up) != hipSuccess) die("upload", WM_ERR_HIP);
    tr.mark("stage");
    {
      OnDevice dev_(b->b.ctx[0]);
      run_or_die("Synthesis", synthesis_prepare_wait(b->b));
      tr.mark("wait");
      if (hipStreamWaitEvent(ws_stream(), g_up_done, 0) != hipSuccess) die("upload", WM_ERR_HIP);
      run_or_die("Synthesis", synthesis_render(b->b, ds, da, dy));
    }
    tr.mark("launch");
    tr.mark("kernels", true);
    get(dy, y, (size_t)y_length);
    g_ws.finish();
    tr.mark("back");
  } catch (const DropInError& e_) {
    drop_in_failed(e_);
  }
}

}  // extern "C"

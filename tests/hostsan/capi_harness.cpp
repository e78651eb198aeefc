// Drives the HOST layer of libworld_mi355 (csrc/capi.cpp, context.cpp and the host halves of the .hip units) under
// AddressSanitizer + UBSan against tests/hostsan/hip_stub.cpp, the way the reference's CLIs call it
// (test/analysis.cpp:93-203, test/synth.cpp:103-106): host pointers, `double**` rows allocated ONE BY ONE at their
// exact size (analysis.cpp:172-176) -- a scatter that is one element off lands in a red zone.  Kernels do not run in
// the stub, so nothing here checks arithmetic; what is checked is that every call returns, outputs are written and
// finite, the error-handler path unwinds cleanly, and the sanitizers stay silent.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "world/cheaptrick.h"
#include "world/codec.h"
#include "world/d4c.h"
#include "world/dio.h"
#include "world/harvest.h"
#include "world/stonemask.h"
#include "world/synthesis.h"
#include "world_mi355.h"

extern "C" long HipStubMallocs();
extern "C" long HipStubLaunches();

static int g_handler_calls = 0;
static void on_error(const char* where, int code, const char* message, void* user) {
  ++g_handler_calls;
  *(int*)user = code;
  fprintf(stderr, "handler: %s code %d: %s\n", where, code, message ? message : "");
}

struct Rows {                         // T rows of `width` doubles, each its own allocation, poisoned with NaN
  std::vector<double*> p;
  int width;
  Rows(int T, int w) : p((size_t)T), width(w) {
    for (auto& r : p) {
      r = new double[(size_t)w];
      for (int j = 0; j < w; ++j) r[j] = NAN;
    }
  }
  ~Rows() { for (auto r : p) delete[] r; }
  bool all_finite() const {
    for (auto r : p)
      for (int j = 0; j < width; ++j)
        if (!isfinite(r[j])) return false;
    return true;
  }
};

static int one_utterance(int fs, double seconds, double frame_period, int fft_override, bool harvest) {
  const int n = (int)(fs * seconds);
  std::vector<double> x((size_t)n);
  for (int i = 0; i < n; ++i) x[(size_t)i] = 0.3 * sin(2.0 * M_PI * 140.0 * i / fs) + 1e-3 * ((i * 2654435761u >> 16) % 1000 - 500) / 500.0;
  DioOption dopt;
  InitializeDioOption(&dopt);
  dopt.frame_period = frame_period;
  dopt.speed = 1;
  dopt.f0_floor = 71.0;
  dopt.allowed_range = 0.1;
  const int T = GetSamplesForDIO(fs, n, frame_period);
  std::vector<double> t((size_t)T, NAN), f0((size_t)T, NAN), rf0((size_t)T, NAN);
  if (harvest) {
    HarvestOption hopt;
    InitializeHarvestOption(&hopt);
    hopt.frame_period = frame_period;
    if (GetSamplesForHarvest(fs, n, frame_period) != T) return 10;
    Harvest(x.data(), n, fs, &hopt, t.data(), f0.data());
  } else {
    Dio(x.data(), n, fs, &dopt, t.data(), f0.data());
  }
  // kernels do not run: give the later stages a plausible contour (time axis and f0 as the analysis would)
  for (int i = 0; i < T; ++i) { t[(size_t)i] = i * frame_period / 1000.0; f0[(size_t)i] = (i % 7 == 0) ? 0.0 : 100.0 + (i % 50); }
  StoneMask(x.data(), n, fs, t.data(), f0.data(), T, rf0.data());
  CheapTrickOption copt;
  InitializeCheapTrickOption(fs, &copt);
  if (fft_override) copt.fft_size = fft_override;
  const int F = copt.fft_size, bins = F / 2 + 1;
  if (fft_override == 0 && F != GetFFTSizeForCheapTrick(fs, &copt)) return 11;
  Rows sp(T, bins), ap(T, bins);
  CheapTrick(x.data(), n, fs, t.data(), f0.data(), T, &copt, sp.p.data());
  D4COption aopt;
  InitializeD4COption(&aopt);
  aopt.threshold = 0.0;
  D4C(x.data(), n, fs, t.data(), f0.data(), T, F, &aopt, ap.p.data());
  if (!sp.all_finite() || !ap.all_finite()) return 12;      // every row written, none past its end (ASan)
  const int ylen = (int)((T - 1) * frame_period / 1000.0 * fs) + 1;
  std::vector<double> y((size_t)ylen, NAN);
  Synthesis(f0.data(), T, sp.p.data(), ap.p.data(), F, frame_period, fs, ylen, y.data());
  for (double v : y)
    if (!isfinite(v)) return 13;
  // the codec entry points the CLIs link (codec.h)
  const int nd = 50 < F / 4 ? 50 : F / 4, na = GetNumberOfAperiodicities(fs);
  Rows mgc(T, nd), sp2(T, bins);
  CodeSpectralEnvelope(sp.p.data(), T, fs, F, nd, mgc.p.data());
  DecodeSpectralEnvelope(mgc.p.data(), T, fs, F, nd, sp2.p.data());
  if (!mgc.all_finite() || !sp2.all_finite()) return 14;
  if (na > 0) {
    Rows bap(T, na), ap2(T, bins);
    CodeAperiodicity(ap.p.data(), T, fs, F, na, bap.p.data());
    DecodeAperiodicity(bap.p.data(), T, fs, na, F, ap2.p.data());   // the reference DEFINITION's order (codec.cpp:237-238)
    if (!bap.all_finite() || !ap2.all_finite()) return 15;
  }
  return 0;
}

int main() {
  int last_code = 0;
  if (getenv("HIP_STUB_DEVICES") && atoi(getenv("HIP_STUB_DEVICES")) == 0) {
    // no device: the drop-in entry points report to the handler and return (reference entry points are void)
    WorldMi355SetErrorHandler(on_error, &last_code);
    DioOption dopt;
    InitializeDioOption(&dopt);
    std::vector<double> x(8000, 0.1), t(401), f0(401);
    Dio(x.data(), 8000, 16000, &dopt, t.data(), f0.data());
    if (g_handler_calls != 1 || last_code == 0) { printf("no-device: handler calls %d code %d\n", g_handler_calls, last_code); return 1; }
    WorldMi355Context* ctx = nullptr;
    if (WorldMi355CreateContext(-1, nullptr, &ctx) == 0 || ctx) return 2;
    printf("capi no-device ok\n");
    return 0;
  }
  WorldMi355SetErrorHandler(on_error, &last_code);
  struct Case { int fs; double sec, fp; int fft; bool hv; } cases[] = {
      {16000, 1.30, 5.0, 0, false}, {16000, 0.21, 5.0, 0, false}, {16000, 0.75, 1.0, 0, true}, {22050, 0.40, 5.0, 0, false},
      {48000, 0.35, 5.0, 0, false}, {48000, 0.30, 5.0, 4096, false}, {8000, 0.50, 10.0, 0, false}, {16000, 0.033, 5.0, 0, false},
      {44100, 0.25, 5.0, 0, true}, {16000, 0.50, 5.0, 2048, false}};
  for (const Case& c : cases) {
    const int before = g_handler_calls;
    const int rc = one_utterance(c.fs, c.sec, c.fp, c.fft, c.hv);
    if (rc != 0 && g_handler_calls == before) {       // a failure must have gone through the handler, never silently
      printf("case fs %d %.3f s fp %.1f fft %d: rc %d without a handler call\n", c.fs, c.sec, c.fp, c.fft, rc);
      return 3;
    }
  }
  // what the library refuses, it refuses through the handler: an fft size it has no transform for, a sampling rate
  // whose D4C sizes differ, an empty utterance
  {
    const int before = g_handler_calls;
    std::vector<double> x(4000, 0.1), t(200, 0.0), f0(200, 120.0);
    for (int i = 0; i < 200; ++i) t[(size_t)i] = i * 0.005;
    CheapTrickOption copt;
    InitializeCheapTrickOption(16000, &copt);
    copt.fft_size = 300;                                    // not a power of two
    Rows sp(200, 151);
    CheapTrick(x.data(), 4000, 16000, t.data(), f0.data(), 200, &copt, sp.p.data());
    DioOption dopt;
    InitializeDioOption(&dopt);
    Dio(x.data(), 0, 16000, &dopt, t.data(), f0.data());    // empty
    if (g_handler_calls == before) { printf("refusals did not reach the handler\n"); return 4; }
  }
  // the batched API: create, analyze into caller buffers, destroy -- and a context torn down with work behind it
  {
    WorldMi355Context* ctx = nullptr;
    const bool inject = getenv("HIP_STUB_FAIL_MALLOC_AFTER") != nullptr;       // allocation failures are the point then
    if (WorldMi355CreateContext(-1, nullptr, &ctx) != 0 || !ctx) return inject ? 0 : 5;
    WorldMi355Params p;
    WorldMi355DefaultParams(16000, 5.0, &p);
    const int lens[3] = {9000, 16000, 4321};
    WorldMi355Batch* b = nullptr;
    if (WorldMi355CreateBatch(ctx, &p, 3, lens, nullptr, nullptr, &b) != 0 || !b) {
      printf("CreateBatch: %s\n", WorldMi355LastError());
      if (!inject) return 6;
    }
    if (b) WorldMi355DestroyBatch(b);
    WorldMi355DestroyContext(ctx);
  }
  WorldMi355SetErrorHandler(nullptr, nullptr);
  printf("capi ok: %d handler calls, %ld device allocations, %ld launches\n", g_handler_calls, HipStubMallocs(), HipStubLaunches());
  return 0;
}

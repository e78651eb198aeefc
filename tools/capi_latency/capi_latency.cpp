// capi_latency.cpp -- steady-state cost of the drop-in WORLD API (host pointers, one utterance per call, double** rows)
// measured from C++, i.e. as the reference's own CLI would call it (test/analysis.cpp:93-203, test/synth.cpp:103-106).
//   make -C tools/capi_latency && tools/capi_latency/capi_latency [n_utt]        (GPU box)
// Utterances of different lengths (3.0 .. 6.1 s), so that nothing is reused between them except what the library
// keeps on purpose (context, randn table, device and pinned buffers).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <vector>

#include "world/cheaptrick.h"
#include "world/d4c.h"
#include "world/dio.h"
#include "world/stonemask.h"
#include "world/synthesis.h"

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  const int n_utt = argc > 1 ? atoi(argv[1]) : 16;
  const int fs = 16000;
  const double fp = 5.0;
  double acc[5] = {0, 0, 0, 0, 0};
  long frames = 0;
  for (int u = -1; u < n_utt; ++u) {                       // u = -1: warm-up (context, tables), not counted
    const int n = u < 0 ? (int)(6.2 * fs) : (int)((3.0 + 3.1 * ((u * 37) % 16) / 15.0) * fs);
    std::vector<double> x((size_t)n);
    double ph = 0.0;
    unsigned s = 12345u + (unsigned)u;
    for (int i = 0; i < n; ++i) {
      const double f0 = 140.0 * (1.0 + 0.2 * sin(2 * M_PI * 0.7 * i / fs)) + (double)((10 * (u + 1)) % 60);
      ph += 2 * M_PI * f0 / fs;
      double v = 0.0;
      for (int h = 1; h <= 20; ++h) v += sin(h * ph) / h;
      s = s * 1664525u + 1013904223u;
      x[(size_t)i] = 0.2 * v / 2.0 + 1e-3 * ((s >> 8) / 8388608.0 - 1.0);
    }
    double t0 = now_ms();
    DioOption dopt;
    InitializeDioOption(&dopt);
    dopt.frame_period = fp; dopt.f0_floor = 71.0; dopt.allowed_range = 0.1; dopt.speed = 1;
    const int nf = GetSamplesForDIO(fs, n, fp);
    std::vector<double> t((size_t)nf), f0((size_t)nf), rf0((size_t)nf);
    Dio(x.data(), n, fs, &dopt, t.data(), f0.data());
    double t1 = now_ms();
    StoneMask(x.data(), n, fs, t.data(), f0.data(), nf, rf0.data());
    double t2 = now_ms();
    CheapTrickOption copt;
    InitializeCheapTrickOption(fs, &copt);
    const int F = copt.fft_size, w = F / 2 + 1;
    std::vector<double*> sp((size_t)nf), ap((size_t)nf);
    for (int i = 0; i < nf; ++i) { sp[(size_t)i] = new double[w]; ap[(size_t)i] = new double[w]; }   // as analysis.cpp:172-176
    CheapTrick(x.data(), n, fs, t.data(), rf0.data(), nf, &copt, sp.data());
    double t3 = now_ms();
    D4COption aopt;
    InitializeD4COption(&aopt);
    aopt.threshold = 0.0;
    D4C(x.data(), n, fs, t.data(), rf0.data(), nf, F, &aopt, ap.data());
    double t4 = now_ms();
    const int ny = (int)((nf - 1) * fp / 1000.0 * fs) + 1;
    std::vector<double> y((size_t)ny);
    Synthesis(rf0.data(), nf, sp.data(), ap.data(), F, fp, fs, ny, y.data());
    double t5 = now_ms();
    if (u >= 0) {
      acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3; acc[4] += t5 - t4;
      frames += nf;
    }
    int voiced = 0;
    for (int i = 0; i < nf; ++i) voiced += rf0[(size_t)i] > 0;
    if (u == 0) printf("first utterance: %d frames, %d voiced, y[1000] = %g\n", nf, voiced, y[1000]);
    for (int i = 0; i < nf; ++i) { delete[] sp[(size_t)i]; delete[] ap[(size_t)i]; }
  }
  const double tot = acc[0] + acc[1] + acc[2] + acc[3] + acc[4];
  printf("%d utterances, %.0f frames each on average: Dio %.2f  StoneMask %.2f  CheapTrick %.2f  D4C %.2f  Synthesis %.2f  "
         "= %.2f ms per utterance (%.0f frames/s)\n", n_utt, (double)frames / n_utt, acc[0] / n_utt, acc[1] / n_utt,
         acc[2] / n_utt, acc[3] / n_utt, acc[4] / n_utt, tot / n_utt, frames / tot * 1e3);
  return 0;
}

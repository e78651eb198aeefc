/* world_oracle_codec.c -- CPU restatement of WORLD's feature codec.
 *
 * TEST INFRASTRUCTURE ONLY (see world_oracle.h).  Follows externs/WORLD_v2/src/codec.cpp; pinned
 * against the compiled reference by tests/test_oracle_vs_ref.py and tests/golden/codec_*.npz.
 *
 * Notes on the reference that shape this file:
 *  - the c2c "backward" wrapper (fft.cpp BackwardFFT, c2c branch) returns conj(sum_j in[j] e^{-j 2 pi jk/n});
 *    DecodeOneFrame only reads the real parts, i.e. Re of the FORWARD DFT of the weighted cepstrum;
 *  - GetParametersForCoding fills fft_size/2 entries of a frequency axis that interp1 is told has
 *    fft_size/2 + 1 knots (codec.cpp:161-180, :126-129); the last knot is never reached because the
 *    largest mel-axis point lies below knot fft_size/2 - 1; it is defined here as the natural next value;
 *  - DecodeAperiodicity's definition takes (.., fs, number_of_aperiodicities, fft_size, ..) while the
 *    header declares (.., fs, fft_size, number_of_aperiodicities, ..) (codec.h:53-54 vs codec.cpp:237-238):
 *    same types, so the ABI is the definition's POSITIONAL meaning, which is what is restated.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "world_oracle.h"

#define ORC_M0 1127.01048                 /* constantnumbers.h: kM0 */
#define ORC_F0 700.0                      /* kF0 */
#define ORC_FLOOR_FREQ 40.0               /* kFloorFrequency */
#define ORC_CEIL_FREQ 20000.0             /* kCeilFrequency */
#define ORC_FREQ_INTERVAL 3000.0          /* kFrequencyInterval */
#define ORC_UPPER_LIMIT 15000.0           /* kUpperLimit */
#define ORC_SAFE 0.000000000001           /* kMySafeGuardMinimum */
#define ORC_PI_C 3.1415926535897932384

static double to_mel(double f) { return ORC_M0 * log(f / ORC_F0 + 1.0); }        /* codec.cpp:59-61 */
static double from_mel(double m) { return ORC_F0 * (exp(m / ORC_M0) - 1.0); }    /* codec.cpp:66-68 */

int orc_num_aperiodicities(int fs) {                                             /* codec.cpp:212-215 */
  double lim = fs / 2.0 - ORC_FREQ_INTERVAL;
  return (int)((ORC_UPPER_LIMIT < lim ? ORC_UPPER_LIMIT : lim) / ORC_FREQ_INTERVAL);
}

/* CodeAperiodicity, codec.cpp:217-235.  ap: [nf][fft_size/2+1], coded: [nf][nap]. */
void orc_code_aperiodicity(const double *ap, int nf, int fs, int fft_size, int nap, double *coded) {
  const int bins = fft_size / 2 + 1;
  double *axis = (double *)malloc(sizeof(double) * (size_t)(nap > 0 ? nap : 1));
  double *la = (double *)malloc(sizeof(double) * (size_t)bins);
  for (int i = 0; i < nap; ++i) axis[i] = ORC_FREQ_INTERVAL * (i + 1.0);
  for (int i = 0; i < nf; ++i) {
    for (int j = 0; j < bins; ++j) la[j] = 20 * log10(ap[(size_t)i * bins + j]);
    orc_interp1q(0, (double)fs / fft_size, la, bins, axis, nap, coded + (size_t)i * nap);
  }
  free(axis); free(la);
}

/* DecodeAperiodicity, codec.cpp:237-266 (+ CheckVUV :30-41, GetAperiodicity :46-54).
 * Positional arguments of the DEFINITION: (coded, nf, fs, nap, fft_size, ap). */
void orc_decode_aperiodicity(const double *coded, int nf, int fs, int nap, int fft_size, double *ap) {
  const int bins = fft_size / 2 + 1;
  double *faxis = (double *)malloc(sizeof(double) * (size_t)bins);
  double *caxis = (double *)malloc(sizeof(double) * (size_t)(nap + 2));
  double *cval = (double *)malloc(sizeof(double) * (size_t)(nap + 2));
  for (int i = 0; i < bins; ++i) faxis[i] = (double)fs / fft_size * i;
  for (int i = 0; i <= nap; ++i) caxis[i] = i * ORC_FREQ_INTERVAL;
  caxis[nap + 1] = fs / 2.0;
  cval[0] = -60.0;
  cval[nap + 1] = -ORC_SAFE;
  for (int i = 0; i < nf; ++i) {
    double *row = ap + (size_t)i * bins;
    for (int j = 0; j < bins; ++j) row[j] = 1.0 - ORC_SAFE;                      /* InitializeAperiodicity :20-25 */
    double tmp = 0.0;
    for (int k = 0; k < nap; ++k) {
      tmp += coded[(size_t)i * nap + k];
      cval[k + 1] = coded[(size_t)i * nap + k];
    }
    tmp /= nap;
    if (tmp > -0.5) continue;                                                    /* CheckVUV == 1 */
    orc_interp1(caxis, cval, nap + 2, faxis, bins, row);
    for (int j = 0; j < bins; ++j) row[j] = pow(10.0, row[j] / 20.0);
  }
  free(faxis); free(caxis); free(cval);
}

/* CodeSpectralEnvelope, codec.cpp:268-295 (+ GetParametersForCoding :161-180, CodeOneFrame :122-133,
 * DCTForCodec :73-88).  sp: [nf][fft_size/2+1], coded: [nf][ndim]. */
void orc_code_spectral_envelope(const double *sp, int nf, int fs, int fft_size, int ndim, double *coded) {
  const int md = fft_size / 2, bins = fft_size / 2 + 1;
  const double ceilf = fs / 2.0 < ORC_CEIL_FREQ ? fs / 2.0 : ORC_CEIL_FREQ;
  const double floor_mel = to_mel(ORC_FLOOR_FREQ), ceil_mel = to_mel(ceilf);
  double *mel_axis = (double *)malloc(sizeof(double) * (size_t)md);
  double *faxis = (double *)malloc(sizeof(double) * (size_t)bins);
  double *wr = (double *)malloc(sizeof(double) * (size_t)md), *wi = (double *)malloc(sizeof(double) * (size_t)md);
  double *ls = (double *)malloc(sizeof(double) * (size_t)bins);
  double *ms = (double *)malloc(sizeof(double) * (size_t)md);
  double *wave = (double *)malloc(sizeof(double) * (size_t)md);
  double *re = (double *)malloc(sizeof(double) * (size_t)(md / 2 + 1)), *im = (double *)malloc(sizeof(double) * (size_t)(md / 2 + 1));
  for (int i = 0; i < md; ++i) {
    mel_axis[i] = (ceil_mel - floor_mel) * i / md + floor_mel;
    wr[i] = 2.0 * cos(i * ORC_PI_C / fft_size) / sqrt((double)fft_size);
    wi[i] = 2.0 * sin(i * ORC_PI_C / fft_size) / sqrt((double)fft_size);
  }
  wr[0] /= sqrt(2.0);
  for (int i = 0; i < md; ++i) faxis[i] = to_mel((double)i * fs / fft_size);
  faxis[md] = to_mel((double)md * fs / fft_size);       /* never reached, see the header comment */
  const double norm = sqrt((double)md);
  for (int t = 0; t < nf; ++t) {
    for (int j = 0; j < bins; ++j) ls[j] = log(sp[(size_t)t * bins + j]);
    orc_interp1(faxis, ls, bins, mel_axis, md, ms);
    const int bias = md / 2;
    for (int i = 0; i < md / 2; ++i) {
      wave[i] = ms[i * 2];
      wave[i + bias] = ms[md - (i * 2) - 1];
    }
    orc_fft_r2c(wave, md, re, im);
    for (int i = 0; i < ndim; ++i) coded[(size_t)t * ndim + i] = (re[i] * wr[i] - im[i] * wi[i]) / norm;
  }
  free(mel_axis); free(faxis); free(wr); free(wi); free(ls); free(ms); free(wave); free(re); free(im);
}

/* own radix-2 complex DFT (forward, e^{-j}); md is a power of two */
static void dft_forward(double *re, double *im, int n) {
  for (int i = 1, j = 0; i < n; ++i) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { double a = re[i]; re[i] = re[j]; re[j] = a; a = im[i]; im[i] = im[j]; im[j] = a; }
  }
  for (int len = 2; len <= n; len <<= 1) {
    const int half = len >> 1;
    for (int base = 0; base < n; base += len)
      for (int k = 0; k < half; ++k) {
        const double ang = -2.0 * ORC_PI_C * k / len, c = cos(ang), s = sin(ang);
        const int p = base + k, q = p + half;
        const double xr = re[q] * c - im[q] * s, xi = re[q] * s + im[q] * c;
        re[q] = re[p] - xr; im[q] = im[p] - xi;
        re[p] += xr; im[p] += xi;
      }
  }
}

/* DecodeSpectralEnvelope, codec.cpp:297-324 (+ GetParametersForDecoding :185-208, DecodeOneFrame :138-157,
 * IDCTForCodec :93-117).  coded: [nf][ndim], sp: [nf][fft_size/2+1]. */
void orc_decode_spectral_envelope(const double *coded, int nf, int fs, int fft_size, int ndim, double *sp) {
  const int md = fft_size / 2, bins = fft_size / 2 + 1;
  const double ceilf = fs / 2.0 < ORC_CEIL_FREQ ? fs / 2.0 : ORC_CEIL_FREQ;
  const double floor_mel = to_mel(ORC_FLOOR_FREQ), ceil_mel = to_mel(ceilf);
  double *mel_axis = (double *)malloc(sizeof(double) * (size_t)(md + 2));
  double *faxis = (double *)malloc(sizeof(double) * (size_t)bins);
  double *wr = (double *)malloc(sizeof(double) * (size_t)md), *wi = (double *)malloc(sizeof(double) * (size_t)md);
  double *ms = (double *)malloc(sizeof(double) * (size_t)(md + 2));
  double *re = (double *)malloc(sizeof(double) * (size_t)md), *im = (double *)malloc(sizeof(double) * (size_t)md);
  for (int i = 0; i < ndim; ++i) {
    wr[i] = cos(i * ORC_PI_C / fft_size) * sqrt((double)fft_size);
    wi[i] = sin(i * ORC_PI_C / fft_size) * sqrt((double)fft_size);
  }
  wr[0] /= sqrt(2.0);
  for (int i = 0; i < md; ++i) mel_axis[i + 1] = from_mel((ceil_mel - floor_mel) * i / md + floor_mel);
  mel_axis[0] = 0;
  mel_axis[md + 1] = fs / 2.0;
  for (int i = 0; i < bins; ++i) faxis[i] = (double)i * fs / fft_size;
  const double norm = sqrt((double)md);
  for (int t = 0; t < nf; ++t) {
    for (int i = 0; i < ndim; ++i) {
      re[i] = coded[(size_t)t * ndim + i] * wr[i] * norm;
      im[i] = -coded[(size_t)t * ndim + i] * wi[i] * norm;
    }
    for (int i = ndim; i < md; ++i) { re[i] = 0.0; im[i] = 0.0; }
    dft_forward(re, im, md);                       /* real part == the wrapper's conj(...) real part */
    for (int i = 0; i < md / 2; ++i) {
      ms[1 + i * 2] = re[i];
      ms[1 + i * 2 + 1] = re[md - i - 1];
    }
    ms[0] = ms[1];
    ms[md + 1] = ms[md];
    double *row = sp + (size_t)t * bins;
    orc_interp1(mel_axis, ms, md + 2, faxis, bins, row);
    for (int i = 0; i < bins; ++i) row[i] = exp(row[i] / md);
  }
  free(mel_axis); free(faxis); free(wr); free(wi); free(ms); free(re); free(im);
}

/* ------------------------------------------------------------------------------------------------ */
/* delta windows and HTK packing of the recipe's `cmp` stage.                 */
/* Restates data/scripts/window.pl:45-146 (dynamic-feature windows with edge clamping and the         */
/* -1e10 "ignore" value) and data/scripts/addhtkheader.pl:45-82 (12-byte header, native endian).     */
/* ------------------------------------------------------------------------------------------------ */
#define ORC_IGNORE (-1.0e+10)

/* one stream of one utterance: in[T][dim] float32 -> out[T][nwin*dim] float32 (window i occupies
 * columns (i-1)*dim .. i*dim-1).  win[i] is the window file's content: size, then `size` coefficients. */
void orc_window_stream(const float *in, int T, int dim, int nwin, const double *const *win, float *out) {
  for (int i = 1; i <= nwin; ++i) {
    const double *w = win[i - 1];
    const int size = (int)w[0];
    const int nlr = (size - 1) / 2;
    int chk[64];
    for (int j = 0; j <= size && j < 64; ++j) chk[j] = 1;            /* window.pl:83-94 */
    for (int j = 1; j <= size; ++j) { if (w[j] != 0.0) break; chk[j] = 0; }
    for (int j = size; j >= 1; --j) { if (w[j] != 0.0) break; chk[j] = 0; }
    for (int t = 0; t < T; ++t)
      for (int j = 0; j < dim; ++j) {
        int boundary = 0;                                            /* :106-123 */
        for (int k = -nlr; k <= nlr; ++k)
          if (chk[k + nlr + 1] == 1) {
            int l = t + k < 0 ? 0 : (t + k >= T ? T - 1 : t + k);
            if ((double)in[(size_t)l * dim + j] == ORC_IGNORE) boundary = 1;
          }
        double acc = ORC_IGNORE;
        if (!boundary) {                                             /* :124-137 */
          acc = 0.0;
          for (int k = -nlr; k <= nlr; ++k) {
            int l = t + k < 0 ? 0 : (t + k >= T ? T - 1 : t + k);
            acc += w[k + nlr + 1] * (double)in[(size_t)l * dim + j];
          }
        }
        out[(size_t)t * nwin * dim + (size_t)dim * (i - 1) + j] = (float)acc;
      }
  }
}

/* addhtkheader.pl:60-75: int32 nframes, int32 frame shift in 100 ns units (integer part of
 * 1e7 * frameshift / samprate), int16 bytes per frame, int16 type; native byte order */
void orc_htk_header(int nframes, int samprate, int frameshift, int bytes_per_frame, int type, unsigned char *out12) {
  int32_t a = nframes, b = (int32_t)(10000000.0 * frameshift / samprate);
  int16_t c = (int16_t)bytes_per_frame, d = (int16_t)type;
  unsigned char *p = out12;
  for (int i = 0; i < 4; ++i) p[i] = ((unsigned char *)&a)[i];
  for (int i = 0; i < 4; ++i) p[4 + i] = ((unsigned char *)&b)[i];
  for (int i = 0; i < 2; ++i) p[8 + i] = ((unsigned char *)&c)[i];
  for (int i = 0; i < 2; ++i) p[10 + i] = ((unsigned char *)&d)[i];
}

/* ---- decode side of the synth CLI's coded-feature form -----------------------
 * test/synth.cpp:151-256 turns the recipe's lf0 / mgc / bap back into f0 / sp / ap before Synthesis:
 *   f0  = exp(lf0), 0 stays 0                                   (ToF0, synth.cpp:80-88, :168-173)
 *   sp  = DecodeSpectralEnvelope(mgc with c0 - 12.0) / 1e4      (:198-217)
 *   ap  = exp(mgc2sp(bap with c0 + 9.210340, order, 0.55, 0)) / 1e4, bins 0 .. order-1 ONLY (:231-246),
 *         order = ap_dim, minus one when ap_dim is odd (:233-235).
 * The reference leaves bins >= order of every ap row uninitialised (new[] without a fill, :240); that is
 * a defect of the caller, not something a restatement can reproduce.  Here those bins are defined as 0.0
 * and flagged in DESIGN.md / INTEGRATION.md.
 * mgc2sp is the CLI's SPTK port (test/sptkfunctions.cpp:186-219): freqt (:596-631) from `order` to F/2
 * with a = -0.55, gnorm / gc2gc / ignorm with both gammas 0 (c0 -> log(exp(c0)), :331-365, the rest copied),
 * then c2sp (:256-274): real part of the F-point DFT of the cepstrum zero-padded to F.               */

void orc_freqt(const double *c1, int m1, double *c2, int m2, double a) {      /* sptkfunctions.cpp:596-631 */
  double *g = (double *)calloc((size_t)m2 + 1, sizeof(double));
  double *d = (double *)calloc((size_t)m2 + 1, sizeof(double));
  const double b = 1 - a * a;
  for (int i = m1; i >= 0; --i) {           /* the reference counts i = -m1 .. 0 and reads c1[-i] */
    d[0] = g[0];
    g[0] = c1[i] + a * d[0];
    if (m2 >= 1) {
      d[1] = g[1];
      g[1] = b * d[0] + a * d[1];
    }
    for (int j = 2; j <= m2; ++j) {
      d[j] = g[j];
      g[j] = d[j - 1] + a * (d[j] - g[j - 1]);
    }
  }
  memcpy(c2, g, sizeof(double) * ((size_t)m2 + 1));
  free(g);
  free(d);
}

/* log-amplitude spectrum x[0 .. nbins) of a mel-generalised cepstrum with gamma 0 (mgc2sp's x output) */
void orc_mgc2sp(const double *mgc, int m, double alpha, int fft_size, int nbins, double *x) {
  const int h = fft_size / 2;
  double *c = (double *)calloc((size_t)fft_size, sizeof(double));
  double *re = (double *)malloc(sizeof(double) * ((size_t)h + 1));
  double *im = (double *)malloc(sizeof(double) * ((size_t)h + 1));
  orc_freqt(mgc, m, c, h, (0.0 - alpha) / (1 - alpha * 0.0));                 /* mgc2mgc :229-254, a2 = 0 */
  c[0] = log(exp(c[0]));                                                      /* gnorm + ignorm, gamma 0  */
  orc_fft_r2c(c, fft_size, re, im);                                           /* c2sp: zero-padded, fftr  */
  for (int k = 0; k < nbins; ++k) x[k] = k <= h ? re[k] : re[fft_size - k];
  free(c);
  free(re);
  free(im);
}

void orc_recipe_decode(const float *lf0, const float *mgc, const float *bap, int nf, int fs, int fft_size,
                       int spec_dim, int ap_dim, double *f0, double *sp, double *ap) {
  const int w = fft_size / 2 + 1;
  const int order = (ap_dim % 2 == 1) ? ap_dim - 1 : ap_dim;
  double *cs = (double *)malloc(sizeof(double) * (size_t)nf * (size_t)spec_dim);
  double *row = (double *)malloc(sizeof(double) * ((size_t)order + 1));
  for (int i = 0; i < nf; ++i) {
    const double l = (double)lf0[i];
    f0[i] = l != 0 ? exp(l) : 0;
    for (int j = 0; j < spec_dim; ++j) cs[(size_t)i * spec_dim + j] = (double)mgc[(size_t)i * spec_dim + j];
    cs[(size_t)i * spec_dim] -= 12.0;
  }
  orc_decode_spectral_envelope(cs, nf, fs, fft_size, spec_dim, sp);
  for (size_t k = 0; k < (size_t)nf * (size_t)w; ++k) sp[k] /= 1e4;
  for (int i = 0; i < nf; ++i) {
    /* mgc2sp reads order + 1 coefficients; the row holds ap_dim of them (the CLI's row buffers are longer) */
    for (int j = 0; j <= order; ++j) row[j] = j < ap_dim ? (double)bap[(size_t)i * ap_dim + j] : 0.0;
    row[0] += 9.210340;
    double *out = ap + (size_t)i * w;
    orc_mgc2sp(row, order, 0.55, fft_size, order, out);
    for (int j = 0; j < order; ++j) out[j] = exp(out[j]) / 1e4;
    for (int j = order; j < w; ++j) out[j] = 0.0;                 /* uninitialised in the reference */
  }
  free(cs);
  free(row);
}

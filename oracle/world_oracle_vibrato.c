/*
 * world_oracle_vibrato.c -- CPU restatement of the recipe's vibrato feature.
 *
 * TEST INFRASTRUCTURE ONLY (see world_oracle.h).
 *
 * Follows data/scripts/Extract.py:115-227 of the reference: per label segment, the voiced runs (f0 >= 55 Hz)
 * that END inside the segment and are longer than 20 frames are detrended with LOWESS (it = 20, delta = 0,
 * default frac = 2/3, Extract.py:220) after subtracting the note's pitch, and getVibrate() (:115-158) turns the
 * zero crossings of the residual into a (depth, period) pair per frame.
 *
 * PARITY UNPINNED.  The reference holds no fixtures for this path, Extract.py cannot run here (statsmodels and
 * progressbar are not installed), and LOWESS itself lives in a dependency the reference does not
 * pin: statsmodels.nonparametric.lowess.  orc_lowess() restates that function's published algorithm (Cleveland
 * 1979 as implemented in statsmodels' _smoothers_lowess.pyx: k = int(frac n + 1e-10) nearest neighbours found
 * by a sliding window, tricube weights over the window radius, weighted linear fit, then `it` robustifying passes
 * with bisquare weights of |residual| / (6 median |residual|), residuals >= 1 clipped to weight 0).
 *
 * Quirks of the script kept as they are:
 *   - getVibrate() APPENDS its (peak, period) pairs to a list that already holds `length` zero pairs
 *     (Extract.py:148,150), so the caller's copy (:223-225) zeroes the run itself and lays the pairs out AFTER
 *     it, over whatever frames follow; a later run overwrites an earlier run's spill-over;
 *   - period = end - start / 2 (precedence: end - (start / 2), :147);
 *   - a half-wave whose peak is below 5 is skipped, which shifts every later pair forward (:145-146), and the
 *     tail (:149-150) repeats the last peak seen (even a skipped one) with the last period assigned;
 *   - a run that reaches the end of its label segment without an unvoiced frame after it is never processed
 *     (oend stays 0, :210-211).
 * Where the script would raise (an empty crossing list, :149; a spill-over past the last frame, :224) this
 * restatement stops writing instead: no pairs, respectively pairs up to the last frame.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "world_oracle.h"

static int cmp_double(const void *a, const void *b) {
  const double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

/* statsmodels.nonparametric.lowess(y, x = 0..n-1, frac, it, delta = 0)[:, 1] */
void orc_lowess(const double *y, int n, double frac, int it, double *fit) {
  int k = (int)(frac * n + 1e-10);
  if (k < 2) k = 2;
  if (k > n) k = n;
  double *rw = (double *)malloc(sizeof(double) * (size_t)n);      /* robustness weights */
  double *w = (double *)malloc(sizeof(double) * (size_t)n);
  double *res = (double *)malloc(sizeof(double) * (size_t)n);
  for (int i = 0; i < n; ++i) rw[i] = 1.0;
  for (int pass = 0; pass <= it; ++pass) {
    int lo = 0, hi = k;                                           /* neighbourhood [lo, hi), carried along */
    for (int i = 0; i < n; ++i) {
      const double xi = (double)i;
      while (hi < n && xi - (double)lo > (double)hi - xi) { ++lo; ++hi; }
      const double r0 = xi - (double)lo, r1 = (double)(hi - 1) - xi;
      const double radius = r0 > r1 ? r0 : r1;
      double sum_w = 0.0;
      int nonzero = 0;
      for (int j = lo; j < hi; ++j) {
        double d = fabs((double)j - xi) / radius;
        d = 1.0 - d * d * d;
        d = d * d * d;                                            /* tricube */
        w[j] = d * rw[j];
        sum_w += w[j];
        nonzero += w[j] > 1e-12;
      }
      if (!(sum_w > 0.0) || nonzero < 2) {                        /* degenerate fit: keep the observation */
        fit[i] = y[i];
        continue;
      }
      for (int j = lo; j < hi; ++j) w[j] /= sum_w;
      double xm = 0.0, sq = 0.0, f = 0.0;
      for (int j = lo; j < hi; ++j) xm += w[j] * (double)j;
      for (int j = lo; j < hi; ++j) sq += w[j] * ((double)j - xm) * ((double)j - xm);
      for (int j = lo; j < hi; ++j)
        f += w[j] * (1.0 + (xi - xm) * ((double)j - xm) / sq) * y[j];
      fit[i] = f;
    }
    if (pass == it) break;
    for (int i = 0; i < n; ++i) res[i] = fabs(y[i] - fit[i]);
    memcpy(w, res, sizeof(double) * (size_t)n);
    qsort(w, (size_t)n, sizeof(double), cmp_double);
    const double med = (n & 1) ? w[n / 2] : 0.5 * (w[n / 2 - 1] + w[n / 2]);
    for (int i = 0; i < n; ++i) {
      double s;
      if (med == 0.0) s = res[i] > 0.0 ? 1.0 : 0.0;
      else s = res[i] / (6.0 * med);
      if (s >= 1.0) s = 1.0;
      s = 1.0 - s * s;
      rw[i] = s * s;                                              /* bisquare */
    }
  }
  free(rw); free(w); free(res);
}

/* getVibrate (Extract.py:115-158).  t receives 2 doubles per entry and needs room for 2 n entries; returns the
 * number of entries (n zero pairs, then the appended pairs). */
int orc_get_vibrate(const double *f, int n, double *t) {
  if (n <= 2) return 0;
  int len = n;
  memset(t, 0, sizeof(double) * 2 * (size_t)n);
  int positive = !(f[0] < 0);
  int *ip = (int *)malloc(sizeof(int) * (size_t)n);
  int nip = 0;
  for (int i = 0; i < n; ++i) {
    if (positive && f[i] <= 0) { ip[nip++] = i; positive = 0; }
    else if (!positive && f[i] >= 0) { ip[nip++] = i; positive = 1; }
  }
  double peak = 0.0, period = 0.0;
  for (int s = 0; s + 1 < nip; ++s) {
    const int start = ip[s], end = ip[s + 1];
    peak = 0.0;
    for (int j = start; j < end; ++j) peak = fabs(f[j]) > peak ? fabs(f[j]) : peak;
    if (peak < 5) continue;
    period = (double)end - (double)start / 2.0;
    for (int j = start; j < end; ++j) { t[2 * len] = peak; t[2 * len + 1] = period; ++len; }
  }
  if (nip > 0)
    for (int i = ip[nip - 1]; i < n; ++i) { t[2 * len] = peak; t[2 * len + 1] = period; ++len; }
  free(ip);
  return len;
}

/* The per-utterance loop of Extract.py:161-227 from f0 in Hz (after soprExp, :98-108) and the label segments
 * as frame ranges [seg_start, seg_end) with the note pitch of each (0 for "xx").  vib and df0 are [T][2], raw
 * (before soprLog).  Returns the number of runs that went through LOWESS. */
int orc_vibrato(const double *f0, int T, const int *seg_start, const int *seg_end, const double *seg_pitch,
                int nseg, double *vib, double *df0) {
  double *df02 = (double *)calloc((size_t)T + 1, sizeof(double));
  double *pf = (double *)malloc(sizeof(double) * ((size_t)T + 1));
  double *fit = (double *)malloc(sizeof(double) * ((size_t)T + 1));
  double *t = (double *)malloc(sizeof(double) * 4 * ((size_t)T + 1));
  memset(vib, 0, sizeof(double) * 2 * (size_t)T);
  memset(df0, 0, sizeof(double) * 2 * (size_t)T);
  int runs = 0;
  for (int s = 0; s < nseg; ++s) {
    const int start = seg_start[s] > 0 ? seg_start[s] : 0;
    const int end = seg_end[s] < T ? seg_end[s] : T;
    const double base = seg_pitch[s];
    for (int j = start; j < end; ++j) {                           /* :189-199 */
      double d = f0[j] - base + 500.0;
      if (d <= 0) d = -1.0;
      df0[2 * j] = f0[j];
      if (f0[j] < 55.0) { df0[2 * j + 1] = 0.0; df02[j] = 0.0; }
      else { df0[2 * j + 1] = d; df02[j] = f0[j] - base; }
    }
    int j = start;                                                /* :202-225 */
    while (j < end) {
      int ostart = j, oend = 0, first = 1;
      while (j < end) {
        if (first && f0[j] >= 55.0) { ostart = j; first = 0; }
        else if (!first && f0[j] < 55.0) { oend = j; break; }
        ++j;
      }
      if (oend == 0) continue;
      if (oend - ostart > 20) {
        const int n = oend - ostart;
        orc_lowess(df02 + ostart, n, 2.0 / 3.0, 20, fit);
        for (int i = 0; i < n; ++i) pf[i] = df02[ostart + i] - fit[i];
        const int len = orc_get_vibrate(pf, n, t);
        for (int k = 0; k < len && ostart + k < T; ++k) {
          vib[2 * (ostart + k)] = t[2 * k];
          vib[2 * (ostart + k) + 1] = t[2 * k + 1];
        }
        ++runs;
      }
    }
  }
  free(df02); free(pf); free(fit); free(t);
  return runs;
}

/* soprLog (Extract.py:86-96) followed by the float32 of saveVector (:52-61) */
void orc_sopr_log(const double *in, int n, float *out) {
  for (int i = 0; i < n; ++i) out[i] = (float)(in[i] <= 0 ? 1e-8 : log(in[i]));
}

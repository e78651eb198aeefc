/*
 * world_oracle.c -- CPU parity oracle for the WORLD hot path (plain C99).
 *
 * TEST INFRASTRUCTURE ONLY (see world_oracle.h).  A from-scratch restatement of
 * the reference algorithm; every function cites the reference file:line it
 * follows (paths relative to /root/reference/externs/WORLD_v2/src/).  The FFT
 * here is this file's own iterative radix-2 transform -- the reference's
 * Ooura FFT is NOT reproduced; only its wrapper conventions (fft.cpp:26-72)
 * are: r2c = standard e^{-j} half spectrum, c2r = unnormalised inverse that
 * ignores Im(DC), Im(Nyquist) and bins above n/2.
 *
 * Pinned against the compiled reference (oracle/_ref) by tests/test_oracle_vs_ref.py
 * and against the vectors under tests/golden generated from that build.
 */
#include "world_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.1415926535897932384      /* constantnumbers.h: kPi */
#define ORC_LOG2 0.69314718055994529       /* kLog2 */
#define ORC_SAFE 0.000000000001            /* kMySafeGuardMinimum */
#define ORC_EPS 0.00000000000000022204460492503131 /* kEps */
#define ORC_BIG 100000.0                   /* kMaximumValue */
#define ORC_DEFAULT_F0 500.0               /* kDefaultF0 */

static double *dalloc(size_t n) {
  double *p = (double *)calloc(n ? n : 1, sizeof(double));
  if (!p) abort();
  return p;
}
static int *ialloc(size_t n) {
  int *p = (int *)calloc(n ? n : 1, sizeof(int));
  if (!p) abort();
  return p;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------ */
/* X1: randn -- matlabfunctions.cpp:247-277.  Every consumer reseeds first,   */
/* so the stream is one universal table R[k].                                 */
/* ------------------------------------------------------------------------ */
typedef struct { uint32_t x, y, z, w; } xs128;
static uint32_t xs_step(xs128 *s) {
  uint32_t t = s->x ^ (s->x << 11);
  s->x = s->y; s->y = s->z; s->z = s->w;
  s->w = (s->w ^ (s->w >> 19)) ^ (t ^ (t >> 8));
  return s->w;
}
void orc_randn_table_u32(uint32_t *out, int count) {
  xs128 s = {123456789u, 362436069u, 521288629u, 88675123u};
  for (int k = 0; k < count; ++k) {
    uint32_t acc = 0;                       /* uint32 wrap-around is intended */
    for (int j = 0; j < 12; ++j) acc += xs_step(&s) >> 4;
    out[k] = acc;
  }
}
void orc_randn_table(double *out, int count) {
  uint32_t *u = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(count ? count : 1));
  orc_randn_table_u32(u, count);
  for (int k = 0; k < count; ++k) out[k] = u[k] / 268435456.0 - 6.0;
  free(u);
}

/* A cursor over R emulating the reference's global generator state. */
typedef struct { double *tab; int cap; int pos; } rng_cursor;
static void rng_init(rng_cursor *r) { r->tab = NULL; r->cap = 0; r->pos = 0; }
static double rng_next(rng_cursor *r) {
  if (r->pos >= r->cap) {
    int ncap = r->cap ? r->cap * 2 : (1 << 16);
    r->tab = (double *)realloc(r->tab, sizeof(double) * (size_t)ncap);
    if (!r->tab) abort();
    orc_randn_table(r->tab, ncap);           /* regenerate prefix: simple, exact */
    r->cap = ncap;
  }
  return r->tab[r->pos++];
}
static void rng_free(rng_cursor *r) { free(r->tab); }

/* ------------------------------------------------------------------------ */
/* small helpers                                                              */
/* ------------------------------------------------------------------------ */
int orc_matlab_round(double x) {             /* matlabfunctions.cpp:212-214 */
  return x > 0 ? (int)(x + 0.5) : (int)(x - 0.5);
}
int orc_suitable_fft_size(int sample) {      /* common.cpp:51-54 */
  return (int)pow(2.0, (int)(log((double)sample) / ORC_LOG2) + 1.0);
}

/* interp1 + histc -- matlabfunctions.cpp:136-182.  histc's merge yields      */
/* k = clamp(#{j : x[j] <= xi}, 1, n-1); linear extrapolation outside.        */
void orc_interp1(const double *x, const double *y, int n, const double *xi,
                 int m, double *yi) {
  for (int i = 0; i < m; ++i) {
    int lo = 0, hi = n;                      /* upper_bound(x, xi[i]) */
    while (lo < hi) {
      int mid = (lo + hi) / 2;
      if (x[mid] <= xi[i]) lo = mid + 1; else hi = mid;
    }
    int k = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    double h = x[k] - x[k - 1];
    double s = (xi[i] - x[k - 1]) / h;
    yi[i] = y[k - 1] + s * (y[k] - y[k - 1]);
  }
}

/* interp1Q -- matlabfunctions.cpp:220-241 (delta_y[n-1] = 0 convention). */
void orc_interp1q(double x0, double dx, const double *y, int n,
                  const double *xi, int m, double *yi) {
  for (int i = 0; i < m; ++i) {
    double q = (xi[i] - x0) / dx;
    int b = (int)q;
    double frac = q - b;
    double dy = (b == n - 1) ? 0.0 : y[b + 1] - y[b];
    yi[i] = y[b] + dy * frac;
  }
}

/* decimate -- matlabfunctions.cpp:27-125 (filter), :184-210 (driver). */
static const double kDecA[13][3] = {
  {0, 0, 0}, {0, 0, 0},
  {0.041156734567757189, -0.42599112459189636, 0.041037215479961225},
  {0.95039378983237421, -0.67429146741526791, 0.15412211621346475},
  {1.4499664446880227, -0.98943497080950582, 0.24578252340690215},
  {1.7610939654280557, -1.2554914843859768, 0.3237186507788215},
  {1.9715352749512141, -1.4686795689225347, 0.3893908434965701},
  {2.1225239019534703, -1.6395144861046302, 0.44469707800587366},
  {2.2357462340187593, -1.7780899984041358, 0.49152555365968692},
  {2.3236003491759578, -1.8921545617463598, 0.53148928133729068},
  {2.3936475118069387, -1.9873904075111861, 0.5658879979027055},
  {2.450743295230728, -2.06794904601978, 0.59574774438332101},
  {2.4981398605924205, -2.1368928194784025, 0.62187513816221485}};
static const double kDecB[13][2] = {
  {0, 0}, {0, 0},
  {0.16797464681802227, 0.50392394045406674},
  {0.071221945171178636, 0.21366583551353591},
  {0.036710750339322612, 0.11013225101796784},
  {0.021334858522387423, 0.06400457556716227},
  {0.013469181309343825, 0.040407543928031475},
  {0.0090366882681608418, 0.027110064804482525},
  {0.0063522763407111993, 0.019056829022133598},
  {0.0046331164041389372, 0.013899349212416812},
  {0.0034818622251927556, 0.010445586675578267},
  {0.0026822508007163792, 0.0080467524021491377},
  {0.0021097275904709001, 0.0063291827714127002}};

static void dec_iir(const double *in, int n, int r, double *out) {
  double a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0;
  if (r >= 2 && r <= 12) {
    a0 = kDecA[r][0]; a1 = kDecA[r][1]; a2 = kDecA[r][2];
    b0 = kDecB[r][0]; b1 = kDecB[r][1];
  }
  double w0 = 0, w1 = 0, w2 = 0;
  for (int i = 0; i < n; ++i) {
    double wt = in[i] + a0 * w0 + a1 * w1 + a2 * w2;
    out[i] = b0 * wt + b1 * w0 + b1 * w1 + b0 * w2;
    w2 = w1; w1 = w0; w0 = wt;
  }
}
void orc_decimate(const double *x, int n, int r, double *y) {
  const int pad = 9;
  int len = n + 2 * pad;
  double *u = dalloc((size_t)len), *v = dalloc((size_t)len);
  for (int i = 0; i < pad; ++i) u[i] = 2 * x[0] - x[pad - i];
  for (int i = 0; i < n; ++i) u[pad + i] = x[i];
  for (int i = 0; i < pad; ++i) u[pad + n + i] = 2 * x[n - 1] - x[n - 2 - i];
  dec_iir(u, len, r, v);
  for (int i = 0; i < len; ++i) u[i] = v[len - 1 - i];
  dec_iir(u, len, r, v);
  for (int i = 0; i < len; ++i) u[i] = v[len - 1 - i];
  int nout = (n - 1) / r + 1;
  int nbeg = r - r * nout + n;
  int c = 0;
  for (int i = nbeg; i < n + pad; i += r) y[c++] = u[i + pad - 1];
  free(u); free(v);
}

void orc_nuttall(int n, double *w) {         /* common.cpp:113-121 */
  for (int i = 0; i < n; ++i) {
    double t = i / (n - 1.0);
    w[i] = 0.355768 - 0.487396 * cos(2.0 * ORC_PI * t) +
           0.144232 * cos(4.0 * ORC_PI * t) - 0.012604 * cos(6.0 * ORC_PI * t);
  }
}

/* DCCorrection -- common.cpp:56-75 */
void orc_dc_correction(const double *in, double f0, int fs, int fft_size,
                       double *out) {
  int upper = 2 + (int)(f0 * fft_size / fs);
  int nrep = upper - 1;
  double *axis = dalloc((size_t)upper), *rep = dalloc((size_t)upper);
  for (int i = 0; i < upper; ++i) axis[i] = (double)i * fs / fft_size;
  orc_interp1q(f0 - axis[0], -(double)fs / fft_size, in, upper + 1, axis, nrep, rep);
  for (int i = 0; i < nrep; ++i) out[i] = in[i] + rep[i];
  free(axis); free(rep);
}

/* LinearSmoothing -- common.cpp:27-46 (set-up), :77-111 */
void orc_linear_smoothing(const double *in, double width, int fs, int fft_size,
                          double *out) {
  int half = fft_size / 2;
  int b = (int)(width * fft_size / fs) + 1;
  int len = half + 2 * b + 1;
  double *mir = dalloc((size_t)len), *seg = dalloc((size_t)len);
  double *ax = dalloc((size_t)half + 1), *lo = dalloc((size_t)half + 1),
         *hi = dalloc((size_t)half + 1);
  for (int i = 0; i < b; ++i) mir[i] = in[b - i];
  for (int i = b; i < half + b; ++i) mir[i] = in[i - b];
  for (int i = half + b; i <= half + 2 * b; ++i) mir[i] = in[half - (i - (half + b))];
  seg[0] = mir[0] * fs / fft_size;
  for (int i = 1; i < len; ++i) seg[i] = mir[i] * fs / fft_size + seg[i - 1];
  for (int i = 0; i <= half; ++i) ax[i] = (double)i / fft_size * fs - width / 2.0;
  double origin = -(b - 0.5) * fs / fft_size;
  double step = (double)fs / fft_size;
  orc_interp1q(origin, step, seg, len, ax, half + 1, lo);
  for (int i = 0; i <= half; ++i) ax[i] += width;
  orc_interp1q(origin, step, seg, len, ax, half + 1, hi);
  for (int i = 0; i <= half; ++i) out[i] = (hi[i] - lo[i]) / width;
  free(mir); free(seg); free(ax); free(lo); free(hi);
}

/* ------------------------------------------------------------------------ */
/* X2: FFT (own radix-2; reference wrapper conventions fft.cpp:26-72)         */
/* ------------------------------------------------------------------------ */
typedef struct { int n; double *c, *s; int *rev; } fft_tab;
#define ORC_MAX_LOG 24
static fft_tab g_tabs[ORC_MAX_LOG + 1];

static const fft_tab *get_tab(int n) {        /* n = complex size, power of two */
  int lg = 0;
  while ((1 << lg) < n) ++lg;
  fft_tab *t = &g_tabs[lg];
  if (t->n == n) return t;
  t->n = n;
  t->c = dalloc((size_t)n); t->s = dalloc((size_t)n); t->rev = ialloc((size_t)n);
  for (int k = 0; k < n; ++k) {
    t->c[k] = cos(2.0 * ORC_PI * k / n);
    t->s[k] = sin(2.0 * ORC_PI * k / n);
    int r = 0;
    for (int b = 0; b < lg; ++b) if (k & (1 << b)) r |= 1 << (lg - 1 - b);
    t->rev[k] = r;
  }
  return t;
}

/* in-place complex FFT, sign = -1 forward (e^{-j}), +1 backward; unnormalised */
static void fft_cplx(double *re, double *im, int n, int sign) {
  if (n == 1) return;
  const fft_tab *t = get_tab(n);
  for (int i = 0; i < n; ++i) {
    int j = t->rev[i];
    if (j > i) {
      double a = re[i]; re[i] = re[j]; re[j] = a;
      a = im[i]; im[i] = im[j]; im[j] = a;
    }
  }
  for (int len = 2; len <= n; len <<= 1) {
    int half = len >> 1, step = n / len;
    for (int base = 0; base < n; base += len) {
      for (int k = 0; k < half; ++k) {
        double wr = t->c[k * step], wi = sign * t->s[k * step];
        int p = base + k, q = p + half;
        double xr = re[q] * wr - im[q] * wi, xi = re[q] * wi + im[q] * wr;
        re[q] = re[p] - xr; im[q] = im[p] - xi;
        re[p] += xr; im[p] += xi;
      }
    }
  }
}

void orc_fft_r2c(const double *x, int n, double *re, double *im) {
  int h = n / 2;
  if (n == 1) { re[0] = x[0]; im[0] = 0; return; }
  double *zr = dalloc((size_t)h), *zi = dalloc((size_t)h);
  for (int j = 0; j < h; ++j) { zr[j] = x[2 * j]; zi[j] = x[2 * j + 1]; }
  fft_cplx(zr, zi, h, -1);
  const fft_tab *t = get_tab(n);
  for (int k = 0; k <= h; ++k) {
    int a = k % h, b = (h - k) % h;
    double er = 0.5 * (zr[a] + zr[b]), ei = 0.5 * (zi[a] - zi[b]);
    double orr = 0.5 * (zi[a] + zi[b]), oi = -0.5 * (zr[a] - zr[b]);
    double wr, wi;
    if (k == h) { wr = -1.0; wi = 0.0; } else { wr = t->c[k]; wi = -t->s[k]; }
    re[k] = er + wr * orr - wi * oi;
    im[k] = ei + wr * oi + wi * orr;
  }
  im[0] = 0.0; im[h] = 0.0;
  free(zr); free(zi);
}

void orc_fft_c2r(const double *re, const double *im, int n, double *x) {
  int h = n / 2;
  double *zr = dalloc((size_t)h), *zi = dalloc((size_t)h);
  const fft_tab *t = get_tab(n);
  for (int k = 0; k < h; ++k) {
    double ar = re[k], ai = (k == 0) ? 0.0 : im[k];
    double br = re[h - k], bi = (h - k == h) ? 0.0 : -im[h - k];  /* conj(X[h-k]) */
    double sr = ar + br, si = ai + bi;         /* 2E */
    double dr = ar - br, di = ai - bi;         /* X[k]-conj(X[h-k]) */
    double wr = t->c[k], wi = t->s[k];         /* e^{+j 2 pi k / n} */
    double orr = dr * wr - di * wi, oi = dr * wi + di * wr;  /* 2O */
    zr[k] = sr - oi; zi[k] = si + orr;         /* 2E + j 2O */
  }
  fft_cplx(zr, zi, h, +1);
  for (int j = 0; j < h; ++j) { x[2 * j] = zr[j]; x[2 * j + 1] = zi[j]; }
  free(zr); free(zi);
}

/* GetMinimumPhaseSpectrum -- common.cpp:182-220; c2c wrapper fft.cpp:61-71   */
/* (output of the forward c2c is conj(sum c[j] e^{+j..}) = DFT of conj(c)).   */
void orc_min_phase(const double *log_spec, int fft_size, double *re, double *im) {
  int n = fft_size, h = n / 2;
  double *ls = dalloc((size_t)n), *cr = dalloc((size_t)n), *ci = dalloc((size_t)n);
  for (int i = 0; i <= h; ++i) ls[i] = log_spec[i];
  for (int i = h + 1; i < n; ++i) ls[i] = ls[n - i];
  orc_fft_r2c(ls, n, cr, ci);
  ci[0] *= -1.0;
  for (int i = 1; i < h; ++i) { cr[i] *= 2.0; ci[i] *= -2.0; }
  ci[h] *= -1.0;
  for (int i = h + 1; i < n; ++i) { cr[i] = 0.0; ci[i] = 0.0; }
  /* forward c2c of the reference == standard DFT of conj(input) */
  for (int i = 0; i < n; ++i) ci[i] = -ci[i];
  fft_cplx(cr, ci, n, -1);
  for (int i = 0; i <= h; ++i) {
    double a = exp(cr[i] / n);
    re[i] = a * cos(ci[i] / n);
    im[i] = a * sin(ci[i] / n);
  }
  free(ls); free(cr); free(ci);
}

/* ------------------------------------------------------------------------ */
/* D1-D6: DIO -- dio.cpp                                                      */
/* ------------------------------------------------------------------------ */
int orc_dio_samples(int fs, int x_length, double frame_period) {  /* dio.cpp:638 */
  return (int)(1000.0 * x_length / fs / frame_period) + 1;
}

/* ZeroCrossingEngine -- dio.cpp:357-393 / harvest.cpp:162-197 */
static int zero_cross(const double *s, int len, double fs, double *loc, double *itv) {
  int *edge = ialloc((size_t)len);
  int cnt = 0;
  for (int i = 0; i < len - 1; ++i)
    if (0.0 < s[i] && s[i + 1] <= 0.0) edge[cnt++] = i + 1;
  if (cnt < 2) { free(edge); return 0; }
  double *fine = dalloc((size_t)cnt);
  for (int i = 0; i < cnt; ++i)
    fine[i] = edge[i] - s[edge[i] - 1] / (s[edge[i]] - s[edge[i] - 1]);
  for (int i = 0; i < cnt - 1; ++i) {
    itv[i] = fs / (fine[i + 1] - fine[i]);
    loc[i] = (fine[i] + fine[i + 1]) / 2.0 / fs;
  }
  free(fine); free(edge);
  return cnt - 1;
}

typedef struct { double *loc[4], *itv[4]; int n[4]; } zc4;

/* GetFourZeroCrossingIntervals -- dio.cpp:402-435 (in-place sign/diff order) */
static void four_crossings(double *s, int len, double fs, zc4 *z) {
  for (int k = 0; k < 4; ++k) { z->loc[k] = dalloc((size_t)len); z->itv[k] = dalloc((size_t)len); }
  z->n[0] = zero_cross(s, len, fs, z->loc[0], z->itv[0]);
  for (int i = 0; i < len; ++i) s[i] = -s[i];
  z->n[1] = zero_cross(s, len, fs, z->loc[1], z->itv[1]);
  for (int i = 0; i < len - 1; ++i) s[i] = s[i] - s[i + 1];
  z->n[2] = zero_cross(s, len - 1, fs, z->loc[2], z->itv[2]);
  for (int i = 0; i < len - 1; ++i) s[i] = -s[i];
  z->n[3] = zero_cross(s, len - 1, fs, z->loc[3], z->itv[3]);
}
static void free_crossings(zc4 *z) {
  for (int k = 0; k < 4; ++k) { free(z->loc[k]); free(z->itv[k]); }
}

/* spectral product helper: (ar,ai) *= (br,bi) over bins 0..h */
static void cmul_inplace(double *ar, double *ai, const double *br, const double *bi, int h) {
  for (int i = 0; i <= h; ++i) {
    double t = ar[i] * br[i] - ai[i] * bi[i];
    ai[i] = ar[i] * bi[i] + ai[i] * br[i];
    ar[i] = t;
  }
}

static void dio_fix_contour(double frame_period, int nb, double **cand,
                            const double *best, int nf, double f0_floor,
                            double allowed, double *out);

void orc_dio(const double *x, int x_length, int fs, double f0_floor,
             double f0_ceil, double channels_in_octave, double frame_period,
             int speed, double allowed_range, double *t, double *f0) {
  /* dio.cpp:578-634 */
  int nb = 1 + (int)(log(f0_ceil / f0_floor) / ORC_LOG2 * channels_in_octave);
  double *bnd = dalloc((size_t)nb);
  for (int i = 0; i < nb; ++i) bnd[i] = f0_floor * pow(2.0, (i + 1) / channels_in_octave);
  int r = imax(imin(speed, 12), 1);
  int ylen = 1 + (int)(x_length / r);
  double afs = (double)fs / r;
  int n = orc_suitable_fft_size(ylen + 4 * (int)(1.0 + afs / bnd[0] / 2.0));
  int h = n / 2;

  /* GetSpectrumForEstimation -- dio.cpp:60-106 */
  double *y = dalloc((size_t)n);
  if (r != 1) orc_decimate(x, x_length, r, y);
  else for (int i = 0; i < x_length; ++i) y[i] = x[i];
  double mean = 0.0;
  for (int i = 0; i < ylen; ++i) mean += y[i];   /* N+1 samples when speed = 1 (quirk) */
  mean /= ylen;
  for (int i = 0; i < ylen; ++i) y[i] -= mean;
  for (int i = ylen; i < n; ++i) y[i] = 0.0;
  double *Yr = dalloc((size_t)h + 1), *Yi = dalloc((size_t)h + 1);
  orc_fft_r2c(y, n, Yr, Yi);
  { /* DesignLowCutFilter -- dio.cpp:40-53 */
    int cut = orc_matlab_round(afs / 50.0);
    int N = cut * 2 + 1, c = (N - 1) / 2;
    double *lc = y;
    for (int i = 1; i <= N; ++i) lc[i - 1] = 0.5 - 0.5 * cos(i * 2.0 * ORC_PI / (N + 1));
    for (int i = N; i < n; ++i) lc[i] = 0.0;
    double sum = 0.0;
    for (int i = 0; i < N; ++i) sum += lc[i];
    for (int i = 0; i < N; ++i) lc[i] = -lc[i] / sum;
    for (int i = 0; i < c; ++i) lc[n - c + i] = lc[i];
    for (int i = 0; i < N; ++i) lc[i] = lc[i + c];
    lc[0] += 1.0;
    double *Hr = dalloc((size_t)h + 1), *Hi = dalloc((size_t)h + 1);
    orc_fft_r2c(lc, n, Hr, Hi);
    cmul_inplace(Yr, Yi, Hr, Hi, h);
    free(Hr); free(Hi);
  }

  int nf = orc_dio_samples(fs, x_length, frame_period);
  for (int i = 0; i < nf; ++i) t[i] = i * frame_period / 1000.0;

  double **cand = (double **)malloc(sizeof(double *) * (size_t)nb);
  double **score = (double **)malloc(sizeof(double *) * (size_t)nb);
  double *flt = dalloc((size_t)n), *Wr = dalloc((size_t)h + 1), *Wi = dalloc((size_t)h + 1);
  double *ip[4];
  for (int k = 0; k < 4; ++k) ip[k] = dalloc((size_t)nf);
  for (int b = 0; b < nb; ++b) {
    cand[b] = dalloc((size_t)nf); score[b] = dalloc((size_t)nf);
    /* GetFilteredSignal -- dio.cpp:296-343 */
    int hal = orc_matlab_round(afs / bnd[b] / 2.0);
    orc_nuttall(hal * 4, flt);
    for (int i = hal * 4; i < n; ++i) flt[i] = 0.0;
    orc_fft_r2c(flt, n, Wr, Wi);
    cmul_inplace(Wr, Wi, Yr, Yi, h);           /* product is commutative */
    orc_fft_c2r(Wr, Wi, n, flt);
    int bias = hal * 2;
    for (int i = 0; i < ylen; ++i) flt[i] = flt[i + bias];
    /* events -- dio.cpp:402-435 */
    zc4 z;
    four_crossings(flt, ylen, afs, &z);
    /* GetF0CandidateContour -- dio.cpp:471-508 (+Sub :441-465) */
    if (z.n[0] > 2 && z.n[1] > 2 && z.n[2] > 2 && z.n[3] > 2) {
      for (int k = 0; k < 4; ++k) orc_interp1(z.loc[k], z.itv[k], z.n[k], t, nf, ip[k]);
      for (int i = 0; i < nf; ++i) {
        double c = (ip[0][i] + ip[1][i] + ip[2][i] + ip[3][i]) / 4.0;
        double s = sqrt(((ip[0][i] - c) * (ip[0][i] - c) + (ip[1][i] - c) * (ip[1][i] - c) +
                         (ip[2][i] - c) * (ip[2][i] - c) + (ip[3][i] - c) * (ip[3][i] - c)) / 3.0);
        if (c > bnd[b] || c < bnd[b] / 2.0 || c > f0_ceil || c < f0_floor) { c = 0.0; s = ORC_BIG; }
        cand[b][i] = c; score[b][i] = s;
      }
    } else {
      for (int i = 0; i < nf; ++i) { cand[b][i] = 0.0; score[b][i] = ORC_BIG; }
    }
    free_crossings(&z);
    for (int i = 0; i < nf; ++i) score[b][i] = score[b][i] / (cand[b][i] + ORC_SAFE);  /* dio.cpp:564 */
  }

  /* GetBestF0Contour -- dio.cpp:112-126 */
  double *best = dalloc((size_t)nf);
  for (int i = 0; i < nf; ++i) {
    double s = score[0][i];
    best[i] = cand[0][i];
    for (int b = 1; b < nb; ++b)
      if (s > score[b][i]) { s = score[b][i]; best[i] = cand[b][i]; }
  }
  dio_fix_contour(frame_period, nb, cand, best, nf, f0_floor, allowed_range, f0);

  for (int b = 0; b < nb; ++b) { free(cand[b]); free(score[b]); }
  for (int k = 0; k < 4; ++k) free(ip[k]);
  free(cand); free(score); free(best); free(flt); free(Wr); free(Wi);
  free(Yr); free(Yi); free(y); free(bnd);
}

/* SelectBestF0 -- dio.cpp:190-209 */
static double dio_select(double cur, double past, double **cand, int nb, int idx, double allowed) {
  double ref = (cur * 3.0 - past) / 2.0;
  double err = fabs(ref - cand[0][idx]), best = cand[0][idx];
  for (int b = 1; b < nb; ++b) {
    double e = fabs(ref - cand[b][idx]);
    if (e < err) { err = e; best = cand[b][idx]; }
  }
  if (fabs(1.0 - best / ref) > allowed) return 0.0;
  return best;
}

/* FixF0Contour + FixStep1-4 -- dio.cpp:132-289.  The reference returns       */
/* without writing f0 when nf <= voice_range_minimum (dio.cpp:266); this       */
/* restatement zero-fills in that case (documented divergence).               */
static void dio_fix_contour(double frame_period, int nb, double **cand,
                            const double *best, int nf, double f0_floor,
                            double allowed, double *out) {
  int vrm = (int)(0.5 + 1000.0 / frame_period / f0_floor) * 2 + 1;
  if (nf <= vrm) { for (int i = 0; i < nf; ++i) out[i] = 0.0; return; }
  double *base = dalloc((size_t)nf), *s1 = dalloc((size_t)nf), *s2 = dalloc((size_t)nf),
         *s3 = dalloc((size_t)nf);
  /* step 1 -- dio.cpp:132-150 */
  for (int i = vrm; i < nf - vrm; ++i) base[i] = best[i];
  for (int i = vrm; i < nf; ++i)
    s1[i] = fabs((base[i] - base[i - 1]) / (ORC_SAFE + base[i])) < allowed ? base[i] : 0.0;
  /* step 2 -- dio.cpp:156-169 */
  for (int i = 0; i < nf; ++i) s2[i] = s1[i];
  int c = (vrm - 1) / 2;
  for (int i = c; i < nf - c; ++i)
    for (int j = -c; j <= c; ++j)
      if (s1[i + j] == 0) { s2[i] = 0.0; break; }
  /* voiced sections -- dio.cpp:174-184 */
  int *pos = ialloc((size_t)nf), *neg = ialloc((size_t)nf), np = 0, nn = 0;
  for (int i = 1; i < nf; ++i) {
    if (s2[i] == 0 && s2[i - 1] != 0) neg[nn++] = i - 1;
    else if (s2[i - 1] == 0 && s2[i] != 0) pos[np++] = i;
  }
  /* step 3 (forward) -- dio.cpp:215-231 */
  for (int i = 0; i < nf; ++i) s3[i] = s2[i];
  for (int i = 0; i < nn; ++i) {
    int limit = i == nn - 1 ? nf - 1 : neg[i + 1];
    for (int j = neg[i]; j < limit; ++j) {
      s3[j + 1] = dio_select(s3[j], s3[j - 1], cand, nb, j + 1, allowed);
      if (s3[j + 1] == 0) break;
    }
  }
  /* step 4 (backward) -- dio.cpp:237-253 */
  for (int i = 0; i < nf; ++i) out[i] = s3[i];
  for (int i = np - 1; i >= 0; --i) {
    int limit = i == 0 ? 1 : pos[i - 1];
    for (int j = pos[i]; j > limit; --j) {
      out[j - 1] = dio_select(out[j], out[j + 1], cand, nb, j - 1, allowed);
      if (out[j - 1] == 0) break;
    }
  }
  free(base); free(s1); free(s2); free(s3); free(pos); free(neg);
}

/* ------------------------------------------------------------------------ */
/* S1-S3: StoneMask -- stonemask.cpp                                          */
/* ------------------------------------------------------------------------ */
/* FixF0 -- stonemask.cpp:96-117.  Bins above fft/2 are an out-of-bounds read  */
/* in the reference (reachable when 6*tentative_f0 > fs/2); here they read as  */
/* zero power (documented divergence in an undefined case).                   */
static double sm_fix_f0(const double *pw, const double *num, int fft_size, int fs,
                        double f0, int nh) {
  double numer = 0.0, denom = 0.0;
  for (int i = 0; i < nh; ++i) {
    int idx = orc_matlab_round(f0 * fft_size / fs * (i + 1));
    double p = idx <= fft_size / 2 ? pw[idx] : 0.0;
    double nm = idx <= fft_size / 2 ? num[idx] : 0.0;
    double inst = p == 0.0 ? 0.0 : (double)idx * fs / fft_size + nm / p * fs / 2.0 / ORC_PI;
    double amp = sqrt(p);
    numer += amp * inst;
    denom += amp * (i + 1);
  }
  return numer / (denom + ORC_SAFE);
}

static double sm_refine(const double *x, int x_length, int fs, double pos, double f0) {
  /* GetRefinedF0 -- stonemask.cpp:184-207 */
  if (f0 <= 40.0 || f0 > fs / 12.0) return 0.0;
  int hw = (int)(1.5 * fs / f0 + 1.0);
  int len = hw * 2 + 1;
  double wlen = (2.0 * hw + 1.0) / fs;
  int n = (int)pow(2.0, 2.0 + (int)(log(hw * 2.0 + 1.0) / ORC_LOG2));
  int h = n / 2;
  /* GetMeanF0 -- stonemask.cpp:136-178 */
  int *raw = ialloc((size_t)len);
  double *mw = dalloc((size_t)len), *dw = dalloc((size_t)len);
  for (int i = 0; i < len; ++i) {
    double bt = (double)(-hw + i) / fs;
    raw[i] = orc_matlab_round((pos + bt) * fs);                 /* :24-28 */
    double tm = (raw[i] - 1.0) / fs - pos;                      /* :33-43 */
    mw[i] = 0.42 + 0.5 * cos(2.0 * ORC_PI * tm / wlen) + 0.08 * cos(4.0 * ORC_PI * tm / wlen);
  }
  dw[0] = -mw[1] / 2.0;                                         /* :49-55 */
  for (int i = 1; i < len - 1; ++i) dw[i] = -(mw[i + 1] - mw[i - 1]) / 2.0;
  dw[len - 1] = mw[len - 2] / 2.0;
  double *buf = dalloc((size_t)n), *mr = dalloc((size_t)h + 1), *mi = dalloc((size_t)h + 1),
         *dr = dalloc((size_t)h + 1), *di = dalloc((size_t)h + 1);
  for (int i = 0; i < len; ++i) buf[i] = x[imax(0, imin(x_length - 1, raw[i] - 1))] * mw[i];
  orc_fft_r2c(buf, n, mr, mi);
  for (int i = 0; i < len; ++i) buf[i] = x[imax(0, imin(x_length - 1, raw[i] - 1))] * dw[i];
  orc_fft_r2c(buf, n, dr, di);
  double *pw = dalloc((size_t)h + 1), *num = dalloc((size_t)h + 1);
  for (int j = 0; j <= h; ++j) {
    num[j] = mr[j] * di[j] - mi[j] * dr[j];
    pw[j] = mr[j] * mr[j] + mi[j] * mi[j];
  }
  /* GetTentativeF0 -- stonemask.cpp:122-131 */
  double tent = sm_fix_f0(pw, num, n, fs, f0, 2);
  double mean = (tent <= 0.0 || tent > f0 * 2) ? 0.0 : sm_fix_f0(pw, num, n, fs, tent, 6);
  if (fabs(mean - f0) / f0 > 0.2) mean = f0;                    /* :202 */
  free(raw); free(mw); free(dw); free(buf); free(mr); free(mi); free(dr); free(di);
  free(pw); free(num);
  return mean;
}

void orc_stonemask(const double *x, int x_length, int fs, const double *t,
                   const double *f0, int nf, double *refined) {
  for (int i = 0; i < nf; ++i) refined[i] = sm_refine(x, x_length, fs, t[i], f0[i]);
}

/* ------------------------------------------------------------------------ */
/* C1-C6: CheapTrick -- cheaptrick.cpp                                        */
/* ------------------------------------------------------------------------ */
int orc_cheaptrick_fft_size(int fs, double f0_floor) {          /* :191-194 */
  return (int)pow(2.0, 1.0 + (int)(log(3.0 * fs / f0_floor + 1) / ORC_LOG2));
}
double orc_cheaptrick_f0_floor(int fs, int fft_size) {           /* :196-198 */
  return 3.0 * fs / (fft_size - 3.0);
}

/* F0-adaptive windowing shared by CheapTrick (cheaptrick.cpp:87-142; Hann,   */
/* ratio 3, L2-normalised) and D4C (d4c.cpp:21-84; type 1 Hann / 2 Blackman,  */
/* ratio given, not normalised).  Writes 2*hw+1 samples, consumes as many     */
/* randn draws.                                                               */
static int windowed_waveform(const double *x, int x_length, int fs, double f0,
                             double pos, int type, double ratio, int normalise,
                             rng_cursor *rng, double *wave) {
  int hw = orc_matlab_round(ratio * fs / f0 / 2.0);
  int len = 2 * hw + 1;
  int origin = orc_matlab_round(pos * fs + 0.001);
  double *win = dalloc((size_t)len);
  double energy = 0.0;
  for (int i = 0; i < len; ++i) {
    int bi = i - hw;
    double p;
    if (normalise) {                      /* cheaptrick.cpp:100-104 */
      p = bi / 1.5 / fs;
      win[i] = 0.5 * cos(ORC_PI * p * f0) + 0.5;
      energy += win[i] * win[i];
    } else {                              /* d4c.cpp:33-45 */
      p = (2.0 * bi / ratio) / fs;
      if (type == 1) win[i] = 0.5 * cos(ORC_PI * p * f0) + 0.5;
      else win[i] = 0.42 + 0.5 * cos(ORC_PI * p * f0) + 0.08 * cos(ORC_PI * p * f0 * 2);
    }
  }
  if (normalise) {
    energy = sqrt(energy);
    for (int i = 0; i < len; ++i) win[i] /= energy;
  }
  for (int i = 0; i < len; ++i) {
    int si = imin(x_length - 1, imax(0, origin + i - hw));
    wave[i] = x[si] * win[i] + rng_next(rng) * ORC_SAFE;
  }
  double s1 = 0, s2 = 0;
  for (int i = 0; i < len; ++i) { s1 += wave[i]; s2 += win[i]; }
  double coef = s1 / s2;
  for (int i = 0; i < len; ++i) wave[i] -= win[i] * coef;
  free(win);
  return len;
}

void orc_cheaptrick(const double *x, int x_length, int fs, const double *t,
                    const double *f0, int nf, double q1, int fft_size, double *sp) {
  int n = fft_size, h = n / 2;
  rng_cursor rng; rng_init(&rng);                                  /* reseed :205 */
  double floor_f0 = orc_cheaptrick_f0_floor(fs, n);
  double *wave = dalloc((size_t)n), *re = dalloc((size_t)h + 1), *im = dalloc((size_t)h + 1);
  double *zero = dalloc((size_t)h + 1);
  for (int f = 0; f < nf; ++f) {
    double cf0 = f0[f] <= floor_f0 ? ORC_DEFAULT_F0 : f0[f];
    /* GetWindowedWaveform -- :112-142 (hw = round(1.5 fs / f0) = ratio 3) */
    int len = windowed_waveform(x, x_length, fs, cf0, t[f], 1, 3.0, 1, &rng, wave);
    /* GetPowerSpectrum -- :64-82 */
    for (int i = len; i < n; ++i) wave[i] = 0.0;
    orc_fft_r2c(wave, n, re, im);
    for (int i = 0; i <= h; ++i) wave[i] = re[i] * re[i] + im[i] * im[i];
    orc_dc_correction(wave, cf0, fs, n, wave);
    /* LinearSmoothing -- :176 */
    orc_linear_smoothing(wave, cf0 * 2.0 / 3.0, fs, n, wave);
    /* AddInfinitesimalNoise -- :147-151 */
    for (int i = 0; i <= h; ++i) wave[i] = wave[i] + fabs(rng_next(&rng)) * ORC_EPS;
    /* SmoothingWithRecovery -- :22-57 */
    for (int i = 0; i <= h; ++i) wave[i] = log(wave[i]);
    for (int i = 1; i < h; ++i) wave[n - i] = wave[i];
    orc_fft_r2c(wave, n, re, im);
    for (int i = 0; i <= h; ++i) {
      double sl, cl;
      if (i == 0) { sl = 1.0; cl = (1.0 - 2.0 * q1) + 2.0 * q1; }
      else {
        double q = (double)i / fs;
        sl = sin(ORC_PI * cf0 * q) / (ORC_PI * cf0 * q);
        cl = (1.0 - 2.0 * q1) + 2.0 * q1 * cos(2.0 * ORC_PI * q * cf0);
      }
      re[i] = re[i] * sl * cl / n;
    }
    orc_fft_c2r(re, zero, n, wave);
    for (int i = 0; i <= h; ++i) sp[(size_t)f * (h + 1) + i] = exp(wave[i]);
  }
  free(wave); free(re); free(im); free(zero); rng_free(&rng);
}

/* ------------------------------------------------------------------------ */
/* A1-A8: D4C -- d4c.cpp                                                      */
/* ------------------------------------------------------------------------ */
static int cmp_double(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

/* GetCentroid -- d4c.cpp:90-119 */
static void d4c_centroid(const double *x, int x_length, int fs, double f0, int n,
                         double pos, rng_cursor *rng, double *cent) {
  int h = n / 2;
  double *wave = dalloc((size_t)n), *r1 = dalloc((size_t)h + 1), *i1 = dalloc((size_t)h + 1),
         *r2 = dalloc((size_t)h + 1), *i2 = dalloc((size_t)h + 1);
  windowed_waveform(x, x_length, fs, f0, pos, 2, 4.0, 0, rng, wave);
  int lim = orc_matlab_round(2.0 * fs / f0) * 2;
  double power = 0.0;
  for (int i = 0; i <= lim; ++i) power += wave[i] * wave[i];
  for (int i = 0; i <= lim; ++i) wave[i] /= sqrt(power);
  orc_fft_r2c(wave, n, r1, i1);
  for (int i = 0; i < n; ++i) wave[i] *= i + 1.0;
  orc_fft_r2c(wave, n, r2, i2);
  for (int i = 0; i <= h; ++i) cent[i] = r2[i] * r1[i] + i1[i] * i2[i];
  free(wave); free(r1); free(i1); free(r2); free(i2);
}

/* D4CGeneralBody -- d4c.cpp:290-316 */
static void d4c_frame(const double *x, int x_length, int fs, double f0, int n,
                      double pos, int nap, const double *win, int wlen,
                      rng_cursor *rng, double *coarse) {
  int h = n / 2;
  double *c1 = dalloc((size_t)h + 1), *c2 = dalloc((size_t)h + 1), *sc = dalloc((size_t)h + 1);
  /* GetStaticCentroid -- :125-142 */
  d4c_centroid(x, x_length, fs, f0, n, pos - 0.25 / f0, rng, c1);
  d4c_centroid(x, x_length, fs, f0, n, pos + 0.25 / f0, rng, c2);
  for (int i = 0; i <= h; ++i) sc[i] = c1[i] + c2[i];
  orc_dc_correction(sc, f0, fs, n, sc);
  /* GetSmoothedPowerSpectrum -- :148-164 */
  double *wave = dalloc((size_t)n), *re = dalloc((size_t)h + 1), *im = dalloc((size_t)h + 1),
         *pw = dalloc((size_t)h + 1);
  windowed_waveform(x, x_length, fs, f0, pos, 1, 4.0, 0, rng, wave);
  orc_fft_r2c(wave, n, re, im);
  for (int i = 0; i <= h; ++i) pw[i] = re[i] * re[i] + im[i] * im[i];
  orc_dc_correction(pw, f0, fs, n, pw);
  orc_linear_smoothing(pw, f0, fs, n, pw);
  /* GetStaticGroupDelay -- :170-186 */
  double *gd = dalloc((size_t)h + 1), *sg = dalloc((size_t)h + 1);
  for (int i = 0; i <= h; ++i) gd[i] = sc[i] / pw[i];
  orc_linear_smoothing(gd, f0 / 2.0, fs, n, gd);
  orc_linear_smoothing(gd, f0, fs, n, sg);
  for (int i = 0; i <= h; ++i) gd[i] -= sg[i];
  /* GetCoarseAperiodicity -- :192-223 */
  int bnd = orc_matlab_round(n * 8.0 / wlen);
  int hw = wlen / 2;
  memset(wave, 0, sizeof(double) * (size_t)n);
  for (int b = 0; b < nap; ++b) {
    int center = (int)(3000.0 * (b + 1) * n / fs);
    for (int j = 0; j <= hw * 2; ++j) wave[j] = gd[center - hw + j] * win[j];
    orc_fft_r2c(wave, n, re, im);
    for (int j = 0; j <= h; ++j) pw[j] = re[j] * re[j] + im[j] * im[j];
    qsort(pw, (size_t)h + 1, sizeof(double), cmp_double);
    for (int j = 1; j <= h; ++j) pw[j] += pw[j - 1];
    coarse[b] = 10 * log10(pw[h - bnd - 1] / pw[h]);
  }
  for (int b = 0; b < nap; ++b) {                              /* :309-311 */
    double v = coarse[b] + (f0 - 100) / 50.0;
    coarse[b] = 0.0 < v ? 0.0 : v;                             /* MyMinDouble(0.0, v), common.h:80: a NaN stays */
  }
  free(c1); free(c2); free(sc); free(wave); free(re); free(im); free(pw); free(gd); free(sg);
}

void orc_d4c(const double *x, int x_length, int fs, const double *t,
             const double *f0, int nf, int fft_size, double threshold, double *ap) {
  int hb = fft_size / 2 + 1;
  rng_cursor rng; rng_init(&rng);                                /* reseed :340 */
  for (size_t i = 0; i < (size_t)nf * hb; ++i) ap[i] = 1.0 - ORC_SAFE;     /* :318-323 */
  int n = (int)pow(2.0, 1.0 + (int)(log(4.0 * fs / 47.0 + 1) / ORC_LOG2));
  double lim = fs / 2.0 - 3000.0;
  int nap = (int)((15000.0 < lim ? 15000.0 : lim) / 3000.0);
  int wlen = (int)(3000.0 * n / fs) * 2 + 1;
  double *win = dalloc((size_t)wlen);
  orc_nuttall(wlen, win);

  /* D4CLoveTrain -- :225-282 */
  double *ap0 = dalloc((size_t)nf);
  {
    int ln = (int)pow(2.0, 1.0 + (int)(log(3.0 * fs / 40.0 + 1) / ORC_LOG2));
    int lh = ln / 2;
    int b0 = (int)ceil(100.0 * ln / fs), b1 = (int)ceil(4000.0 * ln / fs),
        b2 = (int)ceil(7900.0 * ln / fs);
    double *wave = dalloc((size_t)ln), *re = dalloc((size_t)lh + 1), *im = dalloc((size_t)lh + 1),
           *pw = dalloc((size_t)ln);
    for (int f = 0; f < nf; ++f) {
      if (f0[f] == 0.0) { ap0[f] = 0.0; continue; }
      double cf0 = f0[f] > 40.0 ? f0[f] : 40.0;
      int len = windowed_waveform(x, x_length, fs, cf0, t[f], 2, 3.0, 0, &rng, wave);
      for (int i = len; i < ln; ++i) wave[i] = 0.0;
      orc_fft_r2c(wave, ln, re, im);
      for (int i = 0; i <= b0; ++i) pw[i] = 0.0;
      for (int i = b0 + 1; i < lh + 1; ++i) pw[i] = re[i] * re[i] + im[i] * im[i];
      for (int i = b0; i <= b2; ++i) pw[i] += pw[i - 1];
      ap0[f] = pw[b1] / pw[b2];
    }
    free(wave); free(re); free(im); free(pw);
  }

  double *coarse = dalloc((size_t)nap + 2), *caxis = dalloc((size_t)nap + 2);
  coarse[0] = -60.0;
  coarse[nap + 1] = -ORC_SAFE;
  for (int i = 0; i <= nap; ++i) caxis[i] = i * 3000.0;
  caxis[nap + 1] = fs / 2.0;
  double *faxis = dalloc((size_t)hb);
  for (int i = 0; i < hb; ++i) faxis[i] = (double)i * fs / fft_size;

  for (int f = 0; f < nf; ++f) {
    if (f0[f] == 0 || ap0[f] <= threshold) continue;
    double cf0 = f0[f] > 47.0 ? f0[f] : 47.0;
    d4c_frame(x, x_length, fs, cf0, n, t[f], nap, win, wlen, &rng, &coarse[1]);
    /* GetAperiodicity -- :325-333 */
    double *row = ap + (size_t)f * hb;
    orc_interp1(caxis, coarse, nap + 2, faxis, hb, row);
    for (int i = 0; i < hb; ++i) row[i] = pow(10.0, row[i] / 20.0);
  }
  free(win); free(ap0); free(coarse); free(caxis); free(faxis); rng_free(&rng);
}

/* ------------------------------------------------------------------------ */
/* Y1-Y7: Synthesis -- synthesis.cpp                                          */
/* ------------------------------------------------------------------------ */
static double safe_ap(double v) {                                  /* common.h:111-113 */
  double m = 0.999999999999 < v ? 0.999999999999 : v;
  return 0.001 > m ? 0.001 : m;
}

void orc_synthesis(const double *f0, int nf, const double *sp, const double *ap,
                   int fft_size, double frame_period, int fs, int y_length, double *y) {
  int n = fft_size, h = n / 2, hb = h + 1;
  rng_cursor rng; rng_init(&rng);                                 /* reseed :341 */
  for (int i = 0; i < y_length; ++i) y[i] = 0.0;
  double fp = frame_period / 1000.0;
  double lowest_f0 = fs / fft_size + 1.0;                          /* integer division, :359 */

  /* GetTimeBase -- :287-320 */
  double *taxis = dalloc((size_t)y_length), *ct = dalloc((size_t)nf + 1),
         *cf0 = dalloc((size_t)nf + 1), *cv = dalloc((size_t)nf + 1);
  for (int i = 0; i < y_length; ++i) taxis[i] = i / (double)fs;
  for (int i = 0; i < nf; ++i) {                                   /* :223-240 */
    ct[i] = i * fp;
    cf0[i] = f0[i] < lowest_f0 ? 0.0 : f0[i];
    cv[i] = cf0[i] == 0.0 ? 0.0 : 1.0;
  }
  ct[nf] = nf * fp;
  cf0[nf] = cf0[nf - 1] * 2 - cf0[nf - 2];
  cv[nf] = cv[nf - 1] * 2 - cv[nf - 2];
  double *if0 = dalloc((size_t)y_length), *vuv = dalloc((size_t)y_length);
  orc_interp1(ct, cf0, nf + 1, taxis, y_length, if0);
  orc_interp1(ct, cv, nf + 1, taxis, y_length, vuv);
  for (int i = 0; i < y_length; ++i) {
    vuv[i] = vuv[i] > 0.5 ? 1.0 : 0.0;
    if0[i] = vuv[i] == 0.0 ? ORC_DEFAULT_F0 : if0[i];
  }
  /* GetPulseLocationsForTimeBase -- :242-285 */
  double *wrap = dalloc((size_t)y_length);
  {
    double total = 2.0 * ORC_PI * if0[0] / fs;
    wrap[0] = fmod(total, 2.0 * ORC_PI);
    for (int i = 1; i < y_length; ++i) {
      total = total + 2.0 * ORC_PI * if0[i] / fs;
      wrap[i] = fmod(total, 2.0 * ORC_PI);
    }
  }
  double *ploc = dalloc((size_t)y_length), *pshift = dalloc((size_t)y_length);
  int *pidx = ialloc((size_t)y_length);
  int np = 0;
  for (int i = 0; i < y_length - 1; ++i) {
    if (fabs(wrap[i + 1] - wrap[i]) > ORC_PI) {
      ploc[np] = taxis[i];
      pidx[np] = i;
      double y1 = wrap[i] - 2.0 * ORC_PI, y2 = wrap[i + 1];
      pshift[np] = (-y1 / (y2 - y1)) / fs;
      ++np;
    }
  }

  /* GetDCRemover -- :322-334 */
  double *dcr = dalloc((size_t)n);
  {
    double dc = 0.0;
    for (int i = 0; i < h; ++i) {
      dcr[i] = 0.5 - 0.5 * cos(2.0 * ORC_PI * (i + 1.0) / (1.0 + n));
      dcr[n - i - 1] = dcr[i];
      dc += dcr[i] * 2.0;
    }
    for (int i = 0; i < h; ++i) { dcr[i] /= dc; dcr[n - i - 1] = dcr[i]; }
  }

  double *env = dalloc((size_t)hb), *ratio = dalloc((size_t)hb), *ls = dalloc((size_t)hb);
  double *mr = dalloc((size_t)hb), *mi = dalloc((size_t)hb), *sr = dalloc((size_t)hb),
         *si = dalloc((size_t)hb), *nr = dalloc((size_t)hb), *ni = dalloc((size_t)hb);
  double *tmp = dalloc((size_t)n), *per = dalloc((size_t)n), *aper = dalloc((size_t)n),
         *noise = dalloc((size_t)n);

  for (int p = 0; p < np; ++p) {
    int noise_size = pidx[imin(np - 1, p + 1)] - pidx[p];         /* :369 */
    double cvuv = vuv[pidx[p]];
    double ctime = ploc[p];
    /* GetSpectralEnvelope / GetAperiodicRatio -- :140-178 */
    int ff = imin(nf - 1, (int)floor(ctime / fp));
    int fc = imin(nf - 1, (int)ceil(ctime / fp));
    double w = ctime / fp - ff;
    const double *s0 = sp + (size_t)ff * hb, *s1 = sp + (size_t)fc * hb;
    const double *a0 = ap + (size_t)ff * hb, *a1 = ap + (size_t)fc * hb;
    if (ff == fc) {
      for (int i = 0; i < hb; ++i) { env[i] = fabs(s0[i]); ratio[i] = pow(safe_ap(a0[i]), 2.0); }
    } else {
      for (int i = 0; i < hb; ++i) {
        env[i] = (1.0 - w) * fabs(s0[i]) + w * fabs(s1[i]);
        ratio[i] = pow((1.0 - w) * safe_ap(a0[i]) + w * safe_ap(a1[i]), 2.0);
      }
    }
    /* GetPeriodicResponse -- :105-138 */
    if (cvuv <= 0.5 || ratio[0] > 0.999) {
      for (int i = 0; i < n; ++i) per[i] = 0.0;
    } else {
      for (int i = 0; i < hb; ++i) ls[i] = log(env[i] * (1.0 - ratio[i]) + ORC_SAFE) / 2.0;
      orc_min_phase(ls, n, mr, mi);
      double coef = 2.0 * ORC_PI * pshift[p] * fs / n;
      for (int i = 0; i < hb; ++i) {                               /* :88-100 */
        double re2 = cos(coef * i);
        double im2 = sqrt(1.0 - re2 * re2);
        sr[i] = mr[i] * re2 + mi[i] * im2;
        si[i] = mi[i] * re2 - mr[i] * im2;
      }
      orc_fft_c2r(sr, si, n, tmp);
      for (int i = 0; i < h; ++i) { per[i] = tmp[i + h]; per[i + h] = tmp[i]; }  /* fftshift */
      double dc = 0.0;                                             /* :73-82 */
      for (int i = h; i < n; ++i) dc += per[i];
      for (int i = 0; i < h; ++i) per[i] = -dc * dcr[i];            /* overwrite (quirk) */
      for (int i = h; i < n; ++i) per[i] -= dc * dcr[i];
    }
    /* GetAperiodicResponse -- :38-68 (noise :19-33) */
    {
      double avg = 0.0;
      for (int i = 0; i < noise_size; ++i) { noise[i] = rng_next(&rng); avg += noise[i]; }
      avg /= noise_size;
      for (int i = 0; i < noise_size; ++i) noise[i] -= avg;
      for (int i = noise_size > 0 ? noise_size : 0; i < n; ++i) noise[i] = 0.0;
      orc_fft_r2c(noise, n, nr, ni);
      if (cvuv != 0.0) for (int i = 0; i < hb; ++i) ls[i] = log(env[i] * ratio[i]) / 2.0;
      else for (int i = 0; i < hb; ++i) ls[i] = log(env[i]) / 2.0;
      orc_min_phase(ls, n, mr, mi);
      for (int i = 0; i < hb; ++i) {
        sr[i] = mr[i] * nr[i] - mi[i] * ni[i];
        si[i] = mr[i] * ni[i] + mi[i] * nr[i];
      }
      orc_fft_c2r(sr, si, n, tmp);
      for (int i = 0; i < h; ++i) { aper[i] = tmp[i + h]; aper[i + h] = tmp[i]; }
    }
    double sq = sqrt((double)noise_size);
    for (int j = 0; j < n; ++j) {                                  /* :211-215, :378-383 */
      int idx = j + pidx[p] - h + 1;
      if (idx < 0 || idx > y_length - 1) continue;
      y[idx] += (per[j] * sq + aper[j]) / n;
    }
  }

  free(taxis); free(ct); free(cf0); free(cv); free(if0); free(vuv); free(wrap);
  free(ploc); free(pshift); free(pidx); free(dcr); free(env); free(ratio); free(ls);
  free(mr); free(mi); free(sr); free(si); free(nr); free(ni);
  free(tmp); free(per); free(aper); free(noise); rng_free(&rng);
}

/* ------------------------------------------------------------------------ */
/* H1-H6: Harvest -- harvest.cpp (restated in world_oracle_harvest.c)         */
/* ------------------------------------------------------------------------ */
int orc_harvest_samples(int fs, int x_length, double frame_period) {  /* harvest.cpp:1219-1221 */
  return (int)(1000.0 * x_length / fs / frame_period) + 1;
}

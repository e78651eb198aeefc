/*
 * world_oracle_harvest.c -- Harvest F0 estimator, CPU parity oracle (plain C99).
 *
 * TEST INFRASTRUCTURE ONLY (see world_oracle.h).  From-scratch restatement of
 * externs/WORLD_v2/src/harvest.cpp; every block cites the lines it follows.
 *
 * Two places of the reference read uninitialised heap memory; the oracle defines them as zero:
 *   - RemoveUnreliableCandidates copies rows 1..T-2 into its scratch but reads rows 0 and T-1
 *     (harvest.cpp:672-684);
 *   - FixStep1 never writes f0_step1[i] where f0_base[i] == 0 (harvest.cpp:710-722).
 */
#include "world_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define H_PI 3.1415926535897932384
#define H_LOG2 0.69314718055994529
#define H_SAFE 0.000000000001

static double *dz(size_t n) {
  double *p = (double *)calloc(n ? n : 1, sizeof(double));
  if (!p) abort();
  return p;
}
static int *iz(size_t n) {
  int *p = (int *)calloc(n ? n : 1, sizeof(int));
  if (!p) abort();
  return p;
}
static int mini(int a, int b) { return a < b ? a : b; }
static int maxi(int a, int b) { return a > b ? a : b; }

/* ZeroCrossingEngine -- harvest.cpp:162-197 */
static int h_zero_cross(const double *s, int len, double fs, double *loc, double *itv) {
  int *edge = iz((size_t)len);
  int cnt = 0;
  for (int i = 0; i < len - 1; ++i)
    if (0.0 < s[i] && s[i + 1] <= 0.0) edge[cnt++] = i + 1;
  if (cnt < 2) { free(edge); return 0; }
  double *fine = dz((size_t)cnt);
  for (int i = 0; i < cnt; ++i) fine[i] = edge[i] - s[edge[i] - 1] / (s[edge[i]] - s[edge[i] - 1]);
  for (int i = 0; i < cnt - 1; ++i) {
    itv[i] = fs / (fine[i + 1] - fine[i]);
    loc[i] = (fine[i] + fine[i + 1]) / 2.0 / fs;
  }
  free(fine); free(edge);
  return cnt - 1;
}

/* one channel: GetFilteredSignal :99-148, GetFourZeroCrossingIntervals :206-238,
 * GetF0CandidateContour(+Sub) :240-293 */
static void h_channel(double bf, int fftn, double fs, const double *Yr, const double *Yi, int ylen,
                      double f0_floor, double f0_ceil, const double *t, int nf, double *out) {
  int h = fftn / 2;
  int half = orc_matlab_round(fs / bf * 2.0);
  double *flt = dz((size_t)fftn), *Wr = dz((size_t)h + 1), *Wi = dz((size_t)h + 1);
  orc_nuttall(half * 2 + 1, flt);
  for (int i = -half; i <= half; ++i) flt[i + half] *= cos(2 * H_PI * bf * i / fs);
  for (int i = half * 2 + 1; i < fftn; ++i) flt[i] = 0.0;
  orc_fft_r2c(flt, fftn, Wr, Wi);
  for (int i = 0; i <= h; ++i) {
    double tr = Yr[i] * Wr[i] - Yi[i] * Wi[i];
    Wi[i] = Yr[i] * Wi[i] + Yi[i] * Wr[i];
    Wr[i] = tr;
  }
  orc_fft_c2r(Wr, Wi, fftn, flt);
  int bias = half + 1;
  for (int i = 0; i < ylen; ++i) flt[i] = flt[i + bias];

  double *loc[4], *itv[4];
  int n[4];
  for (int k = 0; k < 4; ++k) { loc[k] = dz((size_t)ylen); itv[k] = dz((size_t)ylen); }
  n[0] = h_zero_cross(flt, ylen, fs, loc[0], itv[0]);
  for (int i = 0; i < ylen; ++i) flt[i] = -flt[i];
  n[1] = h_zero_cross(flt, ylen, fs, loc[1], itv[1]);
  for (int i = 0; i < ylen - 1; ++i) flt[i] = flt[i] - flt[i + 1];
  n[2] = h_zero_cross(flt, ylen - 1, fs, loc[2], itv[2]);
  for (int i = 0; i < ylen - 1; ++i) flt[i] = -flt[i];
  n[3] = h_zero_cross(flt, ylen - 1, fs, loc[3], itv[3]);

  if (n[0] > 2 && n[1] > 2 && n[2] > 2 && n[3] > 2) {
    double *ip[4];
    for (int k = 0; k < 4; ++k) { ip[k] = dz((size_t)nf); orc_interp1(loc[k], itv[k], n[k], t, nf, ip[k]); }
    double upper = bf * 1.1, lower = bf * 0.9;
    for (int i = 0; i < nf; ++i) {
      double c = (ip[0][i] + ip[1][i] + ip[2][i] + ip[3][i]) / 4.0;
      if (c > upper || c < lower || c > f0_ceil || c < f0_floor) c = 0.0;
      out[i] = c;
    }
    for (int k = 0; k < 4; ++k) free(ip[k]);
  } else {
    for (int i = 0; i < nf; ++i) out[i] = 0.0;
  }
  for (int k = 0; k < 4; ++k) { free(loc[k]); free(itv[k]); }
  free(flt); free(Wr); free(Wi);
}

/* GetRefinedF0 / GetMeanF0 / FixF0 -- harvest.cpp:434-617 */
static void h_refine(const double *x, int xl, double fs, double pos, double f0, double f0_floor,
                     double f0_ceil, double *rf0, double *rscore) {
  if (f0 <= 0.0) { *rf0 = 0.0; *rscore = 0.0; return; }
  int hw = (int)(1.5 * fs / f0 + 1.0);
  int len = hw * 2 + 1;
  double wlen = (2.0 * hw + 1.0) / fs;
  int fftn = (int)pow(2.0, 2.0 + (int)(log(hw * 2.0 + 1.0) / H_LOG2));
  int h = fftn / 2;
  double bt0 = (-hw + 0) / fs;
  int basic = orc_matlab_round((pos + bt0) * fs + 0.001);              /* :434-441 */
  double *mw = dz((size_t)len), *dw = dz((size_t)len), *buf = dz((size_t)fftn);
  double *mr = dz((size_t)h + 1), *mi = dz((size_t)h + 1), *dr = dz((size_t)h + 1), *di = dz((size_t)h + 1);
  for (int i = 0; i < len; ++i) {                                      /* :446-456 */
    double tm = ((basic + i) - 1.0) / fs - pos;
    mw[i] = 0.42 + 0.5 * cos(2.0 * H_PI * tm / wlen) + 0.08 * cos(4.0 * H_PI * tm / wlen);
  }
  dw[0] = -mw[1] / 2.0;                                                /* :462-468 */
  for (int i = 1; i < len - 1; ++i) dw[i] = -(mw[i + 1] - mw[i - 1]) / 2.0;
  dw[len - 1] = mw[len - 2] / 2.0;
  for (int i = 0; i < len; ++i) buf[i] = x[maxi(0, mini(xl - 1, basic + i - 1))] * mw[i];   /* :474-505 */
  orc_fft_r2c(buf, fftn, mr, mi);
  for (int i = 0; i < len; ++i) buf[i] = x[maxi(0, mini(xl - 1, basic + i - 1))] * dw[i];
  orc_fft_r2c(buf, fftn, dr, di);
  int nh = mini((int)(fs / 2.0 / f0), 6);                              /* :571-572 */
  double numer = 0.0, denom = 0.0, sc = 0.0;                           /* FixF0 :507-536 */
  for (int i = 0; i < nh; ++i) {
    int idx = orc_matlab_round(f0 * fftn / fs * (i + 1));
    double p = 0.0, nm = 0.0;
    if (idx <= h) { p = mr[idx] * mr[idx] + mi[idx] * mi[idx]; nm = mr[idx] * di[idx] - mi[idx] * dr[idx]; }
    double inst = p == 0.0 ? 0.0 : (double)idx * fs / fftn + nm / p * fs / 2.0 / H_PI;
    double amp = sqrt(p);
    numer += amp * inst;
    denom += amp * (i + 1.0);
    sc += fabs((inst / (i + 1.0) - f0) / f0);
  }
  *rf0 = numer / (denom + H_SAFE);
  *rscore = 1.0 / (sc / nh + H_SAFE);
  if (*rf0 < f0_floor || *rf0 > f0_ceil || *rscore < 2.5) { *rf0 = 0.0; *rscore = 0.0; }   /* :610-614 */
  free(mw); free(dw); free(buf); free(mr); free(mi); free(dr); free(di);
}

/* SelectBestF0 -- harvest.cpp:636-650 */
static double h_select(double ref, const double *c, int n, double allowed, double *best_err) {
  double best = 0.0;
  *best_err = allowed;
  for (int i = 0; i < n; ++i) {
    double e = fabs(ref - c[i]) / ref;
    if (e > *best_err) continue;
    best = c[i];
    *best_err = e;
  }
  return best;
}

/* GetBoundaryList -- harvest.cpp:727-743 */
static int h_boundaries(const double *f0, int n, int *list) {
  int cnt = 0;
  int prev = 0;                                   /* vuv[0] = 0 */
  for (int i = 1; i < n; ++i) {
    int v = (i == n - 1) ? 0 : (f0[i] > 0 ? 1 : 0);
    if (v - prev != 0) { list[cnt] = i - cnt % 2; cnt++; }
    prev = v;
  }
  return cnt;
}

/* ExtendF0 -- harvest.cpp:791-820 */
static int h_extend_f0(int origin, int last_point, int shift, double **cand, int ncand, double allowed,
                       double *ext) {
  double tmp_f0 = ext[origin];
  int shifted_origin = origin;
  int distance = abs(last_point - origin);
  int count = 0;
  double dummy;
  for (int i = 0; i <= distance; ++i) {
    int idx = origin + shift * i;
    ext[idx + shift] = h_select(tmp_f0, cand[idx + shift], ncand, allowed, &dummy);
    if (ext[idx + shift] == 0.0) {
      count++;
    } else {
      tmp_f0 = ext[idx + shift];
      count = 0;
      shifted_origin = idx + shift;
    }
    if (count == 4) break;
  }
  return shifted_origin;
}

/* SearchScore -- harvest.cpp:901-907 */
static double h_search_score(double f0, const double *c, const double *s, int n) {
  double score = 0.0;
  for (int i = 0; i < n; ++i)
    if (f0 == c[i] && score < s[i]) score = s[i];
  return score;
}

/* FilteringF0 -- harvest.cpp:1049-1074 */
static void h_filtering(const double *a, const double *b, double *x, int n, int st, int ed, double *y) {
  double w0 = 0.0, w1 = 0.0;
  double *tmp = dz((size_t)n);
  for (int i = 0; i < st; ++i) x[i] = x[st];
  for (int i = ed + 1; i < n; ++i) x[i] = x[ed];
  for (int i = 0; i < n; ++i) {
    double wt = x[i] + a[0] * w0 + a[1] * w1;
    tmp[n - i - 1] = b[0] * wt + b[1] * w0 + b[0] * w1;
    w1 = w0; w0 = wt;
  }
  w0 = w1 = 0.0;
  for (int i = 0; i < n; ++i) {
    double wt = tmp[i] + a[0] * w0 + a[1] * w1;
    y[n - i - 1] = b[0] * wt + b[1] * w0 + b[0] * w1;
    w1 = w0; w0 = wt;
  }
  free(tmp);
}

/* HarvestGeneralBody -- harvest.cpp:1145-1215 */
static void h_body(const double *x, int x_length, int fs, int frame_period, double f0_floor,
                   double f0_ceil, double ch_oct, int speed, double *t, double *f0) {
  double adj_floor = f0_floor * 0.9, adj_ceil = f0_ceil * 1.1;
  int nch = 1 + (int)(log(adj_ceil / adj_floor) / H_LOG2 * ch_oct);
  double *bnd = dz((size_t)nch);
  for (int i = 0; i < nch; ++i) bnd[i] = adj_floor * pow(2.0, (i + 1) / ch_oct);
  int r = maxi(mini(speed, 12), 1);
  int ylen = (int)ceil((double)x_length / r);
  double afs = (double)fs / r;
  int fftn = orc_suitable_fft_size(ylen + 5 + 2 * (int)(2.0 * afs / bnd[0]));
  int h = fftn / 2;

  /* GetWaveformAndSpectrum(+Sub) -- :43-93 */
  double *y = dz((size_t)fftn);
  if (r == 1) {
    for (int i = 0; i < x_length; ++i) y[i] = x[i];
  } else {
    int lag = (int)(ceil(140.0 / r) * r);
    int nlen = x_length + lag * 2;
    double *nx = dz((size_t)nlen), *ny = dz((size_t)nlen);
    for (int i = 0; i < lag; ++i) nx[i] = x[0];
    for (int i = lag; i < lag + x_length; ++i) nx[i] = x[i - lag];
    for (int i = lag + x_length; i < nlen; ++i) nx[i] = x[x_length - 1];
    orc_decimate(nx, nlen, r, ny);
    for (int i = 0; i < ylen; ++i) y[i] = ny[lag / r + i];
    free(nx); free(ny);
  }
  double mean = 0.0;
  for (int i = 0; i < ylen; ++i) mean += y[i];
  mean /= ylen;
  for (int i = 0; i < ylen; ++i) y[i] -= mean;
  for (int i = ylen; i < fftn; ++i) y[i] = 0.0;
  double *Yr = dz((size_t)h + 1), *Yi = dz((size_t)h + 1);
  orc_fft_r2c(y, fftn, Yr, Yi);

  int nf = orc_harvest_samples(fs, x_length, frame_period);
  for (int i = 0; i < nf; ++i) { t[i] = i * frame_period / 1000.0; f0[i] = 0.0; }

  int overlap = 7;
  int max_cand = orc_matlab_round(nch / 10.0) * overlap;
  double **cand = (double **)malloc(sizeof(double *) * (size_t)nf);
  double **score = (double **)malloc(sizeof(double *) * (size_t)nf);
  for (int i = 0; i < nf; ++i) { cand[i] = dz((size_t)max_cand); score[i] = dz((size_t)max_cand); }

  /* HarvestGeneralBodySub -- :1118-1140; GetRawF0Candidates :334-343 */
  double **raw = (double **)malloc(sizeof(double *) * (size_t)nch);
  for (int c = 0; c < nch; ++c) {
    raw[c] = dz((size_t)nf);
    h_channel(bnd[c], fftn, afs, Yr, Yi, ylen, f0_floor, f0_ceil, t, nf, raw[c]);
  }
  /* DetectOfficialF0Candidates(+Sub1/2) -- :348-412 */
  int ncand1 = 0;
  {
    int *vuv = iz((size_t)nch), *st = iz((size_t)nch), *ed = iz((size_t)nch);
    for (int i = 0; i < nf; ++i) {
      for (int j = 0; j < nch; ++j) vuv[j] = raw[j][i] > 0 ? 1 : 0;
      vuv[0] = vuv[nch - 1] = 0;
      int nsec = 0;
      for (int j = 1; j < nch; ++j) {
        int d = vuv[j] - vuv[j - 1];
        if (d == 1) st[nsec] = j;
        if (d == -1) ed[nsec++] = j;
      }
      int k = 0;
      for (int s = 0; s < nsec; ++s) {
        if (ed[s] - st[s] < 10) continue;
        double tmp = 0.0;
        for (int j = st[s]; j < ed[s]; ++j) tmp += raw[j][i];
        tmp /= (ed[s] - st[s]);
        cand[i][k++] = tmp;
      }
      for (int j = k; j < max_cand; ++j) cand[i][j] = 0.0;
      ncand1 = maxi(ncand1, k);
    }
    free(vuv); free(st); free(ed);
  }
  /* OverlapF0Candidates -- :417-429 */
  for (int i = 1; i <= 3; ++i)
    for (int j = 0; j < ncand1; ++j) {
      for (int k = i; k < nf; ++k) cand[k][j + ncand1 * i] = cand[k - i][j];
      for (int k = 0; k < nf - i; ++k) cand[k][j + ncand1 * (i + 3)] = cand[k + i][j];
    }
  int ncand = ncand1 * overlap;

  /* RefineF0Candidates -- :622-631 */
  for (int i = 0; i < nf; ++i)
    for (int j = 0; j < ncand; ++j)
      h_refine(y, ylen, afs, t[i], cand[i][j], f0_floor, f0_ceil, &cand[i][j], &score[i][j]);

  /* RemoveUnreliableCandidates(+Sub) -- :652-688 (rows 0 and nf-1 of the scratch read as zero) */
  {
    double **tmp = (double **)malloc(sizeof(double *) * (size_t)nf);
    for (int i = 0; i < nf; ++i) tmp[i] = dz((size_t)(ncand ? ncand : 1));
    for (int i = 1; i < nf - 1; ++i) memcpy(tmp[i], cand[i], sizeof(double) * (size_t)ncand);
    for (int i = 1; i < nf - 1; ++i)
      for (int j = 0; j < ncand; ++j) {
        double ref = cand[i][j];
        if (ref == 0) continue;
        double e1, e2;
        h_select(ref, tmp[i + 1], ncand, 1.0, &e1);
        h_select(ref, tmp[i - 1], ncand, 1.0, &e2);
        double me = e1 < e2 ? e1 : e2;
        if (me <= 0.05) continue;
        cand[i][j] = 0;
        score[i][j] = 0;
      }
    for (int i = 0; i < nf; ++i) free(tmp[i]);
    free(tmp);
  }

  /* FixF0Contour -- :1027-1044 */
  double *c1 = dz((size_t)nf), *c2 = dz((size_t)nf), *best = dz((size_t)nf);
  for (int i = 0; i < nf; ++i) {                                     /* SearchF0Base :693-705 */
    double bs = 0.0;
    c1[i] = 0.0;
    for (int j = 0; j < ncand; ++j)
      if (score[i][j] > bs) { c1[i] = cand[i][j]; bs = score[i][j]; }
  }
  for (int i = 2; i < nf; ++i) {                                     /* FixStep1 :710-722, allowed 0.008 */
    if (c1[i] == 0.0) continue;
    double ref = c1[i - 1] * 2 - c1[i - 2];
    c2[i] = (fabs((c1[i] - ref) / ref) > 0.008 && fabs((c1[i] - c1[i - 1])) / c1[i - 1] > 0.008) ? 0.0 : c1[i];
  }
  int *bl = iz((size_t)nf + 4);
  {                                                                  /* FixStep2 :748-762, minimum 6 */
    for (int i = 0; i < nf; ++i) c1[i] = c2[i];
    int nb = h_boundaries(c2, nf, bl);
    for (int i = 0; i < nb / 2; ++i) {
      if (bl[i * 2 + 1] - bl[i * 2] >= 6) continue;
      for (int j = bl[i * 2]; j <= bl[i * 2 + 1]; ++j) c1[j] = 0.0;
    }
  }
  {                                                                  /* FixStep3 :968-995, allowed 0.18 */
    for (int i = 0; i < nf; ++i) c2[i] = c1[i];
    int nb = h_boundaries(c1, nf, bl);
    int nsec = nb / 2;
    double **multi = (double **)malloc(sizeof(double *) * (size_t)(nsec ? nsec : 1));
    double **rows = (double **)malloc(sizeof(double *) * (size_t)(nsec ? nsec : 1));   /* for freeing */
    for (int i = 0; i < nsec; ++i) {                                 /* GetMultiChannelF0 :767-778 */
      multi[i] = rows[i] = dz((size_t)nf);
      for (int j = bl[i * 2]; j <= bl[i * 2 + 1]; ++j) multi[i][j] = c1[j];
    }
    for (int i = 0; i < nsec; ++i) {                                 /* Extend :861-878 (in place) */
      bl[i * 2 + 1] = h_extend_f0(bl[i * 2 + 1], mini(nf - 2, bl[i * 2 + 1] + 100), 1, cand, ncand, 0.18, multi[i]);
      bl[i * 2] = h_extend_f0(bl[i * 2], maxi(1, bl[i * 2] - 100), -1, cand, ncand, 0.18, multi[i]);
    }
    int nchn = 0;                                                    /* ExtendSub :840-856 */
    {
      double mean_f0 = 0.0;                                          /* not reset between sections (quirk) */
      for (int i = 0; i < nsec; ++i) {
        int st = bl[i * 2], ed = bl[i * 2 + 1];
        for (int j = st; j < ed; ++j) mean_f0 += multi[i][j];
        mean_f0 /= ed - st;
        if (2200.0 / mean_f0 < ed - st) {                            /* Swap :826-838 */
          double *tp = multi[nchn]; multi[nchn] = multi[i]; multi[i] = tp;
          int ti = bl[nchn * 2]; bl[nchn * 2] = bl[i * 2]; bl[i * 2] = ti;
          ti = bl[nchn * 2 + 1]; bl[nchn * 2 + 1] = bl[i * 2 + 1]; bl[i * 2 + 1] = ti;
          nchn++;
        }
      }
    }
    if (nchn != 0) {                                                 /* MergeF0 :937-963 */
      int *order = iz((size_t)nchn);
      for (int i = 0; i < nchn; ++i) order[i] = i;                   /* MakeSortedOrder :883-896 */
      for (int i = 1; i < nchn; ++i)
        for (int j = i - 1; j >= 0; --j) {
          if (bl[order[j] * 2] > bl[order[i] * 2]) { int tv = order[i]; order[i] = order[j]; order[j] = tv; }
          else break;
        }
      for (int i = 0; i < nf; ++i) c2[i] = multi[0][i];
      for (int i = 1; i < nchn; ++i) {
        int o = order[i];
        if (bl[o * 2] - bl[1] > 0) {
          for (int j = bl[o * 2]; j <= bl[o * 2 + 1]; ++j) c2[j] = multi[o][j];
          bl[0] = bl[o * 2];
          bl[1] = bl[o * 2 + 1];
        } else {                                                     /* MergeF0Sub :912-932 */
          int st1 = bl[0], ed1 = bl[1], st2 = bl[o * 2], ed2 = bl[o * 2 + 1];
          const double *f2 = multi[o];
          if (st1 <= st2 && ed1 >= ed2) {
            bl[1] = ed1;
          } else {
            double s1 = 0.0, s2 = 0.0;
            for (int k = st2; k <= ed1; ++k) {
              s1 += h_search_score(c2[k], cand[k], score[k], ncand);
              s2 += h_search_score(f2[k], cand[k], score[k], ncand);
            }
            if (s1 > s2) for (int k = ed1; k <= ed2; ++k) c2[k] = f2[k];
            else for (int k = st2; k <= ed2; ++k) c2[k] = f2[k];
            bl[1] = ed2;
          }
        }
      }
      free(order);
    }
    for (int i = 0; i < nsec; ++i) free(rows[i]);
    free(rows); free(multi);
  }
  {                                                                  /* FixStep4 :1000-1022, threshold 9 */
    for (int i = 0; i < nf; ++i) best[i] = c2[i];
    int nb = h_boundaries(c2, nf, bl);
    for (int i = 0; i < nb / 2 - 1; ++i) {
      int distance = bl[(i + 1) * 2] - bl[i * 2 + 1] - 1;
      if (distance >= 9) continue;
      double tmp0 = c2[bl[i * 2 + 1]] + 1, tmp1 = c2[bl[(i + 1) * 2]] - 1;
      double coef = (tmp1 - tmp0) / (distance + 1.0);
      int count = 1;
      for (int j = bl[i * 2 + 1] + 1; j <= bl[(i + 1) * 2] - 1; ++j) best[j] = tmp0 + coef * count++;
    }
  }
  {                                                                  /* SmoothF0Contour :1079-1113 */
    const double b[2] = {0.0078202080334971724, 0.015640416066994345};
    const double a[2] = {1.7347257688092754, -0.76600660094326412};
    int lag = 300, nn = nf + lag * 2;
    double *ctr = dz((size_t)nn);
    for (int i = 0; i < nf; ++i) ctr[lag + i] = best[i];
    int *bl2 = iz((size_t)nn + 4);
    int nb = h_boundaries(ctr, nn, bl2);
    int nsec = nb / 2;
    double **multi = (double **)malloc(sizeof(double *) * (size_t)(nsec ? nsec : 1));
    for (int i = 0; i < nsec; ++i) {
      multi[i] = dz((size_t)nn);
      for (int j = bl2[i * 2]; j <= bl2[i * 2 + 1]; ++j) multi[i][j] = ctr[j];
    }
    for (int i = 0; i < nsec; ++i) {
      h_filtering(a, b, multi[i], nn, bl2[i * 2], bl2[i * 2 + 1], ctr);
      for (int j = bl2[i * 2]; j <= bl2[i * 2 + 1]; ++j) f0[j - lag] = ctr[j];
    }
    for (int i = 0; i < nsec; ++i) free(multi[i]);
    free(multi); free(bl2); free(ctr);
  }

  for (int c = 0; c < nch; ++c) free(raw[c]);
  free(raw);
  for (int i = 0; i < nf; ++i) { free(cand[i]); free(score[i]); }
  free(cand); free(score); free(c1); free(c2); free(best); free(bl);
  free(y); free(Yr); free(Yi); free(bnd);
}

/* Harvest -- harvest.cpp:1223-1255 */
void orc_harvest(const double *x, int x_length, int fs, double f0_floor, double f0_ceil,
                 double frame_period, double *t, double *f0) {
  int ratio = orc_matlab_round(fs / 8000.0);
  if (frame_period == 1.0) {
    h_body(x, x_length, fs, 1, f0_floor, f0_ceil, 40, ratio, t, f0);
    return;
  }
  int bn = orc_harvest_samples(fs, x_length, 1);
  double *bf0 = dz((size_t)bn), *bt = dz((size_t)bn);
  h_body(x, x_length, fs, 1, f0_floor, f0_ceil, 40, ratio, bt, bf0);
  int nf = orc_harvest_samples(fs, x_length, frame_period);
  for (int i = 0; i < nf; ++i) {
    t[i] = i * frame_period / 1000.0;
    f0[i] = bf0[mini(bn - 1, orc_matlab_round(t[i] * 1000.0))];
  }
  free(bf0); free(bt);
}

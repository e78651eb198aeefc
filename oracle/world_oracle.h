/*
 * world_oracle.h -- CPU parity oracle for the WORLD analysis/synthesis hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a from-scratch plain-C restatement of the
 * reference algorithm (turbocast/HTS-train-WORLD, externs/WORLD_v2/src, all .cpp files).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it.  The product path (hts-train-world_amd/csrc, libworld_mi355.so) never
 * links, loads or calls anything in this directory.
 *
 * Parity pinning: the reference ships no golden vectors,
 * so this oracle is pinned against the real reference compiled from its own
 * sources into oracle/_ref/libworld_ref.so (oracle/Makefile, target `ref`):
 * tests/test_oracle_vs_ref.py compares every entry point below with the
 * reference on seeded inputs, and tests/golden/ holds vectors generated from
 * that reference build by oracle/gen_golden.py.
 *
 * Arrays are contiguous row-major (sp/ap: [frames][fft_size/2+1]) instead of
 * the reference's double** so that numpy/ctypes can hand them over directly.
 */
#ifndef WORLD_ORACLE_H_
#define WORLD_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- primitives (reference: matlabfunctions.cpp, common.cpp, fft.cpp) ---- */
void orc_randn_table_u32(uint32_t *out, int count);            /* matlabfunctions.cpp:247-277 */
void orc_randn_table(double *out, int count);
int  orc_matlab_round(double x);                                /* matlabfunctions.cpp:212-214 */
int  orc_suitable_fft_size(int sample);                         /* common.cpp:51-54 */
void orc_interp1(const double *x, const double *y, int n,
                 const double *xi, int m, double *yi);          /* matlabfunctions.cpp:136-182 */
void orc_interp1q(double x0, double dx, const double *y, int n,
                  const double *xi, int m, double *yi);         /* matlabfunctions.cpp:220-241 */
void orc_decimate(const double *x, int n, int r, double *y);    /* matlabfunctions.cpp:27-125,184-210 */
void orc_nuttall(int n, double *w);                             /* common.cpp:113-121 */
void orc_dc_correction(const double *in, double f0, int fs, int fft_size,
                       double *out);                            /* common.cpp:56-75 */
void orc_linear_smoothing(const double *in, double width, int fs,
                          int fft_size, double *out);           /* common.cpp:27-46,77-111 */
void orc_fft_r2c(const double *x, int n, double *re, double *im);   /* fft.cpp:48-59 semantics */
void orc_fft_c2r(const double *re, const double *im, int n, double *x); /* fft.cpp:27-35 semantics */
void orc_min_phase(const double *log_spec, int fft_size,
                   double *re, double *im);                     /* common.cpp:182-220 */

/* ---- DIO (dio.cpp) ---- */
int  orc_dio_samples(int fs, int x_length, double frame_period);    /* dio.cpp:638-640 */
void orc_dio(const double *x, int x_length, int fs, double f0_floor,
             double f0_ceil, double channels_in_octave, double frame_period,
             int speed, double allowed_range, double *t, double *f0);   /* dio.cpp:578-647 */

/* ---- StoneMask (stonemask.cpp) ---- */
void orc_stonemask(const double *x, int x_length, int fs, const double *t,
                   const double *f0, int nf, double *refined);     /* stonemask.cpp:184-217 */

/* ---- CheapTrick (cheaptrick.cpp) ---- */
int    orc_cheaptrick_fft_size(int fs, double f0_floor);           /* cheaptrick.cpp:191-194 */
double orc_cheaptrick_f0_floor(int fs, int fft_size);              /* cheaptrick.cpp:196-198 */
void   orc_cheaptrick(const double *x, int x_length, int fs, const double *t,
                      const double *f0, int nf, double q1, int fft_size,
                      double *sp);                                 /* cheaptrick.cpp:200-228 */

/* ---- D4C (d4c.cpp) ---- */
void orc_d4c(const double *x, int x_length, int fs, const double *t,
             const double *f0, int nf, int fft_size, double threshold,
             double *ap);                                          /* d4c.cpp:337-397 */

/* ---- Synthesis (synthesis.cpp) ---- */
void orc_synthesis(const double *f0, int nf, const double *sp,
                   const double *ap, int fft_size, double frame_period,
                   int fs, int y_length, double *y);               /* synthesis.cpp:338-397 */

/* ---- Harvest (harvest.cpp) ---- */
int  orc_harvest_samples(int fs, int x_length, double frame_period);   /* harvest.cpp:1219-1221 */
void orc_harvest(const double *x, int x_length, int fs, double f0_floor,
                 double f0_ceil, double frame_period, double *t,
                 double *f0);                                      /* harvest.cpp:1223-1255 */

/* ---- Feature codec (codec.cpp), world_oracle_codec.c; flat row-major arrays ---- */
int  orc_num_aperiodicities(int fs);                                       /* codec.cpp:212-215 */
void orc_code_aperiodicity(const double *ap, int nf, int fs, int fft_size, int nap,
                           double *coded);                                 /* codec.cpp:217-235 */
void orc_decode_aperiodicity(const double *coded, int nf, int fs, int nap, int fft_size,
                             double *ap);                                  /* codec.cpp:237-266 (definition's order) */
void orc_code_spectral_envelope(const double *sp, int nf, int fs, int fft_size, int ndim,
                                double *coded);                            /* codec.cpp:268-295 */
void orc_decode_spectral_envelope(const double *coded, int nf, int fs, int fft_size, int ndim,
                                  double *sp);                             /* codec.cpp:297-324 */

/* ---- cmp composition (data/scripts/window.pl, addhtkheader.pl), world_oracle_codec.c ---- */
void orc_window_stream(const float *in, int T, int dim, int nwin, const double *const *win,
                       float *out);                                        /* window.pl:45-146 */
/* decode side of the synth CLI's coded features (test/synth.cpp:151-256, test/sptkfunctions.cpp) */
void orc_freqt(const double *c1, int m1, double *c2, int m2, double a);            /* sptkfunctions.cpp:596-631 */
void orc_mgc2sp(const double *mgc, int m, double alpha, int fft_size, int nbins, double *x); /* :186-219, :256-274 */
void orc_recipe_decode(const float *lf0, const float *mgc, const float *bap, int nf, int fs, int fft_size,
                       int spec_dim, int ap_dim, double *f0, double *sp, double *ap);
void orc_htk_header(int nframes, int samprate, int frameshift, int bytes_per_frame, int type,
                    unsigned char *out12);                                 /* addhtkheader.pl:45-82 */

/* ---- vibrato (data/scripts/Extract.py:115-227; world_oracle_vibrato.c) -- PARITY UNPINNED ---- */
void orc_lowess(const double *y, int n, double frac, int it, double *fit);   /* statsmodels lowess on x = 0..n-1 */
int  orc_get_vibrate(const double *f, int n, double *t);                      /* Extract.py:115-158 */
int  orc_vibrato(const double *f0, int T, const int *seg_start, const int *seg_end, const double *seg_pitch,
                 int nseg, double *vib, double *df0);                         /* Extract.py:161-227 */
void orc_sopr_log(const double *in, int n, float *out);                       /* Extract.py:86-96 */

#ifdef __cplusplus
}
#endif
#endif  /* WORLD_ORACLE_H_ */

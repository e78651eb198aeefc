/*
 * world_mi355.h -- batched, device-resident extension of WORLD's C API for MI355X.
 *
 * The drop-in boundary of this repo is WORLD's own public C ABI (include/world/,
 * one header per reference header, same names/signatures/struct layouts).  Those
 * entry points take HOST pointers, one utterance per call, exactly like the
 * reference (externs/WORLD_v2/src/world/{dio,stonemask,cheaptrick,d4c,synthesis,harvest}.h).
 *
 * This header is the extension underneath them: a batch of utterances whose
 * waveforms and features stay in HBM, processed by the same kernels.  The
 * per-utterance functions are thin wrappers over a batch of one.  Everything is
 * plain C: opaque handles, device pointers as `double*`, sizes as int/int64_t,
 * int error codes (0 = ok) -- no torch, no C++ types.
 *
 * Layout of a batch (B utterances):
 *   x   : double[sum x_length]          concatenated waveforms, utterance u at x_offset[u]
 *   t,f0: double[sum f0_length]         frames concatenated, utterance u at frame_offset[u]
 *   sp,ap: double[sum f0_length][fft_size/2+1]   row-major, same frame order
 *   y   : double[sum y_length]          synthesised waveforms at y_offset[u]
 * with f0_length[u] = GetSamplesForDIO(fs, x_length[u], frame_period)  (dio.cpp:638-640)
 * and  y_length[u]  = int((f0_length-1)*frame_period/1000*fs)+1        (test/synth.cpp:259)
 * unless given explicitly.
 */
#ifndef WORLD_MI355_H_
#define WORLD_MI355_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libworld_mi355.so is built with -fvisibility=hidden; the declarations below are its exported ABI */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define WM_OK 0
#define WM_ERR_HIP 1              /* a HIP runtime call failed; see WorldMi355LastError() */
#define WM_ERR_BAD_ARG 2
#define WM_ERR_UNSUPPORTED_FFT 3  /* fft_size outside {512,1024,2048,4096}; D4C's own size outside {1024,2048,4096,8192} */
#define WM_ERR_NO_DEVICE 4        /* no HIP device: the product path never falls back to CPU */
#define WM_ERR_UNSUPPORTED 5
#define WM_ERR_IO 6                /* a file could not be written (WorldMi355WriteFiles); see WorldMi355LastError() */

typedef struct WorldMi355Context WorldMi355Context;
typedef struct WorldMi355Batch WorldMi355Batch;

/* All option fields of the reference's four option structs in one POD
 * (dio.h:16-23, cheaptrick.h:16-20, d4c.h:16-18, harvest.h:16-20). */
typedef struct {
  int fs;
  double frame_period;        /* ms */
  double f0_floor;            /* DIO / Harvest */
  double f0_ceil;
  double channels_in_octave;  /* DIO */
  int speed;                  /* DIO decimation ratio 1..12 */
  double allowed_range;       /* DIO */
  double q1;                  /* CheapTrick */
  int fft_size;               /* CheapTrick / D4C / Synthesis; 0 = GetFFTSizeForCheapTrick(fs) */
  double d4c_threshold;       /* D4C */
} WorldMi355Params;

/* Defaults as the reference's analysis CLI sets them (test/analysis.cpp:93-203):
 * floor 71, ceil 800, 2 ch/oct, speed 1, range 0.1, q1 -0.15, threshold 0 (NOT 0.85). */
void WorldMi355DefaultParams(int fs, double frame_period, WorldMi355Params* p);

/* The entry points of WORLD's own API (include/world/*.h) return void: a failure inside them (no HIP device, a HIP
 * error, an unsupported size) prints a message and calls abort(), as the reference aborts on bad_alloc.  A host that
 * would rather lose one utterance than the process installs a handler: it is called INSTEAD of abort() with the name
 * of the step that failed, a WM_ERR_* code and WorldMi355LastError()'s text, and the entry point then returns to its
 * caller with its outputs unspecified.  NULL restores abort().  The functions of this header return codes and never
 * call the handler. */
typedef void (*WorldMi355ErrorHandler)(const char* where, int code, const char* message, void* user);
void WorldMi355SetErrorHandler(WorldMi355ErrorHandler handler, void* user);

/* device < 0: current device.  stream == NULL: the legacy default stream (stream 0). */
int WorldMi355CreateContext(int device, void* hip_stream, WorldMi355Context** out);
void WorldMi355DestroyContext(WorldMi355Context* ctx);
int WorldMi355SetStream(WorldMi355Context* ctx, void* hip_stream);
int WorldMi355Synchronize(WorldMi355Context* ctx);
const char* WorldMi355LastError(void);

/* f0_lengths / y_lengths may be NULL (derived as documented above).  For a
 * synthesis-only batch pass x_lengths = NULL and give f0_lengths (+ y_lengths). */
int WorldMi355CreateBatch(WorldMi355Context* ctx, const WorldMi355Params* params, int n_utt,
                          const int* x_lengths, const int* f0_lengths, const int* y_lengths,
                          WorldMi355Batch** out);
void WorldMi355DestroyBatch(WorldMi355Batch* b);
int64_t WorldMi355BatchTotalSamples(const WorldMi355Batch* b);
int64_t WorldMi355BatchTotalFrames(const WorldMi355Batch* b);
int64_t WorldMi355BatchTotalOutputSamples(const WorldMi355Batch* b);
int WorldMi355BatchFftSize(const WorldMi355Batch* b);
const int64_t* WorldMi355BatchSampleOffsets(const WorldMi355Batch* b);  /* host, n_utt+1 */
const int64_t* WorldMi355BatchFrameOffsets(const WorldMi355Batch* b);   /* host, n_utt+1 */
const int64_t* WorldMi355BatchOutputOffsets(const WorldMi355Batch* b);  /* host, n_utt+1 */

/* Stage entry points; every pointer is a DEVICE pointer.  Asynchronous on the
 * context's stream unless stated.  Same meaning as the reference functions:
 *   Dio        dio.cpp:642-647        StoneMask  stonemask.cpp:211-217
 *   CheapTrick cheaptrick.cpp:200-228 D4C        d4c.cpp:337-397
 *   Synthesis  synthesis.cpp:338-397 (synchronises once internally: the pulse count
 *              decides the size of the overlap-add scratch)
 *   Harvest    harvest.cpp:1223-1255
 * refined_f0 of StoneMask must not be the f0 array (WM_ERR_BAD_ARG): the output is cleared first. */
int WorldMi355Dio(WorldMi355Batch* b, const double* x, double* t, double* f0);
int WorldMi355StoneMask(WorldMi355Batch* b, const double* x, const double* t, const double* f0,
                        double* refined_f0);
int WorldMi355CheapTrick(WorldMi355Batch* b, const double* x, const double* t, const double* f0,
                         double* sp);
int WorldMi355D4C(WorldMi355Batch* b, const double* x, const double* t, const double* f0, double* ap);
int WorldMi355Synthesis(WorldMi355Batch* b, const double* f0, const double* sp, const double* ap,
                        double* y);
int WorldMi355Harvest(WorldMi355Batch* b, const double* x, double* t, double* f0);

/* The sample formats of the callers (test/audioio.cpp), over the batch's concatenated samples, DEVICE pointers:
 *   x[i] = pcm[i] / 32768          wavread of a 16-bit file (:236-249); total samples of the batch
 *   pcm[i] = clamp(int(y[i] * 32767), -32768, 32767), the cast truncating towards zero: wavwrite (:160-167);
 *            total output samples of the batch.  A NaN sample writes 0 (the reference's cast is undefined there). */
int WorldMi355SamplesFromPcm16(WorldMi355Batch* b, const int16_t* pcm, double* x);
int WorldMi355SamplesToPcm16(WorldMi355Batch* b, const double* y, int16_t* pcm);

/* Dio -> StoneMask -> CheapTrick -> D4C, as test/analysis.cpp:243-390 chains them. */
int WorldMi355Analyze(WorldMi355Batch* b, const double* x, double* t, double* f0, double* sp,
                      double* ap);
/* Analyze followed by Synthesis of the features it produced -- the round trip BASELINE.json's metric times
 * (test/analysis.cpp then test/synth.cpp on the same utterances).  Same kernels, same results as the two
 * calls; the f0-only first part of Synthesis (time base, pulse list, the host round trip for the pulse
 * count) runs on a second stream beside CheapTrick and D4C instead of after them. */
int WorldMi355AnalyzeSynthesize(WorldMi355Batch* b, const double* x, double* t, double* f0, double* sp,
                                double* ap, double* y);

/* Per-utterance status of a batch: status[u] (DEVICE pointer, n_utt ints) receives a bit mask.  Utterances of a
 * batch never exchange data, so one with a flag set does not affect the results of the others (SURVEY.md section 5;
 * the reference has no error reporting: every entry point returns void).  x, f0, sp, ap are the arrays of the
 * analysis calls; any of them may be NULL and is then not scanned.  Asynchronous on the context's stream. */
#define WM_UTT_INPUT_NONFINITE 1   /* NaN / Inf among the samples of x */
#define WM_UTT_TOO_SHORT 2         /* f0_length <= Dio's voice_range_minimum: f0 is all zero (dio.cpp:263-266) */
#define WM_UTT_OUTPUT_NONFINITE 4  /* NaN / Inf among the utterance's f0 / sp / ap */
#define WM_UTT_D4C_DEFAULT_ROWS 8  /* never set since round 5 (kept for ABI compatibility): it marked frames with f0 >= fs / 16
                                    * at fs above 48.1 kHz, which kept D4C's default row; they are analysed now */
int WorldMi355UtteranceStatus(WorldMi355Batch* b, const double* x, const double* f0, const double* sp,
                              const double* ap, int* status);

/* ---- Feature codec (externs/WORLD_v2/src/codec.cpp), SURVEY.md section 8(f) ---------------------------
 * All arrays are device pointers over the batch's frames (row-major, frame order of the batch).
 *   coded sp : double[total_frames][number_of_dimensions]
 *   coded ap : double[total_frames][WorldMi355GetNumberOfAperiodicities(fs)]
 * fs and fft_size are the batch's. */
int WorldMi355GetNumberOfAperiodicities(int fs);                                  /* codec.cpp:212-215 */
int WorldMi355CodeSpectralEnvelope(WorldMi355Batch* b, const double* sp, int number_of_dimensions,
                                   double* coded);                                /* codec.cpp:268-295 */
int WorldMi355DecodeSpectralEnvelope(WorldMi355Batch* b, const double* coded, int number_of_dimensions,
                                     double* sp);                                 /* codec.cpp:297-324 */
int WorldMi355CodeAperiodicity(WorldMi355Batch* b, const double* ap, double* coded);    /* codec.cpp:217-235 */
int WorldMi355DecodeAperiodicity(WorldMi355Batch* b, const double* coded, double* ap);  /* codec.cpp:237-266 */
/* The coded float32 feature set of the recipe's call `analysis wav lf0 mgc bap 5 2048 50 25`
 * (data/Makefile.in:214, test/analysis.cpp:292-366): lf0[total_frames], mgc[total_frames][spec_dim],
 * bap[total_frames][ap_dim] from resident f0 / sp / ap. */
int WorldMi355RecipeFeatures(WorldMi355Batch* b, const double* f0, const double* sp, const double* ap,
                             int spec_dim, int ap_dim, float* lf0, float* mgc, float* bap);
/* The way back, as the synth CLI does it before Synthesis (test/synth.cpp:151-256):
 *   f0 = exp(lf0), 0 stays 0;  sp = DecodeSpectralEnvelope(mgc with c0 - 12.0) / 1e4;
 *   ap[.][j] = exp(mgc2sp(bap with c0 + 9.210340, order, alpha 0.55, gamma 0)[j]) / 1e4 for j < order,
 *   order = ap_dim (minus one when odd), mgc2sp being the CLI's SPTK port (test/sptkfunctions.cpp:186-274,
 *   :596-631).  The reference leaves bins >= order of every ap row uninitialised (synth.cpp:240-245); they
 *   are written as 0.0 here.  An even ap_dim makes mgc2sp read one coefficient past the row; it is taken as 0.
 *   ap_dim <= 64. */
int WorldMi355RecipeDecode(WorldMi355Batch* b, const float* lf0, const float* mgc, const float* bap, int spec_dim,
                           int ap_dim, double* f0, double* sp, double* ap);

/* ---- `cmp` composition (data/scripts/window.pl:45-146, addhtkheader.pl:45-82), SURVEY.md section 8(f) rank 3 ----
 * Applies each stream's dynamic-feature windows and lays the results side by side per frame:
 *   out[frame] = [stream 0: window 0 (dim) | window 1 | ...][stream 1: ...]...      (float32)
 * streams[s]: device pointer, float32 [total_frames][dims[s]].  windows[s][i]: HOST pointer to the
 * window_sizes[s][i] coefficients of window i of stream s (the content of data/win/NAME.win<i> without the
 * leading size).  Frames are clamped per utterance; -1e10 is the ignore value of unvoiced lf0.
 * Limits: 4 streams, 4 windows per stream, odd window sizes up to 15. */
int WorldMi355ComposeCmp(WorldMi355Batch* b, int n_streams, const float* const* streams, const int* dims,
                         const int* n_windows, const double* const* const* windows,
                         const int* const* window_sizes, float* out);
/* The 12-byte HTK header of addhtkheader.pl:60-75 (host only, native byte order). */
void WorldMi355HtkHeader(int n_frames, int sampling_rate, int frame_shift_samples, int bytes_per_frame,
                         int htk_type, unsigned char out12[12]);

/* The fwrite loops at the end of the reference's `analysis` CLI (test/analysis.cpp:360-390), for a whole batch at
 * once: file i = paths[i] receives bytes[i] bytes from data[i] (HOST memory: the pinned slabs a sweep copies the
 * float32 features into), created or truncated.  n_threads plain threads work the list off; returns when every
 * file is written and closed.  Host only: no context, no HIP call.  WM_ERR_IO names the first file that failed. */
int WorldMi355WriteFiles(int n_files, const char* const* paths, const void* const* data, const size_t* bytes,
                         int n_threads);


/* ---- Vibrato feature (data/scripts/Extract.py:115-227, invoked at data/Makefile.in:215), SURVEY.md 8(f) rank 4 ----
 * lf0: DEVICE, float32 [total_frames], log f0 with 0 for unvoiced frames (WorldMi355RecipeFeatures' lf0).
 * Label segments: HOST arrays.  Utterance u owns segments seg_utt_off[u] .. seg_utt_off[u+1]-1 (n_utt + 1 offsets);
 * segment s covers frames [seg_start[s], seg_end[s]) of its utterance (floor(label time / frame period), clamped as
 * Extract.py:181-182) and has note pitch seg_pitch[s] in Hz (0 for "xx").
 * Outputs (DEVICE, float32, after the script's soprLog): vib [total_frames][2] = log depth, log period;
 * lf0_out [total_frames][2] = log f0, log(f0 - pitch + 500) -- what the script writes back over the lf0 file.
 * *n_too_long (may be NULL): voiced runs longer than 3072 frames, left without vibrato.  Synchronises the stream.
 * Parity of this entry point is UNPINNED: the reference holds no fixtures for it and its LOWESS is an unpinned
 * third-party dependency (statsmodels); DESIGN.md section 4. */
int WorldMi355Vibrato(WorldMi355Batch* b, const float* lf0, const int* seg_utt_off, const int* seg_start,
                      const int* seg_end, const double* seg_pitch, float* vib, float* lf0_out, int* n_too_long);

/* Per-kernel timing with HIP events recorded on the context's stream around each launch of the
 * named kernels ("dio_lowcut_kernel", "dio_band_kernel", "stonemask_kernel", "cheaptrick_kernel",
 * "d4c_lovetrain_kernel", "d4c_kernel", "synth_timebase_kernel", "synth_pulse_kernel",
 * "synth_ola_kernel").  Enable clears earlier records; Query synchronises the stream and returns
 * the summed duration and the number of launches since Enable. */
int WorldMi355TimingEnable(WorldMi355Context* ctx, int on);
int WorldMi355TimingQuery(WorldMi355Context* ctx, const char* kernel, double* total_ms, int* launches);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* WORLD_MI355_H_ */

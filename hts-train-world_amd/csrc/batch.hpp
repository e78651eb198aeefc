// batch.hpp -- host-side state shared by the kernel launchers and the C ABI.
//
// Context : one per (process, device): HIP stream, the universal randn table
//           (matlabfunctions.cpp:247-277 as data), launch geometry.
// Batch   : one per set of utterances: lengths/offsets on host and device, the
//           frame->utterance map and every scratch buffer the kernels need, all
//           allocated at creation so the analysis path never calls hipMalloc.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/world_mi355.h"

namespace wm {

int wm_check(hipError_t e);   // maps to WM_ERR_HIP and records the message
void set_error(const char* msg);

// Device memory of batches and stage workspaces comes from a process-wide cache of freed blocks (context.cpp):
// the drop-in C API builds a batch per call signature, and hipMalloc / hipFree (which also synchronises the
// device) per utterance were a third of its latency.  A block goes back to the cache when it is freed and is
// handed out again for a request of its size class; callers free only memory whose users have finished or are
// ordered on the stream that will use it next (DestroyBatch synchronises its context's stream first).
hipError_t dev_alloc_bytes(void** p, size_t bytes);
void dev_free(void* p);
void dev_cache_trim(size_t keep_bytes);       // hipFree cached blocks until at most keep_bytes stay cached
template <class T> inline hipError_t dev_alloc(T** p, size_t bytes) { return dev_alloc_bytes((void**)p, bytes); }

struct Context {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int num_cu = 256;
  int frame_grid = 256 * 8;          // upper bound on the workgroups of a grid-stride per-frame kernel
  int oversub = 6;                   // workgroups per resident slot of those kernels (persistent_grid): measured on
                                     // configs[1], round 2: 21.2 / 18.0 / 17.7 / 17.3 / 17.3 ms per step at 1 / 2 / 4 / 8 /
                                     // 16; end of round 3 (tools/oversub_sweep.sh): 15.0 / 14.2 / 14.1 / 14.3 / 13.8 / 14.2 /
                                     // 14.0 / 14.7 / 17.0 at 2 / 3 / 4 / 5 / 6 / 7 / 8 / 16 / 32 -- a workgroup's start (twiddle
                                     // bases, the blocking fill of the scalar pipe, its share of the default rows) is
                                     // worth most of a frame, a large share makes the last round ragged
  uint32_t* d_rng = nullptr;         // universal randn table, uint32 sums
  int64_t rng_cap = 0;
  uint32_t rng_state[4] = {123456789u, 362436069u, 521288629u, 88675123u};   // matlabfunctions.cpp:247-250
  double* d_scratch = nullptr;       // growable scratch (synthesis responses)
  int64_t scratch_cap = 0;           // in doubles
  int ensure_rng(int64_t count);     // grow + (re)generate, returns error code
  int ensure_scratch(int64_t doubles);
  int64_t* h_pulse_info = nullptr;   // pinned, mapped: {total pulses, largest per-utterance count} of the last Synthesis
  int64_t* d_pulse_info = nullptr;   // the same memory as the device sees it
  // second stream + events of launch_analyze_synthesize (created on first use)
  hipStream_t side = nullptr;
  hipStream_t aux = nullptr;         // D4C's RARE launch (d4c_rare)
  hipStream_t prep = nullptr;        // launch_synthesis: the f0-only kernels of the batch's second part
  hipEvent_t ev_call = nullptr, ev_prep_b = nullptr;
  hipEvent_t ev_f0 = nullptr, ev_prep = nullptr, ev_d4c = nullptr, ev_rare = nullptr;
  hipEvent_t ev_pulse[2] = {nullptr, nullptr}, ev_ola[2] = {nullptr, nullptr};   // synthesis_render's two response halves
  int ensure_side();                 // the second stream and its events, created on first use
  // Dio's filters -- low-cut taps, Nuttall windows and their block spectra (fftconv.hpp) -- depend on the options and
  // the sampling rate only, not on the utterances: built once per context and configuration and shared by every batch
  // (the drop-in API makes a batch per utterance length: rebuilding them per batch was 0.1 ms of every Dio call)
  struct DioFilters {
    int fs, speed;
    double f0_floor, f0_ceil, channels;
    double* d_lowcut;
    double* d_win;
    void* d_H;
  };
  std::vector<DioFilters> dio_filters;
  // likewise: D4C's Nuttall window by length (a function of the sampling rate) and StoneMask's DFT twiddle table
  std::vector<std::pair<int, double*>> nuttall_windows;
  std::vector<std::pair<int, double*>> dc_removers;   // Synthesis' GetDCRemover table by fft_size (a one-thread kernel:
                                                      // 0.29 ms in front of every Synthesis() of a new utterance length)
  void* d_sm_twid = nullptr;
  // optional per-kernel HIP-event timing on `stream` (bench.py's roofline leg)
  bool timing = false;
  std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> timed;
  void timing_clear();
};

// Grid of a grid-stride per-frame kernel: a small multiple (Context::oversub) of the workgroups that are resident
// at once.  Exactly the resident number gives every wave slot the same share when the kernel has the machine to
// itself, but then a slot that is busy with another stream's kernel when this one starts (the f0-only first part
// of Synthesis runs beside CheapTrick and D4C) delays a whole share by that much: the launch ends late by the
// overlap.  With a few workgroups per slot the hardware dispatcher hands the shares out as slots become free; the
// price is a last round that is not full (at most one share of 1 / oversub of a slot's work).
// The occupancy query is made once per (device, kernel, block size) -- a process may hold contexts on several GPUs --
// and remembered (context.cpp).  slot_get / slot_raise read and update an entry under the table's lock: two threads
// that drive contexts concurrently may both ask the runtime the first time, neither reads a half-written value.
int slot_get(int device, const void* kernel, int tag);                 // 0 = never set
int slot_raise(int device, const void* kernel, int tag, int value);    // entry = max(entry, value); returns the old entry
template <class K> inline int persistent_grid(const Context& c, K kernel, int block, int64_t items) {
  int per_cu = slot_get(c.device, (const void*)kernel, block);
  if (per_cu == 0) {
    int q = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kernel, block, 0) != hipSuccess || q < 1) q = 4;
    per_cu = q;
    (void)slot_raise(c.device, (const void*)kernel, block, q);
  }
  const int64_t g = (int64_t)c.num_cu * per_cu * c.oversub;
  return (int)(items < g ? items : g);
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize): once per (device, kernel), and again whenever a launch asks for more
// than the largest size granted so far (hv_refine_kernel's size varies with the batch).
template <class K> inline void allow_dynamic_lds(const Context& c, K kernel, int bytes) {
  if (slot_raise(c.device, (const void*)kernel, -1, bytes) >= bytes) return;
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// Every entry point that takes a batch or a context runs on the context's device, whatever device is current on
// the calling thread, and leaves the caller's current device as it found it.
struct OnDevice {
  int prev = -1;
  explicit OnDevice(const Context& c) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != c.device && hipSetDevice(c.device) == hipSuccess) prev = cur;
  }
  ~OnDevice() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  OnDevice(const OnDevice&) = delete;
  OnDevice& operator=(const OnDevice&) = delete;
};

// RAII bracket: records a start/stop event pair around the launches in its scope.
struct TimedScope {
  Context* c;
  hipEvent_t a = nullptr, b = nullptr;
  TimedScope(Context* ctx, const char* name) : c(ctx) {
    if (!c->timing) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
    (void)hipEventRecord(a, c->stream);
    c->timed[name].push_back(std::make_pair(a, b));
  }
  ~TimedScope() {
    if (b) (void)hipEventRecord(b, c->stream);
  }
};

struct Batch {
  Context* ctx = nullptr;
  WorldMi355Params p{};
  int n_utt = 0;
  std::vector<int> x_len, f0_len, y_len;
  std::vector<int64_t> x_off, f_off, y_off;
  int64_t total_x = 0, total_f = 0, total_y = 0;
  int max_x_len = 0, max_f0_len = 0, max_y_len = 0;
  // device descriptors: sections of one allocation (d_arena), laid out by WorldMi355CreateBatch
  void* d_arena = nullptr;
  int64_t *d_x_off = nullptr, *d_f_off = nullptr, *d_y_off = nullptr;
  int *d_x_len = nullptr, *d_f0_len = nullptr, *d_y_len = nullptr;
  int* d_frame_utt = nullptr;        // [total_f]
  int* d_rng_off = nullptr;          // [total_f] per-frame randn offsets (CheapTrick / D4C phase 2)
  int* d_rng_off2 = nullptr;         // [total_f] D4C LoveTrain offsets
  double* d_ap0 = nullptr;           // [total_f] D4C LoveTrain result
  double* d_f0_tmp = nullptr;        // [total_f] raw DIO f0 before StoneMask
  int* d_perm = nullptr;             // [total_f] costly frames first (partition.hpp)
  int* d_perm2 = nullptr;            // [total_f] D4C: frames that need the wide-margin kernel first
  int* d_part_cnt = nullptr;         // [total_f / 1024 + 2]
  int* d_part_n = nullptr;           // [4] number of listed frames
  // D4C's own offsets and lists: its preparation may run beside CheapTrick (d4c_prepare, d4c.hip)
  int* d_rng_off_d4c = nullptr;      // [total_f]
  int* d_perm_d4c = nullptr;         // [total_f]
  int* d_part_cnt_d4c = nullptr;     // [total_f / 1024 + 2]
  int* d_part_n_d4c = nullptr;       // [4]
  void* d_sm_twid = nullptr;         // StoneMask's DFT twiddle table (stonemask.hip): the context's, not owned
  // D4C tables
  double* d_d4c_window = nullptr;    // Nuttall window of GetCoarseAperiodicity: the context's, not owned (non-null = D4C's tables are ready)
  int* d_utt_total = nullptr;        // [n_utt] LoveTrain randn totals
  double* d_d4c_big = nullptr;       // fft_size_d4c 4096: centroid quarters, centroid, group delay, coarse values per frame
  // DIO workspace
  bool dio_ready = false;
  void* dio_host = nullptr;          // DioHost (dio.hip)
  double* d_dio_lowcut = nullptr;    // low-cut FIR taps by lag
  double* d_dio_win = nullptr;       // Nuttall low-pass windows, all bands
  void* d_dio_desc = nullptr;        // ONE block holding the per-utterance tables below (fft, ylen, offsets): one upload
  int* d_dio_fft = nullptr;          // [n_utt] the reference's fft_size (circular indexing)
  double* d_dio_ws = nullptr;        // [3][total_f] contour work arrays
  void* d_dio_H = nullptr;           // filter spectra of the FFT-convolution path (fftconv.hpp)
  int* d_dio_edges = nullptr;        // [n_utt][2][edge_cap] edge lists of the contour fix when they outgrow LDS
  int* d_dio_ylen = nullptr;         // [n_utt] y_length = 1 + N / speed
  double* d_dio_y = nullptr;         // decimated signals (speed > 1)
  double* d_dio_tmp = nullptr;       // decimation pass-1 output
  int64_t* d_dio_yoff = nullptr;
  int64_t* d_dio_toff = nullptr;
  int64_t dio_tot_y = 0;
  int dio_bands = 0;
  double* d_dio_mean = nullptr;      // [n_utt]
  double* d_dio_mean_part = nullptr; // [n_utt][32] partial sums
  double* d_dio_z = nullptr;         // low-cut output, per utterance y_len + 2*pad
  int64_t* d_dio_z_off = nullptr;
  std::vector<int64_t> dio_z_off;
  int dio_pad = 0;
  double* d_dio_events = nullptr;    // fine edges, [utt][band][4][cap]
  int64_t* d_dio_ev_off = nullptr;   // per utterance base into events (in doubles)
  std::vector<int64_t> dio_ev_off;
  int* d_dio_ev_cnt = nullptr;       // [utt][band][4]
  int* d_dio_tile_cnt = nullptr;     // [utt][band][tile + 1][4] per-tile event counts, then offsets
  double* d_dio_slots = nullptr;     // staged events, [utt][band][4][tiles(utt)][kZcSlot]
  int64_t* d_dio_slot_off = nullptr; // per utterance base into d_dio_slots
  double* d_dio_cand = nullptr;      // [band][total_f]
  double* d_dio_score = nullptr;     // [band][total_f]
  void* harvest_ws = nullptr;        // HarvestWs (harvest.hip)
  void* codec_tables = nullptr;      // CodecTables (codec.hip)
  void* vibrato_ws = nullptr;        // VibWs (vibrato.hip)
  // Synthesis workspace: sections of one allocation (d_syn_arena), laid out by synthesis_prepare
  void* d_syn_arena = nullptr;
  int* d_pulse_idx = nullptr;        // [total_y]
  double* d_pulse_shift = nullptr;   // [total_y]
  double* d_vuv = nullptr;           // [total_y] interpolated vuv
  double* d_phase = nullptr;         // [total_y] scratch (increments / wrapped phase)
  int* d_pulse_cnt = nullptr;        // [n_utt]
  int* d_pulse_tile_cnt = nullptr;   // [n_utt][tiles] pulses per search tile
  int64_t* d_pulse_off = nullptr;    // [n_utt+1]
  int* d_pulse_first = nullptr;      // first pulse at or after every 128th sample of an utterance (the overlap-add's table)
  int* d_syn_order = nullptr;        // [2 n_utt] the identity, then the utterances by output length (shortest first)
  std::vector<int> syn_sorted;       // host copy of the second half
  std::vector<int> syn_order_host;   // what d_syn_order was uploaded from: kept alive, so the upload needs no wait
  void* d_pulse_rec = nullptr;       // [pulse_rec_cap] PulseRec (synthesis.hip), grown on demand
  int64_t pulse_rec_cap = 0;
  int* d_pulse_perm = nullptr;       // [cap] voiced-first pulse order of a chunk, then n, then block counts
  double* d_dc_remover = nullptr;    // [fft_size]: the context's, not owned
  int64_t syn_total_p = 0, syn_chunk = 0;   // pulses of the prepared synthesis, pulses per piece (half of the response scratch)
  bool syn_warm = false;                    // launch_analyze_synthesize has run once on this batch

  int64_t rng_bound_cheaptrick() const;
  int64_t rng_bound_d4c() const;
  int64_t rng_bound_synthesis() const;
};

// kernel launchers (one translation unit each)
int launch_dio(Batch& b, const double* d_x, double* d_t, double* d_f0);
int launch_stonemask(Batch& b, const double* d_x, const double* d_t, const double* d_f0, double* d_out,
                     double f0_lower = 0.0);
int launch_cheaptrick(Batch& b, const double* d_x, const double* d_t, const double* d_f0, double* d_sp);
int launch_d4c(Batch& b, const double* d_x, const double* d_t, const double* d_f0, double* d_ap);
int launch_analyze(Batch& b, const double* d_x, double* d_t, double* d_f0, double* d_sp, double* d_ap);
int d4c_prepare(Batch& b, const double* d_x, const double* d_t, const double* d_f0);
int d4c_rare(Batch& b, const double* d_x, const double* d_t, const double* d_f0, double* d_ap);
int d4c_run(Batch& b, const double* d_x, const double* d_t, const double* d_f0, double* d_ap);
int launch_synthesis(Batch& b, const double* d_f0, const double* d_sp, const double* d_ap, double* d_y);
int synthesis_prepare(Batch& b, const double* d_f0, double* d_y);
int synthesis_begin(Batch& b, const double* d_f0, double* d_y);      // the f0-only kernels, queued
int synthesis_prepare_wait(Batch& b);                                 // their host round trip
int synthesis_render(Batch& b, const double* d_sp, const double* d_ap, double* d_y);
int launch_analyze_synthesize(Batch& b, const double* d_x, double* d_t, double* d_f0, double* d_sp, double* d_ap,
                              double* d_y);
int launch_utterance_status(Batch& b, const double* d_x, const double* d_f0, const double* d_sp, const double* d_ap,
                            int* d_status);
int launch_vibrato(Batch& b, const float* d_lf0, const int* seg_utt_off, const int* seg_start, const int* seg_end,
                   const double* seg_pitch, float* d_vib, float* d_lf0_out, int* n_too_long);
int launch_pcm16_to_samples(Batch& b, const int16_t* d_pcm, double* d_x);
int launch_samples_to_pcm16(Batch& b, const double* d_y, int16_t* d_pcm);
int codec_num_aperiodicities(int fs);
int launch_code_spectral_envelope(Batch& b, const double* d_sp, int ndim, double* d_coded);
int launch_decode_spectral_envelope(Batch& b, const double* d_coded, int ndim, double* d_sp);
int launch_code_aperiodicity(Batch& b, const double* d_ap, double* d_coded);
int launch_decode_aperiodicity(Batch& b, const double* d_coded, double* d_ap);
int launch_compose_cmp(Batch& b, int n_streams, const float* const* d_data, const int* dims, const int* n_windows,
                       const double* const* const* windows, const int* const* window_sizes, float* d_out);
int launch_recipe_features(Batch& b, const double* d_f0, const double* d_sp, const double* d_ap, int spec_dim,
                           int ap_dim, float* d_lf0, float* d_mgc, float* d_bap);
int launch_recipe_decode(Batch& b, const float* d_lf0, const float* d_mgc, const float* d_bap, int spec_dim,
                         int ap_dim, double* d_f0, double* d_sp, double* d_ap);

}  // namespace wm

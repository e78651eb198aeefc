// context.cpp -- Context / Batch life cycle for libworld_mi355.so (host side, C++).
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include <mutex>
#include <unordered_map>
#include <iterator>

#include "batch.hpp"
#include "common.hpp"

namespace wm {

static thread_local std::string g_last_error;

int wm_check(hipError_t e) {
  if (e == hipSuccess) return WM_OK;
  g_last_error = std::string("HIP: ") + hipGetErrorString(e);
  return WM_ERR_HIP;
}

namespace {
// Blocks are cached per (device, size class): a process may hold contexts on several GPUs
// (WorldMi355CreateContext(device, ...)), and a block must only ever be handed back to the device it lives on.
struct DevCache {
  typedef std::pair<int, size_t> Key;                // (device, size class)
  std::mutex mu;
  std::multimap<Key, void*> free_blocks;
  std::unordered_map<void*, Key> owner;              // every live or cached block -> its device and size class
  size_t cached = 0;                                 // idle bytes, all devices
  size_t limit = (size_t)8 << 30;                    // cached (idle) bytes kept at most; WORLD_MI355_CACHE_MB
  DevCache() {
    if (const char* e = getenv("WORLD_MI355_CACHE_MB")) limit = (size_t)strtoull(e, nullptr, 10) << 20;
  }
};
DevCache& dev_cache() {
  static DevCache* c = new DevCache;                 // never destroyed: frees may come from static destructors
  return *c;
}
// size classes: powers of two up to 1 MiB, eighths of a power of two above (at most 12.5 % over-allocation)
size_t size_class(size_t bytes) {
  if (bytes < 256) bytes = 256;
  size_t p2 = 256;
  while (p2 < bytes) p2 <<= 1;
  if (p2 <= ((size_t)1 << 20)) return p2;
  const size_t step = p2 >> 4;                       // p2/2 < bytes <= p2: steps of p2/16
  return (bytes + step - 1) / step * step;
}
int current_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d;
}
// hipFree cached blocks (largest first) until at most keep_bytes stay cached; device < 0: of every device
void trim_locked_out(size_t keep_bytes, int device) {
  DevCache& c = dev_cache();
  std::vector<std::pair<int, void*>> victims;
  {
    std::lock_guard<std::mutex> g(c.mu);
    // largest first: walk the size classes from the top, across devices
    while (c.cached > keep_bytes && !c.free_blocks.empty()) {
      auto best = c.free_blocks.end();
      for (auto it = c.free_blocks.begin(); it != c.free_blocks.end(); ++it)
        if ((device < 0 || it->first.first == device) && (best == c.free_blocks.end() || it->first.second > best->first.second))
          best = it;
      if (best == c.free_blocks.end()) break;
      victims.push_back(std::make_pair(best->first.first, best->second));
      c.cached -= best->first.second;
      c.owner.erase(best->second);
      c.free_blocks.erase(best);
    }
  }
  if (victims.empty()) return;
  const int here = current_device();
  int at = here;
  for (auto& v : victims) {
    if (v.first != at) { (void)hipSetDevice(v.first); at = v.first; }
    (void)hipFree(v.second);
  }
  if (at != here) (void)hipSetDevice(here);
}
}  // namespace

hipError_t dev_alloc_bytes(void** p, size_t bytes) {
  DevCache& c = dev_cache();
  const DevCache::Key key(current_device(), size_class(bytes));
  {
    std::lock_guard<std::mutex> g(c.mu);
    auto it = c.free_blocks.find(key);
    if (it != c.free_blocks.end()) {
      *p = it->second;
      c.free_blocks.erase(it);
      c.cached -= key.second;
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(p, key.second);
  if (e != hipSuccess) {                             // out of memory with idle blocks around: give this device's back and retry
    (void)hipGetLastError();
    trim_locked_out(0, key.first);
    e = hipMalloc(p, key.second);
  }
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> g(c.mu);
    c.owner[*p] = key;
  }
  return e;
}

void dev_free(void* p) {
  if (!p) return;
  DevCache& c = dev_cache();
  size_t over = 0;
  {
    std::lock_guard<std::mutex> g(c.mu);
    auto it = c.owner.find(p);
    if (it == c.owner.end()) {                       // not ours (should not happen): plain free
      (void)hipFree(p);
      return;
    }
    c.free_blocks.emplace(it->second, p);
    c.cached += it->second.second;
    over = c.cached > c.limit ? c.limit / 2 : (size_t)-1;
  }
  if (over != (size_t)-1) dev_cache_trim(over);
}

void dev_cache_trim(size_t keep_bytes) { trim_locked_out(keep_bytes, -1); }
namespace {
struct SlotKey {
  int device;
  const void* kernel;
  int tag;                 // block size of an occupancy entry, -1 for the dynamic-LDS entry
  bool operator<(const SlotKey& o) const {
    return device != o.device ? device < o.device : kernel != o.kernel ? kernel < o.kernel : tag < o.tag;
  }
};
std::mutex g_slot_mu;
std::map<SlotKey, int>& slot_table() {
  static std::map<SlotKey, int>* t = new std::map<SlotKey, int>;   // never destroyed: used from atexit paths
  return *t;
}
}  // namespace
int slot_get(int device, const void* kernel, int tag) {
  std::lock_guard<std::mutex> g(g_slot_mu);
  auto it = slot_table().find(SlotKey{device, kernel, tag});
  return it == slot_table().end() ? 0 : it->second;
}
int slot_raise(int device, const void* kernel, int tag, int value) {
  std::lock_guard<std::mutex> g(g_slot_mu);
  int& e = slot_table()[SlotKey{device, kernel, tag}];
  const int old = e;
  if (value > e) e = value;
  return old;
}
const char* last_error() { return g_last_error.c_str(); }
void set_error(const char* msg) { g_last_error = msg; }

// xorshift128 stream of the reference's randn(), as uint32 sums of 12 draws
// (matlabfunctions.cpp:247-277); index k = k-th sample after randn_reseed().
// `st` carries the generator state from one call to the next, so a table can be extended.
static void fill_randn_u32(uint32_t* out, int64_t count, uint32_t (&st)[4]) {
  uint32_t x = st[0], y = st[1], z = st[2], w = st[3];
  for (int64_t k = 0; k < count; ++k) {
    uint32_t acc = 0;
    for (int j = 0; j < 12; ++j) {
      uint32_t t = x ^ (x << 11);
      x = y; y = z; z = w;
      w = (w ^ (w >> 19)) ^ (t ^ (t >> 8));
      acc += w >> 4;
    }
    out[k] = acc;
  }
  st[0] = x; st[1] = y; st[2] = z; st[3] = w;
}

// The table only ever grows: entries [0, rng_cap) stay, the tail continues the xorshift stream from
// the saved state (a longer utterance after a shorter one costs its extra entries, not a new table).
int Context::ensure_rng(int64_t count) {
  // consumers index the table with 32-bit per-utterance offsets (d_rng_off, randn_at): an utterance whose draws
  // could pass 2^31 (about 135 k frames at 48 kHz) is refused instead of wrapping silently
  if (count > (int64_t)INT32_MAX) {
    set_error("utterance too long: its randn offsets would not fit 32 bits");
    return WM_ERR_UNSUPPORTED;
  }
  if (count <= rng_cap) return WM_OK;
  int64_t cap = count + count / 4 + 4096;
  if (cap < 2 * rng_cap) cap = 2 * rng_cap;
  const int64_t add = cap - rng_cap;
  std::vector<uint32_t> host((size_t)add);
  fill_randn_u32(host.data(), add, rng_state);
  // the old table may still be read by kernels in flight on the stream
  int rc = wm_check(hipStreamSynchronize(stream));
  if (rc) return rc;
  uint32_t* grown = nullptr;
  rc = wm_check(dev_alloc(&grown, sizeof(uint32_t) * (size_t)cap));
  if (rc) return rc;
  if (rng_cap > 0)
    rc = wm_check(hipMemcpy(grown, d_rng, sizeof(uint32_t) * (size_t)rng_cap, hipMemcpyDeviceToDevice));
  if (!rc)
    rc = wm_check(hipMemcpy(grown + rng_cap, host.data(), sizeof(uint32_t) * (size_t)add, hipMemcpyHostToDevice));
  if (rc) { dev_free(grown); return rc; }
  if (d_rng) dev_free(d_rng);
  d_rng = grown;
  rng_cap = cap;
  return WM_OK;
}

int Context::ensure_scratch(int64_t doubles) {
  if (doubles <= scratch_cap) return WM_OK;
  int rc = wm_check(hipStreamSynchronize(stream));
  if (rc) return rc;
  if (d_scratch) dev_free(d_scratch);
  d_scratch = nullptr;
  scratch_cap = 0;
  rc = wm_check(dev_alloc(&d_scratch, sizeof(double) * (size_t)doubles));
  if (rc) return rc;
  scratch_cap = doubles;
  return WM_OK;
}

int Context::ensure_side() {
  if (side) return WM_OK;
  int rc = wm_check(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  rc = rc ? rc : wm_check(hipEventCreateWithFlags(&ev_f0, hipEventDisableTiming));
  rc = rc ? rc : wm_check(hipEventCreateWithFlags(&ev_prep, hipEventDisableTiming));
  rc = rc ? rc : wm_check(hipEventCreateWithFlags(&ev_d4c, hipEventDisableTiming));
  rc = rc ? rc : wm_check(hipEventCreateWithFlags(&ev_rare, hipEventDisableTiming));
  rc = rc ? rc : wm_check(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
  for (int h = 0; h < 2 && !rc; ++h) {
    rc = wm_check(hipEventCreateWithFlags(&ev_pulse[h], hipEventDisableTiming));
    rc = rc ? rc : wm_check(hipEventCreateWithFlags(&ev_ola[h], hipEventDisableTiming));
  }
  return rc;
}

// D4C around CheapTrick in the one-call forms, after ev_f0 has been recorded on the caller's stream and the second
// stream waits for it: the preparation on the second stream, the RARE launch on the third (its waves wait for whole
// SIMDs, i.e. for CheapTrick to drain, and would hold up whatever came behind them on a shared stream); d4c_after()
// queues the usual kernel behind the preparation and makes the caller's stream wait for the RARE launch.
static int d4c_beside(Batch& b, const double* d_x, const double* d_t, const double* d_f0, double* d_ap) {
  Context& c = *b.ctx;
  hipStream_t main_stream = c.stream;
  c.stream = c.side;
  int rc = d4c_prepare(b, d_x, d_t, d_f0);
  rc = rc ? rc : wm_check(hipEventRecord(c.ev_d4c, c.side));
  rc = rc ? rc : wm_check(hipStreamWaitEvent(c.aux, c.ev_d4c, 0));
  c.stream = c.aux;
  rc = rc ? rc : d4c_rare(b, d_x, d_t, d_f0, d_ap);
  rc = rc ? rc : wm_check(hipEventRecord(c.ev_rare, c.aux));
  c.stream = main_stream;
  return rc;
}
static int d4c_after(Batch& b, const double* d_x, const double* d_t, const double* d_f0, double* d_ap) {
  Context& c = *b.ctx;
  int rc = wm_check(hipStreamWaitEvent(c.stream, c.ev_d4c, 0));
  rc = rc ? rc : d4c_run(b, d_x, d_t, d_f0, d_ap);
  return rc ? rc : wm_check(hipStreamWaitEvent(c.stream, c.ev_rare, 0));
}

// Dio -> StoneMask -> CheapTrick -> D4C as one call: D4C's preparation runs on the second stream beside CheapTrick
// once the batch has its D4C tables (the first use allocates them, in the plain order).
int launch_analyze(Batch& b, const double* d_x, double* d_t, double* d_f0, double* d_sp, double* d_ap) {
  Context& c = *b.ctx;
  int rc = launch_dio(b, d_x, d_t, b.d_f0_tmp);
  rc = rc ? rc : launch_stonemask(b, d_x, d_t, b.d_f0_tmp, d_f0, b.p.f0_floor);
  if (rc) return rc;
  if (!b.d_d4c_window) {
    rc = launch_cheaptrick(b, d_x, d_t, d_f0, d_sp);
    return rc ? rc : launch_d4c(b, d_x, d_t, d_f0, d_ap);
  }
  int64_t need = b.rng_bound_cheaptrick();                 // the randn table must not move under two streams
  if (b.rng_bound_d4c() > need) need = b.rng_bound_d4c();
  rc = c.ensure_side();
  rc = rc ? rc : c.ensure_rng(need);
  rc = rc ? rc : wm_check(hipEventRecord(c.ev_f0, c.stream));
  rc = rc ? rc : wm_check(hipStreamWaitEvent(c.side, c.ev_f0, 0));
  if (rc) return rc;
  rc = d4c_beside(b, d_x, d_t, d_f0, d_ap);
  rc = rc ? rc : launch_cheaptrick(b, d_x, d_t, d_f0, d_sp);
  return rc ? rc : d4c_after(b, d_x, d_t, d_f0, d_ap);
}

// Analysis followed by Synthesis of the same features (BASELINE.json's metric), as one call.  Identical
// launches and results; the only difference is where the f0-only first part of Synthesis runs: on a second
// stream, as soon as StoneMask has produced f0, beside CheapTrick and D4C.  Its kernels are latency chains on
// one wavefront per utterance (the phase accumulation) plus a host round trip for the pulse count, which would
// otherwise sit between D4C and the pulse kernel with the machine idle.
int launch_analyze_synthesize(Batch& b, const double* d_x, double* d_t, double* d_f0, double* d_sp, double* d_ap,
                              double* d_y) {
  Context& c = *b.ctx;
  int rc = c.ensure_side();
  if (rc) return rc;
  // the randn table may be reallocated when it grows: settle its size before two streams read it
  int64_t need = b.rng_bound_cheaptrick();
  if (b.rng_bound_d4c() > need) need = b.rng_bound_d4c();
  if (b.rng_bound_synthesis() > need) need = b.rng_bound_synthesis();
  rc = c.ensure_rng(need);
  if (!rc && !b.syn_warm) {
    // first use of this batch: Synthesis allocates its work arrays and sizes the response scratch, and a
    // hipMalloc of gigabytes stalls kernels running beside it -- take the plain order once
    b.syn_warm = true;
    rc = launch_dio(b, d_x, d_t, b.d_f0_tmp);
    rc = rc ? rc : launch_stonemask(b, d_x, d_t, b.d_f0_tmp, d_f0, b.p.f0_floor);
    rc = rc ? rc : launch_cheaptrick(b, d_x, d_t, d_f0, d_sp);
    rc = rc ? rc : launch_d4c(b, d_x, d_t, d_f0, d_ap);
    return rc ? rc : launch_synthesis(b, d_f0, d_sp, d_ap, d_y);
  }
  rc = rc ? rc : launch_dio(b, d_x, d_t, b.d_f0_tmp);
  rc = rc ? rc : launch_stonemask(b, d_x, d_t, b.d_f0_tmp, d_f0, b.p.f0_floor);
  rc = rc ? rc : wm_check(hipEventRecord(c.ev_f0, c.stream));
  if (rc) return rc;
  hipStream_t main_stream = c.stream;
  // second stream, from f0 on: D4C's preparation (offsets, LoveTrain, frame lists), then Synthesis's
  rc = wm_check(hipStreamWaitEvent(c.side, c.ev_f0, 0));
  rc = rc ? rc : d4c_beside(b, d_x, d_t, d_f0, d_ap);
  rc = rc ? rc : launch_cheaptrick(b, d_x, d_t, d_f0, d_sp);
  rc = rc ? rc : d4c_after(b, d_x, d_t, d_f0, d_ap);
  if (!rc) {
    c.stream = c.side;
    rc = synthesis_prepare(b, d_f0, d_y);
    if (!rc) rc = wm_check(hipEventRecord(c.ev_prep, c.side));
    c.stream = main_stream;
  }
  rc = rc ? rc : wm_check(hipStreamWaitEvent(main_stream, c.ev_prep, 0));
  return rc ? rc : synthesis_render(b, d_sp, d_ap, d_y);
}

void Context::timing_clear() {
  for (auto& kv : timed)
    for (auto& pr : kv.second) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  timed.clear();
}

int64_t Batch::rng_bound_cheaptrick() const {
  // per frame: window 2*hw+1 <= fft_size+1 (hw < (fft_size-3)/2 by the f0 floor) + fft_size/2+1
  return (int64_t)max_f0_len * (p.fft_size + 1 + p.fft_size / 2 + 1) + 64;
}
int64_t Batch::rng_bound_d4c() const {
  // LoveTrain: 2*round(1.5 fs/40)+1 per voiced frame; body: 3 windows of 2*round(2 fs/47)+1
  int64_t lt = 2 * (int64_t)matlab_round(1.5 * p.fs / 40.0) + 1;
  int64_t body = 3 * (2 * (int64_t)matlab_round(2.0 * p.fs / 47.0) + 1);
  return (int64_t)max_f0_len * (lt + body) + 64;
}
int64_t Batch::rng_bound_synthesis() const { return (int64_t)max_y_len + 64; }

}  // namespace wm

namespace wm {
void dio_free_host(void* h);
void harvest_free(void* p);
void codec_free(void* p);
void vibrato_free(void* p);

void free_batch_buffers(Batch& b) {
  void* ptrs[] = {b.d_arena, b.d_perm2, b.d_utt_total, b.d_d4c_big,
                  // (Dio's filters are the context's, its per-utterance tables live in d_dio_desc)
                  b.d_dio_desc, b.d_dio_ws, b.d_dio_edges, b.d_dio_y, b.d_dio_tmp, b.d_dio_mean, b.d_dio_mean_part, b.d_dio_z,
                  b.d_dio_events, b.d_dio_ev_cnt, b.d_dio_tile_cnt, b.d_dio_slots, b.d_dio_cand,
                  b.d_dio_score, b.d_syn_arena, b.d_pulse_rec, b.d_pulse_perm};
  for (void* p : ptrs)
    if (p) dev_free(p);
  if (b.dio_host) dio_free_host(b.dio_host);
  b.dio_host = nullptr;
  if (b.harvest_ws) harvest_free(b.harvest_ws);
  b.harvest_ws = nullptr;
  if (b.codec_tables) codec_free(b.codec_tables);
  b.codec_tables = nullptr;
  if (b.vibrato_ws) vibrato_free(b.vibrato_ws);
  b.vibrato_ws = nullptr;
}
}  // namespace wm

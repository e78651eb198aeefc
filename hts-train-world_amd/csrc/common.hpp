// common.hpp -- device helpers shared by the WORLD hot-path kernels (gfx950).
//
// Restates, for one 64-lane wavefront working on LDS-resident spectra, the
// reference's small numeric helpers:
//   matlab_round       externs/WORLD_v2/src/matlabfunctions.cpp:212-214
//   randn (as table)   matlabfunctions.cpp:247-277
//   interp1Q           matlabfunctions.cpp:220-241
//   DCCorrection       common.cpp:56-75
//   LinearSmoothing    common.cpp:27-46, 77-111
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wavesync.hpp"

namespace wm {

constexpr double kPi = 3.1415926535897932384;
constexpr double kLog2 = 0.69314718055994529;
constexpr double kSafe = 0.000000000001;                       // kMySafeGuardMinimum
constexpr double kEps = 0.00000000000000022204460492503131;   // kEps
constexpr double kDefaultF0 = 500.0;
constexpr double kBig = 100000.0;                              // kMaximumValue

__host__ __device__ __forceinline__ int matlab_round(double x) {
  return x > 0 ? (int)(x + 0.5) : (int)(x - 0.5);
}
__host__ __device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__host__ __device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// k-th draw of the universal randn stream from its uint32 table.
__device__ __forceinline__ double randn_at(const uint32_t* __restrict__ tab, int k) {
  return (double)tab[k] / 268435456.0 - 6.0;
}

// Frames are dealt to the persistent workgroups one per round (runs of consecutive frames per workgroup
// were measured: 8 per run costs 15 % through voiced / unvoiced imbalance).  Within a round the
// workgroups of one XCD (workgroup id mod 8, a placement heuristic, not a guarantee) take CONSECUTIVE
// frames: neighbouring frames read almost the same samples (windows of 3-4 pitch periods every 80
// samples), so they are served by that XCD's L2 instead of every XCD fetching every sample.
__device__ __forceinline__ int64_t xcd_dealt(int64_t round_base, int64_t total) {
  const unsigned g = blockIdx.x, G = gridDim.x;
  if (G & 7u) return round_base + g;                     // grid not a multiple of 8: plain dealing
  const unsigned per = G >> 3;
  (void)total;
  return round_base + (int64_t)((g & 7u) * per + (g >> 3));
}
#define WM_FOR_EACH_FRAME(frame, total)                                                                   \
  for (int64_t base_ = 0, frame = xcd_dealt(0, (total)); base_ < (int64_t)(total);                        \
       base_ += gridDim.x, frame = xcd_dealt(base_, (total)))                                             \
    if (frame < (int64_t)(total))

// Compiler fence for the lane index inside persistent frame loops (see FftTw::fence): address
// arithmetic derived from the returned value cannot be hoisted out of the loop and spilled.
__device__ __forceinline__ int opaque_lane(int lane) {
  asm volatile("" : "+v"(lane));
  __builtin_assume((unsigned)lane < 64u);     // what the fence hides: predicates such as 2 (lane + 64 m) <= H fold again
  return lane;
}

// Compiler fence for a kernel-uniform scalar (an argument such as fs) inside persistent frame loops: whatever is
// derived from the returned value is recomputed per frame instead of being hoisted out of the loop, where it would
// hold registers -- often vector registers, for FP64 quotients -- for the whole kernel.
// The same for a per-lane double: what is derived from the returned value is not merged with what was derived from
// the argument elsewhere (common subexpressions kept alive across a transform cost registers, recomputing is cheap).
__device__ __forceinline__ double opaque_d(double v) {
  asm volatile("" : "+v"(v));
  return v;
}

__device__ __forceinline__ int opaque_uniform(int v) {
  asm volatile("" : "+s"(v));
  return v;
}

// A wave-uniform double moved into scalar registers (two v_readfirstlane): per-frame quantities such as f0, the
// frame position or a normalisation factor are computed by the vector ALU (there is no scalar FP64) and would
// otherwise occupy a vector register pair each for the whole frame.
__device__ __forceinline__ double uniform_d(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// Four consecutive doubles from an address that is only 8-byte aligned, as two 16-byte loads (global_load_dwordx4 takes
// any dword-aligned address).  For kernels in which every LANE streams through its own run of samples: a load
// instruction then costs a tag lookup per lane whatever its width, so the wider load halves the lookups per sample.
typedef double double2_a8 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void load4_a8(const double* __restrict__ p, double (&v)[4]) {
  const double2_a8 a = *reinterpret_cast<const double2_a8*>(p);
  const double2_a8 b = *reinterpret_cast<const double2_a8*>(p + 2);
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

// ---- phase timing of a per-frame kernel (debug builds only: make EXTRA=-DWM_PHASE) ------------------------------
// WM_PHASE_MARK(k) adds the shader clock since the previous mark to slot k of a per-wave array; WM_PHASE_FLUSH() adds
// the array to this translation unit's wm_phase_cycles[] in device memory (WorldMi355DebugPhases(unit, out) reads and
// clears it; tools/phase_probe.py).
// In product builds the macros are empty.
#ifdef WM_PHASE
static __device__ unsigned long long wm_phase_cycles[32];     // one per translation unit (no relocatable device code)
static inline int wm_phase_read(unsigned long long* out32) {  // host: read and clear this unit's totals
  unsigned long long zero[32] = {};
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(wm_phase_cycles), sizeof(zero)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(wm_phase_cycles), zero, sizeof(zero)) != hipSuccess;
}
struct PhaseClock {
  unsigned long long last, acc[16];
  __device__ __forceinline__ void start() {
    last = __builtin_readcyclecounter();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0;
  }
  template <int K> __device__ __forceinline__ void mark() {
    const unsigned long long now = __builtin_readcyclecounter();
    acc[K] += now - last;
    last = now;
  }
  __device__ __forceinline__ void flush(int slot0) {
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (acc[k]) atomicAdd(&wm_phase_cycles[slot0 + k], acc[k]);
    }
  }
};
#define WM_PHASE_DECL PhaseClock phase_clock_; phase_clock_.start();
#define WM_PHASE_MARK(k) phase_clock_.mark<k>();
#define WM_PHASE_FLUSH(slot0) phase_clock_.flush(slot0);
#else
#define WM_PHASE_DECL
#define WM_PHASE_MARK(k)
#define WM_PHASE_FLUSH(slot0)
#endif

// ---- wavefront collectives (64 lanes) ----------------------------------------
// Built on DPP row shifts / row broadcasts (gfx9 family) instead of ds_bpermute shuffles: six
// dependent VALU steps with no LDS round trip.  The six steps are a complete inclusive scan over
// the 64 lanes, so the same sequence serves sums, maxima (lane 63 holds the total, broadcast with
// v_readlane) and prefix sums.  Lanes that receive no source read 0.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_get(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, true);
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, true);
  return __hiloint2double(hi2, lo2);
}
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_get_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, true);
}
__device__ __forceinline__ double lane63(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
// inclusive prefix sum across lanes (lane argument kept for call-site compatibility)
__device__ __forceinline__ double wave_scan_incl(double v, int /*lane*/ = 0) {
  v += dpp_get<0x111, 0xf, 0xf>(v);      // row_shr:1
  v += dpp_get<0x112, 0xf, 0xf>(v);      // row_shr:2
  v += dpp_get<0x114, 0xf, 0xf>(v);      // row_shr:4
  v += dpp_get<0x118, 0xf, 0xf>(v);      // row_shr:8
  v += dpp_get<0x142, 0xa, 0xf>(v);      // row_bcast:15 -> rows 1, 3
  v += dpp_get<0x143, 0xc, 0xf>(v);      // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ double wave_sum(double v) { return lane63(wave_scan_incl(v)); }
// Four wave-wide sums at once (gfx950 v_permlane32_swap / v_permlane16_swap): swapping halves of
// two registers and adding reduces BOTH by one level with a single add, so the four values cost two
// levels of swaps (9 instructions), four row-rotate steps on one register and four broadcasts --
// about a third of four separate scans.  The summation tree differs from wave_sum()'s.
__device__ __forceinline__ double permswap_add32(double a, double b) {
  // vdst' = [a(0:31), b(0:31)], src' = [a(32:63), b(32:63)]
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double permswap_add16(double a, double b) {
  // vdst' = rows [a0, b0, a2, b2], src' = rows [a1, b1, a3, b3]
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wave_sum4(double& a, double& b, double& c, double& d) {
  const double ab = permswap_add32(a, b);        // lanes 0-31: partials of a, 32-63: of b
  const double cd = permswap_add32(c, d);
  double r = permswap_add16(ab, cd);             // rows: a, c, b, d (16 partials each)
  r += dpp_get<0x128, 0xf, 0xf>(r);              // row_ror:8
  r += dpp_get<0x124, 0xf, 0xf>(r);              // row_ror:4
  r += dpp_get<0x122, 0xf, 0xf>(r);              // row_ror:2
  r += dpp_get<0x121, 0xf, 0xf>(r);              // row_ror:1
  a = readlane_d(r, 0);
  c = readlane_d(r, 16);
  b = readlane_d(r, 32);
  d = readlane_d(r, 48);
}

// sum over the 16 lanes of a DPP row, returned to every lane of the row
__device__ __forceinline__ double row_sum16(double v) {
  v += dpp_get<0x128, 0xf, 0xf>(v);              // row_ror:8
  v += dpp_get<0x124, 0xf, 0xf>(v);              // row_ror:4
  v += dpp_get<0x122, 0xf, 0xf>(v);              // row_ror:2
  v += dpp_get<0x121, 0xf, 0xf>(v);              // row_ror:1
  return v;
}

// maximum of NON-NEGATIVE values (lanes without a DPP source contribute 0)
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_get<0x111, 0xf, 0xf>(v));
  v = fmax(v, dpp_get<0x112, 0xf, 0xf>(v));
  v = fmax(v, dpp_get<0x114, 0xf, 0xf>(v));
  v = fmax(v, dpp_get<0x118, 0xf, 0xf>(v));
  v = fmax(v, dpp_get<0x142, 0xa, 0xf>(v));
  v = fmax(v, dpp_get<0x143, 0xc, 0xf>(v));
  return lane63(v);
}
__device__ __forceinline__ int wave_scan_incl_i(int v, int /*lane*/ = 0) {
  v += dpp_get_i<0x111, 0xf, 0xf>(v);
  v += dpp_get_i<0x112, 0xf, 0xf>(v);
  v += dpp_get_i<0x114, 0xf, 0xf>(v);
  v += dpp_get_i<0x118, 0xf, 0xf>(v);
  v += dpp_get_i<0x142, 0xa, 0xf>(v);
  v += dpp_get_i<0x143, 0xc, 0xf>(v);
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) { return __builtin_amdgcn_readlane(wave_scan_incl_i(v), 63); }
// maximum of unsigned values over the wave (lanes without a DPP source contribute 0)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  auto mx = [](unsigned a, int b) { const unsigned ub = (unsigned)b; return a > ub ? a : ub; };
  v = mx(v, dpp_get_i<0x111, 0xf, 0xf>((int)v));
  v = mx(v, dpp_get_i<0x112, 0xf, 0xf>((int)v));
  v = mx(v, dpp_get_i<0x114, 0xf, 0xf>((int)v));
  v = mx(v, dpp_get_i<0x118, 0xf, 0xf>((int)v));
  v = mx(v, dpp_get_i<0x142, 0xa, 0xf>((int)v));
  v = mx(v, dpp_get_i<0x143, 0xc, 0xf>((int)v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// exact minimum of NON-NEGATIVE doubles over the wave (+inf for lanes that have none): the bit patterns of
// non-negative doubles order like the values, so the minimum is the maximum of the inverted patterns, high word
// first, then the low word among the lanes that hold that high word
__device__ __forceinline__ double wave_min_nonneg(double e) {
  const unsigned hi = ~(unsigned)__double2hiint(e), lo = ~(unsigned)__double2loint(e);
  const unsigned hi_max = wave_max_u32(hi);
  const unsigned lo_max = wave_max_u32(hi == hi_max ? lo : 0u);
  return __hiloint2double((int)~hi_max, (int)~lo_max);
}

// interp1Q on an LDS array (matlabfunctions.cpp:220-241): y has n entries,
// delta_y[n-1] := 0.
__device__ __forceinline__ double interp1q_lds(double x0, double dx, const double* y, int n, double xi) {
  double q = (xi - x0) / dx;
  int b = (int)q;
  double frac = q - b;
  double y0 = y[b];
  double dy = (b == n - 1) ? 0.0 : y[b + 1] - y0;
  return y0 + dy * frac;
}

// same with the reciprocal of the knot spacing supplied (the quotient may differ from the division's
// by one ulp; the interpolant is continuous across knots, so the value moves by ~1e-16 relative)
__device__ __forceinline__ double interp1q_lds_r(double x0, double inv_dx, const double* y, int n, double xi) {
  const double q = (xi - x0) * inv_dx;
  const int b = (int)q;
  const double frac = q - b;
  const double y0 = y[b];
  const double dy = (b == n - 1) ? 0.0 : y[b + 1] - y0;
  return y0 + dy * frac;
}

// DCCorrection (common.cpp:56-75) in place on pw[0..half] (LDS).  One wavefront.
// `scratch` (LDS, >= upper doubles) holds the replica so that every read of pw
// happens before any write, as in the reference.
__device__ __forceinline__ void dc_correction_lds(double* pw, double f0, int fs, int fft_size,
                                                  double* scratch, int lane) {
  const double inv_fft = 1.0 / fft_size;               // power of two: exact
  const int upper = 2 + (int)(f0 * fft_size / fs);
  const int nrep = upper - 1;
  const double inv_dx = -(double)fft_size / fs;
  for (int i = lane; i < nrep; i += 64) {
    const double axis = (double)i * fs * inv_fft;
    scratch[i] = interp1q_lds_r(f0, inv_dx, pw, upper + 1, axis);
  }
  wave_sync();
  for (int i = lane; i < nrep; i += 64) pw[i] += scratch[i];
  wave_sync();
}

// LinearSmoothing (common.cpp:77-111).  in[0..half] (LDS) -> out[0..half] (LDS; may alias in).
// seg is an LDS scratch of >= 64 * CH doubles; CH is an ODD compile-time bound on ceil(len / 64),
// len = half + 2 b + 1, b = int(width * fft_size / fs) + 1.
//
// Cumulative sum of the mirrored spectrum (the reference's is sequential, common.cpp:38-41): lane l
// sums its own CH consecutive elements in registers (an odd lane stride makes these LDS accesses
// conflict-free), one wave scan stitches the 64 partial sums.
// Interpolation (interp1Q at x -/+ width/2, common.cpp:99-108): on the uniform grid the query of bin i
// is knot i + c exactly, with c = b - 0.5 -/+ width / (2 step) the same for every bin, so the knot
// offset and the fraction are computed once per frame; the reference's per-bin quotient differs from
// i + c by rounding only, and the interpolant is continuous across knots.
// Ends with a barrier.
template <int CH>
__device__ __forceinline__ void linear_smoothing_lds(const double* in, double width, int fs, int fft_size,
                                                     double* seg, double* out, int lane) {
  static_assert(CH % 2 == 1, "odd per-lane chunk keeps the strided LDS accesses conflict-free");
  const int half = fft_size / 2;
  const double inv_fft = 1.0 / fft_size;               // power of two: x * inv_fft == x / fft_size exactly
  const double wq = width * fft_size / fs;             // width in bins
  const int b = (int)wq + 1;
  const int len = half + 2 * b + 1;
  const int beg = lane * CH;
  double v[CH];
#pragma unroll
  for (int q = 0; q < CH; ++q) {
    const int i = beg + q;
    int src = i < b ? b - i : (i < half + b ? i - b : half - (i - (half + b)));
    src = imax(0, imin(half, src));
    v[q] = in[src];
  }
#pragma unroll
  for (int q = 0; q < CH; ++q) {
    const double term = (beg + q < len) ? v[q] * fs * inv_fft : 0.0;
    v[q] = q == 0 ? term : v[q - 1] + term;
  }
  const double carry = wave_scan_incl(v[CH - 1]) - v[CH - 1];
  // unconditional: seg has 64 * CH entries, the ones beyond len are never read (a predicate per element
  // turns into CH exec-mask branches)
#pragma unroll
  for (int q = 0; q < CH; ++q) seg[beg + q] = v[q] + carry;
  wave_sync();
  const double c_lo = (b - 0.5) - 0.5 * wq, c_hi = c_lo + wq;
  const int bl = (int)c_lo, bh = (int)c_hi;
  const double fl = c_lo - bl, fh = c_hi - bh;
  const double inv_width = 1.0 / width;
  constexpr int GI = 8;
  int i0 = 0;
  for (; i0 + 64 * GI <= half + 1; i0 += 64 * GI) {          // full groups: no predicates
    double lo0[GI], lo1[GI], hi0[GI], hi1[GI];
#pragma unroll
    for (int q = 0; q < GI; ++q) {
      const int i = i0 + 64 * q + lane;
      lo0[q] = seg[i + bl]; lo1[q] = seg[i + bl + 1];
      hi0[q] = seg[i + bh]; hi1[q] = seg[i + bh + 1];
    }
#pragma unroll
    for (int q = 0; q < GI; ++q) {
      const double lo = lo0[q] + (lo1[q] - lo0[q]) * fl;
      const double hi = hi0[q] + (hi1[q] - hi0[q]) * fh;
      out[i0 + 64 * q + lane] = (hi - lo) * inv_width;
    }
  }
  for (int i = i0 + lane; i <= half; i += 64) {               // tail (bin half of a power-of-two spectrum)
    const double lo = seg[i + bl] + (seg[i + bl + 1] - seg[i + bl]) * fl;
    const double hi = seg[i + bh] + (seg[i + bh + 1] - seg[i + bh]) * fh;
    out[i] = (hi - lo) * inv_width;
  }
  wave_sync();
}

}  // namespace wm

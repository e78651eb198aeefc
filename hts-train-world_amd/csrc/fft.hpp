// fft.hpp -- one-wavefront FP64 FFTs for gfx950 (CDNA4), LDS-exchanged.
//
// Replaces the reference's Ooura rdft/cdft + FFTW-style wrappers
// (externs/WORLD_v2/src/fft.cpp:26-166) for every per-frame transform on the
// hot path.  Only the wrapper CONVENTIONS are kept (r2c = e^{-j} half spectrum,
// c2r = unnormalised inverse ignoring Im(DC)/Im(Nyquist)); the factorisation is
// this file's own:
//
//   * one 64-lane wavefront owns one transform of N complex points
//     (N = 512/1024/2048; 256 rides on the 512-point plan, see the end of the file),
//     lane l holding elements l + 64*m in registers;
//   * three Stockham passes (radix 4/8/16) with exactly two LDS exchanges;
//     pass-1 input and pass-3 output never touch LDS;
//   * LDS image padded by one element per R1 so the stride-R1 stores of pass 1
//     spread over all 32 store banks (ds_write_b128: 8-lane groups);
//   * no twiddle tables: per-lane base twiddles come from sincospi() once per
//     kernel, powers by short complex-multiply chains.
//
// A workgroup is ONE wavefront (__launch_bounds__(64)), so __syncthreads()
// lowers to a wave-local fence; every LDS exchange below is bracketed by it.
#pragma once
#include <hip/hip_runtime.h>

#include "wavesync.hpp"

namespace wm {

typedef double2 cpx;

__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cpx cmul(cpx a, cpx b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cpx cconj(cpx a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ cpx cmul_mi(cpx a) { return make_double2(a.y, -a.x); }   // a * (-i)
__device__ __forceinline__ cpx cis_neg2pi(double frac) {   // exp(-2*pi*i*frac)
  double s, c;
  sincospi(2.0 * frac, &s, &c);
  return make_double2(c, -s);
}

// ---- in-register DFTs, natural order in and out -----------------------------
// Decimation in time by halves, every butterfly in fused form:
//
//   X[q]         = E[q] + t_q O[q]            E, O: the transforms of the even / odd inputs
//   X[q + R / 2] = E[q] - t_q O[q]            t_q = W_R^q (Dft) or w W_R^q (TwDft)
//
// * a constant t = (c, -s) is factored by its larger component: x +- c (y.x + tau y.y), tau = s / c a compile-time
//   constant -- two fused multiply-adds for the bracket and one per output component: six for a butterfly where
//   the product and the sum / difference take eight;
// * a per-lane t (TwDft: the inter-pass twiddles w^r are folded into the butterflies, nothing is multiplied
//   beforehand) has no tangent at hand: the sum is two chained fused multiply-adds per component and the
//   difference is 2 x - sum (one): six as well;
// * TwDft needs w, w^2, w^4 (, w^8) and the products w W_R^q of the last level only (q and q + R / 4 share one:
//   W_R^(R/4) = -i) -- three to five live twiddles where a table of powers w^1 .. w^(R-1) held seven to fifteen.
// 1024 points: 499 FP64 instructions (round 3: 612, of which 97 built powers and 112 applied them).
template <int K16> struct W16c {                    // W_16^K = (cos, -sin)(2 pi K / 16), K = 0 .. 4 tabulated
  static constexpr double kC[5] = {1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173, 0.0};
  static constexpr double c = kC[K16], s = kC[4 - K16];
};

// x, y  <-  x + t y, x - t y   with t = W_16^K16 (0 <= K16 < 8), compile-time
template <int K16> __device__ __forceinline__ void bfly_const(cpx& x, cpx& y) {
  if constexpr (K16 == 0) {
    const cpx a = x;
    x = cadd(a, y);
    y = csub(a, y);
  } else if constexpr (K16 == 4) {                  // t = -i: t y = (y.y, -y.x)
    const cpx a = x, b = y;
    x = make_double2(a.x + b.y, a.y - b.x);
    y = make_double2(a.x - b.y, a.y + b.x);
  } else if constexpr (K16 > 4) {                   // W^K = -i W^(K-4): t y = (u.y, -u.x), u = W^(K-4) y
    constexpr double c = W16c<K16 - 4>::c, s = W16c<K16 - 4>::s;
    const cpx a = x, b = y;
    if constexpr (c >= s) {
      constexpr double tau = s / c;
      const double ur = __builtin_fma(tau, b.y, b.x), ui = __builtin_fma(-tau, b.x, b.y);   // u / c
      x = make_double2(__builtin_fma(c, ui, a.x), __builtin_fma(-c, ur, a.y));
      y = make_double2(__builtin_fma(-c, ui, a.x), __builtin_fma(c, ur, a.y));
    } else {
      constexpr double kap = c / s;
      const double ur = __builtin_fma(kap, b.x, b.y), ui = __builtin_fma(kap, b.y, -b.x);   // u / s
      x = make_double2(__builtin_fma(s, ui, a.x), __builtin_fma(-s, ur, a.y));
      y = make_double2(__builtin_fma(-s, ui, a.x), __builtin_fma(s, ur, a.y));
    }
  } else {
    constexpr double c = W16c<K16>::c, s = W16c<K16>::s;
    const cpx a = x, b = y;
    if constexpr (c >= s) {                         // t y = c (b.x + tau b.y, b.y - tau b.x)
      constexpr double tau = s / c;
      const double ur = __builtin_fma(tau, b.y, b.x), ui = __builtin_fma(-tau, b.x, b.y);
      x = make_double2(__builtin_fma(c, ur, a.x), __builtin_fma(c, ui, a.y));
      y = make_double2(__builtin_fma(-c, ur, a.x), __builtin_fma(-c, ui, a.y));
    } else {                                        // t y = s (kap b.x + b.y, kap b.y - b.x)
      constexpr double kap = c / s;
      const double ur = __builtin_fma(kap, b.x, b.y), ui = __builtin_fma(kap, b.y, -b.x);
      x = make_double2(__builtin_fma(s, ur, a.x), __builtin_fma(s, ui, a.y));
      y = make_double2(__builtin_fma(-s, ur, a.x), __builtin_fma(-s, ui, a.y));
    }
  }
}

// x, y  <-  x + t y, x - t y   with a per-lane t
__device__ __forceinline__ void bfly_tw(cpx& x, cpx& y, cpx t) {
  const double sx = __builtin_fma(y.x, t.x, __builtin_fma(-y.y, t.y, x.x));
  const double sy = __builtin_fma(y.x, t.y, __builtin_fma(y.y, t.x, x.y));
  y = make_double2(__builtin_fma(2.0, x.x, -sx), __builtin_fma(2.0, x.y, -sy));
  x = make_double2(sx, sy);
}

// w W_16^K16, K16 = 0 .. 3 (the others follow by -i)
template <int K16> __device__ __forceinline__ cpx mul_w16(cpx w) {
  if constexpr (K16 == 0) {
    return w;
  } else if constexpr (K16 == 2) {
    constexpr double h = W16c<2>::c;
    return make_double2(h * (w.x + w.y), h * (w.y - w.x));
  } else {
    constexpr double c = W16c<K16>::c, s = W16c<K16>::s;
    return make_double2(__builtin_fma(w.x, c, w.y * s), __builtin_fma(w.y, c, -(w.x * s)));
  }
}
__device__ __forceinline__ cpx csqr(cpx w) { return make_double2(__builtin_fma(w.x, w.x, -(w.y * w.y)), (w.x + w.x) * w.y); }

template <int R> struct Dft {
  __device__ static __forceinline__ void run(cpx (&v)[R]) {
    if constexpr (R == 2) {
      bfly_const<0>(v[0], v[1]);
    } else {
      cpx e[R / 2], o[R / 2];
#pragma unroll
      for (int r = 0; r < R / 2; ++r) {
        e[r] = v[2 * r];
        o[r] = v[2 * r + 1];
      }
      Dft<R / 2>::run(e);
      Dft<R / 2>::run(o);
      combine<0>(e, o);
#pragma unroll
      for (int q = 0; q < R / 2; ++q) {
        v[q] = e[q];
        v[q + R / 2] = o[q];
      }
    }
  }
  template <int Q> __device__ static __forceinline__ void combine(cpx (&e)[R / 2], cpx (&o)[R / 2]) {
    if constexpr (Q < R / 2) {
      bfly_const<Q * (16 / R)>(e[Q], o[Q]);
      combine<Q + 1>(e, o);
    }
  }
};

// The same with only the first NZ inputs possibly non-zero (the analysis frames are a few hundred samples in a
// transform of one or two thousand: most of what the first pass reads is padding).  By halves as above: the even inputs
// have ceil(NZ / 2) leading non-zeros, the odd ones NZ / 2; a sub-transform of one non-zero input is that input at every
// output, of none it is zero; the combining butterflies stay (their operands are dense).  Radix 16: 32 butterflies
// become 28 / 24 / 20 / 16 / 8 / 0 for NZ = 12 / 8 / 6 / 4 / 2 / 1.  v[r], r >= NZ, is not read.
template <int R, int NZ> struct DftNz {
  __device__ static __forceinline__ void run(cpx (&v)[R]) {
    if constexpr (NZ >= R) {
      Dft<R>::run(v);
    } else if constexpr (NZ <= 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) v[r] = make_double2(0.0, 0.0);
    } else if constexpr (NZ == 1) {
#pragma unroll
      for (int r = 1; r < R; ++r) v[r] = v[0];
    } else {
      cpx e[R / 2], o[R / 2];
#pragma unroll
      for (int r = 0; r < R / 2; ++r) {
        e[r] = v[2 * r];
        o[r] = v[2 * r + 1];
      }
      DftNz<R / 2, (NZ + 1) / 2>::run(e);
      DftNz<R / 2, NZ / 2>::run(o);
      Dft<R>::template combine<0>(e, o);
#pragma unroll
      for (int q = 0; q < R / 2; ++q) {
        v[q] = e[q];
        v[q + R / 2] = o[q];
      }
    }
  }
};

// X[q] = sum_r v[r] w^r W_R^(r q): the DFT of inputs that still lack their inter-pass twiddles w^r.
// pw[k] = w^(2^k), k = 0 .. log2(R) - 1.
template <int R> struct TwDft {
  static constexpr int L = R == 2 ? 1 : (R == 4 ? 2 : (R == 8 ? 3 : 4));
  __device__ static __forceinline__ void run(cpx (&v)[R], const cpx (&pw)[4]) { level<R>(v, pw); }
  // level<RR>: v holds RR inputs that are every (R / RR)-th input of the whole transform; their twiddle base is
  // w^(R / RR) = pw[log2(R / RR)]
  template <int RR> __device__ static __forceinline__ void level(cpx (&v)[RR], const cpx (&pw)[4]) {
    constexpr int LG = R / RR == 1 ? 0 : (R / RR == 2 ? 1 : (R / RR == 4 ? 2 : 3));
    if constexpr (RR == 2) {
      bfly_tw(v[0], v[1], pw[LG]);
    } else {
      cpx e[RR / 2], o[RR / 2];
#pragma unroll
      for (int r = 0; r < RR / 2; ++r) {
        e[r] = v[2 * r];
        o[r] = v[2 * r + 1];
      }
      level<RR / 2>(e, pw);
      level<RR / 2>(o, pw);
      comb<RR, 0>(e, o, pw[LG]);
#pragma unroll
      for (int q = 0; q < RR / 2; ++q) {
        v[q] = e[q];
        v[q + RR / 2] = o[q];
      }
    }
  }
  // t_q = w W_RR^q for q < RR / 4 (computed), t_(q + RR / 4) = -i t_q
  template <int RR, int Q> __device__ static __forceinline__ void comb(cpx (&e)[RR / 2], cpx (&o)[RR / 2], cpx w) {
    if constexpr (RR == 2) {
      bfly_tw(e[0], o[0], w);
    } else if constexpr (Q < RR / 4) {
      const cpx t = mul_w16<Q * (16 / RR)>(w);
      bfly_tw(e[Q], o[Q], t);
      bfly_tw(e[Q + RR / 4], o[Q + RR / 4], make_double2(t.y, -t.x));
      comb<RR, Q + 1>(e, o, w);
    }
  }
};

#ifndef WM_FFT_ILP
#define WM_FFT_ILP 1
#endif
// Between the LDS exchanges of ONE wavefront's transform nothing needs to be waited for: a wave's DS instructions
// execute in issue order, so a read that follows the writes of the same wave sees them, and writes that follow reads
// do not overtake them.  What is needed is that the COMPILER keeps that order (lanes read what other lanes wrote:
// to the compiler those are unrelated addresses).  WM_FFT_LIGHT_SYNC = 1: a compiler barrier only; 0: wave_sync(),
// i.e. s_waitcnt lgkmcnt(0) at every exchange (all writes drained before the first read is issued).
#ifndef WM_FFT_LIGHT_SYNC
#define WM_FFT_LIGHT_SYNC 0
#endif
__device__ __forceinline__ void fft_sync() {
#if WM_FFT_LIGHT_SYNC
  asm volatile("" ::: "memory");
#else
  wave_sync();
#endif
}

template <int N> struct FftCfg;
template <> struct FftCfg<512>  { static constexpr int R1 = 8,  R2 = 8,  R3 = 8; };
#ifndef WM_FFT_1024_SWAP
#define WM_FFT_1024_SWAP 1
#endif
#if WM_FFT_1024_SWAP
template <> struct FftCfg<1024> { static constexpr int R1 = 16, R2 = 16, R3 = 4; };   // second exchange by permlane swaps
#else
template <> struct FftCfg<1024> { static constexpr int R1 = 16, R2 = 8,  R3 = 8; };
#endif
template <> struct FftCfg<2048> { static constexpr int R1 = 16, R2 = 16, R3 = 8; };

// LDS doubles2 needed by one transform of N points (padded image)
template <int N> struct FftLds { static constexpr int kElems = N + N / FftCfg<N>::R1; };

// exp(-2 pi i k / 64), k = 0..16 (a quarter turn): the lane-uniform twiddle factors below are entries of it
__device__ __forceinline__ cpx cis64(int k) {
  constexpr double c[17] = {1.0, 0.9951847266721969, 0.9807852804032304, 0.9569403357322088, 0.9238795325112867,
                            0.881921264348355, 0.8314696123025452, 0.773010453362737, 0.7071067811865476,
                            0.6343932841636455, 0.5555702330196022, 0.47139673682599764, 0.3826834323650898,
                            0.2902846772544624, 0.19509032201612828, 0.0980171403295606, 0.0};
  return make_double2(c[k], -c[16 - k]);
}

// Per-lane twiddle bases, computed once per kernel.  Only what depends on the lane in a way that cannot be
// derived cheaply stays in registers for the whole kernel (two complex values); the rest is rebuilt per use:
//   W_N^(lane + 64 b) = (W_2N^lane)^2 * W_N^(64 b), the second factor a compile-time constant;
//   W_2N^64 (the step of the real-FFT split twiddles) is a compile-time constant.
// Persistent frame loops keep these alive across everything else a frame needs, so every register here is one
// less for the frame (d4c_kernel was spilling exactly these).
template <int N> struct FftTw {
  static constexpr int M = N / 64;
  static constexpr int S3 = M / FftCfg<N>::R3;
  static_assert(N >= 512 && 2048 % N == 0, "constant twiddles are tabulated in 64ths of a turn");
  cpx w2;            // W_{R1*R2}^(lane % R1)
  cpx wsplit;        // W_{2N}^lane          (real-FFT split; its square is W_N^lane)
  __device__ __forceinline__ void init(int lane) {
    constexpr int R1 = FftCfg<N>::R1, R2 = FftCfg<N>::R2;
    w2 = cis_neg2pi((double)(lane % R1) / (double)(R1 * R2));
    wsplit = cis_neg2pi((double)lane / (double)(2 * N));
  }
  // W_N^(lane + 64 b): pass-3 twiddle base of butterfly b
  __device__ __forceinline__ cpx w3(int b) const {
    const cpx sq = make_double2(wsplit.x * wsplit.x - wsplit.y * wsplit.y, 2.0 * wsplit.x * wsplit.y);
    return b == 0 ? sq : cmul(sq, cis64(b * (4096 / N)));          // 64 b / N turns = b * 4096 / N 64ths
  }
  // W_{2N}^64: chain step of the split twiddles (64 / 2N turns = 2048 / N 64ths)
  __device__ __forceinline__ cpx wstep() const { return cis64(2048 / N); }
  // Compiler fence for persistent (grid-stride) kernels: called at the top of every frame, it makes
  // the twiddle bases opaque so that nothing derived from them (powers, products) is hoisted out of
  // the frame loop -- LICM otherwise precomputes dozens of twiddle powers and LDS addresses once per
  // kernel and then spills them, which costs far more than recomputing a few FMAs per transform.
  __device__ __forceinline__ void fence() {
    asm volatile("" : "+v"(w2.x), "+v"(w2.y), "+v"(wsplit.x), "+v"(wsplit.y));
  }
};

// unsigned on purpose: with signed ints the compiler cannot prove (b + R1 * r) / R1 == b / R1 + r and emits
// a full index computation per element instead of one base plus immediate offsets
template <int N> __device__ __forceinline__ unsigned fft_pad(unsigned i) { return i + i / (unsigned)FftCfg<N>::R1; }

// pass 1 of a transform whose inputs are zero from element 64 NZM on: butterfly b (of S1) holds the inputs
// b + r S1, of which the first ceil((NZM - b) / S1) may be non-zero.  b is a loop counter (a constant once unrolled):
// the switch folds.
template <int R1, int S1, int NZM> __device__ __forceinline__ void fft_pass1_pruned(cpx (&a)[R1], int b) {
  static_assert(S1 <= 2, "radix plans here have one or two first-pass butterflies per lane");
  constexpr int NZ0 = (NZM + S1 - 1) / S1 < R1 ? (NZM + S1 - 1) / S1 : R1;                     // b = 0
  constexpr int NZ1 = NZM - 1 <= 0 ? 0 : ((NZM - 1 + S1 - 1) / S1 < R1 ? (NZM - 1 + S1 - 1) / S1 : R1);   // b = 1
  if (b == 0) DftNz<R1, NZ0>::run(a);
  else DftNz<R1, NZ1>::run(a);
}

// The three passes of the forward complex FFT (e^{-j}) of one wavefront.  v[m] holds element lane + 64 m on entry
// and on exit; `lds` must provide FftLds<N>::kElems cpx.  Caller guarantees no other use of `lds` is in flight (ends
// with registers only; starts with a barrier).
// fft_pass1<N, NZM>: the first pass, registers -> LDS image.  NZM: only v[0 .. NZM) may be non-zero on entry (elements
// below 64 NZM); the others are not read.  Everything the pass produces is in LDS when it ends, so the variants of a
// run-time choice (fft_forward_nz) meet with no register live across them.
template <int N, int NZM>
__device__ __forceinline__ void fft_pass1(const cpx (&v)[N / 64], cpx* lds, int lane) {
  constexpr int M = N / 64;
  constexpr int R1 = FftCfg<N>::R1;
  constexpr int S1 = M / R1;
  static_assert(S1 >= 1, "radix plan does not fit 64 lanes");
  fft_sync();
  // ---- pass 1: radix R1, Ns = 1, no twiddles; store to out[j*R1 + r]
#pragma unroll
  for (int b = 0; b < S1; ++b) {
    cpx a[R1];
    if constexpr (NZM >= M) {
#pragma unroll
      for (int r = 0; r < R1; ++r) a[r] = v[b + r * S1];
      Dft<R1>::run(a);
    } else {
#pragma unroll
      for (int r = 0; r < R1; ++r) a[r] = b + r * S1 < NZM ? v[b + r * S1] : make_double2(0.0, 0.0);
      fft_pass1_pruned<R1, S1, NZM>(a, b);
    }
    const unsigned j = (unsigned)lane + 64u * b;
#pragma unroll
    for (int r = 0; r < R1; ++r) lds[fft_pad<N>(j * R1 + r)] = a[r];
#if !WM_FFT_ILP
    __builtin_amdgcn_sched_barrier(0);      // one butterfly at a time: bounds the live registers
#endif
  }
}

// Two registers trade halves (v_permlane32_swap: a' = [a(0:31), b(0:31)], b' = [a(32:63), b(32:63)]) or odd / even
// rows of sixteen lanes (v_permlane16_swap: a' = rows [a0, b0, a2, b2], b' = rows [a1, b1, a3, b3]): a transposition of
// one LANE bit (5 resp. 4) with the bit that tells the two registers apart, one instruction per pair of dwords.
__device__ __forceinline__ void lane_swap32(double& a, double& b) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi[0], (int)lo[0]);
  b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void lane_swap16(double& a, double& b) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi[0], (int)lo[0]);
  b = __hiloint2double((int)hi[1], (int)lo[1]);
}

// passes 2 and 3 of the 16 x 16 x 4 plan (N = 1024): LDS image -> registers with ONE trip through LDS.
// Pass 2 is one radix-16 butterfly per lane whose output r belongs at image position
//   (lane / 16) 256 + lane % 16 + 16 r,
// and butterfly b of pass 3 in lane L reads the positions L + 64 b + 256 r3, r3 < 4: the value comes from lane
// L % 16 + 16 r3, register r = L / 16 + 4 b.  That second exchange moves nothing between the sixteen-lane rows'
// COLUMNS: it is the transposition of lane bits (4, 5) with register bits (0, 1), i.e. two rounds of permlane swaps
// (64 VALU instructions) instead of sixteen ds_write_b128 (13 LDS cycles each on gfx950, MI355X_MICROARCH.md) and
// sixteen ds_read_b128 on a CU whose LDS pipe the transforms keep busy -- and one
// wait for LDS less on the transform's critical path.  After the swaps register slot r3 + 4 b holds input r3 of
// butterfly b.
template <int N>
__device__ __forceinline__ void fft_finish_swap4(cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw, int lane) {
  constexpr int M = N / 64;
  static_assert(M == 16 && FftCfg<N>::R1 == 16 && FftCfg<N>::R2 == 16 && FftCfg<N>::R3 == 4, "16 x 16 x 4");
  fft_sync();
  cpx a[16];
#pragma unroll
  for (int m = 0; m < M; ++m) a[m] = lds[fft_pad<N>((unsigned)lane + 64u * m)];
  fft_sync();
  {
    cpx pw[4];
    pw[0] = tw.w2;
    pw[1] = csqr(pw[0]);
    pw[2] = csqr(pw[1]);
    pw[3] = csqr(pw[2]);
    TwDft<16>::run(a, pw);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if ((r & 2) == 0) {
      lane_swap32(a[r].x, a[r + 2].x);
      lane_swap32(a[r].y, a[r + 2].y);
    }
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if ((r & 1) == 0) {
      lane_swap16(a[r].x, a[r + 1].x);
      lane_swap16(a[r].y, a[r + 1].y);
    }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    cpx c[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = a[r + 4 * b];
    cpx pw[4];
    pw[0] = tw.w3(b);
    pw[1] = csqr(pw[0]);
    pw[2] = pw[1];
    pw[3] = pw[1];
    TwDft<4>::run(c, pw);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[b + 4 * r] = c[r];
  }
}

// passes 2 and 3: LDS image -> registers
template <int N>
__device__ __forceinline__ void fft_finish(cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw, int lane) {
  constexpr int M = N / 64;
  constexpr int R1 = FftCfg<N>::R1, R2 = FftCfg<N>::R2, R3 = FftCfg<N>::R3;
  constexpr int S2 = M / R2, S3 = M / R3;
  static_assert(S2 >= 1 && S3 >= 1, "radix plan does not fit 64 lanes");
  static_assert(R1 * R2 * R3 == N, "radix plan");
  if constexpr (R3 == 4 && M == 16) {
    fft_finish_swap4<N>(v, lds, tw, lane);
    return;
  }
  fft_sync();
#pragma unroll
  for (int m = 0; m < M; ++m) v[m] = lds[fft_pad<N>((unsigned)lane + 64u * m)];
  fft_sync();
  // ---- pass 2: radix R2, Ns = R1; twiddle W_{R1 R2}^{k r}, k = j % R1 = lane % R1
#pragma unroll
  for (int b = 0; b < S2; ++b) {
    cpx a[R2];
#pragma unroll
    for (int r = 0; r < R2; ++r) a[r] = v[b + r * S2];
    {
      cpx pw[4];
      pw[0] = tw.w2;
      pw[1] = csqr(pw[0]);
      pw[2] = csqr(pw[1]);
      pw[3] = R2 > 8 ? csqr(pw[2]) : pw[2];
      TwDft<R2>::run(a, pw);
    }
    const unsigned j = (unsigned)lane + 64u * b;
    const unsigned base = (j / R1) * (R1 * R2) + (j % R1);
#pragma unroll
    for (int r = 0; r < R2; ++r) lds[fft_pad<N>(base + (unsigned)r * R1)] = a[r];
#if !WM_FFT_ILP
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
  fft_sync();
#pragma unroll
  for (int m = 0; m < M; ++m) v[m] = lds[fft_pad<N>((unsigned)lane + 64u * m)];
  // ---- pass 3: radix R3, Ns = N / R3; twiddle W_N^{j r}; output lands in the register layout
#pragma unroll
  for (int b = 0; b < S3; ++b) {
    cpx a[R3];
#pragma unroll
    for (int r = 0; r < R3; ++r) a[r] = v[b + r * S3];
    {
      cpx pw[4];
      pw[0] = tw.w3(b);
      pw[1] = csqr(pw[0]);
      pw[2] = csqr(pw[1]);
      pw[3] = R3 > 8 ? csqr(pw[2]) : pw[2];
      TwDft<R3>::run(a, pw);
    }
#pragma unroll
    for (int r = 0; r < R3; ++r) v[b + r * S3] = a[r];
#if !WM_FFT_ILP
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
}

template <int N, int NZM = N / 64>
__device__ __forceinline__ void fft_forward(cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw, int lane) {
  // per-call fences: every transform derives its own LDS addresses and twiddle powers (a few
  // integer ops / FMAs) instead of sharing hoisted copies that the register allocator then spills
  asm volatile("" : "+v"(lane));
  const_cast<FftTw<N>&>(tw).fence();
  fft_pass1<N, NZM>(v, lds, lane);
  fft_finish<N>(v, lds, tw, lane);
}

// fft_forward with the number of leading registers that may be non-zero known at run time (wave-uniform `nz`: the
// window of an analysis frame): the first pass comes in four sizes.
template <int N>
__device__ __forceinline__ void fft_forward_nz(cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw, int lane, int nz) {
  constexpr int M = N / 64;
  if constexpr (N < 512) {                               // 256 points ride on the 512-point plan (end of the file)
    fft_forward<N>(v, lds, tw, lane);
  } else {
    asm volatile("" : "+v"(lane));
    const_cast<FftTw<N>&>(tw).fence();
    if (8 * nz <= M) fft_pass1<N, M / 8>(v, lds, lane);
    else if (4 * nz <= M) fft_pass1<N, M / 4>(v, lds, lane);
    else if (2 * nz <= M) fft_pass1<N, M / 2>(v, lds, lane);
    else fft_pass1<N, M>(v, lds, lane);
    fft_finish<N>(v, lds, tw, lane);
  }
}

// Unnormalised inverse complex FFT (e^{+j}) via conj . forward . conj.
template <int N>
__device__ __forceinline__ void fft_backward(cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw, int lane) {
#pragma unroll
  for (int m = 0; m < N / 64; ++m) v[m].y = -v[m].y;
  fft_forward<N>(v, lds, tw, lane);
#pragma unroll
  for (int m = 0; m < N / 64; ++m) v[m].y = -v[m].y;
}

// ---- real transforms of length F = 2N ----------------------------------------
// r2c: on entry v[m] = (x[2n], x[2n+1]), n = lane + 64 m.  On exit the half
// spectrum X[0..N] is in `spec` (LDS, N+1 cpx, plain layout): spec may alias the
// FFT image (it is only written after the last exchange).
// The split after the complex transform (shared by rfft_forward and rfft_forward_nz).
template <int N>
__device__ __forceinline__ void rfft_split(cpx (&v)[N / 64], cpx* lds, cpx* spec, const FftTw<N>& tw, int lane);

template <int N, int NZM = N / 64>
__device__ __forceinline__ void rfft_forward(cpx (&v)[N / 64], cpx* lds, cpx* spec, const FftTw<N>& tw,
                                             int lane) {
  fft_forward<N, NZM>(v, lds, tw, lane);
  rfft_split<N>(v, lds, spec, tw, lane);
}
// nz: the packed registers v[0 .. nz) may be non-zero (wave-uniform)
template <int N>
__device__ __forceinline__ void rfft_forward_nz(cpx (&v)[N / 64], cpx* lds, cpx* spec, const FftTw<N>& tw, int lane,
                                                int nz) {
  fft_forward_nz<N>(v, lds, tw, lane, nz);
  rfft_split<N>(v, lds, spec, tw, lane);
}

template <int N>
__device__ __forceinline__ void rfft_split(cpx (&v)[N / 64], cpx* lds, cpx* spec, const FftTw<N>& tw, int lane) {
  constexpr int M = N / 64;
  asm volatile("" : "+v"(lane));
  wave_sync();
#pragma unroll
  for (int m = 0; m < M; ++m) lds[lane + 64 * m] = v[m];     // Z, plain layout
  wave_sync();
  cpx xk[M];
  // X[k] = (a + b) / 2 + W_2N^k (a - b) / (2 i), a = Z[k], b = conj Z[N - k]: with the half folded into the twiddle
  // three fused multiply-adds per component.  W_2N^(k + N / 2) = -i W_2N^k: the lane's twiddles of the upper half of
  // its elements are those of the lower half turned, so M / 2 - 1 products with constants serve M elements.
  cpx wh[M / 2 > 0 ? M / 2 : 1];
  wh[0] = make_double2(0.5 * tw.wsplit.x, 0.5 * tw.wsplit.y);
#pragma unroll
  for (int m = 1; m < M / 2; ++m) wh[m] = cmul(wh[0], cis64(m * (2048 / N)));
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int k = lane + 64 * m;
    const cpx a = v[m];
    const cpx bz = lds[(N - k) & (N - 1)];                     // b = conj(bz)
    const cpx w = m < M / 2 ? wh[m] : make_double2(wh[m - M / 2].y, -wh[m - M / 2].x);
    const double sx = a.x + bz.x, sy = a.y - bz.y, dx = a.x - bz.x, dy = a.y + bz.y;
    xk[m] = make_double2(__builtin_fma(0.5, sx, __builtin_fma(w.x, dy, w.y * dx)),
                         __builtin_fma(0.5, sy, __builtin_fma(w.y, dy, -(w.x * dx))));
  }
  cpx z0 = lds[0];
  wave_sync();
#pragma unroll
  for (int m = 0; m < M; ++m) spec[lane + 64 * m] = xk[m];
  if (lane == 0) {
    spec[0] = make_double2(z0.x + z0.y, 0.0);
    spec[N] = make_double2(z0.x - z0.y, 0.0);
  }
  wave_sync();
}

// The forward real transform for consumers that take every bin where the split computes it (a power spectrum): the
// plain form stores the half spectrum to LDS only for the caller to load the same values back into the same lanes --
// sixteen ds_write_b128 (13 LDS cycles each on gfx950, MI355X_MICROARCH.md) and sixteen reads per transform on a CU whose
// LDS pipe the transforms' exchanges already keep busy.  Here it stays in registers, and
// the split goes by PAIRS: X[k] and X[N - k] are the same four sums and differences of Z[k] and Z[N - k] under twiddles that
// differ in sign only, so the lane that holds Z[k], k = lane + 64 m < N / 2, computes both:
//   xk[m] = X[k],  xr[m] = X[N - k]   (lane 0, m = 0: X[0] and X[N]),   m < M / 2;   xh = X[N / 2] in lane 0.
// Only the upper half of Z goes through LDS (half the stores, reads and twiddles of rfft_split); the bins
// above N / 2 come out in the lanes of their partners, which is all the same to a consumer that stores them by index or
// sums / sorts them.
// The functor forms hand each pair on as it is computed -- f(m, X[k], X[N - k]) for m < M / 2, then f(M / 2, X[N / 2],
// X[N / 2]) (lane 0's is the bin) -- so that a consumer which stores or transforms the values at once keeps neither
// array alive; the array forms below are wrappers.
template <int N, class F>
__device__ __forceinline__ void rfft_split_pairs_f(const cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw, int lane, F f) {
  constexpr int M = N / 64;
  asm volatile("" : "+v"(lane));
  wave_sync();
#pragma unroll
  for (int m = M / 2; m < M; ++m) lds[lane + 64 * m] = v[m];
  wave_sync();
  cpx wh = make_double2(0.5 * tw.wsplit.x, 0.5 * tw.wsplit.y);          // W_2N^k / 2, k = lane + 64 m
#pragma unroll
  for (int m = 0; m < M / 2; ++m) {
    const int k = lane + 64 * m;
    const cpx a = v[m];
    cpx bz = lds[(N - k) & (N - 1)];
    if (m == 0) {                                                       // Z[0] pairs with itself (not stored)
      bz.x = lane == 0 ? a.x : bz.x;
      bz.y = lane == 0 ? a.y : bz.y;
    }
    const cpx w = m == 0 ? wh : cmul(wh, cis64(m * (2048 / N)));
    const double sx = a.x + bz.x, sy = a.y - bz.y, dx = a.x - bz.x, dy = a.y + bz.y;
    const double pr = __builtin_fma(w.x, dy, w.y * dx), pi = __builtin_fma(w.y, dy, -(w.x * dx));
    f(m, make_double2(__builtin_fma(0.5, sx, pr), __builtin_fma(0.5, sy, pi)),
      make_double2(__builtin_fma(0.5, sx, -pr), __builtin_fma(-0.5, sy, pi)));
  }
  const cpx xh = make_double2(v[M / 2].x, -v[M / 2].y);                 // lane 0: X[N / 2] = conj Z[N / 2]
  f(M / 2, xh, xh);
  wave_sync();
}
template <int N>
__device__ __forceinline__ void rfft_split_pairs(const cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw, int lane,
                                                 cpx (&xk)[N / 128], cpx (&xr)[N / 128], cpx& xh) {
  rfft_split_pairs_f<N>(v, lds, tw, lane, [&](int m, cpx a, cpx b) {
    if (m < N / 128) {
      xk[m < N / 128 ? m : 0] = a;
      xr[m < N / 128 ? m : 0] = b;
    } else {
      xh = a;
    }
  });
}
template <int N>
__device__ __forceinline__ void rfft_forward_nz_pairs(cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw, int lane, int nz,
                                                      cpx (&xk)[N / 128], cpx (&xr)[N / 128], cpx& xh) {
  fft_forward_nz<N>(v, lds, tw, lane, nz);
  rfft_split_pairs<N>(v, lds, tw, lane, xk, xr, xh);
}

// c2r (unnormalised; fft.cpp:27-35 semantics): X[0..N] in `spec` (LDS, plain).
// On exit v[m] = (x[2n], x[2n+1]), n = lane + 64 m.  spec may alias `lds`.
template <int N>
__device__ __forceinline__ void rfft_backward(const cpx* spec, cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw,
                                              int lane) {
  constexpr int M = N / 64;
  asm volatile("" : "+v"(lane));
  const_cast<FftTw<N>&>(tw).fence();
  // v = (a + b) + j (a - b) conj(W_2N^k), a = X[k], b = conj X[N - k]: two chained fused multiply-adds per component;
  // the twiddles of the upper half of a lane's elements are those of the lower half turned (see rfft_forward)
  cpx wc[M / 2 > 0 ? M / 2 : 1];                             // conj(W_2N^k) = e^{+j pi k / N}
  wc[0] = cconj(tw.wsplit);
#pragma unroll
  for (int m = 1; m < M / 2; ++m) wc[m] = cmul(wc[0], cconj(cis64(m * (2048 / N))));
  wave_sync();
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int k = lane + 64 * m;
    cpx a = spec[k];
    cpx bz = spec[N - k];                                    // b = conj(bz)
    if (m == 0 && lane == 0) { a.y = 0.0; bz.y = 0.0; }      // Im(DC), Im(Nyquist) ignored
    const cpx w = m < M / 2 ? wc[m] : make_double2(-wc[m - M / 2].y, wc[m - M / 2].x);
    const double sx = a.x + bz.x, sy = a.y - bz.y, dx = a.x - bz.x, dy = a.y + bz.y;
    v[m] = make_double2(__builtin_fma(-dx, w.y, __builtin_fma(-dy, w.x, sx)),
                        __builtin_fma(dx, w.x, __builtin_fma(-dy, w.y, sy)));
  }
  fft_backward<N>(v, lds, tw, lane);
}

// The inverse real transform from a half spectrum held by PAIRS as rfft_split_pairs leaves it (xk[m] = X[k],
// xr[m] = X[N - k], k = lane + 64 m < N / 2; lane 0, m = 0: X[0] and X[N], imaginary parts ignored; xh = X[N / 2] in
// lane 0): Z[k] and Z[N - k] are again the same sums and differences under twiddles that differ in sign, so one lane
// computes both and only Z[N - k] travels (through LDS) to the lane and register it belongs to -- half a store and half
// a read per element where rfft_backward reads a stored spectrum twice.
// The un-split of one pair: Z[k] (returned) and Z[N - k] from X[k] = a and X[N - k] = bz under w = conj(W_2N^k).
__device__ __forceinline__ cpx rfft_unsplit_pair(cpx a, cpx bz, cpx w, cpx& zr) {
  const double sx = a.x + bz.x, sy = a.y - bz.y, dx = a.x - bz.x, dy = a.y + bz.y;
  const double pr = __builtin_fma(dy, w.x, dx * w.y), pi = __builtin_fma(dx, w.x, -(dy * w.y));
  zr = make_double2(sx + pr, pi - sy);
  return make_double2(sx - pr, sy + pi);
}
// g(m, xk, xr) fills the pair m < M / 2; g(M / 2, xh, .) the middle bin.  Leaves v = Z, the operand of fft_backward.
template <int N, class G>
__device__ __forceinline__ void rfft_unsplit_pairs_f(G g, cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw, int lane) {
  constexpr int M = N / 64;
  asm volatile("" : "+v"(lane));
  const_cast<FftTw<N>&>(tw).fence();
  const cpx wc = cconj(tw.wsplit);                           // conj(W_2N^lane) = e^{+j pi lane / N}
  wave_sync();
#pragma unroll
  for (int m = 0; m < M / 2; ++m) {
    const int k = lane + 64 * m;
    cpx a, bz;
    g(m, a, bz);
    if (m == 0) {                                            // Im(DC), Im(Nyquist) ignored
      a.y = lane == 0 ? 0.0 : a.y;
      bz.y = lane == 0 ? 0.0 : bz.y;
    }
    const cpx w = m == 0 ? wc : cmul(wc, cconj(cis64(m * (2048 / N))));
    cpx zr;
    v[m] = rfft_unsplit_pair(a, bz, w, zr);
    if (m > 0 || lane > 0) lds[N - k] = zr;
  }
  {
    cpx xh, unused;
    g(M / 2, xh, unused);
    if (lane == 0) lds[N / 2] = make_double2(2.0 * xh.x, -2.0 * xh.y);
  }
  wave_sync();
#pragma unroll
  for (int m = M / 2; m < M; ++m) v[m] = lds[lane + 64 * m];
}
template <int N>
__device__ __forceinline__ void rfft_backward_pairs(const cpx (&xk)[N / 128], const cpx (&xr)[N / 128], cpx xh,
                                                    cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw, int lane) {
  rfft_unsplit_pairs_f<N>([&](int m, cpx& a, cpx& b) {
    if (m < N / 128) {
      a = xk[m < N / 128 ? m : 0];
      b = xr[m < N / 128 ? m : 0];
    } else {
      a = xh;
    }
  }, v, lds, tw, lane);
  fft_backward<N>(v, lds, tw, lane);
}
template <int N, class G>
__device__ __forceinline__ void rfft_backward_pairs_f(G g, cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw, int lane) {
  rfft_unsplit_pairs_f<N>(g, v, lds, tw, lane);
  fft_backward<N>(v, lds, tw, lane);
}

// x -> irfft(mul . rfft(x)) in one go: forward transform (only v[0 .. nz) may be non-zero), then every pair is split,
// handed to mul(m, X[k], X[N - k]) (m = M / 2: the middle bin, lane 0's counts) to be changed in place, and un-split
// again at once: the spectrum never exists outside the pair in flight.  A lane reads its partner Z[N - k] from LDS and
// later writes the new Z[N - k] to the same address, which no other lane reads: no barrier in between.
template <int N, class Mul>
__device__ __forceinline__ void rfft_filter_pairs(cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw, int lane, int nz,
                                                  Mul mul) {
  constexpr int M = N / 64;
  fft_forward_nz<N>(v, lds, tw, lane, nz);
  asm volatile("" : "+v"(lane));
  wave_sync();
#pragma unroll
  for (int m = M / 2; m < M; ++m) lds[lane + 64 * m] = v[m];
  wave_sync();
  const cpx wh = make_double2(0.5 * tw.wsplit.x, 0.5 * tw.wsplit.y);    // W_2N^k / 2
#pragma unroll
  for (int m = 0; m < M / 2; ++m) {
    const int k = lane + 64 * m;
    const cpx a = v[m];
    cpx bz = lds[(N - k) & (N - 1)];
    if (m == 0) {
      bz.x = lane == 0 ? a.x : bz.x;
      bz.y = lane == 0 ? a.y : bz.y;
    }
    const cpx w = m == 0 ? wh : cmul(wh, cis64(m * (2048 / N)));
    const double sx = a.x + bz.x, sy = a.y - bz.y, dx = a.x - bz.x, dy = a.y + bz.y;
    const double pr = __builtin_fma(w.x, dy, w.y * dx), pi = __builtin_fma(w.y, dy, -(w.x * dx));
    cpx xk = make_double2(__builtin_fma(0.5, sx, pr), __builtin_fma(0.5, sy, pi));
    cpx xr = make_double2(__builtin_fma(0.5, sx, -pr), __builtin_fma(-0.5, sy, pi));
    mul(m, xk, xr);
    if (m == 0) {                                            // Im(DC), Im(Nyquist) ignored
      xk.y = lane == 0 ? 0.0 : xk.y;
      xr.y = lane == 0 ? 0.0 : xr.y;
    }
    cpx zr;
    v[m] = rfft_unsplit_pair(xk, xr, make_double2(2.0 * w.x, -2.0 * w.y), zr);      // conj(W_2N^k)
    if (m > 0 || lane > 0) lds[N - k] = zr;
  }
  {
    cpx xh = make_double2(v[M / 2].x, -v[M / 2].y), unused = xh;
    mul(M / 2, xh, unused);
    if (lane == 0) lds[N / 2] = make_double2(2.0 * xh.x, -2.0 * xh.y);
  }
  wave_sync();
#pragma unroll
  for (int m = M / 2; m < M; ++m) v[m] = lds[lane + 64 * m];
  fft_backward<N>(v, lds, tw, lane);
}

template <int N, int NZM = N / 64, class F>
__device__ __forceinline__ void rfft_forward_pairs_f(cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw, int lane, F f) {
  if constexpr (N < 512) fft_forward<N>(v, lds, tw, lane);          // 256 points ride on the 512-point plan, unpruned
  else fft_forward<N, NZM>(v, lds, tw, lane);
  rfft_split_pairs_f<N>(v, lds, tw, lane, f);
}
template <int N, int NZM = N / 64>
__device__ __forceinline__ void rfft_forward_pairs(cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw, int lane,
                                                   cpx (&xk)[N / 128], cpx (&xr)[N / 128], cpx& xh) {
  if constexpr (N < 512) fft_forward<N>(v, lds, tw, lane);          // 256 points ride on the 512-point plan, unpruned
  else fft_forward<N, NZM>(v, lds, tw, lane);
  rfft_split_pairs<N>(v, lds, tw, lane, xk, xr, xh);
}

// ---- N = 256 (fft_size 512: CheapTrick / Synthesis / codec at fs <= 12.8 kHz, cheaptrick.cpp:191-194) ----------
// Four elements per lane leave no room for a three-pass radix plan, and the case is rare (8 kHz speech), so the
// 256-point transform rides on the 512-point one: interleaving the input with zeros, z'[2 n] = z[n], z'[2 n + 1] = 0,
// gives Z'[k] = Z[k mod 256].  Twice the arithmetic of a dedicated plan; same conventions, same call sites (the
// real-transform wrappers above are generic in N once fft_forward<256> and FftTw<256> exist).
template <> struct FftLds<256> { static constexpr int kElems = FftLds<512>::kElems; };

template <> struct FftTw<256> {
  FftTw<512> t;
  cpx wsplit;        // W_512^lane
  __device__ __forceinline__ void init(int lane) {
    t.init(lane);
    wsplit = cis_neg2pi((double)lane / 512.0);
  }
  __device__ __forceinline__ cpx wstep() const { return cis64(8); }      // W_512^64
  __device__ __forceinline__ void fence() {
    t.fence();
    asm volatile("" : "+v"(wsplit.x), "+v"(wsplit.y));
  }
};

template <>
__device__ __forceinline__ void fft_forward<256>(cpx (&v)[4], cpx* lds, const FftTw<256>& tw, int lane) {
  wave_sync();
#pragma unroll
  for (int m = 0; m < 4; ++m) lds[lane + 64 * m] = v[m];
  wave_sync();
  cpx c[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int i = lane + 64 * m;                       // element of the zero-interleaved sequence
    const cpx a = lds[i >> 1];
    c[m] = (i & 1) ? make_double2(0.0, 0.0) : a;
  }
  fft_forward<512>(c, lds, tw.t, lane);
#pragma unroll
  for (int m = 0; m < 4; ++m) v[m] = c[m];             // Z'[k] = Z[k] for k < 256
}

}  // namespace wm

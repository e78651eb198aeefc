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
template <int R> struct Dft;

template <> struct Dft<2> {
  __device__ static __forceinline__ void run(cpx (&v)[2]) {
    cpx a = v[0];
    v[0] = cadd(a, v[1]);
    v[1] = csub(a, v[1]);
  }
};

template <> struct Dft<4> {
  __device__ static __forceinline__ void run(cpx (&v)[4]) {
    cpx s0 = cadd(v[0], v[2]), d0 = csub(v[0], v[2]);
    cpx s1 = cadd(v[1], v[3]), d1 = cmul_mi(csub(v[1], v[3]));
    v[0] = cadd(s0, s1);
    v[2] = csub(s0, s1);
    v[1] = cadd(d0, d1);
    v[3] = csub(d0, d1);
  }
};

template <> struct Dft<8> {
  __device__ static __forceinline__ void run(cpx (&v)[8]) {
    const double h = 0.70710678118654752440;
    cpx e[4] = {v[0], v[2], v[4], v[6]};
    cpx o[4] = {v[1], v[3], v[5], v[7]};
    Dft<4>::run(e);
    Dft<4>::run(o);
    // W8^1 = h(1-i), W8^2 = -i, W8^3 = h(-1-i)
    cpx t1 = make_double2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x));
    cpx t2 = cmul_mi(o[2]);
    cpx t3 = make_double2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y));
    v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
    v[1] = cadd(e[1], t1);   v[5] = csub(e[1], t1);
    v[2] = cadd(e[2], t2);   v[6] = csub(e[2], t2);
    v[3] = cadd(e[3], t3);   v[7] = csub(e[3], t3);
  }
};

template <> struct Dft<16> {
  __device__ static __forceinline__ void run(cpx (&v)[16]) {
    const double h = 0.70710678118654752440;
    const double c1 = 0.92387953251128675613;   // cos(pi/8)
    const double s1 = 0.38268343236508977173;   // sin(pi/8)
    cpx e[8] = {v[0], v[2], v[4], v[6], v[8], v[10], v[12], v[14]};
    cpx o[8] = {v[1], v[3], v[5], v[7], v[9], v[11], v[13], v[15]};
    Dft<8>::run(e);
    Dft<8>::run(o);
    // W16^k = (cos(k pi/8), -sin(k pi/8))
    cpx t[8];
    t[0] = o[0];
    t[1] = cmul(o[1], make_double2(c1, -s1));
    t[2] = make_double2(h * (o[2].x + o[2].y), h * (o[2].y - o[2].x));
    t[3] = cmul(o[3], make_double2(s1, -c1));
    t[4] = cmul_mi(o[4]);
    t[5] = cmul(o[5], make_double2(-s1, -c1));
    t[6] = make_double2(h * (o[6].y - o[6].x), -h * (o[6].x + o[6].y));
    t[7] = cmul(o[7], make_double2(-c1, -s1));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = cadd(e[k], t[k]);
      v[k + 8] = csub(e[k], t[k]);
    }
  }
};

// multiply a[r] by w^r, r = 1..R-1.
// WM_FFT_ILP (default): powers from a shallow product tree (depth log2 R) -- more live registers,
// short dependency chains; the one-wave-per-SIMD kernels are latency bound and want this.
// Otherwise two interleaved chains stepping by w^2 (three live complex values).
#ifndef WM_FFT_ILP
#define WM_FFT_ILP 1
#endif
template <int R> __device__ __forceinline__ void apply_twiddle_powers(cpx (&a)[R], cpx w) {
#if WM_FFT_ILP
  cpx p[R];
  p[1] = w;
#pragma unroll
  for (int r = 2; r < R; ++r) p[r] = (r & 1) ? cmul(p[r - 1], w) : cmul(p[r / 2], p[r / 2]);
#pragma unroll
  for (int r = 1; r < R; ++r) a[r] = cmul(a[r], p[r]);
#else
  const cpx w2 = cmul(w, w);
  cpx odd = w, even = w2;
  a[1] = cmul(a[1], odd);
#pragma unroll
  for (int r = 2; r < R; r += 2) {
    a[r] = cmul(a[r], even);
    if (r + 1 < R) {
      odd = cmul(odd, w2);
      a[r + 1] = cmul(a[r + 1], odd);
    }
    if (r + 2 < R) even = cmul(even, w2);
  }
#endif
}

template <int N> struct FftCfg;
template <> struct FftCfg<512>  { static constexpr int R1 = 8,  R2 = 8,  R3 = 8; };
template <> struct FftCfg<1024> { static constexpr int R1 = 16, R2 = 8,  R3 = 8; };
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

// Forward complex FFT (e^{-j}).  v[m] holds element lane + 64 m on entry and on
// exit.  `lds` must provide FftLds<N>::kElems cpx.  Caller guarantees no other
// use of `lds` is in flight (ends with registers only; starts with a barrier).
template <int N>
__device__ __forceinline__ void fft_forward(cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw, int lane) {
  constexpr int M = N / 64;
  // per-call fences: every transform derives its own LDS addresses and twiddle powers (a few
  // integer ops / FMAs) instead of sharing hoisted copies that the register allocator then spills
  asm volatile("" : "+v"(lane));
  const_cast<FftTw<N>&>(tw).fence();
  constexpr int R1 = FftCfg<N>::R1, R2 = FftCfg<N>::R2, R3 = FftCfg<N>::R3;
  constexpr int S1 = M / R1, S2 = M / R2, S3 = M / R3;
  static_assert(S1 >= 1 && S2 >= 1 && S3 >= 1, "radix plan does not fit 64 lanes");
  static_assert(R1 * R2 * R3 == N, "radix plan");

  wave_sync();
  // ---- pass 1: radix R1, Ns = 1, no twiddles; store to out[j*R1 + r]
#pragma unroll
  for (int b = 0; b < S1; ++b) {
    cpx a[R1];
#pragma unroll
    for (int r = 0; r < R1; ++r) a[r] = v[b + r * S1];
    Dft<R1>::run(a);
    const unsigned j = (unsigned)lane + 64u * b;
#pragma unroll
    for (int r = 0; r < R1; ++r) lds[fft_pad<N>(j * R1 + r)] = a[r];
#if !WM_FFT_ILP
    __builtin_amdgcn_sched_barrier(0);      // one butterfly at a time: bounds the live registers
#endif
  }
  wave_sync();
#pragma unroll
  for (int m = 0; m < M; ++m) v[m] = lds[fft_pad<N>((unsigned)lane + 64u * m)];
  wave_sync();
  // ---- pass 2: radix R2, Ns = R1; twiddle W_{R1 R2}^{k r}, k = j % R1 = lane % R1
#pragma unroll
  for (int b = 0; b < S2; ++b) {
    cpx a[R2];
#pragma unroll
    for (int r = 0; r < R2; ++r) a[r] = v[b + r * S2];
    apply_twiddle_powers<R2>(a, tw.w2);
    Dft<R2>::run(a);
    const unsigned j = (unsigned)lane + 64u * b;
    const unsigned base = (j / R1) * (R1 * R2) + (j % R1);
#pragma unroll
    for (int r = 0; r < R2; ++r) lds[fft_pad<N>(base + (unsigned)r * R1)] = a[r];
#if !WM_FFT_ILP
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
  wave_sync();
#pragma unroll
  for (int m = 0; m < M; ++m) v[m] = lds[fft_pad<N>((unsigned)lane + 64u * m)];
  // ---- pass 3: radix R3, Ns = N / R3; twiddle W_N^{j r}; output lands in the register layout
#pragma unroll
  for (int b = 0; b < S3; ++b) {
    cpx a[R3];
#pragma unroll
    for (int r = 0; r < R3; ++r) a[r] = v[b + r * S3];
    apply_twiddle_powers<R3>(a, tw.w3(b));
    Dft<R3>::run(a);
#pragma unroll
    for (int r = 0; r < R3; ++r) v[b + r * S3] = a[r];
#if !WM_FFT_ILP
    __builtin_amdgcn_sched_barrier(0);
#endif
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
template <int N>
__device__ __forceinline__ void rfft_forward(cpx (&v)[N / 64], cpx* lds, cpx* spec, const FftTw<N>& tw,
                                             int lane) {
  constexpr int M = N / 64;
  fft_forward<N>(v, lds, tw, lane);
  asm volatile("" : "+v"(lane));
  wave_sync();
#pragma unroll
  for (int m = 0; m < M; ++m) lds[lane + 64 * m] = v[m];     // Z, plain layout
  wave_sync();
  cpx xk[M];
  cpx w = tw.wsplit;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int k = lane + 64 * m;
    cpx a = v[m];
    cpx b = cconj(lds[(N - k) & (N - 1)]);
    cpx e = make_double2(0.5 * (a.x + b.x), 0.5 * (a.y + b.y));
    cpx d = csub(a, b);
    cpx o = make_double2(0.5 * d.y, -0.5 * d.x);             // (a-b)/(2i)
    xk[m] = cadd(e, cmul(w, o));
    w = cmul(w, tw.wstep());
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

// c2r (unnormalised; fft.cpp:27-35 semantics): X[0..N] in `spec` (LDS, plain).
// On exit v[m] = (x[2n], x[2n+1]), n = lane + 64 m.  spec may alias `lds`.
template <int N>
__device__ __forceinline__ void rfft_backward(const cpx* spec, cpx (&v)[N / 64], cpx* lds, const FftTw<N>& tw,
                                              int lane) {
  constexpr int M = N / 64;
  asm volatile("" : "+v"(lane));
  const_cast<FftTw<N>&>(tw).fence();
  cpx w = cconj(tw.wsplit);                                  // e^{+j pi k / N}
  const cpx wst = cconj(tw.wstep());
  wave_sync();
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int k = lane + 64 * m;
    cpx a = spec[k];
    cpx b = cconj(spec[N - k]);
    if (k == 0) { a.y = 0.0; b.y = 0.0; }                    // Im(DC), Im(Nyquist) ignored
    cpx s = cadd(a, b);                                      // 2E
    cpx o = cmul(csub(a, b), w);                             // 2O
    v[m] = make_double2(s.x - o.y, s.y + o.x);               // 2E + j 2O
    w = cmul(w, wst);
  }
  fft_backward<N>(v, lds, tw, lane);
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

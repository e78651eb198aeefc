// bfft.hpp -- workgroup-cooperative FP64 FFTs in LDS for gfx950 (CDNA4).
//
// Second FFT engine of this library (the first, fft.hpp, gives one transform to one wavefront and
// keeps N/64 points per lane in registers; at N >= 1024 that costs > 256 VGPRs and one wave per
// SIMD).  Here NT threads share one transform of N complex points held in LDS: every thread owns
// one radix-4 butterfly per pass (Stockham autosort, log4 N passes + one radix-2 when log2 N is
// odd), i.e. four complex values in registers at a time, so the kernels built on it need < 128
// VGPRs and run 4 workgroups per CU.
//
// Conventions as fft.hpp / the reference wrapper (externs/WORLD_v2/src/fft.cpp:26-72):
// forward = e^{-j}, r2c gives bins 0..N of a length-2N real sequence, c2r is the unnormalised
// inverse that ignores Im(DC) and Im(Nyquist).
//
// Each pass reads its four inputs, synchronises, then writes its four outputs (in place in one
// LDS image of N+1 complex values); twiddle bases are per-thread registers computed once per kernel.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "fft.hpp"

namespace wm {

constexpr int bfft_ilog2(int n) { return n <= 1 ? 0 : 1 + bfft_ilog2(n / 2); }

template <int N, int NT> struct BFft {
  static constexpr int LOG = bfft_ilog2(N);
  static constexpr int L4 = LOG / 2;                 // radix-4 passes
  static constexpr bool HAS2 = (LOG % 2) == 1;       // trailing radix-2 pass
  static constexpr int NB4 = (N / 4 + NT - 1) / NT;  // radix-4 butterflies per thread
  static constexpr int NB2 = (N / 2 + NT - 1) / NT;  // radix-2 butterflies per thread
  static constexpr int NPAIR = (N / 2 + NT - 1) / NT;   // (k, N-k) pairs per thread, k = 1 .. N/2
  static constexpr int kLdsElems = N + 1;

  cpx tw4[L4 > 1 ? L4 - 1 : 1][NB4];   // pass p >= 1: W_{4 Ns}^{j % Ns}, Ns = 4^p
  cpx tw2[NB2];                        // radix-2 pass: W_N^j
  cpx ws[NPAIR];                       // real split: W_{2N}^k for this thread's pairs

  __device__ __forceinline__ void init(int tid) {
#pragma unroll
    for (int p = 1; p < L4; ++p) {
      const int ns = 1 << (2 * p);
#pragma unroll
      for (int b = 0; b < NB4; ++b) {
        const int j = tid + NT * b;
        tw4[p - 1][b] = cis_neg2pi((double)(j % ns) / (double)(4 * ns));
      }
    }
    if (HAS2) {
#pragma unroll
      for (int b = 0; b < NB2; ++b) tw2[b] = cis_neg2pi((double)(tid + NT * b) / (double)N);
    }
#pragma unroll
    for (int q = 0; q < NPAIR; ++q) ws[q] = cis_neg2pi((double)(1 + tid + NT * q) / (double)(2 * N));
  }

  // in-place forward complex FFT of buf[0..N) (natural order in and out).  All NT threads call.
  __device__ __forceinline__ void forward(cpx* buf, int tid) const {
#pragma unroll
    for (int p = 0; p < L4; ++p) {
      const int ns = 1 << (2 * p);
      cpx a[NB4][4];
      __syncthreads();
#pragma unroll
      for (int b = 0; b < NB4; ++b) {
        const int j = tid + NT * b;
        if (j < N / 4) {
#pragma unroll
          for (int r = 0; r < 4; ++r) a[b][r] = buf[j + r * (N / 4)];
        }
      }
      __syncthreads();
#pragma unroll
      for (int b = 0; b < NB4; ++b) {
        const int j = tid + NT * b;
        if (j < N / 4) {
          if (p > 0) {
            const cpx w = tw4[p > 0 ? p - 1 : 0][b];
            const cpx w2 = cmul(w, w);
            a[b][1] = cmul(a[b][1], w);
            a[b][2] = cmul(a[b][2], w2);
            a[b][3] = cmul(a[b][3], cmul(w2, w));
          }
          Dft<4>::run(a[b]);
          const int base = (j / ns) * (4 * ns) + (j % ns);
#pragma unroll
          for (int r = 0; r < 4; ++r) buf[base + r * ns] = a[b][r];
        }
      }
    }
    if (HAS2) {                     // Ns = N/2: outputs land on the inputs' own slots
      __syncthreads();
#pragma unroll
      for (int b = 0; b < NB2; ++b) {
        const int j = tid + NT * b;
        if (j < N / 2) {
          const cpx x0 = buf[j];
          const cpx x1 = cmul(buf[j + N / 2], tw2[b]);
          buf[j] = cadd(x0, x1);
          buf[j + N / 2] = csub(x0, x1);
        }
      }
    }
    __syncthreads();
  }

  // Real-FFT split in place: buf holds Z = FFT_N(x[2n] + j x[2n+1]); on exit buf[0..N] is the
  // half spectrum X of the length-2N real sequence.  Each (k, N-k) pair is owned by one thread.
  __device__ __forceinline__ void split(cpx* buf, int tid) const {
#pragma unroll
    for (int q = 0; q < NPAIR; ++q) {
      const int k = 1 + tid + NT * q;
      if (k <= N / 2) {
        const cpx a = buf[k];
        const cpx b = cconj(buf[N - k]);
        const cpx e = make_double2(0.5 * (a.x + b.x), 0.5 * (a.y + b.y));
        const cpx d = csub(a, b);
        const cpx t = cmul(ws[q], make_double2(0.5 * d.y, -0.5 * d.x));   // W^k * (a-b)/(2i)
        buf[k] = cadd(e, t);
        buf[N - k] = cconj(csub(e, t));
      }
    }
    if (tid == 0) {
      const cpx z0 = buf[0];
      buf[0] = make_double2(z0.x + z0.y, 0.0);
      buf[N] = make_double2(z0.x - z0.y, 0.0);
    }
    __syncthreads();
  }

  // r2c of the real sequence stored as packed pairs in buf (buf[n] = (x[2n], x[2n+1])).
  __device__ __forceinline__ void rfft_forward(cpx* buf, int tid) const {
    forward(buf, tid);
    split(buf, tid);
  }

  // c2r: buf[0..N] half spectrum -> buf[n] = (x[2n], x[2n+1]) of the unnormalised inverse.
  __device__ __forceinline__ void rfft_backward(cpx* buf, int tid) const {
    __syncthreads();
    if (tid == 0) {
      const double x0 = buf[0].x, xn = buf[N].x;               // Im(DC), Im(Nyquist) ignored
      buf[0] = make_double2(x0 + xn, -(x0 - xn));              // conj(Z[0]), Z[0] = (X0+XN) + j (X0-XN)
    }
#pragma unroll
    for (int q = 0; q < NPAIR; ++q) {
      const int k = 1 + tid + NT * q;
      if (k <= N / 2) {
        // Z[k] = (A + conj(B)) + j W^{-k} (A - conj(B)), A = X[k], B = X[N-k]; Z[N-k] by symmetry.
        const cpx a = buf[k];
        const cpx b = cconj(buf[N - k]);
        const cpx s = cadd(a, b);
        const cpx o = cmul(csub(a, b), cconj(ws[q]));
        const cpx zk = make_double2(s.x - o.y, s.y + o.x);
        const cpx zn = make_double2(s.x + o.y, -(s.y - o.x));  // Z[N-k] = conj(s) + j W^{k}... see DESIGN
        // store conjugates: inverse = conj(forward(conj(Z)))
        buf[k] = cconj(zk);
        if (k != N - k) buf[N - k] = cconj(zn);
      }
    }
    forward(buf, tid);
#pragma unroll
    for (int q = 0; q < (N + NT - 1) / NT; ++q) {
      const int n = tid + NT * q;
      if (n < N) buf[n].y = -buf[n].y;
    }
    __syncthreads();
  }
};

// ---- workgroup collectives for NT = 64 * NW threads ----------------------------------------------
template <int NW> struct BlockOps {
  // sum over the workgroup, returned to every thread.  `red` is an LDS array of >= NW doubles.
  __device__ static __forceinline__ double sum(double v, double* red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w];
    return s;
  }
  // two sums at once (one barrier pair); red >= 2 NW doubles
  __device__ static __forceinline__ void sum2(double& a, double& b, double* red, int tid) {
    a = wave_sum(a);
    b = wave_sum(b);
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = a; red[NW + (tid >> 6)] = b; }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { sa += red[w]; sb += red[NW + w]; }
    a = sa; b = sb;
  }
};

}  // namespace wm

// decimate.hpp -- zero-phase 3rd-order IIR decimation shared by Harvest and DIO (speed > 1).
//
// Replaces decimate / FilterForDecimate (externs/WORLD_v2/src/matlabfunctions.cpp:27-125, 184-210):
// reflect-pad by 9 samples, run the IIR forward, reverse, run it again, reverse, keep every r-th
// sample.  The recursion is sequential in the reference; here each thread runs it over its own
// 256-sample chunk after a warm-up from zero state whose length follows the filter's largest pole radius
// (0.66 at r = 2, 0.80 at r = 6, 0.89 at r = 12: 160 ... 512 samples), so that the warm-up state differs from
// the sequential one by < 1e-26 relative and the outputs round to the same doubles.  FMA contraction is off
// to keep the reference's operation order.
#pragma once
#include "common.hpp"

namespace wm {

struct DecMeta {
  int r;                       // decimation ratio 2..12
  int lag;                     // edge padding applied before decimating (Harvest, harvest.cpp:50-59); 0 for DIO
  int warm;                    // warm-up samples in front of a chunk: radius^warm < 1e-26
  double a0, a1, a2, b0, b1;   // matlabfunctions.cpp:29-113
};

static const double kDecimateA[13][3] = {
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
static const double kDecimateB[13][2] = {
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

inline DecMeta make_dec_meta(int r, int lag) {
  DecMeta d;
  d.r = r; d.lag = lag;
  // largest pole radius of z^3 - a0 z^2 - a1 z - a2 by ratio: 0.657 0.686 0.731 0.769 0.799 0.822 0.841 0.857 0.869 0.880 0.889
  static const int kWarm[13] = {0, 0, 160, 176, 208, 256, 288, 336, 384, 432, 464, 512, 512};
  d.warm = kWarm[r];
  d.a0 = kDecimateA[r][0]; d.a1 = kDecimateA[r][1]; d.a2 = kDecimateA[r][2];
  d.b0 = kDecimateB[r][0]; d.b1 = kDecimateB[r][1];
  return d;
}

__device__ __forceinline__ double hv_nx(const double* __restrict__ x, int n, int lag, int j) {
  // new_x of GetWaveformAndSpectrumSub (harvest.cpp:55-59): x edge-padded by lag samples
  return x[imin(n - 1, imax(0, j - lag))];
}
constexpr int kDecChunk = 256;
constexpr int kDecBlk = 16;     // samples fetched per trip to memory: the recursion is the chain, the loads must not be on it

// pass 1: tmp2[i] = IIR(tmp1)[i]
static __global__ __launch_bounds__(64) void decim_fwd_kernel(const double* __restrict__ x,
                                                          const int64_t* __restrict__ x_off,
                                                          const int* __restrict__ x_len, DecMeta m,
                                                          const int64_t* __restrict__ toff,
                                                          double* __restrict__ tmp) {
#pragma clang fp contract(off)
  const int u = blockIdx.y;
  const int n = x_len[u], nn = n + 2 * m.lag, len = nn + 18;
  const int c0 = (blockIdx.x * 64 + threadIdx.x) * kDecChunk;
  if (c0 >= len) return;
  const double* xu = x + x_off[u];
  double* out = tmp + toff[u];
  double w0 = 0.0, w1 = 0.0, w2 = 0.0;
  const int start = imax(0, c0 - m.warm);
  const int end = imin(len, c0 + kDecChunk);
  // tmp1 of decimate (matlabfunctions.cpp:189-192): new_x reflect-padded by 9 samples, 2 edge - mirrored inside the pads
  const double e_lo = hv_nx(xu, n, m.lag, 0), e_hi = hv_nx(xu, n, m.lag, nn - 1);
  for (int i0 = start; i0 < end; i0 += kDecBlk) {
    double v[kDecBlk];
#pragma unroll
    for (int r = 0; r < kDecBlk; ++r) {                          // sixteen independent loads, then the chain
      const int i = imin(i0 + r, len - 1);
      const int j = i < 9 ? 9 - i : (i >= 9 + nn ? nn - 2 - (i - (9 + nn)) : i - 9);
      v[r] = hv_nx(xu, n, m.lag, j);
    }
#pragma unroll
    for (int r = 0; r < kDecBlk; ++r) {
      const int i = i0 + r;
      if (i < end) {
        const double in = i < 9 ? 2 * e_lo - v[r] : (i >= 9 + nn ? 2 * e_hi - v[r] : v[r]);
        const double wt = in + m.a0 * w0 + m.a1 * w1 + m.a2 * w2;
        const double o = m.b0 * wt + m.b1 * w0 + m.b1 * w1 + m.b0 * w2;
        w2 = w1; w1 = w0; w0 = wt;
        if (i >= c0) out[i] = o;
      }
    }
  }
}

// pass 2 on the reversed pass-1 output; only the decimated samples are kept:
// y[c] = tmp1_final[nbeg + c r + 8] (matlabfunctions.cpp:201-206), then y[lag/r + i] (harvest.cpp:62)
static __global__ __launch_bounds__(64) void decim_bwd_kernel(const int* __restrict__ x_len, DecMeta m,
                                                          const int64_t* __restrict__ toff,
                                                          const double* __restrict__ tmp,
                                                          const int64_t* __restrict__ yoff,
                                                          const int* __restrict__ ylen_a, double* __restrict__ y) {
#pragma clang fp contract(off)
  const int u = blockIdx.y;
  const int n = x_len[u], nn = n + 2 * m.lag, len = nn + 18;
  const int c0 = (blockIdx.x * 64 + threadIdx.x) * kDecChunk;
  if (c0 >= len) return;
  const double* in = tmp + toff[u];
  double* yu = y + yoff[u];
  const int ylen = ylen_a[u];
  const int nout = (nn - 1) / m.r + 1;
  const int nbeg = m.r - m.r * nout + nn;
  const int shift = m.lag / m.r;
  double w0 = 0.0, w1 = 0.0, w2 = 0.0;
  const int start = imax(0, c0 - m.warm);
  const int end = imin(len, c0 + kDecChunk);
  // sample i lands at j = len - 1 - i of the re-reversed array, which is output c = (j - 8 - nbeg) / r when that
  // divides: q = j - 8 - nbeg falls by one per step, so its remainder and quotient are counted down, not divided out
  int q = len - 1 - start - 8 - nbeg;
  int ph = q % m.r;
  if (ph < 0) ph += m.r;
  int cq = (q - ph) / m.r;                                       // floor(q / r)
  for (int i0 = start; i0 < end; i0 += kDecBlk) {
    double v[kDecBlk];
#pragma unroll
    for (int r = 0; r < kDecBlk; ++r) v[r] = in[len - 1 - imin(i0 + r, len - 1)];
#pragma unroll
    for (int r = 0; r < kDecBlk; ++r) {
      const int i = i0 + r;
      if (i < end) {
        const double wt = v[r] + m.a0 * w0 + m.a1 * w1 + m.a2 * w2;
        const double o = m.b0 * wt + m.b1 * w0 + m.b1 * w1 + m.b0 * w2;
        w2 = w1; w1 = w0; w0 = wt;
        if (i >= c0 && ph == 0 && q >= 0 && q + nbeg < nn + 9) {
          const int c = cq - shift;
          if (c >= 0 && c < ylen) yu[c] = o;
        }
        --q;
        if (ph == 0) { ph = m.r - 1; --cq; } else { --ph; }
      }
    }
  }
}


}  // namespace wm

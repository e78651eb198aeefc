// micro-benchmark: cost of a dependent v_add_f64 chain in one wavefront (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#define STEP_A "v_add_f64 %[t], %[t], %[b]\n\t"
#define STEP_B "v_add_f64 %[t], %[t], %[b]\n\ts_lshl_b64 exec, exec, 1\n\t"
#define STEP_C "v_add_f64 %[t], %[t], %[b]\n\ts_nop 0\n\t"
#define R16(S) S S S S S S S S S S S S S S S S
template <int V> __global__ void k(double* out, double inc, int iters, long long* cyc) {
  __shared__ double lds[256];
  for (int q = threadIdx.x; q < 256; q += 64) lds[q] = 1e-3 * q;
  __syncthreads();
  double t = threadIdx.x;
  long long c0 = __builtin_readcyclecounter();
  long long m0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    if (V == 0) asm volatile(R16(STEP_A) : [t] "+v"(t) : [b] "s"(inc));
    if (V == 1) asm volatile("s_mov_b64 exec, -1\n\t" R16(STEP_B) "s_mov_b64 exec, -1" : [t] "+v"(t) : [b] "s"(inc) : "scc");
    if (V == 6) {   // 16 broadcast LDS reads, then the chain with exec shifts
      double r[16];
      const unsigned a = (i & 7) * 128;
#pragma unroll
      for (int q = 0; q < 16; ++q) r[q] = lds[(a >> 3) + q];
      asm volatile("s_mov_b64 exec, -1\n\t"
        "v_add_f64 %[t], %[t], %[r0]\n\ts_lshl_b64 exec, exec, 1\n\t" "v_add_f64 %[t], %[t], %[r1]\n\ts_lshl_b64 exec, exec, 1\n\t"
        "v_add_f64 %[t], %[t], %[r2]\n\ts_lshl_b64 exec, exec, 1\n\t" "v_add_f64 %[t], %[t], %[r3]\n\ts_lshl_b64 exec, exec, 1\n\t"
        "v_add_f64 %[t], %[t], %[r4]\n\ts_lshl_b64 exec, exec, 1\n\t" "v_add_f64 %[t], %[t], %[r5]\n\ts_lshl_b64 exec, exec, 1\n\t"
        "v_add_f64 %[t], %[t], %[r6]\n\ts_lshl_b64 exec, exec, 1\n\t" "v_add_f64 %[t], %[t], %[r7]\n\ts_lshl_b64 exec, exec, 1\n\t"
        "v_add_f64 %[t], %[t], %[r8]\n\ts_lshl_b64 exec, exec, 1\n\t" "v_add_f64 %[t], %[t], %[r9]\n\ts_lshl_b64 exec, exec, 1\n\t"
        "v_add_f64 %[t], %[t], %[r10]\n\ts_lshl_b64 exec, exec, 1\n\t" "v_add_f64 %[t], %[t], %[r11]\n\ts_lshl_b64 exec, exec, 1\n\t"
        "v_add_f64 %[t], %[t], %[r12]\n\ts_lshl_b64 exec, exec, 1\n\t" "v_add_f64 %[t], %[t], %[r13]\n\ts_lshl_b64 exec, exec, 1\n\t"
        "v_add_f64 %[t], %[t], %[r14]\n\ts_lshl_b64 exec, exec, 1\n\t" "v_add_f64 %[t], %[t], %[r15]\n\t"
        "s_mov_b64 exec, -1" : [t] "+v"(t) : [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [r3] "v"(r[3]), [r4] "v"(r[4]), [r5] "v"(r[5]),
        [r6] "v"(r[6]), [r7] "v"(r[7]), [r8] "v"(r[8]), [r9] "v"(r[9]), [r10] "v"(r[10]), [r11] "v"(r[11]), [r12] "v"(r[12]), [r13] "v"(r[13]),
        [r14] "v"(r[14]), [r15] "v"(r[15]) : "scc");
    }
    if (V == 7) {   // readlane x2 + add + exec shift
      double src = t * 1e-9;
      asm volatile("s_mov_b64 exec, -1\n\t"
        R16("v_readlane_b32 s20, %[lo], 5\n\tv_readlane_b32 s21, %[hi], 5\n\tv_add_f64 %[t], %[t], s[20:21]\n\ts_lshl_b64 exec, exec, 1\n\t")
        "s_mov_b64 exec, -1" : [t] "+v"(t) : [lo] "v"((int)__double2loint(src)), [hi] "v"((int)__double2hiint(src)) : "scc", "s20", "s21");
    }
    if (V == 8) {   // add + LDS store of the running value (no exec games)
      asm volatile(R16("v_add_f64 %[t], %[t], %[b]\n\tds_write_b64 %[ad], %[t]\n\t") : [t] "+v"(t) : [b] "s"(inc), [ad] "v"(0u) : "memory");
    }
    if (V == 2) asm volatile(R16(STEP_C) : [t] "+v"(t) : [b] "s"(inc));
    if (V == 3) asm volatile(R16("v_fma_f64 %[t], %[t], 1.0, %[b]\n\t") : [t] "+v"(t) : [b] "s"(inc));
    if (V == 4) asm volatile(R16("v_add_f32 %[t], %[t], %[b]\n\t") : [t] "+v"(*(float*)&t) : [b] "s"((float)inc));
    if (V == 5) { double u = t * 0.5; asm volatile(R16("v_add_f64 %[t], %[t], %[b]\n\tv_add_f64 %[u], %[u], %[b]\n\t") : [t] "+v"(t), [u] "+v"(u) : [b] "s"(inc)); t += u; }
  }
  long long c1 = __builtin_readcyclecounter();
  long long m1 = wall_clock64();
  out[blockIdx.x * 64 + threadIdx.x] = t;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = m1 - m0; }
}
template <int V> void run(const char* name, int nblk) {
  double* out; long long* cyc; hipMalloc(&out, 8 * 64 * 4096); hipMalloc(&cyc, 16);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<V><<<nblk, 64>>>(out, 1e-3, iters, cyc); hipDeviceSynchronize();
  hipEventRecord(e0); k<V><<<nblk, 64>>>(out, 1e-3, iters, cyc); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  const double steps = iters * 16.0 * (V == 5 ? 2 : 1);
  printf("%-28s blocks %4d: %.3f ms  %.2f ns/op  shader cycles/op %.2f  wallclk(100MHz) ticks %lld -> %.2f GHz\n", name, nblk, ms,
         ms * 1e6 / steps, h[0] / steps, h[1], h[0] / (h[1] * 10.0) );
}
int main() {
  for (int nb : {256}) {
    run<0>("add_f64 chain", nb); run<1>("add_f64 + exec shift", nb); run<2>("add_f64 + s_nop", nb);
    run<3>("fma_f64 chain", nb); run<4>("add_f32 chain", nb); run<5>("2 indep add_f64 chains", nb);
    run<6>("ldsbcast + add + shift", nb); run<7>("2 readlane + add + shift", nb); run<8>("add + ds_write", nb);
  }
  return 0;
}

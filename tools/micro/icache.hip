// micro-benchmark: cost of streaming a loop body larger than the instruction cache (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#define BODY4 "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
template <int K4> __global__ void k(double* out, int iters, double a, double b) {
  double t0 = threadIdx.x, t1 = t0 + 1, t2 = t0 + 2, t3 = t0 + 3;
  for (int i = 0; i < iters; ++i) {
    asm volatile(".rept %6\n" BODY4 ".endr\n" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(a), "v"(b), "n"(K4));
  }
  out[blockIdx.x * 64 + threadIdx.x] = t0 + t1 + t2 + t3;
}
template <int K4> void run(int nblk, int wg) {
  double* out; (void)hipMalloc(&out, 8 * 64 * 16384);
  const long total = 1 << 22;              // instructions per wave
  const int iters = (int)(total / (4 * K4));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<K4><<<nblk, wg>>>(out, iters, 0.999, 1e-9); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<K4><<<nblk, wg>>>(out, iters, 0.999, 1e-9); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("body %4d KB  blocks %5d x %3d thr: %.3f ms  -> %.2f cycles/instr/wave @2.4GHz\n", K4 * 32 / 1024, nblk, wg, ms,
         ms * 1e-3 * 2.4e9 / total);
  (void)hipFree(out);
}
int main() {
  for (int nb : {256, 1024, 2048}) {
    run<256>(nb, 64); run<1024>(nb, 64); run<2048>(nb, 64); run<3072>(nb, 64); run<3840>(nb, 64);
  }
  return 0;
}

// What an (almost) empty launch costs on gfx950 as a function of the kernel's resources: LDS bytes per workgroup,
// registers per wave, workgroups.  Every wave loads one flag and leaves.  Times 200 back-to-back launches with events.
//   make -C tools/micro launchcost && tools/micro/launchcost
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDS, int REGS>
__global__ __launch_bounds__(64, 1) void probe(const int* flag, double* out) {
  __shared__ double s[LDS / 8];
  if (*flag == 0) return;
  // never taken at run time: keeps the LDS array and the registers in the kernel's resource record
  s[threadIdx.x] = out[threadIdx.x];
  __syncthreads();
  if (REGS >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (REGS >= 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
  if (REGS >= 384) asm volatile("v_accvgpr_write_b32 a127, 0" ::: "a127");
  if (REGS >= 512) asm volatile("v_accvgpr_write_b32 a255, 0" ::: "a255");
  out[threadIdx.x] = s[(threadIdx.x + 1) & 63];
}

template <int LDS, int REGS> void run(const int* flag, double* out, int wgs) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<LDS, REGS>), dim3(wgs), dim3(64), 0, 0, flag, out);
  hipEventRecord(a, 0);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((probe<LDS, REGS>), dim3(wgs), dim3(64), 0, 0, flag, out);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  printf("lds %6d regs %3d wgs %6d : %7.2f us per launch\n", LDS, REGS, wgs, ms * 1000.0 / 200);
}

int main() {
  int* flag;
  double* out;
  hipMalloc(&flag, 4);
  hipMalloc(&out, 4096);
  hipMemset(flag, 0, 4);
  for (int wgs : {256, 1024, 6144, 18432}) {
    run<1024, 64>(flag, out, wgs);
    run<12496, 64>(flag, out, wgs);
    run<32768, 64>(flag, out, wgs);
    run<33296, 64>(flag, out, wgs);
    run<65536, 64>(flag, out, wgs);
    run<1024, 256>(flag, out, wgs);
    run<1024, 384>(flag, out, wgs);
    run<1024, 512>(flag, out, wgs);
    run<33296, 384>(flag, out, wgs);
    run<33296, 512>(flag, out, wgs);
  }
  return 0;
}

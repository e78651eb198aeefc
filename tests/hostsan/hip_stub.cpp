// A stand-in for the HIP runtime, for the HOST layer of libworld_mi355 under AddressSanitizer / UBSan on a box without
// a GPU (tests/test_sanitizers.py).  Test infrastructure only: "device" memory is host heap (so a marshalling copy that
// is one byte too long is an ASan report), copies are memcpy, streams and events are tokens, and KERNELS DO NOT RUN --
// what a kernel would have written is the fill pattern of hipMalloc (finite doubles) or what a memset left.  The
// arithmetic is not under test here (tests/test_gpu_parity.py, on the device); the host side is: option validation,
// arena sizing, the `double**` gather / scatter of the drop-in entry points, the block cache, the error-handler path.
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

namespace {
std::atomic<int> g_device{0};
std::atomic<long> g_mallocs{0}, g_frees{0}, g_launches{0};
int g_devices = 1;                     // HIP_STUB_DEVICES=0: a box without a device
bool g_fail_malloc_after_set = false;
long g_fail_malloc_after = 0;
struct Init {
  Init() {
    if (const char* e = getenv("HIP_STUB_DEVICES")) g_devices = atoi(e);
    if (const char* e = getenv("HIP_STUB_FAIL_MALLOC_AFTER")) { g_fail_malloc_after_set = true; g_fail_malloc_after = atol(e); }
  }
} g_init;
}  // namespace

extern "C" {
long HipStubMallocs() { return g_mallocs.load(); }
long HipStubFrees() { return g_frees.load(); }
long HipStubLaunches() { return g_launches.load(); }

hipError_t hipGetDeviceCount(int* n) { *n = g_devices; return g_devices > 0 ? hipSuccess : hipErrorNoDevice; }
hipError_t hipGetDevice(int* d) { *d = g_device.load(); return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= g_devices) return hipErrorInvalidDevice; g_device = d; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) {
  memset(p, 0, sizeof(*p));
  p->multiProcessorCount = 256;
  strcpy(p->name, "hip_stub");
  return hipSuccess;
}
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stub error"; }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }

hipError_t hipMalloc(void** p, size_t bytes) {
  if (g_fail_malloc_after_set && g_mallocs.load() >= g_fail_malloc_after) { *p = nullptr; return hipErrorOutOfMemory; }
  ++g_mallocs;
  void* q = malloc(bytes ? bytes : 1);
  if (!q) return hipErrorOutOfMemory;
  // a finite pattern: what a kernel "wrote" (doubles 0.25, 0.5, ...; harmless as int32 / float too)
  double* d = (double*)q;
  for (size_t i = 0; i + 1 <= bytes / 8; ++i) d[i] = 0.25 * (double)(1 + (i & 1023));
  memset((char*)q + (bytes / 8) * 8, 0, bytes % 8);
  *p = q;
  return hipSuccess;
}
hipError_t hipFree(void* p) { if (p) ++g_frees; free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
  *p = calloc(1, bytes ? bytes : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) { if (n) memcpy(dst, src, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t) { if (n) memset(dst, v, n); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free((void*)s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free((void*)e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 2; return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { ++g_launches; return hipSuccess; }

// what the host halves of the .hip translation units reference for kernel registration and launches
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
static thread_local struct { dim3 g, b; size_t shm; hipStream_t st; } t_cfg;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t shm, hipStream_t st) { t_cfg.g = g; t_cfg.b = b; t_cfg.shm = shm; t_cfg.st = st; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* shm, hipStream_t* st) { *g = t_cfg.g; *b = t_cfg.b; *shm = t_cfg.shm; *st = t_cfg.st; return hipSuccess; }
}

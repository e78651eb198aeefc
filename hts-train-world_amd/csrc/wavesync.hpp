// wavesync.hpp -- synchronisation of a one-wavefront workgroup
#pragma once
#include <hip/hip_runtime.h>

namespace wm {

// Synchronisation inside a ONE-wavefront workgroup (64 threads): for such a workgroup the compiler
// lowers __syncthreads() to a wave barrier (no s_barrier) plus `s_waitcnt lgkmcnt(0)`.
// (A lighter form -- wavefront-scope fences around __builtin_amdgcn_wave_barrier(), which emits no
// wait at all -- aborted on the device in round 1 and is not used.)
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

}  // namespace wm

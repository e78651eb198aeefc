// harvest.hip -- Harvest F0 estimation (externs/WORLD_v2/src/harvest.cpp:43-1262).
// Not implemented on the device yet: the entry point reports WM_ERR_UNSUPPORTED (there is
// deliberately no CPU fallback).
#include "batch.hpp"

namespace wm {
int launch_harvest(Batch& b, const double* d_x, double* d_t, double* d_f0) {
  (void)b; (void)d_x; (void)d_t; (void)d_f0;
  return WM_ERR_UNSUPPORTED;
}
}  // namespace wm

/* stonemask.h -- drop-in for externs/WORLD_v2/src/world/stonemask.h:27-29. */
#ifndef WORLD_MI355_STONEMASK_H_
#define WORLD_MI355_STONEMASK_H_
#include "world/macrodefinitions.h"
WORLD_BEGIN_C_DECLS

/* replaces StoneMask, stonemask.cpp:211-217 */
void StoneMask(const double *x, int x_length, int fs, const double *temporal_positions,
               const double *f0, int f0_length, double *refined_f0);

WORLD_END_C_DECLS
#endif

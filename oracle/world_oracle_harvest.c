/* placeholder until the Harvest restatement lands (TEST INFRASTRUCTURE ONLY) */
#include "world_oracle.h"
#include <stdlib.h>
void orc_harvest(const double *x, int x_length, int fs, double f0_floor, double f0_ceil,
                 double frame_period, double *t, double *f0) {
  (void)x; (void)x_length; (void)fs; (void)f0_floor; (void)f0_ceil; (void)frame_period; (void)t; (void)f0;
  abort();
}

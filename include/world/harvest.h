/* harvest.h -- drop-in for externs/WORLD_v2/src/world/harvest.h (struct :16-20, functions :35-59). */
#ifndef WORLD_MI355_HARVEST_H_
#define WORLD_MI355_HARVEST_H_
#include "world/macrodefinitions.h"
WORLD_BEGIN_C_DECLS

typedef struct {
  double f0_floor;
  double f0_ceil;
  double frame_period; /* ms */
} HarvestOption;

/* replaces Harvest, harvest.cpp:1223-1255 */
void Harvest(const double *x, int x_length, int fs, const HarvestOption *option,
             double *temporal_positions, double *f0);
/* replaces InitializeHarvestOption, harvest.cpp:1257-1262 */
void InitializeHarvestOption(HarvestOption *option);
/* replaces GetSamplesForHarvest, harvest.cpp:1219-1221 */
int GetSamplesForHarvest(int fs, int x_length, double frame_period);

WORLD_END_C_DECLS
#endif

/* d4c.h -- drop-in for externs/WORLD_v2/src/world/d4c.h (struct :16-18, functions :35-46). */
#ifndef WORLD_MI355_D4C_H_
#define WORLD_MI355_D4C_H_
#include "world/macrodefinitions.h"
WORLD_BEGIN_C_DECLS

typedef struct {
  double threshold;
} D4COption;

/* replaces D4C, d4c.cpp:337-397 */
void D4C(const double *x, int x_length, int fs, const double *temporal_positions,
         const double *f0, int f0_length, int fft_size, const D4COption *option,
         double **aperiodicity);
/* replaces InitializeD4COption, d4c.cpp:399-401 */
void InitializeD4COption(D4COption *option);

WORLD_END_C_DECLS
#endif

/* synthesis.h -- drop-in for externs/WORLD_v2/src/world/synthesis.h:30-32. */
#ifndef WORLD_MI355_SYNTHESIS_H_
#define WORLD_MI355_SYNTHESIS_H_
#include "world/macrodefinitions.h"
WORLD_BEGIN_C_DECLS

/* replaces Synthesis, synthesis.cpp:338-397 */
void Synthesis(const double *f0, int f0_length, const double *const *spectrogram,
               const double *const *aperiodicity, int fft_size, double frame_period, int fs,
               int y_length, double *y);

WORLD_END_C_DECLS
#endif

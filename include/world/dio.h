/* dio.h -- drop-in for externs/WORLD_v2/src/world/dio.h (struct :16-23, functions :38-61).
 * Implemented by libworld_mi355.so: host pointers in/out, HIP kernels inside. */
#ifndef WORLD_MI355_DIO_H_
#define WORLD_MI355_DIO_H_
#include "world/macrodefinitions.h"
WORLD_BEGIN_C_DECLS

typedef struct {
  double f0_floor;
  double f0_ceil;
  double channels_in_octave;
  double frame_period; /* ms */
  int speed;           /* decimation ratio, 1..12 */
  double allowed_range;
} DioOption;

/* replaces Dio, dio.cpp:642-647.  One divergence: when f0_length <= voice_range_minimum the
 * reference returns without writing f0 (dio.cpp:266); this implementation writes zeros. */
void Dio(const double *x, int x_length, int fs, const DioOption *option,
         double *temporal_positions, double *f0);
/* replaces InitializeDioOption, dio.cpp:649-665 */
void InitializeDioOption(DioOption *option);
/* replaces GetSamplesForDIO, dio.cpp:638-640 */
int GetSamplesForDIO(int fs, int x_length, double frame_period);

WORLD_END_C_DECLS
#endif

/* cheaptrick.h -- drop-in for externs/WORLD_v2/src/world/cheaptrick.h (struct :16-20, functions :38-80). */
#ifndef WORLD_MI355_CHEAPTRICK_H_
#define WORLD_MI355_CHEAPTRICK_H_
#include "world/macrodefinitions.h"
WORLD_BEGIN_C_DECLS

typedef struct {
  double q1;
  double f0_floor;
  int fft_size;
} CheapTrickOption;

/* replaces CheapTrick, cheaptrick.cpp:200-228.  spectrogram is f0_length caller-allocated rows
 * of fft_size/2+1 doubles (arbitrary row pointers are honoured). */
void CheapTrick(const double *x, int x_length, int fs, const double *temporal_positions,
                const double *f0, int f0_length, const CheapTrickOption *option,
                double **spectrogram);
/* replaces InitializeCheapTrickOption, cheaptrick.cpp:230-239 */
void InitializeCheapTrickOption(int fs, CheapTrickOption *option);
/* replaces GetFFTSizeForCheapTrick, cheaptrick.cpp:191-194 */
int GetFFTSizeForCheapTrick(int fs, const CheapTrickOption *option);
/* replaces GetF0FloorForCheapTrick, cheaptrick.cpp:196-198 */
double GetF0FloorForCheapTrick(int fs, int fft_size);

WORLD_END_C_DECLS
#endif

/* codec.h -- drop-in for externs/WORLD_v2/src/world/codec.h (declarations :23-88).
 *
 * Same five entry points and signatures; one call = one utterance, host pointers, caller-allocated
 * `double**` rows, exactly like the reference.  The batched device forms are in world_mi355.h.
 *
 * DecodeAperiodicity: the reference's header names the 4th/5th arguments (fft_size,
 * number_of_aperiodicities) but its definition (codec.cpp:237-238) takes them as
 * (number_of_aperiodicities, fft_size).  Both are `int`, so the ABI is positional and this library
 * behaves like the DEFINITION: callers written against the header's names get the reference's
 * behaviour, including that swap.
 */
#ifndef WORLD_MI355_CODEC_H_
#define WORLD_MI355_CODEC_H_
#include "world/macrodefinitions.h"
WORLD_BEGIN_C_DECLS

/* replaces GetNumberOfAperiodicities, codec.cpp:212-215 */
int GetNumberOfAperiodicities(int fs);
/* replaces CodeAperiodicity, codec.cpp:217-235 */
void CodeAperiodicity(const double * const *aperiodicity, int f0_length, int fs, int fft_size,
                      int number_of_aperiodicities, double **coded_aperiodicity);
/* replaces DecodeAperiodicity, codec.cpp:237-266 (4th argument = number of aperiodicities, 5th = fft size) */
void DecodeAperiodicity(const double * const *coded_aperiodicity, int f0_length, int fs, int fft_size,
                        int number_of_aperiodicities, double **aperiodicity);
/* replaces CodeSpectralEnvelope, codec.cpp:268-295 */
void CodeSpectralEnvelope(const double * const *spectrogram, int f0_length, int fs, int fft_size,
                          int number_of_dimensions, double **coded_spectral_envelope);
/* replaces DecodeSpectralEnvelope, codec.cpp:297-324 */
void DecodeSpectralEnvelope(const double * const *coded_spectral_envelope, int f0_length, int fs,
                            int fft_size, int number_of_dimensions, double **spectrogram);

WORLD_END_C_DECLS
#endif

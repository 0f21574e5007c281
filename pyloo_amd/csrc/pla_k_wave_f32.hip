// wave-per-observation kernels (pla_wave.h, pla_is.h) for f32 rows: see pla_k_wave.inc
#define PLA_K_DTYPE float
#include "pla_k_wave.inc"

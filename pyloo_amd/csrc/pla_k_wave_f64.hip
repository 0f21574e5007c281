// wave-per-observation kernels (pla_wave.h, pla_is.h) for f64 rows: see pla_k_wave.inc
#define PLA_K_DTYPE double
#include "pla_k_wave.inc"

// chunked wave-per-observation kernels (pla_chunked.h) for f32 rows: see pla_k_chunked.inc
#define PLA_K_DTYPE float
#include "pla_k_chunked.inc"

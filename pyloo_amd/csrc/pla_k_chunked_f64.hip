// chunked wave-per-observation kernels (pla_chunked.h) for f64 rows: see pla_k_chunked.inc
#define PLA_K_DTYPE double
#include "pla_k_chunked.inc"

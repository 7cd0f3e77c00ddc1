// One tile family of the implicit-GEMM conv kernel (see conv_kernels.h).
#include "conv_kernels.h"

void vsd_launch_conv_128x128(const ConvParams& p, int grid, int stages, hipStream_t s) { launch<128, 128>(p, grid, stages, s); }

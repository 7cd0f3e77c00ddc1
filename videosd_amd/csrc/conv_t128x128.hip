// One tile family of the implicit-GEMM conv kernel (see conv_kernels.h).
#include "conv_kernels.h"

void vsd_launch_conv_128x128(const ConvParams& p, int grid, int stages, hipStream_t s) { launch<128, 128>(p, grid, stages, s); }
void vsd_launch_conv_group_128x128(const ConvGroup& g, int grid, int stages, hipStream_t s) { launch_group<128, 128>(g, grid, stages, s); }

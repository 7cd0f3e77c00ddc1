// One tile family of the implicit-GEMM conv kernel (see conv_kernels.h).
#include "conv_kernels.h"

void vsd_launch_conv_64x64(const ConvParams& p, int grid, int stages, hipStream_t s) { launch<64, 64>(p, grid, stages, s); }
void vsd_launch_conv_group_64x64(const ConvGroup& g, int grid, int stages, hipStream_t s) { launch_group<64, 64>(g, grid, stages, s); }

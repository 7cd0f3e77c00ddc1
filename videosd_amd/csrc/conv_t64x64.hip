// One tile family of the implicit-GEMM conv kernel (see conv_kernels.h).
#include "conv_kernels.h"

void vsd_launch_conv_64x64(const ConvParams& p, int grid, int stages, hipStream_t s) {
  if (stages == 8) FastLaunch<64, 64, 8, false>::go(p, grid, s);  // weight-streaming layers (tiny M, deep K): 7 tiles of 16 KB in flight per workgroup
  else launch<64, 64>(p, grid, stages, s);
}

// One tile family of the implicit-GEMM conv kernel (see conv_kernels.h).
#include "conv_kernels.h"

void vsd_launch_conv_256x128(const ConvParams& p, int grid, int stages, hipStream_t s) {
  // 2x2 waves of 128x64: 85 FLOP per byte staged through LDS (128x128: 64, 64x64: 32)
  if (stages >= 8) launch8<256, 128>(p, grid, stages, s);  // 4x2 waves of 64x64
  else if (stages == 3) FastLaunch<256, 128, 3, false>::go(p, grid, s);
  else FastLaunch<256, 128, 3, true>::go(p, grid, s);
}

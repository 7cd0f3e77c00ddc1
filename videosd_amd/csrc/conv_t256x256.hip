// One tile family of the implicit-GEMM conv kernel (see conv_kernels.h): 256 x 256 on eight waves (4 x 2 waves of 64 x 128).
// 128 FLOP per byte staged through LDS (256 x 128: 85, 128 x 128: 64, 64 x 128: 43) -- the L2 -> LDS fill is what the GEMM-form
// layers bind on: this is the tile that asks least of it.  Two ring slots of 64 KB, buffer-load path only,
// unsplit; the epilogue runs in four row bands of 64 (conv_epilogue.inc).
#include "conv_kernels.h"

void vsd_launch_conv_256x256(const ConvParams& p, int grid, int stages, hipStream_t s) { launch8<256, 256, 2>(p, grid, stages, s); }

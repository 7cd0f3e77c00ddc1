// 3x3 conv, 64 -> 64 channels, with the WEIGHTS RESIDENT IN REGISTERS and persistent workgroups (pipeline 10): the TAESD
// encoder / decoder blocks at the large image sizes (512 x 512 x B frames: 10 240 patches of 8 x 16 pixels per conv).
//
// The halo-patch kernel (conv_halo.hip) streams the nine 8 KB weight tiles of a 64-channel block through its LDS ring for
// EVERY 128-pixel patch, reads both MFMA operands from LDS and transposes its accumulators through LDS to store them.
// Here a wave keeps the weights of ITS 32 output channels for all nine taps in registers (36 fragments = 144 VGPRs, loaded
// once per workgroup), the grid is two workgroups per CU slot-wise (512; the register count allows one resident per CU), each
// walks patches blockIdx.x, + gridDim.x, ...; only the pixel fragments come from LDS (8 reads per 16 MFMAs, one tap
// ahead), the next patch's halo DMA (24 KB, double buffered) is in flight under the current patch's 144 MFMAs per wave,
// and the MFMAs run with the operands SWAPPED (weights first): an accumulator tile is 16 channels x 16 pixels, a lane holds
// 4 consecutive channels of one pixel = 8 contiguous output bytes, so the epilogue is register work + one 8-byte store per
// tile -- no LDS transpose, no barrier.  Same patch geometry, swizzles, nearest-resize fold and plain epilogue (bias + time
// vector, ReLU / SiLU, one residual, ReLU after it) as conv_halo_kernel<64, 2, .>; bit-identical results.
//
// MEASURED on MI355X (scripts/taesd_bench.py, scripts/resident_probe.py; round 2), three versions:
//   (1) weights resident in LDS, everything else as the halo kernel: 193 us at 5 x 512 x 512 against 179-182 us for the halo
//       kernel -- the weight re-streaming (755 MB of L2 -> LDS fill per conv) is NOT what holds these layers at ~500 TFLOP/s;
//   (2) weights in registers, transposing epilogue: 206 us.  Stamps per patch (cycles): top barrier 240 | halo DMA issue +
//       residual loads 1 560 | nine taps 4 800 (144 MFMAs = 2 300) | epilogue 4 260 (two LDS transposes, four barriers);
//   (3) this version (DMA index math hoisted, fragments one tap ahead, register epilogue): 170 us (-7 %); stamps 210 |
//       1 920 | 3 370 | 3 030.  With one wave per SIMD (390 registers) every scalar / vector / memory instruction's issue
//       latency is exposed: the ~5 k cycles of non-MFMA work per patch are the bound, and the halo kernel hides the same work
//       behind the second and third workgroup of its CU.  At 256 x 256 and below, and for one frame, the halo kernel wins.
//   Kept as a tuner candidate behind VSD_TUNE_RESIDENT=1 (explicit pipeline = 10 always works) with its parity tests.
#include "conv_kernels.h"

namespace {

#ifdef RS_PROBE  // development: shader-clock totals per phase of wave 0 of workgroup 0, written to the workspace pointer
#define RSP(I_) { if (tid == 0) { long long t_ = __builtin_readcyclecounter(); pacc[I_] += t_ - plast; plast = t_; } }
#else
#define RSP(I_)
#endif

constexpr int RS_BN = 64, RS_BM = 128, RS_PW = 16, RS_PH = 8, RS_HW = RS_PW + 2;  // 18 halo columns
constexpr int RS_HUSED = (RS_PH + 2) * RS_HW;                                     // 180 halo rows (pixels)
constexpr int RS_HROWS = 192;                                                     // padded to 6 wave-instructions per wave
constexpr int RS_AI = RS_HROWS / 32, RS_BR = RS_BN / 32;
constexpr int RS_A_HALFS = RS_HROWS * BK;
constexpr int RS_LDS_BYTES = 2 * RS_A_HALFS * 2;  // 49 152: two workgroups (and more) per CU

// workgroup barrier for LDS data only (__syncthreads() would also drain the vector-memory counter: the output stores of
// the previous patch would then be waited for at every barrier)
__device__ __forceinline__ void rs_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(256) void conv_resident_kernel(const ConvParams p) {
  prefetch_kernargs();
  __shared__ __attribute__((aligned(16))) unsigned char smem[RS_LDS_BYTES];
  half_t* Abuf = reinterpret_cast<half_t*>(smem);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  constexpr int OOB = (int)0x80000000;
  constexpr int FM = 4, FN = 2, TM = 64, TN = 32;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;

  const int ppr = (p.wo + RS_PW - 1) / RS_PW, tpi = ((p.ho + RS_PH - 1) / RS_PH) * ppr;
  const int npatch = p.batch * tpi;
  const int anr0 = (int)((size_t)p.batch * p.img_in * p.c0 * 2);
  const int cs2 = p.c0 * 2;

  // ---- this wave's weights, once: output channels [32 wn, 32 wn + 32), all nine taps, as MFMA B fragments
  // (lane = channel fr of fragment j, k chunk fq: 8 consecutive k of one weight row = one 16-byte load)
  half8 bw[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bw[t][ks][j] = *reinterpret_cast<const half8*>(p.w + (size_t)(wn * TN + j * 16 + fr) * p.Kp + t * BK + ks * 32 + fq * 8);
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)p.src0, 0, anr0, 0x00020000);
  // halo DMA of patch `pt` into buffer `buf`: this wave's instructions q cover halo rows 8 (wave + 4 q) .. + 7.  The lane's
  // halo position and swizzled slot do not depend on the patch: computed once.
  int hyx[RS_AI], hsw[RS_AI];
#pragma unroll
  for (int q = 0; q < RS_AI; ++q) {
    const int row = 8 * (wave + 4 * q) + (lane >> 3);
    const int hy = row / RS_HW, hx = row - hy * RS_HW;
    hyx[q] = row < RS_HUSED ? (hy << 8) | hx : -1;
    hsw[q] = ((lane & 7) ^ (row & 7)) << 4;
  }
  const float inv_tpi = 1.0f / (float)tpi, inv_ppr = 1.0f / (float)ppr;
  auto patch_origin = [&](int pt, int& img, int& y0, int& x0) __attribute__((always_inline)) {
    img = (int)(((float)pt + 0.5f) * inv_tpi);
    const int trem = pt - img * tpi;
    const int pr = (int)(((float)trem + 0.5f) * inv_ppr);
    y0 = pr * RS_PH;
    x0 = (trem - pr * ppr) * RS_PW;
  };
  auto issue_patch = [&](int pt, int buf) __attribute__((always_inline)) {
    int img, y0, x0;
    patch_origin(pt, img, y0, x0);
    const int pix0 = img * p.img_in;
#pragma unroll
    for (int q = 0; q < RS_AI; ++q) {
      const int y = y0 - 1 + (hyx[q] >> 8), x = x0 - 1 + (hyx[q] & 255);
      const bool in = hyx[q] >= 0 && (unsigned)y < (unsigned)p.hi && (unsigned)x < (unsigned)p.wi;
      const int sy = (int)(((unsigned)y * p.rmul_y) >> p.rshift), sx = (int)(((unsigned)x * p.rmul_x) >> p.rshift);
      const int vo = in ? __mul24(pix0 + sy * p.ws + sx, cs2) + hsw[q] : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr_t)(Abuf + buf * RS_A_HALFS + 8 * (wave_s + 4 * q) * BK), 16, vo, 0, 0, 0);
    }
  };

  // The MFMAs run with the operands swapped (weights first): an accumulator tile is 16 CHANNELS x 16 pixels, a lane holds 4
  // consecutive channels (4 fq .. + 3 of channel group j) of ONE pixel (fr) -- 8 contiguous output bytes.  The epilogue
  // is then register work and one 8-byte store per tile: no LDS transpose, no barrier (measured on the first version of
  // this kernel: the transposing epilogue cost 4.3 k cycles per patch, as much as the nine taps of MFMAs).
  float brv[2][4];  // bias + time vector of this lane's channels
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = wn * TN + j * 16 + 4 * fq + r;
      brv[j][r] = (p.bias ? (float)p.bias[c] : 0.f) + (p.rowvec ? (float)p.rowvec[c] : 0.f);
    }
  const int act = p.act & 0xff;
  const int actk = (act == VSD_ACT_RELU && (p.act & VSD_ACT_POST)) ? 3 : act == VSD_ACT_RELU ? 1 : act == VSD_ACT_SILU ? 2 : 0;

  int hr0[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) hr0[i] = (wm * (TM / RS_PW) + i) * RS_HW + fr;

#ifdef RS_PROBE
  long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long plast = __builtin_readcyclecounter();
#endif
  int pt = blockIdx.x;
  if (pt < npatch) issue_patch(pt, 0);
  RSP(0)
  for (int it = 0; pt < npatch; pt += gridDim.x, ++it) {
    const int cur = it & 1;
    // This patch's halo has landed: every wave waited for its own DMA instructions after the previous patch's MFMAs (below),
    // the first time here; the barrier publishes them and ends the previous epilogue's use of the other buffer.
    if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    rs_barrier();
    RSP(1)
    if (pt + (int)gridDim.x < npatch) issue_patch(pt + gridDim.x, cur ^ 1);

    int img, y0, x0;
    patch_origin(pt, img, y0, x0);
    // this lane's pixels: column fr of patch rows 4 wm + i
    int orow[FM];  // output row (pixel index) or -1 outside the image
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int y = y0 + wm * 4 + i, x = x0 + fr;
      orow[i] = (y < p.ho && x < p.wo) ? img * p.hw_out + y * p.wo + x : -1;
    }
    // residual values of this lane's outputs: requested now, used after the MFMAs
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    half4 rpre[FM][FN];
    if (p.residual) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          rpre[i][j] = *reinterpret_cast<const half4*>(p.residual + (orow[i] >= 0 ? (size_t)orow[i] * p.ldr + wn * TN + j * 16 + 4 * fq : 0));
    }
    RSP(2)
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const half_t* a = Abuf + cur * RS_A_HALFS;
    // A fragments one tap ahead of their MFMAs (one wave per SIMD: nobody else hides the LDS latency)
    half8 af[2][2][FM];
    auto fetch = [&](int tap) __attribute__((always_inline)) {
      const int toff = (tap / 3) * RS_HW + (tap % 3);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int hr = hr0[i] + toff;
        const half_t* row = a + hr * BK;
        const int sw = hr & 7;
        af[tap & 1][0][i] = *reinterpret_cast<const half8*>(row + ((fq ^ sw) << 3));
        af[tap & 1][1][i] = *reinterpret_cast<const half8*>(row + (((4 + fq) ^ sw) << 3));
      }
    };
    fetch(0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) fetch(tap + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bw[tap][ks][j], af[tap & 1][ks][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    RSP(3)
    // the next patch's halo (issued a whole patch of MFMAs ago) and the residual values: all loads, nothing exposed.  Waiting
    // HERE, before this patch's output stores are issued, keeps the stores out of every later wait.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RSP(4)
    // ---- epilogue from registers: 4 channels of one pixel per tile and lane
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        half4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[i][j][r] + brv[j][r];
          if (actk == 1) x = fmaxf(x, 0.f);
          if (actk == 2) x = silu_f(x);
          if (p.residual) x += (float)rpre[i][j][r];
          if (actk == 3) x = fmaxf(x, 0.f);
          o[r] = (half_t)x;
        }
        if (orow[i] >= 0) *reinterpret_cast<half4*>(p.out + (size_t)orow[i] * p.ldo + wn * TN + j * 16 + 4 * fq) = o;
      }
    RSP(5)
  }
#ifdef RS_PROBE
  if (tid == 0 && blockIdx.x == 0 && p.ws_partial)
    for (int i = 0; i < 8; ++i) reinterpret_cast<long long*>(p.ws_partial)[i] = pacc[i];
#endif
}

}  // namespace

// grid: two workgroups per CU (or one per patch when there are fewer); the caller has checked eligibility
void vsd_launch_conv_resident(const ConvParams& p, int patches, hipStream_t s) {
  const int grid = patches < 512 ? patches : 512;
  hipLaunchKernelGGL(conv_resident_kernel, dim3(grid), dim3(256), 0, s, p);
}

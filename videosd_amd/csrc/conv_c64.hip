// 3x3 stride-1 convs with 64 input and 64 output channels -- every conv of a TAESD block (the reference's
// AutoencoderTiny at lcm_controlnet.py:299 and :594) -- as ONE persistent launch with the weights in registers (pipeline 10).
//
// The halo-patch kernel (conv_halo.hip) runs such a layer as 5 120 workgroups (five 512 x 512 frames) that each fetch a patch, wait
// for it, run nine taps of one channel block behind nine barriers and transpose 256 x 64 accumulators through LDS: 2 300 cycles of
// MFMA in a workgroup life of ~10 000 (541 TFLOP/s).  Here a workgroup stays on its CU and walks over patches:
//   * all 64 x 576 weights live in REGISTERS for the workgroup's life: wave (wm, wn) of the 4 x 2 keeps the 32 output channels of
//     half wn for all 18 K steps (36 MFMA operands = 144 registers); two waves per SIMD (<= 256 registers each);
//   * 16 x 16-pixel output patches, the (16+2) x (16+2) x 64-channel input patch by LDS-DMA into one of TWO buffers: patch i+1
//     is in flight while patch i is computed; ONE barrier per patch (none per tap: no operand streams through LDS but the patch);
//   * operands swapped: D = W X^T, so a lane's accumulators are 4 + 4 CONSECUTIVE channels of one pixel (the weight rows of a
//     wave's two channel fragments are interleaved in fours): the epilogue -- bias, activation, residual, ReLU -- is register work
//     and one 16-byte store per pixel and lane, no LDS transpose, no barrier: a wave's epilogue runs beside its SIMD partner's MFMAs.
// Round 2 measured a one-wave-per-SIMD form of the same idea at -7 % (docs/NOTEBOOK.md: every instruction's issue latency exposed,
// 8-byte stores); this form has the second wave, 16-byte stores and half the patch fetches per pixel.
// Same sums in the same order as the halo kernel (tap outer, two K steps of 32 inner, fp32 accumulators; bias then activation then
// residual): parity tests hold it to the halo form's bits.
// Algorithmic work per launch: 2 * M * 64 * 576 FLOP; bytes: M * 64 * 2 in + M * 64 * 2 out (+ the residual) + 72 KB of weights.
#include "conv_kernels.h"

namespace {

constexpr int C64_PW = 16, C64_PH = 16;                      // output patch 16 x 16 pixels
constexpr int C64_HS = 24;                                   // LDS rows per halo row: 18 pixels + 6 unused (a multiple of 8: see below)
constexpr int C64_NW = 8;                                    // waves: 4 (patch rows) x 2 (channel halves)
constexpr int C64_NJ = (C64_PH + 2) * (C64_HS / 8);          // 60 wave-instructions of 8 LDS rows per patch
constexpr int C64_AI = (C64_NJ + C64_NW - 1) / C64_NW;       // 7 per wave: 56 issued, the last two land in padding rows
constexpr int C64_A_HALFS = C64_AI * C64_NW * 8 * BK;        // 448 rows of 64 halfs = 57 344 bytes per buffer
constexpr int C64_RTAPS = 5;                                 // taps whose weights live in registers (the other three: LDS, fragment-major)
constexpr int C64_WLDS_BYTES = (9 - C64_RTAPS) * 2 * 2 * 2 * 1024;  // [tap][K step][channel half][fragment] x 1 KB = 24 KB

// ACT: 0 none, 1 ReLU, 2 SiLU, 3 ReLU after the residual; RES: a residual is added
template <int ACT, bool RES>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_c64_kernel(const ConvParams p) {
  VSD_CUT(VSD_CUT_CONV_HALO, p.cut)
  prefetch_kernargs();
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * C64_A_HALFS * 2 + C64_WLDS_BYTES];
  unsigned char* wlds = smem + 2 * C64_A_HALFS * 2;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  constexpr int OOB = (int)0x80000000;
  constexpr int NW = C64_NW, AI = C64_AI, HS = C64_HS, PW = C64_PW, PH = C64_PH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wave_s >> 1, wn = wave_s & 1;
  const int fp = lane & 15, fq = lane >> 4;  // fragment pixel (B operand column) / K chunk; D: channel block fq, pixel fp

  // ---- the weights of this wave's 32 output channels, all 18 K steps.  Row r of channel fragment j is channel
  // 32 wn + 8 (r >> 2) + 4 j + (r & 3): lane (fp, fq) then holds D rows 4 fq .. 4 fq + 3 of both fragments = channels
  // 32 wn + 8 fq .. + 7, contiguous.
  // Taps 0-4 (10 K steps, 80 registers) stay in registers; taps 5-8 go to LDS in fragment order (a wave-read is 1 KB contiguous:
  // conflict-free), 16 ds_read_b128 per wave and patch -- with all 18 steps in registers the kernel spilled 14-51 of its 256.
  half8 wreg[2 * C64_RTAPS][2];
  {
    const int r = fp;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ch = 32 * wn + 8 * (r >> 2) + 4 * j + (r & 3);
      const half_t* wr = p.w + (size_t)ch * p.Kp + 8 * fq;
#pragma unroll
      for (int s = 0; s < 2 * C64_RTAPS; ++s) wreg[s][j] = *reinterpret_cast<const half8*>(wr + 32 * s);
      if (wm == 0) {
#pragma unroll
        for (int s = 2 * C64_RTAPS; s < 18; ++s)
          *reinterpret_cast<half8*>(wlds + ((((s - 2 * C64_RTAPS) * 2 + wn) * 2 + j) << 10) + lane * 16) = *reinterpret_cast<const half8*>(wr + 32 * s);
      }
    }
  }
  const int ch0 = 32 * wn + 8 * fq;  // this lane's 8 output channels
  // bias and time vector stay fp16 in four registers each (converted at use: the sum 0 + bias + rowvec in fp32 as the halo kernel's)
  const half8 z8 = (half8){0, 0, 0, 0, 0, 0, 0, 0};
  const half8 bias8 = p.bias ? *reinterpret_cast<const half8*>(p.bias + ch0) : z8;
  const half8 rv8 = p.rowvec ? *reinterpret_cast<const half8*>(p.rowvec + ch0) : z8;

  // ---- LDS image of a patch: halo pixel (hy, hx), hy < 18, hx < 18, is row hy * 24 + hx (128 bytes: its 64 channels), the 16-byte
  // chunks of a row XOR-swizzled by (row & 7) = (hx & 7).  24 rows per halo row: (1) an LDS-DMA wave-instruction (8 rows, 1 KB,
  // lane-linear) never straddles two halo rows -- instruction j covers hy = j / 3 (a scalar), hx = 8 (j % 3) + (lane >> 3); (2) the
  // swizzle of a fragment read depends on the tap's kx alone, so ALL 72 fragment reads of a patch are three per-lane base addresses
  // (one per kx, and their ^ 64 for the second K step) plus compile-time offsets.
  const int alc = ((lane & 7) ^ (lane >> 3)) << 4;  // source chunk of this lane's LDS slot (rows 8 j + (lane >> 3): row & 7 = lane >> 3)
  const int hxl = lane >> 3;
  const int anr0 = (int)((size_t)p.batch * p.img_in * p.c0 * 2);
  const int ppr = (p.wo + PW - 1) / PW, tpi = ((p.ho + PH - 1) / PH) * ppr;

  // patch t -> (image, origin); issue its input patch into buffer BUF_
#define C64_COORDS(T_, IMG_, Y0_, X0_)                         \
  const int IMG_ = fdiv((T_), p.fd_tpi);                       \
  const int trem_##IMG_ = (T_) - IMG_ * tpi;                   \
  const int prow_##IMG_ = fdiv(trem_##IMG_, p.fd_ppr);         \
  const int Y0_ = prow_##IMG_ * PH, X0_ = (trem_##IMG_ - prow_##IMG_ * ppr) * PW;
#define C64_ISSUE(IMG_, Y0_, X0_, BUF_)                                                                              \
  {                                                                                                                  \
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.src0, 0, anr0, 0x00020000);        \
    half_t* dst_ = reinterpret_cast<half_t*>(smem) + (BUF_) * C64_A_HALFS;                                           \
    const int pix0_ = (IMG_) * p.img_in;                                                                             \
    _Pragma("unroll") for (int q = 0; q < AI; ++q) {                                                                 \
      const int j = wave_s + NW * q;                                                                                 \
      const int hy = j / 3, hx = 8 * (j - 3 * hy) + hxl;                                                             \
      const int y = (Y0_) - 1 + hy, x = (X0_) - 1 + hx;                                                              \
      const bool in = (unsigned)y < (unsigned)p.hi && (unsigned)x < (unsigned)p.wi && hx < PW + 2 && hy < PH + 2;    \
      const int sy = (int)(((unsigned)y * p.rmul_y) >> p.rshift), sx = (int)(((unsigned)x * p.rmul_x) >> p.rshift);  \
      const int vo_ = in ? (pix0_ + sy * p.ws + sx) * (BK * 2) + alc : OOB;                                          \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(dst_ + 8 * j * BK), 16, vo_, 0, 0, 0);               \
    }                                                                                                                \
  }

  // fragment reads: per-lane byte offsets inside a buffer for tap column kx, K step 0 (K step 1: ^ 64); row (4 wm + i + ky) adds
  // (i + ky) * 24 * 128 at compile time
  int fb[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) fb[kx] = ((4 * wm) * HS + fp + kx) * (BK * 2) + ((fq ^ ((fp + kx) & 7)) << 4);

  const int ntiles = p.tiles_m;
  int t = blockIdx.x;
  if (t < ntiles) {
    C64_COORDS(t, img, y0, x0)
    C64_ISSUE(img, y0, x0, 0)
  }
  const int onr = (int)((size_t)p.M * p.ldo * 2);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (the first patch has no stores in front of it; the LDS weight stores are done)
  __syncthreads();  // the LDS-resident weight fragments (written by waves 0 / 1) are visible to every wave
  int buf = 0;
  for (; t < ntiles; t += gridDim.x, buf ^= 1) {
    C64_COORDS(t, img, y0, x0)
    // This patch has landed (every wave's pieces: barrier), and every wave is done reading the other buffer.
    // (vmcnt retires in order and counts stores: the only operations younger than this patch's DMA are the previous patch's four
    //  output stores -- buffer stores that are ALWAYS issued, pixels outside the image as out-of-range offsets -- so "all but four")
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // this lane's output pixels: patch rows 4 wm + i, column fp
    const int ox = x0 + fp;
    int mrow[4];
    bool mok[4];
    half8 rres[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int oy = y0 + 4 * wm + i;
      mok[i] = oy < p.ho && ox < p.wo;
      mrow[i] = img * p.hw_out + oy * p.wo + ox;
      if (RES) rres[i] = *reinterpret_cast<const half8*>(p.residual + (mok[i] ? (size_t)mrow[i] * p.ldr + ch0 : 0));
    }
    const int tn = t + gridDim.x;
    if (tn < ntiles) {
      C64_COORDS(tn, img2, y2, x2)
      C64_ISSUE(img2, y2, x2, buf ^ 1)
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned char* a = smem + buf * (C64_A_HALFS * 2);
    const unsigned char* a0[3];
    const unsigned char* a1[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      a0[kx] = a + fb[kx];
      a1[kx] = a + (fb[kx] ^ 64);
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
      half8 xf[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xf[0][i] = *reinterpret_cast<const half8*>(a0[kx] + (i + ky) * (HS * BK * 2));
        xf[1][i] = *reinterpret_cast<const half8*>(a1[kx] + (i + ky) * (HS * BK * 2));
      }
      half8 wf[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          wf[ks][j] = tap < C64_RTAPS ? wreg[2 * tap + ks][j]
                                      : *reinterpret_cast<const half8*>(wlds + (((((tap - C64_RTAPS) * 2 + ks) * 2 + wn) * 2 + j) << 10) + lane * 16);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][j], xf[ks][i], acc[i][j], 0, 0, 0);
    }
    // ---- epilogue in registers: channels ch0 .. ch0 + 7 of pixel (4 wm + i, fp)
    // (descriptor made next to its use: hipcc drops the host stub of a kernel that reads one declared in an outer scope)
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, onr, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      half8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float brv = 0.f;
        if (p.bias) brv += (float)bias8[e];
        if (p.rowvec) brv += (float)rv8[e];
        float x = acc[i][e >> 2][e & 3] + brv;
        if (ACT == 1) x = fmaxf(x, 0.f);
        if (ACT == 2) x = silu_f(x);
        if (RES) x += (float)rres[i][e];
        if (ACT == 3) x = fmaxf(x, 0.f);
        o[e] = (half_t)x;
      }
      __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&o), ors, mok[i] ? (mrow[i] * p.ldo + ch0) * 2 : OOB, 0, VSD_OUT_AUX);
    }
  }
#undef C64_COORDS
#undef C64_ISSUE
}

// ---- the same for Cout <= 8 (TAESD's last decoder conv, 64 -> 3 channels at full image size: lcm_controlnet.py:594; its encoder's
// 64 -> 4 projection): HBM-bound on reading the 64-channel input once.  Eight waves of two patch rows each, ONE weight fragment (output
// channels 0-15, rows past Cout zero) for all 18 K steps in 72 registers; lane (pixel, q) holds channels 4 q .. 4 q + 3 and lanes
// q < 2 store 8 bytes: the output row is 8 halfs wide (ldo = 8), channels Cout .. 7 are written as zeros.  The GEMM-form tile ran
// this layer at 17 TFLOP/s (258 us for five 512 x 512 frames): 64 of its 64 tile columns but three are padding.
template <int ACT>
__global__ __launch_bounds__(512) void conv_c64_thin_kernel(const ConvParams p) {
  VSD_CUT(VSD_CUT_CONV_HALO, p.cut)
  prefetch_kernargs();
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * C64_A_HALFS * 2];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  constexpr int OOB = (int)0x80000000;
  constexpr int NW = C64_NW, AI = C64_AI, HS = C64_HS, PW = C64_PW, PH = C64_PH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int fp = lane & 15, fq = lane >> 4;
  half8 wreg[18];
  {
    const half8 z8 = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    const half_t* wr = p.w + (size_t)(fp < p.N ? fp : 0) * p.Kp + 8 * fq;
#pragma unroll
    for (int s = 0; s < 18; ++s) {
      const half8 v = *reinterpret_cast<const half8*>(wr + 32 * s);
      wreg[s] = fp < p.N ? v : z8;
    }
  }
  const int ch0 = 4 * fq;  // this lane's four output channels
  float brv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    brv[e] = 0.f;
    if (p.bias && ch0 + e < p.N) brv[e] += (float)p.bias[ch0 + e];
    if (p.rowvec && ch0 + e < p.N) brv[e] += (float)p.rowvec[ch0 + e];
  }
  const int alc = ((lane & 7) ^ (lane >> 3)) << 4;
  const int hxl = lane >> 3;
  const int anr0 = (int)((size_t)p.batch * p.img_in * p.c0 * 2);
  const int ppr = (p.wo + PW - 1) / PW, tpi = ((p.ho + PH - 1) / PH) * ppr;
#define C64_COORDS(T_, IMG_, Y0_, X0_)                         \
  const int IMG_ = fdiv((T_), p.fd_tpi);                       \
  const int trem_##IMG_ = (T_) - IMG_ * tpi;                   \
  const int prow_##IMG_ = fdiv(trem_##IMG_, p.fd_ppr);         \
  const int Y0_ = prow_##IMG_ * PH, X0_ = (trem_##IMG_ - prow_##IMG_ * ppr) * PW;
#define C64_ISSUE(IMG_, Y0_, X0_, BUF_)                                                                              \
  {                                                                                                                  \
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.src0, 0, anr0, 0x00020000);        \
    half_t* dst_ = reinterpret_cast<half_t*>(smem) + (BUF_) * C64_A_HALFS;                                           \
    const int pix0_ = (IMG_) * p.img_in;                                                                             \
    _Pragma("unroll") for (int q = 0; q < AI; ++q) {                                                                 \
      const int j = wave_s + NW * q;                                                                                 \
      const int hy = j / 3, hx = 8 * (j - 3 * hy) + hxl;                                                             \
      const int y = (Y0_) - 1 + hy, x = (X0_) - 1 + hx;                                                              \
      const bool in = (unsigned)y < (unsigned)p.hi && (unsigned)x < (unsigned)p.wi && hx < PW + 2 && hy < PH + 2;    \
      const int sy = (int)(((unsigned)y * p.rmul_y) >> p.rshift), sx = (int)(((unsigned)x * p.rmul_x) >> p.rshift);  \
      const int vo_ = in ? (pix0_ + sy * p.ws + sx) * (BK * 2) + alc : OOB;                                          \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(dst_ + 8 * j * BK), 16, vo_, 0, 0, 0);               \
    }                                                                                                                \
  }
  int fb[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) fb[kx] = ((2 * wave_s) * HS + fp + kx) * (BK * 2) + ((fq ^ ((fp + kx) & 7)) << 4);
  const int ntiles = p.tiles_m;
  int t = blockIdx.x;
  if (t < ntiles) {
    C64_COORDS(t, img, y0, x0)
    C64_ISSUE(img, y0, x0, 0)
  }
  const int onr = (int)((size_t)p.M * p.ldo * 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int buf = 0;
  for (; t < ntiles; t += gridDim.x, buf ^= 1) {
    C64_COORDS(t, img, y0, x0)
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // (all but the previous patch's two output stores: see conv_c64_kernel)
    __builtin_amdgcn_s_barrier();
    const int tn = t + gridDim.x;
    if (tn < ntiles) {
      C64_COORDS(tn, img2, y2, x2)
      C64_ISSUE(img2, y2, x2, buf ^ 1)
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[2];
    acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned char* a = smem + buf * (C64_A_HALFS * 2);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const half8 xf = *reinterpret_cast<const half8*>(a + (ks ? fb[kx] ^ 64 : fb[kx]) + (i + ky) * (HS * BK * 2));
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[2 * tap + ks], xf, acc[i], 0, 0, 0);
        }
    }
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, onr, 0x00020000);
    const int ox = x0 + fp;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oy = y0 + 2 * wave_s + i;
      const bool ok = oy < p.ho && ox < p.wo && fq < 2;
      const int mrow = img * p.hw_out + oy * p.wo + ox;
      half4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = acc[i][e] + brv[e];
        if (ACT == 1) x = fmaxf(x, 0.f);
        if (ACT == 2) x = silu_f(x);
        o[e] = (half_t)x;
      }
      __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2*>(&o), ors, ok ? (mrow * p.ldo + ch0) * 2 : OOB, 0, 0);
    }
  }
#undef C64_COORDS
#undef C64_ISSUE
}

}  // namespace

template <int ACT>
static void c64_go(const ConvParams& p, int grid, hipStream_t s) {
  if (p.residual) hipLaunchKernelGGL((conv_c64_kernel<ACT, true>), dim3(grid), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((conv_c64_kernel<ACT, false>), dim3(grid), dim3(512), 0, s, p);
}
void vsd_launch_conv_c64(const ConvParams& p, int grid, hipStream_t s) {
  const int act = p.act & 0xff;
  if (p.N <= 8) {  // (host checks: ldo == 8, no residual, no activation after one)
    if (act == VSD_ACT_RELU) hipLaunchKernelGGL((conv_c64_thin_kernel<1>), dim3(grid), dim3(512), 0, s, p);
    else if (act == VSD_ACT_SILU) hipLaunchKernelGGL((conv_c64_thin_kernel<2>), dim3(grid), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv_c64_thin_kernel<0>), dim3(grid), dim3(512), 0, s, p);
    return;
  }
  if (act == VSD_ACT_RELU && (p.act & VSD_ACT_POST)) c64_go<3>(p, grid, s);
  else if (act == VSD_ACT_RELU) c64_go<1>(p, grid, s);
  else if (act == VSD_ACT_SILU) c64_go<2>(p, grid, s);
  else c64_go<0>(p, grid, s);
}

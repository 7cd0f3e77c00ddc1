// Weight-streaming conv / linear for the small-image levels (pipeline 9): M = batch * H * W <= 192 output pixels
// (8 x 8 x 3 frames, 12 x 12 x 1, ...), deep K (Cin >= 256, 3x3 or 1x1, stride 1).
//
// At these sizes the layer is its WEIGHTS (1280 x 11520 fp16 = 29.5 MB against 0.5 MB of activations): the job is to
// pull every weight byte across the fabric exactly once, with as many bytes in flight as the chip can hold, and to use
// each byte against all M rows while it sits in a register.  The tiled kernels cannot do that: with 64/128-row tiles the
// weight panel is re-read once per M tile, a workgroup keeps 3 ring slots (24-48 KB) in flight and the B fragments
// make an LDS round trip (measured 1.5 TB/s of weights on 192 x 1280 x 11520).  Here:
//   * grid = (N / 64) x (Cin / 128): a workgroup owns 64 output channels x ALL M rows x a 128-channel slice of the input
//     (x 9 taps); its four waves split that slice 32 channels each, so no two waves of the chip ever load the same weight;
//   * weights are stored fragment-major (packing.pack_mfma_frag): a wave-instruction loads one contiguous KB straight into
//     the B operand registers of v_mfma_f32_16x16x32_f16.  A wave issues ALL its loads (taps x 4 fragments = 36 KB for a
//     3x3 layer) before it waits for anything: ~144 KB in flight per CU, the whole layer in flight chip-wide;
//   * the activations of the slice -- the zero-padded (H+2) x (W+2) patch of every image, 128 channels -- are staged in LDS
//     once (<= 80 KB); a tap is an address offset into that panel (as in conv_halo.hip).  A-fragment reads are 16-byte,
//     granule-swizzled by the pixel index; each is used for four MFMAs (64 columns);
//   * the four waves' accumulators are added in a fixed order through LDS (the panel is dead by then), every wave
//     finishing a quarter of the tile, and written as the fp32 slab of this channel slice; splitk_reduce_kernel
//     (conv_halo.hip) adds the slabs in slice order and applies the layer's epilogue.  Deterministic.
// Bound: HBM / fabric weight streaming (algorithmic bytes = N x K x 2 per launch); the MFMA work under it is
// M_padded x N x K x 2 / 2.5 PFLOP/s (2.3 us for 192 x 1280 x 11520 against 3.7 us of weights at 8 TB/s).
//
// MEASURED on MI355X (scripts/skinny_bench.py, skinny_probe.py; round 2) -- and why this form is NOT a tuner default:
//   192 x 1280 x 11520 (3 frames, 3x3): 15.4 us + 5-7 us reducer against 17 + 5.4 us for the table's 64x128 tile with
//   split-K 8 (1.2-1.3 TB/s of weights either way); 64 x 1280 x 11520 (one frame): 15.2 against 16.9 us; the linear
//   layers of the level (K = 1280..5120) are 1.5-3x SLOWER here (Cin / 128 slabs of fp32 partials: 39 MB at K = 5120).
//   Shader-clock stamps of one workgroup at M = 192 (cycles): issue panel + weight loads 10.0k | park the panel 2.6k |
//   taps x MFMA 10.2k (432 MFMAs = 6.9k of it) | LDS reduction + slab stores 7.6k.  The load phase is not issue-bound: the
//   vector-memory queue fills and the wave stalls until earlier loads return, i.e. a CU pulls its 147 KB of weights +
//   48 KB of panel at ~30-40 GB/s however many loads the wave has queued (the per-CU miss capacity over a ~1.5 us fabric
//   round trip).  256 CUs x 30 GB/s is the ~6-7 TB/s the whole chip can stream, but only while EVERY CU streams for the
//   whole kernel; here each CU streams for a third of it, computes, then spends as long again adding four 48 KB
//   accumulator sets through LDS (192 KB of LDS traffic per round at 128 B/clk).  So "1.5 TB/s" on these layers is the
//   per-CU fill latency times the serial phases of a short kernel, not something more bytes in flight per wave fixes.
//   The form also takes a whole CU (96 KB LDS, 512 registers per lane), so the second launch in flight cannot share it.
//   Kept (explicit pipeline = 9, VSD_TUNE_STREAMING=1 adds it to the tuner's candidates) with its parity tests.
#include "conv_kernels.h"

namespace {

#ifdef SK_PROBE  // development: shader-clock stamps of wave 0 of the first and the last workgroup, behind the slabs
#define SKP(I_) { if (lane == 0 && wave == 0) stamp[I_] = __builtin_readcyclecounter(); }
#else
#define SKP(I_)
#endif

constexpr int SK_CS = 128;        // input channels per workgroup (4 waves x 32)
constexpr int SK_MAX_PIX = 320;   // padded pixels of all images the LDS panel can hold (x 256 bytes = 80 KB)

template <int MT>
struct SkinnyShape {
  static constexpr int ROUNDS = MT > 4 ? 2 : 1;          // halves of the accumulator tile reduced per LDS round
  static constexpr int TPR = MT * 4 / ROUNDS;            // 16x16 tiles per wave and round
  static constexpr int RED_BYTES = 4 * TPR * 1024;       // [wave][tile][lane][4 floats]
  static constexpr int PANEL_BYTES = SK_MAX_PIX * SK_CS * 2;
  static constexpr int LDS_BYTES = RED_BYTES > PANEL_BYTES ? RED_BYTES : PANEL_BYTES;
};

// MT: 16-row fragments covering M (4 / 8 / 12).  NT: taps (9 for a 3x3 layer, 1 for a linear one).
template <int MT, int NT>
__global__ __launch_bounds__(256) void conv_skinny_kernel(const ConvParams p, const half_t* __restrict__ wfrag) {
  using S = SkinnyShape<MT>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[S::LDS_BYTES];
  half_t* panel = reinterpret_cast<half_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = p.N >> 6;
  const int nt = (int)blockIdx.x % tiles_n, slice = (int)blockIdx.x / tiles_n;
  const int cbase = slice * SK_CS;  // first input channel of this workgroup (counted across the two concat sources)
  const half_t* src = p.src0;
  int csrc = p.c0, coff = cbase;
  if (cbase >= p.c0) {
    src = p.src1;
    csrc = p.c1;
    coff = cbase - p.c0;
  }
  const int H = p.hs, W = p.ws, pad = p.ksize >> 1;
  const int PW = W + 2 * pad, PPI = PW * (H + 2 * pad);  // padded row length / padded pixels per image
  const int P = p.batch * PPI;

#ifdef SK_PROBE
  long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  SKP(0)
  // ---- 1. the activation panel: every (padded pixel, 16-byte granule) once; loads first, all in flight
  constexpr int NFILL = SK_MAX_PIX * 16 / 256;  // 20
  half8 fill[NFILL];
  unsigned inside = 0;  // bit i: slot i is a pixel of the image (the others -- the zero border -- load a valid dummy address:
                        // no branch around a load, the compiler would wait for it inside the branch)
  const float inv_ppi = 1.0f / (float)PPI, inv_pw = 1.0f / (float)PW;
#pragma unroll
  for (int i = 0; i < NFILL; ++i) {
    const int q = tid + i * 256;
    const int pix = q >> 4, g = q & 15;
    const int b = (int)(((float)pix + 0.5f) * inv_ppi);
    const int rem = pix - b * PPI;
    const int py = (int)(((float)rem + 0.5f) * inv_pw);
    const int y = py - pad, x = rem - py * PW - pad;
    const bool ok = pix < P && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    inside |= ok ? (1u << i) : 0u;
    const size_t off = ok ? ((size_t)(b * H + y) * W + x) * csrc + coff + g * 8 : (size_t)coff;
    fill[i] = *reinterpret_cast<const half8*>(src + off);
  }

  // ---- 2. this wave's weights: channels [cbase + 32 wave, +32) of every tap, 4 fragments of 16 output channels
  half8 bw[NT][4];
  {
    const int kblocks = p.K >> 5;
    const half_t* wl = wfrag + (size_t)lane * 8;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int kb = (t * p.cin + cbase + 32 * wave) >> 5;
#pragma unroll
      for (int j = 0; j < 4; ++j) bw[t][j] = *reinterpret_cast<const half8*>(wl + ((size_t)(nt * 4 + j) * kblocks + kb) * 512);
    }
  }

  SKP(1)
  // ---- 3. park the panel (granule g of pixel pix at slot g ^ (pix & 7): the 16 rows of an A fragment -- two runs of
  // consecutive pixels -- then spread over all banks)
#pragma unroll
  for (int i = 0; i < NFILL; ++i) {
    const int q = tid + i * 256;
    const int pix = q >> 4, g = q & 15;
    const half8 v = (inside >> i) & 1u ? fill[i] : (half8){0, 0, 0, 0, 0, 0, 0, 0};
    if (pix < P) *reinterpret_cast<half8*>(panel + ((size_t)pix * 16 + (g ^ (pix & 7))) * 8) = v;
  }

  // rows of this lane's A fragments: m = 16 mi + (lane & 15) -> top-left pixel of its window in the padded panel
  int pb[MT];
  {
    const float inv_hw = 1.0f / (float)p.hw_out, inv_w = 1.0f / (float)W;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      int m = mi * 16 + (lane & 15);
      m = m < p.M ? m : p.M - 1;  // (rows past M compute on a valid pixel and are never stored)
      const int b = (int)(((float)m + 0.5f) * inv_hw);
      const int r = m - b * p.hw_out;
      const int y = (int)(((float)r + 0.5f) * inv_w);
      pb[mi] = b * PPI + y * PW + (r - y * W);
    }
  }
  f32x4 acc[MT][4];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[mi][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  SKP(2)
  __syncthreads();
  SKP(3)

  // ---- 4. taps x rows x 4 fragments.  A fragments in groups of four rows-of-16, read THREE groups (768 MFMA cycles) ahead
  // of their MFMAs through a ring of four register groups: left to itself the compiler keeps one fragment in flight and
  // every four MFMAs then wait out an LDS round trip (measured: the kernel took 16 us at M = 192 and 8 us at M = 64 for
  // the same 29.5 MB of weights).
  const int gq = wave * 4 + (lane >> 4);  // this lane's granule of the slice: its wave's 32 channels, 8 per quarter-wave
  constexpr int GP = MT / 4, STEPS = NT * GP, LEAD = 3;
  half8 ring[4][4];
  auto fetch = [&](int step) __attribute__((always_inline)) {
    const int t = step / GP, gi = step % GP;
    const int toff = NT == 1 ? 0 : (t / 3) * PW + (t % 3);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pix = pb[gi * 4 + k] + toff;
      ring[step & 3][k] = *reinterpret_cast<const half8*>(panel + ((size_t)pix * 16 + (gq ^ (pix & 7))) * 8);
    }
  };
#pragma unroll
  for (int st = 0; st < LEAD && st < STEPS; ++st) fetch(st);
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    if (st + LEAD < STEPS) fetch(st + LEAD);
    __builtin_amdgcn_sched_barrier(0);
    const int t = st / GP, gi = st % GP;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[gi * 4 + k][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[st & 3][k], bw[t][j], acc[gi * 4 + k][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }

  SKP(4)
  // ---- 5. add the four waves' tiles in wave order through LDS; each wave finishes a quarter and stores it to the slab
  float* red = reinterpret_cast<float*>(smem);
  float* slab = p.ws_partial + (size_t)slice * p.M * p.N;
#pragma unroll
  for (int rd = 0; rd < S::ROUNDS; ++rd) {
    __syncthreads();  // the panel (round 0) / the previous round's sums have been read
#pragma unroll
    for (int tl = 0; tl < S::TPR; ++tl) {
      const int mi = rd * (MT / S::ROUNDS) + tl / 4, j = tl & 3;
      *reinterpret_cast<f32x4*>(red + ((size_t)(wave * S::TPR + tl) * 64 + lane) * 4) = acc[mi][j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < S::TPR / 4; ++i) {
      const int tl = wave * (S::TPR / 4) + i;
      f32x4 v = *reinterpret_cast<const f32x4*>(red + ((size_t)tl * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(red + ((size_t)(w * S::TPR + tl) * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += u[e];
      }
      const int mi = rd * (MT / S::ROUNDS) + tl / 4, j = tl & 3;
      const int row0 = mi * 16 + 4 * (lane >> 4), col = nt * 64 + j * 16 + (lane & 15);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (row0 + e < p.M) slab[(size_t)(row0 + e) * p.N + col] = v[e];
    }
  }
#ifdef SK_PROBE
  SKP(5)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SKP(6)
  if (lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
    long long* dst = reinterpret_cast<long long*>(p.ws_partial + (size_t)p.split_k * p.M * p.N) + (blockIdx.x == 0 ? 0 : 8);
    for (int i = 0; i < 7; ++i) dst[i] = stamp[i];
  }
#endif
}

template <int MT>
void launch_mt(const ConvParams& p, const half_t* wfrag, int grid, hipStream_t s) {
  if (p.ksize == 3) hipLaunchKernelGGL((conv_skinny_kernel<MT, 9>), dim3(grid), dim3(256), 0, s, p, wfrag);
  else hipLaunchKernelGGL((conv_skinny_kernel<MT, 1>), dim3(grid), dim3(256), 0, s, p, wfrag);
}

}  // namespace

int vsd_conv_skinny_max_pixels() { return SK_MAX_PIX; }

// grid = (N / 64) * (Cin / 128); the caller has checked eligibility and runs splitk_reduce_kernel afterwards
void vsd_launch_conv_skinny(const ConvParams& p, const half_t* wfrag, int grid, hipStream_t s) {
  if (p.M <= 64) launch_mt<4>(p, wfrag, grid, s);
  else if (p.M <= 128) launch_mt<8>(p, wfrag, grid, s);
  else launch_mt<12>(p, wfrag, grid, s);
}

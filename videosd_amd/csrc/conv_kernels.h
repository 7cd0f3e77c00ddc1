// (shared by the conv_*.hip translation units: the kernel templates are compiled where they are instantiated, one tile
// family per file, so that the files build in parallel)
//
// Implicit-GEMM convolution / linear layer on MFMA (gfx950, wave64, v_mfma_f32_16x16x32_f16).
//
//   out[m][n] = epilogue( sum_k A[m][k] * W[n][k] )
//
// A is never materialised: each 64-wide K tile is gathered straight from the NHWC source(s)
// (3x3 taps, stride, zero padding, nearest resize and channel concat are all address arithmetic in
// the tile loader).  Tiles are staged global -> registers -> LDS (XOR-swizzled 128-B rows, conflict
// free for ds_read_b128 fragment reads), double buffered with one barrier per K tile; the next
// tile's global loads are issued before the current tile's MFMAs (issue-early / write-late).
// The accumulator tile is transposed through LDS so that bias / residual reads and the output
// stores are 16-byte row-contiguous.  K can be split across workgroups (fp32 slabs + a reduce
// kernel that applies the same epilogue) for the small-M, huge-K layers of the 16x16 / 8x8 levels.
//
// Algorithmic work per launch: 2*M*N*K FLOP, fp16 bytes: N*Kp (weights) + M*Cin (input) + M*N (output).
#pragma once
#include <stdarg.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

constexpr int BK = 64;  // halfs per K tile (128-byte LDS rows)

struct ConvParams {
  const half_t* src0;
  const half_t* src1;
  int c0, c1, cin;
  int hs, ws, hi, wi, ho, wo;
  int ksize, stride, pad;
  int resize;   // hi != hs || wi != ws
  unsigned rmul_y, rmul_x;  // resize: ceil(hs * 2^22 / hi), source row = (iy * rmul_y) >> 22; else 1 and shift 0
  int rshift;
  int generic;  // cin % 64 != 0: per-chunk tap computation
  int fast;     // buffer-load address path usable: !generic, no resize, every operand < 2 GB
  int pointwise;  // ksize 1, stride 1, no padding, no resize: output row m reads source pixel m (no per-row index arithmetic)
  int halo_ok;  // the same without the "no resize" condition (the halo kernel folds the nearest resize into its patch fetch)
  const half_t* w;
  int M, N, K, Kp;
  const half_t* bias;
  const half_t* rowvec;
  const half_t* residual;
  const half_t* residual2;
  int ldr;
  float out_scale;
  const float* out_scale_dev;  // optional: the scale lives in device memory (read at run time; replaces out_scale)
  int act;
  half_t* out;
  int ldo;
  half_t* out2;
  const half_t* add2;
  half_t* out_t;
  int ldt, t_col0;
  int split_k, kt_per_split;
  float* ws_partial;
  float* rowstat_out;    // [M][N/64][2]: per-row (sum, sumsq) of the fp16 outputs over each 64-column group
  float* chanstat_part;  // [tiles_m][N][2] scratch: per-tile column (sum, sumsq) of the fp16 outputs
  float* chanstat_out;   // [N][2]: per-channel (sum, sumsq) over all M rows -- the next GroupNorm's statistics
  int* chan_counters;    // [tiles_n] arrival tickets (all zero between launches)
  const float* ln_part;  // fused input LayerNorm: row partials of the A operand, [M][ln_groups][2]
  int ln_groups;
  float ln_eps;
  const float* ln_s;     // [N] sum_k W'[n][k]  (W' = W * gamma)
  const float* ln_t;     // [N] sum_k beta[k] W[n][k] + bias[n]
  const half_t* zeros;  // >= 16 zero bytes: source of out-of-bounds chunks for the direct-to-LDS loader
  int* counters;  // per-tile arrival tickets for the in-kernel split-K reduction (all zero between launches)
  int tiles_m, tiles_n;
  VSD_CUT_FIELD
  int order;    // block_to_tile: 0 = workgroups sharing a weight tile share an XCD, 1 = workgroups sharing input rows do,
                // 2 / 3 = the XCDs as a 2 x 4 / 4 x 2 grid over (M tiles, weight-tile groups)
  int gx;       // orders 2 / 3: weight-tile groups per XCD
#ifdef VSD_CONV_PROBE
  long long* probe;  // scripts/conv_probe.cpp: per-section shader-clock totals of wave 0 of workgroup 0
#endif
#ifdef VSD_WG_TIMELINE
  unsigned long long* wgtl;  // scripts/wg_timeline.py: per workgroup {start, main loop done, end} in 10 ns ticks + hardware id
#endif
  int softmax_cols;  // VSD_ACT_SOFTMAX: valid columns of every 128-column group
  int batch;    // images stacked along M: M = batch * ho * wo, image b's source pixels start at b * hs * ws
  int hw_out;   // ho * wo
  int img_in;   // hs * ws
  int t_img;    // transposed output: columns per image (image b's rows m land at b * t_img + (m - b * hw_out))
  // launch constants as multipliers (common.h fdiv): block -> tile (span = 8 S, S, tiles_n: see block_to_tile), output row ->
  // (image, y, x) (hw_out, wo), halo patch -> (image, patch row / column) (tiles per image, patches per row)
  FastDiv fd_span, fd_s, fd_tiles_n, fd_hw_out, fd_wo, fd_tpi, fd_ppr, fd_gx;
};

#ifdef VSD_CONV_PROBE
inline long long* g_conv_probe = nullptr;
#define CPROBE(I_)                                         \
  {                                                        \
    long long t_ = __builtin_readcyclecounter();           \
    pacc[I_] += t_ - plast;                                \
    plast = t_;                                            \
  }
#define CPROBE_OUT()                                                                       \
  if (p.probe && blockIdx.x == 0 && threadIdx.x == 0)                                      \
    for (int i_ = 0; i_ < 8; ++i_) p.probe[i_] = pacc[i_];                                 \
  if (p.probe && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) /* a workgroup of the LAST round: warm caches */ \
    for (int i_ = 0; i_ < 8; ++i_) p.probe[8 + i_] = pacc[i_];
#else
#define CPROBE(I_)
#define CPROBE_OUT()
#endif



// launchers of the tile families (one translation unit each; conv_gemm.hip dispatches)
void vsd_launch_conv_128x128(const ConvParams& p, int grid, int stages, hipStream_t s);
void vsd_launch_conv_128x64(const ConvParams& p, int grid, int stages, hipStream_t s);
void vsd_launch_conv_64x64(const ConvParams& p, int grid, int stages, hipStream_t s);
void vsd_launch_conv_64x128(const ConvParams& p, int grid, int stages, hipStream_t s);
void vsd_launch_conv_256x128(const ConvParams& p, int grid, int stages, hipStream_t s);
void vsd_launch_conv_halo(const ConvParams& p, int bm, int bn, int grid, hipStream_t s);
void vsd_launch_splitk_reduce(const ConvParams& p, int grid, hipStream_t s);

namespace {

// ---------------------------------------------------------------- epilogue (shared with the reducer)
// has_res / res_val, has_brv / brv_lo, brv_hi: the residual chunk / bias + rowvec of these 8 columns when the caller loaded
// them BEFORE its first store (vmcnt counts stores too: a load issued after a store is only waited for once that store has
// retired).  Passed BY VALUE: the earlier form took `flag ? &array[j] : nullptr`, and that conditional address-of kept the
// caller's arrays in scratch memory (112 bytes per lane in every instantiation: the "prefetched" residual was stored to
// scratch as soon as it was loaded -- a full memory latency exposed in front of the accumulator transpose -- and read back
// from scratch inside the walk; found with scripts/wg_timeline.py: half of a short-K workgroup's life was its epilogue).
__device__ __forceinline__ void epilogue_store8(const ConvParams& p, int m, int n, float (&v)[8], float& rsum, float& rsq,
                                                const bool has_res, const half8 res_val, const bool has_brv, const f32x4 brv_lo,
                                                const f32x4 brv_hi) {
  // n is a multiple of 8; handles n + 8 > N by scalar fallback
  const bool full = (n + 8 <= p.N);
  if (has_brv) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] += brv_lo[i];
      v[4 + i] += brv_hi[i];
    }
  } else {
  if (p.bias) {
    if (full) {
      half8 b = *reinterpret_cast<const half8*>(p.bias + n);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += (float)b[i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (n + i < p.N) v[i] += (float)p.bias[n + i];
    }
  }
  if (p.rowvec) {
    if (full) {
      half8 b = *reinterpret_cast<const half8*>(p.rowvec + n);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += (float)b[i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (n + i < p.N) v[i] += (float)p.rowvec[n + i];
    }
  }
  }
  const int act = p.act & 0xff;
  const bool post = (p.act & VSD_ACT_POST) != 0;
  auto apply_act = [&](float x) -> float {
    if (act == VSD_ACT_RELU) return fmaxf(x, 0.0f);
    if (act == VSD_ACT_SILU) return silu_f(x);
    if (act == VSD_ACT_QUICKGELU) return quick_gelu_f(x);
    if (act == VSD_ACT_GELU) return gelu_erf_f(x);
    return x;
  };
  if (act != VSD_ACT_NONE && !post) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = apply_act(v[i]);
  }
  if (p.out_scale_dev) {
    const float sc = *p.out_scale_dev;  // uniform address: a scalar load
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= sc;
  } else if (p.out_scale != 1.0f) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= p.out_scale;
  }
  if (p.out_t && n >= p.t_col0) {
    int col = m;
    if (p.batch > 1) {
      const int b = m / p.hw_out;
      col = b * p.t_img + (m - b * p.hw_out);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (n + i < p.N) p.out_t[(size_t)(n + i - p.t_col0) * p.ldt + col] = (half_t)v[i];
    return;
  }
  if (full) {
    if (p.residual) {
      half8 r = res_val;
      if (!has_res) r = *reinterpret_cast<const half8*>(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += (float)r[i];
    }
    if (p.residual2) {
      half8 r = *reinterpret_cast<const half8*>(p.residual2 + (size_t)m * p.ldr + n);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += (float)r[i];
    }
    if (post) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = apply_act(v[i]);
    }
    half8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      o[i] = (half_t)v[i];
      float f = (float)o[i];
      v[i] = f;  // hand the rounded value back: the fused statistics are those of the stored tensor
      rsum += f;
      rsq += f * f;
    }
    *reinterpret_cast<half8*>(p.out + (size_t)m * p.ldo + n) = o;
    if (p.out2) {
      half8 a = *reinterpret_cast<const half8*>(p.add2 + (size_t)m * p.ldo + n);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (half_t)(v[i] + (float)a[i]);
      *reinterpret_cast<half8*>(p.out2 + (size_t)m * p.ldo + n) = o;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (n + i >= p.N) continue;
      float x = v[i];
      if (p.residual) x += (float)p.residual[(size_t)m * p.ldr + n + i];
      if (p.residual2) x += (float)p.residual2[(size_t)m * p.ldr + n + i];
      if (post) x = apply_act(x);
      p.out[(size_t)m * p.ldo + n + i] = (half_t)x;
      if (p.out2) p.out2[(size_t)m * p.ldo + n + i] = (half_t)(x + (float)p.add2[(size_t)m * p.ldo + n + i]);
    }
  }
}

__device__ __forceinline__ void epilogue_store8(const ConvParams& p, int m, int n, float (&v)[8], float& rsum, float& rsq) {
  const half8 z8 = (half8){0, 0, 0, 0, 0, 0, 0, 0};
  const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  epilogue_store8(p, m, n, v, rsum, rsq, false, z8, false, z4, z4);
}

// Fused input LayerNorm: the GEMM ran on the raw rows x with W' = W*gamma, so
//   LN(x) W^T + b = rstd * (x W'^T - mean * s) + t,   s[n] = sum_k W'[n][k],  t[n] = sum_k beta[k] W[n][k] + b[n].
// Row mean / rstd come from the (sum, sumsq) partials the producing kernel's epilogue left per 64-column group.
__device__ __forceinline__ void ln_row_stats(const ConvParams& p, int m, float& mean, float& rstd) {
  // partials of one row are contiguous: [M][ln_groups][2].  ALL of a row's loads are issued before the first is consumed
  // (one memory round trip: the first form fetched them four at a time -- five dependent round trips for a 1280-wide row,
  // 3-4 us in front of the main loop of every LayerNorm-consuming GEMM by scripts/wg_timeline.py); the sums run in
  // group order either way (bit-identical).
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2* src = reinterpret_cast<const f32x2*>(p.ln_part) + (size_t)m * p.ln_groups;
  float S = 0.f, Q = 0.f;
  const int G = p.ln_groups;
  auto batch = [&](auto n_tag) __attribute__((always_inline)) {
    constexpr int NG = decltype(n_tag)::value;
    f32x2 a[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) a[g] = src[g < G ? g : 0];  // (clamped: the surplus loads hit the row's first line)
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (g < G) {
        S += a[g][0];
        Q += a[g][1];
      }
  };
  if (G <= 5) batch(std::integral_constant<int, 5>{});
  else if (G <= 10) batch(std::integral_constant<int, 10>{});
  else if (G <= 20) batch(std::integral_constant<int, 20>{});
  else {
    for (int g = 0; g < G; ++g) {
      f32x2 a = src[g];
      S += a[0];
      Q += a[1];
    }
  }
  const float inv = 1.0f / (float)p.K;
  mean = S * inv;
  rstd = rsqrtf(fmaxf(Q * inv - mean * mean, 0.f) + p.ln_eps);
}
__device__ __forceinline__ void ln_transform8(const ConvParams& p, int n, float mean, float rstd, float (&v)[8]) {
  if (n + 8 <= p.N) {
    f32x4 s0 = *reinterpret_cast<const f32x4*>(p.ln_s + n), s1 = *reinterpret_cast<const f32x4*>(p.ln_s + n + 4);
    f32x4 t0 = *reinterpret_cast<const f32x4*>(p.ln_t + n), t1 = *reinterpret_cast<const f32x4*>(p.ln_t + n + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = rstd * (v[i] - mean * s0[i]) + t0[i];
      v[4 + i] = rstd * (v[4 + i] - mean * s1[i]) + t1[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (n + i < p.N) v[i] = rstd * (v[i] - mean * p.ln_s[n + i]) + p.ln_t[n + i];
  }
}

// 8 consecutive fp32 outputs of row r / column c8 of this tile: from the LDS-staged accumulators, or (last
// arriver of a split-K tile) the sum of all slabs in the fixed order 0..split_k-1 (deterministic).
__device__ __forceinline__ void load_chunk8(const ConvParams& p, const float* Cs, int pitch, bool from_slabs, int nparts, int r,
                                            int c8, int m, int n, float (&v)[8]) {
  if (!from_slabs) {
    f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + r * pitch + c8);
    f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + r * pitch + c8 + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = lo[i];
      v[4 + i] = hi[i];
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.f;
  const size_t slab = (size_t)p.M * p.N;
  const float* s = p.ws_partial + (size_t)m * p.N + n;
  if (n + 8 <= p.N) {
    // Up to 8 slabs per round trip (as splitk_reduce_kernel): the first form's runtime-bounded loop was a chain of `nparts`
    // dependent memory round trips in the LAST workgroup of every tile -- a 12-way split paid ~12 fabric latencies behind its
    // ticket, which is why the tuner kept the two-launch form at the deep levels.  Same order of additions: same bits.
    constexpr int NB = 8;
    for (int k0 = 0; k0 < nparts; k0 += NB) {
      f32x4 lo[NB], hi[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int kk = k0 + k < nparts ? k0 + k : 0;  // (clamped: surplus loads hit slab 0's line again)
        lo[k] = *reinterpret_cast<const f32x4*>(s + kk * slab);
        hi[k] = *reinterpret_cast<const f32x4*>(s + kk * slab + 4);
      }
#pragma unroll
      for (int k = 0; k < NB; ++k)
        if (k0 + k < nparts) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] += lo[k][i];
            v[4 + i] += hi[k][i];
          }
        }
    }
  } else {
    for (int k = 0; k < nparts; ++k) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (n + i < p.N) v[i] += s[k * slab + i];
    }
  }
}

// The argument block (ConvParams, ~330 bytes = 6 cache lines) is read with scalar loads wherever a field is first
// used: the compiler spreads those loads over the prologue and every first touch of a new 64-byte line is a miss of its
// own (ISA of round 1: 8 s_load / s_waitcnt round trips before the first barrier, ~1k of the prologue's 3.7k cycles).
// Touching one dword of every line at the very top brings all lines in together; the later loads hit the scalar cache.
__device__ __forceinline__ void prefetch_kernargs() {
  typedef const __attribute__((address_space(4))) unsigned* kptr_t;
  kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr int NL = (sizeof(ConvParams) + 63) / 64;
  unsigned t[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) t[i] = ka[i * 16];
#pragma unroll
  for (int i = 0; i < NL; ++i) asm volatile("" ::"s"(t[i]));
}

// ---------------------------------------------------------------- block -> tile (XCD-aware)
// Workgroups b and b+8 share an XCD (round-robin dispatch) and each XCD has its own L2, so the order decides how often an
// operand crosses the fabric.  order 0: the tiles_m workgroups that stream the SAME weight tile (same tile_n / split) get
// ids b, b+8, b+16, ...: a weight tile is fetched into ONE L2, the activation rows into every L2 whose workgroups need
// them (up to 8) -- right when the weights are the big operand (the deep levels).  order 1: the tiles_n * split_k
// workgroups that read the SAME activation rows (same tile_m) share the XCD instead: the rows cross the fabric once, the
// (small) weights up to 8 times -- right for the wide, shallow layers of the 64x64 / 32x32 levels and TAESD, where round 1's
// counters showed every N tile's XCD re-fetching the input rows (51 MB of fabric traffic per launch against 21 MB of
// operands).  orders 2 / 3: the eight XCDs form a 2 x 4 / 4 x 2 grid -- XCD (xm, xn) owns the M tiles of class xm and the
// weight-tile groups of class xn, so the rows cross the fabric 4 / 2 times and the weights 2 / 4 times: cheaper than either
// one-dimensional order when the two operands are of similar size (the 32x32 and 16x16 levels with several frames per
// launch; needs tiles_m and the group count divisible by the grid).  The host picks the cheapest per launch.  Pure speed:
// any placement gives the same result.
__device__ __forceinline__ void block_to_tile(const ConvParams& p, int bid, int& tile_m, int& grp) {
  if (p.order >= 2) {
    const int x = bid & 7, q = bid >> 3;   // XCD, sequence number inside it
    const int mc = p.order == 2 ? 2 : 4, nc = 8 / mc;
    const int xm = x & (mc - 1), xn = x >> (p.order == 2 ? 1 : 2);
    const int i = fdiv(q, p.fd_gx), j = q - i * p.gx;  // consecutive workgroups of an XCD share their M tile
    tile_m = i * mc + xm;
    grp = j * nc + xn;
    return;
  }
  const int G = p.tiles_n * p.split_k;
  const int P = p.order ? p.tiles_m : G;   // spread over the XCDs
  const int S = p.order ? G : p.tiles_m;   // share one XCD
  const int full = (P >> 3) << 3;
  int prim, sec;
  if (bid < full * S) {
    const int span = 8 * S;
    const int chunk = fdiv(bid, p.fd_span), r = bid - chunk * span;
    prim = chunk * 8 + (r & 7);
    sec = r >> 3;
  } else {
    const int rem = bid - full * S;
    const int q = fdiv(rem, p.fd_s);
    prim = full + q;
    sec = rem - q * S;
  }
  tile_m = p.order ? prim : sec;
  grp = p.order ? sec : prim;
}

// ---------------------------------------------------------------- main kernel
// STAGES == 0: register-staged double buffer (global -> VGPR -> LDS), one tile of prefetch.
// STAGES >= 3: direct-to-LDS ring (global_load_lds, 16 B per lane) with STAGES-1 tiles in flight behind counted
//              vmcnt waits and raw barriers; the XOR swizzle is applied on the per-lane SOURCE address because a
//              wave's LDS-DMA destination is lane-linear; out-of-bounds chunks read a zero page.
// FAST (direct-to-LDS ring, cin % 64 == 0, no resize): the per-tile operand addresses come almost for free.  The
//   generic issue path recomputes tap / channel / bounds / 64-bit addresses for every 16-byte chunk of every K tile
//   (~160 instructions, a dozen quarter-rate integer multiplies and a scalar division per tile -- 3x the issue time
//   of the 16 MFMAs they feed).  Here every chunk is a raw BUFFER load to LDS: the per-row byte offset of the tile
//   row's CENTRE pixel and a 9-bit "which taps are inside the image" mask are computed once; per K tile the tap /
//   channel displacement is ONE scalar (the instruction's soffset, kept by an incremental scalar cursor instead of a
//   division), an out-of-image tap turns the lane's offset into an out-of-range one (the buffer unit then writes
//   zeros into LDS: the conv's zero padding), and the weight rows need no vector instruction at all.
template <int BM, int BN, bool GENERIC, int STAGES, bool ILV, bool FAST>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvParams p) {
  VSD_CUT(VSD_CUT_CONV_GEMM, p.cut)
  prefetch_kernargs();
  WGTL_START()
  constexpr int WM = 2, WN = 2;             // 2x2 waves
  constexpr int TM = BM / WM, TN = BN / WN;  // wave tile
  constexpr int FM = TM / 16, FN = TN / 16;  // 16x16 fragments per wave
  constexpr int AR = BM / 32, BR = BN / 32;  // 16-byte chunks per thread per K tile
  constexpr int BNP = BN + 4;                // fp32 epilogue row pitch
  constexpr int NBUF = STAGES == 0 ? 2 : STAGES;
  constexpr int STAGE_HALFS = (BM + BN) * BK;
  constexpr int STAGE_BYTES = NBUF * STAGE_HALFS * 2;
  constexpr int EPI_BYTES = BM * BNP * 4;
  constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES + BM * 8];  // + per-row (mean, rstd) of a fused LN
  half_t* As = reinterpret_cast<half_t*>(smem);                                        // register path: [2][BM][64]
  half_t* Bs = reinterpret_cast<half_t*>(smem) + 2 * BM * BK;                          //                [2][BN][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // block -> (tile_m, tile_n, split).  Workgroups b and b+8 share an XCD (round-robin dispatch), so the tiles_m
  // workgroups that stream the SAME weight tile (same tile_n / split) are given ids b, b+8, b+16, ...: the tile is
  // then fetched into one XCD's L2 once instead of once per XCD.  Pure speed: any placement gives the same result.
  int tile_m, grp;
  block_to_tile(p, blockIdx.x, tile_m, grp);
  const int split = fdiv(grp, p.fd_tiles_n);
  const int tile_n = grp - split * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int KT = p.Kp / BK;
  const int kt_begin = split * p.kt_per_split;
  const int kt_end = min(KT, kt_begin + p.kt_per_split);

  // ---- loader coordinates
  const int cc = tid & 7;    // 16-byte chunk within the 128-byte tile row
  const int lr = tid >> 3;   // 0..31
  int iy0[AR], ix0[AR], ib[AR];  // ib: first source pixel of the row's image
  bool mvalid[AR];
  // (a pointwise layer on the buffer-load path needs none of this: its row m reads source pixel m -- two integer divisions
  // per row, ~0.3 us of a short-K workgroup's prologue)
  const bool pointwise = FAST && p.pointwise;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int m = m0 + lr + 32 * i;
    mvalid[i] = m < p.M;
    ib[i] = iy0[i] = ix0[i] = 0;
    if (!pointwise) {
      int mm = mvalid[i] ? m : 0;
      int b = 0;
      if (p.batch > 1) {
        b = fdiv(mm, p.fd_hw_out);
        mm -= b * p.hw_out;
      }
      ib[i] = b * p.img_in;
      int oy = fdiv(mm, p.fd_wo), ox = mm - oy * p.wo;
      iy0[i] = oy * p.stride - p.pad;
      ix0[i] = ox * p.stride - p.pad;
    }
  }
  const half_t* wrow[BR];
  bool nvalid[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    int n = n0 + lr + 32 * i;
    nvalid[i] = n < p.N;
    wrow[i] = p.w + (size_t)(nvalid[i] ? n : 0) * p.Kp + cc * 8;
  }

  u32x4 areg[AR], breg[BR];
  const u32x4 zero4 = (u32x4){0u, 0u, 0u, 0u};

// Tile loader (macro, not a lambda: keeps areg/breg in registers).  Loads are unconditional from a
// clamped, always-valid address and zeroed by a select, so there is no divergent control flow.
#define VSD_LOAD_TILE(KT_)                                                                          \
  {                                                                                                 \
    const int kt_ = (KT_);                                                                          \
    _Pragma("unroll") for (int i = 0; i < BR; ++i) {                                                \
      u32x4 v = *reinterpret_cast<const u32x4*>(wrow[i] + (size_t)kt_ * BK);                        \
      breg[i] = nvalid[i] ? v : zero4;                                                              \
    }                                                                                               \
    int k_, cs_;                                                                                    \
    const half_t* src_;                                                                             \
    bool kok_ = true;                                                                               \
    if (!GENERIC) {                                                                                 \
      k_ = kt_ * BK; /* uniform: the whole tile lies inside one tap and one source */              \
    } else {                                                                                        \
      k_ = kt_ * BK + cc * 8;                                                                       \
      kok_ = k_ < p.K;                                                                              \
    }                                                                                               \
    const int tap_ = k_ / p.cin;                                                                    \
    int c_ = k_ - tap_ * p.cin;                                                                     \
    const int ky_ = tap_ / p.ksize, kx_ = tap_ - ky_ * p.ksize;                                     \
    if (!GENERIC && c_ >= p.c0) {                                                                   \
      src_ = p.src1; cs_ = p.c1; c_ -= p.c0;                                                        \
    } else {                                                                                        \
      src_ = p.src0; cs_ = p.c0;                                                                    \
    }                                                                                               \
    if (!GENERIC) c_ += cc * 8;                                                                     \
    _Pragma("unroll") for (int i = 0; i < AR; ++i) {                                                \
      int iy = iy0[i] + ky_, ix = ix0[i] + kx_;                                                     \
      bool ok = kok_ && mvalid[i] && (unsigned)iy < (unsigned)p.hi && (unsigned)ix < (unsigned)p.wi; \
      /* nearest resize as a fixed-point multiply: floor(i*hs/hi) exactly for i*hi < 2^22 (identity: 2^22) */ \
      const int sy = (int)(((unsigned)iy * p.rmul_y) >> p.rshift), sx = (int)(((unsigned)ix * p.rmul_x) >> p.rshift); \
      size_t off = ok ? ((size_t)(ib[i] + sy * p.ws + sx)) * cs_ + c_ : 0;                                  \
      u32x4 v = *reinterpret_cast<const u32x4*>(src_ + off);                                        \
      areg[i] = ok ? v : zero4;                                                                     \
    }                                                                                               \
  }
#define VSD_STORE_TILE(BUF_)                                                                        \
  {                                                                                                 \
    half_t* a_ = As + (BUF_) * BM * BK;                                                             \
    half_t* b_ = Bs + (BUF_) * BN * BK;                                                             \
    _Pragma("unroll") for (int i = 0; i < AR; ++i) {                                                \
      int r = lr + 32 * i;                                                                          \
      *reinterpret_cast<u32x4*>(a_ + r * BK + ((cc ^ (r & 7)) << 3)) = areg[i];                     \
    }                                                                                               \
    _Pragma("unroll") for (int i = 0; i < BR; ++i) {                                                \
      int r = lr + 32 * i;                                                                          \
      *reinterpret_cast<u32x4*>(b_ + r * BK + ((cc ^ (r & 7)) << 3)) = breg[i];                     \
    }                                                                                               \
  }

#ifdef VSD_CONV_PROBE
  long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long plast = __builtin_readcyclecounter();
#endif
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float* rowms = reinterpret_cast<float*>(smem + LDS_BYTES);  // per-row (mean, rstd) of a fused input LayerNorm
// issued right after the prologue tile loads so that its memory latency overlaps theirs
#define VSD_LN_ROWSTATS()                                              \
  if (p.ln_part && tid < BM) {                                         \
    float mean = 0.f, rstd = 0.f;                                      \
    if (m0 + tid < p.M) ln_row_stats(p, m0 + tid, mean, rstd);         \
    rowms[2 * tid] = mean;                                             \
    rowms[2 * tid + 1] = rstd;                                         \
  }
  const int fr = lane & 15;  // fragment row (A) / column (B)
  const int fq = lane >> 4;  // k-chunk quarter

  if constexpr (STAGES == 0) {
    if (kt_begin < kt_end) {
      VSD_LOAD_TILE(kt_begin)
      VSD_LN_ROWSTATS()
      VSD_STORE_TILE(0)
    } else {
      VSD_LN_ROWSTATS()
    }
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const int buf = (kt - kt_begin) & 1;
      const bool more = kt + 1 < kt_end;
      if (more) VSD_LOAD_TILE(kt + 1)
      const half_t* a = As + buf * BM * BK;
      const half_t* b = Bs + buf * BN * BK;
  #pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        half8 af[FM], bf[FN];
  #pragma unroll
        for (int i = 0; i < FM; ++i) {
          int r = wm * TM + i * 16 + fr;
          af[i] = *reinterpret_cast<const half8*>(a + r * BK + (((ks * 4 + fq) ^ (r & 7)) << 3));
        }
  #pragma unroll
        for (int j = 0; j < FN; ++j) {
          int r = wn * TN + j * 16 + fr;
          bf[j] = *reinterpret_cast<const half8*>(b + r * BK + (((ks * 4 + fq) ^ (r & 7)) << 3));
        }
  #pragma unroll
        for (int i = 0; i < FM; ++i)
  #pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      if (more) VSD_STORE_TILE(buf ^ 1)
      __syncthreads();
    }
  } else {
    // ------------------------------------------------------------ direct-to-LDS ring
    constexpr int LPT = AR + BR;            // LDS-DMA instructions per thread per tile
    const int lc = cc ^ (lr & 7);           // logical 16-byte chunk this lane fetches; it lands in slot cc of its row
    const int nt = kt_end - kt_begin;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
    // ---- FAST path state (see the kernel's header comment)
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);  // scalar: the DMA's LDS base goes to M0 without a v_readfirstlane
    [[maybe_unused]] int apix[AR];          // centre pixel (oy*stride, ox*stride) of the row, in pixels from the tensor start
    [[maybe_unused]] unsigned tapmask[AR];  // bit (ky*ksize + kx): that tap of this row lies inside the image
    [[maybe_unused]] int bvoff[BR];         // weight row byte offset (+ this lane's chunk), or out of range
    [[maybe_unused]] int cur_c = 0, cur_tap = 0, cur_ky = 0, cur_kx = 0, cur_kt = kt_begin;  // scalar cursor: next tile to fetch
    // the A descriptors start (pad*ws + pad) pixels BEFORE the tensor: soffset = (ky*ws + kx)*cs*2 + c*2 is then
    // never negative; a lane only ever adds it to a centre pixel whose tap is inside the image.  (Descriptors are
    // rebuilt from these scalars per tile: a handful of SALU moves.)
    const int neg_pix = p.pad * p.ws + p.pad;
    [[maybe_unused]] const half_t* abase0 = p.src0 - (size_t)neg_pix * p.c0;
    [[maybe_unused]] const half_t* abase1 = (p.src1 ? p.src1 : p.src0) - (size_t)neg_pix * p.c1;
    [[maybe_unused]] const int anr0 = (int)(((size_t)p.batch * p.img_in + neg_pix) * p.c0 * 2);
    [[maybe_unused]] const int anr1 = (int)(((size_t)p.batch * p.img_in + neg_pix) * p.c1 * 2);
    [[maybe_unused]] const int bnr = (int)((size_t)p.N * p.Kp * 2);
    constexpr int OOB = (int)0x80000000;
    if constexpr (FAST) {
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        if (pointwise) {
          apix[i] = mvalid[i] ? m0 + lr + 32 * i : 0;
          tapmask[i] = mvalid[i] ? 1u : 0u;
          continue;
        }
        apix[i] = ib[i] + (iy0[i] + p.pad) * p.ws + (ix0[i] + p.pad);
        unsigned mk = 0;
        for (int ky = 0; ky < p.ksize; ++ky)
          for (int kx = 0; kx < p.ksize; ++kx) {
            const bool in = mvalid[i] && (unsigned)(iy0[i] + ky) < (unsigned)p.hi && (unsigned)(ix0[i] + kx) < (unsigned)p.wi;
            mk |= (in ? 1u : 0u) << (ky * p.ksize + kx);
          }
        tapmask[i] = mk;
      }
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const int n = n0 + lr + 32 * i;
        bvoff[i] = n < p.N ? n * p.Kp * 2 + lc * 16 : OOB;
      }
      if (kt_begin > 0) {  // (only a later K split starts inside the tap / channel sequence)
        const int k0 = kt_begin * BK;
        cur_tap = k0 / p.cin;
        cur_c = k0 - cur_tap * p.cin;
        cur_ky = cur_tap / p.ksize;
        cur_kx = cur_tap - cur_ky * p.ksize;
      }
    }
// fetch the cursor's tile into ring slot SLOT_, then (ADV_) move the cursor one K tile on
#define VSD_ISSUE_FAST(SLOT_, ADV_)                                                                    \
  {                                                                                                    \
    half_t* a_ = reinterpret_cast<half_t*>(smem) + (SLOT_) * STAGE_HALFS;                              \
    half_t* b_ = a_ + BM * BK;                                                                         \
    const int soff_b_ = cur_kt * (BK * 2);                                                             \
    /* (descriptors are made next to their use: hipcc drops the host stub of a kernel that reads one declared in an \
       outer scope) */                                                                                 \
    const __amdgpu_buffer_rsrc_t rsb_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, bnr, 0x00020000); \
    _Pragma("unroll") for (int i = 0; i < BR; ++i) {                                                   \
      const int bv_ = bvoff[i] + 0;                                                                    \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb_, (lds_ptr_t)(b_ + (8 * wave_s + 32 * i) * BK), 16, bv_, soff_b_, 0, 0); \
    }                                                                                                  \
    const bool second_ = cur_c >= p.c0;                                                                \
    const int cs2_ = (second_ ? p.c1 : p.c0) * 2;                                                      \
    const int soff_a_ = (cur_ky * p.ws + cur_kx) * cs2_ + (second_ ? cur_c - p.c0 : cur_c) * 2;        \
    const __amdgpu_buffer_rsrc_t rs_ =                                                                 \
        __builtin_amdgcn_make_buffer_rsrc((void*)(second_ ? abase1 : abase0), 0, second_ ? anr1 : anr0, 0x00020000); \
    const unsigned bit_ = 1u << cur_tap;                                                               \
    _Pragma("unroll") for (int i = 0; i < AR; ++i) {                                                   \
      const int vo_ = (tapmask[i] & bit_) ? __mul24(apix[i], cs2_) + lc * 16 : OOB; /* < 2^24 pixels: host check */                            \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(a_ + (8 * wave_s + 32 * i) * BK), 16, vo_, soff_a_, 0, 0); \
    }                                                                                                  \
    if (ADV_) {                                                                                        \
      ++cur_kt;                                                                                        \
      cur_c += BK;                                                                                     \
      if (cur_c >= p.cin) {                                                                            \
        cur_c = 0;                                                                                     \
        ++cur_tap;                                                                                     \
        if (++cur_kx == p.ksize) {                                                                     \
          cur_kx = 0;                                                                                  \
          ++cur_ky;                                                                                    \
        }                                                                                              \
      }                                                                                                \
    }                                                                                                  \
  }
#define VSD_ISSUE_TILE(KT_, SLOT_)                                                                    \
  {                                                                                                   \
    const int kt_ = (KT_);                                                                            \
    half_t* a_ = reinterpret_cast<half_t*>(smem) + (SLOT_) * STAGE_HALFS;                             \
    half_t* b_ = a_ + BM * BK;                                                                        \
    _Pragma("unroll") for (int i = 0; i < BR; ++i) {                                                  \
      const half_t* g_ = nvalid[i] ? (wrow[i] - cc * 8 + lc * 8 + (size_t)kt_ * BK) : p.zeros;        \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)g_, (lds_ptr_t)(b_ + (8 * wave + 32 * i) * BK), 16, 0, 0); \
    }                                                                                                 \
    int k_, cs_;                                                                                      \
    const half_t* src_;                                                                               \
    bool kok_ = true;                                                                                 \
    if (!GENERIC) {                                                                                   \
      k_ = kt_ * BK;                                                                                  \
    } else {                                                                                          \
      k_ = kt_ * BK + lc * 8;                                                                         \
      kok_ = k_ < p.K;                                                                                \
    }                                                                                                 \
    const int tap_ = k_ / p.cin;                                                                      \
    int c_ = k_ - tap_ * p.cin;                                                                       \
    const int ky_ = tap_ / p.ksize, kx_ = tap_ - ky_ * p.ksize;                                       \
    if (!GENERIC && c_ >= p.c0) {                                                                     \
      src_ = p.src1; cs_ = p.c1; c_ -= p.c0;                                                          \
    } else {                                                                                          \
      src_ = p.src0; cs_ = p.c0;                                                                      \
    }                                                                                                 \
    if (!GENERIC) c_ += lc * 8;                                                                       \
    _Pragma("unroll") for (int i = 0; i < AR; ++i) {                                                  \
      int iy = iy0[i] + ky_, ix = ix0[i] + kx_;                                                       \
      bool ok = kok_ && mvalid[i] && (unsigned)iy < (unsigned)p.hi && (unsigned)ix < (unsigned)p.wi;  \
      const int sy = (int)(((unsigned)iy * p.rmul_y) >> p.rshift), sx = (int)(((unsigned)ix * p.rmul_x) >> p.rshift); \
      const half_t* g_ = ok ? src_ + ((size_t)(ib[i] + sy * p.ws + sx)) * cs_ + c_ : p.zeros;                 \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)g_, (lds_ptr_t)(a_ + (8 * wave + 32 * i) * BK), 16, 0, 0); \
    }                                                                                                 \
  }
    if constexpr (!ILV) {
  #pragma unroll
      for (int st = 0; st < STAGES - 1; ++st)
        if (st < nt) {
          if constexpr (FAST) VSD_ISSUE_FAST(st, true)
          else VSD_ISSUE_TILE(kt_begin + st, st)
        }
      VSD_LN_ROWSTATS()
      CPROBE(0)
      WGTL_MARK(c)
      int slot = 0;
      for (int t = 0; t < nt; ++t) {
        // tile t has landed once all but the younger tiles' loads are done; then everyone's has (barrier)
        const int rem = min(STAGES - 2, nt - 1 - t);
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CPROBE(1)
        __builtin_amdgcn_s_barrier();
        CPROBE(2)
        if (t + STAGES - 1 < nt) {
          int ns = slot + STAGES - 1;
          if (ns >= STAGES) ns -= STAGES;
          if constexpr (FAST) VSD_ISSUE_FAST(ns, true)
          else VSD_ISSUE_TILE(kt_begin + t + STAGES - 1, ns)
        }
        CPROBE(3)
        const half_t* a = reinterpret_cast<const half_t*>(smem) + slot * STAGE_HALFS;
        const half_t* b = a + BM * BK;
  #pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          half8 af[FM], bf[FN];
  #pragma unroll
          for (int i = 0; i < FM; ++i) {
            int r = wm * TM + i * 16 + fr;
            af[i] = *reinterpret_cast<const half8*>(a + r * BK + (((ks * 4 + fq) ^ (r & 7)) << 3));
          }
  #pragma unroll
          for (int j = 0; j < FN; ++j) {
            int r = wn * TN + j * 16 + fr;
            bf[j] = *reinterpret_cast<const half8*>(b + r * BK + (((ks * 4 + fq) ^ (r & 7)) << 3));
          }
  #pragma unroll
          for (int i = 0; i < FM; ++i)
  #pragma unroll
            for (int j = 0; j < FN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        CPROBE(4)
        if (++slot == STAGES) slot = 0;
      }

    } else {
      // Interleaved form: every iteration is ONE basic block -- the next tile's LDS-DMA issue is unconditional (tile
      // index clamped: the last iterations re-fetch the final tile into a slot nobody reads) so the wait count is a
      // constant and the scheduler may spread the DMA issues and LDS fragment reads between the MFMAs
      // (sched_group_barrier), instead of running "all loads, then all reads, then all MFMAs" back to back.
      const int kt_last = kt_end - 1;
#pragma unroll
      for (int st = 0; st < STAGES - 1; ++st) {
        if constexpr (FAST) VSD_ISSUE_FAST(st, cur_kt < kt_last)
        else VSD_ISSUE_TILE(min(kt_begin + st, kt_last), st)
      }
      VSD_LN_ROWSTATS()
      WGTL_MARK(c)
      int slot = 0;
      constexpr int NM = FM * FN * 2;       // MFMAs per tile per wave
      for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * LPT) : "memory");
        __builtin_amdgcn_s_barrier();
        int ns = slot + STAGES - 1;
        if (ns >= STAGES) ns -= STAGES;
        // ---- per-tile scalars of the tile to fetch (same arithmetic as VSD_ISSUE_TILE / VSD_ISSUE_FAST)
        half_t* na = reinterpret_cast<half_t*>(smem) + ns * STAGE_HALFS;
        half_t* nb = na + BM * BK;
        [[maybe_unused]] int ktn = 0, k_ = 0, cs_ = 0, c_ = 0, ky_ = 0, kx_ = 0;
        [[maybe_unused]] const half_t* src_ = nullptr;
        [[maybe_unused]] bool kok_ = true;
        [[maybe_unused]] int f_soff_b = 0, f_soff_a = 0, f_cs2 = 0;
        [[maybe_unused]] unsigned f_bit = 0;
        const bool f_second = FAST && cur_c >= p.c0;
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t f_rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(f_second ? abase1 : abase0), 0, f_second ? anr1 : anr0, 0x00020000);
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t f_rsb = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, bnr, 0x00020000);
        if constexpr (FAST) {
          f_soff_b = cur_kt * (BK * 2);
          f_cs2 = (f_second ? p.c1 : p.c0) * 2;
          f_soff_a = (cur_ky * p.ws + cur_kx) * f_cs2 + (f_second ? cur_c - p.c0 : cur_c) * 2;
          f_bit = 1u << cur_tap;
        } else {
          ktn = min(kt_begin + t + STAGES - 1, kt_last);
          if (!GENERIC) {
            k_ = ktn * BK;
          } else {
            k_ = ktn * BK + lc * 8;
            kok_ = k_ < p.K;
          }
          const int tap_ = k_ / p.cin;
          c_ = k_ - tap_ * p.cin;
          ky_ = tap_ / p.ksize;
          kx_ = tap_ - ky_ * p.ksize;
          if (!GENERIC && c_ >= p.c0) {
            src_ = p.src1; cs_ = p.c1; c_ -= p.c0;
          } else {
            src_ = p.src0; cs_ = p.c0;
          }
          if (!GENERIC) c_ += lc * 8;
        }
        const half_t* a = reinterpret_cast<const half_t*>(smem) + slot * STAGE_HALFS;
        const half_t* b = a + BM * BK;
        half8 af[2][FM], bf[2][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          int r = wm * TM + i * 16 + fr;
          af[0][i] = *reinterpret_cast<const half8*>(a + r * BK + (((fq) ^ (r & 7)) << 3));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          int r = wn * TN + j * 16 + fr;
          bf[0][j] = *reinterpret_cast<const half8*>(b + r * BK + (((fq) ^ (r & 7)) << 3));
        }
        // ---- LPT pieces: one LDS-DMA issue, then MPP MFMAs; the k-step-1 fragments are read half way
#pragma unroll
        for (int pc = 0; pc < LPT; ++pc) {
          if constexpr (FAST) {
            if (pc < BR) {
              const int bv_ = bvoff[pc < BR ? pc : 0] + 0;
              __builtin_amdgcn_raw_ptr_buffer_load_lds(f_rsb, (lds_ptr_t)(nb + (8 * wave_s + 32 * pc) * BK), 16, bv_, f_soff_b, 0, 0);
            } else {
              const int i = pc - BR;
              const int vo_ = (tapmask[i] & f_bit) ? __mul24(apix[i], f_cs2) + lc * 16 : OOB;
              __builtin_amdgcn_raw_ptr_buffer_load_lds(f_rs, (lds_ptr_t)(na + (8 * wave_s + 32 * i) * BK), 16, vo_, f_soff_a, 0, 0);
            }
          } else if (pc < BR) {
            const int i = pc;
            const half_t* g_ = nvalid[i] ? (wrow[i] - cc * 8 + lc * 8 + (size_t)ktn * BK) : p.zeros;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)g_, (lds_ptr_t)(nb + (8 * wave + 32 * i) * BK), 16, 0, 0);
          } else {
            const int i = pc - BR;
            int iy = iy0[i] + ky_, ix = ix0[i] + kx_;
            bool ok = kok_ && mvalid[i] && (unsigned)iy < (unsigned)p.hi && (unsigned)ix < (unsigned)p.wi;
            const int sy = (int)(((unsigned)iy * p.rmul_y) >> p.rshift), sx = (int)(((unsigned)ix * p.rmul_x) >> p.rshift);
            const half_t* g_ = ok ? src_ + ((size_t)(ib[i] + sy * p.ws + sx)) * cs_ + c_ : p.zeros;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)g_, (lds_ptr_t)(na + (8 * wave + 32 * i) * BK), 16, 0, 0);
          }
          // (256-row tiles: both k-steps' fragments live at once are 96 registers on top of 128 accumulators -- the compiler then
          //  shuffles accumulators through AGPR copies inside the loop (340 v_accvgpr moves per tile seen); their k-step-1
          //  fragments are therefore read two pieces before they are needed instead of at the top)
          if (pc == (BM >= 256 ? LPT / 2 - 2 : 0)) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
              int r = wm * TM + i * 16 + fr;
              af[1][i] = *reinterpret_cast<const half8*>(a + r * BK + (((4 + fq) ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
              int r = wn * TN + j * 16 + fr;
              bf[1][j] = *reinterpret_cast<const half8*>(b + r * BK + (((4 + fq) ^ (r & 7)) << 3));
            }
          }
#pragma unroll
          for (int idx = pc * NM / LPT; idx < (pc + 1) * NM / LPT; ++idx) {  // this piece's share of the NM MFMAs
            const int ks = idx / (FM * FN), ij = idx % (FM * FN);        // k-step 0 first, then k-step 1
            const int i = ij / FN, j = ij % FN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);  // keep the DMA / MFMA alternation as written
        }
        if constexpr (FAST) {
          if (cur_kt < kt_last) {  // clamped like ktn: past the end the last tile is fetched again into a slot nobody reads
            ++cur_kt;
            cur_c += BK;
            if (cur_c >= p.cin) {
              cur_c = 0;
              ++cur_tap;
              if (++cur_kx == p.ksize) {
                cur_kx = 0;
                ++cur_ky;
              }
            }
          }
        }
        if (++slot == STAGES) slot = 0;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant tail fetches must land before the LDS is reused
    }
    __syncthreads();  // every wave is done reading the ring before the epilogue reuses the LDS
#undef VSD_ISSUE_TILE
#undef VSD_ISSUE_FAST
  }

  WGTL_LOOP()
#define EPI_PART split
#define EPI_NPARTS p.split_k
#define EPI_EXIT { WGTL_END(0) return; }
#include "conv_epilogue.inc"
#undef EPI_PART
#undef EPI_NPARTS
#undef EPI_EXIT
  CPROBE(6)
  CPROBE_OUT()
  WGTL_END(0)
}

// the FAST form exists for the direct-to-LDS rings only (STAGES >= 3)
template <int BM, int BN, int STAGES, bool ILV>
struct FastLaunch {
  static void go(const ConvParams& p, int grid, hipStream_t s) {
    hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, false, STAGES, ILV, true>), dim3(grid), dim3(256), 0, s, p);
  }
};
template <int BM, int BN, bool ILV>
struct FastLaunch<BM, BN, 0, ILV> {
  static void go(const ConvParams& p, int grid, hipStream_t s) {
    hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, false, 0, ILV, false>), dim3(grid), dim3(256), 0, s, p);
  }
};

template <int BM, int BN, int STAGES, bool ILV>
void launch2(const ConvParams& p, int grid, hipStream_t s) {
  if (p.generic) hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, true, STAGES, ILV, false>), dim3(grid), dim3(256), 0, s, p);
  else if (STAGES >= 3 && p.fast) FastLaunch<BM, BN, STAGES, ILV>::go(p, grid, s);
  else hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, false, STAGES, ILV, false>), dim3(grid), dim3(256), 0, s, p);
}
template <int BM, int BN>
void launch(const ConvParams& p, int grid, int stages, hipStream_t s) {
  if (stages == 0) launch2<BM, BN, 0, false>(p, grid, s);
  else if (stages == 3) launch2<BM, BN, 3, false>(p, grid, s);
  else if (stages == 4) launch2<BM, BN, 4, false>(p, grid, s);
  else if (stages == 5) launch2<BM, BN, 3, true>(p, grid, s);
  else launch2<BM, BN, 4, true>(p, grid, s);
}


}  // namespace

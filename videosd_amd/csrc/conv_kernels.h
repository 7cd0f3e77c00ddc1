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
#include <stddef.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

constexpr int BK = 64;  // halfs per K tile (128-byte LDS rows)
constexpr int VSD_GROUP_MAX = VSD_CONV_GROUP_MAX;  // problems per grouped launch (conv_gemm_group_kernel; include/vsd.h)

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

// the argument block of a grouped launch (conv_gemm_group_kernel below): problems start[i] <= blockIdx.x < start[i + 1]
struct ConvGroup {
  ConvParams p[VSD_GROUP_MAX];
  int start[VSD_GROUP_MAX + 1];
  int n;
};
static_assert(sizeof(ConvGroup) <= 4000, "the group's argument block must fit the 4 KB kernel-argument segment");

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
void vsd_launch_conv_256x256(const ConvParams& p, int grid, int stages, hipStream_t s);
void vsd_launch_conv_halo(const ConvParams& p, int bm, int bn, int grid, hipStream_t s);
void vsd_launch_conv_c64(const ConvParams& p, int grid, hipStream_t s);  // conv_c64.hip: pipeline 10
void vsd_launch_conv_group_64x64(const ConvGroup& g, int grid, int stages, hipStream_t s);
void vsd_launch_conv_group_64x128(const ConvGroup& g, int grid, int stages, hipStream_t s);
void vsd_launch_conv_group_128x64(const ConvGroup& g, int grid, int stages, hipStream_t s);
void vsd_launch_conv_group_128x128(const ConvGroup& g, int grid, int stages, hipStream_t s);
void vsd_launch_splitk_reduce_group(const ConvGroup& g, int grid, hipStream_t s);
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
                                                const f32x4 brv_hi, const bool has_raw = false,
                                                const half8 bias_raw = (half8){0, 0, 0, 0, 0, 0, 0, 0},
                                                const half8 rowvec_raw = (half8){0, 0, 0, 0, 0, 0, 0, 0}) {
  // n is a multiple of 8; handles n + 8 > N by scalar fallback
  // has_raw: bias / time vector of these 8 columns as the caller fetched them beside its other operands (the reducer kernel: its
  // own loads here sat behind the slab sums, one more dependent round trip per launch); added one after the other, as below
  const bool full = (n + 8 <= p.N);
  if (has_brv) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] += brv_lo[i];
      v[4 + i] += brv_hi[i];
    }
  } else if (has_raw) {
    if (p.bias) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += (float)bias_raw[i];
    }
    if (p.rowvec) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += (float)rowvec_raw[i];
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
  // scale and first residual as ONE fused multiply-add wherever both exist -- here, in the straight-line walks of conv_epilogue.inc
  // and therefore behind the reducer kernel and the in-launch split-K tails alike.  (Round 5: this function multiplied inside a
  // branch and added later -- two roundings -- while the walks' `x *= sc; x += r` was contracted by the compiler: a scaled layer
  // with a residual, the ControlNet merges, differed in the last bit between its reducer form and its in-launch form.)  x * 1.0f
  // and fma(x, 1.0f, r) are exact: unscaled layers keep their bits.
  const float sc = p.out_scale_dev ? *p.out_scale_dev : p.out_scale;  // (uniform address: a scalar load)
  if (p.out_t && n >= p.t_col0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= sc;
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
      for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], sc, (float)r[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= sc;
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
    VSD_OUT_STORE8(p.out + (size_t)m * p.ldo + n, o);
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
      float x = p.residual ? __builtin_fmaf(v[i], sc, (float)p.residual[(size_t)m * p.ldr + n + i]) : v[i] * sc;
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
// NW = waves per workgroup: 4 (2 x 2, one wave per SIMD) or 8 (4 x 2, two waves per SIMD: the same tile with half the accumulators
//   and half the LDS-DMA instructions per wave -- one wave's DMA issue / LDS wait beside its SIMD partner's MFMAs; round 6).
template <int BM, int BN, bool GENERIC, int STAGES, bool ILV, bool FAST, int NW = 4>
__global__ __launch_bounds__(NW * 64) void conv_gemm_kernel(const ConvParams p) {
  VSD_CUT(VSD_CUT_CONV_GEMM, p.cut)
  prefetch_kernargs();
  WGTL_START()
#define CONV_BID blockIdx.x
#include "conv_gemm_body.inc"
#undef CONV_BID
}

// the eight-wave form as a kernel of its own: two waves per SIMD is ALL it is launched for, so the register allocator may use the
// 256 registers per lane that leaves (without the attribute it aimed lower and spilled 64 registers of the 256 x 128 tile)
template <int BM, int BN, int STAGES, bool ILV>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_gemm8_kernel(const ConvParams p) {
  constexpr bool GENERIC = false, FAST = true;
  constexpr int NW = 8;
  VSD_CUT(VSD_CUT_CONV_GEMM, p.cut)
  prefetch_kernargs();
  WGTL_START()
#define CONV_BID blockIdx.x
#include "conv_gemm_body.inc"
#undef CONV_BID
}

// ---------------------------------------------------------------- several problems in one launch
// Up to VSD_GROUP_MAX independent problems of ONE kernel form (same tile, pipeline, buffer-load path) in one grid: workgroup b
// belongs to problem i with start[i] <= b < start[i + 1] and runs exactly what the plain kernel would run for it (its own split-K
// slabs / tickets included: every problem carries its own workspace and counter pointers).  For chains of small launches that do
// not depend on each other -- the ControlNet's 13 zero-conv merges per denoising step (lcm_controlnet.py:558-577's
// down_block_additional_residuals): 11.5 us of a lone frame each as launches of their own, measured by leaving them out.

template <int BM, int BN, bool GENERIC, int STAGES, bool ILV, bool FAST, int NW = 4>
__global__ __launch_bounds__(NW * 64) void conv_gemm_group_kernel(const ConvGroup g) {
  int prob = 0, first = 0;
#pragma unroll
  for (int i = 1; i < VSD_GROUP_MAX; ++i)
    if (i < g.n && (int)blockIdx.x >= g.start[i]) {
      prob = i;
      first = g.start[i];
    }
  // This problem's argument block, read from the KERNEL-ARGUMENT SEGMENT at a uniform offset (scalar loads).  `g.p[prob]` -- a
  // runtime index into a by-value struct -- made the compiler copy all eight blocks to scratch memory in the 128 x 128 and two more
  // instantiations (3.5 KB per lane, every later p.field a scratch load: scripts/kernel_resources.py, round 5).
  typedef const __attribute__((address_space(4))) ConvParams* kparams_t;
  typedef const __attribute__((address_space(4))) char* kbytes_t;
  const kparams_t kp = (kparams_t)((kbytes_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ConvGroup, p)) +
                       __builtin_amdgcn_readfirstlane(prob);
  const ConvParams p = *(const ConvParams*)kp;  // (the copy is by scalar loads: the compiler sees through the cast to the constant address space)
  VSD_CUT(VSD_CUT_CONV_GEMM, p.cut)
  WGTL_START()
  const int conv_bid = (int)blockIdx.x - first;
#define CONV_BID conv_bid
#include "conv_gemm_body.inc"
#undef CONV_BID
}

template <int BM, int BN>
void launch_group(const ConvGroup& g, int grid, int stages, hipStream_t s) {
  // (buffer-load path only: every member has Cin % 64 == 0 per source and no resize; pipelines 3 / 5 = the 3-stage ring, plain /
  //  interleaved)
  if (stages >= 8) {
    if constexpr (BM * BN >= 128 * 128) {
      if (stages == 8) hipLaunchKernelGGL((conv_gemm_group_kernel<BM, BN, false, 3, false, true, 8>), dim3(grid), dim3(512), 0, s, g);
      else hipLaunchKernelGGL((conv_gemm_group_kernel<BM, BN, false, 3, true, true, 8>), dim3(grid), dim3(512), 0, s, g);
    }
  } else if (stages == 5) hipLaunchKernelGGL((conv_gemm_group_kernel<BM, BN, false, 3, true, true>), dim3(grid), dim3(256), 0, s, g);
  else hipLaunchKernelGGL((conv_gemm_group_kernel<BM, BN, false, 3, false, true>), dim3(grid), dim3(256), 0, s, g);
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
// pipelines 8 / 9: the 3-stage ring (plain / interleaved) on EIGHT waves, buffer-load path only
template <int BM, int BN, int STAGES = 3>
void launch8(const ConvParams& p, int grid, int stages, hipStream_t s) {
  if (stages == 8) hipLaunchKernelGGL((conv_gemm8_kernel<BM, BN, STAGES, false>), dim3(grid), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((conv_gemm8_kernel<BM, BN, STAGES, true>), dim3(grid), dim3(512), 0, s, p);
}
template <int BM, int BN>
void launch(const ConvParams& p, int grid, int stages, hipStream_t s) {
  if (stages >= 8) {
    if constexpr (BM * BN >= 128 * 128) launch8<BM, BN>(p, grid, stages, s);  // (smaller tiles: refused by the host, conv_gemm.hip)
  } else if (stages == 0) launch2<BM, BN, 0, false>(p, grid, s);
  else if (stages == 3) launch2<BM, BN, 3, false>(p, grid, s);
  else if (stages == 4) launch2<BM, BN, 4, false>(p, grid, s);
  else if (stages == 5) launch2<BM, BN, 3, true>(p, grid, s);
  else launch2<BM, BN, 4, true>(p, grid, s);
}


}  // namespace

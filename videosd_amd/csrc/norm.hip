// GroupNorm(+SiLU) and LayerNorm for NHWC fp16 tensors (gfx950).  HBM/L2-bound streaming kernels:
// 16-byte vector loads, per-thread fixed channel chunk so that per-channel scale/shift live in
// registers, wave-shuffle / LDS reductions in fp32.
//
// GroupNorm is two launches: `gn_stats` writes per-workgroup partial (sum, sumsq) per group in a fixed
// order (deterministic, no atomics), `gn_apply` folds the partials, then streams y = x*a[c] + b[c]
// (+SiLU).  The input may be the channel concat of two tensors (UNet up blocks) and the output is the
// concatenated, normalised tensor.  Algorithmic bytes: 2 reads + 1 write of hw*C fp16.
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int GN_MAX_PART = 128;

struct GnParams {
  const half_t* src0;
  const half_t* src1;
  int c0, c1, c, c8;  // c8 = c / 8
  int hw, groups, cpg;
  float eps;
  const half_t* gamma;
  const half_t* beta;
  int silu;
  half_t* out;
  float* part;  // [nblk][groups][2]
  int nblk;     // stats workgroups (per image)
  int rpp;      // rows per pass = blockDim / c8
  int batch;    // images stacked along the rows (blockIdx.y): each normalised with its own statistics
  VSD_CUT_FIELD
};

__device__ __forceinline__ half8 gn_load(const GnParams& p, int row, int ch8) {
  int c = ch8 * 8;
  const half_t* s = (c < p.c0) ? p.src0 + (size_t)row * p.c0 + c : p.src1 + (size_t)row * p.c1 + (c - p.c0);
  return *reinterpret_cast<const half8*>(s);
}

// Both kernels are latency- rather than bandwidth-bound at these sizes (a few MB, L2 resident): every thread issues
// its row loads four at a time before consuming them, and the partial folds are spread over all threads with every
// load of a thread independent of the others, so each phase costs about one memory round trip.
__device__ __forceinline__ void gn_stats_body(const GnParams& p) {
  VSD_CUT(VSD_CUT_GROUPNORM, p.cut)
  extern __shared__ float sm[];  // [16 planes][threads] per-thread channel sums (see below), folded in a fixed order
  const int t = threadIdx.x;
  const int ch8 = t % p.c8;
  const int rl = t / p.c8;
  const int rows_per_blk = (p.hw + p.nblk - 1) / p.nblk;
  const int img0 = blockIdx.y * p.hw;  // first row of this image
  const int r0 = img0 + blockIdx.x * rows_per_blk;
  const int r1 = min(img0 + p.hw, r0 + rows_per_blk);
  float s[8], q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = q[i] = 0.f;
  // rows in batches of four PREDICATED loads (an out-of-range row re-reads the block's first row and is masked to zero):
  // every batch is one memory round trip, the ragged tail included -- a per-row tail loop pays one round trip per row
  const half8 zero8 = (half8){0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = r0 + rl; r < r1; r += 4 * p.rpp) {
    const bool ok1 = r + p.rpp < r1, ok2 = r + 2 * p.rpp < r1, ok3 = r + 3 * p.rpp < r1;
    half8 x0 = gn_load(p, r, ch8), x1 = gn_load(p, ok1 ? r + p.rpp : r, ch8);
    half8 x2 = gn_load(p, ok2 ? r + 2 * p.rpp : r, ch8), x3 = gn_load(p, ok3 ? r + 3 * p.rpp : r, ch8);
    x1 = ok1 ? x1 : zero8;
    x2 = ok2 ? x2 : zero8;
    x3 = ok3 ? x3 : zero8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v0 = (float)x0[i], v1 = (float)x1[i], v2 = (float)x2[i], v3 = (float)x3[i];
      s[i] += v0; q[i] += v0 * v0;
      s[i] += v1; q[i] += v1 * v1;
      s[i] += v2; q[i] += v2 * v2;
      s[i] += v3; q[i] += v3 * v3;
    }
  }
  // LDS layout: 16 planes (channel-in-chunk i, sum | sumsq) of one word per THREAD -- sm[(2 i + which) * T + t].  A wave's
  // store then hits 64 consecutive words (no bank conflict); the first form kept each thread's 16 values together
  // (sm[rl][c][2]: a 64-byte stride between lanes, every second lane on the same bank -- LDS_BANK_CONFLICT / IDX_ACTIVE 0.62
  // in the round-3 counters).  The additions below run in the same order as before: same bits.
  const int T = blockDim.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sm[(2 * i) * T + t] = s[i];
    sm[(2 * i + 1) * T + t] = q[i];
  }
  __syncthreads();
  // Fold in two steps, both in a fixed order (deterministic): (1) every (channel, sum | sumsq) over the rpp row lanes --
  // thread j owns (plane, channel chunk): consecutive threads read consecutive words, rpp reads each, the result goes to row
  // lane 0's word (read and written by this thread only); (2) every (group, sum | sumsq) over its cpg channels.
  // (The first form did both in one loop of rpp * cpg dependent LDS reads on 2 * groups threads -- 120 reads deep at
  // C = 320 with the other 400 threads idle, a third of the kernel's ~10 us.)
  const int nval = 16 * p.c8;
  for (int j = t; j < nval; j += T) {
    const int pl = j / p.c8, c8i = j - pl * p.c8;
    float* w = sm + pl * T + c8i;
    float acc = 0.f;
    for (int rr = 0; rr < p.rpp; ++rr) acc += w[rr * p.c8];
    w[0] = acc;
  }
  __syncthreads();
  for (int i = t; i < p.groups * 2; i += T) {
    const int g = i >> 1, which = i & 1;
    float acc = 0.f;
    for (int k = 0; k < p.cpg; ++k) {
      const int c = g * p.cpg + k;
      acc += sm[(2 * (c & 7) + which) * T + (c >> 3)];
    }
    p.part[((size_t)blockIdx.y * p.nblk + blockIdx.x) * p.groups * 2 + i] = acc;
  }
}

__global__ void gn_stats_kernel(const GnParams p) { gn_stats_body(p); }
// (two tensors as one grid: common.h launch_pairable; blockIdx.z = which)
__global__ void gn_stats_pair_kernel(const Pair<GnParams> g) { gn_stats_body(g.p[blockIdx.z]); }

__device__ __forceinline__ void gn_apply_body(const GnParams& p) {
  VSD_CUT(VSD_CUT_GROUPNORM, p.cut)
  extern __shared__ float sm[];  // [groups][2] mean, rstd | [nch][groups*2] partial folds
  const int t = threadIdx.x;
  const int npairs = p.groups * 2;
  // The first batch of rows, gamma and beta do not depend on the statistics: their loads are issued FIRST, so that the
  // partial fold below (a dependent round trip of its own) runs under their latency instead of in front of it.
  const int ch8 = t % p.c8;
  const int rl = t / p.c8;
  const int rows_per_blk = (p.hw + gridDim.x - 1) / gridDim.x;
  const int img0 = blockIdx.y * p.hw;
  const int r0 = img0 + blockIdx.x * rows_per_blk;
  const int r1 = min(img0 + p.hw, r0 + rows_per_blk);
  const int rf = r0 + rl;  // this thread's first row
  const bool okf0 = rf < r1, okf1 = rf + p.rpp < r1, okf2 = rf + 2 * p.rpp < r1, okf3 = rf + 3 * p.rpp < r1;
  const int rsafe = okf0 ? rf : img0;
  half8 xf0 = gn_load(p, rsafe, ch8), xf1 = gn_load(p, okf1 ? rf + p.rpp : rsafe, ch8);
  half8 xf2 = gn_load(p, okf2 ? rf + 2 * p.rpp : rsafe, ch8), xf3 = gn_load(p, okf3 ? rf + 3 * p.rpp : rsafe, ch8);
  const half8 ga = *reinterpret_cast<const half8*>(p.gamma + ch8 * 8);
  const half8 be = *reinterpret_cast<const half8*>(p.beta + ch8 * 8);
  {
    // fold the statistics partials: (pair, chunk) per thread, chunk ch sums blocks ch, ch+nch, ... (16 loads at a time,
    // all independent), then the chunks are added in order: deterministic
    int nch = (int)blockDim.x / npairs;
    nch = nch < 1 ? 1 : (nch > 8 ? 8 : nch);
    float* fold = sm + npairs;
    for (int i = t; i < npairs * nch; i += blockDim.x) {
      const int pair = i % npairs, ch = i / npairs;
      const float* src = p.part + (size_t)blockIdx.y * p.nblk * npairs + pair;
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int b = ch + j * nch;
        v[j] = b < p.nblk ? src[(size_t)b * npairs] : 0.f;
      }
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += v[j];
      for (int b = ch + 16 * nch; b < p.nblk; b += nch) acc += src[(size_t)b * npairs];
      fold[ch * npairs + pair] = acc;
    }
    __syncthreads();
    for (int i = t; i < npairs; i += blockDim.x) {
      float acc = 0.f;
      for (int ch = 0; ch < nch; ++ch) acc += fold[ch * npairs + i];
      sm[i] = acc;
    }
  }
  __syncthreads();
  float mean_r = 0.f, rstd_r = 0.f;
  for (int g = t; g < p.groups; g += blockDim.x) {  // groups <= blockDim is not guaranteed for tiny C
    float n = (float)p.hw * (float)p.cpg;
    float mean = sm[g * 2] / n;
    float var = fmaxf(sm[g * 2 + 1] / n - mean * mean, 0.f);
    mean_r = mean;
    rstd_r = rsqrtf(var + p.eps);
    sm[g * 2] = mean_r;       // in place: entry g is read and written by this thread only
    sm[g * 2 + 1] = rstd_r;
  }
  __syncthreads();
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int g = (ch8 * 8 + i) / p.cpg;
    float mean = sm[g * 2], rstd = sm[g * 2 + 1];
    a[i] = rstd * (float)ga[i];
    b[i] = (float)be[i] - mean * a[i];
  }
  auto norm_store = [&](int r, const half8& x) {
    half8 y;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = (float)x[i] * a[i] + b[i];
      if (p.silu) v = silu_f(v);
      y[i] = (half_t)v;
    }
    *reinterpret_cast<half8*>(p.out + (size_t)r * p.c + ch8 * 8) = y;
  };
  if (okf0) norm_store(rf, xf0);
  if (okf1) norm_store(rf + p.rpp, xf1);
  if (okf2) norm_store(rf + 2 * p.rpp, xf2);
  if (okf3) norm_store(rf + 3 * p.rpp, xf3);
  for (int r = rf + 4 * p.rpp; r < r1; r += 4 * p.rpp) {  // (more than four rows per thread: large images only)
    const bool ok1 = r + p.rpp < r1, ok2 = r + 2 * p.rpp < r1, ok3 = r + 3 * p.rpp < r1;
    half8 x0 = gn_load(p, r, ch8), x1 = gn_load(p, ok1 ? r + p.rpp : r, ch8);
    half8 x2 = gn_load(p, ok2 ? r + 2 * p.rpp : r, ch8), x3 = gn_load(p, ok3 ? r + 3 * p.rpp : r, ch8);
    norm_store(r, x0);
    if (ok1) norm_store(r + p.rpp, x1);
    if (ok2) norm_store(r + 2 * p.rpp, x2);
    if (ok3) norm_store(r + 3 * p.rpp, x3);
  }
}
__global__ void gn_apply_kernel(const GnParams p) { gn_apply_body(p); }
__global__ void gn_apply_pair_kernel(const Pair<GnParams> g) { gn_apply_body(g.p[blockIdx.z]); }

// ------------------------------------------------------------------ one-launch GroupNorm for small images
// One workgroup per (image, group): the group's cpg channels of every pixel (a cpg*2-byte piece of each NHWC row) are
// loaded ONCE into registers, reduced (wave shuffles + a fixed-order LDS fold: deterministic), normalised from the
// registers and stored.  The two-launch form above spends ~9 + ~6 us on these few-hundred-KB tensors, all of it launch
// and memory latency; this is one launch and one round trip.  VW = halfs per vector load (the alignment cpg allows),
// NV = vectors per row piece (cpg / VW), RMAX rows per thread.
// GPW = groups per workgroup (1 in every shipped instantiation).  Round 5 measured 2 groups of 20 channels / 4 of 10 per workgroup
// (80-byte row pieces, 16-byte vectors, half the workgroups) at 32 x 32 x 640: 9.6 against 8.3 us for one image, 12.5 against 13.2 for
// five (scripts/gn_bench.py) -- the kernel is launch + dependent round trips, not line traffic: not used.
template <int VW, int NV, int RMAX, int GPW = 1>
__device__ __forceinline__ void gn_fused_body(const GnParams& p) {
  VSD_CUT(VSD_CUT_GROUPNORM, p.cut)
  typedef _Float16 vec_t __attribute__((ext_vector_type(VW)));
  constexpr int CPG = NV * VW / GPW;  // channels per group (compile time: the element -> group map is static)
  __shared__ float red[GPW * (2 * 16 + 2)];
  const int t = threadIdx.x, T = blockDim.x;
  const int g0 = blockIdx.x * GPW;
  const size_t img0 = (size_t)blockIdx.y * p.hw;
  const int ch0 = g0 * CPG;
  vec_t x[RMAX][NV];
  float s[GPW], q[GPW];
#pragma unroll
  for (int g = 0; g < GPW; ++g) s[g] = q[g] = 0.f;
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    const int row = t + r * T;
    if (row < p.hw) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = ch0 + v * VW;  // a vector never straddles the two concat sources (c0 % VW == 0, host check)
        const half_t* src = (c < p.c0) ? p.src0 + (img0 + row) * p.c0 + c : p.src1 + (img0 + row) * p.c1 + (c - p.c0);
        x[r][v] = *reinterpret_cast<const vec_t*>(src);
      }
    }
  }
  // gamma / beta do not depend on the statistics: fetched with the rows (behind the barriers below their loads were a second,
  // dependent memory round trip in every one-launch GroupNorm)
  vec_t gav[NV], bev[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    gav[v] = *reinterpret_cast<const vec_t*>(p.gamma + ch0 + v * VW);
    bev[v] = *reinterpret_cast<const vec_t*>(p.beta + ch0 + v * VW);
  }
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    const int row = t + r * T;
    if (row < p.hw) {
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < VW; ++i) {
          const float f = (float)x[r][v][i];
          s[(v * VW + i) / CPG] += f;
          q[(v * VW + i) / CPG] += f * f;
        }
    }
  }
#pragma unroll
  for (int g = 0; g < GPW; ++g)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s[g] += __shfl_xor(s[g], o);
      q[g] += __shfl_xor(q[g], o);
    }
  const int wave = t >> 6, nw = T >> 6;
  if ((t & 63) == 0) {
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
      red[g * 34 + 2 * wave] = s[g];
      red[g * 34 + 2 * wave + 1] = q[g];
    }
  }
  __syncthreads();
  if (t < GPW) {
    float S = 0.f, Q = 0.f;
    for (int w = 0; w < nw; ++w) {
      S += red[t * 34 + 2 * w];
      Q += red[t * 34 + 2 * w + 1];
    }
    const float n = (float)p.hw * (float)CPG;
    const float mean = S / n;
    const float var = fmaxf(Q / n - mean * mean, 0.f);
    red[t * 34 + 32] = mean;
    red[t * 34 + 33] = rsqrtf(var + p.eps);
  }
  __syncthreads();
  float a[NV][VW], b[NV][VW];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const vec_t ga = gav[v], be = bev[v];
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      const int g = (v * VW + i) / CPG;
      a[v][i] = red[g * 34 + 33] * (float)ga[i];
      b[v][i] = (float)be[i] - red[g * 34 + 32] * a[v][i];
    }
  }
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    const int row = t + r * T;
    if (row < p.hw) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        vec_t y;
#pragma unroll
        for (int i = 0; i < VW; ++i) {
          float f = (float)x[r][v][i] * a[v][i] + b[v][i];
          if (p.silu) f = silu_f(f);
          y[i] = (half_t)f;
        }
        *reinterpret_cast<vec_t*>(p.out + (img0 + row) * p.c + ch0 + v * VW) = y;
      }
    }
  }
}

// (groups of more than 20 channels run on images of at most 256 pixels -- gn_try_fused -- i.e. at most 256 threads: said here,
//  the 40-channel form keeps its row, gamma and beta in registers; under the default 1024-thread bound it had 128 and spilled 18)
template <int VW, int NV, int RMAX>
__global__ __launch_bounds__(VW * NV > 20 ? 256 : 1024) void gn_fused_kernel(const GnParams p) {
  gn_fused_body<VW, NV, RMAX>(p);
}
template <int VW, int NV, int RMAX>
__global__ __launch_bounds__(VW * NV > 20 ? 256 : 1024) void gn_fused_pair_kernel(const Pair<GnParams> g) {
  gn_fused_body<VW, NV, RMAX>(g.p[blockIdx.z]);
}

// launches the one-kernel form when the shape fits it (dry: only says whether it would); returns false otherwise
static bool gn_try_fused(vsd_ctx* ctx, const GnParams& p, int batch, hipStream_t s, bool dry) {
  if (getenv("VSD_GN_NO_FUSED")) return false;
  // measured on MI355X (us, one launch vs two): 8x8x1280 5.5 vs 14.5; 16x16x1280 6.3 vs 14.2; 32x32x640 11.5 vs 13.6;
  // it loses with more registers per thread (32x32x1920: 133 vs 14) and with 20-byte row pieces at 64x64 (31 vs 15)
  if (!((p.hw <= 256 && p.cpg <= 40) || (p.hw <= 1024 && p.cpg <= 20))) return false;
  const int threads = p.hw <= 64 ? 64 : (p.hw <= 128 ? 128 : (p.hw <= 256 ? 256 : 1024));
  dim3 grid(p.groups, batch), block(threads);
  const bool a8 = p.cpg % 8 == 0 && p.c0 % 8 == 0 && p.c1 % 8 == 0;
  const bool a4 = p.cpg % 4 == 0 && p.c0 % 4 == 0 && p.c1 % 4 == 0;
#define GN_GO(VW_, NV_)                                                                           \
  {                                                                                               \
    if (dry) return true;                                                                         \
    launch_pairable(ctx, gn_fused_kernel<VW_, NV_, 1>, gn_fused_pair_kernel<VW_, NV_, 1>, grid, block, 0, s, p); /* threads >= hw: one row per thread */ \
    return true;                                                                                  \
  }
  if (a8 && p.cpg == 40) GN_GO(8, 5)
  if (a8 && p.cpg == 8) GN_GO(8, 1)
  if (a8 && p.cpg == 16) GN_GO(8, 2)
  if (a4 && p.cpg == 20) GN_GO(4, 5)
  if (a4 && p.cpg == 4) GN_GO(4, 1)
  if (a4 && p.cpg == 12) GN_GO(4, 3)
  if (p.cpg == 10 && p.c0 % 2 == 0 && p.c1 % 2 == 0) GN_GO(2, 5)
  if (p.cpg == 2 && p.c0 % 2 == 0 && p.c1 % 2 == 0) GN_GO(2, 1)
  if (p.cpg == 6 && p.c0 % 2 == 0 && p.c1 % 2 == 0) GN_GO(2, 3)
#undef GN_GO
  return false;
}

// ------------------------------------------------------------------ LayerNorm: one wave per row
template <int NCH>  // 16-byte chunks per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, int rows, int c, const half_t* gamma,
                                                        const half_t* beta, float eps, half_t* out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c8 = c >> 3;
  half8 v[NCH];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int ch = lane + 64 * i;
    if (ch < c8) {
      v[i] = *reinterpret_cast<const half8*>(x + (size_t)row * c + ch * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)v[i][j];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)c;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int ch = lane + 64 * i;
    if (ch < c8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float d = (float)v[i][j] - mean;
        q += d * d;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / (float)c + eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int ch = lane + 64 * i;
    if (ch < c8) {
      half8 g = *reinterpret_cast<const half8*>(gamma + ch * 8);
      half8 b = *reinterpret_cast<const half8*>(beta + ch * 8);
      half8 y;
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = (half_t)(((float)v[i][j] - mean) * rstd * (float)g[j] + (float)b[j]);
      *reinterpret_cast<half8*>(out + (size_t)row * c + ch * 8) = y;
    }
  }
}

}  // namespace

extern "C" int64_t vsd_groupnorm_workspace_bytes(int hw, int c, int groups) {
  (void)hw;
  (void)c;
  return (int64_t)GN_MAX_PART * groups * 2 * sizeof(float);
}

// kernel launches vsd_groupnorm_batched issues for this shape (1: one workgroup per (image, group); 2: statistics + apply)
extern "C" int vsd_groupnorm_launches(int c0, int c1, int hw, int batch, int groups) {
  GnParams p;
  memset(&p, 0, sizeof p);
  p.c0 = c0; p.c1 = c1; p.c = c0 + c1;
  if (hw <= 0 || groups <= 0 || batch < 1 || p.c0 % 8 || p.c1 % 8 || p.c % groups) return 0;
  p.c8 = p.c / 8; p.hw = hw; p.groups = groups; p.cpg = p.c / groups; p.batch = batch;
  return gn_try_fused(nullptr, p, batch, nullptr, true) ? 1 : 2;
}

extern "C" int vsd_groupnorm(vsd_ctx* ctx, const void* src0, const void* src1, int c0, int c1, int hw, int groups,
                             float eps, const void* gamma, const void* beta, int silu, void* out, void* workspace,
                             void* stream) {
  return vsd_groupnorm_batched(ctx, src0, src1, c0, c1, hw, 1, groups, eps, gamma, beta, silu, out, workspace, stream);
}

extern "C" int vsd_groupnorm_batched(vsd_ctx* ctx, const void* src0, const void* src1, int c0, int c1, int hw, int batch,
                                     int groups, float eps, const void* gamma, const void* beta, int silu, void* out,
                                     void* workspace, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
#ifdef VSD_PROBE
  { static const bool skip = getenv("VSD_SKIP_GN") != nullptr; if (skip) return VSD_OK; }  // (what-if probe builds only)
#endif
  if (batch < 1 || batch > 65535) return vsd_fail(ctx, VSD_ERR_ARG, "groupnorm: batch=%d", batch);
  GnParams p;
  p.batch = batch;
  VSD_CUT_SET(p)
  p.src0 = (const half_t*)src0;
  p.src1 = (const half_t*)src1;
  p.c0 = c0;
  p.c1 = src1 ? c1 : 0;
  p.c = p.c0 + p.c1;
  if (!src0 || !out || !gamma || !beta || !workspace) return vsd_fail(ctx, VSD_ERR_ARG, "groupnorm: null pointer");
  if (p.c0 % 8 || p.c1 % 8 || hw <= 0 || groups <= 0 || p.c % groups)
    return vsd_fail(ctx, VSD_ERR_ARG, "groupnorm: bad shape c0=%d c1=%d hw=%d groups=%d", c0, c1, hw, groups);
  p.c8 = p.c / 8;
  if (p.c8 > 1024) return vsd_fail(ctx, VSD_ERR_ARG, "groupnorm: C=%d too large", p.c);
  if (groups > 256) return vsd_fail(ctx, VSD_ERR_ARG, "groupnorm: groups=%d too large", groups);
  p.hw = hw; p.groups = groups; p.cpg = p.c / groups; p.eps = eps;
  p.gamma = (const half_t*)gamma; p.beta = (const half_t*)beta; p.silu = silu;
  p.out = (half_t*)out; p.part = (float*)workspace;
  p.rpp = p.c8 >= 512 ? 1 : 512 / p.c8;
  if (p.rpp > hw) p.rpp = hw;
  const int threads = p.c8 * p.rpp;
  int nblk = cdiv(hw, 4 * p.rpp);
  // statistics workgroups per image: with several images per launch fewer, longer workgroups win (fewer partials for
  // the apply kernel to fold, the grid still covers the chip) -- MI355X, 5 images: 64x64x640 28.1 -> 23.7 us per pair,
  // 64x64x960 31.3 -> 27.3, 32x32x1920 25.2 -> 21.0 at 32 instead of 128 (16: slower again)
  static const int nblk_env = getenv("VSD_GN_NBLK") ? atoi(getenv("VSD_GN_NBLK")) : 0;  // (benchmarking)
  int nblk_cap = 160 / batch;
  nblk_cap = nblk_cap < 32 ? 32 : nblk_cap;
  if (nblk_env > 0) nblk_cap = nblk_env;
  if (nblk > nblk_cap) nblk = nblk_cap;
  if (nblk > GN_MAX_PART) nblk = GN_MAX_PART;
  if (nblk < 1) nblk = 1;
  p.nblk = nblk;
  hipStream_t s = (hipStream_t)stream;
  if (gn_try_fused(ctx, p, batch, s, true)) {
    LaunchScope ls(ctx, s, VSD_FAM_GROUPNORM, 0.0);
    gn_try_fused(ctx, p, batch, s, false);
    return ls.finish();
  }
  const size_t smem = (size_t)groups * 2 * 9 * sizeof(float);
  {
    const size_t smem_stats = (size_t)p.rpp * p.c * 2 * sizeof(float);
    LaunchScope ls(ctx, s, VSD_FAM_GROUPNORM, 0.0);
    launch_pairable(ctx, gn_stats_kernel, gn_stats_pair_kernel, dim3(nblk, batch), dim3(threads), (unsigned)smem_stats, s, p);
    int rc = ls.finish();
    if (rc) return rc;
  }
  {
    int ablk = cdiv(hw, 2 * p.rpp);
    static const int ablk_cap = getenv("VSD_GN_ABLK") ? atoi(getenv("VSD_GN_ABLK")) : 256;  // (benchmarking: apply workgroups per image)
    if (ablk > ablk_cap) ablk = ablk_cap;
    LaunchScope ls(ctx, s, VSD_FAM_GROUPNORM, 0.0);
    launch_pairable(ctx, gn_apply_kernel, gn_apply_pair_kernel, dim3(ablk, batch), dim3(threads), (unsigned)smem, s, p);
    return ls.finish();
  }
}

extern "C" int vsd_layernorm(vsd_ctx* ctx, const void* x, int rows, int c, const void* gamma, const void* beta,
                             float eps, void* out, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!x || !out || !gamma || !beta) return vsd_fail(ctx, VSD_ERR_ARG, "layernorm: null pointer");
  if (c % 8 || c <= 0 || c > 8 * 64 * 4 || rows <= 0) return vsd_fail(ctx, VSD_ERR_ARG, "layernorm: bad shape rows=%d c=%d", rows, c);
  hipStream_t s = (hipStream_t)stream;
  const int nch = cdiv(c / 8, 64);
  LaunchScope ls(ctx, s, VSD_FAM_LAYERNORM, 0.0);
  dim3 grid(cdiv(rows, 4)), block(256);
  const half_t* xx = (const half_t*)x;
  const half_t* g = (const half_t*)gamma;
  const half_t* b = (const half_t*)beta;
  half_t* o = (half_t*)out;
  switch (nch) {
    case 1: hipLaunchKernelGGL((layernorm_kernel<1>), grid, block, 0, s, xx, rows, c, g, b, eps, o); break;
    case 2: hipLaunchKernelGGL((layernorm_kernel<2>), grid, block, 0, s, xx, rows, c, g, b, eps, o); break;
    case 3: hipLaunchKernelGGL((layernorm_kernel<3>), grid, block, 0, s, xx, rows, c, g, b, eps, o); break;
    default: hipLaunchKernelGGL((layernorm_kernel<4>), grid, block, 0, s, xx, rows, c, g, b, eps, o); break;
  }
  return ls.finish();
}

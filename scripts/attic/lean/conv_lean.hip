// Lean form of the pointwise (1x1) layers for SMALL problems (pipeline 8, round 5): the deep levels of a lone frame.
//
// A 256x1280x1280 linear layer of the 16x16 level streams 3.3 MB of weights and 0.65 MB of activations -- 1 us of HBM time -- and
// took 8.7 us on the tiled kernel whatever its tile or split (profiles/round5_small_gemm_candidates.txt): a 64x64 workgroup walks
// 20 K tiles one behind the other (wait, barrier, LDS-DMA issue, fragment reads, MFMAs: ~275 ns each) around a 1-2 us prologue and
// a 1-3 us epilogue.  Here the K loop has no LDS and no barrier in it:
//   * a workgroup owns a SMALL output tile (16 or 32 rows x 32 or 64 columns: 80-320 workgroups where the tiled kernel had 20-80);
//   * its four waves each take one quarter of K (the K tiles [w * kt_per_split, (w + 1) * kt_per_split) -- exactly the K ranges of the
//     tiled kernel's split_k = 4) and load their MFMA operand fragments STRAIGHT from global memory into registers (a fragment of
//     v_mfma_f32_16x16x32_f16 is 8 consecutive halfs of one row: one 16-byte buffer load per lane, rows past M / N out of range =
//     zeros), ALL of them in flight at once (kt_per_split <= 5: K <= 1280 in four parts);
//   * the four partial tiles meet in LDS (one barrier) and are added in part order 0, 1, 2, 3 -- the order in which the tiled
//     kernel's split-K tail (and splitk_reduce_kernel) add their slabs -- then the layer's epilogue runs on the sum.
// Same MFMA instruction, same operand values, same order of accumulation inside a part and across parts: the result is
// BIT-IDENTICAL to vsd_conv_gemm at split_k = (K tiles / kt_per_split) in any tile (tests/test_ops_gpu.py), so a layer may take this
// form in one program form and the tiled one inside a grouped launch of another without changing a frame.
// Epilogues: bias / time vector, one activation (none, ReLU, SiLU, quick-GELU; ReLU after the residual), scale, one residual, row
// statistics of the output (64-column tiles).  Not here: the fused LayerNorm of the input (its consumers in these networks are
// the QKV, GEGLU and score layers, which have the epilogues below), GEGLU, tile softmax, transposed output, channel statistics,
// second output / residual (the host refuses; the tuner does not offer the form for them).
#include "conv_kernels.h"

namespace {

template <int FM, int FN, int NT>
__device__ __forceinline__ void lean_body(const ConvParams& p, const int bid) {
  constexpr int BM = 16 * FM, BN = 16 * FN, BNP = BN + 4, CH = BN / 8;
  constexpr int PART = BM * BNP;                       // floats of one wave's partial tile
  constexpr int NIT = (BM * CH + 255) / 256;           // 8-wide output chunks per thread
  __shared__ __attribute__((aligned(16))) float parts[4 * PART];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  int tile_m, tile_n;
  block_to_tile(p, bid, tile_m, tile_n);               // (the grid is not split over K: group = column tile)
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int KT = p.Kp / BK, kps = p.kt_per_split;
  const int nparts = (KT + kps - 1) / kps;             // <= 4
  const int kt0 = wave * kps;
  const int kt1 = min(KT, kt0 + kps);
  const int nt = max(0, kt1 - kt0);                    // this wave's K tiles

  // ---- epilogue operands that do not depend on the GEMM: issued first, they land under the K loop
  half8 rpre[NIT];
  const half8 z8 = (half8){0, 0, 0, 0, 0, 0, 0, 0};
  const int ec8 = (tid % CH) * 8, en = n0 + ec8;       // (the same columns in every chunk of a thread: 256 % CH == 0)
  const bool ncol = en < p.N;                          // N % 8 == 0 (host check): a chunk is inside or outside as a whole
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int r = (tid + j * 256) / CH, m = m0 + r;
    const bool ok = r < BM && m < p.M && ncol;
    rpre[j] = z8;
    if (p.residual) rpre[j] = *reinterpret_cast<const half8*>(p.residual + (ok ? (size_t)m * p.ldr + en : 0));
  }
  half8 braw = z8, rvraw = z8;
  if (ncol) {
    if (p.bias) braw = *reinterpret_cast<const half8*>(p.bias + en);
    if (p.rowvec) rvraw = *reinterpret_cast<const half8*>(p.rowvec + en);
  }

  // ---- operand fragments: lane (fr, fq) holds halfs [fq * 8, fq * 8 + 8) of row fr of each 16 x 32 block
  constexpr int OOB = (int)0x80000000;
  int va0[FM], va1[FM], vb[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + i * 16 + fr;
    va0[i] = m < p.M ? (m * p.c0 + fq * 8) * 2 : OOB;
    va1[i] = m < p.M ? (m * p.c1 + fq * 8) * 2 : OOB;
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = n0 + j * 16 + fr;
    vb[j] = n < p.N ? (n * p.Kp + fq * 8) * 2 : OOB;
  }
  const int anr0 = (int)((size_t)p.M * p.c0 * 2), anr1 = (int)((size_t)p.M * p.c1 * 2), bnr = (int)((size_t)p.N * p.Kp * 2);
  u32x4 ra[NT][2][FM], rb[NT][2][FN];
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

// fetch K tile KT_ of this wave's range into register slot S_ (descriptors made next to their use, as in conv_gemm_body.inc)
#define LEAN_ISSUE(S_, KT_)                                                                                          \
  {                                                                                                                  \
    const int kb_ = (KT_) * BK;                                                                                      \
    const bool second_ = kb_ >= p.c0;                                                                                \
    const int soa_ = (second_ ? kb_ - p.c0 : kb_) * 2, sob_ = kb_ * 2;                                               \
    const bool in_ = (KT_) < kt1; /* past the wave's range: empty descriptors, every lane reads zeros, nothing is fetched */ \
    const __amdgpu_buffer_rsrc_t rsa_ = __builtin_amdgcn_make_buffer_rsrc(                                           \
        (void*)(second_ ? p.src1 : p.src0), 0, in_ ? (second_ ? anr1 : anr0) : 0, 0x00020000);                       \
    const __amdgpu_buffer_rsrc_t rsb_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, in_ ? bnr : 0, 0x00020000); \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                               \
      _Pragma("unroll") for (int j = 0; j < FN; ++j)                                                                 \
        rb[S_][ks][j] = __builtin_amdgcn_raw_buffer_load_b128(rsb_, vb[j], sob_ + ks * 64, 0);                       \
      _Pragma("unroll") for (int i = 0; i < FM; ++i)                                                                 \
        ra[S_][ks][i] = __builtin_amdgcn_raw_buffer_load_b128(rsa_, second_ ? va1[i] : va0[i], soa_ + ks * 64, 0);   \
    }                                                                                                                \
  }
  // ALL of the wave's K tiles in flight at once (NT = kt_per_split is a template parameter: straight-line code, so the compiler's
  // own vmcnt before tile t's MFMAs is exactly "all but the NT - 1 - t younger tiles"; a runtime loop over register slots made it
  // wait for everything at the loop head -- seen in the ISA).  A wave whose range is shorter (the last part) reads its surplus
  // tiles through empty descriptors: zeros, nothing fetched, and the MFMAs add nothing.
#pragma unroll
  for (int t = 0; t < NT; ++t) LEAN_ISSUE(t, kt0 + t)
  __builtin_amdgcn_sched_barrier(0);  // (left alone, the scheduler sinks the later tiles' loads between the MFMAs to save registers)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ra[t][ks][i]),
                                                             __builtin_bit_cast(half8, rb[t][ks][j]), acc[i][j], 0, 0, 0);
  }
#undef LEAN_ISSUE

  // ---- the four partial tiles meet in LDS
  if (nt > 0) {
    float* Cs = parts + wave * PART;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = j * 16 + fr, row = i * 16 + fq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) Cs[(row + r) * BNP + col] = acc[i][j][r];
      }
  }
  __syncthreads();

  // ---- epilogue (the arithmetic of conv_epilogue.inc's straight-line walks, term by term)
  float brv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) brv[i] = (float)braw[i] + (float)rvraw[i];
  const float sc = p.out_scale_dev ? *p.out_scale_dev : p.out_scale;
  const int act = p.act & 0xff;
  const bool post = (p.act & VSD_ACT_POST) != 0;
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int r = (tid + j * 256) / CH, m = m0 + r;
    const bool valid = r < BM && m < p.M && ncol;
    float rs = 0.f, rq = 0.f;
    if (valid) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.f;
      const float* s = parts + r * BNP + ec8;
      for (int k = 0; k < nparts; ++k) {  // part order: the order of the split-K tails
        const f32x4 lo = *reinterpret_cast<const f32x4*>(s + k * PART), hi = *reinterpret_cast<const f32x4*>(s + k * PART + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] += lo[i];
          v[4 + i] += hi[i];
        }
      }
      half8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float x = v[i];
        x += brv[i];
        if (!post) {
          if (act == VSD_ACT_RELU) x = fmaxf(x, 0.f);
          else if (act == VSD_ACT_SILU) x = silu_f(x);
          else if (act == VSD_ACT_QUICKGELU) x = quick_gelu_f(x);
        }
        x *= sc;
        if (p.residual) x += (float)rpre[j][i];
        if (post) x = fmaxf(x, 0.f);  // (the one post-residual activation the host lets through: ReLU)
        o[i] = (half_t)x;
        const float f = (float)o[i];  // statistics of the STORED (rounded) values
        rs += f;
        rq += f * f;
      }
      *reinterpret_cast<half8*>(p.out + (size_t)m * p.ldo + en) = o;
    }
    if (CH == 8 && p.rowstat_out) {  // the 8 lanes of one (row, 64-column tile) are consecutive: fold, the first writes
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        rs += __shfl_xor(rs, o);
        rq += __shfl_xor(rq, o);
      }
      if ((tid & 7) == 0 && valid) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<f32x2*>(p.rowstat_out + ((size_t)m * (p.N >> 6) + (en >> 6)) * 2) = (f32x2){rs, rq};
      }
    }
  }
}

template <int FM, int FN, int NT>
__global__ __launch_bounds__(256) void conv_lean_kernel(const ConvParams p) {
  VSD_CUT(VSD_CUT_CONV_GEMM, p.cut)
  prefetch_kernargs();
  lean_body<FM, FN, NT>(p, (int)blockIdx.x);
}

// several independent problems in one grid (as conv_gemm_group_kernel): the twin layers of the two encoders
template <int FM, int FN, int NT>
__global__ __launch_bounds__(256) void conv_lean_group_kernel(const ConvGroup g) {
  int prob = 0;
#pragma unroll
  for (int i = 1; i < VSD_GROUP_MAX; ++i)
    if (i < g.n && (int)blockIdx.x >= g.start[i]) prob = i;
  const ConvParams& p = g.p[prob];
  VSD_CUT(VSD_CUT_CONV_GEMM, p.cut)
  lean_body<FM, FN, NT>(p, (int)blockIdx.x - g.start[prob]);
}

}  // namespace

// bm x bn: 16x64, 32x32 or 32x64, nt = kt_per_split (checked by the caller: vsd_lean_form_ok)
template <int FM, int FN>
static void lean_launch(const ConvParams& p, int nt, int grid, hipStream_t s) {
  switch (nt) {
    case 1: hipLaunchKernelGGL((conv_lean_kernel<FM, FN, 1>), dim3(grid), dim3(256), 0, s, p); break;
    case 2: hipLaunchKernelGGL((conv_lean_kernel<FM, FN, 2>), dim3(grid), dim3(256), 0, s, p); break;
    case 3: hipLaunchKernelGGL((conv_lean_kernel<FM, FN, 3>), dim3(grid), dim3(256), 0, s, p); break;
    case 4: hipLaunchKernelGGL((conv_lean_kernel<FM, FN, 4>), dim3(grid), dim3(256), 0, s, p); break;
    default: hipLaunchKernelGGL((conv_lean_kernel<FM, FN, 5>), dim3(grid), dim3(256), 0, s, p); break;
  }
}
template <int FM, int FN>
static void lean_launch_group(const ConvGroup& g, int nt, int grid, hipStream_t s) {
  switch (nt) {
    case 1: hipLaunchKernelGGL((conv_lean_group_kernel<FM, FN, 1>), dim3(grid), dim3(256), 0, s, g); break;
    case 2: hipLaunchKernelGGL((conv_lean_group_kernel<FM, FN, 2>), dim3(grid), dim3(256), 0, s, g); break;
    case 3: hipLaunchKernelGGL((conv_lean_group_kernel<FM, FN, 3>), dim3(grid), dim3(256), 0, s, g); break;
    case 4: hipLaunchKernelGGL((conv_lean_group_kernel<FM, FN, 4>), dim3(grid), dim3(256), 0, s, g); break;
    default: hipLaunchKernelGGL((conv_lean_group_kernel<FM, FN, 5>), dim3(grid), dim3(256), 0, s, g); break;
  }
}
// what exists: every tile with kt_per_split <= 5 (K <= 1280 in four parts): 160 / 200 / 240 registers of operands in flight per lane
bool vsd_lean_form_ok(int bm, int bn, int nt) {
  if (!((bm == 16 && bn == 64) || (bm == 32 && bn == 32) || (bm == 32 && bn == 64))) return false;
  return nt >= 1 && nt <= 5;
}
void vsd_launch_conv_lean(const ConvParams& p, int bm, int bn, int grid, hipStream_t s) {
  if (bm == 16 && bn == 64) lean_launch<1, 4>(p, p.kt_per_split, grid, s);
  else if (bm == 32 && bn == 32) lean_launch<2, 2>(p, p.kt_per_split, grid, s);
  else lean_launch<2, 4>(p, p.kt_per_split, grid, s);
}
// (every member has the kt_per_split of member 0: checked by the caller)
void vsd_launch_conv_lean_group(const ConvGroup& g, int bm, int bn, int grid, hipStream_t s) {
  const int nt = g.p[0].kt_per_split;
  if (bm == 16 && bn == 64) lean_launch_group<1, 4>(g, nt, grid, s);
  else if (bm == 32 && bn == 32) lean_launch_group<2, 2>(g, nt, grid, s);
  else lean_launch_group<2, 4>(g, nt, grid, s);
}

// Fused "transformer tail" for the 320-wide level of the SD1.5 UNet / ControlNet (gfx950, wave64, MFMA 16x16x32 f16).
//
// Behind the self-attention every operation of a BasicTransformerBlock is per token: out-projection + residual,
// LayerNorm, cross-attention query projection, [cross-attention over 77 text keys], out-projection + residual, LayerNorm,
// GEGLU feed-forward, + residual, and Transformer2DModel's proj_out + residual (diffusers under lcm_controlnet.py:568).
// As separate launches these are seven GEMMs of 2.5 - 20 GFLOP whose workgroups live ~8 us of prologue / epilogue latency
// around ~1 us of MFMA work (profiles/round1f_layer_table_batch3.txt: 12288x320x320 at 147 TFLOP/s).  Here ONE workgroup
// owns 64 tokens for a whole chain: the token tile stays in LDS / registers, only the weights stream (straight into
// registers as MFMA B fragments, a whole GEGLU chunk ahead per wave -- see load_b), and the intermediate tensors (h1, q,
// h2, the 1280-wide GEGLU hidden state, h3) never go to HBM.
//
//   vsd_tail_a:  att, h            -> h1 = att Wo1^T + b + h ;  q = LN(h1) Wq2'^T            (h1, q written: the 77-key
//                                     cross-attention between the two kernels is the existing vsd_attention)
//   vsd_tail_b:  att2, h1, x       -> h2 = att2 Wo2^T + b + h1 ; h3 = GEGLU-FF(LN(h2)) + h2 ; out = h3 Wp^T + b + x
//
// Layouts: wave w of 4 owns output columns [80w, 80w+80) of all 64 rows (4 x 5 accumulator fragments = 80 VGPRs);
// residual streams (h1, h2) stay in registers in that layout, fp32.  The A operand of every GEMM is the fp16 token tile
// X[64][328] in LDS (pitch 328 halfs: conflict-free ds_read_b128 fragment reads).  LayerNorms are the folded form of
// packing.pack_linear_ln (weights hold W*gamma; out = rstd (acc - mean s) + t) with the row statistics computed here from
// the fp16-rounded tile -- the same arithmetic as the unfused path (vsd_conv_gemm rowstat_out / ln_part).
// Bound: each workgroup streams the chain's weights (0.4 MB for tail_a, 2.9 MB for tail_b) from L2 at the per-CU fill
// rate (~70 GB/s, MI355X_MICROARCH.md) -- ~6 us and ~42 us; the MFMA work under it is ~3 / ~20 us.
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int C = 320;        // channel width this kernel is built for
constexpr int BM0 = 64;       // tokens per workgroup (tail_b also exists with 80: see vsd_tail_b)
constexpr int XP = 328;       // LDS pitch of the token tile (halfs)
constexpr int HP = 72;        // LDS pitch of the 64-wide GEGLU chunk (halfs)
constexpr int FF = 4 * C;     // GEGLU hidden width (1280)
constexpr int NCHUNK = FF / 64;

struct TailParams {
  int M;
  const half_t* a_in;   // [M][C] the chain's input tile (attention output)
  const half_t* res0;   // [M][C] residual of the first projection
  const half_t* res_x;  // [M][C] residual of proj_out (tail_b)
  // tail_a: out1 (w0, b0), q2 (w1 with ln_s1 / ln_t1);  tail_b: out2 (w0, b0), ff1 (w1, ln_s1 / ln_t1, tile packed), ff2 (w2, b2), proj_out (w3, b3)
  // every w*: fragment-major (see load_b)
  const half_t* w0; const half_t* b0;
  const half_t* w1; const float* ln_s1; const float* ln_t1;
  const half_t* w2; const half_t* b2;
  const half_t* w3; const half_t* b3;
  half_t* out0;         // tail_a: h1 [M][C];  tail_b: block output [M][C]
  half_t* out1;         // tail_a: q [M][C]
  float ln_eps;
  VSD_CUT_FIELD
};

// ---- weights: straight from global memory into registers, already in MFMA B-fragment layout.
// With the column split every wave needs rows of W nobody else in the workgroup needs, so staging them through LDS buys
// no reuse and costs a workgroup-wide ring, counted waits and a barrier per tile (first version of this kernel: one tile
// in flight per workgroup = one L2 round trip per tile, 2.9 MB in 130-170 us = 20 GB/s per CU).  A fragment of the
// 16x16x32 MFMA is 8 consecutive k of one row per lane: a 16-byte global load.  Each wave keeps a whole GEGLU chunk's
// worth (30 KB) of such loads in flight; the compiler counts the waits (plain loads), and there is no barrier in a GEMM.
template <int NF>
struct BFrag {
  half8 v[2][NF];  // [k-step of 32][fragment of 16 rows]
};

// W is stored FRAGMENT-MAJOR (packing.pack_mfma_frag): for every (16-row block nb, 32-column block kb) one 1 KB block
// [q = k/8][n][8 halfs], blocks ordered [nb][kb].  Lane l = n + 16 q of a fragment then reads bytes [16 l, 16 l + 16) of
// its block: one wave-instruction = one contiguous KB.  (In the plain [N][K] layout the 16 lanes of a quarter-wave read 16
// different rows: 64 cache-line accesses per instruction instead of 8 -- measured 25 GB/s per CU instead of ~70.)
// nb[j] = first row of fragment j (a multiple of 16), k0 a multiple of 64, kblocks = K / 32.
template <int NF>
__device__ __forceinline__ void load_b(BFrag<NF>& b, const half_t* wf, int kblocks, int k0, const int (&nb)[NF], int lane) {
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const half_t* blk = wf + ((size_t)(nb[j] >> 4) * kblocks + (k0 >> 5)) * 512 + lane * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b.v[ks][j] = *reinterpret_cast<const half8*>(blk + ks * 512);
  }
}

// Waves: WM row groups x 4 column groups.  A wave owns rows [32 wm .. ] (MI = 4 / WM accumulator row fragments) and
// columns [80 wn, 80 wn + 80).  WM = 2 (512 threads) puts two waves on every SIMD: one's GEGLU / epilogue VALU work runs
// under the other's MFMAs (with WM = 1 the single wave per SIMD serialises them: measured 80 vs ... us for tail_b); the two
// waves of a column group request the same weight fragments, the second request hits the CU's L1.  Measured (us, 12288 /
// 4096 tokens): tail_a 17.5 / 14.2 with WM = 2 against 19.4 / 16.6; tail_b 80 / 70 either way.
// (tail_a runs WM = 2; tail_b's feed-forward loop needs ~330 registers per wave and runs WM = 1)

// acc[mi][j] += A[rows of fragment mi][ka0 .. ka0 + 63] * B^T, A from the LDS tile `a` (pitch lda), rows from arow0
template <int MI>
struct AFrag {
  half8 v[2][MI];  // [k-step of 32][row fragment]
};

template <int MI>
__device__ __forceinline__ void load_a(AFrag<MI>& a, const half_t* x, int lda, int arow0, int ka0, int lane) {
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a.v[ks][mi] = *reinterpret_cast<const half8*>(x + (arow0 + mi * 16 + r) * lda + ka0 + ks * 32 + q * 8);
}

template <int MI, int NF>
__device__ __forceinline__ void mma_ab(const AFrag<MI>& a, const BFrag<NF>& b, f32x4 (&acc)[MI][NF]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v[ks][mi], b.v[ks][j], acc[mi][j], 0, 0, 0);
}

// acc[mi][j] += A[rows of fragment mi][ka0 .. ka0 + 63] * B^T, A from the LDS tile `a` (pitch lda), rows from arow0
template <int MI, int NF>
__device__ __forceinline__ void mma_b(const half_t* a, int lda, int arow0, int ka0, const BFrag<NF>& b, int lane, f32x4 (&acc)[MI][NF]) {
  AFrag<MI> af;
  load_a(af, a, lda, arow0, ka0, lane);
  mma_ab(af, b, acc);
}

// acc (rows of this wave x 80) = X[.][320] W[80 rows of this wave][320]^T: five K tiles, three fragment sets in flight
template <int MI>
__device__ __forceinline__ void gemm320(const half_t* X, int arow0, const half_t* w, int kblocks, int wn, int lane, f32x4 (&acc)[MI][5]) {
  const int nb[5] = {80 * wn, 80 * wn + 16, 80 * wn + 32, 80 * wn + 48, 80 * wn + 64};
  BFrag<5> b0, b1, b2;
  load_b(b0, w, kblocks, 0, nb, lane);
  load_b(b1, w, kblocks, 64, nb, lane);
  load_b(b2, w, kblocks, 128, nb, lane);
  // A fragments one K tile ahead of their MFMAs: the LDS latency (~130 cycles per fragment set) hides under a tile's MFMAs
  AFrag<MI> a0, a1;
  load_a(a0, X, XP, arow0, 0, lane);
  load_a(a1, X, XP, arow0, 64, lane);
  mma_ab(a0, b0, acc);
  load_b(b0, w, kblocks, 192, nb, lane);
  load_a(a0, X, XP, arow0, 128, lane);
  mma_ab(a1, b1, acc);
  load_b(b1, w, kblocks, 256, nb, lane);
  load_a(a1, X, XP, arow0, 192, lane);
  mma_ab(a0, b2, acc);
  load_a(a0, X, XP, arow0, 256, lane);
  mma_ab(a1, b0, acc);
  mma_ab(a0, b1, acc);
}

// Workgroup barrier for LDS data ONLY.  __syncthreads() also waits for every outstanding vector-memory operation
// (vmcnt(0)) -- here that would drain the weight fragments in flight for the next chunk.  The compiler still inserts the
// vmcnt waits the loaded registers need.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// accumulator element (mi, j, rr) of lane -> (row, col) of the 64 x 320 tile
#define TAIL_ROW(mi, rr) (arow0 + (mi) * 16 + 4 * (lane >> 4) + (rr))
#define TAIL_COL(j) (80 * wn + 16 * (j) + (lane & 15))

// residual tile in accumulator layout straight from global memory (2-byte loads)
template <int MI>
__device__ __forceinline__ void load_res(const half_t* g, int m0, int M, int arow0, int wn, int lane, half_t (&r)[MI][5][4]) {
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      int m = m0 + TAIL_ROW(mi, rr);
      m = m < M ? m : M - 1;
#pragma unroll
      for (int j = 0; j < 5; ++j) r[mi][j][rr] = g[(size_t)m * C + TAIL_COL(j)];
    }
}

// fp16 tile X <- v (rounded), and the per-row (mean, rstd) of the ROUNDED values -> ms[row][2]
template <int BM, int MI>
__device__ __forceinline__ void store_tile_and_stats(half_t* X, float* part, float* ms, float (&v)[MI][5][4], int arow0, int wn, int lane,
                                                     int tid, float eps, bool want_stats) {
  float s[MI][4], q2[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      float ss = 0.f, qq = 0.f;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const half_t hv = (half_t)v[mi][j][rr];
        X[TAIL_ROW(mi, rr) * XP + TAIL_COL(j)] = hv;
        const float f = (float)hv;
        ss += f;
        qq += f * f;
      }
      s[mi][rr] = ss;
      q2[mi][rr] = qq;
    }
  if (!want_stats) return;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        s[mi][rr] += __shfl_xor(s[mi][rr], o);
        q2[mi][rr] += __shfl_xor(q2[mi][rr], o);
      }
      if ((lane & 15) == 0) {
        part[(TAIL_ROW(mi, rr) * 4 + wn) * 2] = s[mi][rr];
        part[(TAIL_ROW(mi, rr) * 4 + wn) * 2 + 1] = q2[mi][rr];
      }
    }
  lds_barrier();
  if (tid < BM) {
    float S = 0.f, Q = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      S += part[(tid * 4 + w) * 2];
      Q += part[(tid * 4 + w) * 2 + 1];
    }
    const float mean = S * (1.0f / C);
    ms[2 * tid] = mean;
    ms[2 * tid + 1] = rsqrtf(fmaxf(Q * (1.0f / C) - mean * mean, 0.f) + eps);
  }
}

// rows of the LDS tile -> global, 16 bytes per lane
template <int NTH, int BM>
__device__ __forceinline__ void write_tile(const half_t* X, half_t* g, int m0, int M, int tid) {
  for (int q = tid; q < BM * (C / 8); q += NTH) {
    const int r = q / (C / 8), c8 = (q - r * (C / 8)) * 8;
    if (m0 + r < M) *reinterpret_cast<half8*>(g + (size_t)(m0 + r) * C + c8) = *reinterpret_cast<const half8*>(X + r * XP + c8);
  }
}

template <int NTH, int BM>
__device__ __forceinline__ void load_a_tile(const half_t* g, half_t* X, int m0, int M, int tid) {
  constexpr int TOTAL = BM * (C / 8), NV = (TOTAL + NTH - 1) / NTH;
  half8 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int q = tid + i * NTH;
    q = q < TOTAL ? q : TOTAL - 1;  // (a ragged last pass re-reads the last piece; it is not stored)
    const int r = q / (C / 8), c8 = (q - r * (C / 8)) * 8;
    int m = m0 + r;
    m = m < M ? m : M - 1;
    v[i] = *reinterpret_cast<const half8*>(g + (size_t)m * C + c8);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int q = tid + i * NTH;
    const int r = q / (C / 8), c8 = (q - r * (C / 8)) * 8;
    if (TOTAL % NTH == 0 || q < TOTAL) *reinterpret_cast<half8*>(X + r * XP + c8) = v[i];
  }
}

template <int KIND, int WM, int BM = BM0>
__device__ __forceinline__ void tail_body(const TailParams& p) {
  VSD_CUT(VSD_CUT_TAIL, p.cut)
  constexpr int MI = BM / 16 / WM, NTH = 256 * WM;
  static_assert(MI * 16 * WM == BM, "rows per workgroup = 16 x row fragments per wave x row groups");
  __shared__ __attribute__((aligned(16))) half_t Xs[BM * XP];
  __shared__ __attribute__((aligned(16))) half_t Hs[2 * BM * HP];
  __shared__ float part[BM * 4 * 2];
  __shared__ float ms[BM * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 3, arow0 = (wave >> 2) * (BM / WM);
  const int m0 = blockIdx.x * BM;

  // ---- everything the chain needs from HBM besides the weights is requested up front
  half_t res0[MI][5][4];
  load_res(p.res0, m0, p.M, arow0, wn, lane, res0);
  load_a_tile<NTH, BM>(p.a_in, Xs, m0, p.M, tid);
  float cb0[5], cb2[5], cb3[5], cs1[5], ct1[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    cb0[j] = (float)p.b0[TAIL_COL(j)];
    if constexpr (KIND == 0) {
      cs1[j] = p.ln_s1[TAIL_COL(j)];
      ct1[j] = p.ln_t1[TAIL_COL(j)];
    }
  }
  (void)cb2; (void)cb3; (void)cs1; (void)ct1;
  f32x4 acc[MI][5];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[mi][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  lds_barrier();  // the input tile is in LDS

  // ---- stage 1: h1 = a_in W0^T + b0 + res0
  zero_acc();
  gemm320(Xs, arow0, p.w0, C / 32, wn, lane, acc);
  float h1[MI][5][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) h1[mi][j][rr] = acc[mi][j][rr] + cb0[j] + (float)res0[mi][j][rr];
  lds_barrier();  // every wave has read the last A fragments of the input tile: X may be overwritten
  store_tile_and_stats<BM>(Xs, part, ms, h1, arow0, wn, lane, tid, p.ln_eps, true);
  lds_barrier();  // X (= h1 rounded to fp16) and the row statistics are published

  if constexpr (KIND == 0) {
    // ---- tail_a: q = LN(h1) W1'^T  (folded LayerNorm), h1 and q to HBM
    write_tile<NTH, BM>(Xs, p.out0, m0, p.M, tid);
    zero_acc();
    gemm320(Xs, arow0, p.w1, C / 32, wn, lane, acc);
    float qv[MI][5][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float mean = ms[2 * TAIL_ROW(mi, rr)], rstd = ms[2 * TAIL_ROW(mi, rr) + 1];
#pragma unroll
        for (int j = 0; j < 5; ++j) qv[mi][j][rr] = rstd * (acc[mi][j][rr] - mean * cs1[j]) + ct1[j];
      }
    lds_barrier();
    store_tile_and_stats<BM>(Xs, part, ms, qv, arow0, wn, lane, tid, p.ln_eps, false);
    lds_barrier();
    write_tile<NTH, BM>(Xs, p.out1, m0, p.M, tid);
    return;
  } else {
    // ---- tail_b: GEGLU feed-forward on LN(h2) (h2 = the tile just stored).
    // Per 64-wide hidden chunk c: S = X W1_c'^T (tile-packed rows: [0,64) hidden, [64,128) gate; column group wn takes
    // hidden columns [16 wn, 16 wn + 16) and their gates), GEGLU -> Hc (LDS), acc3 += Hc W2[:, 64c : 64c+64]^T.  The B
    // fragments (5 x 2 for W1, 1 x 5 for W2) have fixed registers that are re-loaded for a later chunk right after their
    // last use: more than a chunk of weights (30 KB per wave) is always in flight.
    f32x4 acc3[MI][5];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc3[mi][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nb2[5] = {80 * wn, 80 * wn + 16, 80 * wn + 32, 80 * wn + 48, 80 * wn + 64};
    BFrag<2> g1[5];
    BFrag<5> g2;
    float lnc[4];  // this chunk's LayerNorm-fold constants of the lane's hidden / gate column: s_h, t_h, s_g, t_g
    // Every workgroup streams the SAME 2.4 MB of feed-forward weights; walking the chunks in the same order makes all CUs
    // of an XCD ask its L2 for the same lines at the same time (each line lives in one L2 channel).  The chunk order is
    // therefore rotated per workgroup (the sum over chunks is order-independent up to fp32 rounding; a given workgroup
    // always uses the same order, so results stay deterministic).
    const int rot = blockIdx.x % NCHUNK;
    auto chunk_of = [&](int i) {
      i = i < NCHUNK ? i : NCHUNK - 1;
      const int c = i + rot;
      return c < NCHUNK ? c : c - NCHUNK;
    };
    auto load_chunk_w1 = [&](int c, int kt) {
      const int nb1[2] = {c * 128 + 16 * wn, c * 128 + 64 + 16 * wn};
      load_b(g1[kt], p.w1, C / 32, kt * 64, nb1, lane);
    };
    auto load_chunk_lnc = [&](int c) {
      const int nh = c * 128 + 16 * wn + (lane & 15);
      lnc[0] = p.ln_s1[nh]; lnc[1] = p.ln_t1[nh]; lnc[2] = p.ln_s1[nh + 64]; lnc[3] = p.ln_t1[nh + 64];
    };
#pragma unroll
    for (int kt = 0; kt < 5; ++kt) load_chunk_w1(chunk_of(0), kt);
    load_b(g2, p.w2, FF / 32, chunk_of(0) * 64, nb2, lane);
    load_chunk_lnc(chunk_of(0));
    f32x4 sh[MI][2];
    auto zero_sh = [&](f32x4 (&t)[MI][2]) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        t[mi][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        t[mi][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    };
    // t += X W1_c'^T for the chunk whose fragments are in g1 (re-loading them for chunk `cnext` as they are used); the A
    // fragments run one K tile ahead of their MFMAs
    auto scores = [&](f32x4 (&t)[MI][2], int cnext) {
      AFrag<MI> a0, a1;
      load_a(a0, Xs, XP, arow0, 0, lane);
      load_a(a1, Xs, XP, arow0, 64, lane);
      mma_ab(a0, g1[0], t);
      load_chunk_w1(cnext, 0);
      load_a(a0, Xs, XP, arow0, 128, lane);
      mma_ab(a1, g1[1], t);
      load_chunk_w1(cnext, 1);
      load_a(a1, Xs, XP, arow0, 192, lane);
      mma_ab(a0, g1[2], t);
      load_chunk_w1(cnext, 2);
      load_a(a0, Xs, XP, arow0, 256, lane);
      mma_ab(a1, g1[3], t);
      load_chunk_w1(cnext, 3);
      mma_ab(a0, g1[4], t);
      load_chunk_w1(cnext, 4);
    };
    auto geglu_to_lds = [&](f32x4 (&t)[MI][2], half_t* Hc, float s_h, float t_h, float s_g, float t_g) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = TAIL_ROW(mi, rr);
          const float mean = ms[2 * row], rstd = ms[2 * row + 1];
          const float hv = rstd * (t[mi][0][rr] - mean * s_h) + t_h;
          const float gv = rstd * (t[mi][1][rr] - mean * s_g) + t_g;
          Hc[row * HP + 16 * wn + (lane & 15)] = (half_t)(hv * gelu_erf_f(gv));
        }
    };
    // The GEGLU output is double buffered in LDS, so ONE barrier per chunk (Hc complete) is enough: a wave writing buffer
    // c & 1 is past the barrier of chunk c-1, which every wave reaches only after its reads of that buffer in chunk c-2.
    if constexpr (WM == 1 && BM == BM0) {
      // one wave per SIMD: software pipeline -- the scores of chunk c+1 (MFMA) sit in the same basic block as the GEGLU of
      // chunk c (VALU, transcendental) so that the two pipes overlap inside the wave
      zero_sh(sh);
      scores(sh, chunk_of(1));
      for (int c = 0; c < NCHUNK; ++c) {
        const int c2 = chunk_of(c + 2);  // (the last iterations re-load the last chunk: harmless)
        const int c1 = chunk_of(c + 1);
        const float s_h = lnc[0], t_h = lnc[1], s_g = lnc[2], t_g = lnc[3];
        load_chunk_lnc(c1);
        f32x4 sn[MI][2];
        zero_sh(sn);
        scores(sn, c2);
        half_t* Hc = Hs + (c & 1) * (BM * HP);
        geglu_to_lds(sh, Hc, s_h, t_h, s_g, t_g);
        lds_barrier();
        mma_b(Hc, HP, arow0, 0, g2, lane, acc3);
        load_b(g2, p.w2, FF / 32, c1 * 64, nb2, lane);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          sh[mi][0] = sn[mi][0];
          sh[mi][1] = sn[mi][1];
        }
      }
    } else {
      // two waves per SIMD: the partner wave fills the other pipe; the plain order keeps the register count at 256
      // (also the taller tiles: their accumulators leave no room for the second score set)
      for (int c = 0; c < NCHUNK; ++c) {
        const int c1 = chunk_of(c + 1);
        zero_sh(sh);
        scores(sh, c1);
        const float s_h = lnc[0], t_h = lnc[1], s_g = lnc[2], t_g = lnc[3];
        load_chunk_lnc(c1);
        half_t* Hc = Hs + (c & 1) * (BM * HP);
        geglu_to_lds(sh, Hc, s_h, t_h, s_g, t_g);
        lds_barrier();
        mma_b(Hc, HP, arow0, 0, g2, lane, acc3);
        load_b(g2, p.w2, FF / 32, c1 * 64, nb2, lane);
      }
    }
    // h3 = acc3 + b2 + h2 -> X.  The residual h2 is read back from the token tile (fp16, as the unfused path reads it):
    // its fp32 registers are not held across the feed-forward loop.
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      cb2[j] = (float)p.b2[TAIL_COL(j)];
      cb3[j] = (float)p.b3[TAIL_COL(j)];
    }
    float h3[MI][5][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) h3[mi][j][rr] = acc3[mi][j][rr] + cb2[j] + (float)Xs[TAIL_ROW(mi, rr) * XP + TAIL_COL(j)];
    lds_barrier();  // every wave has its residual values (and is done with X as the scores' A operand)
    store_tile_and_stats<BM>(Xs, part, ms, h3, arow0, wn, lane, tid, p.ln_eps, false);
    lds_barrier();
    // ---- proj_out: out = h3 W3^T + b3 + x   (x requested here: its latency hides under the GEMM)
    half_t resx[MI][5][4];
    load_res(p.res_x, m0, p.M, arow0, wn, lane, resx);
    zero_acc();
    gemm320(Xs, arow0, p.w3, C / 32, wn, lane, acc);
    float ov[MI][5][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) ov[mi][j][rr] = acc[mi][j][rr] + cb3[j] + (float)resx[mi][j][rr];
    lds_barrier();
    store_tile_and_stats<BM>(Xs, part, ms, ov, arow0, wn, lane, tid, p.ln_eps, false);
    lds_barrier();
    write_tile<NTH, BM>(Xs, p.out0, m0, p.M, tid);
  }
}

template <int KIND, int WM, int BM = BM0>
__global__ __launch_bounds__(256 * WM) void tail_kernel(const TailParams p) {
  tail_body<KIND, WM, BM>(p);
}
// two token sets with their own weights as one grid (common.h launch_pairable; blockIdx.z = which)
template <int KIND, int WM, int BM = BM0>
__global__ __launch_bounds__(256 * WM) void tail_pair_kernel(const Pair<TailParams> g) {
  tail_body<KIND, WM, BM>(g.p[blockIdx.z]);
}

}  // namespace

// Tokens per workgroup.  tail_b runs one workgroup per CU (330+ registers per lane), so a launch runs in rounds of 256
// workgroups, and a round costs t0 + t1 x rows: every workgroup streams the chain's 2.9 MB of weights whatever its height,
// the MFMA / VALU work grows with the rows.  Measured on MI355X (us per round, scripts/tail_bm_bench.sh): 16 rows 44,
// 32: 49, 48: 60, 64: 74 (the software-pipelined form; the others use the plain loop), 80: 89.  4096 tokens as 64 tiles
// of 64 rows used to take 72 us on a quarter of the chip (and lost to the four unfused launches, 59 us); as 256 tiles of
// 16 rows they take 44; 12288 tokens: 61 us on 48-row tiles against 74; 20480: 91 us on 80-row tiles against 145.
// tail_a (0.4 MB of weights, four-wave forms fit two per CU): 16-row tiles up to 4096 tokens (9.8 against 15 us), 32 above.
// VSD_TAIL_BM forces a height (benchmarking).
static int pick_tail_bm(int m, int kind) {
  static const int force = getenv("VSD_TAIL_BM") ? atoi(getenv("VSD_TAIL_BM")) : 0;
  if (force == 16 || force == 32 || force == 48 || force == 64 || force == 80) return force;
  if (kind == 0) return cdiv(m, 16) <= 256 ? 16 : 32;
  static const int cand[5] = {16, 32, 48, 64, 80};
  static const double round_us[5] = {44.0, 49.0, 60.0, 74.0, 89.0};
  int best = 64;
  double best_t = 1e30;
  for (int i = 0; i < 5; ++i) {
    const double t = (double)cdiv(cdiv(m, cand[i]), 256) * round_us[i];
    if (t < best_t - 1e-9) {
      best_t = t;
      best = cand[i];
    }
  }
  return best;
}

// tokens x 320: out-projection + residual, LayerNorm (folded), query projection of the cross-attention
extern "C" int vsd_tail_a(vsd_ctx* ctx, const void* att, const void* h, int m, const void* w_out, const void* b_out,
                          const void* w_q, const void* ln_s, const void* ln_t, float ln_eps, void* h1_out, void* q_out,
                          void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!att || !h || !w_out || !b_out || !w_q || !ln_s || !ln_t || !h1_out || !q_out || m <= 0)
    return vsd_fail(ctx, VSD_ERR_ARG, "tail_a: bad arguments");
  TailParams p = {};
  p.M = m; p.a_in = (const half_t*)att; p.res0 = (const half_t*)h;
  p.w0 = (const half_t*)w_out; p.b0 = (const half_t*)b_out;
  p.w1 = (const half_t*)w_q; p.ln_s1 = (const float*)ln_s; p.ln_t1 = (const float*)ln_t;
  p.out0 = (half_t*)h1_out; p.out1 = (half_t*)q_out; p.ln_eps = ln_eps;
  VSD_CUT_SET(p)
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls(ctx, s, VSD_FAM_CONV_GEMM, 2.0 * m * 2.0 * C * C);
  // tile height: see pick_tail_bm (64 = the eight-wave form; the others run four waves)
  switch (pick_tail_bm(m, 0)) {
    case 16: launch_pairable(ctx, tail_kernel<0, 1, 16>, tail_pair_kernel<0, 1, 16>, dim3(cdiv(m, 16)), dim3(256), 0, s, p); break;
    case 32: launch_pairable(ctx, tail_kernel<0, 1, 32>, tail_pair_kernel<0, 1, 32>, dim3(cdiv(m, 32)), dim3(256), 0, s, p); break;
    case 48: launch_pairable(ctx, tail_kernel<0, 1, 48>, tail_pair_kernel<0, 1, 48>, dim3(cdiv(m, 48)), dim3(256), 0, s, p); break;
    case 80: launch_pairable(ctx, tail_kernel<0, 1, 80>, tail_pair_kernel<0, 1, 80>, dim3(cdiv(m, 80)), dim3(256), 0, s, p); break;
    default: launch_pairable(ctx, tail_kernel<0, 2>, tail_pair_kernel<0, 2>, dim3(cdiv(m, BM0)), dim3(512), 0, s, p);
  }
  return ls.finish();
}

// tokens x 320: out-projection + residual, LayerNorm (folded), GEGLU feed-forward + residual, proj_out + residual
extern "C" int vsd_tail_b(vsd_ctx* ctx, const void* att2, const void* h1, const void* x, int m, const void* w_out, const void* b_out,
                          const void* w_ff1, const void* ln_s, const void* ln_t, float ln_eps, const void* w_ff2, const void* b_ff2,
                          const void* w_proj, const void* b_proj, void* out, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!att2 || !h1 || !x || !w_out || !b_out || !w_ff1 || !ln_s || !ln_t || !w_ff2 || !b_ff2 || !w_proj || !b_proj || !out || m <= 0)
    return vsd_fail(ctx, VSD_ERR_ARG, "tail_b: bad arguments");
  TailParams p = {};
  p.M = m; p.a_in = (const half_t*)att2; p.res0 = (const half_t*)h1; p.res_x = (const half_t*)x;
  p.w0 = (const half_t*)w_out; p.b0 = (const half_t*)b_out;
  p.w1 = (const half_t*)w_ff1; p.ln_s1 = (const float*)ln_s; p.ln_t1 = (const float*)ln_t;
  p.w2 = (const half_t*)w_ff2; p.b2 = (const half_t*)b_ff2;
  p.w3 = (const half_t*)w_proj; p.b3 = (const half_t*)b_proj;
  p.out0 = (half_t*)out; p.ln_eps = ln_eps;
  VSD_CUT_SET(p)
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls(ctx, s, VSD_FAM_CONV_GEMM, 2.0 * m * (2.0 * C * C + 3.0 * C * FF));
  switch (pick_tail_bm(m, 1)) {
    case 16: launch_pairable(ctx, tail_kernel<1, 1, 16>, tail_pair_kernel<1, 1, 16>, dim3(cdiv(m, 16)), dim3(256), 0, s, p); break;
    case 32: launch_pairable(ctx, tail_kernel<1, 1, 32>, tail_pair_kernel<1, 1, 32>, dim3(cdiv(m, 32)), dim3(256), 0, s, p); break;
    case 48: launch_pairable(ctx, tail_kernel<1, 1, 48>, tail_pair_kernel<1, 1, 48>, dim3(cdiv(m, 48)), dim3(256), 0, s, p); break;
    case 80: launch_pairable(ctx, tail_kernel<1, 1, 80>, tail_pair_kernel<1, 1, 80>, dim3(cdiv(m, 80)), dim3(256), 0, s, p); break;
    default: launch_pairable(ctx, tail_kernel<1, 1>, tail_pair_kernel<1, 1>, dim3(cdiv(m, BM0)), dim3(256), 0, s, p);
  }
  return ls.finish();
}

// Fused attention forward softmax(Q K^T * scale) V for gfx950 (wave64, v_mfma_f32_32x32x16_f16).
//
// One wave owns QB blocks of 32 queries of one head (of one image) and walks the keys in tiles of 64 staged in LDS.
//   S^T = K Q^T  (swapped operands): the accumulator has the QUERY on the lane and 16 keys in the
//   registers, so the row max / row sum are in-lane reductions plus one cross-half swap, and the
//   running max / sum / rescale factor are per-lane scalars.
//   O^T += V^T P^T: the S^T accumulator, converted to fp16 pairs, IS the B operand of the second
//   MFMA (k index = key, permuted inside each 16-key step exactly as the accumulator rows are), so P
//   never touches LDS.  V arrives pre-transposed ([d][keys], written by the QKV GEMM epilogue); its LDS image
//   stores the keys of every 16-key step in that same permuted order so that a fragment is ONE ds_read_b128.
// What bounds it at head_dim 40..80 (in-kernel cycle probe, scripts/attn_probe.cpp): per 32x64 score block a wave spends
// 14-18 MFMAs (~0.45k cycles) but ~150 VALU issues for the softmax (max, sub, exp2, cvt: ~0.8k cycles with the MFMA
// drain), plus ~0.3k each for issuing the next tile's fetch and parking it in LDS.  Hence:
//   KSP = 2  a workgroup holds two wave groups that take alternate key tiles of the SAME queries and merge their
//            (max, sum, O) at the end: two waves per SIMD even when the grid is only one workgroup per CU, so one
//            wave's softmax VALU runs under the other's MFMAs / LDS waits (4096 keys, d = 40: 87 -> 55 us);
//   tiles are fetched two ahead through two register sets (one ahead exposed an L2-miss latency per tile), from
//   wave-uniform tile bases, and never through a select on the loaded value;
//   the row sums come out of the PV MFMA (a row of ones in the V^T padding) when head_dim leaves a padded row;
//   the O rescale is skipped when no query of the wave raised its running maximum;
//   QB = 2 (two query blocks per wave sharing every K / V^T fragment read) is implemented but not instantiated: it
//   measured 1.3x slower everywhere (> 256 VGPRs -> one wave per SIMD; the kernel is issue-bound, not LDS-bound).
// head_dim d (multiple of 8, <= 160) is zero-padded to NQK*16 for QK^T and NPV*32 for PV.
// Algorithmic FLOPs per launch: 4*sq*sk*heads*d (per image).
#include <stdarg.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

struct AttnParams {
  const half_t* q;
  const half_t* k;
  const half_t* vt;
  half_t* out;
  int ldq, ldk, ldvt, ldo;
  int sq, sk, heads, d;
  float scale_log2;  // scale * log2(e)
  int causal;
  int k_brows, vt_bcols;  // batch (blockIdx.z): image b reads K rows from b*k_brows and V^T columns from b*vt_bcols
  VSD_CUT_FIELD
#ifdef VSD_ATTN_PROBE
  long long* probe;  // scripts/attn_probe.cpp: per-section shader-clock totals of wave 0 of workgroup 0
#endif
};

#ifdef VSD_ATTN_PROBE
long long* g_probe = nullptr;
#define PROBE(I_)                                          \
  {                                                        \
    long long t_ = __builtin_readcyclecounter();           \
    pacc[I_] += t_ - plast;                                \
    plast = t_;                                            \
  }
#else
#define PROBE(I_)
#endif

constexpr float NEG_BIG = -1.0e30f;

#ifndef VSD_ATTN_PV_PRIO
#define VSD_ATTN_PV_PRIO 3
#endif

__device__ __forceinline__ float xhalf_max(float x) {  // max over the two half-waves (lanes l and l ^ 32)
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// (Round 4's two experiment bodies -- the PV MFMAs woven into the exponentials, -DVSD_ATTN_WEAVE, and a forced fourth wave per SIMD,
//  -DVSD_ATTN_WAVES -- measured -2 ... -5 % / +60 % and left this file in round 5: docs/dropped_experiments.patch (attention_weave.inc) keeps the woven
//  body, profiles/round4_attention_weave.txt the ISA listing and the numbers.  Round 5 looked at the VALU stream itself
//  (profiles/round5_attention_valu.txt): per 64-key tile a wave issues 33 v_sub + 33 v_exp + 17 v_max3 + 16 v_cvt_pk; the scale is
//  already folded into Q, the row sums already come out of the PV MFMA (the ones row below), O is already rescaled only when a
//  maximum moved; the max-subtract as 16 v_pk_add_f32 measured 9-34 % SLOWER (the compiler no longer shares the two row-sum
//  branches' code, 174 registers) and was not kept.)
// LAZY (round 6; head_dim % 16 == 8, i.e. d = 40 -- the 64 x 64 level's 4096-key self-attention -- and not causal): the running maximum
// is subtracted INSIDE the QK^T MFMA.  The last K step of Q / K has eight padding columns; column d of every key is 1.0 and column d of
// query q holds -m_off[q] (fp16), so the score tile arrives as S - m_off and the 33 v_sub per 64-key tile -- a quarter of the softmax's
// VALU stream, which is what this kernel binds on -- are gone.  m_off follows the true maximum lazily: it moves (O rescaled, the pad
// column rewritten) only when a tile's maximum exceeds it by more than LAZY_T in log2 units, so P stays in [0, 2^LAZY_T] and the largest
// P of a row in [1, 2^LAZY_T]; the row sum comes out of the PV MFMA's ones row, in the same units.  Other rounding than the plain
// form (an offset that is not the exact maximum), the same tolerance.
constexpr float LAZY_T = 5.0f;
template <int NQK, int NPV, int NW, int QB, int KSP, bool LAZY = false>
__device__ __forceinline__ void attention_body(const AttnParams& pp, const unsigned image) {
  VSD_CUT(VSD_CUT_ATTENTION, pp.cut)
  constexpr int NTG = 64 * NW;       // threads of one key-split group (they stage that group's tiles)
  constexpr int KS = NQK * 16 + 8;   // K tile row pitch (halfs): 4 * odd dwords -> conflict-free ds_read_b128
  constexpr int VS = 64 + 8;         // V^T tile row pitch (halfs): 36 dwords, same property
  constexpr int DV = NPV * 32;
  constexpr int KTILE = 64 * KS, VTILE = DV * VS;
  constexpr int GROUP_HALFS = 2 * (KTILE + VTILE);   // double buffered
  constexpr int KCH = (64 * NQK * 2 + NTG - 1) / NTG;  // 16-byte chunks per thread per K tile (upper bound)
  constexpr int VCH = (DV * 8 + NTG - 1) / NTG;        // ... per V^T tile
  constexpr int MERGE_FLOATS = KSP > 1 ? NW * QB * (NPV * 16 + 2) * 64 : 0;
  constexpr int SMEM_HALFS = (KSP * GROUP_HALFS > 2 * MERGE_FLOATS) ? KSP * GROUP_HALFS : 2 * MERGE_FLOATS;
  __shared__ __attribute__((aligned(16))) half_t smem[SMEM_HALFS];

  AttnParams p = pp;
  {  // batch: this workgroup's image
    const size_t b = image;
    p.q += b * p.sq * p.ldq;
    p.out += b * p.sq * p.ldo;
    p.k += b * p.k_brows * p.ldk;
    p.vt += b * p.vt_bcols;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int grp = wave / NW;          // key-split group: takes key tiles grp, grp + KSP, ...
  const int gwave = wave - grp * NW;  // wave within the group <-> query blocks
  const int gtid = tid - grp * NTG;
  const int lr = lane & 31;
  const int lh = lane >> 5;
  // blockIdx.x = head: consecutive workgroup ids go to consecutive XCDs, so with 8 heads every XCD works on ONE head
  // and that head's K / V^T (0.66 MB at 4096 keys, d = 40) stay in its 4 MB L2 instead of all heads thrashing all L2s
  const int head = blockIdx.x;
  const int qblk = blockIdx.y;
  const int q0 = qblk * (32 * NW * QB) + gwave * (32 * QB);
  const int d = p.d;
  const int dch = d >> 3;  // 16-byte chunks per row
  half_t* Ks = smem + grp * GROUP_HALFS;
  half_t* Vs = Ks + 2 * KTILE;

  // zero the LDS once (padding columns / rows stay zero for the whole kernel)
  for (int i = tid; i < KSP * GROUP_HALFS / 8; i += NTG * KSP) reinterpret_cast<u32x4*>(smem)[i] = (u32x4){0u, 0u, 0u, 0u};
  // Row sums for free: when head_dim leaves a padded V^T row (d < DV), the LAST padded row is set to 1.0, so the PV
  // MFMA accumulates sum_k P[q][k] (of the same fp16-rounded P that multiplies V) into O^T row DV-1 -- no VALU adds,
  // and the online rescale of O rescales it too.
  const bool ones_row = d < DV;

  // ---- per-thread staging coordinates (fixed for the whole kernel)
  unsigned kg[KCH];
  int kl[KCH], krow[KCH];
#pragma unroll
  for (int i = 0; i < KCH; ++i) {
    int c = gtid + i * NTG;
    bool ok = c < 64 * dch;
    int kr = ok ? c / dch : 0, kc = ok ? c - kr * dch : 0;
    krow[i] = ok ? kr : 1 << 28;  // never < sk
    kg[i] = ok ? (unsigned)(kr * p.ldk + head * d + kc * 8) : 0u;  // (idle threads load a valid address they never store)
    kl[i] = kr * KS + kc * 8;
  }
  unsigned vg[VCH];
  int vl[VCH];
  bool vok[VCH];
#pragma unroll
  for (int i = 0; i < VCH; ++i) {
    int c = gtid + i * NTG;
    vok[i] = c < d * 8;
    int vr = vok[i] ? c >> 3 : 0, vc = c & 7;
    vg[i] = vok[i] ? (unsigned)((head * d + vr) * p.ldvt + vc * 8) : 0u;
    // keys 8vc..8vc+7 of the tile; inside every 16-key step the LDS order is [0-3, 8-11, 4-7, 12-15] (the order in
    // which the S^T accumulator registers hold the keys): first half -> +0 (vc even) / +4 (odd), second -> +8 / +12
    vl[i] = vr * VS + (vc >> 1) * 16 + (vc & 1) * 4;
  }
  // Two register sets: the tile after next is already in flight while the next one waits in registers for its LDS
  // slot.  One tile of look-ahead is not enough here: a tile's math is ~1k cycles, a K / V^T fetch that misses the
  // XCD's L2 takes 2-4k (the kernel was running at that latency per tile).
  u32x4 kreg[2][KCH], vreg[2][VCH];
// The tile's base addresses are wave-uniform (scalar); each thread adds a fixed unsigned 32-bit offset, so a load costs
// no per-tile address arithmetic.  Rows past the last key (ragged last tile only) re-read row 0 (a valid address), NOT
// zeroed by a select: a select on the loaded value makes the wave wait for the load on the spot (a whole memory latency
// per tile); those keys are masked to -inf in the score tile anyway.
#define ATT_LOAD_K(SET_, KEY0_)                                                                     \
  {                                                                                                 \
    const half_t* kb_ = p.k + (size_t)(KEY0_) * p.ldk;                                              \
    if ((KEY0_) + 64 <= p.sk) {                                                                     \
      _Pragma("unroll") for (int i = 0; i < KCH; ++i) kreg[SET_][i] = *reinterpret_cast<const u32x4*>(kb_ + kg[i]); \
    } else {                                                                                        \
      _Pragma("unroll") for (int i = 0; i < KCH; ++i) {                                             \
        const bool ok = (KEY0_) + krow[i] < p.sk;                                                   \
        kreg[SET_][i] = *reinterpret_cast<const u32x4*>(ok ? kb_ + kg[i] : p.k);                    \
      }                                                                                             \
    }                                                                                               \
  }
#define ATT_LOAD_V(SET_, KEY0_)                                                                     \
  {                                                                                                 \
    const half_t* vb_ = p.vt + (KEY0_);                                                             \
    _Pragma("unroll") for (int i = 0; i < VCH; ++i) vreg[SET_][i] = *reinterpret_cast<const u32x4*>(vb_ + vg[i]); \
  }
#define ATT_STORE_K(SET_, BUF_)                                                                     \
  {                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < KCH; ++i)                                                 \
      if (krow[i] < 64) *reinterpret_cast<u32x4*>(Ks + (BUF_) * KTILE + kl[i]) = kreg[SET_][i];     \
  }
#define ATT_STORE_V(SET_, BUF_)                                                                     \
  {                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < VCH; ++i)                                                 \
      if (vok[i]) {                                                                                 \
        half_t* dst = Vs + (BUF_) * VTILE + vl[i];                                                  \
        *reinterpret_cast<u32x2*>(dst) = (u32x2){vreg[SET_][i][0], vreg[SET_][i][1]};               \
        *reinterpret_cast<u32x2*>(dst + 8) = (u32x2){vreg[SET_][i][2], vreg[SET_][i][3]};           \
      }                                                                                             \
  }
#define ATT_LOAD(SET_, KEY0_) { ATT_LOAD_K(SET_, KEY0_) ATT_LOAD_V(SET_, KEY0_) }
#define ATT_STORE(SET_, BUF_) { ATT_STORE_K(SET_, BUF_) ATT_STORE_V(SET_, BUF_) }

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + 32 qb + lr][ks*16 + lh*8 .. +8], pre-scaled
  half8 qf[QB][NQK];
  int qrow[QB];
  bool qvalid[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    qrow[qb] = q0 + 32 * qb + lr;
    qvalid[qb] = qrow[qb] < p.sq;
#pragma unroll
    for (int ks = 0; ks < NQK; ++ks) {
      int doff = ks * 16 + lh * 8;
      half8 v = (half8){0, 0, 0, 0, 0, 0, 0, 0};
      if (qvalid[qb] && doff < d) {
        v = *reinterpret_cast<const half8*>(p.q + (size_t)qrow[qb] * p.ldq + head * d + doff);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (half_t)((float)v[j] * p.scale_log2);
      }
      qf[qb][ks] = v;
    }
  }

  f32x16 o[QB][NPV];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = NEG_BIG;
    l_run[qb] = 0.f;
#pragma unroll
    for (int i = 0; i < NPV; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][i][r] = 0.f;
  }

  [[maybe_unused]] bool lazy_first = true;  // LAZY: m_run is the offset in the Q pad column (fp16-representable; zero until the first tile)
  if constexpr (LAZY) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) m_run[qb] = 0.f;
  }
  int sk_end = p.sk;
  if (p.causal) sk_end = min(p.sk, qblk * (32 * NW * QB) + 32 * NW * QB);  // later keys are never visible
  const int ntiles = (sk_end + 63) / 64;
  const int niter = (ntiles + KSP - 1) / KSP;

  // this group's tiles are grp, grp + KSP, ...; "slot j" = its j-th tile, living in register set j & 1 / LDS buffer j & 1
  auto slot_key0 = [&](int j) { return (j * KSP + grp) * 64; };
  auto slot_ok = [&](int j) { return j * KSP + grp < ntiles; };
#ifdef VSD_ATTN_PROBE
  long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long plast = __builtin_readcyclecounter();
#endif
  // ---- S^T = K Q^T of the tile in LDS buffer `buf`: two independent 32-key accumulation chains, interleaved (each K
  // fragment feeds all QB blocks)
  auto qk_tile = [&](const int buf, f32x16 (&s)[QB][2]) __attribute__((always_inline)) {
      const half_t* Kb = Ks + buf * KTILE;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[qb][kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < NQK; ++ks) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          half8 kf = *reinterpret_cast<const half8*>(Kb + (kb * 32 + lr) * KS + ks * 16 + lh * 8);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) s[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qb][ks], s[qb][kb], 0, 0, 0);
        }
      }
      PROBE(1)
      // Issue priority: LOW while this wave runs its softmax VALU stream, HIGH from the PV MFMAs through the next tile's QK
      // MFMAs.  A SIMD holds two or three of these waves in different phases; with equal priorities a wave's MFMAs wait their
      // turn behind the other waves' VALU instructions and the matrix pipe drains between them.  Measured on MI355X
      // (scripts/attn_bench.py, same box): 4096 keys d = 40, 5 images 243.9 -> 228.2 us, 3 images 128.2 -> 123.5, 1024 keys d = 80
      // 32.5 -> 30.9; the opposite assignment (softmax high) 248.6 / 123.1 / 32.5.  Bit-identical (scheduling only).
      __builtin_amdgcn_s_setprio(0);
  };
  // ---- masking, online softmax of the score tile `s` (keys key0..key0+63), O^T += V^T P^T with V^T in LDS buffer `buf`
  auto softmax_pv = [&](const int buf, const int key0, auto masked, f32x16 (&s)[QB][2]) __attribute__((always_inline)) {
      const half_t* Vb = Vs + buf * VTILE;
      // ---- masking (tail keys / causal)
      if constexpr (decltype(masked)::value) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
              bool ok = key < p.sk && (!p.causal || key <= qrow[qb]);
              if (!ok) s[qb][kb][r] = NEG_BIG;
            }
      }
      if constexpr (LAZY) {
        // ---- the score tile is S - m_off already: exponentials straight away unless some query's maximum outgrew its offset
        float mx[QB];
        bool need = false;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          float v = NEG_BIG;
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) v = fmaxf(v, s[qb][kb][r]);
          mx[qb] = xhalf_max(v);
          need = need || mx[qb] > LAZY_T;
        }
        const bool moved = __builtin_amdgcn_ballot_w64(need) != 0;
        if (lazy_first || moved) {
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
            const bool mine = lazy_first || mx[qb] > LAZY_T;
            float m_new = mine ? fminf(fmaxf(m_run[qb] + mx[qb], -60000.f), 60000.f) : m_run[qb];
            m_new = (float)(half_t)m_new;            // what the pad column can hold
            const float delta = m_new - m_run[qb];   // (both fp16 values: exact)
            const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
              for (int r = 0; r < 16; ++r) s[qb][kb][r] = __builtin_amdgcn_exp2f(s[qb][kb][r] - delta);
            if (!lazy_first) {  // (the first tile: O is zero and its offset was zero)
#pragma unroll
              for (int i = 0; i < NPV; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qb][i][r] *= alpha;
            }
            m_run[qb] = m_new;
            if (lh == 1) qf[qb][NQK - 1][0] = (half_t)(-m_new);  // column d of this query (upper half of the last K step)
          }
          lazy_first = false;
        } else {
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
              for (int r = 0; r < 16; ++r) s[qb][kb][r] = __builtin_amdgcn_exp2f(s[qb][kb][r]);
        }
      } else {
      // ---- online softmax (per lane = per query; the two half-waves hold different keys of the same query)
      bool grew_any = false;
      float alpha[QB];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float mx = NEG_BIG;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][kb][r]);
        mx = xhalf_max(mx);
        const float m_new = fmaxf(m_run[qb], mx);
        grew_any = grew_any || (m_new > m_run[qb]);
        alpha[qb] = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
        m_run[qb] = m_new;
        if (ones_row) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[qb][kb][r] = __builtin_amdgcn_exp2f(s[qb][kb][r] - m_new);
        } else {
          float psum = 0.f;
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float e = __builtin_amdgcn_exp2f(s[qb][kb][r] - m_new);
              s[qb][kb][r] = e;
              psum += e;
            }
          l_run[qb] = l_run[qb] * alpha[qb] + psum;
        }
      }
      // the running maximum settles after the first tiles: rescale O only when some query of this wave moved it
      // (alpha == 1 exactly for the others, so skipping is bit-identical)
      if (__builtin_amdgcn_ballot_w64(grew_any) != 0) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int i = 0; i < NPV; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][i][r] *= alpha[qb];
      }
      }  // (!LAZY)

      PROBE(2)
      __builtin_amdgcn_s_setprio(VSD_ATTN_PV_PRIO);
      // ---- O^T += V^T P^T (each V^T fragment feeds all QB query blocks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          half8 pf[QB];
#pragma unroll
          for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[qb][j] = (half_t)s[qb][kb][8 * st + j];
#pragma unroll
          for (int db = 0; db < NPV; ++db) {
            half8 vf = *reinterpret_cast<const half8*>(Vb + (db * 32 + lr) * VS + kb * 32 + 16 * st + 8 * lh);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[qb], o[qb][db], 0, 0, 0);
          }
        }
      }
      PROBE(3)
  };
  // two copies of the softmax body: the masked one only runs for a ragged last tile (or causal attention) -- as ONE
  // body the compiler if-converts the mask into ~300 selects that every tile would execute
  auto run_softmax_pv = [&](int j, const int buf, f32x16 (&s)[QB][2]) __attribute__((always_inline)) {
    const int key0 = slot_key0(j);
    if ((key0 + 64 > p.sk) || p.causal) softmax_pv(buf, key0, std::true_type{}, s);
    else softmax_pv(buf, key0, std::false_type{}, s);
  };

  if (slot_ok(0)) ATT_LOAD(0, slot_key0(0))
  if (slot_ok(1)) ATT_LOAD(1, slot_key0(1))
  __syncthreads();  // zero fill complete before the first tile lands on top of it
  if (ones_row && gtid < 2 * 64) Vs[(gtid >> 6) * VTILE + (DV - 1) * VS + (gtid & 63)] = (half_t)1.0f;
  if (LAZY && gtid < 2 * 64) Ks[(gtid >> 6) * KTILE + (gtid & 63) * KS + d] = (half_t)1.0f;  // column d of every key (the staging stores write columns < d)
  if (slot_ok(0)) ATT_STORE(0, 0)
  __syncthreads();
  auto run_tile = [&](int j) __attribute__((always_inline)) {
    f32x16 s[QB][2];
    qk_tile(j & 1, s);
    run_softmax_pv(j, j & 1, s);
  };
  for (int it = 0; it < niter; it += 2) {
    // even slot `it` (LDS buffer 0, its registers -- set 0 -- are free again): fetch slot it + 2, compute, park slot it + 1
    if (slot_ok(it + 2)) ATT_LOAD(0, slot_key0(it + 2))
    PROBE(0)
    if (slot_ok(it)) run_tile(it);
    if (slot_ok(it + 1)) ATT_STORE(1, 1)
    PROBE(4)
    __syncthreads();
    PROBE(5)
    // odd slot it + 1 (LDS buffer 1, register set 1)
    if (it + 1 < niter) {
      if (slot_ok(it + 3)) ATT_LOAD(1, slot_key0(it + 3))
      PROBE(0)
      if (slot_ok(it + 1)) run_tile(it + 1);
      if (slot_ok(it + 2)) ATT_STORE(0, 0)
      PROBE(4)
    }
    __syncthreads();
    PROBE(5)
  }
#ifdef VSD_ATTN_PROBE
  if (p.probe && blockIdx.x == 0 && blockIdx.y == 0 && image == 0 && tid == 0)
    for (int i = 0; i < 8; ++i) p.probe[i] = pacc[i];
#endif
#undef ATT_LOAD
#undef ATT_STORE
#undef ATT_LOAD_K
#undef ATT_LOAD_V
#undef ATT_STORE_K
#undef ATT_STORE_V

  // ---- merge the key-split groups: group 1 hands (max, sum, O) of its keys to group 0 through LDS
  if constexpr (KSP > 1) {
    constexpr int PER_WAVE = QB * (NPV * 16 + 2) * 64;
    float* mg = reinterpret_cast<float*>(smem) + gwave * PER_WAVE + lane;
    if (grp == 1) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float* b = mg + qb * (NPV * 16 + 2) * 64;
        b[0] = m_run[qb];
        b[64] = l_run[qb];
#pragma unroll
        for (int i = 0; i < NPV; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) b[(2 + i * 16 + r) * 64] = o[qb][i][r];
      }
    }
    __syncthreads();
    if (grp != 0) return;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const float* b = mg + qb * (NPV * 16 + 2) * 64;
      const float m1 = b[0], l1 = b[64];
      const float m = fmaxf(m_run[qb], m1);
      const float a0 = __builtin_amdgcn_exp2f(m_run[qb] - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
      l_run[qb] = l_run[qb] * a0 + l1 * a1;
#pragma unroll
      for (int i = 0; i < NPV; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[qb][i][r] = o[qb][i][r] * a0 + b[(2 + i * 16 + r) * 64] * a1;
    }
  }

  // ---- epilogue: normalise and write O[q][head*d + dd]
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float l_tot;
    if (ones_row) {  // O^T row DV-1 = register 15 of the last block on the upper half-wave
      const float mine = o[qb][NPV - 1][15];
      const float other = __shfl_xor(mine, 32);
      l_tot = lh ? mine : other;
    } else {
      l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32);
    }
    const float inv = 1.0f / l_tot;
    if (qvalid[qb]) {
      half_t* orow = p.out + (size_t)qrow[qb] * p.ldo + head * d;
#pragma unroll
      for (int db = 0; db < NPV; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          int dd = db * 32 + 8 * g + 4 * lh;
          if (dd < d) {
            half4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (half_t)(o[qb][db][4 * g + j] * inv);
            *reinterpret_cast<half4*>(orow + dd) = v;
          }
        }
    }
  }
}

template <int NQK, int NPV, int NW, int QB, int KSP, bool LAZY = false>
__global__ __launch_bounds__(64 * NW * KSP) void attention_kernel(const AttnParams pp) {
  attention_body<NQK, NPV, NW, QB, KSP, LAZY>(pp, blockIdx.z);
}
// (the LAZY body of the four-wave form allocates 170 registers on its own -- two past the 168 that three waves per SIMD allow, which is
//  what the plain form runs at: said here)
template <int NQK, int NPV, int NW, int QB, int KSP>
__global__ __launch_bounds__(64 * NW * KSP) __attribute__((amdgpu_waves_per_eu(3))) void attention_lazy_kernel(const AttnParams pp) {
  attention_body<NQK, NPV, NW, QB, KSP, true>(pp, blockIdx.z);
}
template <int NQK, int NPV, int NW, int QB, int KSP>
__global__ __launch_bounds__(64 * NW * KSP) __attribute__((amdgpu_waves_per_eu(3))) void attention_lazy_pair_kernel(const Pair<AttnParams> g) {
  const unsigned nb = gridDim.z >> 1, which = blockIdx.z >= nb;
  attention_body<NQK, NPV, NW, QB, KSP, true>(g.p[which], blockIdx.z - which * nb);
}
// two attention problems of one shape as one grid (common.h launch_pairable): the upper half of gridDim.z is the second one
template <int NQK, int NPV, int NW, int QB, int KSP, bool LAZY = false>
__global__ __launch_bounds__(64 * NW * KSP) void attention_pair_kernel(const Pair<AttnParams> g) {
  const unsigned nb = gridDim.z >> 1, which = blockIdx.z >= nb;
  attention_body<NQK, NPV, NW, QB, KSP, LAZY>(g.p[which], blockIdx.z - which * nb);
}

// (waves per group, query blocks per wave, key-split groups)
struct AttnShape {
  int nw, qb, ksp;
};

template <int NQK, int NPV, int NW, int QB, int KSP>
void launch_one(vsd_ctx* ctx, const AttnParams& p, int batch, hipStream_t s) {
  dim3 grid(p.heads, (p.sq + 32 * NW * QB - 1) / (32 * NW * QB), batch);
  if constexpr (NQK == 3 && NPV == 2) {  // head_dim 40: the maximum subtracted inside the QK^T MFMA (attention_body, LAZY)
    static const bool off = getenv("VSD_ATTN_NO_LAZY") != nullptr;
    if (p.d == 40 && !p.causal && !off) {
      if constexpr (NW == 4 && KSP == 1)
        launch_pairable(ctx, attention_lazy_kernel<NQK, NPV, NW, QB, KSP>, attention_lazy_pair_kernel<NQK, NPV, NW, QB, KSP>, grid, dim3(64 * NW * KSP), 0, s, p);
      else
        launch_pairable(ctx, attention_kernel<NQK, NPV, NW, QB, KSP, true>, attention_pair_kernel<NQK, NPV, NW, QB, KSP, true>, grid, dim3(64 * NW * KSP), 0, s, p);
      return;
    }
  }
  launch_pairable(ctx, attention_kernel<NQK, NPV, NW, QB, KSP>, attention_pair_kernel<NQK, NPV, NW, QB, KSP>, grid, dim3(64 * NW * KSP), 0, s, p);
}

template <int NQK, int NPV>
void launch_attn(vsd_ctx* ctx, const AttnParams& p, AttnShape sh, int batch, hipStream_t s) {
  // QB = 2 (two query blocks per wave) is implemented but not instantiated: measured 1.3x SLOWER on MI355X at every
  // shape (the kernel is issue-bound, not LDS-bound; two blocks per wave need > 256 VGPRs, i.e. one wave per SIMD).
  constexpr bool SMALL = NPV <= 3;  // LDS budget of the KSP = 2 form
  if constexpr (SMALL) {
    if (sh.ksp == 2) return launch_one<NQK, NPV, 4, 1, 2>(ctx, p, batch, s);
  }
  if (sh.nw == 2) return launch_one<NQK, NPV, 2, 1, 1>(ctx, p, batch, s);
  return launch_one<NQK, NPV, 4, 1, 1>(ctx, p, batch, s);
}

}  // namespace

#ifdef VSD_ATTN_PROBE
extern "C" void vsd_attn_set_probe(void* buf) { g_probe = (long long*)buf; }
#endif

extern "C" int vsd_attention(vsd_ctx* ctx, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                             void* out, int ldo, int sq, int sk, int heads, int d, float scale, int causal,
                             void* stream) {
  return vsd_attention_batched(ctx, q, ldq, k, ldk, vt, ldvt, out, ldo, sq, sk, heads, d, scale, causal, 1, 0, 0, stream);
}

extern "C" int vsd_attention_batched(vsd_ctx* ctx, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                                     void* out, int ldo, int sq, int sk, int heads, int d, float scale, int causal,
                                     int batch, int k_batch_rows, int vt_batch_cols, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
#ifdef VSD_PROBE
  { static const bool skip = getenv("VSD_SKIP_ATTN") != nullptr; if (skip) return VSD_OK; }  // (what-if probe builds only)
#endif
  if (batch < 1 || batch > 65535 || k_batch_rows < 0 || vt_batch_cols < 0 || vt_batch_cols % 8)
    return vsd_fail(ctx, VSD_ERR_ARG, "attention: batch=%d k_batch_rows=%d vt_batch_cols=%d", batch, k_batch_rows, vt_batch_cols);
  if (!q || !k || !vt || !out) return vsd_fail(ctx, VSD_ERR_ARG, "attention: null pointer");
  if (d % 8 || d < 8 || d > 160) return vsd_fail(ctx, VSD_ERR_ARG, "attention: head_dim %d unsupported", d);
  if (sq <= 0 || sk <= 0 || heads <= 0) return vsd_fail(ctx, VSD_ERR_ARG, "attention: empty problem");
  if (ldq % 8 || ldk % 8 || ldvt % 8 || ldo % 4 || ldvt < (batch - 1) * vt_batch_cols + ((sk + 63) / 64) * 64)
    return vsd_fail(ctx, VSD_ERR_ARG, "attention: bad leading dimensions (ldvt=%d must cover round_up(sk=%d,64))", ldvt, sk);
  AttnParams p;
  p.q = (const half_t*)q; p.k = (const half_t*)k; p.vt = (const half_t*)vt; p.out = (half_t*)out;
  p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo;
  p.sq = sq; p.sk = sk; p.heads = heads; p.d = d;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.causal = causal;
  p.k_brows = k_batch_rows; p.vt_bcols = vt_batch_cols;
  VSD_CUT_SET(p)
#ifdef VSD_ATTN_PROBE
  p.probe = g_probe;
#endif
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls(ctx, s, VSD_FAM_ATTENTION, 4.0 * sq * (double)sk * heads * d * batch);
  const int nqk = (d + 15) / 16;
  // Shape of the launch (measured on MI355X, scripts/attn_bench.py):
  //   long key loops and at most one 128-query workgroup per CU -> KSP = 2: two wave groups split the keys, so every
  //       SIMD holds two waves (4096 keys, d=40, 8 heads: 87 -> 58 us); with more workgroups than CUs the plain form
  //       already has that occupancy and the merge only costs;
  //   tiny problems -> 2-wave workgroups so that more CUs get one.
  const long wgs = (long)((sq + 127) / 128) * heads * batch;
  AttnShape sh = {4, 1, 1};
  const int ntiles = (sk + 63) / 64;
  if (d <= 96 && !causal && ntiles >= 8 && wgs <= 256) sh.ksp = 2;
  if (sh.ksp == 1 && wgs < 64) sh.nw = 2;
  if (const char* e = getenv("VSD_ATTN_SHAPE")) {  // "nw,qb,ksp" (benchmarking)
    int a = 4, b = 1, c = 1;
    if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3) {
      sh.nw = a == 2 ? 2 : 4;
      sh.qb = 1;
      (void)b;
      sh.ksp = (c == 2 && d <= 96) ? 2 : 1;
    }
  }
  if (nqk <= 1) launch_attn<1, 1>(ctx, p, sh, batch, s);
  else if (nqk == 2) launch_attn<2, 1>(ctx, p, sh, batch, s);
  else if (nqk == 3) launch_attn<3, 2>(ctx, p, sh, batch, s);
  else if (nqk == 4) launch_attn<4, 2>(ctx, p, sh, batch, s);
  else if (nqk == 5) launch_attn<5, 3>(ctx, p, sh, batch, s);
  else if (nqk == 6) launch_attn<6, 3>(ctx, p, sh, batch, s);
  else if (nqk <= 8) launch_attn<8, 4>(ctx, p, sh, batch, s);
  else launch_attn<10, 5>(ctx, p, sh, batch, s);
  return ls.finish();
}

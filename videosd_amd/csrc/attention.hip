// Fused attention forward softmax(Q K^T * scale) V for gfx950 (wave64, v_mfma_f32_32x32x16_f16).
//
// One workgroup = 4 waves = 128 queries of one head; each wave owns 32 queries and the whole key
// loop.  K/V^T tiles of 64 keys are staged in LDS and shared by the 4 waves.
//   S^T = K Q^T  (swapped operands): the accumulator has the QUERY on the lane and 16 keys in the
//   registers, so the row max / row sum are in-lane reductions plus one cross-half shuffle, and the
//   running max / sum / rescale factor are per-lane scalars.
//   O^T += V^T P^T: the S^T accumulator, converted to fp16 pairs, IS the B operand of the second
//   MFMA (k index = key, permuted inside each 16-key step exactly as the accumulator rows are), so P
//   never touches LDS.  V arrives pre-transposed ([d][keys], written by the QKV GEMM epilogue).
// head_dim d (multiple of 8, <= 160) is zero-padded to NQK*16 for QK^T and NPV*32 for PV.
// Algorithmic FLOPs per launch: 4*sq*sk*heads*d.
#include <stdarg.h>
#include <stdlib.h>

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
};

constexpr float NEG_BIG = -1.0e30f;

template <int NQK, int NPV, int NW>
__global__ __launch_bounds__(64 * NW) void attention_kernel(const AttnParams p) {
  constexpr int NT = 64 * NW;        // threads
  constexpr int KS = NQK * 16 + 8;   // K tile row pitch (halfs): odd number of 16-byte slots
  constexpr int VS = 64 + 4;         // V^T tile row pitch (halfs): 136 bytes
  constexpr int DV = NPV * 32;
  constexpr int KTILE = 64 * KS, VTILE = DV * VS;
  constexpr int KCH = (64 * NQK * 2 + NT - 1) / NT;  // 16-byte chunks per thread per K tile (upper bound)
  constexpr int VCH = (DV * 8 + NT - 1) / NT;        // ... per V^T tile
  __shared__ __attribute__((aligned(16))) half_t smem[2 * (KTILE + VTILE)];  // double buffered
  half_t* Ks = smem;
  half_t* Vs = smem + 2 * KTILE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lr = lane & 31;
  const int lh = lane >> 5;
  const int head = blockIdx.y;
  const int q0 = blockIdx.x * (32 * NW) + wave * 32;
  const int d = p.d;
  const int dch = d >> 3;  // 16-byte chunks per row

  // zero both LDS buffers once (padding columns / rows stay zero for the whole kernel)
  for (int i = tid; i < 2 * (KTILE + VTILE) / 8; i += NT) reinterpret_cast<u32x4*>(smem)[i] = (u32x4){0u, 0u, 0u, 0u};

  // ---- per-thread staging coordinates (fixed for the whole kernel)
  int kg[KCH], kl[KCH], krow[KCH];
#pragma unroll
  for (int i = 0; i < KCH; ++i) {
    int c = tid + i * NT;
    bool ok = c < 64 * dch;
    int kr = ok ? c / dch : 0, kc = ok ? c - kr * dch : 0;
    krow[i] = ok ? kr : 1 << 28;  // never < sk
    kg[i] = kr * p.ldk + head * d + kc * 8;
    kl[i] = kr * KS + kc * 8;
  }
  int vg[VCH], vl[VCH];
  bool vok[VCH];
#pragma unroll
  for (int i = 0; i < VCH; ++i) {
    int c = tid + i * NT;
    vok[i] = c < d * 8;
    int vr = vok[i] ? c >> 3 : 0, vc = c & 7;
    vg[i] = (head * d + vr) * p.ldvt + vc * 8;
    vl[i] = vr * VS + vc * 8;
  }
  u32x4 kreg[KCH], vreg[VCH];
  const u32x4 zero4 = (u32x4){0u, 0u, 0u, 0u};
#define ATT_LOAD(KEY0_)                                                                             \
  {                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < KCH; ++i) {                                               \
      bool ok = (KEY0_) + krow[i] < p.sk;                                                           \
      u32x4 v = *reinterpret_cast<const u32x4*>(p.k + (ok ? (size_t)(KEY0_) * p.ldk + kg[i] : 0));  \
      kreg[i] = ok ? v : zero4;                                                                     \
    }                                                                                               \
    _Pragma("unroll") for (int i = 0; i < VCH; ++i) {                                               \
      u32x4 v = *reinterpret_cast<const u32x4*>(p.vt + (vok[i] ? (size_t)vg[i] + (KEY0_) : 0));     \
      vreg[i] = v;                                                                                  \
    }                                                                                               \
  }
#define ATT_STORE(BUF_)                                                                             \
  {                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < KCH; ++i)                                                 \
      if (krow[i] < 64) *reinterpret_cast<u32x4*>(Ks + (BUF_) * KTILE + kl[i]) = kreg[i];           \
    _Pragma("unroll") for (int i = 0; i < VCH; ++i)                                                 \
      if (vok[i]) {                                                                                 \
        u32x2* dst = reinterpret_cast<u32x2*>(Vs + (BUF_) * VTILE + vl[i]);                         \
        dst[0] = (u32x2){vreg[i][0], vreg[i][1]};                                                   \
        dst[1] = (u32x2){vreg[i][2], vreg[i][3]};                                                   \
      }                                                                                             \
  }

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q0+lr][ks*16 + lh*8 .. +8], pre-scaled
  half8 qf[NQK];
  const int qrow = q0 + lr;
  const bool qvalid = qrow < p.sq;
#pragma unroll
  for (int ks = 0; ks < NQK; ++ks) {
    int doff = ks * 16 + lh * 8;
    half8 v = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    if (qvalid && doff < d) {
      v = *reinterpret_cast<const half8*>(p.q + (size_t)qrow * p.ldq + head * d + doff);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (half_t)((float)v[j] * p.scale_log2);
    }
    qf[ks] = v;
  }

  f32x16 o[NPV];
#pragma unroll
  for (int i = 0; i < NPV; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;

  int sk_end = p.sk;
  if (p.causal) sk_end = min(p.sk, (int)blockIdx.x * (32 * NW) + 32 * NW);  // later keys are never visible
  const int ntiles = (sk_end + 63) / 64;

  ATT_LOAD(0)
  __syncthreads();  // zero fill complete before the first tile lands on top of it
  ATT_STORE(0)
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * 64;
    const int buf = t & 1;
    const bool more = t + 1 < ntiles;
    if (more) ATT_LOAD(key0 + 64)  // next tile's global loads fly during this tile's MFMAs
    const half_t* Kb = Ks + buf * KTILE;
    const half_t* Vb = Vs + buf * VTILE;

    // ---- S^T = K Q^T for two 32-key chains
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < NQK; ++ks) {
        half8 kf = *reinterpret_cast<const half8*>(Kb + (kb * 32 + lr) * KS + ks * 16 + lh * 8);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[kb], 0, 0, 0);
      }
    }
    // ---- masking (tail keys / causal)
    const bool need_mask = (key0 + 64 > p.sk) || p.causal;
    if (need_mask) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          bool ok = key < p.sk && (!p.causal || key <= qrow);
          if (!ok) s[kb][r] = NEG_BIG;
        }
    }
    // ---- online softmax (per lane = per query; the two half-waves hold different keys of the same query)
    float mx = NEG_BIG;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float e = __builtin_amdgcn_exp2f(s[kb][r] - m_new);
        s[kb][r] = e;
        psum += e;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int i = 0; i < NPV; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;

    // ---- O^T += V^T P^T
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        half8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (half_t)s[kb][8 * st + j];
#pragma unroll
        for (int db = 0; db < NPV; ++db) {
          const half_t* vrow = Vb + (db * 32 + lr) * VS + kb * 32 + 16 * st + 4 * lh;
          half4 lo = *reinterpret_cast<const half4*>(vrow);
          half4 hi = *reinterpret_cast<const half4*>(vrow + 8);
          half8 vf = (half8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[db], 0, 0, 0);
        }
      }
    }
    if (more) ATT_STORE(buf ^ 1)
    __syncthreads();
  }
#undef ATT_LOAD
#undef ATT_STORE

  // ---- epilogue: normalise and write O[q][head*d + dd]
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (qvalid) {
    half_t* orow = p.out + (size_t)qrow * p.ldo + head * d;
#pragma unroll
    for (int db = 0; db < NPV; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int dd = db * 32 + 8 * g + 4 * lh;
        if (dd < d) {
          half4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (half_t)(o[db][4 * g + j] * inv);
          *reinterpret_cast<half4*>(orow + dd) = v;
        }
      }
  }
}

template <int NQK, int NPV>
void launch_attn(const AttnParams& p, int nw, hipStream_t s) {
  if (nw == 2) {
    dim3 grid((p.sq + 63) / 64, p.heads);
    hipLaunchKernelGGL((attention_kernel<NQK, NPV, 2>), grid, dim3(128), 0, s, p);
  } else {
    dim3 grid((p.sq + 127) / 128, p.heads);
    hipLaunchKernelGGL((attention_kernel<NQK, NPV, 4>), grid, dim3(256), 0, s, p);
  }
}

}  // namespace

extern "C" int vsd_attention(vsd_ctx* ctx, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt,
                             void* out, int ldo, int sq, int sk, int heads, int d, float scale, int causal,
                             void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!q || !k || !vt || !out) return vsd_fail(ctx, VSD_ERR_ARG, "attention: null pointer");
  if (d % 8 || d < 8 || d > 160) return vsd_fail(ctx, VSD_ERR_ARG, "attention: head_dim %d unsupported", d);
  if (sq <= 0 || sk <= 0 || heads <= 0) return vsd_fail(ctx, VSD_ERR_ARG, "attention: empty problem");
  if (ldq % 8 || ldk % 8 || ldvt % 8 || ldo % 4 || ldvt < ((sk + 63) / 64) * 64)
    return vsd_fail(ctx, VSD_ERR_ARG, "attention: bad leading dimensions (ldvt=%d must cover round_up(sk=%d,64))", ldvt, sk);
  AttnParams p;
  p.q = (const half_t*)q; p.k = (const half_t*)k; p.vt = (const half_t*)vt; p.out = (half_t*)out;
  p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo;
  p.sq = sq; p.sk = sk; p.heads = heads; p.d = d;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.causal = causal;
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls(ctx, s, VSD_FAM_ATTENTION, 4.0 * sq * (double)sk * heads * d);
  const int nqk = (d + 15) / 16, npv = (d + 31) / 32;
  // 64-query workgroups (2 waves) when 128-query ones would leave the 256 CUs with < 2 workgroups each
  int nw = (((sq + 127) / 128) * heads < 64) ? 2 : 4;
  if (const char* e = getenv("VSD_ATTN_NW")) nw = atoi(e) == 2 ? 2 : 4;
  if (nqk <= 1) launch_attn<1, 1>(p, nw, s);
  else if (nqk == 2) launch_attn<2, 1>(p, nw, s);
  else if (nqk == 3) launch_attn<3, 2>(p, nw, s);
  else if (nqk == 4) launch_attn<4, 2>(p, nw, s);
  else if (nqk == 5) launch_attn<5, 3>(p, nw, s);
  else if (nqk == 6) launch_attn<6, 3>(p, nw, s);
  else if (nqk <= 8) launch_attn<8, 4>(p, nw, s);
  else launch_attn<10, 5>(p, nw, s);
  (void)npv;
  return ls.finish();
}

// Per-prompt constants of the "absorbed" cross-attention (gfx950): the text's key / value projections folded into the
// query / output weights of a wide transformer block, on the GPU.
//
// Cross-attention over a FIXED key set (the 77 text tokens; Attention.forward of attn2 in diffusers' BasicTransformerBlock
// under lcm_controlnet.py:558,568) is run as two plain GEMMs (include/vsd.h, VSD_ACT_SOFTMAX):
//     out = softmax_h( LN(x) G^T ) Z^T + bo,
//     G[h*128 + j, :] = scale * sum_d K_h[j, d] Wq[h*dh + d, :]        (j < tl; zero rows above)
//     Z[:, h*128 + j] = sum_d Wo[:, h*dh + d] V_h[j, d]
// with the LayerNorm in front of the query projection folded into G like every LN-consuming layer of this library
// (videosd_amd/packing.py _fold_ln): the stored weights are fp16(G * gamma), s[n] = sum_c of those fp16 values,
// t[n] = sum_c G[n, c] * beta[c].  Round 2 built G / Z on the HOST in fp32 (download of K / V^T of 16 layers, ~0.1 s of torch
// matmuls, upload) -- a prompt edit stalled the stream; here it is two small kernels per layer (fp32 FMA in registers,
// 0.4 + 0.4 GFLOP at C = 1280: no MFMA needed, the work is far below one launch's fixed cost ... x 16 layers).
// Bound: latency of the weight-row loads (L2 resident); algorithmic bytes per layer: 2 C^2 fp16 weights + 2 * 1024 * C outputs.
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int XG = 128;  // columns per head (include/vsd.h: one 128-column tile per head)

// One workgroup = one row n = h * 128 + j of G.  Thread t owns 8 consecutive input channels (16-byte weight loads).
__global__ __launch_bounds__(256) void xattn_fold_g_kernel(const half_t* __restrict__ k, int ldk, int tl, const half_t* __restrict__ wq,
                                                           const half_t* __restrict__ gamma, const half_t* __restrict__ beta, int c,
                                                           int dh, float scale, half_t* __restrict__ w_out, float* __restrict__ s_out,
                                                           float* __restrict__ t_out) {
  __shared__ float red[2 * 4];
  const int n = blockIdx.x, h = n / XG, j = n - h * XG;
  const int tid = threadIdx.x;
  const int nch = c >> 3;
  float s_part = 0.f, t_part = 0.f;
  for (int ch = tid; ch < nch; ch += 256) {  // (c <= 2048: one pass)
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (j < tl) {
      const half_t* krow = k + (size_t)j * ldk + h * dh;
      const half_t* wcol = wq + (size_t)h * dh * c + ch * 8;
      int d = 0;
      for (; d + 4 <= dh; d += 4) {  // four independent weight rows in flight; the sum runs in d order
        const half8 w0 = *reinterpret_cast<const half8*>(wcol + (size_t)d * c);
        const half8 w1 = *reinterpret_cast<const half8*>(wcol + (size_t)(d + 1) * c);
        const half8 w2 = *reinterpret_cast<const half8*>(wcol + (size_t)(d + 2) * c);
        const half8 w3 = *reinterpret_cast<const half8*>(wcol + (size_t)(d + 3) * c);
        const float k0 = (float)krow[d], k1 = (float)krow[d + 1], k2 = (float)krow[d + 2], k3 = (float)krow[d + 3];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] = fmaf(k0, (float)w0[i], acc[i]);
          acc[i] = fmaf(k1, (float)w1[i], acc[i]);
          acc[i] = fmaf(k2, (float)w2[i], acc[i]);
          acc[i] = fmaf(k3, (float)w3[i], acc[i]);
        }
      }
      for (; d < dh; ++d) {
        const half8 w0 = *reinterpret_cast<const half8*>(wcol + (size_t)d * c);
        const float k0 = (float)krow[d];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(k0, (float)w0[i], acc[i]);
      }
    }
    const half8 ga = *reinterpret_cast<const half8*>(gamma + ch * 8);
    const half8 be = *reinterpret_cast<const half8*>(beta + ch * 8);
    half8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float g = scale * acc[i];
      o[i] = (half_t)(g * (float)ga[i]);
      s_part += (float)o[i];          // summed over the SAME rounded values the GEMM multiplies with (packing._fold_ln)
      t_part += g * (float)be[i];
    }
    *reinterpret_cast<half8*>(w_out + (size_t)n * c + ch * 8) = o;
  }
  // fixed-order fold: lanes by xor shuffles, then the four waves in order (deterministic)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s_part += __shfl_xor(s_part, o);
    t_part += __shfl_xor(t_part, o);
  }
  if ((tid & 63) == 0) {
    red[2 * (tid >> 6)] = s_part;
    red[2 * (tid >> 6) + 1] = t_part;
  }
  __syncthreads();
  if (tid == 0) {
    s_out[n] = (red[0] + red[2]) + (red[4] + red[6]);
    t_out[n] = (red[1] + red[3]) + (red[5] + red[7]);
  }
}

// One workgroup = 32 output rows (channels c) of one head's 128-column group of Z.  Thread -> (row, 16 key columns).
__global__ __launch_bounds__(256) void xattn_fold_z_kernel(const half_t* __restrict__ vt, int ldvt, int tl, const half_t* __restrict__ wo,
                                                           int c, int dh, int heads, half_t* __restrict__ z_out) {
  const int h = blockIdx.y;
  const int row = blockIdx.x * 32 + (threadIdx.x >> 3);
  const int j0 = (threadIdx.x & 7) * 16;
  if (row >= c) return;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const half_t* wrow = wo + (size_t)row * c + h * dh;
  const bool lo_ok = j0 < tl && j0 + 8 <= ldvt, hi_ok = j0 + 8 < tl && j0 + 16 <= ldvt;  // (V^T is zero beyond tl by contract)
  if (lo_ok) {
    const half8 z8 = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    for (int d = 0; d < dh; ++d) {
      const float w = (float)wrow[d];
      const half_t* vrow = vt + (size_t)(h * dh + d) * ldvt + j0;
      const half8 a = *reinterpret_cast<const half8*>(vrow);
      const half8 b = hi_ok ? *reinterpret_cast<const half8*>(vrow + 8) : z8;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = fmaf(w, (float)a[i], acc[i]);
        acc[8 + i] = fmaf(w, (float)b[i], acc[8 + i]);
      }
    }
  }
  half8 o0, o1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    o0[i] = (half_t)(j0 + i < tl ? acc[i] : 0.f);
    o1[i] = (half_t)(j0 + 8 + i < tl ? acc[8 + i] : 0.f);
  }
  half_t* dst = z_out + (size_t)row * (heads * XG) + h * XG + j0;
  *reinterpret_cast<half8*>(dst) = o0;
  *reinterpret_cast<half8*>(dst + 8) = o1;
}

}  // namespace

extern "C" int vsd_xattn_fold(vsd_ctx* ctx, const void* k, int ldk, const void* vt, int ldvt, int tl, const void* wq, const void* wo,
                              const void* gamma, const void* beta, int c, int heads, float scale, void* xa1_w, void* xa1_s, void* xa1_t,
                              void* xa2_w, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!k || !vt || !wq || !wo || !gamma || !beta || !xa1_w || !xa1_s || !xa1_t || !xa2_w)
    return vsd_fail(ctx, VSD_ERR_ARG, "xattn_fold: null pointer");
  if (c <= 0 || c % 64 || heads <= 0 || c % heads || (c / heads) % 8 || tl <= 0 || tl > XG || ldk < c || ldk % 8 || ldvt < tl || ldvt % 8)
    return vsd_fail(ctx, VSD_ERR_ARG, "xattn_fold: bad shape c=%d heads=%d tl=%d ldk=%d ldvt=%d (c %% 64 == 0, head size %% 8 == 0, tl <= 128)",
                    c, heads, tl, ldk, ldvt);
  hipStream_t s = (hipStream_t)stream;
  const int dh = c / heads;
  {
    LaunchScope ls(ctx, s, VSD_FAM_ELEMENTWISE, 0.0);
    hipLaunchKernelGGL(xattn_fold_g_kernel, dim3(heads * XG), dim3(256), 0, s, (const half_t*)k, ldk, tl, (const half_t*)wq,
                       (const half_t*)gamma, (const half_t*)beta, c, dh, scale, (half_t*)xa1_w, (float*)xa1_s, (float*)xa1_t);
    int rc = ls.finish();
    if (rc) return rc;
  }
  LaunchScope ls(ctx, s, VSD_FAM_ELEMENTWISE, 0.0);
  hipLaunchKernelGGL(xattn_fold_z_kernel, dim3(cdiv(c, 32), heads), dim3(256), 0, s, (const half_t*)vt, ldvt, tl, (const half_t*)wo, c, dh,
                     heads, (half_t*)xa2_w);
  return ls.finish();
}

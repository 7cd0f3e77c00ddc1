// Per-frame elementwise kernels (gfx950): image pre/post-processing, Sobel edge map, LCM scheduler
// arithmetic.  All are tiny and HBM/launch-bound; they exist so that a frame needs no host round trip
// between the u8 upload and the u8 download (the reference syncs twice per frame: canny_gpu.py:44 and
// lcm_controlnet.py:609-611).
#include <stdarg.h>

#include "common.h"

namespace {

__global__ void preprocess_rgb_kernel(const unsigned char* __restrict__ rgb, int hw, half_t* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hw) return;
  half8 o = (half8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float x = (float)rgb[i * 3 + c] / 255.0f;          // pil_to_numpy
    half_t y = (half_t)(2.0f * x - 1.0f);              // normalize, cast to fp16 (prepare_latents dtype)
    half_t z = (half_t)((float)y + 1.0f);              // EncoderTiny: x.add(1)
    o[c] = (half_t)((float)z * 0.5f);                  //              .div(2)
  }
  *reinterpret_cast<half8*>(out + (size_t)i * 8) = o;
}

// ---- Sobel control image ---------------------------------------------------------------------------------------------
// One workgroup = SOBEL_TW pixels of one image row. The bytes of rows y-1..y+1 (one pixel of apron either side) come in with
// 16-byte aligned loads, one per lane, into LDS; the gray values (PIL convert("L"), ToTensor) are formed once per pixel
// there; the 3x3 taps are LDS reads.
constexpr int SOBEL_TW = 256;
constexpr int SOBEL_RAW16 = 52;  // 16-byte pieces per row: (SOBEL_TW + 2) * 3 bytes + up to 15 of alignment slack, rounded up

struct SobelTile {
  uint4 raw[3][SOBEL_RAW16];
  float g[3][SOBEL_TW + 2];
};

__device__ __forceinline__ float sobel_mag_tile(const unsigned char* __restrict__ rgb, int h, int w, int y, int x0,
                                                SobelTile& t) {
  const int tid = threadIdx.x;
  const unsigned char* end = rgb + (size_t)h * w * 3;
  const int xs = x0 > 0 ? x0 - 1 : 0, xe = min(x0 + SOBEL_TW + 1, w);  // pixel columns [xs, xe) this tile needs
  // phase 1: raw bytes
  if (tid < 3 * 64) {
    const int r = tid >> 6, j = tid & 63, yy = y + r - 1;
    if ((unsigned)yy < (unsigned)h) {
      const unsigned char* b0 = rgb + ((size_t)yy * w + xs) * 3;
      const unsigned char* b1 = rgb + ((size_t)yy * w + xe) * 3;
      const unsigned char* a0 = reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(b0) & ~(uintptr_t)15);
      const unsigned char* q = a0 + 16 * j;
      if (q < b1 && j < SOBEL_RAW16) {
        uint4 v;
        if (q >= rgb && q + 16 <= end) {
          v = *reinterpret_cast<const uint4*>(q);
        } else {  // first / last piece of the frame: stay inside the buffer
          unsigned char tmp[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) tmp[k] = (q + k >= rgb && q + k < end) ? q[k] : (unsigned char)0;
          v = *reinterpret_cast<const uint4*>(tmp);
        }
        t.raw[r][j] = v;
      }
    }
  }
  __syncthreads();
  // phase 2: gray value of every pixel of the 3 x (SOBEL_TW + 2) patch; zero outside the image (conv2d padding=1)
  for (int q = tid; q < 3 * (SOBEL_TW + 2); q += blockDim.x) {
    const int r = q / (SOBEL_TW + 2), c = q - r * (SOBEL_TW + 2);
    const int yy = y + r - 1, px = x0 - 1 + c;
    float gv = 0.f;
    if ((unsigned)yy < (unsigned)h && (unsigned)px < (unsigned)w) {
      const unsigned char* b0 = rgb + ((size_t)yy * w + xs) * 3;
      const int skew = (int)(reinterpret_cast<uintptr_t>(b0) & 15);
      const unsigned char* rb = reinterpret_cast<const unsigned char*>(t.raw[r]) + skew + (px - xs) * 3;
      unsigned l = (rb[0] * 19595u + rb[1] * 38470u + rb[2] * 7471u + 0x8000u) >> 16;  // PIL convert("L")
      gv = (float)l / 255.0f;                                                          // ToTensor
    }
    t.g[r][c] = gv;
  }
  __syncthreads();
  if (x0 + tid >= w) return 0.f;
  const float a = t.g[0][tid], b = t.g[0][tid + 1], c = t.g[0][tid + 2];
  const float d = t.g[1][tid], f = t.g[1][tid + 2];
  const float g = t.g[2][tid], hh = t.g[2][tid + 1], i = t.g[2][tid + 2];
  const float ex = (c - a) + 2.0f * (f - d) + (i - g);
  const float ey = (g - a) + 2.0f * (hh - b) + (i - c);
  return sqrtf(ex * ex + ey * ey);
}

// Global maximum in two plain steps, no atomics and no memset node: every block of sobel_max stores ITS maximum (one
// float per block); every block of sobel_apply folds those (a few KB, L2 resident).
__global__ void __launch_bounds__(SOBEL_TW) sobel_max_kernel(const unsigned char* __restrict__ rgb, int h, int w,
                                                              float* __restrict__ blockmax) {
  __shared__ SobelTile t;
  __shared__ float wm[SOBEL_TW / 64];
  float m = sobel_mag_tile(rgb, h, w, blockIdx.y, blockIdx.x * SOBEL_TW, t);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    blockmax[blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));  // order independent
}

__global__ void __launch_bounds__(SOBEL_TW) sobel_apply_kernel(const unsigned char* __restrict__ rgb, int h, int w, float low,
                                                                float high, const float* __restrict__ blockmax, int nblk,
                                                                unsigned char* __restrict__ edge, half_t* __restrict__ ctrl) {
  __shared__ SobelTile t;
  __shared__ float wm[SOBEL_TW / 64];
  float mx = 0.f;
  for (int j = threadIdx.x; j < nblk; j += blockDim.x) mx = fmaxf(mx, blockmax[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = mx;
  const int y = blockIdx.y, x = blockIdx.x * SOBEL_TW + threadIdx.x;
  const float mag = sobel_mag_tile(rgb, h, w, y, blockIdx.x * SOBEL_TW, t);  // its barriers also publish wm
  mx = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
  if (x >= w) return;
  float e = mag / mx;  // 0/0 -> NaN on an all-black frame, like the reference
  if (e >= high) e = 1.0f;
  if (e <= low) e = 0.0f;
  float s = e * 255.0f;
  unsigned char u = (s == s) ? (unsigned char)(int)s : 0;  // .byte() truncation; NaN -> 0
  const size_t i = (size_t)y * w + x;
  edge[i] = u;
  half_t v = (half_t)((float)u / 255.0f);
  *reinterpret_cast<half8*>(ctrl + i * 8) = (half8){v, v, v, 0, 0, 0, 0, 0};
}

__global__ void add_noise_kernel(const half_t* __restrict__ x0, const float* __restrict__ noise, float sa, float sb, int hw,
                                 half_t* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hw) return;
  half8 x = *reinterpret_cast<const half8*>(x0 + (size_t)i * 8);
  half8 o = (half8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 4; ++c) o[c] = (half_t)(sa * (float)x[c] + sb * noise[(size_t)c * hw + i]);
  *reinterpret_cast<half8*>(out + (size_t)i * 8) = o;
}

struct StepCoef {
  float sa, sb, cskip, cout, sap, sbp;
};

// Same arithmetic with the coefficients read from device memory: a captured graph then follows a new strength /
// schedule without being captured again (the host rewrites the few floats between replays).  blockIdx.y = image of a
// batch (every image uses the same noise draw: the reference resets its RNG per frame).
__global__ void add_noise_dev_kernel(const half_t* __restrict__ x0, const float* __restrict__ noise, const float* __restrict__ coef,
                                     int hw, half_t* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hw) return;
  const float sa = coef[0], sb = coef[1];
  const size_t row = (size_t)blockIdx.y * hw + i;
  half8 x = *reinterpret_cast<const half8*>(x0 + row * 8);
  half8 o = (half8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 4; ++c) o[c] = (half_t)(sa * (float)x[c] + sb * noise[(size_t)c * hw + i]);
  *reinterpret_cast<half8*>(out + row * 8) = o;
}

__global__ void lcm_step_kernel(const half_t* __restrict__ eps, const half_t* __restrict__ sample,
                                const float* __restrict__ noise, StepCoef k, int hw, half_t* __restrict__ prev,
                                half_t* __restrict__ den, half_t* __restrict__ dec_in) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hw) return;
  half8 e = *reinterpret_cast<const half8*>(eps + (size_t)i * 8);
  half8 x = *reinterpret_cast<const half8*>(sample + (size_t)i * 8);
  half8 op = (half8){0, 0, 0, 0, 0, 0, 0, 0}, od = op, oi = op;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float xs = (float)x[c];
    float px0 = (xs - k.sb * (float)e[c]) / k.sa;
    float d = k.cout * px0 + k.cskip * xs;
    od[c] = (half_t)d;
    float pv = noise ? k.sap * d + k.sbp * noise[(size_t)c * hw + i] : d;
    op[c] = (half_t)pv;
    oi[c] = (half_t)(tanhf((float)od[c] / 3.0f) * 3.0f);
  }
  if (prev) *reinterpret_cast<half8*>(prev + (size_t)i * 8) = op;
  if (den) *reinterpret_cast<half8*>(den + (size_t)i * 8) = od;
  if (dec_in) *reinterpret_cast<half8*>(dec_in + (size_t)i * 8) = oi;
}

__global__ void lcm_step_dev_kernel(const half_t* __restrict__ eps, const half_t* __restrict__ sample,
                                    const float* __restrict__ noise, const float* __restrict__ coef, int hw,
                                    half_t* __restrict__ prev, half_t* __restrict__ den, half_t* __restrict__ dec_in) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hw) return;
  const StepCoef k = {coef[0], coef[1], coef[2], coef[3], coef[4], coef[5]};
  const size_t row = (size_t)blockIdx.y * hw + i;
  half8 e = *reinterpret_cast<const half8*>(eps + row * 8);
  half8 x = *reinterpret_cast<const half8*>(sample + row * 8);
  half8 op = (half8){0, 0, 0, 0, 0, 0, 0, 0}, od = op, oi = op;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float xs = (float)x[c];
    float px0 = (xs - k.sb * (float)e[c]) / k.sa;
    float d = k.cout * px0 + k.cskip * xs;
    od[c] = (half_t)d;
    float pv = noise ? k.sap * d + k.sbp * noise[(size_t)c * hw + i] : d;
    op[c] = (half_t)pv;
    oi[c] = (half_t)(tanhf((float)od[c] / 3.0f) * 3.0f);
  }
  if (prev) *reinterpret_cast<half8*>(prev + row * 8) = op;
  if (den) *reinterpret_cast<half8*>(den + row * 8) = od;
  if (dec_in) *reinterpret_cast<half8*>(dec_in + row * 8) = oi;
}

// AdaIN of the reference-only mode (lcm_reference_pipeline.py:593-603): per channel, re-normalise x (its own spatial
// mean / variance, population form) to the banked statistics of the reference pass.  Both statistics arrive as
// per-channel (sum, sum of squares) over `rows` pixels -- what vsd_conv_gemm's chanstat_out leaves behind.
__global__ void adain_kernel(const half_t* __restrict__ x, const float* __restrict__ st, const float* __restrict__ st_ref,
                             int rows, int c8, float eps, half_t* __restrict__ out) {
  const size_t total = (size_t)rows * c8;
  const float inv = 1.0f / (float)rows;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(q % c8) * 8;
    half8 v = *reinterpret_cast<const half8*>(x + q * 8);
    half8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float mean = st[(ch + i) * 2] * inv, var = fmaxf(st[(ch + i) * 2 + 1] * inv - mean * mean, 0.f);
      const float mean_r = st_ref[(ch + i) * 2] * inv, var_r = fmaxf(st_ref[(ch + i) * 2 + 1] * inv - mean_r * mean_r, 0.f);
      const float sd = sqrtf(fmaxf(var, eps)), sd_r = sqrtf(fmaxf(var_r, eps));
      o[i] = (half_t)((((float)v[i] - mean) / sd) * sd_r + mean_r);
    }
    *reinterpret_cast<half8*>(out + q * 8) = o;
  }
}

// CLIP text embeddings: out[i] = fp16(fp32(token_embedding[ids[i]]) + fp32(position_embedding[i]))
__global__ void embed_tokens_kernel(const long long* __restrict__ ids, const half_t* __restrict__ tok, const half_t* __restrict__ pos,
                                    int n, int c8, int vocab, half_t* __restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n * c8) return;
  const int i = q / c8, ch = (q - i * c8) * 8;
  long long id = ids[i];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const half8 a = *reinterpret_cast<const half8*>(tok + (size_t)id * c8 * 8 + ch);
  const half8 b = *reinterpret_cast<const half8*>(pos + (size_t)i * c8 * 8 + ch);
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)a[j] + (float)b[j]);
  *reinterpret_cast<half8*>(out + (size_t)i * c8 * 8 + ch) = o;
}

__global__ void postprocess_kernel(const half_t* __restrict__ img, int ld, int hw, unsigned char* __restrict__ rgb) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hw) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    half_t y = (half_t)((float)img[(size_t)i * ld + c] * 2.0f - 1.0f);   // DecoderTiny: x.mul(2).sub(1)
    half_t z = (half_t)((float)y * 0.5f + 0.5f);                          // postprocess denormalize (fp16)
    float f = fminf(fmaxf((float)z, 0.0f), 1.0f);
    rgb[(size_t)i * 3 + c] = (unsigned char)rintf(f * 255.0f);            // numpy round: half to even
  }
}

__global__ void axpy_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b, float scale, int64_t n8,
                            half_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    half8 x = reinterpret_cast<const half8*>(a)[i];
    half8 y = reinterpret_cast<const half8*>(b)[i];
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)x[j] + scale * (float)y[j]);
    reinterpret_cast<half8*>(out)[i] = o;
  }
}

}  // namespace

extern "C" int vsd_preprocess_rgb(vsd_ctx* ctx, const void* rgb_u8, int h, int w, void* out, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!rgb_u8 || !out || h <= 0 || w <= 0) return vsd_fail(ctx, VSD_ERR_ARG, "preprocess_rgb: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls(ctx, s, VSD_FAM_ELEMENTWISE, 0.0);
  hipLaunchKernelGGL(preprocess_rgb_kernel, dim3(cdiv(h * w, 256)), dim3(256), 0, s, (const unsigned char*)rgb_u8, h * w,
                     (half_t*)out);
  return ls.finish();
}

extern "C" int64_t vsd_sobel_workspace_bytes(int h, int w) { return (int64_t)cdiv(w, SOBEL_TW) * h * sizeof(float); }

extern "C" int vsd_sobel_control(vsd_ctx* ctx, const void* rgb_u8, int h, int w, float low, float high, void* edge_u8,
                                 void* control_out, void* workspace, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!rgb_u8 || !edge_u8 || !control_out || !workspace || h <= 0 || w <= 0)
    return vsd_fail(ctx, VSD_ERR_ARG, "sobel_control: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(cdiv(w, SOBEL_TW), h);
  const int nblk = (int)(grid.x * grid.y);
  {
    LaunchScope ls(ctx, s, VSD_FAM_ELEMENTWISE, 0.0);
    hipLaunchKernelGGL(sobel_max_kernel, grid, dim3(SOBEL_TW), 0, s, (const unsigned char*)rgb_u8, h, w, (float*)workspace);
    int rc = ls.finish();
    if (rc) return rc;
  }
  LaunchScope ls(ctx, s, VSD_FAM_ELEMENTWISE, 0.0);
  hipLaunchKernelGGL(sobel_apply_kernel, grid, dim3(SOBEL_TW), 0, s, (const unsigned char*)rgb_u8, h, w, low, high,
                     (const float*)workspace, nblk, (unsigned char*)edge_u8, (half_t*)control_out);
  return ls.finish();
}

extern "C" int vsd_add_noise(vsd_ctx* ctx, const void* x0, const void* noise_f32, float sqrt_a, float sqrt_b, int hw,
                             void* out, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!x0 || !noise_f32 || !out || hw <= 0) return vsd_fail(ctx, VSD_ERR_ARG, "add_noise: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls(ctx, s, VSD_FAM_ELEMENTWISE, 0.0);
  hipLaunchKernelGGL(add_noise_kernel, dim3(cdiv(hw, 256)), dim3(256), 0, s, (const half_t*)x0, (const float*)noise_f32,
                     sqrt_a, sqrt_b, hw, (half_t*)out);
  return ls.finish();
}

extern "C" int vsd_lcm_step(vsd_ctx* ctx, const void* eps, const void* sample, const void* noise_f32,
                            const float* coef_host, int hw, void* prev, void* denoised, void* dec_in, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!eps || !sample || !coef_host || hw <= 0) return vsd_fail(ctx, VSD_ERR_ARG, "lcm_step: bad arguments");
  StepCoef k = {coef_host[0], coef_host[1], coef_host[2], coef_host[3], coef_host[4], coef_host[5]};
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls(ctx, s, VSD_FAM_ELEMENTWISE, 0.0);
  hipLaunchKernelGGL(lcm_step_kernel, dim3(cdiv(hw, 256)), dim3(256), 0, s, (const half_t*)eps, (const half_t*)sample,
                     (const float*)noise_f32, k, hw, (half_t*)prev, (half_t*)denoised, (half_t*)dec_in);
  return ls.finish();
}

extern "C" int vsd_add_noise_dev(vsd_ctx* ctx, const void* x0, const void* noise_f32, const void* coef_dev, int hw, int batch,
                                 void* out, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!x0 || !noise_f32 || !coef_dev || !out || hw <= 0 || batch < 1 || batch > 65535)
    return vsd_fail(ctx, VSD_ERR_ARG, "add_noise_dev: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls(ctx, s, VSD_FAM_ELEMENTWISE, 0.0);
  hipLaunchKernelGGL(add_noise_dev_kernel, dim3(cdiv(hw, 256), batch), dim3(256), 0, s, (const half_t*)x0,
                     (const float*)noise_f32, (const float*)coef_dev, hw, (half_t*)out);
  return ls.finish();
}

extern "C" int vsd_lcm_step_dev(vsd_ctx* ctx, const void* eps, const void* sample, const void* noise_f32, const void* coef_dev,
                                int hw, int batch, void* prev, void* denoised, void* dec_in, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!eps || !sample || !coef_dev || hw <= 0 || batch < 1 || batch > 65535)
    return vsd_fail(ctx, VSD_ERR_ARG, "lcm_step_dev: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls(ctx, s, VSD_FAM_ELEMENTWISE, 0.0);
  hipLaunchKernelGGL(lcm_step_dev_kernel, dim3(cdiv(hw, 256), batch), dim3(256), 0, s, (const half_t*)eps,
                     (const half_t*)sample, (const float*)noise_f32, (const float*)coef_dev, hw, (half_t*)prev,
                     (half_t*)denoised, (half_t*)dec_in);
  return ls.finish();
}

extern "C" int vsd_adain(vsd_ctx* ctx, const void* x, const void* stats, const void* stats_ref, int rows, int c, float eps,
                         void* out, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!x || !stats || !stats_ref || !out || rows <= 0 || c <= 0 || c % 8) return vsd_fail(ctx, VSD_ERR_ARG, "adain: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)rows * (c / 8);
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  LaunchScope ls(ctx, s, VSD_FAM_ELEMENTWISE, 0.0);
  hipLaunchKernelGGL(adain_kernel, dim3(grid), dim3(256), 0, s, (const half_t*)x, (const float*)stats, (const float*)stats_ref,
                     rows, c / 8, eps, (half_t*)out);
  return ls.finish();
}

extern "C" int vsd_embed_tokens(vsd_ctx* ctx, const void* ids_i64, const void* token_emb, const void* pos_emb, int n, int c,
                                int vocab, void* out, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!ids_i64 || !token_emb || !pos_emb || !out || n <= 0 || c <= 0 || c % 8 || vocab <= 0)
    return vsd_fail(ctx, VSD_ERR_ARG, "embed_tokens: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls(ctx, s, VSD_FAM_ELEMENTWISE, 0.0);
  hipLaunchKernelGGL(embed_tokens_kernel, dim3(cdiv(n * (c / 8), 256)), dim3(256), 0, s, (const long long*)ids_i64,
                     (const half_t*)token_emb, (const half_t*)pos_emb, n, c / 8, vocab, (half_t*)out);
  return ls.finish();
}

extern "C" int vsd_postprocess_rgb(vsd_ctx* ctx, const void* img, int ld, int hw, void* rgb_u8, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!img || !rgb_u8 || hw <= 0 || ld < 3) return vsd_fail(ctx, VSD_ERR_ARG, "postprocess_rgb: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls(ctx, s, VSD_FAM_ELEMENTWISE, 0.0);
  hipLaunchKernelGGL(postprocess_kernel, dim3(cdiv(hw, 256)), dim3(256), 0, s, (const half_t*)img, ld, hw,
                     (unsigned char*)rgb_u8);
  return ls.finish();
}

extern "C" int vsd_axpy(vsd_ctx* ctx, const void* a, const void* b, float scale, int64_t n, void* out, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!a || !b || !out || n <= 0 || n % 8) return vsd_fail(ctx, VSD_ERR_ARG, "axpy: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  int64_t n8 = n / 8;
  int grid = (int)((n8 + 255) / 256);
  if (grid > 2048) grid = 2048;
  LaunchScope ls(ctx, s, VSD_FAM_ELEMENTWISE, 0.0);
  hipLaunchKernelGGL(axpy_kernel, dim3(grid), dim3(256), 0, s, (const half_t*)a, (const half_t*)b, scale, n8, (half_t*)out);
  return ls.finish();
}

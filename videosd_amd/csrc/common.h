// Shared host/device helpers for libvsd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/vsd.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_ __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ProfEvent {
  hipEvent_t a, b;
  int fam;
};

struct vsd_ctx {
  int device;
  std::string err;
  bool profiling = false;
  bool capturing = false;
  std::vector<ProfEvent> events;
  void* zero_page = nullptr;  // 4 KiB of zeros in HBM (out-of-bounds source for LDS-DMA loads)
  double fam_flops[VSD_FAM_COUNT];
  int64_t fam_launch[VSD_FAM_COUNT];
};

static inline int vsd_fail(vsd_ctx* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
static inline int vsd_fail(vsd_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

#define VSD_HIP(ctx, expr)                                                                              \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess) return vsd_fail(ctx, VSD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));    \
  } while (0)

// RAII bracket: records events around a launch when profiling, and checks the launch error.
struct LaunchScope {
  vsd_ctx* ctx;
  hipStream_t s;
  int fam;
  hipEvent_t a = nullptr, b = nullptr;
  LaunchScope(vsd_ctx* c, hipStream_t st, int f, double flops) : ctx(c), s(st), fam(f) {
    if (ctx->profiling) {
      (void)hipEventCreate(&a);
      (void)hipEventCreate(&b);
      (void)hipEventRecord(a, s);
      ctx->fam_flops[fam] += flops;
      ctx->fam_launch[fam] += 1;
    }
  }
  int finish() {
    hipError_t e = hipGetLastError();
    if (ctx->profiling) {
      (void)hipEventRecord(b, s);
      ctx->events.push_back({a, b, fam});
    }
    if (e != hipSuccess) return vsd_fail(ctx, VSD_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return VSD_OK;
  }
};

// (rcp instead of an IEEE division: 1 ulp, far below the fp16 rounding of the result, and 10 instructions fewer per element)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 rounding of the result): one rcp, one exp
// and five FMAs instead of libdevice's branchy erff -- the GEGLU epilogue evaluates this for 15 M elements per launch.
__device__ __forceinline__ float erf_as_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as_f(x * 0.70710678118654752f)); }
__device__ __forceinline__ float quick_gelu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

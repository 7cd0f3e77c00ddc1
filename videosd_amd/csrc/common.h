// Shared host/device helpers for libvsd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <type_traits>
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

// ---- two independent launches of ONE kernel as one grid (include/vsd.h vsd_pair_begin): the first operation's launches are
// held here, the second operation's launches join them one by one -- the lower half of gridDim.z runs the held argument block, the upper half the joining one
template <class P>
struct Pair {
  P p[2];
};
struct PairHeld {
  const void* pair_kernel;  // identity of the two-problem kernel this launch may join
  const void* kernel;       // ... and the one-problem kernel, for launching it alone after all
  void (*alone)(const void* kernel, const void* params, dim3 grid, dim3 block, unsigned shmem, hipStream_t s);
  dim3 grid, block;
  unsigned shmem, size;
  hipStream_t stream;
  alignas(16) unsigned char params[480];
};

struct vsd_ctx {
  int device;
  int pair_state = 0;  // 0: launches go out as they come; 1: hold (first operation); 2: join (second operation); 3: pass through
  std::vector<PairHeld> pair_held;
  size_t pair_next = 0;
  int pair_joined = 0;
  std::string err;
  bool profiling = false;
  bool capturing = false;
  std::vector<ProfEvent> events;
  void* zero_page = nullptr;  // 4 KiB of zeros in HBM (out-of-bounds source for LDS-DMA loads)
  int num_cus = 256;          // compute units of the device (grid of the persistent kernels)
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

template <class P>
static void pair_launch_alone(const void* kernel, const void* params, dim3 grid, dim3 block, unsigned shmem, hipStream_t s) {
  hipLaunchKernelGGL(reinterpret_cast<void (*)(const P)>(const_cast<void*>(kernel)), grid, block, shmem, s, *reinterpret_cast<const P*>(params));
}

// Launch `k1(p)`, or -- between vsd_pair_begin and vsd_pair_end -- hold it / join it with the held launch of the same kernel,
// launch geometry and stream as ONE grid of `k2` (gridDim.z doubled: its upper half is the joining problem).  A launch that finds no partner goes out alone, in order.
template <class P>
static inline void launch_pairable(vsd_ctx* ctx, void (*k1)(const P), void (*k2)(const Pair<P>), dim3 grid, dim3 block, unsigned shmem,
                                   hipStream_t s, const P& p) {
  static_assert(sizeof(P) <= sizeof(PairHeld::params) && std::is_trivially_copyable<P>::value, "argument block too large to hold");
  if (ctx->pair_state == 0 || ctx->pair_state == 3) {
    hipLaunchKernelGGL(k1, grid, block, shmem, s, p);
    return;
  }
  if (ctx->pair_state == 1) {
    PairHeld h;
    h.pair_kernel = reinterpret_cast<const void*>(k2);
    h.kernel = reinterpret_cast<const void*>(k1);
    h.alone = &pair_launch_alone<P>;
    h.grid = grid; h.block = block; h.shmem = shmem; h.size = sizeof(P); h.stream = s;
    memcpy(h.params, &p, sizeof(P));
    ctx->pair_held.push_back(h);
    return;
  }
  if (ctx->pair_next < ctx->pair_held.size()) {
    const PairHeld& h = ctx->pair_held[ctx->pair_next++];
    if (h.pair_kernel == reinterpret_cast<const void*>(k2) && h.grid.x == grid.x && h.grid.y == grid.y && h.grid.z == grid.z && h.block.x == block.x &&
        h.block.y == block.y && h.block.z == block.z && h.shmem == shmem && h.stream == s) {
      Pair<P> g;
      memcpy(&g.p[0], h.params, sizeof(P));
      g.p[1] = p;
      hipLaunchKernelGGL(k2, dim3(grid.x, grid.y, 2 * grid.z), block, shmem, s, g);
      ctx->pair_joined += 1;
      return;
    }
    h.alone(h.kernel, h.params, h.grid, h.block, h.shmem, h.stream);
  }
  hipLaunchKernelGGL(k1, grid, block, shmem, s, p);
}

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

// Output stores of the conv epilogues.  -DVSD_NT_STORES (experiment builds, scripts/r6_*): as non-temporal stores.
#ifdef VSD_NT_STORES
#define VSD_OUT_STORE8(PTR_, VAL_) __builtin_nontemporal_store((VAL_), reinterpret_cast<half8*>(PTR_))
#define VSD_OUT_AUX 2
#else
#define VSD_OUT_STORE8(PTR_, VAL_) (*reinterpret_cast<half8*>(PTR_) = (VAL_))
#define VSD_OUT_AUX 0
#endif

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Division by a launch constant without the ~40-instruction runtime integer division (five to eight of them sat in the
// prologue of every conv workgroup: block -> tile, row -> (image, y, x)).  q = (umulhi(x, mul) + x) >> shr, exact for
// 0 <= x < 2^31 and 1 <= d < 2^31 (the round-up multiplier of Granlund / Montgomery; d = 1 gives mul = 1, shr = 0).
struct FastDiv {
  unsigned mul, shr;
};
static inline FastDiv fast_div(unsigned d) {
  FastDiv f;
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  f.shr = s;
  f.mul = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
  return f;
}
__device__ __forceinline__ int fdiv(int x, FastDiv f) { return (int)((__umulhi((unsigned)x, f.mul) + (unsigned)x) >> f.shr); }

// Workgroup timeline (development builds only: -DVSD_WG_TIMELINE, videosd_amd.build.build_timeline): every workgroup of the
// instrumented kernels leaves its start / main-loop-done / end time (s_memrealtime, 100 MHz) and where it ran (HW_ID,
// XCC_ID) in a log buffer, one region per launch [grid, kind, 8 words per workgroup ...], so that a launch's duration can
// be split into dispatch ramp, workgroup life and tail (scripts/wg_timeline.py).  Never compiled into libvsd.so.
#ifdef VSD_WG_TIMELINE
inline unsigned long long* g_wgtl = nullptr;
inline size_t g_wgtl_off = 0, g_wgtl_cap = 0;
static inline unsigned long long* wgtl_claim(int grid) {  // host: the next launch's region (nullptr: logging off / log full)
  if (!g_wgtl || g_wgtl_off + 2 + 8 * (size_t)grid > g_wgtl_cap) return nullptr;
  unsigned long long* r = g_wgtl + g_wgtl_off + 2;
  g_wgtl_off += 2 + 8 * (size_t)grid;
  return r;
}
#define WGTL_START() const unsigned long long wgtl_t0 = __builtin_amdgcn_s_memrealtime(); unsigned long long wgtl_t1 = 0, wgtl_ta = 0, wgtl_tb = 0, wgtl_tc = 0;
#define WGTL_LOOP() wgtl_t1 = __builtin_amdgcn_s_memrealtime();
#define WGTL_MARK(W_) wgtl_t##W_ = __builtin_amdgcn_s_memrealtime();  /* W_ = a, b or c: further points inside the epilogue */
/* the tile epilogue is a function of its own (conv_tile_epilogue): its marks a / b are the kernel's variables, by reference */
#define WGTL_EPI_PARAM , unsigned long long& wgtl_ta, unsigned long long& wgtl_tb
#define WGTL_EPI_ARG , wgtl_ta, wgtl_tb
#define WGTL_EPI_MARK(W_) wgtl_t##W_ = __builtin_amdgcn_s_memrealtime();
#define WGTL_END(KIND_)                                                                                   \
  if (p.wgtl && threadIdx.x == 0) {                                                                       \
    unsigned long long* d_ = p.wgtl + 8 * (size_t)blockIdx.x;                                             \
    d_[0] = wgtl_t0; d_[1] = wgtl_t1; d_[2] = __builtin_amdgcn_s_memrealtime();                           \
    d_[4] = wgtl_ta; d_[5] = wgtl_tb; d_[6] = wgtl_tc;                                                    \
    d_[3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |                               \
            ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);                       \
    if (blockIdx.x == 0) { p.wgtl[-2] = gridDim.x; p.wgtl[-1] = (KIND_); }                                \
  }
// Workgroup-seconds per kernel family under ANY load (round 5; scripts/wg_cu_time.py): every workgroup of the conv / reducer /
// GroupNorm / attention / fused-tail kernels adds its life (first instruction -> thread 0's exit, 10 ns ticks), a count and
// life x waves to per-family counters in a device buffer the host hands over (vsd_cut_set; the launchers copy the pointer into
// the kernel's argument block).  Unlike the per-launch log above this works under graph replay with several launch lanes busy: what
// a kernel family OCCUPIES of the chip while the timed 5 x 4 program runs, not what a launch takes alone.
enum { VSD_CUT_CONV_GEMM = 0, VSD_CUT_CONV_HALO = 1, VSD_CUT_REDUCE = 2, VSD_CUT_GROUPNORM = 3, VSD_CUT_ATTENTION = 4, VSD_CUT_TAIL = 5,
       VSD_CUT_FAMS = 8 };
inline unsigned long long* g_cut = nullptr;  // [3 * VSD_CUT_FAMS]: per family ticks, workgroups, ticks x waves (device memory)
struct CuTimer {
  unsigned long long t0;
  unsigned long long* buf;
  int fam;
  __device__ __forceinline__ CuTimer(int f, unsigned long long* b) : t0(__builtin_amdgcn_s_memrealtime()), buf(b), fam(f) {}
  __device__ __forceinline__ ~CuTimer() {
    if (buf && threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
      const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t0;
      const unsigned long long waves = (blockDim.x * blockDim.y * blockDim.z + 63) / 64;
      atomicAdd(buf + 3 * fam, dt);
      atomicAdd(buf + 3 * fam + 1, 1ull);
      atomicAdd(buf + 3 * fam + 2, dt * waves);
    }
  }
};
#define VSD_CUT(FAM_, PTR_) CuTimer cut_timer_(FAM_, PTR_);
#define VSD_CUT_FIELD unsigned long long* cut;
#define VSD_CUT_SET(P_) (P_).cut = g_cut;
#else
#define VSD_CUT(FAM_, PTR_)
#define VSD_CUT_FIELD
#define VSD_CUT_SET(P_)
#define WGTL_START()
#define WGTL_LOOP()
#define WGTL_MARK(W_)
#define WGTL_EPI_PARAM
#define WGTL_EPI_ARG
#define WGTL_EPI_MARK(W_)
#define WGTL_END(KIND_)
#endif

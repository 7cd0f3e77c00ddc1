// Context, hipGraph capture/replay and per-family timing for libvsd.
#include <stdarg.h>

#include "common.h"

extern "C" int vsd_version(void) { return VSD_VERSION; }
extern "C" int vsd_conv_desc_size(void) { return (int)sizeof(vsd_conv_desc); }

extern "C" vsd_ctx* vsd_create(int device_id) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return nullptr;
  if (hipSetDevice(device_id) != hipSuccess) return nullptr;
  vsd_ctx* c = new vsd_ctx();
  c->device = device_id;
  if (hipMalloc(&c->zero_page, 4096) != hipSuccess || hipMemset(c->zero_page, 0, 4096) != hipSuccess) {
    delete c;
    return nullptr;
  }
  for (int i = 0; i < VSD_FAM_COUNT; ++i) {
    c->fam_flops[i] = 0;
    c->fam_launch[i] = 0;
  }
  return c;
}

static void drop_events(vsd_ctx* ctx) {
  for (auto& e : ctx->events) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  ctx->events.clear();
}

extern "C" void vsd_destroy(vsd_ctx* ctx) {
  if (!ctx) return;
  drop_events(ctx);
  if (ctx->zero_page) (void)hipFree(ctx->zero_page);
  delete ctx;
}

extern "C" const char* vsd_last_error(vsd_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int vsd_graph_begin(vsd_ctx* ctx, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (ctx->capturing) return vsd_fail(ctx, VSD_ERR_STATE, "graph_begin: already capturing");
  if (ctx->profiling) return vsd_fail(ctx, VSD_ERR_STATE, "graph_begin: profiling is on");
  VSD_HIP(ctx, hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  ctx->capturing = true;
  return VSD_OK;
}

extern "C" int vsd_graph_end(vsd_ctx* ctx, void* stream, void** graph_exec_out) {
  if (!ctx || !graph_exec_out) return VSD_ERR_ARG;
  if (!ctx->capturing) return vsd_fail(ctx, VSD_ERR_STATE, "graph_end: not capturing");
  ctx->capturing = false;
  hipGraph_t g = nullptr;
  VSD_HIP(ctx, hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return vsd_fail(ctx, VSD_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
  *graph_exec_out = (void*)ge;
  return VSD_OK;
}

extern "C" int vsd_graph_launch(vsd_ctx* ctx, void* graph_exec, void* stream) {
  if (!ctx || !graph_exec) return VSD_ERR_ARG;
  VSD_HIP(ctx, hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return VSD_OK;
}

extern "C" int vsd_graph_destroy(vsd_ctx* ctx, void* graph_exec) {
  if (!ctx || !graph_exec) return VSD_ERR_ARG;
  VSD_HIP(ctx, hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return VSD_OK;
}

extern "C" int vsd_profile_begin(vsd_ctx* ctx) {
  if (!ctx) return VSD_ERR_ARG;
  if (ctx->capturing) return vsd_fail(ctx, VSD_ERR_STATE, "profile_begin: capturing a graph");
  drop_events(ctx);
  for (int i = 0; i < VSD_FAM_COUNT; ++i) {
    ctx->fam_flops[i] = 0;
    ctx->fam_launch[i] = 0;
  }
  ctx->profiling = true;
  return VSD_OK;
}

extern "C" int vsd_profile_end(vsd_ctx* ctx) {
  if (!ctx) return VSD_ERR_ARG;
  ctx->profiling = false;
  return VSD_OK;
}

extern "C" int vsd_stage_times(vsd_ctx* ctx, float* ms, int64_t* launches, double* flops) {
  if (!ctx || !ms) return VSD_ERR_ARG;
  for (int i = 0; i < VSD_FAM_COUNT; ++i) ms[i] = 0.f;
  for (auto& e : ctx->events) {
    VSD_HIP(ctx, hipEventSynchronize(e.b));
    float t = 0.f;
    VSD_HIP(ctx, hipEventElapsedTime(&t, e.a, e.b));
    ms[e.fam] += t;
  }
  for (int i = 0; i < VSD_FAM_COUNT; ++i) {
    if (launches) launches[i] = ctx->fam_launch[i];
    if (flops) flops[i] = ctx->fam_flops[i];
  }
  return VSD_OK;
}

// Average elapsed time (ms) of an EMPTY event bracket on `stream`: what the per-launch HIP-event timing of
// vsd_stage_times adds to every kernel it brackets (bench.py subtracts launches * this from a family's total).
extern "C" int vsd_profile_overhead(vsd_ctx* ctx, void* stream, int n, float* ms_out) {
  if (!ctx || !ms_out || n <= 0) return VSD_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  std::vector<hipEvent_t> ev(2 * n);
  for (auto& e : ev) VSD_HIP(ctx, hipEventCreate(&e));
  for (int i = 0; i < n; ++i) {
    VSD_HIP(ctx, hipEventRecord(ev[2 * i], s));
    VSD_HIP(ctx, hipEventRecord(ev[2 * i + 1], s));
  }
  VSD_HIP(ctx, hipStreamSynchronize(s));
  double tot = 0;
  for (int i = 0; i < n; ++i) {
    float t = 0.f;
    VSD_HIP(ctx, hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]));
    tot += t;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  *ms_out = (float)(tot / n);
  return VSD_OK;
}

// Context, hipGraph capture/replay and per-family timing for libvsd.
#include <stdarg.h>
#include <stdlib.h>

#include <chrono>
#include <mutex>

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
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) c->num_cus = cus;
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

// ---- two operations as one grid per kernel (common.h launch_pairable)
static void pair_flush(vsd_ctx* ctx) {
  for (; ctx->pair_next < ctx->pair_held.size(); ++ctx->pair_next) {
    const PairHeld& h = ctx->pair_held[ctx->pair_next];
    h.alone(h.kernel, h.params, h.grid, h.block, h.shmem, h.stream);
  }
  ctx->pair_held.clear();
  ctx->pair_next = 0;
}

extern "C" int vsd_pair_begin(vsd_ctx* ctx) {
  if (!ctx) return VSD_ERR_ARG;
  if (ctx->pair_state != 0) return vsd_fail(ctx, VSD_ERR_STATE, "pair_begin: a pair is already open");
  // (while profiling, per-family events bracket single launches: the pair is accepted and its launches go out one by one)
  ctx->pair_state = ctx->profiling ? 3 : 1;
  ctx->pair_joined = 0;
  return VSD_OK;
}

extern "C" int vsd_pair_join(vsd_ctx* ctx) {
  if (!ctx) return VSD_ERR_ARG;
  if (ctx->pair_state != 1 && ctx->pair_state != 3) return vsd_fail(ctx, VSD_ERR_STATE, "pair_join: no pair_begin before it");
  if (ctx->pair_state == 1) ctx->pair_state = 2;
  return VSD_OK;
}

extern "C" int vsd_pair_end(vsd_ctx* ctx, int* joined_out) {
  if (!ctx) return VSD_ERR_ARG;
  if (ctx->pair_state == 0) return vsd_fail(ctx, VSD_ERR_STATE, "pair_end: no pair is open");
  pair_flush(ctx);  // (held launches that found no partner, or everything when the second operation never came)
  ctx->pair_state = 0;
  if (joined_out) *joined_out = ctx->pair_joined;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vsd_fail(ctx, VSD_ERR_HIP, "pair_end: kernel launch failed: %s", hipGetErrorString(e));
  return VSD_OK;
}

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

// ---- streams with a hardware queue AND a command-processor pipe of their own ----------------------------------------
// Measured on MI355X (scripts/queue_probe.cpp, scripts/pipe_probe.cpp; DESIGN.md section 3 "launches in flight"):
//  * plain HIP streams share GPU_MAX_HW_QUEUES (4) hardware queues in creation order, the null stream and every stream any
//    library of the process made included (ten streams -> queues 0 1 2 2 1 0 3 2 1 0): which launches in flight alias is
//    an accident of history;
//  * a stream created with a CU mask (also the mask of all CUs) gets a hardware queue of its own;
//  * but the command processor serves those queues through 4 pipes, queue i of a process sits on pipe (i mod 4), and two
//    BUSY queues on one pipe take turns in long slices: two chains of 100 dependent kernels on queues 0 and 4 take 2.55x
//    the time of one chain (worse than back to back), on queues 0 and 1 1.00x; four chains on queues {0,1,2,3} 1.02x.
// So a process gets exactly four launch streams per device, created together (consecutive queues = four different pipes),
// kept for its lifetime, and everything that can be in flight at once is placed on different ones of them.
__global__ void vsd_spin_kernel(unsigned long long ticks) {
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
  }
}

static std::mutex g_pool_mu;
static hipStream_t g_pool[VSD_MAX_DEVICES][VSD_POOL_STREAMS];
static bool g_pool_made[VSD_MAX_DEVICES];

static int full_cu_mask(vsd_ctx* ctx, std::vector<uint32_t>& all) {
  hipDeviceProp_t prop;
  VSD_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
  all.assign((size_t)(prop.multiProcessorCount + 31) / 32, 0xffffffffu);
  if (prop.multiProcessorCount % 32) all.back() = (1u << (prop.multiProcessorCount % 32)) - 1u;
  return VSD_OK;
}

extern "C" int vsd_stream_create(vsd_ctx* ctx, const uint32_t* cu_mask, int words, void** stream_out) {
  if (!ctx || !stream_out || (cu_mask && words <= 0)) return VSD_ERR_ARG;
  VSD_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<uint32_t> all;
  if (!cu_mask) {
    int rc = full_cu_mask(ctx, all);
    if (rc != VSD_OK) return rc;
    cu_mask = all.data();
    words = (int)all.size();
  }
  hipStream_t s = nullptr;
  VSD_HIP(ctx, hipExtStreamCreateWithCUMask(&s, (uint32_t)words, cu_mask));
  *stream_out = (void*)s;
  return VSD_OK;
}

extern "C" int vsd_stream_destroy(vsd_ctx* ctx, void* stream) {
  if (!ctx || !stream) return VSD_ERR_ARG;
  VSD_HIP(ctx, hipStreamDestroy((hipStream_t)stream));
  return VSD_OK;
}

// The pool's streams are destroyed at process exit, BEFORE the HIP runtime's own teardown (atexit handlers run in reverse
// order of registration, and the runtime registered its own long before the pool exists): left to the runtime, CU-masked
// queues were torn down in an order that crashed rocprofv3's finalisation (SIGSEGV in __cxa_finalize under --kernel-trace).
static void pool_destroy_at_exit() {
  for (int d = 0; d < VSD_MAX_DEVICES; ++d) {
    if (!g_pool_made[d]) continue;
    if (hipSetDevice(d) != hipSuccess) continue;
    for (int i = 0; i < VSD_POOL_STREAMS; ++i) {
      (void)hipStreamSynchronize(g_pool[d][i]);
      (void)hipStreamDestroy(g_pool[d][i]);
    }
    g_pool_made[d] = false;
  }
}

extern "C" int vsd_stream_pool(vsd_ctx* ctx, void** streams_out) {
  if (!ctx || !streams_out) return VSD_ERR_ARG;
  if (ctx->device < 0 || ctx->device >= VSD_MAX_DEVICES) return vsd_fail(ctx, VSD_ERR_ARG, "stream_pool: device %d", ctx->device);
  std::lock_guard<std::mutex> lock(g_pool_mu);
  if (!g_pool_made[ctx->device]) {
    VSD_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> all;
    int rc = full_cu_mask(ctx, all);
    if (rc != VSD_OK) return rc;
    // VSD_POOL_MASK (experiments, scripts/mask_probe.cpp): "xcd" = stream i gets the mask bits b with b % 8 in {2i, 2i+1},
    // "blk" = the i-th quarter of the bits; default: every stream may use every CU
    const char* mode = getenv("VSD_POOL_MASK");
    const int ncu = ctx->num_cus > 0 ? ctx->num_cus : (int)all.size() * 32;
    for (int i = 0; i < VSD_POOL_STREAMS; ++i) {
      std::vector<uint32_t> m = all;
      if (mode && (mode[0] == 'x' || mode[0] == 'b')) {
        std::fill(m.begin(), m.end(), 0u);
        for (int b = 0; b < ncu; ++b) {
          const bool on = mode[0] == 'x' ? (b % 8) / 2 == i : b * VSD_POOL_STREAMS / ncu == i;
          if (on) m[(size_t)b / 32] |= 1u << (b % 32);
        }
      }
      VSD_HIP(ctx, hipExtStreamCreateWithCUMask(&g_pool[ctx->device][i], (uint32_t)m.size(), m.data()));
    }
    g_pool_made[ctx->device] = true;
    static bool registered = false;
    if (!registered) {
      registered = true;
      atexit(pool_destroy_at_exit);
    }
  }
  for (int i = 0; i < VSD_POOL_STREAMS; ++i) streams_out[i] = (void*)g_pool[ctx->device][i];
  return VSD_OK;
}

// Self-test of the placement: a chain of `chain` dependent 10 us kernels on every pool stream at once against one chain
// alone.  ~1.0 = the four streams run side by side; >= 2 = two of them share a pipe (another CU-masked stream was created
// between ours, or the runtime's mapping changed): callers report it (bench.py) or refuse to go on (tests).
extern "C" int vsd_stream_pool_check(vsd_ctx* ctx, int chain, float* ratio_out) {
  if (!ctx || !ratio_out || chain <= 0) return VSD_ERR_ARG;
  void* st[VSD_POOL_STREAMS];
  int rc = vsd_stream_pool(ctx, st);
  if (rc != VSD_OK) return rc;
  auto run = [&](int k, double* us) -> int {
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      for (int i = 0; i < k; ++i) VSD_HIP(ctx, hipStreamSynchronize((hipStream_t)st[i]));
      auto t0 = std::chrono::steady_clock::now();
      for (int j = 0; j < chain; ++j)
        for (int i = 0; i < k; ++i) hipLaunchKernelGGL(vsd_spin_kernel, dim3(64), dim3(64), 0, (hipStream_t)st[i], 1000ull);
      for (int i = 0; i < k; ++i) VSD_HIP(ctx, hipStreamSynchronize((hipStream_t)st[i]));
      double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (t < best) best = t;
    }
    *us = best;
    return VSD_OK;
  };
  double one = 0, all = 0;
  if ((rc = run(1, &one)) != VSD_OK || (rc = run(VSD_POOL_STREAMS, &all)) != VSD_OK) return rc;
  *ratio_out = (float)(all / one);
  return VSD_OK;
}

// ---- launch sequences ----------------------------------------------------------------------------------------------
// A frame's program = single-branch graphs on the caller's streams + event edges between those streams, issued in order.
// (Why not one graph with parallel branches: this runtime executes a forked graph on device-wide internal streams, and two
// forked graph executables in flight run one after the other -- queue_probe: 2 x (2 branches of 300 us) = 675 us, three:
// 1039 us -- while single-branch graphs on streams of different queues overlap fully, 4 x 300 us in 328 us, and the fork
// written with events outside the graphs costs the same ~50 us as the runtime's own.)
struct vsd_seq_item {
  int kind;  // 0 graph, 1 record, 2 wait
  hipGraphExec_t graph;
  hipStream_t stream;
  int event;
};
struct vsd_seq {
  std::vector<vsd_seq_item> items;
  std::vector<hipEvent_t> events;
};

extern "C" int vsd_seq_create(vsd_ctx* ctx, vsd_seq** seq_out) {
  if (!ctx || !seq_out) return VSD_ERR_ARG;
  *seq_out = new vsd_seq();
  return VSD_OK;
}

extern "C" int vsd_seq_add_graph(vsd_ctx* ctx, vsd_seq* seq, void* graph_exec, void* stream) {
  if (!ctx || !seq || !graph_exec) return VSD_ERR_ARG;
  seq->items.push_back({0, (hipGraphExec_t)graph_exec, (hipStream_t)stream, -1});
  return VSD_OK;
}

extern "C" int vsd_seq_add_record(vsd_ctx* ctx, vsd_seq* seq, void* stream, int* event_out) {
  if (!ctx || !seq || !event_out) return VSD_ERR_ARG;
  hipEvent_t e = nullptr;
  VSD_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  seq->events.push_back(e);
  *event_out = (int)seq->events.size() - 1;
  seq->items.push_back({1, nullptr, (hipStream_t)stream, *event_out});
  return VSD_OK;
}

extern "C" int vsd_seq_add_wait(vsd_ctx* ctx, vsd_seq* seq, void* stream, int event) {
  if (!ctx || !seq) return VSD_ERR_ARG;
  if (event < 0 || event >= (int)seq->events.size()) return vsd_fail(ctx, VSD_ERR_ARG, "seq_add_wait: no such event %d", event);
  seq->items.push_back({2, nullptr, (hipStream_t)stream, event});
  return VSD_OK;
}

extern "C" int vsd_seq_count(vsd_ctx* ctx, vsd_seq* seq, int* graphs, int* edges) {
  if (!ctx || !seq) return VSD_ERR_ARG;
  int g = 0, w = 0;
  for (auto& it : seq->items) {
    g += it.kind == 0;
    w += it.kind == 2;
  }
  if (graphs) *graphs = g;
  if (edges) *edges = w;
  return VSD_OK;
}

extern "C" int vsd_seq_launch(vsd_ctx* ctx, vsd_seq* seq) {
  if (!ctx || !seq) return VSD_ERR_ARG;
  for (auto& it : seq->items) {
    if (it.kind == 0) {
      VSD_HIP(ctx, hipGraphLaunch(it.graph, it.stream));
    } else if (it.kind == 1) {
      VSD_HIP(ctx, hipEventRecord(seq->events[it.event], it.stream));
    } else {
      VSD_HIP(ctx, hipStreamWaitEvent(it.stream, seq->events[it.event], 0));
    }
  }
  return VSD_OK;
}

extern "C" int vsd_seq_destroy(vsd_ctx* ctx, vsd_seq* seq) {
  if (!ctx || !seq) return VSD_ERR_ARG;
  for (auto& it : seq->items)
    if (it.kind == 0) (void)hipGraphExecDestroy(it.graph);
  for (auto e : seq->events) (void)hipEventDestroy(e);
  delete seq;
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

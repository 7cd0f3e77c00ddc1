// A prepared frame program from a file, for hosts without Python (include/vsd.h vsd_plan_*; format and export: videosd_amd/plan.py).
//
// SURVEY.md section 8b sketched whole-frame C entry points; a frame's sequencing lives in engine.py, so what a non-Python host gets
// is the RECORDED program: every C-ABI call of the engine's one-stream form with its arguments, device pointers as (region, offset).
// vsd_plan_load allocates the regions, uploads the saved ones (weights, constants, prompt block, counters), patches the pointers,
// replays the calls under stream capture and instantiates the graph; vsd_plan_infer = upload the frame(s), one graph launch,
// download.  Same kernels with the same arguments as the Python engine: the same bits (tests/test_plan_gpu.py).
#include <stdarg.h>
#include <algorithm>

#include "common.h"

namespace {

enum { T_I32 = 0, T_F32 = 1, T_PTR = 2, T_NULL = 3, T_STREAM = 4, T_DESC = 5 };
constexpr int PLAN_MAX_ARGS = 24;

union PlanArg {
  void* p;
  int i;
  float f;
};

struct PlanCall {
  int fn, n;
  PlanArg a[PLAN_MAX_ARGS];
};

}  // namespace

struct vsd_plan {
  vsd_ctx* ctx = nullptr;
  int H = 0, W = 0, batch = 0;
  std::vector<void*> regions;
  std::vector<std::vector<unsigned char>> blobs;  // descriptor arrays of the calls (the calls point into them: a moved vector keeps its buffer)
  std::vector<PlanCall> calls;
  hipStream_t stream = nullptr;
  bool own_stream = true;  // (a launch lane's stream belongs to the process-wide pool)
  void* graph = nullptr;
  void *in = nullptr, *out = nullptr, *prompt = nullptr;
  size_t io_bytes = 0, prompt_bytes = 0;
};

namespace {

int plan_dispatch(vsd_ctx* ctx, int fn, int n, const PlanArg* a) {
  switch (fn) {
#include "plan_dispatch.inc"
    default: return -1;
  }
}

struct Reader {
  FILE* f;
  bool ok = true;
  template <class T>
  T get() {
    T v{};
    if (ok && fread(&v, sizeof(T), 1, f) != 1) ok = false;
    return v;
  }
  bool bytes(void* dst, size_t n) {
    if (ok && n && fread(dst, 1, n, f) != n) ok = false;
    return ok;
  }
};

void plan_release(vsd_plan* p) {
  if (!p) return;
  if (p->graph) (void)hipGraphExecDestroy((hipGraphExec_t)p->graph);
  if (p->stream && p->own_stream) (void)hipStreamDestroy(p->stream);
  for (void* r : p->regions)
    if (r) (void)hipFree(r);
  delete p;
}

}  // namespace

extern "C" int vsd_plan_load(vsd_ctx* ctx, const char* path, vsd_plan** plan_out) { return vsd_plan_load_lane(ctx, path, -1, plan_out); }

// lane >= 0: the plan launches on launch stream `lane` of the process's pool (vsd_stream_pool: four CU-masked streams = four hardware
// queues on four command-processor pipes) -- several plans in flight side by side, as the Python workers' launch lanes
extern "C" int vsd_plan_load_lane(vsd_ctx* ctx, const char* path, int lane, vsd_plan** plan_out) {
  if (!ctx || !path || !plan_out) return VSD_ERR_ARG;
  if (lane >= VSD_POOL_STREAMS) return vsd_fail(ctx, VSD_ERR_ARG, "plan_load: lane %d (0..%d, or -1 for a stream of the plan's own)", lane, VSD_POOL_STREAMS - 1);
  *plan_out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) return vsd_fail(ctx, VSD_ERR_ARG, "plan_load: cannot open %s", path);
  Reader r{f};
  char magic[8];
  r.bytes(magic, 8);
  if (!r.ok || memcmp(magic, "VSDPLAN1", 8) != 0) {
    fclose(f);
    return vsd_fail(ctx, VSD_ERR_ARG, "plan_load: %s is not a plan file", path);
  }
  const uint32_t version = r.get<uint32_t>();
  vsd_plan* p = new vsd_plan;
  p->ctx = ctx;
  p->H = (int)r.get<uint32_t>();
  p->W = (int)r.get<uint32_t>();
  p->batch = (int)r.get<uint32_t>();
  const uint32_t nreg = r.get<uint32_t>(), ncall = r.get<uint32_t>();
  const uint32_t in_r = r.get<uint32_t>();
  const uint64_t in_off = r.get<uint64_t>();
  const uint32_t out_r = r.get<uint32_t>();
  const uint64_t out_off = r.get<uint64_t>();
  const uint32_t pr_r = r.get<uint32_t>();
  const uint64_t pr_off = r.get<uint64_t>(), pr_bytes = r.get<uint64_t>();
  auto fail = [&](const char* what) {
    fclose(f);
    plan_release(p);
    return vsd_fail(ctx, VSD_ERR_ARG, "plan_load: %s (%s)", what, path);
  };
  if (!r.ok || version != 1 || nreg == 0 || nreg > (1u << 20) || ncall == 0 || ncall > (1u << 22) || in_r >= nreg || out_r >= nreg)
    return fail("bad header");
  std::vector<uint64_t> size(nreg);
  std::vector<uint32_t> saved(nreg);
  for (uint32_t i = 0; i < nreg; ++i) {
    size[i] = r.get<uint64_t>();
    saved[i] = r.get<uint32_t>();
    (void)r.get<uint32_t>();
  }
  if (!r.ok) return fail("truncated region table");
  (void)hipSetDevice(ctx->device);
  p->regions.assign(nreg, nullptr);
  for (uint32_t i = 0; i < nreg; ++i) {
    if (hipMalloc(&p->regions[i], size[i] ? size[i] : 16) != hipSuccess) return fail("out of device memory");
    if (!saved[i] && hipMemset(p->regions[i], 0, size[i]) != hipSuccess) return fail("hipMemset failed");
  }
  if (lane >= 0) {
    void* pool[VSD_POOL_STREAMS];
    if (vsd_stream_pool(ctx, pool) != VSD_OK) return fail("no launch stream pool");
    p->stream = (hipStream_t)pool[lane];
    p->own_stream = false;
  } else if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) {
    return fail("hipStreamCreate failed");
  }
  auto at = [&](uint32_t reg, uint64_t off, void** out) {
    if (reg >= nreg || off >= size[reg]) return false;
    *out = (char*)p->regions[reg] + off;
    return true;
  };
  p->calls.resize(ncall);
  for (uint32_t c = 0; c < ncall; ++c) {
    PlanCall& pc = p->calls[c];
    pc.fn = (int)r.get<uint32_t>();
    pc.n = (int)r.get<uint32_t>();
    if (!r.ok || pc.n < 0 || pc.n > PLAN_MAX_ARGS) return fail("bad call record");
    for (int k = 0; k < pc.n; ++k) {
      const uint32_t tag = r.get<uint32_t>(), aux = r.get<uint32_t>();
      const uint64_t val = r.get<uint64_t>();
      if (!r.ok) return fail("truncated call list");
      pc.a[k].p = nullptr;
      switch (tag) {
        case T_I32: pc.a[k].i = (int)(int64_t)val; break;
        case T_F32: { const uint32_t b = (uint32_t)val; memcpy(&pc.a[k].f, &b, 4); break; }
        case T_PTR: if (!at(aux, val, &pc.a[k].p)) return fail("pointer outside its region"); break;
        case T_NULL: break;
        case T_STREAM: pc.a[k].p = (void*)p->stream; break;
        case T_DESC: {
          if (aux == 0 || aux > VSD_CONV_GROUP_MAX || val != (uint64_t)aux * sizeof(vsd_conv_desc)) return fail("descriptor array of another interface version");
          p->blobs.emplace_back((size_t)val);
          std::vector<unsigned char>& blob = p->blobs.back();
          r.bytes(blob.data(), blob.size());
          const uint32_t nfix = r.get<uint32_t>();
          if (!r.ok || nfix > 64 * aux) return fail("bad descriptor record");
          for (uint32_t j = 0; j < nfix; ++j) {
            const uint32_t boff = r.get<uint32_t>(), reg = r.get<uint32_t>();
            const uint64_t off = r.get<uint64_t>();
            void* q = nullptr;
            if (!r.ok || boff + sizeof(void*) > blob.size() || !at(reg, off, &q)) return fail("bad descriptor pointer");
            memcpy(blob.data() + boff, &q, sizeof(void*));
          }
          pc.a[k].p = blob.data();
          break;
        }
        default: return fail("unknown argument tag");
      }
    }
  }
  for (uint32_t i = 0; i < nreg; ++i) {
    if (!saved[i]) continue;
    std::vector<unsigned char> host(1 << 24);
    uint64_t done = 0;
    while (done < size[i]) {
      const size_t n = (size_t)std::min<uint64_t>(host.size(), size[i] - done);
      if (!r.bytes(host.data(), n)) return fail("truncated region contents");
      if (hipMemcpy((char*)p->regions[i] + done, host.data(), n, hipMemcpyHostToDevice) != hipSuccess) return fail("upload failed");
      done += n;
    }
  }
  fclose(f);
  f = nullptr;
  if (!at(in_r, in_off, &p->in) || !at(out_r, out_off, &p->out) || !at(pr_r, pr_off, &p->prompt) || pr_off + pr_bytes > size[pr_r]) {
    plan_release(p);
    return vsd_fail(ctx, VSD_ERR_ARG, "plan_load: frame buffers / prompt block outside their regions");
  }
  p->prompt_bytes = (size_t)pr_bytes;
  p->io_bytes = (size_t)p->batch * p->H * p->W * 3;
  // one eager pass (first-touch of every kernel, the error reports of the ops), then the captured one
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      int rc = vsd_graph_begin(ctx, (void*)p->stream);
      if (rc != VSD_OK) { plan_release(p); return rc; }
    }
    int rc = VSD_OK;
    for (const PlanCall& pc : p->calls) {
      rc = plan_dispatch(ctx, pc.fn, pc.n, pc.a);
      if (rc != VSD_OK) break;
    }
    if (pass == 1) {
      const int rc2 = vsd_graph_end(ctx, (void*)p->stream, &p->graph);
      if (rc == VSD_OK) rc = rc2;
    }
    if (rc != VSD_OK) {
      std::string why = ctx->err;
      plan_release(p);
      return vsd_fail(ctx, rc < 0 && rc > -1000 && why.empty() ? VSD_ERR_ARG : rc, "plan_load: replaying the program failed (%d): %s", rc, why.c_str());
    }
    if (hipStreamSynchronize(p->stream) != hipSuccess) { plan_release(p); return vsd_fail(ctx, VSD_ERR_HIP, "plan_load: the eager pass faulted"); }
  }
  *plan_out = p;
  return VSD_OK;
}

extern "C" int vsd_plan_info(vsd_ctx* ctx, vsd_plan* plan, int* dims) {
  if (!ctx || !plan || !dims) return VSD_ERR_ARG;
  dims[0] = plan->H; dims[1] = plan->W; dims[2] = plan->batch;
  return VSD_OK;
}

extern "C" int vsd_plan_submit(vsd_ctx* ctx, vsd_plan* plan, const void* frame_u8_host, void* out_u8_host) {
  if (!ctx || !plan || !frame_u8_host || !out_u8_host) return VSD_ERR_ARG;
  VSD_HIP(ctx, hipMemcpyAsync(plan->in, frame_u8_host, plan->io_bytes, hipMemcpyHostToDevice, plan->stream));
  VSD_HIP(ctx, hipGraphLaunch((hipGraphExec_t)plan->graph, plan->stream));
  VSD_HIP(ctx, hipMemcpyAsync(out_u8_host, plan->out, plan->io_bytes, hipMemcpyDeviceToHost, plan->stream));
  return VSD_OK;
}

extern "C" int vsd_plan_wait(vsd_ctx* ctx, vsd_plan* plan) {
  if (!ctx || !plan) return VSD_ERR_ARG;
  VSD_HIP(ctx, hipStreamSynchronize(plan->stream));
  return VSD_OK;
}

extern "C" int vsd_plan_infer(vsd_ctx* ctx, vsd_plan* plan, const void* frame_u8_host, void* out_u8_host) {
  const int rc = vsd_plan_submit(ctx, plan, frame_u8_host, out_u8_host);
  return rc != VSD_OK ? rc : vsd_plan_wait(ctx, plan);
}

// Another prompt for a loaded plan: the bytes of an engine.PromptBlock of the same layout (videosd_amd/plan.py export_prompt) replace
// the plan's -- the C counterpart of VideoSDPipeline's prompt cache (the reference re-encodes the prompt every frame,
// lcm_controlnet.py:115-198; here a prompt is ~40 MB of constants built once).  Waits for the plan's stream first.
extern "C" int vsd_plan_load_prompt(vsd_ctx* ctx, vsd_plan* plan, const char* path) {
  if (!ctx || !plan || !path) return VSD_ERR_ARG;
  FILE* f = fopen(path, "rb");
  if (!f) return vsd_fail(ctx, VSD_ERR_ARG, "plan_load_prompt: cannot open %s", path);
  char magic[8];
  uint64_t n = 0;
  const bool head = fread(magic, 1, 8, f) == 8 && fread(&n, 8, 1, f) == 1 && memcmp(magic, "VSDPRMT1", 8) == 0;
  if (!head || n != plan->prompt_bytes) {
    fclose(f);
    return vsd_fail(ctx, VSD_ERR_ARG, "plan_load_prompt: %s is not a prompt file of this plan's layout (%llu bytes, the plan's block has %zu)", path,
                    (unsigned long long)n, plan->prompt_bytes);
  }
  std::vector<unsigned char> host((size_t)n);
  const bool ok = fread(host.data(), 1, host.size(), f) == host.size();
  fclose(f);
  if (!ok) return vsd_fail(ctx, VSD_ERR_ARG, "plan_load_prompt: %s is truncated", path);
  VSD_HIP(ctx, hipStreamSynchronize(plan->stream));
  VSD_HIP(ctx, hipMemcpy(plan->prompt, host.data(), host.size(), hipMemcpyHostToDevice));
  return VSD_OK;
}

// page-locked host memory for a plan's frames (copies from pageable memory are staged by the runtime and do not overlap other lanes)
extern "C" void* vsd_pinned_alloc(vsd_ctx* ctx, size_t bytes) {
  void* p = nullptr;
  if (!ctx || hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
extern "C" void vsd_pinned_free(vsd_ctx* ctx, void* p) {
  (void)ctx;
  if (p) (void)hipHostFree(p);
}

extern "C" void vsd_plan_free(vsd_ctx* ctx, vsd_plan* plan) {
  (void)ctx;
  plan_release(plan);
}

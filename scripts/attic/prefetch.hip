// Weights run ahead of the dependency: a lone frame's persistent weight-prefetch kernel (gfx950).
//
// A lone frame streams its 2.45 GB of weights once per denoising step through ~270 dependent conv launches per network; every
// launch starts on weights no cache holds (the 256 MB memory-side cache is 10x smaller than a step's weights), and its K loop
// then runs at the latency of HBM misses instead of memory-side-cache hits (docs/NOTEBOOK.md, "Cold weights": 10-25 % of a
// weight-heavy layer, +8 % of a lone frame).  Rounds 3-4 tried to touch the next layer's weights from inside the conv /
// GroupNorm kernels (coupled to their wait counters: no gain) and from per-layer touch kernels behind graph edges (the edges
// cost more than the cold weights).  This form has neither coupling: ONE kernel per frame on a launch stream of its own walks
// the frame's weight table in program order and touches one dword of every 128-byte line, kept a bounded number of bytes ahead
// of the consumer by a progress word that every conv launch publishes when it starts (ConvParams::progress).  It reads only,
// writes nothing but its own exit record, and every wait is bounded (it gives up when the consumer stops moving, when the host
// raises the stop word -- another launch wants the stream -- and at a hard time limit), so a frame never depends on it.
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"

namespace {

struct PfEntry {
  const void* ptr;
  uint32_t bytes;
  uint32_t cum_kb;  // KiB of all entries before this one
};

struct PfParams {
  const PfEntry* table;
  int n;
  const int* progress;       // index of the latest conv launch that has STARTED (device memory, written by the conv kernels)
  const int* stop;           // device word: non-zero = leave now
  uint32_t lookahead_kb;     // stay at most this far ahead of the consumer
  uint32_t min_lead_kb;      // ... and do not touch what the consumer reaches within this many KiB (it would fetch it first anyway)
  uint32_t min_entry_kb;     // entries smaller than this are not worth a touch (their launch is latency, not weights)
  uint32_t stall_ticks;      // give up after this many 10 ns ticks without consumer progress
  uint32_t limit_ticks;      // ... and after this many ticks in all
  int nt;                    // touch with non-temporal loads (experiments)
  int* exit_record;          // [4]: entries walked, reason (0 done, 1 stall, 2 stop, 3 limit), ticks, entries actually touched
};

typedef const __attribute__((address_space(1))) int* gptr_i32;
constexpr int PF_INFLIGHT = 16;  // loads in flight per lane

template <bool NT>
__device__ __forceinline__ unsigned touch_lines(const char* base, long long lines, int gtid, int nthr) {
  unsigned sink = 0;
  for (long long i = gtid; i < lines; i += (long long)PF_INFLIGHT * nthr) {
    unsigned v[PF_INFLIGHT];
#pragma unroll
    for (int j = 0; j < PF_INFLIGHT; ++j) {
      const long long l = i + (long long)j * nthr;
      const unsigned* a = (const unsigned*)(base + ((l < lines ? l : i) << 7));  // (clamped: a surplus load hits this lane's first line)
      v[j] = NT ? __builtin_nontemporal_load(a) : __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
#pragma unroll
    for (int j = 0; j < PF_INFLIGHT; ++j) sink ^= v[j];
  }
  return sink;
}

__global__ __launch_bounds__(256) void prefetch_kernel(const PfParams p) {
  const int tid = threadIdx.x, nthr = blockDim.x * gridDim.x, gtid = blockIdx.x * blockDim.x + tid;
  const unsigned long long t_begin = wall_clock64();
  unsigned long long t_progress = t_begin;
  int last_seen = -1, reason = 0, e = 0, touched = 0;
  unsigned sink = 0;
  uint32_t at_kb = 0;  // (tid 0) the consumer's place at the last poll, KiB into the table
  __shared__ int sh_go;
  for (; e < p.n; ++e) {
    const PfEntry ent = p.table[e];
    // ---- throttle: one lane polls (relaxed agent-scope loads, s_sleep between polls), the workgroup follows its verdict:
    //      1 = touch this entry, 2 = skip it (the consumer is at or almost at it, or it is too small to matter), 0 = leave
    if (tid == 0) {
      int go = 1;
      if ((ent.bytes >> 10) < p.min_entry_kb) {
        go = 2;
      } else if (ent.cum_kb > at_kb + p.lookahead_kb || ent.cum_kb < at_kb + p.min_lead_kb) {  // (else: inside the window seen last time)
        for (;;) {
          const int seen = __hip_atomic_load((gptr_i32)p.progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int at = seen < 0 ? 0 : (seen >= p.n ? p.n - 1 : seen);
          const unsigned long long now = wall_clock64();
          if (seen != last_seen) {
            last_seen = seen;
            t_progress = now;
          }
          at_kb = p.table[at].cum_kb;
          if (__hip_atomic_load((gptr_i32)p.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { go = 0; reason = 2; break; }
          if (now - t_begin > p.limit_ticks) { go = 0; reason = 3; break; }
          if (e <= at || ent.cum_kb < at_kb + p.min_lead_kb) { go = 2; break; }
          if (ent.cum_kb <= at_kb + p.lookahead_kb) break;  // close enough to the consumer: touch this entry
          if (now - t_progress > p.stall_ticks) { go = 0; reason = 1; break; }
          __builtin_amdgcn_s_sleep(32);
        }
      }
      sh_go = go;
    }
    __syncthreads();
    const int go = sh_go;
    __syncthreads();
    if (!go) break;
    if (go == 2) continue;
    // ---- touch: one dword of every 128-byte line, PF_INFLIGHT loads in flight per lane
    const long long lines = ((long long)ent.bytes + 127) >> 7;
    sink ^= p.nt ? touch_lines<true>((const char*)ent.ptr, lines, gtid, nthr) : touch_lines<false>((const char*)ent.ptr, lines, gtid, nthr);
    ++touched;
  }
  if (sink == 0x9e3779b9u && p.exit_record) p.exit_record[3] = -1;  // (keeps the loads alive; never true in practice, harmless if it is)
  if (blockIdx.x == 0 && tid == 0 && p.exit_record) {
    p.exit_record[0] = e;
    p.exit_record[1] = reason;
    p.exit_record[2] = (int)(wall_clock64() - t_begin);
    p.exit_record[3] = touched;
  }
}

}  // namespace

extern "C" int vsd_prefetch_weights(vsd_ctx* ctx, const void* table, int n, const void* progress, const void* stop, int lookahead_kb,
                                    int workgroups, float stall_ms, float limit_ms, void* exit_record, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!table || n <= 0 || !progress || !stop || lookahead_kb <= 0 || workgroups <= 0 || workgroups > 1024 || stall_ms <= 0.f || limit_ms <= 0.f ||
      limit_ms > 2000.f)
    return vsd_fail(ctx, VSD_ERR_ARG, "prefetch_weights: bad arguments (n=%d lookahead_kb=%d workgroups=%d stall_ms=%g limit_ms=%g)", n,
                    lookahead_kb, workgroups, stall_ms, limit_ms);
  PfParams p;
  p.table = (const PfEntry*)table;
  p.n = n;
  p.progress = (const int*)progress;
  p.stop = (const int*)stop;
  p.lookahead_kb = (uint32_t)lookahead_kb;
  // (experiments: VSD_PF_MIN_LEAD_KB, VSD_PF_MIN_ENTRY_KB, VSD_PF_NT)
  const int min_lead = getenv("VSD_PF_MIN_LEAD_KB") ? atoi(getenv("VSD_PF_MIN_LEAD_KB")) : 2048;
  const int min_entry = getenv("VSD_PF_MIN_ENTRY_KB") ? atoi(getenv("VSD_PF_MIN_ENTRY_KB")) : 512;
  const int nt = getenv("VSD_PF_NT") ? atoi(getenv("VSD_PF_NT")) : 0;
  p.min_lead_kb = (uint32_t)(min_lead < lookahead_kb ? min_lead : lookahead_kb / 2);
  p.min_entry_kb = (uint32_t)min_entry;
  p.nt = nt;
  p.stall_ticks = (uint32_t)(stall_ms * 1e5f);  // wall_clock64: 100 MHz
  p.limit_ticks = (uint32_t)(limit_ms * 1e5f);
  p.exit_record = (int*)exit_record;
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls(ctx, s, VSD_FAM_ELEMENTWISE, 0.0);
  hipLaunchKernelGGL(prefetch_kernel, dim3(workgroups), dim3(256), 0, s, p);
  return ls.finish();
}

extern "C" int vsd_fill32(vsd_ctx* ctx, void* dst, int value, int count, void* stream) {
  if (!ctx) return VSD_ERR_ARG;
  if (!dst || count <= 0) return vsd_fail(ctx, VSD_ERR_ARG, "fill32: bad arguments");
  VSD_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)dst, value, (size_t)count, (hipStream_t)stream));
  return VSD_OK;
}

// (without -DVSD_ATTN_PROBE: launch time only -- the form the -DVSD_ATTN_EXP=n ablation builds use, see attention.hip)
// Standalone timing probe of csrc/attention.hip (not part of libvsd): builds the kernel with -DVSD_ATTN_PROBE, runs one
// self-attention problem and prints (a) the launch time and (b) where wave 0 of workgroup 0 spent its shader clocks.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -DVSD_ATTN_PROBE scripts/attn_probe.cpp \
//         videosd_amd/csrc/attention.hip videosd_amd/csrc/api.hip -o gpurun_out/attn_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../include/vsd.h"
#ifdef VSD_ATTN_PROBE
extern "C" void vsd_attn_set_probe(void* buf);
#endif
int main(int argc, char** argv) {
  int sq = argc > 1 ? atoi(argv[1]) : 4096, heads = argc > 2 ? atoi(argv[2]) : 8, d = argc > 3 ? atoi(argv[3]) : 40;
  int batch = argc > 4 ? atoi(argv[4]) : 1;
  int c = heads * d, timg = (sq + 63) / 64 * 64;
  vsd_ctx* ctx = vsd_create(0);
  if (!ctx) { printf("no device\n"); return 1; }
  size_t nq = (size_t)batch * sq * c, nv = (size_t)c * batch * timg;
  std::vector<_Float16> h(nq > nv ? nq : nv);
  srand(1);
  for (auto& x : h) x = (_Float16)((rand() % 2001 - 1000) / 500.0f);
  _Float16 *q, *k, *vt, *o; long long* probe;
  hipMalloc(&q, nq * 2); hipMalloc(&k, nq * 2); hipMalloc(&vt, nv * 2); hipMalloc(&o, nq * 2); hipMalloc(&probe, 64);
  hipMemcpy(q, h.data(), nq * 2, hipMemcpyHostToDevice); hipMemcpy(k, h.data(), nq * 2, hipMemcpyHostToDevice);
  hipMemcpy(vt, h.data(), nv * 2, hipMemcpyHostToDevice); hipMemset(probe, 0, 64);
#ifdef VSD_ATTN_PROBE
  vsd_attn_set_probe(probe);
#endif
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&]() { return vsd_attention_batched(ctx, q, c, k, c, vt, batch * timg, o, c, sq, sq, heads, d, 0.158f, 0, batch, sq, timg, s); };
  for (int i = 0; i < 3; ++i) if (run()) { printf("err %s\n", vsd_last_error(ctx)); return 1; }
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int i = 0; i < 20; ++i) run();
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long pr[8]; hipMemcpy(pr, probe, 64, hipMemcpyDeviceToHost);
  long long tot = 0; for (int i = 0; i < 6; ++i) tot += pr[i];
  printf("sq=%d heads=%d d=%d batch=%d: %.1f us/launch; wave0 clocks: load-issue %lld | QK %lld | softmax %lld | PV %lld | lds-store %lld | barrier %lld | total %lld (%d tiles)\n",
         sq, heads, d, batch, ms / 20 * 1e3, pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], tot, (sq + 63) / 64);
  return 0;
}

// Standalone timing probe of csrc/conv_gemm.hip (not part of libvsd): builds the kernel with -DVSD_CONV_PROBE, runs one
// conv / linear problem and prints the launch time and where wave 0 of workgroup 0 spent its shader clocks.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DVSD_CONV_PROBE scripts/conv_probe.cpp videosd_amd/csrc/conv_gemm.hip videosd_amd/csrc/conv_t*.hip videosd_amd/csrc/conv_halo.hip \
//         videosd_amd/csrc/api.hip -o scripts/conv_probe.bin
// usage: conv_probe.bin H W Cin Cout ksize tile pipeline split batch [residual]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/vsd.h"
extern "C" void vsd_conv_set_probe(void* buf);
int main(int argc, char** argv) {
  if (argc < 10) { printf("usage: H W Cin Cout ksize tile pipeline split batch [residual]\n"); return 1; }
  int H = atoi(argv[1]), W = atoi(argv[2]), cin = atoi(argv[3]), cout = atoi(argv[4]), ks = atoi(argv[5]);
  int tile = atoi(argv[6]), pl = atoi(argv[7]), split = atoi(argv[8]), batch = atoi(argv[9]);
  int res = argc > 10 ? atoi(argv[10]) : 0;
  vsd_ctx* ctx = vsd_create(0);
  if (!ctx) { printf("no device\n"); return 1; }
  size_t M = (size_t)batch * H * W, K = (size_t)ks * ks * cin, Kp = (K + 63) / 64 * 64;
  std::vector<_Float16> h(std::max(M * cin, (size_t)cout * Kp));
  srand(1);
  for (auto& x : h) x = (_Float16)((rand() % 2001 - 1000) / 4000.0f);
  _Float16 *x, *w, *o, *r; long long* probe; float* ws; int* cnt;
  hipMalloc(&x, M * cin * 2); hipMalloc(&w, cout * Kp * 2); hipMalloc(&o, M * cout * 2); hipMalloc(&r, M * cout * 2);
  hipMalloc(&probe, 128); hipMalloc(&ws, (size_t)split * M * cout * 4 + 256); hipMalloc(&cnt, VSD_SPLITK_MAX_TILES * 4);
  hipMemcpy(x, h.data(), M * cin * 2, hipMemcpyHostToDevice); hipMemcpy(w, h.data(), cout * Kp * 2, hipMemcpyHostToDevice);
  hipMemset(r, 0, M * cout * 2); hipMemset(probe, 0, 128); hipMemset(cnt, 0, VSD_SPLITK_MAX_TILES * 4);
  vsd_conv_set_probe(probe);
  vsd_conv_desc d; memset(&d, 0, sizeof d);
  d.src0 = x; d.c0 = cin; d.hs = d.hi = d.ho = H; d.ws = d.wi = d.wo = W; d.ksize = ks; d.stride = 1; d.pad = ks / 2;
  d.weight = w; d.n = cout; d.k = (int)K; d.kp = (int)Kp; d.residual = res ? r : nullptr; d.ldr = cout; d.out_scale = 1.f;
  d.out = o; d.ldo = cout; d.tile = tile; d.split_k = split; d.workspace = ws; d.counters = cnt; d.pipeline = pl; d.batch = batch;
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) if (vsd_conv_gemm(ctx, &d, s)) { printf("err %s\n", vsd_last_error(ctx)); return 1; }
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int i = 0; i < 20; ++i) vsd_conv_gemm(ctx, &d, s);
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long pr[16]; hipMemcpy(pr, probe, 128, hipMemcpyDeviceToHost);
  long long tot = 0; for (int i = 0; i < 7; ++i) tot += pr[i];
  double us = ms / 20 * 1e3, fl = 2.0 * M * cout * K;
  printf("M=%zu N=%d K=%zu ks=%d tile=%d pl=%d split=%d: %.1f us  %.0f TF/s | wave0 clocks: prologue %lld | vmcnt-wait %lld | barrier %lld | issue %lld | lds+mfma %lld | acc->lds %lld | epilogue %lld | total %lld (%zu k-tiles)\n",
         M, cout, K, ks, tile, pl, split, us, fl / us / 1e6, pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], tot, Kp / 64 / split);
  long long tot2 = 0; for (int i = 8; i < 15; ++i) tot2 += pr[i];
  printf("   last workgroup (a later round when the grid exceeds the resident slots): prologue %lld | vmcnt-wait %lld | barrier %lld | issue %lld | lds+mfma %lld | acc->lds %lld | epilogue %lld | total %lld\n",
         pr[8], pr[9], pr[10], pr[11], pr[12], pr[13], pr[14], tot2);
  return 0;
}

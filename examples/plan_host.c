/* A host WITHOUT Python driving frames through the denoising path (include/vsd.h vsd_plan_*): what SURVEY.md section 8b's
 * whole-frame entry points are for.  The plan file comes from videosd_amd.plan.export_plan (one frame size / step count / prompt).
 *
 *   gcc -O2 examples/plan_host.c -Iinclude -Lvideosd_amd -lvsd -Wl,-rpath,$PWD/videosd_amd -o /tmp/plan_host
 *   /tmp/plan_host frame.vsdplan in.raw out.raw [repeats]
 * in.raw / out.raw: uint8 [frames per launch][H][W][3].  Prints the frame rate of `repeats` launches one after the other.
 * (the reference's caller is a Python loop, server.py:104-143; this is the same loop for a C / C++ / Go-via-cgo media server) */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "vsd.h"

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s plan in.raw out.raw [repeats]\n", argv[0]);
    return 2;
  }
  const int repeats = argc > 4 ? atoi(argv[4]) : 1;
  vsd_ctx* ctx = vsd_create(0);
  if (!ctx) { fprintf(stderr, "no HIP device\n"); return 1; }
  vsd_plan* plan = NULL;
  if (vsd_plan_load(ctx, argv[1], &plan) != VSD_OK) { fprintf(stderr, "%s\n", vsd_last_error(ctx)); return 1; }
  int dims[3];
  vsd_plan_info(ctx, plan, dims);
  const size_t n = (size_t)dims[2] * dims[0] * dims[1] * 3;
  unsigned char* in = malloc(n);
  unsigned char* out = malloc(n);
  FILE* f = fopen(argv[2], "rb");
  if (!f || fread(in, 1, n, f) != n) { fprintf(stderr, "%s: need %zu bytes (%d x %d x %d x 3)\n", argv[2], n, dims[2], dims[0], dims[1]); return 1; }
  fclose(f);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < repeats; ++i)
    if (vsd_plan_infer(ctx, plan, in, out) != VSD_OK) { fprintf(stderr, "%s\n", vsd_last_error(ctx)); return 1; }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  const double s = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
  f = fopen(argv[3], "wb");
  if (!f || fwrite(out, 1, n, f) != n) { fprintf(stderr, "cannot write %s\n", argv[3]); return 1; }
  fclose(f);
  printf("%d x %d, %d frame(s) per launch: %d launches in %.3f s = %.1f frames/s, %.2f ms per launch\n", dims[1], dims[0], dims[2], repeats, s,
         repeats * dims[2] / s, 1e3 * s / repeats);
  vsd_plan_free(ctx, plan);
  vsd_destroy(ctx);
  return 0;
}

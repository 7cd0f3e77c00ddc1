/* A host WITHOUT Python driving frames through the denoising path (include/vsd.h vsd_plan_*): what SURVEY.md section 8b's
 * whole-frame entry points are for.  The plan file comes from videosd_amd.plan.export_plan / VideoSDPipeline.export_plan (one frame
 * size / step count / prompt / frames per launch).
 *
 *   gcc -O2 examples/plan_host.c -Iinclude -Lvideosd_amd -lvsd -Wl,-rpath,$PWD/videosd_amd -o /tmp/plan_host
 *   /tmp/plan_host frame.vsdplan in.raw out.raw [launches] [lanes]
 * in.raw / out.raw: uint8 [frames per launch][H][W][3].  lanes (1..4): that many copies of the plan in flight, one per launch
 * stream (the reference keeps N actors per node, server.py:132-137; inside one process the launch lanes do the same) -- frame k goes
 * to lane k mod lanes.  Prints the frame rate; out.raw is lane 0's last result.
 * (the reference's caller is a Python loop, server.py:104-143; this is that loop for a C / C++ / Go-via-cgo media server) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "vsd.h"

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s plan in.raw out.raw [launches] [lanes]\n", argv[0]);
    return 2;
  }
  const int launches = argc > 4 ? atoi(argv[4]) : 1;
  int lanes = argc > 5 ? atoi(argv[5]) : 1;
  if (lanes < 1 || lanes > VSD_POOL_STREAMS) lanes = 1;
  vsd_ctx* ctx = vsd_create(0);
  if (!ctx) { fprintf(stderr, "no HIP device\n"); return 1; }
  vsd_plan* plan[VSD_POOL_STREAMS];
  unsigned char *in[VSD_POOL_STREAMS], *out[VSD_POOL_STREAMS];
  int dims[3];
  size_t n = 0;
  for (int l = 0; l < lanes; ++l) {
    /* one lane: a stream of the plan's own; several: the process's launch streams (own hardware queues, vsd_stream_pool) */
    const int rc = lanes == 1 ? vsd_plan_load(ctx, argv[1], &plan[l]) : vsd_plan_load_lane(ctx, argv[1], l, &plan[l]);
    if (rc != VSD_OK) { fprintf(stderr, "%s\n", vsd_last_error(ctx)); return 1; }
    vsd_plan_info(ctx, plan[l], dims);
    n = (size_t)dims[2] * dims[0] * dims[1] * 3;
    in[l] = vsd_pinned_alloc(ctx, n);
    out[l] = vsd_pinned_alloc(ctx, n);
    if (!in[l] || !out[l]) { fprintf(stderr, "no pinned host memory\n"); return 1; }
  }
  FILE* f = fopen(argv[2], "rb");
  if (!f || fread(in[0], 1, n, f) != n) { fprintf(stderr, "%s: need %zu bytes (%d x %d x %d x 3)\n", argv[2], n, dims[2], dims[0], dims[1]); return 1; }
  fclose(f);
  for (int l = 1; l < lanes; ++l) memcpy(in[l], in[0], n);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < launches; ++i) {
    const int l = i % lanes;
    if (i >= lanes && vsd_plan_wait(ctx, plan[l]) != VSD_OK) { fprintf(stderr, "%s\n", vsd_last_error(ctx)); return 1; }  /* the lane's previous frame */
    if (vsd_plan_submit(ctx, plan[l], in[l], out[l]) != VSD_OK) { fprintf(stderr, "%s\n", vsd_last_error(ctx)); return 1; }
  }
  for (int l = 0; l < lanes; ++l)
    if (vsd_plan_wait(ctx, plan[l]) != VSD_OK) { fprintf(stderr, "%s\n", vsd_last_error(ctx)); return 1; }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  const double s = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
  f = fopen(argv[3], "wb");
  if (!f || fwrite(out[0], 1, n, f) != n) { fprintf(stderr, "cannot write %s\n", argv[3]); return 1; }
  fclose(f);
  printf("%d x %d, %d frame(s) per launch, %d lane(s): %d launches in %.3f s = %.1f frames/s, %.2f ms per launch per lane\n", dims[1], dims[0], dims[2], lanes,
         launches, s, launches * dims[2] / s, 1e3 * s / launches * lanes);
  for (int l = 0; l < lanes; ++l) {
    vsd_plan_free(ctx, plan[l]);
    vsd_pinned_free(ctx, in[l]);
    vsd_pinned_free(ctx, out[l]);
  }
  vsd_destroy(ctx);
  return 0;
}

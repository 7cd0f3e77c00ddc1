// Host side of vsd_conv_gemm: argument checks, tile / pipeline dispatch (kernels: conv_kernels.h, conv_t*.hip, conv_halo.hip).
#include "conv_kernels.h"


#ifdef VSD_CONV_PROBE
extern "C" void vsd_conv_set_probe(void* buf) { g_conv_probe = (long long*)buf; }
#endif
#ifdef VSD_WG_TIMELINE
// development builds only (common.h): `buf` = device memory of `words` 8-byte words, zero-filled; every instrumented launch
// appends [grid, kind, 4 words per workgroup]; returns the words used so far
extern "C" void vsd_wgtl_set(void* buf, int64_t words) { g_wgtl = (unsigned long long*)buf; g_wgtl_cap = (size_t)words; g_wgtl_off = 0; }
extern "C" int64_t vsd_wgtl_used(void) { return (int64_t)g_wgtl_off; }
// buf: device memory of 3 * VSD_CUT_FAMS 8-byte words, zeroed by the caller (per kernel family, common.h VSD_CUT_*: workgroup ticks
// of 10 ns, workgroups, ticks x waves); null switches the accounting off
extern "C" void vsd_cut_set(void* buf) { g_cut = (unsigned long long*)buf; }
#endif

// one launch, ready to go: the kernel's argument block, its form (tile, pipeline, halo) and its grid
struct ConvLaunch {
  ConvParams p;
  int BM, BN, grid, stages;
  bool halo, c64;
};

// argument checks + everything the host derives from a descriptor (shared by vsd_conv_gemm and vsd_conv_gemm_group)
static int conv_setup(vsd_ctx* ctx, const vsd_conv_desc* d, ConvLaunch& cl, bool grouped) {
  ConvParams& p = cl.p;
  p.src0 = (const half_t*)d->src0;
  p.src1 = (const half_t*)d->src1;
  p.c0 = d->c0;
  p.c1 = d->src1 ? d->c1 : 0;
  p.cin = p.c0 + p.c1;
  p.hs = d->hs; p.ws = d->ws; p.hi = d->hi; p.wi = d->wi; p.ho = d->ho; p.wo = d->wo;
  p.ksize = d->ksize; p.stride = d->stride; p.pad = d->pad;
  p.resize = (d->hi != d->hs) || (d->wi != d->ws);
  // nearest resize = a 32-bit fixed-point multiply in the loaders: source row = (iy * ceil(hs * 2^22 / hi)) >> 22, exactly
  // floor(iy * hs / hi) while iy * hi < 2^22 (hi <= 2048) and without overflow while hs * 2^22 + hi < 2^32 (hs <= 1023): frames
  // up to 2046 pixels a side through the TAESD decoder's 2x upsamples (1920 x 1080 included)
  if (d->hi <= 0 || d->wi <= 0 || d->hs > d->hi || d->ws > d->wi ||
      (p.resize && (d->hi > 2048 || d->wi > 2048 || d->hs > 1023 || d->ws > 1023)))
    return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: input size %dx%d (stored %dx%d) unsupported (a resized input may be at most 2048 a side, "
                    "its stored form 1023)", d->hi, d->wi, d->hs, d->ws);
  if (p.resize) {
    p.rshift = 22;
    p.rmul_y = (unsigned)((((unsigned long long)d->hs << 22) + d->hi - 1) / d->hi);
    p.rmul_x = (unsigned)((((unsigned long long)d->ws << 22) + d->wi - 1) / d->wi);
  } else {
    p.rshift = 0;
    p.rmul_y = p.rmul_x = 1;
  }
  p.generic = (p.cin % 64) != 0;
  {
    const size_t a_pix = (size_t)(d->batch < 1 ? 1 : d->batch) * d->hs * d->ws + (size_t)d->pad * d->ws + d->pad;
    const size_t cmax = (size_t)(d->c0 > d->c1 ? d->c0 : d->c1);
    p.halo_ok = !p.generic && a_pix < (1u << 24) && a_pix * cmax * 2 < 0x7fffffffull && (size_t)d->n * d->kp * 2 < 0x7fffffffull;
    p.fast = !p.generic && !p.resize && a_pix < (1u << 24) && a_pix * cmax * 2 < 0x7fffffffull && (size_t)d->n * d->kp * 2 < 0x7fffffffull &&
             d->ksize * d->ksize <= 32 && !getenv("VSD_CONV_NO_FAST");
  }
  p.w = (const half_t*)d->weight;
  p.pointwise = d->ksize == 1 && d->stride == 1 && d->pad == 0 && !p.resize && d->ho == d->hs && d->wo == d->ws;
  p.batch = d->batch < 1 ? 1 : d->batch;
  p.hw_out = d->ho * d->wo;
  p.img_in = d->hs * d->ws;
  p.t_img = d->t_img > 0 ? d->t_img : p.hw_out;
  p.M = p.batch * p.hw_out;
  p.N = d->n; p.K = d->k; p.Kp = d->kp;
  p.bias = (const half_t*)d->bias;
  p.rowvec = (const half_t*)d->rowvec;
  p.residual = (const half_t*)d->residual;
  p.residual2 = (const half_t*)d->residual2;
  p.ldr = d->ldr;
  p.out_scale = d->out_scale;
  p.out_scale_dev = (const float*)d->out_scale_dev;
  p.softmax_cols = d->softmax_cols;
  p.act = d->act;
#ifdef VSD_PROBE
  { static const bool skip = getenv("VSD_SKIP_EPI") != nullptr; if (skip) p.act |= 0x4000; }  // (what-if probe builds only: scripts/whatif_probe.sh)
#endif
  p.out = (half_t*)d->out; p.ldo = d->ldo;
  p.out2 = (half_t*)d->out2; p.add2 = (const half_t*)d->add2;
  p.out_t = (half_t*)d->out_t; p.ldt = d->ldt; p.t_col0 = d->t_col0;
  p.split_k = d->split_k < 1 ? 1 : d->split_k;
  p.ws_partial = (float*)d->workspace;
  p.counters = (int*)d->counters;
  p.zeros = (const half_t*)ctx->zero_page;
#ifdef VSD_CONV_PROBE
  p.probe = g_conv_probe;
#endif
#ifdef VSD_WG_TIMELINE
  p.wgtl = nullptr;
#endif
  p.rowstat_out = (float*)d->rowstat_out;
  p.chanstat_part = (float*)d->chanstat_part;
  p.chanstat_out = (float*)d->chanstat_out;
  p.chan_counters = (int*)d->chan_counters;
  p.ln_part = (const float*)d->ln_part;
  p.ln_groups = d->ln_groups;
  p.ln_eps = d->ln_eps;
  p.ln_s = (const float*)d->ln_s;
  p.ln_t = (const float*)d->ln_t;
  VSD_CUT_SET(p)
  const int stages = d->pipeline;
  if (stages != 0 && (stages < 3 || stages > 10)) return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: pipeline %d (0, 3..10)", stages);
  if ((stages == 8 || stages == 9) && !p.fast)
    return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: the eight-wave forms (pipelines 8, 9) exist for the buffer-load path only (Cin %% 64 == 0 per "
                    "source, no resize)");
  const bool c64 = stages == 10;  // the persistent 64 -> 64 channel form (conv_c64.hip): patches as the halo form's 16 x 16
  const bool halo = stages == 7 || c64;

  if (!p.src0 || !p.w || !p.out) return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: null src/weight/out");
  if (p.M <= 0 || p.N <= 0) return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: empty problem M=%d N=%d", p.M, p.N);
  if (p.c0 % 8 || p.c1 % 8) return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: channels must be multiples of 8 (%d,%d)", p.c0, p.c1);
  if (p.c1 && (p.c0 % 64 || p.c1 % 64)) return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: concat sources need C %% 64 == 0");
  if (p.K != p.ksize * p.ksize * p.cin) return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: K=%d != ks^2*Cin=%d", p.K, p.ksize * p.ksize * p.cin);
  if (p.Kp % BK || p.Kp < p.K) return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: Kp=%d must be K rounded up to 64", p.Kp);
  if (p.ksize != 1 && p.ksize != 3) return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: ksize %d", p.ksize);
  if (p.ldo % 8 || (p.residual && p.ldr % 8)) return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: ld must be a multiple of 8");
  if (p.batch > 1 && p.chanstat_out) return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: chanstat_out is per tensor, not per image (batch must be 1)");
  if (p.out_t && p.batch > 1 && (p.t_img < p.hw_out || (int64_t)p.batch * p.t_img > p.ldt))
    return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: transposed output needs hw <= t_img and batch * t_img <= ldt");
  if (p.out2 && !p.add2) return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: out2 without add2");
  if ((p.act & 0xff) > VSD_ACT_GELU) return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: activation %d", p.act & 0xff);
  if ((p.act & 0xff) == VSD_ACT_GELU && (p.out_t || p.rowstat_out || p.ln_part || (p.act & VSD_ACT_POST)))
    return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: erf GELU exists in the general epilogue walk only (no transposed output, row statistics, "
                    "fused LayerNorm or post-residual form)");
  if (p.rowstat_out && (p.N % 64 || p.out_t || (p.split_k > 1 && !d->counters)))
    return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: rowstat_out needs N %% 64 == 0, no transposed output and the in-kernel split-K form");
  if (p.chanstat_out && (!p.chanstat_part || !p.chan_counters || p.N % 8 || p.out_t || (p.act & 0xff) == VSD_ACT_GEGLU ||
                         (p.split_k > 1 && !d->counters)))
    return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: chanstat_out needs chanstat_part + chan_counters, N %% 8 == 0, a plain (non-transposed, "
                    "non-GEGLU) output and the in-kernel split-K form");
  if (p.ln_part && (p.ksize != 1 || !p.ln_s || !p.ln_t || p.ln_groups <= 0 || p.bias))
    return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: fused LayerNorm needs a 1x1 layer, ln_s/ln_t and no separate bias");
  if (p.split_k > 1 && !p.ws_partial) return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: split_k needs a workspace");
  int BM, BN;
  switch (d->tile) {
    case VSD_TILE_128x128: BM = 128; BN = 128; break;
    case VSD_TILE_128x64: BM = 128; BN = 64; break;
    case VSD_TILE_64x64: BM = 64; BN = 64; break;
    case VSD_TILE_64x128: BM = 64; BN = 128; break;
    case VSD_TILE_256x128: BM = 256; BN = 128; break;
    case VSD_TILE_256x64: BM = 256; BN = 64; break;
    case VSD_TILE_256x256: BM = 256; BN = 256; break;
    default: return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: tile %d", d->tile);
  }
  if (halo) {
    const bool simple_epi = !p.out2 && !p.residual2 && !p.out_t && !p.rowstat_out && !p.chanstat_out && !p.ln_part &&
                            p.out_scale == 1.0f && !p.out_scale_dev &&
                            ((p.act & 0xff) == VSD_ACT_NONE || (p.act & 0xff) == VSD_ACT_RELU || (p.act & 0xff) == VSD_ACT_SILU) &&
                            !((p.act & VSD_ACT_POST) && (p.act & 0xff) != VSD_ACT_RELU);
    if (!p.halo_ok || p.ksize != 3 || p.stride != 1 || p.pad != 1 || BM < 128 || (p.N % 8 && !c64) || p.c0 % 64 || p.c1 % 64 || !simple_epi)
      return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: the halo-patch form (pipeline 7) needs a 3x3 stride-1 conv, Cin %% 64 == 0 per "
                      "source, a 128- or 256-row tile and the plain epilogue");
  }
  if (c64) {
    const bool thin = p.N <= 8 && p.ldo == 8 && !p.residual && !(p.act & VSD_ACT_POST);  // (the 64 -> 3 / 64 -> 4 projections: conv_c64_thin_kernel)
    if ((p.N != 64 && !thin) || p.cin != 64 || p.c1 != 0 || BM != 256 || BN != 64 || p.split_k != 1 || p.ldo % 8 || (size_t)p.M * p.ldo * 2 >= 0x7fffffffull)
      return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: the persistent 64-channel form (pipeline 10) needs Cin = 64 from one source, Cout = 64 (or Cout <= 8 "
                      "with ldo = 8, no residual), tile 256x64, no split over K and an output below 2 GB");
  }
  if (BN == 256 && (stages < 8 || p.split_k != 1 || p.chanstat_out))
    return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: the 256x256 tile exists on eight waves (pipeline 8 or 9), unsplit and without fused channel "
                    "statistics (its epilogue runs in row bands)");
  if (!halo && BM == 256 && BN != 256 && (BN != 128 || !p.fast || (stages != 3 && stages != 5 && stages < 8)))
    return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: the 256x128 tile exists for the buffer-load path (Cin %% 64 == 0, no resize) "
                    "with the 3-stage ring (pipeline 3, 5, 8 or 9) only");
  if ((stages == 8 || stages == 9) && BM * BN < 128 * 128)
    return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: the eight-wave forms (pipelines 8, 9) exist for tiles of 128 x 128 and larger");
  if ((p.act & 0xff) == VSD_ACT_GEGLU) {
    if (BN % 128 || p.N % 128 || p.split_k != 1 || (!p.bias && !p.ln_part) || p.out_t)
      return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: GEGLU needs BN=128 or 256, N %% 128 == 0, bias, no split-K");
  }
  if ((p.act & 0xff) == VSD_ACT_SOFTMAX) {
    if (BN != 128 || halo || p.N % 128 || (p.split_k != 1 && !d->counters) || p.out_t || p.out2 || p.residual || p.rowstat_out || p.chanstat_out ||
        p.softmax_cols < 1 || p.softmax_cols > 128 || p.ldo % 8)
      return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: the softmax epilogue needs BN=128, N %% 128 == 0, split-K only in the in-kernel form, a plain output and "
                      "1 <= softmax_cols <= 128");
  }
  if (p.out_t && (p.t_col0 % BN)) return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: t_col0 must be a multiple of BN");
  p.tiles_m = halo ? p.batch * cdiv(p.ho, BM / 16) * cdiv(p.wo, 16) : cdiv(p.M, BM);  // halo: 8x16 / 16x16 pixel patches
  p.tiles_n = cdiv(p.N, BN);
  const int KT = halo ? p.cin / BK : p.Kp / BK;  // the halo form splits over channel blocks (each = 9 K tiles)
  if (p.split_k > KT) p.split_k = KT;
  p.kt_per_split = cdiv(KT, p.split_k);
  p.split_k = cdiv(KT, p.kt_per_split);
  if (p.split_k == 1) p.counters = nullptr;
  if (p.counters && p.tiles_m * p.tiles_n > VSD_SPLITK_MAX_TILES)
    return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm: %d tiles exceed the split-K counter buffer", p.tiles_m * p.tiles_n);
  const int grid = p.tiles_m * p.tiles_n * p.split_k;
  {
    // fabric bytes of the two workgroup orders (see block_to_tile): an operand shared inside an XCD crosses once, the
    // other one once per XCD that needs it
    const double a_bytes = (double)p.batch * p.img_in * p.cin * 2.0, w_bytes = (double)p.N * p.Kp * 2.0;
    const int G = p.tiles_n * p.split_k;
    const int gx = cdiv(G, 8);                                        // groups per XCD in order 0
    const int nslice = p.split_k < gx ? p.split_k : gx;               // K slices of the input one XCD touches
    const double by_w = w_bytes + a_bytes * (G < 8 ? G : 8) * nslice / (double)p.split_k;
    const double by_a = a_bytes + w_bytes * (p.tiles_m < 8 ? p.tiles_m : 8);
    // the XCDs as a 2 x 4 / 4 x 2 grid over (M tiles, groups): rows cross 4 / 2 times, weights 2 / 4 times
    const bool ok24 = p.tiles_m % 2 == 0 && G % 4 == 0, ok42 = p.tiles_m % 4 == 0 && G % 2 == 0;
    const double by_24 = ok24 ? 4.0 * a_bytes + 2.0 * w_bytes : 1e300, by_42 = ok42 ? 2.0 * a_bytes + 4.0 * w_bytes : 1e300;
    const char* force = getenv("VSD_CONV_ORDER");  // (tests / benchmarking: 0 / 1 / 2 / 3 where valid, "1d" = never a grid)
    const bool no_grid = force && force[0] == '1' && force[1] == 'd';
    p.order = by_a < by_w ? 1 : 0;
    double best = by_a < by_w ? by_a : by_w;
    if (!no_grid && by_24 < best) { p.order = 2; best = by_24; }
    if (!no_grid && by_42 < best) { p.order = 3; best = by_42; }
    if (force && !no_grid) {
      const int f = atoi(force);
      if (f <= 1 || (f == 2 && ok24) || (f == 3 && ok42)) p.order = f;
    }
    p.gx = p.order == 2 ? G / 4 : (p.order == 3 ? G / 2 : 1);
    p.fd_gx = fast_div((unsigned)p.gx);
    const int S = p.order ? G : p.tiles_m;  // (block_to_tile)
    p.fd_span = fast_div(8u * (unsigned)S);
    p.fd_s = fast_div((unsigned)S);
    p.fd_tiles_n = fast_div((unsigned)p.tiles_n);
    p.fd_hw_out = fast_div((unsigned)p.hw_out);
    p.fd_wo = fast_div((unsigned)p.wo);
    const int ppr = cdiv(p.wo, 16), tpi = cdiv(p.ho, BM / 16) * ppr;  // halo: patches per row / per image (8x16 or 16x16 pixels)
    p.fd_ppr = fast_div((unsigned)ppr);
    p.fd_tpi = fast_div((unsigned)tpi);
  }
#ifdef VSD_WG_TIMELINE
  p.wgtl = grouped ? nullptr : wgtl_claim(grid);
#endif
  cl.BM = BM;
  cl.BN = BN;
  cl.grid = grid;
  cl.stages = stages;
  cl.halo = halo;
  cl.c64 = c64;
  if (c64 && cl.grid > ctx->num_cus) cl.grid = ctx->num_cus;  // persistent: a workgroup per CU walks over the patches
  return VSD_OK;
}

extern "C" int vsd_conv_gemm(vsd_ctx* ctx, const vsd_conv_desc* d, void* stream) {
  if (!ctx || !d) return VSD_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  ConvLaunch cl;
  int rc0 = conv_setup(ctx, d, cl, false);
  if (rc0 != VSD_OK) return rc0;
  const ConvParams& p = cl.p;
  const int BM = cl.BM, BN = cl.BN, grid = cl.grid, stages = cl.stages;
  const bool halo = cl.halo;
  {
    LaunchScope ls(ctx, s, VSD_FAM_CONV_GEMM, 2.0 * p.M * (double)p.N * p.K);
    if (cl.c64) vsd_launch_conv_c64(p, grid, s);
    else if (halo) vsd_launch_conv_halo(p, BM, BN, grid, s);
    else if (BM == 256 && BN == 256) vsd_launch_conv_256x256(p, grid, stages, s);
    else if (BM == 256) vsd_launch_conv_256x128(p, grid, stages, s);
    else if (BM == 128 && BN == 128) vsd_launch_conv_128x128(p, grid, stages, s);
    else if (BM == 128 && BN == 64) vsd_launch_conv_128x64(p, grid, stages, s);
    else if (BM == 64 && BN == 64) vsd_launch_conv_64x64(p, grid, stages, s);
    else vsd_launch_conv_64x128(p, grid, stages, s);
    int rc = ls.finish();
    if (rc) return rc;
  }
#ifdef VSD_PROBE
  { static const bool skip = getenv("VSD_SKIP_REDUCE") != nullptr; if (skip) return VSD_OK; }
#endif
  if (p.split_k > 1 && !p.counters) {
    LaunchScope ls(ctx, s, VSD_FAM_SPLITK_REDUCE, 0.0);
    size_t total = (size_t)p.M * ((p.N + 7) / 8);
    int g = (int)((total + 255) / 256);
    if (g > 2048) g = 2048;
    vsd_launch_splitk_reduce(p, g, s);
    int rc = ls.finish();
    if (rc) return rc;
  }
  return VSD_OK;
}

// Several independent conv / linear problems as ONE launch (conv_gemm_group_kernel, conv_kernels.h): every descriptor as for
// vsd_conv_gemm, with these restrictions -- the same tile (64x64, 64x128, 128x64 or 128x128) and pipeline (3 or 5) for all,
// the buffer-load operand path for all (Cin % 64 == 0 per source, no resize); a member split over K brings a workspace of its own
// (and, for the in-launch reduction, a counter slice of its own).  The members that leave slabs behind are reduced by ONE more
// launch for the whole group (splitk_reduce_group_kernel).
extern "C" int vsd_conv_gemm_group(vsd_ctx* ctx, const vsd_conv_desc* descs, int n, void* stream) {
  if (!ctx || !descs) return VSD_ERR_ARG;
  if (n < 1 || n > VSD_GROUP_MAX) return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm_group: %d problems (1..%d)", n, VSD_GROUP_MAX);
  hipStream_t s = (hipStream_t)stream;
  ConvGroup g;
  int BM = 0, BN = 0, stages = 0, grid = 0, rgrid = 0, nreduce = 0, last_reduce = 0;
  double flops = 0.0;
  for (int i = 0; i < n; ++i) {
    ConvLaunch cl;
    int rc = conv_setup(ctx, &descs[i], cl, true);
    if (rc != VSD_OK) return rc;
    if (cl.halo || cl.c64 || cl.BM == 256 || !cl.p.fast || cl.p.generic || (cl.stages != 3 && cl.stages != 5 && cl.stages < 8))
      return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm_group: member %d needs a 64- / 128-row tile, pipeline 3, 5, 8 or 9 and the buffer-load operand "
                      "path (Cin %% 64 == 0 per source, no resize)", i);
    if (cl.p.split_k > 1 && !cl.p.counters) {
      size_t total = (size_t)cl.p.M * ((cl.p.N + 7) / 8);
      int rg = (int)((total + 255) / 256);
      rgrid = rg > rgrid ? rg : rgrid;
      ++nreduce;
      last_reduce = i;
    }
    if (i == 0) {
      BM = cl.BM; BN = cl.BN; stages = cl.stages;
    } else if (cl.BM != BM || cl.BN != BN || cl.stages != stages) {
      return vsd_fail(ctx, VSD_ERR_ARG, "conv_gemm_group: member %d has another tile / pipeline than member 0", i);
    }
    g.p[i] = cl.p;
    g.start[i] = grid;
    grid += cl.grid;
    flops += 2.0 * cl.p.M * (double)cl.p.N * cl.p.K;
  }
  for (int i = n; i <= VSD_GROUP_MAX; ++i) g.start[i] = grid;
  for (int i = n; i < VSD_GROUP_MAX; ++i) g.p[i] = g.p[0];
  g.n = n;
  {
    LaunchScope ls(ctx, s, VSD_FAM_CONV_GEMM, flops);
    if (BM == 128 && BN == 128) vsd_launch_conv_group_128x128(g, grid, stages, s);
    else if (BM == 128 && BN == 64) vsd_launch_conv_group_128x64(g, grid, stages, s);
    else if (BM == 64 && BN == 64) vsd_launch_conv_group_64x64(g, grid, stages, s);
    else vsd_launch_conv_group_64x128(g, grid, stages, s);
    int rc = ls.finish();
    if (rc) return rc;
  }
  if (nreduce) {
    LaunchScope ls(ctx, s, VSD_FAM_SPLITK_REDUCE, 0.0);
    if (rgrid > 2048) rgrid = 2048;
    if (nreduce == 1) vsd_launch_splitk_reduce(g.p[last_reduce], rgrid, s);
    else vsd_launch_splitk_reduce_group(g, rgrid, s);
    return ls.finish();
  }
  return VSD_OK;
}

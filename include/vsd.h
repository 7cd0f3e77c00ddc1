/* libvsd — C-ABI of the MI355X (gfx950) per-frame SD1.5/LCM denoising kernels.
 *
 * Drop-in boundary: the reference has no FFI; its seam is the Python class VideoSDPipeline
 * (/root/reference/diffusert/videopipeline.py:11-128) whose `infer` drives diffusers modules on
 * PyTorch-CUDA.  This library replaces everything below that class: each entry point here stands in
 * for the library kernels one reference call site launches (cited per function).  The host side
 * (videosd_amd/engine.py, Python like the reference) sequences these calls once, captures them into a
 * hipGraph and replays the graph per frame.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name says `host`; caller-owned; no torch types.
 *   - activations: fp16, channels-last ("NHWC"): a [H*W][C] row-major matrix, row stride given as ld*.
 *   - weights: fp16, [N][Kp] row-major with K = ksize*ksize*Cin ordered (ky, kx, c), Kp = K rounded
 *     up to 64 and zero-filled (packing: videosd_amd/packing.py).
 *   - `stream` is a hipStream_t passed as void*.
 *   - return value: 0 = ok, negative = vsd_status; vsd_last_error() gives the text.  Never aborts.
 */
#ifndef VSD_H
#define VSD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vsd_ctx vsd_ctx;

enum vsd_status {
  VSD_OK = 0,
  VSD_ERR_ARG = -1,      /* bad shape / unsupported configuration */
  VSD_ERR_HIP = -2,      /* a HIP runtime call failed */
  VSD_ERR_NOMEM = -3,
  VSD_ERR_STATE = -4     /* call order (e.g. graph end without begin) */
};

enum vsd_act { VSD_ACT_NONE = 0, VSD_ACT_RELU = 1, VSD_ACT_SILU = 2, VSD_ACT_GEGLU = 3, VSD_ACT_QUICKGELU = 4,
               VSD_ACT_SOFTMAX = 5 /* row softmax inside every 128-column tile over its first softmax_cols columns (tile
                                      128-wide, N % 128 == 0, split-K only with `counters`): cross-attention probabilities, see softmax_cols */,
               VSD_ACT_GELU = 6 /* erf GELU (the MLP of SDXL's second text encoder; general epilogue walk) */,
               VSD_ACT_POST = 256 /* flag: apply the activation AFTER the residual adds (TAESD block) */ };

/* tile shapes of the implicit-GEMM kernel (BM x BN output tile per 256-thread workgroup) */
enum vsd_tile { VSD_TILE_128x128 = 0, VSD_TILE_128x64 = 1, VSD_TILE_64x64 = 2, VSD_TILE_64x128 = 3,
                VSD_TILE_256x128 = 4 /* Cin % 64 == 0, no resize, pipeline 3, 5 or 7 only */,
                VSD_TILE_256x64 = 5 /* pipeline 7 (halo patch) only */,
                VSD_TILE_256x256 = 6 /* eight waves (pipeline 8 or 9), Cin % 64 == 0, no resize, unsplit, no chanstat_out */ };

/* kernel families for vsd_stage_times */
enum vsd_family {
  VSD_FAM_CONV_GEMM = 0, VSD_FAM_SPLITK_REDUCE = 1, VSD_FAM_GROUPNORM = 2, VSD_FAM_LAYERNORM = 3,
  VSD_FAM_ATTENTION = 4, VSD_FAM_ELEMENTWISE = 5, VSD_FAM_COUNT = 6
};

/* Version of this interface (bumped whenever a struct grows or an entry point is added; round 3 = 3, round 4 = 4, round 5 = 5: pipeline 8 -- the stream-K form -- left the library; round 6 = 6: pipelines 8 / 9 / 10, vsd_groupnorm_launches, vsd_plan_*) and the size in
 * bytes of vsd_conv_desc as the LIBRARY was built: a caller compares both with its own header before the first call
 * (videosd_amd/lib.py does) instead of passing a short struct to a stale libvsd.so. */
#define VSD_VERSION 6
int vsd_version(void);
int vsd_conv_desc_size(void);

/* One context per GPU / per worker process (reference: one Ray actor per GPU, videopipeline.py:11,20). */
vsd_ctx* vsd_create(int device_id);
void vsd_destroy(vsd_ctx* ctx);
const char* vsd_last_error(vsd_ctx* ctx);

/* ---- implicit-GEMM convolution / linear layer -------------------------------------------------
 * out[m][n] = epilogue( sum_k A[m][k] * W[n][k] ),  m = output pixel, k = (ky,kx,c).
 * A is gathered on the fly from one or two NHWC sources (channel concat), with optional nearest
 * resize of the source to (hi,wi) before the convolution and stride 1/2.
 * Replaces: F.conv2d / F.linear / torch.cat / F.interpolate(nearest) launched by diffusers'
 * ResnetBlock2D, Transformer2DModel, Downsample2D, Upsample2D, Attention and FeedForward under
 * lcm_controlnet.py:558 (ControlNet), :568 (UNet), :299/:594 (TAESD).
 * epilogue: v = acc + bias[n] + rowvec[n]; v = act(v); v *= out_scale; v += residual + residual2;
 *           [act | VSD_ACT_POST: the activation is applied here instead]  out = v;  out2 = v + add2.
 * act == GEGLU: weights/bias are tile-packed (64 hidden + 64 gate rows per 128-row tile); the output
 * has n/2 columns: out[m][j] = (h_j + b) * gelu_erf(g_j + b).
 * Columns >= t_col0 (when out_t != NULL) are written TRANSPOSED to out_t[(n - t_col0) * ldt + m]
 * (this is how the attention V operand is produced as V^T).                                        */
typedef struct vsd_conv_desc {
  const void* src0;
  const void* src1;       /* second concat source or NULL */
  int32_t c0, c1;         /* channels of src0 / src1 (multiples of 8; of 64 when c1 != 0) */
  int32_t hs, ws;         /* stored spatial size of the sources */
  int32_t hi, wi;         /* logical input size after nearest resize (== hs, ws when no resize) */
  int32_t ho, wo;         /* output spatial size; M = ho * wo */
  int32_t ksize, stride, pad;
  const void* weight;     /* fp16 [n][kp] */
  int32_t n, k, kp;
  const void* bias;       /* fp16 [n] or NULL */
  const void* rowvec;     /* fp16 [n] or NULL */
  const void* residual;   /* fp16 [M][ldr] or NULL */
  const void* residual2;  /* fp16 [M][ldr] or NULL */
  int32_t ldr;
  float out_scale;
  int32_t act;            /* vsd_act */
  void* out;              /* fp16 [M][ldo] */
  int32_t ldo;
  void* out2;             /* optional: out2 = out + add2, both [M][ldo] */
  const void* add2;
  void* out_t;            /* optional transposed output */
  int32_t ldt, t_col0;
  int32_t tile;           /* vsd_tile */
  int32_t split_k;        /* >= 1; > 1 needs workspace of split_k * M * n floats */
  void* workspace;
  int32_t pipeline;       /* main-loop form: 0 = register-staged double buffer; 3 or 4 = direct-to-LDS ring with
                             that many stages (global_load_lds, counted vmcnt); 5 / 6 = the 3- / 4-stage ring with the
                             DMA issues interleaved between the MFMAs (single-basic-block iterations); 7 = halo patch:
                             3x3 stride-1 convs only (Cin % 64 == 0 per source, tile 128x128, 128x64, 256x128 or
                             256x64, plain epilogue): the (8+2)x(16+2) input patch of a 64-channel block is
                             staged in LDS once and serves all nine taps; 8 / 9 = the 3-stage ring (plain / interleaved)
                             on eight waves (tiles of 128x128 and larger, buffer-load path); 10 = the persistent form for
                             3x3 stride-1 convs with Cin = Cout = 64 from one source (every conv of a TAESD block; tile
                             256x64, unsplit, plain epilogue): weights resident in registers / LDS, one workgroup per CU
                             walking over 16x16-pixel patches, double-buffered patch fetch, register epilogue (round 6;
                             csrc/conv_c64.hip.  Round 2's one-wave-per-SIMD form of it measured -7 % and was removed in
                             round 3, as were an 8-stage ring and a weight-streaming form for M <= 192). */
  void* rowstat_out;      /* optional fp32 [M][n/64][2]: per output row, (sum, sum of squares) of the fp16 outputs over
                             each 64-column group -- the LayerNorm statistics of the NEXT layer, for free */
  void* chanstat_out;     /* optional fp32 [n][2]: per output CHANNEL, (sum, sum of squares) of the fp16 outputs over all M
                             rows (the reference-only mode's AdaIN statistics, vsd_adain).  Needs
                             chanstat_part (fp32 scratch, ceil(M/BM) * n * 2 floats, BM >= 64) and chan_counters
                             (ceil(n/64) int32, all zero; left at zero).  The last workgroup of each column block folds
                             the per-tile partials in tile order: deterministic. */
  void* chanstat_part;
  void* chan_counters;
  const void* ln_part;    /* optional: fused LayerNorm of the A operand.  Row partials as written by the producer's
                             rowstat_out, fp32 [M][ln_groups][2]; the weights must hold W*gamma, ln_s[n] = sum_k of
                             those fp16 weights, ln_t[n] = sum_k beta[k] W[n][k] + bias[n] (fp32 [n], packed like the
                             weights); then out = rstd*(acc - mean*ln_s) + ln_t replaces acc + bias.  1x1 layers only. */
  int32_t ln_groups;
  float ln_eps;
  const void* ln_s;
  const void* ln_t;
  void* counters;         /* optional: VSD_SPLITK_MAX_TILES int32, all zero.  When given, a split-K launch reduces
                             inside the kernel (the last workgroup to arrive at a tile sums the slabs in a fixed
                             order and runs the epilogue, leaving its counter at zero); when NULL a second
                             kernel (splitk_reduce) does it.  Results are bit-identical either way. */
  int32_t batch;          /* images stacked along M (0 or 1: a single image): M = batch * ho * wo, every image has the
                             geometry above, image b's rows are [b*ho*wo, (b+1)*ho*wo) of the output / residuals and
                             its source pixels start at row b*hs*ws of src0 / src1.  Several frames (of independent
                             streams) share one launch and one pass over the weights. */
  int32_t t_img;          /* transposed output with batch > 1: image b's row m goes to column b*t_img + (m - b*ho*wo)
                             of out_t (t_img >= ho*wo, a multiple of 8; 0 = ho*wo) */
  const void* out_scale_dev; /* optional: ONE fp32 in device memory that replaces out_scale (read by the kernel at run
                             time): the ControlNet zero-convs' conditioning scale (lcm_controlnet.py:558-566) can then
                             change under a captured graph.  General epilogue only (not the halo-patch form). */
  int32_t softmax_cols;   /* VSD_ACT_SOFTMAX: valid columns per 128-column group (1..128); the others are written as 0.
                             With the key projections of a fixed key set folded into the query weights
                             (scores_h = LN(x) (scale K_h Wq_h)^T, one 128-column group per head), the GEMM tile IS the score
                             block of one head and this epilogue turns it into probabilities: cross-attention over the 77
                             text tokens (Attention.forward of attn2 under lcm_controlnet.py:568) as two plain GEMMs. */
} vsd_conv_desc;
#define VSD_SPLITK_MAX_TILES 16384

int vsd_conv_gemm(vsd_ctx* ctx, const vsd_conv_desc* d, void* stream);

/* `n` (1..VSD_CONV_GROUP_MAX) INDEPENDENT problems as one launch: the reference runs the ControlNet's 13 zero-conv residual
 * merges of a denoising step (`down_block_additional_residuals`, lcm_controlnet.py:558-577) as 13 cuDNN launches plus 13 adds;
 * here each is one descriptor: a 1x1 conv with a device-side scale and the UNet tensor as residual, and seven / six of them share a
 * grid.  Every member as for vsd_conv_gemm, restricted to ONE kernel form for all: the same tile (64x64, 64x128, 128x64 or
 * 128x128) and pipeline (3 or 5), the buffer-load operand path (Cin % 64 == 0 per source, no resize), split-K only with
 * `counters` (and then a workspace and a counter slice per member).  Same bits as the members launched one by one. */
#define VSD_CONV_GROUP_MAX 8
int vsd_conv_gemm_group(vsd_ctx* ctx, const vsd_conv_desc* descs, int n, void* stream);

/* ---- fused per-token chains of a BasicTransformerBlock at the 320-wide level (csrc/fused_tail.hip) ---------------------
 * One workgroup owns 64 tokens for the whole chain; only the weights stream.  All matrices fp16 row-major, C = 320.
 * Weight operands are the packed forms of videosd_amd/packing.py ([N][K]; LayerNorm-consuming layers hold W*gamma with
 * ln_s / ln_t fp32 [N]; the GEGLU layer is tile-packed, 64 hidden + 64 gate rows per 128).  Replace, for that level, the
 * Attention.to_out / Attention.to_q / FeedForward / proj_out launches of diffusers' BasicTransformerBlock and
 * Transformer2DModel under lcm_controlnet.py:558,568.
 *   vsd_tail_a:  h1 = att W_out^T + b_out + h ;  q = LN(h1) W_q'^T                      (h1, q: [m][320])
 *   vsd_tail_b:  h2 = att2 W_out^T + b_out + h1 ;  h3 = FF2(GEGLU(LN(h2) W_ff1'^T)) + b_ff2 + h2 ;
 *                out = h3 W_proj^T + b_proj + x                                           (out: [m][320])          */
int vsd_tail_a(vsd_ctx* ctx, const void* att, const void* h, int m, const void* w_out, const void* b_out, const void* w_q,
               const void* ln_s, const void* ln_t, float ln_eps, void* h1_out, void* q_out, void* stream);
int vsd_tail_b(vsd_ctx* ctx, const void* att2, const void* h1, const void* x, int m, const void* w_out, const void* b_out,
               const void* w_ff1, const void* ln_s, const void* ln_t, float ln_eps, const void* w_ff2, const void* b_ff2,
               const void* w_proj, const void* b_proj, void* out, void* stream);

/* ---- GroupNorm (+SiLU) over an NHWC tensor, optionally the channel-concat of two tensors ---------
 * Replaces torch.nn.GroupNorm + SiLU in ResnetBlock2D / Transformer2DModel / conv_norm_out.
 * workspace: >= vsd_groupnorm_workspace_bytes(hw, c0 + c1, groups) bytes.                            */
int64_t vsd_groupnorm_workspace_bytes(int hw, int c, int groups);
int vsd_groupnorm(vsd_ctx* ctx, const void* src0, const void* src1, int c0, int c1, int hw, int groups, float eps,
                  const void* gamma, const void* beta, int silu, void* out, void* workspace, void* stream);
/* `batch` images stacked along the rows ([batch*hw][C]), each normalised with its own statistics.
 * workspace: >= batch * vsd_groupnorm_workspace_bytes(hw, c0 + c1, groups) bytes.                     */
int vsd_groupnorm_batched(vsd_ctx* ctx, const void* src0, const void* src1, int c0, int c1, int hw, int batch, int groups,
                          float eps, const void* gamma, const void* beta, int silu, void* out, void* workspace,
                          void* stream);
/* kernel launches vsd_groupnorm_batched issues for a shape: 1 (one workgroup per (image, group): small images) or 2 (statistics +
 * apply); 0 for a shape it refuses.  For launch accounting (Engine.launches_by_kind): the library's own decision, not a restatement. */
int vsd_groupnorm_launches(int c0, int c1, int hw, int batch, int groups);
/* ---- LayerNorm over the last dimension (BasicTransformerBlock.norm1/2/3, CLIP layer norms) -------- */
int vsd_layernorm(vsd_ctx* ctx, const void* x, int rows, int c, const void* gamma, const void* beta, float eps,
                  void* out, void* stream);

/* ---- fused softmax(Q K^T * scale) V, flash style (replaces F.scaled_dot_product_attention) ---------
 * q: [sq][ldq] with head h at columns [h*d, (h+1)*d); k likewise; vt: V transposed, row (h*d + j) holds
 * component j of every key, row stride ldvt >= round_up(sk, 64) and ZERO beyond sk.
 * d in {8..160}, multiple of 8.  causal != 0: key j visible to query i iff j <= i (CLIP).            */
int vsd_attention(vsd_ctx* ctx, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* out,
                  int ldo, int sq, int sk, int heads, int d, float scale, int causal, void* stream);

/* `batch` independent problems in one launch: image b uses q / out rows [b*sq, (b+1)*sq), k rows from b*k_batch_rows
 * (sk for self-attention, 0 when all images share the keys, e.g. the text) and V^T columns from b*vt_batch_cols
 * (a multiple of 8; 0 when shared).  V^T beyond an image's sk keys must be finite (the next image or zeros).   */
int vsd_attention_batched(vsd_ctx* ctx, const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* out,
                          int ldo, int sq, int sk, int heads, int d, float scale, int causal, int batch, int k_batch_rows,
                          int vt_batch_cols, void* stream);

/* ---- per-frame elementwise kernels -----------------------------------------------------------------
 * Latent-like tensors (4 channels) are stored with row stride 8 (channels 4..7 zero).                 */

/* u8 RGB HWC -> fp16 [h*w][8]: channels 0..2 = TAESD encoder input ((x+1)/2 of the [-1,1] image),
 * channels 3..7 = 0.  Replaces VaeImageProcessor.preprocess (lcm_controlnet.py:457) + H2D.           */
int vsd_preprocess_rgb(vsd_ctx* ctx, const void* rgb_u8, int h, int w, void* out, void* stream);

/* Sobel "canny" of the reference (canny_gpu.py:27-44) on device: L conversion, two 3x3 filters,
 * magnitude, division by the global max, thresholds, byte truncation.  Writes the u8 edge map (h*w) and
 * the ControlNet conditioning tensor fp16 [h*w][8] (edge/255 in channels 0..2).
 * workspace: >= vsd_sobel_workspace_bytes(h, w) bytes (one partial maximum per workgroup; no atomics, no
 * memset: two launches on the stream).                                                              */
int64_t vsd_sobel_workspace_bytes(int h, int w);
int vsd_sobel_control(vsd_ctx* ctx, const void* rgb_u8, int h, int w, float low, float high, void* edge_u8,
                      void* control_out, void* workspace, void* stream);

/* latents = sqrt_a * x0 + sqrt_b * noise  (LCMScheduler_X.add_noise, lcm_controlnet.py:1046-1071).
 * x0/out: fp16 [hw][8]; noise: fp32 NCHW [4][hw] (as torch.randn produces it).                        */
int vsd_add_noise(vsd_ctx* ctx, const void* x0, const void* noise_f32, float sqrt_a, float sqrt_b, int hw, void* out,
                  void* stream);

/* One LCMScheduler_X.step (lcm_controlnet.py:948-1043) on fp16 [hw][8] tensors:
 *   pred_x0 = (sample - sqrt_b * eps) / sqrt_a;  denoised = c_out * pred_x0 + c_skip * sample;
 *   prev = sqrt_a_prev * denoised + sqrt_b_prev * noise   (noise NULL => prev = denoised).
 * coef = {sqrt_a, sqrt_b, c_skip, c_out, sqrt_a_prev, sqrt_b_prev}.
 * dec_in (optional): 3*tanh(denoised/3), the TAESD decoder input clamp (DecoderTiny.forward).        */
int vsd_lcm_step(vsd_ctx* ctx, const void* eps, const void* sample, const void* noise_f32, const float* coef_host,
                 int hw, void* prev, void* denoised, void* dec_in, void* stream);

/* The two scheduler kernels with their coefficients in DEVICE memory (fp32: {sqrt_a, sqrt_b} / the six of
 * vsd_lcm_step) and `batch` images per launch (image b = rows [b*hw, (b+1)*hw); all images use the same noise draw,
 * as the reference's per-frame RNG reset implies).  A captured graph built on these follows a new `strength` (another
 * set of timesteps of the same count, lcm_controlnet.py:929-936) by rewriting the floats -- no re-capture
 * (server.py:163-197 patches options live, one slider step at a time). */
int vsd_add_noise_dev(vsd_ctx* ctx, const void* x0, const void* noise_f32, const void* coef_dev, int hw, int batch, void* out,
                      void* stream);
int vsd_lcm_step_dev(vsd_ctx* ctx, const void* eps, const void* sample, const void* noise_f32, const void* coef_dev, int hw,
                     int batch, void* prev, void* denoised, void* dec_in, void* stream);

/* AdaIN of the reference-only mode (lcm_reference_pipeline.py:593-603, dead at v2 but still exposed as `ref`):
 * out[r][c] = (x[r][c] - mean_c) / std_c * std_ref_c + mean_ref_c, statistics over the `rows` pixels of one image,
 * population variance clamped at eps before the square root.  stats / stats_ref: fp32 [c][2] per-channel (sum, sum of
 * squares) over those rows, as vsd_conv_gemm's chanstat_out writes them.  x, out: fp16 [rows][c]; may alias. */
int vsd_adain(vsd_ctx* ctx, const void* x, const void* stats, const void* stats_ref, int rows, int c, float eps, void* out,
              void* stream);

/* Per-prompt constants of the "absorbed" cross-attention (see softmax_cols above; csrc/prompt_fold.hip): for one transformer
 * block, fold the text's key / value projections into its query / output weights.  k: fp16 [tl][ldk] (to_k of the text),
 * vt: fp16 [c][ldvt] (to_v of the text, transposed, zero beyond tl), wq / wo: the raw fp16 [c][c] to_q / to_out.0 weights of attn2
 * (diffusers' Attention under lcm_controlnet.py:558,568), gamma / beta: fp16 [c] of the LayerNorm in front (norm2), scale =
 * head_dim^-0.5.  Writes xa1_w fp16 [heads*128][c] = (scale K_h Wq_h) * gamma with xa1_s / xa1_t fp32 [heads*128] (the
 * ln_s / ln_t of a LayerNorm-consuming vsd_conv_gemm layer) and xa2_w fp16 [c][heads*128] = Wo_h V_h^T; rows / columns of the
 * keys tl..127 of every head are zero.  c % 64 == 0, (c / heads) % 8 == 0, tl <= 128.  Runs once per prompt and layer. */
int vsd_xattn_fold(vsd_ctx* ctx, const void* k, int ldk, const void* vt, int ldvt, int tl, const void* wq, const void* wo,
                   const void* gamma, const void* beta, int c, int heads, float scale, void* xa1_w, void* xa1_s, void* xa1_t,
                   void* xa2_w, void* stream);

/* CLIP text embeddings (CLIPTextEmbeddings under lcm_controlnet.py:175): out[i] = token_emb[ids[i]] + pos_emb[i], i < n.
 * ids: int64 [n] in device memory (clamped to [0, vocab)); token_emb fp16 [vocab][c], pos_emb fp16 [>= n][c], out fp16 [n][c]. */
int vsd_embed_tokens(vsd_ctx* ctx, const void* ids_i64, const void* token_emb, const void* pos_emb, int n, int c, int vocab,
                     void* out, void* stream);

/* decoder output fp16 [hw][ld] (3 channels used; value c of the last conv) -> u8 RGB HWC:
 * y = fp16(2c - 1) (DecoderTiny), (y/2 + 0.5).clamp(0,1)*255 rounded half-to-even
 * (VaeImageProcessor.postprocess, lcm_controlnet.py:609-611).                                         */
int vsd_postprocess_rgb(vsd_ctx* ctx, const void* img, int ld, int hw, void* rgb_u8, void* stream);

/* out = a + b * scale   (fp16, n elements, n % 8 == 0) */
int vsd_axpy(vsd_ctx* ctx, const void* a, const void* b, float scale, int64_t n, void* out, void* stream);

/* ---- two independent operations as ONE grid per kernel ------------------------------------------------
 * The reference runs the ControlNet and then the UNet encoder of a denoising step -- the same topology with two weight sets on the
 * same latents (lcm_controlnet.py:539-577) -- as two sequences of cuDNN / cuBLAS launches.  Here the two walk in lock step:
 *     pair_begin(ctx); op(ctx, A...); pair_join(ctx); op(ctx, B...); pair_end(ctx, &joined);
 * every launch of the first operation is held back, and the second operation's k-th launch joins the k-th held one when it is
 * the same kernel with the same launch geometry on the same stream (one grid, gridDim.z = 2: half the launches, twice the
 * workgroups per launch); a launch that finds no partner goes out alone, in order -- the results never depend on what joined.
 * Pairable today: vsd_groupnorm*, vsd_attention*, vsd_tail_a / vsd_tail_b (two convolutions share a grid through
 * vsd_conv_gemm_group).  The two operations must not depend on each other and must use separate scratch.  `joined_out` (may be
 * null) receives the number of launches that went out as pairs.  While profiling (vsd_profile_begin) every launch goes out alone:
 * the per-family times are those of single launches. */
int vsd_pair_begin(vsd_ctx* ctx);
int vsd_pair_join(vsd_ctx* ctx);
int vsd_pair_end(vsd_ctx* ctx, int* joined_out);

/* ---- a prepared frame program from a FILE (round 6; SURVEY.md section 8b's whole-frame entry points) ---------------------------
 * The reference's seam is a Python class (videopipeline.py:75-128) and the sequencing of a frame lives in videosd_amd/engine.py; for a
 * host without Python the engine's program is EXPORTED (videosd_amd/plan.py export_plan: every C-ABI call of the one-stream form with
 * its arguments, device pointers as (region, offset), the bytes of weights / constants / prompt block) and replayed here.
 * vsd_plan_load: allocate, upload, patch, replay under capture (one hipGraph); the plan is one (frame size, steps, strength, ControlNet
 * scale, prompt, frames per launch).  vsd_plan_infer: frame(s) uint8 [batch][H][W][3] on the HOST in, the same shape out; synchronous.
 * The result is bit for bit the Python engine's.  vsd_plan_info: dims[0..2] = H, W, frames per launch. */
typedef struct vsd_plan vsd_plan;
int vsd_plan_load(vsd_ctx* ctx, const char* path, vsd_plan** plan_out);
/* lane 0..3: the plan launches on that launch stream of the process's pool (vsd_stream_pool) -- up to four plans in flight side by side,
 * as the Python workers' launch lanes; lane -1 = vsd_plan_load (a stream of the plan's own). */
int vsd_plan_load_lane(vsd_ctx* ctx, const char* path, int lane, vsd_plan** plan_out);
int vsd_plan_info(vsd_ctx* ctx, vsd_plan* plan, int* dims);
/* vsd_plan_infer in two halves: enqueue (upload, launch, download; the host buffers must stay valid and should be pinned for the
 * copies to overlap other lanes' work) | wait for this plan's stream. */
int vsd_plan_submit(vsd_ctx* ctx, vsd_plan* plan, const void* frame_u8_host, void* out_u8_host);
int vsd_plan_wait(vsd_ctx* ctx, vsd_plan* plan);
int vsd_plan_infer(vsd_ctx* ctx, vsd_plan* plan, const void* frame_u8_host, void* out_u8_host);
/* another prompt for a loaded plan: a file written by videosd_amd.plan.export_prompt (the prompt's constant block, same layout) */
int vsd_plan_load_prompt(vsd_ctx* ctx, vsd_plan* plan, const char* path);
void vsd_plan_free(vsd_ctx* ctx, vsd_plan* plan);
/* page-locked host memory for a plan's frames (NULL on failure) */
void* vsd_pinned_alloc(vsd_ctx* ctx, size_t bytes);
void vsd_pinned_free(vsd_ctx* ctx, void* p);

/* ---- hipGraph capture / replay (reference intent: compile_model, videopipeline.py:35-47) ----------- */
int vsd_graph_begin(vsd_ctx* ctx, void* stream);
int vsd_graph_end(vsd_ctx* ctx, void* stream, void** graph_exec_out);
int vsd_graph_launch(vsd_ctx* ctx, void* graph_exec, void* stream);
int vsd_graph_destroy(vsd_ctx* ctx, void* graph_exec);

/* ---- launch streams with a hardware queue and a command-processor pipe of their own; launch sequences (round 4) -----
 * Replace the reference's N independent actors per node (server.py:132-137, 317-321: one process, one CUDA context, one
 * set of streams per GPU worker) INSIDE one worker: several launches in flight on one GPU, placed deterministically.
 * What the placement has to respect on MI355X (scripts/queue_probe.cpp, scripts/pipe_probe.cpp): plain HIP streams share 4
 * hardware queues in creation order (aliasing = an accident of what the process created before); a CU-masked stream has a
 * queue of its own, but queue i of a process is served by command-processor pipe i mod 4 and two busy queues on one pipe
 * take turns in long slices (two frame-like chains: 2.55x the time of one; on different pipes 1.00x).
 * vsd_stream_pool: THE four launch streams of this process on the context's device (VSD_POOL_STREAMS; created together on
 *   the first call = four different pipes, never destroyed; every context of the process gets the same four).  Ordinary
 *   hipStream_t values (pass them as `stream` everywhere; torch wraps them with torch.cuda.ExternalStream).  The host side
 *   puts launch lane l on stream l mod 4 and its side branch on stream (l + 2) mod 4.
 * vsd_stream_pool_check: chains of `chain` dependent 10 us kernels on all four streams at once / one chain alone: ~1.0 when
 *   the four run side by side, >= 2 when two of them share a pipe.
 * vsd_stream_create / _destroy: a further CU-masked stream (cu_mask NULL = all CUs; `words` 32-bit words, bit i = CU i) for
 *   callers that partition the CUs themselves; its pipe is (number of CU-masked streams the process created before) mod 4.
 * vsd_seq: a frame's program as an ordered list of (single-branch graph executable -> stream), (record event on stream) and
 *   (stream waits for event) items; vsd_seq_launch issues them in order without waiting.  Graphs with parallel branches are
 *   avoided on purpose: two such executables in flight run one after the other on this runtime, single-branch graphs on
 *   different pipes overlap.  The sequence owns its graph executables and events. */
#define VSD_POOL_STREAMS 4
#define VSD_MAX_DEVICES 16
int vsd_stream_pool(vsd_ctx* ctx, void** streams_out /* [VSD_POOL_STREAMS] */);
int vsd_stream_pool_check(vsd_ctx* ctx, int chain, float* ratio_out);
typedef struct vsd_seq vsd_seq;
int vsd_stream_create(vsd_ctx* ctx, const uint32_t* cu_mask, int words, void** stream_out);
int vsd_stream_destroy(vsd_ctx* ctx, void* stream);
int vsd_seq_create(vsd_ctx* ctx, vsd_seq** seq_out);
int vsd_seq_add_graph(vsd_ctx* ctx, vsd_seq* seq, void* graph_exec, void* stream);
int vsd_seq_add_record(vsd_ctx* ctx, vsd_seq* seq, void* stream, int* event_out);
int vsd_seq_add_wait(vsd_ctx* ctx, vsd_seq* seq, void* stream, int event);
int vsd_seq_count(vsd_ctx* ctx, vsd_seq* seq, int* graphs, int* edges);
int vsd_seq_launch(vsd_ctx* ctx, vsd_seq* seq);
int vsd_seq_destroy(vsd_ctx* ctx, vsd_seq* seq);

/* ---- per-family device timing ------------------------------------------------------------------------
 * While profiling is on, every launch is bracketed by HIP events on its stream (do not capture graphs
 * in this mode).  vsd_stage_times synchronises and returns, per vsd_family: total ms, launch count and
 * the algorithmic FLOPs (2*M*N*K for conv_gemm; 4*sq*sk*heads*d for attention) accumulated since
 * vsd_profile_begin.                                                                                   */
int vsd_profile_begin(vsd_ctx* ctx);
int vsd_profile_end(vsd_ctx* ctx);
int vsd_stage_times(vsd_ctx* ctx, float* ms, int64_t* launches, double* flops);
/* average elapsed ms of an EMPTY event bracket on `stream` (the per-launch cost of the timing itself) */
int vsd_profile_overhead(vsd_ctx* ctx, void* stream, int n, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* VSD_H */

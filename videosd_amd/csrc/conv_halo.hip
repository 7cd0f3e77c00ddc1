// 3x3 convs with an LDS halo patch + the split-K reducer (see conv_kernels.h for the shared epilogue helpers).
#include "conv_kernels.h"

namespace {

// ---------------------------------------------------------------- split-K reducer
// One thread = 8 consecutive outputs of one row: the sum of the slabs in slab order, then the layer's epilogue.  A launch of
// this kernel is a few hundred KB to a few MB of fp32 spread over the chip -- latency, not bandwidth: every load of a thread
// (up to 8 slabs at a time and the residual chunk) is issued BEFORE the first is consumed, so that the
// thread pays one memory round trip instead of one per slab (the first form's runtime-bounded loop: a dependent chain of
// split_k round trips, then the epilogue's own loads).  The additions run in the same order: same bits.
__device__ __forceinline__ void splitk_reduce_body(const ConvParams& p) {
  VSD_CUT(VSD_CUT_REDUCE, p.cut)
  const int nch = (p.N + 7) / 8;
  const size_t total = (size_t)p.M * nch;
  const size_t slab = (size_t)p.M * p.N;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
    int m = (int)(q / nch);
    int n = (int)(q - (size_t)m * nch) * 8;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float* s = p.ws_partial + (size_t)m * p.N + n;
    const bool full = n + 8 <= p.N;
    const bool transposed = p.out_t && n >= p.t_col0;
    // the epilogue's operands of these 8 columns, fetched beside the slabs
    const bool pre_res = full && p.residual != nullptr && !transposed;
    // (bias and time vector are added one after the other by epilogue_store8, as the halo kernel's in-launch epilogue does -- the
    //  two forms of a split halo conv must give the same bits --, from the raw values fetched HERE, beside the slabs: loaded
    //  inside epilogue_store8 they were one more dependent round trip behind the slab sums in each of a lone frame's ~280 reducers)
    const bool pre_brv = false;
    const bool pre_raw = full && !p.ln_part;
    half8 rres = (half8){0, 0, 0, 0, 0, 0, 0, 0}, braw = rres, rvraw = rres;
    if (full) {
      constexpr int NB = 8;
      f32x4 lo[NB], hi[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int kk = k < p.split_k ? k : 0;  // (clamped: surplus loads hit slab 0's line again)
        lo[k] = *reinterpret_cast<const f32x4*>(s + kk * slab);
        hi[k] = *reinterpret_cast<const f32x4*>(s + kk * slab + 4);
      }
      if (pre_res) rres = *reinterpret_cast<const half8*>(p.residual + (size_t)m * p.ldr + n);
      if (pre_raw && p.bias) braw = *reinterpret_cast<const half8*>(p.bias + n);
      if (pre_raw && p.rowvec) rvraw = *reinterpret_cast<const half8*>(p.rowvec + n);
#pragma unroll
      for (int k = 0; k < NB; ++k)
        if (k < p.split_k) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] += lo[k][i];
            v[4 + i] += hi[k][i];
          }
        }
      for (int k0 = NB; k0 < p.split_k; k0 += NB) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          const int kk = k0 + k < p.split_k ? k0 + k : 0;
          lo[k] = *reinterpret_cast<const f32x4*>(s + kk * slab);
          hi[k] = *reinterpret_cast<const f32x4*>(s + kk * slab + 4);
        }
#pragma unroll
        for (int k = 0; k < NB; ++k)
          if (k0 + k < p.split_k) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              v[i] += lo[k][i];
              v[4 + i] += hi[k][i];
            }
          }
      }
    } else {
      for (int k = 0; k < p.split_k; ++k) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (n + i < p.N) v[i] += s[k * slab + i];
      }
    }
    if (p.ln_part) {
      float mean, rstd;
      ln_row_stats(p, m, mean, rstd);
      ln_transform8(p, n, mean, rstd, v);
    }
    float rs = 0.f, rq = 0.f;
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    epilogue_store8(p, m, n, v, rs, rq, pre_res, rres, pre_brv, z4, z4, pre_raw, braw, rvraw);
  }
}
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvParams p) { splitk_reduce_body(p); }
// the reducers of a conv group (vsd_conv_gemm_group) as one grid: blockIdx.y = member; members that were not split, or reduced in
// their own launch, have nothing to do
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(const ConvGroup g) {
  const ConvParams& p = g.p[blockIdx.y];
  if (p.split_k <= 1 || p.counters) return;
  splitk_reduce_body(p);
}

// ---------------------------------------------------------------- 3x3 conv with an LDS halo patch
// The implicit-GEMM kernel above fetches every input pixel nine times per output-channel tile (once per tap), and at
// these tile sizes it is the L2 -> LDS fill path that binds.  Here a workgroup's 128 output pixels are an 8 x 16 PATCH
// of one image; for each 64-channel block the (8+2) x (16+2) input patch is staged in LDS ONCE and all nine taps read
// it at shifted row indices (a tap is an LDS address offset, not a memory fetch): input traffic / 6.4, LDS-DMA issues
// for the input / 6 (6 per wave per channel block instead of 4 per wave per tap).  Only the weight tiles stream per tap
// (3-slot ring, counted vmcnt).  Zero padding = out-of-range buffer offsets, as in the FAST path.
// K order inside a workgroup: channel block outer, tap inner (weights stay [N][(ky,kx,c)]).  Split-K over channel
// blocks writes fp32 slabs for splitk_reduce_kernel, or (p.counters) for the last workgroup to arrive at the tile.  Epilogue: bias + time vector, activation, one residual.
// Eligible: ksize 3, stride 1, pad 1, Cin % 64 == 0 (each concat source), with or without the folded nearest resize
// (Upsample2D + conv: the patch fetch reads each source pixel into the halo rows that show it); patches hanging over the
// right / bottom edge compute but do not store their outside pixels.
// WMN = waves along M (2: 128-pixel 8x16 patch, 256 threads; 4: 256-pixel 16x16 patch, 512 threads -- the weight tile
// is then shared by twice the pixels: half the weight traffic per FLOP, two waves per SIMD at one workgroup per CU).
// NSB = slots of the weight-tile ring (lead = NSB - 1 tiles).
template <int BN, int WMN, int NSB>
__global__ __launch_bounds__(128 * WMN) void conv_halo_kernel(const ConvParams p) {
  VSD_CUT(VSD_CUT_CONV_HALO, p.cut)
  prefetch_kernargs();
  WGTL_START()
  constexpr int NW = 2 * WMN, NT = 64 * NW;
  constexpr int BM = 64 * WMN, PW = 16, PH = 4 * WMN, HW_ = PW + 2;  // halo row length 18
  constexpr int HUSED = (PH + 2) * HW_;                                // 180 / 324 halo rows
  constexpr int HROWS = (HUSED + 8 * NW - 1) / (8 * NW) * (8 * NW);    // padded to whole wave-instructions per wave: 192 / 384
  constexpr int AI = HROWS / (8 * NW), BR = BN / (8 * NW);
  static_assert(BR >= 1, "tile too narrow for this many waves");
  constexpr int TM = 64, TN = BN / 2, FM = TM / 16, FN = TN / 16;
  constexpr int BNP = BN + 4;
  constexpr int A_HALFS = HROWS * BK, B_HALFS = BN * BK;
  constexpr int STAGE_BYTES = (2 * A_HALFS + NSB * B_HALFS) * 2;
  constexpr int EPI_BYTES = BM * BNP * 4;
  constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  half_t* Abuf = reinterpret_cast<half_t*>(smem);
  half_t* Bbuf = Abuf + 2 * A_HALFS;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  constexpr int OOB = (int)0x80000000;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;

  int tile_m, grp;
  block_to_tile(p, blockIdx.x, tile_m, grp);  // XCD-aware order, as in conv_gemm_kernel
  const int split = fdiv(grp, p.fd_tiles_n);
  const int tile_n = grp - split * p.tiles_n;
  const int n0 = tile_n * BN;
  // tile -> (image, patch origin)
  const int ppr = (p.wo + PW - 1) / PW, tpi = ((p.ho + PH - 1) / PH) * ppr;  // patches may hang over the right / bottom edge
  const int img = fdiv(tile_m, p.fd_tpi), trem = tile_m - img * tpi;  // (fd_tpi / fd_ppr: the host's multipliers for tpi / ppr)
  const int prow = fdiv(trem, p.fd_ppr);
  const int y0 = prow * PH, x0 = (trem - prow * ppr) * PW;
  const int pix0 = img * p.img_in;   // first SOURCE pixel of the image (hs x ws; nearest-resized to hi x wi = ho x wo on the fly)
  const int opix0 = img * p.hw_out;  // first output row of the image
  auto row_ok = [&](int r) { return y0 + (r >> 4) < p.ho && x0 + (r & 15) < p.wo; };
  auto row_m = [&](int r) { return opix0 + (y0 + (r >> 4)) * p.wo + x0 + (r & 15); };

  // channel blocks of this split
  const int ncb_all = p.cin / BK;
  const int cb_begin = split * p.kt_per_split;  // (kt_per_split counts channel blocks here)
  const int cb_end = min(ncb_all, cb_begin + p.kt_per_split);
  const int T = (cb_end - cb_begin) * 9;

  // ---- A patch DMA: this wave's instructions j = wave + 4q cover halo rows 8j..8j+7; lane -> (row, 16-byte slot)
  int apx[AI], alc[AI];
#pragma unroll
  for (int q = 0; q < AI; ++q) {
    const int row = 8 * (wave + NW * q) + (lane >> 3);
    const int hy = row / HW_, hx = row - hy * HW_;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool in = row < HUSED && (unsigned)y < (unsigned)p.hi && (unsigned)x < (unsigned)p.wi;
    // nearest resize folded in: several halo rows may show the same source pixel (fixed-point floor(y * hs / hi))
    const int sy = (int)(((unsigned)y * p.rmul_y) >> p.rshift), sx = (int)(((unsigned)x * p.rmul_x) >> p.rshift);
    apx[q] = in ? pix0 + sy * p.ws + sx : -1;
    alc[q] = (((lane & 7) ^ (row & 7)) << 4);
  }
  int bvoff[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const int r = 8 * (wave + NW * i) + (lane >> 3);
    const int n = n0 + r;
    bvoff[i] = n < p.N ? n * p.Kp * 2 + (((lane & 7) ^ (r & 7)) << 4) : OOB;
  }
  const int anr0 = (int)((size_t)p.batch * p.img_in * p.c0 * 2), anr1 = (int)((size_t)p.batch * p.img_in * p.c1 * 2);
  const int bnr = (int)((size_t)p.N * p.Kp * 2);

#define HALO_ISSUE_A(CB_, BUF_)                                                                         \
  {                                                                                                     \
    const int ch_ = (CB_) * BK;                                                                         \
    const bool second_ = ch_ >= p.c0;                                                                   \
    const int cs2_ = (second_ ? p.c1 : p.c0) * 2;                                                       \
    const int soff_ = (second_ ? ch_ - p.c0 : ch_) * 2;                                                 \
    const __amdgpu_buffer_rsrc_t rs_ =                                                                  \
        __builtin_amdgcn_make_buffer_rsrc((void*)(second_ ? p.src1 : p.src0), 0, second_ ? anr1 : anr0, 0x00020000); \
    half_t* dst_ = Abuf + (BUF_) * A_HALFS;                                                             \
    _Pragma("unroll") for (int q = 0; q < AI; ++q) {                                                    \
      const int vo_ = apx[q] >= 0 ? __mul24(apx[q], cs2_) + alc[q] : OOB;                               \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(dst_ + 8 * (wave_s + NW * q) * BK), 16, vo_, soff_, 0, 0); \
    }                                                                                                   \
  }
/* one wave-instruction (8 halo rows per wave) of the patch of channel block CB_: piece Q_ of AI */                     \
/* (vmcnt retires in order: a whole patch issued at once would have to land within the two tiles of lead the weight   \
   ring has; one piece per tap gives every piece that lead and the patch six taps of slack) */
#define HALO_ISSUE_A_PIECE(CB_, BUF_, Q_)                                                               \
  {                                                                                                     \
    const int ch_ = (CB_) * BK;                                                                         \
    const bool second_ = ch_ >= p.c0;                                                                   \
    const int cs2_ = (second_ ? p.c1 : p.c0) * 2;                                                       \
    const int soff_ = (second_ ? ch_ - p.c0 : ch_) * 2;                                                 \
    const __amdgpu_buffer_rsrc_t rs_ =                                                                  \
        __builtin_amdgcn_make_buffer_rsrc((void*)(second_ ? p.src1 : p.src0), 0, second_ ? anr1 : anr0, 0x00020000); \
    half_t* dst_ = Abuf + (BUF_) * A_HALFS;                                                             \
    int apx_ = apx[0], alc_ = alc[0];                                                                   \
    _Pragma("unroll") for (int q = 1; q < AI; ++q)                                                      \
      if (q == (Q_)) {                                                                                  \
        apx_ = apx[q];                                                                                  \
        alc_ = alc[q];                                                                                  \
      }                                                                                                 \
    const int vo_ = apx_ >= 0 ? __mul24(apx_, cs2_) + alc_ : OOB;                                       \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(dst_ + 8 * (wave_s + NW * (Q_)) * BK), 16, vo_, soff_, 0, 0); \
  }
#define HALO_ISSUE_B(TAP_, CB_, SLOT_)                                                                  \
  {                                                                                                     \
    const int soff_ = ((TAP_) * p.cin + (CB_) * BK) * 2;                                                \
    const __amdgpu_buffer_rsrc_t rsb_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, bnr, 0x00020000); \
    half_t* dst_ = Bbuf + (SLOT_) * B_HALFS;                                                            \
    _Pragma("unroll") for (int i = 0; i < BR; ++i) {                                                    \
      const int bv_ = bvoff[i] + 0;                                                                     \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb_, (lds_ptr_t)(dst_ + 8 * (wave_s + NW * i) * BK), 16, bv_, soff_, 0, 0); \
    }                                                                                                   \
  }

#ifdef VSD_CONV_PROBE
  long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long plast = __builtin_readcyclecounter();
#endif
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment bases: A halo row of this lane's pixel for tap (0,0); B row
  int hr0[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) hr0[i] = (wm * (TM / PW) + i) * HW_ + fr;

  if (T > 0) {
    HALO_ISSUE_A(cb_begin, 0)
#pragma unroll
    for (int st = 0; st < NSB - 1; ++st)
      if (st < T) HALO_ISSUE_B(st % 9, cb_begin + st / 9, st)
  }
  constexpr int LEAD = NSB - 1;
  int tap = 0, cb = cb_begin, slot = 0;                 // of iteration t
  int tap2 = LEAD % 9, cb2 = cb_begin + LEAD / 9;        // of the tile fetched in iteration t (t + LEAD)
  // An iteration issues [weight tile t+LEAD, then one piece of the next patch].  vmcnt retires in order, so "tile t has
  // landed" = all but the operations issued after it are done: the pieces of iterations t-LEAD .. t-1 and the weight
  // tiles t+1 .. t+LEAD-1.  (Counted waits need immediates: the steady-state values are exact, the tail over-waits.)
  int hist = 0;  // bit i: iteration t-1-i issued a patch piece
  CPROBE(0)
  for (int t = 0; t < T; ++t) {
    const int later_b = min(LEAD - 1, T - 1 - t) * BR;
    const int nwait = later_b + __builtin_popcount(hist & ((1 << LEAD) - 1));
    if (nwait >= (LEAD - 1) * BR + LEAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LEAD - 1) * BR + LEAD) : "memory");
    else if (nwait >= (LEAD - 1) * BR + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LEAD - 1) * BR + 1) : "memory");
    else if (nwait >= (LEAD - 1) * BR) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LEAD - 1) * BR) : "memory");
    else if (nwait >= BR) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BR) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CPROBE(1)
    __builtin_amdgcn_s_barrier();
    CPROBE(2)
    if (t + LEAD < T) {
      int ns = slot + LEAD;
      if (ns >= NSB) ns -= NSB;
      HALO_ISSUE_B(tap2, cb2, ns)
    }
    hist <<= 1;
    if (tap < AI && cb + 1 < cb_end) {  // next channel block's patch, one piece per tap (taps 0..AI-1)
      HALO_ISSUE_A_PIECE(cb + 1, (cb + 1 - cb_begin) & 1, tap)
      hist |= 1;
    }
    CPROBE(3)
    const half_t* a = Abuf + ((cb - cb_begin) & 1) * A_HALFS;
    const half_t* b = Bbuf + slot * B_HALFS;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int toff = ky * HW_ + kx;
    // all fragment reads of the tile first (both k-steps), then the MFMAs: one LDS latency per tile instead of two
    half8 af[2][FM], bf[2][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int hr = hr0[i] + toff;
      const half_t* row = a + hr * BK;
      const int sw = hr & 7;
      af[0][i] = *reinterpret_cast<const half8*>(row + ((fq ^ sw) << 3));
      af[1][i] = *reinterpret_cast<const half8*>(row + (((4 + fq) ^ sw) << 3));
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int r = wn * TN + j * 16 + fr;
      const half_t* row = b + r * BK;
      bf[0][j] = *reinterpret_cast<const half8*>(row + ((fq ^ (r & 7)) << 3));
      bf[1][j] = *reinterpret_cast<const half8*>(row + (((4 + fq) ^ (r & 7)) << 3));
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
    CPROBE(4)
    if (++slot == NSB) slot = 0;
    if (++tap == 9) { tap = 0; ++cb; }
    if (++tap2 == 9) { tap2 = 0; ++cb2; }
  }
#undef HALO_ISSUE_A
#undef HALO_ISSUE_A_PIECE
#undef HALO_ISSUE_B
  __syncthreads();  // every wave is done reading the tiles before the epilogue reuses the LDS
  WGTL_LOOP()

  // ---- epilogue.  Residual and bias + time vector are loaded before the accumulator transpose (and any store).
  constexpr int CH = BN / 8;
  constexpr int NIT = BM * CH / NT;
  half8 rpre[NIT];
  float brv[8], brv2[8];  // x = (acc + brv) + brv2: bias + time vector summed first (one launch) | bias, then the time vector
  const int pre_n = n0 + (tid % CH) * 8;  // (the in-launch split-K reduction: the reducer kernel's order of additions, same bits)
  const bool ncol_ok = pre_n + 8 <= p.N;
#pragma unroll
  for (int i = 0; i < 8; ++i) brv[i] = brv2[i] = 0.f;
  if (p.split_k == 1) {
    if (ncol_ok && p.bias) {
      half8 v = *reinterpret_cast<const half8*>(p.bias + pre_n);
#pragma unroll
      for (int i = 0; i < 8; ++i) brv[i] += (float)v[i];
    }
    if (ncol_ok && p.rowvec) {
      half8 v = *reinterpret_cast<const half8*>(p.rowvec + pre_n);
#pragma unroll
      for (int i = 0; i < 8; ++i) brv[i] += (float)v[i];
    }
    if (p.residual) {
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const int q = tid + j * NT;
        const int r = q / CH;
        rpre[j] = *reinterpret_cast<const half8*>(p.residual + (ncol_ok && row_ok(r) ? (size_t)row_m(r) * p.ldr + pre_n : 0));
      }
    }
  }
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = wn * TN + j * 16 + fr;
      const int row = wm * TM + i * 16 + fq * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) Cs[(row + r) * BNP + col] = acc[i][j][r];
    }
  __syncthreads();
  CPROBE(5)
  bool from_slabs = false;
  if (p.split_k > 1) {  // fp32 slab of this split
    float* slab = p.ws_partial + (size_t)split * p.M * p.N;
    // in-launch reduction (p.counters): write-through stores, so that the reducing workgroup sees them from any XCD
    // without an L2 write-back on the producers (as in conv_gemm_kernel)
    const bool wt = p.counters != nullptr;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        slab, 0, (int)min((size_t)p.M * p.N * sizeof(float), (size_t)0x7fffffff), 0x00020000);
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int q = tid + j * NT;
      const int r = q / CH, c8 = (q - r * CH) * 8;
      const int n = n0 + c8;
      if (n + 8 <= p.N && row_ok(r)) {
        const float* sp = Cs + r * BNP + c8;
        if (wt) {
          const int off = (int)(((size_t)row_m(r) * p.N + n) * sizeof(float));
          __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(sp), rsrc, off, 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(sp + 4), rsrc, off + 16, 0, 16);
        } else {
          float* d = slab + (size_t)row_m(r) * p.N + n;
          *reinterpret_cast<f32x4*>(d) = *reinterpret_cast<const f32x4*>(sp);
          *reinterpret_cast<f32x4*>(d + 4) = *reinterpret_cast<const f32x4*>(sp + 4);
        }
      }
    }
    if (!p.counters) { WGTL_END(1) return; }  // two-kernel form: splitk_reduce_kernel applies the epilogue
    // The LAST workgroup to arrive at this (patch, column tile) sums the slabs in the order 0..split_k-1 (the reducer
    // kernel's order: same bits) and runs the epilogue.  Ticket protocol as in conv_gemm_kernel.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* lastflag = reinterpret_cast<int*>(smem);
    if (tid == 0) {
      int* cnt = p.counters + tile_n * p.tiles_m + tile_m;
      const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == p.split_k - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      *lastflag = last;
    }
    __syncthreads();
    if (!*lastflag) { WGTL_END(1) return; }
    from_slabs = true;
    if (ncol_ok && p.bias) {
      half8 v = *reinterpret_cast<const half8*>(p.bias + pre_n);
#pragma unroll
      for (int i = 0; i < 8; ++i) brv[i] += (float)v[i];
    }
    if (ncol_ok && p.rowvec) {
      half8 v = *reinterpret_cast<const half8*>(p.rowvec + pre_n);
#pragma unroll
      for (int i = 0; i < 8; ++i) brv2[i] = (float)v[i];
    }
    if (p.residual) {
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const int q = tid + j * NT;
        const int r = q / CH;
        rpre[j] = *reinterpret_cast<const half8*>(p.residual + (ncol_ok && row_ok(r) ? (size_t)row_m(r) * p.ldr + pre_n : 0));
      }
    }
  }
  const int act = p.act & 0xff;
  const bool post = (p.act & VSD_ACT_POST) != 0;
  auto finish = [&](auto act_tag) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;  // 0 none, 1 relu, 2 silu, 3 relu after the residual
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int q = tid + j * NT;
      const int r = q / CH, c8 = (q - r * CH) * 8;
      const int n = n0 + c8;
      if (n + 8 <= p.N && row_ok(r)) {
        f32x4 lo, hi;
        if (from_slabs) {
          lo = hi = (f32x4){0.f, 0.f, 0.f, 0.f};
          const float* sp = p.ws_partial + (size_t)row_m(r) * p.N + n;
          const size_t slab_sz = (size_t)p.M * p.N;
          // eight slabs per round trip (see load_chunk8 in conv_kernels.h); the additions run in slab order: same bits
          constexpr int NB = 8;
          for (int k0 = 0; k0 < p.split_k; k0 += NB) {
            f32x4 a[NB], b[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
              const int kk = k0 + k < p.split_k ? k0 + k : 0;
              a[k] = *reinterpret_cast<const f32x4*>(sp + kk * slab_sz);
              b[k] = *reinterpret_cast<const f32x4*>(sp + kk * slab_sz + 4);
            }
#pragma unroll
            for (int k = 0; k < NB; ++k)
              if (k0 + k < p.split_k) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  lo[i] += a[k][i];
                  hi[i] += b[k][i];
                }
              }
          }
        } else {
          lo = *reinterpret_cast<const f32x4*>(Cs + r * BNP + c8);
          hi = *reinterpret_cast<const f32x4*>(Cs + r * BNP + c8 + 4);
        }
        const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        half8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float x = (v[i] + brv[i]) + brv2[i];
          if (ACT == 1) x = fmaxf(x, 0.f);
          if (ACT == 2) x = silu_f(x);
          if (p.residual) x += (float)rpre[j][i];
          if (ACT == 3) x = fmaxf(x, 0.f);
          o[i] = (half_t)x;
        }
        VSD_OUT_STORE8(p.out + (size_t)row_m(r) * p.ldo + n, o);
      }
    }
  };
  if (act == VSD_ACT_RELU && post) finish(std::integral_constant<int, 3>{});
  else if (act == VSD_ACT_RELU) finish(std::integral_constant<int, 1>{});
  else if (act == VSD_ACT_SILU) finish(std::integral_constant<int, 2>{});
  else finish(std::integral_constant<int, 0>{});
  CPROBE(6)
  CPROBE_OUT()
  WGTL_END(1)
}

}  // namespace

void vsd_launch_conv_halo(const ConvParams& p, int BM, int BN, int grid, hipStream_t s) {
  // (a 4-slot weight ring measured the same as 3 slots: the wait per tile is fill throughput, not lead)
  if (BM == 256 && BN == 128) hipLaunchKernelGGL((conv_halo_kernel<128, 4, 3>), dim3(grid), dim3(512), 0, s, p);
  else if (BM == 256) hipLaunchKernelGGL((conv_halo_kernel<64, 4, 3>), dim3(grid), dim3(512), 0, s, p);
  else if (BN == 128) hipLaunchKernelGGL((conv_halo_kernel<128, 2, 3>), dim3(grid), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((conv_halo_kernel<64, 2, 3>), dim3(grid), dim3(256), 0, s, p);
}

void vsd_launch_splitk_reduce(const ConvParams& p, int grid, hipStream_t s) {
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, p);
}
void vsd_launch_splitk_reduce_group(const ConvGroup& g, int grid, hipStream_t s) {
  hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(grid, g.n), dim3(256), 0, s, g);
}

// Persistent stream-K form of the implicit-GEMM conv / linear kernel (pipeline 8; round 4).
//
// Why: the short-K layers of the transformer blocks (K <= 1280: attention projections, proj_in / proj_out, zero-convs,
// shortcuts) ran one output tile per workgroup.  By scripts/wg_timeline.py more than half of such a workgroup's life was
// NOT its K loop (prologue: argument block, index arithmetic, the first tiles' memory round trip; tail: accumulator transpose,
// residual round trip, stores), and their grids are small: 1280x1280x1280 = 200 tiles on 256 CUs, one workgroup per CU, 56 CUs
// idle, every CU pulling its 0.5 MB of operands alone at the ~35 GB/s one CU gets from beyond the L2 -- 14.8 us for 4.2 GFLOP.
//
// Form: the launch's work is the list of (tile, K step) units in tile-major order, U = tiles x KT.  The grid is a fixed number
// of workgroups (<= the resident slots of the chip); workgroup g walks units [g q, (g + 1) q): the tail of one tile, whole
// tiles, the head of another -- so 200 tiles fill all CUs evenly, a workgroup pays the argument / lane setup once, and the
// first K tile of its NEXT segment is already on its way (LDS-DMA into the ring slot the accumulator transpose does not use)
// while the current segment's epilogue runs.  A tile whose K range is shared by several workgroups is finished like the
// in-launch split-K of conv_kernels.h: every part leaves an fp32 slab (write-through) and takes a ticket, the LAST arriver
// adds the slabs in part order -- a fixed order, so the bits do not depend on who arrives when -- and runs the epilogue.
// Nobody waits for anybody (no spinning: two persistent launches on two streams cannot deadlock each other).
//
// Operand path: the buffer-load-to-LDS ("FAST") path of conv_gemm_kernel -- Cin % 64 == 0 per source, no resize, 1x1 or 3x3.
#include "conv_kernels.h"

namespace {

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_streamk_kernel(const ConvParams p) {
  prefetch_kernargs();
  WGTL_START()
  constexpr int WM = 2, WN = 2;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / 16, FN = TN / 16;
  constexpr int AR = BM / 32, BR = BN / 32;
  constexpr int BNP = BN + 4;
  constexpr int STAGE_HALFS = (BM + BN) * BK;
  constexpr int STAGE_BYTES = STAGE_HALFS * 2;
  constexpr int EPI_BYTES = BM * BNP * 4;
  // ring slots: the accumulator tile of the epilogue lies in slots [0, NSLOT - 1); the LAST slot is the one a segment's first K
  // tile goes to, so that it can be fetched while the previous segment's epilogue still uses the others
  constexpr int NSLOT = EPI_BYTES <= 2 * STAGE_BYTES ? 3 : 4;
  static_assert(EPI_BYTES <= (NSLOT - 1) * STAGE_BYTES, "accumulator tile must leave the last ring slot free");
  constexpr int LDS_BYTES = NSLOT * STAGE_BYTES;
  constexpr int LPT = AR + BR;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES + BM * 8];
  float* rowms = reinterpret_cast<float*>(smem + LDS_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int cc = tid & 7, lr = tid >> 3;
  const int lc = cc ^ (lr & 7);
  const int fr = lane & 15, fq = lane >> 4;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  constexpr int OOB = (int)0x80000000;

  const int KT = p.Kp / BK;
  const int u_end = min(p.sk_units, (int)(blockIdx.x + 1) * p.sk_q);
  int u = blockIdx.x * p.sk_q;
  if (u >= u_end) return;

  const int neg_pix = p.pad * p.ws + p.pad;
  const half_t* abase0 = p.src0 - (size_t)neg_pix * p.c0;
  const half_t* abase1 = (p.src1 ? p.src1 : p.src0) - (size_t)neg_pix * p.c1;
  const int anr0 = (int)(((size_t)p.batch * p.img_in + neg_pix) * p.c0 * 2);
  const int anr1 = (int)(((size_t)p.batch * p.img_in + neg_pix) * p.c1 * 2);
  const int bnr = (int)((size_t)p.N * p.Kp * 2);

  // ---- per-segment state (a segment = this workgroup's K steps [kt_begin, kt_begin + nt) of one tile)
  int tile_m = 0, tile_n = 0, kt_begin = 0, nt = 0, part = 0, nparts = 1;
  int apix[AR];
  unsigned tapmask[AR];
  int bvoff[BR];
  int cur_c = 0, cur_tap = 0, cur_ky = 0, cur_kx = 0, cur_kt = 0;

// unit U_ -> tile, K range, part bookkeeping, per-row operand offsets, cursor
#define SK_SETUP(U_)                                                                                         \
  {                                                                                                          \
    const int u_ = (U_);                                                                                     \
    const int t_ = fdiv(u_, p.fd_kt);                                                                        \
    kt_begin = u_ - t_ * KT;                                                                                 \
    nt = min(KT - kt_begin, u_end - u_);                                                                     \
    int grp_;                                                                                                \
    block_to_tile(p, t_, tile_m, grp_);                                                                      \
    tile_n = grp_;                                                                                           \
    /* the workgroups that share this tile: first = owner of unit t*KT, last = owner of unit (t+1)*KT - 1 */ \
    const int g_first_ = fdiv(t_ * KT, p.fd_q), g_last_ = fdiv(t_ * KT + KT - 1, p.fd_q);                    \
    nparts = g_last_ - g_first_ + 1;                                                                         \
    part = (int)blockIdx.x - g_first_;                                                                       \
    const int m0_ = tile_m * BM, n0_ = tile_n * BN;                                                          \
    _Pragma("unroll") for (int i = 0; i < AR; ++i) {                                                         \
      const int m = m0_ + lr + 32 * i;                                                                       \
      const bool mv = m < p.M;                                                                               \
      if (p.pointwise) {                                                                                     \
        apix[i] = mv ? m : 0;                                                                                \
        tapmask[i] = mv ? 1u : 0u;                                                                           \
      } else {                                                                                               \
        int mm = mv ? m : 0, b = 0;                                                                          \
        if (p.batch > 1) {                                                                                   \
          b = fdiv(mm, p.fd_hw_out);                                                                         \
          mm -= b * p.hw_out;                                                                                \
        }                                                                                                    \
        const int oy = fdiv(mm, p.fd_wo), ox = mm - oy * p.wo;                                               \
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;                                 \
        apix[i] = b * p.img_in + (iy0 + p.pad) * p.ws + (ix0 + p.pad);                                       \
        unsigned mk = 0;                                                                                     \
        /* (ksize is 1 or 3: compile-time loop bounds keep the row loop unrolled and apix / tapmask in registers) */ \
        _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)                                                     \
          _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                                 \
            const bool in = mv && ky < p.ksize && kx < p.ksize && (unsigned)(iy0 + ky) < (unsigned)p.hi &&   \
                            (unsigned)(ix0 + kx) < (unsigned)p.wi;                                           \
            mk |= (in ? 1u : 0u) << (ky * p.ksize + kx);                                                     \
          }                                                                                                  \
        tapmask[i] = mk;                                                                                     \
      }                                                                                                      \
    }                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < BR; ++i) {                                                         \
      const int n = n0_ + lr + 32 * i;                                                                       \
      bvoff[i] = n < p.N ? n * p.Kp * 2 + lc * 16 : OOB;                                                     \
    }                                                                                                        \
    cur_kt = kt_begin;                                                                                       \
    cur_c = cur_tap = cur_ky = cur_kx = 0;                                                                   \
    if (kt_begin > 0) {                                                                                      \
      const int k0 = kt_begin * BK;                                                                          \
      cur_tap = k0 / p.cin;                                                                                  \
      cur_c = k0 - cur_tap * p.cin;                                                                          \
      cur_ky = cur_tap / p.ksize;                                                                            \
      cur_kx = cur_tap - cur_ky * p.ksize;                                                                   \
    }                                                                                                        \
  }

// fetch the cursor's K tile into ring slot SLOT_ and move the cursor on (conv_gemm_kernel's VSD_ISSUE_FAST)
#define SK_ISSUE(SLOT_)                                                                                      \
  {                                                                                                          \
    half_t* a_ = reinterpret_cast<half_t*>(smem) + (SLOT_) * STAGE_HALFS;                                    \
    half_t* b_ = a_ + BM * BK;                                                                               \
    const int soff_b_ = cur_kt * (BK * 2);                                                                   \
    const __amdgpu_buffer_rsrc_t rsb_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, bnr, 0x00020000);   \
    _Pragma("unroll") for (int i = 0; i < BR; ++i) {                                                         \
      const int bv_ = bvoff[i] + 0; /* (a local copy: hipcc drops the kernel's host stub when the builtin reads the array itself) */ \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb_, (lds_ptr_t)(b_ + (8 * wave_s + 32 * i) * BK), 16, bv_, soff_b_, 0, 0); \
    }                                                                                                        \
    const bool second_ = cur_c >= p.c0;                                                                      \
    const int cs2_ = (second_ ? p.c1 : p.c0) * 2;                                                            \
    const int soff_a_ = (cur_ky * p.ws + cur_kx) * cs2_ + (second_ ? cur_c - p.c0 : cur_c) * 2;              \
    const __amdgpu_buffer_rsrc_t rs_ =                                                                       \
        __builtin_amdgcn_make_buffer_rsrc((void*)(second_ ? abase1 : abase0), 0, second_ ? anr1 : anr0, 0x00020000); \
    const unsigned bit_ = 1u << cur_tap;                                                                     \
    _Pragma("unroll") for (int i = 0; i < AR; ++i) {                                                         \
      const int vo_ = (tapmask[i] & bit_) ? __mul24(apix[i], cs2_) + lc * 16 : OOB;                          \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(a_ + (8 * wave_s + 32 * i) * BK), 16, vo_, soff_a_, 0, 0); \
    }                                                                                                        \
    ++cur_kt;                                                                                                \
    cur_c += BK;                                                                                             \
    if (cur_c >= p.cin) {                                                                                    \
      cur_c = 0;                                                                                             \
      ++cur_tap;                                                                                             \
      if (++cur_kx == p.ksize) {                                                                             \
        cur_kx = 0;                                                                                          \
        ++cur_ky;                                                                                            \
      }                                                                                                      \
    }                                                                                                        \
  }

  SK_SETUP(u)
  SK_ISSUE(NSLOT - 1)
  WGTL_MARK(c)
  while (true) {
    const int m0 = tile_m * BM;
    // ---- this segment: K tile t sits in slot (NSLOT - 1 + t) mod NSLOT; tile 0 is already on its way
    if (nt > 1) SK_ISSUE(0)
    if (p.ln_part && tid < BM) {  // fused input LayerNorm: per-row (mean, rstd) of this tile's rows
      float mean = 0.f, rstd = 0.f;
      if (m0 + tid < p.M) ln_row_stats(p, m0 + tid, mean, rstd);
      rowms[2 * tid] = mean;
      rowms[2 * tid + 1] = rstd;
    }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int slot = NSLOT - 1;
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 2 < nt) {
        int ns = slot + 2;
        if (ns >= NSLOT) ns -= NSLOT;
        SK_ISSUE(ns)
      }
      const half_t* a = reinterpret_cast<const half_t*>(smem) + slot * STAGE_HALFS;
      const half_t* b = a + BM * BK;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        half8 af[FM], bf[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int r = wm * TM + i * 16 + fr;
          af[i] = *reinterpret_cast<const half8*>(a + r * BK + (((ks * 4 + fq) ^ (r & 7)) << 3));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int r = wn * TN + j * 16 + fr;
          bf[j] = *reinterpret_cast<const half8*>(b + r * BK + (((ks * 4 + fq) ^ (r & 7)) << 3));
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      if (++slot == NSLOT) slot = 0;
    }
    __syncthreads();  // every wave is done with the ring before the epilogue's accumulator tile (slots 0 .. NSLOT - 2) is written
    WGTL_LOOP()
    // ---- the next segment's first K tile goes out now (into the slot the epilogue leaves alone), then this one's epilogue
    const int e_tile_m = tile_m, e_tile_n = tile_n, e_part = part, e_nparts = nparts;
    (void)lane;
    u += nt;
    const bool more = u < u_end;
    if (more) {
      SK_SETUP(u)
      SK_ISSUE(NSLOT - 1)
    }
    {
      // (the kernel-level tile_m / tile_n already belong to the NEXT segment: the epilogue text sees this segment's through
      //  these shadowing names)
      const int tile_m = e_tile_m, tile_n = e_tile_n;
      const int m0 = tile_m * BM, n0 = tile_n * BN;
#define EPI_PART e_part
#define EPI_NPARTS e_nparts
#define EPI_EXIT goto epilogue_done;  /* (timeline builds: a workgroup's LAST segment is the one logged) */
#include "conv_epilogue.inc"
#undef EPI_PART
#undef EPI_NPARTS
#undef EPI_EXIT
    }
  epilogue_done:
    if (!more) break;
    __syncthreads();  // the accumulator tile has been read: its slots may take the next segment's K tiles
  }
  WGTL_END(2)
#undef SK_SETUP
#undef SK_ISSUE
}

template <int BM, int BN>
void launch_sk(const ConvParams& p, int grid, hipStream_t s) {
  hipLaunchKernelGGL((conv_streamk_kernel<BM, BN>), dim3(grid), dim3(256), 0, s, p);
}

}  // namespace

int vsd_streamk_lds_bytes(int bm, int bn) {
  const int stage = (bm + bn) * BK * 2, epi = bm * (bn + 4) * 4;
  return (epi <= 2 * stage ? 3 : 4) * stage + bm * 8;
}

void vsd_launch_conv_streamk(const ConvParams& p, int bm, int bn, int grid, hipStream_t s) {
  if (bm == 128 && bn == 128) launch_sk<128, 128>(p, grid, s);
  else if (bm == 128) launch_sk<128, 64>(p, grid, s);
  else if (bn == 128) launch_sk<64, 128>(p, grid, s);
  else launch_sk<64, 64>(p, grid, s);
}

// How fast can a CU take GEMM operands, and does it matter which way they come?  (development probe, round 4)
// The conv / linear kernels are bound by the L2 -> LDS fill.  This probe runs the OPERAND TRAFFIC of a
// 64 x 128-tile GEMM without the MFMAs, for a given (M, N, K), in three forms:
//   lds   both operands through the LDS-DMA ring as conv_gemm_kernel does (3 slots, two K tiles in flight): 24 KB per K step
//   mix   A (8 KB per K step) through the ring, B straight into registers from FRAGMENT-MAJOR weights (every wave loads the
//         64 columns x 64 k it needs: 8 contiguous 1 KB pieces; the two waves of a column half load the same bytes), D steps ahead
//   mixh  as mix, but a wave loads only HALF of its column half's B bytes (what a form that shares B between the two waves of
//         a column half through a wave-pair exchange would load): the no-redundancy bound
// and prints us per launch and operand TB/s (useful bytes: 24 KB per workgroup and K step).
//   hipcc -O3 --offload-arch=gfx950 scripts/fill_probe.cpp -o scripts/fill_probe.bin ; ./scripts/fill_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int BM = 64, BN = 128, BK = 64;
constexpr int STAGE = (BM + BN) * BK * 2;  // bytes

struct P {
  const char* a;   // [M][K] fp16
  const char* w;   // [N][K] fp16 (lds form) or fragment-major (mix forms): per 128-col tile and K step a contiguous 16 KB block
  int M, N, K, tiles_m, tiles_n;
  unsigned* sink;
};

template <int MODE, int DEPTH>  // MODE 0 lds, 1 mix, 2 mixh
__global__ __launch_bounds__(256) void fill_kernel(const P p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * STAGE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int tile = blockIdx.x;
  const int tile_n = tile % p.tiles_n, tile_m = tile / p.tiles_n;
  const int KT = p.K / BK;
  const int lr = tid >> 3, cc = tid & 7;
  const int a_nr = p.M * p.K * 2, w_nr = p.N * p.K * 2;
  unsigned acc = 0;
  // per-lane byte offsets of the rows this thread fetches (A: rows lr, lr+32; W: rows lr + 32 i)
  int aoff[2], woff[4];
  for (int i = 0; i < 2; ++i) aoff[i] = ((tile_m * BM + lr + 32 * i) * p.K) * 2 + cc * 16;
  for (int i = 0; i < 4; ++i) woff[i] = ((tile_n * BN + lr + 32 * i) * p.K) * 2 + cc * 16;
  auto issue = [&](int kt, int slot) __attribute__((always_inline)) {
    unsigned char* a_ = smem + slot * STAGE;
    unsigned char* b_ = a_ + BM * BK * 2;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, a_nr, 0x00020000);
    for (int i = 0; i < 2; ++i) {
      const int vo = aoff[i] + 0;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(a_ + (8 * wave_s + 32 * i) * BK * 2), 16, vo, kt * BK * 2, 0, 0);
    }
    if (MODE == 0) {
      const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, w_nr, 0x00020000);
      for (int i = 0; i < 4; ++i) {
        const int vo = woff[i] + 0;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(b_ + (8 * wave_s + 32 * i) * BK * 2), 16, vo, kt * BK * 2, 0, 0);
      }
    }
  };
  constexpr int LPT = MODE == 0 ? 6 : 2;
  // B in registers: fragment-major block of (tile_n, kt): 16 KB = [wn (2)][piece (8)][lane (64)][16 B]; wave (wm, wn) loads column half wn
  constexpr int NB = MODE == 0 ? 1 : (MODE == 1 ? 8 : 4);
  u32x4 breg[DEPTH][NB];
  const int wn = wave & 1, wm = wave >> 1;
  auto bload = [&](int kt, int d) __attribute__((always_inline)) {
    if (MODE == 0) return;
    const char* blk = p.w + ((size_t)tile_n * KT + kt) * (BN * BK * 2) + wn * 8192 + lane * 16;
    // (inline assembly: the compiler's wait-count pass does not see these loads, so it cannot put its own -- conservative:
    //  it merges the prologue's state into the loop -- waits in front of their first use; the counted s_waitcnt below names the
    //  registers it protects as in / out operands, which keeps their uses behind it)
    for (int j = 0; j < NB; ++j) {
      const int piece = MODE == 1 ? j : (wm * 4 + j);
      const char* ptr = blk + piece * 1024;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(breg[d][j]) : "v"(ptr) : "memory");
    }
  };
#define BWAIT8(D_, N_) asm volatile("s_waitcnt vmcnt(%8)" : "+v"(breg[D_][0]), "+v"(breg[D_][1]), "+v"(breg[D_][2]), "+v"(breg[D_][3]), \
    "+v"(breg[D_][4]), "+v"(breg[D_][5]), "+v"(breg[D_][6]), "+v"(breg[D_][7]) : "n"(N_) : "memory")
#define BWAIT4(D_, N_) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(breg[D_][0]), "+v"(breg[D_][1]), "+v"(breg[D_][2]), "+v"(breg[D_][3]) : "n"(N_) : "memory")
  // prologue in the steady state's issue order (step s issues A(s + 2), then B(s + DEPTH)): B(0 .. DEPTH-3) | A0 B(DEPTH-2) | A1 B(DEPTH-1)
  if (MODE == 0) {
    for (int s = 0; s < 2 && s < KT; ++s) issue(s, s);
  } else {
#pragma unroll
    for (int d = 0; d + 2 < DEPTH; ++d)
      if (d < KT) bload(d, d);
    issue(0, 0);
    if (DEPTH - 2 < KT) bload(DEPTH - 2, DEPTH - 2);
    if (1 < KT) issue(1, 1);
    if (DEPTH - 1 < KT) bload(DEPTH - 1, DEPTH - 1);
  }
  int slot = 0;
  // (the K loop is unrolled DEPTH-fold so that the register batch a step consumes is a compile-time choice: with a run-time
  //  index the compiler must wait for EVERY outstanding batch before the first use)
  for (int t0 = 0; t0 < KT; t0 += DEPTH) {
#pragma unroll
    for (int dd = 0; dd < DEPTH; ++dd) {
      const int t = t0 + dd;
      if (t >= KT) break;
      // everything older than the two youngest issue groups has landed: A(t) and B(t)
      if (MODE == 0) {
        if (t + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        constexpr int NW = DEPTH == 2 ? NB + 2 : 2 * NB + 2;
        if constexpr (NB == 8) {
          if (t + DEPTH < KT) BWAIT8(dd, NW);
          else BWAIT8(dd, 0);
        } else if constexpr (NB == 4) {
          if (t + DEPTH < KT) BWAIT4(dd, NW);
          else BWAIT4(dd, 0);
        }
      }
      __builtin_amdgcn_s_barrier();
      if (t + 2 < KT) issue(t + 2, (slot + 2) % 3);
      // consume: A from LDS (every wave reads its 32 rows' fragments), B from LDS or registers
      const unsigned char* a = smem + slot * STAGE;
      u32x4 v = *reinterpret_cast<const u32x4*>(a + ((wm * 32 + (lane & 31)) * BK * 2) + ((lane >> 5) * 16));
      acc ^= v[0] ^ v[3];
      if (MODE == 0) {
        const unsigned char* b = a + BM * BK * 2;
        u32x4 u = *reinterpret_cast<const u32x4*>(b + ((wn * 64 + lane) * BK * 2));
        acc ^= u[1];
      } else {
#pragma unroll
        for (int j = 0; j < NB; ++j) acc ^= breg[dd][j][0] ^ breg[dd][j][2];
        if (t + DEPTH < KT) bload(t + DEPTH, dd);
      }
      if (++slot == 3) slot = 0;
    }
  }
  if (acc == 0x12345678u) p.sink[0] = acc;
}

template <int MODE, int DEPTH>
static double run(const P& p, int grid, hipStream_t s) {
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((fill_kernel<MODE, DEPTH>), dim3(grid), dim3(256), 0, s, p);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((fill_kernel<MODE, DEPTH>), dim3(grid), dim3(256), 0, s, p);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms / 20 * 1e3 < best) best = ms / 20 * 1e3;
  }
  return best;
}

int main() {
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const int shapes[][3] = {{1280, 1280, 1280}, {5120, 640, 640}, {5120, 1920, 640}, {20480, 320, 320}, {5120, 5120, 640}, {1280, 1280, 5120}};
  for (auto& sh : shapes) {
    P p;
    p.M = sh[0]; p.N = (sh[1] + 127) / 128 * 128; p.K = sh[2];
    p.tiles_m = p.M / BM; p.tiles_n = p.N / BN;
    char *a, *w;
    unsigned* sink;
    CK(hipMalloc(&a, (size_t)p.M * p.K * 2));
    CK(hipMalloc(&w, (size_t)p.N * p.K * 2));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, (size_t)p.M * p.K * 2));
    CK(hipMemset(w, 2, (size_t)p.N * p.K * 2));
    p.a = a; p.w = w; p.sink = sink;
    const int grid = p.tiles_m * p.tiles_n;
    const double useful = (double)grid * (p.K / BK) * STAGE;
    const double t0 = run<0, 1>(p, grid, s), t1 = run<1, 2>(p, grid, s), t2 = run<1, 3>(p, grid, s), t3 = run<2, 3>(p, grid, s), t4 = run<2, 4>(p, grid, s);
    printf("M=%5d N=%5d K=%5d  %4d workgroups x %3d K steps, operands %.0f MB:  lds %.1f us (%.1f TB/s) | mix d2 %.1f us (%.1f) | mix d3 %.1f us (%.1f) | mixh d3 %.1f us (%.1f) | mixh d4 %.1f us (%.1f)\n",
           p.M, p.N, p.K, grid, p.K / BK, useful / 1e6, t0, useful / t0 / 1e6, t1, useful / t1 / 1e6, t2, useful / t2 / 1e6, t3, useful / t3 / 1e6, t4, useful / t4 / 1e6);
    CK(hipFree(a)); CK(hipFree(w)); CK(hipFree(sink));
  }
  return 0;
}

// Does a deeper LDS-DMA ring move the operand fill of the small-grid short-K layers?  The fill of conv_gemm_kernel's 64 x 128 tile
// (24 KB per K step: both operands by raw buffer loads to LDS, counted s_waitcnt, one barrier per step, one fragment read per
// wave standing in for the MFMAs) with NS ring slots = NS - 1 K steps in flight, one workgroup per CU, on the round's small grids.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ring_depth_probe.cpp -o scripts/ring_depth_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int BM = 64, BN = 128, BK = 64, STAGE = (BM + BN) * BK * 2;
struct P { const char* a; const char* w; int M, N, K, tiles_n; unsigned* sink; };

template <int NS>
__global__ __launch_bounds__(256) void fill_kernel(const P p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
  const int KT = p.K / BK, lr = tid >> 3, cc = tid & 7, wm = wave >> 1, wn = wave & 1;
  unsigned acc = 0;
  int aoff[2], woff[4];
  for (int i = 0; i < 2; ++i) aoff[i] = ((tile_m * BM + lr + 32 * i) * p.K) * 2 + cc * 16;
  for (int i = 0; i < 4; ++i) woff[i] = ((tile_n * BN + lr + 32 * i) * p.K) * 2 + cc * 16;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.M * p.K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.N * p.K * 2, 0x00020000);
  auto issue = [&](int kt, int slot) __attribute__((always_inline)) {
    unsigned char* a_ = smem + slot * STAGE;
    unsigned char* b_ = a_ + BM * BK * 2;
    for (int i = 0; i < 2; ++i) {
      const int vo = aoff[i] + 0;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(a_ + (8 * wave_s + 32 * i) * BK * 2), 16, vo, kt * BK * 2, 0, 0);
    }
    for (int i = 0; i < 4; ++i) {
      const int vo = woff[i] + 0;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(b_ + (8 * wave_s + 32 * i) * BK * 2), 16, vo, kt * BK * 2, 0, 0);
    }
  };
  constexpr int AHEAD = NS - 1;  // K steps in flight
  for (int s = 0; s < AHEAD && s < KT; ++s) issue(s, s);
  int slot = 0;
  for (int t = 0; t < KT; ++t) {
    // stage t has landed when at most (stages t+1 .. t+AHEAD-1 that exist) x 6 loads are outstanding
    const int left = KT - 1 - t;
    const int out = left < AHEAD - 1 ? left : AHEAD - 1;
    switch (out) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    }
    __builtin_amdgcn_s_barrier();
    if (t + AHEAD < KT) issue(t + AHEAD, (slot + AHEAD) % NS);
    const unsigned char* a = smem + slot * STAGE;
    u32x4 v = *reinterpret_cast<const u32x4*>(a + ((wm * 32 + (lane & 31)) * BK * 2) + ((lane >> 5) * 16));
    u32x4 u = *reinterpret_cast<const u32x4*>(a + BM * BK * 2 + ((wn * 64 + lane) * BK * 2));
    acc ^= v[0] ^ v[3] ^ u[1];
    if (++slot == NS) slot = 0;
  }
  if (acc == 0x12345678u) p.sink[0] = acc;
}

template <int NS>
static double run(const P& p, int grid, hipStream_t s, char* flush, size_t flush_bytes) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    double tot = 0;
    for (int i = 0; i < 10; ++i) {
      if (flush) CK(hipMemsetAsync(flush, i, flush_bytes, s));  // operands cold (as in the frame): push them out of L2 / MALL
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL((fill_kernel<NS>), dim3(grid), dim3(256), 0, s, p);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      tot += ms;
    }
    if (rep && tot / 10 * 1e3 < best) best = tot / 10 * 1e3;
  }
  return best;
}

int main() {
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  char* flush;
  const size_t flush_bytes = 768u << 20;
  CK(hipMalloc(&flush, flush_bytes));
  const int shapes[][3] = {{1280, 1280, 1280}, {5120, 640, 640}, {320, 1280, 1280}, {1280, 1280, 5120}, {5120, 1920, 640}};
  for (int cold = 0; cold < 2; ++cold)
    for (auto& sh : shapes) {
      P p;
      p.M = sh[0]; p.N = sh[1]; p.K = sh[2];
      p.tiles_n = p.N / BN;
      char *a, *w;
      unsigned* sink;
      CK(hipMalloc(&a, (size_t)p.M * p.K * 2));
      CK(hipMalloc(&w, (size_t)p.N * p.K * 2));
      CK(hipMalloc(&sink, 64));
      CK(hipMemset(a, 1, (size_t)p.M * p.K * 2));
      CK(hipMemset(w, 2, (size_t)p.N * p.K * 2));
      p.a = a; p.w = w; p.sink = sink;
      const int grid = (p.M / BM) * p.tiles_n;
      const double mb = (double)grid * (p.K / BK) * STAGE / 1e6;
      char* f = cold ? flush : nullptr;
      const double t3 = run<3>(p, grid, s, f, flush_bytes), t4 = run<4>(p, grid, s, f, flush_bytes), t6 = run<6>(p, grid, s, f, flush_bytes);
      printf("%s M=%5d N=%5d K=%5d %4d workgroups x %3d K steps, %4.0f MB through LDS: 3 slots %5.1f us | 4 slots %5.1f | 6 slots %5.1f   (events around single launches: + ~5 us each)\n",
             cold ? "cold" : "warm", p.M, p.N, p.K, grid, p.K / BK, mb, t3, t4, t6);
      CK(hipFree(a)); CK(hipFree(w)); CK(hipFree(sink));
    }
  return 0;
}

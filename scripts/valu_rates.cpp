// Issue rates of the instructions in the attention softmax on gfx950, and whether they overlap with MFMAs:
//   hipcc -O3 --offload-arch=gfx950 scripts/valu_rates.cpp -o /tmp/valu_rates && /tmp/valu_rates
// One workgroup on one CU.  256 threads = one wave per SIMD; 512 = two per SIMD (waves w and w+4 share a SIMD).
// Every test runs N instructions per wave in 8 independent dependency chains and reports shader cycles per instruction
// (per wave, and per SIMD when two waves run beside each other).
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum { T_EXP, T_SUB, T_MAX3, T_PKMUL, T_CVTPK, T_MFMA, T_MFMA_EXP_SAMEWAVE, T_FMA, T_MIX, T_MIXVALU, T_MIX_BURST, T_N };

template <int WHAT>
__device__ __forceinline__ void body(float (&v)[8], f32x16 (&acc)[2], half8 a, half8 b, int iters) {
  for (int it = 0; it < iters; ++it) {
    if (WHAT == T_EXP) {
#define X(i) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(v[i]));
      REP8(X) REP8(X)
#undef X
    } else if (WHAT == T_SUB) {
#define X(i) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
      REP8(X) REP8(X)
#undef X
    } else if (WHAT == T_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
      REP8(X) REP8(X)
#undef X
    } else if (WHAT == T_MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "v"(v[(i + 2) & 7]));
      REP8(X) REP8(X)
#undef X
    } else if (WHAT == T_PKMUL) {
      // (two packed multiplies over the four register pairs)
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[0])) : "v"(*reinterpret_cast<double*>(&v[2])));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[2])) : "v"(*reinterpret_cast<double*>(&v[4])));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[4])) : "v"(*reinterpret_cast<double*>(&v[6])));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[6])) : "v"(*reinterpret_cast<double*>(&v[0])));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[0])) : "v"(*reinterpret_cast<double*>(&v[2])));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[2])) : "v"(*reinterpret_cast<double*>(&v[4])));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[4])) : "v"(*reinterpret_cast<double*>(&v[6])));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[6])) : "v"(*reinterpret_cast<double*>(&v[0])));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[0])) : "v"(*reinterpret_cast<double*>(&v[2])));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[2])) : "v"(*reinterpret_cast<double*>(&v[4])));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[4])) : "v"(*reinterpret_cast<double*>(&v[6])));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[6])) : "v"(*reinterpret_cast<double*>(&v[0])));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[0])) : "v"(*reinterpret_cast<double*>(&v[2])));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[2])) : "v"(*reinterpret_cast<double*>(&v[4])));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[4])) : "v"(*reinterpret_cast<double*>(&v[6])));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[6])) : "v"(*reinterpret_cast<double*>(&v[0])));
    } else if (WHAT == T_CVTPK) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
      REP8(X) REP8(X)
#undef X
    } else if (WHAT == T_MFMA) {
      // 16 MFMAs in two chains
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[1], 0, 0, 0);
      }
    } else if (WHAT == T_MIX || WHAT == T_MIXVALU || WHAT == T_MIX_BURST) {
      // the attention tile's mix per MFMA: 7 VALU (2 exp, 2 sub, 1 max3, 1 cvt_pk, 1 fma).  T_MIX: one MFMA, then its 7
      // VALU, 16 times; T_MIX_BURST: the 16 MFMAs back to back, then the 112 VALU; T_MIXVALU: the VALU alone.
#define VALU7(k)                                                                                         \
      asm volatile("v_exp_f32_e32 %0, %0" : "+v"(v[(k) & 7]));                                           \
      asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(v[(k + 1) & 7]) : "v"(v[(k + 5) & 7]));             \
      asm volatile("v_exp_f32_e32 %0, %0" : "+v"(v[(k + 2) & 7]));                                       \
      asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(v[(k + 3) & 7]) : "v"(v[(k + 6) & 7]));             \
      asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[(k + 4) & 7]) : "v"(v[(k + 5) & 7]), "v"(v[(k + 6) & 7])); \
      asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[(k + 5) & 7]) : "v"(v[(k + 7) & 7]));          \
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(k + 6) & 7]) : "v"(v[(k + 7) & 7]));
      if (WHAT == T_MIX_BURST) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[1], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 16; ++k) { VALU7(k) }
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          if (WHAT == T_MIX) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[k & 1]) : "v"(a), "v"(b));
          }
          VALU7(k)
        }
      }
#undef VALU7
    } else if (WHAT == T_MFMA_EXP_SAMEWAVE) {
      // per MFMA (32 cycles of the matrix pipe) two independent exps of the same wave
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
        asm volatile("v_exp_f32_e32 %0, %0" : "+v"(v[(2 * k) & 7]));
        asm volatile("v_exp_f32_e32 %0, %0" : "+v"(v[(2 * k + 1) & 7]));
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[1], 0, 0, 0);
        asm volatile("v_exp_f32_e32 %0, %0" : "+v"(v[(2 * k + 2) & 7]));
        asm volatile("v_exp_f32_e32 %0, %0" : "+v"(v[(2 * k + 3) & 7]));
      }
    }
  }
}

// waves 0..3 run LO, waves 4.. (the second / third wave of each SIMD, when launched) run HI
template <int LO, int HI>
__global__ void k(long long* out, float* sink, int iters) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = -0.001f * (threadIdx.x + i + 1);
  f32x16 acc[2];
  for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.01f; b[i] = (_Float16)0.02f; }
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  if (wave < 4) body<LO>(v, acc, a, b, iters);
  else body<HI>(v, acc, a, b, iters);
  // the wave's own results must be complete before the clock is read
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
  asm volatile("s_nop 0" ::"v"(s));
  long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
  sink[threadIdx.x] = s;
}

template <int LO, int HI>
void run(const char* name, int threads, int per_iter_lo, int per_iter_hi) {
  long long* d; float* sink;
  hipMalloc(&d, 16 * sizeof(long long));
  hipMalloc(&sink, 1024 * sizeof(float));
  const int iters = 512;
  long long h[16] = {0};
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<LO, HI>), dim3(1), dim3(threads), 0, 0, d, sink, iters);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-52s wave0 %8.2f cyc/instr", name, (double)h[0] / (iters * per_iter_lo));
  if (threads > 256) printf("   wave4 %8.2f cyc/instr   (wave0 total %lld, wave4 total %lld)", (double)h[4] / (iters * per_iter_hi), h[0], h[4]);
  printf("\n");
  hipFree(d); hipFree(sink);
}

// The shader clock under load: s_memtime (shader cycles) against s_memrealtime (constant 100 MHz) around a long loop of
// MFMAs (+ the softmax VALU mix) on `grid` workgroups of 512 threads.
template <int WHAT>
__global__ void clk_kernel(long long* out, float* sink, int iters) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = -0.001f * (threadIdx.x + i + 1);
  f32x16 acc[2];
  for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.01f; b[i] = (_Float16)0.02f; }
  long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  body<WHAT>(v, acc, a, b, iters);
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
  asm volatile("s_nop 0" ::"v"(s));
  long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
  if (s == 123.f) sink[threadIdx.x] = s;
}

template <int WHAT>
void clk(const char* name, int grid, int iters) {
  long long* d; float* sink;
  hipMalloc(&d, 16 * sizeof(long long));
  hipMalloc(&sink, 1024 * sizeof(float));
  long long h[2] = {0, 0};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((clk_kernel<WHAT>), dim3(grid), dim3(512), 0, 0, d, sink, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s grid %5d: %.2f ms; workgroup 0: %lld shader cycles in %lld ticks of 100 MHz = %.0f MHz\n", name, grid, ms, h[0], h[1],
         h[1] ? (double)h[0] / ((double)h[1] / 100.0) : 0.0);
  hipFree(d); hipFree(sink);
}

int main() {
  clk<T_MFMA>("clock: mfma only", 1, 20000);
  clk<T_MFMA>("clock: mfma only", 256, 20000);
  clk<T_MFMA>("clock: mfma only", 2048, 20000);
  clk<T_MIX>("clock: mfma + softmax VALU mix", 256, 20000);
  clk<T_MIX>("clock: mfma + softmax VALU mix", 2048, 20000);
  clk<T_MIXVALU>("clock: VALU mix only", 2048, 20000);
  run<T_EXP, T_EXP>("v_exp_f32, one wave per SIMD", 256, 16, 16);
  run<T_SUB, T_SUB>("v_sub_f32, one wave per SIMD", 256, 16, 16);
  run<T_FMA, T_FMA>("v_fma_f32, one wave per SIMD", 256, 16, 16);
  run<T_MAX3, T_MAX3>("v_max3_f32, one wave per SIMD", 256, 16, 16);
  run<T_PKMUL, T_PKMUL>("v_pk_mul_f32, one wave per SIMD", 256, 16, 16);
  run<T_CVTPK, T_CVTPK>("v_cvt_pk_f16_f32, one wave per SIMD", 256, 16, 16);
  run<T_MFMA, T_MFMA>("mfma 32x32x16 f16, one wave per SIMD", 256, 16, 16);
  run<T_MFMA_EXP_SAMEWAVE, T_MFMA_EXP_SAMEWAVE>("mfma + 2 exp per mfma, SAME wave (per mfma)", 256, 16, 16);
  run<T_EXP, T_EXP>("v_exp_f32, two waves per SIMD", 512, 16, 16);
  run<T_SUB, T_SUB>("v_sub_f32, two waves per SIMD", 512, 16, 16);
  run<T_MFMA, T_MFMA>("mfma, two waves per SIMD", 512, 16, 16);
  run<T_EXP, T_MFMA>("wave0 exp | wave4 mfma (same SIMD)", 512, 16, 16);
  run<T_SUB, T_MFMA>("wave0 sub | wave4 mfma (same SIMD)", 512, 16, 16);
  run<T_EXP, T_SUB>("wave0 exp | wave4 sub (same SIMD)", 512, 16, 16);
  printf("-- per MFMA group (1 mfma + 7 valu: 2 exp, 2 sub, max3, cvt_pk, fma)\n");
  run<T_MIXVALU, T_MIXVALU>("the 7 VALU alone, one wave per SIMD", 256, 16, 16);
  run<T_MIXVALU, T_MIXVALU>("the 7 VALU alone, two waves per SIMD", 512, 16, 16);
  run<T_MIX, T_MIX>("mfma + 7 VALU interleaved, one wave per SIMD", 256, 16, 16);
  run<T_MIX, T_MIX>("mfma + 7 VALU interleaved, two waves per SIMD", 512, 16, 16);
  run<T_MIX, T_MIX>("mfma + 7 VALU interleaved, three waves per SIMD", 768, 16, 16);
  run<T_MIX_BURST, T_MIX_BURST>("16 mfma THEN 112 VALU, one wave per SIMD", 256, 16, 16);
  run<T_MIX_BURST, T_MIX_BURST>("16 mfma THEN 112 VALU, two waves per SIMD", 512, 16, 16);
  run<T_MIX_BURST, T_MIX_BURST>("16 mfma THEN 112 VALU, three waves per SIMD", 768, 16, 16);
  return 0;
}

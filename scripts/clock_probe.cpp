// Shader clock of the GPU as a kernel sees it, while ANOTHER process loads the chip (scripts/clock_probe.sh): one wave spins for
// ~20 ms between two readings of the shader-clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime).
// MFMA peaks are quoted at the 2.4 GHz boost clock; what the chip holds under a power-bound MFMA load is lower (MI355X_MICROARCH.md,
// "DVFS give-back").    hipcc --offload-arch=gfx950 -O2 scripts/clock_probe.cpp -o scripts/clock_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
__global__ void probe(unsigned long long* out, unsigned long long ticks) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memrealtime() - r0 < ticks) __builtin_amdgcn_s_sleep(8);
  out[0] = __builtin_amdgcn_s_memtime() - c0;
  out[1] = __builtin_amdgcn_s_memrealtime() - r0;
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 40;
  unsigned long long* d;
  hipMalloc(&d, 16);
  for (int i = 0; i < n; ++i) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 2000000ull);  // 20 ms at 100 MHz
    unsigned long long h[2];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%.0f MHz (%llu shader clocks in %.2f ms)\n", (double)h[0] / ((double)h[1] / 100.0), h[0], (double)h[1] / 1e5);
    fflush(stdout);
    usleep(200000);
  }
  return 0;
}

// Where does a CU mask put workgroups?  hipExtStreamCreateWithCUMask with three mask shapes -- every CU, the bits b with
// b % 8 in {0, 1} ("two XCDs" if mask bits go round the XCDs), the first quarter of the bits -- and a kernel whose workgroups
// record XCC_ID and the CU part of HW_ID.  Prints, per mask, workgroups per XCD and the number of distinct (XCD, SE, SH, CU).
// build: hipcc -O2 --offload-arch=gfx950 scripts/mask_probe.cpp -o scripts/mask_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where(unsigned* out, int spin) {
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_ID
  }
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32, grid = 2048;
  printf("CUs %d, mask words %d\n", ncu, words);
  unsigned* d;
  CK(hipMalloc(&d, grid * 2 * sizeof(unsigned)));
  std::vector<unsigned> h(grid * 2);
  const char* names[] = {"all", "bits b %% 8 in {0,1}", "bits 0..ncu/4-1", "bits b %% 8 in {2,3}", "bits ncu/4..ncu/2-1"};
  for (int m = 0; m < 5; ++m) {
    std::vector<uint32_t> mask(words, 0);
    for (int b = 0; b < ncu; ++b) {
      bool on = m == 0 || (m == 1 && b % 8 < 2) || (m == 2 && b < ncu / 4) || (m == 3 && (b % 8 == 2 || b % 8 == 3)) || (m == 4 && b >= ncu / 4 && b < ncu / 2);
      if (on) mask[b / 32] |= 1u << (b % 32);
    }
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, words, mask.data()));
    CK(hipMemsetAsync(d, 0xff, grid * 2 * sizeof(unsigned), s));
    hipLaunchKernelGGL(where, dim3(grid), dim3(256), 0, s, d, 20000);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, grid * 2 * sizeof(unsigned), hipMemcpyDeviceToHost));
    std::map<int, int> per_xcc;
    std::set<unsigned> cus;
    for (int i = 0; i < grid; ++i) {
      const unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
      per_xcc[xcc]++;
      cus.insert((xcc << 16) | ((hw >> 8) & 0xff) | (((hw >> 13) & 0x7) << 8));  // cu_id + sh_id (bits 8..12), se_id (13..15)
    }
    printf("mask %-24s: distinct CUs %3zu; workgroups per XCD:", names[m], cus.size());
    for (auto& kv : per_xcc) printf(" %d:%d", kv.first, kv.second);
    printf("\n  first 16 workgroups' XCDs:");
    for (int i = 0; i < 16; ++i) printf(" %u", h[2 * i] & 0xf);
    printf("\n");
    CK(hipStreamDestroy(s));
  }
  return 0;
}

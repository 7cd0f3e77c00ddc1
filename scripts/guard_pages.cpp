// Device buffers that END at an unmapped page (and start right after one): HIP's virtual-memory API reserves [guard][mapping][guard]
// and maps only the middle, so a kernel that reads or writes past either end of a buffer takes a GPU memory fault at once, wherever
// the allocator would otherwise have put a neighbour.  Used by scripts/guard_page_fuzz.py.
// build: hipcc -O2 -shared -fPIC scripts/guard_pages.cpp -o scripts/libguardpages.so
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
struct Rec { void* base; size_t reserved, mapped; hipMemGenericAllocationHandle_t h; };
static std::map<void*, Rec> g_recs;
static size_t g_gran = 0;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "guard_pages: %s: %s\n", #x, hipGetErrorString(e_)); return nullptr; } } while (0)

extern "C" size_t guard_granularity() { return g_gran; }

// at_end != 0: the buffer's last byte (rounded up to 16) is the mapping's last byte; else its first byte is the mapping's first
extern "C" void* guard_alloc(size_t bytes, int at_end) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  if (!g_gran) CK(hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum));
  const size_t need = (bytes + 15) / 16 * 16;
  const size_t mapped = (need + g_gran - 1) / g_gran * g_gran;
  Rec r;
  r.reserved = mapped + 2 * g_gran;
  r.mapped = mapped;
  CK(hipMemAddressReserve(&r.base, r.reserved, g_gran, nullptr, 0));
  CK(hipMemCreate(&r.h, mapped, &prop, 0));
  char* m = (char*)r.base + g_gran;
  CK(hipMemMap(m, mapped, 0, r.h, 0));
  hipMemAccessDesc d = {};
  d.location = prop.location;
  d.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(m, mapped, &d, 1));
  void* p = at_end ? (void*)(m + mapped - need) : (void*)m;
  g_recs[p] = r;
  return p;
}

extern "C" int guard_free(void* p) {
  auto it = g_recs.find(p);
  if (it == g_recs.end()) return -1;
  Rec r = it->second;
  g_recs.erase(it);
  (void)hipDeviceSynchronize();
  char* m = (char*)r.base + g_gran;
  if (hipMemUnmap(m, r.mapped) != hipSuccess) return -2;
  if (hipMemRelease(r.h) != hipSuccess) return -3;
  if (hipMemAddressFree(r.base, r.reserved) != hipSuccess) return -4;
  return 0;
}

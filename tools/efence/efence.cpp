// GPU "electric fence": a torch pluggable allocator for hunting out-of-bounds and uninitialised reads of the HIP engine.
// TEST INFRASTRUCTURE (tests/conftest.py activates it under DIG3D_EFENCE=1) — never on the product path.
//
// Every tensor gets its own virtual range [granule-aligned mapping][unmapped guard].  The tensor is placed so that its
// END (rounded up to 16 bytes — float4 loads of a row tail are legal) touches the guard: a kernel that reads or writes
// one vector past any allocation faults deterministically, on every box, instead of landing in the caching
// allocator's neighbouring block.  DIG3D_EFENCE=lo puts the guard BELOW the tensor instead (negative indices).
// The payload is filled with DIG3D_EFENCE_FILL (default 0x7f: int32 2139062143 — an index far outside anything,
// float 3.39e38), so a slot read before it is written shows up as a fault or as inf / NaN.
//
// free() waits for the device before unmapping: the engine's kernels are asynchronous and torch hands blocks back as
// soon as the Python object dies.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>

#include <mutex>
#include <unordered_map>

namespace {
struct Block {
  void* base;       // start of the reserved range
  size_t reserved;  // bytes reserved (mapping + guard)
  size_t mapped;    // bytes mapped
  void* map_at;     // where the mapping starts
  hipMemGenericAllocationHandle_t handle;
};
std::mutex mu;
std::unordered_map<void*, Block> live;
size_t gran = 0;
int fill = 0x7f;
bool guard_low = false;
size_t n_alloc = 0, peak_live = 0;

void die(const char* what, hipError_t e) {
  fprintf(stderr, "[efence] %s failed: %s\n", what, hipGetErrorString(e));
  abort();
}
#define CK(x)                          \
  do {                                 \
    hipError_t e_ = (x);               \
    if (e_ != hipSuccess) die(#x, e_); \
  } while (0)
}  // namespace

extern "C" {

void* efence_malloc(ssize_t size, int device, hipStream_t stream) {
  if (size <= 0) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  if (!gran) {
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    const char* f = getenv("DIG3D_EFENCE_FILL");
    if (f) fill = !strcmp(f, "none") ? -1 : (int)strtol(f, nullptr, 0);
    const char* m = getenv("DIG3D_EFENCE");
    guard_low = m && !strcmp(m, "lo");
    fprintf(stderr, "[efence] active: granule %zu bytes, fill 0x%02x, guard %s\n", gran, fill & 0xff,
            guard_low ? "below" : "above");
  }
  const size_t need = ((size_t)size + 15) & ~(size_t)15;
  const size_t mapped = (need + gran - 1) / gran * gran;
  Block b;
  b.reserved = mapped + 2 * gran;      // one unmapped granule on either side
  b.mapped = mapped;
  CK(hipMemAddressReserve(&b.base, b.reserved, gran, nullptr, 0));
  b.map_at = (char*)b.base + gran;
  CK(hipMemCreate(&b.handle, mapped, &prop, 0));
  CK(hipMemMap(b.map_at, mapped, 0, b.handle, 0));
  hipMemAccessDesc acc;
  memset(&acc, 0, sizeof(acc));
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(b.map_at, mapped, &acc, 1));
  // SYNCHRONOUS fill: a small pageable host->device copy is written by the host through the BAR as soon as it is issued —
  // it is not ordered behind an asynchronous memset still queued on the stream (first version of this file: the fill
  // landed AFTER the copy and the batch vector became 0x7f7f7f7f)
  (void)stream;
  if (fill >= 0) {
    CK(hipMemset(b.map_at, fill, mapped));
    CK(hipDeviceSynchronize());
  }
  void* p = guard_low ? b.map_at : (void*)((char*)b.map_at + mapped - need);
  live[p] = b;
  ++n_alloc;
  if (live.size() > peak_live) peak_live = live.size();
  return p;
}

void efence_free(void* ptr, ssize_t size, int device, hipStream_t stream) {
  if (!ptr) return;
  (void)size;
  (void)device;
  (void)stream;
  CK(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lk(mu);
  auto it = live.find(ptr);
  if (it == live.end()) {
    fprintf(stderr, "[efence] free of unknown pointer %p\n", ptr);
    abort();
  }
  Block b = it->second;
  live.erase(it);
  CK(hipMemUnmap(b.map_at, b.mapped));
  CK(hipMemRelease(b.handle));
  // the virtual range is NOT given back: a recycled range was observed to serve stale translations to the next tenant
  // (torch reductions over fresh tensors returned inconsistent results); 47 bits of address space outlast any test run,
  // and a dangling pointer into a dead tensor now faults for the rest of the process
}

void efence_stats(size_t* allocs, size_t* live_now, size_t* peak) {
  std::lock_guard<std::mutex> lk(mu);
  *allocs = n_alloc;
  *live_now = live.size();
  *peak = peak_live;
}

}  // extern "C"

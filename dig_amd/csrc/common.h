// dig3d — MI355X (gfx950 / CDNA4) message-passing engine for DIG's dig.threedgraph hot path.
// Shared device helpers.  Wavefront = 64 lanes everywhere in this code base.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DIG3D_OK 0
#define DIG3D_ERR_ARG (-1)      // bad size / null pointer / unsupported channel count
#define DIG3D_ERR_LAUNCH (-2)   // hipGetLastError() != hipSuccess after a launch

#define DIG3D_WAVE 64

// hipGetLastError() is per-thread sticky state shared with the host framework (PyTorch leaves benign
// errors such as hipErrorNotReady behind): clear it on entry, so a post-launch check reports OUR launch.
#define DIG3D_ENTER() (void)hipGetLastError()

#define DIG3D_CHECK_LAUNCH()                                   \
  do {                                                         \
    if (hipGetLastError() != hipSuccess) return DIG3D_ERR_LAUNCH; \
  } while (0)

// compute units of the current device (hipDeviceProp_t::multiProcessorCount; 256 on MI355X), read once: every worker /
// block-count heuristic of the library is a multiple of it instead of a constant tuned to one part
static inline int dig3d_num_cus() {
  static const int n = [] {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
    return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }();
  return n;
}

// Zero fill as a KERNEL, never hipMemsetAsync: a memset issued by this library inside a HIP-graph capture was not replayed
// reliably on ROCm 7 (round 5: the `add` buffer of ComENet's second arg-min, zeroed by hipMemsetAsync inside the captured
// step, came back with stale contents after a few replays interleaved with eager work — tools/diag_comenet_geom.py; the
// captured steps of rounds 1-4 contained no memset node of ours).  A kernel node has no such problem.
static __global__ void __launch_bounds__(256) k_dig3d_zero_words(uint32_t* __restrict__ p, size_t nwords) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
// bytes must be a multiple of 4 and p 4-byte aligned (every use in this library zeroes float / int32 / int64 arrays)
static inline hipError_t dig3d_zero_async(void* p, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  const size_t nwords = bytes >> 2;
  size_t nb = (nwords + 255) / 256;
  if (nb > 4096) nb = 4096;
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_dig3d_zero_words, dim3((unsigned)nb), dim3(256), 0, st, (uint32_t*)p, nwords);
  return hipGetLastError();
}

static inline int dig3d_blocks(int64_t work, int per_block) {
  int64_t b = (work + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > 2147483647LL) b = 2147483647LL;
  return (int)b;
}

// XCD-aware block order for the gather kernels.  Workgroups are dealt to the 8 XCDs round-robin (block b runs on XCD
// b mod 8) and every XCD has its own 4-MB L2, so with the natural order the rows one molecule gathers are fetched into
// all eight L2s.  The logical block id gives XCD x the CONTIGUOUS range [x * per, (x + 1) * per) of the work (whole
// molecules), so a gathered row is fetched by one L2.  The host rounds the grid up to a multiple of 8
// (dig3d_xcd_grid; the kernels guard the tail); `swz` == 0 keeps the natural order (A/B switch
// DIG3D_NO_XCD_SWIZZLE, read once per translation unit).
__device__ __forceinline__ int dig3d_xcd_block(int swz) {
  const int b = blockIdx.x;
  return swz ? (b & 7) * (int)(gridDim.x >> 3) + (b >> 3) : b;
}
static inline int dig3d_xcd_grid(int nblk) { return (nblk + 7) & ~7; }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ uint64_t lanemask_lt() {
  return (1ull << (threadIdx.x & 63)) - 1ull;
}

// ---- "reference float" helpers -------------------------------------------------------------
// The reference (torch CPU, AVX2/AVX512 builds) evaluates its float32 geometry with these exact
// IEEE operation sequences (probed; see DESIGN.md "reference float semantics").  This file is
// compiled with -ffp-contract=off, so a*b+c below is two roundings unless __fmaf_rn is spelled.
struct f3 {
  float x, y, z;
};

__device__ __forceinline__ f3 f3_sub(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 f3_neg(f3 a) { return {-a.x, -a.y, -a.z}; }

// torch.cross on x86+FMA: c0 = fma(a1, b2, -fl(a2*b1)) etc. (ATen CrossKernel.cpp contracted by GCC).
__device__ __forceinline__ f3 ref_cross(f3 a, f3 b) {
  f3 c;
  c.x = __fmaf_rn(a.y, b.z, -(a.z * b.y));
  c.y = __fmaf_rn(a.z, b.x, -(a.x * b.z));
  c.z = __fmaf_rn(a.x, b.y, -(a.y * b.x));
  return c;
}
// (a*b).sum(-1): three rounded products, summed left to right.
__device__ __forceinline__ float ref_dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// x.pow(2).sum(-1).sqrt()
__device__ __forceinline__ float ref_len(f3 a) { return sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z); }

// x.norm(dim=-1) (ATen norm kernel on x86+FMA): acc = fma(x, x, acc) left to right, then sqrt.
__device__ __forceinline__ float ref_norm(f3 a) {
  return sqrtf(__fmaf_rn(a.z, a.z, __fmaf_rn(a.y, a.y, a.x * a.x)));
}

__device__ __forceinline__ f3 load3(const float* __restrict__ p, int i) {
  const float* q = p + 3ll * i;
  return {q[0], q[1], q[2]};
}

#define DIG3D_2PI_F 6.28318530717958647692f
#define DIG3D_PI_F 3.14159265358979323846f

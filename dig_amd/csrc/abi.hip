// dig3d library identity: the hash of the sources + header this binary was built from (checked by the host at load time,
// dig_amd/_hip.py: a library from another checkout must raise, not write through mismatched arguments), and the device
// figures every worker-count heuristic of the library derives from (common.h dig3d_num_cus).
#include <string.h>

#include "common.h"

#ifndef DIG3D_ABI_HASH
#define DIG3D_ABI_HASH "unhashed"
#endif

extern "C" {

// out[cap] <- NUL-terminated hex hash of csrc/* + include/dig3d.h at build time; returns its length (0 if cap is too small)
int dig3d_abi_hash(char* out, int cap) {
  const char* h = DIG3D_ABI_HASH;
  const int n = (int)strlen(h);
  if (!out || cap < n + 1) return 0;
  memcpy(out, h, n + 1);
  return n;
}

// info[8] (host): [0] compute units the library's heuristics use, [1] wavefront size, [2] XCDs assumed by the block
// swizzle, [3] LDS bytes per workgroup, [4] device ordinal, [5] clock kHz, [6..7] total HBM bytes (lo, hi 32 bits)
int dig3d_device_info(int* info) {
  DIG3D_ENTER();
  if (!info) return DIG3D_ERR_ARG;
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return DIG3D_ERR_LAUNCH;
  info[0] = dig3d_num_cus();
  info[1] = p.warpSize;
  info[2] = 8;
  info[3] = (int)p.sharedMemPerBlock;
  info[4] = dev;
  info[5] = p.clockRate;
  info[6] = (int)(p.totalGlobalMem & 0xffffffffu);
  info[7] = (int)(p.totalGlobalMem >> 32);
  return DIG3D_OK;
}

}  // extern "C"

// Launch-time facts of one kernel instantiation (dynamic-LDS attribute, resident
// workgroups per CU) cached PER DEVICE and safe to use from several host threads.
// A process that switches HIP devices gets the attribute set and the occupancy
// queried again on the new device; two threads launching at once serialise on
// the instantiation's mutex only for the (rare) cache fill.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>

namespace fr {

constexpr int kMaxDevices = 64;

inline int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return dev;
}

inline int device_cu_count() {
  static std::mutex mu;
  static int cus[kMaxDevices] = {0};
  const int dev = current_device();
  const bool cached = dev >= 0 && dev < kMaxDevices;
  std::lock_guard<std::mutex> g(mu);
  if (cached && cus[dev] > 0) return cus[dev];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  if (n <= 0) n = 256;
  if (cached) cus[dev] = n;
  return n;
}

struct LaunchCache {
  struct Facts {
    size_t lds_attr = 0;            // largest dynamic LDS size the attribute was raised to
    size_t occ_lds = (size_t)-1;    // LDS size the occupancy below was queried for
    int per_cu = 0;
  };
  std::mutex mu;
  Facts dev[kMaxDevices];

  // Raises MaxDynamicSharedMemorySize when `lds` needs it and returns the resident
  // workgroups per CU of `kernel` at (`threads`, `lds`) on the CURRENT device.
  template <class K>
  hipError_t facts(K kernel, int threads, size_t lds, int *per_cu) {
    const int d = current_device();
    Facts local;
    std::unique_lock<std::mutex> g(mu);
    Facts &f = (d >= 0 && d < kMaxDevices) ? dev[d] : local;
    if (lds > 64 * 1024 && lds > f.lds_attr) {
      hipError_t e = hipFuncSetAttribute((const void *)kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      f.lds_attr = lds;
    }
    if (per_cu != nullptr) {
      if (f.occ_lds != lds || f.per_cu < 1) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, lds) !=
                hipSuccess ||
            nb < 1) {
          (void)hipGetLastError();
          nb = 1;
        }
        f.per_cu = nb;
        f.occ_lds = lds;
      }
      *per_cu = f.per_cu;
    }
    return hipSuccess;
  }
};

}  // namespace fr

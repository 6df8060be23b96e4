// Host-side launchers of the walk kernels (device code: walk_device.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_runtime_api.h>

#include "kernels.h"
#include "launch_cache.h"
#include "walk_device.h"
#include "walk_fused.h"

namespace fr {

// ---------------------------------------------------------------- launch
template <int E, int P, int LV, int MULTI, bool VEC, bool W, int TEAM = 4, int MODE = 0,
          int SEMI = 0, bool NT = false, bool TI = false>
static hipError_t launch_walk_cfg(const IssArgs &a, hipStream_t st) {
  using C = WalkCfg<E, P, LV, MULTI, VEC, W, TEAM, MODE, SEMI, NT, TI>;
  // rows, wave totals, LDS carries, then (fused, cooperative) the feature window: value,
  // population (MPI) and column of every slot
  const size_t lds = ((size_t)a.R * C::CHUNK + 2 * C::NW + 2 * C::NW +
                      (MULTI == 1 ? a.carry_slots : 0)) *
                         sizeof(double) +
                     ((MODE == 1 && TEAM != 1) ? feat_window_bytes(a.feat_window, a.has_mpi != 0) : 0);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static LaunchCache cache;  // per instantiation; per-device entries, thread-safe
  int per_cu = 1;
  hipError_t e = cache.facts(iss_walk_kernel<C>, kWalkThreads, lds,
                             a.persistent ? &per_cu : nullptr);
  if (e != hipSuccess) return e;
  const int64_t units = a.N * a.G;
  if (units > 0x7fffffffLL) return hipErrorInvalidValue;  // unit indices are 32-bit in the kernel
  int64_t blocks = units;
  if (a.persistent) {
    // at most one resident round of workgroups (a multiple of 8 for the XCD numbering)
    if (a.persistent > 1 && per_cu > a.persistent) per_cu = a.persistent;  // experiments: cap per CU
    int64_t resident = (int64_t)per_cu * device_cu_count();
    resident -= resident % 8;
    if (resident < 8) resident = 8;
    if (a.resident_out != nullptr) {  // the host only asks how many workgroups are resident
      *a.resident_out = (int32_t)resident;
      return hipSuccess;
    }
    if (blocks > resident) blocks = resident;
  } else if (a.resident_out != nullptr) {
    *a.resident_out = 0;
    return hipSuccess;
  }
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(iss_walk_kernel<C>, dim3((unsigned)blocks), dim3(kWalkThreads), lds, st, a);
  return hipGetLastError();
}

// launch of the fused walk (walk_fused.h): one workgroup per (series, group) unit
template <int E, int LV, int MULTI, bool W, int SEMI, bool TI, bool TOTAL, int MODE = 1, bool HO = false>
static hipError_t launch_fused_mode(const IssArgs &a, hipStream_t st) {
  // (MODE 2, the tensor is written: "E = 4" is two pieces of two elements per lane - every store
  // instruction of a wave then covers its 1 KiB without holes)
  using C = WalkCfg<MODE == 2 ? 2 : E, (MODE == 2 && E == 4) ? 2 : 1, LV, MULTI, true, W, 4, MODE, SEMI, false, TI, HO>;
  const size_t lds = ((size_t)a.R * C::CHUNK + 16 + 8 + (MULTI == 1 ? a.carry_slots : 0)) * sizeof(double) +
                     feat_window_bytes(a.feat_window, a.has_mpi != 0, false) + (size_t)a.lds_pad;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static LaunchCache cache;  // per instantiation; per-device entries, thread-safe
  int per_cu = 1;
  hipError_t e = cache.facts(iss_fused_kernel<C, TOTAL>, kWalkThreads, lds,
                             a.resident_out != nullptr ? &per_cu : nullptr);
  if (e != hipSuccess) return e;
  if (a.resident_out != nullptr) {  // the host only asks how many workgroups are resident
    int64_t resident = (int64_t)per_cu * device_cu_count();
    resident -= resident % 8;
    *a.resident_out = (int32_t)(resident < 8 ? 8 : resident);
    return hipSuccess;
  }
  const int64_t units = a.N * a.G;
  if (units > 0x7fffffffLL) return hipErrorInvalidValue;  // unit indices are 32-bit in the kernel
  hipLaunchKernelGGL((iss_fused_kernel<C, TOTAL>), dim3((unsigned)units), dim3(kWalkThreads), lds, st, a);
  return hipGetLastError();
}
// (the weighting mode of a plan - total or not - is a compile-time property of the fused walk)
template <int E, int LV, int MULTI, bool W, int SEMI = 0, bool TI = false, int MODE = 1, bool HO = false>
static hipError_t launch_fused_cfg(const IssArgs &a, hipStream_t st) {
  if constexpr (TI) return launch_fused_mode<E, LV, MULTI, W, SEMI, true, true, MODE, HO>(a, st);
  if constexpr (W) {
    if (a.total_weighting) return launch_fused_mode<E, LV, MULTI, true, SEMI, false, true, MODE, HO>(a, st);
  }
  return launch_fused_mode<E, LV, MULTI, W, SEMI, false, false, MODE, HO>(a, st);
}

// launch of a static program: same persistent grid as the interpreter's
template <class C, class PG>
static hipError_t launch_walk_static(const IssArgs &a, hipStream_t st) {
  const size_t lds = ((size_t)PG::rows * C::CHUNK + 4 * C::NW) * sizeof(double) + (size_t)a.lds_pad;
  static LaunchCache cache;
  int per_cu = 1;
  hipError_t e = cache.facts(iss_walk_static_kernel<C, PG>, kWalkThreads, lds, &per_cu);
  if (e != hipSuccess) return e;
  const int64_t units = a.N * PG::groups;
  if (units > 0x7fffffffLL || a.G != PG::groups) return hipErrorInvalidValue;
  if (a.persistent > 1 && per_cu > a.persistent) per_cu = a.persistent;  // experiments: cap per CU
  int64_t resident = (int64_t)per_cu * device_cu_count();
  resident -= resident % 8;
  if (resident < 8) resident = 8;
  if (a.resident_out != nullptr) {
    *a.resident_out = (int32_t)resident;
    return hipSuccess;
  }
  const int64_t blocks = (units < resident || !a.persistent) ? units : resident;
  hipLaunchKernelGGL((iss_walk_static_kernel<C, PG>), dim3((unsigned)blocks), dim3(kWalkThreads),
                     lds, st, a);
  return hipGetLastError();
}

}  // namespace fr

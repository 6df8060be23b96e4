// Wave64 DPP scan primitives and small device helpers shared by the kernels.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

namespace fr {

typedef double vd2 __attribute__((ext_vector_type(2)));

// The program tables are read-only for the whole launch.  Reading them through
// the constant address space makes every (wave-uniform) access a scalar load
// (s_load, counted by lgkmcnt).  As plain global loads they would be VECTOR loads
// counted by vmcnt, and waiting for one of those also waits for every output
// store issued before it - serialising the store stream node by node.
template <class T>
using cptr = const T __attribute__((address_space(4))) *;
template <class T>
__device__ __forceinline__ cptr<T> as_const(const T *p) {
  return (cptr<T>)(p);
}

// Pointers the compiler KNOWS to be LDS: an add through one is a ds_add_f64 (no return value,
// counted by lgkmcnt), through a generic pointer it would be a flat atomic.
typedef double __attribute__((address_space(3))) lds_f64;
typedef int __attribute__((address_space(3))) lds_i32;
__device__ __forceinline__ void lds_add(lds_f64 *p, double v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---------------------------------------------------------------- wave scan
// The scan is written for both semirings: SEMI 0 = Reals (the "sum" is +, identity
// 0.0), SEMI 1 = Arctic and SEMI 2 = Bayesian (the "sum" is max, identity -inf).
template <int SEMI>
__device__ __forceinline__ double semi_add(double a, double b) {
  if constexpr (SEMI == 0) return a + b;
  return fmax(a, b);
}
template <int SEMI>
__device__ __forceinline__ double semi_zero() {
  if constexpr (SEMI == 0) return 0.0;
  return -__builtin_inf();
}

// DPP fetch of a double: the value of the source lane, the identity where the lane
// has no source.  Reals with all rows enabled: bound_ctrl supplies the zeros and no
// "old" value has to be materialised; otherwise lanes without a source keep old.
template <int CTRL, int ROW_MASK, int SEMI = 0>
__device__ __forceinline__ double dpp_fetch(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  if constexpr (SEMI == 0 && ROW_MASK == 0xf) {
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  } else {
    constexpr int id_hi = SEMI == 0 ? 0 : (int)0xfff00000;  // high word of -inf
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(id_hi, hi, CTRL, ROW_MASK, 0xf, false);
  }
  return __hiloint2double(hi, lo);
}

template <int SEMI = 0>
__device__ __forceinline__ double wave_inclusive_scan(double v) {
  v = semi_add<SEMI>(v, dpp_fetch<0x111, 0xf, SEMI>(v));  // row_shr:1
  v = semi_add<SEMI>(v, dpp_fetch<0x112, 0xf, SEMI>(v));  // row_shr:2
  v = semi_add<SEMI>(v, dpp_fetch<0x114, 0xf, SEMI>(v));  // row_shr:4
  v = semi_add<SEMI>(v, dpp_fetch<0x118, 0xf, SEMI>(v));  // row_shr:8
  v = semi_add<SEMI>(v, dpp_fetch<0x142, 0xa, SEMI>(v));  // row_bcast:15 -> rows 1,3
  v = semi_add<SEMI>(v, dpp_fetch<0x143, 0xc, SEMI>(v));  // row_bcast:31 -> rows 2,3
  return v;
}

// P independent scans, advanced step by step so their DPP latencies overlap
template <int P, int SEMI = 0>
__device__ __forceinline__ void wave_inclusive_scan_multi(double (&v)[P]) {
#pragma unroll
  for (int h = 0; h < P; ++h) v[h] = semi_add<SEMI>(v[h], dpp_fetch<0x111, 0xf, SEMI>(v[h]));
#pragma unroll
  for (int h = 0; h < P; ++h) v[h] = semi_add<SEMI>(v[h], dpp_fetch<0x112, 0xf, SEMI>(v[h]));
#pragma unroll
  for (int h = 0; h < P; ++h) v[h] = semi_add<SEMI>(v[h], dpp_fetch<0x114, 0xf, SEMI>(v[h]));
#pragma unroll
  for (int h = 0; h < P; ++h) v[h] = semi_add<SEMI>(v[h], dpp_fetch<0x118, 0xf, SEMI>(v[h]));
#pragma unroll
  for (int h = 0; h < P; ++h) v[h] = semi_add<SEMI>(v[h], dpp_fetch<0x142, 0xa, SEMI>(v[h]));
#pragma unroll
  for (int h = 0; h < P; ++h) v[h] = semi_add<SEMI>(v[h], dpp_fetch<0x143, 0xc, SEMI>(v[h]));
}

template <int SEMI = 0>
__device__ __forceinline__ double wave_shift_right1(double v) {
  return dpp_fetch<0x138, 0xf, SEMI>(v);  // wave_shr:1, lane 0 gets the identity
}

__device__ __forceinline__ double wave_last_lane(double v) {
  // lane 63's value as a wave-uniform (scalar) double
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for
// vmcnt(0), i.e. for the acknowledgement of every global store the wave has in
// flight - that would serialise each node's output stores with the next node's
// scan.  Waiting for lgkmcnt(0) alone keeps the stores streaming.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace fr

// numpy's summation order for a contiguous float64 row, as a workgroup routine.
//
// STD (fruits/preparation/transform.py:125-147) is np.mean / np.std along the contiguous
// axis, i.e. np.add.reduce: the reduction runs through the ufunc machinery in buffers of
// 8192 elements, `res = 0.0; for every buffer: res = res + pairwise_sum(buffer)`, and
// pairwise_sum (numpy/_core/src/umath/loops_utils.h.src) is
//     n < 8:    a plain left-to-right sum;
//     n <= 128: eight accumulators r[j] = a[j], r[j] += a[i + j] (i = 8, 16, ...), combined
//               as ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)), then the n % 8 trailing elements
//               added one by one;
//     else:     n2 = n / 2 rounded down to a multiple of 8, sum(a[:n2]) + sum(a[n2:]).
// A deterministic order, so it can be reproduced bit for bit: the leaves of the recursion are
// independent (eight lanes per leaf, one per accumulator), the tree above them is a handful of
// additions that one thread performs in the recursion's order.  Checked against numpy itself
// for T = 1 ... 200 000 (tests/test_host.py::test_numpy_sum_model, tests/test_hip_parity.py).
#pragma once
#include <cstdint>
// (the order logic below also compiles for the host: tests/native/pairwise_host.cpp runs it
// against numpy on the CPU)
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define FR_PW_FN __device__ inline
#else
#define FR_PW_FN inline
#endif

namespace fr {

constexpr int kNpBuffer = 8192;      // numpy's ufunc buffer size (elements)
constexpr int kPwBlock = 128;        // PW_BLOCKSIZE
constexpr int kPwMaxLeaves = kNpBuffer / 64;   // a leaf holds > 64 elements once n > 128

struct PairwiseShared {
  int32_t leaf_off[kPwMaxLeaves];
  int32_t leaf_len[kPwMaxLeaves];
  double leaf_sum[kPwMaxLeaves];
  int32_t stack_off[32];
  int32_t stack_len[32];
  double values[16];
  int32_t n_leaves;
  int32_t table_n;       // the buffer length the leaf table was built for
  double result;
};

// thread 0: the leaves of pairwise_sum's recursion over n elements, left to right
FR_PW_FN void pw_build_leaves(PairwiseShared &sh, int n) {
  int sp = 0, leaves = 0;
  sh.stack_off[0] = 0;
  sh.stack_len[0] = n;
  sp = 1;
  while (sp > 0) {
    --sp;
    const int o = sh.stack_off[sp], m = sh.stack_len[sp];
    if (m <= kPwBlock) {
      sh.leaf_off[leaves] = o;
      sh.leaf_len[leaves] = m;
      ++leaves;
    } else {
      int m2 = m / 2;
      m2 -= m2 % 8;
      sh.stack_off[sp] = o + m2;          // right half below the left one: left is popped first
      sh.stack_len[sp] = m - m2;
      sh.stack_off[sp + 1] = o;
      sh.stack_len[sp + 1] = m2;
      sp += 2;
    }
  }
  sh.n_leaves = leaves;
  sh.table_n = n;
}

// thread 0: the additions above the leaves, in the recursion's order (left + right)
FR_PW_FN double pw_combine(PairwiseShared &sh, int n) {
#pragma clang fp contract(off)
  int sp = 1, vp = 0, leaf = 0;
  sh.stack_len[0] = n;
  while (sp > 0) {
    --sp;
    const int m = sh.stack_len[sp];
    if (m < 0) {                          // both halves are on the value stack
      sh.values[vp - 2] = sh.values[vp - 2] + sh.values[vp - 1];
      --vp;
    } else if (m <= kPwBlock) {
      sh.values[vp++] = sh.leaf_sum[leaf++];
    } else {
      int m2 = m / 2;
      m2 -= m2 % 8;
      sh.stack_len[sp] = -1;
      sh.stack_len[sp + 1] = m - m2;
      sh.stack_len[sp + 2] = m2;
      sp += 3;
    }
  }
  return sh.values[0];
}

// lane j (0 ... 7) of a leaf of m >= 8 elements at o: its accumulator r[j]
template <class F>
FR_PW_FN double pw_leaf_lane(F &value, int64_t o, int m, int j) {
#pragma clang fp contract(off)
  const int full = m - (m % 8);
  double r = value(o + j);
  for (int i = 8; i < full; i += 8) r = r + value(o + i + j);
  return r;
}

// the n % 8 trailing elements of a leaf (all of it when m < 8), added one by one
template <class F>
FR_PW_FN double pw_leaf_tail(F &value, int64_t o, int m, double res) {
#pragma clang fp contract(off)
  for (int i = m < 8 ? 0 : m - (m % 8); i < m; ++i) res = res + value(o + i);
  return res;
}

#ifdef __HIPCC__
// np.add.reduce over value(0) ... value(T - 1); every thread of the workgroup calls it
// (blockDim.x a multiple of 64), every thread gets the sum.
template <class F>
__device__ inline double np_sum_row(F value, int64_t T, PairwiseShared &sh) {
#pragma clang fp contract(off)
  const int tid = (int)threadIdx.x, j = tid & 7, grp = tid >> 3;
  const int groups = (int)(blockDim.x >> 3);
  double total = 0.0;
  if (tid == 0) sh.table_n = -1;
  for (int64_t c0 = 0; c0 < T; c0 += kNpBuffer) {
    const int n = (int)((T - c0) < (int64_t)kNpBuffer ? (T - c0) : (int64_t)kNpBuffer);
    __syncthreads();
    if (tid == 0 && sh.table_n != n) pw_build_leaves(sh, n);
    __syncthreads();
    const int leaves = sh.n_leaves;
    for (int l = grp; l < leaves; l += groups) {
      const int64_t o = c0 + sh.leaf_off[l];
      const int m = sh.leaf_len[l];
      double res = 0.0;
      if (m >= 8) {
        double r = pw_leaf_lane(value, o, m, j);
        r = r + __shfl_xor(r, 1);          // (r0+r1), (r2+r3), (r4+r5), (r6+r7)
        r = r + __shfl_xor(r, 2);          // lane 0: (r0+r1) + (r2+r3); lane 4: (r4+r5) + (r6+r7)
        res = r + __shfl_xor(r, 4);        // lane 0: left + right
      }
      if (j == 0) sh.leaf_sum[l] = pw_leaf_tail(value, o, m, res);
    }
    __syncthreads();
    if (tid == 0) total = total + pw_combine(sh, n);
  }
  if (tid == 0) sh.result = total;
  __syncthreads();
  const double out = sh.result;
  __syncthreads();
  return out;
}
#endif

}  // namespace fr

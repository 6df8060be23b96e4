// Host harness for csrc/pairwise.h: the order logic of the device routine (leaf table, the
// eight accumulators of a leaf and their combination as the lanes form it, the additions above
// the leaves, numpy's 8192-element buffers) run on the CPU so that tests/test_host.py can
// compare it with numpy itself without a GPU.  The lanes' xor-shuffles are emulated on an array.
#include <cstdint>
#include "../../fruits_amd/csrc/pairwise.h"

extern "C" double pw_host_sum(const double *a, int64_t T) {
  fr::PairwiseShared sh;
  sh.table_n = -1;
  auto value = [&](int64_t t) { return a[t]; };
  double total = 0.0;
  for (int64_t c0 = 0; c0 < T; c0 += fr::kNpBuffer) {
    const int n = (int)((T - c0) < (int64_t)fr::kNpBuffer ? (T - c0) : (int64_t)fr::kNpBuffer);
    if (sh.table_n != n) fr::pw_build_leaves(sh, n);
    for (int l = 0; l < sh.n_leaves; ++l) {
      const int64_t o = c0 + sh.leaf_off[l];
      const int m = sh.leaf_len[l];
      double res = 0.0;
      if (m >= 8) {
        double r[8], s[8];
        for (int j = 0; j < 8; ++j) r[j] = fr::pw_leaf_lane(value, o, m, j);
        for (int x = 1; x <= 4; x <<= 1) {          // r = r + shfl_xor(r, x) on every lane
          for (int j = 0; j < 8; ++j) s[j] = r[j] + r[j ^ x];
          for (int j = 0; j < 8; ++j) r[j] = s[j];
        }
        res = r[0];
      }
      sh.leaf_sum[l] = fr::pw_leaf_tail(value, o, m, res);
    }
    total = total + fr::pw_combine(sh, n);
  }
  return total;
}

// Small HIP kernels around the trie walk (exp tables, INC, path-length lookups,
// standalone sieves, STD) and the dispatcher over the walk-kernel instances.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <vector>

#include "kernels.h"
#include "walk_scan.h"
#include "pairwise.h"

namespace fr {

// ---------------------------------------------------------------- exp tables
// aux[2a]   = exp( g * alpha_a)   (np.exp(weights * alpha[k]),  semiring.py:123,150)
// aux[2a+1] = exp(-g * alpha_a)   (np.exp(-weights * alpha[k]), semiring.py:119,153,157)
// Arctic (linear = 1): aux[a] = g * alpha_a   (weights * alpha[k], semiring.py:297,306,330)
__global__ void exp_tables_kernel(const double *__restrict__ g, int64_t count,
                                  const float *__restrict__ alphas, int n_alpha,
                                  double *__restrict__ aux, int linear) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const double w = g[i];
  if (linear) {
    for (int a = 0; a < n_alpha; ++a) aux[(int64_t)a * count + i] = w * (double)alphas[a];
    return;
  }
  for (int a = 0; a < n_alpha; ++a) {
    const double al = (double)alphas[a];
    aux[(int64_t)(2 * a) * count + i] = exp(w * al);
    aux[(int64_t)(2 * a + 1) * count + i] = exp(-w * al);
  }
}

// ---------------------------------------------------------------- increments
__global__ void increments_kernel(const double *__restrict__ X, int64_t rows, int64_t T,
                                  int64_t shift, double *__restrict__ out,
                                  const double *__restrict__ head_src, int64_t head) {
  const int64_t total = rows * T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i % T;
    double v = (t >= shift) ? X[i] - X[i - shift] : 0.0;
    if (head_src != nullptr && t < head) v = head_src[i];
    out[i] = v;
  }
}

// ---------------------------------------------------------------- path-length lookup
// One workgroup per series: r = cumsum_t |dx_0| (or dx_0^2), optional /(last+1e-5),
// min-max normalise, * scale.  fruits/iss/weighting.py:148-160, cache.py:25-40,
// preparation/transform.py:184-198.
__device__ __forceinline__ double block_reduce_minmax(double v, bool is_max, double *sm) {
  for (int o = 32; o > 0; o >>= 1) {
    double w = __shfl_xor(v, o);
    v = is_max ? fmax(v, w) : fmin(v, w);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  double r = sm[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = is_max ? fmax(r, sm[w]) : fmin(r, sm[w]);
  return r;
}

__global__ __launch_bounds__(256) void pathlen_lookup_kernel(const double *__restrict__ X,
                                                              int64_t D, int64_t T, int norm,
                                                              int relative, double scale,
                                                              int exact,
                                                              double *__restrict__ out) {
  __shared__ double sm_red[4];
  __shared__ double sm_last;
  __shared__ double sm_tot[2][4];
  const int64_t n = blockIdx.x;
  const double *x = X + n * D * T;  // dimension 0 only
  double *o = out + n * T;
  const int tid = threadIdx.x;
  // The cumulative path length is summed SEQUENTIALLY (one lane per series) like
  // np.cumsum in the reference (fruits/cache.py:25-40): the lookup is then bit-identical
  // to the reference's, and with it every max-plus (Arctic) result - a running maximum
  // has long plateaus, and a fitted quantile that equals a plateau value is an exact tie
  // for all of its elements (a 1-ulp difference in g would move the whole plateau to the
  // other band).  O(T) dependent adds per series, all series in parallel: ~20 us at T = 4096.
  // (the summands are formed and the results stored by the whole workgroup through LDS;
  // only the adds themselves are serial)
  // exact == 0 (Reals plans, whose sums are re-associated anyway): a parallel scan.
  constexpr int kSeg = 4096;
  __shared__ double seg[kSeg];
  double acc = 0.0;  // meaningful in thread 0 only
  if (!exact) {
    const int lane = tid & 63, wave = tid >> 6;
    double run_carry = 0.0;
    int buf = 0;
    for (int64_t t0 = 0; t0 < T; t0 += 512) {
      double s2[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int64_t t = t0 + tid * 2 + e;
        double d = 0.0;
        if (t < T && t >= 1) d = x[t] - x[t - 1];
        s2[e] = (norm == 1) ? fabs(d) : d * d;
        if (t >= T) s2[e] = 0.0;
      }
      const double l1 = s2[0] + s2[1];
      const double incl = wave_inclusive_scan(l1);
      const double excl = wave_shift_right1(incl);
      if (lane == 63) sm_tot[buf][wave] = incl;
      __syncthreads();
      double run = run_carry, base = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if (w == wave) base = run;
        run += sm_tot[buf][w];
      }
      run_carry = run;
      buf ^= 1;
      const double off = base + excl;
      const int64_t t = t0 + tid * 2;
      if (t < T) o[t] = off + s2[0];
      if (t + 1 < T) o[t + 1] = off + l1;
    }
    acc = run_carry;  // the same value in every thread
  }
  for (int64_t c0 = 0; exact && c0 < T; c0 += kSeg) {
    const int len = (int)((T - c0) < kSeg ? (T - c0) : kSeg);
    for (int i = tid; i < len; i += blockDim.x) {
      const int64_t t = c0 + i;
      const double d = t >= 1 ? x[t] - x[t - 1] : 0.0;
      seg[i] = (norm == 1) ? fabs(d) : d * d;
    }
    __syncthreads();
    if (tid == 0) {
      int i = 0;
      for (; i + 8 <= len; i += 8) {  // 8 LDS reads in flight, then the 8 dependent adds
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = seg[i + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc += v[j];
          v[j] = acc;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) seg[i + j] = v[j];
      }
      for (; i < len; ++i) {
        acc += seg[i];
        seg[i] = acc;
      }
    }
    __syncthreads();
    for (int i = tid; i < len; i += blockDim.x) o[c0 + i] = seg[i];
    __syncthreads();
  }
  if (tid == 0) sm_last = acc;
  __syncthreads();  // (also: every thread re-reads elements other threads wrote)
  const double carry = sm_last;
  if (relative == 2) return;  // raw cumulative path length (SharedSeedCache entry)
  const double last = carry;
  double mn = INFINITY, mx = -INFINITY;
  for (int64_t t0 = 0; t0 < T; t0 += 512)
    for (int e = 0; e < 2; ++e) {
      const int64_t t = t0 + tid * 2 + e;
      if (t < T) {
        double v = o[t];
        if (relative) v = v / (last + 1e-5);
        mn = fmin(mn, v);
        mx = fmax(mx, v);
      }
    }
  mn = block_reduce_minmax(mn, false, sm_red);
  mx = block_reduce_minmax(mx, true, sm_red);
  for (int64_t t0 = 0; t0 < T; t0 += 512)
    for (int e = 0; e < 2; ++e) {
      const int64_t t = t0 + tid * 2 + e;
      if (t < T) {
        double v = o[t];
        if (relative) v = v / (last + 1e-5);
        o[t] = (mn != mx) ? ((v - mn) / (mx - mn)) * scale : 0.0 * scale;
      }
    }
}

// ---------------------------------------------------------------- sieves on (N,T)
// value of the inc-times differenced series at t (IncrementSieve._pre_transform,
// fruits/sieving/increment.py:63-71 with _increments of fruits/cache.py:8-13):
// D_0 = A, D_k[t] = D_{k-1}[t] - D_{k-1}[t-1] for t >= 1, D_k[0] = 0.
constexpr int kMaxInc = 8;
__device__ __forceinline__ double diff_at(const double *__restrict__ row, int64_t t, int inc) {
  double v[kMaxInc + 1];
#pragma unroll
  for (int j = 0; j <= kMaxInc; ++j) v[j] = (j <= inc && t - j >= 0) ? row[t - j] : 0.0;
#pragma unroll
  for (int lvl = 1; lvl <= kMaxInc; ++lvl) {
    if (lvl <= inc) {
#pragma unroll
      for (int j = 0; j + lvl <= kMaxInc; ++j)
        if (j <= inc - lvl) v[j] = (t - j >= 1) ? v[j] - v[j + 1] : 0.0;
    }
  }
  return v[0];
}

__global__ __launch_bounds__(256) void sieve_kernel(int kind, const double *__restrict__ A,
                                                     int64_t T, int64_t a_stride, int inc,
                                                     const int64_t *__restrict__ cuts,
                                                     int64_t cut_rows, int C1,
                                                     const double *__restrict__ q, int Q1,
                                                     double *__restrict__ out,
                                                     int64_t out_stride) {
  __shared__ double sm_sum[4];
  __shared__ double sm_cnt[4];
  const int64_t n = blockIdx.x;
  const double *row = A + n * a_stride;
  const int64_t *cut = cuts + (cut_rows == 1 ? 0 : n * C1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (kind == FR_SIEVE_END_K) {
    // out[n, j] = A[n, cut_{j+1} - 1]; index -1 wraps like numpy (segment.py:213-218)
    for (int j = tid; j < C1 - 1; j += blockDim.x) {
      int64_t idx = cut[j + 1] - 1;
      if (idx < 0) idx += T;
      // out of range: the reference raises IndexError (the host validates integer cuts);
      // a device cut table that slipped through yields NaN, never a stray read
      out[n * out_stride + j] = (idx >= 0 && idx < T) ? row[idx] : __builtin_nan("");
    }
    return;
  }
  const int Q = Q1 - 1;
  for (int j = 0; j < C1 - 1; ++j) {
    int64_t lo = cut[j], hi = cut[j + 1];
    if (lo < 0) lo = 0;
    if (hi > T) hi = T;
    for (int k = 0; k < Q; ++k) {
      const double qlo = q[k], qhi = q[k + 1];
      double sum = 0.0, cnt = 0.0;
      for (int64_t t = lo + tid; t < hi; t += blockDim.x) {
        const double v = diff_at(row, t, inc);
        if (qlo < v && v <= qhi) {
          sum += v;
          cnt += 1.0;
        }
      }
      for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o);
        cnt += __shfl_xor(cnt, o);
      }
      __syncthreads();
      if (lane == 0) {
        sm_sum[wave] = sum;
        sm_cnt[wave] = cnt;
      }
      __syncthreads();
      if (tid == 0) {
        double s = 0.0, c = 0.0;
        for (int w = 0; w < 4; ++w) {
          s += sm_sum[w];
          c += sm_cnt[w];
        }
        out[n * out_stride + j * Q + k] =
            (kind == FR_SIEVE_NPI_K) ? c : (c > 0.0 ? s / c : 0.0);
      }
    }
  }
}

// IncrementSieve._pre_transform (inc >= 0) materialised: out[n,t] = D_inc[n,t]
__global__ void pre_transform_kernel(const double *__restrict__ A, int64_t N, int64_t T,
                                     int64_t a_stride, int inc, double *__restrict__ out) {
  const int64_t total = N * T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / T, t = i % T;
    out[i] = diff_at(A + n * a_stride, t, inc);
  }
}

// STD preparateur, separately=True (fruits/preparation/transform.py:141-147):
// per (series, dimension) row: (x - mean) / (std + eps), std = population std
// (np.std), or 1 when var=False.  mean = np.add.reduce(x) / T and
// std = sqrt(np.add.reduce((x - mean)^2) / T) in NUMPY'S summation order (pairwise.h):
// the statistics, and with them every standardised value, are bit-identical to the
// reference's - a last-bit difference here flips the plateau ties of the max-plus
// semirings behind it.
__global__ __launch_bounds__(256) void standardize_kernel(const double *__restrict__ X, int64_t T,
                                                           int div_std, double eps,
                                                           double *__restrict__ out) {
#pragma clang fp contract(off)
  __shared__ PairwiseShared sh;
  const double *x = X + (int64_t)blockIdx.x * T;
  double *o = out + (int64_t)blockIdx.x * T;
  const double mean = np_sum_row([&](int64_t t) { return x[t]; }, T, sh) / (double)T;
  double sd = 1.0;
  if (div_std) {
    const double v = np_sum_row(
        [&](int64_t t) {
          const double d = x[t] - mean;
          return d * d;
        },
        T, sh);
    sd = sqrt(v / (double)T);
  }
  const double den = sd + eps;
  for (int64_t t = threadIdx.x; t < T; t += blockDim.x) o[t] = (x[t] - mean) / den;
}

// Statistics of the PREPARED rows for the fused preparation of the walk kernel (walk.h):
// prepared dimension d' = prep[4 d'] (raw dimension), prep[4 d' + 1] (increment lag, 0 none).
// The same sums in the same order as standardize_kernel over the materialised rows (numpy's),
// so the fused pipeline reproduces the unfused one, and both the reference, bit for bit.
// stats[(n * n_prep + d') * 2] = mean, [.. + 1] = std + eps (1 + eps when div_std == 0).
__global__ __launch_bounds__(256) void row_stats_kernel(const double *__restrict__ X, int64_t D,
                                                         int64_t T, const int32_t *__restrict__ prep,
                                                         int n_prep, int div_std, double eps,
                                                         double *__restrict__ stats) {
#pragma clang fp contract(off)
  __shared__ PairwiseShared sh;
  const int64_t n = blockIdx.x / n_prep;
  const int dp = (int)(blockIdx.x % n_prep);
  const int raw = prep[4 * dp], lag = prep[4 * dp + 1];
  const double *x = X + (n * D + raw) * T;
  auto value = [&](int64_t t) -> double {
    if (lag <= 0) return x[t];
    return t >= lag ? x[t] - x[t - lag] : 0.0;
  };
  const double mean = np_sum_row(value, T, sh) / (double)T;
  double sd = 1.0;
  if (div_std) {
    const double v = np_sum_row(
        [&](int64_t t) {
          const double d = value(t) - mean;
          return d * d;
        },
        T, sh);
    sd = sqrt(v / (double)T);
  }
  if (threadIdx.x == 0) {
    stats[(n * n_prep + dp) * 2] = mean;
    stats[(n * n_prep + dp) * 2 + 1] = sd + eps;
  }
}

hipError_t launch_row_stats(const double *X, int64_t N, int64_t D, int64_t T, const int32_t *prep,
                            int n_prep, int div_std, double eps, double *stats, hipStream_t st) {
  if (N <= 0 || T <= 0 || n_prep <= 0) return hipSuccess;
  if (N * n_prep > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)(N * n_prep)), dim3(256), 0, st, X, D, T,
                     prep, n_prep, div_std, eps, stats);
  return hipGetLastError();
}

// ---------------------------------------------------------------- Arctic argmax
// Arctic(argmax=True), fruits/iss/semiring.py:239-284.  The running maxima V of every
// prefix of every word come from the walk kernel (bit-exact); this is the rest:
// (1) positions: result[2k+1, i] of the reference is the index at which the running
//     maximum was last raised (`>=` keeps the earlier index), i.e. a running maximum of
//     i * [V[i] > V[i-1]] - computable from the materialised row alone;
// (2) the back-tracking of :275-283: for prefix k (index = k + k(k+1)/2) row index is V_k,
//     row index+k+1 is P_k, and for s = k..1 row index+s is P_(s-1) frozen from the final
//     position of row index+s+1 on: R_s[t] = P_(s-1)[min(t, m_s)], m_(k+1) = T-1,
//     m_s = P_s[m_(s+1)].
__global__ __launch_bounds__(256) void argmax_positions_kernel(const double *__restrict__ V,
                                                                int64_t T,
                                                                double *__restrict__ P) {
  __shared__ double sm[256];
  const double *v = V + (int64_t)blockIdx.x * T;
  double *p = P + (int64_t)blockIdx.x * T;
  const int tid = threadIdx.x;
  const int64_t per = (T + 255) / 256, lo = tid * per, hi = lo + per < T ? lo + per : T;
  double best = 0.0;   // positions are exact small integers in a double
  for (int64_t t = lo > 0 ? lo : 1; t < hi; ++t)
    if (v[t] > v[t - 1]) best = (double)t;
  sm[tid] = best;
  __syncthreads();
  double before = 0.0;
  for (int i = 0; i < tid; ++i) before = fmax(before, sm[i]);
  double run = before;
  for (int64_t t = lo; t < hi; ++t) {
    if (t > 0 && v[t] > v[t - 1]) run = (double)t;
    p[t] = run;
  }
}

// jobs (n_jobs, 3) int32: {first V / P row of the word, level k, first output row of prefix k}
__global__ __launch_bounds__(256) void argmax_assemble_kernel(
    const double *__restrict__ V, const double *__restrict__ P, int64_t N, int64_t T,
    const int32_t *__restrict__ jobs, double *__restrict__ out) {
  __shared__ int64_t m[64];   // m_s for s = 1..k+1 (words of <= 63 letters)
  const int64_t n = blockIdx.x;
  const int32_t *jb = jobs + 3 * (int64_t)blockIdx.y;
  const int64_t row0 = jb[0], index = jb[2];
  const int k = jb[1];
  auto prow = [&](int level) { return P + ((row0 + level) * N + n) * T; };
  if (threadIdx.x == 0) {
    m[k + 1] = T - 1;
    for (int s_ = k; s_ >= 1; --s_) m[s_] = (int64_t)prow(s_)[m[s_ + 1]];
  }
  __syncthreads();
  const double *v = V + ((row0 + k) * N + n) * T;
  for (int64_t t = threadIdx.x; t < T; t += blockDim.x) {
    out[(index * N + n) * T + t] = v[t];
    for (int s_ = 1; s_ <= k + 1; ++s_) {
      const int64_t tt = t < m[s_] ? t : m[s_];
      out[((index + s_) * N + n) * T + t] = prow(s_ - 1)[tt];
    }
  }
}

hipError_t launch_arctic_argmax(const double *V, int64_t rows, int64_t N, int64_t T, int n_jobs,
                                const int32_t *jobs, double *P, double *out, hipStream_t st) {
  if (rows <= 0 || N <= 0 || T <= 0 || n_jobs <= 0) return hipSuccess;
  if (rows * N > 0x7fffffffLL || N > 0x7fffffffLL || n_jobs > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(argmax_positions_kernel, dim3((unsigned)(rows * N)), dim3(256), 0, st, V, T, P);
  hipLaunchKernelGGL(argmax_assemble_kernel, dim3((unsigned)N, (unsigned)n_jobs), dim3(256), 0, st,
                     V, P, N, T, jobs, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------- Arctic argmax + sieves
// The rows of Arctic(argmax=True) straight into NPI / MPI / END features (fr_pipeline_set_argmax):
// one workgroup per (series, word).  Of a word of L letters the running maxima V_0 .. V_(L-1) of
// its prefixes exist (the walk kernel wrote them); prefix k contributes the row V_k and k + 1
// position rows (see above), L + L (L + 1) / 2 rows in all, every one a function of V alone - so
// none of them is written: V_k is staged in LDS, its positions P_k join the positions of the
// prefixes in front (LDS, 16 bits each: T < 65536), and every row is formed element by element
// for the feature ops that look at it - R_s[t] = P_(s-1)[min(t, m_s)].
struct ArgmaxWord {
  int32_t v_row0, L, out_row0, pad;
};
// values of a row: V_k, or a frozen position row
struct ArgmaxRow {
  const double *v;            // LDS: V_k, or nullptr
  const unsigned short *p;    // LDS: P_(s-1)
  int m;                      // frozen from here on
  __device__ __forceinline__ double operator()(int t) const {
    if (v) return v[t];
    return (double)p[t < m ? t : m];
  }
};
// the inc-th zero-padded difference at t (fruits/cache.py:8-13; the triangle of the selection)
template <int INC>
__device__ __forceinline__ double argmax_diff(const ArgmaxRow &r, int t) {
  double v[INC + 1];
#pragma unroll
  for (int j = 0; j <= INC; ++j) v[j] = t - j >= 0 ? r(t - j) : 0.0;
#pragma unroll
  for (int lvl = 1; lvl <= INC; ++lvl)
#pragma unroll
    for (int j = 0; j + lvl <= INC; ++j) v[j] = (t - j >= 1) ? v[j] - v[j + 1] : 0.0;
  return v[0];
}
// one band op over the row: the sum and the number of the elements t in [lo, hi) whose
// difference lies in (qlo, qhi]; the whole workgroup takes part, thread 0 gets the totals
template <int INC>
__device__ __forceinline__ void argmax_band(const ArgmaxRow &r, int lo, int hi, double qlo, double qhi,
                                            double *red, double &sum, double &cnt) {
  double s = 0.0, c = 0.0;
  for (int t = lo + (int)threadIdx.x; t < hi; t += (int)blockDim.x) {
    const double d = argmax_diff<INC>(r, t);
    if (qlo < d && d <= qhi) {
      s += d;
      c += 1.0;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    c += __shfl_xor(c, o);
  }
  __syncthreads();   // (red is reused op after op)
  if ((threadIdx.x & 63) == 0) {
    red[2 * (threadIdx.x >> 6)] = s;
    red[2 * (threadIdx.x >> 6) + 1] = c;
  }
  __syncthreads();
  sum = cnt = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
      sum += red[2 * w];
      cnt += red[2 * w + 1];
    }
}

__global__ __launch_bounds__(256) void argmax_sieve_kernel(
    const double *__restrict__ V, int64_t N, int64_t T, const ArgmaxWord *__restrict__ words,
    const FeatOp *__restrict__ ops, int n_ops, int n_ops_padded, double *__restrict__ feats,
    double *__restrict__ cnt, int64_t feat_stride, const int32_t *__restrict__ series_cuts,
    int cut_slots) {
  extern __shared__ double dyn_lds[];
  __shared__ double red[8];
  __shared__ int part[256];
  __shared__ int m[66];   // m_s for s = 1 .. k + 1 (words of <= 63 letters)
  const int64_t n = blockIdx.x;
  const ArgmaxWord w = words[blockIdx.y];
  const int Ti = (int)T, tid = (int)threadIdx.x;
  double *vrow = dyn_lds;
  unsigned short *pos = reinterpret_cast<unsigned short *>(dyn_lds + T);   // [L][T]
  const int32_t *cut_row = series_cuts ? series_cuts + n * cut_slots : nullptr;
  double *frow = feats + n * feat_stride, *crow = cnt + n * feat_stride;
  int out_row = w.out_row0;
  for (int k = 0; k < w.L; ++k) {
    // V_k into LDS; P_k[t] = the index at which the running maximum was last raised
    const double *v = V + ((int64_t)(w.v_row0 + k) * N + n) * T;
    __syncthreads();   // (the previous prefix's ops are done with vrow)
    for (int t = tid; t < Ti; t += 256) vrow[t] = v[t];
    __syncthreads();
    unsigned short *pk = pos + (int64_t)k * T;
    const int per = (Ti + 255) / 256, lo = tid * per, hi = lo + per < Ti ? lo + per : Ti;
    int best = 0;
    for (int t = lo > 0 ? lo : 1; t < hi; ++t)
      if (vrow[t] > vrow[t - 1]) best = t;
    part[tid] = best;
    __syncthreads();
    int run = 0;
    for (int i = 0; i < tid; ++i) run = part[i] > run ? part[i] : run;
    for (int t = lo; t < hi; ++t) {
      if (t > 0 && vrow[t] > vrow[t - 1]) run = t;
      pk[t] = (unsigned short)run;
    }
    __syncthreads();
    if (tid == 0) {
      m[k + 1] = Ti - 1;
      for (int s = k; s >= 1; --s) m[s] = pos[(int64_t)s * T + m[s + 1]];
    }
    __syncthreads();
    // the k + 2 rows of this prefix: V_k, then R_1 .. R_(k+1)
    for (int s = 0; s <= k + 1; ++s, ++out_row) {
      ArgmaxRow r{s == 0 ? vrow : nullptr, s == 0 ? nullptr : pos + (int64_t)(s - 1) * T, s == 0 ? 0 : m[s]};
      const FeatOp *row_ops = ops + (int64_t)out_row * n_ops_padded;
      for (int i = 0; i < n_ops; ++i) {
        const FeatOp op = row_ops[i];
        const int kind = op.kind_inc & 0xff, inc = (int)(int8_t)((op.kind_inc >> 8) & 0xff);
        const bool cuts = (op.kind_inc & (1 << 16)) != 0;
        if (kind == FR_SIEVE_END_K) {
          int pick = op.lo;
          if (cuts) {   // X[:, cut - 1], index -1 wrapping like numpy
            pick = cut_row[op.lo] - 1;
            if (pick < 0) pick += Ti;
          }
          if (tid == 0 && pick >= 0 && pick < Ti) frow[op.col] = r(pick);   // (else: a padding op)
          continue;
        }
        int lo_t = op.lo, hi_t = op.hi;
        if (cuts) {
          lo_t = cut_row[op.lo];
          hi_t = cut_row[op.hi];
        }
        lo_t = lo_t < 0 ? 0 : lo_t;
        hi_t = hi_t > Ti ? Ti : hi_t;
        double sum, c;
        if (inc == 0) argmax_band<0>(r, lo_t, hi_t, op.qlo, op.qhi, red, sum, c);
        else if (inc == 1) argmax_band<1>(r, lo_t, hi_t, op.qlo, op.qhi, red, sum, c);
        else argmax_band<2>(r, lo_t, hi_t, op.qlo, op.qhi, red, sum, c);
        if (tid == 0) {
          if (kind == FR_SIEVE_MPI_K) {
            frow[op.col] = sum;
            crow[op.col] = c;
          } else {
            frow[op.col] = c;
          }
        }
      }
    }
  }
}

size_t argmax_sieve_lds(int64_t T, int max_len) {
  return (size_t)T * 8 + (size_t)max_len * (size_t)T * 2 + 16;
}

hipError_t launch_argmax_sieves(const double *V, int64_t N, int64_t T, const void *words, int n_words,
                                int max_len, const FeatOp *ops, int n_ops, int n_ops_padded,
                                double *feats, double *cnt, int64_t feat_stride,
                                const int32_t *series_cuts, int cut_slots, hipStream_t st) {
  if (N <= 0 || T <= 0 || n_words <= 0) return hipSuccess;
  if (N > 0x7fffffffLL || n_words > 65535 || T > 65535 || max_len > 63) return hipErrorInvalidValue;
  const size_t lds = argmax_sieve_lds(T, max_len);
  if (lds > kArgmaxSieveLds) return hipErrorInvalidValue;
  hipLaunchKernelGGL(argmax_sieve_kernel, dim3((unsigned)N, (unsigned)n_words), dim3(256), lds, st, V, N, T,
                     static_cast<const ArgmaxWord *>(words), ops, n_ops, n_ops_padded, feats, cnt,
                     feat_stride, series_cuts, cut_slots);
  return hipGetLastError();
}

// ---------------------------------------------------------------- rank selection (fit)
// SegmentSieve._fit needs np.quantile of the pre-transformed fit sample
// (fruits/sieving/segment.py:66-75, increment.py:73-74).  np.quantile interpolates
// between two ORDER STATISTICS; those are found here exactly by a radix select over the
// order-preserving 64-bit image of the doubles (one job per wanted rank), so the
// (N_fit, T) rows never leave the device.  Three histogram passes fix the leading 24 bits
// (sign, exponent, 12 mantissa bits); the few elements that share them (<= kSelSmall,
// else the histogram passes simply go on) are gathered in ONE more pass over the data and
// the remaining 40 bits are settled inside a workgroup: 4 passes over the data instead of 9.
constexpr int kSelGroupJobs = 8;   // jobs of one group (they read the same (N, T) row block)
struct SelJob {
  const double *base;        // (N, T) row block of one iterated sum
  unsigned long long prefix; // key bits fixed so far
  long long k;               // rank among the elements that match the prefix
  int inc;
  int pad;
};

__device__ __forceinline__ unsigned long long order_key(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_to_double(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)u);
}

// The data passes of the selection (histogram, gather, successor) walk the differencing orders
// 0 .. MI of an element in ONE unrolled loop: the triangle of differences advances a level (D_k:
// k-th differences, zero-padded, see diff_at) and the jobs of that level - a group's jobs are
// sorted by order - look at its value.  MI is the launch's largest order (0 / 1 / 2, or kMaxInc for
// anything beyond).  Nothing is indexed by a run-time value: an array of the levels' keys picked
// by a job's order ends up in LDS (the compiler's promotion of private arrays), which is what
// the first version of these kernels spent its time on.
template <int MI>
__device__ __forceinline__ void element_load(const double *__restrict__ row, int t, double (&v)[MI + 1]) {
  v[0] = row[t];
#pragma unroll
  for (int j = 1; j <= MI; ++j) v[j] = t - j >= 0 ? row[t - j] : 0.0;
}
// level LVL - 1 -> LVL: afterwards v[0] = D_LVL[t]
template <int MI, int LVL>
__device__ __forceinline__ void next_level(int t, double (&v)[MI + 1]) {
#pragma unroll
  for (int j = 0; j + LVL <= MI; ++j) v[j] = (t - j >= 1) ? v[j] - v[j + 1] : 0.0;
}
// The element loop of the data passes: block b takes series b, b + grid, ... and its threads stride
// over the time axis (no per-element 64-bit division; few series: the time axis is split over the
// blocks).  A wave takes FOUR elements per lane at a time wherever all of them exist (f4; the rest
// one by one, f1): the loads of the four are in flight together, only the first of them can lie in
// the zero-padded head of a series (the others need no bounds tests), and whatever a pass reads
// per JOB - its prefix, its histogram row - is read once for the four.
constexpr int kSelUnroll = 4;
constexpr int kSelBlocks = 4096;
template <int MI>
__device__ __forceinline__ void element_load_inner(const double *__restrict__ row, int t, double (&v)[MI + 1]) {
#pragma unroll
  for (int j = 0; j <= MI; ++j) v[j] = row[t - j];
}
template <int MI, int LVL>
__device__ __forceinline__ void next_level_inner(double (&v)[MI + 1]) {
#pragma unroll
  for (int j = 0; j + LVL <= MI; ++j) v[j] = v[j] - v[j + 1];
}
// level LVL - 1 -> LVL of four elements; only element 0 can be one of a series' first MI
template <int MI, int LVL>
__device__ __forceinline__ void next_level4(const int (&t)[kSelUnroll], double (&v)[kSelUnroll][MI + 1]) {
  next_level<MI, LVL>(t[0], v[0]);
#pragma unroll
  for (int u = 1; u < kSelUnroll; ++u) next_level_inner<MI, LVL>(v[u]);
}
template <int MI, class F4, class F1>
__device__ __forceinline__ void for_elements(const double *__restrict__ base, int64_t N, int64_t T, F4 f4, F1 f1) {
  static_assert(MI < 64, "elements 1 .. 3 of a group of four lie behind the padded head");
  const int64_t per_series = (N >= (int64_t)gridDim.x) ? 1 : ((int64_t)gridDim.x + N - 1) / N;
  const int64_t n_first = (int64_t)blockIdx.x / per_series, part = (int64_t)blockIdx.x % per_series;
  const int64_t n_step = ((int64_t)gridDim.x + per_series - 1) / per_series;
  const int64_t t_len = (T + per_series - 1) / per_series;
  const int t_lo = (int)(part * t_len), t_hi = (int)((part * t_len + t_len < T) ? part * t_len + t_len : T);
  const int step = (int)blockDim.x;
  const int wave_last = (int)(threadIdx.x | 63u);   // the wave's last lane
  for (int64_t n = n_first; n < N; n += n_step) {
    const double *__restrict__ row = base + n * T;
    for (int tb = t_lo; tb < t_hi; tb += step * kSelUnroll) {
      const int t0 = tb + (int)threadIdx.x;
      if (tb + (kSelUnroll - 1) * step + wave_last < t_hi) {   // (uniform in the wave)
        int t[kSelUnroll];
        double v[kSelUnroll][MI + 1];
#pragma unroll
        for (int u = 0; u < kSelUnroll; ++u) t[u] = t0 + u * step;
        element_load<MI>(row, t[0], v[0]);
#pragma unroll
        for (int u = 1; u < kSelUnroll; ++u) element_load_inner<MI>(row, t[u], v[u]);
        f4(t, v);
      } else {
        for (int t = t0; t < t_hi; t += step) {
          double v[MI + 1];
          element_load<MI>(row, t, v);
          f1(t, v);
        }
      }
    }
  }
}

// The leading 32 bits of the order-preserving key: all that the first three digits and the bucket
// tests of the gather pass look at (32-bit operations instead of 64-bit shifts and compares).
__device__ __forceinline__ unsigned int order_key_hi(double v) {
  const unsigned int h = (unsigned int)((unsigned long long)__double_as_longlong(v) >> 32);
  return (h >> 31) ? ~h : (h | 0x80000000u);
}

// A group's jobs in LDS, once per workgroup and pass: the descriptors, and the jobs that take
// part in this pass compacted by differencing order (level i: act[lvl[i]] .. act[lvl[i + 1])).
constexpr int kSelTrack = kSelTrackJobs;   // (kernels.h: the host flags the jobs)
struct SelGroup {
  unsigned long long prefix[kSelGroupJobs];
  int inc[kSelGroupJobs], pad[kSelGroupJobs];
  int act[kSelGroupJobs];
  unsigned int act_hi[kSelGroupJobs];   // leading dword of the job's prefix
  int lvl[kMaxInc + 2];
};
// `take(j)`: does job j take part in this pass?  Returns the number of jobs that do.
template <int MI, class F>
__device__ __forceinline__ int load_group(SelGroup &g, const SelJob *__restrict__ jobs, int jb, int nj,
                                          F take) {
  if ((int)threadIdx.x < nj) {
    g.prefix[threadIdx.x] = jobs[jb + threadIdx.x].prefix;
    g.inc[threadIdx.x] = jobs[jb + threadIdx.x].inc;
    g.pad[threadIdx.x] = jobs[jb + threadIdx.x].pad;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int n = 0;
    for (int i = 0; i <= MI; ++i) {
      g.lvl[i] = n;
      for (int j = 0; j < nj; ++j)
        if ((g.inc[j] == i || (i == MI && g.inc[j] > MI)) && take(j)) {
          g.act[n] = j;
          g.act_hi[n] = (unsigned int)(g.prefix[j] >> 32);
          ++n;
        }
    }
    g.lvl[MI + 1] = n;
  }
  __syncthreads();
  return g.lvl[MI + 1];
}

// SelJob::pad: bit 0 the caller also wants the NEXT order statistic; bit 1 that one lies
// outside what this job has seen (select_succ_kernel finds it); bit 2 the candidates that
// share the job's leading 24 bits fit a workgroup (their number in pad >> 8): no further
// histogram passes, select_gather_kernel + select_small_kernel finish the job; bit 3 more
// candidates than that (heavy ties); bit 4 (set by the host for the first kSelTrack jobs of a
// group and differencing order that want the next statistic) the gather pass leaves the smallest
// key above the job's bucket in succ[]
constexpr int kSelSmall = kSelSmallCap;   // (the host sizes the candidate lists with it)
constexpr int kSelSmallShift = 40;   // bits below this are settled among the gathered candidates

// One block column per GROUP of jobs that read the same (N, T) row block (the ranks and
// differencing orders one iterated sum is asked for): every element is loaded once per
// pass for all of them.
template <int MI, int LVL>
__device__ __forceinline__ void hist_level(const SelGroup &g, unsigned int (*lh)[256], int t,
                                           double (&v)[MI + 1], int shift) {
  if constexpr (LVL <= MI) {
    if constexpr (LVL > 0) next_level<MI, LVL>(t, v);
    const int kb = __builtin_amdgcn_readfirstlane(g.lvl[LVL]), ke = __builtin_amdgcn_readfirstlane(g.lvl[LVL + 1]);
    if (kb != ke) {
      const unsigned int kh = order_key_hi(v[0]);
      const unsigned long long key = order_key(v[0]);
      for (int k = kb; k < ke; ++k) {
        const int job = g.act[k];
        bool match;
        unsigned int bin;
        if (shift >= 32) {   // (uniform) a digit of the leading dword
          match = shift == 56 || (kh >> (shift - 24)) == (g.act_hi[k] >> (shift - 24));
          bin = (kh >> (shift - 32)) & 255u;
        } else {
          match = (key >> (shift + 8)) == (g.prefix[job] >> (shift + 8));
          bin = (unsigned int)(key >> shift) & 255u;
        }
        // The leading digits (sign, exponent, high mantissa bits) are shared by almost all
        // elements: 64 lanes adding to ONE LDS counter serialise.  The lanes that hold the
        // first matching lane's digit are counted together - all of them in the usual case -
        // and in the first two digits the others add one by one (both signs of an increment);
        // later digits are spread out: there the split costs more than it saves.
        const unsigned long long m = __ballot(match);
        if (m == 0) continue;
        const int leader = __ffsll((long long)m) - 1;
        const unsigned int lead_bin = (unsigned int)__builtin_amdgcn_readlane((int)bin, leader);
        const bool same = match && bin == lead_bin;
        const unsigned long long ms = __ballot(same);
        bool todo = match;
        if (ms == m || shift >= 48) {
          if ((int)(threadIdx.x & 63) == leader) atomicAdd(&lh[job][lead_bin], (unsigned int)__popcll(ms));
          todo = match && !same;
        }
        if (todo) atomicAdd(&lh[job][bin], 1u);
      }
    }
    hist_level<MI, LVL + 1>(g, lh, t, v, shift);
  }
}

// The same for a group of four elements (for_elements) and a digit known at compile time (the
// four digits of the leading dword: every pass of a usual fit): a job's row and prefix are read
// once for the four, the digit and the prefix test are immediates.
template <int SHIFT>
__device__ __forceinline__ void hist_count(unsigned int *__restrict__ row, unsigned int kh, unsigned int ph) {
  static_assert(SHIFT >= 32 && SHIFT <= 56, "a digit of the leading dword");
  const bool match = SHIFT == 56 || ((kh ^ ph) >> (SHIFT - 24)) == 0u;
  const unsigned int bin = (kh >> (SHIFT - 32)) & 255u;
  unsigned long long m = __ballot(match);
  if (m == 0) return;
  bool todo = match;
  // the first digit (sign, seven exponent bits) has two to four values in a wave: two of them are
  // counted lane group by lane group (one: 2.26 ms for 64 groups, two: 2.14, three: 2.19)
  constexpr int kRounds = SHIFT == 56 ? 2 : 1;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int leader = __ffsll((long long)m) - 1;
    const unsigned int lead_bin = (unsigned int)__builtin_amdgcn_readlane((int)bin, leader);
    const bool same = todo && bin == lead_bin;
    const unsigned long long ms = __ballot(same);
    // (the first digit alone counts a partial lane group: in the second one - four exponent and
    // four mantissa bits - the rest of the wave is spread out already, 1.81 -> 1.74 ms)
    if (SHIFT >= 56 || ms == m) {
      if ((int)(threadIdx.x & 63) == leader) atomicAdd(&row[lead_bin], (unsigned int)__popcll(ms));
      todo = todo && !same;
      m &= ~ms;
    }
    if (m == 0) return;
  }
  if (todo) atomicAdd(&row[bin], 1u);
}
template <int MI, int SHIFT, int LVL>
__device__ __forceinline__ void hist_level4(const SelGroup &g, unsigned int (*lh)[256],
                                            const int (&t)[kSelUnroll], double (&v)[kSelUnroll][MI + 1]) {
  if constexpr (LVL <= MI) {
    if constexpr (LVL > 0) next_level4<MI, LVL>(t, v);
    const int kb = __builtin_amdgcn_readfirstlane(g.lvl[LVL]), ke = __builtin_amdgcn_readfirstlane(g.lvl[LVL + 1]);
    if (kb != ke) {
      unsigned int kh[kSelUnroll];
#pragma unroll
      for (int u = 0; u < kSelUnroll; ++u) kh[u] = order_key_hi(v[u][0]);
      for (int k = kb; k < ke; ++k) {
        unsigned int *row = lh[__builtin_amdgcn_readfirstlane(g.act[k])];
        const unsigned int ph = (unsigned int)__builtin_amdgcn_readfirstlane((int)g.act_hi[k]);
#pragma unroll
        for (int u = 0; u < kSelUnroll; ++u) hist_count<SHIFT>(row, kh[u], ph);
      }
    }
    hist_level4<MI, SHIFT, LVL + 1>(g, lh, t, v);
  }
}

// The FIRST digit (sign, seven exponent bits): every element takes part and a block sees a handful of
// values - as LDS adds they collide (the pass was bound by LDS conflicts: 1.9 TB/s against the third
// digit's 3.9).  A thread counts the digit's values it meets in kBinPairs register pairs per
// differencing order and only a further value evicts one to LDS; the pairs are added to the block's
// histogram once, at the end.
constexpr unsigned int kBinNone = 0xffffffffu;
// (pairs per differencing order, Fruit.fit of fruit_reduced on one box: none 19.3 ms, one 20.0, two 17.9,
// three 18.3, four 18.5, eight 18.8 - the compares are paid per element, two pairs hold the two signs)
constexpr int kBinPairs = 2;
struct BinCache {
  unsigned int bin[kBinPairs], cnt[kBinPairs];
};
__device__ __forceinline__ void bin_cache_add(BinCache &c, unsigned int b, unsigned int *__restrict__ row) {
  bool hit = false;
#pragma unroll
  for (int i = 0; i < kBinPairs; ++i) {
    const bool h = b == c.bin[i];
    c.cnt[i] += h ? 1u : 0u;
    hit = hit || h;
  }
  if (!hit) {   // (rare: a free pair, else the last one goes to LDS)
    bool placed = false;
#pragma unroll
    for (int i = 0; i + 1 < kBinPairs; ++i)
      if (!placed && c.bin[i] == kBinNone) {
        c.bin[i] = b;
        c.cnt[i] = 1u;
        placed = true;
      }
    if (!placed) {
      if (c.bin[kBinPairs - 1] != kBinNone) atomicAdd(&row[c.bin[kBinPairs - 1]], c.cnt[kBinPairs - 1]);
      c.bin[kBinPairs - 1] = b;
      c.cnt[kBinPairs - 1] = 1u;
    }
  }
}
template <int MI, int LVL>
__device__ __forceinline__ void hist_first4(const SelGroup &g, unsigned int (*lh)[256], BinCache (&bc)[MI + 1],
                                            const int (&t)[kSelUnroll], double (&v)[kSelUnroll][MI + 1]) {
  if constexpr (LVL <= MI) {
    if constexpr (LVL > 0) next_level4<MI, LVL>(t, v);
    const int kb = __builtin_amdgcn_readfirstlane(g.lvl[LVL]), ke = __builtin_amdgcn_readfirstlane(g.lvl[LVL + 1]);
    if (kb != ke) {   // (one job per order takes part in the first pass: the order's histogram)
      unsigned int *row = lh[__builtin_amdgcn_readfirstlane(g.act[kb])];
#pragma unroll
      for (int u = 0; u < kSelUnroll; ++u) bin_cache_add(bc[LVL], order_key_hi(v[u][0]) >> 24, row);
    }
    hist_first4<MI, LVL + 1>(g, lh, bc, t, v);
  }
}
template <int MI, int LVL>
__device__ __forceinline__ void hist_first_flush(const SelGroup &g, unsigned int (*lh)[256],
                                                 const BinCache (&bc)[MI + 1]) {
  if constexpr (LVL <= MI) {
    if (g.lvl[LVL] != g.lvl[LVL + 1]) {
      unsigned int *row = lh[g.act[g.lvl[LVL]]];
#pragma unroll
      for (int i = 0; i < kBinPairs; ++i)
        if (bc[LVL].bin[i] != kBinNone) atomicAdd(&row[bc[LVL].bin[i]], bc[LVL].cnt[i]);
    }
    hist_first_flush<MI, LVL + 1>(g, lh, bc);
  }
}

// SHIFT: the digit when it is one of the leading dword's (the four-wide path), else 0 - then the
// run-time `shift` counts (the low digits: jobs with heavy ties only)
template <int MI, int SHIFT>
__global__ __launch_bounds__(256) void select_hist_kernel(const SelJob *__restrict__ jobs,
                                                           const int2 *__restrict__ groups,
                                                           int64_t N, int64_t T, int shift,
                                                           unsigned int *__restrict__ hist) {
  __shared__ unsigned int lh[kSelGroupJobs][256];
  __shared__ SelGroup g;
  const int jb = groups[blockIdx.y].x, nj = groups[blockIdx.y].y;
  // jobs that finish among their gathered candidates take no part in later passes; first pass:
  // no prefix yet - jobs of one differencing order see the same histogram, which is counted
  // once and copied below
  if constexpr (SHIFT != 0) shift = SHIFT;
  const int n_act = load_group<MI>(g, jobs, jb, nj, [&](int j) {
    return !(g.pad[j] & 4) && !(shift == 56 && j > 0 && g.inc[j] == g.inc[j - 1]);
  });
  if (n_act == 0) return;
  for (int j = 0; j < nj; ++j) lh[j][threadIdx.x] = 0;
  __syncthreads();
  if constexpr (SHIFT == 56 && MI <= 2) {
    BinCache bc[MI + 1];
#pragma unroll
    for (int l = 0; l <= MI; ++l)
#pragma unroll
      for (int i = 0; i < kBinPairs; ++i) {
        bc[l].bin[i] = kBinNone;
        bc[l].cnt[i] = 0u;
      }
    for_elements<MI>(
        jobs[jb].base, N, T,
        [&](const int (&t)[kSelUnroll], double (&v)[kSelUnroll][MI + 1]) { hist_first4<MI, 0>(g, lh, bc, t, v); },
        [&](int t, double (&v)[MI + 1]) { hist_level<MI, 0>(g, lh, t, v, SHIFT); });
    hist_first_flush<MI, 0>(g, lh, bc);
  } else if constexpr (SHIFT != 0) {
    for_elements<MI>(
        jobs[jb].base, N, T,
        [&](const int (&t)[kSelUnroll], double (&v)[kSelUnroll][MI + 1]) { hist_level4<MI, SHIFT, 0>(g, lh, t, v); },
        [&](int t, double (&v)[MI + 1]) { hist_level<MI, 0>(g, lh, t, v, SHIFT); });
  } else {
    const auto one = [&](int t, double (&v)[MI + 1]) { hist_level<MI, 0>(g, lh, t, v, shift); };
    for_elements<MI>(jobs[jb].base, N, T,
                     [&](const int (&t)[kSelUnroll], double (&v)[kSelUnroll][MI + 1]) {
#pragma unroll
                       for (int u = 0; u < kSelUnroll; ++u) one(t[u], v[u]);
                     },
                     one);
  }
  __syncthreads();
  for (int j = 0; j < nj; ++j) {
    if (g.pad[j] & 4) continue;
    int src = j;   // first pass: the histogram of the first job of this differencing order
    if (shift == 56)
      while (src > 0 && g.inc[src - 1] == g.inc[j]) --src;
    if (lh[src][threadIdx.x]) atomicAdd(&hist[(jb + j) * 256 + threadIdx.x], lh[src][threadIdx.x]);
  }
}

__global__ void select_pick_kernel(SelJob *__restrict__ jobs, int shift,
                                   unsigned int *__restrict__ hist, double *__restrict__ out,
                                   unsigned long long *__restrict__ succ,
                                   unsigned int *__restrict__ n_big,
                                   unsigned long long *__restrict__ cand,
                                   unsigned int *__restrict__ cnt) {
  const int job = blockIdx.x;
  if (jobs[job].pad & 4) return;   // (its histogram received nothing)
  if (threadIdx.x == 0) {
    long long k = jobs[job].k, run = 0;
    int d = 0;
    for (; d < 255; ++d) {
      const long long c = hist[job * 256 + d];
      if (k < run + c) break;
      run += c;
    }
    jobs[job].k = k - run;
    jobs[job].prefix |= (unsigned long long)d << shift;
    if (shift == kSelSmallShift) {
      if (hist[job * 256 + d] <= (unsigned int)kSelSmall)
        jobs[job].pad |= 4 | ((int)hist[job * 256 + d] << 8);
      else {
        // too many candidates for a workgroup - usually ONE value many times (the zeros among
        // the increments of a running maximum): select_gather_kernel checks whether they are
        // all equal (cand[0] = the first one seen, cand[1] != 0 = another one exists)
        jobs[job].pad |= 8;
        cnt[job] = hist[job * 256 + d];
        cand[(int64_t)job * kSelSmall] = ~0ull;
        cand[(int64_t)job * kSelSmall + 1] = 0ull;
        atomicAdd(n_big, 1u);   // a job that stays in the histogram passes (unless resolved)
      }
    }
    if (shift == 0) {
      out[job] = key_to_double(jobs[job].prefix);
      // the NEXT order statistic (np.quantile interpolates between two neighbours): the same
      // value when more copies of it remain, else the smallest larger element (one more pass,
      // select_succ_kernel) - instead of a second 8-pass selection
      if (jobs[job].pad & 1) {
        if (k - run + 1 < (long long)hist[job * 256 + d])
          succ[job] = jobs[job].prefix;
        else
          jobs[job].pad |= 2;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[job * 256 + i] = 0;
}

// One pass over the data: the keys that share a small job's leading bits go to its
// candidate list (cand[job][0 .. kSelSmall), filled through cnt[job]).
//
// The smallest key ABOVE a job's bucket is the next order statistic when the selected one is the
// bucket's largest - the usual case for the median of increments, which lies near zero where the
// 24-bit buckets of floating-point numbers hold an element or two.  It is found here, in the
// pass that reads everything anyway (jobs flagged with pad bit 4; per thread a running minimum
// in registers, `above`), instead of in a pass of its own (select_succ_kernel).
struct GatherBig {   // up to two jobs of the group whose candidates may all be equal
  int big0, big1;
  unsigned long long ref0, ref1;   // the first candidate anybody saw
  bool other0, other1;
};
// one element's key against job j of the group (hit: it lies in the job's bucket)
__device__ __forceinline__ void gather_job(int jb, int j, bool hit, unsigned long long key, GatherBig &gb,
                                           unsigned long long *__restrict__ cand,
                                           unsigned int *__restrict__ cnt) {
  if (j == gb.big0 || j == gb.big1) {
    unsigned long long &ref = j == gb.big0 ? gb.ref0 : gb.ref1;
    // nobody has published a candidate yet: ONE lane of the wave tries (a compare-and-swap
    // per thread on one address would serialise a hundred thousand of them) and tells
    // the others what the reference is
    const unsigned long long ask = __ballot(hit && ref == ~0ull);
    if (ask != 0) {
      const int leader = __ffsll((long long)ask) - 1;
      unsigned long long got = 0;
      if ((int)(threadIdx.x & 63) == leader) {
        const unsigned long long old = atomicCAS(&cand[(int64_t)(jb + j) * kSelSmall], ~0ull, key);
        got = old == ~0ull ? key : old;
      }
      const unsigned long long told = __shfl(got, leader);
      if (ref == ~0ull) ref = told;
    }
    if (hit && key != ref) (j == gb.big0 ? gb.other0 : gb.other1) = true;
  } else if (hit) {
    const unsigned int slot = atomicAdd(&cnt[jb + j], 1u);
    if (slot < (unsigned int)kSelSmall) cand[(int64_t)(jb + j) * kSelSmall + slot] = key;
  }
}
template <int MI, int LVL>
__device__ __forceinline__ void gather_level(const SelGroup &g, int jb, int t, double (&v)[MI + 1],
                                             unsigned long long (&above)[MI + 1][kSelTrack],
                                             const unsigned int (&track)[MI + 1][kSelTrack],
                                             GatherBig &gb, unsigned long long *__restrict__ cand,
                                             unsigned int *__restrict__ cnt) {
  if constexpr (LVL <= MI) {
    if constexpr (LVL > 0) next_level<MI, LVL>(t, v);
    const int kb = __builtin_amdgcn_readfirstlane(g.lvl[LVL]), ke = __builtin_amdgcn_readfirstlane(g.lvl[LVL + 1]);
    if (kb != ke) {
      // the bucket (leading 24 bits) of this element
      const unsigned int bucket = order_key_hi(v[0]) >> (kSelSmallShift - 32);
      const unsigned long long key = order_key(v[0]);
#pragma unroll
      for (int a = 0; a < kSelTrack; ++a)   // (the level's first jobs: the host flags only those)
        if (bucket > track[LVL][a] && key < above[LVL][a]) above[LVL][a] = key;
      for (int k = kb; k < ke; ++k)
        gather_job(jb, g.act[k], bucket == (g.act_hi[k] >> (kSelSmallShift - 32)), key, gb, cand, cnt);
    }
    gather_level<MI, LVL + 1>(g, jb, t, v, above, track, gb, cand, cnt);
  }
}
// ... of a group of four elements (for_elements): a job's bucket is read once for the four, and a
// job none of the wave's 256 elements falls to - nearly every job, nearly every time: a bucket holds
// at most kSelSmall of the millions - costs four compares and a branch
template <int MI, int LVL>
__device__ __forceinline__ void gather_level4(const SelGroup &g, int jb, const int (&t)[kSelUnroll],
                                              double (&v)[kSelUnroll][MI + 1],
                                              unsigned long long (&above)[MI + 1][kSelTrack],
                                              const unsigned int (&track)[MI + 1][kSelTrack],
                                              GatherBig &gb, unsigned long long *__restrict__ cand,
                                              unsigned int *__restrict__ cnt) {
  if constexpr (LVL <= MI) {
    if constexpr (LVL > 0) next_level4<MI, LVL>(t, v);
    const int kb = __builtin_amdgcn_readfirstlane(g.lvl[LVL]), ke = __builtin_amdgcn_readfirstlane(g.lvl[LVL + 1]);
    if (kb != ke) {
      unsigned int bucket[kSelUnroll];
      unsigned long long key[kSelUnroll];
#pragma unroll
      for (int u = 0; u < kSelUnroll; ++u) {
        bucket[u] = order_key_hi(v[u][0]) >> (kSelSmallShift - 32);
        key[u] = order_key(v[u][0]);
      }
#pragma unroll
      for (int a = 0; a < kSelTrack; ++a) {
        const unsigned int tr = (unsigned int)__builtin_amdgcn_readfirstlane((int)track[LVL][a]);
        if (tr == ~0u) continue;   // (nothing is tracked in this place)
#pragma unroll
        for (int u = 0; u < kSelUnroll; ++u)
          if (bucket[u] > tr && key[u] < above[LVL][a]) above[LVL][a] = key[u];
      }
      for (int k = kb; k < ke; ++k) {
        const int j = __builtin_amdgcn_readfirstlane(g.act[k]);
        const unsigned int jbucket =
            (unsigned int)__builtin_amdgcn_readfirstlane((int)g.act_hi[k]) >> (kSelSmallShift - 32);
        bool any = false;
#pragma unroll
        for (int u = 0; u < kSelUnroll; ++u) any = any || bucket[u] == jbucket;
        if (__ballot(any) == 0) continue;
#pragma unroll
        for (int u = 0; u < kSelUnroll; ++u) gather_job(jb, j, bucket[u] == jbucket, key[u], gb, cand, cnt);
      }
    }
    gather_level4<MI, LVL + 1>(g, jb, t, v, above, track, gb, cand, cnt);
  }
}
template <int MI, int LVL>
__device__ __forceinline__ void gather_publish(const SelGroup &g, int jb,
                                               const unsigned long long (&above)[MI + 1][kSelTrack],
                                               unsigned long long *__restrict__ succ) {
  if constexpr (LVL <= MI) {
    const int kb = g.lvl[LVL], ke = g.lvl[LVL + 1];
#pragma unroll
    for (int a = 0; a < kSelTrack; ++a) {
      if (kb + a < ke && (g.pad[g.act[kb + a]] & 16)) {
        unsigned long long b = above[LVL][a];
        for (int o = 32; o > 0; o >>= 1) {
          const unsigned long long w = __shfl_xor(b, o);
          b = w < b ? w : b;
        }
        if ((threadIdx.x & 63) == 0 && b != ~0ull) atomicMin(&succ[jb + g.act[kb + a]], b);
      }
    }
    gather_publish<MI, LVL + 1>(g, jb, above, succ);
  }
}

template <int MI>
__global__ __launch_bounds__(256) void select_gather_kernel(const SelJob *__restrict__ jobs,
                                                             const int2 *__restrict__ groups,
                                                             int64_t N, int64_t T,
                                                             unsigned long long *__restrict__ cand,
                                                             unsigned int *__restrict__ cnt,
                                                             unsigned long long *__restrict__ succ) {
  static_assert(kSelSmallShift >= 32, "the bucket test reads the leading dword of a key");
  __shared__ SelGroup g;
  const int jb = groups[blockIdx.y].x, nj = groups[blockIdx.y].y;
  if (load_group<MI>(g, jobs, jb, nj, [&](int j) { return (g.pad[j] & 12) != 0; }) == 0) return;
  GatherBig gb{-1, -1, ~0ull, ~0ull, false, false};
  for (int j = 0; j < nj; ++j) {
    if (g.pad[j] & 8) {
      if (gb.big0 < 0) gb.big0 = j;
      else if (gb.big1 < 0) gb.big1 = j;
    }
  }
  if (gb.big0 >= 0) gb.ref0 = cand[(int64_t)(jb + gb.big0) * kSelSmall];
  if (gb.big1 >= 0) gb.ref1 = cand[(int64_t)(jb + gb.big1) * kSelSmall];
  unsigned long long above[MI + 1][kSelTrack];
  unsigned int track[MI + 1][kSelTrack];   // bucket of a tracked job (else: nothing lies above it)
#pragma unroll
  for (int i = 0; i <= MI; ++i)
#pragma unroll
    for (int a = 0; a < kSelTrack; ++a) {
      above[i][a] = ~0ull;
      const int k = g.lvl[i] + a;
      const bool on = k < g.lvl[i + 1] && (g.pad[g.act[k < kSelGroupJobs ? k : 0]] & 16);
      track[i][a] = on ? g.act_hi[k < kSelGroupJobs ? k : 0] >> (kSelSmallShift - 32) : ~0u;
    }
  for_elements<MI>(
      jobs[jb].base, N, T,
      [&](const int (&t)[kSelUnroll], double (&v)[kSelUnroll][MI + 1]) {
        gather_level4<MI, 0>(g, jb, t, v, above, track, gb, cand, cnt);
      },
      [&](int t, double (&v)[MI + 1]) { gather_level<MI, 0>(g, jb, t, v, above, track, gb, cand, cnt); });
  if (gb.other0) cand[(int64_t)(jb + gb.big0) * kSelSmall + 1] = 1ull;
  if (gb.other1) cand[(int64_t)(jb + gb.big1) * kSelSmall + 1] = 1ull;
  gather_publish<MI, 0>(g, jb, above, succ);
}

// One workgroup per small job: the k-th smallest of its candidates (and the next one) by
// counting - every thread ranks its candidates against all of them in LDS.
__global__ __launch_bounds__(256) void select_small_kernel(SelJob *__restrict__ jobs,
                                                            const unsigned long long *__restrict__ cand,
                                                            const unsigned int *__restrict__ cnt,
                                                            double *__restrict__ out,
                                                            unsigned long long *__restrict__ succ,
                                                            unsigned int *__restrict__ n_big) {
  __shared__ unsigned long long next_key, s_prefix;
  __shared__ unsigned int lh[256];
  __shared__ int s_k, s_eq;
  const int job = blockIdx.x;
  if (jobs[job].pad & 8) {
    // more candidates than a workgroup settles: done all the same when they are ONE value
    if (threadIdx.x == 0) {
      const unsigned long long key = cand[(int64_t)job * kSelSmall];
      if (key != ~0ull && cand[(int64_t)job * kSelSmall + 1] == 0ull) {
        out[job] = key_to_double(key);
        jobs[job].prefix = key;
        // (the next one: another copy, else the smallest key above the bucket - already in
        // succ[job], select_gather_kernel)
        if (jobs[job].pad & 1) {
          if (jobs[job].k + 1 < (long long)cnt[job]) succ[job] = key;
          else if (!(jobs[job].pad & 16)) jobs[job].pad |= 2;
        }
        jobs[job].pad |= 4;
        atomicSub(n_big, 1u);
      }
    }
    return;
  }
  if (!(jobs[job].pad & 4)) return;
  int n = (int)cnt[job];
  if (n > kSelSmall) n = kSelSmall;   // (cannot happen: the histogram counted the same elements)
  // (the candidates stay where the gather pass put them: five sweeps over a list that is in L2)
  const unsigned long long *__restrict__ keys = cand + (int64_t)job * kSelSmall;
  if (threadIdx.x == 0) {
    next_key = ~0ull;
    s_prefix = jobs[job].prefix;   // (the leading 24 bits: every candidate has them)
    s_k = (int)jobs[job].k;
  }
  // the remaining five digits by the same radix selection, inside LDS (ranking every candidate
  // against all the others - 4 million compares for a full list - took as long as a pass over
  // the data)
  for (int shift = kSelSmallShift - 8; shift >= 0; shift -= 8) {
    lh[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long prefix = s_prefix;
    const int k = s_k;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
      if ((keys[i] >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&lh[(keys[i] >> shift) & 255u], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {   // lane l: bins 4 l .. 4 l + 3
      const int l = threadIdx.x;
      const unsigned int c0 = lh[4 * l], c1 = lh[4 * l + 1], c2 = lh[4 * l + 2], c3 = lh[4 * l + 3];
      const unsigned int mine = c0 + c1 + c2 + c3;
      unsigned int incl = mine;
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned int w = __shfl_up(incl, o);
        if (l >= o) incl += w;
      }
      const unsigned int excl = incl - mine;
      if ((unsigned int)k >= excl && (unsigned int)k < incl) {   // (one lane: k < the number counted)
        unsigned int r = (unsigned int)k - excl;
        int d = 4 * l;
        unsigned int c = c0;
        if (r >= c0) { r -= c0; ++d; c = c1;
          if (r >= c1) { r -= c1; ++d; c = c2;
            if (r >= c2) { r -= c2; ++d; c = c3; } } }
        s_k = (int)r;
        s_eq = (int)c;
        s_prefix = prefix | ((unsigned long long)d << shift);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const unsigned long long mine = s_prefix;   // the k-th key; s_eq copies of it, s_k of them in front
    out[job] = key_to_double(mine);
    jobs[job].prefix = mine;
    if (jobs[job].pad & 1) {
      if (s_k + 1 < s_eq) succ[job] = mine;
      else next_key = ~0ull - 1;   // marks: look for the smallest larger key
    }
  }
  __syncthreads();
  if (!(jobs[job].pad & 1) || next_key == ~0ull) return;
  // the next order statistic is the smallest candidate above the selected key - or, when the
  // selected key is the largest candidate, the smallest key above the bucket
  const unsigned long long sel = s_prefix;
  unsigned long long best = ~0ull;
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    if (keys[i] > sel && keys[i] < best) best = keys[i];
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long w = __shfl_xor(best, o);
    best = w < best ? w : best;
  }
  __syncthreads();
  if (threadIdx.x == 0) next_key = ~0ull;
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && best != ~0ull) atomicMin(&next_key, best);
  __syncthreads();
  // (none: the selected key is the bucket's largest; succ[job] already holds the smallest key
  // above the bucket when the gather pass tracked it - pad bit 4 - else a pass of its own finds it)
  if (threadIdx.x == 0) {
    if (next_key != ~0ull) succ[job] = next_key;
    else if (!(jobs[job].pad & 16)) jobs[job].pad |= 2;
  }
}

// smallest key above the selected one, for the jobs flagged pad & 2 (select_pick_kernel at the
// last digit; select_small_kernel for jobs whose successor the gather pass did not track);
// succ[] starts at the largest key
template <int MI, int LVL>
__device__ __forceinline__ void succ_level(const SelGroup &g, int jb, int t, double (&v)[MI + 1],
                                           unsigned long long (&best)[MI + 1],
                                           unsigned long long *__restrict__ succ) {
  if constexpr (LVL <= MI) {
    if constexpr (LVL > 0) next_level<MI, LVL>(t, v);
    const int kb = __builtin_amdgcn_readfirstlane(g.lvl[LVL]), ke = __builtin_amdgcn_readfirstlane(g.lvl[LVL + 1]);
    if (kb != ke) {
      const unsigned long long key = order_key(v[0]);
      // the level's first job: a running minimum in a register, published at the end
      if (key > g.prefix[g.act[kb]] && key < best[LVL]) best[LVL] = key;
      for (int k = kb + 1; k < ke; ++k) {
        // (further jobs of a level are rare: the wave's smallest candidate straight to memory)
        unsigned long long b = key > g.prefix[g.act[k]] ? key : ~0ull;
        if (__ballot(b != ~0ull) == 0) continue;
        for (int o = 32; o > 0; o >>= 1) {
          const unsigned long long w = __shfl_xor(b, o);
          b = w < b ? w : b;
        }
        if ((threadIdx.x & 63) == 0 && b < succ[jb + g.act[k]]) atomicMin(&succ[jb + g.act[k]], b);
      }
    }
    succ_level<MI, LVL + 1>(g, jb, t, v, best, succ);
  }
}
template <int MI, int LVL>
__device__ __forceinline__ void succ_publish(const SelGroup &g, int jb, const unsigned long long (&best)[MI + 1],
                                             unsigned long long *__restrict__ succ) {
  if constexpr (LVL <= MI) {
    if (g.lvl[LVL] != g.lvl[LVL + 1]) {
      unsigned long long b = best[LVL];
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long w = __shfl_xor(b, o);
        b = w < b ? w : b;
      }
      if ((threadIdx.x & 63) == 0 && b != ~0ull) atomicMin(&succ[jb + g.act[g.lvl[LVL]]], b);
    }
    succ_publish<MI, LVL + 1>(g, jb, best, succ);
  }
}
template <int MI>
__global__ __launch_bounds__(256) void select_succ_kernel(const SelJob *__restrict__ jobs,
                                                           const int2 *__restrict__ groups,
                                                           int64_t N, int64_t T,
                                                           unsigned long long *succ) {
  __shared__ SelGroup g;
  const int jb = groups[blockIdx.y].x, nj = groups[blockIdx.y].y;
  if (load_group<MI>(g, jobs, jb, nj, [&](int j) { return (g.pad[j] & 2) != 0; }) == 0) return;
  unsigned long long best[MI + 1];
#pragma unroll
  for (int i = 0; i <= MI; ++i) best[i] = ~0ull;
  const auto one = [&](int t, double (&v)[MI + 1]) { succ_level<MI, 0>(g, jb, t, v, best, succ); };
  for_elements<MI>(jobs[jb].base, N, T,
                   [&](const int (&t)[kSelUnroll], double (&v)[kSelUnroll][MI + 1]) {
#pragma unroll
                     for (int u = 0; u < kSelUnroll; ++u) one(t[u], v[u]);
                   },
                   one);
  succ_publish<MI, 0>(g, jb, best, succ);
}

// MI: the largest differencing order of the launch's jobs (kernels are compiled for 0, 1, 2 and
// kMaxInc)
template <int MI>
static hipError_t select_ranks_mi(SelJob *jb, int n_jobs, const int2 *gr, int n_groups,
                                  const int32_t *h_groups, int2 *gr_active, bool untracked, int64_t N, int64_t T,
                                  unsigned int *hist, double *out, unsigned long long *succ,
                                  unsigned long long *cand, unsigned int *cand_count, hipStream_t st) {
  // blocks per group: about kSelBlocks in all (16 per CU) - a block zeroes and publishes its
  // histograms whatever it counts (64 groups: 32768 blocks 2.62 ms, 8192 2.33, 4096 2.24, 2048 2.30)
  int64_t bpj = (N * T + 256 * 16 - 1) / (256 * 16);
  if (bpj > 512) bpj = 512;
  if (bpj * n_groups > kSelBlocks) bpj = (kSelBlocks + n_groups - 1) / n_groups;
  if (bpj < 1) bpj = 1;
  const int2 *pass_groups = gr;
  int pass_n = n_groups;
  bool trailing = false;   // some jobs go through all eight digits
  for (int shift = 56; shift >= 0; shift -= 8) {
    const dim3 hgrid((unsigned)bpj, (unsigned)pass_n);
    switch (shift) {
      case 56: hipLaunchKernelGGL((select_hist_kernel<MI, 56>), hgrid, dim3(256), 0, st, jb, pass_groups, N, T, shift, hist); break;
      case 48: hipLaunchKernelGGL((select_hist_kernel<MI, 48>), hgrid, dim3(256), 0, st, jb, pass_groups, N, T, shift, hist); break;
      case 40: hipLaunchKernelGGL((select_hist_kernel<MI, 40>), hgrid, dim3(256), 0, st, jb, pass_groups, N, T, shift, hist); break;
      case 32: hipLaunchKernelGGL((select_hist_kernel<MI, 32>), hgrid, dim3(256), 0, st, jb, pass_groups, N, T, shift, hist); break;
      default: hipLaunchKernelGGL((select_hist_kernel<MI, 0>), hgrid, dim3(256), 0, st, jb, pass_groups, N, T, shift, hist);
    }
    hipLaunchKernelGGL(select_pick_kernel, dim3((unsigned)n_jobs), dim3(64), 0, st, jb, shift,
                       hist, out, succ, cand_count + n_jobs, cand, cand_count);
    if (shift == kSelSmallShift) {
      hipLaunchKernelGGL(select_gather_kernel<MI>, dim3((unsigned)bpj, (unsigned)n_groups), dim3(256),
                         0, st, jb, gr, N, T, cand, cand_count, succ);
      hipLaunchKernelGGL(select_small_kernel, dim3((unsigned)n_jobs), dim3(256), 0, st, jb, cand,
                         cand_count, out, succ, cand_count + n_jobs);
      // (h_groups == nullptr - fr_select_ranks_begin: nothing is read back, the host does not wait.
      // The five remaining passes are launched whatever is left for them, over all groups: the
      // workgroups of a group without a job in them leave at once - eleven launches of a few
      // microseconds each in the usual case)
      if (h_groups == nullptr) {
        trailing = true;
        continue;
      }
      // no job left in the histogram passes (the usual case): done.  Else only the jobs with
      // too many candidates for a workgroup - heavy ties that are not ONE value - go on, and
      // the five remaining passes run over THEIR groups alone (the host reads the jobs' flags)
      unsigned int n_big = 1;
      if (hipMemcpyAsync(&n_big, cand_count + n_jobs, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess) {
        (void)hipGetLastError();
        n_big = 1;
      }
      if (n_big == 0) break;
      std::vector<SelJob> hj((size_t)n_jobs);
      if (hipMemcpy(hj.data(), jb, (size_t)n_jobs * sizeof(SelJob), hipMemcpyDeviceToHost) == hipSuccess) {
        std::vector<int32_t> active;
        for (int g = 0; g < n_groups; ++g) {
          bool left = false;
          for (int j = 0; j < h_groups[2 * g + 1]; ++j) left = left || !(hj[h_groups[2 * g] + j].pad & 4);
          if (left) {
            active.push_back(h_groups[2 * g]);
            active.push_back(h_groups[2 * g + 1]);
          }
        }
        if (active.empty()) break;
        if (hipMemcpy(gr_active, active.data(), active.size() * 4, hipMemcpyHostToDevice) == hipSuccess) {
          pass_groups = gr_active;
          pass_n = (int)active.size() / 2;
        } else {
          (void)hipGetLastError();
        }
      } else {
        (void)hipGetLastError();
      }
      trailing = true;
    }
  }
  // (jobs that went through all eight digits and need a neighbour outside what they saw: one
  // more pass over their groups; everybody else has it from the gather pass - unless a group
  // asks for more neighbours per differencing order than that pass tracks)
  if (untracked)
    hipLaunchKernelGGL(select_succ_kernel<MI>, dim3((unsigned)bpj, (unsigned)n_groups), dim3(256), 0, st,
                       jb, gr, N, T, succ);
  else if (trailing)
    hipLaunchKernelGGL(select_succ_kernel<MI>, dim3((unsigned)bpj, (unsigned)pass_n), dim3(256), 0, st,
                       jb, pass_groups, N, T, succ);
  return hipGetLastError();
}

hipError_t launch_select_ranks(void *jobs, int n_jobs, const void *groups, int n_groups,
                               const int32_t *h_groups, void *groups_scratch, int max_inc,
                               bool untracked, int64_t N, int64_t T, unsigned int *hist, double *out,
                               unsigned long long *succ, unsigned long long *cand,
                               unsigned int *cand_count, hipStream_t st) {
  if (n_jobs <= 0 || n_groups <= 0 || N * T <= 0) return hipSuccess;
  SelJob *jb = static_cast<SelJob *>(jobs);
  const int2 *gr = static_cast<const int2 *>(groups);
  int2 *ga = static_cast<int2 *>(groups_scratch);
  switch (max_inc) {
    case 0: return select_ranks_mi<0>(jb, n_jobs, gr, n_groups, h_groups, ga, untracked, N, T, hist, out, succ, cand, cand_count, st);
    case 1: return select_ranks_mi<1>(jb, n_jobs, gr, n_groups, h_groups, ga, untracked, N, T, hist, out, succ, cand, cand_count, st);
    case 2: return select_ranks_mi<2>(jb, n_jobs, gr, n_groups, h_groups, ga, untracked, N, T, hist, out, succ, cand, cand_count, st);
    default: return select_ranks_mi<kMaxInc>(jb, n_jobs, gr, n_groups, h_groups, ga, untracked, N, T, hist, out, succ, cand, cand_count, st);
  }
}

// ---------------------------------------------------------------- launchers
int walk_chunk_elems(int64_t T) { return T <= 512 ? 512 : 1024; }

#define DECL_INST(m, l) hipError_t walk_inst_m##m##_l##l(const IssArgs &, int, hipStream_t);
DECL_INST(0, 2) DECL_INST(0, 4) DECL_INST(0, 6) DECL_INST(0, 8)
DECL_INST(1, 2) DECL_INST(1, 4) DECL_INST(1, 6) DECL_INST(1, 8)
DECL_INST(2, 2) DECL_INST(2, 4) DECL_INST(2, 6) DECL_INST(2, 8)
hipError_t walk_static_launch(const IssArgs &, hipStream_t);
hipError_t walk_packed_inst_m0(const IssArgs &, int, hipStream_t);
hipError_t walk_packed_inst_m1(const IssArgs &, int, hipStream_t);

// short series: four series per workgroup, one wave each (walk_packed.h)
bool packed_supported(int64_t T, int levels, int semiring) {
  (void)semiring;  // all three semirings are instantiated
  // measured against the cooperative kernel: T = 300 138 -> 88 us, T = 384 143 -> 100 us;
  // beyond (4 pieces per wave, 8 elements per lane) it is no faster (T = 512: 73 vs 79 us)
  return (T <= 256 && levels <= 8) || (T <= 384 && levels <= 4);
}

hipError_t launch_iss_walk(IssArgs &a, int levels, hipStream_t st) {
  const int chunk = walk_chunk_elems(a.T);
  a.nchunks = (int32_t)((a.T + chunk - 1) / chunk);
  if (a.N * a.G <= 0) return hipSuccess;
  if (a.nchunks > 1 && a.carry == nullptr) return hipErrorInvalidValue;
  if (a.packed) {
    if (!packed_supported(a.T, levels, a.semiring)) return hipErrorInvalidValue;
    return a.feats ? walk_packed_inst_m1(a, levels, st) : walk_packed_inst_m0(a, levels, st);
  }
  if (a.feats) {
    if (levels <= 2) return walk_inst_m1_l2(a, chunk, st);
    if (levels <= 4) return walk_inst_m1_l4(a, chunk, st);
    if (levels <= 6) return walk_inst_m1_l6(a, chunk, st);
    return walk_inst_m1_l8(a, chunk, st);
  }
  if (a.static_prog != 0) return walk_static_launch(a, st);
  if (a.lean) {   // the fused walk's node loop with a store epilogue (walk_fused.h, MODE 2)
    if (levels <= 2) return walk_inst_m2_l2(a, chunk, st);
    if (levels <= 4) return walk_inst_m2_l4(a, chunk, st);
    if (levels <= 6) return walk_inst_m2_l6(a, chunk, st);
    return walk_inst_m2_l8(a, chunk, st);
  }
  if (levels <= 2) return walk_inst_m0_l2(a, chunk, st);
  if (levels <= 4) return walk_inst_m0_l4(a, chunk, st);
  if (levels <= 6) return walk_inst_m0_l6(a, chunk, st);
  return walk_inst_m0_l8(a, chunk, st);
}

// CosWISS: one kernel per (series, word, frequency) unit, see coswiss.h
hipError_t coswiss_inst_s1(const IssArgs &, int, hipStream_t);
hipError_t coswiss_inst_s2(const IssArgs &, int, hipStream_t);
hipError_t coswiss_inst_s3(const IssArgs &, int, hipStream_t);
hipError_t coswiss_inst_s4(const IssArgs &, int, hipStream_t);
hipError_t coswiss_inst_s5(const IssArgs &, int, hipStream_t);
hipError_t coswiss_inst_s6(const IssArgs &, int, hipStream_t);
hipError_t coswiss_inst_s7(const IssArgs &, int, hipStream_t);
hipError_t coswiss_inst_s8(const IssArgs &, int, hipStream_t);

hipError_t launch_coswiss(IssArgs &a, int exponent, hipStream_t st) {
  const int chunk = walk_chunk_elems(a.T);
  a.nchunks = (int32_t)((a.T + chunk - 1) / chunk);
  if (a.N * a.cw_W * a.cw_F <= 0) return hipSuccess;
  switch (exponent) {
    case 1: return coswiss_inst_s1(a, chunk, st);
    case 2: return coswiss_inst_s2(a, chunk, st);
    case 3: return coswiss_inst_s3(a, chunk, st);
    case 4: return coswiss_inst_s4(a, chunk, st);
    case 5: return coswiss_inst_s5(a, chunk, st);
    case 6: return coswiss_inst_s6(a, chunk, st);
    case 7: return coswiss_inst_s7(a, chunk, st);
    case 8: return coswiss_inst_s8(a, chunk, st);
    default: return hipErrorInvalidValue;
  }
}

// _ffn of fruits/iss/cos.py:93-113 for one (word, frequency): per time step
// Z = C relu(A x + b), sums over the input / hidden dimension in index order (numba's
// np.sum), Y * (Y > 0) as the ReLU.  One thread per (series, time step).
constexpr int kFfnMaxHidden = 64, kFfnMaxDims = 16;
__global__ void coswiss_ffn_kernel(const double *__restrict__ X, int64_t N, int64_t D, int64_t T,
                                   const double *__restrict__ A, const double *__restrict__ b,
                                   const double *__restrict__ Cm, int hidden,
                                   double *__restrict__ Z) {
#pragma clang fp contract(off)
  const int64_t total = N * T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / T, t = i - n * T;
    double x[kFfnMaxDims], y[kFfnMaxHidden];
    for (int d = 0; d < D; ++d) x[d] = X[(n * D + d) * T + t];
    for (int h = 0; h < hidden; ++h) {
      double acc = 0.0;
      for (int d = 0; d < D; ++d) acc = acc + A[h * D + d] * x[d];
      const double v = acc + b[h];
      y[h] = v * (v > 0.0 ? 1.0 : 0.0);
    }
    for (int d = 0; d < D; ++d) {
      double acc = 0.0;
      for (int h = 0; h < hidden; ++h) acc = acc + Cm[d * hidden + h] * y[h];
      Z[(n * D + d) * T + t] = acc;
    }
  }
}

hipError_t launch_coswiss_ffn(const double *X, int64_t N, int64_t D, int64_t T, const double *A,
                              const double *b, const double *Cm, int hidden, double *Z,
                              hipStream_t st) {
  const int64_t total = N * T;
  if (total <= 0) return hipSuccess;
  if (hidden < 1 || hidden > kFfnMaxHidden || D < 1 || D > kFfnMaxDims) return hipErrorInvalidValue;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(coswiss_ffn_kernel, dim3((unsigned)blocks), dim3(256), 0, st, X, N, D, T, A, b,
                     Cm, hidden, Z);
  return hipGetLastError();
}

// sin / cos tables of fruits/iss/cos.py:23-24; the float32 frequency is promoted to
// double before the product with T-1 (numba's typing of the reference's f4 argument)
__global__ void trig_tables_kernel(const float *__restrict__ freqs, int F, int64_t T,
                                   double *__restrict__ out) {
  const int64_t total = (int64_t)F * T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = i / T, t = i % T;
    const double ang = (3.14159265358979323846 * (double)t) / ((double)freqs[f] * (double)(T - 1));
    out[(f * 2) * T + t] = sin(ang);
    out[(f * 2 + 1) * T + t] = cos(ang);
  }
}

hipError_t launch_trig_tables(const float *freqs, int F, int64_t T, double *out, hipStream_t st) {
  const int64_t total = (int64_t)F * T;
  if (total <= 0) return hipSuccess;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(trig_tables_kernel, dim3((unsigned)blocks), dim3(256), 0, st, freqs, F, T,
                     out);
  return hipGetLastError();
}

// MPI features: mean = sum / population (0 for an empty band, increment.py:158-161).
// `pairs` (npi column, mpi column inside one iterated sum's block): NPI features whose
// band, cut and differencing order equal an MPI feature's ARE that population - the walk
// kernel skipped them (fr_pipeline_set_quantiles) and they are filled in here.
__global__ void mpi_finalize_kernel(double *__restrict__ feats, const double *__restrict__ cnt,
                                    int64_t N, int64_t stride, const int32_t *__restrict__ cols,
                                    int n_cols, const int32_t *__restrict__ pairs, int n_pairs,
                                    int per_sum, int K) {
  const int64_t per_n = (int64_t)K * (n_cols + n_pairs);
  const int64_t total = N * per_n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / per_n;
    const int64_t r = i % per_n;
    const int64_t k = r / (n_cols + n_pairs);
    const int j = (int)(r % (n_cols + n_pairs));
    if (j < n_cols) {
      const int64_t f = k * per_sum + cols[j];
      const double c = cnt[n * stride + f];
      feats[n * stride + f] = c > 0.0 ? feats[n * stride + f] / c : 0.0;
    } else {
      const int p = j - n_cols;
      feats[n * stride + k * per_sum + pairs[2 * p]] = cnt[n * stride + k * per_sum + pairs[2 * p + 1]];
    }
  }
}

hipError_t launch_mpi_finalize(double *feats, const double *cnt, int64_t N, int64_t stride,
                               const int32_t *cols, int n_cols, const int32_t *pairs, int n_pairs,
                               int per_sum, int K, hipStream_t st) {
  const int64_t total = N * (int64_t)K * (n_cols + n_pairs);
  if (total <= 0) return hipSuccess;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mpi_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, st, feats, cnt, N,
                     stride, cols, n_cols, pairs, n_pairs, per_sum, K);
  return hipGetLastError();
}

// A plan in pieces leaves its features in WALK order (plan.h, PiecedProgram): the blocks of
// `per_sum` columns of one iterated sum back into the reference's row order,
// dst[n, k * per_sum + j] = src[n, walk_of_row[k] * per_sum + j].  Writes are coalesced, reads
// gather 8 * per_sum byte runs inside the series' row (a few KB: cache hits).
__global__ __launch_bounds__(256) void gather_row_blocks_kernel(const double *__restrict__ src,
                                                                 double *__restrict__ dst, int64_t N,
                                                                 int64_t src_stride, int64_t dst_stride,
                                                                 int K, int per_sum,
                                                                 const int32_t *__restrict__ walk_of_row) {
  const int64_t F = (int64_t)K * per_sum;
  const int64_t total = N * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / F;
    const int c = (int)(i - n * F);
    const int k = c / per_sum, j = c - k * per_sum;
    dst[n * dst_stride + c] = src[n * src_stride + (int64_t)walk_of_row[k] * per_sum + j];
  }
}

hipError_t launch_gather_row_blocks(const double *src, double *dst, int64_t N, int64_t src_stride,
                                    int64_t dst_stride, int K, int per_sum, const int32_t *walk_of_row,
                                    hipStream_t st) {
  const int64_t total = N * (int64_t)K * per_sum;
  if (total <= 0) return hipSuccess;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(gather_row_blocks_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, N,
                     src_stride, dst_stride, K, per_sum, walk_of_row);
  return hipGetLastError();
}

hipError_t launch_exp_tables(const double *g, int64_t count, const float *alphas, int n_alpha,
                             double *aux, bool linear, hipStream_t st) {
  if (count <= 0 || n_alpha <= 0) return hipSuccess;
  const int bs = 256;
  hipLaunchKernelGGL(exp_tables_kernel, dim3((unsigned)((count + bs - 1) / bs)), dim3(bs), 0, st,
                     g, count, alphas, n_alpha, aux, linear ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_increments(const double *X, int64_t rows, int64_t T, int64_t shift, double *out,
                             const double *head_src, int64_t head, hipStream_t st) {
  const int64_t total = rows * T;
  if (total <= 0) return hipSuccess;
  const int bs = 256;
  int64_t blocks = (total + bs - 1) / bs;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(increments_kernel, dim3((unsigned)blocks), dim3(bs), 0, st, X, rows, T,
                     shift, out, head_src, head);
  return hipGetLastError();
}

hipError_t launch_pathlen_lookup(const double *X, int64_t N, int64_t D, int64_t T, int norm,
                                 int relative, double scale, int exact, double *out,
                                 hipStream_t st) {
  if (N <= 0 || T <= 0) return hipSuccess;
  hipLaunchKernelGGL(pathlen_lookup_kernel, dim3((unsigned)N), dim3(256), 0, st, X, D, T, norm,
                     relative, scale, exact, out);
  return hipGetLastError();
}

hipError_t launch_sieve(int kind, const double *A, int64_t N, int64_t T, int64_t a_stride, int inc,
                        const int64_t *cuts, int64_t cut_rows, int C1, const double *q, int Q1,
                        double *out, int64_t out_stride, hipStream_t st) {
  if (N <= 0) return hipSuccess;
  hipLaunchKernelGGL(sieve_kernel, dim3((unsigned)N), dim3(256), 0, st, kind, A, T, a_stride, inc,
                     cuts, cut_rows, C1, q, Q1, out, out_stride);
  return hipGetLastError();
}

hipError_t launch_pre_transform(const double *A, int64_t N, int64_t T, int64_t a_stride, int inc,
                                double *out, hipStream_t st) {
  const int64_t total = N * T;
  if (total <= 0) return hipSuccess;
  const int bs = 256;
  int64_t blocks = (total + bs - 1) / bs;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(pre_transform_kernel, dim3((unsigned)blocks), dim3(bs), 0, st, A, N, T,
                     a_stride, inc, out);
  return hipGetLastError();
}

// CosWISS term reduction (fruits/iss/cos.py:38-48): block (j, n) sums the terms of output
// row j for series n in the reference's order, `res += coeff * (tmp * sin^a * cos^b)`.
__global__ __launch_bounds__(256) void coswiss_combine_kernel(
    const double *__restrict__ A, int64_t N, int64_t T, const int32_t *__restrict__ begin,
    const double *__restrict__ coeff, const int32_t *__restrict__ desc,
    const double *__restrict__ trig, double *__restrict__ out, int64_t out_row_stride) {
#pragma clang fp contract(off)
  const int64_t j = blockIdx.x / N, n = blockIdx.x % N;
  const int i0 = begin[j], i1 = begin[j + 1];
  for (int64_t t = threadIdx.x; t < T; t += 256) {
    const double sn = trig[t], cs = trig[T + t];
    double acc = 0.0;
    for (int i = i0; i < i1; ++i) {
      double v = A[((int64_t)desc[3 * i] * N + n) * T + t];
      for (int k = desc[3 * i + 1]; k > 0; --k) v = v * sn;
      for (int k = desc[3 * i + 2]; k > 0; --k) v = v * cs;
      acc += coeff[i] * v;
    }
    out[j * out_row_stride + n * T + t] = acc;
  }
}

hipError_t launch_coswiss_combine(const double *A, int64_t N, int64_t T, int n_out,
                                  const int32_t *begin, const double *coeff, const int32_t *desc,
                                  const double *trig, double *out, int64_t out_row_stride,
                                  hipStream_t st) {
  if (N <= 0 || T <= 0 || n_out <= 0) return hipSuccess;
  hipLaunchKernelGGL(coswiss_combine_kernel, dim3((unsigned)(N * n_out)), dim3(256), 0, st, A, N,
                     T, begin, coeff, desc, trig, out, out_row_stride);
  return hipGetLastError();
}

// np.nan_to_num(features, nan=0.0) of Fruit.transform (fruits/fruit.py:172): NaN -> 0,
// +-inf -> the largest / lowest finite double
__global__ void nan_to_num_kernel(double *__restrict__ x, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double v = x[i];
    if (v != v) x[i] = 0.0;
    else if (v == __builtin_inf()) x[i] = 1.7976931348623157e308;
    else if (v == -__builtin_inf()) x[i] = -1.7976931348623157e308;
  }
}

hipError_t launch_nan_to_num(double *x, int64_t count, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  int64_t blocks = (count + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(nan_to_num_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, count);
  return hipGetLastError();
}

hipError_t launch_standardize(const double *X, int64_t rows, int64_t T, int div_std, double eps,
                              double *out, hipStream_t st) {
  if (rows <= 0 || T <= 0) return hipSuccess;
  hipLaunchKernelGGL(standardize_kernel, dim3((unsigned)rows), dim3(256), 0, st, X, T, div_std,
                     eps, out);
  return hipGetLastError();
}

}  // namespace fr

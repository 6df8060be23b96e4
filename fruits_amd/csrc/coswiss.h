// Cosine weighted ISS (fruits/iss/cos.py) for gfx950 - the "next" row behind the ISS
// hot path, built from the same pieces as the trie walk (walk.h).
//
// The reference expands every cos(g_a - g_b)^S between two consecutive letters into
// sum_m C(S,m) (sin g_a sin g_b)^(S-m) (cos g_a cos g_b)^m and then evaluates ALL
// (S+1)^(p-1) products of those terms as separate iterated sums (cos.py:11-49,
// 265-287).  The sum over the terms factorises letter by letter: with
// q_m[t] = sin^(S-m)[t] * cos^m[t],
//     A_m^(0)  = cumsum(letter_0 * q_m)
//     u^(k)    = sum_m C(S,m) * q_m * shift(A_m^(k-1))
//     A_m^(k)  = cumsum(u^(k) * letter_k * q_m)          (S+1 scans per letter)
// and the result is cumsum(u^(L-1) * letter_(L-1)) or, with total weighting,
// sum_m C(S,m) q_m A_m^(L-1).  That is (S+1)*L scans instead of up to
// (S+1)^L * L, the same value up to the rounding of a re-associated sum (the
// reference itself is compiled with fastmath=True).
//
// One 256-thread workgroup per (series n, word w, frequency f); the S+1 running
// prefixes live in registers, rows are read straight from global memory in the
// walk kernel's lane layout (they are L2 hits: all W*F units of a series run
// back to back), only wave totals cross waves (block_scan of walk.h).
#pragma once
#include "walk.h"

namespace fr {

constexpr int kCosMaxLetters = 16;  // carry slots reserved per unit (multi-chunk series)

template <class C>
__device__ __forceinline__ void load_global_row(const WalkCtx &cx, const double *gp,
                                                double (&v)[C::EP]) {
  // gp = row + t0; the lane's elements in the layout of emit_store (E consecutive
  // elements per piece, 16-byte accesses)
  constexpr int E = C::E, P = C::P;
  static_assert(E % 2 == 0, "pairs of doubles");
  const int64_t T = cx.a->T;
#pragma unroll
  for (int h = 0; h < P; ++h)
#pragma unroll
    for (int e = 0; e < E; e += 2) {
      const int idx = cx.wave * C::SPAN + h * C::PIECE + cx.lane * E + e;
      const int64_t t = cx.t0 + idx;
      if (cx.a->vec_ok) {
        vd2 q = {0.0, 0.0};
        if (cx.full_chunk || t < T) q = *reinterpret_cast<const vd2 *>(gp + idx);
        v[h * E + e] = q.x;
        v[h * E + e + 1] = q.y;
      } else {
        v[h * E + e] = t < T ? gp[idx] : 0.0;
        v[h * E + e + 1] = t + 1 < T ? gp[idx + 1] : 0.0;
      }
    }
}

// The same elements of PREPARED dimension `dp` of series n, formed from the RAW input on the way
// (fused preparation, IssArgs::prep - INC / NEW(INC) / STD exactly as walk_fused.h's staging forms
// them: x[t] - x[t - lag] with the zero padding of fruits/cache.py:8-13, then (x - mean) / (std +
// eps) with the statistics of row_stats_kernel): no prepared tensor is written or read.
template <class C>
__device__ __forceinline__ void load_prepared_row(const WalkCtx &cx, const IssArgs &a, int64_t n, int dp,
                                                  double (&v)[C::EP]) {
  constexpr int E = C::E, P = C::P;
  const int raw = as_const(a.prep)[4 * dp], lag = as_const(a.prep)[4 * dp + 1];
  const bool standardise = as_const(a.prep)[4 * dp + 2] != 0;
  const double *gp = a.X + (n * a.D + raw) * a.T;
  double mean = 0.0, den = 1.0;
  if (standardise) {
    mean = as_const(a.stats)[(n * a.n_prep + dp) * 2];
    den = as_const(a.stats)[(n * a.n_prep + dp) * 2 + 1];
  }
#pragma unroll
  for (int h = 0; h < P; ++h)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int64_t t = cx.t0 + cx.wave * C::SPAN + h * C::PIECE + cx.lane * E + e;
      double x = 0.0;
      if (t < a.T) {
        x = gp[t];
        if (lag > 0) x = t >= lag ? x - gp[t - lag] : 0.0;
        if (standardise) x = (x - mean) / den;
      }
      v[h * E + e] = x;
    }
}

// v *= sin^(S-M) * cos^M by repeated multiplication, sines first (cos.py:37-40)
template <int S, int M, int EP>
__device__ __forceinline__ void mul_trig(double (&v)[EP], const double (&sn)[EP],
                                         const double (&cs)[EP]) {
#pragma unroll
  for (int k = 0; k < S - M; ++k)
#pragma unroll
    for (int i = 0; i < EP; ++i) v[i] = v[i] * sn[i];
#pragma unroll
  for (int k = 0; k < M; ++k)
#pragma unroll
    for (int i = 0; i < EP; ++i) v[i] = v[i] * cs[i];
}

template <int S, int M, int EP>
__device__ __forceinline__ void build_q(double (&q)[S + 1][EP], const double (&sn)[EP],
                                        const double (&cs)[EP]) {
#pragma unroll
  for (int i = 0; i < EP; ++i) q[M][i] = 1.0;
  mul_trig<S, M, EP>(q[M], sn, cs);
  if constexpr (M < S) build_q<S, M + 1, EP>(q, sn, cs);
}

constexpr double cos_binom(int s, int m) {
  double c = 1.0;
  for (int k = 0; k < m; ++k) c = c * (s - k) / (k + 1);
  return c;
}

template <class C, int S>
struct CosState {
  double xa[S + 1][C::EP];  // exclusive prefixes of the previous letter's S+1 scans
  double q[S + 1][C::EP];   // q_m = sin^(S-m) cos^m, formed once per unit
  bool last;
};

// local (per lane, per piece) inclusive sums of s * q_M for M, M+1, ... S
template <class C, int S, int M>
__device__ __forceinline__ void cos_local(const CosState<C, S> &st, const double (&s)[C::EP],
                                          double (&l)[S + 1][C::EP]) {
  constexpr int E = C::E, P = C::P, EP = C::EP;
  double v[EP];
#pragma unroll
  for (int i = 0; i < EP; ++i) v[i] = s[i] * st.q[M][i];
#pragma unroll
  for (int h = 0; h < P; ++h) {
    l[M][h * E] = v[h * E];
#pragma unroll
    for (int e = 1; e < E; ++e) l[M][h * E + e] = l[M][h * E + e - 1] + v[h * E + e];
  }
  if constexpr (M < S) cos_local<C, S, M + 1>(st, s, l);
}

// total weighting: result += C(S,m) * (A_m * q_m), cos.py:42-48 (m is an unrolled index)
template <class C, int S, int M = 0>
__device__ __forceinline__ void cos_add_result(const CosState<C, S> &st, int m,
                                               double (&c)[C::EP], double (&res)[C::EP]) {
  if (m == M) {
#pragma unroll
    for (int i = 0; i < C::EP; ++i) res[i] += cos_binom(S, M) * (c[i] * st.q[M][i]);
  } else if constexpr (M < S) {
    cos_add_result<C, S, M + 1>(st, m, c, res);
  }
}

// The S+1 scans of one letter, A_m = cumsum(s * q_m), behind ONE workgroup barrier: local
// sums of all of them, their wave scans interleaved (DPP latencies overlap), one LDS
// exchange of S+1 totals per wave.  Values are formed exactly as in block_scan (walk.h).
template <class C, int S>
__device__ __forceinline__ void cos_scan_all(WalkCtx &cx, CosState<C, S> &st,
                                             const double (&s)[C::EP], int slot,
                                             double (&res)[C::EP], double *tot_all) {
  constexpr int E = C::E, P = C::P, EP = C::EP, NW = C::NW, M = S + 1;
  double l[M][EP];
  double incl[M * P], excl[M * P], ptot[M * P];
  cos_local<C, S, 0>(st, s, l);
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int h = 0; h < P; ++h) incl[m * P + h] = l[m][h * E + E - 1];
  wave_inclusive_scan_multi<M * P, 0>(incl);
#pragma unroll
  for (int i = 0; i < M * P; ++i) {
    excl[i] = wave_shift_right1<0>(incl[i]);
    ptot[i] = wave_last_lane(incl[i]);
  }
  double carry_in[M], base[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    carry_in[m] = 0.0;
    if constexpr (C::MULTI != 0) {
      if (!cx.first_chunk) carry_in[m] = cx.carry[slot + m];
    }
  }
  if constexpr (NW == 1) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      base[m] = carry_in[m];
      if constexpr (C::MULTI != 0) {
        double total = ptot[m * P];
#pragma unroll
        for (int h = 1; h < P; ++h) total += ptot[m * P + h];
        if (cx.lane == 0) cx.carry[slot + m] = carry_in[m] + total;
      }
    }
  } else {
    static_assert(NW == 1 || NW == 4, "cross-wave prefix is written for 4 waves");
    double *tot = tot_all + cx.buf * (NW * M);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      double wave_total = ptot[m * P];
#pragma unroll
      for (int h = 1; h < P; ++h) wave_total += ptot[m * P + h];
      if (cx.lane == 0) tot[cx.wave * M + m] = wave_total;
    }
    lds_barrier();
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const double t0 = tot[m], t1 = tot[M + m], t2 = tot[2 * M + m], t3 = tot[3 * M + m];
      const double p2 = t0 + t1, p3 = p2 + t2;
      base[m] = cx.wave == 0 ? 0.0 : (cx.wave == 1 ? t0 : (cx.wave == 2 ? p2 : p3));
      if constexpr (C::MULTI != 0) {
        base[m] += carry_in[m];
        if (cx.wave == 0 && cx.lane == 0) cx.carry[slot + m] = carry_in[m] + (p3 + t3);
      }
    }
    cx.buf ^= 1;
  }
#pragma unroll
  for (int m = 0; m < M; ++m) {
    double c[EP];
    double b = base[m];
#pragma unroll
    for (int h = 0; h < P; ++h) {
      const double off = b + excl[m * P + h];
      st.xa[m][h * E] = off;
#pragma unroll
      for (int e = 0; e + 1 < E; ++e) {
        c[h * E + e] = off + l[m][h * E + e];
        st.xa[m][h * E + e + 1] = c[h * E + e];
      }
      c[h * E + E - 1] = b + incl[m * P + h];
      b += ptot[m * P + h];
    }
    if (st.last) cos_add_result<C, S>(st, m, c, res);
  }
}

template <class C, int S, int M>
__device__ __forceinline__ void cos_combine(const CosState<C, S> &st, double (&u)[C::EP]) {
  constexpr int EP = C::EP;
  double v[EP];
#pragma unroll
  for (int i = 0; i < EP; ++i) v[i] = st.xa[M][i] * st.q[M][i];
#pragma unroll
  for (int i = 0; i < EP; ++i) u[i] += cos_binom(S, M) * v[i];
  if constexpr (M < S) cos_combine<C, S, M + 1>(st, u);
}

// one (series, word, frequency) unit for one time chunk
template <class C, int S>
__device__ __forceinline__ void coswiss_unit(WalkCtx &cx, const double *xrow, const double *trig,
                                             int lb, int le, bool total, int k_out,
                                             double *tot_all, const double *mask = nullptr) {
  constexpr int EP = C::EP;
  const IssArgs &a = *cx.a;
  CosState<C, S> st;
  double res[EP], resx[EP];
  {
    double sn[EP], cs[EP];
    load_global_row<C>(cx, trig, sn);
    load_global_row<C>(cx, trig + a.T, cs);
    build_q<S, 0, EP>(st.q, sn, cs);
  }
  Ops2 pre;
  if constexpr (C::MODE == 1) pre = load_ops2(a, k_out, 0);
#pragma unroll
  for (int i = 0; i < EP; ++i) res[i] = resx[i] = 0.0;
  const int L = le - lb;
  double s[EP];
  for (int k = 0; k < L; ++k) {
    if (k == 0) {
#pragma unroll
      for (int i = 0; i < EP; ++i) s[i] = 1.0;
    } else {
#pragma unroll
      for (int i = 0; i < EP; ++i) s[i] = 0.0;
      cos_combine<C, S, 0>(st, s);
    }
    // the letters, one factor per occurrence in ascending dimension (cos.py:30-36)
    const int fb = as_const(a.cw_fac_begin)[lb + k], fe = as_const(a.cw_fac_begin)[lb + k + 1];
    for (int f = fb; f < fe; ++f) {
      const int code = as_const(a.factors)[f];
      double v[EP];
      if (C::MODE == 1 && a.prep != nullptr)
        load_prepared_row<C>(cx, a, cx.series, code & FAC_ROW_MASK, v);
      else
        load_global_row<C>(cx, xrow + (int64_t)(code & FAC_ROW_MASK) * a.T, v);
      if (code & FAC_DIV) {
#pragma unroll
        for (int i = 0; i < EP; ++i) s[i] = s[i] / v[i];
      } else {
#pragma unroll
        for (int i = 0; i < EP; ++i) s[i] = s[i] * v[i];
      }
    }
    if (mask != nullptr) {
      // dropout (cos.py:80): the summand of letter k is zeroed at the drawn indices before
      // its cumsum - a select, so that a dropped inf / nan is an exact 0 like in the reference
      double keep[EP];
      load_global_row<C>(cx, mask + (int64_t)k * a.T, keep);
#pragma unroll
      for (int i = 0; i < EP; ++i) s[i] = keep[i] != 0.0 ? s[i] : 0.0;
    }
    const int slot = k * (S + 1);
    st.last = k == L - 1;
    if (st.last && !total) {
      block_scan<C>(cx, s, res, resx, slot);
      break;
    }
    cos_scan_all<C, S>(cx, st, s, slot, res, tot_all);
  }
  if constexpr (C::MODE == 1) {
    Rec nd;
#pragma unroll
    for (int i = 0; i < 16; ++i) nd.w[i] = 0;
    nd.w[6] = 1;
    nd.w[7] = k_out;
    cx.slot = kCosMaxLetters * (S + 1);
    if (total) prev_first_differences<C>(cx, res, resx);  // resx[t] = res[t-1]
    cx.slot += 3;
    fused_all<C>(cx, nd, pre, res, resx, s, !total);
  } else {
    emit_store<C>(cx, res, cx.out_base + (int64_t)k_out * a.out_k_stride);
  }
}

template <class C, int S>
__global__ __launch_bounds__(kWalkThreads) void coswiss_kernel(const IssArgs a) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  WalkCtx cx;
  cx.a = &a;
  cx.rows = nullptr;
  double *tot_all = lds;  // [2][NW][S+1] wave totals of cos_scan_all
  cx.tot = lds + 2 * C::NW * (S + 1);
  cx.tail = cx.tot + 2 * C::NW;
  cx.carry = cx.tail + 2 * C::NW;
  cx.tid = tid;
  cx.lane = tid & 63;
  cx.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  cx.team = 0;
  cx.buf = 0;
  cx.tail_buf = 0;
  cx.pc_begin = 0;
  if constexpr (C::MODE == 1) {
    // feature window (walk_device.h, feat_flush): the features of the unit's ONE output row,
    // kept over its time chunks and flushed behind the last one
    double *fw = cx.carry + (kCosMaxLetters * (S + 1) + 8);
    cx.fl_val = (lds_f64 *)fw;
    cx.fl_cnt = (lds_f64 *)(fw + a.feat_window);
    cx.fl_col = (lds_i32 *)(fw + (a.has_mpi ? 2 : 1) * a.feat_window);
    for (int sl = tid; sl < a.feat_window; sl += kWalkThreads) {
      cx.fl_val[sl] = 0.0;
      if (a.has_mpi) cx.fl_cnt[sl] = 0.0;
    }
    cx.fslot = 0;
    cx.fused_used = 0;
  }
  const int64_t per_series = (int64_t)a.cw_W * a.cw_F;
  const int64_t units = a.N * per_series;
  for (int64_t u = blockIdx.x; u < units; u += gridDim.x) {
    const int64_t n = u / per_series;
    const int j = (int)(u % per_series);
    const int w = j / a.cw_F, f = j % a.cw_F;
    const int lb = as_const(a.cw_letter_begin)[w], le = as_const(a.cw_letter_begin)[w + 1];
    for (int64_t chunk = 0; chunk < a.nchunks; ++chunk) {
      const int64_t t0 = chunk * C::CHUNK;
      cx.t0 = t0;
      cx.first_chunk = chunk == 0;
      cx.full_chunk = t0 + C::CHUNK <= a.T;
      cx.out_base = a.out + n * a.out_n_stride + t0;
      if constexpr (C::MODE == 1) {
        cx.feat_row = a.feats + n * a.feat_stride;
        cx.cnt_row = a.cnt + n * a.feat_stride;
        cx.cut_row = a.series_cuts ? a.series_cuts + n * a.cut_slots : nullptr;
      cx.series = n;
      }
      coswiss_unit<C, S>(cx, a.X + (int64_t)j * a.cw_x_unit_stride + n * a.D * a.T + t0,
                         a.aux + (int64_t)f * 2 * a.T + t0, lb, le, a.cw_total != 0, j, tot_all,
                         a.cw_mask ? a.cw_mask + (int64_t)j * a.cw_Lmax * a.T + t0 : nullptr);
    }
    if constexpr (C::MODE == 1) {
      cx.fused_used = a.n_ops;
      feat_flush<C>(cx, false);
    }
  }
}

// Short series (T <= 384): four units per workgroup, one wave each - the scans are
// wave-local (no barrier, no LDS), like iss_walk_packed_kernel (walk_packed.h).
template <class C, int S>
__global__ __launch_bounds__(kWalkThreads) void coswiss_packed_kernel(const IssArgs a) {
  static_assert(C::TEAM == 1 && C::MULTI == 0, "wave-per-unit configuration");
  const int tid = threadIdx.x;
  WalkCtx cx;
  cx.a = &a;
  cx.rows = nullptr;
  cx.tot = nullptr;
  cx.tail = nullptr;
  cx.carry = nullptr;
  cx.tid = tid;
  cx.lane = tid & 63;
  cx.wave = 0;
  cx.team = __builtin_amdgcn_readfirstlane(tid >> 6);
  cx.buf = 0;
  cx.tail_buf = 0;
  cx.pc_begin = 0;
  cx.t0 = 0;
  cx.first_chunk = true;
  cx.full_chunk = C::CHUNK <= a.T;
  const int64_t per_series = (int64_t)a.cw_W * a.cw_F;
  const int64_t units = a.N * per_series;
  for (int64_t u = (int64_t)blockIdx.x * C::TEAMS + cx.team; u < units;
       u += (int64_t)gridDim.x * C::TEAMS) {
    const int64_t n = u / per_series;
    const int j = (int)(u % per_series);
    const int w = j / a.cw_F, f = j % a.cw_F;
    const int lb = as_const(a.cw_letter_begin)[w], le = as_const(a.cw_letter_begin)[w + 1];
    cx.out_base = a.out + n * a.out_n_stride;
    if constexpr (C::MODE == 1) {
      cx.feat_row = a.feats + n * a.feat_stride;
      cx.cnt_row = a.cnt + n * a.feat_stride;
      cx.cut_row = a.series_cuts ? a.series_cuts + n * a.cut_slots : nullptr;
      cx.series = n;
    }
    coswiss_unit<C, S>(cx, a.X + (int64_t)j * a.cw_x_unit_stride + n * a.D * a.T,
                       a.aux + (int64_t)f * 2 * a.T, lb, le, a.cw_total != 0, j, nullptr,
                       a.cw_mask ? a.cw_mask + (int64_t)j * a.cw_Lmax * a.T : nullptr);
  }
}

template <int P, bool VEC, int MODE, int S>
static hipError_t launch_coswiss_packed_cfg(const IssArgs &a, hipStream_t st) {
  using C = WalkCfg<(P == 2 ? 4 : 2), (P == 2 ? 1 : P), 1, 0, VEC, false, 1, MODE, 0>;
  const int64_t units = a.N * a.cw_W * a.cw_F;
  static LaunchCache cache;  // per instantiation; per-device entries, thread-safe
  int per_cu = 1;
  hipError_t e = cache.facts(coswiss_packed_kernel<C, S>, kWalkThreads, 0, &per_cu);
  if (e != hipSuccess) return e;
  int64_t blocks = (units + C::TEAMS - 1) / C::TEAMS;
  const int64_t resident = (int64_t)per_cu * device_cu_count();
  if (blocks > resident) blocks = resident;
  if (blocks < 1) return hipSuccess;
  hipLaunchKernelGGL((coswiss_packed_kernel<C, S>), dim3((unsigned)blocks), dim3(kWalkThreads), 0,
                     st, a);
  return hipGetLastError();
}

template <int P, int MULTI, bool VEC, int MODE, int S>
static hipError_t launch_coswiss_cfg(const IssArgs &a, hipStream_t st) {
  // 1024-element chunks: 4 consecutive elements per lane in ONE piece - half the wave-scan
  // chains of the walk kernel's 2 x 2 layout (this kernel is bound by vector issue, and
  // the S+1 scans of a letter already overlap their DPP latencies)
  using C = WalkCfg<(P == 2 ? 4 : 2), (P == 2 ? 1 : P), 1, MULTI, VEC, false, 4, MODE, 0>;
  const size_t lds = (2 * C::NW * (S + 1) + 4 * C::NW + (kCosMaxLetters * (S + 1) + 8)) *
                         sizeof(double) +
                     (MODE == 1 ? feat_window_bytes(a.feat_window, a.has_mpi != 0) : 0);
  const int64_t units = a.N * a.cw_W * a.cw_F;
  static LaunchCache cache;  // per instantiation; per-device entries, thread-safe
  int per_cu = 1;
  hipError_t e = cache.facts(coswiss_kernel<C, S>, kWalkThreads, lds, &per_cu);
  if (e != hipSuccess) return e;
  int64_t blocks = (int64_t)per_cu * device_cu_count();
  if (blocks > units || !a.persistent) blocks = units;   // persistent grid or one workgroup per unit
  if (blocks < 1) return hipSuccess;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL((coswiss_kernel<C, S>), dim3((unsigned)blocks), dim3(kWalkThreads), lds, st,
                     a);
  return hipGetLastError();
}

}  // namespace fr

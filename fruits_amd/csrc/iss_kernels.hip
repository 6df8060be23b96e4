// HIP kernels of the ISS hot path for gfx950 (CDNA4, wave64).
//
// iss_walk_kernel: one 256-thread workgroup per (series n, group of sub-tries).
//   * the X rows (and exp tables) the words reference are staged ONCE per time
//     chunk into LDS with coalesced 16-byte loads;
//   * the workgroup walks the prefix trie in DFS order; the running exclusive
//     prefix of every open ancestor lives in registers (one "frame" per level);
//   * per node: multiply / divide the letters into the parent's prefix in the
//     reference's order (fruits/iss/semiring.py:143-149), then an inclusive
//     scan along time = thread-local sums -> wave64 DPP scan -> LDS cross-wave
//     offsets (+ the carry of earlier chunks), emit with 16-byte coalesced
//     stores into the reference's (K,N,T) layout (fruits/iss/iss.py:46).
// The kernel is HBM-bound (one write per output element, X read once per
// group); there is no contraction anywhere, so no MFMA.
#include <hip/hip_runtime.h>

#include "kernels.h"

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

// ---------------------------------------------------------------- wave scan
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch(double v) {
  // value of the DPP source lane, 0.0 where there is no source / row disabled
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_inclusive_scan(double v) {
  v += dpp_fetch<0x111, 0xf>(v);  // row_shr:1
  v += dpp_fetch<0x112, 0xf>(v);  // row_shr:2
  v += dpp_fetch<0x114, 0xf>(v);  // row_shr:4
  v += dpp_fetch<0x118, 0xf>(v);  // row_shr:8
  v += dpp_fetch<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
  v += dpp_fetch<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
  return v;
}

__device__ __forceinline__ double wave_shift_right1(double v) {
  return dpp_fetch<0x138, 0xf>(v);  // wave_shr:1, lane 0 gets 0.0
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for
// vmcnt(0), i.e. for the acknowledgement of every global store the wave has in
// flight - that would serialise each node's output stores with the next node's
// scan.  Waiting for lgkmcnt(0) alone keeps the stores streaming.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---------------------------------------------------------------- walk kernel
template <int E_, int P_, int MAXLV_, bool MULTI_>
struct WalkCfg {
  static constexpr bool MULTI = MULTI_;  // more than one time chunk (carries in memory)
  static constexpr int E = E_;          // contiguous elements per thread per piece
  static constexpr int P = P_;          // pieces per thread
  static constexpr int EP = E_ * P_;
  static constexpr int MAXLV = MAXLV_;
  static constexpr int PIECE = kWalkThreads * E_;  // elements per piece
  static constexpr int CHUNK = PIECE * P_;         // elements per time chunk
};

struct WalkCtx {
  const IssArgs *a;
  const double *rows;   // LDS: staged rows [R][CHUNK]
  double *tot;          // LDS: wave totals [2][P*NW]
  double *out_base;     // out + n*out_n_stride + t0
  double *carry;        // carry slots of this series (multi-chunk) or nullptr
  int64_t t0;           // first time index of the chunk
  int node_end;
  int tid, lane, wave;
  int buf;
  bool first_chunk;
};

template <class C>
__device__ __forceinline__ void block_scan(WalkCtx &cx, const double (&s)[C::EP],
                                           double (&c)[C::EP], double (&x)[C::EP],
                                           int carry_slot) {
  constexpr int E = C::E, P = C::P, NW = kWalkThreads / 64;
  double l[C::EP];
  double incl[P], excl[P];
#pragma unroll
  for (int h = 0; h < P; ++h) {
    l[h * E] = s[h * E];
#pragma unroll
    for (int e = 1; e < E; ++e) l[h * E + e] = l[h * E + e - 1] + s[h * E + e];
  }
#pragma unroll
  for (int h = 0; h < P; ++h) incl[h] = wave_inclusive_scan(l[h * E + E - 1]);
#pragma unroll
  for (int h = 0; h < P; ++h) excl[h] = wave_shift_right1(incl[h]);
  double *tot = cx.tot + cx.buf * (P * NW);
  if (cx.lane == 63) {
#pragma unroll
    for (int h = 0; h < P; ++h) tot[h * NW + cx.wave] = incl[h];
  }
  double run = 0.0;
  if constexpr (C::MULTI) {
    if (!cx.first_chunk) run = cx.carry[carry_slot];
  }
  lds_barrier();
  double base[P] = {};
#pragma unroll
  for (int h = 0; h < P; ++h) {
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      if (w == cx.wave) base[h] = run;
      run += tot[h * NW + w];
    }
  }
  cx.buf ^= 1;
  // every wave stores the same value; a wave only ever re-reads its own store
  if constexpr (C::MULTI) {
    if (cx.lane == 0) cx.carry[carry_slot] = run;
  }
#pragma unroll
  for (int h = 0; h < P; ++h) {
    const double off = base[h] + excl[h];
    x[h * E] = off;
    c[h * E] = off + l[h * E];
#pragma unroll
    for (int e = 1; e < E; ++e) {
      x[h * E + e] = c[h * E + e - 1];
      c[h * E + e] = off + l[h * E + e];
    }
  }
}

template <class C>
__device__ __forceinline__ void emit_store(const WalkCtx &cx, const double (&v)[C::EP],
                                           double *dst) {
  constexpr int E = C::E, P = C::P;
  const int64_t T = cx.a->T;
#pragma unroll
  for (int h = 0; h < P; ++h) {
    const int idx = h * C::PIECE + cx.tid * E;
    const int64_t t = cx.t0 + idx;
    if (cx.a->vec_ok) {
      static_assert(E % 2 == 0, "pieces are stored as double2");
#pragma unroll
      for (int e = 0; e < E; e += 2)
        if (t + e < T) {
          vd2 val = {v[h * E + e], v[h * E + e + 1]};
          if (cx.a->nt_store)
            __builtin_nontemporal_store(val, reinterpret_cast<vd2 *>(dst + idx + e));
          else
            *reinterpret_cast<vd2 *>(dst + idx + e) = val;
        }
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (t + e < T) dst[idx + e] = v[h * E + e];
    }
  }
}

template <class C>
__device__ __forceinline__ void process_node(WalkCtx &cx, const NodeDesc &nd, int node_slot,
                                             const double (&pin)[C::EP],
                                             double (&pout)[C::EP]) {
  constexpr int E = C::E, P = C::P, EP = C::EP;
  const IssArgs &a = *cx.a;
  double s[EP];
#pragma unroll
  for (int i = 0; i < EP; ++i) s[i] = pin[i];
  for (int f = 0; f < nd.fac_count; ++f) {
    const int fe = as_const(a.factors)[nd.fac_begin + f];
    const double *row = cx.rows + (fe & FAC_ROW_MASK) * C::CHUNK + cx.tid * E;
    if (fe & FAC_DIV) {
#pragma unroll
      for (int h = 0; h < P; ++h)
#pragma unroll
        for (int e = 0; e < E; ++e) s[h * E + e] = s[h * E + e] / row[h * C::PIECE + e];
    } else {
#pragma unroll
      for (int h = 0; h < P; ++h)
#pragma unroll
        for (int e = 0; e < E; ++e) s[h * E + e] = s[h * E + e] * row[h * C::PIECE + e];
    }
  }
  const bool has_children = (nd.flags & F_CHILDREN) != 0;
  const bool need2 = has_children && nd.z_mul >= 0;
  const bool need1 = nd.emit_count > 0 || (has_children && !need2);
  if (need1) {
    double c[EP], x[EP];
    block_scan<C>(cx, s, c, x, 2 * node_slot);
    if (nd.emit_count > 0) {
      if (nd.emit_mul >= 0) {
        const double *row = cx.rows + nd.emit_mul * C::CHUNK + cx.tid * E;
#pragma unroll
        for (int h = 0; h < P; ++h)
#pragma unroll
          for (int e = 0; e < E; ++e) c[h * E + e] = c[h * E + e] * row[h * C::PIECE + e];
      }
      for (int j = 0; j < nd.emit_count; ++j) {
        const int64_t k = as_const(a.emit_rows)[nd.emit_begin + j];
        emit_store<C>(cx, c, cx.out_base + k * a.out_k_stride);
      }
    }
    if (has_children && !need2) {
#pragma unroll
      for (int i = 0; i < EP; ++i) pout[i] = x[i];
    }
  }
  if (need2) {
    const double *row = cx.rows + nd.z_mul * C::CHUNK + cx.tid * E;
    double s2[EP], c[EP], x[EP];
#pragma unroll
    for (int h = 0; h < P; ++h)
#pragma unroll
      for (int e = 0; e < E; ++e) s2[h * E + e] = s[h * E + e] * row[h * C::PIECE + e];
    block_scan<C>(cx, s2, c, x, 2 * node_slot + 1);
#pragma unroll
    for (int i = 0; i < EP; ++i) pout[i] = x[i];
  }
}

__device__ __forceinline__ NodeDesc load_node(const NodeDesc *nodes, int pc) {
  // uniform address -> scalar loads
  cptr<int32_t> q = as_const(reinterpret_cast<const int32_t *>(nodes + pc));
  NodeDesc nd;
  nd.level = q[0]; nd.flags = q[1]; nd.fac_begin = q[2]; nd.fac_count = q[3];
  nd.emit_begin = q[4]; nd.emit_count = q[5]; nd.emit_mul = q[6]; nd.z_mul = q[7];
  return nd;
}

template <class C, int LV>
__device__ __forceinline__ void walk(WalkCtx &cx, const double (&pin)[C::EP], int &pc) {
  const IssArgs &a = *cx.a;
  while (pc < cx.node_end) {
    NodeDesc nd = load_node(a.nodes, pc);
    if (nd.level != LV) break;
    double pout[C::EP];
    process_node<C>(cx, nd, as_const(a.node_ids)[pc], pin, pout);
    ++pc;
    // only children continue in place in this frame
    while (pc < cx.node_end) {
      NodeDesc nc = load_node(a.nodes, pc);
      if (!(nc.flags & F_CHAIN) || nc.level != LV) break;
      process_node<C>(cx, nc, as_const(a.node_ids)[pc], pout, pout);
      ++pc;
    }
    if constexpr (LV + 1 < C::MAXLV) {
      if (pc < cx.node_end && load_node(a.nodes, pc).level == LV + 1) walk<C, LV + 1>(cx, pout, pc);
    }
  }
}

template <class C>
__global__ __launch_bounds__(kWalkThreads) void iss_walk_kernel(const IssArgs a) {
  extern __shared__ double lds[];
  constexpr int E = C::E, P = C::P, NW = kWalkThreads / 64;
  const int tid = threadIdx.x;
  // (series, group) of this workgroup.  Workgroups are dealt round-robin over
  // the 8 XCDs, so b and b+8 share an L2: keep the G groups of one series on
  // one XCD when N is a multiple of 8 (speed only).
  int64_t n;
  int g;
  {
    const int64_t b = blockIdx.x;
    if (a.xcd_map) {
      const int64_t q = b >> 3, r = b & 7;
      n = (q / a.G) * 8 + r;
      g = (int)(q % a.G);
    } else {
      n = b / a.G;
      g = (int)(b % a.G);
    }
  }
  WalkCtx cx;
  cx.a = &a;
  cx.rows = lds;
  cx.tot = lds + (int64_t)a.R * C::CHUNK;
  cx.tid = tid;
  cx.lane = tid & 63;
  cx.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  cx.buf = 0;
  cx.carry = a.carry ? a.carry + n * (2 * (int64_t)a.total_nodes) : nullptr;
  const int node_begin = as_const(a.group_begin)[g];
  cx.node_end = as_const(a.group_begin)[g + 1];
  double *rows_w = lds;

  for (int64_t chunk = 0; chunk < a.nchunks; ++chunk) {
    const int64_t t0 = chunk * C::CHUNK;
    cx.t0 = t0;
    cx.first_chunk = chunk == 0;
    cx.out_base = a.out + n * a.out_n_stride + t0;
    if (chunk > 0) lds_barrier();
    // stage the referenced rows of this chunk
    for (int r = 0; r < a.R; ++r) {
      const int src = as_const(a.row_src)[r];
      const double *gp = src >= 0
                             ? a.X + (n * a.D + src) * a.T
                             : a.aux + (int64_t)(-src - 1) * a.aux_tab_stride + n * a.aux_n_stride;
#pragma unroll
      for (int h = 0; h < P; ++h) {
        const int idx = h * C::PIECE + tid * E;
        const int64_t t = t0 + idx;
        if (a.vec_ok) {
#pragma unroll
          for (int e = 0; e < E; e += 2) {
            vd2 v = {0.0, 0.0};
            if (t + e < a.T) v = *reinterpret_cast<const vd2 *>(gp + t + e);
            *reinterpret_cast<vd2 *>(rows_w + r * C::CHUNK + idx + e) = v;
          }
        } else {
#pragma unroll
          for (int e = 0; e < E; ++e)
            rows_w[r * C::CHUNK + idx + e] = (t + e < a.T) ? gp[t + e] : 0.0;
        }
      }
    }
    __syncthreads();
    double ones[C::EP];
#pragma unroll
    for (int i = 0; i < C::EP; ++i) ones[i] = 1.0;
    int pc = node_begin;
    walk<C, 0>(cx, ones, pc);
  }
  (void)NW;
}

// ---------------------------------------------------------------- exp tables
// aux[2a]   = exp( g * alpha_a)   (np.exp(weights * alpha[k]),  semiring.py:123,150)
// aux[2a+1] = exp(-g * alpha_a)   (np.exp(-weights * alpha[k]), semiring.py:119,153,157)
__global__ void exp_tables_kernel(const double *__restrict__ g, int64_t count,
                                  const float *__restrict__ alphas, int n_alpha,
                                  double *__restrict__ aux) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const double w = g[i];
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
                                                              double *__restrict__ out) {
  __shared__ double sm_tot[2][4];
  __shared__ double sm_red[4];
  const int64_t n = blockIdx.x;
  const double *x = X + n * D * T;  // dimension 0 only
  double *o = out + n * T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double carry = 0.0;
  int buf = 0;
  for (int64_t t0 = 0; t0 < T; t0 += 512) {
    double s[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int64_t t = t0 + tid * 2 + e;
      double d = 0.0;
      if (t < T && t >= 1) d = x[t] - x[t - 1];
      s[e] = (norm == 1) ? fabs(d) : d * d;
      if (t >= T) s[e] = 0.0;
    }
    const double l1 = s[0] + s[1];
    const double incl = wave_inclusive_scan(l1);
    const double excl = wave_shift_right1(incl);
    if (lane == 63) sm_tot[buf][wave] = incl;
    __syncthreads();
    double run = carry, base = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w == wave) base = run;
      run += sm_tot[buf][w];
    }
    carry = run;
    buf ^= 1;
    const double off = base + excl;
    const int64_t t = t0 + tid * 2;
    if (t < T) o[t] = off + s[0];
    if (t + 1 < T) o[t + 1] = off + l1;
  }
  if (relative == 2) return;  // raw cumulative path length (SharedSeedCache entry)
  __syncthreads();
  // every thread re-reads only elements it wrote itself (same t -> same thread)
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
      out[n * out_stride + j] = row[idx];
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
// (np.std), or 1 when var=False.
__device__ __forceinline__ double block_reduce_sum(double v, double *sm) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sm[w];
  return r;
}

__global__ __launch_bounds__(256) void standardize_kernel(const double *__restrict__ X, int64_t T,
                                                           int div_std, double eps,
                                                           double *__restrict__ out) {
  __shared__ double sm[4];
  const double *x = X + (int64_t)blockIdx.x * T;
  double *o = out + (int64_t)blockIdx.x * T;
  double acc = 0.0;
  for (int64_t t = threadIdx.x; t < T; t += blockDim.x) acc += x[t];
  const double mean = block_reduce_sum(acc, sm) / (double)T;
  double sd = 1.0;
  if (div_std) {
    double v = 0.0;
    for (int64_t t = threadIdx.x; t < T; t += blockDim.x) {
      const double d = x[t] - mean;
      v += d * d;
    }
    sd = sqrt(block_reduce_sum(v, sm) / (double)T);
  }
  const double den = sd + eps;
  for (int64_t t = threadIdx.x; t < T; t += blockDim.x) o[t] = (x[t] - mean) / den;
}

// ---------------------------------------------------------------- launchers
template <int E, int P, int LV, bool MULTI>
static hipError_t launch_walk_cfg(const IssArgs &a, int64_t blocks, hipStream_t st) {
  using C = WalkCfg<E, P, LV, MULTI>;
  const size_t lds = ((size_t)a.R * C::CHUNK + 2 * P * (kWalkThreads / 64)) * sizeof(double);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)iss_walk_kernel<C>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(iss_walk_kernel<C>, dim3((unsigned)blocks), dim3(kWalkThreads), lds, st, a);
  return hipGetLastError();
}

template <int E, int P, bool MULTI>
static hipError_t launch_walk_lv(const IssArgs &a, int levels, int64_t blocks, hipStream_t st) {
  if (levels <= 2) return launch_walk_cfg<E, P, 2, MULTI>(a, blocks, st);
  if (levels <= 4) return launch_walk_cfg<E, P, 4, MULTI>(a, blocks, st);
  if (levels <= 8) return launch_walk_cfg<E, P, 8, MULTI>(a, blocks, st);
  return launch_walk_cfg<E, P, kMaxLevels, MULTI>(a, blocks, st);
}

int walk_chunk_elems(int64_t T) { return T <= 512 ? 512 : 1024; }

hipError_t launch_iss_walk(IssArgs &a, int levels, hipStream_t st) {
  const int chunk = walk_chunk_elems(a.T);
  a.nchunks = (int32_t)((a.T + chunk - 1) / chunk);
  const int64_t blocks = a.N * a.G;
  if (blocks <= 0) return hipSuccess;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  if (chunk == 512) return launch_walk_lv<2, 1, false>(a, levels, blocks, st);
  if (a.nchunks == 1) return launch_walk_lv<2, 2, false>(a, levels, blocks, st);
  if (a.carry == nullptr) return hipErrorInvalidValue;
  return launch_walk_lv<2, 2, true>(a, levels, blocks, st);
}

hipError_t launch_exp_tables(const double *g, int64_t count, const float *alphas, int n_alpha,
                             double *aux, hipStream_t st) {
  if (count <= 0 || n_alpha <= 0) return hipSuccess;
  const int bs = 256;
  hipLaunchKernelGGL(exp_tables_kernel, dim3((unsigned)((count + bs - 1) / bs)), dim3(bs), 0, st,
                     g, count, alphas, n_alpha, aux);
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
                                 int relative, double scale, double *out, hipStream_t st) {
  if (N <= 0 || T <= 0) return hipSuccess;
  hipLaunchKernelGGL(pathlen_lookup_kernel, dim3((unsigned)N), dim3(256), 0, st, X, D, T, norm,
                     relative, scale, out);
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

hipError_t launch_standardize(const double *X, int64_t rows, int64_t T, int div_std, double eps,
                              double *out, hipStream_t st) {
  if (rows <= 0 || T <= 0) return hipSuccess;
  hipLaunchKernelGGL(standardize_kernel, dim3((unsigned)rows), dim3(256), 0, st, X, T, div_std,
                     eps, out);
  return hipGetLastError();
}

}  // namespace fr

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
// DPP fetch of a double: the value of the source lane, 0.0 where the lane has no
// source.  FULL = all rows enabled: bound_ctrl supplies the zeros and no "old"
// value has to be materialised; otherwise disabled rows keep old = 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  if constexpr (ROW_MASK == 0xf) {
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  } else {
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  }
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

// P independent scans, advanced step by step so their DPP latencies overlap
template <int P>
__device__ __forceinline__ void wave_inclusive_scan_multi(double (&v)[P]) {
#pragma unroll
  for (int h = 0; h < P; ++h) v[h] += dpp_fetch<0x111, 0xf>(v[h]);
#pragma unroll
  for (int h = 0; h < P; ++h) v[h] += dpp_fetch<0x112, 0xf>(v[h]);
#pragma unroll
  for (int h = 0; h < P; ++h) v[h] += dpp_fetch<0x114, 0xf>(v[h]);
#pragma unroll
  for (int h = 0; h < P; ++h) v[h] += dpp_fetch<0x118, 0xf>(v[h]);
#pragma unroll
  for (int h = 0; h < P; ++h) v[h] += dpp_fetch<0x142, 0xa>(v[h]);
#pragma unroll
  for (int h = 0; h < P; ++h) v[h] += dpp_fetch<0x143, 0xc>(v[h]);
}

__device__ __forceinline__ double wave_shift_right1(double v) {
  return dpp_fetch<0x138, 0xf>(v);  // wave_shr:1, lane 0 gets 0.0
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

// ---------------------------------------------------------------- walk kernel
// Time layout of one chunk: wave w owns the contiguous span [w*SPAN, (w+1)*SPAN);
// the span is P pieces of 64*E elements; lane l holds E consecutive elements of
// every piece.  So every 16-byte global access of a wave is lane-contiguous
// (E = 2: 1 KiB per instruction) and only NW wave totals cross waves.
template <int E_, int P_, int MAXLV_, bool MULTI_, bool VEC_, bool WEIGHTED_, int TEAM_ = 4>
struct WalkCfg {
  // TEAM waves scan one row together.  TEAM = 4: the whole workgroup works on one
  // (series, group) unit and waves exchange totals through LDS once per node.
  // TEAM = 1: every wave scans whole rows alone (no barrier, no LDS exchange) and
  // the 4 waves of a workgroup walk 4 different groups of the SAME series, so they
  // still share the staged rows.
  static constexpr int TEAM = TEAM_;
  static constexpr int TEAMS = (kWalkThreads / 64) / TEAM_;
  static constexpr bool WEIGHTED = WEIGHTED_;  // exp tables in play (emit_mul / z_mul)
  static constexpr bool VEC = VEC_;      // 16-byte global accesses are aligned
  static constexpr int E = E_;          // contiguous elements per lane per piece
  static constexpr int P = P_;          // pieces per wave
  static constexpr int EP = E_ * P_;
  static constexpr int MAXLV = MAXLV_;
  static constexpr bool MULTI = MULTI_;  // more than one time chunk (carries in memory)
  static constexpr int NW = TEAM_;
  static constexpr int PIECE = 64 * E_;          // elements per wave piece
  static constexpr int SPAN = PIECE * P_;        // elements per wave
  static constexpr int CHUNK = SPAN * NW;        // elements per time chunk
};

// LDS position of chunk element i (i even: 16-byte units never straddle).  For
// E = 4 the two halves of a lane's 4 elements live in two planes so that every
// ds_read_b128 of a wave is lane-contiguous (bank-conflict free).
template <class C>
__device__ __forceinline__ int lds_pos(int i) {
  if constexpr (C::E == 2) {
    return i;
  } else {
    static_assert(C::E == 4, "E must be 2 or 4");
    const int blk = i >> 8, r = i & 255;
    return (blk << 8) + ((r & 2) << 6) + ((r >> 2) << 1) + (r & 1);
  }
}

struct WalkCtx {
  const IssArgs *a;
  const double *rows;   // LDS: staged rows [R][CHUNK]
  double *tot;          // LDS: wave totals [2][NW]
  double *out_base;     // out + n*out_n_stride + t0
  double *carry;        // carry slots of this series (multi-chunk) or nullptr
  int64_t t0;           // first time index of the chunk
  int tid, lane, wave, team;   // wave = index inside the team
  int buf;
  bool first_chunk;
  bool full_chunk;      // every element of the chunk is < T (no per-lane bounds checks)
#ifdef FRUITS_HIP_TIMING_BUILD
  unsigned long long seg[8];   // s_memtime sums per code segment (diagnostic build only)
  unsigned long long last;
#endif
};

#ifdef FRUITS_HIP_TIMING_BUILD
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP(cx, i)                                  \
  do {                                                \
    if ((cx).a->debug & 16) {                         \
      const unsigned long long t_ = stamp_now();      \
      (cx).seg[i] += t_ - (cx).last;                  \
      (cx).last = t_;                                 \
    }                                                 \
  } while (0)
#else
#define STAMP(cx, i) do { } while (0)
#endif

// reads the lane's EP elements of staged row `row`
template <class C>
__device__ __forceinline__ void read_row(const WalkCtx &cx, int row, double (&v)[C::EP]) {
  constexpr int E = C::E, P = C::P;
  const double *base = cx.rows + row * C::CHUNK + cx.wave * C::SPAN;
#pragma unroll
  for (int h = 0; h < P; ++h) {
    if constexpr (E == 2) {
      const vd2 q = *reinterpret_cast<const vd2 *>(base + h * C::PIECE + cx.lane * 2);
      v[h * 2] = q.x;
      v[h * 2 + 1] = q.y;
    } else {
      const vd2 q0 = *reinterpret_cast<const vd2 *>(base + h * C::PIECE + cx.lane * 2);
      const vd2 q1 = *reinterpret_cast<const vd2 *>(base + h * C::PIECE + 128 + cx.lane * 2);
      v[h * 4] = q0.x;
      v[h * 4 + 1] = q0.y;
      v[h * 4 + 2] = q1.x;
      v[h * 4 + 3] = q1.y;
    }
  }
}

template <class C>
__device__ __forceinline__ void block_scan(WalkCtx &cx, const double (&s)[C::EP],
                                           double (&c)[C::EP], double (&x)[C::EP],
                                           int carry_slot) {
  constexpr int E = C::E, P = C::P, NW = C::NW;
  double l[C::EP];
  double incl[P], excl[P], ptot[P];
#pragma unroll
  for (int h = 0; h < P; ++h) {
    l[h * E] = s[h * E];
#pragma unroll
    for (int e = 1; e < E; ++e) l[h * E + e] = l[h * E + e - 1] + s[h * E + e];
  }
#pragma unroll
  for (int h = 0; h < P; ++h) incl[h] = l[h * E + E - 1];
  wave_inclusive_scan_multi<P>(incl);
#pragma unroll
  for (int h = 0; h < P; ++h) {
    excl[h] = wave_shift_right1(incl[h]);
    ptot[h] = wave_last_lane(incl[h]);
  }
  double carry_in = 0.0;
  if constexpr (C::MULTI) {
    if (!cx.first_chunk) carry_in = cx.carry[carry_slot];
  }
  double base = 0.0;
  if constexpr (NW == 1) {
    STAMP(cx, 2);  // local sums + wave scans
    if constexpr (C::MULTI) {
      double total = ptot[0];
#pragma unroll
      for (int h = 1; h < P; ++h) total += ptot[h];
      base = carry_in;
      if (cx.lane == 0) cx.carry[carry_slot] = carry_in + total;
    }
  } else {
    double wave_total = ptot[0];
#pragma unroll
    for (int h = 1; h < P; ++h) wave_total += ptot[h];
    STAMP(cx, 2);  // local sums + wave scans
    double *tot = cx.tot + cx.buf * NW;
    if (cx.lane == 0) tot[cx.wave] = wave_total;
    lds_barrier();
    STAMP(cx, 3);  // LDS write + barrier
    // exclusive prefix of the wave totals
    static_assert(NW == 1 || NW == 4, "cross-wave prefix is written for 4 waves");
    const double t0 = tot[0], t1 = tot[1], t2 = tot[2], t3 = tot[3];
    const double p2 = t0 + t1, p3 = p2 + t2;
    base = cx.wave == 0 ? 0.0 : (cx.wave == 1 ? t0 : (cx.wave == 2 ? p2 : p3));
    cx.buf ^= 1;
    if constexpr (C::MULTI) {
      base += carry_in;
      // every wave stores the same value; a wave only ever re-reads its own store
      if (cx.lane == 0) cx.carry[carry_slot] = carry_in + (p3 + t3);
    }
  }
#pragma unroll
  for (int h = 0; h < P; ++h) {
    const double off = base + excl[h];
    x[h * E] = off;
    c[h * E] = off + l[h * E];
#pragma unroll
    for (int e = 1; e < E; ++e) {
      x[h * E + e] = c[h * E + e - 1];
      c[h * E + e] = off + l[h * E + e];
    }
    base += ptot[h];
  }
  STAMP(cx, 4);  // cross-wave prefix + final adds
}

template <class C>
__device__ __forceinline__ void emit_store(const WalkCtx &cx, const double (&v)[C::EP],
                                           double *dst) {
  constexpr int E = C::E, P = C::P;
  const int64_t T = cx.a->T;
#pragma unroll
  for (int h = 0; h < P; ++h) {
    const int idx = cx.wave * C::SPAN + h * C::PIECE + cx.lane * E;
#ifdef FRUITS_HIP_TIMING_BUILD
    // timing experiments only: keep the arithmetic alive, drop the stores
    if ((cx.a->debug & 1) && v[h * E] != 1.2345678e300) continue;
#endif
    if constexpr (C::VEC) {
#pragma unroll
      for (int e = 0; e < E; e += 2) {
        const vd2 val = {v[h * E + e], v[h * E + e + 1]};
        if (cx.full_chunk || cx.t0 + idx + e < T)
          *reinterpret_cast<vd2 *>(dst + idx + e) = val;
      }
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (cx.full_chunk || cx.t0 + idx + e < T) dst[idx + e] = v[h * E + e];
    }
  }
}

// a NodeRec in registers (all wave-uniform, i.e. SGPRs)
struct Rec {
  int32_t w[16];
  __device__ __forceinline__ int level() const { return w[0] & 0xff; }
  __device__ __forceinline__ int flags() const { return w[0] >> 8; }
  __device__ __forceinline__ int fac_count() const { return w[1]; }
  __device__ __forceinline__ int emit_count() const { return w[6]; }
  __device__ __forceinline__ int node_id() const { return w[9]; }
  __device__ __forceinline__ int emit_mul() const { return w[10]; }
  __device__ __forceinline__ int z_mul() const { return w[11]; }
  __device__ __forceinline__ int fac_begin() const { return w[12]; }
  __device__ __forceinline__ int emit_begin() const { return w[13]; }
};

__device__ __forceinline__ Rec load_rec(const NodeRec *recs, int pc) {
  // uniform, 64-byte aligned address in the constant address space -> one
  // s_load_dwordx16; issued one node ahead of its use (see walk)
  cptr<int32_t> q = as_const(reinterpret_cast<const int32_t *>(
      __builtin_assume_aligned(recs + pc, 64)));
  Rec r;
#pragma unroll
  for (int i = 0; i < 16; ++i) r.w[i] = q[i];
  return r;
}

template <class C>
__device__ __forceinline__ void mul_row(const WalkCtx &cx, int row, double (&s)[C::EP]) {
  double v[C::EP];
  read_row<C>(cx, row, v);
#pragma unroll
  for (int i = 0; i < C::EP; ++i) s[i] = s[i] * v[i];
}

// Letters with a reciprocal factor or more than kRecInlineFactors factors: walk
// the factor table (codes row | FAC_DIV), one factor at a time, in order.
template <class C>
__device__ __forceinline__ void slow_factors(const WalkCtx &cx, int fac_begin, int nf,
                                          double (&s)[C::EP]) {
  for (int f = 0; f < nf; ++f) {
    const int code = as_const(cx.a->factors)[fac_begin + f];
    double v[C::EP];
    read_row<C>(cx, code & FAC_ROW_MASK, v);
    if (code & FAC_DIV) {
#pragma unroll
      for (int i = 0; i < C::EP; ++i) s[i] = s[i] / v[i];
    } else {
#pragma unroll
      for (int i = 0; i < C::EP; ++i) s[i] = s[i] * v[i];
    }
  }
}

template <class C>
__device__ __forceinline__ void emit_all(const WalkCtx &cx, const Rec &nd,
                                         const double (&c)[C::EP]) {
  const IssArgs &a = *cx.a;
  const int ne = nd.emit_count();
  if (ne > 0) emit_store<C>(cx, c, cx.out_base + (int64_t)nd.w[7] * a.out_k_stride);
  if (ne > 1) {
    emit_store<C>(cx, c, cx.out_base + (int64_t)nd.w[8] * a.out_k_stride);
    for (int j = kRecInlineEmits; j < ne; ++j) {
      const int64_t k = as_const(a.emit_rows)[nd.emit_begin() + j];
      emit_store<C>(cx, c, cx.out_base + k * a.out_k_stride);
    }
  }
}

template <class C>
__device__ __forceinline__ void process_node(WalkCtx &cx, const Rec &nd,
                                             const double (&pin)[C::EP],
                                             double (&pout)[C::EP]) {
  constexpr int EP = C::EP;
  double s[EP];
#pragma unroll
  for (int i = 0; i < EP; ++i) s[i] = pin[i];
  const int nf = nd.fac_count();
  if (nd.flags() & F_SLOW) {
    slow_factors<C>(cx, nd.fac_begin(), nf, s);
  } else {
    // multiply-only letter, factors in the reference's order (ascending dimension)
    if (nf > 0) mul_row<C>(cx, nd.w[2], s);
    if (nf > 1) {
      mul_row<C>(cx, nd.w[3], s);
      if (nf > 2) mul_row<C>(cx, nd.w[4], s);
      if (nf > 3) mul_row<C>(cx, nd.w[5], s);
    }
  }
  STAMP(cx, 1);  // factors
  const bool has_children = (nd.flags() & F_CHILDREN) != 0;
  const int z_mul = nd.z_mul(), emit_mul = nd.emit_mul();
  const bool need2 = C::WEIGHTED && has_children && z_mul >= 0;
  const bool need1 = nd.emit_count() > 0 || (has_children && !need2);
#ifdef FRUITS_HIP_TIMING_BUILD
  if (cx.a->debug & 2) {  // timing experiments only: stores without the scan
    emit_all<C>(cx, nd, s);
#pragma unroll
    for (int i = 0; i < EP; ++i) pout[i] = s[i];
    return;
  }
#endif
  if (need1) {
    double c[EP], x[EP];
    block_scan<C>(cx, s, c, x, 2 * nd.node_id());
    if (nd.emit_count() > 0) {
      if (C::WEIGHTED && emit_mul >= 0) mul_row<C>(cx, emit_mul, c);
      emit_all<C>(cx, nd, c);
      STAMP(cx, 5);  // stores
    }
    if (has_children && !need2) {
#pragma unroll
      for (int i = 0; i < EP; ++i) pout[i] = x[i];
    }
  }
  if constexpr (C::WEIGHTED) {
    if (need2) {
      double s2[EP], c[EP], x[EP];
#pragma unroll
      for (int i = 0; i < EP; ++i) s2[i] = s[i];
      mul_row<C>(cx, z_mul, s2);
      block_scan<C>(cx, s2, c, x, 2 * nd.node_id() + 1);
#pragma unroll
      for (int i = 0; i < EP; ++i) pout[i] = x[i];
    }
  }
}

// Walks the records of one group.  cx.cur always holds the record at cx.pc; the
// record after it is requested BEFORE the current node is processed, so its
// scalar-memory latency hides behind the node's vector work.  The sentinel at
// the end of every group (level 0xff) terminates all loops.
template <class C, int LV>
__device__ __forceinline__ void walk(WalkCtx &cx, Rec &cur, int &pc,
                                     const double (&pin)[C::EP]) {
  const IssArgs &a = *cx.a;
  while (cur.level() == LV) {
    const Rec nd = cur;
    ++pc;
    cur = load_rec(a.recs, pc);
    double pout[C::EP];
    STAMP(cx, 0);  // interpreter: record decode / prefetch issue
    process_node<C>(cx, nd, pin, pout);
    // only children continue in place in this frame
    while (cur.level() == LV && (cur.flags() & F_CHAIN)) {
      const Rec nc = cur;
      ++pc;
      cur = load_rec(a.recs, pc);
      process_node<C>(cx, nc, pout, pout);
    }
    if constexpr (LV + 1 < C::MAXLV) {
      if (cur.level() == LV + 1) walk<C, LV + 1>(cx, cur, pc, pout);
    }
  }
}

template <class C>
__global__ __launch_bounds__(kWalkThreads) void iss_walk_kernel(const IssArgs a) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  WalkCtx cx;
  cx.a = &a;
  cx.rows = lds;
  cx.tot = lds + (int64_t)a.R * C::CHUNK;
  cx.tid = tid;
  cx.lane = tid & 63;
  {
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    cx.wave = w % C::TEAM;
    cx.team = w / C::TEAM;
  }
  cx.buf = 0;
  double *rows_w = lds;
  // TEAM = 1: a unit is a series, its G = TEAMS groups go to the 4 waves
  const int64_t units = C::TEAM == 1 ? a.N : a.N * a.G;
  bool first_unit = true;
#ifdef FRUITS_HIP_TIMING_BUILD
  if (a.debug & 4) return;
  for (int i = 0; i < 8; ++i) cx.seg[i] = 0;
  cx.last = stamp_now();
  const unsigned long long t_begin = cx.last;
#endif
  // Persistent workgroups: the grid holds one resident round of workgroups and
  // each walks units b, b + grid, ...  A unit is (series n, group g of sub-tries).
  // Workgroups are dealt round-robin over the 8 XCDs and the grid is a multiple
  // of 8, so with the mapping below all groups of one series meet in one XCD's
  // L2 (speed only, never correctness).
  for (int64_t u = blockIdx.x; u < units; u += gridDim.x) {
    int64_t n;
    int g;
    if constexpr (C::TEAM == 1) {
      n = u;
      g = cx.team;
    } else if (a.xcd_map) {
      const int64_t q = u >> 3, r = u & 7;
      n = (q / a.G) * 8 + r;
      g = (int)(q % a.G);
    } else {
      n = u / a.G;
      g = (int)(u % a.G);
    }
    cx.carry = a.carry ? a.carry + n * (2 * (int64_t)a.total_nodes) : nullptr;
    const int node_begin = as_const(a.group_begin)[g];
    for (int64_t chunk = 0; chunk < a.nchunks; ++chunk) {
      const int64_t t0 = chunk * C::CHUNK;
      cx.t0 = t0;
      cx.first_chunk = chunk == 0;
      cx.full_chunk = t0 + C::CHUNK <= a.T;
      cx.out_base = a.out + n * a.out_n_stride + t0;
      if (!first_unit || chunk > 0) lds_barrier();  // all reads of the old rows are done
      // stage the referenced rows of this chunk: coalesced 16-byte units, the
      // loads of up to 4 rows in flight before the first LDS write
#ifdef FRUITS_HIP_TIMING_BUILD
      if (!(a.debug & 8))
#endif
      for (int r0 = 0; r0 < a.R; r0 += 4) {
        constexpr int U = C::CHUNK / 2 / kWalkThreads;
        vd2 v[4][U];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          if (r0 + rr < a.R) {
            const int src = as_const(a.row_src)[r0 + rr];
            const double *gp =
                src >= 0 ? a.X + (n * a.D + src) * a.T
                         : a.aux + (int64_t)(-src - 1) * a.aux_tab_stride + n * a.aux_n_stride;
#pragma unroll
            for (int k = 0; k < U; ++k) {
              const int i = 2 * (k * kWalkThreads + tid);
              const int64_t t = t0 + i;
              v[rr][k] = vd2{0.0, 0.0};
              if constexpr (C::VEC) {
                if (cx.full_chunk || t < a.T) v[rr][k] = *reinterpret_cast<const vd2 *>(gp + t);
              } else {
                if (t < a.T) v[rr][k].x = gp[t];
                if (t + 1 < a.T) v[rr][k].y = gp[t + 1];
              }
            }
          }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          if (r0 + rr < a.R) {
#pragma unroll
            for (int k = 0; k < U; ++k) {
              const int i = 2 * (k * kWalkThreads + tid);
              *reinterpret_cast<vd2 *>(rows_w + (r0 + rr) * C::CHUNK + lds_pos<C>(i)) = v[rr][k];
            }
          }
        }
      }
      __syncthreads();
      STAMP(cx, 6);  // staging
      double ones[C::EP];
#pragma unroll
      for (int i = 0; i < C::EP; ++i) ones[i] = 1.0;
      int pc = node_begin;
      Rec cur = load_rec(a.recs, pc);
      walk<C, 0>(cx, cur, pc, ones);
    }
    first_unit = false;
  }
#ifdef FRUITS_HIP_TIMING_BUILD
  if ((a.debug & 16) && a.dbg != nullptr && cx.lane == 0) {
    unsigned long long *o = a.dbg + ((int64_t)blockIdx.x * 4 + cx.team * C::TEAM + cx.wave) * 10;
    for (int i = 0; i < 8; ++i) o[i] = cx.seg[i];
    o[8] = stamp_now() - t_begin;
    o[9] = t_begin;
  }
#endif
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
static int device_cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

template <int E, int P, int LV, bool MULTI, bool VEC, bool W, int TEAM = 4>
static hipError_t launch_walk_cfg(const IssArgs &a, hipStream_t st) {
  using C = WalkCfg<E, P, LV, MULTI, VEC, W, TEAM>;
  const size_t lds = ((size_t)a.R * C::CHUNK + 2 * C::NW) * sizeof(double);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static size_t lds_attr = 0;  // per instantiation
  if (lds > 64 * 1024 && lds > lds_attr) {
    hipError_t e = hipFuncSetAttribute((const void *)iss_walk_kernel<C>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    lds_attr = lds;
  }
  const int64_t units = TEAM == 1 ? a.N : a.N * a.G;
  int64_t blocks = units;
  if (a.persistent) {
    // one resident round of workgroups (a multiple of 8 for the XCD mapping)
    static size_t cached_lds = (size_t)-1;
    static int per_cu = 0;
    if (cached_lds != lds) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, iss_walk_kernel<C>, kWalkThreads,
                                                       lds) != hipSuccess || nb < 1)
        nb = 1;
      per_cu = nb;
      cached_lds = lds;
    }
    int64_t resident = (int64_t)per_cu * device_cu_count();
    resident -= resident % 8;
    if (resident < 8) resident = 8;
    if (blocks > resident) blocks = resident;
  }
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(iss_walk_kernel<C>, dim3((unsigned)blocks), dim3(kWalkThreads), lds, st, a);
  return hipGetLastError();
}

template <int E, int P, int LV, bool MULTI, bool VEC>
static hipError_t launch_walk_w(const IssArgs &a, hipStream_t st) {
  return a.aux ? launch_walk_cfg<E, P, LV, MULTI, VEC, true>(a, st)
               : launch_walk_cfg<E, P, LV, MULTI, VEC, false>(a, st);
}

template <int E, int P, bool MULTI, bool VEC>
static hipError_t launch_walk_lv(const IssArgs &a, int levels, hipStream_t st) {
  if (levels <= 2) return launch_walk_w<E, P, 2, MULTI, VEC>(a, st);
  if (levels <= 4) return launch_walk_w<E, P, 4, MULTI, VEC>(a, st);
  if (levels <= 8) return launch_walk_w<E, P, 8, MULTI, VEC>(a, st);
  return launch_walk_w<E, P, kMaxLevels, MULTI, VEC>(a, st);
}

template <int E, int P>
static hipError_t launch_walk_ep(const IssArgs &a, int levels, hipStream_t st) {
  const bool multi = a.nchunks > 1;
  if (multi && a.carry == nullptr) return hipErrorInvalidValue;
  if (multi)
    return a.vec_ok ? launch_walk_lv<E, P, true, true>(a, levels, st)
                    : launch_walk_lv<E, P, true, false>(a, levels, st);
  return a.vec_ok ? launch_walk_lv<E, P, false, true>(a, levels, st)
                  : launch_walk_lv<E, P, false, false>(a, levels, st);
}

int walk_chunk_elems(int64_t T) { return T <= 512 ? 512 : 1024; }

// wave-per-row variant (TEAM = 1): single chunk, aligned 16-byte accesses,
// shallow tries (register frames of 2 * E * P VGPRs per level)
bool wave_rows_supported(int64_t T, int levels, bool vec_ok) {
  return vec_ok && T <= 1024 && levels <= 4;
}

template <int P>
static hipError_t launch_wave_rows(const IssArgs &a, int levels, hipStream_t st) {
  if (levels <= 2)
    return a.aux ? launch_walk_cfg<2, P, 2, false, true, true, 1>(a, st)
                 : launch_walk_cfg<2, P, 2, false, true, false, 1>(a, st);
  return a.aux ? launch_walk_cfg<2, P, 4, false, true, true, 1>(a, st)
               : launch_walk_cfg<2, P, 4, false, true, false, 1>(a, st);
}

hipError_t launch_iss_walk(IssArgs &a, int levels, hipStream_t st) {
  const int chunk = walk_chunk_elems(a.T);
  a.nchunks = (int32_t)((a.T + chunk - 1) / chunk);
  if (a.N * a.G <= 0) return hipSuccess;
  if (a.wave_rows) {
    if (a.G != 4 || !wave_rows_supported(a.T, levels, a.vec_ok != 0)) return hipErrorInvalidValue;
    return chunk == 512 ? launch_wave_rows<4>(a, levels, st) : launch_wave_rows<8>(a, levels, st);
  }
  if (chunk == 512) return launch_walk_ep<2, 1>(a, levels, st);
  return launch_walk_ep<2, 2>(a, levels, st);
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

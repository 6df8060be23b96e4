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
#pragma once
#include "walk_types.h"
#include "walk_scan.h"

namespace fr {

// ---------------------------------------------------------------- walk kernel
// Time layout of one chunk: wave w owns the contiguous span [w*SPAN, (w+1)*SPAN);
// the span is P pieces of 64*E elements; lane l holds E consecutive elements of
// every piece.  So every 16-byte global access of a wave is lane-contiguous
// (E = 2: 1 KiB per instruction) and only NW wave totals cross waves.
template <int E_, int P_, int MAXLV_, int MULTI_, bool VEC_, bool WEIGHTED_, int TEAM_ = 4,
          int MODE_ = 0, int SEMI_ = 0, bool NT_ = false, bool TOTALINC_ = false,
          bool HIGHORD_ = false>
struct WalkCfg {
  // TOTALINC: fused epilogue of a TOTALLY weighted plan with differencing sieves - the
  // increments need the weight one step to the left (previous_weighted).  Its own
  // instantiation: compiled into the common weighted kernels it costs them 130-400 more
  // SGPR spills and 20 % of their speed (config 4: 21.5 -> 26.5 ms).
  static constexpr bool TOTALINC = TOTALINC_;
  // HIGHORD: fused epilogue with differencing orders >= 3 on series of SEVERAL time chunks (a
  // carry per order between chunks).  Its own instantiation like TOTALINC: compiled into the
  // common multi-chunk kernels it took them past 128 VGPRs (config 5: 24.9 -> 30.5 ms).
  static constexpr bool HIGHORD = HIGHORD_;
  // NT: the rows of X are staged with non-temporal loads (load_input; the host asks for it
  // when every row is read once per launch and the batch is about the size of the cache)
  static constexpr bool NT = NT_;
  // SEMI 0: Reals (+, x) with an exclusive shift between letters; SEMI 1: Arctic
  // (max, +), letters add el * x and children continue from the INCLUSIVE maximum;
  // SEMI 2: Bayesian (max, x): the letters and weights of Reals, the scan of Arctic
  static constexpr int SEMI = SEMI_;
  // MODE 0: write the (K,N,T) tensor.  MODE 1: fused sieve epilogue - the values of
  // a node go straight into NPI / MPI / END features, no tensor is written.
  static constexpr int MODE = MODE_;
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
  // more than one time chunk: 0 no, 1 per-node carries in LDS, 2 carries in global memory
  // (a global carry load is a vector load: waiting for it also waits for every output
  // store in flight, so LDS is preferred whenever the group's carries fit)
  static constexpr int MULTI = MULTI_;
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

constexpr int kCarrySlots = 3;  // per node: scan, second scan (non-total), d1 tail (inc = 2)
// fused walk, series of several chunks (WalkCfg::HIGHORD): a slot pair per differencing order
// 3 .. 8 at kCarrySlots + 2 (k - 3), and per cumulation 1 .. 8 (inc < 0) from kCumCarryBase on
constexpr int kCumCarryBase = kCarrySlots + 2 * 6;
constexpr int kCarrySlotsAll = kCumCarryBase + 2 * 8;
constexpr int kStageRows = 3;  // rows staged per batch (registers: 4 * U VGPRs per row)

struct WalkCtx {
  const IssArgs *a;
  const double *rows;   // LDS: staged rows [R][CHUNK]
  double *tot;          // LDS: wave totals [2][NW]
  double *tail;         // LDS: last first-difference of every wave [2][NW] (fused inc = 2)
  int tail_buf;
  int slot;             // carry slot base of the node being processed
  int parity;           // fused walk: time chunk & 1 (the carries of differencing orders >= 3 have two
                        // buffers: several ops of a node read the old one and write the same new one)
  double *out_base;     // out + n*out_n_stride + t0
  double *feat_row;     // MODE 1: feats + n*feat_stride
  double *cnt_row;      // MODE 1: band population of MPI features
  // MODE 1, cooperative kernels: the features of a unit accumulate in an LDS window (value,
  // MPI population, column of every slot) and leave with plain stores (feat_flush); a unit
  // owns its feature columns, so nothing in global memory is ever added to atomically
  lds_f64 *fl_val;
  lds_f64 *fl_cnt;
  lds_i32 *fl_col;
  int fslot;            // first window slot of the node being processed
  int fused_used;       // slots of the window in use
  int frow0;            // fused walk: position in IssArgs::slot_rows of the window's first output row
  int feat_window;      // slots of the window (a huge number when every group's features fit)
  const int32_t *cut_row;  // MODE 1: this series' row of IssArgs::series_cuts (or nullptr)
  int64_t series;          // MODE 1: index of the series being walked
  double *carry;        // carry slots of this series (multi-chunk; LDS or global) or nullptr
  int64_t t0;           // first time index of the chunk
  int tid, lane, wave, team;   // wave = index inside the team
  int pc_begin;                // first record of the group being walked
  int buf;
  int next_unit;               // static programs: the series whose rows are loaded next
  bool have_rows;              // static programs: the unit's rows are already in registers
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

// 16 bytes of an input row.  With one group per series every input element is read exactly
// once per launch: a NON-TEMPORAL load then keeps it from allocating in the Infinity Cache,
// where it would only evict output lines (measured with a pure data mover of the headline's
// traffic, tools/stream_mix.hip: 64 -> 51 us at 352 MB; no difference beyond 1 GB).
// Compile-time: a run-time select of the two loads is folded into one plain load.
template <bool NT>
__device__ __forceinline__ vd2 load_input(const double *p) {
  const vd2 *q = reinterpret_cast<const vd2 *>(p);
  if constexpr (NT) return __builtin_nontemporal_load(q);
  return *q;
}

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
    for (int e = 1; e < E; ++e) l[h * E + e] = semi_add<C::SEMI>(l[h * E + e - 1], s[h * E + e]);
  }
#pragma unroll
  for (int h = 0; h < P; ++h) incl[h] = l[h * E + E - 1];
  wave_inclusive_scan_multi<P, C::SEMI>(incl);
#pragma unroll
  for (int h = 0; h < P; ++h) {
    excl[h] = wave_shift_right1<C::SEMI>(incl[h]);
    ptot[h] = wave_last_lane(incl[h]);
  }
  double carry_in = semi_zero<C::SEMI>();
  if constexpr (C::MULTI != 0) {
    if (!cx.first_chunk) carry_in = cx.carry[carry_slot];
  }
  double base = semi_zero<C::SEMI>();
  if constexpr (NW == 1) {
    STAMP(cx, 2);  // local sums + wave scans
    if constexpr (C::MULTI != 0) {
      double total = ptot[0];
#pragma unroll
      for (int h = 1; h < P; ++h) total = semi_add<C::SEMI>(total, ptot[h]);
      base = carry_in;
      if (cx.lane == 0) cx.carry[carry_slot] = semi_add<C::SEMI>(carry_in, total);
    }
  } else {
    double wave_total = ptot[0];
#pragma unroll
    for (int h = 1; h < P; ++h) wave_total = semi_add<C::SEMI>(wave_total, ptot[h]);
    STAMP(cx, 2);  // local sums + wave scans
    double *tot = cx.tot + cx.buf * NW;
    if (cx.lane == 0) tot[cx.wave] = wave_total;
#ifdef FRUITS_HIP_TIMING_BUILD
    if (cx.a->debug & 32)  // timing experiments only: the cost of the rendezvous itself
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else
#endif
    lds_barrier();
    STAMP(cx, 3);  // LDS write + barrier
    // exclusive prefix of the wave totals
    static_assert(NW == 1 || NW == 4, "cross-wave prefix is written for 4 waves");
    const double t0 = tot[0], t1 = tot[1], t2 = tot[2], t3 = tot[3];
    const double p2 = semi_add<C::SEMI>(t0, t1), p3 = semi_add<C::SEMI>(p2, t2);
    base = cx.wave == 0 ? semi_zero<C::SEMI>() : (cx.wave == 1 ? t0 : (cx.wave == 2 ? p2 : p3));
    cx.buf ^= 1;
    if constexpr (C::MULTI == 1) {
      // LDS carry: every wave read it before the barrier above; one lane updates it
      base = semi_add<C::SEMI>(base, carry_in);
      if (cx.wave == 0 && cx.lane == 0)
        cx.carry[carry_slot] = semi_add<C::SEMI>(carry_in, semi_add<C::SEMI>(p3, t3));
    } else if constexpr (C::MULTI == 2) {
      base = semi_add<C::SEMI>(base, carry_in);
      // every wave stores the same value; a wave only ever re-reads its own store
      if (cx.lane == 0)
        cx.carry[carry_slot] = semi_add<C::SEMI>(carry_in, semi_add<C::SEMI>(p3, t3));
    }
  }
  // The last value of a lane is formed as base + (inclusive wave scan), the first
  // exclusive value of the NEXT lane as base + (that same scan value, shifted): the
  // two are bit-identical, so the stored row and the exclusive prefixes handed to
  // children / sieves are consistent across lanes and pieces (an increment that the
  // running sum absorbs is exactly 0, as in a sequential cumsum).
#pragma unroll
  for (int h = 0; h < P; ++h) {
    const double off = semi_add<C::SEMI>(base, excl[h]);
    x[h * E] = off;
#pragma unroll
    for (int e = 0; e + 1 < E; ++e) {
      c[h * E + e] = semi_add<C::SEMI>(off, l[h * E + e]);
      x[h * E + e + 1] = c[h * E + e];
    }
    c[h * E + E - 1] = semi_add<C::SEMI>(base, incl[h]);
    base = semi_add<C::SEMI>(base, ptot[h]);
  }
  STAMP(cx, 4);  // cross-wave prefix + final adds
}

template <class C>
__device__ __forceinline__ void emit_store(const WalkCtx &cx, const double (&v)[C::EP],
                                           double *dst) {
  constexpr int E = C::E, P = C::P;
  const int64_t T = cx.a->T;
#ifdef FRUITS_HIP_TIMING_BUILD
  // timing experiments only: keep the arithmetic alive, drop the stores
  if ((cx.a->debug & 1) && v[0] != 1.2345678e300) return;
#endif
  if constexpr (C::VEC) {
    if (cx.full_chunk) {
      // the common case, decided once per chunk: no per-lane bounds checks (each would
      // cost an exec-mask save / branch / restore around every store)
#pragma unroll
      for (int h = 0; h < P; ++h) {
        const int idx = cx.wave * C::SPAN + h * C::PIECE + cx.lane * E;
#pragma unroll
        for (int e = 0; e < E; e += 2)
          *reinterpret_cast<vd2 *>(dst + idx + e) = vd2{v[h * E + e], v[h * E + e + 1]};
      }
      return;
    }
  }
#pragma unroll
  for (int h = 0; h < P; ++h) {
    const int idx = cx.wave * C::SPAN + h * C::PIECE + cx.lane * E;
    if constexpr (C::VEC) {
#pragma unroll
      for (int e = 0; e < E; e += 2) {
        const vd2 val = {v[h * E + e], v[h * E + e + 1]};
        if (cx.t0 + idx + e < T) *reinterpret_cast<vd2 *>(dst + idx + e) = val;
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
  __device__ __forceinline__ int fac_count() const { return w[1] & 0xffff; }
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

// s (x)= factor.  Reals: s *= row.  Arctic: s += el * row with the multiplier in bits
// 8-15 of the code; the product is rounded before the add like the reference's
// `tmp + el * Z[dim]` (no fused multiply-add), so max-plus results are bit-exact.
template <class C>
__device__ __forceinline__ void mul_row(const WalkCtx &cx, int code, double (&s)[C::EP]) {
  double v[C::EP];
  read_row<C>(cx, code & FAC_ROW_MASK, v);
  if constexpr (C::SEMI != 1) {
#pragma unroll
    for (int i = 0; i < C::EP; ++i) s[i] = s[i] * v[i];
  } else {
#pragma clang fp contract(off)  // the product must round before the add (no FMA)
    const double el = (double)(int)(int8_t)((code >> 8) & 0xff);
#pragma unroll
    for (int i = 0; i < C::EP; ++i) {
      const double prod = el * v[i];
      s[i] = s[i] + prod;
    }
  }
}

// Letters with a reciprocal factor or more than kRecInlineFactors factors: walk
// the factor table (codes row | FAC_DIV), one factor at a time, in order.
template <class C>
__device__ __forceinline__ void slow_factors(const WalkCtx &cx, int fac_begin, int nf,
                                          double (&s)[C::EP], const double (&pin)[C::EP],
                                          bool letter_sum) {
  bool folded = !letter_sum;
  for (int f = 0; f < nf; ++f) {
    const int code = as_const(cx.a->factors)[fac_begin + f];
    if constexpr (C::SEMI == 1) {
      if (!folded && (code & FAC_FOLD)) {   // prefix + (sum of the letter's terms), then weights
#pragma unroll
        for (int i = 0; i < C::EP; ++i) s[i] = pin[i] + s[i];
        folded = true;
      }
      mul_row<C>(cx, code, s);
      continue;
    }
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
  if constexpr (C::SEMI == 1) {
    if (!folded) {
#pragma unroll
      for (int i = 0; i < C::EP; ++i) s[i] = pin[i] + s[i];
    }
  }
}

// Address of output row k of the current series / chunk.  Rows are at most 4 GiB apart in
// every layout the host code uses, so the byte offset is ONE 32 x 32 -> 64 bit scalar
// multiply (the general 64 x 64 product costs ten scalar instructions per emitted row).
__device__ __forceinline__ double *emit_ptr(const WalkCtx &cx, int k) {
  const IssArgs &a = *cx.a;
  if (a.k_stride_bytes32 != 0) {
    const uint64_t off = (uint64_t)(uint32_t)k * (uint64_t)a.k_stride_bytes32;
    return reinterpret_cast<double *>(reinterpret_cast<char *>(cx.out_base) + off);
  }
  return cx.out_base + (int64_t)k * a.out_k_stride;
}

// `more` / `n_more`: further output rows of a STATIC program's node (compile-time constants,
// kSchedEmits entries); the interpreter reads rows beyond the inline two from the table
template <class C>
__device__ __forceinline__ void emit_all(const WalkCtx &cx, const Rec &nd,
                                         const double (&c)[C::EP], const int32_t *more = nullptr,
                                         int n_more = 0) {
  const IssArgs &a = *cx.a;
  const int ne = nd.emit_count();
  if (ne > 0) emit_store<C>(cx, c, emit_ptr(cx, nd.w[7]));
  if (ne > 1) {
    emit_store<C>(cx, c, emit_ptr(cx, nd.w[8]));
    for (int j = kRecInlineEmits; j < ne; ++j)
      emit_store<C>(cx, c, emit_ptr(cx, as_const(a.emit_rows)[nd.emit_begin() + j]));
  }
  for (int j = 0; j < n_more; ++j) emit_store<C>(cx, c, emit_ptr(cx, more[j]));
}

// ---------------------------------------------------------------- fused sieves
// Features of ONE output row k from the node's inclusive values c and their
// exclusive shifts x (x[t] = c[t-1]): what IncrementSieve._pre_transform +
// NPI/MPI._backend (fruits/sieving/increment.py:63-71,107-163) and END._transform
// (fruits/sieving/segment.py:210-219) compute on the materialised row.  The host
// flattens the sieves of row k into n_ops fixed-size "feature ops" (FeatOp) whose
// cuts and fitted thresholds are already resolved, so one scalar load per op
// brings everything and there are no dependent table look-ups.
struct Ops2 {
  int32_t w[16];  // two FeatOps
};

__device__ __forceinline__ Ops2 load_ops2(const IssArgs &a, int64_t k, int first) {
  cptr<int32_t> q = as_const(reinterpret_cast<const int32_t *>(
      __builtin_assume_aligned(a.ops + (k * a.n_ops_padded + first), 64)));
  Ops2 o;
#pragma unroll
  for (int i = 0; i < 16; ++i) o.w[i] = q[i];
  return o;
}

__device__ __forceinline__ double bits_to_double(int lo, int hi) {
  return __hiloint2double(hi, lo);
}

// First differences d1[t] = c[t] - c[t-1] of the element BEFORE each of the lane's
// elements (second differences need them): inside a lane the neighbour, across lanes
// a DPP shift, across pieces lane 63, across waves an LDS exchange (one extra barrier,
// taken by every wave since the op list is uniform), across chunks a carry slot.
// `carry_at`: the node's carry slot of this exchange (2: the first differences' - advanced once
// per node and chunk, FusedScratch); `two_buffers` (orders >= 3 of the fused walk): the slot pair
// carry_at, carry_at + 1 - this chunk reads buffer `parity` and writes the other.
template <class C>
__device__ __forceinline__ void prev_first_differences(WalkCtx &cx, const double (&d1)[C::EP],
                                                       double (&dp)[C::EP], int carry_at = 2,
                                                       bool two_buffers = false) {
  constexpr int E = C::E, P = C::P, NW = C::NW;
  const int rd = cx.slot + carry_at + (two_buffers ? cx.parity : 0);
  const int wr = cx.slot + carry_at + (two_buffers ? (cx.parity ^ 1) : 0);
  (void)rd;
  (void)wr;
  double last[P];
#pragma unroll
  for (int h = 0; h < P; ++h) {
    last[h] = wave_last_lane(d1[h * E + E - 1]);
    const double from_left = wave_shift_right1<0>(d1[h * E + E - 1]);
    dp[h * E] = from_left;
#pragma unroll
    for (int e = 1; e < E; ++e) dp[h * E + e] = d1[h * E + e - 1];
  }
  double before_wave = 0.0;  // d1 of the element just before this wave's span
  if constexpr (NW > 1) {
    double *tl = cx.tail + cx.tail_buf * NW;
    if (cx.lane == 0) tl[cx.wave] = last[P - 1];
    lds_barrier();
    if (cx.wave > 0) before_wave = tl[cx.wave - 1];
    cx.tail_buf ^= 1;
    if constexpr (C::MULTI != 0) {
      const double carried = cx.first_chunk ? 0.0 : cx.carry[rd];
      if (cx.wave == 0) before_wave = carried;
      const double chunk_last = tl[NW - 1];
      if (C::MULTI == 1 ? (cx.wave == 0 && cx.lane == 0) : (cx.lane == 0))
        cx.carry[wr] = chunk_last;
    }
  } else if constexpr (C::MULTI != 0) {
    before_wave = cx.first_chunk ? 0.0 : cx.carry[rd];
    if (cx.lane == 0) cx.carry[wr] = last[P - 1];
  }
  if (cx.lane == 0) {
    dp[0] = before_wave;
#pragma unroll
    for (int h = 1; h < P; ++h) dp[h * E] = last[h - 1];
  }
}

// per-node scratch of the epilogue: the previous first differences are computed (and
// their chunk carry advanced) at most once per node, whatever the number of inc = 2 ops
template <int EP>
struct FusedScratch {
  double dp[EP];
  bool have_dp = false;
};

// `s` is the scan input the values came from (c = cumsum(s)) when seq_steps.  Then a
// Reals first difference is formed as fl(x + s) - x, the step a SEQUENTIAL cumsum
// takes from the same prefix (np.cumsum in the reference): a summand the running sum
// absorbs gives an increment of exactly 0, as in the reference, where the difference of
// two parallel-scan values is +-1 ulp of noise - which "number of positive increments"
// (NPI with its default q = (0, 1)) would count at random.
template <class C>
__device__ __forceinline__ void fused_op(WalkCtx &cx, const int32_t *w, int slot,
                                         const double (&c)[C::EP], const double (&x)[C::EP],
                                         const double (&s)[C::EP], bool seq_steps,
                                         FusedScratch<C::EP> &sc) {
  constexpr int E = C::E, P = C::P, EP = C::EP;
  const int kind = w[0] & 0xff, inc = (int)(int8_t)((w[0] >> 8) & 0xff), col = w[1];   // inc: signed
  // per-series cuts (coquantile positions): lo / hi name slots of the series' cut row
  const bool series_cuts = (w[0] >> 16) & 1;
  if constexpr (C::TEAM != 1) {
    // (the flush needs the column of every slot, also of one no lane adds to)
    if (cx.wave == 0 && cx.lane == 0) cx.fl_col[slot] = col;
  }
  if (kind == FR_SIEVE_END_K) {
    int pick = w[2];                    // index of the value to pick
    if (series_cuts) {                  // X[:, cut - 1], index -1 wrapping like numpy
      pick = as_const(cx.cut_row)[w[2]] - 1;
      if (pick < 0) pick += (int)cx.a->T;
    }
    const int rel = pick - (int)cx.t0;
    if (rel >= 0 && rel < C::CHUNK) {
      const int wv = rel / C::SPAN;
      if (cx.wave == wv) {
        // the position is uniform: pick the register with uniform selects (static register
        // indices: a computed index into c[] would send the array through scratch), then ONE
        // lane stores it
        const int r = rel - wv * C::SPAN;
        const int hu = r / C::PIECE, q = r - hu * C::PIECE;
        const int owner = q / E, eu = q - owner * E;
        double v = c[0];
#pragma unroll
        for (int h = 0; h < P; ++h)
#pragma unroll
          for (int e = 0; e < E; ++e)
            if (h * E + e > 0 && hu == h && eu == e) v = c[h * E + e];
        if (cx.lane == owner) {
          if constexpr (C::TEAM != 1) cx.fl_val[slot] = v;
          else cx.feat_row[col] = v;
        }
      }
    }
    return;
  }
  int lo = w[2], hi = w[3];
  if (series_cuts) {
    lo = as_const(cx.cut_row)[w[2]];
    hi = as_const(cx.cut_row)[w[3]];
  }
  const double qlo = bits_to_double(w[4], w[5]), qhi = bits_to_double(w[6], w[7]);
  const int t_first = (int)cx.t0 + cx.wave * C::SPAN + cx.lane * E;  // element (h=0, e=0)
  double d[EP];
  if (inc <= 0) {
#pragma unroll
    for (int i = 0; i < EP; ++i) d[i] = c[i];
    if constexpr (C::MULTI == 0) {
      // inc < 0: the row cumulated -inc times (np.cumsum, fruits/sieving/increment.py:68-70) -
      // a plain sum whatever the semiring of the plan; one-chunk series only (the host
      // refuses longer ones: every cumulation would need its own carry)
      using CR = WalkCfg<C::E, C::P, C::MAXLV, 0, C::VEC, C::WEIGHTED, C::TEAM, C::MODE, 0>;
      for (int k = inc; k < 0; ++k) {
        double cs[EP], xs[EP];
        block_scan<CR>(cx, d, cs, xs, 0);
#pragma unroll
        for (int i = 0; i < EP; ++i) d[i] = cs[i];
      }
    }
  } else {
    // increments are zero-padded at t = 0 (fruits/cache.py:8-13)
#pragma unroll
    for (int h = 0; h < P; ++h)
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int i = h * E + e;
        double step = c[i] - x[i];
        if (C::SEMI == 0 && seq_steps) step = (x[i] + s[i]) - x[i];
        d[i] = (t_first + h * C::PIECE + e == 0) ? 0.0 : step;
      }
    if (inc >= 2) {
      if (!sc.have_dp) {
        prev_first_differences<C>(cx, d, sc.dp);
        sc.have_dp = true;
      }
#pragma unroll
      for (int h = 0; h < P; ++h)
#pragma unroll
        for (int e = 0; e < E; ++e)
          d[h * E + e] = (t_first + h * C::PIECE + e == 0) ? 0.0 : d[h * E + e] - sc.dp[h * E + e];
      if constexpr (C::MULTI == 0) {
        // third to eighth differences (IncrementSieve._pre_transform applies _increments inc
        // times, fruits/sieving/increment.py:63-71): one more neighbour exchange per order.
        // Single-chunk series only (the host refuses longer ones: an order needs its own
        // carry between chunks).
        for (int k = 3; k <= inc; ++k) {
          double dq[EP];
          prev_first_differences<C>(cx, d, dq);
#pragma unroll
          for (int h = 0; h < P; ++h)
#pragma unroll
            for (int e = 0; e < E; ++e)
              d[h * E + e] = (t_first + h * C::PIECE + e == 0) ? 0.0 : d[h * E + e] - dq[h * E + e];
        }
      }
    }
  }
  int cnt = 0;
  double sum = 0.0;
#pragma unroll
  for (int h = 0; h < P; ++h)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int t = t_first + h * C::PIECE + e;
      const double v = d[h * E + e];
      const bool in = t >= lo && t < hi && qlo < v && v <= qhi;
      cnt += __popcll(__ballot(in));
      if (kind == FR_SIEVE_MPI_K) sum += in ? v : 0.0;
    }
  // every op leaves its result, zero counts included: every feature column is written exactly
  // once per series and the feature tensor needs no clearing
  if (kind == FR_SIEVE_MPI_K)
    sum = wave_last_lane(wave_inclusive_scan<0>(sum));  // wave total by DPP (no LDS permutes)
  if (cx.lane == 0) {
    if constexpr (C::TEAM != 1) {
      // one LDS add per wave (ds_add_f64, nothing returned)
      if (kind == FR_SIEVE_MPI_K) {
        lds_add(cx.fl_val + slot, sum);
        lds_add(cx.fl_cnt + slot, (double)cnt);
      } else {
        lds_add(cx.fl_val + slot, (double)cnt);
      }
    } else {
      // wave-per-unit kernels: the wave holds the whole (single-chunk) row
      if (kind == FR_SIEVE_MPI_K) {
        cx.feat_row[col] = sum;
        cx.cnt_row[col] = (double)cnt;
      } else {
        cx.feat_row[col] = (double)cnt;
      }
    }
  }
}

// (The IssArgs struct is the only kernel argument of every kernel that flushes: it starts the
// kernel-argument segment.)
// Flushes the LDS feature window of a cooperative kernel: slot s -> column fl_col[s] of the
// series' feature row with plain stores (`add`: onto what earlier time chunks of the unit left
// there - the unit owns its columns, so a plain read-modify-write by the thread that wrote them),
// and clears the window.  The next adds come behind the next node's scan barrier.
// TABLE (the fused walk): no columns in the window - a slot is (output row r of the unit's walk,
// op i), its column the one the row's i-th op names (slot_rows, GroupedProgram).
template <class C, bool TABLE = false>
__device__ __forceinline__ void feat_flush(WalkCtx &cx, bool add) {
  lds_barrier();   // the adds of every wave are in
  // (the kernel arguments through a pointer the optimiser cannot look through: loaded here,
  // where they are used, not kept in scalar registers over the whole walk)
  cptr<IssArgs> ap = (cptr<IssArgs>)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ap));
  const bool mpi = ap->has_mpi != 0;
  double *feat_row = ap->feats + cx.series * ap->feat_stride;
  double *cnt_row = ap->cnt + cx.series * ap->feat_stride;
  const int n_ops = ap->n_ops;
  for (int sl = cx.tid; sl < cx.fused_used; sl += kWalkThreads) {
    int col;
    if constexpr (TABLE) {
      const int r = sl / n_ops, i = sl - r * n_ops;
      // (a plan in pieces numbers its output rows in walk order: no table)
      const int64_t k = ap->slot_rows != nullptr ? as_const(ap->slot_rows)[cx.frow0 + r] : cx.frow0 + r;
      col = as_const(ap->ops)[k * ap->n_ops_padded + i].col;
    } else {
      col = cx.fl_col[sl];
    }
    double v = cx.fl_val[sl];
    if (add) v = feat_row[col] + v;
    feat_row[col] = v;
    cx.fl_val[sl] = 0.0;
    if (mpi) {
      double n = cx.fl_cnt[sl];
      if (add) n = cnt_row[col] + n;
      cnt_row[col] = n;
      cx.fl_cnt[sl] = 0.0;
    }
  }
  if constexpr (TABLE) cx.frow0 += cx.fused_used / n_ops;
  cx.fused_used = 0;
}

// Window slots of the next node (`need` = output rows x feature ops); a full window leaves first.
template <class C, bool TABLE = false>
__device__ __forceinline__ void feat_reserve(WalkCtx &cx, int need) {
  if (cx.fused_used + need > cx.feat_window) feat_flush<C, TABLE>(cx, !cx.first_chunk);
  cx.fslot = cx.fused_used;
  cx.fused_used += need;
}

template <class C>
__device__ __forceinline__ void fused_all(WalkCtx &cx, const Rec &nd, const Ops2 &pre,
                                          const double (&c)[C::EP], const double (&x)[C::EP],
                                          const double (&s)[C::EP], bool seq_steps) {
  const IssArgs &a = *cx.a;
  const int ne = nd.emit_count(), n = a.n_ops;
  int64_t k = nd.w[7];
  Ops2 o = pre;  // ops 0-1 of the first row were requested at the start of the node
  FusedScratch<C::EP> sc;
  int slot = cx.fslot;   // window slot of (output row j, op i): fslot + j * n + i
  for (int j = 0;;) {
    for (int i = 0;;) {
      fused_op<C>(cx, o.w, slot + i, c, x, s, seq_steps, sc);
      if (i + 1 < n) fused_op<C>(cx, o.w + 8, slot + i + 1, c, x, s, seq_steps, sc);
      i += 2;
      if (i >= n) break;
      o = load_ops2(a, k, i);
    }
    if (++j >= ne) break;
    slot += n;
    k = j == 1 ? (int64_t)nd.w[8] : (int64_t)as_const(a.emit_rows)[nd.emit_begin() + j];
    o = load_ops2(a, k, 0);
  }
}

// Totally weighted sums in the fused epilogue: the row the sieves see is c[t] (x) w[t] (Reals /
// Bayesian: c * exp(-g alpha); Arctic: c - g alpha), so its increment at t is
// c[t] (x) w[t] - c[t-1] (x) w[t-1] - the second term formed here from the exclusive prefix
// (x[t] = c[t-1] bit for bit) and the weight row read one element to the left (staged in LDS;
// the element in front of the chunk from the table itself).  The result takes x's place in
// fused_op, whose plain difference then is the increment of the stored row, rounded like it.
template <class C>
__device__ __forceinline__ void previous_weighted(const WalkCtx &cx, int emit_mul,
                                                  const double (&x)[C::EP], double (&xs)[C::EP]) {
  constexpr int E = C::E, P = C::P;
  const IssArgs &a = *cx.a;
  const double *row = cx.rows + emit_mul * C::CHUNK;
#pragma unroll
  for (int h = 0; h < P; ++h)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int idx = cx.wave * C::SPAN + h * C::PIECE + cx.lane * E + e;
      double w = C::SEMI != 1 ? 1.0 : 0.0;
      if (idx > 0) {
        w = row[lds_pos<C>(idx - 1)];
      } else if (cx.t0 > 0) {
        const int src = as_const(a.row_src)[emit_mul];     // (an exp table: src < 0)
        w = (a.aux + (int64_t)(-src - 1) * a.aux_tab_stride + cx.series * a.aux_n_stride)[cx.t0 - 1];
      }
      if constexpr (C::SEMI != 1) {
        xs[h * E + e] = x[h * E + e] * w;
      } else {
#pragma clang fp contract(off)
        const double neg = -1.0 * w;                        // (as mul_row forms c + (-1) * row)
        xs[h * E + e] = x[h * E + e] + neg;
      }
    }
}

template <class C>
__device__ __forceinline__ void process_node(WalkCtx &cx, const Rec &nd, int slot,
                                             const double (&pin)[C::EP],
                                             double (&pout)[C::EP], const int32_t *more = nullptr,
                                             int n_more = 0) {
  constexpr int EP = C::EP;
  cx.slot = slot;
  double s[EP];
#pragma unroll
  for (int i = 0; i < EP; ++i) s[i] = pin[i];
  Ops2 pre;  // MODE 1: the first two feature ops of the first output row, requested early
  if constexpr (C::MODE == 1) {
    if (nd.emit_count() > 0) pre = load_ops2(*cx.a, nd.w[7], 0);
  }
  const int nf = nd.fac_count();
  bool letter_sum = false;
  if constexpr (C::SEMI == 1) {
    // Arctic argmax plans: C = sum of the letter's terms, then prefix + C
    // (fruits/iss/semiring.py:252-256); 0 + term is exact, so starting from zero changes
    // nothing but the association
    letter_sum = cx.a->letter_sum != 0;
    if (letter_sum) {
#pragma unroll
      for (int i = 0; i < EP; ++i) s[i] = 0.0;
    }
  }
  if (nd.flags() & F_SLOW) {
    slow_factors<C>(cx, nd.fac_begin(), nf, s, pin, letter_sum);
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
    block_scan<C>(cx, s, c, x, slot);
    // Reals: children start from the exclusive shift (strictly increasing indices);
    // Arctic: from the inclusive maximum (semiring.py:282-338 has no shift).  Taken
    // BEFORE the emitted values are rescaled in place below.  Written whether or not
    // the node has children (a leaf's frame is never read: its next sibling starts from
    // the frame below, and a second scan - need2 - overwrites it): a conditional
    // hand-over compiles to a select per register.
#pragma unroll
    for (int i = 0; i < EP; ++i) pout[i] = C::SEMI == 0 ? x[i] : c[i];
    if (nd.emit_count() > 0) {
      // total weighting: Reals emit c * exp(-g alpha_k), Arctic emit c - g alpha_k
      if (C::WEIGHTED && emit_mul >= 0)
        mul_row<C>(cx, C::SEMI != 1 ? emit_mul : fac_arctic(emit_mul, -1), c);
      if constexpr (C::MODE == 1 && C::TOTALINC) {
        if (emit_mul >= 0) {
          double xs[EP];
          previous_weighted<C>(cx, emit_mul, x, xs);
          fused_all<C>(cx, nd, pre, c, xs, s, false);
        } else {
          fused_all<C>(cx, nd, pre, c, x, s, true);
        }
      } else if constexpr (C::MODE == 1) {
        fused_all<C>(cx, nd, pre, c, x, s, !(C::WEIGHTED && emit_mul >= 0));
      } else
        emit_all<C>(cx, nd, c, more, n_more);
      STAMP(cx, 5);  // stores
    }
  }
  if constexpr (C::WEIGHTED) {
    if (need2) {
      double s2[EP], c[EP], x[EP];
#pragma unroll
      for (int i = 0; i < EP; ++i) s2[i] = s[i];
      mul_row<C>(cx, C::SEMI != 1 ? z_mul : fac_arctic(z_mul, 1), s2);
      block_scan<C>(cx, s2, c, x, slot + 1);
#pragma unroll
      for (int i = 0; i < EP; ++i) pout[i] = C::SEMI == 0 ? x[i] : c[i];
    }
  }
}

// carry slot of a node: LDS carries are indexed by the node's position inside its
// group, global ones by its plan-wide id
template <class C>
__device__ __forceinline__ int carry_slot_of(const WalkCtx &cx, const Rec &nd, int pc) {
  if constexpr (C::MULTI == 1) return kCarrySlots * (pc - cx.pc_begin);
  return kCarrySlots * nd.node_id();
}

// Walks the records of one group.  cx.cur always holds the record at cx.pc; the
// record after it is requested BEFORE the current node is processed, so its
// scalar-memory latency hides behind the node's vector work.  The sentinel at
// the end of every group (level 0xff) terminates all loops.
template <class C, int LV>
__device__ __forceinline__ void walk(WalkCtx &cx, Rec &cur, int &pc,
                                     const double (&pin)[C::EP]) {
  const IssArgs &a = *cx.a;
  if constexpr (C::MODE == 0 && C::MAXLV <= 4) {
    // two code sites per level (head of a chain / in-place continuation): no
    // register copies; affordable for the small materialising kernels
    while (cur.level() == LV) {
      const Rec nd = cur;
      const int slot = carry_slot_of<C>(cx, nd, pc);
      ++pc;
      cur = load_rec(a.recs, pc);
      double pout[C::EP];
      STAMP(cx, 0);  // interpreter: record decode / prefetch issue
      process_node<C>(cx, nd, slot, pin, pout);
      while (cur.level() == LV && (cur.flags() & F_CHAIN)) {
        const Rec nc = cur;
        const int slot2 = carry_slot_of<C>(cx, nc, pc);
        ++pc;
        cur = load_rec(a.recs, pc);
        process_node<C>(cx, nc, slot2, pout, pout);
      }
      if constexpr (LV + 1 < C::MAXLV) {
        if (cur.level() == LV + 1) walk<C, LV + 1>(cx, cur, pc, pout);
      }
    }
  } else {
    // one code site per level (code size matters for deep / fused kernels): an only
    // child continues in place in this frame (reads its parent's prefix from pout);
    // every other node of the level reads the frame below (pin)
    double pout[C::EP];
#pragma unroll
    for (int i = 0; i < C::EP; ++i) pout[i] = 0.0;
    while (cur.level() == LV) {
      const Rec nd = cur;
      const int slot = carry_slot_of<C>(cx, nd, pc);
      ++pc;
      cur = load_rec(a.recs, pc);
      double src[C::EP];
      if (nd.flags() & F_CHAIN) {
#pragma unroll
        for (int i = 0; i < C::EP; ++i) src[i] = pout[i];
      } else {
#pragma unroll
        for (int i = 0; i < C::EP; ++i) src[i] = pin[i];
      }
      STAMP(cx, 0);  // interpreter: record decode / prefetch issue
      process_node<C>(cx, nd, slot, src, pout);
      if constexpr (LV + 1 < C::MAXLV) {
        if (cur.level() == LV + 1) walk<C, LV + 1>(cx, cur, pc, pout);
      }
    }
  }
}

// ---------------------------------------------------------------- static programs
// The same walk with the program as a COMPILE-TIME constant: PG::w holds a SCHEDULE,
// 16 words per entry like a NodeRec, of one group (the whole plan):
//   node entries     the NodeRec of the node; w[14] / w[15] = the register frame its prefix
//                    is read from (-1: the semiring's one) / written to (-1: nobody reads it);
//   kSchedStage      w[1] = staged row whose registers go to LDS now (it is first read by
//                    the next node entry);
//   kSchedEmits      w[1] further output rows (w[2..]) of the node entry in front of it (a
//                    record holds two; SINGLE-mode plans with repeated words have more);
//   kSchedPrefetch   behind the last stage entry: the registers are free again - issue the
//                    loads of the workgroup's next unit into them;
//   sentinel         end of the schedule.
// Everything the interpreter decodes per node becomes an immediate, the walk is straight-line
// code, and - what the interpreter cannot do - the rows of a unit are loaded together but
// each is completed only in front of the first node that reads it: the nodes that need row 0
// alone run while the other rows are still in flight.  The host orders the schedule for that
// (plan.cpp, static_schedule): any order in which a parent precedes its children is a valid
// walk, since frames are registers named by constants, not a stack.

template <class PG, int PC>
__device__ __forceinline__ constexpr Rec static_rec() {
  Rec r{};
  for (int i = 0; i < 16; ++i) r.w[i] = PG::w[PC * 16 + i];
  return r;
}
template <class PG>
constexpr int static_kind(int pc) { return PG::w[pc * 16] & 0xff; }
template <class PG>
constexpr int static_more_emits(int pc) {   // rows in the kSchedEmits entries starting at pc
  int n = 0;
  while (static_kind<PG>(pc) == kSchedEmits) {
    n += PG::w[pc * 16 + 1];
    ++pc;
  }
  return n;
}

template <class C>
struct StaticRegs {
  static constexpr int U = C::CHUNK / 2 / kWalkThreads;
  vd2 v[kStaticMaxRows][U];                  // rows in flight (global -> registers -> LDS)
  double fr[kStaticMaxFrames][C::EP];        // open prefixes
};

// issues the loads of the rows in MASK of series n into rg.v
template <class C, class PG, int MASK>
__device__ __forceinline__ void static_load_rows(const WalkCtx &cx, StaticRegs<C> &rg, int64_t n) {
  const IssArgs &a = *cx.a;
#pragma unroll
  for (int r = 0; r < PG::rows; ++r) {
    if (!(MASK & (1 << r))) continue;
    const double *gp = a.X + (n * a.D + PG::row_src[r]) * a.T;
#pragma unroll
    for (int k = 0; k < StaticRegs<C>::U; ++k) {
      const int i = 2 * (k * kWalkThreads + cx.tid);
      rg.v[r][k] = vd2{0.0, 0.0};
      if (cx.full_chunk || i < a.T) rg.v[r][k] = load_input<PG::groups == 1>(gp + i);
    }
  }
}

template <class C, class PG, int PC>
__device__ __forceinline__ void walk_static(WalkCtx &cx, StaticRegs<C> &rg, double *rows_w) {
  constexpr int kind = static_kind<PG>(PC);
  if constexpr (kind == kRecSentinelLevel) {
    return;
  } else if constexpr (kind == kSchedPrefetch) {
    // (single-group programs only: the next unit is the same program on another series)
    cx.have_rows = cx.next_unit < (int)cx.a->N;
    if (cx.have_rows) static_load_rows<C, PG, PG::group_rows[0]>(cx, rg, cx.next_unit);
    walk_static<C, PG, PC + 1>(cx, rg, rows_w);
  } else if constexpr (kind == kSchedStage) {
    constexpr int r = PG::w[PC * 16 + 1];
#pragma unroll
    for (int k = 0; k < StaticRegs<C>::U; ++k) {
      const int i = 2 * (k * kWalkThreads + cx.tid);
      *reinterpret_cast<vd2 *>(rows_w + r * C::CHUNK + lds_pos<C>(i)) = rg.v[r][k];
    }
    lds_barrier();
    walk_static<C, PG, PC + 1>(cx, rg, rows_w);
  } else {
    constexpr Rec nd = static_rec<PG, PC>();
    constexpr int fin = nd.w[14], fout = nd.w[15];
    // further output rows of this node: kSchedEmits entries right behind it
    constexpr int n_more = static_more_emits<PG>(PC + 1);
    constexpr int skip = (n_more + kSchedEmitsPerEntry - 1) / kSchedEmitsPerEntry;
    int32_t more[n_more > 0 ? n_more : 1];
#pragma unroll
    for (int j = 0; j < n_more; ++j)
      more[j] = PG::w[(PC + 1 + j / kSchedEmitsPerEntry) * 16 + 2 + j % kSchedEmitsPerEntry];
    double ones[C::EP], dead[C::EP];
#pragma unroll
    for (int i = 0; i < C::EP; ++i) ones[i] = C::SEMI != 1 ? 1.0 : 0.0;
    if constexpr (fin < 0 && fout < 0)
      process_node<C>(cx, nd, 0, ones, dead, more, n_more);
    else if constexpr (fin < 0)
      process_node<C>(cx, nd, 0, ones, rg.fr[fout < 0 ? 0 : fout], more, n_more);
    else if constexpr (fout < 0)
      process_node<C>(cx, nd, 0, rg.fr[fin < 0 ? 0 : fin], dead, more, n_more);
    else
      process_node<C>(cx, nd, 0, rg.fr[fin], rg.fr[fout], more, n_more);
    walk_static<C, PG, PC + 1 + skip>(cx, rg, rows_w);
  }
}

template <class C, class PG, int GI>
__device__ __forceinline__ void static_run_group(WalkCtx &cx, StaticRegs<C> &rg, double *rows_w,
                                                 int64_t n, int g) {
  if constexpr (GI < PG::groups) {
    if (g == GI) {
      if (!cx.have_rows) static_load_rows<C, PG, PG::group_rows[GI]>(cx, rg, n);
      walk_static<C, PG, PG::group_begin[GI]>(cx, rg, rows_w);
    } else {
      static_run_group<C, PG, GI + 1>(cx, rg, rows_w, n, g);
    }
  }
}

// Materialising walk of a static program: one aligned time chunk (MULTI = 0, VEC), a unit is
// (series, group of the schedule), at most kStaticMaxRows staged rows, all of them rows of X.
template <class C, class PG>
__global__ __launch_bounds__(kWalkThreads) void iss_walk_static_kernel(const IssArgs a) {
  static_assert(C::MODE == 0 && C::MULTI == 0 && C::VEC && C::TEAM == 4 && !C::WEIGHTED,
                "static programs: materialising, single chunk, aligned, unweighted");
  static_assert(PG::rows <= kStaticMaxRows && PG::frames <= kStaticMaxFrames, "static program too wide");
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  WalkCtx cx;
  cx.a = &a;
  cx.rows = lds;
  cx.tot = lds + PG::rows * C::CHUNK;
  cx.tail = cx.tot + 2 * C::NW;
  cx.tid = tid;
  cx.lane = tid & 63;
  cx.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  cx.team = 0;
  cx.buf = 0;
  cx.tail_buf = 0;
  cx.carry = nullptr;
  cx.pc_begin = 0;
  cx.t0 = 0;
  cx.first_chunk = true;
  cx.full_chunk = C::CHUNK <= a.T;
  StaticRegs<C> rg;
  // The rows of a unit travel global -> registers -> LDS.  A unit loads its rows when it
  // starts (all of them at once, each completed in front of its first reader) - unless the
  // kSchedPrefetch entry of the workgroup's previous unit has already loaded them, many nodes
  // ahead of their use and in front of most of that unit's stores (vmcnt counts in order: a
  // load issued behind a store waits for it).
  cx.have_rows = false;
  bool first_unit = true;
  constexpr int G = PG::groups;
  const int u_end = (int)(a.N * G);
#ifdef FRUITS_HIP_TIMING_BUILD
  unsigned long long t_unit[4] = {0, 0, 0, 0};
  int n_unit = 0;
  const unsigned long long real_begin = __builtin_amdgcn_s_memrealtime();
#endif
  for (int u = blockIdx.x; u < u_end; u += gridDim.x) {
    int64_t n = u;
    int g = 0;
    if constexpr (G > 1) {
      if (a.xcd_map) {   // the groups of one series meet in one XCD's L2
        const int q = u >> 3, r = u & 7;
        n = (int64_t)(q / G) * 8 + r;
        g = q % G;
      } else {
        n = u / G;
        g = u - (int)n * G;
      }
    }
    cx.out_base = a.out + n * a.out_n_stride;
    cx.next_unit = u + (int)gridDim.x;
    if (!first_unit) lds_barrier();  // all reads of the previous unit's rows are done
    first_unit = false;
#ifdef FRUITS_HIP_TIMING_BUILD
    if (n_unit < 4) t_unit[n_unit++] = __builtin_amdgcn_s_memrealtime();
#endif
    static_run_group<C, PG, 0>(cx, rg, lds, n, g);
  }
#ifdef FRUITS_HIP_TIMING_BUILD
  if ((a.debug & 16) && a.dbg != nullptr && cx.lane == 0) {
    unsigned long long *o = a.dbg + ((int64_t)blockIdx.x * 4 + cx.wave) * 12;
    for (int i = 0; i < 4; ++i) o[i] = t_unit[i];
    unsigned hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    o[4] = hw_id;
    o[5] = xcc_id;
    o[8] = 1;
    o[10] = real_begin;
    o[11] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// WALK_SPILL_LV: fused kernels with at least this many register levels are built for three
// waves per SIMD (<= 168 VGPRs, no scratch) instead of four (128 VGPRs, the deepest frames
// spill 20-92 bytes per lane).  Measured (fruits_amd.build --variant, one process per arm,
// interleaved, r02h): four waves WITH the spills are faster - config 4 (6 levels) 23.9 vs
// 27.8 ms, config 5 (8 levels, 4 chunks) 42.0 vs 49.5 ms - so the default keeps four waves.
#ifndef WALK_SPILL_LV
#define WALK_SPILL_LV 99
#endif
// The fused kernels for 1024-element chunks sit just above 128 VGPRs; at least 4 waves
// per SIMD (<= 128 VGPRs) is worth the compiler's effort there.
#if defined(WALK_MODE) && WALK_MODE == 1 && defined(WALK_LV) && WALK_LV >= WALK_SPILL_LV
// deep tries: 8 VGPRs per register frame on top of the epilogue - three waves per SIMD
// (<= 168 VGPRs) hold them without scratch
#define WALK_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(3)))
#elif defined(WALK_MODE) && WALK_MODE == 1
#define WALK_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(4)))
#else
#define WALK_KERNEL_ATTR
#endif

template <class C>
__global__ __launch_bounds__(kWalkThreads) WALK_KERNEL_ATTR void iss_walk_kernel(const IssArgs a) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  WalkCtx cx;
  cx.a = &a;
  cx.rows = lds;
  cx.tot = lds + (int64_t)a.R * C::CHUNK;
  cx.tail = cx.tot + 2 * C::NW;
  cx.tid = tid;
  cx.lane = tid & 63;
  {
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    cx.wave = w % C::TEAM;
    cx.team = w / C::TEAM;
  }
  cx.buf = 0;
  cx.tail_buf = 0;
  double *rows_w = lds;
  bool first_unit = true;
#ifdef FRUITS_HIP_TIMING_BUILD
  if (a.debug & 4) return;
  for (int i = 0; i < 8; ++i) cx.seg[i] = 0;
  cx.last = stamp_now();
  const unsigned long long t_begin = cx.last;
  const unsigned long long real_begin = __builtin_amdgcn_s_memrealtime();  // 100 MHz, global
#endif
  // Persistent workgroups: the grid holds (at most) one resident round of workgroups and
  // each walks units b, b + grid, ...  A unit is (series n, group g of root sub-tries).
  // Workgroups are dealt round-robin over the 8 XCDs, so with the xcd_map numbering all
  // groups of one series meet in one XCD's L2 (speed only, never correctness).  In the
  // first round the resident workgroups write CONSECUTIVE series of every output plane.
  // (Round 2 tried contiguous spans of units per workgroup - even spans, one staging shared
  // by the groups of a series: 5-10 % slower at every batch size, docs/history.md 4.6.)
  int sink = 0;  // next-series prefetch (see below): one word per 128-byte line of the rows
  int pf_val = 0, pf_off = -1;
  static_assert(C::TEAM == 4 && C::MODE == 0,
                "the materialising interpreter (fused: walk_fused.h; short series: walk_packed.h)");
  {
    const int lines = (int)((a.T * 8 + 127) >> 7);
    if (a.prefetch_next && a.nchunks == 1 && a.D * a.T < (1 << 30) && tid < a.R * lines) {
      const int r = tid / lines, line = tid - r * lines;
      const int src = a.row_src[r];
      if (src >= 0) pf_off = src * (int)a.T + line * 16;
    }
  }
  const int u_end = (int)(a.N * a.G);  // (the host checks < 2^31)
  for (int u = blockIdx.x; u < u_end; u += gridDim.x) {
    int64_t n;
    int g0;
    if (a.xcd_map) {
      const int q = u >> 3, r = u & 7;
      n = (int64_t)(q / a.G) * 8 + r;
      g0 = q % a.G;
    } else {
      const int ni = u / a.G;
      n = ni;
      g0 = u - ni * a.G;
    }
    const int node_begin = as_const(a.group_begin)[g0];
    if constexpr (C::MULTI == 1)
      cx.carry = lds + (int64_t)a.R * C::CHUNK + 4 * C::NW;
    else
      cx.carry = a.carry ? a.carry + n * (kCarrySlots * (int64_t)a.total_nodes) : nullptr;
    cx.pc_begin = node_begin;  // LDS carry slots are indexed from the group's first record
    for (int64_t chunk = 0; chunk < a.nchunks; ++chunk) {
      const int64_t t0 = chunk * C::CHUNK;
      cx.t0 = t0;
      cx.first_chunk = chunk == 0;
      cx.full_chunk = t0 + C::CHUNK <= a.T;
      cx.out_base = a.out + n * a.out_n_stride + t0;
      if (!first_unit || chunk > 0) lds_barrier();  // all reads of the old rows are done
      // stage the referenced rows of this chunk: coalesced 16-byte units, the
      // loads of kStageRows rows in flight before the first LDS write
#ifdef FRUITS_HIP_TIMING_BUILD
      if (!(a.debug & 8))
#endif
      for (int r0 = 0; r0 < a.R; r0 += kStageRows) {
        constexpr int U = C::CHUNK / 2 / kWalkThreads;
        vd2 v[kStageRows][U];
#pragma unroll
        for (int rr = 0; rr < kStageRows; ++rr) {
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
              if (a.vec_ok) {
                if (cx.full_chunk || t < a.T) v[rr][k] = load_input<C::NT>(gp + t);
              } else {
                if (t < a.T) v[rr][k].x = gp[t];
                if (t + 1 < a.T) v[rr][k].y = gp[t + 1];
              }
            }
          }
        }
#pragma unroll
        for (int rr = 0; rr < kStageRows; ++rr) {
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
      {
        if (a.prefetch_next && a.nchunks == 1) {
          // Touch one word per 128-byte line of the rows of this workgroup's NEXT unit, so
          // that its staging - issued when the memory system is full of this kernel's
          // stores - finds them in the L2 / Infinity Cache.  The loaded value is only
          // consumed behind the next staging wait (no extra stall).
          sink += pf_val;
          pf_val = 0;
          // (lines touched a whole long unit ahead are evicted before they are used:
          // units of more than prefetch_next nodes do not prefetch)
          const int n_rec = as_const(a.group_begin)[g0 + 1] - node_begin;
          const int un = u + (int)gridDim.x;
          if (un < u_end && n_rec <= a.prefetch_next && pf_off >= 0) {
            const int64_t n_next =
                a.xcd_map ? (int64_t)((un >> 3) / a.G) * 8 + (un & 7) : (int64_t)(un / a.G);
            pf_val = *reinterpret_cast<const int *>(a.X + n_next * a.D * a.T + pf_off);
          }
        }
      }
      STAMP(cx, 6);  // staging
      double ones[C::EP];  // identity of the semiring's product: 1 (Reals, Bayesian), 0 (Arctic)
#pragma unroll
      for (int i = 0; i < C::EP; ++i) ones[i] = C::SEMI != 1 ? 1.0 : 0.0;
      int pc = node_begin;
      Rec cur = load_rec(a.recs, pc);
      walk<C, 0>(cx, cur, pc, ones);
    }
    first_unit = false;
  }
  if (sink + pf_val == 0x7fffffff) a.out[0] = (double)sink;  // keeps the prefetch loads alive
#ifdef FRUITS_HIP_TIMING_BUILD
  if ((a.debug & 16) && a.dbg != nullptr && cx.lane == 0) {
    unsigned long long *o = a.dbg + ((int64_t)blockIdx.x * 4 + cx.team * C::TEAM + cx.wave) * 12;
    for (int i = 0; i < 8; ++i) o[i] = cx.seg[i];
    o[8] = stamp_now() - t_begin;
    o[9] = t_begin;
    o[10] = real_begin;
    o[11] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

}  // namespace fr

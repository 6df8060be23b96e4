// The fused walk: INC -> ISS -> NPI / MPI / END as ONE launch (what FruitSlice.transform runs,
// fruits/fruit.py:538-550 - K x |sieves| Python calls over a materialised (K,N,T) tensor there).
//
// Same plan, records and arithmetic as the materialising walk (walk_device.h); what differs is
// the shape of the code.  These kernels write almost nothing, so they are bound by instruction
// issue, and the interpreter that serves the materialising walk well spent as many scalar as
// vector instructions per node here (and 100-500 SGPR spills).  This kernel is written for
// the fused case alone:
//   * one short-lived workgroup per (series, group) unit, no persistent loop;
//   * the first 8 dwords of a record hold everything a node of a multiply-only letter needs
//     (the rest is fetched on demand), so two records in flight cost 16 SGPRs, not 32;
//   * ONE node body for every register level (the interpreter's recursion over levels copies
//     it once per level): frames are indexed through two small uniform switches, one around
//     the first factor's multiply, one around the hand-over to the children;
//   * the cross-wave step of a scan reads a zero-padded WINDOW of the wave totals
//     (wave w reads slots w .. w+2 of {Z, Z, Z, t0, t1, t2, t3}), so the exclusive prefix of
//     the totals is two adds for every wave and no select; lane 63 publishes its wave's
//     total straight from the scan register (no readlane / broadcast);
//   * feature ops carry host-resolved flags (whole series, one-sided band), the epilogue
//     takes the short path for them;
//   * features accumulate in the LDS window of walk_device.h (feat_flush) and leave with
//     plain stores; a slot's column follows from the walk order (GroupedProgram::slot_rows).
// Results are bit-identical to the interpreter's (same association everywhere).
//
// What a kernel of this file knows at compile time is a template parameter:
//   iss_fused_kernel<C, TOTAL>                     the generic instances (walk_inst.hip): records
//                                                  and feature ops are decoded as they are read
//   iss_fused_kernel<C, TOTAL, JitOps>             a pipeline's own kernel (jit.cpp,
//                                                  fr_pipeline_prepare): the sieves as immediates
//   iss_fused_kernel<C, TOTAL, JitOps, JitPlan>    ... and the plan (fr_pipeline_compile_plan): a
//                                                  small one as straight-line code (fwalk_static),
//                                                  of a large one the node shapes (fwalk_shaped)
// and C::MODE == 2 is the same node loop with a store epilogue: the lean MATERIALISING walk of the
// plans that have no static program of walk_device.h's kind (emit_lean, two pieces per wave).
#pragma once
#include "walk_device.h"

namespace fr {

// feature-op flags (FeatOp::kind_inc), resolved by the host (capi.cpp, fr_pipeline_set_quantiles)
constexpr int32_t OPF_SERIES_CUTS = 1 << 16;
// bits 20-22: the op's SHAPE when it is one of the common ones (0: none of them)
constexpr int OPF_SHAPE_SHIFT = 20;
constexpr int OPF_SHAPE_END = 1;     // END at a fixed index (lo)
constexpr int OPF_SHAPE_BAND1 = 2;   // NPI over the whole series, qlo < v (qhi = +inf); | 1: of the first differences
constexpr int OPF_SHAPE_BAND2 = 4;   // NPI over the whole series, qlo < v <= qhi;       | 1: of the first differences

// Table reads of the node loop: base pointer + unsigned 32-bit BYTE offset (one scalar load with
// a register offset; a 64-bit index costs five scalar instructions of address arithmetic).
__device__ __forceinline__ cptr<int32_t> at_offset(const void *base, uint32_t off) {
  return (cptr<int32_t>)((cptr<char>)(base) + off);
}

// Single-lane LDS operations of a whole wave in uniform control flow (EXEC all ones before and
// after): the lane is selected by writing EXEC, not by a compare + saved mask + branch - two
// scalar instructions, no mask registers kept alive.  `off` is an LDS BYTE address.
__device__ __forceinline__ void lds_store_lane63(int off, double v) {
  asm volatile("s_lshl_b64 exec, 1, 63\n\tds_write_b64 %0, %1\n\ts_mov_b64 exec, -1"
               :: "v"(off), "v"(v) : "memory", "scc");   // (s_lshl_b64 writes SCC)
}
__device__ __forceinline__ void lds_store_lane(int lane, int off, double v) {
  asm volatile("s_lshl_b64 exec, 1, %2\n\tds_write_b64 %0, %1\n\ts_mov_b64 exec, -1"
               :: "v"(off), "v"(v), "s"(__builtin_amdgcn_readfirstlane(lane)) : "memory", "scc");
}
__device__ __forceinline__ void lds_store_lane0_i32(int off, int v) {
  asm volatile("s_mov_b64 exec, 1\n\tds_write_b32 %0, %1\n\ts_mov_b64 exec, -1"
               :: "v"(off), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_add_lane0(int off, double v) {
  asm volatile("s_mov_b64 exec, 1\n\tds_add_f64 %0, %1\n\ts_mov_b64 exec, -1"
               :: "v"(off), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_add2_lane63(int off0, double v0, int off1, double v1) {
  asm volatile("s_lshl_b64 exec, 1, 63\n\tds_add_f64 %0, %1\n\tds_add_f64 %2, %3\n\ts_mov_b64 exec, -1"
               :: "v"(off0), "v"(v0), "v"(off1), "v"(v1) : "memory", "scc");
}
__device__ __forceinline__ int lds_offset(const void __attribute__((address_space(3))) *p) {
  return (int)(unsigned)(__UINTPTR_TYPE__)p;
}

// The kernel arguments as seen from code that rarely runs (flushes, reciprocal factors, further
// output rows, per-series cuts): read through a pointer the optimiser cannot look through, so
// that they are loaded where they are used instead of being hoisted out of the node loop into
// scalar registers - there are only 102 of those, and the hot path needs them.
// (The kernel's only argument is the IssArgs struct: it starts the kernel-argument segment.)
__device__ __forceinline__ cptr<IssArgs> cold_args() {
  cptr<IssArgs> ap = (cptr<IssArgs>)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ap));
  return ap;
}

__device__ __forceinline__ const void *uniform_ptr(const void *p) {
  const uint64_t u = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return reinterpret_cast<const void *>(((uint64_t)hi << 32) | lo);
}

// One dword of a table line, loaded a node ahead of the real access: the line is then in the
// scalar cache and the real load (issued right where its registers are needed - prefetching INTO
// registers keeps 16 SGPRs alive across the node and ends in spills) is a cache hit.
__device__ __forceinline__ int touch(const void *base, uint32_t off) {
  return at_offset(base, off)[0];
}

// the first half of a NodeRec (w[0..7]): level | flags, factor count | weights, four inline
// factors, emit count, first output row
struct Rec8 {
  int32_t w[8];
  __device__ __forceinline__ int level() const { return w[0] & 0xff; }
  __device__ __forceinline__ int flags() const { return w[0] >> 8; }
  __device__ __forceinline__ int fac_count() const { return w[1] & 0xffff; }
  __device__ __forceinline__ int z_mul() const { return ((w[1] >> 16) & 0xff) - 1; }
  __device__ __forceinline__ int emit_mul() const { return ((w[1] >> 24) & 0xff) - 1; }
  __device__ __forceinline__ int emit_count() const { return w[6]; }
};

__device__ __forceinline__ Rec8 load_rec8(const NodeRec *recs, uint32_t rec_off) {
  cptr<int32_t> q = at_offset(recs, rec_off);
  Rec8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.w[i] = q[i];
  return r;
}

// a field of the second half of a record (second output row, factor / emit table offsets)
__device__ __forceinline__ int rec_field(const NodeRec *recs, uint32_t rec_off, int i) {
  return at_offset(recs, rec_off)[i];
}

// s = src (x) factor: the first factor of a letter reads the prefix it continues
template <class C>
__device__ __forceinline__ void mul_row_from(const WalkCtx &cx, int code, const double (&src)[C::EP],
                                             double (&s)[C::EP]) {
  double v[C::EP];
  // (multiply-only codes of Reals / Bayesian records are bare row numbers: nothing to mask)
  read_row<C>(cx, C::SEMI != 1 ? code : (code & FAC_ROW_MASK), v);
  if constexpr (C::SEMI != 1) {
#pragma unroll
    for (int i = 0; i < C::EP; ++i) s[i] = src[i] * v[i];
  } else {
#pragma clang fp contract(off)  // the product must round before the add (no FMA)
    const double el = (double)(int)(int8_t)((code >> 8) & 0xff);
#pragma unroll
    for (int i = 0; i < C::EP; ++i) {
      const double prod = el * v[i];
      s[i] = src[i] + prod;
    }
  }
}

// s (x)= factor for the codes the fused walk meets outside the factor table: bare row numbers
// (Reals / Bayesian), row | multiplier << 8 (Arctic)
template <class C>
__device__ __forceinline__ void mul_rowp(const WalkCtx &cx, int code, double (&s)[C::EP]) {
  if constexpr (C::SEMI == 1) {
    mul_row<C>(cx, code, s);
  } else {
    double v[C::EP];
    read_row<C>(cx, code, v);
#pragma unroll
    for (int i = 0; i < C::EP; ++i) s[i] = s[i] * v[i];
  }
}

// The same scan for P pieces per wave (the materialising walk, MODE 2: lane l holds E = 2
// consecutive elements of each of its wave's two pieces, so that every 16-byte store instruction
// of a wave covers 1 KiB without holes): one DPP chain per piece, interleaved; the sums and their
// order are block_scan's.
template <class C>
__device__ __forceinline__ void fscan_pieces(WalkCtx &cx, const double (&s)[C::EP], double (&c)[C::EP],
                                             double (&x)[C::EP], int carry_slot) {
  constexpr int E = C::E, P = C::P;
  double l[C::EP], incl[P], excl[P];
#pragma unroll
  for (int h = 0; h < P; ++h) {
    l[h * E] = s[h * E];
#pragma unroll
    for (int e = 1; e < E; ++e) l[h * E + e] = semi_add<C::SEMI>(l[h * E + e - 1], s[h * E + e]);
    incl[h] = l[h * E + E - 1];
  }
  wave_inclusive_scan_multi<P, C::SEMI>(incl);
#pragma unroll
  for (int h = 0; h < P; ++h) excl[h] = wave_shift_right1<C::SEMI>(incl[h]);
  // lane 63: the totals of the pieces in order = the wave's total
  double total = incl[0];
#pragma unroll
  for (int h = 1; h < P; ++h) total = semi_add<C::SEMI>(total, incl[h]);
  double carry_in = semi_zero<C::SEMI>();
  if constexpr (C::MULTI != 0) {
    if (!cx.first_chunk) carry_in = cx.carry[carry_slot];
  }
  lds_f64 *tw = (lds_f64 *)(cx.tot + cx.buf * 8);   // {Z, Z, Z, t0, t1, t2, t3, -}
  lds_store_lane63(lds_offset(tw + 3 + cx.wave), total);
  lds_barrier();
  const double a0 = tw[cx.wave], a1 = tw[cx.wave + 1], a2 = tw[cx.wave + 2];
  double base = semi_add<C::SEMI>(semi_add<C::SEMI>(a0, a1), a2);
  cx.buf ^= 1;
  if constexpr (C::MULTI != 0) {
    if (cx.wave == 3)
      lds_store_lane63(lds_offset((lds_f64 *)cx.carry + carry_slot),
                       semi_add<C::SEMI>(carry_in, semi_add<C::SEMI>(base, total)));
    base = semi_add<C::SEMI>(base, carry_in);
  }
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
    if (h + 1 < P) base = semi_add<C::SEMI>(base, wave_last_lane(incl[h]));
  }
}

// Inclusive scan of one chunk row over the workgroup (P = 1: lane l holds E consecutive
// elements, wave w the span [w * 64 E, (w + 1) * 64 E)).  Same sums in the same order as
// block_scan; the cross-wave step reads the totals window (see the file comment).
// WINDOW = false: the identity slots hold another semiring's zero (the plain sums of a max-plus
// plan's cumulated rows, inc < 0) - select instead.
// `carry_wr`: the slot the advanced chunk carry goes to (default: the one it was read from; the
// cumulations of the epilogue keep two buffers, see fop).
template <class C, bool WINDOW = true>
__device__ __forceinline__ void fscan(WalkCtx &cx, const double (&s)[C::EP], double (&c)[C::EP],
                                      double (&x)[C::EP], int carry_slot, int carry_wr = -1) {
  static_assert(C::NW == 4, "four waves");
  if constexpr (C::P != 1) {
    fscan_pieces<C>(cx, s, c, x, carry_slot);
    return;
  }
  constexpr int E = C::E;
  double l[E];
  l[0] = s[0];
#pragma unroll
  for (int e = 1; e < E; ++e) l[e] = semi_add<C::SEMI>(l[e - 1], s[e]);
  const double incl = wave_inclusive_scan<C::SEMI>(l[E - 1]);
  const double excl = wave_shift_right1<C::SEMI>(incl);
  double carry_in = semi_zero<C::SEMI>();
  if constexpr (C::MULTI != 0) {
    if (!cx.first_chunk) carry_in = cx.carry[carry_slot];
  }
  lds_f64 *tw = (lds_f64 *)(cx.tot + cx.buf * 8);   // {Z, Z, Z, t0, t1, t2, t3, -}
  lds_store_lane63(lds_offset(tw + 3 + cx.wave), incl);
  lds_barrier();
  double base;
  if constexpr (WINDOW) {
    const double a0 = tw[cx.wave], a1 = tw[cx.wave + 1], a2 = tw[cx.wave + 2];
    base = semi_add<C::SEMI>(semi_add<C::SEMI>(a0, a1), a2);
  } else {
    const double t0 = tw[3], t1 = tw[4], t2 = tw[5];
    const double p2 = semi_add<C::SEMI>(t0, t1), p3 = semi_add<C::SEMI>(p2, t2);
    base = cx.wave == 0 ? semi_zero<C::SEMI>() : (cx.wave == 1 ? t0 : (cx.wave == 2 ? p2 : p3));
  }
  cx.buf ^= 1;
  if constexpr (C::MULTI != 0) {
    // the carry of earlier chunks: every wave read it before the barrier, the last lane of the
    // chunk advances it
    if (cx.wave == 3)
      lds_store_lane63(lds_offset((lds_f64 *)cx.carry + (carry_wr < 0 ? carry_slot : carry_wr)),
                       semi_add<C::SEMI>(carry_in, semi_add<C::SEMI>(base, incl)));
    base = semi_add<C::SEMI>(base, carry_in);
  }
  const double off = semi_add<C::SEMI>(base, excl);
  x[0] = off;
#pragma unroll
  for (int e = 0; e + 1 < E; ++e) {
    c[e] = semi_add<C::SEMI>(off, l[e]);
    x[e + 1] = c[e];
  }
  c[E - 1] = semi_add<C::SEMI>(base, incl);
}

// c[eu] for a uniform eu: a chain of uniform branches (the empty asm keeps the arms apart: as
// selects, or as an index, the compiler would send the whole array through scratch memory)
template <int E, int K>
__device__ __forceinline__ double pick_uniform(int eu, const double (&c)[E]) {
  if constexpr (K + 1 >= E) {
    return c[K];
  } else {
    double v;
    if (eu == K) {
      v = c[K];
      asm volatile("" : "+v"(v));
    } else {
      v = pick_uniform<E, K + 1>(eu, c);
    }
    return v;
  }
}

// END at a position the host resolved (no per-series cut): the value at series index `pick`
template <class C>
__device__ __forceinline__ void end_pick(WalkCtx &cx, int pick, int slot, const double (&c)[C::EP]) {
  constexpr int E = C::E;
  const unsigned rel = (unsigned)(pick - (int)cx.t0);
  if (rel >= (unsigned)C::CHUNK) return;           // another time chunk's element
  if ((int)(rel / C::SPAN) != cx.wave) return;     // another wave's
  // the position is uniform: uniform selects of the register, ONE lane stores
  const unsigned q = rel % C::SPAN;
  const double v = pick_uniform<E, 0>((int)(q % E), c);
  lds_store_lane((int)(q / E), lds_offset(cx.fl_val + slot), v);
}

template <class C>
__device__ __forceinline__ void fop(WalkCtx &cx, const int32_t *w, int slot, const double (&c)[C::EP],
                                    const double (&x)[C::EP], const double (&s)[C::EP],
                                    bool seq_steps, FusedScratch<C::EP> &sc) {
  constexpr int E = C::E;
  // The shapes the host flags (OPF_SHAPE_*: END at a fixed index; a counting band over the whole
  // series, of the values or of their first differences, with or without an upper threshold)
  // are short straight-line paths; everything else takes the general one below.
  const int shape = (w[0] >> OPF_SHAPE_SHIFT) & 7;
  if (shape == OPF_SHAPE_END) {
    end_pick<C>(cx, w[2], slot, c);
    return;
  }
  if (shape != 0 && cx.full_chunk) {
    const double qlo = bits_to_double(w[4], w[5]);
    double d[E];
    if (shape & 1) {   // first differences
#pragma unroll
      for (int i = 0; i < E; ++i) {
        double step = c[i] - x[i];
        if (C::SEMI == 0 && seq_steps) step = (x[i] + s[i]) - x[i];
        d[i] = step;
      }
      // element 0 of the series: increments are zero-padded there (fruits/cache.py:8-13)
      d[0] = (cx.first_chunk && cx.wave == 0 && cx.lane == 0) ? 0.0 : d[0];
    } else {
#pragma unroll
      for (int i = 0; i < E; ++i) d[i] = c[i];
    }
    int cnt = 0;
    if (shape >= OPF_SHAPE_BAND2) {
      const double qhi = bits_to_double(w[6], w[7]);
#pragma unroll
      for (int e = 0; e < E; ++e) cnt += __popcll(__ballot(qlo < d[e]) & __ballot(d[e] <= qhi));
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) cnt += __popcll(__ballot(qlo < d[e]));
    }
    lds_add_lane0(lds_offset(cx.fl_val + slot), (double)cnt);
    return;
  }
  const int kind = w[0] & 0xff, inc = (int)(int8_t)((w[0] >> 8) & 0xff);   // inc: signed
  const bool series_cuts = (w[0] & OPF_SERIES_CUTS) != 0;
  if (kind == FR_SIEVE_END_K) {
    int pick = w[2];                    // index of the value to pick
    if (series_cuts) {                  // X[:, cut - 1], index -1 wrapping like numpy
      cptr<IssArgs> ca = cold_args();
      pick = as_const(ca->series_cuts + cx.series * ca->cut_slots)[w[2]] - 1;
      if (pick < 0) pick += (int)ca->T;
    }
    end_pick<C>(cx, pick, slot, c);
    return;
  }
  int lo = w[2], hi = w[3];
  if (series_cuts) {
    cptr<IssArgs> ca = cold_args();
    cptr<int32_t> cut_row = as_const(ca->series_cuts + cx.series * ca->cut_slots);
    lo = cut_row[w[2]];
    hi = cut_row[w[3]];
  }
  const double qlo = bits_to_double(w[4], w[5]), qhi = bits_to_double(w[6], w[7]);
  // element 0 of the series: increments are zero-padded there (fruits/cache.py:8-13)
  const bool at0 = cx.first_chunk && cx.wave == 0 && cx.lane == 0;
  double d[E];
  if (inc <= 0) {
#pragma unroll
    for (int i = 0; i < E; ++i) d[i] = c[i];
    if constexpr (C::MULTI == 0 || C::HIGHORD) {
      // inc < 0: the row cumulated -inc times (np.cumsum, fruits/sieving/increment.py:68-70) -
      // a plain sum whatever the semiring of the plan.  On series of several chunks
      // (WalkCfg::HIGHORD) the j-th cumulation carries its running sum in a slot pair of its
      // own behind the differencing orders' (kCumCarryBase + 2 (j - 1)): this chunk reads buffer
      // `parity` and writes the other, so several sieves of a node read the same old value
      using CR = WalkCfg<C::E, C::P, C::MAXLV, C::MULTI, C::VEC, C::WEIGHTED, C::TEAM, C::MODE, 0>;
      for (int k = inc; k < 0; ++k) {
        double cs[E], xs[E];
        const int at = cx.slot + kCumCarryBase + 2 * (k - inc);
        fscan<CR, false>(cx, d, cs, xs, at + cx.parity, at + (cx.parity ^ 1));
#pragma unroll
        for (int i = 0; i < E; ++i) d[i] = cs[i];
      }
    }
  } else {
    // Reals: fl(x + s) - x, the step a SEQUENTIAL cumsum takes from the same prefix (see
    // fused_op in walk_device.h)
#pragma unroll
    for (int i = 0; i < E; ++i) {
      double step = c[i] - x[i];
      if (C::SEMI == 0 && seq_steps) step = (x[i] + s[i]) - x[i];
      d[i] = step;
    }
    d[0] = at0 ? 0.0 : d[0];
    if (inc >= 2) {
      if (!sc.have_dp) {
        prev_first_differences<C>(cx, d, sc.dp);
        sc.have_dp = true;
      }
#pragma unroll
      for (int e = 0; e < E; ++e) d[e] = d[e] - sc.dp[e];
      d[0] = at0 ? 0.0 : d[0];
      // third to eighth differences: one more neighbour exchange per order; on series of several
      // chunks (WalkCfg::HIGHORD) every order carries its last value of the chunk in a slot pair
      // of its own (IssArgs::carry_per_node = 3 + 2 * (highest order - 2))
      if constexpr (C::MULTI == 0 || C::HIGHORD) {
        for (int k = 3; k <= inc; ++k) {
          double dq[E];
          prev_first_differences<C>(cx, d, dq, 3 + 2 * (k - 3), C::MULTI != 0);
#pragma unroll
          for (int e = 0; e < E; ++e) d[e] = d[e] - dq[e];
          d[0] = at0 ? 0.0 : d[0];
        }
      }
    }
  }
  const bool mpi = kind == FR_SIEVE_MPI_K;
  const int t_first = (int)cx.t0 + cx.wave * C::SPAN + cx.lane * E;
  int cnt = 0;
  double sum = 0.0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int t = t_first + e;
    const double v = d[e];
    const bool in_t = t >= lo && t < hi;
    if (mpi) {
      const bool in = in_t && qlo < v && v <= qhi;
      cnt += __popcll(__ballot(in));
      sum += in ? v : 0.0;
    } else {
      // (the AND of the compare masks: as one predicate it goes through a vector register)
      cnt += __popcll(__ballot(in_t) & __ballot(qlo < v) & __ballot(v <= qhi));
    }
  }
  // one LDS add per wave (ds_add_f64, nothing returned)
  if (mpi) {
    sum = wave_inclusive_scan<0>(sum);  // lane 63 holds the wave total
    lds_add2_lane63(lds_offset(cx.fl_val + slot), sum, lds_offset(cx.fl_cnt + slot), (double)cnt);
  } else {
    lds_add_lane0(lds_offset(cx.fl_val + slot), (double)cnt);
  }
}

// one feature op (32 bytes = one s_load_dwordx8)
struct Op1 {
  int32_t w[8];
};
// what the node loop reads of the kernel arguments, as scalars of their own
struct Hot {
  const NodeRec *recs;
  const FeatOp *ops;
  int n_ops;
  uint32_t op_row_bytes;   // bytes of one output row's ops (the host checks the table < 4 GiB)
  int cps;                 // carry slots per node (kCarrySlots; more with differencing orders >= 3)
  // a body of a plan in pieces (fwalk_pieces): its output rows and carry slots count from these
  // (0 everywhere else)
  uint32_t op_base;        // byte offset of the body's first output row in the op table
  int slot_base;           // carry slot of the body's first node
  // MODE 2 (the tensor is written): row 0 of this series' chunk, bytes between two output rows
  // (0: beyond 32 bits - the general product), whole aligned chunk (no per-lane checks)
  char *out;
  uint32_t k_stride;
  bool fast_store;
};
__device__ __forceinline__ Op1 load_op1(const Hot &a, uint32_t op_off) {
  cptr<int32_t> q = at_offset(a.ops, op_off);
  Op1 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o.w[j] = q[j];
  return o;
}

// What a kernel knows of the sieves at compile time.  DynOps: nothing - every op is decoded from
// its 32-byte record (kind, differencing order, shape, cuts).  A run-time compiled kernel (jit.cpp,
// fr_pipeline_prepare) is instantiated over a struct like
//   struct JitOps { static constexpr bool is_static = true; static constexpr int n = 2;
//                   static constexpr int32_t w0[n] = {...}, lo[n] = {...}, hi[n] = {...}; };
// - the ops of EVERY output row of a pipeline are the same sieves, only the thresholds and the
// column differ - so kind_inc / lo / hi are immediates: the op loop is unrolled and the decode of
// fop folds away (an END at a fixed index of a one-chunk series becomes one predicated LDS store).
struct DynOps {
  static constexpr bool is_static = false, window_fits = false, full_chunks = false;
  static constexpr int n = 0;
};

// Window slots of the next node.  OPS::window_fits (a kernel of a piece type whose largest unit
// fits the feature window: the host knows when it compiles the kernel): no test, and no copy of
// the flush behind every node of straight-line code.
template <class C, class OPS>
__device__ __forceinline__ void freserve(WalkCtx &cx, int need) {
  if constexpr (OPS::window_fits) {
    cx.fslot = cx.fused_used;
    cx.fused_used += need;
  } else {
    feat_reserve<C, true>(cx, need);
  }
}

template <class C, class OPS, bool SEQ, int I>
__device__ __forceinline__ void fops_static(WalkCtx &cx, const Hot &a, uint32_t op_off, int slot,
                                            const double (&c)[C::EP], const double (&x)[C::EP],
                                            const double (&s)[C::EP], FusedScratch<C::EP> &sc) {
  if constexpr (I < OPS::n) {
    const Op1 o = load_op1(a, op_off + 32u * (uint32_t)I);   // (column and thresholds: the row's own)
    const int32_t w[8] = {OPS::w0[I], o.w[1], OPS::lo[I], OPS::hi[I], o.w[4], o.w[5], o.w[6], o.w[7]};
    fop<C>(cx, w, slot + I, c, x, s, SEQ, sc);
    fops_static<C, OPS, SEQ, I + 1>(cx, a, op_off, slot, c, x, s, sc);
  }
}

// The feature ops of every output row of a node: ONE code site for an op; an op is loaded where
// it is evaluated (its line was touched at the start of the node: a scalar-cache hit).
template <class C, bool SEQ, class OPS>
__device__ __forceinline__ void fops_all(WalkCtx &cx, const Hot &a, int ne, uint32_t op_off,
                                         uint32_t rec_off, const double (&c)[C::EP],
                                         const double (&x)[C::EP], const double (&s)[C::EP]) {
  const int n = OPS::is_static ? OPS::n : a.n_ops;
  FusedScratch<C::EP> sc;
  int slot = cx.fslot;   // window slot of (output row j, op i): fslot + j * n + i
  for (int j = 0;;) {
    if constexpr (OPS::is_static) {
      fops_static<C, OPS, SEQ, 0>(cx, a, op_off, slot, c, x, s, sc);
    } else {
      for (int i = 0; i < n; ++i) {
        const Op1 o = load_op1(a, op_off + 32u * (uint32_t)i);
        fop<C>(cx, o.w, slot + i, c, x, s, SEQ, sc);
      }
    }
    if (++j >= ne) break;
    slot += n;
    const int k = j == 1 ? rec_field(a.recs, rec_off, 8)
                         : as_const(cold_args()->emit_rows)[rec_field(a.recs, rec_off, 13) + j];
    op_off = a.op_base + (uint32_t)k * a.op_row_bytes;
  }
}

// MODE 2: the node's values into every output row it names (two inline in the record, further
// ones - SINGLE-mode plans with repeated words - from the table).  A lane holds E consecutive
// elements: 16-byte stores, base = the row's uniform address, offset = one 32-bit register.
template <class C>
__device__ __forceinline__ void emit_lean(const WalkCtx &cx, const Hot &a, int ne, int k,
                                          uint32_t rec_off, const double (&c)[C::EP]) {
  constexpr int E = C::E, P = C::P;
  const uint32_t lane_bytes = (uint32_t)(cx.wave * C::SPAN + cx.lane * E) * 8u;
  for (int j = 0;;) {
    char *row = a.k_stride != 0
                    ? a.out + (uint64_t)(uint32_t)k * (uint64_t)a.k_stride
                    : a.out + (int64_t)k * cold_args()->out_k_stride * 8;
    if (a.fast_store) {
#pragma unroll
      for (int h = 0; h < P; ++h)
#pragma unroll
        for (int e = 0; e < E; e += 2)
          *reinterpret_cast<vd2 *>(row + lane_bytes + 8u * (h * C::PIECE + e)) = vd2{c[h * E + e], c[h * E + e + 1]};
    } else {
      const int64_t T = cold_args()->T;
#pragma unroll
      for (int h = 0; h < P; ++h)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const uint32_t at = lane_bytes + 8u * (h * C::PIECE + e);
          if (cx.t0 + (at >> 3) < T) *reinterpret_cast<double *>(row + at) = c[h * E + e];
        }
    }
    if (++j >= ne) break;
    k = j == 1 ? rec_field(a.recs, rec_off, 8)
               : as_const(cold_args()->emit_rows)[rec_field(a.recs, rec_off, 13) + j];
  }
}

// Register frames: f[k] holds the prefix the children of the open node of level k continue
// from.  They are only ever indexed by compile-time constants: a node's level selects a case
// through a chain of single-bit tests of 1 << level, deepest level first (that is where most
// nodes are; as a chain of equality tests the compiler builds a balanced switch with flag
// variables out of it).  Case K < 0 is the semiring's one (a first letter).
template <class C, int K>
__device__ __forceinline__ void frame_mul(const WalkCtx &cx, unsigned rd_bit, int code,
                                          const double (&f)[C::MAXLV][C::EP], double (&s)[C::EP]) {
  if constexpr (K < 0) {
    double ones[C::EP];  // identity of the semiring's product: 1 (Reals, Bayesian), 0 (Arctic)
#pragma unroll
    for (int i = 0; i < C::EP; ++i) ones[i] = C::SEMI != 1 ? 1.0 : 0.0;
    mul_row_from<C>(cx, code, ones, s);
  } else {
    if (rd_bit & (2u << K)) {
      mul_row_from<C>(cx, code, f[K], s);
      asm volatile("" ::: "memory");  // (keeps the cases apart: merged, they cost selects)
    } else {
      frame_mul<C, K - 1>(cx, rd_bit, code, f, s);
    }
  }
}
template <class C, int K>
__device__ __forceinline__ void frame_get(unsigned rd_bit, const double (&f)[C::MAXLV][C::EP],
                                          double (&s)[C::EP]) {
  if constexpr (K < 0) {
#pragma unroll
    for (int i = 0; i < C::EP; ++i) s[i] = C::SEMI != 1 ? 1.0 : 0.0;
  } else {
    if (rd_bit & (2u << K)) {
#pragma unroll
      for (int i = 0; i < C::EP; ++i) s[i] = f[K][i];
      asm volatile("" ::: "memory");
    } else {
      frame_get<C, K - 1>(rd_bit, f, s);
    }
  }
}
template <class C, int K>
__device__ __forceinline__ void frame_put(unsigned lv_bit, double (&f)[C::MAXLV][C::EP],
                                          const double (&x)[C::EP]) {
  if constexpr (K >= 0) {
    if (lv_bit & (1u << K)) {
#pragma unroll
      for (int i = 0; i < C::EP; ++i) f[K][i] = x[i];
      asm volatile("" ::: "memory");
    } else {
      frame_put<C, K - 1>(lv_bit, f, x);
    }
  }
}

// One node of the record loop: the letters into the prefix it continues (the frame of the level
// below; its own level's for an only child, F_CHAIN), the scan, the feature ops (MODE 2: the
// stores), the hand-over to the children.  `nd` comes in as the node's record and leaves as the
// NEXT one (requested behind the first scan: a scalar-cache hit, in flight over the second).
template <class C, bool TOTAL, class OPS>
__device__ __forceinline__ void fnode(WalkCtx &cx, const Hot &a, double (&f)[C::MAXLV][C::EP],
                                      Rec8 &nd, uint32_t me, int slot, int &sink) {
  constexpr int EP = C::EP;
  const uint32_t rec_off = me + 64u;
  const int ne = nd.emit_count();
  const uint32_t op_off = (uint32_t)nd.w[7] * a.op_row_bytes;
  // the lines of the next record and of this node's ops, on their way to the scalar cache
  // (a node without output rows names row 0: a harmless touch)
  const int t_rec = touch(a.recs, rec_off);
  int t_ops = 0;
  if constexpr (C::MODE == 1) {
    t_ops = touch(a.ops, op_off);
    freserve<C, OPS>(cx, ne * (OPS::is_static ? OPS::n : a.n_ops));
  }
  cx.slot = slot;
  const int nf = nd.fac_count(), flags = nd.flags(), lv = nd.level();
  const unsigned lv_bit = 1u << lv;
  const unsigned rd_bit = (flags & F_CHAIN) ? (lv_bit << 1) : lv_bit;   // 2 << (level read)
  double s[EP];
  if (flags & F_SLOW) {
    // a reciprocal factor, more than four or none: the factor table, one factor at a time
    frame_get<C, C::MAXLV - 1>(rd_bit, f, s);
    slow_factors<C>(cx, rec_field(a.recs, me, 12), nf, s, s, false);
  } else {
    frame_mul<C, C::MAXLV - 1>(cx, rd_bit, nd.w[2], f, s);
    if (nf > 1) {
      mul_rowp<C>(cx, nd.w[3], s);
      if (nf > 2) mul_rowp<C>(cx, nd.w[4], s);
      if (nf > 3) mul_rowp<C>(cx, nd.w[5], s);
    }
  }
  const int w1 = nd.w[1];
  if (flags & F_NEED1) {
    double c[EP], x[EP];
    fscan<C>(cx, s, c, x, slot);
    // Reals: children start from the exclusive shift; Arctic / Bayesian: from the inclusive
    // maximum (taken before the emitted values are rescaled)
    if ((flags & (F_CHILDREN | F_NEED2)) == F_CHILDREN) {
      if constexpr (C::SEMI == 0) frame_put<C, C::MAXLV - 1>(lv_bit, f, x);
      else frame_put<C, C::MAXLV - 1>(lv_bit, f, c);
    }
    if (flags & F_EMIT) {
      if constexpr (C::WEIGHTED && TOTAL) {
        // total weighting: the sieves see c * exp(-g alpha_k) (Arctic: c - g alpha_k)
        const int emit_mul = ((w1 >> 24) & 0xff) - 1;
        mul_rowp<C>(cx, C::SEMI != 1 ? emit_mul : fac_arctic(emit_mul, -1), c);
        if constexpr (C::MODE == 2) {
          emit_lean<C>(cx, a, ne, nd.w[7], me, c);
        } else if constexpr (C::TOTALINC) {
          double xs[EP];
          previous_weighted<C>(cx, emit_mul, x, xs);
          fops_all<C, false, OPS>(cx, a, ne, op_off, me, c, xs, s);
        } else {
          fops_all<C, false, OPS>(cx, a, ne, op_off, me, c, x, s);
        }
      } else if constexpr (C::MODE == 2) {
        emit_lean<C>(cx, a, ne, nd.w[7], me, c);
      } else {
        fops_all<C, true, OPS>(cx, a, ne, op_off, me, c, x, s);
      }
    }
  }
  sink += t_rec + t_ops;
  nd = load_rec8(a.recs, rec_off);   // (a scalar-cache hit; in flight over the second scan)
  if constexpr (C::WEIGHTED && !TOTAL) {
    if (flags & F_NEED2) {
      // non-total weighting: the children continue from the scan of s * exp(+g alpha_k)
      const int z_mul = ((w1 >> 16) & 0xff) - 1;
      double c2[EP], x2[EP];
      mul_rowp<C>(cx, C::SEMI != 1 ? z_mul : fac_arctic(z_mul, 1), s);
      fscan<C>(cx, s, c2, x2, slot + 1);
      if constexpr (C::SEMI == 0) frame_put<C, C::MAXLV - 1>(lv_bit, f, x2);
      else frame_put<C, C::MAXLV - 1>(lv_bit, f, c2);
    }
  }
}

// what the node loop reads of the kernel arguments (see Hot)
template <class C>
__device__ __forceinline__ Hot hot_args(const WalkCtx &cx) {
  Hot a;
  const IssArgs &ka = *cx.a;
  // (through readfirstlane: scalars of their own, not pieces of one wide kernel-argument load
  // that is spilled and restored as a whole)
  a.recs = reinterpret_cast<const NodeRec *>(uniform_ptr(ka.recs));
  a.ops = reinterpret_cast<const FeatOp *>(uniform_ptr(ka.ops));
  a.n_ops = __builtin_amdgcn_readfirstlane(ka.n_ops);
  a.op_row_bytes = __builtin_amdgcn_readfirstlane(ka.n_ops_padded * 32);
  a.cps = C::MULTI != 0 ? __builtin_amdgcn_readfirstlane(ka.carry_per_node) : kCarrySlots;
  a.op_base = 0;
  a.slot_base = 0;
  if constexpr (C::MODE == 2) {
    a.out = static_cast<char *>(const_cast<void *>(uniform_ptr(cx.out_base)));
    a.k_stride = __builtin_amdgcn_readfirstlane(ka.k_stride_bytes32);
    a.fast_store = cx.full_chunk && ka.vec_ok != 0;
  }
  return a;
}

// Walks the records of one group (DFS order, sentinel at the end).
template <class C, bool TOTAL, class OPS>
__device__ __forceinline__ void fwalk(WalkCtx &cx, int node_begin, int &sink) {
  constexpr int EP = C::EP;
  const Hot a = hot_args<C>(cx);
  double f[C::MAXLV][EP];
#pragma unroll
  for (int k = 0; k < C::MAXLV; ++k)
#pragma unroll
    for (int i = 0; i < EP; ++i) f[k][i] = 0.0;
  uint32_t rec_off = (uint32_t)node_begin * 64u;
  int slot = 0;   // carry slots of the node: three per record of the group
  Rec8 nd = load_rec8(a.recs, rec_off);
  while (nd.level() != kRecSentinelLevel) {
    fnode<C, TOTAL, OPS>(cx, a, f, nd, rec_off, slot, sink);
    rec_off += 64u;
    slot += a.cps;
  }
}

// ---------------------------------------------------------------- node shapes as immediates
// Between the record loop and a plan that is straight-line code (below): a run-time compiled
// kernel of a LARGE plan knows the plan's node SHAPES - what the body of a node branches on:
// level, flags, output rows (GroupedProgram::shapes, the most frequent first).  The loop stays a loop over records, but a node's shape index (one more dword per
// node, IssArgs::shape_ids) selects a body compiled for that shape: frames indexed directly, no
// level dispatch, no flag or count tests - the rows, weights and offsets still come from the
// record.  Shapes beyond the SH::n compiled ones take the generic body.
template <class SH, int I>
struct SShape {
  static constexpr int d = SH::desc[I];
  static constexpr int level = d & 0xff, flags = (d >> 8) & 0xff;
  static constexpr int ne = (d >> 20) & 0xf;   // output rows; 3: more than two (read from the record)
};
template <class C, bool TOTAL, class OPS, class SH, int I>
__device__ __forceinline__ void fnode_shaped(WalkCtx &cx, const Hot &a, double (&f)[C::MAXLV][C::EP],
                                             Rec8 &nd, uint32_t me, int slot, int &sink) {
  using S = SShape<SH, I>;
  constexpr int EP = C::EP;
  static_assert(S::level < C::MAXLV, "a shape deeper than the kernel's register frames");
  const uint32_t rec_off = me + 64u;
  const int ne = S::ne < 3 ? S::ne : nd.emit_count();
  const uint32_t op_off = (uint32_t)nd.w[7] * (uint32_t)(OPS::n_padded * 32);
  const int t_rec = touch(a.recs, rec_off);
  int t_ops = 0;
  if constexpr ((S::flags & F_EMIT) != 0) {
    t_ops = touch(a.ops, op_off);
    freserve<C, OPS>(cx, ne * OPS::n);
  }
  cx.slot = slot;
  double s[EP];
  if constexpr ((S::flags & F_SLOW) != 0) {
    if constexpr ((S::flags & F_CHAIN) != 0) {
#pragma unroll
      for (int i = 0; i < EP; ++i) s[i] = f[S::level][i];
    } else if constexpr (S::level > 0) {
#pragma unroll
      for (int i = 0; i < EP; ++i) s[i] = f[S::level - 1][i];
    } else {
#pragma unroll
      for (int i = 0; i < EP; ++i) s[i] = C::SEMI != 1 ? 1.0 : 0.0;
    }
    slow_factors<C>(cx, rec_field(a.recs, me, 12), nd.fac_count(), s, s, false);
  } else {
    if constexpr ((S::flags & F_CHAIN) != 0) {
      mul_row_from<C>(cx, nd.w[2], f[S::level], s);
    } else if constexpr (S::level > 0) {
      mul_row_from<C>(cx, nd.w[2], f[S::level - 1], s);
    } else {
      double ones[EP];
#pragma unroll
      for (int i = 0; i < EP; ++i) ones[i] = C::SEMI != 1 ? 1.0 : 0.0;
      mul_row_from<C>(cx, nd.w[2], ones, s);
    }
    const int nf = nd.fac_count();
    if (nf > 1) {
      mul_rowp<C>(cx, nd.w[3], s);
      if (nf > 2) mul_rowp<C>(cx, nd.w[4], s);
      if (nf > 3) mul_rowp<C>(cx, nd.w[5], s);
    }
  }
  const int w1 = nd.w[1];
  if constexpr ((S::flags & F_NEED1) != 0) {
    double c[EP], x[EP];
    fscan<C>(cx, s, c, x, slot);
    if constexpr ((S::flags & (F_CHILDREN | F_NEED2)) == F_CHILDREN) {
#pragma unroll
      for (int i = 0; i < EP; ++i) f[S::level][i] = C::SEMI == 0 ? x[i] : c[i];
    }
    if constexpr ((S::flags & F_EMIT) != 0) {
      if constexpr (C::WEIGHTED && TOTAL) {
        const int emit_mul = ((w1 >> 24) & 0xff) - 1;
        mul_rowp<C>(cx, C::SEMI != 1 ? emit_mul : fac_arctic(emit_mul, -1), c);
        if constexpr (C::TOTALINC) {
          double xs[EP];
          previous_weighted<C>(cx, emit_mul, x, xs);
          fops_all<C, false, OPS>(cx, a, ne, op_off, me, c, xs, s);
        } else {
          fops_all<C, false, OPS>(cx, a, ne, op_off, me, c, x, s);
        }
      } else {
        fops_all<C, true, OPS>(cx, a, ne, op_off, me, c, x, s);
      }
    }
  }
  sink += t_rec + t_ops;
  nd = load_rec8(a.recs, rec_off);
  if constexpr (C::WEIGHTED && !TOTAL && (S::flags & F_NEED2) != 0) {
    const int z_mul = ((w1 >> 16) & 0xff) - 1;
    double c2[EP], x2[EP];
    mul_rowp<C>(cx, C::SEMI != 1 ? z_mul : fac_arctic(z_mul, 1), s);
    fscan<C>(cx, s, c2, x2, slot + 1);
#pragma unroll
    for (int i = 0; i < EP; ++i) f[S::level][i] = C::SEMI == 0 ? x2[i] : c2[i];
  }
}
// a chain of uniform tests, the most frequent shape first
template <class C, bool TOTAL, class OPS, class SH, int I>
__device__ __forceinline__ void fnode_by_shape(int shape, WalkCtx &cx, const Hot &a,
                                               double (&f)[C::MAXLV][C::EP], Rec8 &nd, uint32_t me,
                                               int slot, int &sink) {
  if constexpr (I < SH::n) {
    if (shape == I) {
      fnode_shaped<C, TOTAL, OPS, SH, I>(cx, a, f, nd, me, slot, sink);
      asm volatile("" ::: "memory");  // (keeps the bodies apart)
    } else {
      fnode_by_shape<C, TOTAL, OPS, SH, I + 1>(shape, cx, a, f, nd, me, slot, sink);
    }
  } else {
    fnode<C, TOTAL, OPS>(cx, a, f, nd, me, slot, sink);
  }
}
template <class C, bool TOTAL, class OPS, class SH>
__device__ __forceinline__ void fwalk_shaped(WalkCtx &cx, int node_begin, int &sink) {
  constexpr int EP = C::EP;
  const Hot a = hot_args<C>(cx);
  const void *shape_ids = uniform_ptr(cx.a->shape_ids);
  double f[C::MAXLV][EP];
#pragma unroll
  for (int k = 0; k < C::MAXLV; ++k)
#pragma unroll
    for (int i = 0; i < EP; ++i) f[k][i] = 0.0;
  uint32_t rec_off = (uint32_t)node_begin * 64u;
  int slot = 0;
  Rec8 nd = load_rec8(a.recs, rec_off);
  int shape = at_offset(shape_ids, rec_off >> 4)[0];
  while (shape >= 0) {   // (the sentinel's shape index is -1)
    const int next_shape = at_offset(shape_ids, (rec_off + 64u) >> 4)[0];   // (in flight over the node)
    fnode_by_shape<C, TOTAL, OPS, SH, 0>(shape, cx, a, f, nd, rec_off, slot, sink);
    shape = next_shape;
    rec_off += 64u;
    slot += a.cps;
  }
}

// ---------------------------------------------------------------- the plan as an immediate too
// A run-time compiled kernel may also know the PLAN (jit.cpp: a pipeline of at most
// kFusedStaticMaxNodes nodes): PG::w holds the records of the group program the launch will use,
// 16 words per record like NodeRec, PG::group_begin the groups' first records.  The walk is then
// straight-line code - a node's level, flags, factor rows, output rows, carry slot and op offsets
// are immediates, the register frames are indexed directly, nothing of a record is loaded or
// decoded - and what is left per node is the arithmetic, the thresholds of its ops and the feature
// window's bookkeeping.  Same sums in the same order as fwalk (same functions).
struct NoProg {
  static constexpr bool is_static = false, is_shaped = false, is_piece = false;
};
template <class PG, int PC>
struct SRec {
  static constexpr int w(int i) { return PG::w[PC * 16 + i]; }
  static constexpr int level = PG::w[PC * 16] & 0xff, flags = PG::w[PC * 16] >> 8;
  static constexpr int nf = PG::w[PC * 16 + 1] & 0xffff;
  static constexpr int z_mul = ((PG::w[PC * 16 + 1] >> 16) & 0xff) - 1;
  static constexpr int emit_mul = ((PG::w[PC * 16 + 1] >> 24) & 0xff) - 1;
  static constexpr int ne = PG::w[PC * 16 + 6];
};
// PC: the record; GB: the first record of its group (carry slots count from there)
template <class C, bool TOTAL, class OPS, class PG, int PC, int GB>
__device__ __forceinline__ void fwalk_static(WalkCtx &cx, const Hot &a, double (&f)[C::MAXLV][C::EP]) {
  using R = SRec<PG, PC>;
  constexpr int EP = C::EP;
  if constexpr (R::level != kRecSentinelLevel) {
    static_assert(R::level < C::MAXLV, "a record deeper than the kernel's register frames");
    const int slot = a.slot_base + (C::MULTI != 0 ? OPS::cps : kCarrySlots) * (PC - GB);
    constexpr uint32_t me = (uint32_t)PC * 64u;
    const uint32_t op_off = a.op_base + (uint32_t)R::w(7) * (uint32_t)(OPS::n_padded * 32);
    freserve<C, OPS>(cx, R::ne * OPS::n);
    cx.slot = slot;
    double s[EP];
    if constexpr ((R::flags & F_SLOW) != 0) {
      if constexpr ((R::flags & F_CHAIN) != 0) {
#pragma unroll
        for (int i = 0; i < EP; ++i) s[i] = f[R::level][i];
      } else if constexpr (R::level > 0) {
#pragma unroll
        for (int i = 0; i < EP; ++i) s[i] = f[R::level - 1][i];
      } else {
#pragma unroll
        for (int i = 0; i < EP; ++i) s[i] = C::SEMI != 1 ? 1.0 : 0.0;
      }
      slow_factors<C>(cx, R::w(12), R::nf, s, s, false);
    } else {
      if constexpr ((R::flags & F_CHAIN) != 0) {
        mul_row_from<C>(cx, R::w(2), f[R::level], s);
      } else if constexpr (R::level > 0) {
        mul_row_from<C>(cx, R::w(2), f[R::level - 1], s);
      } else {
        double ones[EP];
#pragma unroll
        for (int i = 0; i < EP; ++i) ones[i] = C::SEMI != 1 ? 1.0 : 0.0;
        mul_row_from<C>(cx, R::w(2), ones, s);
      }
      if constexpr (R::nf > 1) mul_rowp<C>(cx, R::w(3), s);
      if constexpr (R::nf > 2) mul_rowp<C>(cx, R::w(4), s);
      if constexpr (R::nf > 3) mul_rowp<C>(cx, R::w(5), s);
    }
    if constexpr ((R::flags & F_NEED1) != 0) {
      double c[EP], x[EP];
      fscan<C>(cx, s, c, x, slot);
      if constexpr ((R::flags & (F_CHILDREN | F_NEED2)) == F_CHILDREN) {
#pragma unroll
        for (int i = 0; i < EP; ++i) f[R::level][i] = C::SEMI == 0 ? x[i] : c[i];
      }
      if constexpr ((R::flags & F_EMIT) != 0) {
        if constexpr (C::WEIGHTED && TOTAL) {
          mul_rowp<C>(cx, C::SEMI != 1 ? R::emit_mul : fac_arctic(R::emit_mul, -1), c);
          if constexpr (C::TOTALINC) {
            double xs[EP];
            previous_weighted<C>(cx, R::emit_mul, x, xs);
            fops_all<C, false, OPS>(cx, a, R::ne, op_off, me, c, xs, s);
          } else {
            fops_all<C, false, OPS>(cx, a, R::ne, op_off, me, c, x, s);
          }
        } else {
          fops_all<C, true, OPS>(cx, a, R::ne, op_off, me, c, x, s);
        }
      }
    }
    if constexpr (C::WEIGHTED && !TOTAL && (R::flags & F_NEED2) != 0) {
      double c2[EP], x2[EP];
      mul_rowp<C>(cx, C::SEMI != 1 ? R::z_mul : fac_arctic(R::z_mul, 1), s);
      fscan<C>(cx, s, c2, x2, slot + 1);
#pragma unroll
      for (int i = 0; i < EP; ++i) f[R::level][i] = C::SEMI == 0 ? x2[i] : c2[i];
    }
    fwalk_static<C, TOTAL, OPS, PG, PC + 1, GB>(cx, a, f);
  }
}
// the group a unit walks is a run-time value: one straight-line walk per group
template <class C, bool TOTAL, class OPS, class PG, int G>
__device__ __forceinline__ void fwalk_static_group(WalkCtx &cx, int g0) {
  if constexpr (G < PG::groups) {
    if (g0 == G) {
      Hot a;
      const IssArgs &ka = *cx.a;
      a.recs = reinterpret_cast<const NodeRec *>(uniform_ptr(ka.recs));
      a.ops = reinterpret_cast<const FeatOp *>(uniform_ptr(ka.ops));
      a.n_ops = OPS::n;
      a.op_row_bytes = (uint32_t)(OPS::n_padded * 32);
      a.cps = OPS::cps;
      a.op_base = 0;
      a.slot_base = 0;
      double f[C::MAXLV][C::EP];
#pragma unroll
      for (int k = 0; k < C::MAXLV; ++k)
#pragma unroll
        for (int i = 0; i < C::EP; ++i) f[k][i] = 0.0;
      fwalk_static<C, TOTAL, OPS, PG, PG::group_begin[G], PG::group_begin[G]>(cx, a, f);
    } else {
      fwalk_static_group<C, TOTAL, OPS, PG, G + 1>(cx, g0);
    }
  }
}

// ---------------------------------------------------------------- a large plan in pieces
// plan.h, PiecedProgram: a kernel of this kind walks the items of ONE piece type - PG::w holds the
// records of the type's body, a forest of whole sub-tries as straight-line code (fwalk_static
// above, its output rows and carry slots counted from the item's) - behind a CHAIN, the path
// from the trie's root to the node the forest hangs below, walked by the record loop in frame 0.
// A unit = the items [unit_begin[u], unit_begin[u + 1]) on one series, one staging of its rows.
template <class C, bool TOTAL, class OPS, class PG>
__device__ __forceinline__ void fwalk_pieces(WalkCtx &cx, int unit, int &sink) {
  constexpr int EP = C::EP;
  Hot a = hot_args<C>(cx);
  a.n_ops = OPS::n;
  a.op_row_bytes = (uint32_t)(OPS::n_padded * 32);
  a.cps = C::MULTI != 0 ? OPS::cps : kCarrySlots;
  const void *items = uniform_ptr(cx.a->piece_items);
  const void *unit_begin = uniform_ptr(cx.a->piece_unit_begin);
  uint32_t it = (uint32_t)at_offset(unit_begin, (uint32_t)unit * 4u)[0] * 16u;
  const uint32_t it_end = (uint32_t)at_offset(unit_begin, (uint32_t)unit * 4u)[1] * 16u;
  double f[C::MAXLV][EP];
#pragma unroll
  for (int k = 1; k < C::MAXLV; ++k)
#pragma unroll
    for (int i = 0; i < EP; ++i) f[k][i] = 0.0;
  for (; it < it_end; it += 16u) {
    cptr<int32_t> iw = at_offset(items, it);
    uint32_t rec_off = (uint32_t)iw[0];
    const int row_base = iw[1], node_base = iw[2];
    // the chain: every node continues frame 0 (the semiring's one in front of the first)
#pragma unroll
    for (int i = 0; i < EP; ++i) f[0][i] = C::SEMI != 1 ? 1.0 : 0.0;
    a.op_base = 0;
    int slot = a.cps * node_base;
    Rec8 nd = load_rec8(a.recs, rec_off);
    while (nd.level() != kRecSentinelLevel) {
      fnode<C, TOTAL, OPS>(cx, a, f, nd, rec_off, slot, sink);
      rec_off += 64u;
      slot += a.cps;
    }
    a.op_base = (uint32_t)row_base * a.op_row_bytes;
    a.slot_base = slot;
    fwalk_static<C, TOTAL, OPS, PG, 0, 0>(cx, a, f);
  }
}

// Stages the rows of one time chunk of series n into LDS (coalesced 16-byte units, the loads of
// kStageRows rows in flight before the first LDS write); with a fused preparation the rows are
// formed from the RAW input on the way (INC / NEW(INC) / STD, see IssArgs::prep).
template <class C>
__device__ __forceinline__ void stage_chunk(const WalkCtx &cx, const IssArgs &a, int64_t n, int64_t t0,
                                            double *rows_w) {
  const int tid = cx.tid;
  for (int r0 = 0; r0 < a.R; r0 += kStageRows) {
    constexpr int U = C::CHUNK / 2 / kWalkThreads;
    vd2 v[kStageRows][U];
#pragma unroll
    for (int rr = 0; rr < kStageRows; ++rr) {
      if (r0 + rr < a.R) {
        const int src = as_const(a.row_src)[r0 + rr];
        if (a.prep != nullptr && src >= 0) {
          // INC (x[t] - x[t - lag], zero-padded: fruits/cache.py:8-13), NEW(INC) (the prepared
          // dimension names a raw dimension and a lag) and STD's apply step ((x - mean) /
          // (std + eps), fruits/preparation/transform.py:141-147; statistics: row_stats_kernel)
          const int raw = as_const(a.prep)[4 * src], lag = as_const(a.prep)[4 * src + 1];
          const bool standardise = as_const(a.prep)[4 * src + 2] != 0;
          const double *gp = a.X + (n * a.D + raw) * a.T;
          double mean = 0.0, den = 1.0;
          if (standardise) {
            mean = as_const(a.stats)[(n * a.n_prep + src) * 2];
            den = as_const(a.stats)[(n * a.n_prep + src) * 2 + 1];
          }
#pragma unroll
          for (int k = 0; k < U; ++k) {
            const int i = 2 * (k * kWalkThreads + tid);
            const int64_t t = t0 + i;
            double e0 = 0.0, e1 = 0.0;
            if (t < a.T) e0 = gp[t];
            if (t + 1 < a.T) e1 = gp[t + 1];
            if (lag > 0) {
              e0 = (t >= lag && t < a.T) ? e0 - gp[t - lag] : 0.0;
              e1 = (t + 1 >= lag && t + 1 < a.T) ? e1 - gp[t + 1 - lag] : 0.0;
            }
            if (standardise) {
              e0 = (e0 - mean) / den;
              e1 = (e1 - mean) / den;
            }
            // (elements beyond T stay what the unfused path stages there: zeros)
            v[rr][k] = vd2{t < a.T ? e0 : 0.0, t + 1 < a.T ? e1 : 0.0};
          }
        } else {
          const double *gp =
              src >= 0 ? a.X + (n * a.D + src) * a.T
                       : a.aux + (int64_t)(-src - 1) * a.aux_tab_stride + n * a.aux_n_stride;
#pragma unroll
          for (int k = 0; k < U; ++k) {
            const int i = 2 * (k * kWalkThreads + tid);
            const int64_t t = t0 + i;
            v[rr][k] = vd2{0.0, 0.0};
            if (a.vec_ok) {
              if (cx.full_chunk || t < a.T) v[rr][k] = *reinterpret_cast<const vd2 *>(gp + t);
            } else {
              if (t < a.T) v[rr][k].x = gp[t];
              if (t + 1 < a.T) v[rr][k].y = gp[t + 1];
            }
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
}

// LDS of the fused kernel: rows [R][CHUNK] | totals window [2][8] | tails [2][4] | carries
// [carry_slots] (MULTI) | feature window (values, populations, columns)
// (No waves-per-SIMD attribute: every instance fits 128 VGPRs - four waves - without one, and
// with it the 8-level one-chunk instances spilled a few registers for nothing.)
template <class C, bool TOTAL, class OPS = DynOps, class PG = NoProg>
__global__ __launch_bounds__(kWalkThreads) void iss_fused_kernel(const IssArgs a) {
  static_assert((C::MODE == 1 || C::MODE == 2) && C::TEAM == 4 && (C::P == 1 || C::MODE == 2) && C::MULTI != 2,
                "fused (MODE 1) or lean materialising (MODE 2) configuration");
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  WalkCtx cx;
  cx.a = &a;
  cx.rows = lds;
  cx.tot = lds + (int64_t)a.R * C::CHUNK;
  cx.tail = cx.tot + 16;
  cx.carry = cx.tail + 8;
  cx.tid = tid;
  cx.lane = tid & 63;
  cx.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  cx.team = 0;
  cx.buf = 0;
  cx.tail_buf = 0;
  if constexpr (C::MODE == 2) {
    if (tid < 6) cx.tot[(tid / 3) * 8 + tid % 3] = semi_zero<C::SEMI>();
  } else {
    double *fw = cx.carry + (C::MULTI == 1 ? a.carry_slots : 0);
    cx.fl_val = (lds_f64 *)fw;
    cx.fl_cnt = (lds_f64 *)(fw + a.feat_window);
    cx.fl_col = nullptr;   // (columns follow from the walk order: feat_flush<C, true>)
    for (int sl = tid; sl < a.feat_window; sl += kWalkThreads) {
      cx.fl_val[sl] = 0.0;
      if (a.has_mpi) cx.fl_cnt[sl] = 0.0;
    }
    // the identity slots of both totals windows
    if (tid < 6) cx.tot[(tid / 3) * 8 + tid % 3] = semi_zero<C::SEMI>();
    cx.fused_used = 0;
    cx.fslot = 0;
    cx.feat_window = a.feat_fits ? 0x3fffffff : a.feat_window;
  }
  // one workgroup per unit (series n, group g0); with the xcd numbering the groups of one
  // series meet in one XCD's L2 (speed only)
  const int u = blockIdx.x;
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
  int node_begin = 0;
  if constexpr (!PG::is_piece) node_begin = as_const(a.group_begin)[g0];
  int sink = 0;
  cx.pc_begin = node_begin;
  cx.series = n;   // (feature rows, cut rows: addressed from it where they are needed)
  for (int chunk = 0; chunk < a.nchunks; ++chunk) {
    const int64_t t0 = (int64_t)chunk * C::CHUNK;
    cx.t0 = t0;
    cx.parity = chunk & 1;
    cx.first_chunk = chunk == 0;
    // (OPS::full_chunks - a run-time compiled kernel for a series length that is a multiple of the
    // chunk: the ops' paths for ragged chunks are not even compiled)
    cx.full_chunk = OPS::full_chunks || t0 + C::CHUNK <= a.T;
    if (chunk > 0) lds_barrier();  // all reads of the old rows are done
    stage_chunk<C>(cx, a, n, t0, lds);
    __syncthreads();
    if constexpr (C::MODE == 2) {
      cx.out_base = a.out + n * a.out_n_stride + t0;
      fwalk<C, TOTAL, OPS>(cx, node_begin, sink);
    } else {
      cx.fused_used = 0;  // same slots in every chunk
      if constexpr (PG::is_piece) {
        static_assert(OPS::is_static && C::MODE == 1, "a piece comes with static ops");
        cx.frow0 = as_const(a.piece_unit_row0)[g0];
        fwalk_pieces<C, TOTAL, OPS, PG>(cx, g0, sink);
      } else {
      cx.frow0 = as_const(a.group_row_begin)[g0];
      if constexpr (PG::is_static) {
        static_assert(OPS::is_static && C::MODE == 1, "a static plan comes with static ops");
        fwalk_static_group<C, TOTAL, OPS, PG, 0>(cx, g0);
      } else if constexpr (PG::is_shaped) {
        static_assert(OPS::is_static && C::MODE == 1, "node shapes come with static ops");
        fwalk_shaped<C, TOTAL, OPS, PG>(cx, node_begin, sink);
      } else {
        fwalk<C, TOTAL, OPS>(cx, node_begin, sink);
      }
      }
      // a unit whose features fit the window keeps them there over its time chunks; else every
      // chunk leaves its share (added onto the earlier chunks' in global memory)
      if (!a.feat_fits || chunk + 1 == a.nchunks) feat_flush<C, true>(cx, !a.feat_fits && chunk > 0);
    }
  }
  if (sink == 0x7fffffff) (C::MODE == 2 ? a.out : a.feats)[0] = 0.0;   // (keeps the cache-touching loads alive)
}

}  // namespace fr

// Types and constants shared by the host plan compiler, the C ABI and the DEVICE code of the
// walk kernels.  Free of host-only headers: this file, walk_scan.h and walk_device.h are also
// the source text the run-time compiler (jit.cpp, hipRTC) builds static programs from.
#pragma once
#ifndef __HIPCC_RTC__
#include <cstdint>
#else   // hipRTC has no <cstdint>; the layout of IssArgs must be the host's (LP64)
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef signed int int32_t;
typedef unsigned int uint32_t;
typedef long int64_t;
typedef unsigned long uint64_t;
#endif

namespace fr {

// node flags
constexpr int32_t F_CHAIN = 1;     // only child: processed in place in its parent's frame
constexpr int32_t F_CHILDREN = 2;  // the exclusive prefix of this node is consumed by children
// what a node scans (resolved by the host, read by the fused walk): NEED1 - its letters' sum (it
// has output rows, or children that continue from it); NEED2 - non-total weighting: a second
// scan over s * exp(+g alpha) for the children
constexpr int32_t F_NEED1 = 8;
constexpr int32_t F_NEED2 = 16;
constexpr int32_t F_EMIT = 32;     // the node has output rows
// factor code: LDS row in the low 7 bits.  Reals: bit 7 = divide instead of multiply
// (one code per occurrence of a letter).  Arctic: bits 8-15 = signed multiplier el of
// the ADDED term el * row (one code per dimension of the extended letter).
constexpr int32_t FAC_DIV = 0x80;
constexpr int32_t FAC_ROW_MASK = 0x7f;
constexpr int kSemiReals = 0, kSemiArctic = 1, kSemiBayesian = 2;
constexpr int32_t fac_arctic(int row, int el) { return row | ((el & 0xff) << 8); }
// letter-sum plans (Arctic argmax): the letter's terms collected so far are added to the
// prefix BEFORE this factor is applied (the weight term follows `tmp = tmp + C`)
constexpr int32_t FAC_FOLD = 1 << 16;

constexpr int kMaxLevels = 8;   // deepest register-frame stack a kernel variant supports

// Device form of a node: one 64-byte record = one s_load_dwordx16, every field a
// whole dword so the walk needs no bit unpacking.  The records of a group are
// contiguous in DFS order and end with a sentinel (level 0xff), so the walk
// prefetches record pc+1 while it works on record pc and needs no bounds checks.
constexpr int kRecInlineFactors = 4;   // multiply-only factors held in the record
constexpr int kRecInlineEmits = 2;
constexpr int kRecSentinelLevel = 0xff;
constexpr int32_t F_SLOW = 4;          // a division or more than kRecInlineFactors factors:
                                       // the factor table is walked instead
struct NodeRec {
  int32_t w[16];
  // w[0]  level | flags << 8          w[1]  fac_count | (z_mul + 1) << 16 | (emit_mul + 1) << 24
  // w[2..5]  inline factor rows       w[6]  emit_count
  // w[7..8]  inline emit rows         w[9]  node id (carry slot)
  // w[10] emit_mul row or -1          w[11] z_mul row or -1
  // w[12] fac_begin (factor table: row | FAC_DIV)   w[13] emit_begin (emit-row table)
};

constexpr int kSchedStage = 0xfe;      // entry kind: complete staged row w[1]
constexpr int kSchedEmits = 0xfc;      // entry kind: w[1] more output rows (w[2..]) of the node entry before it
constexpr int kSchedEmitsPerEntry = 14;
constexpr int kSchedPrefetch = 0xfd;   // entry kind: load the next unit's rows (registers are free)
constexpr int kStaticMaxRows = 4;      // staged rows held in registers while in flight
constexpr int kStaticMaxFrames = 4;    // open prefixes
constexpr int kStaticMaxNodes = 32;
// fused pipelines: plans of at most this many nodes may be compiled with the plan as straight-line
// code (walk_fused.h, fwalk_static).  The limit is the compiler's time, not the instruction cache:
// one function of ~1.7 KB per node - 33 nodes 7 s, 115 nodes (of_weight(4,2), 200 KB, a quarter
// faster than the record loop although two CUs share 64 KB of cache) 15 s, growing faster than
// linearly.
constexpr int kFusedStaticMaxNodes = 128;
// larger plans: this many of their most frequent node shapes get a body of their own in the
// pipeline's kernel (fwalk_shaped); the others take the generic body
constexpr int kFusedShapes = 24;
// ... or, the default since round 4, the plan in PIECES of at most this many nodes, each piece
// type straight-line code in a kernel of its own (plan.h, PiecedProgram).  Measured (config 4 /
// config 5, hipRTC on the GPU box's host, cold cache): pieces of <= 128 nodes 6.57 / 11.65 ms
// and 17 s of compiler, <= 64 nodes 6.74 / 12.03 ms and 12 s (the record loop: 8.46 / 15.57 ms).
constexpr int kFusedPieceNodes = 128;
// nodes of a unit (a few items of one type on one series; one staging of the series' rows).
// Measured on config 4 with pieces of <= 64 nodes: units of one item 7.76 ms, <= 96 nodes 7.57,
// <= 160 7.25, <= 256 8.31 (long workgroups: the last round of every launch runs half empty)
constexpr int kPieceUnitNodes = 160;

constexpr int kWalkThreads = 256;

// LDS bytes of a feature window of `slots` slots (8-byte value, 8-byte population with MPI
// sieves, 4-byte column - not in the fused walk, whose columns follow from the walk order; slots even)
constexpr uint64_t feat_window_bytes(int slots, bool mpi, bool cols = true) {
  return (uint64_t)slots * ((mpi ? 16u : 8u) + (cols ? 4u : 0u));
}

constexpr int FR_SIEVE_NPI_K = 0;
constexpr int FR_SIEVE_MPI_K = 1;
constexpr int FR_SIEVE_END_K = 2;

// One feature of one iterated sum, everything resolved on the host (32 bytes, read
// with one scalar load): END picks the value at index `lo`; NPI / MPI look at
// t in [lo, hi) and values in (qlo, qhi] of the inc-times differenced row.
struct FeatOp {
  int32_t kind_inc;   // kind | inc << 8 | per-series cuts << 16 (lo / hi are then slots of the
                      // series' row of IssArgs::series_cuts)
  int32_t col;        // absolute feature column
  int32_t lo, hi;
  double qlo, qhi;
};

struct IssArgs {
  const double *X;          // (N, D, T)
  const double *aux;        // exp tables [2A][aux rows][T] or nullptr
  double *out;
  double *carry;            // (N, 2*total_nodes) chunk carries or nullptr (single chunk)
  const NodeRec *recs;       // 64-byte aligned
  const int32_t *factors;
  const int32_t *emit_rows;
  const int32_t *slot_rows;        // fused walk: output rows in the order a group emits them,
  const int32_t *group_row_begin;  // and where a group's rows start in it (GroupedProgram)
  const int32_t *shape_ids;        // per record: index of its shape (GroupedProgram::shapes; fwalk_shaped)
  const int32_t *group_begin;
  const int32_t *row_src;
  int64_t N, D, T;
  int64_t aux_tab_stride;   // elements between two exp tables
  int64_t aux_n_stride;     // T for per-series lookups, 0 for a broadcast lookup
  int64_t out_k_stride, out_n_stride;
  int32_t G;                // groups per series
  int32_t R;                // staged rows
  int32_t total_nodes;
  int32_t vec_ok;           // 16-byte accesses are aligned
  int32_t nchunks;
  int32_t xcd_map;          // strided schedule: the groups of one series share an XCD
  int32_t persistent;       // grid = one resident round of workgroups
  int32_t carry_slots;      // carry_per_node * (records of the largest group): LDS carry slots
  int32_t carry_per_node;   // fused walk: 3, + 2 per differencing order >= 3 of the sieves (0 = 3)
  int32_t carry_in_lds;     // multi-chunk carries fit in LDS
  int32_t letter_sum;       // Arctic: sum a letter's terms before adding them to the prefix
  int32_t semiring;         // kSemiReals / kSemiArctic
  int32_t prefetch_next;    // units of at most this many nodes touch the next unit's rows (0: off)
  int32_t packed;           // short series: wave-per-series kernel (walk_packed.h)
  // fused sieve epilogue (MODE 1 kernels): features instead of the (K,N,T) tensor
  const FeatOp *ops;        // (K, n_ops_padded) feature ops per output row, 64-byte aligned rows
  double *feats;            // (N, feat_stride) features; every column is written, none is read first
  double *cnt;              // same shape: band population of MPI features
  int64_t feat_stride;
  int32_t n_ops, n_ops_padded;
  // LDS feature window of the cooperative kernels (walk_device.h, feat_flush): slots of
  // {value, population (has_mpi), column}; feat_fits: every group's features fit the window,
  // so a unit flushes once, after its last time chunk
  int32_t feat_window, feat_fits, has_mpi;
  // per-series segment boundaries (coquantile cuts, fruits/sieving/segment.py:51-64): (N,
  // cut_slots) int32 in [0, T], the rows of every such sieve sorted; nullptr: none
  const int32_t *series_cuts;
  int32_t cut_slots;
  // CosWISS programs (coswiss.h): letters of word w = [cw_letter_begin[w], ..+1), factors of
  // letter l = factors[cw_fac_begin[l] .. cw_fac_begin[l+1]) as dimension | FAC_DIV; aux
  // holds the (F, 2, T) sin / cos tables
  const int32_t *cw_letter_begin;
  const int32_t *cw_fac_begin;
  int32_t cw_W, cw_F, cw_total;
  // randomised CosWISS variants (fruits/iss/cos.py:51-164): cw_mask (W, F, cw_Lmax, T) holds
  // 0.0 where the summand of letter k is dropped before its cumsum (dropout), else nullptr;
  // cw_x_unit_stride != 0: unit j = word * F + freq reads its own input at X + j * stride
  // (the ffn-transformed copies), else all units share X
  const double *cw_mask;
  int64_t cw_x_unit_stride;
  int32_t cw_Lmax;
  int32_t lds_pad;          // experiments: extra dynamic LDS bytes per workgroup (fewer resident ones)
  int32_t high_order;       // 1: a fused sieve differences more than twice (WalkCfg::HIGHORD on multi-chunk series)
  int32_t total_inc;        // 1: totally weighted plan whose fused sieves difference (WalkCfg::TOTALINC)
  int32_t total_weighting;  // 1: the plan's weighting is total (the fused walk is compiled per mode)
  int32_t nt_input;         // 1: stage the rows of X with non-temporal loads (interpreter, one group)
  int32_t lean;             // 1: materialising launch through the fused walk's node loop (walk_fused.h, MODE 2)
  int32_t static_prog;      // != 0: the records equal pre-compiled static program #n (walk_static_inst.hip)
  uint32_t k_stride_bytes32; // out_k_stride * 8 when that fits 32 bits (and is > 0), else 0
  int32_t *resident_out;    // HOST pointer; non-null: the launcher stores the number of resident
                            // workgroups of the kernel it would launch there and launches nothing
  // fused preparation (MODE 1 cooperative kernels): X is the RAW (N, D, T) input and the
  // staging forms the prepared rows on the fly.  prep[4 * d'] for prepared dimension d':
  // {raw dimension, increment lag (0: none), standardise (0 / 1), unused}; stats holds
  // (N, n_prep, 2) = {mean, std + eps} of the prepared rows (written by row_stats_kernel)
  const int32_t *prep;
  const double *stats;
  int32_t n_prep;
  // a plan in pieces (plan.h, PiecedProgram; walk_fused.h, fwalk_pieces): the launch of ONE piece
  // type - G = its units per series, recs / emit_rows its own tables, ops / feats in walk order
  // (slot_rows == nullptr: a window slot's output row is its walk position)
  const int32_t *piece_items;       // 4 words per item: chain byte offset, first body row, nodes in front, 0
  const int32_t *piece_unit_begin;  // G + 1 offsets into the items
  const int32_t *piece_unit_row0;   // walk position of a unit's first output row
  unsigned long long *dbg;  // diagnostic stamps (timing build only)
  int32_t debug;            // timing experiments (FRUITS_HIP_DEBUG), 0 in production
};

}  // namespace fr

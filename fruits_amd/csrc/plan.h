// Host-side plan compiler for the ISS trie walk (internal header).
//
// A plan is the device "program" that replaces the reference's Python word loop
// (_calculate_ISS, fruits/iss/iss.py:21-67): the words are merged into a prefix
// trie (what CachePlan, fruits/iss/cache.py:6-81, only uses to drop duplicate
// OUTPUTS is used here to drop duplicate WORK), laid out in DFS preorder and
// annotated with register-frame levels so a workgroup can walk it with the
// running prefixes held in registers.
#pragma once
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "walk_types.h"

namespace fr {

struct NodeDesc {  // 32 bytes, read by the kernel with scalar loads
  int32_t level;       // register frame this node writes (reads level-1; level 0 reads ones)
  int32_t flags;       // F_*
  int32_t fac_begin;   // [fac_begin, fac_begin+fac_count) in the factor table
  int32_t fac_count;
  int32_t emit_begin;  // [emit_begin, emit_begin+emit_count) in the emit-row table
  int32_t emit_count;
  int32_t emit_mul;    // LDS row multiplied into emitted values (total weighting) or -1
  int32_t z_mul;       // LDS row multiplied into the summand of the child scan (non-total) or -1
};

struct GroupedProgram {       // node order for one choice of G (groups per series)
  int groups = 0;
  std::vector<NodeRec> recs;          // per group: its units' nodes, then a sentinel
  std::vector<int32_t> group_begin;   // G+1 offsets into recs
  // output rows in the order a group's walk emits them (the fused walk's feature window
  // maps slot -> column through it when it leaves: walk_device.h, feat_flush) + G offsets
  std::vector<int32_t> slot_rows, group_row_begin;
  // SHAPES of the nodes (walk_fused.h, fwalk_shaped): what a node's body branches on - level,
  // flags, output rows - as one word, the distinct ones by falling
  // frequency (`shapes`), and per record the index of its shape (`shape_ids`; the sentinel: -1)
  std::vector<int32_t> shapes, shape_ids;
  // device copies
  void *d_blob = nullptr;
  const NodeRec *d_recs = nullptr;
  const int32_t *d_group_begin = nullptr;
  const int32_t *d_slot_rows = nullptr, *d_group_row_begin = nullptr;
  const int32_t *d_shape_ids = nullptr;
  const int32_t *d_factors = nullptr;
  const int32_t *d_emit_rows = nullptr;
  const int32_t *d_row_src = nullptr;
  const float *d_alphas = nullptr;
};

// Program of a cosine weighted ISS (fruits/iss/cos.py): the letters of every word as
// factor codes (dimension | FAC_DIV, one per occurrence, ascending dimension) plus the
// frequencies; evaluated by coswiss_kernel (coswiss.h), one unit per (word, frequency).
struct CosProgram {
  int W = 0, F = 0, exponent = 0;
  bool total = false;
  std::vector<float> freqs;
  std::vector<int32_t> letter_begin;   // W+1
  std::vector<int32_t> fac_begin;      // letters+1
  std::vector<int32_t> factors;
  void *d_blob = nullptr;
  const int32_t *d_letter_begin = nullptr, *d_fac_begin = nullptr, *d_factors = nullptr;
  const float *d_freqs = nullptr;
  // randomised variants: dropout mask (W, F, Lmax, mask_T) of 1.0 / 0.0, per-unit inputs
  void *d_mask = nullptr;
  int Lmax = 0;
  int64_t mask_T = 0;
  int64_t x_unit_stride = 0;
};

// A large plan cut into PIECES for the fused walk (walk_fused.h, fwalk_pieces): straight-line
// code is what makes the fused walk fast (no record decode, no level dispatch), but one
// function over a whole plan of a thousand nodes does not compile in useful time.  So the trie
// is covered by items: a CHAIN - the path from the root to a node P, walked by the record loop,
// leaving P's prefix in frame 0 - and a BODY, a forest of whole sub-tries below P as
// straight-line code with its levels counted from P.  Bodies that are equal (same letters, same
// shape: in of_weight(w, d) every sub-trie below a prefix of the same weight) are ONE type,
// compiled once (one kernel per type; a unit = a few items of one type on one series).  The
// output rows are numbered in WALK order (type by type, unit by unit, a chain's rows, then the
// body's in record order), so a body addresses its thresholds relative to its first row and a
// unit's feature columns are contiguous; `row_of_walk` maps back.
struct PieceType {
  std::vector<int32_t> body_w;      // 16 words per record (rebased levels, body-relative output
                                    // rows), sentinel included - the immediates of the type's kernel
  int body_nodes = 0, body_rows = 0, levels = 0;
  std::vector<NodeRec> recs;        // device records: the body's (record p at byte 64 p), then the
                                    // chains of the items, each closed by a sentinel
  std::vector<int32_t> emit_rows;   // output rows beyond the two a record holds (body: relative)
  std::vector<int32_t> items;       // 4 words per item: chain byte offset in recs, walk position of
                                    // the body's first row, nodes of the unit in front of the item, 0
  std::vector<int32_t> unit_begin;  // U + 1 offsets into the items
  std::vector<int32_t> unit_row0;   // walk position of a unit's first output row
  int max_unit_nodes = 0, max_unit_rows = 0, widest_node = 0;   // (carry slots, feature window)
  void *d_blob = nullptr;
  const NodeRec *d_recs = nullptr;
  const int32_t *d_emit_rows = nullptr, *d_items = nullptr, *d_unit_begin = nullptr,
                *d_unit_row0 = nullptr;
  int units() const { return (int)unit_begin.size() - 1; }
};
struct PiecedProgram {
  bool ok = false;
  int max_piece = 0;
  std::vector<PieceType> types;
  std::vector<int32_t> row_of_walk;   // K entries: the output row at walk position q
  int chain_nodes = 0;                // node executions spent in chains (all types, per series)
};

struct Plan {
  CosProgram *cos = nullptr;  // non-null: a CosWISS program (no trie, K = W*F)
  int W = 0;
  int weighting = 0;
  int semiring = kSemiReals;
  bool shared = true;
  // Arctic: the terms el * Z[dim] of one extended letter are summed FIRST and their sum is
  // added to the prefix (the argmax body, fruits/iss/semiring.py:252-256) instead of being
  // added to the prefix one by one (:294-295) - the two round differently
  bool letter_sum = false;
  int K = 0;                 // output rows
  int levels = 0;            // register frames needed
  int max_dim = 0;           // highest dimension referenced (1-based)
  std::vector<NodeDesc> nodes;       // DFS preorder, unit after unit
  std::vector<int32_t> unit_of;      // unit (root sub-trie) index of each node
  std::vector<int32_t> unit_begin;   // U+1 offsets into nodes
  std::vector<double> unit_cost;
  std::vector<int32_t> factors;
  std::vector<int32_t> emit_rows;
  std::vector<int32_t> row_src;      // LDS row -> source: d (>=0) = X dimension d, -(1+j) = aux table j
  // distinct alphas.  Reals: aux table 2a = exp(+g*alpha_a), 2a+1 = exp(-g*alpha_a);
  // Arctic: aux table a = g*alpha_a
  std::vector<float> alphas;
  int aux_tables() const { return (int)alphas.size() * (semiring == kSemiArctic ? 1 : 2); }
  // Bayesian (max, x) shares the Reals encoding of letters and exp weights
  bool multiplicative() const { return semiring != kSemiArctic; }
  int dims_used = 0;
  std::map<int, GroupedProgram> programs;  // per G
  std::map<int, PiecedProgram> pieced;     // per largest piece (nodes)
  // pre-compiled static programs for 1 / 2 / 3 groups per series (1 + index; 0: none;
  // -1: not looked up yet)
  int static_prog[4] = {-1, -1, -1, -1};
  void *jit = nullptr;   // run-time compiled static programs (capi.cpp: JitState), or nullptr
  int device = -1;     // HIP device the uploaded tables live on (-1: nothing uploaded yet)
  std::mutex mu;       // guards `programs`, `cos->d_blob` and `device` (uploads at run time)

  int units() const { return (int)unit_begin.size() - 1; }
  int rows_staged() const { return (int)row_src.size(); }
};

// Builds the plan; returns nullptr and sets `err` on invalid input.
Plan *build_plan(int W, const int32_t *exps, const int32_t *L, const int32_t *Dw,
                 const float *alpha, const int32_t *depth, int weighting, int flags,
                 std::string &err);
// CosWISS program of W simple words x F frequencies (rows word-major, cos.py:167-181).
Plan *build_coswiss_plan(int W, const int32_t *exps, const int32_t *L, const int32_t *Dw, int F,
                         const float *freqs, int exponent, int total, std::string &err);
// Schedule of a static program (walk.h, walk_static): the whole plan as one group, nodes in
// an order that needs the staged rows as late as possible, register frames named explicitly.
struct StaticSchedule {
  bool ok = false;                     // the plan qualifies (small, unweighted, inline letters)
  int groups = 0, rows = 0, frames = 0;
  std::vector<NodeRec> entries;        // per group: node / stage entries, then a sentinel
  std::vector<int32_t> group_begin;    // first entry of every group
  std::vector<int32_t> group_rows;     // bit r: the group reads staged row r
  std::vector<int32_t> row_src;
};
StaticSchedule static_schedule(Plan &p, int G);
// Node order for G groups (LPT assignment of units to groups), cached in the plan.
GroupedProgram &grouped(Plan &p, int G);
// The plan in pieces of at most `max_piece` nodes (see PiecedProgram), cached in the plan;
// !ok: the plan has no such cover (CosWISS, letter sums, nothing to walk).
// (`unit_nodes`: nodes of a unit - a few items of one type on one series; 0: the default.  The
// first call for a `max_piece` decides.)
PiecedProgram &pieced(Plan &p, int max_piece, int unit_nodes = 0);

}  // namespace fr

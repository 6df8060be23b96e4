// Run-time compiled static programs (hipRTC): what static_programs.h holds for the standard
// word sets, for ANY plan the scheduler accepts.  See jit.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "plan.h"
#include "walk_types.h"

namespace fr {

struct JitProgram {
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;
  int groups = 0;
  size_t lds_bytes = 0;
  int per_cu = 0;          // resident workgroups per CU
  int device = -1;
};

// The HIP source of the static program of `sc` (self-contained: device headers + schedule).
std::string jit_source(const StaticSchedule &sc);
// Compiles `sc` for gfx950 (code object in `code`); false + `err` when hipRTC is missing or
// the compilation fails.  Needs no GPU.
// `from_cache`: set when the code object came from the disk cache (a caller whose load of it
// fails drops the file - jit_cache_drop - and compiles again).
bool jit_compile(const StaticSchedule &sc, std::string &code, std::string &err,
                 bool *from_cache = nullptr);
void jit_cache_drop(const StaticSchedule &sc);
// Loads a compiled program on the current device.
bool jit_load(const std::string &code, const StaticSchedule &sc, JitProgram &out, std::string &err);
void jit_unload(JitProgram &p);

// The fused walk of one pipeline with its sieves as compile-time constants (walk_fused.h, JitOps).
struct FusedOps {        // per op of an output row, the same for every row
  std::vector<int32_t> w0, lo, hi;   // FeatOp::kind_inc (kind, differencing order, shape), cuts
  int n_padded = 0;                  // ops per row of the device table (row stride / 32 bytes)
  int cps = 3;                       // carry slots per node (IssArgs::carry_per_node)
  bool window_fits = false;          // a piece type whose largest unit fits the feature window
  bool full_chunks = false;          // the series length is a multiple of the time chunk
};
// The plan as an immediate too (walk_fused.h, fwalk_static): the records of the group program
// the launch will use (16 words each, sentinels included) and the groups' first records.
struct FusedPlan {
  std::vector<int32_t> w;
  std::vector<int32_t> group_begin;   // groups entries
  // ... or, for a large plan, only its most frequent node SHAPES (walk_fused.h, fwalk_shaped;
  // GroupedProgram::shapes): `w` empty, `n_groups` = the group program they were counted on
  std::vector<int32_t> shapes;
  int n_groups = 0;
  // ... or ONE piece type of a plan in pieces (plan.h, PiecedProgram; walk_fused.h, fwalk_pieces):
  // the records of its body (`w`), `piece` set
  bool piece = false;
  int groups() const { return piece ? 0 : (shapes.empty() ? (int)group_begin.size() : n_groups); }
};
struct FusedKey {        // the WalkCfg instantiation a (plan, series length, sieves) selects
  int E, LV, MULTI, W, SEMI, TI, TOTAL, HO;
  uint32_t packed() const {
    return (uint32_t)(E | LV << 4 | MULTI << 8 | W << 9 | SEMI << 10 | TI << 12 | TOTAL << 13 | HO << 14);
  }
};
std::string jit_fused_source(const FusedOps &ops, const FusedPlan *plan = nullptr);
// Compiles (or takes from the disk cache) and loads on the current device; needs hipRTC.
// `plan`: the straight-line walk of exactly this group program (JitProgram::groups says which).
// `cache_only`: only what the disk cache holds (milliseconds); a miss is jit_not_cached(err).
bool jit_fused(const FusedOps &ops, const FusedKey &key, JitProgram &out, std::string &err,
               const FusedPlan *plan = nullptr, bool cache_only = false);
bool jit_not_cached(const std::string &err);
// Compiles into the directory `dir` (the cache's file format) without loading anything: the
// kernels shipped with the build (fruits_amd/gen_bundle.py).  Needs hipRTC, no GPU.
bool jit_fused_into(const FusedOps &ops, const FusedKey &key, const FusedPlan *plan, const char *dir,
                    std::string &err);
hipError_t jit_launch_fused(const JitProgram &p, const IssArgs &a, size_t lds_bytes, hipStream_t st);
hipError_t jit_launch(const JitProgram &p, const IssArgs &a, hipStream_t st);

}  // namespace fr

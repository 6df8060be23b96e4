// Walk kernels whose program is a compile-time constant (walk.h, walk_static).  One
// translation unit per pre-generated program (STATIC_PROG = its index in
// static_programs.h, fruits_amd/gen_static.py), plus the registry (STATIC_REGISTRY) that
// capi.cpp asks whether the plan it is about to run is one of them.
#include "walk.h"

#include <cstring>

#include "static_programs.h"

namespace fr {

#define SP_CAT2(a, b) a##b
#define SP_CAT(a, b) SP_CAT2(a, b)

#ifdef STATIC_REGISTRY
#define SP_DECL(i) hipError_t walk_static_launch_##i(const IssArgs &, hipStream_t);
SP_DECL(0) SP_DECL(1) SP_DECL(2) SP_DECL(3) SP_DECL(4) SP_DECL(5) SP_DECL(6) SP_DECL(7) SP_DECL(8) SP_DECL(9) SP_DECL(10) SP_DECL(11) SP_DECL(12) SP_DECL(13) SP_DECL(14) SP_DECL(15) SP_DECL(16) SP_DECL(17) SP_DECL(18) SP_DECL(19) SP_DECL(20) SP_DECL(21) SP_DECL(22)
static_assert(kStaticPrograms == 23, "list the generated programs above and below");

struct StaticEntry {
  const int32_t *src;
  const int32_t *row_src;   // staged row -> input dimension: part of the program's identity
  int n_src, rows, groups;
  hipError_t (*launch)(const IssArgs &, hipStream_t);
};
#define SP_ENTRY(i) {StaticProg##i::src, StaticProg##i::row_src, StaticProg##i::n_src, StaticProg##i::rows, StaticProg##i::groups, walk_static_launch_##i}
static const StaticEntry kStaticTable[kStaticPrograms] = {
    SP_ENTRY(0), SP_ENTRY(1), SP_ENTRY(2), SP_ENTRY(3), SP_ENTRY(4), SP_ENTRY(5), SP_ENTRY(6), SP_ENTRY(7), SP_ENTRY(8), SP_ENTRY(9), SP_ENTRY(10), SP_ENTRY(11), SP_ENTRY(12), SP_ENTRY(13), SP_ENTRY(14), SP_ENTRY(15), SP_ENTRY(16), SP_ENTRY(17), SP_ENTRY(18), SP_ENTRY(19), SP_ENTRY(20), SP_ENTRY(21), SP_ENTRY(22)};

// 1 + index of the static program for `groups` groups per series whose interpreter records
// (one group) equal `recs` AND whose staged rows come from the same input dimensions (the
// records name LDS rows; `[4]` alone has the records of `[1]` alone), or 0
int static_program_for(const NodeRec *recs, int n, int groups, const int32_t *row_src, int rows) {
  for (int i = 0; i < kStaticPrograms; ++i)
    if (groups == kStaticTable[i].groups && n == kStaticTable[i].n_src &&
        rows == kStaticTable[i].rows &&
        std::memcmp(row_src, kStaticTable[i].row_src, (size_t)rows * 4) == 0 &&
        std::memcmp(recs, kStaticTable[i].src, (size_t)n * 64) == 0)
      return i + 1;
  return 0;
}

// materialising, one aligned 1024-element chunk, unweighted Reals
hipError_t walk_static_launch(const IssArgs &a, hipStream_t st) {
  if (a.static_prog < 1 || a.static_prog > kStaticPrograms ||
      kStaticTable[a.static_prog - 1].groups != a.G)
    return hipErrorInvalidValue;
  return kStaticTable[a.static_prog - 1].launch(a, st);
}
#else
hipError_t SP_CAT(walk_static_launch_, STATIC_PROG)(const IssArgs &a, hipStream_t st) {
  return launch_walk_static<WalkCfg<2, 2, 2, 0, true, false, 4, 0, 0>,
                            SP_CAT(StaticProg, STATIC_PROG)>(a, st);
}
#endif

}  // namespace fr

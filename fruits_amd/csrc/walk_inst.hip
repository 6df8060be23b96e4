// One translation unit per (WALK_MODE, WALK_LV) [or WALK_TEAM1]: the walk kernel has
// many template variants; compiling them in separate objects lets the build run in
// parallel (fruits_amd/build.py).
#include "walk.h"

namespace fr {

#if WALK_MODE == 1 || WALK_MODE == 2
// The fused walk (walk_fused.h; MODE 2: the same walk writing the tensor): 4 consecutive elements
// per lane for 1024-element chunks, 2 for 512-element ones; carries of a multi-chunk walk always
// live in LDS (the host sizes the groups for that).
template <int E, int MULTI>
static hipError_t inst_f(const IssArgs &a, hipStream_t st) {
  if (a.semiring == kSemiArctic)
    return a.aux ? launch_fused_cfg<E, WALK_LV, MULTI, true, 1, false, WALK_MODE>(a, st)
                 : launch_fused_cfg<E, WALK_LV, MULTI, false, 1, false, WALK_MODE>(a, st);
  if (a.semiring == kSemiBayesian)
    return a.aux ? launch_fused_cfg<E, WALK_LV, MULTI, true, 2, false, WALK_MODE>(a, st)
                 : launch_fused_cfg<E, WALK_LV, MULTI, false, 2, false, WALK_MODE>(a, st);
  return a.aux ? launch_fused_cfg<E, WALK_LV, MULTI, true, 0, false, WALK_MODE>(a, st)
               : launch_fused_cfg<E, WALK_LV, MULTI, false, 0, false, WALK_MODE>(a, st);
}
template <int P>
static hipError_t inst_p(const IssArgs &a, hipStream_t st) {
  constexpr int E = P == 2 ? 4 : 2;
  if (a.nchunks > 1 && !a.carry_in_lds) return hipErrorInvalidValue;
  return a.nchunks > 1 ? inst_f<E, 1>(a, st) : inst_f<E, 0>(a, st);
}
#else
template <int P, int MULTI, bool VEC>
static hipError_t inst_w(const IssArgs &a, hipStream_t st) {
  constexpr int E = 2, PP = P;
  if (a.semiring == kSemiArctic)
    return a.aux ? launch_walk_cfg<E, PP, WALK_LV, MULTI, VEC, true, 4, WALK_MODE, 1>(a, st)
                 : launch_walk_cfg<E, PP, WALK_LV, MULTI, VEC, false, 4, WALK_MODE, 1>(a, st);
  if (a.semiring == kSemiBayesian)
    return a.aux ? launch_walk_cfg<E, PP, WALK_LV, MULTI, VEC, true, 4, WALK_MODE, 2>(a, st)
                 : launch_walk_cfg<E, PP, WALK_LV, MULTI, VEC, false, 4, WALK_MODE, 2>(a, st);
  // unweighted Reals, one aligned chunk, one group per series on a cache-sized batch: the
  // input is staged with non-temporal loads (capi.cpp sets nt_input)
  if constexpr (MULTI == 0 && VEC) {
    if (a.nt_input && !a.aux)
      return launch_walk_cfg<E, PP, WALK_LV, MULTI, VEC, false, 4, WALK_MODE, 0, true>(a, st);
  }
  return a.aux ? launch_walk_cfg<E, PP, WALK_LV, MULTI, VEC, true, 4, WALK_MODE>(a, st)
               : launch_walk_cfg<E, PP, WALK_LV, MULTI, VEC, false, 4, WALK_MODE>(a, st);
}
template <int P>
static hipError_t inst_p(const IssArgs &a, hipStream_t st) {
  // carries of a multi-chunk walk live in LDS when the group's slots fit (see WalkCfg)
  const int multi = a.nchunks > 1 ? (a.carry_in_lds ? 1 : 2) : 0;
  if (multi == 1) return a.vec_ok ? inst_w<P, 1, true>(a, st) : inst_w<P, 1, false>(a, st);
  if (multi == 2) return a.vec_ok ? inst_w<P, 2, true>(a, st) : inst_w<P, 2, false>(a, st);
  return a.vec_ok ? inst_w<P, 0, true>(a, st) : inst_w<P, 0, false>(a, st);
}
#endif
#define WALK_CAT2(a, b, c, d) a##b##c##d
#define WALK_CAT(a, b, c, d) WALK_CAT2(a, b, c, d)
#ifdef WALK_HO
// Translation unit of the HIGHORD instantiations (WalkCfg::HIGHORD: differencing orders >= 3 on
// series of several time chunks), apart from the common fused kernels like the TOTALINC ones.
template <int E>
static hipError_t inst_ho(const IssArgs &a, hipStream_t st) {
  if (a.aux && a.total_inc) {   // ... on a totally weighted plan whose sieves difference: both variants
    if (a.semiring == kSemiArctic) return launch_fused_cfg<E, WALK_LV, 1, true, 1, true, 1, true>(a, st);
    if (a.semiring == kSemiBayesian) return launch_fused_cfg<E, WALK_LV, 1, true, 2, true, 1, true>(a, st);
    return launch_fused_cfg<E, WALK_LV, 1, true, 0, true, 1, true>(a, st);
  }
  if (a.semiring == kSemiArctic)
    return a.aux ? launch_fused_cfg<E, WALK_LV, 1, true, 1, false, 1, true>(a, st)
                 : launch_fused_cfg<E, WALK_LV, 1, false, 1, false, 1, true>(a, st);
  if (a.semiring == kSemiBayesian)
    return a.aux ? launch_fused_cfg<E, WALK_LV, 1, true, 2, false, 1, true>(a, st)
                 : launch_fused_cfg<E, WALK_LV, 1, false, 2, false, 1, true>(a, st);
  return a.aux ? launch_fused_cfg<E, WALK_LV, 1, true, 0, false, 1, true>(a, st)
               : launch_fused_cfg<E, WALK_LV, 1, false, 0, false, 1, true>(a, st);
}
hipError_t WALK_CAT(walk_inst_ho, , _l, WALK_LV)(const IssArgs &a, int chunk, hipStream_t st) {
  if (a.nchunks < 2 || !a.carry_in_lds) return hipErrorInvalidValue;
  return chunk == 512 ? inst_ho<2>(a, st) : inst_ho<4>(a, st);
}
#elif defined(WALK_TI)
// Translation unit of the TOTALINC instantiations (WalkCfg::TOTALINC: fused epilogue of a
// totally weighted plan with differencing sieves), apart from the common fused kernels so the
// build stays parallel.
template <int E, int MULTI>
static hipError_t inst_ti(const IssArgs &a, hipStream_t st) {
  if (a.semiring == kSemiArctic) return launch_fused_cfg<E, WALK_LV, MULTI, true, 1, true>(a, st);
  if (a.semiring == kSemiBayesian) return launch_fused_cfg<E, WALK_LV, MULTI, true, 2, true>(a, st);
  return launch_fused_cfg<E, WALK_LV, MULTI, true, 0, true>(a, st);
}
template <int P>
static hipError_t inst_ti_p(const IssArgs &a, hipStream_t st) {
  constexpr int E = P == 2 ? 4 : 2;
  if (a.nchunks > 1 && !a.carry_in_lds) return hipErrorInvalidValue;
  return a.nchunks > 1 ? inst_ti<E, 1>(a, st) : inst_ti<E, 0>(a, st);
}
hipError_t WALK_CAT(walk_inst_ti, , _l, WALK_LV)(const IssArgs &a, int chunk, hipStream_t st) {
  return chunk == 512 ? inst_ti_p<1>(a, st) : inst_ti_p<2>(a, st);
}
#else
#if WALK_MODE == 1
hipError_t WALK_CAT(walk_inst_ti, , _l, WALK_LV)(const IssArgs &, int, hipStream_t);
hipError_t WALK_CAT(walk_inst_ho, , _l, WALK_LV)(const IssArgs &, int, hipStream_t);
#endif
hipError_t WALK_CAT(walk_inst_m, WALK_MODE, _l, WALK_LV)(const IssArgs &a, int chunk,
                                                          hipStream_t st) {
#if WALK_MODE == 1
  if (a.high_order && a.nchunks > 1) return WALK_CAT(walk_inst_ho, , _l, WALK_LV)(a, chunk, st);
  if (a.aux && a.total_inc) return WALK_CAT(walk_inst_ti, , _l, WALK_LV)(a, chunk, st);
#endif
  return chunk == 512 ? inst_p<1>(a, st) : inst_p<2>(a, st);
}
#endif

}  // namespace fr

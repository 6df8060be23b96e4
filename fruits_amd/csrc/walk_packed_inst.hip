// Instances of the wave-per-series kernel (walk_packed.h), one translation unit per
// WALK_MODE: chunk 128 / 256 (<= 4 / <= 8 register levels) and 384 (<= 4 levels), weighted or
// not, aligned or not, the three semirings.
#include "walk_packed.h"

namespace fr {

template <int P, int LV, int SEMI>
static hipError_t packed_pls(const IssArgs &a, hipStream_t st) {
#if WALK_MODE == 1
  return a.aux ? launch_walk_packed_cfg<P, LV, true, true, 1, SEMI>(a, st)
               : launch_walk_packed_cfg<P, LV, true, false, 1, SEMI>(a, st);
#else
  if (a.vec_ok)
    return a.aux ? launch_walk_packed_cfg<P, LV, true, true, 0, SEMI>(a, st)
                 : launch_walk_packed_cfg<P, LV, true, false, 0, SEMI>(a, st);
  return a.aux ? launch_walk_packed_cfg<P, LV, false, true, 0, SEMI>(a, st)
               : launch_walk_packed_cfg<P, LV, false, false, 0, SEMI>(a, st);
#endif
}

template <int P, int LV>
static hipError_t packed_pl(const IssArgs &a, hipStream_t st) {
  if (a.semiring == kSemiArctic) return packed_pls<P, LV, 1>(a, st);
  if (a.semiring == kSemiBayesian) return packed_pls<P, LV, 2>(a, st);
  return packed_pls<P, LV, 0>(a, st);
}

#define PACK_CAT2(a, b) a##b
#define PACK_CAT(a, b) PACK_CAT2(a, b)
hipError_t PACK_CAT(walk_packed_inst_m, WALK_MODE)(const IssArgs &a, int levels, hipStream_t st) {
  if (a.T <= 128) return levels <= 4 ? packed_pl<1, 4>(a, st) : packed_pl<1, 8>(a, st);
  if (a.T <= 256) return levels <= 4 ? packed_pl<2, 4>(a, st) : packed_pl<2, 8>(a, st);
  if (levels > 4 || a.T > 384) return hipErrorInvalidValue;
  return packed_pl<3, 4>(a, st);
}

}  // namespace fr

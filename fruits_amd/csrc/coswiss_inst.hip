// One translation unit per cosine exponent (COS_S): the variants of the CosWISS
// kernel for that exponent (chunk size, multi-chunk carries, aligned stores, fused).
#include "coswiss.h"

namespace fr {

template <int P, int MULTI>
static hipError_t cos_inst_pm(const IssArgs &a, hipStream_t st) {
  if (a.feats) return launch_coswiss_cfg<P, MULTI, true, 1, COS_S>(a, st);
  return a.vec_ok ? launch_coswiss_cfg<P, MULTI, true, 0, COS_S>(a, st)
                  : launch_coswiss_cfg<P, MULTI, false, 0, COS_S>(a, st);
}

template <int P>
static hipError_t cos_inst_p(const IssArgs &a, hipStream_t st) {
  return a.nchunks > 1 ? cos_inst_pm<P, 1>(a, st) : cos_inst_pm<P, 0>(a, st);
}

template <int P>
static hipError_t cos_packed_p(const IssArgs &a, hipStream_t st) {
  if (a.feats) return launch_coswiss_packed_cfg<P, true, 1, COS_S>(a, st);
  return a.vec_ok ? launch_coswiss_packed_cfg<P, true, 0, COS_S>(a, st)
                  : launch_coswiss_packed_cfg<P, false, 0, COS_S>(a, st);
}

#define COS_CAT2(a, b) a##b
#define COS_CAT(a, b) COS_CAT2(a, b)
hipError_t COS_CAT(coswiss_inst_s, COS_S)(const IssArgs &a, int chunk, hipStream_t st) {
  if (a.packed) {  // short series: one wave per (series, word, frequency) unit
    if (a.T <= 128) return cos_packed_p<1>(a, st);
    return a.T <= 256 ? cos_packed_p<2>(a, st) : cos_packed_p<3>(a, st);
  }
  return chunk == 512 ? cos_inst_p<1>(a, st) : cos_inst_p<2>(a, st);
}

}  // namespace fr

// Short series (T <= 256, T <= 384 for shallow tries): four series per workgroup, one wave each.
//
// The cooperative kernel (walk.h) spreads ONE series over the 256 threads of a
// workgroup in chunks of >= 512 elements: with T = 256 half of the lanes hold
// padding, with T = 128 three quarters.  Here every wave walks a whole (series,
// group) unit alone - TEAM = 1 of WalkCfg: the scan is wave-local (DPP only, no
// barrier, no LDS exchange) and a lane stages exactly the row elements it reads
// back, so the four waves of a workgroup never synchronise.  Same records, same
// walk / emit / fused-sieve code as the cooperative kernel.
#pragma once
#include "walk.h"

namespace fr {

template <class C>
__global__ __launch_bounds__(kWalkThreads) void iss_walk_packed_kernel(const IssArgs a) {
  static_assert(C::TEAM == 1 && C::MULTI == 0 && C::E == 2, "wave-per-series configuration");
  extern __shared__ double lds[];
  constexpr int EP = C::EP;
  const int tid = threadIdx.x;
  WalkCtx cx;
  cx.a = &a;
  cx.tid = tid;
  cx.lane = tid & 63;
  cx.wave = 0;
  cx.team = __builtin_amdgcn_readfirstlane(tid >> 6);
  double *rows_w = lds + (int64_t)cx.team * a.R * C::CHUNK;
  cx.rows = rows_w;
  cx.tot = nullptr;   // NW == 1: no cross-wave exchange
  cx.tail = nullptr;
  cx.tail_buf = 0;
  cx.buf = 0;
  cx.carry = nullptr;
  cx.t0 = 0;
  cx.first_chunk = true;
  cx.full_chunk = C::CHUNK <= a.T;
  cx.slot = 0;
  const int64_t units = a.N * a.G;
  for (int64_t u = (int64_t)blockIdx.x * C::TEAMS + cx.team; u < units;
       u += (int64_t)gridDim.x * C::TEAMS) {
    const int64_t n = u / a.G;
    const int g = (int)(u % a.G);
    const int node_begin = as_const(a.group_begin)[g];
    cx.pc_begin = node_begin;
    cx.out_base = a.out + n * a.out_n_stride;
    if constexpr (C::MODE == 1) {
      cx.feat_row = a.feats + n * a.feat_stride;
      cx.cnt_row = a.cnt + n * a.feat_stride;
      cx.cut_row = a.series_cuts ? a.series_cuts + n * a.cut_slots : nullptr;
      cx.series = n;
    }
    // stage: lane l keeps elements [h*128 + 2l, +2) of every row - the ones read_row
    // hands back to it (wave-local, no barrier)
    for (int r = 0; r < a.R; ++r) {
      const int src = as_const(a.row_src)[r];
      if (C::MODE == 1 && a.prep != nullptr && src >= 0) {
        // fused preparation (IssArgs::prep): the prepared row formed from the RAW input on the
        // way - INC / NEW(INC) / STD exactly as walk_fused.h's staging forms them
        const int raw = as_const(a.prep)[4 * src], lag = as_const(a.prep)[4 * src + 1];
        const bool standardise = as_const(a.prep)[4 * src + 2] != 0;
        const double *xp = a.X + (n * a.D + raw) * a.T;
        double mean = 0.0, den = 1.0;
        if (standardise) {
          mean = as_const(a.stats)[(n * a.n_prep + src) * 2];
          den = as_const(a.stats)[(n * a.n_prep + src) * 2 + 1];
        }
#pragma unroll
        for (int h = 0; h < C::P; ++h) {
          const int i = h * C::PIECE + cx.lane * 2;
          double e[2] = {0.0, 0.0};
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int64_t t = i + q;
            if (t < a.T) {
              double x = xp[t];
              if (lag > 0) x = t >= lag ? x - xp[t - lag] : 0.0;
              if (standardise) x = (x - mean) / den;
              e[q] = x;
            }
          }
          *reinterpret_cast<vd2 *>(rows_w + r * C::CHUNK + i) = vd2{e[0], e[1]};
        }
        continue;
      }
      const double *gp = src >= 0
                             ? a.X + (n * a.D + src) * a.T
                             : a.aux + (int64_t)(-src - 1) * a.aux_tab_stride + n * a.aux_n_stride;
#pragma unroll
      for (int h = 0; h < C::P; ++h) {
        const int i = h * C::PIECE + cx.lane * 2;
        vd2 v = {0.0, 0.0};
        if (a.vec_ok) {
          if (cx.full_chunk || i < a.T) v = *reinterpret_cast<const vd2 *>(gp + i);
        } else {
          if (i < a.T) v.x = gp[i];
          if (i + 1 < a.T) v.y = gp[i + 1];
        }
        *reinterpret_cast<vd2 *>(rows_w + r * C::CHUNK + i) = v;
      }
    }
    double ones[EP];
#pragma unroll
    for (int i = 0; i < EP; ++i) ones[i] = C::SEMI != 1 ? 1.0 : 0.0;
    int pc = node_begin;
    Rec cur = load_rec(a.recs, pc);
    walk<C, 0>(cx, cur, pc, ones);
  }
}

template <int P, int LV, bool VEC, bool W, int MODE, int SEMI>
static hipError_t launch_walk_packed_cfg(const IssArgs &a, hipStream_t st) {
  using C = WalkCfg<2, P, LV, 0, VEC, W, 1, MODE, SEMI>;
  const size_t lds = (size_t)C::TEAMS * a.R * C::CHUNK * sizeof(double);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  const int64_t units = a.N * a.G;
  int64_t blocks = (units + C::TEAMS - 1) / C::TEAMS;
  static LaunchCache cache;  // per instantiation; per-device entries, thread-safe
  int per_cu = 1;
  hipError_t e = cache.facts(iss_walk_packed_kernel<C>, kWalkThreads, lds, &per_cu);
  if (e != hipSuccess) return e;
  const int64_t resident = (int64_t)per_cu * device_cu_count();
  if (a.persistent && blocks > resident) blocks = resident;
  if (blocks < 1) return hipSuccess;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(iss_walk_packed_kernel<C>, dim3((unsigned)blocks), dim3(kWalkThreads), lds,
                     st, a);
  return hipGetLastError();
}

}  // namespace fr

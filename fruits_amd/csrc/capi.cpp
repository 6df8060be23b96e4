// C ABI of libfruits_hip.so (see include/fruits_hip.h for the contract and the
// reference interfaces each entry point replaces).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <set>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/fruits_hip.h"
#include "jit.h"
#include "kernels.h"
#include "launch_cache.h"
#include "plan.h"

struct fr_plan {
  fr::Plan *p;
};

namespace {

thread_local std::string g_err;

// Scratch of the fit's selections (fr_select_ranks_begin / _end), per device: a few blobs - device
// memory (jobs, histograms, candidate lists, results) and page-locked host memory (the job table
// on its way up, the results on their way down) - that selections in flight own and later ones
// reuse; grow-only, freed by fr_release_scratch
struct Scratch {
  void *ptr = nullptr, *host = nullptr;
  size_t bytes = 0, host_bytes = 0;
  bool busy = false;
};
constexpr int kScratchDevices = 64, kScratchBlobs = 8;
std::mutex g_scratch_mu[kScratchDevices];   // one per device: fits on different devices do not queue
Scratch g_scratch[kScratchDevices][kScratchBlobs];

thread_local int g_last_code = 0;   // of the entry points that return a handle (fr_select_ranks_begin)

int fail(int code, const std::string &msg) {
  g_err = msg;
  g_last_code = code;
  return code;
}

int hip_fail(hipError_t e, const char *what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  g_last_code = FR_E_HIP;
  return FR_E_HIP;
}

#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return hip_fail(e_, #expr);   \
  } while (0)

int env_int(const char *name, int dflt) {
  const char *v = std::getenv(name);
  return v && *v ? std::atoi(v) : dflt;
}

// Developer knobs live in ONE variable: FRUITS_HIP_DEBUG="name=value,name=value" with
//   groups=G    groups of root sub-tries per series instead of the host's choice
//   persist=P   1 / 0: persistent grid / one workgroup per unit for the materialising walk
//   packed=0    cooperative kernels also for short series (the wave-per-series ones are default)
//   stamps=M    the diagnostic timing build's mask (IssArgs::debug), dbg_bytes=B its stamp buffer
// Nothing here changes a result; the product reads none of them in normal operation.
int debug_knob(const char *name, int dflt) {
  const char *v = std::getenv("FRUITS_HIP_DEBUG");
  if (!v || !*v) return dflt;
  const size_t n = std::strlen(name);
  for (const char *p = v; *p;) {
    if (std::strncmp(p, name, n) == 0 && p[n] == '=') return std::atoi(p + n + 1);
    while (*p && *p != ',') ++p;
    if (*p == ',') ++p;
  }
  return dflt;
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// A stream that is being captured into a hipGraph must not see allocations or
// synchronous copies: the one-time uploads below refuse to run then (the caller
// prepares the plan first: fr_plan_prepare / fr_pipeline_prepare).
bool stream_is_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return cs != hipStreamCaptureStatusNone;
}

int current_device_id() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
  return dev;
}

// The tables of a plan live on the device that was current at their first upload.
int claim_device(fr::Plan &p, const char *who) {
  const int dev = current_device_id();
  if (p.device < 0) p.device = dev;
  if (p.device != dev)
    return fail(FR_E_ARG, std::string(who) + ": the plan's tables live on device " +
                              std::to_string(p.device) + " but device " + std::to_string(dev) +
                              " is current (plans are per device)");
  return FR_OK;
}

// Uploads the node order for G groups once per plan.  Allocates and copies
// synchronously: done by fr_plan_prepare, or by the first run outside a capture.
// Caller holds p.mu.
int ensure_device_program(fr::Plan &p, fr::GroupedProgram &gp, hipStream_t st, const char *who) {
  int rc = claim_device(p, who);
  if (rc != FR_OK) return rc;
  if (gp.d_blob) return FR_OK;
  if (stream_is_capturing(st))
    return fail(FR_E_ARG, std::string(who) + ": the stream is being captured and the plan's "
                          "tables for this shape are not on the device yet - call "
                          "fr_plan_prepare / fr_pipeline_prepare before the capture");
  const size_t n_recs = gp.recs.size();
  size_t off = 0;
  const size_t o_nodes = off;       off = align_up(off + n_recs * sizeof(fr::NodeRec), 64);
  const size_t o_gb = off;          off = align_up(off + gp.group_begin.size() * 4, 64);
  const size_t o_fac = off;         off = align_up(off + p.factors.size() * 4, 64);
  const size_t o_emit = off;        off = align_up(off + p.emit_rows.size() * 4, 64);
  const size_t o_rows = off;        off = align_up(off + p.row_src.size() * 4, 64);
  const size_t o_alpha = off;       off = align_up(off + p.alphas.size() * 4, 64);
  const size_t o_srows = off;       off = align_up(off + gp.slot_rows.size() * 4, 64);
  const size_t o_grb = off;         off = align_up(off + gp.group_row_begin.size() * 4, 64);
  const size_t o_shape = off;       off = align_up(off + gp.shape_ids.size() * 4, 64);
  std::vector<char> host(off + 64, 0);
  std::memcpy(host.data() + o_srows, gp.slot_rows.data(), gp.slot_rows.size() * 4);
  std::memcpy(host.data() + o_grb, gp.group_row_begin.data(), gp.group_row_begin.size() * 4);
  std::memcpy(host.data() + o_shape, gp.shape_ids.data(), gp.shape_ids.size() * 4);
  std::memcpy(host.data() + o_nodes, gp.recs.data(), n_recs * sizeof(fr::NodeRec));
  std::memcpy(host.data() + o_gb, gp.group_begin.data(), gp.group_begin.size() * 4);
  std::memcpy(host.data() + o_fac, p.factors.data(), p.factors.size() * 4);
  std::memcpy(host.data() + o_emit, p.emit_rows.data(), p.emit_rows.size() * 4);
  std::memcpy(host.data() + o_rows, p.row_src.data(), p.row_src.size() * 4);
  std::memcpy(host.data() + o_alpha, p.alphas.data(), p.alphas.size() * 4);
  void *d = nullptr;
  HIP_TRY(hipMalloc(&d, host.size()));
  hipError_t e = hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(d);
    return hip_fail(e, "hipMemcpy(program)");
  }
  char *b = static_cast<char *>(d);
  gp.d_blob = d;
  gp.d_recs = reinterpret_cast<const fr::NodeRec *>(b + o_nodes);
  gp.d_group_begin = reinterpret_cast<const int32_t *>(b + o_gb);
  gp.d_factors = reinterpret_cast<const int32_t *>(b + o_fac);
  gp.d_emit_rows = reinterpret_cast<const int32_t *>(b + o_emit);
  gp.d_row_src = reinterpret_cast<const int32_t *>(b + o_rows);
  gp.d_alphas = reinterpret_cast<const float *>(b + o_alpha);
  gp.d_slot_rows = reinterpret_cast<const int32_t *>(b + o_srows);
  gp.d_group_row_begin = reinterpret_cast<const int32_t *>(b + o_grb);
  gp.d_shape_ids = reinterpret_cast<const int32_t *>(b + o_shape);
  return FR_OK;
}

// Caller holds p.mu.
int ensure_cos_program(fr::Plan &p, fr::CosProgram &c, hipStream_t st, const char *who) {
  int rc = claim_device(p, who);
  if (rc != FR_OK) return rc;
  if (c.d_blob) return FR_OK;
  if (stream_is_capturing(st))
    return fail(FR_E_ARG, std::string(who) + ": the stream is being captured and the CosWISS "
                          "program is not on the device yet - call fr_plan_prepare / "
                          "fr_pipeline_prepare before the capture");
  size_t off = 0;
  const size_t o_lb = off;    off = align_up(off + c.letter_begin.size() * 4, 64);
  const size_t o_fb = off;    off = align_up(off + c.fac_begin.size() * 4, 64);
  const size_t o_fac = off;   off = align_up(off + c.factors.size() * 4, 64);
  const size_t o_fr = off;    off = align_up(off + c.freqs.size() * 4, 64);
  std::vector<char> host(off + 64, 0);
  std::memcpy(host.data() + o_lb, c.letter_begin.data(), c.letter_begin.size() * 4);
  std::memcpy(host.data() + o_fb, c.fac_begin.data(), c.fac_begin.size() * 4);
  std::memcpy(host.data() + o_fac, c.factors.data(), c.factors.size() * 4);
  std::memcpy(host.data() + o_fr, c.freqs.data(), c.freqs.size() * 4);
  void *d = nullptr;
  HIP_TRY(hipMalloc(&d, host.size()));
  hipError_t e = hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(d);
    return hip_fail(e, "hipMemcpy(coswiss program)");
  }
  char *b = static_cast<char *>(d);
  c.d_blob = d;
  c.d_letter_begin = reinterpret_cast<const int32_t *>(b + o_lb);
  c.d_fac_begin = reinterpret_cast<const int32_t *>(b + o_fb);
  c.d_factors = reinterpret_cast<const int32_t *>(b + o_fac);
  c.d_freqs = reinterpret_cast<const float *>(b + o_fr);
  return FR_OK;
}

int choose_groups(const fr::Plan &p, int64_t N, int requested) {
  const int U = p.units();
  if (U <= 1) return 1;
  int G = requested;
  if (G <= 0) G = debug_knob("groups", 0);
  if (G <= 0) {
    // aim for a few thousand workgroups (256 CUs x several resident each)
    const int64_t target = 2048;
    G = (int)((target + N - 1) / (N > 0 ? N : 1));
  }
  if (G > U) G = U;
  if (G < 1) G = 1;
  return G;
}

// How a (plan, N, T, groups) launch is shaped: decided identically by the run path
// and by fr_plan_prepare (which uploads the node order the run will ask for).
struct LaunchShape {
  bool packed = false;    // wave-per-series kernel (short series)
  bool fits = true;       // the staged rows of one time chunk fit the LDS
  int G = 1;              // groups of root sub-tries per series
};

bool staged_rows_fit(const fr::Plan &p, int64_t T) {
  return (size_t)p.rows_staged() * fr::walk_chunk_elems(T) * 8 <= 150 * 1024;
}

LaunchShape launch_shape(const fr::Plan &p, int64_t N, int64_t T, int requested_groups) {
  LaunchShape s;
  s.fits = staged_rows_fit(p, T);
  // short series: four series per workgroup, one wave each (their rows side by side in LDS)
  const int64_t packed_chunk = T <= 128 ? 128 : (T <= 256 ? 256 : 384);
  s.packed = debug_knob("packed", 1) != 0 &&
             fr::packed_supported(T, p.levels, p.semiring) &&
             (size_t)4 * p.rows_staged() * packed_chunk * 8 <= 64 * 1024;
  // (a packed workgroup holds four units: ask for four times the units)
  s.G = choose_groups(p, s.packed ? (N + 3) / 4 : N, requested_groups);
  return s;
}

// LDS carry slots of a multi-chunk walk: 3 per record of the program (nodes + one sentinel
// per group); sized for up to kSpanGroupsMax groups so that the kernel's LDS footprint -
// and with it the number of resident workgroups the group choice is made for - does not
// depend on the choice itself
constexpr int kSpanGroupsMax = 12;
int carry_slots_for(const fr::Plan &p, int G) {
  return 3 * ((int)p.nodes.size() + std::max(G, kSpanGroupsMax));
}
bool carries_fit_lds(const fr::Plan &p, int64_t T, int G) {
  // rows + carries must leave room for >= 4 workgroups per CU (160 KiB LDS)
  const size_t rows_bytes = (size_t)p.rows_staged() * fr::walk_chunk_elems(T) * 8;
  return rows_bytes + (size_t)carry_slots_for(p, G) * 8 <= 40 * 1024;
}

// LDS feature window of a fused cooperative launch (walk_device.h, feat_flush): as many slots
// as fit next to the rows and carries while four workgroups still share a CU's 160 KiB, at
// least what the widest node needs (output rows x feature ops).  `fits`: every group's
// features fit, so a unit flushes once.  0: the widest node does not fit the LDS at all.
int feat_window_sized(int widest, int largest_group, size_t other_lds_bytes, bool mpi, bool &fits);
int feat_window_for(const fr::GroupedProgram &gp, size_t other_lds_bytes, int n_ops, bool mpi,
                    bool &fits) {
  int widest = 0, largest_group = 0;
  for (int g = 0; g < gp.groups; ++g) {
    int total = 0;
    for (int i = gp.group_begin[g]; i < gp.group_begin[g + 1]; ++i) {
      if ((gp.recs[i].w[0] & 0xff) == fr::kRecSentinelLevel) continue;
      const int need = gp.recs[i].w[6] * n_ops;
      widest = std::max(widest, need);
      total += need;
    }
    largest_group = std::max(largest_group, total);
  }
  return feat_window_sized(widest, largest_group, other_lds_bytes, mpi, fits);
}
// (widest: the slots one node needs; largest_group: the slots of the largest unit)
int feat_window_sized(int widest, int largest_group, size_t other_lds_bytes, bool mpi, bool &fits) {
  const size_t budget = 38 * 1024;
  int W = 64;
  while (W < 1024 && W < largest_group &&
         other_lds_bytes + fr::feat_window_bytes(2 * W, mpi, false) <= budget)
    W *= 2;
  if (W < widest) W = (widest + 1) / 2 * 2;
  if (other_lds_bytes + fr::feat_window_bytes(W, mpi, false) > 160 * 1024) return 0;
  fits = largest_group <= W;
  return W;
}

// Resident workgroups of the cooperative walk kernel instance that (plan, T, fused,
// vec_ok) selects: a dry run of the launcher (nothing is enqueued).
int64_t query_resident(const fr::Plan &p, int64_t N, int64_t T, bool fused, bool vec_ok) {
  if (debug_knob("persist", 1) == 0) return 0;
  fr::IssArgs a{};
  int32_t resident = 0;
  double *const dummy = reinterpret_cast<double *>(uintptr_t(256));  // never dereferenced
  a.N = N;
  a.D = std::max(1, p.max_dim);
  a.T = T;
  a.G = 1;
  a.R = p.rows_staged();
  a.total_nodes = (int32_t)p.nodes.size();
  a.aux = p.weighting != 0 ? dummy : nullptr;
  a.carry = T > fr::walk_chunk_elems(T) ? dummy : nullptr;
  a.vec_ok = vec_ok ? 1 : 0;
  a.persistent = 1;
  a.semiring = p.semiring;
  a.carry_slots = carry_slots_for(p, 1);
  a.carry_in_lds = (fused || carries_fit_lds(p, T, 1)) ? 1 : 0;   // (the fused walk: always)
  a.feats = fused ? dummy : nullptr;
  a.feat_window = fused ? 128 : 0;   // (the launch's own window may differ a little)
  a.resident_out = &resident;
  if (fr::launch_iss_walk(a, p.levels, nullptr) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return resident;
}

// Groups per series for the cooperative kernel (walk.h): a unit is (series, group of root
// sub-tries) and stages the series' rows itself, so groups only pay where finer units help.
// Measured on config 2 and its 48-word tiling (tools/gpu_sched.sh: FRUITS_HIP_GROUPS = 1, 2,
// 3, 6, 9 against N = 64 ... 8192, `resident` = one round of workgroups, 1536 for these
// kernels):
//   N < resident        the batch alone cannot fill the chip: ceil(resident / N) groups, at
//                       most 6 (N = 64: G = 6 7.5 us vs 15.7 with 1; 256: 3; 512: 3; 768 and
//                       1000: 2)
//   N < 2 x resident    whole series (N = 1536: 41.7 us, G = 3 46.9; N = 2048: 64.5, G = 3 73.2)
//   beyond              small plans (<= 32 nodes): 3 groups - finer units even out the last
//                       rounds and their restaging hits the XCD's L2 (N = 4096: 135.5 vs
//                       146.6 us, N = 8192: 272 vs 291 us); larger plans keep whole series
//                       (config 3 / 4 / 5: no difference measured)
int choose_groups_walk(const fr::Plan &p, int64_t N, int64_t T, int64_t resident, bool fused = false) {
  const int U = p.units();
  if (U <= 1 || N <= 0) return 1;
  if (resident <= 0) return choose_groups(p, N, 0);
  if (N < resident) {
    const int64_t G = std::min<int64_t>((resident + N - 1) / N, 6);
    return (int)std::max<int64_t>(1, std::min<int64_t>(G, U));
  }
  // fused launches run one short-lived workgroup per unit (run_walk): two groups per series
  // balance a little better than whole series on LONG plans (config 4, 1351 nodes: 13.08 vs
  // 13.32 ms); on shorter ones every extra unit is one more staging of the series' rows - the
  // word shards of config 4 over 8 ranks (~170 nodes each): 1.86 ms with whole series, 2.40 ms
  // with two groups (tools/bench_shards.py)
  // (round 3, the pipeline's own kernels: one-chunk plans of 668 / 683 nodes - two word shards of
  // config 4 - 4.74 / 4.84 ms with whole series, 5.47 / 5.48 ms with two groups; 1351 nodes: 9.53
  // vs 9.35 ms; config 5, 511 nodes over four time chunks - two groups also halve the LDS carries
  // of a unit: 17.8 vs 17.3 ms)
  if (fused)
    return p.nodes.size() >= (T > fr::walk_chunk_elems(T) ? 400u : 1000u) ? std::min(U, 2) : 1;
  // materialising launches of long plans run one short-lived workgroup per unit too (the lean
  // walk, run_walk): two groups per series shorten the last round (of_weight(4,2), N = 2048:
  // 390 -> 372 us; the same at N = 8192) for one more staging of the series' rows
  if (!p.letter_sum && p.nodes.size() >= 64 && debug_knob("lean", 1) != 0) return std::min(U, 2);
  if (N < 2 * resident || p.nodes.size() > 32) return 1;
  return std::min(U, 3);
}

// Run-time compiled static programs of a plan (jit.cpp), by groups per series.
struct JitState {
  std::map<int, fr::JitProgram> progs;
  bool tried = false;
  std::string error;     // why the plan has none (not an error of the caller's)
};

// batches below this many series run a static program with all its groups (to fill the chip)
constexpr int kStaticSplitBelow = 768;

// (T in (384, 512]: the 1024-element chunk with half of its lanes idle - still ahead of the
// interpreter's 512-element chunk on cache-sized batches, see run_walk)
bool static_shape_ok(const fr::Plan &p, int64_t T) {
  return !p.cos && p.weighting == 0 && p.semiring == fr::kSemiReals && T > debug_knob("static_min_T", 384) && T <= 1024;
}

// Compiles and loads the plan's static programs (one group and min(3, units) groups per
// series) unless an ahead-of-time program covers it.  Called with p.mu held; failures leave
// the plan on the interpreter.
void ensure_jit(fr::Plan &p) {
  if (p.jit == nullptr) p.jit = new JitState;
  JitState &js = *static_cast<JitState *>(p.jit);
  if (js.tried) return;
  js.tried = true;
  const int gmax = std::max(1, std::min(3, p.units()));
  for (int g : {1, gmax}) {
    if (js.progs.count(g)) continue;
    const fr::StaticSchedule sc = fr::static_schedule(p, g);
    if (!sc.ok || sc.groups != g) {
      js.error = "the plan does not qualify for a static program";
      return;
    }
    std::string code, err;
    fr::JitProgram prog;
    bool from_cache = false;
    bool ok = fr::jit_compile(sc, code, err, &from_cache) && fr::jit_load(code, sc, prog, err);
    if (!ok && from_cache) {
      // a cached code object the loader refuses (another ROCm, a damaged file): compile afresh
      fr::jit_cache_drop(sc);
      ok = fr::jit_compile(sc, code, err) && fr::jit_load(code, sc, prog, err);
    }
    if (!ok) {
      js.error = err;
      return;
    }
    js.progs[g] = prog;
  }
}

// One-time uploads for the node order a run of this (N, T, groups) asks for.
int prepare_plan(fr::Plan &p, int64_t N, int64_t T, int32_t groups, bool fused, const char *who) {
  std::lock_guard<std::mutex> lock(p.mu);
  if (p.cos) return ensure_cos_program(p, *p.cos, nullptr, who);
  if (N == 0 || T == 0 || p.K == 0 || p.nodes.empty()) return FR_OK;
  const LaunchShape shape = launch_shape(p, N, T, groups);
  if (!shape.fits)
    return fail(FR_E_LIMIT, std::string(who) + ": the plan stages " +
                                std::to_string(p.rows_staged()) +
                                " rows per time chunk, more than the LDS holds - split the word list");
  std::vector<int> Gs;
  const bool auto_groups = !shape.packed && groups <= 0 && debug_knob("groups", 0) <= 0;
  if (auto_groups) {
    // (the choice depends on the kernel instance - fused or not, 16-byte aligned or not -
    // which is only known when the pointers are: upload what either would ask for)
    Gs.push_back(choose_groups_walk(p, N, T, query_resident(p, N, T, fused, true), fused));
    if (!fused) Gs.push_back(choose_groups_walk(p, N, T, query_resident(p, N, T, false, false)));
  } else {
    Gs.push_back(shape.G);
  }
  for (int G : Gs) {
    int rc = ensure_device_program(p, fr::grouped(p, G), nullptr, who);
    if (rc != FR_OK) return rc;
  }
  return FR_OK;
}

struct WorkLayout {
  size_t aux_bytes = 0, carry_bytes = 0;
  size_t total() const { return aux_bytes + carry_bytes; }
};

WorkLayout work_layout(const fr::Plan &p, int64_t N, int64_t T, int64_t lookup_rows) {
  WorkLayout w;
  if (p.cos) {  // the (F, 2, T) sin / cos tables
    w.aux_bytes = align_up((size_t)p.cos->F * 2 * (size_t)T * 8, 256);
    return w;
  }
  if (p.weighting != 0)
    w.aux_bytes = align_up((size_t)p.aux_tables() * (size_t)lookup_rows * (size_t)T * 8, 256);
  if (T > fr::walk_chunk_elems(T))
    w.carry_bytes = align_up((size_t)N * 3 * p.nodes.size() * 8, 256);
  return w;
}

}  // namespace

extern "C" {

const char *fr_last_error(void) { return g_err.c_str(); }

int fr_version(void) { return 133; }

int fr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int fr_malloc(void **d_ptr, int64_t bytes) {
  if (!d_ptr || bytes < 0) return fail(FR_E_ARG, "fr_malloc: bad argument");
  *d_ptr = nullptr;
  if (bytes == 0) return FR_OK;
  hipError_t e = hipMalloc(d_ptr, (size_t)bytes);
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    return fail(FR_E_NOMEM, "fr_malloc: out of device memory");
  }
  if (e != hipSuccess) return hip_fail(e, "hipMalloc");
  return FR_OK;
}

int fr_free(void *d_ptr) {
  if (!d_ptr) return FR_OK;
  HIP_TRY(hipFree(d_ptr));
  return FR_OK;
}

int fr_memcpy_h2d(void *d_dst, const void *h_src, int64_t bytes, void *stream) {
  if (bytes == 0) return FR_OK;
  if (!d_dst || !h_src || bytes < 0) return fail(FR_E_ARG, "fr_memcpy_h2d: bad argument");
  HIP_TRY(hipMemcpyAsync(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return FR_OK;
}

int fr_memcpy_d2h(void *h_dst, const void *d_src, int64_t bytes, void *stream) {
  if (bytes == 0) return FR_OK;
  if (!h_dst || !d_src || bytes < 0) return fail(FR_E_ARG, "fr_memcpy_d2h: bad argument");
  HIP_TRY(hipMemcpyAsync(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return FR_OK;
}

int fr_stream_sync(void *stream) {
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return FR_OK;
}

fr_plan_t *fr_plan_create(int32_t W, const int32_t *exps, const int32_t *L, const int32_t *Dw,
                          const float *alpha, const int32_t *depth, int32_t weighting,
                          int32_t flags) {
  std::string err;
  fr::Plan *p = fr::build_plan(W, exps, L, Dw, alpha, depth, weighting, flags, err);
  if (!p) {
    g_err = err;
    return nullptr;
  }
  fr_plan_t *h = new fr_plan_t;
  h->p = p;
  return h;
}

fr_plan_t *fr_plan_create_coswiss(int32_t W, const int32_t *exps, const int32_t *L,
                                  const int32_t *Dw, int32_t n_freqs, const float *freqs,
                                  int32_t exponent, int32_t total_weighting) {
  std::string err;
  fr::Plan *p = fr::build_coswiss_plan(W, exps, L, Dw, n_freqs, freqs, exponent,
                                       total_weighting, err);
  if (!p) {
    g_err = err;
    return nullptr;
  }
  fr_plan_t *h = new fr_plan_t;
  h->p = p;
  return h;
}

void fr_plan_destroy(fr_plan_t *plan) {
  if (!plan) return;
  if (plan->p) {
    if (plan->p->cos) {
      if (plan->p->cos->d_mask) (void)hipFree(plan->p->cos->d_mask);
      if (plan->p->cos->d_blob) (void)hipFree(plan->p->cos->d_blob);
      delete plan->p->cos;
    }
    for (auto &kv : plan->p->programs)
      if (kv.second.d_blob) (void)hipFree(kv.second.d_blob);
    for (auto &kv : plan->p->pieced)
      for (fr::PieceType &t : kv.second.types)
        if (t.d_blob) (void)hipFree(t.d_blob);
    if (plan->p->jit) {
      JitState *js = static_cast<JitState *>(plan->p->jit);
      for (auto &kv : js->progs) fr::jit_unload(kv.second);
      delete js;
    }
    delete plan->p;
  }
  delete plan;
}

int64_t fr_plan_info(const fr_plan_t *plan, int32_t what) {
  if (!plan || !plan->p) return fail(FR_E_ARG, "fr_plan_info: null plan");
  const fr::Plan &p = *plan->p;
  switch (what) {
    case FR_INFO_ROWS: return p.K;
    case FR_INFO_NODES: return (int64_t)p.nodes.size();
    case FR_INFO_LEVELS: return p.levels;
    case FR_INFO_DIMS_USED: return p.dims_used;
    case FR_INFO_MAX_DIM: return p.max_dim;
    case FR_INFO_ALPHAS: return (int64_t)p.alphas.size();
    case FR_INFO_GROUPS: return p.units();
    case FR_INFO_SHARED: return p.shared ? 1 : 0;
    case FR_INFO_STAGED_ROWS: return p.cos ? 0 : p.rows_staged();
    case FR_INFO_AOT_PROGRAM: {
      fr::Plan &q = *plan->p;
      if (q.cos) return 0;
      std::lock_guard<std::mutex> lock(q.mu);
      const fr::GroupedProgram &g1 = fr::grouped(q, 1);
      return fr::static_program_for(g1.recs.data(), (int)g1.recs.size(), 1, q.row_src.data(),
                                    (int)q.row_src.size());
    }
    case FR_INFO_JIT_PROGRAMS:
      return p.jit ? (int64_t)static_cast<const JitState *>(p.jit)->progs.size() : 0;
    default: return fail(FR_E_ARG, "fr_plan_info: unknown selector");
  }
}

int32_t fr_plan_dump(const fr_plan_t *plan, int32_t *buf, int32_t cap) {
  if (!plan || !plan->p) return fail(FR_E_ARG, "fr_plan_dump: null plan");
  const fr::Plan &p = *plan->p;
  const int32_t n = (int32_t)p.nodes.size();
  for (int32_t i = 0; i < n && buf && (i + 1) * 8 <= cap; ++i) {
    const fr::NodeDesc &nd = p.nodes[i];
    int32_t *o = buf + (size_t)i * 8;
    o[0] = nd.level;
    o[1] = nd.flags;
    o[2] = nd.fac_count;
    o[3] = nd.emit_count;
    o[4] = nd.emit_count ? p.emit_rows[nd.emit_begin] : -1;
    o[5] = nd.emit_mul;
    o[6] = nd.z_mul;
    o[7] = p.unit_of[i];
  }
  return n;
}

int32_t fr_plan_records(fr_plan_t *plan, int32_t groups, int32_t *buf, int64_t cap_words) {
  if (!plan || !plan->p || plan->p->cos) return fail(FR_E_ARG, "fr_plan_records: not a trie plan");
  fr::Plan &p = *plan->p;
  std::lock_guard<std::mutex> lock(p.mu);
  const fr::GroupedProgram &gp = fr::grouped(p, groups);
  const int64_t n = (int64_t)gp.recs.size();
  if (buf != nullptr && cap_words >= n * 16)
    std::memcpy(buf, gp.recs.data(), (size_t)n * 64);
  return (int32_t)n;
}

int64_t fr_plan_pieces(fr_plan_t *plan, int32_t max_piece, int32_t *buf, int64_t cap_words) {
  if (!plan || !plan->p) return fail(FR_E_ARG, "fr_plan_pieces: null plan");
  fr::Plan &p = *plan->p;
  std::lock_guard<std::mutex> lock(p.mu);
  const fr::PiecedProgram &pp = fr::pieced(p, max_piece > 0 ? max_piece : fr::kFusedPieceNodes);
  if (!pp.ok) return 0;
  // header: types, K, node executions in chains, nodes of the plan; per type 8 words + its
  // records, items and unit tables; then the output row at every walk position
  std::vector<int32_t> out{(int32_t)pp.types.size(), p.K, pp.chain_nodes, (int32_t)p.nodes.size()};
  for (const fr::PieceType &t : pp.types) {
    out.insert(out.end(), {t.body_nodes, t.body_rows, t.levels, t.units(), (int32_t)t.items.size() / 4,
                           t.max_unit_nodes, (int32_t)t.recs.size(), t.max_unit_rows});
    for (const fr::NodeRec &r : t.recs) out.insert(out.end(), r.w, r.w + 16);
    out.insert(out.end(), t.items.begin(), t.items.end());
    out.insert(out.end(), t.unit_begin.begin(), t.unit_begin.end());
    out.insert(out.end(), t.unit_row0.begin(), t.unit_row0.end());
  }
  out.insert(out.end(), pp.row_of_walk.begin(), pp.row_of_walk.end());
  if (buf != nullptr && cap_words >= (int64_t)out.size()) std::memcpy(buf, out.data(), out.size() * 4);
  return (int64_t)out.size();
}

int32_t fr_plan_static_schedule(fr_plan_t *plan, int32_t groups, int32_t *buf, int64_t cap_words) {
  if (!plan || !plan->p) return fail(FR_E_ARG, "fr_plan_static_schedule: null plan");
  fr::Plan &p = *plan->p;
  std::lock_guard<std::mutex> lock(p.mu);
  const fr::StaticSchedule sc = fr::static_schedule(p, groups);
  if (!sc.ok) return 0;
  const int64_t n = (int64_t)sc.entries.size();
  // header (32 words): entries, rows, frames, groups, row sources [4..8), group_begin
  // [8..16), rows read by each group [16..24); then the entries
  const int64_t words = 32 + n * 16;
  if (sc.groups > 8) return 0;
  if (buf != nullptr && cap_words >= words) {
    std::memset(buf, 0, 128);
    buf[0] = (int32_t)n;
    buf[1] = sc.rows;
    buf[2] = sc.frames;
    buf[3] = sc.groups;
    for (int r = 0; r < sc.rows; ++r) buf[4 + r] = sc.row_src[r];
    for (int g = 0; g < sc.groups; ++g) {
      buf[8 + g] = sc.group_begin[g];
      buf[16 + g] = sc.group_rows[g];
    }
    std::memcpy(buf + 32, sc.entries.data(), (size_t)n * 64);
  }
  return (int32_t)n;
}

int64_t fr_plan_workspace_bytes(const fr_plan_t *plan, int64_t N, int64_t T, int64_t lookup_rows) {
  if (!plan || !plan->p || N < 0 || T < 0 || lookup_rows < 0)
    return fail(FR_E_ARG, "fr_plan_workspace_bytes: bad argument");
  return (int64_t)work_layout(*plan->p, N, T, lookup_rows).total();
}

int32_t fr_plan_fits(const fr_plan_t *plan, int64_t T) {
  if (!plan || !plan->p || T < 0) return fail(FR_E_ARG, "fr_plan_fits: bad argument");
  const fr::Plan &p = *plan->p;
  if (p.cos) return (p.cos->exponent <= fr::kCosMaxExponent && p.levels <= 16) ? 1 : 0;
  return staged_rows_fit(p, T) ? 1 : 0;
}

int fr_plan_prepare(fr_plan_t *plan, int64_t N, int64_t T, int32_t groups) {
  if (!plan || !plan->p || N < 0 || T < 0) return fail(FR_E_ARG, "fr_plan_prepare: bad argument");
  fr::Plan &p = *plan->p;
  int rc = prepare_plan(p, N, T, groups, false, "fr_plan_prepare");
  if (rc != FR_OK) return rc;
  // a small plan without an ahead-of-time static program gets one compiled now (hipRTC,
  // cached on disk); a failure is not the caller's: the interpreter runs the plan
  if (static_shape_ok(p, T) && env_int("FRUITS_HIP_JIT", 1) != 0 &&
      env_int("FRUITS_HIP_STATIC", 1) != 0) {
    std::lock_guard<std::mutex> lock(p.mu);
    if (p.static_prog[0] < 0) {
      const fr::GroupedProgram &g1 = fr::grouped(p, 1);
      for (int g = 1; g <= 3; ++g)
        p.static_prog[g] = fr::static_program_for(g1.recs.data(), (int)g1.recs.size(), g, p.row_src.data(),
                                                       (int)p.row_src.size());
      p.static_prog[0] = 0;
    }
    if (p.static_prog[1] <= 0 && fr::static_schedule(p, 1).ok) ensure_jit(p);
  }
  return FR_OK;
}

int32_t fr_plan_jit(fr_plan_t *plan, int32_t groups, int32_t compile_only, char *msg, int64_t msg_cap) {
  if (!plan || !plan->p) return fail(FR_E_ARG, "fr_plan_jit: null plan");
  fr::Plan &p = *plan->p;
  std::lock_guard<std::mutex> lock(p.mu);
  auto say = [&](const std::string &s) {
    if (msg && msg_cap > 0) {
      const size_t n = std::min((size_t)msg_cap - 1, s.size());
      std::memcpy(msg, s.data(), n);
      msg[n] = 0;
    }
  };
  say("");
  if (compile_only) {   // needs no GPU: the code object's size, 0 when the plan has no schedule
    const fr::StaticSchedule sc = fr::static_schedule(p, groups);
    if (!sc.ok) return 0;
    std::string code, err;
    if (!fr::jit_compile(sc, code, err)) {
      say(err);
      return fail(FR_E_LIMIT, "fr_plan_jit: " + err);
    }
    return (int32_t)code.size();
  }
  ensure_jit(p);
  JitState &js = *static_cast<JitState *>(p.jit);
  say(js.error);
  return (int32_t)js.progs.size();
}

}  // extern "C"

struct PipeSieve {
  int32_t kind, inc, Q1, col, q_off;
  bool series_cuts = false;    // cuts are slots of the per-series table (coquantile cuts)
  std::vector<int32_t> cuts;   // transformed, clamped to [0, T]
};

struct fr_pipeline {
  fr_plan_t *plan = nullptr;
  int64_t T = 0;
  int32_t per_sum = 0, q_stride = 0, n_ops = 0, n_ops_padded = 0;
  int32_t n_ops_eff = 0;           // ops per row after dropping NPI ops an MPI op covers
  std::vector<int32_t> npi_pairs;  // (npi column, mpi column) inside one iterated sum's block
  void *d_npi_pairs = nullptr;
  std::vector<PipeSieve> sieves;
  std::vector<int32_t> mpi_cols;   // columns inside one iterated sum's block
  // Arctic argmax (fr_pipeline_set_argmax): the OUTPUT rows are the L + L (L + 1) / 2 rows of every
  // word (running maxima and back-tracked positions, fruits/iss/semiring.py:239-284), not the
  // plan's; (n_words, 4) {first plan row, letters, first output row, 0}
  std::vector<int32_t> argmax_words;
  void *d_argmax_words = nullptr;
  int32_t argmax_rows = 0, argmax_max_len = 0;
  int rows() const { return argmax_words.empty() ? plan->p->K : argmax_rows; }   // output rows
  void *d_ops = nullptr;           // (rows, n_ops_padded) FeatOp
  void *d_mpi_cols = nullptr;
  bool have_quantiles = false;
  // per-series cut table (fr_pipeline_set_series_cuts): device (cuts_N, cut_slots) int32, owned
  // by the caller; cut_slots_needed = 1 + the highest slot a sieve names
  const int32_t *d_series_cuts = nullptr;
  int64_t cuts_N = 0;
  int32_t cut_slots = 0, cut_slots_needed = 0;
  // fused preparation (fr_pipeline_set_preparation): 0 dims = none
  int32_t prep_D = 0, prep_n = 0, prep_std = 0;
  double prep_eps = 0.0;
  void *d_prep = nullptr;          // (prep_n, 4) int32
  // run-time compiled fused kernels (jit.cpp, walk_fused.h JitOps): the sieves' kind /
  // differencing order / shape / cuts, the same for every output row, as immediates; compiled by
  // fr_pipeline_prepare for the kernel instantiation the plan and T select, dropped when the
  // thresholds (and with them the ops) are set again
  // (fr_pipeline_prepare may run on another thread than fr_pipeline_run - a caller that does
  // not want to wait for the compiler: jit_mu guards this block, jit_gen says whether the ops a
  // compilation started from are still the pipeline's)
  std::mutex jit_mu;
  uint64_t jit_gen = 0;
  fr::FusedOps jit_ops;
  bool jit_uniform = false;        // every row's ops agree in what becomes an immediate
  std::map<uint32_t, fr::JitProgram> jit;
  std::map<uint32_t, std::string> jit_failed;
  std::set<uint32_t> jit_pending;  // being compiled right now
  // the same with the PLAN as an immediate too (small plans: walk_fused.h, fwalk_static), by
  // instantiation and groups per series: id | groups << 32
  std::map<uint64_t, fr::JitProgram> jit_static;
  std::set<uint64_t> jit_static_tried;
  // ... or, for a large plan, the plan in PIECES (plan.h, PiecedProgram; walk_fused.h,
  // fwalk_pieces): one kernel per piece type, by instantiation; the op table in walk order and
  // the walk position of every output row (uploaded by fr_pipeline_prepare on the caller's
  // thread; the kernels may come from a helper thread)
  std::vector<fr::FeatOp> h_ops;   // host copy of the op table
  struct Pieces {
    int max_piece = 0, device = -1;
    std::vector<fr::JitProgram> progs;   // one per piece type; empty: not compiled (yet)
    std::vector<char> fits;              // per type: compiled for units whose features fit the window
    void *d_tables = nullptr;
    const fr::FeatOp *d_ops_walk = nullptr;
    const int32_t *d_walk_of_row = nullptr;
  };
  std::map<uint32_t, Pieces> jit_pieces;
  std::set<uint32_t> jit_pieces_tried;
  void drop_pieces() {               // (caller holds jit_mu)
    for (auto &kv : jit_pieces) {
      for (fr::JitProgram &pr : kv.second.progs) fr::jit_unload(pr);
      if (kv.second.d_tables) (void)hipFree(kv.second.d_tables);
    }
    jit_pieces.clear();
    jit_pieces_tried.clear();
  }
};


namespace {

struct FusedArgs {          // non-null feats selects the fused sieve kernels
  const fr::FeatOp *ops = nullptr;
  double *feats = nullptr, *cnt = nullptr;
  int64_t feat_stride = 0;
  int32_t n_ops = 0, n_ops_padded = 0;
  bool has_mpi = false;     // cnt is a population table of its own
  fr_pipeline *pl = nullptr;   // the pipeline (its run-time compiled kernels), if any
  const int32_t *series_cuts = nullptr;   // device (N, cut_slots) per-series boundaries
  int32_t cut_slots = 0;
  bool total_inc = false;   // a differencing sieve on a totally weighted plan
  int carry_per_node = 3;   // chunk-carry slots of a node: 3, + 2 per differencing order >= 3
  // fused preparation: d_X is the raw input, the staging forms the prepared rows
  const int32_t *prep = nullptr;   // device (n_prep, 4) table
  const double *stats = nullptr;   // device (N, n_prep, 2) or nullptr (no STD)
  int32_t n_prep = 0;
  // a plan in pieces writes its features in walk order: (N, K * per_sum) scratch, and says so
  double *walk_feats = nullptr;
  const int32_t **walk_of_row = nullptr;   // set by the launch: the walk position of every output row
};

// The instantiation of the fused walk a (plan, series length, sieves) selects - what
// walk_inst.hip's dispatch picks at launch time, as a key for the run-time compiled variants.
fr::FusedKey fused_key_for(const fr::Plan &p, int64_t T, bool total_inc, bool high_order) {
  const int64_t chunk = fr::walk_chunk_elems(T);
  fr::FusedKey k{};
  k.E = chunk == 512 ? 2 : 4;
  k.LV = p.levels <= 2 ? 2 : (p.levels <= 4 ? 4 : (p.levels <= 6 ? 6 : 8));
  k.MULTI = T > chunk ? 1 : 0;
  k.W = p.weighting != 0 ? 1 : 0;
  k.SEMI = p.semiring;
  k.TI = (k.W && total_inc && p.weighting == FR_W_TOTAL) ? 1 : 0;
  k.TOTAL = (k.W && p.weighting == FR_W_TOTAL) ? 1 : 0;
  k.HO = (k.MULTI && high_order) ? 1 : 0;
  return k;
}

// Compiles (hipRTC, disk cache) and loads the pipeline's fused kernel with its sieves as
// immediates, once per instantiation; a failure leaves the pipeline on the generic kernel.
// (`cache_only`: from the disk cache or not at all - a miss leaves no trace, a later call compiles)
void ensure_fused_jit(fr_pipeline &pl, const fr::FusedKey &key, bool cache_only = false) {
  const uint32_t id = key.packed();
  fr::FusedOps ops;
  uint64_t gen;
  {
    std::lock_guard<std::mutex> lock(pl.jit_mu);
    if (!pl.jit_uniform || pl.jit.count(id) || pl.jit_failed.count(id) || pl.jit_pending.count(id))
      return;
    pl.jit_pending.insert(id);
    ops = pl.jit_ops;
    gen = pl.jit_gen;
  }
  fr::JitProgram prog;
  std::string err;
  const bool ok = fr::jit_fused(ops, key, prog, err, nullptr, cache_only);   // (seconds: nobody waits on a lock for it)
  std::lock_guard<std::mutex> lock(pl.jit_mu);
  pl.jit_pending.erase(id);
  if (!ok && cache_only && fr::jit_not_cached(err)) return;
  if (gen != pl.jit_gen) {   // the thresholds were set again meanwhile: not this pipeline's kernel
    if (ok) fr::jit_unload(prog);
    return;
  }
  if (ok)
    pl.jit[id] = prog;
  else
    pl.jit_failed[id] = err;
}

// The straight-line variant for the group program `gp` (a copy of its records goes into the source).
void ensure_fused_static(fr_pipeline &pl, const fr::FusedKey &key, const fr::FusedPlan &plan,
                         bool cache_only = false) {
  const uint64_t id = (uint64_t)key.packed() | (uint64_t)plan.groups() << 32;
  fr::FusedOps ops;
  uint64_t gen;
  {
    std::lock_guard<std::mutex> lock(pl.jit_mu);
    if (!pl.jit_uniform || pl.jit_static.count(id) || pl.jit_static_tried.count(id)) return;
    pl.jit_static_tried.insert(id);
    ops = pl.jit_ops;
    gen = pl.jit_gen;
  }
  fr::JitProgram prog;
  std::string err;
  const bool ok = fr::jit_fused(ops, key, prog, err, &plan, cache_only);
  std::lock_guard<std::mutex> lock(pl.jit_mu);
  if (!ok && cache_only && fr::jit_not_cached(err)) pl.jit_static_tried.erase(id);
  if (gen != pl.jit_gen) {
    if (ok) fr::jit_unload(prog);
    return;
  }
  if (ok) pl.jit_static[id] = prog;
}

// ---- a large plan in pieces (plan.h, PiecedProgram) ----------------------------------------
// Plans of more than kFusedStaticMaxNodes nodes (developer knobs: pieces=0 - never;
// piece_min=M - from M nodes on; piece_nodes=P - pieces of at most P nodes).
// Nodes of the largest piece: a body's code grows with nodes x feature ops per node, and the
// compiler's time faster than that - pipelines of more than two ops per output row (the experiment
// fruits' seven sieves: four ops with MPI sums and second differences) get pieces of half the size
// (the 115-node body of of_weight(6,2) with four ops: ~200 s on the build host, its 33 / 47 / 62-node
// bodies ~60 s together).
int piece_nodes_knob(const fr_pipeline &pl) {
  return debug_knob("piece_nodes", pl.n_ops_eff > 2 ? fr::kFusedPieceNodes / 2 : fr::kFusedPieceNodes);
}
bool pieces_eligible(const fr_pipeline &pl) {
  const fr::Plan &p = *pl.plan->p;
  // (with more than two feature ops per output row a whole plan of ~100 nodes is as much code as
  // a 200-node plan with two - a minute and more of compiler: in pieces from 65 nodes on)
  const int from = pl.n_ops_eff > 2 ? fr::kFusedPieceNodes / 2 + 1 : fr::kFusedStaticMaxNodes + 1;
  return !p.cos && !p.letter_sum && debug_knob("pieces", 1) != 0 && env_int("FRUITS_HIP_JIT", 1) != 0 &&
         (int)p.nodes.size() >= debug_knob("piece_min", from);
}

// Uploads the tables of every piece type once per plan.  Caller holds p.mu; never inside a capture.
int ensure_piece_tables(fr::Plan &p, fr::PiecedProgram &pp, const char *who) {
  int rc = claim_device(p, who);
  if (rc != FR_OK) return rc;
  for (fr::PieceType &t : pp.types) {
    if (t.d_blob) continue;
    size_t off = 0;
    const size_t o_recs = off;   off = align_up(off + t.recs.size() * sizeof(fr::NodeRec), 64);
    const size_t o_emit = off;   off = align_up(off + t.emit_rows.size() * 4, 64);
    const size_t o_items = off;  off = align_up(off + t.items.size() * 4, 64);
    const size_t o_ub = off;     off = align_up(off + t.unit_begin.size() * 4, 64);
    const size_t o_ur = off;     off = align_up(off + t.unit_row0.size() * 4, 64);
    std::vector<char> host(off + 64, 0);
    std::memcpy(host.data() + o_recs, t.recs.data(), t.recs.size() * sizeof(fr::NodeRec));
    std::memcpy(host.data() + o_emit, t.emit_rows.data(), t.emit_rows.size() * 4);
    std::memcpy(host.data() + o_items, t.items.data(), t.items.size() * 4);
    std::memcpy(host.data() + o_ub, t.unit_begin.data(), t.unit_begin.size() * 4);
    std::memcpy(host.data() + o_ur, t.unit_row0.data(), t.unit_row0.size() * 4);
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, host.size()));
    hipError_t e = hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(d);
      return hip_fail(e, "hipMemcpy(piece tables)");
    }
    char *b = static_cast<char *>(d);
    t.d_blob = d;
    t.d_recs = reinterpret_cast<const fr::NodeRec *>(b + o_recs);
    t.d_emit_rows = reinterpret_cast<const int32_t *>(b + o_emit);
    t.d_items = reinterpret_cast<const int32_t *>(b + o_items);
    t.d_unit_begin = reinterpret_cast<const int32_t *>(b + o_ub);
    t.d_unit_row0 = reinterpret_cast<const int32_t *>(b + o_ur);
  }
  return FR_OK;
}

// The pipeline's side of a plan in pieces: the op table in walk order (an op's column = its
// row's walk position x features per sum + its place in the block) and the walk position of
// every output row, for the instantiation `key`.  Synchronous uploads: the caller's thread,
// never inside a capture.  FR_OK also when the plan has no cover.
int ensure_pieces_tables(fr_pipeline &pl, const fr::FusedKey &key, const char *who) {
  fr::Plan &p = *pl.plan->p;
  const int max_piece = piece_nodes_knob(pl);
  fr::PiecedProgram *pp;
  {
    std::lock_guard<std::mutex> lock(p.mu);
    pp = &fr::pieced(p, max_piece, debug_knob("piece_unit", 0));
    if (!pp->ok) return FR_OK;
    int rc = ensure_piece_tables(p, *pp, who);
    if (rc != FR_OK) return rc;
  }
  std::lock_guard<std::mutex> lock(pl.jit_mu);
  if (!pl.jit_uniform) return FR_OK;
  fr_pipeline::Pieces &pcs = pl.jit_pieces[key.packed()];
  if (pcs.d_tables) return FR_OK;
  const int K = p.K, npad = pl.n_ops_padded;
  std::vector<fr::FeatOp> walk((size_t)K * npad);
  std::vector<int32_t> walk_of_row(K, 0);
  for (int q = 0; q < K; ++q) {
    const int k = pp->row_of_walk[q];
    walk_of_row[k] = q;
    for (int i = 0; i < npad; ++i) {
      fr::FeatOp o = pl.h_ops[(size_t)k * npad + i];
      if (i < pl.n_ops_eff) o.col = q * pl.per_sum + (o.col - k * pl.per_sum);
      walk[(size_t)q * npad + i] = o;
    }
  }
  const size_t ops_bytes = align_up(walk.size() * sizeof(fr::FeatOp), 256);
  void *d = nullptr;
  HIP_TRY(hipMalloc(&d, ops_bytes + (size_t)K * 4));
  hipError_t e = hipMemcpy(d, walk.data(), walk.size() * sizeof(fr::FeatOp), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = hipMemcpy(static_cast<char *>(d) + ops_bytes, walk_of_row.data(), (size_t)K * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(d);
    return hip_fail(e, "hipMemcpy(ops in walk order)");
  }
  pcs.max_piece = max_piece;
  pcs.device = current_device_id();
  pcs.d_tables = d;
  pcs.d_ops_walk = static_cast<const fr::FeatOp *>(d);
  pcs.d_walk_of_row = reinterpret_cast<const int32_t *>(static_cast<char *>(d) + ops_bytes);
  return FR_OK;
}

// LDS of a launch of piece type `pt` next to the feature window, and the window (0: none fits)
size_t piece_other_lds(const fr::Plan &p, const fr::PieceType &pt, int64_t T, int carry_per_node) {
  const int64_t chunk = fr::walk_chunk_elems(T);
  return ((size_t)p.rows_staged() * chunk + 24 + (T > chunk ? (size_t)carry_per_node * pt.max_unit_nodes : 0)) * 8;
}
int piece_window(const fr::Plan &p, const fr::PieceType &pt, int64_t T, int carry_per_node, int n_ops,
                 bool mpi, bool &fits) {
  return feat_window_sized(pt.widest_node * n_ops, pt.max_unit_rows * n_ops,
                           piece_other_lds(p, pt, T, carry_per_node), mpi, fits);
}

int piece_level_variant(int levels) { return levels <= 2 ? 2 : (levels <= 4 ? 4 : (levels <= 6 ? 6 : 8)); }

// Compiles (hipRTC, one helper thread per piece type; disk cache) and loads the kernels of the
// plan's piece types for the instantiation `key`; all of them or none.
void ensure_fused_pieces(fr_pipeline &pl, const fr::FusedKey &key, bool cache_only) {
  fr::Plan &p = *pl.plan->p;
  const uint32_t id = key.packed();
  fr::FusedOps ops;
  uint64_t gen;
  int max_piece;
  const int n_ops_eff = pl.n_ops_eff;
  const bool has_mpi = !pl.mpi_cols.empty();
  {
    std::lock_guard<std::mutex> lock(pl.jit_mu);
    auto it = pl.jit_pieces.find(id);
    if (!pl.jit_uniform || it == pl.jit_pieces.end() || !it->second.d_tables || !it->second.progs.empty() ||
        pl.jit_pieces_tried.count(id))
      return;
    pl.jit_pieces_tried.insert(id);
    ops = pl.jit_ops;
    gen = pl.jit_gen;
    max_piece = it->second.max_piece;
  }
  const fr::PiecedProgram *pp;
  {
    std::lock_guard<std::mutex> lock(p.mu);
    pp = &fr::pieced(p, max_piece);   // (map nodes are stable; built and uploaded by ensure_pieces_tables)
  }
  const int n_types = (int)pp->types.size();
  std::vector<fr::JitProgram> progs(n_types);
  std::vector<std::string> errs(n_types);
  std::vector<char> good(n_types, 0), type_fits(n_types, 0);
  const int dev = current_device_id();
  std::atomic<int> next{0};
  auto worker = [&] {
    (void)hipSetDevice(dev);   // (the current device is per thread)
    for (int t = next++; t < n_types; t = next++) {
      fr::FusedPlan fp;
      fp.w = pp->types[t].body_w;
      fp.piece = true;
      fr::FusedKey k = key;
      k.LV = piece_level_variant(pp->types[t].levels);
      // (whether the largest unit's features fit the window is known here: no flush test then)
      fr::FusedOps type_ops = ops;
      bool fits = false;
      (void)piece_window(p, pp->types[t], pl.T, ops.cps, n_ops_eff, has_mpi, fits);
      type_ops.window_fits = fits;
      type_fits[t] = fits ? 1 : 0;
      good[t] = fr::jit_fused(type_ops, k, progs[t], errs[t], &fp, cache_only) ? 1 : 0;
    }
  };
  const int hw = (int)std::thread::hardware_concurrency();
  const int n_threads = cache_only ? 1 : std::max(1, std::min({n_types, hw > 0 ? hw : 4, 16}));
  std::vector<std::thread> pool;
  for (int i = 1; i < n_threads; ++i) pool.emplace_back(worker);
  worker();
  for (std::thread &th : pool) th.join();
  bool all = true, missing = false;
  for (int t = 0; t < n_types; ++t) {
    if (!good[t]) all = false;
    if (!good[t] && cache_only && fr::jit_not_cached(errs[t])) missing = true;
  }
  std::lock_guard<std::mutex> lock(pl.jit_mu);
  auto it = pl.jit_pieces.find(id);
  if (!all || gen != pl.jit_gen || it == pl.jit_pieces.end()) {
    for (int t = 0; t < n_types; ++t)
      if (good[t]) fr::jit_unload(progs[t]);
    // (a miss of the cache-only look leaves no trace: a later call compiles)
    if (!all && missing && gen == pl.jit_gen) pl.jit_pieces_tried.erase(id);
    else if (!all)
      for (int t = 0; t < n_types; ++t)
        if (!good[t]) {
          pl.jit_failed[id | 0x80000000u] = errs[t];
          break;
        }
    return;
  }
  it->second.progs = std::move(progs);
  it->second.fits = std::move(type_fits);
}

// Shared body of fr_iss_run and fr_pipeline_run: validates, lays out the
// workspace, fills the exp tables and launches the trie walk.
int run_walk(const char *who, fr::Plan &p, const double *d_X, int64_t N, int64_t D, int64_t T,
             const double *d_lookup, int64_t lookup_rows, double *d_out, int64_t out_k_stride,
             int64_t out_n_stride, void *d_work, int64_t work_bytes, int32_t groups,
             hipStream_t st, const FusedArgs *fu) {
  const std::string w(who);
  if (N < 0 || D < 1 || T < 0) return fail(FR_E_ARG, w + ": bad shape");
  const int64_t D_words = (fu && fu->prep) ? fu->n_prep : D;   // dimensions the words may name
  if (p.max_dim > D_words)
    return fail(FR_E_DIM, w + ": a word references dimension " + std::to_string(p.max_dim) +
                              " but the input has only " + std::to_string(D_words));
  if (N == 0 || T == 0 || p.K == 0 || (!p.cos && p.nodes.empty())) return FR_OK;
  if (!d_X || (!fu && !d_out)) return fail(FR_E_ARG, w + ": null device pointer");
  if (p.cos) {
    fr::CosProgram &c = *p.cos;
    if (c.exponent > fr::kCosMaxExponent || p.levels > 16)
      return fail(FR_E_LIMIT, w + ": CosWISS kernels cover exponents <= 4 and words of <= 16 "
                              "letters");
    const size_t need = work_layout(p, N, T, 0).total();
    if (!d_work || (size_t)work_bytes < need)
      return fail(FR_E_NOMEM, w + ": workspace too small (need " + std::to_string(need) +
                                  " bytes)");
    {
      std::lock_guard<std::mutex> lock(p.mu);
      int rc = ensure_cos_program(p, c, st, who);
      if (rc != FR_OK) return rc;
    }
    double *trig = static_cast<double *>(d_work);
    hipError_t e = fr::launch_trig_tables(c.d_freqs, c.F, T, trig, st);
    if (e != hipSuccess) return hip_fail(e, "trig_tables launch");
    fr::IssArgs a{};
    a.X = d_X;
    a.aux = trig;
    a.out = d_out;
    a.N = N;
    a.D = D;
    a.T = T;
    a.out_k_stride = out_k_stride;
    a.out_n_stride = out_n_stride;
    a.factors = c.d_factors;
    a.cw_letter_begin = c.d_letter_begin;
    a.cw_fac_begin = c.d_fac_begin;
    a.cw_W = c.W;
    a.cw_F = c.F;
    a.cw_total = c.total ? 1 : 0;
    if (c.d_mask) {
      if (c.mask_T != T)
        return fail(FR_E_ARG, w + ": the dropout mask was set for series of length " +
                                  std::to_string(c.mask_T));
      a.cw_mask = static_cast<const double *>(c.d_mask);
      a.cw_Lmax = c.Lmax;
    }
    a.cw_x_unit_stride = c.x_unit_stride;
    if (fu && fu->prep) {
      if (c.x_unit_stride != 0)
        return fail(FR_E_LIMIT, w + ": a CosWISS with per-unit inputs (ffn) has no fused preparation");
      a.prep = fu->prep;
      a.stats = fu->stats;
      a.n_prep = fu->n_prep;
    }
    a.packed = (T <= 384 && debug_knob("packed", 1) != 0) ? 1 : 0;
    a.vec_ok = (T % 2 == 0) && aligned16(d_X) && aligned16(trig) &&
               (fu || (aligned16(d_out) && (out_k_stride % 2 == 0) && (out_n_stride % 2 == 0)));
    if (fu) {
      a.ops = fu->ops;
      a.feats = fu->feats;
      a.cnt = fu->cnt;
      a.feat_stride = fu->feat_stride;
      a.n_ops = fu->n_ops;
      a.n_ops_padded = fu->n_ops_padded;
      a.series_cuts = fu->series_cuts;
      a.cut_slots = fu->cut_slots;
      // feature window of the cooperative kernel: the ops of the unit's one output row
      a.has_mpi = fu->has_mpi ? 1 : 0;
      a.feat_window = (fu->n_ops + 1) / 2 * 2;
      a.feat_fits = 1;
      if (a.feat_window > 4096)
        return fail(FR_E_LIMIT, w + ": too many sieve features per iterated sum for the fused launch");
    }
    // one short-lived workgroup per (series, word, frequency) unit: 0-2.5 % faster than a
    // persistent grid (exponent 2: 1515 -> 1478 us)
    a.persistent = 0;
    e = fr::launch_coswiss(a, c.exponent, st);
    if (e != hipSuccess) return hip_fail(e, "coswiss launch");
    return FR_OK;
  }
  if (p.weighting != 0) {
    if (!d_lookup) return fail(FR_E_ARG, w + ": weighted plan needs a lookup");
    if (lookup_rows != 1 && lookup_rows != N)
      return fail(FR_E_ARG, w + ": lookup_rows must be 1 or N");
  }
  const WorkLayout wl = work_layout(p, N, T, p.weighting ? lookup_rows : 0);
  if (wl.total() > 0 && (!d_work || (size_t)work_bytes < wl.total()))
    return fail(FR_E_NOMEM, w + ": workspace too small (need " + std::to_string(wl.total()) +
                                " bytes)");
  const bool vec_ok_pre = (T % 2 == 0) && aligned16(d_X) &&
                          (fu || (aligned16(d_out) && (out_k_stride % 2 == 0) &&
                                  (out_n_stride % 2 == 0)));
  // the staged rows (input dimensions + exp tables) of one time chunk must fit the LDS
  LaunchShape shape = launch_shape(p, N, T, groups);
  if (!shape.fits)
    return fail(FR_E_LIMIT, w + ": the plan stages " + std::to_string(p.rows_staged()) +
                                " rows per time chunk (input dimensions + exp tables of " +
                                std::to_string(p.alphas.size()) +
                                " distinct alphas), more than the LDS holds - split the word list");
  // (a totally weighted plan with differencing sieves runs the cooperative kernels, which have
  // the instantiation for it, also on short series)
  const bool packed = shape.packed &&
                      !(fu && fu->total_inc && p.weighting == FR_W_TOTAL);
  const bool auto_groups = !packed && groups <= 0 && debug_knob("groups", 0) <= 0;
  const int64_t resident = auto_groups ? query_resident(p, N, T, fu != nullptr, vec_ok_pre) : 0;
  fr::GroupedProgram *gpp = nullptr;
  int static_lds_pad = 0;
  int static_prog = 0;                      // > 0: ahead-of-time program, -1: run-time compiled
  const fr::JitProgram *jit_prog = nullptr;
  {
    std::lock_guard<std::mutex> lock(p.mu);
    // A pre-compiled static program (walk_static_inst.hip) runs plans whose records equal one
    // of the standard word sets': materialising, one aligned 1024-element chunk, unweighted
    // Reals, the group count its schedule was generated for.  It reads no device tables, so
    // nothing is uploaded for it (and a run of it is capturable without fr_plan_prepare).
    int static_groups = 0;
    const int asked = groups > 0 ? groups : debug_knob("groups", 0);
    if (!fu && !packed && vec_ok_pre && static_shape_ok(p, T) && N > 0 &&
        asked <= 3 && env_int("FRUITS_HIP_STATIC", 1) != 0) {
      if (p.static_prog[0] < 0) {
        const fr::GroupedProgram &g1 = fr::grouped(p, 1);
        for (int g = 1; g <= 3; ++g)
          p.static_prog[g] = fr::static_program_for(g1.recs.data(), (int)g1.recs.size(), g, p.row_src.data(),
                                                       (int)p.row_src.size());
        p.static_prog[0] = 0;
      }
      // no ahead-of-time program: one compiled at run time by fr_plan_prepare - or right here
      // with FRUITS_HIP_JIT=2 (a couple of seconds, once per plan; never inside a capture)
      const bool aot = p.static_prog[1] > 0;
      if (!aot && env_int("FRUITS_HIP_JIT", 1) == 2 && !stream_is_capturing(st)) ensure_jit(p);
      JitState *js = aot ? nullptr : static_cast<JitState *>(p.jit);
      auto have = [&](int g) {
        return p.static_prog[g] > 0 || (js != nullptr && js->progs.count(g) != 0);
      };
      // Groups per series.  Small batches: as many groups as the schedule has, to fill the
      // chip.  Batches whose input + output are at most 1.4 x the 256 MiB Infinity Cache: ONE
      // group - every input row is then read once, with non-temporal loads that do not
      // allocate in that cache, where the input would only evict output lines (config 2:
      // 69 -> 56 us).  Larger batches stream through HBM whatever is done; there the
      // finer units balance better (N = 8192: 273 vs 283 us).
      // (round 4, tools/static_window.py, fraction of 8 TB/s, one group + nt / three groups /
      // no static program: N = 2048 (1.31 x the cache) 0.763 / 0.654 / 0.656; 2304 (1.48 x)
      // 0.637 / 0.679 / 0.623; 3072 (1.97 x) 0.638 / 0.712 / 0.675; 4096 0.719 / 0.713 / 0.680 -
      // the window used to end at 2 x, where the sweep showed the static program behind the
      // walk without one)
      const int gmax = have(3) ? 3 : (have(2) ? 2 : 1);
      const double footprint = 8.0 * (double)N * (double)T * (double)(p.dims_used + p.K);
      const bool cache_sized = N >= kStaticSplitBelow &&
                               footprint <= 0.01 * debug_knob("static_cache_x100", 140) * 256.0 * 1024.0 * 1024.0;
      static_groups = asked > 0 ? asked : (cache_sized ? 1 : gmax);
      // Batches that stream through HBM (beyond twice the cache): FOUR resident workgroups per
      // CU instead of six - fewer concurrent write streams suit the memory system better
      // (N = 4096 / 8192 / 16384: 134 -> 122, 263 -> 241, 525 -> 493 us); 16 KB of unused LDS
      // per workgroup is how a launch asks for that.  Cache-sized and small batches keep six
      // (N = 2048: 56.2 vs 58.6 us with four).
      static_lds_pad = (!cache_sized && N >= kStaticSplitBelow) ? 16384 : 0;
      // Series of 385 ... 512 elements fill half of the program's 1024-element chunk: measured
      // (round 4, of_weight(2,3), fraction of 8 TB/s, interpreter with its 512-element chunk /
      // static program) T = 512: N = 2048 0.572 / 0.625, 4096 0.567 / 0.744, 8192 (streams through
      // HBM) 0.641 / 0.538; T = 400: 0.506 / 0.594, 0.559 / 0.736, 0.503 / 0.443 - the program on
      // cache-sized batches only
      if (have(static_groups) && (T > 512 || cache_sized)) {
        static_prog = p.static_prog[static_groups] > 0 ? p.static_prog[static_groups] : -1;
        if (static_prog < 0) {
          jit_prog = &js->progs[static_groups];
          // (a module is loaded on ONE device: elsewhere the interpreter runs the plan)
          if (jit_prog->device != fr::current_device()) {
            jit_prog = nullptr;
            static_prog = 0;
          }
        }
      }
    }
    const int G = static_prog ? static_groups
                              : (auto_groups ? choose_groups_walk(p, N, T, resident, fu != nullptr) : shape.G);
    gpp = &fr::grouped(p, G);   // (map nodes are stable: the reference outlives the lock)
    if (!static_prog) {
      int rc = ensure_device_program(p, *gpp, st, who);
      if (rc != FR_OK) return rc;
    }
  }
  fr::GroupedProgram &gp = *gpp;

  fr::IssArgs a{};
  a.X = d_X;
  a.out = d_out;
  a.N = N;
  a.D = D;
  a.T = T;
  a.out_k_stride = out_k_stride;
  a.out_n_stride = out_n_stride;
  a.recs = gp.d_recs;
  a.factors = gp.d_factors;
  a.emit_rows = gp.d_emit_rows;
  a.slot_rows = gp.d_slot_rows;
  a.group_row_begin = gp.d_group_row_begin;
  a.shape_ids = gp.d_shape_ids;
  a.group_begin = gp.d_group_begin;
  a.row_src = gp.d_row_src;
  a.G = gp.groups;
  a.R = p.rows_staged();
  a.total_nodes = (int32_t)p.nodes.size();
  char *work = static_cast<char *>(d_work);
  if (p.weighting != 0) {
    double *aux = reinterpret_cast<double *>(work);
    const int64_t count = lookup_rows * T;
    hipError_t e = fr::launch_exp_tables(d_lookup, count, gp.d_alphas, (int)p.alphas.size(), aux,
                                         p.semiring == fr::kSemiArctic, st);
    if (e != hipSuccess) return hip_fail(e, "exp_tables launch");
    a.aux = aux;
    a.aux_tab_stride = count;
    a.aux_n_stride = lookup_rows == 1 ? 0 : T;
  }
  if (wl.carry_bytes) a.carry = reinterpret_cast<double *>(work + wl.aux_bytes);
  a.vec_ok = vec_ok_pre && (!a.aux || aligned16(a.aux));
  a.debug = debug_knob("stamps", 0);
  if (a.debug & 16) {
    // diagnostic build only: stamps go to the tail of the workspace if the caller
    // sized it with FRUITS_HIP_DBG_BYTES extra bytes
    const int64_t extra = debug_knob("dbg_bytes", 0);
    if (extra > 0 && d_work && work_bytes >= (int64_t)wl.total() + extra)
      a.dbg = reinterpret_cast<unsigned long long *>(work + align_up(wl.total(), 256));
  }
  // Persistent grid (one resident round of workgroups striding over the units) or one
  // short-lived workgroup per unit.  Measured (tools/gpu_persist.sh, gpu_static2.sh): the fused
  // kernels gain 3-12 % from the hardware dispatcher's balancing (config 4: 25.1 -> 22.1 ms,
  // config 5: 42.6 -> 37.3 ms); the materialising interpreter keeps the persistent grid up
  // to two resident rounds (config 2: 68.9 vs 73.7 us) and drops it beyond (N = 8192:
  // 287 -> 256 us).
  {
    const int64_t round = resident > 0 ? resident : 1536;
    int by_shape = fu ? 0 : ((N * (int64_t)gp.groups < 2 * round) ? 1 : 0);
    // wave-per-series kernels (short series), materialising: T <= 128 without the persistent
    // grid (16384 x 128: 86 -> 77 us, 32768 x 64: 120 -> 100 us), longer ones with (8192 x 256:
    // 71 vs 75 us)
    if (packed && !fu) by_shape = T > 192 ? 1 : 0;
    a.persistent = debug_knob("persist", by_shape);
  }
  a.packed = packed ? 1 : 0;
  a.prefetch_next = 24;  // longest unit (nodes) that touches its successor's rows
  a.semiring = p.semiring;
  a.letter_sum = p.letter_sum ? 1 : 0;
  a.k_stride_bytes32 = (out_k_stride > 0 && out_k_stride < (int64_t(1) << 29))
                           ? (uint32_t)(out_k_stride * 8) : 0u;
  a.xcd_map = (a.G > 1 && N % 8 == 0) ? 1 : 0;
  a.carry_slots = carry_slots_for(p, gp.groups);
  a.carry_per_node = 3;
  a.carry_in_lds = carries_fit_lds(p, T, gp.groups) ? 1 : 0;
  if (fu && !packed) {
    // the fused walk keeps its chunk carries in LDS, three slots per record of the largest group
    int most = 0;
    for (int g = 0; g < gp.groups; ++g) most = std::max(most, gp.group_begin[g + 1] - gp.group_begin[g]);
    a.carry_per_node = fu->carry_per_node;
    a.carry_slots = a.carry_per_node * most;
    a.carry_in_lds = 1;
  }
  if (fu) {
    a.ops = fu->ops;
    a.feats = fu->feats;
    a.cnt = fu->cnt;
    a.feat_stride = fu->feat_stride;
    a.n_ops = fu->n_ops;
    a.n_ops_padded = fu->n_ops_padded;
    a.series_cuts = fu->series_cuts;
    a.cut_slots = fu->cut_slots;
    a.total_inc = (fu->total_inc && p.weighting == FR_W_TOTAL) ? 1 : 0;
    a.high_order = fu->carry_per_node > 3 ? 1 : 0;
    a.total_weighting = p.weighting == FR_W_TOTAL ? 1 : 0;
    a.has_mpi = fu->has_mpi ? 1 : 0;
    if ((int64_t)p.K * fu->n_ops_padded * 32 >= (int64_t(1) << 32) || gp.recs.size() >= (size_t(1) << 26))
      return fail(FR_E_LIMIT, w + ": the program tables exceed 4 GiB - split the word list");
    if (!packed) {
      const int64_t chunk = fr::walk_chunk_elems(T);
      const size_t other = ((size_t)a.R * chunk + 24 + (T > chunk ? a.carry_slots : 0)) * 8;
      bool fits = false;
      a.feat_window = feat_window_for(gp, other, fu->n_ops, fu->has_mpi, fits);
      a.feat_fits = fits ? 1 : 0;
      if (a.feat_window == 0)
        return fail(FR_E_LIMIT, w + ": the chunk carries and the features of one node (output rows "
                                    "x sieve features) do not fit the LDS - split the word list");
      if (p.letter_sum)
        return fail(FR_E_LIMIT, w + ": letter-sum (argmax) plans have no fused walk");
    }
    if (fu->prep) {
      a.prep = fu->prep;
      a.stats = fu->stats;
      a.n_prep = fu->n_prep;
    }
  }
  a.static_prog = static_prog > 0 ? static_prog : 0;
  a.lds_pad = static_lds_pad;
  // Materialising launches of the interpreter's plans run through the fused walk's node loop with
  // a store epilogue (walk_fused.h, MODE 2: half the instructions per node) whenever that walk
  // covers the plan: chunk carries in LDS, no letter sums (Arctic argmax).
  const int64_t resident_round = resident > 0 ? resident : 1536;
  if (!fu && !packed && !static_prog && !p.letter_sum && debug_knob("lean", 1) != 0 &&
      (p.nodes.size() > 32 || N * (int64_t)gp.groups >= 2 * resident_round)) {
    // (short plans on batches of less than two resident rounds keep the interpreter's persistent
    // grid and its prefetch of the next unit's rows: of_weight(2,3) at N = 2048 66 vs 75 us)
    int most = 0;
    for (int g = 0; g < gp.groups; ++g) most = std::max(most, gp.group_begin[g + 1] - gp.group_begin[g]);
    const int64_t chunk = fr::walk_chunk_elems(T);
    const size_t lds = ((size_t)a.R * chunk + 24 + (T > chunk ? 3 * (size_t)most : 0)) * 8;
    if (lds <= 40 * 1024 || T <= chunk) {
      a.lean = 1;
      a.carry_slots = 3 * most;
      a.carry_in_lds = 1;
      a.persistent = 0;
      a.total_weighting = p.weighting == FR_W_TOTAL ? 1 : 0;
    }
  }
  // The interpreter's share of the same finding, in the window where it was measured to pay:
  // one group per series and a batch just above the Infinity Cache (1 to 1.5 times its
  // 256 MiB - config 2: 70 -> 65 us; 264 MB: 43 -> 45 us, 440 MB: 88 -> 93 us, so not there).
  {
    const double footprint = 8.0 * (double)N * (double)T * (double)(p.dims_used + p.K);
    const double cache = 256.0 * 1024.0 * 1024.0;
    a.nt_input = (!fu && !packed && a.G == 1 && footprint > cache &&
                  footprint <= 1.5 * cache) ? 1 : 0;
  }
  // static programs of several groups run one short-lived workgroup per unit: the hardware
  // dispatcher balances them and keeps the write front compact
  if (static_prog) a.persistent = 0;
  if (fu && fu->pl && !packed && fu->walk_feats != nullptr && fu->walk_of_row != nullptr) {
    // a large plan in pieces (plan.h, PiecedProgram): one launch per piece type, each over
    // (series x the type's units); the features leave in walk order
    const fr::FusedKey key = fused_key_for(p, T, fu->total_inc, fu->carry_per_node > 3);
    fr_pipeline::Pieces pcs;
    {
      std::lock_guard<std::mutex> lock(fu->pl->jit_mu);
      auto it = fu->pl->jit_pieces.find(key.packed());
      if (it != fu->pl->jit_pieces.end() && !it->second.progs.empty() &&
          it->second.device == fr::current_device())
        pcs = it->second;
    }
    if (!pcs.progs.empty()) {
      const fr::PiecedProgram *pp;
      {
        std::lock_guard<std::mutex> lock(p.mu);
        pp = &p.pieced.at(pcs.max_piece);
      }
      const int64_t chunk = fr::walk_chunk_elems(T);
      const int64_t F = (int64_t)p.K * fu->pl->per_sum;
      // (one launch per type, back to back on the caller's stream.  Forked onto side streams -
      // normal or low priority - behind an event and joined again the launches were 0.5-2 %
      // SLOWER on configs 4 / 5: kernels of different code on one CU share its instruction cache)
      std::vector<size_t> order(pp->types.size());
      for (size_t t = 0; t < order.size(); ++t) order[t] = t;
      for (size_t oi = 0; oi < order.size(); ++oi) {
        const size_t t = order[oi];
        const fr::PieceType &pt = pp->types[t];
        fr::IssArgs b = a;
        b.recs = pt.d_recs;
        b.emit_rows = pt.d_emit_rows;
        b.piece_items = pt.d_items;
        b.piece_unit_begin = pt.d_unit_begin;
        b.piece_unit_row0 = pt.d_unit_row0;
        b.group_begin = nullptr;
        b.slot_rows = nullptr;
        b.group_row_begin = nullptr;
        b.shape_ids = nullptr;
        b.G = pt.units();
        b.xcd_map = (b.G > 1 && N % 8 == 0) ? 1 : 0;
        b.ops = pcs.d_ops_walk;
        b.feats = fu->walk_feats;
        b.feat_stride = F;
        b.carry_per_node = fu->carry_per_node;
        b.carry_slots = b.carry_per_node * pt.max_unit_nodes;
        b.carry_in_lds = 1;
        b.persistent = 0;
        b.nchunks = (int32_t)((T + chunk - 1) / chunk);
        const size_t other = piece_other_lds(p, pt, T, b.carry_per_node);
        bool fits = false;
        b.feat_window = piece_window(p, pt, T, b.carry_per_node, fu->n_ops, fu->has_mpi, fits);
        b.feat_fits = fits ? 1 : 0;
        if ((pcs.fits[t] != 0) != fits)   // (the kernel was compiled for exactly this: ensure_fused_pieces)
          return fail(FR_E_LIMIT, w + ": a piece kernel was compiled for another feature window");
        if (b.feat_window == 0)
          return fail(FR_E_LIMIT, w + ": the chunk carries and the features of one node do not fit the LDS");
        const size_t lds = other + fr::feat_window_bytes(b.feat_window, b.has_mpi != 0, false);
        hipError_t je = fr::jit_launch_fused(pcs.progs[t], b, lds, st);
        if (je != hipSuccess) return hip_fail(je, "fused walk (a piece type) launch");
      }
      *fu->walk_of_row = pcs.d_walk_of_row;
      return FR_OK;
    }
  }
  if (fu && fu->pl && !packed) {
    // the pipeline's run-time compiled kernel for this instantiation (fr_pipeline_prepare)
    const fr::FusedKey key = fused_key_for(p, T, fu->total_inc, fu->carry_per_node > 3);
    fr::JitProgram own{};
    {
      std::lock_guard<std::mutex> lock(fu->pl->jit_mu);
      auto st_it = fu->pl->jit_static.find((uint64_t)key.packed() | (uint64_t)gp.groups << 32);
      if (st_it != fu->pl->jit_static.end() && st_it->second.device == fr::current_device()) {
        own = st_it->second;   // (the plan as straight-line code, for exactly this group program)
      } else {
        auto it = fu->pl->jit.find(key.packed());
        if (it != fu->pl->jit.end()) own = it->second;
      }
    }
    if (own.fn != nullptr && own.device == fr::current_device()) {
      const int64_t chunk = fr::walk_chunk_elems(T);
      a.nchunks = (int32_t)((T + chunk - 1) / chunk);
      const size_t lds = ((size_t)a.R * chunk + 16 + 8 + (a.nchunks > 1 ? a.carry_slots : 0)) * 8 +
                         fr::feat_window_bytes(a.feat_window, a.has_mpi != 0, false);
      hipError_t je = fr::jit_launch_fused(own, a, lds, st);
      if (je != hipSuccess) return hip_fail(je, "fused walk (run-time compiled) launch");
      return FR_OK;
    }
  }
  hipError_t e = jit_prog ? fr::jit_launch(*jit_prog, a, st) : fr::launch_iss_walk(a, p.levels, st);
  if (e != hipSuccess) return hip_fail(e, "iss_walk launch");
  return FR_OK;
}

}  // namespace

extern "C" {

int fr_iss_run(fr_plan_t *plan, const double *d_X, int64_t N, int64_t D, int64_t T,
               const double *d_lookup, int64_t lookup_rows, double *d_out, int64_t out_k_stride,
               int64_t out_n_stride, void *d_work, int64_t work_bytes, int32_t groups,
               void *stream) {
  if (!plan || !plan->p) return fail(FR_E_ARG, "fr_iss_run: null plan");
  return run_walk("fr_iss_run", *plan->p, d_X, N, D, T, d_lookup, lookup_rows, d_out,
                  out_k_stride, out_n_stride, d_work, work_bytes, groups, (hipStream_t)stream,
                  nullptr);
}

fr_pipeline_t *fr_pipeline_create(fr_plan_t *plan, int32_t n_sieves, const int32_t *kinds,
                                  const int32_t *incs, const int32_t *C1, const int32_t *Q1,
                                  const int64_t *cuts, int64_t T) {
  if (!plan || !plan->p || n_sieves < 1 || !kinds || !incs || !C1 || !Q1 || !cuts || T < 1) {
    fail(FR_E_ARG, "fr_pipeline_create: bad argument");
    return nullptr;
  }
  fr_pipeline_t *pl = new fr_pipeline_t;
  pl->plan = plan;
  pl->T = T;
  int32_t col = 0, qoff = 0, n_ops = 0;
  for (int i = 0; i < n_sieves; ++i) {
    PipeSieve sv;
    sv.kind = kinds[i] & 0xff;
    sv.series_cuts = (kinds[i] & FR_SIEVE_SERIES_CUTS) != 0;
    sv.inc = incs[i];
    sv.Q1 = Q1[i];
    const int c1 = C1[i];
    std::string bad;
    if (sv.kind < 0 || sv.kind > 2 || c1 < 2) bad = "bad sieve " + std::to_string(i);
    int code = FR_E_ARG;
    if (bad.empty() && sv.kind != FR_SIEVE_END) {
      if (sv.Q1 < 2) bad = "a band sieve needs >= 2 thresholds";
      else if (sv.inc < -8 || sv.inc > 8) {
        bad = "the fused epilogue supports inc -8 to 8";
        code = FR_E_LIMIT;
      }
    }
    if (!bad.empty()) {
      fail(code, "fr_pipeline_create: " + bad);
      delete pl;
      return nullptr;
    }
    for (int j = 0; j < c1 && sv.series_cuts; ++j) {
      // the "cuts" of such a sieve are slots of the per-series table
      const int64_t c = *cuts++;
      if (c < 0 || c > 0xffff) {
        fail(FR_E_ARG, "fr_pipeline_create: bad cut slot " + std::to_string(c));
        delete pl;
        return nullptr;
      }
      sv.cuts.push_back((int32_t)c);
      pl->cut_slots_needed = std::max(pl->cut_slots_needed, (int32_t)c + 1);
    }
    for (int j = 0; j < c1 && !sv.series_cuts; ++j) {
      const int64_t c = *cuts++;
      // END reads X[:, c - 1] (index -1 wraps like numpy): the reference raises IndexError
      // outside [-T, T-1] (np.take_along_axis, fruits/sieving/segment.py:213-218)
      if (sv.kind == FR_SIEVE_END && j > 0 && (c - 1 < -T || c - 1 > T - 1)) {
        fail(FR_E_INDEX, "fr_pipeline_create: END cut " + std::to_string(c) +
                             " is out of bounds for series of length " + std::to_string(T));
        delete pl;
        return nullptr;
      }
      sv.cuts.push_back((int32_t)(c < 0 ? 0 : (c > T ? T : c)));
    }
    const int nf = sv.kind == FR_SIEVE_END ? c1 - 1 : (c1 - 1) * (sv.Q1 - 1);
    sv.col = col;
    sv.q_off = qoff;
    if (sv.kind == FR_SIEVE_MPI)
      for (int f = 0; f < nf; ++f) pl->mpi_cols.push_back(col + f);
    if (sv.kind != FR_SIEVE_END) qoff += sv.Q1;
    col += nf;
    n_ops += nf;
    pl->sieves.push_back(sv);
  }
  pl->per_sum = col;
  pl->q_stride = qoff > 0 ? qoff : 1;
  pl->n_ops = n_ops;
  pl->n_ops_padded = (n_ops + 1) / 2 * 2;
  return pl;
}

void fr_pipeline_destroy(fr_pipeline_t *pl) {
  if (!pl) return;
  {
    // (launches of the pipeline's own kernels may still run, a compilation may still be about to
    // store its result: wait for the device, and take the lock the compilations store under)
    std::lock_guard<std::mutex> lock(pl->jit_mu);
    ++pl->jit_gen;
    if (!pl->jit.empty() || !pl->jit_static.empty() || !pl->jit_pieces.empty()) (void)hipDeviceSynchronize();
  }
  if (pl->d_ops) (void)hipFree(pl->d_ops);
  if (pl->d_mpi_cols) (void)hipFree(pl->d_mpi_cols);
  if (pl->d_npi_pairs) (void)hipFree(pl->d_npi_pairs);
  if (pl->d_prep) (void)hipFree(pl->d_prep);
  if (pl->d_argmax_words) (void)hipFree(pl->d_argmax_words);
  for (auto &kv : pl->jit) fr::jit_unload(kv.second);
  for (auto &kv : pl->jit_static) fr::jit_unload(kv.second);
  pl->drop_pieces();
  delete pl;
}

int64_t fr_pipeline_info(const fr_pipeline_t *pl, int32_t what) {
  if (!pl) return fail(FR_E_ARG, "fr_pipeline_info: null pipeline");
  switch (what) {
    case 0: return pl->per_sum;
    case 1: return pl->q_stride;
    case 2: return (int64_t)pl->per_sum * pl->rows();
    case 3: {                                    // run-time compiled kernels loaded
      fr_pipeline *m = const_cast<fr_pipeline *>(pl);
      std::lock_guard<std::mutex> lock(m->jit_mu);
      return (int64_t)(m->jit.size() + m->jit_static.size());
    }
    case 4: {                                    // ... of them with the plan as straight-line code
      fr_pipeline *m = const_cast<fr_pipeline *>(pl);
      std::lock_guard<std::mutex> lock(m->jit_mu);
      return (int64_t)m->jit_static.size();
    }
    case 5: {                                    // kernels of piece types loaded (a plan in pieces)
      fr_pipeline *m = const_cast<fr_pipeline *>(pl);
      std::lock_guard<std::mutex> lock(m->jit_mu);
      int64_t n = 0;
      for (const auto &kv : m->jit_pieces) n += (int64_t)kv.second.progs.size();
      return n;
    }
    default: return fail(FR_E_ARG, "fr_pipeline_info: unknown selector");
  }
}

// The host half of fr_pipeline_set_quantiles: the table of feature ops (and the NPI / MPI pairs
// that share a population), nothing on the device.
static void build_pipeline_ops(fr_pipeline_t *pl, const double *h_quant, std::vector<fr::FeatOp> &ops,
                               std::vector<int32_t> &pairs) {
  const int K = pl->rows();
  ops.assign((size_t)K * pl->n_ops_padded, fr::FeatOp{});
  // An NPI feature whose band, cut and differencing order equal an MPI feature's is that
  // MPI op's population (experiments/fruit_reduced.py pairs NPI and MPI sieves with the
  // same arguments, fitted on the same values): such NPI ops are dropped from the walk
  // and filled in from the population table by mpi_finalize_kernel.  Only when the same
  // pairs match in every row (the finalize kernel works on column patterns).
  pairs.clear();
  for (int pass = 0; pass < 2; ++pass) {
    const bool merge = pass == 0;
    bool uniform = true;
    int n_ops_eff = 0;
    for (int k = 0; k < K && uniform; ++k) {
      std::vector<fr::FeatOp> row;
      for (const PipeSieve &sv : pl->sieves) {
        const int C = (int)sv.cuts.size() - 1;
        if (sv.kind == FR_SIEVE_END) {
          for (int j = 0; j < C; ++j) {
            if (sv.series_cuts) {   // the kernel reads cut_row[slot] - 1 (and wraps -1)
              row.push_back(fr::FeatOp{FR_SIEVE_END | (1 << 16), k * pl->per_sum + sv.col + j,
                                       sv.cuts[j + 1], 0, 0.0, 0.0});
              continue;
            }
            int idx = sv.cuts[j + 1] - 1;
            if (idx < 0) idx += (int)pl->T;  // numpy's wrap of index -1 (segment.py:213-218)
            row.push_back(fr::FeatOp{FR_SIEVE_END | (1 << 20), k * pl->per_sum + sv.col + j, idx, 0, 0.0, 0.0});
          }
          continue;
        }
        const double *q = h_quant + (size_t)k * pl->q_stride + sv.q_off;
        for (int j = 0; j < C; ++j)
          for (int b = 0; b + 1 < sv.Q1; ++b) {
            // the common shapes get a short path in the fused walk (walk_fused.h, OPF_SHAPE_*):
            // a counting band over the whole series, of the values or their first differences,
            // with or without an upper threshold
            int32_t flags = sv.series_cuts ? (1 << 16) : 0;
            if (sv.kind == FR_SIEVE_NPI && !sv.series_cuts && sv.cuts[j] <= 0 &&
                sv.cuts[j + 1] >= pl->T && (sv.inc == 0 || sv.inc == 1)) {
              const bool no_hi = q[b + 1] == std::numeric_limits<double>::infinity();
              flags |= ((no_hi ? 2 : 4) | sv.inc) << 20;
            }
            row.push_back(fr::FeatOp{sv.kind | ((sv.inc & 0xff) << 8) | flags,
                                     k * pl->per_sum + sv.col + j * (sv.Q1 - 1) + b, sv.cuts[j],
                                     sv.cuts[j + 1], q[b], q[b + 1]});
          }
      }
      std::vector<int32_t> row_pairs;
      std::vector<char> drop(row.size(), 0), used(row.size(), 0);
      if (merge && !pl->mpi_cols.empty()) {
        for (size_t i = 0; i < row.size(); ++i) {
          if ((row[i].kind_inc & 0xff) != FR_SIEVE_NPI) continue;
          for (size_t m = 0; m < row.size(); ++m) {
            if ((row[m].kind_inc & 0xff) != FR_SIEVE_MPI || used[m]) continue;
            // (same differencing order and kind of cuts; the shape bits above are the walk's)
            if ((((row[m].kind_inc ^ row[i].kind_inc) >> 8) & 0x1ff) == 0 && row[m].lo == row[i].lo &&
                row[m].hi == row[i].hi && std::memcmp(&row[m].qlo, &row[i].qlo, 8) == 0 &&
                std::memcmp(&row[m].qhi, &row[i].qhi, 8) == 0) {
              drop[i] = used[m] = 1;
              row_pairs.push_back(row[i].col - k * pl->per_sum);
              row_pairs.push_back(row[m].col - k * pl->per_sum);
              break;
            }
          }
        }
      }
      if (k == 0) pairs = row_pairs;
      else if (row_pairs != pairs) uniform = false;
      fr::FeatOp *o = ops.data() + (size_t)k * pl->n_ops_padded;
      int i = 0;
      for (size_t j = 0; j < row.size(); ++j)
        if (!drop[j]) o[i++] = row[j];
      n_ops_eff = std::max(n_ops_eff, i);
      for (; i < pl->n_ops_padded; ++i)  // padding op: an END that never matches a chunk
        o[i] = fr::FeatOp{FR_SIEVE_END, 0, -(1 << 30), 0, 0.0, 0.0};
    }
    pl->n_ops_eff = n_ops_eff;
    if (uniform) break;
    pairs.clear();  // rows disagree: second pass without merging
  }
}

// What a run-time compiled kernel takes as immediates: per op kind | differencing order | shape
// and the cuts - the same for every output row (the shape only if every row's thresholds agree
// on it: an infinite threshold in one row alone keeps the generic band).  Caller holds jit_mu.
static void set_pipeline_jit_ops(fr_pipeline_t *pl, const std::vector<fr::FeatOp> &ops) {
  const int K = pl->rows();
  pl->jit_ops = fr::FusedOps{};
  pl->jit_ops.n_padded = pl->n_ops_padded;
  pl->jit_ops.full_chunks = pl->T % fr::walk_chunk_elems(pl->T) == 0;
  for (const PipeSieve &sv : pl->sieves) {   // (a slot pair per differencing order >= 3 and per
    if (sv.kind == FR_SIEVE_END) continue;    // cumulation of a row: walk_fused.h, fop)
    if (sv.inc > 2) pl->jit_ops.cps = std::max(pl->jit_ops.cps, 3 + 2 * (sv.inc - 2));
    if (sv.inc < 0) pl->jit_ops.cps = std::max(pl->jit_ops.cps, 15 + 2 * (-sv.inc));
  }
  pl->jit_uniform = K > 0 && pl->n_ops_eff > 0 && pl->cut_slots_needed == 0;
  for (int i = 0; i < pl->n_ops_eff && pl->jit_uniform; ++i) {
    int32_t w0 = ops[i].kind_inc;
    for (int k = 1; k < K; ++k) {
      const fr::FeatOp &o = ops[(size_t)k * pl->n_ops_padded + i];
      if (o.kind_inc != w0) {
        if (((o.kind_inc ^ w0) & ~(7 << 20)) != 0) pl->jit_uniform = false;
        w0 &= ~(7 << 20);
      }
      if (o.lo != ops[i].lo || o.hi != ops[i].hi) pl->jit_uniform = false;
    }
    pl->jit_ops.w0.push_back(w0);
    pl->jit_ops.lo.push_back(ops[i].lo);
    pl->jit_ops.hi.push_back(ops[i].hi);
  }
}

int fr_pipeline_set_argmax(fr_pipeline_t *pl, int32_t n_words, const int32_t *lengths) {
  if (!pl || !pl->plan || !pl->plan->p || n_words < 1 || !lengths)
    return fail(FR_E_ARG, "fr_pipeline_set_argmax: bad argument");
  const fr::Plan &p = *pl->plan->p;
  if (!p.letter_sum || p.semiring != fr::kSemiArctic || p.cos)
    return fail(FR_E_ARG, "fr_pipeline_set_argmax: the plan must be an Arctic letter-sum plan "
                          "(FR_PLAN_ARCTIC | FR_PLAN_LETTER_SUM) with every prefix of every word as a row");
  if (pl->have_quantiles)
    return fail(FR_E_ARG, "fr_pipeline_set_argmax: call it before fr_pipeline_set_quantiles");
  std::vector<int32_t> words;
  int64_t v0 = 0, o0 = 0;
  int max_len = 0;
  for (int w = 0; w < n_words; ++w) {
    const int L = lengths[w];
    if (L < 1 || L > 63) return fail(FR_E_LIMIT, "fr_pipeline_set_argmax: words of 1 to 63 letters");
    words.insert(words.end(), {(int32_t)v0, L, (int32_t)o0, 0});
    v0 += L;
    o0 += L + L * (L + 1) / 2;
    max_len = std::max(max_len, L);
  }
  if (v0 != p.K)
    return fail(FR_E_ARG, "fr_pipeline_set_argmax: the plan has " + std::to_string(p.K) +
                              " rows, the words' prefixes are " + std::to_string(v0));
  if (o0 > 0x7fffffffLL / std::max(1, pl->per_sum) || n_words > 65535)
    return fail(FR_E_LIMIT, "fr_pipeline_set_argmax: too many rows");
  for (const PipeSieve &sv : pl->sieves)
    if (sv.kind != FR_SIEVE_END && (sv.inc < 0 || sv.inc > 2))
      return fail(FR_E_LIMIT, "fr_pipeline_set_argmax: differencing orders 0 to 2");
  if (pl->T > 65535 || fr::argmax_sieve_lds(pl->T, max_len) > fr::kArgmaxSieveLds)
    return fail(FR_E_LIMIT, "fr_pipeline_set_argmax: a row of maxima and the positions of a word's "
                            "prefixes must fit a workgroup's LDS");
  if (pl->d_argmax_words) (void)hipFree(pl->d_argmax_words);
  pl->d_argmax_words = nullptr;
  HIP_TRY(hipMalloc(&pl->d_argmax_words, words.size() * 4));
  HIP_TRY(hipMemcpy(pl->d_argmax_words, words.data(), words.size() * 4, hipMemcpyHostToDevice));
  pl->argmax_words = words;
  pl->argmax_rows = (int32_t)o0;
  pl->argmax_max_len = max_len;
  return FR_OK;
}

int fr_pipeline_set_quantiles(fr_pipeline_t *pl, const double *h_quant) {
  if (!pl || !h_quant) return fail(FR_E_ARG, "fr_pipeline_set_quantiles: bad argument");
  std::vector<fr::FeatOp> ops;
  std::vector<int32_t> pairs;
  build_pipeline_ops(pl, h_quant, ops, pairs);
  if (pairs != pl->npi_pairs || (!pairs.empty() && !pl->d_npi_pairs)) {
    if (pl->d_npi_pairs) (void)hipFree(pl->d_npi_pairs);
    pl->d_npi_pairs = nullptr;
    pl->npi_pairs = pairs;
    if (!pairs.empty()) {
      HIP_TRY(hipMalloc(&pl->d_npi_pairs, pairs.size() * 4));
      HIP_TRY(hipMemcpy(pl->d_npi_pairs, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice));
    }
  }
  // (new thresholds: the kernels compiled for the old ops go - once the launches that may still
  // be running them are done: fr_pipeline_run returns without a synchronisation)
  std::lock_guard<std::mutex> jit_lock(pl->jit_mu);
  if (!pl->jit.empty() || !pl->jit_static.empty() || !pl->jit_pieces.empty()) (void)hipDeviceSynchronize();
  ++pl->jit_gen;
  for (auto &kv : pl->jit) fr::jit_unload(kv.second);
  for (auto &kv : pl->jit_static) fr::jit_unload(kv.second);
  pl->jit.clear();
  pl->jit_static.clear();
  pl->jit_static_tried.clear();
  pl->jit_failed.clear();
  pl->drop_pieces();
  set_pipeline_jit_ops(pl, ops);
  const size_t bytes = ops.size() * sizeof(fr::FeatOp);
  pl->h_ops = ops;
  if (!pl->d_ops && bytes) HIP_TRY(hipMalloc(&pl->d_ops, bytes));
  if (bytes) HIP_TRY(hipMemcpy(pl->d_ops, ops.data(), bytes, hipMemcpyHostToDevice));
  if (!pl->mpi_cols.empty() && !pl->d_mpi_cols) {
    HIP_TRY(hipMalloc(&pl->d_mpi_cols, pl->mpi_cols.size() * 4));
    HIP_TRY(hipMemcpy(pl->d_mpi_cols, pl->mpi_cols.data(), pl->mpi_cols.size() * 4,
                      hipMemcpyHostToDevice));
  }
  pl->have_quantiles = true;
  return FR_OK;
}

int64_t fr_pipeline_workspace_bytes(const fr_pipeline_t *pl, int64_t N, int64_t lookup_rows) {
  if (!pl || N < 0) return fail(FR_E_ARG, "fr_pipeline_workspace_bytes: bad argument");
  const fr::Plan &p = *pl->plan->p;
  size_t b = align_up(work_layout(p, N, pl->T, p.weighting ? lookup_rows : 0).total(), 256);
  if (!pl->mpi_cols.empty()) b += align_up((size_t)N * pl->per_sum * pl->rows() * 8, 256);
  // (argmax: the running maxima of the plan's rows are materialised, the argmax rows are not)
  if (!pl->argmax_words.empty()) b += align_up((size_t)p.K * N * pl->T * 8, 256);
  if (pl->prep_n > 0 && pl->prep_std != 0) b += align_up((size_t)N * pl->prep_n * 16, 256);
  // (a plan in pieces leaves its features in walk order first)
  if (pieces_eligible(*pl)) b += align_up((size_t)N * pl->per_sum * p.K * 8, 256);
  return (int64_t)b;
}

int fr_pipeline_set_preparation(fr_pipeline_t *pl, int32_t D, int32_t inc_lag, int32_t as_new,
                                int32_t standardize, double std_eps) {
  if (!pl || !pl->plan || !pl->plan->p || D < 1 || inc_lag < 0 || standardize < 0 ||
      standardize > 2 || (as_new && inc_lag < 1))
    return fail(FR_E_ARG, "fr_pipeline_set_preparation: bad argument");
  const fr::Plan &p = *pl->plan->p;
  if (pl->d_prep) (void)hipFree(pl->d_prep);
  pl->d_prep = nullptr;
  pl->prep_D = pl->prep_n = pl->prep_std = 0;
  if (inc_lag == 0 && standardize == 0) return FR_OK;   // nothing to fuse
  if (!pl->argmax_words.empty())
    return fail(FR_E_LIMIT, "fr_pipeline_set_preparation: an argmax pipeline materialises the running "
                            "maxima from the prepared input");
  // (every fused kernel forms the prepared rows itself since round 4: the cooperative walk in its
  // staging, the wave-per-series kernels in theirs, CosWISS where it reads a letter's rows - all
  // but a CosWISS with the randomised ffn, whose units read transformed copies of the input)
  if (p.cos && p.cos->x_unit_stride != 0)
    return fail(FR_E_LIMIT, "fr_pipeline_set_preparation: a CosWISS with per-unit inputs (ffn) reads "
                            "transformed copies of the prepared input");
  const int n_prep = as_new ? 2 * D : D;
  std::vector<int32_t> tab((size_t)n_prep * 4, 0);
  for (int d = 0; d < n_prep; ++d) {
    const bool inc_row = as_new ? d >= D : inc_lag > 0;
    tab[4 * d] = as_new ? d % D : d;
    tab[4 * d + 1] = inc_row ? inc_lag : 0;
    tab[4 * d + 2] = standardize != 0 ? 1 : 0;
  }
  HIP_TRY(hipMalloc(&pl->d_prep, tab.size() * 4));
  hipError_t e = hipMemcpy(pl->d_prep, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(pl->d_prep);
    pl->d_prep = nullptr;
    return hip_fail(e, "hipMemcpy(preparation table)");
  }
  pl->prep_D = D;
  pl->prep_n = n_prep;
  pl->prep_std = standardize;
  pl->prep_eps = std_eps;
  return FR_OK;
}

// The kernel instantiation a fused launch of this pipeline over N series takes, or false when
// it has none of its own (CosWISS, wave-per-series kernels, letter sums, nothing fits).
static bool fused_instance_of(fr_pipeline_t *pl, int64_t N, int32_t groups, fr::FusedKey &key,
                              LaunchShape &shape) {
  fr::Plan &p = *pl->plan->p;
  if (p.cos || N <= 0 || env_int("FRUITS_HIP_JIT", 1) == 0 || p.letter_sum) return false;
  bool total_inc = false;
  for (const PipeSieve &sv : pl->sieves)
    if (sv.kind != FR_SIEVE_END && sv.inc >= 1) total_inc = true;
  shape = launch_shape(p, N, pl->T, groups);
  const bool packed = shape.packed && !(total_inc && p.weighting == FR_W_TOTAL);
  if (packed || !shape.fits) return false;
  key = fused_key_for(p, pl->T, total_inc, pl->jit_ops.cps > 3);
  return true;
}

static int pipeline_compile_plan(fr_pipeline_t *pl, int64_t N, int32_t groups, bool cache_only);

static int pipeline_prepare(fr_pipeline_t *pl, int64_t N, int32_t groups, bool cache_only) {
  if (!pl || !pl->plan || !pl->plan->p || N < 0)
    return fail(FR_E_ARG, "fr_pipeline_prepare: bad argument");
  if (!pl->have_quantiles)
    return fail(FR_E_ARG, "fr_pipeline_prepare: call fr_pipeline_set_quantiles first");
  fr::Plan &p = *pl->plan->p;
  if (!pl->argmax_words.empty())   // (the plan runs as a materialising walk, the sieves in a kernel of the library)
    return prepare_plan(p, N, pl->T, groups, false, "fr_pipeline_prepare");
  int rc = prepare_plan(p, N, pl->T, groups, true, "fr_pipeline_prepare");
  if (rc != FR_OK) return rc;
  // The pipeline's own kernel: the fused walk with the sieves as compile-time constants (hipRTC,
  // 1-2 s once per pipeline shape, cached on disk); a failure is not the caller's - the generic
  // kernel runs the pipeline.  Not for the wave-per-series kernels (T <= 384) and CosWISS.
  fr::FusedKey key;
  LaunchShape shape;
  if (fused_instance_of(pl, N, groups, key, shape)) {
    // (a large plan runs in pieces: their tables go up here, on the caller's thread)
    if (pieces_eligible(*pl)) {
      rc = ensure_pieces_tables(*pl, key, "fr_pipeline_prepare");
      if (rc != FR_OK) return rc;
    }
    ensure_fused_jit(*pl, key, cache_only);
  }
  return FR_OK;
}

int fr_pipeline_prepare(fr_pipeline_t *pl, int64_t N, int32_t groups) {
  return pipeline_prepare(pl, N, groups, false);
}

int fr_pipeline_compile_plan(fr_pipeline_t *pl, int64_t N, int32_t groups) {
  return pipeline_compile_plan(pl, N, groups, false);
}

int fr_pipeline_prepare_cached(fr_pipeline_t *pl, int64_t N, int32_t groups) {
  int rc = pipeline_prepare(pl, N, groups, true);
  return rc != FR_OK ? rc : pipeline_compile_plan(pl, N, groups, true);
}

static int pipeline_compile_plan(fr_pipeline_t *pl, int64_t N, int32_t groups, bool cache_only) {
  if (!pl || !pl->plan || !pl->plan->p || N < 0)
    return fail(FR_E_ARG, "fr_pipeline_compile_plan: bad argument");
  if (!pl->have_quantiles)
    return fail(FR_E_ARG, "fr_pipeline_compile_plan: call fr_pipeline_set_quantiles first");
  fr::Plan &p = *pl->plan->p;
  fr::FusedKey key;
  LaunchShape shape;
  if (!fused_instance_of(pl, N, groups, key, shape) || debug_knob("fused_static", 1) == 0) return FR_OK;
  // a large plan: in pieces, every piece type straight-line code in a kernel of its own; the
  // node shapes below only where the plan has no such cover
  if (pieces_eligible(*pl)) {
    if (!cache_only) {
      int rc = ensure_pieces_tables(*pl, key, "fr_pipeline_compile_plan");
      if (rc != FR_OK) return rc;
    }
    ensure_fused_pieces(*pl, key, cache_only);
    std::lock_guard<std::mutex> lock(pl->jit_mu);
    auto it = pl->jit_pieces.find(key.packed());
    if (it != pl->jit_pieces.end() && !it->second.progs.empty()) return FR_OK;
    if (cache_only) return FR_OK;   // (not in the cache: the loop kernels' cached variants are not looked up either)
  }
  // for the group program a launch over N series will pick (another group count at run time
  // simply takes the kernel of fr_pipeline_prepare)
  const int asked = groups > 0 ? groups : debug_knob("groups", 0);
  fr::FusedPlan fp;
  {
    std::lock_guard<std::mutex> lock(p.mu);
    const int G = asked > 0 ? shape.G
                            : choose_groups_walk(p, N, pl->T, query_resident(p, N, pl->T, true, true), true);
    const fr::GroupedProgram &gp = fr::grouped(p, G);
    if ((int)p.nodes.size() <= fr::kFusedStaticMaxNodes) {
      // a small plan: the records themselves (straight-line code)
      fp.w.reserve(gp.recs.size() * 16);
      for (const fr::NodeRec &r : gp.recs) fp.w.insert(fp.w.end(), r.w, r.w + 16);
      fp.group_begin.assign(gp.group_begin.begin(), gp.group_begin.begin() + gp.groups);
    } else {
      // a large one: its most frequent node shapes (the loop stays, the bodies are compiled
      // per shape; the rest takes the generic body)
      const size_t n = std::min<size_t>(gp.shapes.size(), fr::kFusedShapes);
      fp.shapes.assign(gp.shapes.begin(), gp.shapes.begin() + n);
      fp.n_groups = gp.groups;
    }
  }
  ensure_fused_static(*pl, key, fp, cache_only);
  return FR_OK;
}

int32_t fr_pipeline_bundle(fr_pipeline_t *pl, const double *h_quant, int32_t groups, const char *dir,
                           char *msg, int64_t msg_cap) {
  if (msg && msg_cap > 0) msg[0] = 0;
  if (!pl || !pl->plan || !pl->plan->p || !h_quant || !dir || !*dir)
    return fail(FR_E_ARG, "fr_pipeline_bundle: bad argument");
  fr::Plan &p = *pl->plan->p;
  std::vector<fr::FeatOp> ops;
  std::vector<int32_t> pairs;
  build_pipeline_ops(pl, h_quant, ops, pairs);
  fr::FusedOps jops;
  {
    std::lock_guard<std::mutex> lock(pl->jit_mu);
    set_pipeline_jit_ops(pl, ops);
    if (!pl->jit_uniform) return 0;
    jops = pl->jit_ops;
  }
  fr::FusedKey key;
  LaunchShape shape;
  if (!fused_instance_of(pl, 1 << 20, groups, key, shape)) return 0;
  std::vector<std::string> errs;
  std::atomic<int> done{0};
  std::mutex err_mu;
  auto one = [&](const fr::FusedOps &o, const fr::FusedKey &k, const fr::FusedPlan *fp) {
    std::string err;
    if (fr::jit_fused_into(o, k, fp, dir, err)) {
      ++done;
    } else {
      std::lock_guard<std::mutex> lock(err_mu);
      errs.push_back(err);
    }
  };
  if (pieces_eligible(*pl)) {                  // a large plan: a kernel per piece type
    const fr::PiecedProgram *pp;
    {
      std::lock_guard<std::mutex> lock(p.mu);
      pp = &fr::pieced(p, piece_nodes_knob(*pl), debug_knob("piece_unit", 0));
    }
    const int n_types = pp->ok ? (int)pp->types.size() : 0;
    // the largest bodies first (the compiler's time grows faster than a body); job -1: the kernel
    // with the sieves as immediates alone (the plan from its records)
    std::vector<int> order{-1};
    for (int t = 0; t < n_types; ++t) order.push_back(t);
    std::stable_sort(order.begin() + 1, order.end(), [&](int x, int y) {
      return pp->types[x].body_nodes > pp->types[y].body_nodes;
    });
    std::atomic<int> next{0};
    auto worker = [&] {
      for (int i = next++; i < (int)order.size(); i = next++) {
        const int t = order[i];
        if (t < 0) {
          one(jops, key, nullptr);
          continue;
        }
        fr::FusedPlan fp;
        fp.w = pp->types[t].body_w;
        fp.piece = true;
        fr::FusedKey k = key;
        k.LV = piece_level_variant(pp->types[t].levels);
        fr::FusedOps type_ops = jops;
        bool fits = false;
        (void)piece_window(p, pp->types[t], pl->T, jops.cps, pl->n_ops_eff, !pl->mpi_cols.empty(), fits);
        type_ops.window_fits = fits;
        one(type_ops, k, &fp);
      }
    };
    const int n_threads = std::max(1, std::min((int)order.size(), env_int("FRUITS_BUNDLE_THREADS", 4)));
    std::vector<std::thread> pool;
    for (int i = 1; i < n_threads; ++i) pool.emplace_back(worker);
    worker();
    for (std::thread &th : pool) th.join();
  } else if ((int)p.nodes.size() <= fr::kFusedStaticMaxNodes && debug_knob("fused_static", 1) != 0) {
    // a small plan as straight-line code, for the group program of `groups` groups per series
    std::thread sieves_only([&] { one(jops, key, nullptr); });
    fr::FusedPlan fp;
    {
      std::lock_guard<std::mutex> lock(p.mu);
      const fr::GroupedProgram &gp = fr::grouped(p, groups > 0 ? groups : 1);
      fp.w.reserve(gp.recs.size() * 16);
      for (const fr::NodeRec &r : gp.recs) fp.w.insert(fp.w.end(), r.w, r.w + 16);
      fp.group_begin.assign(gp.group_begin.begin(), gp.group_begin.begin() + gp.groups);
    }
    one(jops, key, &fp);
    sieves_only.join();
  } else {
    one(jops, key, nullptr);
  }
  if (!errs.empty()) {
    if (msg && msg_cap > 0) snprintf(msg, (size_t)msg_cap, "%s", errs[0].c_str());
    return fail(FR_E_LIMIT, "fr_pipeline_bundle: " + errs[0]);
  }
  return done.load();
}

int fr_pipeline_set_series_cuts(fr_pipeline_t *pl, const int32_t *d_cuts, int64_t N, int32_t slots) {
  if (!pl || N < 0 || slots < 0 || (N * slots > 0 && !d_cuts))
    return fail(FR_E_ARG, "fr_pipeline_set_series_cuts: bad argument");
  if (slots < pl->cut_slots_needed)
    return fail(FR_E_ARG, "fr_pipeline_set_series_cuts: the sieves name " +
                              std::to_string(pl->cut_slots_needed) + " slots, the table has " +
                              std::to_string(slots));
  pl->d_series_cuts = d_cuts;
  pl->cuts_N = N;
  pl->cut_slots = slots;
  return FR_OK;
}

int fr_pipeline_run(fr_pipeline_t *pl, const double *d_X, int64_t N, int64_t D, int64_t T,
                    const double *d_lookup, int64_t lookup_rows, double *d_feats,
                    int64_t feat_stride, void *d_work, int64_t work_bytes, int32_t groups,
                    void *stream) {
  if (!pl || !pl->plan || !pl->plan->p) return fail(FR_E_ARG, "fr_pipeline_run: null pipeline");
  fr::Plan &p = *pl->plan->p;
  if (T != pl->T) return fail(FR_E_ARG, "fr_pipeline_run: pipeline was created for another T");
  if (!pl->have_quantiles)
    return fail(FR_E_ARG, "fr_pipeline_run: call fr_pipeline_set_quantiles first");
  const int64_t F = (int64_t)pl->per_sum * pl->rows();
  if (N == 0 || F == 0) return FR_OK;
  if (!d_feats || feat_stride < F) return fail(FR_E_ARG, "fr_pipeline_run: bad feature buffer");
  const int64_t need = fr_pipeline_workspace_bytes(pl, N, lookup_rows);
  if (need > 0 && (!d_work || work_bytes < need))
    return fail(FR_E_NOMEM, "fr_pipeline_run: workspace too small (need " +
                                std::to_string(need) + " bytes)");
  hipStream_t st = (hipStream_t)stream;
  const size_t plan_ws = align_up(work_layout(p, N, T, p.weighting ? lookup_rows : 0).total(), 256);
  FusedArgs fu;
  fu.pl = pl;
  fu.ops = static_cast<const fr::FeatOp *>(pl->d_ops);
  fu.feats = d_feats;
  fu.feat_stride = feat_stride;
  fu.n_ops = pl->n_ops_eff;
  fu.n_ops_padded = pl->n_ops_padded;
  for (const PipeSieve &sv : pl->sieves)
    if (sv.kind != FR_SIEVE_END && sv.inc >= 1) fu.total_inc = true;
  fu.carry_per_node = pl->jit_ops.cps;
  if (pl->cut_slots_needed > 0) {
    if (!pl->d_series_cuts || pl->cuts_N != N || pl->cut_slots < pl->cut_slots_needed)
      return fail(FR_E_ARG, "fr_pipeline_run: a sieve has per-series cuts - call "
                            "fr_pipeline_set_series_cuts with a table for these " +
                                std::to_string(N) + " series first");
    fu.series_cuts = pl->d_series_cuts;
    fu.cut_slots = pl->cut_slots;
  }
  // (the population table of MPI features shares the feature row stride)
  if (!pl->mpi_cols.empty() && feat_stride != F)
    return fail(FR_E_ARG, "fr_pipeline_run: MPI needs feat_stride == F");
  if (N < 0 || D < 1 || !d_X) return fail(FR_E_ARG, "fr_pipeline_run: bad input");
  if (pl->prep_n > 0 && D != pl->prep_D)
    return fail(FR_E_ARG, "fr_pipeline_run: the fused preparation was set for " +
                              std::to_string(pl->prep_D) + " raw dimensions, the input has " +
                              std::to_string(D));
  const int64_t D_words = pl->prep_n > 0 ? pl->prep_n : D;
  if (p.max_dim > D_words)
    return fail(FR_E_DIM, "fr_pipeline_run: a word references dimension " +
                              std::to_string(p.max_dim) + " but the input has only " +
                              std::to_string(D_words));
  if (p.weighting != 0 && !p.cos && (!d_lookup || (lookup_rows != 1 && lookup_rows != N)))
    return fail(FR_E_ARG, "fr_pipeline_run: weighted plan needs a lookup of 1 or N rows");
  // (every feature column - and population entry - is written exactly once by the unit that
  // owns it: nothing to clear)
  if (!pl->mpi_cols.empty()) {
    fu.cnt = reinterpret_cast<double *>(static_cast<char *>(d_work) + plan_ws);
    fu.has_mpi = true;
  } else {
    fu.cnt = d_feats;  // never touched without MPI sieves
  }
  if (pl->prep_n > 0) {
    fu.prep = static_cast<const int32_t *>(pl->d_prep);
    fu.n_prep = pl->prep_n;
    if (pl->prep_std != 0) {
      // STD's statistics of the prepared rows: a small pre-pass over the raw input
      size_t off = plan_ws;
      if (!pl->mpi_cols.empty()) off += align_up((size_t)N * F * 8, 256);
      double *stats = reinterpret_cast<double *>(static_cast<char *>(d_work) + off);
      hipError_t e = fr::launch_row_stats(d_X, N, D, T, fu.prep, pl->prep_n,
                                          pl->prep_std == 2 ? 1 : 0, pl->prep_eps, stats, st);
      if (e != hipSuccess) return hip_fail(e, "row_stats launch");
      fu.stats = stats;
    }
  }
  if (!pl->argmax_words.empty()) {
    // Arctic argmax: the running maxima of every prefix (the plan's rows) as a (K, N, T) block of
    // the workspace, then ONE kernel that forms the argmax rows and their features
    size_t off = plan_ws;
    if (!pl->mpi_cols.empty()) off += align_up((size_t)N * F * 8, 256);
    double *V = reinterpret_cast<double *>(static_cast<char *>(d_work) + off);
    int rc = run_walk("fr_pipeline_run", p, d_X, N, D, T, d_lookup, lookup_rows, V, N * T, T, d_work,
                      (int64_t)plan_ws, groups, st, nullptr);
    if (rc != FR_OK) return rc;
    hipError_t e = fr::launch_argmax_sieves(V, N, T, pl->d_argmax_words, (int)pl->argmax_words.size() / 4,
                                            pl->argmax_max_len, fu.ops, fu.n_ops, fu.n_ops_padded, d_feats,
                                            fu.cnt, feat_stride, fu.series_cuts, fu.cut_slots, st);
    if (e != hipSuccess) return hip_fail(e, "argmax_sieves launch");
    if (!pl->mpi_cols.empty()) {
      e = fr::launch_mpi_finalize(d_feats, fu.cnt, N, feat_stride,
                                  static_cast<const int32_t *>(pl->d_mpi_cols), (int)pl->mpi_cols.size(),
                                  static_cast<const int32_t *>(pl->d_npi_pairs),
                                  (int)pl->npi_pairs.size() / 2, pl->per_sum, pl->rows(), st);
      if (e != hipSuccess) return hip_fail(e, "mpi_finalize launch");
    }
    return FR_OK;
  }
  const int32_t *walk_of_row = nullptr;
  if (pieces_eligible(*pl)) {
    size_t off = plan_ws;
    if (!pl->mpi_cols.empty()) off += align_up((size_t)N * F * 8, 256);
    if (pl->prep_n > 0 && pl->prep_std != 0) off += align_up((size_t)N * pl->prep_n * 16, 256);
    fu.walk_feats = reinterpret_cast<double *>(static_cast<char *>(d_work) + off);
    fu.walk_of_row = &walk_of_row;
  }
  int rc = run_walk("fr_pipeline_run", p, d_X, N, D, T, d_lookup, lookup_rows, nullptr, 0, 0,
                    d_work, (int64_t)plan_ws, groups, st, &fu);
  if (rc != FR_OK) return rc;
  if (walk_of_row != nullptr) {
    // the features are in walk order (plan.h, PiecedProgram): band means there, then the blocks
    // of every iterated sum to their columns
    if (!pl->mpi_cols.empty()) {
      hipError_t e = fr::launch_mpi_finalize(fu.walk_feats, fu.cnt, N, F,
                                             static_cast<const int32_t *>(pl->d_mpi_cols),
                                             (int)pl->mpi_cols.size(),
                                             static_cast<const int32_t *>(pl->d_npi_pairs),
                                             (int)pl->npi_pairs.size() / 2, pl->per_sum, p.K, st);
      if (e != hipSuccess) return hip_fail(e, "mpi_finalize launch");
    }
    hipError_t e = fr::launch_gather_row_blocks(fu.walk_feats, d_feats, N, F, feat_stride, p.K,
                                                pl->per_sum, walk_of_row, st);
    if (e != hipSuccess) return hip_fail(e, "gather_row_blocks launch");
    return FR_OK;
  }
  if (!pl->mpi_cols.empty()) {
    hipError_t e = fr::launch_mpi_finalize(d_feats, fu.cnt, N, feat_stride,
                                           static_cast<const int32_t *>(pl->d_mpi_cols),
                                           (int)pl->mpi_cols.size(),
                                           static_cast<const int32_t *>(pl->d_npi_pairs),
                                           (int)pl->npi_pairs.size() / 2, pl->per_sum, p.K, st);
    if (e != hipSuccess) return hip_fail(e, "mpi_finalize launch");
  }
  return FR_OK;
}

int fr_iterated_sum_fast_host(const double *h_Z, int64_t N, int64_t D, int64_t T,
                              const int32_t *word, int32_t L, int32_t Dw, const float *alpha,
                              const double *h_lookup, int64_t extended, int32_t total_weighting,
                              double *h_out) {
  if (!h_Z || !word || !h_out || N < 0 || D < 1 || T < 0 || L < 1 || Dw < 1)
    return fail(FR_E_ARG, "fr_iterated_sum_fast_host: bad argument");
  if (extended < 1 || extended > L)
    return fail(FR_E_ARG, "fr_iterated_sum_fast_host: extended must be in [1, L]");
  const int weighting = h_lookup ? ((total_weighting & 1) ? FR_W_TOTAL : FR_W_NONTOTAL) : FR_W_NONE;
  const int plan_flags = (total_weighting & 2) ? FR_PLAN_ARCTIC
                                               : ((total_weighting & 4) ? FR_PLAN_BAYESIAN : 0);
  if (weighting != FR_W_NONE && !alpha)
    return fail(FR_E_ARG, "fr_iterated_sum_fast_host: weighted call needs alpha");
  const int32_t depth = (int32_t)extended;
  fr_plan_t *plan = fr_plan_create(1, word, &L, &Dw, alpha, &depth, weighting, plan_flags);
  if (!plan) return FR_E_ARG;
  int rc = FR_OK;
  void *dZ = nullptr, *dL = nullptr, *dO = nullptr, *dW = nullptr;
  const int64_t zb = N * D * T * 8, lb = h_lookup ? N * T * 8 : 0, ob = N * extended * T * 8;
  const int64_t wb = fr_plan_workspace_bytes(plan, N, T, h_lookup ? N : 0);
  do {
    if (N == 0 || T == 0) break;
    if ((rc = fr_malloc(&dZ, zb)) != FR_OK) break;
    if ((rc = fr_malloc(&dO, ob)) != FR_OK) break;
    if (lb && (rc = fr_malloc(&dL, lb)) != FR_OK) break;
    if (wb && (rc = fr_malloc(&dW, wb)) != FR_OK) break;
    if ((rc = fr_memcpy_h2d(dZ, h_Z, zb, nullptr)) != FR_OK) break;
    if (lb && (rc = fr_memcpy_h2d(dL, h_lookup, lb, nullptr)) != FR_OK) break;
    rc = fr_iss_run(plan, (const double *)dZ, N, D, T, (const double *)dL, h_lookup ? N : 0,
                    (double *)dO, /*k stride*/ T, /*n stride*/ extended * T, dW, wb, 0, nullptr);
    if (rc != FR_OK) break;
    if ((rc = fr_memcpy_d2h(h_out, dO, ob, nullptr)) != FR_OK) break;
    rc = fr_stream_sync(nullptr);
  } while (0);
  const std::string keep = g_err;
  (void)hipFree(dZ);
  (void)hipFree(dL);
  (void)hipFree(dO);
  (void)hipFree(dW);
  fr_plan_destroy(plan);
  if (rc != FR_OK) g_err = keep;
  return rc;
}

int fr_increments(const double *d_X, int64_t rows, int64_t T, int64_t shift, double *d_out,
                  const double *d_head_src, int64_t head, void *stream) {
  if (rows < 0 || T < 0 || shift < 0) return fail(FR_E_ARG, "fr_increments: bad shape");
  if (rows == 0 || T == 0) return FR_OK;
  if (!d_X || !d_out) return fail(FR_E_ARG, "fr_increments: null device pointer");
  hipError_t e = fr::launch_increments(d_X, rows, T, shift, d_out, d_head_src, head,
                                       (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "increments launch");
  return FR_OK;
}

int fr_pathlen_lookup(const double *d_X, int64_t N, int64_t D, int64_t T, int32_t norm,
                      int32_t relative, double scale, double *d_out, void *stream) {
  const int exact = (norm & FR_LOOKUP_FAST) ? 0 : 1;
  norm &= ~FR_LOOKUP_FAST;
  if (N < 0 || D < 1 || T < 0 || (norm != 1 && norm != 2))
    return fail(FR_E_ARG, "fr_pathlen_lookup: bad argument");
  if (N == 0 || T == 0) return FR_OK;
  if (!d_X || !d_out) return fail(FR_E_ARG, "fr_pathlen_lookup: null device pointer");
  hipError_t e = fr::launch_pathlen_lookup(d_X, N, D, T, norm, relative, scale, exact, d_out,
                                           (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "pathlen_lookup launch");
  return FR_OK;
}

int fr_sieve(int32_t kind, const double *d_A, int64_t N, int64_t T, int64_t a_stride, int32_t inc,
             const int64_t *d_cuts, int64_t cut_rows, int32_t C1, const double *d_q, int32_t Q1,
             double *d_out, int64_t out_stride, void *stream) {
  if (kind < 0 || kind > 2) return fail(FR_E_ARG, "fr_sieve: unknown kind");
  if (N < 0 || T < 1 || C1 < 2 || (cut_rows != 1 && cut_rows != N))
    return fail(FR_E_ARG, "fr_sieve: bad shape");
  if (kind != FR_SIEVE_END && (Q1 < 2 || !d_q)) return fail(FR_E_ARG, "fr_sieve: bad quantiles");
  if (inc < 0 || inc > 8) return fail(FR_E_LIMIT, "fr_sieve: inc must be in [0, 8]");
  if (N == 0) return FR_OK;
  if (!d_A || !d_cuts || !d_out) return fail(FR_E_ARG, "fr_sieve: null device pointer");
  hipError_t e = fr::launch_sieve(kind, d_A, N, T, a_stride, inc, d_cuts, cut_rows, C1, d_q, Q1,
                                  d_out, out_stride, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "sieve launch");
  return FR_OK;
}

struct fr_selection {
  int dev = -1, blob = -1, n_jobs = 0, n_dev = 0;
  hipEvent_t done = nullptr;
  std::vector<int> order, dev_of, via_succ;   // per sorted position (see fr_select_ranks_begin)
  const double *h_out_dev = nullptr;          // page-locked: the jobs' values ...
  const unsigned long long *h_succ = nullptr; // ... and the keys of the next order statistics
};

static void release_selection(fr_selection *sel) {
  if (!sel) return;
  if (sel->done) (void)hipEventDestroy(sel->done);
  if (sel->dev >= 0 && sel->blob >= 0) {
    std::lock_guard<std::mutex> lock(g_scratch_mu[sel->dev]);
    g_scratch[sel->dev][sel->blob].busy = false;
  }
  delete sel;
}

fr_selection_t *fr_select_ranks_begin(const double *d_A, int64_t rows, int64_t N, int64_t T,
                                      int32_t n_jobs, const int32_t *job_row, const int32_t *job_inc,
                                      const int64_t *job_rank, void *stream) {
  auto bad = [](int code, const std::string &msg) -> fr_selection_t * {
    fail(code, msg);
    return nullptr;
  };
  if (rows < 0 || N < 1 || T < 1 || n_jobs < 1 || !job_row || !job_inc || !job_rank)
    return bad(FR_E_ARG, "fr_select_ranks: bad argument");
  if (!d_A) return bad(FR_E_ARG, "fr_select_ranks: null device pointer");
  if (T >= (int64_t(1) << 31))
    return bad(FR_E_LIMIT, "fr_select_ranks: series of 2^31 elements or more (time indices are 32-bit)");
  struct HostJob {
    const double *base;
    unsigned long long prefix;
    long long k;
    int inc, pad;
  };
  static_assert(sizeof(HostJob) == fr::kSelJobBytes, "job layout");
  for (int j = 0; j < n_jobs; ++j)
    if (job_row[j] < 0 || job_row[j] >= rows || job_inc[j] < 0 || job_inc[j] > 8 ||
        job_rank[j] < 0 || job_rank[j] >= N * T)
      return bad(FR_E_ARG, "fr_select_ranks: job " + std::to_string(j) + " out of range");
  // jobs that read the same row block share their passes over it (sorted by row, then
  // differencing order, then rank, so a group computes every difference once per element);
  // identical jobs are run once, and a job that asks for rank r + 1 of the same values as
  // its predecessor's rank r rides on it (one extra pass instead of eight: np.quantile
  // always asks for such neighbours)
  fr_selection *sel = new fr_selection;
  sel->n_jobs = n_jobs;
  std::vector<int> &order = sel->order, &dev_of = sel->dev_of, &via_succ = sel->via_succ;
  order.resize(n_jobs);
  dev_of.resize(n_jobs);
  via_succ.assign(n_jobs, 0);
  for (int j = 0; j < n_jobs; ++j) order[j] = j;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
    if (job_row[x] != job_row[y]) return job_row[x] < job_row[y];
    if (job_inc[x] != job_inc[y]) return job_inc[x] < job_inc[y];
    return job_rank[x] < job_rank[y];
  });
  std::vector<HostJob> jobs;
  std::vector<int32_t> groups;  // {first, count} pairs
  std::vector<int> group_row;
  for (int s = 0; s < n_jobs; ++s) {
    const int j = order[s];
    if (s > 0) {
      const int q = order[s - 1];
      if (job_row[q] == job_row[j] && job_inc[q] == job_inc[j]) {
        if (job_rank[q] == job_rank[j]) {          // duplicate
          dev_of[s] = dev_of[s - 1];
          via_succ[s] = via_succ[s - 1];
          continue;
        }
        if (!via_succ[s - 1] && job_rank[q] + 1 == job_rank[j]) {   // neighbour
          jobs[dev_of[s - 1]].pad |= 1;
          dev_of[s] = dev_of[s - 1];
          via_succ[s] = 1;
          continue;
        }
      }
    }
    dev_of[s] = (int)jobs.size();
    jobs.push_back(HostJob{d_A + (int64_t)job_row[j] * N * T, 0ull, (long long)job_rank[j],
                           job_inc[j], 0});
    if (!groups.empty() && group_row.back() == job_row[j] && groups.back() < fr::kSelGroupMax)
      ++groups.back();
    else {
      groups.push_back(dev_of[s]);
      groups.push_back(1);
      group_row.push_back(job_row[j]);
    }
  }
  const int n_dev = (int)jobs.size();
  const int n_groups = (int)groups.size() / 2;
  sel->n_dev = n_dev;
  // The gather pass tracks the smallest key above the bucket of the first kSelTrackJobs jobs per
  // group and differencing order that want a neighbour (pad bit 4); any further one costs a
  // pass of its own (select_succ_kernel)
  bool untracked = false;
  for (int g = 0; g < n_groups; ++g) {
    int pos = 0, last_inc = -1;   // position of a job among its group's jobs of one order
    for (int j = groups[2 * g]; j < groups[2 * g] + groups[2 * g + 1]; ++j) {
      pos = jobs[j].inc == last_inc ? pos + 1 : 0;
      last_inc = jobs[j].inc;
      if (!(jobs[j].pad & 1)) continue;
      if (pos < fr::kSelTrackJobs) jobs[j].pad |= 16;
      else untracked = true;
    }
  }
  int max_inc = 0;
  for (int j = 0; j < n_jobs; ++j) max_inc = std::max(max_inc, (int)job_inc[j]);
  hipStream_t st = (hipStream_t)stream;
  // device: jobs | groups | (groups still in the passes) | histograms | results | successors |
  // counts | candidate lists; host (page-locked): jobs | groups | results | successors
  const size_t o_jobs = 0;
  const size_t o_groups = align_up(o_jobs + jobs.size() * sizeof(HostJob), 256);
  const size_t o_groups2 = align_up(o_groups + groups.size() * 4, 256);
  const size_t o_hist = align_up(o_groups2 + groups.size() * 4, 256);
  const size_t o_out = align_up(o_hist + (size_t)n_dev * 256 * 4, 256);
  const size_t o_succ = align_up(o_out + (size_t)n_dev * 8, 256);
  const size_t o_cnt = align_up(o_succ + (size_t)n_dev * 8, 256);
  const size_t o_cand = align_up(o_cnt + ((size_t)n_dev + 1) * 4, 256);   // (+1: jobs left in the passes)
  const size_t need = align_up(o_cand + (size_t)n_dev * fr::kSelSmallCap * 8, 256);
  const size_t ho_groups = align_up(jobs.size() * sizeof(HostJob), 256);
  const size_t ho_out = align_up(ho_groups + groups.size() * 4, 256);
  const size_t ho_succ = align_up(ho_out + (size_t)n_dev * 8, 256);
  const size_t need_host = align_up(ho_succ + (size_t)n_dev * 8, 256);
  const int dev = current_device_id();
  if (dev < 0 || dev >= kScratchDevices) {
    delete sel;
    return bad(FR_E_ARG, "fr_select_ranks: device id");
  }
  Scratch *sc = nullptr;
  {
    // a free blob that is large enough, else the free one that is grown (a blob in use - a
    // selection that has begun and not ended - is never touched)
    std::lock_guard<std::mutex> lock(g_scratch_mu[dev]);
    int pick = -1;
    for (int i = 0; i < kScratchBlobs && pick < 0; ++i)
      if (!g_scratch[dev][i].busy && g_scratch[dev][i].bytes >= need && g_scratch[dev][i].host_bytes >= need_host)
        pick = i;
    for (int i = 0; i < kScratchBlobs && pick < 0; ++i)
      if (!g_scratch[dev][i].busy) pick = i;
    if (pick < 0) {
      delete sel;
      return bad(FR_E_LIMIT, "fr_select_ranks: " + std::to_string(kScratchBlobs) +
                                 " selections are in flight on this device - end one first");
    }
    sc = &g_scratch[dev][pick];
    if (sc->bytes < need) {
      if (sc->ptr) (void)hipFree(sc->ptr);
      sc->ptr = nullptr;
      sc->bytes = 0;
      const size_t want = std::max(need + need / 2, (size_t)32 << 20);   // (the slices' needs differ)
      if (hipMalloc(&sc->ptr, want) != hipSuccess) {
        (void)hipGetLastError();
        sc->ptr = nullptr;
        delete sel;
        return bad(FR_E_NOMEM, "fr_select_ranks: device scratch");
      }
      sc->bytes = want;
    }
    if (sc->host_bytes < need_host) {
      if (sc->host) (void)hipHostFree(sc->host);
      sc->host = nullptr;
      sc->host_bytes = 0;
      const size_t want = std::max(need_host * 2, (size_t)1 << 20);
      if (hipHostMalloc(&sc->host, want, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        sc->host = nullptr;
        delete sel;
        return bad(FR_E_NOMEM, "fr_select_ranks: page-locked scratch");
      }
      sc->host_bytes = want;
    }
    sc->busy = true;
    sel->dev = dev;
    sel->blob = pick;
  }
  char *b = static_cast<char *>(sc->ptr), *h = static_cast<char *>(sc->host);
  std::memcpy(h, jobs.data(), jobs.size() * sizeof(HostJob));
  std::memcpy(h + ho_groups, groups.data(), groups.size() * 4);
  sel->h_out_dev = reinterpret_cast<const double *>(h + ho_out);
  sel->h_succ = reinterpret_cast<const unsigned long long *>(h + ho_succ);
  hipError_t e;
  if ((e = hipEventCreateWithFlags(&sel->done, hipEventDisableTiming)) != hipSuccess ||
      (e = hipMemcpyAsync(b + o_jobs, h, jobs.size() * sizeof(HostJob), hipMemcpyHostToDevice, st)) != hipSuccess ||
      (e = hipMemcpyAsync(b + o_groups, h + ho_groups, groups.size() * 4, hipMemcpyHostToDevice, st)) != hipSuccess ||
      (e = hipMemsetAsync(b + o_hist, 0, (size_t)n_dev * 256 * 4, st)) != hipSuccess ||
      (e = hipMemsetAsync(b + o_succ, 0xff, (size_t)n_dev * 8, st)) != hipSuccess ||
      (e = hipMemsetAsync(b + o_cnt, 0, ((size_t)n_dev + 1) * 4, st)) != hipSuccess ||
      // (no host copy of the groups: nothing is read back between the passes)
      (e = fr::launch_select_ranks(b + o_jobs, n_dev, b + o_groups, n_groups, nullptr,
                                   b + o_groups2, max_inc, untracked, N, T,
                                   reinterpret_cast<unsigned int *>(b + o_hist),
                                   reinterpret_cast<double *>(b + o_out),
                                   reinterpret_cast<unsigned long long *>(b + o_succ),
                                   reinterpret_cast<unsigned long long *>(b + o_cand),
                                   reinterpret_cast<unsigned int *>(b + o_cnt), st)) != hipSuccess ||
      (e = hipMemcpyAsync(h + ho_out, b + o_out, (size_t)n_dev * 8, hipMemcpyDeviceToHost, st)) != hipSuccess ||
      (e = hipMemcpyAsync(h + ho_succ, b + o_succ, (size_t)n_dev * 8, hipMemcpyDeviceToHost, st)) != hipSuccess ||
      (e = hipEventRecord(sel->done, st)) != hipSuccess) {
    // (what was queued may still run: the blob is released once the stream has drained)
    (void)hipStreamSynchronize(st);
    release_selection(sel);
    hip_fail(e, "fr_select_ranks");
    return nullptr;
  }
  return sel;
}

int fr_select_ranks_end(fr_selection_t *sel, double *h_out) {
  if (!sel) return fail(FR_E_ARG, "fr_select_ranks_end: null selection");
  hipError_t e = hipEventSynchronize(sel->done);
  if (e != hipSuccess) {
    release_selection(sel);
    return hip_fail(e, "fr_select_ranks_end");
  }
  if (h_out)
    for (int s = 0; s < sel->n_jobs; ++s) {
      double v = sel->h_out_dev[sel->dev_of[s]];
      if (sel->via_succ[s]) {
        const unsigned long long k = sel->h_succ[sel->dev_of[s]];
        if (k != ~0ull) {   // the order-preserving key back to the double (kernels_misc.hip)
          const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
          std::memcpy(&v, &u, 8);
        }
      }
      h_out[sel->order[s]] = v;
    }
  release_selection(sel);
  return FR_OK;
}

int fr_select_ranks(const double *d_A, int64_t rows, int64_t N, int64_t T, int32_t n_jobs,
                    const int32_t *job_row, const int32_t *job_inc, const int64_t *job_rank,
                    double *h_out, void *stream) {
  if (n_jobs == 0 && rows >= 0 && N >= 1 && T >= 1) return FR_OK;
  if (n_jobs > 0 && !h_out) return fail(FR_E_ARG, "fr_select_ranks: bad argument");
  fr_selection_t *sel = fr_select_ranks_begin(d_A, rows, N, T, n_jobs, job_row, job_inc, job_rank, stream);
  if (!sel) return g_last_code;
  return fr_select_ranks_end(sel, h_out);
}

int fr_release_scratch(void) {
  for (int d = 0; d < kScratchDevices; ++d) {
    std::lock_guard<std::mutex> lock(g_scratch_mu[d]);
    for (int i = 0; i < kScratchBlobs; ++i) {
      Scratch &sc = g_scratch[d][i];
      if (sc.busy) continue;   // (a selection in flight keeps its blob)
      if (sc.ptr) (void)hipFree(sc.ptr);
      if (sc.host) (void)hipHostFree(sc.host);
      sc = Scratch{};
    }
  }
  return FR_OK;
}

int fr_pre_transform(const double *d_A, int64_t N, int64_t T, int64_t a_stride, int32_t inc,
                     double *d_out, void *stream) {
  if (N < 0 || T < 0 || inc < 0 || inc > 8) return fail(FR_E_ARG, "fr_pre_transform: bad argument");
  if (N == 0 || T == 0) return FR_OK;
  if (!d_A || !d_out) return fail(FR_E_ARG, "fr_pre_transform: null device pointer");
  hipError_t e = fr::launch_pre_transform(d_A, N, T, a_stride, inc, d_out, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "pre_transform launch");
  return FR_OK;
}

int fr_standardize(const double *d_X, int64_t rows, int64_t T, int32_t div_std, double eps,
                   double *d_out, void *stream) {
  if (rows < 0 || T < 0) return fail(FR_E_ARG, "fr_standardize: bad shape");
  if (rows == 0 || T == 0) return FR_OK;
  if (!d_X || !d_out) return fail(FR_E_ARG, "fr_standardize: null device pointer");
  hipError_t e = fr::launch_standardize(d_X, rows, T, div_std, eps, d_out, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "standardize launch");
  return FR_OK;
}

int fr_coswiss_set_dropout(fr_plan_t *plan, const int32_t *h_indices, int32_t Lmax, int32_t rate,
                           int64_t T) {
  if (!plan || !plan->p || !plan->p->cos || Lmax < 0 || rate < 0 || T < 1 ||
      (rate > 0 && Lmax > 0 && !h_indices))
    return fail(FR_E_ARG, "fr_coswiss_set_dropout: bad argument");
  fr::Plan &p = *plan->p;
  fr::CosProgram &c = *p.cos;
  std::lock_guard<std::mutex> lock(p.mu);
  if (c.d_mask) (void)hipFree(c.d_mask);
  c.d_mask = nullptr;
  c.Lmax = 0;
  c.mask_T = 0;
  if (Lmax == 0) return FR_OK;   // dropout off
  if (Lmax < p.levels)
    return fail(FR_E_ARG, "fr_coswiss_set_dropout: Lmax is shorter than the longest word");
  const size_t rows = (size_t)c.W * c.F * Lmax;
  std::vector<double> mask(rows * (size_t)T, 1.0);
  for (size_t r = 0; r < rows; ++r)
    for (int i = 0; i < rate; ++i) {
      const int32_t idx = h_indices[r * rate + i];
      if (idx < 0 || idx >= T)
        return fail(FR_E_INDEX, "fr_coswiss_set_dropout: index " + std::to_string(idx) +
                                    " is out of bounds for series of length " + std::to_string(T));
      mask[r * (size_t)T + idx] = 0.0;
    }
  int rc = claim_device(p, "fr_coswiss_set_dropout");
  if (rc != FR_OK) return rc;
  HIP_TRY(hipMalloc(&c.d_mask, mask.size() * 8));
  hipError_t e = hipMemcpy(c.d_mask, mask.data(), mask.size() * 8, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(c.d_mask);
    c.d_mask = nullptr;
    return hip_fail(e, "hipMemcpy(dropout mask)");
  }
  c.Lmax = Lmax;
  c.mask_T = T;
  return FR_OK;
}

int fr_coswiss_set_input_stride(fr_plan_t *plan, int64_t unit_stride) {
  if (!plan || !plan->p || !plan->p->cos || unit_stride < 0)
    return fail(FR_E_ARG, "fr_coswiss_set_input_stride: bad argument");
  plan->p->cos->x_unit_stride = unit_stride;
  return FR_OK;
}

int fr_coswiss_ffn(const double *d_X, int64_t N, int64_t D, int64_t T, const double *d_A,
                   const double *d_b, const double *d_C, int32_t hidden, double *d_Z,
                   void *stream) {
  if (N < 0 || D < 1 || T < 0 || hidden < 1) return fail(FR_E_ARG, "fr_coswiss_ffn: bad shape");
  if (N == 0 || T == 0) return FR_OK;
  if (!d_X || !d_A || !d_b || !d_C || !d_Z)
    return fail(FR_E_ARG, "fr_coswiss_ffn: null device pointer");
  hipError_t e = fr::launch_coswiss_ffn(d_X, N, D, T, d_A, d_b, d_C, hidden, d_Z,
                                        (hipStream_t)stream);
  if (e == hipErrorInvalidValue) {
    (void)hipGetLastError();
    return fail(FR_E_LIMIT, "fr_coswiss_ffn: at most 64 hidden units and 16 input dimensions");
  }
  if (e != hipSuccess) return hip_fail(e, "coswiss ffn launch");
  return FR_OK;
}

int fr_arctic_argmax(const double *d_V, int64_t rows, int64_t N, int64_t T, int32_t n_jobs,
                     const int32_t *d_jobs, double *d_P, double *d_out, void *stream) {
  if (rows < 0 || N < 0 || T < 0 || n_jobs < 0)
    return fail(FR_E_ARG, "fr_arctic_argmax: bad shape");
  if (rows == 0 || N == 0 || T == 0 || n_jobs == 0) return FR_OK;
  if (!d_V || !d_jobs || !d_P || !d_out)
    return fail(FR_E_ARG, "fr_arctic_argmax: null device pointer");
  hipError_t e = fr::launch_arctic_argmax(d_V, rows, N, T, n_jobs, d_jobs, d_P, d_out,
                                          (hipStream_t)stream);
  if (e == hipErrorInvalidValue) {
    (void)hipGetLastError();
    return fail(FR_E_LIMIT, "fr_arctic_argmax: grid too large (rows * N, N or jobs)");
  }
  if (e != hipSuccess) return hip_fail(e, "arctic argmax launch");
  return FR_OK;
}

int fr_nan_to_num(double *d_x, int64_t count, void *stream) {
  if (count < 0) return fail(FR_E_ARG, "fr_nan_to_num: bad count");
  if (count == 0) return FR_OK;
  if (!d_x) return fail(FR_E_ARG, "fr_nan_to_num: null device pointer");
  hipError_t e = fr::launch_nan_to_num(d_x, count, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "nan_to_num launch");
  return FR_OK;
}

int fr_coswiss_combine(const double *d_terms, int64_t n_terms, int64_t N, int64_t T,
                       int32_t n_out, const int32_t *d_begin, const double *d_coeff,
                       const int32_t *d_desc, const double *d_trig, double *d_out,
                       int64_t out_row_stride, void *stream) {
  if (n_terms < 0 || N < 0 || T < 0 || n_out < 0)
    return fail(FR_E_ARG, "fr_coswiss_combine: bad shape");
  if (N == 0 || T == 0 || n_out == 0) return FR_OK;
  if (N * (int64_t)n_out > 0x7fffffffLL) return fail(FR_E_LIMIT, "fr_coswiss_combine: grid too large");
  if (out_row_stride < N * T) return fail(FR_E_ARG, "fr_coswiss_combine: rows of d_out overlap");
  if (!d_terms || !d_begin || !d_coeff || !d_desc || !d_trig || !d_out)
    return fail(FR_E_ARG, "fr_coswiss_combine: null device pointer");
  hipError_t e = fr::launch_coswiss_combine(d_terms, N, T, n_out, d_begin, d_coeff, d_desc, d_trig,
                                            d_out, out_row_stride, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "coswiss combine launch");
  return FR_OK;
}

}  // extern "C"

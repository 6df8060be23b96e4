// Internal interface between the C ABI (capi.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "plan.h"

namespace fr {

constexpr int kWalkThreads = 256;

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
  int32_t carry_slots;      // 3 * (records of the program): LDS carry slots
  int32_t carry_in_lds;     // multi-chunk carries fit in LDS
  int32_t letter_sum;       // Arctic: sum a letter's terms before adding them to the prefix
  int32_t semiring;         // kSemiReals / kSemiArctic
  int32_t prefetch_next;    // units of at most this many nodes touch the next unit's rows (0: off)
  int32_t packed;           // short series: wave-per-series kernel (walk_packed.h)
  int32_t wave_rows;        // TEAM = 1 kernel: one wave per row, 4 groups per workgroup
  // fused sieve epilogue (MODE 1 kernels): features instead of the (K,N,T) tensor
  const FeatOp *ops;        // (K, n_ops_padded) feature ops per output row, 64-byte aligned rows
  double *feats;            // (N, feat_stride) zero-initialised features
  double *cnt;              // same shape: band population of MPI features
  int64_t feat_stride;
  int32_t n_ops, n_ops_padded;
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
  int32_t nt_input;         // 1: stage the rows of X with non-temporal loads (interpreter, one group)
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
  unsigned long long *dbg;  // diagnostic stamps (timing build only)
  int32_t debug;            // timing experiments (FRUITS_HIP_DEBUG), 0 in production
};

int walk_chunk_elems(int64_t T);
bool wave_rows_supported(int64_t T, int levels, bool vec_ok);
bool packed_supported(int64_t T, int levels, int semiring);
int static_program_for(const NodeRec *recs, int n, int groups);
hipError_t launch_iss_walk(IssArgs &a, int levels, hipStream_t st);
hipError_t launch_mpi_finalize(double *feats, const double *cnt, int64_t N, int64_t stride,
                               const int32_t *cols, int n_cols, const int32_t *pairs, int n_pairs,
                               int per_sum, int K, hipStream_t st);
hipError_t launch_exp_tables(const double *g, int64_t count, const float *alphas, int n_alpha,
                             double *aux, bool linear, hipStream_t st);
hipError_t launch_increments(const double *X, int64_t rows, int64_t T, int64_t shift, double *out,
                             const double *head_src, int64_t head, hipStream_t st);
hipError_t launch_pathlen_lookup(const double *X, int64_t N, int64_t D, int64_t T, int norm,
                                 int relative, double scale, int exact, double *out,
                                 hipStream_t st);
hipError_t launch_sieve(int kind, const double *A, int64_t N, int64_t T, int64_t a_stride, int inc,
                        const int64_t *cuts, int64_t cut_rows, int C1, const double *q, int Q1,
                        double *out, int64_t out_stride, hipStream_t st);
// jobs sorted so that the jobs of one group (<= kSelGroupMax, same row block) are adjacent;
// groups = int2 {first job, count}
constexpr int kSelGroupMax = 8;
constexpr int kSelSmallCap = 2048;   // candidates a job settles inside one workgroup (kernels_misc.hip)
// job.pad bit 0: also return the next order statistic (succ[job] = its order key; all-ones
// when there is none); succ must be preset to all-ones
hipError_t launch_select_ranks(void *jobs, int n_jobs, const void *groups, int n_groups, int64_t N,
                               int64_t T, unsigned int *hist, double *out,
                               unsigned long long *succ, unsigned long long *cand,
                               unsigned int *cand_count, hipStream_t st);
constexpr int kSelJobBytes = 32;
hipError_t launch_pre_transform(const double *A, int64_t N, int64_t T, int64_t a_stride, int inc,
                                double *out, hipStream_t st);
constexpr int kCosMaxExponent = 8;
hipError_t launch_coswiss(IssArgs &a, int exponent, hipStream_t st);
hipError_t launch_coswiss_ffn(const double *X, int64_t N, int64_t D, int64_t T, const double *A,
                              const double *b, const double *Cm, int hidden, double *Z,
                              hipStream_t st);
hipError_t launch_trig_tables(const float *freqs, int F, int64_t T, double *out, hipStream_t st);
hipError_t launch_coswiss_combine(const double *A, int64_t N, int64_t T, int n_out,
                                  const int32_t *begin, const double *coeff, const int32_t *desc,
                                  const double *trig, double *out, int64_t out_row_stride,
                                  hipStream_t st);
hipError_t launch_nan_to_num(double *x, int64_t count, hipStream_t st);
hipError_t launch_standardize(const double *X, int64_t rows, int64_t T, int div_std, double eps,
                              double *out, hipStream_t st);
hipError_t launch_arctic_argmax(const double *V, int64_t rows, int64_t N, int64_t T, int n_jobs,
                                const int32_t *jobs, double *P, double *out, hipStream_t st);
hipError_t launch_row_stats(const double *X, int64_t N, int64_t D, int64_t T, const int32_t *prep,
                            int n_prep, int div_std, double eps, double *stats, hipStream_t st);

}  // namespace fr

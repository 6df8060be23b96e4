// Internal interface between the C ABI (capi.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "plan.h"
#include "walk_types.h"

namespace fr {


int walk_chunk_elems(int64_t T);
bool packed_supported(int64_t T, int levels, int semiring);
int static_program_for(const NodeRec *recs, int n, int groups, const int32_t *row_src, int rows);
hipError_t launch_iss_walk(IssArgs &a, int levels, hipStream_t st);
hipError_t launch_mpi_finalize(double *feats, const double *cnt, int64_t N, int64_t stride,
                               const int32_t *cols, int n_cols, const int32_t *pairs, int n_pairs,
                               int per_sum, int K, hipStream_t st);
hipError_t launch_gather_row_blocks(const double *src, double *dst, int64_t N, int64_t src_stride,
                                    int64_t dst_stride, int K, int per_sum, const int32_t *walk_of_row,
                                    hipStream_t st);
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
constexpr int kSelTrackJobs = 2;     // jobs per group and differencing order whose successor the gather pass tracks
// (fruit_reduced's fit: 2048 23.0 ms - a few jobs per slice went on through five more digits -
// 4096 20.1, 8192 20.3, 16384 20.3)
constexpr int kSelSmallCap = 4096;   // candidates a job settles inside one workgroup (kernels_misc.hip)
// job.pad bit 0: also return the next order statistic (succ[job] = its order key; all-ones
// when there is none); succ must be preset to all-ones
hipError_t launch_select_ranks(void *jobs, int n_jobs, const void *groups, int n_groups,
                               const int32_t *h_groups, void *groups_scratch, int max_inc,
                               bool untracked, int64_t N, int64_t T, unsigned int *hist, double *out,
                               unsigned long long *succ,
                               unsigned long long *cand, unsigned int *cand_count, hipStream_t st);
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
// Arctic argmax rows -> features (kernels_misc.hip, argmax_sieve_kernel): `words` = device
// (n_words, 4) int32 {first V row, letters, first output row, 0}; the dynamic LDS of a workgroup
// (one row of V, the positions of a word's prefixes) must fit kArgmaxSieveLds
constexpr size_t kArgmaxSieveLds = 64 * 1024 - 4096;
size_t argmax_sieve_lds(int64_t T, int max_len);
hipError_t launch_argmax_sieves(const double *V, int64_t N, int64_t T, const void *words, int n_words,
                                int max_len, const FeatOp *ops, int n_ops, int n_ops_padded,
                                double *feats, double *cnt, int64_t feat_stride,
                                const int32_t *series_cuts, int cut_slots, hipStream_t st);
hipError_t launch_arctic_argmax(const double *V, int64_t rows, int64_t N, int64_t T, int n_jobs,
                                const int32_t *jobs, double *P, double *out, hipStream_t st);
hipError_t launch_row_stats(const double *X, int64_t N, int64_t D, int64_t T, const int32_t *prep,
                            int n_prep, int div_std, double eps, double *stats, hipStream_t st);

}  // namespace fr

/*
 * fruits_hip.h - C ABI of libfruits_hip.so: the MI355X (gfx950) implementation
 * of the FRUITS iterated-sums hot path (INC -> Reals ISS over SimpleWords ->
 * NPI / END [+ MPI]).
 *
 * Plain C: pointers, sizes, opaque handles.  No torch / HIP types in any
 * signature.  Every entry point names the reference interface it replaces
 * (paths relative to the irkri/fruits tree @ 2025-09-05).
 *
 * Conventions
 *   - "d_" pointers are DEVICE pointers (hipMalloc / torch CUDA tensors);
 *     "h_" pointers are HOST pointers.  All float data is IEEE binary64,
 *     C-contiguous, exactly like the reference's numba signatures
 *     (f8 arrays; i4 word tables; f4 alpha; i8 cuts).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *     Device entry points only ENQUEUE work once their plan has been PREPARED
 *     for the shape (fr_plan_prepare / fr_pipeline_prepare: the one-time upload
 *     of the plan's tables): they then never allocate, free or synchronise, so
 *     they may be captured into a hipGraph.  An unprepared plan is uploaded by
 *     its first run (an allocation + a synchronous copy); while the stream is
 *     being captured that run fails with FR_E_ARG instead of breaking the
 *     capture.  A plan's tables live on the device that was current at the
 *     first upload; running it with another device current fails with FR_E_ARG.
 *     Entry points are safe to call from several host threads (plans guard
 *     their own tables; launch-time caches are per device).
 *   - Return value: 0 = ok, <0 = error (FR_E_*); fr_last_error() returns a
 *     thread-local message.  Nothing here ever falls back to a CPU path: with
 *     no HIP device every compute call fails with FR_E_HIP.
 */
#ifndef FRUITS_HIP_H
#define FRUITS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FR_OK 0
#define FR_E_ARG (-1)    /* bad shape / null pointer / out-of-range value      */
#define FR_E_DIM (-2)    /* a word names a dimension > D (the reference reads
                            out of bounds silently, fruits/iss/semiring.py:146) */
#define FR_E_HIP (-3)    /* HIP runtime error (no device, launch failure, ...)  */
#define FR_E_NOMEM (-4)  /* workspace too small / allocation failure            */
#define FR_E_LIMIT (-5)  /* plan exceeds a compiled-in limit                    */
#define FR_E_INDEX (-6)  /* an index the reference would raise IndexError for
                            (END cut outside the series, fruits/sieving/segment.py:213-218) */

/* weighting modes of fr_plan_create */
#define FR_W_NONE 0      /* weighting is None  -> semiring.py:27-28,35 (alpha=0, lookup=0, total) */
#define FR_W_NONTOTAL 1  /* Weighting(total=False) -> _reals_single, semiring.py:93-125           */
#define FR_W_TOTAL 2     /* Weighting(total=True)  -> _total_weighted_reals_single, :128-158      */

/* plan flags */
#define FR_PLAN_SHARE_PREFIXES 1 /* walk the prefix trie (K scans); 0 = one chain
                                    per word, prefixes recomputed like the reference */
#define FR_PLAN_ARCTIC 2         /* (max, +) semiring instead of (+, x):
                                    Arctic._iterated_sum_fast, fruits/iss/semiring.py:282-400
                                    (argmax=False); "next" row of SURVEY.md 8f */
#define FR_PLAN_BAYESIAN 4       /* (max, x) semiring: Bayesian._iterated_sum_fast,
                                    fruits/iss/semiring.py:461-571 - the letters and exp
                                    weights of Reals, a running maximum, no shift */

#define FR_PLAN_ARCTIC_LETTER_SUM 8 /* with FR_PLAN_ARCTIC: the terms of one extended letter are
                                    summed first and their sum added to the prefix - the
                                    rounding of the argmax body (semiring.py:252-256) */

/* fr_plan_info selectors */
#define FR_INFO_ROWS 0       /* K = number of output rows (iterated sums)        */
#define FR_INFO_NODES 1      /* scan passes the device performs                  */
#define FR_INFO_LEVELS 2     /* register frames the walk needs                   */
#define FR_INFO_DIMS_USED 3  /* distinct input dimensions referenced             */
#define FR_INFO_MAX_DIM 4    /* highest dimension referenced (1-based)           */
#define FR_INFO_ALPHAS 5     /* distinct alpha values (exp tables = 2 per alpha) */
#define FR_INFO_GROUPS 6     /* independent sub-tries (upper bound of `groups`)  */
#define FR_INFO_SHARED 7     /* 1 if prefixes are shared                         */
#define FR_INFO_STAGED_ROWS 8 /* rows a workgroup stages in LDS per time chunk (input dimensions +
                                exp tables); fr_iss_run returns FR_E_LIMIT when they do not fit -
                                the caller then splits the word list */
#define FR_INFO_JIT_PROGRAMS 9 /* run-time compiled static programs loaded for this plan (fr_plan_jit) */
#define FR_INFO_AOT_PROGRAM 10 /* 1 + index of the pre-compiled static program (one group per series)
                                  the plan's records equal, 0: none */

/* sieve kinds of fr_sieve_* and the fused pipeline */
#define FR_SIEVE_NPI 0 /* fruits/sieving/increment.py:101-129 */
#define FR_SIEVE_MPI 1 /* fruits/sieving/increment.py:132-163 */
#define FR_SIEVE_END 2 /* fruits/sieving/segment.py:203-225   */
/* OR-ed into a sieve's kind at fr_pipeline_create: its cuts are PER SERIES (coquantile
 * cuts, fruits/sieving/segment.py:51-64) - the sieve's `cuts` entries then name columns
 * ("slots") of the table given to fr_pipeline_set_series_cuts */
#define FR_SIEVE_SERIES_CUTS 0x100

typedef struct fr_plan fr_plan_t;
typedef struct fr_pipeline fr_pipeline_t;

/* ------------------------------------------------------------------ misc */
const char *fr_last_error(void);
int fr_version(void);
/* Number of HIP devices (0 when there is none; never an error). */
int fr_device_count(void);
/* Convenience for callers without their own allocator (cgo, plain ctypes). */
int fr_malloc(void **d_ptr, int64_t bytes);
int fr_free(void *d_ptr);
int fr_memcpy_h2d(void *d_dst, const void *h_src, int64_t bytes, void *stream);
int fr_memcpy_d2h(void *h_dst, const void *d_src, int64_t bytes, void *stream);
int fr_stream_sync(void *stream);

/* ------------------------------------------------------------------ plan
 * Replaces the word loop of _calculate_ISS (fruits/iss/iss.py:21-67) plus the
 * per-word marshalling of Semiring.iterated_sums (fruits/iss/semiring.py:14-41)
 * and CachePlan's depths (fruits/iss/cache.py:17-43).
 *
 *   W        number of words
 *   exps     concatenated exponent tables; word i is (L[i], Dw[i]) int32
 *            row-major == np.array(list(word), dtype=np.int32) (semiring.py:31)
 *   alpha    concatenated per-letter alphas (sum of L[i] floats) == word.alpha
 *            (fruits/iss/words/word.py:71-82); NULL when weighting == FR_W_NONE
 *   depth    depth[i] = CachePlan.unique_el_depth(i) in EXTENDED mode, 1 in
 *            SINGLE mode: number of trailing prefixes of word i that are output
 *            (0 is legal: CachePlan gives 0 to a word that is a prefix of an
 *            earlier word; it then contributes no rows)
 *   Output row order is the reference's: words in order, within a word the
 *   shortest emitted prefix first (iss.py:55-63).
 */
fr_plan_t *fr_plan_create(int32_t W, const int32_t *exps, const int32_t *L,
                          const int32_t *Dw, const float *alpha,
                          const int32_t *depth, int32_t weighting, int32_t flags);
void fr_plan_destroy(fr_plan_t *plan);
int64_t fr_plan_info(const fr_plan_t *plan, int32_t what);
/* Debug / test view of the compiled program: writes up to `cap` int32 words
 * (8 per node: level, flags, n_factors, n_emit, first_emit_row, emit_mul,
 * z_mul, group) and returns the number of nodes. */
int32_t fr_plan_dump(const fr_plan_t *plan, int32_t *buf, int32_t cap);
/* Debug / test view of the DEVICE program for `groups` groups per series: the
 * 64-byte node records (16 int32 words each, sentinel records included) in walk
 * order.  Returns the number of records; copies them when `cap_words` holds them. */
int32_t fr_plan_records(fr_plan_t *plan, int32_t groups, int32_t *buf, int64_t cap_words);
/* Debug / test view of the plan IN PIECES (csrc/plan.h, PiecedProgram): what the fused walk of a
 * large plan runs - the trie covered by items (a chain from the root, walked by the record loop,
 * and a body, a forest of whole sub-tries compiled as straight-line code), equal bodies one
 * piece TYPE with a kernel of its own, the output rows renumbered in walk order.  `max_piece`:
 * nodes of the largest body (0: the default).  Words: {types, K, node executions in chains,
 * nodes}; per type {body nodes, body rows, frames, units, items, nodes of the largest unit,
 * records, rows of the largest unit}, its records (16 words each: the body's - levels counted
 * from the chain's end, output rows from the body's first - then the chains, each closed by a
 * sentinel), its items (4 words: chain byte offset, walk position of the body's first row, nodes
 * of the unit in front, 0), units + 1 item offsets, a walk position per unit; then the output
 * row at every walk position.  Returns the number of words (copied when `cap_words` holds
 * them), 0 when the plan has no such cover. */
int64_t fr_plan_pieces(fr_plan_t *plan, int32_t max_piece, int32_t *buf, int64_t cap_words);
/* Debug / test view of the plan's STATIC schedule for `groups` groups per series (small
 * unweighted plans whose walk is compiled as straight-line code): 32 header words {entries,
 * staged rows, frames, groups, row sources [4..8), first entry of each group [8..16), rows
 * read by each group as bit masks [16..24)} followed by 16 words per entry (node records
 * with the input / output frame in words 14 / 15, "complete row r" entries of kind 0xfe,
 * "load the next unit's rows" 0xfd, a sentinel per group).  Returns the number of entries,
 * 0 when the plan does not qualify. */
int32_t fr_plan_static_schedule(fr_plan_t *plan, int32_t groups, int32_t *buf, int64_t cap_words);
/* Bytes of device workspace fr_iss_run needs for this plan and shape
 * (exp tables for weighted plans, chunk carries for T > one chunk). */
int64_t fr_plan_workspace_bytes(const fr_plan_t *plan, int64_t N, int64_t T,
                                int64_t lookup_rows);
/* 1 when a workgroup can stage the plan's rows (input dimensions + exp tables) of one
 * time chunk of a length-T series in LDS, 0 when fr_iss_run would return FR_E_LIMIT
 * (the caller then splits the word list). */
/* Run-time compiled static program of a small plan (hipRTC): what fr_plan_prepare does for a
 * materialising plan of at most 32 nodes that has no ahead-of-time program.  compile_only != 0:
 * compiles the schedule for `groups` groups per series WITHOUT touching a GPU and returns the
 * size of the code object (0: the plan has no static schedule; FR_E_LIMIT + message: hipRTC
 * missing or the compilation failed).  compile_only == 0: compiles and loads the programs on
 * the current device and returns how many the plan now has; `msg` receives why not. */
int32_t fr_plan_jit(fr_plan_t *plan, int32_t groups, int32_t compile_only, char *msg,
                    int64_t msg_cap);
int32_t fr_plan_fits(const fr_plan_t *plan, int64_t T);
/* One-time upload of the plan's device tables for batches of N series of length T
 * (`groups` as in fr_iss_run).  Allocates and synchronises - call it OUTSIDE a stream
 * capture; afterwards fr_iss_run for that (N, T, groups) only enqueues work.  The
 * reference's callers are synchronous and stateless (fruits/iss/semiring.py:43-52):
 * this is the explicit form of the state a device plan needs.  Idempotent. */
int fr_plan_prepare(fr_plan_t *plan, int64_t N, int64_t T, int32_t groups);

/* ------------------------------------------------------------------ ISS
 * Replaces Reals._iterated_sum_fast for ALL words of the plan in one launch
 * (fruits/iss/semiring.py:167-201, called per word from iss.py:55-63).
 *
 *   d_X            (N, D, T) f64
 *   d_lookup       (lookup_rows, T) f64, lookup_rows in {1, N}: Weighting.get_lookup
 *                  (fruits/iss/weighting.py:100-110, 148-160); a 1-row lookup is
 *                  broadcast over N (Indices); NULL iff the plan is FR_W_NONE
 *   d_out          element (k, n, t) is written at
 *                  d_out[k*out_k_stride + n*out_n_stride + t];
 *                  (K,N,T) results of iss.py:46 => strides (N*T, T);
 *                  (N,E,T) result of semiring.py:177 => strides (T, E*T)
 *   d_work         workspace of >= fr_plan_workspace_bytes() bytes (may be NULL
 *                  when that is 0)
 *   groups         0 = choose; otherwise number of sub-trie groups per series
 */
int fr_iss_run(fr_plan_t *plan, const double *d_X, int64_t N, int64_t D, int64_t T,
               const double *d_lookup, int64_t lookup_rows, double *d_out,
               int64_t out_k_stride, int64_t out_n_stride, void *d_work,
               int64_t work_bytes, int32_t groups, void *stream);

/* Drop-in for Semiring.iterated_sum_fast (fruits/iss/semiring.py:43-52,203-219),
 * host arrays in, host array out, synchronous:
 *   Z (N,D,T) f64, word (L,Dw) i32, alpha (L) f32, lookup (N,T) f64 or NULL
 *   (NULL = the unweighted call of semiring.py:27-28), extended in [1,L],
 *   total_weighting (bit 0; bit 1 set = Arctic semiring, semiring.py:354-400; bit 2 set =
 *   Bayesian semiring, semiring.py:530-571)
 *   -> out (N, extended, T) f64 (caller allocated). */
int fr_iterated_sum_fast_host(const double *h_Z, int64_t N, int64_t D, int64_t T,
                              const int32_t *word, int32_t L, int32_t Dw,
                              const float *alpha, const double *h_lookup,
                              int64_t extended, int32_t total_weighting,
                              double *h_out);

/* ------------------------------------------------------------------ INC
 * Replaces _increments (fruits/cache.py:8-13) as used by INC._transform
 * (fruits/preparation/transform.py:60-75): rows = N*D series of length T,
 * out[r, t] = x[r, t] - x[r, t-shift] for t >= shift, else 0; when
 * keep_head != 0 the first `head` values are copied from d_head_src instead
 * (zero_padding=False, transform.py:73-74). */
int fr_increments(const double *d_X, int64_t rows, int64_t T, int64_t shift,
                  double *d_out, const double *d_head_src, int64_t head, void *stream);

/* ------------------------------------------------------------------ lookups
 * L1.get_lookup (fruits/iss/weighting.py:148-160) on top of _L1_sum
 * (fruits/cache.py:25-31) and NRM (fruits/preparation/transform.py:184-198):
 * g[n,t] = scale * minmax_t(cumsum_t |x_0[t]-x_0[t-1]|), optionally divided by
 * (last + 1e-5) first (relative = 1); constant rows give 0.  relative = 2 returns
 * the raw cumulative path length (the SharedSeedCache entry, cache.py:108-112).
 * d_X is (N, D, T), only dimension 0 is read (cache.py:27).
 * norm: 1 = L1 (abs), 2 = L2 (square).  The path length is summed sequentially like
 * np.cumsum, so the lookup is bit-identical to the reference's (it decides exact ties of
 * max-plus results against fitted quantiles); norm | FR_LOOKUP_FAST uses a parallel scan
 * instead (re-associated sums, ~1e-16 relative) - enough for Reals plans. */
#define FR_LOOKUP_FAST 16
int fr_pathlen_lookup(const double *d_X, int64_t N, int64_t D, int64_t T,
                      int32_t norm, int32_t relative, double scale,
                      double *d_out /* (N,T) */, void *stream);

/* ------------------------------------------------------------------ sieves
 * Replace IncrementSieve._pre_transform + NPI/MPI._backend
 * (fruits/sieving/increment.py:63-71, 107-129, 138-163) and END._transform
 * (fruits/sieving/segment.py:210-219) on a materialised (N, T) iterated sum.
 *
 *   d_A      (N, T) f64, row stride a_stride elements
 *   inc      >= 0: number of increment passes fused into the load
 *   d_cuts   (cut_rows, C1) i64, cut_rows in {1, N}; sorted, leading 0
 *            (segment.py:51-64); a 1-row table is broadcast over N
 *   d_q      (Q1) f64 sorted quantile thresholds (segment.py:66-85)
 *   d_out    feature (n, j*(Q1-1)+k) at d_out[n*out_stride + j*(Q1-1)+k]
 *            (END: Q1 is ignored, feature j at d_out[n*out_stride + j])
 */
int fr_sieve(int32_t kind, const double *d_A, int64_t N, int64_t T, int64_t a_stride,
             int32_t inc, const int64_t *d_cuts, int64_t cut_rows, int32_t C1,
             const double *d_q, int32_t Q1, double *d_out, int64_t out_stride,
             void *stream);

/* ------------------------------------------------------------------ fused pipeline
 * ISS + sieves in ONE launch: replaces the loop of FruitSlice.transform
 * (fruits/fruit.py:538-550) - for every iterated sum, for every sieve,
 * sieve.transform(itsum) - without ever materialising the (K, N, T) tensor.
 * Sieves: NPI / MPI with inc in {0, 1, 2} and END, integer cuts (the same for all
 * series).  Feature (n, k*per_sum + col_s + j*(Q1_s-1) + q) is the reference's
 * column order (iterated sum, then sieve, then segment, then band).
 *
 *   kinds/incs/C1/Q1   per sieve; Q1 is ignored for END
 *   cuts               concatenated transformed cut rows (C1[s] values each: sorted,
 *                      leading 0, negative cuts already resolved for length T -
 *                      fruits/sieving/segment.py:51-64); an END cut c with c - 1 outside
 *                      [-T, T-1] fails with FR_E_INDEX (the reference raises IndexError)
 *   h_quant            HOST (K, q_stride) thresholds of every iterated sum (the fitted
 *                      sieve copies of fruit.py:484-496), q_stride =
 *                      fr_pipeline_info(p, 1) = sum of Q1 over the NPI/MPI sieves;
 *                      fr_pipeline_set_quantiles resolves them with the cuts into a
 *                      device table of per-row feature ops (call once after fit; it
 *                      allocates and synchronises)
 *   d_feats            (N, feat_stride >= F) output, F = fr_pipeline_info(p, 2)
 * Returns FR_E_LIMIT from create when a sieve is outside the fused set (the caller
 * then uses fr_iss_run + fr_sieve).  fr_pipeline_info: 0 features per iterated sum,
 * 1 q_stride, 2 total features, 3 run-time compiled kernels the pipeline holds
 * (fr_pipeline_prepare), 4 those of them with the plan as straight-line code, 5 the kernels
 * of piece types loaded (a large plan in pieces, fr_pipeline_compile_plan). */
fr_pipeline_t *fr_pipeline_create(fr_plan_t *plan, int32_t n_sieves, const int32_t *kinds,
                                  const int32_t *incs, const int32_t *C1, const int32_t *Q1,
                                  const int64_t *cuts, int64_t T);
void fr_pipeline_destroy(fr_pipeline_t *pipeline);
int64_t fr_pipeline_info(const fr_pipeline_t *pipeline, int32_t what);
int64_t fr_pipeline_workspace_bytes(const fr_pipeline_t *pipeline, int64_t N,
                                    int64_t lookup_rows);
int fr_pipeline_set_quantiles(fr_pipeline_t *pipeline, const double *h_quant);
/* Build-time companion of fr_pipeline_prepare / fr_pipeline_compile_plan: compiles the kernels
 * a pipeline with thresholds of this KIND would get at run time - `h_quant` (K, q_stride) needs
 * the right infinities only (a band's shape is an immediate), every finite value stands for any
 * other - into the directory `dir` in the cache's file format, without a device: the sieves as
 * immediates, and the plan (a small one as straight-line code for `groups` groups per series, a
 * large one in pieces).  fruits_amd/gen_bundle.py builds fruits_amd/jit_bundle with it; the
 * library looks there behind the user's cache and in front of the compiler.  Returns the number
 * of code objects the directory now holds for this pipeline, < 0 on error (`msg`). */
int32_t fr_pipeline_bundle(fr_pipeline_t *pipeline, const double *h_quant, int32_t groups,
                           const char *dir, char *msg, int64_t msg_cap);
/* Fuses the slice's preparateurs into the launch: fr_pipeline_run then takes the RAW
 * (N, D, T) input and forms the prepared rows while it stages them - no prepared tensor is
 * written or read.  Covers the chains of the experiment fruits:
 *   inc_lag > 0, as_new == 0   INC(shift=inc_lag)           fruits/preparation/transform.py:60-75
 *   inc_lag > 0, as_new != 0   NEW(INC(shift=inc_lag))      fruits/preparation/wrapper.py:78-93
 *                              (2 D prepared dimensions: the raw ones, then their increments)
 *   standardize 1 / 2          followed by STD(var=False / True), std_eps
 *                                                           fruits/preparation/transform.py:141-147
 * (INC with depth 1 and zero padding, _increments of fruits/cache.py:8-13; STD per series
 * and dimension).  With STD one small pre-pass computes the row statistics into the
 * workspace (np.mean / np.std in numpy's own summation order: bit-identical); everything else
 * is the one fused launch.  D = raw input dimensions.  Every fused kernel forms the rows itself
 * (the cooperative walk and the wave-per-series kernels in their staging, CosWISS where it reads
 * a letter's rows); FR_E_LIMIT only for a CosWISS with per-unit inputs (the randomised ffn): the
 * caller then materialises the prepared input as before.
 * Call before fr_pipeline_workspace_bytes / fr_pipeline_prepare; (0, 0, 0) switches it off. */
int fr_pipeline_set_preparation(fr_pipeline_t *pipeline, int32_t D, int32_t inc_lag,
                                int32_t as_new, int32_t standardize, double std_eps);
/* fr_plan_prepare for the pipeline's plan and series length (after
 * fr_pipeline_set_quantiles): fr_pipeline_run for batches of N series then only
 * enqueues work (the exp-table kernel, the walk, the MPI finalize).  It also compiles the
 * pipeline's OWN walk kernel - the fused walk with the sieves' kinds, differencing orders
 * and cuts as immediates (hipRTC, one code object per sieve list and kernel instantiation,
 * cached on disk; ~2 s the first time on a machine, FRUITS_HIP_JIT=0: not) - which later
 * runs on this device launch instead of the generic instance; results are identical
 * (fr_pipeline_compile_plan: a second kernel that knows the plan too).  No
 * hipRTC, per-series cuts or rows with different op lists: the generic instance stays.
 * May be called from another thread than the one that runs the pipeline (a caller that does
 * not want to wait for the compiler): fr_pipeline_run takes the compiled kernel once it is
 * there. */
int fr_pipeline_prepare(fr_pipeline_t *pipeline, int64_t N, int32_t groups);
/* The second kernel of a pipeline: one that also knows the PLAN, for the group program a
 * batch of N series selects.  A plan of at most 128 nodes becomes straight-line code - nothing
 * of a node is loaded or decoded at run time (BASELINE configs[2]: 243 -> 177 us); of a larger
 * one the kernel knows the node SHAPES (level, flags, output rows: a body per shape, the record
 * loop stays; configs[3] / [4] on one GPU: 9.3 -> 8.7 ms, 17.7 -> 15.9 ms).  Apart from
 * fr_pipeline_prepare because of what it costs: seconds to tens of seconds of hipRTC per plan
 * (115 nodes as straight-line code: ~15 s), cached on disk like the other.  Same results; safe
 * from another thread like fr_pipeline_prepare. */
int fr_pipeline_compile_plan(fr_pipeline_t *pipeline, int64_t N, int32_t groups);
/* fr_pipeline_prepare + fr_pipeline_compile_plan with what the disk cache holds and nothing
 * else: the one-time uploads, and the pipeline's own kernels when an earlier process on this
 * machine compiled them (milliseconds); a miss compiles nothing and leaves no trace. */
int fr_pipeline_prepare_cached(fr_pipeline_t *pipeline, int64_t N, int32_t groups);
/* Per-series segment boundaries for the sieves created with FR_SIEVE_SERIES_CUTS: device
 * table (N, slots) int32, row n = the boundaries of series n (values in [0, T]; the slots of
 * one sieve sorted ascending, the first one its leading 0) - what
 * SegmentSieve._get_transformed_cuts (fruits/sieving/segment.py:51-64) returns for float
 * ("coquantile") cuts.  The table belongs to the caller and must stay valid until the runs
 * that use it have finished; it applies to the following fr_pipeline_run calls with exactly
 * N series. */
int fr_pipeline_set_series_cuts(fr_pipeline_t *pipeline, const int32_t *d_cuts, int64_t N,
                                int32_t slots);
int fr_pipeline_run(fr_pipeline_t *pipeline, const double *d_X, int64_t N, int64_t D, int64_t T,
                    const double *d_lookup, int64_t lookup_rows, double *d_feats,
                    int64_t feat_stride, void *d_work, int64_t work_bytes, int32_t groups,
                    void *stream);

/* IncrementSieve._pre_transform alone (fruits/sieving/increment.py:63-71, inc >= 0),
 * materialised: d_out (N, T) contiguous.  Used by fit, which needs the values for
 * np.quantile (fruits/sieving/segment.py:66-75). */
int fr_pre_transform(const double *d_A, int64_t N, int64_t T, int64_t a_stride, int32_t inc,
                     double *d_out, void *stream);

/* ------------------------------------------------------------------ fit ("next" row)
 * Order statistics for SegmentSieve._fit / IncrementSieve._fit
 * (fruits/sieving/segment.py:66-75, increment.py:73-74): np.quantile(X, q)
 * interpolates between the two order statistics around q*(n-1); job j returns the
 * job_rank[j]-th smallest (0-based) of the job_inc[j]-times differenced
 * (N, T) block number job_row[j] of the device tensor d_A (rows, N, T).  Exact
 * (radix select); synchronous (fit is not a capture path); its scratch is kept between calls -
 * fr_release_scratch frees it. */
int fr_select_ranks(const double *d_A, int64_t rows, int64_t N, int64_t T, int32_t n_jobs,
                    const int32_t *job_row, const int32_t *job_inc, const int64_t *job_rank,
                    double *h_out, void *stream);
/* The same in two halves, for a fit of several slices (fruits/fruit.py:110-121 fits them one after
 * the other): _begin queues every pass of the selection on `stream` and returns at once - nothing
 * is read back between the passes - so the caller can queue the next slice's iterated sums behind
 * it; _end waits for exactly this selection (an event), writes the n_jobs values and frees the
 * handle (h_out NULL: only waits and frees).  d_A must stay valid until _end.  A selection in
 * flight owns one of 8 scratch blobs per device (device memory + page-locked host memory);
 * FR_E_LIMIT (NULL, fr_last_error) when all are taken.  fr_select_ranks = _begin + _end. */
typedef struct fr_selection fr_selection_t;
fr_selection_t *fr_select_ranks_begin(const double *d_A, int64_t rows, int64_t N, int64_t T,
                                      int32_t n_jobs, const int32_t *job_row, const int32_t *job_inc,
                                      const int64_t *job_rank, void *stream);
int fr_select_ranks_end(fr_selection_t *selection, double *h_out);

/* frees the scratch blobs no selection is using */
int fr_release_scratch(void);

/* ------------------------------------------------------------------ STD ("next" row)
 * STD._transform with separately=True (fruits/preparation/transform.py:141-147):
 * every one of the rows = N*D series becomes (x - mean) / (std + eps); std is the
 * population standard deviation (np.std), or 1 when div_std == 0. */
int fr_standardize(const double *d_X, int64_t rows, int64_t T, int32_t div_std, double eps,
                   double *d_out, void *stream);

/* ------------------------------------------------------------------ CosWISS ("next" row)
 * CosWISS.batch_transform without ffn / dropout (fruits/iss/cos.py:11-49,167-181,
 * 289-330): the cosine weighted iterated sums of W simple words (exps / L / Dw as in
 * fr_plan_create) for n_freqs frequencies (float32 like the reference's f4 argument),
 * cosine exponent 1..8, optionally with total weighting.  The result is a plan like
 * any other: fr_iss_run writes its K = W*n_freqs rows (row = word*n_freqs + freq,
 * cos.py:167-181; d_lookup is ignored), fr_plan_workspace_bytes sizes the sin / cos
 * tables the run computes, and fr_pipeline_create fuses the sieves onto it.  The device
 * kernel sums the reference's (exponent+1)^(p-1) terms in factorised form, letter by
 * letter ((exponent+1) scans per letter) - equal up to re-association rounding. */
fr_plan_t *fr_plan_create_coswiss(int32_t W, const int32_t *exps, const int32_t *L,
                                  const int32_t *Dw, int32_t n_freqs, const float *freqs,
                                  int32_t exponent, int32_t total_weighting);

/* The randomised variants of CosWISS (fruits/iss/cos.py:51-164; state drawn by
 * CosWISS._fit :248-263).
 * dropout: _leaky_coswiss_single zeroes the summand of letter k at dropout[k] before its
 *   cumsum (:80).  h_indices (W, n_freqs, Lmax, rate) int32 HOST array =
 *   CosWISS._dropout_indices; T = series length the indices were drawn for.  The plan keeps
 *   a device mask; fr_iss_run / fr_pipeline_run then apply it (Lmax = 0 switches it off).
 *   Allocates and synchronises (call it after fit, outside a capture).
 * ffn: _ffn_coswiss sends the input of every (word, frequency) through _ffn first (:93-113,
 *   :128-135): fr_coswiss_ffn computes Z = C relu(A x + b) per time step for ONE (word,
 *   frequency) (d_A (hidden, D), d_b (hidden), d_C (D, hidden), d_Z (N, D, T); <= 64
 *   hidden units, <= 16 dimensions; enqueues one kernel), and
 *   fr_coswiss_set_input_stride(plan, s) makes unit j = word * n_freqs + freq of the plan
 *   read its input at d_X + j * s (s = N*D*T for the stacked Z's; 0 = all units share d_X). */
int fr_coswiss_set_dropout(fr_plan_t *plan, const int32_t *h_indices, int32_t Lmax, int32_t rate,
                           int64_t T);
int fr_coswiss_set_input_stride(fr_plan_t *plan, int64_t unit_stride);
int fr_coswiss_ffn(const double *d_X, int64_t N, int64_t D, int64_t T, const double *d_A,
                   const double *d_b, const double *d_C, int32_t hidden, double *d_Z,
                   void *stream);

/* The reference's own formulation, term by term, for exponents beyond the kernels above:
 * the cosine weighted ISS (fruits/iss/cos.py:11-49) expands cos(a-b)^s into products of
 * sin / cos powers (cos.py:265-287); every product ("term") is an ordinary Reals ISS of
 * the word over the input extended by one sin and one cos row per frequency, i.e. a
 * fr_plan_create / fr_iss_run program (the terms of all words share prefixes).  This
 * entry point is the remaining reduction `result += weightings[i,0] * tmp` with the
 * total-weighting factors of cos.py:38-48:
 *   d_out[j*out_row_stride + n*T + t] =
 *       sum_{i in [d_begin[j], d_begin[j+1])} d_coeff[i] * d_terms[d_desc[3i], n, t]
 *                                             * sin[t]^d_desc[3i+1] * cos[t]^d_desc[3i+2]
 * summed in ascending i.  d_terms (n_terms, N, T); d_trig (2, T) = sin, cos of one
 * frequency; all pointers device memory.  Enqueues one kernel, no allocation. */
int fr_coswiss_combine(const double *d_terms, int64_t n_terms, int64_t N, int64_t T,
                       int32_t n_out, const int32_t *d_begin, const double *d_coeff,
                       const int32_t *d_desc, const double *d_trig, double *d_out,
                       int64_t out_row_stride, void *stream);

/* ------------------------------------------------------------------ Arctic argmax ("next" row)
 * Arctic(argmax=True): _arctic_argmax_single (fruits/iss/semiring.py:239-284, dispatched
 * from Arctic._iterated_sum_fast :370-378).  The running maxima of EVERY prefix of every word
 * come from an FR_PLAN_ARCTIC plan whose depths are the word lengths (fr_iss_run ->
 * d_V (rows, N, T), rows = sum of L); this entry point derives the positions of the maxima
 * (d_P, scratch of the same shape: the index at which the running maximum was last raised)
 * and assembles the reference's rows: per word and prefix k the values, then the positions
 * of letters 1..k+1 back-tracked from the final position of the next letter (:275-283).
 * d_jobs (n_jobs, 3) int32 on the device, one job per (word, prefix k):
 * {first row of the word in d_V, k, first output row of the prefix = base_w + k + k(k+1)/2};
 * d_out (sum of L + L(L+1)/2, N, T).  Words of <= 63 letters.  Enqueues two kernels. */
int fr_arctic_argmax(const double *d_V, int64_t rows, int64_t N, int64_t T, int32_t n_jobs,
                     const int32_t *d_jobs, double *d_P, double *d_out, void *stream);

/* The same rows straight into features: a pipeline over an FR_PLAN_ARCTIC |
 * FR_PLAN_ARCTIC_LETTER_SUM plan whose rows are all prefixes of `n_words` words of `lengths[w]`
 * letters (in the plan's order) then has sum of L + L(L+1)/2 OUTPUT rows - what FruitSlice.transform
 * sieves for ISS(semiring=Arctic(argmax=True)), fruits/fruit.py:538-550 over iss.py:140-146 - and
 * fr_pipeline_set_quantiles takes one row of thresholds per output row.  fr_pipeline_run materialises
 * the running maxima (sum of L rows, in the workspace) and ONE more kernel forms every argmax row
 * in LDS for the NPI / MPI / END ops that look at it: neither the positions nor the argmax rows are
 * written.  Call between fr_pipeline_create and fr_pipeline_set_quantiles.  FR_E_LIMIT (the caller
 * keeps fr_arctic_argmax + fr_sieve): a sieve differences more than twice or cumulates, or a row
 * of maxima and the positions of a word's prefixes (8 T + 2 L T bytes) exceed 60 KB of LDS. */
int fr_pipeline_set_argmax(fr_pipeline_t *pipeline, int32_t n_words, const int32_t *lengths);

/* ------------------------------------------------------------------ Fruit.transform epilogue
 * np.nan_to_num(result, copy=False, nan=0.0) of Fruit.transform (fruits/fruit.py:172) on the
 * device-resident feature matrix, in place: NaN -> 0, +inf / -inf -> the largest / lowest
 * finite double (numpy's defaults).  d_x: `count` contiguous doubles. */
int fr_nan_to_num(double *d_x, int64_t count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FRUITS_HIP_H */

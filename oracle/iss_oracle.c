/*
 * CPU oracle in C: a restatement of the reference's compiled (numba) kernels on
 * the ISS hot path.  TEST / BASELINE INFRASTRUCTURE ONLY - never linked into
 * fruits_amd/, never a fallback of the product path.  Used by tests/ as a fast
 * checker at full BASELINE sizes and by bench.py's cpu_baseline leg
 * ("kind": "port").  Pinned by tests/test_oracle_c.py against the golden
 * vectors generated from the reference (tests/golden/make_golden.py).
 *
 * Structure follows the reference on purpose (paths relative to the reference
 * root): one pass over a T-vector per numpy statement, prefixes recomputed per
 * word, parallel loop over the series axis N only (numba prange ==
 * `#pragma omp parallel for`).
 *
 *   orc_iterated_sum_fast   fruits/iss/semiring.py:167-201
 *     total branch          fruits/iss/semiring.py:128-158
 *     non-total branch      fruits/iss/semiring.py:93-125
 *   orc_iss_batch           fruits/iss/iss.py:21-67 (+ semiring.py:14-41 marshalling)
 *   orc_increments          fruits/cache.py:8-13
 *   orc_npi_backend         fruits/sieving/increment.py:107-129
 *   orc_mpi_backend         fruits/sieving/increment.py:138-163
 *   orc_end                 fruits/sieving/segment.py:210-219
 *   orc_l1_sum              fruits/cache.py:25-31
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void letters(double *tmp, const double *Z, int64_t T, const int32_t *el, int Dw)
{
    /* semiring.py:111-117 / 143-149 : repeated multiply, or divide for e < 0 */
    for (int d = 0; d < Dw; ++d) {
        int occ = el[d];
        const double *z = Z + (int64_t)d * T;
        if (occ > 0) {
            for (int r = 0; r < occ; ++r)
                for (int64_t t = 0; t < T; ++t) tmp[t] = tmp[t] * z[t];
        } else if (occ < 0) {
            for (int r = 0; r < -occ; ++r)
                for (int64_t t = 0; t < T; ++t) tmp[t] = tmp[t] / z[t];
        }
    }
}

static void cumsum(double *v, int64_t T)
{
    double acc = 0.0;
    for (int64_t t = 0; t < T; ++t) { acc += v[t]; v[t] = acc; }
}

static void shift1(double *v, int64_t T)
{
    /* np.roll(tmp, 1); tmp[0] = 0 */
    memmove(v + 1, v, (size_t)(T - 1) * sizeof(double));
    v[0] = 0.0;
}

static void single_total(const double *Z, int64_t T, const int32_t *word, int L, int Dw,
                         const float *alpha, const double *w, int E, double *out,
                         double *tmp)
{
    for (int64_t t = 0; t < T; ++t) tmp[t] = 1.0;
    for (int k = 0; k < L; ++k) {
        double a = (double)alpha[k];
        letters(tmp, Z, T, word + k * Dw, Dw);
        for (int64_t t = 0; t < T; ++t) tmp[t] = tmp[t] * exp(w[t] * a);
        cumsum(tmp, T);
        if (L - k <= E) {
            double *o = out + (int64_t)(E - (L - k)) * T;
            for (int64_t t = 0; t < T; ++t) o[t] = tmp[t] * exp(-w[t] * a);
        }
        if (k < L - 1) {
            shift1(tmp, T);
            for (int64_t t = 0; t < T; ++t) tmp[t] = tmp[t] * exp(-w[t] * a);
        }
    }
}

static void single_nontotal(const double *Z, int64_t T, const int32_t *word, int L, int Dw,
                            const float *alpha, const double *w, int E, double *out,
                            double *tmp)
{
    for (int64_t t = 0; t < T; ++t) tmp[t] = 1.0;
    for (int k = 0; k < L; ++k) {
        if (k > 0) shift1(tmp, T);
        letters(tmp, Z, T, word + k * Dw, Dw);
        if (k > 0) {
            double a = (double)alpha[k - 1];
            for (int64_t t = 0; t < T; ++t) tmp[t] = tmp[t] * exp(-w[t] * a);
        }
        if (L - k <= E) {
            double *o = out + (int64_t)(E - (L - k)) * T;
            double acc = 0.0;
            for (int64_t t = 0; t < T; ++t) { acc += tmp[t]; o[t] = acc; }
        }
        if (k < L - 1) {
            double a = (double)alpha[k];
            for (int64_t t = 0; t < T; ++t) tmp[t] = tmp[t] * exp(w[t] * a);
            cumsum(tmp, T);
        }
    }
}

/* Unweighted fast lane of the total branch: the reference passes alpha = 0 and
 * lookup = 0 (semiring.py:27-28), so every exp() factor is exactly 1.0 and a
 * multiplication by it is the identity in IEEE arithmetic; skipping it changes
 * no bit. */
static void single_unweighted(const double *Z, int64_t T, const int32_t *word, int L, int Dw,
                              int E, double *out, double *tmp)
{
    for (int64_t t = 0; t < T; ++t) tmp[t] = 1.0;
    for (int k = 0; k < L; ++k) {
        letters(tmp, Z, T, word + k * Dw, Dw);
        cumsum(tmp, T);
        if (L - k <= E)
            memcpy(out + (int64_t)(E - (L - k)) * T, tmp, (size_t)T * sizeof(double));
        if (k < L - 1) shift1(tmp, T);
    }
}

/* Arctic semiring (max, +): fruits/iss/semiring.py:282-309 (_arctic_single) and
 * :317-338 (_total_weighted_arctic_single).  Letters ADD el * Z[dim]; the scan is a
 * running maximum; there is no shift between letters.  w == NULL: unweighted call
 * (alpha = 0, lookup = 0, total branch, semiring.py:27-35) - adding 0.0 is skipped. */
static void cummax(double *v, int64_t T)
{
    for (int64_t t = 1; t < T; ++t) v[t] = v[t - 1] > v[t] ? v[t - 1] : v[t];
}

static void single_arctic(const double *Z, int64_t T, const int32_t *word, int L, int Dw,
                          const float *alpha, const double *w, int total, int E, double *out,
                          double *tmp)
{
    for (int64_t t = 0; t < T; ++t) tmp[t] = 0.0;
    for (int k = 0; k < L; ++k) {
        for (int d = 0; d < Dw; ++d) {
            double el = (double)word[k * Dw + d];
            const double *z = Z + (int64_t)d * T;
            for (int64_t t = 0; t < T; ++t) tmp[t] = tmp[t] + el * z[t];
        }
        double *o = (L - k <= E) ? out + (int64_t)(E - (L - k)) * T : NULL;
        if (w == NULL || total) {
            double a = w ? (double)alpha[k] : 0.0;
            if (w) for (int64_t t = 0; t < T; ++t) tmp[t] = tmp[t] + w[t] * a;
            cummax(tmp, T);
            if (o) for (int64_t t = 0; t < T; ++t) o[t] = w ? tmp[t] - w[t] * a : tmp[t];
            if (k < L - 1 && w) for (int64_t t = 0; t < T; ++t) tmp[t] = tmp[t] - w[t] * a;
        } else {
            if (k > 0) {
                double a = (double)alpha[k - 1];
                for (int64_t t = 0; t < T; ++t) tmp[t] = tmp[t] - w[t] * a;
            }
            if (o) {
                o[0] = tmp[0];
                for (int64_t t = 1; t < T; ++t) o[t] = o[t - 1] > tmp[t] ? o[t - 1] : tmp[t];
            }
            if (k < L - 1) {
                double a = (double)alpha[k];
                for (int64_t t = 0; t < T; ++t) tmp[t] = tmp[t] + w[t] * a;
                cummax(tmp, T);
            }
        }
    }
}

/* Bayesian semiring (max, x): fruits/iss/semiring.py:461-493 (_bayesian_single) and
 * :496-527 (_total_weighted_bayesian_single): the letters and exp weights of the Reals
 * kernels, a running maximum instead of the cumulative sum, no shift.  w == NULL:
 * unweighted call (total branch with exp(0) = 1 factors, skipped). */
static void single_bayesian(const double *Z, int64_t T, const int32_t *word, int L, int Dw,
                            const float *alpha, const double *w, int total, int E,
                            double *out, double *tmp)
{
    for (int64_t t = 0; t < T; ++t) tmp[t] = 1.0;
    for (int k = 0; k < L; ++k) {
        letters(tmp, Z, T, word + k * Dw, Dw);
        double *o = (L - k <= E) ? out + (int64_t)(E - (L - k)) * T : NULL;
        if (w == NULL || total) {
            double a = w ? (double)alpha[k] : 0.0;
            if (w) for (int64_t t = 0; t < T; ++t) tmp[t] = tmp[t] * exp(w[t] * a);
            cummax(tmp, T);
            if (o) for (int64_t t = 0; t < T; ++t) o[t] = w ? tmp[t] * exp(-w[t] * a) : tmp[t];
            if (k < L - 1 && w) for (int64_t t = 0; t < T; ++t) tmp[t] = tmp[t] * exp(-w[t] * a);
        } else {
            if (k > 0) {
                double a = (double)alpha[k - 1];
                for (int64_t t = 0; t < T; ++t) tmp[t] = tmp[t] * exp(-w[t] * a);
            }
            if (o) {
                o[0] = tmp[0];
                for (int64_t t = 1; t < T; ++t) o[t] = o[t - 1] > tmp[t] ? o[t - 1] : tmp[t];
            }
            if (k < L - 1) {
                double a = (double)alpha[k];
                for (int64_t t = 0; t < T; ++t) tmp[t] = tmp[t] * exp(w[t] * a);
                cummax(tmp, T);
            }
        }
    }
}

/* Z (N,D,T); word (L,Dw); alpha (L); lookup (N,T) or NULL (= unweighted);
 * out (N,E,T) with arbitrary strides so the batch entry can write (K,N,T). */
static void iterated_sum_fast_strided(const double *Z, int64_t N, int64_t D, int64_t T,
                                      const int32_t *word, int L, int Dw,
                                      const float *alpha, const double *lookup,
                                      int E, int total, double *out,
                                      int64_t out_n_stride, int64_t out_e_stride,
                                      int nthreads)
{
    const int arctic = total & 2;   /* bit 1 selects the Arctic semiring */
    const int bayesian = total & 4; /* bit 2 the Bayesian one */
    total &= 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        double *tmp = (double *)malloc((size_t)T * sizeof(double));
        double *res = (double *)malloc((size_t)T * (size_t)E * sizeof(double));
#pragma omp for schedule(static)
        for (int64_t j = 0; j < N; ++j) {
            const double *Zj = Z + j * D * T;
            memset(res, 0, (size_t)T * (size_t)E * sizeof(double));
            if (arctic)
                single_arctic(Zj, T, word, L, Dw, alpha, lookup ? lookup + j * T : NULL, total,
                              E, res, tmp);
            else if (bayesian)
                single_bayesian(Zj, T, word, L, Dw, alpha, lookup ? lookup + j * T : NULL,
                                total, E, res, tmp);
            else if (lookup == NULL)
                single_unweighted(Zj, T, word, L, Dw, E, res, tmp);
            else if (total)
                single_total(Zj, T, word, L, Dw, alpha, lookup + j * T, E, res, tmp);
            else
                single_nontotal(Zj, T, word, L, Dw, alpha, lookup + j * T, E, res, tmp);
            for (int e = 0; e < E; ++e)
                memcpy(out + j * out_n_stride + (int64_t)e * out_e_stride,
                       res + (int64_t)e * T, (size_t)T * sizeof(double));
        }
        free(tmp);
        free(res);
    }
}

int orc_iterated_sum_fast(const double *Z, int64_t N, int64_t D, int64_t T,
                          const int32_t *word, int L, int Dw, const float *alpha,
                          const double *lookup, int64_t extended, int total,
                          double *out /* (N,E,T) */, int nthreads)
{
    if (Dw > D || extended < 1 || extended > L) return -1;
    iterated_sum_fast_strided(Z, N, D, T, word, L, Dw, alpha, lookup, (int)extended,
                              total, out, extended * T, T, nthreads);
    return 0;
}

/* The word loop of _calculate_ISS: words are given flattened; word i has L[i]
 * letters of Dw[i] exponents starting at word_off[i] in `exps`, alphas at
 * alpha_off[i] in `alpha`, and emits depth[i] trailing prefixes.  Output is the
 * reference's (K,N,T) `results` array (iss.py:46,55-63). */
int orc_iss_batch(const double *X, int64_t N, int64_t D, int64_t T, int W,
                  const int32_t *exps, const int64_t *word_off, const int32_t *L,
                  const int32_t *Dw, const float *alpha, const int64_t *alpha_off,
                  const int32_t *depth, const double *lookup, int total,
                  double *out /* (K,N,T) */, int nthreads)
{
    int64_t row = 0;
    for (int i = 0; i < W; ++i) {
        if (Dw[i] > D) return -1;
        iterated_sum_fast_strided(X, N, D, T, exps + word_off[i], L[i], Dw[i],
                                  alpha ? alpha + alpha_off[i] : NULL, lookup, depth[i],
                                  total, out + row * N * T, T, N * T, nthreads);
        row += depth[i];
    }
    return 0;
}

void orc_increments(const double *X, int64_t rows, int64_t T, int64_t k, double *out)
{
    /* cache.py:8-13 on (rows = N*D, T) */
    for (int64_t r = 0; r < rows; ++r) {
        const double *x = X + r * T;
        double *o = out + r * T;
        for (int64_t t = 0; t < T; ++t) o[t] = (t >= k) ? x[t] - x[t - k] : 0.0;
    }
}

void orc_l1_sum(const double *X, int64_t N, int64_t D, int64_t T, double *out /* (N,T) */)
{
    for (int64_t n = 0; n < N; ++n) {
        const double *x = X + n * D * T; /* dimension 0 only (cache.py:27) */
        double acc = 0.0;
        for (int64_t t = 0; t < T; ++t) {
            double inc = (t >= 1) ? x[t] - x[t - 1] : 0.0;
            acc += fabs(inc);
            out[n * T + t] = acc;
        }
    }
}

void orc_npi_backend(const double *A, int64_t N, int64_t T, const int64_t *cuts, int C1,
                     const double *q, int Q1, double *out)
{
    int C = C1 - 1, Q = Q1 - 1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i)
        for (int j = 0; j < C; ++j)
            for (int k = 0; k < Q; ++k) {
                int64_t lo = cuts[i * C1 + j], hi = cuts[i * C1 + j + 1];
                if (hi > T) hi = T;
                double cnt = 0.0;
                for (int64_t t = lo; t < hi; ++t) {
                    double v = A[i * T + t];
                    if (q[k] < v && v <= q[k + 1]) cnt += 1.0;
                }
                out[i * (C * Q) + j * Q + k] = cnt;
            }
}

void orc_mpi_backend(const double *A, int64_t N, int64_t T, const int64_t *cuts, int C1,
                     const double *q, int Q1, double *out)
{
    int C = C1 - 1, Q = Q1 - 1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i)
        for (int j = 0; j < C; ++j)
            for (int k = 0; k < Q; ++k) {
                int64_t lo = cuts[i * C1 + j], hi = cuts[i * C1 + j + 1];
                if (hi > T) hi = T;
                double sum = 0.0;
                int64_t cnt = 0;
                for (int64_t t = lo; t < hi; ++t) {
                    double v = A[i * T + t];
                    if (q[k] < v && v <= q[k + 1]) { sum += v; ++cnt; }
                }
                out[i * (C * Q) + j * Q + k] = cnt ? sum / (double)cnt : 0.0;
            }
}

void orc_end(const double *A, int64_t N, int64_t T, const int64_t *cuts, int C1, double *out)
{
    int C = C1 - 1;
    for (int64_t i = 0; i < N; ++i)
        for (int j = 0; j < C; ++j) {
            int64_t idx = cuts[i * C1 + j + 1] - 1;
            if (idx < 0) idx += T; /* numpy take_along_axis wraps -1 */
            out[i * C + j] = A[i * T + idx];
        }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

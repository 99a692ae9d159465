/*
 * xc_oracle_mt_impl.h -- row-parallel (OpenMP) forms of the two row-independent passes, for the
 * CPU baseline legs of bench.py only.  TEST INFRASTRUCTURE, like everything under oracle/.
 *
 * The reference runs exactly these loops under numba's `prange` when XCOLUMNS_NUMBA_PARALLEL=1
 * (numba_csr_functions.py:19-21; `for i in prange(n)` at :166, :242 and :604;
 * experiments/numba_perf_tests.sh:5-6 sets NUMBA_NUM_THREADS=8).  Row work is the serial oracle's
 * (same row routines); only the loop over rows is split over threads.  The confusion pass keeps a
 * private float64 column vector per thread and sums them at the end (numba turns `result[idx] += ..`
 * inside prange into the same per-thread reduction).  The BCA sweep itself has no threaded form:
 * the reference's row loop is serial (block_coordinate.py:448).
 *
 * Included by xc_oracle.c once per value type (T, SFX), after xc_oracle_impl.h.
 */
#include <omp.h>
#define XC_PASTE2(a, b) a##b
#define XC_PASTE(a, b) XC_PASTE2(a, b)
#define FN(name) XC_PASTE(name, SFX)

void FN(oracle_topk_csr_mt)(int64_t n, const int32_t *indptr, const int32_t *indices, const T *data, int k,
                            const T *a, const T *b, int keep_scores, int32_t *out_indices, T *out_data,
                            int n_threads)
{
    int cap = 0;
    for (int64_t i = 0; i < n; ++i) {
        int sz = indptr[i + 1] - indptr[i];
        if (sz > cap) cap = sz;
    }
#pragma omp parallel num_threads(n_threads)
    {
        T *gains = (T *)malloc(sizeof(T) * (size_t)(cap > 0 ? cap : 1));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            for (int q = 0; q < k; ++q) { /* :599-600: pre-filled with column 0 / value 1 */
                out_indices[i * k + q] = 0;
                out_data[i * k + q] = (T)1;
            }
            int32_t s = indptr[i];
            int size = indptr[i + 1] - s;
            for (int p = 0; p < size; ++p) {
                T g = data[s + p];
                if (a) g = g * a[indices[s + p]];
                if (b) g = g + b[indices[s + p]];
                gains[p] = g;
            }
            FN(oracle_topk_row)(gains, indices + s, size, k, out_indices + i * k,
                                keep_scores ? out_data + i * k : NULL);
        }
        free(gains);
    }
}

void FN(oracle_confusion_csr_mt)(int64_t n, int64_t m, const int32_t *t_indptr, const int32_t *t_indices,
                                 const T *t_data, const int32_t *p_indptr, const int32_t *p_indices,
                                 const T *p_data, double *tp, double *fp, double *fn, int n_threads)
{
    int cap = 1;
    for (int64_t i = 0; i < n; ++i) {
        int sz = (t_indptr[i + 1] - t_indptr[i]) + (p_indptr[i + 1] - p_indptr[i]);
        if (sz > cap) cap = sz;
    }
    for (int64_t j = 0; j < m; ++j) tp[j] = fp[j] = fn[j] = 0.0;
    if (n_threads < 1) n_threads = 1;
    /* one private {tp | fp | fn} block per thread, summed over the threads label-parallel at the end */
    double *all = (double *)calloc((size_t)n_threads * 3 * (size_t)m, sizeof(double));
#pragma omp parallel num_threads(n_threads)
    {
        T *od = (T *)malloc(sizeof(T) * (size_t)cap);
        int32_t *oi = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
        double *tmp = (double *)malloc(sizeof(double) * (size_t)cap);
        double *mine = all + (size_t)omp_get_thread_num() * 3 * (size_t)m;
        for (int pass = 0; pass < 3; ++pass) {
            double *dst = mine + (int64_t)pass * m;
#pragma omp for schedule(static) nowait
            for (int64_t i = 0; i < n; ++i) {
                const T *td = t_data + t_indptr[i];
                const int32_t *ti = t_indices + t_indptr[i];
                int tn_ = t_indptr[i + 1] - t_indptr[i];
                const T *pd = p_data + p_indptr[i];
                const int32_t *pi = p_indices + p_indptr[i];
                int pn = p_indptr[i + 1] - p_indptr[i];
                int c;
                if (pass == 0) c = FN(oracle_vec_mul_vec)(pd, pi, pn, td, ti, tn_, od, oi);
                else if (pass == 1) c = FN(oracle_vec_mul_ones_minus_vec)(pd, pi, pn, td, ti, tn_, od, oi);
                else c = FN(oracle_vec_mul_ones_minus_vec)(td, ti, tn_, pd, pi, pn, od, oi);
                FN(oracle_scatter)(dst, oi, od, c, 1.0, tmp);
            }
        }
        free(od);
        free(oi);
        free(tmp);
#pragma omp barrier
        const int nt = omp_get_num_threads();
#pragma omp for schedule(static)
        for (int64_t j = 0; j < m; ++j) {
            double a = 0.0, b = 0.0, c = 0.0;
            for (int t = 0; t < nt; ++t) {
                const double *blk = all + (size_t)t * 3 * (size_t)m;
                a += blk[j];
                b += blk[m + j];
                c += blk[2 * m + j];
            }
            tp[j] = a;
            fp[j] = b;
            fn[j] = c;
        }
    }
    free(all);
}

#undef FN
#undef XC_PASTE
#undef XC_PASTE2

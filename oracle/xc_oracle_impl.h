/*
 * xc_oracle_impl.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Type-generic body of the CPU oracle; included twice by xc_oracle.c with
 *   T    = float / double  (dtype of y_proba / y_pred data)
 *   SFX  = _f32 / _f64
 * Every function restates, sequentially and op-for-op, a function of the
 * reference (paths relative to /root/reference/xcolumns/).  Compile with
 * -ffp-contract=off: the reference never fuses multiply and add.
 *
 * Conventions shared by all functions:
 *   - CSR index arrays (indptr, indices) are int32, indices sorted ascending
 *     in each row (numba_csr_functions.py:121 requires it, never checks it);
 *   - confusion vectors tp/fp/fn/tn are float64 (types.py:14);
 *   - ties between equal gains at the top-k boundary go to the LOWER column
 *     id (np.argpartition leaves the choice unspecified; fixtures avoid ties).
 */

#define XC_PASTE2(a, b) a##b
#define XC_PASTE(a, b) XC_PASTE2(a, b)
#define FN(name) XC_PASTE(name, SFX)

/* --------------------------------------------------------------------------
 * `dst[idx] += sign * val` with numpy/numba fancy-index semantics
 * (numba_csr_functions.py:178, :254, :403-417): the right-hand side
 * `dst[idx] + val` is gathered first and then scattered, so when a column id
 * repeats in idx (non-canonical rows, e.g. the column-0 padding top-k leaves in
 * rows shorter than k) only its LAST occurrence takes effect.
 * tmp: scratch with room for c doubles.
 * ------------------------------------------------------------------------ */
static void FN(oracle_scatter)(double *dst, const int32_t *oi, const T *od,
                               int c, double sign, double *tmp)
{
    for (int q = 0; q < c; ++q) tmp[q] = dst[oi[q]] + sign * (double)od[q];
    for (int q = 0; q < c; ++q) dst[oi[q]] = tmp[q];
}

/* --------------------------------------------------------------------------
 * numba_argtopk_csr / numba_topk_csr (numba_csr_functions.py:455-484).
 * gains of type T; writes up to k (column id, gain) pairs, ids ascending;
 * returns how many were written (all `size` of them, in stored order, when
 * size <= k -- :465-466).
 * ------------------------------------------------------------------------ */
/* NaN gains: np.argpartition(-g, k) sorts NaN last, i.e. a NaN gain is the
 * least preferred; compare on a key that maps NaN to -inf. */
static T FN(oracle_key)(T g) { return (g != g) ? (T)-INFINITY : g; }

static int FN(oracle_topk_row)(const T *gains, const int32_t *ids, int size,
                               int k, int32_t *out_ids, T *out_vals)
{
    if (size <= k) {
        for (int p = 0; p < size; ++p) {
            out_ids[p] = ids[p];
            if (out_vals) out_vals[p] = gains[p];
        }
        return size;
    }
    /* selected[] = positions of the k largest gains; position order is column
     * order because the row is sorted, so scanning positions ascending emits
     * ascending ids (the reference sorts the chosen ids, :463, :479-481). */
    int cnt = 0;
    for (int p = 0; p < size; ++p) {
        /* rank of p = number of entries that beat it */
        int rank = 0;
        T kp = FN(oracle_key)(gains[p]);
        for (int q = 0; q < size; ++q) {
            T kq = FN(oracle_key)(gains[q]);
            if (kq > kp || (kq == kp && q < p)) ++rank;
        }
        if (rank < k) {
            out_ids[cnt] = ids[p];
            if (out_vals) out_vals[cnt] = gains[p];
            ++cnt;
        }
    }
    return cnt;
}

/* --------------------------------------------------------------------------
 * numba_predict_weighted_per_instance_csr, k > 0 branch
 * (numba_csr_functions.py:585-629).  out_indices / out_data have n*k slots and
 * are pre-filled here with 0 / 1 exactly like :599-600, so a row with fewer
 * than k entries keeps the reference's padding (column 0, value 1).
 * a, b may be NULL; they are already cast to T (weighted_prediction.py:72-75).
 * ------------------------------------------------------------------------ */
void FN(oracle_topk_csr)(int64_t n, const int32_t *indptr,
                         const int32_t *indices, const T *data, int k,
                         const T *a, const T *b, int keep_scores,
                         int32_t *out_indices, T *out_data)
{
    for (int64_t p = 0; p < n * (int64_t)k; ++p) {
        out_indices[p] = 0;
        out_data[p] = (T)1;
    }
    int cap = 0;
    for (int64_t i = 0; i < n; ++i) {
        int sz = indptr[i + 1] - indptr[i];
        if (sz > cap) cap = sz;
    }
    T *gains = (T *)malloc(sizeof(T) * (size_t)(cap > 0 ? cap : 1));
    for (int64_t i = 0; i < n; ++i) {
        int32_t s = indptr[i];
        int size = indptr[i + 1] - s;
        for (int p = 0; p < size; ++p) {
            T g = data[s + p];
            if (a) g = g * a[indices[s + p]]; /* :608-609 */
            if (b) g = g + b[indices[s + p]]; /* :610-611 */
            gains[p] = g;
        }
        FN(oracle_topk_row)(gains, indices + s, size, k, out_indices + i * k,
                            keep_scores ? out_data + i * k : NULL);
    }
    free(gains);
}

/* --------------------------------------------------------------------------
 * numba_predict_weighted_per_instance_csr, k == 0 branch
 * (numba_csr_functions.py:631-653 -> :549-582 -> numba_set_gains_csr :516-517
 * with is_insert=False): row i keeps the ids whose gain >= th, stored order.
 * out_indptr has n+1 slots; out_indices has nnz(y_proba) slots (upper bound).
 * Data of the result is all ones (:599, numba_resize fill 1.0 :533).
 * ------------------------------------------------------------------------ */
void FN(oracle_threshold_csr)(int64_t n, const int32_t *indptr,
                              const int32_t *indices, const T *data, T th,
                              const T *a, const T *b, int32_t *out_indptr,
                              int32_t *out_indices)
{
    int32_t w = 0;
    out_indptr[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        for (int32_t p = indptr[i]; p < indptr[i + 1]; ++p) {
            T g = data[p];
            if (a) g = g * a[indices[p]];
            if (b) g = g + b[indices[p]];
            if (g >= th) out_indices[w++] = indices[p];
        }
        out_indptr[i + 1] = w;
    }
}

/* --------------------------------------------------------------------------
 * _predict_weighted_per_instance_dense, numpy branch
 * (weighted_prediction.py:25-60).  The caller passes gains already computed
 * in numpy's promoted dtype (y_proba * a + b, :37-41) as type T; this picks
 * the top-k columns of every row (:46) and writes 1 (or the gain) into the
 * zero-initialised y_pred (:35, :47-49).  k == 0: gains >= th (:58).
 * ------------------------------------------------------------------------ */
void FN(oracle_topk_dense)(int64_t n, int64_t m, const T *gains, int k, T th,
                           int keep_scores, T *y_pred)
{
    for (int64_t p = 0; p < n * m; ++p) y_pred[p] = (T)0;
    for (int64_t i = 0; i < n; ++i) {
        const T *g = gains + i * m;
        T *o = y_pred + i * m;
        if (k > 0) {
            for (int64_t p = 0; p < m; ++p) {
                int64_t rank = 0;
                T kp = FN(oracle_key)(g[p]);
                for (int64_t q = 0; q < m && rank < k; ++q) {
                    T kq = FN(oracle_key)(g[q]);
                    if (kq > kp || (kq == kp && q < p)) ++rank;
                }
                if (rank < k) o[p] = keep_scores ? g[p] : (T)1;
            }
        } else {
            for (int64_t p = 0; p < m; ++p)
                if (g[p] >= th) o[p] = (T)1;
        }
    }
}

/* --------------------------------------------------------------------------
 * numba_csr_vec_mul_vec (numba_csr_functions.py:115-140): products of the
 * entries two sorted sparse vectors share; result dtype = dtype of `a`.
 * Returns the count written to (od, oi).
 * ------------------------------------------------------------------------ */
static int FN(oracle_vec_mul_vec)(const T *ad, const int32_t *ai, int an,
                                  const T *bd, const int32_t *bi, int bn,
                                  T *od, int32_t *oi)
{
    int i = 0, j = 0, k = 0;
    while (i < an && j < bn) {
        if (ai[i] < bi[j]) {
            ++i;
        } else if (ai[i] == bi[j]) {
            od[k] = ad[i] * bd[j];
            oi[k] = ai[i];
            ++k; ++i; ++j;
        } else {
            ++j;
        }
    }
    return k;
}

/* --------------------------------------------------------------------------
 * numba_csr_vec_mul_ones_minus_vec (numba_csr_functions.py:185-213):
 * a * (1 - b) over the support of a; an entry of a that b lacks keeps a's
 * value (:200-203).  `1.0 - b` is evaluated in float64 by numba (the literal
 * is a float64) and the product is rounded to T on the store into
 * new_data (:197, :206); un-JIT'd numpy computes it in T.  Both agree
 * whenever a's value is exactly 1 (binary predictions) or b's is (fn pass),
 * the only cases on the hot path; the float64 form is restated here.
 * ------------------------------------------------------------------------ */
static int FN(oracle_vec_mul_ones_minus_vec)(const T *ad, const int32_t *ai,
                                             int an, const T *bd,
                                             const int32_t *bi, int bn, T *od,
                                             int32_t *oi)
{
    int i = 0, j = 0, k = 0;
    while (i < an) {
        if (j >= bn || ai[i] < bi[j]) {
            od[k] = ad[i];
            oi[k] = ai[i];
            ++k; ++i;
        } else if (ai[i] == bi[j]) {
            od[k] = (T)((double)ad[i] * (1.0 - (double)bd[j]));
            oi[k] = ai[i];
            ++k; ++i; ++j;
        } else {
            ++j;
        }
    }
    return k;
}

/* --------------------------------------------------------------------------
 * calculate_confusion_matrix, CSR branch, axis=0, normalize=False
 * (confusion_matrix.py:364-399 -> :174-228 -> numba_csr_functions.py:143-182,
 * :216-258): three independent passes over the rows in order, each adding the
 * row's merged products into a float64 column vector (:178, :254).
 * tp <- pred (.) true ; fp <- pred (.) (1 - true) ; fn <- true (.) (1 - pred).
 * tn is left to the caller (confusion_matrix.py:391-397).
 * ------------------------------------------------------------------------ */
void FN(oracle_confusion_csr)(int64_t n, int64_t m, const int32_t *t_indptr,
                              const int32_t *t_indices, const T *t_data,
                              const int32_t *p_indptr, const int32_t *p_indices,
                              const T *p_data, double *tp, double *fp,
                              double *fn)
{
    int cap = 1;
    for (int64_t i = 0; i < n; ++i) {
        int sz = (t_indptr[i + 1] - t_indptr[i]) + (p_indptr[i + 1] - p_indptr[i]);
        if (sz > cap) cap = sz;
    }
    T *od = (T *)malloc(sizeof(T) * (size_t)cap);
    int32_t *oi = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)cap);
    for (int64_t j = 0; j < m; ++j) tp[j] = fp[j] = fn[j] = 0.0;
    for (int pass = 0; pass < 3; ++pass) {
        for (int64_t i = 0; i < n; ++i) {
            const T *td = t_data + t_indptr[i];
            const int32_t *ti = t_indices + t_indptr[i];
            int tn_ = t_indptr[i + 1] - t_indptr[i];
            const T *pd = p_data + p_indptr[i];
            const int32_t *pi = p_indices + p_indptr[i];
            int pn = p_indptr[i + 1] - p_indptr[i];
            int c;
            double *dst;
            if (pass == 0) {
                c = FN(oracle_vec_mul_vec)(pd, pi, pn, td, ti, tn_, od, oi);
                dst = tp;
            } else if (pass == 1) {
                c = FN(oracle_vec_mul_ones_minus_vec)(pd, pi, pn, td, ti, tn_, od, oi);
                dst = fp;
            } else {
                c = FN(oracle_vec_mul_ones_minus_vec)(td, ti, tn_, pd, pi, pn, od, oi);
                dst = fn;
            }
            FN(oracle_scatter)(dst, oi, od, c, 1.0, tmp);
        }
    }
    free(od);
    free(oi);
    free(tmp);
}

/* --------------------------------------------------------------------------
 * calculate_confusion_matrix, dense branch, axis=0
 * (confusion_matrix.py:160-166, :187-202): np.sum(y_true * y_pred, axis=0,
 * dtype=float64) etc.  The elementwise products are formed in T, the
 * reduction over rows adds them one row at a time into float64.
 * ------------------------------------------------------------------------ */
void FN(oracle_confusion_dense)(int64_t n, int64_t m, const T *y_true,
                                const T *y_pred, double *tp, double *fp,
                                double *fn)
{
    for (int64_t j = 0; j < m; ++j) tp[j] = fp[j] = fn[j] = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const T *t = y_true + i * m;
        const T *p = y_pred + i * m;
        for (int64_t j = 0; j < m; ++j) {
            T one = (T)1;
            T vtp = t[j] * p[j];
            T vfp = (one - t[j]) * p[j];
            T vfn = t[j] * (one - p[j]);
            tp[j] += (double)vtp;
            fp[j] += (double)vfp;
            fn[j] += (double)vfn;
        }
    }
}

/* --------------------------------------------------------------------------
 * numba_sub_from_/numba_add_to_unnormalized_confusion_matrix_csr
 * (numba_csr_functions.py:385-452).  sign = -1 subtracts, +1 adds.
 * scratch: od/oi with room for (tn + pn) entries each.
 * ------------------------------------------------------------------------ */
static void FN(oracle_update_conf_row)(double *tp, double *fp, double *fn,
                                       double *tn, int64_t m, const T *td,
                                       const int32_t *ti, int tcount,
                                       const T *pd, const int32_t *pi,
                                       int pcount, int skip_tn, double sign,
                                       T *od, int32_t *oi, T *od2, int32_t *oi2,
                                       T *od3, int32_t *oi3, double *tmp)
{
    int c_tp = FN(oracle_vec_mul_vec)(pd, pi, pcount, td, ti, tcount, od, oi);
    FN(oracle_scatter)(tp, oi, od, c_tp, sign, tmp);
    int c_fp = FN(oracle_vec_mul_ones_minus_vec)(pd, pi, pcount, td, ti, tcount, od2, oi2);
    FN(oracle_scatter)(fp, oi2, od2, c_fp, sign, tmp);
    int c_fn = FN(oracle_vec_mul_ones_minus_vec)(td, ti, tcount, pd, pi, pcount, od3, oi3);
    FN(oracle_scatter)(fn, oi3, od3, c_fn, sign, tmp);
    if (!skip_tn) {
        /* :413-417 / :448-452 */
        for (int64_t j = 0; j < m; ++j) tn[j] += sign;
        FN(oracle_scatter)(tn, oi, od, c_tp, -sign, tmp);
        FN(oracle_scatter)(tn, oi2, od2, c_fp, -sign, tmp);
        FN(oracle_scatter)(tn, oi3, od3, c_fn, -sign, tmp);
    }
}

/* --------------------------------------------------------------------------
 * One BCA sweep over CSR rows: the loop at block_coordinate.py:448-463 with
 * _bc_with_0approx_step_csr (:212-293) as its body, k > 0, fixed-length rows
 * (y_pred has exactly k entries per row and the new selection has k entries
 * too, so numba_set_gains_csr takes its in-place branch, :523-524; requires
 * every visited row of y_proba to hold >= k entries).
 *
 *   n_norm   the divisor `n` of the step (:229-231 -- the step always
 *            normalises by the row count; the driver never forwards
 *            normalize_conf_matrix, :449-463)
 *   order    row ids to visit, in this order (:448)
 *   p_indices  y_pred.indices, stride k, updated in place; y_pred.data is all
 *            ones on this path and is passed as p_data (stride k)
 *   tp/fp/fn/tn  running float64 statistics, updated in place
 *   greedy   skip the "remove" step (:243)
 *   maximize gains negated when minimising (:281-282)
 * ------------------------------------------------------------------------ */
void FN(oracle_bca_sweep_csr)(int64_t n_norm, int64_t m, int64_t n_order,
                              const int64_t *order, const int32_t *t_indptr,
                              const int32_t *t_indices, const T *t_data,
                              int32_t *p_indices, const T *p_data, int k,
                              double *tp, double *fp, double *fn, double *tn,
                              const xc_oracle_metric *metric, int greedy,
                              int maximize, int skip_tn)
{
    int cap = 1;
    for (int64_t q = 0; q < n_order; ++q) {
        int64_t i = order[q];
        int sz = t_indptr[i + 1] - t_indptr[i];
        if (sz > cap) cap = sz;
    }
    cap += k;
    T *od = (T *)malloc(sizeof(T) * (size_t)cap * 3);
    int32_t *oi = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap * 3);
    double *gains = (double *)malloc(sizeof(double) * (size_t)cap);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)cap);
    int32_t *new_ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)(k > 0 ? k : 1));
    const double nn = (double)n_norm;

    for (int64_t q = 0; q < n_order; ++q) {
        int64_t i = order[q];
        const T *td = t_data + t_indptr[i];
        const int32_t *ti = t_indices + t_indptr[i];
        int tc = t_indptr[i + 1] - t_indptr[i];
        int32_t *pi = p_indices + i * k;
        const T *pd = p_data + i * k;

        if (!greedy) /* :243-246 */
            FN(oracle_update_conf_row)(tp, fp, fn, tn, m, td, ti, tc, pd, pi, k,
                                       skip_tn, -1.0, od, oi, od + cap, oi + cap,
                                       od + 2 * cap, oi + 2 * cap, tmp);

        for (int p = 0; p < tc; ++p) {
            int32_t j = ti[p];
            T eta = td[p];
            T one_minus = (T)1 - eta;          /* (1 - t_data) in T, :253 */
            double neg_tp = tp[j], neg_fp = fp[j], pos_fn = fn[j];
            double pos_tpp = (neg_tp + (double)eta) / nn;       /* :252 */
            double pos_fpp = (neg_fp + (double)one_minus) / nn; /* :253 */
            double neg_fnn = (pos_fn + (double)eta) / nn;       /* :254 */
            neg_tp /= nn; neg_fp /= nn; pos_fn /= nn;           /* :256-258 */
            double pos_tn = tn[j];                              /* :260 */
            double neg_tnn = pos_tn;                            /* :261 */
            if (!skip_tn) {                                     /* :262-264 */
                neg_tnn = (pos_tn + (double)one_minus) / nn;
                pos_tn /= nn;
            }
            double g = xc_oracle_metric_eval(metric, pos_tpp, pos_fpp, pos_fn, pos_tn)
                     - xc_oracle_metric_eval(metric, neg_tp, neg_fp, neg_fnn, neg_tnn);
            if (!maximize) g = -g;                              /* :281-282 */
            gains[p] = g;
        }

        /* numba_set_gains_csr (:499-524) -> numba_argtopk_csr (:455-466) */
        int cnt = xc_oracle_topk_row_f64(gains, ti, tc, k, new_ids);
        for (int p = 0; p < cnt; ++p) pi[p] = new_ids[p];

        /* :290-293 */
        FN(oracle_update_conf_row)(tp, fp, fn, tn, m, td, ti, tc, pd, pi, k,
                                   skip_tn, +1.0, od, oi, od + cap, oi + cap,
                                   od + 2 * cap, oi + 2 * cap, tmp);
    }
    free(od); free(oi); free(gains); free(new_ids); free(tmp);
}

/* --------------------------------------------------------------------------
 * One BCA sweep over dense rows: block_coordinate.py:448-463 with
 * _bc_with_0approx_step_dense (:132-209) as its body.  k > 0 picks the k
 * best gains (:193-198); k == 0 predicts every label whose (negated) gain is
 * <= 0 (:199-200).  y_pred (n x m, values 0/1 of type T) is updated in place.
 * ------------------------------------------------------------------------ */
void FN(oracle_bca_sweep_dense)(int64_t n_norm, int64_t m, int64_t n_order,
                                const int64_t *order, const T *y_proba,
                                T *y_pred, int k, double *tp, double *fp,
                                double *fn, double *tn,
                                const xc_oracle_metric *metric, int greedy,
                                int maximize, int skip_tn)
{
    double *gains = (double *)malloc(sizeof(double) * (size_t)m);
    const double nn = (double)n_norm;
    const T one = (T)1;
    for (int64_t q = 0; q < n_order; ++q) {
        int64_t i = order[q];
        const T *eta = y_proba + i * m;
        T *pred = y_pred + i * m;

        if (!greedy) { /* :157-163 */
            for (int64_t j = 0; j < m; ++j) {
                tp[j] -= (double)(T)(pred[j] * eta[j]);
                fp[j] -= (double)(T)(pred[j] * (one - eta[j]));
                fn[j] -= (double)(T)((one - pred[j]) * eta[j]);
                if (!skip_tn) tn[j] -= (double)(T)((one - pred[j]) * (one - eta[j]));
            }
        }
        for (int64_t j = 0; j < m; ++j) { /* :166-188 */
            T om = one - eta[j];
            double pos_tp = tp[j] + (double)eta[j];
            double pos_fp = fp[j] + (double)om;
            double neg_fn = fn[j] + (double)eta[j];
            double neg_tn = tn[j];
            if (!skip_tn) neg_tn = tn[j] + (double)om;
            double g = xc_oracle_metric_eval(metric, pos_tp / nn, pos_fp / nn, fn[j] / nn, tn[j] / nn)
                     - xc_oracle_metric_eval(metric, tp[j] / nn, fp[j] / nn, neg_fn / nn, neg_tn / nn);
            if (maximize) g = -g; /* :187-188: smaller is better from here on */
            gains[j] = g;
        }
        for (int64_t j = 0; j < m; ++j) pred[j] = (T)0; /* :191 */
        if (k > 0) { /* :193-198: k smallest negated gains */
            for (int64_t p = 0; p < m; ++p) {
                int64_t rank = 0;
                /* gains are negated here: NaN sorts last = key +inf */
                double kp = (gains[p] != gains[p]) ? INFINITY : gains[p];
                for (int64_t r = 0; r < m && rank < k; ++r) {
                    double kr = (gains[r] != gains[r]) ? INFINITY : gains[r];
                    if (kr < kp || (kr == kp && r < p)) ++rank;
                }
                if (rank < k) pred[p] = one;
            }
        } else { /* :199-200 */
            for (int64_t j = 0; j < m; ++j)
                if (gains[j] <= 0.0) pred[j] = one;
        }
        for (int64_t j = 0; j < m; ++j) { /* :203-209 */
            tp[j] += (double)(T)(pred[j] * eta[j]);
            fp[j] += (double)(T)(pred[j] * (one - eta[j]));
            fn[j] += (double)(T)((one - pred[j]) * eta[j]);
            if (!skip_tn) tn[j] += (double)(T)((one - pred[j]) * (one - eta[j]));
        }
    }
    free(gains);
}

#undef FN
#undef XC_PASTE
#undef XC_PASTE2

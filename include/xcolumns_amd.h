/*
 * xcolumns_amd.h -- C ABI of libxcolumns_amd.so (MI355X / gfx950).
 *
 * The drop-in boundary for xCOLUMNs' block-coordinate-ascent prediction path.
 * The reference (mwydmuch/xCOLUMNs 0.0.3) has no FFI of its own: its native
 * layer is the set of numba-JIT functions in xcolumns/numba_csr_functions.py,
 * flat functions over raw (data, indices, indptr) arrays.  Every entry point
 * below replaces one of those (or the numpy code of one dense branch) and cites
 * it as  <file>:<lines>  relative to /root/reference/xcolumns/.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch-ROCm storage) unless
 *     its name ends in _host; the callee never allocates, frees or synchronises
 *     (xc_utility_finish_host is the one blocking call);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - CSR index arrays are int32, column ids sorted ascending within a row and
 *     distinct (the reference's merges need the same, numba_csr_functions.py:121);
 *   - `dtype` is XC_F32 or XC_F64: the element type of y_proba / y_pred data;
 *   - return value: 0 on success, a negative XC_ERR_* for argument errors, a
 *     positive hipError_t for runtime failures; xc_last_error() gives the text.
 *   - no torch types, no C++ types: plain pointers and sizes only.
 */
#ifndef XCOLUMNS_AMD_H
#define XCOLUMNS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XC_ABI_VERSION 1

enum { XC_F32 = 0, XC_F64 = 1 };

enum {
    XC_OK = 0,
    XC_ERR_BAD_ARG = -1,      /* NULL pointer, negative size, unknown enum        */
    XC_ERR_K_RANGE = -2,      /* k outside 1..XC_MAX_K                            */
    XC_ERR_ROW_TOO_LONG = -3, /* a row holds more than XC_MAX_ROW_NNZ entries     */
    XC_ERR_NO_DEVICE = -4,    /* no gfx950 device visible                         */
};

#define XC_MAX_K 64          /* budget per row handled by the wavefront top-k    */
#define XC_MAX_ROW_NNZ 1024  /* 16 candidates per lane x 64 lanes                */

/* Binary metric evaluated on device: xcolumns/metrics.py binary_*_on_conf_matrix.
 * With `mixed` set the value is (1 - alpha) * tp / kf + alpha * base(...) / mf,
 * the closures of block_coordinate.py:848-1045. */
enum {
    XC_M_PRECISION_AT_K = 0, /* metrics.py:497-513 */
    XC_M_PRECISION = 1,      /* metrics.py:585-605 */
    XC_M_RECALL = 2,         /* metrics.py:633-652 */
    XC_M_FBETA = 3,          /* metrics.py:683-730 (F1 = beta 1) */
    XC_M_JACCARD = 4,        /* metrics.py:778-797 */
    XC_M_BALANCED_ACC = 5,   /* metrics.py:824-845 */
    XC_M_GMEAN = 6,          /* metrics.py:873-894 */
    XC_M_HMEAN = 7,          /* metrics.py:922-944 */
    XC_M_ACCURACY = 8,       /* metrics.py:400-419 */
    XC_M_RECALL_PRECISION_MIX = 9, /* (1 - alpha) * recall + alpha * precision, frank_wolfe.py:925-929 */
    XC_M_COUNT = 10
};

typedef struct xc_metric {
    int32_t base;   /* XC_M_* */
    int32_t mixed;  /* 0 / 1 */
    double epsilon;
    double beta;
    double kf;      /* the k of precision@k */
    double alpha;
    double mf;      /* number of labels, for the mixed utilities */
} xc_metric;

/* Per-label BCA statistics (see DESIGN.md "data layout"):
 *   tpfp    float64[m][2]  one 16-byte record {tp, fp} per label: a candidate label
 *                          costs ONE 16-byte gather in the sweep;
 *   colsum  float64[m]     s = tp + fn = column sum of y_proba over the rows counted
 *                          so far (constant during a non-greedy sweep);
 *   s_entry float64[nnz]   colsum expanded per stored entry of y_proba, so it
 *                          streams in with the row instead of being gathered.
 *   packed  12 B x nnz     optional (float32 scores, m <= 2^25): {col | hot << 25 |
 *                          sel << 31, eta, (float) s} per stored entry -- indices, data,
 *                          sel and s_entry (and the label's hot slot, 0 = none)
 *                          interleaved, so a candidate streams in as ONE 12-byte lane
 *                          load (xc_bca_pack_rows; the sweep keeps its sel bits
 *                          current).  Read by the concurrent sweeps only: s is rounded
 *                          to float32 like the shadow records; the exact sequential
 *                          sweep (n_waves == 1) reads the separate float64 streams.
 *   shadow  float32[m][2]  optional rounded copy of tpfp (8-byte records): what the
 *                          CONCURRENT sweep gathers, so twice as many labels stay in
 *                          an XCD's 4 MiB L2; written by xc_bca_commit_utility, kept
 *                          in step by float32 atomics; the exact sequential sweep
 *                          (n_waves == 1) and the greedy sweep read tpfp itself.
 * fn and tn are derived: fn = s - tp, tn = n_counted - fp - s. */

/* ---- library / device -------------------------------------------------- */

int xc_abi_version(void);
const char *xc_last_error(void);
/* cu_count, waves_per_cu (resident wave slots), and the gcn arch name copied
 * into arch[arch_len].  XC_ERR_NO_DEVICE when no HIP device is usable. */
int xc_device_info(int *cu_count, int *waves_per_cu, char *arch, int arch_len);

/* ---- weighted per-instance top-k -------------------------------------- */

/* numba_predict_weighted_per_instance_csr, k > 0 (numba_csr_functions.py:585-629):
 * gains = data (* a[col]) (+ b[col]) in `dtype`, multiply then add, never fused;
 * top-k per row, ties to the lower column id; out_indices[n*k] ascending within
 * a row; a row with fewer than k entries keeps all of them followed by the
 * reference's padding (column 0, value 1).  out_data[n*k] receives 1 (or the
 * gain when keep_scores).  out_eta (optional, n*k, `dtype`) receives the
 * y_proba value of each chosen entry (0 for padding) -- the BCA driver keeps it
 * next to the prediction.  out_sel (optional, one byte per stored entry of
 * y_proba) receives 1 for the chosen entries and 0 for the others.
 * a, b: optional, length m, already of `dtype`.
 * max_row_nnz: longest row (<= XC_MAX_ROW_NNZ); it selects how many candidates a
 * lane keeps in registers. */
int xc_topk_csr(int64_t n, const int32_t *indptr, const int32_t *indices,
                const void *data, int dtype, int max_row_nnz, int k, const void *a,
                const void *b, int keep_scores, int32_t *out_indices, void *out_data,
                void *out_eta, uint8_t *out_sel, void *stream);

/* k == 0 branch (numba_csr_functions.py:631-653, :516-517): a row keeps the
 * entries whose gain >= th.  Two calls: count -> (caller does the exclusive
 * scan into out_indptr) -> fill. */
int xc_threshold_count_csr(int64_t n, const int32_t *indptr, const int32_t *indices,
                           const void *data, int dtype, double th, const void *a,
                           const void *b, int32_t *out_counts, void *stream);
int xc_threshold_fill_csr(int64_t n, const int32_t *indptr, const int32_t *indices,
                          const void *data, int dtype, double th, const void *a,
                          const void *b, const int32_t *out_indptr,
                          int32_t *out_indices, void *stream);

/* _predict_weighted_per_instance_dense (weighted_prediction.py:25-60).
 * gains[n x m] (row stride ld elements) are ALREADY y_proba*a+b in the promoted
 * dtype `gdtype` (the host forms them with the framework's own promotion rules);
 * y_pred[n x m] (`pdtype`, contiguous) is overwritten: 0 everywhere, 1 (or the
 * gain) at the k best columns of each row, ties to the lower column;
 * k == 0: 1 where gain >= th. */
int xc_topk_dense(int64_t n, int64_t m, int64_t ld, const void *gains, int gdtype,
                  int k, double th, int keep_scores, void *y_pred, int pdtype,
                  void *stream);

/* A dense 0/1 prediction with k ones per row (xc_topk_dense's output) as k column ids per row, ascending, and --
 * when out_val is given -- the entries of `gains` (n x m, row-contiguous) at those columns: the fixed-width CSR
 * form.  experiments/utils.py:198-210 (load_npy_full_pred: dense scores -> top-k CSR) rests on it. */
int xc_dense_pred_to_fixed(int64_t n, int64_t m, const void *y_pred, int pdtype, const void *gains,
                           int gdtype, int k, int32_t *out_idx, void *out_val, void *stream);

/* ---- confusion matrix ---------------------------------------------------- */

/* calculate_confusion_matrix, CSR branch, axis 0 (confusion_matrix.py:364-399 ->
 * numba_csr_functions.py:143-182, :216-258), the three passes fused into one:
 *   tp[j] += p*t, fp[j] += p*(1-t) (p alone when t is absent), fn[j] += t*(1-p).
 * tp/fp/fn: float64[m], ACCUMULATED into (zero them first).  y_pred rows may
 * repeat a column id (the top-k padding); the reference's last-write-wins
 * scatter is reproduced for that case. */
int xc_confusion_csr(int64_t n, int64_t m, const int32_t *t_indptr,
                     const int32_t *t_indices, const void *t_data,
                     const int32_t *p_indptr, const int32_t *p_indices,
                     const void *p_data, int dtype, double *tp, double *fp,
                     double *fn, void *stream);

/* The same statistics with atomics for the PREDICTED entries only, given the column sums of y_true:
 *   fn[j] = colsum_t[j] - sum over the entries of y_true that meet a predicted one of [t - (T)(t (1 - p))].
 * The caller puts colsum_t (float64 sums of y_true's columns; it depends on y_true alone and can be kept) into fn and
 * zeros into tp, fp; nnz(y_pred) + 2 matches atomics instead of nnz(y_pred) + nnz(y_true).  Both matrices need rows of
 * strictly ascending column ids (numba_csr_functions.py:121): y_true's are the caller's to check
 * (xc_csr_rows_ascending: *flag |= 1 if some row is not), a row of y_pred that is not -- the top-k padding of a
 * short row -- sets *flag |= 1 and the results are to be discarded (xc_confusion_csr reproduces that case). */
int xc_csr_rows_ascending(int64_t n, const int32_t *indptr, const int32_t *indices, int32_t *flag,
                          void *stream);
int xc_confusion_csr_pred_side(int64_t n, int64_t m, const int32_t *t_indptr,
                               const int32_t *t_indices, const void *t_data,
                               const int32_t *p_indptr, const int32_t *p_indices,
                               const void *p_data, int dtype, double *tp, double *fp, double *fn,
                               int32_t *flag, void *stream);

/* The prediction-side form for a 0/1 prediction without global atomics: val[q] <- y_true's value at the label of predicted
 * entry q (0 where the row does not store it), float32.  Then xc_scatter_sum_f32(nnz(y_pred), p_indices, val, m, pair = 1)
 * gives {tp, fp} per label (tp = sum val, fp = sum (1 - val) in float32: (T)(p (1 - t)) with p = 1,
 * numba_csr_functions.py:197-206) and fn = column sums of y_true - tp.  *flag |= 1: a y_pred row is not strictly
 * ascending, |= 2: a stored y_pred value is not 1 (take xc_confusion_csr_pred_side / xc_confusion_csr then). */
int xc_confusion_csr_match(int64_t n, const int32_t *t_indptr, const int32_t *t_indices, const float *t_data,
                           const int32_t *p_indptr, const int32_t *p_indices, const float *p_data, float *val,
                           int32_t *flag, void *stream);

/* dense branch (confusion_matrix.py:160-166, :187-202): products in `dtype`,
 * column sums in float64.  y_true, y_pred: n x m contiguous. */
int xc_confusion_dense(int64_t n, int64_t m, const void *y_true, const void *y_pred,
                       int dtype, double *tp, double *fp, double *fn, void *stream);

/* ---- block coordinate ascent, CSR ------------------------------------- */

/* packed[p] <- {indices[p] | hot_slot[indices[p]] << 25 | sel[p] << 31, data[p], (float) s_entry[p]}
 * (three 32-bit words per entry) for float32 scores (column ids < 2^25).  hot_slot: optional uint8[m], 1..63 for the
 * labels whose deltas a sweeping wave batches (see xc_bca_sweep_csr `hot_labels`), 0 else. */
int xc_bca_pack_rows(int64_t nnz, const int32_t *indices, const float *data,
                     const uint8_t *sel, const double *s_entry, const uint8_t *hot_slot,
                     void *packed, void *stream);

/* For a prediction given as column ids (pred_indices[n*k], k per row), look up
 * each id in its row of y_proba: pred_eta[n*k] <- the stored value, or 0 when the
 * row does not hold that column (the reference treats it as eta = 0,
 * numba_csr_functions.py:200-203).  sel (optional, one byte per stored entry,
 * zeroed by the caller) gets 1 at the entries found; orphans (optional, n*k)
 * gets the column id of every predicted column its row does NOT store, else -1. */
int xc_bca_gather_pred_eta(int64_t n, const int32_t *indptr, const int32_t *indices,
                           const void *data, int dtype, const int32_t *pred_indices,
                           int k, void *pred_eta, uint8_t *sel, int32_t *orphans,
                           void *stream);

/* colsum[j] += sum of the stored entries of column j (one-off per run; colsum
 * zeroed by the caller; all-reduce it when the rows are sharded). */
int xc_bca_colsum_csr(int64_t nnz, const int32_t *indices, const void *data,
                      int dtype, double *colsum, void *stream);

/* The packed stream straight from the per-label column sums (a gather of colsum[indices[p]] instead of
 * a read of s_entry[p]): a run whose sweeps all read the packed stream then never builds s_entry. */
int xc_bca_pack_rows_from_colsum(int64_t nnz, const int32_t *indices, const float *data,
                                 const uint8_t *sel, const double *colsum, const uint8_t *hot_slot,
                                 void *packed, void *stream);

/* s_entry[p] = colsum[indices[p]] for every stored entry p. */
int xc_bca_expand_colsum(int64_t nnz, const int32_t *indices, const double *colsum,
                         double *s_entry, void *stream);

/* Expected tp / fp of the current prediction, from scratch:
 *   acc[j*2+0] += eta, acc[j*2+1] += (1 - eta) for every predicted (row, j)
 * -- the tp and fp passes of calculate_confusion_matrix(y_proba, y_pred)
 * (block_coordinate.py:430-436); fn follows from s.  acc: float64 [2m (+1)],
 * accumulated into (zero first; all-reduce it across ranks when the rows are
 * sharded).  Needed for the initial statistics only: later sweeps fill acc
 * themselves (xc_bca_sweep_csr). */
int xc_bca_accumulate_pred(int64_t n_k, const int32_t *pred_indices,
                           const void *pred_eta, int dtype, double *acc,
                           void *stream);

/* tpfp[j] <- acc[j] and, in the same pass, the per-label metric values of
 * _calculate_utility (block_coordinate.py:54-90) on (tp/n, fp/n, fn/n, tn/n) with
 * fn = s - tp and tn = n_counted - fp - s (tn = -1 when skip_tn,
 * confusion_matrix.py:391-393), reduced to partials[XC_UTILITY_PARTIALS] in a
 * fixed order.  acc may be NULL (evaluate tpfp as it stands).  With acc:
 * partials[XC_UTILITY_PARTIALS] <- acc[2m] (the changed-row count the sweep left
 * there) and, when clear_acc, acc is zeroed for the next sweep.
 * partials: float64[XC_UTILITY_PARTIALS + 1]. */
#define XC_UTILITY_PARTIALS 1024
int xc_bca_commit_utility(int64_t m, int64_t n_norm, double n_counted, double *acc,
                          int clear_acc, double *tpfp, float *shadow, const double *colsum,
                          const xc_metric *metric_host, int skip_tn,
                          double *partials, void *stream);
/* Blocking: waits for `stream`, sums the partials in index order on the host and
 * writes the sum to *out_host (divide by m for "mean"); *out_extra_host (optional)
 * receives partials[XC_UTILITY_PARTIALS]. */
int xc_utility_finish_host(const double *partials, double *out_host,
                           double *out_extra_host, void *stream);

/* One sweep of block_coordinate.py:448-463 with _bc_with_0approx_step_csr
 * (:212-293) as the body, for k > 0 and rows holding >= k entries.
 *   order        int32[n_order] row ids in visiting order, or NULL for 0..n_order-1
 *   n_norm       the divisor `n` of the step (:229-231)
 *   pred_*       the prediction (k column ids + their y_proba values per row),
 *                rewritten for the rows that change
 *   sel          one byte per stored entry of y_proba: 1 = in the prediction; this
 *                is what the sweep reads to know a row's current prediction
 *   orphans      optional [n*k] from xc_bca_gather_pred_eta: predicted columns a
 *                row does not store; they leave the prediction when the row is
 *                visited (pass it for the first sweep after a foreign
 *                initialisation, NULL afterwards)
 *   m            number of labels
 *   tpfp/colsum  per-label statistics, updated with float64 atomics
 *   shadow       optional float32 copy of tpfp (see above); NULL = gather tpfp
 *   s_entry      colsum per stored entry (xc_bca_expand_colsum); may be NULL when
 *                greedy (colsum is then gathered and grows during the sweep)
 *   packed       optional packed row stream (float32 scores, non-greedy, n_waves > 1); NULL = read
 *                indices / data / sel / s_entry separately.  A sweep that does not read it
 *                (greedy, n_waves == 1) leaves its sel bits stale: call xc_bca_pack_rows again
 *                before the next sweep that does
 *   hot_labels   optional int32[64] (with packed, shadow and acc): label id of hot slot
 *                h = 1..63 (-1 = unused; entry 0 unused).  A wave sums its deltas to these
 *                labels in LDS and publishes them every few rows as one atomic per label
 *                (same-address float atomics serialise at ~11 ns); they are exempt from
 *                the optimistic validation
 *   acc          optional float64[2m + 1], zeroed by the caller: the sweep adds every
 *                visited row's NEW prediction into it ({tp, fp} per label) -- when
 *                all rows are visited this IS the sweep-boundary recompute
 *                (block_coordinate.py:465-467), no second pass -- and the number of
 *                changed rows into acc[2m].  With `shadow` AND `acc` (concurrent sweep)
 *                the float64 records `tpfp` are left as they were at the sweep start:
 *                the sweep reads only the shadow, and xc_bca_commit_utility rewrites
 *                tpfp and shadow from acc afterwards
 *   greedy       first sweep of init_y_pred="greedy": rows are added as they are
 *                visited (:243 skipped, stats start from zero)
 *   n_waves      number of wavefronts that walk `order` concurrently: wave w takes
 *                positions w, w + n_waves, ...  1 = the reference's exact
 *                sequential sweep; larger values bound how many rows see
 *                statistics that miss each other's update (DESIGN.md "staleness")
 *   changed      optional int64[1], += number of rows whose prediction changed */
int xc_bca_sweep_csr(int64_t n_order, const int32_t *order, int64_t n_norm,
                     const int32_t *indptr, const int32_t *indices, const void *data,
                     int dtype, int max_row_nnz, int32_t *pred_indices,
                     void *pred_eta, uint8_t *sel, const int32_t *orphans, int k,
                     int64_t m, double *tpfp, float *shadow, double *colsum,
                     const double *s_entry, void *packed, const int32_t *hot_labels,
                     double *acc, const xc_metric *metric_host, int maximize, int greedy,
                     int skip_tn, int n_waves, int64_t *changed, void *stream);

/* Plan: bind the per-run constants of a CSR BCA once; afterwards a sweep and its boundary
 * are two short calls (same kernels as xc_bca_sweep_csr / xc_bca_commit_utility +
 * xc_utility_finish_host; it only saves argument marshalling in the host language).
 * The buffers stay owned by the caller and must outlive the plan. */
int xc_bca_plan_create(void **plan, int64_t n, int64_t m, int64_t n_total,
                       const int32_t *indptr, const int32_t *indices, const void *data,
                       int dtype, int max_row_nnz, int k, int32_t *pred_indices,
                       void *pred_eta, uint8_t *sel, double *tpfp, float *shadow,
                       double *colsum, const double *s_entry, void *packed,
                       const int32_t *hot_labels, double *acc, double *partials,
                       const xc_metric *gain_metric,
                       const xc_metric *utility_metric, int maximize, int skip_tn);
int xc_bca_plan_destroy(void *plan);
/* with_acc: accumulate the new prediction into acc (pass 1 when every row is visited) */
int xc_bca_plan_sweep(void *plan, const int32_t *order, int64_t n_order,
                      const int32_t *orphans, int greedy, int n_waves, int with_acc,
                      int use_packed, int64_t *changed, void *stream);
/* commit (acc -> tpfp / shadow, clear acc) when `commit`, utility partials, blocking D2H */
int xc_bca_plan_boundary(void *plan, int64_t n_norm_utility, double n_counted, int commit,
                         int skip_tn, double *out_sum_host, double *out_extra_host,
                         void *stream);

/* Measurement helpers (bench.py).  HIP events owned by the library; the pair given to
 * xc_bca_time_next_sweep is attached to the NEXT xc_bca_sweep_csr dispatch of the calling
 * thread (hipExtLaunchKernelGGL start/stop events), so xc_event_elapsed_ms returns the
 * kernel's own duration on its stream -- what rocprofv3 --kernel-trace reports --
 * rather than kernel + dispatch gap. */
int xc_event_create(void **ev);
int xc_event_destroy(void *ev);
int xc_event_elapsed_ms(void *start, void *stop, float *ms_host);
int xc_bca_time_next_sweep(void *start, void *stop);
/* The event pair pending from xc_bca_time_next_sweep spans the next `launches` sweep launches instead of one: row shards walk
 * a sweep in parts with exchanges in between (start rides on the first part's dispatch, stop on the last one's).  No-op
 * when no pair is pending. */
int xc_bca_time_span(int launches);

/* How a concurrent sweep guards a row that CHANGES its prediction against other rows in flight
 * that change the same labels (process-wide switch, mainly for studies):
 *   2 (default)  commit protocol: the deltas of the labels the row adds or drops are pushed with
 *                RETURNING atomics and the returned values compared with the records the row was
 *                scored on; a difference means another row changed that label first, and the row is
 *                processed again on fresh records (its new prediction is what memory holds by then);
 *   1            round-1 form: re-read those records before committing, re-score if they moved;
 *   0            none. */
int xc_bca_set_validation(int mode);

/* Where a pipelined concurrent sweep leaves the statistics of its new prediction (block_coordinate.py:465-467
 * recomputes them from scratch at every boundary).  on = 1 (default): with the commit protocol every committed
 * change is pushed into the float64 records tpfp (next to the returning float32 atomics on the shadow records; the
 * sweeps that gather tpfp itself commit there anyway) -- exact float64 sums of float32 values, atomics for the
 * rows that change only -- and the boundary reads them there.  on = 0: every
 * row adds its k labels to `acc` (from scratch, as round 1 did: one 16-byte atomic per predicted label, 19 % of
 * a converged sweep at 1 M x 500 K).  xc_bca_plan_delta: does this plan's next pipelined sweep use the first form. */
int xc_bca_set_acc_delta(int on);
int xc_bca_plan_delta(void *plan);
/* Row shards with the first form: d[0..m2) = tpfp - base, d[m2] = count[0] (this rank's changed rows) before the
 * all-reduce of d; tpfp = base = base + d, count[0] = d[m2] after it (m2 = 2 m). */
int xc_bca_delta_pack(int64_t m2, const double *tpfp, const double *base, const double *count, double *d,
                      void *stream);
int xc_bca_delta_unpack(int64_t m2, double *tpfp, double *base, double *count, const double *d, void *stream);

/* Two constants of the concurrent sweep (process-wide, studies; negative = leave as is):
 *   conflict_rel     commit protocol: a returned record that differs from the scored one by less than this
 *                    share of (tp + fp) does not count as a conflict (default 1/512: one other row changing a
 *                    label that holds fewer than 512 predicted rows re-processes this row);
 *   hot_unpublished  hot labels: share of the rows whose deltas may wait in the workgroups' LDS tables
 *                    (default 0.05; sets how often a narrow sweep publishes them). */
int xc_bca_set_tuning(double conflict_rel, double hot_unpublished);

/* Host-side: one `Generator.shuffle(order)` of numpy's PCG64 stream -- the reference's visiting order of a sweep
 * (block_coordinate.py:413-419) -- in place on an int32 array of n entries (host memory).  state_io =
 * {state_hi, state_lo, inc_hi, inc_lo} of rng.bit_generator.state["state"], has_uint32_io / uinteger_io the
 * generator's buffered 32-bit half; all are left as numpy would leave them.  Same permutation as numpy, less time
 * (draws generated a block ahead, the swap partner prefetched). */
int xc_host_shuffle_pcg64(uint64_t *state_io, int *has_uint32_io, uint32_t *uinteger_io, int64_t n,
                          int32_t *order);
/* The same walk in two halves, for two host threads: the partners of all n - 1 steps (js[t] belongs to step
 * i = n-1-t; advances the generator exactly like the shuffle), then the swaps. */
int xc_host_shuffle_draws(uint64_t *state_io, int *has_uint32_io, uint32_t *uinteger_io, int64_t n,
                          uint32_t *js);
int xc_host_shuffle_apply(int64_t n, const uint32_t *js, int32_t *order);

/* Per-label sums of n_items (label, float32 value) pairs without a global float atomic per pair
 * (csrc/xc_scatter.hip: counting sort into buckets of consecutive labels, then one workgroup per bucket sums in LDS).
 * pair = 0: out[label] += value (the column sums of y_proba, numba_csr_functions.py:143-182 over all rows);
 * pair = 1: out[2 label] += value, out[2 label + 1] += 1 - value (float32 subtraction: the from-scratch {tp, fp} of a
 * prediction, block_coordinate.py:430-436).  Every label of [0, m) is written.  `workspace`: device memory of
 * xc_scatter_sum_workspace_bytes(n_items, m) bytes. */
int xc_scatter_sum_workspace_bytes(int64_t n_items, int64_t m, int64_t *bytes);
int xc_scatter_sum_f32(int64_t n_items, const int32_t *idx, const float *val, int64_t m, int pair,
                       double *out, void *workspace, void *stream);

/* Busy labels: counts[m] <- how often each label occurs among every `stride`-th stored entry (indices[0], indices[stride],
 * ...), then list <- (label, sampled count) pairs of the labels with sampled count >= min_count, at most `cap` of
 * them, in arrival order; *n_list <- how many reached the threshold.  The engine picks its <= 63 hot labels from it. */
int xc_label_busy_list(int64_t nnz, const int32_t *indices, int64_t stride, int64_t m, int min_count, int cap,
                       int32_t *counts, int32_t *list, int32_t *n_list, void *stream);

/* Row shards, overlapped mid-sweep exchange (xcolumns_amd/block_coordinate.py:pipeline_step): one element-wise step
 * over the n = 2m float32 record values.  fold != 0: first take in the other ranks' part of the exchange issued one
 * step earlier (records += sum - mine, base likewise; `sum` = the all-reduced changes, `mine` = this rank's share
 * of them).  Then publish: mine = sum = records - base (what this rank's rows changed since its last publication),
 * base = records.  The caller starts an asynchronous all-reduce on `sum` and lets the next segment of the sweep run. */
int xc_bca_exchange_step(int64_t n, float *records, float *base, float *sum, float *mine, int fold,
                         void *stream);

/* Unpack the per-label statistics into the reference's four vectors
 * (tp, fp, fn, tn: float64[m]); tn = -1 when skip_tn. */
int xc_bca_state_unpack(int64_t m, const double *tpfp, const double *colsum,
                        double n_counted, int skip_tn, double *tp, double *fp, double *fn,
                        double *tn, void *stream);

/* ---- block coordinate ascent, dense ------------------------------------ */

/* One sweep of block_coordinate.py:448-463 with _bc_with_0approx_step_dense
 * (:132-209): y_proba, y_pred n x m contiguous of `dtype`; y_pred updated in
 * place; k == 0 predicts every label with non-negative gain (:199-200).
 * stats: float64[4*m] as four vectors tp | fp | fn | tn, updated in place.
 * workspace: float64[m] scratch.  Rows are visited strictly in `order` (one
 * workgroup walks them), m <= 65536. */
int xc_bca_sweep_dense(int64_t n_order, const int32_t *order, int64_t n_norm,
                       int64_t m, const void *y_proba, void *y_pred, int dtype, int k,
                       double *stats, double *workspace, const xc_metric *metric_host,
                       int maximize, int greedy, int skip_tn, void *stream);

/* The same step with `n_blocks` workgroups walking `order` concurrently (block b takes
 * positions b, b + n_blocks, ...; non-greedy sweeps, m <= 8192): statistics are read
 * coherently and only the labels whose prediction flips are pushed back, as float64
 * atomics.  Rows in flight miss each other's update, exactly as in xc_bca_sweep_csr
 * with n_waves > 1.  changed: optional int64[1], += rows whose prediction changed. */
int xc_bca_sweep_dense_concurrent(int64_t n_order, const int32_t *order, int64_t n_norm,
                                  int64_t m, const void *y_proba, void *y_pred, int dtype, int k,
                                  double *stats, const xc_metric *metric_host, int maximize,
                                  int skip_tn, int n_blocks, int64_t *changed, void *stream);

/* Utility of four plain vectors (dense path): partials as above. */
int xc_utility_vectors(int64_t m, int64_t n_norm, const double *stats,
                       const xc_metric *metric_host, double *partials, void *stream);

/* ---------------------------------------------------------------------------
 * The sweep loop without a host round trip per iteration.
 *
 * predict_using_bc_with_0approx decides after every sweep whether to run another
 * (block_coordinate.py:486-493).  Here that rule -- and the wavefront-count policy
 * of DESIGN.md "staleness" -- is evaluated on the GPU at the sweep boundary and
 * left in a small control block; sweep j + 1 can then be enqueued before the host
 * has seen the utility of sweep j: if the rule fired, the later launches find the
 * stop flag and do nothing.  The host reads every boundary's result (utility sum,
 * changed rows, wavefronts used, stop flag) one iteration late from a pinned ring.
 *
 * ctrl: float64[XC_CTRL_SIZE] on the device.  Ring slot s (s < XC_CTRL_RING_SLOTS) is
 * XC_CTRL_RING_STRIDE doubles at ctrl[XC_CTRL_RING + XC_CTRL_RING_STRIDE s]: {utility sum,
 * rows changed, wavefronts of the sweep, 0 = continue | 1 = rule fired | 2 = step
 * skipped | 3 = paused: the rule left the NEXT sweep fewer than exact_below wavefronts, it is the
 * host's to run exactly (the steps enqueued behind report 2), sequence number (host ring only)}.  The host ring (pinned, host-mapped:
 * xc_host_alloc_pinned of XC_CTRL_RING_STRIDE * XC_CTRL_RING_SLOTS doubles) has the
 * same slots; the boundary kernel stores into it directly, sequence number last.
 * ------------------------------------------------------------------------- */
#define XC_CTRL_STOP 0
#define XC_CTRL_OLD_SUM 1
#define XC_CTRL_WAVES 2
#define XC_CTRL_TOLERANCE 3
#define XC_CTRL_DIVISOR 4
#define XC_CTRL_MAXIMIZE 5
#define XC_CTRL_POLICY_NUM 6
#define XC_CTRL_WORLD 7
#define XC_CTRL_MIN_WAVES 8
#define XC_CTRL_MAX_WAVES 9
#define XC_CTRL_FIXED_WAVES 10
#define XC_CTRL_EXACT_BELOW 11 /* a sweep the rule leaves fewer wavefronts than this is the host's to run exactly: the loop pauses */
#define XC_CTRL_RING 16
#define XC_CTRL_RING_SLOTS 8
#define XC_CTRL_RING_STRIDE 8 /* doubles per slot: 4 results, the sequence number, padding */
#define XC_CTRL_SIZE (XC_CTRL_RING + XC_CTRL_RING_STRIDE * XC_CTRL_RING_SLOTS)

/* Arm the control block: `old_utility_sum` = utility sum of the current prediction;
 * the rule is (new/divisor - old/divisor < tolerance) for maximize, (> tolerance)
 * otherwise (divisor = m for metric_aggregation "mean", 1 for "sum").  Wavefronts of
 * the next sweep = fixed_waves if > 0, else
 * clamp(floor(policy_num / max(1, changed / world)), min_waves, max_waves) (changed = rows
 * changed over all ranks);
 * `first_waves` is used by the first sweep; `exact_below` (0: never): see flag 3 above. */
int xc_bca_pipeline_begin(double *ctrl, double old_utility_sum, double tolerance, double divisor,
                          int maximize, double policy_num, int world, int min_waves,
                          int max_waves, int fixed_waves, int first_waves, int exact_below, void *stream);

/* Positions [first, first + count) of a full, non-greedy sweep over `order`
 * (xc_bca_plan_sweep with acc), launched for max_waves wavefronts; runs only if the
 * stop flag is clear, with ctrl's count.  A sweep is one call with (0, n) or several
 * consecutive segments that together cover the order (sharded rows: the ranks exchange
 * their changes between segments); `acc` is complete after the last one. */
int xc_bca_plan_sweep_pipelined(void *plan, const int32_t *order, int64_t first, int64_t count,
                                int use_packed, int max_waves, const double *ctrl, void *stream);

/* The boundary after it (the caller all-reduces acc in between when rows are
 * sharded): commit + utility partials, then the rule and the policy on the GPU; the
 * results go to ring slot `slot` on the device and, stamped with `seq`, in the host
 * ring.  Nothing here blocks, no copy and no event enter the stream. */
int xc_bca_plan_boundary_pipelined(void *plan, int64_t n_norm_utility, double n_counted,
                                   int skip_tn, double *ctrl, int slot, double *host_ring,
                                   double seq, void *stream);

/* Wait (spinning) until host ring slot `slot` carries sequence number `seq`; fails if
 * `stream` reports an error or drains without producing it, or after timeout_ms. */
int xc_bca_ring_wait(const double *host_ring, int slot, double seq, double timeout_ms,
                     void *stream);

int xc_event_synchronize(void *event);
int xc_host_alloc_pinned(void **ptr, int64_t bytes);
int xc_host_free_pinned(void *ptr);

/* ---------------------------------------------------------------------------
 * Frank-Wolfe search for a randomized weighted classifier
 * (xcolumns/frank_wolfe.py:407-690).  An iteration is
 * xc_topk_csr with weights (a, b) -> xc_confusion_csr against y_true -> the two
 * O(m) steps below.  stats / cur / nxt: float64[4*m] as tp | fp | fn | tn.
 * ------------------------------------------------------------------------- */

/* Gradient of the utility in the confusion entries, folded into the next weighted
 * classifier (frank_wolfe.py:585-596; replaces autograd.grad, :368-376):
 *   G_x[j] = d metric(tp_j, fp_j, fn_j, tn_j) / d x / div      (div = m for a macro
 *            average -- the mean's 1/m --, 1 for a sum)
 *   a[j] = G_tp - G_fp - G_fn + G_tn,  b[j] = G_fp - G_tn,  both negated if `negate`
 * (minimisation, :594-596).  For a micro average pass the four label sums as an
 * m = 1 problem and broadcast the result. */
int xc_fw_gradient(int64_t m, const double *stats, const xc_metric *metric_host, double div,
                   int negate, double *a, double *b, void *stream);

/* Utility along the segment between two confusion matrices (frank_wolfe.py:379-404,
 * _find_best_alpha): for every alphas[t]
 *   sum_j metric((1 - alpha) * cur_j + alpha * nxt_j)
 * left as partial sums partials[c * n_alpha + t], c < xc_fw_alpha_chunks(m), to be
 * added over c in ascending order.  The caller picks the step (first maximum for the
 * uniform search, utils.py:174-184; two points per step for the ternary one,
 * :187-201).  Up to 16 points are evaluated with the reference's exact expression
 * and IEEE division; a longer scan mixes with one fma per entry and divides to
 * ~1 ulp (differences of the size of the summation-order ones). */
int xc_fw_alpha_chunks(int64_t m);
int xc_fw_alpha_curve(int64_t m, const double *cur, const double *nxt,
                      const xc_metric *metric_host, int n_alpha, const double *alphas,
                      double *partials, void *stream);

/* Counts for BINARY y_true (sorted distinct column ids per row, values all 1) against a
 * fixed-stride 0/1 prediction of k DISTINCT labels per row: tp[j] += rows that predict and hold
 * label j, cnt[j] += rows that predict j.  Then fp = cnt - tp and fn = (label count of y_true) - tp:
 * the same numbers as xc_confusion_csr with less than half its atomics (the Frank-Wolfe iteration). */
int xc_confusion_counts_csr(int64_t n, int k, const int32_t *t_indptr, const int32_t *t_indices,
                            const int32_t *p_indices, double *tp, double *cnt, void *stream);

/* xc_topk_csr with the weights interleaved, ab[2 col] = a[col], ab[2 col + 1] = b[col]
 * (y_proba's dtype): one gather per candidate instead of two.  Same gains bit for bit. */
int xc_topk_csr_ab(int64_t n, const int32_t *indptr, const int32_t *indices, const void *data,
                   int dtype, int max_row_nnz, int k, const void *ab, int keep_scores,
                   int32_t *out_indices, void *out_data, void *out_eta, uint8_t *out_sel,
                   void *stream);

/* xc_topk_csr / xc_threshold_*_csr with one weighted classifier PER ROW
 * (predict_using_randomized_weighted_classifier, frank_wolfe.py:127-172): row i
 * uses a[row_classifier[i] * ld + col], b[...] (tables of y_proba's dtype). */
int xc_topk_csr_rowwise(int64_t n, const int32_t *indptr, const int32_t *indices,
                        const void *data, int dtype, int max_row_nnz, int k, const void *a,
                        const void *b, int64_t ld, const int32_t *row_classifier,
                        int32_t *out_indices, void *stream);
int xc_threshold_count_csr_rowwise(int64_t n, const int32_t *indptr, const int32_t *indices,
                                   const void *data, int dtype, double th, const void *a,
                                   const void *b, int64_t ld, const int32_t *row_classifier,
                                   int32_t *out_counts, void *stream);
int xc_threshold_fill_csr_rowwise(int64_t n, const int32_t *indptr, const int32_t *indices,
                                  const void *data, int dtype, double th, const void *a,
                                  const void *b, int64_t ld, const int32_t *row_classifier,
                                  const int32_t *out_indptr, int32_t *out_indices, void *stream);

/* ---------------------------------------------------------------------------
 * Block coordinate ascent for coverage (block_coordinate.py:600-701).
 * ef: float64[m], Ef_j = prod_i (1 - pred_ij * eta_ij), the probability that no row
 * covers label j; utility = 1 - mean(Ef) (mixed with precision@k when alpha < 1).
 * ------------------------------------------------------------------------- */

/* One sweep of _bc_for_coverage_step_csr (:539-582) over `order`: per row divide its own
 * factors out of Ef, gain = Ef * eta (alpha * gain + (1 - alpha) * eta / k when alpha < 1),
 * keep the k largest (ties: lower column), multiply the new factors in.  pred_indices /
 * pred_eta / sel as in xc_bca_sweep_csr (rows hold >= k entries).  n_waves = 1 is the
 * reference's sequence bit for bit; more wavefronts update Ef with compare-and-swap
 * multiplies for the labels that enter or leave a prediction.  greedy: the first sweep of
 * init_y_pred="greedy" (Ef starts from ones, nothing is divided out). */
int xc_coverage_sweep_csr(int64_t n_order, const int32_t *order, const int32_t *indptr,
                          const int32_t *indices, const void *data, int dtype, int max_row_nnz,
                          int32_t *pred_indices, void *pred_eta, uint8_t *sel, int k, double *ef,
                          double alpha, int greedy, int n_waves, int64_t *changed, void *stream);

/* Ef from scratch (numba_calculate_prod_csr_mat_mul_ones_minus_mat, numba_csr_functions.py:324-382):
 * ef must hold ones; every predicted entry multiplies (1 - eta) into its label (eta = 0:
 * a predicted label the row does not store, no factor). */
int xc_coverage_product(int64_t n_k, const int32_t *pred_indices, const void *pred_eta, int dtype,
                        double *ef, void *stream);

/* ---------------------------------------------------------------------------
 * The visiting order generated ON the GPU (csrc/xc_order_dev.hip): numpy's `Generator.shuffle` stream -- the
 * reference's `np.random.default_rng(seed)`, one array shuffled cumulatively once per sweep
 * (block_coordinate.py:413-419) -- with no host arithmetic and no host synchronisation per sweep: the PCG64
 * outputs by jump-ahead, the masked rejection settled for all batches of 8192 candidates at once in rounds (a batch's
 * entering bound from what the batches before it kept in the previous round -- a workgroup per batch that waits only
 * for lower-numbered workgroups, no grid barrier: the fixed point is the sequential walk, reached in ~18 rounds at
 * 1 M rows), the Fisher-Yates swaps resolved in parallel.
 * Same permutation and same generator position as numpy (the wrapper checks both against numpy at first use).
 * ------------------------------------------------------------------------- */

/* 32-bit candidates a shuffle of n entries may draw (expectation of the masked rejection + 8 sigma). */
int xc_order_dev_candidates(int64_t n, int64_t *count);
int xc_order_dev_workspace_bytes(int64_t n, int64_t *bytes);
/* state_inc = {state_hi, state_lo, inc_hi, inc_lo} of rng.bit_generator.state["state"]: the NEXT 64-bit output is the
 * XSL-RR of the state stepped once.  consumed = 0, or 1 when the generator holds a buffered 32-bit half (pass the
 * state stepped BACK once then).  order (int32[n], device) <- 0 .. n - 1. */
int xc_order_dev_begin(void *workspace, const uint64_t *state_inc, int consumed, int64_t n, int32_t *order, void *stream);
/* order_out <- one Generator.shuffle of order_in (int32[n] each, device, distinct); asynchronous on `stream`. */
int xc_order_dev_shuffle(void *workspace, int64_t n, const int32_t *order_in, int32_t *order_out, void *stream);
/* The same in two halves, for two streams: the draw (candidate stream, rejection walk: the swap partners of all steps into
 * set `slot` (0 / 1) of the workspace; advances the generator's position -- the next draw needs nothing else of this
 * shuffle) and the apply (the Fisher-Yates swaps with the partners of `slot`).  The caller orders apply k behind draw k and
 * draw k + 2 behind apply k (events). */
int xc_order_dev_draw(void *workspace, int64_t n, int slot, void *stream);
int xc_order_dev_apply(void *workspace, int64_t n, int slot, const int32_t *order_in, int32_t *order_out, void *stream);
/* out8_host = {failure flag (0: every shuffle so far is numpy's), 32-bit draws consumed, shuffles, shader cycles and
 * 100 MHz ticks of the last rejection walk, its rounds and batches, walks redone by the one-wavefront fallback};
 * blocks on `stream`. */
int xc_order_dev_status(void *workspace, int64_t *out8_host, void *stream);
/* Test knob: rounds the grid-wide rejection walk may take (0 = default, 96; 16-45 are needed).  With too few it gives up and
 * the one-wavefront walk behind it redoes the shuffle: the same permutation, ~8 ms per million rows.  Negative: the
 * one-wavefront walk only. */
int xc_order_dev_set_rounds(int rounds);
/* Diagnostics of the last rejection walk, per batch of 8192 candidates: {start, end (100 MHz ticks since the shuffle
 * began), ticks waiting << 32 | ticks settling, rounds << 40 | settles << 24 | inner rounds}; out = NULL: only the number
 * of batches.  Blocks on `stream`. */
int xc_order_dev_walk_trace(void *workspace, int64_t n, int64_t *out, int64_t *batches_out, void *stream);

/* ---------------------------------------------------------------------------
 * The ORDERED parallel sweep (csrc/xc_bca_ord.hip): the loop of block_coordinate.py:448-463 with thousands of
 * rows in flight AND its sequential semantics -- every row decides on the statistics as the rows BEFORE it in
 * the visiting order left them.  Replaces the one-wavefront sweep wherever the reference's own trajectory is
 * wanted (xc_bca_sweep_csr with n_waves = 1 computes the same predictions one row at a time).
 * The order is walked in windows of (workgroups x 16) rows; inside a window the rows iterate on
 * "committed records + changes of the earlier rows of the window" until no decision moves (the fixed point is
 * the sequential result), then the window commits.  One launch per sweep.
 * ------------------------------------------------------------------------- */

/* Workgroups of an ordered sweep on this device (one 1024-thread workgroup = 16 wavefronts per CU) and the rows in
 * flight with ONE row per wavefront; a sweep holds rows_per_wave times as many. */
int xc_bca_ord_window(int *workgroups, int *window);

/* Bytes of the workspace of xc_bca_ord_sweep: m labels, `total_cap` change-list entries over all labels,
 * n_hot (<= 255) labels with dense tables, windows of `window` = workgroups x 16 x rows_per_wave rows. */
int xc_bca_ord_workspace_bytes(int64_t m, int64_t total_cap, int n_hot, int window, int64_t *bytes);

/* One full sweep over `order` (NULL: rows 0 .. n_order - 1); arguments as xc_bca_sweep_csr where the names agree
 * (tpfp: the float64 records of the CURRENT prediction, updated in place; s_entry: the column sum per stored
 * entry).  workspace: zeroed once by the caller, and again after an error.  lab_dir[m][2]: per label {offset of
 * its change list in entries, capacity} or {-(h + 1), 0} for the label of hot slot h (hot_labels[h]): a label
 * gets a dense table when a window holds many rows that store it.  rows_per_wave: 1, 2 or 4 rows per wavefront
 * (rows x candidates per lane <= 4 fit the registers: lowered for rows of more than 64 entries); the workspace must
 * have been sized for the window this gives.  epoch0 grows by 2^20 from launch to launch on one workspace (lists
 * are tagged, never cleared).  changed (device, optional): += rows whose prediction changed.
 * status_host[8] (the call blocks on the stream): [0] positions of the order committed; [1] error -- 0, or 1: a
 * change list overflowed, positions from status_host[0] on are untouched and left to the caller (xc_bca_sweep_csr
 * with n_waves = 1 continues the same sweep exactly), 2: barrier timeout, 3: iteration limit; [2] iterations;
 * [3] windows; [4] / [5] 100 MHz ticks workgroup 0 spent in grid barriers / in the kernel; [6] rows per wavefront
 * used. */
int xc_bca_ord_sweep(void *workspace, int64_t n_order, const int32_t *order, int64_t n_norm, const int32_t *indptr,
                     const int32_t *indices, const void *data, int dtype, int max_row_nnz, int32_t *pred_indices,
                     void *pred_eta, uint8_t *sel, const int32_t *orphans, int k, int64_t m, double *tpfp,
                     const double *s_entry, const int32_t *lab_dir, int64_t total_cap, const int32_t *hot_labels,
                     int n_hot, int workgroups, int rows_per_wave, const xc_metric *metric_host, int maximize,
                     int skip_tn, unsigned epoch0, int64_t *changed, int64_t *status_host, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* XCOLUMNS_AMD_H */

// xc_confusion.hip -- per-label confusion statistics (tp / fp / fn column sums).
//
// Replaces calculate_confusion_matrix's CSR and dense branches, axis 0
// (/root/reference/xcolumns/confusion_matrix.py:364-399):
//   CSR   -> numba_calculate_sum_csr_mat_mul_mat (numba_csr_functions.py:143-182)
//            and numba_calculate_sum_csr_mat_mul_ones_minus_mat (:216-258),
//            three passes there, ONE fused pass here;
//   dense -> np.sum(y_true * y_pred, axis=0, dtype=float64) etc.
//            (confusion_matrix.py:160-166, :187-202).
// Column accumulation: float64 global atomics (global_atomic_add_f64; the memory side retires ~23.5 G scattered atomic
// adds per second whatever their scope, tools/atomic_scope_probe.hip).  Large well-formed inputs send atomics for the
// PREDICTED entries only (xc_confusion_csr_pred_side: fn starts from y_true's cached column sums).  A counting-sort form
// (contributions bucketed by label and summed in LDS) was built in round 2, lost to both (2.7 vs 2.4 / 0.8 ms at 1 M x
// 500 K, profiles/r02_confusion_timing.txt) and was removed in round 3.
#include "xc_common.h"
#include "xc_host.h"

namespace xc {

enum { CF_TP = 0, CF_FP = 1, CF_FN = 2 };

// sink of the row logic below: straight to the result with global atomics
struct AtomicSink {
    double *tp, *fp, *fn;
    __device__ __forceinline__ void emit(int stat, int col, double v) const {
        atomic_add_f64((stat == CF_TP ? tp : stat == CF_FP ? fp : fn) + col, v);
    }
};

// first position in [lo, hi) whose column is >= col
__device__ __forceinline__ int lower_bound_col(const int32_t *indices, int lo, int hi, int col) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (indices[mid] < col) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// One wavefront per row; lanes stride over the row's y_pred entries and then
// over its y_true entries, each locating its partner in the other (sorted) row
// by binary search -- the rows are a few hundred bytes and stay in L1/L2.
//
// The reference's merges + `result[idx] += data` scatter give these rules when
// a column id repeats in a sorted y_pred row:
//   tp: only the first of the repeats can match y_true (:128-139);
//   fp: every repeat emits a value but the scatter keeps the last one, and only
//       the first repeat is "matched" (:199-212, :254);
//   fn: a y_true entry pairs with the first repeat (:199-212).
// Rows that are NOT sorted (the reference's own top-k leaves "[3, 0, 0]" in a row
// with fewer than k entries: stored ids, then column-0 padding,
// numba_csr_functions.py:599-601) go through `merge_row_sequential`, which replays
// the three two-pointer merges literally in one lane; it covers rows of up to
// XC_SEQ_CAP entries each -- longer unsorted rows are outside the contract
// (:121 "requires sorted indices").
#define XC_SEQ_CAP 64

template <typename Sink>
__device__ void scatter_last_wins(Sink &sink, int stat, const int *eidx, const double *eval, int c) {
    for (int e = 0; e < c; ++e) {
        bool later = false;
        for (int f = e + 1; f < c; ++f) later = later || (eidx[f] == eidx[e]);
        if (!later) sink.emit(stat, eidx[e], eval[e]);
    }
}

template <typename T, typename Sink>
__device__ void merge_row_sequential(const int32_t *ti, const T *td, int tc, const int32_t *pi, const T *pd,
                                     int pc, Sink &sink, int *eidx, double *eval) {
    // tp: numba_csr_vec_mul_vec(pred, true)  (:124-140)
    int i = 0, j = 0, c = 0;
    while (i < pc && j < tc) {
        if (pi[i] < ti[j]) ++i;
        else if (pi[i] == ti[j]) { eidx[c] = pi[i]; eval[c] = (double)(T)(pd[i] * td[j]); ++c; ++i; ++j; }
        else ++j;
    }
    scatter_last_wins(sink, CF_TP, eidx, eval, c);
    // fp: numba_csr_vec_mul_ones_minus_vec(pred, true)  (:195-213)
    i = j = c = 0;
    while (i < pc) {
        if (j >= tc || pi[i] < ti[j]) { eidx[c] = pi[i]; eval[c] = (double)pd[i]; ++c; ++i; }
        else if (pi[i] == ti[j]) { eidx[c] = pi[i]; eval[c] = (double)(T)((double)pd[i] * (1.0 - (double)td[j])); ++c; ++i; ++j; }
        else ++j;
    }
    scatter_last_wins(sink, CF_FP, eidx, eval, c);
    // fn: numba_csr_vec_mul_ones_minus_vec(true, pred)
    i = j = c = 0;
    while (i < tc) {
        if (j >= pc || ti[i] < pi[j]) { eidx[c] = ti[i]; eval[c] = (double)td[i]; ++c; ++i; }
        else if (ti[i] == pi[j]) { eidx[c] = ti[i]; eval[c] = (double)(T)((double)td[i] * (1.0 - (double)pd[j])); ++c; ++i; ++j; }
        else ++j;
    }
    scatter_last_wins(sink, CF_FN, eidx, eval, c);
}

// the contributions of one row, by the 64 lanes of a wavefront (eidx / eval: XC_SEQ_CAP slots of LDS of this wavefront)
template <typename T, typename Sink>
__device__ __forceinline__ void confusion_row(const int32_t *t_indptr, const int32_t *t_indices, const T *t_data,
                                              const int32_t *p_indptr, const int32_t *p_indices, const T *p_data,
                                              int64_t row, int lane, Sink &sink, int *eidx, double *eval) {
    const int ts = t_indptr[row], te = t_indptr[row + 1];
    const int ps = p_indptr[row], pe = p_indptr[row + 1];
    // sortedness (non-decreasing) of both rows
    bool bad = false;
    for (int q = ps + 1 + lane; q < pe; q += XC_WAVE) bad = bad || (p_indices[q] < p_indices[q - 1]);
    for (int t = ts + 1 + lane; t < te; t += XC_WAVE) bad = bad || (t_indices[t] < t_indices[t - 1]);
    if (__ballot(bad) != 0ull && (pe - ps) <= XC_SEQ_CAP && (te - ts) <= XC_SEQ_CAP) {
        if (lane == 0)
            merge_row_sequential<T>(t_indices + ts, t_data + ts, te - ts, p_indices + ps, p_data + ps, pe - ps, sink, eidx, eval);
        return;
    }
    for (int q = ps + lane; q < pe; q += XC_WAVE) {
        const int col = p_indices[q];
        const T pv = p_data[q];
        const bool first = (q == ps) || (p_indices[q - 1] != col);
        const bool last = (q == pe - 1) || (p_indices[q + 1] != col);
        const int t = lower_bound_col(t_indices, ts, te, col);
        const bool found = first && t < te && t_indices[t] == col;
        if (found) {
            const T tv = t_data[t];
            sink.emit(CF_TP, col, (double)(T)(pv * tv)); // :133
            if (last) // a * (1.0 - b): float64 inside numba, stored as T (:197, :206)
                sink.emit(CF_FP, col, (double)(T)((double)pv * (1.0 - (double)tv)));
        } else if (last) {
            sink.emit(CF_FP, col, (double)pv); // :200-203
        }
    }
    for (int t = ts + lane; t < te; t += XC_WAVE) {
        const int col = t_indices[t];
        const T tv = t_data[t];
        const int q = lower_bound_col(p_indices, ps, pe, col);
        T v = tv;
        if (q < pe && p_indices[q] == col) v = (T)((double)tv * (1.0 - (double)p_data[q]));
        if (v != (T)0) sink.emit(CF_FN, col, (double)v);
    }
}

template <typename T>
__global__ __launch_bounds__(XC_BLOCK) void confusion_csr_kernel(
    int64_t n, const int32_t *t_indptr, const int32_t *t_indices, const T *t_data,
    const int32_t *p_indptr, const int32_t *p_indices, const T *p_data, double *tp, double *fp,
    double *fn, int n_waves) {
    __shared__ int s_eidx[XC_BLOCK / XC_WAVE][XC_SEQ_CAP];
    __shared__ double s_eval[XC_BLOCK / XC_WAVE][XC_SEQ_CAP];
    const int lane = lane_id();
    const int wib = threadIdx.x >> 6;
    const int wave = blockIdx.x * (XC_BLOCK / XC_WAVE) + wib;
    if (wave >= n_waves) return;
    AtomicSink sink{tp, fp, fn};
    for (int64_t row = wave; row < n; row += n_waves)
        confusion_row<T>(t_indptr, t_indices, t_data, p_indptr, p_indices, p_data, row, lane, sink, s_eidx[wib], s_eval[wib]);
}

// ---- prediction-side form ---------------------------------------------------------------
// fn[j] = sum_i t_ij (1 - p_ij) runs over the stored entries of y_true -- 50 M atomics at 1 M x 500 K, of which
// only the entries that MEET a predicted one (at most nnz(y_pred)) differ from the plain column sum of y_true:
//   fn[j] = colsum_t[j] - sum over matched (i, j) of [ t - (T)(t (1 - p)) ]
// (exact differences of two T values in float64).  With colsum_t known -- it depends on y_true alone, the host
// caches it per matrix -- a thread per row walks the row's PREDICTED entries only, finds each in the row of y_true by
// binary search and issues tp / fp / the fn correction: nnz(y_pred) + 2 * matches atomics instead of
// nnz(y_pred) + nnz(y_true).  Requires rows of both matrices with strictly ascending column ids (what the reference
// requires of a csr_matrix, numba_csr_functions.py:121); a row of y_pred that breaks this -- the top-k padding of a
// short row -- raises `flag` and the host falls back to the general kernel.
template <typename T>
__global__ __launch_bounds__(XC_BLOCK) void confusion_pred_side_kernel(int64_t n, const int32_t *t_indptr,
                                                                       const int32_t *t_indices, const T *t_data,
                                                                       const int32_t *p_indptr, const int32_t *p_indices,
                                                                       const T *p_data, double *tp, double *fp, double *fn,
                                                                       int *flag) {
    for (int64_t row = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x; row < n; row += (int64_t)gridDim.x * XC_BLOCK) {
        const int ts = t_indptr[row], te = t_indptr[row + 1];
        const int ps = p_indptr[row], pe = p_indptr[row + 1];
        int prev = -1;
        for (int q = ps; q < pe; ++q) {
            const int col = p_indices[q];
            if (col <= prev) { // not strictly ascending: the general kernel's repeat / merge rules apply
                atomicOr(flag, 1);
                break;
            }
            prev = col;
            const T pv = p_data[q];
            const int t = lower_bound_col(t_indices, ts, te, col);
            if (t < te && t_indices[t] == col) {
                const T tv = t_data[t];
                // (zeros are not sent: a 0/1 y_true makes every matched fp term one, an explicit zero in it every tp term)
                const double a = (double)(T)(pv * tv);                                        // :133
                const double b = (double)(T)((double)pv * (1.0 - (double)tv));                // :197, :206
                const double corr = (double)tv - (double)(T)((double)tv * (1.0 - (double)pv)); // what this entry no longer adds
                if (a != 0.0) atomic_add_f64(tp + col, a);
                if (b != 0.0) atomic_add_f64(fp + col, b);
                if (corr != 0.0) atomic_add_f64(fn + col, -corr);
            } else {
                atomic_add_f64(fp + col, (double)pv); // :200-203
            }
        }
    }
}

// The matching half of the prediction-side form for a 0/1 prediction (every stored y_pred value is 1, float32): val[q] <- the
// y_true value at predicted entry q's label (0 where the row does not store it).  Then tp = sum val, fp = sum (1 - val)
// (float32 subtraction: (T)(p (1 - t)) with p = 1, :197-206) per label -- xc_scatter_sum_f32(pair = 1) on (p_indices, val):
// a counting sort and LDS sums instead of 3 global atomics per predicted entry -- and fn = colsum(y_true) - tp (a matched
// entry's (T)(t (1 - p)) is 0, an unmatched one's is t).  flag |= 1: a y_pred row is not strictly ascending; |= 2: a stored
// y_pred value is not 1 -- the caller takes another form then.
// A 16-lane group per row: the row of y_true streams in coalesced, 4 entries per lane and chunk of 64, and every predicted
// label of the row is compared against the lanes' entries (a thread per row with a binary search per predicted entry: 0.6 ms
// at 1 M x 50 -- every probe its own cache line request).
__global__ __launch_bounds__(XC_BLOCK) void confusion_match_kernel(int64_t n, const int32_t *t_indptr, const int32_t *t_indices,
                                                                   const float *t_data, const int32_t *p_indptr,
                                                                   const int32_t *p_indices, const float *p_data, float *val, int *flag) {
    const int l16 = threadIdx.x & 15, grp = (threadIdx.x & 63) >> 4;
    const int64_t groups = (int64_t)gridDim.x * (XC_BLOCK / 16);
    int bad = 0;
    for (int64_t row = (int64_t)blockIdx.x * (XC_BLOCK / 16) + (threadIdx.x >> 4); row < n; row += groups) {
        const int ts = t_indptr[row], te = t_indptr[row + 1];
        const int ps = p_indptr[row], pe = p_indptr[row + 1];
        int last = -1; // the last predicted label of the previous chunk
        for (int pb = ps; pb < pe; pb += 16) {
            const int q = pb + l16;
            const int col = q < pe ? p_indices[q] : 0x7fffffff;
            if (q < pe) bad |= p_data[q] != 1.0f ? 2 : 0;
            int prev = __shfl_up(col, 1, 16);
            if (l16 == 0) prev = last;
            if (q < pe) bad |= col <= prev ? 1 : 0;
            last = __shfl(col, 15, 16);
            const int npred = pe - pb < 16 ? pe - pb : 16;
            float tv = 0.0f;
            for (int tb = ts; tb < te; tb += 64) {
                const int e = tb + 4 * l16;
                const int e0 = e < te ? t_indices[e] : -2, e1 = e + 1 < te ? t_indices[e + 1] : -2;
                const int e2 = e + 2 < te ? t_indices[e + 2] : -2, e3 = e + 3 < te ? t_indices[e + 3] : -2;
                for (int j = 0; j < npred; ++j) {
                    const int c = __shfl(col, j, 16);
                    const int hit = e0 == c ? 0 : (e1 == c ? 1 : (e2 == c ? 2 : (e3 == c ? 3 : -1)));
                    const unsigned gb = (unsigned)(__ballot(hit >= 0) >> (16 * grp)) & 0xFFFFu; // this row's lanes
                    if (gb != 0u) { // (one lane at most: y_true's rows are strictly ascending)
                        float v = hit >= 0 ? t_data[e + hit] : 0.0f;
                        v = __shfl(v, __builtin_ctz(gb), 16);
                        if (l16 == j) tv = v;
                    }
                }
            }
            if (q < pe) val[q] = tv;
        }
    }
    if (bad) atomicOr(flag, bad);
}

// strictly ascending column ids in every row?  A 16-lane group per row; flag |= 1 otherwise.
__global__ __launch_bounds__(XC_BLOCK) void csr_rows_ascending_kernel(int64_t n, const int32_t *indptr, const int32_t *indices,
                                                                      int *flag) {
    const int l16 = threadIdx.x & 15;
    const int64_t groups = (int64_t)gridDim.x * (XC_BLOCK / 16);
    bool bad = false;
    for (int64_t row = (int64_t)blockIdx.x * (XC_BLOCK / 16) + (threadIdx.x >> 4); row < n; row += groups) {
        const int s = indptr[row], e = indptr[row + 1];
        for (int q = s + 1 + l16; q < e; q += 16) bad = bad || (indices[q] <= indices[q - 1]);
    }
    if (bad) atomicOr(flag, 1);
}

// Dense: a workgroup owns a strip of XC_BLOCK columns x ROWS_PER_BLOCK rows,
// each thread sums its column over the strip's rows in row order (coalesced
// row-major reads) and pushes one atomic per statistic.
#define XC_CONF_ROWS_PER_BLOCK 256
template <typename T>
__global__ __launch_bounds__(XC_BLOCK) void confusion_dense_kernel(int64_t n, int64_t m, const T *y_true,
                                                                   const T *y_pred, double *tp, double *fp,
                                                                   double *fn) {
    const int64_t j = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.y * XC_CONF_ROWS_PER_BLOCK;
    const int64_t i1 = (i0 + XC_CONF_ROWS_PER_BLOCK < n) ? i0 + XC_CONF_ROWS_PER_BLOCK : n;
    if (j >= m) return;
    double stp = 0.0, sfp = 0.0, sfn = 0.0;
    const T one = (T)1;
    for (int64_t i = i0; i < i1; ++i) {
        const T t = y_true[i * m + j];
        const T p = y_pred[i * m + j];
        stp += (double)(T)(t * p);          // confusion_matrix.py:166
        sfp += (double)(T)((one - t) * p);  // :193
        sfn += (double)(T)(t * (one - p));  // :202
    }
    if (stp != 0.0) atomic_add_f64(tp + j, stp);
    if (sfp != 0.0) atomic_add_f64(fp + j, sfp);
    if (sfn != 0.0) atomic_add_f64(fn + j, sfn);
}

// ---- binary labels against a fixed-stride 0/1 prediction: counts only -----------------
// With y_true in {0, 1}, predictions of exactly k distinct stored-or-not labels per row and no padding,
// fp = (#rows predicting j) - tp and fn = (#rows labelled j) - tp: only tp needs the row-wise match, and
// the label counts of y_true never change between calls.  One thread per predicted entry: count its
// label, binary-search it in the row's sorted true labels.  Less than half the atomics of the general
// kernel (k + matches per row instead of k + |true row| - matches); used by the Frank-Wolfe iteration.
__global__ __launch_bounds__(XC_BLOCK) void confusion_counts_kernel(int64_t n_k, int k, const int32_t *t_indptr,
                                                                   const int32_t *t_indices, const int32_t *p_indices,
                                                                   double *tp, double *cnt) {
    const int64_t t = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x;
    if (t >= n_k) return;
    const int64_t row = t / k;
    const int label = p_indices[t];
    atomic_add_f64(cnt + label, 1.0);
    int lo = t_indptr[row], hi = t_indptr[row + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int v = t_indices[mid];
        if (v < label) lo = mid + 1;
        else hi = mid;
    }
    if (lo < t_indptr[row + 1] && t_indices[lo] == label) atomic_add_f64(tp + label, 1.0);
}

} // namespace xc

extern "C" {

int xc_confusion_csr(int64_t n, int64_t m, const int32_t *t_indptr, const int32_t *t_indices,
                     const void *t_data, const int32_t *p_indptr, const int32_t *p_indices,
                     const void *p_data, int dtype, double *tp, double *fp, double *fn, void *stream) {
    if (n < 0 || m < 0 || !t_indptr || !p_indptr || !tp || !fp || !fn)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_confusion_csr: NULL pointer or negative size");
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_confusion_csr: unknown dtype %d", dtype);
    if (n == 0) return XC_OK;
    const int n_waves = xc::default_row_waves(n);
    const int blocks = (n_waves + 3) / 4;
    hipStream_t st = xc::as_stream(stream);
    if (dtype == XC_F32)
        hipLaunchKernelGGL((xc::confusion_csr_kernel<float>), dim3(blocks), dim3(XC_BLOCK), 0, st, n, t_indptr, t_indices,
                           static_cast<const float *>(t_data), p_indptr, p_indices, static_cast<const float *>(p_data),
                           tp, fp, fn, n_waves);
    else
        hipLaunchKernelGGL((xc::confusion_csr_kernel<double>), dim3(blocks), dim3(XC_BLOCK), 0, st, n, t_indptr, t_indices,
                           static_cast<const double *>(t_data), p_indptr, p_indices, static_cast<const double *>(p_data),
                           tp, fp, fn, n_waves);
    XC_CHECK_LAUNCH("confusion_csr_kernel");
    return XC_OK;
}

int xc_csr_rows_ascending(int64_t n, const int32_t *indptr, const int32_t *indices, int32_t *flag, void *stream) {
    if (n < 0 || !indptr || !flag) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_csr_rows_ascending: bad argument");
    if (n == 0) return XC_OK;
    int64_t blocks = (n + (XC_BLOCK / 16) - 1) / (XC_BLOCK / 16);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(xc::csr_rows_ascending_kernel, dim3((unsigned)blocks), dim3(XC_BLOCK), 0, xc::as_stream(stream), n, indptr,
                       indices, flag);
    XC_CHECK_LAUNCH("csr_rows_ascending_kernel");
    return XC_OK;
}

int xc_confusion_csr_pred_side(int64_t n, int64_t m, const int32_t *t_indptr, const int32_t *t_indices, const void *t_data,
                               const int32_t *p_indptr, const int32_t *p_indices, const void *p_data, int dtype, double *tp,
                               double *fp, double *fn, int32_t *flag, void *stream) {
    if (n < 0 || m < 1 || !t_indptr || !p_indptr || !tp || !fp || !fn || !flag)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_confusion_csr_pred_side: NULL pointer or negative size");
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_confusion_csr_pred_side: unknown dtype %d", dtype);
    if (n == 0) return XC_OK;
    int64_t blocks = (n + XC_BLOCK - 1) / XC_BLOCK;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipStream_t st = xc::as_stream(stream);
    if (dtype == XC_F32)
        hipLaunchKernelGGL((xc::confusion_pred_side_kernel<float>), dim3((unsigned)blocks), dim3(XC_BLOCK), 0, st, n, t_indptr,
                           t_indices, static_cast<const float *>(t_data), p_indptr, p_indices, static_cast<const float *>(p_data),
                           tp, fp, fn, flag);
    else
        hipLaunchKernelGGL((xc::confusion_pred_side_kernel<double>), dim3((unsigned)blocks), dim3(XC_BLOCK), 0, st, n, t_indptr,
                           t_indices, static_cast<const double *>(t_data), p_indptr, p_indices, static_cast<const double *>(p_data),
                           tp, fp, fn, flag);
    XC_CHECK_LAUNCH("confusion_pred_side_kernel");
    return XC_OK;
}

int xc_confusion_csr_match(int64_t n, const int32_t *t_indptr, const int32_t *t_indices, const float *t_data,
                           const int32_t *p_indptr, const int32_t *p_indices, const float *p_data, float *val, int32_t *flag,
                           void *stream) {
    if (n < 0 || !t_indptr || !p_indptr || !flag) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_confusion_csr_match: NULL pointer or negative size");
    if (n == 0) return XC_OK;
    int64_t blocks = (n + XC_BLOCK / 16 - 1) / (XC_BLOCK / 16);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(xc::confusion_match_kernel, dim3((unsigned)blocks), dim3(XC_BLOCK), 0, xc::as_stream(stream), n, t_indptr, t_indices,
                       t_data, p_indptr, p_indices, p_data, val, flag);
    XC_CHECK_LAUNCH("confusion_match_kernel");
    return XC_OK;
}

int xc_confusion_counts_csr(int64_t n, int k, const int32_t *t_indptr, const int32_t *t_indices,
                            const int32_t *p_indices, double *tp, double *cnt, void *stream) {
    if (n < 0 || k < 1 || !t_indptr || !tp || !cnt || (n > 0 && !p_indices))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_confusion_counts_csr: NULL pointer or bad size");
    if (n == 0) return XC_OK;
    const int64_t n_k = n * (int64_t)k;
    const int64_t blocks = (n_k + XC_BLOCK - 1) / XC_BLOCK;
    hipLaunchKernelGGL(xc::confusion_counts_kernel, dim3((unsigned)blocks), dim3(XC_BLOCK), 0, xc::as_stream(stream), n_k, k,
                       t_indptr, t_indices, p_indices, tp, cnt);
    XC_CHECK_LAUNCH("confusion_counts_kernel");
    return XC_OK;
}

int xc_confusion_dense(int64_t n, int64_t m, const void *y_true, const void *y_pred, int dtype, double *tp,
                       double *fp, double *fn, void *stream) {
    if (n < 0 || m < 0 || !tp || !fp || !fn || (n * m > 0 && (!y_true || !y_pred)))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_confusion_dense: NULL pointer or negative size");
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_confusion_dense: unknown dtype %d", dtype);
    if (n == 0 || m == 0) return XC_OK;
    dim3 grid((unsigned)((m + XC_BLOCK - 1) / XC_BLOCK), (unsigned)((n + XC_CONF_ROWS_PER_BLOCK - 1) / XC_CONF_ROWS_PER_BLOCK));
    hipStream_t st = xc::as_stream(stream);
    if (dtype == XC_F32)
        hipLaunchKernelGGL((xc::confusion_dense_kernel<float>), grid, dim3(XC_BLOCK), 0, st, n, m,
                           static_cast<const float *>(y_true), static_cast<const float *>(y_pred), tp, fp, fn);
    else
        hipLaunchKernelGGL((xc::confusion_dense_kernel<double>), grid, dim3(XC_BLOCK), 0, st, n, m,
                           static_cast<const double *>(y_true), static_cast<const double *>(y_pred), tp, fp, fn);
    XC_CHECK_LAUNCH("confusion_dense_kernel");
    return XC_OK;
}

} // extern "C"

// xc_coverage.hip -- block coordinate ascent for coverage:
// predict_optimizing_coverage_using_bc, xcolumns/block_coordinate.py:600-701, with the CSR step
// _bc_for_coverage_step_csr (:539-582).
//
// The statistic is one float64 per label, Ef_j = prod_i (1 - pred_ij * eta_ij) -- the probability that
// label j is covered by no row -- updated multiplicatively.  One wavefront per row, CH x 64 candidates
// in registers, as in the other row kernels:
//   gather Ef of the row's candidates (sc1: other waves' updates are seen)
//   divide the row's own factor (1 - eta) out of the labels it currently predicts        (:561-563)
//   gain = Ef * eta  (mixed with precision@k: alpha * gain + (1 - alpha) * eta / k)      (:566-568)
//   keep the k largest gains (ties: lower column)                                         (:569-575)
//   multiply (1 - eta) into the labels now predicted                                      (:580-582)
// n_waves = 1 is the reference's sequence bit for bit: the divided values are stored, then the
// multiplied ones, unchanged labels included (x / f * f is not always x).  With more wavefronts the
// labels that leave or enter the prediction are updated with a compare-and-swap multiply and the
// unchanged ones are left alone; rows in flight miss each other's update as in the BCA sweep.
#include "xc_common.h"
#include "xc_host.h"

namespace xc {

template <typename T>
struct CovParams {
    int64_t n_order;
    const int32_t *order;
    const int32_t *indptr;
    const int32_t *indices;
    const T *data;
    int32_t *pred_indices;
    T *pred_eta;
    uint8_t *sel;
    int k;
    double *ef;
    double alpha;
    int greedy;
    int n_waves;
    unsigned long long *changed;
};

// Ef[j] <- Ef[j] * f (or / f), atomically: a 64-bit compare-and-swap loop
template <bool DIVIDE>
__device__ __forceinline__ void atomic_scale(double *p, double f) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
    unsigned long long old = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true) {
        const double cur = __longlong_as_double((long long)old);
        const double nxt = DIVIDE ? cur / f : cur * f;
        unsigned long long expected = old;
        if (__hip_atomic_compare_exchange_strong(q, &expected, (unsigned long long)__double_as_longlong(nxt),
                                                 __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            break;
        old = expected;
    }
}

template <typename T, int CH, bool EXACT>
__global__ __launch_bounds__(XC_BLOCK) void coverage_sweep_csr_kernel(CovParams<T> P) {
    const int lane = lane_id();
    const int wave = blockIdx.x * (XC_BLOCK / XC_WAVE) + (threadIdx.x >> 6);
    if (wave >= P.n_waves) return;
    const int k = P.k;
    const bool greedy = P.greedy != 0;
    const T one = (T)1;
    unsigned long long n_changed = 0;
    for (int64_t pos = wave; pos < P.n_order; pos += P.n_waves) {
        const int64_t row = P.order ? (int64_t)P.order[pos] : pos;
        const int s = P.indptr[row], r = P.indptr[row + 1] - s;
        int idx[CH];
        T eta[CH];
        bool in_old[CH], in_new[CH], valid[CH];
        double e[CH];
        unsigned long long key[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int p = lane + XC_WAVE * c;
            valid[c] = p < r;
            const int pc = valid[c] ? p : (r > 0 ? r - 1 : 0);
            idx[c] = r > 0 ? P.indices[s + pc] : 0;
            eta[c] = r > 0 ? P.data[s + pc] : (T)0;
            in_old[c] = valid[c] && !greedy && P.sel[s + pc] != 0;
            in_new[c] = false;
            e[c] = load_coherent(P.ef + idx[c]);
        }
        const double a = P.alpha;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (in_old[c]) e[c] = e[c] / (double)(one - eta[c]); // :561-563, (1 - eta) in eta's dtype
            double g = e[c] * (double)eta[c];                    // :566
            if (a < 1.0) // :567-568: the precision@k part is formed in eta's dtype (numpy promotion)
                g = a * g + (double)(((T)(1.0 - a) * eta[c]) / (T)k);
            key[c] = valid[c] ? sortable_key(nan_to_neg_inf(g)) : 0ull;
        }
        if (r <= k) { // :572-575: a row of at most k entries keeps them all
#pragma unroll
            for (int c = 0; c < CH; ++c) in_new[c] = valid[c];
        } else {
            for (int round = 0; round < k; ++round) {
                unsigned long long lmax = 0ull;
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    if (!in_new[c] && key[c] > lmax) lmax = key[c];
                const unsigned long long M = wave_umax64(lmax);
                bool found = false;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const unsigned long long mask = __ballot(!in_new[c] && key[c] == M && valid[c]);
                    if (!found && mask != 0ull) {
                        if (lane == __ffsll((long long)mask) - 1) in_new[c] = true;
                        found = true;
                    }
                }
            }
        }
        bool any_change = false;
#pragma unroll
        for (int c = 0; c < CH; ++c) any_change = any_change || (in_new[c] != in_old[c]);
        const bool row_changed = __ballot(any_change) != 0ull;

        // statistics
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (!valid[c]) continue;
            const double f = (double)(one - eta[c]);
            if (EXACT) { // the reference's two in-place passes, unchanged labels included
                if (in_old[c] || in_new[c]) P.ef[idx[c]] = in_new[c] ? e[c] * f : e[c];
            } else if (in_new[c] != in_old[c]) {
                if (in_new[c]) atomic_scale<false>(P.ef + idx[c], f);
                else atomic_scale<true>(P.ef + idx[c], f);
            }
        }
        // prediction (ascending columns), per-entry flags
        if (row_changed || greedy) {
            int base = 0;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const unsigned long long mask = __ballot(in_new[c]);
                if (in_new[c]) {
                    const int slot = base + __popcll(mask & lanemask_lt());
                    P.pred_indices[row * k + slot] = idx[c];
                    P.pred_eta[row * k + slot] = eta[c];
                }
                base += __popcll(mask);
                if (valid[c]) P.sel[s + lane + XC_WAVE * c] = in_new[c] ? 1 : 0;
            }
            if (row_changed) ++n_changed;
        }
        // sequential mode: this wave's stores must have landed before it gathers for its next row
        if (EXACT) __builtin_amdgcn_s_waitcnt(0);
    }
    if (lane == 0 && n_changed && P.changed) atomicAdd(P.changed, n_changed);
}

// Ef from scratch (numba_csr_functions.py:324-382): ef preset to ones; every (row, slot) multiplies
// (1 - eta) into its label.  eta = 0 marks a predicted label the row does not store: no factor.
template <typename T>
__global__ __launch_bounds__(XC_BLOCK) void coverage_product_kernel(int64_t n_k, const int32_t *pred_indices,
                                                                   const T *pred_eta, double *ef) {
    const int64_t t = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x;
    if (t >= n_k) return;
    const T eta = pred_eta[t];
    if (eta != (T)0) atomic_scale<false>(ef + pred_indices[t], (double)((T)1 - eta));
}

template <typename T, bool EXACT>
static void launch_cov(const CovParams<T> &P, int ch, hipStream_t st) {
    const int blocks = (P.n_waves + 3) / 4;
    if (ch <= 1) hipLaunchKernelGGL((coverage_sweep_csr_kernel<T, 1, EXACT>), dim3(blocks), dim3(XC_BLOCK), 0, st, P);
    else if (ch <= 4) hipLaunchKernelGGL((coverage_sweep_csr_kernel<T, 4, EXACT>), dim3(blocks), dim3(XC_BLOCK), 0, st, P);
    else hipLaunchKernelGGL((coverage_sweep_csr_kernel<T, 16, EXACT>), dim3(blocks), dim3(XC_BLOCK), 0, st, P);
}

} // namespace xc

extern "C" {

int xc_coverage_sweep_csr(int64_t n_order, const int32_t *order, const int32_t *indptr, const int32_t *indices,
                          const void *data, int dtype, int max_row_nnz, int32_t *pred_indices, void *pred_eta,
                          uint8_t *sel, int k, double *ef, double alpha, int greedy, int n_waves, int64_t *changed,
                          void *stream) {
    if (n_order < 0 || !indptr || !pred_indices || !pred_eta || !sel || !ef)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_coverage_sweep_csr: NULL pointer or bad size");
    if (k < 1 || k > XC_MAX_K) return xc::fail_arg(XC_ERR_K_RANGE, "xc_coverage_sweep_csr: k=%d outside 1..%d", k, XC_MAX_K);
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_coverage_sweep_csr: unknown dtype %d", dtype);
    const int ch = xc::chunks_for(max_row_nnz);
    if (ch == 0)
        return xc::fail_arg(XC_ERR_ROW_TOO_LONG, "xc_coverage_sweep_csr: a row holds %d entries, limit %d", max_row_nnz, XC_MAX_ROW_NNZ);
    if (n_waves < 1) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_coverage_sweep_csr: n_waves must be >= 1");
    if (n_order == 0) return XC_OK;
    if (n_waves > n_order) n_waves = (int)n_order;
    hipStream_t st = xc::as_stream(stream);
    unsigned long long *chg = reinterpret_cast<unsigned long long *>(changed);
    if (dtype == XC_F32) {
        xc::CovParams<float> P{n_order, order, indptr, indices, static_cast<const float *>(data), pred_indices,
                               static_cast<float *>(pred_eta), sel, k, ef, alpha, greedy, n_waves, chg};
        if (n_waves == 1) xc::launch_cov<float, true>(P, ch, st);
        else xc::launch_cov<float, false>(P, ch, st);
    } else {
        xc::CovParams<double> P{n_order, order, indptr, indices, static_cast<const double *>(data), pred_indices,
                                static_cast<double *>(pred_eta), sel, k, ef, alpha, greedy, n_waves, chg};
        if (n_waves == 1) xc::launch_cov<double, true>(P, ch, st);
        else xc::launch_cov<double, false>(P, ch, st);
    }
    XC_CHECK_LAUNCH("coverage_sweep_csr_kernel");
    return XC_OK;
}

int xc_coverage_product(int64_t n_k, const int32_t *pred_indices, const void *pred_eta, int dtype, double *ef,
                        void *stream) {
    if (n_k < 0 || (n_k > 0 && (!pred_indices || !pred_eta || !ef)))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_coverage_product: NULL pointer or bad size");
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_coverage_product: unknown dtype %d", dtype);
    if (n_k == 0) return XC_OK;
    const int blocks = (int)((n_k + XC_BLOCK - 1) / XC_BLOCK);
    hipStream_t st = xc::as_stream(stream);
    if (dtype == XC_F32)
        hipLaunchKernelGGL(xc::coverage_product_kernel<float>, dim3(blocks), dim3(XC_BLOCK), 0, st, n_k, pred_indices,
                           static_cast<const float *>(pred_eta), ef);
    else
        hipLaunchKernelGGL(xc::coverage_product_kernel<double>, dim3(blocks), dim3(XC_BLOCK), 0, st, n_k, pred_indices,
                           static_cast<const double *>(pred_eta), ef);
    XC_CHECK_LAUNCH("coverage_product_kernel");
    return XC_OK;
}

} // extern "C"

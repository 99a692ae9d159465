// xc_dense.hip -- dense-matrix branches of the path.
//
//   xc_topk_dense      <- _predict_weighted_per_instance_dense
//                         (/root/reference/xcolumns/weighted_prediction.py:25-60)
//   xc_bca_sweep_dense <- the row loop block_coordinate.py:448-463 with
//                         _bc_with_0approx_step_dense (:132-209) as its body
//
// Dense rows touch all m labels, so there is no sparsity to exploit: top-k is
// one workgroup per row streaming the row once (HBM-bound, 2 x sizeof per
// element); the BCA sweep is inherently sequential in the rows (row i+1 reads
// the statistics row i wrote for EVERY label), so one 1024-thread workgroup
// walks the order array and parallelises over labels inside a row, keeping the
// reference's exact operation order (bit-identical running statistics).
#include "xc_common.h"
#include "xc_host.h"

namespace xc {

// ---- dense top-k --------------------------------------------------------------
// Workgroup-wide max of a uint64 (DPP wave max, then 4 values through LDS).
template <int BLOCK>
__device__ __forceinline__ unsigned long long block_umax64(unsigned long long v, unsigned long long *lds) {
    v = wave_umax64(v);
    if (lane_id() == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long r = 0ull;
#pragma unroll
    for (int w = 0; w < BLOCK / XC_WAVE; ++w) r = lds[w] > r ? lds[w] : r;
    __syncthreads();
    return r;
}

template <int BLOCK>
__device__ __forceinline__ unsigned block_umin32(unsigned v, unsigned *lds) {
    v = wave_umin32(v);
    if (lane_id() == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned r = ~0u;
#pragma unroll
    for (int w = 0; w < BLOCK / XC_WAVE; ++w) r = lds[w] < r ? lds[w] : r;
    __syncthreads();
    return r;
}

// gains: G (promoted dtype), y_pred: P (y_proba's dtype).  One workgroup per row;
// the row (<= a few tens of KB) is streamed once from HBM and re-read from L1/L2 in
// the k selection rounds.  Round t picks the best element strictly worse than round
// t-1's winner in the total order (gain desc, column asc): no selected-set
// bookkeeping.  float32 gains use ONE 64-bit key (sortable gain << 32 | ~column), so
// a round is a single block-wide max; float64 gains take a max round on the sortable
// gain and a min round on the column among its holders.
template <typename G, typename P>
__global__ __launch_bounds__(XC_BLOCK) void topk_dense_kernel(int64_t m, int64_t ld, const G *gains, int k, G th,
                                                              int keep_scores, P *y_pred) {
    __shared__ unsigned long long red64[XC_BLOCK / XC_WAVE];
    __shared__ unsigned red32[XC_BLOCK / XC_WAVE];
    const int64_t row = blockIdx.x;
    const G *g = gains + row * ld;
    P *o = y_pred + row * m;
    if (k == 0) { // weighted_prediction.py:58
        for (int64_t j = threadIdx.x; j < m; j += XC_BLOCK) o[j] = (g[j] >= th) ? (P)1 : (P)0;
        return;
    }
    for (int64_t j = threadIdx.x; j < m; j += XC_BLOCK) o[j] = (P)0; // :35
    const int rounds = (int64_t)k < m ? k : (int)m;
    if (sizeof(G) == 4) {
        unsigned long long prev = ~0ull;
        for (int round = 0; round < rounds; ++round) {
            unsigned long long best = 0ull;
            for (int64_t j = threadIdx.x; j < m; j += XC_BLOCK) {
                const unsigned long long key =
                    ((unsigned long long)sortable_key32(nan_to_neg_inf((float)g[j])) << 32) | (unsigned)(~(unsigned)j);
                if (key < prev && key > best) best = key;
            }
            best = block_umax64<XC_BLOCK>(best, red64);
            const unsigned col = ~(unsigned)best;
            if (threadIdx.x == 0) o[col] = keep_scores ? (P)g[col] : (P)1; // :47-49
            prev = best;
        }
    } else {
        unsigned long long prev_key = ~0ull;
        unsigned prev_col = 0u;
        bool first = true;
        for (int round = 0; round < rounds; ++round) {
            // remaining = strictly after (prev_key, prev_col) in (key desc, col asc)
            unsigned long long best = 0ull;
            for (int64_t j = threadIdx.x; j < m; j += XC_BLOCK) {
                const unsigned long long key = sortable_key(nan_to_neg_inf((double)g[j]));
                const bool rem = first || key < prev_key || (key == prev_key && (unsigned)j > prev_col);
                if (rem && key > best) best = key;
            }
            best = block_umax64<XC_BLOCK>(best, red64);
            unsigned col = ~0u;
            for (int64_t j = threadIdx.x; j < m; j += XC_BLOCK) {
                const unsigned long long key = sortable_key(nan_to_neg_inf((double)g[j]));
                const bool rem = first || key < prev_key || (key == prev_key && (unsigned)j > prev_col);
                if (rem && key == best && (unsigned)j < col) col = (unsigned)j;
            }
            col = block_umin32<XC_BLOCK>(col, red32);
            if (threadIdx.x == 0) o[col] = keep_scores ? (P)g[col] : (P)1;
            prev_key = best;
            prev_col = col;
            first = false;
        }
    }
}

// float32 gains of at most 32 * XC_BLOCK labels: the row is read ONCE into registers as 64-bit keys (sortable gain << 32 | ~column) and the k rounds run
// on registers; the output row is written once (zeros and the winners together).
template <typename P, int EPT>
__global__ __launch_bounds__(XC_BLOCK) void topk_dense_reg_kernel(int64_t m, int64_t ld, const float *gains, int k,
                                                                  int keep_scores, P *y_pred) {
    __shared__ unsigned long long red64[XC_BLOCK / XC_WAVE];
    const int64_t row = blockIdx.x;
    const float *g = gains + row * ld;
    P *o = y_pred + row * m;
    float val[EPT];
    unsigned long long key[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t j = threadIdx.x + (int64_t)e * XC_BLOCK;
        val[e] = j < m ? __builtin_nontemporal_load(g + j) : 0.0f;
        key[e] = j < m ? (((unsigned long long)sortable_key32(nan_to_neg_inf(val[e])) << 32) | (unsigned)(~(unsigned)j))
                       : 0ull;
    }
    unsigned picked = 0u; // bit e: this thread's e-th label is among the k winners
    const int rounds = (int64_t)k < m ? k : (int)m;
    unsigned long long prev = ~0ull;
    for (int round = 0; round < rounds; ++round) {
        unsigned long long best = 0ull;
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (key[e] < prev && key[e] > best) best = key[e];
        best = block_umax64<XC_BLOCK>(best, red64);
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (key[e] == best && best != 0ull) picked |= 1u << e;
        prev = best;
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t j = threadIdx.x + (int64_t)e * XC_BLOCK;
        if (j < m) o[j] = ((picked >> e) & 1u) ? (keep_scores ? (P)val[e] : (P)1) : (P)0; // weighted_prediction.py:35, :47-49
    }
}

// ---- dense BCA sweep -----------------------------------------------------------
#define XC_DENSE_BLOCK 1024
#define XC_DENSE_MAX_EPT 64 /* labels per thread tracked in the selection mask */

template <typename T>
struct DenseSweepParams {
    int64_t n_order;
    const int32_t *order;
    int64_t m;
    const T *y_proba;
    T *y_pred;
    int k;
    double *tp, *fp, *fn, *tn;
    double *gains; // workspace, m doubles
    xc_metric metric;
    double nn;
    int maximize, greedy, skip_tn;
};

// EPT > 0: every thread keeps the sortable keys of its <= EPT labels in registers
// for the k selection rounds (m <= EPT * 1024); EPT == 0: they are re-read from the
// gains workspace each round (any m up to 65536).
template <typename T, int EPT>
__global__ __launch_bounds__(XC_DENSE_BLOCK) void bca_sweep_dense_kernel(DenseSweepParams<T> P) {
    __shared__ unsigned long long red64[XC_DENSE_BLOCK / XC_WAVE];
    __shared__ unsigned red32[XC_DENSE_BLOCK / XC_WAVE];
    const int64_t m = P.m;
    const double nn = P.nn;
    const T one = (T)1;
    for (int64_t pos = 0; pos < P.n_order; ++pos) {
        const int64_t row = P.order ? (int64_t)P.order[pos] : pos;
        const T *eta = P.y_proba + row * m;
        T *pred = P.y_pred + row * m;

        for (int64_t j = threadIdx.x; j < m; j += XC_DENSE_BLOCK) {
            const T t = eta[j];
            const T p = pred[j];
            const T om = one - t;
            double tp = P.tp[j], fp = P.fp[j], fn = P.fn[j], tn = P.tn[j];
            if (!P.greedy) { // block_coordinate.py:157-163
                tp -= (double)(T)(p * t);
                fp -= (double)(T)(p * om);
                fn -= (double)(T)((one - p) * t);
                if (!P.skip_tn) tn -= (double)(T)((one - p) * om);
            }
            // :166-185
            const double pos_tp = tp + (double)t;
            const double pos_fp = fp + (double)om;
            const double neg_fn = fn + (double)t;
            double neg_tn = tn;
            if (!P.skip_tn) neg_tn = tn + (double)om;
            double g = metric_eval(P.metric, pos_tp / nn, pos_fp / nn, fn / nn, tn / nn) -
                       metric_eval(P.metric, tp / nn, fp / nn, neg_fn / nn, neg_tn / nn);
            if (!P.maximize) g = -g;
            P.tp[j] = tp; P.fp[j] = fp; P.fn[j] = fn; P.tn[j] = tn;
            P.gains[j] = g; // larger is better here; the reference negates and takes the smallest (:187-198)
            pred[j] = (T)0; // :191
        }
        __syncthreads();
        unsigned long long keys[EPT > 0 ? EPT : 1];
        if (EPT > 0) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const int64_t j = threadIdx.x + (int64_t)e * XC_DENSE_BLOCK;
                keys[e] = j < m ? sortable_key(nan_to_neg_inf(P.gains[j])) : 0ull;
            }
        }

        unsigned long long mine = 0ull; // selected labels among this thread's (bit e <-> j = tid + e * BLOCK)
        if (P.k > 0) {
            // k rounds; each: block max of the sortable gain among the labels strictly
            // after the previous winner in (gain desc, column asc), then block min of
            // the column among its holders -- DPP wave reductions, 16 values via LDS
            unsigned long long prev_key = ~0ull;
            unsigned prev_col = 0u;
            bool first = true;
            const int rounds = (int64_t)P.k < m ? P.k : (int)m;
            for (int round = 0; round < rounds; ++round) {
                unsigned long long best = 0ull;
                unsigned col = ~0u;
                if (EPT > 0) {
#pragma unroll
                    for (int e = 0; e < EPT; ++e) {
                        const unsigned j = threadIdx.x + e * XC_DENSE_BLOCK;
                        const bool rem = first || keys[e] < prev_key || (keys[e] == prev_key && j > prev_col);
                        if (rem && keys[e] > best) best = keys[e]; // key 0 = past the row's end
                    }
                    best = block_umax64<XC_DENSE_BLOCK>(best, red64);
#pragma unroll
                    for (int e = 0; e < EPT; ++e) {
                        const unsigned j = threadIdx.x + e * XC_DENSE_BLOCK;
                        const bool rem = first || keys[e] < prev_key || (keys[e] == prev_key && j > prev_col);
                        if (rem && keys[e] == best && j < col) col = j;
                    }
                } else {
                    for (int64_t j = threadIdx.x; j < m; j += XC_DENSE_BLOCK) {
                        const unsigned long long key = sortable_key(nan_to_neg_inf(P.gains[j]));
                        const bool rem = first || key < prev_key || (key == prev_key && (unsigned)j > prev_col);
                        if (rem && key > best) best = key;
                    }
                    best = block_umax64<XC_DENSE_BLOCK>(best, red64);
                    for (int64_t j = threadIdx.x; j < m; j += XC_DENSE_BLOCK) {
                        const unsigned long long key = sortable_key(nan_to_neg_inf(P.gains[j]));
                        const bool rem = first || key < prev_key || (key == prev_key && (unsigned)j > prev_col);
                        if (rem && key == best && (unsigned)j < col) col = (unsigned)j;
                    }
                }
                col = block_umin32<XC_DENSE_BLOCK>(col, red32);
                if ((col % XC_DENSE_BLOCK) == threadIdx.x) mine |= 1ull << (col / XC_DENSE_BLOCK);
                prev_key = best;
                prev_col = col;
                first = false;
            }
        }

        int e = 0;
        for (int64_t j = threadIdx.x; j < m; j += XC_DENSE_BLOCK, ++e) {
            bool sel;
            if (P.k > 0) sel = (mine >> e) & 1ull;
            else sel = (-P.gains[j]) <= 0.0; // :199-200 on the negated gains
            const T t = eta[j];
            const T p = sel ? one : (T)0;
            const T om = one - t;
            if (sel) pred[j] = one;
            // :203-209
            P.tp[j] += (double)(T)(p * t);
            P.fp[j] += (double)(T)(p * om);
            P.fn[j] += (double)(T)((one - p) * t);
            if (!P.skip_tn) P.tn[j] += (double)(T)((one - p) * om);
        }
        __syncthreads();
    }
}

// Register-resident variant for m <= EPT * 1024: every thread owns EPT fixed labels
// and keeps their four statistics in registers across the whole sweep, so a row costs
// one coalesced read of (eta, pred), the gains, k x 2 block reductions and the
// prediction stores -- no statistics traffic, no gains workspace.  Same operation
// order per label as the reference (bit-identical running statistics).
template <typename T, int EPT>
__global__ __launch_bounds__(XC_DENSE_BLOCK) void bca_sweep_dense_reg_kernel(DenseSweepParams<T> P) {
    __shared__ unsigned long long red64[XC_DENSE_BLOCK / XC_WAVE];
    __shared__ unsigned red32[XC_DENSE_BLOCK / XC_WAVE];
    const int64_t m = P.m;
    const double nn = P.nn;
    const T one = (T)1;
    double tp[EPT], fp[EPT], fn[EPT], tn[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t j = threadIdx.x + (int64_t)e * XC_DENSE_BLOCK;
        const bool v = j < m;
        tp[e] = v ? P.tp[j] : 0.0;
        fp[e] = v ? P.fp[j] : 0.0;
        fn[e] = v ? P.fn[j] : 0.0;
        tn[e] = v ? P.tn[j] : 0.0;
    }
    auto row_of = [&](int64_t pos) -> int64_t {
        const int64_t q = pos < P.n_order ? pos : P.n_order - 1;
        return P.order ? (int64_t)P.order[q] : q;
    };
    T t_cur[EPT], p_cur[EPT];
    {
        const int64_t row = row_of(0);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int64_t j = threadIdx.x + (int64_t)e * XC_DENSE_BLOCK;
            t_cur[e] = j < m ? P.y_proba[row * m + j] : (T)0;
            p_cur[e] = j < m ? P.y_pred[row * m + j] : (T)0;
        }
    }
    for (int64_t pos = 0; pos < P.n_order; ++pos) {
        const int64_t row = row_of(pos);
        T *pred = P.y_pred + row * m;
        // prefetch the next row (a row is visited once per sweep, so its prediction is
        // not being rewritten meanwhile)
        const int64_t nrow = row_of(pos + 1);
        T t_nxt[EPT], p_nxt[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int64_t j = threadIdx.x + (int64_t)e * XC_DENSE_BLOCK;
            t_nxt[e] = j < m ? P.y_proba[nrow * m + j] : (T)0;
            p_nxt[e] = j < m ? P.y_pred[nrow * m + j] : (T)0;
        }

        unsigned long long keys[EPT];
        bool nonneg[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int64_t j = threadIdx.x + (int64_t)e * XC_DENSE_BLOCK;
            const T t = t_cur[e], p = p_cur[e];
            const T om = one - t;
            if (!P.greedy) { // block_coordinate.py:157-163
                tp[e] -= (double)(T)(p * t);
                fp[e] -= (double)(T)(p * om);
                fn[e] -= (double)(T)((one - p) * t);
                if (!P.skip_tn) tn[e] -= (double)(T)((one - p) * om);
            }
            // :166-185
            const double pos_tp = tp[e] + (double)t;
            const double pos_fp = fp[e] + (double)om;
            const double neg_fn = fn[e] + (double)t;
            double neg_tn = tn[e];
            if (!P.skip_tn) neg_tn = tn[e] + (double)om;
            double g = metric_eval(P.metric, pos_tp / nn, pos_fp / nn, fn[e] / nn, tn[e] / nn) -
                       metric_eval(P.metric, tp[e] / nn, fp[e] / nn, neg_fn / nn, neg_tn / nn);
            if (!P.maximize) g = -g;
            keys[e] = j < m ? sortable_key(nan_to_neg_inf(g)) : 0ull;
            nonneg[e] = (-g) <= 0.0; // :199-200 on the negated gains (k == 0)
        }

        unsigned long long mine = 0ull;
        if (P.k > 0) {
            unsigned long long prev_key = ~0ull;
            unsigned prev_col = 0u;
            bool first = true;
            const int rounds = (int64_t)P.k < m ? P.k : (int)m;
            for (int round = 0; round < rounds; ++round) {
                unsigned long long best = 0ull;
                unsigned col = ~0u;
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const unsigned j = threadIdx.x + e * XC_DENSE_BLOCK;
                    const bool rem = first || keys[e] < prev_key || (keys[e] == prev_key && j > prev_col);
                    if (rem && keys[e] > best) best = keys[e];
                }
                best = block_umax64<XC_DENSE_BLOCK>(best, red64);
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const unsigned j = threadIdx.x + e * XC_DENSE_BLOCK;
                    const bool rem = first || keys[e] < prev_key || (keys[e] == prev_key && j > prev_col);
                    if (rem && keys[e] == best && j < col) col = j;
                }
                col = block_umin32<XC_DENSE_BLOCK>(col, red32);
                if ((col % XC_DENSE_BLOCK) == threadIdx.x) mine |= 1ull << (col / XC_DENSE_BLOCK);
                prev_key = best;
                prev_col = col;
                first = false;
            }
        }

#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int64_t j = threadIdx.x + (int64_t)e * XC_DENSE_BLOCK;
            const bool sel = P.k > 0 ? (((mine >> e) & 1ull) != 0ull) : nonneg[e];
            const T t = t_cur[e];
            const T p = sel ? one : (T)0;
            const T om = one - t;
            if (j < m) {
                if (p != p_cur[e]) pred[j] = p; // :191-200
                // :203-209
                tp[e] += (double)(T)(p * t);
                fp[e] += (double)(T)(p * om);
                fn[e] += (double)(T)((one - p) * t);
                if (!P.skip_tn) tn[e] += (double)(T)((one - p) * om);
            }
            t_cur[e] = t_nxt[e];
            p_cur[e] = p_nxt[e];
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t j = threadIdx.x + (int64_t)e * XC_DENSE_BLOCK;
        if (j < m) {
            P.tp[j] = tp[e];
            P.fp[j] = fp[e];
            P.fn[j] = fn[e];
            P.tn[j] = tn[e];
        }
    }
}

// ---- concurrent dense sweep ------------------------------------------------------
// The dense step touches every label of a row, but only the labels whose prediction flips
// change the statistics.  So, as in the CSR sweep, `n_blocks` workgroups walk the order
// interleaved (block b takes positions b, b + n_blocks, ...): each reads the four statistic
// vectors coherently (sc1, L2-served), removes its row's own contribution in registers,
// scores all m labels, selects with the same block-wide rounds as above, and pushes float64
// atomics for the flipped labels only.  Rows in flight miss each other's update (DESIGN.md
// "staleness"); n_blocks = 1 is NOT this kernel but the sequential one above.
template <typename T, int EPT>
__global__ __launch_bounds__(XC_DENSE_BLOCK) void bca_sweep_dense_conc_kernel(DenseSweepParams<T> P, xc_metric fast,
                                                                           unsigned long long *changed) {
    __shared__ unsigned long long red64[XC_DENSE_BLOCK / XC_WAVE];
    __shared__ unsigned red32[XC_DENSE_BLOCK / XC_WAVE];
    const int64_t m = P.m;
    const T one = (T)1;
    unsigned long long n_changed = 0;
    for (int64_t pos = blockIdx.x; pos < P.n_order; pos += gridDim.x) {
        const int64_t row = P.order ? (int64_t)P.order[pos] : pos;
        const T *prob = P.y_proba + row * m;
        T *pred = P.y_pred + row * m;
        T t_cur[EPT], p_cur[EPT];
        unsigned long long keys[EPT];
        bool nonneg[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int64_t j = threadIdx.x + (int64_t)e * XC_DENSE_BLOCK;
            const bool v = j < m;
            const T t = v ? prob[j] : (T)0, p = v ? pred[j] : (T)0;
            t_cur[e] = t;
            p_cur[e] = p;
            const T om = one - t;
            // block_coordinate.py:157-163 in registers, on statistics other rows keep updating
            double tp = v ? load_coherent(P.tp + j) : 0.0, fp = v ? load_coherent(P.fp + j) : 0.0;
            double fn = v ? load_coherent(P.fn + j) : 0.0, tn = v ? load_coherent(P.tn + j) : 0.0;
            tp -= (double)(T)(p * t);
            fp -= (double)(T)(p * om);
            fn -= (double)(T)((one - p) * t);
            if (!P.skip_tn) tn -= (double)(T)((one - p) * om);
            // :166-185; psi(x / n; eps, k) = psi(x; eps * n, k * n): `fast` carries the rescaled constants
            double neg_tn = tn;
            if (!P.skip_tn) neg_tn = tn + (double)om;
            double g = metric_eval_t<false>(fast, tp + (double)t, fp + (double)om, fn, tn) -
                       metric_eval_t<false>(fast, tp, fp, fn + (double)t, neg_tn);
            if (!P.maximize) g = -g;
            keys[e] = v ? sortable_key(nan_to_neg_inf(g)) : 0ull;
            nonneg[e] = (-g) <= 0.0; // :199-200 on the negated gains (k == 0)
        }

        unsigned long long mine = 0ull;
        if (P.k > 0) {
            unsigned long long prev_key = ~0ull;
            unsigned prev_col = 0u;
            bool first = true;
            const int rounds = (int64_t)P.k < m ? P.k : (int)m;
            for (int round = 0; round < rounds; ++round) {
                unsigned long long best = 0ull;
                unsigned col = ~0u;
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const unsigned j = threadIdx.x + e * XC_DENSE_BLOCK;
                    const bool rem = first || keys[e] < prev_key || (keys[e] == prev_key && j > prev_col);
                    if (rem && keys[e] > best) best = keys[e];
                }
                best = block_umax64<XC_DENSE_BLOCK>(best, red64);
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const unsigned j = threadIdx.x + e * XC_DENSE_BLOCK;
                    const bool rem = first || keys[e] < prev_key || (keys[e] == prev_key && j > prev_col);
                    if (rem && keys[e] == best && j < col) col = j;
                }
                col = block_umin32<XC_DENSE_BLOCK>(col, red32);
                if ((col % XC_DENSE_BLOCK) == threadIdx.x) mine |= 1ull << (col / XC_DENSE_BLOCK);
                prev_key = best;
                prev_col = col;
                first = false;
            }
        }

        bool flipped = false;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int64_t j = threadIdx.x + (int64_t)e * XC_DENSE_BLOCK;
            const bool sel = P.k > 0 ? (((mine >> e) & 1ull) != 0ull) : nonneg[e];
            const T p = sel ? one : (T)0;
            if (j < m && p != p_cur[e]) { // :191-209 net of :157-163: only a flipped label moves
                const T t = t_cur[e], om = one - t;
                const double d = (double)p - (double)p_cur[e]; // +1 or -1 for 0/1 predictions
                pred[j] = p;
                atomic_add_f64(P.tp + j, d * (double)t);
                atomic_add_f64(P.fp + j, d * (double)om);
                atomic_add_f64(P.fn + j, -d * (double)t);
                if (!P.skip_tn) atomic_add_f64(P.tn + j, -d * (double)om);
                flipped = true;
            }
        }
        if (__syncthreads_or(flipped ? 1 : 0)) ++n_changed;
    }
    if (threadIdx.x == 0 && n_changed && changed) atomicAdd(changed, n_changed);
}

// ---- a 0/1 prediction with exactly k ones per row -> k column ids per row (ascending) and the scores there ----
// one wavefront per row; ballot / popcount compaction in column order
template <typename P, typename G>
__global__ __launch_bounds__(XC_BLOCK) void dense_pred_to_fixed_kernel(int64_t n, int64_t m, int64_t ld_pred, const P *y_pred,
                                                                       int64_t ld_gain, const G *gains, int k,
                                                                       int32_t *out_idx, G *out_val) {
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * (XC_BLOCK / XC_WAVE) + (threadIdx.x >> 6);
    if (row >= n) return;
    const P *p = y_pred + row * ld_pred;
    int base = 0;
    for (int64_t j0 = 0; j0 < m && base < k; j0 += XC_WAVE) {
        const int64_t j = j0 + lane;
        const bool on = j < m && p[j] != (P)0;
        const unsigned long long mask = __ballot(on);
        const int slot = base + __popcll(mask & lanemask_lt());
        if (on && slot < k) {
            out_idx[row * k + slot] = (int32_t)j;
            if (out_val) out_val[row * k + slot] = gains[row * ld_gain + j];
        }
        base += __popcll(mask);
    }
    for (int q = base + lane; q < k; q += XC_WAVE) { // fewer than k ones (m < k): pad like the CSR top-k does
        out_idx[row * k + q] = 0;
        if (out_val) out_val[row * k + q] = (G)0;
    }
}

} // namespace xc

extern "C" {

int xc_dense_pred_to_fixed(int64_t n, int64_t m, const void *y_pred, int pdtype, const void *gains, int gdtype, int k,
                           int32_t *out_idx, void *out_val, void *stream) {
    if (n < 0 || m < 0 || k < 1 || (n * m > 0 && (!y_pred || !out_idx)) || (out_val && !gains))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_dense_pred_to_fixed: bad argument");
    if ((pdtype != XC_F32 && pdtype != XC_F64) || (gdtype != XC_F32 && gdtype != XC_F64))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_dense_pred_to_fixed: unknown dtype");
    if (n == 0) return XC_OK;
    hipStream_t st = xc::as_stream(stream);
    dim3 grid((unsigned)((n + 3) / 4)), block(XC_BLOCK);
#define XC_P2F(PT, GT)                                                                                                   \
    hipLaunchKernelGGL((xc::dense_pred_to_fixed_kernel<PT, GT>), grid, block, 0, st, n, m, m, static_cast<const PT *>(y_pred), \
                       m, static_cast<const GT *>(gains), k, out_idx, static_cast<GT *>(out_val))
    if (pdtype == XC_F32 && gdtype == XC_F32) XC_P2F(float, float);
    else if (pdtype == XC_F32) XC_P2F(float, double);
    else if (gdtype == XC_F32) XC_P2F(double, float);
    else XC_P2F(double, double);
#undef XC_P2F
    XC_CHECK_LAUNCH("dense_pred_to_fixed_kernel");
    return XC_OK;
}

int xc_topk_dense(int64_t n, int64_t m, int64_t ld, const void *gains, int gdtype, int k, double th,
                  int keep_scores, void *y_pred, int pdtype, void *stream) {
    if (n < 0 || m < 0 || ld < m || (n * m > 0 && (!gains || !y_pred)))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_topk_dense: NULL pointer or bad size");
    if (k < 0) return xc::fail_arg(XC_ERR_K_RANGE, "xc_topk_dense: k=%d is negative", k);
    if ((gdtype != XC_F32 && gdtype != XC_F64) || (pdtype != XC_F32 && pdtype != XC_F64))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_topk_dense: unknown dtype");
    if (m > (int64_t)INT32_MAX) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_topk_dense: m too large");
    if (n == 0 || m == 0) return XC_OK;
    hipStream_t st = xc::as_stream(stream);
    dim3 grid((unsigned)n), block(XC_BLOCK);
    if (gdtype == XC_F32 && k > 0 && m <= 32 * XC_BLOCK) { // the row fits in registers: one read, one write
        const float *gf = static_cast<const float *>(gains);
#define XC_TOPK_REG(PT, EPT)                                                                                     \
    hipLaunchKernelGGL((xc::topk_dense_reg_kernel<PT, EPT>), grid, block, 0, st, m, ld, gf, k, keep_scores,      \
                       static_cast<PT *>(y_pred))
#define XC_TOPK_REG_BY_M(PT)                       \
    do {                                           \
        if (m <= 2 * XC_BLOCK) XC_TOPK_REG(PT, 2); \
        else if (m <= 4 * XC_BLOCK) XC_TOPK_REG(PT, 4); \
        else if (m <= 8 * XC_BLOCK) XC_TOPK_REG(PT, 8); \
        else if (m <= 16 * XC_BLOCK) XC_TOPK_REG(PT, 16); \
        else XC_TOPK_REG(PT, 32);                  \
    } while (0)
        if (pdtype == XC_F32) XC_TOPK_REG_BY_M(float);
        else XC_TOPK_REG_BY_M(double);
#undef XC_TOPK_REG_BY_M
#undef XC_TOPK_REG
    } else if (gdtype == XC_F32 && pdtype == XC_F32)
        hipLaunchKernelGGL((xc::topk_dense_kernel<float, float>), grid, block, 0, st, m, ld, static_cast<const float *>(gains), k, (float)th, keep_scores, static_cast<float *>(y_pred));
    else if (gdtype == XC_F64 && pdtype == XC_F32)
        hipLaunchKernelGGL((xc::topk_dense_kernel<double, float>), grid, block, 0, st, m, ld, static_cast<const double *>(gains), k, th, keep_scores, static_cast<float *>(y_pred));
    else if (gdtype == XC_F32 && pdtype == XC_F64)
        hipLaunchKernelGGL((xc::topk_dense_kernel<float, double>), grid, block, 0, st, m, ld, static_cast<const float *>(gains), k, (float)th, keep_scores, static_cast<double *>(y_pred));
    else
        hipLaunchKernelGGL((xc::topk_dense_kernel<double, double>), grid, block, 0, st, m, ld, static_cast<const double *>(gains), k, th, keep_scores, static_cast<double *>(y_pred));
    XC_CHECK_LAUNCH("topk_dense_kernel");
    return XC_OK;
}

int xc_bca_sweep_dense(int64_t n_order, const int32_t *order, int64_t n_norm, int64_t m, const void *y_proba,
                       void *y_pred, int dtype, int k, double *stats, double *workspace,
                       const xc_metric *metric_host, int maximize, int greedy, int skip_tn, void *stream) {
    if (n_order < 0 || n_norm < 1 || m < 1 || !y_proba || !y_pred || !stats || !workspace || !metric_host)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_sweep_dense: NULL pointer or bad size");
    if (k < 0) return xc::fail_arg(XC_ERR_K_RANGE, "xc_bca_sweep_dense: k=%d is negative", k);
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_sweep_dense: unknown dtype %d", dtype);
    if (metric_host->base < 0 || metric_host->base >= XC_M_COUNT)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_sweep_dense: unknown metric %d", metric_host->base);
    if (m > (int64_t)XC_DENSE_BLOCK * XC_DENSE_MAX_EPT)
        return xc::fail_arg(XC_ERR_ROW_TOO_LONG, "xc_bca_sweep_dense: m=%lld exceeds %d", (long long)m, XC_DENSE_BLOCK * XC_DENSE_MAX_EPT);
    if (n_order == 0) return XC_OK;
    hipStream_t st = xc::as_stream(stream);
    if (dtype == XC_F32) {
        xc::DenseSweepParams<float> P{n_order, order, m, static_cast<const float *>(y_proba), static_cast<float *>(y_pred), k,
                                      stats, stats + m, stats + 2 * m, stats + 3 * m, workspace, *metric_host,
                                      (double)n_norm, maximize, greedy, skip_tn};
        if (m <= 1 * XC_DENSE_BLOCK) hipLaunchKernelGGL((xc::bca_sweep_dense_reg_kernel<float, 1>), dim3(1), dim3(XC_DENSE_BLOCK), 0, st, P);
        else if (m <= 4 * XC_DENSE_BLOCK) hipLaunchKernelGGL((xc::bca_sweep_dense_reg_kernel<float, 4>), dim3(1), dim3(XC_DENSE_BLOCK), 0, st, P);
        else if (m <= 8 * XC_DENSE_BLOCK) hipLaunchKernelGGL((xc::bca_sweep_dense_reg_kernel<float, 8>), dim3(1), dim3(XC_DENSE_BLOCK), 0, st, P);
        else hipLaunchKernelGGL((xc::bca_sweep_dense_kernel<float, 0>), dim3(1), dim3(XC_DENSE_BLOCK), 0, st, P);
    } else {
        xc::DenseSweepParams<double> P{n_order, order, m, static_cast<const double *>(y_proba), static_cast<double *>(y_pred), k,
                                       stats, stats + m, stats + 2 * m, stats + 3 * m, workspace, *metric_host,
                                       (double)n_norm, maximize, greedy, skip_tn};
        if (m <= 1 * XC_DENSE_BLOCK) hipLaunchKernelGGL((xc::bca_sweep_dense_reg_kernel<double, 1>), dim3(1), dim3(XC_DENSE_BLOCK), 0, st, P);
        else if (m <= 4 * XC_DENSE_BLOCK) hipLaunchKernelGGL((xc::bca_sweep_dense_reg_kernel<double, 4>), dim3(1), dim3(XC_DENSE_BLOCK), 0, st, P);
        else if (m <= 8 * XC_DENSE_BLOCK) hipLaunchKernelGGL((xc::bca_sweep_dense_reg_kernel<double, 8>), dim3(1), dim3(XC_DENSE_BLOCK), 0, st, P);
        else hipLaunchKernelGGL((xc::bca_sweep_dense_kernel<double, 0>), dim3(1), dim3(XC_DENSE_BLOCK), 0, st, P);
    }
    XC_CHECK_LAUNCH("bca_sweep_dense_kernel");
    return XC_OK;
}

int xc_bca_sweep_dense_concurrent(int64_t n_order, const int32_t *order, int64_t n_norm, int64_t m,
                                  const void *y_proba, void *y_pred, int dtype, int k, double *stats,
                                  const xc_metric *metric_host, int maximize, int skip_tn, int n_blocks,
                                  int64_t *changed, void *stream) {
    if (n_order < 0 || n_norm < 1 || m < 1 || !y_proba || !y_pred || !stats || !metric_host || n_blocks < 1)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_sweep_dense_concurrent: NULL pointer or bad size");
    if (k < 0) return xc::fail_arg(XC_ERR_K_RANGE, "xc_bca_sweep_dense_concurrent: k=%d is negative", k);
    if (dtype != XC_F32 && dtype != XC_F64)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_sweep_dense_concurrent: unknown dtype %d", dtype);
    if (metric_host->base < 0 || metric_host->base >= XC_M_COUNT)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_sweep_dense_concurrent: unknown metric %d", metric_host->base);
    if (m > (int64_t)XC_DENSE_BLOCK * 8)
        return xc::fail_arg(XC_ERR_ROW_TOO_LONG, "xc_bca_sweep_dense_concurrent: m=%lld exceeds %d", (long long)m,
                            XC_DENSE_BLOCK * 8);
    if (n_order == 0) return XC_OK;
    if (n_blocks > n_order) n_blocks = (int)n_order;
    hipStream_t st = xc::as_stream(stream);
    xc_metric fast = *metric_host; // psi(x / n; eps, k) = psi(x; eps * n, k * n)
    fast.epsilon *= (double)n_norm;
    fast.kf *= (double)n_norm;
    unsigned long long *ch = reinterpret_cast<unsigned long long *>(changed);
    if (dtype == XC_F32) {
        xc::DenseSweepParams<float> P{n_order, order, m, static_cast<const float *>(y_proba), static_cast<float *>(y_pred), k,
                                      stats, stats + m, stats + 2 * m, stats + 3 * m, nullptr, *metric_host,
                                      (double)n_norm, maximize, 0, skip_tn};
        if (m <= 1 * XC_DENSE_BLOCK) hipLaunchKernelGGL((xc::bca_sweep_dense_conc_kernel<float, 1>), dim3(n_blocks), dim3(XC_DENSE_BLOCK), 0, st, P, fast, ch);
        else if (m <= 4 * XC_DENSE_BLOCK) hipLaunchKernelGGL((xc::bca_sweep_dense_conc_kernel<float, 4>), dim3(n_blocks), dim3(XC_DENSE_BLOCK), 0, st, P, fast, ch);
        else hipLaunchKernelGGL((xc::bca_sweep_dense_conc_kernel<float, 8>), dim3(n_blocks), dim3(XC_DENSE_BLOCK), 0, st, P, fast, ch);
    } else {
        xc::DenseSweepParams<double> P{n_order, order, m, static_cast<const double *>(y_proba), static_cast<double *>(y_pred), k,
                                       stats, stats + m, stats + 2 * m, stats + 3 * m, nullptr, *metric_host,
                                       (double)n_norm, maximize, 0, skip_tn};
        if (m <= 1 * XC_DENSE_BLOCK) hipLaunchKernelGGL((xc::bca_sweep_dense_conc_kernel<double, 1>), dim3(n_blocks), dim3(XC_DENSE_BLOCK), 0, st, P, fast, ch);
        else if (m <= 4 * XC_DENSE_BLOCK) hipLaunchKernelGGL((xc::bca_sweep_dense_conc_kernel<double, 4>), dim3(n_blocks), dim3(XC_DENSE_BLOCK), 0, st, P, fast, ch);
        else hipLaunchKernelGGL((xc::bca_sweep_dense_conc_kernel<double, 8>), dim3(n_blocks), dim3(XC_DENSE_BLOCK), 0, st, P, fast, ch);
    }
    XC_CHECK_LAUNCH("bca_sweep_dense_conc_kernel");
    return XC_OK;
}

} // extern "C"

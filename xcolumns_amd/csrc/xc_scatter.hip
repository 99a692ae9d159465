// xc_scatter.hip -- per-label sums of many (label, value) pairs without one global float atomic per pair.
//
// The column sums of y_proba (s = tp + fn, block_coordinate.py:430-436 via calculate_confusion_matrix) and the
// from-scratch {tp, fp} of an initial prediction are scatter-adds of tens of millions of float32 values into an
// m-vector.  One global float64 atomic per value runs at the memory side at ~23.5 G adds/s whatever its scope
// (tools/atomic_scope_probe.hip): 2.1 ms for the 50 M entries of a 1 M x 500 K matrix.  Here the pairs are first
// distributed into buckets of 2^s consecutive labels (a counting sort: count, scan, scatter -- all traffic is
// streaming), then one workgroup per bucket sums its pairs in an LDS table with LDS atomics and writes its slice
// of the result: three streaming passes over 8 B per pair instead of 50 M atomics at the memory side.
#include "xc_common.h"
#include "xc_host.h"

namespace xc {

#define XC_SC_CHUNKS_MAX 2048
#define XC_SC_BUCKET_LABELS_MAX 2048 /* LDS table of a bucket: 16 KB of float64 (32 KB with pairs) */
#define XC_SC_BUCKETS_MAX 16384       /* per-chunk bucket counters in LDS: 64 KB */

struct ScatterPlan {
    int64_t n_items, m;
    int shift;     // bucket = label >> shift
    int n_buckets;
    int n_chunks;
    int64_t chunk; // items per chunk
    int pair;      // 0: out[label] += v;  1: out[2 label] += v, out[2 label + 1] += (1 - v) (float32 subtraction, :253)
};

__global__ __launch_bounds__(XC_BLOCK) void scatter_count_kernel(ScatterPlan S, const int32_t *idx, int32_t *counts) {
    extern __shared__ int s_hist[];
    for (int b = threadIdx.x; b < S.n_buckets; b += XC_BLOCK) s_hist[b] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * S.chunk, hi = lo + S.chunk < S.n_items ? lo + S.chunk : S.n_items;
    for (int64_t t = lo + threadIdx.x; t < hi; t += XC_BLOCK) atomicAdd(&s_hist[idx[t] >> S.shift], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < S.n_buckets; b += XC_BLOCK) counts[(int64_t)b * S.n_chunks + blockIdx.x] = s_hist[b];
}

// one workgroup per bucket: exclusive scan of its column of counts over the chunks, and the bucket's total
__global__ __launch_bounds__(XC_BLOCK) void scatter_scan_chunks_kernel(ScatterPlan S, int32_t *counts, int32_t *totals) {
    __shared__ int s_part[XC_BLOCK];
    int32_t *col = counts + (int64_t)blockIdx.x * S.n_chunks;
    const int per = (S.n_chunks + XC_BLOCK - 1) / XC_BLOCK;
    const int g0 = threadIdx.x * per, g1 = g0 + per < S.n_chunks ? g0 + per : S.n_chunks;
    int sum = 0;
    for (int g = g0; g < g1; ++g) sum += col[g];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < XC_BLOCK; ++i) {
            const int v = s_part[i];
            s_part[i] = run;
            run += v;
        }
        totals[blockIdx.x] = run;
    }
    __syncthreads();
    int run = s_part[threadIdx.x];
    for (int g = g0; g < g1; ++g) {
        const int v = col[g];
        col[g] = run;
        run += v;
    }
}

// exclusive scan of the bucket totals (one workgroup; the totals are staged in LDS so the serial part runs on LDS latency)
__global__ __launch_bounds__(XC_BLOCK) void scatter_scan_buckets_kernel(int n_buckets, const int32_t *totals, int64_t *base) {
    extern __shared__ int s_tot[];
    for (int b = threadIdx.x; b < n_buckets; b += XC_BLOCK) s_tot[b] = totals[b];
    __syncthreads();
    __shared__ long long s_part[XC_BLOCK];
    const int per = (n_buckets + XC_BLOCK - 1) / XC_BLOCK;
    const int b0 = threadIdx.x * per, b1 = b0 + per < n_buckets ? b0 + per : n_buckets;
    long long sum = 0;
    for (int b = b0; b < b1; ++b) sum += s_tot[b];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int i = 0; i < XC_BLOCK; ++i) {
            const long long v = s_part[i];
            s_part[i] = run;
            run += v;
        }
        base[n_buckets] = run;
    }
    __syncthreads();
    long long run = s_part[threadIdx.x];
    for (int b = b0; b < b1; ++b) {
        base[b] = run;
        run += s_tot[b];
    }
}

struct __attribute__((aligned(8))) sc_item_t {
    int32_t label;
    float value;
};

__global__ __launch_bounds__(XC_BLOCK) void scatter_move_kernel(ScatterPlan S, const int32_t *idx, const float *val,
                                                                const int32_t *counts, const int64_t *base, sc_item_t *items) {
    extern __shared__ int s_cur[]; // position (relative to the bucket's base) of this chunk's next pair, per bucket
    for (int b = threadIdx.x; b < S.n_buckets; b += XC_BLOCK) s_cur[b] = counts[(int64_t)b * S.n_chunks + blockIdx.x];
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * S.chunk, hi = lo + S.chunk < S.n_items ? lo + S.chunk : S.n_items;
    for (int64_t t = lo + threadIdx.x; t < hi; t += XC_BLOCK) {
        const int32_t label = idx[t];
        const int b = label >> S.shift;
        const int p = atomicAdd(&s_cur[b], 1);
        sc_item_t it;
        it.label = label;
        it.value = val[t];
        items[base[b] + p] = it;
    }
}

__global__ __launch_bounds__(XC_BLOCK) void scatter_reduce_kernel(ScatterPlan S, const int64_t *base, const sc_item_t *items,
                                                                  double *out) {
    extern __shared__ double s_acc[];
    const int labels = 1 << S.shift;
    const int width = S.pair ? 2 * labels : labels;
    for (int i = threadIdx.x; i < width; i += XC_BLOCK) s_acc[i] = 0.0;
    __syncthreads();
    const int64_t lo = base[blockIdx.x], hi = base[blockIdx.x + 1];
    const int32_t first = (int32_t)blockIdx.x << S.shift;
    for (int64_t t = lo + threadIdx.x; t < hi; t += XC_BLOCK) {
        const sc_item_t it = items[t];
        const int j = it.label - first;
        if (S.pair) {
            atomicAdd(&s_acc[2 * j], (double)it.value);
            atomicAdd(&s_acc[2 * j + 1], (double)(1.0f - it.value));
        } else {
            atomicAdd(&s_acc[j], (double)it.value);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < width; i += XC_BLOCK) {
        const int64_t label = (int64_t)first + (S.pair ? i / 2 : i);
        if (label < S.m) out[(S.pair ? 2 * (int64_t)first : (int64_t)first) + i] = s_acc[i];
    }
}

static ScatterPlan make_plan(int64_t n_items, int64_t m, int pair) {
    ScatterPlan S;
    S.n_items = n_items;
    S.m = m;
    S.pair = pair;
    int shift = 6;
    while (((m + (1ll << shift) - 1) >> shift) > 512 && (1 << shift) < XC_SC_BUCKET_LABELS_MAX) ++shift;
    S.shift = shift;
    S.n_buckets = (int)((m + (1ll << shift) - 1) >> shift);
    // many more chunks than CUs (an even load; 256 CUs take a 305-chunk grid in two uneven rounds), at least 2048 pairs each
    int64_t chunks = (n_items + 2047) / 2048;
    if (chunks > XC_SC_CHUNKS_MAX) chunks = XC_SC_CHUNKS_MAX;
    if (chunks < 1) chunks = 1;
    S.n_chunks = (int)chunks;
    S.chunk = (n_items + chunks - 1) / chunks;
    return S;
}

} // namespace xc

extern "C" {

// out[label] (pair = 0) or out[2 label], out[2 label + 1] (pair = 1: value and 1 - value) <- sums over the n_items
// (idx, val) pairs; every label of [0, m) is written (zeros where no pair falls).  float32 values, float64 sums.
int xc_scatter_sum_workspace_bytes(int64_t n_items, int64_t m, int64_t *bytes) {
    if (!bytes || n_items < 0 || m < 1) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_scatter_sum_workspace_bytes: bad argument");
    const xc::ScatterPlan S = xc::make_plan(n_items, m, 0);
    *bytes = (int64_t)S.n_buckets * S.n_chunks * 4 + (int64_t)S.n_buckets * 4 + 64 + ((int64_t)S.n_buckets + 1) * 8 + 64 +
             n_items * 8 + 64;
    return XC_OK;
}

int xc_scatter_sum_f32(int64_t n_items, const int32_t *idx, const float *val, int64_t m, int pair, double *out,
                       void *workspace, void *stream) {
    if (n_items < 0 || m < 1 || !out || !workspace || (n_items > 0 && (!idx || !val)))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_scatter_sum_f32: bad argument");
    if (m > (int64_t)0x7fffffff) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_scatter_sum_f32: m too large");
    const xc::ScatterPlan S = xc::make_plan(n_items, m, pair ? 1 : 0);
    if (S.n_buckets > XC_SC_BUCKETS_MAX)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_scatter_sum_f32: label space too large for the bucket tables (m <= %lld)",
                            (long long)XC_SC_BUCKETS_MAX * XC_SC_BUCKET_LABELS_MAX);
    hipStream_t st = xc::as_stream(stream);
    char *w = static_cast<char *>(workspace);
    int32_t *counts = reinterpret_cast<int32_t *>(w);
    w += ((int64_t)S.n_buckets * S.n_chunks * 4 + 63) / 64 * 64;
    int32_t *totals = reinterpret_cast<int32_t *>(w);
    w += ((int64_t)S.n_buckets * 4 + 63) / 64 * 64;
    int64_t *base = reinterpret_cast<int64_t *>(w);
    w += (((int64_t)S.n_buckets + 1) * 8 + 63) / 64 * 64;
    xc::sc_item_t *items = reinterpret_cast<xc::sc_item_t *>(w);
    const size_t hist_bytes = (size_t)S.n_buckets * 4;
    hipLaunchKernelGGL(xc::scatter_count_kernel, dim3(S.n_chunks), dim3(XC_BLOCK), hist_bytes, st, S, idx, counts);
    hipLaunchKernelGGL(xc::scatter_scan_chunks_kernel, dim3(S.n_buckets), dim3(XC_BLOCK), 0, st, S, counts, totals);
    hipLaunchKernelGGL(xc::scatter_scan_buckets_kernel, dim3(1), dim3(XC_BLOCK), hist_bytes, st, S.n_buckets, totals, base);
    hipLaunchKernelGGL(xc::scatter_move_kernel, dim3(S.n_chunks), dim3(XC_BLOCK), hist_bytes, st, S, idx, val, counts, base, items);
    const size_t acc_bytes = (size_t)(1 << S.shift) * (pair ? 16 : 8);
    hipLaunchKernelGGL(xc::scatter_reduce_kernel, dim3(S.n_buckets), dim3(XC_BLOCK), acc_bytes, st, S, base, items, out);
    XC_CHECK_LAUNCH("scatter_sum kernels");
    return XC_OK;
}

} // extern "C"

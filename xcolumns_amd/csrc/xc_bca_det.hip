// xc_bca_det.hip -- a DETERMINISTIC concurrent BCA sweep on CSR rows (float32 scores, packed row stream).
//
// The default sweep (xc_bca.hip) lets wavefronts read statistics other wavefronts are changing: which
// update a row happens to see depends on timing, and the float atomics add in arrival order, so two runs with
// the same `seed` may return predictions that differ in a few rows.  The reference is deterministic given
// `seed` (/root/reference/xcolumns/block_coordinate.py:413-419).  This file trades speed for that property
// without going back to one row at a time:
//
//   The visiting order is walked in BLOCKS of up to B rows (B = the wavefront count the policy allows).  A block is
//   three launches:
//     prepare  the block's row list: first the rows the previous block could not settle ("carried", oldest
//              first), then the next rows of the order;
//     decide   one wavefront per row scores its candidates on the per-label records AS THEY STOOD WHEN THE BLOCK
//              BEGAN (nothing writes them during this launch) and takes the top-k -- the arithmetic of the
//              concurrent sweep.  A row that wants to change labels claims each of them with an integer
//              atomicMax of (block number, seniority in the block);
//     commit   a row changes its prediction only if it holds EVERY claim it made; a row that lost one is carried
//              into the next block, where it is senior to every fresh row -- the oldest row always wins all its
//              claims, so every block makes progress.  Only SMALL labels are claimed (fewer than 512 predicted
//              rows, the commit protocol's conflict tolerance): a label that sums hundreds of rows cannot be moved
//              by one row, and on skewed label popularity nearly every row of the first sweep drops the same
//              head labels -- claiming those would let one row per block through.
//   During the sweep the per-label records live in 2^-38 FIXED POINT (int64 {tp, fp}): a row's change is an
//   integer atomicAdd, and a sum of integers does not depend on the order of the adds -- any number of rows may
//   update a label in one block.  The from-scratch statistics of the sweep boundary (block_coordinate.py:465-467)
//   are accumulated the same way; they agree with the float64 sums to ~1e-12 relative (a score below 2^-14 is
//   rounded to a multiple of 2^-38).
//
// Nothing in a block depends on timing: the decisions are functions of the frozen records, integer atomicMax /
// atomicAdd are order-independent, no floating-point value is accumulated.  Same seed => same prediction, bit for bit.
// Rows of one block do not see each other's changes (at most B rows of staleness, as in the default sweep) and
// two rows that want the same label are serialised by the claims (the later one re-decides a block later).
#include "xc_common.h"
#include "xc_host.h"

namespace xc {

struct __attribute__((packed, aligned(4))) det_pack3_t {
    unsigned x, y, z; // the packed row entry of xc_bca.hip: col | hot << 25 | sel << 31, eta, (float) s
};
#define XC_DET_COL_MASK 0x01ffffffu
#define XC_DET_FX_SCALE 274877906944.0 /* 2^38 */
#define XC_DET_SMALL_FX (512ll << 38)  /* tp + fp below this: the label is claimed before it is changed */
#define XC_DET_MAX_BLOCK 8192
// state words (int64) of a deterministic sweep
#define XC_DET_CURSOR 0   /* next position of the order not yet handed to a block */
#define XC_DET_NCUR 1     /* rows of the current block */
#define XC_DET_EPOCH 2    /* block number (claims of older blocks lose against any claim of this one) */
#define XC_DET_CHANGED 3  /* rows that changed their prediction in this sweep */
#define XC_DET_WHICH 4    /* which of the two row-list buffers holds the current block */
#define XC_DET_WORDS 8

struct DetParams {
    int64_t n_order;
    const int32_t *order;
    const int32_t *indptr;
    det_pack3_t *packed;
    int32_t *pred_indices;
    float *pred_eta;
    uint8_t *sel;
    long long *rec_fx;             // [m][2] the records {tp, fp} of this sweep in 2^-38 fixed point
    unsigned long long *claim;     // [m]
    long long *acc_fx;             // [2m] fixed-point from-scratch {tp, fp}
    long long *state;              // [XC_DET_WORDS]
    int32_t *blk_rows;             // [2][XC_DET_MAX_BLOCK] row ids of the block (ping-pong)
    uint8_t *blk_retry;            // [2][XC_DET_MAX_BLOCK] 1 = the row lost a claim and is carried
    unsigned long long *dec_bits;  // [XC_DET_MAX_BLOCK][2 * CH] ballots per 64-candidate chunk: the new membership, then
                                   // the lanes whose label the row claimed
    uint8_t *dec_flag;             // [XC_DET_MAX_BLOCK] 1 = the row wants to change
    int k;
    int block;                     // B
    int skip_tn;
    int maximize;
    xc_metric metric_fast;
    double nn, n_counted;
};

// ---- prepare: carried rows (in order) + fresh rows ------------------------------------------------------
__global__ __launch_bounds__(1024) void det_prepare_kernel(DetParams P) {
    __shared__ int s_cnt[1024];
    __shared__ int s_base[1024];
    long long *st = P.state;
    const int t = threadIdx.x;
    const int which = (int)st[XC_DET_WHICH];
    const int n_prev = (int)st[XC_DET_NCUR];
    const int32_t *prev_rows = P.blk_rows + which * XC_DET_MAX_BLOCK;
    const uint8_t *prev_retry = P.blk_retry + which * XC_DET_MAX_BLOCK;
    int32_t *rows = P.blk_rows + (1 - which) * XC_DET_MAX_BLOCK;
    uint8_t *retry = P.blk_retry + (1 - which) * XC_DET_MAX_BLOCK;
    // every thread owns 8 consecutive slots of the previous block: count, scan, scatter in slot order
    int mine[8], c = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int s = t * 8 + q;
        mine[q] = (s < n_prev && prev_retry[s]) ? 1 : 0;
        c += mine[q];
    }
    s_cnt[t] = c;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int i = 0; i < 1024; ++i) {
            s_base[i] = run;
            run += s_cnt[i];
        }
        s_cnt[0] = run; // carried rows in total
    }
    __syncthreads();
    const int carried = s_cnt[0];
    int o = s_base[t];
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (mine[q]) {
            rows[o] = prev_rows[t * 8 + q];
            retry[o] = 0;
            ++o;
        }
    const long long cursor = st[XC_DET_CURSOR];
    long long take = P.n_order - cursor;
    if (take > P.block - carried) take = P.block - carried;
    if (take < 0) take = 0;
    for (int s = t; s < (int)take; s += 1024) {
        const long long pos = cursor + s;
        rows[carried + s] = P.order ? P.order[pos] : (int32_t)pos;
        retry[carried + s] = 0;
    }
    __syncthreads();
    if (t == 0) {
        st[XC_DET_CURSOR] = cursor + take;
        st[XC_DET_NCUR] = carried + take;
        st[XC_DET_EPOCH] = st[XC_DET_EPOCH] + 1;
        st[XC_DET_WHICH] = 1 - which;
    }
}

template <int CH>
struct DetRow {
    int idx[CH];
    float eta[CH];
    float sc[CH];
    bool sel[CH];
};

template <int CH>
__device__ __forceinline__ void det_load_row(const DetParams &P, int s, int r, int lane, DetRow<CH> &d) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int p = lane + XC_WAVE * c;
        const int pc = p < r ? p : (r > 0 ? r - 1 : 0);
        const det_pack3_t *e = P.packed + s + pc;
        const unsigned wx = e->x;
        d.idx[c] = (int)(wx & XC_DET_COL_MASK);
        d.sel[c] = (wx >> 31) != 0 && p < r;
        d.eta[c] = __uint_as_float(e->y);
        d.sc[c] = __uint_as_float(e->z);
    }
}

// ---- decide ---------------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(XC_BLOCK) void det_decide_kernel(DetParams P) {
    const int lane = lane_id();
    const int slot = blockIdx.x * (XC_BLOCK / XC_WAVE) + (threadIdx.x >> 6);
    const int n_cur = (int)P.state[XC_DET_NCUR];
    if (slot >= n_cur) return;
    const int which = (int)P.state[XC_DET_WHICH];
    const int row = P.blk_rows[which * XC_DET_MAX_BLOCK + slot];
    const int s0 = P.indptr[row], r = P.indptr[row + 1] - s0;
    const int k = P.k, kk = r < k ? r : k;
    DetRow<CH> cur;
    det_load_row<CH>(P, s0, r, lane, cur);
    unsigned long long key[CH];
    bool small[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        key[c] = 0ull;
        small[c] = false;
        if (lane + XC_WAVE * c < r) {
            // the records as they stood when the block began: nothing writes them during this launch
            const longlong2 rec = *reinterpret_cast<const longlong2 *>(P.rec_fx + (int64_t)cur.idx[c] * 2);
            small[c] = rec.x + rec.y < XC_DET_SMALL_FX;
            const float e = cur.eta[c];
            const double ed = (double)e, omd = (double)(1.0f - e);
            double tpc = (double)rec.x * (1.0 / XC_DET_FX_SCALE), fpc = (double)rec.y * (1.0 / XC_DET_FX_SCALE);
            if (cur.sel[c]) { // statistics without this row (block_coordinate.py:243-246, in registers)
                tpc -= ed;
                fpc -= omd;
            }
            const double scc = (double)cur.sc[c] - ed;
            const double fn = scc - tpc;
            const double tn = (P.n_counted - 1.0) - fpc - scc;
            double pos_tn = -P.nn, neg_tn = -P.nn; // skip_tn: the constant -1 of the reference, times n
            if (!P.skip_tn) {
                neg_tn = tn + omd;
                pos_tn = tn;
            }
            double g = metric_eval_t<false>(P.metric_fast, tpc + ed, fpc + omd, fn, pos_tn) -
                       metric_eval_t<false>(P.metric_fast, tpc, fpc, fn + ed, neg_tn);
            if (!P.maximize) g = -g;
            key[c] = sortable_key(nan_to_neg_inf(g));
        }
    }
    // the k-th largest key by bisection on the key bits (ties: lower position = lower column), as in xc_bca.hip
    bool in_new[CH];
    unsigned long long thr = 0ull;
    int n_ge = 0;
    for (int bit = 63; bit >= 0; --bit) {
        const unsigned long long cand = thr | (1ull << bit);
        int cnt = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) cnt += __popcll(__ballot(key[c] >= cand));
        if (cnt >= kk) {
            thr = cand;
            n_ge = cnt;
            if (cnt == kk) break;
        }
    }
    if (n_ge == kk) {
#pragma unroll
        for (int c = 0; c < CH; ++c) in_new[c] = key[c] >= thr && key[c] != 0ull;
    } else {
        int n_gt = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) n_gt += __popcll(__ballot(key[c] > thr));
        int need = kk - n_gt;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const bool eq = key[c] == thr && key[c] != 0ull;
            const unsigned long long m_eq = __ballot(eq);
            const int before = __popcll(m_eq & lanemask_lt());
            in_new[c] = (key[c] > thr) || (eq && before < need);
            need -= __popcll(m_eq);
            if (need < 0) need = 0;
        }
    }
    bool any = false;
#pragma unroll
    for (int c = 0; c < CH; ++c) any = any || (in_new[c] != cur.sel[c]);
    const bool changed = __ballot(any) != 0ull;
    // seniority: slot 0 is the oldest row of the block; a larger claim value wins
    const unsigned long long mine = ((unsigned long long)P.state[XC_DET_EPOCH] << 16) | (unsigned long long)(65535 - slot);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const bool claims = changed && in_new[c] != cur.sel[c] && small[c];
        const unsigned long long bits = __ballot(in_new[c]), cbits = __ballot(claims);
        if (lane == 0) {
            P.dec_bits[(int64_t)slot * 2 * CH + c] = bits;
            P.dec_bits[(int64_t)slot * 2 * CH + CH + c] = cbits;
        }
        if (claims) atomicMax(P.claim + cur.idx[c], mine);
    }
    if (lane == 0) P.dec_flag[slot] = changed ? 1 : 0;
}

// ---- commit ---------------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(XC_BLOCK) void det_commit_kernel(DetParams P) {
    const int lane = lane_id();
    const int slot = blockIdx.x * (XC_BLOCK / XC_WAVE) + (threadIdx.x >> 6);
    const int n_cur = (int)P.state[XC_DET_NCUR];
    if (slot >= n_cur) return;
    const int which = (int)P.state[XC_DET_WHICH];
    const int row = P.blk_rows[which * XC_DET_MAX_BLOCK + slot];
    const int s0 = P.indptr[row], r = P.indptr[row + 1] - s0;
    const int k = P.k;
    DetRow<CH> cur;
    det_load_row<CH>(P, s0, r, lane, cur);
    const bool changed = P.dec_flag[slot] != 0;
    const unsigned long long mine = ((unsigned long long)P.state[XC_DET_EPOCH] << 16) | (unsigned long long)(65535 - slot);
    bool in_new[CH];
    bool lost = false;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        in_new[c] = (P.dec_bits[(int64_t)slot * 2 * CH + c] >> lane) & 1ull;
        const bool claimed = (P.dec_bits[(int64_t)slot * 2 * CH + CH + c] >> lane) & 1ull;
        if (claimed) lost = lost || (P.claim[cur.idx[c]] != mine);
    }
    if (changed && __ballot(lost) != 0ull) { // another row of this block holds one of the labels: decide again next block
        if (lane == 0) P.blk_retry[which * XC_DET_MAX_BLOCK + slot] = 1;
        return;
    }
    // this row is settled for the sweep: its (new) prediction enters the from-scratch statistics ...
    int base = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if (in_new[c]) {
            const float e = cur.eta[c];
            atomicAdd(reinterpret_cast<unsigned long long *>(P.acc_fx) + (int64_t)cur.idx[c] * 2,
                      (unsigned long long)__double2ll_rn((double)e * XC_DET_FX_SCALE));
            atomicAdd(reinterpret_cast<unsigned long long *>(P.acc_fx) + (int64_t)cur.idx[c] * 2 + 1,
                      (unsigned long long)__double2ll_rn((double)(1.0f - e) * XC_DET_FX_SCALE));
        }
        // ... and, if it changed, its deltas go into the fixed-point records (integer adds: any order, same sum)
        if (changed) {
            const unsigned long long mask = __ballot(in_new[c]);
            if (in_new[c]) {
                const int o = base + __popcll(mask & lanemask_lt());
                P.pred_indices[(int64_t)row * k + o] = cur.idx[c];
                P.pred_eta[(int64_t)row * k + o] = cur.eta[c];
            }
            base += __popcll(mask);
            if (lane + XC_WAVE * c < r && in_new[c] != cur.sel[c]) {
                const long long d_tp = __double2ll_rn((double)cur.eta[c] * XC_DET_FX_SCALE);
                const long long d_fp = __double2ll_rn((double)(1.0f - cur.eta[c]) * XC_DET_FX_SCALE);
                unsigned long long *rec = reinterpret_cast<unsigned long long *>(P.rec_fx) + (int64_t)cur.idx[c] * 2;
                atomicAdd(rec, (unsigned long long)(in_new[c] ? d_tp : -d_tp));
                atomicAdd(rec + 1, (unsigned long long)(in_new[c] ? d_fp : -d_fp));
                P.sel[s0 + lane + XC_WAVE * c] = in_new[c] ? 1 : 0;
                det_pack3_t *e = P.packed + s0 + lane + XC_WAVE * c;
                e->x = (e->x & 0x7fffffffu) | (in_new[c] ? 0x80000000u : 0u);
            }
        }
    }
    if (changed && lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(P.state) + XC_DET_CHANGED, 1ull);
}

// acc[j] = acc_fx[j] * 2^-38 (float64), acc_fx cleared; acc[2m] = rows changed
__global__ __launch_bounds__(XC_BLOCK) void det_acc_to_f64_kernel(int64_t m2, long long *acc_fx, double *acc, long long *state) {
    const int64_t stride = (int64_t)gridDim.x * XC_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x; j < m2; j += stride) {
        acc[j] = (double)acc_fx[j] * (1.0 / XC_DET_FX_SCALE);
        acc_fx[j] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) acc[m2] = (double)state[XC_DET_CHANGED];
}

__global__ __launch_bounds__(XC_BLOCK) void det_begin_kernel(long long *state, int64_t m2, const double *tpfp, long long *rec_fx) {
    if (blockIdx.x == 0 && threadIdx.x < XC_DET_WORDS) state[threadIdx.x] = 0;
    const int64_t stride = (int64_t)gridDim.x * XC_BLOCK;
    for (int64_t j = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x; j < m2; j += stride)
        rec_fx[j] = __double2ll_rn(tpfp[j] * XC_DET_FX_SCALE);
}

template <int CH>
static void det_launch_blocks(const DetParams &P, int iterations, hipStream_t st) {
    const int grid = (P.block + 3) / 4;
    for (int i = 0; i < iterations; ++i) {
        hipLaunchKernelGGL(det_prepare_kernel, dim3(1), dim3(1024), 0, st, P);
        hipLaunchKernelGGL((det_decide_kernel<CH>), dim3(grid), dim3(XC_BLOCK), 0, st, P);
        hipLaunchKernelGGL((det_commit_kernel<CH>), dim3(grid), dim3(XC_BLOCK), 0, st, P);
    }
}

} // namespace xc

extern "C" {

int xc_bca_det_workspace_bytes(int max_row_nnz, int64_t m, int64_t *bytes) {
    const int ch = xc::chunks_for(max_row_nnz);
    if (!bytes || ch == 0 || m < 1) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_det_workspace_bytes: bad argument");
    // claim[m] u64 | acc_fx[2m] i64 | rec_fx[2m] i64 | state | blk_rows[2][B] | dec_bits[B][2 CH] | blk_retry[2][B] | dec_flag[B]
    *bytes = m * 8 + 2 * m * 8 + 2 * m * 8 + XC_DET_WORDS * 8 + 2 * XC_DET_MAX_BLOCK * 4 +
             (int64_t)XC_DET_MAX_BLOCK * 2 * ch * 8 + 2 * XC_DET_MAX_BLOCK + XC_DET_MAX_BLOCK + 64;
    return XC_OK;
}

// Start a deterministic sweep: clears the claims, the fixed-point boundary statistics and the cursor, and takes
// the records of the sweep from the float64 master copy tpfp[m][2] (as the last boundary left it).
int xc_bca_det_begin(void *workspace, int64_t m, const double *tpfp, void *stream) {
    if (!workspace || m < 1 || !tpfp) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_det_begin: bad argument");
    hipStream_t st = xc::as_stream(stream);
    char *w = static_cast<char *>(workspace);
    XC_HIP_TRY(hipMemsetAsync(w, 0, (size_t)(m * 8 + 2 * m * 8), st));
    int64_t b = (2 * m + XC_BLOCK - 1) / XC_BLOCK;
    if (b > 8192) b = 8192;
    hipLaunchKernelGGL(xc::det_begin_kernel, dim3((int)b), dim3(XC_BLOCK), 0, st, reinterpret_cast<long long *>(w + m * 40), 2 * m,
                       tpfp, reinterpret_cast<long long *>(w + m * 24));
    XC_CHECK_LAUNCH("det_begin_kernel");
    return XC_OK;
}

// Run `iterations` blocks (prepare + decide + commit each) of a deterministic sweep; blocks after the order is
// exhausted and nothing is carried are no-ops.  Reads the cursor and the carried count back when `progress_host`
// is given (blocks on the stream): {cursor, rows of the last block} -- the sweep is over when cursor == n_order
// and the last block was empty.
int xc_bca_det_blocks(void *workspace, int64_t n_order, const int32_t *order, int64_t n_norm, const int32_t *indptr,
                      int max_row_nnz, int32_t *pred_indices, float *pred_eta, uint8_t *sel, int k, int64_t m,
                      void *packed, const xc_metric *metric_host, int maximize, int skip_tn, int block,
                      int iterations, int64_t *progress_host, void *stream) {
    if (!workspace || n_order < 0 || n_norm < 1 || m < 1 || !indptr || !pred_indices || !pred_eta || !sel ||
        !packed || !metric_host || block < 1 || block > XC_DET_MAX_BLOCK || iterations < 0)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_det_blocks: bad argument");
    if (k < 1 || k > XC_MAX_K) return xc::fail_arg(XC_ERR_K_RANGE, "xc_bca_det_blocks: k=%d outside 1..%d", k, XC_MAX_K);
    if (metric_host->base < 0 || metric_host->base >= XC_M_COUNT)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_det_blocks: unknown metric %d", metric_host->base);
    const int ch = xc::chunks_for(max_row_nnz);
    if (ch == 0) return xc::fail_arg(XC_ERR_ROW_TOO_LONG, "xc_bca_det_blocks: a row holds %d entries, limit %d", max_row_nnz, XC_MAX_ROW_NNZ);
    hipStream_t st = xc::as_stream(stream);
    char *w = static_cast<char *>(workspace);
    xc::DetParams P;
    P.n_order = n_order;
    P.order = order;
    P.indptr = indptr;
    P.packed = static_cast<xc::det_pack3_t *>(packed);
    P.pred_indices = pred_indices;
    P.pred_eta = pred_eta;
    P.sel = sel;
    P.claim = reinterpret_cast<unsigned long long *>(w);
    P.acc_fx = reinterpret_cast<long long *>(w + m * 8);
    P.rec_fx = reinterpret_cast<long long *>(w + m * 24);
    P.state = reinterpret_cast<long long *>(w + m * 40);
    char *q = w + m * 40 + XC_DET_WORDS * 8;
    P.blk_rows = reinterpret_cast<int32_t *>(q);
    q += 2 * XC_DET_MAX_BLOCK * 4;
    P.dec_bits = reinterpret_cast<unsigned long long *>(q);
    q += (size_t)XC_DET_MAX_BLOCK * 2 * ch * 8;
    P.blk_retry = reinterpret_cast<uint8_t *>(q);
    q += 2 * XC_DET_MAX_BLOCK;
    P.dec_flag = reinterpret_cast<uint8_t *>(q);
    P.k = k;
    P.block = block;
    P.skip_tn = skip_tn;
    P.maximize = maximize;
    P.metric_fast = *metric_host; // psi(x / n; eps, k) = psi(x; eps * n, k * n), as in the concurrent sweep
    P.metric_fast.epsilon *= (double)n_norm;
    P.metric_fast.kf *= (double)n_norm;
    P.nn = (double)n_norm;
    P.n_counted = (double)n_norm;
    switch (ch) {
    case 1: xc::det_launch_blocks<1>(P, iterations, st); break;
    case 2: xc::det_launch_blocks<2>(P, iterations, st); break;
    case 4: xc::det_launch_blocks<4>(P, iterations, st); break;
    case 8: xc::det_launch_blocks<8>(P, iterations, st); break;
    default: xc::det_launch_blocks<16>(P, iterations, st); break;
    }
    XC_CHECK_LAUNCH("det block kernels");
    if (progress_host) {
        long long tmp[XC_DET_WORDS];
        XC_HIP_TRY(hipMemcpyAsync(tmp, P.state, sizeof(tmp), hipMemcpyDeviceToHost, st));
        XC_HIP_TRY(hipStreamSynchronize(st));
        progress_host[0] = tmp[XC_DET_CURSOR];
        progress_host[1] = tmp[XC_DET_NCUR];
    }
    return XC_OK;
}

// The sweep boundary of a deterministic sweep: the fixed-point statistics become the float64 `acc` (2m + 1
// values, the last one the changed-row count) that xc_bca_commit_utility consumes.
int xc_bca_det_finish(void *workspace, int64_t m, double *acc, void *stream) {
    if (!workspace || m < 1 || !acc) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_det_finish: bad argument");
    char *w = static_cast<char *>(workspace);
    int64_t b = (2 * m + XC_BLOCK - 1) / XC_BLOCK;
    if (b > 8192) b = 8192;
    hipLaunchKernelGGL(xc::det_acc_to_f64_kernel, dim3((int)b), dim3(XC_BLOCK), 0, xc::as_stream(stream), 2 * m,
                       reinterpret_cast<long long *>(w + m * 8), acc, reinterpret_cast<long long *>(w + m * 40));
    XC_CHECK_LAUNCH("det_acc_to_f64_kernel");
    return XC_OK;
}

} // extern "C"

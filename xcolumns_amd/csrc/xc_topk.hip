// xc_topk.hip -- weighted per-instance top-k on CSR rows.
//
// Replaces numba_predict_weighted_per_instance_csr
// (/root/reference/xcolumns/numba_csr_functions.py:585-655) and the
// numba_argtopk_csr / numba_topk_csr helpers it calls (:455-484).
//
// One wavefront per row.  The row's (column, value) pairs are read with
// lane-contiguous 4-byte loads (a 50-entry row is two 200-byte segments), the
// optional weights a[col], b[col] are gathered, the gain is formed in the input
// dtype as multiply-then-add (never fused), and the k best are taken by k rounds
// of a wavefront arg-max.  The winners are emitted in ascending column order by
// a ballot/popcount compaction -- the row is sorted, so lane order is column
// order.  HBM-bound: 8 B per stored entry in, 4 k (+ 2 x sizeof(T) k) B out.
#include "xc_common.h"
#include <stdlib.h>

#include "xc_host.h"

namespace xc {

template <typename T>
struct TopkParams {
    int64_t n;
    const int32_t *indptr;
    const int32_t *indices;
    const T *data;
    const T *a;
    const T *b;
    const T *ab;            // optional instead of a, b: interleaved {a[col], b[col]}, ONE gather per candidate
    const int32_t *row_cls; // optional: row i uses a + row_cls[i] * ld, b + row_cls[i] * ld
    int64_t ld;
    int k;
    int keep_scores;
    int32_t *out_indices;
    T *out_data;
    T *out_eta;
    uint8_t *out_sel;
    int n_waves;
};

// sortable key of a gain in its own dtype: uint32 for float32 rows, uint64 for float64
template <typename T> struct KeyOf;
template <> struct KeyOf<float> {
    typedef unsigned type;
    static __device__ __forceinline__ type make(float g) { return sortable_key32(nan_to_neg_inf(g)); }
    static __device__ __forceinline__ type wave_max(type v) { return wave_umax32(v); }
};
template <> struct KeyOf<double> {
    typedef unsigned long long type;
    static __device__ __forceinline__ type make(double g) { return sortable_key(nan_to_neg_inf(g)); }
    static __device__ __forceinline__ type wave_max(type v) { return wave_umax64(v); }
};

template <typename T, int CH>
struct TopkRow {
    int idx[CH];
    T eta[CH];
};

template <typename T, int CH>
__device__ __forceinline__ void topk_load_row(const TopkParams<T> &P, int s, int r, int lane, TopkRow<T, CH> &d) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int p = lane + XC_WAVE * c;
        const int pc = p < r ? p : (r > 0 ? r - 1 : 0);
        // read-once streams (clamped, all lanes: straight-line loads)
        d.idx[c] = r > 0 ? __builtin_nontemporal_load(P.indices + s + pc) : 0;
        d.eta[c] = r > 0 ? __builtin_nontemporal_load(P.data + s + pc) : (T)0;
    }
}

template <typename T, int CH>
__global__ __launch_bounds__(XC_BLOCK) void topk_csr_kernel(TopkParams<T> P) {
    typedef typename KeyOf<T>::type key_t;
    const int lane = lane_id();
    const int wave = blockIdx.x * (XC_BLOCK / XC_WAVE) + (threadIdx.x >> 6);
    if (wave >= P.n_waves) return;
    const int k = P.k;
    const int64_t W = P.n_waves;
    const int64_t last = P.n - 1;
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto clampr = [&](int64_t r) { return r < last ? r : last; };

    // software pipeline: the entries of row t+1 and the indptr pair of row t+2 are
    // in flight while row t is selected
    int64_t row = wave;
    int s0 = uni(P.indptr[row]), e0 = uni(P.indptr[row + 1]);
    int s1 = uni(P.indptr[clampr(row + W)]), e1 = uni(P.indptr[clampr(row + W) + 1]);
    TopkRow<T, CH> cur;
    topk_load_row<T, CH>(P, s0, e0 - s0, lane, cur);

    for (; row < P.n; row += W) {
        const int s = s0, r = e0 - s0;
        TopkRow<T, CH> nxt;
        topk_load_row<T, CH>(P, s1, e1 - s1, lane, nxt);
        const int64_t row2 = clampr(row + 2 * W);
        const int s2 = P.indptr[row2], e2 = P.indptr[row2 + 1];

        int32_t *o_idx = P.out_indices + row * k;
        T *o_dat = P.out_data ? P.out_data + row * k : nullptr;
        T *o_eta = P.out_eta ? P.out_eta + row * k : nullptr;

        T gain[CH];
        key_t key[CH];
        bool sel[CH];
        // one weighted classifier per row (frank_wolfe.py:153-168) or one for all rows
        const int64_t w_off = P.row_cls ? (int64_t)P.row_cls[row] * P.ld : 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const bool valid = lane + XC_WAVE * c < r;
            T g = cur.eta[c];
            if (P.ab) {
                typedef T pair_t __attribute__((ext_vector_type(2)));
                const pair_t w = *reinterpret_cast<const pair_t *>(P.ab + 2 * (int64_t)cur.idx[c]);
                g = g * w.x; // numba_csr_functions.py:608-609
                g = g + w.y; // :610-611
            }
            if (P.a) g = g * P.a[w_off + cur.idx[c]];
            if (P.b) g = g + P.b[w_off + cur.idx[c]];
            gain[c] = g;
            key[c] = valid ? KeyOf<T>::make(g) : (key_t)0;
            sel[c] = false;
        }

        int n_sel;
        if (r <= k) {
            // :465-466 / :483-484: all entries, stored order
#pragma unroll
            for (int c = 0; c < CH; ++c) sel[c] = (lane + XC_WAVE * c) < r;
            n_sel = r;
        } else {
            // k rounds: wave max of the remaining keys (DPP), winner = its first
            // holder in position order (ties go to the lower column)
            for (int round = 0; round < k; ++round) {
                key_t lmax = 0;
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    if (!sel[c] && key[c] > lmax) lmax = key[c];
                const key_t M = KeyOf<T>::wave_max(lmax);
                bool found = false;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const unsigned long long mask = __ballot(!sel[c] && key[c] == M);
                    if (!found && mask != 0ull) {
                        if (lane == __ffsll((long long)mask) - 1) sel[c] = true;
                        found = true;
                    }
                }
            }
            n_sel = k;
        }

        // ascending-column emission
        int base = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const unsigned long long mask = __ballot(sel[c]);
            if (sel[c]) {
                const int slot = base + __popcll(mask & lanemask_lt());
                o_idx[slot] = cur.idx[c];
                if (o_dat) o_dat[slot] = P.keep_scores ? gain[c] : (T)1;
                if (o_eta) o_eta[slot] = cur.eta[c];
            }
            base += __popcll(mask);
            if (P.out_sel && lane + XC_WAVE * c < r) P.out_sel[s + lane + XC_WAVE * c] = sel[c] ? 1 : 0;
        }
        // :599-601: slots a short row leaves unused keep column 0 / value 1
        if (lane >= n_sel && lane < k) {
            o_idx[lane] = 0;
            if (o_dat) o_dat[lane] = (T)1;
            if (o_eta) o_eta[lane] = (T)0;
        }

        cur = nxt;
        s0 = s1; e0 = e1;
        s1 = uni(s2); e1 = uni(e2);
    }
}

// ---- float32 rows of at most 64 entries: FOUR rows per wavefront ---------------------
// The one-row-per-wave kernel above is VALU- and load-issue-bound at these row lengths (k
// serial wave reductions and dword loads for 50 useful lanes).  Here a 16-lane DPP row owns a
// matrix row; a lane holds 4 CONSECUTIVE entries (position p = 4 * lane16 + c), fetched as one
// 16-byte buffer load each from `indices` and `data` (reads past the end of the arrays return 0
// by the buffer bounds check); the reductions are 4-step butterflies inside the DPP row, and one
// instruction stream selects for four rows at once.
typedef unsigned int uint4q_t __attribute__((ext_vector_type(4)));
#define XC_Q4_RSRC_WORD3 0x00020000 /* raw buffer, 32-bit data format (gfx9) */
#define XC_Q4_CPOL_NT 2             /* streaming loads: do not keep the lines */

// Template flags turn the optional features into straight-line code (the kernel is instruction-bound: ~1300 issue
// cycles per four rows with every feature behind a run-time branch): WMODE 0 = no weights, 1 = interleaved (a, b) pairs,
// 2 = separate a and / or b (with per-row classifier offsets when CLS); EXTRA = eta and membership flags are written too
// (the BCA initial prediction).
template <int WMODE, bool CLS, bool EXTRA>
__global__ __launch_bounds__(XC_BLOCK) void topk_csr_q4_kernel(TopkParams<float> P) {
    const int lane = lane_id();
    const int l16 = lane & 15;
    const int shift = lane & 48; // bit offset of this DPP row in a wave ballot
    const int64_t wave = (int64_t)blockIdx.x * (XC_BLOCK / XC_WAVE) + (threadIdx.x >> 6);
    if (wave >= P.n_waves) return;
    const int64_t G = (int64_t)P.n_waves * 4;
    const int64_t grp = wave * 4 + (lane >> 4);
    const int k = P.k;
    const int64_t last = P.n - 1;
    const unsigned nnz_bytes = (unsigned)P.indptr[P.n] * 4u;
    const __amdgpu_buffer_rsrc_t r_idx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(P.indices), 0, nnz_bytes, XC_Q4_RSRC_WORD3);
    const __amdgpu_buffer_rsrc_t r_eta =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.data), 0, nnz_bytes, XC_Q4_RSRC_WORD3);
    auto clampr = [&](int64_t r) { return r < last ? r : last; };
    auto seg16 = [&](unsigned long long mask) { return (unsigned)(mask >> shift) & 0xFFFFu; };

    struct Row { uint4q_t idx, eta; };
    auto load = [&](int s, Row &d) {
        const int off = (s + 4 * l16) * 4;
        d.idx = __builtin_amdgcn_raw_buffer_load_b128(r_idx, off, 0, XC_Q4_CPOL_NT);
        d.eta = __builtin_amdgcn_raw_buffer_load_b128(r_eta, off, 0, XC_Q4_CPOL_NT);
    };
    // software pipeline per DPP row: the entries of its next matrix row and the indptr pair of the
    // one after are in flight while the current one is selected.  Groups past the end keep
    // working on (clamped) row n-1 without storing, so the wave stays convergent.
    int64_t row = grp;
    int64_t rc = clampr(row);
    int s0 = P.indptr[rc], e0 = P.indptr[rc + 1];
    rc = clampr(row + G);
    int s1 = P.indptr[rc], e1 = P.indptr[rc + 1];
    Row cur;
    load(s0, cur);
    const int64_t iters = (P.n - wave * 4 + G - 1) / G; // of the wave's first DPP row, the longest
    for (int64_t it = 0; it < iters; ++it, row += G) {
        const bool live = row < P.n;
        const int s = s0, r = e0 - s0;
        Row nxt;
        load(s1, nxt);
        rc = clampr(row + 2 * G);
        const int s2 = P.indptr[rc], e2 = P.indptr[rc + 1];
        const int64_t w_off = (CLS && live) ? (int64_t)P.row_cls[row] * P.ld : 0;

        int idx[4];
        float eta[4], gain[4];
        unsigned key[4];
        bool sel[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool valid = 4 * l16 + c < r;
            idx[c] = valid ? (int)cur.idx[c] : 0;
            eta[c] = __uint_as_float(cur.eta[c]);
            float g = eta[c];
            if (WMODE == 1) {
                typedef float pair_t __attribute__((ext_vector_type(2)));
                const pair_t w = *reinterpret_cast<const pair_t *>(P.ab + 2 * (int64_t)idx[c]);
                g = g * w.x; // numba_csr_functions.py:608-609
                g = g + w.y; // :610-611
            }
            if (WMODE == 2) {
                if (P.a) g = g * P.a[w_off + idx[c]];
                if (P.b) g = g + P.b[w_off + idx[c]];
            }
            gain[c] = g;
            // (computed in all lanes, selected afterwards: left to itself hipcc wraps each entry's key in its own
            // exec-masked block -- four saveexec / branch / restore sequences per iteration)
            unsigned kk = sortable_key32(nan_to_neg_inf(g));
            asm volatile("" : "+v"(kk));
            key[c] = valid ? kk : 0u;
            sel[c] = false;
        }
        const int want = r < k ? r : k; // :465-466: a row of at most k entries keeps them all
        // Selection = a tournament of the 16 lanes' sorted lists (the kernel is bound by these instructions, not
        // by HBM: 82 us without them at 1M x 50).  Each lane first sorts its four keys in descending order with a
        // stable network of adjacent exchanges (equal keys keep their position order; q* = entry index of a
        // slot); a round is then ONE 16-lane DPP max over the list heads, a ballot that names the first lane
        // holding it (ties go to the lower column) and a pop in that lane.
        unsigned k0 = key[0], k1 = key[1], k2 = key[2], k3 = key[3];
        unsigned q0 = 0, q1 = 1, q2 = 2, q3 = 3;
#define XC_Q4_CSWAP(ka, kb, qa, qb)                                                                                \
    {                                                                                                              \
        const bool sw = kb > ka;                                                                                   \
        const unsigned tk = sw ? kb : ka, tq = sw ? qb : qa;                                                       \
        kb = sw ? ka : kb;                                                                                         \
        qb = sw ? qa : qb;                                                                                         \
        ka = tk;                                                                                                   \
        qa = tq;                                                                                                   \
    }
        XC_Q4_CSWAP(k0, k1, q0, q1);
        XC_Q4_CSWAP(k2, k3, q2, q3);
        XC_Q4_CSWAP(k1, k2, q1, q2);
        XC_Q4_CSWAP(k0, k1, q0, q1);
        XC_Q4_CSWAP(k2, k3, q2, q3);
        XC_Q4_CSWAP(k1, k2, q1, q2);
#undef XC_Q4_CSWAP
        unsigned selmask = 0u;
#ifdef XC_EXP_TOPK_NOSELECT /* diagnostic build (tools/build_exp.sh): memory-only floor, wrong results */
        const int rounds = 0;
        selmask = l16 < want ? 1u : 0u;
#else
        // the four rows of a wave may need different numbers of rounds: run the maximum
        // (all four rows hold at least k entries in the common case: one ballot instead of a wave reduction)
        const int rounds = __ballot(r < k) == 0ull ? k : (int)wave_umax32((unsigned)want);
#endif
        const unsigned my_bit = 1u << l16;
        // A lane's wins take its list from the head, so after the rounds its selected entries are the first `cnt`
        // slots of the sorted list: the entry masks of those prefixes are formed once, not tracked per round.
        const unsigned pm1 = 1u << q0, pm2 = pm1 | (1u << q1), pm3 = pm2 | (1u << q2), pm4 = pm3 | (1u << q3);
        unsigned cnt = 0u;
        // (no `round < want` test: a row with fewer than k entries runs out of non-zero heads by itself)
        for (int round = 0; round < rounds; ++round) {
            const unsigned M = row16_umax32(k0);
            const unsigned holders = seg16(__builtin_amdgcn_ballot_w64(k0 == M && k0 != 0u));
            const bool win = (holders & (0u - holders)) == my_bit; // lowest holder of this DPP row
            cnt += win ? 1u : 0u;
            k0 = win ? k1 : k0;
            k1 = win ? k2 : k1;
            k2 = win ? k3 : k2;
            k3 = win ? 0u : k3;
        }
#ifndef XC_EXP_TOPK_NOSELECT
        selmask = cnt == 0u ? 0u : (cnt == 1u ? pm1 : (cnt == 2u ? pm2 : (cnt == 3u ? pm3 : pm4)));
#endif
#pragma unroll
        for (int c = 0; c < 4; ++c) sel[c] = (selmask >> c) & 1u;

        // ascending-column emission inside the DPP row: positions are lane-major
        int32_t *o_idx = P.out_indices + row * k;
        float *o_dat = P.out_data ? P.out_data + row * k : nullptr;
        float *o_eta = (EXTRA && P.out_eta) ? P.out_eta + row * k : nullptr;
        // selected entries held by the lower lanes of the DPP row: a prefix sum of the lanes' counts inside the row
        unsigned mine = (unsigned)__popc(selmask), upto = mine;
        upto += dpp_src<0x111, 0xF>(0u, upto); // row_shr:1
        upto += dpp_src<0x112, 0xF>(0u, upto); // row_shr:2
        upto += dpp_src<0x114, 0xF>(0u, upto); // row_shr:4
        upto += dpp_src<0x118, 0xF>(0u, upto); // row_shr:8
        int slot = (int)(upto - mine);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (sel[c] && live) {
                o_idx[slot] = idx[c];
                if (o_dat) o_dat[slot] = P.keep_scores ? gain[c] : 1.0f;
                if (o_eta) o_eta[slot] = eta[c];
            }
            slot += sel[c] ? 1 : 0;
            if (EXTRA && P.out_sel && live && 4 * l16 + c < r) P.out_sel[s + 4 * l16 + c] = sel[c] ? 1 : 0;
        }
        // :599-601: slots a short row leaves unused keep column 0 / value 1
        if (live)
            for (int q = want + l16; q < k; q += 16) {
                o_idx[q] = 0;
                if (o_dat) o_dat[q] = 1.0f;
                if (o_eta) o_eta[q] = 0.0f;
            }

        cur = nxt;
        s0 = s1; e0 = e1;
        s1 = s2; e1 = e2;
    }
}

// ---- k == 0: keep entries with gain >= th (numba_csr_functions.py:516-517) ----
template <typename T, bool FILL>
__global__ __launch_bounds__(XC_BLOCK) void threshold_csr_kernel(
    int64_t n, const int32_t *indptr, const int32_t *indices, const T *data, T th,
    const T *a, const T *b, int64_t ld, const int32_t *row_cls, int32_t *out_counts, const int32_t *out_indptr,
    int32_t *out_indices, int n_waves) {
    const int lane = lane_id();
    const int wave = blockIdx.x * (XC_BLOCK / XC_WAVE) + (threadIdx.x >> 6);
    if (wave >= n_waves) return;
    for (int64_t row = wave; row < n; row += n_waves) {
        const int s = indptr[row];
        const int r = indptr[row + 1] - s;
        const int64_t w_off = row_cls ? (int64_t)row_cls[row] * ld : 0;
        int base = 0;
        for (int p0 = 0; p0 < r; p0 += XC_WAVE) {
            const int p = p0 + lane;
            bool keep = false;
            int col = -1;
            if (p < r) {
                col = indices[s + p];
                T g = data[s + p];
                if (a) g = g * a[w_off + col];
                if (b) g = g + b[w_off + col];
                keep = g >= th;
            }
            const unsigned long long mask = __ballot(keep);
            if (FILL && keep)
                out_indices[out_indptr[row] + base + __popcll(mask & lanemask_lt())] = col;
            base += __popcll(mask);
        }
        if (!FILL && lane == 0) out_counts[row] = base;
    }
}

template <typename T>
static int launch_topk(int64_t n, const int32_t *indptr, const int32_t *indices, const void *data,
                       int k, const void *a, const void *b, int64_t ld, const int32_t *row_cls, int keep_scores,
                       int32_t *out_indices,
                       void *out_data, void *out_eta, uint8_t *out_sel, int ch, hipStream_t st) {
    TopkParams<T> P;
    P.n = n;
    P.indptr = indptr;
    P.indices = indices;
    P.data = static_cast<const T *>(data);
    P.a = static_cast<const T *>(a);
    P.b = static_cast<const T *>(b);
    P.ab = nullptr;
    if (ld < 0) { // `a` carries the interleaved table
        P.ab = P.a;
        P.a = P.b = nullptr;
        ld = 0;
    }
    P.row_cls = row_cls;
    P.ld = ld;
    P.k = k;
    P.keep_scores = keep_scores;
    P.out_indices = out_indices;
    P.out_data = static_cast<T *>(out_data);
    P.out_eta = static_cast<T *>(out_eta);
    P.out_sel = out_sel;
    P.n_waves = default_row_waves(n);
    const int blocks = (P.n_waves + 3) / 4;
    switch (ch) {
    case 1: hipLaunchKernelGGL((topk_csr_kernel<T, 1>), dim3(blocks), dim3(XC_BLOCK), 0, st, P); break;
    case 2: hipLaunchKernelGGL((topk_csr_kernel<T, 2>), dim3(blocks), dim3(XC_BLOCK), 0, st, P); break;
    case 4: hipLaunchKernelGGL((topk_csr_kernel<T, 4>), dim3(blocks), dim3(XC_BLOCK), 0, st, P); break;
    case 8: hipLaunchKernelGGL((topk_csr_kernel<T, 8>), dim3(blocks), dim3(XC_BLOCK), 0, st, P); break;
    default: hipLaunchKernelGGL((topk_csr_kernel<T, 16>), dim3(blocks), dim3(XC_BLOCK), 0, st, P); break;
    }
    return 0;
}

// XCOLUMNS_TOPK_ONE_ROW_PER_WAVE=1 keeps the one-row-per-wave kernel for short float32 rows too
// (A/B measurements)
static const bool g_topk_one_row_per_wave = [] {
    const char *e = getenv("XCOLUMNS_TOPK_ONE_ROW_PER_WAVE");
    return e && e[0] == '1';
}();

static int launch_topk_q4(int64_t n, const int32_t *indptr, const int32_t *indices, const void *data, int k,
                          const void *a, const void *b, int64_t ld, const int32_t *row_cls, int keep_scores,
                          int32_t *out_indices, void *out_data, void *out_eta, uint8_t *out_sel, hipStream_t st) {
    TopkParams<float> P;
    P.n = n;
    P.indptr = indptr;
    P.indices = indices;
    P.data = static_cast<const float *>(data);
    P.a = static_cast<const float *>(a);
    P.b = static_cast<const float *>(b);
    P.ab = nullptr;
    if (ld < 0) { // `a` carries the interleaved table
        P.ab = P.a;
        P.a = P.b = nullptr;
        ld = 0;
    }
    P.row_cls = row_cls;
    P.ld = ld;
    P.k = k;
    P.keep_scores = keep_scores;
    P.out_indices = out_indices;
    P.out_data = static_cast<float *>(out_data);
    P.out_eta = static_cast<float *>(out_eta);
    P.out_sel = out_sel;
    P.n_waves = default_row_waves((n + 3) / 4);
    const dim3 grid((P.n_waves + 3) / 4), block(XC_BLOCK);
    const int wmode = P.ab ? 1 : ((P.a || P.b) ? 2 : 0);
    const bool extra = P.out_eta || P.out_sel;
#define XC_Q4(W, C, E) hipLaunchKernelGGL((topk_csr_q4_kernel<W, C, E>), grid, block, 0, st, P)
    if (wmode == 0) {
        if (extra) XC_Q4(0, false, true);
        else XC_Q4(0, false, false);
    } else if (wmode == 1) {
        if (extra) XC_Q4(1, false, true);
        else XC_Q4(1, false, false);
    } else if (P.row_cls) {
        if (extra) XC_Q4(2, true, true);
        else XC_Q4(2, true, false);
    } else {
        if (extra) XC_Q4(2, false, true);
        else XC_Q4(2, false, false);
    }
#undef XC_Q4
    return 0;
}

} // namespace xc

extern "C" {

int xc_topk_csr(int64_t n, const int32_t *indptr, const int32_t *indices, const void *data, int dtype,
                int max_row_nnz, int k, const void *a, const void *b, int keep_scores,
                int32_t *out_indices, void *out_data, void *out_eta, uint8_t *out_sel, void *stream) {
    if (n < 0 || !indptr || (n > 0 && (!out_indices || !out_data)))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_topk_csr: NULL pointer or negative n");
    if (k < 1 || k > XC_MAX_K) return xc::fail_arg(XC_ERR_K_RANGE, "xc_topk_csr: k=%d outside 1..%d", k, XC_MAX_K);
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_topk_csr: unknown dtype %d", dtype);
    const int ch = xc::chunks_for(max_row_nnz);
    if (ch == 0)
        return xc::fail_arg(XC_ERR_ROW_TOO_LONG, "xc_topk_csr: a row holds %d entries, limit %d", max_row_nnz, XC_MAX_ROW_NNZ);
    if (n == 0) return XC_OK;
    hipStream_t st = xc::as_stream(stream);
    if (dtype == XC_F32 && max_row_nnz <= 64 && n < (1 << 24) && !xc::g_topk_one_row_per_wave)
        xc::launch_topk_q4(n, indptr, indices, data, k, a, b, 0, nullptr, keep_scores, out_indices, out_data, out_eta, out_sel, st);
    else if (dtype == XC_F32)
        xc::launch_topk<float>(n, indptr, indices, data, k, a, b, 0, nullptr, keep_scores, out_indices, out_data, out_eta, out_sel, ch, st);
    else
        xc::launch_topk<double>(n, indptr, indices, data, k, a, b, 0, nullptr, keep_scores, out_indices, out_data, out_eta, out_sel, ch, st);
    XC_CHECK_LAUNCH("topk_csr_kernel");
    return XC_OK;
}

static int threshold_common(bool fill, int64_t n, const int32_t *indptr, const int32_t *indices,
                            const void *data, int dtype, double th, const void *a, const void *b,
                            int64_t ld, const int32_t *row_cls, int32_t *out_counts, const int32_t *out_indptr, int32_t *out_indices,
                            void *stream) {
    if (n < 0 || !indptr) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_threshold_csr: NULL pointer or negative n");
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_threshold_csr: unknown dtype %d", dtype);
    if (n == 0) return XC_OK;
    hipStream_t st = xc::as_stream(stream);
    const int n_waves = xc::default_row_waves(n);
    const int blocks = (n_waves + 3) / 4;
    if (dtype == XC_F32) {
        auto d = static_cast<const float *>(data);
        auto aa = static_cast<const float *>(a);
        auto bb = static_cast<const float *>(b);
        if (fill)
            hipLaunchKernelGGL((xc::threshold_csr_kernel<float, true>), dim3(blocks), dim3(XC_BLOCK), 0, st, n, indptr, indices, d, (float)th, aa, bb, ld, row_cls, out_counts, out_indptr, out_indices, n_waves);
        else
            hipLaunchKernelGGL((xc::threshold_csr_kernel<float, false>), dim3(blocks), dim3(XC_BLOCK), 0, st, n, indptr, indices, d, (float)th, aa, bb, ld, row_cls, out_counts, out_indptr, out_indices, n_waves);
    } else {
        auto d = static_cast<const double *>(data);
        auto aa = static_cast<const double *>(a);
        auto bb = static_cast<const double *>(b);
        if (fill)
            hipLaunchKernelGGL((xc::threshold_csr_kernel<double, true>), dim3(blocks), dim3(XC_BLOCK), 0, st, n, indptr, indices, d, th, aa, bb, ld, row_cls, out_counts, out_indptr, out_indices, n_waves);
        else
            hipLaunchKernelGGL((xc::threshold_csr_kernel<double, false>), dim3(blocks), dim3(XC_BLOCK), 0, st, n, indptr, indices, d, th, aa, bb, ld, row_cls, out_counts, out_indptr, out_indices, n_waves);
    }
    XC_CHECK_LAUNCH("threshold_csr_kernel");
    return XC_OK;
}

int xc_threshold_count_csr(int64_t n, const int32_t *indptr, const int32_t *indices, const void *data,
                           int dtype, double th, const void *a, const void *b, int32_t *out_counts,
                           void *stream) {
    if (n > 0 && !out_counts) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_threshold_count_csr: out_counts is NULL");
    return threshold_common(false, n, indptr, indices, data, dtype, th, a, b, 0, nullptr, out_counts, nullptr, nullptr, stream);
}

int xc_threshold_fill_csr(int64_t n, const int32_t *indptr, const int32_t *indices, const void *data,
                          int dtype, double th, const void *a, const void *b, const int32_t *out_indptr,
                          int32_t *out_indices, void *stream) {
    if (n > 0 && !out_indptr) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_threshold_fill_csr: out_indptr is NULL");
    return threshold_common(true, n, indptr, indices, data, dtype, th, a, b, 0, nullptr, nullptr, out_indptr, out_indices, stream);
}

// ---- weights given interleaved: ab[2 col] = a[col], ab[2 col + 1] = b[col] ----
int xc_topk_csr_ab(int64_t n, const int32_t *indptr, const int32_t *indices, const void *data, int dtype,
                   int max_row_nnz, int k, const void *ab, int keep_scores, int32_t *out_indices, void *out_data,
                   void *out_eta, uint8_t *out_sel, void *stream) {
    if (n < 0 || !indptr || !ab || (n > 0 && !out_indices))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_topk_csr_ab: NULL pointer or negative n");
    if (k < 1 || k > XC_MAX_K) return xc::fail_arg(XC_ERR_K_RANGE, "xc_topk_csr_ab: k=%d outside 1..%d", k, XC_MAX_K);
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_topk_csr_ab: unknown dtype %d", dtype);
    const int ch = xc::chunks_for(max_row_nnz);
    if (ch == 0)
        return xc::fail_arg(XC_ERR_ROW_TOO_LONG, "xc_topk_csr_ab: a row holds %d entries, limit %d", max_row_nnz, XC_MAX_ROW_NNZ);
    if (n == 0) return XC_OK;
    hipStream_t st = xc::as_stream(stream);
    if (dtype == XC_F32 && max_row_nnz <= 64 && n < (1 << 24) && !xc::g_topk_one_row_per_wave)
        xc::launch_topk_q4(n, indptr, indices, data, k, ab, nullptr, -1, nullptr, keep_scores, out_indices, out_data, out_eta, out_sel, st);
    else if (dtype == XC_F32)
        xc::launch_topk<float>(n, indptr, indices, data, k, ab, nullptr, -1, nullptr, keep_scores, out_indices, out_data, out_eta, out_sel, ch, st);
    else
        xc::launch_topk<double>(n, indptr, indices, data, k, ab, nullptr, -1, nullptr, keep_scores, out_indices, out_data, out_eta, out_sel, ch, st);
    XC_CHECK_LAUNCH("topk_csr_kernel (ab)");
    return XC_OK;
}

// ---- one weighted classifier per row (frank_wolfe.py:127-172) ----
int xc_topk_csr_rowwise(int64_t n, const int32_t *indptr, const int32_t *indices, const void *data, int dtype,
                        int max_row_nnz, int k, const void *a, const void *b, int64_t ld,
                        const int32_t *row_classifier, int32_t *out_indices, void *stream) {
    if (n < 0 || !indptr || (n > 0 && (!out_indices || !a || !b || !row_classifier)) || ld < 0)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_topk_csr_rowwise: NULL pointer or bad size");
    if (k < 1 || k > XC_MAX_K) return xc::fail_arg(XC_ERR_K_RANGE, "xc_topk_csr_rowwise: k=%d outside 1..%d", k, XC_MAX_K);
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_topk_csr_rowwise: unknown dtype %d", dtype);
    const int ch = xc::chunks_for(max_row_nnz);
    if (ch == 0)
        return xc::fail_arg(XC_ERR_ROW_TOO_LONG, "xc_topk_csr_rowwise: a row holds %d entries, limit %d", max_row_nnz, XC_MAX_ROW_NNZ);
    if (n == 0) return XC_OK;
    hipStream_t st = xc::as_stream(stream);
    // only the chosen column ids are produced: the prediction's values are all ones
    if (dtype == XC_F32 && max_row_nnz <= 64 && n < (1 << 24) && !xc::g_topk_one_row_per_wave)
        xc::launch_topk_q4(n, indptr, indices, data, k, a, b, ld, row_classifier, 0, out_indices, nullptr, nullptr, nullptr, st);
    else if (dtype == XC_F32)
        xc::launch_topk<float>(n, indptr, indices, data, k, a, b, ld, row_classifier, 0, out_indices, nullptr, nullptr, nullptr, ch, st);
    else
        xc::launch_topk<double>(n, indptr, indices, data, k, a, b, ld, row_classifier, 0, out_indices, nullptr, nullptr, nullptr, ch, st);
    XC_CHECK_LAUNCH("topk_csr_kernel (rowwise)");
    return XC_OK;
}

int xc_threshold_count_csr_rowwise(int64_t n, const int32_t *indptr, const int32_t *indices, const void *data,
                                   int dtype, double th, const void *a, const void *b, int64_t ld,
                                   const int32_t *row_classifier, int32_t *out_counts, void *stream) {
    if (n > 0 && (!out_counts || !row_classifier))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_threshold_count_csr_rowwise: NULL pointer");
    return threshold_common(false, n, indptr, indices, data, dtype, th, a, b, ld, row_classifier, out_counts, nullptr, nullptr, stream);
}

int xc_threshold_fill_csr_rowwise(int64_t n, const int32_t *indptr, const int32_t *indices, const void *data,
                                  int dtype, double th, const void *a, const void *b, int64_t ld,
                                  const int32_t *row_classifier, const int32_t *out_indptr, int32_t *out_indices,
                                  void *stream) {
    if (n > 0 && (!out_indptr || !row_classifier))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_threshold_fill_csr_rowwise: NULL pointer");
    return threshold_common(true, n, indptr, indices, data, dtype, th, a, b, ld, row_classifier, nullptr, out_indptr, out_indices, stream);
}

} // extern "C"

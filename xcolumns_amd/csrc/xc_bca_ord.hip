// xc_bca_ord.hip -- the ORDERED parallel BCA sweep on CSR rows: thousands of rows in flight AND the reference's
// visiting-order semantics (/root/reference/xcolumns/block_coordinate.py:448-463: row i + 1 sees what row i wrote).
//
// The concurrent sweep of xc_bca.hip lets rows in flight miss each other's updates (bounded, measured, tuned -- but
// not the reference's sequence); its exact fallback was ONE wavefront.  This kernel is exact and wide:
//
//   The order is walked in WINDOWS of W rows, one wavefront per row, row data and the committed records of its
//   candidates held in registers.  Inside a window the rows iterate (Jacobi on the decisions):
//     iteration t   every row decides (gains + top-k, the reference's arithmetic: statistics divided by n, IEEE
//                   divisions) on  committed record + the changes that the EARLIER rows of the window decided on in
//                   iteration t - 1,  and publishes its own change list (labels it adds / drops, signed eta);
//     until no row's decision moved.  By induction over the positions the fixed point is what the sequential sweep
//   does with these rows (row 1 of the window sees only committed records: final after iteration 0; row p is final once
//   rows 1 .. p - 1 are), so the iteration ends after at most W rounds -- measured: 3-7 (tests/studies/ordered_sim.py,
//   profiles/r03_ordered_sim.txt: a row's decision rarely depends on WHICH earlier row touched a candidate).
//   Then the window commits (float64 atomics on the records, the new prediction) and the next window starts.
//
//   Change lists.  Per label a small array of {signed eta, window slot} entries, refilled every iteration: the count
//   word carries the iteration number in its high half (atomicMax installs the current iteration with count 0, the
//   returning atomicAdd hands out the index), so nothing is ever cleared; two copies (iteration parity) separate this
//   iteration's writers from the readers of the previous one.  A reader sums the entries with a smaller slot.  Labels
//   stored in many rows ("hot": a window holds ~100 readers and writers of each) would make that quadratic: they get a
//   dense [label][slot] table instead, prefix-summed by one workgroup per label between two iterations, so a reader
//   takes its correction with one 16-byte load.
//
//   One launch per sweep; windows and iterations are separated by a hand-rolled grid barrier (all workgroups are
//   resident: one 1024-thread workgroup per CU; cross-workgroup data moves by agent-scope atomics and sc1 loads /
//   stores only, so the barrier needs no cache maintenance -- MI355X_MICROARCH.md, inter-workgroup visibility).
#include "xc_common.h"
#include "xc_host.h"

namespace xc {

#define XC_ORD_BLOCK 1024
#define XC_ORD_WAVES (XC_ORD_BLOCK / XC_WAVE) /* rows per workgroup */
#define XC_ORD_MAX_HOT 255
#define XC_ORD_SCAN_E 4 /* the hot tables' scan: slots per thread, so a window holds at most 4096 rows */
#define XC_ORD_EPOCHS_PER_LAUNCH (1u << 20)
// words of the sync block (zeroed before every launch)
#define XC_ORD_BAR 0      /* barrier arrivals */
#define XC_ORD_ABORT 1    /* != 0: leave (error code) */
#define XC_ORD_MOVED 2    /* [3] rows whose decision moved, by iteration % 3 */
#define XC_ORD_OVF 5      /* a label's change list overflowed in this iteration */
#define XC_ORD_SYNC_WORDS 64
// status words (int64, device)
#define XC_ORD_ST_DONE 0  /* positions of the order committed */
#define XC_ORD_ST_ERROR 1 /* 0 ok, 1 change list overflow (the rest of the order is left to the caller), 2 barrier timeout, 3 iteration limit */
#define XC_ORD_ST_ITERS 2
#define XC_ORD_ST_WINDOWS 3

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
#define XC_ORD_RSRC_WORD3 0x00020000
#define XC_ORD_SC1 16

struct OrdEntry {
    double eta_signed; // +eta: the row adds the label, -eta: it drops it (an orphan: -0.0)
    int32_t slot;      // the row's slot in the window
    int32_t pad;
};

template <typename T>
struct OrdParams {
    int64_t n_order;
    const int32_t *order;
    const int32_t *indptr;
    const int32_t *indices;
    const T *data;
    int32_t *pred_indices;
    T *pred_eta;
    uint8_t *sel;
    const int32_t *orphans; // optional [n * k]
    int k;
    double *tpfp;          // [m][2]
    const double *s_entry; // [nnz] column sum per stored entry
    int64_t m;
    const int32_t *lab_dir; // [m][2] {offset of the label's entries | -(hot slot + 1), capacity}
    unsigned long long *cnt; // [2][m] (iteration << 32) | entries
    OrdEntry *ent;           // [2][total_cap]
    int64_t total_cap;
    double *hot_delta;       // [2][n_hot][W] signed eta of the row in slot s (0: none)
    double2_t *hot_prefix;   // [2][n_hot][W] changes of the slots before s
    double2_t *hot_total;    // [2][n_hot]
    const int32_t *hot_labels; // [n_hot]
    int n_hot;
    unsigned *sync;
    long long *status;
    xc_metric metric;
    double nn, n_counted;
    int maximize, skip_tn;
    unsigned epoch0;
    unsigned long long *changed;
};

__device__ __forceinline__ unsigned ld_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u32(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_f64(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Grid barrier: every wave has drained its stores, one lane per workgroup arrives on a monotonic counter and polls
// it.  Returns false when the launch is being abandoned (another workgroup timed out or found an error).
__device__ __forceinline__ bool ord_barrier(unsigned *sync, unsigned &target, int *s_ok) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        target += gridDim.x;
        (void)__hip_atomic_fetch_add(sync + XC_ORD_BAR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        unsigned spins = 0;
        while ((int)(ld_u32(sync + XC_ORD_BAR) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63u) == 0) {
                if (ld_u32(sync + XC_ORD_ABORT) != 0u) { ok = 0; break; }
                if (spins > (1u << 24)) { // seconds: a workgroup is not resident or died
                    st_u32(sync + XC_ORD_ABORT, 2u);
                    ok = 0;
                    break;
                }
            }
        }
        if (ok && ld_u32(sync + XC_ORD_ABORT) != 0u) ok = 0;
        *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0;
}

// top-k of the row's keys (xc_bca.hip: swap the worst member for the best outsider while strictly better, exact
// bisection with "lower position wins" otherwise)
template <int CH>
__device__ __forceinline__ void ord_select(const unsigned long long (&key)[CH], const bool (&in_cur)[CH], int n_cur, int kk,
                                           bool (&in_new)[CH]) {
#pragma unroll
    for (int c = 0; c < CH; ++c) in_new[c] = in_cur[c];
    bool exact_path = (n_cur != kk);
    if (!exact_path) {
        for (int it = 0; it <= kk; ++it) {
            unsigned long long lmin = ~0ull, lmax = 0ull;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (in_new[c]) lmin = key[c] < lmin ? key[c] : lmin;
                else lmax = key[c] > lmax ? key[c] : lmax;
            }
            const unsigned long long smin = wave_umin64(lmin), umax = wave_umax64(lmax);
            if (umax < smin) break;
            int n_min = 0, n_max = 0;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                n_min += __popcll(__ballot(in_new[c] && key[c] == smin));
                n_max += __popcll(__ballot(!in_new[c] && key[c] == umax));
            }
            if (umax == smin || n_min != 1 || n_max != 1 || it == kk) {
                exact_path = true;
                break;
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (in_new[c] && key[c] == smin) in_new[c] = false;
                else if (!in_new[c] && key[c] == umax) in_new[c] = true;
            }
        }
    }
    if (exact_path) {
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
    }
}

// one entry into the change list of a label (this iteration's copy): returns false when the list is full
__device__ __forceinline__ bool ord_insert(unsigned long long *cnt, OrdEntry *ent, int off, int cap, unsigned epoch, double eta_signed,
                                           int slot) {
    (void)__hip_atomic_fetch_max(cnt, (unsigned long long)epoch << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long old = __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned i = (unsigned)old;
    if ((int)i >= cap) return false;
    OrdEntry *e = ent + off + i;
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(&e->eta_signed), (unsigned long long)__double_as_longlong(eta_signed),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(&e->slot), (unsigned long long)(unsigned)slot, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

template <typename T>
__device__ __forceinline__ void ord_delta(double eta_signed, double &dtp, double &dfp) {
    const bool neg = __builtin_signbit(eta_signed);
    const double a = __builtin_fabs(eta_signed);
    const double om = (double)((T)1 - (T)a); // (1 - eta) in the input dtype, block_coordinate.py:253
    dtp = neg ? -a : a;
    dfp = neg ? -om : om;
}

// prefix sums of one hot label's dense change table over the window's slots (one workgroup)
template <typename T>
__device__ void ord_scan_hot(const OrdParams<T> &P, int par, int h, int W, double (*s_w)[2]) {
    double *delta = P.hot_delta + ((int64_t)par * P.n_hot + h) * W;
    double2_t *prefix = P.hot_prefix + ((int64_t)par * P.n_hot + h) * W;
    constexpr int E = XC_ORD_SCAN_E; // slots per thread: W <= E * 1024
    const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    double vt[E], vf[E];
    double st = 0.0, sf = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid * E + e;
        double d = 0.0;
        if (i < W) d = ld_f64(delta + i);
        const bool some = d != 0.0 || __builtin_signbit(d); // -0.0: an orphan leaving
        double a = 0.0, b = 0.0;
        if (some) ord_delta<T>(d, a, b);
        vt[e] = st; // exclusive within the thread
        vf[e] = sf;
        st += a;
        sf += b;
        if (some) st_f64(delta + i, 0.0);
    }
    // inclusive scan of the thread totals over the wave
    double it = st, iff = sf;
#pragma unroll
    for (int o = 1; o < XC_WAVE; o <<= 1) {
        const double ut = __shfl_up(it, o, XC_WAVE), uf = __shfl_up(iff, o, XC_WAVE);
        if (lane >= o) {
            it += ut;
            iff += uf;
        }
    }
    if (lane == XC_WAVE - 1) {
        s_w[wv][0] = it;
        s_w[wv][1] = iff;
    }
    __syncthreads();
    double bt = 0.0, bf = 0.0;
    for (int w = 0; w < wv; ++w) {
        bt += s_w[w][0];
        bf += s_w[w][1];
    }
    const double ext = bt + (it - st), exf = bf + (iff - sf); // changes of all slots before this thread's first
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid * E + e;
        if (i < W) {
            st_f64(reinterpret_cast<double *>(prefix + i), ext + vt[e]);
            st_f64(reinterpret_cast<double *>(prefix + i) + 1, exf + vf[e]);
        }
    }
    if (tid == XC_ORD_BLOCK - 1) {
        double2_t *tot = P.hot_total + (int64_t)par * P.n_hot + h;
        st_f64(reinterpret_cast<double *>(tot), bt + it);
        st_f64(reinterpret_cast<double *>(tot) + 1, bf + iff);
    }
    __syncthreads(); // s_w is reused by the next label
}

template <typename T, int CH>
__global__ __launch_bounds__(XC_ORD_BLOCK) void bca_ordered_sweep_kernel(OrdParams<T> P) {
    __shared__ int s_ok;
    __shared__ int s_moved, s_ovf;
    __shared__ double s_w[XC_ORD_WAVES][2];
    const int lane = lane_id();
    const int wib = threadIdx.x >> 6;
    const int slot = blockIdx.x * XC_ORD_WAVES + wib;
    const int W = gridDim.x * XC_ORD_WAVES;
    const int k = P.k;
    const double nn = P.nn;
    const bool skip_tn = P.skip_tn != 0;
    unsigned bar_target = 0;
    unsigned epoch = P.epoch0;
    const unsigned epoch_end = P.epoch0 + XC_ORD_EPOCHS_PER_LAUNCH - 4;
    long long iters = 0, windows = 0;
    unsigned long long n_changed = 0;
    int err = 0;
    int64_t base = 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(P.tpfp, 0, (unsigned)(P.m * 16), XC_ORD_RSRC_WORD3);
    if (threadIdx.x == 0) s_moved = s_ovf = 0;
    __syncthreads();

    for (; base < P.n_order; base += W) {
        const int64_t pos = base + slot;
        const bool active = pos < P.n_order;
        const int64_t row = active ? (P.order ? P.order[pos] : (int32_t)pos) : 0;
        const int s0 = active ? P.indptr[row] : 0;
        const int r = active ? P.indptr[row + 1] - s0 : 0;
        const int kk = r < k ? r : k;
        // ---- the row, the directory entries of its labels and their committed records: fixed for the window ----
        int idx[CH], off[CH], cap[CH];
        T eta[CH];
        double sc[CH], rtp[CH], rfp[CH];
        bool in_old[CH], in_prev[CH];
        int n_old = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int p = lane + XC_WAVE * c;
            const bool have = p < r;
            const int pc = have ? p : (r > 0 ? r - 1 : 0);
            idx[c] = active && r > 0 ? P.indices[s0 + pc] : 0;
            eta[c] = active && r > 0 ? P.data[s0 + pc] : (T)0;
            sc[c] = active && r > 0 ? P.s_entry[s0 + pc] : 0.0;
            in_old[c] = have && P.sel[s0 + pc] != 0;
            in_prev[c] = in_old[c];
            n_old += __popcll(__ballot(in_old[c]));
            const int2 d = *reinterpret_cast<const int2 *>(P.lab_dir + (int64_t)idx[c] * 2);
            off[c] = d.x;
            cap[c] = d.y;
            const double2_t rec = __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, idx[c] * 16, 0, XC_ORD_SC1));
            rtp[c] = rec.x;
            rfp[c] = rec.y;
        }
        // predicted columns the row does not store ("orphans": foreign / random initial predictions): they leave
        // the prediction at the row's visit, fp -= 1 (numba_csr_functions.py:200-203) -- a change known in advance
        int oid = -1, ooff = 0, ocap = 0;
        if (P.orphans && active && lane < k) {
            oid = P.orphans[row * k + lane];
            if (oid >= 0) {
                const int2 d = *reinterpret_cast<const int2 *>(P.lab_dir + (int64_t)oid * 2);
                ooff = d.x;
                ocap = d.y;
            }
        }
        const bool has_orphans = __ballot(oid >= 0) != 0ull;
        bool in_new[CH];
        bool row_moved_ever = false;
        int t = 0;
        bool converged = false;
        for (;; ++t) {
            const int wpar = (int)(epoch & 1u), rpar = wpar ^ 1;
            // ---- what the earlier rows of the window decided in the previous iteration ----
            double ctp[CH], cfp[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) ctp[c] = cfp[c] = 0.0;
            if (t > 0 && active) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const bool have = lane + XC_WAVE * c < r;
                    if (have && off[c] < 0) { // hot label: the scanned table
                        const int h = -off[c] - 1;
                        const double *pf = reinterpret_cast<const double *>(P.hot_prefix + ((int64_t)rpar * P.n_hot + h) * W + slot);
                        ctp[c] = ld_f64(pf);
                        cfp[c] = ld_f64(pf + 1);
                    }
                    unsigned n_e = 0;
                    if (have && off[c] >= 0) {
                        const unsigned long long cw = ld_u64(P.cnt + (int64_t)rpar * P.m + idx[c]);
                        if ((unsigned)(cw >> 32) == epoch - 1u) n_e = (unsigned)cw;
                        if ((int)n_e > cap[c]) n_e = (unsigned)cap[c];
                    }
                    const OrdEntry *e0 = P.ent + (int64_t)rpar * P.total_cap + (off[c] >= 0 ? off[c] : 0);
                    for (unsigned i = 0; __ballot(i < n_e) != 0ull; ++i) {
                        if (i < n_e) {
                            const unsigned long long w0 = ld_u64(reinterpret_cast<const unsigned long long *>(&e0[i].eta_signed));
                            const unsigned long long w1 = ld_u64(reinterpret_cast<const unsigned long long *>(&e0[i].slot));
                            if ((int)(unsigned)w1 < slot) {
                                double a, b;
                                ord_delta<T>(__longlong_as_double((long long)w0), a, b);
                                ctp[c] += a;
                                cfp[c] += b;
                            }
                        }
                    }
                }
            }
            // ---- gains (block_coordinate.py:248-282, the reference's arithmetic) and top-k ----
            unsigned long long key[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                key[c] = 0ull;
                if (lane + XC_WAVE * c < r) {
                    const T e = eta[c];
                    const T om = (T)1 - e;
                    const double ed = (double)e, omd = (double)om;
                    double tpc = rtp[c] + ctp[c], fpc = rfp[c] + cfp[c];
                    if (in_old[c]) { // statistics without this row (:243-246, in registers)
                        tpc -= ed;
                        fpc -= omd;
                    }
                    const double scc = sc[c] - ed;
                    const double fn = scc - tpc;
                    const double tn = (P.n_counted - 1.0) - fpc - scc;
                    const double pos_tp = (tpc + ed) / nn, pos_fp = (fpc + omd) / nn, neg_fn = (fn + ed) / nn;
                    const double neg_tp = tpc / nn, neg_fp = fpc / nn, pos_fn = fn / nn;
                    double pos_tn = -1.0, neg_tn = -1.0;
                    if (!skip_tn) {
                        neg_tn = (tn + omd) / nn;
                        pos_tn = tn / nn;
                    }
                    double g = metric_eval_t<true>(P.metric, pos_tp, pos_fp, pos_fn, pos_tn) -
                               metric_eval_t<true>(P.metric, neg_tp, neg_fp, neg_fn, neg_tn);
                    if (!P.maximize) g = -g;
                    key[c] = sortable_key(nan_to_neg_inf(g));
                }
            }
            ord_select<CH>(key, in_old, n_old, kk, in_new);
            bool moved = false, ovf = false;
#pragma unroll
            for (int c = 0; c < CH; ++c) moved = moved || (in_new[c] != in_prev[c]);
            moved = (__ballot(moved) != 0ull) || (t == 0 && has_orphans);
            // ---- publish this iteration's change list ----
            if (active) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if (lane + XC_WAVE * c < r && in_new[c] != in_old[c]) {
                        const double es = in_new[c] ? (double)eta[c] : -(double)eta[c];
                        if (off[c] < 0) st_f64(P.hot_delta + ((int64_t)wpar * P.n_hot + (-off[c] - 1)) * W + slot, es);
                        else if (!ord_insert(P.cnt + (int64_t)wpar * P.m + idx[c], P.ent + (int64_t)wpar * P.total_cap, off[c], cap[c], epoch, es, slot))
                            ovf = true;
                    }
                    in_prev[c] = in_new[c];
                }
                if (oid >= 0) {
                    if (ooff < 0) st_f64(P.hot_delta + ((int64_t)wpar * P.n_hot + (-ooff - 1)) * W + slot, -0.0);
                    else if (!ord_insert(P.cnt + (int64_t)wpar * P.m + oid, P.ent + (int64_t)wpar * P.total_cap, ooff, ocap, epoch, -0.0, slot))
                        ovf = true;
                }
            }
            ovf = __ballot(ovf) != 0ull;
            if (lane == 0) {
                if (moved) s_moved = 1;
                if (ovf) s_ovf = 1;
            }
            row_moved_ever = row_moved_ever || moved;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                if (s_moved) (void)__hip_atomic_fetch_add(P.sync + XC_ORD_MOVED + epoch % 3u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (s_ovf) st_u32(P.sync + XC_ORD_OVF, 1u);
                s_moved = s_ovf = 0;
                if (blockIdx.x == 0) st_u32(P.sync + XC_ORD_MOVED + (epoch + 1u) % 3u, 0u); // next iteration's counter
            }
            ++iters;
            if (!ord_barrier(P.sync, bar_target, &s_ok)) { err = 2; break; }
            const unsigned mv = ld_u32(P.sync + XC_ORD_MOVED + epoch % 3u);
            if (ld_u32(P.sync + XC_ORD_OVF) != 0u) { err = 1; break; }
            if (mv == 0u) {
                converged = true;
                break;
            }
            if (epoch >= epoch_end || t >= W + 2) { err = 3; break; }
            // ---- hot labels: prefix sums of this iteration's dense tables ----
            if (P.n_hot > 0) {
                for (int h = blockIdx.x; h < P.n_hot; h += gridDim.x) ord_scan_hot<T>(P, wpar, h, W, s_w);
                if (!ord_barrier(P.sync, bar_target, &s_ok)) { err = 2; break; }
            }
            ++epoch;
        }
        if (!converged) break;
        // ---- commit the window: the decisions of the last iteration ARE the sequential sweep's ----
        const int fpar = (int)(epoch & 1u); // this iteration's copies; the scanned totals are the previous iteration's (equal lists)
        if (active) {
            bool any = false;
#pragma unroll
            for (int c = 0; c < CH; ++c) any = any || (in_new[c] != in_old[c]);
            const bool row_changed = __ballot(any) != 0ull;
            if (row_changed || has_orphans) {
                int32_t *p_idx = P.pred_indices + row * k;
                T *p_eta = P.pred_eta + row * k;
                int o = 0;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const unsigned long long mask = __ballot(in_new[c]);
                    if (in_new[c]) {
                        const int q = o + __popcll(mask & lanemask_lt());
                        p_idx[q] = idx[c];
                        p_eta[q] = eta[c];
                    }
                    o += __popcll(mask);
                    if (lane + XC_WAVE * c < r && in_new[c] != in_old[c]) {
                        P.sel[s0 + lane + XC_WAVE * c] = in_new[c] ? 1 : 0;
                        const double sgn = in_new[c] ? 1.0 : -1.0;
                        if (off[c] < 0) {
                            st_f64(P.hot_delta + ((int64_t)fpar * P.n_hot + (-off[c] - 1)) * W + slot, 0.0); // not scanned: clear
                        } else {
                            atomic_add_f64(P.tpfp + (int64_t)idx[c] * 2, sgn * (double)eta[c]);
                            atomic_add_f64(P.tpfp + (int64_t)idx[c] * 2 + 1, sgn * (double)((T)1 - eta[c]));
                        }
                    }
                }
                if (oid >= 0) {
                    if (ooff < 0) st_f64(P.hot_delta + ((int64_t)fpar * P.n_hot + (-ooff - 1)) * W + slot, 0.0);
                    else atomic_add_f64(P.tpfp + (int64_t)oid * 2 + 1, -1.0);
                }
                if (row_changed || has_orphans) ++n_changed;
            }
        }
        if (P.n_hot > 0 && t > 0 && threadIdx.x == 0) { // hot labels: one add of the window's total per label
            for (int h = blockIdx.x; h < P.n_hot; h += gridDim.x) {
                const double *tot = reinterpret_cast<const double *>(P.hot_total + (int64_t)(fpar ^ 1) * P.n_hot + h);
                const double a = ld_f64(tot), b = ld_f64(tot + 1);
                const int64_t j = P.hot_labels[h];
                if (a != 0.0) atomic_add_f64(P.tpfp + j * 2, a);
                if (b != 0.0) atomic_add_f64(P.tpfp + j * 2 + 1, b);
            }
        }
        ++epoch;
        ++windows;
        if (!ord_barrier(P.sync, bar_target, &s_ok)) { err = 2; break; }
    }
    if (lane == 0 && n_changed && P.changed) atomicAdd(P.changed, n_changed);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned ab = ld_u32(P.sync + XC_ORD_ABORT);
        P.status[XC_ORD_ST_DONE] = base < P.n_order ? base : P.n_order;
        P.status[XC_ORD_ST_ERROR] = err ? err : (int)ab;
        P.status[XC_ORD_ST_ITERS] = iters;
        P.status[XC_ORD_ST_WINDOWS] = windows;
    }
    if (err && err != 2 && threadIdx.x == 0) st_u32(P.sync + XC_ORD_ABORT, (unsigned)err); // nobody waits any more, but be explicit
}

template <typename T>
static int ord_launch(const OrdParams<T> &P, int ch, int blocks, hipStream_t st) {
    switch (ch) {
    case 1: hipLaunchKernelGGL((bca_ordered_sweep_kernel<T, 1>), dim3(blocks), dim3(XC_ORD_BLOCK), 0, st, P); break;
    case 2: hipLaunchKernelGGL((bca_ordered_sweep_kernel<T, 2>), dim3(blocks), dim3(XC_ORD_BLOCK), 0, st, P); break;
    case 4: hipLaunchKernelGGL((bca_ordered_sweep_kernel<T, 4>), dim3(blocks), dim3(XC_ORD_BLOCK), 0, st, P); break;
    default: return fail_arg(XC_ERR_ROW_TOO_LONG, "xc_bca_ord_sweep: rows of more than 256 entries take the sequential sweep");
    }
    return XC_OK;
}

template <typename T, int CH>
static int ord_blocks_per_cu() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bca_ordered_sweep_kernel<T, CH>, XC_ORD_BLOCK, 0) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return nb;
}

} // namespace xc

extern "C" {

// Rows in flight of an ordered sweep on this device: one 1024-thread workgroup (16 rows) per CU.
int xc_bca_ord_window(int *workgroups, int *window) {
    int cu = 0;
    int rc = xc_device_info(&cu, nullptr, nullptr, 0);
    if (rc) return rc;
    if (workgroups) *workgroups = cu;
    if (window) *window = cu * XC_ORD_WAVES;
    return XC_OK;
}

int xc_bca_ord_workspace_bytes(int64_t m, int64_t total_cap, int n_hot, int workgroups, int64_t *bytes) {
    if (!bytes || m < 1 || total_cap < 0 || n_hot < 0 || n_hot > XC_ORD_MAX_HOT || workgroups < 1)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ord_workspace_bytes: bad argument");
    const int64_t W = (int64_t)workgroups * XC_ORD_WAVES;
    // sync | status | cnt[2][m] | ent[2][total_cap] | hot_delta[2][n_hot][W] | hot_prefix[2][n_hot][W] | hot_total[2][n_hot]
    *bytes = 256 + 64 + 2 * m * 8 + 2 * total_cap * 16 + 2 * (int64_t)n_hot * W * 8 + 2 * (int64_t)n_hot * W * 16 +
             2 * (int64_t)n_hot * 16 + 256;
    return XC_OK;
}

// One full sweep over `order` (NULL: rows 0 .. n_order - 1) with the reference's semantics and `workgroups` x 16 rows
// in flight.  workspace: xc_bca_ord_workspace_bytes bytes, ZEROED once by the caller (and again after an error).
// lab_dir[m][2]: per label {offset of its change list in units of entries, capacity}, or {-(h + 1), 0} for hot slot h
// (hot_labels[h] = the label); the lists of all labels are disjoint and end below total_cap.  epoch0: a number that
// grows by 2^20 from launch to launch on one workspace (the lists are tagged with it instead of being cleared).
// status_host[4] (blocks on the stream): positions committed, error (0; 1 = a change list overflowed: the positions
// from status[0] on are untouched and are the caller's to sweep), iterations, windows.
int xc_bca_ord_sweep(void *workspace, int64_t n_order, const int32_t *order, int64_t n_norm, const int32_t *indptr,
                     const int32_t *indices, const void *data, int dtype, int max_row_nnz, int32_t *pred_indices,
                     void *pred_eta, uint8_t *sel, const int32_t *orphans, int k, int64_t m, double *tpfp,
                     const double *s_entry, const int32_t *lab_dir, int64_t total_cap, const int32_t *hot_labels, int n_hot,
                     int workgroups, const xc_metric *metric_host, int maximize, int skip_tn, unsigned epoch0,
                     int64_t *changed, int64_t *status_host, void *stream) {
    if (!workspace || n_order < 0 || n_norm < 1 || m < 1 || !indptr || !indices || !data || !pred_indices || !pred_eta || !sel ||
        !tpfp || !s_entry || !lab_dir || !metric_host || !status_host || total_cap < 0 || n_hot < 0 || n_hot > XC_ORD_MAX_HOT ||
        (n_hot > 0 && !hot_labels) || workgroups < 1)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ord_sweep: bad argument");
    if (k < 1 || k > XC_MAX_K) return xc::fail_arg(XC_ERR_K_RANGE, "xc_bca_ord_sweep: k=%d outside 1..%d", k, XC_MAX_K);
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ord_sweep: unknown dtype %d", dtype);
    if (metric_host->base < 0 || metric_host->base >= XC_M_COUNT)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ord_sweep: unknown metric %d", metric_host->base);
    if (m > (int64_t)(0xFFFFFFFFu / 16)) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ord_sweep: m too large for 32-bit record offsets");
    const int ch = xc::chunks_for(max_row_nnz);
    if (ch == 0 || ch > 4)
        return xc::fail_arg(XC_ERR_ROW_TOO_LONG, "xc_bca_ord_sweep: a row holds %d entries, limit 256", max_row_nnz);
    int cu = 0;
    int rc = xc_device_info(&cu, nullptr, nullptr, 0);
    if (rc) return rc;
    // every workgroup must be resident (the grid barrier): at most what the occupancy query admits
    int per_cu = 0;
    if (dtype == XC_F32) per_cu = ch == 1 ? xc::ord_blocks_per_cu<float, 1>() : ch == 2 ? xc::ord_blocks_per_cu<float, 2>() : xc::ord_blocks_per_cu<float, 4>();
    else per_cu = ch == 1 ? xc::ord_blocks_per_cu<double, 1>() : ch == 2 ? xc::ord_blocks_per_cu<double, 2>() : xc::ord_blocks_per_cu<double, 4>();
    if (per_cu < 1) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ord_sweep: the kernel does not fit a CU");
    if (workgroups > cu || workgroups * XC_ORD_WAVES > XC_ORD_SCAN_E * XC_ORD_BLOCK)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ord_sweep: %d workgroups on %d CUs (at most %d)", workgroups, cu,
                            XC_ORD_SCAN_E * XC_ORD_BLOCK / XC_ORD_WAVES);
    status_host[0] = status_host[1] = status_host[2] = status_host[3] = 0;
    if (n_order == 0) return XC_OK;
    hipStream_t st = xc::as_stream(stream);
    char *w = static_cast<char *>(workspace);
    const int64_t W = (int64_t)workgroups * XC_ORD_WAVES;
    unsigned *sync = reinterpret_cast<unsigned *>(w);
    long long *status = reinterpret_cast<long long *>(w + 256);
    char *q = w + 320;
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(q);
    q += 2 * m * 8;
    xc::OrdEntry *ent = reinterpret_cast<xc::OrdEntry *>(q);
    q += 2 * total_cap * 16;
    double *hot_delta = reinterpret_cast<double *>(q);
    q += 2 * (int64_t)n_hot * W * 8;
    xc::double2_t *hot_prefix = reinterpret_cast<xc::double2_t *>(q);
    q += 2 * (int64_t)n_hot * W * 16;
    xc::double2_t *hot_total = reinterpret_cast<xc::double2_t *>(q);
    XC_HIP_TRY(hipMemsetAsync(w, 0, 320, st));
    if (dtype == XC_F32) {
        xc::OrdParams<float> P{n_order, order, indptr, indices, static_cast<const float *>(data), pred_indices,
                               static_cast<float *>(pred_eta), sel, orphans, k, tpfp, s_entry, m, lab_dir, cnt, ent, total_cap,
                               hot_delta, hot_prefix, hot_total, hot_labels, n_hot, sync, status, *metric_host, (double)n_norm,
                               (double)n_norm, maximize, skip_tn, epoch0, reinterpret_cast<unsigned long long *>(changed)};
        rc = xc::ord_launch(P, ch, workgroups, st);
    } else {
        xc::OrdParams<double> P{n_order, order, indptr, indices, static_cast<const double *>(data), pred_indices,
                                static_cast<double *>(pred_eta), sel, orphans, k, tpfp, s_entry, m, lab_dir, cnt, ent, total_cap,
                                hot_delta, hot_prefix, hot_total, hot_labels, n_hot, sync, status, *metric_host, (double)n_norm,
                                (double)n_norm, maximize, skip_tn, epoch0, reinterpret_cast<unsigned long long *>(changed)};
        rc = xc::ord_launch(P, ch, workgroups, st);
    }
    if (rc) return rc;
    XC_CHECK_LAUNCH("bca_ordered_sweep_kernel");
    long long tmp[4];
    XC_HIP_TRY(hipMemcpyAsync(tmp, status, sizeof(tmp), hipMemcpyDeviceToHost, st));
    XC_HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < 4; ++i) status_host[i] = tmp[i];
    return XC_OK;
}

} // extern "C"

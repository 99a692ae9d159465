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
//   Change lists.  Per label a small array of {signed eta, window slot} entries, refilled every iteration: ONE returning
//   atomicAdd on the label's count hands out the index.  Three copies rotate: iteration t fills copy t % 3, reads copy
//   (t - 1) % 3 and takes its own entries of iteration t - 2 out of the counts of copy (t - 2) % 3 again (no-return
//   atomics, nobody reads or fills that copy meanwhile), so a count is zero whenever its copy is filled.  A reader sums
//   the entries with a smaller slot (all candidates of all its rows in one loop: their loads overlap).  Labels
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

#ifndef XC_ORD_BLOCK
#define XC_ORD_BLOCK 1024 /* threads per workgroup, one workgroup per CU (experiment builds: -DXC_ORD_BLOCK=512) */
#endif
#define XC_ORD_WAVES (XC_ORD_BLOCK / XC_WAVE) /* wavefronts per workgroup */
#if XC_ORD_BLOCK <= 512
#define XC_ORD_MAX_RPW 8 /* 8 wavefronts per CU hold 256 registers each */
#else
#define XC_ORD_MAX_RPW 4
#endif
#define XC_ORD_MAX_HOT 255
#define XC_ORD_EPOCHS_PER_LAUNCH (1u << 20)
// A window settles in 3-7 iterations; one that has not after 48 is not going to (decisions on a knife edge that the order of
// a change list's float64 sum tips back and forth: ill-conditioned metrics on float64 scores, found by the exact fuzz): the
// launch ends with status 3 before that window and the caller walks the rest of the order with one wavefront.
#define XC_ORD_MAX_ITERATIONS 48
// words of the sync block (zeroed before every launch)
#define XC_ORD_BAR 0      /* barrier arrivals */
#define XC_ORD_ABORT 1    /* != 0: leave (error code) */
#define XC_ORD_MOVED 2    /* [3] rows whose decision moved, by iteration % 3 */
#define XC_ORD_OVF 5      /* a label's change list overflowed in this iteration */
#define XC_ORD_TOP 6      /* XCDs arrived (hierarchical barrier) */
#define XC_ORD_CENSUS 8   /* [8] workgroups per XCD */
#define XC_ORD_XCD_CNT 32 /* [8] x 32 words: arrivals per XCD, a 128-byte line each */
#define XC_ORD_XCD_GEN 288 /* [8] x 32 words: generation per XCD */
#define XC_ORD_SYNC_WORDS 544
#define XC_ORD_SYNC_BYTES 2304 /* the sync block (544 words), padded to a multiple of 256 bytes */
// status words (int64, device)
#define XC_ORD_ST_DONE 0  /* positions of the order committed */
#define XC_ORD_ST_ERROR 1 /* 0 ok, 1 change list overflow (the rest of the order is left to the caller), 2 barrier timeout, 3 iteration limit */
#define XC_ORD_ST_ITERS 2
#define XC_ORD_ST_WINDOWS 3
#define XC_ORD_ST_BAR_TICKS 4 /* 100 MHz ticks workgroup 0 spent in grid barriers */
#define XC_ORD_ST_ALL_TICKS 5 /* ... and in the kernel */
#define XC_ORD_ST_ROWS_PER_WAVE 6
#define XC_ORD_ST_WORDS 8

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
    unsigned *cnt;           // [3][m] entries of the label's change list, per copy
    OrdEntry *ent;           // [3][total_cap]
    int64_t total_cap;
    double *hot_delta;       // [2][n_hot][W] signed eta of the row in slot s (0: none)
    double2_t *hot_prefix;   // [2][n_hot][W] changes of the slots before s
    double2_t *hot_total;    // [2][n_hot]
    unsigned *hot_dirty;     // [2][XC_ORD_MAX_HOT] iteration in which a row last changed the hot label
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

// Grid barrier, XCD-hierarchical (MI355X_MICROARCH.md price list, barrier-xcd): every wave has drained its stores,
// one lane per workgroup arrives on its XCD's counter, the last arrival of an XCD arrives on the top counter and
// polls it, then opens its XCD's generation word that the other workgroups of the XCD poll.  All counters are
// monotonic (zeroed per launch).  `extra` (optional): two words thread 0 reads after the barrier and hands to the
// whole workgroup through LDS (the moved-rows counter and the overflow flag).  Returns false when the launch is being
// abandoned (a workgroup timed out or found an error).
struct OrdBar {
    unsigned round;   // hierarchical barriers passed (the first barrier of a launch is flat and not counted)
    unsigned xcc;     // this workgroup's XCD
    unsigned n_here;  // workgroups on this XCD
    unsigned n_xcd;   // XCDs that hold workgroups
};
#define XC_ORD_SPIN_LIMIT (1u << 24)
__device__ __forceinline__ bool ord_spin(const unsigned *word, unsigned want, unsigned *sync) {
    unsigned spins = 0;
    while ((int)(ld_u32(word) - want) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0) {
            if (ld_u32(sync + XC_ORD_ABORT) != 0u) return false;
            if (spins > XC_ORD_SPIN_LIMIT) { // seconds: a workgroup is not resident or died
                st_u32(sync + XC_ORD_ABORT, 2u);
                return false;
            }
        }
    }
    return true;
}

__device__ __forceinline__ bool ord_barrier(unsigned *sync, OrdBar &b, int *s_ok, const unsigned *extra0 = nullptr,
                                            const unsigned *extra1 = nullptr) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        bool ok = true;
        if (b.n_xcd == 0) { // the first barrier of a launch: flat (nobody knows the census yet)
            (void)__hip_atomic_fetch_add(sync + XC_ORD_BAR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ord_spin(sync + XC_ORD_BAR, gridDim.x, sync);
        } else {
            ++b.round;
            unsigned *cnt = sync + XC_ORD_XCD_CNT + 32 * b.xcc, *gen = sync + XC_ORD_XCD_GEN + 32 * b.xcc;
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1u == b.round * b.n_here) { // the last workgroup of this XCD
                (void)__hip_atomic_fetch_add(sync + XC_ORD_TOP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ord_spin(sync + XC_ORD_TOP, b.round * b.n_xcd, sync);
                st_u32(gen, b.round);
            } else {
                ok = ord_spin(gen, b.round, sync);
            }
        }
        if (ok && ld_u32(sync + XC_ORD_ABORT) != 0u) ok = false;
        s_ok[0] = ok ? 1 : 0;
        s_ok[1] = extra0 ? (int)ld_u32(extra0) : 0;
        s_ok[2] = extra1 ? (int)ld_u32(extra1) : 0;
    }
    __syncthreads();
    return s_ok[0] != 0;
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

template <typename T>
__device__ __forceinline__ void ord_delta(double eta_signed, double &dtp, double &dfp) {
    const bool neg = __builtin_signbit(eta_signed);
    const double a = __builtin_fabs(eta_signed);
    const double om = (double)((T)1 - (T)a); // (1 - eta) in the input dtype, block_coordinate.py:253
    dtp = neg ? -a : a;
    dfp = neg ? -om : om;
}

// prefix sums of one hot label's dense change table over the window's slots (one workgroup; two passes over the
// table instead of per-thread arrays).  A label no row touched in this iteration is skipped: readers check the
// same word (hot_dirty) before they use its prefix.
template <typename T>
__device__ void ord_scan_hot(const OrdParams<T> &P, int par, int h, int W, unsigned epoch, double (*s_w)[2]) {
    if (ld_u32(P.hot_dirty + par * XC_ORD_MAX_HOT + h) != epoch) return; // workgroup-uniform
    double *delta = P.hot_delta + ((int64_t)par * P.n_hot + h) * W;
    double2_t *prefix = P.hot_prefix + ((int64_t)par * P.n_hot + h) * W;
    const int E = (W + XC_ORD_BLOCK - 1) / XC_ORD_BLOCK;
    const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    double st = 0.0, sf = 0.0;
    for (int e = 0; e < E; ++e) {
        const int i = tid * E + e;
        if (i < W) {
            const double d = ld_f64(delta + i);
            if (d != 0.0 || __builtin_signbit(d)) { // -0.0: an orphan leaving
                double a, b;
                ord_delta<T>(d, a, b);
                st += a;
                sf += b;
            }
        }
    }
    // inclusive scan of the thread totals over the wave, then over the workgroup's waves
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
    double rt = bt + (it - st), rf = bf + (iff - sf); // changes of all slots before this thread's first
    for (int e = 0; e < E; ++e) {
        const int i = tid * E + e;
        if (i < W) {
            st_f64(reinterpret_cast<double *>(prefix + i), rt);
            st_f64(reinterpret_cast<double *>(prefix + i) + 1, rf);
            const double d = ld_f64(delta + i);
            if (d != 0.0 || __builtin_signbit(d)) {
                double a, b;
                ord_delta<T>(d, a, b);
                rt += a;
                rf += b;
                st_f64(delta + i, 0.0);
            }
        }
    }
    if (tid == XC_ORD_BLOCK - 1) {
        double2_t *tot = P.hot_total + (int64_t)par * P.n_hot + h;
        st_f64(reinterpret_cast<double *>(tot), bt + it);
        st_f64(reinterpret_cast<double *>(tot) + 1, bf + iff);
    }
    __syncthreads(); // s_w is reused by the next label
}

// R rows per wavefront (W = workgroups x 16 x R rows per window), CH candidates per lane and row; ORPH: the
// prediction holds labels their rows do not store (first sweep of a foreign / random initial prediction).
template <typename T, int R, int CH, bool ORPH>
__global__ __launch_bounds__(XC_ORD_BLOCK) void bca_ordered_sweep_kernel(OrdParams<T> P) {
    __shared__ int s_ok[4];
    __shared__ int s_moved, s_ovf;
    __shared__ double s_w[XC_ORD_WAVES][2];
    const int lane = lane_id();
    const int wib = threadIdx.x >> 6;
    const int waves = gridDim.x * XC_ORD_WAVES;
    const int wslot = blockIdx.x * XC_ORD_WAVES + wib; // row q of this wave sits in window slot q * waves + wslot
    const int W = waves * R;
    const int k = P.k;
    const double nn = P.nn;
    const bool skip_tn = P.skip_tn != 0;
    unsigned epoch = P.epoch0;
    const unsigned epoch_end = P.epoch0 + XC_ORD_EPOCHS_PER_LAUNCH - 4;
    long long iters = 0, windows = 0;
    unsigned long long n_changed = 0;
    int err = 0;
    int64_t base = 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(P.tpfp, 0, (unsigned)(P.m * 16), XC_ORD_RSRC_WORD3);
    const __amdgpu_buffer_rsrc_t rsrc_ent = __builtin_amdgcn_make_buffer_rsrc(P.ent, 0, (unsigned)(3 * P.total_cap * 16), XC_ORD_RSRC_WORD3);
    // ---- census: which XCD is this workgroup on, how many share it (the hierarchical barrier's counts) ----
    OrdBar bar{0u, 0u, 0u, 0u};
    unsigned long long t_bar = 0, t_begin = 0;
    if (threadIdx.x == 0) {
        s_moved = s_ovf = 0;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        bar.xcc = xcc & 7u;
        (void)__hip_atomic_fetch_add(P.sync + XC_ORD_CENSUS + bar.xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t_begin = __builtin_amdgcn_s_memrealtime();
    }
    if (!ord_barrier(P.sync, bar, s_ok)) return; // nothing has been touched yet
    if (threadIdx.x == 0) {
        unsigned nx = 0;
        for (int x = 0; x < 8; ++x) nx += ld_u32(P.sync + XC_ORD_CENSUS + x) != 0u;
        bar.n_here = ld_u32(P.sync + XC_ORD_CENSUS + bar.xcc);
        bar.n_xcd = nx;
    }

    // the next window's row ids and bounds, fetched while this window iterates
    int64_t nrow[R];
    int ns0[R], ne0[R];
    bool next_rows = false, next_ptrs = false;
    for (; base < P.n_order; base += W) {
        // ---- the window's rows, the directory entries of their labels, their committed records: fixed for the window ----
        bool active[R];
        int64_t row[R];
        int s0[R], r[R], kk[R], n_old[R];
        int idx[R][CH], off[R][CH];
        T eta[R][CH];
        double sc[R][CH], rtp[R][CH], rfp[R][CH];
        unsigned f_old = 0u, f_prev = 0u; // bit q * CH + c: the candidate is in the row's prediction (before / last iteration)
        int oid[R], ooff[R];
        unsigned orph_rows = 0u, had_corr = 0u; // bit q (wave-uniform)
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int64_t pos = base + (int64_t)q * waves + wslot;
            active[q] = pos < P.n_order;
            row[q] = next_rows ? (nrow[q] < 0 ? 0 : nrow[q]) : (active[q] ? (P.order ? P.order[pos] : (int32_t)pos) : 0);
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            s0[q] = next_ptrs ? ns0[q] : (active[q] ? P.indptr[row[q]] : 0);
            r[q] = (next_ptrs ? ne0[q] : (active[q] ? P.indptr[row[q] + 1] : 0)) - s0[q];
            if (!active[q]) r[q] = 0;
            kk[q] = r[q] < k ? r[q] : k;
        }
        next_rows = next_ptrs = false;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            n_old[q] = 0;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int p = lane + XC_WAVE * c;
                const bool have = p < r[q];
                const int pc = have ? p : (r[q] > 0 ? r[q] - 1 : 0);
                const bool any = active[q] && r[q] > 0;
                idx[q][c] = any ? P.indices[s0[q] + pc] : 0;
                eta[q][c] = any ? P.data[s0[q] + pc] : (T)0;
                sc[q][c] = any ? P.s_entry[s0[q] + pc] : 0.0;
                const bool in = have && P.sel[s0[q] + pc] != 0;
                if (in) f_old |= 1u << (q * CH + c);
                n_old[q] += __popcll(__ballot(in));
            }
            oid[q] = -1;
            ooff[q] = 0;
            if (ORPH && active[q] && lane < k) oid[q] = P.orphans[row[q] * k + lane];
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                off[q][c] = P.lab_dir[(int64_t)idx[q][c] * 2];
                const double2_t rec = __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, idx[q][c] * 16, 0, XC_ORD_SC1));
                rtp[q][c] = rec.x;
                rfp[q][c] = rec.y;
            }
            // predicted columns the row does not store ("orphans"): they leave the prediction at the row's visit,
            // fp -= 1 (numba_csr_functions.py:200-203) -- a change known before the row is scored
            if (ORPH) {
                if (oid[q] >= 0) ooff[q] = P.lab_dir[(int64_t)oid[q] * 2];
                if (__ballot(oid[q] >= 0) != 0ull) orph_rows |= 1u << q;
            }
        }
        f_prev = f_old;
        unsigned f_new = f_old;
        // candidates (bit q * CH + c) / orphans (bit q) this lane entered into the list copy filled in the previous
        // iteration (_r: being read now) and the one before (_c: to be taken out of the counts now)
        unsigned ins_r = 0u, ins_c = 0u, oins_r = 0u, oins_c = 0u;
        int t = 0;
        bool converged = false;
        for (;; ++t) {
            const unsigned wbuf = epoch % 3u, rbuf = (epoch + 2u) % 3u, cbuf = (epoch + 1u) % 3u;
            const int wpar = (int)(epoch & 1u), rpar = wpar ^ 1; // the hot tables keep two copies (the scan clears them)
            bool wg_moved = false, ovf = false;
            // ---- what the earlier rows of the window decided in the previous iteration: counts first (all rows) ----
            unsigned n_e[R][CH];
            unsigned n_max = 0u;
            double ctp[R][CH], cfp[R][CH];
#pragma unroll
            for (int q = 0; q < R; ++q)
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    n_e[q][c] = 0u;
                    ctp[q][c] = cfp[q][c] = 0.0;
                    if (t > 0 && active[q] && lane + XC_WAVE * c < r[q] && off[q][c] >= 0)
                        n_e[q][c] = ld_u32(P.cnt + (int64_t)rbuf * P.m + idx[q][c]);
                }
            // ... the next window's rows meanwhile (nothing of this depends on the statistics)
            if (t == 0) {
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const int64_t pos = base + W + (int64_t)q * waves + wslot;
                    nrow[q] = pos < P.n_order ? (P.order ? P.order[pos] : (int32_t)pos) : -1;
                }
                next_rows = true;
            } else if (t == 1) {
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    ns0[q] = nrow[q] >= 0 ? P.indptr[nrow[q]] : 0;
                    ne0[q] = nrow[q] >= 0 ? P.indptr[nrow[q] + 1] : 0;
                }
                next_ptrs = true;
            }
            if (t > 0) {
                // take this lane's entries of iteration t - 2 out of their counts (copy cbuf is idle in this iteration)
#pragma unroll
                for (int q = 0; q < R; ++q) {
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        if ((ins_c >> (q * CH + c)) & 1u)
                            (void)__hip_atomic_fetch_sub(P.cnt + (int64_t)cbuf * P.m + idx[q][c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (ORPH && ((oins_c >> q) & 1u))
                        (void)__hip_atomic_fetch_sub(P.cnt + (int64_t)cbuf * P.m + oid[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int q = 0; q < R; ++q)
#pragma unroll
                    for (int c = 0; c < CH; ++c) n_max = n_e[q][c] > n_max ? n_e[q][c] : n_max;
                // one loop over the entry index for all candidates of all rows: their loads are in flight together
                const unsigned ebase = rbuf * (unsigned)P.total_cap;
                for (unsigned i = 0; __ballot(i < n_max) != 0ull; i += 2) {
                    uint4_t e[R][CH][2];
#pragma unroll
                    for (int q = 0; q < R; ++q)
#pragma unroll
                        for (int c = 0; c < CH; ++c)
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                // past the list's end: an offset beyond the buffer (no request, zeros)
                                const unsigned o = (i + u < n_e[q][c]) ? (ebase + (unsigned)off[q][c] + i + u) * 16u : 0xFFFFFFF0u;
                                e[q][c][u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_ent, (int)o, 0, XC_ORD_SC1);
                            }
#pragma unroll
                    for (int q = 0; q < R; ++q)
#pragma unroll
                        for (int c = 0; c < CH; ++c)
#pragma unroll
                            for (int u = 0; u < 2; ++u)
                                if (i + u < n_e[q][c] && (int)e[q][c][u].z < q * waves + wslot) {
                                    double a, b;
                                    ord_delta<T>(__longlong_as_double((long long)(((unsigned long long)e[q][c][u].y << 32) | e[q][c][u].x)), a, b);
                                    ctp[q][c] += a;
                                    cfp[q][c] += b;
                                }
                }
                // hot labels: the scanned table, if any row touched the label
#pragma unroll
                for (int q = 0; q < R; ++q)
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        if (active[q] && lane + XC_WAVE * c < r[q] && off[q][c] < 0) {
                            const int h = -off[q][c] - 1;
                            if (ld_u32(P.hot_dirty + rpar * XC_ORD_MAX_HOT + h) == epoch - 1u) {
                                const double *pf = reinterpret_cast<const double *>(P.hot_prefix + ((int64_t)rpar * P.n_hot + h) * W + q * waves + wslot);
                                ctp[q][c] = ld_f64(pf);
                                cfp[q][c] = ld_f64(pf + 1);
                            }
                        }
            }
            unsigned ins_w = 0u, oins_w = 0u;
            unsigned got[R][CH]; // index the label's list handed out (this iteration's copy)
            int ocap_got[R];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                bool corr = false;
#pragma unroll
                for (int c = 0; c < CH; ++c) corr = corr || ctp[q][c] != 0.0 || cfp[q][c] != 0.0;
                corr = __ballot(corr) != 0ull;
                // a row whose records are what they were in the previous iteration (no changes then, none now) keeps
                // its decision; everybody decides in iteration 0
                const bool decide = active[q] && (t == 0 || corr || ((had_corr >> q) & 1u));
                had_corr = corr ? (had_corr | (1u << q)) : (had_corr & ~(1u << q));
                bool in_new[CH], in_old[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    in_old[c] = (f_old >> (q * CH + c)) & 1u;
                    in_new[c] = (f_new >> (q * CH + c)) & 1u;
                }
                if (decide) {
                    // ---- gains (block_coordinate.py:248-282, the reference's arithmetic) and top-k ----
                    unsigned long long key[CH];
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        key[c] = 0ull;
                        if (lane + XC_WAVE * c < r[q]) {
                            const T e = eta[q][c];
                            const T om = (T)1 - e;
                            const double ed = (double)e, omd = (double)om;
                            double tpc = rtp[q][c] + ctp[q][c], fpc = rfp[q][c] + cfp[q][c];
                            if (in_old[c]) { // statistics without this row (:243-246, in registers)
                                tpc -= ed;
                                fpc -= omd;
                            }
                            const double scc = sc[q][c] - ed;
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
                    ord_select<CH>(key, in_old, n_old[q], kk[q], in_new);
                }
                bool moved = false;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    moved = moved || (in_new[c] != (bool)((f_prev >> (q * CH + c)) & 1u));
                    f_new = in_new[c] ? (f_new | (1u << (q * CH + c))) : (f_new & ~(1u << (q * CH + c)));
                }
                moved = (__ballot(moved) != 0ull) || (ORPH && t == 0 && ((orph_rows >> q) & 1u));
                wg_moved = wg_moved || moved;
                // ---- publish this iteration's change list: the returning adds of all rows first ... ----
                const int slot = q * waves + wslot;
                ocap_got[q] = -1;
                if (active[q]) {
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        got[q][c] = 0u;
                        if (lane + XC_WAVE * c < r[q] && in_new[c] != in_old[c]) {
                            if (off[q][c] < 0) {
                                const int h = -off[q][c] - 1;
                                st_f64(P.hot_delta + ((int64_t)wpar * P.n_hot + h) * W + slot, in_new[c] ? (double)eta[q][c] : -(double)eta[q][c]);
                                if (ld_u32(P.hot_dirty + wpar * XC_ORD_MAX_HOT + h) != epoch) st_u32(P.hot_dirty + wpar * XC_ORD_MAX_HOT + h, epoch);
                            } else {
                                got[q][c] = __hip_atomic_fetch_add(P.cnt + (int64_t)wbuf * P.m + idx[q][c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                ins_w |= 1u << (q * CH + c);
                            }
                        }
                    }
                    if (ORPH && oid[q] >= 0) {
                        if (ooff[q] < 0) {
                            const int h = -ooff[q] - 1;
                            st_f64(P.hot_delta + ((int64_t)wpar * P.n_hot + h) * W + slot, -0.0);
                            if (ld_u32(P.hot_dirty + wpar * XC_ORD_MAX_HOT + h) != epoch) st_u32(P.hot_dirty + wpar * XC_ORD_MAX_HOT + h, epoch);
                        } else {
                            ocap_got[q] = (int)__hip_atomic_fetch_add(P.cnt + (int64_t)wbuf * P.m + oid[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            oins_w |= 1u << q;
                        }
                    }
                }
            }
            // ... then the entries
            {
                OrdEntry *ew = P.ent + (int64_t)wbuf * P.total_cap;
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const int slot = q * waves + wslot;
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        if ((ins_w >> (q * CH + c)) & 1u) {
                            const int cap = P.lab_dir[(int64_t)idx[q][c] * 2 + 1];
                            if ((int)got[q][c] >= cap) ovf = true;
                            else {
                                const bool inn = (f_new >> (q * CH + c)) & 1u;
                                OrdEntry *e = ew + off[q][c] + got[q][c];
                                st_f64(&e->eta_signed, inn ? (double)eta[q][c] : -(double)eta[q][c]);
                                __hip_atomic_store(reinterpret_cast<unsigned long long *>(&e->slot), (unsigned long long)(unsigned)slot, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    if (ORPH && ((oins_w >> q) & 1u)) {
                        const int cap = P.lab_dir[(int64_t)oid[q] * 2 + 1];
                        if (ocap_got[q] >= cap) ovf = true;
                        else {
                            OrdEntry *e = ew + ooff[q] + ocap_got[q];
                            st_f64(&e->eta_signed, -0.0);
                            __hip_atomic_store(reinterpret_cast<unsigned long long *>(&e->slot), (unsigned long long)(unsigned)slot, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
            ins_c = ins_r;
            ins_r = ins_w;
            oins_c = oins_r;
            oins_r = oins_w;
            f_prev = f_new;
            ovf = __ballot(ovf) != 0ull;
            if (lane == 0) {
                if (wg_moved) s_moved = 1;
                if (ovf) s_ovf = 1;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                if (s_moved) (void)__hip_atomic_fetch_add(P.sync + XC_ORD_MOVED + epoch % 3u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (s_ovf) st_u32(P.sync + XC_ORD_OVF, 1u);
                s_moved = s_ovf = 0;
                if (blockIdx.x == 0) st_u32(P.sync + XC_ORD_MOVED + (epoch + 1u) % 3u, 0u); // next iteration's counter
            }
            ++iters;
            unsigned long long tb = 0;
            if (threadIdx.x == 0 && blockIdx.x == 0) tb = __builtin_amdgcn_s_memrealtime();
            if (!ord_barrier(P.sync, bar, s_ok, P.sync + XC_ORD_MOVED + epoch % 3u, P.sync + XC_ORD_OVF)) { err = 2; break; }
            if (threadIdx.x == 0 && blockIdx.x == 0) t_bar += __builtin_amdgcn_s_memrealtime() - tb;
            const unsigned mv = (unsigned)s_ok[1];
            if (s_ok[2] != 0) { err = 1; break; }
            if (mv == 0u) {
                converged = true;
                break;
            }
            if (epoch >= epoch_end || t >= W + 2 || t >= XC_ORD_MAX_ITERATIONS) { err = 3; break; }
            // ---- hot labels: prefix sums of this iteration's dense tables ----
            if (P.n_hot > 0) {
                for (int h = blockIdx.x; h < P.n_hot; h += gridDim.x) ord_scan_hot<T>(P, wpar, h, W, epoch, s_w);
                if (threadIdx.x == 0 && blockIdx.x == 0) tb = __builtin_amdgcn_s_memrealtime();
                if (!ord_barrier(P.sync, bar, s_ok)) { err = 2; break; }
                if (threadIdx.x == 0 && blockIdx.x == 0) t_bar += __builtin_amdgcn_s_memrealtime() - tb;
            }
            ++epoch;
        }
        if (!converged) break;
        // ---- commit the window: the decisions of the last iteration ARE the sequential sweep's ----
        const int fpar = (int)(epoch & 1u); // this iteration's copies; the scanned totals are the previous iteration's (equal lists)
#pragma unroll
        for (int q = 0; q < R; ++q) {
            if (!active[q]) continue;
            const int slot = q * waves + wslot;
            // this lane's entries of the last two iterations leave the counts: every copy is empty for the next window
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if ((ins_r >> (q * CH + c)) & 1u)
                    (void)__hip_atomic_fetch_sub(P.cnt + (int64_t)(epoch % 3u) * P.m + idx[q][c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((ins_c >> (q * CH + c)) & 1u)
                    (void)__hip_atomic_fetch_sub(P.cnt + (int64_t)((epoch + 2u) % 3u) * P.m + idx[q][c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (ORPH && ((oins_r >> q) & 1u))
                (void)__hip_atomic_fetch_sub(P.cnt + (int64_t)(epoch % 3u) * P.m + oid[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ORPH && ((oins_c >> q) & 1u))
                (void)__hip_atomic_fetch_sub(P.cnt + (int64_t)((epoch + 2u) % 3u) * P.m + oid[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned diff = (f_new ^ f_old) >> (q * CH) & ((1u << CH) - 1u);
            const bool row_changed = __ballot(diff != 0u) != 0ull;
            const bool orph = ORPH && ((orph_rows >> q) & 1u);
            if (!(row_changed || orph)) continue;
            int32_t *p_idx = P.pred_indices + row[q] * k;
            T *p_eta = P.pred_eta + row[q] * k;
            int o = 0;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const bool inn = (f_new >> (q * CH + c)) & 1u, ino = (f_old >> (q * CH + c)) & 1u;
                const unsigned long long mask = __ballot(inn);
                if (inn) {
                    const int qq = o + __popcll(mask & lanemask_lt());
                    p_idx[qq] = idx[q][c];
                    p_eta[qq] = eta[q][c];
                }
                o += __popcll(mask);
                if (lane + XC_WAVE * c < r[q] && inn != ino) {
                    P.sel[s0[q] + lane + XC_WAVE * c] = inn ? 1 : 0;
                    const double sgn = inn ? 1.0 : -1.0;
                    if (off[q][c] < 0) {
                        st_f64(P.hot_delta + ((int64_t)fpar * P.n_hot + (-off[q][c] - 1)) * W + slot, 0.0); // not scanned: clear
                    } else {
                        atomic_add_f64(P.tpfp + (int64_t)idx[q][c] * 2, sgn * (double)eta[q][c]);
                        atomic_add_f64(P.tpfp + (int64_t)idx[q][c] * 2 + 1, sgn * (double)((T)1 - eta[q][c]));
                    }
                }
            }
            if (ORPH && oid[q] >= 0) {
                if (ooff[q] < 0) st_f64(P.hot_delta + ((int64_t)fpar * P.n_hot + (-ooff[q] - 1)) * W + slot, 0.0);
                else atomic_add_f64(P.tpfp + (int64_t)oid[q] * 2 + 1, -1.0);
            }
            ++n_changed;
        }
        if (P.n_hot > 0 && t > 0 && threadIdx.x == 0) { // hot labels: one add of the window's total per label
            for (int h = blockIdx.x; h < P.n_hot; h += gridDim.x) {
                if (ld_u32(P.hot_dirty + (fpar ^ 1) * XC_ORD_MAX_HOT + h) != epoch - 1u) continue;
                const double *tot = reinterpret_cast<const double *>(P.hot_total + (int64_t)(fpar ^ 1) * P.n_hot + h);
                const double a = ld_f64(tot), b = ld_f64(tot + 1);
                const int64_t j = P.hot_labels[h];
                if (a != 0.0) atomic_add_f64(P.tpfp + j * 2, a);
                if (b != 0.0) atomic_add_f64(P.tpfp + j * 2 + 1, b);
            }
        }
        ++epoch;
        ++windows;
        unsigned long long tb = 0;
        if (threadIdx.x == 0 && blockIdx.x == 0) tb = __builtin_amdgcn_s_memrealtime();
        if (!ord_barrier(P.sync, bar, s_ok)) { err = 2; break; }
        if (threadIdx.x == 0 && blockIdx.x == 0) t_bar += __builtin_amdgcn_s_memrealtime() - tb;
    }
    if (lane == 0 && n_changed && P.changed) atomicAdd(P.changed, n_changed);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned ab = ld_u32(P.sync + XC_ORD_ABORT);
        P.status[XC_ORD_ST_DONE] = base < P.n_order ? base : P.n_order;
        P.status[XC_ORD_ST_ERROR] = err ? err : (int)ab;
        P.status[XC_ORD_ST_ITERS] = iters;
        P.status[XC_ORD_ST_WINDOWS] = windows;
        P.status[XC_ORD_ST_BAR_TICKS] = (long long)t_bar;                                        // 100 MHz ticks inside barriers (workgroup 0)
        P.status[XC_ORD_ST_ALL_TICKS] = (long long)(__builtin_amdgcn_s_memrealtime() - t_begin); // ... and in the whole kernel
    }
}

template <typename T, int R, int CH>
static void ord_launch_one(const OrdParams<T> &P, int blocks, hipStream_t st) {
    if (P.orphans) hipLaunchKernelGGL((bca_ordered_sweep_kernel<T, R, CH, true>), dim3(blocks), dim3(XC_ORD_BLOCK), 0, st, P);
    else hipLaunchKernelGGL((bca_ordered_sweep_kernel<T, R, CH, false>), dim3(blocks), dim3(XC_ORD_BLOCK), 0, st, P);
}

// rows per wavefront x candidates per lane <= XC_ORD_MAX_RPW (a row's entries and records stay in registers for the window)
template <typename T>
static int ord_launch(const OrdParams<T> &P, int ch, int rows_per_wave, int blocks, hipStream_t st) {
#if XC_ORD_MAX_RPW >= 8
    if (ch == 1 && rows_per_wave >= 8) ord_launch_one<T, 8, 1>(P, blocks, st);
    else if (ch == 2 && rows_per_wave >= 4) ord_launch_one<T, 4, 2>(P, blocks, st);
    else if (ch == 4 && rows_per_wave >= 2) ord_launch_one<T, 2, 4>(P, blocks, st);
    else
#endif
    if (ch == 1 && rows_per_wave >= 4) ord_launch_one<T, 4, 1>(P, blocks, st);
    else if (ch == 1 && rows_per_wave >= 2) ord_launch_one<T, 2, 1>(P, blocks, st);
    else if (ch == 1) ord_launch_one<T, 1, 1>(P, blocks, st);
    else if (ch == 2 && rows_per_wave >= 2) ord_launch_one<T, 2, 2>(P, blocks, st);
    else if (ch == 2) ord_launch_one<T, 1, 2>(P, blocks, st);
    else if (ch == 4) ord_launch_one<T, 1, 4>(P, blocks, st);
    else return fail_arg(XC_ERR_ROW_TOO_LONG, "xc_bca_ord_sweep: rows of more than 256 entries take the sequential sweep");
    return XC_OK;
}

} // namespace xc

extern "C" {

// Rows in flight of an ordered sweep on this device: one 1024-thread workgroup per CU, 16 wavefronts each.
int xc_bca_ord_window(int *workgroups, int *window) {
    int cu = 0;
    int rc = xc_device_info(&cu, nullptr, nullptr, 0);
    if (rc) return rc;
    if (workgroups) *workgroups = cu;
    if (window) *window = cu * XC_ORD_WAVES;
    return XC_OK;
}

static int64_t ord_align(int64_t b) { return (b + 255) / 256 * 256; }

int xc_bca_ord_workspace_bytes(int64_t m, int64_t total_cap, int n_hot, int window, int64_t *bytes) {
    if (!bytes || m < 1 || total_cap < 0 || n_hot < 0 || n_hot > XC_ORD_MAX_HOT || window < 1)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ord_workspace_bytes: bad argument");
    const int64_t W = window;
    // sync | status | hot_dirty[2][255] | cnt[3][m] | ent[3][total_cap] | hot_delta[2][n_hot][W] | hot_prefix[2][n_hot][W] | hot_total[2][n_hot]
    *bytes = XC_ORD_SYNC_BYTES + 256 + ord_align(2 * XC_ORD_MAX_HOT * 4) + ord_align(3 * m * 4) + ord_align(3 * total_cap * 16) +
             ord_align(2 * (int64_t)n_hot * W * 8) + ord_align(2 * (int64_t)n_hot * W * 16) + ord_align(2 * (int64_t)n_hot * 16) + 256;
    return XC_OK;
}

int xc_bca_ord_sweep(void *workspace, int64_t n_order, const int32_t *order, int64_t n_norm, const int32_t *indptr,
                     const int32_t *indices, const void *data, int dtype, int max_row_nnz, int32_t *pred_indices,
                     void *pred_eta, uint8_t *sel, const int32_t *orphans, int k, int64_t m, double *tpfp,
                     const double *s_entry, const int32_t *lab_dir, int64_t total_cap, const int32_t *hot_labels, int n_hot,
                     int workgroups, int rows_per_wave, const xc_metric *metric_host, int maximize, int skip_tn,
                     unsigned epoch0, int64_t *changed, int64_t *status_host, void *stream) {
    if (!workspace || n_order < 0 || n_norm < 1 || m < 1 || !indptr || !indices || !data || !pred_indices || !pred_eta || !sel ||
        !tpfp || !s_entry || !lab_dir || !metric_host || !status_host || total_cap < 0 || n_hot < 0 || n_hot > XC_ORD_MAX_HOT ||
        (n_hot > 0 && !hot_labels) || workgroups < 1 || rows_per_wave < 1)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ord_sweep: bad argument");
    if (k < 1 || k > XC_MAX_K) return xc::fail_arg(XC_ERR_K_RANGE, "xc_bca_ord_sweep: k=%d outside 1..%d", k, XC_MAX_K);
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ord_sweep: unknown dtype %d", dtype);
    if (metric_host->base < 0 || metric_host->base >= XC_M_COUNT)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ord_sweep: unknown metric %d", metric_host->base);
    if (m > (int64_t)(0xFFFFFFFFu / 16)) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ord_sweep: m too large for 32-bit record offsets");
    if (3 * total_cap * 16 >= (int64_t)0xFFFFFFF0u)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ord_sweep: the change lists exceed 4 GB");
    const int ch = xc::chunks_for(max_row_nnz);
    if (ch == 0 || ch > 4)
        return xc::fail_arg(XC_ERR_ROW_TOO_LONG, "xc_bca_ord_sweep: a row holds %d entries, limit 256", max_row_nnz);
    // rows per wavefront: what the registers hold (rows x candidates per lane <= 4)
    int rpw = rows_per_wave >= 8 ? 8 : (rows_per_wave >= 4 ? 4 : (rows_per_wave >= 2 ? 2 : 1));
    if (rpw * ch > XC_ORD_MAX_RPW) rpw = XC_ORD_MAX_RPW / ch;
    int cu = 0;
    int rc = xc_device_info(&cu, nullptr, nullptr, 0);
    if (rc) return rc;
    // every workgroup must be resident (the grid barrier): one 1024-thread workgroup per CU always is
    if (workgroups > cu) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ord_sweep: %d workgroups on %d CUs", workgroups, cu);
    for (int i = 0; i < XC_ORD_ST_WORDS; ++i) status_host[i] = 0;
    status_host[XC_ORD_ST_ROWS_PER_WAVE] = rpw;
    if (n_order == 0) return XC_OK;
    hipStream_t st = xc::as_stream(stream);
    char *w = static_cast<char *>(workspace);
    const int64_t W = (int64_t)workgroups * XC_ORD_WAVES * rpw;
    unsigned *sync = reinterpret_cast<unsigned *>(w);
    long long *status = reinterpret_cast<long long *>(w + XC_ORD_SYNC_BYTES);
    char *q = w + XC_ORD_SYNC_BYTES + 256;
    unsigned *hot_dirty = reinterpret_cast<unsigned *>(q);
    q += ord_align(2 * XC_ORD_MAX_HOT * 4);
    unsigned *cnt = reinterpret_cast<unsigned *>(q);
    q += ord_align(3 * m * 4);
    xc::OrdEntry *ent = reinterpret_cast<xc::OrdEntry *>(q);
    q += ord_align(3 * total_cap * 16);
    double *hot_delta = reinterpret_cast<double *>(q);
    q += ord_align(2 * (int64_t)n_hot * W * 8);
    xc::double2_t *hot_prefix = reinterpret_cast<xc::double2_t *>(q);
    q += ord_align(2 * (int64_t)n_hot * W * 16);
    xc::double2_t *hot_total = reinterpret_cast<xc::double2_t *>(q);
    XC_HIP_TRY(hipMemsetAsync(w, 0, XC_ORD_SYNC_BYTES + 256, st));
    if (dtype == XC_F32) {
        xc::OrdParams<float> P{n_order, order, indptr, indices, static_cast<const float *>(data), pred_indices,
                               static_cast<float *>(pred_eta), sel, orphans, k, tpfp, s_entry, m, lab_dir, cnt, ent, total_cap,
                               hot_delta, hot_prefix, hot_total, hot_dirty, hot_labels, n_hot, sync, status, *metric_host, (double)n_norm,
                               (double)n_norm, maximize, skip_tn, epoch0, reinterpret_cast<unsigned long long *>(changed)};
        rc = xc::ord_launch(P, ch, rpw, workgroups, st);
    } else {
        xc::OrdParams<double> P{n_order, order, indptr, indices, static_cast<const double *>(data), pred_indices,
                                static_cast<double *>(pred_eta), sel, orphans, k, tpfp, s_entry, m, lab_dir, cnt, ent, total_cap,
                                hot_delta, hot_prefix, hot_total, hot_dirty, hot_labels, n_hot, sync, status, *metric_host, (double)n_norm,
                                (double)n_norm, maximize, skip_tn, epoch0, reinterpret_cast<unsigned long long *>(changed)};
        rc = xc::ord_launch(P, ch, rpw, workgroups, st);
    }
    if (rc) return rc;
    XC_CHECK_LAUNCH("bca_ordered_sweep_kernel");
    long long tmp[XC_ORD_ST_WORDS];
    XC_HIP_TRY(hipMemcpyAsync(tmp, status, sizeof(tmp), hipMemcpyDeviceToHost, st));
    XC_HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < XC_ORD_ST_WORDS; ++i)
        if (i != XC_ORD_ST_ROWS_PER_WAVE) status_host[i] = tmp[i];
    return XC_OK;
}

} // extern "C"

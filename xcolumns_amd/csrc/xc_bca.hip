// xc_bca.hip -- block coordinate ascent on CSR rows (the dominant kernel).
//
// Replaces the row loop of predict_using_bc_with_0approx
// (/root/reference/xcolumns/block_coordinate.py:448-463), its body
// _bc_with_0approx_step_csr (:212-293) and the numba routines that body calls:
// numba_sub_from_/add_to_unnormalized_confusion_matrix_csr
// (numba_csr_functions.py:385-452) and numba_set_gains_csr (:499-546).
//
// Data layout.  Per label j one 16-byte record {tp, fp} (float64) in `tpfp`: one
// candidate label = ONE 16-byte gather (buffer_load_dwordx4 sc1) from L2 /
// Infinity Cache instead of three 8-byte gathers from three vectors -- the sweep
// is bound by the number of scattered L2 requests per row, not by bytes.  The
// third statistic is carried as s = tp + fn = the column sum of y_proba over the
// rows counted so far (`colsum`, float64 per label): it does not change during a
// sweep, so it is expanded once per run into `s_entry`, one float64 per STORED
// ENTRY of y_proba, and streams in with the row (coalesced) instead of being
// gathered.  fn = s - tp and tn = n_counted - fp - s are derived in registers, so
// a change of prediction touches only tp and fp (two adjacent float64 atomics).
//
// One wavefront per row.  "Remove the row's contribution" (:243-246) is done in
// registers on the gathered values instead of by atomics on memory, and "add it
// back" (:290-293) becomes atomics only for labels whose membership changed --
// algebraically the same statistics, far fewer memory-side atomics.
//
// Sweep order and staleness.  The reference's loop is Gauss-Seidel: row i+1 sees
// row i's update.  Here `n_waves` wavefronts walk the order array interleaved
// (wave w takes positions w, w + n_waves, ...), all progressing at the same
// pace, so a row misses at most the updates of the ~n_waves rows around it in
// the order.  n_waves = 1 is the exact sequential sweep (used by the parity
// tests); the driver picks n_waves as a small fraction of n.
#include <hip/hip_ext.h>

#include <atomic>
#include <chrono>

#include "xc_common.h"
#include "xc_host.h"

namespace xc {

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
// one packed row entry: three dwords, dword-aligned (global_load_dwordx3)
struct __attribute__((packed, aligned(4))) pack3_t {
    unsigned x, y, z;
};

template <typename T>
struct SweepParams {
    int64_t n_order;
    const int32_t *order;
    const int32_t *indptr;
    const int32_t *indices;
    const T *data;
    int32_t *pred_indices;
    T *pred_eta;
    uint8_t *sel;           // per stored entry of y_proba: 1 = currently predicted
    const int32_t *orphans; // optional [n*k]: predicted columns the row does not store, -1 = none
    int k;
    double *tpfp;           // [m][2] float64 records {tp, fp} (master copy, atomics)
    float *shadow;          // optional [m][2] float32 copy of tpfp: the gather target of the
                            // concurrent sweep (8-byte records: twice the labels per L2 byte)
    double *colsum;         // [m] s = tp + fn
    const double *s_entry;  // [nnz] colsum expanded per stored entry (NULL in the greedy sweep)
    pack3_t *packed;        // optional [nnz] 12-byte entries {col | hot << 25 | sel << 31, eta (f32), s (f32)}:
                            // the row streams interleaved so a candidate is ONE 12-byte lane load
                            // (concurrent sweeps only: the exact mode reads the float64 s_entry)
    const int32_t *hot_labels; // optional [64] (with packed): label id of hot slot h = 1..63, -1 = unused
    double *acc;            // optional [2m + 1]: from-scratch {tp, fp} of the NEW prediction, [2m] += changed rows
    int acc_delta;          // with acc, commit protocol only: do NOT rebuild the statistics from scratch in `acc`;
                            // push every committed change into the float64 records `tpfp` instead (a sweep that
                            // gathers the float32 shadow never reads them) -- the boundary takes them from there
    int64_t m;
    unsigned tpfp_bytes;
    xc_metric metric;      // as given (EXACT path)
    xc_metric metric_fast; // epsilon * n, kf * n: evaluates the raw statistics (non-exact path)
    double nn;             // divisor n of the step (block_coordinate.py:229-231)
    double n_counted; // rows counted in the statistics when not greedy
    int maximize;
    int greedy;
    int skip_tn;
    int n_waves;
    int validate; // concurrent mode: 0 = none, 1 = re-read the records of the labels a row is about to
                  // change, 2 = commit them with returning atomics and compare (the default)
    float hot_unpublished; // HOT: share of the rows whose hot-label deltas may wait in the workgroups' LDS tables
    float conflict_rel;   // commit protocol: a returned record that differs from the scored one by less than
                          // this share of (tp + fp) is not a conflict (a label that sums hundreds of rows moves
                          // all the time and one row's change cannot move its gain)
    unsigned long long *changed;
    unsigned long long *stamps; // diagnostic builds only (-DXC_STAMPS): per-phase cycle sums
    const double *ctrl;         // optional device-side loop control (XC_CTRL_*): stop flag and wave count
};

// In-kernel phase stamps (cdna_hip_programming.md section 7): compiled in only with
// -DXC_STAMPS, never in the shipped library.  One asm statement = s_memtime plus
// its own lgkmcnt wait; the sums go to a buffer no other code reads.
#ifdef XC_STAMPS
#define XC_NSTAMP 8
#define XC_STAMP_DECL unsigned long long st_prev = 0, st_sum[XC_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0}
#define XC_STAMP_START() do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_prev = t_; } while (0)
#define XC_STAMP(i) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); st_sum[i] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define XC_STAMP_DECL
#define XC_STAMP_START()
#define XC_STAMP(i)
#endif

// Normalisation by n (block_coordinate.py:252-264).  EXACT divides every entry by
// n like the reference (bit-identical gains; the sequential mode that must
// reproduce the reference's trajectory).  The concurrent mode skips it: every
// metric here is a ratio of the entries, so psi(x / n; eps, k) = psi(x; eps * n,
// k * n) -- the host passes the rescaled constants in `metric_fast` and the
// kernel feeds the raw statistics, saving six divisions per candidate.
// The row a wavefront works on: everything that does not depend on the
// statistics, so it can be fetched ahead of time.
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
#define XC_RSRC_WORD3 0x00020000 /* raw buffer, 32-bit data format (gfx9) */
#define XC_CPOL_SC1 16           /* cache policy bit 4 = sc1 on gfx94x/gfx950 */
#define XC_MAX_RETRY 3           /* optimistic validation: re-score a row at most this often */
/* packed entry word 0: column id (25 bits) | hot slot (6 bits) | sel (1 bit) */
#define XC_PACK_COL_MASK 0x01ffffffu
#define XC_PACK_HOT_SHIFT 25
#define XC_PACK_HOT_MASK 63u
#define XC_HOT_FLUSH_ROWS 8      /* a wave publishes its hot-label deltas every this many rows */

template <typename T, int CH>
struct RowData {
    int idx[CH];
    T eta[CH];
    uint8_t sel[CH]; // is the entry in the row's current prediction
    uint8_t hot[CH]; // packed stream only: hot slot 1..63 of the entry's label, 0 = none
    double sc[CH];   // column sum of the entry's label (s_entry), non-greedy sweeps
};

// All lanes load (clamped to the row's last entry): straight-line code keeps the
// loads in flight under precise vmcnt waits instead of exec-masked blocks; lanes
// past the row end are masked out by `p < r` later.
template <typename T, int CH, bool PACKED>
__device__ __forceinline__ void load_row(const SweepParams<T> &P, int s, int r, int lane, RowData<T, CH> &d) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int p = lane + XC_WAVE * c;
        // an empty row (refused by the host wrappers, but never trusted here) must not read before
        // the row's first entry
        const int pc = p < r ? p : (r > 0 ? r - 1 : 0);
        // read-once streams: non-temporal, so they do not evict the {tp, fp} records
        // (the gather table) from the XCD's L2
        if (PACKED) {
            // one 12-byte load per lane = 768 contiguous bytes per wave instruction
            const pack3_t *e = P.packed + s + pc;
            const unsigned wx = __builtin_nontemporal_load(&e->x);
            const unsigned wy = __builtin_nontemporal_load(&e->y);
            const unsigned wz = __builtin_nontemporal_load(&e->z);
            d.idx[c] = (int)(wx & XC_PACK_COL_MASK);
            d.hot[c] = (uint8_t)((wx >> XC_PACK_HOT_SHIFT) & XC_PACK_HOT_MASK);
            d.sel[c] = (uint8_t)(wx >> 31);
            d.eta[c] = (T)__uint_as_float(wy);
            d.sc[c] = (double)__uint_as_float(wz);
        } else {
            d.idx[c] = __builtin_nontemporal_load(P.indices + s + pc);
            d.eta[c] = __builtin_nontemporal_load(P.data + s + pc);
            d.sel[c] = __builtin_nontemporal_load(P.sel + s + pc);
            d.hot[c] = 0;
            d.sc[c] = P.s_entry ? __builtin_nontemporal_load(P.s_entry + s + pc) : 0.0;
        }
    }
}

// SHADOW (only with !EXACT, never greedy): gather the float32 copy of the records.
// PACKED (float32 scores, never greedy): the row streams come interleaved from `packed`.
// HOT (with SHADOW, PACKED and acc): deltas to the hot labels are batched per workgroup.
#ifdef XC_SWEEP_WAVES_PER_EU /* experiment builds (tools/build_variant.sh): trade spills for resident wavefronts */
#define XC_SWEEP_OCC __attribute__((amdgpu_waves_per_eu(XC_SWEEP_WAVES_PER_EU, 8)))
#else
#define XC_SWEEP_OCC
#endif
template <typename T, int CH, bool EXACT, bool HAS_ORDER, bool SHADOW, bool PACKED, bool HOT>
__global__ __launch_bounds__(XC_BLOCK) XC_SWEEP_OCC void bca_sweep_csr_kernel(SweepParams<T> P) {
    const int lane = lane_id();
    const int wave = blockIdx.x * (XC_BLOCK / XC_WAVE) + (threadIdx.x >> 6);
    // hot-label delta table of the workgroup (see flush_hot below): zeroed before any wave leaves
    __shared__ float s_hot[XC_WAVE][2];
    // ... and the workgroup's copy of the hot RECORDS: the published value as of its last flush.  Rows read
    // their hot candidates from here (published + this workgroup's pending deltas) instead of gathering the
    // 63 lines every publication of every workgroup rewrites -- 17 M gathers of lines that the atomics keep
    // dropping from the L2s were the largest single cost of a first sweep on Zipf popularity.
    __shared__ float s_hotrec[XC_WAVE][2];
    // acc_delta: the workgroup's changes of the hot labels in float64, pushed into tpfp once, when it ends
    __shared__ double s_hot64[XC_WAVE][2];
    __shared__ int s_hot_ticks, s_hot_done;
    if (HOT) {
        if (threadIdx.x < XC_WAVE) {
            s_hot[threadIdx.x][0] = 0.0f;
            s_hot[threadIdx.x][1] = 0.0f;
            s_hot64[threadIdx.x][0] = 0.0;
            s_hot64[threadIdx.x][1] = 0.0;
            const int hl = P.hot_labels[threadIdx.x];
            s_hotrec[threadIdx.x][0] = hl >= 0 ? __hip_atomic_load(P.shadow + (int64_t)hl * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
            s_hotrec[threadIdx.x][1] = hl >= 0 ? __hip_atomic_load(P.shadow + (int64_t)hl * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
        }
        if (threadIdx.x == 0) s_hot_ticks = s_hot_done = 0;
        __syncthreads();
    }
    // device-side loop control (xc_bca_plan_*_pipelined): the previous boundary decided on
    // the GPU whether this sweep runs at all and with how many wavefronts; the grid is
    // launched for the largest count and the surplus waves leave here
    int n_walk = P.n_waves;
    if (P.ctrl) {
        if (P.ctrl[XC_CTRL_STOP] != 0.0) return;
        n_walk = (int)P.ctrl[XC_CTRL_WAVES];
    }
    if (wave >= n_walk) return;
    const int k = P.k;
    const double nn = P.nn;
    const bool greedy = P.greedy != 0;
    const bool skip_tn = P.skip_tn != 0;
    const bool commit_mode = !EXACT && !greedy && P.validate == 2;
    // does this sweep rebuild the new prediction's statistics from scratch in `acc` (every row adds its k labels),
    // or push the committed changes into the float64 records (acc_delta: changing rows only)
    const bool acc_scratch = P.acc != nullptr && !(P.acc_delta != 0 && commit_mode);
    const int64_t W = n_walk;
    const int64_t last = P.n_order - 1;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(P.tpfp, 0, P.tpfp_bytes, XC_RSRC_WORD3);
    // float32 shadow of the records: only the concurrent (non-exact), non-greedy sweep reads it
    // descriptors are built from kernel arguments (scalar registers); a runtime select
    // between pointers would make hipcc wrap every buffer load in a waterfall loop
    const __amdgpu_buffer_rsrc_t rsrc32 =
        __builtin_amdgcn_make_buffer_rsrc(P.shadow, 0, SHADOW ? P.tpfp_bytes / 2 : 0u, XC_RSRC_WORD3);
    unsigned long long n_changed = 0;

    // Software pipeline over the wave's positions pos, pos+W, pos+2W, ...:
    // while row t is being scored, the CSR entries of row t+1, the indptr pair
    // of row t+2 and the order entry of row t+3 are in flight.  None of these
    // depends on the statistics, so fetching them early adds no staleness; the
    // statistics themselves are gathered at the last moment.
    auto row_at = [&](int64_t pos) -> int {
        const int64_t q = pos < last ? pos : last; // clamp: tail prefetches stay in bounds
        return HAS_ORDER ? P.order[q] : (int)q; // compile-time: no branch around the load
    };
    int64_t pos = wave;
    // row ids and row bounds are the same in all lanes: readfirstlane moves them to
    // scalar registers, so row addresses are SGPR base + lane offset
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    int row0 = uni(row_at(pos)), row1 = uni(row_at(pos + W)), row2 = uni(row_at(pos + 2 * W));
    int s0 = uni(P.indptr[row0]), e0 = uni(P.indptr[row0 + 1]);
    int s1 = uni(P.indptr[row1]), e1 = uni(P.indptr[row1 + 1]);
    RowData<T, CH> cur;
    load_row<T, CH, PACKED>(P, s0, e0 - s0, lane, cur);
    // The from-scratch accumulation of a row's new prediction into `acc` is not
    // urgent, and vmcnt retires in issue order: issued right after the decision it
    // would sit in front of the NEXT row's gathers and their wait would pay the
    // atomics' ~700-cycle round trip.  So a row's acc atomics are issued one
    // iteration later, just after the next row's gathers.  They are also packed:
    // the (label, eta) of the q-th predicted entry is moved (through LDS) to lanes
    // 2q and 2q+1, which add eta and 1-eta to the label's two ADJACENT float64 slots
    // in ONE wave instruction -- one 16-byte request per label at the memory-side
    // atomic units instead of two (they are request-bound: ~21 G scattered/s).
    __shared__ int s_pack_idx[XC_BLOCK / XC_WAVE][XC_MAX_K];
    __shared__ T s_pack_eta[XC_BLOCK / XC_WAVE][XC_MAX_K];
    const int wib = threadIdx.x >> 6;
    int pend_idx = 0;      // lane L < 2 * pend_n: label of predicted entry L / 2
    double pend_val = 0.0; // eta (even lane) or 1 - eta (odd lane)
    int pend_n = 0;
    auto flush_pending = [&]() {
#ifndef XC_EXP_SKIP_ACC /* diagnostic builds only (tools/build_exp.sh): what does this traffic cost? */
        if (lane < 2 * pend_n) atomic_add_f64(P.acc + (int64_t)pend_idx * 2 + (lane & 1), pend_val);
#endif
    };
    // Hot labels (the head of a Zipf-like popularity): in the first sweeps nearly every row
    // changes them, and float atomics on ONE address run one after the other at the memory side
    // (11-35 ns each: 1.8 ms of a 2.7 ms sweep at C2-Zipf).  The waves of a workgroup therefore
    // sum their deltas to the (at most 63) hot labels in an LDS table -- slot h = lane h at flush
    // time -- and the workgroup publishes it about every XC_HOT_FLUSH_ROWS rows per wave as one
    // atomic per label.  Their statistics are sums over thousands of rows, so a few rows' delay
    // moves a gain by a relative 1e-3 at most; for the same reason hot labels are left out of the
    // optimistic validation (their records move all the time).
    constexpr bool hot_on = HOT;
    const int my_hot_label = hot_on ? P.hot_labels[lane] : -1;
    const int waves_here = (int)((n_walk - (int64_t)blockIdx.x * (XC_BLOCK / XC_WAVE)) < (XC_BLOCK / XC_WAVE)
                                     ? (n_walk - (int64_t)blockIdx.x * (XC_BLOCK / XC_WAVE))
                                     : (XC_BLOCK / XC_WAVE));
    int hot_rows = 0;
    // A wave ticks the workgroup's publication counter every hot_flush_rows rows, so about hot_flush_rows / 2
    // rows per wavefront are unpublished at any time: keep that below P.hot_unpublished of the rows whatever
    // the width -- a narrow sweep publishes less often (publications serialise at the memory side, ~80 ns
    // each per record: n / (4 * hot_flush_rows) of them per hot record and sweep).
    int hot_flush_rows = (int)(2.0f * P.hot_unpublished * (float)P.n_order / (float)n_walk);
    hot_flush_rows = hot_flush_rows < XC_HOT_FLUSH_ROWS ? XC_HOT_FLUSH_ROWS : (hot_flush_rows > 64 ? 64 : hot_flush_rows);
    auto flush_hot = [&]() { // lane h publishes the workgroup's sum for hot slot h and refreshes its copy of the record
        if (my_hot_label >= 0) {
            const float a = __hip_atomic_exchange(&s_hot[lane][0], 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const float b = __hip_atomic_exchange(&s_hot[lane][1], 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // until the atomics return, readers keep seeing this workgroup's own part (copy + a)
            s_hotrec[lane][0] += a;
            s_hotrec[lane][1] += b;
#ifndef XC_EXP_SKIP_HOT
            float *rec = P.shadow + (int64_t)my_hot_label * 2;
            // the returning add hands back what every OTHER workgroup has published meanwhile
            const float na = a != 0.0f ? atomic_add_ret_f32(rec, a) + a
                                       : __hip_atomic_load(rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float nb = b != 0.0f ? atomic_add_ret_f32(rec + 1, b) + b
                                       : __hip_atomic_load(rec + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_hotrec[lane][0] = na;
            s_hotrec[lane][1] = nb;
#endif
        }
    };
    // a candidate's float32 record: hot labels from the workgroup's LDS copy (no request leaves the lane: an
    // offset past the buffer's end returns zeros without touching memory), the others gathered coherently
    auto gather32 = [&](int col, int hot) -> float2_t {
        const unsigned off = (hot_on && hot != 0) ? 0xFFFFFFF0u : (unsigned)col * 8u;
        float2_t v = __builtin_bit_cast(float2_t, __builtin_amdgcn_raw_buffer_load_b64(rsrc32, (int)off, 0, XC_CPOL_SC1));
        if (hot_on && hot != 0) {
            v.x = s_hotrec[hot][0] + s_hot[hot][0];
            v.y = s_hotrec[hot][1] + s_hot[hot][1];
        }
        return v;
    };
    XC_STAMP_DECL;
    XC_STAMP_START();
    for (; pos < P.n_order; pos += W) {
        const int64_t row = row0;
        const int r = e0 - s0;
        int32_t *p_idx = P.pred_indices + row * k;
        T *p_eta = P.pred_eta + row * k;

        // ---- gather the statistics of the candidate labels: one 16-byte
        // {tp, fp} record per candidate, agent-coherent (sc1: served by L2, never by
        // this CU's L1, so other waves' atomics are seen) ----
        double2_t rec64[CH];
        float2_t rec32[CH];
        double sc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (SHADOW) {
                rec32[c] = gather32(cur.idx[c], cur.hot[c]);
            } else {
                rec64[c] = __builtin_bit_cast(
                    double2_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, cur.idx[c] * 16, 0, XC_CPOL_SC1));
            }
            sc[c] = greedy ? load_coherent(P.colsum + cur.idx[c]) : cur.sc[c];
        }

        // ---- prefetch for the following rows (issued AFTER the gathers so the
        // wait on the gathers does not also wait for these; the scheduling barrier
        // keeps hipcc from hoisting them above the gathers) ----
        __builtin_amdgcn_sched_barrier(0);
        RowData<T, CH> nxt;
        load_row<T, CH, PACKED>(P, s1, e1 - s1, lane, nxt);
        const int s2 = P.indptr[row2], e2 = P.indptr[row2 + 1];
        const int row3 = row_at(pos + 3 * W);
        if (acc_scratch) flush_pending(); // the previous row's contribution to acc (younger than the gathers)
        XC_STAMP(0); // issue gathers + prefetches

        // ---- membership of the candidates in the current prediction comes with
        // the row (sel flags).  Predicted columns the row does not store ("orphans":
        // eta = 0, they contributed fp += 1, numba_csr_functions.py:200-203) exist
        // only in foreign initial predictions; they leave the prediction here ----
        bool in_old[CH];
        int n_old = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            in_old[c] = !greedy && cur.sel[c] != 0 && (lane + XC_WAVE * c < r);
            n_old += __popcll(__ballot(in_old[c]));
        }
        if (P.orphans && !greedy && lane < k) {
            const int oid = P.orphans[row * k + lane];
            if (oid >= 0) {
                if (!(SHADOW && acc_scratch)) atomic_add_f64(P.tpfp + (int64_t)oid * 2 + 1, -1.0);
                if (P.shadow) atomic_add_f32(P.shadow + (int64_t)oid * 2 + 1, -1.0f);
            }
        }
        XC_STAMP(1); // membership
        // Optimistic validation (concurrent mode): a row that decides to CHANGE its
        // prediction re-reads the records of all its candidates; if a record of a
        // label it is about to add or drop moved meanwhile (another wave changed the
        // same label), the row is re-scored on the fresh records.  Two rows in
        // flight that both want the same label are thereby serialised like in the
        // sequential sweep; rows that change nothing pay nothing.
        // Commit protocol (concurrent mode, P.validate == 2, the default): a row that decides to CHANGE
        // its prediction pushes the deltas of the labels it adds or drops at once, with RETURNING
        // atomics, and compares what they return with the record values it scored on.  Equal: nobody
        // touched those labels between this row's gather and its commit -- the commit is exactly what
        // the sequential sweep would have done.  Different: another row in flight changed the same label
        // first; this row's prediction in memory is now `in_new`, and it is simply processed again on
        // fresh records (its own contribution removed in registers as always).  Two rows that want the
        // same label are thereby serialised whatever their timing -- the validate-then-commit form
        // below leaves the atomics' flight time (~1 us) as a window in which both still take it.
        bool in_new[CH], in_cur[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) in_cur[c] = in_old[c];
        int n_cur = n_old;
        bool row_changed = false;
        const int kk = r < k ? r : k;
        // The records are first USED here, in straight-line code: the wait hipcc puts in front of this is the
        // exact one (vmcnt = the prefetches issued after the gathers).  Inside the retry loop, where a second
        // pass reads re-gathered records, its bookkeeping merges the two paths into vmcnt(0) -- every row
        // would also wait for the next row's stream loads and for the previous row's `acc` atomics, both
        // younger than the gathers and both slower.
        double tp_now[CH], fp_now[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            tp_now[c] = SHADOW ? (double)rec32[c].x : rec64[c].x;
            fp_now[c] = SHADOW ? (double)rec32[c].y : rec64[c].y;
        }
        for (int attempt = 0;; ++attempt) {
        // ---- gains (block_coordinate.py:248-282) ----
        unsigned long long key[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            key[c] = 0ull;
            if (lane + XC_WAVE * c < r) {
                const T e = cur.eta[c];
                const T om = (T)1 - e; // (1 - t_data) in the input dtype, :253
                const double ed = (double)e;
                const double omd = (double)om;
                double tpc = tp_now[c];
                double fpc = fp_now[c];
                double scc = sc[c];
                // statistics without this row (:243-246, done in registers)
                if (in_cur[c]) {
                    tpc -= ed;
                    fpc -= omd;
                }
                if (!greedy) scc -= ed;
                const double fn = scc - tpc;
                const double n1 = greedy ? (double)pos : (P.n_counted - 1.0); // rows counted besides this one
                const double tn = n1 - fpc - scc;
                double g;
                if (EXACT) {
                    // :252-264
                    const double pos_tp = (tpc + ed) / nn;
                    const double pos_fp = (fpc + omd) / nn;
                    const double neg_fn = (fn + ed) / nn;
                    const double neg_tp = tpc / nn;
                    const double neg_fp = fpc / nn;
                    const double pos_fn = fn / nn;
                    // skip_tn: Etn is the constant -1, undivided (:260-261); zeros in the
                    // greedy first sweep (:427)
                    double pos_tn = greedy ? 0.0 : -1.0, neg_tn = pos_tn;
                    if (!skip_tn) {
                        neg_tn = (tn + omd) / nn;
                        pos_tn = tn / nn;
                    }
                    // :267-282
                    g = metric_eval_t<true>(P.metric, pos_tp, pos_fp, pos_fn, pos_tn) -
                        metric_eval_t<true>(P.metric, neg_tp, neg_fp, neg_fn, neg_tn);
                } else {
                    double pos_tn = greedy ? 0.0 : -nn, neg_tn = pos_tn; // -1 * n
                    if (!skip_tn) {
                        neg_tn = tn + omd;
                        pos_tn = tn;
                    }
                    if (P.metric_fast.base == XC_M_FBETA && !P.metric_fast.mixed) {
                        // F-beta (the headline metric): psi(with) - psi(without) over ONE denominator.  With D = b2 (tp + fp) + tp +
                        // fn + eta + eps and s = eta + (1 - eta) (the reference's 1 - eta is rounded to the scores' type, so s is 1 only
                        // up to that rounding) the two values are (1 + b2)(tp + eta) / (D + b2 s) and (1 + b2) tp / D, so the gain is
                        // (1 + b2)(eta D - b2 s tp) / (D (D + b2 s)): one reciprocal per candidate instead of two, and no cancellation
                        // between two nearly equal quotients
                        const double b2 = P.metric_fast.beta * P.metric_fast.beta;
                        const double b2s = b2 * (ed + omd);
                        const double den = (b2 * (tpc + fpc)) + tpc + (fn + ed) + P.metric_fast.epsilon;
                        g = fdiv<false>((1.0 + b2) * (ed * den - b2s * tpc), den * (den + b2s));
                    } else if (P.metric_fast.base == XC_M_JACCARD && !P.metric_fast.mixed) {
                        // (tp + eta) / (D + 1 - eta) - tp / D with D = tp + fp + fn + eta + eps
                        const double den = tpc + fpc + (fn + ed) + P.metric_fast.epsilon;
                        g = fdiv<false>(ed * den - tpc * omd, den * (den + omd));
                    } else if (P.metric_fast.base == XC_M_PRECISION && !P.metric_fast.mixed) {
                        // (tp + eta) / (D + s) - tp / D with D = tp + fp + eps, s = eta + (1 - eta)
                        const double den = tpc + fpc + P.metric_fast.epsilon, s1 = ed + omd;
                        g = fdiv<false>(ed * den - tpc * s1, den * (den + s1));
                    } else if (P.metric_fast.base == XC_M_RECALL && !P.metric_fast.mixed) {
                        // (tp + eta) / D - tp / D with D = tp + fn + eta + eps (the row's eta is on one side or the other)
                        g = fdiv<false>(ed, tpc + (fn + ed) + P.metric_fast.epsilon);
                    } else {
                        g = metric_eval_t<false>(P.metric_fast, tpc + ed, fpc + omd, fn, pos_tn) -
                            metric_eval_t<false>(P.metric_fast, tpc, fpc, fn + ed, neg_tn);
                    }
                }
                if (!P.maximize) g = -g;
                key[c] = sortable_key(nan_to_neg_inf(g));
            }
        }

        XC_STAMP(2); // wait for the gathers + gains
        // ---- top-k (numba_set_gains_csr -> numba_argtopk_csr,
        // numba_csr_functions.py:455-466, :514-524).
        // Fast path: start from the current prediction and swap its worst member
        // for the best outsider while the outsider is strictly better -- the set
        // is the top-k exactly when no such swap is left, and most rows need none.
        // Two DPP wave reductions per check.  Any tie at the boundary, or a
        // prediction that does not hold exactly k of the row's entries, goes to
        // the exact path below.
#pragma unroll
        for (int c = 0; c < CH; ++c) in_new[c] = in_cur[c];
        bool exact_path = (n_cur != kk);
        if (!exact_path) {
            for (int it = 0; it <= kk; ++it) {
                unsigned long long lmin = ~0ull, lmax = 0ull;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if (in_new[c]) lmin = key[c] < lmin ? key[c] : lmin;
                    else lmax = key[c] > lmax ? key[c] : lmax; // key 0 = no candidate
                }
                // high words first: they almost always decide; low words only when
                // the two boundary keys share their high word
                const unsigned smin_hi = wave_umin32((unsigned)(lmin >> 32));
                const unsigned umax_hi = wave_umax32((unsigned)(lmax >> 32));
                if (umax_hi < smin_hi) break;
                const unsigned smin_lo = wave_umin32((unsigned)(lmin >> 32) == smin_hi ? (unsigned)lmin : ~0u);
                const unsigned umax_lo = wave_umax32((unsigned)(lmax >> 32) == umax_hi ? (unsigned)lmax : 0u);
                const unsigned long long smin = ((unsigned long long)smin_hi << 32) | smin_lo;
                const unsigned long long umax = ((unsigned long long)umax_hi << 32) | umax_lo;
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
            // The k-th largest key by bisection on the key bits with wave ballots:
            // `thr` grows bit by bit while at least k candidates stay >= thr and
            // stops early once exactly k do.  Equal keys at the boundary go to the
            // lower position (= lower column id).
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
                // thr is the k-th largest key and it repeats: all larger keys, then
                // the first (kk - #larger) of the equal ones in position order
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

        // ---- write the new prediction (ascending columns) and push the change
        // of the statistics (:290-293 minus :243-246) to memory ----
        bool any_change = false;
#pragma unroll
        for (int c = 0; c < CH; ++c) any_change = any_change || (in_new[c] != in_cur[c]);
        row_changed = __ballot(any_change) != 0ull;
        if (commit_mode) {
            if (!row_changed) break; // nothing (more) to change: memory holds `in_cur`
            bool conflict = false;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (in_new[c] != in_cur[c]) {
                    const double sgn = in_new[c] ? 1.0 : -1.0;
                    const double ed = (double)cur.eta[c];
                    const double omd = (double)((T)1 - cur.eta[c]);
                    if (hot_on && cur.hot[c] != 0) { // summed per workgroup, published by flush_hot
                        (void)__hip_atomic_fetch_add(&s_hot[cur.hot[c]][0], (float)(sgn * ed), __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP);
                        (void)__hip_atomic_fetch_add(&s_hot[cur.hot[c]][1], (float)(sgn * omd), __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (!acc_scratch) { // exact float64 sums for the records, one push per workgroup at its end
                            (void)__hip_atomic_fetch_add(&s_hot64[cur.hot[c]][0], sgn * ed, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WORKGROUP);
                            (void)__hip_atomic_fetch_add(&s_hot64[cur.hot[c]][1], sgn * omd, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    } else if (SHADOW) {
                        float *sh = P.shadow + (int64_t)cur.idx[c] * 2;
                        const float was_tp = atomic_add_ret_f32(sh + 0, (float)(sgn * ed));
                        const float was_fp = atomic_add_ret_f32(sh + 1, (float)(sgn * omd));
                        conflict = conflict || (fabsf(was_tp - rec32[c].x) + fabsf(was_fp - rec32[c].y) >
                                                P.conflict_rel * (rec32[c].x + rec32[c].y));
                        if (!acc_scratch) { // the float64 records are what the boundary reads (no acc, or acc_delta)
                            atomic_add_f64(P.tpfp + (int64_t)cur.idx[c] * 2, sgn * ed);
                            atomic_add_f64(P.tpfp + (int64_t)cur.idx[c] * 2 + 1, sgn * omd);
                        }
                    } else {
                        double *st = P.tpfp + (int64_t)cur.idx[c] * 2;
                        const double was_tp = atomic_add_ret_f64(st + 0, sgn * ed);
                        const double was_fp = atomic_add_ret_f64(st + 1, sgn * omd);
                        conflict = conflict || (fabs(was_tp - rec64[c].x) + fabs(was_fp - rec64[c].y) >
                                                (double)P.conflict_rel * (rec64[c].x + rec64[c].y));
                        if (P.shadow) { // keep the float32 copy in step
                            atomic_add_f32(P.shadow + (int64_t)cur.idx[c] * 2, (float)(sgn * ed));
                            atomic_add_f32(P.shadow + (int64_t)cur.idx[c] * 2 + 1, (float)(sgn * omd));
                        }
                    }
                }
                in_cur[c] = in_new[c];
            }
            n_cur = kk;
            if (__ballot(conflict) == 0ull || attempt >= XC_MAX_RETRY) break;
            // both of this row's atomics on a label have returned, i.e. have been performed at the
            // memory side (and dropped the line from this XCD's L2): the gather below sees them
        } else {
        if (EXACT || greedy || P.validate != 1 || !row_changed || attempt >= XC_MAX_RETRY) break;
        // first only the records of the labels this row adds or drops (2-3 of its ~50
        // candidates: inactive lanes issue no request) ...
        bool moved = false;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (in_new[c] != in_old[c] && !(hot_on && cur.hot[c] != 0)) {
                if (SHADOW) {
                    const float2_t now = __builtin_bit_cast(
                        float2_t, __builtin_amdgcn_raw_buffer_load_b64(rsrc32, cur.idx[c] * 8, 0, XC_CPOL_SC1));
                    moved = moved || now.x != rec32[c].x || now.y != rec32[c].y;
                } else {
                    const double2_t now = __builtin_bit_cast(
                        double2_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, cur.idx[c] * 16, 0, XC_CPOL_SC1));
                    moved = moved || now.x != rec64[c].x || now.y != rec64[c].y;
                }
            }
        }
        if (__ballot(moved) == 0ull) break;
        }
        // ... and only after a conflict every candidate's record, to score the row again
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (SHADOW)
                rec32[c] = gather32(cur.idx[c], cur.hot[c]);
            else
                rec64[c] = __builtin_bit_cast(
                    double2_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, cur.idx[c] * 16, 0, XC_CPOL_SC1));
            tp_now[c] = SHADOW ? (double)rec32[c].x : rec64[c].x;
            fp_now[c] = SHADOW ? (double)rec32[c].y : rec64[c].y;
        }
        } // retry
        if (commit_mode) { // in_new == in_cur == what memory holds; did the row end where it started?
            bool differs = false;
#pragma unroll
            for (int c = 0; c < CH; ++c) differs = differs || (in_new[c] != in_old[c]);
            row_changed = __ballot(differs) != 0ull;
        }
        // The prefetched row, its bounds and the next order entry are consumed HERE, before this row's stores:
        // a first use at the rotation below would have to wait (vmcnt(0): the stores are conditional, hipcc
        // cannot count past them) for every store and atomic of this row to be acknowledged.
        const int s2u = uni(s2), e2u = uni(e2), row3u = uni(row3);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            int hs = (int)nxt.hot[c] | ((int)nxt.sel[c] << 8);
            asm volatile("" : "+v"(nxt.idx[c]), "+v"(nxt.eta[c]), "+v"(nxt.sc[c]), "+v"(hs));
            nxt.hot[c] = (uint8_t)(hs & 0xff);
            nxt.sel[c] = (uint8_t)(hs >> 8);
        }
        XC_STAMP(3); // top-k
        // The from-scratch recompute of the sweep boundary (block_coordinate.py:465-467:
        // tp / fp of the new prediction summed over ALL rows) is accumulated row by row
        // instead of by a separate pass over the prediction afterwards; the atomics
        // themselves go out in the next iteration (flush_pending).
        if (acc_scratch) {
            int nsel = 0;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const unsigned long long mask = __ballot(in_new[c]);
                if (in_new[c]) {
                    const int slot = nsel + __popcll(mask & lanemask_lt());
                    s_pack_idx[wib][slot] = cur.idx[c];
                    s_pack_eta[wib][slot] = cur.eta[c];
                }
                nsel += __popcll(mask);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // one deferred wave instruction carries 32 predicted entries (two lanes each) ...
            pend_n = nsel < XC_WAVE / 2 ? nsel : XC_WAVE / 2;
            const int q = (lane >> 1) < nsel ? (lane >> 1) : 0;
            pend_idx = s_pack_idx[wib][q];
            const T e = s_pack_eta[wib][q];
            pend_val = (lane & 1) ? (double)((T)1 - e) : (double)e;
            // ... the rest (budgets k > 32 only) goes out at once
            for (int q0 = XC_WAVE / 2; q0 < nsel; q0 += XC_WAVE / 2) {
                const int q2 = q0 + (lane >> 1);
                if (q2 < nsel) {
                    const T e2 = s_pack_eta[wib][q2];
#ifndef XC_EXP_SKIP_ACC
                    atomic_add_f64(P.acc + (int64_t)s_pack_idx[wib][q2] * 2 + (lane & 1),
                                   (lane & 1) ? (double)((T)1 - e2) : (double)e2);
#endif
                }
            }
        }
        if (row_changed || greedy) {
            int base = 0;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const unsigned long long mask = __ballot(in_new[c]);
                if (in_new[c]) {
                    const int slot = base + __popcll(mask & lanemask_lt());
                    p_idx[slot] = cur.idx[c];
                    p_eta[slot] = cur.eta[c];
                }
                base += __popcll(mask);
                if (lane + XC_WAVE * c < r && in_new[c] != (cur.sel[c] != 0)) {
                    P.sel[s0 + lane + XC_WAVE * c] = in_new[c] ? 1 : 0;
                    if (PACKED)
                        P.packed[s0 + lane + XC_WAVE * c].x =
                            (unsigned)cur.idx[c] | ((unsigned)cur.hot[c] << XC_PACK_HOT_SHIFT) |
                            (in_new[c] ? 0x80000000u : 0u);
                }
                if (lane + XC_WAVE * c < r) {
                    double *st = P.tpfp + (int64_t)cur.idx[c] * 2;
                    const double ed = (double)cur.eta[c];
                    const double omd = (double)((T)1 - cur.eta[c]);
                    if (greedy) {
                        atomic_add_f64(P.colsum + cur.idx[c], ed);
                        if (in_new[c]) {
                            atomic_add_f64(st + 0, ed);
                            atomic_add_f64(st + 1, omd);
                        }
                    } else if (!commit_mode && in_new[c] != in_old[c]) {
                        const double sgn = in_new[c] ? 1.0 : -1.0;
                        // a sweep that gathers the shadow AND accumulates the boundary's from-scratch
                        // statistics never reads the float64 records before the commit kernel
                        // overwrites them from `acc`: their delta atomics would be dead work
                        if (!(SHADOW && acc_scratch)) {
                            atomic_add_f64(st + 0, sgn * ed);
                            atomic_add_f64(st + 1, sgn * omd);
                        }
                        if (hot_on && cur.hot[c] != 0) { // summed per workgroup, published by flush_hot
                            (void)__hip_atomic_fetch_add(&s_hot[cur.hot[c]][0], (float)(sgn * ed), __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WORKGROUP);
                            (void)__hip_atomic_fetch_add(&s_hot[cur.hot[c]][1], (float)(sgn * omd), __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else if (P.shadow) { // keep the float32 copy in step
#ifndef XC_EXP_SKIP_DELTA
                            float *sh = P.shadow + (int64_t)cur.idx[c] * 2;
                            atomic_add_f32(sh + 0, (float)(sgn * ed));
                            atomic_add_f32(sh + 1, (float)(sgn * omd));
#endif
                        }
                    }
                }
            }
            if (row_changed) ++n_changed;
        }
        if (hot_on && ++hot_rows >= hot_flush_rows) { // one publication per round of the workgroup's waves
            hot_rows = 0;
            int tick = 0;
            if (lane == 0) tick = __hip_atomic_fetch_add(&s_hot_ticks, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            tick = __builtin_amdgcn_readfirstlane(tick);
            if ((tick + 1) % waves_here == 0) flush_hot();
        }

        // sequential mode: this wave's atomics must have been performed before it
        // gathers statistics for its next row
        if (P.n_waves == 1) __builtin_amdgcn_s_waitcnt(0);

        XC_STAMP(4); // stores + atomics
        // ---- rotate the pipeline ----
        cur = nxt;
        row0 = row1; s0 = s1; e0 = e1;
        row1 = row2; s1 = s2u; e1 = e2u;
        row2 = row3u;
#ifdef XC_STAMPS
        // the rotation consumes the prefetched registers: wait for them here so the
        // stamp prices it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        XC_STAMP(5); // prefetch landing
    }
    if (acc_scratch) flush_pending();
    if (hot_on) { // the last wave of the workgroup to finish publishes what is left
        int done = 0;
        if (lane == 0) done = __hip_atomic_fetch_add(&s_hot_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__builtin_amdgcn_readfirstlane(done) == waves_here - 1) {
            flush_hot();
            if (!acc_scratch && my_hot_label >= 0) {
                const double a = s_hot64[lane][0], b = s_hot64[lane][1];
                if (a != 0.0) atomic_add_f64(P.tpfp + (int64_t)my_hot_label * 2, a);
                if (b != 0.0) atomic_add_f64(P.tpfp + (int64_t)my_hot_label * 2 + 1, b);
            }
        }
    }
#ifdef XC_STAMPS
    if (P.stamps && lane == 0)
        for (int i = 0; i < XC_NSTAMP; ++i) atomicAdd(P.stamps + i, st_sum[i]);
#endif
    if (lane == 0 && n_changed) {
        if (P.changed) atomicAdd(P.changed, n_changed);
        if (P.acc) atomic_add_f64(P.acc + 2 * P.m, (double)n_changed);
    }
}

// ---- pred_eta lookup --------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(XC_BLOCK) void gather_pred_eta_kernel(
    int64_t n_k, int k, const int32_t *indptr, const int32_t *indices, const T *data,
    const int32_t *pred_indices, T *pred_eta, uint8_t *sel, int32_t *orphans) {
    const int64_t t = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x;
    if (t >= n_k) return;
    const int64_t row = t / k;
    const int col = pred_indices[t];
    int lo = indptr[row], hi = indptr[row + 1];
    while (lo < hi) { // lower_bound
        const int mid = (lo + hi) >> 1;
        if (indices[mid] < col) lo = mid + 1;
        else hi = mid;
    }
    const bool found = lo < indptr[row + 1] && indices[lo] == col;
    pred_eta[t] = found ? data[lo] : (T)0;
    if (found && sel) sel[lo] = 1;
    if (orphans) orphans[t] = found ? -1 : col;
}

// ---- s = column sums of y_proba ------------------------------------------------
template <typename T>
__global__ __launch_bounds__(XC_BLOCK) void colsum_csr_kernel(int64_t nnz, const int32_t *indices,
                                                              const T *data, double *colsum) {
    const int64_t stride = (int64_t)gridDim.x * XC_BLOCK;
    for (int64_t t = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x; t < nnz; t += stride)
        atomic_add_f64(colsum + indices[t], (double)data[t]);
}

// s_entry[p] = colsum[indices[p]]: the column sum carried next to every stored entry
__global__ __launch_bounds__(XC_BLOCK) void expand_colsum_kernel(int64_t nnz, const int32_t *indices,
                                                                 const double *colsum, double *s_entry) {
    const int64_t stride = (int64_t)gridDim.x * XC_BLOCK;
    for (int64_t t = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x; t < nnz; t += stride)
        s_entry[t] = colsum[indices[t]];
}

// packed[p] = {col | hot << 25 | sel << 31, eta, (float)s}: the four row streams of a float32 matrix interleaved
__global__ __launch_bounds__(XC_BLOCK) void pack_rows_kernel(int64_t nnz, const int32_t *indices, const float *data,
                                                             const uint8_t *sel, const double *s_entry,
                                                             const uint8_t *hot_slot, pack3_t *packed) {
    const int64_t stride = (int64_t)gridDim.x * XC_BLOCK;
    for (int64_t t = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x; t < nnz; t += stride) {
        pack3_t w;
        const unsigned col = (unsigned)indices[t];
        w.x = col | ((hot_slot ? (unsigned)hot_slot[col] & XC_PACK_HOT_MASK : 0u) << XC_PACK_HOT_SHIFT) |
              (sel[t] ? 0x80000000u : 0u);
        w.y = __float_as_uint(data[t]);
        w.z = __float_as_uint((float)s_entry[t]);
        packed[t] = w;
    }
}

// the same straight from the per-label column sums (one L2-resident gather per entry): the float64
// per-entry expansion is then needed only by the sweeps that do not read the packed stream
__global__ __launch_bounds__(XC_BLOCK) void pack_rows_colsum_kernel(int64_t nnz, const int32_t *indices,
                                                                    const float *data, const uint8_t *sel,
                                                                    const double *colsum, const uint8_t *hot_slot,
                                                                    pack3_t *packed) {
    const int64_t stride = (int64_t)gridDim.x * XC_BLOCK;
    for (int64_t t = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x; t < nnz; t += stride) {
        pack3_t w;
        const unsigned col = (unsigned)__builtin_nontemporal_load(indices + t);
        w.x = col | ((hot_slot ? (unsigned)hot_slot[col] & XC_PACK_HOT_MASK : 0u) << XC_PACK_HOT_SHIFT) |
              (__builtin_nontemporal_load(sel + t) ? 0x80000000u : 0u);
        w.y = __float_as_uint(__builtin_nontemporal_load(data + t));
        w.z = __float_as_uint((float)colsum[col]);
        packed[t] = w;
    }
}

// ---- tp / fp of the current prediction from scratch ------------------------------
template <typename T>
__global__ __launch_bounds__(XC_BLOCK) void accumulate_pred_kernel(int64_t n_k, const int32_t *pred_indices,
                                                                   const T *pred_eta, double *acc) {
    const int64_t stride = (int64_t)gridDim.x * XC_BLOCK;
    for (int64_t t = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x; t < n_k; t += stride) {
        const T e = pred_eta[t];
        double *a = acc + (int64_t)pred_indices[t] * 2;
        atomic_add_f64(a + 0, (double)e);            // pred * true
        atomic_add_f64(a + 1, (double)((T)1 - e));   // pred * (1 - true), 1 when the row lacks the column
    }
}

// ---- commit + utility ---------------------------------------------------------
// Labels are split into XC_UTILITY_PARTIALS contiguous strips, one workgroup
// each; the strip sum is a fixed-shape LDS tree, so the partials (and their
// in-order host sum) do not depend on timing.
__global__ __launch_bounds__(XC_BLOCK) void commit_utility_kernel(int64_t m, double nn, double n_counted,
                                                                  double *acc, int clear_acc, double *tpfp,
                                                                  float *shadow, const double *colsum,
                                                                  xc_metric metric, int skip_tn, double *partials,
                                                                  const double *ctrl) {
    __shared__ double red[XC_BLOCK];
    if (ctrl && ctrl[XC_CTRL_STOP] != 0.0) return; // the loop has stopped: the sweep before was a no-op
    const int64_t per = (m + XC_UTILITY_PARTIALS - 1) / XC_UTILITY_PARTIALS;
    const int64_t j0 = (int64_t)blockIdx.x * per;
    const int64_t j1 = (j0 + per < m) ? j0 + per : m;
    double sum = 0.0;
    for (int64_t j = j0 + threadIdx.x; j < j1; j += XC_BLOCK) {
        double tp, fp;
        if (acc && clear_acc != 2) {
            tp = acc[2 * j];
            fp = acc[2 * j + 1];
            tpfp[2 * j] = tp;
            tpfp[2 * j + 1] = fp;
            if (shadow) {
                shadow[2 * j] = (float)tp;
                shadow[2 * j + 1] = (float)fp;
            }
            if (clear_acc) {
                acc[2 * j] = 0.0;
                acc[2 * j + 1] = 0.0;
            }
        } else {
            tp = tpfp[2 * j];
            fp = tpfp[2 * j + 1];
            if (acc && shadow) { // clear_acc == 2: the sweep pushed its changes into tpfp; the float32 copy follows
                shadow[2 * j] = (float)tp;
                shadow[2 * j + 1] = (float)fp;
            }
        }
        const double sc = colsum[j];
        const double fn = sc - tp;
        // confusion_matrix.py:391-397; _calculate_utility gets Etn / n (:438-445)
        const double tn = skip_tn ? -1.0 : (n_counted - fp - sc);
        sum += metric_eval(metric, tp / nn, fp / nn, fn / nn, tn / nn);
    }
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int w = XC_BLOCK / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
    if (acc && blockIdx.x == 0 && threadIdx.x == 0) { // the changed-row count rides in acc[2m]
        partials[XC_UTILITY_PARTIALS] = acc[2 * m];
        if (clear_acc) acc[2 * m] = 0.0;
    }
}

// ---- device-side sweep loop control -------------------------------------------------
// The stopping rule of predict_using_bc_with_0approx (block_coordinate.py:486-493) and the
// wavefront-count policy evaluated on the GPU, so the host can enqueue sweep j + 1 before it has
// seen the utility of sweep j: a sweep launched after the rule fired is a no-op.
__global__ void bca_ctrl_init_kernel(double *ctrl, double old_sum, double tolerance, double divisor, int maximize,
                                     double policy_num, double world, double min_waves, double max_waves,
                                     double fixed_waves, double first_waves, double exact_below) {
    if (threadIdx.x != 0) return;
    ctrl[XC_CTRL_STOP] = 0.0;
    ctrl[XC_CTRL_OLD_SUM] = old_sum;
    ctrl[XC_CTRL_WAVES] = first_waves;
    ctrl[XC_CTRL_TOLERANCE] = tolerance;
    ctrl[XC_CTRL_DIVISOR] = divisor;
    ctrl[XC_CTRL_MAXIMIZE] = maximize ? 1.0 : 0.0;
    ctrl[XC_CTRL_POLICY_NUM] = policy_num;
    ctrl[XC_CTRL_WORLD] = world;
    ctrl[XC_CTRL_MIN_WAVES] = min_waves;
    ctrl[XC_CTRL_MAX_WAVES] = max_waves;
    ctrl[XC_CTRL_FIXED_WAVES] = fixed_waves;
    ctrl[XC_CTRL_EXACT_BELOW] = exact_below;
}

// The order every sum of the XC_UTILITY_PARTIALS partials uses, on the GPU and on the host
// (xc_utility_finish_host): a fixed pairwise tree, s[i] += s[i + stride] for stride = 512 .. 1.
__global__ __launch_bounds__(XC_UTILITY_PARTIALS / 2) void bca_boundary_finish_kernel(const double *partials,
                                                                                    double *ctrl, int slot,
                                                                                    double *host_ring, double seq) {
    __shared__ double s[XC_UTILITY_PARTIALS];
    double *ring = ctrl + XC_CTRL_RING + XC_CTRL_RING_STRIDE * slot;
    volatile double *hring = host_ring ? host_ring + XC_CTRL_RING_STRIDE * slot : nullptr;
    if (ctrl[XC_CTRL_STOP] != 0.0) { // already stopped: report that this step did not run
        if (threadIdx.x == 0) {
            ring[3] = 2.0;
            if (hring) {
                hring[3] = 2.0;
                __threadfence_system();
                hring[4] = seq;
            }
        }
        return;
    }
    const int t = threadIdx.x;
    s[t] = partials[t];
    s[t + XC_UTILITY_PARTIALS / 2] = partials[t + XC_UTILITY_PARTIALS / 2];
    __syncthreads();
    for (int stride = XC_UTILITY_PARTIALS / 2; stride > 0; stride >>= 1) {
        if (t < stride) s[t] += s[t + stride];
        __syncthreads();
    }
    if (t != 0) return;
    const double total = s[0];
    const double changed = partials[XC_UTILITY_PARTIALS];
    const double div = ctrl[XC_CTRL_DIVISOR];
    const double new_u = total / div, old_u = ctrl[XC_CTRL_OLD_SUM] / div;
    const double tol = ctrl[XC_CTRL_TOLERANCE];
    const bool stop = ctrl[XC_CTRL_MAXIMIZE] != 0.0 ? (new_u - old_u < tol) : (new_u - old_u > tol); // :486-489
    const double used = ctrl[XC_CTRL_WAVES]; // wavefronts the sweep before this boundary used
    // wavefronts of the next sweep: block_coordinate.WavePolicy.next
    const double max_w = ctrl[XC_CTRL_MAX_WAVES];
    double w;
    if (ctrl[XC_CTRL_FIXED_WAVES] > 0.0) {
        w = ctrl[XC_CTRL_FIXED_WAVES];
    } else {
        double c = changed / ctrl[XC_CTRL_WORLD];
        if (c < 1.0) c = 1.0;
        w = floor(ctrl[XC_CTRL_POLICY_NUM] / c);
        if (w < ctrl[XC_CTRL_MIN_WAVES]) w = ctrl[XC_CTRL_MIN_WAVES];
    }
    // A sweep the rule leaves fewer wavefronts than EXACT_BELOW belongs to the exact (ordered) sweep, which the host
    // paces: the loop PAUSES -- the sweep already enqueued behind this boundary finds the stop flag and does nothing,
    // the host reads flag 3, runs that sweep itself and may arm the loop again (block_coordinate.run_bca_sweeps).
    const bool pause = !stop && ctrl[XC_CTRL_FIXED_WAVES] <= 0.0 && w < ctrl[XC_CTRL_EXACT_BELOW];
    const double flag = stop ? 1.0 : (pause ? 3.0 : 0.0);
    ring[0] = total;
    ring[1] = changed;
    ring[2] = used;
    ring[3] = flag;
    if (hring) { // straight into the host's pinned ring: no copy engine, no event in the stream
        hring[0] = total;
        hring[1] = changed;
        hring[2] = used;
        hring[3] = flag;
        __threadfence_system();
        hring[4] = seq;
    }
    ctrl[XC_CTRL_OLD_SUM] = total;
    ctrl[XC_CTRL_STOP] = (stop || pause) ? 1.0 : 0.0;
    if (w > max_w) w = max_w;
    if (w < 1.0) w = 1.0;
    ctrl[XC_CTRL_WAVES] = w;
}

__global__ __launch_bounds__(XC_BLOCK) void utility_vectors_kernel(int64_t m, double nn, const double *stats,
                                                                   xc_metric metric, double *partials) {
    __shared__ double red[XC_BLOCK];
    const int64_t per = (m + XC_UTILITY_PARTIALS - 1) / XC_UTILITY_PARTIALS;
    const int64_t j0 = (int64_t)blockIdx.x * per;
    const int64_t j1 = (j0 + per < m) ? j0 + per : m;
    double sum = 0.0;
    for (int64_t j = j0 + threadIdx.x; j < j1; j += XC_BLOCK)
        sum += metric_eval(metric, stats[j] / nn, stats[m + j] / nn, stats[2 * m + j] / nn, stats[3 * m + j] / nn);
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int w = XC_BLOCK / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(XC_BLOCK) void state_unpack_kernel(int64_t m, const double *tpfp, const double *colsum,
                                                                double n_counted, int skip_tn, double *tp, double *fp,
                                                                double *fn, double *tn) {
    const int64_t j = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x;
    if (j >= m) return;
    tp[j] = tpfp[2 * j];
    fp[j] = tpfp[2 * j + 1];
    fn[j] = colsum[j] - tpfp[2 * j];
    tn[j] = skip_tn ? -1.0 : (n_counted - tpfp[2 * j + 1] - colsum[j]);
}

// ---- which labels are busy (stored in many rows)?  a strided sample of the stored entries is enough ----------
__global__ __launch_bounds__(XC_BLOCK) void label_hist_sampled_kernel(int64_t n_samples, int64_t stride, const int32_t *indices,
                                                                      int32_t *counts) {
    const int64_t step = (int64_t)gridDim.x * XC_BLOCK;
    for (int64_t t = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x; t < n_samples; t += step)
        atomicAdd(counts + indices[t * stride], 1);
}

// labels whose sampled count reaches `min_count`, as (label, count) pairs in arrival order (the caller sorts the few)
__global__ __launch_bounds__(XC_BLOCK) void label_busy_list_kernel(int64_t m, const int32_t *counts, int min_count, int cap,
                                                                   int32_t *list, int32_t *n_list) {
    const int64_t j = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x;
    if (j >= m) return;
    const int c = counts[j];
    if (c >= min_count) {
        const int o = atomicAdd(n_list, 1);
        if (o < cap) {
            list[2 * o] = (int32_t)j;
            list[2 * o + 1] = c;
        }
    }
}

// ---- row shards: one step of the overlapped mid-sweep exchange ----------------------------------------------
// `records` (float32) is this rank's working copy of the per-label records, `base` the same without the rank's own
// not-yet-published changes.  A step (a) folds in the OTHER ranks' part of the exchange issued one step earlier
// (`sum` = all-reduced changes, `mine` = this rank's contribution to it), then (b) publishes what this rank's rows
// changed since: it leaves it in `mine` and in `sum` (the buffer the next all-reduce runs on) and moves `base` up.
__global__ __launch_bounds__(XC_BLOCK) void exchange_step_kernel(int64_t n, float *records, float *base, float *sum,
                                                                 float *mine, int fold) {
    const int64_t stride = (int64_t)gridDim.x * XC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x; i < n; i += stride) {
        float r = records[i], b = base[i];
        if (fold) {
            const float others = sum[i] - mine[i];
            r += others;
            b += others;
        }
        const float own = r - b; // what this rank's rows changed since its last publication
        records[i] = r;
        base[i] = r;
        mine[i] = own;
        sum[i] = own;
    }
}

static unsigned long long *g_stamp_buffer = nullptr; // set by xc_debug_set_stamp_buffer
// one-shot HIP events recorded tightly around the NEXT sweep launch (xc_bca_time_next_sweep)
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
static thread_local int g_ev_span = 1; // sweep launches the pending event pair spans (xc_bca_time_span)
static int g_validate = 2;                           // xc_bca_set_validation: 2 = commit protocol
static int g_acc_delta = 1;                          // xc_bca_set_acc_delta: pipelined sweeps push changes instead of rebuilding acc
static float g_conflict_rel = 1.0f / 512.0f;         // xc_bca_set_tuning
static float g_hot_unpublished = 0.05f;              // xc_bca_set_tuning: share of the rows whose hot-label deltas may be unpublished

template <typename T, int CH, bool EXACT, bool HAS_ORDER, bool SHADOW, bool PACKED, bool HOT>
static void launch_sweep_one(const SweepParams<T> &P, hipStream_t st) {
    const int blocks = (P.n_waves + 3) / 4;
    if (g_ev_stop) {
        // start / stop events attached to the dispatch itself: the measured span is the
        // kernel, not the kernel plus the dispatch gap an event pair around it would add.  A sweep that row shards walk in
        // parts (xc_bca_time_span): the start event rides on the first part's launch, the stop event on the last one's
        const bool last = g_ev_span <= 1;
        hipExtLaunchKernelGGL((bca_sweep_csr_kernel<T, CH, EXACT, HAS_ORDER, SHADOW, PACKED, HOT>), dim3(blocks),
                              dim3(XC_BLOCK), 0, st, g_ev_start, last ? g_ev_stop : nullptr, 0, P);
        g_ev_start = nullptr;
        if (last) {
            g_ev_stop = nullptr;
            g_ev_span = 1;
        } else {
            --g_ev_span;
        }
    } else {
        hipLaunchKernelGGL((bca_sweep_csr_kernel<T, CH, EXACT, HAS_ORDER, SHADOW, PACKED, HOT>), dim3(blocks),
                           dim3(XC_BLOCK), 0, st, P);
    }
}

template <typename T, bool EXACT, bool HAS_ORDER, bool SHADOW, bool PACKED, bool HOT>
static void launch_sweep_impl(const SweepParams<T> &P, int ch, hipStream_t st) {
    switch (ch) {
    case 1: launch_sweep_one<T, 1, EXACT, HAS_ORDER, SHADOW, PACKED, HOT>(P, st); break;
    case 2: launch_sweep_one<T, 2, EXACT, HAS_ORDER, SHADOW, PACKED, HOT>(P, st); break;
    case 4: launch_sweep_one<T, 4, EXACT, HAS_ORDER, SHADOW, PACKED, HOT>(P, st); break;
    case 8: launch_sweep_one<T, 8, EXACT, HAS_ORDER, SHADOW, PACKED, HOT>(P, st); break;
    default: launch_sweep_one<T, 16, EXACT, HAS_ORDER, SHADOW, PACKED, HOT>(P, st); break;
    }
}

template <typename T, bool PACKED>
static void launch_sweep_mode(const SweepParams<T> &P, int ch, hipStream_t st) {
    const bool exact = P.n_waves == 1 && !P.ctrl;
    const bool shadow = !exact && !P.greedy && P.shadow != nullptr;
    const bool hot = PACKED && shadow && P.hot_labels != nullptr && P.acc != nullptr;
    if (P.order) {
        if (exact) launch_sweep_impl<T, true, true, false, PACKED, false>(P, ch, st);
        else if (hot) launch_sweep_impl<T, false, true, true, PACKED, PACKED>(P, ch, st);
        else if (shadow) launch_sweep_impl<T, false, true, true, PACKED, false>(P, ch, st);
        else launch_sweep_impl<T, false, true, false, PACKED, false>(P, ch, st);
    } else {
        if (exact) launch_sweep_impl<T, true, false, false, PACKED, false>(P, ch, st);
        else if (hot) launch_sweep_impl<T, false, false, true, PACKED, PACKED>(P, ch, st);
        else if (shadow) launch_sweep_impl<T, false, false, true, PACKED, false>(P, ch, st);
        else launch_sweep_impl<T, false, false, false, PACKED, false>(P, ch, st);
    }
}

// n_waves == 1 is the sequential mode that must reproduce the reference's
// trajectory: it keeps the reference's divisions and reads the float64 records.
// The float32 shadow is read by the concurrent, non-greedy sweep only; the packed row
// stream exists for float32 scores and non-greedy sweeps.
static void launch_sweep(const SweepParams<float> &P, int ch, hipStream_t st) {
    const bool exact = P.n_waves == 1 && !P.ctrl; // the packed stream carries s in float32
    if (P.packed && !P.greedy && !exact) launch_sweep_mode<float, true>(P, ch, st);
    else launch_sweep_mode<float, false>(P, ch, st);
}

static void launch_sweep(const SweepParams<double> &P, int ch, hipStream_t st) {
    launch_sweep_mode<double, false>(P, ch, st);
}

// Row shards with acc_delta: a rank's float64 records hold the statistics all ranks agreed on at the last
// boundary (`base`) plus what ITS rows changed since.  d <- records - base (exact: sums of float32 values in
// float64), d[m2] <- this rank's changed-row count; after the all-reduce of d, records = base <- base + d.
__global__ __launch_bounds__(XC_BLOCK) void delta_pack_kernel(int64_t m2, const double *tpfp, const double *base,
                                                              const double *count, double *d) {
    for (int64_t i = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x; i < m2; i += (int64_t)gridDim.x * XC_BLOCK)
        d[i] = tpfp[i] - base[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) d[m2] = count[0];
}

__global__ __launch_bounds__(XC_BLOCK) void delta_unpack_kernel(int64_t m2, double *tpfp, double *base, double *count,
                                                                const double *d) {
    for (int64_t i = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x; i < m2; i += (int64_t)gridDim.x * XC_BLOCK) {
        const double v = base[i] + d[i];
        tpfp[i] = v;
        base[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) count[0] = d[m2];
}

static int grid_for(int64_t n_items) {
    int64_t b = (n_items + XC_BLOCK - 1) / XC_BLOCK;
    const int64_t cap = 256 * 8 * 4; // grid-stride beyond a few blocks per CU
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

} // namespace xc

extern "C" {

int xc_bca_gather_pred_eta(int64_t n, const int32_t *indptr, const int32_t *indices, const void *data,
                           int dtype, const int32_t *pred_indices, int k, void *pred_eta, uint8_t *sel,
                           int32_t *orphans, void *stream) {
    if (n < 0 || k < 1 || (n > 0 && (!indptr || !pred_indices || !pred_eta)))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_gather_pred_eta: bad argument");
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_gather_pred_eta: unknown dtype %d", dtype);
    if (n == 0) return XC_OK;
    const int64_t n_k = n * k;
    const int blocks = (int)((n_k + XC_BLOCK - 1) / XC_BLOCK);
    hipStream_t st = xc::as_stream(stream);
    if (dtype == XC_F32)
        hipLaunchKernelGGL((xc::gather_pred_eta_kernel<float>), dim3(blocks), dim3(XC_BLOCK), 0, st, n_k, k, indptr, indices,
                           static_cast<const float *>(data), pred_indices, static_cast<float *>(pred_eta), sel, orphans);
    else
        hipLaunchKernelGGL((xc::gather_pred_eta_kernel<double>), dim3(blocks), dim3(XC_BLOCK), 0, st, n_k, k, indptr, indices,
                           static_cast<const double *>(data), pred_indices, static_cast<double *>(pred_eta), sel, orphans);
    XC_CHECK_LAUNCH("gather_pred_eta_kernel");
    return XC_OK;
}

int xc_bca_colsum_csr(int64_t nnz, const int32_t *indices, const void *data, int dtype, double *colsum,
                      void *stream) {
    if (nnz < 0 || (nnz > 0 && (!indices || !data || !colsum)))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_colsum_csr: bad argument");
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_colsum_csr: unknown dtype %d", dtype);
    if (nnz == 0) return XC_OK;
    hipStream_t st = xc::as_stream(stream);
    const int blocks = xc::grid_for(nnz);
    if (dtype == XC_F32)
        hipLaunchKernelGGL((xc::colsum_csr_kernel<float>), dim3(blocks), dim3(XC_BLOCK), 0, st, nnz, indices,
                           static_cast<const float *>(data), colsum);
    else
        hipLaunchKernelGGL((xc::colsum_csr_kernel<double>), dim3(blocks), dim3(XC_BLOCK), 0, st, nnz, indices,
                           static_cast<const double *>(data), colsum);
    XC_CHECK_LAUNCH("colsum_csr_kernel");
    return XC_OK;
}

int xc_bca_expand_colsum(int64_t nnz, const int32_t *indices, const double *colsum, double *s_entry, void *stream) {
    if (nnz < 0 || (nnz > 0 && (!indices || !colsum || !s_entry)))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_expand_colsum: bad argument");
    if (nnz == 0) return XC_OK;
    hipLaunchKernelGGL(xc::expand_colsum_kernel, dim3(xc::grid_for(nnz)), dim3(XC_BLOCK), 0, xc::as_stream(stream), nnz,
                       indices, colsum, s_entry);
    XC_CHECK_LAUNCH("expand_colsum_kernel");
    return XC_OK;
}

int xc_bca_pack_rows(int64_t nnz, const int32_t *indices, const float *data, const uint8_t *sel,
                     const double *s_entry, const uint8_t *hot_slot, void *packed, void *stream) {
    if (nnz < 0 || (nnz > 0 && (!indices || !data || !sel || !s_entry || !packed)))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_pack_rows: bad argument");
    if (nnz == 0) return XC_OK;
    hipLaunchKernelGGL(xc::pack_rows_kernel, dim3(xc::grid_for(nnz)), dim3(XC_BLOCK), 0, xc::as_stream(stream), nnz, indices,
                       data, sel, s_entry, hot_slot, static_cast<xc::pack3_t *>(packed));
    XC_CHECK_LAUNCH("pack_rows_kernel");
    return XC_OK;
}

int xc_bca_pack_rows_from_colsum(int64_t nnz, const int32_t *indices, const float *data, const uint8_t *sel,
                                 const double *colsum, const uint8_t *hot_slot, void *packed, void *stream) {
    if (nnz < 0 || (nnz > 0 && (!indices || !data || !sel || !colsum || !packed)))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_pack_rows_from_colsum: bad argument");
    if (nnz == 0) return XC_OK;
    hipLaunchKernelGGL(xc::pack_rows_colsum_kernel, dim3(xc::grid_for(nnz)), dim3(XC_BLOCK), 0, xc::as_stream(stream), nnz,
                       indices, data, sel, colsum, hot_slot, static_cast<xc::pack3_t *>(packed));
    XC_CHECK_LAUNCH("pack_rows_colsum_kernel");
    return XC_OK;
}

int xc_bca_accumulate_pred(int64_t n_k, const int32_t *pred_indices, const void *pred_eta, int dtype,
                           double *acc, void *stream) {
    if (n_k < 0 || (n_k > 0 && (!pred_indices || !pred_eta || !acc)))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_accumulate_pred: bad argument");
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_accumulate_pred: unknown dtype %d", dtype);
    if (n_k == 0) return XC_OK;
    hipStream_t st = xc::as_stream(stream);
    const int blocks = xc::grid_for(n_k);
    if (dtype == XC_F32)
        hipLaunchKernelGGL((xc::accumulate_pred_kernel<float>), dim3(blocks), dim3(XC_BLOCK), 0, st, n_k, pred_indices,
                           static_cast<const float *>(pred_eta), acc);
    else
        hipLaunchKernelGGL((xc::accumulate_pred_kernel<double>), dim3(blocks), dim3(XC_BLOCK), 0, st, n_k, pred_indices,
                           static_cast<const double *>(pred_eta), acc);
    XC_CHECK_LAUNCH("accumulate_pred_kernel");
    return XC_OK;
}

static int commit_utility_impl(int64_t m, int64_t n_norm, double n_counted, double *acc, int clear_acc, double *tpfp,
                               float *shadow, const double *colsum, const xc_metric *metric_host, int skip_tn,
                               double *partials, const double *ctrl, void *stream) {
    if (m < 0 || n_norm < 1 || !tpfp || !colsum || !metric_host || !partials)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_commit_utility: bad argument");
    if (metric_host->base < 0 || metric_host->base >= XC_M_COUNT)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_commit_utility: unknown metric %d", metric_host->base);
    hipLaunchKernelGGL(xc::commit_utility_kernel, dim3(XC_UTILITY_PARTIALS), dim3(XC_BLOCK), 0, xc::as_stream(stream), m,
                       (double)n_norm, n_counted, acc, clear_acc, tpfp, shadow, colsum, *metric_host, skip_tn, partials,
                       ctrl);
    XC_CHECK_LAUNCH("commit_utility_kernel");
    return XC_OK;
}

int xc_bca_commit_utility(int64_t m, int64_t n_norm, double n_counted, double *acc, int clear_acc, double *tpfp,
                          float *shadow, const double *colsum, const xc_metric *metric_host, int skip_tn,
                          double *partials, void *stream) {
    return commit_utility_impl(m, n_norm, n_counted, acc, clear_acc, tpfp, shadow, colsum, metric_host, skip_tn,
                               partials, nullptr, stream);
}

int xc_utility_vectors(int64_t m, int64_t n_norm, const double *stats, const xc_metric *metric_host,
                       double *partials, void *stream) {
    if (m < 0 || n_norm < 1 || !stats || !metric_host || !partials)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_utility_vectors: bad argument");
    if (metric_host->base < 0 || metric_host->base >= XC_M_COUNT)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_utility_vectors: unknown metric %d", metric_host->base);
    hipLaunchKernelGGL(xc::utility_vectors_kernel, dim3(XC_UTILITY_PARTIALS), dim3(XC_BLOCK), 0, xc::as_stream(stream), m,
                       (double)n_norm, stats, *metric_host, partials);
    XC_CHECK_LAUNCH("utility_vectors_kernel");
    return XC_OK;
}

int xc_utility_finish_host(const double *partials, double *out_host, double *out_extra_host, void *stream) {
    if (!partials || !out_host) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_utility_finish_host: NULL pointer");
    // pinned staging buffer, allocated once per host thread: a D2H into pageable
    // memory goes through the runtime's own staging copy and costs several us more
    static thread_local double *buf = nullptr;
    if (!buf) XC_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&buf), sizeof(double) * (XC_UTILITY_PARTIALS + 1), hipHostMallocDefault));
    hipStream_t st = xc::as_stream(stream);
    XC_HIP_TRY(hipMemcpyAsync(buf, partials, sizeof(double) * (XC_UTILITY_PARTIALS + 1), hipMemcpyDeviceToHost, st));
    XC_HIP_TRY(hipStreamSynchronize(st));
    // the fixed pairwise tree bca_boundary_finish_kernel uses, so both give the same bits
    for (int stride = XC_UTILITY_PARTIALS / 2; stride > 0; stride >>= 1)
        for (int i = 0; i < stride; ++i) buf[i] += buf[i + stride];
    *out_host = buf[0];
    if (out_extra_host) *out_extra_host = buf[XC_UTILITY_PARTIALS];
    return XC_OK;
}

static int sweep_csr_impl(int64_t n_order, const int32_t *order, int64_t n_norm, const int32_t *indptr,
                          const int32_t *indices, const void *data, int dtype, int max_row_nnz,
                          int32_t *pred_indices, void *pred_eta, uint8_t *sel, const int32_t *orphans, int k,
                          int64_t m, double *tpfp, float *shadow, double *colsum, const double *s_entry, void *packed,
                          const int32_t *hot_labels, double *acc, const xc_metric *metric_host, int maximize, int greedy,
                          int skip_tn, int n_waves, int64_t *changed, const double *ctrl, int acc_delta, void *stream) {
    if (n_order < 0 || n_norm < 1 || m < 1 || !indptr || !pred_indices || !pred_eta || !sel || !tpfp || !colsum ||
        !metric_host)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_sweep_csr: NULL pointer or bad size");
    if (!greedy && !s_entry && !(packed && n_waves > 1))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_sweep_csr: s_entry is required unless greedy or a concurrent sweep over the packed stream");
    if (m > (int64_t)(0xFFFFFFFFu / 16))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_sweep_csr: m too large for 32-bit record offsets");
    if (packed && m > (int64_t)XC_PACK_COL_MASK + 1)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_sweep_csr: the packed row stream holds 25-bit column ids");
    if (k < 1 || k > XC_MAX_K) return xc::fail_arg(XC_ERR_K_RANGE, "xc_bca_sweep_csr: k=%d outside 1..%d", k, XC_MAX_K);
    if (dtype != XC_F32 && dtype != XC_F64) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_sweep_csr: unknown dtype %d", dtype);
    if (metric_host->base < 0 || metric_host->base >= XC_M_COUNT)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_sweep_csr: unknown metric %d", metric_host->base);
    const int ch = xc::chunks_for(max_row_nnz);
    if (ch == 0)
        return xc::fail_arg(XC_ERR_ROW_TOO_LONG, "xc_bca_sweep_csr: a row holds %d entries, limit %d", max_row_nnz, XC_MAX_ROW_NNZ);
    if (n_waves < 1) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_sweep_csr: n_waves must be >= 1");
    if (n_order == 0) return XC_OK;
    if (n_waves > n_order) n_waves = (int)n_order;
    hipStream_t st = xc::as_stream(stream);
    xc_metric fast = *metric_host; // psi(x / n; eps, k) = psi(x; eps * n, k * n)
    fast.epsilon *= (double)n_norm;
    fast.kf *= (double)n_norm;
    if (dtype == XC_F32) {
        xc::SweepParams<float> P{n_order, order, indptr, indices, static_cast<const float *>(data), pred_indices,
                                 static_cast<float *>(pred_eta), sel, orphans, k, tpfp, shadow, colsum, greedy ? nullptr : s_entry,
                                 static_cast<xc::pack3_t *>(packed), packed ? hot_labels : nullptr, acc, acc_delta, m,
                                 (unsigned)(m * 16), *metric_host, fast, (double)n_norm,
                                 (double)n_norm, maximize, greedy, skip_tn, n_waves, xc::g_validate, xc::g_hot_unpublished, xc::g_conflict_rel,
                                 reinterpret_cast<unsigned long long *>(changed), xc::g_stamp_buffer, ctrl};
        xc::launch_sweep(P, ch, st);
    } else {
        xc::SweepParams<double> P{n_order, order, indptr, indices, static_cast<const double *>(data), pred_indices,
                                  static_cast<double *>(pred_eta), sel, orphans, k, tpfp, shadow, colsum, greedy ? nullptr : s_entry,
                                  nullptr, nullptr, acc, acc_delta, m, (unsigned)(m * 16), *metric_host, fast, (double)n_norm,
                                  (double)n_norm, maximize, greedy, skip_tn, n_waves, xc::g_validate, xc::g_hot_unpublished, xc::g_conflict_rel,
                                  reinterpret_cast<unsigned long long *>(changed), xc::g_stamp_buffer, ctrl};
        xc::launch_sweep(P, ch, st);
    }
    XC_CHECK_LAUNCH("bca_sweep_csr_kernel");
    return XC_OK;
}

int xc_bca_sweep_csr(int64_t n_order, const int32_t *order, int64_t n_norm, const int32_t *indptr,
                     const int32_t *indices, const void *data, int dtype, int max_row_nnz,
                     int32_t *pred_indices, void *pred_eta, uint8_t *sel, const int32_t *orphans, int k,
                     int64_t m, double *tpfp, float *shadow, double *colsum, const double *s_entry, void *packed,
                     const int32_t *hot_labels, double *acc, const xc_metric *metric_host, int maximize, int greedy,
                     int skip_tn, int n_waves, int64_t *changed, void *stream) {
    return sweep_csr_impl(n_order, order, n_norm, indptr, indices, data, dtype, max_row_nnz, pred_indices, pred_eta,
                          sel, orphans, k, m, tpfp, shadow, colsum, s_entry, packed, hot_labels, acc, metric_host,
                          maximize, greedy, skip_tn, n_waves, changed, nullptr, 0, stream);
}

// ---- measurement helpers (bench.py): HIP events owned by the library ---------------
int xc_event_create(void **ev) {
    if (!ev) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_event_create: NULL");
    hipEvent_t e;
    XC_HIP_TRY(hipEventCreate(&e));
    *ev = e;
    return XC_OK;
}

int xc_event_destroy(void *ev) {
    if (ev) XC_HIP_TRY(hipEventDestroy(static_cast<hipEvent_t>(ev)));
    return XC_OK;
}

int xc_event_elapsed_ms(void *start, void *stop, float *ms_host) {
    if (!start || !stop || !ms_host) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_event_elapsed_ms: NULL");
    XC_HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
    XC_HIP_TRY(hipEventElapsedTime(ms_host, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
    return XC_OK;
}

int xc_bca_time_next_sweep(void *start, void *stop) {
    xc::g_ev_start = static_cast<hipEvent_t>(start);
    xc::g_ev_stop = static_cast<hipEvent_t>(stop);
    xc::g_ev_span = 1;
    return XC_OK;
}

int xc_bca_time_span(int launches) {
    if (launches < 1) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_time_span: launches < 1");
    if (xc::g_ev_stop) xc::g_ev_span = launches;
    return XC_OK;
}

int xc_bca_set_tuning(double conflict_rel, double hot_unpublished) {
    if (conflict_rel >= 0.0) xc::g_conflict_rel = (float)conflict_rel;
    if (hot_unpublished >= 0.0) xc::g_hot_unpublished = (float)hot_unpublished;
    return XC_OK;
}

int xc_bca_set_validation(int mode) {
    xc::g_validate = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
    return XC_OK;
}

int xc_bca_set_acc_delta(int on) {
    xc::g_acc_delta = on ? 1 : 0;
    return XC_OK;
}


// Diagnostic builds (-DXC_STAMPS) only: device buffer of 8 uint64 phase-cycle sums.
// Not part of include/xcolumns_amd.h; the shipped library ignores the pointer.
int xc_debug_set_stamp_buffer(void *buf) {
    xc::g_stamp_buffer = static_cast<unsigned long long *>(buf);
    return XC_OK;
}

// ---- plan: the per-run constants bound once, two short calls per sweep --------------
struct xc_bca_plan_s {
    int64_t n, m, n_total;
    const int32_t *indptr, *indices;
    const void *data;
    int dtype, max_row_nnz, k;
    int32_t *pred_indices;
    void *pred_eta;
    uint8_t *sel;
    double *tpfp;
    float *shadow;
    double *colsum;
    const double *s_entry;
    void *packed;
    const int32_t *hot_labels;
    double *acc, *partials;
    xc_metric gain_metric, utility_metric;
    int maximize, skip_tn;
    int last_delta; // did the last pipelined sweep leave its changes in tpfp (acc_delta) rather than rebuild acc
};

// A pipelined sweep leaves the statistics of the new prediction either rebuilt from scratch in `acc` (every row
// adds its k labels: one 16-byte atomic each, 19 % of a converged sweep at 1 M x 500 K) or, with the commit
// protocol on float32 shadow records, as the committed changes pushed into the float64 records -- exact float64
// sums of float32 values either way, the second with atomics for the rows that CHANGE only.
static int plan_delta(const xc_bca_plan_s *p) {
    // (without a float32 shadow the commit protocol's returning atomics act on tpfp itself: nothing more to push)
    return xc::g_acc_delta && xc::g_validate == 2 && p->acc != nullptr;
}

int xc_bca_plan_create(void **plan, int64_t n, int64_t m, int64_t n_total, const int32_t *indptr,
                       const int32_t *indices, const void *data, int dtype, int max_row_nnz, int k,
                       int32_t *pred_indices, void *pred_eta, uint8_t *sel, double *tpfp, float *shadow,
                       double *colsum, const double *s_entry, void *packed, const int32_t *hot_labels, double *acc,
                       double *partials, const xc_metric *gain_metric, const xc_metric *utility_metric, int maximize,
                       int skip_tn) {
    if (!plan || !gain_metric || !utility_metric) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_plan_create: NULL pointer");
    xc_bca_plan_s *p = new xc_bca_plan_s{n, m, n_total, indptr, indices, data, dtype, max_row_nnz, k, pred_indices,
                                         pred_eta, sel, tpfp, shadow, colsum, s_entry, packed, hot_labels, acc, partials, *gain_metric,
                                         *utility_metric, maximize, skip_tn, 0};
    *plan = p;
    return XC_OK;
}

int xc_bca_plan_destroy(void *plan) {
    delete static_cast<xc_bca_plan_s *>(plan);
    return XC_OK;
}

int xc_bca_plan_delta(void *plan) { return plan ? plan_delta(static_cast<const xc_bca_plan_s *>(plan)) : 0; }

int xc_bca_delta_pack(int64_t m2, const double *tpfp, const double *base, const double *count, double *d, void *stream) {
    if (m2 < 1 || !tpfp || !base || !count || !d) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_delta_pack: bad argument");
    hipLaunchKernelGGL(xc::delta_pack_kernel, dim3(xc::grid_for(m2)), dim3(XC_BLOCK), 0, xc::as_stream(stream), m2, tpfp, base,
                       count, d);
    XC_CHECK_LAUNCH("delta_pack_kernel");
    return XC_OK;
}

int xc_bca_delta_unpack(int64_t m2, double *tpfp, double *base, double *count, const double *d, void *stream) {
    if (m2 < 1 || !tpfp || !base || !count || !d) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_delta_unpack: bad argument");
    hipLaunchKernelGGL(xc::delta_unpack_kernel, dim3(xc::grid_for(m2)), dim3(XC_BLOCK), 0, xc::as_stream(stream), m2, tpfp, base,
                       count, d);
    XC_CHECK_LAUNCH("delta_unpack_kernel");
    return XC_OK;
}

int xc_bca_plan_sweep(void *plan, const int32_t *order, int64_t n_order, const int32_t *orphans, int greedy,
                      int n_waves, int with_acc, int use_packed, int64_t *changed, void *stream) {
    if (!plan) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_plan_sweep: NULL plan");
    const xc_bca_plan_s *p = static_cast<const xc_bca_plan_s *>(plan);
    return xc_bca_sweep_csr(n_order, order, p->n_total, p->indptr, p->indices, p->data, p->dtype, p->max_row_nnz,
                            p->pred_indices, p->pred_eta, p->sel, orphans, p->k, p->m, p->tpfp, p->shadow, p->colsum,
                            p->s_entry, use_packed ? p->packed : nullptr, p->hot_labels, with_acc ? p->acc : nullptr,
                            &p->gain_metric, p->maximize, greedy, p->skip_tn, n_waves, changed, stream);
}

int xc_bca_plan_boundary(void *plan, int64_t n_norm_utility, double n_counted, int commit, int skip_tn,
                         double *out_sum_host, double *out_extra_host, void *stream) {
    if (!plan) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_plan_boundary: NULL plan");
    const xc_bca_plan_s *p = static_cast<const xc_bca_plan_s *>(plan);
    int rc = xc_bca_commit_utility(p->m, n_norm_utility, n_counted, commit ? p->acc : nullptr, 1, p->tpfp, p->shadow,
                                   p->colsum, &p->utility_metric, skip_tn, p->partials, stream);
    if (rc) return rc;
    return xc_utility_finish_host(p->partials, out_sum_host, out_extra_host, stream);
}

// ---- the sweep loop without a host round trip per iteration ---------------------------
int xc_bca_pipeline_begin(double *ctrl, double old_utility_sum, double tolerance, double divisor, int maximize,
                          double policy_num, int world, int min_waves, int max_waves, int fixed_waves, int first_waves,
                          int exact_below, void *stream) {
    if (!ctrl || max_waves < 1 || first_waves < 1 || divisor <= 0.0)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_pipeline_begin: bad argument");
    hipLaunchKernelGGL(xc::bca_ctrl_init_kernel, dim3(1), dim3(XC_WAVE), 0, xc::as_stream(stream), ctrl, old_utility_sum,
                       tolerance, divisor, maximize, policy_num, (double)(world < 1 ? 1 : world), (double)min_waves,
                       (double)max_waves, (double)fixed_waves, (double)first_waves, (double)(exact_below < 0 ? 0 : exact_below));
    XC_CHECK_LAUNCH("bca_ctrl_init_kernel");
    return XC_OK;
}

int xc_bca_plan_sweep_pipelined(void *plan, const int32_t *order, int64_t first, int64_t count, int use_packed,
                                int max_waves, const double *ctrl, void *stream) {
    if (!plan || !ctrl || max_waves < 2)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_plan_sweep_pipelined: bad argument");
    xc_bca_plan_s *p = static_cast<xc_bca_plan_s *>(plan);
    if (first < 0 || count < 0 || first + count > p->n || (!order && (first != 0 || count != p->n)))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_plan_sweep_pipelined: bad segment (a partial segment needs `order`)");
    // the boundary that follows must read the statistics where this sweep leaves them
    p->last_delta = plan_delta(p);
    return sweep_csr_impl(count, order ? order + first : nullptr, p->n_total, p->indptr, p->indices, p->data, p->dtype, p->max_row_nnz,
                          p->pred_indices, p->pred_eta, p->sel, nullptr, p->k, p->m, p->tpfp, p->shadow, p->colsum,
                          p->s_entry, use_packed ? p->packed : nullptr, p->hot_labels, p->acc, &p->gain_metric,
                          p->maximize, 0, p->skip_tn, max_waves, nullptr, ctrl, p->last_delta, stream);
}

int xc_bca_plan_boundary_pipelined(void *plan, int64_t n_norm_utility, double n_counted, int skip_tn, double *ctrl,
                                   int slot, double *host_ring, double seq, void *stream) {
    if (!plan || !ctrl || slot < 0 || slot >= XC_CTRL_RING_SLOTS)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_plan_boundary_pipelined: bad argument");
    const xc_bca_plan_s *p = static_cast<const xc_bca_plan_s *>(plan);
    int rc = commit_utility_impl(p->m, n_norm_utility, n_counted, p->acc, p->last_delta ? 2 : 1, p->tpfp, p->shadow, p->colsum,
                                 &p->utility_metric, skip_tn, p->partials, ctrl, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(xc::bca_boundary_finish_kernel, dim3(1), dim3(XC_UTILITY_PARTIALS / 2), 0, xc::as_stream(stream),
                       p->partials, ctrl, slot, host_ring, seq);
    XC_CHECK_LAUNCH("bca_boundary_finish_kernel");
    return XC_OK;
}

// Spin (GIL-free under ctypes) until ring slot `slot` of the pinned host ring carries sequence
// number `seq`; XC_ERR_BAD_ARG after timeout_ms without it (a stuck stream, not a slow one).
int xc_bca_ring_wait(const double *host_ring, int slot, double seq, double timeout_ms, void *stream) {
    if (!host_ring || slot < 0 || slot >= XC_CTRL_RING_SLOTS)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ring_wait: bad argument");
    const volatile double *r = host_ring + XC_CTRL_RING_STRIDE * slot;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (r[4] != seq) {
        __builtin_ia32_pause();
        if ((++spins & 0xFFFFu) == 0) { // about once a millisecond: is the stream still alive?
            const hipError_t q = hipStreamQuery(xc::as_stream(stream));
            if (q != hipSuccess && q != hipErrorNotReady) return xc::fail_hip(q, "xc_bca_ring_wait: stream");
            if (q == hipSuccess && r[4] != seq)
                return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ring_wait: the stream drained without boundary %.0f", seq);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms > timeout_ms) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_ring_wait: no result after %.0f ms", timeout_ms);
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return XC_OK;
}

int xc_event_synchronize(void *event) {
    if (!event) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_event_synchronize: NULL");
    XC_HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(event)));
    return XC_OK;
}

int xc_host_alloc_pinned(void **ptr, int64_t bytes) {
    if (!ptr || bytes <= 0) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_host_alloc_pinned: bad argument");
    XC_HIP_TRY(hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault));
    return XC_OK;
}

int xc_host_free_pinned(void *ptr) {
    if (ptr) XC_HIP_TRY(hipHostFree(ptr));
    return XC_OK;
}

int xc_label_busy_list(int64_t nnz, const int32_t *indices, int64_t stride, int64_t m, int min_count, int cap,
                       int32_t *counts, int32_t *list, int32_t *n_list, void *stream) {
    if (nnz < 0 || stride < 1 || m < 1 || cap < 1 || (nnz > 0 && !indices) || !counts || !list || !n_list)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_label_busy_list: bad argument");
    hipStream_t st = xc::as_stream(stream);
    XC_HIP_TRY(hipMemsetAsync(counts, 0, (size_t)m * 4, st));
    XC_HIP_TRY(hipMemsetAsync(n_list, 0, 4, st));
    const int64_t n_samples = nnz / stride;
    if (n_samples > 0)
        hipLaunchKernelGGL(xc::label_hist_sampled_kernel, dim3(xc::grid_for(n_samples)), dim3(XC_BLOCK), 0, st, n_samples, stride,
                           indices, counts);
    hipLaunchKernelGGL(xc::label_busy_list_kernel, dim3((unsigned)((m + XC_BLOCK - 1) / XC_BLOCK)), dim3(XC_BLOCK), 0, st, m,
                       counts, min_count, cap, list, n_list);
    XC_CHECK_LAUNCH("label_busy_list_kernel");
    return XC_OK;
}

int xc_bca_exchange_step(int64_t n, float *records, float *base, float *sum, float *mine, int fold, void *stream) {
    if (n < 0 || (n > 0 && (!records || !base || !sum || !mine)))
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_exchange_step: bad argument");
    if (n == 0) return XC_OK;
    hipLaunchKernelGGL(xc::exchange_step_kernel, dim3(xc::grid_for(n)), dim3(XC_BLOCK), 0, xc::as_stream(stream), n, records,
                       base, sum, mine, fold);
    XC_CHECK_LAUNCH("exchange_step_kernel");
    return XC_OK;
}

int xc_bca_state_unpack(int64_t m, const double *tpfp, const double *colsum, double n_counted, int skip_tn,
                        double *tp, double *fp, double *fn, double *tn, void *stream) {
    if (m < 0 || !tpfp || !colsum || !tp || !fp || !fn || !tn)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_bca_state_unpack: bad argument");
    if (m == 0) return XC_OK;
    const int blocks = (int)((m + XC_BLOCK - 1) / XC_BLOCK);
    hipLaunchKernelGGL(xc::state_unpack_kernel, dim3(blocks), dim3(XC_BLOCK), 0, xc::as_stream(stream), m, tpfp, colsum,
                       n_counted, skip_tn, tp, fp, fn, tn);
    XC_CHECK_LAUNCH("state_unpack_kernel");
    return XC_OK;
}

} // extern "C"

/*
 * ordered_sim.c -- STUDY (test infrastructure, CPU only): how far can a BCA sweep be parallelised
 * while keeping the reference's visiting-order semantics
 * (/root/reference/xcolumns/block_coordinate.py:448-463: row i + 1 sees row i)?
 *
 * Pass 1 runs the sequential sweep (macro F-beta, skip_tn, the in-register form of
 * xcolumns_amd/csrc/xc_bca.hip) and records, per order position, the labels the row changed and the
 * deltas.  Pass 2 replays the trace under the "certified prefix" rule of the ordered sweep: rows
 * [base, base + window) are scored on the statistics as committed up to `base`; a row is CERTAIN when no
 * earlier uncommitted row can change its decision, all rows before the first uncertain row commit in one
 * round, the first uncertain row becomes the next base (exact by definition).  It counts the rounds.
 *
 *   mode 0  read/write sets: uncertain = an earlier uncommitted row changed one of the row's candidates
 *   mode 1  as 0, but labels flagged `boxed` are handled by interval arithmetic: the record lies in
 *           [committed - pending_minus, committed + pending_plus] (pending = changes of the uncommitted
 *           rows before the row, or of the whole window with `superset`), and the row is certain when
 *           its top-k set is the same for every record in the boxes
 *   (mode 1 with every label boxed = decision-level certification for all labels)
 *
 * gcc -O2 -ffp-contract=off -shared -fPIC ordered_sim.c -o _build/libordered_sim.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXR 1024
#define MAXK 64

typedef struct {
    double eps, beta, nn;
} sim_metric;

static inline double fbeta(const sim_metric *mt, double tp, double fp, double fn) {
    const double b2 = mt->beta * mt->beta;
    return (1.0 + b2) * tp / ((b2 * (tp + fp)) + tp + fn + mt->eps);
}

/* gain of predicting the label for this row, statistics WITHOUT the row: tpc, fpc, fn (= s' - tpc) */
static inline double gain(const sim_metric *mt, double tpc, double fpc, double scc, double ed, double omd) {
    const double nn = mt->nn;
    const double fn = scc - tpc;
    return fbeta(mt, (tpc + ed) / nn, (fpc + omd) / nn, fn / nn) - fbeta(mt, tpc / nn, fpc / nn, (fn + ed) / nn);
}
/* interval form: F is increasing in tp and decreasing in fp (tp + fn is constant) */
static inline void gain_box(const sim_metric *mt, double tl, double th, double fl, double fh, double scc, double ed,
                            double omd, double *glo, double *ghi) {
    const double nn = mt->nn;
    const double fn_l = scc - th, fn_h = scc - tl; /* fn decreases with tp */
    *ghi = fbeta(mt, (th + ed) / nn, (fl + omd) / nn, fn_l / nn) - fbeta(mt, tl / nn, fh / nn, (fn_h + ed) / nn);
    *glo = fbeta(mt, (tl + ed) / nn, (fh + omd) / nn, fn_h / nn) - fbeta(mt, th / nn, fl / nn, (fn_l + ed) / nn);
}

static int topk_select(const double *g, int r, int k, uint8_t *in_new) {
    /* larger gain first, ties -> lower position */
    for (int p = 0; p < r; ++p) {
        int rank = 0;
        for (int q = 0; q < r; ++q)
            if (g[q] > g[p] || (g[q] == g[p] && q < p)) ++rank;
        in_new[p] = rank < k;
    }
    return 0;
}

/* trace layout per position: ntr[pos] changes at trace[pos * 2k ...]: label, dtp, dfp */
typedef struct {
    int32_t label;
    double dtp, dfp;
} change_t;

/* Pass 1: sequential sweep.  sel (uint8 per stored entry) and tp/fp are updated in place. */
int64_t sim_sequential(int64_t n_order, const int64_t *order, const int32_t *indptr, const int32_t *indices,
                       const float *data, uint8_t *sel, int k, double *tp, double *fp, const double *colsum,
                       const sim_metric *mt, int32_t *ntr, change_t *trace) {
    double g[MAXR];
    uint8_t in_new[MAXR];
    int64_t changed = 0;
    for (int64_t pos = 0; pos < n_order; ++pos) {
        const int64_t i = order[pos];
        const int s0 = indptr[i], r = indptr[i + 1] - s0;
        for (int p = 0; p < r; ++p) {
            const int j = indices[s0 + p];
            const float e = data[s0 + p];
            const double ed = (double)e, omd = (double)(1.0f - e);
            double tpc = tp[j], fpc = fp[j];
            if (sel[s0 + p]) {
                tpc -= ed;
                fpc -= omd;
            }
            g[p] = gain(mt, tpc, fpc, colsum[j] - ed, ed, omd);
        }
        topk_select(g, r, k, in_new);
        int nc = 0;
        for (int p = 0; p < r; ++p)
            if (in_new[p] != sel[s0 + p]) {
                const int j = indices[s0 + p];
                const float e = data[s0 + p];
                const double sgn = in_new[p] ? 1.0 : -1.0;
                change_t *c = trace + pos * 2 * k + nc++;
                c->label = j;
                c->dtp = sgn * (double)e;
                c->dfp = sgn * (double)(1.0f - e);
                tp[j] += c->dtp;
                fp[j] += c->dfp;
                sel[s0 + p] = in_new[p];
            }
        ntr[pos] = nc;
        changed += nc > 0;
    }
    return changed;
}

/* Pass 2: replay under the certified-prefix rule.  tp/fp/sel: the state at the START of the sweep
 * (advanced here as rows commit).  Returns the number of rounds; stats[0] = rows that ended a round because
 * of a non-boxed hit, stats[1] = because of a box, stats[2] = because the window was full,
 * stats[3] = boxed rows that were evaluated with a non-degenerate box and found certain. */
int64_t sim_replay(int64_t n_order, const int64_t *order, const int32_t *indptr, const int32_t *indices,
                   const float *data, uint8_t *sel, int k, double *tp, double *fp, const double *colsum,
                   const sim_metric *mt, const int32_t *ntr, const change_t *trace, int64_t m, int mode,
                   const uint8_t *boxed, int superset, int64_t window, int64_t *stats, int32_t *seg_len_hist /* [32] log2 */) {
    int64_t *lastw = (int64_t *)malloc(sizeof(int64_t) * m); /* last uncommitted/any writer position per label */
    for (int64_t j = 0; j < m; ++j) lastw[j] = -1;
    /* pending sums per label (boxed labels only matter) */
    double *pp_tp = (double *)calloc(m, sizeof(double)), *pn_tp = (double *)calloc(m, sizeof(double));
    double *pp_fp = (double *)calloc(m, sizeof(double)), *pn_fp = (double *)calloc(m, sizeof(double));
    double g[MAXR], glo[MAXR], ghi[MAXR];
    uint8_t in_new[MAXR];
    int64_t rounds = 0, base = 0;
    int64_t wend = 0; /* superset: pending sums cover [base, wend) */
    memset(stats, 0, sizeof(int64_t) * 8);
    memset(seg_len_hist, 0, sizeof(int32_t) * 32);

#define ADD_PENDING(pos, sgnmul)                                                      \
    for (int c_ = 0; c_ < ntr[pos]; ++c_) {                                           \
        const change_t *ch = trace + (pos) * 2 * k + c_;                              \
        if (ch->dtp > 0) pp_tp[ch->label] += (sgnmul) * ch->dtp;                      \
        else pn_tp[ch->label] += (sgnmul) * -ch->dtp;                                 \
        if (ch->dfp > 0) pp_fp[ch->label] += (sgnmul) * ch->dfp;                      \
        else pn_fp[ch->label] += (sgnmul) * -ch->dfp;                                 \
    }
#define COMMIT(pos)                                                                   \
    do {                                                                              \
        const int64_t i_ = order[pos];                                                \
        const int s_ = indptr[i_], r_ = indptr[i_ + 1] - s_;                          \
        for (int c_ = 0; c_ < ntr[pos]; ++c_) {                                       \
            const change_t *ch = trace + (pos) * 2 * k + c_;                          \
            tp[ch->label] += ch->dtp;                                                 \
            fp[ch->label] += ch->dfp;                                                 \
            for (int p_ = 0; p_ < r_; ++p_)                                           \
                if (indices[s_ + p_] == ch->label) sel[s_ + p_] = ch->dtp > 0;        \
        }                                                                             \
    } while (0)

    while (base < n_order) {
        /* a round: base is exact by definition */
        ++rounds;
        if (mode == 1 && superset) {
            const int64_t want = base + window < n_order ? base + window : n_order;
            for (; wend < want; ++wend) ADD_PENDING(wend, 1.0);
        }
        int64_t p = base;
        int why = 2;
        /* the pending contributions of rows [base, p) in prefix mode are accumulated as p advances */
        for (; p < n_order && p - base < window; ++p) {
            const int64_t i = order[p];
            const int s0 = indptr[i], r = indptr[i + 1] - s0;
            int uncertain = 0, boxes = 0;
            if (p > base) {
                for (int q = 0; q < r && !uncertain; ++q) {
                    const int j = indices[s0 + q];
                    if (mode == 1 && boxed[j]) continue;
                    if (lastw[j] >= base) uncertain = 1;
                }
                if (uncertain) why = 0;
                if (!uncertain && mode == 1) {
                    /* speculative decision on the committed records; boxes for the boxed candidates */
                    for (int q = 0; q < r; ++q) {
                        const int j = indices[s0 + q];
                        const float e = data[s0 + q];
                        const double ed = (double)e, omd = (double)(1.0f - e);
                        double tpc = tp[j], fpc = fp[j];
                        if (sel[s0 + q]) {
                            tpc -= ed;
                            fpc -= omd;
                        }
                        g[q] = gain(mt, tpc, fpc, colsum[j] - ed, ed, omd);
                        glo[q] = ghi[q] = g[q];
                        if (boxed[j] && (pp_tp[j] > 1e-9 || pn_tp[j] > 1e-9 || pp_fp[j] > 1e-9 || pn_fp[j] > 1e-9)) {
                            /* own tentative delta is part of a superset box: it is in the sums already */
                            gain_box(mt, tpc - pn_tp[j], tpc + pp_tp[j], fpc - pn_fp[j], fpc + pp_fp[j], colsum[j] - ed,
                                     ed, omd, &glo[q], &ghi[q]);
                            ++boxes;
                        }
                    }
                    if (boxes) {
                        topk_select(g, r, k, in_new);
                        double tin = INFINITY, tout = -INFINITY;
                        for (int q = 0; q < r; ++q) {
                            if (in_new[q]) tin = glo[q] < tin ? glo[q] : tin;
                            else tout = ghi[q] > tout ? ghi[q] : tout;
                        }
                        if (!(tin > tout)) {
                            uncertain = 1;
                            why = 1;
                        } else
                            ++stats[3];
                    }
                }
            }
            if (uncertain) break;
            /* certain: it will commit in this round; later rows of the round see it as an uncommitted writer */
            for (int c = 0; c < ntr[p]; ++c) lastw[trace[p * 2 * k + c].label] = p;
            if (mode == 1 && !superset) ADD_PENDING(p, 1.0);
        }
        if (p < n_order && p - base >= window) why = 2;
        if (p < n_order) ++stats[why];
        /* commit [base, p) */
        int64_t len = p - base;
        int b = 0;
        while ((1ll << (b + 1)) <= len && b < 31) ++b;
        ++seg_len_hist[b];
        for (int64_t q = base; q < p; ++q) {
            if (mode == 1) ADD_PENDING(q, -1.0);
            COMMIT(q);
        }
        if (mode == 1 && !superset) {
            /* floating residue of add/remove: clear exactly */
            for (int64_t q = base; q < p; ++q)
                for (int c = 0; c < ntr[q]; ++c) {
                    const int j = trace[q * 2 * k + c].label;
                    pp_tp[j] = pn_tp[j] = pp_fp[j] = pn_fp[j] = 0.0;
                }
        }
        base = p;
    }
    free(lastw);
    free(pp_tp);
    free(pn_tp);
    free(pp_fp);
    free(pn_fp);
    return rounds;
}

/* Depth of the TRUE dependency graph of a sweep (the schedule an oracle that knew every decision could run):
 * row p must score after every earlier row that changed one of its candidates has committed (level + 1) and a
 * row may not commit a label before every earlier row that reads it has scored (same level or later).
 * `ignore` (optional, per label): labels left out (handled by other means).  Returns the number of levels;
 * width_hist[log2(rows in the level)] is filled. */
int64_t sim_dag_depth(int64_t n_order, const int64_t *order, const int32_t *indptr, const int32_t *indices, int k,
                      const int32_t *ntr, const change_t *trace, int64_t m, const uint8_t *ignore, int32_t *level_out) {
    int32_t *lw = (int32_t *)malloc(sizeof(int32_t) * m), *lr = (int32_t *)malloc(sizeof(int32_t) * m);
    for (int64_t j = 0; j < m; ++j) lw[j] = lr[j] = -1;
    int32_t depth = 0;
    for (int64_t p = 0; p < n_order; ++p) {
        const int64_t i = order[p];
        const int s0 = indptr[i], r = indptr[i + 1] - s0;
        int32_t lv = 0;
        for (int q = 0; q < r; ++q) {
            const int j = indices[s0 + q];
            if (ignore && ignore[j]) continue;
            if (lw[j] + 1 > lv) lv = lw[j] + 1;
        }
        for (int c = 0; c < ntr[p]; ++c) {
            const int j = trace[p * 2 * k + c].label;
            if (ignore && ignore[j]) continue;
            if (lr[j] > lv) lv = lr[j];
        }
        for (int q = 0; q < r; ++q) {
            const int j = indices[s0 + q];
            if (lr[j] < lv) lr[j] = lv;
        }
        for (int c = 0; c < ntr[p]; ++c) {
            const int j = trace[p * 2 * k + c].label;
            if (lw[j] < lv) lw[j] = lv;
        }
        if (level_out) level_out[p] = lv;
        if (lv + 1 > depth) depth = lv + 1;
    }
    free(lw);
    free(lr);
    return depth;
}

/* Jacobi fix-point per window: every row of the window decides on (committed records + the changes the EARLIER rows
 * of the window made in the previous iteration); iterate until no decision moves -- the fixed point is the sequential
 * result (induction over the positions).  tp/fp/sel: state at the start of the sweep, advanced here.  Returns the
 * total number of iterations (sum over windows; an iteration = one parallel pass over the window's rows);
 * out[0] = windows, out[1] = max iterations of a window, out[2] = row evaluations whose input had changed,
 * out[3] = max writers of one label in a window, out[4] = labels with more than `slots` writers (summed over iterations). */
int64_t sim_fixpoint(int64_t n_order, const int64_t *order, const int32_t *indptr, const int32_t *indices,
                     const float *data, uint8_t *sel, int k, double *tp, double *fp, const double *colsum,
                     const sim_metric *mt, int64_t m, int64_t window, int slots, int64_t *out) {
    int32_t *head = (int32_t *)malloc(sizeof(int32_t) * m);
    int32_t *cnt = (int32_t *)calloc(m, sizeof(int32_t));
    for (int64_t j = 0; j < m; ++j) head[j] = -1;
    const int64_t cap = window * 2 * k;
    change_t *cur = (change_t *)malloc(sizeof(change_t) * cap), *nxt = (change_t *)malloc(sizeof(change_t) * cap);
    int32_t *ncur = (int32_t *)calloc(window, sizeof(int32_t)), *nnxt = (int32_t *)calloc(window, sizeof(int32_t));
    int32_t *link = (int32_t *)malloc(sizeof(int32_t) * cap); /* next entry of the same label */
    int32_t *epos = (int32_t *)malloc(sizeof(int32_t) * cap); /* window slot of the entry's row */
    double g[MAXR];
    uint8_t in_new[MAXR];
    int64_t total_iters = 0;
    memset(out, 0, sizeof(int64_t) * 8);
    for (int64_t base = 0; base < n_order; base += window) {
        const int64_t B = base + window <= n_order ? window : n_order - base;
        for (int64_t s = 0; s < B; ++s) ncur[s] = 0;
        int iters = 0;
        for (;;) {
            ++iters;
            /* per-label lists of the current decisions' changes (entries in slot order: ascending position) */
            int64_t ne = 0;
            for (int64_t s = B - 1; s >= 0; --s) /* reverse: the list heads end up in ascending order */
                for (int c = ncur[s] - 1; c >= 0; --c) {
                    const change_t *ch = cur + s * 2 * k + c;
                    const int64_t e = s * 2 * k + c;
                    link[e] = head[ch->label];
                    head[ch->label] = (int32_t)e;
                    epos[e] = (int32_t)s;
                    if (++cnt[ch->label] > out[3]) out[3] = cnt[ch->label];
                    if (cnt[ch->label] == slots + 1) ++out[4];
                    ++ne;
                }
            int moved = 0;
            for (int64_t s = 0; s < B; ++s) {
                const int64_t i = order[base + s];
                const int s0 = indptr[i], r = indptr[i + 1] - s0;
                int touched = 0;
                for (int q = 0; q < r; ++q) {
                    const int j = indices[s0 + q];
                    const float e = data[s0 + q];
                    const double ed = (double)e, omd = (double)(1.0f - e);
                    double tpc = tp[j], fpc = fp[j];
                    for (int32_t en = head[j]; en >= 0 && epos[en] < s; en = link[en]) {
                        tpc += cur[en].dtp;
                        fpc += cur[en].dfp;
                        touched = 1;
                    }
                    if (sel[s0 + q]) {
                        tpc -= ed;
                        fpc -= omd;
                    }
                    g[q] = gain(mt, tpc, fpc, colsum[j] - ed, ed, omd);
                }
                out[2] += touched;
                topk_select(g, r, k, in_new);
                int nc = 0;
                for (int q = 0; q < r; ++q)
                    if (in_new[q] != sel[s0 + q]) {
                        change_t *c = nxt + s * 2 * k + nc++;
                        const float e = data[s0 + q];
                        const double sgn = in_new[q] ? 1.0 : -1.0;
                        c->label = indices[s0 + q];
                        c->dtp = sgn * (double)e;
                        c->dfp = sgn * (double)(1.0f - e);
                    }
                nnxt[s] = nc;
                if (nc != ncur[s]) moved = 1;
                else
                    for (int c = 0; c < nc; ++c)
                        if (nxt[s * 2 * k + c].label != cur[s * 2 * k + c].label || nxt[s * 2 * k + c].dtp != cur[s * 2 * k + c].dtp)
                            moved = 1;
            }
            /* clear the lists */
            for (int64_t s = 0; s < B; ++s)
                for (int c = 0; c < ncur[s]; ++c) {
                    head[cur[s * 2 * k + c].label] = -1;
                    cnt[cur[s * 2 * k + c].label] = 0;
                }
            change_t *t = cur; cur = nxt; nxt = t;
            int32_t *tn = ncur; ncur = nnxt; nnxt = tn;
            if (!moved) break;
        }
        total_iters += iters;
        ++out[0];
        if (iters > out[1]) out[1] = iters;
        /* commit the window */
        for (int64_t s = 0; s < B; ++s) {
            const int64_t i = order[base + s];
            const int s0 = indptr[i], r = indptr[i + 1] - s0;
            for (int c = 0; c < ncur[s]; ++c) {
                const change_t *ch = cur + s * 2 * k + c;
                tp[ch->label] += ch->dtp;
                fp[ch->label] += ch->dfp;
                for (int q = 0; q < r; ++q)
                    if (indices[s0 + q] == ch->label) sel[s0 + q] = ch->dtp > 0;
            }
        }
    }
    free(head); free(cnt); free(cur); free(nxt); free(ncur); free(nnxt); free(link); free(epos);
    return total_iters;
}

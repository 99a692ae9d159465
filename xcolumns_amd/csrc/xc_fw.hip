// xc_fw.hip -- the O(m) side of the Frank-Wolfe search (xcolumns/frank_wolfe.py:407-690):
// the gradient of the utility in the 4 x m confusion entries (:368-376, :585-596) and
// the utility along the segment between two confusion matrices (:379-404).
//
// The reference differentiates its numpy metric formulas with the `autograd` package;
// here the same formulas (metrics.py:497-944, written once as fw_metric<N>) are
// instantiated on forward-mode dual numbers carrying the four partial derivatives,
// so value and gradient come from one evaluation per label.
//
// Both kernels are float64 VALU work on vectors that fit in L2 / MALL (32 B per label):
// the gradient is one pass; the step-size search evaluates the metric m x n_alpha times
// (n_alpha = 10^4 for the default uniform search), tiled so that a workgroup stages 256
// labels in LDS and every thread owns one alpha.
#include "xc_common.h"
#include "xc_host.h"

#define XC_FW_TILE 256
#define XC_FW_MAX_CHUNKS 128 /* label chunks of the step scan (grid y) */
#define XC_FW_EXACT_POINTS 16 /* up to this many alphas: the reference's exact arithmetic */

namespace xc {

struct Dual {
    double v, d0, d1, d2, d3;
    __device__ Dual() {}
    __device__ Dual(double x) : v(x), d0(0.0), d1(0.0), d2(0.0), d3(0.0) {}
    __device__ Dual(double x, double a, double b, double c, double d) : v(x), d0(a), d1(b), d2(c), d3(d) {}
};

__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d0 + b.d0, a.d1 + b.d1, a.d2 + b.d2, a.d3 + b.d3); }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d0 - b.d0, a.d1 - b.d1, a.d2 - b.d2, a.d3 - b.d3); }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) {
    return Dual(a.v * b.v, a.d0 * b.v + a.v * b.d0, a.d1 * b.v + a.v * b.d1, a.d2 * b.v + a.v * b.d2,
                a.d3 * b.v + a.v * b.d3);
}
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
    const double q = a.v / b.v;
    return Dual(q, (a.d0 - q * b.d0) / b.v, (a.d1 - q * b.d1) / b.v, (a.d2 - q * b.d2) / b.v, (a.d3 - q * b.d3) / b.v);
}
__device__ __forceinline__ Dual nsqrt(Dual a) {
    const double s = sqrt(a.v);
    const double h = 0.5 / s;
    return Dual(s, h * a.d0, h * a.d1, h * a.d2, h * a.d3);
}
__device__ __forceinline__ double nsqrt(double a) { return sqrt(a); }

// float64 with the ~1-ulp division of xc_common.h (v_rcp_f64 + two Newton steps): used by the step-size
// scan, whose label sums already differ from numpy's pairwise ones in the last bits
struct FastD {
    double v;
    __device__ FastD() {}
    __device__ FastD(double x) : v(x) {}
};
__device__ __forceinline__ FastD operator+(FastD a, FastD b) { return FastD(a.v + b.v); }
__device__ __forceinline__ FastD operator-(FastD a, FastD b) { return FastD(a.v - b.v); }
__device__ __forceinline__ FastD operator*(FastD a, FastD b) { return FastD(a.v * b.v); }
__device__ __forceinline__ FastD operator/(FastD a, FastD b) { return FastD(fdiv<false>(a.v, b.v)); }
__device__ __forceinline__ FastD nsqrt(FastD a) { return FastD(sqrt(a.v)); }

// metrics.py formulas in the reference's operation order (the file is compiled with
// -ffp-contract=off), for N = double or Dual
// BASE >= 0 fixes the formula at compile time (the step scan's inner loop), -1 reads mt.base
template <typename N, int BASE = -1>
__device__ __forceinline__ N fw_base(const xc_metric &mt, N tp, N fp, N fn, N tn) {
    const double eps = mt.epsilon;
    switch (BASE >= 0 ? BASE : mt.base) {
    case XC_M_PRECISION_AT_K: return tp / N(mt.kf);                       // :513
    case XC_M_PRECISION: return tp / (tp + fp + N(eps));                   // :605
    case XC_M_RECALL: return tp / (tp + fn + N(eps));                      // :652
    case XC_M_FBETA: {                                                     // :703
        const double b2 = mt.beta * mt.beta;
        return (N(1.0 + b2) * tp) / ((N(b2) * (tp + fp)) + tp + fn + N(eps));
    }
    case XC_M_JACCARD: return tp / (tp + fp + fn + N(eps));                // :797
    case XC_M_BALANCED_ACC: {                                              // :843-845
        const N tpr = tp / (tp + fn + N(eps));
        const N tnr = tn / (tn + fp + N(eps));
        return (tpr + tnr) / N(2.0);
    }
    case XC_M_GMEAN: {                                                     // :892-894
        const N tpr = tp / (tp + fn + N(eps));
        const N tnr = tn / (tn + fp + N(eps));
        return nsqrt(tpr * tnr);
    }
    case XC_M_HMEAN: {                                                     // :942-944
        const N tpr = tp / (tp + fn + N(eps));
        const N tnr = tn / (tn + fp + N(eps));
        return (N(2.0) * tpr * tnr) / (tpr + tnr);
    }
    case XC_M_ACCURACY: return (tp + tn) / (tp + fp + fn + tn);            // :416-419
    case XC_M_RECALL_PRECISION_MIX:                                        // frank_wolfe.py:925-929
        return N(1.0 - mt.alpha) * (tp / (tp + fn + N(eps))) + N(mt.alpha) * (tp / (tp + fp + N(eps)));
    default: return N(__builtin_nan(""));
    }
}

template <typename N, int BASE = -1>
__device__ __forceinline__ N fw_metric(const xc_metric &mt, N tp, N fp, N fn, N tn) {
    N v = fw_base<N, BASE>(mt, tp, fp, fn, tn);
    if (mt.mixed) // frank_wolfe.py:832-838
        v = N(1.0 - mt.alpha) * (tp / N(mt.kf)) + (N(mt.alpha) * v) / N(mt.mf);
    return v;
}

// a_j = G_tp - G_fp - G_fn + G_tn, b_j = G_fp - G_tn (frank_wolfe.py:592-593); G = dpsi_j / d(.) / div
__global__ __launch_bounds__(XC_BLOCK) void fw_gradient_kernel(int64_t m, const double *stats, xc_metric metric,
                                                               double div, int negate, double *a, double *b) {
    const int64_t j = (int64_t)blockIdx.x * XC_BLOCK + threadIdx.x;
    if (j >= m) return;
    const Dual tp(stats[j], 1.0, 0.0, 0.0, 0.0), fp(stats[m + j], 0.0, 1.0, 0.0, 0.0);
    const Dual fn(stats[2 * m + j], 0.0, 0.0, 1.0, 0.0), tn(stats[3 * m + j], 0.0, 0.0, 0.0, 1.0);
    const Dual u = fw_metric<Dual>(metric, tp, fp, fn, tn);
    const double gtp = u.d0 / div, gfp = u.d1 / div, gfn = u.d2 / div, gtn = u.d3 / div;
    double aj = gtp - gfp - gfn + gtn;
    double bj = gfp - gtn;
    if (negate) { // :594-596
        aj = -aj;
        bj = -bj;
    }
    a[j] = aj;
    b[j] = bj;
}

// partials[chunk][t] = sum over the chunk's labels of psi((1 - alpha_t) * cur_j + alpha_t * nxt_j)
// EXACT: the reference's expression and IEEE division (two-point ternary steps).  Otherwise (the
// 10^4-point uniform scan, VALU-bound): the mix as one fma per entry, cur + alpha * (nxt - cur), and
// the ~1-ulp division -- differences of the size of the summation-order ones.
template <bool EXACT, int BASE>
__global__ __launch_bounds__(XC_FW_TILE) void fw_alpha_curve_kernel(int64_t m, const double *cur, const double *nxt,
                                                                   xc_metric metric, int n_alpha,
                                                                   const double *alphas, int64_t per_chunk,
                                                                   double *partials) {
    __shared__ double s_cur[XC_FW_TILE][4];
    __shared__ double s_nxt[XC_FW_TILE][4]; // !EXACT: nxt - cur
    const int t = blockIdx.x * XC_FW_TILE + threadIdx.x;
    const bool live = t < n_alpha;
    const double alpha = live ? alphas[t] : 0.0;
    const double keep = 1.0 - alpha;
    const int64_t j0 = (int64_t)blockIdx.y * per_chunk;
    const int64_t j1 = (j0 + per_chunk < m) ? j0 + per_chunk : m;
    double sum = 0.0;
    for (int64_t base = j0; base < j1; base += XC_FW_TILE) {
        const int64_t j = base + threadIdx.x;
        const int cnt = (int)((j1 - base < XC_FW_TILE) ? j1 - base : XC_FW_TILE);
        __syncthreads();
        if (j < j1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double c = cur[q * m + j], x = nxt[q * m + j];
                s_cur[threadIdx.x][q] = c;
                s_nxt[threadIdx.x][q] = EXACT ? x : x - c;
            }
        }
        __syncthreads();
        if (live) {
#pragma unroll 4
            for (int i = 0; i < cnt; ++i) {
                if (EXACT) { // frank_wolfe.py:392-397, the same expression per entry
                    const double tp = keep * s_cur[i][0] + alpha * s_nxt[i][0];
                    const double fp = keep * s_cur[i][1] + alpha * s_nxt[i][1];
                    const double fn = keep * s_cur[i][2] + alpha * s_nxt[i][2];
                    const double tn = keep * s_cur[i][3] + alpha * s_nxt[i][3];
                    sum += fw_metric<double, BASE>(metric, tp, fp, fn, tn);
                } else {
                    const FastD tp(__builtin_fma(alpha, s_nxt[i][0], s_cur[i][0]));
                    const FastD fp(__builtin_fma(alpha, s_nxt[i][1], s_cur[i][1]));
                    const FastD fn(__builtin_fma(alpha, s_nxt[i][2], s_cur[i][2]));
                    const FastD tn(__builtin_fma(alpha, s_nxt[i][3], s_cur[i][3]));
                    sum += fw_metric<FastD, BASE>(metric, tp, fp, fn, tn).v;
                }
            }
        }
    }
    if (live) partials[(int64_t)blockIdx.y * n_alpha + t] = sum;
}

// The long scan for the linear-fractional metrics (precision, recall, F-beta, Jaccard, plain or mixed with
// precision@k): along the segment both numerator and denominator are linear in alpha,
//   psi_j(alpha) = (N0 + alpha dN) / (D0 + alpha dD)  [+ L0 + alpha dL for the mixed utilities],
// so the six per-label constants are formed once while the tile is staged and an evaluation is two (three)
// fmas and one ~1-ulp division -- about half the float64 instructions of the general fast path, again within
// the rounding differences the scan already has against numpy's pairwise sums.
// Every thread carries A step sizes, so a label's constants are read from LDS once per A evaluations, and
// the evaluation itself is cut to its reciprocal.  While a tile is staged each label is put in one of three
// classes:
//   linear  dD == 0 (the step does not move the denominator; most labels of a large label space):
//           psi = N0/D0 + alpha dN/D0 -- two per-label numbers that are summed over the labels ONCE and
//           applied to every step size at the end; nothing per evaluation;
//   smooth  psi = c1 + c2 / d(alpha) with c1 = dN/dD, c2 = N0 - c1 D0 (n - c1 d does not depend on alpha):
//           c1 joins the per-label sums, an evaluation is d = fma(alpha, dD, D0), its reciprocal (v_rcp_f64 +
//           one cubic step) and one fma -- 6 float64 instructions instead of 9.  Taken when |c1| <= 64 and
//           the two end-point denominators are within 1024x of each other, which bounds the cancellation in
//           c1 + c2 r to ~1e-11 absolute per label (the label sums are O(m));
//   rough   everything else (denominators that start or end near epsilon, 0/0): numerator * reciprocal.
// The linear part of the mixed utilities, (1 - a)/k (tp + alpha dtp), is a per-label sum as well.  Smooth
// labels are staged from the front of the tile, rough ones from the back: two branch-free loops.
#define XC_FW_SMOOTH_C1 64.0
#define XC_FW_SMOOTH_RATIO 1024.0
// v_rcp_f64 is good to 2^-24.4 on gfx950 (tools/rcp_probe.hip); one cubic step r (1 + e + e^2), e = 1 - d r,
// takes that to 2^-53 in three fmas (two Newton steps need four)
__device__ __forceinline__ double fw_rcp(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    const double e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
}

template <int BASE, int A, bool MIXED>
__global__ __launch_bounds__(XC_FW_TILE) void fw_alpha_curve_linfrac_kernel(int64_t m, const double *cur,
                                                                           const double *nxt, xc_metric metric,
                                                                           int n_alpha, const double *alphas,
                                                                           int64_t per_chunk, double *partials) {
    constexpr int NW = XC_FW_TILE / XC_WAVE;
    __shared__ double s_c[XC_FW_TILE][4]; // smooth: c2, -, D0, dD   rough: N0, dN, D0, dD
    __shared__ double s_lin[NW][2];
    __shared__ int s_smooth[NW], s_rough[NW];
    const int lane = threadIdx.x & (XC_WAVE - 1), wave = threadIdx.x / XC_WAVE;
    int t[A];
    double alpha[A], sum[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
        t[a] = (blockIdx.x * A + a) * XC_FW_TILE + threadIdx.x;
        alpha[a] = t[a] < n_alpha ? alphas[t[a]] : 0.0;
        sum[a] = 0.0;
    }
    const int64_t j0 = (int64_t)blockIdx.y * per_chunk;
    const int64_t j1 = (j0 + per_chunk < m) ? j0 + per_chunk : m;
    const double eps = metric.epsilon, b2 = metric.beta * metric.beta;
    const double scale = MIXED ? metric.alpha / metric.mf : 1.0;      // frank_wolfe.py:832-838
    const double lin = MIXED ? (1.0 - metric.alpha) / metric.kf : 0.0;
    double lin0 = 0.0, lin1 = 0.0; // this thread's labels: sum of the constant / of the alpha coefficient
    for (int64_t base = j0; base < j1; base += XC_FW_TILE) {
        const int64_t j = base + threadIdx.x;
        double N0 = 0.0, dN = 0.0, D0 = 1.0, dD = 0.0;
        bool smooth = false, rough = false;
        if (j < j1) {
            const double tp = cur[j], fp = cur[m + j], fn = cur[2 * m + j];
            const double dtp = nxt[j] - tp, dfp = nxt[m + j] - fp, dfn = nxt[2 * m + j] - fn;
            if (BASE == XC_M_PRECISION) {
                N0 = tp, dN = dtp, D0 = tp + fp + eps, dD = dtp + dfp;
            } else if (BASE == XC_M_RECALL) {
                N0 = tp, dN = dtp, D0 = tp + fn + eps, dD = dtp + dfn;
            } else if (BASE == XC_M_FBETA) {
                N0 = (1.0 + b2) * tp, dN = (1.0 + b2) * dtp;
                D0 = (b2 * (tp + fp)) + tp + fn + eps, dD = (b2 * (dtp + dfp)) + dtp + dfn;
            } else { // XC_M_JACCARD
                N0 = tp, dN = dtp, D0 = tp + fp + fn + eps, dD = dtp + dfp + dfn;
            }
            N0 *= scale, dN *= scale;
            if (MIXED) {
                lin0 += lin * tp;
                lin1 += lin * dtp;
            }
            const double D1 = D0 + dD;
            const double dlow = fmin(D0, D1), dhigh = fmax(D0, D1);
            if (dD == 0.0 && D0 > 0.0) {
                lin0 += N0 / D0;
                lin1 += dN / D0;
            } else if (dlow > 0.0 && dhigh <= XC_FW_SMOOTH_RATIO * dlow && fabs(dN) <= XC_FW_SMOOTH_C1 * fabs(dD)) {
                const double c1 = dN / dD;
                lin0 += c1;
                N0 = __builtin_fma(-c1, D0, N0); // c2
                smooth = true;
            } else {
                rough = true;
            }
        }
        const unsigned long long b_smooth = __ballot(smooth), b_rough = __ballot(rough);
        __syncthreads(); // the previous tile has been read by every thread
        if (lane == 0) {
            s_smooth[wave] = __popcll(b_smooth);
            s_rough[wave] = __popcll(b_rough);
        }
        __syncthreads();
        int n_smooth = 0, n_rough = 0, smooth_before = 0, rough_before = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            if (w < wave) {
                smooth_before += s_smooth[w];
                rough_before += s_rough[w];
            }
            n_smooth += s_smooth[w];
            n_rough += s_rough[w];
        }
        if (smooth || rough) {
            const unsigned long long below = (1ull << lane) - 1ull;
            const int pos = smooth ? smooth_before + __popcll(b_smooth & below)
                                   : XC_FW_TILE - 1 - (rough_before + __popcll(b_rough & below));
            s_c[pos][0] = N0;
            s_c[pos][1] = dN;
            s_c[pos][2] = D0;
            s_c[pos][3] = dD;
        }
        __syncthreads();
#pragma unroll 2
        for (int i = 0; i < n_smooth; ++i) {
            const double c2 = s_c[i][0], cD0 = s_c[i][2], cdD = s_c[i][3];
#pragma unroll
            for (int a = 0; a < A; ++a)
                sum[a] = __builtin_fma(c2, fw_rcp(__builtin_fma(alpha[a], cdD, cD0)), sum[a]);
        }
#pragma unroll 2
        for (int i = XC_FW_TILE - n_rough; i < XC_FW_TILE; ++i) {
            const double cN0 = s_c[i][0], cdN = s_c[i][1], cD0 = s_c[i][2], cdD = s_c[i][3];
#pragma unroll
            for (int a = 0; a < A; ++a)
                sum[a] += __builtin_fma(alpha[a], cdN, cN0) * fw_rcp(__builtin_fma(alpha[a], cdD, cD0));
        }
    }
    // the per-label sums of the whole chunk, in a fixed order: lanes, then wavefronts
#pragma unroll
    for (int off = XC_WAVE / 2; off > 0; off >>= 1) {
        lin0 += __shfl_xor(lin0, off);
        lin1 += __shfl_xor(lin1, off);
    }
    __syncthreads();
    if (lane == 0) {
        s_lin[wave][0] = lin0;
        s_lin[wave][1] = lin1;
    }
    __syncthreads();
    lin0 = s_lin[0][0], lin1 = s_lin[0][1];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        lin0 += s_lin[w][0];
        lin1 += s_lin[w][1];
    }
#pragma unroll
    for (int a = 0; a < A; ++a)
        if (t[a] < n_alpha) partials[(int64_t)blockIdx.y * n_alpha + t[a]] = sum[a] + __builtin_fma(alpha[a], lin1, lin0);
}

} // namespace xc

extern "C" {

int xc_fw_gradient(int64_t m, const double *stats, const xc_metric *metric_host, double div, int negate,
                   double *a, double *b, void *stream) {
    if (m < 1 || !stats || !metric_host || !a || !b) return xc::fail_arg(XC_ERR_BAD_ARG, "xc_fw_gradient: bad argument");
    if (metric_host->base < 0 || metric_host->base >= XC_M_COUNT)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_fw_gradient: unknown metric %d", metric_host->base);
    const int blocks = (int)((m + XC_BLOCK - 1) / XC_BLOCK);
    hipLaunchKernelGGL(xc::fw_gradient_kernel, dim3(blocks), dim3(XC_BLOCK), 0, xc::as_stream(stream), m, stats,
                       *metric_host, div, negate, a, b);
    XC_CHECK_LAUNCH("fw_gradient_kernel");
    return XC_OK;
}

int xc_fw_alpha_chunks(int64_t m) {
    const int64_t tiles = (m + XC_FW_TILE - 1) / XC_FW_TILE;
    return (int)(tiles < 1 ? 1 : (tiles > XC_FW_MAX_CHUNKS ? XC_FW_MAX_CHUNKS : tiles));
}

int xc_fw_alpha_curve(int64_t m, const double *cur, const double *nxt, const xc_metric *metric_host, int n_alpha,
                      const double *alphas, double *partials, void *stream) {
    if (m < 1 || n_alpha < 1 || !cur || !nxt || !metric_host || !alphas || !partials)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_fw_alpha_curve: bad argument");
    if (metric_host->base < 0 || metric_host->base >= XC_M_COUNT)
        return xc::fail_arg(XC_ERR_BAD_ARG, "xc_fw_alpha_curve: unknown metric %d", metric_host->base);
    const int chunks = xc_fw_alpha_chunks(m);
    // whole tiles per chunk, so a chunk's labels are summed in ascending order tile by tile
    const int64_t tiles = (m + XC_FW_TILE - 1) / XC_FW_TILE;
    const int64_t per_chunk = ((tiles + chunks - 1) / chunks) * XC_FW_TILE;
    const int gx = (n_alpha + XC_FW_TILE - 1) / XC_FW_TILE;
    hipStream_t st = xc::as_stream(stream);
    if (n_alpha <= XC_FW_EXACT_POINTS) {
        hipLaunchKernelGGL((xc::fw_alpha_curve_kernel<true, -1>), dim3(gx, chunks), dim3(XC_FW_TILE), 0, st, m, cur, nxt,
                           *metric_host, n_alpha, alphas, per_chunk, partials);
    } else {
        // the long scan is float64-VALU-bound: the formula is fixed at compile time so the inner loop
        // carries no switch
#define XC_FW_SCAN(B)                                                                                              \
    case B:                                                                                                        \
        hipLaunchKernelGGL((xc::fw_alpha_curve_kernel<false, B>), dim3(gx, chunks), dim3(XC_FW_TILE), 0, st, m, cur, \
                           nxt, *metric_host, n_alpha, alphas, per_chunk, partials);                               \
        break
#define XC_FW_SCAN_LINFRAC_AM(B, A, M)                                                                              \
    hipLaunchKernelGGL((xc::fw_alpha_curve_linfrac_kernel<B, A, M>), dim3((gx + A - 1) / A, chunks),                 \
                       dim3(XC_FW_TILE), 0, st, m, cur, nxt, *metric_host, n_alpha, alphas, per_chunk, partials)
#define XC_FW_SCAN_LINFRAC(B)                                                                                      \
    case B:                                                                                                        \
        if (n_alpha >= 4 * XC_FW_TILE) {                                                                           \
            if (metric_host->mixed) XC_FW_SCAN_LINFRAC_AM(B, 4, true);                                             \
            else XC_FW_SCAN_LINFRAC_AM(B, 4, false);                                                               \
        } else {                                                                                                   \
            if (metric_host->mixed) XC_FW_SCAN_LINFRAC_AM(B, 1, true);                                             \
            else XC_FW_SCAN_LINFRAC_AM(B, 1, false);                                                               \
        }                                                                                                          \
        break
        switch (metric_host->base) {
            XC_FW_SCAN(XC_M_PRECISION_AT_K);
            XC_FW_SCAN_LINFRAC(XC_M_PRECISION);
            XC_FW_SCAN_LINFRAC(XC_M_RECALL);
            XC_FW_SCAN_LINFRAC(XC_M_FBETA);
            XC_FW_SCAN_LINFRAC(XC_M_JACCARD);
            XC_FW_SCAN(XC_M_BALANCED_ACC);
            XC_FW_SCAN(XC_M_GMEAN);
            XC_FW_SCAN(XC_M_HMEAN);
            XC_FW_SCAN(XC_M_ACCURACY);
            XC_FW_SCAN(XC_M_RECALL_PRECISION_MIX);
        }
#undef XC_FW_SCAN_LINFRAC
#undef XC_FW_SCAN_LINFRAC_AM
#undef XC_FW_SCAN
    }
    XC_CHECK_LAUNCH("fw_alpha_curve_kernel");
    return XC_OK;
}

} // extern "C"

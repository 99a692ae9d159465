// xc_common.h -- shared device helpers for libxcolumns_amd (gfx950 only).
//
// Wave = 64 lanes everywhere.  One wavefront owns one row of y_proba: lane l
// holds candidates l, l + 64, ... (CH of them, compile-time), which keeps the
// candidates of a sorted CSR row in ascending column order when read lane-major
// chunk by chunk.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xcolumns_amd.h"

#define XC_WAVE 64
#define XC_BLOCK 256 /* 4 waves: one per SIMD of a CU */

namespace xc {

__device__ __forceinline__ int lane_id() { return threadIdx.x & (XC_WAVE - 1); }

__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

// Statistics records are written by other waves' float64 atomics (performed at
// the memory side of the XCD's L2), so they are read with agent-scope relaxed
// atomic loads: `global_load_dwordx2 ... sc1`, which skips this CU's L1 (never
// refreshed by other CUs' writes -- MI355X_MICROARCH "inter-workgroup
// visibility").
__device__ __forceinline__ double load_coherent(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// No-return float64 add: one `global_atomic_add_f64`.
__device__ __forceinline__ void atomic_add_f64(double *p, double v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void atomic_add_f32(float *p, float v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Returning forms (`global_atomic_add_f32/f64 ... sc0`): the value the memory held before this add.
__device__ __forceinline__ float atomic_add_ret_f32(float *p, float v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double atomic_add_ret_f64(double *p, double v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- top-k selection key ---------------------------------------------------
// Order: larger gain first; equal gains -> lower candidate position (= lower
// column id in a sorted row); NaN gains last (numpy sorts NaN last in -gains).
template <typename G>
struct Best {
    G g;
    int p;
};

template <typename G>
__device__ __forceinline__ G nan_to_neg_inf(G g) {
    return (g != g) ? -INFINITY : g;
}

template <typename G>
__device__ __forceinline__ bool beats(G g, int p, G og, int op) {
    return (g > og) || (g == og && p < op);
}

template <typename G>
__device__ __forceinline__ Best<G> wave_argmax(Best<G> b) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        G og = __shfl_xor(b.g, off, XC_WAVE);
        int op = __shfl_xor(b.p, off, XC_WAVE);
        if (beats(og, op, b.g, b.p)) {
            b.g = og;
            b.p = op;
        }
    }
    return b; // identical in all lanes
}

// ---- wavefront reductions on DPP (gfx9 row_shr / row_bcast) -------------------
// v_max_u32_dpp etc.: one VALU instruction per step, no LDS crossbar.  After the
// four row_shr steps lane 15 of each 16-lane row holds the row's result;
// row_bcast:15 / row_bcast:31 carry it across rows so lane 63 holds the wave's.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_src(unsigned identity, unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xF, false);
}

__device__ __forceinline__ unsigned wave_umax32(unsigned v) {
    v = max(v, dpp_src<0x111, 0xF>(0u, v)); // row_shr:1
    v = max(v, dpp_src<0x112, 0xF>(0u, v)); // row_shr:2
    v = max(v, dpp_src<0x114, 0xF>(0u, v)); // row_shr:4
    v = max(v, dpp_src<0x118, 0xF>(0u, v)); // row_shr:8
    v = max(v, dpp_src<0x142, 0xA>(0u, v)); // row_bcast:15 into rows 1, 3
    v = max(v, dpp_src<0x143, 0xC>(0u, v)); // row_bcast:31 into rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ unsigned wave_umin32(unsigned v) {
    v = min(v, dpp_src<0x111, 0xF>(~0u, v));
    v = min(v, dpp_src<0x112, 0xF>(~0u, v));
    v = min(v, dpp_src<0x114, 0xF>(~0u, v));
    v = min(v, dpp_src<0x118, 0xF>(~0u, v));
    v = min(v, dpp_src<0x142, 0xA>(~0u, v));
    v = min(v, dpp_src<0x143, 0xC>(~0u, v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- all-reduce inside each 16-lane DPP row (quad swaps, then the two row mirrors): every
// lane ends with its row's result; four VALU instructions, no cross-row traffic ----
__device__ __forceinline__ unsigned row16_umax32(unsigned v) {
    v = max(v, dpp_src<0xB1, 0xF>(0u, v));  // quad_perm:[1,0,3,2]
    v = max(v, dpp_src<0x4E, 0xF>(0u, v));  // quad_perm:[2,3,0,1]
    v = max(v, dpp_src<0x141, 0xF>(0u, v)); // row_half_mirror
    v = max(v, dpp_src<0x140, 0xF>(0u, v)); // row_mirror
    return v;
}

__device__ __forceinline__ unsigned row16_umin32(unsigned v) {
    v = min(v, dpp_src<0xB1, 0xF>(~0u, v));
    v = min(v, dpp_src<0x4E, 0xF>(~0u, v));
    v = min(v, dpp_src<0x141, 0xF>(~0u, v));
    v = min(v, dpp_src<0x140, 0xF>(~0u, v));
    return v;
}

// 64-bit max / min as two 32-bit reductions: high words first, then the low words
// of the lanes that hold the winning high word
__device__ __forceinline__ unsigned long long wave_umax64(unsigned long long v) {
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    const unsigned H = wave_umax32(hi);
    const unsigned L = wave_umax32(hi == H ? lo : 0u);
    return ((unsigned long long)H << 32) | L;
}

__device__ __forceinline__ unsigned long long wave_umin64(unsigned long long v) {
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    const unsigned H = wave_umin32(hi);
    const unsigned L = wave_umin32(hi == H ? lo : ~0u);
    return ((unsigned long long)H << 32) | L;
}

// Order-preserving map from float64 to uint64 (larger gain -> larger key).  NaN
// was mapped to -inf before.  Every finite/infinite double maps to a key >= 1
// (-inf -> 0x000fffffffffffff), so key 0 marks "no candidate".
// -0.0 and +0.0 compare equal in the reference's numpy selection (a tie: the lower column
// wins), so both map to the key of +0.0 (x + 0.0 turns -0.0 into +0.0 and nothing else).
__device__ __forceinline__ unsigned long long sortable_key(double g) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(g + 0.0);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// float32 flavour: every non-NaN float maps to a key >= 1 (0 = no candidate)
__device__ __forceinline__ unsigned sortable_key32(float g) {
    const unsigned u = __float_as_uint(g + 0.0f);
    return (u >> 31) ? ~u : (u | 0x80000000u);
}

// ---- division ------------------------------------------------------------------
// EXACT: IEEE division, bit-identical to the reference's numpy `/`.
// Otherwise: v_rcp_f64 refined by two Newton steps, then one multiply -- about
// 1 ulp instead of 0.5 ulp, half the instructions of the IEEE sequence.  Used by
// the concurrent sweep, whose trajectory is not the reference's anyway.
template <bool EXACT>
__device__ __forceinline__ double fdiv(double a, double b) {
    if (EXACT) return a / b;
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    return a * r;
}

// ---- binary metrics (xcolumns/metrics.py) ------------------------------------
// Same operation order as the reference's numpy expressions; the translation
// unit is compiled with -ffp-contract=off so no multiply is fused into an add.
template <bool EXACT>
__device__ __forceinline__ double metric_base_t(const xc_metric &mt, double tp, double fp, double fn,
                                                double tn) {
    const double eps = mt.epsilon;
    switch (mt.base) {
    case XC_M_PRECISION_AT_K: // metrics.py:513
        return fdiv<EXACT>(tp, mt.kf);
    case XC_M_PRECISION: // :605
        return fdiv<EXACT>(tp, tp + fp + eps);
    case XC_M_RECALL: // :652
        return fdiv<EXACT>(tp, tp + fn + eps);
    case XC_M_FBETA: { // :703
        const double b2 = mt.beta * mt.beta;
        return fdiv<EXACT>((1.0 + b2) * tp, (b2 * (tp + fp)) + tp + fn + eps);
    }
    case XC_M_JACCARD: // :797
        return fdiv<EXACT>(tp, tp + fp + fn + eps);
    case XC_M_BALANCED_ACC: { // :843-845
        const double tpr = fdiv<EXACT>(tp, tp + fn + eps);
        const double tnr = fdiv<EXACT>(tn, tn + fp + eps);
        return (tpr + tnr) / 2.0;
    }
    case XC_M_GMEAN: { // :892-894
        const double tpr = fdiv<EXACT>(tp, tp + fn + eps);
        const double tnr = fdiv<EXACT>(tn, tn + fp + eps);
        return sqrt(tpr * tnr);
    }
    case XC_M_HMEAN: { // :942-944
        const double tpr = fdiv<EXACT>(tp, tp + fn + eps);
        const double tnr = fdiv<EXACT>(tn, tn + fp + eps);
        return fdiv<EXACT>(2.0 * tpr * tnr, tpr + tnr);
    }
    case XC_M_ACCURACY: // :416-419
        return fdiv<EXACT>(tp + tn, tp + fp + fn + tn);
    case XC_M_RECALL_PRECISION_MIX: // frank_wolfe.py:925-929
        return (1.0 - mt.alpha) * fdiv<EXACT>(tp, tp + fn + eps) + mt.alpha * fdiv<EXACT>(tp, tp + fp + eps);
    default:
        return __builtin_nan("");
    }
}

template <bool EXACT>
__device__ __forceinline__ double metric_eval_t(const xc_metric &mt, double tp, double fp, double fn,
                                                double tn) {
    double v = metric_base_t<EXACT>(mt, tp, fp, fn, tn);
    if (mt.mixed) // block_coordinate.py:862-865 and siblings
        v = (1.0 - mt.alpha) * fdiv<EXACT>(tp, mt.kf) + fdiv<EXACT>(mt.alpha * v, mt.mf);
    return v;
}

__device__ __forceinline__ double metric_eval(const xc_metric &mt, double tp, double fp, double fn,
                                              double tn) {
    return metric_eval_t<true>(mt, tp, fp, fn, tn);
}

} // namespace xc

"""CPU oracle for the Frank-Wolfe path -- TEST INFRASTRUCTURE.

A numpy restatement of ``/root/reference/xcolumns/frank_wolfe.py``:
``find_classifier_using_fw`` (:407-690), ``_find_best_alpha`` (:379-404 with
``utils.py:174-201``) and ``predict_using_randomized_weighted_classifier``
(:85-172).  The reference differentiates the metric with the ``autograd`` package
(:368-376), which this image does not have; here the gradients are the closed
forms of the same formulas (metrics.py:497-944), written out per metric.

Parity status: PINNED by ``tests/test_oracle_golden.py`` against fixtures made by
``tests/golden/make_golden.py`` from the reference's own ``frank_wolfe.py``, run
with a stand-in for ``autograd.grad`` that differentiates with ``torch.autograd``
(what the reference's own torch branch does, frank_wolfe.py:18-41).  The pin is
therefore one step weaker than the BCA one: the differentiation engine was
substituted, everything else executed is the reference's code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module.
"""
from __future__ import annotations

from dataclasses import dataclass
from time import time
from typing import Any, Dict, Optional, Tuple

import numpy as np
from scipy.sparse import csr_matrix

from . import ref

(PRECISION_AT_K, PRECISION, RECALL, FBETA, JACCARD, BALANCED_ACC, GMEAN, HMEAN, ACCURACY,
 RECALL_PRECISION_MIX) = range(10)


@dataclass
class FwMetric:
    """A scalar utility of the four confusion vectors.

    average: "macro" (mean over labels, metrics.py:51-58), "micro" (the binary formula
    on the label sums, :83-90) or "sum" (the mixed utilities, frank_wolfe.py:832-838).
    mixed: value_j = (1 - alpha) * tp_j / k + alpha * base_j / m   (frank_wolfe.py:832-838)
    base RECALL_PRECISION_MIX: (1 - alpha) * recall_j + alpha * precision_j (:925-929)
    """
    base: int
    average: str = "macro"
    epsilon: float = 1e-9
    beta: float = 1.0
    k: float = 1.0
    mixed: bool = False
    alpha: float = 1.0
    m: float = 1.0


def _base_value_and_grad(mt: FwMetric, tp, fp, fn, tn):
    """(psi, dpsi/dtp, dpsi/dfp, dpsi/dfn, dpsi/dtn), elementwise."""
    e = mt.epsilon
    z = np.zeros_like(tp)
    if mt.base == PRECISION_AT_K:  # metrics.py:513
        return tp / mt.k, z + 1.0 / mt.k, z, z, z
    if mt.base == PRECISION:  # :605
        d = tp + fp + e
        return tp / d, (d - tp) / (d * d), -tp / (d * d), z, z
    if mt.base == RECALL:  # :652
        d = tp + fn + e
        return tp / d, (d - tp) / (d * d), z, -tp / (d * d), z
    if mt.base == FBETA:  # :703
        b2 = mt.beta * mt.beta
        num = (1.0 + b2) * tp
        d = (b2 * (tp + fp)) + tp + fn + e
        return (num / d, ((1.0 + b2) * d - num * (1.0 + b2)) / (d * d), -num * b2 / (d * d),
                -num / (d * d), z)
    if mt.base == JACCARD:  # :797
        d = tp + fp + fn + e
        return tp / d, (d - tp) / (d * d), -tp / (d * d), -tp / (d * d), z
    if mt.base in (BALANCED_ACC, GMEAN, HMEAN):  # :843-845, :892-894, :942-944
        dp = tp + fn + e
        dn = tn + fp + e
        tpr = tp / dp
        tnr = tn / dn
        tpr_tp, tpr_fn = (dp - tp) / (dp * dp), -tp / (dp * dp)
        tnr_tn, tnr_fp = (dn - tn) / (dn * dn), -tn / (dn * dn)
        if mt.base == BALANCED_ACC:
            return (tpr + tnr) / 2.0, tpr_tp / 2.0, tnr_fp / 2.0, tpr_fn / 2.0, tnr_tn / 2.0
        if mt.base == GMEAN:
            v = np.sqrt(tpr * tnr)
            with np.errstate(divide="ignore", invalid="ignore"):
                h = 0.5 / v
                return v, h * tnr * tpr_tp, h * tpr * tnr_fp, h * tnr * tpr_fn, h * tpr * tnr_tn
        s = tpr + tnr
        with np.errstate(divide="ignore", invalid="ignore"):
            v = 2.0 * tpr * tnr / s
            d_tpr = 2.0 * tnr * tnr / (s * s)
            d_tnr = 2.0 * tpr * tpr / (s * s)
        return v, d_tpr * tpr_tp, d_tnr * tnr_fp, d_tpr * tpr_fn, d_tnr * tnr_tn
    if mt.base == ACCURACY:  # :416-419
        d = tp + fp + fn + tn
        num = tp + tn
        return num / d, (d - num) / (d * d), -num / (d * d), -num / (d * d), (d - num) / (d * d)
    if mt.base == RECALL_PRECISION_MIX:  # frank_wolfe.py:925-929
        dr = tp + fn + e
        dq = tp + fp + e
        a = mt.alpha
        return ((1 - a) * (tp / dr) + a * (tp / dq),
                (1 - a) * (dr - tp) / (dr * dr) + a * (dq - tp) / (dq * dq),
                -a * tp / (dq * dq), -(1 - a) * tp / (dr * dr), z)
    raise ValueError(f"unknown metric {mt.base}")


def _label_value_and_grad(mt: FwMetric, tp, fp, fn, tn):
    v, gtp, gfp, gfn, gtn = _base_value_and_grad(mt, tp, fp, fn, tn)
    if mt.mixed:
        a = mt.alpha
        v = (1 - a) * (tp / mt.k) + a * v / mt.m
        gtp = (1 - a) / mt.k + a * gtp / mt.m
        gfp, gfn, gtn = a * gfp / mt.m, a * gfn / mt.m, a * gtn / mt.m
    return v, gtp, gfp, gfn, gtn


def metric_value(mt: FwMetric, tp, fp, fn, tn):
    """The arithmetic runs in the arrays' own dtype, as the reference's numpy expressions do
    (float32 confusion vectors for float32 inputs, confusion_matrix.py:181, :212, :226)."""
    if mt.average == "micro":
        return _label_value_and_grad(mt, tp.sum(), fp.sum(), fn.sum(), tn.sum())[0]
    v = _label_value_and_grad(mt, tp, fp, fn, tn)[0]
    return v.mean() if mt.average == "macro" else v.sum()


def metric_value_and_gradient(mt: FwMetric, tp, fp, fn, tn):
    """frank_wolfe.py:368-376: the utility and its gradient in each of the 4 x m entries."""
    m = tp.shape[0]
    if mt.average == "micro":
        v, *g = _label_value_and_grad(mt, tp.sum(), fp.sum(), fn.sum(), tn.sum())
        return (float(v),) + tuple(np.full(m, x) for x in g)
    v, *g = _label_value_and_grad(mt, tp, fp, fn, tn)
    if mt.average == "macro":
        return (float(v.mean()),) + tuple(x / m for x in g)
    return (float(v.sum()),) + tuple(g)


def uniform_search(low, high, step, func):  # utils.py:174-184
    best = low
    best_val = func(low)
    for i in np.arange(low + step, high, step):
        score = func(i)
        if score > best_val:
            best = i
            best_val = score
    return best, best_val


def ternary_search(low, high, eps, func):  # utils.py:187-201
    while high - low > eps:
        mid1 = low + (high - low) / 3
        mid2 = high - (high - low) / 3
        if func(mid1) < func(mid2):
            high = mid2
        else:
            low = mid1
    best = (low + high) / 2
    return best, func(best)


def find_classifier_using_fw(
    y_true, y_proba, metric: FwMetric, k: int, max_iters: int = 100, init_classifier="top",
    maximize: bool = True, normalize_conf_matrix: bool = True, tolerance: float = 1e-6,
    search_for_best_alpha: bool = True, alpha_search_algo: str = "uniform",
    alpha_tolerance: float = 0.001, alpha_uniform_search_step: float = 0.0001,
    skip_tn: bool = False, seed: Optional[int] = None,
) -> Tuple[np.ndarray, np.ndarray, np.ndarray, Dict[str, Any]]:
    """frank_wolfe.py:407-690.  Returns (classifiers_a, classifiers_b, classifiers_proba, meta)."""
    n, m = y_proba.shape
    f32 = np.float32  # types.py:13, the dtype of the classifier tables (:501-503)
    y_freq = np.array(y_true.sum(axis=0), dtype=f32).flatten()
    rng = np.random.default_rng(seed)
    A = np.zeros((max_iters + 1, m), dtype=f32)
    B = np.zeros((max_iters + 1, m), dtype=f32)
    P = np.ones(max_iters + 1, dtype=f32)
    if isinstance(init_classifier, str):
        if init_classifier == "top":  # :505-507
            A[0] = 1.0
            B[0] = -0.5
        elif init_classifier == "random":  # :508-510
            A[0] = rng.random(m)
            B[0] = rng.random(m) - 0.5
        elif init_classifier == "prior":  # :511-516
            A[0] = 1.0 / ((y_freq + 0.1) / y_true.shape[0])
            B[0] = 0.0
        else:
            raise ValueError("Unsupported type of init_classifier")
    else:
        A[0], B[0] = init_classifier

    def predict_conf(i):  # :560-566 / :599-604
        y_pred_i = ref.predict_weighted_per_instance(y_proba, k, th=0.0, a=A[i], b=B[i])
        # exact counts, then the reference's dtype flow: accumulators of y_true's dtype
        # (confusion_matrix.py:181 `dtype if dtype else y_true.dtype`), `/ n` in that dtype (:265-266)
        tp, fp, fn, _ = ref.calculate_confusion_matrix(y_true, y_pred_i, normalize=False, skip_tn=True)
        dt = y_true.dtype
        tp, fp, fn = tp.astype(dt), fp.astype(dt), fn.astype(dt)
        if normalize_conf_matrix:
            tp, fp, fn = tp / n, fp / n, fn / n
        if skip_tn:  # :391-393
            tn = np.full_like(tp, -1)
        else:  # :397
            tn = -tp - fp - fn + (1 if normalize_conf_matrix else n)
        return tp, fp, fn, tn

    tp, fp, fn, tn = predict_conf(0)
    utility_i = metric_value(metric, tp, fp, fn, tn)
    meta = {"alphas": [], "classifiers_utilities": [utility_i], "utilities": [utility_i], "time": time()}
    new_utility = utility_i
    i = 0
    for i in range(1, max_iters + 1):
        old_utility, Gtp, Gfp, Gfn, Gtn = metric_value_and_gradient(metric, tp, fp, fn, tn)
        a_i = Gtp - Gfp - Gfn + Gtn  # :592-596
        b_i = Gfp - Gtn
        A[i] = a_i
        B[i] = b_i
        if not maximize:
            A[i] *= -1
            B[i] *= -1
        tp_i, fp_i, fn_i, tn_i = predict_conf(i)
        utility_i = metric_value(metric, tp_i, fp_i, fn_i, tn_i)
        if search_for_best_alpha:  # :379-404
            def comb(alpha):
                return metric_value(metric, (1 - alpha) * tp + alpha * tp_i, (1 - alpha) * fp + alpha * fp_i,
                                    (1 - alpha) * fn + alpha * fn_i, (1 - alpha) * tn + alpha * tn_i)
            if alpha_search_algo == "uniform":
                alpha, _ = uniform_search(0, 1, alpha_uniform_search_step, comb)
            elif alpha_search_algo == "ternary":
                alpha, _ = ternary_search(0, 1, alpha_tolerance, comb)
            else:
                raise ValueError(f"Unknown search algorithm {alpha_search_algo}")
        else:
            alpha = 2 / (i + 1)
        tp = (1 - alpha) * tp + alpha * tp_i
        fp = (1 - alpha) * fp + alpha * fp_i
        fn = (1 - alpha) * fn + alpha * fn_i
        tn = (1 - alpha) * tn + alpha * tn_i
        new_utility = metric_value(metric, tp, fp, fn, tn)
        if alpha < alpha_tolerance or ((maximize and new_utility - old_utility < tolerance)
                                       or (not maximize and old_utility - new_utility < tolerance)):  # :640-660
            A, B, P = A[:i], B[:i], P[:i]
            break
        meta["alphas"].append(float(alpha))
        meta["classifiers_utilities"].append(utility_i)
        meta["utilities"].append(new_utility)
        P[:i] *= 1 - alpha
        P[i] = alpha
    meta["time"] = time() - meta["time"]
    meta["iters"] = i
    return A, B, P, meta


def draw_classifiers(n: int, classifiers_proba, seed) -> np.ndarray:
    """The classifier index of every row: n sequential ``rng.choice(range(c), p=p)``
    calls (frank_wolfe.py:99, :153)."""
    rng = np.random.default_rng(seed)
    rows = np.arange(len(classifiers_proba))
    return np.array([rng.choice(rows, p=classifiers_proba) for _ in range(n)], dtype=np.int64)


def predict_using_randomized_weighted_classifier(y_proba, k: int, A, B, P, dtype=None, seed=None):
    """frank_wolfe.py:85-172 (numpy and CSR branches)."""
    n, m = y_proba.shape
    cls = draw_classifiers(n, P, seed)
    if isinstance(y_proba, csr_matrix):
        # :127-172 -> numba_csr_functions.py:550-582, :499-546: the ids of the k largest gains in
        # ascending id order (every id when the row has <= k entries), or gain >= 0 when k == 0
        indptr, indices, data = y_proba.indptr, y_proba.indices, y_proba.data
        out_idx, out_ptr = [], [0]
        for i in range(n):
            s, e = indptr[i], indptr[i + 1]
            g = data[s:e] * A[cls[i]][indices[s:e]]
            g = g + B[cls[i]][indices[s:e]]
            ids = indices[s:e]
            if k > 0:
                if g.size > k:
                    # largest gain first, lower position on ties (fixtures avoid boundary ties)
                    order = np.lexsort((np.arange(g.size), -g))[:k]
                    ids = np.sort(ids[order])
            else:
                ids = ids[g >= 0.0]
            out_idx.extend(int(x) for x in ids)
            out_ptr.append(len(out_idx))
        return csr_matrix((np.ones(len(out_idx), dtype=dtype if dtype else y_proba.dtype),
                           np.array(out_idx, dtype=y_proba.indices.dtype),
                           np.array(out_ptr, dtype=y_proba.indptr.dtype)), shape=(n, m))
    y_pred = np.zeros(y_proba.shape, dtype=y_proba.dtype if dtype is None else dtype)
    for i in range(n):  # :97-108
        g = y_proba[i] * A[cls[i]] + B[cls[i]]
        if k > 0:
            order = np.lexsort((np.arange(m), -g))[:k]
            y_pred[i, order] = 1.0
        else:
            y_pred[i, g > 0] = 1.0
    return y_pred

"""CPU oracle for coverage BCA -- TEST INFRASTRUCTURE.

A numpy restatement of ``predict_optimizing_coverage_using_bc`` for CSR input
(/root/reference/xcolumns/block_coordinate.py:600-701 with ``_bc_for_coverage_step_csr`` :539-582,
``_calculate_coverage_utility`` :585-597 and ``numba_calculate_prod_csr_mat_mul_ones_minus_mat``,
numba_csr_functions.py:324-382).  The statistic is the per-label probability of NOT being covered,
Ef_j = prod_i (1 - pred_ij * eta_ij), kept in float64 (types.py:14) and updated multiplicatively.

Parity status: PINNED by ``tests/test_oracle_golden.py`` against fixtures generated from the reference
itself (``tests/golden/make_golden.py``, CSR only: the reference's dense branch calls ``np.product``,
which numpy 2 no longer has).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
cpu_baseline leg may import this module.
"""
from __future__ import annotations

from time import time
from typing import Optional

import numpy as np
from scipy.sparse import csr_matrix

from . import ref


def failure_probabilities(y_proba: csr_matrix, pred_idx: np.ndarray, k: int) -> np.ndarray:
    """numba_csr_functions.py:324-382 with a = y_pred (ones), b = y_proba: rows in order, every predicted
    label the row stores multiplies Ef by (1 - eta), computed in eta's dtype."""
    n, m = y_proba.shape
    Ef = np.ones(m, dtype=np.float64)
    indptr, indices, data = y_proba.indptr, y_proba.indices, y_proba.data
    one = data.dtype.type(1)
    for i in range(n):
        cols = indices[indptr[i]:indptr[i + 1]]
        p = pred_idx[i * k:(i + 1) * k]
        pos = np.searchsorted(cols, p)
        ok = (pos < cols.size) & (cols[np.minimum(pos, cols.size - 1)] == p)
        sel = pos[ok]
        Ef[cols[sel]] *= one - data[indptr[i] + sel]
    return Ef


def coverage_utility(y_proba: csr_matrix, pred_idx: np.ndarray, Ef: np.ndarray, k: int, alpha: float) -> float:
    """_calculate_coverage_utility, block_coordinate.py:585-597."""
    n, m = y_proba.shape
    cov = 1 - Ef.mean()
    if alpha < 1:
        pred = csr_matrix((np.ones(n * k, dtype=y_proba.dtype), pred_idx, np.arange(n + 1) * k), shape=(n, m))
        tp = ref.calculate_confusion_matrix(y_proba, pred, skip_tn=True)[0].astype(y_proba.dtype)  # :181 dtype
        cov = alpha * cov + (1 - alpha) * (tp / n / k).sum()
    return cov


def predict_optimizing_coverage_using_bc(y_proba: csr_matrix, k: int, alpha: float = 1, tolerance: float = 1e-6,
                                         init_y_pred="top", max_iters: int = 100, shuffle_order: bool = True,
                                         seed: Optional[int] = None):
    """block_coordinate.py:600-701 (CSR).  Returns (y_pred csr_matrix, meta)."""
    n, m = y_proba.shape
    meta = {"utilities": [], "iters": 0, "time": time()}
    greedy = isinstance(init_y_pred, str) and init_y_pred == "greedy"
    if isinstance(init_y_pred, str) and init_y_pred in ("random", "greedy"):
        pred_idx = np.ascontiguousarray(ref.random_at_k_csr((n, m), k, dtype=y_proba.dtype, seed=seed).indices)
    elif isinstance(init_y_pred, str) and init_y_pred == "top":
        pred_idx = np.ascontiguousarray(ref.predict_top_k(y_proba, k).indices)
    else:
        pred_idx = np.ascontiguousarray(init_y_pred.indices).copy()
    indptr, indices, data = y_proba.indptr, y_proba.indices, y_proba.data
    one = data.dtype.type(1)
    rng = np.random.default_rng(seed)
    order = np.arange(n)
    for j in range(1, max_iters + 1):
        if shuffle_order:
            rng.shuffle(order)
        Ef = np.ones(m, dtype=np.float64) if greedy else failure_probabilities(y_proba, pred_idx, k)
        old_cov = coverage_utility(y_proba, pred_idx, Ef, k, alpha)
        for i in order:
            s, e = indptr[i], indptr[i + 1]
            cols, eta = indices[s:e], data[s:e]
            p = pred_idx[i * k:(i + 1) * k]
            if not greedy:  # :561-563
                pos = np.searchsorted(cols, p)
                ok = (pos < cols.size) & (cols[np.minimum(pos, cols.size - 1)] == p)
                Ef[cols[pos[ok]]] /= one - eta[pos[ok]]
            gains = Ef[cols] * eta  # :566
            if alpha < 1:
                gains = alpha * gains + (1 - alpha) * eta / k
            if gains.size > k:
                top = np.lexsort((np.arange(gains.size), -gains))[:k]  # largest first, lower position on ties
                new = np.sort(cols[top])
            else:
                new = np.sort(cols)
            pred_idx[i * k:(i + 1) * k] = new
            pos = np.searchsorted(cols, new)
            Ef[new] *= one - eta[pos]  # :580-582
        Ef = failure_probabilities(y_proba, pred_idx, k)
        new_cov = coverage_utility(y_proba, pred_idx, Ef, k, alpha)
        greedy = False
        meta["iters"] = j
        meta["utilities"].append(new_cov)
        if new_cov <= old_cov + tolerance:
            break
    meta["time"] = time() - meta["time"]
    pred = csr_matrix((np.ones(n * k, dtype=y_proba.dtype), pred_idx.astype(y_proba.indices.dtype),
                       (np.arange(n + 1) * k).astype(y_proba.indptr.dtype)), shape=(n, m))
    return pred, meta

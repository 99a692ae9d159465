"""CPU oracle for the xCOLUMNs BCA / weighted top-k path -- TEST INFRASTRUCTURE.

Python front of ``oracle/xc_oracle.c``: a sequential restatement of the
reference's algorithm (mwydmuch/xCOLUMNs 0.0.3; citations are relative to
``/root/reference/xcolumns/``).  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import this module; the product
package ``xcolumns_amd`` never does.

Parity status: PINNED by ``tests/test_oracle_golden.py`` against fixtures
generated from the reference itself (``tests/golden/make_golden.py``).

The driver functions below restate the *Python* control flow of the reference
(`predict_using_bc_with_0approx`, block_coordinate.py:296-499) and delegate the
per-row work to the C functions, which restate the numba kernels.
"""
from __future__ import annotations

import ctypes
import os
import random
import subprocess
from time import time
from typing import Any, Dict, Optional, Tuple, Union

import numpy as np
from scipy.sparse import csr_matrix

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libxc_oracle.so")

# metric ids, same numbering as the enum in xc_oracle.c
PRECISION_AT_K, PRECISION, RECALL, FBETA, JACCARD, BALANCED_ACC, GMEAN, HMEAN, ACCURACY = range(9)


class _Metric(ctypes.Structure):
    _fields_ = [
        ("base", ctypes.c_int32),
        ("mixed", ctypes.c_int32),
        ("epsilon", ctypes.c_double),
        ("beta", ctypes.c_double),
        ("kf", ctypes.c_double),
        ("alpha", ctypes.c_double),
        ("mf", ctypes.c_double),
    ]


def make_metric(base: int, epsilon: float = 1e-9, beta: float = 1.0, k: float = 1.0,
                mixed: bool = False, alpha: float = 1.0, m: float = 1.0) -> _Metric:
    return _Metric(int(base), int(bool(mixed)), float(epsilon), float(beta), float(k),
                   float(alpha), float(m))


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Returns the .so path."""
    src_newer = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("xc_oracle.c", "xc_oracle_impl.h", "xc_oracle_mt_impl.h", "Makefile")
    )
    if force or src_newer:
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(arr: Optional[np.ndarray]):
    return None if arr is None else arr.ctypes.data_as(ctypes.c_void_p)


def _sfx(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32"
    if dtype == np.float64:
        return "_f64"
    raise ValueError(f"oracle supports float32/float64 data, got {dtype}")


def _csr_parts(mat: csr_matrix) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    return (
        np.ascontiguousarray(mat.indptr, dtype=np.int32),
        np.ascontiguousarray(mat.indices, dtype=np.int32),
        np.ascontiguousarray(mat.data),
    )


# ---------------------------------------------------------------------------
# weighted per-instance prediction (weighted_prediction.py:91-188)
# ---------------------------------------------------------------------------

def predict_weighted_per_instance(y_proba, k: int, th: float = 0.0, a=None, b=None,
                                  keep_scores: bool = False):
    if isinstance(y_proba, csr_matrix):
        n, m = y_proba.shape
        dt = y_proba.dtype
        indptr, indices, data = _csr_parts(y_proba)
        # weighted_prediction.py:72-75: weights are cast to y_proba.dtype
        a_ = None if a is None else np.ascontiguousarray(a, dtype=dt)
        b_ = None if b is None else np.ascontiguousarray(b, dtype=dt)
        if k > 0:
            out_idx = np.empty(n * k, dtype=np.int32)
            out_dat = np.empty(n * k, dtype=dt)
            getattr(lib(), "oracle_topk_csr" + _sfx(dt))(
                ctypes.c_int64(n), _p(indptr), _p(indices), _p(data), ctypes.c_int(k),
                _p(a_), _p(b_), ctypes.c_int(int(keep_scores)), _p(out_idx), _p(out_dat))
            out_indptr = (np.arange(n + 1, dtype=np.int64) * k).astype(y_proba.indptr.dtype)
            return csr_matrix((out_dat, out_idx.astype(y_proba.indices.dtype), out_indptr),
                              shape=(n, m))
        out_indptr = np.empty(n + 1, dtype=np.int32)
        out_idx = np.empty(max(1, indices.size), dtype=np.int32)
        th_c = ctypes.c_float(th) if dt == np.float32 else ctypes.c_double(th)
        getattr(lib(), "oracle_threshold_csr" + _sfx(dt))(
            ctypes.c_int64(n), _p(indptr), _p(indices), _p(data), th_c, _p(a_), _p(b_),
            _p(out_indptr), _p(out_idx))
        nnz = int(out_indptr[-1])
        return csr_matrix((np.ones(nnz, dtype=dt), out_idx[:nnz].astype(y_proba.indices.dtype),
                           out_indptr.astype(y_proba.indptr.dtype)), shape=(n, m))

    # dense: gains are formed by numpy itself so dtype promotion is numpy's
    # (weighted_prediction.py:37-41); y_pred keeps y_proba's dtype (:35)
    y_proba = np.asarray(y_proba)
    n, m = y_proba.shape
    gains = y_proba
    if a is not None:
        gains = gains * a
    if b is not None:
        gains = gains + b
    gains = np.ascontiguousarray(gains)
    gdt = gains.dtype
    pred_g = np.empty((n, m), dtype=gdt)
    th_c = ctypes.c_float(th) if gdt == np.float32 else ctypes.c_double(th)
    getattr(lib(), "oracle_topk_dense" + _sfx(gdt))(
        ctypes.c_int64(n), ctypes.c_int64(m), _p(gains), ctypes.c_int(k), th_c,
        ctypes.c_int(int(keep_scores)), _p(pred_g))
    return pred_g.astype(y_proba.dtype)


def predict_top_k(y_proba, k: int, keep_scores: bool = False):
    return predict_weighted_per_instance(y_proba, k, keep_scores=keep_scores)


def predict_top_k_threads(y_proba: csr_matrix, k: int, n_threads: int) -> csr_matrix:
    """predict_top_k on a csr_matrix with the loop over rows split over `n_threads` OpenMP threads:
    the reference under XCOLUMNS_NUMBA_PARALLEL=1 (numba_csr_functions.py:604 `prange`).  Baseline
    leg of bench.py; same result as the serial form."""
    n, m = y_proba.shape
    dt = y_proba.dtype
    indptr, indices, data = _csr_parts(y_proba)
    out_idx = np.empty(n * k, dtype=np.int32)
    out_dat = np.empty(n * k, dtype=dt)
    getattr(lib(), "oracle_topk_csr_mt" + _sfx(dt))(
        ctypes.c_int64(n), _p(indptr), _p(indices), _p(data), ctypes.c_int(k), None, None, ctypes.c_int(0),
        _p(out_idx), _p(out_dat), ctypes.c_int(int(n_threads)))
    out_indptr = (np.arange(n + 1, dtype=np.int64) * k).astype(y_proba.indptr.dtype)
    return csr_matrix((out_dat, out_idx.astype(y_proba.indices.dtype), out_indptr), shape=(n, m))


def calculate_confusion_matrix_threads(y_true: csr_matrix, y_pred: csr_matrix, n_threads: int):
    """(tp, fp, fn) of calculate_confusion_matrix on CSR input, rows split over `n_threads` OpenMP
    threads with per-thread column vectors (numba_csr_functions.py:166 / :242 `prange`).  Sums are
    formed in another order than the serial pass: equal to rounding only."""
    n, m = y_true.shape
    dt = y_true.dtype
    tp, fp, fn = (np.empty(m, dtype=np.float64) for _ in range(3))
    t_indptr, t_indices, t_data = _csr_parts(y_true)
    p_indptr, p_indices, p_data = _csr_parts(y_pred.astype(dt))
    getattr(lib(), "oracle_confusion_csr_mt" + _sfx(dt))(
        ctypes.c_int64(n), ctypes.c_int64(m), _p(t_indptr), _p(t_indices), _p(t_data), _p(p_indptr), _p(p_indices),
        _p(p_data), _p(tp), _p(fp), _p(fn), ctypes.c_int(int(n_threads)))
    return tp, fp, fn


# ---------------------------------------------------------------------------
# confusion matrix (confusion_matrix.py:364-399), axis=0
# ---------------------------------------------------------------------------

def calculate_confusion_matrix(y_true, y_pred, normalize: bool = False, skip_tn: bool = False):
    n, m = y_true.shape
    tp = np.empty(m, dtype=np.float64)
    fp = np.empty(m, dtype=np.float64)
    fn = np.empty(m, dtype=np.float64)
    if isinstance(y_true, csr_matrix):
        dt = y_true.dtype
        t_indptr, t_indices, t_data = _csr_parts(y_true)
        p_indptr, p_indices, p_data = _csr_parts(y_pred.astype(dt))
        getattr(lib(), "oracle_confusion_csr" + _sfx(dt))(
            ctypes.c_int64(n), ctypes.c_int64(m), _p(t_indptr), _p(t_indices), _p(t_data),
            _p(p_indptr), _p(p_indices), _p(p_data), _p(tp), _p(fp), _p(fn))
    else:
        dt = y_true.dtype
        yt = np.ascontiguousarray(y_true)
        yp = np.ascontiguousarray(y_pred, dtype=dt)
        getattr(lib(), "oracle_confusion_dense" + _sfx(dt))(
            ctypes.c_int64(n), ctypes.c_int64(m), _p(yt), _p(yp), _p(tp), _p(fp), _p(fn))
    if normalize:  # confusion_matrix.py:265-266
        tp, fp, fn = tp / n, fp / n, fn / n
    if skip_tn:  # :391-393
        tn = tp.copy()
        tn[:] = -1
    else:  # :397
        tn = -tp - fp - fn + (1.0 if normalize else n)
    return tp, fp, fn, tn


def metric_values(metric: _Metric, tp, fp, fn, tn) -> np.ndarray:
    m = tp.shape[0]
    out = np.empty(m, dtype=np.float64)
    lib().xc_oracle_metric_values(
        ctypes.byref(metric), ctypes.c_int64(m),
        _p(np.ascontiguousarray(tp)), _p(np.ascontiguousarray(fp)),
        _p(np.ascontiguousarray(fn)), _p(np.ascontiguousarray(tn)), _p(out))
    return out


def calculate_utility(metric: _Metric, aggregation: str, tp, fp, fn, tn) -> float:
    """_calculate_utility, block_coordinate.py:54-90."""
    vals = metric_values(metric, tp, fp, fn, tn)
    if aggregation == "sum":
        return vals.sum()
    if aggregation == "mean":
        return vals.mean()
    raise ValueError(f"Unsupported utility aggregation function: {aggregation}")


# ---------------------------------------------------------------------------
# initial predictions (block_coordinate.py:28-51, utils.py:103-136)
# ---------------------------------------------------------------------------

def random_at_k_np(shape, k, dtype=None, seed=None) -> np.ndarray:
    """utils.py:103-116."""
    n, m = shape
    y_pred = np.zeros(shape, dtype=dtype)
    rng = np.random.default_rng(seed)
    labels_range = np.arange(m)
    for i in range(n):
        y_pred[i, rng.choice(labels_range, k, replace=False, shuffle=False)] = 1.0
    return y_pred


def random_at_k_csr(shape, k, dtype=None, seed=None) -> csr_matrix:
    """utils.py:119-136 -> numba_random_at_k (numba_csr_functions.py:92-112) with
    numba_fast_random_choice (:77-89): a partial Fisher-Yates shuffle driven by
    the `random` module.  Outside numba that is CPython's Mersenne Twister
    (the stream numba's own generator produces is not claimed)."""
    n, m = shape
    if seed is not None:
        random.seed(seed)
    indices = np.zeros(n * k, dtype=np.int32)
    for i in range(n):
        index = np.arange(m, dtype=np.int32)
        for t in range(k):
            j = random.randint(t, m - 1)
            index[t], index[j] = index[j], index[t]
        indices[i * k:(i + 1) * k] = index[:k]
    indptr = (np.arange(n + 1) * k).astype(np.int32)
    mat = csr_matrix((np.ones(n * k, dtype=np.float32).astype(dtype), indices, indptr),
                     dtype=dtype, shape=shape)
    mat.sort_indices()
    return mat


# ---------------------------------------------------------------------------
# predict_using_bc_with_0approx (block_coordinate.py:296-499)
# ---------------------------------------------------------------------------

def predict_using_bc_with_0approx(
    y_proba,
    metric: _Metric,
    k: int,
    metric_aggregation: str = "mean",
    normalize_conf_matrix: bool = True,
    maximize: bool = True,
    tolerance: float = 1e-6,
    init_y_pred: Union[str, np.ndarray, csr_matrix] = "top",
    max_iters: int = 100,
    shuffle_order: bool = True,
    skip_tn: bool = False,
    seed: Optional[int] = None,
    utility_metric: Optional[_Metric] = None,
) -> Tuple[Any, Dict[str, Any]]:
    """`metric` drives the per-row gains (it carries metric_kwargs,
    block_coordinate.py:267-278); `utility_metric` is what _calculate_utility
    evaluates -- the reference calls it WITHOUT metric_kwargs (:438-445,
    :469-476, :63), so it defaults to the same formula with default kwargs
    only when the caller passes it that way."""
    if utility_metric is None:
        utility_metric = metric
    meta: Dict[str, Any] = {"utilities": [], "iters": 0, "time": time()}
    is_csr = isinstance(y_proba, csr_matrix)
    n_rows, m = y_proba.shape
    n = n_rows if normalize_conf_matrix else 1          # :403-405
    dt = y_proba.dtype

    greedy = isinstance(init_y_pred, str) and init_y_pred == "greedy"   # :409
    if isinstance(init_y_pred, str) and init_y_pred in ("random", "greedy"):
        y_pred = (random_at_k_csr if is_csr else random_at_k_np)((n_rows, m), k, dtype=dt, seed=seed)
    elif isinstance(init_y_pred, str) and init_y_pred == "top":
        y_pred = predict_top_k(y_proba, k)
    else:
        y_pred = init_y_pred                              # used as is, :46

    if is_csr:
        if k <= 0:
            raise NotImplementedError("oracle BCA on CSR restates the k > 0 path only")
        t_indptr, t_indices, t_data = _csr_parts(y_proba)
        if (np.diff(t_indptr) < k).any():
            raise NotImplementedError("oracle BCA on CSR requires >= k stored entries per row")
        if not (np.diff(y_pred.indptr) == k).all():
            raise ValueError("y_pred must hold exactly k entries per row")
        p_indices = np.ascontiguousarray(y_pred.indices, dtype=np.int32).copy()
        p_data = np.ascontiguousarray(y_pred.data, dtype=dt)
        sweep = getattr(lib(), "oracle_bca_sweep_csr" + _sfx(dt))
    else:
        y_proba_c = np.ascontiguousarray(y_proba)
        y_pred = np.ascontiguousarray(y_pred, dtype=dt)
        sweep = getattr(lib(), "oracle_bca_sweep_dense" + _sfx(dt))

    def current_pred():
        if is_csr:
            return csr_matrix((p_data, p_indices, y_pred.indptr), shape=(n_rows, m))
        return y_pred

    rng = np.random.default_rng(seed)                     # :413
    order = np.arange(n)                                  # :414 (int64)
    for j in range(1, max_iters + 1):
        if shuffle_order:
            rng.shuffle(order)                            # :418-419
        if greedy:                                        # :423-427
            tp = np.zeros(m); fp = np.zeros(m); fn = np.zeros(m); tn = np.zeros(m)
        else:                                             # :430-436
            tp, fp, fn, tn = calculate_confusion_matrix(y_proba, current_pred(), skip_tn=skip_tn)
        old_utility = calculate_utility(utility_metric, metric_aggregation, tp / n, fp / n, fn / n, tn / n)

        order_c = np.ascontiguousarray(order, dtype=np.int64)
        if is_csr:
            sweep(ctypes.c_int64(n_rows), ctypes.c_int64(m), ctypes.c_int64(order_c.size),
                  _p(order_c), _p(t_indptr), _p(t_indices), _p(t_data), _p(p_indices),
                  _p(p_data), ctypes.c_int(k), _p(tp), _p(fp), _p(fn), _p(tn),
                  ctypes.byref(metric), ctypes.c_int(int(greedy)), ctypes.c_int(int(maximize)),
                  ctypes.c_int(int(skip_tn)))
        else:
            sweep(ctypes.c_int64(n_rows), ctypes.c_int64(m), ctypes.c_int64(order_c.size),
                  _p(order_c), _p(y_proba_c), _p(y_pred), ctypes.c_int(k), _p(tp), _p(fp),
                  _p(fn), _p(tn), ctypes.byref(metric), ctypes.c_int(int(greedy)),
                  ctypes.c_int(int(maximize)), ctypes.c_int(int(skip_tn)))

        tp, fp, fn, tn = calculate_confusion_matrix(y_proba, current_pred(), skip_tn=skip_tn)  # :465
        new_utility = calculate_utility(utility_metric, metric_aggregation, tp / n, fp / n, fn / n, tn / n)
        greedy = False
        meta["iters"] = j
        meta["utilities"].append(float(new_utility))
        if (maximize and new_utility - old_utility < tolerance) or (
            not maximize and new_utility - old_utility > tolerance
        ):                                                # :486-493
            break

    meta["time"] = time() - meta["time"]
    if is_csr:
        out = csr_matrix((p_data, p_indices.astype(y_pred.indices.dtype), y_pred.indptr),
                         shape=(n_rows, m))
        return out, meta
    return y_pred, meta

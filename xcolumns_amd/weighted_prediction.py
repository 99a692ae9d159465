"""Weighted per-instance prediction on MI355X.

Same functions, arguments, validation and return types as
/root/reference/xcolumns/weighted_prediction.py (:91-188, :196-220 and the
closed-form weightings :223-533); the row-wise top-k runs in
``xc_topk_csr`` / ``xc_topk_dense`` (csrc/xc_topk.hip, csrc/xc_dense.hip).
"""
from __future__ import annotations

import ctypes
from time import time
from typing import Optional, Tuple, Union

import numpy as np
import torch
from scipy.sparse import csr_matrix

from . import _device as D
from . import _lib
from .types import DenseMatrix, DType, Matrix, is_dense, is_matrix


# ---------------------------------------------------------------------------
# device-level entry points (torch tensors in HBM in, torch tensors out)
# ---------------------------------------------------------------------------

def topk_csr_device(csr: D.DeviceCSR, k: int, a: Optional[torch.Tensor] = None,
                    b: Optional[torch.Tensor] = None, keep_scores: bool = False,
                    want_eta: bool = False, out_sel: Optional[torch.Tensor] = None):
    """``xc_topk_csr`` on a device-resident CSR matrix.  Returns
    (indices[n*k] int32, data[n*k], eta[n*k] or None); `out_sel` (uint8 per stored
    entry) is filled with the chosen-entry flags when given."""
    n = csr.n
    dev = csr.data.device
    out_idx = torch.empty(n * k, dtype=torch.int32, device=dev)
    out_dat = torch.empty(n * k, dtype=csr.data.dtype, device=dev)
    out_eta = torch.empty(n * k, dtype=csr.data.dtype, device=dev) if want_eta else None
    if a is not None and b is not None:
        # both weights: interleave them (an O(m) copy) so a candidate costs one gather, not two -- same gains
        ab = torch.stack([a, b], dim=1).contiguous()
        _lib.call("xc_topk_csr_ab", n, D.ptr(csr.indptr), D.ptr(csr.indices), D.ptr(csr.data), csr.code,
                  int(csr.max_row_nnz), int(k), D.ptr(ab), int(bool(keep_scores)),
                  D.ptr(out_idx), D.ptr(out_dat), D.ptr(out_eta), D.ptr(out_sel), D.stream())
        return out_idx, out_dat, out_eta
    _lib.call("xc_topk_csr", n, D.ptr(csr.indptr), D.ptr(csr.indices), D.ptr(csr.data), csr.code,
              int(csr.max_row_nnz), int(k), D.ptr(a), D.ptr(b), int(bool(keep_scores)),
              D.ptr(out_idx), D.ptr(out_dat), D.ptr(out_eta), D.ptr(out_sel), D.stream())
    return out_idx, out_dat, out_eta


def threshold_csr_device(csr: D.DeviceCSR, th: float, a=None, b=None):
    """k == 0: entries with gain >= th.  Returns (indptr[n+1] int32, indices int32)."""
    n = csr.n
    dev = csr.data.device
    counts = torch.zeros(n, dtype=torch.int32, device=dev)
    _lib.call("xc_threshold_count_csr", n, D.ptr(csr.indptr), D.ptr(csr.indices), D.ptr(csr.data),
              csr.code, float(th), D.ptr(a), D.ptr(b), D.ptr(counts), D.stream())
    indptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    indptr[1:] = torch.cumsum(counts, 0, dtype=torch.int64).to(torch.int32)
    nnz = int(indptr[-1].item()) if n > 0 else 0
    indices = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    _lib.call("xc_threshold_fill_csr", n, D.ptr(csr.indptr), D.ptr(csr.indices), D.ptr(csr.data),
              csr.code, float(th), D.ptr(a), D.ptr(b), D.ptr(indptr), D.ptr(indices), D.stream())
    return indptr, indices[:nnz]


def topk_dense_device(gains: torch.Tensor, k: int, th: float, keep_scores: bool,
                      out_dtype: torch.dtype) -> torch.Tensor:
    """``xc_topk_dense``: gains (n x m, row-contiguous, on the GPU) -> y_pred (n x m)."""
    n, m = gains.shape
    if gains.stride(1) != 1:
        gains = gains.contiguous()
    y_pred = torch.empty((n, m), dtype=out_dtype, device=gains.device)
    _lib.call("xc_topk_dense", n, m, gains.stride(0) if n > 1 else m, D.ptr(gains), D.dtype_code(gains.dtype),
              int(k), float(th), int(bool(keep_scores)), D.ptr(y_pred), D.dtype_code(out_dtype), D.stream())
    return y_pred


# ---------------------------------------------------------------------------
# reference API
# ---------------------------------------------------------------------------

def _predict_weighted_per_instance_dense(y_proba, k, th=0.0, a=None, b=None, keep_scores=False, dtype=None):
    """weighted_prediction.py:25-60.  gains = y_proba (* a) (+ b) with the input
    framework's own type promotion (numpy promotes float32 * float64 to float64,
    :37-41), then the row top-k kernel."""
    dev = D.require_gpu()
    is_torch = isinstance(y_proba, torch.Tensor)
    y = y_proba if is_torch else torch.from_numpy(np.ascontiguousarray(y_proba))
    D.dtype_code(y.dtype)
    out_dtype = y.dtype if dtype is None else D.torch_dtype(dtype)
    D.dtype_code(out_dtype)
    gains = y.to(dev)
    if a is not None:
        gains = gains * D.to_device(a, device=dev)
    if b is not None:
        gains = gains + D.to_device(b, device=dev)
    D.dtype_code(gains.dtype)
    y_pred = topk_dense_device(gains, k, th, keep_scores, out_dtype)
    if is_torch:
        return y_pred.to(y_proba.device)
    return y_pred.cpu().numpy()


def _predict_weighted_per_instance_csr(y_proba: csr_matrix, k, th=0.0, a=None, b=None,
                                       keep_scores=False, dtype=None) -> csr_matrix:
    """weighted_prediction.py:63-88 -> numba_csr_functions.py:585-655."""
    dev = D.require_gpu()
    n, m = tuple(y_proba.shape)
    csr = D.as_device_csr(y_proba, dev)
    tdt = csr.data.dtype
    # weights are cast to y_proba's dtype (:72-75)
    a_d = None if a is None else D.to_device(a, dtype=tdt, device=dev)
    b_d = None if b is None else D.to_device(b, dtype=tdt, device=dev)
    if not isinstance(y_proba, csr_matrix):
        # resident in HBM (DeviceCSR / torch sparse_csr): the prediction stays there, same kind of object
        out_dt = tdt if dtype is None else D.torch_dtype(dtype)
        if k > 0:
            idx, dat, _ = topk_csr_device(csr, k, a_d, b_d, keep_scores)
            return D.fixed_width_prediction(y_proba, idx, k, n, m, values=dat.to(out_dt))
        indptr, idx = threshold_csr_device(csr, th, a_d, b_d)
        ones = torch.ones(idx.numel(), dtype=out_dt, device=dev)
        if isinstance(y_proba, D.DeviceCSR):
            return D.DeviceCSR.from_parts(indptr, idx, ones, (n, m), dev, check=False)
        idt = y_proba.crow_indices().dtype
        return torch.sparse_csr_tensor(indptr.to(idt), idx.to(idt), ones, size=(n, m))
    if k > 0:
        idx, dat, _ = topk_csr_device(csr, k, a_d, b_d, keep_scores)
        out_indptr = (np.arange(n + 1, dtype=np.int64) * k).astype(y_proba.indptr.dtype)
        out = csr_matrix((dat.cpu().numpy(), idx.cpu().numpy().astype(y_proba.indices.dtype, copy=False),
                          out_indptr), shape=(n, m), dtype=dtype)
    else:
        indptr, idx = threshold_csr_device(csr, th, a_d, b_d)
        idx_h = idx.cpu().numpy().astype(y_proba.indices.dtype, copy=False)
        out = csr_matrix((np.ones(idx_h.size, dtype=y_proba.dtype), idx_h,
                          indptr.cpu().numpy().astype(y_proba.indptr.dtype, copy=False)),
                         shape=(n, m), dtype=dtype)
    return out


def predict_weighted_per_instance(
    y_proba: Matrix,
    k: int,
    th: float = 0.0,
    a: Optional[DenseMatrix] = None,
    b: Optional[DenseMatrix] = None,
    dtype: Optional[DType] = None,
    keep_scores: bool = False,
    return_meta: bool = False,
    return_weights: bool = False,
) -> Union[Matrix, Tuple[Matrix, dict]]:
    r"""For each row of `y_proba` compute gains :math:`g = a \odot \eta_i + b` and
    predict the `k` labels with the highest gains (or, when `k` is 0, every label
    whose gain is at least `th`).  Same contract as the reference's function of
    this name (weighted_prediction.py:91-188): the result has the type, shape and
    dtype of `y_proba`; with `return_meta` a ``(y_pred, meta)`` tuple."""
    if not is_matrix(y_proba):
        raise ValueError("y_proba must be either np.ndarray, torch.Tensor, or csr_matrix")
    if len(y_proba.shape) == 1:
        y_proba = y_proba.reshape(1, -1)
    elif len(y_proba.shape) > 2:
        raise ValueError("y_proba must be 1d or 2d")
    if not isinstance(k, int):
        raise ValueError("k must be an integer")
    n, m = y_proba.shape
    if a is not None:
        if not is_dense(a):
            raise ValueError("a must be np.ndarray or torch.Tensor")
        if a.shape != (m,):
            raise ValueError("a must be of shape (y_proba[1],)")
    if b is not None:
        if not is_dense(b):
            raise ValueError("b must be np.ndarray or torch.Tensor")
        if b.shape != (m,):
            raise ValueError("b must be of shape (y_proba[1],)")

    if return_meta:
        meta = {"iters": 1, "time": time()}

    if is_dense(y_proba):
        y_pred = _predict_weighted_per_instance_dense(y_proba, k, th=th, a=a, b=b, dtype=dtype,
                                                      keep_scores=keep_scores)
    else:
        y_pred = _predict_weighted_per_instance_csr(y_proba, k, th=th, a=a, b=b, dtype=dtype,
                                                    keep_scores=keep_scores)

    if return_meta:
        meta["time"] = time() - meta["time"]
        if return_weights:
            meta["a"] = a
            meta["b"] = b
        return y_pred, meta
    return y_pred


def predict_top_k(y_proba: Matrix, k: int, dtype: Optional[DType] = None, keep_scores: bool = False,
                  return_meta: bool = False):
    """Top-`k` labels of every row (weighted_prediction.py:196-220)."""
    return predict_weighted_per_instance(y_proba, k=k, dtype=dtype, keep_scores=keep_scores,
                                         return_meta=return_meta)


def _check_priors(priors, y_proba):
    if priors.shape[0] != y_proba.shape[1]:
        raise ValueError("priors must be of shape (y_proba[1],)")


def predict_optimizing_macro_recall(y_proba, k, priors, epsilon: float = 1e-6, keep_scores=False,
                                    dtype=None, return_meta=False, return_weights=False):
    """a = 1 / (priors + epsilon)  (weighted_prediction.py:223-264)."""
    _check_priors(priors, y_proba)
    return predict_weighted_per_instance(y_proba, k=k, a=1.0 / (priors + epsilon), dtype=dtype,
                                         keep_scores=keep_scores, return_meta=return_meta,
                                         return_weights=return_weights)


def predict_optimizing_macro_balanced_accuracy(y_proba, k, priors, epsilon: float = 1e-6, dtype=None,
                                               return_meta=False):
    """gains = eta / pi - (1 - eta) / (1 - pi), pi = priors + epsilon
    (weighted_prediction.py:267-368).  The gains are linear in eta:
    a = 1/pi + 1/(1 - pi), b = -1/(1 - pi) -- they are formed element-wise here
    with the reference's own expression and handed to the top-k kernels."""
    _check_priors(priors, y_proba)
    if return_meta:
        meta = {"iters": 1, "time": time()}
    if not is_matrix(y_proba):
        raise ValueError("y_proba must be either np.ndarray, torch.Tensor, or csr_matrix")
    dev = D.require_gpu()
    pri = priors + epsilon
    if is_dense(y_proba):
        is_torch = isinstance(y_proba, torch.Tensor)
        y = (y_proba if is_torch else torch.from_numpy(np.ascontiguousarray(y_proba))).to(dev)
        p = D.to_device(pri, device=dev)
        gains = y / p - (1 - y) / (1 - p)
        out_dtype = y.dtype if dtype is None else D.torch_dtype(dtype)
        y_pred = topk_dense_device(gains, k, 0.0, False, out_dtype)
        y_pred = y_pred.to(y_proba.device) if is_torch else y_pred.cpu().numpy()
    else:
        n, m = y_proba.shape
        if k <= 0:
            raise ValueError("k must be > 0 for sparse y_proba")
        csr = D.DeviceCSR.from_scipy(y_proba, dev)
        p = D.to_device(np.asarray(pri), device=dev)
        marg = p[csr.indices.long()]
        # row_data / row_marginals - (1 - row_data) / (1 - row_marginals)  (numba_csr_functions.py:678-679)
        gains = csr.data / marg - (1 - csr.data) / (1 - marg)
        gcsr = D.DeviceCSR(csr.indptr, csr.indices, gains.to(csr.data.dtype).contiguous(), csr.shape, csr.max_row_nnz)
        idx, _, _ = topk_csr_device(gcsr, k)
        out_indptr = (np.arange(n + 1, dtype=np.int64) * k).astype(y_proba.indptr.dtype)
        y_pred = csr_matrix((np.ones(n * k, dtype=y_proba.dtype),
                             idx.cpu().numpy().astype(y_proba.indices.dtype, copy=False), out_indptr),
                            shape=(n, m), dtype=y_proba.dtype if dtype is None else dtype)
    if return_meta:
        meta["time"] = time() - meta["time"]
        return y_pred, meta
    return y_pred


def predict_log_weighted_per_instance(y_proba, k, priors, epsilon: float = 1e-9, keep_scores=False,
                                      dtype=None, return_meta=False, return_weights=False):
    """a = -log(priors + epsilon)  (weighted_prediction.py:371-416)."""
    _check_priors(priors, y_proba)
    weights = -np.log(priors + epsilon) if isinstance(priors, np.ndarray) else -torch.log(priors + epsilon)
    return predict_weighted_per_instance(y_proba, k=k, a=weights, keep_scores=keep_scores, dtype=dtype,
                                         return_meta=return_meta, return_weights=return_weights)


def predict_power_law_weighted_per_instance(y_proba, k, priors, beta: float, epsilon: float = 1e-9,
                                            keep_scores=False, dtype=None, return_meta=False,
                                            return_weights=False):
    """a = (priors + epsilon) ** -beta  (weighted_prediction.py:419-465)."""
    _check_priors(priors, y_proba)
    return predict_weighted_per_instance(y_proba, k=k, a=(priors + epsilon) ** -beta, keep_scores=keep_scores,
                                         dtype=dtype, return_meta=return_meta, return_weights=return_weights)


def predict_optimizing_instance_precision(y_proba, k, keep_scores=False, dtype=None, return_meta=False):
    """Top-k (weighted_prediction.py:468-497)."""
    if k <= 0:
        raise ValueError("k must be > 0")
    return predict_top_k(y_proba, k=k, keep_scores=keep_scores, dtype=dtype, return_meta=return_meta)


def predict_optimizing_instance_propensity_scored_precision(y_proba, k, inverse_propensities=None,
                                                            propensities=None, keep_scores=False, dtype=None,
                                                            return_meta=False, return_weights=False):
    """a = inverse propensities, or 1 / propensities with zeros mapped to 1
    (weighted_prediction.py:500-533)."""
    if inverse_propensities is not None:
        if inverse_propensities.shape[0] != y_proba.shape[1]:
            raise ValueError("inverse_propensities must be of shape (y_proba[1],)")
    elif propensities is not None:
        if propensities.shape[0] != y_proba.shape[1]:
            raise ValueError("propensities must be of shape (y_proba[1],)")
        propensities[propensities == 0] = 1.0
        inverse_propensities = 1.0 / propensities
    else:
        raise ValueError("either inverse_propensities or propensities must be provided")
    return predict_weighted_per_instance(y_proba, k=k, a=inverse_propensities, keep_scores=keep_scores,
                                         dtype=dtype, return_meta=return_meta, return_weights=return_weights)

"""Per-label confusion statistics on MI355X.

Same public surface as /root/reference/xcolumns/confusion_matrix.py:
``ConfusionMatrix`` (:16-152), ``calculate_tp/fp/fn`` (:271-361) and
``calculate_confusion_matrix`` (:364-399).  The column sums run in
``xc_confusion_csr`` / ``xc_confusion_dense`` (csrc/xc_confusion.hip): one
fused pass with float64 atomics instead of the reference's three merge passes.

Divergences, all on purpose: sums are always accumulated in float64 and cast to
`dtype` at the end (the reference accumulates in `dtype`, which defaults to
y_true's); torch tensors are supported (the reference raises TypeError for
them, SURVEY.md section 8 a-10 vi) and come back on the input's device.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Union

import numpy as np
import torch
from scipy.sparse import csr_matrix

from . import _device as D
from . import _lib
from .types import DenseMatrix, DType, Matrix, Number, is_dense


class ConfusionMatrix:
    """tp / fp / fn / tn (numbers or per-label vectors) with element-wise
    arithmetic; unpacks as ``tp, fp, fn, tn`` so it can be splatted into the
    ``*_on_conf_matrix`` metric functions."""

    __slots__ = ("tp", "fp", "fn", "tn")

    def __init__(self, tp, fp, fn, tn):
        self.tp, self.fp, self.fn, self.tn = tp, fp, fn, tn

    def _entries(self):
        return (self.tp, self.fp, self.fn, self.tn)

    def __iter__(self):
        return iter(self._entries())

    def __eq__(self, other: object) -> bool:
        if not isinstance(other, ConfusionMatrix):
            return False
        for x, y in zip(self._entries(), other._entries()):
            same = x == y
            if not (bool(same) if isinstance(same, (bool, np.bool_)) else bool(same.all())):
                return False
        return True

    def _map2(self, other, op) -> "ConfusionMatrix":
        if isinstance(other, ConfusionMatrix):
            return ConfusionMatrix(*(op(x, y) for x, y in zip(self._entries(), other._entries())))
        return ConfusionMatrix(*(op(x, other) for x in self._entries()))

    def _imap2(self, other, op) -> "ConfusionMatrix":
        res = self._map2(other, op)
        self.tp, self.fp, self.fn, self.tn = res.tp, res.fp, res.fn, res.tn
        return self

    def __add__(self, other):
        return self._map2(other, lambda x, y: x + y)

    def __iadd__(self, other):
        return self._imap2(other, lambda x, y: x + y)

    def __sub__(self, other):
        return self._map2(other, lambda x, y: x - y)

    def __isub__(self, other):
        return self._imap2(other, lambda x, y: x - y)

    def __mul__(self, other):
        return self._map2(other, lambda x, y: x * y)

    def __imul__(self, other):
        return self._imap2(other, lambda x, y: x * y)

    def __truediv__(self, other):
        return self._map2(other, lambda x, y: x / y)

    def __itruediv__(self, other):
        return self._imap2(other, lambda x, y: x / y)

    def __floordiv__(self, other):
        return self._map2(other, lambda x, y: x // y)

    def __ifloordiv__(self, other):
        return self._imap2(other, lambda x, y: x // y)

    def normalize(self) -> "ConfusionMatrix":
        """Rates instead of counts: every entry divided by tp + fp + fn + tn."""
        total = self.tp + self.fp + self.fn + self.tn
        return ConfusionMatrix(self.tp / total, self.fp / total, self.fn / total, self.tn / total)


# ---------------------------------------------------------------------------
# device-level
# ---------------------------------------------------------------------------

_PRED_SIDE_MIN_ITEMS = 1_000_000      # below this the general kernel's single pass is as fast


def confusion_csr_device(t: D.DeviceCSR, p: D.DeviceCSR, keeps: bool = True) -> torch.Tensor:
    """tp | fp | fn as a (3, m) float64 tensor on the GPU.  `keeps`: `t` is an object the caller holds on to (a
    DeviceCSR of theirs), so what is derived from it alone -- column sums, the row check -- pays off on later calls."""
    out = torch.zeros((3, t.m), dtype=torch.float64, device=t.data.device)
    items = 2 * p.nnz + t.nnz
    # Atomics for the PREDICTED entries only (xc_confusion_csr_pred_side): fn starts from the column sums of y_true --
    # kept with the matrix, so a second prediction scored against the same y_true pays nnz(y_pred) + 2 matches
    # atomics instead of nnz(y_pred) + nnz(y_true) (1 M x 500 K x 50, k = 5: 2.4 -> 0.7 ms; the first call also
    # sums the columns and checks the rows once).  A y_pred row that is not strictly ascending -- the reference's top-k
    # pads a short row with column 0 -- or a y_true that is not falls back to the general kernel.
    # XCOLUMNS_CONFUSION_PRED_SIDE=0 disables.
    if (keeps and os.environ.get("XCOLUMNS_CONFUSION_PRED_SIDE", "1") != "0" and t.n > 0
            and items >= _PRED_SIDE_MIN_ITEMS and t.nnz > 2 * p.nnz and t.rows_ascending()):
        flag = torch.zeros(1, dtype=torch.int32, device=t.data.device)
        # a 0/1 float32 prediction (what predict_* return): no global atomics at all -- match, then a counting sort + LDS
        # sums of (label, y_true value) pairs (xc_scatter_sum_f32): 0.80 -> 0.3 ms at 1 M x 500 K
        if (t.data.dtype == torch.float32 and p.data.dtype == torch.float32 and t.m <= 16384 * 2048
                and os.environ.get("XCOLUMNS_CONFUSION_SCATTER", "1") != "0"):
            val = torch.empty(max(1, p.nnz), dtype=torch.float32, device=t.data.device)
            _lib.call("xc_confusion_csr_match", t.n, D.ptr(t.indptr), D.ptr(t.indices), D.ptr(t.data), D.ptr(p.indptr),
                      D.ptr(p.indices), D.ptr(p.data), D.ptr(val), D.ptr(flag), D.stream())
            nbytes = ctypes.c_int64(0)
            _lib.call("xc_scatter_sum_workspace_bytes", int(p.nnz), t.m, ctypes.byref(nbytes))
            ws = torch.empty(int(nbytes.value), dtype=torch.uint8, device=t.data.device)
            pairs = torch.empty((t.m, 2), dtype=torch.float64, device=t.data.device)
            _lib.call("xc_scatter_sum_f32", int(p.nnz), D.ptr(p.indices), D.ptr(val), t.m, 1, D.ptr(pairs), D.ptr(ws), D.stream())
            out[0].copy_(pairs[:, 0])
            out[1].copy_(pairs[:, 1])
            torch.sub(t.column_sums(), out[0], out=out[2])
            if int(flag.item()) == 0:
                return out
            out.zero_()
            flag.zero_()
        out[2].copy_(t.column_sums())
        _lib.call("xc_confusion_csr_pred_side", t.n, t.m, D.ptr(t.indptr), D.ptr(t.indices), D.ptr(t.data),
                  D.ptr(p.indptr), D.ptr(p.indices), D.ptr(p.data), t.code, D.ptr(out[0]), D.ptr(out[1]), D.ptr(out[2]),
                  D.ptr(flag), D.stream())
        if int(flag.item()) == 0:
            return out
        out.zero_()
    _lib.call("xc_confusion_csr", t.n, t.m, D.ptr(t.indptr), D.ptr(t.indices), D.ptr(t.data),
              D.ptr(p.indptr), D.ptr(p.indices), D.ptr(p.data), t.code, D.ptr(out[0]), D.ptr(out[1]),
              D.ptr(out[2]), D.stream())
    return out


def confusion_dense_device(y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
    n, m = y_true.shape
    out = torch.zeros((3, m), dtype=torch.float64, device=y_true.device)
    _lib.call("xc_confusion_dense", n, m, D.ptr(y_true), D.ptr(y_pred), D.dtype_code(y_true.dtype),
              D.ptr(out[0]), D.ptr(out[1]), D.ptr(out[2]), D.stream())
    return out


def _column_stats(y_true: Matrix, y_pred: Matrix, axis):
    """Validation of confusion_matrix.py:237-268, then the fused kernel.
    Returns (tp, fp, fn) float64 torch tensors on the GPU, and a converter that
    puts a result vector back where the inputs live."""
    on_device = False
    if is_dense(y_true) and is_dense(y_pred):
        dense = True
    elif isinstance(y_true, csr_matrix) and isinstance(y_pred, csr_matrix):
        dense = False
    elif D.is_device_sparse(y_true) and D.is_device_sparse(y_pred):
        # both already resident in HBM (DeviceCSR / torch sparse_csr): no host round trip, the result
        # vectors are torch tensors on the GPU
        dense, on_device = False, True
    else:
        raise ValueError("y_true and y_pred must be both np.ndarray, both torch.Tensor, or csr_matrix")
    if tuple(y_true.shape) != tuple(y_pred.shape):
        raise ValueError("y_true and y_pred must have the same shape")
    if axis not in (0, 1):
        raise ValueError("axis must be 0 or 1")
    dev = D.require_gpu()
    if dense:
        is_torch = isinstance(y_true, torch.Tensor)
        yt = y_true if is_torch else torch.from_numpy(np.ascontiguousarray(y_true))
        if not yt.dtype.is_floating_point or yt.dtype not in (torch.float32, torch.float64):
            yt = yt.to(torch.float64)
        yp = y_pred if isinstance(y_pred, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(y_pred))
        yt = yt.to(dev).contiguous()
        yp = yp.to(device=dev, dtype=yt.dtype).contiguous()
        if axis == 1:
            yt, yp = yt.t().contiguous(), yp.t().contiguous()
        stats = confusion_dense_device(yt, yp)
        home = y_true.device if is_torch else None
    elif on_device:
        t, p = D.as_device_csr(y_true, dev), D.as_device_csr(y_pred, dev)
        if axis == 1:   # the rare direction: transpose with torch's sparse conversion, stays on the GPU
            t = D.DeviceCSR.from_torch(t.to_torch(torch.int64).t().to_sparse_csr(), dev)
            p = D.DeviceCSR.from_torch(p.to_torch(torch.int64).t().to_sparse_csr(), dev)
        if p.data.dtype != t.data.dtype:
            p = D.DeviceCSR(p.indptr, p.indices, p.data.to(t.data.dtype), p.shape, p.max_row_nnz, p.min_row_nnz)
        stats = confusion_csr_device(t, p, keeps=isinstance(y_true, D.DeviceCSR) and axis == 0)
        home = dev
        is_torch = True
    else:
        if axis == 1:
            y_true, y_pred = y_true.T.tocsr(), y_pred.T.tocsr()
            y_true.sort_indices()
            y_pred.sort_indices()
        if y_true.dtype not in (np.float32, np.float64):
            y_true = y_true.astype(np.float64)
        t = D.DeviceCSR.from_scipy(y_true, dev)
        p = D.DeviceCSR.from_scipy(y_pred.astype(y_true.dtype, copy=False), dev)
        stats = confusion_csr_device(t, p, keeps=False)     # uploaded for this call: nothing to keep
        home = None
        is_torch = False

    # dtype=None: CSR results take y_true's dtype (confusion_matrix.py:184, :214, :228); dense ones the dtype
    # of the summed product, i.e. numpy's promotion of the two inputs (:166, :193, :202)
    if on_device:
        default = t.data.dtype
    elif dense and is_torch:
        default = torch.result_type(y_true, y_pred) if isinstance(y_pred, torch.Tensor) else y_true.dtype
    elif dense:
        default = np.result_type(y_true.dtype, y_pred.dtype if hasattr(y_pred, "dtype") else np.float64)
        if default.kind in "bi":
            default = np.dtype(np.int64)
        elif default.kind == "u":
            default = np.dtype(np.uint64)
    else:
        default = D.numpy_dtype(y_true.dtype)

    def back(vec: torch.Tensor, dtype):
        if is_torch:
            v = vec.to(home)
            return v.to(D.torch_dtype(dtype)) if dtype is not None else v.to(default)
        out = vec.cpu().numpy()
        return out.astype(dtype if dtype is not None else default, copy=False)

    return stats, back


def _single(which: int, y_true, y_pred, normalize, axis, dtype):
    stats, back = _column_stats(y_true, y_pred, axis)
    val = back(stats[which], dtype)
    if normalize:  # confusion_matrix.py:265-266
        val = val / y_true.shape[0]
    return val


def calculate_tp(y_true: Matrix, y_pred: Matrix, normalize: bool = False, axis: Optional[int] = 0,
                 dtype: Optional[DType] = None):
    """True positives along `axis` (confusion_matrix.py:271-299)."""
    return _single(0, y_true, y_pred, normalize, axis, dtype)


def calculate_fp(y_true: Matrix, y_pred: Matrix, normalize: bool = False, axis: Optional[int] = 0,
                 dtype: Optional[DType] = None):
    """False positives along `axis` (confusion_matrix.py:302-330)."""
    return _single(1, y_true, y_pred, normalize, axis, dtype)


def calculate_fn(y_true: Matrix, y_pred: Matrix, normalize: bool = False, axis: Optional[int] = 0,
                 dtype: Optional[DType] = None):
    """False negatives along `axis` (confusion_matrix.py:333-361)."""
    return _single(2, y_true, y_pred, normalize, axis, dtype)


def calculate_confusion_matrix(y_true: Matrix, y_pred: Matrix, normalize: bool = False, skip_tn: bool = False,
                               axis: Optional[int] = 0, dtype: Optional[DType] = None) -> ConfusionMatrix:
    """tp, fp, fn (one fused GPU pass) and tn derived from them
    (confusion_matrix.py:364-399): tn = -1 when `skip_tn` (:391-393), else
    ``-tp - fp - fn + (1 if normalize else n or m)`` (:397)."""
    stats, back = _column_stats(y_true, y_pred, axis)
    tp, fp, fn = (back(stats[i], dtype) for i in range(3))
    n, m = tuple(y_true.shape)
    if normalize:
        tp, fp, fn = tp / n, fp / n, fn / n
    if skip_tn:
        tn = tp.copy() if isinstance(tp, np.ndarray) else tp.clone()
        tn[:] = -1
    else:
        tn = -tp - fp - fn + (1.0 if normalize else (n if axis == 0 else m))
    return ConfusionMatrix(tp, fp, fn, tn)

"""Frank-Wolfe search for a randomized weighted classifier on MI355X.

Same functions, arguments, validation, return types and ``meta`` as
/root/reference/xcolumns/frank_wolfe.py: ``find_classifier_using_fw`` (:407-690),
``make_frank_wolfe_wrapper`` and its 16 products (:698-815), the four mixed-utility
wrappers (:818-932), ``RandomizedWeightedClassifier`` (:295-360) and
``predict_using_randomized_weighted_classifier`` (:175-292).

One iteration is exactly data-parallel over rows:

    weighted top-k of every row     xc_topk_csr / xc_topk_dense        (HBM-bound, n rows)
    confusion against y_true        xc_confusion_csr / _dense          (HBM-bound, n rows)
    [all-reduce of 3m counts when rows are sharded over ranks]
    gradient -> next (a, b)         xc_fw_gradient                     (m labels)
    step size along the segment     xc_fw_alpha_curve                  (m x n_alpha)

The reference differentiates the metric with ``autograd``; the kernels evaluate the
same formulas on dual numbers, so `metric_func` must be one of the library's
utilities (macro / micro of the binary metrics, or the mixed ones built here):
arbitrary Python callables cannot run on the GPU and there is no CPU fallback.

dtype flow kept from the reference: the classifier tables are float32
(types.py:13); each classifier's confusion vectors are rounded to y_true's dtype
(confusion_matrix.py:181 ``dtype if dtype else y_true.dtype``, ``/ n`` in that
dtype, :265-266).  The O(m) arithmetic on them runs in float64 here, where the
reference -- for float32 inputs only -- runs its first evaluations in float32.
"""
from __future__ import annotations

import ctypes
import functools
from dataclasses import dataclass, replace
from time import time
from typing import Any, Callable, Dict, Optional, Tuple, Union

import numpy as np
import torch
from scipy.sparse import csr_matrix

from . import _device as D
from . import _lib
from . import metrics as M
from .confusion_matrix import confusion_csr_device, confusion_dense_device
from .metrics import MetricSpec
from .types import DenseMatrix, DType, Matrix, is_dense, is_matrix
from .utils import add_kwargs_to_signature, log_info, log_warning
from .weighted_prediction import threshold_csr_device, topk_csr_device, topk_dense_device

DefaultDataDType = np.float32  # types.py:13


# ---------------------------------------------------------------------------
# the utility: a binary formula + how it is aggregated over labels
# ---------------------------------------------------------------------------

@dataclass(frozen=True)
class FwObjective:
    """What the kernels evaluate for `metric_func(tp, fp, fn, tn)`.

    average "macro": mean over labels (metrics.py:51-58); "micro": the formula on the
    label sums (:83-90); "sum": sum over labels (the mixed utilities, frank_wolfe.py:832-838).
    """
    spec: MetricSpec
    average: str


class FwMetric:
    """A callable utility that carries its device descriptor -- what the mixed wrappers
    pass as `metric_func` (the reference builds a closure there, frank_wolfe.py:832-838)."""

    def __init__(self, objective: FwObjective, name: str):
        self.objective = objective
        self.__name__ = name

    def __call__(self, tp, fp, fn, tn, **kwargs):
        o = self.objective
        spec = replace(o.spec, epsilon=float(kwargs["epsilon"])) if "epsilon" in kwargs else o.spec
        if o.average == "micro":
            return M.host_values(spec, tp.sum(), fp.sum(), fn.sum(), tn.sum())
        v = M.host_values(spec, tp, fp, fn, tn)
        return v.mean() if o.average == "macro" else v.sum()


_STEM_BASE = {
    "precision": _lib.XC_M_PRECISION, "recall": _lib.XC_M_RECALL, "fbeta_score": _lib.XC_M_FBETA,
    "f1_score": _lib.XC_M_FBETA, "jaccard_score": _lib.XC_M_JACCARD,
    "balanced_accuracy": _lib.XC_M_BALANCED_ACC, "gmean": _lib.XC_M_GMEAN, "hmean": _lib.XC_M_HMEAN,
}


def resolve_fw_metric(metric_func, metric_kwargs: Optional[Dict[str, Any]] = None) -> FwObjective:
    """Map `metric_func` (+ `metric_kwargs`) to the descriptor the kernels evaluate.  Known:
    ``{macro,micro}_{precision,recall,fbeta_score,f1_score,jaccard_score,balanced_accuracy,gmean,hmean}
    _on_conf_matrix`` of this package or of the reference (recognised by name), and :class:`FwMetric`."""
    kwargs = dict(metric_kwargs or {})
    func = metric_func
    if isinstance(func, functools.partial):
        if func.args:
            raise NotImplementedError("positional functools.partial arguments are not supported")
        kwargs = {**func.keywords, **kwargs}
        func = func.func
    if isinstance(func, FwObjective):
        obj = func
    elif isinstance(func, FwMetric):
        obj = func.objective
    else:
        name = getattr(func, "__name__", "")
        module = (getattr(func, "__module__", "") or "").split(".")[-1]
        obj = None
        if module == "metrics" and name.endswith("_on_conf_matrix") and name.split("_", 1)[0] in ("macro", "micro"):
            avg, stem = name[: -len("_on_conf_matrix")].split("_", 1)
            if stem in _STEM_BASE:
                obj = FwObjective(MetricSpec(base=_STEM_BASE[stem]), avg)
        if obj is None:
            raise NotImplementedError(
                f"metric_func={name or func!r} is not one of the utilities the MI355X kernels differentiate "
                "(macro_/micro_{precision,recall,fbeta_score,f1_score,jaccard_score,balanced_accuracy,gmean,hmean}"
                "_on_conf_matrix or an xcolumns_amd.frank_wolfe.FwMetric); arbitrary Python callables cannot run "
                "on the GPU and there is no CPU fallback")
    unknown = set(kwargs) - {"epsilon", "beta"}
    if unknown:
        raise NotImplementedError(f"metric_kwargs {sorted(unknown)} are not supported on device")
    spec = obj.spec
    if "epsilon" in kwargs:
        spec = replace(spec, epsilon=float(kwargs["epsilon"]))
    if "beta" in kwargs:
        if spec.base != _lib.XC_M_FBETA:
            raise ValueError("metric_kwargs['beta'] only applies to the F-beta score")
        spec = replace(spec, beta=float(kwargs["beta"]))
    return FwObjective(spec, obj.average)


# ---------------------------------------------------------------------------
# device-level pieces
# ---------------------------------------------------------------------------

def _finish_sum(partials: torch.Tensor) -> float:
    out = ctypes.c_double(0.0)
    _lib.call("xc_utility_finish_host", D.ptr(partials), ctypes.byref(out), None, D.stream())
    return out.value


class FwEngine:
    """Device-resident state of one Frank-Wolfe run over the rows held by THIS rank."""

    def __init__(self, y_true, y_proba, k: int, objective: FwObjective, maximize: bool, normalize: bool,
                 skip_tn: bool, comm=None, n_total: Optional[int] = None):
        self.dev = D.require_gpu()
        self.k = int(k)
        self.obj = objective
        self.metric = objective.spec.to_c()
        self.maximize = bool(maximize)
        self.normalize = bool(normalize)
        self.skip_tn = bool(skip_tn)
        self.comm = comm
        self.n, self.m = (int(x) for x in y_proba.shape)
        self.n_total = int(self.n if n_total is None else n_total)
        self.sparse = isinstance(y_proba, csr_matrix)
        if self.sparse:
            yt = y_true if y_true.dtype in (np.float32, np.float64) else y_true.astype(np.float64)
            self.true = D.DeviceCSR.from_scipy(yt, self.dev)
            self.proba = D.DeviceCSR.from_scipy(y_proba, self.dev)
            self.stat_dtype = self.true.data.dtype
            self.weight_dtype = self.proba.data.dtype  # weighted_prediction.py:72-75
            if self.k > 0:  # every classifier's prediction has the same fixed-stride frame
                self._pred_indptr = (torch.arange(self.n + 1, device=self.dev, dtype=torch.int64) * self.k).to(torch.int32)
                self._pred_ones = torch.ones(self.n * self.k, dtype=self.stat_dtype, device=self.dev)
            # binary labels, k distinct predictions per row, no padded rows: tp and per-label counts are
            # enough (fp = count - tp, fn = label count - tp; xc_confusion_counts_csr)
            row_nnz = np.diff(y_proba.indptr)
            self._counts_path = bool(self.k > 0 and self.n > 0 and row_nnz.min() >= self.k
                                     and bool((self.true.data == 1).all().item()) and yt.has_sorted_indices)
            if self._counts_path:
                # rows per label of y_true (xc_label_busy_list with stride 1 leaves the full histogram in `counts`)
                counts = torch.empty(self.m, dtype=torch.int32, device=self.dev)
                scratch = torch.empty(4, dtype=torch.int32, device=self.dev)
                _lib.call("xc_label_busy_list", self.true.nnz, D.ptr(self.true.indices), 1, self.m, 2 ** 31 - 1, 1,
                          D.ptr(counts), D.ptr(scratch[1:]), D.ptr(scratch[:1]), D.stream())
                self._true_count = counts.to(torch.float64)
        else:
            yt = y_true if isinstance(y_true, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(y_true))
            yp = y_proba if isinstance(y_proba, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(y_proba))
            if yt.dtype not in (torch.float32, torch.float64):
                yt = yt.to(torch.float64)
            D.dtype_code(yp.dtype)
            self.true = yt.to(self.dev).contiguous()
            self.proba = yp.to(self.dev).contiguous()
            self.stat_dtype = self.true.dtype
        self.partials = torch.zeros(_lib.XC_UTILITY_PARTIALS + 1, dtype=torch.float64, device=self.dev)

    # -- one weighted classifier -> its confusion matrix (frank_wolfe.py:560-566, :599-604) --
    def confusion_of(self, a: np.ndarray, b: np.ndarray) -> torch.Tensor:
        """float64 (4, m) on the GPU: tp | fp | fn | tn of predict_weighted_per_instance(y_proba, k,
        th=0.0, a, b) against y_true, normalised / rounded as the reference's calculate_confusion_matrix."""
        if self.sparse:
            # `a`, `b`: the float32 table rows, on the host or already on the GPU (next_classifier)
            a_d = D.to_device(a, dtype=self.weight_dtype, device=self.dev)
            b_d = D.to_device(b, dtype=self.weight_dtype, device=self.dev)
            if self.k > 0:
                # {a, b} interleaved: one gather per candidate (same gains bit for bit)
                ab = torch.stack((a_d, b_d), dim=1).contiguous()
                idx = torch.empty(self.n * self.k, dtype=torch.int32, device=self.dev)
                c = self.proba
                _lib.call("xc_topk_csr_ab", c.n, D.ptr(c.indptr), D.ptr(c.indices), D.ptr(c.data), c.code,
                          int(c.max_row_nnz), self.k, D.ptr(ab), 0, D.ptr(idx), None, None, None, D.stream())
                if self._counts_path:
                    tc = torch.zeros((2, self.m), dtype=torch.float64, device=self.dev)
                    _lib.call("xc_confusion_counts_csr", self.n, self.k, D.ptr(self.true.indptr), D.ptr(self.true.indices),
                              D.ptr(idx), D.ptr(tc[0]), D.ptr(tc[1]), D.stream())
                    counts = torch.stack((tc[0], tc[1] - tc[0], self._true_count - tc[0]))
                    pred = None
                else:
                    pred = D.DeviceCSR(self._pred_indptr, idx, self._pred_ones, (self.n, self.m), self.k)
            else:
                indptr, idx = threshold_csr_device(self.proba, 0.0, a_d, b_d)
                ones = torch.ones(max(1, idx.numel()), dtype=self.stat_dtype, device=self.dev)
                pred = D.DeviceCSR(indptr, idx, ones, (self.n, self.m), 0)
            if pred is not None:
                counts = confusion_csr_device(self.true, pred)
        else:
            gains = self.proba * D.to_device(a, device=self.dev) + D.to_device(b, device=self.dev)  # :37-41
            D.dtype_code(gains.dtype)
            y_pred = topk_dense_device(gains, self.k, 0.0, False, self.stat_dtype)
            counts = confusion_dense_device(self.true, y_pred)
        if self.comm is not None:
            self.comm.all_reduce(counts)
        # the reference accumulates in y_true's dtype and divides in it (counts are exact either way)
        c = counts.to(self.stat_dtype)
        if self.normalize:
            c = c / self.n_total
        stats = torch.empty((4, self.m), dtype=torch.float64, device=self.dev)
        stats[:3] = c
        if self.skip_tn:  # confusion_matrix.py:391-393
            stats[3] = -1.0
        else:  # :397
            stats[3] = (-c[0] - c[1] - c[2] + (1.0 if self.normalize else float(self.n_total))).to(torch.float64)
        return stats

    def _reduced(self, stats: torch.Tensor) -> Tuple[torch.Tensor, int]:
        """micro average: the formula on the label sums, an m = 1 problem"""
        if self.obj.average == "micro":
            return stats.sum(dim=1).contiguous(), 1
        return stats, self.m

    def utility(self, stats: torch.Tensor) -> float:
        s, m = self._reduced(stats)
        _lib.call("xc_utility_vectors", m, 1, D.ptr(s), ctypes.byref(self.metric), D.ptr(self.partials), D.stream())
        total = _finish_sum(self.partials)
        return total / self.m if self.obj.average == "macro" else total

    def next_classifier(self, stats: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """frank_wolfe.py:585-596: (a, b) from the gradient at `stats`, rounded to the classifier
        tables' float32 (types.py:13) and left on the GPU: the next prediction reads them there, the
        host keeps a copy for the returned classifier."""
        s, m = self._reduced(stats)
        ab = torch.empty((2, m), dtype=torch.float64, device=self.dev)
        div = float(self.m) if self.obj.average == "macro" else 1.0
        _lib.call("xc_fw_gradient", m, D.ptr(s), ctypes.byref(self.metric), div, int(not self.maximize),
                  D.ptr(ab[0]), D.ptr(ab[1]), D.stream())
        ab = ab.to(torch.float32)
        if m == 1:
            ab = ab.expand(2, self.m).contiguous()
        return ab[0], ab[1]

    def curve(self, cur: torch.Tensor, nxt: torch.Tensor, alphas: np.ndarray) -> np.ndarray:
        """utility((1 - alpha) * cur + alpha * nxt) up to the constant 1/m, for every alpha."""
        c, m = self._reduced(cur)
        x, _ = self._reduced(nxt)
        n_alpha = int(alphas.size)
        chunks = int(_lib.load().xc_fw_alpha_chunks(m))
        al = torch.from_numpy(np.ascontiguousarray(alphas, dtype=np.float64)).to(self.dev)
        part = torch.empty((chunks, n_alpha), dtype=torch.float64, device=self.dev)
        _lib.call("xc_fw_alpha_curve", m, D.ptr(c), D.ptr(x), ctypes.byref(self.metric), n_alpha, D.ptr(al),
                  D.ptr(part), D.stream())
        # one reduction kernel over the chunk axis (no atomics: the same result on every run)
        return part.sum(dim=0).cpu().numpy()

    def best_alpha(self, cur, nxt, algo: str, eps: float, step: float):
        """_find_best_alpha, frank_wolfe.py:379-404."""
        if algo == "uniform":  # utils.py:174-184: first strict maximum over [0, step, 2 step, ...) < 1
            grid = np.concatenate([[0.0], np.arange(0 + step, 1, step)])
            vals = self.curve(cur, nxt, grid)
            if np.isnan(vals[0]):
                return 0
            best = int(np.argmax(np.where(np.isnan(vals), -np.inf, vals)))
            return 0 if best == 0 else grid[best]
        if algo == "ternary":  # utils.py:187-201
            low, high = 0, 1
            while high - low > eps:
                mid1 = low + (high - low) / 3
                mid2 = high - (high - low) / 3
                v = self.curve(cur, nxt, np.array([mid1, mid2]))
                if v[0] < v[1]:
                    high = mid2
                else:
                    low = mid1
            return (low + high) / 2
        raise ValueError(f"Unknown search algorithm {algo}")


# ---------------------------------------------------------------------------
# randomized classifier (frank_wolfe.py:85-360)
# ---------------------------------------------------------------------------

def draw_classifiers(n: int, classifiers_proba, seed) -> np.ndarray:
    """The classifier of every row: what n sequential ``rng.choice(range(c), p=p)`` calls return
    (frank_wolfe.py:99, :153) -- numpy's Generator.choice draws one uniform double per call and looks it
    up in the normalised cumulative sum, which is done here for all rows at once."""
    p = np.asarray(classifiers_proba.cpu().numpy() if isinstance(classifiers_proba, torch.Tensor) else classifiers_proba)
    rng = np.random.default_rng(seed)
    if p.ndim != 1 or p.size == 0:
        raise ValueError("classifiers_proba must be a non-empty vector")
    atol = max(np.sqrt(np.finfo(np.float64).eps),
               np.sqrt(np.finfo(p.dtype).eps) if np.issubdtype(p.dtype, np.floating) else 0.0)
    pd = p.astype(np.float64)
    if np.isnan(pd).any():
        raise ValueError("probabilities contain NaN")
    if (pd < 0).any():
        raise ValueError("probabilities are not non-negative")
    if abs(pd.sum() - 1.0) > atol:
        raise ValueError("probabilities do not sum to 1")
    cdf = pd.cumsum()
    cdf /= cdf[-1]
    return cdf.searchsorted(rng.random(n), side="right").astype(np.int32)


def _table(x, dtype, dev) -> torch.Tensor:
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
    return t.to(device=dev, dtype=dtype).contiguous()


def _predict_randomized_csr(y_proba: csr_matrix, k, A, B, cls, dtype) -> csr_matrix:
    """frank_wolfe.py:127-172 -> numba_csr_functions.py:550-582, :499-546: per row the ids of the k
    largest gains in ascending order (every id when the row has <= k entries), or gain >= 0 for k == 0."""
    dev = D.require_gpu()
    n, m = y_proba.shape
    a_np = A.cpu().numpy() if isinstance(A, torch.Tensor) else np.asarray(A)
    # gains = data * a[ids] + b[ids] with numpy's promotion (:568-570)
    gdt = np.result_type(y_proba.dtype, a_np.dtype, (B.cpu().numpy() if isinstance(B, torch.Tensor) else np.asarray(B)).dtype)
    D.dtype_code(gdt)
    csr = D.DeviceCSR.from_scipy(y_proba if y_proba.dtype == gdt else y_proba.astype(gdt), dev)
    tdt = D.torch_dtype(gdt)
    a_d, b_d = _table(A, tdt, dev), _table(B, tdt, dev)
    cls_d = torch.from_numpy(cls).to(dev)
    if k > 0:
        idx = torch.empty(n * k, dtype=torch.int32, device=dev)
        _lib.call("xc_topk_csr_rowwise", n, D.ptr(csr.indptr), D.ptr(csr.indices), D.ptr(csr.data), csr.code,
                  int(csr.max_row_nnz), int(k), D.ptr(a_d), D.ptr(b_d), m, D.ptr(cls_d), D.ptr(idx), D.stream())
        lens = torch.clamp(csr.indptr[1:] - csr.indptr[:-1], max=k)
        keep = torch.arange(k, device=dev)[None, :] < lens[:, None]  # a short row keeps its own ids only
        idx_h = idx.view(n, k)[keep].cpu().numpy()
        indptr_h = np.concatenate([[0], np.cumsum(lens.cpu().numpy(), dtype=np.int64)])
    else:
        counts = torch.zeros(n, dtype=torch.int32, device=dev)
        _lib.call("xc_threshold_count_csr_rowwise", n, D.ptr(csr.indptr), D.ptr(csr.indices), D.ptr(csr.data),
                  csr.code, 0.0, D.ptr(a_d), D.ptr(b_d), m, D.ptr(cls_d), D.ptr(counts), D.stream())
        indptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        indptr[1:] = torch.cumsum(counts, 0, dtype=torch.int64).to(torch.int32)
        nnz = int(indptr[-1].item()) if n > 0 else 0
        idx = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
        _lib.call("xc_threshold_fill_csr_rowwise", n, D.ptr(csr.indptr), D.ptr(csr.indices), D.ptr(csr.data),
                  csr.code, 0.0, D.ptr(a_d), D.ptr(b_d), m, D.ptr(cls_d), D.ptr(indptr), D.ptr(idx), D.stream())
        idx_h = idx[:nnz].cpu().numpy()
        indptr_h = indptr.cpu().numpy()
    data = np.ones(idx_h.size, dtype=dtype if dtype else y_proba.data.dtype)  # :145-147
    return csr_matrix((data, idx_h.astype(y_proba.indices.dtype, copy=False),
                       indptr_h.astype(y_proba.indptr.dtype, copy=False)), shape=(n, m))


def _predict_randomized_dense(y_proba, k, A, B, cls, dtype):
    """frank_wolfe.py:85-108 (numpy: gains > 0 for k == 0) and :43-74 (torch: gains >= 0)."""
    dev = D.require_gpu()
    is_torch = isinstance(y_proba, torch.Tensor)
    y = (y_proba if is_torch else torch.from_numpy(np.ascontiguousarray(y_proba))).to(dev)
    D.dtype_code(y.dtype)
    a_d = (A if isinstance(A, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(A))).to(dev)
    b_d = (B if isinstance(B, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(B))).to(dev)
    cls_d = torch.from_numpy(cls).to(dev).long()
    out_dtype = y.dtype if dtype is None else D.torch_dtype(dtype)
    n, m = y.shape
    y_pred = torch.empty((n, m), dtype=out_dtype, device=dev)
    rows = max(1, (1 << 28) // max(1, m * 8))  # bound the gains scratch to ~256 MiB
    for lo in range(0, n, rows):
        hi = min(n, lo + rows)
        gains = y[lo:hi] * a_d[cls_d[lo:hi]] + b_d[cls_d[lo:hi]]
        if k > 0:
            D.dtype_code(gains.dtype)
            y_pred[lo:hi] = topk_dense_device(gains.contiguous(), k, 0.0, False, out_dtype)
        else:
            y_pred[lo:hi] = ((gains >= 0) if is_torch else (gains > 0)).to(out_dtype)
    return y_pred.to(y_proba.device) if is_torch else y_pred.cpu().numpy()


def predict_using_randomized_weighted_classifier(
    y_proba: Matrix,
    k: int,
    classifiers_a: DenseMatrix,
    classifiers_b: DenseMatrix,
    classifiers_proba: DenseMatrix,
    dtype: Optional[DType] = None,
    seed: Optional[int] = None,
) -> Matrix:
    """Prediction of a randomized weighted classifier (frank_wolfe.py:175-292): every row draws one of
    the weighted classifiers (rows of `classifiers_a` / `classifiers_b`) with probabilities
    `classifiers_proba`, then keeps the k labels with the largest ``a_c * eta + b_c`` (k > 0) or the
    labels with positive gain (k == 0).  Same type, shape and dtype as `y_proba`."""
    if not is_matrix(y_proba):
        raise ValueError("y_proba must be either np.ndarray, torch.Tensor, or csr_matrix")
    if len(y_proba.shape) == 1:
        y_proba = y_proba.reshape(1, -1)
    elif len(y_proba.shape) > 2:
        raise ValueError("y_proba must be 1d or 2d")
    if not isinstance(k, int):
        raise ValueError("k must be an integer")
    if not is_dense(classifiers_a) or not is_dense(classifiers_b) or not is_dense(classifiers_proba):
        raise ValueError("classifiers_a, classifiers_b, and classifiers_proba must be ndarray")
    n, m = y_proba.shape
    if classifiers_a.shape[1] != m or classifiers_b.shape[1] != m:
        raise ValueError(
            "classifiers_a, classifier_b, and classifiers_proba must have the same number of columns as y_proba")
    if classifiers_a.shape[0] != classifiers_b.shape[0] or classifiers_a.shape[0] != classifiers_proba.shape[0]:
        raise ValueError("classifiers_a, classifier_b, and classifiers_proba must have the same number of rows")
    cls = draw_classifiers(n, classifiers_proba, seed)
    if isinstance(y_proba, csr_matrix):
        return _predict_randomized_csr(y_proba, k, classifiers_a, classifiers_b, cls, dtype)
    return _predict_randomized_dense(y_proba, k, classifiers_a, classifiers_b, cls, dtype)


class RandomizedWeightedClassifier:
    """A set of weighted classifiers, one of which is drawn for every instance
    (frank_wolfe.py:295-360)."""

    def __init__(self, k: int, a: DenseMatrix, b: DenseMatrix, p: DenseMatrix):
        if not isinstance(k, int):
            raise ValueError("k must be an integer")
        if not is_dense(a) or not is_dense(b) or not is_dense(p):
            raise ValueError("a, b, and p must be ndarray")
        if a.shape != b.shape or a.shape[0] != p.shape[0]:
            raise ValueError(
                "a, b must have the same shape and the number of rows must be equal to the number of rows of p")
        self.k = k
        self.a = a
        self.b = b
        self.p = p

    def predict(self, y_proba: Matrix, dtype: Optional[DType] = None, seed: Optional[int] = None) -> Matrix:
        if y_proba.shape[1] != self.a.shape[1] or y_proba.shape[1] != self.b.shape[1]:
            raise ValueError(
                f"This classifier support the input matrix with {self.a.shape[1]} columns (labels), got {y_proba.shape[1]}")
        return predict_using_randomized_weighted_classifier(y_proba, self.k, self.a, self.b, self.p, dtype=dtype,
                                                            seed=seed)


# ---------------------------------------------------------------------------
# find_classifier_using_fw
# ---------------------------------------------------------------------------

def find_classifier_using_fw(
    y_true: Matrix,
    y_proba: Matrix,
    metric_func: Callable,
    k: int,
    max_iters: int = 100,
    init_classifier: Union[str, Tuple[DenseMatrix, DenseMatrix]] = "top",
    maximize: bool = True,
    normalize_conf_matrix: bool = True,
    metric_kwargs: Optional[Dict[str, Any]] = None,
    tolerance: float = 1e-6,
    search_for_best_alpha: bool = True,
    alpha_search_algo: str = "uniform",
    alpha_tolerance: float = 0.001,
    alpha_uniform_search_step: float = 0.0001,
    skip_tn: bool = False,
    seed: Optional[int] = None,
    verbose: bool = False,
    return_meta: bool = False,
    **kwargs,
) -> Union[RandomizedWeightedClassifier, Tuple[RandomizedWeightedClassifier, Dict[str, Any]]]:
    """Frank-Wolfe search for the randomized weighted classifier maximising (or minimising)
    `metric_func` of the confusion matrix on (`y_true`, `y_proba`); frank_wolfe.py:407-690, same
    arguments, stopping rule and ``meta`` ("alphas", "classifiers_utilities", "utilities", "time",
    "iters").  Extra keyword arguments: ``comm`` / ``n_total`` (rows sharded over ranks: `y_true`,
    `y_proba` are this rank's rows, counts are all-reduced once per iteration; see
    :func:`xcolumns_amd.distributed.find_classifier_using_fw_sharded`)."""
    log_info("Starting searching for optimal randomized classifier using Frank-Wolfe algorithm ...", verbose)
    log_info(f"  Optimization direction: {'maximize' if maximize else 'minimize'}, "
             f"{f'budget k: {k}' if k > 0 else ''}", verbose)
    log_info(f"  Tolerance (stopping condition): {alpha_tolerance}, max iterations: {max_iters}", verbose)

    if type(y_true) != type(y_proba) and is_matrix(y_true):  # :484-487
        raise ValueError(
            f"y_true and y_proba have unsupported combination of types {type(y_true)} and {type(y_proba)}, "
            "should be both np.ndarray, both torch.Tensor, or both csr_matrix")
    if not is_matrix(y_proba):
        raise ValueError("y_proba must be either np.ndarray, torch.Tensor, or csr_matrix")
    if y_true.shape != y_proba.shape:
        raise ValueError(f"y_true and y_proba must have the same shape, got {y_true.shape} and {y_proba.shape}")
    if not isinstance(k, int):
        raise ValueError("k must be an integer")
    objective = resolve_fw_metric(metric_func, metric_kwargs)
    n, m = y_proba.shape
    comm = kwargs.pop("comm", None)
    n_total = kwargs.pop("n_total", None)
    engine_factory = kwargs.pop("engine_factory", FwEngine)

    log_info(f"  Initializing initial {init_classifier if isinstance(init_classifier, str) else 'custom'} classifier ...",
             verbose)
    rng = np.random.default_rng(seed)
    A = np.zeros((max_iters + 1, m), dtype=DefaultDataDType)
    B = np.zeros((max_iters + 1, m), dtype=DefaultDataDType)
    P = np.ones(max_iters + 1, dtype=DefaultDataDType)
    if isinstance(init_classifier, str) and init_classifier == "top":  # :505-507
        A[0] = 1.0
        B[0] = -0.5
    elif isinstance(init_classifier, str) and init_classifier == "random":  # :508-510
        A[0] = rng.random(m)
        B[0] = rng.random(m) - 0.5
    elif isinstance(init_classifier, str) and init_classifier == "prior":  # :511-516
        freq = y_true.sum(dim=0).cpu().numpy() if isinstance(y_true, torch.Tensor) else y_true.sum(axis=0)
        y_freq = np.array(freq, dtype=DefaultDataDType).flatten()
        if comm is not None:
            t = torch.from_numpy(y_freq.astype(np.float64)).to(D.require_gpu())
            comm.all_reduce(t)
            y_freq = t.cpu().numpy().astype(DefaultDataDType)
        A[0] = 1.0 / ((y_freq + 0.1) / (n if n_total is None else n_total))
        B[0] = 0.0
    elif (isinstance(init_classifier, (tuple, list)) and len(init_classifier) == 2
          and is_dense(init_classifier[0]) and is_dense(init_classifier[1])
          and tuple(init_classifier[0].shape) == (m,) and tuple(init_classifier[1].shape) == (m,)):  # :517-534
        A[0] = init_classifier[0].cpu().numpy() if isinstance(init_classifier[0], torch.Tensor) else init_classifier[0]
        B[0] = init_classifier[1].cpu().numpy() if isinstance(init_classifier[1], torch.Tensor) else init_classifier[1]
    else:
        raise ValueError(
            "Unsupported type of init_classifier, it should be in ['random', 'top'], or a tuple of two np.ndarray "
            "or torch.Tensor of shape (y_true.shape[1], )")

    eng = engine_factory(y_true, y_proba, k, objective, maximize, normalize_conf_matrix, skip_tn, comm=comm,
                         n_total=n_total)
    stats = eng.confusion_of(A[0], B[0])
    utility_i = eng.utility(stats)
    meta: Dict[str, Any] = {"alphas": [], "classifiers_utilities": [utility_i], "utilities": [utility_i], "time": time()}
    log_info(f"    Metric value of the first (sub)classifier 0: {utility_i}", verbose)

    new_utility = utility_i
    i = 0
    for i in range(1, max_iters + 1):
        log_info(f"  Starting iteration {i}/{max_iters} ...", verbose)
        old_utility = new_utility  # the utility at `stats` (:585-587)
        a_i, b_i = eng.next_classifier(stats)
        on_device = isinstance(a_i, torch.Tensor)
        if on_device:
            # the host's copy of the float32 rows travels while the GPU predicts, counts and scans
            if a_i.is_cuda:
                if i == 1:
                    stage = torch.empty((2, m), dtype=torch.float32).pin_memory()
                    side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):   # a copy-engine stream of its own: the kernels do not queue behind it
                    stage[0].copy_(a_i, non_blocking=True)
                    stage[1].copy_(b_i, non_blocking=True)
                    staged = torch.cuda.Event()
                    staged.record(side)
            else:
                A[i], B[i] = a_i.numpy(), b_i.numpy()
        else:
            A[i], B[i] = a_i, b_i
        stats_i = eng.confusion_of(a_i if on_device else A[i], b_i if on_device else B[i])
        utility_i = eng.utility(stats_i)
        if on_device and a_i.is_cuda:
            staged.synchronize()
            A[i], B[i] = stage[0].numpy(), stage[1].numpy()   # copied into the tables (float32 rows)
        log_info(f"    Metric value of new (sub)classifier {i}: {utility_i}", verbose)
        if search_for_best_alpha:
            alpha = eng.best_alpha(stats, stats_i, alpha_search_algo, alpha_tolerance, alpha_uniform_search_step)
        else:
            alpha = 2 / (i + 1)
        stats = (1 - alpha) * stats + alpha * stats_i  # :625-628
        new_utility = eng.utility(stats)
        log_info(f"    Iteration {i}/{max_iters} finished, alpha: {alpha}, metric: {old_utility} -> {new_utility}", verbose)
        if alpha < alpha_tolerance or ((maximize and new_utility - old_utility < tolerance)
                                       or (not maximize and old_utility - new_utility < tolerance)):  # :640-660
            log_info(f"  Stopping because alpha is smaller than {alpha_tolerance}" if alpha < alpha_tolerance
                     else f"  Stopping because the improvement is smaller than {tolerance}", verbose)
            A, B, P = A[:i], B[:i], P[:i]
            break
        meta["alphas"].append(alpha)
        meta["classifiers_utilities"].append(utility_i)
        meta["utilities"].append(new_utility)
        P[:i] *= 1 - alpha
        P[i] = alpha
    else:
        log_info("  Stopping because max iterations reached", verbose)
    log_info(f"  Final utility of the randomized classifier: {new_utility}, number of sub-classifiers: {len(A)}", verbose)

    if isinstance(y_proba, torch.Tensor):  # :539-549: the tables follow y_proba's dtype and device
        A, B, P = (torch.tensor(x, dtype=y_proba.dtype, device=y_proba.device) for x in (A, B, P))
    rnd_classifier = RandomizedWeightedClassifier(k, A, B, P)
    if return_meta:
        meta["time"] = time() - meta["time"]
        meta["iters"] = i
        return rnd_classifier, meta
    return rnd_classifier


# ---------------------------------------------------------------------------
# wrappers (frank_wolfe.py:698-932)
# ---------------------------------------------------------------------------

def make_frank_wolfe_wrapper(metric_func: Callable, metric_name: str, maximize: bool = True, skip_tn: bool = False,
                             warn_k_eq_0: bool = False):
    """Factory of ``f(y_true, y_proba, k, **kwargs)`` wrappers around :func:`find_classifier_using_fw`
    for one utility (frank_wolfe.py:698-748)."""

    def find_classifier_for_metric_using_fw(y_true: Matrix, y_proba: Matrix, k: int, **kwargs):
        if warn_k_eq_0 and k == 0:
            log_warning(f"Warning: k=0 results in degenerated solution for {metric_name}!")
        return find_classifier_using_fw(y_true, y_proba, metric_func, k, maximize=maximize, skip_tn=skip_tn, **kwargs)

    find_classifier_for_metric_using_fw.__doc__ = (
        f"Find a randomized classifier that maximizes {metric_name} with the Frank-Wolfe algorithm: "
        f"``find_classifier_using_fw(y_true, y_proba, {metric_func.__name__}, k, ..., maximize={maximize}, "
        f"skip_tn={skip_tn})``.")
    return add_kwargs_to_signature(find_classifier_for_metric_using_fw, find_classifier_using_fw,
                                   skip=["metric_func", "maximize", "skip_tn"])


def _publish_wrappers():
    table = {  # stem: (label, skip_tn, warn_k_eq_0)
        "precision": ("precision", True, True), "recall": ("recall", True, True),
        "f1_score": ("F1 score", True, False), "jaccard_score": ("Jaccard score", True, False),
        "balanced_accuracy": ("balanced accuracy", False, False), "hmean": ("H-mean", False, False),
        "gmean": ("G-mean", False, False),
    }
    g = globals()
    for stem, (label, skip_tn, warn) in table.items():
        for avg in ("macro", "micro"):
            metric = getattr(M, f"{avg}_{stem}_on_conf_matrix")
            g[f"find_classifier_optimizing_{avg}_{stem}_using_fw"] = make_frank_wolfe_wrapper(
                metric, f"{avg}-averaged {label}", maximize=True, skip_tn=skip_tn, warn_k_eq_0=warn)


_publish_wrappers()


def _mixed_with_instance_precision(base: int, name: str):
    def wrapper(y_true: Matrix, y_proba: Matrix, k: int, alpha: float = 1, **kwargs):
        m = y_true.shape[1]
        spec = MetricSpec(base=base, mixed=True, kf=float(k), alpha=float(alpha), mf=float(m))
        return find_classifier_using_fw(y_true, y_proba, FwMetric(FwObjective(spec, "sum"), "mixed_metric_fn"), k,
                                        **kwargs)

    wrapper.__name__ = f"find_classifier_optimizing_mixed_instance_precision_and_macro_{name}_using_fw"
    wrapper.__doc__ = (f"Frank-Wolfe search maximising (1 - alpha) * instance precision@k + alpha * macro-averaged "
                       f"{name.replace('_', ' ')} (frank_wolfe.py:818-903).")
    return wrapper


find_classifier_optimizing_mixed_instance_precision_and_macro_precision_using_fw = _mixed_with_instance_precision(
    _lib.XC_M_PRECISION, "precision")
find_classifier_optimizing_mixed_instance_precision_and_macro_f1_score_using_fw = _mixed_with_instance_precision(
    _lib.XC_M_FBETA, "f1_score")
find_classifier_optimizing_mixed_instance_precision_and_macro_recall_using_fw = _mixed_with_instance_precision(
    _lib.XC_M_RECALL, "recall")


def find_classifier_optimizing_mixed_macro_recall_and_macro_precision_using_fw(
        y_true: Matrix, y_proba: Matrix, k: int, alpha: float = 1, **kwargs):
    """Frank-Wolfe search maximising the label sum of (1 - alpha) * recall + alpha * precision
    (frank_wolfe.py:906-932)."""
    spec = MetricSpec(base=_lib.XC_M_RECALL_PRECISION_MIX, alpha=float(alpha))
    return find_classifier_using_fw(y_true, y_proba, FwMetric(FwObjective(spec, "sum"), "mixed_metric_fn"), k, **kwargs)

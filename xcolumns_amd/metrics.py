"""Binary metrics on confusion-matrix entries and their device descriptors.

The ``binary_*_on_conf_matrix`` functions keep the names, signatures and
formulas of /root/reference/xcolumns/metrics.py (:400-419, :497-513, :585-605,
:633-652, :683-730, :778-797, :824-845, :873-894, :922-944) and work on numbers,
numpy arrays or torch tensors, so user code that evaluates them on the host
keeps working.  Inside the BCA kernels the same formulas run on the GPU: an
arbitrary Python callable cannot, so :func:`resolve_metric` maps a callable to a
:class:`MetricSpec` (``struct xc_metric``) by identity / name and raises for
anything it does not know -- the optimiser never silently optimises a different
metric and there is no CPU fallback.
"""
from __future__ import annotations

import functools
from dataclasses import dataclass, replace
from typing import Any, Callable, Dict, Optional

import numpy as np

import torch

from . import _lib


@dataclass(frozen=True)
class MetricSpec:
    """Host mirror of ``struct xc_metric`` (include/xcolumns_amd.h)."""

    base: int
    mixed: bool = False
    epsilon: float = 1e-9
    beta: float = 1.0
    kf: float = 1.0
    alpha: float = 1.0
    mf: float = 1.0

    def to_c(self) -> _lib.XcMetric:
        return _lib.XcMetric(int(self.base), int(self.mixed), float(self.epsilon), float(self.beta),
                             float(self.kf), float(self.alpha), float(self.mf))

    @property
    def uses_tn(self) -> bool:
        return self.base in (_lib.XC_M_BALANCED_ACC, _lib.XC_M_GMEAN, _lib.XC_M_HMEAN, _lib.XC_M_ACCURACY)


def _tag(base: int):
    def deco(fn):
        fn._xc_base = base
        return fn
    return deco


@_tag(_lib.XC_M_ACCURACY)
def binary_accuracy_on_conf_matrix(tp, fp, fn, tn, normalize: bool = True):
    r"""(TP + TN) / (TP + FP + FN + TN); the plain sum TP + TN when not normalised."""
    acc = tp + tn
    if normalize:
        acc = acc / (tp + fp + fn + tn)
    return acc


@_tag(_lib.XC_M_PRECISION_AT_K)
def binary_precision_at_k_on_conf_matrix(tp, fp, fn, tn, k: int):
    r"""TP / k."""
    return tp / k


@_tag(_lib.XC_M_PRECISION)
def binary_precision_on_conf_matrix(tp, fp, fn, tn, epsilon: float = 1e-9):
    r"""TP / (TP + FP + eps)."""
    return tp / (tp + fp + epsilon)


@_tag(_lib.XC_M_RECALL)
def binary_recall_on_conf_matrix(tp, fp, fn, tn, epsilon: float = 1e-9):
    r"""TP / (TP + FN + eps)."""
    return tp / (tp + fn + epsilon)


@_tag(_lib.XC_M_FBETA)
def binary_fbeta_score_on_conf_matrix(tp, fp, fn, tn, beta: float = 1.0, epsilon: float = 1e-9):
    r"""(1 + beta^2) TP / (beta^2 (TP + FP) + TP + FN + eps)."""
    return (1 + beta**2) * tp / ((beta**2 * (tp + fp)) + tp + fn + epsilon)


@_tag(_lib.XC_M_FBETA)
def binary_f1_score_on_conf_matrix(tp, fp, fn, tn, epsilon: float = 1e-9):
    r"""F-beta with beta = 1."""
    return binary_fbeta_score_on_conf_matrix(tp, fp, fn, tn, beta=1.0, epsilon=epsilon)


@_tag(_lib.XC_M_JACCARD)
def binary_jaccard_score_on_conf_matrix(tp, fp, fn, tn, epsilon: float = 1e-9):
    r"""TP / (TP + FP + FN + eps)."""
    return tp / (tp + fp + fn + epsilon)


@_tag(_lib.XC_M_BALANCED_ACC)
def binary_balanced_accuracy_on_conf_matrix(tp, fp, fn, tn, epsilon: float = 1e-9):
    r"""(TPR + TNR) / 2 with TPR = TP / (TP + FN + eps), TNR = TN / (TN + FP + eps)."""
    tpr = tp / (tp + fn + epsilon)
    tnr = tn / (tn + fp + epsilon)
    return (tpr + tnr) / 2


@_tag(_lib.XC_M_GMEAN)
def binary_gmean_on_conf_matrix(tp, fp, fn, tn, epsilon: float = 1e-9):
    r"""sqrt(TPR * TNR)."""
    tpr = tp / (tp + fn + epsilon)
    tnr = tn / (tn + fp + epsilon)
    return (tpr * tnr) ** 0.5


@_tag(_lib.XC_M_HMEAN)
def binary_hmean_on_conf_matrix(tp, fp, fn, tn, epsilon: float = 1e-9):
    r"""2 TPR TNR / (TPR + TNR)."""
    tpr = tp / (tp + fn + epsilon)
    tnr = tn / (tn + fp + epsilon)
    return (2 * tpr * tnr) / (tpr + tnr)


# the reference's own functions, recognised by name when a caller passes them
_BASE_BY_NAME = {
    "binary_accuracy_on_conf_matrix": _lib.XC_M_ACCURACY,
    "binary_precision_at_k_on_conf_matrix": _lib.XC_M_PRECISION_AT_K,
    "binary_precision_on_conf_matrix": _lib.XC_M_PRECISION,
    "binary_recall_on_conf_matrix": _lib.XC_M_RECALL,
    "binary_fbeta_score_on_conf_matrix": _lib.XC_M_FBETA,
    "binary_f1_score_on_conf_matrix": _lib.XC_M_FBETA,
    "binary_jaccard_score_on_conf_matrix": _lib.XC_M_JACCARD,
    "binary_balanced_accuracy_on_conf_matrix": _lib.XC_M_BALANCED_ACC,
    "binary_gmean_on_conf_matrix": _lib.XC_M_GMEAN,
    "binary_hmean_on_conf_matrix": _lib.XC_M_HMEAN,
}


class DeviceMetric:
    """A callable binary metric that also carries its device descriptor.

    Used for the closures the reference builds inside its wrappers (instance
    precision with a fixed k, block_coordinate.py:821-822; the mixed utilities,
    :862-865 and siblings): calling it evaluates the formula on the host exactly
    like those closures, ``spec`` is what the kernels run.
    """

    def __init__(self, spec: MetricSpec, host_base: Callable, name: str):
        self.spec = spec
        self._host_base = host_base
        self.__name__ = name

    def __call__(self, tp, fp, fn, tn, **kwargs):
        s = self.spec
        if s.base == _lib.XC_M_PRECISION_AT_K:
            base = tp / s.kf
        elif s.base == _lib.XC_M_FBETA:
            base = binary_fbeta_score_on_conf_matrix(tp, fp, fn, tn, beta=s.beta,
                                                     epsilon=kwargs.get("epsilon", s.epsilon))
        elif s.base == _lib.XC_M_ACCURACY:
            base = self._host_base(tp, fp, fn, tn)
        else:
            base = self._host_base(tp, fp, fn, tn, epsilon=kwargs.get("epsilon", s.epsilon))
        if s.mixed:
            return (1 - s.alpha) * (tp / s.kf) + s.alpha * base / s.mf
        return base


def resolve_metric(binary_metric_func, metric_kwargs: Optional[Dict[str, Any]] = None) -> MetricSpec:
    """Map `binary_metric_func` (+ `metric_kwargs`) to the descriptor the kernels
    evaluate.  Raises ``NotImplementedError`` for callables that are not one of
    the known formulas (no silent substitution, no host fallback)."""
    kwargs = dict(metric_kwargs or {})
    func = binary_metric_func
    if isinstance(func, (list, tuple)):
        if len(func) == 0:
            raise ValueError("binary_metric_func list is empty")
        specs = [resolve_metric(f, metric_kwargs) for f in func]
        if any(s != specs[0] for s in specs[1:]):
            raise NotImplementedError(
                "a different binary metric per label is not supported by the MI355X kernels; "
                "pass one metric for all labels")
        return specs[0]
    if isinstance(func, functools.partial):
        if func.args:
            raise NotImplementedError("positional functools.partial arguments are not supported")
        kwargs = {**func.keywords, **kwargs}
        func = func.func
    if isinstance(func, DeviceMetric):
        spec = func.spec
    elif isinstance(func, MetricSpec):
        spec = func
    else:
        base = getattr(func, "_xc_base", None)
        if base is None and callable(func):
            name = getattr(func, "__name__", "")
            module = getattr(func, "__module__", "") or ""
            if name in _BASE_BY_NAME and module.split(".")[-1] == "metrics":
                base = _BASE_BY_NAME[name]
        if base is None:
            raise NotImplementedError(
                f"binary_metric_func={getattr(func, '__name__', func)!r} is not one of the metrics the "
                "MI355X kernels evaluate (binary_{precision,recall,fbeta_score,f1_score,jaccard_score,"
                "balanced_accuracy,gmean,hmean,accuracy,precision_at_k}_on_conf_matrix or a DeviceMetric); "
                "arbitrary Python callables cannot run on the GPU and there is no CPU fallback")
        spec = MetricSpec(base=base)
    allowed = {"epsilon", "beta", "k"}
    unknown = set(kwargs) - allowed
    if unknown:
        raise NotImplementedError(f"metric_kwargs {sorted(unknown)} are not supported on device")
    if "epsilon" in kwargs:
        spec = replace(spec, epsilon=float(kwargs["epsilon"]))
    if "beta" in kwargs:
        if spec.base != _lib.XC_M_FBETA:
            raise ValueError("metric_kwargs['beta'] only applies to the F-beta score")
        spec = replace(spec, beta=float(kwargs["beta"]))
    if "k" in kwargs:
        spec = replace(spec, kf=float(kwargs["k"]))
    return spec


def host_values(spec: MetricSpec, tp, fp, fn, tn):
    """Evaluate `spec` on host arrays (small vectors; used for reporting only)."""
    table = {
        _lib.XC_M_PRECISION_AT_K: lambda: tp / spec.kf,
        _lib.XC_M_PRECISION: lambda: binary_precision_on_conf_matrix(tp, fp, fn, tn, spec.epsilon),
        _lib.XC_M_RECALL: lambda: binary_recall_on_conf_matrix(tp, fp, fn, tn, spec.epsilon),
        _lib.XC_M_FBETA: lambda: binary_fbeta_score_on_conf_matrix(tp, fp, fn, tn, spec.beta, spec.epsilon),
        _lib.XC_M_JACCARD: lambda: binary_jaccard_score_on_conf_matrix(tp, fp, fn, tn, spec.epsilon),
        _lib.XC_M_BALANCED_ACC: lambda: binary_balanced_accuracy_on_conf_matrix(tp, fp, fn, tn, spec.epsilon),
        _lib.XC_M_GMEAN: lambda: binary_gmean_on_conf_matrix(tp, fp, fn, tn, spec.epsilon),
        _lib.XC_M_HMEAN: lambda: binary_hmean_on_conf_matrix(tp, fp, fn, tn, spec.epsilon),
        _lib.XC_M_ACCURACY: lambda: binary_accuracy_on_conf_matrix(tp, fp, fn, tn),
        # frank_wolfe.py:925-929
        _lib.XC_M_RECALL_PRECISION_MIX: lambda: (
            (1 - spec.alpha) * binary_recall_on_conf_matrix(tp, fp, fn, tn, spec.epsilon)
            + spec.alpha * binary_precision_on_conf_matrix(tp, fp, fn, tn, spec.epsilon)),
    }
    base = table[spec.base]()
    if spec.mixed:
        return (1 - spec.alpha) * (tp / spec.kf) + spec.alpha * base / spec.mf
    return base


# ---------------------------------------------------------------------------
# Evaluation on true labels: the step right after prediction in
# every driver of the reference.  The per-label / per-instance statistics come from
# the fused GPU pass (xc_confusion_csr / xc_confusion_dense); the O(m) or O(n)
# arithmetic on them stays on the host, as in /root/reference/xcolumns/metrics.py
# (:38-224 factories, :284-328 label statistics).
# ---------------------------------------------------------------------------

def _with_kwargs_of(func: Callable, source: Callable) -> Callable:
    from .utils import add_kwargs_to_signature

    return add_kwargs_to_signature(func, source)


def make_macro_metric_on_conf_matrix(binary_metric: Callable, name: str) -> Callable:
    """Mean over labels of `binary_metric(tp, fp, fn, tn, **kwargs)`."""

    def macro_metric_on_conf_matrix(tp, fp, fn, tn, **kwargs):
        return binary_metric(tp, fp, fn, tn, **kwargs).mean()

    macro_metric_on_conf_matrix.__doc__ = f"Macro-averaged {name}: the mean of {binary_metric.__name__} over labels."
    return _with_kwargs_of(macro_metric_on_conf_matrix, binary_metric)


def make_micro_metric_on_conf_matrix(binary_metric: Callable, metric_name: str) -> Callable:
    """`binary_metric` of the label-summed entries."""

    def micro_metric_on_conf_matrix(tp, fp, fn, tn, **kwargs):
        return binary_metric(tp.sum(), fp.sum(), fn.sum(), tn.sum(), **kwargs)

    micro_metric_on_conf_matrix.__doc__ = f"Micro-averaged {metric_name}: {binary_metric.__name__} of the summed entries."
    return _with_kwargs_of(micro_metric_on_conf_matrix, binary_metric)


def make_metric_on_y_true_and_y_pred(metric_on_conf_matrix: Callable, metric_name: str,
                                     skip_tn: bool = False) -> Callable:
    """metric(y_true, y_pred, **kwargs) from metric(tp, fp, fn, tn, **kwargs): per-label
    rates (normalised confusion matrix, axis 0) from one GPU pass."""

    def metric_on_y_true_and_y_pred(y_true, y_pred, **kwargs):
        from .confusion_matrix import calculate_confusion_matrix

        C = calculate_confusion_matrix(y_true, y_pred, normalize=True, skip_tn=skip_tn, axis=0)
        return metric_on_conf_matrix(*C, **kwargs)

    metric_on_y_true_and_y_pred.__doc__ = f"{metric_name} of `y_pred` against `y_true` ({metric_on_conf_matrix.__name__})."
    return _with_kwargs_of(metric_on_y_true_and_y_pred, metric_on_conf_matrix)


def make_instance_metric_on_y_true_and_y_pred(binary_metric: Callable, metric_name: str,
                                              skip_tn: bool = False) -> Callable:
    """Instance-averaged metric: `binary_metric` on every row's counts (axis 1), then the mean."""

    def instance_metric_on_y_true_and_y_pred(y_true, y_pred, **kwargs):
        from .confusion_matrix import calculate_confusion_matrix

        C = calculate_confusion_matrix(y_true, y_pred, normalize=False, skip_tn=skip_tn, axis=1)
        return binary_metric(*C, **kwargs).mean()

    instance_metric_on_y_true_and_y_pred.__doc__ = f"Instance-averaged {metric_name} ({binary_metric.__name__} per row)."
    return _with_kwargs_of(instance_metric_on_y_true_and_y_pred, binary_metric)


def _publish_metric_family():
    """macro_* / micro_* / instance_* / binary_* evaluation functions for each formula,
    named as in the reference (metrics.py:608-623 and its siblings)."""
    family = {
        "precision": (binary_precision_on_conf_matrix, True),
        "recall": (binary_recall_on_conf_matrix, True),
        "fbeta_score": (binary_fbeta_score_on_conf_matrix, True),
        "f1_score": (binary_f1_score_on_conf_matrix, True),
        "jaccard_score": (binary_jaccard_score_on_conf_matrix, True),
        "balanced_accuracy": (binary_balanced_accuracy_on_conf_matrix, False),
        "gmean": (binary_gmean_on_conf_matrix, False),
        "hmean": (binary_hmean_on_conf_matrix, False),
    }
    g = globals()
    for stem, (binary, skip_tn) in family.items():
        label = stem.replace("_", " ")
        macro_cm = make_macro_metric_on_conf_matrix(binary, label)
        micro_cm = make_micro_metric_on_conf_matrix(binary, label)
        macro_cm.__name__ = f"macro_{stem}_on_conf_matrix"
        micro_cm.__name__ = f"micro_{stem}_on_conf_matrix"
        g[macro_cm.__name__] = macro_cm
        g[micro_cm.__name__] = micro_cm
        g[f"binary_{stem}"] = make_metric_on_y_true_and_y_pred(binary, f"binary {label}", skip_tn=skip_tn)
        g[f"macro_{stem}"] = make_metric_on_y_true_and_y_pred(macro_cm, f"macro-averaged {label}", skip_tn=skip_tn)
        g[f"micro_{stem}"] = make_metric_on_y_true_and_y_pred(micro_cm, f"micro-averaged {label}", skip_tn=skip_tn)
        g[f"instance_{stem}"] = make_instance_metric_on_y_true_and_y_pred(binary, label, skip_tn=skip_tn)


_publish_metric_family()
# the reference's own spelling of this one (metrics.py:815), kept so call sites keep working
instance_jaccard_score_score = globals()["instance_jaccard_score"]


def label_counts(y):
    """Occurrences of each label (column sums)."""
    if len(y.shape) > 2:
        raise ValueError("y must be a binary matrix")
    return y.sum(axis=0)


def label_priors(y):
    """Prior probability of each label: counts / number of rows."""
    return label_counts(y) / y.shape[0]


def inverse_label_priors(y):
    return 1.0 / label_priors(y)


# ---------------------------------------------------------------------------
# the rest of the reference's evaluation set (metrics.py:17-35, :175-281, :331-397, :422-493,
# :516-583, :972-1003): accuracy / Hamming, precision@k and its propensity-weighted form,
# coverage / abandonment, the tail variants and the Jain et al. propensity model.  All of it is
# O(m) or O(n) arithmetic on the statistics the fused GPU pass returns.
# ---------------------------------------------------------------------------

def check_if_y_pred_at_k(y_pred, k: int) -> bool:
    """True if every row of `y_pred` sums to k > 0 (metrics.py:17-35; its "binary matrix" check
    tests ``(y == 0) & (y == 1)``, which never holds, and is kept as it is)."""
    from scipy.sparse import csr_matrix

    values = y_pred.data if isinstance(y_pred, csr_matrix) else y_pred
    if ((values == 0) & (values == 1)).any():
        raise ValueError("y_pred must be a binary matrix")
    return k > 0 and bool((y_pred.sum(axis=1) == k).all())


def binary_0_1_loss_on_conf_matrix(tp, fp, fn, tn, normalize: bool = True):
    """(fp + fn) / (tp + fp + fn + tn), or fp + fn when not normalised (metrics.py:422-441)."""
    loss = fp + fn
    if normalize:
        loss /= tp + fp + fn + tn
    return loss


def hamming_score_on_conf_matrix(tp, fp, fn, tn, normalize: bool = True):
    """Mean over labels of the binary accuracy (metrics.py:444-458)."""
    return binary_accuracy_on_conf_matrix(tp, fp, fn, tn, normalize=normalize).mean()


def hamming_loss_on_conf_matrix(tp, fp, fn, tn, normalize: bool = True):
    """Mean over labels of the binary 0/1 loss (metrics.py:461-475)."""
    return binary_0_1_loss_on_conf_matrix(tp, fp, fn, tn, normalize=normalize).mean()


def precision_at_k_on_conf_matrix(tp, fp, fn, tn, k: int):
    """Sum over labels of tp / k (metrics.py:516-527)."""
    return binary_precision_at_k_on_conf_matrix(tp, fp, fn, tn, k).sum()


def binary_weighted_precision_at_k_on_conf_matrix(tp, fp, fn, tn, k: int, w):
    """w * tp / k per label (metrics.py:540-557); w: e.g. inverse propensities."""
    return w * tp / k


def weighted_precision_at_k_on_conf_matrix(tp, fp, fn, tn, k: int, w):
    """Sum over labels of w * tp / k (metrics.py:560-572)."""
    return binary_weighted_precision_at_k_on_conf_matrix(tp, fp, fn, tn, k, w).sum()


def coverage_on_conf_matrix(tp, fp, fn, tn):
    """Share of labels (or, per row, of instances) with at least one true positive (metrics.py:972-990)."""
    return (tp > 0).mean()


def make_tail_instance_metric_on_y_true_and_y_pred(binary_metric: Callable, metric_name: str,
                                                   skip_tn: bool = False) -> Callable:
    """Instance-averaged metric restricted to tail labels: labels whose prior is above the given
    percentile are removed from y_true first (metrics.py:175-225)."""

    def tail_instance_metric_on_y_true_and_y_pred(y_true, y_pred, priors, percentile: float = 0.5, **kwargs):
        import numpy as np
        from scipy.sparse import issparse

        from .confusion_matrix import calculate_confusion_matrix

        w = np.array(priors, copy=True)
        p = np.percentile(w, percentile * 100)
        w[w <= p] = 1.0
        w[w > p] = 0.0
        y_tail = y_true.multiply(w).tocsr() if issparse(y_true) else y_true * w
        C = calculate_confusion_matrix(y_tail, y_pred, normalize=False, skip_tn=skip_tn, axis=1)
        return binary_metric(*C, **kwargs).mean()

    tail_instance_metric_on_y_true_and_y_pred.__doc__ = (
        f"Tail instance-averaged {metric_name}: {binary_metric.__name__} per row on the labels at or below the "
        "prior percentile.")
    return _with_kwargs_of(tail_instance_metric_on_y_true_and_y_pred, binary_metric)


def _tail_mask(w, percentile: float):
    import numpy as np

    w = np.array(w, copy=True)
    p = np.percentile(w, percentile * 100)
    w[w < p] = 0.0
    w[w >= p] = 1.0
    return w


def instance_tail_metric(y_true, y_pred, metric_on_conf_matrix_func: Callable, k: int, w, percentile: float = 0.5,
                         epsilon: float = 1e-9):
    """metrics.py:232-252: `metric_on_conf_matrix_func` per row on the labels whose weight is at or above
    the percentile (note the opposite side to the tail_* factories), averaged over rows."""
    from .confusion_matrix import calculate_confusion_matrix

    y_tail = y_true.multiply(_tail_mask(w, percentile)).tocsr()
    C = calculate_confusion_matrix(y_tail, y_pred, normalize=False, skip_tn=True, axis=1)
    return metric_on_conf_matrix_func(*C, epsilon=epsilon).mean()


def instance_tail_recall_at_k(y_true, y_pred, k: int, w, percentile: float = 0.5, epsilon: float = 1e-9):
    """metrics.py:255-281: :func:`instance_tail_metric` with the binary recall."""
    return instance_tail_metric(y_true, y_pred, binary_recall_on_conf_matrix, k, w, percentile=percentile,
                                epsilon=epsilon)


def jpv_inverse_propensities(y, a: float = 0.55, b: float = 1.5):
    """Inverse label propensities of Jain et al. 2016: 1 + C (n_j + b)^-a, C = (log n - 1)(b + 1)^a
    (metrics.py:361-397)."""
    from math import log

    import numpy as np

    n = y.shape[0]
    C = (log(n) - 1) * (b + 1) ** a
    counts = label_counts(y)
    if not isinstance(counts, torch.Tensor):
        counts = np.asarray(counts).ravel()   # scipy gives a 1 x m matrix, on which ** would be a matrix power
    return 1 + C * (counts + b) ** -a


def jpv_propensities(y, a: float = 0.55, b: float = 1.5):
    """1 / :func:`jpv_inverse_propensities` (metrics.py:331-358)."""
    return 1.0 / jpv_inverse_propensities(y, a, b)


binary_accuracy = make_metric_on_y_true_and_y_pred(binary_accuracy_on_conf_matrix, "accuracy")
binary_0_1_loss = make_metric_on_y_true_and_y_pred(binary_0_1_loss_on_conf_matrix, "0/1 loss")
hamming_score = make_metric_on_y_true_and_y_pred(hamming_score_on_conf_matrix, "Hamming score")
hamming_loss = make_metric_on_y_true_and_y_pred(hamming_loss_on_conf_matrix, "Hamming loss")
precision_at_k = make_metric_on_y_true_and_y_pred(precision_at_k_on_conf_matrix, "precision at k", skip_tn=True)
weighted_precision_at_k = make_metric_on_y_true_and_y_pred(weighted_precision_at_k_on_conf_matrix,
                                                           "weighted precision at k", skip_tn=True)
coverage = make_metric_on_y_true_and_y_pred(coverage_on_conf_matrix, "coverage", skip_tn=True)
abandonment = make_instance_metric_on_y_true_and_y_pred(coverage_on_conf_matrix, "abandonment", skip_tn=True)
tail_abandonment = make_tail_instance_metric_on_y_true_and_y_pred(coverage_on_conf_matrix, "tail abandonment", skip_tn=True)
tail_recall = make_tail_instance_metric_on_y_true_and_y_pred(binary_recall_on_conf_matrix, "recall", skip_tn=True)

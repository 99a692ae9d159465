#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the reference itself.

Run once, in the build container (needs /root/reference; the GPU box never
sees it):

    python tests/golden/make_golden.py

The reference (mwydmuch/xCOLUMNs 0.0.3) is pure Python whose only compiled
dependency, numba, is not installable offline.  Every ``@njit`` body on this
path is plain numpy-compatible Python, so the script registers an in-memory
module named ``numba`` whose ``njit``/``jit`` are identity decorators and
``prange`` is ``range``; the reference's own code then runs unmodified and its
outputs are stored here as data (inputs + expected outputs).  No reference
source is copied.

Known differences between this un-JIT'd run and real numba:
tie choice inside np.argpartition (fixtures have no ties at the k-th
boundary -- checked below for the top-k cases), and the random stream of
``numba_random_at_k`` (CPython's ``random`` here).
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
from scipy.sparse import csr_matrix

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("XCOLUMNS_REFERENCE", "/root/reference")


def _install_numba_identity():
    mod = types.ModuleType("numba")

    def _deco(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f

    mod.njit = mod.jit = _deco
    mod.prange = range
    mod.get_num_threads = lambda: 1
    mod.set_num_threads = lambda n: None
    sys.modules["numba"] = mod


_install_numba_identity()
sys.path.insert(0, REFERENCE)

import xcolumns.block_coordinate as ref_bc  # noqa: E402
import xcolumns.metrics as ref_metrics  # noqa: E402
from xcolumns.confusion_matrix import calculate_confusion_matrix  # noqa: E402
from xcolumns.weighted_prediction import predict_weighted_per_instance  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def csr_fields(prefix, mat):
    return {
        prefix + "_data": mat.data,
        prefix + "_indices": mat.indices,
        prefix + "_indptr": mat.indptr,
        prefix + "_shape": np.asarray(mat.shape, dtype=np.int64),
    }


def ragged_csr(rng, n, m, rmax, dtype, zipf=False):
    """Random CSR with 0..rmax sorted distinct columns per row (includes empty
    rows and rows shorter than k)."""
    indptr = [0]
    cols = []
    if zipf:
        w = 1.0 / np.arange(1, m + 1)
        w /= w.sum()
        perm = rng.permutation(m)
    for i in range(n):
        r = int(rng.integers(0, rmax + 1))
        if i < 4:
            r = i  # rows with 0,1,2,3 entries for sure
        if zipf:
            c = perm[rng.choice(m, size=r, replace=False, p=w)]
        else:
            c = rng.choice(m, size=r, replace=False)
        cols.append(np.sort(c))
        indptr.append(indptr[-1] + r)
    indices = np.concatenate(cols).astype(np.int32)
    data = rng.random(indices.size).astype(dtype)
    return csr_matrix((data, indices, np.asarray(indptr, dtype=np.int32)), shape=(n, m))


def fixed_csr(rng, n, m, r, dtype, zipf=False, skew=False):
    """Random CSR with exactly r sorted distinct columns per row."""
    if zipf:
        w = 1.0 / np.arange(1, m + 1)
        w /= w.sum()
        perm = rng.permutation(m)
        cols = np.concatenate([np.sort(perm[rng.choice(m, r, replace=False, p=w)]) for _ in range(n)])
    else:
        cols = np.concatenate([np.sort(rng.choice(m, r, replace=False)) for _ in range(n)])
    data = rng.random(n * r)
    if skew:
        data = data ** 3
    return csr_matrix((data.astype(dtype), cols.astype(np.int32),
                       (np.arange(n + 1) * r).astype(np.int32)), shape=(n, m))


def assert_no_boundary_ties_csr(mat, k, a=None, b=None):
    for i in range(mat.shape[0]):
        s, e = mat.indptr[i], mat.indptr[i + 1]
        g = mat.data[s:e]
        if a is not None:
            g = g * a[mat.indices[s:e]].astype(mat.dtype)
        if b is not None:
            g = g + b[mat.indices[s:e]].astype(mat.dtype)
        if g.size > k:
            gs = np.sort(g)[::-1]
            assert gs[k - 1] != gs[k], f"tie at the top-k boundary in row {i}"


# ---------------------------------------------------------------------------
# A. weighted top-k, CSR
# ---------------------------------------------------------------------------

def gen_topk_csr():
    for dtype, tag in ((np.float32, "f32"), (np.float64, "f64")):
        rng = np.random.default_rng(101 if tag == "f32" else 102)
        n, m, k = 300, 120, 3
        Y = ragged_csr(rng, n, m, 30, dtype)
        a = (rng.random(m) * 2).astype(np.float64)
        b = (rng.random(m) - 0.5).astype(np.float64)
        assert_no_boundary_ties_csr(Y, k)
        assert_no_boundary_ties_csr(Y, k, a, b)
        out = {}
        out.update(csr_fields("y", Y))
        out["a"] = a
        out["b"] = b
        out["k"] = np.int64(k)
        cases = {
            "plain": dict(),
            "scores": dict(keep_scores=True),
            "ab": dict(a=a, b=b),
            "ab_scores": dict(a=a, b=b, keep_scores=True),
            "a_only": dict(a=a),
            "b_only": dict(b=b),
        }
        for name, kw in cases.items():
            P = predict_weighted_per_instance(Y, k, **kw)
            assert P.dtype == dtype
            out.update(csr_fields("pred_" + name, P))
        # k = 0: threshold on the gains
        th = 0.4
        P0 = predict_weighted_per_instance(Y, 0, th=th, a=a, b=b)
        out["th"] = np.float64(th)
        out.update(csr_fields("pred_k0_ab", P0))
        P0 = predict_weighted_per_instance(Y, 0, th=th)
        out.update(csr_fields("pred_k0_plain", P0))
        save("topk_csr_" + tag, **out)


# ---------------------------------------------------------------------------
# B. weighted top-k, dense
# ---------------------------------------------------------------------------

def gen_topk_dense():
    for dtype, tag in ((np.float32, "f32"), (np.float64, "f64")):
        rng = np.random.default_rng(201 if tag == "f32" else 202)
        n, m, k = 200, 60, 3
        Y = rng.random((n, m)).astype(dtype)
        a = (rng.random(m) * 2).astype(np.float64)
        b = (rng.random(m) - 0.5).astype(np.float64)
        a32 = a.astype(np.float32)
        b32 = b.astype(np.float32)
        out = {"y": Y, "a": a, "b": b, "a32": a32, "b32": b32, "k": np.int64(k)}
        out["pred_plain"] = predict_weighted_per_instance(Y, k)
        out["pred_scores"] = predict_weighted_per_instance(Y, k, keep_scores=True)
        out["pred_ab"] = predict_weighted_per_instance(Y, k, a=a, b=b)
        out["pred_ab_scores"] = predict_weighted_per_instance(Y, k, a=a, b=b, keep_scores=True)
        out["pred_ab32"] = predict_weighted_per_instance(Y, k, a=a32, b=b32)
        out["pred_ab32_scores"] = predict_weighted_per_instance(Y, k, a=a32, b=b32, keep_scores=True)
        th = 0.6
        out["th"] = np.float64(th)
        out["pred_k0_ab"] = predict_weighted_per_instance(Y, 0, th=th, a=a, b=b)
        out["pred_k0_plain"] = predict_weighted_per_instance(Y, 0, th=th)
        for key in list(out):
            if key.startswith("pred_"):
                assert out[key].dtype == dtype and out[key].shape == (n, m)
        save("topk_dense_" + tag, **out)


# ---------------------------------------------------------------------------
# C. confusion matrix
# ---------------------------------------------------------------------------

def gen_confusion():
    for dtype, tag in ((np.float32, "f32"), (np.float64, "f64")):
        rng = np.random.default_rng(301 if tag == "f32" else 302)
        n, m, k = 400, 90, 4
        Y = ragged_csr(rng, n, m, 25, dtype)
        P = predict_weighted_per_instance(Y, k)       # includes rows with r < k (padding quirk)
        # a prediction with entries outside the support of Y
        Prand = fixed_csr(rng, n, m, k, dtype)
        Prand.data[:] = 1
        # binary ground truth as float CSR
        L = ragged_csr(rng, n, m, 6, dtype)
        L.data[:] = 1
        out = {}
        out.update(csr_fields("y", Y))
        out.update(csr_fields("p", P))
        out.update(csr_fields("prand", Prand))
        out.update(csr_fields("l", L))
        for tname, T_ in (("y", Y), ("l", L)):
            for pname, P_ in (("p", P), ("prand", Prand)):
                for skip_tn in (False, True):
                    for normalize in (False, True):
                        C = calculate_confusion_matrix(T_, P_, normalize=normalize, skip_tn=skip_tn,
                                                       dtype=np.float64)
                        key = f"C_{tname}_{pname}_skip{int(skip_tn)}_norm{int(normalize)}"
                        out[key] = np.stack([C.tp, C.fp, C.fn, C.tn])
        # dense
        Yd = Y.toarray()
        Pd = (Prand.toarray() > 0).astype(dtype)
        Ld = L.toarray()
        out["yd"], out["pd"], out["ld"] = Yd, Pd, Ld
        for tname, T_ in (("yd", Yd), ("ld", Ld)):
            for skip_tn in (False, True):
                C = calculate_confusion_matrix(T_, Pd, normalize=False, skip_tn=skip_tn, dtype=np.float64)
                out[f"C_{tname}_pd_skip{int(skip_tn)}_norm0"] = np.stack([C.tp, C.fp, C.fn, C.tn])
        save("confusion_" + tag, **out)


# ---------------------------------------------------------------------------
# D. BCA on CSR
# ---------------------------------------------------------------------------

def run_bca(Y, spec):
    """spec: dict(entry=..., k=..., kwargs=...) -> (y_pred, meta)."""
    entry = spec["entry"]
    kw = dict(spec.get("kwargs", {}))
    if "init_matrix" in spec:
        kw["init_y_pred"] = spec["init_matrix"]
    if entry == "generic":
        metric = getattr(ref_metrics, spec["metric"])
        return ref_bc.predict_using_bc_with_0approx(Y, metric, spec["k"], return_meta=True, **kw)
    fn = getattr(ref_bc, entry)
    return fn(Y, spec["k"], return_meta=True, **kw)


def spec_json(spec):
    s = {k: v for k, v in spec.items() if k != "init_matrix"}
    s["has_init_matrix"] = "init_matrix" in spec
    return json.dumps(s, sort_keys=True)


def gen_bca_csr():
    # D1: the SURVEY section 8c anchor
    rng = np.random.default_rng(0)
    n, m, r, k = 2000, 300, 20, 5
    cols = np.concatenate([np.sort(rng.choice(m, r, replace=False)) for _ in range(n)])
    data = rng.random(n * r).astype(np.float32)
    Y = csr_matrix((data, cols.astype(np.int32), (np.arange(n + 1) * r).astype(np.int32)), shape=(n, m))
    spec = dict(entry="predict_optimizing_macro_f1_score_using_bc", k=k,
                kwargs=dict(seed=1, max_iters=5))
    P, meta = run_bca(Y, spec)
    assert abs(meta["utilities"][0] - 0.5711350158138998) < 1e-15, meta["utilities"]
    out = {}
    out.update(csr_fields("y", Y))
    out.update(csr_fields("pred", P))
    out["utilities"] = np.asarray(meta["utilities"], dtype=np.float64)
    out["iters"] = np.int64(meta["iters"])
    out["spec"] = np.asarray(spec_json(spec))
    save("bca_csr_anchor_f32", **out)

    # D2: many metrics / options on smaller inputs
    for dtype, tag in ((np.float32, "f32"), (np.float64, "f64")):
        rng = np.random.default_rng(401 if tag == "f32" else 402)
        n, m, r, k = 600, 80, 12, 3
        Yu = fixed_csr(rng, n, m, r, dtype, skew=True)
        Yz = fixed_csr(rng, n, m, r, dtype, zipf=True, skew=True)
        init = fixed_csr(rng, n, m, k, dtype)   # explicit initial prediction, partly outside the support
        init.data[:] = 1
        specs = {
            "f1": dict(entry="predict_optimizing_macro_f1_score_using_bc", k=k, kwargs=dict(seed=13, max_iters=4)),
            "f1_zipf": dict(entry="predict_optimizing_macro_f1_score_using_bc", k=k, kwargs=dict(seed=13, max_iters=4), data="z"),
            "precision": dict(entry="predict_optimizing_macro_precision_using_bc", k=k, kwargs=dict(seed=2, max_iters=3)),
            "recall": dict(entry="predict_optimizing_macro_recall_using_bc", k=k, kwargs=dict(seed=3, max_iters=3)),
            "jaccard": dict(entry="predict_optimizing_macro_jaccard_score_using_bc", k=k, kwargs=dict(seed=4, max_iters=3)),
            "balacc": dict(entry="predict_optimizing_macro_balanced_accuracy_using_bc", k=k, kwargs=dict(seed=5, max_iters=3)),
            "hmean": dict(entry="predict_optimizing_macro_hmean_using_bc", k=k, kwargs=dict(seed=6, max_iters=3)),
            "gmean": dict(entry="predict_optimizing_macro_gmean_using_bc", k=k, kwargs=dict(seed=7, max_iters=3), data="z"),
            "inst_prec_top": dict(entry="predict_optimizing_instance_precision_using_bc", k=k,
                                  kwargs=dict(seed=8, max_iters=3, init_y_pred="top")),
            "inst_prec_random": dict(entry="predict_optimizing_instance_precision_using_bc", k=k,
                                     kwargs=dict(seed=8, max_iters=3)),
            "mixed_f1": dict(entry="predict_optimizing_mixed_instance_precision_and_macro_f1_score_using_bc", k=k,
                             kwargs=dict(seed=9, max_iters=3, alpha=0.3)),
            "mixed_balacc": dict(entry="predict_optimizing_mixed_instance_precision_and_macro_balanced_accuracy_using_bc",
                                 k=k, kwargs=dict(seed=10, max_iters=3, alpha=0.7)),
            "mixed_recall": dict(entry="predict_optimizing_mixed_instance_precision_and_macro_recall_using_bc", k=k,
                                 kwargs=dict(seed=11, max_iters=3, alpha=0.5), data="z"),
            "fbeta2": dict(entry="generic", metric="binary_fbeta_score_on_conf_matrix", k=k,
                           kwargs=dict(seed=12, max_iters=3, skip_tn=True, metric_kwargs=dict(beta=2.0, epsilon=1e-7))),
            "recall_min": dict(entry="generic", metric="binary_recall_on_conf_matrix", k=k,
                               kwargs=dict(seed=14, max_iters=3, maximize=False, skip_tn=True)),
            "f1_noshuffle": dict(entry="predict_optimizing_macro_f1_score_using_bc", k=k,
                                 kwargs=dict(seed=15, max_iters=3, shuffle_order=False)),
            "f1_greedy": dict(entry="predict_optimizing_macro_f1_score_using_bc", k=k,
                              kwargs=dict(seed=16, max_iters=3, init_y_pred="greedy")),
            "f1_random": dict(entry="predict_optimizing_macro_f1_score_using_bc", k=k,
                              kwargs=dict(seed=17, max_iters=3, init_y_pred="random")),
            "balacc_greedy": dict(entry="predict_optimizing_macro_balanced_accuracy_using_bc", k=k,
                                  kwargs=dict(seed=18, max_iters=2, init_y_pred="greedy")),
            "f1_initmat": dict(entry="predict_optimizing_macro_f1_score_using_bc", k=k,
                               kwargs=dict(seed=19, max_iters=3), init_matrix=True),
            "f1_sum_nonorm_tol": dict(entry="generic", metric="binary_f1_score_on_conf_matrix", k=k,
                                      kwargs=dict(seed=20, max_iters=6, tolerance=1e-4, metric_aggregation="sum",
                                                  skip_tn=True)),
            "accuracy": dict(entry="generic", metric="binary_accuracy_on_conf_matrix", k=k,
                             kwargs=dict(seed=21, max_iters=2)),
        }
        out = {}
        out.update(csr_fields("yu", Yu))
        out.update(csr_fields("yz", Yz))
        out.update(csr_fields("init", init))
        names = []
        for name, spec in specs.items():
            Yin = Yz if spec.get("data") == "z" else Yu
            spec_run = dict(spec)
            if spec.get("init_matrix"):
                spec_run["init_matrix"] = init.copy()
            else:
                spec_run.pop("init_matrix", None)
            P, meta = run_bca(Yin, spec_run)
            assert P.dtype == dtype and (np.diff(P.indptr) == k).all()
            out.update(csr_fields("pred_" + name, P))
            out["utilities_" + name] = np.asarray(meta["utilities"], dtype=np.float64)
            out["iters_" + name] = np.int64(meta["iters"])
            out["spec_" + name] = np.asarray(spec_json(spec))
            names.append(name)
            print(f"  bca_csr_{tag}/{name}: iters={meta['iters']} utilities={meta['utilities']}")
        out["names"] = np.asarray(names)
        save("bca_csr_" + tag, **out)


# ---------------------------------------------------------------------------
# E. BCA on dense
# ---------------------------------------------------------------------------

def gen_bca_dense():
    for dtype, tag in ((np.float32, "f32"), (np.float64, "f64")):
        rng = np.random.default_rng(501 if tag == "f32" else 502)
        n, m, k = 250, 30, 3
        Y = (rng.random((n, m)) ** 3).astype(dtype)
        specs = {
            "f1_top": dict(entry="predict_optimizing_macro_f1_score_using_bc", k=k, kwargs=dict(seed=31, max_iters=3)),
            "f1_random": dict(entry="predict_optimizing_macro_f1_score_using_bc", k=k,
                              kwargs=dict(seed=32, max_iters=3, init_y_pred="random")),
            "f1_greedy": dict(entry="predict_optimizing_macro_f1_score_using_bc", k=k,
                              kwargs=dict(seed=33, max_iters=3, init_y_pred="greedy")),
            "recall": dict(entry="predict_optimizing_macro_recall_using_bc", k=k, kwargs=dict(seed=34, max_iters=3)),
            "balacc": dict(entry="predict_optimizing_macro_balanced_accuracy_using_bc", k=k,
                           kwargs=dict(seed=35, max_iters=3)),
            "gmean_greedy": dict(entry="predict_optimizing_macro_gmean_using_bc", k=k,
                                 kwargs=dict(seed=36, max_iters=2, init_y_pred="greedy")),
            "mixed_hmean": dict(entry="predict_optimizing_mixed_instance_precision_and_macro_hmean_using_bc", k=k,
                                kwargs=dict(seed=37, max_iters=3, alpha=0.4)),
            "f1_k0": dict(entry="predict_optimizing_macro_f1_score_using_bc", k=0,
                          kwargs=dict(seed=38, max_iters=3, init_y_pred="random")),
            "jaccard_k0_top": dict(entry="predict_optimizing_macro_jaccard_score_using_bc", k=0,
                                   kwargs=dict(seed=39, max_iters=3)),
        }
        out = {"y": Y}
        names = []
        for name, spec in specs.items():
            P, meta = run_bca(Y, spec)
            assert P.dtype == dtype and P.shape == (n, m)
            out["pred_" + name] = P
            out["utilities_" + name] = np.asarray(meta["utilities"], dtype=np.float64)
            out["iters_" + name] = np.int64(meta["iters"])
            out["spec_" + name] = np.asarray(spec_json(spec))
            names.append(name)
            print(f"  bca_dense_{tag}/{name}: iters={meta['iters']} utilities={meta['utilities']}")
        out["names"] = np.asarray(names)
        save("bca_dense_" + tag, **out)


# ---------------------------------------------------------------------------
# F. evaluation metrics on true labels (metrics.py:38-224 factories)
# ---------------------------------------------------------------------------

EVAL_NAMES = [f"{avg}_{stem}" for stem in ("precision", "recall", "f1_score", "fbeta_score", "jaccard_score",
                                            "balanced_accuracy", "gmean", "hmean")
              for avg in ("macro", "micro", "instance")]


def gen_eval():
    rng = np.random.default_rng(601)
    n, m, k = 500, 60, 4
    Yp = fixed_csr(rng, n, m, 15, np.float64, zipf=True, skew=True)
    P = predict_weighted_per_instance(Yp, k)
    L = ragged_csr(rng, n, m, 7, np.float64, zipf=True)
    L.data[:] = 1
    out = {}
    out.update(csr_fields("l", L))
    out.update(csr_fields("p", P))
    Ld, Pd = L.toarray(), P.toarray()
    out["ld"], out["pd"] = Ld, Pd
    for name in EVAL_NAMES:
        # the reference spells one of them `instance_jaccard_score_score` (metrics.py:815)
        f = getattr(ref_metrics, name, None) or getattr(ref_metrics, name + "_score")
        out["csr_" + name] = np.float64(f(L, P))
        out["dense_" + name] = np.float64(f(Ld, Pd))
        assert abs(out["csr_" + name] - out["dense_" + name]) < 1e-12, name
    out["names"] = np.asarray(EVAL_NAMES)
    out["label_priors"] = np.asarray(ref_metrics.label_priors(Ld))
    # the rest of the evaluation set (metrics.py:17-35, :175-281, :331-397, :422-583, :972-1003)
    priors = np.asarray(ref_metrics.label_priors(Ld)).ravel()
    inv_ps = np.asarray(ref_metrics.jpv_inverse_propensities(Ld)).ravel()   # (the reference's own fails on CSR: matrix ** -a)
    out["x_priors"], out["x_k"] = priors, np.int64(k)
    out["x_jpv_inverse_propensities"] = inv_ps
    out["x_jpv_propensities"] = np.asarray(ref_metrics.jpv_propensities(Ld, a=0.6, b=2.6)).ravel()
    extra = {
        "binary_accuracy": {}, "binary_0_1_loss": {}, "hamming_score": {}, "hamming_loss": {},
        "precision_at_k": dict(k=k), "weighted_precision_at_k": dict(k=k, w=inv_ps), "coverage": {},
        "abandonment": {}, "tail_abandonment": dict(priors=priors), "tail_recall": dict(priors=priors, percentile=0.3),
    }
    for name, kw in extra.items():
        f = getattr(ref_metrics, name)
        out["xcsr_" + name] = np.asarray(f(L, P, **kw), dtype=np.float64)
        out["xdense_" + name] = np.asarray(f(Ld, Pd, **kw), dtype=np.float64)
    out["xcsr_instance_tail_recall_at_k"] = np.float64(ref_metrics.instance_tail_recall_at_k(L, P, k, priors, percentile=0.4))
    out["xcsr_instance_tail_metric"] = np.float64(ref_metrics.instance_tail_metric(
        L, P, ref_metrics.binary_precision_on_conf_matrix, k, priors, percentile=0.6))
    out["x_check_at_k"] = np.asarray([ref_metrics.check_if_y_pred_at_k(P, k), ref_metrics.check_if_y_pred_at_k(P, k + 1),
                                      ref_metrics.check_if_y_pred_at_k(Pd, k), ref_metrics.check_if_y_pred_at_k(L, k)])
    out["extra_names"] = np.asarray(list(extra))
    save("eval_f64", **out)


# ---------------------------------------------------------------------------
# G. Frank-Wolfe (frank_wolfe.py:407-690) and the randomized classifier (:85-172)
# ---------------------------------------------------------------------------
# frank_wolfe.py imports the `autograd` package (:4-5), absent from this image, for ONE
# call: autograd.grad(metric_func, argnum=[0, 1, 2, 3]) (:368-376).  The stand-in below
# differentiates the reference's own metric functions with torch.autograd on CPU tensors
# of the arrays' own dtype -- what the reference's torch branch does (:18-41) -- and hands
# numpy itself out as `autograd.numpy`.  Everything else that runs is the reference's code.

def _install_autograd_standin():
    import torch

    mod = types.ModuleType("autograd")

    def grad(func, argnum):
        def gradient(*args):
            ts = [torch.tensor(np.asarray(x), requires_grad=True) for x in args]
            value = func(*ts)
            gs = torch.autograd.grad(value, [ts[i] for i in argnum], allow_unused=True, materialize_grads=True)
            return tuple(g.numpy() for g in gs)
        return gradient

    mod.grad = grad
    mod.numpy = np
    sys.modules["autograd"] = mod
    sys.modules["autograd.numpy"] = np


FW_CASES = [
    # name, entry (wrapper name or "generic:<metric>"), layout, dtype, k, kwargs
    ("macro_recall_init", "generic:macro_recall_on_conf_matrix", "csr", "f32", 3, dict(seed=2024, init="ab")),
    ("macro_f1_top", "find_classifier_optimizing_macro_f1_score_using_fw", "csr", "f32", 3, dict(max_iters=10)),
    ("micro_f1", "find_classifier_optimizing_micro_f1_score_using_fw", "csr", "f64", 3,
     dict(max_iters=6, init_classifier="random", seed=11)),
    ("macro_precision_dense", "find_classifier_optimizing_macro_precision_using_fw", "dense", "f32", 3, dict(max_iters=8)),
    ("macro_jaccard_random", "find_classifier_optimizing_macro_jaccard_score_using_fw", "csr", "f32", 4,
     dict(init_classifier="random", seed=7, max_iters=8)),
    ("macro_balanced_accuracy_dense", "find_classifier_optimizing_macro_balanced_accuracy_using_fw", "dense", "f64", 3,
     dict(max_iters=6)),
    ("macro_hmean", "find_classifier_optimizing_macro_hmean_using_fw", "csr", "f32", 3, dict(max_iters=6)),
    ("macro_gmean", "find_classifier_optimizing_macro_gmean_using_fw", "csr", "f64", 3, dict(max_iters=6)),
    ("micro_jaccard_dense", "find_classifier_optimizing_micro_jaccard_score_using_fw", "dense", "f32", 2,
     dict(max_iters=5, init_classifier="random", seed=5)),
    ("mixed_precision_f1", "find_classifier_optimizing_mixed_instance_precision_and_macro_f1_score_using_fw", "csr",
     "f32", 3, dict(alpha=0.5, max_iters=8)),
    ("mixed_recall_precision", "find_classifier_optimizing_mixed_macro_recall_and_macro_precision_using_fw", "csr",
     "f32", 3, dict(alpha=0.3, max_iters=8)),
    ("macro_f1_ternary", "find_classifier_optimizing_macro_f1_score_using_fw", "csr", "f32", 3,
     dict(alpha_search_algo="ternary", max_iters=8, init_classifier="random", seed=3)),
    ("macro_f1_fixed_step", "find_classifier_optimizing_macro_f1_score_using_fw", "csr", "f32", 3,
     dict(search_for_best_alpha=False, max_iters=6, init_classifier="random", seed=4, tolerance=-1.0)),
    ("macro_f1_prior_init", "find_classifier_optimizing_macro_f1_score_using_fw", "csr", "f32", 3,
     dict(init_classifier="prior", max_iters=6)),
    ("macro_f1_k0", "find_classifier_optimizing_macro_f1_score_using_fw", "csr", "f32", 0, dict(max_iters=6)),
    ("macro_recall_unnormalized", "generic:macro_recall_on_conf_matrix", "csr", "f32", 3,
     dict(normalize_conf_matrix=False, max_iters=6, skip_tn=True)),
    ("macro_f1_beta_kwargs", "generic:macro_fbeta_score_on_conf_matrix", "csr", "f32", 3,
     dict(metric_kwargs={"beta": 2.0, "epsilon": 1e-6}, max_iters=6, skip_tn=True)),
]


def gen_fw():
    _install_autograd_standin()
    import xcolumns.frank_wolfe as ref_fw

    rng = np.random.default_rng(701)
    n, m, r = 700, 90, 12
    out = {}
    mats = {}
    for tag, dt in (("f32", np.float32), ("f64", np.float64)):
        Yp = fixed_csr(rng, n, m, r, dt, skew=False)
        Yp.data = (Yp.data.astype(np.float64) ** 2).astype(dt)
        Yt = Yp.copy()
        Yt.data = (rng.random(Yp.nnz) < Yp.data).astype(dt)
        Yt.eliminate_zeros()
        Yt.sort_indices()
        Ytest = fixed_csr(rng, 300, m, r, dt)
        mats[tag] = (Yt, Yp, Ytest)
        out.update(csr_fields(f"true_{tag}", Yt))
        out.update(csr_fields(f"proba_{tag}", Yp))
        out.update(csr_fields(f"test_{tag}", Ytest))
    init_a = rng.random(m)
    init_b = rng.random(m)
    out["init_a"], out["init_b"] = init_a, init_b
    specs = []
    for ci, (name, entry, layout, tag, k, kw) in enumerate(FW_CASES):
        Yt, Yp, Ytest = mats[tag]
        if layout == "dense":
            Yt, Yp, Ytest = Yt.toarray(), Yp.toarray(), Ytest.toarray()
        kw = dict(kw)
        if kw.pop("init", None) == "ab":
            kw["init_classifier"] = (init_a, init_b)
        if entry.startswith("generic:"):
            metric = getattr(ref_metrics, entry.split(":")[1])
            clf, meta = ref_fw.find_classifier_using_fw(Yt, Yp, metric, k, return_meta=True, **kw)
        else:
            clf, meta = getattr(ref_fw, entry)(Yt, Yp, k, return_meta=True, **kw)
        assert np.isfinite(clf.a).all() and np.isfinite(clf.b).all(), name
        pred = clf.predict(Ytest, seed=2024)
        pred_csr = pred if isinstance(pred, csr_matrix) else csr_matrix(pred)
        pred_csr.sort_indices()
        assert clf.a.dtype == np.float32 and clf.p.dtype == np.float32
        out[f"c{ci}_a"], out[f"c{ci}_b"], out[f"c{ci}_p"] = clf.a, clf.b, clf.p
        out[f"c{ci}_alphas"] = np.asarray(meta["alphas"], dtype=np.float64)
        out[f"c{ci}_utilities"] = np.asarray(meta["utilities"], dtype=np.float64)
        out[f"c{ci}_classifiers_utilities"] = np.asarray(meta["classifiers_utilities"], dtype=np.float64)
        out[f"c{ci}_iters"] = np.int64(meta["iters"])
        out[f"c{ci}_pred_indices"] = pred_csr.indices
        out[f"c{ci}_pred_indptr"] = pred_csr.indptr
        out[f"c{ci}_pred_dtype"] = np.asarray(str(pred.dtype))
        kw_json = {k_: v for k_, v in kw.items() if k_ != "init_classifier" or isinstance(v, str)}
        specs.append(json.dumps(dict(name=name, entry=entry, layout=layout, dtype=tag, k=k, kwargs=kw_json,
                                     init_ab="init_classifier" in kw and not isinstance(kw["init_classifier"], str)),
                                sort_keys=True))
        print(f"  fw case {ci} {name}: iters {meta['iters']}, classifiers {clf.a.shape[0]}, "
              f"utility {meta['utilities'][0]:.6f} -> {meta['utilities'][-1]:.6f}, alphas {meta['alphas'][:3]}")
    out["specs"] = np.asarray(specs)
    save("fw", **out)


# ---------------------------------------------------------------------------
# H. coverage BCA (block_coordinate.py:600-701), CSR only: the dense branch calls np.product,
#    which numpy 2 no longer has
# ---------------------------------------------------------------------------

COVERAGE_CASES = [
    ("top_f32", "f32", 3, dict(seed=2024)),
    ("top_f64", "f64", 3, dict(seed=7, max_iters=6)),
    ("mixed_alpha_f64", "f64", 4, dict(seed=3, alpha=0.5, max_iters=6)),
    ("mixed_alpha_f32", "f32", 3, dict(seed=3, alpha=0.7, max_iters=6)),
    ("greedy_f64", "f64", 3, dict(seed=11, init_y_pred="greedy", max_iters=5)),
    ("no_shuffle_f32", "f32", 5, dict(seed=5, shuffle_order=False, max_iters=4, tolerance=-1.0)),
    ("explicit_init_f64", "f64", 3, dict(seed=9, max_iters=5)),
]


def gen_coverage():
    rng = np.random.default_rng(801)
    n, m, r = 900, 1500, 14
    out, specs = {}, []
    mats = {}
    for tag, dt in (("f32", np.float32), ("f64", np.float64)):
        Y = fixed_csr(rng, n, m, r, dt, zipf=True, skew=True)
        mats[tag] = Y
        out.update(csr_fields(f"y_{tag}", Y))
    for ci, (name, tag, k, kw) in enumerate(COVERAGE_CASES):
        Y = mats[tag]
        kw = dict(kw)
        if name.startswith("explicit_init"):
            init = predict_weighted_per_instance(Y, k, a=rng.random(m))
            out[f"c{ci}_init_indices"] = init.indices.copy()
            kw["init_y_pred"] = init
        P, meta = ref_bc.predict_optimizing_coverage_using_bc(Y, k, return_meta=True, **kw)
        assert (np.diff(P.indptr) == k).all()
        out[f"c{ci}_pred_indices"] = P.indices[: n * k].copy()
        out[f"c{ci}_utilities"] = np.asarray(meta["utilities"], dtype=np.float64)
        out[f"c{ci}_iters"] = np.int64(meta["iters"])
        specs.append(json.dumps(dict(name=name, dtype=tag, k=k,
                                     kwargs={a: b for a, b in kw.items() if a != "init_y_pred" or isinstance(b, str)},
                                     explicit_init=name.startswith("explicit_init")), sort_keys=True))
        print(f"  coverage case {ci} {name}: iters {meta['iters']}, utilities {meta['utilities'][:3]} ...")
    out["specs"] = np.asarray(specs)
    save("coverage", **out)


# ---------------------------------------------------------------------------
# I. on-disk inputs of the experiment drivers (experiments/utils.py:113-228)
# ---------------------------------------------------------------------------

def gen_io():
    import importlib.util
    import tempfile

    spec = importlib.util.spec_from_file_location("ref_experiments_utils", os.path.join(REFERENCE, "experiments", "utils.py"))
    ref_io = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_io)
    rng = np.random.default_rng(901)
    n, m = 60, 40
    # XMC-repository style label file: header, "l1,l2 f:v f:v", rows without labels, one unsorted row
    lines = [f"{n} 25 {m}"]
    for i in range(n):
        r = int(rng.integers(0, 5))
        labs = rng.choice(m, r, replace=False)
        if i % 7 != 3:
            labs = np.sort(labs)
        feats = " ".join(f"{int(f)}:{rng.random():.4f}" for f in np.sort(rng.choice(25, 3, replace=False)))
        lines.append(",".join(str(int(x)) for x in labs) + " " + feats)
    labels_txt = "\n".join(lines) + "\n"
    # libsvm-like prediction files: one with sorted rows, one with rows in score order
    def pred_text(sort_rows):
        out = []
        for i in range(n):
            r = int(rng.integers(0, 6))
            cols = rng.choice(m, r, replace=False)
            vals = rng.random(r)
            order = np.argsort(cols) if sort_rows else np.argsort(-vals)
            out.append(" ".join(f"{int(cols[j])}:{vals[j]:.6f}" for j in order))
        return "\n".join(out) + "\n"
    pred_sorted_txt, pred_unsorted_txt = pred_text(True), pred_text(False)
    npy_labels = np.stack([rng.choice(m, 4, replace=False) for _ in range(n)]).astype(np.int64)
    npy_scores = rng.random((n, 4)).astype(np.float32)
    out = {"labels_txt": np.asarray(labels_txt), "pred_sorted_txt": np.asarray(pred_sorted_txt),
           "pred_unsorted_txt": np.asarray(pred_unsorted_txt), "npy_labels": npy_labels, "npy_scores": npy_scores}
    with tempfile.TemporaryDirectory() as d:
        def write(name, text):
            path = os.path.join(d, name)
            with open(path, "w") as f:
                f.write(text)
            return path
        out.update(csr_fields("labels", ref_io.load_txt_labels(write("labels.txt", labels_txt))))
        out.update(csr_fields("pred_sorted", ref_io.load_txt_sparse_pred(write("ps.txt", pred_sorted_txt))))
        out.update(csr_fields("pred_unsorted", ref_io.load_txt_sparse_pred(write("pu.txt", pred_unsorted_txt))))
        base = os.path.join(d, "top")
        np.save(base + "-labels.npy", npy_labels)
        np.save(base + "-scores.npy", npy_scores)
        out.update(csr_fields("npy_pair", ref_io.load_npy_sparse_pred(base)))
    save("io", **out)


def gen_wrappers():
    """The weighted-prediction wrappers the reference's tests call (tests/test_weighted_prediction.py:69-103):
    macro recall / balanced accuracy, log and power-law weights, propensity-scored precision -- CSR float32
    and dense float64 inputs; rows whose k-th and (k+1)-th gains tie are regenerated."""
    import xcolumns.weighted_prediction as wp
    rng = np.random.default_rng(404)
    n, m, r, k = 240, 90, 18, 3
    priors = np.clip(rng.random(m) ** 3, 2e-3, 0.6)
    inv_prop = 1.0 + rng.random(m) * 20.0
    prop = 1.0 / inv_prop
    prop[::17] = 0.0  # zeros are mapped to 1 (weighted_prediction.py:524-526)
    cases = {
        "macro_recall": (wp.predict_optimizing_macro_recall, dict(priors=priors)),
        "macro_balanced_accuracy": (wp.predict_optimizing_macro_balanced_accuracy, dict(priors=priors)),
        "log_weighted": (wp.predict_log_weighted_per_instance, dict(priors=priors)),
        "power_law": (wp.predict_power_law_weighted_per_instance, dict(priors=priors, beta=0.5)),
        "instance_precision": (wp.predict_optimizing_instance_precision, dict()),
        "ps_precision_inverse": (wp.predict_optimizing_instance_propensity_scored_precision,
                                 dict(inverse_propensities=inv_prop)),
        "ps_precision_propensities": (wp.predict_optimizing_instance_propensity_scored_precision,
                                      dict(propensities=prop)),
    }

    def gains_of(name, eta, cols):
        pri = priors[cols]
        if name == "macro_recall":
            return eta * (1.0 / (pri + 1e-6))
        if name == "macro_balanced_accuracy":
            return eta / (pri + 1e-6) - (1 - eta) / (1 - (pri + 1e-6))
        if name == "log_weighted":
            return eta * -np.log(pri + 1e-9)
        if name == "power_law":
            return eta * (pri + 1e-9) ** -0.5
        if name == "instance_precision":
            return eta
        if name == "ps_precision_inverse":
            return eta * inv_prop[cols]
        p = prop[cols].copy()
        p[p == 0] = 1.0
        return eta * (1.0 / p)

    cols_all, data_all = [], []
    for i in range(n):
        while True:
            cols = np.sort(rng.choice(m, r, replace=False))
            eta = rng.random(r)
            ok = True
            for name in cases:
                g = np.sort(gains_of(name, eta.astype(np.float32).astype(np.float64), cols))[::-1]
                if g[k - 1] - g[k] < 1e-4 * max(1.0, abs(g[k - 1])):
                    ok = False
            if ok:
                break
        cols_all.append(cols)
        data_all.append(eta)
    cols_all = np.concatenate(cols_all).astype(np.int32)
    data_all = np.concatenate(data_all)
    indptr = (np.arange(n + 1) * r).astype(np.int32)
    Y32 = csr_matrix((data_all.astype(np.float32), cols_all, indptr), shape=(n, m))
    Yd = Y32.astype(np.float64).toarray()
    out = {"k": np.int64(k), "priors": priors, "inverse_propensities": inv_prop, "propensities": prop}
    out.update(csr_fields("y", Y32))
    for name, (fn, kw) in cases.items():
        kw_c = {kk: (v.copy() if isinstance(v, np.ndarray) else v) for kk, v in kw.items()}
        P = fn(Y32, k, **kw_c)
        assert P.dtype == np.float32 and (np.diff(P.indptr) == k).all()
        out["csr_" + name] = P.indices.reshape(n, k).astype(np.int32)
        kw_d = {kk: (v.copy() if isinstance(v, np.ndarray) else v) for kk, v in kw.items()}
        Pd = fn(Yd, k, **kw_d)
        assert Pd.dtype == np.float64 and (Pd.sum(axis=1) == k).all()
        out["dense_" + name] = np.sort(np.argsort(-Pd, axis=1, kind="stable")[:, :k], axis=1).astype(np.int32)
        assert np.array_equal(out["csr_" + name], out["dense_" + name]) or name == "macro_balanced_accuracy", name
    save("wp_wrappers", **out)


def gen_api():
    """The public surface of the reference's library modules as data: every module-level public name, and for
    plain ``def``s the parameter names with their literal defaults (read with ``ast``; wrappers produced by
    factories are listed by name only)."""
    import ast
    api = {}
    for mod in ("metrics", "block_coordinate", "weighted_prediction", "frank_wolfe", "confusion_matrix", "utils", "types"):
        tree = ast.parse(open(os.path.join(REFERENCE, "xcolumns", mod + ".py")).read())
        entry = {}
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and not node.name.startswith("_"):
                a = node.args
                names = [x.arg for x in a.posonlyargs + a.args]
                defaults = [None] * (len(names) - len(a.defaults)) + list(a.defaults)
                params = []
                for nm, d in zip(names, defaults):
                    item = {"name": nm}
                    if isinstance(d, ast.Constant):
                        item["default"] = d.value
                    params.append(item)
                for x, d in zip(a.kwonlyargs, a.kw_defaults):
                    item = {"name": x.arg}
                    if isinstance(d, ast.Constant):
                        item["default"] = d.value
                    params.append(item)
                entry[node.name] = {"kind": "def", "params": params, "var_kw": a.kwarg is not None}
            elif isinstance(node, ast.ClassDef) and not node.name.startswith("_"):
                entry[node.name] = {"kind": "class", "methods": sorted(
                    n.name for n in node.body if isinstance(n, ast.FunctionDef) and not n.name.startswith("_"))}
            elif isinstance(node, ast.Assign):
                for tg in node.targets:
                    if isinstance(tg, ast.Name) and not tg.id.startswith("_"):
                        entry[tg.id] = {"kind": "name"}
        api[mod] = entry
    with open(os.path.join(HERE, "public_api.json"), "w") as f:
        json.dump(api, f, indent=1, sort_keys=True)
    print("public_api.json:", {k: len(v) for k, v in api.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["topk_csr", "topk_dense", "confusion", "bca_csr", "bca_dense", "eval", "fw", "coverage", "io", "api", "wrappers"]
    for w in which:
        globals()["gen_" + w]()

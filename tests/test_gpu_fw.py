"""GPU parity tests of the Frank-Wolfe path: the HIP path against
the fixtures generated from the reference (tests/golden/fw.npz) and against the CPU oracle
(oracle/fw_ref.py) on seeded inputs.

Bars.  Discrete results -- number of iterations, the step sizes of the uniform search, the drawn
classifiers and every predicted label set -- must be identical.  float64 inputs: utilities within
1e-12 (summation order of the label sums).  float32 inputs: the reference evaluates the first
iteration's metric in float32 (its confusion vectors inherit y_true's dtype), the kernels always in
float64, so utilities agree to float32 resolution (2e-6 relative) and the float32 classifier tables
to a few float32 ulps."""
import numpy as np
import pytest
import torch
from scipy.sparse import csr_matrix

import _golden as G

pytestmark = pytest.mark.gpu


def _product_call(z, spec):
    import xcolumns_amd.frank_wolfe as xfw
    import xcolumns_amd.metrics as xm

    y_true, y_proba, _ = G.fw_inputs(z, spec)
    kw = dict(spec["kwargs"])
    if spec["init_ab"]:
        kw["init_classifier"] = (z["init_a"], z["init_b"])
    entry = spec["entry"]
    if entry.startswith("generic:"):
        return xfw.find_classifier_using_fw(y_true, y_proba, getattr(xm, entry.split(":")[1]), spec["k"],
                                            return_meta=True, **kw)
    return getattr(xfw, entry)(y_true, y_proba, spec["k"], return_meta=True, **kw)


def test_frank_wolfe_golden():
    z = G.load("fw")
    for ci, spec in enumerate(G.fw_cases(z)):
        name = spec["name"]
        clf, meta = _product_call(z, spec)
        f64 = spec["dtype"] == "f64"
        assert meta["iters"] == int(z[f"c{ci}_iters"]), name
        assert clf.a.shape == z[f"c{ci}_a"].shape and clf.a.dtype == np.float32 and clf.p.dtype == np.float32, name
        np.testing.assert_array_equal(np.asarray(meta["alphas"], dtype=np.float64), z[f"c{ci}_alphas"], err_msg=name)
        rtol_u = 1e-12 if f64 else 2e-6
        np.testing.assert_allclose(meta["utilities"], z[f"c{ci}_utilities"], rtol=rtol_u, err_msg=name)
        np.testing.assert_allclose(meta["classifiers_utilities"], z[f"c{ci}_classifiers_utilities"], rtol=rtol_u,
                                   err_msg=name)
        scale = float(np.abs(z[f"c{ci}_a"]).max())
        np.testing.assert_allclose(clf.a, z[f"c{ci}_a"], rtol=5e-6, atol=5e-7 * scale, err_msg=name)
        np.testing.assert_allclose(clf.b, z[f"c{ci}_b"], rtol=5e-6, atol=5e-7 * scale, err_msg=name)
        np.testing.assert_array_equal(clf.p, z[f"c{ci}_p"], err_msg=name)


def test_randomized_classifier_golden():
    """RandomizedWeightedClassifier.predict with the REFERENCE's tables: identical label sets."""
    from xcolumns_amd.frank_wolfe import RandomizedWeightedClassifier

    z = G.load("fw")
    for ci, spec in enumerate(G.fw_cases(z)):
        name = spec["name"]
        _, _, y_test = G.fw_inputs(z, spec)
        clf = RandomizedWeightedClassifier(spec["k"], z[f"c{ci}_a"], z[f"c{ci}_b"], z[f"c{ci}_p"])
        pred = clf.predict(y_test, seed=2024)
        assert type(pred) == type(y_test) and pred.shape == y_test.shape, name
        assert str(pred.dtype) == str(z[f"c{ci}_pred_dtype"]), name
        pc = pred if isinstance(pred, csr_matrix) else csr_matrix(pred)
        pc.sort_indices()
        np.testing.assert_array_equal(pc.indptr, z[f"c{ci}_pred_indptr"], err_msg=name)
        np.testing.assert_array_equal(pc.indices, z[f"c{ci}_pred_indices"], err_msg=name)
        if isinstance(pred, csr_matrix):
            assert pred.indices.dtype == y_test.indices.dtype and pred.indptr.dtype == y_test.indptr.dtype
            assert (pred.data == 1).all()


def _problem(seed, n, m, r, dtype):
    """Scores with skewed label priors (so that weighting labels matters) and labels drawn from them."""
    rng = np.random.default_rng(seed)
    cols = np.concatenate([np.sort(rng.choice(m, r, replace=False)) for _ in range(n)]).astype(np.int32)
    w = 0.03 + 0.97 * rng.random(m) ** 3
    eta = ((rng.random(n * r) ** 2) * w[cols]).astype(dtype)
    indptr = (np.arange(n + 1) * r).astype(np.int32)
    Yp = csr_matrix((eta, cols, indptr), shape=(n, m))
    keep = rng.random(n * r) < eta
    Yt = csr_matrix((keep.astype(dtype), cols.copy(), indptr.copy()), shape=(n, m))  # eliminate_zeros works in place
    Yt.eliminate_zeros()
    return Yt, Yp


@pytest.mark.parametrize("metric,avg,skip_tn", [("f1_score", "macro", True), ("balanced_accuracy", "macro", False),
                                                ("jaccard_score", "micro", True), ("recall", "macro", True)])
def test_frank_wolfe_vs_oracle_f64(metric, avg, skip_tn):
    from oracle import fw_ref as fw

    import xcolumns_amd.frank_wolfe as xfw
    import xcolumns_amd.metrics as xm

    Yt, Yp = _problem(31, 6000, 400, 20, np.float64)
    k = 4
    base = getattr(fw, G.FW_STEMS[metric])
    A, B, P, meta_o = fw.find_classifier_using_fw(Yt, Yp, fw.FwMetric(base=base, average=avg), k, max_iters=6,
                                                 skip_tn=skip_tn, init_classifier="random", seed=5)
    clf, meta = xfw.find_classifier_using_fw(Yt, Yp, getattr(xm, f"{avg}_{metric}_on_conf_matrix"), k, max_iters=6,
                                             skip_tn=skip_tn, init_classifier="random", seed=5, return_meta=True)
    assert meta["iters"] == meta_o["iters"]
    np.testing.assert_array_equal(np.asarray(meta["alphas"], dtype=np.float64), np.asarray(meta_o["alphas"]))
    np.testing.assert_allclose(meta["utilities"], meta_o["utilities"], rtol=1e-12)
    scale = float(np.abs(A).max())
    np.testing.assert_allclose(clf.a, A, rtol=5e-6, atol=5e-7 * scale)
    np.testing.assert_allclose(clf.b, B, rtol=5e-6, atol=5e-7 * scale)
    np.testing.assert_array_equal(clf.p, P)
    # the two predictions from the oracle's tables
    pred = xfw.predict_using_randomized_weighted_classifier(Yp, k, A, B, P, seed=9)
    pred_o = fw.predict_using_randomized_weighted_classifier(Yp, k, A, B, P, seed=9)
    np.testing.assert_array_equal(pred.indptr, pred_o.indptr)
    np.testing.assert_array_equal(pred.indices, pred_o.indices)


def test_frank_wolfe_reference_test_properties():
    """tests/test_frank_wolfe.py:63-103 of the reference: FW for macro recall beats top-k and lands within
    0.02 of the closed-form optimum; predictions keep type, dtype and k labels per row."""
    from xcolumns_amd.confusion_matrix import calculate_confusion_matrix
    from xcolumns_amd.frank_wolfe import find_classifier_using_fw
    from xcolumns_amd.metrics import macro_recall_on_conf_matrix
    from xcolumns_amd.weighted_prediction import predict_optimizing_macro_recall, predict_top_k

    rng = np.random.default_rng(3)
    Yt, Yp = _problem(41, 8000, 300, 25, np.float64)
    y_val, p_val, y_test, p_test = Yt[:5000], Yp[:5000], Yt[5000:], Yp[5000:]
    k = 3
    init_a, init_b = rng.random(300), rng.random(300)
    scores = []
    for conv in (lambda x: x, lambda x: x.toarray()):
        clf, meta = find_classifier_using_fw(conv(y_val), conv(p_val), macro_recall_on_conf_matrix, k,
                                             return_meta=True, seed=2024, init_classifier=(init_a, init_b))
        y_pred = clf.predict(conv(p_test), seed=2024)
        assert type(y_pred) == type(conv(p_test)) and y_pred.dtype == p_test.dtype
        assert (np.asarray(y_pred.sum(axis=1)).ravel() == k).all()
        scores.append(macro_recall_on_conf_matrix(*calculate_confusion_matrix(conv(y_test), y_pred)))
    assert abs(scores[0] - scores[1]) < 1e-12
    top = macro_recall_on_conf_matrix(*calculate_confusion_matrix(y_test, predict_top_k(p_test, k)))
    priors = np.asarray(y_val.mean(axis=0)).ravel()
    opt = macro_recall_on_conf_matrix(*calculate_confusion_matrix(
        y_test, predict_optimizing_macro_recall(p_test, k, priors=priors)))
    assert scores[0] >= top
    assert abs(opt - scores[0]) < 0.02


def test_frank_wolfe_api_contract():
    import xcolumns_amd.frank_wolfe as xfw
    import xcolumns_amd.metrics as xm

    Yt, Yp = _problem(51, 500, 40, 8, np.float32)
    with pytest.raises(ValueError):
        xfw.find_classifier_using_fw(Yt.toarray(), Yp, xm.macro_f1_score_on_conf_matrix, 3)
    with pytest.raises(ValueError):
        xfw.find_classifier_using_fw(Yt[:10], Yp, xm.macro_f1_score_on_conf_matrix, 3)
    with pytest.raises(ValueError):
        xfw.find_classifier_using_fw(Yt, Yp, xm.macro_f1_score_on_conf_matrix, 3, init_classifier="best")
    with pytest.raises(NotImplementedError):
        xfw.find_classifier_using_fw(Yt, Yp, lambda tp, fp, fn, tn: tp.sum(), 3)
    clf = xfw.find_classifier_optimizing_macro_f1_score_using_fw(Yt, Yp, 3, max_iters=3)
    assert isinstance(clf, xfw.RandomizedWeightedClassifier) and clf.a.dtype == np.float32
    with pytest.raises(ValueError):
        clf.predict(Yp[:, :10])
    with pytest.raises(ValueError):
        xfw.RandomizedWeightedClassifier(3, clf.a, clf.b[:, :5], clf.p)
    with pytest.raises(ValueError):
        xfw.predict_using_randomized_weighted_classifier(Yp, 3, clf.a, clf.b, clf.p * 0.5, seed=1)
    # torch tensors in -> torch tables / predictions out (frank_wolfe.py:539-549)
    t_true, t_proba = torch.from_numpy(Yt.toarray()), torch.from_numpy(Yp.toarray())
    clf_t, meta = xfw.find_classifier_optimizing_macro_f1_score_using_fw(t_true, t_proba, 3, max_iters=3, return_meta=True)
    assert isinstance(clf_t.a, torch.Tensor) and clf_t.a.dtype == t_proba.dtype
    clf_d = xfw.find_classifier_optimizing_macro_f1_score_using_fw(Yt.toarray(), Yp.toarray(), 3, max_iters=3)
    np.testing.assert_array_equal(clf_t.a.numpy(), clf_d.a)  # same path as dense numpy input
    pred = clf_t.predict(t_proba, seed=3)
    assert isinstance(pred, torch.Tensor) and pred.dtype == t_proba.dtype and (pred.sum(dim=1) == 3).all()
    # a reference metric function is recognised by name
    import types
    mod = types.ModuleType("xcolumns.metrics")
    def macro_f1_score_on_conf_matrix(tp, fp, fn, tn, epsilon=1e-9):
        raise AssertionError("never evaluated on the host")
    macro_f1_score_on_conf_matrix.__module__ = "xcolumns.metrics"
    clf2 = xfw.find_classifier_using_fw(Yt, Yp, macro_f1_score_on_conf_matrix, 3, max_iters=3, skip_tn=True)
    np.testing.assert_array_equal(clf2.a, clf.a)


def test_frank_wolfe_ragged_rows_vs_oracle():
    """Rows shorter than k (the weighted top-k pads them with column 0, duplicated: SURVEY a-10 i) and
    empty rows go through the same confusion arithmetic as the reference's numba kernels."""
    from oracle import fw_ref as fw

    import xcolumns_amd.frank_wolfe as xfw
    import xcolumns_amd.metrics as xm

    rng = np.random.default_rng(8)
    n, m, k = 4000, 120, 5
    lens = rng.integers(0, 21, size=n)
    lens[:6] = np.arange(6)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cols = np.concatenate([np.sort(rng.choice(m, l, replace=False)) for l in lens]).astype(np.int32)
    w = 0.05 + 0.95 * rng.random(m) ** 2
    eta = ((rng.random(cols.size) ** 2) * w[cols]).astype(np.float64)
    Yp = csr_matrix((eta, cols, indptr), shape=(n, m))
    Yt = csr_matrix(((rng.random(cols.size) < eta).astype(np.float64), cols.copy(), indptr.copy()), shape=(n, m))
    Yt.eliminate_zeros()
    for avg, metric in (("macro", "f1_score"), ("micro", "precision")):
        A, B, P, meta_o = fw.find_classifier_using_fw(Yt, Yp, fw.FwMetric(base=getattr(fw, G.FW_STEMS[metric]), average=avg),
                                                     k, max_iters=5, skip_tn=True, alpha_uniform_search_step=0.001)
        clf, meta = xfw.find_classifier_using_fw(Yt, Yp, getattr(xm, f"{avg}_{metric}_on_conf_matrix"), k, max_iters=5,
                                                 skip_tn=True, alpha_uniform_search_step=0.001, return_meta=True)
        assert meta["iters"] == meta_o["iters"]
        np.testing.assert_array_equal(np.asarray(meta["alphas"], dtype=np.float64), np.asarray(meta_o["alphas"]))
        np.testing.assert_allclose(meta["utilities"], meta_o["utilities"], rtol=1e-12)
        np.testing.assert_array_equal(clf.p, P)
        pred = clf.predict(Yp, seed=4)
        pred_o = fw.predict_using_randomized_weighted_classifier(Yp, k, clf.a, clf.b, clf.p, seed=4)
        np.testing.assert_array_equal(pred.indptr, pred_o.indptr)
        np.testing.assert_array_equal(pred.indices, pred_o.indices)
        assert (np.diff(pred.indptr) == np.minimum(lens, k)).all()


@pytest.mark.parametrize("base", ["PRECISION", "RECALL", "FBETA", "JACCARD"])
@pytest.mark.parametrize("mixed", [False, True])
def test_step_scan_fast_path_matches_the_exact_expression(base, mixed):
    """The long step scan (> 16 points: per-label linear-fractional form, reciprocal seeded from the
    neighbouring step size where the denominator moves slowly) against the exact kernel (the reference's
    expression with IEEE division, used for <= 16 points) at the same step sizes -- on labels that stress the
    seed: denominators that start or end at epsilon, labels untouched by one of the two classifiers, identical
    statistics, and a non-uniform grid."""
    import ctypes
    from xcolumns_amd import _device as D, _lib
    from xcolumns_amd.metrics import MetricSpec
    dev = D.require_gpu()
    rng = np.random.default_rng(3)
    m, n = 5000, 200000.0

    def stats():
        pred = rng.integers(0, 400, m).astype(np.float64)
        pos = rng.integers(0, 300, m).astype(np.float64)
        pred[rng.random(m) < 0.2] = 0            # never predicted: precision's denominator is epsilon
        pos[rng.random(m) < 0.1] = 0             # no positives
        tp = np.floor(np.minimum(pred, pos) * rng.random(m))
        fp, fn = pred - tp, pos - tp
        return np.stack([tp, fp, fn, n - tp - fp - fn]) / n

    cur, nxt = stats(), stats()
    nxt[:, :50] = cur[:, :50]                    # labels the step does not move
    spec = MetricSpec(base=getattr(_lib, "XC_M_" + base), mixed=mixed, alpha=0.7, kf=3.0, mf=float(m)) if mixed \
        else MetricSpec(base=getattr(_lib, "XC_M_" + base))
    metric = spec.to_c()
    grid = np.concatenate([[0.0], np.arange(1e-4, 1, 1e-4)])
    grid[5000:] = np.sort(rng.random(grid.size - 5000))   # second half: irregular spacing
    c_d, x_d = (torch.from_numpy(v).to(dev).contiguous() for v in (cur, nxt))

    def curve(alphas):
        chunks = int(_lib.load().xc_fw_alpha_chunks(m))
        al = torch.from_numpy(np.ascontiguousarray(alphas)).to(dev)
        part = torch.empty((chunks, alphas.size), dtype=torch.float64, device=dev)
        _lib.call("xc_fw_alpha_curve", m, D.ptr(c_d), D.ptr(x_d), ctypes.byref(metric), int(alphas.size), D.ptr(al),
                  D.ptr(part), D.stream())
        return part.sum(dim=0).cpu().numpy()

    fast = curve(grid)
    picks = np.concatenate([np.arange(0, 64), rng.choice(grid.size, 192, replace=False), [grid.size - 1]])
    exact = np.concatenate([curve(grid[picks[i:i + 16]]) for i in range(0, picks.size, 16)])
    rel = np.abs(fast[picks] - exact) / np.abs(exact)
    print(base, mixed, "max relative difference", rel.max())
    assert rel.max() < 1e-12

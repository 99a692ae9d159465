"""Pins the CPU oracle (oracle/) against fixtures generated from the reference
itself (tests/golden/make_golden.py).  Bit-exact for index sets and for values
the reference computes without reductions; utilities / confusion vectors must be
bit-identical too because the oracle repeats the reference's operation order."""
import numpy as np
import pytest
from scipy.sparse import csr_matrix

import _golden as G


def _same_csr(a, b, data_exact=True):
    assert a.shape == b.shape
    assert np.array_equal(np.asarray(a.indptr, dtype=np.int64), np.asarray(b.indptr, dtype=np.int64))
    nnz = int(b.indptr[-1])
    assert np.array_equal(a.indices[:nnz], b.indices[:nnz])
    if data_exact:
        assert np.array_equal(a.data[:nnz], b.data[:nnz])


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_topk_csr_golden(oref, tag):
    z = G.load("topk_csr_" + tag)
    Y = G.csr_from(z, "y")
    k = int(z["k"])
    a, b = z["a"], z["b"]
    cases = {
        "plain": dict(), "scores": dict(keep_scores=True), "ab": dict(a=a, b=b),
        "ab_scores": dict(a=a, b=b, keep_scores=True), "a_only": dict(a=a), "b_only": dict(b=b),
    }
    for name, kw in cases.items():
        P = oref.predict_weighted_per_instance(Y, k, **kw)
        assert P.dtype == Y.dtype
        _same_csr(P, G.csr_from(z, "pred_" + name))
    th = float(z["th"])
    _same_csr(oref.predict_weighted_per_instance(Y, 0, th=th, a=a, b=b), G.csr_from(z, "pred_k0_ab"))
    _same_csr(oref.predict_weighted_per_instance(Y, 0, th=th), G.csr_from(z, "pred_k0_plain"))


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_topk_dense_golden(oref, tag):
    z = G.load("topk_dense_" + tag)
    Y, k = z["y"], int(z["k"])
    a, b, a32, b32 = z["a"], z["b"], z["a32"], z["b32"]
    th = float(z["th"])
    cases = {
        "plain": dict(), "scores": dict(keep_scores=True), "ab": dict(a=a, b=b),
        "ab_scores": dict(a=a, b=b, keep_scores=True), "ab32": dict(a=a32, b=b32),
        "ab32_scores": dict(a=a32, b=b32, keep_scores=True),
    }
    for name, kw in cases.items():
        P = oref.predict_weighted_per_instance(Y, k, **kw)
        assert P.dtype == Y.dtype
        assert np.array_equal(P, z["pred_" + name]), name
    assert np.array_equal(oref.predict_weighted_per_instance(Y, 0, th=th, a=a, b=b), z["pred_k0_ab"])
    assert np.array_equal(oref.predict_weighted_per_instance(Y, 0, th=th), z["pred_k0_plain"])


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_confusion_golden(oref, tag):
    z = G.load("confusion_" + tag)
    mats = {n: G.csr_from(z, n) for n in ("y", "p", "prand", "l")}
    for tname in ("y", "l"):
        for pname in ("p", "prand"):
            for skip_tn in (False, True):
                for normalize in (False, True):
                    C = oref.calculate_confusion_matrix(mats[tname], mats[pname], normalize=normalize,
                                                        skip_tn=skip_tn)
                    exp = z[f"C_{tname}_{pname}_skip{int(skip_tn)}_norm{int(normalize)}"]
                    assert np.array_equal(np.stack(C), exp), (tname, pname, skip_tn, normalize)
    for tname in ("yd", "ld"):
        for skip_tn in (False, True):
            C = oref.calculate_confusion_matrix(z[tname], z["pd"], skip_tn=skip_tn)
            assert np.array_equal(np.stack(C), z[f"C_{tname}_pd_skip{int(skip_tn)}_norm0"])


def test_bca_csr_anchor_golden(oref):
    z = G.load("bca_csr_anchor_f32")
    Y = G.csr_from(z, "y")
    spec = G.spec_of(z)
    P, meta = G.oracle_call_from_spec(oref, spec, Y)
    assert meta["iters"] == int(z["iters"])
    assert np.array_equal(np.asarray(meta["utilities"]), z["utilities"])
    # the SURVEY section 8c anchor values, re-derived when the fixture was generated
    assert meta["utilities"][0] == 0.5711350158138998
    assert meta["utilities"][-1] == 0.5716832858107608
    _same_csr(P, G.csr_from(z, "pred"))


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_bca_csr_golden(oref, tag):
    z = G.load("bca_csr_" + tag)
    Yu, Yz, init = G.csr_from(z, "yu"), G.csr_from(z, "yz"), G.csr_from(z, "init")
    for name in [str(s) for s in z["names"]]:
        spec = G.spec_of(z, name)
        Y = Yz if spec.get("data") == "z" else Yu
        P, meta = G.oracle_call_from_spec(oref, spec, Y, init_matrix=init.copy())
        assert meta["iters"] == int(z["iters_" + name]), name
        assert np.array_equal(np.asarray(meta["utilities"]), z["utilities_" + name]), (
            name, meta["utilities"], z["utilities_" + name])
        _same_csr(P, G.csr_from(z, "pred_" + name))


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_bca_dense_golden(oref, tag):
    z = G.load("bca_dense_" + tag)
    Y = z["y"]
    for name in [str(s) for s in z["names"]]:
        spec = G.spec_of(z, name)
        P, meta = G.oracle_call_from_spec(oref, spec, Y)
        assert meta["iters"] == int(z["iters_" + name]), name
        assert np.array_equal(np.asarray(meta["utilities"]), z["utilities_" + name]), (
            name, meta["utilities"], z["utilities_" + name])
        assert P.dtype == Y.dtype
        assert np.array_equal(P, z["pred_" + name]), name


def test_frank_wolfe_golden():
    """oracle/fw_ref.py against the reference's find_classifier_using_fw + RandomizedWeightedClassifier.predict.
    The reference differentiates with autograd, the oracle with closed forms: the float32 classifier
    tables may differ in the last bit, everything discrete (iterations, alphas, predictions) must match."""
    from oracle import fw_ref as fw

    z = G.load("fw")
    for ci, spec in enumerate(G.fw_cases(z)):
        A, B, P, meta = G.fw_oracle_call(fw, z, spec)
        name = spec["name"]
        assert meta["iters"] == int(z[f"c{ci}_iters"]), name
        assert A.shape == z[f"c{ci}_a"].shape and A.dtype == np.float32, name
        # a few float32 ulps: for float32 inputs the first gradient is float32 arithmetic on both sides
        scale = float(np.abs(z[f"c{ci}_a"]).max())
        np.testing.assert_allclose(A, z[f"c{ci}_a"], rtol=5e-6, atol=5e-7 * scale, err_msg=name)
        np.testing.assert_allclose(B, z[f"c{ci}_b"], rtol=5e-6, atol=5e-7 * scale, err_msg=name)
        np.testing.assert_array_equal(P, z[f"c{ci}_p"], err_msg=name)
        np.testing.assert_array_equal(np.asarray(meta["alphas"]), z[f"c{ci}_alphas"], err_msg=name)
        np.testing.assert_allclose(meta["utilities"], z[f"c{ci}_utilities"], rtol=1e-13, err_msg=name)
        np.testing.assert_allclose(meta["classifiers_utilities"], z[f"c{ci}_classifiers_utilities"], rtol=1e-13,
                                   err_msg=name)
        # the randomized classifier, with the REFERENCE's tables so the check is independent of the above
        _, _, y_test = G.fw_inputs(z, spec)
        pred = fw.predict_using_randomized_weighted_classifier(y_test, spec["k"], z[f"c{ci}_a"], z[f"c{ci}_b"],
                                                               z[f"c{ci}_p"], seed=2024)
        assert str(pred.dtype) == str(z[f"c{ci}_pred_dtype"]), name
        pc = pred if isinstance(pred, csr_matrix) else csr_matrix(pred)
        pc.sort_indices()
        np.testing.assert_array_equal(pc.indptr, z[f"c{ci}_pred_indptr"], err_msg=name)
        np.testing.assert_array_equal(pc.indices, z[f"c{ci}_pred_indices"], err_msg=name)


def test_coverage_bca_golden():
    """oracle/coverage_ref.py against the reference's predict_optimizing_coverage_using_bc (CSR): identical
    predictions and iteration counts; utilities to 1e-12 (float64) -- for float32 inputs the reference sums
    the precision@k part of a mixed utility in float32, the oracle in float64 then rounds: 1e-6."""
    from oracle import coverage_ref as cov

    z = G.load("coverage")
    for ci, spec in enumerate(G.coverage_cases(z)):
        Y, kw = G.coverage_inputs(z, ci, spec)
        P, meta = cov.predict_optimizing_coverage_using_bc(Y, spec["k"], **kw)
        name = spec["name"]
        assert meta["iters"] == int(z[f"c{ci}_iters"]), name
        tol = 1e-6 if (spec["dtype"] == "f32" and kw.get("alpha", 1) < 1) else 1e-12
        np.testing.assert_allclose(meta["utilities"], z[f"c{ci}_utilities"], rtol=0, atol=tol, err_msg=name)
        np.testing.assert_array_equal(P.indices, z[f"c{ci}_pred_indices"], err_msg=name)
        assert P.dtype == Y.dtype and (np.diff(P.indptr) == spec["k"]).all()


def test_threaded_baseline_legs_equal_the_serial_oracle(oref):
    """The OpenMP row-parallel top-k / confusion passes (bench.py's T = 8 / all-core CPU baseline legs,
    the reference's prange loops) give the serial oracle's results: index sets bit-exact, column sums
    to rounding."""
    from xcolumns_amd.synthetic import make_csr

    for dtype in (np.float32, np.float64):
        Y = make_csr(3000, 400, 30, seed=5, k=4, dtype=dtype)
        P = oref.predict_top_k(Y, 4)
        for t in (1, 3, 8):
            Pt = oref.predict_top_k_threads(Y, 4, t)
            assert np.array_equal(P.indices, Pt.indices) and np.array_equal(P.indptr, Pt.indptr)
            tp, fp, fn, _ = oref.calculate_confusion_matrix(Y, P, skip_tn=True)
            tpt, fpt, fnt = oref.calculate_confusion_matrix_threads(Y, P, t)
            np.testing.assert_allclose(tpt, tp, rtol=0, atol=1e-9)
            np.testing.assert_allclose(fpt, fp, rtol=0, atol=1e-9)
            np.testing.assert_allclose(fnt, fn, rtol=0, atol=1e-9)

"""The ORDERED parallel sweep (csrc/xc_bca_ord.hip) against the sequential oracle: thousands of rows in flight, the
reference's visiting-order semantics (/root/reference/xcolumns/block_coordinate.py:448-463) -- identical
predictions, utilities to 1e-12, on uniform and skewed label popularity, several windows, foreign initial
predictions, every metric family, float64 scores, rows of more than 64 entries, and the overflow hand-over."""
import os

import numpy as np
import pytest
from scipy.sparse import csr_matrix

pytestmark = pytest.mark.gpu


def _csr(n, m, r, seed, zipf=False, dtype=np.float32, k=5):
    from xcolumns_amd.synthetic import make_csr
    return make_csr(n, m, r, seed=seed, zipf=zipf, dtype=dtype, k=k)


def _same(P, Po):
    assert np.array_equal(P.indptr, Po.indptr)
    assert np.array_equal(P.indices, Po.indices), f"{int((P.indices != Po.indices).sum())} predicted labels differ"


def _run(Y, k, oref, base, skip_tn=True, sweeps=3, entry="predict_using_bc_with_0approx", **kw):
    import xcolumns_amd.block_coordinate as bc
    import xcolumns_amd.metrics as pm
    names = {oref.FBETA: "binary_f1_score_on_conf_matrix", oref.PRECISION: "binary_precision_on_conf_matrix",
             oref.RECALL: "binary_recall_on_conf_matrix", oref.JACCARD: "binary_jaccard_score_on_conf_matrix",
             oref.BALANCED_ACC: "binary_balanced_accuracy_on_conf_matrix", oref.GMEAN: "binary_gmean_on_conf_matrix",
             oref.HMEAN: "binary_hmean_on_conf_matrix"}
    n, m = Y.shape
    metric = oref.make_metric(base, k=float(k), m=float(m))
    okw = {a: b for a, b in kw.items() if a in ("init_y_pred", "maximize")}
    if isinstance(okw.get("init_y_pred"), csr_matrix):
        okw["init_y_pred"] = okw["init_y_pred"].copy()
    tol = -1.0 if kw.get("maximize", True) else 1.0     # never stop early (block_coordinate.py:486-489)
    Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=skip_tn, seed=13, max_iters=sweeps, tolerance=tol, **okw)
    if isinstance(kw.get("init_y_pred"), csr_matrix):
        kw["init_y_pred"] = kw["init_y_pred"].copy()
    Pg, mg = bc.predict_using_bc_with_0approx(Y, getattr(pm, names[base]), k, skip_tn=skip_tn, seed=13, max_iters=sweeps,
                                              tolerance=tol, return_meta=True, bca_waves=1, bca_ordered=True, **kw)
    assert mg["iters"] == mo["iters"] == sweeps
    d = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
    print("ordered vs oracle, |utility difference| per sweep:", d)
    assert d.max() < 1e-12, (mg["utilities"], mo["utilities"])
    _same(Pg, Po)
    return Pg, mg


@pytest.mark.parametrize("zipf", [False, True])
@pytest.mark.parametrize("workgroups", [None, 3])
def test_ordered_equals_sequential_oracle(oref, monkeypatch, zipf, workgroups):
    """20 K x 3 K, macro-F1: windows of 7680 rows (3 windows) and, with 3 workgroups, 96-row windows (209 windows); with
    Zipf popularity the head labels are put on the dense tables (from 48 readers per window on: the default, 1600,
    leaves this matrix to the change lists alone -- that form is the Zipf cases of the other tests)."""
    if workgroups:
        monkeypatch.setenv("XCOLUMNS_BCA_ORD_WORKGROUPS", str(workgroups))
    if zipf:
        monkeypatch.setenv("XCOLUMNS_BCA_ORD_HOT_READERS", "48")
    Y = _csr(20000, 3000, 30, 11 + int(zipf), zipf=zipf)
    _run(Y, 5, oref, oref.FBETA, sweeps=3)


@pytest.mark.parametrize("metric,skip_tn,maximize", [
    ("PRECISION", True, True), ("RECALL", True, True), ("JACCARD", True, True), ("BALANCED_ACC", False, True),
    ("GMEAN", False, True), ("HMEAN", False, True), ("FBETA", True, False)])
def test_ordered_every_metric(oref, metric, skip_tn, maximize):
    Y = _csr(6000, 900, 24, 5, zipf=True, k=3)
    _run(Y, 3, oref, getattr(oref, metric), skip_tn=skip_tn, sweeps=3, maximize=maximize)


def test_ordered_float64_scores_and_long_rows(oref):
    """float64 y_proba; 100 entries per row = two candidates per lane."""
    Y = _csr(5000, 4000, 100, 9, dtype=np.float64)
    _run(Y, 5, oref, oref.FBETA, sweeps=2)


def test_ordered_foreign_initial_prediction(oref):
    """A random initial prediction: most predicted labels are not stored in their rows (orphans: they leave at the
    row's visit, fp -= 1) -- changes that are known before the row is scored."""
    from xcolumns_amd.utils import random_at_k_csr
    Y = _csr(8000, 2500, 30, 3, zipf=True)
    init = random_at_k_csr(Y.shape, 5, dtype=Y.dtype, seed=4)
    P, _ = _run(Y, 5, oref, oref.FBETA, sweeps=3, init_y_pred=init)
    assert P is not None


def test_ordered_overflow_hands_over_to_one_wavefront(oref, monkeypatch):
    """Change lists that are too short (forced: 1 entry per label besides the dense tables): the kernel stops at the
    window that overflows, the one-wavefront sweep walks the rest of the order -- the same sweep."""
    import xcolumns_amd.block_coordinate as bc
    orig = bc.BcaCsrEngine._ordered_setup

    def tiny(self, orphans):
        d = orig(self, orphans)
        if not d.get("_shrunk"):
            cap = d["lab_dir"][:, 1]
            d["lab_dir"][:, 1] = (cap > 0).to(cap.dtype)      # one entry per label
            d["_shrunk"] = True
        return d

    monkeypatch.setattr(bc.BcaCsrEngine, "_ordered_setup", tiny)
    monkeypatch.setenv("XCOLUMNS_BCA_ORD_HOT", "0")
    Y = _csr(12000, 1500, 30, 21)
    _run(Y, 5, oref, oref.FBETA, sweeps=2)


def test_ordered_is_the_default_where_the_policy_wants_the_exact_sweep(oref):
    """About one predicted row per label (the shape of BASELINE configs[2]) is a shape the default policy runs as the
    reference's exact sequence: with the ordered sweep that no longer means one wavefront."""
    import xcolumns_amd.block_coordinate as bc
    n, m, k = 30000, 134000, 5
    Y = _csr(n, m, 50, 31)
    metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
    Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=13, max_iters=3, tolerance=-1.0)
    Pg, mg = bc.predict_optimizing_macro_f1_score_using_bc(Y, k, seed=13, max_iters=3, tolerance=-1.0, return_meta=True,
                                                           bca_diagnostics=True)
    assert mg["wavefronts"] == [1, 1, 1]
    assert np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"])).max() < 1e-12
    _same(Pg, Po)


def test_ordered_small_random_problems(oref):
    """Sixty random small problems (the generator of tests/studies/fuzz_exact.py: 1-400 rows, ragged rows, a third of them
    with values on a grid of eighths -- exact ties --, every metric, top / random starts, sum and mean aggregation, both
    directions): the ordered parallel sweep behind bca_waves=1 returns the sequential oracle's labels and utilities."""
    import xcolumns_amd.block_coordinate as bc
    metrics = [("binary_precision_on_conf_matrix", oref.PRECISION), ("binary_recall_on_conf_matrix", oref.RECALL),
               ("binary_f1_score_on_conf_matrix", oref.FBETA), ("binary_jaccard_score_on_conf_matrix", oref.JACCARD),
               ("binary_balanced_accuracy_on_conf_matrix", oref.BALANCED_ACC)]
    seen = []
    orig = bc.BcaCsrEngine.sweep_ordered

    def spy(self, order, n_order):
        orig(self, order, n_order)
        seen.append(self.ordered_stats["error"])

    bc.BcaCsrEngine.sweep_ordered = spy
    try:
        for seed in range(60):
            rng = np.random.default_rng(70_000 + seed)
            n, m = int(rng.integers(2, 400)), int(rng.integers(2, 300))
            k = int(rng.integers(1, min(m, 12) + 1))
            rmax = int(rng.integers(k, min(m, 70) + 1))
            lens = rng.integers(k, rmax + 1, size=n)
            cols = np.concatenate([np.sort(rng.choice(m, l, replace=False)) for l in lens]).astype(np.int32)
            vals = (rng.integers(1, 9, size=cols.size) / 8.0) if rng.random() < 0.35 else rng.random(cols.size) ** rng.integers(1, 4)
            Y = csr_matrix((vals.astype(np.float32), cols, np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)), shape=(n, m))
            name, base = metrics[int(rng.integers(len(metrics)))]
            maximize = bool(rng.random() < 0.85)
            kw = dict(seed=int(rng.integers(1000)), max_iters=int(rng.integers(1, 4)), tolerance=-1.0 if maximize else 1.0,
                      skip_tn=base != oref.BALANCED_ACC and bool(rng.random() < 0.5), maximize=maximize,
                      metric_aggregation=str(rng.choice(["mean", "sum"])), init_y_pred=str(rng.choice(["top", "random"])))
            metric = oref.make_metric(base, k=float(k), m=float(m))
            Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, **kw)
            Pg, mg = bc.predict_using_bc_with_0approx(Y, getattr(bc, name), k, return_meta=True, bca_waves=1, **kw)
            assert mg["iters"] == mo["iters"], (seed, kw)
            assert np.allclose(mg["utilities"], mo["utilities"], rtol=1e-13, atol=1e-12), (seed, name, kw, mg["utilities"], mo["utilities"])
            assert np.array_equal(Pg.indices, Po.indices), (seed, name, kw)
    finally:
        bc.BcaCsrEngine.sweep_ordered = orig
    assert len(seen) >= 60 and all(e in (0, 1, 3) for e in seen)     # the ordered sweep ran (1 / 3: handed the rest to one wavefront)
    print("ordered sweeps:", len(seen), "handed over:", sum(e != 0 for e in seen))

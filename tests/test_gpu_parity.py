"""GPU parity tests: the HIP path (through the C ABI) against the golden fixtures
generated from the reference and against the CPU oracle on seeded inputs.

Bars: bit-exact for index sets and for values computed without reductions;
1e-12 for float64 column sums (atomic order); BCA utilities 1e-12 and identical
predictions in the exact mode (bca_waves=1: the ordered parallel sweep, or ONE
wavefront with bca_ordered=False -- both are the reference's sequence).  In the
concurrent mode (the product default where it holds the bar) rows in flight miss
each other's updates, so the trajectory is not the sequential one: the utility after
EVERY sweep must be within north_star's 1e-5 of the sequential oracle's, with a
margin (tests/_parity.py: fail above 0.5 of the bar, three visiting orders each)."""
import numpy as np
import pytest
import torch
from scipy.sparse import csr_matrix

import _golden as G
import _parity

pytestmark = pytest.mark.gpu

FINAL_TOL = 1e-5       # north_star: utility within 1e-5 after the same number of iterations
PER_SWEEP_TOL = 1e-5   # ... after ANY number of iterations: every intermediate sweep too (default bca_parity="per_sweep")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    from xcolumns_amd import _lib
    info = _lib.device_info()
    assert info["arch"].startswith("gfx950"), info


def _same_csr(a, b, data_exact=True):
    assert a.shape == b.shape
    assert np.array_equal(np.asarray(a.indptr, dtype=np.int64), np.asarray(b.indptr, dtype=np.int64))
    nnz = int(b.indptr[-1])
    assert np.array_equal(a.indices[:nnz], b.indices[:nnz])
    if data_exact:
        assert np.array_equal(a.data[:nnz], b.data[:nnz])


# ---------------------------------------------------------------------------
# weighted top-k
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_topk_csr_golden(tag):
    from xcolumns_amd.weighted_prediction import predict_weighted_per_instance
    z = G.load("topk_csr_" + tag)
    Y = G.csr_from(z, "y")
    k = int(z["k"])
    a, b = z["a"], z["b"]
    cases = {
        "plain": dict(), "scores": dict(keep_scores=True), "ab": dict(a=a, b=b),
        "ab_scores": dict(a=a, b=b, keep_scores=True), "a_only": dict(a=a), "b_only": dict(b=b),
    }
    for name, kw in cases.items():
        P = predict_weighted_per_instance(Y, k, **kw)
        assert isinstance(P, csr_matrix) and P.dtype == Y.dtype and P.shape == Y.shape
        assert P.indices.dtype == Y.indices.dtype and P.indptr.dtype == Y.indptr.dtype
        _same_csr(P, G.csr_from(z, "pred_" + name))
    th = float(z["th"])
    _same_csr(predict_weighted_per_instance(Y, 0, th=th, a=a, b=b), G.csr_from(z, "pred_k0_ab"))
    _same_csr(predict_weighted_per_instance(Y, 0, th=th), G.csr_from(z, "pred_k0_plain"))


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("kind", ["numpy", "torch_cpu", "torch_gpu"])
def test_topk_dense_golden(tag, kind):
    from xcolumns_amd.weighted_prediction import predict_weighted_per_instance

    z = G.load("topk_dense_" + tag)
    Y, k = z["y"], int(z["k"])
    conv = {
        "numpy": lambda x: x,
        "torch_cpu": lambda x: torch.from_numpy(x),
        "torch_gpu": lambda x: torch.from_numpy(x).cuda(),
    }[kind]
    back = (lambda p: p) if kind == "numpy" else (lambda p: p.cpu().numpy())
    a, b, a32, b32 = z["a"], z["b"], z["a32"], z["b32"]
    th = float(z["th"])
    cases = {
        "plain": dict(), "scores": dict(keep_scores=True), "ab": dict(a=conv(a), b=conv(b)),
        "ab_scores": dict(a=conv(a), b=conv(b), keep_scores=True), "ab32": dict(a=conv(a32), b=conv(b32)),
        "ab32_scores": dict(a=conv(a32), b=conv(b32), keep_scores=True),
    }
    Yin = conv(Y)
    for name, kw in cases.items():
        P = predict_weighted_per_instance(Yin, k, **kw)
        assert type(P) == type(Yin) and P.dtype == Yin.dtype and tuple(P.shape) == Y.shape
        if kind != "numpy":
            assert P.device == Yin.device
        assert np.array_equal(back(P), z["pred_" + name]), name
    assert np.array_equal(back(predict_weighted_per_instance(Yin, 0, th=th, a=conv(a), b=conv(b))), z["pred_k0_ab"])
    assert np.array_equal(back(predict_weighted_per_instance(Yin, 0, th=th)), z["pred_k0_plain"])


@pytest.mark.parametrize("k", [1, 5, 17, 64])
def test_topk_csr_short_rows_adversarial_vs_oracle(oref, k):
    """float32 rows of at most 64 entries take the four-rows-per-wavefront kernel: rows of every length
    0..64 (so k > r, k == r, k < r all occur), heavily tied scores (values on a grid of 8: the lowest
    column must win), negative and zero weights, +-inf and NaN gains, keep_scores, eta / sel outputs of
    the BCA initialisation.  Index sets must be identical to the oracle's."""
    from xcolumns_amd import _device as D
    from xcolumns_amd.weighted_prediction import predict_weighted_per_instance, topk_csr_device
    rng = np.random.default_rng(100 + k)
    n, m = 4099, 500   # not a multiple of 4 rows per wave
    lens = rng.integers(0, 65, size=n)
    lens[:65] = np.arange(65)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cols = np.concatenate([np.sort(rng.choice(m, r, replace=False)) for r in lens]).astype(np.int32)
    data = (rng.integers(0, 8, size=cols.size) / 8.0).astype(np.float32)      # ties everywhere
    Y = csr_matrix((data, cols, indptr), shape=(n, m))
    a = rng.normal(size=m).astype(np.float32)
    a[::7] = 0.0
    b = rng.normal(size=m).astype(np.float32)
    b[3], b[11], b[19] = np.inf, -np.inf, np.nan
    for kw in (dict(), dict(a=a, b=b), dict(a=a), dict(b=b, keep_scores=True)):
        Pg = predict_weighted_per_instance(Y, k, **kw)
        Po = oref.predict_weighted_per_instance(Y, k, **kw)
        _same_csr(Pg, Po, data_exact=False)
        if kw.get("keep_scores"):
            assert np.array_equal(Pg.data, Po.data, equal_nan=True)
    # the BCA initialisation outputs (eta of the chosen entries, per-entry flags); rows need >= k entries
    keep = lens >= k
    Yk = Y[keep]
    csr = D.DeviceCSR.from_scipy(Yk, D.require_gpu())
    sel = torch.zeros(csr.nnz, dtype=torch.uint8, device="cuda")
    idx, dat, eta = topk_csr_device(csr, k, want_eta=True, out_sel=sel)
    Po = oref.predict_top_k(Yk, k)
    assert np.array_equal(idx.cpu().numpy(), Po.indices)
    dense = Yk.toarray()
    rows = np.repeat(np.arange(Yk.shape[0]), k)
    assert np.array_equal(eta.cpu().numpy(), dense[rows, Po.indices])
    flags = sel.cpu().numpy()
    assert flags.sum() == Yk.shape[0] * k
    chosen = csr_matrix((flags.astype(np.float32), Yk.indices, Yk.indptr), shape=Yk.shape)
    chosen.eliminate_zeros()
    assert np.array_equal(chosen.indices, Po.indices)


@pytest.mark.parametrize("m", [1, 255, 257, 1000, 4097, 8192, 9000])
def test_topk_dense_adversarial_vs_oracle(oref, m):
    """Dense rows of float32 gains: register-resident kernel up to 8192 labels, the streaming one beyond;
    tied scores on a coarse grid (lowest column wins), -0.0 / +0.0, +-inf and NaN, k >= m."""
    from xcolumns_amd.weighted_prediction import predict_weighted_per_instance
    rng = np.random.default_rng(m)
    n = 67
    Y = (rng.integers(-4, 5, size=(n, m)) / 4.0).astype(np.float32)
    Y[Y == 0] = np.where(rng.random((Y == 0).sum()) < 0.5, -0.0, 0.0)
    if m > 10:
        Y[3, 5], Y[3, 7], Y[4, 2], Y[5, :4] = np.inf, -np.inf, np.nan, np.nan
    for k in sorted({1, min(5, m), min(64, m), m if m <= 1000 else min(300, m)}):
        for keep in (False, True):
            Pg = predict_weighted_per_instance(Y, k, keep_scores=keep)
            Po = oref.predict_weighted_per_instance(Y, k, keep_scores=keep)
            assert Pg.dtype == Y.dtype and Pg.shape == Y.shape
            assert np.array_equal(Pg, Po, equal_nan=True), (m, k, keep, np.argwhere(~((Pg == Po) | (np.isnan(Pg) & np.isnan(Po))))[:5])


def test_topk_csr_long_rows_vs_oracle(oref):
    """Rows of up to 1000 entries (the *_1000_* prediction files of the reference's
    experiments) exercise the multi-chunk register path; oracle is the checker."""
    from xcolumns_amd.weighted_prediction import predict_weighted_per_instance
    rng = np.random.default_rng(7)
    n, m, k = 300, 5000, 5
    lens = rng.integers(5, 1001, size=n)
    lens[:4] = [64, 65, 128, 1000]
    cols = [np.sort(rng.choice(m, l, replace=False)) for l in lens]
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    Y = csr_matrix((rng.random(indptr[-1]).astype(np.float32), np.concatenate(cols).astype(np.int32), indptr),
                   shape=(n, m))
    a = rng.random(m) + 0.5
    for kw in (dict(), dict(a=a), dict(a=a, keep_scores=True)):
        _same_csr(predict_weighted_per_instance(Y, k, **kw), oref.predict_weighted_per_instance(Y, k, **kw))


# ---------------------------------------------------------------------------
# confusion matrix
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_confusion_golden(tag):
    """The CSR and dense kernels against the reference's outputs, incl. the padded-prediction quirk."""
    from xcolumns_amd.confusion_matrix import calculate_confusion_matrix
    z = G.load("confusion_" + tag)
    mats = {n: G.csr_from(z, n) for n in ("y", "p", "prand", "l")}
    for tname in ("y", "l"):
        for pname in ("p", "prand"):
            for skip_tn in (False, True):
                for normalize in (False, True):
                    C = calculate_confusion_matrix(mats[tname], mats[pname], normalize=normalize,
                                                   skip_tn=skip_tn, dtype=np.float64)
                    exp = z[f"C_{tname}_{pname}_skip{int(skip_tn)}_norm{int(normalize)}"]
                    got = np.stack(list(C))
                    assert got.dtype == np.float64
                    assert np.allclose(got, exp, rtol=1e-12, atol=1e-12), (tname, pname, skip_tn, normalize)
    for tname in ("yd", "ld"):
        for skip_tn in (False, True):
            C = calculate_confusion_matrix(z[tname], z["pd"], skip_tn=skip_tn, dtype=np.float64)
            assert np.allclose(np.stack(list(C)), z[f"C_{tname}_pd_skip{int(skip_tn)}_norm0"], rtol=1e-12, atol=1e-12)
            Ct = calculate_confusion_matrix(torch.from_numpy(z[tname]).cuda(), torch.from_numpy(z["pd"]).cuda(),
                                            skip_tn=skip_tn, dtype=torch.float64)
            assert Ct.tp.is_cuda
            assert np.allclose(np.stack([v.cpu().numpy() for v in Ct]), z[f"C_{tname}_pd_skip{int(skip_tn)}_norm0"],
                               rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_confusion_vs_oracle_large(oref, dtype):
    """300 K x 200 K, ragged rows (some shorter than k: the reference's top-k pads them with column 0, unsorted --
    the general kernel's merge replay): the kernel and the oracle agree."""
    from xcolumns_amd.confusion_matrix import calculate_confusion_matrix
    from xcolumns_amd.weighted_prediction import predict_top_k
    rng = np.random.default_rng(11)
    n, m, k = 300_000, 200_000, 5
    lens = rng.integers(0, 24, size=n)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cols = rng.integers(0, m, size=int(indptr[-1])).astype(np.int32)
    Y = csr_matrix((rng.random(indptr[-1]).astype(dtype), cols, indptr), shape=(n, m))
    Y.sum_duplicates()
    Y.sort_indices()
    P = predict_top_k(Y, k)
    exp = np.stack(list(oref.calculate_confusion_matrix(Y, P, skip_tn=True)))[:3]
    C = calculate_confusion_matrix(Y, P, skip_tn=True, dtype=np.float64)
    assert np.allclose(np.stack([C.tp, C.fp, C.fn]), exp, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_confusion_prediction_side_vs_oracle(oref, dtype, monkeypatch):
    """Large inputs with well-formed rows take xc_confusion_csr_pred_side (atomics for the predicted entries only, fn
    from the cached column sums of y_true): same statistics as the oracle and as the general kernel, first call and
    cached; a prediction with a padded short row falls back to the general kernel."""
    from xcolumns_amd import DeviceCSR
    from xcolumns_amd.confusion_matrix import calculate_confusion_matrix
    from xcolumns_amd.weighted_prediction import predict_top_k
    rng = np.random.default_rng(12)
    n, m, k = 200_000, 150_000, 5
    lens = rng.integers(k, 30, size=n)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cols = rng.integers(0, m, size=int(indptr[-1])).astype(np.int32)
    Y = csr_matrix((rng.random(indptr[-1]).astype(dtype), cols, indptr), shape=(n, m))
    Y.sum_duplicates()
    Y.sort_indices()
    Y = Y[np.diff(Y.indptr) >= k]                      # duplicates may have shortened a row below k
    n = Y.shape[0]
    T = Y.copy()
    T.data = (rng.random(T.nnz) < 0.4).astype(dtype) * T.data     # y_true: some entries kept, some explicit zeros
    Td, Yd = DeviceCSR.from_scipy(T), DeviceCSR.from_scipy(Y)
    # a 0/1 prediction (float32: matched, then summed by a counting sort -- no global atomics: xc_confusion_csr_match +
    # xc_scatter_sum_f32) and one that keeps its scores (atomics for the predicted entries)
    for keep_scores in (False, True):
        Pd = predict_top_k(Yd, k, keep_scores=keep_scores)
        P = Pd.to_scipy()
        exp = np.stack(oref.calculate_confusion_matrix(T, P, skip_tn=True)[:3])
        for call in range(2):                           # second call: column sums and row check come from the cache
            C = calculate_confusion_matrix(Td, Pd, skip_tn=True, dtype=torch.float64)
            got = np.stack([C.tp.cpu().numpy(), C.fp.cpu().numpy(), C.fn.cpu().numpy()])
            assert np.allclose(got, exp, rtol=1e-12, atol=1e-12), (keep_scores, call)
            assert "_colsum64" in Td.__dict__ and Td.rows_ascending()
        if not keep_scores:
            monkeypatch.setenv("XCOLUMNS_CONFUSION_SCATTER", "0")   # the same through the atomic form
            C = calculate_confusion_matrix(Td, Pd, skip_tn=True, dtype=torch.float64)
            assert np.allclose(np.stack([C.tp.cpu().numpy(), C.fp.cpu().numpy(), C.fn.cpu().numpy()]), got, rtol=1e-12, atol=1e-12)
            monkeypatch.delenv("XCOLUMNS_CONFUSION_SCATTER")
    monkeypatch.setenv("XCOLUMNS_CONFUSION_PRED_SIDE", "0")
    C = calculate_confusion_matrix(Td, Pd, skip_tn=True, dtype=torch.float64)
    assert np.allclose(np.stack([C.tp.cpu().numpy(), C.fp.cpu().numpy(), C.fn.cpu().numpy()]), got, rtol=1e-12, atol=1e-12)
    monkeypatch.delenv("XCOLUMNS_CONFUSION_PRED_SIDE")
    # a padded row in the prediction (k + 1 labels from rows that hold only k): the fast kernel flags it, the general one answers
    short = int(np.argmin(np.diff(Y.indptr)))
    if Y.indptr[short + 1] - Y.indptr[short] == k:
        P6 = predict_top_k(Y, k + 1)
        exp6 = np.stack(oref.calculate_confusion_matrix(T, P6, skip_tn=True)[:3])
        C6 = calculate_confusion_matrix(Td, DeviceCSR.from_scipy(P6), skip_tn=True, dtype=torch.float64)
        assert np.allclose(np.stack([C6.tp.cpu().numpy(), C6.fp.cpu().numpy(), C6.fn.cpu().numpy()]), exp6, rtol=1e-12, atol=1e-12)


# ---------------------------------------------------------------------------
# BCA, exact sequential mode (one wavefront walks the order): must reproduce the
# reference's trajectory
# ---------------------------------------------------------------------------

def _check_bca(P, meta, z, name, pred_key, tol):
    exp_u = z["utilities" + ("_" + name if name else "")]
    assert meta["iters"] == int(z["iters" + ("_" + name if name else "")]), (name, meta)
    assert np.allclose(np.asarray(meta["utilities"]), exp_u, rtol=0, atol=tol), (name, meta["utilities"], exp_u)
    assert set(meta) == {"utilities", "iters", "time"}


@pytest.mark.parametrize("ordered", [True, False])
def test_bca_csr_anchor_exact(ordered):
    z = G.load("bca_csr_anchor_f32")
    Y = G.csr_from(z, "y")
    P, meta = G.product_call_from_spec(G.spec_of(z), Y, bca_waves=1, bca_ordered=ordered)
    _check_bca(P, meta, z, None, "pred", 1e-12)
    assert isinstance(P, csr_matrix) and P.dtype == Y.dtype
    _same_csr(P, G.csr_from(z, "pred"))


@pytest.mark.parametrize("ordered", [True, False])   # the ordered parallel sweep / one wavefront: the same sequence
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_bca_csr_golden_exact(tag, ordered):
    z = G.load("bca_csr_" + tag)
    Yu, Yz, init = G.csr_from(z, "yu"), G.csr_from(z, "yz"), G.csr_from(z, "init")
    for name in [str(s) for s in z["names"]]:
        spec = G.spec_of(z, name)
        Y = Yz if spec.get("data") == "z" else Yu
        init_m = init.copy()
        P, meta = G.product_call_from_spec(spec, Y, init_matrix=init_m, bca_waves=1, bca_ordered=ordered)
        _check_bca(P, meta, z, name, "pred_" + name, 1e-12)
        assert P.dtype == Y.dtype and (np.diff(P.indptr) == spec["k"]).all()
        _same_csr(P, G.csr_from(z, "pred_" + name))
        if spec.get("has_init_matrix"):
            assert P is init_m  # an explicit init_y_pred is updated in place and returned


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("kind", ["numpy", "torch_gpu"])
def test_bca_dense_golden(tag, kind):
    z = G.load("bca_dense_" + tag)
    Y = z["y"]
    Yin = Y if kind == "numpy" else torch.from_numpy(Y).cuda()
    for name in [str(s) for s in z["names"]]:
        spec = G.spec_of(z, name)
        P, meta = G.product_call_from_spec(spec, Yin, bca_waves=1)
        _check_bca(P, meta, z, name, "pred_" + name, 1e-12)
        assert type(P) == type(Yin) and P.dtype == Yin.dtype
        got = P if kind == "numpy" else P.cpu().numpy()
        assert np.array_equal(got, z["pred_" + name]), name


# ---------------------------------------------------------------------------
# BCA, concurrent mode (the product default) against the oracle
# ---------------------------------------------------------------------------

def _synthetic_csr(n, m, r, seed, zipf=False, dtype=np.float32):
    from xcolumns_amd.synthetic import make_csr
    return make_csr(n, m, r, seed=seed, zipf=zipf, dtype=dtype)


@pytest.mark.parametrize("zipf", [False, True])
def test_bca_csr_concurrent_vs_oracle(oref, zipf):
    """n=20K rows, default concurrency policy: the final utility is within 1e-5 of the
    sequential oracle's after the same number of sweeps, every sweep within
    PER_SWEEP_TOL, and the result is a valid prediction."""
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc
    n, m, r, k = 20000, 3000, 30, 5
    Y = _synthetic_csr(n, m, r, 11 + int(zipf), zipf=zipf)
    metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
    for seed in _parity.SEEDS:
        Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=seed, max_iters=4, tolerance=-1.0)
        Pg, mg = predict_optimizing_macro_f1_score_using_bc(Y, k, seed=seed, max_iters=4, tolerance=-1.0, return_meta=True)
        assert mg["iters"] == mo["iters"] == 4
        diff = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
        _parity.check(diff, f"20K x 3K {'zipf' if zipf else 'uniform'}, seed {seed}")
    assert (np.diff(Pg.indptr) == k).all() and Pg.dtype == Y.dtype
    # every predicted label is stored in its row, ids ascending and distinct
    for i in range(0, n, 997):
        pi = Pg.indices[i * k:(i + 1) * k]
        assert (np.diff(pi) > 0).all() and np.isin(pi, Y.indices[Y.indptr[i]:Y.indptr[i + 1]]).all()
    # utility improves on top-k and never decreases by more than the staleness noise
    u = np.asarray(mg["utilities"])
    assert (np.diff(u) > -1e-6).all()


def test_bca_csr_full_size_properties(oref):
    """BASELINE.json configs[1] shape (n=100K, m=30K, 50 entries/row, k=5): size-independent
    properties + the oracle's utilities (the oracle needs ~2 s here)."""
    from xcolumns_amd import _device as D
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc
    from xcolumns_amd.confusion_matrix import calculate_confusion_matrix
    from xcolumns_amd.metrics import binary_f1_score_on_conf_matrix
    n, m, r, k = 100_000, 30_000, 50, 5
    Y = _synthetic_csr(n, m, r, 20240001)
    Pg, mg = predict_optimizing_macro_f1_score_using_bc(Y, k, seed=13, max_iters=4, tolerance=-1.0, return_meta=True)
    assert (np.diff(Pg.indptr) == k).all()
    ids = Pg.indices.reshape(n, k)
    assert (np.diff(ids, axis=1) > 0).all()
    # independent kernel (xc_confusion_csr) recomputes the statistics of the returned prediction:
    # macro-F1 of them must equal the last utility the driver reported
    C = calculate_confusion_matrix(Y, Pg, normalize=True, skip_tn=True, dtype=np.float64)
    f1 = binary_f1_score_on_conf_matrix(C.tp, C.fp, C.fn, C.tn).mean()
    assert abs(f1 - mg["utilities"][-1]) < 1e-10, (f1, mg["utilities"])
    metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
    for seed in _parity.SEEDS:
        Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=seed, max_iters=4, tolerance=-1.0)
        mg2 = mg if seed == 13 else predict_optimizing_macro_f1_score_using_bc(Y, k, seed=seed, max_iters=4, tolerance=-1.0,
                                                                                return_meta=True)[1]
        _parity.check(np.abs(np.asarray(mg2["utilities"]) - np.asarray(mo["utilities"])), f"C2 100K x 30K, seed {seed}")
    # idempotence at convergence is not guaranteed after 3 sweeps, but top-k must be improved upon
    top = oref.predict_top_k(Y, k)
    tp, fp, fn, tn = oref.calculate_confusion_matrix(Y, top, skip_tn=True)
    u_top = oref.calculate_utility(metric, "mean", tp / n, fp / n, fn / n, tn / n)
    assert mg["utilities"][0] > u_top


# ---------------------------------------------------------------------------
# API contract
# ---------------------------------------------------------------------------

def test_api_contract_and_errors():
    from xcolumns_amd.block_coordinate import predict_using_bc_with_0approx
    from xcolumns_amd.metrics import binary_recall_on_conf_matrix
    from xcolumns_amd.weighted_prediction import predict_top_k, predict_weighted_per_instance
    rng = np.random.default_rng(5)
    Yd = rng.random((50, 12)).astype(np.float32)
    Y = csr_matrix(Yd * (Yd > 0.3))
    with pytest.raises(ValueError):
        predict_using_bc_with_0approx(Y, binary_recall_on_conf_matrix, 2.0)
    with pytest.raises(ValueError):
        predict_using_bc_with_0approx([[1.0]], binary_recall_on_conf_matrix, 2)
    with pytest.raises(ValueError):
        predict_using_bc_with_0approx(Yd, binary_recall_on_conf_matrix, 2, init_y_pred=np.zeros((3, 3)))
    with pytest.raises(ValueError):
        predict_using_bc_with_0approx(Yd, binary_recall_on_conf_matrix, 2, metric_aggregation="median")
    with pytest.raises(NotImplementedError):
        predict_using_bc_with_0approx(Yd, lambda tp, fp, fn, tn: tp, 2)
    with pytest.raises(ValueError):
        predict_weighted_per_instance(Yd, 2, a=np.ones(3))
    with pytest.raises(ValueError):
        predict_weighted_per_instance(Yd, "2")
    with pytest.raises(ValueError):
        predict_weighted_per_instance(np.zeros((2, 2, 2)), 1)
    # column ids outside [0, m) never reach a kernel (scipy does not validate them)
    bad = csr_matrix(Yd)
    bad.indices = bad.indices.copy()
    bad.indices[7] = 12
    with pytest.raises(ValueError, match="column ids"):
        predict_top_k(bad, 2)
    with pytest.raises(ValueError, match="column ids"):
        predict_using_bc_with_0approx(bad, binary_recall_on_conf_matrix, 2)
    init = predict_top_k(csr_matrix(Yd), 2)
    init.indices[3] = -1
    with pytest.raises(ValueError, match="column ids"):
        predict_using_bc_with_0approx(csr_matrix(Yd), binary_recall_on_conf_matrix, 2, init_y_pred=init)
    # 1-d input is one row (weighted_prediction.py:142-143)
    one = predict_top_k(Yd[0], 3)
    assert one.shape == (1, 12) and one.sum() == 3
    # same test the reference runs: type / dtype preserved, k ones per row
    for inp in (Yd, Yd.astype(np.float64), csr_matrix(Yd), torch.from_numpy(Yd), torch.from_numpy(Yd).cuda()):
        for init in ("random", "greedy", "top"):
            if isinstance(inp, csr_matrix) and init != "top":
                pass
            P, meta = predict_using_bc_with_0approx(inp, binary_recall_on_conf_matrix, 3, return_meta=True,
                                                    seed=2024, init_y_pred=init)
            assert type(P) == type(inp) and P.dtype == inp.dtype
            s = P.sum(axis=1) if not isinstance(P, torch.Tensor) else P.sum(dim=1).cpu().numpy()
            assert (np.asarray(s).ravel() == 3).all()


# ---------------------------------------------------------------------------
# harder shapes for the sweep kernel, exact sequential mode vs the oracle
# ---------------------------------------------------------------------------

def _ragged_csr(n, m, rmin, rmax, seed, dtype=np.float32, quantize=None):
    rng = np.random.default_rng(seed)
    lens = rng.integers(rmin, rmax + 1, size=n)
    lens[:5] = np.minimum([rmin, 64, 65, 129, rmax], rmax)
    cols = np.concatenate([np.sort(rng.choice(m, l, replace=False)) for l in lens]).astype(np.int32)
    data = rng.random(cols.size) ** 2
    if quantize:
        data = np.ceil(data * quantize) / quantize      # few distinct values -> many exactly equal gains
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    return csr_matrix((np.clip(data, 1e-3, 1.0).astype(dtype), cols, indptr), shape=(n, m))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_bca_csr_long_ragged_rows_exact(oref, dtype):
    """Rows of 5..700 entries: 1, 2, 4, 8 and 16 candidates per lane, chunked ballots,
    the swap loop and the bisection fallback across chunks."""
    from xcolumns_amd.block_coordinate import predict_using_bc_with_0approx
    from xcolumns_amd.metrics import binary_f1_score_on_conf_matrix, binary_balanced_accuracy_on_conf_matrix
    n, m, k = 500, 3000, 5
    Y = _ragged_csr(n, m, k, 700, 31, dtype=dtype)
    for fn_, base, skip_tn in ((binary_f1_score_on_conf_matrix, oref.FBETA, True),
                              (binary_balanced_accuracy_on_conf_matrix, oref.BALANCED_ACC, False)):
        metric = oref.make_metric(base, k=float(k), m=float(m))
        Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=skip_tn, seed=3, max_iters=3, tolerance=-1.0)
        Pg, mg = predict_using_bc_with_0approx(Y, fn_, k, skip_tn=skip_tn, seed=3, max_iters=3, tolerance=-1.0,
                                               return_meta=True, bca_waves=1)
        assert np.allclose(mg["utilities"], mo["utilities"], rtol=0, atol=1e-12), (mg["utilities"], mo["utilities"])
        assert np.array_equal(Pg.indices, Po.indices)
        # concurrent mode on the same input: valid prediction, close utility
        Pc, mc = predict_using_bc_with_0approx(Y, fn_, k, skip_tn=skip_tn, seed=3, max_iters=3, tolerance=-1.0,
                                               return_meta=True, bca_waves=4)
        dc = np.abs(np.asarray(mc["utilities"]) - np.asarray(mo["utilities"]))
        print("long ragged rows, 4 wavefronts on 500 rows:", dc)
        assert dc.max() < PER_SWEEP_TOL    # a forced width (the default policy runs this matrix exactly): the bar, no margin
        assert (np.diff(Pc.indices.reshape(n, k), axis=1) > 0).all()


def test_bca_csr_ties_exact(oref):
    """Scores quantised to 8 levels: many exactly equal gains at the top-k boundary;
    the lower column must win, as in the oracle."""
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_recall_using_bc, predict_optimizing_macro_f1_score_using_bc
    n, m, k = 600, 40, 3
    Y = _ragged_csr(n, m, k, 30, 77, quantize=8)
    for fn_, base in ((predict_optimizing_macro_f1_score_using_bc, oref.FBETA),
                      (predict_optimizing_macro_recall_using_bc, oref.RECALL)):
        metric = oref.make_metric(base, k=float(k), m=float(m))
        Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=5, max_iters=3, tolerance=-1.0)
        Pg, mg = fn_(Y, k, seed=5, max_iters=3, tolerance=-1.0, return_meta=True, bca_waves=1)
        assert np.allclose(mg["utilities"], mo["utilities"], rtol=0, atol=1e-12)
        assert np.array_equal(Pg.indices, Po.indices)
    # top-k itself with ties
    from xcolumns_amd.weighted_prediction import predict_top_k
    assert np.array_equal(predict_top_k(Y, k).indices, oref.predict_top_k(Y, k).indices)


def test_bca_csr_not_normalized_quirk(oref):
    """normalize_conf_matrix=False: the reference then visits ONLY row 0 and divides the
    utility by 1 (block_coordinate.py:403-405, :414); the step still divides by n."""
    from xcolumns_amd.block_coordinate import predict_using_bc_with_0approx
    from xcolumns_amd.metrics import binary_f1_score_on_conf_matrix
    n, m, k = 300, 50, 3
    Y = _ragged_csr(n, m, k, 20, 9)
    metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
    Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=1, max_iters=2, tolerance=-1.0,
                                                normalize_conf_matrix=False)
    Pg, mg = predict_using_bc_with_0approx(Y, binary_f1_score_on_conf_matrix, k, skip_tn=True, seed=1, max_iters=2,
                                           tolerance=-1.0, normalize_conf_matrix=False, return_meta=True, bca_waves=1)
    assert np.allclose(mg["utilities"], mo["utilities"], rtol=0, atol=1e-12)
    assert np.array_equal(Pg.indices, Po.indices)


def test_bca_shadow_on_off_agree():
    """The float32 shadow of the {tp, fp} records (gathered by the concurrent sweep) and
    the float64 records themselves lead to the same optimum within the parity bar."""
    from xcolumns_amd import _device as D, _lib
    from xcolumns_amd.block_coordinate import BcaCsrEngine, WavePolicy, run_bca_sweeps
    from xcolumns_amd.metrics import MetricSpec
    n, m, k = 30000, 5000, 5
    Y = _synthetic_csr(n, m, 40, 4242)
    dev = D.require_gpu()
    csr = D.DeviceCSR.from_scipy(Y, dev)
    spec = MetricSpec(base=_lib.XC_M_FBETA)
    finals = []
    for shadow in (True, False):
        eng = BcaCsrEngine(csr, k, spec, spec, skip_tn=True, use_shadow=shadow)
        eng.init_top()
        rng = np.random.default_rng(13)
        order = np.arange(n)

        def nxt():
            rng.shuffle(order)
            return torch.from_numpy(order.astype(np.int32)).to(dev)

        meta = {"utilities": [], "iters": 0}
        run_bca_sweeps(eng, nxt, n, n, m, "mean", True, -1.0, 5, False, WavePolicy(n, m=m, row_nnz=Y.nnz / n), False, meta)
        finals.append(meta["utilities"])
    d = np.abs(np.asarray(finals[0]) - np.asarray(finals[1]))
    print("shadow on/off utility diff per sweep:", d)
    assert d[-1] < FINAL_TOL and d.max() < PER_SWEEP_TOL


# ---------------------------------------------------------------------------
# the concurrent (non-exact) arithmetic for EVERY metric: psi(x / n; eps, k) is
# evaluated as psi(x; eps * n, k * n) with reciprocal-based divisions
# ---------------------------------------------------------------------------

CONCURRENT_METRIC_CASES = [
    # (xcolumns_amd wrapper or generic metric name, oracle base, skip_tn, aggregation, mixed, metric_kwargs)
    ("binary_precision_on_conf_matrix", "PRECISION", True, "mean", False, {"epsilon": 0.05}),
    ("binary_recall_on_conf_matrix", "RECALL", True, "mean", False, {"epsilon": 0.05}),
    ("binary_fbeta_score_on_conf_matrix", "FBETA", True, "mean", False, {"beta": 2.0, "epsilon": 0.05}),
    ("binary_jaccard_score_on_conf_matrix", "JACCARD", True, "mean", False, {"epsilon": 0.05}),
    ("binary_balanced_accuracy_on_conf_matrix", "BALANCED_ACC", False, "mean", False, {"epsilon": 0.05}),
    ("binary_gmean_on_conf_matrix", "GMEAN", False, "mean", False, {"epsilon": 0.05}),
    ("binary_hmean_on_conf_matrix", "HMEAN", False, "mean", False, {"epsilon": 0.05}),
    ("binary_accuracy_on_conf_matrix", "ACCURACY", False, "mean", False, {}),
]


@pytest.mark.parametrize("case", CONCURRENT_METRIC_CASES, ids=[c[1] for c in CONCURRENT_METRIC_CASES])
def test_bca_concurrent_arithmetic_all_metrics(oref, case):
    """A large epsilon makes the eps * n rescaling of the concurrent path visible: a
    wrong scaling would move the optimum by far more than the tolerance."""
    import xcolumns_amd.metrics as pm
    from xcolumns_amd.block_coordinate import predict_using_bc_with_0approx
    name, base, skip_tn, agg, mixed, mk = case
    n, m, k = 8000, 400, 4
    Y = _synthetic_csr(n, m, 25, 555)
    metric = oref.make_metric(getattr(oref, base), epsilon=mk.get("epsilon", 1e-9), beta=mk.get("beta", 1.0),
                              k=float(k), m=float(m))
    umetric = oref.make_metric(getattr(oref, base), k=float(k), m=float(m))
    Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, metric_aggregation=agg, skip_tn=skip_tn, seed=21,
                                                max_iters=5, tolerance=-1.0, utility_metric=umetric)
    Pg, mg = predict_using_bc_with_0approx(Y, getattr(pm, name), k, metric_aggregation=agg, skip_tn=skip_tn, seed=21,
                                           max_iters=5, tolerance=-1.0, metric_kwargs=mk or None, return_meta=True,
                                           bca_waves=8)
    d = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
    print(base, "concurrent(8 waves) vs sequential:", d)
    assert d.max() < PER_SWEEP_TOL, (mg["utilities"], mo["utilities"])   # every sweep (measured <= 5e-6 with the commit protocol)
    same = (Pg.indices.reshape(n, k) == Po.indices.reshape(n, k)).all(axis=1).mean()
    assert same > 0.97, same


@pytest.mark.parametrize("entry", ["predict_optimizing_instance_precision_using_bc",
                                   "predict_optimizing_mixed_instance_precision_and_macro_f1_score_using_bc",
                                   "predict_optimizing_mixed_instance_precision_and_macro_hmean_using_bc"])
def test_bca_concurrent_arithmetic_kf_scaling(oref, entry):
    """precision@k and the mixed utilities carry the k * n rescaling."""
    import xcolumns_amd.block_coordinate as bc
    n, m, k = 8000, 400, 4
    Y = _synthetic_csr(n, m, 25, 556)
    spec = {"entry": entry, "k": k, "kwargs": dict(seed=22, max_iters=4, tolerance=-1.0, init_y_pred="top")}
    if "mixed" in entry:
        spec["kwargs"]["alpha"] = 0.6
    Po, mo = G.oracle_call_from_spec(oref, spec, Y)
    Pg, mg = getattr(bc, entry)(Y, k, return_meta=True, bca_waves=8, **spec["kwargs"])
    d = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
    scale = max(1.0, abs(mo["utilities"][-1]))
    print(entry, d, mo["utilities"])
    assert d.max() < PER_SWEEP_TOL * scale
    same = (Pg.indices.reshape(n, k) == Po.indices.reshape(n, k)).all(axis=1).mean()
    assert same > 0.97, same


def test_bca_csr_north_star_size_properties():
    """BASELINE.json's north-star shape (n=1M, m=500K, 50 entries/row, k=5): too large for
    the oracle in a test, so size-independent properties: valid prediction, the
    reported utility equals macro-F1 recomputed by the independent confusion kernel,
    BCA improves on top-k and does not get worse sweep over sweep."""
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc
    from xcolumns_amd.confusion_matrix import calculate_confusion_matrix
    from xcolumns_amd.metrics import binary_f1_score_on_conf_matrix
    from xcolumns_amd.weighted_prediction import predict_top_k
    n, m, r, k = 1_000_000, 500_000, 50, 5
    Y = _synthetic_csr(n, m, r, 20240009)
    P, meta = predict_optimizing_macro_f1_score_using_bc(Y, k, seed=13, max_iters=4, tolerance=-1.0, return_meta=True)
    ids = P.indices.reshape(n, k)
    assert (np.diff(ids, axis=1) > 0).all()
    sup = Y.indices.reshape(n, r)
    for i in range(0, n, 49999):                       # predicted labels are stored in their rows
        assert np.isin(ids[i], sup[i]).all()
    C = calculate_confusion_matrix(Y, P, normalize=True, skip_tn=True, dtype=np.float64)
    f1 = binary_f1_score_on_conf_matrix(C.tp, C.fp, C.fn, C.tn).mean()
    assert abs(f1 - meta["utilities"][-1]) < 1e-10
    T = predict_top_k(Y, k)
    Ct = calculate_confusion_matrix(Y, T, normalize=True, skip_tn=True, dtype=np.float64)
    f1_top = binary_f1_score_on_conf_matrix(Ct.tp, Ct.fp, Ct.fn, Ct.tn).mean()
    u = np.asarray(meta["utilities"])
    assert u[0] > f1_top and (np.diff(u) > -1e-6).all(), (f1_top, u)
    # top-k property at full size: every chosen score >= every unchosen score of its row
    chosen = np.zeros((n, r), dtype=bool)
    tid = T.indices.reshape(n, k)
    for q in range(k):
        chosen |= sup == tid[:, q:q + 1]
    assert (chosen.sum(axis=1) == k).all()
    data = Y.data.reshape(n, r)
    assert (np.where(chosen, data, np.inf).min(axis=1) >= np.where(~chosen, data, -np.inf).max(axis=1)).all()


# ---------------------------------------------------------------------------
# edges: empty input, maximum row length, limits
# ---------------------------------------------------------------------------

def test_edges_and_limits(oref):
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc
    from xcolumns_amd.weighted_prediction import predict_top_k, predict_weighted_per_instance
    # no rows / rows without entries
    E = csr_matrix((0, 7), dtype=np.float32)
    P = predict_top_k(E, 3)
    assert P.shape == (0, 7) and P.nnz == 0
    Z = csr_matrix((5, 7), dtype=np.float32)       # five empty rows: the reference's padding (column 0, value 1)
    _same_csr(predict_top_k(Z, 2), oref.predict_top_k(Z, 2))
    # maximum row length handled in registers: exactly 1024 entries
    rng = np.random.default_rng(0)
    m = 4096
    rows = [np.sort(rng.choice(m, l, replace=False)) for l in (1024, 1023, 5, 1024)]
    indptr = np.concatenate([[0], np.cumsum([r.size for r in rows])]).astype(np.int32)
    Y = csr_matrix((rng.random(indptr[-1]).astype(np.float32), np.concatenate(rows).astype(np.int32), indptr), shape=(4, m))
    _same_csr(predict_top_k(Y, 5), oref.predict_top_k(Y, 5))
    metric = oref.make_metric(oref.FBETA, k=5.0, m=float(m))
    Po, mo = oref.predict_using_bc_with_0approx(Y, metric, 5, skip_tn=True, seed=1, max_iters=2, tolerance=-1.0)
    Pg, mg = predict_optimizing_macro_f1_score_using_bc(Y, 5, seed=1, max_iters=2, tolerance=-1.0, return_meta=True, bca_waves=1)
    assert np.allclose(mg["utilities"], mo["utilities"], rtol=0, atol=1e-12) and np.array_equal(Pg.indices, Po.indices)
    # one entry more than the limit: a clear error, no silent truncation
    big = np.sort(rng.choice(m, 1025, replace=False)).astype(np.int32)
    B = csr_matrix((rng.random(1025).astype(np.float32), big, np.array([0, 1025], dtype=np.int32)), shape=(1, m))
    with pytest.raises(ValueError, match="1025"):
        predict_top_k(B, 5)
    with pytest.raises(ValueError):
        predict_weighted_per_instance(Y, 65)                      # k above XC_MAX_K
    with pytest.raises(ValueError, match="at least k"):
        predict_optimizing_macro_f1_score_using_bc(Y, 6)          # a row stores only 5 entries
    with pytest.raises(NotImplementedError):
        predict_optimizing_macro_f1_score_using_bc(Y, 0)          # CSR k = 0
    # int64 index arrays are accepted; the result carries the input's index dtype
    # (scipy itself narrows to int32 when the values fit)
    Y64 = csr_matrix((Y.data, Y.indices.astype(np.int64), Y.indptr.astype(np.int64)), shape=Y.shape)
    P64 = predict_top_k(Y64, 5)
    assert P64.indices.dtype == Y64.indices.dtype and P64.indptr.dtype == Y64.indptr.dtype
    assert np.array_equal(P64.indices, predict_top_k(Y, 5).indices)


def test_evaluation_metrics_golden():
    """macro / micro / instance metrics on true labels against values the
    reference itself produced; statistics come from the GPU confusion kernels."""
    import xcolumns_amd.metrics as pm
    z = G.load("eval_f64")
    L, P = G.csr_from(z, "l"), G.csr_from(z, "p")
    for name in [str(s) for s in z["names"]]:
        f = getattr(pm, name)
        assert abs(f(L, P) - float(z["csr_" + name])) < 1e-12, name
        assert abs(f(z["ld"], z["pd"]) - float(z["dense_" + name])) < 1e-12, name
    assert np.allclose(pm.label_priors(z["ld"]), z["label_priors"], rtol=0, atol=1e-15)
    # the rest of the evaluation set: accuracy / Hamming, precision@k (plain and propensity-weighted),
    # coverage / abandonment, tail variants, the propensity model, the at-k check
    k, priors, inv_ps = int(z["x_k"]), z["x_priors"], z["x_jpv_inverse_propensities"]
    kwargs = {"precision_at_k": dict(k=k), "weighted_precision_at_k": dict(k=k, w=inv_ps),
              "tail_abandonment": dict(priors=priors), "tail_recall": dict(priors=priors, percentile=0.3)}
    for name in [str(s) for s in z["extra_names"]]:
        f = getattr(pm, name)
        kw = kwargs.get(name, {})
        np.testing.assert_allclose(np.asarray(f(L, P, **kw), dtype=np.float64), z["xcsr_" + name], rtol=0, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(np.asarray(f(z["ld"], z["pd"], **kw), dtype=np.float64), z["xdense_" + name], rtol=0,
                                   atol=1e-12, err_msg=name)
    assert abs(pm.instance_tail_recall_at_k(L, P, k, priors, percentile=0.4) - float(z["xcsr_instance_tail_recall_at_k"])) < 1e-12
    assert abs(pm.instance_tail_metric(L, P, pm.binary_precision_on_conf_matrix, k, priors, percentile=0.6)
               - float(z["xcsr_instance_tail_metric"])) < 1e-12
    np.testing.assert_allclose(np.asarray(pm.jpv_inverse_propensities(z["ld"])).ravel(), inv_ps, rtol=1e-15)
    np.testing.assert_allclose(np.asarray(pm.jpv_inverse_propensities(L)).ravel(), inv_ps, rtol=1e-15)
    np.testing.assert_allclose(np.asarray(pm.jpv_propensities(z["ld"], a=0.6, b=2.6)).ravel(), z["x_jpv_propensities"], rtol=1e-15)
    got = [pm.check_if_y_pred_at_k(P, k), pm.check_if_y_pred_at_k(P, k + 1), pm.check_if_y_pred_at_k(z["pd"], k),
           pm.check_if_y_pred_at_k(L, k)]
    assert got == [bool(x) for x in z["x_check_at_k"]]


# ---------------------------------------------------------------------------
# the sweep loop with the stopping rule on the GPU (xc_bca_plan_*_pipelined)
# ---------------------------------------------------------------------------

def test_pipelined_loop_matches_the_stopping_rule_and_policy(oref):
    """Default (concurrent) runs hand the stopping rule and the wavefront policy to the GPU and read
    the results one iteration late.  The trace must be what the host-paced loop would produce: every
    sweep but the last improved by >= tolerance, the last did not (or max_iters was hit); the
    wavefronts of sweep j follow WavePolicy from the rows changed in sweep j - 1; the last utility is
    the utility of the returned prediction; nothing ran after the rule fired."""
    from xcolumns_amd.block_coordinate import WavePolicy, predict_optimizing_macro_f1_score_using_bc
    from xcolumns_amd.synthetic import make_csr

    n, m, k = 30000, 3000, 5
    Y = make_csr(n, m, 40, seed=77, k=k)
    for tol, max_iters in ((1e-5, 50), (-1.0, 4), (1e-6, 100)):
        P, meta = predict_optimizing_macro_f1_score_using_bc(Y, k, seed=3, tolerance=tol, max_iters=max_iters,
                                                             return_meta=True, bca_diagnostics=True)
        u = meta["utilities"]
        assert meta["iters"] == len(u) <= max_iters
        top = oref.predict_top_k(Y, k)
        metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))

        def util(pred):
            tp, fp, fn, tn = oref.calculate_confusion_matrix(Y, pred, skip_tn=True)
            return oref.calculate_utility(metric, "mean", tp / n, fp / n, fn / n, tn / n)

        trace = [util(top)] + u
        gains = np.diff(trace)
        assert (gains[:-1] >= tol).all(), gains
        assert gains[-1] < tol or len(u) == max_iters
        assert abs(util(P) - u[-1]) < 1e-12           # the prediction returned is the one of the last boundary
        pol = WavePolicy(n, m=m, row_nnz=Y.nnz / n)
        want = [pol.next(None)] + [pol.next(c) for c in meta["rows_changed"][:-1]]
        assert meta["wavefronts"] == want, (meta["wavefronts"], want)


def test_pipelined_and_host_paced_loops_agree(monkeypatch, oref):
    """XCOLUMNS_BCA_PIPELINE=0 keeps the host in the loop; both loops must stop after the same sweep
    with utilities that differ only by the concurrent sweep's run-to-run noise."""
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc
    from xcolumns_amd.synthetic import make_csr

    Y = make_csr(20000, 2000, 40, seed=78, k=5)
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("XCOLUMNS_BCA_PIPELINE", mode)
        _, runs[mode] = predict_optimizing_macro_f1_score_using_bc(Y, 5, seed=3, tolerance=-1.0, max_iters=6,
                                                                   bca_waves=64, return_meta=True,
                                                                   bca_diagnostics=True)
    a, b = runs["1"]["utilities"], runs["0"]["utilities"]
    assert len(a) == len(b) == 6
    assert np.abs(np.asarray(a) - np.asarray(b)).max() < PER_SWEEP_TOL
    assert abs(a[-1] - b[-1]) < FINAL_TOL
    assert runs["1"]["wavefronts"] == runs["0"]["wavefronts"] == [64] * 6


# ---------------------------------------------------------------------------
# dense BCA, concurrent mode (the product default) against the oracle
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("dtype,k,skip_tn,entry", [
    (np.float32, 5, True, "predict_optimizing_macro_f1_score_using_bc"),
    (np.float64, 3, False, "predict_optimizing_macro_balanced_accuracy_using_bc"),
    (np.float32, 0, True, "predict_optimizing_macro_f1_score_using_bc"),
])
def test_bca_dense_concurrent_vs_oracle(oref, dtype, k, skip_tn, entry):
    """Dense y_proba with the default policy (several workgroups walk the order): final utility within
    1e-5 of the sequential oracle after the same number of sweeps, every sweep within PER_SWEEP_TOL;
    bca_waves=1 reproduces it to 1e-12."""
    import xcolumns_amd.block_coordinate as bc
    rng = np.random.default_rng(21)
    n, m = 6000, 1500
    Y = (rng.random((n, m)) ** 6).astype(dtype)
    base = {"predict_optimizing_macro_f1_score_using_bc": oref.FBETA,
            "predict_optimizing_macro_balanced_accuracy_using_bc": oref.BALANCED_ACC}[entry]
    metric = oref.make_metric(base, k=float(max(k, 1)), m=float(m))
    Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=skip_tn, seed=13, max_iters=4, tolerance=-1.0)
    Pg, mg = getattr(bc, entry)(Y, k, seed=13, max_iters=4, tolerance=-1.0, return_meta=True, bca_diagnostics=True)
    assert mg["iters"] == 4 and type(Pg) == type(Y) and Pg.dtype == Y.dtype and Pg.shape == Y.shape
    diff = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
    print("dense concurrent-vs-sequential utility diff per sweep:", diff)
    if k == 0:   # no budget: the product keeps the sequential sweep
        assert diff.max() < 1e-12
    assert diff[-1] < FINAL_TOL and diff.max() < PER_SWEEP_TOL, (mg["utilities"], mo["utilities"])
    if k > 0:
        assert (Pg.sum(axis=1) == k).all()
    assert np.isin(Pg, (0, 1)).all()
    # the reported utility is the utility of the returned prediction
    tp, fp, fn, tn = oref.calculate_confusion_matrix(Y, Pg, skip_tn=skip_tn)
    u = oref.calculate_utility(metric, "mean", tp / n, fp / n, fn / n, tn / n)
    assert abs(u - mg["utilities"][-1]) < 1e-12
    Pe, me = getattr(bc, entry)(Y, k, seed=13, max_iters=4, tolerance=-1.0, return_meta=True, bca_waves=1)
    assert np.allclose(me["utilities"], mo["utilities"], rtol=0, atol=1e-12) and np.array_equal(Pe, Po)


@pytest.mark.parametrize("init", ["greedy", "random", "matrix_with_foreign_labels"])
def test_pipelined_loop_after_a_host_paced_first_sweep(oref, init):
    """Greedy first sweeps and initial predictions holding labels a row does not store cannot use the
    device-side loop for sweep 1; the loop switches over from sweep 2.  Same bars as the other
    concurrent runs against the sequential oracle."""
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc
    from xcolumns_amd.synthetic import make_csr

    n, m, k = 20000, 2000, 4
    Y = make_csr(n, m, 30, seed=91, k=k)
    kw = dict(seed=13, max_iters=5, tolerance=-1.0)
    if init == "matrix_with_foreign_labels":
        rng = np.random.default_rng(5)
        idx = np.sort(np.stack([rng.choice(m, k, replace=False) for _ in range(n)]), axis=1).astype(np.int32)
        mk = lambda: csr_matrix((np.ones(n * k, dtype=np.float32), idx.ravel().copy(), np.arange(n + 1, dtype=np.int32) * k),
                                shape=(n, m))
        init_o, init_g = mk(), mk()
    else:
        init_o = init_g = init
    metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
    Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, init_y_pred=init_o, **kw)
    Pg, mg = predict_optimizing_macro_f1_score_using_bc(Y, k, init_y_pred=init_g, return_meta=True, **kw)
    assert mg["iters"] == mo["iters"] == 5
    diff = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
    print(init, "concurrent-vs-sequential utility diff per sweep:", diff)
    assert diff[-1] < FINAL_TOL and diff.max() < PER_SWEEP_TOL, (mg["utilities"], mo["utilities"])
    tp, fp, fn, tn = oref.calculate_confusion_matrix(Y, Pg, skip_tn=True)
    assert abs(oref.calculate_utility(metric, "mean", tp / n, fp / n, fn / n, tn / n) - mg["utilities"][-1]) < 1e-12
    if init == "matrix_with_foreign_labels":
        assert Pg is init_g   # updated in place and returned, like the reference


@pytest.mark.parametrize("k", [33, 64])
def test_bca_csr_large_budget(oref, k):
    """Budgets above 32 labels per row (XC_MAX_K = 64): the from-scratch statistics a sweep accumulates
    carry 32 predicted entries per deferred wave instruction, the rest separately.  Exact mode reproduces
    the oracle; the concurrent default stays within the bars and reports the utility of what it returns."""
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc
    rng = np.random.default_rng(k)
    n, m, r = 6000, 900, 100
    cols = np.concatenate([np.sort(rng.choice(m, r, replace=False)) for _ in range(n)]).astype(np.int32)
    Y = csr_matrix(((rng.random(n * r) ** 3).astype(np.float32), cols, (np.arange(n + 1) * r).astype(np.int32)), shape=(n, m))
    metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
    Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=2, max_iters=8, tolerance=-1.0)
    Pe, me = predict_optimizing_macro_f1_score_using_bc(Y, k, seed=2, max_iters=8, tolerance=-1.0, return_meta=True, bca_waves=1)
    assert np.allclose(me["utilities"], mo["utilities"], rtol=0, atol=1e-12) and np.array_equal(Pe.indices, Po.indices)
    Pg, mg = predict_optimizing_macro_f1_score_using_bc(Y, k, seed=2, max_iters=8, tolerance=-1.0, return_meta=True)
    diff = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
    assert diff[-1] < FINAL_TOL and diff.max() < PER_SWEEP_TOL, (mg["utilities"], mo["utilities"])
    tp, fp, fn, tn = oref.calculate_confusion_matrix(Y, Pg, skip_tn=True)
    assert abs(oref.calculate_utility(metric, "mean", tp / n, fp / n, fn / n, tn / n) - mg["utilities"][-1]) < 1e-12


def _ragged_problem(rng, n, m, rmin, rmax, dtype, zipf):
    lens = rng.integers(rmin, rmax + 1, size=n)
    if zipf:
        w = 1.0 / np.arange(1, m + 1)
        w /= w.sum()
        perm = rng.permutation(m)
        cols = [np.sort(perm[rng.choice(m, l, replace=False, p=w)]) for l in lens]
    else:
        cols = [np.sort(rng.choice(m, l, replace=False)) for l in lens]
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    data = (rng.random(indptr[-1]) ** 3).astype(dtype)
    return csr_matrix((data, np.concatenate(cols).astype(np.int32), indptr), shape=(n, m))


@pytest.mark.parametrize("case", ["f64_scores", "minimize", "long_rows_zipf", "rows_of_exactly_k", "tiny", "balanced_accuracy_tn"])
def test_bca_csr_concurrent_corner_cases(oref, case):
    """The concurrent default against the sequential oracle on inputs that take the less travelled
    kernel variants: float64 scores (no packed stream), minimisation, rows of 65..300 entries with a
    skewed popularity (several candidates per lane + hot labels), rows that hold exactly k entries,
    tiny matrices, a metric that needs tn."""
    import xcolumns_amd.block_coordinate as bc
    rng = np.random.default_rng(len(case))
    k, kw, entry, base, skip_tn = 5, {}, "predict_optimizing_macro_f1_score_using_bc", oref.FBETA, True
    tol = -1.0
    if case == "f64_scores":
        Y = _ragged_problem(rng, 20000, 2500, 20, 60, np.float64, False)
    elif case == "minimize":
        Y = _ragged_problem(rng, 20000, 2500, 20, 60, np.float32, False)
        entry, kw, tol = None, dict(maximize=False), 1.0   # minimising: (new - old > tolerance) never fires
    elif case == "long_rows_zipf":
        Y = _ragged_problem(rng, 6000, 4000, 65, 300, np.float32, True)
    elif case == "rows_of_exactly_k":
        Y = _ragged_problem(rng, 5000, 300, 5, 5, np.float32, False)
    elif case == "tiny":
        Y = _ragged_problem(rng, 3, 7, 5, 6, np.float32, False)
    else:
        Y = _ragged_problem(rng, 20000, 2500, 20, 60, np.float32, False)
        entry, base, skip_tn = "predict_optimizing_macro_balanced_accuracy_using_bc", oref.BALANCED_ACC, False
    n, m = Y.shape
    metric = oref.make_metric(base, k=float(k), m=float(m))
    Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=skip_tn, seed=7, max_iters=4, tolerance=tol, **kw)
    if entry is None:
        Pg, mg = bc.predict_using_bc_with_0approx(Y, bc.binary_f1_score_on_conf_matrix, k, skip_tn=True, seed=7,
                                                  max_iters=4, tolerance=tol, return_meta=True, **kw)
    else:
        Pg, mg = getattr(bc, entry)(Y, k, seed=7, max_iters=4, tolerance=tol, return_meta=True)
    diff = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
    print(case, "concurrent-vs-sequential utility diff per sweep:", diff)
    assert mg["iters"] == mo["iters"] == 4
    _parity.check(diff, case)
    assert Pg.dtype == Y.dtype and (np.diff(Pg.indptr) == k).all()
    tp, fp, fn, tn = oref.calculate_confusion_matrix(Y, Pg, skip_tn=skip_tn)
    assert abs(oref.calculate_utility(metric, "mean", tp / n, fp / n, fn / n, tn / n) - mg["utilities"][-1]) < 1e-12
    if case == "rows_of_exactly_k":
        assert np.array_equal(Pg.indices, Y.indices)


# ---------------------------------------------------------------------------
# coverage BCA
# ---------------------------------------------------------------------------

def test_coverage_bca_golden_exact():
    """predict_optimizing_coverage_using_bc with bca_waves=1 against the reference's own outputs:
    identical predictions and iteration counts, utilities to 1e-12 (1e-6 where the reference sums the
    precision@k part in float32)."""
    from xcolumns_amd.block_coordinate import predict_optimizing_coverage_using_bc
    z = G.load("coverage")
    for ci, spec in enumerate(G.coverage_cases(z)):
        Y, kw = G.coverage_inputs(z, ci, spec)
        init = kw.get("init_y_pred")
        P, meta = predict_optimizing_coverage_using_bc(Y, spec["k"], return_meta=True, bca_waves=1, **kw)
        name = spec["name"]
        assert meta["iters"] == int(z[f"c{ci}_iters"]), name
        tol = 1e-6 if (spec["dtype"] == "f32" and kw.get("alpha", 1) < 1) else 1e-12
        np.testing.assert_allclose(meta["utilities"], z[f"c{ci}_utilities"], rtol=0, atol=tol, err_msg=name)
        np.testing.assert_array_equal(P.indices, z[f"c{ci}_pred_indices"], err_msg=name)
        assert isinstance(P, csr_matrix) and P.dtype == Y.dtype and (np.diff(P.indptr) == spec["k"]).all()
        assert set(meta) == {"utilities", "iters", "time"}
        if spec["explicit_init"]:
            assert P is init   # updated in place and returned


@pytest.mark.parametrize("alpha", [1, 0.6])
def test_coverage_bca_vs_oracle(alpha):
    """Default (sequential, one wavefront) = the oracle to 1e-12 at 20K x 30K; bca_waves = 16 (rows in
    flight, compare-and-swap multiplies) lands within 1e-4 -- the coverage landscape has many nearly
    equivalent optima, so trajectories that differ at all end a few 1e-5 apart."""
    from oracle import coverage_ref as cov
    from xcolumns_amd.block_coordinate import predict_optimizing_coverage_using_bc
    from xcolumns_amd.synthetic import make_csr
    n, m, k = 20000, 30000, 3
    Y = make_csr(n, m, 20, seed=31, k=k, zipf=True)
    Po, mo = cov.predict_optimizing_coverage_using_bc(Y, k, alpha=alpha, seed=4, max_iters=5, tolerance=-1.0)
    Pe, me = predict_optimizing_coverage_using_bc(Y, k, alpha=alpha, seed=4, max_iters=5, tolerance=-1.0, return_meta=True)
    tol = 1e-12 if alpha == 1 else 1e-6   # float32 scores: the precision@k part is summed in float64 here
    np.testing.assert_allclose(me["utilities"], mo["utilities"], rtol=0, atol=tol)
    np.testing.assert_array_equal(Pe.indices, Po.indices)
    Pg, mg = predict_optimizing_coverage_using_bc(Y, k, alpha=alpha, seed=4, max_iters=5, tolerance=-1.0, return_meta=True,
                                                  bca_waves=16)
    diff = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
    print("coverage, 16 wavefronts vs sequential, utility diff per sweep:", diff)
    assert mg["iters"] == mo["iters"] == 5 and diff.max() < 1e-4, (mg["utilities"], mo["utilities"])
    assert (np.diff(Pg.indptr) == k).all() and Pg.dtype == Y.dtype
    # the reported utility is the utility of the returned prediction
    Ef = cov.failure_probabilities(Y, Pg.indices, k)
    assert abs(cov.coverage_utility(Y, Pg.indices, Ef, k, alpha) - mg["utilities"][-1]) < 1e-6
    with pytest.raises(NotImplementedError):
        predict_optimizing_coverage_using_bc(Y[:10].toarray(), k)
    with pytest.raises(ValueError):
        predict_optimizing_coverage_using_bc(Y, 0)


def test_io_dense_scores_to_topk_csr_and_device(tmp_path):
    """load_npy_full_pred (experiments/utils.py:198-210): the k largest scores of every row of a dense .npy
    matrix as a CSR matrix with sorted column ids -- selected on the GPU -- and its upload for the kernels."""
    from xcolumns_amd import io as xio
    rng = np.random.default_rng(12)
    dense = rng.random((300, 500)).astype(np.float32)
    path = str(tmp_path / "scores.npy")
    np.save(path, dense)
    k = 7
    mat = xio.load_npy_full_pred(path, keep_top_k=k)
    assert mat.shape == (300, 500) and mat.dtype == np.float32 and (np.diff(mat.indptr) == k).all()
    top = np.sort(np.argsort(-dense, axis=1)[:, :k], axis=1)
    assert np.array_equal(mat.indices.reshape(300, k), top)
    assert np.array_equal(mat.data.reshape(300, k), np.take_along_axis(dense, top, axis=1))
    assert xio.load_npy_full_pred(path).nnz == 0   # keep_top_k = 0: nothing kept
    d = xio.to_device(mat)
    assert d.indices.is_cuda and d.indices.dtype == torch.int32 and d.max_row_nnz == k


def test_block_coordinate_reference_test_properties():
    """The reference's own BCA test (tests/test_block_coordinate.py:33-96), on synthetic data: for every
    input type the prediction keeps type / dtype / k labels per row; BCA for macro recall scores at least
    as well as top-k on the true labels and lands within 0.02 of the closed-form optimum
    (predict_optimizing_macro_recall with the label priors)."""
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_recall_using_bc
    from xcolumns_amd.confusion_matrix import calculate_confusion_matrix
    from xcolumns_amd.metrics import macro_recall_on_conf_matrix
    from xcolumns_amd.weighted_prediction import predict_optimizing_macro_recall, predict_top_k
    rng = np.random.default_rng(17)
    n, m, k = 6000, 400, 3
    w = 0.03 + 0.97 * rng.random(m) ** 3                       # skewed label priors
    dense_proba = ((rng.random((n, m)) ** 4) * w).astype(np.float32)
    dense_true = (rng.random((n, m)) < dense_proba).astype(np.float32)
    scores = {}
    for kind, conv in (("numpy", lambda x: x), ("csr", lambda x: csr_matrix(x)), ("torch", lambda x: torch.from_numpy(x).cuda())):
        y_proba, y_true = conv(dense_proba), conv(dense_true)
        y_pred, meta = predict_optimizing_macro_recall_using_bc(y_proba, k, return_meta=True, seed=2024)
        assert type(y_pred) == type(y_proba) and y_pred.dtype == y_proba.dtype
        rows = y_pred.sum(axis=1) if kind != "torch" else y_pred.sum(dim=1).cpu().numpy()
        assert (np.asarray(rows).ravel() == k).all()
        C = calculate_confusion_matrix(y_true, y_pred, normalize=False, skip_tn=False)
        scores[kind] = float(macro_recall_on_conf_matrix(*C))
    assert abs(scores["numpy"] - scores["torch"]) < 1e-6 and abs(scores["numpy"] - scores["csr"]) < 5e-3
    top = float(macro_recall_on_conf_matrix(*calculate_confusion_matrix(dense_true, predict_top_k(dense_proba, k))))
    opt = float(macro_recall_on_conf_matrix(*calculate_confusion_matrix(
        dense_true, predict_optimizing_macro_recall(dense_proba, k, priors=dense_true.mean(axis=0)))))
    print(f"top-k {top:.4f}  BCA {scores}  closed form {opt:.4f}")
    assert scores["numpy"] >= top and abs(opt - scores["numpy"]) < 0.02


WRAPPER_CASES = {
    "macro_recall": ("predict_optimizing_macro_recall", ("priors",), {}),
    "macro_balanced_accuracy": ("predict_optimizing_macro_balanced_accuracy", ("priors",), {}),
    "log_weighted": ("predict_log_weighted_per_instance", ("priors",), {}),
    "power_law": ("predict_power_law_weighted_per_instance", ("priors",), {"beta": 0.5}),
    "instance_precision": ("predict_optimizing_instance_precision", (), {}),
    "ps_precision_inverse": ("predict_optimizing_instance_propensity_scored_precision", ("inverse_propensities",), {}),
    "ps_precision_propensities": ("predict_optimizing_instance_propensity_scored_precision", ("propensities",), {}),
}


@pytest.mark.parametrize("case", sorted(WRAPPER_CASES))
def test_weighted_prediction_wrappers_golden(case):
    """The wrappers of tests/test_weighted_prediction.py:69-103 against label sets the reference produced
    (tests/golden/wp_wrappers.npz), for CSR float32, dense float64 and torch inputs: bit-exact."""
    import xcolumns_amd.weighted_prediction as wp
    z = G.load("wp_wrappers")
    k = int(z["k"])
    Y = G.csr_from(z, "y")
    n, m = Y.shape
    name, vec_args, extra = WRAPPER_CASES[case]
    fn = getattr(wp, name)
    kw = lambda conv=(lambda v: v): {**{a: conv(z[a].copy()) for a in vec_args}, **extra}  # noqa: E731
    P = fn(Y, k, **kw())
    assert isinstance(P, csr_matrix) and P.dtype == np.float32 and (np.diff(P.indptr) == k).all()
    assert np.array_equal(P.indices.reshape(n, k), z["csr_" + case])
    Yd = Y.astype(np.float64).toarray()
    Pd = fn(Yd, k, **kw())
    assert isinstance(Pd, np.ndarray) and Pd.dtype == np.float64 and (Pd.sum(axis=1) == k).all()
    assert np.array_equal(np.sort(np.argsort(-Pd, axis=1, kind="stable")[:, :k], axis=1), z["dense_" + case])
    Pt = fn(torch.from_numpy(Yd).cuda(), k, **kw(lambda v: torch.from_numpy(v).cuda()))
    assert isinstance(Pt, torch.Tensor) and Pt.is_cuda and Pt.dtype == torch.float64
    assert np.array_equal(Pt.cpu().numpy(), Pd)


@pytest.mark.parametrize("switch", ["XCOLUMNS_BCA_PACKED", "XCOLUMNS_BCA_HOT", "XCOLUMNS_BCA_SHADOW"])
@pytest.mark.parametrize("waves", [1, None])
def test_bca_csr_kernel_variants_behind_the_switches(oref, monkeypatch, switch, waves):
    """The kernel variants a default run does not take for float32 scores -- the separate index / score
    streams (what m > 2^25 labels fall back to), no hot-label table, the float64 records instead of their
    float32 copy -- on skewed data (hot labels exist): with one wavefront each is the sequential oracle to
    1e-12, with the default concurrency it meets the same bars as the default variant."""
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc
    n, m, r, k = 20000, 3000, 30, 5
    Y = _synthetic_csr(n, m, r, 12, zipf=True)
    metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
    Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=3, max_iters=4, tolerance=-1.0)
    monkeypatch.setenv(switch, "0")
    Pg, mg = predict_optimizing_macro_f1_score_using_bc(Y, k, seed=3, max_iters=4, tolerance=-1.0, return_meta=True,
                                                        bca_waves=waves)
    diff = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
    print(switch, waves, diff)
    if waves == 1:
        assert diff.max() < 1e-12 and np.array_equal(Pg.indices, Po.indices)
    else:
        assert diff[-1] < FINAL_TOL and diff.max() < PER_SWEEP_TOL


def test_bca_csr_label_space_beyond_the_packed_stream(oref):
    """m = 40 M labels (> 2^25, the packed 16-byte entries hold 25-bit column ids): the sweep falls back to the
    separate index / score streams by itself; record offsets stay within 32 bits.  Sequential = the oracle
    to 1e-12, default concurrency within the bars."""
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc
    n, m, r, k = 3000, 40_000_000, 24, 3
    rng = np.random.default_rng(77)
    cols = np.sort(rng.integers(0, m, size=(n, r), dtype=np.int64), axis=1)
    cols[:, -1] = m - 1 - rng.integers(0, 50, size=n)          # the far end of the tables is touched
    cols = np.sort(cols, axis=1)
    assert (np.diff(cols, axis=1) > 0).all()
    Y = csr_matrix(((rng.random(n * r) ** 2).astype(np.float32).clip(1e-3, 1), cols.ravel().astype(np.int32),
                    (np.arange(n + 1) * r).astype(np.int32)), shape=(n, m))
    metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
    Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=5, max_iters=2, tolerance=-1.0)
    for waves in (1, None):
        Pg, mg = predict_optimizing_macro_f1_score_using_bc(Y, k, seed=5, max_iters=2, tolerance=-1.0,
                                                            return_meta=True, bca_waves=waves)
        diff = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
        if waves == 1:
            assert diff.max() < 1e-12 and np.array_equal(Pg.indices, Po.indices)
        else:
            assert diff[-1] < FINAL_TOL and diff.max() < PER_SWEEP_TOL


def test_device_side_loop_pauses_for_exact_sweeps(oref, monkeypatch):
    """The device-side loop hands a sweep back to the host when its wavefront rule falls below the policy's
    exact-sweep threshold (XC_CTRL_EXACT_BELOW, flag 3): the sweep that was already enqueued does nothing, the host runs
    it with the SAME visiting order, and the loop is armed again afterwards.  Forced here after every sweep: every
    order is consumed exactly once, in sequence, and the trace stays the sequential oracle's (the host's sweeps are the
    exact ones, the device's run 64 wavefronts on 30 K rows)."""
    import xcolumns_amd.block_coordinate as bc
    n, m, k, sweeps = 30000, 3000, 5, 6
    Y = _synthetic_csr(n, m, 30, 77)
    calls = {"host": 0, "device": 0}

    class AlternatingPolicy(bc.WavePolicy):
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            self.sequential_below = 10 ** 9          # the device rule always pauses after the sweep it ran
            self._turn = 0

        def next(self, changed_prev=None, greedy=False):
            # the host's question "how wide is the next sweep": concurrent (64) and exact (1) alternately
            return 64

    seen = []
    orig_step, orig_sweep = bc.BcaCsrEngine.pipeline_step, bc.BcaCsrEngine.sweep

    def spy_step(self, order, j, n_u):
        seen.append(("device", j, order.cpu().numpy().copy()))
        return orig_step(self, order, j, n_u)

    def spy_sweep(self, order, n_order, n_waves, greedy=False):
        seen.append(("host", None, order.cpu().numpy().copy()))
        return orig_sweep(self, order, n_order, n_waves, greedy=greedy)

    monkeypatch.setattr(bc, "WavePolicy", AlternatingPolicy)
    monkeypatch.setattr(bc.BcaCsrEngine, "pipeline_step", spy_step)
    monkeypatch.setattr(bc.BcaCsrEngine, "sweep", spy_sweep)
    Pg, mg = bc.predict_optimizing_macro_f1_score_using_bc(Y, k, seed=13, max_iters=sweeps, tolerance=-1.0, return_meta=True)
    metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
    _, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=13, max_iters=sweeps, tolerance=-1.0)
    assert mg["iters"] == sweeps and len(mg["utilities"]) == sweeps
    d = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
    print("paused after every sweep, |utility - oracle| per sweep:", d)
    assert d.max() < PER_SWEEP_TOL
    # the sweeps that ran: device sweeps j = 1 .. K, each followed by a no-op'd enqueue of j + 1 that is re-issued
    ran = [s for s in seen if s[0] == "device"]
    rng, order = np.random.default_rng(13), np.arange(n)
    expected = []
    for _ in range(sweeps):
        rng.shuffle(order)
        expected.append(order.astype(np.int32).copy())
    by_j = {}
    for _, j, o in ran:
        by_j.setdefault(j, []).append(o)
    assert sorted(by_j) == list(range(1, sweeps + 1)) or sorted(by_j) == list(range(1, sweeps + 2))
    for j in range(1, sweeps + 1):
        for o in by_j[j]:                    # enqueued (and paused), then re-issued: the same order both times
            assert np.array_equal(o, expected[j - 1]), j

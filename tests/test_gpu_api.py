"""-m gpu tests of the round-2 API surface: sparse inputs already resident in HBM (DeviceCSR, torch sparse_csr),
the visiting-order worker, RCCL through TorchComm, and bench.py starting its own ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(n=30000, m=4000, r=40, k=5, seed=3):
    from xcolumns_amd.synthetic import make_csr
    return make_csr(n, m, r, seed=seed, k=k), k


def test_device_resident_inputs_top_k_and_confusion(oref):
    """predict_top_k / predict_weighted_per_instance / calculate_confusion_matrix on a DeviceCSR and on a torch
    sparse_csr tensor: same index sets as the host csr_matrix call (bit-exact vs the oracle), the result is the
    same kind of object and stays on the GPU."""
    from xcolumns_amd import DeviceCSR
    from xcolumns_amd.confusion_matrix import calculate_confusion_matrix
    from xcolumns_amd.weighted_prediction import predict_top_k, predict_weighted_per_instance

    Y, k = _problem()
    n, m = Y.shape
    want = oref.predict_top_k(Y, k)
    d = DeviceCSR.from_scipy(Y)
    Pd = predict_top_k(d, k)
    assert isinstance(Pd, DeviceCSR) and Pd.indices.is_cuda and Pd.shape == (n, m) and Pd.dtype == Y.dtype
    assert np.array_equal(Pd.indices.cpu().numpy(), want.indices)
    t = torch.sparse_csr_tensor(torch.from_numpy(Y.indptr.astype(np.int64)), torch.from_numpy(Y.indices.astype(np.int64)),
                                torch.from_numpy(Y.data), size=Y.shape).cuda()
    Pt = predict_top_k(t, k)
    assert isinstance(Pt, torch.Tensor) and Pt.layout == torch.sparse_csr and Pt.is_cuda and tuple(Pt.shape) == (n, m)
    assert Pt.col_indices().dtype == torch.int64 and np.array_equal(Pt.col_indices().cpu().numpy(), want.indices)
    # weights + keep_scores
    rng = np.random.default_rng(0)
    a, b = rng.random(m).astype(np.float32) + 0.5, rng.random(m).astype(np.float32) * 0.1
    Wh = predict_weighted_per_instance(Y, k, a=a, b=b, keep_scores=True)
    Wd = predict_weighted_per_instance(d, k, a=a, b=b, keep_scores=True)
    assert np.array_equal(Wd.indices.cpu().numpy(), Wh.indices) and np.array_equal(Wd.data.cpu().numpy(), Wh.data)
    # k = 0: threshold
    Th = predict_weighted_per_instance(Y, 0, th=0.3)
    Td = predict_weighted_per_instance(d, 0, th=0.3)
    assert np.array_equal(Td.indptr.cpu().numpy(), Th.indptr) and np.array_equal(Td.indices.cpu().numpy(), Th.indices)
    # confusion matrix of device-resident matrices: torch vectors on the GPU, equal to the host call
    Ch = calculate_confusion_matrix(Y, want, skip_tn=True, dtype=np.float64)
    Cd = calculate_confusion_matrix(d, Pd, skip_tn=True, dtype=np.float64)
    for h, g in zip(Ch, Cd):
        assert isinstance(g, torch.Tensor) and g.is_cuda
        np.testing.assert_allclose(g.cpu().numpy(), h, rtol=0, atol=1e-9)
    Ct = calculate_confusion_matrix(t, Pt, skip_tn=True, dtype=np.float64)
    np.testing.assert_allclose(Ct.tp.cpu().numpy(), Ch.tp, rtol=0, atol=1e-9)
    with pytest.raises(ValueError):
        calculate_confusion_matrix(d, want)          # mixing a device matrix with a host one is refused


def test_device_resident_inputs_bca(oref):
    """predict_optimizing_macro_f1_score_using_bc on DeviceCSR / torch sparse_csr: exact mode reproduces the
    oracle's prediction, the default mode stays within the bar, init_y_pred of the same kind is updated in place."""
    from xcolumns_amd import DeviceCSR
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f

    Y, k = _problem(20000, 3000, 30)
    n, m = Y.shape
    metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
    Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=5, max_iters=3, tolerance=-1.0)
    d = DeviceCSR.from_scipy(Y)
    Pe, me = f(d, k, seed=5, max_iters=3, tolerance=-1.0, return_meta=True, bca_waves=1)
    assert isinstance(Pe, DeviceCSR) and np.array_equal(Pe.indices.cpu().numpy(), Po.indices)
    assert np.allclose(me["utilities"], mo["utilities"], rtol=0, atol=1e-12)
    t = d.to_torch(torch.int64)
    Pt, mt = f(t, k, seed=5, max_iters=3, tolerance=-1.0, return_meta=True)
    assert Pt.layout == torch.sparse_csr and Pt.is_cuda and (Pt.crow_indices().diff() == k).all()
    assert np.abs(np.asarray(mt["utilities"]) - np.asarray(mo["utilities"])).max() < 1e-5
    init = f(d, k, seed=1, max_iters=1, tolerance=-1.0)
    out = f(d, k, seed=5, max_iters=2, tolerance=-1.0, init_y_pred=init)
    assert out is init
    with pytest.raises(ValueError):
        f(DeviceCSR.from_scipy(Y[:, :]), k + 100)     # rows shorter than k are refused before any launch
    with pytest.raises(ValueError):
        DeviceCSR.from_parts(d.indptr, d.indices + m, d.data, d.shape)   # column ids out of range


@pytest.mark.parametrize("where", ["gpu", "host_threads"])
def test_order_source_delivers_the_reference_stream(where, monkeypatch):
    """The visiting orders prepared ahead -- generated on the GPU (the default for large matrices), or walked by the
    host's worker threads into a pinned buffer and copied on a side stream (XCOLUMNS_ORDER_DEVICE=0) -- are numpy's."""
    from xcolumns_amd import _device as D
    from xcolumns_amd.block_coordinate import _OrderSource

    # (unset, the faster of the two on this machine is taken: block_coordinate._orders_on_device)
    monkeypatch.setenv("XCOLUMNS_ORDER_DEVICE", "0" if where == "host_threads" else "1")
    n = 120_000
    dev = D.require_gpu()
    src = _OrderSource(n, 13, True, "numpy", dev)
    rng = np.random.default_rng(13)
    ref = np.arange(n)
    try:
        assert src._threaded == (where == "host_threads") and (src._devgen is not None) == (where == "gpu")
        for _ in range(6):
            rng.shuffle(ref)
            got = src.next()
            torch.cuda.synchronize()
            assert got.dtype == torch.int32 and got.is_cuda and np.array_equal(got.cpu().numpy(), ref.astype(np.int32))
    finally:
        src.close()
    assert not src._threaded and src._devgen is None


def test_rccl_all_reduce_through_torchcomm():
    """backend "nccl" (= RCCL) with the one rank a one-GPU box allows: the collective the sharded sweep issues at
    every boundary -- an in-place float64 sum over 2m + 1 values on the compute stream -- runs through RCCL."""
    code = (
        "import os, torch, torch.distributed as dist\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29571')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "from xcolumns_amd.distributed import TorchComm\n"
        "c = TorchComm(); c.start_timing()\n"
        "t = torch.arange(1000001, dtype=torch.float64, device='cuda'); c.all_reduce(t); ms = c.stop_timing()\n"
        "assert dist.get_backend() == 'nccl' and float(t[-1]) == 1e6 and c.calls == 1 and c.bytes_reduced == 8 * 1000001\n"
        "print('rccl ok', ms)\n"
        "dist.destroy_process_group()\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` outside torch.distributed.run: the parent starts the ranks as children and relays
    ONE JSON line (rehearsed with gloo on the one GPU: RCCL refuses two ranks on one device)."""
    env = dict(os.environ, XC_BENCH_BACKEND="gloo", XC_BENCH_ONE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c2_100Kx30K",
                          "--steps", "3", "--warmup", "1", "--repeats", "2", "--no-cpu-baseline"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 3 and j["value"] > 0
    assert j["comm"]["world_size"] == 2 and j["comm"]["backend"] == "gloo" and j["comm"]["all_reduce_bytes_per_sweep"] > 0
    assert j["comm"]["world_size_seen"] == 2 and j["comm"]["ms_per_collective"] > 0 and j["comm"]["boundary_all_reduce_bytes"] == (2 * 30000 + 1) * 8
    assert j["config"]["rows_total"] == 200_000 and j["strong_scaling"]["rows_total"] == 100_000
    sc = j["strong_scaling"]["comm"]
    assert sc["world_size_seen"] == 2 and sc["ms_per_collective"] > 0
    sp = sc["sharded_vs_oracle"]     # the public sharded call on the split matrix against the sequential oracle (rank 0)
    print("bench sharded_vs_oracle:", sp)
    # two shards of 50 K rows exchange three times per sweep (a part is not cut below 16 K rows): the first sweeps of a
    # sharded run trail the sequential reference by about 1.8e-3 / exchanges at two predicted rows per label and close in
    # by 3-4x per sweep (profiles/r03_shard_gpu_study.txt: the GPU engine = the checker engine's figures)
    assert len(sp["abs_diff_by_sweep"]) == 2 and sp["abs_diff_by_sweep"][0] < 1e-3 and sp["abs_diff_by_sweep"][1] < 3e-4
    assert len(j["roofline"]["frac_by_sweep"]) == 3 and 0 < j["roofline"]["frac"] < 1


@pytest.mark.parametrize("zipf", [False, True])
def test_deterministic_mode_is_the_reference_sequence(oref, zipf):
    """bca_deterministic=True: every sweep is the ordered parallel sweep -- the reference's own sequence, which is
    deterministic given `seed` (block_coordinate.py:413-419): runs with the same seed return the same prediction,
    it is the sequential oracle's prediction, and the utilities are the oracle's to 1e-12 (the default mode differs
    in a few rows run to run).  Thousands of rows are in flight meanwhile (not one wavefront)."""
    from xcolumns_amd import block_coordinate as bc
    from xcolumns_amd.synthetic import make_csr

    f = bc.predict_optimizing_macro_f1_score_using_bc
    n, m, r, k = 60000, 20000, 40, 5
    Y = make_csr(n, m, r, seed=77, zipf=zipf, k=k)
    metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
    Po, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=5, max_iters=4, tolerance=-1.0)
    seen = []
    orig = bc.BcaCsrEngine.sweep_ordered

    def spy(self, order, n_order):
        orig(self, order, n_order)
        seen.append(dict(self.ordered_stats))

    bc.BcaCsrEngine.sweep_ordered = spy
    try:
        runs = [f(Y, k, seed=5, max_iters=4, tolerance=-1.0, return_meta=True, bca_deterministic=True) for _ in range(3)]
    finally:
        bc.BcaCsrEngine.sweep_ordered = orig
    assert len(seen) == 12 and all(s_["error"] == 0 and s_["window"] >= 4096 for s_ in seen)
    P0, m0 = runs[0]
    for P, mt in runs[1:]:
        assert np.array_equal(P.indices, P0.indices)
        assert np.allclose(mt["utilities"], m0["utilities"], rtol=0, atol=1e-14)
    assert np.array_equal(P0.indices, Po.indices)
    d = np.abs(np.asarray(m0["utilities"]) - np.asarray(mo["utilities"]))
    print("deterministic mode: windows of", seen[0]["window"], "rows,", [s_["iterations"] for s_ in seen[:4]],
          "iterations per sweep; |utility - oracle| per sweep:", d)
    assert d.max() < 1e-12
    assert (np.diff(P0.indptr) == k).all() and (np.diff(P0.indices.reshape(n, k), axis=1) > 0).all()


@pytest.mark.parametrize("delta", ["1", "0"])
@pytest.mark.parametrize("zipf", [False, True])
def test_boundary_statistics_are_those_of_the_prediction(oref, zipf, delta, monkeypatch):
    """The reference recomputes the confusion matrix from scratch at every sweep boundary (block_coordinate.py:465-467).
    The pipelined sweeps either rebuild it in `acc` (XCOLUMNS_BCA_ACC_DELTA=0) or push every committed change into
    the float64 records (default; hot labels through a float64 LDS table): both ways the utility reported after
    the last sweep is the utility of the returned prediction computed from scratch, and every sweep holds the bar."""
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f
    from xcolumns_amd.synthetic import make_csr

    monkeypatch.setenv("XCOLUMNS_BCA_ACC_DELTA", delta)
    n, m, k = 100_000, 30_000, 5
    Y = make_csr(n, m, 50, seed=20240001, zipf=zipf, k=k)
    metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
    _, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=13, max_iters=6, tolerance=-1.0)
    P, mg = f(Y, k, seed=13, max_iters=6, tolerance=-1.0, return_meta=True)
    tp, fp, fn, tn = oref.calculate_confusion_matrix(Y, P, skip_tn=True)
    u = oref.calculate_utility(metric, "mean", tp / n, fp / n, fn / n, tn / n)
    d = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
    print("acc_delta", delta, "zipf", zipf, "|reported - from scratch|", abs(u - mg["utilities"][-1]), "per sweep vs oracle", d)
    assert abs(u - mg["utilities"][-1]) < 1e-11
    assert d.max() < 1e-5
    if not zipf:
        # float64 scores (no packed stream; the reference's own dtype for its statistics), longer budget, shorter run
        Y64 = Y[:40_000].astype(np.float64)
        n64, k64 = Y64.shape[0], 8
        metric64 = oref.make_metric(oref.FBETA, k=float(k64), m=float(m))
        _, mo64 = oref.predict_using_bc_with_0approx(Y64, metric64, k64, skip_tn=True, seed=3, max_iters=4, tolerance=-1.0)
        P64, mg64 = f(Y64, k64, seed=3, max_iters=4, tolerance=-1.0, return_meta=True)
        tp, fp, fn, tn = oref.calculate_confusion_matrix(Y64, P64, skip_tn=True)
        u64 = oref.calculate_utility(metric64, "mean", tp / n64, fp / n64, fn / n64, tn / n64)
        d64 = np.abs(np.asarray(mg64["utilities"]) - np.asarray(mo64["utilities"]))
        print("  float64 scores, k = 8: |reported - from scratch|", abs(u64 - mg64["utilities"][-1]), "per sweep vs oracle", d64)
        assert abs(u64 - mg64["utilities"][-1]) < 1e-11 and d64.max() < 1e-5


def test_final_parity_policy(oref):
    """bca_parity="final": wider sweeps than the default (four times), the utility after the LAST sweep within 1e-5 of
    the sequential oracle (intermediate sweeps may sit a few 1e-5 away); "per_sweep" (default) holds every sweep."""
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f
    from xcolumns_amd.synthetic import make_csr

    n, m, k = 100_000, 30_000, 5
    Y = make_csr(n, m, 50, seed=20240001, k=k)
    metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
    _, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=13, max_iters=5, tolerance=-1.0)
    _, md = f(Y, k, seed=13, max_iters=5, tolerance=-1.0, return_meta=True, bca_diagnostics=True)
    _, mf = f(Y, k, seed=13, max_iters=5, tolerance=-1.0, return_meta=True, bca_diagnostics=True, bca_parity="final")
    dd = np.abs(np.asarray(md["utilities"]) - np.asarray(mo["utilities"]))
    df = np.abs(np.asarray(mf["utilities"]) - np.asarray(mo["utilities"]))
    print("per_sweep", md["wavefronts"], dd, "\nfinal    ", mf["wavefronts"], df)
    assert mf["wavefronts"][0] == 4 * md["wavefronts"][0]
    assert dd.max() < 1e-5 and df[-1] < 1e-5 and df.max() < 5e-5
    with pytest.raises(ValueError):
        f(Y, k, bca_parity="sometimes")
    # skewed labels: the default walks this matrix's first sweep sequentially (0.26 s); "final" takes the whole GPU from
    # the first sweep on -- what that leaves behind (a few 1e-4) heals in the second sweep
    Yz = make_csr(n, m, 50, seed=20240001, zipf=True, k=k)
    _, moz = oref.predict_using_bc_with_0approx(Yz, metric, k, skip_tn=True, seed=13, max_iters=4, tolerance=-1.0)
    _, mfz = f(Yz, k, seed=13, max_iters=4, tolerance=-1.0, return_meta=True, bca_diagnostics=True, bca_parity="final")
    dz = np.abs(np.asarray(mfz["utilities"]) - np.asarray(moz["utilities"]))
    print("final, Zipf", mfz["wavefronts"], dz)
    assert mfz["wavefronts"][0] > 1000 and dz[1:].max() < 1e-5 and dz[0] < 2e-3


def test_visiting_order_paths_agree():
    """The visiting order reaches the sweeps through a worker thread, a pinned buffer and -- when numpy's generator
    is the PCG64 this build was checked against -- the library's own walk of numpy's stream; whichever path is
    taken, the exact mode returns the same prediction (the orders are the same)."""
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f
    from xcolumns_amd.synthetic import make_csr

    Y = make_csr(60_000, 8_000, 30, seed=4, k=5)
    runs = []
    for env in ({}, {"XCOLUMNS_ORDER_FAST_SHUFFLE": "0"}, {"XCOLUMNS_ORDER_PREFETCH": "0"},
                {"XCOLUMNS_ORDER_PREFETCH": "0", "XCOLUMNS_ORDER_FAST_SHUFFLE": "0"}):
        os.environ.update(env)
        try:
            P, meta = f(Y, 5, seed=21, max_iters=2, tolerance=-1.0, return_meta=True, bca_waves=1)
        finally:
            for key in env:
                os.environ.pop(key)
        runs.append((P.indices.copy(), meta["utilities"]))
    for idx, u in runs[1:]:
        assert np.array_equal(idx, runs[0][0]) and u == runs[0][1]


@pytest.mark.parametrize("m", [1, 63, 1000, 30_000, 500_001, 2_800_000])
def test_scatter_sum_against_index_add(m):
    """xc_scatter_sum_f32 (bucketed counting sort + LDS sums) = the plain float64 scatter-add, for column sums
    (pair = 0) and the {value, 1 - value} statistics of a prediction (pair = 1): every label written, duplicates summed,
    label ids at both ends of the range, label spaces that are not a multiple of the bucket size."""
    import ctypes

    from xcolumns_amd import _device as D, _lib

    dev = D.require_gpu()
    g = torch.Generator(device=dev)
    g.manual_seed(m)
    n_items = 700_001
    idx = torch.randint(0, m, (n_items,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    idx[:3] = 0
    idx[-3:] = m - 1
    val = torch.rand(n_items, generator=g, device=dev, dtype=torch.float32) ** 3
    nbytes = ctypes.c_int64(0)
    _lib.call("xc_scatter_sum_workspace_bytes", n_items, m, ctypes.byref(nbytes))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    for pair in (0, 1):
        out = torch.full(((2 if pair else 1) * m + 1,), -7.0, dtype=torch.float64, device=dev)
        _lib.call("xc_scatter_sum_f32", n_items, D.ptr(idx), D.ptr(val), m, pair, D.ptr(out), D.ptr(ws), D.stream())
        if pair:
            want = torch.zeros(2 * m, dtype=torch.float64, device=dev)
            want.index_add_(0, idx.long() * 2, val.double())
            want.index_add_(0, idx.long() * 2 + 1, (1.0 - val).double())
        else:
            want = torch.zeros(m, dtype=torch.float64, device=dev)
            want.index_add_(0, idx.long(), val.double())
        torch.testing.assert_close(out[:-1], want, rtol=1e-13, atol=1e-9)   # float64 summation order
        assert float(out[-1]) == -7.0                       # nothing written past the last label

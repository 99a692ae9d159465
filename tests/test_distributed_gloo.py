"""world_size-2 `gloo` test of the row-sharded BCA control flow on CPU.

No GPU here, so the per-rank engine is a checker-backed stand-in built on the
CPU oracle (tests may use the oracle; the product never does): it implements the
same engine interface as BcaCsrEngine -- local sweep against "global statistics
at the sweep start + own updates", local from-scratch tp/fp/fn, ONE all-reduce
per sweep boundary.  What is under test is the product's driver
(xcolumns_amd.distributed.predict_bca_csr_sharded + block_coordinate.run_bca_sweeps):
sharding, per-rank visiting orders, the all-reduce payload, identical utilities
and stopping decision on every rank."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from scipy.sparse import csr_matrix, vstack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShardEngine:
    """CPU stand-in for BcaCsrEngine over one rank's rows (test infrastructure)."""

    def __init__(self, shard, k, gain_spec, utility_spec, maximize, skip_tn, n_total, comm):
        from oracle import ref as oref
        self.o = oref
        self.Y, self.k, self.comm, self.n_total = shard, k, comm, n_total
        self.maximize, self.skip_tn = maximize, skip_tn
        self.metric = oref.make_metric(gain_spec.base, epsilon=gain_spec.epsilon, beta=gain_spec.beta,
                                       k=gain_spec.kf, mixed=gain_spec.mixed, alpha=gain_spec.alpha, m=gain_spec.mf)
        self.umetric = oref.make_metric(utility_spec.base, epsilon=utility_spec.epsilon, beta=utility_spec.beta,
                                        k=utility_spec.kf, mixed=utility_spec.mixed, alpha=utility_spec.alpha,
                                        m=utility_spec.mf)
        self.m = shard.shape[1]
        self.stats = np.zeros((4, self.m))
        self._changed = 0

    def init_top(self):
        P = self.o.predict_top_k(self.Y, self.k)
        self.pred_idx = np.ascontiguousarray(P.indices, dtype=np.int32)
        self.pred_data = np.ones(self.pred_idx.size, dtype=self.Y.dtype)

    def reset_state(self, greedy):
        assert not greedy

    def _pred(self):
        n = self.Y.shape[0]
        return csr_matrix((self.pred_data, self.pred_idx, np.arange(n + 1, dtype=np.int32) * self.k), shape=self.Y.shape)

    def recompute_utility_sum(self, n_u):
        tp, fp, fn, _ = self.o.calculate_confusion_matrix(self.Y, self._pred(), skip_tn=True)
        t = torch.from_numpy(np.stack([tp, fp, fn]))
        self.comm.all_reduce(t)                       # the ONE collective of a sweep boundary
        tp, fp, fn = t.numpy()
        tn = np.full(self.m, -1.0) if self.skip_tn else self.n_total - tp - fp - fn
        self.stats = np.stack([tp, fp, fn, tn]).copy()
        return float(self.o.metric_values(self.umetric, tp / n_u, fp / n_u, fn / n_u, tn / n_u).sum())

    def reset_changed(self):
        self._before = self.pred_idx.copy()

    def sweep(self, order, n_order, n_waves, greedy=False):
        o, Y = self.o, self.Y
        order = np.arange(n_order, dtype=np.int64) if order is None else np.ascontiguousarray(order, dtype=np.int64)
        indptr = np.ascontiguousarray(Y.indptr, dtype=np.int32)
        indices = np.ascontiguousarray(Y.indices, dtype=np.int32)
        tp, fp, fn, tn = (np.ascontiguousarray(v) for v in self.stats)
        sfx = "_f32" if Y.dtype == np.float32 else "_f64"
        getattr(o.lib(), "oracle_bca_sweep_csr" + sfx)(
            ctypes.c_int64(self.n_total), ctypes.c_int64(self.m), ctypes.c_int64(order.size), o._p(order),
            o._p(indptr), o._p(indices), o._p(np.ascontiguousarray(Y.data)), o._p(self.pred_idx), o._p(self.pred_data),
            ctypes.c_int(self.k), o._p(tp), o._p(fp), o._p(fn), o._p(tn), ctypes.byref(self.metric),
            ctypes.c_int(0), ctypes.c_int(int(self.maximize)), ctypes.c_int(int(self.skip_tn)))

    def sync_column_sums(self):
        pass

    def rows_changed(self):
        k = self.k
        c = int((self._before.reshape(-1, k) != self.pred_idx.reshape(-1, k)).any(axis=1).sum())
        t = torch.tensor([c], dtype=torch.int64)
        self.comm.all_reduce(t)
        return int(t.item())


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xcolumns_amd.distributed import TorchComm, predict_bca_csr_sharded, shard_csr
    from xcolumns_amd.metrics import binary_f1_score_on_conf_matrix
    from xcolumns_amd.synthetic import make_csr

    Y = make_csr(3001, 200, 20, seed=99, k=4)           # odd row count: uneven shards
    comm = TorchComm()
    shard = shard_csr(Y, world, rank)
    P, meta = predict_bca_csr_sharded(shard, binary_f1_score_on_conf_matrix, 4, comm, skip_tn=True, seed=13,
                                      max_iters=6, tolerance=1e-7,
                                      engine_factory=lambda *a: OracleShardEngine(*a))
    q.put((rank, meta["utilities"], meta["iters"], P.indices.copy(), comm.calls, comm.bytes_reduced))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_bca_two_ranks_gloo(oref):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, u0, it0, idx0, calls0, bytes0), (r1, u1, it1, idx1, calls1, bytes1) = res
    # identical trace and stopping decision on both ranks
    assert u0 == u1 and it0 == it1 and 1 <= it0 <= 6
    # collectives: one row-count exchange, then (1 stats + 1 changed-rows) per recompute; nothing per row
    assert calls0 == calls1 == 1 + 1 + 2 * it0
    from xcolumns_amd.synthetic import make_csr
    Y = make_csr(3001, 200, 20, seed=99, k=4)
    n, m, k = Y.shape[0], Y.shape[1], 4
    # the reported last utility is the macro-F1 of the assembled prediction, recomputed sequentially
    P = csr_matrix((np.ones(n * k, dtype=np.float32), np.concatenate([idx0, idx1]), np.arange(n + 1) * k), shape=(n, m))
    metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
    tp, fp, fn, tn = oref.calculate_confusion_matrix(Y, P, skip_tn=True)
    u = oref.calculate_utility(metric, "mean", tp / n, fp / n, fn / n, tn / n)
    assert abs(u - u0[-1]) < 1e-12
    # against the unsharded sequential oracle: better than top-k, close to its optimum
    _, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=13, max_iters=6, tolerance=1e-7)
    top = oref.predict_top_k(Y, k)
    tp, fp, fn, tn = oref.calculate_confusion_matrix(Y, top, skip_tn=True)
    u_top = oref.calculate_utility(metric, "mean", tp / n, fp / n, fn / n, tn / n)
    assert u0[0] > u_top
    assert abs(u0[-1] - mo["utilities"][-1]) < 2e-4, (u0, mo["utilities"])

"""Row-sharded BCA over the GPUs of one node: one process per GPU
(``torch.distributed``, backend "nccl" = RCCL over xGMI).

The reference is single-process (no collective call sites); the
shard/all-reduce scheme is this build's own.  Rows ("instances") are split into
contiguous blocks, one per rank; the per-label records are replicated.  Per
sweep every rank walks its own rows against ``global statistics at the start of
the sweep + its own updates`` and the sweep boundary is the reference's
from-scratch recompute (block_coordinate.py:465-467) done as: local tp/fp of the
rank's rows -> ONE all-reduce (sum, float64, 2*m values) -> identical global
statistics and utility on every rank, hence an identical stopping decision.
No other collective is on the data path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist
from scipy.sparse import csr_matrix

from . import _lib


class TorchComm:
    """Thin view of a torch.distributed process group: sum all-reduce in place."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.bytes_reduced = 0
        self.calls = 0
        self._timed = None   # list of (start, stop) event pairs while start_timing() is in effect

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        timed = self._timed is not None and t.is_cuda
        if timed:
            # on the stream the kernels run on: the collective's own stream is ordered behind `e0` and
            # this stream waits for the collective before `e1`, so the span is the all-reduce as the
            # sweep loop pays for it
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        if timed:
            e1.record()
            self._timed.append((e0, e1))
        self.bytes_reduced += t.numel() * t.element_size()
        self.calls += 1
        return t

    def start_timing(self) -> None:
        """bench.py: time every all-reduce of GPU tensors from here on with an event pair."""
        self._timed = []

    def stop_timing(self) -> float:
        """Total milliseconds of the all-reduces since start_timing() (synchronises)."""
        pairs, self._timed = self._timed or [], None
        if not pairs:
            return 0.0
        torch.cuda.synchronize()
        return float(sum(a.elapsed_time(b) for a, b in pairs))

    def all_reduce_async(self, t: torch.Tensor):
        """Start a sum all-reduce of `t` and return at once; ``.wait()`` on the returned handle orders the
        current stream (RCCL) / the host (gloo) behind its completion.  The overlapped mid-sweep exchange
        (block_coordinate.BcaCsrEngine.pipeline_step) runs the next segment of the sweep meanwhile."""
        self.bytes_reduced += t.numel() * t.element_size()
        self.calls += 1
        if self.world > 1:
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return _Done()

    def all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        """Control-plane reduction (argument checks that every rank must agree on); not counted
        as data-path traffic."""
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t


class _Done:
    """Handle of a collective that needed no communication (one rank)."""

    def wait(self):
        return True


def exchange_changes(comm: "TorchComm", records: torch.Tensor, snapshot: torch.Tensor,
                     buf: Optional[torch.Tensor] = None) -> None:
    """Mid-sweep exchange between row shards: every rank publishes what ITS rows changed in the
    replicated per-label `records` since `snapshot` was taken and takes in the other ranks' changes
    (one all-reduce of the difference), in place; `snapshot` is left equal to the merged records.
    Works on any device (the GPU engine passes its float32 records, the CPU tests a float64 table).
    Nothing else touches `records` between the three steps (the sweep waits for the exchange), so the merged
    records are simply snapshot + sum over ranks of (records - snapshot): three element-wise passes, no
    temporaries when `buf` (same shape) is given."""
    buf = torch.sub(records, snapshot, out=buf)       # what my rows changed
    comm.all_reduce(buf)                              # what everybody's rows changed
    torch.add(snapshot, buf, out=records)
    snapshot.copy_(records)


# Cross-rank staleness: within a sweep a rank does not see what the other ranks' rows change -- the sweep
# is Gauss-Seidel inside a shard and Jacobi across shards.  Measured with the sequential checker engine
# on 2 and 8 gloo ranks (tests/studies/shard_exchange_study.py, profiles/r02_shard_exchange_study.txt;
# macro-F1, k = 5, |utility - sequential reference| per sweep):
#   * ONE exchange per sweep (only the boundary all-reduce, north_star's scheme) does not merely lag: with
#     8 shards every rank corrects the same imbalance of a label with its own rows, the corrections add up
#     8-fold, and on uniform label popularity the utility FALLS from sweep to sweep (20 K x 6 K:
#     4.9e-3, 6.4e-3, 8.6e-3, ... ; 40 K x 2.5 K: 1e-3 -> 3e-3); 2 shards converge, slowly (x0.7 / sweep).
#   * S exchanges per sweep contract: 8 shards, S = 2: x0.45 per sweep; S = 4: x0.3 (1.1e-4, 3.2e-5, 8.3e-6,
#     2.7e-6, 8.9e-7, 2.2e-7); S = 8: 4.9e-5, 1.3e-5, 3.3e-6, 8.2e-7, ...  The first-sweep difference is about
#     4e-4 / S (uniform) .. 6e-3 / S (Zipf, where every row changes) and heals in the following sweeps.
# "auto": as many exchanges as the sweep is expected to change rows (up to EXCHANGES_MAX), never fewer than
# the number that keeps the iteration contracting for the shard count.
import os as _os

# Round 3: 16 / 32 (round 2: 8 / 16) -- 8 shards, 24 K x 1.5 K: 2.0e-5, 6.1e-6, 2.2e-6, ... per sweep (was 4.5e-5, 1.5e-5,
# 4.3e-6); 32 / 64 holds 1e-5 from the first sweep on (7.6e-6, 3.5e-6, 1.5e-6) at twice the exchanges.
EXCHANGES_MAX = int(_os.environ.get("XCOLUMNS_BCA_EXCHANGES_MAX", "16"))
_EXCHANGE_PER_CHANGED_SHARE = float(_os.environ.get("XCOLUMNS_BCA_EXCHANGE_FACTOR", "32"))   # S = ceil(this * share of rows the sweep is expected to change)


def min_exchanges(world: int) -> int:
    """Fewest exchanges per sweep at which the sharded iteration contracts towards the sequential one."""
    return 1 if world <= 1 else (2 if world <= 4 else 4)


def exchanges_for_sweep(setting, changed_rows: Optional[float], n_total: int, world: int) -> int:
    """Exchanges of a sweep: `setting` = an integer (fixed) or "auto" (from the number of rows the last KNOWN
    sweep changed over all ranks; None before any is known = half of the rows)."""
    if world <= 1:
        return 1
    if setting not in (None, "auto", 0):
        return max(1, int(setting))
    share = 0.5 if changed_rows is None else float(changed_rows) / max(1, n_total)
    want = int(-(-_EXCHANGE_PER_CHANGED_SHARE * share // 1))
    return int(max(min_exchanges(world), min(EXCHANGES_MAX, want)))


def global_policy_inputs(comm: "TorchComm", skewed_local: bool, nnz_local: int, n_total: int, device=None) -> Tuple[bool, float]:
    """(skewed, mean stored entries per row) over ALL ranks' rows: what a sharded run feeds block_coordinate.WavePolicy,
    identical on every rank (one MAX and one SUM reduction of a scalar)."""
    flag = torch.tensor([int(bool(skewed_local))], dtype=torch.int64, device=device)
    comm.all_reduce_max(flag)
    nnz = torch.tensor([int(nnz_local)], dtype=torch.int64, device=device)
    if comm.world > 1:
        dist.all_reduce(nnz, op=dist.ReduceOp.SUM, group=comm.group)
    return bool(flag.item()), float(nnz.item()) / max(1, int(n_total))


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block of `rank`: sizes differ by at most one row."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_csr(y_proba: csr_matrix, world: int, rank: int) -> csr_matrix:
    """Rows [lo, hi) of `y_proba` as their own CSR matrix (indptr rebased; the
    indices / data slices are contiguous views)."""
    lo, hi = shard_bounds(y_proba.shape[0], world, rank)
    s, e = int(y_proba.indptr[lo]), int(y_proba.indptr[hi])
    indptr = (y_proba.indptr[lo:hi + 1] - y_proba.indptr[lo]).astype(y_proba.indptr.dtype)
    return csr_matrix((y_proba.data[s:e], y_proba.indices[s:e], indptr), shape=(hi - lo, y_proba.shape[1]))


def local_order(global_order: np.ndarray, lo: int, hi: int) -> np.ndarray:
    """The visiting order a rank uses: the global permutation restricted to its
    row block, kept in the global order, rebased to local row ids."""
    sel = global_order[(global_order >= lo) & (global_order < hi)]
    return (sel - lo).astype(np.int32)


def predict_bca_csr_sharded(
    y_proba_shard: csr_matrix,
    binary_metric_func,
    k: int,
    comm: TorchComm,
    metric_aggregation: str = "mean",
    metric_kwargs=None,
    maximize: bool = True,
    tolerance: float = 1e-6,
    max_iters: int = 100,
    shuffle_order: bool = True,
    skip_tn: bool = False,
    seed: Optional[int] = None,
    verbose: bool = False,
    bca_waves: Optional[int] = None,
    bca_exchanges: Optional[int] = None,
    engine_factory=None,
):
    """BCA (init "top") with the rows sharded over the ranks of `comm`: every rank
    passes ITS contiguous block of rows and gets back the prediction for them
    (csr_matrix) plus the meta dict, identical on all ranks.

    The global visiting order is the reference's stream
    (``np.random.default_rng(seed)``, block_coordinate.py:413-419) over the global
    row ids, restricted to the rank's block (:func:`local_order`).
    `engine_factory(csr_shard, k, gain_spec, utility_spec, maximize, skip_tn, n_total,
    comm)` builds the per-rank engine; the default is the GPU engine.
    `bca_exchanges` (default "auto", env XCOLUMNS_BCA_EXCHANGES): how often per sweep the ranks
    exchange what their rows changed; 1 = only the all-reduce of the from-scratch statistics
    at the sweep boundary (north_star's scheme), S > 1 = S - 1 more all-reduces of the float32
    records (8 bytes per label) inside the sweep, which cuts the cross-rank staleness about S-fold;
    "auto" = :func:`exchanges_for_sweep`: several in the sweeps that change many rows, 1 afterwards."""
    from time import time

    from . import block_coordinate as bc
    from .metrics import resolve_metric

    if not isinstance(k, int):
        raise ValueError("k must be an integer")
    n_local, m = y_proba_shard.shape
    counts = torch.zeros(comm.world, dtype=torch.int64)
    counts[comm.rank] = n_local
    counts_dev = counts.device
    if engine_factory is None:
        from . import _device as D
        dev = D.require_gpu()
        counts = counts.to(dev)
        counts_dev = dev
    comm.all_reduce(counts)
    counts = counts.cpu().numpy()
    lo = int(counts[:comm.rank].sum())
    hi = lo + n_local
    n_total = int(counts.sum())
    # The checks _bc_csr applies to the whole matrix, on the shards: the verdict is all-reduced so
    # that every rank raises together instead of one rank leaving the others in a collective.
    row_nnz = np.diff(y_proba_shard.indptr)
    shortest = int(row_nnz.min()) if n_local > 0 else (1 << 30)
    verdict = torch.tensor([-shortest, int(row_nnz.max()) if n_local > 0 else 0], dtype=torch.int64,
                           device=counts_dev)
    comm.all_reduce_max(verdict)
    shortest, longest = -int(verdict[0].item()), int(verdict[1].item())
    if k < 1 or k > _lib.XC_MAX_K:
        raise ValueError(f"k must be in 1..{_lib.XC_MAX_K} for sparse y_proba on the GPU, got {k}")
    if n_total > 0 and shortest < k:
        raise ValueError(
            f"every row of a sparse y_proba must store at least k={k} entries on the GPU path "
            f"(shortest row over all ranks has {shortest})")
    if longest > _lib.XC_MAX_ROW_NNZ:
        raise ValueError(f"a row stores {longest} entries, the GPU path handles at most {_lib.XC_MAX_ROW_NNZ}")

    gain_spec = resolve_metric(binary_metric_func, metric_kwargs)
    utility_spec = resolve_metric(binary_metric_func, None)
    if engine_factory is None:
        csr = D.DeviceCSR.from_scipy(y_proba_shard, dev)
        eng = bc.BcaCsrEngine(csr, k, gain_spec, utility_spec, maximize=maximize, skip_tn=skip_tn,
                              n_total=n_total, comm=comm)
        if bca_exchanges:
            eng.exchanges = "auto" if bca_exchanges == "auto" else max(1, int(bca_exchanges))
        to_dev = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    else:
        eng = engine_factory(y_proba_shard, k, gain_spec, utility_spec, maximize, skip_tn, n_total, comm)
        eng.exchanges = "auto" if bca_exchanges in (None, "auto") else max(1, int(bca_exchanges))
        to_dev = lambda a: a  # noqa: E731
    eng.init_top()

    rng = np.random.default_rng(seed)
    order = np.arange(n_total)

    def next_order():
        if not shuffle_order:
            return None
        rng.shuffle(order)
        return to_dev(local_order(order, lo, hi))

    meta = {"utilities": [], "iters": 0, "time": time()}
    if engine_factory is None:
        # every input of the policy is GLOBAL: a rank that saw a skewed shard (or shorter rows) and chose another sweep
        # path than its peers would issue other collectives than they do
        skewed, row_nnz = global_policy_inputs(comm, eng.skewed, y_proba_shard.nnz, n_total, counts_dev)
        policy = bc.WavePolicy(n_local, fixed=bca_waves, world=comm.world, k=k, m=m, row_nnz=row_nnz, skewed=skewed)
    else:
        policy = _FixedWaves(bca_waves or 1)
    bc.run_bca_sweeps(eng, next_order, n_local, n_total, m, metric_aggregation, maximize, tolerance, max_iters,
                      False, policy, verbose, meta)
    if getattr(eng, "exchanges_used", None):
        meta["exchanges"] = list(eng.exchanges_used)
    meta["time"] = time() - meta["time"]
    idx = eng.pred_idx.cpu().numpy() if isinstance(eng.pred_idx, torch.Tensor) else np.asarray(eng.pred_idx)
    out_indptr = (np.arange(n_local + 1, dtype=np.int64) * k).astype(y_proba_shard.indptr.dtype)
    y_pred = csr_matrix((np.ones(n_local * k, dtype=y_proba_shard.dtype),
                         idx.astype(y_proba_shard.indices.dtype, copy=False), out_indptr), shape=(n_local, m))
    return y_pred, meta


class _FixedWaves:
    def __init__(self, w):
        self.w = int(w)

    def next(self, changed_prev=None, greedy=False):
        return self.w


# ---------------------------------------------------------------------------
# Frank-Wolfe over row shards
# ---------------------------------------------------------------------------

def find_classifier_using_fw_sharded(y_true_local: csr_matrix, y_proba_local: csr_matrix, metric_func, k: int,
                                     comm: TorchComm, **kwargs):
    """`find_classifier_using_fw` with the rows sharded over the ranks of `comm`: every rank holds
    rows [lo, hi) of `y_true` / `y_proba` and runs the same iteration on them; the only exchange is
    ONE all-reduce of the 3m label counts per weighted classifier (the FW iteration is exactly
    data-parallel: top-k and confusion counts are per row, everything else is O(m) and replicated).
    The counts are integers, so every rank sees bit-identical statistics and takes the same steps;
    the returned classifier is identical on all ranks and equal to the single-process one."""
    from .frank_wolfe import find_classifier_using_fw

    rows = torch.tensor([y_proba_local.shape[0]], dtype=torch.int64)
    dev = None
    if torch.cuda.is_available() and dist.get_backend(comm.group) == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        rows = rows.to(dev)
    comm.all_reduce(rows)
    return find_classifier_using_fw(y_true_local, y_proba_local, metric_func, k, comm=comm,
                                    n_total=int(rows.item()), **kwargs)

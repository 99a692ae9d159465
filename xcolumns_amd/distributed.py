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


class TorchComm:
    """Thin view of a torch.distributed process group: sum all-reduce in place."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.bytes_reduced = 0
        self.calls = 0

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        self.bytes_reduced += t.numel() * t.element_size()
        self.calls += 1
        return t


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

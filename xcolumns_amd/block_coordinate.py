"""Block coordinate ascent (BCA) prediction on MI355X.

Drop-in for the generic-BCA part of /root/reference/xcolumns/block_coordinate.py:
``predict_using_bc_with_0approx`` (:296-499), ``make_bc_wrapper`` and its products
(:709-801), ``predict_optimizing_instance_precision_using_bc`` (:804-835) and the
mixed-utility wrappers (:848-1045) -- same names, arguments, defaults, return
types, ``meta`` keys and ValueErrors.  The Python ``for i in order`` loop of the
reference (:448-463) and everything under it run in HIP kernels
(csrc/xc_bca.hip, csrc/xc_dense.hip) behind the C ABI; this module keeps only
the control flow around the sweeps: initial prediction, visiting order (the
reference's own ``np.random.default_rng(seed)`` stream), utility trace and the
stopping rule.

Extra keyword arguments (all optional, swallowed by ``**kwargs`` in the
reference's signature so call sites stay source-compatible):

``bca_waves``      number of wavefronts that walk the visiting order concurrently
                   (1 = the reference's exact sequential sweep; default: the
                   staleness-budget policy, :class:`WavePolicy`).
``bca_parity``     "per_sweep" (default): every sweep's utility within 1e-5 of the sequential reference -- shapes on
                   which nothing but the reference's own sequence achieves that (about one predicted row per label,
                   small skewed label spaces' first sweep, random / greedy starts' first sweep, ...) run the exact
                   one-wavefront sweep; "final": wider sweeps, the whole GPU on those shapes too, the bar holds for
                   the utility after the last sweep -- on the one-row-per-label shapes: within the reference's own
                   seed-to-seed spread (:class:`WavePolicy`; env XCOLUMNS_BCA_PARITY).
``bca_deterministic`` True: every sweep runs as the reference's own sequence (the ordered parallel sweep, csrc/xc_bca_ord.hip:
                   a window of rows in flight iterated to the sequential fixed point) -- the reference is deterministic given
                   `seed` (:413-419), and so is the prediction then, run to run; several times the default sweep's time.
                   Default False: the faster concurrent sweep whose result depends on timing in a few rows
                   (env XCOLUMNS_BCA_DETERMINISTIC).
``bca_ordered``    False: exact sweeps (``bca_waves=1`` and the shapes the policy runs exactly) use ONE wavefront instead of
                   the ordered parallel sweep (csrc/xc_bca_ord.hip: a window of rows in flight, iterated to the fixed
                   point that IS the sequential result) -- same predictions, for cross-checks (env XCOLUMNS_BCA_ORDERED=0).
``bca_diagnostics`` True: ``meta`` also carries "wavefronts" and "rows_changed" per sweep
                   (by default ``meta`` has exactly the reference's keys).
``order_backend``  "numpy" (default: the reference's RNG stream, generated on the
                   host and uploaded) or "device" (``torch.randperm`` on the GPU,
                   a different stream, no host work per sweep).
"""
from __future__ import annotations

import ctypes
import os
import time as _time
from time import time
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import torch
from scipy.sparse import csr_matrix

from . import _device as D
from . import _lib
from .metrics import (
    DeviceMetric,
    MetricSpec,
    binary_balanced_accuracy_on_conf_matrix,
    binary_f1_score_on_conf_matrix,
    binary_gmean_on_conf_matrix,
    binary_hmean_on_conf_matrix,
    binary_jaccard_score_on_conf_matrix,
    binary_precision_at_k_on_conf_matrix,
    binary_precision_on_conf_matrix,
    binary_recall_on_conf_matrix,
    resolve_metric,
)
from .types import DenseMatrix, Matrix, is_dense, is_matrix, is_sparse
from .utils import add_kwargs_to_signature, log_info, log_warning, random_at_k_csr, random_at_k_np
from .weighted_prediction import topk_csr_device, topk_dense_device

# ---------------------------------------------------------------------------
# how many wavefronts walk the visiting order concurrently
# ---------------------------------------------------------------------------
# The reference's sweep is sequential; here W rows are in flight at once.  A row that CHANGES its
# prediction is serialised against the other rows in flight by the kernel's commit protocol
# (csrc/xc_bca.hip); what is left is a row that was scored on a record another row changed a moment
# later and kept its (by then slightly wrong) decision.  Two rows interact when one changes a label the
# other holds as a candidate, so the natural measure of concurrency is rows in flight per LABEL, W / m,
# not per row.  Measured on MI355X, |utility - sequential oracle| of a sweep from the top-k start
# (profiles/r02_policy_study.txt; macro-F1, k = 5, 50 entries per row):
#     uniform label popularity:  ~2.5e-4 * (rows changed / n) * (W / m)     (n = 20 K .. 1 M)
#     Zipf(1) popularity, sweep 1 (every row changes, mostly into tail labels that hold one or two
#     rows): ~2e-3 * (W / m); from sweep 2 on below the uniform figure
# and the difference heals in the following sweeps.  The reference itself moves by 2.5e-6 (uniform) /
# 3.3e-5 (Zipf) standard deviation in sweep 1 when only its visiting order (`seed`) changes.
#
# bca_parity = "per_sweep" (default): EVERY sweep within north_star's 1e-5 of the sequential reference --
#     W_j = BETA * m * (50 / entries per row) * (5 / k)^2 * min(1, n k / 12 m)^2.5 * min(1, m / 30K)^0.5 * n / (2 * rows changed in sweep j-1)
#     (half of the rows assumed before sweep 1), a sixteenth of it (less when m < 200 K) for the first sweep when the label
#     popularity is skewed (hot labels present); converged sweeps use the whole GPU.
# bca_parity = "final": four times wider -- the utility after the LAST sweep is what is held to 1e-5
#     (intermediate sweeps of a top-k start stay within ~5e-5), for callers that only use the result.
# A changing row moves more labels the larger k is and the trajectories then settle in different, nearly
# equivalent optima (k = 64, n = 6000, m = 900: 1.6e-4 in sweep 1 at the k = 5 width, 1.3e-5 with (5 / k)^1.5), hence (5 / k)^2.
# round 3: 0.05 -> 0.025.  Over five visiting orders the worst sweep sat at 0.95 / 0.97 of the bar on 100 K x 30 K / 200 K x 60 K
# with 0.05, at 0.34 / 0.30 with 0.025, and no lower with 0.0125 (0.47 / 0.31: the floor of this chaotic quantity) --
# tests/studies/beta_margin_study.py, profiles/r03_beta_margin_study.txt; the tests hold 0.5 of the bar (tests/_parity.py)
_BETA = float(os.environ.get("XCOLUMNS_BCA_BETA", "0.025"))
_SKEWED_FIRST_SWEEP = 0.1232   # x 2 with _BETA / 2 in round 3: the skewed first sweep keeps the width measured in round 2 (976 wavefronts at 1 M x 500 K)
_FINAL_PARITY_FACTOR = 4.0
_STALE_BUDGET = float(os.environ["XCOLUMNS_BCA_STALE_BUDGET"]) if "XCOLUMNS_BCA_STALE_BUDGET" in os.environ else None
_MIN_WAVES = 1


def default_parity() -> str:
    p = os.environ.get("XCOLUMNS_BCA_PARITY", "per_sweep")
    if p not in ("per_sweep", "final"):
        raise ValueError("XCOLUMNS_BCA_PARITY / bca_parity must be 'per_sweep' or 'final'")
    return p


class WavePolicy:
    """Number of concurrent wavefronts for the next sweep."""

    def __init__(self, n_order: int, fixed: Optional[int] = None, budget: Optional[float] = None,
                 world: int = 1, k: int = 5, first_changed: float = 0.5, m: Optional[int] = None,
                 row_nnz: float = 50.0, skewed: bool = False, parity: Optional[str] = None, scale: float = 1.0,
                 first_sequential: bool = False, sweeps: Optional[int] = None):
        """`n_order`: rows THIS rank visits per sweep; `world`: ranks sharing the rows (the changed-row
        count fed to :meth:`next` is the global one); `k`: labels per row; `first_changed`: expected share
        of rows the first sweep changes (about half from the top-k prediction, all of them from a random
        or foreign one); `m`: labels, `row_nnz`: mean stored entries per row, `skewed`: hot labels present
        (first sweep narrower); `parity`: "per_sweep" | "final"; `scale`: extra factor on the width; `sweeps`:
        the most sweeps the run may do (max_iters), if known.
        `budget` (or XCOLUMNS_BCA_STALE_BUDGET) selects the round-1 rule instead: W = budget * n^2 / changed."""
        self.first_changed = float(first_changed)
        # the first sweep of a random / foreign start changes every label of every row: it runs as the
        # reference's sequential sweep (exact) unless the caller asked for "final" parity or a fixed width
        self.first_sequential = bool(first_sequential)
        self.sequential_below = 0      # a sweep the rule leaves fewer wavefronts than this runs sequentially (skewed labels)
        self.world = max(1, int(world))
        env = os.environ.get("XCOLUMNS_BCA_WAVES")
        self.fixed = int(fixed) if fixed else (int(env) if env else None)
        self.n = max(1, int(n_order))
        self.parity = default_parity() if parity is None else parity
        if self.parity not in ("per_sweep", "final"):
            raise ValueError("bca_parity must be 'per_sweep' or 'final'")
        k_scale = min(1.0, 5.0 / max(1, int(k))) ** 2
        if budget is None and _STALE_BUDGET is not None:
            budget = _STALE_BUDGET
        if budget is not None or m is None:
            b = 8e-3 if budget is None else float(budget)
            self.num = b * k_scale * float(self.n) * float(self.n)
            self.first_factor = 1.0
        else:
            # labels that hold about one predicted row each (n * k / m ~ 1: 150 K rows x 670 K labels) react to a
            # single row in flight with their whole statistic, and the differences no longer heal -- the
            # landscape is full of nearly equivalent optima (C3: 1e-5 at 0.4 % of the rows in flight, in every
            # sweep; 131 wavefronts: 0.3e-6 .. 1.3e-5 from run to run): the width shrinks with the 2.5th power of the
            # predicted rows per label below 12 (C3: 72 wavefronts; the north-star shape, 10 rows per label, keeps the whole GPU)
            per_label = float(self.n) * self.world * max(1, int(k)) / float(m)
            # and the bar is absolute: one label's F1 weighs 1 / m in the utility, so on a small label space a
            # handful of decisions that fall the other way are already 1e-5 (20 K x 5 K: 1.0-1.7e-5 at 0.064 m
            # rows in flight, 4e-6 at half of that)
            # (linear below 30 K labels since the fuzz of round 2: 31 K x 7 K, k = 3 measured 1.2-1.5e-5 at 358 wavefronts
            # and 4-7e-6 at 225 with the square root; tests/studies/fuzz_case.py 1042)
            small_m = min(1.0, float(m) / 30000.0)
            # (longer rows than 50 entries narrow the sweep -- more candidates per row meet a row in flight; shorter ones
            # do not widen it: 10 entries per row at 5x measured 2.6e-5 in sweep 1 on 31 K x 7 K)
            width = (_BETA * float(m) * min(1.0, 50.0 / max(1.0, float(row_nnz))) * k_scale * float(scale)
                     * min(1.0, per_label / 12.0) ** 2.5 * small_m)
            if self.parity == "final":
                width *= _FINAL_PARITY_FACTOR
            self.num = width * self.n / 2.0
            # ... and since the differences do not heal there, converging sweeps may not widen freely either: at most
            # twice the first sweep's width while a label holds fewer than four predicted rows (C3, third sweep at
            # 1.5x the first width: 9.2e-6; constant width 146: 8.6e-6, 3.8e-6, 6.1e-6, 4.1e-6, 3.1e-6)
            if per_label < 2.0 and self.parity == "per_sweep" and not self.fixed:
                # About ONE predicted row per label (150 K rows x 670 K labels): 35 / 19 / 55 wavefronts still ended a
                # sweep at 8.0e-6 and 1.02e-5 in two of five runs, and nothing heals afterwards.  The reference itself
                # moves by 1.2-3.8e-5 (standard deviation) there when only its visiting order changes, so its trajectory
                # cannot be tracked to 1e-5 by anything but its own sequence: these shapes run the EXACT sweep -- since
                # round 3 the ordered parallel sweep (csrc/xc_bca_ord.hip: 0.5-1.7 ms per sweep at C3; one wavefront took
                # 0.25 s) -- unless the caller asks for bca_parity="final" or a width.  (fixed = 1 means "exact" below too.)
                self.fixed = 1
            if per_label < 8.0 and float(m) < 30000.0 and self.parity == "per_sweep" and not self.fixed:
                # few predicted rows per label on a SMALL label space: one decision that falls the other way is the
                # whole bar (1 / m per label), and 17-35 wavefronts still measured 1.5e-5 on 37 K x 20 K, k = 2
                # (3.7 rows per label; the fuzz of round 2, seed 1065), 11-27 wavefronts 1.3e-5 on 22 K x 10.5 K, k = 2
                # (4.1; seed 3059): the exact sweep
                self.fixed = 1
            if 3.0 * max(1, int(k)) > float(row_nnz) and self.parity == "per_sweep" and not self.fixed:
                # a budget of more than a THIRD of a row's candidates (35 K x 14 K, 12 entries per row, k = 7: the fuzz of
                # round 2; half of a row until the fuzz of round 3 found 14 K x 5 K, 12 entries, k = 6 at 2.3e-5 with 7-29
                # wavefronts and 34 K x 19 K, 12 entries, k = 5 at 1.1e-5 with 18-43) is another shape whose trajectory
                # nothing but its own sequence tracks: TWO wavefronts already sit 1-5e-5 from it, and nearly half of the rows
                # still change in the second sweep
                self.fixed = 1
            full_width = False
            if per_label < 2.0 and self.parity == "final" and not self.fixed:
                # ... and with "final" parity they take the whole GPU: with the commit protocol 8, 64, 1024 and 8192
                # wavefronts all end a sweep 1e-7 .. 1.1e-5 from the sequential trajectory (three visiting orders each,
                # profiles/r02_c3_width.txt) -- what moves the utility there is the order in which conflicting rows
                # happen to commit, not how many rows are in flight, so a narrow sweep buys nothing
                full_width = True
                self.num = float(self.n) * float(self.n)
            if per_label < 4.0 and not full_width:
                width *= 0.5     # three runs at 71 / 39 / 110 wavefronts ended sweep 3 at 2.9e-6, 8.7e-6, 9.0e-6: too close
                self.num = width * self.n / 2.0
                self.width_cap = max(1, int(2.0 * width))
            # skewed popularity: every row changes in sweep 1, mostly into tail labels that hold one or two rows --
            # a single decision that falls the other way moves such a label's F1 by ~0.3, i.e. the utility by
            # 0.3 / m: on a 30 K-label space three of them are the whole 1e-5
            # (uniform popularity, round 3: HALF the width for the first sweep.  At 1 M x 500 K the whole GPU -- 7924 wavefronts
            # -- put sweep 1 at 0.78 / 0.53 of the bar on two of five visiting orders, reproducibly per order; the
            # difference falls in proportion to the width (0.29 at 2535, 0.13 at 1267 wavefronts), later sweeps sit below
            # 0.2, and the sweep kernel is throughput-bound from ~2500 wavefronts on, so the narrower first sweep costs
            # little: tests/studies/ns_first_sweep_margin.py, profiles/r03_ns_first_sweep_margin.txt)
            self.first_factor = _SKEWED_FIRST_SWEEP * min(1.0, float(m) / 200000.0) if skewed else 0.5
            if skewed and self.parity == "final" and (sweeps is None or int(sweeps) >= 2):
                # what a wide first sweep leaves behind on skewed labels heals in the next one: 100 K x 30 K Zipf at 8192
                # wavefronts 1.9-3.6e-4 after sweep 1, 2.9-4.3e-6 after sweep 2, <= 3e-7 from sweep 3 on (three visiting
                # orders, profiles/r02_c3_width.txt); 1 M x 500 K: 3.2e-5, then 3.8e-7.  "final" promises the last sweep.
                self.first_factor = 1.0
            # ... and when that leaves fewer than 64 wavefronts the first sweep runs as the reference's own sequence
            # (the exact sweep: 4.9 ms at 100 K x 30 K with the ordered parallel sweep, 0.26 s with one wavefront): on 100 K x 30 K Zipf ONE tail label that is predicted by a different row is 1-2e-5 of utility,
            # and 22 .. 250 wavefronts all measure 0.5-2e-5 in sweep 1 (six runs each, profiles/r02_c2_zipf_width.txt)
            # -- anything but one row at a time is a coin toss against the bar there.
            if (skewed and self.parity == "per_sweep" and not self.fixed
                    and self.num * self.first_factor / (self.n * self.first_changed) < 64.0):
                self.first_sequential = True
            # the same holds for a LATER sweep that still changes so many rows that the rule leaves it fewer than 64
            # wavefronts (the sweep after a greedy or random start on 22 K x 7 K Zipf: 54 wavefronts, 0.8-1.3e-5)
            if skewed and self.parity == "per_sweep" and not self.fixed:
                self.sequential_below = 64
            # round 3: on ANY shape.  A sweep the rule leaves a handful of wavefronts is both off the bar (25 K x 7 K, hmean after
            # a greedy start: 1.1e-5 at 5-9 wavefronts, the concurrent fuzz) and slower than the exact sweep, now that the exact
            # sweep is the ordered parallel one (a wavefront takes ~2.6 us per row: 7 ms for 25 K rows on 9 of them, 1-2 ms exact)
            if self.parity == "per_sweep" and not self.fixed:
                self.sequential_below = 64
                if self.num * self.first_factor / (self.n * self.first_changed) < 64.0:
                    self.first_sequential = True
        self.budget = self.num / (float(self.n) * float(self.n))   # the same rule as a share of n (diagnostics)
        info = _lib.device_info()
        self.cap = info["cu_count"] * info["waves_per_cu"]
        if getattr(self, "width_cap", None) is not None:
            self.cap = min(self.cap, self.width_cap)

    def next(self, changed_prev: Optional[int] = None, greedy: bool = False) -> int:
        """`greedy`: the first sweep of init_y_pred="greedy" -- every row is added to statistics that
        start from zero and no row is validated, which measures ~0.2 * W / n (5.6e-4 at the ordinary
        width, C2 shape): it runs 32x narrower, on every rank."""
        if self.fixed:
            return max(1, min(self.fixed, self.n))
        if greedy:
            if self.parity == "per_sweep":
                # ... and under the per-sweep bar it runs as the reference's sequential sweep like the first sweep of the
                # other non-top-k starts: 13 wavefronts on 214 K x 106 K Zipf measured 2.7e-5 (medium fuzz, seed 5005)
                return 1
            return int(max(1, min(self.cap, self.n, self.num / self.n / 32)))
        # sharded rows: the other ranks' updates are invisible within a sweep whatever W is (DESIGN.md
        # section 7), but this rank's own rows still follow the rule, on its share of the changes
        if changed_prev is None:
            if self.first_sequential:
                return 1
            want = int(self.num * self.first_factor / (self.n * self.first_changed))
        else:
            want = int(self.num / max(1.0, changed_prev / self.world))
            if want < self.sequential_below:
                return 1
        return int(max(1, min(self.cap, self.n, max(_MIN_WAVES, want))))

    def device_params(self):
        """(policy_num, world, min_waves, max_waves, fixed_waves) for xc_bca_pipeline_begin: the same
        rule as :meth:`next`, evaluated by the boundary kernel."""
        max_w = int(max(1, min(self.cap, self.n)))
        fixed = int(max(1, min(self.fixed, self.n))) if self.fixed else 0
        return self.num, self.world, _MIN_WAVES, max_w, fixed

    @property
    def sequential(self) -> bool:
        """bca_waves=1: the reference's exact sweep (one wavefront, host-paced)."""
        return bool(self.fixed) and min(self.fixed, self.n) == 1


def default_bca_waves(n_order: int) -> int:
    """Concurrent wavefronts of a first sweep under the default policy."""
    return WavePolicy(n_order).next(None)


# ---------------------------------------------------------------------------
# device-level engine (CSR)
# ---------------------------------------------------------------------------

class BcaCsrEngine:
    """Device-resident state of one BCA run over CSR rows held by THIS rank.

    HBM layout: CSR y_proba (int32 indptr/indices, f32|f64 data);
    prediction as fixed-stride ``pred_idx[n*k]`` + ``pred_eta[n*k]`` plus one byte per
    stored entry (``sel``); per-label records ``tpfp[m][2] = {tp, fp}`` float64,
    ``colsum[m]`` (s = tp + fn) and its per-entry expansion ``s_entry[nnz]``;
    ``acc[m][2]`` float64 for the from-scratch tp/fp of every sweep boundary (the
    all-reduce payload when rows are sharded over ranks).
    """

    def __init__(self, csr: D.DeviceCSR, k: int, gain_spec: MetricSpec, utility_spec: MetricSpec,
                 maximize: bool = True, skip_tn: bool = False, n_total: Optional[int] = None,
                 comm=None, use_shadow: Optional[bool] = None, deterministic: bool = False):
        if k < 1 or k > _lib.XC_MAX_K:
            raise ValueError(f"k must be in 1..{_lib.XC_MAX_K} for sparse y_proba on the GPU, got {k}")
        if csr.n > 0 and csr.min_row_nnz < k:
            # a short row would leave stale ids in its prediction slots and make the sweep read before
            # the row's first entry (xc_bca.hip load_row): refuse it here, whoever builds the engine
            raise ValueError(
                f"every row of a sparse y_proba must store at least k={k} entries on the GPU path "
                f"(shortest row has {int(csr.min_row_nnz)})")
        self.csr = csr
        self.k = int(k)
        # bca_deterministic: every sweep is the exact (ordered) one -- the reference's sequence, same seed same prediction
        self.deterministic = bool(deterministic)
        # one-wavefront (exact) sweeps run as the ordered parallel sweep (csrc/xc_bca_ord.hip) where it applies
        self.ordered = os.environ.get("XCOLUMNS_BCA_ORDERED", "1") != "0"
        self.gain_metric = gain_spec.to_c()
        self.utility_metric = utility_spec.to_c()
        self.maximize = bool(maximize)
        self.skip_tn = bool(skip_tn)
        self.n_total = int(csr.n if n_total is None else n_total)  # rows over all ranks
        self.comm = comm
        dev = csr.data.device
        self.dev = dev
        m = csr.m
        self.tpfp = torch.zeros((m, 2), dtype=torch.float64, device=dev)
        # float32 copy of tpfp gathered by the concurrent sweep (XCOLUMNS_BCA_SHADOW=0 disables)
        if use_shadow is None:
            use_shadow = os.environ.get("XCOLUMNS_BCA_SHADOW", "1") != "0"
        self.shadow = torch.zeros((m, 2), dtype=torch.float32, device=dev) if use_shadow else None
        self.colsum = torch.zeros(m, dtype=torch.float64, device=dev)
        # float64 colsum per stored entry: read by the sweeps that do not read the packed stream (exact
        # one-wavefront sweeps, float64 scores, XCOLUMNS_BCA_PACKED=0); allocated and filled on first need
        self.s_entry: Optional[torch.Tensor] = None
        self._s_entry_dirty = True
        # float32 scores: indices / data / sel / (float) s_entry interleaved in 12-byte entries, so a
        # candidate streams in as one 12-byte lane load (XCOLUMNS_BCA_PACKED=0 disables)
        self.packed = (torch.empty((max(1, csr.nnz), 3), dtype=torch.int32, device=dev)
                       if csr.data.dtype == torch.float32 and m <= (1 << 25)
                       and os.environ.get("XCOLUMNS_BCA_PACKED", "1") != "0" else None)
        self._pack_dirty = True
        # Hot labels (the head of a skewed popularity): stored in >= n/32 rows, at most 63 of them.
        # A sweeping wave batches its deltas to them (xc_bca_sweep_csr `hot_labels`);
        # XCOLUMNS_BCA_HOT=0 disables.  Uniform popularity has none.
        self.hot_slot = self.hot_labels = None
        self.skewed = False      # a label stored in >= n/32 rows: the first sweep from top-k changes nearly every row
        if csr.nnz > 0:
            # label frequencies from a strided sample of the stored entries (at most ~512 K of them: a label stored in n / 32 rows is hit ~300 times): a label
            # stored in n / 32 rows shows up thousands of times in it, and the hot set only steers how the
            # kernel batches its atomics -- not worth a full histogram of the matrix (2.1 ms at 1M x 500K)
            stride = max(1, csr.nnz // (1 << 19))
            # "hot" presumes a record that sums so many rows that a few rows' delay cannot move a gain: at
            # least 4096 stored entries (a label stored in 500 of 6000 rows is busy, but its tp is ~10)
            busy_min, hot_min = max(64, csr.n // 32), max(4096, csr.n // 32)
            # at most 32 * (entries per row) labels can be stored in n / 32 rows each; sampling noise on top
            cap = int(min(m, 40 * (csr.nnz // max(1, csr.n) + 1) + 64))
            counts = torch.empty(m, dtype=torch.int32, device=dev)
            found = torch.empty(2 * cap + 1, dtype=torch.int32, device=dev)
            _lib.call("xc_label_busy_list", csr.nnz, D.ptr(csr.indices), stride, m, max(1, busy_min // stride), cap,
                      D.ptr(counts), D.ptr(found[1:]), D.ptr(found[:1]), D.stream())
            found = found.cpu().numpy()
            n_found = min(int(found[0]), cap)
            pairs = found[1:1 + 2 * n_found].reshape(n_found, 2).astype(np.int64)
            pairs[:, 1] *= stride
            # most frequent first, ties by label id: the same table whatever order the kernel listed them in
            pairs = pairs[np.lexsort((pairs[:, 0], -pairs[:, 1]))]
            self.skewed = n_found > 0
            hot = pairs[pairs[:, 1] >= hot_min][:max(0, min(63, int(os.environ.get("XCOLUMNS_BCA_HOT_MAX", "63"))))]
            n_hot = int(hot.shape[0])
            labels = torch.from_numpy(hot[:, 0].astype(np.int32)).to(dev)
            if not (self.packed is not None and self.shadow is not None
                    and os.environ.get("XCOLUMNS_BCA_HOT", "1") != "0"):
                n_hot = 0
            if n_hot > 0:
                self.hot_labels = torch.full((64,), -1, dtype=torch.int32, device=dev)
                self.hot_labels[1:n_hot + 1] = labels
                self.hot_slot = torch.zeros(m, dtype=torch.uint8, device=dev)
                self.hot_slot[labels.long()] = torch.arange(1, n_hot + 1, dtype=torch.uint8, device=dev)
        # from-scratch {tp, fp} of a sweep boundary; slot 2m carries the changed-row count
        self.acc = torch.zeros(2 * m + 1, dtype=torch.float64, device=dev)
        # pipelined concurrent sweeps push their committed changes into the float64 records instead of rebuilding
        # `acc` (xc_bca_set_acc_delta; XCOLUMNS_BCA_ACC_DELTA=0 = from scratch in every sweep, as round 1 did)
        _lib.call("xc_bca_set_acc_delta", int(os.environ.get("XCOLUMNS_BCA_ACC_DELTA", "1") != "0"))
        if os.environ.get("XCOLUMNS_BCA_HOT_UNPUBLISHED"):     # study knob: share of the rows whose hot-label deltas may wait
            _lib.call("xc_bca_set_tuning", -1.0, float(os.environ["XCOLUMNS_BCA_HOT_UNPUBLISHED"]))
        # sharded rows: exchanges of the ranks' changes per sweep (1 = only the all-reduce of the
        # from-scratch statistics at the sweep boundary, the north-star scheme)
        # sweep), S > 1 = S - 1 more all-reduces of the float32 records inside the sweep, "auto" = several in
        # the sweeps that change many rows and 1 once converged (distributed.exchanges_for_sweep)
        env_x = os.environ.get("XCOLUMNS_BCA_EXCHANGES", "auto")
        self.exchanges = "auto" if env_x == "auto" else max(1, int(env_x))
        self._changed_known = None   # rows changed (all ranks) in the last sweep whose result the host has read
        self._acc_filled = False    # did the last sweep leave the new prediction's statistics in acc
        self._changed_last = 0
        self.partials = torch.zeros(_lib.XC_UTILITY_PARTIALS + 1, dtype=torch.float64, device=dev)
        self.changed = torch.zeros(1, dtype=torch.int64, device=dev)
        self.pred_idx: Optional[torch.Tensor] = None
        self.pred_eta: Optional[torch.Tensor] = None
        # one byte per stored entry of y_proba: is it in the row's current prediction
        self.sel = torch.zeros(max(1, csr.nnz), dtype=torch.uint8, device=dev)
        self.orphans: Optional[torch.Tensor] = None

    # -- initial prediction --------------------------------------------------
    def init_top(self):
        """predict_top_k (block_coordinate.py:40)."""
        self.pred_idx, _, self.pred_eta = topk_csr_device(self.csr, self.k, want_eta=True, out_sel=self.sel)
        self.orphans = None
        self._pack_dirty = True

    def init_indices(self, pred_idx: torch.Tensor):
        """An explicit prediction: k column ids per row.  Columns a row does not store
        ("orphans") are kept aside; they leave the prediction in the first sweep."""
        c = self.csr
        self.pred_idx = pred_idx.to(device=self.dev, dtype=torch.int32).contiguous()
        self.pred_eta = torch.empty(c.n * self.k, dtype=c.data.dtype, device=self.dev)
        self.sel.zero_()
        orphans = torch.empty(c.n * self.k, dtype=torch.int32, device=self.dev)
        _lib.call("xc_bca_gather_pred_eta", c.n, D.ptr(c.indptr), D.ptr(c.indices), D.ptr(c.data), c.code,
                  D.ptr(self.pred_idx), self.k, D.ptr(self.pred_eta), D.ptr(self.sel), D.ptr(orphans),
                  D.stream())
        self.orphans = orphans if bool((orphans >= 0).any().item()) else None
        self._pack_dirty = True

    # -- plan: per-run constants bound once on the C side ------------------------------
    def _plan_handle(self):
        """(Re)bind when the prediction buffers were replaced by an init_* call."""
        key = (self.pred_idx.data_ptr(), self.pred_eta.data_ptr(),
               0 if self.s_entry is None else self.s_entry.data_ptr())
        if getattr(self, "_plan_key", None) != key:
            self._drop_plan()
            h = ctypes.c_void_p()
            c = self.csr
            _lib.call("xc_bca_plan_create", ctypes.byref(h), c.n, c.m, self.n_total, D.ptr(c.indptr), D.ptr(c.indices),
                      D.ptr(c.data), c.code, int(c.max_row_nnz), self.k, D.ptr(self.pred_idx), D.ptr(self.pred_eta),
                      D.ptr(self.sel), D.ptr(self.tpfp), D.ptr(self.shadow), D.ptr(self.colsum), D.ptr(self.s_entry),
                      D.ptr(self.packed), D.ptr(self.hot_labels), D.ptr(self.acc), D.ptr(self.partials), ctypes.byref(self.gain_metric),
                      ctypes.byref(self.utility_metric), int(self.maximize), int(self.skip_tn))
            self._plan, self._plan_key = h, key
        return self._plan

    def _drop_plan(self):
        if getattr(self, "_plan", None) is not None:
            _lib.load().xc_bca_plan_destroy(self._plan)
            self._plan = None
            self._plan_key = None

    # -- statistics -------------------------------------------------------------
    def reset_state(self, greedy: bool):
        """Zero the statistics; unless `greedy`, colsum <- column sums of y_proba over
        all ranks, expanded per stored entry."""
        self.tpfp.zero_()
        if self.shadow is not None:
            self.shadow.zero_()
        self.colsum.zero_()
        if not greedy:
            # the column sums of y_proba depend on the matrix alone: kept with the DeviceCSR (a caller that holds on to its
            # matrix -- another k, another metric -- pays the 50 M-entry scatter, 1.25 ms at 1 M x 500 K, once)
            self.colsum.copy_(self.csr.column_sums())
            if self.comm is not None:
                self.comm.all_reduce(self.colsum)
            self._expand_colsum()

    def _scatter_sum(self, n_items: int, idx: torch.Tensor, val: torch.Tensor, out: torch.Tensor, pair: bool) -> bool:
        """Per-label float64 sums of (label, float32 value) pairs by a bucketed counting sort + LDS sums
        (xc_scatter_sum_f32: 2.1 -> ~0.6 ms for the 50 M entries of 1M x 500K) instead of one global float atomic per
        pair.  False: not applicable (float64 values, tiny inputs, a label space beyond the bucket tables) -- the
        caller takes the atomic kernel.  XCOLUMNS_BCA_SCATTER=0 disables."""
        if (val.dtype != torch.float32 or n_items < (1 << 18) or self.csr.m > 16384 * 2048
                or os.environ.get("XCOLUMNS_BCA_SCATTER", "1") == "0"):
            return False
        nbytes = ctypes.c_int64(0)
        _lib.call("xc_scatter_sum_workspace_bytes", int(n_items), self.csr.m, ctypes.byref(nbytes))
        ws = torch.empty(int(nbytes.value), dtype=torch.uint8, device=self.dev)
        _lib.call("xc_scatter_sum_f32", int(n_items), D.ptr(idx), D.ptr(val), self.csr.m, int(pair), D.ptr(out), D.ptr(ws),
                  D.stream())
        ws.record_stream(torch.cuda.current_stream())
        return True

    def _expand_colsum(self):
        """colsum changed: its per-entry copies (s_entry, the packed stream) are stale."""
        self._s_entry_dirty = True
        self._pack_dirty = True

    def _ensure_s_entry(self):
        if self.s_entry is None:
            self.s_entry = torch.empty(max(1, self.csr.nnz), dtype=torch.float64, device=self.dev)
            self._s_entry_dirty = True
        if self._s_entry_dirty:
            c = self.csr
            _lib.call("xc_bca_expand_colsum", c.nnz, D.ptr(c.indices), D.ptr(self.colsum), D.ptr(self.s_entry),
                      D.stream())
            self._s_entry_dirty = False

    def _repack(self):
        c = self.csr
        _lib.call("xc_bca_pack_rows_from_colsum", c.nnz, D.ptr(c.indices), D.ptr(c.data), D.ptr(self.sel),
                  D.ptr(self.colsum), D.ptr(self.hot_slot), D.ptr(self.packed), D.stream())
        self._pack_dirty = False

    def sync_column_sums(self):
        """After a greedy sweep: colsum holds this rank's rows only -> all ranks;
        then expand it for the following (non-greedy) sweeps."""
        if self.comm is not None:
            self.comm.all_reduce(self.colsum)
        self._expand_colsum()

    def recompute_utility_sum(self, n_norm_utility: int) -> float:
        """calculate_confusion_matrix(y_proba, y_pred) + _calculate_utility
        (block_coordinate.py:430-445 / :465-476): tp, fp of the current prediction from
        scratch (one all-reduce of 2m+1 doubles when sharded), then the sum over labels
        of the binary metric on the normalised entries.  A sweep that visited every
        row has already accumulated them (xc_bca_sweep_csr `acc`); otherwise one pass
        over the prediction does.  Blocks on the result."""
        if not self._acc_filled:
            if self._scatter_sum(self.csr.n * self.k, self.pred_idx, self.pred_eta, self.acc, pair=True):
                self.acc[2 * self.csr.m] = 0.0
            else:
                self.acc.zero_()
                _lib.call("xc_bca_accumulate_pred", self.csr.n * self.k, D.ptr(self.pred_idx), D.ptr(self.pred_eta),
                          self.csr.code, D.ptr(self.acc), D.stream())
        if self.comm is not None:
            self.comm.all_reduce(self.acc)
        self._acc_filled = False
        return self.utility_sum(n_norm_utility, commit=True)

    def utility_sum(self, n_norm_utility: int, commit: bool = False, n_counted: Optional[float] = None,
                    skip_tn: Optional[bool] = None) -> float:
        out, extra = ctypes.c_double(0.0), ctypes.c_double(0.0)
        _lib.call("xc_bca_plan_boundary", self._plan_handle(), int(n_norm_utility),
                  float(self.n_total if n_counted is None else n_counted), int(bool(commit)),
                  int(self.skip_tn if skip_tn is None else skip_tn), ctypes.byref(out), ctypes.byref(extra),
                  D.stream())
        if commit:
            self._changed_last = int(round(extra.value))
        return out.value

    # -- one sweep ------------------------------------------------------------------
    def sweep(self, order: Optional[torch.Tensor], n_order: int, n_waves: int, greedy: bool = False):
        """block_coordinate.py:448-463 over `order` (int32 row ids on the GPU, or
        None for 0..n_order-1).  When every row is visited the kernel also leaves the
        new prediction's from-scratch statistics in `acc` (zero on entry: the commit
        kernel clears it)."""
        c = self.csr
        full = n_order >= c.n
        if self.deterministic:
            n_waves = 1
        if int(n_waves) == 1 and self.can_sweep_ordered(n_order, greedy):
            # the reference's sequence, thousands of rows in flight (csrc/xc_bca_ord.hip); same predictions as the
            # one-wavefront sweep below
            self.sweep_ordered(order, n_order)
            return
        if not full:
            self.changed.zero_()
        # the packed stream serves the concurrent sweeps (its s is float32); greedy and one-wavefront (exact)
        # sweeps read the separate streams and leave the packed copy's membership bits stale
        use_packed = self.packed is not None and not greedy and int(n_waves) > 1
        if use_packed and self._pack_dirty:
            self._repack()
        if not use_packed and not greedy:
            self._ensure_s_entry()
        _lib.call("xc_bca_plan_sweep", self._plan_handle(), D.ptr(order), int(n_order), D.ptr(self.orphans),
                  int(bool(greedy)), int(n_waves), int(full), int(use_packed),
                  None if full else D.ptr(self.changed), D.stream())
        if not full and self.shadow is not None and int(n_waves) == 1 and not greedy:
            self.shadow.copy_(self.tpfp)     # an exact partial sweep updated the float64 records only
        if not use_packed:
            self._pack_dirty = True  # sel was rewritten without touching the packed copy
        self._acc_filled = full
        self._partial_sweep = not full
        # every row was visited: no orphan is left in any prediction
        if full:
            self.orphans = None

    # -- row shards, host-paced sweeps: exchanges inside the sweep ------------------------------------------
    supports_segments = True

    def sweep_segments(self, order: Optional[torch.Tensor], n_order: int, n_waves: int, n_seg: int):
        """One host-paced sweep of a row shard cut into `n_seg` parts of the order; between two parts every rank
        publishes what ITS rows changed in the float64 records since the last exchange and takes in the other ranks'
        changes (distributed.exchange_changes: one all-reduce of 16 bytes per label).  The exact (ordered /
        one-wavefront) sweeps of a sharded run go through here -- one exchange per sweep does not contract with many
        shards."""
        from .distributed import exchange_changes
        n = int(n_order)
        if order is None:
            order = torch.arange(n, dtype=torch.int32, device=self.dev)
        if getattr(self, "_seg_base", None) is None:
            self._seg_base = torch.empty_like(self.tpfp)
            self._seg_buf = torch.empty_like(self.tpfp)
        self._seg_base.copy_(self.tpfp)
        bounds = [n * s // n_seg for s in range(n_seg + 1)]
        total = 0
        for s in range(n_seg):
            cnt = bounds[s + 1] - bounds[s]
            if cnt > 0:
                self.sweep(order[bounds[s]:bounds[s + 1]], cnt, n_waves)
                total += int(self.changed.item())
            if s < n_seg - 1:
                exchange_changes(self.comm, self.tpfp, self._seg_base, self._seg_buf)
                if self.shadow is not None:
                    self.shadow.copy_(self.tpfp)
        self.changed.fill_(total)
        self._acc_filled = False
        self._partial_sweep = True
        self.orphans = None if n >= self.csr.n else self.orphans

    # -- ordered parallel sweep (csrc/xc_bca_ord.hip) ------------------------------------------------------
    def can_sweep_ordered(self, n_order: int, greedy: bool) -> bool:
        """The reference's sequential semantics with a window of rows in flight: rows of at most 256 stored entries
        (longer rows and greedy first sweeps take the one-wavefront sweep).  XCOLUMNS_BCA_ORDERED=0 /
        `ordered=False` disables."""
        # float32 scores only: their sums in float64 are exact, so "committed record + the earlier rows' changes" is the
        # number the sequential sweep holds whatever the order of the additions; with float64 scores every addition rounds,
        # the statistics differ from the sequential ones in the last bit, and ill-conditioned gains (tn-based metrics with
        # skip_tn) turn that into other decisions (the exact fuzz: 1 case in ~100): those take the one-wavefront sweep
        # (`bca_ordered=True` forces it for float64 scores too)
        return (self.ordered and not greedy and n_order > 1 and self.csr.n > 0 and int(self.csr.max_row_nnz) <= 256
                and (self.csr.data.dtype == torch.float32 or getattr(self, "ordered_forced", False)))

    def _ordered_setup(self, orphans: Optional[torch.Tensor]):
        """Label directory of the ordered sweep: which labels get a dense table (a window holds many rows that store
        them) and how many change-list entries the others can hold (a row changes a label only if it stores it, or
        holds it as an orphan of a foreign initial prediction)."""
        c = self.csr
        dev = self.dev
        if getattr(self, "_ord_counts", None) is None:
            wg, win = ctypes.c_int(0), ctypes.c_int(0)
            _lib.call("xc_bca_ord_window", ctypes.byref(wg), ctypes.byref(win))
            env_wg = os.environ.get("XCOLUMNS_BCA_ORD_WORKGROUPS")
            # Two CUs per XCD stay free (workgroups go to the XCDs round robin, one per CU): the kernel's grid barrier needs
            # all its workgroups resident, and the visiting-order generator's rejection walk (csrc/xc_order_dev.hip: a
            # workgroup per batch, each waiting for lower-numbered ones) must always find a CU for its lowest unfinished
            # workgroup -- with every CU taken by a workgroup that spins at the barrier, a walk that is only partly resident
            # and this kernel would wait for each other until their spin limits.  With CUs to spare the two overlap freely
            # (fencing them apart with events cost 0.3 ms per sweep at 150 K rows).
            spare = int(os.environ.get("XCOLUMNS_BCA_ORD_SPARE_CUS", "16"))
            dflt = wg.value - spare if wg.value >= 8 * spare else wg.value
            self._ord_wg = max(1, min(int(env_wg), wg.value)) if env_wg else dflt
            self._ord_waves = max(1, win.value // max(1, wg.value))            # wavefronts per workgroup
            self._ord_max_rpw = 4 if self._ord_waves >= 16 else 8              # rows x candidates per lane a wavefront's registers hold
            # rows per wavefront: rows x candidates per lane <= 4 stay in registers for a window's iterations; small
            # matrices take fewer (a window should not swallow the whole sweep: its first rows would wait for nothing)
            rpw = int(os.environ.get("XCOLUMNS_BCA_ORD_ROWS", "2"))   # measured: profiles/r03_ordered_builds_rows.txt
            ch = 1 if c.max_row_nnz <= 64 else (2 if c.max_row_nnz <= 128 else 4)
            rpw = max(1, min(8 if rpw >= 8 else (4 if rpw >= 4 else (2 if rpw >= 2 else 1)), self._ord_max_rpw // ch))
            while rpw > 1 and self._ord_wg * self._ord_waves * rpw * 2 > max(1, c.n):
                rpw //= 2
            self._ord_rpw = rpw
            self._ord_counts = torch.bincount(c.indices, minlength=c.m)
            self._ord_epoch = 1 << 20
            self._ord_dirs = {}
        key = orphans is not None
        if key in self._ord_dirs and not key:
            return self._ord_dirs[key]
        W = self._ord_wg * self._ord_waves * self._ord_rpw
        counts = self._ord_counts
        if orphans is not None:
            counts = counts + torch.bincount(orphans[orphans >= 0], minlength=c.m)
        readers = counts.to(torch.float64) * (float(W) / float(max(1, c.n)))     # expected rows of a window that store the label
        # dense tables: a list of L entries read by L rows is L^2 entry reads, a table W slots scanned once -- but ANY table
        # costs the window a second barrier and a scan per iteration (~25 us), and the lists win far beyond the point where
        # L^2 = W: measured at 100 K x 30 K and 150 K x 670 K with Zipf(1) labels, top-k and random starts
        # (profiles/r03_ordered_hot_threshold.txt), first sweep 6.1 / 8.2 ms with tables from 48 readers on (255 tables), 4.2 /
        # 5.5 ms from 400 (81 / 56 tables), 3.2 / 4.0 ms with none.  Tables remain for labels that a fifth of a window's rows
        # store (1600 readers of 7680): there a window in which most of those rows change the label would be 10^6-10^7 entry
        # reads per iteration.
        hot_min = float(os.environ.get("XCOLUMNS_BCA_ORD_HOT_READERS", "1600"))
        n_hot = int(min(255, int((readers >= hot_min).sum().item())))
        if os.environ.get("XCOLUMNS_BCA_ORD_HOT", "1") == "0":
            n_hot = 0
        cap = torch.minimum(counts, 8 + 2 * torch.ceil(readers).to(torch.int64))
        hot_labels = None
        if n_hot > 0:
            hot_labels = torch.topk(counts, n_hot).indices.to(torch.int32)
            cap[hot_labels.long()] = 0
        off = torch.cumsum(cap, 0) - cap
        total_cap = int(cap.sum().item())
        if total_cap >= (1 << 31):
            raise ValueError("xcolumns_amd: the ordered sweep's change lists exceed 2^31 entries")
        lab_dir = torch.stack([off, cap], dim=1).to(torch.int32)
        if n_hot > 0:
            lab_dir[hot_labels.long(), 0] = -(torch.arange(n_hot, device=dev, dtype=torch.int32) + 1)
        nbytes = ctypes.c_int64(0)
        _lib.call("xc_bca_ord_workspace_bytes", c.m, total_cap, n_hot, W, ctypes.byref(nbytes))
        ws = torch.zeros(int(nbytes.value), dtype=torch.uint8, device=dev)
        d = {"lab_dir": lab_dir.contiguous(), "hot_labels": hot_labels, "n_hot": n_hot, "total_cap": total_cap, "ws": ws}
        self._ord_dirs[key] = d
        return d

    def sweep_ordered(self, order: Optional[torch.Tensor], n_order: int):
        """One full sweep with the reference's visiting-order semantics (block_coordinate.py:448-463) and a window of
        rows in flight (xc_bca_ord_sweep).  Updates the prediction and the float64 records; the boundary statistics
        are rebuilt from the prediction afterwards (recompute_utility_sum), as the reference does (:465-467)."""
        c = self.csr
        self._ensure_s_entry()
        d = self._ordered_setup(self.orphans)
        self.changed.zero_()
        status = (ctypes.c_int64 * 8)()
        if self._ord_epoch >= (1 << 31) - (1 << 21):      # the lists' iteration tags wrap: start over on cleared lists
            for dd in self._ord_dirs.values():
                dd["ws"].zero_()
            self._ord_epoch = 1 << 20
        _lib.call("xc_bca_ord_sweep", D.ptr(d["ws"]), int(n_order), D.ptr(order), self.n_total, D.ptr(c.indptr),
                  D.ptr(c.indices), D.ptr(c.data), c.code, int(c.max_row_nnz), D.ptr(self.pred_idx), D.ptr(self.pred_eta),
                  D.ptr(self.sel), D.ptr(self.orphans), self.k, c.m, D.ptr(self.tpfp), D.ptr(self.s_entry), D.ptr(d["lab_dir"]),
                  d["total_cap"], D.ptr(d["hot_labels"]), d["n_hot"], self._ord_wg, self._ord_rpw, ctypes.byref(self.gain_metric),
                  int(self.maximize), int(self.skip_tn), ctypes.c_uint32(self._ord_epoch), D.ptr(self.changed), status,
                  D.stream())
        self._ord_epoch += 1 << 20
        done, err = int(status[0]), int(status[1])
        self.ordered_stats = {"iterations": int(status[2]), "windows": int(status[3]), "done": done, "error": err,
                              "barrier_us": status[4] / 100.0, "kernel_us": status[5] / 100.0, "rows_per_wave": int(status[6]),
                              "window": self._ord_wg * self._ord_waves * int(status[6]), "n_hot": d["n_hot"]}
        if err in (1, 3):
            # a change list overflowed (more rows of one window changed a label than its list holds), or a window did not
            # settle within the iteration limit: the committed prefix stands, the one-wavefront sweep walks the rest of the
            # order -- the same sweep, exactly
            d["ws"].zero_()
            rest = (order if order is not None else torch.arange(n_order, dtype=torch.int32, device=self.dev))[done:]
            _lib.call("xc_bca_plan_sweep", self._plan_handle(), D.ptr(rest), int(n_order - done), D.ptr(self.orphans), 0, 1, 0,
                      0, D.ptr(self.changed), D.stream())
        elif err != 0:
            d["ws"].zero_()
            raise RuntimeError(f"ordered BCA sweep failed (status {err}: barrier timeout) after {done} rows")
        if self.orphans is not None:
            self._ord_dirs.pop(True, None)        # the directory with the orphans' allowance served its one sweep
        self._pack_dirty = True          # sel was rewritten without touching the packed copy
        self._acc_filled = False         # the boundary statistics come from the prediction (a from-scratch pass)
        self._partial_sweep = True       # rows_changed() reads the kernel's counter
        if n_order >= c.n:
            self.orphans = None          # every row was visited: no orphan is left in any prediction

    # -- the sweep loop without a host round trip per iteration (include/xcolumns_amd.h) --------
    def can_pipeline(self, n_order: int) -> bool:
        return (self.orphans is None and n_order >= self.csr.n and not self.deterministic
                and os.environ.get("XCOLUMNS_BCA_PIPELINE", "1") != "0")

    def pipeline_begin(self, old_utility_sum: float, tolerance: float, divisor: float, maximize: bool,
                       policy: "WavePolicy", first_waves: int):
        """Arm the device-side stopping rule and wavefront policy."""
        if getattr(self, "_ctrl", None) is None:
            self._ctrl = torch.zeros(_lib.XC_CTRL_SIZE, dtype=torch.float64, device=self.dev)
            n_ring = _lib.XC_CTRL_RING_STRIDE * _lib.XC_CTRL_RING_SLOTS
            ring = ctypes.c_void_p()
            _lib.call("xc_host_alloc_pinned", ctypes.byref(ring), n_ring * 8)
            self._ring_ptr = ring
            self._ring = (ctypes.c_double * n_ring).from_address(ring.value)
            for i in range(n_ring):
                self._ring[i] = -1.0
            self._pipe_seq = 0.0      # grows by one per boundary over the engine's lifetime
            self._seq_of = {}
        num, world, min_w, max_w, fixed = policy.device_params()
        if self._delta_sharded():
            # the statistics every rank holds now (all-reduced, committed): the base of the sweeps' changes
            if getattr(self, "_tpfp_base", None) is None:
                self._tpfp_base = torch.empty_like(self.tpfp)
                self._delta_buf = torch.empty(2 * self.csr.m + 1, dtype=torch.float64, device=self.dev)
            self._tpfp_base.copy_(self.tpfp)
        self._changed_known = None
        self._pipe_max_waves = max(2, max_w)
        _lib.call("xc_bca_pipeline_begin", D.ptr(self._ctrl), float(old_utility_sum), float(tolerance), float(divisor),
                  int(bool(maximize)), float(num), int(world), int(min_w), int(max_w), int(fixed),
                  int(max(1, min(first_waves, max_w))), int(getattr(policy, "sequential_below", 0)), D.stream())

    def _delta_sharded(self) -> bool:
        """Row shards whose pipelined sweeps leave their changes in the float64 records (xc_bca_set_acc_delta)
        instead of rebuilding `acc` from scratch."""
        return self.comm is not None and bool(_lib.load().xc_bca_plan_delta(self._plan_handle()))

    def pipeline_step(self, order: Optional[torch.Tensor], j: int, n_norm_utility: int):
        """Enqueue sweep j and its boundary; returns at once.  The sweep runs only if the rule has
        not fired at an earlier boundary."""
        if self.packed is not None and self._pack_dirty:
            self._repack()
        use_packed = self.packed is not None
        if not use_packed:
            self._ensure_s_entry()
        n = self.csr.n
        segments = 1
        if self.comm is not None and self.shadow is not None:
            from .distributed import exchange_changes, exchanges_for_sweep
            # the host runs one boundary behind the GPU: the newest changed-row count it has is sweep j - 2's
            # (identical on every rank -- it comes out of the all-reduced statistics -- so all ranks agree)
            segments = exchanges_for_sweep(self.exchanges, self._changed_known, self.n_total, self.comm.world)
        self.exchanges_used = getattr(self, "exchanges_used", [])
        self.exchanges_used.append(segments)
        if segments <= 1:
            _lib.call("xc_bca_plan_sweep_pipelined", self._plan_handle(), D.ptr(order), 0, n, int(use_packed),
                      int(self._pipe_max_waves), D.ptr(self._ctrl), D.stream())
        else:
            # Sharded rows, several exchanges per sweep (XCOLUMNS_BCA_EXCHANGES / bca_exchanges): the
            # order is walked in `segments` parts; between two parts every rank publishes what its rows
            # changed in the float32 records since the last exchange and takes in the others' changes,
            # so a rank misses the other ranks' updates of 1/segments of a sweep instead of a whole one.
            if order is None:
                order = torch.arange(n, dtype=torch.int32, device=self.dev)
            if getattr(self, "_snap", None) is None:
                self._snap = torch.empty_like(self.shadow)
                self._xbuf = torch.empty_like(self.shadow)
                self._xmine = torch.empty_like(self.shadow)
            self._snap.copy_(self.shadow)
            # Overlapped form (default; XCOLUMNS_BCA_EXCHANGE_OVERLAP=0 = the blocking form): twice as many parts,
            # and what the ranks publish at the end of part p is folded in at the end of part p + 1 -- the
            # all-reduce runs beside part p + 1 instead of in front of it, and one fused element-wise pass
            # (xc_bca_exchange_step) replaces the three of the blocking form.  A rank then misses the others' updates
            # of 1/2 .. 1 blocking segments (blocking: 0 .. 1): about the same staleness, no waiting.
            overlap = (os.environ.get("XCOLUMNS_BCA_EXCHANGE_OVERLAP", "1") != "0" and hasattr(self.comm, "all_reduce_async"))
            parts = min(2 * segments, 16) if overlap else segments
            # A part is its own launch (ramp-up and drain ~10 us): a small shard is not cut finer than 16 K rows per
            # part -- then in the blocking form (few parts, the waiting is cheap), and never below the number the
            # iteration needs to contract (distributed.min_exchanges).  From the GLOBAL row count: every rank must cut
            # its sweep into the same number of parts (the collectives pair up) and shard sizes differ.
            from .distributed import min_exchanges
            fit = max(1, (self.n_total // self.comm.world) // 16384)
            if parts > fit:
                overlap = False
                parts = max(min(segments, fit), min(segments, min_exchanges(self.comm.world)))
            self.exchanges_used[-1] = parts
            bounds = [n * s // parts for s in range(parts + 1)]
            _lib.call("xc_bca_time_span", int(parts))     # (bench.py's event pair, if one is pending, spans all the parts)
            pending = None
            for s in range(parts):
                _lib.call("xc_bca_plan_sweep_pipelined", self._plan_handle(), D.ptr(order), bounds[s],
                          bounds[s + 1] - bounds[s], int(use_packed), int(self._pipe_max_waves), D.ptr(self._ctrl),
                          D.stream())
                if s < parts - 1:
                    if not overlap:
                        exchange_changes(self.comm, self.shadow, self._snap, self._xbuf)
                        continue
                    if pending is not None:
                        pending.wait()
                    _lib.call("xc_bca_exchange_step", 2 * self.csr.m, D.ptr(self.shadow), D.ptr(self._snap), D.ptr(self._xbuf),
                              D.ptr(self._xmine), int(pending is not None), D.stream())
                    pending = self.comm.all_reduce_async(self._xbuf)
            if pending is not None:
                pending.wait()   # the last publication is superseded by the boundary's from-scratch statistics
        if self.comm is not None:
            if self._delta_sharded():
                # the sweep pushed this rank's changes into its float64 records: what the ranks exchange is
                # (records - the statistics agreed at the last boundary), exact in float64, same 2m + 1 doubles
                m2 = 2 * self.csr.m
                _lib.call("xc_bca_delta_pack", m2, D.ptr(self.tpfp), D.ptr(self._tpfp_base), D.ptr(self.acc[m2:]),
                          D.ptr(self._delta_buf), D.stream())
                self.comm.all_reduce(self._delta_buf)
                _lib.call("xc_bca_delta_unpack", m2, D.ptr(self.tpfp), D.ptr(self._tpfp_base), D.ptr(self.acc[m2:]),
                          D.ptr(self._delta_buf), D.stream())
            else:
                self.comm.all_reduce(self.acc)
        slot = j % _lib.XC_CTRL_RING_SLOTS
        self._pipe_seq += 1.0
        self._seq_of[slot] = self._pipe_seq
        _lib.call("xc_bca_plan_boundary_pipelined", self._plan_handle(), int(n_norm_utility), float(self.n_total),
                  int(self.skip_tn), D.ptr(self._ctrl), slot, self._ring_ptr, self._pipe_seq, D.stream())
        self._acc_filled = False
        self._partial_sweep = False

    def pipeline_result(self, j: int):
        """(utility sum, rows changed, wavefronts used, flag) of boundary j; flag 0 = continue, 1 = the stopping rule
        fired there, 2 = the step did not run, 3 = the loop paused there: the policy wants the NEXT sweep exact
        (WavePolicy.sequential_below), it is the host's to run.  Blocks until it is known."""
        slot = j % _lib.XC_CTRL_RING_SLOTS
        _lib.call("xc_bca_ring_wait", self._ring_ptr, slot, self._seq_of[slot], 600000.0, D.stream())
        r, o = self._ring, _lib.XC_CTRL_RING_STRIDE * slot
        total, changed, waves, flag = r[o], r[o + 1], r[o + 2], r[o + 3]
        if flag != 2.0:
            self._changed_last = int(round(changed))
            self._changed_known = self._changed_last
        return total, int(round(changed)), int(waves), int(flag)

    def close(self):
        """Release what the engine holds outside torch's allocator: the C-side plan and the pinned
        result ring (after the stream has drained: the boundary kernels write into it)."""
        self._drop_plan()
        if getattr(self, "_ring_ptr", None) is not None:
            torch.cuda.synchronize()
            _lib.call("xc_host_free_pinned", self._ring_ptr)
            self._ring_ptr = None
            self._ring = None
            self._ctrl = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset_changed(self):
        """Kept for engine-interface compatibility: counters are reset by the kernels."""
        self._partial_sweep = False

    def rows_changed(self) -> int:
        """Rows whose prediction changed in the last sweep, over all ranks (read after
        recompute_utility_sum: the count travels with the statistics)."""
        if getattr(self, "_partial_sweep", False):
            if self.comm is not None:
                self.comm.all_reduce(self.changed)
            return int(self.changed.item())
        return self._changed_last

    def confusion_vectors(self):
        """(tp, fp, fn, tn) float64 tensors on the GPU, the reference's four vectors."""
        m = self.csr.m
        out = torch.empty((4, m), dtype=torch.float64, device=self.dev)
        _lib.call("xc_bca_state_unpack", m, D.ptr(self.tpfp), D.ptr(self.colsum), float(self.n_total), int(self.skip_tn),
                  D.ptr(out[0]), D.ptr(out[1]), D.ptr(out[2]), D.ptr(out[3]), D.stream())
        return out


_ORDER_PREFETCH_ROWS = 50_000   # below this a sweep's order is cheaper to make in line than to hand over
_ORDER_PREFETCH_DEPTH = 3        # orders generated ahead of the sweep that consumes them


class _OrderWorkers:
    """Two persistent host threads that prepare visiting orders ahead of the sweeps: one generates the swap partners
    (the sequential PCG64 arithmetic), the other applies the swaps and copies the order to the GPU on a side stream.
    Persistent because a thread's first HIP calls (device context, stream, pinned allocation) cost tens of
    milliseconds -- threads started per call made the call 2-4x slower and erratic (measured: 40-175 ms against 27-41
    ms in line at 1 M rows; 23 ms with the persistent pair).  One run at a time uses them; a concurrent caller simply
    generates its orders in line."""

    _lock = None
    _busy = False
    _queues = None
    _resources = {}     # copy thread only: (device index) -> side stream; (device index, n) -> pinned slots

    @classmethod
    def acquire(cls) -> bool:
        import queue
        import threading
        if cls._lock is None:
            cls._lock = threading.Lock()
        with cls._lock:
            if cls._busy:
                return False
            if cls._queues is None:
                cls._queues = (queue.Queue(), queue.Queue())
                for q, name in zip(cls._queues, ("xcolumns-order-draws", "xcolumns-order-copy")):
                    threading.Thread(target=cls._serve, args=(q,), name=name, daemon=True).start()
            cls._busy = True
            return True

    @classmethod
    def release(cls) -> None:
        with cls._lock:
            cls._busy = False

    @staticmethod
    def _serve(q):
        while True:
            job, done = q.get()
            try:
                job()
            finally:
                done.set()

    @classmethod
    def submit(cls, which: int, job):
        import threading
        done = threading.Event()
        cls._queues[which].put((job, done))
        return done


_DEVICE_WALK_NS_PER_ROW = 0.25     # the device generator: 0.28 ms + 0.25 ns per row and order (csrc/xc_order_dev.hip; tools/order_walk_trace.py)
_DEVICE_ORDER_NS_FIXED = 280_000.0
_order_choice = None


def _orders_on_device(n: int = 1_000_000) -> bool:
    """Where numpy's visiting-order stream is generated for a matrix of n rows: XCOLUMNS_ORDER_DEVICE=1 / 0 forces the GPU /
    the host's two worker threads; by default whichever is faster on THIS machine for THIS size.  The device generator takes
    0.28 ms + 0.25 ns per row (its rejection walk is a chain of rounds whatever the size: 0.28 / 0.37 / 0.64 / 1.1 ms at 50 K
    / 100 K / 1 M / 3 M rows), the host's sequential walk 1.3 ns per row on a fast idle core and 4 ns and more on a slow or
    busy one (measured once per process on 256 K rows): small matrices on a fast host stay on the host (100 K rows: 10 sweeps
    in 2.4 ms against 6.1), large ones and slow hosts go to the GPU."""
    global _order_choice
    env = os.environ.get("XCOLUMNS_ORDER_DEVICE")
    if env is not None:
        return env != "0"
    if _order_choice is None:
        host_ns = float("inf")
        try:
            from .utils import Pcg64Shuffler
            if Pcg64Shuffler.usable():
                sh = Pcg64Shuffler(np.random.default_rng(0), 1 << 18)
                sh.apply(sh.draws())                       # first touch of the buffers
                t0 = _time.perf_counter()
                for _ in range(3):
                    js = sh.draws()
                draws = (_time.perf_counter() - t0) / 3    # the two halves run on two threads: the slower one paces
                t0 = _time.perf_counter()
                for _ in range(3):
                    sh.apply(js)
                swaps = (_time.perf_counter() - t0) / 3
                host_ns = max(draws, swaps) / (1 << 18) * 1e9
        except Exception:
            host_ns = float("inf")
        _order_choice = host_ns
    return _order_choice * float(n) > 0.9 * (_DEVICE_ORDER_NS_FIXED + _DEVICE_WALK_NS_PER_ROW * float(n))


class _OrderSource:
    """Visiting order per sweep (block_coordinate.py:413-419): the reference's stream --
    ``np.random.default_rng(seed)``, ONE array shuffled cumulatively, once per iteration.

    Large matrices (>= 50 K rows): the stream is generated ON the GPU (utils.DeviceNumpyOrders, csrc/xc_order_dev.hip:
    PCG64 outputs by jump-ahead, the masked rejection by one wavefront, the Fisher-Yates swaps resolved in parallel) --
    the same permutations, no host work per sweep; XCOLUMNS_ORDER_DEVICE=0 selects the host walk below.

    On the host the shuffle is sequential work (numpy: 8 ms for 1 M rows, 12x the sweep it feeds).  There it is
    (a) done by the library's own walk of numpy's stream (utils.Pcg64Shuffler: same permutation, 2.3 ms of draws +
    0.8 ms of swaps) and (b) taken off the critical path: the draws of sweep j + 2 and the swaps + upload of sweep
    j + 1 run on two persistent worker threads (:class:`_OrderWorkers`; the C routines release the GIL) while sweep j
    runs; ``next()`` makes the compute stream wait for the copy only.  Same stream, same orders -- the loop just
    stops waiting for them."""

    def __init__(self, n: int, seed, shuffle: bool, backend: str, dev, prefetch: Optional[bool] = None,
                 limit: Optional[int] = None):
        self.n, self.shuffle, self.backend, self.dev = n, shuffle, backend, dev
        # orders the run can ask for at most (max_iters): nothing is generated ahead beyond them, so a run that uses
        # them all finds nothing in flight when it closes the source (3 ms per call at 1 M rows otherwise)
        self.limit = (1 << 62) if limit is None else max(0, int(limit))
        if backend not in ("numpy", "device"):
            raise ValueError("order_backend must be 'numpy' or 'device'")
        self.rng = np.random.default_rng(seed)       # :413
        self.order = np.arange(n)                     # :414, shuffled cumulatively
        self.gen = None
        self._threaded = False
        # the same walk and stream as rng.shuffle, 2-3x faster (utils.Pcg64Shuffler); numpy's own when it is not
        # the PCG64 generator this build was checked against, or for small orders
        self._fast = None
        if (backend == "numpy" and shuffle and n >= _ORDER_PREFETCH_ROWS
                and os.environ.get("XCOLUMNS_ORDER_FAST_SHUFFLE", "1") != "0"):
            from .utils import Pcg64Shuffler
            if Pcg64Shuffler.usable():
                self._fast = Pcg64Shuffler(self.rng, n)
        # numpy's stream generated ON the GPU (csrc/xc_order_dev.hip, utils.DeviceNumpyOrders): no host work per sweep
        self._devgen = None
        if backend == "numpy" and shuffle and n >= _ORDER_PREFETCH_ROWS and _orders_on_device(n):
            from .utils import DeviceNumpyOrders
            if DeviceNumpyOrders.usable(dev):
                self._devgen = DeviceNumpyOrders(self.rng, n, dev, limit=self.limit)
                self._fast = None
        if backend == "device" and shuffle:
            self.gen = torch.Generator(device=dev)
            self.gen.manual_seed(int(seed) if seed is not None else int(self.rng.integers(2 ** 31)))
        if prefetch is None:
            prefetch = os.environ.get("XCOLUMNS_ORDER_PREFETCH", "1") != "0"
        if (self._devgen is None and backend == "numpy" and shuffle and prefetch and n >= _ORDER_PREFETCH_ROWS
                and _OrderWorkers.acquire()):
            import queue
            import threading
            self._threaded = True
            self._q = queue.Queue(maxsize=_ORDER_PREFETCH_DEPTH - 1)
            self._jq = queue.Queue(maxsize=2)
            self._stop = threading.Event()
            self._error = None
            self._trace = [] if os.environ.get("XCOLUMNS_ORDER_TRACE") == "1" else None   # per-stage host times
            self._jobs = []
            if self._fast is not None:
                self._jobs.append(_OrderWorkers.submit(0, self._produce_draws))
            self._jobs.append(_OrderWorkers.submit(1, self._produce))

    # Hand-over between the threads: blocking queue operations, woken by close() / a failing worker (which empty
    # the queues and push a None) rather than by polling a flag -- a 20 ms poll was the larger part of a call on
    # a 100 K-row matrix.  The long timeouts are a safety net only.
    def _put(self, q, item) -> bool:
        while not self._stop.is_set() and self._error is None:
            try:
                q.put(item, timeout=0.25)
                return True
            except Exception:
                continue
        return False

    def _get(self, q):
        while not self._stop.is_set() and self._error is None:
            try:
                return q.get(timeout=0.25)       # None: woken to stop
            except Exception:
                continue
        return None

    def _wake(self):
        """Unblock whoever waits on either queue: make room, then leave a None for the getters."""
        for q in (self._q, self._jq):
            try:
                while True:
                    q.get_nowait()
            except Exception:
                pass
            try:
                q.put_nowait(None)
            except Exception:
                pass

    def _produce_draws(self):
        try:
            made = 0
            while not self._stop.is_set() and made < self.limit:
                made += 1
                t0 = _time.perf_counter()
                js = self._fast.draws()
                if self._trace is not None:
                    self._trace.append(("draws", _time.perf_counter() - t0))
                if not self._put(self._jq, js):
                    break
        except BaseException as e:
            self._error = e
            self._wake()

    def _produce(self):
        try:
            torch.cuda.set_device(self.dev)
            res = _OrderWorkers._resources
            di = torch.device(self.dev).index or 0
            side = res.get(di)
            if side is None:
                side = res[di] = torch.cuda.Stream(device=self.dev, priority=-1)   # (a queue of its own: see utils.DeviceNumpyOrders)
            slots = _ORDER_PREFETCH_DEPTH + 1      # one being filled, DEPTH - 1 queued, one in the consumer's hands
            key = (di, self.n)
            if key not in res:
                for old in [k for k in res if isinstance(k, tuple)]:
                    del res[old]                  # one size at a time: the buffers of the previous matrix go
                res[key] = [torch.empty(self.n, dtype=torch.int32).pin_memory() for _ in range(slots)]
            pinned = res[key]
            done = [None] * slots
            i = 0
            while not self._stop.is_set() and i < self.limit:
                slot = i % slots
                if done[slot] is not None:
                    done[slot].synchronize()          # its previous copy has left the pinned buffer
                if self._fast is not None:
                    t0 = _time.perf_counter()
                    js = self._get(self._jq)
                    if js is None:
                        break
                    t1 = _time.perf_counter()
                    np.copyto(pinned[slot].numpy(), self._fast.apply(js))   # :418-419, the swaps of the int32 walk
                    if self._trace is not None:
                        self._trace.append(("wait_draws", t1 - t0))
                        self._trace.append(("apply+copy", _time.perf_counter() - t1))
                else:
                    self.rng.shuffle(self.order)      # :418-419 (GIL released)
                    np.copyto(pinned[slot].numpy(), self.order, casting="unsafe")
                with torch.cuda.stream(side):
                    d = pinned[slot].to(self.dev, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(side)
                done[slot] = ev
                i += 1
                if not self._put(self._q, (d, ev)):
                    break
            for ev in done:                           # the pinned buffers outlive this run: let their copies finish
                if ev is not None:
                    ev.synchronize()
        except BaseException as e:                    # surfaced by next()
            self._error = e
            self._wake()

    def next(self) -> Optional[torch.Tensor]:
        if not self.shuffle:
            return None
        if self._devgen is not None:
            return self._devgen.next()
        if self._threaded:
            item = self._get(self._q)
            if item is None:
                raise RuntimeError(f"visiting-order worker failed: {self._error!r}")
            d, ev = item
            torch.cuda.current_stream().wait_event(ev)
            d.record_stream(torch.cuda.current_stream())
            return d
        if self.backend == "numpy":
            if self._fast is not None:
                return torch.from_numpy(self._fast.shuffle().copy()).to(self.dev, non_blocking=True)
            self.rng.shuffle(self.order)              # :418-419
            return torch.from_numpy(self.order.astype(np.int32)).to(self.dev, non_blocking=True)
        return torch.randperm(self.n, generator=self.gen, device=self.dev, dtype=torch.int32)

    def close(self):
        """Stop the workers' jobs (orders generated ahead of an early stop are dropped) and hand the workers back."""
        if self._devgen is not None:
            gen, self._devgen = self._devgen, None
            gen.finish()       # raises if a shuffle on the device failed; leaves the generator where numpy would
        if self._threaded:
            self._stop.set()
            deadline = _time.monotonic() + 10.0
            for done in self._jobs:
                self._wake()
                while not done.wait(timeout=0.001) and _time.monotonic() < deadline:
                    self._wake()
            self._threaded = False
            _OrderWorkers.release()
            if self._trace:
                import sys
                by = {}
                for name, t in self._trace:
                    by.setdefault(name, []).append(t * 1e3)
                print("order workers, ms per order:", {k_: [round(x, 2) for x in v] for k_, v in by.items()}, file=sys.stderr)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------
# predict_using_bc_with_0approx
# ---------------------------------------------------------------------------

def run_bca_sweeps(eng, next_order: Callable, n_order: int, n_u: int, m: int, metric_aggregation: str,
                   maximize: bool, tolerance: float, max_iters: int, greedy: bool, policy, verbose: bool,
                   meta: Dict[str, Any]) -> None:
    """The sweep loop of predict_using_bc_with_0approx (block_coordinate.py:415-493)
    over an engine that holds this rank's rows: order -> (recompute, utility) ->
    sweep -> recompute, utility -> stopping rule.  `eng` is a :class:`BcaCsrEngine`
    (or anything with its methods: the multi-rank CPU tests inject a checker-backed
    one); with sharded rows every rank runs this loop and reaches the same decision
    because utilities come from all-reduced statistics.

    Sweeps the policy runs concurrently are handed to the GPU in one go (:func:`_run_pipelined`: stopping rule and
    wavefront policy on the device); sweeps it wants exact (the ordered parallel sweep / one wavefront) are paced
    here.  The device-side loop hands back ("pauses") when its rule asks for an exact sweep."""
    changed_prev = None
    new_utility = None
    new_utility_sum = None
    pending_order = None      # the order of a sweep the device-side loop enqueued but did not run (it paused before it)
    div = m if metric_aggregation == "mean" else 1
    j = 1
    while j <= max_iters:
        if (not greedy and not getattr(policy, "sequential", True) and hasattr(eng, "can_pipeline")
                and eng.can_pipeline(n_order) and policy.next(changed_prev) > 1):
            # the remaining sweeps are concurrent ones (until the rule says otherwise): hand the stopping rule to the GPU
            if new_utility_sum is None:
                if j == 1:
                    eng.reset_state(False)
                log_info("    Calculating expected confusion matrix ...", verbose)
                new_utility_sum = eng.recompute_utility_sum(n_u)
            paused = _run_pipelined(eng, next_order, n_u, div, maximize, tolerance, max_iters, j, new_utility_sum,
                                    changed_prev, policy, verbose, meta, pending_order)
            if paused is None:
                return
            j, pending_order, new_utility_sum, changed_prev = paused
            new_utility = new_utility_sum / div
            continue
        log_info(f"  Starting iteration {j}/{max_iters} ...", verbose)
        order, pending_order = (pending_order, None) if pending_order is not None else (next_order(), None)
        if j == 1:
            eng.reset_state(greedy)
        if greedy:
            # all four vectors are zeros, tn included (:423-427)
            old_utility = eng.utility_sum(n_u, n_counted=0.0, skip_tn=False)
        elif new_utility is None:
            log_info("    Calculating expected confusion matrix ...", verbose)
            old_utility = eng.recompute_utility_sum(n_u)
        else:
            # the end-of-sweep recompute of sweep j-1 IS the start-of-sweep one of j (:430, :465)
            old_utility = new_utility_sum
        if metric_aggregation == "mean":
            old_utility = old_utility / m

        log_info("    Doing block coordinate optimization steps ...", verbose)
        eng.reset_changed()
        n_waves = policy.next(changed_prev, greedy=True) if greedy else policy.next(changed_prev)
        n_seg = 1
        if not greedy and getattr(eng, "supports_segments", False) and getattr(eng, "comm", None) is not None:
            from .distributed import exchanges_for_sweep
            n_seg = exchanges_for_sweep(getattr(eng, "exchanges", 1), changed_prev, n_u, eng.comm.world)
        if n_seg > 1:
            eng.sweep_segments(order, n_order, n_waves, n_seg)   # S - 1 mid-sweep exchanges between row shards
        else:
            eng.sweep(order, n_order, n_waves, greedy=greedy)
        meta.setdefault("exchanges", []).append(n_seg)
        if greedy:
            eng.sync_column_sums()
        new_utility_sum = eng.recompute_utility_sum(n_u)
        changed_prev = eng.rows_changed()
        new_utility = new_utility_sum / m if metric_aggregation == "mean" else new_utility_sum

        greedy = False
        meta["iters"] = j
        meta["utilities"].append(new_utility)
        meta.setdefault("wavefronts", []).append(n_waves)
        meta.setdefault("rows_changed", []).append(changed_prev)
        log_info(f"    Iteration {j}/{max_iters} finished, expected metric value: {old_utility} -> {new_utility}", verbose)
        if (maximize and new_utility - old_utility < tolerance) or (not maximize and new_utility - old_utility > tolerance):
            log_info(f"  Stopping because improvement of expected metric value is smaller than {tolerance}", verbose)
            break
        j += 1


def _run_pipelined(eng, next_order: Callable, n_u: int, div: float, maximize: bool, tolerance: float, max_iters: int,
                   j0: int, old_sum: float, changed_prev, policy, verbose: bool, meta: Dict[str, Any], first_order=None):
    """Sweeps j0.. of the loop above with the stopping rule (block_coordinate.py:486-493) and the
    wavefront policy evaluated on the GPU: sweep j + 1 is enqueued before the utility of sweep j has
    reached the host; if the rule fired at boundary j it is a no-op.  Same utilities, same decision,
    same prediction as the host-paced loop -- the GPU just never waits for Python.

    Returns None when the run is over, or (j, order, utility sum, rows changed) when the device-side policy PAUSED after
    sweep j - 1: it wants sweep j exact; `order` is the visiting order that sweep was enqueued with (it did not run)."""
    eng.pipeline_begin(old_sum, tolerance, div, maximize, policy, policy.next(changed_prev))
    prev = [old_sum / div]
    last = {}

    def collect(j) -> int:
        total, changed, waves, flag = eng.pipeline_result(j)
        if flag == 2:
            return 2
        new_utility = total / div
        meta["iters"] = j
        meta["utilities"].append(new_utility)
        meta.setdefault("wavefronts", []).append(waves)
        meta.setdefault("rows_changed", []).append(changed)
        log_info(f"    Iteration {j}/{max_iters} finished ({waves} wavefronts, {changed} rows changed), expected "
                 f"metric value: {prev[0]} -> {new_utility}", verbose)
        prev[0] = new_utility
        last.update(total=total, changed=changed)
        if flag == 1:
            log_info(f"  Stopping because improvement of expected metric value is smaller than {tolerance}", verbose)
        return flag

    orders = {}
    for j in range(j0, max_iters + 1):
        orders[j] = first_order if (j == j0 and first_order is not None) else next_order()
        eng.pipeline_step(orders[j], j, n_u)
        if j > j0:
            flag = collect(j - 1)
            orders.pop(j - 1, None)
            if flag == 1:
                return None
            if flag == 3:      # paused after sweep j - 1: sweep j was enqueued with orders[j] and did nothing
                eng.pipeline_result(j)           # (its boundary reports "did not run": drain it before the host takes over)
                return j, orders[j], last["total"], last["changed"]
    flag = collect(max_iters)
    return None


def _initial_csr_indices(y_proba, init_y_pred, k: int, seed):
    """block_coordinate.py:28-51 for sparse input; None means "top" (done on the GPU).  Returns the
    k column ids per row as a numpy array (host input) or an int32 tensor (device-resident input)."""
    n, m = tuple(y_proba.shape)
    if isinstance(init_y_pred, str) and init_y_pred in ("random", "greedy"):
        dt = y_proba.dtype if isinstance(y_proba, (csr_matrix, D.DeviceCSR)) else D.numpy_dtype(y_proba.dtype)
        mat = random_at_k_csr((n, m), k, dtype=dt, seed=seed)
        return mat.indices
    if isinstance(init_y_pred, str) and init_y_pred == "top":
        return None
    if is_matrix(init_y_pred):
        if tuple(init_y_pred.shape) != (n, m):
            raise ValueError(f"init_y_pred must have shape (n, m) = ({n}, {m}), but has shape {tuple(init_y_pred.shape)}")
        if isinstance(y_proba, csr_matrix):
            if not isinstance(init_y_pred, csr_matrix):
                raise ValueError("init_y_pred must be a csr_matrix when y_proba is a csr_matrix")
            if not (np.diff(init_y_pred.indptr) == k).all():
                raise ValueError(
                    "on the GPU path init_y_pred must hold exactly k stored entries per row "
                    "(variable-length rows are not supported)")
            return init_y_pred.indices
        if not D.is_device_sparse(init_y_pred):
            raise ValueError("init_y_pred must be a DeviceCSR / torch sparse_csr tensor when y_proba is one")
        init = D.as_device_csr(init_y_pred)
        if n > 0 and not (init.min_row_nnz == init.max_row_nnz == k):
            raise ValueError(
                "on the GPU path init_y_pred must hold exactly k stored entries per row "
                "(variable-length rows are not supported)")
        return init.indices
    raise ValueError(
        "init_y_pred must be np.ndarray, Torch.tensor, csr_matrix or str in ['random', 'greedy', 'top'], "
        f"but has type {type(init_y_pred)}")


def _bc_csr(y_proba, gain_spec, utility_spec, k, metric_aggregation, n_u, maximize, tolerance,
            init_y_pred, max_iters, shuffle_order, skip_tn, seed, verbose, meta, bca_waves, order_backend,
            bca_parity=None, bca_deterministic=False, bca_ordered=None):
    """Sparse y_proba: a csr_matrix in host memory (uploaded here, result downloaded) or a matrix already
    resident in HBM -- DeviceCSR or torch sparse_csr tensor -- in which case nothing crosses PCIe but the
    visiting orders and the per-sweep utility."""
    n_rows, m = tuple(y_proba.shape)
    host = isinstance(y_proba, csr_matrix)
    if k == 0:
        raise NotImplementedError(
            "k=0 (no budget) with sparse y_proba is not implemented on the GPU path yet; "
            "pass a dense y_proba or k > 0")
    if host:
        row_nnz = np.diff(y_proba.indptr)
        if n_rows > 0 and row_nnz.min() < k:
            raise ValueError(
                f"every row of a sparse y_proba must store at least k={k} entries on the GPU path "
                f"(shortest row has {int(row_nnz.min())})")
    dev = D.require_gpu()
    # the host starts walking the visiting orders NOW: the first ones are ready by the time the matrix is uploaded,
    # packed and its initial prediction made (the walk is what bounds a call on a resident matrix: ~1.3 ms per
    # 1 M-row order behind a 0.5 ms sweep)
    orders = _OrderSource(n_u, seed, shuffle_order, order_backend, dev, limit=max_iters)
    try:
        csr = D.as_device_csr(y_proba, dev)
        eng = BcaCsrEngine(csr, k, gain_spec, utility_spec, maximize=maximize, skip_tn=skip_tn,
                           deterministic=bca_deterministic)
        if bca_ordered is not None:
            eng.ordered = bool(bca_ordered)
            eng.ordered_forced = bool(bca_ordered)

        log_info("  Initializing initial prediction ...", verbose)
        greedy = isinstance(init_y_pred, str) and init_y_pred == "greedy"
        init_idx = _initial_csr_indices(y_proba, init_y_pred, k, seed)
        if init_idx is None:
            eng.init_top()
        else:
            if isinstance(init_idx, torch.Tensor):
                init_dev = init_idx.to(device=dev, dtype=torch.int32).clone()
            else:
                init_dev = torch.from_numpy(np.ascontiguousarray(init_idx, dtype=np.int32)).to(dev)
            D.check_column_ids(init_dev, m, "init_y_pred")
            eng.init_indices(init_dev)

        if bca_deterministic:
            bca_waves = 1        # every sweep exact: the reference's sequence (ordered parallel sweep where it applies)
        if bca_waves is None and gain_spec.base == _lib.XC_M_PRECISION:
            # Macro precision tp / (tp + fp) jumps when a rarely predicted label gains or loses a row: rows in
            # flight that pick the same attractive label all take it, and the run ends 1e-4 .. 1e-3 BELOW the
            # sequential one even with two wavefronts (tests/studies/fuzz_concurrent.py).  Parity first: the
            # reference's sequential sweep unless the caller sets bca_waves.
            log_info("  macro precision: using the exact sweep (bca_waves=1: the reference's sequence, a window of rows in flight); "
                     "set bca_waves to trade parity for speed", verbose)
            bca_waves = 1
        # a random / foreign / greedy start changes every row in sweep 1 and keeps many rows moving for several
        # sweeps (measured 5e-5 .. 1.2e-4 at the top-k width): half the budget for the whole run
        # a random / foreign / greedy start -- and a minimisation from the top-k start, the worst point for it --
        # changes every row in sweep 1 and keeps many rows moving for several sweeps (measured 1.0-1.1e-5 in sweep 1
        # at half the top-k width): a quarter of the width for the whole run
        normalize_first = n_u >= n_rows
        calm = init_idx is None and maximize   # a minimisation starts from its worst point: every row changes
        parity = default_parity() if bca_parity is None else bca_parity
        policy = WavePolicy(n_u, fixed=bca_waves, k=k, first_changed=0.5 if calm else 1.0,
                            m=m, row_nnz=csr.nnz / max(1, n_rows), skewed=eng.skewed,
                            parity=parity, scale=1.0 if calm else 0.25, sweeps=max_iters,
                            first_sequential=not calm and not greedy and parity == "per_sweep" and normalize_first)
        run_bca_sweeps(eng, orders.next, n_u, n_u, m, metric_aggregation, maximize, tolerance, max_iters, greedy,
                       policy, verbose, meta)
    finally:
        orders.close()

    pred_idx = eng.pred_idx
    eng.close()
    if not host:
        # device-resident input -> device-resident prediction of the same kind; an explicit init_y_pred is
        # updated in place and returned, like the reference does with its matrices (:46, :285-287)
        if isinstance(init_y_pred, D.DeviceCSR):
            init_y_pred.indices.copy_(pred_idx)
            init_y_pred.forget_cached()      # column sums / ordering cached with the old column ids are stale
            return init_y_pred
        if D.is_torch_sparse_csr(init_y_pred):
            init_y_pred.col_indices().copy_(pred_idx.to(init_y_pred.col_indices().dtype))
            return init_y_pred
        return D.fixed_width_prediction(y_proba, pred_idx, k, n_rows, m)
    new_indices = pred_idx.cpu().numpy()
    if isinstance(init_y_pred, csr_matrix):
        # the reference updates an explicit init_y_pred in place and returns it (:46, :285-287)
        init_y_pred.indices[:] = new_indices.astype(init_y_pred.indices.dtype, copy=False)
        return init_y_pred
    out_indptr = (np.arange(n_rows + 1, dtype=np.int64) * k).astype(y_proba.indptr.dtype)
    return csr_matrix((np.ones(n_rows * k, dtype=y_proba.dtype),
                       new_indices.astype(y_proba.indices.dtype, copy=False), out_indptr), shape=(n_rows, m))


def _dense_confusion(y: torch.Tensor, y_pred: torch.Tensor, stats: torch.Tensor, skip_tn: bool):
    """calculate_confusion_matrix(y_proba, y_pred, normalize=False, skip_tn, float64)
    into stats = tp | fp | fn | tn (block_coordinate.py:430-436)."""
    n, m = y.shape
    stats.zero_()
    _lib.call("xc_confusion_dense", n, m, D.ptr(y), D.ptr(y_pred), D.dtype_code(y.dtype),
              D.ptr(stats[0]), D.ptr(stats[1]), D.ptr(stats[2]), D.stream())
    if skip_tn:
        stats[3].fill_(-1.0)
    else:
        stats[3] = -stats[0] - stats[1] - stats[2] + float(n)   # confusion_matrix.py:397


def _dense_utility(stats: torch.Tensor, n_u: int, metric_c, partials: torch.Tensor) -> float:
    m = stats.shape[1]
    _lib.call("xc_utility_vectors", m, int(n_u), D.ptr(stats), ctypes.byref(metric_c), D.ptr(partials), D.stream())
    out = ctypes.c_double(0.0)
    _lib.call("xc_utility_finish_host", D.ptr(partials), ctypes.byref(out), None, D.stream())
    return out.value


_DENSE_BLOCK_WAVES = 16       # a dense row is scored by one 1024-thread workgroup
_DENSE_MAX_LABELS = 8192      # xc_bca_sweep_dense_concurrent keeps <= 8 labels per thread in registers


def _bc_dense(y_proba, gain_spec, utility_spec, k, metric_aggregation, n_u, maximize, tolerance, init_y_pred,
              max_iters, shuffle_order, skip_tn, seed, verbose, meta, order_backend, bca_waves=None, bca_parity=None):
    n_rows, m = y_proba.shape
    if k < 0:
        raise ValueError("k must be >= 0")
    dev = D.require_gpu()
    is_torch = isinstance(y_proba, torch.Tensor)
    y_host = y_proba if is_torch else torch.from_numpy(np.ascontiguousarray(y_proba))
    D.dtype_code(y_host.dtype)
    y = y_host.to(dev).contiguous()

    log_info("  Initializing initial prediction ...", verbose)
    greedy = isinstance(init_y_pred, str) and init_y_pred == "greedy"
    init_is_matrix = False
    if isinstance(init_y_pred, str) and init_y_pred in ("random", "greedy"):
        y_pred = torch.from_numpy(random_at_k_np((n_rows, m), k, dtype=D.numpy_dtype(y.dtype), seed=seed)).to(dev)
    elif isinstance(init_y_pred, str) and init_y_pred == "top":
        y_pred = topk_dense_device(y, k, 0.0, False, y.dtype)
    elif is_matrix(init_y_pred):
        if tuple(init_y_pred.shape) != (n_rows, m):
            raise ValueError(f"init_y_pred must have shape (n, m) = ({n_rows}, {m}), but has shape {init_y_pred.shape}")
        if not is_dense(init_y_pred):
            raise ValueError("init_y_pred must be dense when y_proba is dense")
        init_is_matrix = True
        src = init_y_pred if isinstance(init_y_pred, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(init_y_pred))
        y_pred = src.to(device=dev, dtype=y.dtype).contiguous()
    else:
        raise ValueError(
            "init_y_pred must be np.ndarray, Torch.tensor, csr_matrix or str in ['random', 'greedy', 'top'], "
            f"but has type {type(init_y_pred)}")

    gain_c, util_c = gain_spec.to_c(), utility_spec.to_c()
    stats = torch.zeros((4, m), dtype=torch.float64, device=dev)
    work = torch.empty(m, dtype=torch.float64, device=dev)
    partials = torch.zeros(_lib.XC_UTILITY_PARTIALS + 1, dtype=torch.float64, device=dev)
    orders = _OrderSource(n_u, seed, shuffle_order, order_backend, dev, prefetch=False)
    # rows in flight: the CSR policy (one workgroup here = one "wavefront" there), bounded by the
    # workgroups the GPU holds; bca_waves=1 is the reference's sequential sweep
    if bca_waves is None and gain_spec.base == _lib.XC_M_PRECISION:
        bca_waves = 1   # macro precision is chaotic under any concurrency (see _bc_csr): sequential by default
    from_top = isinstance(init_y_pred, str) and init_y_pred == "top"
    # the dense sweep keeps the round-1 rule (validate-then-commit, no commit protocol): rows in flight as
    # a share of n, W = 4e-3 * n^2 / rows changed, half of it for a random or foreign start
    policy = WavePolicy(n_u, fixed=bca_waves, k=max(1, k), first_changed=0.5 if from_top else 1.0,
                        budget=(4e-3 if from_top else 2e-3) * (_FINAL_PARITY_FACTOR if bca_parity == "final" else 1.0),
                        parity=bca_parity)
    max_blocks = max(1, policy.cap // _DENSE_BLOCK_WAVES)
    changed = torch.zeros(1, dtype=torch.int64, device=dev)
    changed_prev = None
    for j in range(1, max_iters + 1):
        log_info(f"  Starting iteration {j}/{max_iters} ...", verbose)
        order = orders.next()
        if greedy:
            stats.zero_()                              # :423-427
        else:
            log_info("    Calculating expected confusion matrix ...", verbose)
            _dense_confusion(y, y_pred, stats, skip_tn)
        old_utility = _dense_utility(stats, n_u, util_c, partials)
        if metric_aggregation == "mean":
            old_utility /= m
        log_info("    Doing block coordinate optimization steps ...", verbose)
        n_blocks = min(policy.next(changed_prev), max_blocks)
        # k == 0 (no budget) flips many labels per row: rows in flight would miss too much of each
        # other (2.6e-4 in the first sweep at 6000 x 1500), so it stays sequential
        if greedy or n_blocks == 1 or m > _DENSE_MAX_LABELS or k == 0:
            _lib.call("xc_bca_sweep_dense", int(n_u), D.ptr(order), int(n_rows), int(m), D.ptr(y), D.ptr(y_pred),
                      D.dtype_code(y.dtype), int(k), D.ptr(stats), D.ptr(work), ctypes.byref(gain_c), int(maximize),
                      int(greedy), int(skip_tn), D.stream())
            changed_prev = None
        else:
            changed.zero_()
            _lib.call("xc_bca_sweep_dense_concurrent", int(n_u), D.ptr(order), int(n_rows), int(m), D.ptr(y),
                      D.ptr(y_pred), D.dtype_code(y.dtype), int(k), D.ptr(stats), ctypes.byref(gain_c), int(maximize),
                      int(skip_tn), int(n_blocks), D.ptr(changed), D.stream())
            changed_prev = int(changed.item())
        _dense_confusion(y, y_pred, stats, skip_tn)   # :465-467
        new_utility = _dense_utility(stats, n_u, util_c, partials)
        if metric_aggregation == "mean":
            new_utility /= m
        greedy = False
        meta["iters"] = j
        meta["utilities"].append(new_utility)
        log_info(f"    Iteration {j}/{max_iters} finished, expected metric value: {old_utility} -> {new_utility}", verbose)
        if (maximize and new_utility - old_utility < tolerance) or (not maximize and new_utility - old_utility > tolerance):
            log_info(f"  Stopping because improvement of expected metric value is smaller than {tolerance}", verbose)
            break

    if init_is_matrix:  # updated in place and returned (:46)
        if isinstance(init_y_pred, torch.Tensor):
            init_y_pred.copy_(y_pred.to(device=init_y_pred.device, dtype=init_y_pred.dtype))
        else:
            init_y_pred[...] = y_pred.cpu().numpy().astype(init_y_pred.dtype, copy=False)
        return init_y_pred
    if is_torch:
        return y_pred.to(y_proba.device)
    return y_pred.cpu().numpy()


def predict_using_bc_with_0approx(
    y_proba: Matrix,
    binary_metric_func: Union[Callable, List[Callable]],
    k: int,
    metric_aggregation: str = "mean",  # "mean" or "sum"
    normalize_conf_matrix: bool = True,
    metric_kwargs: Optional[Dict[str, Any]] = None,
    maximize: bool = True,
    tolerance: float = 1e-6,
    init_y_pred: Union[str, Matrix] = "top",  # "random", "top", "greedy", Matrix
    max_iters: int = 100,
    shuffle_order: bool = True,
    skip_tn: bool = False,
    return_meta: bool = False,
    seed: Optional[int] = None,
    verbose: bool = False,
    **kwargs,
) -> Union[Matrix, Tuple[Matrix, Dict[str, Any]]]:
    r"""Block coordinate ascent / descent with the 0-th order approximation of the
    expected-test-utility objective for a metric that decomposes over labels
    (block_coordinate.py:296-499).

    Sweeps over the rows of `y_proba`; for a row it takes the row's contribution
    out of the expected per-label confusion statistics, scores every candidate
    label by ``binary_metric(with the label predicted) - binary_metric(without)``,
    keeps the `k` best and puts the contribution back.  Stops when a sweep
    improves the utility by less than `tolerance` or after `max_iters` sweeps.

    `binary_metric_func` must be one of the ``binary_*_on_conf_matrix`` metrics
    (this package's or the reference's), a ``functools.partial`` of one, or a
    :class:`~xcolumns_amd.metrics.DeviceMetric`; see
    :func:`~xcolumns_amd.metrics.resolve_metric`.

    Returns a matrix of the same type, shape and dtype as `y_proba`; with
    `return_meta` also ``{"utilities": [...], "iters": int, "time": seconds}``.
    """
    log_info(
        f"Starting optimization of ETU metric using block coordinate {'ascent (maximization)' if maximize else 'descent (minimization)'} algorithm ...",
        verbose,
    )
    if isinstance(k, int) and k > 0:
        log_info(f"  Budget k: {k}", verbose)
    log_info(f"  Tolerance (stopping condition): {tolerance}, max iterations: {max_iters}", verbose)

    meta: Dict[str, Any] = {"utilities": [], "iters": 0, "time": time()}

    if not isinstance(k, int):
        raise ValueError("k must be an integer")
    if not (is_dense(y_proba) or is_sparse(y_proba)):
        raise ValueError("y_proba must be either np.ndarray, torch.Tensor, or csr_matrix")
    if metric_aggregation not in ("mean", "sum"):
        raise ValueError(
            f"Unsupported utility aggregation function: {metric_aggregation}, must be either 'mean' or 'sum'")

    gain_spec = resolve_metric(binary_metric_func, metric_kwargs)
    # _calculate_utility is called without metric_kwargs (:438-445, :469-476)
    utility_spec = resolve_metric(binary_metric_func, None)

    n, m = y_proba.shape
    if n == 0 or m == 0:
        raise ValueError(f"y_proba must have at least one row and one column, got shape {tuple(y_proba.shape)}")
    n_u = n if normalize_conf_matrix else 1   # :403-405 (also the length of the visiting order, :414)

    bca_waves = kwargs.pop("bca_waves", None)
    bca_parity = kwargs.pop("bca_parity", None)
    if bca_parity not in (None, "per_sweep", "final"):
        raise ValueError("bca_parity must be 'per_sweep' or 'final'")
    bca_diagnostics = kwargs.pop("bca_diagnostics", False)
    bca_deterministic = bool(kwargs.pop("bca_deterministic", os.environ.get("XCOLUMNS_BCA_DETERMINISTIC", "0") == "1"))
    order_backend = kwargs.pop("order_backend", os.environ.get("XCOLUMNS_ORDER_BACKEND", "numpy"))
    bca_ordered = kwargs.pop("bca_ordered", None)

    if is_sparse(y_proba):
        y_pred = _bc_csr(y_proba, gain_spec, utility_spec, k, metric_aggregation, n_u, maximize, tolerance,
                         init_y_pred, max_iters, shuffle_order, skip_tn, seed, verbose, meta, bca_waves,
                         order_backend, bca_parity, bca_deterministic, bca_ordered)
    else:
        y_pred = _bc_dense(y_proba, gain_spec, utility_spec, k, metric_aggregation, n_u, maximize, tolerance,
                           init_y_pred, max_iters, shuffle_order, skip_tn, seed, verbose, meta, order_backend,
                           bca_waves, bca_parity)

    if not bca_diagnostics:
        meta.pop("wavefronts", None)
        meta.pop("rows_changed", None)
        meta.pop("exchanges", None)
    if return_meta:
        meta["time"] = time() - meta["time"]
        return y_pred, meta
    return y_pred


# ---------------------------------------------------------------------------
# coverage (block_coordinate.py:507-701)
# ---------------------------------------------------------------------------

def predict_optimizing_coverage_using_bc(
    y_proba: Matrix,
    k: int,
    alpha: float = 1,
    tolerance: float = 1e-6,
    init_y_pred: Union[str, Matrix] = "top",
    max_iters: int = 100,
    shuffle_order: bool = True,
    return_meta: bool = False,
    seed: Optional[int] = None,
    verbose: bool = False,
    **kwargs,
):
    """Block coordinate ascent for the expected coverage@k -- the mean over labels of the probability
    of being predicted correctly at least once -- optionally mixed with instance precision@k
    (``alpha * coverage + (1 - alpha) * precision@k``); block_coordinate.py:600-701, same arguments,
    stopping rule (``new <= old + tolerance``) and ``meta``.

    The statistic is Ef_j = prod_i (1 - pred_ij * eta_ij); a sweep is ``xc_coverage_sweep_csr``, the
    from-scratch Ef of every sweep boundary ``xc_coverage_product``.  Sparse (csr_matrix) y_proba with at
    least k stored entries per row; the reference's dense branch relies on ``np.product`` (gone in
    numpy 2) and is not reproduced.  Extra keyword: ``bca_waves`` (default 1 = the reference's sequential
    sweep, bit-exact; W > 1 wavefronts in flight end within a few 1e-5 of it)."""
    log_info(f"Starting optimization of ETU coverage@{k} metric using block coordinate ascent algorithm ...", verbose)
    if not isinstance(k, int) or k <= 0:
        raise ValueError("k must be an integer > 0")
    if not isinstance(y_proba, csr_matrix):
        if is_dense(y_proba):
            raise NotImplementedError(
                "coverage BCA runs on sparse (csr_matrix) y_proba on the GPU; the reference's dense branch needs "
                "np.product, which numpy 2 removed, and is not reproduced")
        raise ValueError("y_proba must be either np.ndarray or csr_matrix")
    bca_waves = kwargs.pop("bca_waves", None)
    order_backend = kwargs.pop("order_backend", os.environ.get("XCOLUMNS_ORDER_BACKEND", "numpy"))
    n, m = y_proba.shape
    if n == 0 or m == 0:
        raise ValueError(f"y_proba must have at least one row and one column, got shape {tuple(y_proba.shape)}")
    if k > _lib.XC_MAX_K:
        raise ValueError(f"k must be in 1..{_lib.XC_MAX_K} for sparse y_proba on the GPU, got {k}")
    row_nnz = np.diff(y_proba.indptr)
    if row_nnz.min() < k:
        raise ValueError(f"every row of a sparse y_proba must store at least k={k} entries on the GPU path "
                         f"(shortest row has {int(row_nnz.min())})")
    meta: Dict[str, Any] = {"utilities": [], "iters": 0, "time": time()}
    if seed is not None:
        np.random.seed(seed)  # :626-627

    dev = D.require_gpu()
    csr = D.DeviceCSR.from_scipy(y_proba, dev)
    log_info("  Initializing starting prediction ...", verbose)
    greedy = isinstance(init_y_pred, str) and init_y_pred == "greedy"
    init_idx = _initial_csr_indices(y_proba, init_y_pred, k, seed)
    sel = torch.zeros(max(1, csr.nnz), dtype=torch.uint8, device=dev)
    if init_idx is None:
        pred_idx, _, pred_eta = topk_csr_device(csr, k, want_eta=True, out_sel=sel)
    else:
        pred_idx = torch.from_numpy(np.ascontiguousarray(init_idx, dtype=np.int32)).to(dev)
        D.check_column_ids(pred_idx, m, "init_y_pred")
        pred_eta = torch.empty(n * k, dtype=csr.data.dtype, device=dev)
        orphans = torch.empty(n * k, dtype=torch.int32, device=dev)
        _lib.call("xc_bca_gather_pred_eta", n, D.ptr(csr.indptr), D.ptr(csr.indices), D.ptr(csr.data), csr.code,
                  D.ptr(pred_idx), k, D.ptr(pred_eta), D.ptr(sel), D.ptr(orphans), D.stream())
    ef = torch.ones(m, dtype=torch.float64, device=dev)
    changed = torch.zeros(1, dtype=torch.int64, device=dev)

    def failure_probabilities():
        ef.fill_(1.0)
        _lib.call("xc_coverage_product", n * k, D.ptr(pred_idx), D.ptr(pred_eta), csr.code, D.ptr(ef), D.stream())

    def utility() -> float:  # _calculate_coverage_utility, :585-597
        cov = 1.0 - float(ef.mean().item())
        if alpha < 1:
            cov = alpha * cov + (1 - alpha) * float(pred_eta.to(torch.float64).sum().item()) / n / k
        return cov

    orders = _OrderSource(n, seed, shuffle_order, order_backend, dev, prefetch=False)
    # The multiplicative statistic reacts far more strongly to a row in flight than the additive ones,
    # and the coverage landscape has many nearly equivalent optima: runs with a few rows in flight end
    # 1-3e-5 from the sequential run even after 8 sweeps (20K x 30K Zipf; 1.6e-4 .. 2.8e-4 per sweep at the
    # BCA width).  The default is therefore the reference's sequential sweep (one wavefront, bit-exact);
    # bca_waves = W > 1 trades that for speed.
    policy = WavePolicy(n, fixed=bca_waves if bca_waves else 1, k=k)
    changed_prev = None
    new_cov = None
    for j in range(1, max_iters + 1):
        log_info(f"  Starting iteration {j}/{max_iters} ...", verbose)
        order = orders.next()
        if greedy:
            ef.fill_(1.0)  # :645-646
            old_cov = utility()
        elif new_cov is None:
            failure_probabilities()
            old_cov = utility()
        else:
            old_cov = new_cov  # the end-of-sweep recompute of sweep j-1 is the start-of-sweep one of j
        n_waves = policy.next(changed_prev, greedy=True) if greedy else policy.next(changed_prev)
        changed.zero_()
        _lib.call("xc_coverage_sweep_csr", n, D.ptr(order), D.ptr(csr.indptr), D.ptr(csr.indices), D.ptr(csr.data),
                  csr.code, int(csr.max_row_nnz), D.ptr(pred_idx), D.ptr(pred_eta), D.ptr(sel), k, D.ptr(ef),
                  float(alpha), int(greedy), int(n_waves), D.ptr(changed), D.stream())
        failure_probabilities()  # :659-665
        new_cov = utility()
        changed_prev = int(changed.item())
        greedy = False
        meta["iters"] = j
        meta["utilities"].append(new_cov)
        log_info(f"    Iteration {j}/{max_iters} finished, expected coverage: {old_cov} -> {new_cov}", verbose)
        if new_cov <= old_cov + tolerance:  # :676
            log_info(f"  Stopping because improvement of expected coverage is smaller than {tolerance}", verbose)
            break

    new_indices = pred_idx.cpu().numpy()
    if isinstance(init_y_pred, csr_matrix):  # updated in place and returned, like the reference (:46)
        init_y_pred.indices[:] = new_indices.astype(init_y_pred.indices.dtype, copy=False)
        y_pred = init_y_pred
    else:
        out_indptr = (np.arange(n + 1, dtype=np.int64) * k).astype(y_proba.indptr.dtype)
        y_pred = csr_matrix((np.ones(n * k, dtype=y_proba.dtype), new_indices.astype(y_proba.indices.dtype, copy=False),
                             out_indptr), shape=(n, m))
    if return_meta:
        meta["time"] = time() - meta["time"]
        return y_pred, meta
    return y_pred


# ---------------------------------------------------------------------------
# wrappers for specific metrics (block_coordinate.py:709-801)
# ---------------------------------------------------------------------------

def make_bc_wrapper(binary_metric_func: Callable, metric_name: str, maximize: bool = True,
                    metric_aggregation: str = "mean", skip_tn: bool = False, warn_k_eq_0: bool = False):
    """Factory of ``f(y_proba, k, **kwargs)`` wrappers around
    :func:`predict_using_bc_with_0approx` for one metric (block_coordinate.py:709-759)."""

    def predict_optimizing_metric_using_bc(y_proba: Matrix, k: int, **kwargs):
        if warn_k_eq_0 and k == 0:
            log_warning(f"Warning: k=0 results in degenerated solution for {metric_name}!")
        return predict_using_bc_with_0approx(y_proba, binary_metric_func, k, metric_aggregation=metric_aggregation,
                                             maximize=maximize, skip_tn=skip_tn, **kwargs)

    predict_optimizing_metric_using_bc.__doc__ = (
        f"Predict optimizing {metric_name} with block coordinate ascent: "
        f"``predict_using_bc_with_0approx(y_proba, {binary_metric_func.__name__}, k, "
        f"metric_aggregation={metric_aggregation!r}, maximize={maximize}, skip_tn={skip_tn}, **kwargs)``."
    )
    return add_kwargs_to_signature(predict_optimizing_metric_using_bc, predict_using_bc_with_0approx,
                                   skip=["metric_func", "metric_aggregation", "maximize", "skip_tn"])


predict_optimizing_macro_precision_using_bc = make_bc_wrapper(
    binary_precision_on_conf_matrix, "macro-averaged precision", metric_aggregation="mean", maximize=True,
    skip_tn=True, warn_k_eq_0=True)
predict_optimizing_macro_recall_using_bc = make_bc_wrapper(
    binary_recall_on_conf_matrix, "macro-averaged recall", metric_aggregation="mean", maximize=True,
    skip_tn=True, warn_k_eq_0=True)
predict_optimizing_macro_f1_score_using_bc = make_bc_wrapper(
    binary_f1_score_on_conf_matrix, "macro-averaged F1 score", metric_aggregation="mean", maximize=True,
    skip_tn=True)
predict_optimizing_macro_jaccard_score_using_bc = make_bc_wrapper(
    binary_jaccard_score_on_conf_matrix, "macro-averaged Jaccard score", maximize=True, skip_tn=True)
predict_optimizing_macro_balanced_accuracy_using_bc = make_bc_wrapper(
    binary_balanced_accuracy_on_conf_matrix, "macro-averaged balanced accuracy", maximize=True)
predict_optimizing_macro_hmean_using_bc = make_bc_wrapper(
    binary_hmean_on_conf_matrix, "macro-averaged H-mean", maximize=True)
predict_optimizing_macro_gmean_using_bc = make_bc_wrapper(
    binary_gmean_on_conf_matrix, "macro-averaged G-mean", maximize=True)


def predict_optimizing_instance_precision_using_bc(
    y_proba,
    k: int,
    tolerance: float = 1e-6,
    init_y_pred="random",
    max_iters: int = 100,
    shuffle_order: bool = True,
    verbose: bool = False,
    return_meta: bool = False,
    **kwargs,
):
    """BCA with instance precision, tp / k summed over labels
    (block_coordinate.py:804-835; note the "random" default initialisation)."""
    metric = DeviceMetric(MetricSpec(base=_lib.XC_M_PRECISION_AT_K, kf=float(k) if k else 1.0),
                          binary_precision_at_k_on_conf_matrix, "instance_precision_with_specific_k")
    return predict_using_bc_with_0approx(
        y_proba, binary_metric_func=metric, k=k, metric_aggregation="sum", tolerance=tolerance,
        init_y_pred=init_y_pred, max_iters=max_iters, shuffle_order=shuffle_order, verbose=verbose,
        return_meta=return_meta, **kwargs)


# ---------------------------------------------------------------------------
# mixed utilities (block_coordinate.py:848-1045):
#   (1 - alpha) * tp / k + alpha * binary_metric(...) / m,  summed over labels
# ---------------------------------------------------------------------------

def _make_mixed_wrapper(base_id: int, host_base: Callable, what: str):
    def predict_optimizing_mixed_using_bc(y_proba, k: int, alpha: float = 1, **kwargs):
        n, m = y_proba.shape
        metric = DeviceMetric(
            MetricSpec(base=base_id, mixed=True, kf=float(k) if k else 1.0, alpha=float(alpha), mf=float(m)),
            host_base, "mixed_utility_fn")
        return predict_using_bc_with_0approx(y_proba, binary_metric_func=metric, k=k, metric_aggregation="sum",
                                             skip_tn=True, **kwargs)

    predict_optimizing_mixed_using_bc.__doc__ = (
        f"BCA with a weighted average of instance precision and macro-averaged {what} "
        "as the target: ``(1 - alpha) * precision@k + alpha * macro metric``.")
    return predict_optimizing_mixed_using_bc


predict_optimizing_mixed_instance_precision_and_macro_precision_using_bc = _make_mixed_wrapper(
    _lib.XC_M_PRECISION, binary_precision_on_conf_matrix, "precision")
predict_optimizing_mixed_instance_precision_and_macro_recall_using_bc = _make_mixed_wrapper(
    _lib.XC_M_RECALL, binary_recall_on_conf_matrix, "recall")
predict_optimizing_mixed_instance_precision_and_macro_f1_score_using_bc = _make_mixed_wrapper(
    _lib.XC_M_FBETA, binary_f1_score_on_conf_matrix, "F1 score")
predict_optimizing_mixed_instance_precision_and_macro_balanced_accuracy_using_bc = _make_mixed_wrapper(
    _lib.XC_M_BALANCED_ACC, binary_balanced_accuracy_on_conf_matrix, "balanced accuracy")
predict_optimizing_mixed_instance_precision_and_macro_jaccard_score_using_bc = _make_mixed_wrapper(
    _lib.XC_M_JACCARD, binary_jaccard_score_on_conf_matrix, "Jaccard score")
predict_optimizing_mixed_instance_precision_and_macro_gmean_using_bc = _make_mixed_wrapper(
    _lib.XC_M_GMEAN, binary_gmean_on_conf_matrix, "G-mean")
predict_optimizing_mixed_instance_precision_and_macro_hmean_using_bc = _make_mixed_wrapper(
    _lib.XC_M_HMEAN, binary_hmean_on_conf_matrix, "H-mean")

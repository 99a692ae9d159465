#!/usr/bin/env python3
"""Randomised sweep of the CONCURRENT default (GPU box): medium random problems through
predict_using_bc_with_0approx against the sequential oracle; prints the worst per-sweep and the final
utility difference of every case and flags those over the bar (1e-5 at EVERY sweep, round 2).

    python tests/studies/fuzz_concurrent.py [cases] [first_seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref as oref  # noqa: E402  (checker)
import xcolumns_amd.block_coordinate as bc  # noqa: E402
from xcolumns_amd.synthetic import make_csr  # noqa: E402

METRICS = [("binary_precision_on_conf_matrix", oref.PRECISION, True), ("binary_recall_on_conf_matrix", oref.RECALL, True),
           ("binary_f1_score_on_conf_matrix", oref.FBETA, True), ("binary_jaccard_score_on_conf_matrix", oref.JACCARD, True),
           ("binary_balanced_accuracy_on_conf_matrix", oref.BALANCED_ACC, False), ("binary_hmean_on_conf_matrix", oref.HMEAN, False)]
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
over = 0
for seed in range(first, first + cases):
    rng = np.random.default_rng(seed)
    if os.environ.get("XC_FUZZ_SCALE") == "medium":     # between BASELINE configs[1] and the north-star size
        n = int(rng.integers(50_000, 250_000))
        m = int(rng.integers(20_000, 400_000))
        r = int(rng.integers(20, 110))
        k = int(rng.integers(1, 9))
    else:
        n = int(rng.integers(3000, 40000))
        m = int(rng.integers(200, 20000))
        r = int(rng.integers(10, 60))
        k = int(rng.integers(1, 9))
    zipf = bool(rng.random() < 0.4)
    dtype = np.float32 if rng.random() < 0.7 else np.float64
    name, base, skip_tn = METRICS[int(rng.integers(len(METRICS)))]
    init = str(rng.choice(["top", "top", "random", "greedy"]))
    Y = make_csr(n, max(m, r + 1), r, seed=seed, zipf=zipf, k=k, dtype=dtype)
    m = Y.shape[1]
    kw = dict(seed=int(rng.integers(100)), max_iters=4 if os.environ.get("XC_FUZZ_SCALE") == "medium" else 6, tolerance=-1.0,
              skip_tn=skip_tn, init_y_pred=init)
    metric = oref.make_metric(base, k=float(k), m=float(m))
    _, mo = oref.predict_using_bc_with_0approx(Y, metric, k, **kw)
    _, mg = bc.predict_using_bc_with_0approx(Y, getattr(bc, name), k, return_meta=True, bca_diagnostics=True, **kw)
    d = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
    flag = "OVER" if d.max() > 1e-5 else "ok"
    over += flag == "OVER"
    print(f"{flag:4s} seed={seed} n={n} m={m} r={r} k={k} zipf={zipf} {np.dtype(dtype).name} {name[7:-15]} init={init} "
          f"W={mg['wavefronts'][:3]} max {d.max():.1e} final {d[-1]:.1e}", flush=True)
print(f"{cases} cases: {over} over the bars", flush=True)

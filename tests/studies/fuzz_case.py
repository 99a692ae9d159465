#!/usr/bin/env python3
"""Round-2 study (GPU box): one case of tests/studies/fuzz_concurrent.py by its seed, per-sweep differences, a few runs,
optionally at fixed widths:  python tests/studies/fuzz_case.py SEED [W ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref as oref  # noqa: E402
import xcolumns_amd.block_coordinate as bc  # noqa: E402
from xcolumns_amd.synthetic import make_csr  # noqa: E402

METRICS = [("binary_precision_on_conf_matrix", oref.PRECISION, True), ("binary_recall_on_conf_matrix", oref.RECALL, True),
           ("binary_f1_score_on_conf_matrix", oref.FBETA, True), ("binary_jaccard_score_on_conf_matrix", oref.JACCARD, True),
           ("binary_balanced_accuracy_on_conf_matrix", oref.BALANCED_ACC, False), ("binary_hmean_on_conf_matrix", oref.HMEAN, False)]
oref.build()
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
n = int(rng.integers(3000, 40000)); m = int(rng.integers(200, 20000)); r = int(rng.integers(10, 60)); k = int(rng.integers(1, 9))
zipf = bool(rng.random() < 0.4)
dtype = np.float32 if rng.random() < 0.7 else np.float64
name, base, skip_tn = METRICS[int(rng.integers(len(METRICS)))]
init = str(rng.choice(["top", "top", "random", "greedy"]))
Y = make_csr(n, max(m, r + 1), r, seed=seed, zipf=zipf, k=k, dtype=dtype)
m = Y.shape[1]
kw = dict(seed=int(rng.integers(100)), max_iters=6, tolerance=-1.0, skip_tn=skip_tn, init_y_pred=init)
metric = oref.make_metric(base, k=float(k), m=float(m))
_, mo = oref.predict_using_bc_with_0approx(Y, metric, k, **kw)
print(f"seed={seed} n={n} m={m} r={r} k={k} zipf={zipf} {np.dtype(dtype).name} {name} init={init}")
for w in [None] + [int(x) for x in sys.argv[2:]]:
    for rep in range(3):
        _, mg = bc.predict_using_bc_with_0approx(Y, getattr(bc, name), k, return_meta=True, bca_diagnostics=True, bca_waves=w, **kw)
        d = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
        print(f"  bca_waves={w} W={mg['wavefronts']} changed={mg['rows_changed']} diff " + " ".join(f"{x:.1e}" for x in d), flush=True)

#!/usr/bin/env python3
"""xcolumns_amd in five minutes (needs an MI355X and the built library: `make -C xcolumns_amd/csrc`).

Every call below has the signature of its namesake in mwydmuch/xCOLUMNs; swap the import and keep the rest.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
from xcolumns_amd.block_coordinate import (predict_optimizing_coverage_using_bc,
                                           predict_optimizing_macro_f1_score_using_bc)
from xcolumns_amd.confusion_matrix import calculate_confusion_matrix
from xcolumns_amd.frank_wolfe import find_classifier_optimizing_macro_f1_score_using_fw
from xcolumns_amd.metrics import macro_f1_score, macro_f1_score_on_conf_matrix
from xcolumns_amd.synthetic import make_csr
from xcolumns_amd.weighted_prediction import predict_top_k

k = 5
y_proba = make_csr(50_000, 10_000, 40, seed=1, k=k, zipf=True)           # scipy CSR, float32 scores
rng = np.random.default_rng(0)
y_true = y_proba.copy()
y_true.data = (rng.random(y_true.nnz) < y_true.data).astype(np.float32)  # labels drawn from the scores
y_true.eliminate_zeros()

top = predict_top_k(y_proba, k)
bca, meta = predict_optimizing_macro_f1_score_using_bc(y_proba, k, seed=13, return_meta=True)
print(f"macro-F1 on the drawn labels: top-k {macro_f1_score(y_true, top):.4f}  BCA {macro_f1_score(y_true, bca):.4f} "
      f"({meta['iters']} sweeps, {meta['time'] * 1e3:.1f} ms; expected utility {meta['utilities'][-1]:.4f})")

# a randomized weighted classifier fitted on one half, applied to the other (Frank-Wolfe)
half = y_proba.shape[0] // 2
clf = find_classifier_optimizing_macro_f1_score_using_fw(y_true[:half], y_proba[:half], k, max_iters=10)
fw_pred = clf.predict(y_proba[half:], seed=7)
print(f"Frank-Wolfe, {clf.a.shape[0]} weighted classifiers: macro-F1 on the held-out half "
      f"{macro_f1_score_on_conf_matrix(*calculate_confusion_matrix(y_true[half:], fw_pred, normalize=True)):.4f}")

cov, meta = predict_optimizing_coverage_using_bc(y_proba[:5000], 3, seed=2, max_iters=5, return_meta=True)
print(f"expected coverage@3 of 5000 rows after {meta['iters']} sweeps: {meta['utilities'][-1]:.4f}")

#!/usr/bin/env python3
"""Round-2 check (GPU box): the DEFAULT policy through the public API against pinned oracle utilities
(tests/studies/drift_study.py / policy_study.py printed them) on the large shapes, three runs each."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xcolumns_amd import DeviceCSR  # noqa: E402
from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f  # noqa: E402
from xcolumns_amd.synthetic import make_csr, make_csr_rows  # noqa: E402

CASES = [
    ("ns uniform (make_csr)", lambda: make_csr(1_000_000, 500_000, 50, seed=20240001),
     [0.47616679986311045, 0.4774164556921846, 0.47758207271633435, 0.47761463305888935, 0.47762111649320843]),
    ("ns zipf (make_csr_rows)", lambda: make_csr_rows(1_000_000, 500_000, 0, 1_000_000, 50, seed=20240001, zipf=True),
     [0.6291719737586741, 0.6365840213827987, 0.636688558167669, 0.6366968086388481, 0.6366977110438439, 0.636697843352911]),
    ("400k zipf (make_csr_rows)", lambda: make_csr_rows(400_000, 200_000, 0, 400_000, 50, seed=20240001, zipf=True),
     [0.6217677945331616, 0.6286271487970402, 0.6287431276170287, 0.6287529114912591, 0.6287541351535234, 0.6287542790881697]),
    ("c2 zipf (make_csr)", lambda: make_csr(100_000, 30_000, 50, seed=20240001, zipf=True),
     [0.648199043445261, 0.6547032431173719, 0.6547503099475628, 0.6547521283506257, 0.6547522939429329]),
    ("c2 uniform (make_csr)", lambda: make_csr(100_000, 30_000, 50, seed=20240001),
     [0.47744958358373496, 0.47820571741710327, 0.47829887894416656, 0.4783173181116867, 0.47832071296902456]),
]
DET = os.environ.get("XC_DET") == "1"   # the deterministic concurrent mode instead of the default sweep
for name, gen, uo in CASES:
    if DET and name.startswith("400k"):
        continue
    Y = DeviceCSR.from_scipy(gen())
    uo = np.asarray(uo)
    for rep in range(3):
        _, meta = f(Y, 5, seed=13, max_iters=len(uo), tolerance=-1.0, return_meta=True, bca_diagnostics=True,
                    bca_deterministic=DET)
        d = np.abs(np.asarray(meta["utilities"]) - uo)
        print(f"{name:28s} W={meta['wavefronts']} diff=[{' '.join('%.1e' % x for x in d)}] max={d.max():.1e} "
              f"time={meta['time'] * 1e3:.1f} ms", flush=True)

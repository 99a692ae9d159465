#!/usr/bin/env python3
"""Round-2 study (GPU box): C3 shape (150 K x 670 K, about one predicted row per label): per-sweep |utility - oracle|
and sweep time against the number of wavefronts walking the order, several visiting-order seeds."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref as oref  # noqa: E402
from xcolumns_amd import DeviceCSR  # noqa: E402
from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f  # noqa: E402
from xcolumns_amd.synthetic import make_csr  # noqa: E402

oref.build()
shape = os.environ.get("XC_SHAPE", "150000x670000")
n, m = (int(x) for x in shape.split("x"))
k, sweeps = 5, int(os.environ.get("XC_SWEEPS", "4"))
Y = make_csr(n, m, 50, seed=20240003, k=k, zipf=os.environ.get("XC_ZIPF") == "1")
Yd = DeviceCSR.from_scipy(Y)
metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
widths = [int(x) for x in sys.argv[1:]] or [8, 16, 32, 64, 128]
for seed in (13, 14, 15):
    t0 = time.time()
    _, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=seed, max_iters=sweeps, tolerance=-1.0)
    print(f"seed {seed}: oracle {time.time() - t0:.1f} s", flush=True)
    for w in widths:
        f(Yd, k, seed=seed, max_iters=1, tolerance=-1.0, bca_waves=w)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, mg = f(Yd, k, seed=seed, max_iters=sweeps, tolerance=-1.0, return_meta=True, bca_waves=w)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
        d = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
        print(f"  W={w:5d}  call {ms:8.1f} ms  |utility - oracle| per sweep " + " ".join(f"{x:.1e}" for x in d), flush=True)

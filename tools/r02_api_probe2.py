#!/usr/bin/env python3
"""Round-2 probe (GPU box): host-side profile of one public BCA call on a device-resident matrix."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xcolumns_amd import DeviceCSR  # noqa: E402
from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f  # noqa: E402
from xcolumns_amd.synthetic import WORKLOADS, make_csr  # noqa: E402

for wl in sys.argv[1:] or ["c2_100Kx30K"]:
    n, m = WORKLOADS[wl]
    Y = make_csr(n, m, 50, seed=20240001, zipf=False, k=5)
    Yd = DeviceCSR.from_scipy(Y)
    for name, inp in (("host", Y), ("device", Yd)):
        ts = []
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            f(inp, 5, tolerance=-1.0, max_iters=10, seed=13, return_meta=True)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(wl, name, " ".join("%.1f" % t for t in ts), "ms", flush=True)
        pr = cProfile.Profile()
        pr.enable()
        f(inp, 5, tolerance=-1.0, max_iters=10, seed=13, return_meta=True)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

#!/usr/bin/env python3
"""Round-2 timing (GPU box): the sequential (one-wavefront, exact) sweep -- ms per sweep and microseconds per row."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xcolumns_amd import DeviceCSR  # noqa: E402
from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f  # noqa: E402
from xcolumns_amd.synthetic import make_csr  # noqa: E402

for n, m, zipf in ((100_000, 30_000, False), (100_000, 30_000, True), (150_000, 670_000, False)):
    Yd = DeviceCSR.from_scipy(make_csr(n, m, 50, seed=20240001, zipf=zipf, k=5))
    f(Yd, 5, seed=13, max_iters=1, tolerance=-1.0, bca_waves=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, meta = f(Yd, 5, seed=13, max_iters=3, tolerance=-1.0, bca_waves=1, return_meta=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{n}x{m} zipf={zipf}: {dt / 3 * 1e3:.1f} ms per sequential sweep = {dt / 3 / n * 1e6:.2f} us per row; utilities {meta['utilities']}", flush=True)

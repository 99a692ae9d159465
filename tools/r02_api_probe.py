#!/usr/bin/env python3
"""Round-2 probe (GPU box): where does the time of one public BCA call at the north-star size go?"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xcolumns_amd import DeviceCSR, _lib  # noqa: E402
from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f  # noqa: E402
from xcolumns_amd.synthetic import make_csr_rows  # noqa: E402
from xcolumns_amd.utils import Pcg64Shuffler  # noqa: E402

n, m = 1_000_000, 500_000
print("shuffler usable:", Pcg64Shuffler.usable(), flush=True)
s = Pcg64Shuffler(np.random.default_rng(13), n)
ref = np.arange(n)
rng = np.random.default_rng(13)
for rep in range(4):
    t0 = time.perf_counter(); js = s.draws(); t1 = time.perf_counter(); s.apply(js); t2 = time.perf_counter()
    rng.shuffle(ref); t3 = time.perf_counter()
    print(f"draws {(t1 - t0) * 1e3:.2f} ms  apply {(t2 - t1) * 1e3:.2f} ms  numpy shuffle {(t3 - t2) * 1e3:.2f} ms  equal {np.array_equal(ref, s.order)}", flush=True)
Y = make_csr_rows(n, m, 0, n, 50, seed=20240001)
Yd = DeviceCSR.from_scipy(Y)
for env in ({}, {"XCOLUMNS_ORDER_FAST_SHUFFLE": "0"}, {"XCOLUMNS_ORDER_PREFETCH": "0"}, {"XCOLUMNS_BCA_SCATTER": "0"}, {}):
    os.environ.update(env)
    ts = []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, meta = f(Yd, 5, tolerance=-1.0, max_iters=10, seed=13, return_meta=True)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    for k_ in env:
        os.environ.pop(k_)
    print(f"device-resident call, 10 sweeps, env {env}: {' '.join('%.1f' % t for t in ts)} ms", flush=True)
for order in ("device",):
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, meta = f(Yd, 5, tolerance=-1.0, max_iters=10, seed=13, return_meta=True, order_backend=order)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"device-resident call, order_backend={order}: {' '.join('%.1f' % t for t in ts)} ms", flush=True)

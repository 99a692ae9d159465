#!/usr/bin/env python3
"""Where does a row's time go?  Runs the BCA sweep from the diagnostic build
(tools/_build/libxcolumns_amd_stamps.so, -DXC_STAMPS) and prints each phase's share
of the wave cycles.  Read the SHARES, not the run time (stamps serialise).

    python tools/stamp_study.py [n m waves]
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xcolumns_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "_build", "libxcolumns_amd_stamps.so")
from xcolumns_amd import _device as D  # noqa: E402
from xcolumns_amd.block_coordinate import BcaCsrEngine  # noqa: E402
from xcolumns_amd.metrics import MetricSpec  # noqa: E402
from xcolumns_amd.synthetic import make_csr  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 30_000
waves_list = [int(x) for x in sys.argv[3:]] or [390, 1562, 8192]
zipf = os.environ.get("XC_STAMP_ZIPF") == "1"
Y = make_csr(n, m, 50, seed=20240001, zipf=zipf, k=5)
dev = D.require_gpu()
lib = _lib.load()
lib.xc_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(8, dtype=torch.int64, device=dev)
lib.xc_debug_set_stamp_buffer(ctypes.c_void_p(stamps.data_ptr()))
csr = D.DeviceCSR.from_scipy(Y, dev)
spec = MetricSpec(base=_lib.XC_M_FBETA)
names = ["issue gathers+prefetch", "membership", "wait gathers + gains", "top-k + commit", "stores+atomics",
         "prefetch landing/rotate"]
rng = np.random.default_rng(13)
order = np.arange(n)
for waves in waves_list:
    eng = BcaCsrEngine(csr, 5, spec, spec, maximize=True, skip_tn=True)
    eng.init_top()
    eng.reset_state(False)
    eng.recompute_utility_sum(n)
    for sweep in range(3):
        rng.shuffle(order)
        o = torch.from_numpy(order.astype(np.int32)).to(dev)
        stamps.zero_()
        eng.sweep(o, n, waves)
        torch.cuda.synchronize()
        t = stamps.cpu().numpy()[:6].astype(np.float64)
        per_row = t / n
        print(f"waves={waves} sweep={sweep + 1}: cycles/row total={per_row.sum():.0f}  " +
              "  ".join(f"{nm}={v:.0f} ({v / per_row.sum():.0%})" for nm, v in zip(names, per_row)), flush=True)
        eng.recompute_utility_sum(n)

#!/usr/bin/env python3
"""Round-2 study (GPU box): first-sweep |utility - oracle| on C2 with Zipf(1) labels for narrow widths, 6 runs each
(the difference is dominated by single tail-label decisions: one label's F1 weighs 1/m = 3.3e-5 here)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xcolumns_amd import _device as D, _lib  # noqa: E402
from xcolumns_amd.block_coordinate import BcaCsrEngine  # noqa: E402
from xcolumns_amd.metrics import MetricSpec  # noqa: E402
from xcolumns_amd.synthetic import make_csr  # noqa: E402

n, m = 100_000, 30_000
u1 = 0.648199043445261
Y = make_csr(n, m, 50, seed=20240001, zipf=True)
dev = D.require_gpu()
csr = D.DeviceCSR.from_scipy(Y, dev)
spec = MetricSpec(base=_lib.XC_M_FBETA)
rng = np.random.default_rng(13)
order = np.arange(n)
rng.shuffle(order)
o = torch.from_numpy(order.astype(np.int32)).to(dev)
if os.environ.get("XC_HOT_UNPUB"):
    _lib.load().xc_bca_set_tuning(-1.0, float(os.environ["XC_HOT_UNPUB"]))
print("hot_unpublished", os.environ.get("XC_HOT_UNPUB", "default"), "hot", os.environ.get("XCOLUMNS_BCA_HOT", "1"), flush=True)
for W in [int(w) for w in os.environ.get("XC_WIDTHS", "22,45,60,90,120,180,250").split(",")]:
    ds, ms = [], []
    for rep in range(6):
        eng = BcaCsrEngine(csr, 5, spec, spec, maximize=True, skip_tn=True)
        eng.init_top(); eng.reset_state(False); eng.recompute_utility_sum(n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.sweep(o, n, W); e1.record()
        ds.append(abs(eng.recompute_utility_sum(n) / m - u1)); ms.append(e0.elapsed_time(e1))
        eng.close()
    print(f"W={W:4d} sweep-1 diff: {' '.join('%.1e' % d for d in ds)}  max {max(ds):.1e}  ms {np.mean(ms):.2f}", flush=True)

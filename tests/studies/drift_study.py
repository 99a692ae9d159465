#!/usr/bin/env python3
"""Staleness study (GPU box): per-sweep utility of the concurrent BCA sweep for
several numbers of concurrent wavefronts against the sequential oracle, with the
sweep-kernel time and the number of rows whose prediction changed.

    python tests/studies/drift_study.py [n m [zipf]]
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref as oref  # noqa: E402  (checker)
from xcolumns_amd import _device as D, _lib  # noqa: E402
from xcolumns_amd.block_coordinate import BcaCsrEngine  # noqa: E402
from xcolumns_amd.metrics import MetricSpec  # noqa: E402
from xcolumns_amd.synthetic import make_csr  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 30_000
zipf = len(sys.argv) > 3 and sys.argv[3] == "zipf"
_lib.load().xc_bca_set_validation(int(os.environ.get("XC_VALIDATE", "2")))
r, k, sweeps = 50, 5, 5
Y = make_csr(n, m, r, seed=20240001, zipf=zipf)
t0 = time.time()
metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
_, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=13, max_iters=sweeps, tolerance=-1.0)
uo = np.asarray(mo["utilities"])
print(f"oracle {n}x{m} zipf={zipf}: {time.time() - t0:.1f}s utilities={uo.tolist()}", flush=True)

dev = D.require_gpu()
csr = D.DeviceCSR.from_scipy(Y, dev)
spec = MetricSpec(base=_lib.XC_M_FBETA)
info = _lib.device_info()
cap = info["cu_count"] * info["waves_per_cu"]
rng0 = np.random.default_rng(13)
order = np.arange(n)
orders = []
for s in range(sweeps):
    rng0.shuffle(order)
    orders.append(torch.from_numpy(order.astype(np.int32)).to(dev))

rows = []
for waves in sorted({1, max(1, n // 4096), max(1, n // 1024), max(1, n // 256), max(1, n // 64), max(1, n // 16), min(cap, n)}):
    if waves == 1 and n > 200_000:
        continue
    eng = BcaCsrEngine(csr, k, spec, spec, maximize=True, skip_tn=True)
    eng.init_top()
    eng.reset_state(False)
    eng.recompute_utility_sum(n)
    us, ms, ch = [], [], []
    for s in range(sweeps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.sweep(orders[s], n, waves)
        e1.record()
        us.append(eng.recompute_utility_sum(n) / m)
        ms.append(e0.elapsed_time(e1))
        ch.append(eng.rows_changed())
    d = np.abs(np.asarray(us) - uo)
    rows.append(dict(waves=waves, frac=waves / n, diff=d.tolist(), sweep_ms=ms, changed=ch))
    print(f"waves={waves:6d} ({waves / n:.4%}) diff={np.array2string(d, precision=2)} "
          f"ms={np.array2string(np.asarray(ms), precision=3)} changed={ch}", flush=True)
print(json.dumps(dict(n=n, m=m, zipf=zipf, oracle=uo.tolist(), rows=rows)))

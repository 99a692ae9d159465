#!/usr/bin/env python3
"""Per-sweep |utility - sequential oracle| and sweep-kernel time of the DEFAULT concurrency
policy at several staleness budgets (XCOLUMNS_BCA_STALE_BUDGET).   python tests/studies/policy_study.py [n m [zipf]]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref as oref  # noqa: E402  (checker)
from xcolumns_amd import _device as D, _lib  # noqa: E402
from xcolumns_amd.block_coordinate import BcaCsrEngine, WavePolicy  # noqa: E402
from xcolumns_amd.metrics import MetricSpec  # noqa: E402
from xcolumns_amd.synthetic import make_csr  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 30_000
zipf = len(sys.argv) > 3 and sys.argv[3] == "zipf"
k, sweeps = 5, int(os.environ.get("XC_STUDY_SWEEPS", "8"))
repeats = int(os.environ.get("XC_STUDY_REPEATS", "1"))
budgets = [float(x) for x in os.environ.get("XC_STUDY_BUDGETS", "1e-3,2e-3,4e-3,8e-3,1.0").split(",")]
if n > 200_000:
    from xcolumns_amd.synthetic import make_csr_rows
    Y = make_csr_rows(n, m, 0, n, 50, seed=20240001, zipf=zipf)
else:
    Y = make_csr(n, m, 50, seed=20240001, zipf=zipf)
metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
_, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=13, max_iters=sweeps, tolerance=-1.0)
uo = np.asarray(mo["utilities"])
dev = D.require_gpu()
csr = D.DeviceCSR.from_scipy(Y, dev)
spec = MetricSpec(base=_lib.XC_M_FBETA)
rng0 = np.random.default_rng(13)
order = np.arange(n)
orders = []
for s in range(sweeps):
    rng0.shuffle(order)
    orders.append(torch.from_numpy(order.astype(np.int32)).to(dev))
print(f"{n}x{m} zipf={zipf}; oracle utilities {uo.tolist()}")
for budget in [b for b in budgets for _ in range(repeats)]:
    eng = BcaCsrEngine(csr, k, spec, spec, skip_tn=True)
    eng.init_top()
    eng.reset_state(False)
    eng.recompute_utility_sum(n)
    pol = WavePolicy(n, budget=budget)
    changed, us, ws, ms = None, [], [], []
    for s in range(sweeps):
        w = pol.next(changed)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.sweep(orders[s], n, w)
        e1.record()
        us.append(eng.recompute_utility_sum(n) / m)
        changed = eng.rows_changed()
        ws.append(w)
        ms.append(e0.elapsed_time(e1))
    d = np.abs(np.asarray(us) - uo)
    print(f"budget={budget:g}: W={ws} diff={np.array2string(d, precision=1)} max={d.max():.1e} "
          f"sweep_ms_total={sum(ms):.3f} -> {n * sweeps / sum(ms) / 1e3:.0f} M rows/s (kernels only)", flush=True)

#!/usr/bin/env python3
"""Round-2 diagnostic (GPU box): what limits the first-sweep parity on Zipf label popularity?
|utility - sequential oracle| after sweeps 1 and 2 for several wavefront counts, with the hot-label
batching on / off and the two validation modes.  The oracle's utilities for these two matrices were
computed once (tests/studies/policy_study.py) and are pinned below."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xcolumns_amd import _device as D, _lib  # noqa: E402
from xcolumns_amd.block_coordinate import BcaCsrEngine  # noqa: E402
from xcolumns_amd.metrics import MetricSpec  # noqa: E402
from xcolumns_amd.synthetic import make_csr_rows  # noqa: E402

ORACLE = {(400_000, 200_000): [0.6217677945331616, 0.6286271487970402],
          (1_000_000, 500_000): [0.6291719737586741, 0.6365840213827987]}
n, m = int(sys.argv[1]), int(sys.argv[2])
uo = np.asarray(ORACLE[(n, m)])
Y = make_csr_rows(n, m, 0, n, 50, seed=20240001, zipf=True)
dev = D.require_gpu()
csr = D.DeviceCSR.from_scipy(Y, dev)
spec = MetricSpec(base=_lib.XC_M_FBETA)
rng = np.random.default_rng(13)
order = np.arange(n)
orders = []
for s in range(2):
    rng.shuffle(order)
    orders.append(torch.from_numpy(order.astype(np.int32)).to(dev))
waves = [int(w) for w in sys.argv[3].split(",")]
for label, env, validate in (("default", {}, 2), ("hot off", {"XCOLUMNS_BCA_HOT": "0"}, 2), ("validate=1", {}, 1)):
    for k_, v in env.items():
        os.environ[k_] = v
    _lib.load().xc_bca_set_validation(validate)
    for W in waves:
        eng = BcaCsrEngine(csr, 5, spec, spec, maximize=True, skip_tn=True)
        eng.init_top(); eng.reset_state(False); eng.recompute_utility_sum(n)
        us, ms = [], []
        for s in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.sweep(orders[s], n, W); e1.record()
            us.append(eng.recompute_utility_sum(n) / m)
            ms.append(e0.elapsed_time(e1))
        d = np.abs(np.asarray(us) - uo)
        print(f"{n}x{m} {label:12s} W={W:5d} diff=[{d[0]:.1e} {d[1]:.1e}] ms=[{ms[0]:.3f} {ms[1]:.3f}]", flush=True)
        eng.close()
    for k_ in env:
        os.environ.pop(k_)
_lib.load().xc_bca_set_validation(2)

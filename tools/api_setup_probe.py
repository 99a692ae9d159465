"""Where ONE public BCA call on the north-star workload spends its time outside the sweeps: every stage of
block_coordinate._bc_csr wrapped with a timer -- once with a device synchronisation after each stage (the stage's own
host + GPU time, no overlap), once without (what the host thread spends there while the GPU runs on).

    python tools/api_setup_probe.py [workload] [sweeps]"""
import os
import sys
import time
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xcolumns_amd import _device as D  # noqa: E402
from xcolumns_amd import block_coordinate as bc  # noqa: E402
from xcolumns_amd.synthetic import WORKLOADS, make_csr_rows  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "ns_1Mx500K"
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n, m = WORKLOADS[wl]
Yd = D.DeviceCSR.from_scipy(make_csr_rows(n, m, 0, n))
acc = defaultdict(float)
cnt = defaultdict(int)
SYNC = [True]


def wrap(owner, name, label=None):
    f = getattr(owner, name)
    label = label or name

    def g(*a, **kw):
        if SYNC[0]:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            return f(*a, **kw)
        finally:
            if SYNC[0]:
                torch.cuda.synchronize()
            acc[label] += time.perf_counter() - t0
            cnt[label] += 1
    setattr(owner, name, g)


E = bc.BcaCsrEngine
for owner, name in ((bc._OrderSource, "__init__"), (bc._OrderSource, "next"), (bc._OrderSource, "close"), (D, "as_device_csr"),
                    (E, "__init__"), (E, "init_top"), (E, "reset_state"), (E, "recompute_utility_sum"), (E, "pipeline_begin"),
                    (E, "pipeline_step"), (E, "pipeline_result"), (E, "close"), (E, "_repack"), (E, "_plan_handle"),
                    (D, "fixed_width_prediction")):
    wrap(owner, name, f"{owner.__name__.split('.')[-1]}.{name}")
f = bc.predict_optimizing_macro_f1_score_using_bc
for sync in (True, False):
    SYNC[0] = sync
    for rep in range(4):
        acc.clear()
        cnt.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f(Yd, 5, tolerance=-1.0, max_iters=sweeps, seed=13)
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
    print(f"== {wl}, {sweeps} sweeps, {'synchronised after every stage' if sync else 'no synchronisation (host time per stage)'}: "
          f"call {total * 1e3:.2f} ms; stages {sum(v for k, v in acc.items() if k not in ('BcaCsrEngine._repack', 'BcaCsrEngine._plan_handle')) * 1e3:.2f} ms")
    for k_, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"   {k_:42s} {v * 1e3:8.3f} ms  ({cnt[k_]} calls)")

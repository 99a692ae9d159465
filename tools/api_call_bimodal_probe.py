"""Call after call: wall time of the public BCA call at the north-star size (10, 10, 20 sweeps, ten calls each) with the order
generator's walk diagnostics -- how profiles/r03_api_call_timing.txt's alternating 10 / 14.5 ms calls were found (pooled side
streams sharing a hardware queue with the sweeps' stream)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["XCOLUMNS_ORDER_DEVICE"] = "1"
from xcolumns_amd import _device as D
from xcolumns_amd import utils
from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f
from xcolumns_amd.synthetic import WORKLOADS, make_csr_rows
n, m = WORKLOADS["ns_1Mx500K"]
Yd = D.DeviceCSR.from_scipy(make_csr_rows(n, m, 0, n))
walks = []
orig = utils.DeviceNumpyOrders.finish
def fin(self, sync_rng=False):
    orig(self, sync_rng); walks.append(dict(self.last_walk))
utils.DeviceNumpyOrders.finish = fin
for sweeps in (10, 10, 20):
    for i in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f(Yd, 5, tolerance=-1.0, max_iters=sweeps, seed=13)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        w = walks[-1]
        print(f"{sweeps} sweeps call {i}: {dt*1e3:.2f} ms; last walk {w['us']:.0f} us, {w['rounds']} rounds, fallbacks {w['fallbacks']}", flush=True)

"""What the fused from-scratch recompute (10 float64 atomics per row into `acc`) costs the
sweep kernel: the same converged sweep timed with and without it."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from xcolumns_amd import _device as D, _lib
from xcolumns_amd.block_coordinate import BcaCsrEngine
from xcolumns_amd.metrics import MetricSpec
from xcolumns_amd.synthetic import WORKLOADS, make_csr

wl = sys.argv[1] if len(sys.argv) > 1 else "ns_1Mx500K"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
n, m = WORKLOADS[wl]
Y = make_csr(n, m, 50, seed=20240001, k=5)
dev = D.require_gpu()
csr = D.DeviceCSR.from_scipy(Y, dev)
spec = MetricSpec(base=_lib.XC_M_FBETA)
eng = BcaCsrEngine(csr, 5, spec, spec, maximize=True, skip_tn=True)
eng.init_top(); eng.reset_state(greedy=False); eng.recompute_utility_sum(n)
rng = np.random.default_rng(13)
order = np.arange(n)
def nxt():
    rng.shuffle(order)
    return torch.from_numpy(order.astype(np.int32)).to(dev)
for _ in range(6):
    eng.sweep(nxt(), n, W); eng.recompute_utility_sum(n)
for label, n_order in (("with acc", n), ("without acc", n - 1), ("with acc", n), ("without acc", n - 1)):
    o = nxt(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.sweep(o, n_order, W); e1.record(); torch.cuda.synchronize()
    print(wl, "W", W, label, "sweep %.4f ms" % e0.elapsed_time(e1), "changed", end=" ", flush=True)
    eng.recompute_utility_sum(n); print(eng.rows_changed(), flush=True)

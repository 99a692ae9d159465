"""ONE public BCA call on the north-star workload (matrix resident in HBM): wall time for 10 and 20 sweeps with the
visiting orders generated on the GPU (default) and on the host (XCOLUMNS_ORDER_DEVICE=0), and with torch.randperm."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xcolumns_amd import _device as D  # noqa: E402
from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f  # noqa: E402
from xcolumns_amd.synthetic import WORKLOADS, make_csr_rows  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "ns_1Mx500K"
n, m = WORKLOADS[wl]
Y = make_csr_rows(n, m, 0, n)
Yd = D.DeviceCSR.from_scipy(Y)
for label, env, kw in (("orders on the GPU (numpy's stream)", {"XCOLUMNS_ORDER_DEVICE": "1"}, {}),
                       ("orders on the host (numpy's stream, two worker threads)", {"XCOLUMNS_ORDER_DEVICE": "0"}, {}),
                       ("order_backend='device' (torch.randperm: another stream)", {}, {"order_backend": "device"})):
    os.environ.update(env)
    for sweeps in (10, 20):
        ts = []
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, meta = f(Yd, 5, tolerance=-1.0, max_iters=sweeps, seed=13, return_meta=True, **kw)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        med = float(np.median(ts[1:]))
        print(f"{wl} {label}: {sweeps} sweeps {med * 1e3:.1f} ms ({min(ts[1:]) * 1e3:.1f}-{max(ts[1:]) * 1e3:.1f}) = "
              f"{n * sweeps / med:.3e} rows/s; last utility {meta['utilities'][-1]:.12f}", flush=True)

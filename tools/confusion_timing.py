"""calculate_confusion_matrix on CSR inputs resident in HBM at the north-star size (1 M x 500 K x 50, a 0/1 top-5 prediction):
first call (column sums of y_true + row check + statistics) and steady state, per form."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xcolumns_amd import _device as D  # noqa: E402
from xcolumns_amd.confusion_matrix import calculate_confusion_matrix  # noqa: E402
from xcolumns_amd.synthetic import WORKLOADS, make_csr_rows  # noqa: E402
from xcolumns_amd.weighted_prediction import predict_top_k  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "ns_1Mx500K"
n, m = WORKLOADS[wl]
Y = make_csr_rows(n, m, 0, n)
Yd = D.DeviceCSR.from_scipy(Y)
Pd = predict_top_k(Yd, 5)
calculate_confusion_matrix(D.DeviceCSR.from_scipy(Y[:200_000]), predict_top_k(D.DeviceCSR.from_scipy(Y[:200_000]), 5), skip_tn=True,
                           dtype=torch.float64)      # the process's first call (library and allocator warm-up) is not what is measured
ref = None
for label, env in (("default (match + counting sort, no global atomics)", {}),
                   ("XCOLUMNS_CONFUSION_SCATTER=0 (atomics for the predicted entries)", {"XCOLUMNS_CONFUSION_SCATTER": "0"}),
                   ("XCOLUMNS_CONFUSION_PRED_SIDE=0 (the general kernel: one atomic per contribution)", {"XCOLUMNS_CONFUSION_PRED_SIDE": "0"})):
    os.environ.update(env)
    Yd.forget_cached()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    C = calculate_confusion_matrix(Yd, Pd, skip_tn=True, dtype=torch.float64)
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    ts = []
    for _ in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        C = calculate_confusion_matrix(Yd, Pd, skip_tn=True, dtype=torch.float64)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    got = torch.stack([C.tp, C.fp, C.fn])
    if ref is None:
        ref = got
    print(f"{wl} {label}: first call {first * 1e3:.2f} ms, steady {np.median(ts) * 1e3:.3f} ms (min {min(ts) * 1e3:.3f}); "
          f"max |difference to the default form| {float((got - ref).abs().max()):.2e}", flush=True)
    for k_ in env:
        os.environ.pop(k_)

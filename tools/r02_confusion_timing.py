#!/usr/bin/env python3
"""Round-2 timing (GPU box): xc_confusion_csr (one global float64 atomic per contribution) against
xc_confusion_csr_bucketed (counting sort by label bucket + LDS sums) at the north-star and the C5 shapes."""
import os
import sys
import time

import numpy as np
import torch
from scipy.sparse import csr_matrix

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xcolumns_amd import DeviceCSR  # noqa: E402
from xcolumns_amd.confusion_matrix import confusion_csr_device  # noqa: E402
from xcolumns_amd.synthetic import make_csr  # noqa: E402
from xcolumns_amd.weighted_prediction import predict_top_k  # noqa: E402


def timed(t, p, mode, reps=6):
    os.environ["XCOLUMNS_CONFUSION_BUCKETED"] = mode
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        out = confusion_csr_device(t, p)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return out, ms


only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, n, m, binary in (("ns_1Mx500K, y_true = y_proba (50 entries/row)", 1_000_000, 500_000, False),
                           ("c5 shape 1.7Mx2.8M, binary y_true drawn from y_proba", 1_700_000, 2_800_000, True),
                           ("c2 100Kx30K, y_true = y_proba", 100_000, 30_000, False)):
    if only and not name.startswith(only):
        continue
    t0 = time.time()
    Yp = make_csr(n, m, 50, seed=20240005, k=5)
    if binary:
        keep = np.random.default_rng(5).random(Yp.nnz) < Yp.data
        Yt = csr_matrix((keep.astype(np.float32), Yp.indices.copy(), Yp.indptr.copy()), shape=Yp.shape)
        Yt.eliminate_zeros()
    else:
        Yt = Yp
    t, p = DeviceCSR.from_scipy(Yt), DeviceCSR.from_scipy(Yp)
    pred = predict_top_k(p, 5)
    print(f"{name}: true entries {t.nnz}, predicted entries {pred.nnz} (generated in {time.time() - t0:.1f} s)", flush=True)
    a, ms_a = timed(t, pred, "0")
    b, ms_b = timed(t, pred, "1")
    rel = float(((a - b).abs() / a.abs().clamp_min(1.0)).max())
    print("  atomic   ms:", " ".join("%.3f" % x for x in ms_a))
    print("  bucketed ms:", " ".join("%.3f" % x for x in ms_b), f"   max relative difference {rel:.2e}", flush=True)
    os.environ.pop("XCOLUMNS_CONFUSION_BUCKETED")
    t.forget_cached()
    ms_c = []
    for _ in range(6):      # the first call also sums the columns of y_true and checks its rows; the others find them cached
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        c = confusion_csr_device(t, pred)
        e1.record()
        torch.cuda.synchronize()
        ms_c.append(e0.elapsed_time(e1))
    rel = float(((a - c).abs() / a.abs().clamp_min(1.0)).max())
    print("  prediction side (default) ms:", " ".join("%.3f" % x for x in ms_c), f"   max relative difference {rel:.2e}", flush=True)
    del t, p, pred, a, b
    torch.cuda.empty_cache()

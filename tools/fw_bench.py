"""Stage timings of one Frank-Wolfe iteration (BASELINE config 5 shape) on the GPU.

    python tools/fw_bench.py [n] [m] [iters]

Synthetic y_proba as in bench.py (r = 50 entries per row, float32), y_true drawn from it."""
import os
import sys
import time

import numpy as np
import torch
from scipy.sparse import csr_matrix

sys.path.insert(0, ".")
from xcolumns_amd import _lib
from xcolumns_amd.frank_wolfe import FwEngine, FwObjective
from xcolumns_amd.metrics import MetricSpec
from xcolumns_amd.synthetic import make_csr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_700_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 2_800_000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
k, r = 5, 50
t0 = time.time()
Yp = make_csr(n, m, r, seed=20240005, k=k)
rng = np.random.default_rng(5)
keep = rng.random(Yp.nnz) < Yp.data
Yt = csr_matrix((keep.astype(np.float32), Yp.indices.copy(), Yp.indptr.copy()), shape=Yp.shape)
Yt.eliminate_zeros()
print(f"generated {n} x {m}, nnz {Yp.nnz}, true labels {Yt.nnz} in {time.time() - t0:.1f}s", flush=True)

eng = FwEngine(Yt, Yp, k, FwObjective(MetricSpec(base=_lib.XC_M_FBETA), "macro"), True, True, True)


def timed(label, fn, acc):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter()
    e0.record()
    out = fn()
    e1.record()
    torch.cuda.synchronize()
    acc.setdefault(label, []).append((e0.elapsed_time(e1), (time.perf_counter() - t) * 1e3))
    return out


acc = {}
a = np.ones(m, dtype=np.float32)
b = np.full(m, -0.5, dtype=np.float32)
stats = timed("predict+confusion", lambda: eng.confusion_of(a, b), acc)
for it in range(iters):
    a, b = timed("gradient", lambda: eng.next_classifier(stats), acc)
    stats_i = timed("predict+confusion", lambda: eng.confusion_of(a, b), acc)
    timed("utility", lambda: eng.utility(stats_i), acc)
    alpha = timed("alpha search (10^4 points)", lambda: eng.best_alpha(stats, stats_i, "uniform", 1e-3, 1e-4), acc)
    if os.environ.get("XC_FW_COUNT_DISTINCT"):  # how many distinct (cur, nxt) label tuples does the scan see
        both = torch.cat([stats[:3], stats_i[:3]]).t().contiguous()
        moved = int((stats[:3] != stats_i[:3]).any(dim=0).sum())
        print(f"  labels {m}, moved by the step {moved}, distinct (cur, nxt) tuples {torch.unique(both, dim=0).shape[0]}", flush=True)
    stats = (1 - alpha) * stats + alpha * stats_i
    print(f"iter {it}: alpha {alpha} utility {eng.utility(stats):.6f}", flush=True)
for kx, v in acc.items():
    dev = np.mean([x[0] for x in v[1:] or v])
    wall = np.mean([x[1] for x in v[1:] or v])
    print(f"{kx:30s} device {dev:9.3f} ms   wall {wall:9.3f} ms   ({len(v)} calls)")

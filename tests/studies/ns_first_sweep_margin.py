"""STUDY: the north-star shape's first two sweeps against the sequential oracle over visiting orders, at the whole GPU
(the default: the width rule exceeds the resident wavefronts) and at narrower first sweeps (XCOLUMNS_BCA_BETA)."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle import ref as oref  # noqa: E402
from xcolumns_amd.synthetic import make_csr_rows  # noqa: E402

n, m, k = 1_000_000, 500_000, 5
Y = make_csr_rows(n, m, 0, n, 50, seed=20240001, k=k)
metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
seeds = (13, 7, 2024, 99, 5)
ref = {}
for s in seeds:
    t0 = time.time()
    _, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=s, max_iters=2, tolerance=-1.0)
    ref[s] = mo["utilities"]
    print("oracle seed", s, "%.0f s" % (time.time() - t0), flush=True)
np.save("/tmp/ns_ref.npy", np.asarray([ref[s] for s in seeds]))
CHILD = r'''
import sys, numpy as np, torch, time
sys.path.insert(0, %r)
from xcolumns_amd import _device as D
from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f
from xcolumns_amd.synthetic import make_csr_rows
Y = D.DeviceCSR.from_scipy(make_csr_rows(1000000, 500000, 0, 1000000, 50, seed=20240001, k=5))
ref = np.load("/tmp/ns_ref.npy")
out = []
for q, seed in enumerate((13, 7, 2024, 99, 5)):
    for rep in range(2):
        _, mg = f(Y, 5, seed=seed, max_iters=2, tolerance=-1.0, return_meta=True, bca_diagnostics=True)
        d = np.abs(np.asarray(mg["utilities"]) - ref[q])
        out.append((seed, [round(float(x) / 1e-5, 3) for x in d], mg["wavefronts"]))
print("RESULT", out)
''' % ROOT
for beta in ("0.025", "0.008", "0.004"):
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, XCOLUMNS_BCA_BETA=beta), capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
    print("beta", beta, line[0][7:] if line else r.stderr[-800:], flush=True)

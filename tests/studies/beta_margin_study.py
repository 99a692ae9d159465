"""STUDY: how far below north_star's 1e-5 do the concurrent (default) sweeps sit, by width budget (XCOLUMNS_BCA_BETA) and
visiting order (seed)?  max over the sweeps of |utility - sequential oracle| / 1e-5 on the shapes whose early sweeps the
policy NARROWS (the width rule stays below the whole GPU)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from oracle import ref as oref
from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f
from xcolumns_amd.synthetic import make_csr
shapes = {"c2": (100000, 30000, 50, False), "40Kx10K": (40000, 10000, 40, False), "200Kx60K": (200000, 60000, 50, False)}
n, m, r, z = shapes[sys.argv[1]]
Y = make_csr(n, m, r, seed=20240001, zipf=z, k=5)
metric = oref.make_metric(oref.FBETA, k=5.0, m=float(m))
out = []
for seed in (13, 7, 2024, 99, 5):
    _, mo = oref.predict_using_bc_with_0approx(Y, metric, 5, skip_tn=True, seed=seed, max_iters=4, tolerance=-1.0)
    _, mg = f(Y, 5, seed=seed, max_iters=4, tolerance=-1.0, return_meta=True, bca_diagnostics=True)
    d = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
    out.append((seed, float(d.max()) / 1e-5, mg["wavefronts"]))
print("RESULT", out)
''' % ROOT

for shape in ("c2", "40Kx10K", "200Kx60K"):
    for beta in ("0.05", "0.025", "0.0125"):
        env = dict(os.environ, XCOLUMNS_BCA_BETA=beta)
        r = subprocess.run([sys.executable, "-c", CHILD, shape], env=env, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
        print(shape, "beta", beta, line[0][7:] if line else r.stderr[-500:], flush=True)

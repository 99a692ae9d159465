"""The ordered parallel sweep from a random and from the top-k start on Zipf(1) workloads: kernel time per sweep, iterations,
hot tables -- run with XCOLUMNS_BCA_ORD_HOT_READERS = 48 and 1600 for profiles/r03_ordered_hot_threshold.txt."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from xcolumns_amd import _device as D
from xcolumns_amd import block_coordinate as bc
from xcolumns_amd.synthetic import WORKLOADS, make_csr_rows
for wl in ("c2_100Kx30K", "c3_amazon670k_150Kx670K"):
    n, m = WORKLOADS[wl]
    Yd = D.as_device_csr(make_csr_rows(n, m, 0, n, zipf=True), D.require_gpu())
    seen = []
    orig = bc.BcaCsrEngine.sweep_ordered
    def spy(self, order, n_order):
        orig(self, order, n_order); seen.append(dict(self.ordered_stats))
    bc.BcaCsrEngine.sweep_ordered = spy
    for init in ("random", "top"):
        for rep in range(2):
            seen.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
            bc.predict_optimizing_macro_f1_score_using_bc(Yd, 5, seed=3, max_iters=3, tolerance=-1.0, bca_waves=1, init_y_pred=init)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(wl, "zipf init", init, "hot readers >=", os.environ.get("XCOLUMNS_BCA_ORD_HOT_READERS", "default"), ": call %.1f ms;" % (dt * 1e3),
              "kernel ms per sweep", [round(s["kernel_us"] / 1e3, 2) for s in seen], "iterations", [s["iterations"] for s in seen], "errors", [s["error"] for s in seen], "hot", seen[0]["n_hot"] if seen else None, flush=True)
    bc.BcaCsrEngine.sweep_ordered = orig

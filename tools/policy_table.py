"""What the default policy (bca_parity="per_sweep") does per workload: which sweeps run concurrently (wavefronts) and which
exactly (the ordered parallel sweep: window, iterations), and what ONE public call costs -- a markdown table for DESIGN.md.

    python tools/policy_table.py [sweeps] > profiles/r03_policy_table.md      (on the GPU box)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xcolumns_amd import _device as D  # noqa: E402
from xcolumns_amd import block_coordinate as bc  # noqa: E402
from xcolumns_amd.synthetic import WORKLOADS, make_csr_rows  # noqa: E402

SWEEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ROWS = [("configs[1] C2 100K x 30K", "c2_100Kx30K", False), ("C2 with Zipf(1) labels", "c2_100Kx30K", True),
        ("configs[2] C3 150K x 670K", "c3_amazon670k_150Kx670K", False), ("C3 with Zipf(1) labels", "c3_amazon670k_150Kx670K", True),
        ("configs[3] C4 780K x 500K (one GPU)", "c4_wiki500k_780Kx500K", False), ("C4 with Zipf(1) labels", "c4_wiki500k_780Kx500K", True),
        ("north star 1M x 500K", "ns_1Mx500K", False), ("north star with Zipf(1) labels", "ns_1Mx500K", True)]
seen = []
orig = bc.BcaCsrEngine.sweep_ordered


def spy(self, order, n_order):
    orig(self, order, n_order)
    seen.append(dict(self.ordered_stats))


bc.BcaCsrEngine.sweep_ordered = spy
print(f"| workload (macro-F1, k = 5, 50 entries per row, {SWEEPS} sweeps, top-k start) | sweeps run exactly (the ordered parallel sweep) | "
      "wavefronts of the concurrent sweeps | one public call, matrix in HBM | rows/s | fraction of the HBM roofline (2068 B per row and sweep) |")
print("|---|---|---|---|---|---|")
for label, wl, zipf in ROWS:
    n, m = WORKLOADS[wl]
    Yd = D.DeviceCSR.from_scipy(make_csr_rows(n, m, 0, n, zipf=zipf))
    seen.clear()
    # the widths come from a diagnostics call (which syncs per sweep); the time from plain calls
    _, meta = bc.predict_optimizing_macro_f1_score_using_bc(Yd, 5, seed=13, max_iters=SWEEPS, tolerance=-1.0, return_meta=True,
                                                            bca_diagnostics=True)
    stats = [dict(s) for s in seen]
    ts = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bc.predict_optimizing_macro_f1_score_using_bc(Yd, 5, seed=13, max_iters=SWEEPS, tolerance=-1.0)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    seen[:] = stats
    med = float(np.median(ts[1:]))
    w = meta["wavefronts"]
    exact = [j + 1 for j, x in enumerate(w) if x == 1]
    ex_txt = "none" if not exact else (f"{exact[0]}..{exact[-1]}" if exact == list(range(exact[0], exact[-1] + 1)) and len(exact) > 1 else ", ".join(map(str, exact)))
    if seen:
        ex_txt += f" (windows of {seen[0]['window']} rows; fixed-point iterations per window, mean per sweep: " \
                  f"{', '.join('%.1f' % (s['iterations'] / max(s['windows'], 1)) for s in seen)}; " \
                  f"kernel {', '.join('%.2f' % (s['kernel_us'] / 1e3) for s in seen)} ms)"
    conc = [x for x in w if x != 1]
    print(f"| {label} | {ex_txt} | {', '.join(map(str, conc)) if conc else '-'} | {med * 1e3:.1f} ms | {n * SWEEPS / med:.2e} | "
          f"{2068 * n * SWEEPS / med / 8e12:.3f} |", flush=True)
    del Yd
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""End-to-end timing of the public API call (host buffers in, host result out):
the PCIe- and host-RNG-inclusive rate DESIGN.md quotes next to bench.py's
HBM-resident number.   python tools/e2e_api_timing.py [workload]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc  # noqa: E402
from xcolumns_amd.synthetic import WORKLOADS, make_csr  # noqa: E402
from xcolumns_amd.weighted_prediction import predict_top_k  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2_100Kx30K"
n, m = WORKLOADS[wl]
Y = make_csr(n, m, 50, seed=20240001)
predict_top_k(Y[:1000], 5)  # warm the library / context
for backend in ("numpy", "device"):
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        P, meta = predict_optimizing_macro_f1_score_using_bc(Y, 5, seed=13, max_iters=10, tolerance=-1.0,
                                                             return_meta=True, order_backend=backend)
        dt = time.perf_counter() - t0
    print(f"{wl} order_backend={backend}: {dt * 1e3:.1f} ms for 10 sweeps -> {n * 10 / dt / 1e6:.1f} M rows/s "
          f"(meta.time {meta['time'] * 1e3:.1f} ms, final utility {meta['utilities'][-1]:.6f})", flush=True)
t0 = time.perf_counter()
T = predict_top_k(Y, 5)
print(f"{wl} predict_top_k end-to-end: {(time.perf_counter() - t0) * 1e3:.1f} ms -> {n / (time.perf_counter() - t0) / 1e6:.1f} M rows/s")

# BASELINE.json configs[0] shape: dense 3865 x 3956 float32 (EURLex-4K-like), k=5
from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as _bc  # noqa: E402
rng = np.random.default_rng(1)
Yd = (1.0 / (1.0 + np.exp(-rng.normal(-2.0, 1.5, size=(3865, 3956))))).astype(np.float32)
predict_top_k(Yd[:64], 5)
for it in range(2):
    t0 = time.perf_counter()
    Pd = predict_top_k(Yd, 5)
    dt = time.perf_counter() - t0
print(f"C1 dense 3865x3956 f32 predict_top_k (host in/out): {dt * 1e3:.2f} ms -> {3865 / dt / 1e3:.1f} K rows/s")
Yg = torch.from_numpy(Yd).cuda()
torch.cuda.synchronize()
t0 = time.perf_counter()
Pg = predict_top_k(Yg, 5)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"C1 dense predict_top_k (torch GPU tensor in/out): {dt * 1e3:.3f} ms -> {3865 / dt / 1e6:.2f} M rows/s")
t0 = time.perf_counter()
Pb, mb = _bc(Yg, 5, seed=0, max_iters=3, tolerance=-1e9, return_meta=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
t1 = time.perf_counter()
Pb1, mb1 = _bc(Yg, 5, seed=0, max_iters=3, tolerance=-1e9, return_meta=True, bca_waves=1)
torch.cuda.synchronize()
dt1 = time.perf_counter() - t1
print(f"C1 dense BCA macro-F1, 3 sweeps, bca_waves=1 (sequential, exact): {dt1 * 1e3:.1f} ms -> {3865 * 3 / dt1 / 1e3:.1f} K rows/s, "
      f"utilities {mb1['utilities']}")
print(f"C1 dense BCA macro-F1, 3 sweeps (torch GPU tensor): {dt * 1e3:.1f} ms -> {3865 * 3 / dt / 1e3:.1f} K rows/s, "
      f"utilities {mb['utilities']}")

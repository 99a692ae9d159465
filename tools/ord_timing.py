"""Time the ordered parallel sweep (csrc/xc_bca_ord.hip) through the public API: ms per sweep, iterations and
windows per sweep, and |utility - sequential oracle| (the oracle only up to --oracle-rows rows)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xcolumns_amd import _device as D  # noqa: E402
from xcolumns_amd import block_coordinate as bc  # noqa: E402
from xcolumns_amd.synthetic import WORKLOADS, make_csr_rows  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3_amazon670k_150Kx670K")
ap.add_argument("--zipf", action="store_true")
ap.add_argument("--sweeps", type=int, default=5)
ap.add_argument("--oracle", action="store_true")
ap.add_argument("--one-wave", action="store_true", help="also time the one-wavefront sweep (slow)")
args = ap.parse_args()

n, m = WORKLOADS[args.workload]
Y = make_csr_rows(n, m, 0, n, zipf=args.zipf)
dev = D.require_gpu()
Yd = D.as_device_csr(Y, dev)
torch.cuda.synchronize()
stats = []
orig = bc.BcaCsrEngine.sweep_ordered


def timed(self, order, n_order):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    orig(self, order, n_order)
    torch.cuda.synchronize()
    stats.append(dict(self.ordered_stats, ms=(time.perf_counter() - t0) * 1e3))


bc.BcaCsrEngine.sweep_ordered = timed
for rep in range(2):
    stats.clear()
    t0 = time.perf_counter()
    P, meta = bc.predict_optimizing_macro_f1_score_using_bc(Yd, 5, seed=13, max_iters=args.sweeps, tolerance=-1.0,
                                                            return_meta=True, bca_waves=1, bca_ordered=True)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
print(f"{args.workload}{' zipf' if args.zipf else ''}: whole call {wall:.1f} ms for {args.sweeps} sweeps "
      f"({n * args.sweeps / wall * 1e3:.3e} rows/s)")
for j, s in enumerate(stats):
    print(f"  sweep {j + 1}: {s['ms']:.3f} ms  {s['iterations']} iterations in {s['windows']} windows  "
          f"({n / s['ms'] * 1e3:.3e} rows/s; window {s['window']}; hot tables {s['n_hot']}; kernel {s['kernel_us'] / 1e3:.3f} ms, "
          f"{s['barrier_us'] / 1e3:.3f} ms of it in barriers; status {s['error']})")
print("  utilities", [f"{u:.12f}" for u in meta["utilities"]])
if args.one_wave:
    t0 = time.perf_counter()
    P1, meta1 = bc.predict_optimizing_macro_f1_score_using_bc(Yd, 5, seed=13, max_iters=args.sweeps, tolerance=-1.0,
                                                              return_meta=True, bca_waves=1, bca_ordered=False)
    torch.cuda.synchronize()
    print(f"  one wavefront: {(time.perf_counter() - t0) * 1e3:.1f} ms; max |utility difference| "
          f"{np.abs(np.asarray(meta1['utilities']) - np.asarray(meta['utilities'])).max():.2e}; "
          f"predictions equal: {bool((P1.indices == P.indices).all().item())}")
if args.oracle:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import ref as oref
    metric = oref.make_metric(oref.FBETA, k=5.0, m=float(m))
    t0 = time.perf_counter()
    Po, mo = oref.predict_using_bc_with_0approx(Y, metric, 5, skip_tn=True, seed=13, max_iters=args.sweeps, tolerance=-1.0)
    print(f"  oracle: {(time.perf_counter() - t0):.1f} s; |utility difference| per sweep "
          f"{np.abs(np.asarray(mo['utilities']) - np.asarray(meta['utilities']))}; predictions equal: "
          f"{bool(np.array_equal(Po.indices, P.indices.cpu().numpy()))}")

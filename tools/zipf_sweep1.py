"""Sweep-1 time of the Zipf C2 workload under toggles (what serialises it?)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from xcolumns_amd import _lib
if os.environ.get("XC_LIB"):
    _lib.LIB_PATH = os.environ["XC_LIB"]
from xcolumns_amd import _device as D
from xcolumns_amd.block_coordinate import BcaCsrEngine
from xcolumns_amd.metrics import MetricSpec
from xcolumns_amd.synthetic import make_csr
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100_000, 30_000)
Y = make_csr(n, m, 50, seed=20240001, zipf=True)
dev = D.require_gpu(); csr = D.DeviceCSR.from_scipy(Y, dev); spec = MetricSpec(base=_lib.XC_M_FBETA)
rng = np.random.default_rng(13); order = np.arange(n); rng.shuffle(order)
o = torch.from_numpy(order.astype(np.int32)).to(dev)
for label, env, validate, n_order in (("default", {}, 1, n), ("no validation", {}, 0, n), ("hot off", {"XCOLUMNS_BCA_HOT": "0"}, 1, n),
                                      ("no acc (n-1 rows)", {}, 1, n - 1), ("hot off, no validation", {"XCOLUMNS_BCA_HOT": "0"}, 0, n),
                                      ("no acc, no validation", {}, 0, n - 1)):
    for k_, v in env.items(): os.environ[k_] = v
    _lib.load().xc_bca_set_validation(validate)
    eng = BcaCsrEngine(csr, 5, spec, spec, maximize=True, skip_tn=True)
    for k_ in env: os.environ.pop(k_)
    eng.init_top(); eng.reset_state(False); eng.recompute_utility_sum(n)
    for W in (800, 8192):
        eng.init_top(); eng.reset_state(False); eng.recompute_utility_sum(n)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.sweep(o, n_order, W); e1.record(); torch.cuda.synchronize()
        u = eng.recompute_utility_sum(n) / m
        print(f"{label:28s} W={W:5d} sweep1 {e0.elapsed_time(e1):7.3f} ms  utility {u:.6f} changed {eng.rows_changed()}", flush=True)
_lib.load().xc_bca_set_validation(1)

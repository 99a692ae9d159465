"""Row top-k (xc_topk_csr) on the bench workloads: kernel time and algorithmic HBM rate
(B_topk = 4 + 8r + 4k per row).  XCOLUMNS_TOPK_ONE_ROW_PER_WAVE=1 for the A/B."""
import sys
import torch
sys.path.insert(0, ".")
import os
from xcolumns_amd import _device as D, _lib
if os.environ.get("XC_LIB"):
    _lib.LIB_PATH = os.environ["XC_LIB"]
from xcolumns_amd.synthetic import WORKLOADS, make_csr
from xcolumns_amd.weighted_prediction import topk_csr_device

for wl in sys.argv[1:] or ["c2_100Kx30K", "ns_1Mx500K"]:
    n, m = WORKLOADS[wl]
    Y = make_csr(n, m, 50, seed=20240001, k=5)
    dev = D.require_gpu()
    csr = D.DeviceCSR.from_scipy(Y, dev)
    a = torch.rand(m, device=dev) + 0.5
    b = torch.rand(m, device=dev) - 0.5
    for label, kw in (("plain", {}), ("weights a, b", dict(a=a, b=b))):
        for _ in range(3):
            topk_csr_device(csr, 5, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            topk_csr_device(csr, 5, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        bytes_row = 4 + 8 * 50 + 4 * 5 + (8 * 50 if kw else 0)
        print(f"{wl:14s} {label:14s} {ms * 1e3:8.1f} us/call (incl. output allocation)  {n / ms / 1e6:7.2f} G rows/s  "
              f"{n * bytes_row / ms / 1e9:6.2f} TB/s algorithmic", flush=True)

"""ONE workload for `rocprofv3 --kernel-trace --stats`: public BCA calls of 20 sweeps on the north-star matrix resident in HBM,
visiting orders generated on the GPU -- which kernels does a call spend its GPU time in?"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["XCOLUMNS_ORDER_DEVICE"] = "1"
from xcolumns_amd import _device as D  # noqa: E402
from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f  # noqa: E402
from xcolumns_amd.synthetic import WORKLOADS, make_csr_rows  # noqa: E402

n, m = WORKLOADS["ns_1Mx500K"]
Yd = D.DeviceCSR.from_scipy(make_csr_rows(n, m, 0, n))
for _ in range(5):
    f(Yd, 5, tolerance=-1.0, max_iters=20, seed=13)
torch.cuda.synchronize()

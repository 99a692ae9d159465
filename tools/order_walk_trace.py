"""Per-batch timeline of the device order generator's grid-wide rejection walk (csrc/xc_order_dev.hip)."""
import ctypes
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xcolumns_amd import _device as D, _lib  # noqa: E402
from xcolumns_amd.utils import DeviceNumpyOrders  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = D.require_gpu()
gen = DeviceNumpyOrders(np.random.default_rng(1), n, dev, ahead=0)
for _ in range(3):
    gen.next()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    gen.next()
torch.cuda.synchronize()
print(f"n = {n}: {(time.perf_counter() - t0) * 100:.3f} ms per order")
nb = ctypes.c_int64(0)
_lib.call("xc_order_dev_walk_trace", D.ptr(gen.ws), n, None, ctypes.byref(nb), D.stream())
out = (ctypes.c_int64 * (4 * nb.value))()
_lib.call("xc_order_dev_walk_trace", D.ptr(gen.ws), n, out, ctypes.byref(nb), D.stream())
a = np.frombuffer(out, dtype=np.uint64).reshape(-1, 4)
gen.finish()
print("status", gen.last_walk)
print("batch: start us, end us, waiting us, settling us, rounds, settles, inner rounds")
for b in sorted(set(list(range(0, min(6, len(a)))) + list(range(6, len(a), max(1, len(a) // 20))) + list(range(max(0, len(a) - 12), len(a))))):
    s_, e_, ws, ri = (int(x) for x in a[b])
    print(f"  {b:5d}: {s_ / 100:9.1f} {e_ / 100:9.1f} {(ws >> 32) / 100:9.1f} {(ws & 0xFFFFFFFF) / 100:9.1f} {ri >> 40:4d} {(ri >> 24) & 0xFFFF:4d} {ri & 0xFFFFFF:5d}")

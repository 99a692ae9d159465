"""Where a call on a scipy csr_matrix in host memory spends its time before the first kernel: the upload of the north-star
matrix (1 M x 500 K x 50: 4 + 200 + 200 MB) from pageable memory -- as DeviceCSR.from_scipy does it, and through a ring of
pinned staging buffers filled by worker threads."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xcolumns_amd import _device as D  # noqa: E402
from xcolumns_amd.synthetic import WORKLOADS, make_csr_rows  # noqa: E402

n, m = WORKLOADS["ns_1Mx500K"]
Y = make_csr_rows(n, m, 0, n)
dev = D.require_gpu()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a = torch.from_numpy(Y.indices).to(dev); torch.cuda.synchronize(); t1 = time.perf_counter()
    b = torch.from_numpy(Y.data).to(dev); torch.cuda.synchronize(); t2 = time.perf_counter()
    Yd = D.DeviceCSR.from_scipy(Y); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"pageable .to(device): indices {1e3*(t1-t0):.1f} ms ({Y.indices.nbytes/(t1-t0)/1e9:.1f} GB/s), data {1e3*(t2-t1):.1f} ms; "
          f"DeviceCSR.from_scipy {1e3*(t3-t2):.1f} ms", flush=True)


def staged(arr, chunk_bytes, n_bufs, threads):
    flat = arr.reshape(-1).view(np.uint8)
    out = torch.empty(flat.size, dtype=torch.uint8, device=dev)
    bufs = [torch.empty(chunk_bytes, dtype=torch.uint8).pin_memory() for _ in range(n_bufs)]
    evs = [None] * n_bufs
    pool = ThreadPoolExecutor(threads)
    chunks = [(o, min(o + chunk_bytes, flat.size)) for o in range(0, flat.size, chunk_bytes)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    futs = {}
    def fill(i, lo, hi):
        np.copyto(bufs[i % n_bufs].numpy()[:hi - lo], flat[lo:hi])
    ahead = n_bufs - 1
    for i, (lo, hi) in enumerate(chunks[:ahead]):
        futs[i] = pool.submit(fill, i, lo, hi)
    for i, (lo, hi) in enumerate(chunks):
        futs.pop(i).result()
        out[lo:hi].copy_(bufs[i % n_bufs][:hi - lo], non_blocking=True)
        ev = torch.cuda.Event(); ev.record(); evs[i % n_bufs] = ev
        nxt = i + ahead
        if nxt < len(chunks):
            j = nxt % n_bufs
            if evs[j] is not None:
                evs[j].synchronize()
            futs[nxt] = pool.submit(fill, nxt, *chunks[nxt])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pool.shutdown()
    ok = bool((out[:1 << 20].cpu().numpy() == flat[:1 << 20]).all()) and bool((out[-(1 << 20):].cpu().numpy() == flat[-(1 << 20):]).all())
    return dt, ok


for chunk_mb, n_bufs, threads in ((16, 4, 2), (16, 6, 4), (32, 4, 2), (32, 6, 4), (8, 8, 4), (64, 4, 4)):
    dt, ok = staged(Y.data, chunk_mb << 20, n_bufs, threads)
    dt2, ok2 = staged(Y.data, chunk_mb << 20, n_bufs, threads)
    print(f"staged upload of data (200 MB), chunks of {chunk_mb} MB, {n_bufs} pinned buffers, {threads} threads: {1e3*dt2:.1f} ms "
          f"({Y.data.nbytes/dt2/1e9:.1f} GB/s; first run incl. pinning {1e3*dt:.1f} ms) ok={ok and ok2}", flush=True)

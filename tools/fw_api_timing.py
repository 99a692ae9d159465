"""End-to-end Frank-Wolfe through the public API at the shape of BASELINE config 5 (host matrices in,
classifier out): per-iteration wall time incl. everything the API does.
    python tools/fw_api_timing.py [n] [m] [iters]"""
import sys, time
import numpy as np
import torch
from scipy.sparse import csr_matrix
sys.path.insert(0, ".")
from xcolumns_amd.frank_wolfe import find_classifier_optimizing_macro_f1_score_using_fw
from xcolumns_amd.synthetic import make_csr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_700_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 2_800_000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
Yp = make_csr(n, m, 50, seed=20240005, k=5)
rng = np.random.default_rng(5)
Yt = csr_matrix(((rng.random(Yp.nnz) < Yp.data).astype(np.float32), Yp.indices.copy(), Yp.indptr.copy()), shape=Yp.shape)
Yt.eliminate_zeros()
find_classifier_optimizing_macro_f1_score_using_fw(Yt[:1000], Yp[:1000], 5, max_iters=2)   # warm the library
for it in (1, iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    clf, meta = find_classifier_optimizing_macro_f1_score_using_fw(Yt, Yp, 5, max_iters=it, tolerance=-1.0, alpha_tolerance=-1.0,
                                                                   return_meta=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{n} x {m}: max_iters={it}: {dt * 1e3:.1f} ms total, {meta['iters']} iterations, utilities {meta['utilities'][0]:.5f} -> {meta['utilities'][-1]:.5f}", flush=True)
    if it == 1: t1 = dt
print(f"per additional iteration: {(dt - t1) / (iters - 1) * 1e3:.1f} ms")

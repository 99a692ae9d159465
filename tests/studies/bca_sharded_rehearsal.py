"""2 ranks sharing ONE GPU over gloo: predict_bca_csr_sharded with the GPU engine (device-side loop,
one all-reduce of acc per sweep).  Checks: same utilities and stopping decision on both ranks, the last
utility is the utility of the assembled prediction, and the trace stays near the sequential oracle's
(other ranks' updates are invisible within a sweep: DESIGN.md section 7).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 tests/studies/bca_sharded_rehearsal.py"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
from scipy.sparse import csr_matrix
sys.path.insert(0, ".")
from xcolumns_amd.distributed import TorchComm, predict_bca_csr_sharded, shard_bounds, shard_csr
from xcolumns_amd.metrics import binary_f1_score_on_conf_matrix
from xcolumns_amd.synthetic import make_csr

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("XC_REHEARSAL_BACKEND", "gloo")   # "nccl" (= RCCL): one GPU per rank, the multi-GPU node's form
if backend == "nccl":
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
else:
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
shape = os.environ.get("XC_BCA_REHEARSAL_SHAPE")   # "n,m,sweeps": the benchmark generator at that size, no sequential oracle
if shape:
    from xcolumns_amd.synthetic import make_csr_rows
    n, m, sweeps = (int(x) for x in shape.split(","))
    k = 5
    lo, hi = shard_bounds(n, world, rank)
    shard = make_csr_rows(n, m, lo, hi, 50, seed=20240004, k=k)
    tol = -1.0
else:
    n, m, k, sweeps, tol = 40001, 3000, 5, 6, 1e-7
    Y = make_csr(n, m, 30, seed=5, k=k)
    shard = shard_csr(Y, world, rank)
comm = TorchComm()
P, meta = predict_bca_csr_sharded(shard, binary_f1_score_on_conf_matrix, k, comm, skip_tn=True,
                                  seed=13, max_iters=sweeps, tolerance=tol)
gathered = [None] * world
dist.all_gather_object(gathered, (meta["utilities"], meta["iters"], P.indices, torch.cuda.current_device()))
if rank == 0:
    sys.path.insert(0, "tests")
    from oracle import ref as oref   # checker (this script is a rehearsal, not product code)
    assert all(g[0] == gathered[0][0] and g[1] == gathered[0][1] for g in gathered)
    idx = np.concatenate([g[2] for g in gathered])
    full = csr_matrix((np.ones(n * k, dtype=np.float32), idx, np.arange(n + 1) * k), shape=(n, m))
    metric = oref.make_metric(oref.FBETA, k=float(k), m=float(m))
    if shape:
        Y = make_csr_rows(n, m, 0, n, 50, seed=20240004, k=k)
    tp, fp, fn, tn = oref.calculate_confusion_matrix(Y, full, skip_tn=True)
    u = oref.calculate_utility(metric, "mean", tp / n, fp / n, fn / n, tn / n)
    if shape and os.environ.get("XC_BCA_REHEARSAL_ORACLE") != "1":
        mo = {"utilities": []}
    elif shape:
        _, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=13, max_iters=sweeps, tolerance=-1.0)
    else:
        _, mo = oref.predict_using_bc_with_0approx(Y, metric, k, skip_tn=True, seed=13, max_iters=6, tolerance=1e-7)
    print("utilities", meta["utilities"], "\noracle   ", mo["utilities"], "\nexchanges", meta.get("exchanges"),
          "\n|last - utility(assembled prediction)| =", abs(u - meta["utilities"][-1]), "all-reduce calls", comm.calls,
          "\nbackend", dist.get_backend(), "devices", sorted({int(g[3]) for g in gathered}), flush=True)
    assert abs(u - meta["utilities"][-1]) < 1e-12
dist.barrier()
dist.destroy_process_group()

"""2 ranks sharing ONE GPU over gloo: the GPU Frank-Wolfe engine with rows sharded must return the
single-process classifier bit for bit (label counts are all-reduced exactly).  Control-flow rehearsal;
real multi-GPU runs use backend nccl.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/fw_sharded_rehearsal.py"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
from scipy.sparse import csr_matrix
sys.path.insert(0, ".")
from xcolumns_amd.distributed import TorchComm, find_classifier_using_fw_sharded, shard_csr
from xcolumns_amd.frank_wolfe import find_classifier_using_fw
from xcolumns_amd.metrics import macro_f1_score_on_conf_matrix

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
rng = np.random.default_rng(3)
shape = os.environ.get("XC_FW_REHEARSAL_SHAPE")   # "n,m,r,k,iters": e.g. BASELINE configs[4]'s 1700000,2800000,50,5,4
if shape:
    from xcolumns_amd.synthetic import make_csr_rows
    n, m, r, k, iters = (int(x) for x in shape.split(","))
    Yp = make_csr_rows(n, m, 0, n, r, seed=20240005, k=k)
    Yt = csr_matrix(((rng.random(Yp.nnz) < Yp.data).astype(np.float32), Yp.indices.copy(), Yp.indptr.copy()), shape=Yp.shape)
else:
    n, m, r, k, iters = 30001, 800, 20, 4, 6
    cols = np.concatenate([np.sort(rng.choice(m, r, replace=False)) for _ in range(n)]).astype(np.int32)
    w = 0.05 + 0.95 * rng.random(m) ** 2
    eta = ((rng.random(n * r) ** 2) * w[cols]).astype(np.float32)
    indptr = (np.arange(n + 1) * r).astype(np.int32)
    Yp = csr_matrix((eta, cols, indptr), shape=(n, m))
    Yt = csr_matrix(((rng.random(n * r) < eta).astype(np.float32), cols.copy(), indptr.copy()), shape=(n, m))
comm = TorchComm()
clf, meta = find_classifier_using_fw_sharded(shard_csr(Yt, world, rank), shard_csr(Yp, world, rank),
                                             macro_f1_score_on_conf_matrix, k, comm, skip_tn=True, max_iters=iters,
                                             return_meta=True)
if rank == 0:
    ref, meta1 = find_classifier_using_fw(Yt, Yp, macro_f1_score_on_conf_matrix, k, skip_tn=True, max_iters=iters, return_meta=True)
    print("utilities", [float(u) for u in meta["utilities"]], flush=True)
    ok = (np.array_equal(clf.a, ref.a) and np.array_equal(clf.b, ref.b) and np.array_equal(clf.p, ref.p)
          and meta["alphas"] == meta1["alphas"] and np.allclose(meta["utilities"], meta1["utilities"], rtol=1e-13))
    print("sharded == single process:", ok, "iters", meta["iters"], "all-reduces", comm.calls, flush=True)
    assert ok
dist.barrier()
dist.destroy_process_group()

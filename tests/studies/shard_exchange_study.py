#!/usr/bin/env python3
"""Row shards on CPU (gloo): per-sweep |utility - sequential oracle| of the sharded BCA for several
numbers of ranks P and exchanges per sweep S, with the checker-backed engine of
tests/test_distributed_gloo.py (every rank sweeps its rows SEQUENTIALLY, so what is measured is the
cross-rank staleness alone).   python tests/studies/shard_exchange_study.py [n m]"""
import os
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def _matrix(n, m, zipf):
    """XC_STUDY_ROWS_GEN=1: the benchmark generator the GPU rehearsal (bca_sharded_rehearsal.py) uses at a custom shape."""
    from xcolumns_amd.synthetic import make_csr, make_csr_rows
    if os.environ.get("XC_STUDY_ROWS_GEN") == "1":
        return make_csr_rows(n, m, 0, n, 50, seed=20240004, zipf=zipf, k=5)
    return make_csr(n, m, 50, seed=20240001, zipf=zipf)


def worker(rank, world, port, n, m, settings, zipf, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.test_distributed_gloo import OracleShardEngine
    from xcolumns_amd.distributed import TorchComm, predict_bca_csr_sharded, shard_csr
    from xcolumns_amd.metrics import binary_f1_score_on_conf_matrix
    from xcolumns_amd.synthetic import make_csr

    Y = _matrix(n, m, zipf)
    comm = TorchComm()
    shard = shard_csr(Y, world, rank)
    for S in settings:
        _, meta = predict_bca_csr_sharded(shard, binary_f1_score_on_conf_matrix, 5, comm, skip_tn=True, seed=13,
                                          max_iters=6, tolerance=-1.0, bca_exchanges=S,
                                          engine_factory=lambda *a: OracleShardEngine(*a))
        if rank == 0:
            q.put((S, meta["utilities"], meta.get("exchanges")))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    from oracle import ref as oref
    from tests.test_distributed_gloo import _free_port
    from xcolumns_amd.synthetic import make_csr

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 12_000
    worlds = [int(x) for x in os.environ.get("XC_STUDY_WORLDS", "2,8").split(",")]
    sets = [x if x == "auto" else int(x) for x in os.environ.get("XC_STUDY_SETTINGS", "1,2,4,8,auto").split(",")]
    for zipf in ((False,) if os.environ.get("XC_STUDY_UNIFORM_ONLY") == "1" else (False, True)):
        Y = _matrix(n, m, zipf)
        metric = oref.make_metric(oref.FBETA, k=5.0, m=float(m))
        _, mo = oref.predict_using_bc_with_0approx(Y, metric, 5, skip_tn=True, seed=13, max_iters=6, tolerance=-1.0)
        uo = np.asarray(mo["utilities"])
        print(f"{n}x{m} zipf={zipf} oracle {uo.tolist()}", flush=True)
        for world in worlds:
            settings = sets
            ctx = mp.get_context("spawn")
            q = ctx.Queue()
            port = _free_port()
            procs = [ctx.Process(target=worker, args=(r, world, port, n, m, settings, zipf, q)) for r in range(world)]
            for p in procs:
                p.start()
            for _ in settings:
                S, us, ex = q.get(timeout=1200)
                d = np.abs(np.asarray(us) - uo)
                ds = " ".join("%.1e" % x for x in d)
                print(f"  P={world} S={S!s:>4}: diff=[{ds}] exchanges={ex}", flush=True)
            for p in procs:
                p.join(timeout=120)

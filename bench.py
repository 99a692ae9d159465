#!/usr/bin/env python3
"""bench.py -- BCA macro-F1 sweeps on synthetic sparse score matrices.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--zipf]

One "step" = one BCA sweep (block_coordinate.py:448-463 of the reference) over
the rank's rows PLUS the sweep-boundary work the reference does every
iteration: the from-scratch recompute of the expected confusion statistics
(:465-467; one all-reduce when N > 1), the utility (:469-476) and its transfer
to the host for the stopping rule.  Inputs (CSR y_proba, the initial top-k
prediction, and the visiting orders of all W+K sweeps, generated with the
reference's own np.random.default_rng stream) are resident in HBM before the
timed region.  Weak scaling: every rank holds the workload's n rows
(global n = N x n); value = N * n * K / max-over-ranks time.

Prints ONE JSON line on rank 0 (see DESIGN.md "measurement" for every field).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

R_NNZ = 50   # stored entries per row (primary)
K = 5
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_row_step(r: int, k: int) -> int:
    """SURVEY.md section 8(d) B_step: indptr 4 + CSR row 8r + three float64 gathers 24r +
    old prediction 4k + new prediction 4k."""
    return 4 + 32 * r + 8 * k


def algorithmic_bytes_per_row_sweep(r: int, k: int) -> int:
    """SURVEY.md section 8(d) / BASELINE.md section 4 B_sweep = 8 + 40r + 12k: the step pass plus the
    from-scratch confusion recompute of the sweep boundary (4 + 8r + 4k).  The sweep kernel does
    both (the recompute is fused into it), so this is the work of ONE launch; it is the figure the
    >= 40 % target is defined on."""
    return 8 + 40 * r + 12 * k


def cpu_baseline(Y, k, seed, budget_s=12.0):
    """The oracle (oracle/, a C restatement of the reference's sequential sweep) timed
    on ONE host core on a bounded sample of the same workload: whole sweeps over
    the first rows of the same matrix until ~budget_s of CPU work."""
    from oracle import ref as oracle_ref

    n, m = Y.shape
    n_s = min(n, 100_000)
    Ys = Y[:n_s]
    metric = oracle_ref.make_metric(oracle_ref.FBETA, k=float(k), m=float(m))
    sweeps, t_total = 0, 0.0
    it = 2
    while t_total < budget_s and sweeps < 64:
        t0 = time.perf_counter()
        _, meta = oracle_ref.predict_using_bc_with_0approx(Ys, metric, k, skip_tn=True, seed=seed, max_iters=it,
                                                            tolerance=-1.0)
        t_total += time.perf_counter() - t0
        sweeps += meta["iters"]
    return {
        "value": n_s * sweeps / t_total,
        "unit": "rows/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{sweeps} sequential sweeps over the first {n_s} rows of the same matrix "
                  f"(incl. top-k init and per-sweep confusion recompute), {t_total:.1f} s, 1 thread",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ns_1Mx500K",
                    help="ns_1Mx500K (the configuration north_star's targets are quoted on; default), "
                         "c2_100Kx30K (BASELINE configs[1]), c3_..., c4_...")
    ap.add_argument("--zipf", action="store_true", help="Zipf(1) label popularity instead of uniform")
    ap.add_argument("--waves", type=int, default=0, help="wavefronts walking the order (0 = product default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from xcolumns_amd import _device as D
    from xcolumns_amd import _lib
    from xcolumns_amd.block_coordinate import BcaCsrEngine, WavePolicy
    from xcolumns_amd.distributed import TorchComm
    from xcolumns_amd.metrics import MetricSpec
    from xcolumns_amd.synthetic import WORKLOADS, make_csr

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # XC_BENCH_BACKEND=gloo + XC_BENCH_ONE_DEVICE=1 rehearse the N > 1 control flow with
    # several ranks on ONE GPU (RCCL refuses duplicate devices); never used for numbers.
    one_device = os.environ.get("XC_BENCH_ONE_DEVICE") == "1"
    torch.cuda.set_device(0 if one_device else local_rank)
    comm = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=os.environ.get("XC_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
        comm = TorchComm()

    n, m = WORKLOADS[args.workload]
    seed = 20240001 + rank
    Y = make_csr(n, m, R_NNZ, seed=seed, zipf=args.zipf, k=K)
    dev = D.require_gpu()
    csr = D.DeviceCSR.from_scipy(Y, dev)
    spec = MetricSpec(base=_lib.XC_M_FBETA)  # macro-F1: binary_f1_score_on_conf_matrix, eps 1e-9
    eng = BcaCsrEngine(csr, K, spec, spec, maximize=True, skip_tn=True, n_total=n * world, comm=comm)
    policy = WavePolicy(n, fixed=args.waves if args.waves > 0 else None, world=world, k=K)   # as _bc_csr builds it
    n_u = n * world   # normalisation of the utility = global row count

    # visiting orders of every sweep, the reference's RNG stream (seed 13), uploaded up front
    total = args.warmup + args.steps
    rng = np.random.default_rng(13 + rank)
    order = np.arange(n)
    orders = torch.empty((total, n), dtype=torch.int32, device=dev)
    for s in range(total):
        rng.shuffle(order)
        orders[s] = torch.from_numpy(order.astype(np.int32)).to(dev)

    eng.init_top()
    eng.reset_state(greedy=False)
    u0 = eng.recompute_utility_sum(n_u) / m

    ev_pairs = []
    waves_used = []
    # events are created up front: only their attachment to the sweep dispatch is in the timed region
    ev_pool = []
    for _ in range(args.steps):
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.call("xc_event_create", ctypes.byref(e0))
        _lib.call("xc_event_create", ctypes.byref(e1))
        ev_pool.append((e0, e1))
    host_t = {"enqueue": 0.0, "wait_result": 0.0}
    pipelined = eng.can_pipeline(n) and not policy.sequential
    NEVER = -1e300  # tolerance: the stopping rule never fires, every run does exactly K sweeps

    def run(first, last, old_sum, timed):
        """Sweeps first..last-1 exactly as block_coordinate.run_bca_sweeps drives them: the stopping
        rule and the wavefront policy are evaluated on the GPU at every boundary, the host enqueues
        sweep j + 1 before it reads the result of sweep j (one D2H of 4 doubles per boundary)."""
        out = []
        if first >= last:
            return out
        if not pipelined:   # bca_waves = 1 (--waves 1): the host-paced exact loop
            changed = None
            for s in range(first, last):
                if timed and os.environ.get("XC_BENCH_NO_EVENTS") != "1":
                    e0, e1 = ev_pool[len(ev_pairs)]
                    _lib.call("xc_bca_time_next_sweep", e0, e1)
                    ev_pairs.append((e0, e1))
                w = policy.next(changed)
                eng.sweep(orders[s], n, w, greedy=False)
                out.append(eng.recompute_utility_sum(n_u))
                changed = eng.rows_changed()
                if timed:
                    waves_used.append(w)
            return out
        eng.pipeline_begin(old_sum, NEVER, float(m), True, policy, policy.next(None))

        def collect(j):
            t = time.perf_counter()
            total, changed, waves, flag = eng.pipeline_result(j)
            if timed:
                host_t["wait_result"] += time.perf_counter() - t
                waves_used.append(waves)
            assert flag == 0, flag
            out.append(total)

        for s in range(first, last):
            t = time.perf_counter()
            if timed and os.environ.get("XC_BENCH_NO_EVENTS") != "1":
                # HIP events attached to the sweep dispatch itself, on the stream it runs on
                e0, e1 = ev_pool[len(ev_pairs)]
                _lib.call("xc_bca_time_next_sweep", e0, e1)
                ev_pairs.append((e0, e1))
            eng.pipeline_step(orders[s], s, n_u)
            if timed:
                host_t["enqueue"] += time.perf_counter() - t
            if s > first:
                collect(s - 1)
        collect(last - 1)
        return out

    u0_sum = u0 * m
    run(0, args.warmup, u0_sum, False)
    # the timed steps are sweeps 1..K of a fresh run: back to the top-k prediction
    # (untimed), so the measured mix of changed / unchanged rows is a real run's
    eng.init_top()
    u0_sum = eng.recompute_utility_sum(n_u)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    utilities = [u / m for u in run(args.warmup, total, u0_sum, True)]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    sweep_ms = []
    for a, b in ev_pairs:
        ms = ctypes.c_float(0.0)
        _lib.call("xc_event_elapsed_ms", a, b, ctypes.byref(ms))
        sweep_ms.append(ms.value)
        _lib.call("xc_event_destroy", a)
        _lib.call("xc_event_destroy", b)
    avg_sweep_s = (sum(sweep_ms) / len(sweep_ms)) / 1e3 if sweep_ms else float("nan")

    if rank == 0:
        b_step = algorithmic_bytes_per_row_step(R_NNZ, K)
        b_sweep = algorithmic_bytes_per_row_sweep(R_NNZ, K)
        achieved = b_sweep * n / avg_sweep_s / 1e9
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same
        # command (tools/profile_bench.sh -> tools/summarize_profile.py), when present
        traffic = None
        tfile = os.path.join(ROOT, "profiles", f"traffic_{args.workload}{'_zipf' if args.zipf else ''}.json")
        if os.path.exists(tfile) and world == 1:
            try:
                traffic = json.load(open(tfile))["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        out = {
            "metric": "BCA iterations/sec x instances (rows/s) at k=5",
            "value": n * world * args.steps / elapsed,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: CSR n={n} rows/GPU, m={m} labels, {R_NNZ} entries/row, "
                            f"{'Zipf(1)' if args.zipf else 'uniform'} label popularity, BCA macro-F1 k={K}, "
                            f"init top-k, skip_tn, float32 scores, float64 statistics",
                "rows_per_gpu": n, "labels": m, "nnz_per_row": R_NNZ, "k": K,
                "concurrent_wavefronts_per_sweep": waves_used,
                "step": "sweep kernel (incl. from-scratch tp/fp recompute) + (all-reduce) + commit/utility "
                        "+ stopping rule + D2H of the result",
            },
            "roofline": {
                "kernel": "bca_sweep_csr_kernel<float,1>",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_row": b_sweep,
                "algorithmic_bytes_note": "SURVEY 8(d) B_sweep = step pass 1644 + fused from-scratch recompute 424",
                "frac_step_pass_only": b_step * n / avg_sweep_s / 1e9 / HBM_PEAK_GBS,
                "frac_whole_step": b_sweep * n * world * args.steps / elapsed / 1e9 / (HBM_PEAK_GBS * world),
                "avg_kernel_ms": avg_sweep_s * 1e3,
            },
            "host_ms_per_step": {"enqueue_sweep_and_boundary": host_t["enqueue"] / args.steps * 1e3,
                                 "wait_for_previous_result": host_t["wait_result"] / args.steps * 1e3},
            "loop": "device-side stopping rule, host one iteration behind" if pipelined else "host-paced (exact)",
            "utility_first_last": [utilities[0], utilities[-1]],
            "utility_top_k": u0,
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU leg runs at N = 1 only
            out["cpu_baseline"] = cpu_baseline(Y, K, seed=13)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- BCA macro-F1 sweeps on synthetic sparse score matrices.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--scaling weak|strong] [--zipf]

One "step" = one BCA sweep (block_coordinate.py:448-463 of the reference) over the rank's rows PLUS
the sweep-boundary work the reference does every iteration: the from-scratch recompute of the
expected confusion statistics (:465-467; one all-reduce when N > 1), the utility (:469-476) and its
transfer to the host for the stopping rule.  Inputs (CSR y_proba, the initial top-k prediction and
the visiting orders of all sweeps -- the reference's np.random.default_rng stream) are resident in
HBM before the timed region.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts
`python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a CHILD (it
never touches the GPU itself and never re-executes), passes rank 0's JSON line on and exits with
the child's code.  Under torch.distributed.run (the driver's own launch) it is a worker.

Scaling: "weak" (default; every rank holds the workload's n rows, global n = N x n) and "strong"
(`--scaling strong`: the workload's n rows split over the N ranks -- BASELINE configs[3]'s
"instance-sharded").  N = 1 is the same run in both.  For N > 1 the weak line also carries a
`strong_scaling` object measured in the same process, so one driver run yields both curves.

Prints ONE JSON line on rank 0 (DESIGN.md "measurement" explains every field).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

R_NNZ = 50   # stored entries per row (primary)
K = 5
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MATRIX_SEED = 20240001
ORDER_SEED = 13


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


# ---------------------------------------------------------------------------
# parent: start the N ranks as children
# ---------------------------------------------------------------------------

def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_children(n_gpus: int) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: run the N ranks as a child process
    tree and relay rank 0's JSON line.  Nothing here imports torch or touches the GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith('{"metric"'):
            line = out
        elif out:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        return 1
    return rc


# ---------------------------------------------------------------------------
# CPU baseline legs (rank 0, N = 1): the oracle timed on the host cores
# ---------------------------------------------------------------------------

def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(Y, k, seed, budget_s=10.0):
    """The oracle (oracle/, a C restatement of the reference's kernels) on the host cores, on a bounded
    sample of the same workload.  (i) The BCA sweep on ONE core -- the reference's row loop is serial
    (block_coordinate.py:448): whole sweeps over the first rows of the same matrix until ~budget_s.
    (ii) The row-independent passes -- top-k and the confusion recompute -- with T = 1, 8 and all host
    cores over rows, the reference under XCOLUMNS_NUMBA_PARALLEL=1 / NUMBA_NUM_THREADS=8
    (experiments/numba_perf_tests.sh:5-6)."""
    from oracle import ref as oracle_ref

    n, m = Y.shape
    n_s = min(n, 100_000)
    Ys = Y[:n_s]
    metric = oracle_ref.make_metric(oracle_ref.FBETA, k=float(k), m=float(m))
    sweeps, t_total = 0, 0.0
    while t_total < budget_s and sweeps < 64:
        t0 = time.perf_counter()
        _, meta = oracle_ref.predict_using_bc_with_0approx(Ys, metric, k, skip_tn=True, seed=seed, max_iters=2,
                                                            tolerance=-1.0)
        t_total += time.perf_counter() - t0
        sweeps += meta["iters"]
    host_cores = os.cpu_count() or 1
    try:
        host_cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    legs = {"topk_rows_per_s": {}, "confusion_rows_per_s": {}}
    n_t = min(n, 1_000_000)
    Yt = Y[:n_t]
    pred = None
    for t in sorted({1, min(8, host_cores), host_cores}):
        t0 = time.perf_counter()
        pred = oracle_ref.predict_top_k_threads(Yt, k, t)
        legs["topk_rows_per_s"][f"T={t}"] = n_t / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        oracle_ref.calculate_confusion_matrix_threads(Yt, pred, t)
        legs["confusion_rows_per_s"][f"T={t}"] = n_t / (time.perf_counter() - t0)
    return {
        "value": n_s * sweeps / t_total,
        "unit": "rows/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{sweeps} sequential sweeps over the first {n_s} rows of the same matrix "
                  f"(incl. top-k init and per-sweep confusion recompute), {t_total:.1f} s, 1 thread",
        "cpu_model": _cpu_model(),
        "host_cores": host_cores,
        "row_parallel_legs": dict(legs, sample=f"one pass over the first {n_t} rows, OpenMP threads over rows "
                                                "(the reference's prange loops); top-k k=5, confusion of that prediction"),
    }


# ---------------------------------------------------------------------------
# worker
# ---------------------------------------------------------------------------

class Timer:
    """K timed sweeps of one engine, repeated; everything the JSON line needs from them."""

    def __init__(self, eng, policy, orders, n_local, n_u, m, comm, world, dev):
        self.eng, self.policy, self.orders = eng, policy, orders
        self.n, self.n_u, self.m, self.comm, self.world, self.dev = n_local, n_u, m, comm, world, dev

    def _barrier(self):
        import torch
        import torch.distributed as dist
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(self, first, last, old_sum, events=None, waves_used=None, host_t=None):
        """Sweeps first..last-1 exactly as block_coordinate.run_bca_sweeps drives them: the stopping rule
        and the wavefront policy are evaluated on the GPU at every boundary, the host enqueues sweep
        j + 1 before it reads the result of sweep j."""
        from xcolumns_amd import _lib
        eng, policy, orders, n, n_u, m = self.eng, self.policy, self.orders, self.n, self.n_u, self.m
        out = []
        if first >= last:
            return out
        pipelined = eng.can_pipeline(n) and not policy.sequential
        NEVER = -1e300  # tolerance: the stopping rule never fires, every run does exactly K sweeps

        def attach(s):
            if events is not None:
                e0, e1 = events[s - first]
                _lib.call("xc_bca_time_next_sweep", e0, e1)   # HIP events attached to the sweep dispatch itself

        if not pipelined:   # bca_waves = 1 (--waves 1): the host-paced exact loop
            changed = None
            for s in range(first, last):
                attach(s)
                w = policy.next(changed)
                eng.sweep(orders[s], n, w, greedy=False)
                out.append(eng.recompute_utility_sum(n_u))
                changed = eng.rows_changed()
                if waves_used is not None:
                    waves_used.append(w)
            return out
        eng.pipeline_begin(old_sum, NEVER, float(m), True, policy, policy.next(None))

        def collect(j):
            t = time.perf_counter()
            total, changed, waves, flag = eng.pipeline_result(j)
            if host_t is not None:
                host_t["wait_result"] += time.perf_counter() - t
            if waves_used is not None:
                waves_used.append(waves)
            assert flag == 0, flag
            out.append(total)

        for s in range(first, last):
            t = time.perf_counter()
            attach(s)
            eng.pipeline_step(orders[s], s, n_u)
            if host_t is not None:
                host_t["enqueue"] += time.perf_counter() - t
            if s > first:
                collect(s - 1)
        collect(last - 1)
        return out

    def measure(self, warmup, steps, repeats):
        """`warmup` untimed sweeps, then `repeats` times: back to the top-k prediction (untimed), barrier,
        EXACTLY `steps` sweeps (sweeps 1..K of a fresh run, the expensive early ones included), barrier."""
        import torch
        import torch.distributed as dist
        from xcolumns_amd import _lib
        eng, n_u, m = self.eng, self.n_u, self.m
        eng.init_top()
        eng.reset_state(greedy=False)
        u0_sum = eng.recompute_utility_sum(n_u)
        self.run(0, warmup, u0_sum)
        no_events = os.environ.get("XC_BENCH_NO_EVENTS") == "1"
        elapsed, kernel_ms, utilities, waves = [], [], None, None
        host_t = {"enqueue": 0.0, "wait_result": 0.0}
        ar_ms = []
        for rep in range(repeats):
            events = None
            if not no_events:
                events = []
                for _ in range(steps):
                    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
                    _lib.call("xc_event_create", ctypes.byref(e0))
                    _lib.call("xc_event_create", ctypes.byref(e1))
                    events.append((e0, e1))
            eng.init_top()
            u0_sum = eng.recompute_utility_sum(n_u)
            if self.comm is not None:
                self.comm.start_timing()
            w_used = []
            self._barrier()
            t0 = time.perf_counter()
            sums = self.run(warmup, warmup + steps, u0_sum, events, w_used, host_t)
            self._barrier()
            dt = time.perf_counter() - t0
            if self.world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            elapsed.append(dt)
            if self.comm is not None:
                ar_ms.append(self.comm.stop_timing())
            if events is not None:
                row = []
                for a, b in events:
                    ms = ctypes.c_float(0.0)
                    _lib.call("xc_event_elapsed_ms", a, b, ctypes.byref(ms))
                    row.append(ms.value)
                    _lib.call("xc_event_destroy", a)
                    _lib.call("xc_event_destroy", b)
                kernel_ms.append(row)
            utilities = [u / m for u in sums]
            waves = w_used
        return {"elapsed": elapsed, "kernel_ms": kernel_ms, "utilities": utilities, "waves": waves,
                "u_top_k": u0_sum / m, "host_t": {k: v / (repeats * steps) * 1e3 for k, v in host_t.items()},
                "allreduce_ms": ar_ms}


def _median(xs):
    return float(np.median(np.asarray(xs, dtype=np.float64)))


def roofline_of(kernel_ms, n_rows, by_sweep=True):
    """Fractions of the HBM roofline from per-launch kernel times [repeat][sweep] (ms)."""
    b_step, b_sweep = algorithmic_bytes_per_row_step(R_NNZ, K), algorithmic_bytes_per_row_sweep(R_NNZ, K)
    a = np.asarray(kernel_ms, dtype=np.float64)
    avg_s = float(a.mean()) / 1e3
    achieved = b_sweep * n_rows / avg_s / 1e9
    out = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "avg_kernel_ms": avg_s * 1e3,
           "frac_step_pass_only": b_step * n_rows / avg_s / 1e9 / HBM_PEAK_GBS}
    if by_sweep:
        med = np.median(a, axis=0)
        out["kernel_ms_by_sweep"] = [round(float(x), 4) for x in med]
        out["frac_by_sweep"] = [round(float(b_sweep * n_rows / (x / 1e3) / 1e9 / HBM_PEAK_GBS), 4) for x in med]
    return out


def worker(args):
    import torch
    import torch.distributed as dist

    from xcolumns_amd import _device as D
    from xcolumns_amd import _lib
    from xcolumns_amd.block_coordinate import BcaCsrEngine, WavePolicy
    from xcolumns_amd.distributed import TorchComm, local_order, shard_bounds
    from xcolumns_amd.metrics import MetricSpec
    from xcolumns_amd.synthetic import WORKLOADS, make_csr_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}, or without it (bench.py starts the ranks itself)")
    # XC_BENCH_BACKEND=gloo + XC_BENCH_ONE_DEVICE=1 rehearse the N > 1 control flow with several ranks on
    # ONE GPU (RCCL refuses duplicate devices); never used for numbers.
    one_device = os.environ.get("XC_BENCH_ONE_DEVICE") == "1"
    torch.cuda.set_device(0 if one_device else local_rank)
    dev = D.require_gpu()
    comm, backend = None, None
    if world > 1 or os.environ.get("XC_BENCH_FORCE_DIST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("XC_BENCH_BACKEND", "nccl")
        kw = {"device_id": dev} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        comm = TorchComm()
        backend = dist.get_backend()

    n, m = WORKLOADS[args.workload]
    spec = MetricSpec(base=_lib.XC_M_FBETA)  # macro-F1: binary_f1_score_on_conf_matrix, eps 1e-9
    total_sweeps = args.warmup + args.steps
    extras = not args.no_extras and world == 1

    parity = args.parity or os.environ.get("XCOLUMNS_BCA_PARITY", "per_sweep")
    matrices = {}

    def build(scaling, zipf, use_shadow=None, packed=True, parity=parity):
        """(engine, policy, orders, local rows, global rows, host matrix) of this rank for one mode."""
        n_global = n * world if scaling == "weak" else n
        lo, hi = shard_bounds(n_global, world, rank)
        key = (scaling if world > 1 else "weak", bool(zipf))
        if key not in matrices:      # the legs of one run share the host matrix and its copy in HBM
            matrices.clear()
            if scaling == "weak":
                # rank r holds the whole n-row workload matrix, generated from its own seeds
                Yh = make_csr_rows(n, m, 0, n, R_NNZ, seed=MATRIX_SEED + 8 * rank, zipf=zipf, k=K)
            else:
                Yh = make_csr_rows(n, m, lo, hi, R_NNZ, seed=MATRIX_SEED, zipf=zipf, k=K)
            matrices[key] = (Yh, D.DeviceCSR.from_scipy(Yh, dev))
        Y, csr = matrices[key]
        if not packed:
            os.environ["XCOLUMNS_BCA_PACKED"] = "0"
        try:
            eng = BcaCsrEngine(csr, K, spec, spec, maximize=True, skip_tn=True, n_total=n_global, comm=comm,
                               use_shadow=use_shadow)
        finally:
            if not packed:
                os.environ.pop("XCOLUMNS_BCA_PACKED", None)
        if args.exchanges:
            eng.exchanges = max(1, args.exchanges)
        n_local = hi - lo
        policy = WavePolicy(n_local, fixed=args.waves if args.waves > 0 else None, world=world, k=K, m=m,
                            row_nnz=R_NNZ, skewed=eng.skewed, parity=parity, sweeps=args.steps)   # as _bc_csr builds it
        # the reference's visiting order over the GLOBAL rows (np.random.default_rng(seed), cumulative
        # shuffles, block_coordinate.py:413-419), restricted to this rank's block
        rng = np.random.default_rng(ORDER_SEED)
        order = np.arange(n_global)
        orders = torch.empty((total_sweeps, n_local), dtype=torch.int32, device=dev)
        for s in range(total_sweeps):
            rng.shuffle(order)
            orders[s] = torch.from_numpy(local_order(order, lo, hi)).to(dev)
        return eng, policy, orders, n_local, n_global, Y

    def one_mode(scaling, zipf, repeats, use_shadow=None, packed=True, parity=parity):
        eng, policy, orders, n_local, n_global, Y = build(scaling, zipf, use_shadow, packed, parity)
        if comm is not None:
            comm.bytes_reduced = comm.calls = 0
        t = Timer(eng, policy, orders, n_local, n_global, m, comm, world, dev)
        res = t.measure(args.warmup, args.steps, repeats)
        used = [int(x) for x in getattr(eng, "exchanges_used", [])[-args.steps:]] or [1] * args.steps
        res.update(n_local=n_local, n_global=n_global, Y=Y, eng=eng,
                   pipelined=eng.can_pipeline(n_local) and not policy.sequential,
                   exchanges=used, hot=eng.hot_labels is not None)
        if comm is not None:
            per_sweep = max(1, (args.warmup + args.steps * repeats))
            extra = (sum(used) / len(used) - 1.0) if eng.shadow is not None else 0.0
            n_exch = sum(used) / len(used) - 1.0 if eng.shadow is not None else 0.0
            ar = _median([x / args.steps for x in res["allreduce_ms"]]) if res["allreduce_ms"] else None
            res["comm"] = {"backend": backend, "world_size": dist.get_world_size(),
                           "world_size_seen": dist.get_world_size(), "devices_visible": torch.cuda.device_count(),
                           "exchanges_per_sweep": used,
                           "boundary_all_reduce_bytes": (2 * m + 1) * 8, "mid_sweep_exchange_bytes": 2 * m * 4,
                           "collectives_per_sweep": 1.0 + n_exch,
                           "ms_per_collective": (ar / (1.0 + n_exch)) if ar is not None else None,
                           "all_reduce_bytes_per_sweep": (2 * m + 1) * 8 + extra * 2 * m * 4,
                           "all_reduce_bytes_note": "boundary: 2m+1 float64 (from-scratch tp/fp + changed-row count); each "
                                                    "mid-sweep exchange: 2m float32 (what the rank's rows changed in the records)",
                           "all_reduce_calls_total": comm.calls, "all_reduce_bytes_total": comm.bytes_reduced,
                           "all_reduce_ms_per_sweep": (_median([x / args.steps for x in res["allreduce_ms"]])
                                                       if res["allreduce_ms"] else None),
                           "sweeps_run": per_sweep}
        return res

    main = one_mode(args.scaling, args.zipf, args.repeats)
    strong = None
    if world > 1 and args.scaling == "weak" and not args.no_extras:
        main["eng"].close()
        del main["eng"]
        torch.cuda.empty_cache()
        strong = one_mode("strong", args.zipf, max(1, min(args.repeats, 3)))
        strong["eng"].close()
        del strong["eng"]
        torch.cuda.empty_cache()
        if not args.no_sharded_parity:
            try:
                strong["comm"]["sharded_vs_oracle"] = sharded_parity_leg(args, comm, rank, world, matrices[("strong", bool(args.zipf))][0])
            except Exception as e:  # identical on every rank up to the oracle call, which only rank 0 makes
                strong["comm"]["sharded_vs_oracle"] = {"error": repr(e)}

    out = None
    if rank == 0:
        b_sweep = algorithmic_bytes_per_row_sweep(R_NNZ, K)
        el = main["elapsed"]
        med = _median(el)
        n_local, n_global = main["n_local"], main["n_global"]
        roof = {"kernel": "bca_sweep_csr_kernel<float,1,false,true,true,true,%s> (one launch = one sweep of the rank's rows, "
                          "from-scratch recompute fused%s)" % ("true" if main["hot"] else "false",
                                                              "; N > 1: a sweep is walked in parts with exchanges of the ranks' changes in between -- "
                                                              "the timed span runs from the first part's dispatch to the end of the last one" if world > 1 else ""),
                "bound": "hbm"}
        if main["kernel_ms"]:
            roof.update(roofline_of(main["kernel_ms"], n_local))
        else:
            roof.update({"achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None})
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
        # (tools/profile_bench.sh -> tools/summarize_profile.py), when present
        traffic = None
        tfile = os.path.join(ROOT, "profiles", f"traffic_{args.workload}{'_zipf' if args.zipf else ''}.json")
        if os.path.exists(tfile) and world == 1:
            try:
                traffic = json.load(open(tfile))["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        roof.update({
            "traffic": traffic,
            "algorithmic_bytes_per_row": b_sweep,
            "algorithmic_bytes_note": "SURVEY 8(d) B_sweep = 8 + 40r + 12k: step pass 4 + 32r + 8k (incl. 24r B of float64 statistic "
                                      "gathers; 1644 at r = 50) + fused from-scratch recompute 4 + 8r + 4k (424); the timed kernel gathers 8-byte float32 records instead "
                                      "(see config.arithmetic, frac_f64_records)",
            "frac_whole_step": b_sweep * n_global * args.steps / med / 1e9 / (HBM_PEAK_GBS * world),
        })
        out = {
            "metric": "BCA iterations/sec x instances (rows/s) at k=5",
            "value": n_global * args.steps / med,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": med / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64 gains/accumulation on f32 gathered records",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: CSR n={n_local} rows/GPU ({n_global} over {world} GPU(s), {args.scaling} scaling), "
                            f"m={m} labels, {R_NNZ} entries/row, {'Zipf(1)' if args.zipf else 'uniform'} label popularity, "
                            f"BCA macro-F1 k={K}, init top-k, skip_tn, float32 scores",
                "arithmetic": "gathered per-label statistics: float32 {tp, fp} shadow records (8 B) + float32 column sum "
                              "streamed with the row; gains, top-k keys, accumulation (acc, tpfp) and utility in float64; "
                              "the reference gathers float64 statistics (types.py:14) -- that variant is roofline.frac_f64_records",
                "rows_per_gpu": n_local, "rows_total": n_global, "labels": m, "nnz_per_row": R_NNZ, "k": K,
                "concurrent_wavefronts_per_sweep": main["waves"],
                "parity_policy": parity,
                "exchanges_per_sweep": main["exchanges"],
                "step": "sweep kernel (incl. from-scratch tp/fp recompute) + (all-reduce) + commit/utility "
                        "+ stopping rule + D2H of the result",
            },
            "repeats": {"n": len(el), "ms_per_step_median": med / args.steps * 1e3,
                        "ms_per_step_min": min(el) / args.steps * 1e3, "ms_per_step_max": max(el) / args.steps * 1e3,
                        "note": "each repeat = reset to the top-k prediction (untimed), barrier, K sweeps, barrier; "
                                "value and ms_per_step are the median repeat"},
            "roofline": roof,
            "host_ms_per_step": {"enqueue_sweep_and_boundary": main["host_t"]["enqueue"],
                                 "wait_for_previous_result": main["host_t"]["wait_result"]},
            "loop": "device-side stopping rule, host one iteration behind" if main["pipelined"] else "host-paced (exact)",
            "utility_by_sweep": main["utilities"],
            "utility_top_k": main["u_top_k"],
        }
        if "comm" in main:
            out["comm"] = main["comm"]
        if strong is not None:
            sm = _median(strong["elapsed"])
            out["strong_scaling"] = {
                "value": strong["n_global"] * args.steps / sm, "unit": "rows/s", "ms_per_step": sm / args.steps * 1e3,
                "rows_per_gpu": strong["n_local"], "rows_total": strong["n_global"],
                "roofline_frac_per_gpu": (roofline_of(strong["kernel_ms"], strong["n_local"])["frac"]
                                          if strong["kernel_ms"] else None),
                "kernel_ms_by_sweep": (roofline_of(strong["kernel_ms"], strong["n_local"])["kernel_ms_by_sweep"]
                                       if strong["kernel_ms"] else None),
                "comm": strong.get("comm"),
                "note": "the SAME n-row workload matrix split over the ranks (BASELINE configs[3] style); "
                        "speed-up = this value / the N=1 value",
            }

    if extras and rank == 0:
        Y = main["Y"]
        main["eng"].close()
        del main["eng"]
        torch.cuda.empty_cache()
        # (1) the same loop on the reference's float64 records: no float32 shadow, no packed float32 stream
        try:
            f64 = one_mode(args.scaling, args.zipf, 1, use_shadow=False, packed=False)
            r64 = roofline_of(f64["kernel_ms"], f64["n_local"])
            out["roofline"]["frac_f64_records"] = r64["frac"]
            out["roofline"]["f64_records"] = {"avg_kernel_ms": r64["avg_kernel_ms"], "frac_by_sweep": r64["frac_by_sweep"],
                                              "value": f64["n_global"] * args.steps / f64["elapsed"][0],
                                              "kernel": "bca_sweep_csr_kernel<float,1,false,true,false,false,false>: 16-byte "
                                                        "float64 {tp, fp} records gathered, float64 s streamed (XCOLUMNS_BCA_SHADOW=0 "
                                                        "XCOLUMNS_BCA_PACKED=0)",
                                              "utility_by_sweep": f64["utilities"]}
            f64["eng"].close()
            del f64
            torch.cuda.empty_cache()
        except Exception as e:  # an extra must never cost the main line
            out["roofline"]["frac_f64_records"] = None
            out["roofline"]["f64_records_error"] = repr(e)
        # (2) BASELINE.md section 4's quantity: ONE public call, fixed number of sweeps
        try:
            out["api_call"] = api_call_leg(Y, args.steps)
            out["roofline"]["frac_api_call"] = out["api_call"]["device_resident"]["roofline_frac"]
        except Exception as e:
            out["api_call"] = {"error": repr(e)}
        # (3) the same workload with Zipf(1) label popularity (real XMLC data is long-tailed)
        if not args.zipf and not args.no_zipf:
            try:
                z = one_mode(args.scaling, True, max(1, min(args.repeats, 3)))
                rz = roofline_of(z["kernel_ms"], z["n_local"])
                zm = _median(z["elapsed"])
                out["zipf"] = {"value": z["n_global"] * args.steps / zm, "unit": "rows/s", "ms_per_step": zm / args.steps * 1e3,
                               "roofline_frac": rz["frac"], "avg_kernel_ms": rz["avg_kernel_ms"],
                               "kernel_ms_by_sweep": rz["kernel_ms_by_sweep"], "frac_by_sweep": rz["frac_by_sweep"],
                               "concurrent_wavefronts_per_sweep": z["waves"], "utility_by_sweep": z["utilities"],
                               "workload": f"{args.workload} with Zipf(1) label popularity over a random permutation of the labels"}
                z["eng"].close()
                del z
                torch.cuda.empty_cache()
                if parity == "per_sweep":
                    # the same with bca_parity="final": the first sweep -- every row changes in it -- runs four times
                    # wider and sits a few 1e-5 from the sequential reference; from sweep 2 on both agree with it
                    zf = one_mode(args.scaling, True, 1, parity="final")
                    rf = roofline_of(zf["kernel_ms"], zf["n_local"])
                    out["zipf"]["with_final_parity"] = {
                        "value": zf["n_global"] * args.steps / zf["elapsed"][0], "roofline_frac": rf["frac"],
                        "kernel_ms_by_sweep": rf["kernel_ms_by_sweep"], "concurrent_wavefronts_per_sweep": zf["waves"]}
                    zf["eng"].close()
                    del zf
                    torch.cuda.empty_cache()
            except Exception as e:
                out["zipf"] = {"error": repr(e)}
        # (5) the shapes whose default sweeps are the EXACT ones (the ordered parallel sweep, csrc/xc_bca_ord.hip): BASELINE
        #     configs[2] (about one predicted row per label) and configs[1] with Zipf(1) labels -- ONE public call each
        for key, wl, zf, sw in (("c3", "c3_amazon670k_150Kx670K", False, args.steps), ("c2_zipf", "c2_100Kx30K", True, 5)):
            if args.workload == wl and bool(args.zipf) == zf:
                continue
            try:
                out[key] = shape_leg(wl, zf, sw)
            except Exception as e:
                out[key] = {"error": repr(e)}
        # (4) per-sweep |utility - sequential oracle| of the default policy on configs[1] (C2: the oracle takes seconds)
        try:
            out["parity_c2"] = parity_leg()
        except Exception as e:
            out["parity_c2"] = {"error": repr(e)}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(Y, K, seed=ORDER_SEED)
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(main["Y"], K, seed=ORDER_SEED)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def sharded_parity_leg(args, comm, rank, world, Y_shard, sweeps=2):
    """The public sharded call (xcolumns_amd.distributed.predict_bca_csr_sharded, default exchange schedule and width
    policy) on the SAME workload matrix split over the ranks, `sweeps` sweeps from top-k; rank 0 then runs the
    sequential oracle on the whole matrix with the same seed: |utility difference| after every sweep (bar 1e-5)."""
    from xcolumns_amd.distributed import predict_bca_csr_sharded
    from xcolumns_amd.metrics import binary_f1_score_on_conf_matrix
    from xcolumns_amd.synthetic import WORKLOADS, make_csr_rows

    n, m = WORKLOADS[args.workload]
    t0 = time.perf_counter()
    _, meta = predict_bca_csr_sharded(Y_shard, binary_f1_score_on_conf_matrix, K, comm, skip_tn=True, seed=ORDER_SEED,
                                      max_iters=sweeps, tolerance=-1.0)
    t_call = time.perf_counter() - t0
    out = {"workload": f"{args.workload} split over {world} ranks, {sweeps} sweeps, seed {ORDER_SEED}, default policy and exchange schedule",
           "bar": 1e-5, "utility_by_sweep": [float(u) for u in meta["utilities"]], "exchanges_per_sweep": meta.get("exchanges"),
           "call_s": t_call}
    if rank == 0:
        from oracle import ref as oracle_ref   # checker only, outside every timed region
        Y = make_csr_rows(n, m, 0, n, R_NNZ, seed=MATRIX_SEED, zipf=args.zipf, k=K)
        metric = oracle_ref.make_metric(oracle_ref.FBETA, k=float(K), m=float(m))
        t0 = time.perf_counter()
        _, mo = oracle_ref.predict_using_bc_with_0approx(Y, metric, K, skip_tn=True, seed=ORDER_SEED, max_iters=sweeps, tolerance=-1.0)
        d = np.abs(np.asarray(meta["utilities"]) - np.asarray(mo["utilities"]))
        out.update(abs_diff_by_sweep=[float(x) for x in d], max=float(d.max()), oracle_s=time.perf_counter() - t0)
    return out


def api_call_leg(Y, sweeps):
    """rows/s of ONE predict_optimizing_macro_f1_score_using_bc(y_proba, k, tolerance < 0, max_iters = K) call:
    BASELINE.md section 4's definition of the metric (wall time of the call).  From a scipy matrix in host memory
    (upload over PCIe included) and from a matrix already resident in HBM (xcolumns_amd.DeviceCSR), at K, 10 and 20
    sweeps; visiting orders: the reference's numpy stream generated on the GPU (the drop-in default,
    csrc/xc_order_dev.hip), the same stream walked on the host (two worker threads, XCOLUMNS_ORDER_DEVICE=0) and
    torch.randperm (order_backend="device": another stream, what the loop costs without numpy's sequential walk)."""
    import torch

    from xcolumns_amd import _device as D
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f

    n = Y.shape[0]

    def timed(inp, k_sweeps, reps=5, **kw):
        times = []
        for _ in range(reps + 1):      # the first call warms the allocator / starts the order machinery: not reported
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, meta = f(inp, K, tolerance=-1.0, max_iters=k_sweeps, seed=ORDER_SEED, return_meta=True, **kw)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        assert meta["iters"] == k_sweeps
        med = float(np.median(times[1:]))
        return {"ms": med * 1e3, "rows_per_s": n * k_sweeps / med, "ms_min": min(times[1:]) * 1e3,
                "ms_max": max(times[1:]) * 1e3, "calls": reps,
                "roofline_frac": algorithmic_bytes_per_row_sweep(R_NNZ, K) * n * k_sweeps / med / 1e9 / HBM_PEAK_GBS}

    res = {"sweeps": sweeps, "order": "numpy default_rng stream (drop-in default): generated on the GPU or walked by two host threads, "
                                      "whichever is faster on the machine (block_coordinate._orders_on_device)"}
    Yd = D.DeviceCSR.from_scipy(Y)
    res["host_csr_matrix"] = timed(Y, sweeps, reps=3)
    timed(Yd, sweeps, reps=2)        # settle after the uploads of the leg above (its first calls scattered: 10-23 ms for a 10 ms call)
    by = {}
    from xcolumns_amd.block_coordinate import _orders_on_device
    res["order_default_on_this_machine"] = "gpu" if _orders_on_device(n) else "host threads"
    for k_sweeps in sorted({10, 20}):
        row = {"default": timed(Yd, k_sweeps)}
        for name, flag in (("numpy_stream_on_gpu", "1"), ("numpy_stream_on_host_threads", "0")):
            os.environ["XCOLUMNS_ORDER_DEVICE"] = flag
            try:
                row[name] = timed(Yd, k_sweeps, reps=3)
            finally:
                os.environ.pop("XCOLUMNS_ORDER_DEVICE", None)
        row["torch_randperm"] = timed(Yd, k_sweeps, reps=3, order_backend="device")
        by[str(k_sweeps)] = row
    res["device_resident_by_sweeps"] = by
    res["device_resident"] = by[str(sweeps)]["default"] if str(sweeps) in by else timed(Yd, sweeps)
    return res


def shape_leg(workload, zipf, sweeps, oracle_sweeps=3):
    """ONE public predict_optimizing_macro_f1_score_using_bc call (default policy, matrix resident in HBM, the
    reference's numpy visiting order) on another shape: wall time, rows/s, which sweep ran how (wavefronts; the exact
    sweeps' windows and iterations), and |utility - sequential oracle| for the first sweeps."""
    import torch

    from oracle import ref as oracle_ref
    from xcolumns_amd import _device as D
    from xcolumns_amd import block_coordinate as bc
    from xcolumns_amd.synthetic import WORKLOADS, make_csr_rows

    n, m = WORKLOADS[workload]
    Y = make_csr_rows(n, m, 0, n, R_NNZ, seed=MATRIX_SEED, zipf=zipf, k=K)
    Yd = D.DeviceCSR.from_scipy(Y)
    f = bc.predict_optimizing_macro_f1_score_using_bc
    seen = []
    orig = bc.BcaCsrEngine.sweep_ordered

    def spy(self, order, n_order):
        orig(self, order, n_order)
        seen.append(dict(self.ordered_stats))

    times, meta = [], None
    bc.BcaCsrEngine.sweep_ordered = spy
    try:
        for _ in range(4):
            seen.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            P, meta = f(Yd, K, tolerance=-1.0, max_iters=sweeps, seed=ORDER_SEED, return_meta=True, bca_diagnostics=True)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    finally:
        bc.BcaCsrEngine.sweep_ordered = orig
    med = float(np.median(times[1:]))
    res = {"workload": f"{workload}{' with Zipf(1) labels' if zipf else ''}: n={n}, m={m}, {R_NNZ} entries/row, k={K}, {sweeps} sweeps, "
                       "default policy, one public call on a matrix resident in HBM",
           "ms": med * 1e3, "rows_per_s": n * sweeps / med, "ms_min": min(times[1:]) * 1e3, "ms_max": max(times[1:]) * 1e3,
           "roofline_frac_of_the_call": algorithmic_bytes_per_row_sweep(R_NNZ, K) * n * sweeps / med / 1e9 / HBM_PEAK_GBS,
           "wavefronts_per_sweep": meta.get("wavefronts"),
           "exact_sweeps": [{"window_rows": s_["window"], "windows": s_["windows"], "iterations": s_["iterations"],
                             "kernel_ms": s_["kernel_us"] / 1e3, "barrier_ms": s_["barrier_us"] / 1e3} for s_ in seen],
           "utility_by_sweep": meta["utilities"]}
    q = min(sweeps, oracle_sweeps)
    metric = oracle_ref.make_metric(oracle_ref.FBETA, k=float(K), m=float(m))
    Po, mo = oracle_ref.predict_using_bc_with_0approx(Y, metric, K, skip_tn=True, seed=ORDER_SEED, max_iters=q, tolerance=-1.0)
    d = np.abs(np.asarray(meta["utilities"][:q]) - np.asarray(mo["utilities"]))
    res["abs_diff_vs_oracle_by_sweep"] = [float(x) for x in d]
    if sweeps == q:
        res["prediction_equals_oracle"] = bool(np.array_equal(P.indices.cpu().numpy(), Po.indices))
    return res


def parity_leg():
    """Default policy against the sequential oracle on BASELINE configs[1] (100K x 30K, uniform and Zipf),
    5 sweeps: |utility difference| after every sweep."""
    from oracle import ref as oracle_ref
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f
    from xcolumns_amd.synthetic import WORKLOADS, make_csr

    n, m = WORKLOADS["c2_100Kx30K"]
    out = {"workload": "c2_100Kx30K, 5 sweeps, seed 13, default policy", "bar": 1e-5}
    for name, zipf in (("uniform", False), ("zipf", True)):
        Y = make_csr(n, m, R_NNZ, seed=MATRIX_SEED, zipf=zipf, k=K)
        metric = oracle_ref.make_metric(oracle_ref.FBETA, k=float(K), m=float(m))
        _, mo = oracle_ref.predict_using_bc_with_0approx(Y, metric, K, skip_tn=True, seed=ORDER_SEED, max_iters=5, tolerance=-1.0)
        _, mg = f(Y, K, tolerance=-1.0, max_iters=5, seed=ORDER_SEED, return_meta=True)
        d = np.abs(np.asarray(mg["utilities"]) - np.asarray(mo["utilities"]))
        out[name] = {"abs_diff_by_sweep": [float(x) for x in d], "max": float(d.max())}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ns_1Mx500K",
                    help="ns_1Mx500K (the configuration north_star's targets are quoted on; default), "
                         "c2_100Kx30K (BASELINE configs[1]), c3_..., c4_...")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank holds the workload's n rows; strong: the n rows are split over the ranks")
    ap.add_argument("--zipf", action="store_true", help="Zipf(1) label popularity instead of uniform")
    ap.add_argument("--entries", type=int, default=50,
                    help="stored entries per row: 50 (SURVEY 8d primary) or 100 (the reference's *_100_* prediction files)")
    ap.add_argument("--waves", type=int, default=0, help="wavefronts walking the order (0 = product default)")
    ap.add_argument("--repeats", type=int, default=5, help="repetitions of the K-sweep timed region (median reported)")
    ap.add_argument("--exchanges", type=int, default=0, help="N > 1: exchanges of the ranks' changes per sweep (0 = product default)")
    ap.add_argument("--parity", choices=("per_sweep", "final"), default=None,
                    help="wave policy: per_sweep (product default: every sweep within 1e-5 of the sequential reference) or "
                         "final (wider sweeps, the bar holds after the last sweep)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-zipf", action="store_true", help="skip the Zipf leg of the default run")
    ap.add_argument("--no-sharded-parity", action="store_true",
                    help="N > 1: skip the sharded-vs-oracle leg (two sweeps of the sequential oracle on rank 0, about 40 s)")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the main timed loop (profiling runs): no f64-record / API / Zipf / parity legs, no strong_scaling object")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0 or args.repeats < 1:
        raise SystemExit("need --gpus >= 1, --steps >= 1, --warmup >= 0, --repeats >= 1")
    global R_NNZ
    R_NNZ = int(args.entries)
    if not (K <= R_NNZ <= 1024):
        raise SystemExit("--entries must lie in 5..1024")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_children(args.gpus))
    worker(args)


if __name__ == "__main__":
    main()

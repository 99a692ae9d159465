"""STUDY (test infrastructure, CPU only): rounds an order-respecting parallel BCA sweep would need
(tests/studies/ordered_sim.c).  python tests/studies/ordered_sim.py [config ...]"""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from xcolumns_amd import synthetic  # noqa: E402


class Metric(ctypes.Structure):
    _fields_ = [("eps", ctypes.c_double), ("beta", ctypes.c_double), ("nn", ctypes.c_double)]


def lib():
    out = os.path.join(HERE, "_build", "libordered_sim.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = os.path.join(HERE, "ordered_sim.c")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", out, "-lm"], check=True)
    L = ctypes.CDLL(out)
    L.sim_sequential.restype = ctypes.c_int64
    L.sim_replay.restype = ctypes.c_int64
    return L


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def run(name, n, m, zipf, sweeps=4, k=5, r=50, window=8192, seed=13):
    L = lib()
    Y = synthetic.make_csr_rows(n, m, 0, n, r=r, zipf=zipf, k=k)
    indptr, indices, data = Y.indptr.astype(np.int32), Y.indices.astype(np.int32), Y.data.astype(np.float32)
    # top-k start
    d2 = data.reshape(n, r)
    top = np.argpartition(-d2, k, axis=1)[:, :k]
    sel = np.zeros((n, r), dtype=np.uint8)
    np.put_along_axis(sel, top, 1, axis=1)
    sel = sel.reshape(-1)
    colsum = np.bincount(indices, weights=data.astype(np.float64), minlength=m)
    tp = np.bincount(indices, weights=data.astype(np.float64) * sel, minlength=m)
    fp = np.bincount(indices, weights=(1.0 - data).astype(np.float32).astype(np.float64) * sel, minlength=m)
    cnt = np.bincount(indices, minlength=m)
    hot = np.zeros(m, dtype=np.uint8)
    hot_min = max(4096, n // 32)
    hot_ids = np.argsort(-cnt)[:63]
    hot[hot_ids[cnt[hot_ids] >= hot_min]] = 1
    boxed_all = np.ones(m, dtype=np.uint8)
    mid = np.zeros(m, dtype=np.uint8)
    mid[cnt >= max(64, n // 2000)] = 1
    mt = Metric(1e-9, 1.0, float(n))
    rng = np.random.default_rng(seed)
    order = np.arange(n)
    print(f"== {name}: n={n} m={m} zipf={zipf} hot={int(hot.sum())} mid={int(mid.sum())}", flush=True)
    ntr = np.zeros(n, dtype=np.int32)
    trace = np.zeros(n * 2 * k * 3, dtype=np.float64)  # change_t = 24 bytes (int32 + pad, 2 doubles)
    for sw in range(1, sweeps + 1):
        rng.shuffle(order)
        o64 = order.astype(np.int64)
        tp0, fp0, sel0 = tp.copy(), fp.copy(), sel.copy()
        t0 = time.time()
        ch = L.sim_sequential(ctypes.c_int64(n), P(o64), P(indptr), P(indices), P(data), P(sel), k, P(tp), P(fp),
                              P(colsum), ctypes.byref(mt), P(ntr), P(trace))
        line = f"  sweep {sw}: changed {ch / n:.3f} (seq {time.time() - t0:.1f}s) |"
        lv = np.zeros(n, dtype=np.int32)
        L.sim_dag_depth.restype = ctypes.c_int64
        d0 = L.sim_dag_depth(ctypes.c_int64(n), P(o64), P(indptr), P(indices), k, P(ntr), P(trace), ctypes.c_int64(m), None, P(lv))
        line += f" DAG depth {d0} ({n / d0:.0f} rows/level)"
        if hot.any():
            for nm_, ig in (("hot", hot), ("mid", mid)):
                d1 = L.sim_dag_depth(ctypes.c_int64(n), P(o64), P(indptr), P(indices), k, P(ntr), P(trace), ctypes.c_int64(m), P(ig), P(lv))
                line += f", without {nm_} labels {d1} ({n / d1:.0f})"
        line += " |"
        L.sim_fixpoint.restype = ctypes.c_int64
        for win in [int(x) for x in os.environ.get("SIM_FIX_WINDOWS", "1024,4096,16384").split(",")]:
            out = np.zeros(8, dtype=np.int64)
            a, b, c = tp0.copy(), fp0.copy(), sel0.copy()
            it = L.sim_fixpoint(ctypes.c_int64(n), P(o64), P(indptr), P(indices), P(data), P(c), k, P(a), P(b), P(colsum),
                                ctypes.byref(mt), ctypes.c_int64(m), ctypes.c_int64(win), 4, P(out))
            assert np.array_equal(c, sel), "fix-point must end in the sequential prediction"
            line += f" fix/{win}: {it} iters in {out[0]} windows (max {out[1]}, touched evals {out[2]}, max writers/label {out[3]}, >4-writer labels {out[4]})"
        line += " |"
        if os.environ.get("SIM_DAG_ONLY"):
            print(line, flush=True)
            continue
        variants = [("rw", 0, None, 0)]
        if hot.any():
            variants += [("hot-box/prefix", 1, hot, 0), ("hot-box/superset", 1, hot, 1),
                         ("mid-box/prefix", 1, mid, 0), ("mid-box/superset", 1, mid, 1)]
        variants += [("all-box/prefix", 1, boxed_all, 0)]
        for vname, mode, boxed, superset in variants:
            for win in ([window] if not superset else [512, 2048]):
                stats = np.zeros(8, dtype=np.int64)
                hist = np.zeros(32, dtype=np.int32)
                a, b, c = tp0.copy(), fp0.copy(), sel0.copy()
                rounds = L.sim_replay(ctypes.c_int64(n), P(o64), P(indptr), P(indices), P(data), P(c), k, P(a), P(b),
                                      P(colsum), ctypes.byref(mt), P(ntr), P(trace), ctypes.c_int64(m), mode,
                                      P(boxed) if boxed is not None else None, superset, ctypes.c_int64(win), P(stats),
                                      P(hist))
                assert np.array_equal(c, sel), "replay must end in the sequential prediction"
                line += f" {vname}{'/' + str(win) if superset else ''}: {rounds} rounds ({n / rounds:.0f} rows/round; ends rw {stats[0]} box {stats[1]} win {stats[2]})"
        print(line, flush=True)


CONFIGS = {
    "c3": (150_000, 670_000, False),
    "c2": (100_000, 30_000, False),
    "c2z": (100_000, 30_000, True),
    "ns": (1_000_000, 500_000, False),
    "nsz": (1_000_000, 500_000, True),
    "small": (20_000, 5_000, False),
    "smallz": (20_000, 5_000, True),
}

if __name__ == "__main__":
    for nm in (sys.argv[1:] or ["c3", "c2", "c2z"]):
        n, m, z = CONFIGS[nm]
        run(nm, n, m, z)

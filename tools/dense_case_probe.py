"""Two cases of tests/studies/fuzz_exact.py replayed with details (dense exact mode against the oracle): seed 40397, a float64
minimisation driven to a utility of 1e-10 (other zero-probability labels in 12 rows, utilities equal), and seed 30332, a
tn-based metric with skip_tn (ill-conditioned)."""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
sys.argv = ["x", "0", "0"]
src = open("tests/studies/fuzz_exact.py").read().split("cases = int(sys.argv[1])")[0]
g = {"__file__": os.path.abspath("tests/studies/fuzz_exact.py")}
exec(compile(src, "fz", "exec"), g)
oref, bc = g["oref"], g["bc"]
for seed in (40397, 30332):
    rng = np.random.default_rng(seed)
    Y, k = g["problem"](rng)
    n, m = Y.shape
    what = rng.choice(["bca_csr", "bca_csr", "bca_dense", "topk", "confusion", "coverage"])
    name, base = g["METRICS"][int(rng.integers(len(g["METRICS"])))]
    kw = dict(seed=int(rng.integers(1000)), max_iters=int(rng.integers(1, 5)), tolerance=float(rng.choice([-1.0, 1e-6])),
              skip_tn=bool(rng.random() < 0.5), maximize=bool(rng.random() < 0.85),
              metric_aggregation=str(rng.choice(["mean", "sum"])), shuffle_order=bool(rng.random() < 0.8),
              init_y_pred=str(rng.choice(["top", "random", "greedy"])))
    if not kw["maximize"]:
        kw["tolerance"] = abs(kw["tolerance"])
    metric = oref.make_metric(base, k=float(k), m=float(m))
    Yin = Y.toarray()
    Po, mo = oref.predict_using_bc_with_0approx(Yin, metric, k, **kw)
    for rep in range(3):
        Pg, mg = bc.predict_using_bc_with_0approx(Yin, getattr(bc, name), k, return_meta=True, bca_waves=1, **kw)
        diff_rows = np.nonzero((Pg != Po).any(axis=1))[0]
        print(seed, name, kw, "iters", mg["iters"], mo["iters"], "utilities", mg["utilities"], mo["utilities"], "rows that differ", diff_rows[:10], len(diff_rows))
        for r in diff_rows[:2]:
            print("   row", r, "gpu", np.nonzero(Pg[r])[0], "oracle", np.nonzero(Po[r])[0], "values", Yin[r][np.nonzero(Pg[r] != Po[r])[0]])

"""Synthetic sparse score matrices shaped like the benchmark configurations
(SURVEY.md section 8d): CSR, int32 indices sorted ascending and distinct within a row,
exactly `r` stored entries per row, float32 scores in (0, 1) skewed towards small
probabilities, label popularity uniform or Zipf(1) over a random permutation of
the labels.  Used by bench.py, __graft_entry__.smoke() and the tests."""
from __future__ import annotations

import numpy as np
from scipy.sparse import csr_matrix

# BASELINE.json configs (n rows, m labels); r = 50 entries per row, k = 5
WORKLOADS = {
    "c2_100Kx30K": (100_000, 30_000),
    "c3_amazon670k_150Kx670K": (150_000, 670_000),
    "c4_wiki500k_780Kx500K": (780_000, 500_000),
    "ns_1Mx500K": (1_000_000, 500_000),
    "c5_amazon3m_1.7Mx2.8M": (1_700_000, 2_800_000),   # configs[4]'s shape (its workload is Frank-Wolfe: tools/fw_bench.py)
}


def _first_r_distinct(draws: np.ndarray, r: int):
    """For each row of `draws` the first r distinct values in draw order, sorted
    ascending; also a mask of the rows that had at least r distinct values."""
    n, w = draws.shape
    order = np.argsort(draws, axis=1, kind="stable")
    sc = np.take_along_axis(draws, order, axis=1)
    dup_sorted = np.zeros((n, w), dtype=bool)
    dup_sorted[:, 1:] = sc[:, 1:] == sc[:, :-1]
    dup = np.empty_like(dup_sorted)
    np.put_along_axis(dup, order, dup_sorted, axis=1)
    rank = np.cumsum(~dup, axis=1)
    ok = rank[:, -1] >= r
    keep = (~dup) & (rank <= r)
    out = np.full((n, r), -1, dtype=np.int64)
    rows_ok = np.nonzero(ok)[0]
    if rows_ok.size:
        sel = draws[rows_ok][keep[rows_ok]].reshape(rows_ok.size, r)
        out[rows_ok] = np.sort(sel, axis=1)
    return out, ok


def sample_columns(rng: np.random.Generator, n: int, m: int, r: int, zipf: bool = False) -> np.ndarray:
    """(n, r) int64 sorted distinct column ids per row."""
    if r > m:
        raise ValueError("r must be <= m")
    if zipf:
        w = 1.0 / np.arange(1, m + 1, dtype=np.float64)
        cdf = np.cumsum(w / w.sum())
        perm = rng.permutation(m)
    cols = np.full((n, r), -1, dtype=np.int64)
    todo = np.arange(n)
    width = r + max(16, r // 2) if not zipf else 4 * r
    while todo.size:
        if zipf:
            u = rng.random((todo.size, width))
            draws = perm[np.minimum(np.searchsorted(cdf, u), m - 1)]
        else:
            draws = rng.integers(0, m, size=(todo.size, width))
        got, ok = _first_r_distinct(draws, r)
        cols[todo[ok]] = got[ok]
        todo = todo[~ok]
        width = min(width * 2, max(width, 64 * r))
    return cols


def make_csr(n: int, m: int, r: int = 50, seed: int = 20240001, zipf: bool = False, k: int = 5,
             dtype=np.float32, values: str = "cube") -> csr_matrix:
    """The benchmark's y_proba.  `values`: "cube" -> u^3, "sigmoid" -> sigmoid(N(-2, 1.5^2))."""
    rng = np.random.default_rng(seed)
    cols = np.empty((n, r), dtype=np.int32)
    chunk = max(1, min(n, 200_000))
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        cols[s:e] = sample_columns(rng, e - s, m, r, zipf=zipf)
    if values == "cube":
        data = rng.random((n, r)) ** 3
    elif values == "sigmoid":
        data = 1.0 / (1.0 + np.exp(-(rng.normal(-2.0, 1.5, size=(n, r)))))
    else:
        raise ValueError("values must be 'cube' or 'sigmoid'")
    data = data.astype(dtype)
    data = np.clip(data, np.finfo(dtype).tiny * 4, 1.0 - np.finfo(dtype).eps)
    # no tie between the k-th and (k+1)-th largest score of a row (top-k must be unambiguous)
    if 0 < k < r:
        srt = np.sort(data, axis=1)[:, ::-1]
        bad = np.nonzero(srt[:, k - 1] == srt[:, k])[0]
        while bad.size:
            data[bad] = np.clip((rng.random((bad.size, r)) ** (3 if values == "cube" else 1)).astype(dtype),
                                np.finfo(dtype).tiny * 4, 1.0 - np.finfo(dtype).eps)
            srt = np.sort(data[bad], axis=1)[:, ::-1]
            bad = bad[srt[:, k - 1] == srt[:, k]]
    indptr = (np.arange(n + 1, dtype=np.int64) * r).astype(np.int32)
    return csr_matrix((data.reshape(-1), cols.reshape(-1), indptr), shape=(n, m))

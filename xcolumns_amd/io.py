"""On-disk inputs of the reference's experiment drivers (/root/reference/experiments/utils.py:113-228):
XMC-repository label files, libsvm-like sparse prediction files, ``*-labels.npy`` / ``*-scores.npy``
top-k pairs, dense ``.npy`` score matrices and the ``.npz`` cache.  Same function names, arguments and
results (scipy CSR, float32, column ids sorted within a row); :func:`to_device` puts a matrix straight
into HBM for the kernels.

The text formats are parsed in bulk (one tokenisation of the whole file) instead of the reference's
per-token Python loop; dense score matrices are reduced to their top-k on the GPU.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Callable, Optional, Union

import numpy as np
import torch
from scipy.sparse import csr_matrix, load_npz, save_npz

from . import _device as D
from .utils import construct_csr_matrix


def _finish(data, indices, indptr) -> csr_matrix:
    """csr_matrix((data, indices, indptr), dtype=float32), rows sorted only when some row is not
    (experiments/utils.py:152-160)."""
    indices = np.asarray(indices, dtype=np.int64)
    indptr = np.asarray(indptr, dtype=np.int64)
    if indices.size > 1:
        starts = np.zeros(indices.size, dtype=bool)
        starts[indptr[:-1][indptr[:-1] < indices.size]] = True
        requires_sort = bool(((indices[1:] < indices[:-1]) & ~starts[1:]).any())
    else:
        requires_sort = False
    return construct_csr_matrix(np.asarray(data, dtype=np.float64), indices, indptr, dtype=np.float32,
                                sort_indices=requires_sort)


def load_txt_labels(path: str, header: bool = True, labels_delimiter: str = ",",
                    labels_features_delimiter: Optional[str] = " ", labels_map: Optional[dict] = None) -> csr_matrix:
    """The sparse label matrix of an XMC-repository text file: an optional ``<n> <features> <labels>``
    header, then per line ``l1,l2,... f1:v1 f2:v2 ...`` (experiments/utils.py:113-160)."""
    with open(path) as file:
        if header:
            file.readline()
        lines = file.read().split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    indptr = [0]
    tokens = []
    for line in lines:
        labels = line
        if labels_features_delimiter is not None:
            labels = line.split(labels_features_delimiter)[0]
        labels = labels.split(labels_delimiter)
        if not (len(labels) == 1 and labels[0].strip() == ""):
            tokens.extend(labels)
        indptr.append(len(tokens))
    if labels_map is not None:
        indices = np.fromiter((labels_map[t.strip()] for t in tokens), dtype=np.int64, count=len(tokens))
    else:
        indices = np.array(tokens, dtype=np.int64) if tokens else np.zeros(0, dtype=np.int64)
    return _finish(np.ones(indices.size), indices, indptr)


def load_txt_sparse_pred(path: str) -> csr_matrix:
    """A sparse prediction matrix in libsvm-like text, per line ``<label>:<value> <label>:<value> ...``
    (experiments/utils.py:163-186)."""
    with open(path) as file:
        lines = file.read().split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    counts = np.fromiter((len(line.split()) for line in lines), dtype=np.int64, count=len(lines))
    flat = " ".join(lines).replace(":", " ").split()
    pairs = np.array(flat, dtype=np.float64).reshape(-1, 2) if flat else np.zeros((0, 2))
    indptr = np.concatenate([[0], np.cumsum(counts)])
    return _finish(pairs[:, 1], pairs[:, 0].astype(np.int64), indptr)


def load_npy_sparse_pred(path: str) -> csr_matrix:
    """``<path>-labels.npy`` (n x k column ids) + ``<path>-scores.npy`` (n x k values)
    (experiments/utils.py:189-195)."""
    indices = np.load(path + "-labels.npy", allow_pickle=True)
    data = np.load(path + "-scores.npy", allow_pickle=True)
    indptr = np.arange(0, indices.shape[0] + 1, 1, dtype=np.int32) * indices.shape[1]
    return construct_csr_matrix(data.flatten(), indices.flatten(), indptr, dtype=np.float32, sort_indices=True)


def load_npy_full_pred(path: str, keep_top_k: int = 0, **kwargs) -> csr_matrix:
    """A dense n x m score matrix reduced to its ``keep_top_k`` largest scores per row
    (experiments/utils.py:198-210); the selection runs on the GPU."""
    dense = np.load(path, allow_pickle=True)
    if keep_top_k < 0:
        raise ValueError("keep_top_k must be >= 0 (the reference's negative branch reads an undefined variable)")
    n = dense.shape[0]
    if keep_top_k == 0:
        # (the reference's default fails here: scipy cannot infer the shape of an empty matrix)
        return construct_csr_matrix(np.zeros(0, dtype=dense.dtype), np.zeros(0, dtype=np.int64),
                                    np.zeros(n + 1, dtype=np.int32), dtype=np.float32, shape=dense.shape,
                                    sort_indices=True)
    dev = D.require_gpu()
    from . import _lib
    from .weighted_prediction import topk_dense_device
    scores = torch.from_numpy(np.ascontiguousarray(dense)).to(dev)
    if scores.dtype not in (torch.float32, torch.float64):
        scores = scores.to(torch.float32)
    m = scores.shape[1]
    if keep_top_k > m:
        raise ValueError(f"keep_top_k={keep_top_k} exceeds the number of columns {m}")
    # the package's own row top-k (xc_topk_dense: ties go to the lower column) and a compaction of its 0/1 rows
    y_pred = topk_dense_device(scores, keep_top_k, 0.0, False, scores.dtype)
    idx = torch.empty(n * keep_top_k, dtype=torch.int32, device=dev)
    vals = torch.empty(n * keep_top_k, dtype=scores.dtype, device=dev)
    _lib.call("xc_dense_pred_to_fixed", n, m, D.ptr(y_pred), D.dtype_code(y_pred.dtype), D.ptr(scores),
              D.dtype_code(scores.dtype), int(keep_top_k), D.ptr(idx), D.ptr(vals), D.stream())
    indptr = np.arange(0, n + 1, 1, dtype=np.int32) * keep_top_k
    return construct_csr_matrix(vals.cpu().numpy(), idx.cpu().numpy().astype(np.int64), indptr, dtype=np.float32,
                                sort_indices=True)


def load_cache_npz_file(path: Union[str, Path], load_func: Callable, recreate: bool = False, **load_func_args):
    """``<path>.npz`` if it exists, else ``load_func(path, **load_func_args)`` saved there
    (experiments/utils.py:213-226)."""
    path = str(path)
    if not os.path.exists(path + ".npz") or recreate:
        data = load_func(path, **load_func_args)
        save_npz(path + ".npz", data)
    else:
        data = load_npz(path + ".npz")
    return data


def to_device(mat: csr_matrix) -> D.DeviceCSR:
    """A loaded matrix as int32 / float32 arrays in HBM (what the kernels take)."""
    return D.DeviceCSR.from_scipy(mat.tocsr() if not isinstance(mat, csr_matrix) else mat)

"""Device plumbing: PyTorch-ROCm tensors own the HBM buffers and streams; the
kernels get raw pointers through the C ABI."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
from scipy.sparse import csr_matrix

from . import _lib

_TORCH_OF_NP = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}
_NP_OF_TORCH = {torch.float32: np.dtype(np.float32), torch.float64: np.dtype(np.float64)}


def require_gpu() -> torch.device:
    """The product path has no CPU fallback: fail loudly without a GPU or the library."""
    _lib.load()
    if not torch.cuda.is_available():
        raise RuntimeError(
            "xcolumns_amd needs an AMD MI355X (gfx950) visible to PyTorch-ROCm; "
            "no GPU is available and there is no CPU fallback."
        )
    return torch.device("cuda", torch.cuda.current_device())


def dtype_code(dt) -> int:
    if isinstance(dt, torch.dtype):
        if dt == torch.float32:
            return _lib.XC_F32
        if dt == torch.float64:
            return _lib.XC_F64
    else:
        dt = np.dtype(dt)
        if dt == np.float32:
            return _lib.XC_F32
        if dt == np.float64:
            return _lib.XC_F64
    raise ValueError(f"y_proba must hold float32 or float64 values, got {dt}")


def torch_dtype(dt) -> torch.dtype:
    if isinstance(dt, torch.dtype):
        return dt
    return _TORCH_OF_NP[np.dtype(dt)]


def numpy_dtype(dt) -> np.dtype:
    if isinstance(dt, torch.dtype):
        return _NP_OF_TORCH[dt]
    return np.dtype(dt)


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_device(a, dtype=None, device=None) -> torch.Tensor:
    device = device or require_gpu()
    if isinstance(a, torch.Tensor):
        t = a.to(device=device, dtype=dtype) if dtype is not None else a.to(device=device)
    else:
        t = torch.from_numpy(np.ascontiguousarray(a))
        t = t.to(device=device, dtype=dtype) if dtype is not None else t.to(device=device)
    return t.contiguous()


def check_column_ids(indices: torch.Tensor, m: int, what: str) -> None:
    """The kernels index m-length tables with these ids unchecked: refuse ids outside [0, m) here, before
    any launch (one min/max pass on the device; scipy itself does not validate `indices`)."""
    if indices.numel() == 0:
        return
    lo, hi = torch.aminmax(indices)
    lo, hi = int(lo), int(hi)
    if lo < 0 or hi >= m:
        raise ValueError(f"{what}: column ids must lie in [0, {m}), found [{lo}, {hi}]")


@dataclass
class DeviceCSR:
    """A CSR matrix resident in HBM: int32 indptr / indices, float data."""

    indptr: torch.Tensor
    indices: torch.Tensor
    data: torch.Tensor
    shape: tuple
    max_row_nnz: int

    @property
    def n(self) -> int:
        return self.shape[0]

    @property
    def m(self) -> int:
        return self.shape[1]

    @property
    def nnz(self) -> int:
        return int(self.indices.numel())

    @property
    def code(self) -> int:
        return dtype_code(self.data.dtype)

    @staticmethod
    def from_scipy(mat: csr_matrix, device=None) -> "DeviceCSR":
        device = device or require_gpu()
        if mat.nnz >= 2 ** 31:
            raise ValueError("matrices with >= 2^31 stored entries are not supported (int32 offsets)")
        dtype_code(mat.dtype)
        indptr = np.ascontiguousarray(mat.indptr, dtype=np.int32)
        row_nnz = np.diff(indptr)
        if indptr.size != mat.shape[0] + 1 or indptr[0] != 0 or indptr[-1] != mat.indices.size or \
                (row_nnz.size and row_nnz.min() < 0) or mat.data.size != mat.indices.size:
            raise ValueError("malformed csr_matrix: indptr must start at 0, be non-decreasing and end at nnz")
        max_row = int(row_nnz.max()) if mat.shape[0] > 0 else 0
        indices = torch.from_numpy(np.ascontiguousarray(mat.indices, dtype=np.int32)).to(device)
        check_column_ids(indices, int(mat.shape[1]), "csr_matrix")
        return DeviceCSR(
            indptr=torch.from_numpy(indptr).to(device),
            indices=indices,
            data=torch.from_numpy(np.ascontiguousarray(mat.data)).to(device),
            shape=tuple(int(x) for x in mat.shape),
            max_row_nnz=max_row,
        )

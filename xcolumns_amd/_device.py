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


def is_torch_sparse_csr(x) -> bool:
    return isinstance(x, torch.Tensor) and x.layout == torch.sparse_csr


@dataclass
class DeviceCSR:
    """A CSR matrix resident in HBM: int32 indptr / indices, float data.

    Also an INPUT / OUTPUT type of the public functions (the reference's type seam is types.py:10-23:
    ndarray, csr_matrix, torch.Tensor): pass a DeviceCSR -- or a torch ``sparse_csr`` tensor -- as
    `y_proba` and the call neither uploads nor downloads the matrix; the prediction comes back as the
    same kind of object, resident in HBM.  Rows must hold sorted, distinct column ids (as the
    reference requires of a csr_matrix, numba_csr_functions.py:121)."""

    indptr: torch.Tensor
    indices: torch.Tensor
    data: torch.Tensor
    shape: tuple
    max_row_nnz: int
    min_row_nnz: int = 0

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

    @property
    def dtype(self) -> np.dtype:
        return numpy_dtype(self.data.dtype)

    @property
    def ndim(self) -> int:
        return 2

    # -- derived quantities that depend on this matrix alone, computed once (a DeviceCSR is a read-only view: whoever
    # rewrites its tensors in place calls forget_cached()) --------------------------------------------------------
    def forget_cached(self) -> None:
        self.__dict__.pop("_colsum64", None)
        self.__dict__.pop("_ascending", None)

    def column_sums(self) -> torch.Tensor:
        """float64[m] sums of the stored values per column (bucketed counting sort + LDS sums for float32 values,
        xc_scatter_sum_f32; one float64 atomic per entry otherwise)."""
        cs = self.__dict__.get("_colsum64")
        if cs is None:
            import ctypes
            from . import _lib
            cs = torch.zeros(self.m, dtype=torch.float64, device=self.data.device)
            if self.nnz > 0:
                if self.data.dtype == torch.float32 and self.nnz >= (1 << 18) and self.m <= 16384 * 2048:
                    nbytes = ctypes.c_int64(0)
                    _lib.call("xc_scatter_sum_workspace_bytes", self.nnz, self.m, ctypes.byref(nbytes))
                    ws = torch.empty(int(nbytes.value), dtype=torch.uint8, device=self.data.device)
                    _lib.call("xc_scatter_sum_f32", self.nnz, ptr(self.indices), ptr(self.data), self.m, 0, ptr(cs), ptr(ws),
                              stream())
                    ws.record_stream(torch.cuda.current_stream())
                else:
                    _lib.call("xc_bca_colsum_csr", self.nnz, ptr(self.indices), ptr(self.data), self.code, ptr(cs), stream())
            self.__dict__["_colsum64"] = cs
        return cs

    def rows_ascending(self) -> bool:
        """Strictly ascending column ids in every row (what the reference requires of a csr_matrix)."""
        asc = self.__dict__.get("_ascending")
        if asc is None:
            from . import _lib
            flag = torch.zeros(1, dtype=torch.int32, device=self.data.device)
            _lib.call("xc_csr_rows_ascending", self.n, ptr(self.indptr), ptr(self.indices), ptr(flag), stream())
            asc = int(flag.item()) == 0
            self.__dict__["_ascending"] = asc
        return asc

    @staticmethod
    def from_parts(indptr: torch.Tensor, indices: torch.Tensor, data: torch.Tensor, shape, device=None,
                   check: bool = True) -> "DeviceCSR":
        """From (indptr, indices, data) tensors (any device / integer width; moved to the GPU and narrowed
        to int32 if needed).  `check`: structure and column-id range are verified on the device (the kernels
        index m-length tables with the ids unchecked); costs one small synchronisation."""
        device = device or require_gpu()
        n, m = int(shape[0]), int(shape[1])
        if indices.numel() >= 2 ** 31:
            raise ValueError("matrices with >= 2^31 stored entries are not supported (int32 offsets)")
        dtype_code(data.dtype)
        indptr = indptr.to(device=device, dtype=torch.int32).contiguous()
        indices = indices.to(device=device, dtype=torch.int32).contiguous()
        data = data.to(device=device).contiguous()
        if indptr.numel() != n + 1 or data.numel() != indices.numel():
            raise ValueError("malformed CSR: indptr must have n + 1 entries and data / indices equal lengths")
        max_row = min_row = 0
        if n > 0:
            row_nnz = indptr[1:] - indptr[:-1]
            lo, hi = torch.aminmax(row_nnz)
            stats = torch.stack([lo, hi, indptr[0], indptr[-1]]).cpu().numpy()   # one D2H for all four
            min_row, max_row = int(stats[0]), int(stats[1])
            if check and (min_row < 0 or int(stats[2]) != 0 or int(stats[3]) != indices.numel()):
                raise ValueError("malformed CSR: indptr must start at 0, be non-decreasing and end at nnz")
        if check:
            check_column_ids(indices, m, "sparse matrix")
        return DeviceCSR(indptr, indices, data, (n, m), max_row, min_row)

    @staticmethod
    def from_torch(t: torch.Tensor, device=None) -> "DeviceCSR":
        """From a torch ``sparse_csr`` tensor (CPU or GPU)."""
        if not is_torch_sparse_csr(t) or t.dim() != 2:
            raise ValueError("expected a 2-d torch tensor with layout torch.sparse_csr")
        return DeviceCSR.from_parts(t.crow_indices(), t.col_indices(), t.values(), t.shape, device)

    def to_torch(self, index_dtype: torch.dtype = torch.int32) -> torch.Tensor:
        return torch.sparse_csr_tensor(self.indptr.to(index_dtype), self.indices.to(index_dtype), self.data,
                                       size=self.shape)

    def to_scipy(self) -> csr_matrix:
        return csr_matrix((self.data.cpu().numpy(), self.indices.cpu().numpy(), self.indptr.cpu().numpy()),
                          shape=self.shape)

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
        min_row = int(row_nnz.min()) if mat.shape[0] > 0 else 0
        indices = torch.from_numpy(np.ascontiguousarray(mat.indices, dtype=np.int32)).to(device)
        check_column_ids(indices, int(mat.shape[1]), "csr_matrix")
        return DeviceCSR(
            indptr=torch.from_numpy(indptr).to(device),
            indices=indices,
            data=torch.from_numpy(np.ascontiguousarray(mat.data)).to(device),
            shape=tuple(int(x) for x in mat.shape),
            max_row_nnz=max_row,
            min_row_nnz=min_row,
        )


def is_device_sparse(x) -> bool:
    """A sparse matrix the public functions take without a host round trip."""
    return isinstance(x, DeviceCSR) or is_torch_sparse_csr(x)


def as_device_csr(x, device=None) -> DeviceCSR:
    """csr_matrix (uploaded), DeviceCSR (as is) or torch sparse_csr tensor (its buffers, int32 ids)."""
    if isinstance(x, DeviceCSR):
        return x
    if isinstance(x, csr_matrix):
        return DeviceCSR.from_scipy(x, device)
    if is_torch_sparse_csr(x):
        return DeviceCSR.from_torch(x, device)
    raise ValueError(f"not a sparse matrix type of this package: {type(x)}")


def fixed_width_prediction(like, pred_idx: torch.Tensor, k: int, n: int, m: int, values: Optional[torch.Tensor] = None):
    """k predicted column ids per row as a sparse matrix of the kind of `like` (DeviceCSR or torch
    sparse_csr), resident where `pred_idx` is: indptr = arange(n + 1) * k, data = 1 (or `values`)."""
    dev = pred_idx.device
    if values is None:
        values = torch.ones(n * k, dtype=like.data.dtype if isinstance(like, DeviceCSR) else like.values().dtype,
                            device=dev)
    if isinstance(like, DeviceCSR):
        indptr = torch.arange(n + 1, dtype=torch.int32, device=dev) * k
        return DeviceCSR(indptr, pred_idx, values, (n, m), k, k)
    idt = like.crow_indices().dtype
    crow = torch.arange(n + 1, dtype=idt, device=dev) * k
    return torch.sparse_csr_tensor(crow, pred_idx.to(idt), values, size=(n, m))

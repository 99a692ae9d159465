"""Matrix type aliases and default dtypes of the path.

PyTorch is always present here (it owns the device buffers), so the torch variants
of the reference's optional aliases are unconditional.
"""
from typing import Tuple, Union

import numpy as np
import torch
from scipy.sparse import csr_matrix

# accumulation happens in float64, inputs default to float32 / int32 indices
DefaultAccDataDType = np.float64
DefaultDataDType = np.float32
DefaultIndDType = np.int32
DefaultTorchDataDType = torch.float32
TORCH_AVAILABLE = True

_DENSE_TYPES = (np.ndarray, torch.Tensor)
_MATRIX_TYPES = _DENSE_TYPES + (csr_matrix,)

Number = Union[int, float, np.number]
DType = Union[np.dtype, torch.dtype]
DenseMatrix = Union[np.ndarray, torch.Tensor]
Matrix = Union[np.ndarray, csr_matrix, torch.Tensor]
CSRMatrixAsTuple = Tuple[np.ndarray, np.ndarray, np.ndarray]  # (data, indices, indptr)


def is_dense(x) -> bool:
    """ndarray or (strided) torch tensor."""
    if isinstance(x, torch.Tensor):
        return x.layout == torch.strided
    return isinstance(x, np.ndarray)


def is_sparse(x) -> bool:
    """csr_matrix, or a sparse matrix already resident in HBM: xcolumns_amd.DeviceCSR or a torch
    ``sparse_csr`` tensor (this build's extension of the reference's type seam, types.py:10-23)."""
    from ._device import is_device_sparse

    return isinstance(x, csr_matrix) or is_device_sparse(x)


def is_matrix(x) -> bool:
    """Any of the input kinds the path accepts."""
    return is_dense(x) or is_sparse(x)

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
    """ndarray or torch tensor."""
    return isinstance(x, _DENSE_TYPES)


def is_matrix(x) -> bool:
    """Any of the three input kinds the path accepts."""
    return isinstance(x, _MATRIX_TYPES)

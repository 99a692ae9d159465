"""Type aliases and default dtypes (mirrors /root/reference/xcolumns/types.py:7-27)."""
from typing import Tuple, Union

import numpy as np
import torch
from scipy.sparse import csr_matrix

DType = Union[np.dtype, torch.dtype]
Number = Union[int, float, np.number]
DenseMatrix = Union[np.ndarray, torch.Tensor]
Matrix = Union[np.ndarray, csr_matrix, torch.Tensor]
CSRMatrixAsTuple = Tuple[np.ndarray, np.ndarray, np.ndarray]
DefaultIndDType = np.int32
DefaultDataDType = np.float32
DefaultAccDataDType = np.float64
DefaultTorchDataDType = torch.float32
TORCH_AVAILABLE = True

_DENSE_TYPES = (np.ndarray, torch.Tensor)
_MATRIX_TYPES = (np.ndarray, csr_matrix, torch.Tensor)


def is_dense(x) -> bool:
    return isinstance(x, _DENSE_TYPES)


def is_matrix(x) -> bool:
    return isinstance(x, _MATRIX_TYPES)

"""CPU checks of the C-ABI boundary: the library builds, loads, and exports every
symbol include/xcolumns_amd.h declares (no compute calls without a GPU); the
product path fails loudly when the HIP library or the GPU is missing."""
import os
import re

import numpy as np
import pytest
from scipy.sparse import csr_matrix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "xcolumns_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from xcolumns_amd import _lib

    _lib.build()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes signature table out of sync with the header"
    assert lib.xc_abi_version() == 1


def test_argument_errors_need_no_gpu():
    """Argument validation happens before any launch: negative codes + text."""
    import ctypes

    from xcolumns_amd import _lib

    lib = _lib.load()
    rc = lib.xc_topk_csr(4, None, None, None, 0, 64, 3, None, None, 0, None, None, None, None, None)
    assert rc == _lib.XC_ERR_BAD_ARG and b"xc_topk_csr" in lib.xc_last_error()
    dummy = ctypes.c_void_p(1)
    rc = lib.xc_topk_csr(4, dummy, dummy, dummy, 0, 64, 0, None, None, 0, dummy, dummy, None, None, None)
    assert rc == _lib.XC_ERR_K_RANGE
    rc = lib.xc_topk_csr(4, dummy, dummy, dummy, 0, 5000, 3, None, None, 0, dummy, dummy, None, None, None)
    assert rc == _lib.XC_ERR_ROW_TOO_LONG
    with pytest.raises(ValueError):
        _lib.call("xc_topk_csr", 4, dummy, dummy, dummy, 7, 64, 3, None, None, 0, dummy, dummy, None, None, None)


def test_fails_loudly_without_library(monkeypatch, tmp_path):
    from xcolumns_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libxcolumns_amd.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc
    from xcolumns_amd.weighted_prediction import predict_top_k

    Y = csr_matrix(np.random.default_rng(0).random((8, 6)).astype(np.float32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        predict_top_k(Y, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        predict_optimizing_macro_f1_score_using_bc(Y, 2)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under xcolumns_amd/ may import, link
    or execute it."""
    pkg = os.path.join(ROOT, "xcolumns_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", "Makefile")):
                for ln in open(os.path.join(dirpath, f)).read().splitlines():
                    low = ln.lower()
                    if "oracle" in low:
                        assert not re.search(r"\b(import|from|include|cdll|-l)\b.*oracle", low), (f, ln)

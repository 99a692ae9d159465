"""Host-side helpers of the path (mirrors the parts of
/root/reference/xcolumns/utils.py the BCA / top-k path uses)."""
from __future__ import annotations

import inspect
import logging
import random
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
from scipy.sparse import csr_matrix

# same logger name and level as the reference, so existing logging configuration applies
logging.basicConfig()
logger = logging.getLogger("xcolumns")
logger.setLevel(logging.INFO)


def log(msg: str, verbose: bool = True, level: int = logging.INFO):
    """Emit `msg` on the package logger when `verbose` (the `verbose=` flag of the API)."""
    if verbose:
        logger.log(level, msg)


def _leveled(level: int):
    def emit(msg: str, verbose: bool = True):
        log(msg, verbose, level=level)
    return emit


log_debug = _leveled(logging.DEBUG)
log_info = _leveled(logging.INFO)
log_warning = _leveled(logging.WARNING)
log_error = _leveled(logging.ERROR)


def _filled_like(a, fill: float, shape, dtype):
    """A dense array of `fill` living where `a` lives: numpy for ndarray / csr_matrix
    inputs, a tensor on `a`'s device for torch inputs; shape / dtype default to a's."""
    shape = a.shape if shape is None else shape
    if isinstance(a, torch.Tensor):
        return torch.full(shape, fill, dtype=a.dtype if dtype is None else dtype, device=a.device)
    if isinstance(a, (np.ndarray, csr_matrix)):
        return np.full(shape, fill, dtype=a.dtype if dtype is None else dtype)
    raise ValueError(f"Unsupported type {type(a)}")


def zeros_like(a, shape: Tuple[int, ...] = None, dtype=None):
    return _filled_like(a, 0, shape, dtype)


def ones_like(a, shape: Tuple[int, ...] = None, dtype=None):
    return _filled_like(a, 1, shape, dtype)


def random_at_k_like(a, shape: Tuple[int, int], k: int, seed: Optional[int] = None):
    """Declared but empty in the reference (utils.py:97-100); kept so imports keep working."""
    return None


def random_at_k_np(shape: Tuple[int, int], k: int, dtype=None, seed: Optional[int] = None) -> np.ndarray:
    """k random labels per row, the numpy Generator stream of utils.py:103-116
    (``rng.choice(m, k, replace=False, shuffle=False)`` row by row)."""
    n, m = shape
    y_pred = np.zeros(shape, dtype=dtype)
    rng = np.random.default_rng(seed)
    labels_range = np.arange(m)
    for i in range(n):
        y_pred[i, rng.choice(labels_range, k, replace=False, shuffle=False)] = 1.0
    return y_pred


def random_at_k_csr(shape: Tuple[int, int], k: int, dtype=None, seed: Optional[int] = None) -> csr_matrix:
    """k random labels per row as CSR (utils.py:119-136).  The reference draws them
    inside numba (numba_csr_functions.py:92-112: a partial Fisher-Yates shuffle on
    ``random.randint``); numba keeps a private Mersenne-Twister whose stream is
    not reproducible outside it, so this uses the same algorithm on CPython's
    ``random`` -- same distribution, same stream as the un-JIT'd reference."""
    n, m = shape
    if seed is not None:
        random.seed(seed)
    indices = np.empty(n * k, dtype=np.int32)
    base = np.arange(m, dtype=np.int32)
    for i in range(n):
        index = base.copy()
        for t in range(k):
            j = random.randint(t, m - 1)
            index[t], index[j] = index[j], index[t]
        indices[i * k:(i + 1) * k] = index[:k]
    indptr = (np.arange(n + 1, dtype=np.int64) * k).astype(np.int32)
    return construct_csr_matrix(np.ones(n * k, dtype=np.float32).astype(dtype), indices, indptr,
                                dtype=dtype, shape=shape, sort_indices=True)


def unpack_csr_matrix(matrix: csr_matrix):
    return matrix.data, matrix.indices, matrix.indptr


def unpack_csr_matrices(*matrices) -> List[np.ndarray]:
    out = []
    for mat in matrices:
        out.extend(unpack_csr_matrix(mat))
    return out


def construct_csr_matrix(data, indices, indptr, dtype=None, shape=None, sort_indices=False) -> csr_matrix:
    mat = csr_matrix((data, indices, indptr), dtype=dtype, shape=shape)
    if sort_indices:
        mat.sort_indices()
    return mat


def uniform_search(low: float, high: float, step: float, func: Callable) -> Tuple[float, float]:
    """utils.py:174-184."""
    best, best_val = low, func(low)
    for i in np.arange(low + step, high, step):
        score = func(i)
        if score > best_val:
            best, best_val = i, score
    return best, best_val


def ternary_search(low: float, high: float, eps: float, func: Callable) -> Tuple[float, float]:
    """utils.py:187-201."""
    while high - low > eps:
        mid1 = low + (high - low) / 3
        mid2 = high - (high - low) / 3
        if func(mid1) < func(mid2):
            high = mid2
        else:
            low = mid1
    best = (low + high) / 2
    return best, func(best)


def add_kwargs_to_signature(func: Callable, func_with_kwargs: Callable, skip: Optional[List] = None) -> Callable:
    """Expose the keyword arguments of `func_with_kwargs` in `func.__signature__`
    (utils.py:209-230); the experiments filter kwargs through it."""
    skip = skip or []
    sig_with_kwargs = inspect.signature(func_with_kwargs)
    sig_new = inspect.signature(func)
    func.__signature__ = sig_new.replace(
        parameters=[p for p in sig_new.parameters.values() if p.kind != inspect.Parameter.VAR_KEYWORD]
        + [p for p in sig_with_kwargs.parameters.values()
           if p.default != inspect.Parameter.empty and p.name not in skip]
    )
    return func


# ---------------------------------------------------------------------------
# the visiting order's shuffle, faster than numpy but the SAME stream
# ---------------------------------------------------------------------------

class Pcg64Shuffler:
    """``rng.shuffle(order)`` of a ``np.random.default_rng`` generator through the library's host routine
    (``xc_host_shuffle_pcg64``: numpy's own Fisher-Yates walk and PCG64 stream on an int32 array, the swap
    partners prefetched -- csrc/xc_order.hip).  The generator object is kept in step: after every shuffle its
    state is what numpy's own shuffle would have left, so mixing both is safe.

    ``Pcg64Shuffler.usable()`` checks the routine against numpy on this machine once (permutation AND the
    generator's next draws); callers fall back to ``rng.shuffle`` when it says no."""

    _ok = None

    def __init__(self, rng: np.random.Generator, n: int):
        self.rng, self.n = rng, int(n)
        self.order = np.arange(self.n, dtype=np.int32)

    @staticmethod
    def _walk(rng: np.random.Generator, order: np.ndarray) -> None:
        import ctypes

        from . import _lib
        st = rng.bit_generator.state
        s, inc = int(st["state"]["state"]), int(st["state"]["inc"])
        mask = (1 << 64) - 1
        words = (ctypes.c_uint64 * 4)(s >> 64, s & mask, inc >> 64, inc & mask)
        has, uint = ctypes.c_int(int(st["has_uint32"])), ctypes.c_uint32(int(st["uinteger"]))
        _lib.call("xc_host_shuffle_pcg64", ctypes.cast(words, ctypes.c_void_p), ctypes.byref(has), ctypes.byref(uint),
                  int(order.size), ctypes.c_void_p(order.ctypes.data))
        st["state"]["state"] = (int(words[0]) << 64) | int(words[1])
        st["has_uint32"], st["uinteger"] = int(has.value), int(uint.value)
        rng.bit_generator.state = st

    @classmethod
    def usable(cls) -> bool:
        if cls._ok is None:
            try:
                ok = True
                for seed, n in ((13, 1), (7, 2), (123, 1000), (5, 70001)):
                    a, b = np.random.default_rng(seed), np.random.default_rng(seed)
                    if type(a.bit_generator).__name__ != "PCG64":
                        ok = False
                        break
                    ref, mine = np.arange(n), np.arange(n, dtype=np.int32)
                    for _ in range(3):               # cumulative, like the sweeps
                        a.shuffle(ref)
                        cls._walk(b, mine)
                    ok = ok and np.array_equal(ref, mine) and a.integers(0, 1 << 62, size=4).tolist() == b.integers(0, 1 << 62, size=4).tolist()
                    # the two-thread form the sweep loop uses by default (draws(), then apply()) against numpy as well
                    c = np.random.default_rng(seed)
                    ref2, two = np.arange(n), cls(np.random.default_rng(seed), n)
                    for _ in range(3):
                        c.shuffle(ref2)
                        two.apply(two.draws())
                    ok = ok and np.array_equal(ref2, two.order) and (c.integers(0, 1 << 62, size=4).tolist()
                                                                     == two.rng.integers(0, 1 << 62, size=4).tolist())
                cls._ok = bool(ok)
            except Exception:
                cls._ok = False
        return cls._ok

    def shuffle(self) -> np.ndarray:
        """Shuffle the int32 order in place (cumulatively) and return it (the caller copies what it keeps)."""
        self._walk(self.rng, self.order)
        return self.order

    # the two halves of the walk, for two threads (block_coordinate._OrderSource): the draws of sweep j + 1 do not
    # depend on the array, so they are generated while the swaps of sweep j are applied
    def draws(self) -> np.ndarray:
        """The swap partners of the NEXT shuffle (uint32[n - 1]); advances the generator like the shuffle itself."""
        import ctypes

        from . import _lib
        # a small ring of buffers: fresh pages cost more to touch than the draws to make (1.7 of 4 ms at 1 M rows);
        # the consumer holds at most two queued + one being applied while the next is filled
        if getattr(self, "_js_pool", None) is None:
            self._js_pool = [np.zeros(max(0, self.n - 1), dtype=np.uint32) for _ in range(5)]
            self._js_next = 0
        js = self._js_pool[self._js_next]
        self._js_next = (self._js_next + 1) % len(self._js_pool)
        st = self.rng.bit_generator.state
        s_, inc = int(st["state"]["state"]), int(st["state"]["inc"])
        mask = (1 << 64) - 1
        words = (ctypes.c_uint64 * 4)(s_ >> 64, s_ & mask, inc >> 64, inc & mask)
        has, uint = ctypes.c_int(int(st["has_uint32"])), ctypes.c_uint32(int(st["uinteger"]))
        _lib.call("xc_host_shuffle_draws", ctypes.cast(words, ctypes.c_void_p), ctypes.byref(has), ctypes.byref(uint),
                  self.n, ctypes.c_void_p(js.ctypes.data))
        st["state"]["state"] = (int(words[0]) << 64) | int(words[1])
        st["has_uint32"], st["uinteger"] = int(has.value), int(uint.value)
        self.rng.bit_generator.state = st
        return js

    def apply(self, js: np.ndarray) -> np.ndarray:
        """Apply the swaps of one shuffle to the order (in place) and return it."""
        import ctypes

        from . import _lib
        _lib.call("xc_host_shuffle_apply", self.n, ctypes.c_void_p(js.ctypes.data), ctypes.c_void_p(self.order.ctypes.data))
        return self.order

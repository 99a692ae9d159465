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
    indices = _random_at_k_vectorised(n, m, k) if n * k >= 4096 else None
    if indices is None:
        indices = _random_at_k_loop(n, m, k)
    indptr = (np.arange(n + 1, dtype=np.int64) * k).astype(np.int32)
    return construct_csr_matrix(np.ones(n * k, dtype=np.float32).astype(dtype), indices, indptr,
                                dtype=dtype, shape=shape, sort_indices=True)


def _random_at_k_loop(n: int, m: int, k: int) -> np.ndarray:
    """numba_csr_functions.py:92-112 on CPython's `random`, row by row."""
    indices = np.empty(n * k, dtype=np.int32)
    randint = random.randint
    for i in range(n):
        # the partial Fisher-Yates shuffle of 0 .. m-1 (swap position t with a random position j >= t, t = 0 .. k-1) on the few
        # positions it touches: a copy of the m-vector per row was 400 GB of memory traffic at 150 K x 670 K (5-9 s)
        moved = {}
        row = i * k
        for t in range(k):
            j = randint(t, m - 1)
            vt, vj = moved.get(t, t), moved.get(j, j)
            moved[j] = vt
            indices[row + t] = vj          # position t is final: later steps swap positions > t only
    return indices


def _random_at_k_vectorised(n: int, m: int, k: int) -> Optional[np.ndarray]:
    """The same draws without a Python call per draw: `random.randint(t, m - 1)` is t + _randbelow(m - t), and _randbelow(w)
    takes 32-bit Mersenne-Twister outputs >> (32 - w.bit_length()) until one is < w.  numpy's MT19937 is the same generator:
    loaded with `random`'s state it yields the same outputs in bulk; a candidate below m - k + 1 is taken at every step and
    one >= m at none, the few in between are settled in order; then the k swaps per row as array operations.  `random` is
    left where the loop would have left it.  None: not applicable (the caller takes the loop)."""
    if k < 1 or m - k + 1 < 1 or (m - k + 1).bit_length() != m.bit_length() or m.bit_length() > 32:
        return None
    try:
        version, internal, gauss = random.getstate()
        if version != 3 or len(internal) != 625:
            return None
        key, pos = np.asarray(internal[:624], dtype=np.uint32), int(internal[624])
        bg = np.random.MT19937()
        bg.state = {"bit_generator": "MT19937", "state": {"key": key, "pos": pos}}
        bits, need = m.bit_length(), n * k
        p_take = (m - k + 1) / float(1 << bits)
        total = int(need / p_take * 1.02) + 4096
        raw = bg.random_raw(total).astype(np.uint64)
        r = (raw >> np.uint64(32 - bits)).astype(np.int64)
        sure = r < (m - k + 1)
        maybe = np.nonzero((r >= m - k + 1) & (r < m))[0]
        taken = sure.copy()
        if maybe.size:
            before = np.cumsum(sure) - sure          # sure candidates in front of each position
            extra = 0
            for q in maybe:                          # a handful: m - t depends on the step, i.e. on how many were taken so far
                t = int(before[q] + extra) % k
                if r[q] < m - t:
                    taken[q] = True
                    extra += 1
        where = np.nonzero(taken)[0]
        if where.size < need:
            return None
        where = where[:need]
        consumed = int(where[-1]) + 1
        J = r[where].reshape(n, k) + np.arange(k, dtype=np.int64)[None, :]      # j_t = t + _randbelow(m - t)
        # `random` ends where the loop would have ended: the same state stepped `consumed` outputs
        bg.state = {"bit_generator": "MT19937", "state": {"key": key, "pos": pos}}
        bg.random_raw(consumed)
        st = bg.state["state"]
        random.setstate((3, tuple(int(x) for x in st["key"]) + (int(st["pos"]),), gauss))
        out = np.empty((n, k), dtype=np.int64)
        VT = np.empty((n, k), dtype=np.int64)        # the value step t moved to position j_t
        for t in range(k):
            j = J[:, t]
            vj, vt = j.copy(), np.full(n, t, dtype=np.int64)
            for s_ in range(t):                      # the latest earlier step that wrote the position decides
                hit = J[:, s_] == j
                vj[hit] = VT[hit, s_]
                hit = J[:, s_] == t
                vt[hit] = VT[hit, s_]
            VT[:, t] = vt
            out[:, t] = vj
        return out.reshape(-1).astype(np.int32)
    except Exception:
        return None


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


# ---------------------------------------------------------------------------
# ... and the same stream generated on the GPU
# ---------------------------------------------------------------------------

class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_PCG64_MULT = (2549297995355413924 << 64) | 4865540595714422341
_MASK128 = (1 << 128) - 1


def pcg64_advance(state: int, inc: int, steps: int) -> int:
    """The PCG64 state `steps` steps further (steps < 0: back), by the LCG's jump-ahead."""
    steps %= 1 << 128
    acc_mult, acc_plus, cur_mult, cur_plus = 1, 0, _PCG64_MULT, inc
    while steps:
        if steps & 1:
            acc_mult = (acc_mult * cur_mult) & _MASK128
            acc_plus = (acc_plus * cur_mult + cur_plus) & _MASK128
        cur_plus = ((cur_mult + 1) * cur_plus) & _MASK128
        cur_mult = (cur_mult * cur_mult) & _MASK128
        steps >>= 1
    return (acc_mult * state + acc_plus) & _MASK128


class DeviceNumpyOrders:
    """``rng.shuffle(order)`` of a ``np.random.default_rng`` generator, cumulatively, ON the GPU
    (csrc/xc_order_dev.hip): :meth:`next` returns the int32 order tensor of the next sweep -- the permutation numpy
    would produce -- with no host work per sweep.  The orders are generated `ahead` shuffles in advance on a side
    stream (the masked rejection is one wavefront's work: it runs beside the sweeps); the consumer's stream waits for
    the order it is handed.  :meth:`finish` raises if a shuffle failed (candidate buffer exhausted: sized for 8
    standard deviations) and, with ``sync_rng`` (and ``ahead=0``), leaves the Python generator where numpy's own
    shuffles would have."""

    _ok = None
    DEPTH = 4     # order buffers in rotation

    def __init__(self, rng: np.random.Generator, n: int, device, ahead: int = 2, limit=None):
        import ctypes

        import torch

        from . import _device as D
        from . import _lib
        self.rng, self.n, self.dev = rng, int(n), device
        self.limit = (1 << 62) if limit is None else int(limit)   # orders the consumer can ask for: none beyond are made
        self.ahead = int(max(0, min(ahead, self.DEPTH - 2)))
        st = rng.bit_generator.state
        if st["bit_generator"] != "PCG64":
            raise ValueError("DeviceNumpyOrders needs a PCG64 generator (np.random.default_rng)")
        self._state0, self._inc = int(st["state"]["state"]), int(st["state"]["inc"])
        has = int(st["has_uint32"])
        base = pcg64_advance(self._state0, self._inc, -1) if has else self._state0
        self._base, self._has0 = base, has
        nbytes = ctypes.c_int64(0)
        _lib.call("xc_order_dev_workspace_bytes", self.n, ctypes.byref(nbytes))
        self.ws = torch.empty(int(nbytes.value), dtype=torch.uint8, device=device)
        self.bufs = [torch.empty(max(1, self.n), dtype=torch.int32, device=device) for _ in range(self.DEPTH)]
        # The two side streams are made ONCE per device, with high priority: torch hands out pooled streams round robin and
        # ROCm maps them onto a few hardware queues -- every other call got a pair that shared a queue with the sweeps'
        # stream and lost the overlap (10 sweeps: 10 / 14.5 / 10 / 14.5 ms, call after call); priority streams have queues of
        # their own.
        self.side, self.side_b = self._side_streams(device) if self.ahead > 0 else (None, None)
        self._applied = {}          # shuffle number -> event of its apply half
        if self.side is not None:
            # the side streams work on these tensors: the allocator must not hand their memory to anybody else before
            # that work has finished, even if this object is dropped without finish()
            for st in (self.side, self.side_b):
                self.ws.record_stream(st)
                for b in self.bufs:
                    b.record_stream(st)
        mask = (1 << 64) - 1
        words = (ctypes.c_uint64 * 4)(base >> 64, base & mask, self._inc >> 64, self._inc & mask)
        with torch.cuda.stream(self.side) if self.side is not None else _nullcontext():
            _lib.call("xc_order_dev_begin", D.ptr(self.ws), ctypes.cast(words, ctypes.c_void_p), has, self.n,
                      D.ptr(self.bufs[0]), D.stream())
        self.generated = 0          # shuffles enqueued
        self.handed = 0             # orders handed to the consumer
        self._done = {}             # shuffle number -> event on the side stream
        self._free = {}             # order number -> event on the consumer's stream recorded when the NEXT order was handed out
        for _ in range(min(self.ahead, self.limit)):
            self._generate()

    _streams = {}

    @classmethod
    def _side_streams(cls, device):
        import torch
        key = torch.device(device).index or 0
        if key not in cls._streams:
            cls._streams[key] = (torch.cuda.Stream(device=device, priority=-1), torch.cuda.Stream(device=device, priority=-1))
        return cls._streams[key]

    def __del__(self):
        try:
            if self.side is not None:
                self.side.synchronize()
                self.side_b.synchronize()
        except Exception:
            pass

    def _generate(self):
        """Enqueue shuffle number generated + 1: bufs[k % DEPTH] <- shuffle of bufs[(k - 1) % DEPTH].  On side streams in two
        halves: the draw (rejection walk, the chain the NEXT shuffle waits for) on one, the apply (Fisher-Yates swaps) on the
        other -- the apply of shuffle k runs beside the draw of shuffle k + 1."""
        import torch

        from . import _device as D
        from . import _lib
        k = self.generated + 1
        src, dst = self.bufs[(k - 1) % self.DEPTH], self.bufs[k % self.DEPTH]
        if self.side is None or self.n < 2:
            _lib.call("xc_order_dev_shuffle", D.ptr(self.ws), self.n, D.ptr(src), D.ptr(dst), D.stream())
            self.generated = k
            return
        slot = k & 1
        with torch.cuda.stream(self.side):
            ev = self._applied.pop(k - 2, None)          # the partners of `slot` were last read by the apply of shuffle k - 2
            if ev is not None:
                self.side.wait_event(ev)
            _lib.call("xc_order_dev_draw", D.ptr(self.ws), self.n, slot, D.stream())
            drawn = torch.cuda.Event()
            drawn.record(self.side)
        with torch.cuda.stream(self.side_b):
            self.side_b.wait_event(drawn)
            # the buffer held order k - DEPTH: its sweep was enqueued before order k - DEPTH + 1 was handed out
            ev = self._free.pop(k - self.DEPTH + 1, None)
            if ev is not None:
                self.side_b.wait_event(ev)
            _lib.call("xc_order_dev_apply", D.ptr(self.ws), self.n, slot, D.ptr(src), D.ptr(dst), D.stream())
            ev = torch.cuda.Event()
            ev.record(self.side_b)
            self._done[k] = ev
            self._applied[k] = ev
        self.generated = k

    def next(self):
        import torch
        k = self.handed + 1
        cur = torch.cuda.current_stream()
        if self.side is not None:
            # whatever the consumer enqueued so far (the sweeps of the orders handed out before) precedes this point
            ev = torch.cuda.Event()
            ev.record(cur)
            self._free[k] = ev
        while self.generated < max(k, min(k + self.ahead, self.limit)):
            self._generate()
        if self.side is not None:
            cur.wait_event(self._done.pop(k))
        self.handed = k
        return self.bufs[k % self.DEPTH][:self.n]

    def status(self):
        import ctypes

        import torch

        from . import _device as D
        from . import _lib
        out = (ctypes.c_int64 * 8)()
        if self.side is not None:
            self.side_b.synchronize()      # (the apply half may flag a failed shuffle too)
        with torch.cuda.stream(self.side) if self.side is not None else _nullcontext():
            _lib.call("xc_order_dev_status", D.ptr(self.ws), out, D.stream())
        self.last_walk = {"cycles": int(out[3]), "us": int(out[4]) / 100.0, "rounds": int(out[5]), "batches": int(out[6]),
                          "fallbacks": int(out[7])}
        return int(out[0]), int(out[1]), int(out[2])

    def finish(self, sync_rng: bool = False) -> None:
        flag, consumed, _ = self.status()
        if flag != 0:
            raise RuntimeError(f"the device visiting-order generator failed (flag {flag}); set XCOLUMNS_ORDER_DEVICE=0")
        if not sync_rng:
            return
        if self.generated != self.handed:
            raise ValueError("sync_rng needs ahead=0 (orders generated in advance moved the generator further)")
        st = self.rng.bit_generator.state
        # draw d is the low / high half of output d // 2 (counted from the base state): after `consumed` draws the
        # generator has made ceil(consumed / 2) outputs and, if consumed is odd, still holds the high half of the last
        outputs = (consumed + 1) // 2
        state = pcg64_advance(self._base, self._inc, outputs)
        st["state"]["state"] = state
        st["has_uint32"] = consumed & 1
        if consumed & 1:
            hi, lo = state >> 64, state & ((1 << 64) - 1)
            x, rot = hi ^ lo, hi >> 58
            out = ((x >> rot) | (x << ((64 - rot) & 63))) & ((1 << 64) - 1)
            st["uinteger"] = out >> 32
        self.rng.bit_generator.state = st

    @classmethod
    def usable(cls, device) -> bool:
        """Checked against numpy once per process: permutations of three cumulative shuffles at sizes around the
        mask boundaries, and the generator's next draws afterwards."""
        if cls._ok is None:
            try:
                ok = True
                for seed, n in ((13, 2), (7, 3), (3, 64), (5, 65), (123, 1000), (9, 4097), (11, 70001)):
                    a, b = np.random.default_rng(seed), np.random.default_rng(seed)
                    if seed == 9:
                        a.integers(0, 10, size=3, dtype=np.uint32)      # leave a buffered 32-bit half behind
                        b.integers(0, 10, size=3, dtype=np.uint32)
                    ref = np.arange(n)
                    gen = cls(b, n, device, ahead=0 if seed != 11 else 2)
                    for _ in range(3):
                        a.shuffle(ref)
                        mine = gen.next().cpu().numpy()
                        ok = ok and np.array_equal(ref, mine)
                    if seed == 11:
                        gen.finish()
                        continue
                    gen.finish(sync_rng=True)
                    ok = ok and a.integers(0, 1 << 62, size=4).tolist() == b.integers(0, 1 << 62, size=4).tolist()
                cls._ok = bool(ok)
            except Exception:
                cls._ok = False
        return cls._ok

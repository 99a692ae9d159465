"""The visiting order generated on the GPU (csrc/xc_order_dev.hip) is numpy's: the reference shuffles ONE array
cumulatively with np.random.default_rng(seed), once per sweep (/root/reference/xcolumns/block_coordinate.py:413-419)."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [2, 3, 5, 64, 65, 127, 129, 1000, 4096, 4097, 65536, 65537, 99991, 1_000_000, 1_048_577])
def test_device_orders_are_numpys(n):
    from xcolumns_amd import _device as D
    from xcolumns_amd.utils import DeviceNumpyOrders
    dev = D.require_gpu()
    for seed in (13, 2024):
        a, b = np.random.default_rng(seed), np.random.default_rng(seed)
        ref = np.arange(n)
        gen = DeviceNumpyOrders(b, n, dev, ahead=0)
        for s in range(4):
            a.shuffle(ref)
            mine = gen.next().cpu().numpy()
            assert np.array_equal(ref, mine), (n, seed, s, int((ref != mine).sum()))
        gen.finish(sync_rng=True)
        # the grid-wide rounds settled every walk themselves (no walk was left to the one-wavefront fallback)
        assert gen.last_walk["fallbacks"] == 0 and 1 <= gen.last_walk["rounds"] <= 40, gen.last_walk
        # the Python generator is where numpy's own shuffles would have left it
        assert a.integers(0, 1 << 62, size=5).tolist() == b.integers(0, 1 << 62, size=5).tolist()


@pytest.mark.parametrize("rounds", [1, 3])
def test_device_orders_fall_back_to_the_one_workgroup_walk(rounds):
    """Given too few rounds the grid-wide walk gives up: the one-wavefront walk behind it redoes the shuffle -- the
    same permutations and generator position (the safety net of csrc/xc_order_dev.hip, never taken by default)."""
    from xcolumns_amd import _device as D
    from xcolumns_amd import _lib
    from xcolumns_amd.utils import DeviceNumpyOrders
    dev = D.require_gpu()
    _lib.call("xc_order_dev_set_rounds", rounds)
    try:
        for n in (70_001, 300_000):
            a, b = np.random.default_rng(11), np.random.default_rng(11)
            ref = np.arange(n)
            gen = DeviceNumpyOrders(b, n, dev, ahead=0)
            for s in range(3):
                a.shuffle(ref)
                assert np.array_equal(ref, gen.next().cpu().numpy()), (n, s)
            gen.finish(sync_rng=True)
            assert gen.last_walk["fallbacks"] == 3, gen.last_walk
            assert a.integers(0, 1 << 62, size=5).tolist() == b.integers(0, 1 << 62, size=5).tolist()
    finally:
        _lib.call("xc_order_dev_set_rounds", 0)


def test_device_orders_with_a_buffered_half_and_many_sweeps():
    from xcolumns_amd import _device as D
    from xcolumns_amd.utils import DeviceNumpyOrders
    dev = D.require_gpu()
    n = 200_003
    a, b = np.random.default_rng(5), np.random.default_rng(5)
    for g in (a, b):
        g.integers(0, 1000, size=7, dtype=np.uint32)     # an odd number of 32-bit draws: a buffered half is left
    assert a.bit_generator.state["has_uint32"] == 1
    ref = np.arange(n)
    gen = DeviceNumpyOrders(b, n, dev, ahead=0)
    for s in range(25):
        a.shuffle(ref)
        mine = gen.next()
        if s % 6 == 0 or s == 24:
            assert np.array_equal(ref, mine.cpu().numpy()), s
    gen.finish(sync_rng=True)
    assert a.random(3).tolist() == b.random(3).tolist()
    # generated ahead on a side stream (the product's form): the same orders, a consumer that keeps reading them
    c, r = np.random.default_rng(5), np.random.default_rng(5)
    for g in (c, r):
        g.integers(0, 1000, size=7, dtype=np.uint32)
    ref2, gen2 = np.arange(n), DeviceNumpyOrders(r, n, dev, ahead=2)
    sums = []
    for s in range(12):
        c.shuffle(ref2)
        o = gen2.next()
        sums.append((o.to(torch.int64) * torch.arange(n, device=o.device)).sum())     # a consumer on the current stream
        if s in (0, 5, 11):
            assert np.array_equal(ref2, o.cpu().numpy()), s
    gen2.finish()
    # a generator dropped without finish() while its side stream still works must not hand its buffers to anybody else
    for _ in range(3):
        g3 = DeviceNumpyOrders(np.random.default_rng(1), n, dev, ahead=2)
        g3.next()
        del g3
        torch.empty(n, dtype=torch.int32, device=dev).zero_()
    torch.cuda.synchronize()


def test_device_orders_time_and_api_use(oref):
    """1 M rows: the device generator's time per order; and the public API uses it (same utilities as with the host
    walk, XCOLUMNS_ORDER_DEVICE=0)."""
    import os

    from xcolumns_amd import _device as D
    from xcolumns_amd.block_coordinate import predict_optimizing_macro_f1_score_using_bc as f
    from xcolumns_amd.synthetic import make_csr
    from xcolumns_amd.utils import DeviceNumpyOrders
    dev = D.require_gpu()
    assert DeviceNumpyOrders.usable(dev)
    gen = DeviceNumpyOrders(np.random.default_rng(1), 1_000_000, dev, ahead=0)
    gen.next()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        gen.next()
    torch.cuda.synchronize()
    print("device order generator: %.3f ms per 1M-row order" % ((time.perf_counter() - t0) * 100))
    gen.finish()
    print("  the rejection walk (grid-wide rounds): %.1f us, %d rounds" % (gen.last_walk["us"], gen.last_walk["rounds"]), gen.last_walk)
    assert gen.last_walk["fallbacks"] == 0
    # beside a kernel that keeps the GPU busy (the product's situation: the sweeps run meanwhile)
    gen = DeviceNumpyOrders(np.random.default_rng(1), 1_000_000, dev, ahead=2)
    a = torch.rand(8192, 8192, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        o = gen.next()
        for _ in range(4):
            a = torch.sin(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 100
    gen.finish()
    print("  beside elementwise kernels on the main stream: %.3f ms per order + 4 kernels; walk %.1f us, %d rounds" % (
        dt, gen.last_walk["us"], gen.last_walk["rounds"]))
    Y = make_csr(120_000, 40_000, 30, seed=8)
    _, m1 = f(Y, 5, seed=3, max_iters=3, tolerance=-1.0, return_meta=True, bca_waves=1)
    os.environ["XCOLUMNS_ORDER_DEVICE"] = "0"
    try:
        _, m2 = f(Y, 5, seed=3, max_iters=3, tolerance=-1.0, return_meta=True, bca_waves=1)
    finally:
        os.environ.pop("XCOLUMNS_ORDER_DEVICE")
    assert np.allclose(m1["utilities"], m2["utilities"], rtol=0, atol=1e-13)

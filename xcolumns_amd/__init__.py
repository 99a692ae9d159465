"""xcolumns_amd -- MI355X-native block-coordinate-ascent prediction for xCOLUMNs.

Drop-in for the hot path of mwydmuch/xCOLUMNs (``xcolumns.block_coordinate``,
``xcolumns.weighted_prediction``, ``xcolumns.confusion_matrix``) and its neighbours
(``xcolumns.frank_wolfe``, ``xcolumns.metrics``; ``xcolumns_amd.io`` for the drivers'
on-disk formats, ``xcolumns_amd.distributed`` for rows sharded over GPUs): same function
names, arguments, return types and error behaviour; the compute runs in
hand-written HIP kernels for gfx950 behind the C ABI of ``include/xcolumns_amd.h``.
There is no CPU fallback: without the built library and a GPU the calls raise.
"""
__version__ = "0.1.0"


def __getattr__(name):   # lazily: importing the package must not import torch
    if name == "DeviceCSR":
        from ._device import DeviceCSR
        return DeviceCSR
    raise AttributeError(f"module 'xcolumns_amd' has no attribute {name!r}")

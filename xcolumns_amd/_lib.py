"""ctypes binding of ``libxcolumns_amd.so`` (the C ABI in ``include/xcolumns_amd.h``).

The library holds the hand-written HIP kernels for gfx950.  There is NO CPU
fallback: if the shared object is missing or a call fails, a ``RuntimeError``
is raised.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C xcolumns_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libxcolumns_amd.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

XC_F32, XC_F64 = 0, 1
XC_MAX_K = 64
XC_MAX_ROW_NNZ = 1024
XC_UTILITY_PARTIALS = 1024
XC_CTRL_RING, XC_CTRL_RING_SLOTS, XC_CTRL_RING_STRIDE = 16, 8, 8
XC_CTRL_SIZE = XC_CTRL_RING + XC_CTRL_RING_STRIDE * XC_CTRL_RING_SLOTS
XC_ERR_BAD_ARG, XC_ERR_K_RANGE, XC_ERR_ROW_TOO_LONG, XC_ERR_NO_DEVICE = -1, -2, -3, -4

# metric ids (include/xcolumns_amd.h)
(XC_M_PRECISION_AT_K, XC_M_PRECISION, XC_M_RECALL, XC_M_FBETA, XC_M_JACCARD,
 XC_M_BALANCED_ACC, XC_M_GMEAN, XC_M_HMEAN, XC_M_ACCURACY, XC_M_RECALL_PRECISION_MIX) = range(10)


class XcMetric(ctypes.Structure):
    """``struct xc_metric``."""

    _fields_ = [
        ("base", c_int32),
        ("mixed", c_int32),
        ("epsilon", c_double),
        ("beta", c_double),
        ("kf", c_double),
        ("alpha", c_double),
        ("mf", c_double),
    ]


# name -> (restype, argtypes); every symbol include/xcolumns_amd.h declares
SIGNATURES = {
    "xc_abi_version": (c_int, []),
    "xc_last_error": (c_char_p, []),
    "xc_device_info": (c_int, [POINTER(c_int), POINTER(c_int), c_char_p, c_int]),
    "xc_topk_csr": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                            c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xc_threshold_count_csr": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_double,
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "xc_threshold_fill_csr": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_double,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xc_topk_dense": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_int, c_int, c_double, c_int,
                              c_void_p, c_int, c_void_p]),
    "xc_confusion_csr": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xc_csr_rows_ascending": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xc_confusion_csr_pred_side": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xc_confusion_csr_match": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "xc_confusion_dense": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                   c_void_p, c_void_p]),
    "xc_bca_gather_pred_eta": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "xc_bca_colsum_csr": (c_int, [c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "xc_bca_expand_colsum": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xc_bca_pipeline_begin": (c_int, [c_void_p, c_double, c_double, c_double, c_int, c_double, c_int, c_int, c_int,
                                      c_int, c_int, c_int, c_void_p]),
    "xc_bca_plan_sweep_pipelined": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "xc_bca_plan_boundary_pipelined": (c_int, [c_void_p, c_int64, c_double, c_int, c_void_p, c_int, c_void_p,
                                               c_double, c_void_p]),
    "xc_bca_ring_wait": (c_int, [c_void_p, c_int, c_double, c_double, c_void_p]),
    "xc_event_synchronize": (c_int, [c_void_p]),
    "xc_host_alloc_pinned": (c_int, [POINTER(c_void_p), c_int64]),
    "xc_host_free_pinned": (c_int, [c_void_p]),
    "xc_coverage_sweep_csr": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                      c_void_p, c_int, c_void_p, c_double, c_int, c_int, c_void_p, c_void_p]),
    "xc_coverage_product": (c_int, [c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "xc_fw_gradient": (c_int, [c_int64, c_void_p, POINTER(XcMetric), c_double, c_int, c_void_p, c_void_p, c_void_p]),
    "xc_fw_alpha_chunks": (c_int, [c_int64]),
    "xc_fw_alpha_curve": (c_int, [c_int64, c_void_p, c_void_p, POINTER(XcMetric), c_int, c_void_p, c_void_p, c_void_p]),
    "xc_confusion_counts_csr": (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xc_topk_csr_ab": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p]),
    "xc_topk_csr_rowwise": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                    c_int64, c_void_p, c_void_p, c_void_p]),
    "xc_threshold_count_csr_rowwise": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p,
                                               c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "xc_threshold_fill_csr_rowwise": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_double, c_void_p,
                                              c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xc_bca_pack_rows": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xc_bca_pack_rows_from_colsum": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p]),
    "xc_bca_accumulate_pred": (c_int, [c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "xc_bca_commit_utility": (c_int, [c_int64, c_int64, c_double, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                      POINTER(XcMetric), c_int, c_void_p, c_void_p]),
    "xc_utility_finish_host": (c_int, [c_void_p, POINTER(c_double), POINTER(c_double), c_void_p]),
    "xc_bca_sweep_csr": (c_int, [c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                 c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(XcMetric),
                                 c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "xc_bca_plan_create": (c_int, [POINTER(c_void_p), c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                   c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, POINTER(XcMetric), POINTER(XcMetric), c_int,
                                   c_int]),
    "xc_bca_plan_destroy": (c_int, [c_void_p]),
    "xc_bca_plan_sweep": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                  c_void_p]),
    "xc_bca_plan_boundary": (c_int, [c_void_p, c_int64, c_double, c_int, c_int, POINTER(c_double), POINTER(c_double),
                                     c_void_p]),
    "xc_bca_set_validation": (c_int, [c_int]),
    "xc_bca_set_acc_delta": (c_int, [c_int]),
    "xc_bca_plan_delta": (c_int, [c_void_p]),
    "xc_bca_delta_pack": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xc_bca_delta_unpack": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xc_bca_set_tuning": (c_int, [c_double, c_double]),
    "xc_dense_pred_to_fixed": (c_int, [c_int64, c_int64, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                       c_void_p]),
    "xc_scatter_sum_workspace_bytes": (c_int, [c_int64, c_int64, POINTER(c_int64)]),
    "xc_scatter_sum_f32": (c_int, [c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "xc_label_busy_list": (c_int, [c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xc_bca_exchange_step": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "xc_host_shuffle_pcg64": (c_int, [c_void_p, POINTER(c_int), POINTER(ctypes.c_uint32), c_int64, c_void_p]),
    "xc_host_shuffle_draws": (c_int, [c_void_p, POINTER(c_int), POINTER(ctypes.c_uint32), c_int64, c_void_p]),
    "xc_host_shuffle_apply": (c_int, [c_int64, c_void_p, c_void_p]),
    "xc_event_create": (c_int, [POINTER(c_void_p)]),
    "xc_event_destroy": (c_int, [c_void_p]),
    "xc_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(ctypes.c_float)]),
    "xc_bca_time_next_sweep": (c_int, [c_void_p, c_void_p]),
    "xc_bca_time_span": (c_int, [c_int]),
    "xc_bca_state_unpack": (c_int, [c_int64, c_void_p, c_void_p, c_double, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p]),
    "xc_bca_sweep_dense": (c_int, [c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int,
                                   c_int, c_void_p, c_void_p, POINTER(XcMetric), c_int, c_int, c_int,
                                   c_void_p]),
    "xc_bca_sweep_dense_concurrent": (c_int, [c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int, c_int,
                                              c_void_p, POINTER(XcMetric), c_int, c_int, c_int, c_void_p, c_void_p]),
    "xc_utility_vectors": (c_int, [c_int64, c_int64, c_void_p, POINTER(XcMetric), c_void_p, c_void_p]),
    "xc_order_dev_candidates": (c_int, [c_int64, POINTER(c_int64)]),
    "xc_order_dev_workspace_bytes": (c_int, [c_int64, POINTER(c_int64)]),
    "xc_order_dev_begin": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "xc_order_dev_shuffle": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "xc_order_dev_draw": (c_int, [c_void_p, c_int64, c_int, c_void_p]),
    "xc_order_dev_apply": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "xc_order_dev_status": (c_int, [c_void_p, POINTER(c_int64), c_void_p]),
    "xc_order_dev_set_rounds": (c_int, [c_int]),
    "xc_order_dev_walk_trace": (c_int, [c_void_p, c_int64, POINTER(c_int64), POINTER(c_int64), c_void_p]),
    "xc_bca_ord_window": (c_int, [POINTER(c_int), POINTER(c_int)]),
    "xc_bca_ord_workspace_bytes": (c_int, [c_int64, c_int64, c_int, c_int, POINTER(c_int64)]),
    "xc_bca_ord_sweep": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                 c_void_p, c_int, c_int, c_int, POINTER(XcMetric), c_int, c_int, ctypes.c_uint32, c_void_p,
                                 POINTER(c_int64), c_void_p]),
}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 with hipcc (``csrc/Makefile``)."""
    proc = subprocess.run(["make", "-C", CSRC_DIR, "-j4"], capture_output=True, text=True)
    if verbose or proc.returncode != 0:
        print(proc.stdout)
        print(proc.stderr)
    if proc.returncode != 0:
        raise RuntimeError("building libxcolumns_amd.so failed (hipcc); see output above")
    return LIB_PATH


def load() -> ctypes.CDLL:
    """Load the library; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the MI355X kernels have not been built. "
            "Run `make -C xcolumns_amd/csrc` (needs hipcc). There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError -> a declared symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    """Call an ``int``-returning entry point and raise on a non-zero status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.xc_last_error()
        msg = msg.decode() if msg else ""
        if rc in (XC_ERR_K_RANGE, XC_ERR_ROW_TOO_LONG, XC_ERR_BAD_ARG):
            raise ValueError(f"{name}: {msg} (code {rc})")
        raise RuntimeError(f"{name} failed: {msg} (code {rc})")


def device_info():
    lib = load()
    cu, wv = c_int(0), c_int(0)
    buf = ctypes.create_string_buffer(64)
    rc = lib.xc_device_info(ctypes.byref(cu), ctypes.byref(wv), buf, 64)
    if rc != 0:
        raise RuntimeError("no MI355X visible: " + (lib.xc_last_error() or b"").decode())
    return {"cu_count": cu.value, "waves_per_cu": wv.value, "arch": buf.value.decode()}

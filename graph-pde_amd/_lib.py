"""ctypes binding of libgpde.so (the C ABI declared in include/gpde.h).

The library is loaded lazily and never stored on a module/`nn.Module` instance, so models stay
picklable (`torch.save(model)`, /root/reference/graph-neural-operator/UAI1_full_resolution.py:317).
There is NO fallback: if the shared object is missing or a symbol is absent this raises.
"""
from __future__ import annotations

import ctypes
import os
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPDE_LIB", os.path.join(_PKG, "libgpde.so"))   # GPDE_LIB: experiments

GPDE_OK = 0
GPDE_VERSION, GPDE_VERSION_ABLATION, GPDE_VERSION_INSTRUMENTED = 100, 0x10000, 0x20000
GPDE_AGGR_ADD, GPDE_AGGR_MEAN, GPDE_AGGR_MAX = 0, 1, 2
GPDE_WECONV_MAX_GROUP = 16
GPDE_FWD_DEFAULT, GPDE_FWD_F16SPLIT = 0, 1
# the other forward flags of include/gpde.h (A/B switches; tests/test_abi.py checks these values against the header)
GPDE_FWD_F16SPLIT_8WAVE, GPDE_FWD_STATIC_RANGES, GPDE_FWD_AGG_F16, GPDE_FWD_AGG_F32, GPDE_FWD_NO_EDGE_PATH = 2, 4, 16, 32, 64
GPDE_WIDTH = 64

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)


class GpdeWeConvDesc(ctypes.Structure):
    """include/gpde.h `GpdeWeConvDesc`: one NNConv call given its per-edge weights (tests/test_abi.py checks the layout)."""
    _fields_ = [("x", ctypes.c_void_p), ("edge_weights", ctypes.c_void_p), ("rowptr", ctypes.c_void_p),
                ("src", ctypes.c_void_p), ("root", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("residual", ctypes.c_void_p), ("out", ctypes.c_void_p), ("n_nodes", ctypes.c_int32),
                ("aggr", ctypes.c_int32), ("relu", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class GpdeNodeAttr(ctypes.Structure):
    """include/gpde.h `GpdeNodeAttr`: edge attributes described by a node table (tests/test_abi.py checks the layout)."""
    _fields_ = [("table", ctypes.c_void_p), ("stride", ctypes.c_int32), ("n_slots", ctypes.c_int32), ("sel", ctypes.c_int32 * 8)]


# name -> (restype, argtypes); mirrors include/gpde.h one to one (tests check the two agree)
SIGNATURES = {
    "gpde_version": (ctypes.c_int, []),
    "gpde_last_error": (ctypes.c_char_p, []),
    "gpde_reload_switches": (ctypes.c_int, []),
    "gpde_csr_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64]),
    "gpde_csr_from_coo": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.c_void_p]),
    "gpde_gather_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "gpde_mlp_pack_bytes": (ctypes.c_size_t, [ctypes.c_int, c_i32p]),
    "gpde_mlp_pack": (ctypes.c_int, [ctypes.c_int, c_i32p, ctypes.POINTER(ctypes.c_void_p),
                                     ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p,
                                     ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_fwd_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64,
                                                          ctypes.c_int, c_i32p]),
    "gpde_nnconv_fwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_i32p,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_fwd_act": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                           ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_i32p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_fwd_hidden_act": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_int, c_i32p, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                                  ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                                  ctypes.c_void_p]),
    "gpde_nnconv_fwd_plan": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, c_i32p,
                                            ctypes.c_size_t, c_i32p, c_i64p, c_i32p, c_i32p]),
    "gpde_nnconv_fwd_kernel": (ctypes.c_char_p, [ctypes.c_int64, ctypes.c_int, c_i32p, ctypes.c_uint32]),
    "gpde_nnconv_bwd_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64,
                                                          ctypes.c_int, c_i32p]),
    "gpde_nnconv_bwd_workspace_bytes_one_chunk": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, c_i32p]),
    "gpde_csr_source_order": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_bwd_ordered": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                               ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int, c_i32p, ctypes.POINTER(ctypes.c_void_p),
                                               ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int,
                                               ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_bwd_hidden_ordered": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                                      ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_void_p, ctypes.c_int, c_i32p,
                                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_bwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_int, c_i32p, ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_fwd_nodeattr": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32,
                                                c_i32p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_int, c_i32p, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                                ctypes.c_void_p]),
    "gpde_hidden_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, c_i32p]),
    "gpde_hidden_fwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_void_p, ctypes.c_int, c_i32p, ctypes.c_void_p,
                                       ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_fwd_hidden": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_int, c_i32p, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                              ctypes.c_void_p]),
    "gpde_nnconv_fwd_mixed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                             c_i32p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_bwd_hidden": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                              ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_i32p,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_hidden_bwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                       c_i32p, ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p,
                                       ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_fwd_keepz": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_int, c_i32p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                             ctypes.c_void_p]),
    "gpde_nnconv_bwd_z": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_i32p, ctypes.POINTER(ctypes.c_void_p),
                                         ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                                         ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_bwd_attr": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_i32p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_bwd_deferred_supported": (ctypes.c_int, [ctypes.c_int, c_i32p]),
    "gpde_nnconv_bwd_deferred_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, c_i32p, ctypes.c_int]),
    "gpde_nnconv_bwd_light": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_i32p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_bwd_deferred": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_i32p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_fwd_mixed_keepz": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_i32p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_edge_weights_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, c_i32p]),
    "gpde_edge_weights_fwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, c_i32p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_bwd_edgeweights_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64]),
    "gpde_nnconv_bwd_edgeweights": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_edge_weights_bwd_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, c_i32p]),
    "gpde_edge_weights_bwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, c_i32p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_fwd_edgeweights_group": (ctypes.c_int, [ctypes.POINTER(GpdeWeConvDesc), ctypes.c_int, ctypes.c_void_p]),
    "gpde_nnconv_fwd_na": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(GpdeNodeAttr), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_i32p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_hidden_fwd_na": (ctypes.c_int, [ctypes.POINTER(GpdeNodeAttr), ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, c_i32p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "gpde_nnconv_bwd_na": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(GpdeNodeAttr), ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_i32p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_bwd_light_na": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(GpdeNodeAttr), ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_i32p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_nnconv_bwd_deferred_na": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.POINTER(GpdeNodeAttr), ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_i32p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_hidden_bwd_na": (ctypes.c_int, [ctypes.POINTER(GpdeNodeAttr), ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_i32p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_radius_graph_count": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                               ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    "gpde_radius_graph_fill": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                              ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_int64, ctypes.c_void_p]),
    "gpde_radius_graph2_count": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                                ctypes.c_double, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]),
    "gpde_radius_graph2_fill": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                               ctypes.c_double, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int64, ctypes.c_void_p]),
    "gpde_radius_csr_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                                          ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "gpde_radius_csr_count": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                             ctypes.c_double, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(ctypes.c_double), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                             ctypes.c_void_p]),
    "gpde_radius_csr_fill": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                            ctypes.c_double, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(ctypes.c_double), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "gpde_profile_begin": (ctypes.c_int, []),
    "gpde_profile_end": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), c_i32p,
                                        ctypes.POINTER(ctypes.c_double)]),
    "gpde_profile_end_kinds": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), c_i32p]),
}
PROF_KINDS = ("fused", "gemm3", "epilogue", "prep", "other")        # GPDE_PROF_* of include/gpde.h


def profile_begin():
    lib().gpde_profile_begin()


def profile_end():
    """{kind: (ms, launches)} of the kernels gpde_nnconv_fwd launched on this thread since profile_begin()."""
    ms = (ctypes.c_double * len(PROF_KINDS))()
    n = (ctypes.c_int32 * len(PROF_KINDS))()
    check(lib().gpde_profile_end_kinds(ms, n), "gpde_profile_end_kinds")
    return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(PROF_KINDS)}

_lib = None
_lock = threading.Lock()
n_native_calls = 0          # incremented by every gpde_nnconv_fwd call (tests assert it moves)


class GpdeError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load libgpde.so (once). Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise GpdeError(
                        f"{LIB_PATH} not found - build it with `python graph-pde_amd/build.py` "
                        "(or __graft_entry__.build()); the NNConv hot path has no non-HIP fallback")
                l = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(l, name)          # AttributeError if a declared symbol is missing
                    fn.restype, fn.argtypes = res, args
                ver = int(l.gpde_version())
                if ver & GPDE_VERSION_ABLATION and os.environ.get("GPDE_ALLOW_ABLATION") != "1":
                    raise GpdeError(
                        f"{LIB_PATH} is an ABLATION build (gpde_version() = {ver:#x}: arithmetic compiled out, results are "
                        "wrong; timing experiments only) - refused.  GPDE_ALLOW_ABLATION=1 loads it for such an experiment")
                _lib = l
    return _lib


def reload_switches() -> None:
    """Re-read the library's GPDE_* developer switches from the environment (they are read once per process otherwise)."""
    if _lib is not None:
        _lib.gpde_reload_switches()


def check(rc: int, what: str) -> None:
    if rc != GPDE_OK:
        msg = lib().gpde_last_error().decode("utf-8", "replace")
        if rc == -2:
            raise NotImplementedError(f"{what}: {msg}")
        raise GpdeError(f"{what} failed (code {rc}): {msg}")


def dims_array(dims):
    return (ctypes.c_int32 * len(dims))(*[int(d) for d in dims])

"""graph-pde_amd: the MI355X-native edge-conditioned graph convolution (NNConv) of
neuraloperator/graph-pde — hand-written HIP for gfx950 behind the reference's module surface.

    from graph_pde_amd import NNConv_old, NNConv          # drop-in modules (nn_conv.py)
    from graph_pde_amd import ops                          # CSR / packing / raw forward
    fwd = graph_pde_amd.capture(model_fn, x)               # opt-in: the call sequence of a sample as ONE HIP graph (capture.py)

(The directory is named `graph-pde_amd`; `graph_pde_amd.py` at the repo root makes it importable.)
"""
from . import _lib, ops, synth          # noqa: F401
from .nn_conv import ECConv, NNConv, NNConv_old, nnconv_group   # noqa: F401
from .ops import NodeAttr                           # noqa: F401  (opt-in: edge attributes from node data)
from .capture import capture                        # noqa: F401  (opt-in: a model function's native calls as one HIP graph)

__all__ = ["NNConv_old", "NNConv", "ECConv", "NodeAttr", "nnconv_group", "capture", "ops", "synth"]

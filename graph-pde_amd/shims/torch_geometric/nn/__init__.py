"""`torch_geometric.nn`: NNConv is the fused MI355X operator; GCNConv is importable only
(neurips1_MGKN.py:10 imports it without using it; the GCN baseline neurips4_GCN.py is out of scope)."""
import _bootstrap  # noqa: F401
from graph_pde_amd.nn_conv import NNConv  # noqa: F401
from . import conv, inits  # noqa: F401
from .conv import MessagePassing  # noqa: F401


class GCNConv:
    def __init__(self, *a, **k):
        raise NotImplementedError("GCNConv (neurips4_GCN.py baseline) is outside the NNConv hot path")

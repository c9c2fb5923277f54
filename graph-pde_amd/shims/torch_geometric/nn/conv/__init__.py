"""`torch_geometric.nn.conv`: `MessagePassing` is the class the native operator modules derive from
(graph_pde_amd.message_passing; reference import: graph-neural-operator/nn_conv.py:3)."""
import _bootstrap  # noqa: F401
from graph_pde_amd.message_passing import MessagePassing  # noqa: F401

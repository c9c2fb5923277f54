"""`reset` / `uniform` as imported by nn_conv.py:4."""
import _bootstrap  # noqa: F401
from graph_pde_amd.nn_conv import _reset as reset, _uniform as uniform  # noqa: F401

"""Minimal torch_geometric facade (see ../README.md)."""
from . import data, nn  # noqa: F401
__version__ = "0.0.gpde-shim"

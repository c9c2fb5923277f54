"""Make `graph_pde_amd` importable from the shims (repo root = three levels up)."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _root not in sys.path:
    sys.path.insert(0, _root)
import graph_pde_amd  # noqa: E402,F401

"""Drop-in for /root/reference/graph-neural-operator/nn_conv.py: same module name, same class
names.  `NNConv_old` (the class every GKN script instantiates, nn_conv.py:197-286) is the fused
MI355X operator.  `NNConv` (diagonal edge kernel, nn_conv.py:8-96) and `NNConv_Gaussian`
(nn_conv.py:99-194) are dead code in the reference (imported by neurips1_GKN.py:10 /
neurips5_GKN.py:10, never instantiated): the names import, constructing them raises."""
import _bootstrap  # noqa: F401
from graph_pde_amd.nn_conv import NNConv_old  # noqa: F401


class _NotBuilt:
    _what = ""

    def __init__(self, *a, **k):
        raise NotImplementedError(
            f"{type(self).__name__}: {self._what} is never instantiated by any graph-pde script and "
            "is outside the MI355X hot path; use NNConv_old (full edge kernel)")


class NNConv(_NotBuilt):
    _what = "the diagonal-kernel variant (nn_conv.py:8-96)"


class NNConv_Gaussian(_NotBuilt):
    _what = "the Gaussian-RBF variant (nn_conv.py:99-194)"


ECConv = NNConv

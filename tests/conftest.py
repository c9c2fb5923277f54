import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
GOLDEN_CASES = ["ckpt_g16", "ragged_add", "mlp2_mean_noroot", "burgers_k4", "mlp5_g16", "ckpt_torus_m100"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """One committed reference vector (tests/golden/make_golden.py) as torch tensors."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    n_layers = int(z["n_layers"])
    case = {
        "name": name,
        "x": torch.from_numpy(z["x"]),
        "edge_index": torch.from_numpy(z["edge_index"]),
        "edge_attr": torch.from_numpy(z["edge_attr"]),
        "aggr": str(z["aggr"]),
        "weights": [torch.from_numpy(z[f"W{i}"]) for i in range(n_layers)],
        "biases": [torch.from_numpy(z[f"b{i}"]) for i in range(n_layers)],
        "root": torch.from_numpy(z["root"]) if "root" in z.files else None,
        "bias": torch.from_numpy(z["bias"]) if "bias" in z.files else None,
        "out_f32": torch.from_numpy(z["out_f32"]),
        "out_f64": torch.from_numpy(z["out_f64"]),
    }
    return case


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    return load_golden(request.param)


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch, plus: libgpde.so reads its GPDE_* developer switches ONCE per process (no getenv() in launch
    paths), so a test that flips one through setenv / delenv has the library re-read them - and once more when the
    environment is restored."""
    from graph_pde_amd import _lib
    touched = []

    def wrap(fn):
        def inner(name, *a, **k):
            r = fn(name, *a, **k)
            if str(name).startswith("GPDE_"):
                touched.append(name)
                _lib.reload_switches()
            return r
        return inner
    monkeypatch.setenv = wrap(monkeypatch.setenv)
    monkeypatch.delenv = wrap(monkeypatch.delenv)
    yield monkeypatch
    monkeypatch.undo()
    if touched:
        _lib.reload_switches()

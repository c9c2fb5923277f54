"""Developer probe: backward gradient errors against the float64 oracle at several edge counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.nnconv_oracle import rel_l2
from tests.test_gpu_parity import _oracle_grads
from tests.test_gpu_bwd import _case, _native
for e in (3000, 8000, 20000):
    dims, n = [6, 256, 256, 4096], 300
    x, ei, ea, ws_, bs_, root, bias, gout = _case(dims, n, e, 77)
    ref = _oracle_grads(x, ei, ea, ws_, bs_, root, bias, "mean", gout)
    for env in ("", "GPDE_BWD_DW2_F32", "GPDE_BWD_GEMM_F32"):
        for k in ("GPDE_BWD_DW2_F32", "GPDE_BWD_GEMM_F32"):
            os.environ.pop(k, None)
        if env:
            os.environ[env] = "1"
        g = _native(x, ei, ea, ws_, bs_, root, gout)
        print(e, env or "split", "dx %.2e" % rel_l2(g[0].cpu(), ref[0]),
              "dW", ["%.2e" % rel_l2(g[1][l].cpu(), ref[1][l]) for l in range(3)],
              "db", ["%.2e" % rel_l2(g[2][l].cpu(), ref[2][l]) for l in range(3)], flush=True)

"""Developer probe: run-to-run equality of every gradient under memory pressure (1 GiB fills between calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_bwd import _case, _native
from graph_pde_amd import _lib
names = ["dx", "dW1", "dW2", "dW3", "db1", "db2", "db3", "droot", "dbias"]
def flat(o):
    return [o[0]] + list(o[1]) + list(o[2]) + [o[3], o[4]]
fills = [1e-30, float("nan"), -3.0e38, 0.5]
for dims, n, e in (([6, 256, 256, 4096], 200, 9000), ([6, 1024, 1024, 4096], 300, 20000)):
    x, ei, ea, ws_, bs_, root, bias, gout = _case(dims, n, e, 5)
    for variant in ("1", "2"):
        os.environ["GPDE_EDGE_BWD"] = variant
        _lib.reload_switches()         # the library reads its switches once per process
        ref = flat(_native(x, ei, ea, ws_, bs_, root, gout))
        nbad = 0
        for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
            if it % 2:
                t = torch.randn(256 << 20, device="cuda:0") * (10.0 ** ((it % 7) - 3)); del t
            else:
                t = torch.full((256 << 20,), fills[it % 4], device="cuda:0"); del t
            cur = flat(_native(x, ei, ea, ws_, bs_, root, gout))
            msg = []
            for nm, a, b in zip(names, ref, cur):
                if not torch.equal(a, b):
                    dif = (a != b).nonzero()
                    msg.append(f"{nm}: {dif.shape[0]} entries, rows {dif[:, 0].unique()[:12].tolist()} rel {float((a-b).norm()/a.norm()):.1e}")
            if msg:
                nbad += 1
                print(dims, "edge kernel", variant, "run", it, "fill", fills[it % 4], " | ".join(msg), flush=True)
        print(dims, "edge kernel", variant, "bad runs:", nbad, flush=True)

"""The import shims (graph-pde_amd/shims) give the unmodified reference scripts what they import
(SURVEY.md Appendix A); checked in a subprocess so sys.path stays clean."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIMS = os.path.join(REPO, "graph-pde_amd", "shims")

CODE = r'''
import torch, nn_conv, h5py
from torch_geometric.data import Data, DataLoader
from torch_geometric.nn import NNConv, GCNConv
from torch_geometric.nn.conv import MessagePassing
from torch_geometric.nn.inits import reset, uniform
import graph_pde_amd
assert nn_conv.NNConv_old is graph_pde_amd.NNConv_old and NNConv is graph_pde_amd.NNConv
# collation rule: *index* keys offset by cumulative node count and concatenated on the last dim
ds = [Data(x=torch.randn(3, 2), edge_index=torch.tensor([[0, 1], [1, 2]]), y=torch.randn(3),
           sample_idx=torch.tensor([5]), edge_index_range=torch.tensor([[0, 2]])) for _ in range(2)]
loader = DataLoader(ds, batch_size=2, shuffle=False)
assert len(loader) == 1 and loader.dataset[0] is ds[0]
b = next(iter(loader))
assert b.x.shape == (6, 2) and b.y.shape == (6,)
assert b.edge_index.tolist() == [[0, 1, 3, 4], [1, 2, 4, 5]]
assert b.sample_idx.tolist() == [5, 5]                 # 'idx' is not 'index': no offset
assert b.to("cpu") is b and b.num_nodes == 6
# dead-code classes import but do not construct
for cls in (nn_conv.NNConv, nn_conv.NNConv_Gaussian, GCNConv):
    try:
        cls(1, 1, None)
    except NotImplementedError:
        pass
    else:
        raise AssertionError(cls)
# generic MessagePassing facade (not the hot path): mean aggregation
class Avg(MessagePassing):
    def __init__(self): super().__init__(aggr="mean")
    def forward(self, x, ei): return self.propagate(ei, x=x)
    def message(self, x_j): return x_j
    def update(self, aggr_out): return aggr_out
x = torch.tensor([[1.0], [3.0], [5.0]])
out = Avg()(x, torch.tensor([[0, 1, 2], [2, 2, 0]]))
assert out.view(-1).tolist() == [5.0, 0.0, 2.0]
print("ok")
'''


def test_shims_import_surface():
    env = dict(os.environ, PYTHONPATH=SHIMS)
    r = subprocess.run([sys.executable, "-c", CODE], cwd="/tmp", env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]

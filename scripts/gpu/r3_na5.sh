timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, graph_pde_amd as gp
from graph_pde_amd import ops, synth
d = torch.device("cuda:0")
s = 221
ei = synth.lattice_radius_graph(s, 0.10, d)
pos = synth.lattice_positions(s, d); a = synth.darcy_coefficient(s, 0).to(d)
ea = synth.darcy_edge_attr(ei, pos, a)
na = gp.NodeAttr.darcy(pos, a)
mat = na.materialize(ei)
E = ei.shape[1]
bad = (mat != ea).any(dim=1).nonzero().view(-1)
print("E", E, "rows differing", bad.numel(), "first", bad[:5].tolist(), "last", bad[-5:].tolist())
# ground truth on CPU for a few differing edges
eic = ei[:, bad[:4]].cpu(); posc = pos.cpu(); ac = a.cpu()
for k in range(eic.shape[1]):
    j, i = int(eic[0, k]), int(eic[1, k])
    truth = [float(posc[j, 0].float()), float(posc[j, 1].float()), float(posc[i, 0].float()), float(posc[i, 1].float()), float(ac[j]), float(ac[i])]
    print("edge", int(bad[k]), "truth", truth, "\n   ea ", ea[bad[k]].tolist(), "\n   mat", mat[bad[k]].tolist())
# which op: column by column
cols = [na.table[ei[ep].long(), col] for ep, col in na.sel]
for c in range(6):
    print("col", c, "stack == col", torch.equal(mat[:, c], cols[c]), "ea == col", torch.equal(ea[:, c], cols[c]))
PY

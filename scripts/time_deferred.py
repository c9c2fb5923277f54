"""Training step of `depth` x one conv module (UAI1_full_resolution.py:29-30, :258-273) when H does NOT fit memory:
every application with its own full backward (GPDE_DEFERRED_BWD=off) against the depth-deferred backward (light pass per
application + ONE deferred pass).  Developer probe.  usage: time_deferred.py <cfg> [depth] [steps]
   g241: H really does not fit; smaller graphs: the hidden-cache budget is forced to 0 so that the same path runs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import graph_pde_amd as gp
from graph_pde_amd import ops, synth, hidden_cache
cfg = sys.argv[1] if len(sys.argv) > 1 else "g121"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 6
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
s, r = {"g241": (241, 0.1), "g121": (121, 0.1), "g61": (61, 0.1)}[cfg]
dev = torch.device("cuda:0")
if os.environ.get("NODEATTR") == "1":
    # row f3 in training: graph from positions (cell list -> CSR), attributes from the node table: no edge list, no [E,6] tensor
    pos = synth.lattice_positions(s, dev)
    a_n = synth.darcy_coefficient(s, 0).to(dev)
    t0 = time.perf_counter()
    ei = ops.radius_csr(pos, r)
    ea = gp.NodeAttr.darcy(pos, a_n)
    torch.cuda.synchronize()
    n, e = ei.n_nodes, ei.n_edges
    print(f"graph from positions: N={n} E={e} in {1e3 * (time.perf_counter() - t0):.1f} ms, attributes from the node table")
else:
    ei, ea, n = synth.darcy_graph(s, r, device=dev)
    e = ei.shape[1]
if cfg != "g241":
    hidden_cache.BUDGET_BYTES = 0
res = {}
for mode in os.environ.get("MODES", "off,auto").split(","):
    torch.manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(6, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 4096))
    conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(dev)
    fc1, fc2 = torch.nn.Linear(6, 64).to(dev), torch.nn.Linear(64, 1).to(dev)
    params = list(conv.parameters()) + list(fc1.parameters()) + list(fc2.parameters())
    opt = torch.optim.Adam(params, lr=1e-4, weight_decay=5e-4)
    a_in, y_t = torch.randn(n, 6, device=dev), torch.randn(n, device=dev)
    hidden_cache.DEFER_MODE = mode
    hidden_cache.clear()
    ops.clear_caches()
    ts, losses = [], []
    for it in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        h = fc1(a_in)
        for _ in range(depth):
            h = torch.relu(conv(h, ei, ea))
        loss = torch.norm(fc2(h).view(-1) - y_t, 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize()
        ts.append((t1 - t0, t2 - t1, time.perf_counter() - t0))
        losses.append(float(loss.detach()))
    res[mode] = (ts, losses)
    print(f"{cfg} E={e} depth={depth} deferred={mode}: " + "  ".join(f"step{k}: fwd {1e3*a:.0f} bwd {1e3*b:.0f} total {1e3*c:.0f} ms" for k, (a, b, c) in enumerate(ts)) +
          f"  -> {depth*e/ts[-1][2]/1e6:.1f} M-edge-applications/s  losses {['%.6g' % l for l in losses]}  stats {dict(hidden_cache.stats)}  peak {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
if len(res) == 2:
    print(f"last step off/auto = {res['off'][0][-1][2] / res['auto'][0][-1][2]:.2f}x")

"""Developer probe: `attr_reorder` - the slot-order gather of edge_attr (ops.attr_in_slot_order ->
gpde_gather_rows, 12.9 GB of traffic at G241) - took 22 ms on one box and 99 ms on the driver's.  Where does a first call's time
go?  Times, per call with HIP events: (a) the gather into a FRESH torch allocation (first touch of 2.3 GB), (b) the same into a
block torch already holds (second call after freeing the first result), (c) the kernel alone into a preallocated buffer, and
the allocation by itself."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_pde_amd import _lib, ops, synth     # noqa: E402

d = torch.device("cuda:0")
ei, ea, n = synth.darcy_graph(241, 0.10, device=d, seed=0)
csr = ops.csr_for(ei, n)
torch.cuda.synchronize()


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = fn()
    b.record()
    torch.cuda.synchronize()
    return r, 1e3 * (time.perf_counter() - t0), a.elapsed_time(b)


torch.cuda.empty_cache()
_, w, g = timed(lambda: torch.empty(ea.shape[0], ea.shape[1], device=d))
print(f"torch.empty of the 2.3 GB result from the driver (cache emptied): wall {w:.1f} ms")
torch.cuda.empty_cache()
for k in range(3):
    out, w, g = timed(lambda: ops.gather_rows(ea, csr.perm))
    print(f"gather_rows call {k} ({'fresh allocation' if k == 0 else 'cached block'}): wall {w:.1f} ms, device {g:.1f} ms")
    del out
buf = torch.empty(ea.shape[0], ea.shape[1], device=d)
lib = _lib.lib()
for k in range(3):
    def run():
        with torch.cuda.device(d):
            _lib.check(lib.gpde_gather_rows(ea.data_ptr(), int(ea.size(1)), csr.perm.data_ptr(), int(csr.perm.numel()), buf.data_ptr(),
                                            int(torch.cuda.current_stream(d).cuda_stream)), "gpde_gather_rows")
    _, w, g = timed(run)
    print(f"kernel alone into a preallocated buffer, call {k}: wall {w:.1f} ms, device {g:.1f} ms")
# a NEW edge_attr tensor each time, as an epoch hands them over (bench.py's distinct-sample figure)
pos = synth.lattice_positions(241, d)
for k in range(3):
    a_s = synth.darcy_coefficient(241, seed=100 + k).to(d)
    ea_s = synth.darcy_edge_attr(ei, pos, a_s)
    _, w, g = timed(lambda: ops.attr_in_slot_order(csr, ea_s))
    print(f"attr_in_slot_order on a new sample's edge_attr, sample {k}: wall {w:.1f} ms, device {g:.1f} ms")
    del ea_s

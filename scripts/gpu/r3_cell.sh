O=gpurun_out/r3_08
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -12
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import time, torch
from graph_pde_amd import ops, synth
d = torch.device("cuda:0")
for s in (121, 241):
    pos = synth.lattice_positions(s, d)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        csr = ops.radius_csr(pos, 0.10)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ei = ops.radius_graph(pos, 0.10)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ref = ops.build_csr(ei, s * s)
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"s={s} E={csr.n_edges}: cell-list CSR {1e3*(t1-t0):.2f} ms | brute-force COO {1e3*(t2-t1):.2f} ms + sort to CSR {1e3*(t3-t2):.2f} ms | equal {torch.equal(csr.src, ref.src) and torch.equal(csr.rowptr, ref.rowptr)}")
PY

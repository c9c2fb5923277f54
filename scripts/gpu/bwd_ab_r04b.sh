# round-4 backward micro-changes on ONE box: gradient tests, then timing A/B of each switch (scripts/time_bwd.py g121, hidden cache
# off) and a kernel-trace of the default build.  Output: gpurun_out/bwd_ab_r04b/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/bwd_ab_r04b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_deferred.py tests/test_gpu_edgeweights.py tests/test_gpu_repeat.py tests/test_gpu_hidden.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -5
for env in "" "GPDE_NT_NO_PREFETCH=1" "GPDE_TN_NO_KS_XCD=1" "GPDE_NT_NO_PREFETCH=1 GPDE_TN_NO_KS_XCD=1" ""; do
  echo "[$env]"; env $env GPDE_HIDDEN_CACHE=off timeout 300 python scripts/time_bwd.py g121 2>&1 | grep "bwd M-edges" | tail -2
done | tee $O/ab.txt
GPDE_HIDDEN_CACHE=off timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o run -- python $R/scripts/time_bwd.py g121 > $O/stats.log 2>&1; echo "stats rc=$?"
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/bwd_ab_r04b/stats/run_kernel_trace.csv")))
# per-dispatch durations of the split GEMM launches (dW_2 = split-K first, then dU_1) and the per-edge kernel
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    key = "gemm_f16s" if "gemm_f16s_nt_kernel<false>" in n else "edge_bwd2" if "edge_bwd2" in n else None
    if key: d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
g = d["gemm_f16s"]
print("split GEMM launches (ms), alternating dW_2 / dU_1:", [round(v, 2) for v in g[-22:]])
print("edge_bwd2 (ms):", [round(v, 2) for v in d["edge_bwd2"][-11:]])
PY
head -12 $O/stats/run_kernel_stats.csv | cut -c1-160
find $O -type f -size +2M -delete

mkdir -p gpurun_out/r2u
timeout 600 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_hidden.py tests/test_gpu_models.py -x -q -m gpu > gpurun_out/r2u/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r2u/pytest.log | tail -3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
GPDE_HIDDEN_CACHE=off timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/r2u/bwd -o run -- python scripts/time_bwd.py g121 > gpurun_out/r2u/bwd.log 2>&1 < /dev/null; echo "trace rc=$?"
grep "bwd M-edges" gpurun_out/r2u/bwd.log | tail -1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2u/bwd/**/*kernel_stats.csv', recursive=True)
if f:
    for i,r in enumerate(csv.DictReader(open(f[0]))):
        if i<8: print(r['Name'][:90], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2),'ms', r['Percentage'])
PY
find gpurun_out/r2u -type f -size +2M -delete

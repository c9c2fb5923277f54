mkdir -p gpurun_out/r3d
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3d/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r3d/pytest.log | tail -6
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
GPDE_HIDDEN_CACHE=off timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/r3d/bwd -o run -- python scripts/time_bwd.py g121 > gpurun_out/r3d/bwd.log 2>&1 < /dev/null; echo "trace rc=$?"
grep "bwd M-edges" gpurun_out/r3d/bwd.log | tail -1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3d/bwd/**/*kernel_stats.csv', recursive=True)
if f:
    for i,r in enumerate(csv.DictReader(open(f[0]))):
        if i<10: print(r['Name'][:90], r['Calls'], round(float(r['TotalDurationNs'])/3e6,2),'ms/bwd', r['Percentage'])
PY
find gpurun_out/r3d -type f -size +2M -delete

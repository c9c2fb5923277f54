mkdir -p gpurun_out/r2n
timeout 600 python -m pytest tests/test_gpu_v6.py -x -q -m gpu > gpurun_out/r2n/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r2n/pytest.log | tail -3
for rep in 1 2; do for prec in f16split f16split_static; do
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reuse-probe --no-mgkn --no-alt --precision $prec > gpurun_out/r2n/bench_${prec}_$rep.log 2>&1 < /dev/null
python - <<PY
import json
for l in open('gpurun_out/r2n/bench_${prec}_$rep.log'):
    if l.startswith('{'):
        j=json.loads(l); print('$prec', $rep, j['value'], j['median_step_ms'], j['roofline']['avg_launch_ms'], j['roofline']['frac'])
PY
done; done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-reuse-probe --no-mgkn --no-alt --steps 1 --warmup 0"
timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r2n/fetch_queue -o run -- $B > gpurun_out/r2n/fetch_queue.log 2>&1 < /dev/null; echo "fetch rc=$?"
python - <<'PY'
import csv,glob
for tag in ('queue',):
    f=glob.glob(f'gpurun_out/r2n/fetch_{tag}/**/*counter_collection.csv', recursive=True)
    if not f: print(tag,'no csv'); continue
    v=[float(r['Counter_Value']) for r in csv.DictReader(open(f[0])) if 'gpde_fused_f16v6' in r['Kernel_Name']]
    print(tag, 'FETCH raw GB per launch:', [round(x*1024/1e9,2) for x in v])
PY
find gpurun_out/r2n -type f -size +2M -delete

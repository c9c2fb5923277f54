mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests/test_gpu_v6.py -x -q -m gpu > gpurun_out/r2b/pytest_v6.log 2>&1; echo "pytest_v6 rc=$?"
tail -25 gpurun_out/r2b/pytest_v6.log
for prec in f16split f16split_8wave; do
timeout 300 python bench.py --config g121 --steps 5 --warmup 2 --no-cpu-baseline --no-reuse-probe --precision $prec > gpurun_out/r2b/bench_g121_$prec.log 2>&1; echo "bench $prec rc=$?"; python -c "
import json,sys
for l in open('gpurun_out/r2b/bench_g121_$prec.log'):
    if l.startswith('{'):
        j=json.loads(l); print('$prec', j['value'], j['ms_per_step'], j['alt_precision'])
"
done

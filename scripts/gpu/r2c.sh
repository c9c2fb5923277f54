mkdir -p gpurun_out/r2c
for prec in f16split f16split_8wave; do
timeout 600 python bench.py --config g241 --steps 5 --warmup 2 --no-cpu-baseline --no-reuse-probe --precision $prec > gpurun_out/r2c/bench_g241_$prec.log 2>&1; echo "bench $prec rc=$?"; python -c "
import json,sys
for l in open('gpurun_out/r2c/bench_g241_$prec.log'):
    if l.startswith('{'):
        j=json.loads(l); print('$prec', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['alt_precision'])
"
done

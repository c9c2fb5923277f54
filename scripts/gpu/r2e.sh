mkdir -p gpurun_out/r2e
timeout 900 python -m pytest tests/test_gpu_v6.py -x -q -m gpu > gpurun_out/r2e/pytest_v6.log 2>&1; echo "pytest_v6 rc=$?"; tail -5 gpurun_out/r2e/pytest_v6.log
for cfg in g241; do GPDE_LIB=$PWD/graph-pde_amd/libgpde_T6.so timeout 300 python scripts/v6_timing.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2e/timing.txt; done
for prec in f16split; do
timeout 600 python bench.py --config g241 --steps 5 --warmup 2 --no-cpu-baseline --no-reuse-probe --precision $prec > gpurun_out/r2e/bench_g241_$prec.log 2>&1; echo "bench $prec rc=$?"; python -c "
import json,sys
for l in open('gpurun_out/r2e/bench_g241_$prec.log'):
    if l.startswith('{'):
        j=json.loads(l); print('$prec', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['alt_precision'])
"
done

mkdir -p gpurun_out/r2h
timeout 1700 python -m pytest tests -x -q -m gpu -s > gpurun_out/r2h/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r2h/pytest_gpu.log | tail -25
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r2h/bench_default.log 2> gpurun_out/r2h/bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/r2h/bench_default.err; python - <<'PY'
import json
for l in open('gpurun_out/r2h/bench_default.log'):
    if l.startswith('{'):
        j=json.loads(l)
        print(j['value'], j['ms_per_step'], j['median_step_ms'], j['rel_l2_sample'])
        print(json.dumps(j['roofline'])[:900])
        print(json.dumps(j['alt_precision']))
        print(json.dumps(j['cpu_baseline']))
        print(json.dumps(j['mgkn'], indent=0)[:2500])
PY
timeout 600 python bench.py --train --steps 3 --warmup 1 > gpurun_out/r2h/bench_train.log 2>&1; echo "train rc=$?"; grep "^{" gpurun_out/r2h/bench_train.log | head -2
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --train --gpus 1 --steps 3 --warmup 1 > gpurun_out/r2h/bench_train_torchrun.log 2>&1; echo "train torchrun rc=$?"; grep "^{" gpurun_out/r2h/bench_train_torchrun.log | head -2

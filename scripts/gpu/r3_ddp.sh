O=gpurun_out/r3_07
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_headline.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -25
timeout 300 python bench.py --gpus 2 --steps 1 --warmup 0 > $O/gpus2.log 2>&1; echo "bench --gpus 2 on a 1-GPU box: rc=$?"; tail -2 $O/gpus2.log

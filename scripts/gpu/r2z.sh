mkdir -p gpurun_out/r2z
timeout 900 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_mgkn.py tests/test_gpu_hidden.py tests/test_gpu_models.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r2z/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r2z/pytest.log | tail -12

find gpurun_out/r2z -type f -size +2M -delete

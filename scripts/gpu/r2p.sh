mkdir -p gpurun_out/r2p
timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_mgkn.py -x -q -m gpu > gpurun_out/r2p/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r2p/pytest.log | tail -15

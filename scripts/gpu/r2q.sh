mkdir -p gpurun_out/r2q
timeout 900 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_mgkn.py -x -q -m gpu > gpurun_out/r2q/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r2q/pytest.log | tail -12
for v in 0 1; do
  if [ $v = 1 ]; then export GPDE_BWD_GEMM_F32=1; fi
  GPDE_HIDDEN_CACHE=off timeout 300 python scripts/time_bwd.py g121 2>&1 < /dev/null | grep -v amdgpu.ids | tail -1
done

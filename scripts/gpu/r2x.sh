mkdir -p gpurun_out/r2x
timeout 800 python scripts/dbg_mgkn_flaky.py 24 2>&1 < /dev/null | grep -v amdgpu.ids | tail -6
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2x/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r2x/pytest.log | tail -4

mkdir -p gpurun_out/r2i
timeout 1700 python -m pytest tests -q -m gpu -s > gpurun_out/r2i/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r2i/pytest_gpu.log | grep -v "^$" | tail -12
timeout 300 python scripts/mgkn_levels.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2i/mgkn_levels.txt
bash scripts/gpu/profile.sh r02

mkdir -p gpurun_out/r2k
timeout 900 python -m pytest tests/test_gpu_mgkn.py tests/test_gpu_parity.py tests/test_gpu_hidden.py tests/test_gpu_models.py -x -q -m gpu > gpurun_out/r2k/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r2k/pytest.log | tail -6
timeout 300 python scripts/mgkn_levels.py 2>&1 < /dev/null | grep -v amdgpu.ids | tee gpurun_out/r2k/mgkn_levels.txt
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2k/mgkn_bench.txt
import sys, json, torch
sys.path.insert(0, '.')
import bench
r = bench.mgkn_probe(torch.device('cuda:0'))
for k, v in r.items():
    print(k, v['ms_per_forward'], v['ms_per_forward_fused_glue'], v['fused_glue_equals_unfused'], v['max_rel_l2_vs_oracle'], v['kernel_time_share'])
PY

mkdir -p gpurun_out/r2y
timeout 300 python scripts/mgkn_levels.py > gpurun_out/r2y/mgkn.log 2>&1 < /dev/null; echo "mgkn rc=$?"; grep -v amdgpu.ids gpurun_out/r2y/mgkn.log | tail -28
timeout 900 python -m pytest tests/test_gpu_mgkn.py tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_repeat.py tests/test_gpu_hidden.py -x -q -m gpu > gpurun_out/r2y/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r2y/pytest.log | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-alt --no-reuse-probe --no-backward-probe 2>/dev/null < /dev/null | tail -1 > gpurun_out/r2y/bench.json; python - <<'PY'
import json
j=json.load(open('gpurun_out/r2y/bench.json'))
print(j['value'], j['roofline']['kernel_ms_per_step'], {k:(v['ms_per_forward'],v['ms_per_forward_fused_glue'],v['max_rel_l2_vs_oracle']) for k,v in j['mgkn'].items()})
PY

mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_edgepath.py tests/test_gpu_mgkn.py tests/test_gpu_parity.py tests/test_gpu_hidden.py tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/r3a/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r3a/pytest.log | tail -15
timeout 300 python scripts/mgkn_levels.py 2>&1 < /dev/null | grep -v amdgpu.ids | head -6
timeout 600 python bench.py --no-cpu-baseline --no-alt --no-reuse-probe --no-backward-probe --steps 3 --warmup 1 2>/dev/null < /dev/null | tail -1 > gpurun_out/r3a/bench.json; python - <<'PY'
import json
j=json.load(open('gpurun_out/r3a/bench.json'))
print(j['value'], {k:(v['ms_per_forward'],v['ms_per_forward_fused_glue'],v['max_rel_l2_vs_oracle']) for k,v in j['mgkn'].items()})
PY

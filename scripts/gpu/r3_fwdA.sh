O=gpurun_out/r3_09
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_v6.py tests/test_gpu_repeat.py tests/test_gpu_parity.py tests/test_gpu_hidden.py tests/test_gpu_edgepath.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -4
GPDE_LIB=$PWD/graph-pde_amd/libgpde_T6.so timeout 300 python scripts/v6_timing.py g241 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-alt --no-mgkn --no-reuse-probe --no-backward-probe --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['rel_l2_sample'])"

O=gpurun_out/r3_01
mkdir -p $O
echo "== repro lib (round-2 buffer plan, guards off) under skew" > $O/repro.log
GPDE_LIB=$PWD/graph-pde_amd/libgpde_repro.so timeout 600 python -m pytest tests/test_gpu_repeat.py -q -m gpu -k "workgroup_timing" 2>&1 | grep -v amdgpu.ids | grep -E "AssertionError|passed|failed" >> $O/repro.log
echo "== fixed lib" >> $O/repro.log
timeout 900 python -m pytest tests/test_gpu_repeat.py -q -m gpu -k "workgroup_timing or backward_is_repro" 2>&1 | grep -v amdgpu.ids | tail -5 >> $O/repro.log
cat $O/repro.log
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -5

# Round 3: proof of the root cause of round 2's intermittent wrong grad_W1 - record: profiles/r03_bwd_race_repro.log.
# libgpde_repro.so was a one-off build of the ROUND-2 buffer plan (csrc/gpde_bwd.hip: `int nb_ = 0;` in mlp_backward, the
# overlap guards of gpde_launch_gemm / gpde_launch_gemm_f16s_nt disabled) made with
#     GPDE_BUILD_SUFFIX=_repro python graph-pde_amd/build.py
# from a temporarily edited tree; it is not kept.  With GPDE_DEBUG_SKEW_US (odd column slices start 150 us late) that build
# fails 8 of 8 cases; the fixed library returns the bits of the unskewed run.
O=gpurun_out/r3_01
mkdir -p $O
echo "== repro lib (round-2 buffer plan, guards off) under skew" > $O/repro.log
GPDE_LIB=$PWD/scripts/ubench/lib/libgpde_repro.so timeout 600 python -m pytest tests/test_gpu_repeat.py -q -m gpu -k "workgroup_timing" 2>&1 | grep -v amdgpu.ids | grep -E "AssertionError|passed|failed" >> $O/repro.log
echo "== fixed lib" >> $O/repro.log
timeout 900 python -m pytest tests/test_gpu_repeat.py -q -m gpu -k "workgroup_timing or backward_is_repro" 2>&1 | grep -v amdgpu.ids | tail -5 >> $O/repro.log
cat $O/repro.log
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -5

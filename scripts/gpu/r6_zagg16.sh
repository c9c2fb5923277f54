cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r6_zagg16; mkdir -p $O
# the Z re-aggregation of the backward on split f16 (default) against the fp32 kernel (GPDE_BWD_ZAGG_F32=1): G241 depth-6 step
MODES=auto timeout 600 python scripts/time_deferred.py g241 6 4 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400 | tee $O/g241_step_f16.txt
GPDE_BWD_ZAGG_F32=1 MODES=auto timeout 600 python scripts/time_deferred.py g241 6 4 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400 | tee $O/g241_step_f32.txt
timeout 1700 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_deferred.py tests/test_gpu_keep_hidden.py tests/test_gpu_repeat.py tests/test_gpu_nodeattr_train.py tests/test_gpu_headline_bwd.py tests/test_gpu_headline_train.py tests/test_gpu_hidden.py tests/test_gpu_dldh_accumulate.py tests/test_gpu_regime_walk.py tests/test_gpu_capture.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids $O/pytest.log | grep "passed\|failed\|^FAILED\|^E  " | cut -c1-220 | tail -15

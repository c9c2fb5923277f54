cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r6_val; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_deferred.py tests/test_gpu_keep_hidden.py tests/test_gpu_repeat.py tests/test_gpu_nodeattr_train.py tests/test_gpu_headline_bwd.py tests/test_gpu_hidden.py tests/test_gpu_dldh_accumulate.py tests/test_gpu_onepass.py tests/test_gpu_parity.py tests/test_gpu_regime_walk.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids $O/pytest.log | grep "passed\|failed\|^FAILED\|^E  " | cut -c1-220 | tail -15
bash scripts/gpu/profile_bwd.sh r06k > $O/prof.log 2>&1; tail -4 $O/prof.log
MODES=auto timeout 600 python scripts/time_deferred.py g241 6 4 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400 | tee $O/g241_step.txt

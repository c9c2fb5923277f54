cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r6_mgkn_iter; mkdir -p $O
timeout 300 python scripts/time_mgkn_train.py all 9 2>&1 | grep "train step" | cut -c1-120


timeout 300 python scripts/time_mgkn_capture.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
timeout 2000 python -m pytest tests/test_gpu_we_accumulate.py tests/test_gpu_mgkn.py tests/test_gpu_edgeweights.py tests/test_gpu_bwd.py tests/test_gpu_hidden.py tests/test_gpu_dldh_accumulate.py tests/test_gpu_capture.py tests/test_gpu_parity.py tests/test_gpu_v6.py tests/test_gpu_headline.py tests/test_gpu_models.py tests/test_gpu_hypothesis.py tests/test_gpu_edgepath.py tests/test_gpu_reference_shapes.py tests/test_gpu_deferred.py tests/test_gpu_headline_bwd.py tests/test_gpu_regime_walk.py tests/test_gpu_keep_hidden.py tests/test_gpu_repeat.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids $O/pytest.log | grep "passed\|failed\|^FAILED\|^E  " | cut -c1-220 | tail -15

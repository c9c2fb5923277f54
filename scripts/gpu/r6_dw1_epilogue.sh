cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do
  for env in "GPDE_X=0" "GPDE_BWD_DW1_PASS=1"; do
    echo "[$env] $(env $env GPDE_HIDDEN_CACHE=off timeout 300 python scripts/time_bwd.py g121 2>&1 | grep 'bwd M-edges' | tail -1)"
  done
done
MODES=auto timeout 500 python scripts/time_deferred.py g241 6 3 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/losses.*stats/stats/' | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_deferred.py tests/test_gpu_repeat.py tests/test_gpu_headline_bwd.py tests/test_gpu_nodeattr_train.py tests/test_gpu_regime_walk.py tests/test_gpu_keep_hidden.py -q -m gpu 2>&1 | grep -v amdgpu.ids | grep "passed\|failed\|^FAILED\|^E  " | cut -c1-200 | tail -8

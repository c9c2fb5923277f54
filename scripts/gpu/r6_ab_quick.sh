cd $GRAFT_REPO_ROOT
python scripts/dbg_dw1b.py 2>&1 | grep -v amdgpu.ids | head -3
python scripts/dbg_dw1.py 41 2>&1 | grep -v amdgpu.ids | head -2
for rep in 1 2 3; do
  for env in "" "GPDE_BWD_DW1_PASS=1" "GPDE_BWD_H1_IMAGE=1" "GPDE_BWD_WS_FRACTION=0.6"; do
    echo "[$env] $(env $env GPDE_HIDDEN_CACHE=off timeout 300 python scripts/time_bwd.py g121 2>&1 | grep 'bwd M-edges' | tail -1)"
  done
done
timeout 900 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_deferred.py tests/test_gpu_repeat.py tests/test_gpu_headline_bwd.py -q -m gpu 2>&1 | grep -v amdgpu.ids | grep "passed\|failed\|^FAILED\|^E  " | cut -c1-200 | tail -8

# backward tier + timing A/B of the backward switches on ONE box (scripts/time_bwd.py g121, hidden cache off)
O=gpurun_out/bwd_check
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_repeat.py tests/test_gpu_parity.py tests/test_gpu_hidden.py tests/test_gpu_ddp.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -4
for env in "" "GPDE_BWD_DU_PASSES=1" "GPDE_BWD_DU_PASSES=1 GPDE_BWD_H1_MATERIALIZE=1" ""; do
  echo "[$env]"; env $env GPDE_HIDDEN_CACHE=off timeout 300 python scripts/time_bwd.py g121 2>&1 | grep "bwd M-edges" | tail -1
done

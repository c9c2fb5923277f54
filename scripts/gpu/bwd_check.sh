# backward tier + timing A/B of the backward switches on ONE box (scripts/time_bwd.py g121, hidden cache off)
O=gpurun_out/bwd_check
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_repeat.py tests/test_gpu_parity.py tests/test_gpu_hidden.py tests/test_gpu_models.py tests/test_gpu_ddp.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -6
for env in "" "GPDE_SAVE_Z_GB=0" "GPDE_SAVE_Z_GB=0 GPDE_BWD_DU_PASSES=1 GPDE_BWD_H1_MATERIALIZE=1" ""; do
  echo "[$env]"; env $env GPDE_HIDDEN_CACHE=off timeout 300 python scripts/time_bwd.py g121 2>&1 | grep "bwd M-edges" | tail -1
done
for env in "" "GPDE_SAVE_Z_GB=0"; do
  echo "[train s=61 depth 6, hidden cache auto: $env]"; env $env MODES=auto timeout 300 python scripts/time_depth.py g61 2>&1 | grep "depth=" | cut -c1-190
done

# the three reference scripts, unmodified, through the import shims on an MI355X (files staged by scripts/stage_reference.py)
O=gpurun_out/refscripts
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_reference_scripts.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -3
for s in UAI1_full_resolution MGKN_general_darcy2d MGKN_orthogonal_burgers1d; do
  ( time timeout 900 python scripts/run_reference_script.py $s.py --set ntrain=4 --set ntest=2 --set epochs=2 ) > $O/$s.log 2>&1; echo "$s rc=$?"; grep -v amdgpu.ids $O/$s.log | grep -v "^$" | tail -4
done

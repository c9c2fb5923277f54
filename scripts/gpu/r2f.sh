mkdir -p gpurun_out/r2f
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu > gpurun_out/r2f/pytest_boundary.log 2>&1; echo "boundary rc=$?"; tail -15 gpurun_out/r2f/pytest_boundary.log
cd /tmp
for s in UAI1_full_resolution.py MGKN_general_darcy2d.py MGKN_orthogonal_burgers1d.py; do
  nt=2; [ $s = UAI1_full_resolution.py ] || nt=1
  ( time timeout 900 python $GRAFT_REPO_ROOT/scripts/run_reference_script.py $s --set ntrain=2 --set ntest=$nt --set epochs=1 ) > $GRAFT_REPO_ROOT/gpurun_out/r2f/ref_$s.log 2>&1; echo "$s rc=$?"; grep -v "amdgpu.ids\|Warning\|warn" $GRAFT_REPO_ROOT/gpurun_out/r2f/ref_$s.log | tail -8
done

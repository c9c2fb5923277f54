mkdir -p gpurun_out/r2g
cd /tmp
for s in MGKN_general_darcy2d.py; do
  ( time timeout 900 python $GRAFT_REPO_ROOT/scripts/run_reference_script.py $s --set ntrain=2 --set ntest=1 --set epochs=1 ) > $GRAFT_REPO_ROOT/gpurun_out/r2g/ref_$s.log 2>&1; echo "$s rc=$?"; grep -v "amdgpu.ids\|Warning\|warn" $GRAFT_REPO_ROOT/gpurun_out/r2g/ref_$s.log | tail -12
done

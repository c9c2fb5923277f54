# the one-pass kernel for the LIGHT pass at G241 (no kept H competes there) - same box, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_onepass; mkdir -p $O
for rep in 1 2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export GPDE_BWD_ONE_PASS=1; else unset GPDE_BWD_ONE_PASS; fi
    MODES=auto timeout 600 python scripts/time_deferred.py g241 6 4 2>&1 | grep -v amdgpu.ids | tail -6 | sed "s/^/one_pass=$v: /"
  done
done | tee $O/g241_onepass_light.txt
unset GPDE_BWD_ONE_PASS
timeout 300 python scripts/time_attr_reorder.py 2>&1 | grep -v amdgpu.ids | tee $O/attr_reorder.txt
cd /tmp
for arm in "" "--composite"; do
  /usr/bin/time -f "UAI7_evaluate.py $arm: %e s wall" timeout 600 python $GRAFT_REPO_ROOT/scripts/run_reference_script.py UAI7_evaluate.py --set ntrain=2 --set ntest=1 --set epochs=1 $arm 2>&1 | grep -v amdgpu.ids | grep "^0 \|wall\|preprocessing\|calls" | cut -c1-200
done | tee $GRAFT_REPO_ROOT/gpurun_out/r6_onepass/uai7_arms.txt

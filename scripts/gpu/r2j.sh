timeout 900 bash scripts/gpu/profile.sh r02 < /dev/null
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/prof_r02/mgkn -o run -- python scripts/mgkn_levels.py > gpurun_out/prof_r02/mgkn.log 2>&1 < /dev/null; echo "mgkn trace rc=$?"
find gpurun_out/prof_r02 -type f -size +2M -delete
find gpurun_out/prof_r02/mgkn -type f < /dev/null | head
du -sh gpurun_out < /dev/null

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/train61
mkdir -p $O
MODES=auto timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/t -o run -- python scripts/time_depth.py g61 > $O/log.txt 2>&1; echo rc=$?; grep "depth=" $O/log.txt
find $O -type f -size +2M -delete

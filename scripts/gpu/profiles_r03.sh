# Round-3 records: headline kernel stats + PMC passes (scripts/gpu/profile.sh), backward trace, MGKN-general trace
bash scripts/gpu/profile.sh r03 < /dev/null | grep "rc="
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03
GPDE_HIDDEN_CACHE=off timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/bwd -o run -- python scripts/time_bwd.py g121 > $O/bwd.log 2>&1; echo "bwd trace rc=$?"; grep "bwd M-edges" $O/bwd.log | tail -1
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/mgkn4 -o run -- python scripts/time_mgkn_we.py mgkn_general_darcy2d > $O/mgkn4.log 2>&1; echo "mgkn4 trace rc=$?"; grep "per forward" $O/mgkn4.log
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/mgkn3 -o run -- python scripts/time_mgkn_we.py mgkn_orthogonal_burgers1d grouped > $O/mgkn3.log 2>&1; echo "mgkn3 trace rc=$?"; grep "per forward" $O/mgkn3.log
find $O -type f -size +2M -delete
find $O -name "*kernel_stats.csv" | head

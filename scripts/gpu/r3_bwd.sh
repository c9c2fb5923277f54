O=gpurun_out/r3_04
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_repeat.py tests/test_gpu_parity.py tests/test_gpu_hidden.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -8
GPDE_HIDDEN_CACHE=off timeout 300 python scripts/time_bwd.py g121 2>&1 | grep "bwd M-edges" | tail -1
GPDE_BWD_H1_MATERIALIZE=1 GPDE_BWD_ROWSCALE_PASS=1 GPDE_HIDDEN_CACHE=off timeout 300 python scripts/time_bwd.py g121 2>&1 | grep "bwd M-edges" | tail -1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
GPDE_HIDDEN_CACHE=off timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/bwd -o run -- python scripts/time_bwd.py g121 > $O/bwd.log 2>&1; echo "trace rc=$?"
find $O -name "*kernel_stats.csv" | head -2
find $O -type f -size +2M -delete

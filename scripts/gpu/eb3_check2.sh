cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/eb3
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_hidden.py tests/test_gpu_edgeweights.py tests/test_gpu_headline.py tests/test_gpu_models.py -x -q -m gpu > $O/pytest2.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest2.log | tail -4
for env in "" "GPDE_BWD_WS_FRACTION=0" ""; do
  echo "[$env]"; env $env GPDE_HIDDEN_CACHE=off timeout 300 python scripts/time_bwd.py g121 2>&1 | grep "bwd M-edges" | tail -2
done | tee $O/ab2.txt
GPDE_HIDDEN_CACHE=off timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats2 -o run -- python scripts/time_bwd.py g121 > $O/stats2.log 2>&1; echo "stats rc=$?"
head -22 $O/stats2/run_kernel_stats.csv | cut -c1-150
MODES=auto timeout 300 python scripts/time_depth.py g121 2>&1 | grep "depth=" | cut -c1-200
find $O -type f -size +2M -delete

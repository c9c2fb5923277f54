# split-f16 per-edge backward kernel: gradient tests (all three kernel variants), then timing A/B against the fp32 staged kernel
# and against the separate transposing pass over dU_2
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/eb3
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_hidden.py tests/test_gpu_edgeweights.py tests/test_gpu_repeat.py tests/test_gpu_headline.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -12
for env in "" "GPDE_BWD_DU_TRANSPOSE_PASS=1" "GPDE_EDGE_BWD=2" ""; do
  echo "[$env]"; env $env GPDE_HIDDEN_CACHE=off timeout 300 python scripts/time_bwd.py g121 2>&1 | grep "bwd M-edges" | tail -2
done | tee $O/ab.txt
GPDE_HIDDEN_CACHE=off timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o run -- python scripts/time_bwd.py g121 > $O/stats.log 2>&1; echo "stats rc=$?"
head -16 $O/stats/run_kernel_stats.csv | cut -c1-150
find $O -type f -size +2M -delete

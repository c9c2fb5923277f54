# round 6: first hidden layer generated inside the backward's split GEMMs (gpde_gemm_f16s_nt_kernel<false, 1 | 2>): gradients,
# then same-box A/B against rounds 3-5's image path (GPDE_BWD_H1_IMAGE=1), then a kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r6_h1gen; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_headline_bwd.py tests/test_gpu_bwd.py tests/test_gpu_deferred.py tests/test_gpu_keep_hidden.py tests/test_gpu_repeat.py tests/test_gpu_nodeattr_train.py -q -m gpu -s > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids $O/pytest.log | grep "dW1\|passed\|failed\|Error\|error" | cut -c1-400 | tail -20
for rep in 1 2; do
  for env in "" "GPDE_BWD_DW1_PASS=1" "GPDE_BWD_H1_IMAGE=1"; do
    echo "[$env]"; env $env GPDE_HIDDEN_CACHE=off timeout 300 python scripts/time_bwd.py g121 2>&1 | grep "bwd M-edges" | tail -1
  done
done | tee $O/ab.txt
GPDE_HIDDEN_CACHE=off timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o run -- python $R/scripts/time_bwd.py g121 > $O/stats.log 2>&1; echo "stats rc=$?"
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-160
cp $f $O/kernel_stats.csv; rm -rf $O/stats

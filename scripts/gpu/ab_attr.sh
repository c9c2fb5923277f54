O=gpurun_out/ab_attr
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_v6.py tests/test_gpu_hidden.py tests/test_gpu_parity.py tests/test_gpu_bwd.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -3
B="python bench.py --no-cpu-baseline --no-alt --no-mgkn --no-reuse-probe --no-backward-probe --steps 6 --warmup 2"
for rep in 1 2; do
  for v in 0 1; do
    GPDE_ATTR_SLOT_ORDER=$v timeout 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slot_order=$v', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
  done
done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B1="python bench.py --no-cpu-baseline --no-alt --no-mgkn --no-reuse-probe --no-backward-probe --steps 1 --warmup 0"
for v in 0 1; do
  GPDE_ATTR_SLOT_ORDER=$v timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $O/fetch$v -o run -- $B1 > $O/fetch$v.log 2>&1; echo "fetch$v rc=$?"
  GPDE_ATTR_SLOT_ORDER=$v timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $O/write$v -o run -- $B1 > $O/write$v.log 2>&1; echo "write$v rc=$?"
done
find $O -type f -size +2M -delete

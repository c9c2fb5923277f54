cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for X in boundary bwd capture ddp deferred dldh_accumulate edgepath edgeweights graph hidden headline_bwd headline; do
  timeout 900 python -m pytest tests/test_gpu_$X.py tests/test_gpu_hypothesis.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | grep "passed\|failed" | tail -1 | sed "s/^/$X: /"
done

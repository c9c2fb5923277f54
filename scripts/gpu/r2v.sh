mkdir -p gpurun_out/r2v
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2v/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r2v/pytest.log | tail -4
timeout 900 python bench.py > gpurun_out/r2v/bench.json 2> gpurun_out/r2v/bench.err < /dev/null; echo "bench rc=$?"
grep "\[bench\]" gpurun_out/r2v/bench.err | tail -12
bash scripts/gpu/profile.sh r02 < /dev/null
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
GPDE_HIDDEN_CACHE=off timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/r2v/bwd -o run -- python scripts/time_bwd.py g121 > gpurun_out/r2v/bwd.log 2>&1 < /dev/null; echo "trace rc=$?"
grep "bwd M-edges" gpurun_out/r2v/bwd.log | tail -1
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/r2v/mgkn -o run -- python scripts/mgkn_levels.py > gpurun_out/r2v/mgkn.log 2>&1 < /dev/null; echo "mgkn rc=$?"
find gpurun_out/r2v -type f -size +2M -delete

mkdir -p gpurun_out/r3b
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | grep -v amdgpu.ids | tail -2
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3b/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r3b/pytest.log | tail -3
timeout 900 python bench.py > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err < /dev/null; echo "bench rc=$?"
grep "\[bench\]" gpurun_out/r3b/bench.err | tail -6
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-mgkn --no-reuse-probe --no-backward-probe 2>/dev/null < /dev/null | tail -1 | cut -c1-200
bash scripts/gpu/profile.sh r02 < /dev/null | grep "rc="
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
GPDE_HIDDEN_CACHE=off timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/r3b/bwd -o run -- python scripts/time_bwd.py g121 > gpurun_out/r3b/bwd.log 2>&1 < /dev/null; echo "trace rc=$?"
grep "bwd M-edges" gpurun_out/r3b/bwd.log | tail -1
timeout 300 python scripts/mgkn_levels.py > gpurun_out/r3b/mgkn.log 2>&1 < /dev/null; echo "mgkn rc=$?"
find gpurun_out/r3b gpurun_out/prof_r02 -type f -size +2M -delete

#!/bin/bash
# One GPU-box pass over everything the round's numbers rest on (run through gpurun from the repo root):
#   smoke() -> full GPU test tier -> default bench line -> the bench under torch.distributed.run (N = 1) ->
#   rocprofv3 passes of scripts/gpu/profile.sh -> kernel trace of the backward probe -> per-level MGKN times.
# Raw output: gpurun_out/validate/ and gpurun_out/prof_<tag>/; scripts/collect_profiles.py makes the profiles/ files.
TAG=${1:-r03}
O=gpurun_out/validate
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | grep -v amdgpu.ids | tail -2
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -3
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null; echo "bench rc=$?"
grep "\[bench\]" $O/bench.err | tail -6
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-mgkn --no-reuse-probe --no-backward-probe 2>/dev/null < /dev/null | tail -1 | cut -c1-200
bash scripts/gpu/profile.sh $TAG < /dev/null | grep "rc="
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
GPDE_HIDDEN_CACHE=off timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/bwd -o run -- python scripts/time_bwd.py g121 > $O/bwd.log 2>&1 < /dev/null; echo "trace rc=$?"
grep "bwd M-edges" $O/bwd.log | tail -1
timeout 300 python scripts/mgkn_levels.py > $O/mgkn.log 2>&1 < /dev/null; echo "mgkn rc=$?"
find $O gpurun_out/prof_$TAG -type f -size +2M -delete

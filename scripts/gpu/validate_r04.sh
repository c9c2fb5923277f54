#!/bin/bash
# Round 4: ONE GPU-box pass over everything the round's numbers rest on (run through gpurun from the repo root):
#   smoke -> full GPU tier -> default bench line -> bench under torch.distributed.run (N = 1, RCCL) -> --split-graph (N = 1, both
#   partition ways) -> rocprofv3 of the headline bench (stats + FETCH/WRITE/busy PMC passes) -> the same of ONE backward ->
#   depth-6 training steps (s=121 forced-deferred with kernel stats; G241 default policy, tensor and node-table attributes) ->
#   MGKN training steps -> the three reference scripts.
# Raw output under gpurun_out/; scripts/collect_profiles.py r04, collect_profiles_bwd.py r04 make the profiles/ files.
TAG=r04
O=gpurun_out/validate_r04
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | grep -v amdgpu.ids | tail -2
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -6
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null; echo "bench rc=$?"
grep "\[bench\]" $O/bench.err | tail -8
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-mgkn --no-reuse-probe --no-backward-probe 2>/dev/null < /dev/null | tail -1 > $O/bench_torchrun_n1.json; cut -c1-160 $O/bench_torchrun_n1.json
timeout 300 python bench.py --split-graph --steps 3 --warmup 1 2>/dev/null < /dev/null | tail -1 > $O/split_positions.json; cut -c1-200 $O/split_positions.json
timeout 300 python bench.py --split-graph --split-from-edges --steps 3 --warmup 1 2>/dev/null < /dev/null | tail -1 > $O/split_edges.json; cut -c1-200 $O/split_edges.json
bash scripts/gpu/profile.sh $TAG < /dev/null | grep "rc="
bash scripts/gpu/profile_bwd.sh $TAG < /dev/null | grep "rc=\|bwd M-edges\|reduced"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
MODES=auto timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/train121 -o run -- python scripts/time_deferred.py g121 6 3 > $O/train121_prof.log 2>&1 < /dev/null; echo "train121 trace rc=$?"
timeout 300 python scripts/time_deferred.py g121 6 3 > $O/train121.log 2>&1 < /dev/null; grep "E=" $O/train121.log | cut -c1-330
MODES=auto timeout 300 python scripts/time_deferred.py g241 6 3 > $O/train241.log 2>&1 < /dev/null; grep "E=" $O/train241.log | cut -c1-330
NODEATTR=1 MODES=auto timeout 300 python scripts/time_deferred.py g241 6 3 > $O/train241_nodeattr.log 2>&1 < /dev/null; grep "E=\|graph from" $O/train241_nodeattr.log | cut -c1-330
timeout 200 python scripts/time_mgkn_train.py all 5 > $O/mgkn_train.log 2>&1 < /dev/null; grep "train step" $O/mgkn_train.log | cut -c1-200
GPDE_EDGE_WEIGHT_CACHE=off timeout 200 python scripts/time_mgkn_train.py all 5 > $O/mgkn_train_we_off.log 2>&1 < /dev/null; grep "train step" $O/mgkn_train_we_off.log | cut -c1-120
bash scripts/gpu/refscripts.sh < /dev/null | grep "rc=\|passed\|skipped"
find $O gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_bwd gpurun_out/refscripts -type f -size +2M -delete

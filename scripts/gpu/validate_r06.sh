#!/bin/bash
# Round 6: one GPU-box pass over what the round's numbers rest on (after the full tier, which runs on its own):
#   smoke -> default bench line (+ bench_detail.json) -> --train under torch.distributed.run (N = 1, RCCL) -> rocprofv3 of the headline bench
#   (stats + FETCH / WRITE / busy PMC passes) -> the same of ONE backward, kept-H form and recompute form -> G241 depth-6 training steps with
#   kernel stats -> MGKN training steps with kernel stats.  Raw output under gpurun_out/; scripts/collect_profiles.py r06,
#   collect_profiles_bwd.py r06k --kept-h / r06 make the profiles/ files.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/validate_r06; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 < /dev/null | grep -v amdgpu.ids | tail -2
( time timeout 900 python bench.py --detail-out $O/bench_detail.json > $O/bench.json 2> $O/bench.err < /dev/null ); echo "bench rc=$?"
grep "\[bench\]" $O/bench.err | tail -12; wc -c $O/bench.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --train --steps 3 --warmup 1 2>/dev/null < /dev/null | tail -1 > $O/bench_train_torchrun_n1.json; cut -c1-400 $O/bench_train_torchrun_n1.json
bash scripts/gpu/profile.sh r06 < /dev/null | grep "rc="
bash scripts/gpu/profile_bwd.sh r06k < /dev/null | grep "rc=\|bwd M-edges\|reduced"
GPDE_SAVE_H_GB=0 bash scripts/gpu/profile_bwd.sh r06 < /dev/null | grep "rc=\|bwd M-edges\|reduced"
cd /tmp; cd $R
MODES=auto timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $O/train241 -o run -- python scripts/time_deferred.py g241 6 3 > $O/train241_prof.log 2>&1 < /dev/null; echo "train241 trace rc=$?"; grep "E=" $O/train241_prof.log | cut -c1-330
cp $(find $O/train241 -name "*kernel_stats.csv" | head -1) $O/train_g241_kernel_stats.csv 2>/dev/null
for wl in mgkn_orthogonal_burgers1d mgkn_general_darcy2d; do
  timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/mt_$wl -o run -- python scripts/time_mgkn_train.py $wl 10 > $O/mgkn_train_$wl.log 2>&1 < /dev/null; echo "$wl trace rc=$?"; grep "train step" $O/mgkn_train_$wl.log | cut -c1-200
  cp $(find $O/mt_$wl -name "*kernel_stats.csv" | head -1) $O/mgkn_${wl}_train_kernel_stats.csv 2>/dev/null
done
find $O gpurun_out/prof_r06 gpurun_out/prof_r06k_bwd gpurun_out/prof_r06_bwd -type f -size +2M -delete
du -sh $O

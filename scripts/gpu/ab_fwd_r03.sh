#!/bin/bash
# Same-box A/B of the headline forward: the round-3 tree (commit cb49ddf, extracted and built under the git-ignored
# scripts/ubench/r03tree/ by `git archive cb49ddf graph-pde_amd include bench.py graph_pde_amd.py oracle | tar -x -C ...`) against
# the current tree, alternating runs - VERDICT r4: "191.75 (r03) -> 185.54 (r04) ... nobody A/B'd the r03 .so".
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_r03; mkdir -p $O
F="--no-cpu-baseline --no-alt --no-mgkn --no-reuse-probe --no-backward-probe --steps 8 --warmup 3"
for rep in 1 2 3; do
  (cd $R/scripts/ubench/r03tree && timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r03tree', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])")
  (cd $R && timeout 300 python bench.py $F --no-measure-traffic 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('current', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])")
done | tee $O/ab.txt

#!/bin/bash
# Round-4 clock evidence for the "power-governed" reading of the headline kernel, all on ONE box, one call:
#   1. scripts/ubench/kloop_model_v6: bare MFMA stream / full K-loop model on random operands, cycles per chunk AND the shader
#      clock each variant ran at (clock64 cycles of the loop / event time), incl. "2 of 3 split products";
#   2. the production kernel and its 2-of-3 ablation build (libgpde_abl2.so: GPDE_ABL_2MFMA, wrong results, timing only):
#      bench line (3 steps), then rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES (kernel-trace only)
#      -> effective clock = GUI_ACTIVE / 8 XCDs / duration, matrix-pipe busy share.
# Output: gpurun_out/clock_r04/ ; scripts/collect_clock_evidence.py writes profiles/r04_clock_evidence.{txt,json}.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/clock_r04
mkdir -p $O
cd $R
timeout 120 scripts/ubench/kloop_model_v6 > $O/kloop_model_v6.txt 2>&1; echo "ubench rc=$?"
B="python $R/bench.py --no-cpu-baseline --no-reuse-probe --no-mgkn --no-alt --no-backward-probe"
for v in base abl2; do
  if [ $v = abl2 ]; then export GPDE_LIB=$R/scripts/ubench/lib/libgpde_abl2.so; export GPDE_ALLOW_ABLATION=1; else unset GPDE_LIB GPDE_ALLOW_ABLATION; fi
  timeout 200 $B --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_$v.json; echo "bench $v rc=$?"
  timeout 300 rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/pmc_$v -o run -- $B --steps 2 --warmup 1 > $O/pmc_$v.log 2>&1; echo "pmc $v rc=$?"
done
unset GPDE_LIB
python $R/scripts/collect_clock_evidence.py --reduce
find $O -type f -size +2M -delete
ls $O

#!/bin/bash
# rocprofv3 records of the headline bench (run on the GPU box through gpurun): kernel-trace stats, and one
# --pmc pass per counter set (never combined with sys/hip/hsa traces: gpurun refuses that).  Raw output goes to
# gpurun_out/prof_<tag>/; scripts/collect_profiles.py turns it into the committed profiles/ files.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
B="python $R/bench.py --no-cpu-baseline --no-reuse-probe --no-mgkn --no-alt --no-backward-probe"
mkdir -p $R/gpurun_out/prof_$TAG
cd $R
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/prof_$TAG/stats -o run -- $B --steps 3 --warmup 1 > gpurun_out/prof_$TAG/stats.log 2>&1; echo "stats rc=$?"
timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_$TAG/fetch -o run -- $B --steps 1 --warmup 0 > gpurun_out/prof_$TAG/fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof_$TAG/write -o run -- $B --steps 1 --warmup 0 > gpurun_out/prof_$TAG/write.log 2>&1; echo "write rc=$?"
timeout 300 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/prof_$TAG/busy -o run -- $B --steps 1 --warmup 0 > gpurun_out/prof_$TAG/busy.log 2>&1; echo "busy rc=$?"
grep "^{" gpurun_out/prof_$TAG/stats.log | head -1 > gpurun_out/prof_$TAG/bench_line.json
# only small summaries travel back (gpurun merges at most 64 MiB)
find gpurun_out/prof_$TAG -type f -size +2M -delete
find gpurun_out/prof_$TAG -name "*.csv" < /dev/null | head -30
du -sh gpurun_out/prof_$TAG < /dev/null

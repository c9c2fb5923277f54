#!/bin/bash
# rocprofv3 records of ONE NNConv backward on the s=121 graph (scripts/time_bwd.py g121: 3 forward + backward pairs, hidden
# cache off): kernel-trace stats, then one --pmc pass per counter set (FETCH_SIZE, WRITE_SIZE, matrix-pipe busy) - never
# combined with other trace domains.  Raw output: gpurun_out/prof_<tag>_bwd/; scripts/collect_profiles_bwd.py makes the
# committed profiles/ files.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
O=gpurun_out/prof_${TAG}_bwd
cd $R
mkdir -p $O
export GPDE_HIDDEN_CACHE=off
B="python $R/scripts/time_bwd.py g121"
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o run -- $B > $O/stats.log 2>&1; echo "stats rc=$?"
timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o run -- $B > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $O/write -o run -- $B > $O/write.log 2>&1; echo "write rc=$?"
timeout 300 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/busy -o run -- $B > $O/busy.log 2>&1; echo "busy rc=$?"
grep "bwd M-edges" $O/stats.log | tail -1
# the per-dispatch counter tables of a backward are a few MB: reduce them on the box, ship the summary
python $R/scripts/collect_profiles_bwd.py $TAG --reduce
find $O -type f -size +2M -delete
du -sh $O < /dev/null

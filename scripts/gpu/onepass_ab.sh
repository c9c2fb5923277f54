#!/bin/bash
# A/B of the backward's one-pass kernel against the two-pass form on ONE box: wall time (scripts/time_bwd.py g121) and the
# per-kernel table of rocprofv3 --kernel-trace --stats for both.  Raw output under gpurun_out/onepass_ab/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/onepass_ab
cd $R
mkdir -p $O
export GPDE_HIDDEN_CACHE=off
B="python $R/scripts/time_bwd.py ${1:-g121}"
for mode in one two; do
  if [ $mode = two ]; then export GPDE_BWD_TWO_PASS=1; else unset GPDE_BWD_TWO_PASS; fi
  timeout 200 $B 2>/dev/null | tail -2 > $O/wall_$mode.txt
  timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_$mode -o run -- $B > $O/stats_$mode.log 2>&1
  f=$(find $O/stats_$mode -name "*kernel_stats.csv" | head -1)
  python - "$f" > $O/kernels_$mode.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over the run (3 forward + backward pairs)")
for r in rows[:18]:
    print(f'{float(r["TotalDurationNs"])/3e6:9.2f} ms/pair  {int(r["Calls"]):6d} calls  {r["Name"][:90]}')
P
  echo "== $mode"; cat $O/wall_$mode.txt; cat $O/kernels_$mode.txt
done
unset GPDE_BWD_TWO_PASS
find $O -type f -size +2M -delete

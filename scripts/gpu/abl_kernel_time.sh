#!/bin/bash
# Kernel time of the backward's one-pass kernel (and wall time of the backward) for a list of developer builds under
# scripts/ubench/lib/ (ablations: GPDE_ALLOW_ABLATION=1): one box, rocprofv3 --kernel-trace --stats of scripts/time_bwd.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/abl
cd $R; mkdir -p $O
export GPDE_HIDDEN_CACHE=off GPDE_ALLOW_ABLATION=1
for lib in "$@"; do
  if [ $lib = prod ]; then unset GPDE_LIB; else export GPDE_LIB=$R/scripts/ubench/lib/libgpde_$lib.so; fi
  rm -rf $O/s_$lib
  timeout 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O/s_$lib -o run -- python $R/scripts/time_bwd.py ${CFG:-g121} > $O/log_$lib.txt 2>&1
  f=$(find $O/s_$lib -name "*kernel_stats.csv" | head -1)
  w=$(grep "bwd M-edges" $O/log_$lib.txt | tail -1 | sed 's/.*bwd \([0-9.]*\) ms.*/\1/')
  k=$(python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'f16v6_kernel<2' in r['Name']: print(round(float(r['AverageNs'])/1e6,2))
")
  echo "$lib: one-pass kernel $k ms, backward wall $w ms"
  rm -rf $O/s_$lib
done

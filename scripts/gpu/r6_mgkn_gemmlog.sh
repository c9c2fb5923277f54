cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r6_mgkn_gemm; mkdir -p $O
for w in mgkn_general_darcy2d mgkn_orthogonal_burgers1d; do
  rm -rf $O/$w
  GPDE_DEBUG_GEMM_LOG=1 timeout 300 rocprofv3 --output-format csv --kernel-trace -d $O/$w -o run -- python scripts/time_mgkn_train.py $w 1 > $O/$w.out 2> $O/$w.err
  grep "train step" $O/$w.out
  grep "^\[gpde_gemm\]" $O/$w.err > $O/$w.gemmlog
  python - "$O/$w" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
gem = [r for r in rows if "gpde_gemm_kernel" in r["Kernel_Name"]]
log = [l.split() for l in open(d + ".gemmlog")]
print(len(gem), "gemm dispatches,", len(log), "log lines")
n = len(log) // 4
agg = collections.OrderedDict()
for r, l in list(zip(gem, log))[-n:]:
    k = " ".join(l[1:])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"last step: {n} fp32 GEMM launches, {tot/1e3:.2f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{v[1]:9.1f} us  x{v[0]:3d}  {k}")
PY
  python scripts/mgkn_step_breakdown.py $(find $O/$w -name "*kernel_trace.csv")
  find $O/$w -type f -size +3M -delete
done

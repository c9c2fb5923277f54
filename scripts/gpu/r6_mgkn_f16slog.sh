cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r6_mgkn_gemm; mkdir -p $O
for w in mgkn_orthogonal_burgers1d mgkn_general_darcy2d; do
  rm -rf $O/$w
  GPDE_DEBUG_GEMM_LOG=1 timeout 300 rocprofv3 --output-format csv --kernel-trace -d $O/$w -o run -- python scripts/time_mgkn_train.py $w 1 > $O/$w.out 2> $O/$w.err
  grep "^\[gpde_gemm_f16s\]" $O/$w.err > $O/$w.f16slog
  python - "$O/$w" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
gem = [r for r in rows if "gpde_gemm_f16s_nt_kernel" in r["Kernel_Name"]]
log = [l.split() for l in open(d + ".f16slog")]
print(d, len(gem), "f16s dispatches,", len(log), "log lines")
n = len(log) // 4
tot = 0
for r, l in list(zip(gem, log))[-n:]:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += us
    M, N, K = int(l[2]), int(l[4]), int(l[6])
    print(f"{us:8.1f} us  {' '.join(l[1:])}   {6.0 * M * N * K / us / 1e6:7.1f} TF-f16/s  grid {r.get('Grid_Size_X', r.get('Grid_Size'))}")
print("total", tot)
PY
  find $O/$w -type f -size +3M -delete
done

"""Per-kernel breakdown of the LAST training step in a rocprofv3 kernel trace of scripts/time_mgkn_train.py <workload> 1 (3 warm-up steps + 1): usage mgkn_step_breakdown.py <kernel_trace.csv>"""
import csv, sys, collections, re
f = sys.argv[1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
gem = [r for r in rows if "gpde_gemm_kernel" in r["Kernel_Name"]]
n = len(gem) // 4
t_lo = int(gem[-n]["Start_Timestamp"])
# step boundary: find the Adam kernels (multi_tensor_apply) before
last = [r for r in rows if int(r["Start_Timestamp"]) >= t_lo]
def short(nm):
    nm = re.sub(r"\(anonymous namespace\)::", "", nm)
    nm = re.sub(r"^void ", "", nm)
    m = re.match(r"at::native::vectorized_elementwise_kernel<\d+, at::native::(\w+)", nm)
    if m: return "torch:" + m.group(1)
    return nm.split("(")[0][:70]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in last:
    a = agg[short(r["Kernel_Name"])]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("launches", len(last), "busy ms", sum(v[1] for v in agg.values()) / 1e3, "span ms", (int(last[-1]["End_Timestamp"]) - t_lo) / 1e6)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{v[1]:9.1f} us x{v[0]:4d}  {k}")
# add sizes
adds = [r for r in last if "CUDAFunctor_add" in r["Kernel_Name"]]
c = collections.Counter()
for r in adds:
    c[(int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]), )] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(sorted(c.items(), key=lambda kv: -kv[1])[:12])

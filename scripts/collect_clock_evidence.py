#!/usr/bin/env python3
"""gpurun_out/clock_r04/ (scripts/gpu/clock_evidence.sh) -> profiles/r04_clock_evidence.json + profiles/r04_kloop_model_clocks.txt.
`--reduce` runs on the GPU box: per-dispatch counter tables + kernel traces -> summary.json (small enough to ship back)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "gpurun_out", "clock_r04")
KERNEL = "gpde_fused_f16v6_kernel"


def reduce_one(tag):
    d = os.path.join(src, f"pmc_{tag}")
    cf = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kf = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not cf or not kf:
        return None
    dur = {}
    for r in csv.DictReader(open(kf[0])):
        if KERNEL in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    ctr = defaultdict(dict)
    for r in csv.DictReader(open(cf[0])):
        if KERNEL in r["Kernel_Name"]:
            ctr[r["Dispatch_Id"]][r["Counter_Name"]] = ctr[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = []
    for k, c in ctr.items():
        if k in dur and "GRBM_GUI_ACTIVE" in c:
            rows.append({"duration_ms": dur[k] / 1e6, "gui_active": c["GRBM_GUI_ACTIVE"], "mfma_busy": c.get("SQ_VALU_MFMA_BUSY_CYCLES"),
                         "sq_busy": c.get("SQ_BUSY_CYCLES")})
    if not rows:
        return None
    n = len(rows)
    t = sum(r["duration_ms"] for r in rows) / n
    ga = sum(r["gui_active"] for r in rows) / n
    mb = sum(r["mfma_busy"] or 0 for r in rows) / n
    return {"dispatches": n, "avg_duration_ms_under_pmc": round(t, 3), "GRBM_GUI_ACTIVE_sum_over_8_xcds": ga,
            "effective_clock_GHz": round(ga / 8 / (t * 1e-3) / 1e9, 3),
            "mfma_busy_share_of_simd_cycles": round(mb / (ga * 128), 4) if ga else None}


if "--reduce" in sys.argv:
    out = {}
    for tag in ("base", "abl2"):
        out[tag] = {"pmc": reduce_one(tag)}
        bl = os.path.join(src, f"bench_{tag}.json")
        if os.path.exists(bl) and os.path.getsize(bl):
            try:
                j = json.loads(open(bl).read())
                out[tag]["bench"] = {"M_edges_per_s": j["value"], "ms_per_step": j["ms_per_step"], "avg_launch_ms": j["roofline"]["avg_launch_ms"],
                                     "frac": j["roofline"]["frac"]}
            except Exception as ex:       # noqa: BLE001
                out[tag]["bench_error"] = repr(ex)
    json.dump(out, open(os.path.join(src, "summary.json"), "w"), indent=1)
    print(json.dumps(out))
    sys.exit(0)

summ = json.load(open(os.path.join(src, "summary.json")))
shutil.copyfile(os.path.join(src, "kloop_model_v6.txt"), os.path.join(REPO, "profiles", "r04_kloop_model_clocks.txt"))
json.dump({"source": "scripts/gpu/clock_evidence.sh: one gpurun call, one box.  `base` = the shipped gpde_fused_f16v6_kernel, `abl2` = the same "
                     "kernel built with -DGPDE_ABL_2MFMA (two of the three split products: a third less matrix work per edge; results WRONG, "
                     "timing only).  effective_clock_GHz = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration under the PMC pass; "
                     "mfma busy share as profiles/r03_pmc_busy_g241.json.  The bench figures are from separate, unprofiled runs of the same builds.",
           "g241": summ}, open(os.path.join(REPO, "profiles", "r04_clock_evidence.json"), "w"), indent=1)
print(json.dumps(summ, indent=1))

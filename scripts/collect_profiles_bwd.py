#!/usr/bin/env python3
"""rocprofv3 output of scripts/gpu/profile_bwd.sh (gpurun_out/prof_<tag>_bwd/) -> committed records of ONE NNConv backward on
the s=121 graph (N = 14,641, E = 5,931,137, kernel MLP 6-1024-1024-4096, hidden cache off):

    profiles/<tag>_bwd_g121_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (all kernels of 3 fwd + bwd pairs)
    profiles/traffic_<tag>_bwd.json            HBM-side bytes per backward, by kernel symbol (bench.py backward.roofline)
    profiles/<tag>_bwd_pmc_busy_g121.json      matrix-pipe busy share per kernel of the backward

`--reduce` (run on the GPU box): collapse the per-dispatch counter tables into <dir>/summary.json first, because the raw
tables exceed what gpurun ships back.  Counter conventions as scripts/collect_profiles.py: FETCH_SIZE / WRITE_SIZE in KiB,
gfx950 FETCH_SIZE counts 128-byte requests as 64 (corrected = 2 x raw)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(REPO, "gpurun_out", f"prof_{tag}_bwd")
out = os.path.join(REPO, "profiles")
PASSES = 3                         # forward + backward pairs of scripts/time_bwd.py
FWD_ONLY = ("gpde_fused_f16v6_kernel<false", "gpde_fused_f16v6_kernel<0", "gpde_gemm3_kernel", "gpde_epilogue_kernel", "k_block_bounds", "k_absmax_x",
            "k_attr_bound", "k_split_x")
if "--kept-h" in sys.argv:      # round 5: the training forward wrote H_2 (store kernel) and aggregated from it - those launches are the forward's
    FWD_ONLY += ("gpde_fused_f16v6_kernel<1", "gpde_zagg_kernel")


def sym(name):
    m = re.search(r"(gpde_\w+(?:<[^>]*>)?|k_\w+|pack_\w+)", name)
    return m.group(1) if m else None


def find(sub, pat):
    fs = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


def reduce_counters(sub):
    f = find(sub, "*counter_collection.csv")
    rows = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    if f:
        for r in csv.DictReader(open(f)):
            s = sym(r["Kernel_Name"])
            if s:
                c = rows[s][r["Counter_Name"]]
                c[0] += float(r["Counter_Value"])
                c[1] += 1
    return {k: {c: v for c, v in d.items()} for k, d in rows.items()}


if "--reduce" in sys.argv:
    summ = {sub: reduce_counters(sub) for sub in ("fetch", "write", "busy")}
    json.dump(summ, open(os.path.join(src, "summary.json"), "w"))
    print("reduced:", {k: len(v) for k, v in summ.items()})
    sys.exit(0)

summ = json.load(open(os.path.join(src, "summary.json")))
f = find("stats", "*kernel_stats.csv")
if f:
    keep = [r for r in csv.reader(open(f))]
    hdr, body = keep[0], [r for r in keep[1:] if "gpde" in r[0] or "k_" in r[0] or "pack_" in r[0]]
    with open(os.path.join(out, f"{tag}_bwd_g121_kernel_stats.csv"), "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(hdr)
        w.writerows(body)
    print("kernel stats:", len(body), "kernels")

kernels, total = {}, 0.0
for s in sorted(set(summ["fetch"]) | set(summ["write"])):
    if s.startswith(FWD_ONLY):
        continue
    fr = summ["fetch"].get(s, {}).get("FETCH_SIZE", [0.0, 0])
    wr = summ["write"].get(s, {}).get("WRITE_SIZE", [0.0, 0])
    f_b, w_b = fr[0] * 1024 / PASSES, wr[0] * 1024 / PASSES
    if f_b + w_b < 1e6:
        continue
    kernels[s] = {"launches_per_backward": round(max(fr[1], wr[1]) / PASSES, 1), "FETCH_SIZE_raw_bytes": round(f_b),
                  "FETCH_SIZE_corrected_bytes": round(2 * f_b), "WRITE_SIZE_bytes": round(w_b), "hbm_bytes": round(2 * f_b + w_b)}
    total += 2 * f_b + w_b
json.dump({
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) of `GPDE_HIDDEN_CACHE=off python "
              f"scripts/time_bwd.py g121` (3 forward + backward pairs; per-backward figures = sums / 3, forward-only kernels left "
              f"out); scripts/gpu/profile_bwd.sh {tag}",
    "graph": "g121 (N=14641, E=5931137)", "kernel_width": 1024,
    "unit_note": "KiB counters of the L2's memory-side requests (Infinity-Cache hits included); gfx950 FETCH_SIZE counts 128-B requests "
                 "as 64 B: corrected = 2 x raw (upper bound for 4-byte loads)",
    "hbm_bytes_per_backward": round(total), "kernels": kernels}, open(os.path.join(out, f"traffic_{tag}_bwd.json"), "w"), indent=1)
print(f"traffic per backward: {total / 1e9:.1f} GB;", {k: f"{v['hbm_bytes'] / 1e9:.2f}" for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]['hbm_bytes'])[:8]})

rec = {}
for s, c in summ["busy"].items():
    tot = {k: v[0] for k, v in c.items()}
    if tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0 or s.startswith(FWD_ONLY):
        continue
    rec[s] = {"launches": c["GRBM_GUI_ACTIVE"][1] if "GRBM_GUI_ACTIVE" in c else None,
              "mfma_busy_share_of_simd_cycles": round(tot["SQ_VALU_MFMA_BUSY_CYCLES"] / max(tot.get("GRBM_GUI_ACTIVE", 0) * 128, 1), 4),
              "wave_wait_any_share": round(tot.get("SQ_WAIT_ANY", 0) / max(tot.get("SQ_WAVE_CYCLES", 1), 1), 4),
              "gui_active_cycles_sum_over_xcds": tot.get("GRBM_GUI_ACTIVE")}
json.dump({"source": f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE "
                     f"--kernel-trace of the same command; scripts/gpu/profile_bwd.sh {tag}",
           "note": "share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE summed over the 8 XCDs x 128 SIMDs per XCD), as profiles/r03_pmc_busy_g241.json",
           "kernels": rec}, open(os.path.join(out, f"{tag}_bwd_pmc_busy_g121.json"), "w"), indent=1)
print("busy:", {k: v["mfma_busy_share_of_simd_cycles"] for k, v in rec.items()})

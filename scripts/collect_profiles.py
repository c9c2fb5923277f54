#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of scripts/gpu/profile.sh (gpurun_out/prof_<tag>/) into the committed records:

    profiles/<tag>_bench_g241_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary (gpde kernels)
    profiles/traffic_<tag>.json                    HBM-side traffic per launch, KEYED BY KERNEL SYMBOL
    profiles/<tag>_pmc_busy_g241.json              matrix-pipe busy / wave wait shares of the fused kernel

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced
reads (MI355X_MICROARCH.md, HBM section): `hbm_bytes_per_launch` = 2 x raw FETCH + WRITE (an upper bound where
part of the traffic is 4-byte loads); the raw numbers are kept next to it."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(REPO, "gpurun_out", f"prof_{tag}")
out = os.path.join(REPO, "profiles")


def sym(name):
    m = re.search(r"(gpde_\w+|k_\w+)", name)
    return m.group(1) if m else None


def find(sub, pat):
    fs = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


def counter_rows(sub):
    f = find(sub, "*counter_collection.csv")
    rows = defaultdict(lambda: defaultdict(list))     # kernel symbol -> counter -> [values per dispatch]
    if not f:
        return rows
    for r in csv.DictReader(open(f)):
        s = sym(r["Kernel_Name"])
        if s:
            rows[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return rows


# ---- kernel stats -------------------------------------------------------------------------------------
f = find("stats", "*kernel_stats.csv")
if f:
    keep = [r for r in csv.reader(open(f))]
    hdr, body = keep[0], [r for r in keep[1:] if "gpde" in r[0] or "k_" in r[0]]
    with open(os.path.join(out, f"{tag}_bench_g241_kernel_stats.csv"), "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(hdr)
        w.writerows(body)
    print("kernel stats:", len(body), "gpde kernels")

# ---- traffic ---------------------------------------------------------------------------------------------
fetch, write = counter_rows("fetch"), counter_rows("write")
kernels = {}
for s in sorted(set(fetch) | set(write)):
    fr = fetch.get(s, {}).get("FETCH_SIZE", [])
    wr = write.get(s, {}).get("WRITE_SIZE", [])
    if not fr and not wr:
        continue
    n = max(len(fr), len(wr), 1)
    f_b = sum(fr) * 1024 / max(len(fr), 1)
    w_b = sum(wr) * 1024 / max(len(wr), 1)
    kernels[s] = {
        "launches": n,
        "FETCH_SIZE_raw_bytes_per_launch": round(f_b),
        "FETCH_SIZE_corrected_bytes_per_launch": round(2 * f_b),
        "WRITE_SIZE_bytes_per_launch": round(w_b),
        "hbm_bytes_per_launch": round(2 * f_b + w_b),
    }
line = {}
bl = os.path.join(src, "bench_line.json")
if os.path.exists(bl) and os.path.getsize(bl):
    line = json.loads(open(bl).read())
if kernels:
    json.dump({
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) of `python bench.py --steps 1 "
                  f"--warmup 0 --no-cpu-baseline --no-reuse-probe --no-mgkn --no-alt`; scripts/gpu/profile.sh {tag}",
        "config": "g241", "kernel_width": 1024,
        "unit_note": "FETCH_SIZE / WRITE_SIZE are KiB counters of the L2's memory-side requests (Infinity-Cache hits included); "
                     "gfx950 FETCH_SIZE counts 128-B requests as 64 B: corrected = 2 x raw (upper bound for 4-byte loads)",
        "kernels": kernels,
    }, open(os.path.join(out, f"traffic_{tag}.json"), "w"), indent=1)
    print("traffic:", {k: f"{v['hbm_bytes_per_launch'] / 1e9:.2f} GB" for k, v in kernels.items()})

# ---- busy counters -----------------------------------------------------------------------------------------
busy = counter_rows("busy")
rec = {}
for s, c in busy.items():
    if not s.startswith("gpde_fused"):
        continue
    tot = {k: sum(v) for k, v in c.items()}
    wave = tot.get("SQ_WAVE_CYCLES", 0) * 4          # quad-cycles -> cycles
    rec[s] = {"launches": len(next(iter(c.values()))), "counters_sum": tot,
              # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 128 SIMDs per XCD (32 CUs x 4): share of SIMD-cycles with the
              # matrix pipe busy (same formula as profiles/r01_pmc_fused_f16v3d_g241.json: 0.515 for the 8-wave kernel)
              "mfma_busy_share_of_simd_cycles": round(tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(tot.get("GRBM_GUI_ACTIVE", 0) * 128, 1), 4)
              if "GRBM_GUI_ACTIVE" in tot else None,
              "effective_clock_GHz": None,
              "wave_wait_any_share": round(tot.get("SQ_WAIT_ANY", 0) / max(tot.get("SQ_WAVE_CYCLES", 1), 1), 4),
              "wave_wait_inst_share": round(tot.get("SQ_WAIT_INST_ANY", 0) / max(tot.get("SQ_WAVE_CYCLES", 1), 1), 4),
              "wave_cycles": wave}
if rec:
    json.dump({"source": f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY "
                         f"GRBM_GUI_ACTIVE --kernel-trace; scripts/gpu/profile.sh {tag}",
               "note": "SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X_MICROARCH.md)",
               "kernels": rec}, open(os.path.join(out, f"{tag}_pmc_busy_g241.json"), "w"), indent=1)
    print("busy:", {k: (v["mfma_busy_share_of_simd_cycles"], v["wave_wait_any_share"]) for k, v in rec.items()})
if line:
    json.dump(line, open(os.path.join(out, f"{tag}_bench_g241_profiled_run.json"), "w"), indent=1)

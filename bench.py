#!/usr/bin/env python3
"""Benchmark of the fused NNConv forward on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config g241|g121|g61|g16] [--kernel-width 1024]
    python bench.py --train [--gpus N] ...          # training-step mode (BASELINE config 5 shape, see below)
    python bench.py --split-graph [--gpus N] ...    # ONE graph split by destination rows over the ranks (strong scaling)

One "step" = one `conv(x, edge_index, edge_attr)` forward of the headline operator on one PDE sample: the
Darcy-241^2 r=0.10 lattice radius graph (N = 58,081 nodes, E = 95,539,625 edges, BASELINE.json configs[1])
with the kernel MLP DenseNet([6,1024,1024,4096]), width 64, aggr='mean', root + bias
(UAI1_full_resolution.py:21,56-59).  Inputs are synthetic of that shape, resident in HBM and
with the destination CSR already built (its build time goes to stderr) when the timed region starts.
N > 1: one process per GPU (torch.distributed / RCCL for the barrier only), each rank runs its own
independent sample: weak scaling, no data-path collective.  Rank 0 prints ONE JSON line.

`value` = whole-job M-edges/s over the K timed steps bracketed by barrier + synchronize (max over ranks);
`median_step_ms` / `value_at_median` come from per-step HIP events on the kernels' stream (SURVEY.md §8d asks
for the median).

Round 6: the printed line is <= 2.5 KB - the contract's keys, a compact `roofline` and `cpu_baseline`, and `summary` with every
other leg's headline figure (graph preparation, MGKN forwards / training steps, the backward in both forms with its roofline
fraction and traffic, the G241 depth-6 training step on the same sample / distinct samples with its peak memory).  EVERYTHING
measured - the full objects described below, per-kernel counters, notes - goes to `bench_detail.json` (`--detail-out`), whose path
the line carries as `detail`.  Objects (in the detail file; `roofline` / `cpu_baseline` compact on the line):
  roofline      of the dominant kernel (its symbol from gpde_nnconv_fwd_kernel).  The path is compute bound
                (SURVEY.md §8d), so `bound` = "mfma".  `achieved` = MFMA FLOPs the kernel EXECUTES per launch /
                average launch duration (HIP events inside libgpde.so on the kernel's own stream), `peak` = the
                dense peak of the pipe it runs on (f16 2.5 PFLOP/s, fp32 157.3 TFLOP/s), `frac` = achieved/peak.
                `algorithmic_ratio` = the reference formulation's FLOPs (10,506,304 per edge at 1024^2) at the
                same duration / the fp32 MFMA peak - above 1 because the kernel re-associates the last layer
                 and runs the hidden layer on the f16 pipe; NOT a roofline fraction.  `traffic` =
                HBM-side bytes per launch from the PMC record of THIS kernel symbol in profiles/traffic_r0N.json (newest),
                or measured in this run with --measure-traffic
                (null when the record is of another kernel), next to the algorithmic bytes per launch.
  alt_precision the exact-fp32 arithmetic (every contraction on fp32 MFMA) on the same inputs: median of >= 5
                steps, and the distance between the two outputs.
  cpu_baseline  the CPU oracle (plain-PyTorch restatement of the reference path, kind "port") timed on this
                host's cores on a bounded sample: all in-edges (>= 1 M) of a stratified subset of destination
                rows of the same graph; the same rows give `rel_l2_sample`.
  mgkn          BASELINE configs 3 and 4 (MGKN-orthogonal Burgers-1D s=8192, MGKN-general Darcy-2D L=5):
                ms per model forward, NNConv calls, M-edge-applications/s, max rel-L2 of the distinct NNConv
                applications vs the fp64 oracle on the same tensors, time share per kernel kind.
  depth_reuse   cross-depth reuse probe (s=61, depth 6; and the headline graph with the partial H the device-sized
                budget allows).

--train: one training step per rank and step = forward + native backward (gpde_nnconv_bwd) of a depth-`--depth`
KernelNN-shaped stack on the rank's own sample + ONE flat RCCL gradient all-reduce (parallel.allreduce_gradients)
+ Adam; reports samples/s, ms/step and the all-reduce share.  Default graph g61 (the reference's own training
resolution, UAI1_full_resolution.py:39-46).

--split-graph: the exchange step of SURVEY.md §8(e): every rank holds the same graph, keeps the in-edges of its block of
destination rows (balanced on in-edges), a step = its rows of one NNConv forward + the RCCL all-gather of the [rows, 64]
blocks (15 MB in total at G241); `value` = the graph's edges / step time, `scaling` = "strong".
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

CONFIGS = {
    # name: (s, r)   BASELINE.json configs[1] is g241; the others are quick-look sizes
    "g241": (241, 0.10),
    "g121": (121, 0.10),
    "g61": (61, 0.10),
    "g16": (16, 0.15),
}
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md
PEAK_F16_MFMA_TFLOPS = 2500.0      # dense, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
# What a bare v_mfma_f32_32x32x16_f16 stream sustains on this chip (scripts/ubench/kloop_model_v6.hip, "48 MFMA only" /
# full K-loop model rows of profiles/r03_kloop_model_power.txt: 1.64 - 1.76 PFLOP/s, the clock falling to ~1.6 GHz under a
# 100 % busy matrix pipe).  Reported NEXT TO `frac` (which stays against the nominal dense peak), never instead of it.
SUSTAINED_F16_MFMA_TFLOPS = 1700.0
TRAFFIC_FILE = next((f for f in (os.path.join(REPO, "profiles", f"traffic_r0{k}.json") for k in (5, 4, 3)) if os.path.exists(f)),
                    os.path.join(REPO, "profiles", "traffic_r05.json"))      # newest committed PMC record of the headline kernel


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_flops_per_edge(dims, w=64):
    """the work the reference performs per edge (nn_conv.py:274-275)."""
    f = 0
    for i in range(len(dims) - 1):
        f += 2 * dims[i] * dims[i + 1]
    return f + 2 * w * w + w


def executed_flops_per_edge(dims, kernel, agg_f16, w=64):
    """MFMA FLOPs the fused kernel issues per edge for a 3-Linear MLP, as (fp32-MFMA, f16-MFMA):
      gpde_fused_kernel        everything on fp32 MFMA; H1 with K padded to 8, regenerated per 128-column slice
      gpde_fused_f16v3_kernel  hidden layer = 3 f16 MFMAs per product; H1 = 2 f16 MFMAs (K = 16) per 32 k1 rows
                               and 64-column wave tile
      gpde_fused_f16v6_kernel  the same with a 128-column wave tile (half the H1 regeneration)
    the 64 x k2 outer product (aggregation) is 3 f16 MFMAs per product when `agg_f16`, else fp32 MFMA."""
    k1p = (dims[1] + 31) // 32 * 32
    k2p = (dims[2] + 127) // 128 * 128
    agg = 2 * w * k2p
    hidden = 2 * k1p * k2p
    if kernel == "gpde_fused_kernel":
        return (2 * 8 * k1p * (k2p // 128) + agg + hidden, 0)
    tile = 64 if kernel == "gpde_fused_f16v3_kernel" else 128
    h1 = 2 * 2 * 16 * k1p * (k2p // tile)
    if agg_f16:
        return (0, 3 * hidden + h1 + 3 * agg)
    return (agg, 3 * hidden + h1)


def traffic_record(config, kw, kernel):
    """HBM-side traffic of `kernel` from the committed PMC record, or (None, reason)."""
    if not os.path.exists(TRAFFIC_FILE):
        return None, f"no {os.path.relpath(TRAFFIC_FILE, REPO)}"
    try:
        tj = json.load(open(TRAFFIC_FILE))
    except Exception as ex:       # noqa: BLE001
        return None, f"unreadable traffic file: {ex}"
    if tj.get("config") != config or tj.get("kernel_width") != kw:
        return None, f"PMC record is for {tj.get('config')} / width {tj.get('kernel_width')}"
    rec = tj.get("kernels", {}).get(kernel)
    if rec is None:
        return None, f"PMC record holds {sorted(tj.get('kernels', {}))}, not {kernel}: re-run scripts/gpu/profile.sh"
    return rec, tj.get("source", "")


def measure_traffic(args, kernel):
    """--measure-traffic: HBM-side bytes per launch of `kernel`, measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE, then
    WRITE_SIZE: separate passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes) over a one-step child run of this
    script.  Same correction as scripts/collect_profiles.py: the counters are KiB, gfx950's FETCH_SIZE counts a 128-byte
    request as 64 (corrected = 2 x raw, an upper bound where part of the traffic is 4-byte loads).  Returns (record, source)
    or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--config", args.config,
             "--kernel-width", str(args.kernel_width), "--no-cpu-baseline", "--no-reuse-probe", "--no-mgkn", "--no-alt",
             "--no-backward-probe", "--no-measure-traffic"] + (["--precision", args.precision] if args.precision else [])
    got = {}
    tmp = tempfile.mkdtemp(prefix="gpde_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            try:       # a pass takes ~40 s (import torch + graph build + one step under the counter service); never let it hold the bench
                r = subprocess.run(["rocprofv3", "--output-format", "csv", "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "run", "--"] + child,
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {ctr} pass exceeded 240 s"
            fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not fs:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode}): {r.stderr[-300:]}"
            vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(fs[0]))
                    if row["Counter_Name"] == ctr and kernel in row["Kernel_Name"]]
            if not vals:
                return None, f"no {ctr} rows for {kernel}"
            got[ctr] = (sum(vals) * 1024 / len(vals), len(vals))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    f_b, w_b = got["FETCH_SIZE"][0], got["WRITE_SIZE"][0]
    rec = {"launches": got["FETCH_SIZE"][1], "FETCH_SIZE_raw_bytes_per_launch": round(f_b),
           "FETCH_SIZE_corrected_bytes_per_launch": round(2 * f_b), "WRITE_SIZE_bytes_per_launch": round(w_b),
           "hbm_bytes_per_launch": round(2 * f_b + w_b)}
    return rec, "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over a one-step child run of bench.py"


def make_conv(kw, dev, k0=6, seed=0):
    import graph_pde_amd as gp
    torch.manual_seed(seed)
    mlp = torch.nn.Sequential(torch.nn.Linear(k0, kw), torch.nn.ReLU(), torch.nn.Linear(kw, kw),
                              torch.nn.ReLU(), torch.nn.Linear(kw, 4096))
    return gp.NNConv_old(64, 64, mlp, aggr="mean").to(dev)          # weights: seed 0, replicated


def median_ms(fn, steps, warmup=1):
    """Median of per-call HIP-event times (events on torch's current stream = the kernels' stream)."""
    for _ in range(warmup):
        fn()
    evs = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in evs)


# ------------------------------------------------------------------------------------------------------
# MGKN configurations (BASELINE configs 3 and 4)
# ------------------------------------------------------------------------------------------------------
def mgkn_probe(dev, steps=10):
    from graph_pde_amd import _lib, mgkn_workloads, ops
    from oracle.nnconv_oracle import nnconv_forward, rel_l2
    out = {}
    for name, build in mgkn_workloads.WORKLOADS.items():
        wl = build(dev)
        wl.forward()                                              # CSR / pack caches warm
        ms = median_ms(wl.forward, steps, warmup=2)
        _lib.profile_begin()
        wl.forward()
        kinds = _lib.profile_end()
        tot = sum(v[0] for v in kinds.values()) or 1.0
        # parity of every distinct NNConv application of the forward against the fp64 oracle, same tensors
        worst = 0.0
        for conv, x, ei, ea in wl.pairs:
            with torch.no_grad():
                y = conv(x, ei, ea)
            lin = ops.mlp_linears(conv.nn)
            ref = nnconv_forward(x.cpu(), ei.cpu(), ea.cpu(), [l.weight.detach().cpu() for l in lin],
                                 [l.bias.detach().cpu() for l in lin],
                                 None if conv.root is None else conv.root.detach().cpu(),
                                 None if conv.bias is None else conv.bias.detach().cpu(), aggr=conv.aggr,
                                 dtype=torch.float64, chunk_edges=8192)
            worst = max(worst, rel_l2(y.cpu(), ref))
        top = max(kinds, key=lambda k: kinds[k][0])
        wl_f = build(dev, fused_glue=True)                        # opt-in: relu(x + conv) inside the last kernel
        y_plain, y_fused = wl.forward(), wl_f.forward()
        ms_fused = median_ms(wl_f.forward, steps, warmup=2)
        ms_grouped = grouped_diff = None
        if name == "mgkn_orthogonal_burgers1d":                   # the 13 independent convs of a sweep in one launch
            wl_g = build(dev, grouped=True)
            y_g = wl_g.forward()
            ms_grouped = median_ms(wl_g.forward, steps, warmup=2)
            grouped_diff = max(float((a_.double() - b_.double()).norm() / a_.double().norm().clamp_min(1e-30))
                               for a_, b_ in zip(y_fused, y_g))
        from graph_pde_amd import hidden_cache as hc_
        # opt-in: plain module calls served from cached per-edge weights where the graph qualifies
        we0 = hc_.WE_MODE
        hc_.WE_MODE = "auto"
        try:
            wl_w = build(dev, fused_glue=True)
            y_w = wl_w.forward()
            ms_we = median_ms(wl_w.forward, steps, warmup=3)
            y_w = wl_w.forward()
            we_diff = max(float((a_.double() - b_.double()).norm() / a_.double().norm().clamp_min(1e-30)) for a_, b_ in zip(y_fused, y_w))
            worst_we = 0.0
            for conv, x, ei, ea in wl_w.pairs:
                with torch.no_grad():
                    for _ in range(4):
                        y = conv(x, ei, ea)
                lin = ops.mlp_linears(conv.nn)
                ref = nnconv_forward(x.cpu(), ei.cpu(), ea.cpu(), [l.weight.detach().cpu() for l in lin],
                                     [l.bias.detach().cpu() for l in lin],
                                     None if conv.root is None else conv.root.detach().cpu(),
                                     None if conv.bias is None else conv.bias.detach().cpu(), aggr=conv.aggr,
                                     dtype=torch.float64, chunk_edges=8192)
                worst_we = max(worst_we, rel_l2(y.cpu(), ref))
        finally:
            hc_.WE_MODE = we0
        # the UNMODIFIED call sequence recorded once into a HIP graph and replayed (gp.capture, opt-in): same kernels, same bits
        import graph_pde_amd as gp_
        hc_.clear()
        wl_c = build(dev)
        for _ in range(3):
            y_direct = [t_.clone() for t_ in wl_c.forward()]
        ms_direct_c = median_ms(wl_c.forward, steps, warmup=1)
        cap_ = gp_.capture(wl_c.forward)
        ms_cap = median_ms(cap_, steps, warmup=2)
        cap_same = all(torch.equal(a_, b_) for a_, b_ in zip(cap_(), wl_c.forward()))
        del cap_
        # one optimisation step of the NNConv stack (forward with autograd + native backward of every call + Adam): the
        # scripts' inner loop (MGKN_general_darcy2d.py:260-282, MGKN_orthogonal_burgers1d.py:226-242), default policy
        # (hidden-activation cache `auto`: every module's H is built once per step and its MLP backward runs once)
        hc_.clear()
        wl_t = build(dev)
        for _ in range(3):
            loss_t = wl_t.train_step()
        torch.cuda.synchronize()
        tts = []
        for _ in range(5):
            tq = time.perf_counter()
            loss_t = wl_t.train_step()
            torch.cuda.synchronize()
            tts.append(time.perf_counter() - tq)
        train_ms = 1e3 * statistics.median(tts)
        train_stats = dict(hc_.stats)
        hc_.clear()
        # ... and the whole step recorded (Adam with its step count on the device)
        train_cap_ms = None
        try:
            wl_tc = build(dev, capturable=True)
            cap_t = gp_.capture(wl_tc.train_step, updates_parameters=True)
            tts = []
            for _ in range(6):
                tq = time.perf_counter()
                loss_c = cap_t()
                torch.cuda.synchronize()
                tts.append(time.perf_counter() - tq)
            if bool(torch.isfinite(loss_c)):
                train_cap_ms = round(1e3 * statistics.median(tts[1:]), 3)
            del cap_t, wl_tc
        except Exception as ex:       # noqa: BLE001
            log(f"[bench] {name}: captured training step failed: {type(ex).__name__}: {str(ex)[:200]}")
        hc_.clear()
        out[name] = {
            "workload": wl.description, "nnconv_calls": wl.calls, "edge_applications": wl.edge_applications,
            "ms_per_forward": round(ms, 3),
            "ms_per_forward_captured": round(ms_cap, 3),
            "captured": {"what": "the same unmodified module calls recorded once into a HIP graph and replayed (graph_pde_amd.capture, "
                                 "opt-in): no host issue per call", "bit_identical_to_direct_calls": bool(cap_same),
                         "ms_direct_same_run": round(ms_direct_c, 3)},
            "train_step_ms": round(train_ms, 3),
            "train_step_captured_ms": train_cap_ms,
            "train_step": {"ms": round(train_ms, 3), "loss_finite": bool(torch.isfinite(loss_t)),
                           "M_edge_applications_per_s": round(wl.edge_applications / train_ms / 1e3, 2),
                           "hidden_cache": {k: train_stats.get(k) for k in ("hits", "builds", "direct")},
                           "note": "median of 5 steps after 3 warm-ups: forward with autograd, squared-norm loss, native backward of "
                                   "every NNConv call (gpde_nnconv_bwd / _hidden + gpde_hidden_bwd once per module), Adam lr 1e-3 wd 5e-4; "
                                   "gradient parity of every distinct call at this size: tests/test_gpu_mgkn.py"},
            "ms_per_forward_fused_glue": round(ms_fused, 3),
            "ms_per_forward_grouped": None if ms_grouped is None else round(ms_grouped, 3),
            "grouped_rel_l2_vs_fused_glue": grouped_diff,
            "ms_per_forward_edge_weight_cache": round(ms_we, 3),
            "edge_weight_cache": {"opt_in": "GPDE_EDGE_WEIGHT_CACHE=auto (plain module calls) / nnconv_group (explicit)",
                                  "hits": hc_.stats["we_hits"], "builds": hc_.stats["we_builds"],
                                  "rel_l2_vs_default_path": we_diff, "max_rel_l2_vs_oracle": worst_we},
            "best_ms_per_forward": round(min(v for v in (ms, ms_fused, ms_we, ms_grouped) if v is not None), 3),
            "fused_glue_rel_l2_vs_unfused": max(float((a_.double() - b_.double()).norm() / a_.double().norm().clamp_min(1e-30))
                                                for a_, b_ in zip(y_plain, y_fused)),
            "M_edge_applications_per_s": round(wl.edge_applications / ms / 1e3, 2),
            "max_rel_l2_vs_oracle": worst,
            "kernel_time_share": {k: round(v[0] / tot, 3) for k, v in kinds.items() if v[1]},
            "kernel_launches": {k: v[1] for k, v in kinds.items() if v[1]},
            "gpu_kernel_ms_per_forward": round(tot, 3),
            "top_kernel_kind": top,
            "bound": "launch / latency (<= 131 k edges per call, SURVEY.md §8d)",
        }
    return out


# ------------------------------------------------------------------------------------------------------
# training-step mode
# ------------------------------------------------------------------------------------------------------
def train_mode(args, rank, world, dev, use_dist, barrier):
    import torch.distributed as dist
    from graph_pde_amd import hidden_cache, parallel, synth
    s, r = CONFIGS[args.config]
    conv = make_conv(args.kernel_width, dev)
    fc1 = torch.nn.Linear(6, 64).to(dev)
    fc2 = torch.nn.Linear(64, 1).to(dev)
    torch.manual_seed(1)
    for m in (fc1, fc2):
        m.reset_parameters()
    params = list(fc1.parameters()) + list(conv.parameters()) + list(fc2.parameters())
    if use_dist:
        for m in (fc1, conv, fc2):
            parallel.broadcast_parameters(m)
    opt = torch.optim.Adam(params, lr=1e-4, weight_decay=5e-4)          # UAI1_full_resolution.py:242
    ei, ea, n = synth.darcy_graph(s, r, device=dev, seed=rank)           # this rank's sample
    e = int(ei.shape[1])
    g = torch.Generator(device=dev).manual_seed(10 + rank)
    feat = torch.randn(n, 6, device=dev, generator=g)
    y = torch.randn(n, device=dev, generator=g)

    def step():
        opt.zero_grad(set_to_none=True)
        h = fc1(feat)
        for _ in range(args.depth):                                       # KernelNN.forward, UAI1:26-33
            h = torch.relu(conv(h, ei, ea))
        out = fc2(h).view(-1)
        loss = torch.norm(out - y, 1)                                      # UAI1:265
        loss.backward()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        n_red = parallel.allreduce_gradients(params, world=world if use_dist else 1)
        b.record()
        opt.step()
        return a, b, n_red

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    evs = [step() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    ar_ms = sum(a.elapsed_time(b) for a, b, _ in evs) / max(args.steps, 1)
    mine = {"rank": rank, "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 2), "allreduce_ms": round(ar_ms, 3),
            "peak_GiB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1)}
    per_rank = [mine]
    if use_dist:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank != 0:
        return None
    ms = 1e3 * elapsed / args.steps
    return {
        "metric": "training samples/s, GKN Darcy-2D (fwd + native bwd + RCCL grad all-reduce + Adam)",
        "value": round(world * 1e3 / ms, 4), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (hidden layer + aggregation: 2-term f16-split MFMA, f32 accumulate)",
        "data": "synthetic",
        "rccl_ranks": world if use_dist else 0,
        "config": {"workload": f"one training step per GPU and step: KernelNN stack fc1 + {args.depth} x relu(NNConv_old) + fc2 on "
                               f"the {s}x{s} r={r} radius graph N={n} E={e}; kernel MLP 6-{args.kernel_width}-"
                               f"{args.kernel_width}-4096; L1 loss; Adam lr 1e-4 wd 5e-4; one sample per GPU per step",
                   "graph": args.config, "depth": args.depth, "hidden_cache": hidden_cache.MODE,
                   "parallelism": f"dp{world} (independent samples, one flat gradient all-reduce)"},
        "M_edge_applications_per_s": round(world * args.depth * e / (ms * 1e-3) / 1e6, 2),
        "allreduce": {"elements": evs[-1][2], "ms_per_step": round(ar_ms, 3), "share_of_step": round(ar_ms / ms, 5),
                      "backend": "nccl (RCCL)" if use_dist else "none (single process)"},
        "per_rank": per_rank,
        "samples_per_step": world, "config5_note": f"BASELINE config 5 = 256 samples per optimizer step: {256 // max(world, 1)} such steps per rank "
                                                    "with gradients accumulated, ONE all-reduce (tests/test_parallel_gloo.py ws-8)",
    }


def split_graph_mode(args, rank, world, dev, use_dist, barrier):
    """`--split-graph`: ONE graph split by destination rows over the ranks, a step = one NNConv forward of the WHOLE graph:
    each rank its rows, then the all-gather of the row blocks (parallel.nnconv_rows, SURVEY.md §8e way 2).  Strong scaling:
    total work fixed.  Not the default line (that one is one sample per GPU).  The rank's block is built from the POSITIONS
    (parallel.partition_rows_by_position: count pass over all nodes, fill pass over its own destinations) - no rank holds
    the whole edge list; `--split-from-edges` keeps round 3's whole-graph-then-filter path for comparison."""
    import torch.distributed as dist
    from graph_pde_amd import ops, parallel, synth
    s, r = CONFIGS[args.config]
    conv = make_conv(args.kernel_width, dev)
    n = s * s
    x = torch.randn(n, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(1000))
    torch.cuda.synchronize()
    tq = time.perf_counter()
    if args.split_from_edges:
        ei, ea, n = synth.darcy_graph(s, r, device=dev, seed=0)
        part = parallel.partition_rows(ei, ea, n, rank=rank, world=world)
        del ei, ea
    else:
        pos = synth.lattice_positions(s, dev)
        a_n = synth.darcy_coefficient(s, 0).to(dev)
        part = parallel.partition_rows_by_position(pos, r, ops.NodeAttr.darcy(pos, a_n), rank=rank, world=world)
    torch.cuda.synchronize()
    build_ms = 1e3 * (time.perf_counter() - tq)
    torch.cuda.empty_cache()
    graph = part.edge_index if part.csr is None else part.csr
    ev = []
    # a step is ONE NNConv forward, as in the default line (which calls the operator directly): the cross-call caches - the
    # same x / graph / weights every step would be served from kept hidden activations - stay out of the timed region
    from graph_pde_amd import hidden_cache
    hidden_cache.MODE = "off"

    def step():
        with torch.no_grad():
            a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            a.record()
            local = conv(x, graph, part.edge_attr)[part.lo:part.hi].contiguous()       # this rank's rows
            b.record()
            if use_dist:                              # (one rank under torchrun: still through the collective = RCCL)
                full = parallel._gather_blocks(local, part)
            else:
                full = local
            c.record()
            ev.append((a, b, c))
            return full

    for _ in range(args.warmup):
        step()
    ev.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    block_ms = statistics.median(a.elapsed_time(b) for a, b, _ in ev)
    gather_ms = statistics.median(b.elapsed_time(c) for _, b, c in ev)
    mine = {"rank": rank, "rows": [part.lo, part.hi], "edges": part.n_edges, "block_ms": round(block_ms, 3),
            "gather_ms": round(gather_ms, 3), "graph_build_ms": round(build_ms, 1)}
    per_rank = [None] * world
    if use_dist:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        dist.all_gather_object(per_rank, mine)
    else:
        per_rank = [mine]
    if rank != 0:
        return None
    e = sum(p_["edges"] for p_ in per_rank)
    ms = 1e3 * elapsed / args.steps
    return {
        "metric": "M-edges/s through fused NNConv fwd (width=64)", "value": round(e / (ms * 1e-3) / 1e6, 3), "unit": "M-edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None,
        "dtype": "f32 (hidden layer + aggregation: 2-term f16-split MFMA, f32 accumulate)", "data": "synthetic",
        "rccl_ranks": world if use_dist else 0,
        "config": {"workload": f"one NNConv forward of ONE {s}x{s} r={r} radius graph N={n} E={e} split by destination rows over "
                               f"{world} rank(s); kernel MLP 6-{args.kernel_width}-{args.kernel_width}-4096; all-gather of the "
                               f"[rows x 64] blocks inside the timed step", "graph": args.config,
                   "parallelism": f"rows{world} (destination-row blocks balanced on in-edges; x replicated)",
                   "partition": "whole edge list on every rank, then filtered (--split-from-edges)" if args.split_from_edges else
                                "built per rank from the positions: in-degree count pass over all nodes + fill pass over the rank's "
                                "own destinations (parallel.partition_rows_by_position); no rank holds the whole edge list"},
        "per_rank": per_rank, "row_bounds": part.bounds, "all_finite": bool(torch.isfinite(out).all()),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS))
    ap.add_argument("--kernel-width", type=int, default=1024)
    ap.add_argument("--cpu-rows", type=int, default=640, help="destination rows of the CPU sample (>= 1 M edges on g241)")
    ap.add_argument("--cpu-threads", type=int, default=64, help="threads of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reuse-probe", action="store_true",
                    help="skip the depth x module probe of the cross-depth hidden-activation reuse")
    ap.add_argument("--no-backward-probe", action="store_true", help="skip the single-call backward timing (g121)")
    ap.add_argument("--no-mgkn", action="store_true", help="skip the MGKN configurations (BASELINE configs 3, 4)")
    ap.add_argument("--no-g241-train", action="store_true", help="skip the G241 backward / depth-6 training-step figures")
    ap.add_argument("--no-alt", action="store_true", help="skip the exact-fp32 leg")
    ap.add_argument("--measure-traffic", action="store_true",
                    help="re-derive roofline.traffic in THIS run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of a one-step "
                         "child run of this script.  Since round 5 this is the DEFAULT for the headline configuration on one GPU "
                         "when rocprofv3 is on the box (adds ~1-2 minutes); --no-measure-traffic keeps the committed record")
    ap.add_argument("--no-measure-traffic", action="store_true", help="roofline.traffic from the committed profiles/traffic_r*.json")
    ap.add_argument("--precision", default=None, choices=["f32", "f16split", "f16split_8wave", "f16split_static", "f16split_agg16", "f16split_agg32", "f16split_noedge"],
                    help="arithmetic of the hidden layer (default: graph_pde_amd.ops.DEFAULT_PRECISION)")
    ap.add_argument("--detail-out", default=None, help="where the full record goes (default: ./bench_detail.json); the JSON line stays <= 2.5 KB")
    ap.add_argument("--train", action="store_true", help="training-step mode (see the module docstring)")
    ap.add_argument("--depth", type=int, default=6, help="--train: NNConv applications per forward")
    ap.add_argument("--split-from-edges", action="store_true",
                    help="--split-graph: build the whole edge list on every rank and filter it (round 3's path) instead of building "
                         "each rank's block from the positions")
    ap.add_argument("--split-graph", action="store_true",
                    help="strong-scaling mode: ONE graph split by destination rows over the ranks (all-gather per forward)")
    args = ap.parse_args()
    if args.config is None:
        args.config = "g61" if args.train else "g241"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # started as plain `python bench.py --gpus N`: re-exec under torch.distributed.run, one rank per GPU over RCCL
        # (the same command line the driver's launcher uses)
        have = torch.cuda.device_count()
        if have < args.gpus:
            log(f"--gpus {args.gpus}: only {have} GPU(s) visible on this node")
            sys.exit(2)
        import socket
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    if args.gpus != world:
        log(f"--gpus {args.gpus} but WORLD_SIZE={world}: the two must agree")
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torchrun (RANK set) the process group is always created, also for one rank, so the
    # N = 1 launch exercises the same RCCL init / barrier / all-reduce path as N = 2, 4, 8
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.train or args.split_graph:
        line = (train_mode if args.train else split_graph_mode)(args, rank, world, dev, use_dist, barrier)
        if line is not None:
            print(json.dumps(line), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return

    from graph_pde_amd import _lib, ops, synth

    s, r = CONFIGS[args.config]
    kw = args.kernel_width
    dims = [6, kw, kw, 4096]
    conv = make_conv(kw, dev)

    t0 = time.time()
    ei, ea, n = synth.darcy_graph(s, r, device=dev, seed=rank)       # independent sample per rank
    e = int(ei.shape[1])
    gx = torch.Generator(device=dev).manual_seed(1000 + rank)
    x = torch.randn(n, 64, device=dev, generator=gx)
    torch.cuda.synchronize()
    t1 = time.time()
    csr = ops.csr_for(ei, n)
    torch.cuda.synchronize()
    t2 = time.time()
    ops.attr_in_slot_order(csr, ea)              # edge_attr rows in CSR slot order (gpde_gather_rows): once per (graph, edge_attr)
    torch.cuda.synchronize()
    t3 = time.time()
    # ... and for a SECOND edge_attr tensor on the same graph (a new sample of the same mesh): the gather alone.  The first call above also
    # pays the process's first launches of these kernels and the identity probe of `perm` (22 - 108 ms across boxes); the cache entry of the
    # probe tensor is dropped again (2.3 GB)
    ea2 = ea.clone()
    torch.cuda.synchronize()
    t4 = time.time()
    ops.attr_in_slot_order(csr, ea2)
    torch.cuda.synchronize()
    t5 = time.time()
    if getattr(csr, "_attr_sorted", None):
        for k_ in [k_ for k_, v_ in csr._attr_sorted.items() if v_[0] is ea2]:
            csr._attr_sorted.pop(k_)
    del ea2
    # per-(graph, edge_attr) preparation every NEW sample pays before its first forward; never inside the timed region
    graph_prep = {"csr_build_ms": round(1e3 * (t2 - t1), 2), "attr_reorder_ms": round(1e3 * (t3 - t2), 2),
                  "attr_reorder_next_sample_ms": round(1e3 * (t5 - t4), 2),
                  "note": "gpde_csr_from_coo (stable sort by destination of the int64 [2,E] list) and gpde_gather_rows (edge_attr into "
                          "CSR slot order); both cached per tensor + version, `depth` applications and every epoch reuse them"}
    if rank == 0:
        log(f"[bench] graph {args.config}: N={n} E={e} generated in {t1 - t0:.2f}s; "
            f"dst-CSR build {1e3 * (t2 - t1):.1f} ms, attribute reorder {1e3 * (t3 - t2):.1f} ms (first call; next sample {1e3 * (t5 - t4):.1f} ms; not in the timed region)")

    lin = ops.mlp_linears(conv.nn)
    pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
    ws = torch.empty(ops.workspace_bytes(n, e, pm), dtype=torch.uint8, device=dev)
    out = torch.empty(n, 64, dtype=torch.float32, device=dev)
    plan = ops.launch_plan(n, e, pm, ws.numel())
    precision = args.precision or ops.DEFAULT_PRECISION
    kernel = ops.fused_kernel_name(n, e, pm, precision)

    def step(prec=precision):
        ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", out=out, ws=ws, precision=prec)

    for _ in range(args.warmup):
        step()
    barrier()
    _lib.profile_begin()
    evs = []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        step()
        b.record()
        evs.append((a, b))
    barrier()
    elapsed = time.perf_counter() - t_start
    kinds = _lib.profile_end()
    step_ms = [a.elapsed_time(b) for a, b in evs]
    if use_dist:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank != 0:
        dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * e / (elapsed / args.steps) / 1e6                  # M-edges/s, whole job
    med = statistics.median(step_ms)

    # ---- roofline of the dominant kernel (rank 0's launches) --------------------------------------
    fused_ms, launches = kinds["fused"]
    n_launch = max(int(launches), 1)
    avg_launch_ms = fused_ms / n_launch
    edges_per_launch = e * args.steps / n_launch
    agg_f16 = kernel != "gpde_fused_kernel" and precision != "f16split_agg32" and (e >= 32768 or precision == "f16split_agg16")
    f_alg = algorithmic_flops_per_edge(dims)
    f_exe32, f_exe16 = executed_flops_per_edge(dims, kernel, agg_f16)
    rate = edges_per_launch / (avg_launch_ms * 1e-3) / 1e12           # 1e12 edges/s through the kernel
    on_f16 = f_exe16 > 0
    achieved = (f_exe16 if on_f16 else f_exe32) * rate
    peak = PEAK_F16_MFMA_TFLOPS if on_f16 else PEAK_FP32_MFMA_TFLOPS
    n_param = sum(p.numel() for p in conv.parameters())
    alg_bytes_per_launch = (40.0 * e + 512.0 * n + 4.0 * n_param) * edges_per_launch / e   # SURVEY §8(d) x units per launch
    rec, src = traffic_record(args.config, kw, kernel)
    import shutil as _sh
    under_profiler = any(k_.startswith(("ROCPROF", "ROCP_")) for k_ in os.environ)        # (this run is itself a rocprofv3 child)
    if not args.measure_traffic and not args.no_measure_traffic and world == 1 and args.config == "g241" and _sh.which("rocprofv3") and \
            not under_profiler:
        args.measure_traffic = True
    if args.measure_traffic and world == 1:
        del ws
        torch.cuda.empty_cache()                      # the child run needs the device's memory
        rec_m, src_m = measure_traffic(args, kernel)
        ws = torch.empty(ops.workspace_bytes(n, e, pm), dtype=torch.uint8, device=dev)
        if rec_m is not None:
            rec, src = rec_m, src_m
        else:
            log(f"[bench] --measure-traffic: {src_m}; keeping the committed record")
            src = f"{src} (--measure-traffic failed: {src_m})" if rec is not None else src_m
    traffic = None if rec is None else rec.get("hbm_bytes_per_launch")
    other_ms = sum(v[0] for k, v in kinds.items() if k != "fused")
    edges_per_s_rank = e / (elapsed / args.steps)
    bytes_per_edge = (40.0 * e + 512.0 * n + 4.0 * n_param) / e
    roofline = {
        "kernel": kernel, "bound": "mfma", "pipe": "f16 MFMA (2-term split operands, fp32 accumulate)" if on_f16 else "fp32 MFMA",
        "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
        "sustained_mfma_tflops": SUSTAINED_F16_MFMA_TFLOPS if on_f16 else None,
        "frac_of_sustained": round(achieved / SUSTAINED_F16_MFMA_TFLOPS, 4) if on_f16 else None,
        "sustained_note": "power-governed chip: a bare f16 MFMA stream measures 1.64-1.76 PFLOP/s here (profiles/r03_kloop_model_power.txt); "
                          "removing 3.2 % of this kernel's cycles changed its time by 0.0 % (profiles/r03_fwd_sideload_interleave_ab.txt)"
                          if on_f16 else None,
        "traffic": traffic, "traffic_source": src if rec is not None else None,
        "traffic_note": None if rec is not None else src,
        "traffic_detail": rec,
        "algorithmic_bytes_per_launch": round(alg_bytes_per_launch),
        "traffic_over_algorithmic": None if traffic is None else round(traffic / alg_bytes_per_launch, 2),
        "avg_launch_ms": round(avg_launch_ms, 3), "launches_per_step": n_launch // args.steps,
        "executed_flop_per_edge": {"fp32_mfma": f_exe32, "f16_mfma": f_exe16},
        "executed_fp32_mfma_tflops": round(f_exe32 * rate, 2),
        "algorithmic_flop_per_edge": f_alg,
        "algorithmic_tflops": round(f_alg * rate, 2),
        "algorithmic_ratio": round(f_alg * rate / PEAK_FP32_MFMA_TFLOPS, 4),
        "algorithmic_ratio_note": "reference-formulation FLOPs / fp32 MFMA peak; > 1 because the last layer is re-associated "
                                  "(DESIGN.md §2) and the hidden layer runs on the f16 pipe - not a roofline fraction",
        "fused_share_of_step": round(fused_ms / (1e3 * elapsed), 4),
        "node_kernels_ms_per_step": round(other_ms / args.steps, 3),
        "kernel_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in kinds.items() if v[1]},
        "hbm_algorithmic_bytes_per_edge": round(bytes_per_edge, 2),
        "hbm_achieved_GBs": round(edges_per_s_rank * bytes_per_edge / 1e9, 2),
        "hbm_frac": round(edges_per_s_rank * bytes_per_edge / 1e9 / PEAK_HBM_GBS, 6),
        "hbm_note": "BASELINE.json's HBM target is not the binding bound: >= 205 FLOP/B against a machine balance of 19.7",
    }

    # ---- row f3 on the headline kernel: attributes read from the node table (no [E, 6] tensor, no perm) -----------
    nodeattr = None
    if world == 1 and not args.no_alt and kernel == "gpde_fused_f16v6_kernel":
        try:
            pos_n = synth.lattice_positions(s, dev)
            a_n = synth.darcy_coefficient(s, rank).to(dev)
            na = ops.NodeAttr.darcy(pos_n, a_n)
            same_attr = bool(torch.equal(na.materialize(ei[:, :4096]), ea[:4096]))
            out_n = torch.empty_like(out)
            fn_n = lambda: ops.nnconv_forward_nodeattr_raw(x, csr, na, pm, conv.root, conv.bias, "mean", out=out_n, ws=ws, precision=precision)
            ms_n = median_ms(fn_n, 3, warmup=1)
            step()
            nodeattr = {"ms_per_forward": round(ms_n, 2), "M_edges_per_s": round(e / ms_n / 1e3, 2),
                        "bitwise_equal_to_tensor_path": bool(torch.equal(out_n, out)) if same_attr else None,
                        "attr_tensor_bytes_not_read": int(ea.numel() * 4 + e * 4),
                        "note": "node_attr of gpde_nnconv_fwd_mixed_keepz on gpde_fused_f16v6_kernel<false, NODEATTR>: slot d of edge (j -> i) from "
                                "node_table[(j or i)][col] (SquareMeshGenerator.attributes recipe, utilities.py:274-277)"}
            log(f"[bench] node-table attributes: {nodeattr['ms_per_forward']} ms, {nodeattr['M_edges_per_s']} M-edges/s, "
                f"bitwise equal: {nodeattr['bitwise_equal_to_tensor_path']}")
        except Exception as ex:        # recorded, not fatal for the headline line
            nodeattr = {"error": repr(ex)}

    # ---- the exact-fp32 arithmetic on the same inputs: median of >= 5 steps -------------------------
    alt = None
    if world == 1 and not args.no_alt:
        other = "f32" if precision != "f32" else "f16split"
        out_main = out.clone()
        alt_steps = max(5, min(args.steps, 10))
        alt_ms = median_ms(lambda: step(other), alt_steps, warmup=1)
        d = (out.double() - out_main.double()).norm() / out.double().norm()
        alt = {"precision": other, "kernel": ops.fused_kernel_name(n, e, pm, other),
               "value": round(e / alt_ms / 1e3, 3), "unit": "M-edges/s", "median_step_ms": round(alt_ms, 3),
               "steps": alt_steps, "rel_l2_between_precisions": float(d)}
        out.copy_(out_main)

    # ---- CPU baseline + parity on a bounded sample --------------------------------------------------
    cpu = None
    rel = None
    if not args.no_cpu_baseline and world == 1:
        from oracle.nnconv_oracle import nnconv_forward, rel_l2
        # plain PyTorch CPU GEMMs stop scaling (and then regress) far below the 256 hardware
        # threads of the GPU box; `cores` reports the threads actually used
        ncores = min(os.cpu_count() or 1, args.cpu_threads)
        torch.set_num_threads(ncores)
        rows = torch.linspace(0, n - 1, min(args.cpu_rows, n)).round().long().unique()
        rowptr = csr.rowptr.cpu().long()
        slots = torch.cat([torch.arange(int(rowptr[i]), int(rowptr[i + 1])) for i in rows.tolist()])
        eid = csr.perm.cpu().long()[slots]
        eid, _ = torch.sort(eid)                                      # reference (input) edge order
        ei_s, ea_s = ei[:, eid.to(dev)].cpu(), ea[eid.to(dev)].cpu()
        x_c = x.cpu()
        ws_c = [l.weight.detach().cpu() for l in lin]
        bs_c = [l.bias.detach().cpu() for l in lin]
        root_c, bias_c = conv.root.detach().cpu(), conv.bias.detach().cpu()
        # warm-up on a slice, then one timed pass over the whole sample
        nnconv_forward(x_c, ei_s[:, :65536], ea_s[:65536], ws_c, bs_c, root_c, bias_c, aggr="mean",
                       chunk_edges=65536)
        tc = time.perf_counter()
        y_cpu = nnconv_forward(x_c, ei_s, ea_s, ws_c, bs_c, root_c, bias_c, aggr="mean",
                               dtype=torch.float32, chunk_edges=65536)
        tcpu = time.perf_counter() - tc
        es = int(ei_s.shape[1])
        cpu = {"value": round(es / tcpu / 1e6, 5), "unit": "M-edges/s", "cores": ncores,
               "kind": "port",
               "sample": f"all {es} in-edges of {len(rows)} stratified destination rows of the same "
                         f"graph, plain-PyTorch fp32 oracle, {ncores} threads, {tcpu:.1f}s"}
        rel = rel_l2(out[rows.to(dev)].cpu(), y_cpu[rows])
        log(f"[bench] CPU oracle sample: {es} edges in {tcpu:.2f}s; rel-L2 GPU vs CPU rows = {rel:.3e}")

    del ws
    torch.cuda.empty_cache()

    # ---- MGKN configurations -----------------------------------------------------------------------------
    mgkn = None
    if not args.no_mgkn and world == 1:
        mgkn = mgkn_probe(dev)

    # ---- cross-depth reuse: depth applications of ONE module, as KernelNN.forward does
    #      (UAI1_full_resolution.py:29-30), on the reference's own training resolution s=61 r=0.10 -- the
    #      headline graph's hidden activations (E x 4 KiB = 391 GB) do not fit one GPU.  Not part of `value`.
    reuse = None
    if not args.no_reuse_probe and world == 1:
        from graph_pde_amd import hidden_cache
        ei6, ea6, n6 = synth.darcy_graph(61, 0.1, device=dev, seed=0)
        x6 = torch.randn(n6, 64, device=dev)
        depth = 6

        def model_fwd(xin):
            hcur = xin
            for _ in range(depth):
                hcur = torch.relu(conv(hcur, ei6, ea6))
            return hcur

        def tm(fn, reps=5):
            fn(); fn()
            torch.cuda.synchronize()
            tq = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - tq) / reps

        def fwd_only():
            with torch.no_grad():
                return model_fwd(x6)

        def fwd_bwd():
            conv.zero_grad(set_to_none=True)
            with torch.no_grad():
                for p_ in conv.parameters():
                    p_.mul_(1.0)                        # new weight version: as after an optimiser step
            model_fwd(x6).square().mean().backward()
        res = {}
        mode0 = hidden_cache.MODE
        for mode in ("off", "auto"):
            hidden_cache.MODE = mode
            hidden_cache.clear()
            y6 = fwd_only()
            res[mode] = (tm(fwd_only), tm(fwd_bwd, 3), y6)
        hidden_cache.MODE = mode0
        d6 = float((res["off"][2].double() - res["auto"][2].double()).norm() / res["off"][2].double().norm())
        e6 = int(ei6.shape[1])
        reuse = {"graph": "g61 (N=%d, E=%d)" % (n6, e6), "depth": depth,
                 "forward_ms": {"direct": round(1e3 * res["off"][0], 3), "reuse": round(1e3 * res["auto"][0], 3)},
                 "forward_backward_ms": {"direct": round(1e3 * res["off"][1], 3), "reuse": round(1e3 * res["auto"][1], 3)},
                 "forward_M_edge_applications_per_s": {"direct": round(depth * e6 / res["off"][0] / 1e6, 1),
                                                       "reuse": round(depth * e6 / res["auto"][0] / 1e6, 1)},
                 "rel_l2_between_paths": d6,
                 "note": "forward: fixed weights, H built once and reused by all later calls; "
                         "forward_backward: new weight version every step, H rebuilt once per step"}
        if args.config == "g241":
            # the headline graph itself: H (391 GB) does not fit, the default budget (hidden_cache.budget_bytes: sized to
            # the device) keeps the in-edges of the leading ~44 % of the nodes; the others are recomputed per layer
            def g_fwd():
                with torch.no_grad():
                    hcur = x
                    for _ in range(depth):
                        hcur = torch.relu(conv(hcur, ei, ea))
                    return hcur
            gres = {}
            for mode in ("off", "auto"):
                hidden_cache.MODE = mode
                hidden_cache.clear()
                torch.cuda.empty_cache()
                budget = hidden_cache.budget_bytes(dev)
                g_fwd()
                torch.cuda.synchronize()
                tq = time.perf_counter()
                yg = g_fwd()
                torch.cuda.synchronize()
                ent = hidden_cache._entries.get(conv)
                gres[mode] = (time.perf_counter() - tq, yg, budget, 0 if ent is None or ent.hidden is None else ent.hn)
            hidden_cache.MODE = mode0
            dg = float((gres["off"][1].double() - gres["auto"][1].double()).norm() / gres["off"][1].double().norm())
            reuse["g241_depth6_forward"] = {
                "direct_ms": round(1e3 * gres["off"][0], 1), "reuse_ms": round(1e3 * gres["auto"][0], 1),
                "M_edge_applications_per_s": {"direct": round(depth * e / gres["off"][0] / 1e6, 1),
                                              "reuse": round(depth * e / gres["auto"][0] / 1e6, 1)},
                "budget_GiB": round(gres["auto"][2] / 2 ** 30, 1), "nodes_served_from_H": gres["auto"][3], "nodes": n,
                "rel_l2_between_paths": dg,
                "note": "partial H: inference only (gpde_nnconv_fwd_mixed_keepz); budget = 70 % of HBM, at most free - 48 GB"}
            del gres, yg, ent                            # `ent` holds the 170 GB partial H: it must not outlive this probe
            hidden_cache.clear()
            torch.cuda.empty_cache()

    # ---- backward of ONE NNConv call on the s=121 graph: what `loss.backward()` costs per
    #      edge through the same operator (dx, dW_1..3, db_1..3, droot, dbias), hidden-activation cache off so that the
    #      recompute is inside the time.  Not part of `value`.
    backward = None
    if not args.no_backward_probe and world == 1:
        from graph_pde_amd import hidden_cache
        mode0 = hidden_cache.MODE
        hidden_cache.MODE = "off"
        hidden_cache.clear()
        try:
            eib, eab, nb_ = synth.darcy_graph(121, 0.1, device=dev, seed=0)
            xb = torch.randn(nb_, 64, device=dev, requires_grad=True)

            def fwd_bwd_pairs(reps):
                """[(training-forward s, backward s)] of `reps` forward + backward pairs of one NNConv call"""
                out_ = []
                for it in range(reps):
                    conv.zero_grad(set_to_none=True)
                    xb.grad = None
                    torch.cuda.synchronize()
                    tq0 = time.perf_counter()
                    yb = conv(xb, eib, eab)
                    lossb = yb.square().mean()
                    torch.cuda.synchronize()
                    tq = time.perf_counter()
                    lossb.backward()
                    torch.cuda.synchronize()
                    out_.append((tq - tq0, time.perf_counter() - tq))
                return out_
            # the DEFAULT policy (round 5): when 4 KiB per edge fit, the training forward KEEPS the last hidden activations (store
            # kernel + aggregation from them) and the backward reads them instead of recomputing (ops.keep_hidden) ...
            kept_h = ops.keep_hidden(ops.csr_for(eib, nb_), [6, kw, kw, 4096], dev)
            prs = fwd_bwd_pairs(4)
            tb_med = sorted(t[1] for t in prs[1:])[1]
            tf_med = sorted(t[0] for t in prs[1:])[1]
            # ... and the recompute form of rounds 2 - 4 (GPDE_SAVE_H_GB=0: nothing but Z goes from the forward to the backward)
            save_h0 = ops.SAVE_H_BYTES
            ops.SAVE_H_BYTES = 0
            try:
                prs_r = fwd_bwd_pairs(4)
            finally:
                ops.SAVE_H_BYTES = save_h0
            tbr_med = sorted(t[1] for t in prs_r[1:])[1]
            tfr_med = sorted(t[0] for t in prs_r[1:])[1]
            eb_ = int(eib.shape[1])
            # executed MFMA work of one backward per edge (3-Linear kernel, split-f16 GEMMs): recompute of H_2 on the forward's
            # kernel (3 x hidden + H1 regeneration), dU_1 and dW_2 (3 x hidden each) and - round 4, gpde_edge_bwd3.hip - the
            # per-edge kernel's two 64 x k2 products (3 MFMAs per product as well) on the f16 pipe; what is left on the fp32
            # pipe is per NODE (dZ = gT . W3, dW_3 = gT^T . Z: 2 x 2 x 64 x 64 x k2 per node; Z comes from the forward: keep-Z)
            kwp = (kw + 127) // 128 * 128
            f16_rec = 3 * 2 * kwp * kwp + 2 * 2 * 16 * kwp * (kwp // 128)         # the recompute's share: hidden GEMM + H1 regeneration
            f16_bwd_r = 3 * (3 * 2 * kwp * kwp) + 2 * 2 * 16 * kwp * (kwp // 128) + 2 * (3 * 2 * 64 * kwp) + \
                (kwp // 256) * 2 * 2 * 16 * kwp          # round 6: H_1 generated inside the dW_2 GEMM (MFMA pair per 32 x 32 block, once per row quad)
            f16_bwd = f16_bwd_r - (f16_rec if kept_h else 0)
            f32_bwd = 2 * (2 * 64 * 64 * kwp) * nb_ / max(int(eib.shape[1]), 1)
            bwd_rate = eb_ / tb_med / 1e12
            trec = None
            tfile = next((f for f in (os.path.join(REPO, "profiles", f"traffic_r06{'k' if kept_h else ''}_bwd.json"),
                                      os.path.join(REPO, "profiles", f"traffic_r05{'k' if kept_h else ''}_bwd.json"),
                                      os.path.join(REPO, "profiles", "traffic_r05_bwd.json"),
                                      os.path.join(REPO, "profiles", "traffic_r04_bwd.json")) if os.path.exists(f)), "")
            if os.path.exists(tfile):
                try:
                    trec = json.load(open(tfile))
                except Exception:       # noqa: BLE001
                    trec = None
            alg_bwd = 40.0 * eb_ + 3 * 256.0 * nb_ + 8.0 * sum(p_.numel() for p_ in conv.parameters())
            backward = {"graph": "g121 (N=%d, E=%d)" % (nb_, eb_), "ms": round(1e3 * tb_med, 2),
                        "M_edges_per_s": round(eb_ / tb_med / 1e6, 2),
                        "hidden_kept_by_forward": bool(kept_h),
                        "training_forward_ms": round(1e3 * tf_med, 2), "pair_ms": round(1e3 * (tf_med + tb_med), 2),
                        "recompute_form": {
                            "ms": round(1e3 * tbr_med, 2), "M_edges_per_s": round(eb_ / tbr_med / 1e6, 2),
                            "training_forward_ms": round(1e3 * tfr_med, 2), "pair_ms": round(1e3 * (tfr_med + tbr_med), 2),
                            "frac_f16_peak": round(f16_bwd_r * eb_ / tbr_med / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
                            "note": "GPDE_SAVE_H_GB=0: the backward of rounds 2 - 4, H_2 recomputed inside it (one more K loop of the hidden "
                                    "layer); `ms` above is the default policy - the forward wrote H_2 (4 KiB per edge) and this backward read it"},
                        "roofline": {
                            "bound": "mfma", "pipe": "f16 MFMA (2-term split operands) for the three k1 x k2 products and the per-edge 64 x k2 products; fp32 MFMA for the per-node dZ / dW_3 products",
                            "executed_flop_per_edge": {"f16_mfma": f16_bwd, "fp32_mfma": f32_bwd},
                            "achieved": round(f16_bwd * bwd_rate, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(f16_bwd * bwd_rate / PEAK_F16_MFMA_TFLOPS, 4),
                            "fp32_mfma_tflops": round(f32_bwd * bwd_rate, 2),
                            "frac_note": "whole backward (all kernels, wall time) against the f16 peak; the fp32-MFMA part adds "
                                         f"{round(f32_bwd * bwd_rate / PEAK_FP32_MFMA_TFLOPS, 4)} of the fp32 peak on top",
                            "algorithmic_bytes": round(alg_bwd),
                            "traffic": None if trec is None else trec.get("hbm_bytes_per_backward"),
                            "traffic_over_algorithmic": None if trec is None or not trec.get("hbm_bytes_per_backward") else
                            round(trec["hbm_bytes_per_backward"] / alg_bwd, 1),
                            "traffic_source": None if trec is None else trec.get("source"),
                            "traffic_by_kernel": None if trec is None else trec.get("kernels")},
                        "grads_finite": bool(torch.isfinite(xb.grad).all()) and
                        all(bool(torch.isfinite(p_.grad).all()) for p_ in conv.parameters()),
                        "arithmetic": "dU_1 and dW_2 GEMMs (gpde_gemm_f16s_nt_kernel) and the per-edge products (gpde_edge_bwd3_kernel) on the "
                                      "2-term f16 split, H_2 kept by the forward (or recomputed on the forward's kernel: recompute_form), "
                                      "per-node products fp32 MFMA",
                        "workspace_GiB": round(ops.bwd_workspace_bytes(_lib.lib(), nb_, eb_, 3, _lib.dims_array([6, kw, kw, 4096]), dev,
                                                                       eb_ * kwp * 4 if kept_h else 0) / 2**30, 1),
                        "kept_hidden_GiB": round(eb_ * kwp * 4 / 2**30, 1) if kept_h else 0.0,
                        "note": "median of 3 timed backward passes after one warm-up; parity of every gradient "
                                "against float64 autograd: tests/test_gpu_bwd.py, tests/test_gpu_parity.py"}
            # the same backward with the opt-in ONE-CHUNK workspace (GPDE_BWD_WS_FRACTION=0.6: the whole s=121 graph as one edge chunk,
            # ~125 GiB) instead of the library's default plan (~26 GB, ten edge chunks: the default since round 6, `ms` above)
            backward["default_workspace_GiB"] = backward["workspace_GiB"]
            try:
                frac0 = ops.BWD_WS_FRACTION
                ops.BWD_WS_FRACTION = 0.6
                td = []
                for it in range(3):
                    conv.zero_grad(set_to_none=True)
                    xb.grad = None
                    yb = conv(xb, eib, eab)
                    lossb = yb.square().mean()
                    torch.cuda.synchronize()
                    tq = time.perf_counter()
                    lossb.backward()
                    torch.cuda.synchronize()
                    td.append(time.perf_counter() - tq)
                backward["one_chunk_workspace_ms"] = round(1e3 * sorted(td[1:])[0], 2)
                backward["one_chunk_workspace_GiB"] = round(ops.bwd_workspace_bytes(_lib.lib(), nb_, eb_, 3, _lib.dims_array([6, kw, kw, 4096]), dev,
                                                                                    eb_ * kwp * 4 if kept_h else 0) / 2**30, 1)
            finally:
                ops.BWD_WS_FRACTION = frac0
            log(f"[bench] backward g121: {backward['ms']} ms, {backward['M_edges_per_s']} M-edges/s; one-chunk workspace (opt-in) {backward.get('one_chunk_workspace_ms')} ms")
            del eib, eab, xb, yb, lossb, prs, prs_r
            # ---- the same at the headline size, and BASELINE config 5's per-GPU unit of work: one training step of the
            #      depth-6 GKN on ONE 241^2 sample (UAI1_full_resolution.py:258-273: forward, L1 loss, backward, Adam).
            #      H of this graph is 391 GB (> HBM): nothing is cached, every layer recomputes its hidden chain.
            if args.config == "g241" and not args.no_g241_train:
                torch.cuda.empty_cache()
                xg = x.detach().clone().requires_grad_(True)
                tg = []
                for it in range(2):
                    conv.zero_grad(set_to_none=True)
                    xg.grad = None
                    yg = conv(xg, ei, ea)
                    lg = yg.square().mean()
                    torch.cuda.synchronize()
                    tq = time.perf_counter()
                    lg.backward()
                    torch.cuda.synchronize()
                    tg.append(time.perf_counter() - tq)
                backward["g241_one_layer"] = {"ms": round(1e3 * tg[-1], 1), "M_edges_per_s": round(e / tg[-1] / 1e6, 2),
                                              "grads_finite": bool(torch.isfinite(xg.grad).all()),
                                              "note": "second of two backward passes of one NNConv call on the headline graph"}
                log(f"[bench] backward g241: {backward['g241_one_layer']['ms']} ms, {backward['g241_one_layer']['M_edges_per_s']} M-edges/s")
                del yg, lg
                fc1 = torch.nn.Linear(6, 64).to(dev)
                fc2 = torch.nn.Linear(64, 1).to(dev)
                params = list(conv.parameters()) + list(fc1.parameters()) + list(fc2.parameters())
                opt = torch.optim.Adam(params, lr=1e-4, weight_decay=5e-4)
                a_in = torch.randn(n, 6, device=dev)
                y_t = torch.randn(n, device=dev)
                depth = 6
                del xg
                # the module's DEFAULT policy (hidden cache `auto`, deferred backward `auto`): H of this graph (391 GB) does
                # not fit, so the applications share a virtual-H node - light backward per application, ONE deferred pass for
                # the hidden layers - and read the part of H that does fit.  Step 1 still sees the first
                # application as a stranger (own full backward); from step 2 on all six hang on the shared node.
                hidden_cache.MODE = "auto"
                hidden_cache.clear()
                torch.cuda.empty_cache()
                t_steps = []
                for it in range(3):
                    torch.cuda.synchronize()
                    tq = time.perf_counter()
                    opt.zero_grad(set_to_none=True)
                    hcur = fc1(a_in)
                    for _ in range(depth):
                        hcur = torch.relu(conv(hcur, ei, ea))
                    loss_t = torch.norm(fc2(hcur).view(-1) - y_t, 1)
                    loss_t.backward()
                    opt.step()
                    torch.cuda.synchronize()
                    t_steps.append(time.perf_counter() - tq)
                t_step = min(t_steps[1:])
                ent_t = hidden_cache._entries.get(conv)
                backward["g241_depth6_train_step"] = {
                    "s": round(t_step, 2), "first_step_s": round(t_steps[0], 2), "steps_s": [round(t_, 2) for t_ in t_steps],
                    "M_edge_applications_per_s": round(depth * e / t_step / 1e6, 2),
                    "loss_finite": bool(torch.isfinite(loss_t)),
                    "deferred_backward": {k: hidden_cache.stats.get(k, 0) for k in ("deferred_builds", "deferred_hits", "builds", "hits")},
                    "nodes_served_from_partial_H": 0 if ent_t is None or ent_t.hidden is None else ent_t.hn, "nodes": n,
                    "peak_GiB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                    "note": "three consecutive steps of fc1 + 6 x relu(NNConv_old) + fc2, L1 loss, backward, Adam (UAI1_full_resolution.py:"
                            "258-273) - BASELINE config 5's unit of work per GPU and sample; `s` = the faster of steps 2 and 3 (steady "
                            "state: all six applications on the shared virtual-H node), `first_step_s` includes the cold start; "
                            "round 3 ran every application's own full backward: 18.98 s"}
                log(f"[bench] g241 depth-6 train step: first {t_steps[0]:.2f} s, steady {t_step:.2f} s")
                # ---- the same step over three DISTINCT samples, as an epoch sees them (UAI1_full_resolution.py:258-273: every batch
                # is another sample): a NEW edge_index tensor (the loader's) and new edge attributes per step, so that the dst-CSR
                # build, the slot-order gather of the attributes and the re-keying of every cache are INSIDE the step time
                del hcur, loss_t
                try:
                    t_dist = []
                    pos = synth.lattice_positions(s, dev)
                    for it in range(3):
                        a_s = synth.darcy_coefficient(s, seed=100 + it).to(dev)
                        ea_s = synth.darcy_edge_attr(ei, pos, a_s)             # this sample's [E, 6] attributes
                        ei_s = ei.clone()                                       # a fresh index tensor, as a DataLoader hands over
                        a_in_s = torch.randn(n, 6, device=dev)
                        torch.cuda.synchronize()
                        tq = time.perf_counter()
                        opt.zero_grad(set_to_none=True)
                        hcur = fc1(a_in_s)
                        for _ in range(depth):
                            hcur = torch.relu(conv(hcur, ei_s, ea_s))
                        loss_s = torch.norm(fc2(hcur).view(-1) - y_t, 1)
                        loss_s.backward()
                        opt.step()
                        torch.cuda.synchronize()
                        t_dist.append(time.perf_counter() - tq)
                        del hcur, ei_s, ea_s
                    backward["g241_depth6_train_step"]["distinct_samples"] = {
                        "steps_s": [round(t_, 2) for t_ in t_dist], "s": round(statistics.median(t_dist), 2),
                        "loss_finite": bool(torch.isfinite(loss_s)),
                        "note": "three steps on three different samples (new edge_index tensor and new edge_attr each): graph / attribute "
                                "preparation and cache re-keying inside the time"}
                    log(f"[bench] g241 depth-6 train step, distinct samples: {[round(t_, 2) for t_ in t_dist]} s")
                    del loss_s
                except Exception as ex:       # noqa: BLE001
                    backward["g241_depth6_train_step"]["distinct_samples"] = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}
                del opt
        finally:
            hidden_cache.MODE = mode0
            hidden_cache.clear()
        torch.cuda.empty_cache()

    def pick(dct, *keys):
        return None if not dct else {k_: dct.get(k_) for k_ in keys}

    g241t = None if not backward else backward.get("g241_depth6_train_step")
    bwd_roof = {} if not backward else backward.get("roofline", {})
    # ---- everything measured (per-kernel counters, notes, every leg's full object): bench_detail.json beside the line
    detail = {
        "metric": "M-edges/s through fused NNConv fwd (width=64)", "value": round(value, 3), "unit": "M-edges/s",
        "precision": precision, "median_step_ms": round(med, 3), "value_at_median": round(world * e / med / 1e3, 3),
        "config": {"workload": f"GKN Darcy-2D {s}x{s} lattice radius graph r={r} N={n} E={e} per sample; NNConv_old fwd "
                               f"width=64; kernel MLP 6-{kw}-{kw}-4096; aggr=mean root+bias; one sample per GPU",
                   "graph": args.config, "edges_per_sample": e, "nodes_per_sample": n, "plan": plan},
        "graph_prep": graph_prep, "rel_l2_sample": rel, "alt_precision": alt, "node_table_attributes": nodeattr,
        "roofline": roofline, "cpu_baseline": cpu, "mgkn": mgkn, "depth_reuse": reuse, "backward": backward,
    }
    detail_path = args.detail_out or os.path.join(os.getcwd(), "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(detail_path) or ".", exist_ok=True)
        with open(detail_path, "w") as fh:
            json.dump(detail, fh, indent=1)
    except OSError as ex:
        detail_path = f"(not written: {ex})"
    # ---- the ONE JSON line, <= 2.5 KB: the contract's keys, a compact roofline / cpu_baseline, and every other leg's headline figure
    # (the driver keeps key names + a 2-3 KB tail - a 20 KB line hid half of what it measured)
    line = {
        "metric": "M-edges/s through fused NNConv fwd (width=64)",
        "value": round(value, 3), "unit": "M-edges/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if precision == "f32" else "f32 (hidden layer + aggregation: 2-term f16-split MFMA, f32 accumulate)",
        "data": "synthetic",
        "config": {"workload": f"GKN Darcy-2D {s}x{s} radius graph r={r} N={n} E={e}; NNConv_old fwd width=64; kernel MLP 6-{kw}-{kw}-4096; "
                               "aggr=mean root+bias; one sample per GPU", "graph": args.config},
        "rccl_ranks": world if use_dist else 0,
        "rel_l2_sample": None if rel is None else float(f"{rel:.3e}"),
        "roofline": None if not roofline else {k_: roofline.get(k_) for k_ in (
            "kernel", "bound", "achieved", "peak", "unit", "frac", "frac_of_sustained", "traffic", "algorithmic_bytes_per_launch",
            "traffic_over_algorithmic", "avg_launch_ms", "launches_per_step", "hbm_frac")},
        "cpu_baseline": None if not cpu else dict(pick(cpu, "value", "unit", "cores", "kind"), sample=str(cpu.get("sample", ""))[:96]),
        "summary": {
            "graph_prep_ms": pick(graph_prep, "csr_build_ms", "attr_reorder_ms", "attr_reorder_next_sample_ms"),
            "node_table_M_edges_s": None if not nodeattr else nodeattr.get("M_edges_per_s", nodeattr.get("value")),
            "mgkn_fwd_ms": None if not mgkn else {k_[5:12]: [v_.get("ms_per_forward"), v_.get("ms_per_forward_captured"), v_.get("ms_per_forward_grouped")]
                                                  for k_, v_ in mgkn.items()},
            "mgkn_train_ms": None if not mgkn else {k_[5:12]: [v_.get("train_step_ms"), v_.get("train_step_captured_ms")] for k_, v_ in mgkn.items()},
            "mgkn_keys": "[unmodified calls, captured, grouped] / [direct, captured]; depth6: [direct, shared H]",
            "bwd_g121": None if not backward else {
                "ms": backward.get("ms"), "M_edges_s": backward.get("M_edges_per_s"), "fwd_ms": backward.get("training_forward_ms"),
                "pair_ms": backward.get("pair_ms"), "frac": bwd_roof.get("frac"), "traffic": bwd_roof.get("traffic"),
                "traffic_over_algorithmic": bwd_roof.get("traffic_over_algorithmic"),
                "recompute_form": pick(backward.get("recompute_form"), "ms", "pair_ms", "frac_f16_peak"),
                "one_chunk_ms": backward.get("one_chunk_workspace_ms"), "workspace_GiB": backward.get("default_workspace_GiB")},
            "bwd_g241_one_layer": None if not backward else pick(backward.get("g241_one_layer"), "ms", "M_edges_per_s"),
            "g241_depth6_train_step": None if not g241t else {
                "s": g241t.get("s"), "first_s": g241t.get("first_step_s"), "distinct_samples_s": g241t.get("distinct_samples", {}).get("s"),
                "peak_GiB": g241t.get("peak_GiB"), "M_edge_applications_s": g241t.get("M_edge_applications_per_s")},
            "depth6_s61_ms": None if not reuse else {"fwd": [reuse.get("forward_ms", {}).get("direct"), reuse.get("forward_ms", {}).get("reuse")],
                                                     "fwd_bwd": [reuse.get("forward_backward_ms", {}).get("direct"), reuse.get("forward_backward_ms", {}).get("reuse")]},
            "depth6_g241_fwd_ms": None if not reuse or "g241_depth6_forward" not in reuse else
            [reuse["g241_depth6_forward"].get("direct_ms"), reuse["g241_depth6_forward"].get("reuse_ms")],
        },
        "detail": detail_path,
    }
    txt = json.dumps(line, separators=(",", ":"))
    if len(txt) > 2500:                                  # never let the line outgrow the driver's tail: drop the least essential
        for k_ in ("depth6_s61_ms", "depth6_g241_fwd_ms", "mgkn_keys", "node_table_M_edges_s", "bwd_g241_one_layer"):
            line["summary"].pop(k_, None)
            txt = json.dumps(line, separators=(",", ":"))
            if len(txt) <= 2500:
                break
    print(txt, flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

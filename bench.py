#!/usr/bin/env python3
"""Benchmark of the fused NNConv forward on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config g241|g61|g16] [--kernel-width 1024]

One "step" = one `conv(x, edge_index, edge_attr)` forward of the headline operator on one PDE
sample: the Darcy-241^2 r=0.10 lattice radius graph (N = 58,081 nodes, E = 95,539,625 edges,
BASELINE.json configs[1]) with the kernel MLP DenseNet([6,1024,1024,4096]), width 64, aggr='mean',
root + bias (UAI1_full_resolution.py:21,56-59).  Inputs are synthetic of that shape (SURVEY.md
§8d), resident in HBM and with the destination CSR already built (its build time is printed to
stderr) when the timed region starts.  N > 1: one process per GPU (torch.distributed / RCCL for
the barrier only), each rank runs its own independent sample: weak scaling, no data-path
collective.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  precision     "f16split" (default): the k1 x k2 hidden layer runs on f16 MFMA with two-term
                operand splitting and fp32 accumulation (error class of fp32);
                "f32": every contraction on fp32 MFMA (exact fmaf chains).  `alt_precision`
                carries the other mode's throughput and the distance between the two outputs.
  roofline      dominant kernel = gpde_fused_*kernel (edge MLP + outer-product aggregation).
                bound 'mfma' (fp32 MFMA, 157.3 TFLOP/s): the path is compute-bound.
                achieved = ALGORITHMIC FLOPs of the reference formulation (10,506,304 FLOP/edge at
                1024^2) x edges per launch / average launch duration (HIP events recorded inside
                libgpde.so on the kernel's own stream).  The kernel re-associates the last layer
                 and executes ~4.4x fewer FLOPs, so `frac` can exceed 1;
                `frac_executed` is executed FLOPs / peak.  The HBM view BASELINE.json asks for is
                in `hbm_*` (40.5 algorithmic B/edge vs 8 TB/s).
  cpu_baseline  the CPU oracle (plain-PyTorch restatement of the reference path, kind "port")
                timed on this host's cores on a bounded sample: all in-edges of a stratified
                subset of destination rows of the same graph; the same rows give `rel_l2_sample`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

CONFIGS = {
    # name: (s, r)   BASELINE.json configs[1] is g241; g61/g16 are quick-look sizes
    "g241": (241, 0.10),
    "g121": (121, 0.10),
    "g61": (61, 0.10),
    "g16": (16, 0.15),
}
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md
PEAK_F16_MFMA_TFLOPS = 2500.0      # dense, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_flops_per_edge(dims, w=64):
    """the work the reference performs per edge (nn_conv.py:274-275)."""
    f = 0
    for i in range(len(dims) - 1):
        f += 2 * dims[i] * dims[i + 1]
    return f + 2 * w * w + w


def executed_flops_per_edge(dims, w=64, precision="f32"):
    """MFMA FLOPs the fused kernel issues per edge for a 3-Linear MLP, as
    (fp32-MFMA FLOPs, f16-MFMA FLOPs): H1 generation (K padded to 8, repeated per 128-column
    slice) and the 64 x k2 outer product are always fp32 MFMA; the k1 x k2 layer is fp32 MFMA
    ("f32") or 3 f16 MFMAs per product ("f16split")."""
    k1p = (dims[1] + 31) // 32 * 32
    k2p = (dims[2] + 127) // 128 * 128
    agg = 2 * w * k2p
    hidden = 2 * k1p * k2p
    if precision == "f32":
        return (2 * 8 * k1p * (k2p // 128) + agg + hidden, 0)
    if precision == "f16split4w":            # 4-wave kernel: H1 on fp32 MFMA per 128-column slice
        return (2 * 8 * k1p * (k2p // 128) + agg, 3 * hidden)
    # default 8-wave kernel: H1 as 2 f16 MFMAs (K = 16) per 32-row chunk and 64-column wave tile
    h1 = 2 * 2 * 16 * k1p * (k2p // 64)
    if precision == "f16split_agg32":        # aggregation kept on fp32 MFMA
        return (agg, 3 * hidden + h1)
    # aggregation on split f16 as well (default from 32768 edges on)
    return (0, 3 * hidden + h1 + 3 * agg)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="g241", choices=sorted(CONFIGS))
    ap.add_argument("--kernel-width", type=int, default=1024)
    ap.add_argument("--cpu-rows", type=int, default=512, help="destination rows of the CPU sample")
    ap.add_argument("--cpu-threads", type=int, default=64, help="threads of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reuse-probe", action="store_true",
                    help="skip the depth x module probe of the cross-depth hidden-activation reuse")
    ap.add_argument("--precision", default=None, choices=["f32", "f16split", "f16split_8wave", "f16split_agg16", "f16split_agg32"],
                    help="arithmetic of the hidden layer (default: graph_pde_amd.ops.DEFAULT_PRECISION)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            log(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus}`")
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

    import graph_pde_amd as gp
    from graph_pde_amd import _lib, ops, synth

    s, r = CONFIGS[args.config]
    kw = args.kernel_width
    dims = [6, kw, kw, 4096]
    torch.manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(6, kw), torch.nn.ReLU(), torch.nn.Linear(kw, kw),
                              torch.nn.ReLU(), torch.nn.Linear(kw, 4096))
    conv = gp.NNConv_old(64, 64, mlp, aggr="mean").to(dev)          # weights: seed 0, replicated

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
    if rank == 0:
        log(f"[bench] graph {args.config}: N={n} E={e} generated in {t1 - t0:.2f}s; "
            f"dst-CSR build {1e3 * (t2 - t1):.1f} ms (not in the timed region)")

    lin = ops.mlp_linears(conv.nn)
    pm = ops.pack_mlp([l.weight for l in lin], [l.bias for l in lin])
    ws = torch.empty(ops.workspace_bytes(n, e, pm), dtype=torch.uint8, device=dev)
    out = torch.empty(n, 64, dtype=torch.float32, device=dev)
    plan = ops.launch_plan(n, e, pm, ws.numel())

    precision = args.precision or ops.DEFAULT_PRECISION

    def step():
        ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", out=out, ws=ws,
                               precision=precision)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    lib = _lib.lib()
    lib.gpde_profile_begin()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t_start
    import ctypes
    fused_ms, launches, other_ms = ctypes.c_double(), ctypes.c_int32(), ctypes.c_double()
    lib.gpde_profile_end(ctypes.byref(fused_ms), ctypes.byref(launches), ctypes.byref(other_ms))
    if use_dist:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank != 0:
        dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * e / (elapsed / args.steps) / 1e6                  # M-edges/s, whole job

    # ---- roofline of the dominant kernel (rank 0's launches) --------------------------------------
    f_alg = algorithmic_flops_per_edge(dims)
    f_exe32, f_exe16 = executed_flops_per_edge(dims, precision=precision)
    n_launch = max(int(launches.value), 1)
    avg_launch_ms = fused_ms.value / n_launch
    edges_per_launch = e * args.steps / n_launch
    achieved_tf = f_alg * edges_per_launch / (avg_launch_ms * 1e-3) / 1e12
    executed_tf = f_exe32 * edges_per_launch / (avg_launch_ms * 1e-3) / 1e12
    executed_tf16 = f_exe16 * edges_per_launch / (avg_launch_ms * 1e-3) / 1e12
    bytes_per_edge = (40.0 * e + 512.0 * n + 4.0 * sum(p.numel() for p in conv.parameters())) / e
    edges_per_s_rank = e / (elapsed / args.steps)
    traffic = None
    tpath = os.path.join(REPO, "profiles", "traffic_r01.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("config") == args.config and tj.get("kernel_width") == kw:
                traffic = tj.get("fused_hbm_bytes_per_launch", {}).get(precision)
        except Exception:
            traffic = None
    roofline = {
        "kernel": {"f16split": "gpde_fused_f16v3_kernel", "f16split4w": "gpde_fused_f16_kernel"}.get(
            precision, "gpde_fused_kernel<1>"),
        "bound": "mfma",
        "achieved": round(achieved_tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved_tf / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
        "avg_launch_ms": round(avg_launch_ms, 3), "launches_per_step": n_launch // args.steps,
        "algorithmic_flop_per_edge": f_alg,
        "executed_f32_mfma_flop_per_edge": f_exe32, "executed_f16_mfma_flop_per_edge": f_exe16,
        "executed_f32_mfma_tflops": round(executed_tf, 2),
        "executed_f16_mfma_tflops": round(executed_tf16, 2),
        # share of the matrix pipe's time the executed MFMAs need at peak rate
        "frac_executed": round(executed_tf / PEAK_FP32_MFMA_TFLOPS + executed_tf16 / PEAK_F16_MFMA_TFLOPS, 4),
        "fused_share_of_step": round(fused_ms.value / (1e3 * elapsed), 4),
        "node_kernels_ms_per_step": round(other_ms.value / args.steps, 3),
        "hbm_algorithmic_bytes_per_edge": round(bytes_per_edge, 2),
        "hbm_achieved_GBs": round(edges_per_s_rank * bytes_per_edge / 1e9, 2),
        "hbm_frac": round(edges_per_s_rank * bytes_per_edge / 1e9 / PEAK_HBM_GBS, 6),
    }

    # ---- the other arithmetic on the same inputs, one step, for reference ---------------------------
    alt = None
    if world == 1:
        other = "f32" if precision == "f16split" else "f16split"
        out_main = out.clone()
        ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", out=out, ws=ws, precision=other)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        ops.nnconv_forward_raw(x, csr, ea, pm, conv.root, conv.bias, "mean", out=out, ws=ws, precision=other)
        torch.cuda.synchronize()
        tb = time.perf_counter() - ta
        d = (out.double() - out_main.double()).norm() / out.double().norm()
        alt = {"precision": other, "value": round(e / tb / 1e6, 3), "unit": "M-edges/s",
               "rel_l2_between_precisions": float(d)}
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
        run = lambda: nnconv_forward(x_c, ei_s, ea_s, ws_c, bs_c, root_c, bias_c, aggr="mean",
                                     dtype=torch.float32, chunk_edges=65536)
        # warm-up on a slice, then one timed pass over the whole sample
        nnconv_forward(x_c, ei_s[:, :65536], ea_s[:65536], ws_c, bs_c, root_c, bias_c, aggr="mean",
                       chunk_edges=65536)
        tc = time.perf_counter()
        y_cpu = run()
        tcpu = time.perf_counter() - tc
        es = int(ei_s.shape[1])
        cpu = {"value": round(es / tcpu / 1e6, 5), "unit": "M-edges/s", "cores": ncores,
               "kind": "port",
               "sample": f"all {es} in-edges of {len(rows)} stratified destination rows of the same "
                         f"graph, plain-PyTorch fp32 oracle, {ncores} threads, {tcpu:.1f}s"}
        rel = rel_l2(out[rows.to(dev)].cpu(), y_cpu[rows])
        log(f"[bench] CPU oracle sample: {es} edges in {tcpu:.2f}s; rel-L2 GPU vs CPU rows = {rel:.3e}")

    # ---- cross-depth reuse: depth applications of ONE module, as KernelNN.forward does
    #      (UAI1_full_resolution.py:29-30), on the reference's own training resolution s=61 r=0.10 -- the
    #      headline graph's hidden activations (E x 4 KiB = 391 GB) do not fit one GPU.  Not part of `value`.
    reuse = None
    if not args.no_reuse_probe and world == 1:
        from graph_pde_amd import hidden_cache
        del ws
        torch.cuda.empty_cache()
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
            for p_ in conv.parameters():
                p_.data.mul_(1.0)                       # new weight version: as after an optimiser step
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

    line = {
        "metric": "M-edges/s through fused NNConv fwd (width=64)",
        "value": round(value, 3), "unit": "M-edges/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if precision == "f32" else "f32 (hidden layer + aggregation: 2-term f16-split MFMA, f32 accumulate)",
        "precision": precision, "data": "synthetic",
        "config": {"workload": f"GKN Darcy-2D {s}x{s} lattice radius graph r={r} "
                               f"(N={n}, E={e} per sample), NNConv_old fwd width=64, kernel MLP "
                               f"[6,{kw},{kw},4096], aggr=mean, root+bias; one sample per GPU",
                   "graph": args.config, "edges_per_sample": e, "nodes_per_sample": n,
                   "plan": plan},
        "rel_l2_sample": rel,
        "alt_precision": alt,
        "depth_reuse": reuse,
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

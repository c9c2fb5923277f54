#!/usr/bin/env python3
"""Run an UNMODIFIED script of the reference (neuraloperator/graph-pde) on top of the MI355X operator.

    python scripts/run_reference_script.py UAI1_full_resolution.py --set ntrain=2 --set ntest=2 --set epochs=1

What this does, and nothing else:
  * puts graph-pde_amd/shims on sys.path AHEAD of the script's own directory, so that `nn_conv`,
    `torch_geometric` and `h5py` resolve to the shims (neither PyG nor h5py is installable here,
    SURVEY.md Appendix A) while `utilities` stays the reference's own file;
  * creates a scratch working directory with the `data/ model/ results/ image/` folders the scripts
    expect and SYNTHETIC .mat files of the shapes they read (the datasets are not available offline);
  * executes the script's bytes as they are (runpy).  `--set name=value` overrides a module-level
    hyper-parameter WITHOUT editing the file: a line tracer on the script's module frame re-imposes the
    value after the script's own assignment has run (ntrain / ntest / epochs, so that a plumbing run
    takes a minute instead of days).
The script is looked up under /root/reference (build container) or oracle/_ref (staged there by
scripts/stage_reference.py for a GPU-box run and removed afterwards; git-ignored, travels with the gpurun snapshot).
"""
from __future__ import annotations

import argparse
import ast
import os
import runpy
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIMS = os.path.join(REPO, "graph-pde_amd", "shims")
PROJECTS = ["graph-neural-operator", "multipole-graph-neural-operator"]
ROOTS = ["/root/reference", os.path.join(REPO, "oracle", "_ref")]


def find_script(name: str) -> str:
    if os.path.isabs(name) and os.path.exists(name):
        return name
    for root in ROOTS:
        for proj in PROJECTS:
            p = os.path.join(root, proj, name)
            if os.path.exists(p):
                return p
    raise FileNotFoundError(f"{name}: not under {ROOTS} (run scripts/stage_reference.py where /root/reference exists)")


def smooth_field(rng, n, s, sigma):
    from scipy.ndimage import gaussian_filter
    return np.stack([gaussian_filter(rng.standard_normal((s, s)), sigma=sigma, mode="wrap") for _ in range(n)])


def make_darcy_mat(path, n, s, seed):
    """Fields read by UAI*.py / MGKN_general_darcy2d.py: coeff, Kcoeff, Kcoeff_x, Kcoeff_y, sol [N,s,s]."""
    import scipy.io
    rng = np.random.default_rng(seed)
    g = smooth_field(rng, n, s, s / 16)
    coeff = np.where(g > 0, 12.0, 3.0)
    from scipy.ndimage import gaussian_filter
    kc = np.stack([gaussian_filter(c, sigma=2.0) for c in coeff])
    kx = np.gradient(kc, axis=2) * (s - 1)
    ky = np.gradient(kc, axis=1) * (s - 1)
    sol = smooth_field(rng, n, s, s / 8) * 0.05 + 0.01
    scipy.io.savemat(path, {"coeff": coeff, "Kcoeff": kc, "Kcoeff_x": kx, "Kcoeff_y": ky, "sol": sol})


def make_burgers_mat(path, n, s, seed):
    """Fields read by MGKN_orthogonal_burgers1d.py: a, u [N, 8192]."""
    import scipy.io
    from scipy.ndimage import gaussian_filter1d
    rng = np.random.default_rng(seed)
    a = np.stack([gaussian_filter1d(rng.standard_normal(s), sigma=s / 64, mode="wrap") for _ in range(n)])
    u = np.stack([gaussian_filter1d(x, sigma=s / 32, mode="wrap") for x in a])
    scipy.io.savemat(path, {"a": a, "u": u})


def prepare_workdir(script: str, workdir: str, n_samples: int):
    for d in ("data", "model", "results", "image"):
        os.makedirs(os.path.join(workdir, d), exist_ok=True)
    # which data set a script reads is in its source: `burgers_data_R10.mat` (a, u [N, 8192]), `piececonst_r241_...` or
    # `piececonst_r421_...` (literally, or as 'piececonst_r'+str(s0) with s0 = 421: UAI7_evaluate*.py:38-41)
    with open(script) as fh:
        src = fh.read()
    if "TRAIN_PATH = 'data/burgers_data_R10.mat'" in src:
        make_burgers_mat(os.path.join(workdir, "data", "burgers_data_R10.mat"), n_samples, 8192, 0)
        return
    res = 421 if ("piececonst_r421" in src and "TRAIN_PATH = 'data/piececonst_r421" in src) or "s0 = 421" in src else 241
    for i in (1, 2):
        make_darcy_mat(os.path.join(workdir, "data", f"piececonst_r{res}_N1024_smooth{i}.mat"), n_samples, res, i)


def install_torch_drift_compat():
    """The 2020-era scripts rely on behaviour of the torch they were written for.  One case breaks a run on
    torch 2.x whatever NNConv implementation is underneath: MGKN_general_darcy2d.py:80-84 does `x = F.relu(x)`
    and then writes slices of x in place (`x[a:b] = conv(...)`).  Old torch differentiated relu from its INPUT
    (derivatives.yaml: threshold_backward(grad, self, 0)), current torch saves the OUTPUT, so the slice write
    invalidates the saved tensor ("modified by an inplace operation").  The environment shim below restores
    the old contract for F.relu only (gradient = grad * (input > 0), identical values); the script's bytes
    stay as they are.  Elementwise glue outside the NNConv hot path."""
    import torch
    import torch.nn.functional as F

    class _ReluSavesInput(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            ctx.save_for_backward(x)
            return x.clamp_min(0)

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            return g * (x > 0).to(g.dtype)

    def relu(input, inplace=False):
        if inplace or not (torch.is_grad_enabled() and input.requires_grad):
            return _orig(input, inplace=inplace)
        return _ReluSavesInput.apply(input)

    _orig = F.relu
    if getattr(_orig, "_gpde_compat", False):
        return
    relu._gpde_compat = True
    F.relu = relu

    # Second drift, same script (MGKN_general_darcy2d.py:304-318, utilities.py:91): after `u_normalizer.cpu()` a
    # CPU tensor is indexed with a CUDA index tensor (`self.std[sample_idx]`, `test_u[i, data.sample_idx]`).  The
    # torch of 2020 moved such an index to the host implicitly; torch 2.x raises.  Restore the old behaviour.
    _getitem = torch.Tensor.__getitem__

    def _to_host(i):
        return i.cpu() if isinstance(i, torch.Tensor) and i.is_cuda else i

    def getitem(self, idx):
        if not self.is_cuda:
            idx = tuple(_to_host(i) for i in idx) if isinstance(idx, tuple) else _to_host(idx)
        return _getitem(self, idx)

    torch.Tensor.__getitem__ = getitem

    # Third drift (graph-neural-operator/utilities.py:607, DownsampleGridSplitter.sample, reached by UAI7_evaluate.py:140):
    # `torch.tensor([x, y])` with x, y one-element tensors from `torch.randint(0, r, (1,))`.  The torch of 2020 read such a list
    # as two scalars; torch 2.x walks into the elements and raises "len() of a 0-d tensor".  Restore the old reading.
    _tensor = torch.tensor

    def tensor(data, *a, **k):
        if isinstance(data, (list, tuple)) and data and all(isinstance(t, torch.Tensor) and t.numel() == 1 for t in data):
            data = [t.item() for t in data]
        return _tensor(data, *a, **k)

    torch.tensor = tensor


def install_cpu_dryrun():
    """Developer aid (build container, no GPU): `.cuda()` / `.to('cuda')` become no-ops so that a script's data pipeline, batch
    collation, splitters and model classes can be exercised on the HOST with the stock-torch composite (--composite is implied).
    Never used by a test: the tests run the scripts on the MI355X."""
    import torch

    def ident(self, *a, **k):
        return self
    torch.Tensor.cuda = ident
    torch.nn.Module.cuda = ident
    _t_to, _m_to = torch.Tensor.to, torch.nn.Module.to

    def fix(a):
        if isinstance(a, torch.device) and a.type == "cuda":
            return torch.device("cpu")
        if isinstance(a, str) and a.startswith("cuda"):
            return "cpu"
        return a
    torch.Tensor.to = lambda self, *a, **k: _t_to(self, *[fix(x) for x in a], **{k_: fix(v) for k_, v in k.items()})
    torch.nn.Module.to = lambda self, *a, **k: _m_to(self, *[fix(x) for x in a], **{k_: fix(v) for k_, v in k.items()})


def run(script: str, overrides: dict, workdir: str | None = None, n_samples: int | None = None, seed: int | None = 0,
        composite: bool = False) -> dict:
    """`seed`: torch / numpy / random are seeded before the script starts (the scripts seed nothing themselves), so two runs see
    the same initial weights, sample order and random sub-graphs.  `composite`: the modules' forward is replaced by the
    stock-torch-ops composite of tests/helpers/composite_nnconv.py - the OTHER arm of the script-level parity test."""
    script = find_script(script)
    install_torch_drift_compat()
    import torch
    torch.set_printoptions(precision=8)            # the scripts print CPU tensors (`tensor(0.0597)`): enough digits to compare runs
    if seed is not None:
        import random
        torch.manual_seed(seed)
        np.random.seed(seed)
        random.seed(seed)
    if composite:
        if REPO not in sys.path:
            sys.path.insert(0, REPO)
        import graph_pde_amd  # noqa: F401
        from tests.helpers import composite_nnconv
        run.composite_counter = composite_nnconv.install()
    if n_samples is None:
        n_samples = max([int(v) for k, v in overrides.items() if k in ("ntrain", "ntest")] + [2])
    own_tmp = None
    if workdir is None:
        own_tmp = tempfile.TemporaryDirectory(prefix="gpde_refrun_")
        workdir = own_tmp.name
    prepare_workdir(script, workdir, n_samples)
    os.environ.setdefault("MPLBACKEND", "Agg")
    old_path, old_cwd, old_argv = list(sys.path), os.getcwd(), list(sys.argv)
    sys.path[:] = [SHIMS, os.path.dirname(script)] + [p for p in old_path if p not in (SHIMS, os.path.dirname(script))]
    os.chdir(workdir)
    sys.argv = [script]
    target = os.path.realpath(script)

    def local_tracer(frame, event, arg):
        g = frame.f_globals
        for k, v in overrides.items():
            if k in g and g[k] != v:
                g[k] = v
        return local_tracer

    def global_tracer(frame, event, arg):
        code = frame.f_code
        if code.co_name == "<module>" and os.path.realpath(code.co_filename) == target:
            return local_tracer
        return None

    try:
        if overrides:
            sys.settrace(global_tracer)
        return runpy.run_path(script, run_name="__main__")
    finally:
        sys.settrace(None)
        os.chdir(old_cwd)
        sys.path[:] = old_path
        sys.argv = old_argv
        if own_tmp is not None:
            own_tmp.cleanup()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("script")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE")
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--samples", type=int, default=None, help="samples in the synthetic .mat files")
    ap.add_argument("--seed", type=int, default=0, help="torch / numpy / random seed set before the script starts")
    ap.add_argument("--cpu-dryrun", action="store_true", help="developer aid: no GPU - .cuda() is a no-op, composite operator on the host")
    ap.add_argument("--composite", action="store_true",
                    help="test harness: run the script on the stock-torch-ops composite of the operator instead of libgpde.so")
    args = ap.parse_args()
    overrides = {}
    for kv in args.set:
        k, v = kv.split("=", 1)
        overrides[k] = ast.literal_eval(v)
    if args.cpu_dryrun:
        install_cpu_dryrun()
        args.composite = True
    ns = run(args.script, overrides, args.workdir, args.samples, args.seed, args.composite)
    print(f"[run_reference_script] {os.path.basename(args.script)} finished; overrides {overrides}")
    from graph_pde_amd import _lib
    print(f"[run_reference_script] native libgpde.so calls: {_lib.n_native_calls}")
    if args.composite:
        print(f"[run_reference_script] composite forward calls: {run.composite_counter['calls']}")
    return ns


if __name__ == "__main__":
    main()
